"""Phase-B kernel time against subtree size: trees of n random points (run under rocprofv3 --kernel-trace
and read the kd_build_small_kernel durations in order; two trees per size, corner then surf)."""
import importlib, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pkg = importlib.import_module("the-cooper-mapper_amd")
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
for n in [int(v) for v in os.environ.get("SIZES", "40,64,100,128,256,418,512,1024,1536").split(",")]:
    pts = np.zeros((n, 4), np.float32)
    pts[:, :3] = rng.uniform(-20, 20, (n, 3))
    for _ in range(2):
        ctx.map_set(pts, pts)
