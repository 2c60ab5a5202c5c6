"""lslam_map_set build time against cloud size (random points on noisy planes); run once per setting of
LSLAM_NO_LEVEL_BUILD to find where the level-synchronous phase starts to pay."""
import importlib, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pkg = importlib.import_module("the-cooper-mapper_amd")
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
out = []
for n in [int(v) for v in os.environ.get("SIZES", "4000,8000,16000,32000,64000,128000,256000,512000").split(",")]:
    pts = np.zeros((n, 4), np.float32)
    pts[:, :2] = rng.uniform(-60, 60, (n, 2))
    pts[:, 2] = rng.normal(0, 0.02, n)
    small = pts[:200]
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.map_set(small, pts)
        best = min(best, time.perf_counter() - t0)
    out.append("%d: %.2f ms (build %.2f)" % (n, 1e3 * best, ctx.map_info().build_ms))
print("; ".join(out))
