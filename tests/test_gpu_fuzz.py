"""Seeded parity fuzz (tools/fuzz_parity.py): random worlds / ring counts / poses, voxel-filtered and 1-cm-rounded maps
(exact distance ties), both search implementations -- sweep taps bit for bit against the oracle, whole loops within
the north-star bar.  Nine problems here; run the tool itself with N_SEEDS=... for more."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_parity_fuzz_nine_problems():
    env = dict(os.environ, N_SEEDS="9")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "9 problems, no mismatch" in r.stdout


@pytest.mark.gpu
def test_posegraph_solver_forms_fuzz_twenty_graphs():
    """tools/fuzz_posegraph.py: random graphs from 2 keyframes up (hubs, aggregates of one member, one workgroup only) --
    the persistent kernels against the launch loop, and a persistent run against itself bit for bit."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_posegraph.py"), "20"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "pose-graph fuzz: 20 graphs" in r.stdout


@pytest.mark.gpu
def test_map_maintenance_fuzz_twenty_maps():
    """tools/fuzz_fmap.py: random cube grids / leaves / walks / clumped clouds through FeatureMap and VoxelGrid, bit for bit
    against the oracle; both forms of addFeatureCloud's rebuild (new points merged in, everything re-sorted) must occur."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_fmap.py"), "20"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "20 maps, no mismatch" in r.stdout
    import re
    m = re.search(r"(\d+) merged, (\d+) re-sorted", r.stdout)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0
