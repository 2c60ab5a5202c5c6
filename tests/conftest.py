import glob
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The library reads the environment once, when the first context is created; the per-call test hooks (LSLAM_DEBUG_NODE_CAP_DIV,
# LSLAM_DEBUG_SPIN_LIMIT, LSLAM_DEBUG_PG_ABORT, LSLAM_HUGE_MIN ...) are looked at only in a process started with this set
os.environ.setdefault("LSLAM_DEBUG_HOOKS", "1")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory has a hyphen in its name: import it via importlib and
    register the alias cooper_mapper_amd."""
    if "cooper_mapper_amd" not in sys.modules:
        pkg = importlib.import_module("the-cooper-mapper_amd")
        sys.modules["cooper_mapper_amd"] = pkg
    return sys.modules["cooper_mapper_amd"]


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def synth():
    load_pkg()
    return importlib.import_module("the-cooper-mapper_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def goldens():
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "knn_*.npz"))):
        out[os.path.basename(f)[4:-4]] = dict(np.load(f))
    assert out, "no golden kNN fixtures found"
    return out


@pytest.fixture(scope="session")
def small_problem(synth):
    """16-ring x 900 scan against a 120 m map: the oracle finishes in ~0.1 s."""
    return synth.make_problem(rings=16, azimuth_steps=900, world_half=60.0)


@pytest.fixture(scope="session")
def _session_ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture
def ctx(_session_ctx):
    """GPU context (one per session); only requested by @pytest.mark.gpu tests.  Settings a test may have left on it
    (LaserMapping switches deferred trees on) are reset for the next one."""
    _session_ctx.defer_trees(False)
    yield _session_ctx
    _session_ctx.defer_trees(False)
