import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")



@pytest.fixture(autouse=True)
def _fresh_self_planning_cache():
    """The suites run on the library's defaults (lazy de-augmentation handles, self-planning mergers); what a merger has learnt about
    a geometry must not travel from one test into the next."""
    yield
    mod = sys.modules.get("pytorch_toolbelt_amd.inference._merge_modes")
    if mod is not None:
        with mod.auto_lock:
            mod.auto_cache.clear()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """A tests/golden/*.npz fixture written by oracle/make_golden.py (outputs of the unmodified reference)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
        self.cases = json.loads(str(self.z["__cases__"]))

    def __getitem__(self, k):
        return self.z[k]

    def by_fn(self, *fns):
        return [c for c in self.cases if c["fn"] in fns]


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope="session")
def golden_tiles():
    return load_golden("tiles.npz")


@pytest.fixture(scope="session")
def golden_tta():
    return load_golden("tta.npz")


@pytest.fixture(scope="session")
def golden_losses():
    return load_golden("losses.npz")


@pytest.fixture(scope="session")
def forced_build(tmp_path_factory):
    """ONE forced rebuild of every HIP translation unit per test session (into a scratch directory: the library the process has loaded
    is not touched), with the compiler's kernel-resource remarks kept -- shared by tests/test_abi.py (does everything still compile,
    link and export the header's symbols?) and tests/test_kernel_resources.py (register / scratch / occupancy budgets)."""
    import __graft_entry__ as g

    out = tmp_path_factory.mktemp("forced_build")
    remarks = out / "remarks"
    remarks.mkdir()
    g.build(force=True, out_dir=str(out / "lib"), remarks_dir=str(remarks))
    return {"lib_dir": str(out / "lib"), "remarks_dir": str(remarks)}
