"""CPU: the reference's OWN test files for this path -- tests/test_tiles.py, tests/test_tta.py, tests/test_losses.py of the unmodified
checkout under /root/reference -- collected and run UNMODIFIED against this package through the drop-in alias (`compat/pytorch_toolbelt`
first on sys.path; `oracle/ref_shim` only supplies the `cv2` module test_tta.py imports at its top).  SURVEY.md section 7, step 0.

The reference checkout exists in the build container only; on the GPU box this test skips (nothing at run time may read it)."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

REF_TESTS = "/root/reference/tests"
FILES = ["test_tiles.py", "test_tta.py", "test_losses.py"]


@pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="the reference checkout is not present on this machine")
def test_reference_tests_pass_unmodified_through_the_alias(tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "compat"), ROOT, os.path.join(ROOT, "oracle", "ref_shim")])
    env.pop("PTB_AUTO_PLAN", None)
    probe = subprocess.run([sys.executable, "-c", "import pytorch_toolbelt, pytorch_toolbelt.inference.tiles as t; print(pytorch_toolbelt.__file__); print(t.TileMerger.__module__)"],
                           env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert probe.returncode == 0, probe.stderr
    where, module = probe.stdout.strip().splitlines()[-2:]
    assert where.startswith(os.path.join(ROOT, "compat")) and module.startswith("pytorch_toolbelt_amd."), (where, module)
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)] + [os.path.join(REF_TESTS, f) for f in FILES],
                         env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
    tail = run.stdout.strip().splitlines()[-1] if run.stdout.strip() else ""
    assert run.returncode == 0, run.stdout[-4000:] + run.stderr[-2000:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 37 and "failed" not in tail and "error" not in tail, tail      # 37 passed, 2 skipped (CUDA) -- as with the reference itself
