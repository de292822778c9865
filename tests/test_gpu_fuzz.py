"""A short run of tools/fuzz_msm.py (randomised differential test HIP MSM vs C oracle) inside the
GPU suite; longer runs (76 000 cases, 0 mismatches in round 1) are done by hand."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_msm_short():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_msm.py"), "12"], capture_output=True, text=True,
                       env=dict(os.environ, FUZZ_SEED="7"), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_fuzz_ops_short():
    """tools/fuzz_ops.py: IPA rounds, folds, batch scalar mults, sums, mod-q bulk ops, decompression
    (24 500 cases clean in the long run of round 1)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_ops.py"), "12"], capture_output=True, text=True,
                       env=dict(os.environ, FUZZ_SEED="8"), timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
