"""Executed proof of INTEGRATION.md's drop-in claim - build container only (the reference tree cannot travel, so this
test is skipped wherever /root/reference is absent, e.g. on the GPU box).  See tests/dropin/run_reference_over_heterobatch.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/algos"), reason="needs the reference tree (build container only)")
def test_reference_env_wrappers_cat_buffer_and_learner_run_unchanged_over_heterobatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin", "run_reference_over_heterobatch.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DROPIN OK: 48 graph arrays bit-exact" in r.stdout
