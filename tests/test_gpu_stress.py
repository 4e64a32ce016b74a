"""All API families from several host threads at once (tests/stress_threads.py, a few seconds of it): the extractor's handles, the
matcher's workspace pool, the FEM block / stream caches and the stereo path share one device and one process in the reference's
threading model (Tracking, LocalMapping, LoopClosing: SURVEY 8b)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_all_families_from_six_threads():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_threads.py")
    r = subprocess.run([sys.executable, script, "4", "6"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "errors []" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
