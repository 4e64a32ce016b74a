"""RCCL on the one-GPU box: the collectives of the N > 1 bench path with ONE rank (more ranks cannot share a card under RCCL;
world size 2 and 8 run over gloo in tests/test_dist_gloo.py).  In a child process: a process group must not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sharded_pipeline_over_rccl_single_rank():
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_pipeline_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-PIPELINE-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
