"""`bench.py --gpus N` starts its own ranks (orb_slam2_e_amd/launch.py): spawn, relay of rank 0's line, exit codes.
CPU only: stand-in rank scripts, plus bench.py itself up to the point where a rank finds no GPU."""
import io
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_e_amd import launch  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launch_module_needs_no_torch():
    # the parent must not touch the GPU: the launcher may import the standard library only
    code = ("import sys; sys.path.insert(0, %r); from orb_slam2_e_amd import launch; "
            "assert 'torch' not in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_should_spawn_and_check_world():
    assert not launch.should_spawn(1, {})
    assert launch.should_spawn(2, {})
    assert not launch.should_spawn(2, {"WORLD_SIZE": "2"})        # already a rank of some launcher
    assert launch.check_world(1, {}) == 1
    assert launch.check_world(4, {"WORLD_SIZE": "4"}) == 4
    with pytest.raises(SystemExit):
        launch.check_world(8, {"WORLD_SIZE": "1"})
    with pytest.raises(SystemExit):
        launch.check_world(1, {"WORLD_SIZE": "2"})
    e = launch.rank_env(3, 8, 1234, {})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "1234")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_spawn_relays_rank0_line_and_checks_n_gpus(tmp_path):
    s = _script(tmp_path, """
        import json, os, sys
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        print("noise from rank", r, file=sys.stderr)
        if r == 0:
            print("not json")
            print(json.dumps({"n_gpus": w, "value": 1.5, "argv": sys.argv[1:]}))
        else:
            print("rank", r, "stdout")        # never reaches the parent's stdout
        """)
    out, err = io.StringIO(), io.StringIO()
    rc = launch.run_parent(s, ["--gpus", "3", "--steps", "7"], 3, timeout=60, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = out.getvalue().strip().splitlines()
    res = json.loads(lines[-1])
    assert res == {"n_gpus": 3, "value": 1.5, "argv": ["--gpus", "3", "--steps", "7"]}
    assert all("stdout" not in l for l in lines)
    e = err.getvalue()
    assert "[rank 1] rank 1 stdout" in e and "[rank 2] noise from rank 2" in e


def test_spawn_fails_when_a_rank_fails_and_stops_the_others(tmp_path):
    s = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)                        # the survivors would hang in a collective
        """)
    out, err = io.StringIO(), io.StringIO()
    import time
    t0 = time.monotonic()
    rc = launch.run_parent(s, [], 3, timeout=120, out=out, err=err)
    assert rc == 7 and time.monotonic() - t0 < 60
    assert "rank 1 exited with 7" in err.getvalue()


def test_spawn_fails_on_wrong_n_gpus_or_missing_line(tmp_path):
    s = _script(tmp_path, """
        import json, os
        if os.environ["RANK"] == "0":
            print(json.dumps({"n_gpus": 1}))     # a bench that ignored --gpus
        """)
    err = io.StringIO()
    assert launch.run_parent(s, [], 2, timeout=60, out=io.StringIO(), err=err) == 4
    assert "n_gpus = 1" in err.getvalue()
    s2 = _script(tmp_path, "pass\n")
    assert launch.run_parent(s2, [], 2, timeout=60, out=io.StringIO(), err=io.StringIO()) == 3


def test_spawn_timeout(tmp_path):
    s = _script(tmp_path, "import time; time.sleep(600)\n")
    assert launch.run_parent(s, [], 2, timeout=1.0, out=io.StringIO(), err=io.StringIO()) == 124


def test_bench_rejects_a_launcher_world_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_gpus2_spawns_two_ranks_and_fails_loudly_without_gpus():
    # no GPU in this container: both ranks must get as far as the device check and the parent must report the failure
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "300"],
                       env=env, capture_output=True, text=True, timeout=400)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: this is the no-GPU failure path")
    assert r.returncode != 0
    assert "needs GPU" in r.stderr and "[rank " in r.stderr
    assert not r.stdout.strip()


def test_rank_cpu_masks():
    """bench.py pins every rank before it touches the GPU: masks are disjoint, non-empty, inside what the process may use, and
    follow the GPUs' locality when sysfs tells it (ranks behind one socket share that socket's CPUs evenly)."""
    from orb_slam2_e_amd import launch
    avail = list(range(384))
    near = lambda r: list(range(0, 96)) + list(range(192, 288)) if r < 4 else list(range(96, 192)) + list(range(288, 384))
    masks = [launch.rank_cpu_mask(r, 8, avail, near) for r in range(8)]
    assert all(len(m) == 48 for m in masks)
    assert len(set().union(*map(set, masks))) == 384
    assert all(set(masks[r]) <= set(near(r)) for r in range(8))
    # no topology: contiguous shares; fewer CPUs than ranks: everybody still gets one
    masks = [launch.rank_cpu_mask(r, 8, list(range(16)), None) for r in range(8)]
    assert masks == [[2 * r, 2 * r + 1] for r in range(8)]
    assert all(len(launch.rank_cpu_mask(r, 8, [3, 5, 9], None)) == 1 for r in range(8))
    # a restricted process (cgroup / taskset): locality is intersected with what is allowed; a rank whose GPU's CPUs are all
    # outside falls back to the contiguous share
    masks = [launch.rank_cpu_mask(r, 2, list(range(0, 8)), lambda r: list(range(4 * r, 4 * r + 4))) for r in range(2)]
    assert masks == [[0, 1, 2, 3], [4, 5, 6, 7]]
    assert launch.rank_cpu_mask(1, 2, [0, 1], lambda r: [100, 101]) == [1]
    assert launch.rank_cpu_mask(0, 1, [4, 5], None) == [4, 5]
    assert launch._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert launch.pin_rank(0, 1) is None


def test_gpu_locality_from_a_fake_sysfs(tmp_path):
    """Both sources of a GPU's local CPUs, on a made-up /sys: the KFD topology (location_id -> PCI function) and, when its per-node
    files are unreadable (they are for unprivileged users on the GPU boxes), the AMD render nodes in bus order."""
    from orb_slam2_e_amd import launch
    sys_ = tmp_path / "sys"
    for i, (bdf, cpus) in enumerate([("0000:05:00.0", "0-3"), ("0000:85:00.0", "4-7")]):
        d = sys_ / "bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n"); (d / "vendor").write_text("0x1002\n")
        r = sys_ / "class/drm" / f"renderD{128 + i}"
        r.mkdir(parents=True)
        (r / "device").symlink_to(d)
    other = sys_ / "bus/pci/devices/0000:01:00.0"; other.mkdir(parents=True)
    (other / "local_cpulist").write_text("0-7\n"); (other / "vendor").write_text("0x1a03\n")
    r = sys_ / "class/drm/renderD130"; r.mkdir(parents=True); (r / "device").symlink_to(other)      # a BMC's VGA: not a GPU of ours
    assert launch.gpu_local_cpus(0, str(sys_)) == [0, 1, 2, 3] and launch.gpu_local_cpus(1, str(sys_)) == [4, 5, 6, 7]
    assert launch.gpu_local_cpus(2, str(sys_)) is None
    nodes = sys_ / "class/kfd/kfd/topology/nodes"
    for n, (simd, loc) in enumerate([(0, 0), (256, 0x8500), (256, 0x0500)]):       # KFD order differs from bus order here: it wins
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / "properties").write_text(f"cpu_cores_count 8\nsimd_count {simd}\ndomain 0\nlocation_id {loc}\n")
    assert launch.gpu_local_cpus(0, str(sys_)) == [4, 5, 6, 7] and launch.gpu_local_cpus(1, str(sys_)) == [0, 1, 2, 3]


def test_visible_device_lists_are_mapped_to_physical_gpus():
    """pin_rank pins a rank to the CPUs of the GPU it really uses: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumber the
    devices, and a gloo rehearsal (every rank on device 0) must not pretend to a locality."""
    from orb_slam2_e_amd import launch
    v = launch.visible_device_index
    assert v(3, {}) == 3
    assert v(1, {"HIP_VISIBLE_DEVICES": "4,6"}) == 6 and v(0, {"CUDA_VISIBLE_DEVICES": "5"}) == 5
    assert v(1, {"ROCR_VISIBLE_DEVICES": "2,3,7"}) == 3
    assert v(1, {"ROCR_VISIBLE_DEVICES": "2,3,7", "HIP_VISIBLE_DEVICES": "2,0"}) == 2      # HIP's list indexes ROCr's
    assert v(2, {"HIP_VISIBLE_DEVICES": "0,1"}) is None and v(0, {"ROCR_VISIBLE_DEVICES": "GPU-abc"}) is None
    import os
    if len(os.sched_getaffinity(0)) >= 2:
        before = os.sched_getaffinity(0)
        try:
            r = launch.pin_rank(1, 2, dev_index=0)          # rehearsal: rank 1 on device 0
            assert r["from"] == "contiguous share" and r["gpu"] == {"hip_device": 0, "physical": launch.visible_device_index(0)}
        finally:
            os.sched_setaffinity(0, before)
