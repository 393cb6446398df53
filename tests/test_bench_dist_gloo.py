"""CPU: the multi-rank protocol of bench.py -- process group, NUMA binding helper, timed rounds bracketed by
barriers, max-over-ranks, rank-0 JSON line -- run exactly as the driver launches it (torch.distributed.run,
world size 2), over gloo, with a sleep standing in for the GPU decode (`--stub-decode`)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from motioncam_decoder_amd import benchlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_bench_protocol_world2_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--stub-decode", "--min-seconds", "0.1", "--frames", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 alone prints, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["rounds"] >= 2                   # 3 steps of 4 ms: several rounds to cover 0.1 s
    # the slower rank (4 ms per step) sets the time; the faster one (2 ms) does not
    # (upper bound generous: sleeps overshoot on a loaded host -- the build may still be running beside the tests)
    assert 3.9 <= d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"] < 60.0
    assert d["bit_exact"] is True
    px = 2 * 5 * 3840 * 2160
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) < 1.0   # whole-job rate over both ranks
    # every rank's own time beside the max over ranks: the stub's rank 1 takes twice as long per step as its rank 0
    own = d["per_rank"]["ms_per_step"]
    assert len(own) == 2 and 1.9 <= own[0] < own[1] <= d["ms_per_step"] + 0.5 and own[1] > 3.9


@pytest.mark.timeout(300)
def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no RANK in the environment: the parent (which never touches HIP) starts two ranks as a
    child process and relays rank 0's line -- an N-GPU request never comes back as a one-rank measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-decode",
           "--min-seconds", "0.02", "--frames", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_bench_refuses_to_measure_fewer_gpus_than_asked_for():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    from motioncam_decoder_amd import benchlib as bl
    have = len(bl.gpu_pci_devices())
    # more GPUs than the machine has (none in the CPU container): no line, a message, a non-zero exit code
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "1"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # a launcher whose world size disagrees with --gpus
    env2 = dict(env, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-decode", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=env2)
    assert r.returncode != 0 and "must agree" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_timed_rounds_single_process():
    calls = []
    comm = benchlib.Comm(None)
    times = benchlib.timed_rounds(lambda i: calls.append(i), lambda: None, comm, steps=4, warmup=2, min_seconds=0.0)
    assert len(times) == 1 and calls[:2] == [0, 1]
    assert len(calls) == 2 + 4 + 4            # warm-up, the sizing round, one timed round
    st = benchlib.round_stats([0.004, 0.008, 0.006], 2)
    assert st == {"median": 3.0, "min": 2.0, "max": 4.0, "rounds": 3}


def test_numa_helpers_do_not_need_a_gpu():
    # no AMD GPU in this container: the helper says so instead of failing
    node, bdf = benchlib.gpu_numa_node(0)
    assert node is None or isinstance(node, int)
    # the GPUs are taken in PCI order and HIP_VISIBLE_DEVICES is honoured (a remapped rank must not bind to another GPU's node)
    os.environ["HIP_VISIBLE_DEVICES"] = "5,2"
    try:
        assert benchlib._visible_ordinals() == [5, 2]
        assert benchlib.gpu_numa_node(2) == (None, None)
    finally:
        del os.environ["HIP_VISIBLE_DEVICES"]
    assert benchlib.bind_to_gpu_numa(63) is None


def test_bench_frame_content_does_not_depend_on_the_world_size():
    """bench.py: global frame g of the job holds image g mod 48 at every --gpus N (so that the digest of the per-frame checksums,
    `frame_checksums`, is one value for the N = 1, 2, 4, 8 lines): every rank's local frame i must map to that image."""
    import importlib.util
    from motioncam_decoder_amd import shard
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for distinct, frames in ((48, 240), (4, 24), (7, 30), (48, 5)):
        for world in (1, 2, 3, 4, 8):
            for rank in range(world):
                gidx = shard.shard_frames(world * frames, rank, world)
                seeds = b.frame_seeds(gidx, distinct, frames, 3)
                d = len(seeds)
                assert 1 <= d <= min(distinct, frames)
                for i, g in enumerate(gidx[:frames]):
                    assert seeds[i % d] == 3000 + g % distinct, (distinct, frames, world, rank, i)


def test_bench_profile_helpers_read_the_committed_profiles():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    t, src = b.traffic_from_profile("3840x2160_12bit_240_nat", 6168487920)
    assert t and 1.0 <= t / 6168487920 < 1.05 and "traffic.json" in src
    t2, src2 = b.traffic_from_profile("7680x4320_12bit_120_nat", 12343588800)      # no --pmc pass of this geometry: the ratio
    assert t2 and abs(t2 / 12343588800 - t / 6168487920) < 1e-9 and "ratio" in src2
    assert b.traffic_from_profile("1x1_12bit_1_other", 100) == (None, None)
    ms, path = b.profile_launch_ms("nat")
    assert ms and 0.5 < ms < 2.0 and path.startswith("profiles/r")
