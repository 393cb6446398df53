"""CPU, world_size 2 over gloo: the frame-sharded job gives the same per-frame results as a
single process, and the timing/flag reductions behave (the N>1 path of bench.py).  The decode
itself is stood in for by the oracle here (no GPU in this container); on the GPU box the same
helpers run under RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _libs as L
from motioncam_decoder_amd import shard

N_FRAMES = 7  # odd on purpose: ranks get 4 and 3 frames


def _frame(i):
    img = L.natural_image_np(192, 16, 12, 12.0, 1000 * 5 + i)
    if i % 3 == 2:
        buf, typ = L.encode6(img), 6
    else:
        buf, typ = L.encode7(img), 7
    return typ, img, buf


def _decode(typ, buf, w, h):
    ret, out = (L.oracle_decode7 if typ == 7 else L.oracle_decode6)(buf, w, h)
    assert ret == w * h
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_frames(N_FRAMES, rank, world)
    local = {}
    for i in mine:
        typ, img, buf = _frame(i)
        local[i] = shard.frame_checksum(_decode(typ, buf, 192, 16))
    allsums = shard.gather_checksums(dist, N_FRAMES, local)
    tmax = shard.reduce_max(dist, 1.0 + rank)          # slowest rank defines the step time
    ok = shard.reduce_min_flag(dist, rank != 1)        # one failing rank fails the job
    dist.barrier()
    if rank == 0:
        q.put((mine, allsums, tmax, ok))
    else:
        q.put((mine, None, tmax, ok))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_frames_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard.shard_frames(13, r, world))
        assert seen == list(range(13))
        assert all(i % world == r for r in range(world) for i in shard.shard_frames(13, r, world))


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process():
    want = []
    for i in range(N_FRAMES):
        typ, img, buf = _frame(i)
        out = _decode(typ, buf, 192, 16)
        assert np.array_equal(out, img)
        want.append(shard.frame_checksum(out))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = sorted(r[0] for r in res)
    assert shards == [[0, 2, 4, 6], [1, 3, 5]]
    full = next(r[1] for r in res if r[1] is not None)
    assert full == want                       # per-frame results independent of the rank count
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res)   # max over ranks
    assert all(r[3] is False for r in res)            # min over ranks of the ok flag
