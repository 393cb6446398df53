"""CPU, plan only: the partition rule of the device pool (include/mcraw_hip.h: mcraw_shard_of / mcraw_shard_count,
frame i -> member i mod G) for G = 1..8 -- every frame decoded exactly once, by the member the Python side of the
job (motioncam_decoder_amd/shard.py, what bench.py --gpus N uses) expects, shares within one frame of each other."""
import motioncam_decoder_amd as M
from motioncam_decoder_amd import shard


def test_partition_rule_matches_shard_py_for_1_to_8_devices():
    lib = M.load()  # the library loads and answers without a GPU
    for G in range(1, 9):
        for n in (0, 1, 7, 8, 240, 960, 961):
            owners = [lib.mcraw_shard_of(i, G) for i in range(n)]
            assert all(0 <= o < G for o in owners)
            for m in range(G):
                mine = [i for i, o in enumerate(owners) if o == m]
                assert mine == shard.shard_frames(n, m, G)
                assert lib.mcraw_shard_count(n, m, G) == len(mine)
            counts = [lib.mcraw_shard_count(n, m, G) for m in range(G)]
            assert sum(counts) == n and max(counts) - min(counts) <= 1


def test_partition_rule_rejects_nonsense():
    lib = M.load()
    assert lib.mcraw_shard_of(3, 0) == -1 and lib.mcraw_shard_of(-1, 4) == -1
    assert lib.mcraw_shard_count(10, 4, 4) == -1 and lib.mcraw_shard_count(-1, 0, 4) == -1


def test_pool_needs_a_gpu():
    import pytest
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(M.McrawError):
        M.Pool()


def test_legacy_launch_order_covers_every_segment_once_and_in_order():
    """mcraw_legacy_launch_order (the table k6_decode's workgroups find their frame and nominal segment in): every
    (frame, segment) exactly once, a frame's segments in rising order, no workgroup for nothing, frames of very
    different sizes and empty ones included."""
    import ctypes as C
    import numpy as np
    import motioncam_decoder_amd as M
    lib = C.CDLL(M.lib_path())
    lib.mcraw_legacy_launch_order.restype = None
    lib.mcraw_legacy_launch_order.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(11)
    cases = [[5], [3, 3, 3], [1, 1000, 7, 7, 0, 250], [0, 0, 4], list(rng.integers(0, 60, size=37)), list(rng.integers(1, 5, size=300))]
    for nseg in cases:
        n = len(nseg)
        a = (C.c_uint32 * n)(*[int(v) for v in nseg])
        tab = (C.c_uint32 * (3 * n + 1))()
        lib.mcraw_legacy_launch_order(a, n, tab)
        t = list(tab)
        base, lo, order = t[: n + 1], t[n + 1: 2 * n + 1], t[2 * n + 1:]
        assert base[n] == sum(int(v) for v in nseg) and sorted(order) == list(range(n))
        assert all(nseg[order[i]] >= nseg[order[i + 1]] for i in range(n - 1))
        seen = {}
        for b in range(base[n]):
            st = max(s for s in range(n) if base[s] <= b)       # the kernel's find_frame: the last stage that starts at or before b
            inplay = n - st
            f = order[(b - base[st]) % inplay]
            seg = lo[st] + (b - base[st]) // inplay
            assert seg < nseg[f], (nseg, b, f, seg)
            assert seen.get(f, -1) == seg - 1, (nseg, b, f, seg)  # a frame's segments in rising order, none skipped
            seen[f] = seg
        assert all(seen.get(f, -1) == int(nseg[f]) - 1 for f in range(n))
        # the rounds every frame takes part in (the kernel's table-free path): workgroup b -> frame b % n, segment b // n
        if n > 1:
            assert base[1] == n * min(int(v) for v in nseg)
