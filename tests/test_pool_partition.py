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
