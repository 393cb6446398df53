"""Frame sharding across the GPUs of one node (one process per GPU).

Frames are independent (raw::Decode / DecodeLegacy are pure functions of one buffer,
lib/Decoder.cpp:184-235), so a batch shards by frame index with NO collective on the data
path: frame i belongs to rank i mod world.  torch.distributed (RCCL on GPUs, gloo on CPU)
carries only the start barrier, the max of the per-rank times and small result gathers.
"""
import zlib

import numpy as np


def shard_frames(n_total, rank, world):
    """Global frame indices decoded by `rank`: i with i % world == rank."""
    return list(range(rank, n_total, world))


def frame_checksum(arr):
    """CRC32 of a decoded uint16 mosaic (device-count independent result check)."""
    return zlib.crc32(np.ascontiguousarray(arr).view(np.uint8)) & 0xFFFFFFFF


def reduce_max(dist, value, device="cpu"):
    """Max over ranks of a python float (the job takes as long as its slowest rank)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_min_flag(dist, ok, device="cpu"):
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def gather_checksums(dist, n_total, local, device="cpu"):
    """local: {global frame index: checksum}.  Returns the full list (every rank gets it):
    each rank fills its own slots of a zero vector, one SUM all-reduce merges them."""
    import torch
    t = torch.zeros(n_total, dtype=torch.int64, device=device)
    for i, c in local.items():
        t[i] = int(c)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
