"""Measurement plumbing shared by bench.py and its CPU test (tests/test_bench_dist_gloo.py).

Nothing here decodes anything: process placement (NUMA), the timed-region protocol
(warm-up, rounds of exactly K steps between barrier + synchronise, max over ranks),
and small statistics.  bench.py passes the step function; the gloo test passes a stub.
"""
import glob
import math
import os
import statistics
import time


# ------------------------------------------------------------------ placement

def gpu_numa_node(local_rank):
    """NUMA node of the local_rank-th AMD GPU (PCI order), read from sysfs without touching HIP.
    None when it cannot be told (no sysfs entry, single-node host reports -1)."""
    devs = []
    for d in glob.glob("/sys/class/drm/renderD*/device"):
        try:
            with open(os.path.join(d, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            devs.append(os.path.realpath(d))
        except OSError:
            continue
    devs = sorted(set(devs))
    if local_rank >= len(devs):
        return None
    try:
        with open(os.path.join(devs[local_rank], "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        return None
    return node if node >= 0 else None


def node_cpus(node):
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            txt = f.read().strip()
    except OSError:
        return []
    cpus = []
    for part in txt.split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return cpus


def bind_to_gpu_numa(local_rank):
    """Pin this process (and the pinned buffers it will first-touch) to the NUMA node of its GPU.
    Call BEFORE the first GPU call.  Returns {"node": n, "cpus": k} or None (nothing done)."""
    node = gpu_numa_node(local_rank)
    if node is None:
        return None
    cpus = node_cpus(node)
    if not cpus:
        return None
    try:
        allowed = os.sched_getaffinity(0)
        want = set(cpus) & allowed
        if not want:
            return None
        os.sched_setaffinity(0, want)
    except (AttributeError, OSError):
        return None
    return {"node": node, "cpus": len(want)}


# ------------------------------------------------------------------ timed region

class Comm:
    """What the timed region needs from torch.distributed (or nothing at world size 1)."""

    def __init__(self, dist=None, device="cpu"):
        self.dist = dist if (dist is not None and dist.is_available() and dist.is_initialized()) else None
        self.device = device

    @property
    def world(self):
        return self.dist.get_world_size() if self.dist else 1

    @property
    def rank(self):
        return self.dist.get_rank() if self.dist else 0

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def _reduce(self, values, op):
        import torch
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device)
        if self.dist:
            self.dist.all_reduce(t, op=op)
        return [float(x) for x in t.tolist()]

    def max(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX if self.dist else None)

    def min(self, values):
        return self._reduce(values, self.dist.ReduceOp.MIN if self.dist else None)

    def sum(self, values):
        return self._reduce(values, self.dist.ReduceOp.SUM if self.dist else None)


def timed_rounds(step, sync, comm, steps, warmup, min_seconds=0.5, max_rounds=200):
    """The bench.py protocol: `warmup` untimed steps, then rounds of EXACTLY `steps` steps, each round
    bracketed by barrier + sync on both sides.  The number of rounds is chosen (by the slowest rank,
    after the first round) so that the timed rounds cover at least `min_seconds`.  Returns the list of
    round times in seconds, each already the MAX over ranks."""
    for i in range(warmup):
        step(i)
    sync()

    def one_round():
        comm.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        sync()
        comm.barrier()
        return time.perf_counter() - t0

    first = comm.max([one_round()])[0]
    rounds = int(min(max_rounds, max(1, math.ceil(min_seconds / max(first, 1e-9)))))
    rounds = int(comm.max([rounds])[0])  # every rank runs the same number
    local = [one_round() for _ in range(rounds)]
    return comm.max(local)


def round_stats(times, steps):
    """ms per step: median over the rounds, with the spread."""
    ms = sorted(1e3 * t / steps for t in times)
    return {"median": statistics.median(ms), "min": ms[0], "max": ms[-1], "rounds": len(ms)}
