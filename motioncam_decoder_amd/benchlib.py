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

def _visible_ordinals():
    """HIP ordinal -> physical ordinal (PCI order) under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (numeric lists only;
    None: no remapping)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None  # (UUID form: the order cannot be told without the runtime)
    return None


def gpu_pci_devices():
    """sysfs device directories of the AMD GPUs, ordered by PCI address (domain:bus:device.function, numerically) --
    the order in which the HIP runtime numbers them when nothing remaps it."""
    devs = {}
    for d in glob.glob("/sys/class/drm/renderD*/device"):
        try:
            with open(os.path.join(d, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            real = os.path.realpath(d)
            bdf = os.path.basename(real)  # e.g. 0000:c1:00.0
            dom, bus, rest = bdf.split(":")
            dev, fn = rest.split(".")
            devs[(int(dom, 16), int(bus, 16), int(dev, 16), int(fn, 16))] = real
        except (OSError, ValueError):
            continue
    return [devs[k] for k in sorted(devs)]


def gpu_numa_node(local_rank, bdf=None):
    """NUMA node of the local_rank-th visible AMD GPU (or of the GPU at PCI address `bdf`, e.g. from
    torch.cuda.get_device_properties), read from sysfs without touching HIP.  Returns (node, bdf); node is None when it
    cannot be told (no sysfs entry, single-node host reports -1)."""
    path = None
    if bdf:
        path = os.path.join("/sys/bus/pci/devices", bdf.lower())
    else:
        devs = gpu_pci_devices()
        vis = _visible_ordinals()
        idx = local_rank
        if vis is not None:
            if local_rank >= len(vis):
                return None, None
            idx = vis[local_rank]
        if idx >= len(devs):
            return None, None
        path = devs[idx]
        bdf = os.path.basename(path)
    try:
        with open(os.path.join(path, "numa_node")) as f:
            node = int(f.read().strip())
    except (OSError, ValueError):
        return None, bdf
    return (node if node >= 0 else None), bdf


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


def bind_to_gpu_numa(local_rank, bdf=None):
    """Pin this process (and the pinned buffers it will first-touch) to the NUMA node of its GPU.
    Call BEFORE the first pinned allocation.  Returns {"node": n, "cpus": k, "bdf": PCI address} or None (nothing done)."""
    node, bdf = gpu_numa_node(local_rank, bdf)
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
    return {"node": node, "cpus": len(want), "bdf": bdf}


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

    def gather(self, value):
        """Every rank's `value` (a number), by rank: one all-reduce of a one-hot vector."""
        w, r = self.world, self.rank
        return self.sum([float(value) if i == r else 0.0 for i in range(w)])


def timed_rounds(step, sync, comm, steps, warmup, min_seconds=0.5, max_rounds=200, own=None):
    """The bench.py protocol: `warmup` untimed steps, then rounds of EXACTLY `steps` steps, each round
    bracketed by barrier + sync on both sides.  The number of rounds is chosen (by the slowest rank,
    after the first round) so that the timed rounds cover at least `min_seconds`.  Returns the list of
    round times in seconds, each already the MAX over ranks.  `own` (a list): this rank's own round times, taken when ITS
    steps were done, in front of the closing barrier -- what the per-rank figures of an N > 1 line are made of."""
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
        mine = time.perf_counter() - t0
        comm.barrier()
        return time.perf_counter() - t0, mine

    first = comm.max([one_round()[0]])[0]
    rounds = int(min(max_rounds, max(1, math.ceil(min_seconds / max(first, 1e-9)))))
    rounds = int(comm.max([rounds])[0])  # every rank runs the same number
    both = [one_round() for _ in range(rounds)]
    if own is not None:
        own.extend(m for _, m in both)
    return comm.max([t for t, _ in both])


def round_stats(times, steps):
    """ms per step: median over the rounds, with the spread."""
    ms = sorted(1e3 * t / steps for t in times)
    return {"median": statistics.median(ms), "min": ms[0], "max": ms[-1], "rounds": len(ms)}
