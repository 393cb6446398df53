#!/usr/bin/env python3
"""What this MI355X's HBM delivers for plain streams (torch kernels, 4 GB buffers > Infinity Cache):
write-only fill, read+write copy, read-only sum.  Context for roofline.frac in bench.py."""
import json
import torch

n = 4 * 1024 ** 3
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a32, b32 = a.view(torch.int32), b.view(torch.int32)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


res = {}
t = timed(lambda: a32.fill_(7)); res["fill_write_only_GBs"] = n / t / 1e9
t = timed(lambda: b32.copy_(a32)); res["copy_read_plus_write_GBs"] = 2 * n / t / 1e9
t = timed(lambda: a32.sum()); res["sum_read_only_GBs"] = n / t / 1e9
print(json.dumps({k: round(v, 1) for k, v in res.items()}))
