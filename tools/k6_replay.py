#!/usr/bin/env python3
"""What would a map stage without vector work be worth?  k6_decode from a -DMCRAW_DIAG library: normal, recording its maps,
and with the maps read back instead of walked (tools/k6_replay.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import _libs as L
import motioncam_decoder_amd as M

w, h, n = 4000, 3000, 32
dev = torch.device("cuda:0")
imgs = [L.synth_image(w, h, 12, 1, 12.0, 6000 + i) for i in range(4)]
bufs = [L.encode6(im) for im in imgs]
tin = [torch.from_numpy(bufs[i % 4]).to(dev) for i in range(n)]
tout = torch.zeros(n * w * h * 2, dtype=torch.uint8, device=dev)
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), w, h, 6, tout.data_ptr() + i * w * h * 2, w * h) for i in range(n)])
ctx = M.Context(0)
lib = M.load()
ctx.profile(True)


def run(mode, reps=10):
    lib.mcraw_diag_k6_maps(mode)
    tout.zero_()
    written, status = ctx.decode_batch(frames)
    assert all(s == 0 for s in status), status
    got = tout[: w * h * 2].cpu().numpy().view(np.uint16).reshape(h, w)
    assert np.array_equal(got, imgs[0]), "mode %d decodes wrong" % mode
    ctx.kernel_ms("k6_decode", reset=True)
    for _ in range(reps):
        ctx.decode_batch(frames, want_status=False)
    torch.cuda.synchronize()
    return ctx.kernel_ms("k6_decode", reset=True)[0] / reps


for rnd in range(3):
    print("normal %.4f  record %.4f  replay %.4f  replay, loads issued early %.4f ms" % (run(0), run(1), run(2), run(3)), flush=True)
