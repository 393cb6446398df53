"""GPU (-m gpu): a legacy workgroup whose look-back words never arrive fails ITS FRAME, it does not hang the launch.

k6_decode's workgroups wait for the record counts of the frame's earlier segments (decoupled look-back, bounded polls:
SPIN6).  A second build of the same sources with -DMCRAW_INJECT_LOST lets segment 3 of the batch's first legacy frame
never publish its words (and polls 2^12 instead of 2^20 times): every later segment of that frame must give up, the frame
must come back with MCRAW_E_DEVICE and nothing written counted, the other frames of the batch must decode bit-exactly,
and the call must return -- run in a child process under a timeout."""
import os
import shutil
import subprocess
import sys

import pytest

from motioncam_decoder_amd import build as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, numpy as np, torch
sys.path[:0] = [%(root)r, %(tests)r]
import _libs as L
import motioncam_decoder_amd as M
dev = torch.device("cuda:0")
ctx = M.Context(0)
imgs = [L.natural_image_np(1024, 256, 12, 12.0, 900 + i) for i in range(3)]        # ~0.26 MB of stream each: 16+ segments
bufs = [L.encode6(im) for im in imgs]
assert all(b.size > 6 * 16384 for b in bufs)
tin = [torch.from_numpy(b).to(dev) for b in bufs]
tout = [torch.zeros(im.size * 2, dtype=torch.uint8, device=dev) for im in imgs]
frames = M.Context.make_frames([(tin[i].data_ptr(), tin[i].numel(), 1024, 256, 6, tout[i].data_ptr(), imgs[i].size) for i in range(3)])
written, status = ctx.decode_batch(frames)
print("status", [hex(s) for s in status], "written", written)
assert status[0] & M.E_DEVICE and written[0] == 0, "the frame with the lost segment must fail"
for i in (1, 2):
    assert status[i] == 0 and written[i] == imgs[i].size
    assert np.array_equal(tout[i].cpu().numpy().view(np.uint16).reshape(imgs[i].shape), imgs[i])
# the context is usable afterwards: the same batch again gives the same answer
written2, status2 = ctx.decode_batch(frames)
assert [bool(s) for s in status2] == [True, False, False]
print("ok")
'''


def test_lost_lookback_word_fails_one_frame_and_returns(tmp_path):
    lib = str(tmp_path / "libmcraw_lost.so")
    B.build_variant(lib, ['-DMCRAW_INJECT_LOST'])
    env = dict(os.environ, MCRAW_LIB_PATH=lib)
    code = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr[-3000:]
