"""GPU (-m gpu): the two forms of k7_side and the side stream (mcraw_abi.hip, MCRAW_SIDE_CUS).

k7_side (lib/RawData.cpp:463-498 as one kernel) is one template with two instances: the fat workgroups every batch runs with,
and thin ones (256 threads, pieces of 12 KiB) that fit beside the tile kernel's workgroups -- what a batch runs whose k7_side
is put on the context's side stream, beside the tile kernel of the batch in front (an experiment that is off by default: it
hides k7_side and slows the tile kernel by as much, docs/lab_notes.md).  Both paths must decode every input like the
reference: the type-7 suites run against the thin instance (MCRAW_SIDE_THIN=1), and batches queued back to back run with the
side stream in both of its forms (lowest-priority stream; CU-masked streams)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _suites(env, suites, timeout=900):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"]
                       + [os.path.join(ROOT, "tests", s) for s in suites], env=env, capture_output=True, text=True, timeout=timeout)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail + r.stderr[-2000:]


def test_type7_suites_with_thin_side_workgroups():
    _suites(dict(os.environ, MCRAW_SIDE_THIN="1"),
            ["test_gpu_parity.py", "test_gpu_fuzz.py", "test_gpu_negative.py", "test_encoder_variants.py"])


def test_type7_suites_with_thin_side_workgroups_and_forced_parts():
    _suites(dict(os.environ, MCRAW_SIDE_THIN="1", MCRAW_SIDE_SPLIT="3,2"), ["test_gpu_parity.py", "test_gpu_negative.py"])


QUEUED = r'''
import sys
import numpy as np
import torch
sys.path[:0] = [%(root)r, %(tests)r]
import _libs as L
import motioncam_decoder_amd as M
dev = torch.device("cuda:0")
ctx = M.Context(0)
rng = np.random.default_rng(5)
shapes = [(3840, 2160, 12, 1), (4032, 3024, 12, 1), (1920, 1080, 10, 1), (640, 480, 14, 0), (4000, 3000, 12, 1)]
sets = []
for k in range(3):
    imgs, tins, touts, descs = [], [], [], []
    for i in range(40):
        w, h, nb, nat = shapes[(i + k) %% len(shapes)]
        typ = 6 if (w == 4000 and k == 1) else 7   # one set holds both encodings
        img = L.synth_image(w, h, nb, nat, 12.0, 100 * k + (i %% 7))
        buf = L.encode7(img) if typ == 7 else L.encode6(img)
        ti = torch.from_numpy(buf).to(dev)
        to = torch.zeros(w * h * 2, dtype=torch.uint8, device=dev)
        imgs.append(img); tins.append(ti); touts.append(to)
        descs.append((ti.data_ptr(), ti.numel(), w, h, typ, to.data_ptr(), w * h))
    sets.append((imgs, tins, touts, M.Context.make_frames(descs)))
torch.cuda.synchronize()
for rep in range(8):           # batches that follow each other on the context's own stream: no status asked for
    for imgs, tins, touts, frames in sets:
        ctx.decode_batch(frames, mem=M.MEM_DEVICE, want_status=False)
st = ctx.synchronize(40)
assert all(s == 0 for s in st), st
assert ctx.errors() == 0
for imgs, tins, touts, frames in sets:
    for img, to in zip(imgs, touts):
        got = to.cpu().numpy().view(np.uint16).reshape(img.shape)
        assert np.array_equal(got, img)
ctx.close()
print("queued ok")
'''


@pytest.mark.parametrize("mode", ["-1", "2"])
def test_queued_batches_with_k7_side_on_the_side_stream(mode, tmp_path):
    src = tmp_path / "queued.py"
    src.write_text(QUEUED % {"root": ROOT, "tests": os.path.join(ROOT, "tests")})
    r = subprocess.run([sys.executable, str(src)], env=dict(os.environ, MCRAW_SIDE_CUS=mode), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "queued ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
