"""tools/verify_clip.py: the product against the reference on a whole .mcraw file, frame by frame.

CPU: the tool's own container reader and stream walkers on a clip written by the test-side writer (both encodings, a legacy
trailer).  GPU: the tool end to end on that clip, and -- when MCRAW_SAMPLE names a real recording (the reference's README points
at one, /root/reference/README.md:26-28; there is no network here) -- on that file."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _libs as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "verify_clip.py")


def _tool():
    spec = importlib.util.spec_from_file_location("verify_clip", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _clip(tmp_path):
    imgs = [L.natural_image_np(256, 64, 12, 12.0, 1), L.natural_image_np(250, 30, 10, 4.0, 2), L.uniform_image_np(128, 16, 14, 3),
            L.natural_image_np(96, 20, 12, 12.0, 4)]
    frames = [(3000, 7, 256, 64, L.encode7(imgs[0])),
              (1000, 6, 250, 30, L.encode6(imgs[1], flags=1)),   # written out of timestamp order; flags=1: with the trailer of restart records
              (2000, 7, 128, 16, L.encode7(imgs[2])),
              (4000, 6, 96, 20, L.encode6(imgs[3]))]
    path = str(tmp_path / "clip.mcraw")
    L.write_mcraw(path, frames, audio_chunks=[(0, np.arange(960, dtype=np.int16))])
    return path, imgs


def test_the_tools_container_reader_and_stream_walkers(tmp_path):
    V = _tool()
    path, imgs = _clip(tmp_path)
    fr = V.read_clip(path)
    assert [f[0] for f in fr] == [1000, 2000, 3000, 4000]                     # by timestamp, like Decoder::getFrames
    assert [(f[1]["width"], f[1]["height"], f[1]["compressionType"]) for f in fr] == [(250, 30, 6), (128, 16, 7), (256, 64, 7), (96, 20, 6)]
    assert len(V.read_clip(path, 2)) == 2
    # payloads decode to the images (the checker side of the tool)
    ret, out = L.oracle_decode7(np.frombuffer(fr[2][2], np.uint8), 256, 64)
    assert ret == 256 * 64 and np.array_equal(out, imgs[0])
    h7 = V.bits_hist7(fr[1][2])                                               # uniform 14-bit noise: every block raw
    assert h7 is not None and sum(h7.values()) == 4 * 2 * 4 and set(h7) <= {"14", "13"}
    h6, tr = V.scan6(fr[0][2], 250, 30)
    assert sum(h6.values()) == 2 * 256 // 32 * 30 and tr["restart_records"] >= 1
    h6b, trb = V.scan6(fr[3][2], 96, 20)
    assert trb["restart_records"] == 0 and trb["bytes_behind_the_records"] == 1  # (the encoder's single trailing byte)
    with open(path, "rb") as f:
        data = f.read()
    bad = tmp_path / "bad.mcraw"
    bad.write_bytes(data[:-7])
    with pytest.raises(V.ClipError):
        V.read_clip(str(bad))


@pytest.mark.gpu
def test_verify_clip_on_a_synthetic_clip(tmp_path):
    path, _ = _clip(tmp_path)
    r = subprocess.run([sys.executable, TOOL, path, "--json"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["verdict"]["frames"] == 4 and d["verdict"]["mismatches"] == 0 and d["verdict"]["types"] == [6, 7]
    assert all(row["equal"] and row["product_crc32"] == row["checker_crc32"] for row in d["rows"])
    # a frame whose payload was damaged in the file: both sides must still agree (pixels or rejection), or the tool says MISMATCH
    with open(path, "rb") as f:
        data = bytearray(f.read())
    data[len(data) // 3] ^= 0x5A
    p2 = tmp_path / "damaged.mcraw"
    p2.write_bytes(bytes(data))
    r2 = subprocess.run([sys.executable, TOOL, str(p2), "--json", "--no-hist", "--checker", "oracle"], capture_output=True, text=True, timeout=600)
    assert r2.returncode in (0, 1, 2)
    if r2.returncode == 0:
        assert json.loads(r2.stdout.strip().splitlines()[-1])["verdict"]["mismatches"] == 0


@pytest.mark.gpu
def test_verify_clip_on_a_real_recording():
    sample = os.environ.get("MCRAW_SAMPLE")
    if not sample or not os.path.exists(sample):
        pytest.skip("NO REAL CLIP CHECKED: set MCRAW_SAMPLE=<file.mcraw> (e.g. the sample the reference's README links) to compare the "
                    "product with the reference on a real recording; every other test decodes the build's own encoder's output")
    n = os.environ.get("MCRAW_SAMPLE_FRAMES", "16")
    r = subprocess.run([sys.executable, TOOL, sample, "-n", n], capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
