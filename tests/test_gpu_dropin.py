"""GPU (-m gpu): end-to-end drop-in.  A synthetic .mcraw (current + legacy frames written out of
timestamp order, audio chunks with and without timestamps) goes through
  * mcraw_export (own CLI over motioncam::Decoder::loadFrames -> one GPU batch, and --single),
  * example_dropin: the reference's example.cpp, compiled unchanged against this repository,
  * example_ref:    the reference built from its own sources (CPU codec),
and the DNG / WAV files of the last two must be byte-identical."""
import os
import subprocess
import zlib

import numpy as np
import pytest

import _libs as L

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXPORT = os.path.join(ROOT, "motioncam_decoder_amd", "lib", "mcraw_export")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "example_dropin")
REFEX = os.path.join(ROOT, "oracle", "_ref", "example_ref")


@pytest.fixture(scope="module")
def container(tmp_path_factory):
    d = tmp_path_factory.mktemp("mcraw")
    specs = [(3000, 7, 1920, 1080, 12, 12.0), (1000, 7, 640, 480, 10, 4.0), (2000, 6, 800, 600, 12, 12.0),
             (4000, 6, 1000, 30, 14, 40.0), (5000, 7, 200, 12, 12, 12.0)]
    frames, images = [], {}
    for ts, typ, w, h, nb, sig in specs:
        img = L.natural_image_np(w, h, nb, sig, ts)
        frames.append((ts, typ, w, h, L.encode7(img) if typ == 7 else L.encode6(img)))
        images[ts] = img
    audio = [(111, np.arange(1920, dtype=np.int16)), (None, (np.arange(1920, dtype=np.int16) * 3).astype(np.int16))]
    path = L.write_mcraw(str(d / "clip.mcraw"), frames, audio)
    return d, path, images, audio


def _run(cmd, cwd):
    env = dict(os.environ)
    return subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=300, env=env)


@pytest.mark.parametrize("mode", ["batch", "single"])
def test_export_tool(container, mode, tmp_path):
    d, path, images, audio = container
    if not os.path.exists(EXPORT):
        from motioncam_decoder_amd import build
        build.build_host()
    cmd = [EXPORT, path, "-o", str(tmp_path)] + (["--single"] if mode == "single" else [])
    r = _run(cmd, str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "Found 5 frames" in r.stdout
    order = sorted(images)
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == 5
    for i, ts in enumerate(order):
        got = np.fromfile(str(tmp_path / ("frame_%06d.u16" % i)), dtype=np.uint16)
        assert np.array_equal(got.reshape(images[ts].shape), images[ts]), (i, ts)
        assert ("crc32 %08x" % (zlib.crc32(images[ts].tobytes()) & 0xFFFFFFFF)) in lines[i]
    pcm = np.fromfile(str(tmp_path / "audio.s16"), dtype=np.int16)
    assert np.array_equal(pcm, np.concatenate([a[1] for a in audio]))


def test_export_tool_fused_post_stage(container, tmp_path):
    # Decoder::loadFrames with FrameOutput{subtractBlackLevel, bitsPerSample 12}: black levels from the
    # container metadata (64 for every CFA position in this clip), rows as 12-bit strips
    d, path, images, audio = container
    if not os.path.exists(EXPORT):
        from motioncam_decoder_amd import build
        build.build_host()
    r = _run([EXPORT, path, "-o", str(tmp_path), "--black", "--bits", "12"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    for i, ts in enumerate(sorted(images)):
        img = images[ts]
        want = L.oracle_post(img, [64, 64, 64, 64], True)
        got = np.fromfile(str(tmp_path / ("frame_%06d.p12" % i)), dtype=np.uint8)
        assert np.array_equal(got.reshape(want.shape), want), (i, ts)
    r = _run([EXPORT, path, "-o", str(tmp_path), "--black"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    for i, ts in enumerate(sorted(images)):
        got = np.fromfile(str(tmp_path / ("frame_%06d.u16" % i)), dtype=np.uint16)
        want = np.maximum(images[ts].astype(np.int32) - 64, 0).astype(np.uint16)
        assert np.array_equal(got.reshape(want.shape), want), (i, ts)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/example_dropin not built (needs /root/reference)")
def test_reference_example_over_gpu_decode(container, tmp_path):
    d, path, images, audio = container
    a = tmp_path / "dropin"
    a.mkdir()
    r = _run([DROPIN, path], str(a))
    assert r.returncode == 0, r.stderr + r.stdout
    order = sorted(images)
    for i, ts in enumerate(order):
        dng = (a / ("frame_%06d.dng" % i)).read_bytes()
        assert images[ts].tobytes() in dng, (i, ts)  # one uncompressed strip (example.cpp:80-92)
    assert (a / "audio.wav").stat().st_size == 44 + 2 * 2 * 1920
    if os.path.exists(REFEX):
        b = tmp_path / "ref"
        b.mkdir()
        r2 = _run([REFEX, path], str(b))
        assert r2.returncode == 0, r2.stderr
        assert r.stdout == r2.stdout
        for name in sorted(os.listdir(str(b))):
            assert (a / name).read_bytes() == (b / name).read_bytes(), name


def test_export_tool_into_caller_buffers(container, tmp_path):
    # Decoder::loadFramesInto: the GPU pipeline downloads straight into (pinned) buffers of the caller;
    # same bytes as the vector form, also with the post stage
    d, path, images, audio = container
    if not os.path.exists(EXPORT):
        from motioncam_decoder_amd import build
        build.build_host()
    r = _run([EXPORT, path, "-o", str(tmp_path), "--pinned"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "loadFramesInto" in r.stdout
    for i, ts in enumerate(sorted(images)):
        got = np.fromfile(str(tmp_path / ("frame_%06d.u16" % i)), dtype=np.uint16)
        assert np.array_equal(got.reshape(images[ts].shape), images[ts]), (i, ts)
    r = _run([EXPORT, path, "-o", str(tmp_path), "--pinned", "--black", "--bits", "12"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    for i, ts in enumerate(sorted(images)):
        want = L.oracle_post(images[ts], [64, 64, 64, 64], True)
        got = np.fromfile(str(tmp_path / ("frame_%06d.p12" % i)), dtype=np.uint8)
        assert np.array_equal(got.reshape(want.shape), want), (i, ts)


SWAP = os.path.join(ROOT, "oracle", "_ref", "example_codec_swap")


@pytest.mark.skipif(not (os.path.exists(SWAP) and os.path.exists(REFEX)),
                    reason="oracle/_ref/example_codec_swap not built (needs /root/reference)")
def test_reference_container_code_with_codec_swapped(container, tmp_path):
    # the reference's example.cpp and lib/Decoder.cpp unchanged, only
    # raw::Decode / raw::DecodeLegacy replaced by the C ABI (host/RawData.cpp) -- same files out
    d, path, images, audio = container
    a, b = tmp_path / "swap", tmp_path / "ref"
    a.mkdir()
    b.mkdir()
    r = _run([SWAP, path], str(a))
    assert r.returncode == 0, r.stderr + r.stdout
    r2 = _run([REFEX, path], str(b))
    assert r2.returncode == 0, r2.stderr
    assert r.stdout == r2.stdout
    names = sorted(os.listdir(str(b)))
    assert len(names) == len(images) + 1
    for name in names:
        assert (a / name).read_bytes() == (b / name).read_bytes(), name


def test_export_tool_strip_width_from_white_level(container, tmp_path):
    """--bits auto: the container says whiteLevel 1023 -> 10-bit strips (1.25 bytes per sample over the link)."""
    d, path, images, audio = container
    r = _run([EXPORT, path, "-o", str(tmp_path), "--bits", "auto"], str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "bits per sample: 10" in r.stdout
    for i, ts in enumerate(sorted(images)):
        img = images[ts]
        got = np.fromfile(str(tmp_path / ("frame_%06d.p10" % i)), dtype=np.uint8)
        assert np.array_equal(got.reshape(img.shape[0], -1), L.oracle_post(img, None, bits=10)), (i, ts)


@pytest.mark.parametrize("order", ["0 1 2 3 4", "4 3 2 1 0", "0 0 1 1 4 4", "0 2 4 1 3 0", "1 b 2 3 b 4 0", "3", "4 4 4",
                                   "0 1 2 o12 3 4 o16 0 1 2 o10 3 o14 4", "0 1 2 3 4 0 1 2 3 4 4 3 3 4"])
def test_load_frame_in_any_order(container, order, tmp_path):
    """The facade reads the frame that FOLLOWS a loadFrame() call in the index ahead into pinned memory and decodes the one
    behind the previous call ahead of time (the reference's loop, example.cpp:182-188, asks for the frames one by one in that
    order).  A call gets its own frame, in the form IT asks for, whatever was read or decoded ahead: backwards, repeats, skips,
    batches in between, other output options from one call to the next, the last frame (nothing to read ahead), a lone call
    that leaves a read or a decode pending when the decoder goes away."""
    d, path, images, audio = container
    inc = os.path.join(ROOT, "motioncam_decoder_amd", "host")
    lib = os.path.join(ROOT, "motioncam_decoder_amd", "lib")
    exe = str(d / "frame_order")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(inc, "include"), "-I" + os.path.join(inc, "thirdparty"),
                               "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "frame_order.cpp"),
                               "-L" + lib, "-lmotioncam_decoder", "-lmcraw_hip", "-lpthread", "-Wl,-rpath," + lib])
    r = _run([exe, path] + order.split(), str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    ts = sorted(images)
    want, bits = [], 16
    for tok in order.split():
        if tok[0] == "o":
            bits = int(tok[1:])
            continue
        for i in (range(len(ts)) if tok == "b" else [int(tok)]):
            img = images[ts[i]]
            data = img.tobytes() if bits == 16 or tok == "b" else L.oracle_post(img, None, bits=bits).tobytes()
            line = "%s%d %d %08x" % ("b" if tok == "b" else "", i, len(data), zlib.crc32(data) & 0xFFFFFFFF)
            want.append(line if tok == "b" else line + " %dx%d" % (img.shape[1], img.shape[0]))
    assert r.stdout.split("\n")[:-1] == want


def test_export_tool_one_frame_per_chunk(container, tmp_path):
    """MCRAW_SLOT_MB=1: every chunk of the loadFrames pipeline is one frame, so the copy-outs of neighbouring chunks (sliced over the
    facade's worker threads, which take one caller at a time) are under way together; run a few times, every frame must be its own."""
    d, path, images, audio = container
    env = dict(os.environ, MCRAW_SLOT_MB="1")
    order = sorted(images)
    for rep in range(4):
        r = subprocess.run([EXPORT, path, "-o", str(tmp_path), "--no-write"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("frame ")]
        assert len(lines) == len(order)
        for i, ts in enumerate(order):
            assert ("crc32 %08x" % (zlib.crc32(images[ts].tobytes()) & 0xFFFFFFFF)) in lines[i], (rep, i)


def test_load_frame_random_walk(container, tmp_path):
    """A few hundred loadFrame calls in a seeded random order -- runs of consecutive frames (where the facade decodes ahead), jumps,
    repeats, output options that change in the middle of a run, a batch now and then: every call's bytes are its frame's in the
    form it asked for.  (MCRAW_WALK=<n> for a longer walk.)"""
    import random
    d, path, images, audio = container
    test_load_frame_in_any_order(container, "0", tmp_path)  # (builds the program)
    exe = str(d / "frame_order")
    rng = random.Random(20261003)
    ts = sorted(images)
    toks, i, bits = [], 0, 16
    for step in range(int(os.environ.get("MCRAW_WALK", "300"))):
        r = rng.random()
        if r < 0.60:
            i = (i + 1) % len(ts)  # the reference's loop: the next frame (wrapping at the end)
        elif r < 0.75:
            i = rng.randrange(len(ts))
        elif r < 0.80:
            pass  # the same frame again
        elif r < 0.90:
            bits = rng.choice([16, 16, 12, 10, 14])
            toks.append("o%d" % bits)
            continue
        else:
            toks.append("b")
            continue
        toks.append(str(i))
    crc = {}
    want, bits = [], 16
    for tok in toks:
        if tok[0] == "o":
            bits = int(tok[1:])
            continue
        for k in (range(len(ts)) if tok == "b" else [int(tok)]):
            key = (k, 16 if tok == "b" else bits)
            if key not in crc:
                img = images[ts[k]]
                data = img.tobytes() if key[1] == 16 else L.oracle_post(img, None, bits=key[1]).tobytes()
                crc[key] = (len(data), zlib.crc32(data) & 0xFFFFFFFF)
            img = images[ts[k]]
            line = "%s%d %d %08x" % ("b" if tok == "b" else "", k, crc[key][0], crc[key][1])
            want.append(line if tok == "b" else line + " %dx%d" % (img.shape[1], img.shape[0]))
    r = _run([exe, path] + toks, str(tmp_path))
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    got = r.stdout.split("\n")[:-1]
    assert len(got) == len(want)
    bad = [(n, g, w) for n, (g, w) in enumerate(zip(got, want)) if g != w]
    assert not bad, bad[:3]


def test_load_frame_walks_past_a_frame_that_does_not_decode(container, tmp_path):
    """Frame 2 of 5 is damaged (its side streams cut off).  Decoded ahead of its call it fails quietly; the call FOR it then
    takes the ordinary path and throws what the reference throws, the calls around it get their frames -- forwards (where the
    damaged frame is decoded ahead), backwards, and asked for twice."""
    d, path, images, audio = container
    test_load_frame_in_any_order(container, "0", tmp_path)  # (builds the program)
    exe = str(d / "frame_order")
    ts = sorted(images)
    specs = []
    for n, t in enumerate(ts):
        img = images[t]
        buf = L.encode7(img)
        if n == 2:
            buf = buf[:len(buf) // 2].copy()  # (the bits / refs streams lie behind the payload: gone)
        specs.append((t, 7, img.shape[1], img.shape[0], buf))
    bad = L.write_mcraw(str(tmp_path / "damaged.mcraw"), specs)
    r = _run([exe, bad] + "0 1 2 3 4 3 2 2 1 0 1 2 3".split(), str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.split("\n")[:-1]
    for line, i in zip(got, [0, 1, 2, 3, 4, 3, 2, 2, 1, 0, 1, 2, 3]):
        if i == 2:
            assert line == "2 failed: Failed to uncompress frame", line
        else:
            img = images[ts[i]]
            assert line == "%d %d %08x %dx%d" % (i, img.nbytes, zlib.crc32(img.tobytes()) & 0xFFFFFFFF, img.shape[1], img.shape[0]), line
    assert len(got) == 13
