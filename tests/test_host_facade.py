"""CPU: the C++ host facade (motioncam::Decoder over the C ABI) builds, exports the reference's
public API, and the reference's own example.cpp compiles UNCHANGED against this repository's
headers (drop-in contract).  Running it needs a GPU: tests/test_gpu_dropin.py."""
import os
import subprocess

import pytest

from motioncam_decoder_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "motioncam_decoder_amd", "lib")


@pytest.fixture(scope="module")
def host_lib():
    build.build_hip()
    return build.build_host()


def test_facade_exports_reference_api(host_lib):
    syms = subprocess.run(["nm", "-DC", "--defined-only", host_lib], capture_output=True, text=True, check=True).stdout
    for want in ("motioncam::Decoder::Decoder(std::", "motioncam::Decoder::Decoder(_IO_FILE*)",
                 "motioncam::Decoder::~Decoder()", "motioncam::Decoder::getFrames() const",
                 "motioncam::Decoder::getContainerMetadata", "motioncam::Decoder::loadFrame(long",
                 "motioncam::Decoder::loadFrames(", "motioncam::Decoder::audioSampleRateHz() const",
                 "motioncam::Decoder::numAudioChannels() const", "motioncam::Decoder::loadAudio(std::vector",
                 "motioncam::Decoder::loadAudio() const", "motioncam::raw::Decode(unsigned short*, int, int, unsigned char const*, unsigned long)",
                 "motioncam::raw::DecodeLegacy(unsigned short*, int, int, unsigned char const*, unsigned long)"):
        assert want in syms, want
    assert os.path.exists(os.path.join(LIB, "mcraw_export"))


def test_facade_has_no_cpu_codec(host_lib):
    # the facade forwards to the C ABI; it must not carry a decoder of its own
    und = subprocess.run(["nm", "-DC", "--undefined-only", host_lib], capture_output=True, text=True, check=True).stdout
    assert "mcraw_decode7" in und and "mcraw_decode6" in und and "mcraw_pool_decode_batch" in und
    for f in ("Decoder.cpp", "RawData.cpp"):
        src = open(os.path.join(ROOT, "motioncam_decoder_amd", "host", f)).read()
        assert "oracle" not in src.lower() and "simde" not in src.lower()


@pytest.mark.skipif(not os.path.exists("/root/reference/example.cpp"), reason="reference checkout absent")
def test_reference_example_compiles_unchanged(host_lib):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dropin"], check=True)
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "example_dropin"))
    und = subprocess.run(["nm", "-DC", "--undefined-only", os.path.join(ROOT, "oracle", "_ref", "example_dropin")],
                         capture_output=True, text=True, check=True).stdout
    assert "mcraw_decode7" in und  # its frames go through the HIP library, not a CPU codec


@pytest.fixture(scope="module")
def probe(host_lib, tmp_path_factory):
    d = tmp_path_factory.mktemp("probe")
    exe = str(d / "facade_probe")
    host = os.path.join(ROOT, "motioncam_decoder_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(host, "include"), "-I" + os.path.join(host, "thirdparty"),
                    "-o", exe, os.path.join(ROOT, "tests", "cpp", "facade_probe.cpp"), "-L" + LIB, "-lmotioncam_decoder",
                    "-lmcraw_hip", "-Wl,-rpath," + LIB], check=True)
    return exe, d


def _probe(exe, *args):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout.splitlines()


def test_container_reader_matches_reference_contract(probe):
    import numpy as np
    import _libs as L
    exe, d = probe
    img = L.natural_image_np(128, 8, 12, 12.0, 1)
    frames = [(3000, 7, 128, 8, L.encode7(img)), (1000, 6, 128, 8, L.encode6(img)), (2000, 7, 128, 8, L.encode7(img))]
    audio = [(111, np.arange(100, dtype=np.int16)), (None, np.full(60, 2, np.int16))]
    good = L.write_mcraw(str(d / "good.mcraw"), frames, audio, audio_rate=44100, audio_channels=1)
    rc, out = _probe(exe, good, "decode")
    assert rc == 0
    assert out[0] == "frames 1000 2000 3000"                       # sorted by timestamp (lib/Decoder.cpp:266-279)
    assert out[1] == "camera rggb 44100 1"
    assert out[2] == "audio 111 100 4950" and out[3] == "audio -1 60 120"   # timestamp -1 without metadata (:62-70)
    assert out[4] == "loader 2"
    assert out[5] == "missing: Frame not found (timestamp: 123456789)"      # :186
    import torch
    if not torch.cuda.is_available():
        assert out[6].startswith("decode: Failed to uncompress legacy frame")  # no GPU: fails, never decodes on the CPU
    else:
        assert out[6] == "decoded 2048 128x8"

    raw = open(good, "rb").read()
    cases = {"missing.mcraw": (None, "error: Failed to open "),
             "version.mcraw": (raw[:7] + bytes([2]) + raw[8:], "error: Invalid container version"),
             "ident.mcraw": (b"NOTION " + raw[7:], "error: Invalid header id"),
             "tail.mcraw": (raw[:-24] + bytes([9, 0, 0, 0]) + raw[-20:], "error: Invalid file"),
             "magic.mcraw": (raw[:-16] + bytes(4) + raw[-12:], "error: Corrupted file"),
             "short.mcraw": (raw[:20], "error: ")}
    for name, (content, want) in cases.items():
        p = str(d / name)
        if content is not None:
            open(p, "wb").write(content)
        rc, out = _probe(exe, p)
        assert rc == 1 and out and out[0].startswith(want), (name, out)


def test_reference_container_vocabulary_compiles_and_reads_a_file(tmp_path):
    """host/include/motioncam/Container.hpp is source compatible with the reference header of that name
    (lib/include/motioncam/Container.hpp:22-72): a reader written with the reference's type, member, enumerator
    and constant names (tests/cpp/container_compat.cpp) compiles against it and walks a synthetic file."""
    import numpy as np
    import _libs as L
    exe = str(tmp_path / "container_compat")
    host = os.path.join(ROOT, "motioncam_decoder_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(host, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "container_compat.cpp")], check=True)
    frames = []
    for i, ts in enumerate((3000, 1000, 2000)):
        img = L.natural_image_np(128, 8, 12, 12.0, 40 + i)
        frames.append((ts, 7 if i != 1 else 6, 128, 8, L.encode7(img) if i != 1 else L.encode6(img)))
    audio = [(111, np.arange(64, dtype=np.int16)), (None, np.arange(64, dtype=np.int16))]
    path = L.write_mcraw(str(tmp_path / "c.mcraw"), frames, audio)
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert r.stdout.strip() == "frames 3 audio 2 first_ts 1000"


def test_copy_out_workers_under_thread_sanitizer(tmp_path):
    """The threads Decoder::loadFrame's copy-out keeps (host/WorkerPool.hpp) take one caller at a time; two chunks' copy-outs call
    them together.  tests/cpp/worker_pool_tsan.cpp: three callers, every run size around the worker count, under -fsanitize=thread."""
    exe = str(tmp_path / "worker_pool_tsan")
    host = os.path.join(ROOT, "motioncam_decoder_amd", "host")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=thread", "-I" + host, "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "worker_pool_tsan.cpp"), "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if "FATAL: ThreadSanitizer" in r.stderr:  # (a kernel whose address-space layout this sanitizer runtime cannot map: not a finding)
        pytest.skip("ThreadSanitizer does not start here: " + r.stderr.strip().splitlines()[0])
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (r.stdout, r.stderr[-2000:])
    assert r.stdout.strip() == "wrong 0"
