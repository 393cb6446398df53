"""CPU: the container reader of the facade (host/Decoder.cpp) under AddressSanitizer + UBSan on mutated
and truncated .mcraw files.  A corrupt file may be rejected (exit 1, IOException text) or parsed, or end
in an uncaught nlohmann::json exception (the reference lets those propagate too) -- but never in a
sanitizer report, a crash, or a hang (a corrupted row count used to turn into a 16 GB read)."""
import os
import subprocess

import numpy as np
import pytest

import _libs as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "motioncam_decoder_amd", "host")


def _asan_runtime():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="libasan not available")
def test_container_reader_survives_mutated_files(tmp_path):
    from motioncam_decoder_amd import build
    import motioncam_decoder_amd as M
    build.build_hip()
    lib_dir = os.path.dirname(M.lib_path())
    exe = str(tmp_path / "probe_asan")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17",
                    "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(HOST, "thirdparty"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_probe.cpp"),
                    os.path.join(HOST, "Decoder.cpp"), os.path.join(HOST, "RawData.cpp"),
                    "-L" + lib_dir, "-lmcraw_hip", "-lpthread", "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    frames = []
    for i in range(4):
        img = L.natural_image_np(128, 16, 12, 12.0, i)
        frames.append((1000 + i, 7 if i % 2 == 0 else 6, 128, 16, L.encode7(img) if i % 2 == 0 else L.encode6(img)))
    audio = [(111, np.arange(480, dtype=np.int16)), (None, np.arange(480, dtype=np.int16))]
    raw = np.fromfile(L.write_mcraw(str(tmp_path / "ok.mcraw"), frames, audio), dtype=np.uint8)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    rng = np.random.default_rng(5)
    outcomes = {}
    for trial in range(240):
        b = raw.copy()
        mode = trial % 4
        if mode == 0:    # byte flips anywhere
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, b.size))] = rng.integers(0, 256)
        elif mode == 1:  # truncation
            b = b[: int(rng.integers(1, b.size))]
        elif mode == 2:  # the index and audio tables at the end
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(max(0, b.size - 200), b.size))] = rng.integers(0, 256)
        else:            # a 32-bit field blown up
            pos = int(rng.integers(0, b.size - 4))
            b[pos:pos + 4] = np.frombuffer(np.uint32(rng.integers(0, 2 ** 32)).tobytes(), np.uint8)
        p = str(tmp_path / "m.mcraw")
        b.tofile(p)
        r = subprocess.run([exe, p, "x"], capture_output=True, env=env, timeout=120)  # a hang fails the test here
        err = r.stderr.decode("utf-8", "replace")
        assert "AddressSanitizer" not in err and "runtime error" not in err, (trial, mode, err[-800:])
        if r.returncode not in (0, 1):
            assert "nlohmann" in err and "terminate called" in err, (trial, mode, r.returncode, err[-800:])
        outcomes[r.returncode] = outcomes.get(r.returncode, 0) + 1
    assert outcomes.get(0, 0) > 20 and outcomes.get(1, 0) > 20, outcomes
