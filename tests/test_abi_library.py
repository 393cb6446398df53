"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mcraw_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import motioncam_decoder_amd as M
from motioncam_decoder_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_hip()
    return M.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mcraw_hip.h")).read()
    declared = set(re.findall(r"\b(mcraw_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(M.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_code_object_is_gfx950():
    raw = open(M.lib_path(), "rb").read()
    assert b"gfx950" in raw
    for k in (b"k7_tiles", b"k7_side", b"k6_decode"):
        assert k in raw, k


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.mcraw_ctx_create(0, ctypes.byref(h)) != 0
    assert not h.value
    assert b"no CPU fallback" in lib.mcraw_last_error() or lib.mcraw_last_error()
    with pytest.raises(M.McrawError):
        M.Context(0)
    # the drop-in single-frame entry fails (returns 0) instead of decoding on the CPU
    import numpy as np
    out = np.zeros(128 * 8, np.uint16)
    buf = np.zeros(64, np.uint8)
    assert lib.mcraw_decode7(out.ctypes.data, 128, 8, buf.ctypes.data, buf.size) == 0


def test_product_does_not_link_oracle():
    # the shipped library must not reference the oracle or the reference build
    raw = open(M.lib_path(), "rb").read()
    assert b"mcraw_oracle" not in raw and b"mcraw_ref_" not in raw
    for f in os.listdir(os.path.join(ROOT, "motioncam_decoder_amd", "csrc")):
        src = open(os.path.join(ROOT, "motioncam_decoder_amd", "csrc", f)).read()
        assert "oracle" not in src.lower(), f


def _build_c_probe(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_probe")
    lib_dir = os.path.dirname(M.lib_path())
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_probe.c"), "-L" + lib_dir, "-lmcraw_hip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    return exe


def test_header_is_plain_c_and_links_from_c(lib, tmp_path):
    # include/mcraw_hip.h compiles as C99 with -pedantic -Werror; a C program links the library and, on a
    # machine without a GPU, every call fails cleanly (no CPU fallback)
    import subprocess
    import torch
    exe = _build_c_probe(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the decode leg runs in test_c_program_decodes")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ctx=no" in r.stdout and "decode7 returned 0" in r.stdout and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_c_program_decodes(lib, tmp_path):
    import subprocess
    exe = _build_c_probe(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ctx=yes" in r.stdout and "decode7 returned 256 first=0" in r.stdout
    assert "set_post rc=0" in r.stdout and "set_post(NULL) rc=0" in r.stdout


def test_tile_order_is_a_permutation_for_every_grid_and_run_length(lib):
    """k7_tiles' block -> logical workgroup mapping (mcraw_tile_order; runs of c workgroups per XCD, or the grid in eight
    parts): every logical workgroup exactly once, whatever the grid size -- a gap or a repeat would be pixels never
    written or written twice.  Within a full run group, XCD x (blocks x, x + 8, ...) gets c consecutive workgroups."""
    import numpy as np
    f = lib.mcraw_tile_order
    for runs in (0, 1, 3, 8, 64, 128, 512):
        for n in (1, 2, 7, 8, 9, 63, 64, 65, 1023, 1024, 1025, 4 * 1024 + 5, 8 * 128 * 3, 8 * 128 * 3 + 77):
            got = np.array([f(b, n, runs) for b in range(n)], dtype=np.int64)
            assert got.min() == 0 and got.max() == n - 1 and np.unique(got).size == n, (runs, n)
    n, c = 8 * 128 * 4, 128
    for x in range(8):
        mine = [f(b, n, c) for b in range(x, 8 * c, 8)]   # XCD x's blocks of the first group
        assert mine == list(range(x * c, (x + 1) * c))
