"""GPU (-m gpu): k7_side with every side stream cut into parts (round 4; round 5: the parts' decode replays their count).

A long side stream of a small batch is resolved by several workgroups: each owns a range of the stream's 32 KiB pieces, counts
the records of its pieces with the walker alone (from a speculative start, for the parts in the middle), is told by the part in
front where the chain really enters its pieces and with which record, and decodes; payload offsets are relative to the part
and k7_tiles adds the earlier parts' totals.  The library picks the number of parts from the batch (MCRAW_SIDE_SPLIT pins it);
here the parity, fuzz, negative, encoder-variant and property suites are run in child processes with 2 and 4 parts forced on
EVERY stream -- short ones, whose later parts own nothing, included --, once more against the build that puts every stream
on the segment walkers, and against a build in which the first part of every stream never speaks (the others give up waiting
and follow the chain from the stream's first record by themselves)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from motioncam_decoder_amd import build as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["test_gpu_parity.py", "test_gpu_fuzz.py", "test_gpu_negative.py", "test_encoder_variants.py", "test_gpu_properties.py"]

# (the two tests that start bench.py's ranks as grandchildren take 16 s a run and add nothing to what a variant of k7_side is asked here)
NOT_HERE = ["--deselect", os.path.join(ROOT, "tests", "test_gpu_properties.py") + "::test_bench_multi_rank_path_on_one_gpu",
            "--deselect", os.path.join(ROOT, "tests", "test_gpu_properties.py") + "::test_frame_checksums_do_not_depend_on_the_gpu_count"]


def _run(env):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + NOT_HERE
                       + [os.path.join(ROOT, "tests", s) for s in SUITES], env=env, capture_output=True, text=True, timeout=1200)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail + r.stderr[-2000:]


@pytest.mark.parametrize("parts", [2, 4])
def test_all_type7_suites_with_every_side_stream_in_parts(parts):
    _run(dict(os.environ, MCRAW_SIDE_SPLIT=str(parts)))


def test_parts_whose_last_one_does_not_count():
    """Round 5: a part's decode replays the record positions its count left behind, and the LAST part of a stream counts too while
    the chip has room for it (the library decides by the number of workgroups; the suites above run with it, their batches are
    small).  Here the other instance of the kernel: the last part waits for the part in front and follows the chain itself."""
    _run(dict(os.environ, MCRAW_SIDE_SPLIT="3,2", MCRAW_SIDE_LASTC="0"))


def _build(tmp_path, flag):
    lib = str(tmp_path / ("libmcraw_%s.so" % flag.lower()))
    B.build_variant(lib, ["-D" + flag])
    return lib


def test_parts_on_the_segment_walkers(tmp_path):
    _run(dict(os.environ, MCRAW_LIB_PATH=_build(tmp_path, "MCRAW_FORCE_SEGW"), MCRAW_SIDE_SPLIT="4"))


def test_parts_whose_predecessor_never_speaks(tmp_path):
    _run(dict(os.environ, MCRAW_LIB_PATH=_build(tmp_path, "MCRAW_INJECT_MUTE7"), MCRAW_SIDE_SPLIT="4"))


def test_the_library_splits_long_streams_of_small_batches_by_itself():
    """One 8K frame and one 12 MP noise frame: the default rule cuts their streams in two; the result equals the oracle's, and a
    frame whose refs stream is cut short is still reported."""
    import torch

    import _libs as L
    import motioncam_decoder_amd as M

    dev = torch.device("cuda:0")
    ctx = M.Context(0)
    imgs = [L.synth_image(7680, 4320, 12, 1, 12.0, 77), L.synth_image(4032, 3024, 14, 0, 0.0, 78)]
    bufs = [L.encode7(im) for im in imgs]
    cut = bufs[0][: bufs[0].size - 40000].copy()  # the refs stream is cut short: the chain ends in front of its last record
    ins = [torch.from_numpy(b).to(dev) for b in bufs + [cut]]
    dims = [(7680, 4320), (4032, 3024), (7680, 4320)]
    outs = [torch.zeros(w * h * 2, dtype=torch.uint8, device=dev) for w, h in dims]
    fr = M.Context.make_frames([(ins[i].data_ptr(), ins[i].numel(), dims[i][0], dims[i][1], 7, outs[i].data_ptr(), dims[i][0] * dims[i][1])
                                for i in range(3)])
    written, status = ctx.decode_batch(fr)
    assert status[0] == 0 and status[1] == 0, status
    for i in range(2):
        w, h = dims[i]
        assert written[i] == w * h
        assert np.array_equal(outs[i].cpu().numpy().view(np.uint16).reshape(h, w), imgs[i])
    ret, _ = L.oracle_decode7(cut, 7680, 4320)
    assert (status[2] != 0) == (ret == 0), (status[2], ret)
    ctx.close()
