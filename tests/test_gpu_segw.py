"""GPU (-m gpu): k7_side's segment walkers on every input of the parity, fuzz and negative suites.

In the product library a side stream is followed by run speculation and handed to the segment walkers only when its
records keep changing size (uniform-noise frames).  A second build of the same sources with -DMCRAW_FORCE_SEGW puts EVERY
stream on the walkers from its first record on; the parity, fuzz, negative and encoder-variant tests are then run against
that library in a child process (MCRAW_LIB_PATH), so that the walkers -- lanes that start on payload bytes, verification
lane against lane, repair rounds, piece and unit boundaries, dead chains -- see every stream shape those suites hold."""
import os
import shutil
import subprocess
import sys

import pytest

from motioncam_decoder_amd import build as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (the two tests that start bench.py's ranks as grandchildren take 16 s a run and add nothing to what a variant of k7_side is asked here)
NOT_HERE = ["--deselect", os.path.join(ROOT, "tests", "test_gpu_properties.py") + "::test_bench_multi_rank_path_on_one_gpu",
            "--deselect", os.path.join(ROOT, "tests", "test_gpu_properties.py") + "::test_frame_checksums_do_not_depend_on_the_gpu_count"]


def test_all_type7_suites_with_every_stream_on_the_segment_walkers(tmp_path):
    lib = str(tmp_path / "libmcraw_segw.so")
    B.build_variant(lib, ['-DMCRAW_FORCE_SEGW'])
    env = dict(os.environ, MCRAW_LIB_PATH=lib)
    suites = ["test_gpu_parity.py", "test_gpu_fuzz.py", "test_gpu_negative.py", "test_encoder_variants.py", "test_gpu_properties.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + NOT_HERE
                       + [os.path.join(ROOT, "tests", s) for s in suites], env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail + r.stderr[-2000:]
