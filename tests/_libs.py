"""Doors for the tests: the build's encoder / image generator (motioncam_decoder_amd/synthlib.py) and the checkers
(oracle/doors.py: the oracle and, where it was built, the real reference).  TEST INFRASTRUCTURE.

The product library (libmcraw_hip.so) is loaded by motioncam_decoder_amd itself."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from motioncam_decoder_amd.synthlib import *  # noqa: F401,F403,E402
from motioncam_decoder_amd.synthlib import _ptr, _strip_bits  # noqa: F401,E402
from doors import *  # noqa: F401,F403,E402
from doors import ORACLE_DIR, _decode, _cpu_has  # noqa: F401,E402
from motioncam_decoder_amd.synthlib import SYNTH_DIR  # noqa: F401,E402
