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
    assert "mcraw_decode7" in und and "mcraw_decode6" in und and "mcraw_decode_batch" in und
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
