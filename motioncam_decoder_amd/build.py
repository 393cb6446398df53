"""Build the native parts of the MCRAW decode path, in-tree.

    python -m motioncam_decoder_amd.build            # everything
    python -m motioncam_decoder_amd.build hip synth  # selected targets

Targets
  hip    motioncam_decoder_amd/lib/libmcraw_hip.so      gfx950 kernels + C ABI (hipcc)
  synth  motioncam_decoder_amd/synth/libmcraw_synth.so  encoder / image generator (gcc, CPU)
  host   motioncam_decoder_amd/lib/libmotioncam_decoder.so + mcraw_export   C++ facade (g++)

hipcc cross-compiles for gfx950 without a GPU.  The .so files are git-ignored
but travel to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")
CSRC = os.path.join(PKG, "csrc")
SYNTH = os.path.join(PKG, "synth")
HOST = os.path.join(PKG, "host")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmcraw_hip.so")
    srcs = [os.path.join(CSRC, f) for f in ("mcraw_abi.hip", "mcraw_pool.hip", "mcraw_type7.hip", "mcraw_type6.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("mcraw_plan.h", "mcraw_dev.h")] + [
        os.path.join(ROOT, "include", "mcraw_hip.h")]
    if force or _newer(out, deps):
        diag = ["-DMCRAW_DIAG"] if os.environ.get("MCRAW_DIAG") else []  # timing-experiment kernels (tools/abl7.sh)
        _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
              "-Wall", "-Wno-unused-function"] + diag + ["-o", out] + srcs + ["-lpthread"])
    return out


def build_timeline(force=False):
    """The diagnostic build of the host-memory pipeline (tools/timeline_host.sh): events with timing, one line per sub-batch when
    it is drained.  Not the product: lib/timeline/libmcraw_hip.so, picked up through LD_LIBRARY_PATH or MCRAW_LIB_PATH."""
    d = os.path.join(LIB, "timeline")
    os.makedirs(d, exist_ok=True)
    out = os.path.join(d, "libmcraw_hip.so")
    srcs = [os.path.join(CSRC, f) for f in ("mcraw_abi.hip", "mcraw_pool.hip", "mcraw_type7.hip", "mcraw_type6.hip")]
    if force or _newer(out, srcs + [os.path.join(CSRC, f) for f in ("mcraw_plan.h", "mcraw_dev.h")]):
        _run([HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
              "-DMCRAW_TIMELINE", "-o", out] + srcs + ["-lpthread"])
    return out


def build_synth(force=False):
    out = os.path.join(SYNTH, "libmcraw_synth.so")
    src = os.path.join(SYNTH, "mcraw_synth.c")
    if force or _newer(out, [src]):
        _run(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c11", "-Wall", "-Wextra", "-shared", "-o", out, src, "-lm"])
    return out


def build_host(force=False):
    """C++ facade (motioncam::Decoder over the C ABI) and the export tool."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmotioncam_decoder.so")
    srcs = [os.path.join(HOST, f) for f in ("Decoder.cpp", "RawData.cpp", "Writer.cpp")]
    if not all(os.path.exists(s) for s in srcs):
        return None
    hdrs = [os.path.join(HOST, "include", "motioncam", f) for f in ("Decoder.hpp", "Container.hpp", "RawData.hpp", "Writer.hpp",
                                                                     "mcraw_container.h")]
    hdrs.append(os.path.join(HOST, "WorkerPool.hpp"))
    inc = ["-I" + os.path.join(HOST, "include"), "-I" + os.path.join(HOST, "thirdparty"), "-I" + os.path.join(ROOT, "include")]
    if force or _newer(out, srcs + hdrs):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + inc + ["-o", out] + srcs +
             ["-L" + LIB, "-lmcraw_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    tool = os.path.join(LIB, "mcraw_export")
    tsrc = os.path.join(HOST, "mcraw_export.cpp")
    if os.path.exists(tsrc) and (force or _newer(tool, [tsrc, out])):
        _run(["g++", "-O2", "-std=c++17", "-Wall"] + inc + ["-o", tool, tsrc, "-L" + LIB, "-lmotioncam_decoder",
             "-lmcraw_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(force=False, targets=("hip", "synth", "host")):
    res = {}
    if "hip" in targets:
        res["hip"] = build_hip(force)
    if "synth" in targets:
        res["synth"] = build_synth(force)
    if "host" in targets:
        res["host"] = build_host(force)
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    build_all(force="--force" in sys.argv, targets=tuple(args) or ("hip", "synth", "host"))
