"""Build the native parts of the MCRAW decode path, in-tree.

    python -m motioncam_decoder_amd.build            # everything
    python -m motioncam_decoder_amd.build hip synth  # selected targets
    python -m motioncam_decoder_amd.build variant /tmp/libx.so -DMCRAW_DIAG   # another build of the HIP sources (tests, tools)

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


# The gfx950 library: kernels (mcraw_type7 / mcraw_type6), the host side of the C ABI in its units (csrc/mcraw_host.h) and the device pool.
HIP_SOURCES = ("mcraw_abi.hip", "mcraw_submit.hip", "mcraw_tune.hip", "mcraw_device.hip", "mcraw_hostmem.hip", "mcraw_pool.hip",
               "mcraw_type7.hip", "mcraw_type6.hip")
HIP_HEADERS = ("mcraw_plan.h", "mcraw_dev.h", "mcraw_host.h")
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]


def hip_sources():
    return [os.path.join(CSRC, f) for f in HIP_SOURCES]


def _objects(flags=(), tag="", force=False, only=None):
    """One object per source under lib/obj/ (rebuilt when the source or a header is newer), compiled side by side."""
    obj = os.path.join(LIB, "obj")
    os.makedirs(obj, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in HIP_HEADERS] + [os.path.join(ROOT, "include", "mcraw_hip.h")]
    objs, jobs = [], []
    for src in hip_sources():
        if only is not None and src not in only:
            continue
        o = os.path.join(obj, os.path.basename(src)[:-4] + tag + ".o")
        objs.append(o)
        if force or _newer(o, [src] + hdrs):
            cmd = [HIPCC] + HIP_FLAGS + ["-Wall", "-Wno-unused-function"] + list(flags) + ["-c", "-o", o, src]
            print("+", " ".join(cmd), flush=True)
            jobs.append((subprocess.Popen(cmd), cmd))
    for pr, cmd in jobs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return objs, bool(jobs)


def _link(out, objs):
    _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out] + objs + ["-lpthread"])


def build_variant(out, flags=()):
    """Another build of the same sources (tests and tools: fault injection, forced paths, diagnostics).  Only the sources that a
    flag can reach are compiled again -- those that mention a -D name themselves, all of them when a header does or when a flag
    is no -D --; the others' objects are the product library's."""
    import hashlib
    import re
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    names = [re.sub(r"^-D([A-Za-z0-9_]+).*$", r"\1", f) for f in flags if f.startswith("-D")]
    everything = len(names) != len(flags) or any(n in open(os.path.join(CSRC, h)).read() for n in names for h in HIP_HEADERS)
    touched = [s for s in hip_sources() if everything or any(n in open(s).read() for n in names)]
    tag = ".v" + hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    plain, _ = _objects(only=[s for s in hip_sources() if s not in touched])
    special, _ = _objects(flags, tag, only=touched)
    _link(out, plain + special)
    return out


# The -D builds that tests/ links on the GPU box (tests/test_gpu_split.py, test_gpu_segw.py, test_gpu_lookback_fault.py).
TEST_VARIANTS = (["-DMCRAW_FORCE_SEGW"], ["-DMCRAW_INJECT_MUTE7"], ["-DMCRAW_INJECT_LOST"])


def prebuild_test_variants():
    """Compile the test variants' objects into lib/obj/ ahead of time: build_variant() on the GPU box then only links."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for i, flags in enumerate(TEST_VARIANTS):
            build_variant(os.path.join(d, "v%d.so" % i), flags)


def build_hip(force=False):
    """lib/libmcraw_hip.so: one object per source (lib/obj/), then the link."""
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libmcraw_hip.so")
    if os.environ.get("MCRAW_DIAG"):  # timing-experiment kernels (tools/abl7.sh) in place of the product library
        objs, built = _objects(["-DMCRAW_DIAG"], ".diag", force)
    else:
        objs, built = _objects(force=force)
    if force or built or _newer(out, objs):
        _link(out, objs)
    return out


def build_timeline(force=False):
    """The diagnostic build of the host-memory pipeline (tools/timeline_host.sh): events with timing, one line per sub-batch when
    it is drained.  Not the product: lib/timeline/libmcraw_hip.so, picked up through LD_LIBRARY_PATH or MCRAW_LIB_PATH."""
    out = os.path.join(LIB, "timeline", "libmcraw_hip.so")
    if force or _newer(out, hip_sources() + [os.path.join(CSRC, f) for f in HIP_HEADERS]):
        build_variant(out, ["-Wall", "-Wno-unused-function", "-DMCRAW_TIMELINE"])
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
    if len(sys.argv) > 2 and sys.argv[1] == "variant":  # python -m motioncam_decoder_amd.build variant <out.so> [compiler flags ...]
        build_variant(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    build_all(force="--force" in sys.argv, targets=tuple(args) or ("hip", "synth", "host"))
