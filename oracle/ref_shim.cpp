// ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doorway into the REAL reference
// codec, compiled by oracle/Makefile from the sources where they lie under
// /root/reference (lib/RawData.cpp, lib/RawData_Legacy.cpp + vendored SIMDe).
// Nothing of the reference is copied: this file only calls its public seam
// (lib/include/motioncam/RawData.hpp:25-37).  Output: oracle/_ref/libmcraw_ref.so
// (git-ignored).  Used to pin oracle/mcraw_oracle.c, to generate tests/golden/,
// and as bench.py's cpu_baseline "reference" leg.
#include <motioncam/RawData.hpp>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" {

size_t mcraw_ref_decode7(uint16_t *out, int width, int height, const uint8_t *in, size_t len)
{
    return motioncam::raw::Decode(out, width, height, in, len);
}

size_t mcraw_ref_decode6(uint16_t *out, int width, int height, const uint8_t *in, size_t len)
{
    return motioncam::raw::DecodeLegacy(out, width, height, in, len);
}

// Frame-parallel timing (one frame per task, private outputs, inputs shared):
// seconds of wall time for `reps` passes over `nframes` buffers, <0 on failure.
double mcraw_ref_time_batch(int type, int width, int height, const uint8_t *const *ins,
                            const size_t *lens, int nframes, int nthreads, int reps)
{
    if (nthreads < 1)
        nthreads = 1;
    std::vector<std::thread> th;
    std::vector<int> err(nthreads, 0);
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nthreads; t++)
        th.emplace_back([=, &err]() {
            // type 7 writes width*encodedHeight; callers pass height % 4 == 0
            std::vector<uint16_t> out((size_t)width * ((size_t)height + 4));
            for (int r = 0; r < reps; r++)
                for (int f = t; f < nframes; f += nthreads) {
                    size_t n = type == 7 ? motioncam::raw::Decode(out.data(), width, height, ins[f], lens[f])
                                         : motioncam::raw::DecodeLegacy(out.data(), width, height, ins[f], lens[f]);
                    if (n == 0)
                        err[t] = 1;
                }
        });
    for (auto &x : th)
        x.join();
    auto t1 = std::chrono::steady_clock::now();
    for (int e : err)
        if (e)
            return -1.0;
    return std::chrono::duration<double>(t1 - t0).count();
}
}
