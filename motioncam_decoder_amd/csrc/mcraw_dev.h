// mcraw_dev.h -- device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mcraw_plan.h"

namespace mcraw {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Storage length in bytes of a 64-sample block coded at `b` bits
// (lib/RawData.cpp:27-45: 0,8,..,48,64,64,80,80,128...).
__device__ __forceinline__ uint32_t len7_of(uint32_t b)
{
    return b <= 6u ? 8u * b : (b <= 8u ? 64u : (b <= 10u ? 80u : 128u));
}

// Row of the term table for `b` bits: 0..6 -> themselves, 7/8 -> 7 (Decode8),
// 9/10 -> 8 (Decode10), >= 11 -> 9 (raw 16, no table row) (RawData.cpp:424-458).
__device__ __forceinline__ uint32_t cls7_of(uint32_t b)
{
    return b <= 6u ? b : (b <= 8u ? 7u : (b <= 10u ? 8u : 9u));
}

// Bounds-checked byte-buffer descriptor over one frame buffer: reads past
// `len` return 0 instead of faulting (corrupt offsets cannot leave the frame).
// The range check of a raw buffer works on whole dwords, so the record count is
// `len` rounded up to 4: the last 1-3 bytes of an odd-sized frame stay readable
// (the dword that holds them lies inside the 4-byte aligned allocation).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frame_rsrc(const uint8_t *in, uint32_t len)
{
    // descriptor inputs must be provably wave-uniform (no waterfall loops)
    uint64_t a = reinterpret_cast<uint64_t>(in);
    uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    uint32_t n = __builtin_amdgcn_readfirstlane((len + 3u) & ~3u);
    void *p = reinterpret_cast<void *>((static_cast<uint64_t>(hi) << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, static_cast<int>(n), 0x00020000);
}

__device__ __forceinline__ uint32_t ld_u8(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    return __builtin_amdgcn_raw_buffer_load_b8(r, off, 0, 0);
}

__device__ __forceinline__ uint4 ld_b128(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// Largest f with base[f] <= item (base has n+1 ascending entries, base[0] = 0).
__device__ __forceinline__ int find_frame(uint32_t item, const uint32_t *__restrict__ base, int n)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (base[mid] <= item)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Consecutive logical work items on one XCD: workgroups are dealt round-robin
// over the 8 XCDs, so blocks b and b+8 share an L2.  Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
}

// Exclusive prefix sum over the 64 lanes (all lanes must be active) and the wave total.
// DPP form (no LDS traffic): Hillis-Steele inside each 16-lane row with row_shr:1,2,4,8, then
// row_bcast:15 / row_bcast:31 carry the row totals across (gfx9 DPP controls).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane, uint32_t *total)
{
    (void)lane;
    uint32_t inc = v;
#define MCRAW_DPP_ADD(ctrl, rowmask)                                                                                   \
    inc += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(inc), ctrl, rowmask, 0xf, true))
    MCRAW_DPP_ADD(0x111, 0xf); // row_shr:1
    MCRAW_DPP_ADD(0x112, 0xf); // row_shr:2
    MCRAW_DPP_ADD(0x114, 0xf); // row_shr:4
    MCRAW_DPP_ADD(0x118, 0xf); // row_shr:8
    MCRAW_DPP_ADD(0x142, 0xa); // row_bcast:15 -> rows 1 and 3
    MCRAW_DPP_ADD(0x143, 0xc); // row_bcast:31 -> rows 2 and 3
#undef MCRAW_DPP_ADD
    *total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(inc), 63));
    return inc - v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}

} // namespace mcraw
