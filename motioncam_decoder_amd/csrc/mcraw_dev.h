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

// 0x00 / 0xFF in every byte lane where `g` holds 0 / 1: one addition and one v_perm_b32 (selector bytes 12 and 13 are
// the constants 0x00 and 0xFF).  Written as (g << 8) - g the compiler emits a v_mul_lo_u32 by 255, which issues at a
// quarter of the rate.
__device__ __forceinline__ uint32_t byte_mask(uint32_t g)
{
    return __builtin_amdgcn_perm(0u, 0u, 0x0C0C0C0Cu + g);
}

// Bounds-checked byte-buffer descriptor over one frame buffer: reads past
// `len` return 0 instead of faulting (corrupt offsets cannot leave the frame).
// The range check of a raw buffer works on whole dwords, so the record count is
// `len` rounded up to 4: the last 1-3 bytes of an odd-sized frame stay readable.
// Up to 3 bytes behind the frame can be fetched that way (never used): inside a
// device allocation, which is page-granular, they always exist -- documented in
// include/mcraw_hip.h.  The base may have any byte alignment.
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

// The same load marked non-temporal (read once, streaming).
__device__ __forceinline__ uint4 ld_b128_nt(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 2));
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
__host__ __device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
}

// The same in chunks: every run of 8 * c consecutive logical items is dealt to the 8 XCDs in pieces of c (c = n / 8 is
// xcd_remap, c = 1 the identity); the tail that does not fill a run keeps its order.  Bijective for any grid size.
__host__ __device__ __forceinline__ uint32_t xcd_chunked(uint32_t b, uint32_t n, uint32_t c)
{
    const uint32_t group = 8u * c, g = b / group, r = b - g * group;
    if ((g + 1u) * group > n)
        return b;
    return g * group + (r & 7u) * c + (r >> 3);
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

// Lane i's value of lane i - 1 (lane 0: `first`) / of lane i + 1 (lane 63: `last`), all 64 lanes active: one DPP move across
// the whole wave (gfx9 wave_shr:1 / wave_shl:1) instead of a ds_bpermute round trip through the LDS (~120 cycles when the
// next instruction needs the result).
__device__ __forceinline__ uint32_t wave_prev(uint32_t v, uint32_t first)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(first), static_cast<int>(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_next(uint32_t v, uint32_t last)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last), static_cast<int>(v), 0x130, 0xf, 0xf, false));
}
// Lane k's value (k wave-uniform) for every lane: v_readlane instead of ds_bpermute.
__device__ __forceinline__ uint32_t wave_lane(uint32_t v, uint32_t k)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(k)));
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
        v += __shfl_xor(v, d, 64);
    return v;
}


// Words that one workgroup leaves for another are 64 bits: the launch's epoch in the high half (the buffers are never
// cleared: words of earlier launches carry older epochs and read as "not there yet"), the payload in the low half.  Every
// word is complete in itself, so publishing one is a single relaxed store at device scope and needs no fence.
__device__ __forceinline__ void look_put(uint64_t *w, uint32_t epoch, uint32_t v)
{
    __hip_atomic_store(w, (static_cast<uint64_t>(epoch) << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool look_get(const uint64_t *w, uint32_t epoch, uint32_t *v)
{
    const uint64_t x = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *v = static_cast<uint32_t>(x);
    return static_cast<uint32_t>(x >> 32) == epoch;
}
// ---- fused post-decode stage (Post in mcraw_plan.h) ---------------------------------------------
//
// 8 consecutive samples of row y starting at an even column arrive as four dwords of (even column |
// odd column << 16).  post_black: saturating subtraction of the row's two black levels.  post_pack12:
// the same 8 samples as 12 bytes of an MSB-first 12-bit stream (sample pair a, b -> a>>4,
// (a&15)<<4 | b>>8, b&255), samples above 4095 saturate.
typedef uint16_t mcraw_u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void post_black(uint32_t p[4], const Post &post, uint32_t y)
{
    const mcraw_u16x2 bl = __builtin_bit_cast(mcraw_u16x2, (y & 1u) ? post.black23 : post.black01);
#pragma unroll
    for (int i = 0; i < 4; i++)
        p[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(mcraw_u16x2, p[i]), bl));
}

__device__ __forceinline__ void post_pack12(const uint32_t p[4], uint32_t o[3])
{
    uint32_t t[4]; // 24-bit big-endian groups a << 12 | b
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t c = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mcraw_u16x2, p[i]),
                                                                                  __builtin_bit_cast(mcraw_u16x2, 0x0FFF0FFFu)));
        // a << 12 | b in ONE instruction: the dot product of (a, b) with (4096, 1) (v_dot2_u32_u16; round 3 shifted the high
        // half down and used v_mad_u32_u16: two)
        t[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(mcraw_u16x2, c), __builtin_bit_cast(mcraw_u16x2, 0x00011000u), 0u, false);
    }
    // memory order: t0.b2 t0.b1 t0.b0 | t1.b2 t1.b1 t1.b0 | ...
    o[0] = __builtin_amdgcn_perm(t[1], t[0], 0x06000102u);
    o[1] = __builtin_amdgcn_perm(t[2], t[1], 0x05060001u);
    o[2] = __builtin_amdgcn_perm(t[3], t[2], 0x04050600u);
}

// The same without the clamp: for samples that are known to be below 4096 (k7_tiles proves it per decode item from the blocks'
// references and storage widths, with the black levels already taken off the references).
__device__ __forceinline__ void post_pack12_lean(const uint32_t p[4], uint32_t o[3])
{
    uint32_t t[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        t[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(mcraw_u16x2, p[i]), __builtin_bit_cast(mcraw_u16x2, 0x00011000u), 0u, false);
    o[0] = __builtin_amdgcn_perm(t[1], t[0], 0x06000102u);
    o[1] = __builtin_amdgcn_perm(t[2], t[1], 0x05060001u);
    o[2] = __builtin_amdgcn_perm(t[3], t[2], 0x04050600u);
}

// The same for B = 10 or 14 bits per sample: 8 samples -> B bytes of an MSB-first B-bit stream (TIFF/DNG
// BitsPerSample B, FillOrder 1), samples above 2^B - 1 saturate unless the caller knows that none is (`CLAMP`).
// Round 5: a sample PAIR (a, b) becomes a << B | b with one v_dot2_u32_u16, like the 12-bit form's; the four 2B-bit groups are
// cut into big-endian words with v_lshl_or_b32 / v_or3_b32 and byte-swapped into memory order: 14 / 15 vector instructions where
// the per-sample shifts of round 2 took 27 / 31.
template <int B, bool CLAMP>
__device__ __forceinline__ void post_pack_be(const uint32_t p[4], uint32_t o[4])
{
    static_assert(B == 10 || B == 14, "strip rows of 10 or 14 bits per sample");
    constexpr uint32_t MAXV = (1u << B) - 1u;
    uint32_t g[4]; // a << B | b: 20 / 28 bits
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t c = p[i];
        if (CLAMP)
            c = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mcraw_u16x2, c), __builtin_bit_cast(mcraw_u16x2, MAXV | (MAXV << 16))));
        g[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(mcraw_u16x2, c), __builtin_bit_cast(mcraw_u16x2, 0x00010000u | (1u << B)), 0u, false);
    }
    uint32_t w[4];
    if (B == 10) { // 80 bits: g0:20 g1:20 g2:20 g3:20
        w[0] = (g[0] << 12) | (g[1] >> 8);
        w[1] = (g[1] << 24) | (g[2] << 4) | (g[3] >> 16);
        w[2] = g[3] << 16;
        w[3] = 0u;
    } else { // 112 bits: g0:28 g1:28 g2:28 g3:28
        w[0] = (g[0] << 4) | (g[1] >> 24);
        w[1] = (g[1] << 8) | (g[2] >> 20);
        w[2] = (g[2] << 12) | (g[3] >> 16);
        w[3] = g[3] << 16;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        o[i] = __builtin_amdgcn_perm(w[i], w[i], 0x00010203u);
}

// The output buffers are global memory, but a pointer that a kernel loads from a table is generic to the compiler,
// which then emits FLAT stores: those also count on the LDS counter, so every wait for an LDS read waits for the
// stores in front of it as well.  gptr<V>(p) is p as a pointer to V in the global address space.
#define MCRAW_GLOBAL __attribute__((address_space(1)))
template <class V, class T>
__device__ __forceinline__ MCRAW_GLOBAL V *gptr(T *p)
{
    return (MCRAW_GLOBAL V *)p;
}

// The decoded rows are written once and never read again by these kernels: 16 (12) bytes per lane as a write-through,
// streaming store (`sc1 nt`: the line does not stay in the XCD's L2, which then holds more of what is being READ; with `nt`
// alone the line stays -- MI355X_MICROARCH.md, "stores of each flavour").  k7_tiles: 0.971 -> 0.958 ms on one box, six
// interleaved pairs of fresh processes (tools/store_sc.sh); `sc1` without `nt` is no faster than `nt`.
#ifndef MCRAW_STORE_POLICY
#define MCRAW_STORE_POLICY "sc1 nt" // (tools/store_sc.sh builds the library with others)
#endif
typedef uint32_t mcraw_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t mcraw_u32x3 __attribute__((ext_vector_type(3)));
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "store_stream16/12: the `sc1 nt` mnemonics and the two wait states behind a store of more than 8 bytes are gfx940+ facts"
#endif
__device__ __forceinline__ void store_stream16(void *dst, mcraw_u32x4 v)
{
    // (the string ends with s_nop 1: nothing inside an asm statement is padded, and the compiler's next instruction may
    // otherwise overwrite the data registers of a store of more than 8 bytes before the store has read them)
    asm volatile("global_store_dwordx4 %0, %1, off " MCRAW_STORE_POLICY "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_stream12(void *dst, mcraw_u32x3 v)
{
    asm volatile("global_store_dwordx3 %0, %1, off " MCRAW_STORE_POLICY "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
}

// Bytes [0, nb) of the dwords o[] to dst, the last one masked by `last` (the cropped end of a strip row).
__device__ __forceinline__ void post_store_bytes(uint8_t *dst, const uint32_t *o, uint32_t nb, uint32_t last, uint32_t maxb)
{
    for (uint32_t i = 0; i < maxb; i++)
        if (i < nb) {
            uint32_t b = (o[i >> 2] >> (8u * (i & 3u))) & 0xffu;
            if (i == nb - 1u)
                b &= last;
            gptr<uint8_t>(dst)[i] = static_cast<uint8_t>(b);
        }
}

// Store 8 samples (columns x..x+7, x % 8 == 0, of which the first `n` exist) of row y in the
// post-stage layout.  `quick`: rows are dword multiples and the buffer is dword aligned (12-bit
// form) / 16-byte friendly (16-bit form).
// PB: bits per output sample, a compile-time property of the kernel instance (16, 12, 10 or 14) so that every
// instance carries one packing only.
// `lean` (strip rows, wave-uniform): the samples need neither the black levels (they are off the references already)
// nor the clamp.
template <bool NT, int PB>
__device__ __forceinline__ void post_store8(uint16_t *out, const Post &post, uint32_t width, uint32_t y, uint32_t x,
                                            uint32_t p[4], uint32_t n, bool quick, bool lean = false)
{
    if (!(PB != 16 && lean) && (post.mode & POST_BLACK))
        post_black(p, post, y);
    if (PB == 10 || PB == 14) { // 10 / 14 bytes per 8 samples; rows start on even bytes when width % 8 == 0
        constexpr uint32_t B = PB == 10 ? 10u : 14u;
        uint32_t o[4];
        if (lean)
            post_pack_be<(PB == 10 ? 10 : 14), false>(p, o);
        else
            post_pack_be<(PB == 10 ? 10 : 14), true>(p, o);
        uint8_t *dst = reinterpret_cast<uint8_t *>(out) + static_cast<size_t>(y) * post_row_bytes(width, post.mode) + (x >> 3) * B;
        if (quick && n == 8u) {
            typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(2)));
            typedef uint32_t u32x3_u __attribute__((ext_vector_type(3), aligned(2)));
            if (PB == 10) {
                const u32x2_u v = {o[0], o[1]};
                *gptr<u32x2_u>(dst) = v;
                *gptr<uint16_t>(dst + 8) = static_cast<uint16_t>(o[2]);
            } else {
                const u32x3_u v = {o[0], o[1], o[2]};
                *gptr<u32x3_u>(dst) = v;
                *gptr<uint16_t>(dst + 12) = static_cast<uint16_t>(o[3]);
            }
        } else { // rows off the 2-byte grid, and the cropped end of a row (a last sample may end inside a byte)
            const uint32_t bitsn = n * B, nb = (bitsn + 7u) >> 3;
            post_store_bytes(dst, o, nb, (bitsn & 7u) ? (0xffu << (8u - (bitsn & 7u))) & 0xffu : 0xffu, 14u);
        }
        return;
    }
    if (PB == 12) {
        uint32_t o[3];
        if (lean)
            post_pack12_lean(p, o);
        else
            post_pack12(p, o);
        uint8_t *dst = reinterpret_cast<uint8_t *>(out) + static_cast<size_t>(y) * post_row_bytes(width, post.mode) + (x >> 3) * 12u;
        if (quick && n == 8u) {
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
            const u32x3 v = {o[0], o[1], o[2]};
            if (NT)
                store_stream12(dst, v);
            else
                *gptr<u32x3>(dst) = v;
        } else if (n == 8u) { // a strip row off the dword grid: one unaligned 12-byte store
            typedef uint32_t u32x3_u __attribute__((ext_vector_type(3), aligned(1)));
            const u32x3_u v = {o[0], o[1], o[2]};
            *gptr<u32x3_u>(dst) = v;
        } else { // the cropped end of a row: bytes; an odd last sample owns the high nibble of its second byte
            const uint32_t nb = (n * 12u + 7u) >> 3;
#pragma unroll
            for (uint32_t i = 0; i < 12u; i++)
                if (i < nb) {
                    uint32_t b = (o[i >> 2] >> (8u * (i & 3u))) & 0xffu;
                    if ((n & 1u) && i == nb - 1u)
                        b &= 0xf0u;
                    gptr<uint8_t>(dst)[i] = static_cast<uint8_t>(b);
                }
        }
        return;
    }
    uint16_t *dst = out + static_cast<size_t>(y) * width + x;
    if (quick && n == 8u) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        if (NT)
            store_stream16(dst, v);
        else
            *gptr<u32x4>(dst) = v;
    } else if (n == 8u) { // rows off the 16-byte grid: one unaligned 16-byte store
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
        const u32x4_u v = {p[0], p[1], p[2], p[3]};
        *gptr<u32x4_u>(dst) = v;
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++)
            if (i < n)
                gptr<uint16_t>(dst)[i] = static_cast<uint16_t>(p[i >> 1] >> (16u * (i & 1u)));
    }
}

// 10- and 14-bit strip rows, the interior of a frame (round 5).  A lane's 8 samples are 10 / 14 bytes; stored as 8 + 2 / 12 + 2
// bytes, a wave's two store instructions touch every 64-byte line of the row piece twice (counters: 2.0 x the write requests
// between L1 and L2 that the bytes need; the 14-bit kernel 1.20 ms where the plain one takes 0.98 for more bytes).  Instead every
// lane stores ONE 16-byte piece: its own bytes and the first 6 / 2 bytes of the row's NEXT 8 samples, fetched from the lane that
// holds them (`src`; ds_bpermute: the LDS crossbar, no memory) -- two lanes write those bytes, both write the same values, so no
// order matters.  A lane whose next 8 samples are not decoded in this pass (`has_next` false: the last piece of the pass's last
// tile, or of the frame's last tile column) stores the last 6 / 2 bytes of the PREVIOUS 8 samples in front of its own instead
// (four lanes down in its group of eight: one DPP row shift).  One store instruction per row for the whole wave: a wave's
// instruction that only a few lanes take costs the memory pipeline as much as a full one (the same exchange with the odd lanes
// on the old 12 + 2 form: 1.38 ms).
// All 64 lanes of the wave must be here (the caller checks), the row piece whole and inside the frame.
template <int PB>
__device__ __forceinline__ void post_store8_merged(uint16_t *out, const Post &post, uint32_t width, uint32_t y, uint32_t x, uint32_t p[4],
                                                   bool lean, uint32_t src, bool has_next)
{
    static_assert(PB == 10 || PB == 14, "strip rows whose 8-sample pieces are no dword multiple");
    constexpr uint32_t B = PB;
    if (!lean && (post.mode & POST_BLACK))
        post_black(p, post, y);
    uint32_t o[4];
    if (lean)
        post_pack_be<PB, false>(p, o);
    else
        post_pack_be<PB, true>(p, o);
    const auto next = [&](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src * 4u), static_cast<int>(v))); };
    const auto prev = [&](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, true)); }; // row_shr:4: lane - 4
    uint8_t *dst = reinterpret_cast<uint8_t *>(out) + static_cast<size_t>(y) * post_row_bytes(width, post.mode) + (x >> 3) * B;
    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
    u32x4_u v;
    if (PB == 10) {
        const uint32_t n0 = next(o[0]), n1 = next(o[1]), m1 = prev(o[1]), m2 = prev(o[2]);
        const u32x4_u fwd = {o[0], o[1], (o[2] & 0xffffu) | (n0 << 16), (n0 >> 16) | (n1 << 16)};                               // at dst
        const u32x4_u back = {m1, (m2 & 0xffffu) | (o[0] << 16), (o[0] >> 16) | (o[1] << 16), (o[1] >> 16) | (o[2] << 16)};     // at dst - 6
        v = has_next ? fwd : back;
        dst -= has_next ? 0 : 6;
    } else {
        const uint32_t n0 = next(o[0]), m3 = prev(o[3]);
        const u32x4_u fwd = {o[0], o[1], o[2], (o[3] & 0xffffu) | (n0 << 16)};                                                   // at dst
        const u32x4_u back = {(m3 & 0xffffu) | (o[0] << 16), (o[0] >> 16) | (o[1] << 16), (o[1] >> 16) | (o[2] << 16), (o[2] >> 16) | (o[3] << 16)}; // at dst - 2
        v = has_next ? fwd : back;
        dst -= has_next ? 0 : 2;
    }
    *gptr<u32x4_u>(dst) = v;
}

} // namespace mcraw
