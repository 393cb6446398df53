// mcraw_type7.hip -- gfx950 kernels for the current MCRAW frame encoding
// (compressionType 7).  Replaces motioncam::raw::Decode, lib/RawData.cpp:528-612.
//
//   k7_walk   side-stream chain resolve   (RawData.cpp:463-498, the inline-header chain)
//   k7_meta   side-stream record decode   (RawData.cpp:485-495) -> bits[], refs[], group lengths
//   k7_scan   payload offsets             (RawData.cpp:562, 576-579: offset += LEN[bits])
//   k7_tiles  tile unpack + reference add + Bayer interleave + crop
//             (RawData.cpp:410-461, 112-408, 581-593, 598-608)   <- the roofline kernel
//
// Integer bit-slicing on byte planes; no MFMA.  A "group" is the 64 payload
// blocks (16 tiles of 64x4 px) described by one record of the bits stream; its
// payload is one contiguous, 8-byte aligned span of <= 8 KiB.
#include "mcraw_dev.h"

#include "../../include/mcraw_hip.h"

namespace mcraw {

// ------------------------------------------------------------------ term table
//
// Every sample of a block stored at <= 10 bits is the OR of at most three
// byte-domain terms ((P[p][j] >> s) & (2^n - 1)) << l, plus (Decode10 only) a
// 2-bit term that lands in bits 8..9.  Lane j of the reference's 8-wide SIMD
// is byte j of an 8-byte plane; sample index = 8*k + j.  One row per
// (class, k): {term0, term1, term2, term_hi}, a term = p*8 | s<<8 | n<<16 | l<<24.
// Rows follow lib/RawData.cpp: Decode1 :112-136, Decode2 :138-162, Decode3 :164-199,
// Decode4 :201-223, Decode5 :225-262, Decode6 :264-304, Decode8 :306-326, Decode10 :328-374.
#define TM(p, s, n, l) ((uint32_t)((p) * 8) | ((uint32_t)(s) << 8) | ((uint32_t)(n) << 16) | ((uint32_t)(l) << 24))
#define Z 0u
#define ROW1(k) {TM(0, k, 1, 0), Z, Z, Z}
#define ROW2(k) {TM((k) >> 2, 2 * ((k)&3), 2, 0), Z, Z, Z}
#define ROW4(k) {TM((k) >> 1, 4 * ((k)&1), 4, 0), Z, Z, Z}
#define ROW8(k) {TM(k, 0, 8, 0), Z, Z, Z}
#define ROW10(k) {TM(5 * ((k) >> 2) + ((k)&3), 0, 8, 0), Z, Z, TM(5 * ((k) >> 2) + 4, 2 * ((k)&3), 2, 0)}
__constant__ uint32_t c_tab7[9 * 8][4] = {
    // class 0: all zero
    {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z}, {Z, Z, Z, Z},
    ROW1(0), ROW1(1), ROW1(2), ROW1(3), ROW1(4), ROW1(5), ROW1(6), ROW1(7),
    ROW2(0), ROW2(1), ROW2(2), ROW2(3), ROW2(4), ROW2(5), ROW2(6), ROW2(7),
    // class 3
    {TM(0, 0, 3, 0), Z, Z, Z}, {TM(0, 3, 3, 0), Z, Z, Z}, {TM(0, 6, 2, 0), TM(2, 6, 1, 2), Z, Z},
    {TM(1, 0, 3, 0), Z, Z, Z}, {TM(1, 3, 3, 0), Z, Z, Z}, {TM(1, 6, 2, 0), TM(2, 7, 1, 2), Z, Z},
    {TM(2, 0, 3, 0), Z, Z, Z}, {TM(2, 3, 3, 0), Z, Z, Z},
    ROW4(0), ROW4(1), ROW4(2), ROW4(3), ROW4(4), ROW4(5), ROW4(6), ROW4(7),
    // class 5
    {TM(0, 0, 5, 0), Z, Z, Z}, {TM(1, 0, 5, 0), Z, Z, Z}, {TM(2, 0, 5, 0), Z, Z, Z}, {TM(3, 0, 5, 0), Z, Z, Z},
    {TM(4, 0, 5, 0), Z, Z, Z}, {TM(0, 5, 3, 0), TM(3, 5, 2, 3), Z, Z}, {TM(1, 5, 3, 0), TM(4, 5, 2, 3), Z, Z},
    {TM(2, 5, 3, 0), TM(3, 7, 1, 3), TM(4, 7, 1, 4), Z},
    // class 6
    {TM(0, 0, 6, 0), Z, Z, Z}, {TM(1, 0, 6, 0), Z, Z, Z}, {TM(2, 0, 6, 0), Z, Z, Z}, {TM(3, 0, 6, 0), Z, Z, Z},
    {TM(4, 0, 6, 0), Z, Z, Z}, {TM(5, 0, 6, 0), Z, Z, Z},
    {TM(0, 6, 2, 0), TM(1, 6, 2, 2), TM(2, 6, 2, 4), Z}, {TM(3, 6, 2, 0), TM(4, 6, 2, 2), TM(5, 6, 2, 4), Z},
    ROW8(0), ROW8(1), ROW8(2), ROW8(3), ROW8(4), ROW8(5), ROW8(6), ROW8(7),
    ROW10(0), ROW10(1), ROW10(2), ROW10(3), ROW10(4), ROW10(5), ROW10(6), ROW10(7),
};
#undef TM
#undef Z

__device__ __forceinline__ uint32_t term_off(uint32_t t) { return t & 0xffu; }
__device__ __forceinline__ uint32_t term_shr(uint32_t t) { return (t >> 8) & 31u; }
__device__ __forceinline__ uint32_t term_bits(uint32_t t) { return (t >> 16) & 31u; }
__device__ __forceinline__ uint32_t term_shl(uint32_t t) { return (t >> 24) & 31u; }

// ------------------------------------------------------------------ k7_walk
//
// One wave per (frame, side stream).  The stream is a chain of records
// {hbits<<4 | ref>>8, ref & 255, LEN[hbits] payload bytes}: where record i+1
// starts is only known from the header of record i (RawData.cpp:485-495).  The
// wave pulls the stream through LDS in 4 KiB pieces (next piece in flight while
// the current one is walked) and chases the headers with scalar code.
constexpr int PIECE = 4096;

__global__ __launch_bounds__(64) void k7_walk(const Work7 W)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_piece[2][PIECE];

    const int f = blockIdx.x >> 1;
    const int s = blockIdx.x & 1;
    const Plan7 *P = W.plans + f;
    int32_t *status = W.status + f;
    const uint32_t lane = threadIdx.x;
    const uint32_t len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);

    // frame header: 4 x u32 LE (RawData.cpp:500-524) and its checks (:547-554)
    const uint4 h = ld_b128(rs, 0);
    const uint32_t encW = h.x, encH = h.y;
    const uint32_t so = s ? h.w : h.z; // refsOffset : bitsOffset
    int32_t err = 0;
    if (len < 16u || h.z > len || h.w > len || (encW & 63u) != 0u || encW < static_cast<uint32_t>(P->width) ||
        encW == 0u || encH == 0u || (encH & 3u) != 0u)
        err = MCRAW_E_HEADER;
    else if (encW != P->encW || encH != P->encH)
        err = E_GEOMETRY;
    else if (so + 4u > len || so + 4u < so)
        err = MCRAW_E_TRUNCATED;
    uint32_t count = 0;
    if (!err) {
        count = ld_u8(rs, so) | (ld_u8(rs, so + 1) << 8) | (ld_u8(rs, so + 2) << 16) | (ld_u8(rs, so + 3) << 24);
        if (count < P->nblk) // the reference would index past the vector (RawData.cpp:573-574)
            err = MCRAW_E_SIDESTREAM;
    }
    if (err) {
        if (lane == 0)
            atomicOr(status, err);
        return;
    }

    const uint32_t R = P->ngroups;
    uint32_t *__restrict__ rec_off = W.rec_off + (static_cast<size_t>(f) * 2u + s) * W.Rmax;
    uint32_t pos = __builtin_amdgcn_readfirstlane(so + 4u);
    uint32_t pb = pos & ~15u;

    uint4 r0 = ld_b128(rs, pb + lane * 16u);
    uint4 r1 = ld_b128(rs, pb + 1024u + lane * 16u);
    uint4 r2 = ld_b128(rs, pb + 2048u + lane * 16u);
    uint4 r3 = ld_b128(rs, pb + 3072u + lane * 16u);
    int buf = 0;
    uint32_t i = 0, mine = 0;
    bool bad = false;
    while (i < R && !bad) {
        uint4 *dst = reinterpret_cast<uint4 *>(s_piece[buf]);
        dst[lane] = r0;
        dst[64 + lane] = r1;
        dst[128 + lane] = r2;
        dst[192 + lane] = r3;
        __syncthreads();
        // next piece goes in flight now, lands while this one is walked
        const uint32_t nb = pb + PIECE;
        r0 = ld_b128(rs, nb + lane * 16u);
        r1 = ld_b128(rs, nb + 1024u + lane * 16u);
        r2 = ld_b128(rs, nb + 2048u + lane * 16u);
        r3 = ld_b128(rs, nb + 3072u + lane * 16u);
        const uint8_t *piece = s_piece[buf];
        while (i < R && pos < nb) {
            const uint32_t hb = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(piece[pos - pb])) >> 4;
            const uint32_t next = pos + 2u + len7_of(hb);
            if (next > len) { // RawData.cpp:419-420 would skip the block and leave stale data
                bad = true;
                break;
            }
            if (lane == (i & 63u))
                mine = pos;
            if ((i & 63u) == 63u)
                rec_off[i - 63u + lane] = mine;
            pos = next;
            ++i;
        }
        pb = nb;
        buf ^= 1;
    }
    if (bad) {
        if (lane == 0)
            atomicOr(status, MCRAW_E_TRUNCATED);
        return;
    }
    const uint32_t tail = i & 63u;
    if (lane < tail)
        rec_off[i - tail + lane] = mine;
}

// ------------------------------------------------------------------ k7_meta
//
// One wave per side-stream record, lane = entry.  Same unpack as a payload block
// (DecodeBlock on the record, RawData.cpp:489) plus the record's reference
// (:491-492).  Bits records also emit the byte length of their 64-block group.
__global__ __launch_bounds__(256) void k7_meta(const Work7 W)
{
    const uint32_t f = blockIdx.y;
    const Plan7 *P = W.plans + f;
    int32_t *status = W.status + f;
    if (*status != 0)
        return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t R = P->ngroups;
    const uint32_t rec = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (rec >= 2u * R)
        return;
    const uint32_t s = rec >= R ? 1u : 0u;
    const uint32_t r = rec - s * R;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, P->len);
    const uint32_t off = __builtin_amdgcn_readfirstlane(W.rec_off[(static_cast<size_t>(f) * 2u + s) * W.Rmax + r]);
    const uint32_t b0 = ld_u8(rs, off), b1 = ld_u8(rs, off + 1u);
    const uint32_t hb = b0 >> 4;                   // RawData.cpp:106-110
    const uint32_t ref = ((b0 & 15u) << 8) | b1;
    const uint32_t pay = off + 2u;
    const uint32_t k = lane >> 3, j = lane & 7u;

    uint32_t v = 0;
    if (hb >= 11u) { // raw 16, little endian (RawData.cpp:376-408)
        v = ld_u8(rs, pay + 2u * lane) | (ld_u8(rs, pay + 2u * lane + 1u) << 8);
    } else if (hb != 0u) {
        const uint32_t *row = c_tab7[cls7_of(hb) * 8u + k];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const uint32_t tm = row[t];
            if (term_bits(tm) != 0u)
                v |= ((ld_u8(rs, pay + term_off(tm) + j) >> term_shr(tm)) & ((1u << term_bits(tm)) - 1u)) << term_shl(tm);
        }
        const uint32_t th = row[3];
        if (term_bits(th) != 0u)
            v |= ((ld_u8(rs, pay + term_off(th) + j) >> term_shr(th)) & 3u) << 8;
    }
    v = (v + ref) & 0xffffu; // uint16 wrap (RawData.cpp:492)

    const uint32_t idx = r * 64u + lane;
    if (s == 0u) {
        const bool used = idx < P->nblk;
        if (used && v > 16u) { // would index past ENCODING_BLOCK_LENGTH (RawData.cpp:419)
            atomicOr(status, MCRAW_E_SIDESTREAM);
            v = 16u;
        }
        W.bits[static_cast<size_t>(f) * W.Rmax * 64u + idx] = static_cast<uint8_t>(v);
        const uint32_t sum = wave_sum(used ? len7_of(v) : 0u);
        if (lane == 0)
            W.grp_off[static_cast<size_t>(f) * (W.Rmax + 1u) + r] = sum; // lengths until k7_scan
    } else {
        W.refs[static_cast<size_t>(f) * W.Rmax * 64u + idx] = static_cast<uint16_t>(v);
    }
}

// ------------------------------------------------------------------ k7_scan
//
// Payload offset of every group: 16 + sum of the lengths before it
// (RawData.cpp:562 `offset = METADATA_OFFSET`, :576-579 `offset += ...`).
__global__ __launch_bounds__(256) void k7_scan(const Work7 W)
{
    __shared__ uint32_t s_w[4];
    const Plan7 *P = W.plans + blockIdx.x;
    int32_t *status = W.status + blockIdx.x;
    if (*status != 0)
        return;
    const uint32_t R = P->ngroups, tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t *__restrict__ g = W.grp_off + static_cast<size_t>(blockIdx.x) * (W.Rmax + 1u);
    uint32_t carry = 16u;
    for (uint32_t base = 0; base < R; base += 256u) {
        const uint32_t i = base + tid;
        const uint32_t v = i < R ? g[i] : 0u;
        uint32_t wtot;
        const uint32_t ex = wave_excl_scan(v, lane, &wtot);
        if (lane == 63u)
            s_w[w] = wtot;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) {
            const uint32_t x = s_w[q];
            before += q < w ? x : 0u;
            total += x;
        }
        if (i < R)
            g[i] = carry + before + ex;
        carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        g[R] = carry;
        if (carry > P->len) // some block crosses `len` (RawData.cpp:419-420)
            atomicOr(status, MCRAW_E_TRUNCATED);
    }
}

// ------------------------------------------------------------------ k7_tiles
//
// One 256-thread workgroup per group of 16 tiles.  The group's payload span is
// pulled into LDS with 16-byte coalesced buffer loads; then every lane owns the
// 8 samples (8*k .. 8*k+7) of two sibling blocks (2r, 2r+1) of one tile -- the
// reference's UInt16x8 vector (RawData.cpp:47-104) as four packed-u16 dwords --
// and emits 16 consecutive pixels of one output row.
//
//   lane -> tile tt = tid>>4, row pair r = (tid>>3)&1, k = tid&7
//   pixel row = 4*ty + r + 2*(k>>2), first column = 64*tx + 16*(k&3)   (RawData.cpp:581-593)
struct Unpacked { uint32_t x[4]; }; // samples (0,1)(2,3)(4,5)(6,7) as packed u16 pairs

__device__ __forceinline__ Unpacked unpack8(const uint8_t *__restrict__ blk, uint32_t cidx, uint32_t k,
                                            const uint4 *__restrict__ s_tab)
{
    Unpacked u;
    if (cidx >= 9u) { // raw 16: samples 8k..8k+7 are 16 bytes, already little-endian u16
        const uint2 a = *reinterpret_cast<const uint2 *>(blk + 16u * k);
        const uint2 b = *reinterpret_cast<const uint2 *>(blk + 16u * k + 8u);
        u.x[0] = a.x; u.x[1] = a.y; u.x[2] = b.x; u.x[3] = b.y;
        return u;
    }
    const uint4 row = s_tab[cidx * 8u + k];
    uint32_t lo = 0, hi = 0; // byte-domain accumulators: samples j=0..3 and j=4..7
    const uint32_t tms[3] = {row.x, row.y, row.z};
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const uint32_t tm = tms[t];
        const uint2 p = *reinterpret_cast<const uint2 *>(blk + term_off(tm));
        const uint32_t m = ((1u << term_bits(tm)) - 1u) * 0x01010101u;
        lo |= ((p.x >> term_shr(tm)) & m) << term_shl(tm);
        hi |= ((p.y >> term_shr(tm)) & m) << term_shl(tm);
    }
    const uint32_t th = row.w; // Decode10's bits 8..9
    const uint2 ph = *reinterpret_cast<const uint2 *>(blk + term_off(th));
    const uint32_t mh = ((1u << term_bits(th)) - 1u) * 0x01010101u;
    const uint32_t lo8 = (ph.x >> term_shr(th)) & mh, hi8 = (ph.y >> term_shr(th)) & mh;
    // bytes -> packed u16 pairs: {lo.b0 | lo8.b0 << 8, lo.b1 | lo8.b1 << 8} ...
    u.x[0] = __builtin_amdgcn_perm(lo8, lo, 0x05010400u);
    u.x[1] = __builtin_amdgcn_perm(lo8, lo, 0x07030602u);
    u.x[2] = __builtin_amdgcn_perm(hi8, hi, 0x05010400u);
    u.x[3] = __builtin_amdgcn_perm(hi8, hi, 0x07030602u);
    return u;
}

constexpr int PAY_LDS = SPAN_MAX + 16 + 32; // span + 16-B alignment head + slack for zero-length tails

__global__ __launch_bounds__(256) void k7_tiles(const Work7 W)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_pay[PAY_LDS];
    __shared__ uint4 s_tab[72];
    __shared__ uint32_t s_blk[GROUP_BLOCKS]; // byte offset in span | class << 16
    __shared__ uint16_t s_ref[GROUP_BLOCKS];

    // grid = (Rmax groups, n7 frames); consecutive items share an XCD (adjacent spans / rows)
    const uint32_t item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const uint32_t f = item / gridDim.x;
    const uint32_t g = item - f * gridDim.x;
    const Plan7 *P = W.plans + f;
    if (g >= P->ngroups || W.status[f] != 0)
        return;
    const uint32_t tid = threadIdx.x;
    const uint32_t nblk = P->nblk;

    // span of this group in the frame buffer
    const uint32_t *grp = W.grp_off + static_cast<size_t>(f) * (W.Rmax + 1u) + g;
    const uint32_t start = __builtin_amdgcn_readfirstlane(grp[0]);
    const uint32_t end = __builtin_amdgcn_readfirstlane(grp[1]);
    const uint32_t base16 = start & ~15u;
    const uint32_t head = start - base16;
    const uint32_t n16 = min((end - base16 + 15u) >> 4, 513u); // never more than SPAN_MAX + head

    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, P->len);
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0;
    if (tid < n16)
        v0 = ld_b128(rs, base16 + tid * 16u);
    if (tid + 256u < n16)
        v1 = ld_b128(rs, base16 + (tid + 256u) * 16u);
    if (tid + 512u < n16)
        v2 = ld_b128(rs, base16 + (tid + 512u) * 16u);

    if (tid < 72u)
        s_tab[tid] = reinterpret_cast<const uint4 *>(c_tab7)[tid];

    if (tid < 64u) { // wave 0: per-block class, reference and offset inside the span
        const uint32_t blk = g * 64u + tid;
        const bool used = blk < nblk;
        const size_t mo = static_cast<size_t>(f) * W.Rmax * 64u + blk;
        const uint32_t b = used ? W.bits[mo] : 0u;
        const uint32_t r = used ? W.refs[mo] : 0u;
        uint32_t total;
        const uint32_t ex = wave_excl_scan(len7_of(b), tid, &total);
        s_blk[tid] = (head + ex) | (cls7_of(b) << 16);
        s_ref[tid] = static_cast<uint16_t>(r);
    }

    uint4 *pay4 = reinterpret_cast<uint4 *>(s_pay);
    if (tid < n16)
        pay4[tid] = v0;
    if (tid + 256u < n16)
        pay4[tid + 256u] = v1;
    if (tid + 512u < n16)
        pay4[tid + 512u] = v2;
    __syncthreads();

    const uint32_t tt = tid >> 4, r = (tid >> 3) & 1u, k = tid & 7u;
    const uint32_t tile = g * GROUP_TILES + tt;
    if (tile * 4u >= nblk)
        return;
    const uint32_t tilesX = P->tilesX;
    const uint32_t ty = tile / tilesX, tx = tile - ty * tilesX;
    const uint32_t y = 4u * ty + r + 2u * (k >> 2);
    const uint32_t x = 64u * tx + 16u * (k & 3u);
    const int32_t width = P->width;
    if (y >= static_cast<uint32_t>(P->rows) || x >= static_cast<uint32_t>(width))
        return;

    const uint32_t bi = 4u * tt + 2u * r;
    const uint2 mb = *reinterpret_cast<const uint2 *>(&s_blk[bi]);
    const uint32_t refs2 = *reinterpret_cast<const uint32_t *>(&s_ref[bi]); // refA | refB << 16
    const Unpacked A = unpack8(s_pay + (mb.x & 0xffffu), mb.x >> 16, k, s_tab);
    const Unpacked B = unpack8(s_pay + (mb.y & 0xffffu), mb.y >> 16, k, s_tab);

    // Bayer interleave (RawData.cpp:582-592): pixel 2i from block 2r, 2i+1 from 2r+1;
    // add both references with uint16 wrap-around in one packed add.
    uint32_t o[8];
    const u16x2 rr = __builtin_bit_cast(u16x2, refs2);
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const uint32_t e = __builtin_amdgcn_perm(B.x[m], A.x[m], 0x05040100u);
        const uint32_t d = __builtin_amdgcn_perm(B.x[m], A.x[m], 0x07060302u);
        o[2 * m] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, e) + rr);
        o[2 * m + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, d) + rr);
    }

    uint16_t *dst = P->out + static_cast<size_t>(y) * static_cast<size_t>(width) + x;
    if (P->fast_store && x + 16u <= static_cast<uint32_t>(width)) {
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        d4[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d4[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else { // cropped or unaligned row: element stores (RawData.cpp:598-608 copies `width` only)
        const uint32_t n = min(16u, static_cast<uint32_t>(width) - x);
#pragma unroll
        for (uint32_t i = 0; i < 16u; i++)
            if (i < n)
                dst[i] = static_cast<uint16_t>(o[i >> 1] >> (16u * (i & 1u)));
    }
}

// ------------------------------------------------------------------ launchers

void launch_k7(const Work7 &W, uint32_t stage, hipStream_t st)
{
    const uint32_t n7 = static_cast<uint32_t>(W.n7);
    switch (stage) {
    case MCRAW_K7_WALK:
        hipLaunchKernelGGL(k7_walk, dim3(2 * n7), dim3(64), 0, st, W);
        break;
    case MCRAW_K7_META:
        hipLaunchKernelGGL(k7_meta, dim3((2 * W.Rmax + 3) / 4, n7), dim3(256), 0, st, W);
        break;
    case MCRAW_K7_SCAN:
        hipLaunchKernelGGL(k7_scan, dim3(n7), dim3(256), 0, st, W);
        break;
    case MCRAW_K7_TILES:
        hipLaunchKernelGGL(k7_tiles, dim3(W.Rmax, n7), dim3(256), 0, st, W);
        break;
    }
}

} // namespace mcraw
