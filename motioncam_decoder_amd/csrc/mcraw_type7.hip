// mcraw_type7.hip -- gfx950 kernels for the current MCRAW frame encoding
// (compressionType 7).  Replaces motioncam::raw::Decode, lib/RawData.cpp:528-612.
//
//   k7_side    one workgroup per side stream: header checks (RawData.cpp:500-524, 547-554), the
//              inline-header record chain (:463-498), record decode (:485-495) -> bits[], refs[],
//              and the payload offset of every decode item (:562, 576-579: offset += LEN[bits])
//   k7_tiles   tile unpack + reference add + Bayer interleave + crop
//              (RawData.cpp:410-461, 112-408, 581-593, 598-608)  <- the roofline kernel
//
// Integer bit-slicing on byte planes; no MFMA.  A "group" is the 64 payload blocks (16 tiles of
// 64x4 px) described by one record of the bits stream; a decode "item" is half of it (32 blocks,
// 8 tiles): its payload is one contiguous, 8-byte aligned span of <= 4 KiB.
#include <algorithm>
#include <cstdlib>

#include "mcraw_dev.h"

#include "../../include/mcraw_hip.h"

namespace mcraw {

// ------------------------------------------------------------------ term table
//
// Every sample of a block stored at <= 10 bits is the OR of at most three
// byte-domain terms ((P[p][j] >> s) & (2^n - 1)) << l, plus (Decode10 only) a
// 2-bit term that lands in bits 8..9.  Lane j of the reference's 8-wide SIMD
// is byte j of an 8-byte plane; sample index = 8*k + j.  One row per
// (class, k): {term0, term1, term2, term_hi}, a term = p*8 | s<<8 | (2^n - 1)<<16 | l<<24.
// Rows follow lib/RawData.cpp: Decode1 :112-136, Decode2 :138-162, Decode3 :164-199,
// Decode4 :201-223, Decode5 :225-262, Decode6 :264-304, Decode8 :306-326, Decode10 :328-374.
#define TM(p, s, n, l) ((uint32_t)((p) * 8) | ((uint32_t)(s) << 8) | ((uint32_t)((1u << (n)) - 1u) << 16) | ((uint32_t)(l) << 24))
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
// the field mask 2^n - 1 of the term in every byte lane (one v_perm_b32; a multiply by 0x01010101 is quarter rate)
__device__ __forceinline__ uint32_t term_mask(uint32_t t) { return __builtin_amdgcn_perm(t, t, 0x02020202u); }
__device__ __forceinline__ uint32_t term_shl(uint32_t t) { return (t >> 24) & 31u; }

// ------------------------------------------------------------------ block unpack
//
// The reference's UInt16x8 vector (RawData.cpp:47-104): 8 samples 8k..8k+7 of one
// 64-sample block as four dwords of packed u16 pairs.
struct Unpacked { uint32_t x[4]; }; // samples (0,1)(2,3)(4,5)(6,7)

// 8 bytes of LDS at byte offset `off` of the array `base`.  Payload blocks are 8-byte
// aligned (ALIGNED8); side-stream records sit at arbitrary byte offsets: three aligned dwords
// and a funnel shift then stand in for the unaligned 64-bit read.
template <bool ALIGNED8>
__device__ __forceinline__ uint2 lds_read8(const uint8_t *__restrict__ base, uint32_t off)
{
    if (ALIGNED8)
        return *reinterpret_cast<const uint2 *>(base + off);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base) + (off >> 2);
    const uint32_t sh = (off & 3u) * 8u; // any byte alignment: a side stream may start at an odd offset
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    return make_uint2(__builtin_amdgcn_alignbit(w1, w0, sh), __builtin_amdgcn_alignbit(w2, w1, sh));
}

template <bool ALIGNED8>
__device__ __forceinline__ Unpacked unpack8(const uint8_t *__restrict__ base, uint32_t blk, uint32_t cidx, uint32_t k,
                                            const uint4 *__restrict__ s_tab)
{
    Unpacked u;
    if (cidx >= 9u) { // raw 16: samples 8k..8k+7 are 16 bytes, already little-endian u16
        const uint2 a = lds_read8<ALIGNED8>(base, blk + 16u * k);
        const uint2 b = lds_read8<ALIGNED8>(base, blk + 16u * k + 8u);
        u.x[0] = a.x; u.x[1] = a.y; u.x[2] = b.x; u.x[3] = b.y;
        return u;
    }
    const uint4 row = s_tab[cidx * 8u + k];
    uint32_t lo = 0, hi = 0; // byte-domain accumulators: samples j=0..3 and j=4..7
    const uint32_t tms[3] = {row.x, row.y, row.z};
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const uint32_t tm = tms[t];
        // payload lanes run all terms branch-free (one aligned 8-byte read each); for side-stream
        // records a term costs three reads and two funnel shifts, and most (class, k) rows have
        // one term only: skip the ones no lane of the wave needs
        if (!ALIGNED8 && t > 0 && !__any(tm != 0u))
            continue;
        const uint2 p = lds_read8<ALIGNED8>(base, blk + term_off(tm));
        const uint32_t m = term_mask(tm);
        lo |= ((p.x >> term_shr(tm)) & m) << term_shl(tm);
        hi |= ((p.y >> term_shr(tm)) & m) << term_shl(tm);
    }
    const uint32_t th = row.w; // Decode10's bits 8..9
    uint32_t lo8 = 0, hi8 = 0;
    if (ALIGNED8 || __any(th != 0u)) {
        const uint2 ph = lds_read8<ALIGNED8>(base, blk + term_off(th));
        const uint32_t mh = term_mask(th);
        lo8 = (ph.x >> term_shr(th)) & mh;
        hi8 = (ph.y >> term_shr(th)) & mh;
    }
    // bytes -> packed u16 pairs: {lo.b0 | lo8.b0 << 8, lo.b1 | lo8.b1 << 8} ...
    u.x[0] = __builtin_amdgcn_perm(lo8, lo, 0x05010400u);
    u.x[1] = __builtin_amdgcn_perm(lo8, lo, 0x07030602u);
    u.x[2] = __builtin_amdgcn_perm(hi8, hi, 0x05010400u);
    u.x[3] = __builtin_amdgcn_perm(hi8, hi, 0x07030602u);
    return u;
}

// ------------------------------------------------------------------ side streams: k7_side
//
// A side stream is a chain of records {hbits<<4 | ref>>8, ref & 255, LEN[hbits] payload
// bytes}: where record i+1 starts is only known from the header of record i
// (RawData.cpp:485-495).  ONE WORKGROUP PER STREAM resolves the chain, decodes the records
// and (bits stream) turns the block lengths into payload offsets -- no hand-off between
// workgroups, one launch for everything in front of k7_tiles.
//
// The stream is taken through LDS in PIECES of 32 KiB (the next two pieces are in registers / on
// their way from HBM meanwhile).  Per piece:
//   B  build: the bytes go to LDS together with a STRIDE TABLE: every second byte is a candidate
//      record start; each thread turns its 16-byte lines into strides (byte-parallel table lookups,
//      in 2-byte units: 1 + LEN/2, RawData.cpp:27-45; 0 = the record would cross `len`, :419-420).
//   W  chain walk by RUN SPECULATION (wave 0): the record at p has stride S; lane j looks up the
//      stride of the candidate at p + j*S.  All lanes up to the first one that finds another stride
//      ARE records -- each is where its predecessor ends -- and that first other lane is the next
//      record, its stride already in hand.  One LDS gather per run of equally long records instead
//      of one dependent read per record: coded frames are made of long runs (~100 steps per UHD
//      stream of 2 025 records; one step per record at worst).  Exact for any content: nothing is
//      assumed, lanes only confirm.
//   D  record decode (waves 1..7, WHILE wave 0 walks the next piece): eight records per wave and
//      pass are unpacked like payload blocks (DecodeBlock on the record, RawData.cpp:489; +
//      reference, :491-492) -> refs[] (u16) / bits[] (u8, validated);
//   S  bits stream: byte length of every decode item, exclusive scan over the workgroup plus the
//      running carry -> payload offset of every item (RawData.cpp:562, :576-579).
//
// The workgroup of the bits stream also publishes the frame's real geometry (Geo7) from the frame
// header (RawData.cpp:500-524, 545-554): everything behind it works from the header, not from
// the caller's width/height.
#ifndef MCRAW_SIDE_T
#define MCRAW_SIDE_T 512
#endif
#ifndef MCRAW_SIDE_LPT
#define MCRAW_SIDE_LPT 4
#endif
#ifndef MCRAW_SIDE_LCAP
#define MCRAW_SIDE_LCAP 512
#endif
#ifndef MCRAW_SPEC_WARM
#define MCRAW_SPEC_WARM 8192
#endif
// (template parameters of the kernel: threads per workgroup, 16-byte lines per thread and piece, records per UNIT of the walk /
// decode pipeline, candidates in front of its pieces at which a speculative count starts.  512 threads, pieces of 32 KiB, 57 KB
// of LDS: the shortest chain.  Round 5 also built a THIN form -- 256 threads, pieces of 12 KiB, 23 KB of LDS -- that fits beside
// the tile kernel's workgroups; what it was for did not pay, docs/lab_notes.md, and it is gone.)
constexpr uint32_t SIDE_XL = 9;                             // lines behind the piece: 130 (reach of its last record) + 8 (read slack) bytes
constexpr uint32_t SIDE_OUT = 0xFFu;                        // stride table: "behind the piece"
constexpr uint32_t SIDE_DEAD = 0xFEu;                       // stride table: a record here would cross `len`
static_assert(ITEM_SPLIT == 2, "k7_side sums the block lengths of half a record per item");

template <int PATTERN>
__device__ __forceinline__ uint32_t swz_xor(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), PATTERN));
}

// Workgroup barrier for LDS traffic only: outstanding buffer loads (the prefetched piece) stay in flight.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 8 candidate record starts of one 16-byte line -> 8 strides in 2-byte units (1 + LEN/2), one byte each.
__device__ __forceinline__ uint2 side_strides(const uint4 v, uint32_t odd)
{
    const uint32_t pick = odd ? 0x07050301u : 0x06040200u; // header bytes sit at every second byte
    uint32_t st[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t h4 = i ? __builtin_amdgcn_perm(v.w, v.z, pick) : __builtin_amdgcn_perm(v.y, v.x, pick);
        const uint32_t hb = (h4 >> 4) & 0x0F0F0F0Fu;
        const uint32_t sel = hb & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(0x08060504u, 0x03020100u, sel); // LEN/8, hbits 0..7
        const uint32_t hi = __builtin_amdgcn_perm(0x10101010u, 0x100A0A08u, sel); // LEN/8, hbits 8..15
        const uint32_t g = (hb >> 3) & 0x01010101u;
        const uint32_t m = byte_mask(g);
        st[i] = (((hi & m) | (lo & ~m)) << 2) + 0x01010101u;
    }
    return make_uint2(st[0], st[1]);
}

#ifdef MCRAW_DIAG // event timeline of one workgroup (timing experiments only; not in the product library)
#ifndef SIDE_PROF_BLOCK
#define SIDE_PROF_BLOCK 1u
#endif
// [0][..]: wave 0 (the walker), [1][..]: wave 1 (a decoder): event id << 48 | shader cycles since the kernel's first stamp;
// entry 0 of each row: number of events
__device__ unsigned long long g_side_prof[2][256];
#define SIDE_STAMP(slot)                                                                                               \
    do {                                                                                                               \
        if (blockIdx.x == SIDE_PROF_BLOCK && (tid == 0u || tid == 64u) && tl_n_ < 255u) {                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
            g_side_prof[tid ? 1 : 0][++tl_n_] = (static_cast<unsigned long long>(slot) << 48) | (now_ - stamp_);       \
            g_side_prof[tid ? 1 : 0][0] = tl_n_;                                                                        \
        }                                                                                                              \
    } while (0)
#else
#define SIDE_STAMP(slot)
#endif

// PARTS: streams may be cut into parts (W.nsplit); false: one workgroup per stream -- the instance every batch of UHD frames runs,
// without the count, the hand-off and the replay in its code (the walker's loop is short of scalar registers as it is).
// LASTC: the last part of a stream counts its pieces too (below) -- an instance of its own for the same reason.
template <uint32_t SIDE_T, uint32_t SIDE_LPT, uint32_t SIDE_LCAP, uint32_t SPEC_WARM, bool PARTS, bool LASTC>
__global__ __launch_bounds__(SIDE_T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k7_side(const Work7 W)
{
    constexpr uint32_t SIDE_PIECE = 16 * SIDE_T * SIDE_LPT;     // stream bytes per piece
    constexpr uint32_t SIDE_HALF = SIDE_PIECE / 2;              // candidate positions per piece (2 bytes apart)
    constexpr uint32_t SIDE_BYTES = SIDE_PIECE + SIDE_XL * 16 + 16;
    __shared__ __attribute__((aligned(16))) uint8_t s_b[SIDE_BYTES];       // bytes of the piece the decoders work on
    __shared__ __attribute__((aligned(16))) uint8_t s_T[SIDE_HALF + 80];   // strides of the piece the walker is in; [SIDE_HALF ..] = SIDE_OUT (the reach of a record)
    __shared__ __attribute__((aligned(16))) uint16_t s_L[2][SIDE_LCAP];
    __shared__ __attribute__((aligned(16))) uint16_t s_len[2][2 * SIDE_LCAP]; // item lengths of a unit (bits stream)
    __shared__ uint4 s_tab[72];
    __shared__ __attribute__((aligned(16))) uint32_t s_st[2][4]; // what the walker reports with list 0 / 1
    __shared__ uint2 s_seg[64]; // segment walkers: {first record of the segment | records in it << 16, records of the piece in front of it}
    __shared__ int32_t s_err;

    // First one side stream of every frame, then the other.  At 240 frames two workgroups share a CU, and they are the ones
    // 256 apart in the launch: with the streams of a frame side by side (2 f + s) those were two of the same kind -- two long
    // ones on half of the CUs --, now they are one of each: k7_side 71 -> 67 us (six interleaved pairs of runs, tools/ab.sh).
    // A long stream of a small batch is cut into PARTS (round 4; W.nsplit[stream]): every part owns a range of the stream's
    // pieces, [m_lo, m_hi), and the records that START in them.  What a part needs to know -- where in its first piece the
    // chain enters, and the index of its first record -- only the chain itself says, so:
    //   phase 1  every part but the last follows the chain through its pieces with the walker alone and COUNTS (no record
    //            is decoded: a unit costs its walk and two barriers).  Part 0 starts on the stream's first record; the others
    //            start half a piece in front of their range on a byte that is most likely no record at all: a wrong chain reads
    //            payload bytes as headers and falls onto the true one within a few hundred bytes (k7_side's segment walkers and
    //            the legacy kernel's rest on the same fact).  All parts do this at the same time.
    //   hand-off part p waits for what part p - 1 says (W.sync: records up to the end of its pieces, and where the chain enters
    //            the next piece), checks that its own chain entered its range exactly there -- nothing is taken on trust --,
    //            and says the same for its own range.  A chain of one hop per part.
    //   phase 2  the pipeline below (walk, decode, offsets) over the part's pieces, from the entry and the record index it now
    //            knows.  A part whose speculation failed starts it from what its predecessor said (and says its own afterwards);
    //            one whose predecessor never speaks (workgroups started out of order and the chip is full) follows the chain
    //            from the stream's first record by itself.  Payload offsets are relative to the part's first item; k7_tiles adds
    //            the totals of the parts in front (Frame7::part_item, part_len).
    const uint32_t nfr = static_cast<uint32_t>(W.n7), nsb = PARTS ? W.nsplit[0] : 1u, nsr = PARTS ? W.nsplit[1] : 1u;
    const uint32_t bq = blockIdx.x / nfr, f = blockIdx.x - bq * nfr;
    const uint32_t s = bq < nsr ? 1u : 0u, part = s ? bq : bq - nsr, nsp = s ? nsr : nsb; // (the longer stream's parts first)
    const uint32_t fs = (nsb + nsr) * f + (s ? nsb : 0u) + part;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MCRAW_DIAG
    const unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    uint32_t tl_n_ = 0, nsteps_ = 0;
#endif
    if (tid < 72u)
        s_tab[tid] = reinterpret_cast<const uint4 *>(c_tab7)[tid];
    if (tid < 80u)
        s_T[SIDE_HALF + tid] = static_cast<uint8_t>(SIDE_OUT);
    if (tid == 0u)
        s_err = 0;

    const Plan7 PL = W.plans[f]; // pinned host memory: one read over the link
    const Plan7 *P = &PL;
    int32_t *status = W.status + fs; // this stream's word
    const uint32_t len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    SIDE_STAMP(10); // plan here

    // ---- frame header (RawData.cpp:500-524) and its checks (:547-554)
    const uint4 hv = ld_b128(rs, 0); // zeros when len < 16
    const uint32_t encW = __builtin_amdgcn_readfirstlane(hv.x), encH = __builtin_amdgcn_readfirstlane(hv.y);
    const uint32_t bitsOff = __builtin_amdgcn_readfirstlane(hv.z), refsOff = __builtin_amdgcn_readfirstlane(hv.w);
    const uint32_t so = s ? refsOff : bitsOff;
    // ---- pieces: s_b[x] holds frame byte A + x; A is 16-byte aligned (relative to the frame buffer: the
    // bounds check of the buffer loads works on whole dwords); the first record sits behind the 4-byte count.
    // Candidate u of a piece is the byte pair at 2u + odd.
    const uint64_t A0 = (static_cast<uint64_t>(so) + 4u) & ~15ull;
    const uint32_t odd = (so + 4u) & 1u;
    struct Lines {
        uint4 v[SIDE_LPT], x;
    };
    auto load_piece = [&](Lines &r, uint32_t piece) {
        const uint64_t base = A0 + static_cast<uint64_t>(piece) * SIDE_PIECE;
#pragma unroll
        for (uint32_t j = 0; j < SIDE_LPT; j++) {
            const uint64_t o = base + 16ull * (tid + SIDE_T * j);
            r.v[j] = o < len ? ld_b128(rs, static_cast<uint32_t>(o)) : make_uint4(0, 0, 0, 0);
        }
        r.x = make_uint4(0, 0, 0, 0);
        if (tid < SIDE_XL) {
            const uint64_t o = base + 16ull * (SIDE_T * SIDE_LPT + tid);
            r.x = o < len ? ld_b128(rs, static_cast<uint32_t>(o)) : make_uint4(0, 0, 0, 0);
        }
    };
    Lines nx, ny; // invariant at the top of the loop below: the two pieces behind the one in s_b (bb + 1, bb + 2)
    load_piece(nx, 0u); // on its way while the header is checked (bounds-checked loads: harmless whatever `so` says)
    int32_t err = 0;
    uint32_t tilesX = 0, nblk = 0, R = 0;
    if (len < 16u || bitsOff > len || refsOff > len || (encW & 63u) != 0u || encW < static_cast<uint32_t>(P->width) ||
        encW == 0u || encH == 0u || (encH & 3u) != 0u ||
        static_cast<uint64_t>(encW) * encH >= (1ull << 31)) {
        err = MCRAW_E_HEADER; // RawData.cpp:547-554 returns 0
    } else {
        tilesX = encW >> 6;
        nblk = 4u * tilesX * (encH >> 2);
        R = (nblk + GROUP_BLOCKS - 1u) / GROUP_BLOCKS;
        // The workspace and the k7_tiles grid were sized from the geometry the host planned with; a frame
        // coded larger than that is planned again from its real header by the host.
        if (R > P->ngroups)
            err = E_GEOMETRY;
        else if (so + 4u > len || so + 4u < so)
            err = MCRAW_E_TRUNCATED;
        else {
            const uint32_t count = ld_u8(rs, so) | (ld_u8(rs, so + 1u) << 8) | (ld_u8(rs, so + 2u) << 16) |
                                   (ld_u8(rs, so + 3u) << 24);
            if (__builtin_amdgcn_readfirstlane(count) < nblk) // the reference would index past the vector (:573-574)
                err = MCRAW_E_SIDESTREAM;
        }
    }
    SIDE_STAMP(11); // header + count here
    Frame7 *const FR = W.frames + f;
    if (s == 0u && part == 0u && tid == 0u) { // (field by field: part_item[q] / part_len[q] of the parts that exist belong to them)
        FR->in = P->in;
        FR->out = P->out;
        FR->len = len;
        FR->width = P->width;
        FR->rows = min(static_cast<uint32_t>(P->height), encH);
        FR->tilesX = tilesX;
        FR->nblk = err ? 0u : nblk; // a frame rejected here is not touched by k7_tiles
        FR->fast_store = P->fast_store;
        FR->encH = encH;
        FR->pad = 0u;
        for (uint32_t q = err ? 0u : nsb - 1u; q < 3u; q++) {
            FR->part_item[q] = 0xFFFFFFFFu;
            FR->part_len[q] = 0u;
        }
        W.status[W.nstatus + f] = static_cast<int32_t>(encH); // read back with the statuses (rows written = min(height, encH))
    }
    if (err) {
        if (tid == 0u)
            *status = err;
        return;
    }
    // The pieces of this part.  Where the stream ends is not known before its chain has been followed; the split is made
    // on a guess -- the stream reaches to where the other one starts, or to the end of the frame, as encoders lay them
    // out -- and any guess gives a partition: the ranges are disjoint, the last part's is open-ended.
    constexpr uint32_t ENDX_ = 0xFFFFFFFFu;
    uint32_t m_lo = 0u, m_hi = 0xFFFFFFFFu, np_guess = 0xFFFFFFFFu;
    if (nsp > 1u) {
        const uint32_t other = s ? bitsOff : refsOff;
        const uint64_t end_guess = other > so ? other : len;
        const uint64_t npieces = (end_guess - A0 + SIDE_PIECE - 1u) / SIDE_PIECE; // (A0 <= so + 4 <= len, end_guess > so)
        np_guess = static_cast<uint32_t>(npieces);
        m_lo = static_cast<uint32_t>(part * npieces / nsp);
        if (part + 1u < nsp)
            m_hi = static_cast<uint32_t>((part + 1u) * npieces / nsp);
    }
    uint64_t *const sync = W.sync + (static_cast<size_t>(2u * f + s) * MAX_SPLIT7 + part) * 2u; // mine; sync[-2], sync[-1]: the part in front
    // Round 5: what a part's count finds out -- where every record of its pieces starts -- is kept (W.rpos, this part's Rmax
    // entries: the candidate index of a record inside its piece, bit 15 on a piece's first record) and its decode REPLAYS it:
    // no second walk along the chain (the bits stream of coded frames is walk-bound: 13 000 cycles per unit of 512 records against
    // 4 400 of decode), no stride tables for pieces that are replayed.  And the LAST part counts as well -- speculatively, up to
    // where the stream is guessed to end -- instead of idling until the part in front has spoken.
    uint16_t *const rp_base = (PARTS && nsp > 1u && W.rpos) ? W.rpos + (static_cast<size_t>(2u * f + s) * MAX_SPLIT7 + part) * W.Rmax : nullptr;
    const __amdgpu_buffer_rsrc_t rsp = frame_rsrc(reinterpret_cast<const uint8_t *>(rp_base), rp_base ? 2u * W.Rmax : 0u);
    uint32_t rleft = 0u, rnext = 0u, rexit = ENDX_; // (uniform, all threads) entries left to replay, the next one, where the chain enters the piece behind them
    bool rdead = false;                            // ... or the chain ended behind them
    constexpr uint32_t ENDX = ENDX_; // hand-off: the stream is over (its last record lies in front, or the chain has ended)

    uint8_t *bits = W.bits + static_cast<size_t>(f) * W.Rmax * 64u;
    uint16_t *refs = W.refs + static_cast<size_t>(f) * W.Rmax * 64u;
    uint32_t *goff = W.grp_off + static_cast<size_t>(f) * (W.Rmax * ITEM_SPLIT + 1u);

    uint32_t upc = 0;                      // piece of the unit just walked (the decoders' next work)
    uint32_t bb = 0;                       // piece whose bytes are in s_b
    uint32_t tb = 0;                       // piece whose strides are in s_T
    uint32_t n = 0;                        // records in front of the unit just walked (their index, when `known_n`)
    uint64_t carry = 0u;                   // payload offset of the next item (RawData.cpp:562), relative to the part's first item
    uint32_t n_first = 0xFFFFFFFFu;        // first record this part decodes (none yet)
    int32_t lane_err = 0;                  // per lane (a bits entry above 16)
    bool dead = false;                     // uniform: the chain ended before R records
    uint32_t decode_from = 0u;             // units of pieces in front of this one are walked, not decoded
    constexpr uint32_t NOENTRY = 0xFFFFFFFEu;
    // candidates (2 bytes each) in front of its pieces at which a speculative count starts.  Half a piece: with 2 KiB (what the
    // segment walkers start with) enough counts of four-part streams arrived on a wrong chain -- and then decode from what the
    // part in front says, one part after the other -- that four parts were slower than two (tools/side_warm2.sh: 16 x 12 MP
    // natural frames 110 -> 85 us, 14-bit noise 205 -> 162 us with 16 KiB)
    static_assert(SPEC_WARM <= SIDE_HALF, "a speculative count starts inside the piece in front of the part's first");

    // registers -> bytes of a piece (what the decoders read)
    auto store_bytes = [&](const Lines &r) {
#pragma unroll
        for (uint32_t j = 0; j < SIDE_LPT; j++)
            reinterpret_cast<uint4 *>(s_b)[tid + SIDE_T * j] = r.v[j];
        if (tid < SIDE_XL)
            reinterpret_cast<uint4 *>(s_b)[SIDE_T * SIDE_LPT + tid] = r.x;
    };
    // registers -> stride table of `piece` (what the walker reads)
    auto build_strides = [&](const Lines &r, uint32_t piece) {
        const uint64_t base = A0 + static_cast<uint64_t>(piece) * SIDE_PIECE;
        const uint64_t left = len - min(static_cast<uint64_t>(len), base);
        const uint32_t lim = static_cast<uint32_t>(min(left, static_cast<uint64_t>(1u << 30))); // frame bytes from `base` on
#pragma unroll
        for (uint32_t j = 0; j < SIDE_LPT; j++) {
            const uint32_t line = tid + SIDE_T * j;
            uint2 st = side_strides(r.v[j], odd);
            if (16u * line + 16u + 130u > lim) { // near the end of the frame: a record that would cross `len` ends the chain
                uint32_t w[2] = {st.x, st.y};
#pragma unroll
                for (uint32_t i = 0; i < 8u; i++) {
                    const uint32_t sv = (w[i >> 2] >> (8u * (i & 3u))) & 0xffu;
                    if (16u * line + 2u * i + odd + 2u * sv > lim)
                        w[i >> 2] = (w[i >> 2] & ~(0xffu << (8u * (i & 3u)))) | (SIDE_DEAD << (8u * (i & 3u)));
                }
                st = make_uint2(w[0], w[1]);
            }
            reinterpret_cast<uint2 *>(s_T)[line] = st;
        }
    };
    // W (wave 0): list up to `room` (>= 1) records of the piece in s_T from candidate pu on; result in s_st:
    // where the chain stands, records listed, why it stopped (1 list full / stream complete, 2 behind the
    // piece, 3 chain dead) and the stride of the record it stands on when that is known (else 0).
    //
    // Two ways to follow the chain.  RUN SPECULATION (above): one pass per run of equally long records -- what coded
    // frames consist of.  A stream whose records keep changing size (the refs of a noise frame: 66- and 82-byte records
    // taking turns, 2.2 records per pass) is followed by SEGMENT WALKERS instead (`segw`, from then on for the rest of
    // the stream): lane j owns segment j of the piece (SEG_C candidates from where the chain stands) and walks one record
    // per step like the reference does, all lanes side by side.  Lanes of the first WARM_C candidates start on the
    // chain itself; the others start WARM_C candidates in front of their segment at a candidate that is most likely no
    // record at all: a wrong chain reads payload bytes as headers and falls onto the true one with probability
    // ~ 1/37 per step, so that after WARM_C candidates it has done so in 97 % of the cases.  Nothing is taken on trust:
    // lane j's first record in its segment must be where lane j - 1 left its own (the lanes on the chain itself are the
    // induction's start); the first lane for which that fails walks its segment again from there, and so on (each lane
    // at most once).  A piece then costs ~ (WARM_C + SEG_C) / 30 dependent LDS reads instead of one pass per run.
    #ifndef MCRAW_SEGW_RATIO
#define MCRAW_SEGW_RATIO 3u
#endif
#ifndef MCRAW_WARM_SEGS
#define MCRAW_WARM_SEGS 4
#endif
    constexpr uint32_t SEG_C = SIDE_HALF / 64u, WARM_C = MCRAW_WARM_SEGS * SEG_C, SEG_NONE = 0x7FFFu, SEG_DEADX = 0xFFFFu;
#ifdef MCRAW_FORCE_SEGW // test builds: every stream on the segment walkers from its first record on (tools/test_segw.sh)
    bool segw = true, seg_valid = false;
#else
    bool segw = false, seg_valid = false;          // (uniform, wave 0)
#endif
    uint32_t seg_total = 0, seg_done = 0, seg_exit = 0;
    // one lane = one chain: from `pos` to the end of the lane's segment [segs, sege); -> first record in the segment,
    // records started in it, and where the chain leaves it (SEG_DEADX: it ended, a record would cross `len`)
    auto seg_chain = [&](bool on, uint32_t pos, uint32_t segs, uint32_t sege, uint32_t &ent, uint32_t &cnt, uint32_t &ex) {
        // branch-free: a lane that has left its segment, or stands in front of a record that would cross `len`, stays
        // where it is (stride 0); the dependent chain of a step is one LDS read, a compare, a select and an addition
        pos = on ? pos : SIDE_HALF;
        sege = on ? sege : 0u;
        bool stop;
        do {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t code = s_T[pos]; // (pos < sege + 65 <= SIDE_HALF + 65)
                stop = pos >= sege || code >= SIDE_DEAD;
                const bool inseg = !stop && pos >= segs;
                ent = min(ent, inseg ? pos : SEG_NONE);
                cnt += inseg ? 1u : 0u;
                pos += stop ? 0u : code;
            }
#ifdef MCRAW_DIAG
            nsteps_ += 2;
#endif
        } while (__any(!stop));
        if (on)
            ex = pos >= sege ? pos : SEG_DEADX;
    };
    auto seg_walk = [&](uint32_t org) {
        const uint32_t segs = org + lane * SEG_C;
        const bool act = segs < SIDE_HALF;
        const uint32_t sege = min(segs + SEG_C, SIDE_HALF);
        const bool on_chain = lane < WARM_C / SEG_C;
        uint32_t ent = SEG_NONE, cnt = 0, ex = SEG_DEADX;
        seg_chain(act, on_chain ? org : segs - WARM_C, segs, sege, ent, cnt, ex);
        // Lane j's chain is the true one in its segment iff it enters it where lane j - 1's leaves its own.  Every lane
        // for which that fails walks its segment again from where its predecessor says, all of them side by side; the
        // lowest such lane is right afterwards for good (the lanes below it are), so the rounds end -- after one or
        // two: wrong lanes are few and seldom neighbours, and a corrected lane mostly leaves its segment where it did before.
        uint32_t src = SEG_NONE; // where a lane that walked again started from (its chain may end right there)
        for (uint32_t round = 0; round < 64u; round++) { // (the lowest wrong lane is right after every round)
            const uint32_t want = wave_prev(ex, ex); // (lane 0: its own, never looked at -- it is on the chain)
            const uint32_t claim = src != SEG_NONE ? src : ent != SEG_NONE ? ent : ex;
            const bool bad = act && !on_chain && claim != want;
            if (__ballot(bad) == 0ull)
                break;
            if (bad) {
                ent = SEG_NONE;
                cnt = 0;
                ex = SEG_DEADX;
                src = want;
            }
            seg_chain(bad && want != SEG_DEADX, want, segs, sege, ent, cnt, ex);
        }
        uint32_t total;
        const uint32_t off = wave_excl_scan(act ? cnt : 0u, lane, &total);
        s_seg[lane] = make_uint2((ent & 0xFFFFu) | (cnt << 16), off);
        const uint32_t nact = (SIDE_HALF - org + SEG_C - 1u) / SEG_C; // (org < SIDE_HALF)
        seg_exit = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ex), static_cast<int>(min(nact, 64u) - 1u)));
        seg_total = total;
        seg_done = 0;
        seg_valid = true;
    };
    // records [seg_done, seg_done + take) of the piece, in chain order, into the list: every lane walks its segment
    // once more, from its (now known) first record
    auto seg_list = [&](uint16_t *L, uint32_t take) {
        const uint2 sv = s_seg[lane];
        const uint32_t cnt = sv.x >> 16, off = sv.y, lo = seg_done, hi = seg_done + take;
        uint32_t pos = sv.x & 0xFFFFu, kk = off;
        const uint32_t kend = min(off + cnt, hi);
        bool part = cnt != 0u && off < hi && off + cnt > lo;
        pos = part ? pos : SIDE_HALF;
        while (__any(part)) {
            const uint32_t code = s_T[pos];
            if (part && kk >= lo)
                L[kk - lo] = static_cast<uint16_t>(pos);
            kk++;
            part = part && kk < kend;
            pos = part ? pos + code : SIDE_HALF;
        }
    };
    auto walk = [&](uint32_t lst, uint32_t pu, uint32_t SU, uint32_t room, bool fresh) {
        uint16_t *L = s_L[lst];
        uint32_t cnt = 0, why = 2u;
        if (!segw && pu < SIDE_HALF) {
            // one pass per run, one exit test per pass
            uint32_t nb, take, nxt, passes = 0;
            bool go;
            do {
                const uint32_t u = pu + lane * SU;
                const uint32_t code = s_T[min(u, SIDE_HALF)];
                const unsigned long long m = __ballot(code != SU);
                nb = m == 0ull ? 64u : static_cast<uint32_t>(__builtin_ctzll(m));
                take = min(nb, room - cnt);
                if (lane < take)
                    L[cnt + lane] = static_cast<uint16_t>(u);
                // lane nb: the record behind the run, if it is one (SIDE_OUT: behind the piece, SIDE_DEAD: it would cross `len`)
                nxt = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(code), static_cast<int>(nb & 63u)));
                cnt += take;
                pu += take * SU;
                go = take == nb && cnt < room && (nb == 64u || nxt < SIDE_DEAD);
                SU = (go && nb != 64u) ? nxt : SU;
                passes++;
                // Fewer than three records per pass: the records keep changing size, runs do not pay here (a pass costs ~350
                // cycles whatever it lists); the segment walkers take over where the chain stands (a record that is not listed
                // yet), for the rest of the stream.  The bar is low on purpose: a chain that starts on payload bytes advances by
                // what those bytes say, and the small residuals of coded frames make that a crawl (two bytes a step) -- a
                // stream that lists 15 records per pass is followed three times FASTER by runs than by the walkers, measured.
                segw = go && passes >= 8u && passes * MCRAW_SEGW_RATIO > cnt;
#ifdef MCRAW_DIAG
                nsteps_++;
#endif
            } while (go && !segw);
            if (segw) {
            } else if (take < nb) {
                why = 1u; // list full / stream complete inside the run: pu is a record of stride SU
            } else if (cnt >= room) {
                why = 1u; // ... at the end of the run
                SU = (nb != 64u && nxt < SIDE_DEAD) ? nxt : 0u;
            } else {
                why = nxt == SIDE_OUT ? 2u : 3u;
                SU = 0u;
            }
            if (segw)
                seg_valid = false;
        } else if (!segw) {
            SU = 0u;
        }
        if (segw && pu < SIDE_HALF) {
            if (fresh || !seg_valid)
                seg_walk(pu);
            const uint32_t take = min(room - cnt, seg_total - seg_done);
            if (take)
                seg_list(L + cnt, take);
            seg_done += take;
            cnt += take;
            SU = 0u;
            if (seg_done == seg_total) { // the piece is through
                why = seg_exit == SEG_DEADX ? 3u : 2u;
                pu = seg_exit == SEG_DEADX ? 0u : seg_exit;
                seg_valid = false;
            } else {
                why = 1u;
                pu = 0u; // (where the chain stands is in s_seg)
            }
        } else if (segw) {
            SU = 0u;
        }
        if (lane == 0u)
            *reinterpret_cast<uint4 *>(s_st[lst]) = make_uint4(pu, cnt, why, SU);
    };

    const uint32_t k = lane & 7u, sub = lane >> 3;
    const uint32_t first_cand = static_cast<uint32_t>((so + 4u) & 15u) >> 1; // the stream's first record, behind the 4-byte count
    // S (wave 0, bits stream): item lengths of one unit -> payload offsets: exclusive scan, eight items per lane and pass
    static_assert((2u * SIDE_LCAP) % 512u == 0u, "whole passes of 512 items");
    auto scan_unit = [&](uint32_t par, uint32_t n0, uint32_t cnt) {
        const uint32_t nitems = 2u * cnt;
        for (uint32_t base = 0; base < nitems; base += 512u) {
            const uint32_t i0 = base + 8u * lane;
            uint4 lv = make_uint4(0u, 0u, 0u, 0u);
            if (i0 < nitems)
                lv = *reinterpret_cast<const uint4 *>(&s_len[par][i0]);
            uint32_t a[8] = {lv.x & 0xffffu, lv.x >> 16, lv.y & 0xffffu, lv.y >> 16,
                             lv.z & 0xffffu, lv.z >> 16, lv.w & 0xffffu, lv.w >> 16};
            uint32_t mine = 0;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                a[i] = i0 + i < nitems ? a[i] : 0u;
                mine += a[i];
            }
            uint32_t utot;
            const uint32_t ex = wave_excl_scan(mine, lane, &utot);
            // offsets past 2^32 only occur in frames that fail the `len` check at the end
            uint32_t o = static_cast<uint32_t>(carry) + ex;
#pragma unroll
            for (uint32_t i = 0; i < 8u; i++) {
                if (i0 + i < nitems)
                    goff[2u * n0 + i0 + i] = o;
                o += a[i];
            }
            carry += utot;
        }
    };

    // ---- hand-off between the parts of a stream
#ifdef MCRAW_INJECT_MUTE7 // test builds (tests/test_gpu_split.py): the first part of every stream never tells, the others give up soon
    constexpr uint32_t SPIN7 = 1u << 6;
#else
    constexpr uint32_t SPIN7 = 1u << 19; // polls before a part stops waiting and follows the chain from the stream's start by itself
#endif
    __shared__ uint32_t s_ask[4];
    bool told = part + 1u >= nsp; // what the next part needs has been said (the last part has nobody to tell)
    auto tell = [&](uint32_t count, uint32_t where) { // records in front of the next part's pieces; candidate at which the chain enters them
        if (!told && tid == 0u) {
#ifdef MCRAW_INJECT_MUTE7
            if (part != 0u)
#endif
            {
                look_put(sync, W.epoch, count);
                look_put(sync + 1, W.epoch, where);
            }
        }
        told = true;
    };
    // what the part in front says (all threads; false: it never spoke)
    auto ask = [&](uint32_t &count, uint32_t &where) {
        lds_barrier();
        if (tid == 0u) {
            uint32_t c = 0u, w = first_cand, ok = 1u;
            if (part) {
                ok = 0u;
                for (uint32_t spin = 0; spin < SPIN7; spin++) {
                    if (look_get(sync - 2, W.epoch, &c) && look_get(sync - 1, W.epoch, &w)) {
                        ok = 1u;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            s_ask[0] = c;
            s_ask[1] = w;
            s_ask[2] = ok;
        }
        lds_barrier();
        count = s_ask[0];
        where = s_ask[1];
        return s_ask[2] != 0u;
    };
    auto nothing_to_decode = [&]() { // (bits stream: this part adds nothing to the payload offsets)
        if (tid == 0u && s == 0u && nsp > 1u) {
            if (part + 1u < nsp)
                FR->part_len[part] = 0u;
            if (part)
                FR->part_item[part - 1u] = 0xFFFFFFFFu;
        }
    };

    // ---- phase 1 (count) and the hand-off: where phase 2 starts
    uint32_t run_piece = 0u, run_cand = first_cand, run_n = 0u; // piece, candidate in it, records in front of it
    uint32_t nx_piece = 0u;                                       // the piece that is on its way into nx
    uint32_t exitc = ENDX;                                        // where the chain enters piece m_hi (phase 2, when it still has to be said)
    bool go = true;
    if (nsp > 1u) {
        const bool empty = m_lo >= m_hi, lastp = part + 1u >= nsp;
        const bool owns0 = !empty && m_lo == 0u; // the part that owns piece 0 (part 0, or the parts in front of this one own nothing)
        // (the last part's pieces are open-ended; its count goes as far as the stream is guessed to reach -- any guess will do: what
        // lies behind the count is walked by the decode as before)
        const uint32_t m_chi = lastp ? np_guess : m_hi;
        // (it pays while the chip has room -- the count is work on top, 8K frames' streams by the hundred run 25 % longer with
        // it --: the host says so, W.side_lastc, and launches the instance that has it)
        const bool lastc = LASTC && lastp && !owns0 && rp_base != nullptr && m_chi > m_lo && m_chi != 0xFFFFFFFFu;
        const bool spec = !empty && !owns0 && (!lastp || lastc);
        uint32_t entry = NOENTRY, cnt1 = 0u, exit1 = ENDX; // where the chain enters piece m_lo; records of the part's pieces; where it enters piece m_hi
        bool dead1 = false;                                // ... it ended inside the part's pieces
        bool counted = false;
        if (!empty && (!lastp || lastc)) {
            counted = true;
            uint32_t spiece = 0xFFFFFFFFu; // piece of the unit stored last (its first record carries the piece flag)
            // The count: the walker alone, over the part's pieces -- from the stream's first record, or (spec) from SPEC_WARM
            // candidates in front of them, on a byte that is most likely no record at all.
            uint32_t cpiece = spec ? m_lo - 1u : 0u, lst = 0u;
            if (spec)
                load_piece(nx, cpiece);
            build_strides(nx, cpiece);
            load_piece(nx, cpiece + 1u);
            nx_piece = cpiece + 1u;
            lds_barrier();
            if (wave == 0u)
                walk(0u, spec ? SIDE_HALF - SPEC_WARM : first_cand, 0u, spec ? SIDE_LCAP : min(R, SIDE_LCAP), true);
            for (;;) {
                lds_barrier();
                const uint4 st4 = *reinterpret_cast<const uint4 *>(s_st[lst]);
                const uint32_t pu = __builtin_amdgcn_readfirstlane(st4.x);
                const uint32_t total = __builtin_amdgcn_readfirstlane(st4.y);
                const uint32_t why = __builtin_amdgcn_readfirstlane(st4.z);
                const uint32_t SU = __builtin_amdgcn_readfirstlane(st4.w);
                const uint32_t npc = why == 2u ? cpiece + 1u : cpiece;
                if (cpiece >= m_lo) {
                    if (rp_base && total) { // the unit's records, for the decode (every thread stores; the walker is on the next unit soon)
                        const uint16_t *Lc = s_L[lst];
                        for (uint32_t i = tid; i < total; i += SIDE_T)
                            if (cnt1 + i < W.Rmax)
                                rp_base[cnt1 + i] = static_cast<uint16_t>(Lc[i] | ((i == 0u && cpiece != spiece) ? 0x8000u : 0u));
                        spiece = cpiece;
                    }
                    cnt1 += total;
                    dead1 = dead1 || why == 3u;
                }
                if (npc > cpiece && npc == m_lo) {
                    entry = pu - SIDE_HALF;
#ifndef MCRAW_FORCE_SEGW // (what the chain looked like while it ran over payload bytes says nothing about the stream)
                    segw = false;
#endif
                    seg_valid = false;
                }
                if (npc > cpiece && npc == m_chi)
                    exit1 = pu - SIDE_HALF;
                if ((!spec && cnt1 >= R) || why == 3u || npc >= m_chi)
                    break;
                if (npc > cpiece) { // (the walker has left the piece whose strides are in s_T)
                    build_strides(nx, npc);
                    load_piece(nx, npc + 1u);
                    nx_piece = npc + 1u;
                    lds_barrier();
                }
                if (wave == 0u)
                    walk(lst ^ 1u, npc > cpiece ? pu - SIDE_HALF : pu, SU, spec ? SIDE_LCAP : min(R - cnt1, SIDE_LCAP), npc > cpiece);
                cpiece = npc;
                lst ^= 1u;
            }
            SIDE_STAMP(20); // count over
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the stored positions are on their way to memory before any barrier below)
        }
        const auto replay_from_count = [&]() { // the count stands: its records are replayed by the decode
            if (rp_base && counted && cnt1) {
                rleft = min(cnt1, W.Rmax);
                rnext = 0u;
                rexit = exit1;
                rdead = dead1;
            }
        };
        // The hand-off: what the part in front says, what this part says, where its decode starts.
        uint32_t pn = 0u, pw = first_cand;
        const bool ok = owns0 ? true : ask(pn, pw);
        SIDE_STAMP(21); // the part in front has spoken
        decode_from = m_lo;
        if (!ok) { // it never spoke: from the stream's first record on, by itself (and this part says its own at the end)
            if (empty) { // (a part that owns nothing only has to find out what to pass on)
                decode_from = 0xFFFFFFFFu;
                if (m_hi == 0u) {
                    tell(0u, first_cand);
                    nothing_to_decode();
                    go = false;
                }
            }
        } else if (!owns0 && (pw == ENDX || pn >= R)) { // the stream is over in front of this part
            tell(pn, ENDX);
            nothing_to_decode();
            go = false;
        } else if (empty) { // nothing of the stream falls to this part: pass it on
            tell(pn, pw);
            nothing_to_decode();
            go = false;
        } else if (lastp) {
            run_piece = m_lo, run_cand = owns0 ? first_cand : pw, run_n = pn;
            if (lastc && pw == entry)
                replay_from_count();
        } else if (owns0 || pw == entry) { // the count stands: say it, then decode
            const uint32_t upto = pn + cnt1;
            tell(min(upto, R), upto >= R || dead1 || exit1 == ENDX ? ENDX : exit1);
            run_piece = m_lo, run_cand = owns0 ? first_cand : entry, run_n = pn;
            replay_from_count();
        } else { // the chain enters this part's pieces elsewhere: decode from there, and say afterwards what comes out of it
            run_piece = m_lo, run_cand = pw, run_n = pn;
        }
        if (go && run_n >= R) { // (nothing left of the stream)
            tell(run_n, ENDX);
            nothing_to_decode();
            go = false;
        }
    }

    uint32_t cur = 0;                 // list (and item-length buffer) of the unit the decoders work on
    uint32_t prev_n = 0, prev_total = 0; // the unit before it, not yet scanned
    if (go) {
    // ---- phase 2.  Prologue: the first piece in LDS, the next two on their way, first unit walked
    lds_barrier(); // (every wave is done with the count)
    upc = bb = tb = run_piece;
    n = run_n;
#ifdef MCRAW_FORCE_SEGW
    segw = true;
#else
    segw = false;
#endif
    seg_valid = false;
    carry = run_piece == 0u && decode_from == 0u ? 16u : 0u; // (the part that decodes the stream's first record starts behind the 16-byte header)
    if (nx_piece != run_piece)
        load_piece(nx, run_piece);
    store_bytes(nx);
    SIDE_STAMP(12); // first piece arrived
    if (!rleft) // (a replayed piece needs no stride table)
        build_strides(nx, run_piece);
    load_piece(nx, run_piece + 1u);
    load_piece(ny, run_piece + 2u);
    // Replay (wave 0): the next unit's records out of the count's notes instead of a walk.  64 lanes x 8 entries are in
    // registers, fetched one unit ahead (past the L1: other waves of this workgroup stored them); a unit ends in front of an
    // entry that opens the next piece, like a walked one.
    uint4 rq = make_uint4(0u, 0u, 0u, 0u);
    auto rfetch = [&](uint32_t at) { rq = ld_b128_nt(rsp, 2u * (at + 8u * lane)); }; // (bounds-checked: zeros behind the part's entries)
    auto replay = [&](uint32_t lst, uint32_t room) {
        const uint32_t n = min(room, rleft);
        const uint32_t w4[4] = {rq.x, rq.y, rq.z, rq.w};
        uint32_t mine = 8u, ment = 0u; // my first entry (behind the unit's first) that opens a piece
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            const uint32_t idx = 8u * lane + static_cast<uint32_t>(i), e = (w4[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            const bool hit = idx >= 1u && idx < n && (e & 0x8000u) != 0u;
            mine = hit ? static_cast<uint32_t>(i) : mine;
            ment = hit ? e : ment;
        }
        const unsigned long long m = __ballot(mine != 8u);
        uint32_t take = n, pu = 0u, why = 1u;
        if (m) {
            const uint32_t fl = static_cast<uint32_t>(__builtin_ctzll(m));
            take = 8u * fl + wave_lane(mine, fl);
            pu = SIDE_HALF + (wave_lane(ment, fl) & 0x7fffu);
            why = 2u;
        } else if (n == rleft) { // the count's last records: what it found behind them
            if (rdead)
                why = 3u;
            else if (rexit != ENDX)
                why = 2u, pu = SIDE_HALF + rexit;
        }
        if (8u * lane < take)
            *reinterpret_cast<uint4 *>(&s_L[lst][8u * lane]) = make_uint4(rq.x & 0x7fff7fffu, rq.y & 0x7fff7fffu, rq.z & 0x7fff7fffu, rq.w & 0x7fff7fffu);
        if (lane == 0u)
            *reinterpret_cast<uint4 *>(s_st[lst]) = make_uint4(pu, take, why, 0u);
        rfetch(rnext + take);
    };
    lds_barrier();
    SIDE_STAMP(13);
    if (wave == 0u) { // a short first unit: the decoders start early
        if (rleft) {
            rfetch(0u);
            replay(0u, min(R - run_n, SIDE_LCAP / 4u));
        } else
            walk(0u, run_cand, 0u, min(R - run_n, SIDE_LCAP / 4u), true);
    }
    SIDE_STAMP(14);
    while (true) {
        lds_barrier();
        SIDE_STAMP(0);
        // the unit just walked: `total` records of piece upc in list `cur`; where the walker stands
        const uint4 st4 = *reinterpret_cast<const uint4 *>(s_st[cur]);
        const uint32_t pu = __builtin_amdgcn_readfirstlane(st4.x);
        const uint32_t total = __builtin_amdgcn_readfirstlane(st4.y);
        const uint32_t why = __builtin_amdgcn_readfirstlane(st4.z);
        const uint32_t SU = __builtin_amdgcn_readfirstlane(st4.w);
        const uint32_t npc = why == 2u ? upc + 1u : upc; // piece of the next unit
        const bool last = n + total >= R || why == 3u || npc >= m_hi; // (... or the next unit is the next part's)
        const bool skip = upc < decode_from;             // a unit in front of this part's pieces: walked, not decoded
        if (!skip && n_first == 0xFFFFFFFFu)
            n_first = n;
        if (npc > upc && npc == m_hi)
            exitc = pu - SIDE_HALF;
        if (rleft) { // (that unit was replayed: all threads keep count of what is left)
            rleft -= min(total, rleft);
            rnext += total;
        }
        const bool moved = upc > bb, build = !last && npc > tb && rleft == 0u;
        if (moved) { // the decoders move on to piece bb + 1: every unit of piece bb has been decoded
            store_bytes(nx);
            bb = upc;
            nx = ny;
            load_piece(ny, bb + 2u);
        }
        if (build) { // the walker moves on to piece bb + 1: it has left the piece whose strides are in s_T
            build_strides(nx, npc);
            tb = npc;
        }
        if (moved || build)
            lds_barrier();
        SIDE_STAMP(1);
        if (wave == 0u) {
            if (!last) {
                if (rleft)
                    replay(cur ^ 1u, min(R - (n + total), SIDE_LCAP));
                else
                    walk(cur ^ 1u, npc > upc ? pu - SIDE_HALF : pu, SU, min(R - (n + total), SIDE_LCAP), npc > upc);
            }
            SIDE_STAMP(2);
            if (s == 0u && prev_total) // the unit the decoders finished before the last barrier
                scan_unit(cur ^ 1u, prev_n, prev_total);
            SIDE_STAMP(5);
        } else if (!skip) {
            // D: record decode, lane = (record, k) owns samples 8k..8k+7; the list entry and header of the
            // next pass are fetched while this pass is unpacked
            const uint8_t *B = s_b;
            const uint16_t *L = s_L[cur];
            constexpr uint32_t STEP = (SIDE_T / 64u - 1u) * 8u;
            uint32_t q = (wave - 1u) * 8u + sub;
            uint32_t ro_n = q < total ? 2u * L[q] + odd : 0u;
            uint32_t hd_n = *reinterpret_cast<const uint16_t *>(B + (ro_n & ~1u)) | (static_cast<uint32_t>(B[(ro_n & ~1u) + 2u]) << 16);
            for (uint32_t qb = (wave - 1u) * 8u; qb < total; qb += STEP, q += STEP) {
                const bool live = q < total;
                const uint32_t ro = ro_n;
                const uint32_t hd = hd_n >> (8u * (ro & 1u)); // header bytes b0, b1 (the stream may sit at odd addresses)
                ro_n = q + STEP < total ? 2u * L[q + STEP] + odd : 0u;
                hd_n = *reinterpret_cast<const uint16_t *>(B + (ro_n & ~1u)) | (static_cast<uint32_t>(B[(ro_n & ~1u) + 2u]) << 16);
                const uint32_t b0 = live ? hd & 0xffu : 0u, b1 = (hd >> 8) & 0xffu;
                const uint32_t hb = b0 >> 4, ref = ((b0 & 15u) << 8) | b1; // RawData.cpp:106-110
                Unpacked U = unpack8<false>(B, ro + 2u, cls7_of(hb), k, s_tab);
                const u16x2 rr = __builtin_bit_cast(u16x2, ref | (ref << 16));
#pragma unroll
                for (int i = 0; i < 4; i++) // uint16 wrap (RawData.cpp:492)
                    U.x[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, U.x[i]) + rr);
                const uint32_t r = n + q;
                const uint32_t idx = r * 64u + 8u * k;
                if (s == 1u) {
                    if (live)
                        *reinterpret_cast<uint4 *>(refs + idx) = make_uint4(U.x[0], U.x[1], U.x[2], U.x[3]);
                    continue;
                }
                // bits stream: validate, narrow to bytes, and add up the byte length of the lane's 8 blocks
                const uint32_t nvalid = live ? min(8u, nblk - min(nblk, idx)) : 0u;
                if (nvalid < 8u) { // the record's entries past the last block are decoded by the reference but never used
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++)
                        U.x[i] &= nvalid >= 2u * i + 2u ? 0xffffffffu : nvalid == 2u * i + 1u ? 0xffffu : 0u;
                }
                uint32_t c[4], over = 0;
#pragma unroll
                for (uint32_t i = 0; i < 4u; i++) { // an entry above 16 would index past ENCODING_BLOCK_LENGTH (RawData.cpp:419)
                    c[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, U.x[i]),
                                                                                  __builtin_bit_cast(u16x2, 0x00100010u)));
                    over |= c[i] ^ U.x[i];
                }
                if (over)
                    lane_err |= MCRAW_E_SIDESTREAM;
                const uint32_t bytes_lo = __builtin_amdgcn_perm(c[1], c[0], 0x06040200u);
                const uint32_t bytes_hi = __builtin_amdgcn_perm(c[3], c[2], 0x06040200u);
                if (live)
                    *reinterpret_cast<uint2 *>(bits + idx) = make_uint2(bytes_lo, bytes_hi);
                // LEN[v] / 8 for four entries at a time (RawData.cpp:27-45): two 8-entry byte tables picked by
                // bit 3, 16 for v == 16; then a byte sum
                uint32_t l8 = 0;
#pragma unroll
                for (uint32_t i = 0; i < 2u; i++) {
                    const uint32_t v4 = i ? bytes_hi : bytes_lo;
                    const uint32_t sel = v4 & 0x07070707u;
                    const uint32_t lo = __builtin_amdgcn_perm(0x08060504u, 0x03020100u, sel); // v = 0..7
                    const uint32_t hi = __builtin_amdgcn_perm(0x10101010u, 0x100A0A08u, sel); // v = 8..15
                    const uint32_t g = (v4 >> 3) & 0x01010101u;
                    const uint32_t m = byte_mask(g);
                    const uint32_t l4 = ((hi & m) | (lo & ~m)) | (v4 & 0x10101010u);
                    l8 = __builtin_amdgcn_sad_u8(l4, 0u, l8);
                }
                // sum over the four lanes of one decode item (ds_swizzle bit-mask mode: lane ^ 1, ^ 2)
                l8 += swz_xor<0x041F>(l8);
                l8 += swz_xor<0x081F>(l8);
                if (live && (k & 3u) == 0u)
                    s_len[cur][2u * q + (k >> 2)] = static_cast<uint16_t>(l8 << 3); // <= 32 * 128 bytes
            }
            SIDE_STAMP(3);
        }
        prev_n = n;
        prev_total = skip ? 0u : total;
        n += total;
        if (last) {
            dead = why == 3u && n < R && !skip; // (a chain that ends in front of this part's pieces is the owner's to report)
            break;
        }
        upc = npc;
        cur ^= 1u;
    }
    tell(min(n, R), n >= R || dead || exitc == ENDX ? ENDX : exitc); // (a part that could not say it behind its count)
    if (s == 0u && prev_total) { // the last unit's offsets
        lds_barrier();
        if (wave == 0u)
            scan_unit(cur, prev_n, prev_total);
    }
    if (tid == 0u) {
        if (dead) // a record crosses `len` before the stream has its R records (RawData.cpp:419-420)
            lane_err |= MCRAW_E_TRUNCATED;
        if (s == 0u) {
            const uint32_t c32 = static_cast<uint32_t>(min(carry, static_cast<uint64_t>(0xffffffffu)));
            if (n >= R && !dead && n_first != 0xFFFFFFFFu) { // the part that decodes the stream's last record: where the payload ends
                goff[ITEM_SPLIT * R] = c32;
                if (nsp == 1u && carry > len) // some block crosses `len` (RawData.cpp:419-420); with several parts k7_tiles
                    lane_err |= MCRAW_E_TRUNCATED; // makes this check, behind the sum over the parts
            }
            if (nsp > 1u) {
                if (part + 1u < nsp)
                    FR->part_len[part] = n_first != 0xFFFFFFFFu ? c32 : 0u;
                if (part)
                    FR->part_item[part - 1u] = n_first != 0xFFFFFFFFu ? ITEM_SPLIT * n_first : 0xFFFFFFFFu;
            }
        }
    }
    }
    // one status word per stream, written once (no initialisation needed in front of the kernel)
    if (lane_err)
        atomicOr(&s_err, lane_err);
    lds_barrier();
    if (tid == 0u)
        *status = s_err;
    SIDE_STAMP(15);
#ifdef MCRAW_DIAG
    if (blockIdx.x == SIDE_PROF_BLOCK && tid == 0u)
        g_side_prof[0][255] = nsteps_;
#endif
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
constexpr int PAY_LDS = ITEM_SPAN + 16 + 32; // span + 16-B alignment head + slack for zero-length tails
constexpr uint32_t PAY_CHUNKS = ITEM_SPAN / 16 + 1; // 16-byte chunks of a staged span

// Wave-uniform description of one work item (frame f, group g).
struct ItemS {
    uint32_t valid;      // 0: nothing to do (group beyond the frame, or the frame already failed)
    uint32_t g, nblk, tilesX;
    uint32_t base16, n16, head; // 16-byte aligned span start, chunks to stage, start - base16
    int32_t width, rows;
    uint32_t fast;
    const uint8_t *in;
    uint32_t len;
    uint16_t *out;
    size_t meta;         // index of the group's first entry in W.bits / W.refs
    uint32_t lean;       // (strip rows) no sample of the item can reach 2^bits, black levels are off its references
};

__device__ __forceinline__ ItemS item_scalars(const Work7 &W, uint32_t item, uint32_t first_frame, uint32_t class_groups)
{
    ItemS I;
    const uint32_t per = W.Rmax * ITEM_SPLIT;         // items of the uniform-stride workspace per frame
    const uint32_t cper = class_groups * ITEM_SPLIT;  // items this launch spends on each frame of its size class
    const uint32_t fc = item / cper;
    const uint32_t f = first_frame + fc;
    const uint32_t g = item - fc * cper;
    const Frame7 *F = W.frames + f; // plan + header geometry, written by k7_side
    const uint32_t nblk = F->nblk;
    const uint32_t *grp = W.grp_off + static_cast<size_t>(f) * (per + 1u) + g;
    // a frame whose side streams failed behind the header checks is decoded from whatever the workspace
    // holds (its status says so; every read below is bounds-checked and every offset clamped)
    I.valid = g * ITEM_BLOCKS < nblk ? 1u : 0u;
    uint32_t start = 0, end = 0;
    if (I.valid) {
        start = grp[0];
        end = grp[1];
        if (W.nsplit[0] > 1u) { // offsets are relative to the first item of the part of k7_side that wrote them
            // the part an item belongs to: the last one whose first item is not behind it; the payload of all parts in front of
            // that one lies in front of the item (parts that had nothing to decode say no first item and a length of 0)
            uint32_t ps = 0u, pe = 0u, sum_s = 0u;
            uint64_t e64 = end;
#pragma unroll
            for (uint32_t q = 0; q < 3u; q++) {
                const uint32_t pi = F->part_item[q];
                ps = g >= pi ? q + 1u : ps;
                pe = g + 1u >= pi && pi != 0xFFFFFFFFu ? q + 1u : pe;
            }
#pragma unroll
            for (uint32_t q = 0; q < 3u; q++) {
                const uint32_t pl = F->part_len[q];
                sum_s += q < ps ? pl : 0u;
                e64 += q < pe ? pl : 0u;
            }
            start += sum_s;
            end = static_cast<uint32_t>(e64);
            // the payload's end against `len` (RawData.cpp:419-420: some block would cross it), checked here because no part of
            // k7_side knows the sum; by the wave of the frame's last item
            if ((g + 1u) * ITEM_BLOCKS >= nblk && e64 > F->len)
                atomicOr(W.status + (W.nsplit[0] + W.nsplit[1]) * f, MCRAW_E_TRUNCATED);
        }
    }
    I.g = g;
    I.nblk = nblk;
    I.tilesX = F->tilesX;
    I.base16 = start & ~15u;
    I.head = start - I.base16;
    I.n16 = I.valid ? min((end - min(end, I.base16) + 15u) >> 4, PAY_CHUNKS) : 0u; // never more than ITEM_SPAN + head
    I.width = F->width;
    I.rows = static_cast<int32_t>(F->rows);
    I.fast = F->fast_store;
    I.in = F->in;
    I.len = F->len;
    I.out = F->out;
    I.meta = static_cast<size_t>(f) * W.Rmax * 64u + static_cast<size_t>(g) * ITEM_BLOCKS;
    I.lean = 0u;
    return I;
}

// Store 8 consecutive pixels (16 B) of row y starting at column x, cropped to `width`
// (RawData.cpp:598-608 copies `width` pixels of the coded row).
template <bool NT = false, int POST = 0> // POST: 0 = the plain mosaic, else bits per sample of the post stage's rows
__device__ __forceinline__ void store_px8(const ItemS &I, const Post &post, uint32_t y, uint32_t x, uint32_t p[4])
{
    const uint32_t width = static_cast<uint32_t>(I.width);
    if (y >= static_cast<uint32_t>(I.rows) || x >= width)
        return;
    if (POST) { // black levels / 12-bit strip rows (mcraw_dev.h)
        post_store8<NT, POST>(I.out, post, width, y, x, p, min(8u, width - x), I.fast != 0u, POST != 16 && I.lean != 0u);
        return;
    }
    uint16_t *dst = I.out + static_cast<size_t>(y) * static_cast<size_t>(width) + x;
    if (I.fast && x + 8u <= width) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        if (NT)
            store_stream16(dst, v);
        else
            *gptr<u32x4>(dst) = v;
    } else if (x + 8u <= width) {
        // a row that does not start on a 16-byte boundary (width % 8 != 0, or an odd output address):
        // still one 16-byte store -- global memory takes it at any 2-byte alignment -- instead of eight
        // 2-byte ones
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
        const u32x4_u v = {p[0], p[1], p[2], p[3]};
        *gptr<u32x4_u>(dst) = v;
    } else { // the cropped end of a row: element stores
        const uint32_t n = width - x;
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++)
            if (i < n)
                gptr<uint16_t>(dst)[i] = static_cast<uint16_t>(p[i >> 1] >> (16u * (i & 1u)));
    }
}

// Decode the staged item: every lane owns 8 samples (8k..8k+7) of two sibling blocks
// (2r, 2r+1) of one tile, i.e. 16 consecutive pixels of row 4ty + r + 2(k>>2).
// Before storing, lanes k and k^4 trade one 16-byte half so that each lane ends up
// with the SAME 16-byte column chunk of rows r and r+2: the 8 lanes of a (tile, r)
// then write one full 128-byte line per store instruction instead of two half-filled
// ones (partial-line writes were the bottleneck of the first version).
//
// ABL (builds with -DMCRAW_DIAG only, env MCRAW_ABLATE): 0 = product; 1 = no global stores;
// 2 = no unpack arithmetic; 3 (caller) = no payload loads.
template <int ABL = 0, bool NT = false, int POST = 0>
__device__ __forceinline__ void item_decode(const ItemS &I, const Post &post, uint32_t tt, uint32_t r, uint32_t k,
                                            const uint8_t *s_pay, const uint32_t *s_blk, const uint16_t *s_ref,
                                            const uint4 *s_tab)
{
    const uint32_t tile = I.g * ITEM_TILES + tt;
    if (!I.valid || tile * 4u >= I.nblk) // uniform over the 16 lanes of a tile
        return;
    const bool whole = __ballot(true) == ~0ull; // (10- / 14-bit strips: the whole wave is here, lanes can fetch from each other)
    const uint32_t ty = tile / I.tilesX, tx = tile - ty * I.tilesX;

    const uint32_t bi = 4u * tt + 2u * r;
    const uint2 mb = *reinterpret_cast<const uint2 *>(&s_blk[bi]);
    const uint32_t refs2 = *reinterpret_cast<const uint32_t *>(&s_ref[bi]); // refA | refB << 16
    Unpacked A, B;
    if (ABL == 2) {
        A.x[0] = A.x[1] = A.x[2] = A.x[3] = mb.x;
        B.x[0] = B.x[1] = B.x[2] = B.x[3] = mb.y;
    } else {
        A = unpack8<true>(s_pay, mb.x & 0xffffu, mb.x >> 16, k, s_tab);
        B = unpack8<true>(s_pay, mb.y & 0xffffu, mb.y >> 16, k, s_tab);
    }

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
    if (ABL == 1) { // keep the values alive without storing them
#pragma unroll
        for (int m = 0; m < 8; m++)
            asm volatile("" ::"v"(o[m]));
        return;
    }

    // lane k < 4 holds chunks (2c, 2c+1) of row r, lane k+4 the same chunks of row r+2
    // (c = k & 3).  Swap "my second chunk" against "partner's first chunk".
    const bool lo = k < 4u;
    uint32_t p0[4], p1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t send = lo ? o[4 + i] : o[i];
        // ds_swizzle bit-mask mode: lane' = ((lane & 0x1f) | 0) ^ 4  -> lane ^ 4, no LDS memory touched
        const uint32_t recv = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(send), 0x101F));
        p0[i] = lo ? o[i] : recv;
        p1[i] = lo ? recv : o[4 + i];
    }
    const uint32_t x = 64u * tx + 8u * (2u * (k & 3u) + (k >> 2));
    const uint32_t y = 4u * ty + r;
    if (POST == 14) { // (10-bit rows the same way -- 16-byte stores 10 bytes apart, six bytes written twice -- are 1.5 % SLOWER than 8 + 2)
        // the interior of the frame: one 16-byte store per lane and row (post_store8_merged).  The row's next 8 samples: lane
        // k + 4 for the first four lanes of a (tile, row pair) group, k - 3 for the next three, and the first lane of the next
        // tile's group for the last one -- when that tile is the neighbour on the right and decoded in this pass (else the
        // lane's store reaches back into the previous 8 samples instead: lane k - 4, the group's fourth).
        const bool in = I.fast != 0u && y + 2u < static_cast<uint32_t>(I.rows) && x + 8u <= static_cast<uint32_t>(I.width);
        if (whole && __ballot(!in) == 0ull) {
            const uint32_t lane = threadIdx.x & 63u;
            const bool has_next = k != 7u || ((tt & 3u) != 3u && tx + 1u < I.tilesX);
            const uint32_t src = k < 4u ? lane + 4u : k < 7u ? lane - 3u : has_next ? lane + 9u : lane;
            const uint32_t width = static_cast<uint32_t>(I.width);
            post_store8_merged<14>(I.out, post, width, y, x, p0, I.lean != 0u, src, has_next);
            post_store8_merged<14>(I.out, post, width, y + 2u, x, p1, I.lean != 0u, src, has_next);
            return;
        }
    }
    store_px8<NT, POST>(I, post, y, x, p0);
    store_px8<NT, POST>(I, post, y + 2u, x, p1);
}

// One WAVE per item, four independent waves per workgroup, no barrier on the data
// path: lane = block for the metadata (bits -> LEN -> wave scan gives every block's
// offset), the wave stages its own span in its own LDS slice, then decodes its 16
// tiles in four rounds of 64 lanes.  The only workgroup-wide event is the barrier
// that publishes the shared term table.
template <int ABL, bool NT, int POST = 0>
__global__ __launch_bounds__(256) void k7_tiles(const Work7 W, uint32_t total, uint32_t first_frame, uint32_t class_groups)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_pay[4][PAY_LDS];
    __shared__ uint4 s_tab[72];
    __shared__ uint32_t s_blk[4][ITEM_BLOCKS];
    __shared__ uint16_t s_ref[4][ITEM_BLOCKS];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = (W.xcd_chunk ? xcd_chunked(blockIdx.x, gridDim.x, W.xcd_chunk) : xcd_remap(blockIdx.x, gridDim.x)) * 4u + wave;
    if (tid < 72u)
        s_tab[tid] = reinterpret_cast<const uint4 *>(c_tab7)[tid];

    ItemS I;
    I.valid = 0;
    I.n16 = 0;
    if (item < total)
        I = item_scalars(W, item, first_frame, class_groups);

    // span -> registers: 16-byte chunks lane, lane+64, ... (ITEM_SPAN + alignment head)
    constexpr uint32_t NV = (PAY_CHUNKS + 63u) / 64u;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(I.in, I.len);
    uint4 v[NV];
#pragma unroll
    for (uint32_t c = 0; c < NV; c++) {
        v[c] = make_uint4(0, 0, 0, 0);
        if (ABL != 3 && lane + 64u * c < I.n16)
            v[c] = ld_b128_nt(rs, I.base16 + (lane + 64u * c) * 16u);
    }
    uint32_t b = 0, r = 0;
    if (I.valid && lane < ITEM_BLOCKS && I.g * ITEM_BLOCKS + lane < I.nblk) {
        b = W.bits[I.meta + lane];
        r = W.refs[I.meta + lane];
    }
    if (POST == 12 || POST == 10 || POST == 14) {
        // strip rows (round 4: 12 bits; round 5: 10 and 14 too): a block is one colour plane of its tile (block k of a tile: row parity k >> 1, column parity
        // k & 1, RawData.cpp:581-593), so its black level is ONE value and can come off its reference -- if the reference is not
        // below it --, and no sample of the block can reach 2^POST when reference - black + (2^storage width - 1) does not: then
        // neither the saturating subtraction nor the clamp of the strip stage can change a sample, and the item's stores skip
        // both (16 of the stage's ~ 54 vector instructions per lane and call).  One ballot per item decides.
        const uint32_t bl2 = (W.post.mode & POST_BLACK) ? ((lane & 2u) ? W.post.black23 : W.post.black01) : 0u;
        const uint32_t bl = (lane & 1u) ? bl2 >> 16 : bl2 & 0xffffu;
        const uint32_t sw = b <= 6u ? b : b <= 8u ? 8u : 10u; // storage width of the block's residuals (RawData.cpp:424-458)
        const bool real = I.valid && lane < ITEM_BLOCKS && I.g * ITEM_BLOCKS + lane < I.nblk;
        // (r + 2^sw - 1 <= 65535: the reference add of the decode does not wrap either, RawData.cpp:581-593)
        const bool fits = !real || (b <= 10u && r >= bl && r - bl + ((1u << sw) - 1u) <= (1u << POST) - 1u && r + ((1u << sw) - 1u) <= 65535u);
        I.lean = __ballot(!fits) == 0ull ? 1u : 0u;
        if (I.lean)
            r -= bl;
    }

    uint32_t tot;
    const uint32_t ex = wave_excl_scan(lane < ITEM_BLOCKS ? len7_of(b) : 0u, lane, &tot);
    if (lane < ITEM_BLOCKS) {
        s_blk[wave][lane] = (I.head + ex) | (cls7_of(b) << 16);
        s_ref[wave][lane] = static_cast<uint16_t>(r);
    }
    uint4 *pay4 = reinterpret_cast<uint4 *>(s_pay[wave]);
#pragma unroll
    for (uint32_t c = 0; c < NV; c++)
        if (lane + 64u * c < I.n16)
            pay4[lane + 64u * c] = v[c];
    __syncthreads(); // (with loads that write the LDS in flight the compiler waits for them here: vmcnt(0) in front of the barrier)

#pragma unroll
    for (uint32_t q = 0; q < ITEM_TILES / 4u; q++)
        item_decode<ABL, NT, POST>(I, W.post, q * 4u + (lane >> 4), (lane >> 3) & 1u, lane & 7u, s_pay[wave], s_blk[wave],
                                   s_ref[wave], s_tab);
}

// ------------------------------------------------------------------ launchers

#ifdef MCRAW_DIAG
extern "C" void mcraw_diag_side_prof(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_side_prof), sizeof(g_side_prof));
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_side_prof), z, sizeof(z));
    }
}
#endif

// The order k7_tiles takes its workgroups' work in (block b of a grid of n -> logical workgroup), exported so that the rule
// -- a permutation for every grid size and run length -- can be checked without a GPU (tests/test_abi_library.py).
extern "C" uint32_t mcraw_tile_order(uint32_t b, uint32_t n, uint32_t runs)
{
    return runs ? xcd_chunked(b, n, runs) : xcd_remap(b, n);
}

void launch_k7(const Work7 &W, uint32_t stage, hipStream_t st)
{
    const uint32_t n7 = static_cast<uint32_t>(W.n7);
    switch (stage) {
    case MCRAW_K7_SIDE: {
        const dim3 sgrid(n7 * (W.nsplit[0] + W.nsplit[1]));
        if (W.nsplit[0] + W.nsplit[1] > 2u && W.side_lastc)
            hipLaunchKernelGGL((k7_side<MCRAW_SIDE_T, MCRAW_SIDE_LPT, MCRAW_SIDE_LCAP, MCRAW_SPEC_WARM, true, true>), sgrid, dim3(MCRAW_SIDE_T), 0, st, W);
        else if (W.nsplit[0] + W.nsplit[1] > 2u)
            hipLaunchKernelGGL((k7_side<MCRAW_SIDE_T, MCRAW_SIDE_LPT, MCRAW_SIDE_LCAP, MCRAW_SPEC_WARM, true, false>), sgrid, dim3(MCRAW_SIDE_T), 0, st, W);
        else
            hipLaunchKernelGGL((k7_side<MCRAW_SIDE_T, MCRAW_SIDE_LPT, MCRAW_SIDE_LCAP, MCRAW_SPEC_WARM, false, false>), sgrid, dim3(MCRAW_SIDE_T), 0, st, W);
        break;
    }
    case MCRAW_K7_TILES: {
#ifdef MCRAW_DIAG // timing experiments of the same kernel (see item_decode); not in the product library
        static const int abl = []() {
            const char *e = std::getenv("MCRAW_ABLATE");
            return e ? std::atoi(e) : 0;
        }();
#endif
        for (uint32_t k = 0; k < W.nclasses; k++) { // one launch per size class (a homogeneous batch has one)
            const uint32_t first = W.class_first[k], count = W.class_first[k + 1] - first, groups = W.class_groups[k];
            const uint32_t total = groups * ITEM_SPLIT * count;
            if (total == 0u)
                continue;
            const dim3 grid((total + 3) / 4);
            if (W.post.mode != 0u) { // one kernel instance per row format
                switch (post_bits(W.post.mode)) {
                case 12: hipLaunchKernelGGL((k7_tiles<0, true, 12>), grid, dim3(256), 0, st, W, total, first, groups); break;
                case 10: hipLaunchKernelGGL((k7_tiles<0, true, 10>), grid, dim3(256), 0, st, W, total, first, groups); break;
                case 14: hipLaunchKernelGGL((k7_tiles<0, true, 14>), grid, dim3(256), 0, st, W, total, first, groups); break;
                default: hipLaunchKernelGGL((k7_tiles<0, true, 16>), grid, dim3(256), 0, st, W, total, first, groups); break;
                }
                continue;
            }
#ifdef MCRAW_DIAG
            if (abl == 1)
                hipLaunchKernelGGL((k7_tiles<1, true>), grid, dim3(256), 0, st, W, total, first, groups);
            else if (abl == 2)
                hipLaunchKernelGGL((k7_tiles<2, true>), grid, dim3(256), 0, st, W, total, first, groups);
            else if (abl == 3)
                hipLaunchKernelGGL((k7_tiles<3, true>), grid, dim3(256), 0, st, W, total, first, groups);
            else if (abl == 4)
                hipLaunchKernelGGL((k7_tiles<0, false>), grid, dim3(256), 0, st, W, total, first, groups);
            else
#endif
                hipLaunchKernelGGL((k7_tiles<0, true>), grid, dim3(256), 0, st, W, total, first, groups);
        }
        break;
    }
    }
}

} // namespace mcraw
