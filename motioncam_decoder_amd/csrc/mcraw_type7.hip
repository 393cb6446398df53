// mcraw_type7.hip -- gfx950 kernels for the current MCRAW frame encoding
// (compressionType 7).  Replaces motioncam::raw::Decode, lib/RawData.cpp:528-612.
//
//   k7_hdr, k7_maps, k7_follow   header checks + side-stream chain resolve
//                                (RawData.cpp:500-524, 547-554; :463-498 the inline-header chain)
//   k7_records (sparse / dense)  side-stream record decode (RawData.cpp:485-495)
//                                -> bits[], refs[], byte length of every decode item
//   k7_scan                      payload offsets (RawData.cpp:562, 576-579: offset += LEN[bits])
//   k7_tiles                     tile unpack + reference add + Bayer interleave + crop
//                                (RawData.cpp:410-461, 112-408, 581-593, 598-608)  <- the roofline kernel
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

// ------------------------------------------------------------------ side-stream chain
//
// A side stream is a chain of records {hbits<<4 | ref>>8, ref & 255, LEN[hbits] payload
// bytes}: where record i+1 starts is only known from the header of record i
// (RawData.cpp:485-495).  A lone wave chasing ~2000 headers per stream costs ~0.3 ms,
// so the chain is resolved in parallel with TRANSITION MAPS: record strides are even and
// at most 130 bytes, hence a fixed 1 KiB chunk of the stream can only be entered at 65
// offsets ("phases" 0,2,..,128 past the chunk start).
//
//   k7_hdr      per stream: validate the frame header (RawData.cpp:547-554), find where the
//               stream starts and how many chunks it can span; build the work list of k7_maps
//   k7_maps     per chunk: table of record strides, then per phase a walk over it to the chunk end
//               -> (exit phase into the next chunk, records started)
//   k7_follow   per stream: follow the true phase through the chunk maps
//               -> (entry phase, first record index) per chunk; work list of k7_records
//   k7_records  per 4 chunks (a lane walks each from its true entry), or per chunk of tiny records
//               (pointer doubling): list the records, unpack eight of them per pass
//               -> bits[], refs[], byte length of every decode item
//
// k7_maps and k7_records are persistent grids looping over device-built work lists, so no
// workgroup is spent on chunks that lie beyond a stream.
constexpr uint32_t DEAD7 = 127u; // phase value: the chain has ended (record past `len`, or all records found)

constexpr uint32_t MAPS_CH = 3; // chunks per 256-thread workgroup: 3 x 65 phases = 195 lanes
constexpr uint32_t NODEAD = 0xFFFFu;
constexpr uint32_t DENSE_RECORDS = 96; // k7_records lists chunks with more records by pointer doubling
constexpr uint32_t REC_GROUP = 4;      // consecutive sparse chunks per k7_records work item (one walking lane each)

// Where the record whose header sits at staged offset `rel` ends (= where the next one
// starts), or NODEAD when it would cross `len` (RawData.cpp:419-420 skips such a block).
__device__ __forceinline__ uint32_t next_of(const uint8_t *s_b, uint32_t head, uint32_t rel, uint32_t abs, uint32_t len)
{
    const uint32_t nx = rel + 2u + len7_of(static_cast<uint32_t>(s_b[head + rel]) >> 4);
    return abs + nx > len ? NODEAD : nx;
}

// One wave per side stream (index fs = 2 * frame + s; s = 0 bits, 1 refs).
__global__ __launch_bounds__(64) void k7_hdr(const Work7 W)
{
    const uint32_t fs = blockIdx.x, lane = threadIdx.x;
    const uint32_t f = fs >> 1, s = fs & 1u;
    const Plan7 *P = W.plans + f;
    const uint32_t len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    const uint4 hv = ld_b128(rs, 0); // frame header: 4 x u32 LE (RawData.cpp:500-524); zeros when len < 16
    const uint32_t encW = __builtin_amdgcn_readfirstlane(hv.x), encH = __builtin_amdgcn_readfirstlane(hv.y);
    const uint32_t bitsOff = __builtin_amdgcn_readfirstlane(hv.z), refsOff = __builtin_amdgcn_readfirstlane(hv.w);
    const uint32_t so = s ? refsOff : bitsOff;
    int32_t err = 0;
    bool frame_ok = false;
    if (len < 16u || bitsOff > len || refsOff > len || (encW & 63u) != 0u || encW < static_cast<uint32_t>(P->width) ||
        encW == 0u || encH == 0u || (encH & 3u) != 0u)
        err = MCRAW_E_HEADER; // RawData.cpp:547-554 returns 0
    else if (encW != P->encW || encH != P->encH)
        err = E_GEOMETRY;
    else {
        frame_ok = true;
        if (so + 4u > len || so + 4u < so)
            err = MCRAW_E_TRUNCATED;
        else {
            const uint32_t count = ld_u8(rs, so) | (ld_u8(rs, so + 1u) << 8) | (ld_u8(rs, so + 2u) << 16) |
                                   (ld_u8(rs, so + 3u) << 24);
            if (__builtin_amdgcn_readfirstlane(count) < P->nblk) // the reference would index past the vector (:573-574)
                err = MCRAW_E_SIDESTREAM;
        }
    }
    if (err && lane == 0)
        atomicOr(W.status + f, err);
    // Extent: the bits stream of a canonically laid out frame ends where the refs stream
    // begins; k7_follow reports E_LAYOUT if its chain is still alive there and the host
    // re-plans that frame with the hint off (Plan7::full_extent).
    const uint32_t s0 = so + 4u;
    uint32_t nchunk = 0, hinted = 0;
    if (frame_ok && !err) {
        uint32_t end = len;
        if (s == 0u && !P->full_extent && refsOff > bitsOff) {
            end = refsOff;
            hinted = 1u;
        }
        if (end > s0)
            nchunk = min(W.nch, (end - s0 + CH7 - 1u) / CH7);
    }
    const uint32_t nwg = (nchunk + MAPS_CH - 1u) / MAPS_CH;
    uint32_t base = 0;
    if (lane == 0) {
        W.sinfo[fs] = make_uint4(s0, nchunk, hinted, 0u);
        if (nwg)
            base = atomicAdd(W.counters, nwg); // the order of the work list does not matter
    }
    base = __builtin_amdgcn_readfirstlane(base);
    // everything k7_maps needs to start loading: stream, first chunk, its byte offset, chunk count
    for (uint32_t i = lane; i < nwg; i += 64u)
        W.list_maps[base + i] = make_uint4(fs, i * MAPS_CH, s0 + i * MAPS_CH * CH7, nchunk);
}

__global__ __launch_bounds__(256) void k7_maps(const Work7 W)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_b[MAPS_CH * CH7 + 32];
    __shared__ uint16_t s_nxt[MAPS_CH * CH7 / 2];
    constexpr uint32_t HALF7 = CH7 / 2;
    __shared__ __attribute__((aligned(16))) uint8_t s_stride[MAPS_CH * HALF7 + 16 + 80]; // + head, + the last stride's reach

    const uint32_t tid = threadIdx.x;
    const uint32_t sub = tid / PH7, ph = tid - sub * PH7;
    const uint32_t nwork = W.counters[0];
    uint4 nextw = make_uint4(0, 0, 0, 0);
    if (blockIdx.x < nwork)
        nextw = W.list_maps[blockIdx.x];
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint4 e = nextw;
        if (wi + gridDim.x < nwork) // descriptor of the next work item rides behind this one
            nextw = W.list_maps[wi + gridDim.x];
        const uint32_t fs = __builtin_amdgcn_readfirstlane(e.x), c0 = __builtin_amdgcn_readfirstlane(e.y);
        const uint32_t abs0 = __builtin_amdgcn_readfirstlane(e.z), nchunk = __builtin_amdgcn_readfirstlane(e.w);
        const Plan7 *P = W.plans + (fs >> 1);
        const uint32_t len = P->len;
        const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
        const uint32_t base16 = abs0 & ~15u, head = abs0 - base16;
        const uint32_t n16 = (head + MAPS_CH * CH7 + 15u) >> 4; // <= 194 lines: one per thread
        const uint32_t c = c0 + sub;
        if (len >= MAPS_CH * CH7 + 130u && abs0 <= len - (MAPS_CH * CH7 + 130u)) {
            // No record that starts in these chunks can reach `len`.  Headers can sit at every
            // second byte from `head` on; each thread turns the 8 candidates of one 16-byte line
            // into record strides in half positions, 1 + LEN/2 (RawData.cpp:27-45), so that the 65
            // walks of a chunk are chains of single-byte LDS reads
            if (tid < n16) {
                const uint4 v = ld_b128(rs, base16 + tid * 16u);
                const uint32_t pick = (head & 1u) ? 0x07050301u : 0x06040200u;
                uint32_t st[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const uint32_t h4 = i ? __builtin_amdgcn_perm(v.w, v.z, pick) : __builtin_amdgcn_perm(v.y, v.x, pick);
                    const uint32_t hb = (h4 >> 4) & 0x0F0F0F0Fu;
                    const uint32_t sel = hb & 0x07070707u;
                    const uint32_t lo = __builtin_amdgcn_perm(0x08060504u, 0x03020100u, sel); // LEN/8, bits 0..7
                    const uint32_t hi = __builtin_amdgcn_perm(0x10101010u, 0x100A0A08u, sel); // LEN/8, bits 8..15
                    const uint32_t g = (hb >> 3) & 0x01010101u;
                    const uint32_t m = (g << 8) - g;
                    st[i] = (((hi & m) | (lo & ~m)) << 2) + 0x01010101u;
                }
                reinterpret_cast<uint2 *>(s_stride)[tid] = make_uint2(st[0], st[1]);
            }
            __syncthreads();
            if (sub < MAPS_CH && c < nchunk) {
                // table index u <-> staged byte 2u + (head & 1); chunk `sub` spans HALF7 indices from here
                const uint8_t *pp = s_stride + (head >> 1) + sub * HALF7 + ph;
                const uint8_t *const pe = s_stride + (head >> 1) + (sub + 1u) * HALF7;
                uint32_t count = 0;
                while (pp < pe) {
                    pp += *pp;
                    count++;
                }
                W.cmap[(static_cast<size_t>(fs) * W.nch + c) * PH7 + ph] = static_cast<uint32_t>(pp - pe) | (count << 8);
            }
        } else {
            // the stream's last chunks: successor table with the `len` check (a record that would
            // cross `len` ends the chain, RawData.cpp:419-420)
            for (uint32_t q = tid; q < n16; q += 256u)
                reinterpret_cast<uint4 *>(s_b)[q] = ld_b128(rs, base16 + q * 16u);
            __syncthreads();
            for (uint32_t i = tid; i < MAPS_CH * CH7 / 2; i += 256u)
                s_nxt[i] = static_cast<uint16_t>(next_of(s_b, head, 2u * i, abs0, len));
            __syncthreads();
            if (sub < MAPS_CH && c < nchunk) {
                const uint32_t hi = (sub + 1u) * CH7; // end of this chunk inside the staged bytes
                uint32_t rel = sub * CH7 + 2u * ph, count = 0;
                while (rel < hi) { // NODEAD ends the loop too
                    rel = s_nxt[rel >> 1];
                    count += rel != NODEAD ? 1u : 0u;
                }
                W.cmap[(static_cast<size_t>(fs) * W.nch + c) * PH7 + ph] =
                    (rel == NODEAD ? DEAD7 : (rel - hi) >> 1) | (count << 8);
            }
        }
        __syncthreads(); // the staging buffers are reused by the next work item
    }
}

constexpr uint32_t FOLLOW_PIECE = 192; // chunk maps staged per pass (192 * 65 * 4 B = 49 KB)

__global__ __launch_bounds__(256) void k7_follow(const Work7 W)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_map[FOLLOW_PIECE * PH7];
    __shared__ uint32_t s_entry[FOLLOW_PIECE];
    __shared__ uint32_t s_state[4];

    const uint32_t fs = blockIdx.x, f = fs >> 1;
    int32_t *status = W.status + f;
    if (*status != 0)
        return;
    const Plan7 *P = W.plans + f;
    const uint32_t tid = threadIdx.x;
    const uint4 si = W.sinfo[fs];
    const uint32_t nchunk = si.y, hinted = si.z;
    const uint32_t R = P->ngroups, nch = W.nch;
    const uint32_t *maps = W.cmap + static_cast<size_t>(fs) * nch * PH7;
    uint32_t *centry = W.centry + static_cast<size_t>(fs) * nch;
    uint32_t p = 0, n = 0, creal = 0; // the first record sits right behind the entry count
    for (uint32_t base = 0; base < nchunk && p != DEAD7; base += FOLLOW_PIECE) {
        const uint32_t cnt = min(FOLLOW_PIECE, nchunk - base);
        const uint32_t words = cnt * PH7;
        const uint32_t *src = maps + static_cast<size_t>(base) * PH7; // 16-byte aligned (base % 4 == 0)
        for (uint32_t i = tid * 4u; i < words; i += 1024u) {
            if (i + 4u <= words) {
                *reinterpret_cast<uint4 *>(&s_map[i]) = *reinterpret_cast<const uint4 *>(&src[i]);
            } else {
                for (uint32_t t = i; t < words; t++)
                    s_map[t] = src[t];
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t c = 0;
            for (; c < cnt && p != DEAD7; c++) {
                s_entry[c] = p | (n << 8);
                const uint32_t m = s_map[c * PH7 + p];
                n += m >> 8;
                p = n >= R ? DEAD7 : (m & 255u); // all records found: nothing beyond this chunk
            }
            s_state[0] = p;
            s_state[1] = n;
            s_state[2] = c; // chunks of this piece that hold records
        }
        __syncthreads();
        const uint32_t live = s_state[2];
        for (uint32_t i = tid; i < live; i += 256u)
            centry[base + i] = s_entry[i];
        p = s_state[0];
        n = s_state[1];
        creal = base + live;
        __syncthreads();
    }
    if (n < R) {
        // the chain stopped short of R records: a record crosses `len` -- or the bits stream
        // runs past the start of the refs stream (non-canonical layout: host re-plans)
        if (tid == 0)
            atomicOr(status, (p != DEAD7 && hinted) ? E_LAYOUT : MCRAW_E_TRUNCATED);
        return;
    }
    // Work lists of k7_records: the chunks [0, creal) of this stream, REC_GROUP consecutive chunks
    // per item (one lane walks each).  A chunk of tiny records (more than DENSE_RECORDS of them) is
    // listed by pointer doubling in a kernel of its own: it goes to the dense list, which grows
    // from the back of the same array, and the rest of its group is listed chunk by chunk.
    if (tid < 4u)
        s_state[tid] = 0; // [0] dense items of this stream, [1] sparse slots handed out, [2] dense slots handed out, [3] sparse items
    __syncthreads();
    const uint32_t ngrp = (creal + REC_GROUP - 1u) / REC_GROUP;
    auto first_record = [&](uint32_t i) { return i < creal ? min(R, centry[i] >> 8) : R; };
    for (uint32_t g = tid; g < ngrp; g += 256u) {
        const uint32_t c0 = g * REC_GROUP, cnt = min(REC_GROUP, creal - c0);
        uint32_t nd = 0;
        for (uint32_t j = 0; j < cnt; j++)
            nd += first_record(c0 + j + 1u) - min(first_record(c0 + j + 1u), centry[c0 + j] >> 8) > DENSE_RECORDS ? 1u : 0u;
        if (nd) {
            atomicAdd(&s_state[0], nd);
            atomicAdd(&s_state[3], cnt - nd);
        } else {
            atomicAdd(&s_state[3], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t nd = s_state[0];
        s_entry[0] = atomicAdd(W.counters + 1, s_state[3]);
        s_entry[1] = nd ? atomicAdd(W.counters + 2, nd) : 0u;
    }
    __syncthreads();
    const uint32_t baseS = s_entry[0], baseD = s_entry[1];
    for (uint32_t g = tid; g < ngrp; g += 256u) {
        const uint32_t c0 = g * REC_GROUP, cnt = min(REC_GROUP, creal - c0);
        uint32_t dmask = 0;
        for (uint32_t j = 0; j < cnt; j++)
            dmask |= (first_record(c0 + j + 1u) - min(first_record(c0 + j + 1u), centry[c0 + j] >> 8) > DENSE_RECORDS ? 1u : 0u) << j;
        if (dmask == 0u) {
            // sparse item: stream, first chunk | chunks << 24, byte offset of the first chunk, end of its record range
            W.list_recs[baseS + atomicAdd(&s_state[1], 1u)] =
                make_uint4(fs, c0 | (cnt << 24), si.x + c0 * CH7, first_record(c0 + cnt));
            continue;
        }
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t i = c0 + j, e = centry[i];
            if ((dmask >> j) & 1u) { // dense item: stream, records in the chunk, its byte offset, entry (phase | first record << 8)
                const uint32_t nrec = first_record(i + 1u) - min(first_record(i + 1u), e >> 8);
                W.list_recs[W.list_cap - 1u - (baseD + atomicAdd(&s_state[2], 1u))] = make_uint4(fs, nrec, si.x + i * CH7, e);
            } else {
                W.list_recs[baseS + atomicAdd(&s_state[1], 1u)] =
                    make_uint4(fs, i | (1u << 24), si.x + i * CH7, first_record(i + 1u));
            }
        }
    }
}

// staged bytes of an item of `chunks` chunks: records starting in the last one may run 130 bytes
// past it (+ read slack), and the first chunk starts up to 15 bytes into the first 16-byte line
constexpr int rec_bytes(int chunks) { return (chunks * CH7 + 130 + 8 + 16 + 15) / 16 * 16 + 16; }
constexpr int REC_MAX = CH7 / 2; // records that can start in one chunk

template <int PATTERN>
__device__ __forceinline__ uint32_t swz_xor(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), PATTERN));
}

// One wave per work item.  Sparse form: REC_GROUP consecutive chunks, lane j walks chunk j from
// its resolved entry and notes payload offset, class and reference of every record it meets (the
// record indices of the item are contiguous, so the notes form one flat list).  Dense form: one
// chunk of tiny records, listed by pointer doubling.  Then EIGHT records are unpacked per pass:
// lane = (record, k) owns samples 8k..8k+7 exactly like a payload lane (DecodeBlock on the
// record, RawData.cpp:489; + reference, :491-492).
template <int ABL, bool DENSE>
__global__ __launch_bounds__(64) void k7_records(const Work7 W)
{
    constexpr int NCHUNK = DENSE ? 1 : static_cast<int>(REC_GROUP);
    constexpr int REC_BYTES = rec_bytes(NCHUNK);
    constexpr uint32_t HDR_CAP = DENSE ? REC_MAX : REC_GROUP * DENSE_RECORDS;
    __shared__ __attribute__((aligned(16))) uint8_t s_b[REC_BYTES];
    __shared__ uint32_t s_hdr[HDR_CAP + 8]; // per record: payload offset | hbits << 14 | reference << 18
    __shared__ __attribute__((aligned(16))) uint16_t s_J[DENSE ? CH7 / 2 + 8 : 8]; // successor table of the chunk's candidates
    __shared__ __attribute__((aligned(8))) uint8_t s_M[DENSE ? CH7 / 2 + 8 : 8];   // chain marks
    __shared__ uint4 s_tab[72];

    const uint32_t lane = threadIdx.x;
    s_tab[lane] = reinterpret_cast<const uint4 *>(c_tab7)[lane];
    if (lane < 8u)
        s_tab[64u + lane] = reinterpret_cast<const uint4 *>(c_tab7)[64u + lane];
    const uint32_t nwork = W.counters[DENSE ? 2 : 1];
    // sparse items are listed from the front of the work list, dense ones from its back
    const uint4 *list = DENSE ? W.list_recs + (W.list_cap - 1u) : W.list_recs;
    constexpr int STEP = DENSE ? -1 : 1;
    uint4 nextw = make_uint4(0, 0, 0, 0);
    if (blockIdx.x < nwork)
        nextw = list[STEP * static_cast<int>(blockIdx.x)];
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
    const uint4 we = nextw;
    if (wi + gridDim.x < nwork) // descriptor of the next work item rides behind this one
        nextw = list[STEP * static_cast<int>(wi + gridDim.x)];
    const uint32_t fs = __builtin_amdgcn_readfirstlane(we.x);
    const uint32_t f = fs >> 1, s = fs & 1u;
    int32_t *status = W.status + f;
    const Plan7 *P = W.plans + f;
    const uint32_t R = P->ngroups, nblk = P->nblk;
    const uint32_t len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    const uint32_t abs = __builtin_amdgcn_readfirstlane(we.z);
    const uint32_t base16 = abs & ~15u, head = abs - base16;
    // sparse: entries of my chunks, one per lane (issued before the staging loads)
    const uint32_t first = __builtin_amdgcn_readfirstlane(we.y) & 0xffffffu;
    const uint32_t nchunk = DENSE ? 1u : __builtin_amdgcn_readfirstlane(we.y) >> 24;
    uint32_t ce = 0;
    if (!DENSE && lane < nchunk)
        ce = W.centry[static_cast<size_t>(fs) * W.nch + first + lane];
    __syncthreads(); // previous work item is done with the staging buffers
    const uint32_t n16 = (head + nchunk * CH7 + 130u + 8u + 16u + 15u) >> 4; // <= REC_BYTES / 16
#pragma unroll
    for (uint32_t q = 0; q < (REC_BYTES / 16 + 63) / 64; q++)
        if (lane + 64u * q < n16)
            reinterpret_cast<uint4 *>(s_b)[lane + 64u * q] = ld_b128(rs, base16 + (lane + 64u * q) * 16u);
    __syncthreads();

    uint32_t n = 0, i0;
    if (!DENSE) {
        // Sparse item: lane j follows chunk j's chain from its true entry up to the first record of
        // chunk j+1 (the item's end for the last lane) and parses each header (RawData.cpp:106-110).
        i0 = __builtin_amdgcn_readfirstlane(ce) >> 8;
        const uint32_t iend = min(R, __builtin_amdgcn_readfirstlane(we.w));
        n = iend - min(iend, i0);
        const uint32_t cnext = __shfl_down(ce, 1, 64);
        if (ABL != 3 && lane < nchunk) {
            const uint32_t myend = lane + 1u < nchunk ? min(iend, cnext >> 8) : iend;
            uint32_t rel = 2u * (ce & 255u), idx = ce >> 8;
            const uint32_t cb = lane * CH7; // my chunk inside the staged bytes
            while (rel < CH7 && idx < myend) {
                const uint32_t ro = head + cb + rel;
                const uint32_t b0 = s_b[ro], b1 = s_b[ro + 1u];
                const uint32_t nx = rel + 2u + len7_of(b0 >> 4);
                if (abs + cb + nx > len)
                    break; // cannot happen for a chunk k7_follow listed
                s_hdr[idx - i0] = (ro + 2u) | ((b0 >> 4) << 14) | ((((b0 & 15u) << 8) | b1) << 18);
                rel = nx;
                idx++;
            }
        }
        if (ABL == 2 || ABL == 3)
            n = 0;
    } else {
    const uint32_t entry = __builtin_amdgcn_readfirstlane(we.w);
    const uint32_t ph = entry & 255u;
    i0 = entry >> 8;
    // Dense chunk (runs of 2-byte records: up to 512 per KiB).  Which of the chunk's 512 even offsets start a record of the true chain?  A serial walk
    // costs ~45 scalar instructions per record on one lane; instead the chain is marked by
    // POINTER DOUBLING over all candidates at once: J1[p] = where the record at p ends;
    // level l marks J_{2^l}(q) for every marked q (marks then cover distance < 2^(l+1) from the
    // entry) and squares the table; ~6 levels for a typical chunk, 10 at most.
    constexpr uint32_t NP = CH7 / 2; // candidates; index NP = "beyond this chunk"
    const uint32_t p0 = lane * 8u;   // this lane owns candidates p0 .. p0+7
    uint32_t jn[8];
#pragma unroll
    for (uint32_t t = 0; t < 8u; t++) {
        const uint32_t rel = 2u * (p0 + t);
        const uint32_t nx = rel + 2u + len7_of(static_cast<uint32_t>(s_b[head + rel]) >> 4);
        jn[t] = (abs + nx > len || nx >= CH7) ? NP : nx >> 1; // a record crossing `len` ends the chain
    }
    *reinterpret_cast<uint4 *>(&s_J[p0]) = make_uint4(jn[0] | (jn[1] << 16), jn[2] | (jn[3] << 16), jn[4] | (jn[5] << 16), jn[6] | (jn[7] << 16));
    *reinterpret_cast<uint2 *>(&s_M[p0]) = make_uint2(0u, 0u);
    if (lane == 0)
        s_J[NP] = static_cast<uint16_t>(NP);
    __syncthreads();
    if (lane == 0 && ABL != 3)
        s_M[ph] = 1; // entry phase = candidate index (offset 2 * ph)
    __syncthreads();
    uint2 m = *reinterpret_cast<const uint2 *>(&s_M[p0]);
    for (uint32_t level = 0; level < 10u; level++) {
#pragma unroll
        for (uint32_t t = 0; t < 8u; t++) {
            const uint32_t mk = ((t < 4u ? m.x : m.y) >> (8u * (t & 3u))) & 1u;
            s_M[mk ? jn[t] : NP + 1u] = 1; // branch-free: unmarked candidates hit a dummy slot (NP, NP+1 are never read as marks)
        }
        __syncthreads();
        const uint2 m2 = *reinterpret_cast<const uint2 *>(&s_M[p0]);
        const bool grew = __any((m2.x != m.x) || (m2.y != m.y));
        m = m2;
        if (!grew)
            break;
        uint32_t sq[8];
#pragma unroll
        for (uint32_t t = 0; t < 8u; t++)
            sq[t] = s_J[jn[t]];
        __syncthreads();
#pragma unroll
        for (uint32_t t = 0; t < 8u; t++)
            jn[t] = sq[t];
        *reinterpret_cast<uint4 *>(&s_J[p0]) = make_uint4(jn[0] | (jn[1] << 16), jn[2] | (jn[3] << 16), jn[4] | (jn[5] << 16), jn[6] | (jn[7] << 16));
        __syncthreads();
    }
    // rank of every marked candidate = its record index inside the chunk (candidates are in stream order)
    const uint32_t mine = static_cast<uint32_t>(__popc(m.x) + __popc(m.y));
    uint32_t ntot;
    const uint32_t rank0 = wave_excl_scan(mine, lane, &ntot);
    n = min(ntot, R - i0);
    if (ABL == 2 || ABL == 3)
        n = 0;
    // payload offset, class and reference of each record (RawData.cpp:106-110)
    uint32_t rk = rank0;
#pragma unroll
    for (uint32_t t = 0; t < 8u; t++) {
        const uint32_t mk = ((t < 4u ? m.x : m.y) >> (8u * (t & 3u))) & 1u;
        if (mk) {
            if (rk < n) {
                const uint32_t ro = head + 2u * (p0 + t);
                const uint32_t b0 = s_b[ro], b1 = s_b[ro + 1u];
                s_hdr[rk] = (ro + 2u) | ((b0 >> 4) << 14) | ((((b0 & 15u) << 8) | b1) << 18);
            }
            rk++;
        }
    }
    }
    if (lane < 8u)
        s_hdr[n + lane] = 0u; // padding: idle lanes of the last pass unpack "class 0"
    __syncthreads();

    const uint32_t k = lane & 7u, sub = lane >> 3;
    uint8_t *bits = W.bits + static_cast<size_t>(f) * W.Rmax * 64u;
    uint16_t *refs = W.refs + static_cast<size_t>(f) * W.Rmax * 64u;
    uint32_t *glen = W.grp_off + static_cast<size_t>(f) * (W.Rmax * ITEM_SPLIT + 1u);
    for (uint32_t qb = 0; qb < (ABL == 1 ? 0u : n); qb += 8u) {
        const uint32_t q = qb + sub;
        const bool live = q < n;
        const uint32_t h = s_hdr[q];
        const uint32_t hb = (h >> 14) & 15u, ref = h >> 18;
        Unpacked U = unpack8<false>(s_b, h & 0x3fffu, cls7_of(hb), k, s_tab);
        const u16x2 rr = __builtin_bit_cast(u16x2, ref | (ref << 16));
#pragma unroll
        for (int i = 0; i < 4; i++) // uint16 wrap (RawData.cpp:492)
            U.x[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, U.x[i]) + rr);
        const uint32_t r = i0 + q;
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
            atomicOr(status, MCRAW_E_SIDESTREAM);
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
            const uint32_t m = (g << 8) - g;
            const uint32_t l4 = ((hi & m) | (lo & ~m)) | (v4 & 0x10101010u);
            l8 = __builtin_amdgcn_sad_u8(l4, 0u, l8);
        }
        // sum over the lanes of one decode item (ds_swizzle bit-mask mode: lane ^ 1, ^ 2, ^ 4)
        l8 += swz_xor<0x041F>(l8);
        l8 += swz_xor<0x081F>(l8);
        if (ITEM_SPLIT == 1)
            l8 += swz_xor<0x101F>(l8);
        if (live && (k & (8u / ITEM_SPLIT - 1u)) == 0u)
            glen[r * ITEM_SPLIT + k / (8u / ITEM_SPLIT)] = l8 << 3; // lengths until k7_scan turns them into offsets
    }
    } // work items
}

// ------------------------------------------------------------------ k7_scan
//
// Payload offset of every group: 16 + sum of the lengths before it
// (RawData.cpp:562 `offset = METADATA_OFFSET`, :576-579 `offset += ...`).
__global__ __launch_bounds__(1024) void k7_scan(const Work7 W)
{
    __shared__ uint32_t s_w[16];
    const Plan7 *P = W.plans + blockIdx.x;
    int32_t *status = W.status + blockIdx.x;
    if (*status != 0)
        return;
    const uint32_t R = P->ngroups * ITEM_SPLIT, tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t *__restrict__ g = W.grp_off + static_cast<size_t>(blockIdx.x) * (W.Rmax * ITEM_SPLIT + 1u);
    uint32_t carry = 16u;
    for (uint32_t base = 0; base < R; base += 1024u) {
        const uint32_t i = base + tid;
        const uint32_t v = i < R ? g[i] : 0u;
        uint32_t wtot;
        const uint32_t ex = wave_excl_scan(v, lane, &wtot);
        if (lane == 63u)
            s_w[w] = wtot;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t q = 0; q < 16u; q++) {
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
};

__device__ __forceinline__ ItemS item_scalars(const Work7 &W, uint32_t item, uint32_t first_frame, uint32_t class_groups)
{
    ItemS I;
    const uint32_t per = W.Rmax * ITEM_SPLIT;         // items of the uniform-stride workspace per frame
    const uint32_t cper = class_groups * ITEM_SPLIT;  // items this launch spends on each frame of its size class
    const uint32_t fc = item / cper;
    const uint32_t f = first_frame + fc;
    const uint32_t g = item - fc * cper;
    const Plan7 *P = W.plans + f;
    const uint32_t *grp = W.grp_off + static_cast<size_t>(f) * (per + 1u) + g;
    const uint32_t start = grp[0], end = grp[1];
    I.valid = (g * ITEM_BLOCKS < P->nblk && W.status[f] == 0) ? 1u : 0u;
    I.g = g;
    I.nblk = P->nblk;
    I.tilesX = P->tilesX;
    I.base16 = start & ~15u;
    I.head = start - I.base16;
    I.n16 = I.valid ? min((end - I.base16 + 15u) >> 4, PAY_CHUNKS) : 0u; // never more than ITEM_SPAN + head
    I.width = P->width;
    I.rows = P->rows;
    I.fast = P->fast_store;
    I.in = P->in;
    I.len = P->len;
    I.out = P->out;
    I.meta = static_cast<size_t>(f) * W.Rmax * 64u + static_cast<size_t>(g) * ITEM_BLOCKS;
    return I;
}

// Store 8 consecutive pixels (16 B) of row y starting at column x, cropped to `width`
// (RawData.cpp:598-608 copies `width` pixels of the coded row).
template <bool NT = false, bool POST = false>
__device__ __forceinline__ void store_px8(const ItemS &I, const Post &post, uint32_t y, uint32_t x, uint32_t p[4])
{
    const uint32_t width = static_cast<uint32_t>(I.width);
    if (y >= static_cast<uint32_t>(I.rows) || x >= width)
        return;
    if (POST) { // black levels / 12-bit strip rows (mcraw_dev.h)
        post_store8<NT>(I.out, post, width, y, x, p, min(8u, width - x), I.fast != 0u);
        return;
    }
    uint16_t *dst = I.out + static_cast<size_t>(y) * static_cast<size_t>(width) + x;
    if (I.fast && x + 8u <= width) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        if (NT)
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst));
        else
            *reinterpret_cast<u32x4 *>(dst) = v;
    } else if (x + 8u <= width) {
        // a row that does not start on a 16-byte boundary (width % 8 != 0, or an odd output address):
        // still one 16-byte store -- global memory takes it at any 2-byte alignment -- instead of eight
        // 2-byte ones
        typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
        const u32x4_u v = {p[0], p[1], p[2], p[3]};
        *reinterpret_cast<u32x4_u *>(dst) = v;
    } else { // the cropped end of a row: element stores
        const uint32_t n = width - x;
#pragma unroll
        for (uint32_t i = 0; i < 8u; i++)
            if (i < n)
                dst[i] = static_cast<uint16_t>(p[i >> 1] >> (16u * (i & 1u)));
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
template <int ABL = 0, bool NT = false, bool POST = false>
__device__ __forceinline__ void item_decode(const ItemS &I, const Post &post, uint32_t tt, uint32_t r, uint32_t k,
                                            const uint8_t *s_pay, const uint32_t *s_blk, const uint16_t *s_ref,
                                            const uint4 *s_tab)
{
    const uint32_t tile = I.g * ITEM_TILES + tt;
    if (!I.valid || tile * 4u >= I.nblk) // uniform over the 16 lanes of a tile
        return;
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
    store_px8<NT, POST>(I, post, y, x, p0);
    store_px8<NT, POST>(I, post, y + 2u, x, p1);
}

// One WAVE per item, four independent waves per workgroup, no barrier on the data
// path: lane = block for the metadata (bits -> LEN -> wave scan gives every block's
// offset), the wave stages its own span in its own LDS slice, then decodes its 16
// tiles in four rounds of 64 lanes.  The only workgroup-wide event is the barrier
// that publishes the shared term table.
template <int ABL, bool NT, bool POST = false>
__global__ __launch_bounds__(256) void k7_tiles(const Work7 W, uint32_t total, uint32_t first_frame, uint32_t class_groups)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_pay[4][PAY_LDS];
    __shared__ uint4 s_tab[72];
    __shared__ uint32_t s_blk[4][ITEM_BLOCKS];
    __shared__ uint16_t s_ref[4][ITEM_BLOCKS];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t item = xcd_remap(blockIdx.x, gridDim.x) * 4u + wave;
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
    __syncthreads();

#pragma unroll
    for (uint32_t q = 0; q < ITEM_TILES / 4u; q++)
        item_decode<ABL, NT, POST>(I, W.post, q * 4u + (lane >> 4), (lane >> 3) & 1u, lane & 7u, s_pay[wave], s_blk[wave],
                                   s_ref[wave], s_tab);
}

// ------------------------------------------------------------------ launchers

// Grid of a persistent kernel: as many workgroups as the device keeps resident.
static uint32_t persistent_grid(int which)
{
    static uint32_t g[2] = {0, 0};
    if (!g[which]) {
        int dev = 0, cus = 256, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            cus = p.multiProcessorCount;
        hipError_t e = which == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k7_maps, 256, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k7_records<0, false>, 64, 0);
        if (e != hipSuccess || per_cu <= 0)
            per_cu = which == 0 ? 8 : 16;
        g[which] = static_cast<uint32_t>(cus * per_cu);
    }
    return g[which];
}

void launch_k7(const Work7 &W, uint32_t stage, hipStream_t st)
{
    const uint32_t n7 = static_cast<uint32_t>(W.n7);
    switch (stage) {
    case MCRAW_K7_WALK:
        hipLaunchKernelGGL(k7_hdr, dim3(2 * n7), dim3(64), 0, st, W);
        hipLaunchKernelGGL(k7_maps, dim3(persistent_grid(0)), dim3(256), 0, st, W);
        hipLaunchKernelGGL(k7_follow, dim3(2 * n7), dim3(256), 0, st, W);
        break;
    case MCRAW_K7_META: {
        const dim3 g(persistent_grid(1));
#ifdef MCRAW_DIAG // timing experiments (tools/abl7.sh builds with -DMCRAW_DIAG); not in the product library
        static const int abl = []() {
            const char *e = std::getenv("MCRAW_ABLATE_REC");
            return e ? std::atoi(e) : 0;
        }();
        if (abl == 1)
            hipLaunchKernelGGL((k7_records<1, false>), g, dim3(64), 0, st, W);
        else if (abl == 2)
            hipLaunchKernelGGL((k7_records<2, false>), g, dim3(64), 0, st, W);
        else if (abl == 3)
            hipLaunchKernelGGL((k7_records<3, false>), g, dim3(64), 0, st, W);
        else
#endif
            hipLaunchKernelGGL((k7_records<0, false>), g, dim3(64), 0, st, W);
        // chunks made of runs of tiny records (flat image regions): usually none
        hipLaunchKernelGGL((k7_records<0, true>), dim3(2048), dim3(64), 0, st, W);
        break;
    }
    case MCRAW_K7_SCAN:
        hipLaunchKernelGGL(k7_scan, dim3(n7), dim3(1024), 0, st, W);
        break;
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
            if (W.post.mode != 0u) {
                hipLaunchKernelGGL((k7_tiles<0, true, true>), grid, dim3(256), 0, st, W, total, first, groups);
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
