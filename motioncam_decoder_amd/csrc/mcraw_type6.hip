// mcraw_type6.hip -- gfx950 kernels for the legacy MCRAW frame encoding
// (compressionType 6).  Replaces motioncam::raw::DecodeLegacy,
// lib/RawData_Legacy.cpp:445-495.
//
// The legacy stream is ONE chain of 16-sample records with inline 2-byte
// headers (RawData_Legacy.cpp:377-442): record i+1 starts where record i ends.
// A frame holds ~w*h/16 of them, so the chain is resolved in parallel with
// transition maps: record strides are even and <= 34 bytes, so a fixed 1 KiB
// chunk of the stream can be entered at only 17 offsets ("phases" 0,2,..,32).
//
//   k6_maps    per chunk: table of record strides, then per phase a walk over it to the chunk end
//              -> (exit phase, records started)
//   k6_super   compose 64 chunk maps into one super-chunk map
//   k6_frame   follow the true phase over the super-chunks   -> entry of every super-chunk
//   k6_chunks  follow it over a super-chunk's 64 chunks      -> entry of every chunk
//   k6_rows    per 4 chunks: lanes list the records of one chunk each from its true entry, then
//              all lanes unpack them (MSB-first bitstreams, RawData_Legacy.cpp:38-370), add the
//              references, interleave even/odd columns (:483-486) and crop the padded row (:490)
#include "mcraw_dev.h"

#include "../../include/mcraw_hip.h"

namespace mcraw {

constexpr uint32_t DEAD = 31; // phase value: the chain ended (a record crossed `len`)

// Payload bytes of a record whose header nibble is `b` (RawData_Legacy.cpp:13-32).
__device__ __forceinline__ uint32_t len6_of(uint32_t b) { return b <= 10u ? 2u * b : 32u; }

// ------------------------------------------------------------------ k6_maps
constexpr int MAP_CH_PER_WAVE = 3;           // 3 x 17 phases = 51 of 64 lanes
constexpr uint32_t HALF6 = CHUNK6 / 2;       // even byte positions ("half positions") per chunk
constexpr uint32_t TAB6 = HALF6 + 32;        // stride table per chunk, padded so a finished lane still reads LDS it owns

// Stride of the record whose header byte is `b`, in half positions: (2 + LEN)/2 = 1 + bits for
// bits <= 10, 17 above (RawData_Legacy.cpp:13-32).  Four header bytes per call, one per byte lane.
__device__ __forceinline__ uint32_t stride4(uint32_t hdr4)
{
    const uint32_t x = (hdr4 >> 4) & 0x0F0F0F0Fu;
    const uint32_t g = ((x + 0x05050505u) >> 4) & 0x01010101u;
    const uint32_t big = (g << 8) - g; // 0xFF in the byte lanes where bits >= 11
    return (big & 0x11111111u) | (~big & (x + 0x01010101u));
}

// One wave per 3 chunks.  All 64 lanes first turn the chunk bytes into a table of record strides
// (one byte per even position -- the only places a header can sit); then lane (chunk, phase)
// follows the table from its entry to the chunk end: 1 LDS read + 3 VALU per record instead of
// decoding the header at every step of all 17 walks.
__global__ __launch_bounds__(256) void k6_maps(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ item_base,
                                               int nframes)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_tab[4][MAP_CH_PER_WAVE * TAB6];

    const int f = find_frame(blockIdx.x, item_base, nframes);
    const Plan6 *P = plans + f;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t c0 = ((blockIdx.x - item_base[f]) * 4u + wave) * MAP_CH_PER_WAVE;
    const uint32_t nchunks = P->nchunks, len = P->len;
    if (c0 >= nchunks)
        return;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);

    // 3 KiB of stream for this wave (reads past `len` give 0); 16 bytes -> 8 strides per lane
    uint2 *tab2 = reinterpret_cast<uint2 *>(s_tab[wave]);
#pragma unroll
    for (int q = 0; q < MAP_CH_PER_WAVE; q++) {
        const uint4 v = ld_b128(rs, (c0 + q) * CHUNK6 + lane * 16u);
        // header candidates are bytes 0 and 2 of every dword
        const uint32_t lo = stride4(__builtin_amdgcn_perm(v.y, v.x, 0x06040200u));
        const uint32_t hi = stride4(__builtin_amdgcn_perm(v.w, v.z, 0x06040200u));
        tab2[q * (TAB6 / 8u) + lane] = make_uint2(lo, hi);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    const uint32_t sub = lane / PHASES6, ph = lane - sub * PHASES6;
    const uint32_t c = c0 + sub;
    if (sub >= MAP_CH_PER_WAVE || c >= nchunks)
        return;
    const uint8_t *tab = s_tab[wave] + sub * TAB6;
    const uint32_t cs = c * CHUNK6;
    // RawData_Legacy.cpp:387-388,398-399: a record must end before len-1, i.e. the record at half
    // position q with stride d is the chain's end when cs + 2*(q + d) >= len
    const uint32_t limq = len > cs ? (len - cs + 1u) >> 1 : 0u;
    uint32_t q = ph, count = 0;
    if (limq > HALF6 + 17u) { // no record of this chunk can reach `len`
        const uint8_t *pp = tab + q, *const pe = tab + HALF6;
        while (pp < pe) {
            pp += *pp;
            count++;
        }
        q = static_cast<uint32_t>(pp - tab);
    } else {
        while (q < HALF6) {
            const uint32_t nq = q + tab[q];
            if (nq >= limq)
                break;
            q = nq;
            count++;
        }
    }
    const uint32_t exitph = q < HALF6 ? DEAD : q - HALF6;
    P->cmap[c * PHASES6 + ph] = exitph | (count << 8);
}

// ------------------------------------------------------------------ k6_super
__global__ __launch_bounds__(64) void k6_super(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ super_base,
                                               int nframes)
{
    __shared__ uint32_t s_map[SUPER6 * PHASES6];
    const int f = find_frame(blockIdx.x, super_base, nframes);
    const Plan6 *P = plans + f;
    const uint32_t sc = blockIdx.x - super_base[f];
    const uint32_t first = sc * SUPER6;
    const uint32_t cnt = min(static_cast<uint32_t>(SUPER6), P->nchunks - first);
    const uint32_t lane = threadIdx.x;
    const uint32_t *src = P->cmap + static_cast<size_t>(first) * PHASES6;
    for (uint32_t i = lane; i < cnt * PHASES6; i += 64u)
        s_map[i] = src[i];
    __syncthreads();
    if (lane >= PHASES6)
        return;
    uint32_t p = lane, n = 0;
    for (uint32_t c = 0; c < cnt && p != DEAD; c++) {
        const uint32_t m = s_map[c * PHASES6 + p];
        n += m >> 8;
        p = m & 255u;
    }
    P->smap[sc * PHASES6 + lane] = p | (n << 8);
}

// ------------------------------------------------------------------ k6_frame
constexpr int FRAME_PIECE = 512; // super-chunk maps staged per pass

__global__ __launch_bounds__(64) void k6_frame(const Plan6 *__restrict__ plans)
{
    __shared__ uint32_t s_map[FRAME_PIECE * PHASES6];
    __shared__ uint32_t s_entry[FRAME_PIECE];
    const Plan6 *P = plans + blockIdx.x;
    const uint32_t lane = threadIdx.x, nsuper = P->nsuper;
    uint32_t p = 0, n = 0; // the stream starts with a record at byte 0 (RawData_Legacy.cpp:476)
    for (uint32_t base = 0; base < nsuper; base += FRAME_PIECE) {
        const uint32_t cnt = min(static_cast<uint32_t>(FRAME_PIECE), nsuper - base);
        const uint32_t *src = P->smap + static_cast<size_t>(base) * PHASES6;
        for (uint32_t i = lane; i < cnt * PHASES6; i += 64u)
            s_map[i] = src[i];
        __syncthreads();
        if (lane == 0) {
            for (uint32_t s = 0; s < cnt; s++) {
                s_entry[s] = p | (n << 8);
                if (p != DEAD) {
                    const uint32_t m = s_map[s * PHASES6 + p];
                    // entries carry the record index in 24 bits: a stream with more records than that (the frame
                    // itself has fewer, the host checks) saturates instead of wrapping back into the frame
                    n = min(n + (m >> 8), 0xFFFFFFu);
                    p = m & 255u;
                }
            }
        }
        __syncthreads();
        for (uint32_t i = lane; i < cnt; i += 64u)
            P->sentry[base + i] = s_entry[i];
        p = __shfl(p, 0, 64);
        n = __shfl(n, 0, 64);
        __syncthreads();
    }
    // fewer records than height * recs_per_row inside `len`: the reference would
    // skip the rest and return stale rows (RawData_Legacy.cpp:387-388)
    if (lane == 0 && n < P->nrec)
        atomicOr(P->status, MCRAW_E_TRUNCATED);
}

// ------------------------------------------------------------------ k6_chunks
__global__ __launch_bounds__(64) void k6_chunks(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ super_base,
                                                int nframes)
{
    __shared__ uint32_t s_map[SUPER6 * PHASES6];
    __shared__ uint32_t s_entry[SUPER6];
    const int f = find_frame(blockIdx.x, super_base, nframes);
    const Plan6 *P = plans + f;
    if (*P->status != 0)
        return;
    const uint32_t sc = blockIdx.x - super_base[f];
    const uint32_t first = sc * SUPER6;
    const uint32_t cnt = min(static_cast<uint32_t>(SUPER6), P->nchunks - first);
    const uint32_t lane = threadIdx.x;
    const uint32_t *src = P->cmap + static_cast<size_t>(first) * PHASES6;
    for (uint32_t i = lane; i < cnt * PHASES6; i += 64u)
        s_map[i] = src[i];
    __syncthreads();
    if (lane == 0) {
        const uint32_t e = P->sentry[sc];
        uint32_t p = e & 255u, n = e >> 8;
        for (uint32_t c = 0; c < cnt; c++) {
            s_entry[c] = p | (n << 8);
            if (p != DEAD) {
                const uint32_t m = s_map[c * PHASES6 + p];
                n = min(n + (m >> 8), 0xFFFFFFu); // (see k6_frame)
                p = m & 255u;
            }
        }
    }
    __syncthreads();
    if (lane < cnt)
        P->centry[first + lane] = s_entry[lane];
}

// ------------------------------------------------------------------ k6_rows

// Samples 4*qt..4*qt+3 of the record at byte `ro` of the staged stream, reference NOT yet added;
// *ref receives the header's 12-bit reference (RawData_Legacy.cpp:372-375).  The payload is an
// MSB-first bitstream of sb-bit fields (sb = header nibble for <= 10, 16 big-endian raw bits above,
// RawData_Legacy.cpp:38-370), so the lane's four fields are the top 4*sb bits of a 64-bit
// big-endian window that starts 4*qt*sb bits into the payload.  The window is cut out of three
// aligned dwords with two byte permutes (alignment and byte swap in one selector).
__device__ __forceinline__ void quad6(const uint8_t *__restrict__ bytes, uint32_t ro, uint32_t qt, uint32_t v[4],
                                      uint32_t *ref)
{
    const uint32_t h = *reinterpret_cast<const uint16_t *>(bytes + ro); // records start on even bytes
    const uint32_t hb = (h >> 4) & 15u;
    *ref = ((h & 15u) << 8) | (h >> 8);
    const uint32_t sb = hb <= 10u ? hb : 16u;
    const uint32_t ob = 4u * qt * sb;       // bit offset of my fields in the payload
    const uint32_t B = ro + 2u + (ob >> 3); // first byte of the window
    const uint32_t *w = reinterpret_cast<const uint32_t *>(bytes + (B & ~3u));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    const uint32_t k = B & 3u;
    const uint32_t sel = 0x00010203u + __builtin_amdgcn_perm(k, k, 0u); // k in every byte lane
    const uint32_t hi0 = __builtin_amdgcn_perm(d1, d0, sel), lo0 = __builtin_amdgcn_perm(d2, d1, sel);
    // odd field width and odd quarter: the fields start on a nibble
    const uint64_t W = ((static_cast<uint64_t>(hi0) << 32) | lo0) << (ob & 4u);
    const uint32_t hi = static_cast<uint32_t>(W >> 32);
    // fields 0 and 1 end within the high dword for every width; 2 and 3 can reach into the low one
    v[0] = __builtin_amdgcn_ubfe(hi, (32u - sb) & 31u, sb);
    v[1] = __builtin_amdgcn_ubfe(hi, (32u - 2u * sb) & 31u, sb);
    v[2] = __builtin_amdgcn_ubfe(static_cast<uint32_t>(W >> ((64u - 3u * sb) & 63u)), 0u, sb);
    v[3] = __builtin_amdgcn_ubfe(static_cast<uint32_t>(W >> ((64u - 4u * sb) & 63u)), 0u, sb);
}

// One wave per ROWS_CH consecutive chunks (4 KiB of stream), four waves per workgroup.  Lanes walk
// one chunk each from its resolved entry and note where every record starts; the records of the wave
// form one contiguous index range, so the list is flat.  Then ALL lanes unpack, four lanes per record
// PAIR: a lane owns samples 4q..4q+3 of the even-column record and of the odd-column record = 8
// consecutive pixels = one 16-byte store; 8 lanes fill a 128-byte line.
//
// A wave owns the pairs whose EVEN record starts in its chunks.  When its range ends on an even
// record, the odd partner starts right behind it, at most 32 bytes into the next wave's first chunk and
// inside this wave's staged slack (STAGE); when its range starts on an odd record, that record belongs
// to the previous wave's last pair.  So every pair is decoded whole, by one lane quartet.
//
// (A wave-uniform walk of ONE chunk per wave spent 64 lanes on a scalar chain and made this kernel
// issue-bound; more chunks per wave spread the walk over more lanes but cost LDS, and 4 measured best.)
// The list holds one entry per record PAIR -- the position of the even record; the odd one starts where
// the even one ends, which the unpacking lane knows from the even record's header -- so a round covers
// 1024 records in 1 KiB of LDS: all data but runs of 2-byte records stays on the single-round path
// (4-byte records, 1-bit residuals of a nearly flat frame, are 1024 per wave).
constexpr uint32_t ROWS_CAP = 256u * ROWS_CH; // records per round (typical: ~70 per chunk; worst case 512 per chunk -> 2 rounds)
static_assert(ROWS_CAP <= 1024u && ROWS_CAP % 2u == 0u, "the division-free row arithmetic in k6_rows assumes at most 512 pairs per round");

#ifndef K6_ABL
#define K6_ABL 0 // timing experiments only: 1 no stores, 3 no walk
#endif

template <int POST> // 0 = the plain mosaic, else bits per sample of the post stage's rows
__global__ __launch_bounds__(256) void k6_rows(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ item_base,
                                               int nframes, const Post post)
{
    constexpr uint32_t STAGE = ROWS_CH * CHUNK6 + 64 + 16; // + the reach of a record that starts 32 bytes past the chunks
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[4][STAGE];
    // the list of a round, one of two layouts: every record r at [r - wlo] (up to ROWS_CAP / 2 records: the
    // common case, one LDS read gives both records of a pair), or one entry per PAIR at [(r - wlo) / 2]
    // holding the even record only (up to ROWS_CAP records; the unpacking lane finds the odd one behind it)
    __shared__ __attribute__((aligned(4))) uint16_t s_pos[4][ROWS_CAP / 2 + 2];
    __shared__ uint32_t s_pairmode[4];
    __shared__ uint32_t s_ent[4 * ROWS_CH];

    // workgroups run over the batch BACKWARDS: k6_maps has just streamed the whole input through the
    // Infinity Cache front to back, so its tail -- what a backward pass touches first -- is still there
    const uint32_t bid = gridDim.x - 1u - blockIdx.x;
    const int f = find_frame(bid, item_base, nframes);
    const Plan6 *P = plans + f;
    if (*P->status != 0)
        return; // whole workgroup
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t nchunks = P->nchunks, nrec = P->nrec, len = P->len;
    const uint32_t c0 = ((bid - item_base[f]) * 4u + wave) * ROWS_CH;
    const bool have = c0 < nchunks;
    // entries of my chunks (lane j) and of the chunk after them (lane ROWS_CH)
    uint32_t e = DEAD;
    if (have && lane <= ROWS_CH && c0 + lane < nchunks)
        e = P->centry[c0 + lane];
    const uint32_t e0 = __builtin_amdgcn_readfirstlane(e);
    const uint32_t enext = __shfl(e, ROWS_CH, 64);
    const bool inner = c0 + ROWS_CH < nchunks && (enext & 255u) != DEAD && (enext >> 8) <= nrec;
    // records that start in my chunks: [I0, Iend); nrec is even (two records per 32 columns)
    const uint32_t I0 = e0 >> 8;
    const uint32_t Iend = min(nrec, (c0 + ROWS_CH < nchunks) ? (enext >> 8) : nrec);
    // records I decode: the pairs whose even record is among them
    const uint32_t R0 = (I0 + 1u) & ~1u, R1 = (Iend + 1u) & ~1u;
    // live: the chain reaches this wave's chunks and there are pairs for it
    const bool live = have && (e0 & 255u) != DEAD && R0 < R1;
    const uint32_t N = live ? R1 - R0 : 0u;

    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    const uint32_t cs0 = c0 * CHUNK6;
    if (live) {
        uint4 *dst = reinterpret_cast<uint4 *>(s_bytes[wave]);
#pragma unroll
        for (uint32_t q = 0; q < (STAGE / 16 + 63) / 64; q++)
            if (lane + 64u * q < STAGE / 16)
                dst[lane + 64u * q] = ld_b128(rs, cs0 + (lane + 64u * q) * 16u);
        if (lane < ROWS_CH)
            s_ent[wave * ROWS_CH + lane] = e;
    }
    // lean: every chunk of this wave runs to its end, the next entry bounds the last one, no record can
    // reach `len`, and the records fit one round of the list
    const bool lean = live && inner && N <= ROWS_CAP && cs0 + STAGE < len;
    const bool pairmode = N > ROWS_CAP / 2u; // which list layout this wave's single round uses
    if (lane == 0)
        s_pairmode[wave] = pairmode ? 1u : 0u;
    // When that holds for all four waves (everywhere but at the ends of a frame and in runs of tiny
    // records), ONE wave walks the 16 chunks of the workgroup, a lane each: a walk keeps a wave busy for
    // ~70 dependent steps whatever the number of walking lanes, so four waves walking four chunks each
    // would spend four times the issue slots on it.
    const bool coop = __syncthreads_and(lean) != 0;

    const uint8_t *bytes = s_bytes[wave];
    const uint32_t width = static_cast<uint32_t>(P->width);
    const bool fast = P->fast_store != 0u;
    uint16_t *const out = P->out;

    // row arithmetic without per-lane division: pairs per row `ppr`; a round spans < 2 rows when
    // ppr >= 512, otherwise n / ppr for n < 1024 is exact as (n * ceil(2^20 / ppr)) >> 20
    const uint32_t ppr = P->recs_per_row >> 1;
    const bool widerow = ppr >= 512u;
    const uint32_t m20 = widerow ? 0u : ((1u << 20) + ppr - 1u) / ppr;

    // Unpack the pairs of records [wlo, whi) (both even) listed in s_pos[wave]: four lanes per pair
    auto unpack_round = [&](uint32_t wlo, uint32_t whi, bool by_pair) {
        const uint32_t pair0 = wlo >> 1;
        const uint16_t *pairs = s_pos[wave];
        const uint32_t *both = reinterpret_cast<const uint32_t *>(s_pos[wave]);
        const uint32_t y0 = pair0 / ppr, r0 = pair0 - y0 * ppr; // wave-uniform
        const uint32_t row0 = y0 * width;
        const uint32_t ntask = 2u * (whi - wlo);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
        for (uint32_t t = lane; t < ntask; t += 64u) {
            // one task: 8 pixels (even columns from record A, odd columns from record B, uint16 wrap
            // on the reference add) and where they go
            const uint32_t q = t >> 2, qt = t & 3u;
            uint32_t roa, rob;
            if (by_pair) { // the odd-column record starts where the even-column one ends (RawData_Legacy.cpp:377-442)
                roa = pairs[q];
                rob = roa + 2u + len6_of(static_cast<uint32_t>(bytes[roa]) >> 4);
            } else {
                const uint32_t ro2 = both[q];
                roa = ro2 & 0xffffu;
                rob = ro2 >> 16;
            }
            uint32_t va[4], vb[4], refa, refb;
            quad6(bytes, roa, qt, va, &refa);
            quad6(bytes, rob, qt, vb, &refb);
            const uint32_t n = r0 + q;
            const uint32_t dy = widerow ? (n >= ppr ? 1u : 0u) : __umul24(n, m20) >> 20;
            const uint32_t x = (n - __umul24(dy, ppr)) * 32u + 8u * qt; // RawData_Legacy.cpp:479-486
            const u16x2 refs = __builtin_bit_cast(u16x2, refa | (refb << 16));
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                o[j] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, va[j] | (vb[j] << 16)) + refs);
            if (POST) { // black levels / 12-bit strip rows (mcraw_dev.h); padded columns are cropped
                if (x < width)
                    post_store8<true, POST>(out, post, width, y0 + dy, x, o, min(8u, width - x), fast);
                continue;
            }
            // (y0 + dy) * width without a per-lane 32-bit multiply: a wide row (width can exceed 24
            // bits) advances by at most one row per round, a narrow one has width < 2^14
            uint16_t *px = out + (row0 + (widerow ? (dy ? width : 0u) : __umul24(dy, width)) + x);
            if (K6_ABL == 1) {
                if ((o[0] ^ o[1] ^ o[2] ^ o[3]) == 0x12345678u)
                    px[0] = 1;
            } else if (fast && x + 8u <= width) {
                const u32x4 v = {o[0], o[1], o[2], o[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(px));
            } else if (x + 8u <= width) { // rows off the 16-byte grid: still one (unaligned) 16-byte store
                typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
                const u32x4_u v = {o[0], o[1], o[2], o[3]};
                *reinterpret_cast<u32x4_u *>(px) = v;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 8u; j++) // padded columns are cropped (RawData_Legacy.cpp:490)
                    if (x + j < width)
                        px[j] = static_cast<uint16_t>(o[j >> 1] >> (16u * (j & 1u)));
            }
        }
    };

    if (coop) {
        if (K6_ABL != 3 && wave == 0u && lane < 4u * ROWS_CH) {
            const uint32_t w = lane / ROWS_CH, j = lane - w * ROWS_CH; // chunk j of wave w
            const uint32_t ej = s_ent[lane], first = ((s_ent[w * ROWS_CH] >> 8) + 1u) & ~1u;
            const uint8_t *base = s_bytes[w];
            const uint8_t *p = base + j * CHUNK6 + 2u * (ej & 255u), *const pe = base + (j + 1u) * CHUNK6;
            uint32_t idx = ej >> 8;
            if (s_pairmode[w]) {
                // an odd first record belongs to the previous wave's last pair: never listed
                while (p < pe) { // the stride decode, and a store for every second record
                    const uint32_t hb = static_cast<uint32_t>(*p) >> 4;
                    if ((idx & 1u) == 0u)
                        s_pos[w][(idx - first) >> 1] = static_cast<uint16_t>(p - base);
                    idx++;
                    p += 2u + len6_of(hb);
                }
            } else {
                uint16_t *lp = s_pos[w] + static_cast<int32_t>(idx - first); // [-1] for an odd first record:
                if (idx < first) {                                             // skipped, see above
                    p += 2u + len6_of(static_cast<uint32_t>(*p) >> 4);
                    lp++;
                }
                while (p < pe) { // nothing but the stride decode in the loop
                    const uint32_t hb = static_cast<uint32_t>(*p) >> 4;
                    *lp++ = static_cast<uint16_t>(p - base);
                    p += 2u + len6_of(hb);
                }
                // the walk of a wave's last chunk stops on the next wave's first record: the partner of my
                // last record when my range ends on an even one (lp is then at an odd list index)
                if (j == ROWS_CH - 1u && ((lp - s_pos[w]) & 1))
                    *lp = static_cast<uint16_t>(p - base);
            }
        }
        __syncthreads();
        unpack_round(R0, R1, pairmode);
        return;
    }
    if (!live)
        return;
    const bool walker = lane < ROWS_CH && (e & 255u) != DEAD && c0 + lane < nchunks;
    // a walker keeps its place from round to round (restarting at the chunk entry every round made a
    // run of 2-byte records cost rounds x 512 steps per lane)
    uint32_t pos = 2u * (e & 255u), idx = e >> 8;
    for (uint32_t base = 0; base < N; base += ROWS_CAP) {
        const uint32_t wlo = R0 + base, whi = min(R1, wlo + ROWS_CAP); // records of this round (both even)
        if (K6_ABL != 3 && walker) {
            const uint32_t off = lane * CHUNK6;
            while (idx < whi && pos < CHUNK6) {
                // Runs of 2-byte records (flat or clipped image regions) are what makes a wave land here:
                // eight of them fill an aligned 16-byte line whose even bytes all have a zero high
                // nibble -- one LDS read then lists four pairs instead of one record.
                const uint32_t a = off + pos;
                if ((a & 15u) == 0u && (idx & 1u) == 0u && pos + 16u <= CHUNK6 && idx >= wlo && idx + 8u <= whi &&
                    cs0 + a + 16u < len) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(bytes + a);
                    if (((q.x | q.y | q.z | q.w) & 0x00F000F0u) == 0u) {
#pragma unroll
                        for (uint32_t k = 0; k < 4u; k++)
                            s_pos[wave][((idx - wlo) >> 1) + k] = static_cast<uint16_t>(a + 4u * k);
                        pos += 16u;
                        idx += 8u;
                        continue;
                    }
                }
                const uint32_t nx = pos + 2u + len6_of(static_cast<uint32_t>(bytes[off + pos]) >> 4);
                if (cs0 + off + nx >= len)
                    break; // k6_frame has already failed the frame if records are missing
                if (idx >= wlo && (idx & 1u) == 0u)
                    s_pos[wave][(idx - wlo) >> 1] = static_cast<uint16_t>(off + pos);
                pos = nx;
                idx++;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        unpack_round(wlo, whi, true);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier(); // the list is rewritten by the next round
    }
}

// ------------------------------------------------------------------ launchers

void launch_k6_maps(const Plan6 *plans, const uint32_t *item_base, int nframes, uint32_t nitems, hipStream_t st)
{
    hipLaunchKernelGGL(k6_maps, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes);
}

void launch_k6_resolve(const Plan6 *plans, const uint32_t *super_base, int nframes, uint32_t nsuper_items,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k6_super, dim3(nsuper_items), dim3(64), 0, st, plans, super_base, nframes);
    hipLaunchKernelGGL(k6_frame, dim3(nframes), dim3(64), 0, st, plans);
    hipLaunchKernelGGL(k6_chunks, dim3(nsuper_items), dim3(64), 0, st, plans, super_base, nframes);
}

void launch_k6_rows(const Plan6 *plans, const uint32_t *item_base, int nframes, uint32_t nitems, const Post &post,
                    hipStream_t st)
{
    if (post.mode == 0u) {
        hipLaunchKernelGGL(k6_rows<0>, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes, post);
        return;
    }
    switch (post_bits(post.mode)) { // one kernel instance per row format
    case 12: hipLaunchKernelGGL(k6_rows<12>, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes, post); break;
    case 10: hipLaunchKernelGGL(k6_rows<10>, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes, post); break;
    case 14: hipLaunchKernelGGL(k6_rows<14>, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes, post); break;
    default: hipLaunchKernelGGL(k6_rows<16>, dim3(nitems), dim3(256), 0, st, plans, item_base, nframes, post); break;
    }
}

} // namespace mcraw
