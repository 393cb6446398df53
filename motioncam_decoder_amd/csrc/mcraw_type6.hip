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
// One kernel, one pass over the stream (k6_decode): a workgroup stages 16 chunks, walks all 17 phases of
// each (and of the chunk in front of them) over a byte table of record strides -> per chunk and phase the exit
// phase and the records started; finds where the TRUE chain enters its chunks -- the map of the chunk in front of a
// chunk almost always sends all 17 phases to one exit, because a wrong chain reads payload bytes as headers and
// falls onto the true one within a few hundred bytes -- and the index of its first record by decoupled look-back
// over the frame's earlier workgroups; lists the records of its chunks from those entries and unpacks them
// (MSB-first bitstreams, RawData_Legacy.cpp:38-370), adds the references, interleaves even/odd columns (:483-486)
// and crops the padded row (:490).
#include "mcraw_dev.h"

#include <cstdlib>

#include "../../include/mcraw_hip.h"

namespace mcraw {

constexpr uint32_t DEAD = 31; // phase value: the chain ended (a record crossed `len`)

// Payload bytes of a record whose header nibble is `b` (RawData_Legacy.cpp:13-32).
__device__ __forceinline__ uint32_t len6_of(uint32_t b) { return b <= 10u ? 2u * b : 32u; }

constexpr uint32_t HALF6 = CHUNK6 / 2; // even byte positions ("half positions") per chunk

// Stride of the record whose header byte is `b`, in half positions: (2 + LEN)/2 = 1 + bits for
// bits <= 10, 17 above (RawData_Legacy.cpp:13-32).  Four header bytes per call, one per byte lane.
__device__ __forceinline__ uint32_t stride4(uint32_t hdr4)
{
    const uint32_t x = (hdr4 >> 4) & 0x0F0F0F0Fu;
    const uint32_t g = ((x + 0x05050505u) >> 4) & 0x01010101u;
    const uint32_t big = byte_mask(g); // 0xFF in the byte lanes where bits >= 11
    return (big & 0x11111111u) | (~big & (x + 0x01010101u));
}

// 24-bit x 24-bit multiply, low 32 bits (full rate).  __umul24 of a per-lane and a uniform operand comes out of the compiler as
// v_and + v_mul_lo_u32, which issues at a quarter of the rate.
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Look-back state words are 64 bits: the launch's epoch in the high half (the state buffer is never cleared: words
// of earlier launches carry older epochs and read as "not there yet"), the payload in the low half.  Every word is
// complete in itself, so publishing one is a single relaxed store at device scope and needs no fence.
constexpr uint32_t RES_AGG = 1u, RES_PREFIX = 2u; // Look6::res payload: state << 30 | records << 5 | exit phase
constexpr uint32_t EX_PHASE = 1u, EX_MAP = 2u;    // Look6::ex payload: kind << 30 | exit phase (kind 1)
#ifdef MCRAW_INJECT_LOST // test builds (tests/test_gpu_lookback_fault.py): segment 3 of the batch's first legacy frame never publishes
constexpr uint32_t SPIN6 = 1u << 12;
#else
constexpr uint32_t SPIN6 = 1u << 20;              // polls before a workgroup gives the frame up (never seen; a hang is worse)
#endif

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

// Unpacking: one wave per ROWS_CH consecutive chunks (4 KiB of stream), four such waves per workgroup.  Lanes
// list where every record of the wave's chunks starts (a quarter chunk per lane, from where the true chain crosses
// into it); the records of the wave form one contiguous index range, so the list is flat.  Then ALL lanes unpack,
// four lanes per record PAIR: a lane owns samples 4q..4q+3 of the even-column record and of the odd-column record =
// 8 consecutive pixels = one 16-byte store; 8 lanes fill a 128-byte line.
//
// A wave owns the pairs whose EVEN record starts in its chunks.  When its range ends on an even
// record, the odd partner starts right behind it, at most 32 bytes into the next wave's first chunk and
// inside the staged slack; when its range starts on an odd record, that record belongs
// to the previous wave's last pair.  So every pair is decoded whole, by one lane quartet.
//
// The list has two layouts: every record (up to ROWS_CAP / 2 per wave), or one entry per record PAIR -- the
// position of the even record; the odd one starts where the even one ends, which the unpacking lane knows from
// the even record's header -- so a round covers 1024 records in 1 KiB of LDS: all data but runs of 2-byte
// records stays on the single-round path (4-byte records, 1-bit residuals of a nearly flat frame, are 1024 per wave).
constexpr uint32_t ROWS_CAP = 256u * ROWS_CH; // records per round (typical: ~70 per chunk; worst case 512 per chunk -> 2 rounds)
static_assert(ROWS_CAP <= 1024u && ROWS_CAP % 2u == 0u, "the division-free row arithmetic of the unpack assumes at most 512 pairs per round");

#ifndef K6_ABL
#define K6_ABL 0 // timing experiments only: 1 no stores, 3 no walk, 5 the front alone (no bytes in LDS, waves 0-3 leave behind the maps, no unpack), 6 the same with the waves staying
#endif

// ------------------------------------------------------------------ k6_decode
constexpr uint32_t TAIL6 = 128;           // tasks (8 pixels each) of every unpacking wave's list that the fifth wave takes over
constexpr uint32_t DEC_CH = 4 * ROWS_CH;  // chunks per workgroup: four unpacking waves
constexpr uint32_t DEC_T = 320;           // ... and a fifth wave: (DEC_CH + 1) * 17 = 289 map walks need five
constexpr uint32_t RUN6 = 16;              // table entry at the first of sixteen 2-byte records in a row: jump over them (no record has this stride)
constexpr uint32_t QTAB = HALF6 / 4;      // byte walk table of a quarter chunk: 128 strides (a walk that has left its quarter is held
                                          // by a select in the walk, not by zeros behind the table)
constexpr uint32_t TABQ = 4 * QTAB + 16;  // ... of a chunk (+ 16: the tables of neighbouring chunks, which one wave walks at about the same
                                          // positions, start 4 LDS banks apart)
static_assert((DEC_CH + 1) * PHASES6 <= DEC_T, "one thread per (chunk, phase)");

#ifdef MCRAW_DIAG // phase stamps of every workgroup (timing experiments only; not in the product library)
constexpr int K6_PROF_WG = 1 << 16;
// [0..13]: stage stamps (s_memtime), [14] live, [15] look-back spins; [16], [17]: s_memrealtime (100 MHz) at wave 0's start and end;
// [18 + w]: at wave w's end, its stores landed; [23], [24]: HW_ID, XCC_ID of wave 0; [27 + w]: HW_ID of wave w
constexpr int K6_PROF_N = 32;
__device__ uint32_t g_k6_prof[K6_PROF_WG][K6_PROF_N];
#define K6_STAMP(slot, who)                                                                                            \
    do {                                                                                                               \
        if (threadIdx.x == (who) && blockIdx.x < K6_PROF_WG) {                                                         \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
            g_k6_prof[blockIdx.x][slot] = static_cast<uint32_t>(now_ - stamp_);                                        \
            stamp_ = now_;                                                                                             \
        }                                                                                                              \
    } while (0)
#define K6_COUNT(slot, v) (blockIdx.x < K6_PROF_WG ? (void)(g_k6_prof[blockIdx.x][slot] = static_cast<uint32_t>(v)) : (void)0)
#else
#define K6_STAMP(slot, who)
#define K6_COUNT(slot, v)
#endif

// A workgroup works on ONE frame (the launch interleaves the frames: a segment's predecessors then started long before
// it) and takes its segment -- DEC_CH chunks -- from the frame's ticket counter: the segments it may have to wait on were all taken by
// workgroups that are running or done, whatever order the hardware starts workgroups in.  (One counter per frame,
// each in its own 256 bytes: one counter for the batch serialised the launch -- 31 000 device-scope atomics on one
// address took 0.37 ms.)  Five waves: 17 x 17 map walks need 289 threads; the fifth wave then resolves the entries
// while the other four wait; those four unpack their chunks, and the fifth a share of every list (TAIL6).
template <int POST> // 0 = the plain mosaic, else bits per sample of the post stage's rows
__global__ __launch_bounds__(DEC_T) void k6_decode(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ wg_tab,
                                                   uint32_t stage0, const Look6 look, uint32_t *__restrict__ tickets,
                                                   uint32_t epoch, uint32_t nframes, uint32_t smax, const Post post)
{
    // the segment's stream: its DEC_CH chunks and the reach of a record that starts 32 bytes past them (the chunk in
    // front of them, whose map tells where the segment is entered, only becomes a walk table)
    constexpr uint32_t OWN = DEC_CH * CHUNK6, SLACK = 64 + 32;
    constexpr uint32_t NPIECE = (CHUNK6 + OWN + SLACK) / 16; // 16-byte pieces, piece 0 at stream offset (cfirst - 1) * CHUNK6
    constexpr uint32_t NROUND = (NPIECE + DEC_T - 1) / DEC_T;
    __shared__ __attribute__((aligned(16))) uint8_t s_own[OWN + SLACK];
    // walk tables of the DEC_CH + 1 chunks, a quarter chunk at a time: 128 strides, then 32 zeros where a walk that
    // has left the quarter stays; dead once the records are counted: the record lists take their place
    __shared__ __attribute__((aligned(16))) uint8_t s_tab[(DEC_CH + 1) * TABQ];
    __shared__ __attribute__((aligned(4))) uint8_t s_qx[(DEC_CH + 1) * 4 * PHASES6]; // [chunk][quarter][entry phase] = phase at which the quarter is left
    __shared__ uint8_t s_cx[(DEC_CH + 1) * PHASES6];     // [chunk][entry phase] = exit phase: the four composed
    __shared__ uint32_t s_exits[DEC_CH + 1]; // per chunk: the set of exit phases its 17 walks reach, one bit each
    // entry of my chunks and of the one behind them (phase | first record << 8), and of every quarter of my chunks: written
    // behind the look-back, when the exit sets and the quarters' maps have served -- they take their LDS (26 996 bytes in all: LDS is
    // handed out in units of 1 280 bytes, five workgroups per CU up to 32 768; six would need <= 26 880 and are no faster)
    uint32_t *const s_ent = s_exits;
    uint32_t *const s_ent4 = reinterpret_cast<uint32_t *>(s_qx);
    static_assert(sizeof(s_qx) >= DEC_CH * 4 * sizeof(uint32_t), "the quarters' entries fit where their maps were");
    __shared__ uint32_t s_ticket, s_coop, s_runs;
    // the list of a round, one of two layouts: every record r at [r - wlo] (up to ROWS_CAP / 2 records: the
    // common case, one LDS read gives both records of a pair), or one entry per PAIR at [(r - wlo) / 2]
    // holding the even record only (up to ROWS_CAP records; the unpacking lane finds the odd one behind it)
    typedef uint16_t PosList[ROWS_CAP / 2 + 2];
    static_assert(sizeof(PosList) * 4 <= sizeof(s_tab), "the lists live where the walk tables were");
    static_assert((DEC_CH + 1) * TABQ < 65536u, "table addresses fit the walkers' 16 bits");
    PosList *const s_pos = reinterpret_cast<PosList *>(s_tab);

#ifdef MCRAW_DIAG
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x < K6_PROF_WG) {
        g_k6_prof[blockIdx.x][6] = static_cast<uint32_t>(stamp_);
        g_k6_prof[blockIdx.x][16] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
        g_k6_prof[blockIdx.x][23] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, all 32 bits
        g_k6_prof[blockIdx.x][24] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
    }
    if ((threadIdx.x & 63u) == 0u && blockIdx.x < K6_PROF_WG)
        g_k6_prof[blockIdx.x][27 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#define K6_END()                                                                                                       \
    do {                                                                                                               \
        if ((threadIdx.x & 63u) == 0u && blockIdx.x < K6_PROF_WG)                                                     \
            g_k6_prof[blockIdx.x][18 + (threadIdx.x >> 6)] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());  \
    } while (0)
#else
#define K6_END()
#endif
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    // Which frame this workgroup works on, and the segment it will most likely be given: the launch goes over the
    // frames round by round -- round r = segment r of every frame that has one --, so a frame's segments start in order
    // and far apart, and no workgroup is launched for nothing however the frames' sizes differ (the host's table: the
    // frames by falling size; stage t = the rounds in which all but the t smallest are still in play).
    // (the rounds that every frame takes part in need no table: most batches hold frames of one size)
    const bool all_in = blockIdx.x < stage0;
    const uint32_t stage = all_in ? 0u : static_cast<uint32_t>(find_frame(blockIdx.x, wg_tab, static_cast<int>(nframes)));
    const uint32_t inplay = nframes - stage, wrel = all_in ? blockIdx.x : blockIdx.x - wg_tab[stage];
    const uint32_t f = all_in ? wrel % inplay : wg_tab[2u * nframes + 1u + wrel % inplay];
    if (tid == 0) {
        s_runs = 0u;
        s_ticket = atomicAdd(tickets + f * TICKET_STRIDE6, 1u);
    }
    const Plan6 *P = plans + f;
    const uint32_t nchunks = P->nchunks, nrec = P->nrec, len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    // the ticket takes a round trip to the frame's counter: start loading what it will almost certainly say
    // (workgroups start in order), and load again if it says otherwise
    uint32_t seg = (all_in ? 0u : wg_tab[nframes + 1u + stage]) + wrel / inplay;
    uint4 v[NROUND];
    auto fetch = [&]() { // piece i at stream offset (seg * DEC_CH - 1) * CHUNK6 + 16 i; past `len`: reads 0
#pragma unroll
        for (uint32_t r = 0; r < NROUND; r++) {
            const uint32_t i = tid + r * DEC_T;
            v[r] = (seg || i >= CHUNK6 / 16u) ? ld_b128_nt(rs, seg * (DEC_CH * CHUNK6) - CHUNK6 + i * 16u) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    fetch();
    __syncthreads();
    K6_STAMP(0, 0);
    K6_STAMP(8, 256);
    if (s_ticket != seg) {
        seg = s_ticket;
        fetch();
    }
    const uint32_t cfirst = seg * DEC_CH;
    if (cfirst >= nchunks)
        return; // whole workgroup

    // ---- stage the stream and turn it into the walk tables (one byte per even position: the record stride in half
    // positions; 32 zeros behind every chunk, where a walk that has left the chunk stays)
#pragma unroll
    for (uint32_t r = 0; r < NROUND; r++) {
        const uint32_t i = tid + r * DEC_T;
        uint32_t lo = 0, hi = 0;
        uint8_t *dst = nullptr;
        bool ones = false;
        if (i < NPIECE && (seg || i >= CHUNK6 / 16u)) {
            if (K6_ABL < 5 && i >= CHUNK6 / 16u)
                *reinterpret_cast<uint4 *>(s_own + (i - CHUNK6 / 16u) * 16u) = v[r];
            if (i < (DEC_CH + 1u) * (CHUNK6 / 16u)) {
                // header candidates are bytes 0 and 2 of every dword
                lo = stride4(__builtin_amdgcn_perm(v[r].y, v[r].x, 0x06040200u));
                hi = stride4(__builtin_amdgcn_perm(v[r].w, v[r].z, 0x06040200u));
                const uint32_t k = i / (CHUNK6 / 16u), j = i % (CHUNK6 / 16u);
                // RawData_Legacy.cpp:387-388,398-399: a record must end before len-1, i.e. the record at half position
                // q of its chunk with stride d is the chain's end when cs + 2*(q + d) >= len: its stride becomes 0,
                // the walk stays in front of it
                const uint32_t cs = (cfirst + k - 1u) * CHUNK6;
                const uint32_t limq = len > cs ? (len - cs + 1u) >> 1 : 0u;
                if (limq <= HALF6 + 17u) {
#pragma unroll
                    for (uint32_t u = 0; u < 8u; u++) {
                        uint32_t &wd = u < 4u ? lo : hi;
                        const uint32_t sh = 8u * (u & 3u);
                        if (j * 8u + u + ((wd >> sh) & 255u) >= limq)
                            wd &= ~(255u << sh);
                    }
                }
                // sixteen 2-byte records in a row (a flat or clipped image region would otherwise cost a step per record,
                // 128 per quarter): a walk that arrives at the first one jumps over all sixteen.  No record has that
                // stride, so the entry also tells the record count what it stands for.
                ones = limq > HALF6 + 17u && lo == 0x01010101u && hi == 0x01010101u;
                dst = s_tab + k * TABQ + (j >> 4) * QTAB + (j & 15u) * 8u;
            }
        }
        // (pieces i and i + 1 sit in neighbouring lanes; an even piece and its successor share a quarter; most waves
        // hold no such piece at all and skip this)
        const unsigned long long om = __ballot(ones);
        if (om & (om >> 1)) {
            const bool next_ones = ((om >> (lane & 63u)) >> 1) & 1ull;
            if (ones && next_ones && (i & 1u) == 0u) {
                lo = 0x01010100u | RUN6;
                s_runs = 1u;
            }
        }
        if (dst)
            *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
    }
    if (tid <= DEC_CH)
        s_exits[tid] = 0u;
    __syncthreads();
    K6_STAMP(1, 0);
    K6_STAMP(9, 256);

    // ---- transition maps, phases only: thread (chunk k, phase) walks the four quarters of its chunk, each from its
    // phase, side by side -- a step is one LDS read and one addition per walk, four independent walks in flight -- and
    // notes at which phase each is left; three lookups then compose the chunk's map from the quarters'.
    const bool mapper = tid < (DEC_CH + 1u) * PHASES6;
    const uint32_t mk = tid / PHASES6, mph = tid - mk * PHASES6; // (mk = 0: the chunk in front of the segment)
    const bool mapped = mapper && (mk || seg) && cfirst + mk - 1u < nchunks;
    if (mapped) {
        const uint32_t q0 = mk * TABQ + mph;
        uint32_t A0 = q0, A1 = q0 + QTAB, A2 = q0 + 2u * QTAB, A3 = q0 + 3u * QTAB;
        const uint32_t end0 = mk * TABQ + HALF6 / 4u; // (end of quarter r: end0 + r * QTAB)
        uint32_t moved;
        do { // until no walk of the wave moved any more: each has left its quarter, or stands in front of the chain's
             // last record (stride 0)
#pragma unroll
            for (int u = 0; u < 2; u++) { // (a walk that has left its quarter stays where it is)
                // (the reads are unconditional -- what lies behind a quarter is LDS of this workgroup -- and a select drops them)
                const uint32_t r0 = s_tab[A0], r1 = s_tab[A1], r2 = s_tab[A2], r3 = s_tab[A3];
                const uint32_t t0 = A0 < end0 ? r0 : 0u, t1 = A1 < end0 + QTAB ? r1 : 0u, t2 = A2 < end0 + 2u * QTAB ? r2 : 0u,
                               t3 = A3 < end0 + 3u * QTAB ? r3 : 0u;
                A0 += t0, A1 += t1, A2 += t2, A3 += t3;
                moved = t0 | t1 | t2 | t3;
            }
        } while (__any(moved != 0u));
        uint8_t *qx = s_qx + mk * 4u * PHASES6 + mph;
        qx[0] = static_cast<uint8_t>(A0 < end0 ? DEAD : A0 - end0);
        qx[PHASES6] = static_cast<uint8_t>(A1 < end0 + QTAB ? DEAD : A1 - (end0 + QTAB));
        qx[2 * PHASES6] = static_cast<uint8_t>(A2 < end0 + 2u * QTAB ? DEAD : A2 - (end0 + 2u * QTAB));
        qx[3 * PHASES6] = static_cast<uint8_t>(A3 < end0 + 3u * QTAB ? DEAD : A3 - (end0 + 3u * QTAB));
    }
    __syncthreads();
    K6_STAMP(2, 0);
    if (mapped) {
        const uint8_t *qx = s_qx + mk * 4u * PHASES6;
        uint32_t x = mph;
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++)
            x = x == DEAD ? DEAD : qx[r * PHASES6 + x];
        s_cx[mk * PHASES6 + mph] = static_cast<uint8_t>(x);
        atomicOr(&s_exits[mk], 1u << x);
    }
    __syncthreads();

    if (K6_ABL == 5 && wave < 4u)
        return;
    // ---- entries of my chunks (wave DEC_T / 64 - 1; the others wait at the barrier below)
    const uint32_t cnt = min(static_cast<uint32_t>(DEC_CH), nchunks - cfirst);
    // What unpacking wave w has to do, from the entries in s_ent (valid once the fifth wave has written them).
    struct Range6 {
        uint32_t R0, R1, N; // records it decodes: the pairs whose even record starts in its chunks
        bool live;          // the chain reaches its chunks and there are pairs for it
        bool lean;          // every chunk runs to its end, the next entry bounds the last one, no record can reach `len`,
                            // and the records fit one round of the list
        bool pairmode;      // which list layout its single round uses
    };
    auto ent_of = [&](uint32_t j) { return cfirst + j < nchunks ? s_ent[j] : DEAD; }; // (no entry behind the last chunk)
    // (e0, enext: the entries of the wave's first chunk and of the chunk behind its last)
    auto range_from = [&](uint32_t w, uint32_t e0, uint32_t enext) {
        constexpr uint32_t STAGE = ROWS_CH * CHUNK6 + 64 + 16; // what a wave may touch: its chunks + the reach of a record 32 bytes past them
        const uint32_t c0 = cfirst + w * ROWS_CH;
        const bool inner = c0 + ROWS_CH < nchunks && (enext & 255u) != DEAD && (enext >> 8) <= nrec;
        // records that start in the wave's chunks: [I0, Iend); nrec is even (two records per 32 columns)
        const uint32_t I0 = e0 >> 8;
        const uint32_t Iend = min(nrec, (c0 + ROWS_CH < nchunks) ? (enext >> 8) : nrec);
        Range6 g;
        g.R0 = (I0 + 1u) & ~1u;
        g.R1 = (Iend + 1u) & ~1u;
        g.live = c0 < nchunks && (e0 & 255u) != DEAD && g.R0 < g.R1;
        g.N = g.live ? g.R1 - g.R0 : 0u;
        g.lean = g.live && inner && g.N <= ROWS_CAP && c0 * CHUNK6 + STAGE < len;
        g.pairmode = g.N > ROWS_CAP / 2u;
        return g;
    };
    auto range_of = [&](uint32_t w) { return range_from(w, ent_of(w * ROWS_CH), ent_of(w * ROWS_CH + ROWS_CH)); };
    if (wave == DEC_T / 64u - 1u) {
        const size_t lme = static_cast<size_t>(f) * smax + seg;
        uint64_t *const res = look.res + lme; // mine; res[-k]: k segments before me
        uint32_t spins = 0, w = 0;
        bool lost = false;
        // Entry phase of chunk j (lane j; lane cnt: of the chunk behind the segment): the map of the chunk in front
        // of it almost always sends all 17 phases to ONE exit (a wrong chain reads payload bytes as headers and falls
        // onto the true one within a few hundred bytes), so every lane knows its own at once; a lane whose map is
        // not unanimous takes the entry of the chunk in front through that map.
        uint32_t myp = DEAD;
        bool known = false;
        if (lane <= cnt) {
            if (lane == 0u && seg == 0u) {
                myp = 0u; // the stream starts with a record at byte 0 (RawData_Legacy.cpp:476)
                known = true;
            } else {
                const uint32_t xs = s_exits[lane]; // (chunk k = lane is the one in front of chunk j = lane)
                myp = xs ? static_cast<uint32_t>(__builtin_ctz(xs)) : DEAD;
                known = (xs & (xs - 1u)) == 0u; // one exit for all 17 phases
            }
        } else {
            known = true;
        }
        auto propagate = [&]() { // (rare; at most cnt rounds)
            for (;;) {
                const uint32_t pp = wave_prev(myp, DEAD);
                const uint32_t kk = wave_prev(known ? 1u : 0u, 0u); // (DPP moves across the wave: no LDS round trip)
                const bool pk = lane != 0u && kk != 0u;
                const bool now = !known && pk;
                if (now) {
                    myp = pp == DEAD ? DEAD : s_cx[lane * PHASES6 + pp];
                    known = true;
                }
                if (__ballot(now) == 0ull)
                    break;
            }
        };
        propagate();
        // What the NEXT segment is entered at, said as early as it can be said: a phase when one of my chunks' maps is
        // unanimous (whatever I am entered at myself), else -- a stream whose every chunk keeps its 17 chains apart --
        // the map from my entry phase to it, for my successors to go through while I still wait for mine.
        const bool full = cnt == DEC_CH;
        const bool exit_known = wave_lane(known ? 1u : 0u, DEC_CH) != 0u;
        if (full && exit_known) {
            const uint32_t xp = wave_lane(myp, DEC_CH);
            if (lane == 0)
                look_put(look.ex + lme, epoch, (EX_PHASE << 30) | xp);
        } else if (full) {
            uint32_t x = lane < PHASES6 ? lane : DEAD;
            for (uint32_t k = 1; k <= DEC_CH; k++)
                x = x == DEAD ? DEAD : s_cx[k * PHASES6 + x];
            uint32_t hw = 0;
#pragma unroll
            for (uint32_t e = 0; e < 6u; e++) {
                const uint32_t src = lane * 6u + e;
                const uint32_t xe = __shfl(x, static_cast<int>(src & 63u), 64);
                hw |= (src < PHASES6 ? xe : DEAD) << (5u * e);
            }
            if (lane < 3u)
                look_put(look.hm + 3u * lme + lane, epoch, hw);
            if (lane == 0)
                look_put(look.ex + lme, epoch, EX_MAP << 30);
        }
        if (!__builtin_amdgcn_readfirstlane(known ? 1u : 0u)) {
            // Lane 0 does not know the segment's own entry: the nearest segment in front whose exit is known -- it has
            // resolved and published its records, or said a phase above --, and from there through the maps of the
            // segments in between (64 segments per poll; the frame itself is entered at phase 0).
            uint32_t ep = DEAD;
            bool got = false;
            while (!got && !lost) {
                const int32_t k = static_cast<int32_t>(seg) - 1 - static_cast<int32_t>(lane); // segment of this lane
                bool kn = false, mp = false;
                uint32_t ph = DEAD, h0 = 0, h1 = 0, h2 = 0;
                if (k == -1) {
                    kn = true;
                    ph = 0u;
                } else if (k >= 0) {
                    uint32_t vr = 0, vx = 0;
                    const size_t ks = lme - seg + static_cast<uint32_t>(k);
                    const bool rok = look_get(look.res + ks, epoch, &vr) && (vr >> 30) != 0u;
                    const bool xok = look_get(look.ex + ks, epoch, &vx);
                    if (rok) {
                        kn = true;
                        ph = vr & 31u;
                    } else if (xok && (vx >> 30) == EX_PHASE) {
                        kn = true;
                        ph = vx & 31u;
                    } else if (xok && (vx >> 30) == EX_MAP) {
                        const bool m0 = look_get(look.hm + 3u * ks, epoch, &h0), m1 = look_get(look.hm + 3u * ks + 1u, epoch, &h1),
                                   m2 = look_get(look.hm + 3u * ks + 2u, epoch, &h2);
                        mp = m0 && m1 && m2;
                    }
                }
                const uint64_t km = __ballot(kn), mm = __ballot(mp);
                const uint32_t i0 = km ? static_cast<uint32_t>(__builtin_ctzll(km)) : 64u;
                const uint64_t front = i0 >= 64u ? ~0ull : (1ull << i0) - 1ull; // the segments between that one and me
                if (i0 < 64u && (mm & front) == front) {
                    uint32_t p = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ph), static_cast<int>(i0)));
                    for (uint32_t t = i0; t-- > 0u && p != DEAD;) {
                        const uint32_t w0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(h0), static_cast<int>(t)));
                        const uint32_t w1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(h1), static_cast<int>(t)));
                        const uint32_t w2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(h2), static_cast<int>(t)));
                        const uint32_t hw = p < 6u ? w0 : p < 12u ? w1 : w2;
                        p = (hw >> (5u * (p % 6u))) & 31u;
                    }
                    ep = p;
                    got = true;
                } else {
                    if (++spins > SPIN6)
                        lost = true;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (lane == 0u) {
                myp = lost ? DEAD : ep;
                known = true;
            }
            propagate();
        }
        const uint32_t lastp = full ? wave_lane(myp, DEC_CH) : DEAD; // the segment's exit phase: what the next one is entered at
        K6_STAMP(10, 256);

        // ---- the true chain, a quarter chunk per lane (chunk j = lane / 4, quarter r = lane % 4): where it crosses into
        // the quarter (its chunk's entry through the quarters in front), then how many records it starts there
        const uint32_t uj = lane >> 2, ur = lane & 3u;
        const uint32_t cph = __shfl(myp, static_cast<int>(uj), 64);             // entry phase of my chunk
        const uint32_t aph = wave_lane(myp, DEC_CH);                            // ... of the chunk behind a full segment
        uint32_t qp = cph;
        {
            const uint8_t *qx = s_qx + (uj + 1u) * 4u * PHASES6;
#pragma unroll
            for (uint32_t r = 0; r < 3u; r++)
                if (r < ur)
                    qp = qp == DEAD ? DEAD : qx[r * PHASES6 + qp];
        }
        if (uj >= cnt)
            qp = DEAD;
        uint32_t qn = 0;
        // ... and notes where: a lane finds one record per step until its walk is over, so its n-th record is the one of
        // step n -- its half position goes to byte n of eight registers (steps are unrolled: static indices).  The record
        // lists below are made from these notes, without a second walk along the chain.
        constexpr uint32_t NOTES6 = 32;
        uint32_t nb[NOTES6 / 4u] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        bool notes_ok = s_runs == 0u;
        {
            const uint32_t qb = (uj + 1u) * TABQ + ur * QTAB;
            const uint32_t qe = qb + HALF6 / 4u;  // a walk that has left its quarter stays where it is
            uint32_t A = qp == DEAD ? qe : qb + qp; // (DEAD: nothing to count)
            uint32_t t;
            if (s_runs) { // (some walk of this segment may meet a jump over sixteen records)
                do {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        t = s_tab[A]; t = A < qe ? t : 0u; // (unconditional read, then the select)
                        qn += t == RUN6 ? RUN6 : (t ? 1u : 0u);
                        A += t;
                    }
                } while (__any(t != 0u));
            } else {
                bool more = true;
#pragma unroll
                for (uint32_t st = 0; st < NOTES6; st += 2u) {
#pragma unroll
                    for (uint32_t u = 0; u < 2u; u++) {
                        t = s_tab[A]; t = A < qe ? t : 0u; // (unconditional read, then the select)
                        nb[(st + u) >> 2] |= (A - qb) << (8u * ((st + u) & 3u)); // (< 256; bytes behind a lane's last record are never used)
                        qn += t ? 1u : 0u;
                        A += t;
                    }
                    if (!__any(t != 0u)) {
                        more = false;
                        break;
                    }
                }
                if (more) { // records of less than 8 bytes on average: counted on, listed by a walk of their own below
                    notes_ok = false;
                    do {
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            t = s_tab[A]; t = A < qe ? t : 0u; // (unconditional read, then the select)
                            qn += t ? 1u : 0u;
                            A += t;
                        }
                    } while (__any(t != 0u));
                }
            }
        }
        K6_STAMP(11, 256);
        // records in front of my quarter within the segment, and those of the whole segment
        uint32_t total; // (quarters behind the stream add nothing)
        const uint32_t qfirst = wave_excl_scan(qn, lane, &total);
        // first record index: records of the frame's earlier segments, 64 of them per poll
        uint32_t base = 0;
        if (seg) {
#ifdef MCRAW_INJECT_LOST
            if (lane == 0 && !(f == 0u && seg == 3u))
#else
            if (lane == 0)
#endif
                look_put(res, epoch, (RES_AGG << 30) | (total << 5) | lastp);
            int32_t jn = static_cast<int32_t>(seg) - 1; // nearest segment of the window (lane 0)
            while (!lost) {
                const int32_t k = jn - static_cast<int32_t>(lane);
                w = RES_AGG << 30; // segments "before the frame": nothing, and never reached (segment 0 has a prefix)
                const bool ok = k < 0 || (look_get(res - seg + k, epoch, &w) && (w >> 30) != 0u);
                const uint64_t okm = __ballot(ok), pm = __ballot(ok && (w >> 30) == RES_PREFIX);
                const uint32_t np = pm ? static_cast<uint32_t>(__builtin_ctzll(pm)) : 64u; // lanes in front of the first prefix
                const uint64_t need = np >= 64u ? ~0ull : ((1ull << np) | ((1ull << np) - 1ull));
                if ((okm & need) != need) { // not all published yet
                    if (++spins > SPIN6)
                        lost = true;
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t add;
                (void)wave_excl_scan(lane <= np ? (w >> 5) & 0xFFFFFFu : 0u, lane, &add);
                base = min(base + add, 0xFFFFFFu);
                if (np < 64u)
                    break;
                jn -= 64;
            }
        }
        K6_STAMP(12, 256);
        K6_COUNT(14, 1);
        K6_COUNT(15, spins);
        if (lost) { // (a predecessor never published: fail the frame rather than wait for ever)
            if (lane == 0)
                atomicOr(P->status, MCRAW_E_DEVICE);
            qp = DEAD;
        }
        // entries carry the record index in 24 bits: a stream with more records than that (the frame itself has
        // fewer, the host checks) saturates instead of wrapping back into the frame
        const uint32_t endn = min(base + total, 0xFFFFFFu);
#ifdef MCRAW_INJECT_LOST
        if (lane == 0 && !(f == 0u && seg == 3u))
#else
        if (lane == 0)
#endif
            look_put(res, epoch, (RES_PREFIX << 30) | (endn << 5) | (lost ? DEAD : lastp));
        const uint32_t qi = min(base + qfirst, 0xFFFFFFu); // index of my quarter's first record
        const uint32_t ent4v = qp | (qi << 8);
        const uint32_t entv = uj <= cnt && !lost ? (cph | (qi << 8)) : DEAD;            // (lanes with ur == 0: chunk uj's)
        const uint32_t ent16 = cnt == DEC_CH && !lost ? (aph | (endn << 8)) : DEAD;     // ... and of the chunk behind a full segment
        s_ent4[lane] = ent4v;
        if (ur == 0u)
            s_ent[uj] = entv;
        if (lane == 0)
            s_ent[DEC_CH] = ent16;
        // fewer records than height * recs_per_row inside `len`: the reference would
        // skip the rest and return stale rows (RawData_Legacy.cpp:387-388)
        if (lane == 0 && cfirst + cnt >= nchunks && endn < nrec && !lost)
            atomicOr(P->status, MCRAW_E_TRUNCATED);

        // ---- the record lists of the four unpacking waves, when every one of them is on the lean path (everywhere
        // but at the ends of a frame and in runs of tiny records): lane = (wave w, chunk j, quarter r) lists a quarter
        // chunk from where the true chain crosses into it -- 18 dependent steps instead of a chunk's 70, on ONE wave
        // (a walk costs a wave its issue slots whatever the number of walking lanes).  The walk tables are dead by now:
        // their LDS holds the lists.
        // (what this wave needs of the entries it has just stored, it takes from its registers: a wait for the LDS here would
        // also wait for the look-back words' way to memory -- stores count on the same counter)
        const uint32_t uw = lane / (4u * ROWS_CH), j = (lane >> 2) & (ROWS_CH - 1u), r = lane & 3u;
        static_assert(ROWS_CH == 4, "an unpacking wave's chunks are sixteen lanes of this wave");
        const uint32_t ew0 = wave_lane(entv, 0u), ew1 = wave_lane(entv, 16u), ew2 = wave_lane(entv, 32u), ew3 = wave_lane(entv, 48u);
        auto ent_reg = [&](uint32_t w) { // entry of chunk w * ROWS_CH, as ent_of() will read it
            const uint32_t v = w == 0u ? ew0 : w == 1u ? ew1 : w == 2u ? ew2 : w == 3u ? ew3 : ent16;
            return cfirst + w * ROWS_CH < nchunks ? v : DEAD;
        };
        const uint32_t enext_reg = ent_reg(uw + 1u);
        const Range6 rg = range_from(uw, ent_reg(uw), enext_reg);
        const bool coop = __ballot(!rg.lean) == 0ull;
        if (lane == 0)
            s_coop = coop ? 1u : 0u;
        // The common case (no jumps over runs of 2-byte records in this segment, at most NOTES6 records per quarter, every
        // record listed): the lists are the notes of the count walk put in their places -- stores that do not depend on one
        // another, instead of a second walk along the chain (which took 5 700 of a workgroup's 34 000 cycles).  A lane stores
        // whole groups of eight entries.  The group its records end in is filled up with the first records of the NEXT quarter
        // (every quarter starts at least seven: a record has at most 34 bytes), taken from the lane behind it: what a lane
        // stores beyond its own records is then exactly what that lane stores there itself, and it does not matter which
        // of the two stores comes last.
        const bool noted = coop && notes_ok && __ballot(rg.pairmode || rg.N + NOTES6 > ROWS_CAP / 2u) == 0ull;
        if (K6_ABL != 3 && noted) {
            const int32_t slot0 = static_cast<int32_t>(qi - rg.R0); // -1: an odd first record belongs to the previous wave's
            uint16_t *lp = s_pos[uw] + slot0;                        // last pair, never listed
            const uint32_t boff = j * CHUNK6 + r * (CHUNK6 / 4u);
            {
                const uint32_t kb = qn & 7u, gp = qn >> 3;
                // the next quarter's first eight notes, 128 half positions further on (lane 63: the chunk behind the segment
                // is entered at phase `aph`)
                uint32_t x0 = wave_next(nb[0], 0u), x1 = wave_next(nb[1], 0u);
                x0 = (lane == 63u ? aph & 31u : x0) | 0x80808080u;
                x1 |= 0x80808080u;
                const uint64_t nx = (static_cast<uint64_t>(x1) << 32) | x0;
                const uint32_t o0 = gp == 0u ? nb[0] : gp == 1u ? nb[2] : gp == 2u ? nb[4] : nb[6];
                const uint32_t o1 = gp == 0u ? nb[1] : gp == 1u ? nb[3] : gp == 2u ? nb[5] : nb[7];
                const uint64_t own = (static_cast<uint64_t>(o1) << 32) | o0;
                const uint64_t keep = (1ull << (8u * kb)) - 1ull; // (kb <= 7)
                const uint64_t m = (own & keep) | (nx << (8u * kb));
#pragma unroll
                for (uint32_t g = 0; g < NOTES6 / 8u; g++) {
                    nb[2u * g] = gp == g ? static_cast<uint32_t>(m) : nb[2u * g];
                    nb[2u * g + 1u] = gp == g ? static_cast<uint32_t>(m >> 32) : nb[2u * g + 1u];
                }
            }
#pragma unroll
            for (uint32_t g = 0; g < NOTES6 / 8u; g++) {
                if (g && __ballot(qn > 8u * g) == 0ull)
                    break;
                if (qn > 8u * g) { // (at most seven entries behind a lane's last record: the next quarter's first seven)
#pragma unroll
                    for (uint32_t i = 8u * g; i < 8u * g + 8u; i++) {
                        const uint32_t v = boff + 2u * ((nb[i >> 2] >> (8u * (i & 3u))) & 255u);
                        if (i > 0u || slot0 >= 0)
                            lp[i] = static_cast<uint16_t>(v);
                    }
                }
            }
            // behind a wave's last record: the next wave's first one (the partner of the wave's last record when its range
            // ends on an even one; a lane whose records end on a group boundary has stored nothing behind them)
            if (j == ROWS_CH - 1u && r == 3u)
                lp[qn] = static_cast<uint16_t>(ROWS_CH * CHUNK6 + 2u * (enext_reg & 255u));
        } else if (K6_ABL != 3 && K6_ABL < 5 && coop) {
            const uint32_t ej = ent4v, first = rg.R0;
            const uint8_t *base = s_own + uw * (ROWS_CH * CHUNK6);
            const uint8_t *p = base + j * CHUNK6 + r * (CHUNK6 / 4u) + 2u * (ej & 255u);
            const uint8_t *const pe = base + j * CHUNK6 + (r + 1u) * (CHUNK6 / 4u);
            uint32_t idx = ej >> 8;
            if (rg.pairmode) {
                // an odd first record belongs to the previous wave's last pair: never listed
                while (p < pe) { // the stride decode, and a store for every second record
                    // eight 2-byte records fill an aligned 16-byte line whose even bytes all have a zero high nibble:
                    // one LDS read then lists four pairs instead of one record
                    if (((p - base) & 15u) == 0u && (idx & 1u) == 0u && idx >= first && p + 16 <= pe) {
                        const uint4 q = *reinterpret_cast<const uint4 *>(p);
                        if (((q.x | q.y | q.z | q.w) & 0x00F000F0u) == 0u) {
#pragma unroll
                            for (uint32_t k = 0; k < 4u; k++)
                                s_pos[uw][((idx - first) >> 1) + k] = static_cast<uint16_t>(p - base + 4u * k);
                            p += 16;
                            idx += 8u;
                            continue;
                        }
                    }
                    const uint32_t hb = static_cast<uint32_t>(*p) >> 4;
                    if ((idx & 1u) == 0u && idx >= first)
                        s_pos[uw][(idx - first) >> 1] = static_cast<uint16_t>(p - base);
                    idx++;
                    p += 2u + len6_of(hb);
                }
            } else {
                uint16_t *lp = s_pos[uw] + static_cast<int32_t>(idx - first); // [-1] for an odd first record:
                if (idx < first && p < pe) {                                   // skipped, see above
                    p += 2u + len6_of(static_cast<uint32_t>(*p) >> 4);
                    lp++;
                }
                while (p < pe) { // nothing but the stride decode in the loop
                    const uint32_t hb = static_cast<uint32_t>(*p) >> 4;
                    *lp++ = static_cast<uint16_t>(p - base);
                    p += 2u + len6_of(hb);
                }
                // the walk of a wave's last quarter stops on the next wave's first record: the partner of my
                // last record when my range ends on an even one (lp is then at an odd list index)
                if (j == ROWS_CH - 1u && r == 3u && ((lp - s_pos[uw]) & 1))
                    *lp = static_cast<uint16_t>(p - base);
            }
        }
        K6_STAMP(13, 256);
    }
    __syncthreads();
    K6_STAMP(3, 0);
    const bool coop = s_coop != 0u;
    if ((wave >= 4u && !coop) || K6_ABL >= 5) {
        K6_END();
        return; // (the fifth wave has no chunks of its own; on the lean path it takes a share of every wave's pairs)
    }

    const Range6 mine = range_of(wave & 3u);
    const uint32_t c0 = cfirst + wave * ROWS_CH, cs0 = c0 * CHUNK6;
    const uint32_t R0 = mine.R0, R1 = mine.R1, N = mine.N;
    const bool live = mine.live, pairmode = mine.pairmode;
    // entries of my chunks (lane j) and of the chunk after them (lane ROWS_CH): the general path walks from them
    const uint32_t e = lane <= ROWS_CH ? ent_of((wave & 3u) * ROWS_CH + lane) : DEAD;

    // (the walk tables are dead: from here on their LDS holds the record lists)
    const uint32_t width = static_cast<uint32_t>(P->width);
    const bool fast = P->fast_store != 0u;
    uint16_t *const out = P->out;

    // row arithmetic without per-lane division: pairs per row `ppr`; a round spans < 2 rows when
    // ppr >= 512, otherwise n / ppr for n < 1024 is exact as (n * ceil(2^20 / ppr)) >> 20
    const uint32_t ppr = P->recs_per_row >> 1;
    const bool widerow = ppr >= 512u;
    const uint32_t m20 = widerow ? 0u : ((1u << 20) + ppr - 1u) / ppr;

    // Unpack the pairs of records [wlo, whi) (both even) of unpacking wave uw, listed in s_pos[uw]: four lanes per pair; of
    // its 2 * (whi - wlo) tasks, those in [tb, te)
    auto unpack_round = [&](uint32_t uw, uint32_t wlo, uint32_t whi, bool by_pair, uint32_t tb, uint32_t te) {
        const uint8_t *bytes = s_own + uw * (ROWS_CH * CHUNK6);
        const uint32_t pair0 = wlo >> 1;
        const uint16_t *pairs = s_pos[uw];
        const uint32_t *both = reinterpret_cast<const uint32_t *>(s_pos[uw]);
        const uint32_t y0 = pair0 / ppr, r0 = pair0 - y0 * ppr; // wave-uniform
        const uint32_t row0 = y0 * width;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
        for (uint32_t t = tb + lane; t < te; t += 64u) {
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
            const uint32_t dy = widerow ? (n >= ppr ? 1u : 0u) : mul_u24(n, m20) >> 20;
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
                __builtin_nontemporal_store(v, gptr<u32x4>(px));
            } else if (x + 8u <= width) { // rows off the 16-byte grid: still one (unaligned) 16-byte store
                typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
                const u32x4_u v = {o[0], o[1], o[2], o[3]};
                *gptr<u32x4_u>(px) = v;
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 8u; j++) // padded columns are cropped (RawData_Legacy.cpp:490)
                    if (x + j < width)
                        gptr<uint16_t>(px)[j] = static_cast<uint16_t>(o[j >> 1] >> (16u * (j & 1u)));
            }
        }
    };

    if (coop) {
        // The wave that resolved the entries has no chunks of its own: it takes the last TAIL6 tasks of each of the four
        // lists (whole passes of 64 lanes: a wave's 580 tasks were nine passes and a tenth of 8 lanes; now seven or eight,
        // and eight for the fifth wave).
        K6_STAMP(4, 0);
        auto tail_of = [&](uint32_t ntask) { return ntask >= 3u * TAIL6 ? ntask - TAIL6 : ntask; }; // first task of the fifth wave's share
        if (wave < 4u) {
            unpack_round(wave, R0, R1, pairmode, 0u, tail_of(2u * (R1 - R0)));
        } else {
#pragma unroll 1
            for (uint32_t uw = 0; uw < 4u; uw++) {
                const Range6 rg = range_of(uw);
                const uint32_t ntask = 2u * (rg.R1 - rg.R0);
                unpack_round(uw, rg.R0, rg.R1, rg.pairmode, tail_of(ntask), ntask);
            }
        }
        K6_STAMP(5, 0);
#ifdef MCRAW_DIAG
        if (threadIdx.x == 0 && blockIdx.x < K6_PROF_WG) {
            g_k6_prof[blockIdx.x][7] = static_cast<uint32_t>(stamp_);
            g_k6_prof[blockIdx.x][17] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
        }
        __builtin_amdgcn_s_waitcnt(0); // (the end stamp is taken when the wave's stores have landed)
        K6_END();
#endif
        return;
    }
    if (!live)
        return;
    const uint8_t *bytes = s_own + wave * (ROWS_CH * CHUNK6);
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
        unpack_round(wave, wlo, whi, true, 0u, 2u * (whi - wlo));
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier(); // the list is rewritten by the next round
    }
}

// ------------------------------------------------------------------ launchers

#ifdef MCRAW_DIAG
extern "C" int mcraw_diag_k6_occupancy(int dyn_lds)
{
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k6_decode<0>, DEC_T, static_cast<size_t>(dyn_lds));
    return n;
}
extern "C" void mcraw_diag_k6_prof(uint32_t *out, int nwg, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k6_prof), sizeof(uint32_t) * K6_PROF_N * nwg);
    if (reset) {
        void *p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_k6_prof));
        (void)hipMemset(p, 0, sizeof(g_k6_prof));
    }
}
#endif

// wg_tab (3 * nframes + 1 words, see k6_decode): [0, nframes]: first workgroup of every stage; then the stages' first
// rounds; then the frames by falling number of segments.  `stage0`: workgroups of the rounds every frame takes part in
// (= wg_tab[1]); `nwg`: segments of all frames together.
void launch_k6_decode(const Plan6 *plans, const uint32_t *wg_tab, uint32_t stage0, uint32_t nwg, const Look6 &look,
                      uint32_t *tickets, uint32_t epoch, int nframes, uint32_t smax, const Post &post, hipStream_t st)
{
    if (!nwg)
        return;
    const dim3 grid(nwg), block(DEC_T);
    const uint32_t nf = static_cast<uint32_t>(nframes);
    if (post.mode == 0u) {
#ifdef MCRAW_DIAG // occupancy experiments: extra LDS per workgroup (tools/k6_occ.sh)
        static const uint32_t pad = getenv("MCRAW_K6_LDSPAD") ? static_cast<uint32_t>(atoi(getenv("MCRAW_K6_LDSPAD"))) : 0u;
        hipLaunchKernelGGL(k6_decode<0>, grid, block, pad, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post);
#else
        hipLaunchKernelGGL(k6_decode<0>, grid, block, 0, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post);
#endif
        return;
    }
    switch (post_bits(post.mode)) { // one kernel instance per row format
    case 12: hipLaunchKernelGGL(k6_decode<12>, grid, block, 0, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post); break;
    case 10: hipLaunchKernelGGL(k6_decode<10>, grid, block, 0, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post); break;
    case 14: hipLaunchKernelGGL(k6_decode<14>, grid, block, 0, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post); break;
    default: hipLaunchKernelGGL(k6_decode<16>, grid, block, 0, st, plans, wg_tab, stage0, look, tickets, epoch, nf, smax, post); break;
    }
}

} // namespace mcraw
