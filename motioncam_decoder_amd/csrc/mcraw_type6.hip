// mcraw_type6.hip -- gfx950 kernels for the legacy MCRAW frame encoding
// (compressionType 6).  Replaces motioncam::raw::DecodeLegacy,
// lib/RawData_Legacy.cpp:445-495.
//
// The legacy stream is ONE chain of 16-sample records with inline 2-byte
// headers (RawData_Legacy.cpp:377-442): record i+1 starts where record i ends.
// A frame holds ~w*h/16 of them.  Record strides are even and <= 34 bytes, so a chain
// can cross any given byte of the stream at only 17 offsets ("phases" 0,2,..,32), and a chain that
// starts on a byte that is no record start reads payload bytes as headers and falls onto the true
// chain within a few hundred bytes.
//
// One kernel, one pass over the stream (k6_decode): a workgroup stages 16 KiB of it (and the KiB in front),
// lets 64 lanes walk it speculatively -- a quarter KiB each, from a start half a KiB further up --, verifies the
// lanes against each other and against the one chain that the 17 possible chains of the KiB in front have become,
// gets the index of its first record by decoupled look-back over the frame's earlier workgroups, lists the records
// and unpacks them (MSB-first bitstreams, RawData_Legacy.cpp:38-370), adds the references, interleaves even/odd
// columns (:483-486) and crops the padded row (:490).
#include "mcraw_dev.h"

#include <cstdlib>

#include "../../include/mcraw_hip.h"

#include <type_traits>

namespace mcraw {

constexpr uint32_t DEAD = 31; // phase value: the chain ended (a record crossed `len`)

// Payload bytes of a record whose header nibble is `b` (RawData_Legacy.cpp:13-32).
__device__ __forceinline__ uint32_t len6_of(uint32_t b) { return b <= 10u ? 2u * b : 32u; }

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

// A lane's record PAIR (round 4): what is computed once per record is computed for both records at once.
//   ha, hb : the two headers (records start on even bytes: 16-bit reads), even-column record first
//   refs   : both 12-bit references, packed like the samples (RawData_Legacy.cpp:372-375: big-endian, low nibble of byte 0 first)
struct Pair6 {
    uint32_t sa, sb;   // field widths
    uint32_t refs;
};
__device__ __forceinline__ Pair6 pair6_head(uint32_t ha, uint32_t hb)
{
    Pair6 r;
    const uint32_t na = __builtin_amdgcn_ubfe(ha, 4u, 4u), nb = __builtin_amdgcn_ubfe(hb, 4u, 4u);
    r.sa = na <= 10u ? na : 16u;
    r.sb = nb <= 10u ? nb : 16u;
    const uint32_t hh = ha | (hb << 16);
    r.refs = __builtin_amdgcn_perm(hh, hh, 0x02030001u) & 0x0FFF0FFFu; // bytes swapped inside each half
    return r;
}
// Samples 4*qt..4*qt+3 (qt4 = 4 * qt) of the record at byte `ro` of the staged stream, field width `s`, reference NOT yet
// added.  The payload is an MSB-first bitstream of s-bit fields (s = header nibble for <= 10, 16 big-endian raw bits above,
// RawData_Legacy.cpp:38-370), so the lane's four fields are the top 4*s bits of a 64-bit big-endian window that starts
// 4*qt*s bits into the payload.
__device__ __forceinline__ void quad6u(const uint8_t *__restrict__ bytes, uint32_t ro, uint32_t qt4, uint32_t s, uint32_t v[4])
{
    const uint32_t ob = mul_u24(qt4, s);                                               // bit offset of my fields in the payload
    // the window: three aligned dwords, cut to the byte it starts on and byte-swapped by two permutes (one selector does both).
    // (gfx950 reads LDS at any alignment, and ONE 8-byte read at the window's first byte holds all four fields -- but such reads
    // stall the LDS pipeline: the kernel 0.383 instead of 0.362 ms, docs/lab_notes.md)
    const uint32_t B = ro + 2u + (ob >> 3);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(bytes + (B & ~3u));
    const uint32_t k = B & 3u;
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    const uint32_t sel = 0x00010203u + __builtin_amdgcn_perm(k, k, 0u); // k in every byte lane
    const uint32_t hi = __builtin_amdgcn_perm(d1, d0, sel), lo = __builtin_amdgcn_perm(d2, d1, sel);
    const uint64_t W = (static_cast<uint64_t>(hi) << 32) | lo; // big-endian window; my fields start (ob & 4) bits below its top
    // field i ends p_i = 64 - (ob & 4) - (i + 1) * s bits above the window's bottom: p_0, p_1 >= 32 for every width
    const uint32_t p0 = (64u - s) - (ob & 4u), p1 = p0 - s, p2 = p1 - s, p3 = p2 - s;
    v[0] = __builtin_amdgcn_ubfe(hi, p0 & 31u, s);
    v[1] = __builtin_amdgcn_ubfe(hi, p1 & 31u, s);
    v[2] = __builtin_amdgcn_ubfe(static_cast<uint32_t>(W >> (p2 & 63u)), 0u, s);
    v[3] = __builtin_amdgcn_ubfe(static_cast<uint32_t>(W >> (p3 & 63u)), 0u, s);
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
// the even record's header -- so a round covers 768 records in 772 bytes of LDS: all data but runs of records of 2 and 4
// bytes (flat or clipped regions; 1-bit residuals of a nearly flat frame: 1024 records per wave) stays on the single-round path.
// (Rounds 4-5: 1024 records in 1 KiB; round 6 gave a quarter of the lists for the eighth workgroup per CU, see k6_decode.)
constexpr uint32_t ROWS_CAP = 192u * ROWS_CH; // records per round (typical: ~70 per chunk; worst case 512 per chunk -> 3 rounds)
static_assert(ROWS_CAP <= 1024u && ROWS_CAP % 2u == 0u, "the division-free row arithmetic of the unpack assumes at most 512 pairs per round");

#ifndef K6_ABL
#define K6_ABL 0 // timing experiments only (wrong pixels): 1 no stores, 2 no unpack, 3 no record lists, 4 load + stage only, 8 no ticket,
                 // 16 forty vector instructions more per unpack pass, 32 no look-back (record indices estimated), 33 one poll, not waited on
#endif

// ------------------------------------------------------------------ k6_decode
constexpr uint32_t UNPACK_W = SEG_WAVES6;         // unpacking waves per workgroup
constexpr uint32_t DEC_CH = UNPACK_W * ROWS_CH;   // chunks per workgroup
// Four waves per workgroup: the last wave resolves the chain (the one in front of it the sure entry beside it), then all four unpack
// their chunks.  Eight workgroups fit a CU (round 6; seven until then): the kernel is a chain of latencies -- load, walk, look-back, lists, unpack --, and what
// hides them is the number of segments in flight.  (Rounds 2-5 also built two-, five- and eight-wave workgroups and a form that
// stages the stream by LDS-DMA, all slower: docs/lab_notes.md.)
constexpr uint32_t DEC_T = 64 * UNPACK_W;
static_assert(DEC_T == 256, "four waves: the resolving wave is the last one");
constexpr uint32_t FRONT6 = CHUNK6;       // bytes staged in front of the segment: the chains that cross into them have become one by the segment's start
#ifndef MCRAW_WARM6
#define MCRAW_WARM6 512
#endif
constexpr uint32_t WARM6 = MCRAW_WARM6;   // bytes in front of its quarter chunk at which a speculative walker starts (tools/k6_warm.sh)
constexpr uint32_t QUART6 = CHUNK6 / 4;   // bytes of stream per walker
constexpr uint32_t NQ6 = 4 * DEC_CH;      // walkers = quarter chunks per segment
constexpr uint32_t NOFRONT = 255;         // s_front's boundary: the chains never became one inside this segment
static_assert(NQ6 == 64, "one walker per lane of the resolving wave");
static_assert(WARM6 <= FRONT6 && WARM6 % 2u == 0u && WARM6 >= 64u, "the walkers of the first quarter start inside the staged front");

// Bytes from a record's header to the next record's (RawData_Legacy.cpp:13-32,377-442); `b` = the header's first byte.
__device__ __forceinline__ uint32_t stride6(uint32_t b)
{
    const uint32_t hb = b >> 4;
    return hb <= 10u ? 2u * hb + 2u : 34u;
}

// Where the record behind the one at byte P of the staged stream starts.  What can be computed from P alone (P + 2, P + 34) is
// computed while the header byte is on its way from the LDS: behind the read the dependent chain is three instructions
// (nibble, shift-add, select).
__device__ __forceinline__ uint32_t next6(const uint8_t *st, uint32_t P)
{
    const uint32_t b = st[P];
    uint32_t big = P + 34u, small = P + 2u;
    asm volatile("" : "+v"(big), "+v"(small)); // (computed here, in front of the wait for the byte)
    uint32_t n; // (both values exist: the select below stays a select -- as a branch over two arms it cost the walk half its speed)
    asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(n) : "v"(__builtin_amdgcn_ubfe(b, 4u, 4u)), "v"(small));
    return b >= 0xB0u ? big : n;
}

// One step of a chain walk over the staged stream, the careful form, for segments that need it (wave-uniform): the record at
// byte P -> bytes to the next one, 0 for a walk that has reached `bound` (it stays where it arrived: P - bound is the phase at
// which the chain crosses the bound); a record that would reach the end of the stream is the chain's end
// (RawData_Legacy.cpp:387-388,398-399: stride 0, the walk stays in front of it; `limP` = the stage position of `len`), and a
// walk that stands on the first byte of a 16-byte piece made of eight 2-byte records (flat or clipped image regions: one bit
// per piece in `flat`) passes all eight at once.  *n = records passed.
__device__ __forceinline__ uint32_t step6(const uint8_t *st, const uint64_t *flat, uint32_t P, uint32_t bound, uint32_t limP, uint32_t *n)
{
    uint32_t t = stride6(st[P]);
    const bool jump = (P & 15u) == 0u && P + 16u <= bound && ((flat[P >> 10] >> ((P >> 4) & 63u)) & 1ull) != 0ull;
    t = jump ? 16u : t;
    const uint32_t c = jump ? 8u : 1u;
    t = P + t >= limP ? 0u : t;
    t = P < bound ? t : 0u;
    *n = t ? c : 0u;
    return t;
}

// All lanes of a wave walk until every walk has reached its `bound` (or has died in front of the end of the stream).
// Returns where the walk arrived: the first record start at or behind the bound (a dead walk: in front of it).
// The fast form runs free -- nothing in the dependent chain holds a walk that has arrived; it notes its arrival and walks on
// until the slowest lane is there, clamped to the stage.
template <uint32_t CLAMP>
__device__ __forceinline__ uint32_t walk_to6(bool careful, const uint8_t *st, const uint64_t *flat, uint32_t P, uint32_t bound, uint32_t limP)
{
    if (careful) {
        uint32_t t, n;
        do {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                t = step6(st, flat, P, bound, limP, &n);
                P += t;
            }
        } while (__any(t != 0u)); // (a walk that does not move never moves again)
        return P;
    }
    uint32_t arr = P;
    do {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t Pn = next6(st, P);
            arr = P < bound ? Pn : arr;
            P = Pn;
        }
        P = min(P, CLAMP);
    } while (__any(P < bound));
    return arr;
}

#ifdef MCRAW_DIAG // phase stamps of every workgroup (timing experiments only; not in the product library)
constexpr int K6_PROF_WG = 1 << 16;
// [0..13]: stage stamps (s_memtime), [14] live, [15] look-back spins; [16], [17]: s_memrealtime (100 MHz) at wave 0's start and end;
// [18 + w]: at wave w's end, its stores landed; [23], [24]: HW_ID, XCC_ID of wave 0; [25]: walk rounds of the resolving wave; [27 + w]: HW_ID of wave w
constexpr int K6_PROF_N = 32;
__device__ uint32_t g_k6_prof[K6_PROF_WG][K6_PROF_N];
// (the stamps are kept in LDS and written at the workgroup's end: a store to memory per stamp made every later wait for the wave's
// loads wait for that store's way out as well -- the phases behind a stamp read thousands of cycles too long)
#define K6_STAMP(slot, who)                                                                                            \
    do {                                                                                                               \
        if (threadIdx.x == (who)) {                                                                                    \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
            s_prof[slot] = static_cast<uint32_t>(now_ - stamp_);                                                       \
            stamp_ = now_;                                                                                             \
        }                                                                                                              \
    } while (0)
#define K6_COUNT(slot, v) ((threadIdx.x & 63u) == 0u ? (void)(s_prof[slot] = static_cast<uint32_t>(v)) : (void)0)
#else
#define K6_STAMP(slot, who)
#define K6_COUNT(slot, v)
#endif

// A workgroup works on ONE frame (the launch interleaves the frames: a segment's predecessors then started long before
// it) and takes its segment -- DEC_CH chunks -- from the frame's ticket counter: the segments it may have to wait on were all taken by
// workgroups that are running or done, whatever order the hardware starts workgroups in.  (One counter per frame,
// each in its own 256 bytes: one counter for the batch serialised the launch -- 31 000 device-scope atomics on one
// address took 0.37 ms.)
//
// Where the records of the segment start (round 4; rounds 1-3 built a table of record strides and walked all 17 phases of
// every chunk over it: 17 walks per byte of stream).  A chain that starts on a byte that is no record start reads payload
// bytes as headers and falls onto the true chain within a few hundred bytes, so:
//  * SPECULATIVE WALKERS (fifth wave, lane = quarter chunk): every lane starts WARM6 bytes in front of its quarter on the
//    stream bytes themselves (no table), notes at which phase it arrives, then walks its quarter: phase at which it leaves,
//    records started, where they start.  Nothing is taken on trust: lane j's arrival must be lane j - 1's exit; every lane for
//    which that fails walks its quarter again from where its predecessor says, all of them side by side, until nothing
//    changes -- the lowest wrong lane is right for good after each round.
//  * THE SURE ENTRY (fourth wave, 17 lanes): every chain that crosses into the KiB in front of the segment starts a record
//    at one of 17 phases there; the 17 walks almost always arrive at the segment as ONE chain -- that arrival is the
//    induction's start.  Where they do not (0.5 % of the segments of natural frames), they walk on, quarter by quarter, until
//    they are one: the lanes behind that boundary are verified from there (the segment then says its exit phase at once),
//    the ones in front of it wait for the segment in front to say at which phase it ends.  A stream whose chains never
//    meet publishes the map entry phase -> exit phase instead, and its segments resolve in wave fronts of 64.
template <int POST> // 0 = the plain mosaic, else bits per sample of the post stage's rows
__global__ __launch_bounds__(DEC_T) void k6_decode(const Plan6 *__restrict__ plans, const uint32_t *__restrict__ wg_tab,
                                                   uint32_t stage0, const Look6 look, uint32_t *__restrict__ tickets,
                                                   uint32_t epoch, uint32_t nframes, uint32_t smax, const Post post)
{
    // the segment's stream: the KiB in front of it, its DEC_CH chunks and the reach of a record that starts 32 bytes past them
    constexpr uint32_t OWN = DEC_CH * CHUNK6, SLACK = 64 + 32;
    constexpr uint32_t NPIECE = (FRONT6 + OWN + SLACK) / 16; // 16-byte pieces, piece 0 at stream offset (cfirst - 1) * CHUNK6
    constexpr uint32_t NROUND = (NPIECE + DEC_T - 1) / DEC_T;
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[FRONT6 + OWN + SLACK];
    uint8_t *const s_own = s_stage + FRONT6;
    __shared__ uint64_t s_flat[NROUND * (DEC_T / 64u)]; // one bit per staged piece: eight 2-byte records (piece i: bit i % 64 of word i / 64)
    // the list of a round, one of two layouts: every record r at [r - wlo] (up to ROWS_CAP / 2 records: the
    // common case, one LDS read gives both records of a pair), or one entry per PAIR at [(r - wlo) / 2]
    // holding the even record only (up to ROWS_CAP records; the unpacking lane finds the odd one behind it)
    typedef uint16_t PosList[ROWS_CAP / 2 + 2];
    static_assert(sizeof(PosList) % 4u == 0u, "the lists are read as dwords");
    // (round 6: 20 352 bytes of LDS = EIGHT workgroups per CU.  The first wave's list lies in the staged front, which nobody reads
    // once the chain is resolved, and a list holds 386 entries instead of 514: 3 - 9 % on natural and noisy frames, tools/ab6n.sh,
    // tools/ab_shapes6.sh -- frames made of runs of 2-byte records take three rounds per wave instead of two and lose 10 %.)
    __shared__ __attribute__((aligned(4))) PosList s_posN[UNPACK_W - 1];
    static_assert(sizeof(PosList) <= FRONT6, "the first list fits the front");
    auto pos_of = [&](uint32_t w) -> uint16_t * { return w == 0u ? reinterpret_cast<uint16_t *>(s_stage) : s_posN[w - 1u]; };
    // entry of my chunks and of the one behind them (phase | first record << 8), and of every quarter of my chunks
    __shared__ uint32_t s_ent[DEC_CH + 1], s_ent4[NQ6];
    __shared__ uint8_t s_fmap[32]; // (only for streams whose chains never meet) my entry phase -> my exit phase
    __shared__ uint32_t s_ticket, s_coop, s_front;

#ifdef MCRAW_DIAG
    unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
    __shared__ uint32_t s_prof[K6_PROF_N];
    if (threadIdx.x < K6_PROF_N)
        s_prof[threadIdx.x] = 0u;
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
    constexpr uint32_t RESOLVER = DEC_T / 64u - 1u; // the wave that resolves the chain
    // Which frame this workgroup works on, and the segment it will most likely be given: the launch goes over the
    // frames round by round -- round r = segment r of every frame that has one --, so a frame's segments start in order
    // and far apart, and no workgroup is launched for nothing however the frames' sizes differ (the host's table: the
    // frames by falling size; stage t = the rounds in which all but the t smallest are still in play).
    // (the rounds that every frame takes part in need no table: most batches hold frames of one size)
    const bool all_in = blockIdx.x < stage0;
    const uint32_t stage = all_in ? 0u : static_cast<uint32_t>(find_frame(blockIdx.x, wg_tab, static_cast<int>(nframes)));
    const uint32_t inplay = nframes - stage, wrel = all_in ? blockIdx.x : blockIdx.x - wg_tab[stage];
    const uint32_t f = all_in ? wrel % inplay : wg_tab[2u * nframes + 1u + wrel % inplay];
    const Plan6 *P = plans + f;
    const uint32_t nchunks = P->nchunks, nrec = P->nrec, len = P->len;
    const __amdgpu_buffer_rsrc_t rs = frame_rsrc(P->in, len);
    // What the unpacking waves need of the plan, read HERE: left to the compiler, these four scalar loads and the division sit behind
    // the workgroup's last barrier, on every wave's way from the record lists to its first store (worth 1 % at most: they hit the scalar cache).
    uint32_t width = static_cast<uint32_t>(P->width), fast_store = P->fast_store, ppr = P->recs_per_row >> 1;
    uint16_t *out = P->out;
    // row arithmetic without per-lane division: pairs per row `ppr`; a round spans < 2 rows when
    // ppr >= 512, otherwise n / ppr for n < 1024 is exact as (n * ceil(2^20 / ppr)) >> 20
    const bool widerow = ppr >= 512u;
    uint32_t m20 = widerow ? 0u : ((1u << 20) + ppr - 1u) / ppr;
    asm volatile("" : "+s"(width), "+s"(fast_store), "+s"(ppr), "+s"(out), "+s"(m20)); // (values, not addresses to load from later)
    // the ticket takes a round trip to the frame's counter: start loading what it will almost certainly say
    // (workgroups start in order), and load again if it says otherwise
    if (tid == 0)
        s_ticket = K6_ABL >= 8 ? 0u : atomicAdd(tickets + f * TICKET_STRIDE6, 1u);
    uint32_t seg = (all_in ? 0u : wg_tab[nframes + 1u + stage]) + wrel / inplay;
    uint4 v[NROUND];
    auto fetch = [&]() { // piece i at stream offset (seg * DEC_CH - 1) * CHUNK6 + 16 i; past `len`: reads 0
#pragma unroll
        for (uint32_t r = 0; r < NROUND; r++) {
            const uint32_t i = tid + r * DEC_T;
            // Plain loads (round 4; round 3's were non-temporal): the KiB in front of the segment is the last KiB of the segment
            // before, and the pieces a thread of the last round fetches BEHIND the stage (3 KiB: the round is not full) are the
            // next segment's first -- both are found in the XCD's L2 by the workgroup that needs them when a frame's segments run on
            // one XCD.  Counters: 1.015 x the stream's bytes fetched from memory (non-temporal: 1.23 - 1.47 x, the lines were
            // fetched again), and the fetch behind the stage is worth 2 - 4 % (tools/ab6n.sh).
            const uint32_t off = seg * OWN - FRONT6 + i * 16u;
            if (!(seg || i >= FRONT6 / 16u))
                v[r] = make_uint4(0u, 0u, 0u, 0u);
            else
                v[r] = ld_b128(rs, off);
        }
    };
    fetch();
    __syncthreads();
    K6_STAMP(0, 0);
    K6_STAMP(8, RESOLVER * 64u);
    if (K6_ABL < 8 && s_ticket != seg) {
        seg = s_ticket;
        fetch();
    }

    // ---- what the resolving wave finds out (lane = quarter chunk: chunk uj, quarter ur)
    const uint32_t uj = lane >> 2, ur = lane & 3u;
    const uint32_t qb = FRONT6 + lane * QUART6, qe = qb + QUART6;
    constexpr uint32_t NOTES6 = 32;
    uint32_t nb[NOTES6 / 4u] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}; // where my quarter's records start (half positions, a byte each)
    uint32_t a = DEAD, x = DEAD, qn = 0u; // phases at which the chain enters and leaves my quarter; records it starts there
    bool notes_ok = false, lost = false;
    uint32_t spins = 0;
    const uint32_t cfirst = seg * DEC_CH;
    if (cfirst >= nchunks) {
        return; // whole workgroup
    }
    const uint32_t cnt = min(static_cast<uint32_t>(DEC_CH), nchunks - cfirst);
    const bool full = cnt == DEC_CH;
    // stage position of the stream's end (the stage starts FRONT6 bytes in front of the segment; cfirst * CHUNK6 < len)
    const uint32_t limP = len - cfirst * CHUNK6 + FRONT6;
    {
        const bool near_end = limP < FRONT6 + OWN + SLACK + 64u; // a record this workgroup walks over can reach `len`

        // ---- stage the stream; note which pieces are eight 2-byte records in a row (a flat or clipped image region would
        // otherwise cost a step per record, 128 per quarter)
#pragma unroll
        for (uint32_t r = 0; r < NROUND; r++) {
            const uint32_t i = tid + r * DEC_T;
            bool ones = false;
            if (i < NPIECE && (seg || i >= FRONT6 / 16u)) {
                const uint4 vr = v[r];
                *reinterpret_cast<uint4 *>(s_stage + i * 16u) = vr;
                // every even byte has a zero high nibble: a walk that arrives on the piece's first byte passes eight records of
                // two bytes; none of them may be the chain's end (RawData_Legacy.cpp:387-388: the last one ends at 16 i + 16)
                ones = ((vr.x | vr.y | vr.z | vr.w) & 0x00F000F0u) == 0u && i * 16u + 16u < limP;
            }
            const unsigned long long om = __ballot(ones);
            if (lane == 0u)
                s_flat[r * (DEC_T / 64u) + wave] = om;
        }
        if (tid < 32u)
            s_fmap[tid] = static_cast<uint8_t>(DEAD);
        __syncthreads();
        K6_STAMP(1, 0);
        K6_STAMP(9, RESOLVER * 64u);
        if (K6_ABL == 4)
            return;
        // (wave-uniform: which walk loop this workgroup runs)
        const bool careful = near_end || __ballot(lane < NROUND * (DEC_T / 64u) && s_flat[lane < NROUND * (DEC_T / 64u) ? lane : 0u] != 0ull) != 0ull;
        constexpr uint32_t CLAMP = FRONT6 + OWN + SLACK - 2u; // (free-running walks stay inside the stage)

        // ---- the sure entry (wave RESOLVER - 1): the 17 chains that can cross into the staged front -- every chain starts a record
        // at one of 17 phases there --, walked to the segment's start in two halves side by side (34 lanes: half as many steps;
        // the second half's 17 walks map the first half's exits to arrivals); and, while the arrivals are not yet one chain,
        // on to the next quarter's start
        if (wave == RESOLVER - 1u) {
            __builtin_amdgcn_s_setprio(3);
            uint32_t fr = 0u; // boundary (quarters into the segment) << 8 | phase of THE chain there; segment 0 starts with a record at byte 0 (RawData_Legacy.cpp:476)
            if (seg) {
                const bool w17 = lane < PHASES6, w34 = lane < 2u * PHASES6;
                const uint32_t b0 = w17 ? FRONT6 / 2u : FRONT6;
                uint32_t Pf = w17 ? 2u * lane : w34 ? FRONT6 / 2u + 2u * (lane - PHASES6) : FRONT6 + OWN; // (the other lanes stand behind every bound)
                Pf = walk_to6<CLAMP>(careful, s_stage, s_flat, Pf, b0, limP);
                uint32_t ph = Pf >= b0 ? (Pf - b0) >> 1 : DEAD; // (a chain that has died in front of the bound stays dead)
                { // lanes 0..16: through the second half
                    const uint32_t ph2 = __shfl(ph, static_cast<int>(PHASES6 + (ph < PHASES6 ? ph : 0u)), 64);
                    ph = w17 ? (ph == DEAD ? DEAD : ph2) : DEAD;
                }
                const uint32_t a0 = ph;
                Pf = w17 && ph != DEAD ? FRONT6 + 2u * ph : FRONT6 + OWN; // (a dead chain takes no further part: it cannot be the frame's)
                const bool alive = w17 && ph != DEAD;
                for (uint32_t kb = 0u;; kb++) {
                    const uint32_t bound = FRONT6 + kb * QUART6;
                    if (kb) {
                        Pf = walk_to6<CLAMP>(careful, s_stage, s_flat, Pf, alive ? bound : 0u, limP);
                        ph = Pf >= bound ? (Pf - bound) >> 1 : DEAD;
                    }
                    const unsigned long long am = __ballot(alive && ph != DEAD);
                    const uint32_t ph0 = am ? wave_lane(ph, static_cast<uint32_t>(__builtin_ctzll(am))) : DEAD;
                    if (kb < 4u * cnt && __ballot(alive && ph != DEAD && ph != ph0) == 0ull) {
                        fr = (kb << 8) | ph0;
                        break;
                    }
                    if (kb >= 4u * cnt) { // the segment's end, and still several chains: my entry phase -> my exit phase
                        if (alive)
                            s_fmap[a0] = static_cast<uint8_t>(ph);
                        fr = NOFRONT << 8;
                        break;
                    }
                }
            }
            if (lane == 0u)
                s_front = fr;
            __builtin_amdgcn_s_setprio(0);
        }

        if (wave != RESOLVER) {
            __syncthreads(); // (the resolving wave's twin of this barrier sits in its verification loop: s_front is there)
            K6_STAMP(2, 0);
        } else {
            // ---- speculative walkers
            __builtin_amdgcn_s_setprio(3);
            const bool inq = uj < cnt; // (quarters behind the stream's last chunk: nothing to walk)
            {
                const uint32_t s0 = seg ? 0u : FRONT6; // (segment 0: the stream starts with a record at byte 0, RawData_Legacy.cpp:476)
                // (how far in front a walker starts: a wrong chain meets the true one after a number of STEPS that grows with the
                // records' size -- the longer the records, the fewer of the even bytes are record starts)
                const uint32_t warm = len <= nrec * 18u ? WARM6 : len <= nrec * 26u ? min(WARM6 + QUART6, FRONT6) : FRONT6;
                const uint32_t Pw = walk_to6<CLAMP>(careful, s_stage, s_flat, inq ? max(qb - warm, s0) : qe, qb, limP);
                a = inq && Pw >= qb ? (Pw - qb) >> 1 : DEAD;
            }
            K6_STAMP(10, RESOLVER * 64u);
            // ... then walks its quarter from there and notes where its records start: a lane finds one record per step until its
            // walk is over, so its n-th record is the one of step n -- its half position goes to byte n of eight registers
            // (steps are unrolled: static indices).  The record lists below are made from these notes, without a second walk
            // along the chain.
            x = DEAD, qn = 0u;
            notes_ok = !careful;
            bool act = a != DEAD;           // lanes that walk their quarter in the coming round
            uint32_t from = 0u, efrom = 0u; // verification: lane `from` is entered at phase `efrom`, every lane behind it where its predecessor ends
            uint32_t vstage = 0u;           // 0: the lanes against each other (lane 0 believes itself), 1: from the sure entry, 2: from the segment in front
            uint32_t rounds = 0u;
            (void)rounds;
            for (;;) {
                // ---- (re)walk the quarters of the lanes whose entry has changed
                if (!careful) { // fast form: notes in registers
                    uint32_t Pw = act ? qb + 2u * a : qe, xc = Pw, cn = 0u;
                    // (the notes go straight to `nb`: a lane that walks its quarter starts them over; one that does not stands behind its
                    // quarter from the first step on and notes nothing -- a working copy cost eight registers)
#pragma unroll
                    for (uint32_t g = 0; g < NOTES6 / 4u; g++)
                        nb[g] = act ? 0u : nb[g];
                    bool more = true;
#pragma unroll
                    for (uint32_t st = 0; st < NOTES6; st += 2u) {
#pragma unroll
                        for (uint32_t u = 0; u < 2u; u++) {
                            const uint32_t Pn = next6(s_stage, Pw);
                            const bool in = Pw < qe;
                            nb[(st + u) >> 2] |= (in ? __builtin_amdgcn_ubfe(Pw - qb, 1u, 8u) : 0u) << (8u * ((st + u) & 3u));
                            cn += in ? 1u : 0u;
                            xc = in ? Pn : xc;
                            Pw = Pn;
                        }
                        Pw = min(Pw, CLAMP);
                        asm volatile("" : "+v"(nb[st >> 2])); // (noted now: left to itself the compiler keeps the shifted positions of every step until the loop's exit)
                        if (!__any(Pw < qe)) {
                            more = false;
                            break;
                        }
                    }
                    if (more) { // records of less than 8 bytes on average: counted on, listed by a walk of their own below
                        notes_ok = false;
                        do {
#pragma unroll
                            for (uint32_t u = 0; u < 2u; u++) {
                                const uint32_t Pn = next6(s_stage, Pw);
                                const bool in = Pw < qe;
                                cn += in ? 1u : 0u;
                                xc = in ? Pn : xc;
                                Pw = Pn;
                            }
                            Pw = min(Pw, CLAMP);
                        } while (__any(Pw < qe));
                    }
                    if (act) {
                        x = (xc - qe) >> 1;
                        qn = cn;
                    }
                } else { // segments that need the careful walk: counts only
                    uint32_t Pw = act ? qb + 2u * a : qe, t, n, cn = 0u;
                    do {
#pragma unroll
                        for (int u = 0; u < 2; u++) {
                            t = step6(s_stage, s_flat, Pw, qe, limP, &n);
                            cn += n;
                            Pw += t;
                        }
                    } while (__any(t != 0u));
                    if (act) {
                        x = Pw >= qe ? (Pw - qe) >> 1 : DEAD;
                        qn = cn;
                    }
                }
                rounds++;
                // ---- verify: lane j is entered where lane j - 1 is left
                bool again = false;
                for (;;) {
                    const uint32_t xp = wave_prev(x, DEAD);
                    const uint32_t e = lane == from ? efrom : xp;
                    const bool bad = inq && lane >= from && (vstage != 0u || lane != 0u) && e != a;
                    if (__ballot(bad) != 0ull) {
                        if (bad) {
                            a = e;
                            if (e == DEAD) { // (the chain ends in front of my quarter)
                                x = DEAD;
                                qn = 0u;
                            }
                        }
                        act = bad && e != DEAD;
                        again = true;
                        break;
                    }
                    if (vstage == 0u) {
                        __syncthreads(); // (twin of the other waves' barrier above: wave RESOLVER - 1 has said where the sure chain is)
                        K6_STAMP(11, RESOLVER * 64u);
                        const uint32_t fr = s_front;
                        from = fr >> 8;
                        efrom = fr & 255u;
                        vstage = from == NOFRONT ? 2u : 1u;
                        if (from != NOFRONT)
                            continue;
                    }
                    if (vstage == 1u && from == 0u)
                        break; // the common case: every lane verified from the segment's own sure entry
                    if (vstage == 1u || from == NOFRONT) {
                        // My entry is not known from my own bytes.  What the NEXT segment is entered at is said as early as it
                        // can be said: a phase when my chains have become one inside me (the lanes behind that boundary are verified),
                        // else -- a stream whose chains never meet -- the map from my entry phase to it, for my successors to go
                        // through while I still wait for mine.
                        const size_t lme = static_cast<size_t>(f) * smax + seg;
                        if (full && from != NOFRONT) {
                            const uint32_t xp63 = wave_lane(x, NQ6 - 1u);
                            if (lane == 0)
                                look_put(look.ex + lme, epoch, (EX_PHASE << 30) | xp63);
                        } else if (full) {
                            const uint32_t xm = lane < PHASES6 ? s_fmap[lane] : DEAD;
                            uint32_t hw = 0;
#pragma unroll
                            for (uint32_t k = 0; k < 6u; k++) {
                                const uint32_t src = lane * 6u + k;
                                const uint32_t xe = __shfl(xm, static_cast<int>(src & 63u), 64);
                                hw |= (src < PHASES6 ? xe : DEAD) << (5u * k);
                            }
                            if (lane < 3u)
                                look_put(look.hm + 3u * lme + lane, epoch, hw);
                            if (lane == 0)
                                look_put(look.ex + lme, epoch, EX_MAP << 30);
                        }
                        // The nearest segment in front whose exit is known -- it has resolved and published its records, or said a
                        // phase above --, and from there through the maps of the segments in between (64 segments per poll; the
                        // frame itself is entered at phase 0).
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
                        if (lost) { // (a predecessor never said anything: the frame fails below)
                            a = DEAD;
                            x = DEAD;
                            qn = 0u;
                            break;
                        }
                        vstage = 2u;
                        from = 0u;
                        efrom = ep;
                        continue;
                    }
                    break; // (vstage 2, nothing bad: verified from the frame's own chain)
                }
                if (!again)
                    break;
            }
            K6_COUNT(25, rounds);
        }
    }

    // What unpacking wave w has to do, from the entries in s_ent (valid once the resolving wave has written them).
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
        g.pairmode = g.N + NOTES6 > ROWS_CAP / 2u; // (the layout by records: as long as the notes' stores stay inside the list)
        return g;
    };
    auto range_of = [&](uint32_t w) { return range_from(w, ent_of(w * ROWS_CH), ent_of(w * ROWS_CH + ROWS_CH)); };

    if (wave == RESOLVER) {
        const bool inq = uj < cnt;
        const size_t lme = static_cast<size_t>(f) * smax + seg;
        uint64_t *const res = look.res + lme; // mine; res[-k]: k segments before me
        uint32_t w = 0;
        // the segment's exit phase: what the next one is entered at
        const uint32_t aph = wave_lane(x, NQ6 - 1u); // entry phase of the chunk behind a full segment
        const uint32_t lastp = full ? aph : DEAD;
        const uint32_t cph = __shfl(a, static_cast<int>(lane & ~3u), 64); // entry phase of my chunk
        uint32_t qp = a;                                                   // ... of my quarter
        K6_STAMP(26, RESOLVER * 64u);
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
                look_put(res, epoch, (RES_AGG << 30) | (total << 5) | (lost ? DEAD : lastp));
            // (a window of LBW segments per poll: the nearest segment that already holds a prefix is mostly less than ten back --
            // what lies between a segment's own count and its prefix is one look-back --, and a poll is LBW small requests to the
            // fabric: with 64 lanes polling, the polls were a quarter of the kernel's fetched bytes)
            constexpr uint32_t LBW = 16;
            int32_t jn = static_cast<int32_t>(seg) - 1; // nearest segment of the window (lane 0)
            if (K6_ABL == 32 || K6_ABL == 33) // (no look-back -- the records land near their rows, by the frame's average record size; timing only)
                base = static_cast<uint32_t>(static_cast<uint64_t>(nrec) * seg * (DEC_CH * CHUNK6) / len);
            if (K6_ABL == 33) { // (... but ONE poll of the window's words, waited for and not looked at: what the reads cost without the waiting)
                const int32_t k = jn - static_cast<int32_t>(lane);
                uint32_t wv = 0;
                const bool ok = lane < LBW && k >= 0 && look_get(res - seg + k, epoch, &wv);
                if (__ballot(ok && wv == 0x12345678u) == 1ull)
                    base++;
            }
            // The first polls are SCALAR loads (round 6): the words of the eight segments in front in one s_load_dwordx16 that passes the
            // scalar cache (glc).  A vector load of the same words is queued behind everything this CU's waves have sent to memory --
            // its workgroups' stores --, and what the kernel was bound by was this hand-off: without the look-back (record indices
            // guessed) 0.27 ms, with vector polls 0.356, with scalar polls 0.30 - 0.31 (tools/ab6n.sh, docs/lab_notes.md).  A word
            // that carries this launch's epoch is what its writer wrote, whatever way it came, so what such a poll finds can be
            // relied on; what it does not find after SCALAR_POLLS6 polls is asked for by vector loads at device scope as before
            // (measured: a frame whose segments run on all eight XCDs resolves by scalar polls alone -- the fallback is there
            // because nothing documents that it must).  Windows that reach in front of the frame's first segment go the vector way too.
            constexpr uint32_t SCALAR_POLLS6 = 32;
            constexpr int SW = 8; // segments per scalar poll: ONE s_load_dwordx16 (two of them for sixteen segments: 2.5 % slower, and
                                  // 26 instead of 15 MB of polls per 380 MB of stream; four segments by s_load_dwordx8: the same as eight)
            typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
            bool prefixed = false;
            for (uint32_t polls = SCALAR_POLLS6; polls && jn >= SW - 1 && K6_ABL != 32 && K6_ABL != 33; polls--) {
                const uint64_t wa = reinterpret_cast<uint64_t>(res - seg + (jn - (SW - 1))); // words of segments jn - SW + 1 .. jn
                const uint64_t wp = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(wa >> 32)))) << 32) |
                                    static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(wa)));
                u32x16 w16;
                asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(w16) : "s"(wp) : "memory");
                uint32_t sum = 0u;
                bool ok = true, found = false;
#pragma unroll
                for (int k = 0; k < SW; k++) { // nearest segment first: word SW - 1 - k of the window
                    const int idx = 2 * (SW - 1 - k);
                    const uint32_t pl = w16[idx], ep = w16[idx + 1];
                    const bool valid = ep == epoch && (pl >> 30) != 0u;
                    if (!found) {
                        ok = ok && valid;
                        sum += valid ? (pl >> 5) & 0xFFFFFFu : 0u;
                        found = valid && (pl >> 30) == RES_PREFIX;
                    }
                }
                if (!ok) { // not all published yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                base = min(base + sum, 0xFFFFFFu);
                if (found) {
                    prefixed = true;
                    break;
                }
                jn -= SW;
            }
            while (!lost && !prefixed && K6_ABL != 32 && K6_ABL != 33) {
                const int32_t k = jn - static_cast<int32_t>(lane);
                w = RES_AGG << 30; // segments "before the frame": nothing, and never reached (segment 0 has a prefix)
                bool got = false;
                if (lane < LBW && k >= 0)
                    got = look_get(res - seg + k, epoch, &w);
                const bool ok = lane < LBW && (k < 0 || (got && (w >> 30) != 0u));
                const uint64_t okm = __ballot(ok), pm = __ballot(ok && (w >> 30) == RES_PREFIX);
                const uint32_t np = pm ? static_cast<uint32_t>(__builtin_ctzll(pm)) : LBW; // lanes in front of the first prefix
                const uint64_t need = np >= LBW ? (1ull << LBW) - 1ull : ((1ull << np) | ((1ull << np) - 1ull));
                if ((okm & need) != need) { // not all published yet
                    if (++spins > SPIN6)
                        lost = true;
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t add;
                (void)wave_excl_scan(lane <= np && lane < LBW ? (w >> 5) & 0xFFFFFFu : 0u, lane, &add);
                base = min(base + add, 0xFFFFFFu);
                if (np < LBW)
                    break;
                jn -= static_cast<int32_t>(LBW);
            }
        }
        K6_STAMP(12, RESOLVER * 64u);
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
        const uint32_t entv = inq && !lost ? (cph | (qi << 8)) : DEAD;                 // (lanes with ur == 0: chunk uj's)
        const uint32_t ent16 = full && !lost ? (aph | (endn << 8)) : DEAD;             // ... and of the chunk behind a full segment
        s_ent4[lane] = ent4v; // (the general path's walkers start from these)
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
        // chunk from where the true chain crosses into it, on ONE wave
        // (a walk costs a wave its issue slots whatever the number of walking lanes).
        // (what this wave needs of the entries it has just stored, it takes from its registers: a wait for the LDS here would
        // also wait for the look-back words' way to memory -- stores count on the same counter)
        const uint32_t uw = lane / (4u * ROWS_CH), j = (lane >> 2) & (ROWS_CH - 1u), r = lane & 3u;
        static_assert(ROWS_CH == 4, "an unpacking wave's chunks are sixteen lanes of this wave");
        const uint32_t ew0 = wave_lane(entv, 0u), ew1 = wave_lane(entv, 16u), ew2 = wave_lane(entv, 32u), ew3 = wave_lane(entv, 48u);
        auto ent_reg = [&](uint32_t w) { // entry of chunk w * ROWS_CH, as ent_of() will read it
            const uint32_t v = w >= UNPACK_W ? ent16 : w == 0u ? ew0 : w == 1u ? ew1 : w == 2u ? ew2 : ew3;
            return cfirst + w * ROWS_CH < nchunks ? v : DEAD;
        };
        const uint32_t enext_reg = ent_reg(uw + 1u);
        const Range6 rg = range_from(uw, ent_reg(uw), enext_reg);
        const bool coop = __ballot(!rg.lean) == 0ull;
        if (lane == 0)
            s_coop = coop ? 1u : 0u;
        // The common case (no jumps over runs of 2-byte records in this segment, at most NOTES6 records per quarter, every
        // record listed): the lists are the notes of the quarter walks put in their places -- stores that do not depend on one
        // another, instead of a second walk along the chain.  A lane stores
        // whole groups of eight entries.  The group its records end in is filled up with the first records of the NEXT quarter
        // (every quarter starts at least seven: a record has at most 34 bytes), taken from the lane behind it: what a lane
        // stores beyond its own records is then exactly what that lane stores there itself, and it does not matter which
        // of the two stores comes last.
        // (a list by pairs holds every second note: up to ROWS_CAP - NOTES6 records per wave; round 6 -- until then the notes served
        // the layout by records only, and a wave of more than 512 records was listed by a second walk along the chain: 12 % of the kernel)
        const bool noted = coop && notes_ok && __ballot(rg.N + NOTES6 > (rg.pairmode ? ROWS_CAP : ROWS_CAP / 2u)) == 0ull;
        if (K6_ABL != 3 && noted) {
            const int32_t slot0 = static_cast<int32_t>(qi - rg.R0); // -1: an odd first record belongs to the previous wave's
            uint16_t *lp = pos_of(uw) + slot0;                        // last pair, never listed
            const uint32_t boff = j * CHUNK6 + r * (CHUNK6 / 4u);
            {
                const uint32_t kb = qn & 7u, gp = qn >> 3;
                // the next quarter's first eight notes, 128 half positions further on (lane 63: the chunk behind the segment
                // is entered at phase `aph`)
                uint32_t x0 = wave_next(nb[0], 0u), x1 = wave_next(nb[1], 0u);
                x0 = (lane == NQ6 - 1u ? aph & 31u : x0) | 0x80808080u;
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
            // by pairs: the records of even number only -- my notes par, par + 2, .. (par: whether my first record is an odd one) --, entry
            // (slot0 + i) / 2 for note i; the neighbour's stores into my last group pick the same records by the same rule
            const uint32_t par = static_cast<uint32_t>(slot0) & 1u;
            uint16_t *const lp2 = pos_of(uw) + ((slot0 + static_cast<int32_t>(par)) >> 1);
#pragma unroll
            for (uint32_t g = 0; g < NOTES6 / 8u; g++) {
                if (g && __ballot(qn > 8u * g) == 0ull)
                    break;
                if (qn > 8u * g) { // (at most seven entries behind a lane's last record: the next quarter's first seven)
                    if (rg.pairmode) {
#pragma unroll
                        for (uint32_t k = 0; k < 4u; k++) {
                            const uint32_t v = boff + 2u * ((nb[2u * g + (k >> 1)] >> (8u * par + 16u * (k & 1u))) & 255u);
                            lp2[4u * g + k] = static_cast<uint16_t>(v);
                        }
                    } else {
#pragma unroll
                        for (uint32_t i = 8u * g; i < 8u * g + 8u; i++) {
                            const uint32_t v = boff + 2u * ((nb[i >> 2] >> (8u * (i & 3u))) & 255u);
                            if (i > 0u || slot0 >= 0)
                                lp[i] = static_cast<uint16_t>(v);
                        }
                    }
                }
            }
            // behind a wave's last record: the next wave's first one (the partner of the wave's last record when its range
            // ends on an even one; a lane whose records end on a group boundary has stored nothing behind them; by pairs: not listed)
            if (j == ROWS_CH - 1u && r == 3u && !rg.pairmode)
                lp[qn] = static_cast<uint16_t>(ROWS_CH * CHUNK6 + 2u * (enext_reg & 255u));
        } else if (K6_ABL != 3 && coop) {
            const uint32_t ej = ent4v, first = rg.R0;
            const uint8_t *base = s_own + uw * (ROWS_CH * CHUNK6);
            const uint8_t *p = base + j * CHUNK6 + r * (CHUNK6 / 4u) + 2u * (ej & 255u);
            const uint8_t *const pe = base + j * CHUNK6 + (r + 1u) * (CHUNK6 / 4u);
            uint32_t idx = ej >> 8;
            if ((ej & 255u) == DEAD)
                p = pe; // (coop implies that the chain reaches every quarter; belt and braces)
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
                                pos_of(uw)[((idx - first) >> 1) + k] = static_cast<uint16_t>(p - base + 4u * k);
                            p += 16;
                            idx += 8u;
                            continue;
                        }
                    }
                    const uint32_t hb = static_cast<uint32_t>(*p) >> 4;
                    if ((idx & 1u) == 0u && idx >= first)
                        pos_of(uw)[(idx - first) >> 1] = static_cast<uint16_t>(p - base);
                    idx++;
                    p += 2u + len6_of(hb);
                }
            } else {
                uint16_t *lp = pos_of(uw) + static_cast<int32_t>(idx - first); // [-1] for an odd first record:
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
                if (j == ROWS_CH - 1u && r == 3u && ((lp - pos_of(uw)) & 1))
                    *lp = static_cast<uint16_t>(p - base);
            }
        }
        K6_STAMP(13, RESOLVER * 64u);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    K6_STAMP(3, 0);
    const bool coop = s_coop != 0u;
    if (K6_ABL == 2)
        return;
    const Range6 mine = range_of(wave);
    const uint32_t c0 = cfirst + wave * ROWS_CH, cs0 = c0 * CHUNK6;
    const uint32_t R0 = mine.R0, R1 = mine.R1, N = mine.N;
    const bool live = mine.live, pairmode = mine.pairmode;

    // (the walk tables are dead: from here on their LDS holds the record lists)
    const bool fast = fast_store != 0u;

    // Unpack the pairs of records [wlo, whi) (both even) of unpacking wave uw, listed in pos_of(uw): four lanes per pair; of
    // its 2 * (whi - wlo) tasks, those in [tb, te).  What is the same for every task of a round -- the list's layout, whether a
    // round can span more than two rows -- is decided once, outside the loop (a workgroup has one ready wave per SIMD most of
    // the time: every scalar branch inside the loop is paid in full).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    auto unpack_loop = [&](auto by_pair_t, auto wide_t, auto fast_t, uint32_t uw, uint32_t wlo, uint32_t tb, uint32_t te) {
        constexpr bool BY_PAIR = decltype(by_pair_t)::value, WIDE = decltype(wide_t)::value, FAST = decltype(fast_t)::value;
        const uint8_t *bytes = s_own + uw * (ROWS_CH * CHUNK6);
        const uint32_t pair0 = wlo >> 1;
        const uint16_t *pairs = pos_of(uw);
        const uint32_t *both = reinterpret_cast<const uint32_t *>(pos_of(uw));
        const uint32_t y0 = pair0 / ppr, r0 = pair0 - y0 * ppr; // wave-uniform
        const uint32_t row0 = y0 * width;
        const uint32_t qt = (tb + lane) & 3u, qt4 = 4u * qt; // (t advances by 64)
        for (uint32_t t = tb + lane; t < te; t += 64u) {
            // one task: 8 pixels (even columns from record A, odd columns from record B, uint16 wrap
            // on the reference add) and where they go
            const uint32_t q = t >> 2;
            uint32_t roa, rob, ha, hb;
            if (BY_PAIR) { // the odd-column record starts where the even-column one ends (RawData_Legacy.cpp:377-442)
                roa = pairs[q];
                ha = *reinterpret_cast<const uint16_t *>(bytes + roa);
                rob = roa + 2u + len6_of(__builtin_amdgcn_ubfe(ha, 4u, 4u));
                hb = *reinterpret_cast<const uint16_t *>(bytes + rob);
            } else {
                const uint32_t ro2 = both[q];
                roa = ro2 & 0xffffu;
                rob = ro2 >> 16;
                ha = *reinterpret_cast<const uint16_t *>(bytes + roa);
                hb = *reinterpret_cast<const uint16_t *>(bytes + rob);
            }
            const Pair6 hd = pair6_head(ha, hb);
            uint32_t va[4], vb[4];
            quad6u(bytes, roa, qt4, hd.sa, va);
            quad6u(bytes, rob, qt4, hd.sb, vb);
#if K6_ABL == 16 // (timing experiment: forty vector instructions more per pass, the same memory accesses)
            {
                uint32_t d = va[0] ^ vb[3];
#pragma unroll
                for (int e = 0; e < 40; e++)
                    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(d) : "v"(va[1]), "v"(vb[2]));
                va[0] ^= d & 0x10000u;
            }
#endif
            const uint32_t n = r0 + q;
            const uint32_t dy = WIDE ? (n >= ppr ? 1u : 0u) : mul_u24(n, m20) >> 20;
            const uint32_t x = (n - __umul24(dy, ppr)) * 32u + 8u * qt; // RawData_Legacy.cpp:479-486
            const u16x2 refs = __builtin_bit_cast(u16x2, hd.refs);
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
            uint16_t *px = out + (row0 + (WIDE ? (dy ? width : 0u) : __umul24(dy, width)) + x);
            const bool whole = x + 8u <= width;
            if (K6_ABL == 1) {
                if ((o[0] ^ o[1] ^ o[2] ^ o[3]) == 0x12345678u)
                    px[0] = 1;
            } else if (__builtin_expect(__ballot(!whole) == 0ull, 1)) { // (no lane of this pass at a cropped row end: the rule)
                if (FAST) {
                    const u32x4 v = {o[0], o[1], o[2], o[3]};
                    store_stream16(px, __builtin_bit_cast(mcraw_u32x4, v)); // (write-through, streaming: as k7_tiles' stores)
                } else { // rows off the 16-byte grid: still one (unaligned) 16-byte store
                    typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(2)));
                    const u32x4_u v = {o[0], o[1], o[2], o[3]};
                    *gptr<u32x4_u>(px) = v;
                }
            } else if (whole) {
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
    auto unpack_round = [&](uint32_t uw, uint32_t wlo, uint32_t whi, bool by_pair, uint32_t tb, uint32_t te) {
        (void)whi;
        typedef std::integral_constant<bool, true> yes;
        typedef std::integral_constant<bool, false> no;
        const bool bp = __builtin_amdgcn_readfirstlane(by_pair ? 1 : 0) != 0; // (the same for all lanes: made known to the compiler)
        auto with_layout = [&](auto wide_t, auto fast_t) {
            if (bp)
                unpack_loop(yes{}, wide_t, fast_t, uw, wlo, tb, te);
            else
                unpack_loop(no{}, wide_t, fast_t, uw, wlo, tb, te);
        };
        if (fast) {
            if (widerow)
                with_layout(yes{}, yes{});
            else
                with_layout(no{}, yes{});
        } else { // rows off the 16-byte grid
            if (widerow)
                with_layout(yes{}, no{});
            else
                with_layout(no{}, no{});
        }
    };

    if (coop) {
        K6_STAMP(4, 0);
        unpack_round(wave, R0, R1, pairmode, 0u, 2u * (R1 - R0));
        K6_STAMP(5, 0);
#ifdef MCRAW_DIAG
        if (threadIdx.x == 0 && blockIdx.x < K6_PROF_WG) {
            g_k6_prof[blockIdx.x][7] = static_cast<uint32_t>(stamp_);
            g_k6_prof[blockIdx.x][17] = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
        }
        __builtin_amdgcn_s_waitcnt(0); // (the end stamp is taken when the wave's stores have landed)
        K6_END();
        __syncthreads();
        if (threadIdx.x < K6_PROF_N && blockIdx.x < K6_PROF_WG && (threadIdx.x < 6 || (threadIdx.x >= 8 && threadIdx.x < 16) || threadIdx.x == 25 || threadIdx.x == 26))
            g_k6_prof[blockIdx.x][threadIdx.x] = s_prof[threadIdx.x];
#endif
        return;
    }
    if (!live)
        return;
    const uint8_t *bytes = s_own + wave * (ROWS_CH * CHUNK6);
    // The lists of the general path: one lane per QUARTER chunk walks from where the chain enters its quarter (s_ent4: sixteen
    // walks side by side per wave -- round 6; until then one lane per chunk: a wave of 2-byte records cost 64 steps a round).
    // A walker keeps its place from round to round (restarting at the entry every round made a run of 2-byte records
    // cost rounds x 128 steps per lane).
    const uint32_t e4 = lane < 4u * ROWS_CH ? s_ent4[wave * (4u * ROWS_CH) + lane] : DEAD;
    const bool walker = lane < 4u * ROWS_CH && (e4 & 255u) != DEAD && c0 + (lane >> 2) < nchunks;
    const uint32_t qend = (lane + 1u) * QUART6; // (positions: bytes from the wave's first chunk)
    uint32_t pos = lane * QUART6 + 2u * (e4 & 255u), idx = e4 >> 8;
    uint16_t *const mylist = pos_of(wave);
    for (uint32_t base = 0; base < N; base += ROWS_CAP) {
        const uint32_t wlo = R0 + base, whi = min(R1, wlo + ROWS_CAP); // records of this round (both even)
        if (K6_ABL != 3 && walker) {
            while (idx < whi && pos < qend) {
                // Runs of 2-byte records (flat or clipped image regions) are what makes a wave land here:
                // eight of them fill an aligned 16-byte line whose even bytes all have a zero high
                // nibble -- one LDS read then lists four pairs instead of one record.
                if ((pos & 15u) == 0u && (idx & 1u) == 0u && idx >= wlo && idx + 8u <= whi && cs0 + pos + 16u < len) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(bytes + pos);
                    if (((q.x | q.y | q.z | q.w) & 0x00F000F0u) == 0u) {
#pragma unroll
                        for (uint32_t k = 0; k < 4u; k++)
                            mylist[((idx - wlo) >> 1) + k] = static_cast<uint16_t>(pos + 4u * k);
                        pos += 16u;
                        idx += 8u;
                        continue;
                    }
                }
                const uint32_t nx = pos + 2u + len6_of(static_cast<uint32_t>(bytes[pos]) >> 4);
                if (cs0 + nx >= len)
                    break; // k6_frame has already failed the frame if records are missing
                if (idx >= wlo && (idx & 1u) == 0u)
                    mylist[(idx - wlo) >> 1] = static_cast<uint16_t>(pos);
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
