// mcraw_abi.hip -- host side of the C ABI declared in include/mcraw_hip.h.
//
// Owns the HIP context of the decode path: streams, a ring of batch slots
// (pinned upload buffer + HBM arena + completion event), per-batch planning
// (geometry, workspace carving, flat work-item tables) and the kernel launches.
// Replaces the per-frame dispatch of lib/Decoder.cpp:216-234 with one batched
// submit.  There is no CPU decode fallback in this file or anywhere behind it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mcraw_hip.h"
#include "mcraw_plan.h"

namespace mcraw {
void launch_k7(const Work7 &W, uint32_t stage, hipStream_t st, bool thin = false);
void launch_k6_decode(const Plan6 *plans, const uint32_t *wg_tab, uint32_t stage0, uint32_t nwg, const Look6 &look,
                      uint32_t *tickets, uint32_t epoch, int nframes, uint32_t smax, const Post &post, hipStream_t st);
} // namespace mcraw

using namespace mcraw;

struct mcraw_ticket;

namespace {

thread_local std::string g_err;

int fail(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return -static_cast<int>(e ? e : hipErrorUnknown);
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail(e_, #expr);                                                                                    \
    } while (0)

constexpr int NSLOT = 16; // host-memory sub-batches in flight (two batches of five, with room)
constexpr int NDSLOT = 4; // device-memory batches the host may run ahead by
constexpr size_t ALIGN = 256;

inline size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Slot {
    Buf pinned;  // host upload image of the batch tables
    Buf arena;   // HBM: tables + workspace
    Buf dev_in;  // HBM staging of inputs  (MCRAW_MEM_HOST)
    Buf dev_out; // HBM staging of outputs (MCRAW_MEM_HOST)
    Buf status_host; // pinned: statuses copied back
    // legacy frames: look-back state of k6_decode.  Never cleared after it was allocated: state words carry the epoch
    // of the launch that wrote them.
    Buf look;
    uint32_t look_epoch = 0;
    // Device statuses are kept in plan order (type-7 frames, then legacy frames) so a kernel
    // finds its word from its frame index alone; `order` maps them back to the caller's
    // frame indices and `host_status` holds what the host decided on its own (bad arguments).
    std::vector<int> order;
    std::vector<int32_t> host_status;
    int n7 = 0; // type-7 frames of the batch in this slot (their coded heights follow the statuses)
    int wpf = 2; // status words per type-7 frame the batch was launched with (one per part of its side streams)
    Buf side_sync; // type-7 frames: what the parts of a side stream tell each other (k7_side); never cleared, epoch-tagged words
    hipEvent_t done = nullptr;
    hipEvent_t fork = nullptr, join = nullptr; // a batch that holds both encodings: its legacy kernel runs on the context's second stream
    hipEvent_t side_done = nullptr; // k7_side of the batch, when it ran on the context's side stream
    ::mcraw_ticket *owner = nullptr; // host-memory batch whose statuses still sit in this slot's arena
    int owner_part = -1;
    hipEvent_t uploaded = nullptr; // host-memory pipeline: inputs of the sub-batch are in HBM
#ifdef MCRAW_TIMELINE
    hipEvent_t tl_begin = nullptr; // in front of the sub-batch's uploads
    double tl_host = 0.0;          // host clock when the sub-batch was queued (ms since the context's first)
#endif
    hipEvent_t decoded = nullptr;  // ... its kernels have run
    hipStream_t stream = nullptr; // the slot's own stream (host-memory pipeline: the kernels of a sub-batch)
    bool busy = false;
    uint64_t seq = 0;    // host-memory pipeline: the order the sub-batches were queued in
    bool landed = false; // ... this one's downloads are known to be over (its statuses may still wait for their ticket)
    // A device-memory batch submitted without a status request: what is needed to plan frames again
    // whose header asks for more workspace than they were given (mcraw_ctx_synchronize, or the
    // next use of the slot, does that before the batch is forgotten).
    std::vector<mcraw_frame> frames;
    size_t status_off = 0;
    Post post{0, 0, 0};
    bool unresolved = false;
    uint64_t serial = 0; // of the device-memory batch in this slot
};

// One sub-batch of a host-memory batch, riding in a slot.
struct Part {
    int slot, first, count;
    size_t status_off;
    bool drained;
    bool sent; // its status words went home behind its kernels (send_status): nothing to fetch when it is drained
};

struct KStat {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double ms = 0.0;
    int launches = 0;
};

} // namespace

// An asynchronous host-memory batch (mcraw_decode_batch_async / mcraw_ticket_wait).
struct mcraw_ticket {
    mcraw_ctx *c = nullptr;
    std::vector<mcraw_frame> frames;
    std::vector<int32_t> status;
    std::vector<uint32_t> encH; // coded heights (type-7 frames)
    std::vector<Part> parts;
    std::vector<int> skipped; // frames that no sub-batch holds (no device memory for their workspace): failed on their own
    Post post{0, 0, 0}; // post stage the batch was submitted with
    bool small = false; // a few sub-batches only: scheduled the short way (host_submit)
    bool send = false;  // ... and its status words go home behind their kernels (send_status)
    int want_send = -1; // (deal_host: what this piece is to do; -1: what the context has decided)
    int trial_way = -1; // a ticket of the context's trial rows (mcraw_decode_batch_async): which row
    size_t trial_bytes = 0;
    bool big_trial = false; // a large batch whose way is being compared (big_way)
    int way = 0;
    std::chrono::steady_clock::time_point t_queued;
    // A large batch queued with mcraw_decode_batch_async is dealt out as a row of short ones (deal_host): this ticket then holds
    // the ones still under way (oldest first, with the index of their first frame) and the results of those that have landed.
    bool composite = false;
    std::vector<std::unique_ptr<mcraw_ticket>> pieces;
    std::vector<int> piece_first;
    std::vector<size_t> got_written;
    std::vector<int32_t> got_status;
};

// Contexts of this process per device.  The short way of the host-memory pipeline (host_submit) is tuned for ONE stream of batches
// on a GPU's copy engines: two contexts on one device (the bench's pool of two members on one GPU) take the long way, as before.
std::atomic<int> g_ctx_on_device[64];

struct mcraw_ctx {
    uint64_t part_seq = 0;
    bool counted = false; // in g_ctx_on_device
    // Host-memory pipeline: do the status words go home behind their kernels (1) or are they fetched when the batch is waited for
    // (0)?  Decided by measurement on the first large batch (deal_host), or by MCRAW_SHORT_WAY=0|1; until then: fetched.
    int send_home = -1;         // ... a large batch in one synchronous call
    int send_home_tickets = -1; // ... a stream of tickets
    double trial_rate[2] = {0.0, 0.0}; // bytes per second of the two trial batches (fetched, sent)
    int big_seen = 0;                  // large batches so far (the first one is not compared)
    int sent_trials = 0;               // ... that sent (the first of them is not compared either)
    // ... and for a caller that streams short tickets instead (the facade's chunks): TRIAL_TICKETS in a row fetch, the next
    // TRIAL_TICKETS send, the rate between the first and the last landing of each row is compared
    struct TicketTrial {
        int way = 0, queued = 0, landed = 0;
        size_t bytes = 0;
        std::chrono::steady_clock::time_point t_first;
        double rate[2] = {0.0, 0.0};
    } tt;
#ifdef MCRAW_TIMELINE
    hipEvent_t tl0 = nullptr; // the timeline's zero: recorded on the upload stream in front of the first sub-batch
    std::chrono::steady_clock::time_point tl_host0;
#endif
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t h2d = nullptr, d2h = nullptr; // host-memory pipeline: one stream per copy direction
    Slot slots[NSLOT];   // host-memory pipeline
    int next_slot = 0;
    Slot dslots[NDSLOT]; // device-memory batches (tables + workspace only)
    int next_dslot = 0;
    Slot rslot;          // frames planned a second time (always drained before the call returns)
    hipStream_t aux = nullptr; // deferred second plans of batches whose caller stream is not known any more
    hipStream_t legacy = nullptr; // the legacy kernel of a batch that holds both encodings (beside the type-7 kernels)
    // EXPERIMENT (off by default, MCRAW_SIDE_CUS; see mcraw_ctx_create).  Batches that follow each other on the context's OWN
    // stream (caller's stream NULL): k7_side is a chain per side stream (lib/RawData.cpp:463-498) that has to be done before the
    // tile loop (:556-562 -> :571-608), a handful of latency-bound workgroups.  k7_side of batch n + 1 can run on `side` -- a
    // stream of the lowest priority, a hardware queue of its own -- while k7_tiles of batch n streams: as THIN workgroups (one
    // wave per SIMD, 23 KB of LDS), which find room on a CU whenever one of the tile kernel's leaves (the fat ones, 57 KB and two
    // waves per SIMD, starve until the tile kernel is through).  `tmain`: only with a CU partition (MCRAW_SIDE_CUS > 0).
    hipStream_t side = nullptr, tmain = nullptr;
    bool side_fat = false;  // MCRAW_SIDE_FAT (timing experiment): the fat workgroups on the side stream too
    bool side_thin = false; // MCRAW_SIDE_THIN (tests, experiments): every k7_side launch as thin workgroups
    hipStream_t last_own = nullptr;  // which of the context's own streams the last own-stream batch went to
    hipEvent_t chain = nullptr;      // orders two own-stream batches that went to different streams
    uint32_t profile = 0; // bit id: bracket launches of kernel id with events
    uint32_t profile_every = 1, profile_tick[MCRAW_K_COUNT] = {0}; // ... every n-th launch of it only
    Post post{0, 0, 0};   // fused post-decode stage of the batches to come (mcraw_ctx_set_post)
    KStat kstat[MCRAW_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    // How k7_tiles' workgroups are dealt to the XCDs (Work7::xcd_chunk), chosen by measurement for large resident batches:
    // which of the candidates is faster depends on where the caller's buffers lie in physical memory (see submit()).  The
    // choice is made PER GEOMETRY (frames, groups), not per buffer: the first launches of a geometry try each candidate twice
    // between events and the faster one stays; afterwards one launch in 64 is timed -- the chosen candidate and the other one
    // in turn --, and the choice moves when the other one has become the faster (a caller whose buffers change is never
    // left measuring, and one whose buffers moved to a place where the other mapping wins gets there).
    struct Tune {
        static constexpr int NC = 2;
        int key_n = 0;       // what the choice was made for: frames, groups, row format (another kernel instance, other rows)
        uint32_t key_R = 0, key_mode = 0;
        int issued[NC] = {0, 0}, done[NC] = {0, 0};
        float best[NC] = {0.f, 0.f}; // first samples: the minimum; afterwards a moving average
        int decided = -1;
        unsigned long long launches = 0; // tunable launches since the decision
        struct Pending {
            hipEvent_t a, b;
            int cand;
        };
        std::vector<Pending> pending;
        unsigned long long used = 0; // (least recently used entry is replaced)
    } tunes[4]; // a few geometries at a time
    // How many workgroups ("parts") resolve a long side stream of a small resident batch (Work7::nsplit[bits, refs]): which of
    // the two streams is the slow one is a matter of content -- the bits stream of coded frames (short runs of equally long
    // records), the refs stream of noise --, and the chip holds 512 workgroups of k7_side at a time.  Chosen like the XCD mapping:
    // the first launches of a geometry try each candidate twice between events, the fastest stays, one launch in 64 re-checks.
    struct SideTune {
        static constexpr int MAXC = 8;
        int key_n = 0;
        uint32_t key_R = 0;
        int nc = 0, cand[MAXC][2] = {{1, 1}};
        int issued[MAXC] = {0}, done[MAXC] = {0};
        float best[MAXC] = {0.f};
        int decided = -1;
        unsigned long long launches = 0, used = 0;
        struct Pending {
            hipEvent_t a, b;
            int cand;
        };
        std::vector<Pending> pending;
    } side_tunes[4];
    int side_last = -1;
    unsigned long long tune_clock = 0;
    int tune_last = -1; // entry of the last tunable batch (mcraw_ctx_xcd_runs)
    // last device-memory batch, for mcraw_ctx_synchronize
    int last_slot = -1;
    int last_n = 0;
    std::vector<int32_t> last_status; // its statuses once resolved
    // every device-memory batch has a serial number; the statuses of the last few that were submitted WITHOUT a status
    // request are kept once they are known (mcraw_ctx_batch_status), and their OR since the last look (mcraw_ctx_errors):
    // a caller that queues batches back to back -- the device pool, from several host threads -- can still tell which failed
    uint64_t serial = 0;
    std::deque<std::pair<uint64_t, std::vector<int32_t>>> settled;
    int32_t sticky = 0;
    std::mutex mu;
};

namespace {

int ensure(Buf &b, size_t bytes, bool pinned)
{
    if (bytes <= b.cap)
        return 0;
    size_t want = std::max(bytes, b.cap + b.cap / 2);
    want = up(want, 1 << 20);
    void *np = nullptr; // the old buffer stays valid until the new one exists
    if (pinned)
        HIP_TRY(hipHostMalloc(&np, want, hipHostMallocDefault));
    else
        HIP_TRY(hipMalloc(&np, want));
    if (b.p)
        (void)(pinned ? hipHostFree(b.p) : hipFree(b.p));
    b.p = np;
    b.cap = want;
    return 0;
}

hipEvent_t get_event(mcraw_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr;
    return e;
}

struct KTimer { // brackets one launch with events on the launch stream
    mcraw_ctx *c;
    int id;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    KTimer(mcraw_ctx *c_, int id_, hipStream_t st_) : c(c_), id(id_), st(st_)
    {
        if ((c->profile & (1u << id)) && (c->profile_tick[id]++ % c->profile_every) == 0u) {
            a = get_event(c);
            b = get_event(c);
            if (a && b)
                (void)hipEventRecord(a, st);
        }
    }
    ~KTimer()
    {
        if (a && b) {
            (void)hipEventRecord(b, st);
            c->kstat[id].pending.emplace_back(a, b);
        }
    }
};

// Geometry the host plans a type-7 frame with unless the header says otherwise.
struct Geom7 {
    uint32_t encW, encH;
};

struct Batch {
    std::vector<Plan7> p7;
    std::vector<int> idx7; // frame index in the caller's array
    std::vector<Plan6> p6;
    std::vector<int> idx6;
};

// Carve `bytes` out of a running arena offset.
inline size_t carve(size_t &off, size_t bytes)
{
    size_t o = off;
    off = up(off + bytes, ALIGN);
    return o;
}

struct Layout { // byte offsets inside the slot arena / upload image
    size_t status = 0;                                   // int32[n + 1 + n7]
    size_t plans7 = 0;                                   // Plan7[n7]
    size_t plans6 = 0, tickets = 0, wg_tab = 0;
    size_t upload_bytes = 0;                             // tables end here, workspace follows
    size_t total = 0;
};

constexpr uint32_t TUNE_CHUNKS[mcraw_ctx::Tune::NC] = {128u, 0u};

// Which candidate the next k7_tiles launch of a large resident batch runs with: -1 = the entry's `decided` (not timed),
// else the candidate to run AND time.  Never blocks: finished event pairs are collected as they come.
int tune_pick(mcraw_ctx *c, int n7, uint32_t R, uint32_t mode)
{
    constexpr int NC = mcraw_ctx::Tune::NC, SAMPLES = 2, NT = static_cast<int>(sizeof(c->tunes) / sizeof(c->tunes[0]));
    constexpr unsigned long long RECHECK = 64; // one launch in this many is timed once the choice is made
    int e = -1, lru = 0;
    for (int i = 0; i < NT; i++) {
        if (c->tunes[i].key_n == n7 && c->tunes[i].key_R == R && c->tunes[i].key_mode == mode)
            e = i;
        if (c->tunes[i].used < c->tunes[lru].used)
            lru = i;
    }
    if (e < 0) { // another geometry: measure, in the entry that was not used for the longest time
        e = lru;
        mcraw_ctx::Tune &t = c->tunes[e];
        for (auto &p : t.pending) { // (their results belong to the old geometry)
            (void)hipEventSynchronize(p.b);
            c->event_pool.push_back(p.a);
            c->event_pool.push_back(p.b);
        }
        t.pending.clear();
        t.key_n = n7;
        t.key_R = R;
        t.key_mode = mode;
        t.decided = -1;
        t.launches = 0;
        for (int k = 0; k < NC; k++)
            t.issued[k] = t.done[k] = 0, t.best[k] = 0.f;
    }
    mcraw_ctx::Tune &t = c->tunes[e];
    t.used = ++c->tune_clock;
    c->tune_last = e;
    for (size_t i = 0; i < t.pending.size();) {
        if (hipEventQuery(t.pending[i].b) != hipSuccess) {
            (void)hipGetLastError(); // (hipErrorNotReady is no error)
            i++;
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.pending[i].a, t.pending[i].b) == hipSuccess && ms > 0.f) {
            const int k = t.pending[i].cand;
            if (t.decided < 0)
                t.best[k] = t.done[k] ? std::min(t.best[k], ms) : ms;
            else
                t.best[k] = 0.75f * t.best[k] + 0.25f * ms;
            t.done[k]++;
        }
        c->event_pool.push_back(t.pending[i].a);
        c->event_pool.push_back(t.pending[i].b);
        t.pending.erase(t.pending.begin() + static_cast<long>(i));
    }
    if (t.decided >= 0) {
        const int other = 1 - t.decided;
        if (t.best[other] < 0.99f * t.best[t.decided]) // (the re-checks say the other mapping has become the faster one)
            t.decided = other;
        t.launches++;
        if (t.launches % RECHECK != 0 || !t.pending.empty())
            return -1;
        return (t.launches / RECHECK) % 2 ? 1 - t.decided : t.decided;
    }
    bool all = true;
    for (int k = 0; k < NC; k++)
        all = all && t.done[k] >= SAMPLES;
    if (all) {
        t.decided = 0;
        for (int k = 1; k < NC; k++)
            if (t.best[k] < t.best[t.decided])
                t.decided = k;
        return -1;
    }
    int pick = -1;
    for (int k = 0; k < NC; k++)
        if (t.issued[k] < SAMPLES + 1 && (pick < 0 || t.issued[k] < t.issued[pick]))
            pick = k;
    if (pick < 0) // every candidate is issued, results still on their way: the caller runs with the first meanwhile
        return -1;
    t.issued[pick]++;
    return pick;
}

// The split of the side streams the next k7_side launch of a resident batch runs with: the index of a candidate to run AND
// time, or -1 = the entry's `decided` (the first candidate while nothing is decided).  c->side_last is the entry.
int side_pick(mcraw_ctx *c, int n7, uint32_t R)
{
    typedef mcraw_ctx::SideTune ST;
    constexpr int SAMPLES = 2, NT = static_cast<int>(sizeof(c->side_tunes) / sizeof(c->side_tunes[0]));
    constexpr unsigned long long RECHECK = 64;
    int e = -1, lru = 0;
    for (int i = 0; i < NT; i++) {
        if (c->side_tunes[i].key_n == n7 && c->side_tunes[i].key_R == R && c->side_tunes[i].nc)
            e = i;
        if (c->side_tunes[i].used < c->side_tunes[lru].used)
            lru = i;
    }
    if (e < 0) {
        e = lru;
        ST &t = c->side_tunes[e];
        for (auto &p : t.pending) {
            (void)hipEventSynchronize(p.b);
            c->event_pool.push_back(p.a);
            c->event_pool.push_back(p.b);
        }
        t.pending.clear();
        t.key_n = n7;
        t.key_R = R;
        t.decided = -1;
        t.launches = 0;
        // (512 workgroups of k7_side are resident at once; parts that own little leave early, so somewhat more can pay:
        // 120 x 8K frames ran fastest with 4 + 1 parts = 600 workgroups, tools/side_split.py)
        const int budget = 1024 / std::max(n7, 1);
        static const int all[][2] = {{4, 4}, {4, 2}, {4, 1}, {2, 2}, {2, 4}, {3, 1}, {1, 3}, {1, 1}}; // (unsplit can win too)
        t.nc = 0;
        for (const auto &cd : all)
            if (cd[0] + cd[1] <= budget && t.nc < ST::MAXC)
                t.cand[t.nc][0] = cd[0], t.cand[t.nc][1] = cd[1], t.nc++;
        if (t.nc == 0)
            t.cand[0][0] = t.cand[0][1] = 1, t.nc = 1;
        for (int k = 0; k < ST::MAXC; k++)
            t.issued[k] = t.done[k] = 0, t.best[k] = 0.f;
    }
    ST &t = c->side_tunes[e];
    t.used = ++c->tune_clock;
    c->side_last = e;
    for (size_t i = 0; i < t.pending.size();) {
        if (hipEventQuery(t.pending[i].b) != hipSuccess) {
            (void)hipGetLastError();
            i++;
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.pending[i].a, t.pending[i].b) == hipSuccess && ms > 0.f) {
            const int k = t.pending[i].cand;
            if (t.decided < 0)
                t.best[k] = t.done[k] ? std::min(t.best[k], ms) : ms;
            else
                t.best[k] = 0.75f * t.best[k] + 0.25f * ms;
            t.done[k]++;
        }
        c->event_pool.push_back(t.pending[i].a);
        c->event_pool.push_back(t.pending[i].b);
        t.pending.erase(t.pending.begin() + static_cast<long>(i));
    }
    if (t.nc == 1) {
        t.decided = 0;
        return -1;
    }
    if (t.decided >= 0) {
        for (int k = 0; k < t.nc; k++)
            if (t.done[k] > 0 && t.best[k] < 0.97f * t.best[t.decided])
                t.decided = k;
        t.launches++;
        if (t.launches % RECHECK != 0 || !t.pending.empty())
            return -1;
        return static_cast<int>((t.launches / RECHECK) % static_cast<unsigned long long>(t.nc));
    }
    bool all_done = true;
    for (int k = 0; k < t.nc; k++)
        all_done = all_done && t.done[k] >= SAMPLES;
    if (all_done) {
        t.decided = 0;
        for (int k = 1; k < t.nc; k++)
            if (t.best[k] < t.best[t.decided])
                t.decided = k;
        return -1;
    }
    int pick = -1;
    for (int k = 0; k < t.nc; k++)
        if (t.issued[k] < SAMPLES + 1 && (pick < 0 || t.issued[k] < t.issued[pick]))
            pick = k;
    if (pick < 0) {
        if (t.pending.empty()) { // every sample is in or was lost (an event that could not be read): decide on what there is
            t.decided = 0;
            for (int k = 1; k < t.nc; k++)
                if (t.done[k] && (!t.done[t.decided] || t.best[k] < t.best[t.decided]))
                    t.decided = k;
        }
        return -1;
    }
    t.issued[pick]++;
    return pick;
}

int submit(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, const std::vector<Geom7> *geom_override,
           const uint8_t *const *dev_in, uint16_t *const *dev_out, hipStream_t st, size_t *status_off, hipStream_t side_st = nullptr)
{
    // ---- plan on the host -------------------------------------------------
    std::vector<int32_t> status(n, 0);
    Batch B;
    for (int i = 0; i < n; i++) {
        const mcraw_frame &f = frames[i];
        const uint8_t *in = dev_in ? dev_in[i] : f.in;
        uint16_t *out = dev_out ? dev_out[i] : f.out;
        if (!in || !out || f.width <= 0 || f.height <= 0 || f.len == 0 || f.len >= (1ull << 32) ||
            (f.type != MCRAW_TYPE_BLOCK && f.type != MCRAW_TYPE_LEGACY) ||
            reinterpret_cast<uintptr_t>(out) % 2 != 0 ||
            static_cast<uint64_t>(f.width) * static_cast<uint64_t>(f.height) >= (1ull << 31)) {
            status[i] = MCRAW_E_ARGS;
            continue;
        }
        const uint32_t pmode = c->post.mode;
        // vector stores: 16-byte rows pieces of uint16, or 12-byte pieces of a 12-bit strip (dword aligned)
        // (10- and 14-bit strips go out as 2-byte aligned pieces: any uint16 pointer will do)
        const uintptr_t oalign = (pmode & POST_PACK12) ? 4 : (pmode & POST_PACKED) ? 2 : 16;
        const bool fast = (reinterpret_cast<uintptr_t>(out) % oalign == 0) && (f.width % 8 == 0);
        if (f.type == MCRAW_TYPE_BLOCK) {
            Plan7 p{};
            p.in = in;
            p.out = out;
            p.len = static_cast<uint32_t>(f.len);
            p.width = f.width;
            // coded geometry the frame gets workspace and grid for: the header's where the host has
            // seen it, else what an encoder makes of width x height (RawData.cpp reads it from the
            // header only, :545-554; k7_side does the same and reports a frame that needs more)
            uint32_t encW = static_cast<uint32_t>(up(f.width, 64)), encH = static_cast<uint32_t>(up(f.height, 4));
            if (geom_override && (*geom_override)[i].encW) {
                encW = (*geom_override)[i].encW;
                encH = (*geom_override)[i].encH;
            }
            if (static_cast<uint64_t>(encW) * encH >= (1ull << 31) || (encW & 63u) || (encH & 3u)) {
                status[i] = MCRAW_E_HEADER;
                continue;
            }
            const uint32_t rows = std::min<uint32_t>(static_cast<uint32_t>(f.height), encH);
            if (f.out_capacity * 2 < static_cast<size_t>(rows) * post_row_bytes(static_cast<uint32_t>(f.width), pmode)) {
                status[i] = MCRAW_E_CAPACITY;
                continue;
            }
            p.height = f.height;
            p.ngroups = (4 * (encW / 64) * (encH / 4) + GROUP_BLOCKS - 1) / GROUP_BLOCKS;
            p.fast_store = fast ? 1u : 0u;
            B.p7.push_back(p);
            B.idx7.push_back(i);
        } else {
            Plan6 p{};
            p.in = in;
            p.out = out;
            p.len = static_cast<uint32_t>(f.len);
            p.width = f.width;
            p.height = f.height;
            if (f.out_capacity * 2 < static_cast<size_t>(f.height) * post_row_bytes(static_cast<uint32_t>(f.width), pmode)) {
                status[i] = MCRAW_E_CAPACITY;
                continue;
            }
            p.padded = static_cast<uint32_t>(up(f.width, 32));
            p.recs_per_row = 2 * p.padded / 32;
            p.nrec = p.recs_per_row * static_cast<uint32_t>(f.height);
            if (p.nrec >= (1u << 24)) { // chunk entries carry the first record index in 24 bits
                status[i] = MCRAW_E_ARGS;
                continue;
            }
            p.nchunks = static_cast<uint32_t>((f.len + CHUNK6 - 1) / CHUNK6);
            p.fast_store = fast ? 1u : 0u;
            B.p6.push_back(p);
            B.idx6.push_back(i);
        }
    }
    const int n7 = static_cast<int>(B.p7.size()), n6 = static_cast<int>(B.p6.size());
    // k7_side on the context's side stream (as thin workgroups that find room beside the tile kernel's), the rest of the batch on
    // `st` behind an event
    hipStream_t sst = side_st && n7 > 0 && s.side_done && !geom_override && !dev_in ? side_st : nullptr;
    const bool no_thin = c->side_fat, all_thin = c->side_thin; // (experiments and tests: MCRAW_SIDE_FAT, MCRAW_SIDE_THIN)

    // Type-7 plans in order of decreasing size, cut into size classes: the unpack kernel is launched
    // once per class with that class's group count, so a batch that mixes small and large frames does
    // not spend the largest frame's grid on every frame (BASELINE config 4 mixes 2 MP and 12 MP frames).
    uint32_t nclasses = 0, class_first[Work7::MAX_CLASSES + 1] = {0}, class_groups[Work7::MAX_CLASSES] = {0};
    if (n7) {
        std::vector<int> perm(n7);
        for (int k = 0; k < n7; k++)
            perm[k] = k;
        std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return B.p7[a].ngroups > B.p7[b].ngroups; });
        bool sorted = true;
        for (int k = 0; k < n7; k++)
            sorted = sorted && perm[k] == k;
        if (!sorted) {
            std::vector<Plan7> p7(n7);
            std::vector<int> idx7(n7);
            for (int k = 0; k < n7; k++) {
                p7[k] = B.p7[perm[k]];
                idx7[k] = B.idx7[perm[k]];
            }
            B.p7.swap(p7);
            B.idx7.swap(idx7);
        }
        for (int k = 0; k < n7; k++) {
            const uint32_t g = B.p7[k].ngroups;
            // a frame joins the current class while it wastes at most a fifth of the class's grid
            if (nclasses == 0 || (g * 5u < class_groups[nclasses - 1] * 4u && nclasses < static_cast<uint32_t>(Work7::MAX_CLASSES))) {
                class_first[nclasses] = static_cast<uint32_t>(k);
                class_groups[nclasses] = g;
                nclasses++;
            }
        }
        class_first[nclasses] = static_cast<uint32_t>(n7);
    }

    // ---- lay out the upload image and the workspace ------------------------
    Layout L;
    size_t off = 0;
    // status words: two per type-7 frame (one per side stream, each written once by its workgroup), one per legacy
    // frame, one spare; then the coded height of every type-7 frame
    // Long side streams of small batches are resolved by several workgroups each (k7_side "parts") when the chip has room for
    // them all at once (two workgroups of k7_side per CU).  What a part saves is the other parts' pieces; what it adds is a
    // count over its own pieces and a hand-off: measured (tools/side_split.py, tools/side_warm2.sh), 16 x 12 MP frames
    // 160 -> 85 us with four parts per stream (14-bit noise 205 -> 162), 120 x 8K 270 -> 208 us with two, UHD frames (streams
    // of two to eight pieces) lose.  Which stream needs the parts is a matter of content, so resident batches measure
    // (side_pick); host-memory batches and re-planned frames take two or four per stream.  MCRAW_SIDE_SPLIT=b,r pins the
    // numbers (tests run the type-7 suites with 2,2 and 4,4).
    int nsplit[2] = {1, 1};
    int side_cand = -1;
    {
        uint32_t rmax = 0;
        for (const Plan7 &p : B.p7)
            rmax = std::max(rmax, p.ngroups);
        const bool longstreams = rmax >= 2900u && n7 * 4 <= 1024;
        if (longstreams && !dev_in && !geom_override && !std::getenv("MCRAW_SIDE_SPLIT")) {
            side_cand = side_pick(c, n7, rmax);
            const mcraw_ctx::SideTune &t = c->side_tunes[c->side_last];
            const int k = side_cand >= 0 ? side_cand : std::max(t.decided, 0);
            nsplit[0] = t.cand[k][0];
            nsplit[1] = t.cand[k][1];
        } else if (longstreams && n7 * 8 <= 512)
            nsplit[0] = nsplit[1] = 4;
        else if (longstreams && n7 * 4 <= 512)
            nsplit[0] = nsplit[1] = 2;
        if (const char *e = std::getenv("MCRAW_SIDE_SPLIT")) {
            int b = 0, r = 0;
            const int got = std::sscanf(e, "%d,%d", &b, &r);
            if (got == 1)
                r = b;
            if (got >= 1 && b >= 1 && b <= MAX_SPLIT7 && r >= 1 && r <= MAX_SPLIT7)
                nsplit[0] = b, nsplit[1] = r;
        }
    }
    const int wpf = nsplit[0] + nsplit[1]; // status words (= workgroups of k7_side) per type-7 frame
    const size_t nstatus = static_cast<size_t>(wpf) * n7 + n6 + 1;
    L.status = carve(off, sizeof(int32_t) * (nstatus + n7));
    L.plans7 = carve(off, sizeof(Plan7) * n7);
    L.plans6 = carve(off, sizeof(Plan6) * n6);
    L.tickets = carve(off, sizeof(uint32_t) * TICKET_STRIDE6 * n6); // k6_decode's segment counters: uploaded as zeros
    L.wg_tab = carve(off, sizeof(uint32_t) * (3 * n6 + 1));         // ... and the order its workgroups take the frames in
    L.upload_bytes = off;

    // type-7 workspace: one stride for every frame (the largest frame's), so the
    // kernels address it from (frame, group) alone
    size_t Rmax = 0;
    for (const Plan7 &p : B.p7)
        Rmax = std::max<size_t>(Rmax, p.ngroups);
    const size_t w_frames = carve(off, sizeof(Frame7) * n7);
    const size_t w_bits = carve(off, Rmax * 64 * n7);
    const size_t w_refs = carve(off, Rmax * 64 * sizeof(uint16_t) * n7);
    const size_t w_grp = carve(off, sizeof(uint32_t) * (Rmax * ITEM_SPLIT + 1) * n7);
    // side streams in parts: where the records of a part's pieces start, left by its count for its decode (k7_side)
    const size_t w_rpos = wpf > 2 ? carve(off, sizeof(uint16_t) * Rmax * 2 * MAX_SPLIT7 * n7) : 0;
    // k6_decode goes over the legacy frames round by round (round r: segment r of every frame that has one): the
    // frames by falling number of segments; stage t = the rounds in which all but the t smallest frames are in play
    uint32_t smax = 0; // segments of the longest legacy stream
    std::vector<uint32_t> wg_tab(3 * n6 + 1, 0);
    {
        std::vector<uint32_t> nseg(n6);
        for (int k = 0; k < n6; k++) {
            nseg[k] = (B.p6[k].nchunks + SEG_CHUNKS6 - 1) / SEG_CHUNKS6;
            smax = std::max(smax, nseg[k]);
        }
        mcraw_legacy_launch_order(nseg.data(), n6, wg_tab.data());
    }
    L.total = off;
    {
        const size_t need6 = sizeof(uint64_t) * 5 * smax * static_cast<size_t>(n6); // res, ex, hm[3] per segment
        if (need6 > s.look.cap) {
            if (int rc = ensure(s.look, need6, false))
                return rc;
            HIP_TRY(hipMemsetAsync(s.look.p, 0, s.look.cap, st)); // epoch 0 = never written
        }
        const size_t need7 = wpf > 2 ? sizeof(uint64_t) * 2 * 2 * MAX_SPLIT7 * static_cast<size_t>(n7) : 0; // two words per part of a side stream
        if (need7 > s.side_sync.cap) {
            if (int rc = ensure(s.side_sync, need7, false))
                return rc;
            HIP_TRY(hipMemsetAsync(s.side_sync.p, 0, s.side_sync.cap, sst ? sst : st));
        }
        if (++s.look_epoch == 0u) { // (2^32 batches later: start over)
            if (s.look.p)
                HIP_TRY(hipMemsetAsync(s.look.p, 0, s.look.cap, st));
            if (s.side_sync.p)
                HIP_TRY(hipMemsetAsync(s.side_sync.p, 0, s.side_sync.cap, sst ? sst : st));
            s.look_epoch = 1;
        }
    }

    if (int rc = ensure(s.arena, L.total, false))
        return rc;
    if (int rc = ensure(s.pinned, L.upload_bytes, true))
        return rc;
    uint8_t *dev = static_cast<uint8_t *>(s.arena.p);
    uint8_t *img = static_cast<uint8_t *>(s.pinned.p);

    std::memset(img + L.status, 0, sizeof(int32_t) * (nstatus + n7));
    s.host_status = status;
    s.n7 = n7;
    s.wpf = wpf;
    s.order = B.idx7;
    s.order.insert(s.order.end(), B.idx6.begin(), B.idx6.end());
    if (n7)
        std::memcpy(img + L.plans7, B.p7.data(), sizeof(Plan7) * n7);

    for (int k = 0; k < n6; k++)
        B.p6[k].status = reinterpret_cast<int32_t *>(dev + L.status) + wpf * n7 + k;
    if (n6) {
        std::memcpy(img + L.wg_tab, wg_tab.data(), sizeof(uint32_t) * wg_tab.size());
        std::memcpy(img + L.plans6, B.p6.data(), sizeof(Plan6) * n6);
        std::memset(img + L.tickets, 0, sizeof(uint32_t) * TICKET_STRIDE6 * n6);
    }

    // Type-7 frames need no upload: k7_side reads their plans straight from this pinned image and every status
    // word of theirs is written by a plain store.  The legacy kernels take their tables (and zeroed status words) from HBM.
    static const bool upload_plans = std::getenv("MCRAW_PLAN_UPLOAD") != nullptr; // timing experiment: plans through HBM
    if (n6 || upload_plans)
        HIP_TRY(hipMemcpyAsync(dev, img, L.upload_bytes, hipMemcpyHostToDevice, st));

    // ---- launches -----------------------------------------------------------
    // A batch that holds both encodings (BASELINE config 4): the two codecs share nothing, and k7_side is a handful of
    // latency-bound workgroups -- the legacy kernel runs beside the type-7 kernels on the context's second stream, forked
    // behind the table upload and joined in front of whatever the caller queues next.
    static const bool no_fork = std::getenv("MCRAW_NO_FORK") != nullptr; // timing experiment: one stream
    const bool both = n7 > 0 && n6 > 0 && s.fork && s.join && c->legacy && !no_fork;
    hipStream_t st6 = st;
    if (both) {
        st6 = c->legacy;
        HIP_TRY(hipEventRecord(s.fork, st));
        HIP_TRY(hipStreamWaitEvent(st6, s.fork, 0));
    }
    if (n7) {
        Work7 W{};
        W.plans = reinterpret_cast<const Plan7 *>((upload_plans ? dev : img) + L.plans7); // pinned host memory, device-visible at the same address
        W.status = reinterpret_cast<int32_t *>(dev + L.status);
        W.frames = reinterpret_cast<Frame7 *>(dev + w_frames);
        W.nstatus = static_cast<uint32_t>(nstatus);
        W.nsplit[0] = static_cast<uint32_t>(nsplit[0]);
        W.nsplit[1] = static_cast<uint32_t>(nsplit[1]);
        W.sync = static_cast<uint64_t *>(s.side_sync.p);
        W.epoch = s.look_epoch;
        W.bits = dev + w_bits;
        W.refs = reinterpret_cast<uint16_t *>(dev + w_refs);
        W.grp_off = reinterpret_cast<uint32_t *>(dev + w_grp);
        W.rpos = wpf > 2 ? reinterpret_cast<uint16_t *>(dev + w_rpos) : nullptr;
        // (one workgroup of k7_side per CU at most: 16 x 12 MP 14-bit noise 162 -> 108 us, natural 87 -> 72; 120 x 8K with 4 + 1
        // parts 164 -> 206 us -- tools/ab_side.sh)
        static const int lastc_env = []() { const char *e = std::getenv("MCRAW_SIDE_LASTC"); return e ? std::atoi(e) : -1; }();
        W.side_lastc = lastc_env >= 0 ? static_cast<uint32_t>(lastc_env) : (static_cast<long>(n7) * wpf <= 256 ? 1u : 0u);
        W.Rmax = static_cast<uint32_t>(Rmax);
        W.n7 = n7;
        W.post = c->post;
        // How k7_tiles' workgroups are dealt to the eight XCDs: in runs of 128 workgroups (2 MiB of output: the eight write
        // streams of a moment sit 2 MiB apart) or the grid in eight parts (one per XCD, a few hundred megabytes apart).
        // Neither is the faster one everywhere: the same launch takes 0.96 - 1.04 ms with the one and 0.97 - 1.01 ms with
        // the other, from box to box and -- for the eight parts -- from one process to the next on one box: the streams
        // meet on memory channels or not, as the physical pages of the caller's buffers fall (runs of 8 MiB are the slow
        // case every time).  So large resident batches measure: the first launches of a geometry take turns between
        // events, the faster candidate stays, and one launch in 64 re-checks it (tune_pick).  MCRAW_XCD_CHUNK pins the
        // choice (0: eight parts, 1: blockIdx order, n: runs of n).
        static const int xcd_env = []() {
            const char *e = std::getenv("MCRAW_XCD_CHUNK");
            return e ? std::atoi(e) : -1;
        }();
        const bool tunable = xcd_env < 0 && !dev_in && !geom_override && n7 >= 32;
        int tune_cand = -1;
        uint32_t xcd_chunk = xcd_env >= 0 ? static_cast<uint32_t>(xcd_env) : 128u;
        if (tunable) {
            tune_cand = tune_pick(c, n7, static_cast<uint32_t>(Rmax), c->post.mode);
            xcd_chunk = TUNE_CHUNKS[tune_cand >= 0 ? tune_cand : std::max(c->tunes[c->tune_last].decided, 0)];
        }
        W.xcd_chunk = xcd_chunk;
        W.nclasses = nclasses;
        for (uint32_t k = 0; k <= nclasses; k++)
            W.class_first[k] = class_first[k];
        for (uint32_t k = 0; k < nclasses; k++)
            W.class_groups[k] = class_groups[k];
        for (uint32_t stage : {MCRAW_K7_SIDE, MCRAW_K7_TILES}) {
            hipEvent_t ta = nullptr, tb = nullptr;
            const bool time_side = stage == MCRAW_K7_SIDE && side_cand >= 0;
            hipStream_t kst = stage == MCRAW_K7_SIDE && sst ? sst : st; // (the side stream's kernels follow each other on it)
            if ((stage == MCRAW_K7_TILES && tune_cand >= 0) || time_side) {
                ta = get_event(c);
                tb = get_event(c);
                if (ta && tb)
                    (void)hipEventRecord(ta, kst);
            }
            {
                KTimer t(c, static_cast<int>(stage), kst);
                launch_k7(W, stage, kst, stage == MCRAW_K7_SIDE && (all_thin || (sst != nullptr && !no_thin)));
            }
            if (ta && tb) {
                (void)hipEventRecord(tb, kst);
                if (time_side)
                    c->side_tunes[c->side_last].pending.push_back({ta, tb, side_cand});
                else
                    c->tunes[c->tune_last].pending.push_back({ta, tb, tune_cand});
            }
            if (stage == MCRAW_K7_SIDE && sst) { // the tile kernel needs what k7_side leaves in the slot's workspace
                HIP_TRY(hipEventRecord(s.side_done, sst));
                HIP_TRY(hipStreamWaitEvent(st, s.side_done, 0));
            }
        }
    }
    if (n6) {
        const Plan6 *dp = reinterpret_cast<const Plan6 *>(dev + L.plans6);
        Look6 lk;
        lk.res = static_cast<uint64_t *>(s.look.p);
        lk.ex = lk.res + static_cast<size_t>(smax) * n6;
        lk.hm = lk.ex + static_cast<size_t>(smax) * n6;
        KTimer t(c, MCRAW_K6_DECODE, st6);
        launch_k6_decode(dp, reinterpret_cast<const uint32_t *>(dev + L.wg_tab), n6 > 1 ? wg_tab[1] : wg_tab[n6], wg_tab[n6], lk,
                         reinterpret_cast<uint32_t *>(dev + L.tickets), s.look_epoch, n6, smax, c->post, st6);
    }
    if (both) {
        HIP_TRY(hipEventRecord(s.join, st6));
        HIP_TRY(hipStreamWaitEvent(st, s.join, 0));
    }
    HIP_TRY(hipGetLastError());
    *status_off = L.status;
    return 0;
}

int fetch_status(mcraw_ctx *c, Slot &s, size_t status_off, int n, hipStream_t st, int32_t *status, uint32_t *encH = nullptr, bool sent = false);

// Bring the statuses of one sub-batch home (they live in its slot's arena) and wait for its downloads:
// the slot is free afterwards.
int drain_part(mcraw_ticket *t, int idx)
{
    Part &p = t->parts[idx];
    if (p.drained)
        return 0;
    Slot &s = t->c->slots[p.slot];
    if (p.sent) {
        HIP_TRY(hipEventSynchronize(s.done)); // its downloads, queued on the download stream behind its kernels and its status words
        if (int rc = fetch_status(t->c, s, p.status_off, p.count, s.stream, t->status.data() + p.first, t->encH.data() + p.first, true))
            return rc;
    } else {
        if (int rc = fetch_status(t->c, s, p.status_off, p.count, s.stream, t->status.data() + p.first, t->encH.data() + p.first))
            return rc;
        HIP_TRY(hipEventSynchronize(s.done)); // its downloads, queued on the download stream
    }
#ifdef MCRAW_TIMELINE
    {
        float b = 0, u = 0, d = 0, e = 0;
        (void)hipEventElapsedTime(&b, t->c->tl0, s.tl_begin);
        (void)hipEventElapsedTime(&u, t->c->tl0, s.uploaded);
        (void)hipEventElapsedTime(&d, t->c->tl0, s.decoded);
        (void)hipEventElapsedTime(&e, t->c->tl0, s.done);
        const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t->c->tl_host0).count();
        std::fprintf(stderr, "[tl] slot %2d frames %3d+%d  queued (host) %8.3f | upload from %8.3f to %8.3f, decoded %8.3f, downloaded %8.3f | drained (host) %8.3f\n",
                     p.slot, p.first, p.count, s.tl_host, b, u, d, e, now);
    }
#endif
    s.busy = false;
    s.owner = nullptr;
    p.drained = true;
    return 0;
}

int resolve_device(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, hipStream_t st, int32_t *status, uint32_t *encH);

// A finished device-memory batch that nobody asked the statuses of: frames whose header wants more than
// the plan gave them are decoded now (second plan on the context's own stream).
int settle_slot(mcraw_ctx *c, Slot &s, std::vector<int32_t> *keep)
{
    HIP_TRY(hipEventSynchronize(s.done));
    int rc = 0;
    if (s.unresolved) {
        const int n = static_cast<int>(s.frames.size());
        std::vector<int32_t> status(n);
        std::vector<uint32_t> encH(n, 0);
        const Post now = c->post;
        c->post = s.post; // a frame planned again gets the post stage its batch was submitted with
        rc = resolve_device(c, s, s.frames.data(), n, c->aux, status.data(), encH.data());
        c->post = now;
        if (rc == 0) {
            for (int32_t v : status)
                c->sticky |= v;
            c->settled.emplace_back(s.serial, status);
            if (c->settled.size() > 64)
                c->settled.pop_front();
        }
        if (keep)
            *keep = status;
        s.unresolved = false;
        s.frames.clear();
    }
    s.busy = false;
    return rc;
}

int acquire_slot(mcraw_ctx *c, Slot **out, bool device_batch = false)
{
    Slot &s = device_batch ? c->dslots[c->next_dslot] : c->slots[c->next_slot];
    if (device_batch)
        c->next_dslot = (c->next_dslot + 1) % NDSLOT;
    else
        c->next_slot = (c->next_slot + 1) % NSLOT;
    if (s.busy) {
        if (s.owner) { // an asynchronous batch still keeps its statuses here
            if (int rc = drain_part(s.owner, s.owner_part))
                return rc;
        } else if (device_batch) {
            const bool last = c->last_slot == static_cast<int>(&s - c->dslots);
            if (int rc = settle_slot(c, s, last ? &c->last_status : nullptr))
                return rc;
            if (last)
                c->last_slot = -1; // its statuses are kept in last_status
        } else {
            HIP_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
    }
    *out = &s;
    return 0;
}

// Fetch statuses of a finished-or-running batch (synchronises on the stream); `encH`: the coded
// height of every type-7 frame, from its header (rows written = min(height, encH), RawData.cpp:571, :611).
// `sent`: the words are in s.status_host already (send_status below, and the caller has waited for what was queued behind it).
int fetch_status(mcraw_ctx *c, Slot &s, size_t status_off, int n, hipStream_t st, int32_t *status, uint32_t *encH, bool sent)
{
    const int ndev = static_cast<int>(s.order.size()), n7 = s.n7, n6 = ndev - n7, w7 = s.wpf;
    const size_t nstatus = static_cast<size_t>(w7) * n7 + n6 + 1;
    const size_t words = nstatus + n7; // statuses (w7 per type-7 frame, one per legacy frame, one spare), coded heights
    if (!sent) {
        if (int rc = ensure(s.status_host, sizeof(int32_t) * words, true))
            return rc;
        if (ndev)
            HIP_TRY(hipMemcpyAsync(s.status_host.p, static_cast<uint8_t *>(s.arena.p) + status_off, sizeof(int32_t) * words,
                                   hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    const int32_t *dev = static_cast<const int32_t *>(s.status_host.p);
    for (int i = 0; i < n && i < static_cast<int>(s.host_status.size()); i++)
        status[i] = s.host_status[i];
    for (int j = 0; j < ndev; j++)
        if (s.order[j] < n) {
            int32_t v = 0;
            if (j < n7)
                for (int w = 0; w < w7; w++)
                    v |= dev[w7 * j + w];
            else
                v = dev[w7 * n7 + (j - n7)];
            status[s.order[j]] |= v;
        }
    if (encH)
        for (int j = 0; j < n7; j++)
            if (s.order[j] < n)
                encH[s.order[j]] = static_cast<uint32_t>(dev[nstatus + j]);
    (void)c;
    return 0;
}

// Host-memory pipeline: the sub-batch's status words go home behind its kernels, on the same stream, written into pinned host
// memory by a kernel of one workgroup.  (Fetched with a copy only when the batch is waited for, they are queued on the copy engine
// behind whatever the NEXT batch has put there, and the wait for batch A ends when batch B's downloads do: tools/timeline_host.sh;
// a stream of ticketed batches then runs no faster than synchronous calls.  Sent with a copy of their own at submit time -- on the
// slot's stream or on the download lane -- they take the engine the frames' download would have had, and with a few batches
// queued every other download or so runs on one that moves 13 GB/s.)
__global__ void k_words_home(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n)
{
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int send_status(Slot &s, size_t status_off, hipStream_t st)
{
    const int ndev = static_cast<int>(s.order.size()), n7 = s.n7, n6 = ndev - n7, w7 = s.wpf;
    const size_t words = static_cast<size_t>(w7) * n7 + n6 + 1 + n7;
    if (int rc = ensure(s.status_host, sizeof(int32_t) * words, true))
        return rc;
    if (ndev) {
        hipLaunchKernelGGL(k_words_home, dim3(1), dim3(256), 0, st, static_cast<uint32_t *>(s.status_host.p),
                           reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(s.arena.p) + status_off), static_cast<uint32_t>(words));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

size_t written_of(const mcraw_frame &f, int32_t status, uint32_t encH)
{
    if (status != 0)
        return 0;
    if (f.type == MCRAW_TYPE_BLOCK) // width * min(height, encodedHeight): RawData.cpp:611 when they agree
        return static_cast<size_t>(f.width) * std::min<size_t>(static_cast<size_t>(f.height), encH);
    return static_cast<size_t>(f.width) * static_cast<size_t>(f.height); // RawData_Legacy.cpp:494
}

// Statuses a caller sees carry no internal bits.
inline int32_t public_status(int32_t st)
{
    return (st & E_GEOMETRY) ? ((st & ~E_GEOMETRY) | MCRAW_E_HEADER) : st;
}

// Statuses (and coded heights) of the device-memory batch in slot `s`, synchronising on `st`.  Frames
// whose header describes more blocks than they were planned with (the caller's width x height is a
// window of a larger coded frame: RawData.cpp takes the geometry from the header alone, :545-554)
// are planned again from the real header and decoded on `st` before this returns.
int resolve_device(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, hipStream_t st, int32_t *status, uint32_t *encH)
{
    if (int rc = fetch_status(c, s, s.status_off, n, st, status, encH))
        return rc;
    std::vector<int> redo;
    for (int i = 0; i < n; i++)
        if (status[i] & E_GEOMETRY)
            redo.push_back(i);
    if (!redo.empty()) {
        std::vector<mcraw_frame> rf(redo.size());
        std::vector<Geom7> rg(redo.size());
        std::vector<uint32_t> hdr(4 * redo.size(), 0u);
        for (size_t k = 0; k < redo.size(); k++) {
            rf[k] = frames[redo[k]];
            HIP_TRY(hipMemcpyAsync(&hdr[4 * k], rf[k].in, 16, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t k = 0; k < redo.size(); k++)
            rg[k] = {hdr[4 * k], hdr[4 * k + 1]};
        Slot &s2 = c->rslot;
        if (int rc = submit(c, s2, rf.data(), static_cast<int>(rf.size()), &rg, nullptr, nullptr, st, &s2.status_off))
            return rc;
        std::vector<int32_t> st2(rf.size());
        std::vector<uint32_t> eh2(rf.size(), 0u);
        if (int rc = fetch_status(c, s2, s2.status_off, static_cast<int>(rf.size()), st, st2.data(), eh2.data()))
            return rc;
        for (size_t k = 0; k < redo.size(); k++) {
            status[redo[k]] = st2[k];
            encH[redo[k]] = eh2[k];
        }
    }
    for (int i = 0; i < n; i++)
        status[i] = public_status(status[i]);
    return 0;
}

// Decode a batch whose buffers are in HBM.  With `written` / `status_out` the call synchronises and
// resolves everything; without, frames that need a second plan get it in mcraw_ctx_synchronize (or when
// the slot comes round again).
int decode_device(mcraw_ctx *c, const mcraw_frame *frames, int n, hipStream_t user, size_t *written, int32_t *status_out)
{
    // The batch in flight before this one (if any): is it still running?  Asked BEFORE acquire_slot may wait for an older one.
    bool busy_before = false;
    if (!user && c->side && c->last_slot >= 0 && c->dslots[c->last_slot].busy) {
        busy_before = hipEventQuery(c->dslots[c->last_slot].done) == hipErrorNotReady;
        (void)hipGetLastError();
    }
    Slot *sp = nullptr;
    if (int rc = acquire_slot(c, &sp, true))
        return rc;
    Slot &s = *sp;
    hipStream_t st = user, side_st = nullptr;
    if (!user) {
        // The context's own stream: nothing of the caller's can be ordered against it except through the host, so the frames'
        // inputs are complete now -- k7_side of this batch may start while the batch in front of it is still unpacking.  It
        // goes to the side stream when it has that company to hide behind; alone on an idle chip its chains are done sooner
        // in line with the tile kernel, as fat workgroups.
        st = c->stream;
        if (c->side && busy_before)
            side_st = c->side;
        if (c->tmain) { // (CU partition, an experiment: batches of the current encoding alone leave k7_side's CUs alone)
            bool all7 = true;
            for (int i = 0; i < n && all7; i++)
                all7 = frames[i].type == MCRAW_TYPE_BLOCK;
            if (all7)
                st = c->tmain;
            if (c->last_own && c->last_own != st) { // (the two streams take turns only when the kind of batch changes)
                HIP_TRY(hipEventRecord(c->chain, c->last_own));
                HIP_TRY(hipStreamWaitEvent(st, c->chain, 0));
            }
            c->last_own = st;
        }
    }
    if (int rc = submit(c, s, frames, n, nullptr, nullptr, nullptr, st, &s.status_off, side_st))
        return rc;
    HIP_TRY(hipEventRecord(s.done, st));
    s.busy = true;
    c->last_slot = static_cast<int>(sp - c->dslots);
    c->last_n = n;
    c->last_status.clear();
    s.serial = ++c->serial;
    if (!written && !status_out) {
        s.frames.assign(frames, frames + n);
        s.post = c->post;
        s.unresolved = true;
        return 0;
    }
    s.unresolved = false;
    std::vector<int32_t> status(n);
    std::vector<uint32_t> encH(n, 0u);
    if (int rc = resolve_device(c, s, frames, n, st, status.data(), encH.data()))
        return rc;
    c->last_status = status;
    for (int i = 0; i < n; i++) {
        if (status_out)
            status_out[i] = status[i];
        if (written)
            written[i] = written_of(frames[i], status[i], encH[i]);
    }
    return 0;
}

// What submit() rejects on the host before any device work (MCRAW_E_ARGS).
inline bool frame_args_ok(const mcraw_frame &f, const void *in, const void *out)
{
    return in && out && f.width > 0 && f.height > 0 && f.len != 0 && f.len < (1ull << 32) &&
           (f.type == MCRAW_TYPE_BLOCK || f.type == MCRAW_TYPE_LEGACY) && reinterpret_cast<uintptr_t>(out) % 2 == 0 &&
           static_cast<uint64_t>(f.width) * static_cast<uint64_t>(f.height) < (1ull << 31);
}

// Coded geometry from the 16-byte frame header (RawData.cpp:500-524) when the host can read it; zeros
// (= plan from width x height) when it is not a header a frame could decode with (:547-554).
inline Geom7 header_geometry(const mcraw_frame &f)
{
    Geom7 g{0u, 0u};
    if (f.type != MCRAW_TYPE_BLOCK || !f.in || f.len < 16 || f.width <= 0)
        return g;
    uint32_t h[2];
    std::memcpy(h, f.in, 8);
    if (h[0] == 0u || h[1] == 0u || (h[0] & 63u) || (h[1] & 3u) || h[0] < static_cast<uint32_t>(f.width) ||
        static_cast<uint64_t>(h[0]) * h[1] >= (1ull << 31))
        return g;
    // a header is untrusted input: N = encW * encH / 64 blocks need two side streams of ceil(N / 64) records of at least two
    // bytes each, behind their 4-byte counts and the 16-byte header (RawData.cpp:463-498) -- a frame buffer shorter than that
    // cannot hold the geometry it claims, and gets no workspace for it (it is planned from width x height, and k7_side then
    // rejects its header)
    const uint64_t nrecords = (static_cast<uint64_t>(h[0]) * h[1] / 64u + GROUP_BLOCKS - 1u) / GROUP_BLOCKS;
    if (16u + 2u * (4u + 2u * nrecords) > f.len)
        return g;
    g.encW = h[0];
    g.encH = h[1];
    return g;
}

// Host-memory batch, cut into sub-batches that flow through three lanes: every upload on one stream,
// the kernels of a sub-batch on its slot's stream, every download on a third stream, chained by
// events -- so each copy engine runs back to back over the sub-batches while the kernels of the next
// one execute (BASELINE config 3: "pinned H2D + decode overlapped on HIP streams").  With the copies
// of a sub-batch on its slot's own stream (first version) the engines idled between sub-batches:
// 2 150 instead of 2 630 UHD frames/s.
// The frame headers are in host memory here, so every frame is planned from its real geometry.
int host_submit_part(mcraw_ticket *t, int first, int count)
{
    mcraw_ctx *c = t->c;
    const mcraw_frame *frames = t->frames.data();
    Slot *sp = nullptr;
    if (int rc = acquire_slot(c, &sp))
        return rc;
    Slot &s = *sp;
    hipStream_t st = s.stream;
    // Device staging mirrors the host layout wherever frames are neighbours in host memory (inputs: up
    // to 256 bytes apart; outputs: exactly adjacent, a copy must not touch bytes between two buffers):
    // such a run moves with ONE copy per direction -- a copy call costs about 6 us, which is what a
    // stream of small frames would otherwise be bound by.  Offsets keep the host address modulo 256.
    // Neighbours are assumed to belong to one allocation (the usual case: slices of one pinned buffer);
    // where the runtime refuses a merged copy (hipErrorInvalidValue: it spans two allocations) the run
    // is copied frame by frame instead.
    struct Run {
        uintptr_t host;
        size_t bytes, dev;
        int first, last; // frames of the run (sub-batch indices)
    };
    std::vector<Run> rin, rout;
    std::vector<size_t> in_off(count, SIZE_MAX), out_off(count, SIZE_MAX), out_len(count, 0);
    std::vector<Geom7> geom(count);
    size_t io = 0, oo = 0;
    uintptr_t lay_host_end = 0;
    size_t lay_dev_end = 0;
    bool lay_ok = false;
    for (int i = 0; i < count; i++) {
        const mcraw_frame &f = frames[first + i];
        geom[i] = header_geometry(f);
        if (!frame_args_ok(f, f.in, f.out))
            continue; // rejected by submit() with MCRAW_E_ARGS: nothing is staged, nothing is copied
        const uintptr_t a = reinterpret_cast<uintptr_t>(f.in);
        if (!rin.empty() && a >= rin.back().host + rin.back().bytes && a - (rin.back().host + rin.back().bytes) <= 256) {
            in_off[i] = rin.back().dev + (a - rin.back().host);
            rin.back().bytes = a + f.len - rin.back().host;
            rin.back().last = i;
        } else {
            const size_t dev = up(io, ALIGN) + (a & (ALIGN - 1));
            rin.push_back({a, f.len, dev, i, i});
            in_off[i] = dev;
        }
        io = rin.back().dev + rin.back().bytes;
        out_len[i] = std::min(f.out_capacity * 2, static_cast<size_t>(f.height) * post_row_bytes(static_cast<uint32_t>(f.width), c->post.mode));
        // the kernels may write a whole frame even when the caller's capacity is smaller (that frame then
        // fails with MCRAW_E_CAPACITY before any kernel runs): reserve the full size on the device
        const size_t full = std::max(out_len[i], static_cast<size_t>(f.width) * f.height * 2);
        const uintptr_t ao = reinterpret_cast<uintptr_t>(f.out);
        if (lay_ok && ao == lay_host_end)
            out_off[i] = lay_dev_end; // adjacent in host memory: adjacent in the staging too
        else
            out_off[i] = up(oo, ALIGN) + (ao & (ALIGN - 1));
        lay_ok = full == out_len[i]; // nothing behind this frame's bytes in the staging
        lay_host_end = ao + out_len[i];
        lay_dev_end = out_off[i] + out_len[i];
        oo = std::max(oo, out_off[i] + full);
    }
    if (int rc = ensure(s.dev_in, io + ALIGN, false))
        return rc;
    if (int rc = ensure(s.dev_out, oo + ALIGN, false))
        return rc;
    std::vector<const uint8_t *> din(count);
    std::vector<uint16_t *> dout(count);
    for (int i = 0; i < count; i++) {
        din[i] = in_off[i] != SIZE_MAX ? static_cast<uint8_t *>(s.dev_in.p) + in_off[i] : nullptr;
        dout[i] = out_off[i] != SIZE_MAX ? reinterpret_cast<uint16_t *>(static_cast<uint8_t *>(s.dev_out.p) + out_off[i]) : nullptr;
    }
#ifdef MCRAW_TIMELINE
    if (!c->tl0) {
        HIP_TRY(hipEventCreate(&c->tl0));
        HIP_TRY(hipEventRecord(c->tl0, c->h2d));
        c->tl_host0 = std::chrono::steady_clock::now();
    }
    HIP_TRY(hipEventRecord(s.tl_begin, c->h2d));
    s.tl_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->tl_host0).count();
#endif
    for (const Run &r : rin) {
        hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(s.dev_in.p) + r.dev, reinterpret_cast<const void *>(r.host), r.bytes,
                                      hipMemcpyHostToDevice, c->h2d);
        if (e == hipErrorInvalidValue && r.last > r.first) {
            (void)hipGetLastError();
            for (int i = r.first; i <= r.last; i++)
                if (in_off[i] != SIZE_MAX)
                    HIP_TRY(hipMemcpyAsync(static_cast<uint8_t *>(s.dev_in.p) + in_off[i], frames[first + i].in, frames[first + i].len,
                                           hipMemcpyHostToDevice, c->h2d));
        } else
            HIP_TRY(e);
    }
    // three lanes: all uploads queue on one stream, all downloads on another (each copy engine then
    // runs back to back over the sub-batches), the kernels of a sub-batch on its slot's stream between
    HIP_TRY(hipEventRecord(s.uploaded, c->h2d));
    HIP_TRY(hipStreamWaitEvent(st, s.uploaded, 0));
    size_t status_off = 0;
    if (int rc = submit(c, s, frames + first, count, &geom, din.data(), dout.data(), st, &status_off))
        return rc;
    if (t->send)
        if (int rc = send_status(s, status_off, st))
            return rc;
    HIP_TRY(hipEventRecord(s.decoded, st));
    HIP_TRY(hipStreamWaitEvent(c->d2h, s.decoded, 0));
    // downloads: only frames the host has not rejected (a rejected frame's buffer stays untouched; the
    // content of a buffer whose frame fails on the device is undefined)
    for (int i = 0; i < count; i++) {
        if (out_off[i] == SIZE_MAX || s.host_status[i] != 0 || out_len[i] == 0)
            continue;
        const uintptr_t a = reinterpret_cast<uintptr_t>(frames[first + i].out);
        if (!rout.empty() && rout.back().last == i - 1 && a == rout.back().host + rout.back().bytes &&
            out_off[i] == rout.back().dev + rout.back().bytes) {
            rout.back().bytes += out_len[i];
            rout.back().last = i;
        } else
            rout.push_back({a, out_len[i], out_off[i], i, i});
    }
    for (const Run &r : rout) {
        hipError_t e = hipMemcpyAsync(reinterpret_cast<void *>(r.host), static_cast<uint8_t *>(s.dev_out.p) + r.dev, r.bytes,
                                      hipMemcpyDeviceToHost, c->d2h);
        if (e == hipErrorInvalidValue && r.last > r.first) {
            (void)hipGetLastError();
            for (int i = r.first; i <= r.last; i++)
                HIP_TRY(hipMemcpyAsync(frames[first + i].out, static_cast<uint8_t *>(s.dev_out.p) + out_off[i], out_len[i],
                                       hipMemcpyDeviceToHost, c->d2h));
        } else
            HIP_TRY(e);
    }
    HIP_TRY(hipEventRecord(s.done, c->d2h));
    s.busy = true;
    s.landed = false;
    s.seq = ++c->part_seq;
    // the slot keeps this sub-batch's statuses until they are drained into the ticket: by
    // mcraw_ticket_wait, or earlier by acquire_slot when the ring comes round (more sub-batches in
    // flight than slots)
    s.owner = t;
    s.owner_part = static_cast<int>(t->parts.size());
    t->parts.push_back({static_cast<int>(sp - c->slots), first, count, status_off, false, t->send});
    return 0;
}

// Is this context the only one of the process on its device?  (Else: the long way for every batch.)
inline bool alone_on_device(const mcraw_ctx *c) { return !c->counted || g_ctx_on_device[c->device].load() <= 1; }

// Queue a host-memory batch (ticket->frames): returns when the last sub-batch is submitted.
int host_submit(mcraw_ticket *t)
{
    mcraw_ctx *c = t->c;
    const mcraw_frame *frames = t->frames.data();
    const int n = static_cast<int>(t->frames.size());
    constexpr size_t SUB_BYTES = 96ull << 20; // compressed + decoded bytes per sub-batch (64-160 MB measure within 3 %)
    t->status.assign(n, 0);
    t->encH.assign(n, 0u);
    t->post = c->post;
    // Workspace of a sub-batch: every type-7 frame gets the stride of the largest one (the kernels address it from (frame,
    // group) alone), so one large frame among many small ones -- or one header that claims a large geometry -- must not be
    // multiplied by the frames around it: a sub-batch is also closed when that product passes WS_BUDGET.
    constexpr size_t WS_BUDGET = 1ull << 30;
    auto groups_of = [&](int i) -> size_t {
        const mcraw_frame &f = frames[i];
        if (f.type != MCRAW_TYPE_BLOCK || !frame_args_ok(f, f.in, f.out))
            return 0;
        const Geom7 g = header_geometry(f);
        const uint64_t encW = g.encW ? g.encW : up(static_cast<size_t>(f.width), 64), encH = g.encW ? g.encH : up(static_cast<size_t>(f.height), 4);
        return static_cast<size_t>((encW * encH / 64u + GROUP_BLOCKS - 1u) / GROUP_BLOCKS);
    };
    constexpr size_t WS_PER_GROUP = 64u * 3u + 4u * ITEM_SPLIT; // bits (u8) + refs (u16) per block, one offset per item
    // the sub-batch that starts at frame `first`
    auto cut = [&](int first) {
        size_t bytes = 0, gmax = 0;
        int count = 0, n7 = 0;
        while (first + count < n) {
            const mcraw_frame &f = frames[first + count];
            const size_t fb = frame_args_ok(f, f.in, f.out) ? f.len + static_cast<size_t>(f.width) * f.height * 2 : 0;
            const size_t g = groups_of(first + count);
            const size_t gm = std::max(gmax, g);
            if (count > 0 && (bytes + fb > SUB_BYTES || gm * WS_PER_GROUP * static_cast<size_t>(n7 + (g ? 1 : 0)) > WS_BUDGET))
                break;
            bytes += fb;
            gmax = gm;
            n7 += g ? 1 : 0;
            count++;
        }
        return count;
    };
    // A batch of a few sub-batches -- a caller that streams tickets, the facade's chunks -- goes the short way: its status
    // words come home behind its kernels (send_status), so waiting for it is waiting for ITS downloads, and the next ticket's
    // uploads run beside them (7-frame UHD tickets, two in flight: 2 560 -> 2 990 frames/s; tools/bench_tickets.py).  That
    // way works while little is queued: with four tickets in flight, or more sub-batches than the ring has slots, the
    // downloads fall to a quarter of their rate (13 GB/s; the runtime's choice of copy engine is the suspect), where the long
    // way -- statuses fetched when the batch is waited for, which queues that fetch behind everything submitted since and so
    // lets the ring run empty now and then -- keeps 2 700-2 900.  So: batches of up to SHORT_PARTS sub-batches are scheduled the
    // short way (queued only when at most ONE other batch still has downloads under way; larger batches are dealt out as such
    // by deal_host), and whether their status words are sent home is the caller's word (`want_send`: big_way / the ticket rows
    // of mcraw_decode_batch_async measure what is faster in this process).
    constexpr int SHORT_PARTS = 6;
    {
        int parts = 0;
        for (int f = 0; f < n && parts <= SHORT_PARTS; parts++)
            f += cut(f);
        t->small = parts <= SHORT_PARTS && alone_on_device(c);
        t->send = t->small && t->want_send == 1;
    }
    while (t->small) {
        int others = 0;
        Slot *oldest = nullptr;
        const ::mcraw_ticket *seen[NSLOT];
        for (Slot &x : c->slots)
            if (x.busy && !x.landed && x.owner && x.owner != t) {
                bool dup = false;
                for (int k = 0; k < others; k++)
                    dup = dup || seen[k] == x.owner;
                if (!dup)
                    seen[others++] = x.owner;
                if (!oldest || x.seq < oldest->seq)
                    oldest = &x;
            }
        if (others <= 1)
            break;
        HIP_TRY(hipEventSynchronize(oldest->done));
        oldest->landed = true;
    }
    int first = 0;
    while (first < n) {
        int count = cut(first);
        int rc = host_submit_part(t, first, count);
        // out of device memory: halve the sub-batch; a single frame that cannot get its workspace fails alone
        // (the failed attempt may have queued uploads from the caller's buffers into a slot that no part of the ticket owns:
        // they are waited for here, so that no copy can still be reading a buffer when the ticket is reported done)
        while (rc == -static_cast<int>(hipErrorOutOfMemory) && count > 1) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->h2d);
            count = (count + 1) / 2;
            rc = host_submit_part(t, first, count);
        }
        if (rc == -static_cast<int>(hipErrorOutOfMemory)) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->h2d);
            t->status[first] |= MCRAW_E_DEVICE;
            t->skipped.push_back(first);
            rc = 0;
        }
        if (rc) {
            // nothing of this batch may still be moving when the caller hears of the failure (it may free its buffers)
            (void)hipStreamSynchronize(c->h2d);
            for (Part &p : t->parts) {
                (void)hipEventSynchronize(c->slots[p.slot].done);
                c->slots[p.slot].busy = false;
                c->slots[p.slot].owner = nullptr;
                p.drained = true;
            }
            (void)hipStreamSynchronize(c->d2h);
            return rc;
        }
        first += count;
    }
    return 0;
}

// Wait for a host-memory batch and resolve its statuses.
int host_finish(mcraw_ticket *t, size_t *written, int32_t *status_out)
{
    const mcraw_frame *frames = t->frames.data();
    const int n = static_cast<int>(t->frames.size());
    for (size_t k = 0; k < t->parts.size(); k++)
        if (int rc = drain_part(t, static_cast<int>(k)))
            return rc;
    for (int i : t->skipped)
        t->status[i] |= MCRAW_E_DEVICE;
    for (int i = 0; i < n; i++) {
        // every frame was planned from its real header (header_geometry), so no frame is left to plan again
        const int32_t st = public_status(t->status[i]);
        if (status_out)
            status_out[i] = st;
        if (written)
            written[i] = written_of(frames[i], st, t->encH[i]);
    }
    return 0;
}

// A ticket that goes away (finished, or failed half way) must not be pointed at by a slot.
void forget_ticket(mcraw_ticket *t)
{
    for (Slot &s : t->c->slots)
        if (s.owner == t) {
            s.owner = nullptr; // the slot stays busy until its `done` event: acquire_slot waits for it
            s.owner_part = -1;
        }
}

// A host-memory batch dealt out as a row of short batches (host_submit: up to SHORT_PARTS sub-batches each), two of them under
// way at a time -- the regime in which the copy lanes never drain and never crowd: 240 UHD frames in one call 2 750 -> 3 000
// frames/s host to host.  `finish`: wait for all of them (the synchronous call); else the last ones stay in `pieces` for
// land_pieces.  Results go to written / status_out (either may be null) at the frames' positions in the batch.
int land_piece(std::vector<std::unique_ptr<mcraw_ticket>> &pieces, std::vector<int> &piece_first, size_t *written, int32_t *status_out)
{
    mcraw_ticket *p = pieces.front().get(); // the oldest piece: wait, file its results
    const int first = piece_first.front();
    const int r = host_finish(p, written ? written + first : nullptr, status_out ? status_out + first : nullptr);
    forget_ticket(p);
    pieces.erase(pieces.begin());
    piece_first.erase(piece_first.begin());
    return r;
}

// Status words home behind their kernels (1), or fetched at the wait (0)?  In a process whose first GPU work was this context
// sending is 10 % faster for a large batch (2 960 against 2 680 UHD frames/s); behind one torch operation -- HIP hands a process four
// hardware queues per stream priority, and which of the context's streams share one depends on what existed before -- the small
// kernel that writes home makes sub-batch k + 1's upload wait for sub-batch k's download there (1 600 against 2 570).  Neither a
// probe on dummy buffers nor the first pieces of a batch show that (it sets in later), so whole batches are compared: of the
// batches of ten pieces or more the first one fetches and only warms the slots up, the second fetches, the third and the fourth
// send (the fourth is the one compared), and the faster way is the context's for large batches from then on (until then:
// fetched).  Streams of short tickets decide for themselves (mcraw_decode_batch_async: sending won wherever it was
// measured).  MCRAW_SHORT_WAY=0|1 decides both beforehand.
constexpr size_t PIECE_BYTES = 4 * (96ull << 20);

size_t host_bytes(const mcraw_frame *frames, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; i++)
        total += static_cast<size_t>(frames[i].len) + (frames[i].width > 0 && frames[i].height > 0 ? static_cast<size_t>(frames[i].width) * frames[i].height * 2 : 0);
    return total;
}

void way_from_env(mcraw_ctx *c)
{
    static const char *e = std::getenv("MCRAW_SHORT_WAY");
    if (e && (e[0] == '0' || e[0] == '1') && c->send_home < 0)
        c->send_home = c->send_home_tickets = e[0] - '0';
}

// The way of a batch of more than one piece; *trial: it is one of the two that are compared (big_way_result when it is over).
int big_way(mcraw_ctx *c, size_t total, bool *trial)
{
    way_from_env(c);
    *trial = c->send_home < 0 && alone_on_device(c) && total / PIECE_BYTES >= 10;
    if (c->send_home >= 0)
        return c->send_home;
    if (*trial && c->big_seen++ == 0) {
        *trial = false; // (the context's first large batch pays for the slots' buffers: fetched, and not compared)
        // ... and what the other way needs is made now, so that its trial batch does not pay for it: the slots' pinned status
        // buffers, the first launch of the kernel that writes into them
        for (Slot &x : c->slots)
            if (ensure(x.status_host, 4096, true) != 0)
                break;
        if (c->slots[0].status_host.p) {
            hipLaunchKernelGGL(k_words_home, dim3(1), dim3(64), 0, c->slots[0].stream, static_cast<uint32_t *>(c->slots[0].status_host.p),
                               static_cast<const uint32_t *>(c->slots[0].status_host.p), 0u);
            (void)hipStreamSynchronize(c->slots[0].stream);
        }
        (void)hipGetLastError();
    }
    return *trial && c->trial_rate[0] != 0.0 ? 1 : 0;
}

void big_way_result(mcraw_ctx *c, int way, size_t total, double seconds)
{
    if (c->send_home >= 0 || seconds <= 0)
        return;
    if (way == 1 && c->sent_trials++ == 0)
        return; // (the first batch that sends is its way's warm-up, as the context's first batch was the other's)
    c->trial_rate[way] = total / seconds;
    if (way == 1) {
        c->send_home = c->trial_rate[1] > c->trial_rate[0] * 1.03 ? 1 : 0;
        if (std::getenv("MCRAW_TRACE"))
            std::fprintf(stderr, "[mcraw] host-memory pipeline: status words fetched %.1f GB/s, sent home %.1f GB/s: %s from here on\n",
                         c->trial_rate[0] / 1e9, c->trial_rate[1] / 1e9, c->send_home ? "sent" : "fetched");
    }
}

int deal_host(mcraw_ctx *c, const mcraw_frame *frames, int n, size_t *written, int32_t *status_out,
              std::vector<std::unique_ptr<mcraw_ticket>> &pieces, std::vector<int> &piece_first, bool finish, int way)
{
    // (a piece is cut by bytes here and into sub-batches of up to 96 MB by host_submit, which ends one in front of the frame that
    // would not fit: four sub-batches' worth of bytes are five or six sub-batches, short by host_submit's count)
    const size_t piece = alone_on_device(c) ? PIECE_BYTES : SIZE_MAX;
    int rc = 0, first = 0;
    while (first < n && rc == 0) {
        size_t bytes = 0;
        int count = 0;
        while (first + count < n) {
            const mcraw_frame &f = frames[first + count];
            const size_t fb = static_cast<size_t>(f.len) + (f.width > 0 && f.height > 0 ? static_cast<size_t>(f.width) * f.height * 2 : 0);
            if (count > 0 && bytes + fb > piece)
                break;
            bytes += fb;
            count++;
        }
        std::unique_ptr<mcraw_ticket> p(new mcraw_ticket());
        p->c = c;
        p->frames.assign(frames + first, frames + first + count);
        p->want_send = way;
        rc = host_submit(p.get());
        if (rc != 0) { // (host_submit has waited for whatever it had queued of this piece)
            forget_ticket(p.get());
            break;
        }
        pieces.push_back(std::move(p));
        piece_first.push_back(first);
        first += count;
        if (pieces.size() >= 2 && (finish || first < n)) {
            rc = land_piece(pieces, piece_first, written, status_out);
        }
    }
    while (!pieces.empty() && (finish || rc != 0)) { // (behind a failure too: nothing of the batch may still be moving then)
        const int r = land_piece(pieces, piece_first, written, status_out);
        if (rc == 0)
            rc = r;
    }
    return rc;
}

// Synchronous host-memory batch.
int decode_host(mcraw_ctx *c, const mcraw_frame *frames, int n, size_t *written, int32_t *status_out)
{
    std::vector<std::unique_ptr<mcraw_ticket>> pieces;
    std::vector<int> piece_first;
    const size_t total = host_bytes(frames, n);
    bool trial = false;
    way_from_env(c);
    const int way = total > PIECE_BYTES ? big_way(c, total, &trial) : std::max(0, c->send_home_tickets);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = deal_host(c, frames, n, written, status_out, pieces, piece_first, true, way);
    if (trial && rc == 0)
        big_way_result(c, way, total, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

mcraw_ctx *g_default = nullptr;
std::mutex g_default_mu;

mcraw_ctx *default_ctx()
{
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        mcraw_ctx *c = nullptr;
        if (mcraw_ctx_create(-1, &c) != 0)
            return nullptr;
        g_default = c;
        std::atexit([]() { // the process-wide context of the five-argument entry points
            std::lock_guard<std::mutex> lk2(g_default_mu);
            mcraw_ctx_destroy(g_default);
            g_default = nullptr;
        });
    }
    return g_default;
}

size_t decode_one(int type, uint16_t *output, int width, int height, const uint8_t *input, size_t len)
{
    mcraw_ctx *c = default_ctx();
    if (!c)
        return 0;
    mcraw_frame f{};
    f.in = input;
    f.len = len;
    f.width = width;
    f.height = height;
    f.type = type;
    f.out = output;
    f.out_capacity = width > 0 && height > 0 ? static_cast<size_t>(width) * static_cast<size_t>(height) : 0;
    size_t written = 0;
    int32_t status = 0;
    if (mcraw_decode_batch(c, &f, 1, MCRAW_MEM_HOST, nullptr, &written, &status) != 0)
        return 0;
    return written;
}

} // namespace

// ------------------------------------------------------------------ C ABI

extern "C" {

const char *mcraw_last_error(void) { return g_err.c_str(); }

int mcraw_ctx_create(int device, mcraw_ctx **out)
{
    if (!out)
        return -1;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_err = "mcraw: no HIP device available (the decode path has no CPU fallback)";
        return e != hipSuccess ? -static_cast<int>(e) : -static_cast<int>(hipErrorNoDevice);
    }
    if (device < 0) {
        const char *env = std::getenv("MCRAW_DEVICE");
        if (env && *env)
            device = std::atoi(env);
        else
            HIP_TRY(hipGetDevice(&device));
    }
    if (device >= ndev) {
        g_err = "mcraw: device index out of range";
        return -static_cast<int>(hipErrorInvalidDevice);
    }
    HIP_TRY(hipSetDevice(device));
    // a half-built context is torn down again on any failure below
    struct Guard {
        mcraw_ctx *c;
        ~Guard() { if (c) mcraw_ctx_destroy(c); }
    } guard{new mcraw_ctx()};
    mcraw_ctx *c = guard.c;
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->legacy, hipStreamNonBlocking));
    {
        // k7_side of the next batch beside the tile kernel of the one in flight (see mcraw_ctx::side): MEASURED, NOT SHIPPED --
        // off unless MCRAW_SIDE_CUS says otherwise (docs/lab_notes.md, round 5, has the tables):
        //   -1   a side stream of the LOWEST priority -- a hardware queue of its own: streams of one priority share four queues,
        //        and two kernels in one queue never overlap -- on which k7_side runs as thin workgroups.  It hides (the step is
        //        the tile kernel + 10-16 us instead of + 57 us), and the tile kernel pays for it: k7_side's 480 thin workgroups
        //        hold a quarter of the chip's wave slots for 0.2 ms, the tile kernel runs 3-5 % longer, the step is where it was;
        //   r>0  r CUs of every XCD for k7_side alone and the rest for the other kernels, as CU masks of two streams (a queue's CU
        //        mask is dealt bit by bit to the XCDs: bit i is XCD i mod 8, then shader engine by shader engine, so the low 8 r
        //        bits are r CUs of every XCD).  k7_side hides completely -- and the tile kernel, which runs at the CUs' memory
        //        pipelines' rate, loses more than its share: +4.6 / +9 / +9 / +8 / +16 % with 8 / 16 / 24 / 32 / 48 CUs away.
        c->side_fat = std::getenv("MCRAW_SIDE_FAT") != nullptr;
        c->side_thin = std::getenv("MCRAW_SIDE_THIN") != nullptr;
        int per_xcd = 0;
        if (const char *e = std::getenv("MCRAW_SIDE_CUS"))
            per_xcd = std::atoi(e);
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount, nxcd = 8;
        if (per_xcd < 0) {
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            int ps = lo, pt = (lo + hi) / 2;
            if (const char *e = std::getenv("MCRAW_SIDE_PRIO")) // experiment: "s,t" = priorities of the side stream and of the other one
                (void)std::sscanf(e, "%d,%d", &ps, &pt);
            HIP_TRY(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, ps));
            // (the context's own stream is created again, beside it: as the stream it was, the first of the context's twenty-odd,
            // the tile kernel ran 8 % longer with k7_side beside it and k7_side did not hide -- tools/ab_env.sh, docs/lab_notes.md)
            hipStream_t own = nullptr;
            HIP_TRY(hipStreamCreateWithPriority(&own, hipStreamNonBlocking, pt));
            (void)hipStreamDestroy(c->stream);
            c->stream = own;
        } else if (per_xcd > 0 && ncu % 32 == 0 && per_xcd * nxcd * 2 <= ncu) {
            const int words = ncu / 32, r = per_xcd * nxcd;
            std::vector<uint32_t> ms(words, 0u), mt(words, 0xffffffffu);
            for (int b = 0; b < r; b++) {
                ms[b / 32] |= 1u << (b % 32);
                mt[b / 32] &= ~(1u << (b % 32));
            }
            hipError_t e1 = hipExtStreamCreateWithCUMask(&c->side, static_cast<uint32_t>(words), ms.data());
            hipError_t e2 = e1 == hipSuccess ? hipExtStreamCreateWithCUMask(&c->tmain, static_cast<uint32_t>(words), mt.data()) : e1;
            if (e1 != hipSuccess || e2 != hipSuccess) { // no partition on this runtime: everything on the one stream
                (void)hipGetLastError();
                if (c->side)
                    (void)hipStreamDestroy(c->side);
                c->side = c->tmain = nullptr;
            } else
                HIP_TRY(hipEventCreateWithFlags(&c->chain, hipEventDisableTiming));
        }
    }
    for (Slot *sp : {&c->rslot}) {
        HIP_TRY(hipEventCreateWithFlags(&sp->fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sp->join, hipEventDisableTiming));
    }
    for (Slot &s : c->slots) {
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
#ifdef MCRAW_TIMELINE // (tools/timeline_host.sh: when did every sub-batch's upload, kernels and download end on the GPU's clock?)
        HIP_TRY(hipEventCreate(&s.done));
        HIP_TRY(hipEventCreate(&s.uploaded));
        HIP_TRY(hipEventCreate(&s.decoded));
        HIP_TRY(hipEventCreate(&s.tl_begin));
#else
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming));
#endif
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    }
    for (Slot &s : c->dslots) {
        HIP_TRY(hipEventCreateWithFlags(&s.side_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.join, hipEventDisableTiming));
    }
    guard.c = nullptr;
    if (c->device >= 0 && c->device < 64) {
        g_ctx_on_device[c->device]++;
        c->counted = true;
    }
    *out = c;
    return 0;
}

void mcraw_ctx_destroy(mcraw_ctx *c)
{
    if (!c)
        return;
    if (c->counted)
        g_ctx_on_device[c->device]--;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    auto release = [](Slot &s) {
        if (s.pinned.p) (void)hipHostFree(s.pinned.p);
        if (s.status_host.p) (void)hipHostFree(s.status_host.p);
        if (s.arena.p) (void)hipFree(s.arena.p);
        if (s.look.p) (void)hipFree(s.look.p);
        if (s.side_sync.p) (void)hipFree(s.side_sync.p);
        if (s.dev_in.p) (void)hipFree(s.dev_in.p);
        if (s.dev_out.p) (void)hipFree(s.dev_out.p);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.fork) (void)hipEventDestroy(s.fork);
        if (s.join) (void)hipEventDestroy(s.join);
        if (s.side_done) (void)hipEventDestroy(s.side_done);
        if (s.uploaded) (void)hipEventDestroy(s.uploaded);
        if (s.decoded) (void)hipEventDestroy(s.decoded);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    };
    for (Slot &s : c->slots)
        release(s);
    for (Slot &s : c->dslots)
        release(s);
    release(c->rslot);
    for (KStat &k : c->kstat)
        for (auto &p : k.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto &t : c->tunes)
        for (auto &p : t.pending) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    for (auto &t : c->side_tunes)
        for (auto &p : t.pending) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
    for (hipEvent_t e : c->event_pool)
        (void)hipEventDestroy(e);
    if (c->chain)
        (void)hipEventDestroy(c->chain);
    if (c->side)
        (void)hipStreamDestroy(c->side);
    if (c->tmain)
        (void)hipStreamDestroy(c->tmain);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    if (c->aux)
        (void)hipStreamDestroy(c->aux);
    if (c->legacy)
        (void)hipStreamDestroy(c->legacy);
    if (c->h2d)
        (void)hipStreamDestroy(c->h2d);
    if (c->d2h)
        (void)hipStreamDestroy(c->d2h);
    delete c;
}

int mcraw_decode_batch(mcraw_ctx *c, const mcraw_frame *frames, int nframes, int mem, void *stream, size_t *written,
                       int32_t *status)
{
    if (!c || (!frames && nframes > 0) || nframes < 0) {
        g_err = "mcraw_decode_batch: bad arguments";
        return -1;
    }
    if (nframes == 0)
        return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (mem == MCRAW_MEM_DEVICE)
        return decode_device(c, frames, nframes, static_cast<hipStream_t>(stream), written, status);
    if (mem == MCRAW_MEM_HOST)
        return decode_host(c, frames, nframes, written, status);
    g_err = "mcraw_decode_batch: unknown memory kind";
    return -1;
}

int mcraw_decode_batch_async(mcraw_ctx *c, const mcraw_frame *frames, int nframes, mcraw_ticket **ticket)
{
    if (ticket)
        *ticket = nullptr;
    if (!c || !ticket || (!frames && nframes > 0) || nframes < 0) {
        g_err = "mcraw_decode_batch_async: bad arguments";
        return -1;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    mcraw_ticket *t = new mcraw_ticket();
    t->c = c;
    // (queued as a row of short batches: the call returns when the last of them is queued, as it did when the ring of slots was
    // shorter than the batch; what has landed by then is kept in the ticket)
    t->composite = true;
    t->got_written.assign(static_cast<size_t>(nframes), 0);
    t->got_status.assign(static_cast<size_t>(nframes), 0);
    constexpr int TRIAL_TICKETS = 12;
    way_from_env(c);
    const size_t total = host_bytes(frames, nframes);
    int way;
    if (total > PIECE_BYTES) { // a large batch as a ticket: compared like the synchronous ones, its time runs until it is waited for
        way = big_way(c, total, &t->big_trial);
        t->trial_bytes = total;
        t->t_queued = std::chrono::steady_clock::now();
    } else if (c->send_home_tickets >= 0) {
        way = c->send_home_tickets;
    } else if (nframes > 0 && alone_on_device(c) && c->tt.queued < TRIAL_TICKETS) {
        way = t->trial_way = c->tt.way; // (undecided: this ticket belongs to the row under way)
        t->trial_bytes = total;
        c->tt.queued++;
    } else {
        way = 0;
    }
    t->way = way;
    if (int rc = deal_host(c, frames, nframes, t->got_written.data(), t->got_status.data(), t->pieces, t->piece_first, false, way)) {
        delete t;
        return rc;
    }
    *ticket = t;
    return 0;
}

int mcraw_ticket_wait(mcraw_ticket *t, size_t *written, int32_t *status)
{
    if (!t)
        return -1;
    mcraw_ctx *c = t->c;
    int rc;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        rc = hipSetDevice(c->device) == hipSuccess ? 0 : -static_cast<int>(hipErrorInvalidDevice);
        while (!t->pieces.empty()) { // (whatever happens: every piece is waited for, the first failure is the one reported)
            const int r = land_piece(t->pieces, t->piece_first, t->got_written.data(), t->got_status.data());
            if (rc == 0)
                rc = r;
        }
        if (t->big_trial && rc == 0)
            big_way_result(c, t->way, t->trial_bytes, std::chrono::duration<double>(std::chrono::steady_clock::now() - t->t_queued).count());
        if (t->trial_way >= 0 && c->send_home_tickets < 0 && t->trial_way == c->tt.way) { // a ticket of the trial row under way has landed
            constexpr int TRIAL_TICKETS = 12;
            mcraw_ctx::TicketTrial &tt = c->tt;
            const auto now = std::chrono::steady_clock::now();
            if (tt.landed++ == 0)
                tt.t_first = now; // (the row's clock starts with its first landing; that ticket's bytes are not counted)
            else
                tt.bytes += t->trial_bytes;
            if (tt.landed == TRIAL_TICKETS) {
                tt.rate[tt.way] = tt.bytes / std::max(1e-9, std::chrono::duration<double>(now - tt.t_first).count());
                if (tt.way == 0) {
                    tt = mcraw_ctx::TicketTrial{1, 0, 0, 0, now, {tt.rate[0], 0.0}};
                } else {
                    c->send_home_tickets = tt.rate[1] > tt.rate[0] * 1.03 ? 1 : 0;
                    if (std::getenv("MCRAW_TRACE"))
                        std::fprintf(stderr, "[mcraw] host-memory pipeline (tickets): status words fetched %.1f GB/s, sent home %.1f GB/s: %s from here on\n",
                                     tt.rate[0] / 1e9, tt.rate[1] / 1e9, c->send_home_tickets ? "sent" : "fetched");
                }
            }
        }
        const size_t n = t->got_status.size();
        for (size_t i = 0; i < n; i++) {
            if (written)
                written[i] = t->got_written[i];
            if (status)
                status[i] = t->got_status[i];
        }
    }
    delete t;
    return rc;
}

int mcraw_ctx_synchronize(mcraw_ctx *c, int32_t *status, int nframes)
{
    if (!c)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    for (Slot &s : c->slots)
        if (s.busy) {
            if (s.owner) { // an asynchronous host-memory batch keeps its statuses: file them in its ticket
                if (int rc = drain_part(s.owner, s.owner_part))
                    return rc;
                continue;
            }
            HIP_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
    for (int k = 0; k < NDSLOT; k++) {
        Slot &s = c->dslots[k];
        if (!s.busy)
            continue;
        const bool last = k == c->last_slot;
        if (s.unresolved) { // submitted without a status request: frames that need a second plan get it now
            if (int rc = settle_slot(c, s, last ? &c->last_status : nullptr))
                return rc;
        } else {
            HIP_TRY(hipEventSynchronize(s.done));
            s.busy = false;
        }
    }
    if (status) {
        const int n = std::min(nframes, c->last_n);
        for (int i = 0; i < n; i++)
            status[i] = i < static_cast<int>(c->last_status.size()) ? c->last_status[i] : 0;
    }
    return 0;
}

uint64_t mcraw_ctx_last_serial(mcraw_ctx *c)
{
    if (!c)
        return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->serial;
}

int mcraw_ctx_batch_status(mcraw_ctx *c, uint64_t serial, int32_t *status, int nframes)
{
    if (!c || nframes < 0 || (nframes > 0 && !status))
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    for (int k = 0; k < NDSLOT; k++) { // still in its slot: wait for it and plan again what needs it
        Slot &s = c->dslots[k];
        if (s.busy && s.unresolved && s.serial == serial) {
            const bool last = k == c->last_slot;
            if (int rc = settle_slot(c, s, last ? &c->last_status : nullptr))
                return rc;
            if (last)
                c->last_slot = -1;
        }
    }
    for (const auto &e : c->settled)
        if (e.first == serial) {
            for (int i = 0; i < nframes; i++)
                status[i] = i < static_cast<int>(e.second.size()) ? e.second[i] : 0;
            return 0;
        }
    return 1; // not a batch submitted without a status request, or more than 64 such batches ago
}

int32_t mcraw_ctx_errors(mcraw_ctx *c, int reset)
{
    if (!c)
        return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    const int32_t v = c->sticky;
    if (reset)
        c->sticky = 0;
    return v;
}

size_t mcraw_decode7(uint16_t *output, int width, int height, const uint8_t *input, size_t len)
{
    return decode_one(MCRAW_TYPE_BLOCK, output, width, height, input, len);
}

size_t mcraw_decode6(uint16_t *output, int width, int height, const uint8_t *input, size_t len)
{
    return decode_one(MCRAW_TYPE_LEGACY, output, width, height, input, len);
}

int mcraw_ctx_set_post(mcraw_ctx *c, const mcraw_post *post)
{
    if (!c)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!post) {
        c->post = Post{0, 0, 0};
        return 0;
    }
    const uint32_t packs = post->flags & (MCRAW_POST_PACK12 | MCRAW_POST_PACK10 | MCRAW_POST_PACK14);
    if ((post->flags & ~(MCRAW_POST_BLACK | MCRAW_POST_PACK12 | MCRAW_POST_PACK10 | MCRAW_POST_PACK14)) != 0u ||
        (packs & (packs - 1u)) != 0u) { // at most one strip width
        g_err = "mcraw: unknown post-stage flags";
        return -1;
    }
    Post p{0, 0, 0};
    if (post->flags & MCRAW_POST_BLACK) {
        p.mode |= POST_BLACK;
        p.black01 = static_cast<uint32_t>(post->black[0]) | (static_cast<uint32_t>(post->black[1]) << 16);
        p.black23 = static_cast<uint32_t>(post->black[2]) | (static_cast<uint32_t>(post->black[3]) << 16);
    }
    if (post->flags & MCRAW_POST_PACK12)
        p.mode |= POST_PACK12;
    if (post->flags & MCRAW_POST_PACK10)
        p.mode |= POST_PACK10;
    if (post->flags & MCRAW_POST_PACK14)
        p.mode |= POST_PACK14;
    c->post = p;
    return 0;
}

void mcraw_legacy_launch_order(const uint32_t *nseg, int n, uint32_t *tab)
{
    if (!nseg || !tab || n <= 0)
        return;
    std::vector<uint32_t> order(n);
    for (int k = 0; k < n; k++)
        order[k] = static_cast<uint32_t>(k);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return nseg[a] > nseg[b]; });
    uint32_t lo = 0, base = 0;
    for (int t = 0; t < n; t++) {
        const uint32_t hi = nseg[order[n - 1 - t]]; // the smallest frame still in play leaves after this round
        tab[t] = base;
        tab[n + 1 + t] = lo;
        tab[2 * n + 1 + t] = order[t];
        base += static_cast<uint32_t>(n - t) * (hi - lo);
        lo = hi;
    }
    tab[n] = base; // = segments of all frames
}

int mcraw_ctx_profile(mcraw_ctx *c, int enable)
{
    if (!c)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profile = enable == 1 ? ~0u : static_cast<uint32_t>(enable) >> 1;
    return 0;
}

int mcraw_ctx_xcd_runs(mcraw_ctx *c)
{
    if (!c)
        return -2;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->tune_last >= 0 && c->tunes[c->tune_last].decided >= 0 ? static_cast<int>(TUNE_CHUNKS[c->tunes[c->tune_last].decided]) : -1;
}

int mcraw_ctx_side_parts(mcraw_ctx *c)
{
    if (!c)
        return -2;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->side_last < 0 || c->side_tunes[c->side_last].decided < 0)
        return -1;
    const mcraw_ctx::SideTune &t = c->side_tunes[c->side_last];
    return t.cand[t.decided][0] * 16 + t.cand[t.decided][1];
}

int mcraw_ctx_profile_every(mcraw_ctx *c, int n)
{
    if (!c || n < 1)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    c->profile_every = static_cast<uint32_t>(n);
    return 0;
}

int mcraw_ctx_kernel_ms(mcraw_ctx *c, int id, double *ms, int *launches, int reset)
{
    if (!c || id < 0 || id >= MCRAW_K_COUNT)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    KStat &k = c->kstat[id];
    for (auto &p : k.pending) {
        HIP_TRY(hipEventSynchronize(p.second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
        k.ms += t;
        k.launches++;
        c->event_pool.push_back(p.first);
        c->event_pool.push_back(p.second);
    }
    k.pending.clear();
    if (ms)
        *ms = k.ms;
    if (launches)
        *launches = k.launches;
    if (reset) {
        k.ms = 0.0;
        k.launches = 0;
    }
    return 0;
}

void *mcraw_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return nullptr;
    return p;
}

void mcraw_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

} // extern "C"
