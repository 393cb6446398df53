// mcraw_host.h -- what the host-side units of the C ABI (include/mcraw_hip.h) share: the context, its batch slots, the tickets of
// the host-memory pipeline, and the functions the units call in each other.
//   mcraw_abi.hip      the extern "C" entry points, context life cycle
//   mcraw_submit.hip   one batch: plan (geometry, workspace carving, launch order), table upload, kernel launches
//   mcraw_tune.hip     the context's run-time measurements (XCD mapping of k7_tiles, parts per side stream)
//   mcraw_device.hip   batches whose buffers are in HBM: slots, statuses, second plans
//   mcraw_hostmem.hip  batches whose buffers are in host memory: the three-lane pipeline, tickets
// Replaces the per-frame dispatch of lib/Decoder.cpp:216-234 with batched submits.  There is no CPU decode fallback in any of them.
#pragma once
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
void launch_k7(const Work7 &W, uint32_t stage, hipStream_t st);
void launch_k6_decode(const Plan6 *plans, const uint32_t *wg_tab, uint32_t stage0, uint32_t nwg, const Look6 &look,
                      uint32_t *tickets, uint32_t epoch, int nframes, uint32_t smax, const Post &post, hipStream_t st);
} // namespace mcraw

struct mcraw_ticket;
struct mcraw_ctx;

namespace mcraw {


extern thread_local std::string g_err;

inline int fail(hipError_t e, const char *what)
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


} // namespace mcraw

using namespace mcraw; // (an internal header: the two C ABI handle types below are global and made of the namespace's types)

// An asynchronous host-memory batch (mcraw_decode_batch_async / mcraw_ticket_wait).
struct mcraw_ticket {
    mcraw_ctx *c = nullptr;
    std::vector<mcraw_frame> frames;
    std::vector<int32_t> status;
    std::vector<uint32_t> encH; // coded heights (type-7 frames)
    std::vector<mcraw::Part> parts;
    std::vector<int> skipped; // frames that no sub-batch holds (no device memory for their workspace): failed on their own
    mcraw::Post post{0, 0, 0}; // post stage the batch was submitted with
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
extern std::atomic<int> g_ctx_on_device[64];

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
    // What the environment says, read ONCE when the context is made (tests and tools set these for a child process):
    int env_side_split[2] = {0, 0}; // MCRAW_SIDE_SPLIT "b,r": parts per bits / refs stream for every type-7 batch (0: measured / none)
    int env_side_lastc = -1;        // MCRAW_SIDE_LASTC: the last part of a stream counts too (0 / 1; -1: by the batch's size)
    int env_xcd_chunk = -1;         // MCRAW_XCD_CHUNK: k7_tiles' XCD mapping pinned (-1: measured)
    int env_short_way = -1;         // MCRAW_SHORT_WAY: how the status words of host-memory batches come home (-1: measured)
    bool env_trace = false;         // MCRAW_TRACE: the measurements' verdicts on stderr
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

namespace mcraw {


int ensure(Buf &b, size_t bytes, bool pinned);
hipEvent_t get_event(mcraw_ctx *c);

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

// ---- mcraw_tune.hip
int tune_pick(mcraw_ctx *c, int n7, uint32_t R, uint32_t mode);
int side_pick(mcraw_ctx *c, int n7, uint32_t R);
// ---- mcraw_submit.hip
int submit(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, const std::vector<Geom7> *geom_override,
           const uint8_t *const *dev_in, uint16_t *const *dev_out, hipStream_t st, size_t *status_off);
// ---- mcraw_device.hip
int fetch_status(mcraw_ctx *c, Slot &s, size_t status_off, int n, hipStream_t st, int32_t *status, uint32_t *encH = nullptr, bool sent = false);
int send_status(Slot &s, size_t status_off, hipStream_t st);
void warm_send_status(hipStream_t st);
int drain_part(mcraw_ticket *t, int idx);
int resolve_device(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, hipStream_t st, int32_t *status, uint32_t *encH);
int settle_slot(mcraw_ctx *c, Slot &s, std::vector<int32_t> *keep);
int acquire_slot(mcraw_ctx *c, Slot **out, bool device_batch = false);
size_t written_of(const mcraw_frame &f, int32_t status, uint32_t encH);
int32_t public_status(int32_t st);
int decode_device(mcraw_ctx *c, const mcraw_frame *frames, int n, hipStream_t user, size_t *written, int32_t *status_out);
// ---- mcraw_hostmem.hip
// (is this context the only one of the process on its device?  The pipeline's scheduling is tuned for ONE stream of batches on a
// GPU's copy engines; contexts that share a device share one count of the batches under way, DevGate)
inline bool alone_on_device(const mcraw_ctx *c) { return !c->counted || g_ctx_on_device[c->device].load() <= 1; }
// What the contexts of one device share (round 6; two contexts on one GPU -- a pool of two members on it -- used to fall back to
// the pipeline's old scheduling, 2 277 against 2 940 UHD frames/s): the short batches of ALL of them that still have downloads
// under way, as the `done` events of their last sub-batches (a batch is queued only when at most one other, of whatever context,
// is still out), and which way home of the status words a context of this device measured as the faster one.
struct DevGate {
    struct Flight {
        hipEvent_t done;
        const mcraw_ctx *c;
    };
    std::mutex mu;
    std::vector<Flight> flights;
    int way = -1;
};
extern DevGate g_gate[64];
void gate_forget(const mcraw_ctx *c); // (a context that goes away: none of its events may stay in the gate)
constexpr size_t PIECE_BYTES = 4 * (96ull << 20); // a large host-memory batch is dealt out in pieces of this size (deal_host)
int host_submit_part(mcraw_ticket *t, int first, int count);
int host_submit(mcraw_ticket *t);
int host_finish(mcraw_ticket *t, size_t *written, int32_t *status_out);
void forget_ticket(mcraw_ticket *t);
int land_piece(std::vector<std::unique_ptr<mcraw_ticket>> &pieces, std::vector<int> &piece_first, size_t *written, int32_t *status_out);
size_t host_bytes(const mcraw_frame *frames, int n);
void way_from_env(mcraw_ctx *c);
int big_way(mcraw_ctx *c, size_t total, bool *trial);
void big_way_result(mcraw_ctx *c, int way, size_t total, double seconds);
int deal_host(mcraw_ctx *c, const mcraw_frame *frames, int n, size_t *written, int32_t *status_out,
              std::vector<std::unique_ptr<mcraw_ticket>> &pieces, std::vector<int> &piece_first, bool finish, int way);
int decode_host(mcraw_ctx *c, const mcraw_frame *frames, int n, size_t *written, int32_t *status_out);

} // namespace mcraw
