// mcraw_abi.hip -- the extern "C" entry points of include/mcraw_hip.h and the life cycle of a context: streams, a ring of batch
// slots (pinned upload buffer + HBM arena + completion event).  The work behind them: mcraw_host.h.
#include "mcraw_host.h"

using namespace mcraw;

std::atomic<int> g_ctx_on_device[64];

namespace mcraw {

thread_local std::string g_err;

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


} // namespace mcraw

namespace {

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
    { // what the environment says (mcraw_ctx: env_*)
        if (const char *e = std::getenv("MCRAW_SIDE_SPLIT")) {
            int b = 0, r = 0;
            const int got = std::sscanf(e, "%d,%d", &b, &r);
            if (got == 1)
                r = b;
            if (got >= 1 && b >= 1 && b <= MAX_SPLIT7 && r >= 1 && r <= MAX_SPLIT7)
                c->env_side_split[0] = b, c->env_side_split[1] = r;
        }
        if (const char *e = std::getenv("MCRAW_SIDE_LASTC"))
            c->env_side_lastc = std::atoi(e);
        if (const char *e = std::getenv("MCRAW_XCD_CHUNK"))
            c->env_xcd_chunk = std::atoi(e);
        if (const char *e = std::getenv("MCRAW_SHORT_WAY"))
            c->env_short_way = std::atoi(e) != 0 ? 1 : 0;
        c->env_trace = std::getenv("MCRAW_TRACE") != nullptr;
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
    gate_forget(c);
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
        if (t->trial_way >= 0 && c->tt.queued > 0)
            c->tt.queued--; // (a ticket that never flew lands nowhere: its place in the trial row is free again)
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
                    if (c->env_trace)
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

int mcraw_ctx_host_way(mcraw_ctx *c)
{
    if (!c)
        return -1;
    std::lock_guard<std::mutex> lk(c->mu);
    return c->send_home;
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
