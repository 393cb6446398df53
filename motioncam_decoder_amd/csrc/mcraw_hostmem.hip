// mcraw_hostmem.hip -- batches whose buffers are in host memory: the three-lane pipeline and its tickets
// (host side of the C ABI, see mcraw_host.h).
#include "mcraw_host.h"

using namespace mcraw;

namespace mcraw {

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

DevGate g_gate[64];

void gate_forget(const mcraw_ctx *c)
{
    if (c->device < 0 || c->device >= 64)
        return;
    DevGate &g = g_gate[c->device];
    std::lock_guard<std::mutex> lk(g.mu);
    g.flights.erase(std::remove_if(g.flights.begin(), g.flights.end(), [&](const DevGate::Flight &f) { return f.c == c; }), g.flights.end());
}

// Contexts that share their device: wait until at most one short batch of ANY of them still has downloads under way.  Called with
// the gate locked; the batches that are over are dropped from the list on the way.
static void gate_wait(DevGate &g)
{
    for (;;) {
        g.flights.erase(std::remove_if(g.flights.begin(), g.flights.end(),
                                       [](const DevGate::Flight &f) { return hipEventQuery(f.done) != hipErrorNotReady; }),
                        g.flights.end());
        (void)hipGetLastError();
        if (g.flights.size() <= 1)
            return;
        (void)hipEventSynchronize(g.flights.front().done); // (the oldest one; its slot's event may have been recorded again since: then longer)
    }
}

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
        t->small = parts <= SHORT_PARTS;
        t->send = t->small && t->want_send == 1;
    }
    // (a device shared by several contexts: one count of the batches under way for all of them, held while this one is queued)
    const bool shared = t->small && !alone_on_device(c) && c->device >= 0 && c->device < 64;
    std::unique_lock<std::mutex> gate_lk;
    if (shared) {
        gate_lk = std::unique_lock<std::mutex>(g_gate[c->device].mu);
        gate_wait(g_gate[c->device]);
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
    if (shared && !t->parts.empty())
        g_gate[c->device].flights.push_back({c->slots[t->parts.back().slot].done, c});
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

size_t host_bytes(const mcraw_frame *frames, int n)
{
    size_t total = 0;
    for (int i = 0; i < n; i++)
        total += static_cast<size_t>(frames[i].len) + (frames[i].width > 0 && frames[i].height > 0 ? static_cast<size_t>(frames[i].width) * frames[i].height * 2 : 0);
    return total;
}

void way_from_env(mcraw_ctx *c)
{
    if (c->env_short_way >= 0 && c->send_home < 0)
        c->send_home = c->send_home_tickets = c->env_short_way;
    // (a context that shares its device does not compare -- the others' traffic is in its times --: it takes what a context of
    // this device found, if one has)
    if (c->send_home < 0 && !alone_on_device(c) && c->device >= 0 && c->device < 64) {
        std::lock_guard<std::mutex> lk(g_gate[c->device].mu);
        if (g_gate[c->device].way >= 0)
            c->send_home = c->send_home_tickets = g_gate[c->device].way;
    }
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
            warm_send_status(c->slots[0].stream);
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
        if (c->device >= 0 && c->device < 64) {
            std::lock_guard<std::mutex> lk(g_gate[c->device].mu);
            g_gate[c->device].way = c->send_home;
        }
        if (c->env_trace)
            std::fprintf(stderr, "[mcraw] host-memory pipeline: status words fetched %.1f GB/s, sent home %.1f GB/s: %s from here on\n",
                         c->trial_rate[0] / 1e9, c->trial_rate[1] / 1e9, c->send_home ? "sent" : "fetched");
    }
}

int deal_host(mcraw_ctx *c, const mcraw_frame *frames, int n, size_t *written, int32_t *status_out,
              std::vector<std::unique_ptr<mcraw_ticket>> &pieces, std::vector<int> &piece_first, bool finish, int way)
{
    // (a piece is cut by bytes here and into sub-batches of up to 96 MB by host_submit, which ends one in front of the frame that
    // would not fit: four sub-batches' worth of bytes are five or six sub-batches, short by host_submit's count)
    const size_t piece = PIECE_BYTES;
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


} // namespace mcraw
