// mcraw_device.hip -- batches whose buffers are in HBM: slots, statuses, second plans (host side of the C ABI, see mcraw_host.h).
#include "mcraw_host.h"

using namespace mcraw;

namespace mcraw {

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

int acquire_slot(mcraw_ctx *c, Slot **out, bool device_batch)
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

void warm_send_status(hipStream_t st) // (the kernel's first launch -- its code object's load -- in front of the first timed one)
{
    hipLaunchKernelGGL(k_words_home, dim3(1), dim3(64), 0, st, nullptr, nullptr, 0u);
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
int32_t public_status(int32_t st)
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
    Slot *sp = nullptr;
    if (int rc = acquire_slot(c, &sp, true))
        return rc;
    Slot &s = *sp;
    hipStream_t st = user ? user : c->stream;
    if (int rc = submit(c, s, frames, n, nullptr, nullptr, nullptr, st, &s.status_off))
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


} // namespace mcraw
