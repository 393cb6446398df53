// mcraw_submit.hip -- one batch: plan on the host (geometry, workspace carving, launch order), table upload, kernel launches
// (host side of the C ABI, see mcraw_host.h).
#include "mcraw_host.h"

using namespace mcraw;

namespace mcraw {

int submit(mcraw_ctx *c, Slot &s, const mcraw_frame *frames, int n, const std::vector<Geom7> *geom_override,
           const uint8_t *const *dev_in, uint16_t *const *dev_out, hipStream_t st, size_t *status_off)
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
        if (longstreams && !dev_in && !geom_override && !c->env_side_split[0]) {
            side_cand = side_pick(c, n7, rmax);
            const mcraw_ctx::SideTune &t = c->side_tunes[c->side_last];
            const int k = side_cand >= 0 ? side_cand : std::max(t.decided, 0);
            nsplit[0] = t.cand[k][0];
            nsplit[1] = t.cand[k][1];
        } else if (longstreams && n7 * 8 <= 512)
            nsplit[0] = nsplit[1] = 4;
        else if (longstreams && n7 * 4 <= 512)
            nsplit[0] = nsplit[1] = 2;
        if (c->env_side_split[0])
            nsplit[0] = c->env_side_split[0], nsplit[1] = c->env_side_split[1];
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
            HIP_TRY(hipMemsetAsync(s.side_sync.p, 0, s.side_sync.cap, st));
        }
        if (++s.look_epoch == 0u) { // (2^32 batches later: start over)
            if (s.look.p)
                HIP_TRY(hipMemsetAsync(s.look.p, 0, s.look.cap, st));
            if (s.side_sync.p)
                HIP_TRY(hipMemsetAsync(s.side_sync.p, 0, s.side_sync.cap, st));
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
    if (n6)
        HIP_TRY(hipMemcpyAsync(dev, img, L.upload_bytes, hipMemcpyHostToDevice, st));

    // ---- launches -----------------------------------------------------------
    // A batch that holds both encodings (BASELINE config 4): the two codecs share nothing, and k7_side is a handful of
    // latency-bound workgroups -- the legacy kernel runs beside the type-7 kernels on the context's second stream, forked
    // behind the table upload and joined in front of whatever the caller queues next.
    const bool both = n7 > 0 && n6 > 0 && s.fork && s.join && c->legacy;
    hipStream_t st6 = st;
    if (both) {
        st6 = c->legacy;
        HIP_TRY(hipEventRecord(s.fork, st));
        HIP_TRY(hipStreamWaitEvent(st6, s.fork, 0));
    }
    if (n7) {
        Work7 W{};
        W.plans = reinterpret_cast<const Plan7 *>(img + L.plans7); // pinned host memory, device-visible at the same address
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
        W.side_lastc = c->env_side_lastc >= 0 ? static_cast<uint32_t>(c->env_side_lastc) : (static_cast<long>(n7) * wpf <= 256 ? 1u : 0u);
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
        const int xcd_env = c->env_xcd_chunk;
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
            hipStream_t kst = st;
            if ((stage == MCRAW_K7_TILES && tune_cand >= 0) || time_side) {
                ta = get_event(c);
                tb = get_event(c);
                if (ta && tb)
                    (void)hipEventRecord(ta, kst);
            }
            {
                KTimer t(c, static_cast<int>(stage), kst);
                launch_k7(W, stage, kst);
            }
            if (ta && tb) {
                (void)hipEventRecord(tb, kst);
                if (time_side)
                    c->side_tunes[c->side_last].pending.push_back({ta, tb, side_cand});
                else
                    c->tunes[c->tune_last].pending.push_back({ta, tb, tune_cand});
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


} // namespace mcraw
