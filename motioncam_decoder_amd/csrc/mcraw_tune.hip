// mcraw_tune.hip -- the context's run-time measurements: which XCD mapping k7_tiles runs with, how many parts resolve a long side stream
// (host side of the C ABI, see mcraw_host.h).
#include "mcraw_host.h"

using namespace mcraw;

namespace mcraw {


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


} // namespace mcraw
