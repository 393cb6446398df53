// mcraw_pool.hip -- several GPUs of one node behind one handle (include/mcraw_hip.h, "device pool").
//
// The reference decodes a clip one frame after the other on one thread (example.cpp:187-195 ->
// lib/Decoder.cpp:184-235).  Frames are independent, so a batch shards by frame index: frame i
// goes to pool member i mod G -- no exchange between devices, nothing collective.
// Every member is a mcraw_ctx of its own (streams, staging, workspace) driven by ONE HOST THREAD
// of its own, which is bound to the CPUs of the GPU's NUMA node before it touches the device, so
// that the plans it writes and the pinned memory it allocates are local to that GPU's PCIe root.
// Results do not depend on the pool size.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mcraw_hip.h"

namespace {

// CPUs of the NUMA node a PCI device hangs off (sysfs), empty when the host does not say.
std::vector<int> cpus_near_pci(const char *bus_id)
{
    std::vector<int> cpus;
    char path[256];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus_id);
    int node = -1;
    if (FILE *f = std::fopen(path, "r")) {
        if (std::fscanf(f, "%d", &node) != 1)
            node = -1;
        std::fclose(f);
    }
    if (node < 0)
        return cpus;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = std::fopen(path, "r");
    if (!f)
        return cpus;
    char buf[4096];
    if (std::fgets(buf, sizeof(buf), f)) {
        for (char *tok = std::strtok(buf, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            if (std::sscanf(tok, "%d-%d", &a, &b) == 2) {
                for (int c = a; c <= b; c++)
                    cpus.push_back(c);
            } else if (std::sscanf(tok, "%d", &a) == 1) {
                cpus.push_back(a);
            }
        }
    }
    std::fclose(f);
    return cpus;
}

// One pool member: a device, its context, and the host thread that drives it.  Tasks are queued and run in order; run()
// returns the task's number and wait() blocks until that task has run, so several host threads may hand a member work
// at the same time (each waits for its own task).
struct Member {
    int device = 0;
    int numa_cpus = 0; // CPUs the thread was bound to (0: not bound)
    mcraw_ctx *ctx = nullptr;
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> tasks;
    uint64_t submitted = 0, completed = 0;
    // the same two counters for waiters that spin a little before they block: a hand-off through the condition variable
    // alone costs a wake-up of 30 - 60 us each way, a quarter of a 240-frame batch's decode time
    std::atomic<uint64_t> submitted_a{0}, completed_a{0};
    bool created = false, quit = false;
    int create_rc = 0;
    std::string create_err;

    uint64_t run(std::function<void()> fn)
    {
        std::unique_lock<std::mutex> lk(mu);
        tasks.push_back(std::move(fn));
        const uint64_t seq = ++submitted;
        submitted_a.store(seq, std::memory_order_release);
        cv.notify_all();
        return seq;
    }
    static constexpr int SPIN_US = 200; // how long a waiter polls before it blocks
    void wait(uint64_t seq)
    {
        const auto t0 = std::chrono::steady_clock::now();
        while (completed_a.load(std::memory_order_acquire) < seq) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) { // (a long task: block; a 240-frame batch is 1 ms)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return completed >= seq; });
                return;
            }
            std::this_thread::yield();
        }
    }
    void wait_created()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return created; });
    }
    void loop()
    {
        // before the first HIP call of this thread: next to the GPU -- on those CPUs of the GPU's NUMA node that the process
        // is allowed to run on (a launcher's taskset / cpuset stays in force; no common CPU: the thread stays where it is)
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof(bus), device) == hipSuccess) {
            for (char *p = bus; *p; p++)
                if (*p >= 'A' && *p <= 'F')
                    *p = static_cast<char>(*p - 'A' + 'a');
            const std::vector<int> cpus = cpus_near_pci(bus);
            cpu_set_t allowed, set;
            CPU_ZERO(&allowed);
            CPU_ZERO(&set);
            int n = 0;
            if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
                for (int c : cpus)
                    if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
                        CPU_SET(c, &set);
                        n++;
                    }
                if (n > 0 && sched_setaffinity(0, sizeof(set), &set) == 0)
                    numa_cpus = n;
            }
        }
        create_rc = mcraw_ctx_create(device, &ctx);
        if (create_rc != 0)
            create_err = mcraw_last_error();
        {
            std::unique_lock<std::mutex> lk(mu);
            created = true;
            cv.notify_all();
        }
        uint64_t taken = 0;
        for (;;) {
            std::function<void()> fn;
            { // (a member that has just finished a task polls for the next one for a moment before it goes to sleep)
                const auto t0 = std::chrono::steady_clock::now();
                while (submitted_a.load(std::memory_order_acquire) == taken &&
                       std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(SPIN_US))
                    std::this_thread::yield();
            }
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !tasks.empty() || quit; });
                if (tasks.empty())
                    break; // (quit, and nothing left to run)
                fn = std::move(tasks.front());
                tasks.pop_front();
                taken++;
            }
            fn();
            {
                std::unique_lock<std::mutex> lk(mu);
                completed++;
                completed_a.store(completed, std::memory_order_release);
                cv.notify_all();
            }
        }
        if (ctx)
            mcraw_ctx_destroy(ctx);
        ctx = nullptr;
    }
};

thread_local std::string g_pool_err;

} // namespace

struct mcraw_pool {
    std::vector<Member *> members;
    std::mutex mu; // one batch at a time
    uint64_t id = 0; // (a later pool at the same address is another pool)
};

namespace {
std::atomic<uint64_t> g_pool_ids{0};
// The calling thread's last resident batch that was only queued (mcraw_pool_decode_batch_device without `written` / `status`):
// which of its frames went to which member, under which serial number of that member's context, and what the host decided
// about a frame by itself (MCRAW_E_ARGS).  One record per pool and host thread: mcraw_pool_synchronize reports a thread its OWN batch.
struct Queued {
    uint64_t pool_id = 0;
    int n = 0;
    std::vector<std::vector<int>> index; // per member: caller's indices of its frames
    std::vector<uint64_t> serial;        // per member: its context's batch serial (0: nothing was submitted)
    std::vector<int32_t> host_status;    // per frame
};
thread_local std::unordered_map<const mcraw_pool *, Queued> g_queued;
} // namespace

// A sharded batch in flight.
struct mcraw_pool_ticket {
    mcraw_pool *pool = nullptr;
    int n = 0;
    std::vector<std::vector<mcraw_frame>> sub; // per member
    std::vector<std::vector<int>> index;       // per member: position of its frames in the caller's batch
    std::vector<mcraw_ticket *> tickets;
    std::vector<int> rc;
    std::vector<std::string> err;
};

extern "C" {

int mcraw_shard_of(long index, int ndevices)
{
    if (ndevices <= 0 || index < 0)
        return -1;
    return static_cast<int>(index % ndevices);
}

int mcraw_shard_count(long nframes, int member, int ndevices)
{
    if (ndevices <= 0 || member < 0 || member >= ndevices || nframes < 0)
        return -1;
    return static_cast<int>((nframes - member + ndevices - 1) / ndevices);
}

const char *mcraw_pool_last_error(void) { return g_pool_err.c_str(); }

int mcraw_pool_create(const int *devices, int ndevices, mcraw_pool **out)
{
    if (!out || ndevices < 0 || (ndevices > 0 && !devices)) {
        g_pool_err = "mcraw_pool_create: bad arguments";
        return -1;
    }
    *out = nullptr;
    int have = 0;
    hipError_t e = hipGetDeviceCount(&have);
    if (e != hipSuccess || have <= 0) {
        g_pool_err = "mcraw: no HIP device available (the decode path has no CPU fallback)";
        return e != hipSuccess ? -static_cast<int>(e) : -static_cast<int>(hipErrorNoDevice);
    }
    std::vector<int> devs;
    if (ndevices > 0) {
        devs.assign(devices, devices + ndevices);
    } else {
        const char *env = std::getenv("MCRAW_DEVICES"); // "all" or a comma separated list
        if (env && std::strcmp(env, "all") == 0) {
            for (int d = 0; d < have; d++)
                devs.push_back(d);
        } else if (env && *env) {
            std::string s(env);
            size_t pos = 0;
            while (pos < s.size()) {
                size_t end = s.find(',', pos);
                if (end == std::string::npos)
                    end = s.size();
                if (end > pos)
                    devs.push_back(std::atoi(s.substr(pos, end - pos).c_str()));
                pos = end + 1;
            }
        }
        if (devs.empty()) { // one member: MCRAW_DEVICE, else the current device (like mcraw_ctx_create(-1))
            const char *one = std::getenv("MCRAW_DEVICE");
            int d = 0;
            if (one && *one)
                d = std::atoi(one);
            else if (hipGetDevice(&d) != hipSuccess)
                d = 0;
            devs.push_back(d);
        }
    }
    for (int d : devs)
        if (d < 0 || d >= have) {
            g_pool_err = "mcraw_pool_create: device index out of range";
            return -static_cast<int>(hipErrorInvalidDevice);
        }
    mcraw_pool *p = new mcraw_pool();
    p->id = ++g_pool_ids;
    for (int d : devs) {
        Member *m = new Member();
        m->device = d;
        m->thread = std::thread([m] { m->loop(); });
        p->members.push_back(m);
    }
    int rc = 0;
    for (Member *m : p->members) {
        m->wait_created();
        if (m->create_rc != 0 && rc == 0) {
            rc = m->create_rc;
            g_pool_err = m->create_err;
        }
    }
    if (rc != 0) {
        mcraw_pool_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

void mcraw_pool_destroy(mcraw_pool *p)
{
    if (!p)
        return;
    g_queued.erase(p); // (the calling thread's record; other threads' records are told apart by the pool's id)
    for (Member *m : p->members) {
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->quit = true;
            m->cv.notify_all();
        }
        if (m->thread.joinable())
            m->thread.join();
        delete m;
    }
    delete p;
}

int mcraw_pool_size(const mcraw_pool *p) { return p ? static_cast<int>(p->members.size()) : 0; }

int mcraw_pool_device(const mcraw_pool *p, int member)
{
    if (!p || member < 0 || member >= static_cast<int>(p->members.size()))
        return -1;
    return p->members[member]->device;
}

int mcraw_pool_numa_cpus(const mcraw_pool *p, int member)
{
    if (!p || member < 0 || member >= static_cast<int>(p->members.size()))
        return -1;
    return p->members[member]->numa_cpus;
}

mcraw_ctx *mcraw_pool_ctx(mcraw_pool *p, int member)
{
    if (!p || member < 0 || member >= static_cast<int>(p->members.size()))
        return nullptr;
    return p->members[member]->ctx;
}

int mcraw_pool_set_post(mcraw_pool *p, const mcraw_post *post)
{
    if (!p)
        return -1;
    std::lock_guard<std::mutex> lk(p->mu); // not while a batch is being dealt: all members of one batch get the same stage
    int rc = 0;
    for (Member *m : p->members)
        if (int r = mcraw_ctx_set_post(m->ctx, post))
            rc = r;
    return rc;
}

void *mcraw_pool_host_alloc(mcraw_pool *p, int member, size_t bytes)
{
    if (!p || member < 0 || member >= static_cast<int>(p->members.size()))
        return nullptr;
    Member *m = p->members[member];
    void *res = nullptr;
    m->wait(m->run([&] { res = mcraw_host_alloc(bytes); })); // pages are taken by the member's (NUMA-bound) thread
    return res;
}

int mcraw_pool_decode_batch_async(mcraw_pool *p, const mcraw_frame *frames, int nframes, mcraw_pool_ticket **out)
{
    if (out)
        *out = nullptr;
    if (!p || !out || nframes < 0 || (nframes > 0 && !frames)) {
        g_pool_err = "mcraw_pool_decode_batch_async: bad arguments";
        return -1;
    }
    const int G = static_cast<int>(p->members.size());
    mcraw_pool_ticket *t = new mcraw_pool_ticket();
    t->pool = p;
    t->n = nframes;
    t->sub.resize(G);
    t->index.resize(G);
    t->tickets.assign(G, nullptr);
    t->rc.assign(G, 0);
    t->err.resize(G);
    for (int i = 0; i < nframes; i++) { // frame i -> member i mod G
        const int m = mcraw_shard_of(i, G);
        t->sub[m].push_back(frames[i]);
        t->index[m].push_back(i);
    }
    std::lock_guard<std::mutex> lk(p->mu);
    std::vector<uint64_t> seq(G, 0);
    for (int m = 0; m < G; m++) {
        Member *mem = p->members[m];
        seq[m] = mem->run([t, m, mem] {
            if (t->sub[m].empty())
                return;
            t->rc[m] = mcraw_decode_batch_async(mem->ctx, t->sub[m].data(), static_cast<int>(t->sub[m].size()), &t->tickets[m]);
            if (t->rc[m] != 0)
                t->err[m] = mcraw_last_error();
        });
    }
    int rc = 0;
    for (int m = 0; m < G; m++) {
        p->members[m]->wait(seq[m]);
        if (t->rc[m] != 0 && rc == 0) {
            rc = t->rc[m];
            g_pool_err = t->err[m];
        }
    }
    if (rc != 0) { // what was queued on the other members still runs into the caller's buffers: wait for it
        for (int m = 0; m < G; m++)
            if (t->tickets[m])
                (void)mcraw_ticket_wait(t->tickets[m], nullptr, nullptr);
        delete t;
        return rc;
    }
    *out = t;
    return 0;
}

int mcraw_pool_ticket_wait(mcraw_pool_ticket *t, size_t *written, int32_t *status)
{
    if (!t)
        return -1;
    mcraw_pool *p = t->pool;
    const int G = static_cast<int>(p->members.size());
    std::vector<std::vector<size_t>> wr(G);
    std::vector<std::vector<int32_t>> st(G);
    {
        // (no pool lock: a wait must not keep another host thread from dealing its batch; the members' queues keep order)
        std::vector<uint64_t> seq(G, 0);
        for (int m = 0; m < G; m++) {
            wr[m].assign(t->sub[m].size(), 0);
            st[m].assign(t->sub[m].size(), 0);
            Member *mem = p->members[m];
            seq[m] = mem->run([t, m, &wr, &st] {
                if (!t->tickets[m])
                    return;
                t->rc[m] = mcraw_ticket_wait(t->tickets[m], wr[m].data(), st[m].data());
                t->tickets[m] = nullptr;
                if (t->rc[m] != 0)
                    t->err[m] = mcraw_last_error();
            });
        }
        for (int m = 0; m < G; m++)
            p->members[m]->wait(seq[m]);
    }
    int rc = 0;
    for (int m = 0; m < G; m++) {
        if (t->rc[m] != 0 && rc == 0) {
            rc = t->rc[m];
            g_pool_err = t->err[m];
        }
        for (size_t k = 0; k < t->index[m].size(); k++) {
            if (written)
                written[t->index[m][k]] = wr[m][k];
            if (status)
                status[t->index[m][k]] = st[m][k];
        }
    }
    delete t;
    return rc;
}

// Is `ptr` device memory of `device`?  hipPointerGetAttributes answers that (a look-up in the runtime's tables: ~0.3 us, 0.15 ms for
// the 480 pointers of a 240-frame batch -- a seventh of the batch's decode time), so what it said about an ALLOCATION is
// remembered per calling thread FOR THE BATCH AT HAND: a pointer inside a range that was already asked about in this batch
// costs a binary search, every allocation is asked about once per batch (two look-ups for the usual batch whose frames are
// slices of one input and one output allocation) -- memory that was freed and allocated again elsewhere between two batches
// is never taken for what it was.
namespace {
struct KnownRange {
    uintptr_t base, end;
    int device; // -1: not device memory
    unsigned long long asked; // the batch (of this thread) in which the runtime was last asked about it
};
thread_local std::vector<KnownRange> g_ranges; // sorted by base
thread_local unsigned long long g_batch_no = 0; // resident batches this thread has submitted
} // namespace

static bool resident_on(const void *ptr, int device)
{
    if (!ptr)
        return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    size_t lo = 0, hi = g_ranges.size();
    while (lo < hi) { // last range with base <= a
        const size_t mid = (lo + hi) / 2;
        if (g_ranges[mid].base <= a)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo > 0 && a < g_ranges[lo - 1].end && g_ranges[lo - 1].asked == g_batch_no)
        return g_ranges[lo - 1].device == device;
    hipPointerAttribute_t at;
    int dev = -1;
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess)
        (void)hipGetLastError(); // (an unregistered host pointer is an error to HIP, not to us)
    else if (at.type == hipMemoryTypeDevice)
        dev = at.device;
    void *base = nullptr;
    size_t size = 0;
    if (dev >= 0 && hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&base), &size, const_cast<void *>(ptr)) == hipSuccess && size) {
        const uintptr_t b = reinterpret_cast<uintptr_t>(base);
        if (lo > 0 && g_ranges[lo - 1].base == b) {
            g_ranges[lo - 1] = {b, b + size, dev, g_batch_no};
        } else {
            if (g_ranges.size() >= 4096) // (a caller with very many allocations: start over)
                g_ranges.clear(), lo = 0;
            g_ranges.insert(g_ranges.begin() + static_cast<long>(lo), KnownRange{b, b + size, dev, g_batch_no});
            // a new allocation may overlap stale neighbours (their memory was freed): drop them
            for (size_t k = g_ranges.size(); k-- > 0;)
                if (k != lo && g_ranges[k].base < b + size && g_ranges[k].end > b)
                    g_ranges.erase(g_ranges.begin() + static_cast<long>(k)), lo -= k < lo ? 1 : 0;
        }
    } else {
        (void)hipGetLastError();
        if (lo > 0 && a < g_ranges[lo - 1].end) // (what was known about this address is no longer true)
            g_ranges.erase(g_ranges.begin() + static_cast<long>(lo - 1));
    }
    return dev == device;
}

int mcraw_pool_decode_batch_device(mcraw_pool *p, const mcraw_frame *frames, int nframes, size_t *written, int32_t *status)
{
    if (!p || nframes < 0 || (nframes > 0 && !frames)) {
        g_pool_err = "mcraw_pool_decode_batch_device: bad arguments";
        return -1;
    }
    const int G = static_cast<int>(p->members.size());
    std::vector<std::vector<mcraw_frame>> sub(G);
    std::vector<std::vector<int>> index(G);
    std::vector<int32_t> host_status(nframes, 0);
    g_batch_no++; // (what was learnt about an allocation in an earlier batch is asked again)
    for (int i = 0; i < nframes; i++) { // frame i -> member i mod G: its buffers live in THAT member's HBM
        const int m = mcraw_shard_of(i, G);
        // (checked, not assumed: a frame on another GPU would be decoded over xGMI at a fraction of the rate or fault, a
        // host pointer would fault; neither shows on a one-GPU box)
        if (!resident_on(frames[i].in, p->members[m]->device) || !resident_on(frames[i].out, p->members[m]->device)) {
            host_status[i] = MCRAW_E_ARGS;
            continue;
        }
        sub[m].push_back(frames[i]);
        index[m].push_back(i);
    }
    std::vector<std::vector<size_t>> wr(G);
    std::vector<std::vector<int32_t>> st(G);
    std::vector<int> rcs(G, 0);
    std::vector<std::string> errs(G);
    std::vector<uint64_t> seq(G, 0), serial(G, 0);
    const bool async = !written && !status; // nobody asks: the members only queue their shares (mcraw_pool_synchronize waits)
    {
        std::lock_guard<std::mutex> lk(p->mu);
        for (int m = 0; m < G; m++) {
            wr[m].assign(sub[m].size(), 0);
            st[m].assign(sub[m].size(), 0);
            Member *mem = p->members[m];
            seq[m] = mem->run([&, m, mem] {
                if (sub[m].empty())
                    return;
                rcs[m] = mcraw_decode_batch(mem->ctx, sub[m].data(), static_cast<int>(sub[m].size()), MCRAW_MEM_DEVICE, nullptr,
                                            async ? nullptr : wr[m].data(), async ? nullptr : st[m].data());
                if (rcs[m] != 0)
                    errs[m] = mcraw_last_error();
                else
                    serial[m] = mcraw_ctx_last_serial(mem->ctx); // (this member's context is driven by this thread alone)
            });
        }
    }
    int rc = 0;
    for (int m = 0; m < G; m++) {
        p->members[m]->wait(seq[m]);
        if (rcs[m] != 0 && rc == 0) {
            rc = rcs[m];
            g_pool_err = errs[m];
        }
        for (size_t k = 0; k < index[m].size(); k++) {
            if (written)
                written[index[m][k]] = wr[m][k];
            if (status)
                status[index[m][k]] = st[m][k];
        }
    }
    for (int i = 0; i < nframes; i++)
        if (host_status[i]) {
            if (written)
                written[i] = 0;
            if (status)
                status[i] = host_status[i];
        }
    if (async) { // what mcraw_pool_synchronize needs to report THIS thread its own batch
        Queued &q = g_queued[p];
        q.pool_id = p->id;
        q.n = nframes;
        q.index.swap(index);
        q.serial.swap(serial);
        q.host_status.swap(host_status);
    }
    return rc;
}

int mcraw_pool_synchronize(mcraw_pool *p, int32_t *status, int nframes)
{
    if (!p)
        return -1;
    const int G = static_cast<int>(p->members.size());
    Queued mine;
    {
        auto it = g_queued.find(p);
        if (it != g_queued.end() && it->second.pool_id == p->id)
            mine = it->second; // (kept: a second synchronize reports the same batch)
    }
    mine.index.resize(G);
    mine.serial.resize(G, 0);
    std::vector<std::vector<int32_t>> st(G);
    std::vector<int> rcs(G, 0);
    std::vector<int32_t> sticky(G, 0);
    std::vector<std::string> errs(G);
    std::vector<uint64_t> seq(G, 0);
    for (int m = 0; m < G; m++) {
        st[m].assign(mine.index[m].size(), 0);
        Member *mem = p->members[m];
        seq[m] = mem->run([&, m, mem] {
            rcs[m] = mcraw_ctx_synchronize(mem->ctx, nullptr, 0);
            if (rcs[m] == 0 && mine.serial[m] && !st[m].empty()) {
                const int r = mcraw_ctx_batch_status(mem->ctx, mine.serial[m], st[m].data(), static_cast<int>(st[m].size()));
                if (r < 0)
                    rcs[m] = r;
                else if (r == 1) // the member has forgotten this batch (other threads queued more than 64 behind it): its frames'
                    st[m].assign(st[m].size(), MCRAW_E_DEVICE); // outcome is UNKNOWN to this caller, which is not "decoded"
            }
            if (rcs[m] != 0)
                errs[m] = mcraw_last_error();
            sticky[m] = mcraw_ctx_errors(mem->ctx, 1);
        });
    }
    int rc = 0;
    int32_t any = 0;
    for (int m = 0; m < G; m++) {
        p->members[m]->wait(seq[m]);
        if (rcs[m] != 0 && rc == 0) {
            rc = rcs[m];
            g_pool_err = errs[m];
        }
        any |= sticky[m];
        for (size_t k = 0; k < mine.index[m].size(); k++) {
            any |= st[m][k];
            if (status && mine.index[m][k] < nframes)
                status[mine.index[m][k]] = st[m][k];
        }
    }
    for (int i = 0; i < mine.n && i < static_cast<int>(mine.host_status.size()); i++)
        if (mine.host_status[i]) {
            any |= mine.host_status[i];
            if (status && i < nframes)
                status[i] = mine.host_status[i];
        }
    if (rc != 0)
        return rc < 0 ? rc : -rc;
    return any;
}

int mcraw_pool_decode_batch(mcraw_pool *p, const mcraw_frame *frames, int nframes, size_t *written, int32_t *status)
{
    mcraw_pool_ticket *t = nullptr;
    if (int rc = mcraw_pool_decode_batch_async(p, frames, nframes, &t))
        return rc;
    return mcraw_pool_ticket_wait(t, written, status);
}

} // extern "C"
