// worker_pool_tsan -- the facade's copy-out workers under ThreadSanitizer: several callers at once (two chunks' copy-outs overlap
// in Decoder::loadFrames), runs of every size around the worker count, pools made and destroyed while idle and right after work.
#include "WorkerPool.hpp"

#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>

int main()
{
    using motioncam::detail::WorkerPool;
    long bad = 0;
    for (int round = 0; round < 20; round++) {
        WorkerPool pool(3 + round % 5);
        std::vector<std::thread> callers;
        std::atomic<long> wrong{0};
        for (int c = 0; c < 3; c++)
            callers.emplace_back([&, c]() {
                for (size_t n = 0; n < 24; n++) {
                    std::vector<unsigned char> src(4096 * (n + 1), static_cast<unsigned char>(c + n)), dst(src.size(), 0);
                    const size_t slice = (src.size() + n) / (n + 1);
                    pool.run(n + 1, [&](size_t t) {
                        const size_t lo = std::min(src.size(), t * slice), hi = std::min(src.size(), lo + slice);
                        std::memcpy(dst.data() + lo, src.data() + lo, hi - lo);
                    });
                    if (dst != src)
                        wrong++;
                }
                pool.run(0, [&](size_t) { wrong++; });
            });
        for (std::thread &t : callers)
            t.join();
        bad += wrong.load();
    }
    { WorkerPool idle(4); }
    { // an exception out of the work -- on the calling thread or on a worker -- comes out of run(), once, after every index was
      // handed out and every thread has stopped calling; the pool works afterwards
        WorkerPool pool(4);
        for (int round = 0; round < 50; round++) {
            std::atomic<int> calls{0};
            bool thrown = false;
            try {
                pool.run(16, [&](size_t i) {
                    calls++;
                    if (i == static_cast<size_t>(round % 16) || i == 15)
                        throw std::runtime_error("slice failed");
                });
            } catch (const std::runtime_error &) {
                thrown = true;
            }
            if (!thrown || calls.load() != 16)
                bad++;
            std::atomic<int> after{0};
            pool.run(8, [&](size_t) { after++; });
            if (after.load() != 8)
                bad++;
        }
    }
    std::printf("wrong %ld\n", bad);
    return bad ? 1 : 0;
}
