// WorkerPool -- internal to the facade (host/Decoder.cpp); a header of its own so that tests/test_host_facade.py can run it
// under ThreadSanitizer without a GPU.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace motioncam {
namespace detail {

// A few host threads that stay (per-frame callers: starting eight threads for every frame's copy-out cost more than the copy).
// run(n, fn) calls fn(i) for i in [0, n) on the workers and the calling thread and returns when all are done; callers take turns.
// An exception out of fn -- on any thread -- is caught where it is thrown, the remaining indices are still handed out (a worker
// that stopped would leave the others' share undone), every thread is waited for, and the first exception is thrown again from run().
class WorkerPool {
public:
    explicit WorkerPool(unsigned workers)
    {
        for (unsigned t = 0; t < workers; t++)
            threads_.emplace_back([this]() { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : threads_)
            t.join();
    }
    void run(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0)
            return;
        std::lock_guard<std::mutex> one(run_); // (two chunks' copy-outs may be under way at once: they take turns)
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            busy_ = threads_.size();
            error_ = nullptr;
            gen_++;
        }
        cv_.notify_all();
        work(fn, n);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_.wait(lk, [this]() { return busy_ == 0; }); // (whatever happened: nobody calls fn any more behind this line)
            fn_ = nullptr;
            err = error_;
            error_ = nullptr;
        }
        if (err)
            std::rethrow_exception(err);
    }

private:
    void work(const std::function<void(size_t)> &fn, size_t n)
    {
        for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) {
            try {
                fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu_);
                if (!error_)
                    error_ = std::current_exception();
            }
        }
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(size_t)> *fn;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&]() { return quit_ || gen_ != seen; });
                if (quit_)
                    return;
                seen = gen_;
                fn = fn_;
                n = n_;
            }
            work(*fn, n);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--busy_ == 0)
                    done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, run_;
    std::condition_variable cv_, done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0, busy_ = 0;
    std::atomic<size_t> next_{0};
    unsigned long long gen_ = 0;
    bool quit_ = false;
    std::exception_ptr error_; // the first exception of the run under way
};

} // namespace detail
} // namespace motioncam
