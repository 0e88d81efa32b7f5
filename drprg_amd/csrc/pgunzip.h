// pgunzip.h -- one plain gzip stream inflated by many threads (NEXT-4: `.fq.gz` is the reference's
// normal input, /root/reference/docs/src/guide/predict.md:25-36, and one zlib / libdeflate thread is three orders of
// magnitude under the kernels).
//
// A deflate stream has no index, but it can be entered at any block boundary if the 32 KB of text before it are treated as
// unknown: the compressed file is cut into chunks, every chunk's thread looks for the first dynamic-Huffman block header
// at or after its offset (bit by bit: header fields in range, complete code-length / literal / distance codes, and the
// block decodes) and inflates from there into 16-bit symbols -- a byte, or 0x8000 + i for "byte i of the 32 KB before my
// start" -- until it reaches the first dynamic block at or after the next chunk's offset.  The chunks are then stitched in
// file order: a chunk is accepted iff it starts exactly where its predecessor ended (anything else -- a header look-alike,
// no dynamic block in range, an over-long chunk -- is inflated again from the known position); the last 32 KB of every
// accepted chunk, resolved, are the next one's window, and with the windows known all symbols are resolved to bytes by all
// threads.  CRC-32 and length of every member are checked like gzip does.  (The approach of pugz / rapidgzip; own code.)
#pragma once
#include "common.h"
#include <condition_variable>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace drprg {

// Threads that stay: run(fn, n) runs fn on n threads (the caller is one of them) and returns when all are back.  The rounds of the
// producer and the read() calls of the consumer each started their threads anew -- 8 + 39 times 31 threads for 4 M reads, a
// millisecond per call on the calling thread.
class Crew {
public:
    explicit Crew(int helpers) : n_(helpers > 0 ? helpers : 0) {}
    ~Crew()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    template <typename F> void run(F& fn, int n)
    {
        const int helpers = std::min(n - 1, n_);
        if (helpers > 0) {
            std::unique_lock<std::mutex> g(mu_);
            while ((int)th_.size() < helpers) th_.emplace_back([this] { loop(); }); // (started when first needed)
            job_ = [&fn] { fn(); };
            tickets_ = helpers;
            running_ = helpers;
            ++gen_;
            g.unlock();
            cv_.notify_all();
        }
        std::exception_ptr helper_failed;
        auto join = [&] {
            if (helpers > 0) {
                std::unique_lock<std::mutex> g(mu_);
                done_.wait(g, [&] { return running_ == 0; });
                job_ = nullptr;
                helper_failed = failed_;
                failed_ = nullptr;
            }
        };
        try {
            fn();
        } catch (...) { // (the helpers still run fn: it must outlive them)
            join();
            throw;
        }
        join();
        if (helper_failed) std::rethrow_exception(helper_failed); // (what a helper threw comes out of run() on the calling thread)
    }

private:
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_.wait(g, [&] { return quit_ || (gen_ != seen && tickets_ > 0); });
            if (quit_) return;
            seen = gen_;
            --tickets_;
            std::function<void()> job = job_;
            g.unlock();
            std::exception_ptr err;
            try {
                job();
            } catch (...) {
                err = std::current_exception();
            }
            g.lock();
            if (err && !failed_) failed_ = err;
            if (--running_ == 0) done_.notify_all();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::function<void()> job_;
    uint64_t gen_ = 0;
    int tickets_ = 0, running_ = 0;
    bool quit_ = false;
    std::exception_ptr failed_;
};


class ParallelGunzip {
public:
    // data/len: the whole compressed file (stays mapped for the lifetime of the object); chunk_bytes 0 = automatic
    ParallelGunzip(const unsigned char* data, size_t len, int threads, size_t chunk_bytes = 0);
    ~ParallelGunzip();
    // up to cap bytes of text into dst; 0 = end of the stream.  Throws Error(DRPRG_EIO) on a corrupt stream.
    size_t read(char* dst, size_t cap);
    // chunks accepted as their threads inflated them / inflated again from the known position (diagnostics, tests)
    uint64_t chunks_accepted() const;
    uint64_t chunks_redone() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace drprg
