// host_threads.h -- work(i) for i in [0, n) on host threads, with nothing escaping a thread.
//
// The library is built with exceptions on and its C-ABI wrappers turn a std::bad_alloc into PHANT_E_OOM -- but an exception that
// leaves a std::thread's body is std::terminate, and so is destroying a joinable std::thread while another exception unwinds.  Here
// every body is caught where it runs, every thread that was started is joined whatever happens (a thread that could not be
// started has its share run by the caller), and the first failure is rethrown on the CALLING thread once all are back.
#pragma once
#include <atomic>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

namespace phant {

template <class F>
inline void parallel_guarded(size_t n, F&& work) {
    if (n == 0) return;
    std::atomic<int> failed{0};  // 1 = out of memory, 2 = anything else
    auto body = [&](size_t i) noexcept {
        try {
            work(i);
        } catch (const std::bad_alloc&) {
            int none = 0;
            failed.compare_exchange_strong(none, 1);
        } catch (...) {
            int none = 0;
            failed.compare_exchange_strong(none, 2);
        }
    };
    std::vector<std::thread> th;
    size_t started = 1;
    try {
        th.reserve(n - 1);
        for (; started < n; ++started) th.emplace_back(body, started);
    } catch (...) {  // (no memory for the vector, or the system refuses another thread: fewer threads, the same work)
    }
    body(0);
    for (size_t i = started; i < n; ++i) body(i);
    for (std::thread& t : th)
        if (t.joinable()) t.join();
    if (failed.load() == 1) throw std::bad_alloc();
    if (failed.load() == 2) throw std::runtime_error("a worker thread failed");
}

}  // namespace phant
