// worker_pool.h -- a small fork/join pool for the host side of the GameContext: independent
// per-game work (the trees of a tree-search batch) is spread over a few host threads between two
// device batches.  The calling thread takes part; run() returns when every index has been done.
// No Python is touched inside (Wait / Steps run with the GIL released).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

namespace gogame {

class WorkerPool {
 public:
  // threads = 0: GOENV_HOST_THREADS if set, else min(16, the CPUs this process may run on); 1 = run inline
  // (tree search, 256 games x 16 search threads on a 16-CPU share: 4 / 8 / 12 / 16 threads 1.46 / 1.89 / 2.14 / 2.30 M leaf
  // evaluations per second)
  explicit WorkerPool(int threads = 0) {
    if (threads <= 0) {
      if (const char* e = std::getenv("GOENV_HOST_THREADS")) threads = std::atoi(e);
      if (threads <= 0) {
        unsigned hw = std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) hw = (unsigned)CPU_COUNT(&set);
        threads = (int)(hw ? (hw < 16u ? hw : 16u) : 1u);
      }
    }
    if (threads > 64) threads = 64;
    for (int i = 1; i < threads; ++i) workers_.emplace_back([this] { loop(); });
  }

  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_.store(true);
      generation_.fetch_add(1, std::memory_order_release);
    }
    wake_.notify_all();
    for (auto& t : workers_) t.join();
  }

  WorkerPool(const WorkerPool&) = delete;
  WorkerPool& operator=(const WorkerPool&) = delete;

  int threads() const { return (int)workers_.size() + 1; }

  // fn(i) for every i in [0, count); the first exception thrown by any fn(i) is rethrown here
  void run(int count, const std::function<void(int)>& fn) {
    if (count <= 0) return;
    if (workers_.empty() || count == 1) {
      for (int i = 0; i < count; ++i) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      count_ = count;
      next_.store(0);
      active_.store((int)workers_.size());
      error_ = nullptr;
      generation_.fetch_add(1, std::memory_order_release);
    }
    wake_.notify_all();   // (for the workers that have gone to sleep; the others are polling)
    work();
    // the workers end within one fn call of this thread's end: poll first, sleep only if that takes long
    for (int spin = 0; active_.load(std::memory_order_acquire) != 0; ++spin) {
      if (spin < (1 << 16)) {
        relax();
      } else {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return active_.load() == 0; });
      }
    }
    fn_ = nullptr;
    if (error_) std::rethrow_exception(error_);
  }

 private:
  static void relax() { __builtin_ia32_pause(); }

  void work() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= count_) return;
      try {
        (*fn_)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(m_);
        if (!error_) error_ = std::current_exception();
      }
    }
  }

  // A worker polls for the next batch for a while before it sleeps on the condition variable: the batches of a running
  // search follow one another within a few hundred microseconds, and waking a sleeping thread costs tens of them.
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      bool got = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int spin = 0;; ++spin) {
        if (generation_.load(std::memory_order_acquire) != seen) { got = true; break; }
        relax();
        if ((spin & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
      }
      if (!got) {
        std::unique_lock<std::mutex> lk(m_);
        wake_.wait(lk, [&] { return generation_.load() != seen; });
      }
      seen = generation_.load(std::memory_order_acquire);
      if (stop_.load()) return;
      work();
      if (active_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        std::lock_guard<std::mutex> lk(m_);
        done_.notify_one();
      }
    }
  }

  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable wake_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0}, active_{0};
  int count_ = 0;
  std::atomic<unsigned long> generation_{0};
  std::atomic<bool> stop_{false};
  std::exception_ptr error_;
};

}  // namespace gogame
