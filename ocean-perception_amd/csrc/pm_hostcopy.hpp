// pm_hostcopy.hpp -- a few persistent host threads that share large memcpys (host side of the host-buffer entry points).
// One core copies ~10 GB/s; a 1280x720 float map is 3.7 MB, and a synchronous Match() packs 9 MB into the pinned staging
// buffer and unpacks 7 MB out of it: ~1.2 ms of a 3.8 ms call on one thread (round 2: 266 pairs/s against 318 with
// resident buffers).  Four threads take that to ~0.35 ms.
#pragma once

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace pm {

class CopyPool {
 public:
  explicit CopyPool(int workers = 3) {
    for (int i = 0; i < workers; ++i) threads_.emplace_back([this, i] { Loop(i); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;

  // rows x row_bytes from src (row stride src_step) to dst (row stride dst_step); the caller's thread copies its share.
  void Copy2D(void* dst, size_t dst_step, const void* src, size_t src_step, size_t row_bytes, int rows) {
    const int parts = (int)threads_.size() + 1;
    if ((size_t)rows * row_bytes < (1u << 20) || rows < parts) {  // small: not worth a wake-up
      Part(dst, dst_step, src, src_step, row_bytes, 0, rows);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = Job{dst, dst_step, src, src_step, row_bytes, rows, parts};
      pending_ = parts - 1;
      ++generation_;
    }
    cv_.notify_all();
    Slice(job_, parts - 1);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
  }

 private:
  struct Job {
    void* dst;
    size_t dst_step;
    const void* src;
    size_t src_step, row_bytes;
    int rows, parts;
  };
  static void Part(void* dst, size_t dst_step, const void* src, size_t src_step, size_t row_bytes, int r0, int r1) {
    if (dst_step == row_bytes && src_step == row_bytes) {
      std::memcpy((char*)dst + (size_t)r0 * row_bytes, (const char*)src + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
      return;
    }
    for (int y = r0; y < r1; ++y) std::memcpy((char*)dst + (size_t)y * dst_step, (const char*)src + (size_t)y * src_step, row_bytes);
  }
  static void Slice(const Job& j, int part) {
    const int r0 = (int)((long long)j.rows * part / j.parts), r1 = (int)((long long)j.rows * (part + 1) / j.parts);
    Part(j.dst, j.dst_step, j.src, j.src_step, j.row_bytes, r0, r1);
  }
  void Loop(int index) {
    unsigned long long seen = 0;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        j = job_;
      }
      Slice(j, index);
      {
        std::lock_guard<std::mutex> lk(m_);
        --pending_;
      }
      done_.notify_one();
    }
  }

  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  Job job_{};
  int pending_ = 0;
  unsigned long long generation_ = 0;
  bool stop_ = false;
};

}  // namespace pm
