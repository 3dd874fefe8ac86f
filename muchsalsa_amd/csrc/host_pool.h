// host_pool.h -- the parked host threads the library's short loops run on (graph stage, first touches of page-locked
// blocks; the loaders' few long loops start their own threads: sharing the pool between the PAF parser and the two
// sequence parsers measured no better).  A loop that starts its own threads pays 0.3-0.5 ms for fifteen of them; the host stages have dozens of loops.
// A loop invites helpers and takes part itself: items are handed out by a counter, whoever is free takes the next one, and
// the caller leaves when the counter has run out and every helper that joined has left (a helper that arrives later finds
// nothing to do and never touches the caller's frame).  Loops of several callers share the pool; it grows until every
// waiting invitation has a parked thread (at most 63 threads) and lives as long as the library.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace msgpu {

class HostPool {
public:
  static HostPool &get() {
    static HostPool p;
    return p;
  }
  // body(i) for every i < n_items, on up to `want` threads including this one; the first exception is rethrown here
  template <class Body> void run(unsigned want, size_t n_items, Body &&body) {
    if (want <= 1 || n_items <= 1) {
      for (size_t i = 0; i < n_items; ++i) body(i);
      return;
    }
    auto r   = std::make_shared<Region>();
    r->n     = n_items;
    r->body  = [&body](size_t i) { body(i); };
    const unsigned helpers = static_cast<unsigned>(std::min<size_t>(want - 1, n_items - 1));
    invite(r, helpers);
    work(*r);
    std::unique_lock<std::mutex> lk(r->m);
    r->cv.wait(lk, [&] { return r->active == 0; });
    if (r->err) std::rethrow_exception(r->err);
  }

private:
  struct Region {
    std::atomic<size_t>          next{0};
    size_t                       n = 0;
    std::function<void(size_t)>  body;
    std::mutex                   m;
    std::condition_variable      cv;
    int                          active = 0; // helpers inside work()
    std::exception_ptr           err;
  };
  static void work(Region &r) {
    for (size_t i = r.next.fetch_add(1); i < r.n; i = r.next.fetch_add(1)) {
      try {
        r.body(i);
      } catch (...) {
        r.next.store(r.n); // nothing more is handed out
        std::lock_guard<std::mutex> g(r.m);
        if (!r.err) r.err = std::current_exception();
      }
    }
  }
  void invite(const std::shared_ptr<Region> &r, unsigned helpers) {
    std::lock_guard<std::mutex> g(m_);
    for (unsigned k = 0; k < helpers; ++k) q_.push_back(r);
    try { // a parked thread for every invitation that is waiting (loops of several callers do not queue behind each other)
      while (workers_.size() - busy_ < q_.size() && workers_.size() < 63) workers_.emplace_back([this] { loop(); });
    } catch (std::system_error const &) {} // (no thread to be had: the loop runs on those there are)
    if (helpers == 1) cv_.notify_one();
    else cv_.notify_all();
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Region> r;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return; // stop_
        r = std::move(q_.front());
        q_.pop_front();
        ++busy_;
      }
      if (r->next.load() < r->n) { // (else the loop is over already)
        {
          std::lock_guard<std::mutex> g(r->m);
          ++r->active;
        }
        work(*r);
        {
          std::lock_guard<std::mutex> g(r->m);
          if (--r->active == 0) r->cv.notify_all();
        }
      }
      r.reset();
      std::lock_guard<std::mutex> g(m_);
      --busy_;
    }
  }
  HostPool() = default;
  ~HostPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      q_.clear();
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
  }
  std::mutex                          m_;
  std::condition_variable             cv_;
  std::deque<std::shared_ptr<Region>> q_;
  std::vector<std::thread>            workers_;
  size_t                              busy_ = 0; // workers inside a loop
  bool                                stop_ = false;
};

} // namespace msgpu
