// taxator-tk_amd/host/channel.hh -- blocking queue between the pipeline stages of the host driver.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>

namespace taxator {

template <class T>
class Channel {  // unbounded; depth is bounded by the number of objects in circulation
 public:
  void push(T v) { { std::lock_guard<std::mutex> g(m_); q_.push_back(v); } cv_.notify_one(); }
  T pop() { std::unique_lock<std::mutex> g(m_); cv_.wait(g, [&] { return !q_.empty(); }); T v = q_.front(); q_.pop_front(); return v; }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<T> q_;
};

}  // namespace taxator
