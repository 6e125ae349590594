/**
 * local_tracker.h — in-process Tracker: one executor thread drains a job queue
 * and hands every result to the monitor.  Same contract as the reference's
 * LocalTracker over AsyncLocalTracker (src/tracker/local_tracker.h,
 * async_local_tracker.h): Issue() returns at once, jobs run one at a time in
 * order, NumRemains() counts queued + running jobs.
 */
#ifndef DIFACTO_HOST_LOCAL_TRACKER_H_
#define DIFACTO_HOST_LOCAL_TRACKER_H_
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include "difacto/tracker.h"

namespace difacto {

class LocalTracker : public Tracker {
 public:
  LocalTracker() : remains_(0), stop_(false) { thread_ = std::thread(&LocalTracker::Loop, this); }
  virtual ~LocalTracker() { Stop(); }
  KWArgs Init(const KWArgs& kwargs) override { return kwargs; }

  void Issue(const std::vector<std::pair<int, std::string>>& jobs) override {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (const auto& j : jobs) queue_.push_back(j);
      remains_ += static_cast<int>(jobs.size());
    }
    cv_.notify_all();
  }
  int NumRemains() override {
    std::lock_guard<std::mutex> lk(mu_);
    return remains_;
  }
  void Clear() override {
    std::lock_guard<std::mutex> lk(mu_);
    remains_ -= static_cast<int>(queue_.size());
    queue_.clear();
  }
  void Stop() override {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [this] { return remains_ == 0 || stop_; });
      stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
  }
  void SetMonitor(const Monitor& monitor) override {
    std::lock_guard<std::mutex> lk(mu_);
    monitor_ = monitor;
  }
  void SetExecutor(const Executor& executor) override {
    std::lock_guard<std::mutex> lk(mu_);
    executor_ = executor;
  }
  void Wait() override {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return stop_; });
  }

 private:
  void Loop() {
    while (true) {
      std::pair<int, std::string> job;
      Executor exec;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || (!queue_.empty() && executor_); });
        if (stop_) return;
        job = queue_.front();
        queue_.pop_front();
        exec = executor_;
      }
      std::string rets;
      exec(job.second, &rets);
      Monitor mon;
      {
        std::lock_guard<std::mutex> lk(mu_);
        mon = monitor_;
      }
      if (mon) mon(job.first, rets);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --remains_;
      }
      done_cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<std::pair<int, std::string>> queue_;
  int remains_;
  bool stop_;
  Monitor monitor_;
  Executor executor_;
  std::thread thread_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_LOCAL_TRACKER_H_
