/**
 * difacto/tracker.h — Tracker: the job queue between the scheduler loop and the
 * executors.  Interface-compatible with the reference's include/difacto/tracker.h
 * (:60-113); the only implementation here is the in-process one.
 */
#ifndef DIFACTO_TRACKER_H_
#define DIFACTO_TRACKER_H_
#include <functional>
#include <string>
#include <utility>
#include <vector>
#include "./base.h"

namespace difacto {

class Tracker {
 public:
  static Tracker* Create();
  Tracker() {}
  virtual ~Tracker() {}
  virtual KWArgs Init(const KWArgs& kwargs) = 0;

  /*! \brief queue one job (args) for node_id; returns at once */
  void Issue(int node_id, std::string args) { Issue({std::make_pair(node_id, args)}); }
  virtual void Issue(const std::vector<std::pair<int, std::string>>& jobs) = 0;
  /*! \brief jobs not finished yet */
  virtual int NumRemains() = 0;
  /*! \brief drop the jobs that have not started */
  virtual void Clear() = 0;
  /*! \brief finish everything, then stop the executors */
  virtual void Stop() = 0;

  typedef std::function<void(int node_id, const std::string& rets)> Monitor;
  virtual void SetMonitor(const Monitor& monitor) = 0;
  typedef std::function<void(const std::string& args, std::string* rets)> Executor;
  virtual void SetExecutor(const Executor& executor) = 0;
  /*! \brief executor side: block until Stop() */
  virtual void Wait() = 0;
};

}  // namespace difacto
#endif  // DIFACTO_TRACKER_H_
