/**
 * difacto/learner.h — Learner: the algorithm driver.  Interface-compatible with
 * the reference's include/difacto/learner.h (:26-69): Create(type), Init(kwargs)
 * returning the unknown kwargs, Run(), Stop(), and the RunScheduler / Process
 * pair connected through the Tracker.
 */
#ifndef DIFACTO_LEARNER_H_
#define DIFACTO_LEARNER_H_
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "./base.h"
#include "./tracker.h"
#include "dmlc/io.h"

namespace difacto {

class Learner {
 public:
  /*! \brief "sgd" is the learner this build accelerates; others are reported as unsupported */
  static Learner* Create(const std::string& type);
  Learner() : tracker_(nullptr) {}
  virtual ~Learner() { delete tracker_; }

  virtual KWArgs Init(const KWArgs& kwargs);

  /**
   * \brief the reference runs the scheduler loop on the scheduler node and parks the other roles in
   * the tracker (learner.h:38-45).  The multi-GPU build has no separate scheduler process: every rank
   * runs the (deterministic) loop itself on the same merged progress, and executes its own share of
   * the jobs — see Tracker::Create and SGDLearner::RunEpoch.
   */
  void Run() { RunScheduler(); }
  void Stop() { tracker_->Stop(); }

 protected:
  /*! \brief scheduler loop: issue jobs, merge their results */
  virtual void RunScheduler() = 0;
  /*! \brief executor: run one job (serialized in args), serialize the result into rets */
  virtual void Process(const std::string& args, std::string* rets) = 0;
  Tracker* tracker_;
};

}  // namespace difacto
#endif  // DIFACTO_LEARNER_H_
