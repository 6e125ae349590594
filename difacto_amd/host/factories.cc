/**
 * factories.cc — the string/role factories of the interfaces
 * (reference: src/loss/loss.cc:13-26, src/store/store.cc:8-15,
 * src/tracker/tracker.cc:8-15, src/learner.cc:15-35).
 */
#include "./device_store.h"
#include "./hip_fm_loss.h"
#include "./local_tracker.h"
#include "./sgd_learner.h"
#include "./sharded_store.h"
#include "difacto/learner.h"
#include "difacto/loss.h"
#include "difacto/store.h"
#include "difacto/tracker.h"

namespace difacto {

Loss* Loss::Create(const std::string& type, int nthreads) {
  Loss* loss = nullptr;
  if (type == "fm") {
    loss = new HipFMLoss();
  } else if (type == "logit") {
    loss = new HipFMLoss(0);  // plain logistic regression = FM with V_dim = 0 (fm_loss.h:77, :168)
  } else {
    LOG(FATAL) << "unknown loss type: " << type << " (this build provides fm and logit)";
  }
  loss->set_nthreads(nthreads);
  return loss;
}

// The reference's distributed store is a TODO (store.cc:9-10: LOG(FATAL) << "not implemented"); here
// DMLC_ROLE selects the key-range-sharded multi-GPU store (one process per GPU, sharded_store.h).
Store* Store::Create() {
  if (IsDistributed()) return new ShardedDeviceStore();
  return new DeviceStore();
}

// Every rank runs the scheduler loop itself (same data split, same merged progress, hence the same
// decisions) and executes its own share of the jobs in-process: no DistTracker is needed.
Tracker* Tracker::Create() { return new LocalTracker(); }

Learner* Learner::Create(const std::string& type) {
  if (type == "sgd") return new SGDLearner();
  LOG(FATAL) << "learner type " << type << " is not part of this build (sgd only)";
  return nullptr;
}

KWArgs Learner::Init(const KWArgs& kwargs) {
  tracker_ = Tracker::Create();
  auto remain = tracker_->Init(kwargs);
  using namespace std::placeholders;
  tracker_->SetExecutor(std::bind(&Learner::Process, this, _1, _2));
  return remain;
}

}  // namespace difacto
