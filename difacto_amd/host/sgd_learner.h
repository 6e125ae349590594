/**
 * sgd_learner.h — SGDLearner: the reference's SGD learner (src/sgd/sgd_learner.{h,cc})
 * with its worker loop running on the GPU.
 *
 * Scheduler side (epochs, job issue, progress merge, stop criteria, epoch-end
 * callbacks) follows the reference line by line.  The worker loop
 * IterateData has two implementations, chosen by the `device_path` key:
 *
 *   fused   (default) every minibatch goes to the device as raw CSR; Localizer,
 *           Pull, Predict, Evaluate, CalcGrad, Push and the FTRL/AdaGrad update
 *           run there (dfh_localize / dfh_sgd_step); batch t+1 is prepared on a
 *           second stream while batch t trains — the overlap the reference gets
 *           from its reader thread + batch tracker (sgd_learner.cc:196-224)
 *   literal the reference's own sequence of interface calls with host arrays:
 *           Localizer::Compact -> Store::Push(kFeaCount) -> Store::Pull ->
 *           GetPos -> Loss::Predict -> Loss::Evaluate -> Loss::CalcGrad ->
 *           Store::Push(kGradient), each served by the device adaptors
 */
#ifndef DIFACTO_HOST_SGD_LEARNER_H_
#define DIFACTO_HOST_SGD_LEARNER_H_
#include <cstdio>
#include <functional>
#include <string>
#include <vector>
#include "./device_store.h"
#include "./sgd_param.h"
#include "./sgd_utils.h"
#include "difacto/learner.h"
#include "difacto/loss.h"
#include "difacto/store.h"

namespace difacto {

class SGDLearner : public Learner {
 public:
  SGDLearner() : store_(nullptr), loss_(nullptr) {}
  virtual ~SGDLearner();

  KWArgs Init(const KWArgs& kwargs) override;

  /*! \brief called after every epoch with the merged training / validation progress */
  void AddEpochEndCallback(const std::function<void(int epoch, const sgd::Progress& train, const sgd::Progress& val)>& cb) {
    epoch_end_callback_.push_back(cb);
  }
  DeviceSGDUpdater* GetUpdater() {
    return CHECK_NOTNULL(static_cast<DeviceSGDUpdater*>(CHECK_NOTNULL(store_)->updater().get()));
  }

 protected:
  void RunScheduler() override;
  void Process(const std::string& args, std::string* rets) override;

 private:
  void RunEpoch(int epoch, int job_type, sgd::Progress* prog);
  /*! \brief task = predict: one forward pass of the loaded model over the data, predictions to pred_out */
  void RunPrediction();
  /*! \brief prediction jobs: the step's logits, one line per example, appended to the job's output file */
  void WritePredictions(dfh_batch* b);
  /*! \brief the data a job reads: data_in for training, data_val for validation, either for prediction */
  const std::string& JobData(const sgd::Job& job) const;
  void IterateData(const sgd::Job& job, sgd::Progress* prog);
  void IterateDataFused(const sgd::Job& job, sgd::Progress* prog);
  void IterateDataLiteral(const sgd::Job& job, sgd::Progress* prog);
  void IterateDataSharded(const sgd::Job& job, sgd::Progress* prog);
  /*! \brief sums the record over the ranks of a sharded run (identity for one process) */
  void MergeAcrossRanks(sgd::Progress* prog);
  real_t EvaluatePenalty(const SArray<real_t>& weights, const SArray<int>& w_pos, const SArray<int>& V_pos);
  void GetPos(const SArray<int>& len, SArray<int>* w_pos, SArray<int>* V_pos);
  void SaveModel();
  void LoadModel();

  Store* store_;
  Loss* loss_;
  SGDLearnerParam param_;
  int blk_nthreads_ = DEFAULT_NTHREADS;
  std::vector<std::function<void(int, const sgd::Progress&, const sgd::Progress&)>> epoch_end_callback_;
  // device batches of the fused path (double-buffered)
  // minibatch objects in rotation.  The fused loop uses kFusedBatches of them: the reader delivers minibatches in bursts
  // (ten at a time, when a shuffle buffer becomes current) and the loop queues a whole burst on the device without
  // waiting for any step — with two or three objects the device idled while the loop waited for the reader and the
  // loop waited while the device caught up.  The sharded loop uses the first two.
  static constexpr int kFusedBatches = 12;
  dfh_batch* batch_[kFusedBatches] = {};
  size_t batch_rows_ = 0, batch_nnz_ = 0;
  bool batch_arena_ = false;   // batch_[] were carved from ONE device allocation (dfh_batch_create_many): freed when the last one goes
  FILE* pred_file_ = nullptr;       // open while a prediction job runs
  std::vector<float> pred_buf_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_SGD_LEARNER_H_
