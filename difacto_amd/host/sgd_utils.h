/**
 * sgd_utils.h — the two plain records the SGD learner moves through the Tracker
 * as byte strings: the job descriptor (scheduler -> worker) and the progress
 * record (worker -> scheduler).  Field names, order and the raw-bytes wire form
 * follow the reference (src/sgd/sgd_utils.h) so that its learner logic reads
 * unchanged; the packing code is shared through PodRecord.
 */
#ifndef DIFACTO_HOST_SGD_UTILS_H_
#define DIFACTO_HOST_SGD_UTILS_H_
#include <cstring>
#include <sstream>
#include <string>
#include <type_traits>
#include "difacto/base.h"

namespace difacto {
namespace sgd {

/*! \brief raw-bytes (de)serialisation for a trivially copyable record */
template <typename Derived>
struct PodRecord {
  void SerializeToString(std::string* out) const {
    static_assert(std::is_trivially_copyable<Derived>::value, "record must be plain data");
    out->assign(reinterpret_cast<const char*>(static_cast<const Derived*>(this)), sizeof(Derived));
  }
  /*! \brief false (and untouched) for an empty message; any other size must match exactly */
  bool Unpack(const char* bytes, size_t size) {
    if (size == 0) return false;
    CHECK_EQ(size, sizeof(Derived));
    std::memcpy(static_cast<Derived*>(this), bytes, sizeof(Derived));
    return true;
  }
};

/*! \brief what a worker is asked to do with which slice of the data */
struct Job : public PodRecord<Job> {
  enum Kind : int { kLoadModel = 1, kSaveModel = 2, kTraining = 3, kValidation = 4, kEvaluation = 5, kPrediction = 6 };
  int type = 0;
  int num_parts = 1;  // the data file is cut into this many byte ranges ...
  int part_idx = 0;   // ... and the job reads this one
  int epoch = 0;
  void ParseFromString(const std::string& msg) {
    CHECK(!msg.empty());
    Unpack(msg.data(), msg.size());
  }
};

/*! \brief sums a worker accumulates over the minibatches of a job (sgd_learner.cc:141-155) */
struct Progress : public PodRecord<Progress> {
  real_t loss = 0;     // logistic objective, summed over examples
  real_t penalty = 0;  // regulariser over the pulled weights
  real_t auc = 0;      // sum over batches of AUC * batch rows
  real_t nnz_w = 0;
  real_t nrows = 0;

  void Merge(const Progress& o) {
    loss += o.loss;
    penalty += o.penalty;
    auc += o.auc;
    nnz_w += o.nnz_w;
    nrows += o.nrows;
  }
  void Merge(const std::string& msg) {
    Progress o;
    if (o.Unpack(msg.data(), msg.size())) Merge(o);
  }
  void ParseFrom(const char* bytes, size_t size) { Unpack(bytes, size); }
  /*! \brief the text of the per-epoch log line (same wording as the reference prints) */
  std::string TextString() const {
    std::ostringstream line;
    line << "loss = " << loss << ", AUC = " << auc / nrows;
    return line.str();
  }
};

}  // namespace sgd
}  // namespace difacto
#endif  // DIFACTO_HOST_SGD_UTILS_H_
