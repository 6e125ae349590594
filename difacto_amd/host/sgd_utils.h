/**
 * sgd_utils.h — the job descriptor and the progress record the SGD learner
 * passes through the Tracker as byte strings; same fields and serialization
 * as the reference's src/sgd/sgd_utils.h.
 */
#ifndef DIFACTO_HOST_SGD_UTILS_H_
#define DIFACTO_HOST_SGD_UTILS_H_
#include <cstring>
#include <sstream>
#include <string>
#include "difacto/base.h"

namespace difacto {
namespace sgd {

struct Job {
  static const int kLoadModel = 1;
  static const int kSaveModel = 2;
  static const int kTraining = 3;
  static const int kValidation = 4;
  static const int kEvaluation = 5;
  int type;
  int num_parts;  // parts the data file is cut into
  int part_idx;   // the part this job reads
  int epoch;
  Job() : type(0), num_parts(1), part_idx(0), epoch(0) {}
  void SerializeToString(std::string* str) const { str->assign(reinterpret_cast<const char*>(this), sizeof(Job)); }
  void ParseFromString(const std::string& str) {
    CHECK_EQ(str.size(), sizeof(Job));
    memcpy(this, str.data(), sizeof(Job));
  }
};

struct Progress {
  real_t loss = 0;     // logistic objective, summed over examples
  real_t penalty = 0;  // regulariser over the pulled weights
  real_t auc = 0;      // sum over batches of AUC * batch rows
  real_t nnz_w = 0;
  real_t nrows = 0;

  std::string TextString() {
    std::stringstream ss;
    ss << "loss = " << loss << ", AUC = " << auc / nrows;
    return ss.str();
  }
  void SerializeToString(std::string* str) const { str->assign(reinterpret_cast<const char*>(this), sizeof(Progress)); }
  void ParseFrom(const char* data, size_t size) {
    if (size == 0) return;
    CHECK_EQ(size, sizeof(Progress));
    memcpy(this, data, sizeof(Progress));
  }
  void Merge(const std::string& str) {
    Progress other;
    other.ParseFrom(str.data(), str.size());
    Merge(other);
  }
  void Merge(const Progress& o) {
    loss += o.loss; penalty += o.penalty; auc += o.auc; nnz_w += o.nnz_w; nrows += o.nrows;
  }
};

}  // namespace sgd
}  // namespace difacto
#endif  // DIFACTO_HOST_SGD_UTILS_H_
