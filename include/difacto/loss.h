/**
 * difacto/loss.h — Loss: forward (Predict), objective (Evaluate) and backward
 * (CalcGrad) over a CSR minibatch.  Interface-compatible with the reference's
 * include/difacto/loss.h (:25-85).
 */
#ifndef DIFACTO_LOSS_H_
#define DIFACTO_LOSS_H_
#include <cmath>
#include <string>
#include <vector>
#include "./base.h"
#include "./sarray.h"
#include "dmlc/data.h"
#include "dmlc/omp.h"

namespace difacto {

class Loss {
 public:
  /**
   * \brief factory.  "fm" (and "logit", its V_dim = 0 case) return the HIP
   *        implementation; nthreads is kept for signature compatibility.
   */
  static Loss* Create(const std::string& type, int nthreads = DEFAULT_NTHREADS);
  Loss() : nthreads_(DEFAULT_NTHREADS) {}
  virtual ~Loss() {}

  virtual KWArgs Init(const KWArgs& kwargs) = 0;

  /*! \brief pred += f(data; param); see the concrete loss for the param layout */
  virtual void Predict(const dmlc::RowBlock<unsigned>& data, const std::vector<SArray<char>>& param,
                       SArray<real_t>* pred) = 0;

  /*! \brief logistic objective sum_i log(1 + exp(-y_i pred_i)), y = label > 0 ? 1 : -1 */
  virtual real_t Evaluate(dmlc::real_t const* label, const SArray<real_t>& pred) const {
    double objv = 0;
    for (size_t i = 0; i < pred.size(); ++i) {
      const real_t y = label[i] > 0 ? 1 : -1;
      objv += std::log(1 + std::exp(static_cast<double>(-y * pred[i])));
    }
    return static_cast<real_t>(objv);
  }

  /*! \brief grad += df/dparam, same shape as the weights in param */
  virtual void CalcGrad(const dmlc::RowBlock<unsigned>& data, const std::vector<SArray<char>>& param,
                        SArray<real_t>* grad) = 0;

  void set_nthreads(int nthreads) {
    CHECK_GT(nthreads, 1);
    CHECK_LT(nthreads, 50);
    nthreads_ = nthreads;
  }
  int nthreads_;
};

}  // namespace difacto
#endif  // DIFACTO_LOSS_H_
