/**
 * hip_fm_loss.h — HipFMLoss : Loss.  The factorization-machine loss of the
 * reference (src/loss/fm_loss.h) behind the same interface, evaluated by HIP
 * kernels through the C ABI:
 *
 *   Predict : pred += X w + 1/2 sum((X V)^2 - (X.X)(V.V), 2), clamped to +-20 iff V_dim > 0
 *   CalcGrad: p = -y / (1 + exp(y pred)); grad_w += X'p; grad_V += X' diag(p) X V - diag((X.X)'p) V
 *
 * param[0] = weights (ragged: per key w [, V[0..V_dim)]), param[1] = w_pos,
 * param[2] = V_pos (-1 = absent) [, param[3] = pred for CalcGrad] — exactly
 * the reference's layout (fm_loss.h:50-65, :130-146).  Empty position arrays
 * mean dense weights without V (the V_dim == 0 / "logit" case).
 *
 * Unlike the reference, CalcGrad does not depend on a preceding Predict of the
 * same object (it recomputes X V on the device), so instances are stateless.
 */
#ifndef DIFACTO_HOST_HIP_FM_LOSS_H_
#define DIFACTO_HOST_HIP_FM_LOSS_H_
#include <vector>
#include "./device_context.h"
#include "difacto/loss.h"
#include "dmlc/parameter.h"

namespace difacto {

/*! \brief same key and range as the reference's FMLossParam (fm_loss.h:19-27) */
struct FMLossParam : public dmlc::Parameter<FMLossParam> {
  int V_dim;
  DMLC_DECLARE_PARAMETER(FMLossParam) { DMLC_DECLARE_FIELD(V_dim).set_range(0, 10000); }
};

class HipFMLoss : public Loss {
 public:
  /*! \brief fixed_V_dim >= 0 pins V_dim (used for "logit": V_dim = 0) */
  explicit HipFMLoss(int fixed_V_dim = -1) : fixed_V_dim_(fixed_V_dim) { param_.V_dim = 0; }
  virtual ~HipFMLoss() {}

  KWArgs Init(const KWArgs& kwargs) override {
    if (fixed_V_dim_ >= 0) {
      param_.V_dim = fixed_V_dim_;
      return kwargs;
    }
    return param_.InitAllowUnknown(kwargs);
  }

  void Predict(const dmlc::RowBlock<unsigned>& data, const std::vector<SArray<char>>& param,
               SArray<real_t>* pred) override {
    CHECK_EQ(param.size(), 3u);
    Predict(data, SArray<real_t>(param[0]), SArray<int>(param[1]), SArray<int>(param[2]), pred);
  }

  void Predict(const dmlc::RowBlock<unsigned>& data, const SArray<real_t>& weights, const SArray<int>& w_pos,
               const SArray<int>& V_pos, SArray<real_t>* pred) {
    CHECK_NOTNULL(pred);
    CHECK_EQ(pred->size(), data.size);
    if (w_pos.size()) CHECK_EQ(w_pos.size(), V_pos.size());
    const int k = w_pos.empty() ? 0 : param_.V_dim;
    DFH_CALL(dfh_fm_predict(DeviceContext::Get(), k, data.size, data.offset, data.index, data.value, weights.data(),
                            weights.size(), w_pos.empty() ? nullptr : w_pos.data(),
                            V_pos.empty() ? nullptr : V_pos.data(), w_pos.size(), pred->data()));
  }

  void CalcGrad(const dmlc::RowBlock<unsigned>& data, const std::vector<SArray<char>>& param,
                SArray<real_t>* grad) override {
    CHECK_EQ(param.size(), 4u);
    CalcGrad(data, SArray<real_t>(param[0]), SArray<int>(param[1]), SArray<int>(param[2]), SArray<real_t>(param[3]), grad);
  }

  void CalcGrad(const dmlc::RowBlock<unsigned>& data, const SArray<real_t>& weights, const SArray<int>& w_pos,
                const SArray<int>& V_pos, const SArray<real_t>& pred, SArray<real_t>* grad) {
    CHECK_NOTNULL(grad);
    CHECK_EQ(pred.size(), data.size);
    CHECK_EQ(grad->size(), weights.size());
    CHECK_NOTNULL(data.label);
    const int k = w_pos.empty() ? 0 : param_.V_dim;
    DFH_CALL(dfh_fm_calcgrad(DeviceContext::Get(), k, data.size, data.offset, data.index, data.value, data.label,
                             weights.data(), weights.size(), w_pos.empty() ? nullptr : w_pos.data(),
                             V_pos.empty() ? nullptr : V_pos.data(), w_pos.size(), pred.data(), grad->data()));
  }

  real_t Evaluate(dmlc::real_t const* label, const SArray<real_t>& pred) const override {
    float objv = 0;
    DFH_CALL(dfh_loss_evaluate(DeviceContext::Get(), label, pred.data(), pred.size(), &objv));
    return objv;
  }

  int V_dim() const { return param_.V_dim; }

 private:
  FMLossParam param_;
  int fixed_V_dim_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_HIP_FM_LOSS_H_
