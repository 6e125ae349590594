/**
 * lbfgs_mini.h — a minimal L-BFGS outer loop over the Loss interface: the second customer of
 * HipFMLoss (the reference's LBFGSLearner creates its Loss through the same factory,
 * src/lbfgs/lbfgs_learner.cc:364-367, and calls Predict -> CalcGrad -> Evaluate per block of rows,
 * :262-270).  It exists to drive a Loss through the golden objective trajectories of
 * tests/cpp/lbfgs_learner_test.cc (Basic: V_dim 0; WithV: FM with V_dim 5), which exercise
 * Predict / CalcGrad with V far more thoroughly than the reference's SGD tests do.
 *
 * Restated from the reference, one worker / one block of rows, no tile store, no thread pool:
 *   RunScheduler          src/lbfgs/lbfgs_learner.cc:14-126   (direction, Wolfe line search)
 *   LBFGSUpdater          src/lbfgs/lbfgs_updater.h:33-205    (s / y history, regulariser, clamp)
 *   Twoloop               src/lbfgs/lbfgs_twoloop.h:19-126    (vector-free two-loop recursion)
 *   Inner / Add / Times   src/lbfgs/lbfgs_utils.h:62-98
 * The BCD / L-BFGS learners themselves stay out of scope.
 */
#ifndef DIFACTO_HOST_LBFGS_MINI_H_
#define DIFACTO_HOST_LBFGS_MINI_H_
#include <algorithm>
#include <functional>
#include <vector>
#include "difacto/loss.h"

namespace difacto {
namespace lbfgs_mini {

typedef std::vector<real_t> Vec;

inline double Inner(const Vec& a, const Vec& b) {  // float products, double sum (lbfgs_utils.h:62-72)
  double r = 0;
  for (size_t i = 0; i < a.size(); ++i) r += a[i] * b[i];
  return r;
}
inline void Add(real_t x, const Vec& a, Vec* b) {  // lbfgs_utils.h:74-88
  if (x == 0) return;
  if (x == 1) {
    for (size_t i = 0; i < a.size(); ++i) (*b)[i] += a[i];
  } else {
    for (size_t i = 0; i < a.size(); ++i) (*b)[i] += x * a[i];
  }
}

/*! \brief lbfgs_twoloop.h:19-126 */
class Twoloop {
 public:
  void CalcIncreB(const std::vector<Vec>& s, const std::vector<Vec>& y, const Vec& grad, std::vector<real_t>* incr_B) {
    const int m = static_cast<int>(s.size());
    incr_B->resize(6 * m + 1);
    for (int i = 0; i < m; ++i) {
      (*incr_B)[i] = Inner(s.back(), s[i]);
      (*incr_B)[i + m] = Inner(s.back(), y[i]);
      (*incr_B)[i + 2 * m] = Inner(y.back(), s[i]);
      (*incr_B)[i + 3 * m] = Inner(y.back(), y[i]);
      (*incr_B)[i + 4 * m] = Inner(grad, s[i]);
      (*incr_B)[i + 5 * m] = Inner(grad, y[i]);
    }
    (*incr_B)[6 * m] = Inner(grad, grad);
  }
  void ApplyIncreB(const std::vector<real_t>& incr_B) {
    const int m = static_cast<int>((incr_B.size() - 1) / 6);
    CHECK(m == m_ + 1 || m == m_);
    const int sh = m == m_ ? 1 : 0;
    std::vector<std::vector<double>> B;
    for (int i = 0; i < 2 * m + 1; ++i) {
      std::vector<double> b(2 * m + 1);
      if (i < m - 1) {
        const auto& old = B_[i + sh];
        for (int j = 0; j <= i; ++j) b[j] = old[j + sh];
      } else if (i == m - 1) {
        for (int j = 0; j <= i; ++j) b[j] = incr_B[j];
      } else if (i < 2 * m - 1) {
        const auto& old = B_[i + (m == m_ ? 1 : -1)];
        for (int j = 0; j < m; ++j) b[j] = old[j + sh];
        b[m - 1] = incr_B[i];
        for (int j = m; j <= i; ++j) b[j] = old[j + (m == m_ ? 1 : -1)];
      } else if (i == 2 * m - 1) {
        for (int j = 0; j < 2 * m; ++j) b[j] = incr_B[2 * m + j];
      } else {
        for (int j = 0; j < 2 * m + 1; ++j) b[j] = incr_B[4 * m + j];
      }
      B.push_back(b);
    }
    for (int i = 0; i < 2 * m + 1; ++i)
      for (int j = 0; j < i; ++j) B[j][i] = B[i][j];
    B_ = B;
    m_ = m;
  }
  void CalcDirection(const std::vector<Vec>& s, const std::vector<Vec>& y, const Vec& grad, Vec* p) {
    p->assign(grad.size(), 0);
    std::vector<double> d(2 * m_ + 1, 0.0), alpha(m_, 0.0);
    d[2 * m_] = -1;
    for (int i = m_ - 1; i >= 0; --i) {
      for (int l = 0; l < 2 * m_ + 1; ++l) alpha[i] += d[l] * B_[l][i];
      alpha[i] /= B_[i][m_ + i] + 1e-10;
      d[m_ + i] -= alpha[i];
    }
    for (int i = 0; i < 2 * m_ + 1; ++i) d[i] *= B_[m_ - 1][2 * m_ - 1] / (B_[2 * m_ - 1][2 * m_ - 1] + 1e-10);
    for (int i = 0; i < m_; ++i) {
      double beta = 0;
      for (int l = 0; l < 2 * m_ + 1; ++l) beta += d[l] * B_[m_ + i][l];
      beta /= B_[i][m_ + i] + 1e-10;
      d[i] += alpha[i] - beta;
    }
    for (int i = 0; i < m_; ++i) Add(static_cast<real_t>(d[i]), s[i], p);
    for (int i = 0; i < m_; ++i) Add(static_cast<real_t>(d[i + m_]), y[i], p);
    Add(static_cast<real_t>(d[2 * m_]), grad, p);
  }

 private:
  int m_ = 0;
  std::vector<std::vector<double>> B_;
};

struct Param {
  int V_dim = 0, m = 10, max_num_epochs = 100, max_num_linesearchs = 5;
  real_t l2 = .1f, V_l2 = .01f, alpha = 1, init_alpha = 1, c1 = 1e-4f, c2 = .9f, rho = .5f;
};

/**
 * \brief minimise  sum_i log(1 + exp(-y_i f(x_i; w))) + 1/2 l2 |w|^2 + 1/2 V_l2 |V|^2  over one block of
 * localized rows; every key carries V when V_dim > 0 (V_threshold = 0).  Returns the objective after
 * every epoch — what LBFGSLearner's epoch-end callback reports.
 */
inline std::vector<real_t> Run(Loss* loss, const dmlc::RowBlock<unsigned>& data, size_t nkeys, const Param& P,
                               const std::function<void(const std::vector<int>&, Vec*)>& initializer = nullptr) {
  const int k = P.V_dim;
  const size_t n = nkeys * (1 + k);
  std::vector<int> lens(k ? nkeys : 0, 1 + k);
  Vec w(n, 0);
  if (initializer) initializer(lens, &w);
  SArray<int> w_pos, V_pos;
  if (k) {
    w_pos.resize(nkeys);
    V_pos.resize(nkeys);
    for (size_t i = 0; i < nkeys; ++i) {
      w_pos[i] = static_cast<int>(i * (1 + k));
      V_pos[i] = w_pos[i] + 1;
    }
  }
  auto reg_coef = [&](size_t i) { return (k && i % (1 + k) != 0) ? P.V_l2 : P.l2; };
  auto reg_eval = [&](const Vec& x) {  // LBFGSUpdater::Evaluate, lbfgs_updater.h:189-203
    real_t o = 0;
    for (size_t i = 0; i < x.size(); ++i) o += .5 * reg_coef(i) * x[i] * x[i];
    return o;
  };
  auto add_reg_grad = [&](const Vec& x, Vec* g) {  // AddRegularizerGrad, :170-184
    for (size_t i = 0; i < x.size(); ++i) (*g)[i] += reg_coef(i) * x[i];
  };
  Vec wg;
  auto loss_grad = [&](const Vec& x) {  // LBFGSLearner::CalcGrad, lbfgs_learner.cc:246-305
    SArray<real_t> ws(x.size()), pred(data.size), g(x.size());
    std::copy(x.begin(), x.end(), ws.data());
    std::vector<SArray<char>> param = {SArray<char>(ws), SArray<char>(w_pos), SArray<char>(V_pos)};
    loss->Predict(data, param, &pred);
    param.push_back(SArray<char>(pred));
    loss->CalcGrad(data, param, &g);
    wg.assign(g.data(), g.data() + g.size());
    return loss->Evaluate(data.label, pred);
  };
  real_t objv = reg_eval(w) + loss_grad(w);
  std::vector<Vec> s, y;
  Vec grads;
  real_t alpha_srv = 0;
  Twoloop tl;
  std::vector<real_t> out;
  for (int ep = 0; ep < P.max_num_epochs; ++ep) {
    Vec new_grads = wg;
    add_reg_grad(w, &new_grads);
    std::vector<real_t> B;
    if (grads.empty()) {  // PrepareCalcDirection, lbfgs_updater.h:86-101
      grads = new_grads;
    } else {
      if (static_cast<int>(y.size()) == P.m) y.erase(y.begin());
      y.push_back(new_grads);
      Add(-1, grads, &y.back());
      grads = new_grads;
      if (alpha_srv != 1) for (auto& v : s.back()) v *= alpha_srv;
      alpha_srv = 0;
      tl.CalcIncreB(s, y, grads, &B);
    }
    Vec dir;  // CalcDirection, :107-123
    if (!y.empty()) {
      tl.ApplyIncreB(B);
      tl.CalcDirection(s, y, grads, &dir);
    } else {
      dir = grads;
      for (auto& v : dir) v *= -1;
    }
    for (auto& v : dir) v = v > 5 ? 5 : (v < -5 ? -5 : v);
    if (static_cast<int>(s.size()) == P.m) s.erase(s.begin());
    s.push_back(dir);
    const real_t p_gf = static_cast<real_t>(Inner(grads, dir));
    real_t alpha = ep != 0 ? P.alpha : P.init_alpha, alpha_w = 0, new_objv = objv;  // lbfgs_learner.cc:52-73
    for (int i = 0; i < P.max_num_linesearchs; ++i) {
      Add(alpha - alpha_w, dir, &w);
      alpha_w = alpha;
      alpha_srv = alpha;
      real_t st0 = 0, st1 = 0;
      st0 += loss_grad(w);
      st1 += static_cast<real_t>(Inner(wg, dir));
      Vec rg(w.size(), 0);
      add_reg_grad(w, &rg);
      st0 += reg_eval(w);
      st1 += static_cast<real_t>(Inner(rg, dir));
      new_objv = st0;
      if (new_objv <= objv + P.c1 * alpha * p_gf && st1 >= P.c2 * p_gf) break;
      alpha *= P.rho;
    }
    out.push_back(new_objv);
    objv = new_objv;
  }
  return out;
}

}  // namespace lbfgs_mini
}  // namespace difacto
#endif  // DIFACTO_HOST_LBFGS_MINI_H_
