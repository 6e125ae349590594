"""TEST INFRASTRUCTURE ONLY — the reference's L-BFGS outer loop, restated minimally, to drive a
Loss implementation through the golden trajectories of tests/cpp/lbfgs_learner_test.cc
(LBFGSLearner.Basic: V_dim 0, 19 epochs to 1e-5; LBFGSLearner.WithV: FM with V_dim 5 to 1e-4).

What it follows (all under /root/reference/src/lbfgs/):
  lbfgs_learner.cc:14-126   RunScheduler: direction, backtracking line search on the Wolfe
                            conditions (c1, c2, rho, max 5 trials), objective bookkeeping
  lbfgs_learner.cc:246-305  CalcGrad: Loss::Predict -> Loss::CalcGrad -> Loss::Evaluate on the data
  lbfgs_updater.h:33-160    InitWeight, PrepareCalcDirection (s, y history of m pairs, regulariser
                            gradient), CalcDirection (clamp to +-5), LineSearch, Evaluate = r(w)
  lbfgs_twoloop.h:19-126    the vector-free two-loop recursion (CalcIncreB / ApplyIncreB / CalcDelta)
  lbfgs_utils.h:62-98       Inner (float products summed in double), Add, Times
The data, tile store, thread pool and parameter-server jobs of the learner are not restated: one
worker, one block of rows.  The loss is a pair of callables, so the same loop runs the CPU oracle
(tests/test_oracle_golden.py) and the device's FMLoss (tests/test_gpu_parity.py); the C++ twin in
difacto_amd/host/host_tests.cc drives HipFMLoss through the Loss interface itself.
"""
import numpy as np

f32 = np.float32

BASIC_OBJV = [34.603421, 12.655075, 5.224232, 2.713903, 1.290586, 0.645131, 0.317889, 0.156723, 0.075331, 0.032091, 0.018044,
              0.008562, 0.004336, 0.002132, 0.001051, 0.000506, 0.000227, 0.000119, 0.000059]   # lbfgs_learner_test.cc:9-28
WITHV_OBJV = [35.224265, 21.631514, 18.394319, 16.077692, 12.389012, 8.888516, 8.446880, 8.146090, 8.023501, 7.981967, 7.955119,
              7.937092, 7.922456, 7.880596, 7.861660, 7.838057, 7.807892, 7.784401, 7.756756]   # :88-111 (first 19 = max_num_epochs)


def withv_initializer(lens, w):
    """the weight initializer of LBFGSLearner.WithV (lbfgs_learner_test.cc:130-141)"""
    n = 0
    for l in lens:
        for i in range(l):
            if i > 0:
                w[n] = (i - (l - 1) / 2) * .01
            n += 1


def inner(a, b):
    return float(np.sum((a * b).astype(np.float32).astype(np.float64)))

class Twoloop:
    def __init__(self): self.m_ = 0; self.B_ = []
    def calc_incre_B(self, s, y, grad):
        m = len(s); out = [0.0]*(6*m+1)
        for i in range(m):
            out[i] = inner(s[-1], s[i]); out[i+m] = inner(s[-1], y[i]); out[i+2*m] = inner(y[-1], s[i])
            out[i+3*m] = inner(y[-1], y[i]); out[i+4*m] = inner(grad, s[i]); out[i+5*m] = inner(grad, y[i])
        out[6*m] = inner(grad, grad)
        return out
    def apply_incre_B(self, incr):
        m = (len(incr)-1)//6; m_ = self.m_
        assert m == m_+1 or m == m_
        same = (m == m_)
        B = []
        for i in range(2*m+1):
            b = [0.0]*(2*m+1)
            if i < m-1:
                old = self.B_[i+(1 if same else 0)]
                for j in range(i+1): b[j] = old[j+(1 if same else 0)]
            elif i == m-1:
                for j in range(i+1): b[j] = incr[j]
            elif i < 2*m-1:
                old = self.B_[i+(1 if same else -1)]
                for j in range(m): b[j] = old[j+(1 if same else 0)]
                b[m-1] = incr[i]
                for j in range(m, i+1): b[j] = old[j+(1 if same else -1)]
            elif i == 2*m-1:
                for j in range(2*m): b[j] = incr[2*m+j]
            else:
                for j in range(2*m+1): b[j] = incr[4*m+j]
            B.append(b)
        for i in range(2*m+1):
            for j in range(i): B[j][i] = B[i][j]
        self.B_ = B; self.m_ = m
    def calc_delta(self):
        m = self.m_; B = self.B_
        d = [0.0]*(2*m+1); d[2*m] = -1.0
        alpha = [0.0]*m
        for i in range(m-1, -1, -1):
            for l in range(2*m+1): alpha[i] += d[l]*B[l][i]
            alpha[i] /= B[i][m+i] + 1e-10
            d[m+i] -= alpha[i]
        for i in range(2*m+1): d[i] *= B[m-1][2*m-1] / (B[2*m-1][2*m-1] + 1e-10)
        for i in range(m):
            beta = 0.0
            for l in range(2*m+1): beta += d[l]*B[m+i][l]
            beta /= B[i][m+i] + 1e-10
            d[i] += alpha[i] - beta
        return d
    def calc_direction(self, s, y, grad):
        m = self.m_; d = self.calc_delta()
        p = np.zeros_like(grad)
        def add(x, a, p):
            x = f32(x)
            if x == 0: return p
            return (p + a) if x == 1 else (p + x*a).astype(f32)
        for i in range(m): p = add(d[i], s[i], p)
        for i in range(m): p = add(d[i+m], y[i], p)
        p = add(d[2*m], grad, p)
        return p.astype(f32)

def run(loss_grad, nkeys, V_dim, l2, V_l2, m, max_epochs, rho=.5, c1=1e-4, c2=.9, alpha_p=1.0, init_alpha=1.0, init=None):
    """loss_grad(w, lens) -> (sum_i log(1 + exp(-y_i pred_i)) as float32, gradient in the layout of w);
    w is the ragged weight array of `nkeys` keys, every key with V when V_dim > 0 (V_threshold = 0).
    Returns the objective after every epoch (what the reference's epoch-end callback sees)."""
    U = nkeys
    lens = np.full(U, 1 + V_dim, np.int32) if V_dim else np.zeros(0, np.int32)
    n = U * (1 + V_dim)
    w = np.zeros(n, f32)
    if init:
        init(lens, w)
    regw = np.full(n, l2, f32)
    if V_dim:
        regw = np.full((U, 1 + V_dim), V_l2, f32)
        regw[:, 0] = l2
        regw = regw.reshape(-1)

    def reg_eval(w):  # LBFGSUpdater::Evaluate: a float accumulator, every term computed in double
        acc = f32(0)
        for x, r in zip(w, regw):
            acc = f32(float(acc) + .5 * float(r) * float(x) * float(x))
        return acc

    def reg_grad(w):
        return (regw * w).astype(f32)

    lobj, wg = loss_grad(w, lens)
    objv = f32(reg_eval(w) + f32(lobj))
    s, y = [], []
    grads = None
    alpha_srv = f32(0)
    tl = Twoloop()
    out = []
    for k in range(max_epochs):
        new_grads = (wg + reg_grad(w)).astype(f32)
        B = None
        if grads is None:
            grads = new_grads
        else:
            if len(y) == m:
                y.pop(0)
            y.append((new_grads - grads).astype(f32))
            grads = new_grads
            if alpha_srv != 1:
                s[-1] = (s[-1] * alpha_srv).astype(f32)
            alpha_srv = f32(0)
            B = tl.calc_incre_B(s, y, grads)
        if y:
            tl.apply_incre_B(B)
            d = tl.calc_direction(s, y, grads)
        else:
            d = (-grads).astype(f32)
        d = np.clip(d, -5, 5).astype(f32)
        if len(s) == m:
            s.pop(0)
        s.append(d)
        p_gf = f32(inner(grads, d))
        alpha = f32(alpha_p if k != 0 else init_alpha)
        alpha_w = f32(0)
        new_objv = objv
        for _ in range(5):
            step = f32(alpha - alpha_w)
            w = (w + d).astype(f32) if step == 1 else (w + step * d).astype(f32)
            alpha_w = alpha
            alpha_srv = alpha
            lobj, wg = loss_grad(w, lens)
            new_objv = f32(f32(lobj) + reg_eval(w))
            st1 = f32(f32(inner(wg, d)) + f32(inner(reg_grad(w), d)))
            if new_objv <= objv + c1 * alpha * p_gf and st1 >= c2 * p_gf:
                break
            alpha = f32(alpha * rho)
        out.append(float(new_objv))
        objv = new_objv
    return out
