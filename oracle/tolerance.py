"""TEST INFRASTRUCTURE ONLY — tolerances for comparing two fp32 evaluations of FMLoss.

north_star: per-example logits and per-key gradients match the reference CPU path on identical
minibatches at fp32 rtol 1e-5.  Two correct fp32 evaluations that add the same n terms in a
different order (the reference: serially, src/common/spmv.h:119-132, spmm.h:105-118, 137-156; the
device: lane groups + wave reductions) can only agree to the rounding noise of the sums, which is
proportional to sum|terms|, not to |result| — a result that cancels to ~0 has no relative accuracy in
either of them.  So every comparison here is

    |got - ref|  <=  rtol * |ref|  +  floor,      floor = C_SIGMA * 2^-24 * sqrt(n) * sum|terms|

(2^-24 = fp32 unit roundoff; sqrt(n): random-walk growth of n roundings; C_SIGMA = 1 since round 4 — it was
4 through round 3, where the worst measured err / tol over all kernel-parity cases was 0.199), evaluated sum by sum — see predict_bound / calcgrad_bound.  This module computes sum|terms| and n in float64 from the very
inputs both sides were given.  Nothing here is imported by the product (difacto_amd/).
"""
import numpy as np
import scipy.sparse as sp

U32 = 2.0 ** -24
RTOL = 1e-5


class Design:
    """the localized minibatch as float64 CSR matrices [rows x unique keys]: X, |X| and X.^2, built
    entry by entry — a feature listed twice in a row stays two terms, as in SpMV / SpMM (the
    reference adds x_j w_j and x_j^2 V_j^2 per nonzero, spmv.h:119-132, fm_loss.h:95-110)"""

    def __init__(self, offset, index, value, ncols):
        offset = np.asarray(offset, np.int64)
        nnz = int(offset[-1])
        index = np.asarray(index, np.int64)[:nnz]
        x = np.ones(nnz, np.float64) if value is None else np.asarray(value[:nnz], np.float64)
        shape = (len(offset) - 1, ncols)
        self.X = sp.csr_matrix((x, index, offset), shape=shape)
        self.A = sp.csr_matrix((np.abs(x), index, offset), shape=shape)
        self.X2 = sp.csr_matrix((x * x, index, offset), shape=shape)
        self.s = np.maximum(np.diff(offset), 1).astype(np.float64)                    # terms per example
        self.cnt = np.maximum(np.bincount(index, minlength=ncols), 1).astype(np.float64)  # terms per key
        self.XT, self.AT, self.X2T = self.X.T.tocsr(), self.A.T.tocsr(), self.X2.T.tocsr()


def design(offset, index, value, ncols):
    return Design(offset, index, value, ncols)


def dense_rows(weights, lens, V_dim):
    """ragged (weights, lens) of SGDUpdater::Get (sgd_updater.cc:32-56) -> w[U], V[U,k], has_V[U] in float64;
    lens empty (V_dim == 0): one weight per key"""
    W = np.asarray(weights, np.float64)
    if V_dim == 0 or len(lens) == 0:
        return W.copy(), np.zeros((len(W), 0)), np.zeros(len(W), bool)
    lens = np.asarray(lens, np.int64)
    start = np.concatenate([[0], np.cumsum(lens)[:-1]])
    has = lens > 1
    V = np.zeros((len(lens), V_dim))
    if has.any():
        V[has] = W[(start[has] + 1)[:, None] + np.arange(V_dim)[None, :]]
    return W[start], V, has


def packed_rows(weights, lens, V_dim, stride):
    """the exchange layout of include/difacto_hip.h: [w, has_V, 0, 0 | V zero-padded] per key, float32"""
    w, V, has = dense_rows(weights, lens, V_dim)
    rows = np.zeros((len(w), stride), np.float32)
    rows[:, 0] = w
    rows[:, 1] = has
    if V_dim:
        rows[:, 4:4 + V_dim] = V
    return rows


def ragged_from_packed(rows, lens, V_dim):
    """inverse of packed_rows for gradient rows: -> the ragged layout FMLoss::CalcGrad writes"""
    if V_dim == 0 or len(lens) == 0:
        return rows[:, 0].copy()
    out = []
    for r, l in zip(rows, lens):
        out.append(r[:1])
        if l > 1:
            out.append(r[4:4 + V_dim])
    return np.concatenate(out).astype(np.float32)


C_SIGMA = 1.0  # floor = C_SIGMA * 2^-24 * (random-walk rounding of the sums below); 4.0 until round 3 (worst measured err/tol 0.199 -> 0.8)


def predict_bound(D, w, V):
    """float64 FMLoss::Predict (fm_loss.h:67-119, clamp included iff V_dim > 0) and the comparison
    floor per example: the rounding of  sum_j x w  (s terms), of every XV_d = sum_j x V_jd (s terms,
    entering squared: 2 |XV_d| dXV_d), of every XXVV_d (s positive terms) and of the final sum over
    the V_dim dimensions"""
    X, A, X2, s = D.X, D.A, D.X2, D.s
    pred = X @ w
    noise = np.sqrt(s) * (A @ np.abs(w))
    k = V.shape[1]
    if k:
        XV = X @ V
        XXVV = X2 @ (V * V)
        AV = A @ np.abs(V)
        pred = pred + 0.5 * ((XV * XV).sum(1) - XXVV.sum(1))
        noise = noise + 0.5 * (np.sqrt(s) * (2 * np.abs(XV) * AV + XXVV).sum(1) + np.sqrt(k) * (XV * XV + XXVV).sum(1))
        pred = np.clip(pred, -20.0, 20.0)
    return pred, C_SIGMA * U32 * noise


def calcgrad_bound(D, label, pred, w, V, has):
    """float64 FMLoss::CalcGrad (fm_loss.h:148-199) on the given logits -> gw[U], gV[U,k] and their
    comparison floors: the rounding of the per-key sums over the key's occurrences (cnt terms) plus,
    for gV, the rounding of the X*V both sides recompute (s terms each, weighted by |x p|)"""
    y = np.where(np.asarray(label) > 0, 1.0, -1.0)
    p = -y / (1.0 + np.exp(y * np.asarray(pred, np.float64)))
    ap = np.abs(p)
    X, A, XT, AT, X2T, s, cnt = D.X, D.A, D.XT, D.AT, D.X2T, D.s, D.cnt
    gw = XT @ p
    fl_w = C_SIGMA * U32 * np.sqrt(cnt) * (AT @ ap)
    k = V.shape[1]
    if not k:
        return gw, np.zeros((len(w), 0)), fl_w, np.zeros((len(w), 0))
    XV = X @ V
    AV = A @ np.abs(V)
    xxp = X2T @ p
    gV = XT @ (XV * p[:, None]) - V * xxp[:, None]
    noise = np.sqrt(cnt)[:, None] * (AT @ (np.abs(XV) * ap[:, None]) + np.abs(V) * (X2T @ ap)[:, None]) \
        + AT @ (AV * (ap * np.sqrt(s))[:, None])
    gV[~has] = 0
    return gw, gV, fl_w, C_SIGMA * U32 * noise


def check(got, ref, floor, what, rtol=RTOL):
    """assert |got - ref| <= rtol |ref| + floor elementwise; returns the worst ratio err / tol"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    floor = np.broadcast_to(np.asarray(floor, np.float64), ref.shape)
    assert got.shape == ref.shape, "%s: shape %r vs %r" % (what, got.shape, ref.shape)
    tol = rtol * np.abs(ref) + floor + 1e-30
    ratio = np.abs(got - ref) / tol
    if ratio.size and not np.all(ratio <= 1.0):
        i = int(np.argmax(ratio))
        raise AssertionError("%s: element %d got %r want %r, |diff| %.3g > tol %.3g (rtol part %.3g, floor %.3g); "
                             "%d of %d outside" % (what, i, got.flat[i], ref.flat[i], abs(got.flat[i] - ref.flat[i]),
                                                   tol.flat[i], rtol * abs(ref.flat[i]), floor.flat[i],
                                                   int((ratio > 1).sum()), ratio.size))
    return float(ratio.max()) if ratio.size else 0.0
