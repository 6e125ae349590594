#!/usr/bin/env python
"""Generates tests/golden/ref_fm.npz with the REFERENCE'S OWN arithmetic (oracle/_ref = /root/reference's
src/data/localizer.cc, src/sgd/sgd_updater.cc, src/loss/fm_loss.h compiled here by oracle/Makefile) — run in the build
container, where /root/reference exists; the fixture travels to the GPU box, the reference does not.

For every case: the worker loop of src/sgd/sgd_learner.cc:129-227 driven for STEPS minibatch passes over one batch on an
empty SGDUpdater (rand_r lazy InitV, i.e. the reference as shipped):

    Localizer::Compact                                           -> <c>_feaids, <c>_feacnt, <c>_index (u32)
    step t:  [t == 0] Store::Push(kFeaCount)                     (sgd_learner.cc:214-217)
             Store::Pull(kWeight)                                -> <c>_vals_<t>, <c>_lens_<t>   (sgd_updater.cc:32-56)
             FMLoss::Predict                                     -> <c>_pred_<t>                 (fm_loss.h:67-119)
             FMLoss::CalcGrad                                    -> <c>_grad_<t>                 (fm_loss.h:148-199)
             Store::Push(kGradient) -> SGDUpdater::Update        (sgd_updater.cc:58-148)
    after the last step: Pull                                    -> <c>_vals_final, <c>_lens_final

    <c>_{offset,rawindex,value,label}: the batch that went in (value absent: binary features); <c>_param: the
    SGDUpdaterParam fields as a JSON string.

Cases: the reference's rcv1 fixture with V_dim 8 (BASELINE config C2) and one ragged random batch with V_dim 0, 5, 64
(binary for V_dim 5).  tests/test_golden_fm.py compares the HIP path with these arrays directly (-m gpu) and the C
restatement with them on CPU.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bindings as ob  # noqa: E402
from conftest import load_libsvm, random_batch  # noqa: E402

STEPS = 3


def get_pos(lens):
    """SGDLearner::GetPos (sgd_learner.cc:113-127)"""
    if len(lens) == 0:
        return None, None
    ends = np.cumsum(lens)
    w_pos = (ends - lens).astype(np.int32)
    V_pos = np.where(lens > 1, w_pos + 1, -1).astype(np.int32)
    return w_pos, V_pos


def run_case(R, name, batch, V_dim, param, out):
    loc = R.localize(batch["offset"], batch["index"])
    keys = loc["feaids"]
    out[name + "_offset"] = np.asarray(batch["offset"], np.uint64)
    out[name + "_rawindex"] = np.asarray(batch["index"], np.uint64)
    if batch["value"] is not None:
        out[name + "_value"] = batch["value"]
    out[name + "_label"] = batch["label"]
    out[name + "_feaids"] = keys
    out[name + "_feacnt"] = loc["feacnt"]
    out[name + "_index"] = loc["index"]
    out[name + "_param"] = np.frombuffer(json.dumps(dict(V_dim=V_dim, **param)).encode(), np.uint8)
    st = R.store_create(V_dim=V_dim, **param)
    for t in range(STEPS):
        if t == 0:
            st.push(keys, ob.FEA_COUNT, loc["feacnt"])
        vals, lens = st.pull(keys)
        vals, lens = vals.copy(), lens.copy()
        w_pos, V_pos = get_pos(lens)
        pred, grad = R.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], batch["value"], batch["label"], vals, w_pos, V_pos)
        st.push(keys, ob.GRADIENT, grad, lens)
        out["%s_vals_%d" % (name, t)] = vals
        out["%s_lens_%d" % (name, t)] = lens
        out["%s_pred_%d" % (name, t)] = pred
        out["%s_grad_%d" % (name, t)] = grad
    vals, lens = st.pull(keys)
    out[name + "_vals_final"] = vals.copy()
    out[name + "_lens_final"] = lens.copy()
    return int(loc["U"]), int((lens > 1).sum()) if len(lens) else 0


def main():
    if not ob.have_ref():
        ob.build(ref=True)
    R = ob.Ref()
    out = {}
    off, idx, val, lab = load_libsvm(os.path.join(ROOT, "tests", "golden", "rcv1_100.libsvm"))
    rcv1 = dict(offset=off, index=idx, value=val, label=lab)
    # example/rcv1_sgd.conf-like hyper-parameters with V on from the first step (V_threshold below every count is not
    # possible for count-1 keys: threshold 0 gives every key with w != 0 its V, sgd_updater.cc:122-126)
    p_rcv1 = dict(l1=0.1, l2=0.0, lr=0.1, lr_beta=1.0, V_lr=0.05, V_lr_beta=1.0, V_l2=0.01, V_threshold=0, V_init_scale=0.1, seed=0)
    info = {"rcv1_k8": run_case(R, "rcv1_k8", rcv1, 8, p_rcv1, out)}
    rng = np.random.default_rng(20260926)
    p_rag = dict(l1=0.02, l2=0.01, lr=0.3, lr_beta=1.0, V_lr=0.05, V_lr_beta=1.0, V_l2=0.02, V_threshold=1, V_init_scale=0.2, seed=9)
    for k in (0, 5, 64):
        b = random_batch(rng, 300, 1500, 40, binary=(k == 5))
        info["ragged_k%d" % k] = run_case(R, "ragged_k%d" % k, b, k, p_rag, out)
    out["cases"] = np.frombuffer(json.dumps(sorted(info)).encode(), np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "ref_fm.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d bytes): %s" % (path, os.path.getsize(path), info))


if __name__ == "__main__":
    main()
