"""Pins the CPU oracle (oracle/difacto_oracle.c) and, where present, the
reference build (oracle/_ref) to the reference's OWN golden vectors:

  tests/cpp/fm_loss_test.cc:12-83       FMLoss.NoV / FMLoss.HasV
  tests/cpp/localizer_test.cc:12-63     Localizer.Base / BaseHash / ReverseBytes
  tests/cpp/sgd_learner_test.cc:9-49    SGDLearner.Basic (20-epoch FTRL trajectory)

and the oracle to the reference itself on random inputs.
"""
import os

import numpy as np
import pytest

from conftest import random_batch
from oracle import bindings as ob

U64MAX = 2 ** 64 - 1


def _localized(chk, rcv1, max_index=U64MAX):
    loc = chk.localize(rcv1["offset"], rcv1["index"], max_index)
    ids = chk.reverse_bytes(loc["feaids"])
    return loc, ids.astype(np.int64)


@pytest.fixture(params=["oracle", "ref"])
def chk(request, oracle):
    if request.param == "oracle":
        return oracle
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    return ob.Ref()


# ---- tests/cpp/localizer_test.cc:12-29 (Base) and :31-49 (BaseHash)
def test_localizer_base(chk, rcv1):
    loc, ids = _localized(chk, rcv1)
    assert loc["U"] == 2775
    assert int(ids.sum()) == 65111856
    assert float(loc["feacnt"].sum()) == 9648
    assert np.array_equal(loc["offset"], rcv1["offset"])
    # ascending reversed keys, compact index consistent with the dictionary
    assert np.all(loc["feaids"][1:] > loc["feaids"][:-1])
    back = chk.reverse_bytes(loc["feaids"])[loc["index"]]
    assert np.array_equal(back, rcv1["index"])


def test_localizer_base_hash(chk, rcv1):
    loc, ids = _localized(chk, rcv1, 1000)
    assert int(ids.sum()) == 478817
    assert float(loc["feacnt"].sum()) == 9648


# ---- tests/cpp/localizer_test.cc:51-63
def test_reverse_bytes_involution(chk):
    n = 20000  # reference uses 1e6 samples of the same lattice
    for i in range(0, 1000000, 1000000 // n):
        j = (U64MAX // 1000000) * i
        assert chk.reverse_bytes(chk.reverse_bytes(j)) == j
    # nibble reversal, spot values
    assert chk.reverse_bytes(0x0123456789ABCDEF) == 0xFEDCBA9876543210
    assert chk.reverse_bytes(1) == 0x1000000000000000


def test_encode_fea_grp_id(chk):
    # include/difacto/base.h:60-63
    assert chk.encode_fea_grp_id(0xABCDEF, 7, 12) == (0xABCDEF << 12) | 7
    assert chk.encode_fea_grp_id(U64MAX, 38, 12) == ((U64MAX << 12) & U64MAX) | 38


# ---- tests/cpp/fm_loss_test.cc:12-40
def test_fm_loss_nov(chk, rcv1):
    loc, ids = _localized(chk, rcv1)
    w = (ids / 5e4).astype(np.float32)
    if chk.name == "ref":
        pred, grad = chk.fm_predict_calcgrad(0, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], w)
        objv = chk.logit_objv(rcv1["label"], pred)
    else:
        pred = chk.fm_predict(0, loc["offset"], loc["index"], rcv1["value"], w)
        grad = chk.fm_calcgrad(0, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], w, pred)
        objv = chk.loss_evaluate(rcv1["label"], pred)
    assert abs(objv - 147.4672) < 1e-3
    assert abs(float((grad.astype(np.float64) ** 2).sum()) - 90.5817) < 1e-3


def _hasv_weights(ids, k):
    U = len(ids)
    W = np.zeros((U, k + 1), np.float32)
    W[:, 0] = (ids / 5e4).astype(np.float32)
    for j in range(1, k + 1):
        W[:, j] = (ids * j / 5e5).astype(np.float32)
    w_pos = (np.arange(U) * (k + 1)).astype(np.int32)
    return W.reshape(-1).copy(), w_pos, w_pos + 1


# ---- tests/cpp/fm_loss_test.cc:42-83
def test_fm_loss_hasv(chk, rcv1):
    loc, ids = _localized(chk, rcv1)
    k = 5
    W, w_pos, V_pos = _hasv_weights(ids, k)
    if chk.name == "ref":
        pred, grad = chk.fm_predict_calcgrad(k, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], W, w_pos, V_pos)
        objv = chk.logit_objv(rcv1["label"], pred)
    else:
        pred = chk.fm_predict(k, loc["offset"], loc["index"], rcv1["value"], W, w_pos, V_pos)
        grad = chk.fm_calcgrad(k, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], W, pred, w_pos, V_pos)
        objv = chk.loss_evaluate(rcv1["label"], pred)
    assert abs(objv - 330.628) < 1e-3
    assert abs(float((grad.astype(np.float64) ** 2).sum()) - 1.2378e3) < 1e-1


SGD_BASIC = [69.314718, 69.314718, 67.151912, 61.414778, 56.244989, 53.218700, 51.248737, 49.846688,
             48.650164, 47.698351, 46.924038, 46.388223, 45.970721, 45.499307, 45.102245, 44.798413,
             44.565211, 44.386417, 44.240657, 44.109764]


# ---- tests/cpp/sgd_learner_test.cc:9-49: V_dim=0,l1=l2=lr=1,batch=100, 1 job/epoch, 20 epochs.
# Epoch loop restated from SGDLearner::IterateData (sgd_learner.cc:196-224): one
# batch of 100 rows per epoch; epoch 0 also pushes the feature counts.
def test_sgd_learner_basic_oracle(oracle, rcv1):
    loc = oracle.localize(rcv1["offset"], rcv1["index"])
    st = oracle.store_create(V_dim=0, l1=1, l2=1, lr=1)
    for epoch in range(20):
        prog = ob.Progress()
        st.sgd_step(loc["offset"], loc["index"], rcv1["value"], rcv1["label"], loc["feaids"],
                    feacnt=loc["feacnt"] if epoch == 0 else None, is_train=True, prog=prog)
        assert abs(prog.loss - SGD_BASIC[epoch]) < 5e-5, (epoch, prog.loss)
        assert prog.nrows == 100


def test_sgd_learner_basic_ref(ref, oracle, rcv1):
    """same trajectory driven through the reference's StoreLocal/SGDUpdater/FMLoss"""
    loc = ref.localize(rcv1["offset"], rcv1["index"])
    st = ref.store_create(V_dim=0, l1=1, l2=1, lr=1)
    for epoch in range(20):
        if epoch == 0:
            st.push(loc["feaids"], ob.FEA_COUNT, loc["feacnt"])
        vals, lens = st.pull(loc["feaids"])
        assert len(lens) == 0  # sgd_updater.cc:40
        pred, grad = ref.fm_predict_calcgrad(0, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], vals)
        loss = ref.loss_evaluate(rcv1["label"], pred)
        assert abs(loss - SGD_BASIC[epoch]) < 5e-5, (epoch, loss)
        st.push(loc["feaids"], ob.GRADIENT, grad)


# ---- oracle restatement vs the reference itself on random inputs
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("binary", [False, True])
def test_oracle_vs_ref_localize(oracle, ref, seed, binary):
    rng = np.random.default_rng(seed)
    b = random_batch(rng, 200, 5000 if seed else 2 ** 63, 30, binary=binary)
    for mx in (U64MAX, 1000):
        a = oracle.localize(b["offset"], b["index"], mx)
        r = ref.localize(b["offset"], b["index"], mx)
        assert a["U"] == r["U"]
        for k in ("feaids", "feacnt", "index", "offset"):
            assert np.array_equal(a[k], r[k]), k


@pytest.mark.parametrize("V_dim", [0, 1, 5, 8, 64])
@pytest.mark.parametrize("binary", [False, True])
def test_oracle_vs_ref_fm_loss(oracle, ref, V_dim, binary):
    rng = np.random.default_rng(100 + V_dim)
    b = random_batch(rng, 64, 300, 40, binary=binary)
    loc = oracle.localize(b["offset"], b["index"])
    U = loc["U"]
    # ragged weights: ~30% of keys without V (lens 1), the rest 1+V_dim
    if V_dim == 0:
        W = rng.normal(size=U).astype(np.float32) * 0.1
        w_pos = V_pos = None
    else:
        lens = np.where(rng.random(U) < 0.3, 1, 1 + V_dim).astype(np.int32)
        w_pos, V_pos = oracle.get_pos(lens)
        W = (rng.normal(size=int(lens.sum())) * 0.1).astype(np.float32)
    pr, gr = ref.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], W, w_pos, V_pos)
    po = oracle.fm_predict(V_dim, loc["offset"], loc["index"], b["value"], W, w_pos, V_pos)
    go = oracle.fm_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], W, po, w_pos, V_pos)
    # same evaluation order => expect equality up to FMA/vectorisation noise
    np.testing.assert_allclose(po, pr, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(go, gr, rtol=1e-5, atol=1e-6)
    assert abs(oracle.loss_evaluate(b["label"], po) - ref.loss_evaluate(b["label"], pr)) < 1e-3
    # AUC on identical inputs (ties in pred are ordered arbitrarily by std::sort, so
    # compare on a tie-free vector)
    pt = np.unique(pr)
    lt = b["label"][:len(pt)]
    assert oracle.auc_times_n(lt, pt) == pytest.approx(ref.auc_times_n(lt, pt), rel=1e-6)


@pytest.mark.parametrize("V_dim", [0, 4, 16])
def test_oracle_vs_ref_updater_trajectory(oracle, ref, V_dim):
    """FTRL(w) + AdaGrad(V) + lazy InitV (rand_r chain) over several steps of
    several batches, incl. the epoch-0 feature-count pushes: entry-for-entry
    equality of what Pull returns.  This is the V_dim>0 SGD coverage the
    reference's own tests lack (SURVEY.md 8c 'Gap')."""
    rng = np.random.default_rng(7 + V_dim)
    kw = dict(V_dim=V_dim, l1=0.05, l2=0.01, lr=0.2, V_lr=0.1, V_l2=0.02, V_threshold=2, V_init_scale=0.3, seed=3)
    so = oracle.store_create(init_mode=ob.INIT_REFRAND, **kw)
    sr = ref.store_create(**kw)
    batches = [random_batch(rng, 50, 120, 12, binary=(i % 2 == 0)) for i in range(4)]
    locs = [oracle.localize(b["offset"], b["index"]) for b in batches]
    for epoch in range(4):
        for b, loc in zip(batches, locs):
            if epoch == 0:
                so.push(loc["feaids"], ob.FEA_COUNT, loc["feacnt"])
                sr.push(loc["feaids"], ob.FEA_COUNT, loc["feacnt"])
            vo, lo = so.pull(loc["feaids"])
            vr, lr_ = sr.pull(loc["feaids"])
            assert np.array_equal(lo, lr_)
            np.testing.assert_allclose(vo, vr, rtol=2e-5, atol=1e-7)
            if V_dim:
                w_pos, V_pos = oracle.get_pos(lo)
            else:
                w_pos = V_pos = None
            pred, grad = ref.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], vr, w_pos, V_pos)
            so.push(loc["feaids"], ob.GRADIENT, grad, lo)
            sr.push(loc["feaids"], ob.GRADIENT, grad, lr_)
    if V_dim:
        assert np.any(lo > 1), "trajectory never allocated V; test is vacuous"


def test_rand_r_matches_libc(oracle):
    import ctypes
    libc = ctypes.CDLL(None)
    libc.rand_r.restype = ctypes.c_int
    for s0 in (0, 1, 12345, 2 ** 31, 2 ** 32 - 1):
        a, b = ctypes.c_uint(s0), ctypes.c_uint(s0)
        for _ in range(50):
            assert oracle.L.orc_rand_r(ctypes.byref(a)) == libc.rand_r(ctypes.byref(b))
            assert a.value == b.value


def test_penalty_and_getpos(oracle):
    lens = np.array([1, 3, 0, 3, 1], np.int32)
    w_pos, V_pos = oracle.get_pos(lens)
    assert w_pos.tolist() == [0, 1, -1, 4, 7]
    assert V_pos.tolist() == [-1, 2, -1, 5, -1]
    p = ob.make_param(V_dim=2, l1=0.5, l2=2.0, V_l2=4.0)
    W = np.array([1, -2, 3, 4, 0.5, 1, 1, -1], np.float32)
    got = oracle.evaluate_penalty(p, W, w_pos, V_pos)
    ws = [1, -2, 0.5, -1]
    exp = sum(0.5 * abs(w) + 0.5 * 2 * w * w for w in ws) + 0.5 * 4 * (9 + 16 + 1 + 1)
    assert got == pytest.approx(exp, rel=1e-6)


# ---- tests/cpp/lbfgs_learner_test.cc: the FM loss with V driven by L-BFGS ("indirectly ... FM with V
# over 23 iterations"): pins fm_predict / fm_calcgrad / loss_evaluate with V_dim > 0 against the reference's numbers
@pytest.mark.parametrize("impl", ["oracle", "ref"])
@pytest.mark.parametrize("case", ["basic", "withv"])
def test_golden_lbfgs_trajectories(request, rcv1, impl, case):
    from oracle import lbfgs_driver as LB
    P = request.getfixturevalue(impl)
    oracle = request.getfixturevalue("oracle")
    loc = oracle.localize(rcv1["offset"], rcv1["index"])
    V_dim = 0 if case == "basic" else 5

    def loss_grad(w, lens):
        wp, vp = oracle.get_pos(lens) if V_dim else (None, None)
        if impl == "ref":
            pred, g = P.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], w, wp, vp)
        else:
            pred = P.fm_predict(V_dim, loc["offset"], loc["index"], rcv1["value"], w, wp, vp)
            g = P.fm_calcgrad(V_dim, loc["offset"], loc["index"], rcv1["value"], rcv1["label"], w, pred, wp, vp)
        return np.float32(P.loss_evaluate(rcv1["label"], pred)), g

    if case == "basic":
        got = LB.run(loss_grad, loc["U"], 0, 0.0, 0.01, 5, 19)
        assert np.max(np.abs(np.array(got) - np.array(LB.BASIC_OBJV))) < 1e-5
    else:
        got = LB.run(loss_grad, loc["U"], 5, 0.1, 0.01, 5, 19, init=LB.withv_initializer)
        assert np.max(np.abs(np.array(got) - np.array(LB.WITHV_OBJV))) < 1e-4


def test_reference_build_recipe_from_scratch(tmp_path):
    """oracle/Makefile compiles the reference's own sources against third_party_shim from scratch (a copy of oracle/ in a
    scratch directory: the in-tree oracle/_ref is up to date most of the time, which once hid a shim change the
    reference's headers did not compile against)"""
    import shutil
    import subprocess
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference absent: oracle/_ref travels prebuilt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = tmp_path / "oracle"
    shutil.copytree(os.path.join(root, "oracle"), work, ignore=shutil.ignore_patterns("_ref", "*.so", "__pycache__"))
    subprocess.check_call(["make", "-s", "-C", str(work), "SHIM=" + os.path.join(root, "third_party_shim"), "ref"])
    assert os.path.getsize(work / "_ref" / "libdifacto_ref.so") > 100000
