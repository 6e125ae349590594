"""ctypes bindings of the two CPU checkers.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

* ``Oracle``  -> oracle/liboracle.so : the plain-C restatement (difacto_oracle.c)
* ``Ref``     -> oracle/_ref/libdifacto_ref.so : the reference's own sources
                 compiled here against the shims (ref_capi.cc); may be absent.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Both classes expose the same numpy-level methods so tests can be
parametrised over them.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libdifacto_ref.so")

FEA_COUNT, WEIGHT, GRADIENT = 1, 2, 3
INIT_REFRAND, INIT_HASH = 0, 1

UPDATER_DEFAULTS = dict(l1=1.0, l2=0.0, V_l2=0.01, lr=0.01, lr_beta=1.0, V_lr=0.01,
                        V_lr_beta=1.0, V_init_scale=0.01, V_dim=0, V_threshold=10, seed=0)


def build(ref=True):
    """(re)build liboracle.so and, if /root/reference is present, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def have_ref():
    return os.path.exists(REF_SO)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _sz(a):
    return np.ascontiguousarray(a, dtype=np.uint64)  # size_t on LP64


class UpdaterParam(C.Structure):
    _fields_ = [("l1", C.c_float), ("l2", C.c_float), ("V_l2", C.c_float), ("lr", C.c_float),
                ("lr_beta", C.c_float), ("V_lr", C.c_float), ("V_lr_beta", C.c_float),
                ("V_init_scale", C.c_float), ("V_dim", C.c_int), ("V_threshold", C.c_int),
                ("seed", C.c_uint), ("init_mode", C.c_int)]


class Progress(C.Structure):
    _fields_ = [("loss", C.c_float), ("penalty", C.c_float), ("auc", C.c_float),
                ("nnz_w", C.c_float), ("nrows", C.c_float)]


def make_param(init_mode=INIT_REFRAND, **kw):
    d = dict(UPDATER_DEFAULTS)
    d.update(kw)
    p = UpdaterParam()
    for k, v in d.items():
        setattr(p, k, v)
    p.init_mode = init_mode
    return p


class Oracle:
    """numpy-level view of oracle/liboracle.so"""
    name = "oracle"

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = self.L = C.CDLL(ORACLE_SO)
        u64, sz, vp, i32, f32 = C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_float
        L.orc_reverse_bytes.restype = u64
        L.orc_reverse_bytes.argtypes = [u64]
        L.orc_encode_fea_grp_id.restype = u64
        L.orc_encode_fea_grp_id.argtypes = [u64, i32, i32]
        L.orc_rand_r.restype = i32
        L.orc_rand_r.argtypes = [C.POINTER(C.c_uint)]
        L.orc_hash_init_value.restype = f32
        L.orc_hash_init_value.argtypes = [u64, i32, C.c_uint, f32]
        L.orc_splitmix64.restype = u64
        L.orc_splitmix64.argtypes = [u64]
        L.orc_localize.restype = sz
        L.orc_localize.argtypes = [sz, vp, vp, u64, vp, vp, vp, vp, vp]
        L.orc_store_create.restype = vp
        L.orc_store_create.argtypes = [C.POINTER(UpdaterParam)]
        L.orc_store_destroy.argtypes = [vp]
        L.orc_store_size.restype = sz
        L.orc_store_size.argtypes = [vp]
        L.orc_store_pull.argtypes = [vp, vp, sz, vp, C.POINTER(sz), vp, C.POINTER(sz)]
        L.orc_store_push.restype = i32
        L.orc_store_push.argtypes = [vp, vp, sz, i32, vp, sz, vp, sz]
        L.orc_store_peek.restype = i32
        L.orc_store_peek.argtypes = [vp, u64, vp, vp, C.POINTER(i32)]
        L.orc_store_poke.argtypes = [vp, u64, vp, vp, i32]
        L.orc_get_pos.argtypes = [vp, sz, vp, vp]
        L.orc_fm_predict.argtypes = [i32, sz, vp, vp, vp, vp, vp, vp, sz, vp, vp]
        L.orc_fm_calcgrad.argtypes = [i32, sz, vp, vp, vp, vp, vp, sz, vp, vp, sz, vp, vp]
        L.orc_loss_evaluate.restype = f32
        L.orc_loss_evaluate.argtypes = [vp, vp, sz]
        L.orc_auc_times_n.restype = f32
        L.orc_auc_times_n.argtypes = [vp, vp, sz]
        L.orc_evaluate_penalty.restype = f32
        L.orc_evaluate_penalty.argtypes = [C.POINTER(UpdaterParam), vp, sz, vp, vp, sz]
        L.orc_sgd_step.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, vp, i32, C.POINTER(Progress), vp]

    # --- a1
    def reverse_bytes(self, x):
        if np.isscalar(x) or isinstance(x, int):
            return int(self.L.orc_reverse_bytes(int(x)))
        return np.array([self.L.orc_reverse_bytes(int(v)) for v in np.asarray(x)], dtype=np.uint64)

    def encode_fea_grp_id(self, x, gid, nbits):
        return int(self.L.orc_encode_fea_grp_id(int(x), gid, nbits))

    # --- a2
    def localize(self, offset, index, max_index=2 ** 64 - 1, want_cnt=True, want_sorted=False):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, dtype=np.uint64)
        n = len(offset) - 1
        nnz = int(offset[-1]) if n > 0 else 0
        uniq = np.zeros(max(nnz, 1), np.uint64)
        cnt = np.zeros(max(nnz, 1), np.float32) if want_cnt else None
        oi = np.zeros(max(nnz, 1), np.uint32)
        oo = np.zeros(n + 1, np.uint64)
        sp = np.zeros(max(nnz, 1), np.uint32) if want_sorted else None
        U = self.L.orc_localize(n, _p(offset), _p(index), max_index, _p(uniq), _p(cnt), _p(oi), _p(oo), _p(sp))
        out = dict(U=U, feaids=uniq[:U].copy(), index=oi[:nnz].copy(), offset=oo)
        if want_cnt:
            out["feacnt"] = cnt[:U].copy()
        if want_sorted:
            out["sorted_pos"] = sp[:nnz].copy()
        return out

    # --- store
    def store_create(self, init_mode=INIT_REFRAND, **kw):
        p = make_param(init_mode, **kw)
        h = self.L.orc_store_create(C.byref(p))
        return OracleStore(self, h, p)

    # --- loss
    def get_pos(self, lens):
        lens = np.ascontiguousarray(lens, np.int32)
        w = np.zeros(len(lens), np.int32)
        v = np.zeros(len(lens), np.int32)
        self.L.orc_get_pos(_p(lens), len(lens), _p(w), _p(v))
        return w, v

    def fm_predict(self, V_dim, offset, index, value, weights, w_pos=None, V_pos=None, want_xv=False):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, np.uint32)
        value = _f32(value)
        weights = _f32(weights)
        n = len(offset) - 1
        pred = np.zeros(n, np.float32)
        xv = np.zeros(max(n * V_dim, 1), np.float32) if want_xv else None
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        self.L.orc_fm_predict(V_dim, n, _p(offset), _p(index), _p(value), _p(weights), _p(w_pos), _p(V_pos),
                              npos, _p(pred), _p(xv))
        return (pred, xv.reshape(n, V_dim)) if want_xv else pred

    def fm_calcgrad(self, V_dim, offset, index, value, label, weights, pred, w_pos=None, V_pos=None):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, np.uint32)
        value = _f32(value)
        weights = _f32(weights)
        label = _f32(label)
        pred = _f32(pred)
        n = len(offset) - 1
        grad = np.zeros(len(weights), np.float32)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        self.L.orc_fm_calcgrad(V_dim, n, _p(offset), _p(index), _p(value), _p(label), _p(weights), len(weights),
                               _p(w_pos), _p(V_pos), npos, _p(pred), _p(grad))
        return grad

    def loss_evaluate(self, label, pred):
        label, pred = _f32(label), _f32(pred)
        return float(self.L.orc_loss_evaluate(_p(label), _p(pred), len(pred)))

    def auc_times_n(self, label, pred):
        label, pred = _f32(label), _f32(pred)
        return float(self.L.orc_auc_times_n(_p(label), _p(pred), len(pred)))

    def evaluate_penalty(self, param, weights, w_pos=None, V_pos=None):
        weights = _f32(weights)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        return float(self.L.orc_evaluate_penalty(C.byref(param), _p(weights), len(weights), _p(w_pos), _p(V_pos), npos))


class OracleStore:
    def __init__(self, orc, handle, param):
        self.o, self.h, self.param = orc, handle, param
        self.V_dim = param.V_dim

    def __del__(self):
        if getattr(self, "h", None):
            self.o.L.orc_store_destroy(self.h)
            self.h = None

    def size(self):
        return int(self.o.L.orc_store_size(self.h))

    def pull(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        n = len(keys)
        vals = np.zeros(max(n * (1 + self.V_dim), 1), np.float32)
        lens = np.zeros(max(n, 1), np.int32)
        nv, nl = C.c_size_t(0), C.c_size_t(0)
        self.o.L.orc_store_pull(self.h, _p(keys), n, _p(vals), C.byref(nv), _p(lens), C.byref(nl))
        return vals[:nv.value].copy(), lens[:nl.value].copy()

    def push(self, keys, val_type, vals, lens=None):
        keys = np.ascontiguousarray(keys, np.uint64)
        vals = _f32(vals)
        lens = np.zeros(0, np.int32) if lens is None else np.ascontiguousarray(lens, np.int32)
        rc = self.o.L.orc_store_push(self.h, _p(keys), len(keys), val_type, _p(vals), len(vals), _p(lens), len(lens))
        if rc != 0:
            raise RuntimeError("oracle push: reference CHECK would fail")

    def peek(self, key):
        s4 = np.zeros(4, np.float32)
        v = np.zeros(max(2 * self.V_dim, 1), np.float32)
        hv = C.c_int(0)
        ok = self.o.L.orc_store_peek(self.h, int(key), _p(s4), _p(v), C.byref(hv))
        if not ok:
            return None
        return dict(fea_cnt=s4[0], w=s4[1], sqrt_g=s4[2], z=s4[3], V=v[:2 * self.V_dim] if hv.value else None)

    def poke(self, key, fea_cnt, w, sqrt_g, z, V2k=None):
        s4 = np.array([fea_cnt, w, sqrt_g, z], np.float32)
        v = None if V2k is None else _f32(V2k)
        self.o.L.orc_store_poke(self.h, int(key), _p(s4), _p(v), 0 if V2k is None else 1)

    def sgd_step(self, offset, index, value, label, feaids, feacnt=None, is_train=True, prog=None):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, np.uint32)
        value = _f32(value)
        label = _f32(label)
        feaids = np.ascontiguousarray(feaids, np.uint64)
        feacnt = _f32(feacnt)
        n = len(offset) - 1
        pred = np.zeros(n, np.float32)
        prog = prog if prog is not None else Progress()
        self.o.L.orc_sgd_step(self.h, n, _p(offset), _p(index), _p(value), _p(label), _p(feaids), len(feaids),
                              _p(feacnt), 1 if is_train else 0, C.byref(prog), _p(pred))
        return pred, prog


class Ref:
    """numpy-level view of oracle/_ref/libdifacto_ref.so (the reference itself)"""
    name = "ref"

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        L = self.L = C.CDLL(REF_SO)
        u64, sz, vp, i32, f32 = C.c_uint64, C.c_size_t, C.c_void_p, C.c_int, C.c_float
        L.ref_reverse_bytes.restype = u64
        L.ref_reverse_bytes.argtypes = [u64]
        L.ref_encode_fea_grp_id.restype = u64
        L.ref_encode_fea_grp_id.argtypes = [u64, i32, i32]
        L.ref_localize.restype = sz
        L.ref_localize.argtypes = [sz, vp, vp, vp, u64, i32, vp, vp, vp, vp, vp]
        L.ref_store_create.restype = vp
        L.ref_store_create.argtypes = [C.c_char_p]
        L.ref_store_destroy.argtypes = [vp]
        L.ref_store_pull.argtypes = [vp, vp, sz, vp, C.POINTER(sz), vp, C.POINTER(sz)]
        L.ref_store_push.argtypes = [vp, vp, sz, i32, vp, sz, vp, sz]
        L.ref_fmloss_create.restype = vp
        L.ref_fmloss_create.argtypes = [i32, i32]
        L.ref_fmloss_destroy.argtypes = [vp]
        L.ref_fmloss_predict.argtypes = [vp, sz, vp, vp, vp, vp, sz, vp, vp, sz, vp]
        L.ref_fmloss_calcgrad.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, vp, vp, sz, vp, vp]
        L.ref_loss_evaluate.restype = f32
        L.ref_loss_evaluate.argtypes = [vp, vp, vp, sz]
        L.ref_auc.restype = f32
        L.ref_auc.argtypes = [vp, vp, sz]
        L.ref_logit_objv.restype = f32
        L.ref_logit_objv.argtypes = [vp, vp, sz]
        # data-format half (present when the build found an lz4.h: oracle/Makefile)
        self.has_ingest = hasattr(L, "ref_crb_compress")
        if self.has_ingest:
            L.ref_crb_compress.restype = C.c_long
            L.ref_crb_compress.argtypes = [sz, vp, vp, vp, vp, vp, vp, sz]
            L.ref_crb_decompress.restype = C.c_long
            L.ref_crb_decompress.argtypes = [vp, sz, sz, sz, vp, vp, vp, vp, vp, vp]
            L.ref_criteo_parse.restype = C.c_long
            L.ref_criteo_parse.argtypes = [C.c_char_p, sz, i32, sz, sz, vp, vp, vp, vp]
            if hasattr(L, "ref_adfea_parse"):
                L.ref_adfea_parse.restype = C.c_long
                L.ref_adfea_parse.argtypes = [C.c_char_p, sz, sz, sz, vp, vp, vp, vp]
            L.ref_city_checker_hash64.restype = u64
            L.ref_city_checker_hash64.argtypes = [C.c_char_p, sz]
        self._loss = {}

    # ---- the reference's on-disk formats (src/data/compressed_row_block.h, src/reader/criteo_parser.h)
    def crb_compress(self, offset, label, index, value=None, weight=None):
        """CompressedRowBlock::Compress<feaid_t> -> the record's bytes"""
        offset = _sz(offset)
        label = _f32(label)
        index = np.ascontiguousarray(index, np.uint64)
        value, weight = _f32(value), _f32(weight)
        n = len(offset) - 1
        nnz = int(offset[-1] - offset[0])
        cap = 64 + 2 * (8 * (n + 1) + 12 * max(nnz, 1) + 8 * max(n, 1)) + 4096
        out = np.zeros(cap, np.uint8)
        got = self.L.ref_crb_compress(n, _p(offset), _p(label), _p(index), _p(value), _p(weight), _p(out), cap)
        assert got >= 0
        return out[:got].tobytes()

    def crb_decompress(self, rec, row_cap=1 << 16, nnz_cap=1 << 22):
        """CompressedRowBlock::Decompress<feaid_t> -> dict(offset, label, index, value | None, weight | None)"""
        rec = bytes(rec)
        buf = np.frombuffer(rec, np.uint8)
        off = np.zeros(row_cap + 1, np.uint64)
        lab = np.zeros(row_cap, np.float32)
        idx = np.zeros(nnz_cap, np.uint64)
        val = np.zeros(nnz_cap, np.float32)
        wgt = np.zeros(row_cap, np.float32)
        cnt = np.zeros(4, np.int64)
        n = self.L.ref_crb_decompress(_p(buf), len(rec), row_cap, nnz_cap, _p(off), _p(lab), _p(idx), _p(val), _p(wgt), _p(cnt))
        assert n >= 0
        nnz, nval, nw, nl = (int(c) for c in cnt)
        return dict(offset=off[:n + 1].copy(), label=lab[:nl].copy(), index=idx[:nnz].copy(),
                    value=val[:nval].copy() if nval else None, weight=wgt[:nw].copy() if nw else None)

    def criteo_parse(self, text, is_train=True, row_cap=1 << 16, nnz_cap=1 << 22):
        """CriteoParser::ParseNext over one chunk -> (offset, label, index)"""
        text = bytes(text)
        off = np.zeros(row_cap + 1, np.uint64)
        lab = np.zeros(row_cap, np.float32)
        idx = np.zeros(nnz_cap, np.uint64)
        nnz = C.c_long(0)
        n = self.L.ref_criteo_parse(text, len(text), 1 if is_train else 0, row_cap, nnz_cap, _p(off), _p(lab), _p(idx),
                                    C.byref(nnz))
        assert n >= 0
        return off[:n + 1].copy(), lab[:n].copy(), idx[:nnz.value].copy()

    def adfea_parse(self, text, row_cap=1 << 16, nnz_cap=1 << 22):
        """AdfeaParser::ParseNext over one chunk -> (offset, label, index)"""
        text = bytes(text)
        off = np.zeros(row_cap + 1, np.uint64)
        lab = np.zeros(row_cap, np.float32)
        idx = np.zeros(nnz_cap, np.uint64)
        nnz = C.c_long(0)
        n = self.L.ref_adfea_parse(text, len(text), row_cap, nnz_cap, _p(off), _p(lab), _p(idx), C.byref(nnz))
        assert n >= 0
        return off[:n + 1].copy(), lab[:n].copy(), idx[:nnz.value].copy()

    def city_checker_hash64(self, s):
        s = bytes(s)
        return int(self.L.ref_city_checker_hash64(s, len(s)))

    def reverse_bytes(self, x):
        if np.isscalar(x) or isinstance(x, int):
            return int(self.L.ref_reverse_bytes(int(x)))
        return np.array([self.L.ref_reverse_bytes(int(v)) for v in np.asarray(x)], dtype=np.uint64)

    def encode_fea_grp_id(self, x, gid, nbits):
        return int(self.L.ref_encode_fea_grp_id(int(x), gid, nbits))

    def localize(self, offset, index, max_index=2 ** 64 - 1, want_cnt=True, nthreads=2, **_):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, dtype=np.uint64)
        n = len(offset) - 1
        nnz = int(offset[-1]) if n > 0 else 0
        uniq = np.zeros(max(nnz, 1), np.uint64)
        cnt = np.zeros(max(nnz, 1), np.float32) if want_cnt else None
        oi = np.zeros(max(nnz, 1), np.uint32)
        oo = np.zeros(n + 1, np.uint64)
        U = self.L.ref_localize(n, _p(offset), _p(index), None, max_index, nthreads, _p(uniq), _p(cnt), _p(oi),
                                _p(oo), None)
        out = dict(U=U, feaids=uniq[:U].copy(), index=oi[:nnz].copy(), offset=oo)
        if want_cnt:
            out["feacnt"] = cnt[:U].copy()
        return out

    def store_create(self, **kw):
        d = dict(UPDATER_DEFAULTS)
        d.update(kw)
        s = ";".join("%s=%r" % (k, v) for k, v in d.items())
        return RefStore(self, self.L.ref_store_create(s.encode()), d["V_dim"])

    def _loss_for(self, V_dim, nthreads=2):
        """one FMLoss per (V_dim, OpenMP threads); 2 threads is the reference's blk_nthreads_ (sgd_learner.h:90),
        Loss::set_nthreads accepts 1 < n < 50 (include/difacto/loss.h:81)"""
        if (V_dim, nthreads) not in self._loss:
            self._loss[(V_dim, nthreads)] = self.L.ref_fmloss_create(V_dim, nthreads)
        return self._loss[(V_dim, nthreads)]

    def fm_predict(self, V_dim, offset, index, value, weights, w_pos=None, V_pos=None, nthreads=2):
        offset = _sz(offset)
        index = np.ascontiguousarray(index, np.uint32)
        value = _f32(value)
        weights = _f32(weights)
        n = len(offset) - 1
        pred = np.zeros(n, np.float32)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        self.L.ref_fmloss_predict(self._loss_for(V_dim, nthreads), n, _p(offset), _p(index), _p(value), _p(weights),
                                  len(weights), _p(w_pos), _p(V_pos), npos, _p(pred))
        return pred

    def fm_predict_calcgrad(self, V_dim, offset, index, value, label, weights, w_pos=None, V_pos=None, nthreads=2):
        """Predict then CalcGrad on the same FMLoss instance (CalcGrad needs Predict's XV_)"""
        offset = _sz(offset)
        index = np.ascontiguousarray(index, np.uint32)
        value = _f32(value)
        weights = _f32(weights)
        label = _f32(label)
        n = len(offset) - 1
        pred = self.fm_predict(V_dim, offset, index, value, weights, w_pos, V_pos, nthreads=nthreads)
        grad = np.zeros(len(weights), np.float32)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        self.L.ref_fmloss_calcgrad(self._loss_for(V_dim, nthreads), n, _p(offset), _p(index), _p(value), _p(label),
                                   _p(weights), len(weights), _p(w_pos), _p(V_pos), npos, _p(pred), _p(grad))
        return pred, grad

    def loss_evaluate(self, label, pred):
        label, pred = _f32(label), _f32(pred)
        return float(self.L.ref_loss_evaluate(self._loss_for(0), _p(label), _p(pred), len(pred)))

    def auc_times_n(self, label, pred):
        label, pred = _f32(label), _f32(pred)
        return float(self.L.ref_auc(_p(label), _p(pred), len(pred)))

    def logit_objv(self, label, pred):
        label, pred = _f32(label), _f32(pred)
        return float(self.L.ref_logit_objv(_p(label), _p(pred), len(pred)))


class RefStore:
    def __init__(self, ref, handle, V_dim):
        self.r, self.h, self.V_dim = ref, handle, V_dim

    def __del__(self):
        if getattr(self, "h", None):
            self.r.L.ref_store_destroy(self.h)
            self.h = None

    def pull(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        n = len(keys)
        vals = np.empty(max(n * (1 + self.V_dim), 1), np.float32)  # views, no second copy: this call is timed by bench.py
        lens = np.empty(max(n, 1), np.int32)
        nv, nl = C.c_size_t(0), C.c_size_t(0)
        self.r.L.ref_store_pull(self.h, _p(keys), n, _p(vals), C.byref(nv), _p(lens), C.byref(nl))
        return vals[:nv.value], lens[:nl.value]

    def push(self, keys, val_type, vals, lens=None):
        keys = np.ascontiguousarray(keys, np.uint64)
        vals = _f32(vals)
        lens = np.zeros(0, np.int32) if lens is None else np.ascontiguousarray(lens, np.int32)
        self.r.L.ref_store_push(self.h, _p(keys), len(keys), val_type, _p(vals), len(vals), _p(lens), len(lens))
