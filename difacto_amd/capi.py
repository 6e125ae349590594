"""ctypes binding of include/difacto_hip.h (the reference-side stub for a Python
host; the C++ host in difacto_amd/host/ binds the same symbols)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFACTO_HIP_LIB: another build of the same library (tools/var_<name>.so of a same-box A/B); it must exist
LIB_PATH = os.environ.get("DIFACTO_HIP_LIB") or os.path.join(_HERE, "libdifacto_hip.so")

FEA_COUNT, WEIGHT, GRADIENT = 1, 2, 3
INIT_REFRAND, INIT_HASH = 0, 1
U64MAX = 2 ** 64 - 1

# every symbol include/difacto_hip.h declares (tests check the .so exports them all)
SYMBOLS = [
    "dfh_last_error", "dfh_updater_param_default", "dfh_ctx_create", "dfh_ctx_destroy", "dfh_ctx_sync",
    "dfh_ctx_stream", "dfh_ctx_device", "dfh_reverse_bytes", "dfh_encode_fea_grp_id", "dfh_table_create",
    "dfh_table_destroy", "dfh_table_size", "dfh_table_param", "dfh_table_bytes", "dfh_pull", "dfh_push",
    "dfh_table_export", "dfh_table_import", "dfh_fm_predict", "dfh_fm_calcgrad", "dfh_loss_evaluate",
    "dfh_auc_times_n", "dfh_batch_create", "dfh_batch_destroy", "dfh_batch_load_host", "dfh_batch_load_device",
    "dfh_localize", "dfh_batch_load_localized_host", "dfh_batch_get_localized", "dfh_batch_shape", "dfh_sgd_step",
    "dfh_batch_progress", "dfh_batch_get_pred", "dfh_row_stride", "dfh_shard_pull", "dfh_shard_push_count",
    "dfh_shard_push_grad", "dfh_batch_forward", "dfh_batch_backward", "dfh_batch_device_keys", "dfh_malloc",
    "dfh_free", "dfh_memcpy_h2d", "dfh_memcpy_d2h", "dfh_ctx_set_timing", "dfh_ctx_get_timing", "dfh_kernel_name",
    "dfh_table_warm_start", "dfh_ctx_set_pipeline", "dfh_batch_lookup", "dfh_localize_lookup", "dfh_rowbuf_create", "dfh_rowbuf_destroy", "dfh_rowbuf_load_host", "dfh_batch_gather_rows", "dfh_batch_set_option", "dfh_batch_key_ranges", "dfh_batch_attach_device",
    "dfh_batch_key_ranges_device", "dfh_shard_resolve", "dfh_shard_pull_resolved", "dfh_shard_push_count_resolved",
    "dfh_shard_push_grad_resolved", "dfh_table_check", "dfh_ctx_set_timing_mask", "dfh_table_save", "dfh_table_load", "dfh_shard_resolve_multi", "dfh_shard_push_count_multi",
    "dfh_shard_push_grad_multi", "dfh_shard_count_pull_multi", "dfh_shard_push_grad_listed", "dfh_shard_release", "dfh_ctx_set_option", "dfh_table_set_has_aux", "dfh_table_has_aux",
    "dfh_comm_unique_id", "dfh_comm_create_rccl", "dfh_comm_create_callback", "dfh_comm_destroy", "dfh_comm_rank", "dfh_comm_world",
    "dfh_comm_allreduce_sum", "dfh_shard_create", "dfh_shard_destroy", "dfh_shard_owned_range", "dfh_shard_step", "dfh_shard_prefetch_counts",
    "dfh_shard_pull_host", "dfh_shard_push_host", "dfh_comm_allgather", "dfh_shard_balanced_splits", "dfh_shard_set_exchange", "dfh_shard_set_timing", "dfh_shard_get_timing",
    "dfh_comm_stats", "dfh_comm_info", "dfh_comm_selfcheck", "dfh_comm_wire_probe", "dfh_table_capacity", "dfh_batch_prepare_rows", "dfh_rowbuf_load_host_slices",
    "dfh_batch_create_many", "dfh_shard_multi_words", "dfh_shard_reserve", "dfh_comm_create_loopback", "dfh_comm_loopback_feed", "dfh_comm_loopback_wire", "dfh_comm_loopback_wire_time",
]
XCHG_COUNTS, XCHG_KEYS, XCHG_CNT, XCHG_ROWS, XCHG_GRADS, XCHG_OTHER = range(6)
SHARD_STAGES = ("counts", "L", "K", "R", "RW", "F", "G", "P")
K_COUNT = 8
K_LOCALIZE, K_LOOKUP, K_FORWARD, K_BACKWARD, K_PULL, K_PUSH, K_MISC, K_AUC = range(8)


class UpdaterParam(C.Structure):
    """SGDUpdaterParam (src/sgd/sgd_param.h:66-107)"""
    _fields_ = [("l1", C.c_float), ("l2", C.c_float), ("V_l2", C.c_float), ("lr", C.c_float),
                ("lr_beta", C.c_float), ("V_lr", C.c_float), ("V_lr_beta", C.c_float),
                ("V_init_scale", C.c_float), ("V_dim", C.c_int), ("V_threshold", C.c_int),
                ("seed", C.c_uint), ("init_mode", C.c_int)]


class Progress(C.Structure):
    """sgd::Progress (src/sgd/sgd_utils.h:40-75)"""
    _fields_ = [("loss", C.c_float), ("penalty", C.c_float), ("auc", C.c_float),
                ("nnz_w", C.c_float), ("nrows", C.c_float)]


# dfh_alltoallv_fn: int (*)(void* user, const void* send, const size_t* send_bytes, void* recv, const size_t* recv_bytes)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t))
COMM_ID_BYTES = 128


class DfhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("difacto_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """load libdifacto_hip.so; raises (no fallback) if it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `python -m difacto_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # Load order matters when PyTorch is used in the same process (multi-GPU driver, bench.py):
    # torch bundles its own libamdhip64; if ours (linked to the system ROCm runtime of the same
    # soname) is mapped first, torch later brings up a second HIP runtime and sees no GPU.
    # Importing torch first makes both share one runtime.
    if os.environ.get("DIFACTO_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, sz, i32, u64, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_float
    PP = C.POINTER
    L.dfh_last_error.restype = C.c_char_p
    L.dfh_updater_param_default.argtypes = [PP(UpdaterParam), i32]
    L.dfh_ctx_create.argtypes = [i32, vp, PP(vp)]
    L.dfh_ctx_destroy.argtypes = [vp]
    L.dfh_ctx_sync.argtypes = [vp]
    L.dfh_ctx_stream.restype = vp
    L.dfh_ctx_stream.argtypes = [vp]
    L.dfh_ctx_device.argtypes = [vp]
    L.dfh_reverse_bytes.restype = u64
    L.dfh_reverse_bytes.argtypes = [u64]
    L.dfh_encode_fea_grp_id.restype = u64
    L.dfh_encode_fea_grp_id.argtypes = [u64, i32, i32]
    L.dfh_table_create.argtypes = [vp, PP(UpdaterParam), u64, PP(vp)]
    L.dfh_table_destroy.argtypes = [vp]
    L.dfh_table_size.argtypes = [vp, PP(u64)]
    L.dfh_table_param.argtypes = [vp, PP(UpdaterParam)]
    L.dfh_table_capacity.argtypes = [vp, PP(u64), PP(u64)]
    L.dfh_table_bytes.restype = u64
    L.dfh_table_bytes.argtypes = [vp]
    L.dfh_pull.argtypes = [vp, vp, sz, vp, PP(sz), vp, PP(sz)]
    L.dfh_push.argtypes = [vp, vp, sz, i32, vp, sz, vp, sz]
    L.dfh_table_export.argtypes = [vp, u64, vp, vp, vp, vp, PP(u64)]
    L.dfh_table_import.argtypes = [vp, u64, vp, vp, vp, vp]
    L.dfh_fm_predict.argtypes = [vp, i32, sz, vp, vp, vp, vp, sz, vp, vp, sz, vp]
    L.dfh_fm_calcgrad.argtypes = [vp, i32, sz, vp, vp, vp, vp, vp, sz, vp, vp, sz, vp, vp]
    L.dfh_loss_evaluate.argtypes = [vp, vp, vp, sz, PP(f32)]
    L.dfh_auc_times_n.argtypes = [vp, vp, vp, sz, PP(f32)]
    L.dfh_batch_create.argtypes = [vp, sz, sz, PP(vp)]
    L.dfh_batch_destroy.argtypes = [vp]
    L.dfh_batch_create_many.argtypes = [vp, i32, sz, sz, PP(vp)]
    L.dfh_batch_load_host.argtypes = [vp, sz, vp, vp, vp, vp]
    L.dfh_batch_load_device.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.dfh_batch_attach_device.argtypes = [vp, sz, sz, vp, vp, vp, vp]
    L.dfh_localize.argtypes = [vp, u64]
    L.dfh_batch_load_localized_host.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, sz]
    L.dfh_batch_get_localized.argtypes = [vp, PP(sz), vp, vp, vp]
    L.dfh_batch_shape.argtypes = [vp, PP(sz), PP(sz), PP(sz)]
    L.dfh_sgd_step.argtypes = [vp, vp, i32, i32]
    L.dfh_batch_progress.argtypes = [vp, PP(Progress), i32]
    L.dfh_batch_get_pred.argtypes = [vp, vp]
    L.dfh_row_stride.restype = sz
    L.dfh_row_stride.argtypes = [i32]
    L.dfh_shard_pull.argtypes = [vp, vp, sz, vp]
    L.dfh_shard_push_count.argtypes = [vp, vp, sz, vp]
    L.dfh_shard_push_grad.argtypes = [vp, vp, sz, vp]
    L.dfh_batch_forward.argtypes = [vp, i32, vp]
    L.dfh_batch_backward.argtypes = [vp, i32, vp, vp]
    L.dfh_batch_device_keys.argtypes = [vp, PP(vp), PP(vp), PP(sz)]
    L.dfh_malloc.argtypes = [vp, sz, PP(vp)]
    L.dfh_free.argtypes = [vp, vp]
    L.dfh_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.dfh_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.dfh_table_warm_start.argtypes = [vp, vp, sz, f32, f32]
    L.dfh_ctx_set_pipeline.argtypes = [vp, i32]
    L.dfh_batch_lookup.argtypes = [vp, vp]
    L.dfh_localize_lookup.argtypes = [vp, vp, C.c_uint64]
    L.dfh_rowbuf_create.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.dfh_rowbuf_destroy.argtypes = [vp]
    L.dfh_rowbuf_load_host.argtypes = [vp, C.c_size_t, vp, vp, vp]
    L.dfh_batch_gather_rows.argtypes = [vp, C.c_size_t, vp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.dfh_batch_prepare_rows.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t), C.c_uint64]
    L.dfh_batch_set_option.argtypes = [vp, C.c_char_p, i32]
    L.dfh_batch_key_ranges.argtypes = [vp, i32, vp]
    L.dfh_batch_key_ranges_device.argtypes = [vp, i32, vp, vp]
    L.dfh_shard_resolve.argtypes = [vp, vp, sz, vp]
    L.dfh_shard_pull_resolved.argtypes = [vp, vp, sz, vp]
    L.dfh_shard_push_count_resolved.argtypes = [vp, vp, vp, sz, vp]
    L.dfh_shard_push_grad_resolved.argtypes = [vp, vp, vp, sz, vp]
    L.dfh_table_check.argtypes = [vp]
    L.dfh_table_save.argtypes = [vp, C.c_char_p, i32, PP(u64)]
    L.dfh_table_load.argtypes = [vp, C.c_char_p, u64, u64, PP(i32), PP(u64)]
    L.dfh_shard_resolve_multi.argtypes = [vp, vp, vp, i32, i32, vp]
    L.dfh_shard_push_count_multi.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.dfh_shard_push_grad_multi.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.dfh_shard_count_pull_multi.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.dfh_shard_push_grad_listed.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.dfh_shard_release.argtypes = [vp, vp, sz, i32]
    L.dfh_shard_multi_words.restype = sz
    L.dfh_shard_multi_words.argtypes = [sz, i32]
    L.dfh_ctx_set_timing.argtypes = [vp, i32]
    L.dfh_ctx_set_timing_mask.argtypes = [vp, C.c_uint32]
    L.dfh_ctx_get_timing.argtypes = [vp, i32, vp, vp]
    L.dfh_kernel_name.restype = C.c_char_p
    L.dfh_kernel_name.argtypes = [i32]
    L.dfh_ctx_set_option.argtypes = [vp, C.c_char_p, i32]
    L.dfh_table_set_has_aux.argtypes = [vp, i32]
    L.dfh_table_has_aux.argtypes = [vp]
    L.dfh_comm_unique_id.argtypes = [vp]
    L.dfh_comm_create_rccl.argtypes = [vp, i32, i32, vp, PP(vp)]
    L.dfh_comm_create_callback.argtypes = [vp, i32, i32, ALLTOALLV_FN, vp, PP(vp)]
    L.dfh_comm_destroy.argtypes = [vp]
    L.dfh_comm_rank.argtypes = [vp]
    L.dfh_comm_world.argtypes = [vp]
    L.dfh_comm_allreduce_sum.argtypes = [vp, vp, i32]
    L.dfh_shard_create.argtypes = [vp, vp, vp, PP(vp)]
    L.dfh_shard_destroy.argtypes = [vp]
    L.dfh_shard_owned_range.argtypes = [vp, vp, PP(u64), PP(u64)]
    L.dfh_shard_step.argtypes = [vp, vp, i32, i32, PP(i32)]
    L.dfh_shard_prefetch_counts.argtypes = [vp, vp]
    L.dfh_shard_pull_host.argtypes = [vp, vp, C.c_size_t, vp, PP(C.c_size_t), vp, PP(C.c_size_t)]
    L.dfh_shard_push_host.argtypes = [vp, vp, C.c_size_t, i32, vp, C.c_size_t, vp, C.c_size_t]
    L.dfh_comm_allgather.argtypes = [vp, vp, C.c_size_t, vp]
    L.dfh_comm_stats.argtypes = [vp, i32, vp, vp, vp]
    L.dfh_comm_info.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.dfh_comm_selfcheck.argtypes = [vp, C.c_double]
    L.dfh_comm_wire_probe.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.dfh_shard_balanced_splits.argtypes = [vp, vp, C.c_size_t, vp]
    L.dfh_shard_set_exchange.argtypes = [vp, i32]
    L.dfh_shard_reserve.argtypes = [vp, sz, sz]
    L.dfh_comm_create_loopback.argtypes = [vp, i32, i32, PP(vp)]
    L.dfh_comm_loopback_feed.argtypes = [vp, i32, vp, i32]
    L.dfh_comm_loopback_wire.argtypes = [vp, C.c_double, C.c_double]
    L.dfh_comm_loopback_wire_time.argtypes = [vp, i32, PP(C.c_double)]
    L.dfh_shard_set_timing.argtypes = [vp, i32]
    L.dfh_shard_get_timing.argtypes = [vp, i32, vp, PP(u64)]
    _lib = L
    return L


def _ck(rc):
    if rc != 0:
        raise DfhError(rc, lib().dfh_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dp(x):
    """device pointer: int, c_void_p or a torch tensor"""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return x if isinstance(x, C.c_void_p) else C.c_void_p(int(x))


def make_param(V_dim=0, init_mode=INIT_HASH, **kw):
    p = UpdaterParam()
    lib().dfh_updater_param_default(C.byref(p), V_dim)
    p.init_mode = init_mode
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def reverse_bytes(x):
    return int(lib().dfh_reverse_bytes(int(x)))


class Context:
    """a HIP device + stream (dfh_ctx)"""

    def __init__(self, device=0, stream=None):
        self.h = C.c_void_p()
        _ck(lib().dfh_ctx_create(device, _dp(stream), C.byref(self.h)))

    def sync(self):
        _ck(lib().dfh_ctx_sync(self.h))

    def set_pipeline(self, on=True):
        """prepare later batches (copy, localize, lookup) on `on` preparation streams (True = 1)
        while an earlier batch trains on the main stream"""
        _ck(lib().dfh_ctx_set_pipeline(self.h, int(on)))

    def set_option(self, name, value):
        """validated launch tuning: fwd_depth, fwd_blocks, bwd_small_blocks, prep_priority"""
        _ck(lib().dfh_ctx_set_option(self.h, name.encode(), int(value)))

    def set_timing(self, on=True):
        _ck(lib().dfh_ctx_set_timing(self.h, 1 if on else 0))

    def set_timing_mask(self, mask):
        """time only the kernels whose bit (1 << K_*) is set"""
        _ck(lib().dfh_ctx_set_timing_mask(self.h, mask))

    def get_timing(self, reset=True):
        """{kernel name: (total_ms, calls)} from HIP events on this context's stream"""
        ms = np.zeros(K_COUNT, np.float64)
        calls = np.zeros(K_COUNT, np.uint64)
        _ck(lib().dfh_ctx_get_timing(self.h, 1 if reset else 0, _p(ms), _p(calls)))
        return {lib().dfh_kernel_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(K_COUNT)}

    @property
    def stream(self):
        return lib().dfh_ctx_stream(self.h)

    def close(self):
        if self.h:
            lib().dfh_ctx_destroy(self.h)
            self.h = None

    # literal Loss API ------------------------------------------------------
    def fm_predict(self, V_dim, offset, index, value, weights, w_pos=None, V_pos=None, pred0=None):
        """FMLoss::Predict (src/loss/fm_loss.h:67-119)"""
        offset = np.ascontiguousarray(offset, np.uint64)
        index = np.ascontiguousarray(index, np.uint32)
        value = None if value is None else np.ascontiguousarray(value, np.float32)
        weights = np.ascontiguousarray(weights, np.float32)
        n = len(offset) - 1
        pred = np.zeros(n, np.float32) if pred0 is None else np.array(pred0, np.float32)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        _ck(lib().dfh_fm_predict(self.h, V_dim, n, _p(offset), _p(index), _p(value), _p(weights), len(weights),
                                 _p(w_pos), _p(V_pos), npos, _p(pred)))
        return pred

    def fm_calcgrad(self, V_dim, offset, index, value, label, weights, pred, w_pos=None, V_pos=None):
        """FMLoss::CalcGrad (src/loss/fm_loss.h:148-199)"""
        offset = np.ascontiguousarray(offset, np.uint64)
        index = np.ascontiguousarray(index, np.uint32)
        value = None if value is None else np.ascontiguousarray(value, np.float32)
        weights = np.ascontiguousarray(weights, np.float32)
        label = np.ascontiguousarray(label, np.float32)
        pred = np.ascontiguousarray(pred, np.float32)
        n = len(offset) - 1
        grad = np.zeros(len(weights), np.float32)
        npos = 0 if w_pos is None else len(w_pos)
        w_pos = None if w_pos is None else np.ascontiguousarray(w_pos, np.int32)
        V_pos = None if V_pos is None else np.ascontiguousarray(V_pos, np.int32)
        _ck(lib().dfh_fm_calcgrad(self.h, V_dim, n, _p(offset), _p(index), _p(value), _p(label), _p(weights),
                                  len(weights), _p(w_pos), _p(V_pos), npos, _p(pred), _p(grad)))
        return grad

    def auc_times_n(self, label, pred):
        """BinClassMetric::AUC (src/loss/bin_class_metric.h:35-56): AUC * n"""
        label = np.ascontiguousarray(label, np.float32)
        pred = np.ascontiguousarray(pred, np.float32)
        o = C.c_float(0)
        _ck(lib().dfh_auc_times_n(self.h, _p(label), _p(pred), len(pred), C.byref(o)))
        return o.value

    def loss_evaluate(self, label, pred):
        label = np.ascontiguousarray(label, np.float32)
        pred = np.ascontiguousarray(pred, np.float32)
        o = C.c_float(0)
        _ck(lib().dfh_loss_evaluate(self.h, _p(label), _p(pred), len(pred), C.byref(o)))
        return o.value


class Table:
    """one shard of the model in HBM (dfh_table): replaces Store + SGDUpdater state"""

    def __init__(self, ctx, capacity, V_dim=0, init_mode=INIT_HASH, **kw):
        self.ctx = ctx
        self.param = make_param(V_dim, init_mode, **kw)
        self.V_dim = V_dim
        self.h = C.c_void_p()
        _ck(lib().dfh_table_create(ctx.h, C.byref(self.param), capacity, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().dfh_table_destroy(self.h)
            self.h = None

    def size(self):
        n = C.c_uint64(0)
        _ck(lib().dfh_table_size(self.h, C.byref(n)))
        return n.value

    def capacity(self):
        """(rows the arrays hold now, number of growths) — a table created with capacity 0 grows"""
        c, g = C.c_uint64(0), C.c_uint64(0)
        _ck(lib().dfh_table_capacity(self.h, C.byref(c), C.byref(g)))
        return c.value, g.value

    def bytes(self):
        return int(lib().dfh_table_bytes(self.h))

    def pull(self, keys):
        """Store::Pull(kWeight) -> (vals ragged, lens)"""
        keys = np.ascontiguousarray(keys, np.uint64)
        n = len(keys)
        vals = np.zeros(max(n * (1 + self.V_dim), 1), np.float32)
        lens = np.zeros(max(n, 1), np.int32)
        nv, nl = C.c_size_t(0), C.c_size_t(0)
        _ck(lib().dfh_pull(self.h, _p(keys), n, _p(vals), C.byref(nv), _p(lens), C.byref(nl)))
        return vals[:nv.value].copy(), lens[:nl.value].copy()

    def push(self, keys, val_type, vals, lens=None):
        """Store::Push(kFeaCount | kGradient)"""
        keys = np.ascontiguousarray(keys, np.uint64)
        vals = np.ascontiguousarray(vals, np.float32)
        lens = np.zeros(0, np.int32) if lens is None else np.ascontiguousarray(lens, np.int32)
        _ck(lib().dfh_push(self.h, _p(keys), len(keys), val_type, _p(vals), len(vals), _p(lens), len(lens)))

    def export(self):
        n = self.size()
        k = self.V_dim
        keys = np.zeros(max(n, 1), np.uint64)
        scal = np.zeros(max(n, 1) * 4, np.float32)
        has = np.zeros(max(n, 1), np.int32)
        V = np.zeros(max(n * 2 * k, 1), np.float32)
        m = C.c_uint64(0)
        _ck(lib().dfh_table_export(self.h, max(n, 1), _p(keys), _p(scal), _p(has), _p(V), C.byref(m)))
        n = m.value
        return dict(keys=keys[:n], scal=scal[:4 * n].reshape(n, 4), has_V=has[:n], V=V[:n * 2 * k].reshape(n, 2 * k) if k else None)

    def import_(self, keys, scal, has_V, V=None):
        keys = np.ascontiguousarray(keys, np.uint64)
        scal = np.ascontiguousarray(scal, np.float32)
        has_V = np.ascontiguousarray(has_V, np.int32)
        V = None if V is None else np.ascontiguousarray(V, np.float32)
        _ck(lib().dfh_table_import(self.h, len(keys), _p(keys), _p(scal), _p(has_V), _p(V)))

    def save(self, path, save_aux=True):
        """Updater::Save to a file (the C++ host's model format); returns the number of entries written"""
        n = C.c_uint64(0)
        _ck(lib().dfh_table_save(self.h, str(path).encode(), 1 if save_aux else 0, C.byref(n)))
        return n.value

    def load(self, path, key_lo=0, key_hi=0):
        """Updater::Load of the keys in [key_lo, key_hi) (key_hi = 0: unbounded) -> (entries loaded, has_aux)"""
        n, aux = C.c_uint64(0), C.c_int(0)
        _ck(lib().dfh_table_load(self.h, str(path).encode(), key_lo, key_hi, C.byref(aux), C.byref(n)))
        return n.value, bool(aux.value)

    @property
    def has_aux(self):
        """SGDUpdater::has_aux_: False after loading a model saved without optimiser state"""
        return bool(lib().dfh_table_has_aux(self.h))

    def set_has_aux(self, on):
        _ck(lib().dfh_table_set_has_aux(self.h, 1 if on else 0))

    def warm_start(self, d_keys, n, w0=0.01, cnt0=100.0):
        _ck(lib().dfh_table_warm_start(self.h, _dp(d_keys), n, w0, cnt0))

    # device-pointer (sharded) calls
    def shard_pull(self, d_keys, n, d_rows):
        _ck(lib().dfh_shard_pull(self.h, _dp(d_keys), n, _dp(d_rows)))

    def shard_push_count(self, d_keys, n, d_cnt):
        _ck(lib().dfh_shard_push_count(self.h, _dp(d_keys), n, _dp(d_cnt)))

    def shard_push_grad(self, d_keys, n, d_grads):
        _ck(lib().dfh_shard_push_grad(self.h, _dp(d_keys), n, _dp(d_grads)))

    # resolved form: probe all received keys once, then work on row ids
    def shard_resolve(self, d_keys, n, d_rowid):
        _ck(lib().dfh_shard_resolve(self.h, _dp(d_keys), n, _dp(d_rowid)))

    def shard_pull_resolved(self, d_rowid, n, d_rows):
        _ck(lib().dfh_shard_pull_resolved(self.h, _dp(d_rowid), n, _dp(d_rows)))

    def shard_push_count_resolved(self, d_rowid, d_keys, n, d_cnt):
        _ck(lib().dfh_shard_push_count_resolved(self.h, _dp(d_rowid), _dp(d_keys), n, _dp(d_cnt)))

    def shard_push_grad_resolved(self, d_rowid, d_keys, n, d_grads):
        _ck(lib().dfh_shard_push_grad_resolved(self.h, _dp(d_rowid), _dp(d_keys), n, _dp(d_grads)))

    # all source ranks of a step in one launch (seg: nsrc+1 host offsets into the concatenated lists)
    @staticmethod
    def _seg(seg):
        a = np.ascontiguousarray(seg, dtype=np.uint64)
        return a, len(a) - 1

    def shard_resolve_multi(self, d_keys, seg, d_rowid, mask_slot=0):
        a, n = self._seg(seg)
        _ck(lib().dfh_shard_resolve_multi(self.h, _dp(d_keys), _p(a), n, mask_slot, _dp(d_rowid)))

    def shard_push_count_multi(self, d_rowid, d_keys, seg, d_cnt, mask_slot=0):
        a, n = self._seg(seg)
        _ck(lib().dfh_shard_push_count_multi(self.h, _dp(d_rowid), _dp(d_keys), _p(a), n, mask_slot, _dp(d_cnt)))

    def shard_push_grad_multi(self, d_rowid, d_keys, seg, d_grads, mask_slot=0):
        a, n = self._seg(seg)
        _ck(lib().dfh_shard_push_grad_multi(self.h, _dp(d_rowid), _dp(d_keys), _p(a), n, mask_slot, _dp(d_grads)))

    def shard_count_pull_multi(self, d_rowid, d_keys, seg, d_cnt, d_rows, mask_slot=0):
        """Push(kFeaCount) of all sources (d_cnt or None) + Pull per distinct key; leaves the key lists of push_grad_listed"""
        a, n = self._seg(seg)
        _ck(lib().dfh_shard_count_pull_multi(self.h, _dp(d_rowid), _dp(d_keys), _p(a), n, mask_slot,
                                             _dp(d_cnt) if d_cnt is not None else None, _dp(d_rows)))

    def shard_push_grad_listed(self, d_rowid, d_keys, seg, d_grads, mask_slot=0):
        a, n = self._seg(seg)
        _ck(lib().dfh_shard_push_grad_listed(self.h, _dp(d_rowid), _dp(d_keys), _p(a), n, mask_slot, _dp(d_grads)))

    def shard_release(self, d_rowid, n, mask_slot=0):
        _ck(lib().dfh_shard_release(self.h, _dp(d_rowid), n, mask_slot))

    def check(self):
        """raise if the device-side error word is set (capacity / duplicate key / V mismatch)"""
        _ck(lib().dfh_table_check(self.h))


class RowBuf:
    """a shuffle buffer in HBM (dfh_rowbuf): rows are gathered out of it on the device (Batch.gather_rows)"""

    def __init__(self, ctx, max_rows, max_nnz):
        self.ctx = ctx
        self.h = C.c_void_p()
        _ck(lib().dfh_rowbuf_create(ctx.h, max_rows, max_nnz, C.byref(self.h)))

    def load_host(self, offset, index, value=None):
        offset = np.ascontiguousarray(offset, np.uint64)
        index = np.ascontiguousarray(index, np.uint64)
        value = None if value is None else np.ascontiguousarray(value, np.float32)
        _ck(lib().dfh_rowbuf_load_host(self.h, len(offset) - 1, _p(offset), _p(index), _p(value)))

    def close(self):
        if self.h:
            lib().dfh_rowbuf_destroy(self.h)
            self.h = None


def create_batches(ctx, n, max_rows, max_nnz):
    """n Batch objects carved from one device allocation (dfh_batch_create_many)"""
    hs = (C.c_void_p * n)()
    _ck(lib().dfh_batch_create_many(ctx.h, n, max_rows, max_nnz, hs))
    out = []
    for i in range(n):
        b = Batch.__new__(Batch)
        b.ctx, b.h = ctx, C.c_void_p(hs[i])
        out.append(b)
    return out


class Batch:
    """a device-resident minibatch + workspace (dfh_batch)"""

    def __init__(self, ctx, max_rows, max_nnz):
        self.ctx = ctx
        self.h = C.c_void_p()
        _ck(lib().dfh_batch_create(ctx.h, max_rows, max_nnz, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().dfh_batch_destroy(self.h)
            self.h = None

    def load_host(self, offset, index, value, label):
        """raw CSR with u64 feature ids (what Reader::Value() yields)"""
        offset = np.ascontiguousarray(offset, np.uint64)
        index = np.ascontiguousarray(index, np.uint64)
        value = None if value is None else np.ascontiguousarray(value, np.float32)
        label = np.ascontiguousarray(label, np.float32)
        _ck(lib().dfh_batch_load_host(self.h, len(offset) - 1, _p(offset), _p(index), _p(value), _p(label)))

    def load_device(self, nrows, nnz, d_offset, d_index, d_value, d_label):
        _ck(lib().dfh_batch_load_device(self.h, nrows, nnz, _dp(d_offset), _dp(d_index), _dp(d_value), _dp(d_label)))

    def attach_device(self, nrows, nnz, d_offset, d_index, d_value, d_label):
        """zero-copy: read the caller's device arrays in place"""
        _ck(lib().dfh_batch_attach_device(self.h, nrows, nnz, _dp(d_offset), _dp(d_index), _dp(d_value), _dp(d_label)))

    def localize(self, max_index=U64MAX, table=None):
        """Localizer::Compact on device; with a table also the key-index probe (dfh_localize_lookup)"""
        if table is None:
            _ck(lib().dfh_localize(self.h, max_index))
        else:
            _ck(lib().dfh_localize_lookup(table.h, self.h, max_index))

    def set_option(self, name, value):
        _ck(lib().dfh_batch_set_option(self.h, name.encode(), int(value)))

    def lookup(self, table):
        _ck(lib().dfh_batch_lookup(table.h, self.h))

    def gather_rows(self, offset, label, segments):
        """dfh_batch_gather_rows: the minibatch = rows `rows` of row buffer `rb` for every (rb, rows) of `segments`, in order;
        offset / label are the minibatch's own (cumulative row lengths, labels)"""
        offset = np.ascontiguousarray(offset, np.uint64)
        label = np.ascontiguousarray(label, np.float32)
        rows = [np.ascontiguousarray(r, np.uint32) for _, r in segments]
        n = len(segments)
        bufs = (C.c_void_p * n)(*[rb.h for rb, _ in segments])
        ptrs = (C.c_void_p * n)(*[r.ctypes.data for r in rows])
        cnts = (C.c_size_t * n)(*[len(r) for r in rows])
        _ck(lib().dfh_batch_gather_rows(self.h, len(label), _p(offset), _p(label), n, bufs, ptrs, cnts))

    def prepare_rows(self, table, offset, label, segments, max_index=2 ** 64 - 1):
        """dfh_batch_prepare_rows: gather_rows + localize + lookup(table) as one preparation phase"""
        offset = np.ascontiguousarray(offset, np.uint64)
        label = np.ascontiguousarray(label, np.float32)
        rows = [np.ascontiguousarray(r, np.uint32) for _, r in segments]
        n = len(segments)
        bufs = (C.c_void_p * n)(*[rb.h for rb, _ in segments])
        ptrs = (C.c_void_p * n)(*[r.ctypes.data for r in rows])
        cnts = (C.c_size_t * n)(*[len(r) for r in rows])
        _ck(lib().dfh_batch_prepare_rows(table.h, self.h, len(label), _p(offset), _p(label), n, bufs, ptrs, cnts, max_index))

    def load_localized_host(self, offset, index, value, label, feaids, feacnt=None):
        offset = np.ascontiguousarray(offset, np.uint64)
        index = np.ascontiguousarray(index, np.uint32)
        value = None if value is None else np.ascontiguousarray(value, np.float32)
        label = np.ascontiguousarray(label, np.float32)
        feaids = np.ascontiguousarray(feaids, np.uint64)
        feacnt = None if feacnt is None else np.ascontiguousarray(feacnt, np.float32)
        _ck(lib().dfh_batch_load_localized_host(self.h, len(offset) - 1, _p(offset), _p(index), _p(value), _p(label),
                                                _p(feaids), _p(feacnt), len(feaids)))

    def shape(self, want_U=True):
        a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        _ck(lib().dfh_batch_shape(self.h, C.byref(a), C.byref(b), C.byref(c) if want_U else None))
        return a.value, b.value, c.value

    def get_localized(self):
        nrows, nnz, U = self.shape()
        feaids = np.zeros(max(U, 1), np.uint64)
        cnt = np.zeros(max(U, 1), np.float32)
        index = np.zeros(max(nnz, 1), np.uint32)
        u = C.c_size_t(0)
        _ck(lib().dfh_batch_get_localized(self.h, C.byref(u), _p(feaids), _p(cnt), _p(index)))
        return dict(U=U, feaids=feaids[:U], feacnt=cnt[:U], index=index[:nnz])

    def sgd_step(self, table, is_train=True, push_cnt=False):
        """the batch executor of SGDLearner::IterateData, on device (async)"""
        _ck(lib().dfh_sgd_step(table.h, self.h, 1 if is_train else 0, 1 if push_cnt else 0))

    def progress(self, reset=True):
        p = Progress()
        _ck(lib().dfh_batch_progress(self.h, C.byref(p), 1 if reset else 0))
        return p

    def pred(self):
        nrows, _, _ = self.shape(want_U=False)
        out = np.zeros(nrows, np.float32)
        _ck(lib().dfh_batch_get_pred(self.h, _p(out)))
        return out

    def forward(self, V_dim, d_rows):
        _ck(lib().dfh_batch_forward(self.h, V_dim, _dp(d_rows)))

    def backward(self, V_dim, d_rows, d_grads):
        _ck(lib().dfh_batch_backward(self.h, V_dim, _dp(d_rows), _dp(d_grads)))

    def key_ranges(self, nparts):
        """bounds[nparts+1]: shard d owns feaids[bounds[d]:bounds[d+1]] (key-range partition)"""
        out = np.zeros(nparts + 1, np.uint32)
        _ck(lib().dfh_batch_key_ranges(self.h, nparts, _p(out)))
        return out

    def device_keys(self):
        a, b, u = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        _ck(lib().dfh_batch_device_keys(self.h, C.byref(a), C.byref(b), C.byref(u)))
        return a.value, b.value, u.value

    def device_key_ptrs(self):
        """device pointers of feaids / feacnt without synchronising (U comes from key_ranges_device)"""
        a, b = C.c_void_p(), C.c_void_p()
        _ck(lib().dfh_batch_device_keys(self.h, C.byref(a), C.byref(b), None))
        return a.value, b.value

    def key_ranges_device(self, nparts, d_bounds, d_splits=None):
        _ck(lib().dfh_batch_key_ranges_device(self.h, nparts, _dp(d_splits), _dp(d_bounds)))


class DeviceBuffer:
    """raw device memory owned through the C ABI (dfh_malloc)"""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = C.c_void_p()
        _ck(lib().dfh_malloc(ctx.h, nbytes, C.byref(self.ptr)))

    @classmethod
    def from_numpy(cls, ctx, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(ctx, max(arr.nbytes, 16))
        if arr.nbytes:
            _ck(lib().dfh_memcpy_h2d(ctx.h, b.ptr, _p(arr), arr.nbytes))
        return b

    def to_numpy(self, dtype, count):
        out = np.zeros(count, dtype)
        if out.nbytes:
            _ck(lib().dfh_memcpy_d2h(self.ctx.h, _p(out), self.ptr, out.nbytes))
        return out

    def at(self, byte_offset):
        return C.c_void_p(self.ptr.value + byte_offset)

    def close(self):
        if self.ptr:
            lib().dfh_free(self.ctx.h, self.ptr)
            self.ptr = None


def row_stride(V_dim):
    return int(lib().dfh_row_stride(V_dim))


def multi_words(n, nsrc):
    """32-bit words of the row-word buffer dfh_shard_resolve_multi fills for n entries of nsrc sources"""
    return int(lib().dfh_shard_multi_words(n, nsrc))


class Comm:
    """dfh_comm: the transport of the sharded store — RCCL (product) or a host callback (tests)"""

    def __init__(self, handle, keep=None):
        self.h, self._keep = handle, keep

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        _ck(lib().dfh_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def rccl(cls, ctx, rank, world, unique_id):
        h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, COMM_ID_BYTES)
        _ck(lib().dfh_comm_create_rccl(ctx.h, rank, world, C.cast(buf, C.c_void_p), C.byref(h)))
        return cls(h)

    @classmethod
    def callback(cls, ctx, rank, world, fn):
        """fn(send: np.uint8[...], send_bytes: list, recv: np.uint8[...] (to fill), recv_bytes: list)"""
        def tramp(_user, p_send, p_sb, p_recv, p_rb):
            try:
                sb = [int(p_sb[i]) for i in range(world)]
                rb = [int(p_rb[i]) for i in range(world)]
                send = np.ctypeslib.as_array(C.cast(p_send, C.POINTER(C.c_uint8)), shape=(max(sum(sb), 1),))[:sum(sb)]
                recv = np.ctypeslib.as_array(C.cast(p_recv, C.POINTER(C.c_uint8)), shape=(max(sum(rb), 1),))[:sum(rb)]
                fn(send, sb, recv, rb)
                return 0
            except Exception:  # noqa: BLE001 - reported through the C return code
                import traceback
                traceback.print_exc()
                return 1
        cb = ALLTOALLV_FN(tramp)
        h = C.c_void_p()
        _ck(lib().dfh_comm_create_callback(ctx.h, rank, world, cb, None, C.byref(h)))
        return cls(h, keep=cb)

    @classmethod
    def loopback(cls, ctx, rank, world):
        """measurement only: rank `rank` of a `world`-rank job alone on its GPU; the peers' messages are device copies
        out of buffers fed with feed() (include/difacto_hip.h, dfh_comm_create_loopback)"""
        h = C.c_void_p()
        _ck(lib().dfh_comm_create_loopback(ctx.h, rank, world, C.byref(h)))
        return cls(h)

    def feed(self, kind, d_src, sticky=False):
        """source (device pointer) of the next exchange of `kind` (XCHG_*), laid out like its receive side"""
        _ck(lib().dfh_comm_loopback_feed(self.h, kind, d_src, 1 if sticky else 0))

    def wire(self, link_gbps, latency_us):
        _ck(lib().dfh_comm_loopback_wire(self.h, float(link_gbps), float(latency_us)))

    def wire_time_us(self, reset=True):
        v = C.c_double(0)
        _ck(lib().dfh_comm_loopback_wire_time(self.h, 1 if reset else 0, C.byref(v)))
        return v.value

    def allreduce_sum(self, vals):
        a = np.ascontiguousarray(vals, np.float64).copy()
        _ck(lib().dfh_comm_allreduce_sum(self.h, _p(a), len(a)))
        return a

    def allgather(self, arr):
        """every rank's array (same dtype and length on every rank) -> [world, len]"""
        a = np.ascontiguousarray(arr)
        out = np.zeros((lib().dfh_comm_world(self.h),) + a.shape, a.dtype)
        _ck(lib().dfh_comm_allgather(self.h, _p(a), a.nbytes, _p(out)))
        return out

    def stats(self, reset=False):
        """(bytes sent to other ranks, bytes received from them, message groups) since the last reset"""
        a, b, g = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _ck(lib().dfh_comm_stats(self.h, 1 if reset else 0, C.byref(a), C.byref(b), C.byref(g)))
        return a.value, b.value, g.value

    def info(self):
        """the bound transport: RCCL version + the file it was bound from, or the host callback"""
        buf = C.create_string_buffer(512)
        _ck(lib().dfh_comm_info(self.h, buf, 512))
        return buf.value.decode()

    def wire_probe(self, bytes_per_peer, reps=20):
        """microseconds per grouped exchange in which every rank sends bytes_per_peer to (and receives as much from) every
        other rank; COLLECTIVE"""
        us = C.c_double(0)
        _ck(lib().dfh_comm_wire_probe(self.h, int(bytes_per_peer), int(reps), C.byref(us)))
        return us.value

    def selfcheck(self, timeout_s=60.0):
        """collective start-up check (first exchange polled with a timeout): raises instead of hanging"""
        _ck(lib().dfh_comm_selfcheck(self.h, float(timeout_s)))

    def balanced_splits(self, sample_keys):
        """collective: split keys (world-1, identical on every rank) at the quantiles of the union of the ranks' key samples"""
        k = np.ascontiguousarray(sample_keys, np.uint64)
        w = lib().dfh_comm_world(self.h)
        out = np.zeros(max(w - 1, 1), np.uint64)
        _ck(lib().dfh_shard_balanced_splits(self.h, _p(k) if len(k) else None, len(k), _p(out)))
        return out[:w - 1]

    def close(self):
        if self.h:
            lib().dfh_comm_destroy(self.h)
            self.h = None


class Shard:
    """dfh_shard: this rank's part of the key-range-sharded model + the per-step exchange"""

    def __init__(self, table, comm, splits=None):
        self.h = C.c_void_p()
        self.table, self.comm = table, comm
        self.splits = None if splits is None else np.ascontiguousarray(splits, np.uint64)
        _ck(lib().dfh_shard_create(table.h, comm.h, _p(self.splits), C.byref(self.h)))

    def step(self, batch, is_train=True, push_cnt=False):
        """collective; batch = None when this rank's data is exhausted.  -> True while any rank had a minibatch"""
        act = C.c_int(0)
        _ck(lib().dfh_shard_step(self.h, batch.h if batch is not None else None, 1 if is_train else 0, 1 if push_cnt else 0,
                                 C.byref(act)))
        return bool(act.value)

    def prefetch_counts(self, next_batch):
        """collective, right before step(): the minibatch of the FOLLOWING step (localize queued) or None"""
        _ck(lib().dfh_shard_prefetch_counts(self.h, next_batch.h if next_batch is not None else None))

    def pull_host(self, keys):
        """collective literal Store::Pull(kWeight) -> (vals ragged, lens)"""
        keys = np.ascontiguousarray(keys, np.uint64)
        n, k = len(keys), self.table.V_dim
        vals = np.zeros(max(n * (1 + k), 1), np.float32)
        lens = np.zeros(max(n, 1), np.int32)
        nv, nl = C.c_size_t(0), C.c_size_t(0)
        _ck(lib().dfh_shard_pull_host(self.h, _p(keys) if n else None, n, _p(vals), C.byref(nv), _p(lens), C.byref(nl)))
        return vals[:nv.value], lens[:nl.value]

    def push_host(self, keys, val_type, vals, lens=None):
        """collective literal Store::Push(kFeaCount | kGradient)"""
        keys = np.ascontiguousarray(keys, np.uint64)
        vals = np.ascontiguousarray(vals, np.float32)
        lens = np.zeros(0, np.int32) if lens is None else np.ascontiguousarray(lens, np.int32)
        _ck(lib().dfh_shard_push_host(self.h, _p(keys) if len(keys) else None, len(keys), val_type, _p(vals) if len(vals) else None,
                                      len(vals), _p(lens) if len(lens) else None, len(lens)))

    def set_exchange(self, mode):
        """"sync" (one minibatch at a time, zero staleness) or "overlap" (two in flight, staleness <= 1); between epochs"""
        _ck(lib().dfh_shard_set_exchange(self.h, {"sync": 0, "overlap": 1}[mode]))

    def reserve(self, batch_keys, recv_keys):
        """exchange buffers for these sizes now, so that no step re-allocates (after set_exchange)"""
        _ck(lib().dfh_shard_reserve(self.h, int(batch_keys), int(recv_keys)))

    def set_timing(self, on):
        _ck(lib().dfh_shard_set_timing(self.h, 1 if on else 0))

    def get_timing(self, reset=True):
        """-> ({stage: ms}, steps covered)"""
        ms = np.zeros(len(SHARD_STAGES), np.float64)
        n = C.c_uint64(0)
        _ck(lib().dfh_shard_get_timing(self.h, 1 if reset else 0, _p(ms), C.byref(n)))
        return dict(zip(SHARD_STAGES, ms.tolist())), n.value

    def owned_range(self):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        _ck(lib().dfh_shard_owned_range(self.h, _p(self.splits), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def close(self):
        if self.h:
            lib().dfh_shard_destroy(self.h)
            self.h = None
