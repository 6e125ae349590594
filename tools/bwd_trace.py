#!/usr/bin/env python
"""Per-block timeline of one k_backward_all launch (measurement build tools/var_trace.so, -DDFH_BWD_TRACE):
when does each role start and end inside the launch.  C3 batch, warm table, main kernels only."""
import ctypes as C, sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from difacto_amd import capi, synth
L = capi.lib()
ctx = capi.Context(0)
ids = 33_000_000
gen = synth.CriteoSynth(total_ids=ids, seed=42)
hyper = dict(l1=0.0, l2=0.0, V_l2=0.01, lr=0.01, V_lr=0.01, V_init_scale=0.01, V_threshold=0, seed=0)
tb = capi.Table(ctx, int(ids * 1.02) + 4 * 390000, V_dim=64, init_mode=capi.INIT_HASH, **hyper)
for g in range(39):
    keys = synth.reverse_bytes_np(gen.all_ids(g))
    for o in range(0, len(keys), 1 << 22):
        part = np.ascontiguousarray(keys[o:o + (1 << 22)])
        db = capi.DeviceBuffer.from_numpy(ctx, part)
        tb.warm_start(db.ptr, len(part)); ctx.sync(); db.close()
bts = []
for i in range(4):
    b = gen.batch(10000)
    bt = capi.Batch(ctx, 10000, 390000)
    bt.load_host(b["offset"], b["index"], None, b["label"]); bt.localize(); bts.append(bt)
for it in range(12):
    bts[it % 4].sgd_step(tb, is_train=True, push_cnt=True)
ctx.sync()
n = 3 * 8192
buf = np.zeros(n, np.uint64)
L.dfh_debug_bwd_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.dfh_debug_bwd_trace(buf.ctypes.data_as(C.c_void_p), n) == 0
t = buf.reshape(-1, 3)
t = t[t[:, 2] > 0]
t0 = t[:, 1].min()
us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0  # wall_clock64: 100 MHz
print("blocks traced", len(t), "kernel span %.1f us" % us(t[:, 2]).max())
for role, name in ((0, "hot"), (1, "mid"), (2, "small")):
    m = t[:, 0] == role
    if not m.any(): continue
    s, e = us(t[m, 1]), us(t[m, 2])
    d = e - s
    print("%-5s blocks %4d  start min/med/max %.1f/%.1f/%.1f  end med/p90/max %.1f/%.1f/%.1f  duration med/p90/max %.1f/%.1f/%.1f"
          % (name, m.sum(), s.min(), np.median(s), s.max(), np.median(e), np.percentile(e, 90), e.max(), np.median(d), np.percentile(d, 90), d.max()))
# resident blocks over time
ev = np.concatenate([np.stack([us(t[:, 1]), np.ones(len(t))], 1), np.stack([us(t[:, 2]), -np.ones(len(t))], 1)])
ev = ev[np.argsort(ev[:, 0])]
res = np.cumsum(ev[:, 1])
for T in range(0, int(us(t[:, 2]).max()) + 1, 5):
    i = np.searchsorted(ev[:, 0], T, side="right") - 1
    print("t=%3d us resident blocks %d" % (T, res[i] if i >= 0 else 0))
