#!/usr/bin/env python
"""Owner side of the sharded store alone, at the size rank 0 of an 8-rank C4 job sees (7 source lists of ~18 k keys each
from other workers' minibatches, data-balanced key ranges, a pre-filled shard): resolve_multi, push_count_multi,
pull_resolved, push_grad_multi in a loop.  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times
; prints HIP-event times of the groups itself.   usage: owner_bench.py [iters] [vdim] [world] [entry|listed]
(listed, round 6: resolve_multi, count_pull_multi, push_grad_listed — the owner side per distinct key, what dfh_shard_step runs)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    listed = (sys.argv[4] if len(sys.argv) > 4 else "listed") == "listed"
    from difacto_amd import capi, sharded, synth
    B, S, ids, r = 10000, synth.NUM_SLOTS, 33_000_000, 0

    class A:
        rows, key_ranges, blend_alpha = B, "data", 0.5
    splits, _ = sharded.bench_splits(A, W, lambda: synth.CriteoSynth(total_ids=ids, seed=42), S)
    lo, hi = 0, int(splits[0])
    gen = synth.CriteoSynth(total_ids=ids, seed=42)
    ctx = capi.Context(0)
    hyper = dict(l1=0.0, l2=0.0, V_l2=0.01, lr=0.01 / W, lr_beta=1.0, V_lr=0.01 / W, V_lr_beta=1.0, V_init_scale=0.01, V_threshold=0, seed=0)
    mine = [x[(x >= np.uint64(lo)) & (x < np.uint64(hi))] for x in (synth.reverse_bytes_np(gen.all_ids(g)) for g in range(S))]
    owned = sum(len(m) for m in mine)
    tb = capi.Table(ctx, int(owned * 1.05) + 8 * B * S, V_dim=k, init_mode=capi.INIT_HASH, **hyper)
    for m in mine:
        if len(m):
            db = capi.DeviceBuffer.from_numpy(ctx, np.ascontiguousarray(m))
            tb.warm_start(db.ptr, len(m), w0=0.01, cnt0=100.0)
            ctx.sync()
            db.close()
    stride = capi.row_stride(k)
    nd = 8
    sets = []
    rng = np.random.default_rng(3)
    for i in range(nd):
        ks, cs = [], []
        for p in range(1, W):
            g = synth.CriteoSynth(total_ids=ids, seed=42)
            g.rng = np.random.default_rng(1000 + 17 * i + p)
            pk, pc = sharded._localized_keys(g.batch(B))
            a, b = np.searchsorted(pk, [np.uint64(lo), np.uint64(hi)])
            ks.append(pk[a:b])
            cs.append(pc[a:b])
        seg = np.concatenate([[0, 0], np.cumsum([len(x) for x in ks])]).astype(np.int64)   # source 0 (the owner itself) sends nothing
        keys = np.concatenate(ks)
        n = len(keys)
        sets.append(dict(n=n, seg=seg, keys=capi.DeviceBuffer.from_numpy(ctx, keys), cnt=capi.DeviceBuffer.from_numpy(ctx, np.concatenate(cs)),
                         uniq=len(np.unique(keys))))
    nmax = max(s["n"] for s in sets)
    rowid = capi.DeviceBuffer(ctx, 4 * capi.multi_words(nmax, W))
    rows = capi.DeviceBuffer(ctx, 4 * stride * nmax)
    gm = np.zeros((nmax, stride), np.float32)
    gm[:, 0] = rng.normal(size=nmax) * 1e-3
    gm[:, 1] = 1.0
    gm[:, 4:4 + k] = rng.normal(size=(nmax, k)) * 1e-4
    grads = capi.DeviceBuffer.from_numpy(ctx, gm)
    ctx.set_timing(True)
    ctx.get_timing(reset=True)
    t0 = time.time()
    for it in range(iters):
        s = sets[it % nd]
        tb.shard_resolve_multi(s["keys"].ptr, s["seg"], rowid.ptr, 0)
        if listed:
            tb.shard_count_pull_multi(rowid.ptr, s["keys"].ptr, s["seg"], s["cnt"].ptr, rows.ptr, 0)
            tb.shard_push_grad_listed(rowid.ptr, s["keys"].ptr, s["seg"], grads.ptr, 0)
        else:
            tb.shard_push_count_multi(rowid.ptr, s["keys"].ptr, s["seg"], s["cnt"].ptr, 0)
            tb.shard_pull_resolved(rowid.ptr, s["n"], rows.ptr)
            tb.shard_push_grad_multi(rowid.ptr, s["keys"].ptr, s["seg"], grads.ptr, 0)
    ctx.sync()
    tb.check()
    tm = ctx.get_timing(reset=True)
    print("owner_bench (%s): W=%d k=%d  entries/step %.0f  unique %.0f  (HIP events, incl. launch) %s  wall %.1f us/iter"
          % ("per distinct key" if listed else "per entry", W, k, np.mean([s["n"] for s in sets]), np.mean([s["uniq"] for s in sets]),
             {n: round(v[0] / max(v[1], 1) * 1e3, 1) for n, v in tm.items() if v[1]}, (time.time() - t0) / iters * 1e6))


if __name__ == "__main__":
    main()
