#!/usr/bin/env python
"""bench.py — examples/sec of the FM/SGD worker step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torchrun)

A "step" is one pass of the hot path over one Criteo-shaped synthetic minibatch
(B rows x 39 slots, ids over 33 M features, V_dim=64, binary values) whose raw
CSR (u64 feature ids) is already resident in HBM:
    device Localizer::Compact -> Pull (table gather) -> FMLoss::Predict ->
    Evaluate -> FMLoss::CalcGrad (segmented sum) -> Push -> FTRL/AdaGrad in place.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10000, help="minibatch rows per GPU per step (criteo_sgd.conf:10)")
    ap.add_argument("--ids", type=int, default=33_000_000, help="feature id space")
    ap.add_argument("--vdim", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic batches cycled through")
    ap.add_argument("--no-prefill", action="store_true", help="start from an empty model instead of a warm one")
    ap.add_argument("--cpu-batches", type=int, default=-1, help="batches in the CPU baseline sample (-1: auto, 0: skip)")
    ap.add_argument("--no-timing", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-pipeline", action="store_true", help="prepare and train on one stream")
    ap.add_argument("--prep-streams", type=int, default=2,
                    help="preparation streams = minibatches localized ahead of the one training (sgd_learner.cc:219-223 "
                         "keeps 2 in flight)")
    ap.add_argument("--prep-lookup", action="store_true", help="resolve key->row on the preparation stream too")
    ap.add_argument("--uniform-ranges", action="store_true",
                    help="N>1: uniform key ranges (owner = key / ceil(2^64/N)) instead of ranges balanced on the id space")
    ap.add_argument("--exchange", choices=["sync", "overlap"], default="overlap",
                    help="N>1: two minibatches in flight with the exchange hidden behind compute (staleness 1, what the "
                         "reference's batch tracker does, sgd_learner.cc:219-223), or one at a time (zero staleness)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (key-range shards + RCCL all_to_all_v) even with one rank")
    return ap.parse_args()


TIMING_EVERY = int(os.environ.get("DFH_TIMING_EVERY", "4"))  # the forward kernel is timed on every n-th step of the timed region

HYPER = dict(l1=0.0, l2=0.0, V_l2=0.01, lr=0.01, lr_beta=1.0, V_lr=0.01, V_lr_beta=1.0, V_init_scale=0.01,
             V_threshold=0, seed=0)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; profiles/r01_pmc_hbm_traffic.json,
    collected with `rocprofv3 --pmc ... -- python bench.py` on the default workload), or None"""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        d = json.load(open(path))
        for name, v in d.items():
            if name.startswith(kernel):
                return v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(batches, V_dim, nbatches):
    """the reference CPU path (oracle/_ref: the reference's own Localizer/SGDUpdater/FMLoss
    compiled here) or the C port, timed on this box's host cores over `nbatches` batches"""
    from oracle import bindings as ob
    kind = "reference" if ob.have_ref() else "port"
    t_total = 0.0
    rows = 0
    if kind == "reference":
        R = ob.Ref()
        st = R.store_create(V_dim=V_dim, **HYPER)
        for b in batches[:nbatches]:
            t0 = time.perf_counter()
            loc = R.localize(b["offset"], b["index"], nthreads=2)
            st.push(loc["feaids"], ob.FEA_COUNT, loc["feacnt"])
            vals, lens = st.pull(loc["feaids"])
            # GetPos (sgd_learner.cc:113-127), vectorised
            ends = np.cumsum(lens)
            w_pos = (ends - lens).astype(np.int32)
            V_pos = np.where(lens > 1, w_pos + 1, -1).astype(np.int32)
            pred, grad = R.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], None, b["label"], vals, w_pos, V_pos)
            R.loss_evaluate(b["label"], pred)
            st.push(loc["feaids"], ob.GRADIENT, grad, lens)
            t_total += time.perf_counter() - t0
            rows += len(b["label"])
        cores = 2  # blk_nthreads_ = DEFAULT_NTHREADS (sgd_learner.h:90); the updater is single-threaded
    else:
        O = ob.Oracle()
        st = O.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **HYPER)
        for b in batches[:nbatches]:
            t0 = time.perf_counter()
            loc = O.localize(b["offset"], b["index"])
            st.sgd_step(loc["offset"], loc["index"], None, b["label"], loc["feaids"], feacnt=loc["feacnt"], is_train=True)
            t_total += time.perf_counter() - t0
            rows += len(b["label"])
        cores = 1
    return dict(value=rows / t_total, unit="examples/sec", cores=cores, kind=kind,
                sample="%d batches x %d rows of the same synthetic stream, model starting empty, "
                       "localize+pull+predict+calcgrad+push, no file I/O" % (nbatches, len(batches[0]["label"])))


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: re-launch under torch.distributed.run, one rank per GPU
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return subprocess.call(cmd)
    if args.gpus > 1 or world > 1 or args.force_sharded:
        from difacto_amd import sharded
        return sharded.bench_main(args, rank, world, local_rank, HYPER)

    import torch  # device plumbing only: barrier-equivalent sync + sanity that a GPU exists
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from difacto_amd import capi, synth
    from difacto_amd.build import build_hip
    build_hip()

    B, k = args.rows, args.vdim
    S = synth.NUM_SLOTS
    ctx = capi.Context(local_rank)
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    table = capi.Table(ctx, int(args.ids * 1.02) + 4 * B * S, V_dim=k, init_mode=capi.INIT_HASH, **HYPER)

    t0 = time.time()
    if not args.no_prefill:
        # warm model: every feature id present with an allocated V row, so every
        # gathered row moves its full (1+k)*4 bytes (worst-case traffic)
        for g in range(S):
            keys = synth.reverse_bytes_np(gen.all_ids(g))
            chunk = 1 << 22
            for o in range(0, len(keys), chunk):
                part = np.ascontiguousarray(keys[o:o + chunk])
                db = capi.DeviceBuffer.from_numpy(ctx, part)
                table.warm_start(db.ptr, len(part), w0=0.01, cnt0=100.0)
                ctx.sync()
                db.close()
    nkeys = table.size()
    t_prefill = time.time() - t0

    nd = max(1, min(args.distinct, args.steps + args.warmup))
    host_batches = [gen.batch(B) for _ in range(nd)]
    dev = []
    for hb in host_batches:
        off32 = hb["offset"].astype(np.uint32)
        dev.append((capi.DeviceBuffer.from_numpy(ctx, off32), capi.DeviceBuffer.from_numpy(ctx, hb["index"]),
                    capi.DeviceBuffer.from_numpy(ctx, hb["label"])))
    # depth+1 batch objects: batches t+1 .. t+depth are localized on the preparation streams while
    # batch t trains on the main stream (updates are still applied strictly in batch order)
    depth = 0 if args.no_pipeline else max(1, min(args.prep_streams, 4))
    ctx.set_pipeline(depth)
    ahead = max(depth, 1)
    # one spare object so that a new Localizer never waits for the step that just ended to release its buffers
    bts = [capi.Batch(ctx, B, B * S) for _ in range(ahead + (2 if depth else 1))]
    bt = bts[0]

    def prep(i):
        o, x, l = dev[i % nd]
        b = bts[i % len(bts)]
        b.attach_device(B, B * S, o.ptr, x.ptr, None, l.ptr)  # inputs are resident in HBM: no copy
        b.localize()
        if args.prep_lookup:
            b.lookup(table)

    def step(i):
        prep(i + ahead)
        bts[i % len(bts)].sgd_step(table, is_train=True, push_cnt=True)

    for i in range(ahead - 1):
        prep(i)
    prep(ahead - 1)
    for i in range(args.warmup):
        step(i)
    ctx.sync()
    torch.cuda.synchronize()
    for b in bts:
        b.progress(reset=True)
    # live timing of the dominant kernel inside the timed region: the k_forward dispatch of every
    # TIMING_EVERY-th step carries a start/stop HIP event pair (hipExtLaunchKernelGGL: the kernel's own
    # begin/end stamps, no marker packets).  Every step would cost the job 3 %, every 4th costs 1 %.
    fwd_mask = 0 if args.no_timing else ((1 << capi.K_FORWARD) | (1 << capi.K_BACKWARD))
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if fwd_mask and i % TIMING_EVERY == 0:
            ctx.set_timing_mask(fwd_mask)
            step(args.warmup + i)
            ctx.set_timing_mask(0)
        else:
            step(args.warmup + i)
    t_enqueued = time.perf_counter() - t0   # host side only: everything is queued, nothing awaited yet
    ctx.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    progs = [b.progress(reset=True) for b in bts]
    prog = capi.Progress()
    prog.loss = sum(p.loss for p in progs)
    prog.nrows = sum(p.nrows for p in progs)
    timing = {} if args.no_timing else ctx.get_timing(reset=True)
    # per-kernel breakdown from a separate, fully instrumented pass (NOT part of the timed region)
    breakdown = {}
    if not args.no_timing:
        nb_steps = min(args.steps, 50)
        ctx.set_timing(True)
        for i in range(nb_steps):
            step(args.warmup + args.steps + i)
        breakdown = {n: v[0] / nb_steps for n, v in ctx.get_timing(reset=True).items() if v[1] > 0}
        ctx.set_timing(False)
        for b in bts:
            b.progress(reset=True)
    _, _, U_last = bt.shape()

    ex_per_s = args.steps * B / dt
    r_g = S * (1 + k) * 4  # algorithmic gather bytes per example
    roofline = None
    if timing and timing["forward"][1] > 0:
        fwd_ms = timing["forward"][0] / timing["forward"][1]
        achieved = B * r_g / (fwd_ms * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="k_forward", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBPS, traffic=pmc_traffic("k_forward<"),
                        algorithmic_bytes_per_launch=B * r_g, avg_launch_ms=fwd_ms,
                        launches_timed=int(timing["forward"][1]))
    # the same for the longest kernel of the step, the fused backward/update (SURVEY 8d "full-step accounting":
    # XV re-read s*k*4 per example + state read and written (8+2k)*4*2 per unique key)
    roofline_bwd = None
    if timing and timing["backward"][1] > 0:
        bwd_ms = timing["backward"][0] / timing["backward"][1]
        bwd_bytes = B * S * k * 4 + int(U_last) * (8 + 2 * k) * 4 * 2
        ach = bwd_bytes / (bwd_ms * 1e-3) / 1e9
        roofline_bwd = dict(bound="hbm", kernel="k_backward_all", achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBPS, traffic=pmc_traffic("k_backward_all<"),
                            algorithmic_bytes_per_launch=bwd_bytes, avg_launch_ms=bwd_ms,
                            launches_timed=int(timing["backward"][1]))
    nb = args.cpu_batches
    cpu = None
    if nb != 0:
        if nb < 0:
            nb = max(2, min(nd, int(200000 / B)))  # ~20 batches of 10k rows: tens of seconds of CPU work
        cpu = cpu_baseline(host_batches, k, nb)

    out = {
        "metric": "examples/sec (FM SGD worker step, Criteo-shape, V_dim=%d)" % k,
        "value": ex_per_s, "unit": "examples/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C3: Criteo-shaped synthetic, %d ids / 39 slots, V_dim=%d, FTRL(w)+AdaGrad(V), 1 MI355X"
                               % (args.ids, k),
                   "rows_per_step": B, "nnz_per_row": S, "unique_keys_last_batch": int(U_last),
                   "step": "device localize + pull + predict + evaluate + calcgrad + push/update",
                   "model_keys": int(nkeys), "prefilled": not args.no_prefill, "hyper": HYPER,
                   "distinct_batches": nd, "pipelined_prep": not args.no_pipeline, "prep_streams": depth},
        "roofline": roofline,
        "roofline_backward": roofline_bwd,
        "cpu_baseline": cpu,
        "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
        "kernel_ms_per_step": breakdown,
        "kernel_ms_per_step_note": "separate instrumented pass after the timed region (HIP events around every kernel group)",
        "train_logloss_per_example": prog.loss / max(prog.nrows, 1),
        "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
        "prefill_seconds": t_prefill,
    }
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
