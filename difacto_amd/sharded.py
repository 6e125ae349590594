"""Multi-GPU FM/SGD worker path, Python side: key-range helpers and the N > 1 bench driver.

The exchange itself lives in libdifacto_hip.so (csrc/dfh_shard.hip: dfh_comm / dfh_shard / dfh_shard_step over RCCL
ncclSend / ncclRecv, sync or overlapped): one process per GPU, the model row-sharded by contiguous ranges of the
reversed keys — what ReverseBytes is for (include/difacto/base.h:29-38) — and because the Localizer emits a minibatch's
keys in ascending order (src/data/localizer.cc:28-48) each owner's keys are one contiguous slice of it.  The ranges are
either the uniform ones (owner = key / ceil(2^64/G)) or given by explicit split keys balanced on a sample
(balanced_splits here, dfh_shard_balanced_splits in the library): with feature-group ids in the low bits of an id
(EncodeFeaGrpID, base.h:60-63) the uniform split is badly skewed.

bench_main_native is what `bench.py --gpus N` runs.  The older torch.distributed transport of the same protocol is
test infrastructure: tests/sharded_harness.py.
"""
import collections
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

U64MAX = 2 ** 64 - 1


def key_span(world):
    """shard d owns reversed keys in [d*span, (d+1)*span)"""
    return U64MAX if world == 1 else U64MAX // world + 1


def uniform_splits(world):
    """first keys of shards 1 .. world-1 under the uniform partition"""
    return np.array([d * key_span(world) for d in range(1, world)], dtype=np.uint64)


def balanced_splits(sample_keys, world):
    """first keys of shards 1 .. world-1 such that every shard owns the same share of
    `sample_keys` (a sample of the id space, identical on every rank).  Feature-group ids
    live in the top bits of a reversed key (EncodeFeaGrpID + ReverseBytes, base.h:39-63), so
    the uniform partition of a 39-group id space gives one shard 2.3x the average at world 8."""
    s = np.sort(np.asarray(sample_keys, dtype=np.uint64))
    if world == 1 or len(s) == 0:
        return np.zeros(0, np.uint64)
    q = [(len(s) * d) // world for d in range(1, world)]
    return s[q].astype(np.uint64)


def owner_of(keys, splits):
    """shard index of every (reversed) key"""
    return np.searchsorted(np.asarray(splits, dtype=np.uint64), np.asarray(keys, dtype=np.uint64), side="right")


# --------------------------------------------------------------------------- bench (N > 1), native transport
XGMI_LINK_GBPS = 153.6   # per link, both directions together; 7 links per MI355X, one to every peer of the node


def bench_main_native(args, rank, world, local_rank, hyper, cpu_baseline_fn=None, pmc_traffic_fn=None):
    """bench.py --gpus N under torchrun, the exchange inside libdifacto_hip.so (dfh_shard_step over RCCL
    ncclSend / ncclRecv): weak scaling, every rank trains its own B-row minibatch per step against the
    key-range-sharded model, zero staleness.  torch.distributed (gloo) only carries the rendezvous id,
    the barriers around the timed region and the max-over-ranks of the time."""
    from . import capi, synth
    from .build import build_hip
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (no CPU fallback)")
    sys.stdout.flush()
    real_stdout = os.dup(1)   # RCCL prints a banner on stdout; the contract is ONE JSON line there
    os.dup2(2, 1)
    shared = os.environ.get("DFH_BENCH_BACKEND", "nccl") == "gloo"   # dry run: ranks share devices, host-staged exchange
    if shared:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        build_hip()
    dist.barrier()
    B, k, S = args.rows, args.vdim, synth.NUM_SLOTS
    hyper = dict(hyper)
    # a key present in every worker's minibatch receives `world` gradient pushes per step: the rates are divided by
    # the number of workers so that they move it about as far as one worker's push would.  No effect on throughput.
    hyper["lr"] = hyper["lr"] / world
    hyper["V_lr"] = hyper["V_lr"] / world
    ctx = capi.Context(local_rank)
    ctx.set_pipeline(1)
    if shared:
        def exchange(send, sb, recv, rb):
            out = torch.empty(sum(rb), dtype=torch.uint8)
            dist.all_to_all_single(out, torch.from_numpy(np.array(send, copy=True)), output_split_sizes=rb, input_split_sizes=sb)
            recv[:] = out.numpy()
        comm = capi.Comm.callback(ctx, rank, world, exchange)
    else:
        ids = [capi.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        comm = capi.Comm.rccl(ctx, rank, world, ids[0])
    # a bad rendezvous (a rank that never joins) fails here within a minute instead of hanging the first step
    comm.selfcheck(float(os.environ.get("DFH_SELFCHECK_TIMEOUT", "60")))
    comm_info = comm.info()
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    splits = None
    if world > 1 and not args.uniform_ranges:
        # balanced on the id space (= on the rows every shard holds): the split keys come out of the library
        # (dfh_shard_balanced_splits: quantiles of the union of the ranks' samples); every rank samples other slots
        sample = [synth.reverse_bytes_np(gen.all_ids(g))[::61] for g in range(S) if g % world == rank]
        splits = comm.balanced_splits(np.concatenate(sample) if sample else np.zeros(0, np.uint64))
    elif world > 1:
        splits = uniform_splits(world)
    owned = 0
    mine = []
    for g in range(S):
        keys = synth.reverse_bytes_np(gen.all_ids(g))
        mine.append(keys[owner_of(keys, splits) == rank] if world > 1 else keys)
        owned += len(mine[-1])
    table = capi.Table(ctx, int(owned * 1.05) + 8 * B * S, V_dim=k, init_mode=capi.INIT_HASH, **hyper)
    if not args.no_prefill:
        for m in mine:
            for o in range(0, len(m), 1 << 22):
                part = np.ascontiguousarray(m[o:o + (1 << 22)])
                db = capi.DeviceBuffer.from_numpy(ctx, part)
                table.warm_start(db.ptr, len(part), w0=0.01, cnt0=100.0)
                ctx.sync()
                db.close()
    del mine
    t_prefill = time.time() - t0
    shard = capi.Shard(table, comm, splits)
    if args.exchange == "overlap":
        shard.set_exchange("overlap")
    gen.rng = np.random.default_rng(1000 + rank)   # every rank draws its own stream (different data parts, sgd_learner.cc:78-89)
    nd = max(1, args.distinct)
    dev = []
    host_sample = []   # rank 0 keeps a few host batches for the CPU baseline
    for _ in range(nd):
        hb = gen.batch(B)
        if rank == 0 and len(host_sample) < 8:
            host_sample.append(hb)
        dev.append((capi.DeviceBuffer.from_numpy(ctx, hb["offset"].astype(np.uint32)), capi.DeviceBuffer.from_numpy(ctx, hb["index"]),
                    capi.DeviceBuffer.from_numpy(ctx, hb["label"])))
    bts = [capi.Batch(ctx, B, B * S) for _ in range(3)]
    if not getattr(args, "no_auc", False):
        for b_ in bts:
            b_.set_option("compute_auc", 1)   # BinClassMetric::AUC of every minibatch (sgd_learner.cc:153-155)

    def prep(i):
        o, x, l = dev[i % nd]
        b = bts[i % len(bts)]
        b.attach_device(B, B * S, o.ptr, x.ptr, None, l.ptr)
        b.localize()   # on the preparation stream, while the previous step's exchange runs

    def step(i):
        prep(i + 1)
        shard.prefetch_counts(bts[(i + 1) % len(bts)])   # the next step will not wait for its counts mid-way
        shard.step(bts[i % len(bts)], is_train=True, push_cnt=True)

    prep(0)
    done = 0
    for _ in range(args.warmup):
        step(done)
        done += 1
    ctx.sync()
    torch.cuda.synchronize()
    dist.barrier()
    for b in bts:
        b.progress(reset=True)
    mask = 0 if (args.no_timing or rank != 0) else (1 << capi.K_FORWARD)
    ctx.get_timing(reset=True)

    def region():
        """K steps between two barriers; the time is the maximum over the ranks (the same number on every rank)"""
        nonlocal done
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if mask and i % 4 == 0:
                ctx.set_timing_mask(mask)
                step(done)
                ctx.set_timing_mask(0)
            else:
                step(done)
            done += 1
        ctx.sync()
        torch.cuda.synchronize()
        dist.barrier()
        dt_t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        return float(dt_t.item())

    # the K-step region is repeated until min_time seconds are timed (every rank derives the same count from the
    # all-reduced time of the first region); the reported time is the median region
    comm.stats(reset=True)
    reps = [region()]
    more = int(min(max(np.ceil(args.min_time / max(reps[0], 1e-9)) - 1, 0), args.max_reps - 1))
    for _ in range(more):
        reps.append(region())
    dt = float(sorted(reps)[len(reps) // 2])
    x_sent, x_recv, x_groups = comm.stats(reset=True)
    x_steps = len(reps) * args.steps
    fwd_t = ctx.get_timing(reset=True).get("forward", (0.0, 0)) if mask else (0.0, 0)
    table.check()
    # per-stage device time from a separate, instrumented pass (NOT part of the timed region): HIP events around the
    # stages of dfh_shard_step on the stream each stage runs on; max over the ranks per stage
    stage_ms = None
    if not args.no_timing:
        shard.set_timing(True)
        shard.get_timing(reset=True)
        n_inst = min(args.steps, 40)
        for _ in range(n_inst):
            step(done)
            done += 1
        ms, n_cov = shard.get_timing(reset=True)
        shard.set_timing(False)
        names = list(capi.SHARD_STAGES)
        per = torch.tensor([ms[n] / max(n_cov, 1) for n in names], dtype=torch.float64)
        dist.all_reduce(per, op=dist.ReduceOp.MAX)
        stage_ms = {n: round(float(v), 5) for n, v in zip(names, per.tolist())}
    progs = [b.progress(reset=True) for b in bts]
    U_last = bts[(done - 1) % len(bts)].shape()[2]
    tot = comm.allreduce_sum([sum(p.loss for p in progs), sum(p.nrows for p in progs), float(U_last),
                              float(x_sent), float(x_recv)])
    cpu = None
    if rank == 0 and cpu_baseline_fn is not None and args.cpu_batches != 0:
        # the reference's CPU path on rank 0's host cores, on a bounded sample of rank 0's own stream (the other ranks
        # wait at the barrier below); the same function as the N = 1 line
        nb = args.cpu_batches if args.cpu_batches > 0 else max(2, min(len(host_sample), int((60000 if k <= 64 else 30000) / max(B, 1))))
        base_hyper = dict(hyper, lr=hyper["lr"] * world, V_lr=hyper["V_lr"] * world)
        cpu = cpu_baseline_fn(host_sample, k, min(nb, len(host_sample)), base_hyper)
    if rank == 0:
        ex_per_s = args.steps * B * world / dt
        r_g = S * (1 + k) * 4
        roofline = None
        if fwd_t[1] > 0:
            fwd_ms = fwd_t[0] / fwd_t[1]
            achieved = B * r_g / (fwd_ms * 1e-3) / 1e9
            tr, tr_src = pmc_traffic_fn(["k_forward<"], "sharded-w1") if pmc_traffic_fn else (None, None)
            roofline = dict(bound="hbm", kernel="k_forward (own keys read in the table, the others in the pulled rows; rank 0)",
                            achieved=achieved, peak=8000.0, unit="GB/s", frac=achieved / 8000.0, traffic=tr,
                            traffic_source=tr_src, traffic_note="counters of a committed ONE-rank run of this code path "
                            "(--force-sharded), not of this run" if tr else None,
                            algorithmic_bytes_per_launch=B * r_g, avg_launch_ms=fwd_ms, launches_timed=int(fwd_t[1]))
        # the exchange against the links: payload this GPU sent + received per step (keys and epoch-0 counts out, rows back,
        # gradient rows out; the same three for the keys it owns) over the device time of the K, RW and G stages
        roofline_x = None
        if True:   # one rank: zero bytes, zero time — the block is still there so that the line has one shape for every N
            per_gpu_step = (tot[3] + tot[4]) / world / max(x_steps, 1)
            x_ms = (stage_ms["K"] + stage_ms["RW"] + stage_ms["G"]) if stage_ms else None
            peak = 7 * XGMI_LINK_GBPS
            links = max(world - 1, 1)
            ach = per_gpu_step / (x_ms * 1e-3) / 1e9 if x_ms else None
            roofline_x = dict(bound="xgmi", stages="K + RW + G (device time of the three all-to-all-v stages, instrumented pass, max over ranks)",
                              bytes_per_gpu_step=per_gpu_step, exchange_ms_per_step=x_ms, achieved=ach, peak=peak, unit="GB/s",
                              frac=(ach / peak) if ach is not None else None, links_in_use=min(world - 1, 7),
                              frac_of_links_in_use=(ach / (links * XGMI_LINK_GBPS)) if ach is not None else None,
                              peak_note="7 xGMI links x 153.6 GB/s per GPU, both directions together; with N ranks N - 1 links carry traffic",
                              message_groups_per_step=x_groups / max(x_steps, 1))
        out = {
            "metric": "examples/sec (FM SGD worker step, Criteo-shape, V_dim=%d)" % k,
            "value": ex_per_s, "unit": "examples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "repetitions": len(reps),
            "ms_per_step_min": min(reps) / args.steps * 1e3, "ms_per_step_max": max(reps) / args.steps * 1e3,
            "value_note": "median of `repetitions` timed regions of `steps` steps each, every region's time = max over ranks",
            "config": {"workload": "C4: Criteo-shaped synthetic, %d ids / 39 slots, V_dim=%d, model row-sharded by key "
                                   "range over %d MI355X, RCCL ncclSend/ncclRecv all-to-all-v inside libdifacto_hip.so"
                                   % (args.ids, k, world),
                       "rows_per_step_per_gpu": B, "nnz_per_row": S, "parallelism": "shard%d" % world,
                       "step": "device localize + key/row/gradient all-to-all-v + predict + calcgrad + in-place update (dfh_shard_step)",
                       "avg_unique_keys_per_batch": tot[2] / world, "prefilled": not args.no_prefill, "hyper": hyper,
                       "dry_run_shared_gpu": shared,
                       "key_ranges": "uniform" if args.uniform_ranges else "balanced on the id space",
                       "exchange": ("overlap: two minibatches in flight inside dfh_shard_step (staleness <= 1 for rows of other "
                                    "owners, sgd_learner.cc:219-223), collectives on their own stream"
                                    if args.exchange == "overlap" else "sync: one minibatch at a time, zero staleness"),
                       "lr_scaled_by_world": {"lr": hyper["lr"], "V_lr": hyper["V_lr"], "divided_by": world,
                                              "why": "a key present in every worker's minibatch receives `world` pushes per step"},
                       "transport": "host callback over gloo (dry run, ranks share devices)" if shared else "RCCL ncclSend/ncclRecv",
                       "transport_bound": comm_info, "distinct_batches_per_rank": nd,
                       "auc_every_minibatch": not getattr(args, "no_auc", False),
                       "owned_keys_rank0": int(owned)},
            "stage_ms_per_step": stage_ms,
            "stage_ms_per_step_note": "separate instrumented pass; max over ranks per stage; counts/K/G/RW run on the "
                                      "collectives' stream in overlap mode and overlap F/R/P on the main stream",
            "roofline": roofline, "roofline_exchange": roofline_x, "cpu_baseline": cpu,
            "train_logloss_per_example": tot[0] / max(tot[1], 1.0),
            "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
            "prefill_seconds": t_prefill,
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    dist.barrier()
    for o_ in [shard] + bts + [table, comm]:
        o_.close()
    ctx.close()
    dist.destroy_process_group()
    return 0
