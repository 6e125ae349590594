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


def blended_splits(id_sample, data_sample, world, alpha):
    """first keys of shards 1 .. world-1 at the quantiles of the MIXTURE alpha * (keys the minibatches carry) + (1 - alpha) *
    (keys of the id space).  alpha = 0 balances the rows every shard holds, alpha = 1 the keys every owner receives per step
    (what dfh_shard_balanced_splits does with samples of the ranks' first minibatches, the C++ store's default).  The two
    differ a lot on Criteo-shaped data: the slot id sits in the top bits of a reversed key (EncodeFeaGrpID + ReverseBytes,
    base.h:39-63), the 13 integer slots hold 0.4 % of the ids and a third of a minibatch's keys, so ranges balanced on the
    ids alone give one owner 2x the average traffic and another 0.13x (8 ranks); balanced on the traffic alone they give
    one owner 30 % of the rows.  Identical on every rank when the samples are."""
    if world == 1:
        return np.zeros(0, np.uint64)
    a = np.sort(np.asarray(id_sample, dtype=np.uint64))
    b = np.sort(np.asarray(data_sample, dtype=np.uint64))
    if alpha <= 0 or len(b) == 0:
        return balanced_splits(a, world)
    if alpha >= 1 or len(a) == 0:
        return balanced_splits(b, world)
    pts = np.concatenate([a, b])
    wts = np.concatenate([np.full(len(a), (1.0 - alpha) / len(a)), np.full(len(b), alpha / len(b))])
    o = np.argsort(pts, kind="stable")
    pts, cw = pts[o], np.cumsum(wts[o])
    return np.array([pts[min(int(np.searchsorted(cw, d / world)), len(pts) - 1)] for d in range(1, world)], dtype=np.uint64)


def bench_splits(args, world, gen_factory, S):
    """the key ranges of the N > 1 bench lines (identical on every rank: the workers' streams are seeded by rank, so every
    rank can draw every worker's first minibatches itself).  --key-ranges data (default: balanced on the keys the minibatches
    carry), ids (on the id space), blend (--blend-alpha between the two; what a model near the HBM capacity needs), uniform."""
    from . import synth
    mode = "uniform" if getattr(args, "uniform_ranges", False) else getattr(args, "key_ranges", "data")
    if world == 1:
        return None, mode
    if mode == "uniform":
        return uniform_splits(world), mode
    alpha = {"data": 1.0, "ids": 0.0, "blend": float(getattr(args, "blend_alpha", 0.5))}[mode]
    g = gen_factory()
    step = 61 if int(g.vocab.sum()) <= 100_000_000 else 997
    ids = np.concatenate([synth.reverse_bytes_np(g.ids_of(q, np.arange(0, int(g.vocab[q]), step, dtype=np.uint64))) for q in range(S)])
    parts = []
    for p in range(world):
        gp = gen_factory()
        gp.rng = np.random.default_rng(5000 + p)   # not the timed streams: ranges come from a sample, as in production
        for _ in range(2):
            parts.append(np.unique(synth.reverse_bytes_np(gp.batch(args.rows)["index"])))
    data = np.concatenate(parts)
    splits = blended_splits(ids, data, world, alpha)
    # what the ranges mean for every owner: its share of the model's rows and of the keys that arrive per step
    args.range_shares = dict(rows=(np.bincount(owner_of(ids, splits), minlength=world) / max(len(ids), 1)).round(4).tolist(),
                             traffic=(np.bincount(owner_of(data, splits), minlength=world) / max(len(data), 1)).round(4).tolist())
    return splits, mode


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
    # transport start-up: a failure here (RCCL initialisation, a rank that never joins: dfh_comm_selfcheck polls for
    # DFH_SELFCHECK_TIMEOUT seconds) becomes a JSON line with "error" on rank 0's stdout and a non-zero exit, not a hang
    comm, err = None, None
    try:
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
        comm.selfcheck(float(os.environ.get("DFH_SELFCHECK_TIMEOUT", "60")))
    except Exception as ex:  # noqa: BLE001 — whatever the transport raised goes into the line
        err = "rank %d: %r" % (rank, ex)
    errs = [None] * world
    try:
        dist.all_gather_object(errs, err)
    except Exception as ex:  # noqa: BLE001 — the rendezvous itself is gone: every rank reports what it has
        errs = [err or ("rank %d: rendezvous lost: %r" % (rank, ex))]
    if any(errs):
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps({"metric": "examples/sec (FM SGD worker step, Criteo-shape, V_dim=%d)" % args.vdim, "value": None,
                              "unit": "examples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "error": "transport start-up failed: " + "; ".join(e for e in errs if e)}), flush=True)
        return 1
    comm_info = comm.info()
    # the wires, measured before anything else uses them: grouped ncclSend / ncclRecv all-to-all of 1 / 5 / 35 MB per peer on
    # the library's own communicator (the sizes of the K, RW / G exchanges at N = 8 and N = 2: DESIGN 6a)
    wire_probe = None
    if world > 1:
        sizes = [int(x) for x in os.environ.get("DFH_WIRE_PROBE_BYTES", "1000000,5000000,35000000").split(",") if x]
        wire_probe = dict(transport=comm_info, peers=world - 1, per_size=[],
                          note="every rank sends and receives `bytes_per_peer` to / from every other rank in ONE grouped exchange "
                               "(dfh_comm_wire_probe, %d timed repetitions after 2 untimed); GB/s per link and direction = "
                               "bytes_per_peer / time; max over ranks of the time" % 10)
        for nb in sizes:
            # a probe that fails on one rank (out of memory for the 2 x peers x bytes buffers, a transport error) must not
            # leave the others waiting in the reduction below: every rank takes part, the failure travels as a flag, the
            # remaining sizes are skipped and the measurement itself goes on
            try:
                us, bad = comm.wire_probe(nb, reps=10), 0.0
            except Exception as ex:  # noqa: BLE001
                us, bad = 0.0, 1.0
                wire_probe.setdefault("errors", []).append("rank %d, %d bytes per peer: %r" % (rank, nb, ex))
            t_ = torch.tensor([us, bad], dtype=torch.float64)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            us = float(t_[0].item())
            if float(t_[1].item()) > 0:
                wire_probe["per_size"].append(dict(bytes_per_peer=nb, us_per_grouped_exchange=None, gbps_per_link_and_direction=None,
                                                   error="the probe failed on at least one rank"))
                break
            wire_probe["per_size"].append(dict(bytes_per_peer=nb, us_per_grouped_exchange=us,
                                               gbps_per_link_and_direction=(nb / us / 1e3) if us > 0 else None))
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    splits, ranges_mode = bench_splits(args, world, lambda: synth.CriteoSynth(total_ids=args.ids, seed=42), S)
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
    shard.reserve(B * S, 2 * B * S)   # no buffer grows (stream drain + re-allocation) inside a step
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
    # minibatches are localized `ahead` steps before they train (--shard-prep-ahead, default 2): the overlapped exchange
    # sends the keys of minibatch t+1 during step t, so its Localizer has to be through when step t STARTS — localized only
    # one ahead (round 4) it ran beside step t's forward and the owners' pull for t+1 waited for it
    ahead = max(1, int(getattr(args, "shard_prep_ahead", 2)))
    bts = [capi.Batch(ctx, B, B * S) for _ in range(ahead + 2)]
    if not getattr(args, "no_auc", False):
        for b_ in bts:
            b_.set_option("compute_auc", 1)   # BinClassMetric::AUC of every minibatch (sgd_learner.cc:153-155)

    def prep(i):
        o, x, l = dev[i % nd]
        b = bts[i % len(bts)]
        b.attach_device(B, B * S, o.ptr, x.ptr, None, l.ptr)
        b.localize()   # on the preparation stream, while earlier steps' exchanges run

    def step(i):
        prep(i + ahead)
        shard.prefetch_counts(bts[(i + 1) % len(bts)])   # the next step will not wait for its counts mid-way
        shard.step(bts[i % len(bts)], is_train=True, push_cnt=True)

    for j in range(ahead):
        prep(j)
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
            if wire_probe and wire_probe["per_size"] and ach is not None:
                # ... and against what THIS node's wires gave the probe: the largest probed size, both directions of N - 1 links
                best = max((p_["gbps_per_link_and_direction"] or 0.0) for p_ in wire_probe["per_size"])
                if best > 0:
                    roofline_x["measured_wire_gbps_per_link_and_direction"] = best
                    roofline_x["frac_of_measured_wire"] = ach / (2.0 * links * best)
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
                       "key_ranges": ranges_mode,
                       "exchange": ("overlap: two minibatches in flight inside dfh_shard_step (staleness <= 1 for rows of other "
                                    "owners, sgd_learner.cc:219-223), collectives on their own stream"
                                    if args.exchange == "overlap" else "sync: one minibatch at a time, zero staleness"),
                       "lr_scaled_by_world": {"lr": hyper["lr"], "V_lr": hyper["V_lr"], "divided_by": world,
                                              "why": "a key present in every worker's minibatch receives `world` pushes per step"},
                       "transport": "host callback over gloo (dry run, ranks share devices)" if shared else "RCCL ncclSend/ncclRecv",
                       "transport_bound": comm_info, "distinct_batches_per_rank": nd,
                       "auc_every_minibatch": not getattr(args, "no_auc", False),
                       "owned_keys_rank0": int(owned)},
            "lr_divided_by_world": world,
            "wire_probe": wire_probe,
            "stage_ms_per_step": stage_ms,
            "stage_ms_per_step_note": "separate instrumented pass; max over ranks per stage; counts/K/G/RW run on the "
                                      "collectives' stream in overlap mode and overlap F/R/P on the main stream",
            "roofline": roofline, "roofline_exchange": roofline_x, "cpu_baseline": cpu,
            "train_logloss_per_example": tot[0] / max(tot[1], 1.0),
            "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
            "prefill_seconds": t_prefill,
        }
        # (the driver keeps the LAST 2 000 characters of the output beside the contract keys: the exchange's blocks go to the end)
        for k_ in ("stage_ms_per_step", "wire_probe", "roofline_exchange"):
            if k_ in out:
                out[k_] = out.pop(k_)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    dist.barrier()
    for o_ in [shard] + bts + [table, comm]:
        o_.close()
    ctx.close()
    dist.destroy_process_group()
    return 0


# --------------------------------------------------------------------------- projection: one GPU as rank r of W
XGMI_LINK_DIR_GBPS = XGMI_LINK_GBPS / 2   # one direction of one link
WIRE_MODELS = (("off", 0.0, 0.0), ("peak", XGMI_LINK_DIR_GBPS, 10.0), ("achievable", 0.6 * XGMI_LINK_DIR_GBPS, 20.0))


def _device_slot_keys(gen, g, r0, r1, device):
    """reversed keys of ranks [r0, r1) of slot g, generated ON THE DEVICE with torch int64 arithmetic (wrap-around like
    synth.splitmix64 / reverse_bytes_np; checked against them in tests/test_bench_contract.py): a 1e9-id space is too much
    for numpy on the host.  -> int64 tensor holding the u64 bit patterns"""
    def c(v):
        return torch.tensor(v - (1 << 64) if v >= (1 << 63) else v, dtype=torch.int64, device=device)

    def lsr(v, n):
        return (v >> n) & ((1 << (64 - n)) - 1)
    ranks = torch.arange(r0, r1, dtype=torch.int64, device=device)
    x = c(int(gen.seed)) ^ c(g << 40) ^ ranks
    x = x + c(0x9E3779B97F4A7C15)
    z = (x ^ lsr(x, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    h = z ^ lsr(z, 31)
    x = (h << 12) | g
    x = (x << 32) | lsr(x, 32)
    x = ((x & c(0x0000FFFF0000FFFF)) << 16) | lsr(x & c(0xFFFF0000FFFF0000), 16)
    x = ((x & c(0x00FF00FF00FF00FF)) << 8) | lsr(x & c(0xFF00FF00FF00FF00), 8)
    x = ((x & c(0x0F0F0F0F0F0F0F0F)) << 4) | lsr(x & c(0xF0F0F0F0F0F0F0F0), 4)
    return x


def _device_owned_keys(gen, lo, hi, device, chunk=1 << 25):
    """yields int64 tensors (u64 bit patterns) of the reversed keys of the whole id space that fall into [lo, hi)"""
    sign = 1 << 63
    lo_s = torch.tensor((lo ^ sign) - (1 << 64) if (lo ^ sign) >= sign else (lo ^ sign), dtype=torch.int64, device=device)
    hi1 = (hi - 1) ^ sign
    hi_s = torch.tensor(hi1 - (1 << 64) if hi1 >= sign else hi1, dtype=torch.int64, device=device)
    flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=device)
    for g in range(len(gen.vocab)):
        v = int(gen.vocab[g])
        for r0 in range(0, v, chunk):
            x = _device_slot_keys(gen, g, r0, min(v, r0 + chunk), device)
            xs = x ^ flip    # unsigned order as signed order
            sel = x[(xs >= lo_s) & (xs <= hi_s)]
            if sel.numel():
                yield sel


def _localized_keys(hb):
    """(ascending unique reversed keys, occurrence counts f32) of a host minibatch: Localizer::Compact's feaids / feacnt
    (src/data/localizer.cc:11-50), in numpy — only to SYNTHESISE what peers would send; the emulated rank's own
    minibatches go through the device Localizer like everywhere else"""
    from . import synth
    k, c = np.unique(synth.reverse_bytes_np(hb["index"]), return_counts=True)
    return k, c.astype(np.float32)


def bench_main_emulated(args, hyper, cpu_baseline_fn=None):
    """bench.py --emulate-world W [--emulate-rank r|all]: ONE GPU carries the load it would carry as rank r of a W-rank job
    (loop-back dfh_comm, include/difacto_hip.h): it trains its own B-row minibatches against the key range it would own
    (1/W of the balanced splits); the keys / counts it receives are those of the W - 1 other workers' minibatches (other
    streams of the same generator) restricted to that range, the rows it pulls for its remote keys come from a shadow buffer
    of valid rows, the gradient rows it receives are W - 1 peers' worth of valid gradient rows.  Everything
    dfh_shard_step launches at N = W runs at its real size on a quiet chip.  A PROJECTION, never the headline: the line says
    so; wire time is modelled (xGMI, one link per peer and direction) at the peak link rate and at a stated fraction of it."""
    import concurrent.futures
    from . import capi, synth
    from .build import build_hip
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    build_hip()
    W = args.emulate_world
    B, k, S = args.rows, args.vdim, synth.NUM_SLOTS
    hyper = dict(hyper)
    hyper["lr"] = hyper["lr"] / W
    hyper["V_lr"] = hyper["V_lr"] / W
    nd = max(2, min(args.distinct, 64))
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    all_keys = [synth.reverse_bytes_np(gen.all_ids(g)) for g in range(S)] if args.ids <= 40_000_000 else None
    splits, ranges_mode = bench_splits(args, W, lambda: synth.CriteoSynth(total_ids=args.ids, seed=42), S)
    shares = getattr(args, "range_shares", None)
    if args.emulate_rank == "all":
        ranks = list(range(W))
    elif args.emulate_rank == "auto":   # the owner that receives the most keys per step (ties: the most rows): the job's slowest rank
        tr, rw = shares["traffic"], shares["rows"]
        ranks = [max(range(W), key=lambda q: (round(tr[q], 3), rw[q]))]
    else:
        ranks = [int(args.emulate_rank)]
    # the W workers' streams (one generator object each: every worker draws its own data part, sgd_learner.cc:78-89).  Only the
    # emulated ranks' minibatches are kept as raw CSR; of the others only the localized key lists are needed.
    def stream(p):
        g = synth.CriteoSynth(total_ids=args.ids, seed=42)
        g.rng = np.random.default_rng(1000 + p)
        raw, loc = [], []
        for _ in range(nd):
            hb = g.batch(B)
            loc.append(_localized_keys(hb))
            raw.append(hb if p in ranks else None)
        return raw, loc
    with concurrent.futures.ThreadPoolExecutor(min(W, 8)) as pool:
        streams = list(pool.map(stream, range(W)))
    t_streams = time.time() - t0
    stride = capi.row_stride(k)
    results = []
    for r in ranks:
        results.append(_emulate_one_rank(args, r, W, splits, streams, all_keys, gen, hyper, stride, nd))
    if args.emulate_wire not in results[0]["models"]:   # DFH_EMUL_MODELS left it out
        args.emulate_wire = next(iter(results[0]["models"]))
    worst = max(results, key=lambda x: x["models"][args.emulate_wire]["ms_per_step"])
    headline = worst["models"][args.emulate_wire]
    proj = {m: W * B / (max(x["models"][m]["ms_per_step"] for x in results) * 1e-3) for m in worst["models"]}
    out = {
        "projection": True,
        "metric": "PROJECTED examples/sec of a %d-GPU job from ONE GPU carrying the load of rank r of %d (loop-back transport; "
                  "FM SGD worker step, Criteo-shape, V_dim=%d)" % (W, W, k),
        "value": proj[args.emulate_wire], "unit": "examples/sec", "n_gpus": 1, "emulated_world": W, "emulated_ranks": ranks,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": headline["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_note": "W x B / (slowest emulated rank's step) under the `%s` wire model; NOT a measurement of %d GPUs: every "
                      "kernel of dfh_shard_step ran at its real size on one quiet GPU, the wires are a model" % (args.emulate_wire, W),
        "projected_examples_per_sec": proj,
        "wire_models": {n: dict(link_gbps_per_direction=g, latency_us_per_exchange=l) for n, g, l in WIRE_MODELS},
        "wire_model_note": "an exchange holds its stream for latency + largest per-peer message / link rate (full mesh: one xGMI link "
                           "per peer, 153.6 GB/s per link both directions together = 76.8 per direction); `achievable` = 0.6 of that "
                           "+ 20 us per grouped ncclSend/ncclRecv exchange (assumed, not measured: no second GPU here)",
        "config": {"workload": "C4/C5 projection: Criteo-shaped synthetic, %d ids / 39 slots, V_dim=%d, model row-sharded by key range "
                               "over %d ranks, rank(s) %s emulated on 1 MI355X" % (args.ids, k, W, ranks),
                   "rows_per_step_per_gpu": B, "parallelism": "shard%d (emulated)" % W, "exchange": args.exchange, "hyper": hyper,
                   "distinct_batches_per_rank": nd, "key_ranges": ranges_mode, "range_shares_per_owner": shares,
                   "transport": "loop-back (device copies of the exact message sizes; sends read once, receives copied from fed buffers)"},
        "ranks": results,
        "streams_seconds": t_streams,
    }
    print(json.dumps(out))
    return 0


def _emulate_one_rank(args, r, W, splits, streams, all_keys, gen, hyper, stride, nd):
    from . import capi, synth
    B, k, S = args.rows, args.vdim, synth.NUM_SLOTS
    lo = 0 if r == 0 else int(splits[r - 1])
    hi = (1 << 64) if r == W - 1 else int(splits[r])
    ctx = capi.Context(0)
    ctx.set_pipeline(1)
    comm = capi.Comm.loopback(ctx, r, W)
    # this rank's shard of the model, pre-filled (every id of its range present with V, like the N = 1 line)
    t0 = time.time()
    owned = 0
    if all_keys is not None:
        mine = []
        for g in range(S):
            keys = all_keys[g]
            sel = keys[(keys >= np.uint64(lo)) & (keys <= np.uint64(hi - 1))]
            mine.append(sel)
            owned += len(sel)
        table = capi.Table(ctx, int(owned * 1.05) + 8 * B * S, V_dim=k, init_mode=capi.INIT_HASH, **hyper)
        for m in mine:
            for o in range(0, len(m), 1 << 22):
                part = np.ascontiguousarray(m[o:o + (1 << 22)])
                db = capi.DeviceBuffer.from_numpy(ctx, part)
                table.warm_start(db.ptr, len(part), w0=0.01, cnt0=100.0)
                ctx.sync()
                db.close()
        del mine
    else:   # an id space beyond the host's means (C5: 1e9 ids): the keys are generated on the device, twice (count, fill)
        dev_t = torch.device("cuda", 0)
        for sel in _device_owned_keys(gen, lo, hi, dev_t):
            owned += int(sel.numel())
        table = capi.Table(ctx, int(owned * 1.02) + 8 * B * S, V_dim=k, init_mode=capi.INIT_HASH, **hyper)
        for sel in _device_owned_keys(gen, lo, hi, dev_t):
            torch.cuda.synchronize()
            table.warm_start(sel.data_ptr(), int(sel.numel()), w0=0.01, cnt0=100.0)
            ctx.sync()
        torch.cuda.empty_cache()
    t_prefill = time.time() - t0
    shard = capi.Shard(table, comm, splits)
    if args.exchange == "overlap":
        shard.set_exchange("overlap")
    shard.reserve(B * S, 2 * B * S)   # no buffer grows (stream drain + re-allocation) inside a step
    raw, own_loc = streams[r]
    # per minibatch: what the W - 1 peers send this owner (their keys inside [lo, hi), ascending per peer, peer order)
    feeds = []
    n_in, n_out, u_own = [], [], []
    max_recv = 0
    for i in range(nd):
        cnt_words = np.zeros((W, 2), np.int64)
        ks, cs = [], []
        for p in range(W):
            if p == r:
                continue
            pk, pc = streams[p][1][i]
            a, b = np.searchsorted(pk, [np.uint64(lo), np.uint64(hi - 1)], side="left")
            b = b + (1 if b < len(pk) and int(pk[b]) == hi - 1 else 0)
            ks.append(pk[a:b])
            cs.append(pc[a:b])
            cnt_words[p] = (b - a, 1)
        keys = np.concatenate(ks) if ks else np.zeros(0, np.uint64)
        cnts = np.concatenate(cs) if cs else np.zeros(0, np.float32)
        max_recv = max(max_recv, len(keys))
        ok, _ = own_loc[i]
        a, b = np.searchsorted(ok, [np.uint64(lo), np.uint64(hi - 1)], side="left")
        b = b + (1 if b < len(ok) and int(ok[b]) == hi - 1 else 0)
        n_in.append(len(keys))
        n_out.append(len(ok) - (b - a))
        u_own.append(b - a)
        feeds.append((capi.DeviceBuffer.from_numpy(ctx, cnt_words), capi.DeviceBuffer.from_numpy(ctx, keys),
                      capi.DeviceBuffer.from_numpy(ctx, cnts)))
    max_U = max(len(x[0]) for x in own_loc)
    rng = np.random.default_rng(7 + r)
    # rows another owner would answer with: [w, has_V = 1, 0, 0 | V];  gradient rows the peers would push: [gw, had_V = 1, 0, 0 | gV]
    def shadow(n, a, b):
        m = np.zeros((n + 1024, stride), np.float32)
        m[:, 0] = rng.normal(size=len(m)) * a
        m[:, 1] = 1.0
        m[:, 4:4 + k] = rng.normal(size=(len(m), k)) * b
        return capi.DeviceBuffer.from_numpy(ctx, m)
    rows_shadow = shadow(max_U, 0.01, 0.01)
    grads_shadow = shadow(max_recv, 1e-3, 1e-4)
    comm.feed(capi.XCHG_ROWS, rows_shadow.ptr, sticky=True)
    comm.feed(capi.XCHG_GRADS, grads_shadow.ptr, sticky=True)
    dev = [(capi.DeviceBuffer.from_numpy(ctx, hb["offset"].astype(np.uint32)), capi.DeviceBuffer.from_numpy(ctx, hb["index"]),
            capi.DeviceBuffer.from_numpy(ctx, hb["label"])) for hb in raw]
    ahead = max(1, int(getattr(args, "shard_prep_ahead", 2)))   # (see bench_main_native)
    bts = [capi.Batch(ctx, B, B * S) for _ in range(ahead + 2)]
    if not getattr(args, "no_auc", False):
        for b_ in bts:
            b_.set_option("compute_auc", 1)
    fed = [0]

    def prep(i):
        o, x, l = dev[i % nd]
        b = bts[i % len(bts)]
        b.attach_device(B, B * S, o.ptr, x.ptr, None, l.ptr)
        b.localize()
        assert fed[0] == i
        c_, k_, f_ = feeds[i % nd]
        comm.feed(capi.XCHG_COUNTS, c_.ptr)
        comm.feed(capi.XCHG_KEYS, k_.ptr)
        comm.feed(capi.XCHG_CNT, f_.ptr)
        fed[0] += 1

    def step(i):
        prep(i + ahead)
        shard.prefetch_counts(bts[(i + 1) % len(bts)])
        shard.step(bts[i % len(bts)], is_train=True, push_cnt=True)

    for j in range(ahead):
        prep(j)
    done = 0
    for _ in range(args.warmup):
        step(done)
        done += 1
    ctx.sync()
    torch.cuda.synchronize()
    table.check()
    models = {}
    min_time = max(0.3, args.min_time / 3)
    stage = {}
    only = os.environ.get("DFH_EMUL_MODELS")   # e.g. "off": one wire model only (a kernel trace of just that case)
    for name, gbps, lat in WIRE_MODELS:
        if only and name not in only.split(","):
            continue
        comm.wire(gbps, lat)
        for _ in range(8):
            step(done)
            done += 1
        reps, t_all = [], 0.0
        comm.stats(reset=True)
        comm.wire_time_us(reset=True)
        while t_all < min_time and len(reps) < args.max_reps:
            ctx.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(done)
                done += 1
            ctx.sync()
            torch.cuda.synchronize()
            reps.append(time.perf_counter() - t0)
            t_all += reps[-1]
        dt = float(sorted(reps)[len(reps) // 2])
        sent, recv, groups = comm.stats(reset=True)
        nsteps = len(reps) * args.steps
        models[name] = dict(ms_per_step=dt / args.steps * 1e3, repetitions=len(reps),
                            examples_per_sec_this_gpu=args.steps * B / dt,
                            modelled_wire_ms_per_step=comm.wire_time_us(reset=True) / nsteps * 1e-3,
                            bytes_sent_per_step=sent / nsteps, bytes_recv_per_step=recv / nsteps, message_groups_per_step=groups / nsteps)
        if not args.no_timing:
            shard.set_timing(True)
            shard.get_timing(reset=True)
            n_inst = min(args.steps, 40)
            for _ in range(n_inst):
                step(done)
                done += 1
            ms, n_cov = shard.get_timing(reset=True)
            shard.set_timing(False)
            stage[name] = {n: round(ms[n] / max(n_cov, 1), 5) for n in capi.SHARD_STAGES}
    table.check()
    progs = [b.progress(reset=True) for b in bts]
    stride_b = stride * 4
    res = dict(rank=r, owned_keys=int(owned), key_range=[lo, hi - 1], prefill_seconds=t_prefill,
               unique_keys_per_batch=float(np.mean([len(x[0]) for x in own_loc])),
               own_keys_per_batch=float(np.mean(u_own)), remote_keys_out_per_batch=float(np.mean(n_out)),
               keys_in_per_batch=float(np.mean(n_in)),
               bytes_per_gpu_step=dict(K_out=float(np.mean(n_out)) * 12, K_in=float(np.mean(n_in)) * 12,
                                       RW_out=float(np.mean(n_in)) * stride_b, RW_in=float(np.mean(n_out)) * stride_b,
                                       G_out=float(np.mean(n_out)) * stride_b, G_in=float(np.mean(n_in)) * stride_b),
               models=models, stage_ms_per_step=stage,
               stage_ms_per_step_note="instrumented pass per wire model (HIP events around the stages on the stream each runs on); "
                                      "K / RW / G include the modelled wire wait when the model is on",
               train_logloss_per_example=sum(p.loss for p in progs) / max(sum(p.nrows for p in progs), 1))
    for o_ in [shard] + bts + [table, comm, rows_shadow, grads_shadow] + [x for f in feeds for x in f] + [x for d in dev for x in d]:
        o_.close()
    ctx.close()
    torch.cuda.empty_cache()
    return res
