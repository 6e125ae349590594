#!/usr/bin/env python
"""bench.py — examples/sec of the FM/SGD worker step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N=1 directly; N>1 under torchrun)

A "step" is one pass of the hot path over one synthetic minibatch whose raw CSR (u64 feature ids)
is already resident in HBM:
    device Localizer::Compact -> Pull (table gather) -> FMLoss::Predict -> Evaluate ->
    FMLoss::CalcGrad (segmented sum) -> Push -> FTRL/AdaGrad in place.
The default workload is BASELINE.json's C3 (Criteo-shaped: B rows x 39 slots, 33 M ids, V_dim 64,
binary values).  --preset selects the other single-GPU lines of SURVEY.md 8(d):
    c3-refdefaults   C3 with the reference's default hyper-parameters (l1 = 1, V_threshold = 10)
    c5-slice         one GPU's share of C5: V_dim 128, l1 = 1, as many ids as fit the HBM (~2e8)
    c2               rcv1-shaped: 100 rows x ~75 real-valued features over 47 236 ids, V_dim 8
The timed region of K steps (sync + barrier on both sides) is repeated until at least --min-time
seconds have been measured; the line reports the MEDIAN repetition and the spread.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement".
"""
import argparse
import glob
import json
import os

# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument`
# otherwise); the boxes export it already, a hand-made environment may not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# steady-state worst-case traffic: every touched key carries V
HYPER = dict(l1=0.0, l2=0.0, V_l2=0.01, lr=0.01, lr_beta=1.0, V_lr=0.01, V_lr_beta=1.0, V_init_scale=0.01,
             V_threshold=0, seed=0)
PRESETS = {
    "c3": dict(),
    # src/sgd/sgd_param.h:95-105
    "c3-refdefaults": dict(hyper=dict(l1=1.0, V_threshold=10)),
    "c5-slice": dict(vdim=128, ids=200_000_000, hyper=dict(l1=1.0)),
    # (single_queue: minibatches this small are launch latency, not throughput — the Localizer's stages ride in the step's own three
    # launches instead of taking five launches and two event hand-overs on a second queue: the same device time, half the host
    # time per step, which is what bounds this preset)
    "c2": dict(vdim=8, rows=100, ids=47_236, hyper=dict(l1=1.0, lr=0.1), single_queue=True),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preset", choices=sorted(PRESETS), default="c3")
    ap.add_argument("--rows", type=int, default=None, help="minibatch rows per GPU per step (criteo_sgd.conf:10: 10000)")
    ap.add_argument("--ids", type=int, default=None, help="feature id space (33 M)")
    ap.add_argument("--vdim", type=int, default=None)
    ap.add_argument("--l1", type=float, default=None)
    ap.add_argument("--v-threshold", type=int, default=None)
    ap.add_argument("--distinct", type=int, default=256,
                    help="distinct synthetic batches cycled through (independent of --steps; SURVEY 8d asks for a stream of "
                         "1 000: 256 x 390 000 ids already revisits a batch only every ~30 ms of device time)")
    ap.add_argument("--bias-slots", type=int, default=0,
                    help="diagnostic (NOT the metric): the first n slots carry ONE id in every row (a bias-like feature, the token of a "
                         "missing value): keys with B occurrences per minibatch, the update kernel's longest segments")
    ap.add_argument("--head-share", type=float, default=1.0,
                    help="with --bias-slots: the share of a slot's rows that carry its one id (1.0: every row; 0.25 over 13 slots is "
                         "the head of an integer feature of real click logs: thirteen keys of ~2 500 occurrences per minibatch)")
    ap.add_argument("--no-auc", action="store_true",
                    help="A/B: leave BinClassMetric::AUC out of the step (the reference computes it for every minibatch, "
                         "sgd_learner.cc:153-155; the default step does too)")
    ap.add_argument("--min-time", type=float, default=3.0, help="repeat the K-step region until this many seconds are timed")
    ap.add_argument("--max-reps", type=int, default=20000)
    ap.add_argument("--no-secondary", action="store_true",
                    help="N=1, preset c3: skip the `secondary` block (c3-refdefaults and c5-slice, each run as a child process "
                         "after the headline measurement has released the GPU)")
    ap.add_argument("--secondary-min-time", type=float, default=1.0)
    ap.add_argument("--no-prefill", action="store_true", help="start from an empty model instead of a warm one")
    ap.add_argument("--cpu-batches", type=int, default=-1, help="batches in the CPU baseline sample (-1: auto, 0: skip)")
    ap.add_argument("--no-timing", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-pipeline", action="store_true", help="prepare and train on one stream")
    ap.add_argument("--prep-streams", type=int, default=1,
                    help="preparation streams = minibatches localized ahead of the one training (sgd_learner.cc:219-223 "
                         "keeps 2 in flight)")
    ap.add_argument("--single-queue", dest="single_queue", action="store_true", default=None,
                    help="the single-queue step (csrc/dfh_riders.hip): no preparation stream; the Localizer's stages of the next "
                         "minibatches ride as extra blocks of this step's own three launches")
    ap.add_argument("--two-queues", dest="single_queue", action="store_false",
                    help="rounds 2-5: Localizer + probe of minibatch t+1 on a low-priority preparation stream")
    ap.add_argument("--ahead", type=int, default=0,
                    help="minibatches prepared ahead of the one training.  Single-queue step: default 2 (count(t+2) rides in the "
                         "update launch of step t); two queues: default = the number of preparation streams")
    ap.add_argument("--no-prep-lookup", dest="prep_lookup", action="store_false",
                    help="probe the key index inside the step instead of on the preparation stream")
    ap.add_argument("--fused-probe", action="store_true",
                    help="A/B: probe the key index inside the Localizer's emit pass (dfh_localize_lookup) instead of in a launch of "
                         "its own (dfh_batch_lookup); measured 0.4 %% slower: the longer emit pass runs beside the forward")
    ap.add_argument("--later-epoch", action="store_true",
                    help="time the step of epochs after the first (no feature-count push, sgd_learner.cc:214-217) instead of "
                         "the default epoch-0 step, which pushes counts every minibatch (the worst case)")
    ap.add_argument("--ctx-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: validated launch tuning passed to dfh_ctx_set_option (fwd_depth, bwd_small_blocks, ...)")
    ap.add_argument("--no-relocalize", action="store_true",
                    help="diagnostic (NOT the metric): localize every batch object once and time lookup + forward + backward alone")
    ap.add_argument("--uniform-ranges", action="store_true",
                    help="N>1: uniform key ranges (owner = key / ceil(2^64/N)); same as --key-ranges uniform")
    ap.add_argument("--key-ranges", choices=["data", "ids", "blend", "uniform"], default=None,
                    help="N>1: what the owners' key ranges balance — the keys the minibatches carry (`data`: every owner receives "
                         "the same number of keys per step; the default, and what the C++ store's shard_ranges = balanced does), "
                         "the rows every shard holds (`ids`), a mixture (`blend`, --blend-alpha; the default of the c5 preset, "
                         "whose model is near the HBM capacity), or the key space (`uniform`)")
    ap.add_argument("--blend-alpha", type=float, default=0.5)
    ap.add_argument("--exchange", choices=["sync", "overlap"], default="overlap",
                    help="N>1: two minibatches in flight with the exchange hidden behind compute (staleness <= 1, what the "
                         "reference's batch tracker does, sgd_learner.cc:219-223; the default), or one at a time (zero staleness)")
    ap.add_argument("--transport", choices=["native", "torch"], default="native",
                    help="N>1: the exchange inside libdifacto_hip.so (dfh_shard_step, RCCL ncclSend/ncclRecv; the product path) or "
                         "the test harness over torch.distributed (tests/sharded_harness.py)")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="W",
                    help="PROJECTION (never the headline): this ONE GPU carries the load of rank r of a W-rank job through the "
                         "loop-back transport (peers' keys / counts / gradient rows synthesised from W - 1 other streams of the "
                         "generator restricted to the rank's key range; wires modelled) — every kernel of dfh_shard_step at N = W "
                         "at its real size on a quiet chip")
    ap.add_argument("--emulate-rank", default="auto",
                    help="which rank(s) to emulate: an index, `auto` (the owner that receives the most keys per step: the job's "
                         "slowest rank) or `all` (one after the other; the projection takes the slowest, like a job's barrier would)")
    ap.add_argument("--emulate-wire", choices=["off", "peak", "achievable"], default="achievable",
                    help="which wire model the projected `value` is quoted under (all three are in the line)")
    ap.add_argument("--shard-prep-ahead", type=int, default=2,
                    help="N>1: minibatches are localized this many steps before they train (2: the keys of minibatch t+1 are ready "
                         "when step t starts and travel at once; 1: round 4's loop)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the N>1 code path (key-range shards + RCCL all_to_all_v) even with one rank")
    args = ap.parse_args()
    pre = PRESETS[args.preset]
    args.rows = args.rows if args.rows is not None else pre.get("rows", 10000)
    args.ids = args.ids if args.ids is not None else pre.get("ids", 33_000_000)
    args.vdim = args.vdim if args.vdim is not None else pre.get("vdim", 64)
    hyper = dict(HYPER)
    hyper.update(pre.get("hyper", {}))
    if args.l1 is not None:
        hyper["l1"] = args.l1
    if args.v_threshold is not None:
        hyper["V_threshold"] = args.v_threshold
    args.hyper = hyper
    if args.single_queue is None:
        args.single_queue = bool(pre.get("single_queue", False))
    if args.key_ranges is None:
        args.key_ranges = "blend" if args.preset == "c5-slice" else "data"
    return args


TIMING_EVERY = int(os.environ.get("DFH_TIMING_EVERY", "8"))  # forward/backward are timed on every n-th step of the timed region


def pmc_traffic(kernels, preset):
    """(HBM bytes per launch, source file) of the first of `kernels` found in the newest committed rocprofv3 PMC
    passes of this preset (FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; profiles/rNN_pmc_hbm_traffic[_<preset>].json,
    collected with `rocprofv3 --pmc ... -- python bench.py`, NOT measured in this run), or (None, None)"""
    suffix = "" if preset in ("c3", "c3-refdefaults") else "_" + preset.replace("-", "_")   # "sharded-w1": the 1-rank sharded path
    if isinstance(kernels, str):
        kernels = [kernels]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic%s.json" % suffix)), reverse=True):
        try:
            d = json.load(open(path))
            for kernel in kernels:
                for name, v in d.items():
                    if name.startswith(kernel):
                        return v["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError, AttributeError):
            continue
    return None, None


def requests_profile(pipelined):
    """(memory-side requests per step, source file) from the newest committed counter pass of the default step
    (profiles/rNN_requests_{pipelined,serial}.json: rocprofv3 --pmc TCC_EA0_RDREQ_sum / TCC_EA0_WRREQ_sum, tools/requests_table.py),
    and the request-rate ceiling of the access pattern (tools/fresh_bench.hip, the newest profiles/rNN_fresh_rows_bench.txt: the
    best of random 256 B row gathers [2 line reads per row] and 512 B row read-modify-writes with their 16 B header [5 line
    reads + 9 writes per row] on rows that are NOT in the memory-side cache).  Counters of committed profiles, NOT of this run."""
    import re
    req = src = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_requests_%s.json" % ("pipelined" if pipelined else "serial"))), reverse=True):
        try:
            req, src = json.load(open(path))["requests_per_step"], os.path.relpath(path, ROOT)
            break
        except (OSError, ValueError, KeyError):
            continue
    ceil = csrc = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fresh_rows_bench.txt")), reverse=True):
        try:
            best = 0.0
            for line in open(path):
                if line.startswith("nsets 1 "):
                    break   # (the second half repeats ONE index set: cached rows, not the ceiling)
                m = re.match(r"(gather256|rmw\+hdr)\s+blocks\s+\d+\s*:\s*([0-9.]+) us", line)
                if m:
                    n = 390000 * 2 if m.group(1) == "gather256" else 147000 * 14
                    best = max(best, n / (float(m.group(2)) * 1e-6))
            if best:
                ceil, csrc = best, os.path.relpath(path, ROOT)
                break
        except (OSError, ValueError):
            continue
    return req, src, ceil, csrc


def secondary_lines(args):
    """the other single-GPU lines of SURVEY 8(d), each by a child process of this same file (own table, own
    measurement, own roofline blocks): c3 with the reference's default hyper-parameters and one GPU's share of C5
    (V_dim 128).  Run after the headline measurement has released its 21 GB; a failure is reported, not fatal."""
    import subprocess
    out = {}
    # (name, arguments): c3-cold = the FIRST 256 steps on an empty table (a real first epoch: index insert + zero row + lazy
    # InitV land in a key's first step); c2 = the reference's quick-start shape with its own CPU baseline
    runs = [("c3-refdefaults", ["--preset", "c3-refdefaults", "--cpu-batches", "0"]),
            ("c5-slice", ["--preset", "c5-slice", "--cpu-batches", "0"]),
            ("c3-cold", ["--preset", "c3", "--no-prefill", "--steps", "256", "--warmup", "0", "--max-reps", "1", "--min-time", "0",
                         "--cpu-batches", "0", "--distinct", "256"]),
            ("c2", ["--preset", "c2", "--cpu-batches", "100"])]
    for name, extra in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--min-time", str(args.secondary_min_time), "--no-secondary"] + extra   # (later flags win: c3-cold sets its own steps)
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            keep = ("value", "unit", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "repetitions", "timed_region_s_total",
                    "dominant_kernel", "roofline", "roofline_backward", "roofline_step", "kernel_ms_per_step", "steps", "warmup")
            e = {k: d.get(k) for k in keep}
            if name == "c2":
                e["cpu_baseline"] = d.get("cpu_baseline")
            e["config"] = {k: d["config"].get(k) for k in ("workload", "rows_per_step", "unique_keys_per_batch", "model_keys",
                                                           "table_bytes", "hyper", "prefilled", "single_queue",
                                                           "minibatches_prepared_ahead")}
            e["wall_seconds"] = time.time() - t0
            out[name] = e
        except Exception as ex:  # noqa: BLE001 — the headline line must survive anything here
            out[name] = dict(error=repr(ex)[:300], wall_seconds=time.time() - t0)
    return out


def roofline_requests(args, s_per_step):
    """the step against the rate at which the chip serves random memory-side requests (every random read is a
    128 B line whatever it uses, writes are 32 / 64 B requests; the update's mix of the two is served at ~44 G requests/s)"""
    if args.preset not in ("c3", "c3-refdefaults") or args.no_relocalize:
        return None
    req, src, ceil, csrc = requests_profile(not args.no_pipeline)
    if not req or not ceil:
        return None
    per_s = req / s_per_step
    return dict(bound="memory-side requests", per_step=req, per_s=per_s, ceiling_per_s=ceil, frac=per_s / ceil,
                per_step_source=src, ceiling_source=csrc,
                note="requests of a committed counter pass of this step (not of this run) over this run's step time; ceiling = best "
                     "request rate of tools/fresh_bench.hip's patterns on fresh rows")


def host_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the CPU time the container may use (cgroup v2 cpu.max "quota period"): the GPU boxes of this build allow 16 cores' worth
    # of a 256-thread host, and threads beyond the quota only get the process throttled
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(period) > 0:
            quota = max(1, -(-int(q) // int(period)))
    except (OSError, ValueError):
        pass
    return dict(nproc=os.cpu_count(), cpu_quota=quota, cpu_model=model)


def cpu_baseline(batches, V_dim, nbatches, hyper, with_auc=True):
    """the reference CPU path (oracle/_ref: the reference's own Localizer / SGDUpdater / FMLoss compiled
    here) — or the C restatement when that build is absent — timed on this box's host cores.  The
    sample is `nbatches` batches of the GPU run's own stream; a first, untimed pass over them fills
    the model (every touched key gets its entry, state and V row, like the GPU's pre-filled table),
    the timed pass is the second.  Two thread settings: as shipped (2 OpenMP threads in
    the loss and the Localizer, single-threaded updater) and scaled to the host (min(nproc, the container's CPU quota, 49))."""
    from oracle import bindings as ob
    info = host_info()
    kind = "reference" if ob.have_ref() else "port"
    B = len(batches[0]["label"])
    sample = ("second pass over %d batches x %d rows of the GPU run's synthetic stream (the first, untimed pass fills the "
              "model); localize + count push + pull + predict + calcgrad + push, no file I/O" % (nbatches, B))
    if kind == "port":
        O = ob.Oracle()
        st = O.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **hyper)
        t_total, rows = 0.0, 0
        for timed in (False, True):
            for b in batches[:nbatches]:
                t0 = time.perf_counter()
                loc = O.localize(b["offset"], b["index"])
                st.sgd_step(loc["offset"], loc["index"], b["value"], b["label"], loc["feaids"], feacnt=loc["feacnt"], is_train=True)
                if timed:
                    t_total += time.perf_counter() - t0
                    rows += B
        return dict(value=rows / t_total, unit="examples/sec", cores=1, kind=kind, sample=sample, **info)
    R = ob.Ref()
    out = None
    for label, nthreads in (("as_shipped", 2), ("scaled", max(2, min(info["nproc"] or 2, info["cpu_quota"] or 49, 49)))):
        st = R.store_create(V_dim=V_dim, **hyper)
        stage = dict(localize=0.0, push_count=0.0, pull=0.0, predict_calcgrad=0.0, evaluate_auc=0.0, push_grad=0.0)
        rows = 0
        for timed in (False, True):
            for b in batches[:nbatches]:
                ts = [time.perf_counter()]
                loc = R.localize(b["offset"], b["index"], nthreads=nthreads)
                ts.append(time.perf_counter())
                st.push(loc["feaids"], ob.FEA_COUNT, loc["feacnt"])
                ts.append(time.perf_counter())
                vals, lens = st.pull(loc["feaids"])
                # GetPos (sgd_learner.cc:113-127), vectorised
                ends = np.cumsum(lens)
                w_pos = (ends - lens).astype(np.int32)
                V_pos = np.where(lens > 1, w_pos + 1, -1).astype(np.int32)
                ts.append(time.perf_counter())
                pred, grad = R.fm_predict_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], vals, w_pos,
                                                   V_pos, nthreads=nthreads)
                ts.append(time.perf_counter())
                R.loss_evaluate(b["label"], pred)
                if with_auc:
                    R.auc_times_n(b["label"], pred)   # BinClassMetric::AUC, sgd_learner.cc:153-155
                ts.append(time.perf_counter())
                st.push(loc["feaids"], ob.GRADIENT, grad, lens)
                ts.append(time.perf_counter())
                if timed:
                    for name, a, c in zip(stage, ts[:-1], ts[1:]):
                        stage[name] += c - a
                    rows += B
        total = sum(stage.values())
        res = dict(value=rows / total, threads=nthreads, stage_ms_per_batch={n: round(v / nbatches * 1e3, 3) for n, v in stage.items()})
        if out is None:
            # blk_nthreads_ = DEFAULT_NTHREADS = 2 (sgd_learner.h:90); the updater is single-threaded
            out = dict(value=res["value"], unit="examples/sec", cores=2, kind=kind, sample=sample,
                       stage_ms_per_batch=res["stage_ms_per_batch"], **info)
        else:
            out["scaled_threads"] = res
    out["note"] = ("compute only: the reference's worker loop adds 10 ms sleep-polls (sgd_learner.cc:93,221) that are not "
                   "reproduced here; Loss::Evaluate and BinClassMetric::AUC of every minibatch %s on both sides"
                   % ("included" if with_auc else "(AUC) excluded"))
    return out


def rcv1_shaped_batches(rng, n, rows, ids):
    """C2: the shape of example/rcv1_sgd.conf — ~75 real-valued features per row over 47 236 ids, batch 100"""
    out = []
    for _ in range(n):
        lens = np.clip(rng.poisson(75, size=rows), 1, None)
        off = np.zeros(rows + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        nnz = int(off[-1])
        idx = (rng.zipf(1.2, size=nnz) % ids + 1).astype(np.uint64)
        val = np.abs(rng.normal(size=nnz) * 0.1).astype(np.float32)
        lab = np.where(rng.random(rows) < 0.47, 1.0, -1.0).astype(np.float32)
        out.append(dict(offset=off, index=idx, value=val, label=lab))
    return out


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
    if args.emulate_world > 1:
        from difacto_amd import sharded
        return sharded.bench_main_emulated(args, args.hyper)
    if args.gpus > 1 or world > 1 or args.force_sharded:
        if args.transport == "native":
            from difacto_amd import sharded
            return sharded.bench_main_native(args, rank, world, local_rank, args.hyper,
                                             cpu_baseline_fn=lambda hb, k_, nb_, hy_: cpu_baseline(hb, k_, nb_, hy_, with_auc=not args.no_auc),
                                             pmc_traffic_fn=pmc_traffic)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sharded_harness   # the torch.distributed transport: test infrastructure, not the product path
        return sharded_harness.bench_main(args, rank, world, local_rank, args.hyper)

    import torch  # device plumbing only: barrier-equivalent sync + sanity that a GPU exists
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    from difacto_amd import capi, synth
    from difacto_amd.build import build_hip
    build_hip()

    B, k, hyper = args.rows, args.vdim, args.hyper
    criteo = args.preset != "c2"
    ctx = capi.Context(local_rank)
    if args.preset == "c5-slice" and not args.no_prefill:
        # one GPU's share of C5 (1e9 ids over 8 GPUs = 1.25e8 rows each; SURVEY 8e "Capacity"): as many rows as
        # fit, leaving room for the batches and the runtime.  per row: 2*kp*4 B V+acc, 32 B header, ~32 B index
        free_b, _total_b = torch.cuda.mem_get_info()
        per_row = 2 * ((k + 3) // 4 * 4) * 4 + 32 + 2 * 16 * 2
        fit = int((free_b - (8 << 30)) / per_row)
        if fit < args.ids:
            args.ids = max(fit, 1_000_000)
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42) if criteo else None
    S = synth.NUM_SLOTS if criteo else 0
    nd = max(1, args.distinct)
    # the synthetic stream is drawn by a helper thread (numpy releases the GIL) while this one fills the model
    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(1)
    if criteo:
        capacity = int(args.ids * 1.02) + 4 * B * S
        bgen = synth.CriteoSynth(total_ids=args.ids, seed=42)   # its own generator object: `gen` serves the prefill

        def draw():
            hb = bgen.batch(B)
            for g in range(min(args.bias_slots, S)):   # (--bias-slots: one id in every row of the slot, or in --head-share of them)
                col = hb["index"].reshape(B, S)[:, g]
                head = bgen.ids_of(g, np.zeros(1, np.int64))[0]
                if args.head_share >= 1.0:
                    col[:] = head
                else:
                    col[bgen.rng.random(B) < args.head_share] = head
            return hb
        fut_batches = pool.submit(lambda: [draw() for _ in range(nd)])
    else:
        capacity = int(args.ids * 1.5) + 4096
        fut_batches = pool.submit(rcv1_shaped_batches, np.random.default_rng(42), nd, B, args.ids)
    table = capi.Table(ctx, capacity, V_dim=k, init_mode=capi.INIT_HASH, **hyper)

    t0 = time.time()
    if not args.no_prefill:
        # warm model: every feature id present with an allocated V row, so every
        # gathered row moves its full (1+k)*4 bytes (worst-case traffic)
        groups = ([synth.reverse_bytes_np(gen.all_ids(g)) for g in range(S)] if criteo and args.ids <= 40_000_000 else
                  (synth.reverse_bytes_np(gen.all_ids(g)) for g in range(S)) if criteo else
                  [synth.reverse_bytes_np(np.arange(1, args.ids + 1, dtype=np.uint64))])
        for keys in groups:
            chunk = 1 << 22
            for o in range(0, len(keys), chunk):
                part = np.ascontiguousarray(keys[o:o + chunk])
                db = capi.DeviceBuffer.from_numpy(ctx, part)
                table.warm_start(db.ptr, len(part), w0=0.01, cnt0=100.0)
                ctx.sync()
                db.close()
    nkeys = table.size()
    t_prefill = time.time() - t0
    host_batches = fut_batches.result()
    pool.shutdown()
    max_nnz = max(int(hb["offset"][-1]) for hb in host_batches)

    dev = []
    for hb in host_batches:
        off32 = hb["offset"].astype(np.uint32)
        dev.append((capi.DeviceBuffer.from_numpy(ctx, off32), capi.DeviceBuffer.from_numpy(ctx, hb["index"]),
                    capi.DeviceBuffer.from_numpy(ctx, hb["label"]),
                    None if hb["value"] is None else capi.DeviceBuffer.from_numpy(ctx, hb["value"]),
                    len(hb["label"]), int(hb["offset"][-1])))
    # depth+1 batch objects: batches t+1 .. t+depth are localized on the preparation streams while
    # batch t trains on the main stream (updates are still applied strictly in batch order)
    sq = args.single_queue and not args.no_pipeline
    depth = 0 if (args.no_pipeline or sq) else max(1, min(args.prep_streams, 4))
    for kv in args.ctx_option:
        name, val = kv.split("=", 1)
        ctx.set_option(name, int(val))
    ctx.set_pipeline(depth)
    if sq:
        ctx.set_option("single_queue", 1)
    ahead = (args.ahead or 2) if sq else max(args.ahead or depth, 1)
    # one spare object so that a new Localizer never waits for the step that just ended to release its buffers
    # (single queue: everything is ordered by the one stream, ahead + 1 objects rotate)
    bts = [capi.Batch(ctx, B, max_nnz) for _ in range(ahead + (2 if depth else 1))]
    bt = bts[0]
    if not args.no_auc:
        for b_ in bts:
            b_.set_option("compute_auc", 1)   # BinClassMetric::AUC of every minibatch, like the reference's loop

    def prep(i):
        o, x, l, v, nr, nz = dev[i % nd]
        b = bts[i % len(bts)]
        b.attach_device(nr, nz, o.ptr, x.ptr, None if v is None else v.ptr, l.ptr)  # inputs are resident in HBM: no copy
        if args.prep_lookup and depth and args.fused_probe:
            b.localize(table=table)      # the Localizer's emit pass probes the key index itself
        else:
            b.localize()
            if args.prep_lookup and depth:   # one stream: the step's own lookup probes and pushes in one pass
                b.lookup(table)

    def step(i):
        if not args.no_relocalize or i + ahead < len(bts):
            prep(i + ahead)
        bts[i % len(bts)].sgd_step(table, is_train=True, push_cnt=not args.later_epoch)

    for i in range(ahead):
        prep(i)
    done = 0
    for i in range(args.warmup):
        step(done)
        done += 1
    ctx.sync()
    torch.cuda.synchronize()
    for b in bts:
        b.progress(reset=True)
    # live timing of the two big kernels inside the timed region: the k_forward / k_backward_all dispatch of every
    # TIMING_EVERY-th step carries a start/stop HIP event pair (hipExtLaunchKernelGGL: the kernel's own
    # begin/end stamps, no marker packets).  Every step would cost the job 3-7 %, every 4th 1 %, every 8th (the default) ~0.5 %:
    # a dispatch that carries events is followed by ~5 us of idle queue instead of ~1 (profiles/r06gap_step_gaps.txt).
    mask = 0 if args.no_timing else ((1 << capi.K_FORWARD) | (1 << capi.K_BACKWARD))
    if mask and os.environ.get("DFH_GAP_TRACE"):   # measurement: the lookup dispatch timed too (csrc: dfh_ctx_get_timing prints the gaps)
        mask |= 1 << capi.K_LOOKUP
    ctx.get_timing(reset=True)
    reps, enq = [], []
    t_all = 0.0
    while True:
        ctx.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if mask and i % TIMING_EVERY == 0:
                ctx.set_timing_mask(mask)
                step(done)
                ctx.set_timing_mask(0)
            else:
                step(done)
            done += 1
        t_enq = time.perf_counter() - t0   # host side only: everything is queued, nothing awaited yet
        ctx.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        reps.append(dt)
        enq.append(t_enq)
        t_all += dt
        if t_all >= args.min_time or len(reps) >= args.max_reps:
            break
    order = np.argsort(reps)
    dt = float(reps[order[len(reps) // 2]])  # the median repetition
    t_enqueued = float(enq[order[len(reps) // 2]])
    progs = [b.progress(reset=True) for b in bts]
    loss_sum = sum(p.loss for p in progs)
    rows_sum = sum(p.nrows for p in progs)
    timing = {} if args.no_timing else ctx.get_timing(reset=True)
    # per-kernel breakdown from a separate, fully instrumented pass (NOT part of the timed region)
    breakdown = {}
    if not args.no_timing:
        nb_steps = min(args.steps, 50)
        ctx.set_timing(True)
        for i in range(nb_steps):
            step(done)
            done += 1
        breakdown = {n: v[0] / nb_steps for n, v in ctx.get_timing(reset=True).items() if v[1] > 0}
        ctx.set_timing(False)
        for b in bts:
            b.progress(reset=True)
    U_all = []
    for b in bts:
        U_all.append(b.shape()[2])
    U_mean = float(np.mean(U_all))
    nnz_mean = float(np.mean([d[5] for d in dev]))
    s_mean = nnz_mean / B

    ex_per_s = args.steps * B / dt
    r_g = s_mean * (1 + k) * 4  # algorithmic gather bytes per example
    u = U_mean / B
    r_step = s_mean * (1 + k) * 4 + s_mean * k * 4 + u * (3 + 2 * k) * 4 * 2  # SURVEY 8d full-step accounting
    roofline = None
    if timing and timing["forward"][1] > 0:
        fwd_ms = timing["forward"][0] / timing["forward"][1]
        achieved = B * r_g / (fwd_ms * 1e-3) / 1e9
        tr, tr_src = pmc_traffic("k_forward<", args.preset)
        roofline = dict(bound="hbm", kernel="k_forward", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBPS, traffic=tr, traffic_source=tr_src,
                        algorithmic_bytes_per_launch=B * r_g, avg_launch_ms=fwd_ms,
                        launches_timed=int(timing["forward"][1]))
    # the same for the longest kernel of the step, the fused backward/update.  algorithmic = SURVEY 8d's
    # full-step accounting (XV re-read s*k*4 per example + state read and written (3+2k)*4*2 per unique key);
    # hbm_necessary = the state bytes alone: the XV re-read is an L2-resident B*kp*4 B array, not HBM traffic
    roofline_bwd = None
    if timing and timing["backward"][1] > 0:
        bwd_ms = timing["backward"][0] / timing["backward"][1]
        nec = U_mean * (3 + 2 * k) * 4 * 2
        bwd_bytes = B * s_mean * k * 4 + nec
        ach = bwd_bytes / (bwd_ms * 1e-3) / 1e9
        tr, tr_src = pmc_traffic(["void dfh::k_update_fused<", "k_update_fused<", "k_backward_all<"], args.preset)
        roofline_bwd = dict(bound="hbm", kernel="k_update_fused", achieved=ach, peak=HBM_PEAK_GBPS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBPS, traffic=tr, traffic_source=tr_src,
                            algorithmic_bytes_per_launch=bwd_bytes, hbm_necessary_bytes_per_launch=nec,
                            achieved_hbm_necessary=nec / (bwd_ms * 1e-3) / 1e9,
                            frac_hbm_necessary=nec / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            avg_launch_ms=bwd_ms, launches_timed=int(timing["backward"][1]))
    secondary = None
    if args.preset == "c3" and not args.no_secondary:
        # release the headline run's table and batches first: the C5 slice takes most of the HBM
        for b in bts:
            b.close()
        for d in dev:
            for x in d[:4]:
                if x is not None:
                    x.close()
        tbytes = table.bytes()
        table.close()
        ctx.close()
        torch.cuda.empty_cache()
        secondary = secondary_lines(args)
    else:
        tbytes = table.bytes()
    nb = args.cpu_batches
    cpu = None
    if nb != 0:
        if nb < 0:
            # ~10-20 s of CPU work: two passes (fill + timed) at two thread settings
            nb = max(2, min(nd, int((60000 if k <= 64 else 30000) / max(B, 1)) if criteo else 50))
        cpu = cpu_baseline(host_batches, k, nb, hyper, with_auc=not args.no_auc)

    names = {"c3": "C3: Criteo-shaped synthetic, %d ids / 39 slots, V_dim=%d, FTRL(w)+AdaGrad(V), 1 MI355X" % (args.ids, k),
             "c3-refdefaults": "C3 with the reference's default hyper-parameters (sgd_param.h:95-105: l1=1, V_threshold=10): "
                               "%d ids / 39 slots, V_dim=%d, 1 MI355X" % (args.ids, k),
             "c5-slice": "C5 slice: one GPU's share of the 1 B-id / V_dim=128 / l1-FTRL config — %d ids / 39 slots, V_dim=%d, "
                         "1 MI355X" % (args.ids, k),
             "c2": "C2 shape: rcv1-like synthetic, %d ids, ~%d real-valued features/row, batch %d, V_dim=%d, 1 MI355X"
                   % (args.ids, round(s_mean), B, k)}
    out = {
        "metric": "examples/sec (FM SGD worker step, %s, V_dim=%d)" % ("Criteo-shape" if criteo else "rcv1-shape", k),
        "value": ex_per_s, "unit": "examples/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": names[args.preset], "preset": args.preset,
                   "rows_per_step": B, "nnz_per_row": s_mean, "unique_keys_per_batch": U_mean,
                   "step": "device localize + pull + predict + evaluate + %scalcgrad + push/update" % ("" if args.no_auc else "AUC + "),
                   "auc_every_minibatch": not args.no_auc,
                   "model_keys": int(nkeys), "table_bytes": tbytes, "prefilled": not args.no_prefill, "hyper": hyper,
                   "distinct_batches": nd,
                   "distinct_batches_note": "SURVEY 8d asks for a stream of 1 000; 256 x 390 000 ids is a 2 GB working set against a 256 MB "
                                            "memory-side cache (a batch recurs every ~30 ms of device time) and 64 / 256 distinct batches "
                                            "measured the same rate (85.55 / 85.54 M, round 4) and so do 256 / 1 000 (84.43 / 84.41 M, profiles/r05w_*); "
                                            "--distinct 1000 adds ~28 s of host-side generation",
                   "pipelined_prep": not args.no_pipeline, "prep_streams": depth, "single_queue": bool(sq),
                   "minibatches_prepared_ahead": ahead,
                   "feature_counts_pushed_every_step": not args.later_epoch},
        "repetitions": len(reps), "timed_region_s_total": t_all,
        "ms_per_step_min": float(min(reps)) / args.steps * 1e3, "ms_per_step_max": float(max(reps)) / args.steps * 1e3,
        "value_note": "median of `repetitions` timed regions of `steps` steps each"
                      + (" — DIAGNOSTIC RUN without the Localizer in the step (--no-relocalize), not the metric" if args.no_relocalize else ""),
        "dominant_kernel": "k_update_fused (fused CalcGrad + Push/update): roofline_backward; `roofline` is the gather "
                           "kernel BASELINE.json's metric names (k_forward)",
        "roofline": roofline,
        "roofline_backward": roofline_bwd,
        "roofline_step": dict(bound="hbm", bytes_per_example=r_step, achieved=ex_per_s * r_step / 1e9, peak=HBM_PEAK_GBPS,
                              unit="GB/s", frac=ex_per_s * r_step / 1e9 / HBM_PEAK_GBPS,
                              note="SURVEY 8d R_step = s(1+k)4 + s k 4 + u(3+2k)8 with the measured u = U/B"),
        "roofline_requests": roofline_requests(args, dt / args.steps),
        "cpu_baseline": cpu,
        "secondary": secondary,
        "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
        "kernel_ms_per_step": breakdown,
        "kernel_ms_per_step_note": "separate instrumented pass after the timed region (HIP events around every kernel group)",
        "train_logloss_per_example": loss_sum / max(rows_sum, 1),
        "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
        "prefill_seconds": t_prefill,
    }
    # the driver keeps the line's parsed contract keys and the LAST 2 000 characters of the output: what a reader of its record
    # should see without the profiles goes to the end of the line — the secondaries in one short dict, the step's request-rate
    # fraction, and the dominant kernel's roofline block last
    if secondary:
        out["secondary_summary"] = {k_: (dict(M_examples_per_sec=round(v_["value"] / 1e6, 3), ms_per_step=round(v_["ms_per_step"], 5))
                                         if isinstance(v_, dict) and v_.get("value") else v_) for k_, v_ in secondary.items()}
    rq = out.pop("roofline_requests")
    if rq:
        out["roofline_requests"] = {k_: rq[k_] for k_ in ("bound", "per_step", "per_s", "ceiling_per_s", "frac", "per_step_source", "ceiling_source")}
    else:
        out["roofline_requests"] = None
    out["roofline_step"] = out.pop("roofline_step")
    out["roofline_backward"] = out.pop("roofline_backward")
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
