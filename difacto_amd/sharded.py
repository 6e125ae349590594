"""Multi-GPU FM/SGD worker path: one process per GPU, the model row-sharded by key
range across the ranks, one exchange each way per step.

This is the parameter-server pattern the reference's interfaces were designed
for — workers own data parts, servers own key ranges (Store::Push/Pull,
include/difacto/store.h:53-73) — with every rank being both.  Rank g owns a
contiguous range of the reversed keys: ReverseBytes exists to make range
partitioning work (include/difacto/base.h:29-38), and because the Localizer emits
keys in ascending order (src/data/localizer.cc:28-48) each destination's keys are
one contiguous slice: send buffers need no permutation.  The ranges are either the
uniform ones (owner = key / ceil(2^64/G)) or given by explicit split keys balanced
on the id space (balanced_splits): with feature-group ids in the low bits of an id
(EncodeFeaGrpID, base.h:60-63) the uniform split is badly skewed.

Per step and rank (torch.distributed all_to_all_single = RCCL over xGMI):
    localize own minibatch                      (device, dfh_localize; one step ahead)
    keys  --all_to_all_v-->  owners             [+ counts in epoch 0, same message]
    owners: resolve keys -> rows once; Push(kFeaCount) per source; Pull -> rows
    rows  --all_to_all_v-->  workers            ((1+V_dim) floats per key, fixed stride)
    worker: Predict / Evaluate / CalcGrad       (dfh_batch_forward / dfh_batch_backward)
    grads --all_to_all_v-->  owners
    owners: Push(kGradient), applied one source rank after the other
There is no all-reduce anywhere: the traffic is key-routed rows, an all-to-all
that uses all seven xGMI links of a GPU at once.

Semantics: G synchronous workers read the same model version, then their
gradients are applied sequentially in source-rank order (FTRL/AdaGrad are not
linear, so sum-then-update would be a different algorithm) — a legal execution of
the reference's asynchronous Push protocol.

The compute is behind a small backend interface so the exchange logic can be
exercised on CPU (gloo, world_size 2) with a test double; the product backend is
HipBackend (HIP kernels through the C ABI).  No CPU fallback is selected
implicitly: ShardedWorker requires an explicit backend.
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


class _DevPtr:
    """expose a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def torch_view(ptr, n, dtype, device):
    """zero-copy torch tensor over device memory owned by the C library"""
    if n == 0:
        return torch.empty(0, dtype=dtype, device=device)
    typestr = {torch.int64: "<i8", torch.float32: "<f4", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevPtr(ptr, (n,), typestr), device=device)


NSLOTS = 2  # minibatches in flight per rank: one stepping, one being localized


class HipBackend:
    """the product backend: HIP kernels through include/difacto_hip.h"""

    def __init__(self, device_index, V_dim, capacity, hyper, max_rows, max_nnz, pipeline=True):
        from . import capi
        self.capi = capi
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # one torch stream carries the step; the library's kernels, torch's copies and the RCCL
        # collectives are all ordered on it.  The Localizer of the NEXT minibatch runs on the
        # library's own preparation stream (dfh_ctx_set_pipeline) and is joined with events.
        self.stream = torch.cuda.Stream(device=self.device)
        torch.cuda.set_stream(self.stream)
        self.ctx = capi.Context(device_index, stream=self.stream.cuda_stream)
        if pipeline:
            self.ctx.set_pipeline(True)
        self.table = capi.Table(self.ctx, capacity, V_dim=V_dim, init_mode=capi.INIT_HASH, **hyper)
        self.batches = [capi.Batch(self.ctx, max_rows, max_nnz) for _ in range(NSLOTS)]
        self._keep = [None] * NSLOTS
        self.V_dim = V_dim
        self.stride = capi.row_stride(V_dim)

    # ---- worker side
    def submit(self, slot, data):
        """enqueue load + Localizer for a minibatch: either the reader's host arrays
        dict(offset u64, index u64 raw ids, value f32|None, label f32), or arrays already in
        HBM as dict(device=True, nrows, nnz, offset i32, index i64, value|None, label)"""
        b = self.batches[slot]
        if data.get("device"):
            self._keep[slot] = data  # the batch reads the caller's tensors in place
            b.attach_device(data["nrows"], data["nnz"], data["offset"], data["index"], data.get("value"), data["label"])
        else:
            b.load_host(data["offset"], data["index"], data["value"], data["label"])
        b.localize()

    def bounds(self, slot, world, out, splits=None):
        """out[world+1] (int64, device): keys of shard d are feaids[out[d]:out[d+1]]; asynchronous.
        splits: device int64 tensor with the bit patterns of the world-1 first keys, or None (uniform)"""
        self.batches[slot].key_ranges_device(world, out, splits)

    def unique_keys(self, slot, U):
        """-> (keys int64 tensor [U] (bit pattern of the u64 keys), counts float32 [U])"""
        pk, pc = self.batches[slot].device_key_ptrs()
        return torch_view(pk, U, torch.int64, self.device), torch_view(pc, U, torch.float32, self.device)

    def forward(self, slot, rows):
        self.batches[slot].forward(self.V_dim, rows.data_ptr())

    def backward(self, slot, rows, grads):
        self.batches[slot].backward(self.V_dim, rows.data_ptr(), grads.data_ptr())

    def progress(self):
        tot = None
        for b in self.batches:
            p = b.progress(reset=True)
            if tot is None:
                tot = p
            else:
                for f, _ in p._fields_:
                    setattr(tot, f, getattr(tot, f) + getattr(p, f))
        return tot

    def pred(self, slot):
        return self.batches[slot].pred()

    # ---- owner side: the keys received in a step (concatenated ascending lists, source s in
    # [seg[s], seg[s+1])) are resolved to rows once; every operation is one launch for all sources
    def owner_resolve(self, keys, seg):
        rowid = torch.empty(keys.numel(), dtype=torch.int32, device=self.device)
        if keys.numel():
            self.table.shard_resolve_multi(keys, seg, rowid)
        return rowid

    def owner_pull(self, rowid, keys, rows, seg):
        if keys.numel():
            self.table.shard_pull_resolved(rowid, rowid.numel(), rows)

    def owner_push_count(self, rowid, keys, cnt, seg):
        """Push(kFeaCount) of every source, applied in source order"""
        if keys.numel():
            self.table.shard_push_count_multi(rowid, keys, seg, cnt)

    def owner_push_grad(self, rowid, keys, grads, seg):
        """Push(kGradient) of every source, applied in source order; ends the step for these rows"""
        if keys.numel():
            self.table.shard_push_grad_multi(rowid, keys, seg, grads)

    def owner_release(self, rowid):
        """ends a step that pushes no gradients (validation)"""
        if rowid.numel():
            self.table.shard_release(rowid, rowid.numel())

    def sync(self):
        self.ctx.sync()

    def check(self):
        self.table.check()

    def close(self):
        self.ctx.sync()
        for b in self.batches:
            b.close()
        self.table.close()
        self.ctx.close()


class _Pending:
    __slots__ = ("slot", "counted")

    def __init__(self, slot):
        self.slot, self.counted = slot, False


class ShardedWorker:
    """one rank of the sharded SGD loop (worker + owner of one key range).

    submit(batch) enqueues the Localizer of a minibatch (up to NSLOTS in flight);
    step() runs the oldest submitted one.  Submitting batch t+1 before step(t) lets its
    Localizer and the exchange of its per-destination key counts overlap step t, so a
    step waits on the host only for an event that was recorded long before."""

    def __init__(self, backend, group=None, stage_through_host=False, splits=None):
        """splits: np.uint64[world-1] first keys of shards 1.. (identical on all ranks); None =
        the uniform partition.  stage_through_host: exchange through host copies (for process groups whose backend
        cannot move device tensors, e.g. gloo when several ranks share one GPU in a test);
        the product path exchanges device buffers over RCCL directly"""
        self.be = backend
        self.group = group
        self.stage = bool(stage_through_host)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = backend.device
        self.stride = backend.stride
        self.cuda = self.device.type == "cuda"
        G = self.world
        self.queue = collections.deque()
        self.next_slot = 0
        self._bounds = [torch.zeros(G + 1, dtype=torch.int64, device=self.device) for _ in range(NSLOTS)]
        self._hcnt = [torch.zeros((2, G), dtype=torch.int64, pin_memory=self.cuda) for _ in range(NSLOTS)]
        self._ev = [torch.cuda.Event() for _ in range(NSLOTS)] if self.cuda else None
        self.splits = None
        if splits is not None and G > 1:
            sp = np.ascontiguousarray(np.asarray(splits, dtype=np.uint64))
            if len(sp) != G - 1 or np.any(sp[1:] < sp[:-1]):
                raise ValueError("splits must be world-1 ascending keys")
            self.splits = torch.from_numpy(sp.view(np.int64).copy()).to(self.device)

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        if self.stage and inp.device.type != "cpu":
            if self.cuda:
                torch.cuda.current_stream().synchronize()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)
            out.copy_(h_out)
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def submit(self, data):
        if len(self.queue) >= NSLOTS:
            raise RuntimeError("at most %d minibatches in flight" % NSLOTS)
        slot = self.next_slot
        self.next_slot = (slot + 1) % NSLOTS
        self.be.submit(slot, data)
        self.queue.append(_Pending(slot))

    def _exchange_counts(self, p):
        """how many keys does every rank send me?  (device -> pinned host, no host wait here)"""
        b = self._bounds[p.slot]
        self.be.bounds(p.slot, self.world, b, self.splits)
        send_t = b[1:] - b[:-1]
        recv_t = torch.empty_like(send_t)
        self._a2a(recv_t, send_t)
        h = self._hcnt[p.slot]
        h[0].copy_(send_t, non_blocking=True)
        h[1].copy_(recv_t, non_blocking=True)
        if self.cuda:
            self._ev[p.slot].record()
        p.counted = True

    def step(self, is_train=True, push_cnt=False):
        """one synchronous step over the oldest submitted minibatch"""
        be, G = self.be, self.world
        p = self.queue.popleft()
        slot = p.slot
        if not p.counted:
            self._exchange_counts(p)
        if self.cuda:
            self._ev[slot].synchronize()
        send = self._hcnt[slot][0].tolist()
        recv = self._hcnt[slot][1].tolist()
        nrecv, U = sum(recv), sum(send)
        keys, cnt = be.unique_keys(slot, U)

        # 1. keys (and epoch-0 counts, riding in the same message) to their owners; owners resolve
        #    them to table rows once
        roff = [0]
        for n in recv:
            roff.append(roff[-1] + n)
        if push_cnt:
            kc = torch.stack((keys, cnt.view(torch.int32).to(torch.int64)), dim=1)   # [U, 2] int64
            rkc = torch.empty((nrecv, 2), dtype=torch.int64, device=self.device)
            self._a2a(rkc, kc, recv, send)
            rkeys = rkc[:, 0].contiguous()
            rcnt = rkc[:, 1].to(torch.int32).view(torch.float32)
        else:
            rkeys = torch.empty(nrecv, dtype=torch.int64, device=self.device)
            self._a2a(rkeys, keys, recv, send)
        rowid = be.owner_resolve(rkeys, roff)
        if push_cnt:
            be.owner_push_count(rowid, rkeys, rcnt, roff)  # Push(kFeaCount), source rank after source rank
        # 2. owners pull rows (every source reads the same model version) and send them back
        rrows = torch.empty((nrecv, self.stride), dtype=torch.float32, device=self.device)
        if nrecv:
            be.owner_pull(rowid, rkeys, rrows, roff)
        rows = torch.empty((U, self.stride), dtype=torch.float32, device=self.device)
        self._a2a(rows, rrows, send, recv)
        # 3. worker math on the pulled rows
        be.forward(slot, rows)
        if is_train:
            grads = torch.empty((U, self.stride), dtype=torch.float32, device=self.device)
            be.backward(slot, rows, grads)
        # the next minibatch's Localizer has been running beside this step: exchange its counts now,
        # ahead of the gradient exchange, so that the next step() finds them on the host
        if self.queue and not self.queue[0].counted:
            self._exchange_counts(self.queue[0])
        if is_train:
            # 4. gradients to the owners, applied in source-rank order
            rgrads = torch.empty((nrecv, self.stride), dtype=torch.float32, device=self.device)
            self._a2a(rgrads, grads, recv, send)
            be.owner_push_grad(rowid, rkeys, rgrads, roff)
        else:
            be.owner_release(rowid)
        return dict(unique=U, sent=send, received=recv, slot=slot)


# --------------------------------------------------------------------------- bench (N > 1)
def bench_main(args, rank, world, local_rank, hyper):
    """bench.py --gpus N under torchrun: weak scaling, every rank trains its own
    B-row minibatch per step against the key-range-sharded model"""
    from . import capi, synth
    from .build import build_hip
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (no CPU fallback)")
    # RCCL prints a version banner on the process's stdout; the bench contract is ONE JSON line there.
    # Keep a private handle on the real stdout and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # DFH_BENCH_BACKEND=gloo: functional dry run of this very code on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange is staged through the host); numbers from it mean nothing
    dry = os.environ.get("DFH_BENCH_BACKEND", "nccl") == "gloo"
    if dry:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if dry:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if rank == 0:
        build_hip()
    dist.barrier()
    B, k, S = args.rows, args.vdim, synth.NUM_SLOTS
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    # key ranges balanced on the id space (every rank derives the same split keys): the feature-group id
    # sits in the top bits of a reversed key, so uniform ranges would give one shard 2.3x the average
    splits = None
    if world > 1 and not args.uniform_ranges:
        splits = balanced_splits(np.concatenate([synth.reverse_bytes_np(gen.all_ids(g))[::61] for g in range(S)]), world)
    elif world > 1:
        splits = uniform_splits(world)
    mine = []
    for g in range(S):
        keys = synth.reverse_bytes_np(gen.all_ids(g))
        mine.append(keys[owner_of(keys, splits) == rank] if world > 1 else keys)
    owned = sum(len(m) for m in mine)
    # owned share of the id space + slack for insert-on-miss
    cap = int(owned * 1.05) + 8 * B * S
    be = HipBackend(local_rank, k, cap, hyper, B, B * S)
    if not args.no_prefill:
        chunk = 1 << 22
        for m in mine:
            for o in range(0, len(m), chunk):
                part = torch.from_numpy(np.ascontiguousarray(m[o:o + chunk]).view(np.int64)).to(be.device)
                be.table.warm_start(part.data_ptr(), part.numel(), w0=0.01, cnt0=100.0)
                be.sync()
    del mine
    t_prefill = time.time() - t0
    # every rank draws its own stream of minibatches (different data parts, sgd_learner.cc:78-89)
    gen.rng = np.random.default_rng(1000 + rank)
    nd = max(1, min(args.distinct, args.steps + args.warmup))
    dev = []
    for _ in range(nd):
        hb = gen.batch(B)
        dev.append(dict(device=True, nrows=B, nnz=B * S, value=None,
                        offset=torch.from_numpy(hb["offset"].astype(np.uint32).view(np.int32)).to(be.device),
                        index=torch.from_numpy(hb["index"].view(np.int64)).to(be.device),
                        label=torch.from_numpy(hb["label"]).to(be.device)))
    worker = ShardedWorker(be, stage_through_host=dry, splits=splits)
    extra = 0 if args.no_timing else min(args.steps, 30)  # instrumented pass after the timed region
    total = args.warmup + args.steps + extra

    def step(i):
        # the reader's overlap (sgd_learner.cc:196-224): minibatch i+1 is localized while i steps
        if i + 1 < total:
            worker.submit(dev[(i + 1) % nd])
        return worker.step(is_train=True, push_cnt=True)

    worker.submit(dev[0])
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dist.barrier()
    be.progress()
    t0 = time.perf_counter()
    for i in range(args.steps):
        info = step(args.warmup + i)
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=be.device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    be.check()
    prog = be.progress()
    timing = {}
    if extra:  # per-kernel breakdown, outside the timed region (event pairs drain the stream)
        be.ctx.set_timing(True)
        for i in range(extra):
            step(args.warmup + args.steps + i)
        timing = {n: round(ms / extra, 4) for n, (ms, calls) in be.ctx.get_timing(reset=True).items() if calls}
        be.ctx.set_timing(False)
        be.progress()
    stats = torch.tensor([prog.loss, prog.nrows, float(info["unique"]), float(sum(info["sent"]) - info["sent"][rank])],
                         dtype=torch.float64, device=be.device)
    dist.all_reduce(stats)
    if rank == 0:
        ex_per_s = args.steps * B * world / dt
        r_g = S * (1 + k) * 4
        out = {
            "metric": "examples/sec (FM SGD worker step, Criteo-shape, V_dim=%d)" % k,
            "value": ex_per_s, "unit": "examples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4: Criteo-shaped synthetic, %d ids / 39 slots, V_dim=%d, model row-sharded by key "
                                   "range over %d MI355X, RCCL all_to_all_v" % (args.ids, k, world),
                       "rows_per_step_per_gpu": B, "nnz_per_row": S, "parallelism": "shard%d" % world,
                       "step": "device localize + key/row/gradient all_to_all_v + predict + calcgrad + in-place update",
                       "avg_unique_keys_per_batch": stats[2].item() / world,
                       "avg_remote_keys_per_batch": stats[3].item() / world,
                       "prefilled": not args.no_prefill, "hyper": hyper, "dry_run_shared_gpu": dry,
                       "key_ranges": "uniform" if args.uniform_ranges else "balanced on the id space",
                       "owned_keys_rank0": int(owned)},
            "roofline": None, "cpu_baseline": None,
            "train_logloss_per_example": stats[0].item() / max(stats[1].item(), 1.0),
            "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
            "prefill_seconds": t_prefill,
            "kernel_ms_per_step_rank0": timing,
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    be.close()
    dist.destroy_process_group()
    return 0
