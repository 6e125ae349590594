"""TEST INFRASTRUCTURE: the sharded worker protocol over torch.distributed (all_to_all_single), kept for the
CPU (gloo) tests of the exchange logic with a test-double backend and as a second transport to compare the native one
with (bench.py --transport torch).  The product path is dfh_shard_step inside libdifacto_hip.so (csrc/dfh_shard.hip),
driven by difacto_amd/sharded.py:bench_main_native and the C++ host.

Per step and rank:
    localize own minibatch                      (device, dfh_localize; one step ahead)
    keys  --all_to_all_v-->  owners             [+ counts in epoch 0, same message]
    owners: resolve keys -> rows once; Push(kFeaCount) per source; Pull -> rows
    rows  --all_to_all_v-->  workers            ((1+V_dim) floats per key, fixed stride)
    worker: Predict / Evaluate / CalcGrad       (dfh_batch_forward / dfh_batch_backward)
    grads --all_to_all_v-->  owners
    owners: Push(kGradient), applied one source rank after the other
The compute is behind a small backend interface so the exchange logic can be exercised on CPU (gloo, world_size 2)
with a test double; HipBackend runs the HIP kernels through the C ABI.
"""
import collections
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from difacto_amd.sharded import U64MAX, balanced_splits, key_span, owner_of, uniform_splits  # noqa: E402,F401

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


NSLOTS = 3  # minibatch objects per rank: one stepping, up to two being prepared / pulled ahead


class HipBackend:
    """the product backend: HIP kernels through include/difacto_hip.h"""

    def __init__(self, device_index, V_dim, capacity, hyper, max_rows, max_nnz, pipeline=True):
        from difacto_amd import capi
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
    def owner_resolve(self, keys, seg, mslot=0):
        if not hasattr(self, "_n_entries"):
            self._n_entries = {}
        self._n_entries[mslot] = keys.numel()
        rowid = torch.empty(max(self.capi.multi_words(keys.numel(), len(seg) - 1), 1), dtype=torch.int32, device=self.device)   # row words + extras
        if keys.numel():
            self.table.shard_resolve_multi(keys, seg, rowid, mslot)
        return rowid

    def owner_pull(self, rowid, keys, rows, seg):
        if keys.numel():
            self.table.shard_pull_resolved(rowid, keys.numel(), rows)

    def owner_push_count(self, rowid, keys, cnt, seg, mslot=0):
        """Push(kFeaCount) of every source, applied in source order"""
        if keys.numel():
            self.table.shard_push_count_multi(rowid, keys, seg, cnt, mslot)

    def owner_push_grad(self, rowid, keys, grads, seg, mslot=0):
        """Push(kGradient) of every source, applied in source order; ends the step for these rows"""
        if keys.numel():
            self.table.shard_push_grad_multi(rowid, keys, seg, grads, mslot)

    def owner_release(self, rowid, mslot=0):
        """ends a step that pushes no gradients (validation)"""
        if rowid.numel():
            self.table.shard_release(rowid, self._n_entries[mslot], mslot)

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


class _Done:
    """stand-in for a collective that already completed (host-staged exchange)"""

    def wait(self):
        return True


class _Pending:
    """one submitted minibatch on its way through the stages of a step"""

    def __init__(self, slot, seq, is_train, push_cnt):
        self.slot, self.seq, self.is_train, self.push_cnt = slot, seq, is_train, push_cnt
        self.mslot = seq & 1  # owner-side source-mask slot: consecutive steps alternate
        self.cnt_issued = self.counted = self.sized = self.k_issued = self.pulled = self.rw_issued = False
        self.w_cnt = self.w_keys = self.w_rows = self.w_grads = None


class ShardedWorker:
    """one rank of the sharded SGD loop (worker + owner of one key range).

    submit(batch, is_train, push_cnt) enqueues the Localizer of a minibatch (up to NSLOTS in
    flight); step() completes the oldest one.  A step is six stages,
        K  keys (+ counts) to the owners          R  owners resolve, count-push, pull
        RW rows back to the workers               F  forward / backward on the pulled rows
        G  gradients to the owners                P  owners apply them, source rank after source rank
    exchange="sync" runs them in that order for one minibatch at a time: every worker reads the
    model all earlier minibatches have updated.  exchange="overlap" keeps TWO minibatches in
    flight, like the reference's batch tracker (sgd_learner.cc:219-223: the next batch is issued
    while one is still pending, so its Pull may be served before the previous Push has landed):
    per step() call the order is  K(t+1) | F(t) G(t) | R(t+1) RW(t+1) | P(t), with the collectives
    asynchronous, so the gradients of t travel while the owners pull for t+1 and the rows of t+1
    travel while the gradients of t are applied.  Every table operation still runs on the one
    compute stream in program order (nothing races on a row); minibatch t+1 reads the model
    without t's update (staleness 1, exactly one batch)."""

    def __init__(self, backend, group=None, stage_through_host=False, splits=None, exchange="sync"):
        """splits: np.uint64[world-1] first keys of shards 1.. (identical on all ranks); None =
        the uniform partition.  stage_through_host: exchange through host copies (for process
        groups whose backend cannot move device tensors, e.g. gloo when several ranks share one
        GPU in a test); the product path exchanges device buffers over RCCL directly"""
        if exchange not in ("sync", "overlap"):
            raise ValueError("exchange must be 'sync' or 'overlap'")
        self.be = backend
        self.group = group
        self.stage = bool(stage_through_host)
        self.overlap = exchange == "overlap"
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = backend.device
        self.stride = backend.stride
        self.cuda = self.device.type == "cuda"
        G = self.world
        self.queue = collections.deque()
        self.next_slot = 0
        self.seq = 0
        self._bounds = [torch.zeros(G + 1, dtype=torch.int64, device=self.device) for _ in range(NSLOTS)]
        self._hcnt = [torch.zeros((2, G), dtype=torch.int64, pin_memory=self.cuda) for _ in range(NSLOTS)]
        self._ev = [torch.cuda.Event() for _ in range(NSLOTS)] if self.cuda else None
        self.splits = None
        self.splits_host = uniform_splits(G) if G > 1 else np.zeros(0, np.uint64)
        if splits is not None and G > 1:
            sp = np.ascontiguousarray(np.asarray(splits, dtype=np.uint64))
            if len(sp) != G - 1 or np.any(sp[1:] < sp[:-1]):
                raise ValueError("splits must be world-1 ascending keys")
            self.splits = torch.from_numpy(sp.view(np.int64).copy()).to(self.device)
            self.splits_host = sp

    # ---- model files: Updater::Save / Load on the sharded table (one part per rank)
    @staticmethod
    def part_path(prefix, rank):
        return "%s.part-%05d" % (prefix, rank)

    def owned_range(self):
        """[lo, hi) of the reversed keys this rank owns (hi = 0: no upper bound)"""
        lo = int(self.splits_host[self.rank - 1]) if self.rank > 0 else 0
        hi = int(self.splits_host[self.rank]) if self.rank < self.world - 1 else 0
        return lo, hi

    def save_model(self, prefix, save_aux=True):
        """every rank writes its shard to <prefix>.part-<rank> (the C++ host's model format); call between steps"""
        if self.queue and any(p.rw_issued or p.pulled for p in self.queue):
            raise RuntimeError("save_model: a step is under way")
        self.be.sync()
        n = self.be.table.save(self.part_path(prefix, self.rank), save_aux)
        dist.barrier(group=self.group)
        return n

    def load_model(self, prefix, nparts):
        """every rank reads all `nparts` part files and keeps the keys of its own range: the model may
        have been written under any number of ranks and any split keys"""
        lo, hi = self.owned_range()
        total = 0
        for r in range(nparts):
            n, _ = self.be.table.load(self.part_path(prefix, r), lo, hi)
            total += n
        dist.barrier(group=self.group)
        return total

    # ---- plumbing
    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        """all_to_all_v; returns a handle whose wait() orders the compute stream after the exchange
        (asynchronous in overlap mode, already ordered in sync mode)"""
        if self.stage and inp.device.type != "cpu":
            if self.cuda:
                torch.cuda.current_stream().synchronize()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                                   group=self.group)
            out.copy_(h_out)
            return _Done()
        if not self.overlap:  # the compute stream waits right here: no handle to create and keep
            dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)
            return _Done()
        return dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits,
                                      group=self.group, async_op=True)

    def submit(self, data, is_train=True, push_cnt=False):
        if len(self.queue) >= NSLOTS:
            raise RuntimeError("at most %d minibatches in flight" % NSLOTS)
        slot = self.next_slot
        self.next_slot = (slot + 1) % NSLOTS
        self.be.submit(slot, data)
        self.queue.append(_Pending(slot, self.seq, is_train, push_cnt))
        self.seq += 1

    # ---- stages
    def _counts_issue(self, p):
        """how many keys does every rank send me?  (asynchronous all_to_all of the per-destination counts)"""
        b = self._bounds[p.slot]
        self.be.bounds(p.slot, self.world, b, self.splits)
        p.send_t = b[1:] - b[:-1]
        p.recv_t = torch.empty_like(p.send_t)
        p.w_cnt = self._a2a(p.recv_t, p.send_t)
        p.cnt_issued = True

    def _counts_finish(self, p):
        """device -> pinned host + event; no host wait here"""
        p.w_cnt.wait()
        h = self._hcnt[p.slot]
        h[0].copy_(p.send_t, non_blocking=True)
        h[1].copy_(p.recv_t, non_blocking=True)
        if self.cuda:
            self._ev[p.slot].record()
        p.counted = True

    def _sizes(self, p):
        """the one host wait of a step: on an event recorded a step (or more) ago"""
        if not p.cnt_issued:
            self._counts_issue(p)
        if not p.counted:
            self._counts_finish(p)
        if self.cuda:
            self._ev[p.slot].synchronize()
        p.send = self._hcnt[p.slot][0].tolist()
        p.recv = self._hcnt[p.slot][1].tolist()
        p.nrecv, p.U = sum(p.recv), sum(p.send)
        p.roff = [0]
        for n in p.recv:
            p.roff.append(p.roff[-1] + n)
        p.sized = True

    def _K(self, p):
        """keys (and epoch-0 counts, riding in the same message) to their owners"""
        if not p.sized:
            self._sizes(p)
        keys, cnt = self.be.unique_keys(p.slot, p.U)
        if p.push_cnt:
            p.kc = torch.stack((keys, cnt.view(torch.int32).to(torch.int64)), dim=1)   # [U, 2] int64
            p.rkc = torch.empty((p.nrecv, 2), dtype=torch.int64, device=self.device)
            p.w_keys = self._a2a(p.rkc, p.kc, p.recv, p.send)
        else:
            p.keys = keys
            p.rkeys = torch.empty(p.nrecv, dtype=torch.int64, device=self.device)
            p.w_keys = self._a2a(p.rkeys, keys, p.recv, p.send)
        p.k_issued = True

    def _R(self, p):
        """owners: resolve the received keys to rows once, Push(kFeaCount), Pull"""
        be = self.be
        p.w_keys.wait()
        if p.push_cnt:
            p.rkeys = p.rkc[:, 0].contiguous()
            rcnt = p.rkc[:, 1].to(torch.int32).view(torch.float32)
        p.rowid = be.owner_resolve(p.rkeys, p.roff, p.mslot)
        if p.push_cnt:
            be.owner_push_count(p.rowid, p.rkeys, rcnt, p.roff, p.mslot)  # source rank after source rank
        p.rrows = torch.empty((p.nrecv, self.stride), dtype=torch.float32, device=self.device)
        if p.nrecv:
            be.owner_pull(p.rowid, p.rkeys, p.rrows, p.roff)   # every source reads the same model version
        p.pulled = True

    def _RW(self, p):
        p.rows = torch.empty((p.U, self.stride), dtype=torch.float32, device=self.device)
        p.w_rows = self._a2a(p.rows, p.rrows, p.send, p.recv)
        p.rw_issued = True

    def _F(self, p):
        """worker math on the pulled rows"""
        p.w_rows.wait()
        self.be.forward(p.slot, p.rows)
        if p.is_train:
            p.grads = torch.empty((p.U, self.stride), dtype=torch.float32, device=self.device)
            self.be.backward(p.slot, p.rows, p.grads)

    def _G(self, p):
        if p.is_train:
            p.rgrads = torch.empty((p.nrecv, self.stride), dtype=torch.float32, device=self.device)
            p.w_grads = self._a2a(p.rgrads, p.grads, p.recv, p.send)

    def _P(self, p):
        """owners: gradients applied in source-rank order (or the step released)"""
        if p.is_train:
            p.w_grads.wait()
            self.be.owner_push_grad(p.rowid, p.rkeys, p.rgrads, p.roff, p.mslot)
        else:
            self.be.owner_release(p.rowid, p.mslot)

    # ---- one step
    def step(self):
        """completes the oldest submitted minibatch"""
        q = self.queue
        p = q[0]
        nxt = q[1] if len(q) > 1 else None
        if not self.overlap:
            self._K(p)
            self._R(p)
            self._RW(p)
            self._F(p)
            # the next minibatch's Localizer has been running beside this step: exchange its counts now,
            # ahead of the gradient exchange, so that the next step() finds them on the host
            if nxt is not None and not nxt.cnt_issued:
                self._counts_issue(nxt)
                self._counts_finish(nxt)
            self._G(p)
            self._P(p)
        else:
            nn = q[2] if len(q) > 2 else None
            if not p.rw_issued:          # pipeline fill: nothing of this minibatch is under way yet
                self._K(p)
                self._R(p)
                self._RW(p)
            if nxt is not None and not nxt.k_issued:
                self._K(nxt)             # small; travels while F(t) computes
            if nn is not None and not nn.cnt_issued:
                self._counts_issue(nn)   # tiny, ahead of the big transfers on the collective stream
            self._F(p)
            self._G(p)                   # gradients of t travel ...
            if nn is not None and nn.cnt_issued and not nn.counted:
                self._counts_finish(nn)  # early in the step: the next step()'s one host wait finds it done
            if nxt is not None:
                self._R(nxt)             # ... while the owners pull for t+1 (before t's update: staleness 1)
                self._RW(nxt)            # rows of t+1 travel ...
            self._P(p)                   # ... while the gradients of t are applied
        q.popleft()
        return dict(unique=p.U, sent=p.send, received=p.recv, slot=p.slot)


# --------------------------------------------------------------------------- bench (N > 1)
def bench_main(args, rank, world, local_rank, hyper):
    """bench.py --gpus N under torchrun: weak scaling, every rank trains its own
    B-row minibatch per step against the key-range-sharded model"""
    from difacto_amd import capi, synth
    from difacto_amd.build import build_hip
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
    # a key present in every worker's minibatch receives `world` gradient pushes per step: the learning
    # rates are divided by the number of workers so that they move it about as far as one worker's push
    # would (with the single-worker rates the 8-worker run drifts: logloss 0.74 after 13 steps, 3.9 with
    # two minibatches in flight).  Throughput does not depend on it.
    hyper = dict(hyper)
    hyper["lr"] = hyper["lr"] / world
    hyper["V_lr"] = hyper["V_lr"] / world
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    # key ranges balanced on the id space (every rank derives the same split keys): the feature-group id
    # sits in the top bits of a reversed key, so uniform ranges would give one shard 2.3x the average
    splits = None
    if world > 1 and not args.uniform_ranges:
        # balanced on the id space (= on the rows every shard holds): the split keys come out of the library
        # (dfh_shard_balanced_splits: quantiles of the union of the ranks' samples); every rank samples other slots
        sample = [synth.reverse_bytes_np(gen.all_ids(g))[::61] for g in range(S) if g % world == rank]
        splits = comm.balanced_splits(np.concatenate(sample) if sample else np.zeros(0, np.uint64))
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
    worker = ShardedWorker(be, stage_through_host=dry, splits=splits, exchange=args.exchange)
    ahead = 2 if args.exchange == "overlap" else 1
    extra = 0 if args.no_timing else min(args.steps, 30)  # instrumented pass after the timed region
    total = args.warmup + args.steps + extra

    def step(i):
        # the reader's overlap (sgd_learner.cc:196-224): later minibatches are localized while i steps
        if i + ahead < total:
            worker.submit(dev[(i + ahead) % nd], is_train=True, push_cnt=True)
        return worker.step()

    for i in range(min(ahead, total)):
        worker.submit(dev[i % nd], is_train=True, push_cnt=True)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dist.barrier()
    be.progress()
    # live timing of the forward kernel (its dispatch carries the event pair) on every 4th step of rank 0
    fwd_mask = 0 if (args.no_timing or rank != 0) else (1 << capi.K_FORWARD)
    be.ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if fwd_mask and i % 4 == 0:
            be.ctx.set_timing_mask(fwd_mask)
            info = step(args.warmup + i)
            be.ctx.set_timing_mask(0)
        else:
            info = step(args.warmup + i)
    torch.cuda.synchronize()
    dist.barrier()
    fwd_t = be.ctx.get_timing(reset=True).get("forward", (0.0, 0)) if fwd_mask else (0.0, 0)
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
        roofline = None
        if fwd_t[1] > 0:
            fwd_ms = fwd_t[0] / fwd_t[1]
            achieved = B * r_g / (fwd_ms * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel="k_forward (rows pulled into the exchange layout, rank 0)", achieved=achieved,
                            peak=8000.0, unit="GB/s", frac=achieved / 8000.0, traffic=None,
                            algorithmic_bytes_per_launch=B * r_g, avg_launch_ms=fwd_ms, launches_timed=int(fwd_t[1]))
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
                       "exchange": "overlap: two minibatches in flight, staleness 1 (sgd_learner.cc:219-223)"
                                   if args.exchange == "overlap" else "sync: one minibatch at a time, zero staleness",
                       "owned_keys_rank0": int(owned)},
            "roofline": roofline, "cpu_baseline": None,
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
