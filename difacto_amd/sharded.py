"""Multi-GPU FM/SGD worker path: one process per GPU, the model row-sharded by key
range across the ranks, one exchange each way per step.

This is the parameter-server pattern the reference's interfaces were designed
for — workers own data parts, servers own key ranges (Store::Push/Pull,
include/difacto/store.h:53-73) — with every rank being both: rank g owns the
reversed keys in [g*span, (g+1)*span), span = ceil(2^64/G).  ReverseBytes exists
to make exactly this range partition uniform (include/difacto/base.h:29-38), and
because the Localizer emits keys in ascending order (src/data/localizer.cc:28-48)
each destination's keys are one contiguous slice: send buffers need no permutation.

Per step and rank (torch.distributed all_to_all_single = RCCL over xGMI):
    localize own minibatch                      (device, dfh_localize)
    keys  --all_to_all_v-->  owners             [+ counts in epoch 0]
    owners: Push(kFeaCount), Pull -> rows       (dfh_shard_push_count / dfh_shard_pull)
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
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

U64MAX = 2 ** 64 - 1


def key_span(world):
    """shard d owns reversed keys in [d*span, (d+1)*span)"""
    return U64MAX if world == 1 else U64MAX // world + 1


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


class HipBackend:
    """the product backend: HIP kernels through include/difacto_hip.h"""

    def __init__(self, device_index, V_dim, capacity, hyper, max_rows, max_nnz):
        from . import capi
        self.capi = capi
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # share torch's current stream so that kernels, copies and RCCL collectives are stream-ordered
        self.ctx = capi.Context(device_index, stream=torch.cuda.current_stream().cuda_stream)
        self.table = capi.Table(self.ctx, capacity, V_dim=V_dim, init_mode=capi.INIT_HASH, **hyper)
        self.batch = capi.Batch(self.ctx, max_rows, max_nnz)
        self.V_dim = V_dim
        self.stride = capi.row_stride(V_dim)

    # ---- worker side
    def load_and_localize(self, b):
        """b: dict(offset u64, index u64 raw ids, value f32|None, label f32) on the host"""
        self.batch.load_host(b["offset"], b["index"], b["value"], b["label"])
        self.batch.localize()

    def load_and_localize_device(self, nrows, nnz, d_offset, d_index, d_value, d_label):
        self.batch.load_device(nrows, nnz, d_offset, d_index, d_value, d_label)
        self.batch.localize()

    def unique_keys(self):
        """-> (keys int64 tensor [U] (bit pattern of the u64 keys), counts float32 [U])"""
        pk, pc, U = self.batch.device_keys()
        return torch_view(pk, U, torch.int64, self.device), torch_view(pc, U, torch.float32, self.device)

    def key_ranges(self, world):
        return self.batch.key_ranges(world).astype(np.int64)

    def forward(self, rows):
        self.batch.forward(self.V_dim, rows.data_ptr())

    def backward(self, rows, grads):
        self.batch.backward(self.V_dim, rows.data_ptr(), grads.data_ptr())

    def progress(self):
        return self.batch.progress(reset=True)

    def pred(self):
        return self.batch.pred()

    # ---- owner side (n unique keys per call)
    def owner_push_count(self, keys, cnt):
        if keys.numel():
            self.table.shard_push_count(keys.data_ptr(), keys.numel(), cnt.data_ptr())

    def owner_pull(self, keys, rows):
        if keys.numel():
            self.table.shard_pull(keys.data_ptr(), keys.numel(), rows.data_ptr())

    def owner_push_grad(self, keys, grads):
        if keys.numel():
            self.table.shard_push_grad(keys.data_ptr(), keys.numel(), grads.data_ptr())

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.batch.close()
        self.table.close()
        self.ctx.close()


class ShardedWorker:
    """one rank of the sharded SGD loop (worker + owner of one key range)"""

    def __init__(self, backend, group=None):
        self.be = backend
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = backend.device
        self.stride = backend.stride

    def _a2a(self, out, inp, out_splits, in_splits):
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def step(self, is_train=True, push_cnt=False):
        """one synchronous step over the batch the backend has localized"""
        be, G = self.be, self.world
        keys, cnt = be.unique_keys()
        bounds = be.key_ranges(G)                       # feaids[bounds[d]:bounds[d+1]] -> owner d
        send = [int(bounds[d + 1] - bounds[d]) for d in range(G)]
        # how many keys does every rank send me?
        send_t = torch.tensor(send, dtype=torch.int64, device=self.device)
        recv_t = torch.empty(G, dtype=torch.int64, device=self.device)
        dist.all_to_all_single(recv_t, send_t, group=self.group)
        recv = [int(x) for x in recv_t.cpu().tolist()]
        nrecv, U = sum(recv), int(keys.numel())

        # 1. keys (and epoch-0 counts) to their owners
        rkeys = torch.empty(nrecv, dtype=torch.int64, device=self.device)
        self._a2a(rkeys, keys, recv, send)
        roff = np.concatenate([[0], np.cumsum(recv)]).astype(np.int64)
        if push_cnt:
            rcnt = torch.empty(nrecv, dtype=torch.float32, device=self.device)
            self._a2a(rcnt, cnt, recv, send)
            for s in range(G):  # Push(kFeaCount), source rank after source rank
                be.owner_push_count(rkeys[roff[s]:roff[s + 1]], rcnt[roff[s]:roff[s + 1]])
        # 2. owners pull rows and send them back
        rrows = torch.empty((nrecv, self.stride), dtype=torch.float32, device=self.device)
        for s in range(G):
            be.owner_pull(rkeys[roff[s]:roff[s + 1]], rrows[roff[s]:roff[s + 1]])
        rows = torch.empty((U, self.stride), dtype=torch.float32, device=self.device)
        self._a2a(rows, rrows, send, recv)
        # 3. worker math on the pulled rows
        be.forward(rows)
        if is_train:
            grads = torch.empty((U, self.stride), dtype=torch.float32, device=self.device)
            be.backward(rows, grads)
            # 4. gradients to the owners, applied in source-rank order
            rgrads = torch.empty((nrecv, self.stride), dtype=torch.float32, device=self.device)
            self._a2a(rgrads, grads, recv, send)
            for s in range(G):
                be.owner_push_grad(rkeys[roff[s]:roff[s + 1]], rgrads[roff[s]:roff[s + 1]])
        return dict(unique=U, sent=send, received=recv)


# --------------------------------------------------------------------------- bench (N > 1)
def bench_main(args, rank, world, local_rank, hyper):
    """bench.py --gpus N under torchrun: weak scaling, every rank trains its own
    B-row minibatch per step against the key-range-sharded model"""
    from . import capi, synth
    from .build import build_hip
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        build_hip()
    dist.barrier()
    B, k, S = args.rows, args.vdim, synth.NUM_SLOTS
    span = key_span(world)
    # owned share of the id space (+ slack for imbalance and insert-on-miss)
    cap = int(args.ids / world * 1.15) + 8 * B * S
    be = HipBackend(local_rank, k, cap, hyper, B, B * S)
    gen = synth.CriteoSynth(total_ids=args.ids, seed=42)
    t0 = time.time()
    if not args.no_prefill:
        for g in range(S):
            keys = synth.reverse_bytes_np(gen.all_ids(g))
            mine = keys[(keys // np.uint64(span)) == np.uint64(rank)] if world > 1 else keys
            chunk = 1 << 22
            for o in range(0, len(mine), chunk):
                part = torch.from_numpy(np.ascontiguousarray(mine[o:o + chunk]).view(np.int64)).to(be.device)
                be.table.warm_start(part.data_ptr(), part.numel(), w0=0.01, cnt0=100.0)
                be.sync()
    t_prefill = time.time() - t0
    # every rank draws its own stream of minibatches (different data parts, sgd_learner.cc:78-89)
    gen.rng = np.random.default_rng(1000 + rank)
    nd = max(1, min(args.distinct, args.steps + args.warmup))
    dev = []
    for _ in range(nd):
        hb = gen.batch(B)
        dev.append((torch.from_numpy(hb["offset"].astype(np.uint32).view(np.int32)).to(be.device),
                    torch.from_numpy(hb["index"].view(np.int64)).to(be.device),
                    torch.from_numpy(hb["label"]).to(be.device)))
    worker = ShardedWorker(be)

    def step(i):
        o, x, l = dev[i % nd]
        be.load_and_localize_device(B, B * S, o.data_ptr(), x.data_ptr(), None, l.data_ptr())
        return worker.step(is_train=True, push_cnt=True)

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
    prog = be.progress()
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
                       "prefilled": not args.no_prefill, "hyper": hyper},
            "roofline": None, "cpu_baseline": None,
            "train_logloss_per_example": stats[0].item() / max(stats[1].item(), 1.0),
            "hbm_gbps_step_algorithmic": ex_per_s * r_g / 1e9,
            "prefill_seconds": t_prefill,
        }
        print(json.dumps(out))
    be.close()
    dist.destroy_process_group()
    return 0
