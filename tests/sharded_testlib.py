"""Test double for difacto_amd.sharded's compute backend: the CPU oracle stands in
for the HIP kernels so the exchange logic (key-range partition, all_to_all_v of
keys / rows / gradients, sequential application in source-rank order) can be
exercised with gloo on CPU.  TEST INFRASTRUCTURE ONLY."""
import numpy as np
import torch

from oracle import bindings as ob

U64MAX = 2 ** 64 - 1


def row_stride(V_dim):
    return 4 + (V_dim + 3) // 4 * 4


class OracleBackend:
    def __init__(self, V_dim, hyper):
        self.o = ob.Oracle()
        self.store = self.o.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **hyper)
        self.V_dim = V_dim
        self.stride = row_stride(V_dim)
        self.device = torch.device("cpu")
        self.loss = 0.0
        self.nrows = 0.0
        self.slots = {}

    # worker side (two minibatches may be in flight, like the HIP backend)
    def submit(self, slot, b):
        self.slots[slot] = dict(b=b, loc=self.o.localize(b["offset"], b["index"]))

    def _use(self, slot):
        self.b, self.loc = self.slots[slot]["b"], self.slots[slot]["loc"]

    def unique_keys(self, slot, U):
        loc = self.slots[slot]["loc"]
        assert U == loc["U"]
        return (torch.from_numpy(loc["feaids"].view(np.int64).copy()), torch.from_numpy(loc["feacnt"].copy()))

    def bounds(self, slot, world, out, splits=None):
        loc = self.slots[slot]["loc"]
        span = U64MAX if world == 1 else U64MAX // world + 1
        firsts = np.array([min(d * span, U64MAX) for d in range(world)], dtype=np.uint64)
        if splits is not None:
            firsts[1:] = splits.numpy().view(np.uint64)
        b = np.searchsorted(loc["feaids"], firsts, side="left").astype(np.int64)
        out.copy_(torch.from_numpy(np.concatenate([b, [loc["U"]]]).astype(np.int64)))

    def _ragged(self, rows):
        rows = rows.numpy()
        k = self.V_dim
        has_v = rows[:, 1] != 0
        lens = np.where(has_v, 1 + k, 1).astype(np.int32)
        vals = []
        for r, hv in zip(rows, has_v):
            vals.append(r[0:1])
            if hv:
                vals.append(r[4:4 + k])
        W = np.concatenate(vals).astype(np.float32) if len(vals) else np.zeros(0, np.float32)
        return W, lens

    def forward(self, slot, rows):
        self._use(slot)
        W, lens = self._ragged(rows)
        self._W, self._lens = W, lens
        if self.V_dim:
            self._wp, self._vp = self.o.get_pos(lens)
        else:
            self._wp = self._vp = None
        self._pred = self.o.fm_predict(self.V_dim, self.loc["offset"], self.loc["index"], self.b["value"], W, self._wp, self._vp)
        self.loss += self.o.loss_evaluate(self.b["label"], self._pred)
        self.nrows += len(self._pred)

    def backward(self, slot, rows, grads):
        self._use(slot)
        g = self.o.fm_calcgrad(self.V_dim, self.loc["offset"], self.loc["index"], self.b["value"], self.b["label"],
                               self._W, self._pred, self._wp, self._vp)
        out = grads.numpy()
        out[:] = 0
        k = self.V_dim
        p = 0
        for u, l in enumerate(self._lens):
            out[u, 0] = g[p]
            p += 1
            if l > 1:
                out[u, 1] = 1.0
                out[u, 4:4 + k] = g[p:p + k]
                p += k

    def pred(self, slot=None):
        return self._pred

    # owner side (the HIP backend's interface: all sources of a step per call; the double
    # walks the sources one Store call at a time, which is what those calls must equal)
    def owner_resolve(self, keys, seg, mslot=0):
        return torch.zeros(keys.numel(), dtype=torch.int32)

    def owner_push_count(self, rowid, keys, cnt, seg, mslot=0):
        for s in range(len(seg) - 1):
            if seg[s + 1] > seg[s]:
                self.store.push(keys[seg[s]:seg[s + 1]].numpy().view(np.uint64), ob.FEA_COUNT, cnt[seg[s]:seg[s + 1]].numpy())

    def owner_pull(self, rowid, keys, rows, seg):
        # one Store::Pull per source rank (keys are unique only within a source)
        for s in range(len(seg) - 1):
            if seg[s + 1] > seg[s]:
                self._pull(keys[seg[s]:seg[s + 1]], rows[seg[s]:seg[s + 1]])

    def owner_push_grad(self, rowid, keys, grads, seg, mslot=0):
        for s in range(len(seg) - 1):
            if seg[s + 1] > seg[s]:
                self._push_grad(keys[seg[s]:seg[s + 1]], grads[seg[s]:seg[s + 1]])

    def owner_release(self, rowid, mslot=0):
        pass

    def _pull(self, keys, rows):
        vals, lens = self.store.pull(keys.numpy().view(np.uint64))
        out = rows.numpy()
        out[:] = 0
        k = self.V_dim
        p = 0
        for u in range(keys.numel()):
            out[u, 0] = vals[p]
            p += 1
            if k and lens[u] > 1:
                out[u, 1] = 1.0
                out[u, 4:4 + k] = vals[p:p + k]
                p += k

    def _push_grad(self, keys, grads):
        g = grads.numpy()
        k = self.V_dim
        vals, lens = [], []
        for r in g:
            vals.append(r[0:1])
            if k and r[1] != 0:
                vals.append(r[4:4 + k])
                lens.append(1 + k)
            else:
                lens.append(1)
        self.store.push(keys.numpy().view(np.uint64), ob.GRADIENT, np.concatenate(vals), np.array(lens, np.int32) if k else None)


def emulate_single_store(oracle, batches, V_dim, hyper, push_cnt_steps, overlap=False):
    """what the sharded step must equal: ONE store receiving the workers' requests in a fixed order.
    sync:    per step all count pushes (source rank order), all pulls (same model version), all
             gradient pushes (source rank order).
    overlap: two minibatches in flight (sgd_learner.cc:219-223): step t's gradient pushes land
             AFTER the count pushes and pulls of step t+1.
    batches[r][i] = minibatch of rank r at step i.  Returns (store, preds[r][i], loss[r])."""
    world, steps = len(batches), len(batches[0])
    store = oracle.store_create(init_mode=ob.INIT_HASH, V_dim=V_dim, **hyper)
    preds = [[] for _ in range(world)]
    loss = [0.0] * world
    locs = [[oracle.localize(batches[r][i]["offset"], batches[r][i]["index"]) for r in range(world)] for i in range(steps)]
    pulled = {}

    def count_and_pull(i):
        if i < push_cnt_steps:
            for r in range(world):
                store.push(locs[i][r]["feaids"], ob.FEA_COUNT, locs[i][r]["feacnt"])
        pulled[i] = [store.pull(locs[i][r]["feaids"]) for r in range(world)]

    for i in range(steps):
        if i not in pulled:
            count_and_pull(i)
        if overlap and i + 1 < steps:
            count_and_pull(i + 1)
        grads = []
        for r in range(world):
            b, loc = batches[r][i], locs[i][r]
            vals, lens = pulled[i][r]
            wp, vp = oracle.get_pos(lens)
            p = oracle.fm_predict(V_dim, loc["offset"], loc["index"], b["value"], vals, wp, vp)
            preds[r].append(p)
            loss[r] += oracle.loss_evaluate(b["label"], p)
            grads.append(oracle.fm_calcgrad(V_dim, loc["offset"], loc["index"], b["value"], b["label"], vals, p, wp, vp))
        for r in range(world):
            store.push(locs[i][r]["feaids"], ob.GRADIENT, grads[r], pulled[i][r][1])
    return store, preds, loss
