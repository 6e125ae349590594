"""Criteo-shaped synthetic minibatches.

One feature per slot g in [0, 39): 13 "integer" slots with 10 000 ids each and
26 "categorical" slots whose vocabulary sizes fall off as 1/(g-12), all summing
to `total_ids` (33 M for the C3/C4 configs).  Within a slot the rank is
Zipf(alpha=1.05) truncated to the slot's vocabulary; the token is hashed with
splitmix64 (stand-in for CityHash64, src/reader/criteo_parser.h:72-84) and
tagged with the slot id in the low 12 bits exactly as the reference does:
    id = EncodeFeaGrpID(hash, g, 12) = (hash << 12) | g     (include/difacto/base.h:60-63)
Values are all ones, so the batch is binary (value == NULL, as BatchReader
drops all-ones values, src/reader/batch_reader.cc:71-73).  Labels ~ Bernoulli(0.25).

The generator is plain numpy so the very same arrays feed the GPU path and the
CPU baseline.
"""
import numpy as np

NUM_SLOTS = 39
NUM_INT_SLOTS = 13
INT_VOCAB = 10000
ALPHA = 1.05
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """vectorised splitmix64 on uint64 arrays (wrap-around arithmetic)"""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return z ^ (z >> np.uint64(31))


def reverse_bytes_np(x):
    """vectorised ReverseBytes (include/difacto/base.h:39-51): nibble reversal of u64"""
    x = x.astype(np.uint64)
    S = np.uint64
    x = (x << S(32)) | (x >> S(32))
    x = ((x & S(0x0000FFFF0000FFFF)) << S(16)) | ((x & S(0xFFFF0000FFFF0000)) >> S(16))
    x = ((x & S(0x00FF00FF00FF00FF)) << S(8)) | ((x & S(0xFF00FF00FF00FF00)) >> S(8))
    x = ((x & S(0x0F0F0F0F0F0F0F0F)) << S(4)) | ((x & S(0xF0F0F0F0F0F0F0F0)) >> S(4))
    return x


def slot_vocab_sizes(total_ids=33_000_000):
    v = np.zeros(NUM_SLOTS, np.int64)
    v[:NUM_INT_SLOTS] = min(INT_VOCAB, max(total_ids // (4 * NUM_SLOTS), 16))
    rest = total_ids - int(v[:NUM_INT_SLOTS].sum())
    wts = 1.0 / np.arange(1, NUM_SLOTS - NUM_INT_SLOTS + 1)
    cat = np.floor(rest * wts / wts.sum()).astype(np.int64)
    cat = np.maximum(cat, 16)
    cat[0] += rest - int(cat.sum())  # make the total exact
    v[NUM_INT_SLOTS:] = cat
    return v


_CDF_CACHE = {}   # (total_ids, alpha) -> per-slot CDFs: generators of one id space share them (8 GB at 1e9 ids)


class CriteoSynth:
    def __init__(self, total_ids=33_000_000, seed=42, alpha=ALPHA, pos_rate=0.25):
        self.vocab = slot_vocab_sizes(total_ids)
        self.seed = np.uint64(seed)
        self.alpha = alpha
        self.pos_rate = pos_rate
        self.rng = np.random.default_rng(seed)
        self._cdf = _CDF_CACHE.setdefault((int(total_ids), float(alpha)), [None] * NUM_SLOTS)

    def _slot_cdf(self, g):
        if self._cdf[g] is None:
            r = np.arange(1, int(self.vocab[g]) + 1, dtype=np.float64)
            np.power(r, -self.alpha, out=r)
            np.cumsum(r, out=r)
            r /= r[-1]
            self._cdf[g] = r
        return self._cdf[g]

    def ids_of(self, g, ranks):
        """raw feature id of (slot, rank): (splitmix64(seed ^ g<<40 ^ rank) << 12) | g"""
        key = self.seed ^ (np.uint64(g) << np.uint64(40)) ^ ranks.astype(np.uint64)
        h = splitmix64(key)
        with np.errstate(over="ignore"):
            return ((h << np.uint64(12)) & M64) | np.uint64(g)

    def all_ids(self, g):
        return self.ids_of(g, np.arange(int(self.vocab[g]), dtype=np.uint64))

    def batch(self, nrows):
        """-> dict(offset u64[B+1], index u64[B*39], value None, label f32[B])"""
        idx = np.empty((nrows, NUM_SLOTS), np.uint64)
        for g in range(NUM_SLOTS):
            u = self.rng.random(nrows)
            ranks = np.searchsorted(self._slot_cdf(g), u, side="left")
            idx[:, g] = self.ids_of(g, ranks)
        label = (self.rng.random(nrows) < self.pos_rate).astype(np.float32)
        offset = (np.arange(nrows + 1, dtype=np.uint64) * np.uint64(NUM_SLOTS))
        return dict(offset=offset, index=idx.reshape(-1), value=None, label=label)
