/*
 * oracle/difacto_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's FM/SGD worker path; see
 * difacto_oracle.h for the contract and the pinning story.  Arithmetic is kept
 * in the reference's evaluation order and in its types (float unless the
 * reference's expression promotes to double); compile with
 * -ffp-contract=off -fno-fast-math.
 */
#include "difacto_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ a1 */

/* include/difacto/base.h:39-51: swap 32/16/8/4-bit groups => nibble reversal */
uint64_t orc_reverse_bytes(uint64_t x) {
  x = x << 32 | x >> 32;
  x = (x & 0x0000FFFF0000FFFFULL) << 16 | (x & 0xFFFF0000FFFF0000ULL) >> 16;
  x = (x & 0x00FF00FF00FF00FFULL) << 8 | (x & 0xFF00FF00FF00FF00ULL) >> 8;
  x = (x & 0x0F0F0F0F0F0F0F0FULL) << 4 | (x & 0xF0F0F0F0F0F0F0F0ULL) >> 4;
  return x;
}

/* include/difacto/base.h:60-63 */
uint64_t orc_encode_fea_grp_id(uint64_t x, int gid, int nbits) { return (x << nbits) | (uint64_t)gid; }

/* include/difacto/base.h:71-73 */
uint64_t orc_decode_fea_grp_id(uint64_t x, int nbits) { return x % (1ULL << nbits); }

/* glibc's rand_r (stdlib/rand_r.c), which src/sgd/sgd_updater.cc:144 calls:
 * three steps of the LCG x <- 1103515245 x + 12345 (mod 2^32), taking 11, 10
 * and 10 bits.  Restated so the device can reproduce the chain. */
int orc_rand_r(unsigned* seed) {
  unsigned next = *seed;
  int result;
  next = next * 1103515245u + 12345u;
  result = (int)((next / 65536u) % 2048u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  next = next * 1103515245u + 12345u;
  result <<= 10;
  result ^= (int)((next / 65536u) % 1024u);
  *seed = next;
  return result;
}

uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* The product's sharding-independent V init (a documented deviation from
 * sgd_updater.cc:140-147, whose rand_r chain depends on global call order):
 * V[j] = (u - 0.5) * scale with u a 24-bit uniform from a hash of (key,seed,j).
 * Same amplitude (+-scale/2) as the reference. */
float orc_hash_init_value(uint64_t key, int j, unsigned seed, float scale) {
  uint64_t a = orc_splitmix64(key ^ (0xD1B54A32D192ED03ULL * ((uint64_t)seed + 1ULL)));
  uint64_t h = orc_splitmix64(a + (uint64_t)j);
  uint32_t r = (uint32_t)(h >> 40);
  float u = (float)r * (1.0f / 16777216.0f);
  return (u - 0.5f) * scale;
}

void orc_updater_param_default(orc_updater_param* p, int V_dim) {
  /* src/sgd/sgd_param.h:95-105 */
  p->l1 = 1.0f;
  p->l2 = 0.0f;
  p->V_l2 = 0.01f;
  p->lr = 0.01f;
  p->lr_beta = 1.0f;
  p->V_lr = 0.01f;
  p->V_lr_beta = 1.0f;
  p->V_init_scale = 0.01f;
  p->V_threshold = 10;
  p->V_dim = V_dim;
  p->seed = 0;
  p->init_mode = ORC_INIT_REFRAND;
}

/* ------------------------------------------------------------------ a2 */

typedef struct {
  uint64_t k;
  uint32_t i;
} orc_pair;

static int pair_cmp(const void* a, const void* b) {
  const orc_pair* x = (const orc_pair*)a;
  const orc_pair* y = (const orc_pair*)b;
  if (x->k != y->k) return x->k < y->k ? -1 : 1;
  /* the reference's sort is unstable (localizer.cc:28 compares keys only); no
   * output depends on the order among equal keys.  Tie-break by position to
   * make sorted_pos deterministic. */
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}

size_t orc_localize(size_t nrows, const size_t* offset, const uint64_t* index, uint64_t max_index,
                    uint64_t* uniq, float* cnt, uint32_t* out_index, size_t* out_offset,
                    uint32_t* sorted_pos) {
  /* CountUniqIndex — src/data/localizer.cc:11-50 */
  if (nrows == 0) return 0; /* :17 */
  /* the reference indexes blk.index[0 .. offset[size]) (:18,:24): offset[0]==0 assumed */
  size_t nnz = offset[nrows];
  if (nnz == 0) { /* reference would read pair_[0] of an empty vector; define as empty */
    for (size_t i = 0; i <= nrows; ++i) out_offset[i] = 0;
    return 0;
  }
  orc_pair* pair = (orc_pair*)malloc(nnz * sizeof(orc_pair));
  for (size_t i = 0; i < nnz; ++i) {
    pair[i].k = orc_reverse_bytes(index[i] % max_index); /* :24 */
    pair[i].i = (uint32_t)i;
  }
  qsort(pair, nnz, sizeof(orc_pair), pair_cmp); /* :27-28 */
  size_t U = 0;
  uint64_t curr = pair[0].k; /* :35 */
  float c = 0;
  for (size_t i = 0; i < nnz; ++i) { /* :37-46 */
    if (pair[i].k != curr) {
      uniq[U] = curr;
      if (cnt) cnt[U] = c;
      ++U;
      curr = pair[i].k;
      c = 0;
    }
    c += 1.0f;
  }
  uniq[U] = curr; /* :47-48 */
  if (cnt) cnt[U] = c;
  ++U;

  /* RemapIndex — src/data/localizer.cc:53-103 with idx_dict == uniq, so every
   * nnz matches (:66-77) and the CSR shape is unchanged (:88-96). */
  size_t d = 0;
  for (size_t i = 0; i < nnz; ++i) {
    while (uniq[d] < pair[i].k) ++d;
    out_index[pair[i].i] = (uint32_t)d; /* stored +1 then -1 in the reference (:71,:92) */
    if (sorted_pos) sorted_pos[i] = pair[i].i;
  }
  for (size_t i = 0; i <= nrows; ++i) out_offset[i] = offset[i] - offset[0];
  free(pair);
  return U;
}

/* ------------------------------------------------------- a4 / a9 / a10 */

/* SGDEntry — src/sgd/sgd_updater.h:19-29 */
typedef struct {
  uint64_t key;
  float fea_cnt;
  float w, sqrt_g, z;
  float* V; /* [V_dim values | V_dim accumulators] or NULL */
} orc_entry;

struct orc_store {
  orc_updater_param p;
  orc_entry* e;
  size_t n, cap;
  /* open-addressing index: slot -> entry index + 1 (0 = empty) */
  uint32_t* ht;
  size_t ht_cap; /* power of two */
};

static size_t ht_probe(const orc_store* s, uint64_t key) {
  size_t mask = s->ht_cap - 1;
  size_t h = (size_t)orc_splitmix64(key) & mask;
  while (s->ht[h] != 0 && s->e[s->ht[h] - 1].key != key) h = (h + 1) & mask;
  return h;
}

static void ht_grow(orc_store* s) {
  size_t ncap = s->ht_cap * 2;
  free(s->ht);
  s->ht = (uint32_t*)calloc(ncap, sizeof(uint32_t));
  s->ht_cap = ncap;
  for (size_t i = 0; i < s->n; ++i) {
    size_t h = ht_probe(s, s->e[i].key);
    s->ht[h] = (uint32_t)(i + 1);
  }
}

orc_store* orc_store_create(const orc_updater_param* p) {
  orc_store* s = (orc_store*)calloc(1, sizeof(orc_store));
  s->p = *p;
  s->cap = 1024;
  s->e = (orc_entry*)calloc(s->cap, sizeof(orc_entry));
  s->ht_cap = 4096;
  s->ht = (uint32_t*)calloc(s->ht_cap, sizeof(uint32_t));
  return s;
}

void orc_store_destroy(orc_store* s) {
  if (!s) return;
  for (size_t i = 0; i < s->n; ++i) free(s->e[i].V);
  free(s->e);
  free(s->ht);
  free(s);
}

size_t orc_store_size(const orc_store* s) { return s->n; }

/* model_[id]: std::unordered_map::operator[] default-constructs a zero entry
 * for an unseen key (sgd_updater.cc:44,:66,:87) */
static orc_entry* entry_of(orc_store* s, uint64_t key) {
  size_t h = ht_probe(s, key);
  if (s->ht[h]) return &s->e[s->ht[h] - 1];
  if (s->n == s->cap) {
    s->cap *= 2;
    s->e = (orc_entry*)realloc(s->e, s->cap * sizeof(orc_entry));
  }
  orc_entry* e = &s->e[s->n];
  memset(e, 0, sizeof(*e));
  e->key = key;
  s->ht[h] = (uint32_t)(++s->n);
  if (s->n * 2 > s->ht_cap) {
    ht_grow(s);
  }
  return &s->e[s->n - 1];
}

/* SGDUpdater::InitV — src/sgd/sgd_updater.cc:140-147 */
static void init_V(orc_store* s, orc_entry* e) {
  int n = s->p.V_dim;
  e->V = (float*)malloc(sizeof(float) * 2 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    if (s->p.init_mode == ORC_INIT_REFRAND) {
      /* (rand_r(&seed) / (real_t)RAND_MAX - 0.5) * V_init_scale  (:144):
       * int/float division in float, then "- 0.5" and "* scale" in double,
       * rounded to float on assignment. */
      float q = (float)orc_rand_r(&s->p.seed) / (float)RAND_MAX;
      e->V[i] = (float)(((double)q - 0.5) * (double)s->p.V_init_scale);
    } else {
      e->V[i] = orc_hash_init_value(e->key, i, s->p.seed, s->p.V_init_scale);
    }
  }
  memset(e->V + n, 0, sizeof(float) * (size_t)n); /* :146 */
}

/* SGDUpdater::Get — src/sgd/sgd_updater.cc:32-56 */
void orc_store_pull(orc_store* s, const uint64_t* keys, size_t n, float* vals, size_t* nvals,
                    int* lens, size_t* nlens) {
  int V_dim = s->p.V_dim;
  size_t p = 0;
  for (size_t i = 0; i < n; ++i) {
    orc_entry* e = entry_of(s, keys[i]); /* :44 inserts a zero entry if absent */
    vals[p++] = e->w;                    /* :46 */
    if (e->V) {                          /* :47-50 */
      memcpy(vals + p, e->V, sizeof(float) * (size_t)V_dim);
      p += (size_t)V_dim;
      lens[i] = V_dim + 1;
    } else if (V_dim != 0) { /* :51-53 */
      lens[i] = 1;
    }
  }
  *nvals = p;                      /* :55 */
  *nlens = V_dim == 0 ? 0 : n;     /* :40 */
}

/* SGDUpdater::UpdateW (FTRL-proximal) — src/sgd/sgd_updater.cc:104-127 */
static void update_w(orc_store* s, float gw, orc_entry* e) {
  const orc_updater_param* P = &s->p;
  float sg = e->sqrt_g;
  float w = e->w;
  gw += w * P->l2;                                /* :108 */
  e->sqrt_g = (float)sqrt((double)(sg * sg + gw * gw)); /* :109 — sqrt of a float expr */
  e->z -= gw - (e->sqrt_g - sg) / P->lr * w;      /* :111 */
  float z = e->z;
  float l1 = P->l1;
  if (z <= l1 && z >= -l1) { /* :115-116 */
    e->w = 0;
  } else {
    float eta = (P->lr_beta + e->sqrt_g) / P->lr; /* :118 */
    e->w = (z > 0 ? z - l1 : z + l1) / eta;       /* :119 */
  }
  if (w == 0 && e->w != 0) { /* :122-126 */
    if (P->V_dim > 0 && e->V == NULL && e->fea_cnt > (float)P->V_threshold) init_V(s, e);
  }
}

/* SGDUpdater::UpdateV (AdaGrad) — src/sgd/sgd_updater.cc:129-138 */
static void update_V(orc_store* s, const float* gV, orc_entry* e) {
  const orc_updater_param* P = &s->p;
  int n = P->V_dim;
  for (int i = 0; i < n; ++i) {
    float g = gV[i] + P->V_l2 * e->V[i];                        /* :132 */
    float cg = e->V[i + n];                                     /* :133 */
    e->V[i + n] = (float)sqrt((double)(cg * cg + g * g));       /* :134 */
    float eta = P->V_lr / (e->V[i + n] + P->V_lr_beta);         /* :135 */
    e->V[i] -= eta * g;                                         /* :136 */
  }
}

/* SGDUpdater::Update — src/sgd/sgd_updater.cc:58-102 */
int orc_store_push(orc_store* s, const uint64_t* keys, size_t n, int val_type, const float* vals,
                   size_t nvals, const int* lens, size_t nlens) {
  const orc_updater_param* P = &s->p;
  if (val_type == ORC_FEA_COUNT) { /* :62-73 */
    if (nvals != n) return -1;
    for (size_t i = 0; i < n; ++i) {
      orc_entry* e = entry_of(s, keys[i]);
      e->fea_cnt += vals[i];
      if (P->V_dim > 0 && e->V == NULL && e->w != 0 && e->fea_cnt > (float)P->V_threshold) {
        init_V(s, e);
      }
    }
    return 0;
  } else if (val_type == ORC_GRADIENT) { /* :74-97 */
    int w_only = nlens == 0; /* :77 */
    if (w_only) {
      if (nvals != n) return -1;
    } else if (nlens != n) {
      return -1;
    }
    size_t p = 0;
    for (size_t i = 0; i < n; ++i) {
      orc_entry* e = entry_of(s, keys[i]);
      update_w(s, vals[p++], e); /* :89 */
      if (!w_only && lens[i] > 1) { /* :90-95 */
        if (lens[i] != P->V_dim + 1) return -1;
        if (e->V == NULL) return -1;
        update_V(s, vals + p, e);
        p += (size_t)P->V_dim;
      }
    }
    if (p != nvals) return -1; /* :96 */
    return 0;
  }
  return -1; /* :99 LOG(FATAL) */
}

int orc_store_peek(orc_store* s, uint64_t key, float* scal4, float* V2k, int* has_V) {
  size_t h = ht_probe(s, key);
  if (!s->ht[h]) return 0;
  orc_entry* e = &s->e[s->ht[h] - 1];
  scal4[0] = e->fea_cnt;
  scal4[1] = e->w;
  scal4[2] = e->sqrt_g;
  scal4[3] = e->z;
  *has_V = e->V != NULL;
  if (e->V && V2k) memcpy(V2k, e->V, sizeof(float) * 2 * (size_t)s->p.V_dim);
  return 1;
}

void orc_store_poke(orc_store* s, uint64_t key, const float* scal4, const float* V2k, int has_V) {
  orc_entry* e = entry_of(s, key);
  e->fea_cnt = scal4[0];
  e->w = scal4[1];
  e->sqrt_g = scal4[2];
  e->z = scal4[3];
  if (has_V) {
    if (!e->V) e->V = (float*)malloc(sizeof(float) * 2 * (size_t)s->p.V_dim);
    memcpy(e->V, V2k, sizeof(float) * 2 * (size_t)s->p.V_dim);
  } else {
    free(e->V);
    e->V = NULL;
  }
}

/* ------------------------------------------------------------------ a5 */

/* SGDLearner::GetPos — src/sgd/sgd_learner.cc:113-127 */
void orc_get_pos(const int* lens, size_t n, int* w_pos, int* V_pos) {
  int p = 0;
  for (size_t i = 0; i < n; ++i) {
    int l = lens[i];
    w_pos[i] = l == 0 ? -1 : p;
    V_pos[i] = l > 1 ? p + 1 : -1;
    p += l;
  }
}

/* ------------------------------------------------------------------ a6 */

/* SpMM::Times (src/common/spmm.h:94-121): y[i,:] += sum_j D[i,j] * x[pos[idx_j] ...],
 * serial in nnz order, absent rows (pos -1) skipped */
static void spmm_times(size_t nrows, const size_t* offset, const uint32_t* index, const float* value,
                       const float* x, const int* x_pos, int k, float* y) {
  for (size_t i = 0; i < nrows; ++i) {
    if (offset[i] == offset[i + 1]) continue;
    float* y_i = y + i * (size_t)k;
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      int pj = x_pos[index[j]];
      if (pj == -1) continue;
      const float* x_j = x + pj;
      if (value) {
        float v = value[j];
        for (int l = 0; l < k; ++l) y_i[l] += x_j[l] * v;
      } else {
        for (int l = 0; l < k; ++l) y_i[l] += x_j[l];
      }
    }
  }
}

void orc_fm_predict(int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                    const float* value, const float* weights, const int* w_pos, const int* V_pos,
                    size_t npos, float* pred, float* XV_out) {
  (void)npos;
  /* pred += X * w : SpMV::Times (src/common/spmv.h:108-134) via fm_loss.h:74 */
  for (size_t i = 0; i < nrows; ++i) {
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      float x_j;
      if (w_pos) {
        int pj = w_pos[index[j]];
        x_j = pj == -1 ? 0.0f : weights[pj]; /* GetVal, spmv.h:174-181 */
      } else {
        x_j = weights[index[j]];
      }
      if (x_j == 0) continue; /* spmv.h:125 */
      if (value) {
        pred[i] += x_j * value[j];
      } else {
        pred[i] += x_j;
      }
    }
  }
  if (V_dim == 0) return; /* fm_loss.h:77 — NO clamp in this case */
  size_t k = (size_t)V_dim;
  size_t nnz = offset[nrows];

  /* XV = X*V (fm_loss.h:81-83) */
  float* XV = XV_out ? XV_out : (float*)malloc(sizeof(float) * (nrows * k + 1));
  memset(XV, 0, sizeof(float) * nrows * k);
  spmm_times(nrows, offset, index, value, weights, V_pos, V_dim, XV);

  /* XX = X.*X (fm_loss.h:86-92): only when values are present */
  float* XX = NULL;
  if (value) {
    XX = (float*)malloc(sizeof(float) * (nnz + 1));
    for (size_t j = 0; j < nnz; ++j) XX[j] = value[j] * value[j];
  }
  /* XXVV = XX * (V.*V) (fm_loss.h:95-105); VV is formed on the fly: the
   * reference squares V into a temporary with the same float multiply. */
  float* XXVV = (float*)calloc(nrows * k + 1, sizeof(float));
  for (size_t i = 0; i < nrows; ++i) {
    float* y_i = XXVV + i * k;
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      int pj = V_pos[index[j]];
      if (pj == -1) continue;
      const float* v = weights + pj;
      if (XX) {
        float xx = XX[j];
        for (size_t l = 0; l < k; ++l) y_i[l] += (v[l] * v[l]) * xx;
      } else {
        for (size_t l = 0; l < k; ++l) y_i[l] += v[l] * v[l];
      }
    }
  }
  /* pred += .5 * sum(XV.^2 - XXVV, 2)  (fm_loss.h:108-115); ".5 * s" is a
   * double product added to a float and rounded once */
  for (size_t i = 0; i < nrows; ++i) {
    const float* t = XV + i * k;
    const float* tt = XXVV + i * k;
    float s = 0;
    for (size_t j = 0; j < k; ++j) s += t[j] * t[j] - tt[j];
    pred[i] = (float)((double)pred[i] + .5 * (double)s);
  }
  /* projection (fm_loss.h:118) */
  for (size_t i = 0; i < nrows; ++i) pred[i] = pred[i] > 20 ? 20 : (pred[i] < -20 ? -20 : pred[i]);
  free(XXVV);
  free(XX);
  if (!XV_out) free(XV);
}

/* ------------------------------------------------------------------ a8 */

void orc_fm_calcgrad(int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                     const float* value, const float* label, const float* weights,
                     size_t nweights, const int* w_pos, const int* V_pos, size_t npos,
                     const float* pred, float* grad) {
  (void)nweights;
  /* p = -y ./ (1 + exp(y .* pred))  (fm_loss.h:157-161; std::exp on float) */
  float* p = (float*)malloc(sizeof(float) * (nrows + 1));
  for (size_t i = 0; i < nrows; ++i) {
    float y = label[i] > 0 ? 1.0f : -1.0f;
    p[i] = -y / (1 + expf(y * pred[i]));
  }
  /* grad_w += X' * p : SpMV::TransTimes (spmv.h:140-171) via fm_loss.h:164;
   * every column is accumulated in ascending row order */
  for (size_t i = 0; i < nrows; ++i) {
    float x_i = p[i];
    if (x_i == 0) continue; /* spmv.h:155 */
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      uint32_t c = index[j];
      float* y_j;
      if (w_pos) {
        int pc = w_pos[c];
        if (pc == -1) continue;
        y_j = grad + pc;
      } else {
        y_j = grad + c;
      }
      if (value) {
        *y_j += x_i * value[j];
      } else {
        *y_j += x_i;
      }
    }
  }
  if (V_dim == 0) { /* fm_loss.h:168 */
    free(p);
    return;
  }
  size_t k = (size_t)V_dim;

  /* XV as left by the preceding Predict on the same batch (fm_loss.h:174,191) */
  float* XV = (float*)calloc(nrows * k + 1, sizeof(float));
  spmm_times(nrows, offset, index, value, weights, V_pos, V_dim, XV);

  /* XXp = (X.*X)' * p  (fm_loss.h:171-178), dense over the npos columns */
  float* XXp = (float*)calloc(npos + 1, sizeof(float));
  for (size_t i = 0; i < nrows; ++i) {
    float x_i = p[i];
    if (x_i == 0) continue;
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      if (value) {
        XXp[index[j]] += x_i * (value[j] * value[j]);
      } else {
        XXp[index[j]] += x_i;
      }
    }
  }
  /* grad_V -= diag(XXp) * V  (fm_loss.h:181-188) */
  for (size_t u = 0; u < npos; ++u) {
    int pu = V_pos[u];
    if (pu < 0) continue;
    for (size_t j = 0; j < k; ++j) grad[(size_t)pu + j] -= weights[(size_t)pu + j] * XXp[u];
  }
  /* XV = diag(p) * XV  (fm_loss.h:192-195) */
  for (size_t i = 0; i < nrows; ++i)
    for (size_t j = 0; j < k; ++j) XV[i * k + j] *= p[i];
  /* grad_V += X' * XV : SpMM::TransTimes (spmm.h:128-159) via fm_loss.h:198 */
  for (size_t i = 0; i < nrows; ++i) {
    const float* x_i = XV + i * k;
    for (size_t j = offset[i]; j < offset[i + 1]; ++j) {
      int pc = V_pos[index[j]];
      if (pc == -1) continue;
      float* y_j = grad + pc;
      if (value) {
        float v = value[j];
        for (size_t l = 0; l < k; ++l) y_j[l] += x_i[l] * v;
      } else {
        for (size_t l = 0; l < k; ++l) y_j[l] += x_i[l];
      }
    }
  }
  free(XXp);
  free(XV);
  free(p);
}

/* ------------------------------------------------------------------ a7 */

/* Loss::Evaluate — include/difacto/loss.h:57-66 (libm exp/log on the
 * double-promoted float argument; float accumulators).  The reference sums
 * under "#pragma omp parallel for reduction(+:objv) num_threads(nthreads_)"
 * with nthreads_ = DEFAULT_NTHREADS = 2 (include/difacto/base.h:28,
 * src/sgd/sgd_learner.h:90): each thread accumulates a contiguous static chunk
 * in float, the partials are then added.  The float rounding of that shape is
 * visible at the 5e-5 tolerance of tests/cpp/sgd_learner_test.cc, so it is
 * restated here (gcc static schedule: the first n%nt threads get one extra). */
float orc_loss_evaluate_nt(const float* label, const float* pred, size_t n, int nthreads) {
  float total = 0;
  size_t q = n / (size_t)nthreads, r = n % (size_t)nthreads, begin = 0;
  for (int t = 0; t < nthreads; ++t) {
    size_t len = q + ((size_t)t < r ? 1 : 0);
    float objv = 0;
    for (size_t i = begin; i < begin + len; ++i) {
      float y = label[i] > 0 ? 1.0f : -1.0f;
      objv = (float)((double)objv + log(1 + exp((double)(-y * pred[i]))));
    }
    total += objv;
    begin += len;
  }
  return total;
}

float orc_loss_evaluate(const float* label, const float* pred, size_t n) {
  return orc_loss_evaluate_nt(label, pred, n, 2);
}

/* ----------------------------------------------------------------- a12 */

typedef struct {
  float label, predict;
} auc_entry;

static int auc_cmp(const void* a, const void* b) {
  float x = ((const auc_entry*)a)->predict, y = ((const auc_entry*)b)->predict;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* BinClassMetric::AUC — src/loss/bin_class_metric.h:35-56 (returns area * n).
 * Ties in predict are ordered arbitrarily by the reference's std::sort; this
 * restatement orders them as qsort leaves them (tests avoid tied predictions
 * or accept the tie ambiguity explicitly). */
float orc_auc_times_n(const float* label, const float* pred, size_t n) {
  auc_entry* buff = (auc_entry*)malloc(sizeof(auc_entry) * (n + 1));
  for (size_t i = 0; i < n; ++i) {
    buff[i].label = label[i];
    buff[i].predict = pred[i];
  }
  qsort(buff, n, sizeof(auc_entry), auc_cmp);
  float area = 0, cum_tp = 0;
  for (size_t i = 0; i < n; ++i) {
    if (buff[i].label > 0) {
      cum_tp += 1;
    } else {
      area += cum_tp;
    }
  }
  free(buff);
  if (cum_tp == 0 || cum_tp == (float)n) return 1; /* :51 */
  area /= cum_tp * ((float)n - cum_tp);
  return (area < 0.5 ? 1 - area : area) * (float)n;
}

/* SGDLearner::EvaluatePenalty — src/sgd/sgd_learner.cc:249-273.  The terms
 * "param.l1 * fabs(w) + .5 * param.l2 * w * w" are double expressions added
 * into a float accumulator. */
float orc_evaluate_penalty(const orc_updater_param* P, const float* weights, size_t nweights,
                           const int* w_pos, const int* V_pos, size_t npos) {
  float objv = 0;
  if (npos) {
    for (size_t i = 0; i < npos; ++i) {
      int p = w_pos[i];
      if (p == -1) continue;
      float w = weights[p];
      objv = (float)((double)objv + ((double)P->l1 * fabs((double)w) + .5 * (double)P->l2 * (double)w * (double)w));
    }
    for (size_t i = 0; i < npos; ++i) {
      int p = V_pos[i];
      if (p == -1) continue;
      for (int j = 0; j < P->V_dim; ++j) {
        float V = weights[p + j];
        objv = (float)((double)objv + .5 * (double)P->V_l2 * (double)V * (double)V);
      }
    }
  } else {
    for (size_t i = 0; i < nweights; ++i) {
      float w = weights[i];
      objv = (float)((double)objv + ((double)P->l1 * fabs((double)w) + .5 * (double)P->l2 * (double)w * (double)w));
    }
  }
  return objv;
}

/* ---------------------------------------------------- the worker step */

/* The batch executor of SGDLearner::IterateData — src/sgd/sgd_learner.cc:131-178,
 * preceded by the epoch-0 feature-count push (:201-202,:214-217). */
void orc_sgd_step(orc_store* s, size_t nrows, const size_t* offset, const uint32_t* index,
                  const float* value, const float* label, const uint64_t* feaids, size_t U,
                  const float* feacnt, int is_train, orc_progress* prog, float* pred_out) {
  int V_dim = s->p.V_dim;
  if (feacnt) orc_store_push(s, feaids, U, ORC_FEA_COUNT, feacnt, U, NULL, 0); /* :214-217 */

  float* values = (float*)malloc(sizeof(float) * (U * (size_t)(1 + V_dim) + 1));
  int* lens = (int*)malloc(sizeof(int) * (U + 1));
  size_t nvals, nlens;
  orc_store_pull(s, feaids, U, values, &nvals, lens, &nlens); /* :177 */

  int* w_pos = NULL;
  int* V_pos = NULL;
  if (nlens) { /* GetPos yields empty arrays when lens is empty (:116-118) */
    w_pos = (int*)malloc(sizeof(int) * (U + 1));
    V_pos = (int*)malloc(sizeof(int) * (U + 1));
    orc_get_pos(lens, U, w_pos, V_pos); /* :144 */
  }
  float* pred = (float*)calloc(nrows + 1, sizeof(float)); /* :142 */
  orc_fm_predict(V_dim, nrows, offset, index, value, values, w_pos, V_pos, nlens, pred, NULL); /* :147 */
  if (prog) {
    prog->nrows += (float)nrows;                                  /* :141 */
    prog->loss += orc_loss_evaluate(label, pred, nrows);          /* :148 */
    prog->penalty += orc_evaluate_penalty(&s->p, values, nvals, w_pos, V_pos, nlens); /* :150 */
    prog->auc += orc_auc_times_n(label, pred, nrows);             /* :153-155 */
  }
  if (pred_out) memcpy(pred_out, pred, sizeof(float) * nrows);
  if (is_train) { /* :158-168 */
    float* grads = (float*)calloc(nvals + 1, sizeof(float));
    orc_fm_calcgrad(V_dim, nrows, offset, index, value, label, values, nvals, w_pos, V_pos, nlens,
                    pred, grads);
    orc_store_push(s, feaids, U, ORC_GRADIENT, grads, nvals, lens, nlens);
    free(grads);
  }
  free(pred);
  free(w_pos);
  free(V_pos);
  free(lens);
  free(values);
}
