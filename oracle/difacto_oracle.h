/*
 * oracle/difacto_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the reference's FM/SGD hot path
 * (dmlc/difacto).  Every function cites the reference file:line it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (difacto_amd/, include/difacto_hip.h)
 * must never call it.
 *
 * Pinning: tests/test_oracle_golden.py checks this restatement against every
 * golden vector the reference's own tests hold for the path
 * (tests/cpp/fm_loss_test.cc, localizer_test.cc, sgd_learner_test.cc) and,
 * when oracle/_ref (the reference's own sources compiled here) is present,
 * against the reference itself on random inputs.
 */
#ifndef DIFACTO_ORACLE_H_
#define DIFACTO_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* value types of Store::Push/Pull — include/difacto/store.h:31-33 */
enum { ORC_FEA_COUNT = 1, ORC_WEIGHT = 2, ORC_GRADIENT = 3 };

/* how a new embedding row is filled */
enum {
  ORC_INIT_REFRAND = 0, /* glibc rand_r chain on a mutated seed: sgd_updater.cc:140-147 */
  ORC_INIT_HASH = 1     /* counter-based hash of (key, j, seed): the product's sharding-independent init */
};

/* SGDUpdaterParam — src/sgd/sgd_param.h:66-107 (same names, same defaults) */
typedef struct {
  float l1, l2, V_l2;
  float lr, lr_beta, V_lr, V_lr_beta;
  float V_init_scale;
  int V_dim;
  int V_threshold;
  unsigned seed;
  int init_mode; /* ORC_INIT_* (not in the reference) */
} orc_updater_param;

void orc_updater_param_default(orc_updater_param* p, int V_dim);

/* ---- a1: id transforms — include/difacto/base.h:39-73 ---- */
uint64_t orc_reverse_bytes(uint64_t x);
uint64_t orc_encode_fea_grp_id(uint64_t x, int gid, int nbits);
uint64_t orc_decode_fea_grp_id(uint64_t x, int nbits);

/* glibc rand_r restated (the reference calls libc's; the device needs the
 * formula) and the product's hash init; see difacto_oracle.c */
int orc_rand_r(unsigned* seed);
float orc_hash_init_value(uint64_t key, int j, unsigned seed, float scale);
uint64_t orc_splitmix64(uint64_t x);

/* ---- a2: Localizer::Compact — src/data/localizer.cc:11-103 ----
 * index: raw u64 ids (nnz = offset[nrows]-offset[0]); outputs caller-allocated:
 * uniq[nnz], cnt[nnz] (may be NULL), out_index[nnz], out_offset[nrows+1],
 * sorted_pos[nnz] (may be NULL: nnz positions ordered by key, ties by position).
 * returns U = number of unique keys. */
size_t orc_localize(size_t nrows, const size_t* offset, const uint64_t* index, uint64_t max_index,
                    uint64_t* uniq, float* cnt, uint32_t* out_index, size_t* out_offset,
                    uint32_t* sorted_pos);

/* ---- a4/a9/a10: SGDUpdater behind StoreLocal ---- */
typedef struct orc_store orc_store;
orc_store* orc_store_create(const orc_updater_param* p);
void orc_store_destroy(orc_store* s);
size_t orc_store_size(const orc_store* s);
/* Store::Pull(kWeight) -> SGDUpdater::Get — src/sgd/sgd_updater.cc:32-56.
 * vals capacity n*(1+V_dim); lens capacity n.  *nlens = 0 iff V_dim == 0. */
void orc_store_pull(orc_store* s, const uint64_t* keys, size_t n, float* vals, size_t* nvals,
                    int* lens, size_t* nlens);
/* Store::Push -> SGDUpdater::Update — src/sgd/sgd_updater.cc:58-102.
 * returns 0, or -1 on the conditions where the reference CHECK-fails. */
int orc_store_push(orc_store* s, const uint64_t* keys, size_t n, int val_type, const float* vals,
                   size_t nvals, const int* lens, size_t nlens);
/* test helper: read / overwrite one entry {fea_cnt,w,sqrt_g,z} + V[2k] (has_V 0/1) */
int orc_store_peek(orc_store* s, uint64_t key, float* scal4, float* V2k, int* has_V);
void orc_store_poke(orc_store* s, uint64_t key, const float* scal4, const float* V2k, int has_V);

/* ---- a5: SGDLearner::GetPos — src/sgd/sgd_learner.cc:113-127 ---- */
void orc_get_pos(const int* lens, size_t n, int* w_pos, int* V_pos);

/* ---- a6: FMLoss::Predict — src/loss/fm_loss.h:67-119.
 * pred is accumulated into (caller zeroes it).  w_pos/V_pos may be NULL
 * (dense: w_pos[i]=i; no V).  XV_out (nrows*V_dim) optional. */
void orc_fm_predict(int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                    const float* value, const float* weights, const int* w_pos, const int* V_pos,
                    size_t npos, float* pred, float* XV_out);

/* ---- a8: FMLoss::CalcGrad — src/loss/fm_loss.h:148-199.
 * grad (same shape as weights) is accumulated into (caller zeroes it). */
void orc_fm_calcgrad(int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                     const float* value, const float* label, const float* weights,
                     size_t nweights, const int* w_pos, const int* V_pos, size_t npos,
                     const float* pred, float* grad);

/* ---- a7: Loss::Evaluate — include/difacto/loss.h:57-66 ---- */
float orc_loss_evaluate(const float* label, const float* pred, size_t n); /* nthreads = 2 */
float orc_loss_evaluate_nt(const float* label, const float* pred, size_t n, int nthreads);

/* ---- a12: BinClassMetric::AUC (src/loss/bin_class_metric.h:35-56, returns AUC*n),
 * SGDLearner::EvaluatePenalty (src/sgd/sgd_learner.cc:249-273) ---- */
float orc_auc_times_n(const float* label, const float* pred, size_t n);
float orc_evaluate_penalty(const orc_updater_param* p, const float* weights, size_t nweights,
                           const int* w_pos, const int* V_pos, size_t npos);

/* sgd::Progress — src/sgd/sgd_utils.h:40-75 */
typedef struct {
  float loss, penalty, auc, nnz_w, nrows;
} orc_progress;

/* ---- the whole worker step of SGDLearner::IterateData's batch executor
 * (src/sgd/sgd_learner.cc:131-178) on an already-localized batch:
 * [push fea counts] -> pull -> GetPos -> Predict -> Evaluate/penalty/AUC ->
 * CalcGrad -> push gradients.  pred_out (nrows) optional. */
void orc_sgd_step(orc_store* s, size_t nrows, const size_t* offset, const uint32_t* index,
                  const float* value, const float* label, const uint64_t* feaids, size_t U,
                  const float* feacnt /* NULL unless epoch-0 training */, int is_train,
                  orc_progress* prog, float* pred_out);

#ifdef __cplusplus
}
#endif
#endif /* DIFACTO_ORACLE_H_ */
