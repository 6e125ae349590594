/**
 * oracle/_ref C API  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin extern "C" wrapper around the REFERENCE's OWN classes, compiled from
 * the sources where they lie under /root/reference (never copied here):
 *   difacto::Localizer   src/data/localizer.{h,cc}
 *   difacto::SGDUpdater  src/sgd/sgd_updater.{h,cc}
 *   difacto::StoreLocal  src/store/store_local.h
 *   difacto::FMLoss      src/loss/fm_loss.h (+ common/spmv.h, common/spmm.h)
 *   difacto::BinClassMetric  src/loss/bin_class_metric.h
 * against the from-scratch dmlc/ps shims in third_party_shim/ (dmlc-core and
 * ps-lite are absent submodules; they contribute containers and macros only,
 * no arithmetic — SURVEY.md §8c).
 *
 * Used by tests/ to (1) pin the C restatement in oracle/difacto_oracle.c and
 * (2) as the "reference" CPU baseline in bench.py.  Nothing in the product
 * path may link or load this.
 */
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "difacto/base.h"
#include "difacto/loss.h"
#include "difacto/store.h"
#include "difacto/updater.h"
#include "data/localizer.h"
#include "loss/fm_loss.h"
#include "loss/bin_class_metric.h"
#include "sgd/sgd_updater.h"
#include "store/store_local.h"

namespace difacto {
// the reference registers these in src/loss/loss.cc and src/updater.cc, which
// also drag in the (out-of-scope) BCD / delta-loss code; register here instead.
DMLC_REGISTER_PARAMETER(FMLossParam);
DMLC_REGISTER_PARAMETER(SGDUpdaterParam);
}  // namespace difacto

using namespace difacto;

namespace {

KWArgs ParseKW(const char* s) {
  // "k=v;k=v"
  KWArgs kw;
  std::string str(s ? s : "");
  size_t p = 0;
  while (p < str.size()) {
    size_t e = str.find(';', p);
    if (e == std::string::npos) e = str.size();
    std::string item = str.substr(p, e - p);
    size_t q = item.find('=');
    if (q != std::string::npos) kw.push_back({item.substr(0, q), item.substr(q + 1)});
    p = e + 1;
  }
  return kw;
}

template <typename T>
SArray<T> View(const T* p, size_t n) {
  // non-owning view
  return SArray<T>(const_cast<T*>(p), n, false);
}

struct RefStore {
  StoreLocal store;
  std::shared_ptr<SGDUpdater> updater;
};

dmlc::RowBlock<unsigned> MakeBlock(size_t nrows, const size_t* offset, const unsigned* index,
                                   const float* value, const float* label) {
  dmlc::RowBlock<unsigned> b;
  b.size = nrows;
  b.offset = offset;
  b.index = index;
  b.value = value;
  b.label = label;
  b.weight = nullptr;
  return b;
}

}  // namespace

extern "C" {

uint64_t ref_reverse_bytes(uint64_t x) { return ReverseBytes(x); }
uint64_t ref_encode_fea_grp_id(uint64_t x, int gid, int nbits) { return EncodeFeaGrpID(x, gid, nbits); }

/**
 * Localizer::Compact (src/data/localizer.h:41-51).  Outputs are caller
 * allocated with capacity nnz (uniq, cnt, out_index) and nrows+1 (out_offset).
 * Returns the number of unique keys.
 */
size_t ref_localize(size_t nrows, const size_t* offset, const uint64_t* index, const float* value,
                    uint64_t max_index, int nthreads, uint64_t* uniq, float* cnt,
                    uint32_t* out_index, size_t* out_offset, float* out_value) {
  dmlc::RowBlock<feaid_t> blk;
  blk.size = nrows;
  blk.offset = offset;
  blk.index = index;
  blk.value = value;
  std::vector<float> fake_label(nrows, 0);
  blk.label = fake_label.data();
  dmlc::data::RowBlockContainer<unsigned> compact;
  std::vector<feaid_t> uidx;
  std::vector<real_t> freq;
  Localizer lc(max_index, nthreads);
  lc.Compact(blk, &compact, &uidx, cnt ? &freq : nullptr);
  memcpy(uniq, uidx.data(), uidx.size() * sizeof(feaid_t));
  if (cnt) memcpy(cnt, freq.data(), freq.size() * sizeof(real_t));
  if (compact.offset.size() == nrows + 1) {
    memcpy(out_offset, compact.offset.data(), (nrows + 1) * sizeof(size_t));
    memcpy(out_index, compact.index.data(), compact.index.size() * sizeof(unsigned));
    if (out_value && !compact.value.empty())
      memcpy(out_value, compact.value.data(), compact.value.size() * sizeof(float));
  }
  return uidx.size();
}

/** SGDUpdater behind StoreLocal (src/store/store_local.h:24-44) */
void* ref_store_create(const char* kwargs) {
  auto* s = new RefStore();
  s->updater.reset(new SGDUpdater());
  auto remain = s->updater->Init(ParseKW(kwargs));
  (void)remain;
  s->store.SetUpdater(s->updater);
  return s;
}
void ref_store_destroy(void* h) { delete static_cast<RefStore*>(h); }

/** Store::Pull(kWeight) -> SGDUpdater::Get (src/sgd/sgd_updater.cc:32-56).
 *  vals capacity n*(1+V_dim), lens capacity n. */
void ref_store_pull(void* h, const uint64_t* keys, size_t n, float* vals, size_t* nvals, int* lens,
                    size_t* nlens) {
  auto* s = static_cast<RefStore*>(h);
  SArray<real_t> v;
  SArray<int> l;
  s->store.Pull(View(keys, n), Store::kWeight, &v, &l, nullptr);
  memcpy(vals, v.data(), v.size() * sizeof(real_t));
  memcpy(lens, l.data(), l.size() * sizeof(int));
  *nvals = v.size();
  *nlens = l.size();
}

/** Store::Push -> SGDUpdater::Update (src/sgd/sgd_updater.cc:58-102) */
void ref_store_push(void* h, const uint64_t* keys, size_t n, int val_type, const float* vals,
                    size_t nvals, const int* lens, size_t nlens) {
  auto* s = static_cast<RefStore*>(h);
  s->store.Push(View(keys, n), val_type, View(vals, nvals), View(lens, nlens), nullptr);
}

/** FMLoss (src/loss/fm_loss.h) */
void* ref_fmloss_create(int V_dim, int nthreads) {
  auto* l = new FMLoss();
  l->Init({{"V_dim", std::to_string(V_dim)}});
  l->set_nthreads(nthreads);
  return l;
}
void ref_fmloss_destroy(void* h) { delete static_cast<FMLoss*>(h); }

/** FMLoss::Predict (src/loss/fm_loss.h:67-119); pred must arrive zeroed */
void ref_fmloss_predict(void* h, size_t nrows, const size_t* offset, const unsigned* index,
                        const float* value, const float* weights, size_t nweights,
                        const int* w_pos, const int* V_pos, size_t npos, float* pred) {
  auto* l = static_cast<FMLoss*>(h);
  auto blk = MakeBlock(nrows, offset, index, value, nullptr);
  SArray<real_t> p = View(pred, nrows);
  l->Predict(blk, View(weights, nweights), View(w_pos, w_pos ? npos : 0),
             View(V_pos, V_pos ? npos : 0), &p);
}

/** FMLoss::CalcGrad (src/loss/fm_loss.h:148-199); grad must arrive zeroed */
void ref_fmloss_calcgrad(void* h, size_t nrows, const size_t* offset, const unsigned* index,
                         const float* value, const float* label, const float* weights,
                         size_t nweights, const int* w_pos, const int* V_pos, size_t npos,
                         const float* pred, float* grad) {
  auto* l = static_cast<FMLoss*>(h);
  auto blk = MakeBlock(nrows, offset, index, value, label);
  SArray<real_t> g = View(grad, nweights);
  l->CalcGrad(blk, View(weights, nweights), View(w_pos, w_pos ? npos : 0),
              View(V_pos, V_pos ? npos : 0), View(pred, nrows), &g);
}

/** Loss::Evaluate (include/difacto/loss.h:57-66) */
float ref_loss_evaluate(void* h, const float* label, const float* pred, size_t n) {
  return static_cast<FMLoss*>(h)->Evaluate(label, View(pred, n));
}

/** BinClassMetric::AUC / LogitObjv (src/loss/bin_class_metric.h:35-56,83-91) */
float ref_auc(const float* label, const float* pred, size_t n) {
  BinClassMetric m(label, pred, n);
  return m.AUC();
}
float ref_logit_objv(const float* label, const float* pred, size_t n) {
  BinClassMetric m(label, pred, n);
  return m.LogitObjv();
}

}  // extern "C"
