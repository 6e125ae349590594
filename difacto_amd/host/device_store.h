/**
 * device_store.h — the model store on the GPU.
 *
 *   DeviceSGDUpdater : Updater   the reference's SGDUpdater (src/sgd/sgd_updater.{h,cc}:
 *                                FTRL on w, AdaGrad on V, lazy InitV) with its
 *                                unordered_map replaced by a row table in HBM (dfh_table)
 *   DeviceStore : Store          the reference's StoreLocal (src/store/store_local.h):
 *                                synchronous in-process Push -> Updater::Update,
 *                                Pull -> Updater::Get, then the callback
 *
 * Both keep the reference's signatures and ragged (vals, lens) layout, so any
 * caller of Store::Push/Pull works unchanged; the SGD learner additionally
 * reaches the table itself for the fused on-device step.
 */
#ifndef DIFACTO_HOST_DEVICE_STORE_H_
#define DIFACTO_HOST_DEVICE_STORE_H_
#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "./device_context.h"
#include "./sgd_param.h"
#include "difacto/store.h"
#include "difacto/updater.h"

namespace difacto {

class DeviceSGDUpdater : public Updater {
 public:
  DeviceSGDUpdater() {}
  virtual ~DeviceSGDUpdater() {
    if (table_) dfh_table_destroy(table_);
  }

  KWArgs Init(const KWArgs& kwargs) override {
    auto remain = param_.InitAllowUnknown(kwargs);
    remain = dev_.InitAllowUnknown(remain);
    CHECK(dev_.V_init == "hash" || dev_.V_init == "refrand") << "V_init must be hash or refrand";
    dfh_updater_param p;
    dfh_updater_param_default(&p, param_.V_dim);
    p.l1 = param_.l1; p.l2 = param_.l2; p.V_l2 = param_.V_l2;
    p.lr = param_.lr; p.lr_beta = param_.lr_beta; p.V_lr = param_.V_lr; p.V_lr_beta = param_.V_lr_beta;
    p.V_init_scale = param_.V_init_scale;
    p.V_threshold = param_.V_threshold;
    p.seed = param_.seed;
    p.init_mode = dev_.V_init == "hash" ? DFH_INIT_HASH : DFH_INIT_REFRAND;
    DFH_CALL(dfh_table_create(DeviceContext::Get(), &p, dev_.table_capacity, &table_));
    return remain;
  }

  /**
   * Model file: "DFHM", u32 version, i32 V_dim, i32 has_aux, u64 n, then n entries
   * {u64 key, f32 w, [f32 fea_cnt, sqrt_g, z if aux], i32 has_V, [V_dim f32 V, [V_dim f32 acc if aux]]}.
   * (Load/Save are TODO stubs in the reference, src/sgd/sgd_updater.h:44-50: the format is ours.)
   */
  void Load(dmlc::Stream* fi, bool* has_aux) override {
    CHECK_NOTNULL(fi);
    char magic[4];
    uint32_t ver;
    int32_t k, aux;
    uint64_t n;
    CHECK_EQ(fi->Read(magic, 4), 4u);
    CHECK(!memcmp(magic, "DFHM", 4)) << "not a difacto-hip model file";
    CHECK(fi->ReadPOD(&ver) && fi->ReadPOD(&k) && fi->ReadPOD(&aux) && fi->ReadPOD(&n));
    CHECK_EQ(k, param_.V_dim) << "model V_dim differs from the configured one";
    if (has_aux) *has_aux = aux != 0;
    const size_t kk = static_cast<size_t>(std::max(k, 1));
    // entries are imported in chunks: the header's n only bounds the loop, it never sizes a buffer
    // (a corrupt count ends in a failed read, not in an allocation)
    const size_t chunk = 1 << 20;
    std::vector<uint64_t> keys;
    std::vector<float> scal, V, row(2 * kk);
    std::vector<int> hasv;
    auto flush = [&]() {
      if (keys.empty()) return;
      DFH_CALL(dfh_table_import(table_, keys.size(), keys.data(), scal.data(), hasv.data(), V.data()));
      keys.clear(); scal.clear(); V.clear(); hasv.clear();
    };
    for (uint64_t i = 0; i < n; ++i) {
      uint64_t key;
      float w, cnt = 0, sg = 0, z = 0;
      int32_t hv;
      CHECK(fi->ReadPOD(&key) && fi->ReadPOD(&w)) << "truncated model file (entry " << i << " of " << n << ")";
      if (aux) CHECK(fi->ReadPOD(&cnt) && fi->ReadPOD(&sg) && fi->ReadPOD(&z)) << "truncated model file";
      CHECK(fi->ReadPOD(&hv)) << "truncated model file";
      std::fill(row.begin(), row.end(), 0.f);
      if (hv && k > 0) {
        CHECK_EQ(fi->Read(row.data(), sizeof(float) * k), sizeof(float) * k) << "truncated model file";
        if (aux) CHECK_EQ(fi->Read(row.data() + k, sizeof(float) * k), sizeof(float) * k) << "truncated model file";
      }
      keys.push_back(key);
      scal.push_back(cnt); scal.push_back(w); scal.push_back(sg); scal.push_back(z);
      hasv.push_back(hv);
      V.insert(V.end(), row.begin(), row.begin() + 2 * kk);
      if (keys.size() == chunk) flush();
    }
    flush();
    // SGDUpdater::has_aux_ (sgd_updater.h:80): without optimiser state the table refuses gradient
    // pushes and training steps, as the reference's CHECK(has_aux_) does (sgd_updater.cc:75)
    if (!aux && n > 0) DFH_CALL(dfh_table_set_has_aux(table_, 0));
  }

  void Save(bool save_aux, dmlc::Stream* fo) const override {
    CHECK_NOTNULL(fo);
    uint64_t n = 0;
    DFH_CALL(dfh_table_size(table_, &n));
    const int k = param_.V_dim;
    const size_t kk = static_cast<size_t>(std::max(k, 1));
    const uint64_t cap = std::max<uint64_t>(n, 1);
    std::vector<uint64_t> keys(cap);
    std::vector<float> scal(cap * 4), V(cap * 2 * kk);
    std::vector<int> hasv(cap);
    uint64_t m = 0;
    DFH_CALL(dfh_table_export(table_, cap, keys.data(), scal.data(), hasv.data(), V.data(), &m));
    const uint32_t ver = 1;
    const int32_t kv = k, aux = save_aux ? 1 : 0;
    // entries with w == 0 and no V carry nothing a predictor needs: skipped unless aux is wanted
    uint64_t kept = 0;
    for (uint64_t i = 0; i < m; ++i) kept += (save_aux || scal[i * 4 + 1] != 0 || hasv[i]) ? 1 : 0;
    fo->Write("DFHM", 4);
    fo->WritePOD(ver); fo->WritePOD(kv); fo->WritePOD(aux); fo->WritePOD(kept);
    for (uint64_t i = 0; i < m; ++i) {
      if (!(save_aux || scal[i * 4 + 1] != 0 || hasv[i])) continue;
      fo->WritePOD(keys[i]);
      fo->WritePOD(scal[i * 4 + 1]);
      if (save_aux) { fo->WritePOD(scal[i * 4 + 0]); fo->WritePOD(scal[i * 4 + 2]); fo->WritePOD(scal[i * 4 + 3]); }
      const int32_t hv = hasv[i];
      fo->WritePOD(hv);
      if (hv && k > 0) {
        fo->Write(&V[i * 2 * k], sizeof(float) * k);
        if (save_aux) fo->Write(&V[i * 2 * k + k], sizeof(float) * k);
      }
    }
  }

  /*! \brief reference: SGDUpdater::Get, sgd_updater.cc:32-56 */
  void Get(const SArray<feaid_t>& fea_ids, int val_type, SArray<real_t>* weights, SArray<int>* lens) override {
    CHECK_EQ(val_type, Store::kWeight);
    CHECK_NOTNULL(weights);
    CHECK_NOTNULL(lens);
    const size_t n = fea_ids.size();
    weights->resize(n * (1 + param_.V_dim));
    lens->resize(n);
    size_t nvals = 0, nlens = 0;
    DFH_CALL(dfh_pull(table_, fea_ids.data(), n, weights->data(), &nvals, lens->data(), &nlens));
    weights->resize(nvals);
    lens->resize(nlens);
  }

  /*! \brief reference: SGDUpdater::Update, sgd_updater.cc:58-102 */
  void Update(const SArray<feaid_t>& fea_ids, int value_type, const SArray<real_t>& values,
              const SArray<int>& lens) override {
    DFH_CALL(dfh_push(table_, fea_ids.data(), fea_ids.size(), value_type, values.data(), values.size(), lens.data(),
                      lens.size()));
  }

  const SGDUpdaterParam& param() const { return param_; }
  const DeviceParam& device_param() const { return dev_; }
  dfh_table* table() { return table_; }

 private:
  SGDUpdaterParam param_;
  DeviceParam dev_;
  dfh_table* table_ = nullptr;
};

class DeviceStore : public Store {
 public:
  DeviceStore() : time_(0) {}
  virtual ~DeviceStore() {}
  KWArgs Init(const KWArgs& kwargs) override { return kwargs; }

  int Push(const SArray<feaid_t>& fea_ids, int val_type, const SArray<real_t>& vals, const SArray<int>& lens,
           const std::function<void()>& on_complete) override {
    CHECK_NOTNULL(updater_.get())->Update(fea_ids, val_type, vals, lens);  // synchronous: inputs are consumed on return
    if (on_complete) on_complete();
    return time_++;
  }
  int Pull(const SArray<feaid_t>& fea_ids, int val_type, SArray<real_t>* vals, SArray<int>* lens,
           const std::function<void()>& on_complete) override {
    CHECK_NOTNULL(updater_.get())->Get(fea_ids, val_type, vals, lens);
    if (on_complete) on_complete();
    return time_++;
  }
  void Wait(int time) override {}
  int Rank() override { return 0; }
  int NumWorkers() override { return 1; }
  int NumServers() override { return 1; }

 private:
  int time_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_DEVICE_STORE_H_
