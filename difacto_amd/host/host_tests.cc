/**
 * host_tests.cc — interface-level tests of the C++ host side, the reference's
 * own gtest cases re-hosted on a tiny harness (no gtest in this image):
 *   FMLoss.NoV / FMLoss.HasV        tests/cpp/fm_loss_test.cc:12-83
 *   Localizer.Base / BaseHash       tests/cpp/localizer_test.cc:12-49
 *   SGDLearner.Basic                tests/cpp/sgd_learner_test.cc:9-49  (fused and literal worker loops)
 *   BatchReader.Read / RandRead / PartRead   tests/cpp/batch_reader_test.cc:9-57
 *   LBFGSLearner.Basic / WithV      tests/cpp/lbfgs_learner_test.cc:8-146 (objective trajectories of an
 *                                   L-BFGS loop over Loss::Predict / CalcGrad / Evaluate — lbfgs_mini.h)
 * plus Store Pull/Push and Updater Save/Load round trips.  Needs a GPU, except the reader cases.
 * usage: difacto_host_tests <path to rcv1_100.libsvm> [reader]     ("reader": only the host-only cases)
 */
#include <cmath>
#include <cstdio>
#include <memory>
#include "./device_store.h"
#include "./hip_fm_loss.h"
#include "./host_localizer.h"
#include "./lbfgs_mini.h"
#include "./libsvm_reader.h"
#include "./sgd_learner.h"
#include "dmlc/memory_io.h"

using namespace difacto;

static int g_fail = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ++g_fail;                                                              \
      fprintf(stderr, "  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
    }                                                                        \
  } while (0)

static std::string g_data;

static void load_data(dmlc::data::RowBlockContainer<unsigned>* data, std::vector<feaid_t>* uidx,
                      std::vector<real_t>* freq = nullptr, feaid_t max_index = ~0ULL) {
  LibsvmBatchReader reader(g_data, 0, 1, 100);
  CHECK(reader.Next());
  Localizer lc(max_index);
  lc.Compact(reader.Value(), data, uidx, freq);
  for (auto& i : *uidx) i = ReverseBytes(i);
}

static double logit_objv(const float* label, const SArray<real_t>& pred) {
  double o = 0;
  for (size_t i = 0; i < pred.size(); ++i) {
    double y = label[i] > 0 ? 1 : -1;
    o += std::log(1 + std::exp(-y * pred[i]));
  }
  return o;
}
static double norm2(const SArray<real_t>& v) {
  double n = 0;
  for (auto x : v) n += (double)x * x;
  return n;
}

static void TestLocalizer() {
  dmlc::data::RowBlockContainer<unsigned> c;
  std::vector<feaid_t> uidx;
  std::vector<real_t> freq;
  load_data(&c, &uidx, &freq);
  uint64_t s = 0;
  double f = 0;
  for (auto i : uidx) s += i;
  for (auto x : freq) f += x;
  EXPECT(s == 65111856ULL);
  EXPECT(f == 9648);
  load_data(&c, &uidx, &freq, 1000);
  s = 0;
  for (auto i : uidx) s += i;
  EXPECT(s == 478817ULL);
}

// RefRand is glibc's rand() from its default state, RefRand::Shuffle libstdc++'s std::random_shuffle over it
static void TestRefRand() {
  srand(1);
  RefRand g;
  for (int i = 0; i < 5000; ++i) EXPECT(g.Next() == rand());
  srand(12345);
  g.Seed(12345);
  for (int i = 0; i < 1000; ++i) EXPECT(g.Next() == rand());
  std::vector<unsigned> a(1000), b(1000);
  for (unsigned i = 0; i < 1000; ++i) a[i] = b[i] = i;
  srand(1);
  g.Seed(1);
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wdeprecated-declarations"
  std::random_shuffle(a.begin(), a.end());
#pragma GCC diagnostic pop
  g.Shuffle(&b);
  EXPECT(a == b);
}

// the minibatch checksums of tests/cpp/batch_reader_test.cc:9-57 (batch_size 37 over the 100 rows)
static void TestBatchReader() {
  const int label[] = {11, 15, 10};
  const int len[] = {37, 37, 26};
  const size_t os[] = {85035, 63968, 31323};
  const uint32_t idx[] = {95285478, 70504854, 62972349};
  const float val[] = {37.0f, 37.0f, 26.0f};
  for (int shuffled = 0; shuffled < 2; ++shuffled) {
    LibsvmBatchReader reader(g_data, 0, 1, 37, shuffled ? 37 : 0);
    int i = 0;
    while (reader.Next()) {
      EXPECT(i < 3);
      if (i >= 3) break;
      const auto& b = reader.Value();
      const int size = static_cast<int>(b.size);
      float lab = 0, v2 = 0;
      size_t o = 0;
      uint32_t ix = 0;
      for (int r = 0; r < size; ++r) lab += b.label[r];
      for (int r = 0; r <= size; ++r) o += b.offset[r] - b.offset[0];
      const size_t nnz = b.offset[size] - b.offset[0];
      for (size_t j = 0; j < nnz; ++j) {
        ix += static_cast<uint32_t>(b.index[b.offset[0] + j]);
        const float x = b.value ? b.value[b.offset[0] + j] : 1.0f;
        v2 += x * x;
      }
      EXPECT(static_cast<int>(lab) == label[i]);
      EXPECT(size == len[i]);
      if (shuffled) EXPECT(o != os[i]); else EXPECT(o == os[i]);   // rows permuted inside the batch
      EXPECT(ix == idx[i]);
      EXPECT(std::fabs(val[i] - v2) <= 1e-4);
      ++i;
    }
    EXPECT(i == 3);
  }
  // PartRead: the second of two parts holds about half of the rows
  LibsvmBatchReader part(g_data, 1, 2, 37);
  int ttl = 0;
  while (part.Next()) ttl += static_cast<int>(part.Value().size);
  EXPECT(ttl >= 40 && ttl <= 60);
}

static void TestFMLossNoV() {
  dmlc::data::RowBlockContainer<unsigned> rowblk;
  std::vector<feaid_t> uidx;
  load_data(&rowblk, &uidx);
  SArray<real_t> w(uidx.size());
  for (size_t i = 0; i < uidx.size(); ++i) w[i] = uidx[i] / 5e4;
  std::unique_ptr<Loss> loss(Loss::Create("fm"));
  loss->Init({{"V_dim", "0"}});
  auto data = rowblk.GetBlock();
  SArray<real_t> pred(data.size);
  static_cast<HipFMLoss*>(loss.get())->Predict(data, w, {}, {}, &pred);
  EXPECT(std::fabs(logit_objv(data.label, pred) - 147.4672) < 1e-3);
  EXPECT(std::fabs(loss->Evaluate(data.label, pred) - 147.4672) < 1e-3);
  SArray<real_t> grad(w.size());
  static_cast<HipFMLoss*>(loss.get())->CalcGrad(data, w, {}, {}, pred, &grad);
  EXPECT(std::fabs(norm2(grad) - 90.5817) < 1e-3);
}

static void TestFMLossHasV() {
  const int V_dim = 5;
  dmlc::data::RowBlockContainer<unsigned> rowblk;
  std::vector<feaid_t> uidx;
  load_data(&rowblk, &uidx);
  SArray<int> w_pos(uidx.size()), V_pos(uidx.size());
  SArray<real_t> w(uidx.size() * (V_dim + 1));
  int p = 0;
  for (size_t i = 0; i < uidx.size(); ++i) {
    w[i * (V_dim + 1)] = uidx[i] / 5e4;
    for (int j = 1; j <= V_dim; ++j) w[i * (V_dim + 1) + j] = uidx[i] * j / 5e5;
    w_pos[i] = p;
    V_pos[i] = p + 1;
    p += V_dim + 1;
  }
  std::unique_ptr<Loss> loss(Loss::Create("fm"));
  auto remain = loss->Init({{"V_dim", std::to_string(V_dim)}, {"foo", "bar"}});
  EXPECT(remain.size() == 1 && remain[0].first == "foo");  // unknown kwargs are handed back
  auto data = rowblk.GetBlock();
  SArray<real_t> pred(data.size);
  // through the virtual interface with the reference's param packing (fm_loss.h:50-65)
  std::vector<SArray<char>> inputs = {SArray<char>(w), SArray<char>(w_pos), SArray<char>(V_pos)};
  loss->Predict(data, inputs, &pred);
  EXPECT(std::fabs(logit_objv(data.label, pred) - 330.628) < 1e-3);
  SArray<real_t> grad(w.size());
  inputs.push_back(SArray<char>(pred));
  loss->CalcGrad(data, inputs, &grad);
  EXPECT(std::fabs(norm2(grad) - 1.2378e3) < 1e-1);
}

// tests/cpp/lbfgs_learner_test.cc: the FM loss (with and without V) driven by L-BFGS through the Loss interface
static void TestLBFGSTrajectory(bool with_v) {
  dmlc::data::RowBlockContainer<unsigned> rowblk;
  std::vector<feaid_t> uidx;
  load_data(&rowblk, &uidx);
  auto data = rowblk.GetBlock();
  lbfgs_mini::Param P;
  P.m = 5;
  P.max_num_epochs = 19;
  P.init_alpha = 1;
  std::unique_ptr<Loss> loss(Loss::Create("fm"));
  if (!with_v) {
    const std::vector<real_t> objv = {34.603421, 12.655075, 5.224232, 2.713903, 1.290586, 0.645131, 0.317889,
                                      0.156723,  0.075331,  0.032091, 0.018044, 0.008562, 0.004336, 0.002132,
                                      0.001051,  0.000506,  0.000227, 0.000119, 0.000059};
    P.V_dim = 0;
    P.l2 = 0;
    loss->Init({{"V_dim", "0"}});
    auto got = lbfgs_mini::Run(loss.get(), data, uidx.size(), P);
    EXPECT(got.size() == objv.size());
    for (size_t i = 0; i < got.size() && i < objv.size(); ++i) EXPECT(std::fabs(got[i] - objv[i]) < 1e-5);
  } else {
    const std::vector<real_t> objv = {35.224265, 21.631514, 18.394319, 16.077692, 12.389012, 8.888516, 8.446880,
                                      8.146090,  8.023501,  7.981967,  7.955119,  7.937092,  7.922456, 7.880596,
                                      7.861660,  7.838057,  7.807892,  7.784401,  7.756756};
    P.V_dim = 5;
    P.l2 = .1f;
    P.V_l2 = .01f;
    P.rho = .5f;
    loss->Init({{"V_dim", "5"}});
    auto init = [](const std::vector<int>& lens, lbfgs_mini::Vec* vals) {  // lbfgs_learner_test.cc:130-141
      int n = 0;
      for (int l : lens) {
        for (int i = 0; i < l; ++i) {
          if (i > 0) {
            real_t v = l - 1;
            (*vals)[n] = (i - v / 2) * .01;
          }
          ++n;
        }
      }
    };
    auto got = lbfgs_mini::Run(loss.get(), data, uidx.size(), P, init);
    EXPECT(got.size() == objv.size());
    double worst = 0;
    for (size_t i = 0; i < got.size() && i < objv.size(); ++i) {
      worst = std::max(worst, std::fabs((double)got[i] - objv[i]));
      EXPECT(std::fabs(got[i] - objv[i]) < 1e-4);
    }
    printf("        LBFGS WithV: worst |objv - golden| = %.2e over %zu epochs\n", worst, got.size());
  }
}

static void TestSGDLearnerBasic(const char* path) {
  std::vector<real_t> objv = {69.314718, 69.314718, 67.151912, 61.414778, 56.244989, 53.218700, 51.248737,
                              49.846688, 48.650164, 47.698351, 46.924038, 46.388223, 45.970721, 45.499307,
                              45.102245, 44.798413, 44.565211, 44.386417, 44.240657, 44.109764};
  SGDLearner learner;
  KWArgs args = {{"data_in", g_data}, {"V_dim", "0"}, {"l2", "1"}, {"l1", "1"}, {"lr", "1"},
                 {"num_jobs_per_epoch", "1"}, {"batch_size", "100"}, {"max_num_epochs", "20"},
                 {"device_path", path}, {"table_capacity", "65536"},
                 // with the reference's default stop_rel_objv = 1e-5 the run ends after epoch 1 (epochs 0 and
                 // 1 have the same loss: FTRL keeps w at 0 while |z| <= l1), in the reference as here, and its
                 // test never looks at the other 18 values; disable the criterion to check all 20
                 {"stop_rel_objv", "-1"}};
  auto remain = learner.Init(args);
  EXPECT(remain.size() == 0);
  int seen = 0;
  learner.AddEpochEndCallback([&](int epoch, const sgd::Progress& train, const sgd::Progress& val) {
    EXPECT(std::fabs(objv[epoch] - train.loss) < 5e-5);
    EXPECT(train.nrows == 100);
    EXPECT(train.auc > 0);
    ++seen;
  });
  learner.Run();
  EXPECT(seen == 20);
}

static void TestStoreAndModelIO() {
  std::shared_ptr<DeviceSGDUpdater> up(new DeviceSGDUpdater());
  auto remain = up->Init({{"V_dim", "3"}, {"V_threshold", "0"}, {"lr", "0.5"}, {"l1", "0.01"}, {"table_capacity", "4096"},
                          {"V_init", "refrand"}, {"unknown_key", "1"}});
  EXPECT(remain.size() == 1);
  std::unique_ptr<Store> store(Store::Create());
  store->SetUpdater(up);
  SArray<feaid_t> keys = {ReverseBytes(3), ReverseBytes(1), ReverseBytes(2)};
  std::sort(keys.begin(), keys.end());
  SArray<real_t> cnt = {5, 6, 7};
  int done = 0;
  store->Wait(store->Push(keys, Store::kFeaCount, cnt, {}, [&done]() { ++done; }));
  SArray<real_t> vals;
  SArray<int> lens;
  store->Wait(store->Pull(keys, Store::kWeight, &vals, &lens, [&done]() { ++done; }));
  EXPECT(done == 2);
  EXPECT(vals.size() == 3 && lens.size() == 3 && lens[0] == 1);
  SArray<real_t> g = {1.f, -2.f, 3.f};
  store->Push(keys, Store::kGradient, g, lens);
  store->Pull(keys, Store::kWeight, &vals, &lens);
  EXPECT(vals.size() == 12 && lens[0] == 4 && lens[2] == 4);  // w left zero -> V allocated (lazy InitV)
  // Save with aux -> Load into a fresh updater -> identical Pull
  std::string buf;
  {
    dmlc::MemoryStringStream fo(&buf);
    up->Save(true, &fo);
  }
  std::shared_ptr<DeviceSGDUpdater> up2(new DeviceSGDUpdater());
  up2->Init({{"V_dim", "3"}, {"V_threshold", "0"}, {"lr", "0.5"}, {"l1", "0.01"}, {"table_capacity", "4096"}});
  {
    dmlc::MemoryStringStream fi(&buf);
    bool aux = false;
    up2->Load(&fi, &aux);
    EXPECT(aux);
  }
  SArray<real_t> vals2;
  SArray<int> lens2;
  up2->Get(keys, Store::kWeight, &vals2, &lens2);
  EXPECT(vals2.size() == vals.size());
  for (size_t i = 0; i < vals.size() && i < vals2.size(); ++i) EXPECT(vals[i] == vals2[i]);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <rcv1_100.libsvm>\n", argv[0]);
    return 2;
  }
  g_data = argv[1];
  if (argc > 2 && std::string(argv[2]) == "reader") {  // host-only cases: no device is touched
    TestRefRand();
    printf("[%s] %s\n", g_fail ? "FAILED" : "  OK  ", "RefRand = glibc rand() / std::random_shuffle");
    TestBatchReader();
    printf("[%s] %s\n", g_fail ? "FAILED" : "  OK  ", "BatchReader.Read+RandRead+PartRead");
    printf("%s\n", g_fail ? "SOME TESTS FAILED" : "ALL HOST TESTS PASSED");
    return g_fail ? 1 : 0;
  }
  struct { const char* name; std::function<void()> fn; } tests[] = {
      {"RefRand = glibc rand() / std::random_shuffle", TestRefRand},
      {"BatchReader.Read+RandRead+PartRead", TestBatchReader},
      {"Localizer.Base+BaseHash", TestLocalizer},
      {"FMLoss.NoV", TestFMLossNoV},
      {"FMLoss.HasV", TestFMLossHasV},
      {"SGDLearner.Basic[fused]", [] { TestSGDLearnerBasic("fused"); }},
      {"SGDLearner.Basic[literal]", [] { TestSGDLearnerBasic("literal"); }},
      {"Store+ModelIO", TestStoreAndModelIO},
      {"LBFGSLearner.Basic (HipFMLoss, V_dim 0)", [] { TestLBFGSTrajectory(false); }},
      {"LBFGSLearner.WithV (HipFMLoss, V_dim 5)", [] { TestLBFGSTrajectory(true); }},
  };
  for (auto& t : tests) {
    int before = g_fail;
    t.fn();
    printf("[%s] %s\n", g_fail == before ? "  OK  " : "FAILED", t.name);
  }
  printf("%s\n", g_fail ? "SOME TESTS FAILED" : "ALL HOST TESTS PASSED");
  return g_fail ? 1 : 0;
}
