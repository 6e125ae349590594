/**
 * sharded_store.h — ShardedDeviceStore : Store, the model row-sharded by key range over the GPUs
 * of one node, one process per GPU.  It takes the place of the reference's distributed store
 * (Store::Create() under DMLC_ROLE, src/store/store.cc:8-15 — KVStoreDist over ps-lite there):
 * every process is a worker AND the server of one key range; NumWorkers() is the number of ranks, so
 * SGDLearner::RunEpoch cuts the data into NumWorkers() x num_jobs_per_epoch parts as the reference
 * does (src/sgd/sgd_learner.cc:78-89).
 *
 * The exchange lives inside libdifacto_hip.so (dfh_shard_step: RCCL ncclSend / ncclRecv over xGMI);
 * this class owns the communicator and the shard handle and exposes them to the learner's sharded
 * worker loop.  The store is COLLECTIVE — every rank takes part in every step — so the literal,
 * one-sided Push / Pull of the interface is not offered here: use device_path = fused (the default).
 *
 * Process environment (one process per GPU; example/run_local_gpus.sh):
 *   DMLC_ROLE=worker            distributed mode (include/difacto/base.h: IsDistributed())
 *   DMLC_NUM_WORKER=<G>         ranks
 *   DIFACTO_RANK=<r>            this rank, 0 .. G-1 (also the default DIFACTO_DEVICE)
 *   DIFACTO_RENDEZVOUS=<path>   rank 0 writes the RCCL unique id there, the others read it
 *   DIFACTO_COMM=rccl|file      transport; "file" (exchange through files in the directory
 *                               DIFACTO_RENDEZVOUS) exists for tests with several ranks on one GPU
 */
#ifndef DIFACTO_HOST_SHARDED_STORE_H_
#define DIFACTO_HOST_SHARDED_STORE_H_
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
#include "./device_store.h"

namespace difacto {

/*! \brief test transport: an all-to-all-v through files in a shared directory */
class FileExchange {
 public:
  FileExchange(const std::string& dir, int rank, int world) : dir_(dir), rank_(rank), world_(world) {}
  static int Call(void* user, const void* send, const size_t* sb, void* recv, const size_t* rb) {
    return static_cast<FileExchange*>(user)->Run(static_cast<const char*>(send), sb, static_cast<char*>(recv), rb);
  }

 private:
  std::string Name(uint64_t seq, int src, int dst) const {
    return dir_ + "/ex" + std::to_string(seq) + "." + std::to_string(src) + "." + std::to_string(dst);
  }
  int Run(const char* send, const size_t* sb, char* recv, const size_t* rb) {
    const uint64_t seq = seq_++;
    size_t off = 0;
    for (int d = 0; d < world_; ++d) {
      const std::string fin = Name(seq, rank_, d), tmp = fin + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f) return 1;
      if (sb[d] && fwrite(send + off, 1, sb[d], f) != sb[d]) return 1;
      fclose(f);
      if (rename(tmp.c_str(), fin.c_str()) != 0) return 1;
      off += sb[d];
    }
    off = 0;
    for (int s = 0; s < world_; ++s) {
      const std::string fin = Name(seq, s, rank_);
      FILE* f = nullptr;
      for (int tries = 0; tries < 600000 && !(f = fopen(fin.c_str(), "rb")); ++tries)
        std::this_thread::sleep_for(std::chrono::microseconds(100));
      if (!f) return 1;
      if (rb[s] && fread(recv + off, 1, rb[s], f) != rb[s]) return 1;
      fclose(f);
      unlink(fin.c_str());
      off += rb[s];
    }
    return 0;
  }
  std::string dir_;
  int rank_, world_;
  uint64_t seq_ = 0;
};

class ShardedDeviceStore : public Store {
 public:
  ShardedDeviceStore() {
    const char* w = getenv("DMLC_NUM_WORKER");
    const char* r = getenv("DIFACTO_RANK");
    // In the reference's launch a scheduler and servers run beside the workers and park in tracker_->Wait().  Here
    // the workers own the model shards and schedule themselves: any other role joining would race the workers for
    // rank 0.  It is refused, not parked: a launcher that starts one is misconfigured for this build.
    const char* role = getenv("DMLC_ROLE");
    CHECK(!role || std::string(role) == "worker")
        << "DMLC_ROLE=" << role << ": this build runs workers only (each worker owns a key range of the model on its GPU and "
        << "runs the scheduler loop itself); start DMLC_NUM_WORKER worker processes, no scheduler, no servers";
    world_ = w ? atoi(w) : 1;
    CHECK(r || world_ <= 1) << "DMLC_NUM_WORKER=" << world_ << " needs DIFACTO_RANK (0.." << world_ - 1
                            << ") in every worker's environment: without it every process would come up as rank 0";
    rank_ = r ? atoi(r) : 0;
    CHECK(world_ >= 1 && world_ <= 32 && rank_ >= 0 && rank_ < world_)
        << "DMLC_NUM_WORKER (1..32) / DIFACTO_RANK (0..DMLC_NUM_WORKER-1) are not consistent";
    // one GPU per rank, unless the launcher already narrowed the visible devices to this rank's
    if (!getenv("DIFACTO_DEVICE") && !getenv("HIP_VISIBLE_DEVICES") && !getenv("ROCR_VISIBLE_DEVICES"))
      setenv("DIFACTO_DEVICE", std::to_string(rank_).c_str(), 0);
  }
  virtual ~ShardedDeviceStore() {
    if (shard_) dfh_shard_destroy(shard_);
    if (comm_) dfh_comm_destroy(comm_);
  }

  /*! \brief connects the ranks; called after SetUpdater (sgd_learner.cc:234-240), so the shard's table exists */
  KWArgs Init(const KWArgs& kwargs) override {
    auto* up = CHECK_NOTNULL(dynamic_cast<DeviceSGDUpdater*>(CHECK_NOTNULL(updater_.get())));
    CHECK(up->device_param().V_init == "hash" || up->param().V_dim == 0)
        << "the sharded store needs V_init=hash: the rand_r chain of the reference depends on the global order of allocations";
    const char* rv = getenv("DIFACTO_RENDEZVOUS");
    const char* kind = getenv("DIFACTO_COMM");
    CHECK(rv || world_ == 1) << "DIFACTO_RENDEZVOUS must name the rendezvous file (rccl) or directory (file transport)";
    dfh_ctx* ctx = DeviceContext::Get();
    if (kind && std::string(kind) == "file") {
      files_.reset(new FileExchange(rv ? rv : "/tmp", rank_, world_));
      DFH_CALL(dfh_comm_create_callback(ctx, rank_, world_, &FileExchange::Call, files_.get(), &comm_));
    } else {
      char id[DFH_COMM_ID_BYTES];
      if (rank_ == 0) {
        DFH_CALL(dfh_comm_unique_id(id));
        if (world_ > 1) {
          const std::string tmp = std::string(rv) + ".tmp";
          FILE* f = CHECK_NOTNULL(fopen(tmp.c_str(), "wb"));
          CHECK_EQ(fwrite(id, 1, sizeof(id), f), sizeof(id));
          fclose(f);
          CHECK_EQ(rename(tmp.c_str(), rv), 0);
        }
      } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 120000 && !(f = fopen(rv, "rb")); ++tries)
          std::this_thread::sleep_for(std::chrono::milliseconds(1));
        CHECK(f) << "rank 0 never wrote the rendezvous file " << rv;
        CHECK_EQ(fread(id, 1, sizeof(id), f), sizeof(id));
        fclose(f);
      }
      DFH_CALL(dfh_comm_create_rccl(ctx, rank_, world_, id, &comm_));
    }
    LOG(INFO) << "sharded store: rank " << rank_ << " of " << world_ << " connected ("
              << (files_ ? "file transport" : "RCCL") << ")";
    return kwargs;
  }

  /**
   * \brief cuts the key space and creates this rank's shard.  COLLECTIVE; called once by the learner after Init,
   * before a model is loaded.  sample: reversed keys of (the beginning of) this rank's part of the data, any number,
   * for shard_ranges = balanced (dfh_shard_balanced_splits: the quantiles of the union of all ranks' samples);
   * ignored for uniform ranges.
   */
  void CreateShard(const std::vector<feaid_t>& sample) {
    CHECK(!shard_) << "the shard exists already";
    auto* up = CHECK_NOTNULL(dynamic_cast<DeviceSGDUpdater*>(CHECK_NOTNULL(updater_.get())));
    const auto& dp = up->device_param();
    CHECK(dp.shard_ranges == "balanced" || dp.shard_ranges == "uniform") << "shard_ranges = balanced | uniform";
    CHECK(dp.shard_exchange == "overlap" || dp.shard_exchange == "sync") << "shard_exchange = overlap | sync";
    std::vector<uint64_t> splits(world_ > 1 ? world_ - 1 : 1, 0);
    const bool balanced = dp.shard_ranges == "balanced" && world_ > 1;
    if (balanced) DFH_CALL(dfh_shard_balanced_splits(comm_, sample.data(), sample.size(), splits.data()));
    DFH_CALL(dfh_shard_create(up->table(), comm_, balanced ? splits.data() : nullptr, &shard_));
    if (dp.shard_exchange == "overlap") DFH_CALL(dfh_shard_set_exchange(shard_, 1));
    if (reserve_keys_) DFH_CALL(dfh_shard_reserve(shard_, reserve_keys_, 2 * reserve_keys_));  // no re-allocation inside a step
    uint64_t lo = 0, hi = 0;
    DFH_CALL(dfh_shard_owned_range(shard_, balanced ? splits.data() : nullptr, &lo, &hi));
    LOG(INFO) << "sharded store: rank " << rank_ << " owns the reversed keys [" << lo << ", " << (hi ? std::to_string(hi) : "2^64")
              << ") (" << dp.shard_ranges << " ranges), exchange = " << dp.shard_exchange;
  }

  /**
   * \brief the literal Store::Push / Pull (include/difacto/store.h:53-73) on the sharded model: keys routed to their owners,
   * rows / acknowledgements back (dfh_shard_push_host / dfh_shard_pull_host).  COLLECTIVE: every rank makes the same
   * sequence of calls (a rank without a minibatch passes empty arrays) — what SGDLearner's literal worker loop does.
   * Synchronous: the arrays are consumed, and on_complete has run, on return.
   */
  int Push(const SArray<feaid_t>& fea_ids, int val_type, const SArray<real_t>& vals, const SArray<int>& lens,
           const std::function<void()>& on_complete) override {
    DFH_CALL(dfh_shard_push_host(CHECK_NOTNULL(shard_), fea_ids.data(), fea_ids.size(), val_type, vals.data(), vals.size(), lens.data(),
                                 lens.size()));
    if (on_complete) on_complete();
    return time_++;
  }
  int Pull(const SArray<feaid_t>& fea_ids, int val_type, SArray<real_t>* vals, SArray<int>* lens,
           const std::function<void()>& on_complete) override {
    CHECK_EQ(val_type, Store::kWeight);
    auto* up = CHECK_NOTNULL(dynamic_cast<DeviceSGDUpdater*>(CHECK_NOTNULL(updater_.get())));
    const size_t n = fea_ids.size();
    CHECK_NOTNULL(vals)->resize(n * (1 + up->param().V_dim));
    CHECK_NOTNULL(lens)->resize(n);
    size_t nvals = 0, nlens = 0;
    DFH_CALL(dfh_shard_pull_host(CHECK_NOTNULL(shard_), fea_ids.data(), n, vals->data(), &nvals, lens->data(), &nlens));
    vals->resize(nvals);
    lens->resize(nlens);
    if (on_complete) on_complete();
    return time_++;
  }
  void Wait(int time) override {}
  int Rank() override { return rank_; }
  int NumWorkers() override { return world_; }
  int NumServers() override { return world_; }

  dfh_shard* shard() { return shard_; }
  /*! \brief unique keys of the largest minibatch to expect (batch_size x ids per row): the exchange buffers are sized for it
   *  when the shard is created; 0: they grow when a step first meets a size */
  void set_reserve_keys(size_t n) { reserve_keys_ = n; }
  dfh_comm* comm() { return comm_; }

 private:
  int rank_ = 0, world_ = 1;
  int time_ = 0;
  dfh_comm* comm_ = nullptr;
  dfh_shard* shard_ = nullptr;
  size_t reserve_keys_ = 0;
  std::unique_ptr<FileExchange> files_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_SHARDED_STORE_H_
