/**
 * sgd_learner.cc — see sgd_learner.h.  Reference: src/sgd/sgd_learner.cc.
 */
#include "./sgd_learner.h"
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <thread>
#include <unistd.h>
#include "./hip_fm_loss.h"
#include "./host_localizer.h"
#include "./libsvm_reader.h"
#include "./sharded_store.h"
#include "data/row_block.h"

namespace difacto {

DMLC_REGISTER_PARAMETER(SGDLearnerParam);
DMLC_REGISTER_PARAMETER(SGDUpdaterParam);
DMLC_REGISTER_PARAMETER(DeviceParam);
DMLC_REGISTER_PARAMETER(FMLossParam);

SGDLearner::~SGDLearner() {
  for (auto& b : batch_)
    if (b) dfh_batch_destroy(b);
  delete loss_;
  delete store_;
}

// reference: SGDLearner::Init, sgd_learner.cc:229-246 — every stage consumes
// the keys it knows and hands the rest on; what is left is returned (and warned about by main)
KWArgs SGDLearner::Init(const KWArgs& kwargs) {
  auto remain = Learner::Init(kwargs);
  remain = param_.InitAllowUnknown(remain);
  CHECK(param_.data_format == "libsvm" || param_.data_format == "criteo" || param_.data_format == "criteo_test" ||
        param_.data_format == "adfea" || param_.data_format == "rec")
      << "data_format " << param_.data_format << " is not one of the reference's (libsvm, criteo, criteo_test, adfea, rec)";
  auto updater = new DeviceSGDUpdater();
  remain = updater->Init(remain);
  remain.push_back(std::make_pair("V_dim", std::to_string(updater->param().V_dim)));
  store_ = Store::Create();
  store_->SetUpdater(std::shared_ptr<Updater>(updater));
  remain = store_->Init(remain);
  loss_ = Loss::Create(param_.loss, blk_nthreads_);
  remain = loss_->Init(remain);
  if (auto* ss = dynamic_cast<ShardedDeviceStore*>(store_)) {
    // the key ranges of the shards: balanced on a sample of the keys this rank is going to see (the first minibatches
    // of its first data part), gathered over the ranks
    std::vector<feaid_t> sample;
    if (GetUpdater()->device_param().shard_ranges == "balanced" && store_->NumWorkers() > 1) {
      sgd::Job job;
      job.type = param_.task == "predict" ? sgd::Job::kPrediction : sgd::Job::kTraining;
      const int n = store_->NumWorkers() * param_.num_jobs_per_epoch;
      BatchReader reader(JobData(job), param_.data_format, store_->Rank(), n, param_.batch_size, 0, 1.0f);
      while (sample.size() < (1u << 18) && reader.Next()) {
        const auto& blk = reader.Value();
        for (size_t i = blk.offset[0]; i < blk.offset[blk.size]; ++i) sample.push_back(ReverseBytes(blk.index[i]));
      }
    }
    // exchange buffers for this job's minibatches up front (batch_size x feed_ids_per_row keys): nothing is re-allocated
    // inside a step unless the data has more ids per row than announced
    ss->set_reserve_keys(static_cast<size_t>(param_.batch_size) * static_cast<size_t>(GetUpdater()->device_param().feed_ids_per_row));
    ss->CreateShard(sample);
  }
  if (param_.model_in.size()) LoadModel();
  return remain;
}

// reference: SGDLearner::RunScheduler, sgd_learner.cc:31-68
void SGDLearner::RunScheduler() {
  if (param_.task == "predict") {
    RunPrediction();
    return;
  }
  CHECK(param_.task == "train") << "unknown task " << param_.task;
  real_t pre_loss = 0, pre_val_auc = 0;
  for (int k = 0; k < param_.max_num_epochs; ++k) {
    sgd::Progress train_prog;
    LOG(INFO) << "Start epoch " << k;
    RunEpoch(k, sgd::Job::kTraining, &train_prog);
    LOG(INFO) << " - Training: " << train_prog.TextString();
    sgd::Progress val_prog;
    if (param_.data_val.size()) {
      RunEpoch(k, sgd::Job::kValidation, &val_prog);
      LOG(INFO) << " - Validation: " << val_prog.TextString();
    }
    for (const auto& cb : epoch_end_callback_) cb(k, train_prog, val_prog);

    real_t eps = std::fabs(train_prog.loss - pre_loss) / pre_loss;
    if (eps < param_.stop_rel_objv) {
      LOG(INFO) << "Change of loss [" << eps << "] < stop_rel_objv [" << param_.stop_rel_objv << "]";
      break;
    }
    if (val_prog.auc > 0) {
      eps = (val_prog.auc - pre_val_auc) / val_prog.nrows;
      if (eps < param_.stop_val_auc) {
        LOG(INFO) << "Change of validation AUC [" << eps << "] < stop_val_auc [" << param_.stop_val_auc << "]";
        break;
      }
    }
    if (k + 1 >= param_.max_num_epochs) LOG(INFO) << "Reach maximal number of epochs";
    pre_loss = train_prog.loss;
    pre_val_auc = val_prog.auc;
  }
  if (param_.model_out.size()) SaveModel();
}

// reference: SGDLearner::RunEpoch, sgd_learner.cc:70-94.  The data is cut into NumWorkers() x
// num_jobs_per_epoch parts; in a sharded run every rank is its own scheduler and takes the parts
// j * NumWorkers() + Rank(), then the ranks' progress records are summed so that all of them see the
// same epoch result (and take the same stopping decision).
void SGDLearner::RunEpoch(int epoch, int job_type, sgd::Progress* prog) {
  tracker_->SetMonitor([prog](int node_id, const std::string& rets) { prog->Merge(rets); });
  const int nworkers = store_->NumWorkers();
  const int n = nworkers * param_.num_jobs_per_epoch;
  const bool sharded = dynamic_cast<ShardedDeviceStore*>(store_) != nullptr;
  std::vector<std::pair<int, std::string>> jobs;
  for (int i = 0; i < n; ++i) {
    if (sharded && i % nworkers != store_->Rank()) continue;
    sgd::Job job;
    job.type = job_type;
    job.epoch = epoch;
    job.num_parts = n;
    job.part_idx = i;
    jobs.emplace_back(0, std::string());
    job.SerializeToString(&jobs.back().second);
  }
  tracker_->Issue(jobs);
  while (tracker_->NumRemains()) std::this_thread::sleep_for(std::chrono::microseconds(200));
  if (sharded) MergeAcrossRanks(prog);
}

// task = predict (src/main.cc:61-62 is a TODO in the reference; sgd_param.h:24-28: "model_in ... should be specified if it
// is a prediction task").  One pass, no shuffling, no sampling, nothing pushed: every data part is a job of kind
// kPrediction whose worker loop is the validation loop (Localizer -> Pull -> FMLoss::Predict) plus the logits going to
// a file.  One process: the parts run in order and append to <pred_out>.  Sharded: rank r takes the parts i = r (mod
// ranks) and writes <pred_out>.part-<i>; the parts are consecutive byte ranges of the data, so concatenating the files
// in part order gives the predictions in file order.
void SGDLearner::RunPrediction() {
  CHECK(param_.model_in.size()) << "task=predict needs model_in (sgd_param.h:24-28)";
  CHECK(param_.pred_out.size()) << "task=predict needs pred_out=<file>";
  sgd::Progress prog;
  RunEpoch(0, sgd::Job::kPrediction, &prog);
  LOG(INFO) << "predicted " << prog.nrows << " examples of " << (param_.data_val.size() ? param_.data_val : param_.data_in)
            << " -> " << param_.pred_out << " (against their labels: " << prog.TextString() << ")";
}

const std::string& SGDLearner::JobData(const sgd::Job& job) const {
  if (job.type == sgd::Job::kTraining) return param_.data_in;
  if (job.type == sgd::Job::kPrediction) return param_.data_val.size() ? param_.data_val : param_.data_in;
  return param_.data_val;
}

void SGDLearner::WritePredictions(dfh_batch* b) {
  size_t nrows = 0;
  DFH_CALL(dfh_batch_shape(b, &nrows, nullptr, nullptr));
  pred_buf_.resize(nrows);
  if (nrows == 0) return;
  DFH_CALL(dfh_batch_get_pred(b, pred_buf_.data()));  // waits for the step
  for (size_t i = 0; i < nrows; ++i) {
    const float v = param_.pred_prob ? 1.0f / (1.0f + std::exp(-pred_buf_[i])) : pred_buf_[i];
    CHECK_GT(fprintf(CHECK_NOTNULL(pred_file_), "%.9g\n", v), 0) << "cannot write " << param_.pred_out;
  }
}

void SGDLearner::MergeAcrossRanks(sgd::Progress* prog) {
  auto* ss = dynamic_cast<ShardedDeviceStore*>(store_);
  if (!ss) return;
  double v[5] = {prog->loss, prog->penalty, prog->auc, prog->nnz_w, prog->nrows};
  DFH_CALL(dfh_comm_allreduce_sum(ss->comm(), v, 5));
  prog->loss = static_cast<real_t>(v[0]);
  prog->penalty = static_cast<real_t>(v[1]);
  prog->auc = static_cast<real_t>(v[2]);
  prog->nnz_w = static_cast<real_t>(v[3]);
  prog->nrows = static_cast<real_t>(v[4]);
}

// reference: SGDLearner::Process, sgd_learner.h:42-53
void SGDLearner::Process(const std::string& args, std::string* rets) {
  sgd::Progress prog;
  sgd::Job job;
  job.ParseFromString(args);
  if (job.type == sgd::Job::kTraining || job.type == sgd::Job::kValidation) {
    IterateData(job, &prog);
  } else if (job.type == sgd::Job::kPrediction) {
    const bool sharded = dynamic_cast<ShardedDeviceStore*>(store_) != nullptr;
    const std::string path = sharded ? param_.pred_out + ".part-" + std::to_string(job.part_idx) : param_.pred_out;
    pred_file_ = fopen(path.c_str(), (sharded || job.part_idx == 0) ? "w" : "a");
    CHECK(pred_file_) << "cannot open " << path;
    IterateData(job, &prog);
    CHECK_EQ(fclose(pred_file_), 0) << "cannot write " << path;
    pred_file_ = nullptr;
  }
  prog.SerializeToString(rets);
}

void SGDLearner::IterateData(const sgd::Job& job, sgd::Progress* prog) {
  if (GetUpdater()->device_param().device_path == "literal" && job.type != sgd::Job::kPrediction) {
    IterateDataLiteral(job, prog);  // sharded store too: its Push / Pull are collective, the loop keeps the ranks in step
  } else if (dynamic_cast<ShardedDeviceStore*>(store_)) {
    IterateDataSharded(job, prog);
  } else {
    IterateDataFused(job, prog);
  }
}

namespace {
// Device feed: the reader's shuffle buffers in HBM.  The reader thread uploads every buffer once, when it becomes current
// (BatchReader::Describe), into a device row buffer; the worker loop then sends 4 B per row of a minibatch and the rows are
// gathered on the device (dfh_batch_gather_rows) instead of being copied twice on the host (into the minibatch, into the
// pinned staging area).  Buffers are named by their serial: `live` holds every uploaded buffer that a minibatch may still
// name; the worker loop hands a buffer back (Release) once the gather of the last minibatch that names it has been queued —
// with down-sampling (neg_sampling < 1) one minibatch may span any number of buffers, so there is no fixed ring.  A buffer
// that is handed back is reused by a later upload, which waits ON THE DEVICE for the gathers queued on it
// (dfh_rowbuf_load_host); buffers are freed on the loop's thread, at the end of the job.
struct DeviceFeed {
  dfh_ctx* ctx = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, dfh_rowbuf*> live;        // serial -> uploaded buffer
  std::vector<dfh_rowbuf*> spare, all;         // handed back / every buffer ever created
  std::map<dfh_rowbuf*, std::pair<size_t, size_t>> cap;
  // A buffer's upload is queued by the thread that builds it, as soon as it exists — one buffer ahead of the one being cut
  // into minibatches (BatchReader's on_built) — and carried out by a few upload threads, each on the stream of the row
  // buffer it fills (one pageable copy per slice: ~1.4 ms per 31 MB buffer on one thread, more than the device needs for
  // the buffer's ten minibatches)
  struct Job {
    std::vector<size_t> offset;     // the buffer's own offsets (copied: the builder's container moves on)
    std::vector<BufSlice> slices;   // shared ownership of the parsed chunks
    uint64_t serial;
  };
  std::deque<Job> jobs;
  std::vector<std::thread> workers;
  bool stop = false;
  // rows / nnz: what a shuffle buffer is expected to hold.  Every upload thread creates one row buffer of that size right
  // away — the buffer's arrays, its stream (a hardware queue: milliseconds) and events — while the reader is still parsing
  // the first chunks, instead of when the first shuffle buffer is already waiting for its upload (DIFACTO_FEED_PRECREATE=0:
  // as before).  A buffer that turns out too small stays spare and a fitting one is created then, as always.
  void Start(size_t rows, size_t nnz) {
    const char* e = getenv("DIFACTO_UPLOAD_THREADS");
    const int n = std::max(1, std::min(e ? atoi(e) : 2, 8));
    const char* pc = getenv("DIFACTO_FEED_PRECREATE");
    const bool precreate = rows > 0 && !(pc && atoi(pc) == 0);
    for (int t = 0; t < n; ++t)
      workers.emplace_back([this, rows, nnz, precreate] {
        if (precreate) {
          dfh_rowbuf* rb = nullptr;
          if (dfh_rowbuf_create(ctx, rows, std::max<size_t>(nnz, 1), &rb) == DFH_OK) {
            std::lock_guard<std::mutex> lk(mu);
            all.push_back(rb);
            cap[rb] = {rows, std::max<size_t>(nnz, 1)};
            spare.push_back(rb);
          }
        }
        Work();
      });
  }
  void Upload(const dmlc::RowBlock<feaid_t>& blk, const std::vector<BufSlice>& slices, uint64_t serial) {   // builder's thread
    Job j;
    j.offset.assign(blk.offset, blk.offset + blk.size + 1);
    j.slices = slices;
    j.serial = serial;
    {
      std::lock_guard<std::mutex> lk(mu);
      jobs.push_back(std::move(j));
    }
    cv.notify_all();
  }
  void Work() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;
        j = std::move(jobs.front());
        jobs.pop_front();
      }
      DoUpload(j);
    }
  }
  void DoUpload(const Job& j) {
    const std::vector<BufSlice>& slices = j.slices;
    const uint64_t serial = j.serial;
    const size_t nrows = j.offset.size() - 1;
    const size_t nnz = j.offset[nrows] - j.offset[0];
    dfh_rowbuf* rb = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < spare.size() && !rb; ++i) {
        const auto& c = cap[spare[i]];
        if (c.first >= nrows && c.second >= nnz) {
          rb = spare[i];
          spare.erase(spare.begin() + i);
        }
      }
    }
    if (!rb) {   // none fits: a new one, in place of a spare that has proved too small (such buffers — ~58 MB and
                 // a stream each — used to stay until the feed was destroyed)
      dfh_rowbuf* small = nullptr;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!spare.empty()) {
          small = spare.back();
          spare.pop_back();
          all.erase(std::remove(all.begin(), all.end(), small), all.end());
          cap.erase(small);
        }
      }
      if (small) DFH_CALL(dfh_rowbuf_destroy(small));   // waits for the gathers queued out of it, nothing else
      const size_t rows = std::max<size_t>(nrows, 1), nz = std::max<size_t>(nnz + nnz / 4, 1);
      DFH_CALL(dfh_rowbuf_create(ctx, rows, nz, &rb));
      std::lock_guard<std::mutex> lk(mu);
      all.push_back(rb);
      cap[rb] = {rows, nz};
    }
    std::vector<const uint64_t*> idx(slices.size());
    std::vector<const float*> val(slices.size());
    std::vector<size_t> cnt(slices.size());
    for (size_t g = 0; g < slices.size(); ++g) {
      idx[g] = slices[g].index();
      val[g] = slices[g].value();
      cnt[g] = slices[g].nnz();
    }
    DFH_CALL(dfh_rowbuf_load_host_slices(rb, nrows, j.offset.data(), static_cast<int>(slices.size()), idx.data(), val.data(), cnt.data()));
    {
      std::lock_guard<std::mutex> lk(mu);
      CHECK(live.emplace(serial, rb).second) << "shuffle buffer " << serial << " uploaded twice";
    }
    cv.notify_all();
  }
  dfh_rowbuf* Of(uint64_t serial) {   // on the worker loop's thread: waits for an upload still under way
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return live.count(serial) != 0; });
    auto it = live.find(serial);
    CHECK(it != live.end()) << "minibatch names shuffle buffer " << serial << ", which is not (or no longer) on the device";
    return it->second;
  }
  // the gathers of every minibatch that names a buffer below `serial` have been queued
  void Release(uint64_t serial) {
    std::lock_guard<std::mutex> lk(mu);
    while (!live.empty() && live.begin()->first < serial) {
      spare.push_back(live.begin()->second);
      live.erase(live.begin());
    }
  }
  ~DeviceFeed() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& w : workers)
      if (w.joinable()) w.join();
    for (auto* rb : all) dfh_rowbuf_destroy(rb);
  }
};
}  // namespace

// ---- the fused worker loop: sgd_learner.cc:129-227 on the device
void SGDLearner::IterateDataFused(const sgd::Job& job, sgd::Progress* progress) {
  const bool train = job.type == sgd::Job::kTraining;
  const bool predict = job.type == sgd::Job::kPrediction;
  const bool push_cnt = train && job.epoch == 0;  // sgd_learner.cc:201-202
  dfh_ctx* ctx = DeviceContext::Get();
  dfh_table* table = GetUpdater()->table();
  // preparation streams (DIFACTO_PREP_STREAMS=1..4).  With the device feed the preparation of a minibatch is the row gather
  // + Localizer + probe — a chain as long as the step itself — and two streams let consecutive preparations overlap:
  // same box, by this loop's clock, .rec 53.6 -> 59.4 and criteo text 48.0 -> 50.6 M rows/s (tools/gpu_r04x.sh).  bench.py,
  // whose inputs are already in HBM (no gather), loses 5 % with two: it keeps one
  const char* ps = getenv("DIFACTO_PREP_STREAMS");
  const bool feed_wanted = job.type == sgd::Job::kTraining && param_.shuffle > 0 && getenv("DIFACTO_HOST_FEED") == nullptr;
  // DIFACTO_SINGLE_QUEUE=1: the single-queue step (csrc/dfh_riders.hip: the Localizer's stages riding in the step's own three
  // launches, no preparation stream, no events; minibatches then prepared TWO ahead so that every stage finds a launch to ride
  // in).  Off by default: measured through this loop it loses at every size (profiles/r06m_*: batch 100, V_dim 8 from libsvm 2.22
  // against 1.59 M rows/s; batch 2 000 Criteo rows 28.7 against 21.9 M) — with the device feed the row gather is one more launch
  // on the ONE queue and a launch with riders is as long as its longest rider, while two queues hide the whole preparation
  // behind the step.  (Inputs resident in HBM, bench.py's C2 preset: the same rate with half the host time per step.)
  const char* sqe = getenv("DIFACTO_SINGLE_QUEUE");
  const bool single_queue = sqe != nullptr && atoi(sqe) != 0;
  DFH_CALL(dfh_ctx_set_option(ctx, "single_queue", single_queue ? 1 : 0));
  // The preparation streams' priority (DIFACTO_PREP_PRIORITY=-1|0|1; decided by the first job of a process: the streams are
  // created once).  With the device feed the preparation chain (row gather + Localizer + probe) is as long as the step, and at
  // the lowest priority — bench.py's setting, whose chain is shorter — it is the chain that sets the pace: default priority,
  // same box, by this loop's clock, .rec 62.9 / 63.2 -> 66.9 / 64.9 M rows/s, criteo text unchanged (58.5 M: the parsers'
  // pace), three streams no better (66.6), highest priority worse (60.8) — profiles/r06e2e_prep_priority.txt.
  static bool prio_set = false;
  if (!prio_set) {
    const char* pp = getenv("DIFACTO_PREP_PRIORITY");
    if (pp || feed_wanted) DFH_CALL(dfh_ctx_set_option(ctx, "prep_priority", pp ? atoi(pp) : 0));
    prio_set = true;
  }
  DFH_CALL(dfh_ctx_set_pipeline(ctx, ps ? std::max(1, std::min(atoi(ps), 4)) : (feed_wanted ? 2 : 1)));  // (ignored on the single queue)
  const int ahead = single_queue ? 2 : 1;
  // minibatches are cut (permutation + row selection) two ahead on the reader's own thread, the reference's reader /
  // executor overlap (sgd_learner.cc:196-224).  Training with a shuffle buffer: the buffers go to HBM and the rows are
  // gathered there (device feed; DIFACTO_HOST_FEED=1 keeps the host-side gather)
  DeviceFeed feed;   // outlives the reader, whose thread uploads into it
  feed.ctx = ctx;
  const bool device_feed = train && param_.shuffle > 0 && getenv("DIFACTO_HOST_FEED") == nullptr;
  BatchReader::SliceFn upload;
  // (feed_ids_per_row, default 48: the criteo rows of the reference's example have 39)
  const size_t ids_per_row = static_cast<size_t>(GetUpdater()->device_param().feed_ids_per_row);
  if (device_feed) feed.Start(static_cast<size_t>(param_.batch_size) * param_.shuffle, static_cast<size_t>(param_.batch_size) * param_.shuffle * ids_per_row);
  if (device_feed)
    upload = [&feed](const dmlc::RowBlock<feaid_t>& blk, const std::vector<BufSlice>& slices, uint64_t serial) {
      feed.Upload(blk, slices, serial);
    };
  // device feed: buffers as slices of the parsed chunks (no host assembly), uploaded by the thread that builds them
  BatchReader* batch_reader = new BatchReader(JobData(job), param_.data_format, job.part_idx, job.num_parts, param_.batch_size,
                                              train ? param_.batch_size * param_.shuffle : 0, train ? param_.neg_sampling : 1.0f,
                                              device_feed, upload);
  if (device_feed) batch_reader->DescribeSlices(nullptr);
  // described minibatches are ~120 KB each: a deeper queue lets the loop ride out the reader's pause at a buffer boundary
  PrefetchSource reader(batch_reader, device_feed ? kFusedBatches : 2);
  // objects in rotation: a dozen only where a dozen steps can be queued (the device feed's bursts); validation, prediction
  // and host-feed jobs have a prefetch depth of 2 and rotate three (six times the HBM and creation time for nothing)
  const int nrot = device_feed ? kFusedBatches : 3;
  auto all_there = [&] {
    for (int q = 0; q < nrot; ++q)
      if (!batch_[q]) return false;
    return true;
  };
  auto ensure = [&](size_t rows, size_t nnz) {
    if (all_there() && rows <= batch_rows_ && nnz <= batch_nnz_) return;
    for (auto& b : batch_) {
      if (b) {
        dfh_progress p;
        DFH_CALL(dfh_batch_progress(b, &p, 1));  // never pending here: drained at the end of every job
        dfh_batch_destroy(b);
        b = nullptr;
      }
    }
    batch_rows_ = std::max(rows, batch_rows_);
    batch_nnz_ = std::max(nnz * 2, batch_nnz_);
    // one device allocation for all of them (dfh_batch_create_many): twelve objects used to be 12 x 61 hipMallocs, ~35 ms of
    // a job's start-up
    DFH_CALL(dfh_batch_create_many(ctx, nrot, batch_rows_, std::max<size_t>(batch_nnz_, 1), batch_));
    batch_arena_ = true;
    for (int q = 0; q < nrot; ++q) DFH_CALL(dfh_batch_set_option(batch_[q], "compute_auc", 1));  // sgd_learner.cc:153-155
  };
  const bool split_prep = getenv("DIFACTO_SPLIT_PREP") != nullptr;
  std::vector<dfh_rowbuf*> bufs;
  std::vector<const uint32_t*> rows;
  std::vector<size_t> cnts;
  const bool prof = getenv("DIFACTO_PROFILE") != nullptr;
  double t_feed = 0;   // DIFACTO_PROFILE: inside "stage + localize + lookup", waiting for a buffer's upload (DeviceFeed::Of)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // prepare batch t+1 (H2D copy, Localizer, key lookup) while batch t trains
  auto prepare = [&](int slot) {
    const auto& blk = reader.Value();
    // a growing batch needs new buffers while the other slot may be in flight: drain first (the
    // main loop has already trained the pending batch)
    if (!all_there() || blk.size > batch_rows_ || blk.offset[blk.size] - blk.offset[0] > batch_nnz_) {
      DFH_CALL(dfh_ctx_sync(ctx));
      sgd::Progress keep;
      for (auto& b : batch_) {
        if (!b) continue;
        dfh_progress p;
        DFH_CALL(dfh_batch_progress(b, &p, 1));
        keep.loss += p.loss; keep.penalty += p.penalty; keep.auc += p.auc; keep.nrows += p.nrows;
      }
      progress->Merge(keep);
      ensure(blk.size, blk.offset[blk.size] - blk.offset[0]);
    }
    dfh_batch* b = batch_[slot];
    if (getenv("DIFACTO_TRACE") && (blk.index || blk.offset[blk.size] == blk.offset[0])) {
      uint64_t cs = 0;
      double ls = 0;
      for (size_t i = blk.offset[0]; i < blk.offset[blk.size]; ++i) cs += blk.index[i] * (i - blk.offset[0] + 1);
      for (size_t i = 0; i < blk.size; ++i) ls += blk.label[i] * (i + 1);
      LOG(INFO) << "batch rows " << blk.size << " nnz " << blk.offset[blk.size] - blk.offset[0] << " off0 " << blk.offset[0]
                << " idxsum " << cs << " labsum " << ls << " value " << (blk.value != nullptr);
    }
    const std::vector<RowSeg>& segs = reader.Aux();
    if (!segs.empty()) {   // a described minibatch: its rows are gathered out of the device-resident buffers
      bufs.resize(segs.size());
      rows.resize(segs.size());
      cnts.resize(segs.size());
      const double tf = prof ? now() : 0;
      for (size_t g = 0; g < segs.size(); ++g) {
        bufs[g] = CHECK_NOTNULL(feed.Of(segs[g].buf));
        rows[g] = segs[g].rows.data();
        cnts[g] = segs[g].rows.size();
      }
      if (prof) t_feed += now() - tf;
      // gather + Localizer (Localizer lc(-1, ...), sgd_learner.cc:203) + key lookup as one preparation phase
      // (DIFACTO_SPLIT_PREP=1: the three calls of round 3, for A/B)
      if (split_prep) {
        DFH_CALL(dfh_batch_gather_rows(b, blk.size, blk.offset, blk.label, static_cast<int>(segs.size()), bufs.data(), rows.data(),
                                       cnts.data()));
      } else {
        DFH_CALL(dfh_batch_prepare_rows(table, b, blk.size, blk.offset, blk.label, static_cast<int>(segs.size()), bufs.data(),
                                        rows.data(), cnts.data(), ~0ULL));
      }
      // minibatches take their rows from the buffers in order: the buffers before this one's last are exhausted
      uint64_t last = 0;
      for (const auto& g : segs) last = std::max<uint64_t>(last, g.buf);
      feed.Release(last);
      if (!split_prep) return;
    } else {
      DFH_CALL(dfh_batch_load_host(b, blk.size, blk.offset, blk.index, blk.value, blk.label));
    }
    DFH_CALL(dfh_localize(b, ~0ULL));  // Localizer lc(-1, ...), sgd_learner.cc:203
    DFH_CALL(dfh_batch_lookup(table, b));  // (dfh_localize_lookup does both in one pass; measured 0.4 % slower per step)
  };
  auto needs_growth = [&](const dmlc::RowBlock<feaid_t>& blk) {
    return !all_there() || blk.size > batch_rows_ || blk.offset[blk.size] - blk.offset[0] > batch_nnz_;
  };
  // DIFACTO_PROFILE=1: where the host thread of this loop spends its time (reader wait + batch assembly,
  // staging + Localizer / lookup queueing, step queueing), printed at the end of the job
  double t_read = 0, t_prep = 0, t_step = 0;
  double t0 = prof ? now() : 0;
  // the batch objects while the reader parses its first chunks (sizes: this job's batch_size at 48 ids per row, or what an
  // earlier job left; a bigger minibatch re-creates them, as before)
  if (device_feed && !all_there() && !(getenv("DIFACTO_FEED_PRECREATE") && atoi(getenv("DIFACTO_FEED_PRECREATE")) == 0))
    ensure(param_.batch_size, static_cast<size_t>(param_.batch_size) * ids_per_row);
  const double t_created = prof ? now() : 0;
  // minibatch i sits in object i mod nrot.  `ahead` minibatches are prepared in front of the one that steps (1: prepare(i + 1),
  // step(i) — the two minibatches the reference keeps in flight, sgd_learner.cc:219-223; 2 on the single queue)
  static_assert(kFusedBatches >= 3, "three objects in rotation at least");
  int prepared = 0, i = 0;   // minibatches prepared / stepped so far
  bool more = true;
  auto step_one = [&] {
    const int cur = i % nrot;
    DFH_CALL(dfh_sgd_step(table, batch_[cur], train ? 1 : 0, push_cnt ? 1 : 0));
    if (predict) WritePredictions(batch_[cur]);
    ++i;
  };
  double t_first = 0;
  auto prepare_next = [&] {   // -> false: the reader is exhausted
    if (!more) return false;
    more = reader.Next();
    if (prepared == 0 && prof) t_first = now();
    if (prof) { const double t1 = now(); t_read += t1 - t0; t0 = t1; }
    if (!more) return false;
    if (needs_growth(reader.Value())) {
      // growing re-creates ALL batch objects: the prepared, not yet trained minibatches go first
      while (i < prepared) step_one();
      if (prof) { const double t1 = now(); t_step += t1 - t0; t0 = t1; }
    }
    prepare(prepared % nrot);
    ++prepared;
    if (prof) { const double t1 = now(); t_prep += t1 - t0; t0 = t1; }
    return true;
  };
  for (int a = 0; a < ahead; ++a) prepare_next();
  if (prof)
    LOG(INFO) << "start-up: batch objects " << t_created - (t_first - t_read) << " s, first minibatch from the reader after "
              << t_first - t_created << " s more, the first preparations queued in " << now() - t_first << " s";
  while (i < prepared) {
    prepare_next();
    step_one();
    if (prof) { const double t1 = now(); t_step += t1 - t0; t0 = t1; }
  }
  if (prof)
    LOG(INFO) << "host loop over " << i << " minibatches: reader " << t_read << " s, stage + localize + lookup " << t_prep
              << " s (" << t_feed << " s of it waiting for a buffer's upload), step " << t_step << " s";
  for (auto& b : batch_) {
    if (!b) continue;
    dfh_progress p;
    DFH_CALL(dfh_batch_progress(b, &p, 1));
    sgd::Progress q;
    q.loss = p.loss; q.penalty = p.penalty; q.auc = p.auc; q.nrows = p.nrows;
    progress->Merge(q);
  }
  uint64_t nkeys;
  DFH_CALL(dfh_table_size(table, &nkeys));  // surfaces a full table as an error
}

// ---- the sharded worker loop: sgd_learner.cc:129-227 with Store::Pull / Push turned into the exchange of
// dfh_shard_step.  The ranks step together; a rank whose part of the data is exhausted keeps serving
// its shard (b = NULL) until nobody has a minibatch left.  Four batch objects in rotation: while minibatch t steps,
// minibatch t+1 — copied in and localized one step EARLIER — has its per-owner key counts (and, overlapped, its keys
// and rows) exchanged inside the running step (dfh_shard_prefetch_counts), and minibatch t+2 is copied in and localized
// on the preparation stream: the keys of t+1 are ready when step t starts (round 5: localized only one ahead, the
// Localizer of t+1 ran beside step t's forward and the exchange waited for it).  The two minibatches the
// reference's batch tracker keeps in flight (sgd_learner.cc:219-223), without its staleness in sync mode.
void SGDLearner::IterateDataSharded(const sgd::Job& job, sgd::Progress* progress) {
  const bool train = job.type == sgd::Job::kTraining;
  const bool predict = job.type == sgd::Job::kPrediction;
  const bool push_cnt = train && job.epoch == 0;  // sgd_learner.cc:201-202
  auto* ss = CHECK_NOTNULL(dynamic_cast<ShardedDeviceStore*>(store_));
  // minibatches are cut (permutation + row gather) two ahead on the reader's own thread, the reference's reader /
  // executor overlap (sgd_learner.cc:196-224)
  PrefetchSource reader(new BatchReader(JobData(job), param_.data_format, job.part_idx, job.num_parts, param_.batch_size,
                                        train ? param_.batch_size * param_.shuffle : 0, train ? param_.neg_sampling : 1.0f), 2);
  dfh_ctx* ctx = DeviceContext::Get();
  DFH_CALL(dfh_ctx_set_option(ctx, "single_queue", 0));  // (a fused job of small minibatches may have left it on)
  DFH_CALL(dfh_ctx_set_pipeline(ctx, 1));
  auto drain = [&](dfh_batch* b) {
    dfh_progress p;
    DFH_CALL(dfh_batch_progress(b, &p, 1));
    sgd::Progress q;
    q.loss = p.loss; q.penalty = p.penalty; q.auc = p.auc; q.nrows = p.nrows;
    progress->Merge(q);
  };
  constexpr int kSlots = 4;
  static_assert(kSlots <= kFusedBatches, "the sharded loop's objects are the first of the fused loop's");
  // objects a fused job left behind share ONE allocation, which lives as long as any of them: all go (this loop's four
  // would otherwise pin twelve objects' worth of HBM), this loop creates its own
  for (int q = batch_arena_ ? 0 : kSlots; q < kFusedBatches; ++q) {  // the fused loop's further objects: this loop rotates four
    if (!batch_[q]) continue;
    drain(batch_[q]);
    dfh_batch_destroy(batch_[q]);
    batch_[q] = nullptr;
  }
  batch_arena_ = false;
  size_t cap_rows[kSlots] = {0}, cap_nnz[kSlots] = {0};
  for (int q = 0; q < kSlots; ++q) {  // objects left by an earlier job keep their size
    if (batch_[q]) {
      cap_rows[q] = batch_rows_;
      cap_nnz[q] = batch_nnz_;
    }
  }
  // the next minibatch of this rank into slot `q` (whose previous occupant has been stepped), or NULL
  auto prepare = [&](int q) -> dfh_batch* {
    if (!reader.Next()) return nullptr;
    const auto& blk = reader.Value();
    const size_t nnz = blk.offset[blk.size] - blk.offset[0];
    if (!batch_[q] || blk.size > cap_rows[q] || nnz > cap_nnz[q]) {
      // only this slot grows: the other one may hold a prepared minibatch that has not stepped yet
      if (batch_[q]) {
        drain(batch_[q]);              // synchronises: its last step is through
        dfh_batch_destroy(batch_[q]);
      }
      cap_rows[q] = std::max<size_t>(blk.size, std::max(cap_rows[q], batch_rows_));
      cap_nnz[q] = std::max(nnz * 2, std::max(cap_nnz[q], batch_nnz_));
      DFH_CALL(dfh_batch_create(ctx, cap_rows[q], std::max<size_t>(cap_nnz[q], 1), &batch_[q]));
      DFH_CALL(dfh_batch_set_option(batch_[q], "compute_auc", 1));  // sgd_learner.cc:153-155
    }
    dfh_batch* b = batch_[q];
    DFH_CALL(dfh_batch_load_host(b, blk.size, blk.offset, blk.index, blk.value, blk.label));
    DFH_CALL(dfh_localize(b, ~0ULL));  // Localizer lc(-1, ...), sgd_learner.cc:203
    return b;
  };
  dfh_batch* cur = prepare(0);
  dfh_batch* nxt = prepare(1);   // localized a step before it is announced
  int active = 1;
  for (int i = 0; active; ++i) {
    dfh_batch* ahead = prepare((i + 2) % kSlots);   // (its slot's previous minibatch, i - 2, has been stepped)
    DFH_CALL(dfh_shard_prefetch_counts(ss->shard(), nxt));
    if (getenv("DIFACTO_TRACE")) LOG(INFO) << "shard step: batch " << (cur ? "yes" : "none");
    DFH_CALL(dfh_shard_step(ss->shard(), cur, train ? 1 : 0, push_cnt ? 1 : 0, &active));
    if (predict && cur) WritePredictions(cur);
    if (getenv("DIFACTO_TRACE")) LOG(INFO) << "shard step done: active " << active;
    cur = nxt;
    nxt = ahead;
  }
  size_t max_rows = 0, max_nnz = 0;
  for (int q = 0; q < kSlots; ++q) {
    max_rows = std::max(max_rows, cap_rows[q]);
    max_nnz = std::max(max_nnz, cap_nnz[q]);
  }
  for (int q = 0; q < kSlots; ++q) {
    if (!batch_[q]) continue;
    drain(batch_[q]);
    // the fused loop sizes its objects alike: keep that invariant for whichever job runs next
    if (cap_rows[q] != max_rows || cap_nnz[q] != max_nnz) {
      dfh_batch_destroy(batch_[q]);
      batch_[q] = nullptr;
    }
  }
  batch_rows_ = max_rows;
  batch_nnz_ = max_nnz;
  for (int q = 0, w = 0; q < kSlots; ++q)   // the survivors to the front
    if (batch_[q]) {
      if (q != w) std::swap(batch_[q], batch_[w]);
      ++w;
    }
  uint64_t nkeys;
  DFH_CALL(dfh_table_size(GetUpdater()->table(), &nkeys));  // surfaces a full shard as an error
}

// reference: SGDLearner::GetPos, sgd_learner.cc:113-127
void SGDLearner::GetPos(const SArray<int>& len, SArray<int>* w_pos, SArray<int>* V_pos) {
  const size_t n = len.size();
  w_pos->resize(n);
  V_pos->resize(n);
  int p = 0;
  for (size_t i = 0; i < n; ++i) {
    const int l = len[i];
    (*w_pos)[i] = l == 0 ? -1 : p;
    (*V_pos)[i] = l > 1 ? p + 1 : -1;
    p += l;
  }
}

// reference: SGDLearner::EvaluatePenalty, sgd_learner.cc:249-273
real_t SGDLearner::EvaluatePenalty(const SArray<real_t>& weights, const SArray<int>& w_pos, const SArray<int>& V_pos) {
  double objv = 0;
  const auto& param = GetUpdater()->param();
  if (w_pos.size()) {
    for (int p : w_pos) {
      if (p == -1) continue;
      const double w = weights[p];
      objv += param.l1 * std::fabs(w) + .5 * param.l2 * w * w;
    }
    for (int p : V_pos) {
      if (p == -1) continue;
      for (int i = 0; i < param.V_dim; ++i) {
        const double V = weights[p + i];
        objv += .5 * param.V_l2 * V * V;
      }
    }
  } else {
    for (auto wf : weights) {
      const double w = wf;
      objv += param.l1 * std::fabs(w) + .5 * param.l2 * w * w;
    }
  }
  return static_cast<real_t>(objv);
}

// ---- the literal worker loop: the reference's sequence of interface calls
void SGDLearner::IterateDataLiteral(const sgd::Job& job, sgd::Progress* progress) {
  const bool train = job.type == sgd::Job::kTraining;
  const bool push_cnt = train && job.epoch == 0;
  PrefetchSource reader(new BatchReader(train ? param_.data_in : param_.data_val, param_.data_format, job.part_idx, job.num_parts,
                                        param_.batch_size, train ? param_.batch_size * param_.shuffle : 0,
                                        train ? param_.neg_sampling : 1.0f), 2);
  // the sharded store's Push / Pull are collective: every rank makes the same calls until no rank has a minibatch
  // left; a rank whose part of the data is exhausted goes on with empty arrays
  auto* ss = dynamic_cast<ShardedDeviceStore*>(store_);
  bool more = true;
  for (;;) {
    const bool have = more && reader.Next();
    more = have;
    if (ss) {
      double any = have ? 1.0 : 0.0;
      DFH_CALL(dfh_comm_allreduce_sum(ss->comm(), &any, 1));
      if (any == 0.0) break;
    } else if (!have) {
      break;
    }
    dmlc::data::RowBlockContainer<unsigned> data;
    auto feaids = std::make_shared<std::vector<feaid_t>>();
    auto feacnt = std::make_shared<std::vector<real_t>>();
    if (have) {
      Localizer lc(-1, blk_nthreads_);
      lc.Compact(reader.Value(), &data, feaids.get(), push_cnt ? feacnt.get() : nullptr);
    }
    SArray<feaid_t> keys(feaids);
    if (push_cnt) store_->Wait(store_->Push(keys, Store::kFeaCount, SArray<real_t>(feacnt), {}));
    SArray<real_t> values;
    SArray<int> lengths;
    store_->Wait(store_->Pull(keys, Store::kWeight, &values, &lengths));
    SArray<real_t> grads;
    if (have) {
      auto blk = data.GetBlock();
      progress->nrows += blk.size;
      SArray<real_t> pred(blk.size);
      SArray<int> w_pos, V_pos;
      GetPos(lengths, &w_pos, &V_pos);
      std::vector<SArray<char>> inputs = {SArray<char>(values), SArray<char>(w_pos), SArray<char>(V_pos)};
      loss_->Predict(blk, inputs, &pred);
      progress->loss += loss_->Evaluate(blk.label, pred);
      progress->penalty += EvaluatePenalty(values, w_pos, V_pos);
      float auc_n = 0;
      DFH_CALL(dfh_auc_times_n(DeviceContext::Get(), blk.label, pred.data(), pred.size(), &auc_n));
      progress->auc += auc_n;
      if (train) {
        grads.resize(values.size());
        inputs.push_back(SArray<char>(pred));
        loss_->CalcGrad(blk, inputs, &grads);
      }
    }
    if (train) store_->Wait(store_->Push(keys, Store::kGradient, grads, lengths));
  }
}

// Updater::Save / Load.  A sharded run writes one part per rank (<model_out>.part-<rank>) plus a manifest
// (<model_out>.parts: the number of parts, written by rank 0, which also removes the parts a run with more ranks left
// behind under the same name) and reads exactly the parts the manifest names, keeping the keys of its own range: a
// model can be re-loaded under any number of ranks, and stale parts of an earlier save are never imported.
void SGDLearner::SaveModel() {
  auto* ss = dynamic_cast<ShardedDeviceStore*>(store_);
  const std::string path = ss ? param_.model_out + ".part-" + std::to_string(store_->Rank()) : param_.model_out;
  // written under a temporary name and renamed when complete: a reader never sees half a file, and a rank that dies
  // mid-save leaves the previous part in place
  const std::string tmp = path + ".tmp";
  {
    std::unique_ptr<dmlc::Stream> fo(dmlc::Stream::Create(tmp.c_str(), "w"));
    GetUpdater()->Save(true, fo.get());
  }
  if (ss) {
    // the manifest is the commit point of a sharded save: it may only name parts that are complete.  Every rank has closed
    // its part when this all-reduce returns (a rank that failed never arrives: no new manifest); the parts are then
    // renamed into place and rank 0 writes the manifest after a second round.
    double ok[1] = {1.0};
    DFH_CALL(dfh_comm_allreduce_sum(ss->comm(), ok, 1));
    CHECK_EQ(static_cast<int>(ok[0]), store_->NumWorkers()) << "a rank did not finish its model part";
  }
  CHECK_EQ(rename(tmp.c_str(), path.c_str()), 0) << "cannot move " << tmp << " to " << path;
  LOG(INFO) << "model saved to " << path;
  if (ss) {
    double done[1] = {1.0};
    DFH_CALL(dfh_comm_allreduce_sum(ss->comm(), done, 1));   // every part is in place
  }
  if (ss && store_->Rank() == 0) {
    const int world = store_->NumWorkers();
    for (int n = world;; ++n) {  // parts of an earlier save with more ranks
      const std::string stale = param_.model_out + ".part-" + std::to_string(n);
      if (access(stale.c_str(), F_OK) != 0) break;
      CHECK_EQ(unlink(stale.c_str()), 0) << "cannot remove the stale model part " << stale;
      LOG(INFO) << "removed the stale model part " << stale;
    }
    const std::string mf = param_.model_out + ".parts";
    FILE* f = fopen((mf + ".tmp").c_str(), "w");
    CHECK(f) << "cannot write " << mf;
    fprintf(f, "%d\n", world);
    CHECK_EQ(fclose(f), 0);
    CHECK_EQ(rename((mf + ".tmp").c_str(), mf.c_str()), 0);
  }
}

void SGDLearner::LoadModel() {
  auto* ss = dynamic_cast<ShardedDeviceStore*>(store_);
  if (!ss) {
    std::unique_ptr<dmlc::Stream> fi(dmlc::Stream::Create(param_.model_in.c_str(), "r"));
    bool has_aux = false;
    GetUpdater()->Load(fi.get(), &has_aux);
    LOG(INFO) << "model loaded from " << param_.model_in << (has_aux ? " (with optimiser state)" : "");
    return;
  }
  uint64_t lo = 0, hi = 0, total = 0;
  DFH_CALL(dfh_shard_owned_range(ss->shard(), nullptr, &lo, &hi));
  // how many parts: the manifest of the save; a model saved before manifests existed: every part up to the first gap
  int want = -1;
  const std::string mf = param_.model_in + ".parts";
  if (FILE* f = fopen(mf.c_str(), "r")) {
    CHECK(fscanf(f, "%d", &want) == 1 && want >= 1 && want <= 4096) << "bad model manifest " << mf;
    fclose(f);
  } else {
    LOG(WARNING) << "no manifest " << mf << ": reading " << param_.model_in << ".part-<n> up to the first gap";
  }
  int parts = 0;
  for (;; ++parts) {
    if (want >= 0 && parts == want) break;
    const std::string path = param_.model_in + ".part-" + std::to_string(parts);
    if (access(path.c_str(), R_OK) != 0) {
      CHECK(want < 0) << "model part " << path << " is missing (the manifest names " << want << " parts)";
      break;
    }
    uint64_t n = 0;
    int aux = 0;
    DFH_CALL(dfh_table_load(GetUpdater()->table(), path.c_str(), lo, hi, &aux, &n));
    total += n;
  }
  CHECK_GT(parts, 0) << "no model part files " << param_.model_in << ".part-<n>";
  LOG(INFO) << "rank " << store_->Rank() << ": " << total << " entries of its key range loaded from " << parts << " part files";
}

}  // namespace difacto
