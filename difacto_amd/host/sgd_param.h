/**
 * sgd_param.h — configuration of the SGD learner and its updater: the same keys,
 * types, ranges and defaults as the reference's src/sgd/sgd_param.h, so that
 * its .conf files work unchanged; plus the keys of the device build.
 */
#ifndef DIFACTO_HOST_SGD_PARAM_H_
#define DIFACTO_HOST_SGD_PARAM_H_
#include <string>
#include "difacto/base.h"
#include "dmlc/parameter.h"

namespace difacto {

/*! \brief reference: SGDLearnerParam, sgd_param.h:12-64 */
struct SGDLearnerParam : public dmlc::Parameter<SGDLearnerParam> {
  std::string data_in;       // training data: a file (required)
  std::string data_val;      // optional validation data
  std::string data_format;   // "libsvm"
  std::string model_out;     // where to save the model after training
  std::string model_in;      // model to start from
  std::string loss;          // "fm" (default) or "logit"
  int max_num_epochs;
  int batch_size;            // required
  int shuffle;               // shuffle buffer = batch_size * shuffle rows
  float neg_sampling;        // keep probability of a negative example
  int num_jobs_per_epoch;    // file parts per worker and epoch
  real_t stop_rel_objv;      // stop if |objv - prev| / prev < this
  real_t stop_val_auc;       // stop if validation AUC gain < this
  // task = predict (the reference leaves it a TODO, src/main.cc:61-62; its sgd_param.h:24-28 names model_in as
  // the model of "a prediction task"): keys of this build, defaults leave training untouched
  std::string task;          // "train" (default) | "predict": forward pass of model_in over data_val (else data_in)
  std::string pred_out;      // predict: one line per example, in file order (<pred_out>.part-<i> per data part of a sharded run)
  int pred_prob;             // predict: 0 writes the logit FMLoss::Predict returns (default), 1 writes 1 / (1 + exp(-logit))
  DMLC_DECLARE_PARAMETER(SGDLearnerParam) {
    DMLC_DECLARE_FIELD(task).set_default("train");
    DMLC_DECLARE_FIELD(pred_out).set_default("");
    DMLC_DECLARE_FIELD(pred_prob).set_default(0);
    DMLC_DECLARE_FIELD(data_format).set_default("libsvm");
    DMLC_DECLARE_FIELD(data_in);
    DMLC_DECLARE_FIELD(data_val).set_default("");
    DMLC_DECLARE_FIELD(model_out).set_default("");
    DMLC_DECLARE_FIELD(model_in).set_default("");
    DMLC_DECLARE_FIELD(loss).set_default("fm");
    DMLC_DECLARE_FIELD(max_num_epochs).set_default(20);
    DMLC_DECLARE_FIELD(num_jobs_per_epoch).set_default(10);
    DMLC_DECLARE_FIELD(batch_size);
    DMLC_DECLARE_FIELD(shuffle).set_default(10);
    DMLC_DECLARE_FIELD(neg_sampling).set_default(1);
    DMLC_DECLARE_FIELD(stop_rel_objv).set_default(1e-5);
    DMLC_DECLARE_FIELD(stop_val_auc).set_default(1e-5);
  }
};

/*! \brief reference: SGDUpdaterParam, sgd_param.h:66-107 */
struct SGDUpdaterParam : public dmlc::Parameter<SGDUpdaterParam> {
  float l1, l2, V_l2;
  float lr, lr_beta, V_lr, V_lr_beta;
  float V_init_scale;
  int V_dim;
  int V_threshold;
  unsigned int seed;
  DMLC_DECLARE_PARAMETER(SGDUpdaterParam) {
    DMLC_DECLARE_FIELD(l1).set_range(0, 1e10).set_default(1);
    DMLC_DECLARE_FIELD(l2).set_range(0, 1e10).set_default(0);
    DMLC_DECLARE_FIELD(V_l2).set_range(0, 1e10).set_default(.01);
    DMLC_DECLARE_FIELD(lr).set_range(0, 10).set_default(.01);
    DMLC_DECLARE_FIELD(lr_beta).set_range(0, 1e10).set_default(1);
    DMLC_DECLARE_FIELD(V_lr).set_range(0, 1e10).set_default(.01);
    DMLC_DECLARE_FIELD(V_lr_beta).set_range(0, 10).set_default(1);
    DMLC_DECLARE_FIELD(V_init_scale).set_range(0, 10).set_default(.01);
    DMLC_DECLARE_FIELD(V_threshold).set_default(10);
    DMLC_DECLARE_FIELD(V_dim);
    DMLC_DECLARE_FIELD(seed).set_default(0);
  }
};

/*! \brief keys that exist only in the device build (all optional) */
struct DeviceParam : public dmlc::Parameter<DeviceParam> {
  /*! \brief rows of the model table in HBM; 0 (default): the table grows as the reference's hash map does
   *  (sgd_updater.h:78), > 0: a fixed capacity, overflow is an error */
  unsigned long long table_capacity;
  /*! \brief "hash": order-independent V init; "refrand": the reference's rand_r chain, bit for bit */
  std::string V_init;
  /*! \brief "fused": the whole worker step on device; "literal": Store/Loss calls with host arrays */
  std::string device_path;
  /*! \brief sharded store: "balanced" cuts the key space at the quantiles of a sample of the data's keys (every rank
   *  samples its own part, the samples are gathered); "uniform": owner = key / ceil(2^64 / ranks) */
  std::string shard_ranges;
  /*! \brief sharded store: "overlap" keeps two minibatches in flight like the reference's batch tracker
   *  (sgd_learner.cc:219-223; rows pulled from other owners are at most one minibatch stale), "sync" one (zero staleness) */
  std::string shard_exchange;
  /*! \brief ids per row the device feed's row buffers and batch objects are first sized for (the criteo rows of the
   *  reference's example have 39); data with more ids per row re-creates them at the size it needs */
  int feed_ids_per_row;
  DMLC_DECLARE_PARAMETER(DeviceParam) {
    DMLC_DECLARE_FIELD(feed_ids_per_row).set_range(1, 1 << 20).set_default(48);
    DMLC_DECLARE_FIELD(shard_ranges).set_default("balanced");
    DMLC_DECLARE_FIELD(shard_exchange).set_default("overlap");
    DMLC_DECLARE_FIELD(table_capacity).set_default(0);
    DMLC_DECLARE_FIELD(V_init).set_default("refrand");
    DMLC_DECLARE_FIELD(device_path).set_default("fused");
  }
};

}  // namespace difacto
#endif  // DIFACTO_HOST_SGD_PARAM_H_
