/**
 * difacto/store.h — Store: how a worker reads and writes the shared model.
 * Interface-compatible with the reference's include/difacto/store.h (:53-100):
 * asynchronous Push/Pull returning a timestamp, Wait(timestamp), group sizes,
 * and the Updater the server side applies.
 */
#ifndef DIFACTO_STORE_H_
#define DIFACTO_STORE_H_
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "./base.h"
#include "./sarray.h"
#include "./updater.h"
#include "dmlc/io.h"
#include "dmlc/parameter.h"

namespace difacto {

class Store {
 public:
  /*! \brief factory: the device-resident store on a GPU box (see difacto_amd/host/device_store.h) */
  static Store* Create();
  Store() {}
  virtual ~Store() {}

  /*! \brief value types (reference: store.h:31-33) */
  static const int kFeaCount = 1;
  static const int kWeight = 2;
  static const int kGradient = 3;

  virtual KWArgs Init(const KWArgs& kwargs) = 0;

  /**
   * \brief send (key, value) data to the model; returns a timestamp for Wait().
   *        vals/lens use the Updater's ragged layout; lens may be empty.
   */
  virtual int Push(const SArray<feaid_t>& fea_ids, int val_type, const SArray<real_t>& vals,
                   const SArray<int>& lens, const std::function<void()>& on_complete = nullptr) = 0;
  /*! \brief fetch the values of the keys; outputs stay valid until on_complete ran */
  virtual int Pull(const SArray<feaid_t>& fea_ids, int val_type, SArray<real_t>* vals, SArray<int>* lens,
                   const std::function<void()>& on_complete = nullptr) = 0;
  /*! \brief block until the Push/Pull with this timestamp finished */
  virtual void Wait(int time) = 0;

  virtual int NumWorkers() = 0;
  virtual int NumServers() = 0;
  virtual int Rank() = 0;

  void SetUpdater(const std::shared_ptr<Updater>& updater) { updater_ = updater; }
  std::shared_ptr<Updater> updater() { return updater_; }

 protected:
  std::shared_ptr<Updater> updater_;
};

}  // namespace difacto
#endif  // DIFACTO_STORE_H_
