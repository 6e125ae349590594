/**
 * difacto/updater.h — Updater: owns the model and applies what the Store
 * receives.  Interface-compatible with the reference's include/difacto/updater.h
 * (Init/Load/Save/Get/Update, :33-69).
 */
#ifndef DIFACTO_UPDATER_H_
#define DIFACTO_UPDATER_H_
#include <string>
#include <vector>
#include "./base.h"
#include "./sarray.h"
#include "dmlc/io.h"

namespace difacto {

class Updater {
 public:
  Updater() {}
  virtual ~Updater() {}
  /*! \brief consume known kwargs, return the unknown remainder */
  virtual KWArgs Init(const KWArgs& kwargs) = 0;
  /*! \brief read a model; *has_aux tells whether optimiser state came with it */
  virtual void Load(dmlc::Stream* fi, bool* has_aux) = 0;
  /*! \brief write the model, with optimiser state iff save_aux */
  virtual void Save(bool save_aux, dmlc::Stream* fo) const = 0;
  /**
   * \brief values of the given features.  For data_type == Store::kWeight the
   *        layout is ragged: per key w, then V[0..V_dim) iff allocated;
   *        data_offset holds the per-key lengths (empty when V_dim == 0).
   */
  virtual void Get(const SArray<feaid_t>& fea_ids, int data_type, SArray<real_t>* data,
                   SArray<int>* data_offset) = 0;
  /*! \brief apply received data (feature counts or gradients, same ragged layout) */
  virtual void Update(const SArray<feaid_t>& fea_ids, int data_type, const SArray<real_t>& data,
                      const SArray<int>& data_offset) = 0;
};

}  // namespace difacto
#endif  // DIFACTO_UPDATER_H_
