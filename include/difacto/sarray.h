/**
 * difacto/sarray.h — SArray<T>: the shared-ownership array every interface of
 * the path trades in (copies share storage; SArray<char>(SArray<T>) reinterprets
 * bytes).  The reference aliases ps-lite's ps::SArray (sarray.h:35); ps-lite is
 * not vendored here, third_party_shim/ps/sarray.h supplies a from-scratch one.
 */
#ifndef DIFACTO_SARRAY_H_
#define DIFACTO_SARRAY_H_
#include "ps/sarray.h"
namespace difacto {
template <typename T>
using SArray = ps::SArray<T>;
}  // namespace difacto
#endif  // DIFACTO_SARRAY_H_
