// Stand-in for dmlc-core's omp.h: pull in OpenMP, or serial fallbacks.
#ifndef SHIM_DMLC_OMP_H_
#define SHIM_DMLC_OMP_H_
#if defined(_OPENMP)
#include <omp.h>
#else
inline int omp_get_thread_num() { return 0; }
inline int omp_get_num_threads() { return 1; }
inline int omp_get_max_threads() { return 1; }
inline int omp_get_num_procs() { return 1; }
inline void omp_set_num_threads(int) {}
#endif
#endif  // SHIM_DMLC_OMP_H_
