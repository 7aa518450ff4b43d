// Internal helpers shared by the HIP translation units of libp2m_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/p2m.h"

namespace p2m {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Device-side CSR of one coarsening level.  `col/a/b` is the *merged* pattern of L and
// L2 = 2*L*L - I: T1 = sum a*x[col], T2 = sum b*x[col] are produced by ONE gather pass.
struct Graph {
  int V = 0;
  int nnz = 0;       // merged pattern
  int nnz_L = 0;     // pattern of L alone (for reporting)
  int max_row = 0;
  int* rowptr = nullptr;   // [V+1]
  int* col = nullptr;      // [nnz]
  float* a = nullptr;      // [nnz]  coefficients of L
  float* b = nullptr;      // [nnz]  coefficients of 2*L*L - I
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

#define P2M_CHECK_ARG(cond, msg)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      p2m::set_error("%s: %s", __func__, msg);         \
      return P2M_ERR_INVALID;                          \
    }                                                  \
  } while (0)

}  // namespace p2m
