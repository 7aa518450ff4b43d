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

// Tile plan of the LDS-staged basis kernel (k_basis_tile): the real rows, in compact order, are cut into tiles of
// consecutive rows whose merged-CSR neighbourhoods have a small UNION (the coarsening-tree order keeps neighbours
// close: 30 consecutive rows touch ~125 distinct rows, not 30 x 21).  A block stages the union rows of one sample in
// LDS once (coalesced whole-row loads) and every output row then gathers from LDS.
struct TilePlan {
  int ntiles = 0;
  int* tile_row = nullptr;    // [ntiles+1]  compact row range of each tile
  int* tile_u = nullptr;      // [ntiles+1]  range of each tile in ucol
  int* ucol = nullptr;        // union source rows (vertex id >> shift), ascending within a tile
  int* erow = nullptr;        // [n_real+1]  entry range of each compact row
  float4* ent = nullptr;      // per entry {a, b, bits(local index into the tile's union), 0}, merged-CSR order
  float* tile_cnt = nullptr;  // [ntiles]    rows of each tile (as float: the weight of a tile's BatchNorm partials)
  // The same operator as a DENSE block per tile, for the gather on the matrix cores (k_cheb_mg_gemm): row (p, i) =
  // coefficients of output row i of plane p (0: a, 1: b) over the tile's union columns, zero padded to TILE_UPAD, times
  // 2^lt_exp, cut into two fp16 slices and laid out as MFMA B fragments in 16-byte units:
  // ltx[tile][u / 16][slice][(u / 8) % 2][p * 32 + i][u % 8].
  unsigned short* ltx = nullptr;
  int lt_exp = 0;
  // (round 5) the same block as three EXACT bf16 slices of the fp32 coefficients, unscaled, same unit layout with three
  // slices per 16-column step: ltx3[tile][u / 16][slice][(u / 8) % 2][p * 32 + i][u % 8] - the matrix-core gather in the
  // three-bf16-slice arithmetic
  unsigned short* ltx3 = nullptr;
};
constexpr int TILE_UPAD = 128;                       // union columns of the dense block (>= TILE_UCAP, a multiple of 16)
constexpr int TILE_LTX_ELEMS = (TILE_UPAD / 16) * 2 * 64 * 16;     // uint16 per tile: 32 KB
constexpr int TILE_LTX3_ELEMS = (TILE_UPAD / 16) * 3 * 64 * 16;    // 48 KB
// (measured and rejected: 64-row tiles / 240 union rows - halving the resident blocks costs more than the 22 % fewer L2-side
//  reads bring; 80 / 512 and 100 / 768 at three resident blocks: DESIGN.md section 6)
constexpr int TILE_RMAX = 32;      // rows per tile
constexpr int TILE_UCAP = 120;     // union rows per tile: 120 x 512 B = 60 KB of LDS at 128 features
constexpr int TILE_ECAP = 896;     // entries per tile: 14 KB of LDS

// Device-side CSR of one coarsening level.  `col/a/b` is the *merged* pattern of L and
// L2 = 2*L*L - I: T1 = sum a*x[col], T2 = sum b*x[col] are produced by ONE gather pass.
struct Graph {
  int V = 0;
  int nnz = 0;       // merged pattern
  int nnz_L = 0;     // pattern of L alone (for reporting)
  int max_row = 0;
  int plane_bits = 0;      // ceil(log2(max row sum of |a|, |b|)) >= 0: |L x|, |L2 x| <= 2^plane_bits max |x|
  int* rowptr = nullptr;   // [V+1]
  int* col = nullptr;      // [nnz]
  float* a = nullptr;      // [nnz]  coefficients of L
  float* b = nullptr;      // [nnz]  coefficients of 2*L*L - I
  // Fake (padding) vertices are isolated: their merged row is the diagonal alone, with the SAME (fake_a, fake_b)
  // for the whole level.  For them T1 = fake_a*x and T2 = fake_b*x, so the contraction needs only K = Fin with
  // W0 + fake_a*W1 + fake_b*W2.  real_ids / fake_ids list the two vertex sets for the row-set kernels (fake_ids
  // ascending; real_ids in the LOCALITY order of the tile plans since round 5 - nothing depends on either being sorted).
  int n_real = 0, n_fake = 0;
  int* real_ids = nullptr;   // [n_real]
  int* fake_ids = nullptr;   // [n_fake]
  float fake_a = 0.f, fake_b = 0.f;
  TilePlan plan[3];          // [in_shift] and [2] = the paired operator; ntiles == 0: no plan
  // Paired operator (plan[2]): output row c = merged row 2c + merged row 2c+1 of this level, i.e. S L and S L2 with S the
  // pair-sum (the transpose of the x2 un-pool).  Defined over the V/2 vertices of the next-coarser level: those with at
  // least one real child here (pair_real_ids, compact plane order) and those whose children are both fake
  // (pair_fake_ids: S L g = fake_a S g, S L2 g = fake_b S g).
  int n_pair_real = 0, n_pair_fake = 0;
  int* pair_real_ids = nullptr;   // [n_pair_real]  coarse vertex ids, ascending
  int* pair_fake_ids = nullptr;   // [n_pair_fake]
  // Classes of IDENTICAL fake rows (p2m_graph_set_classes).  All descendants of a fake vertex of a coarser level are fake,
  // isolated, and went through the same per-row arithmetic from the same un-pooled value: they are bitwise equal, and
  // in the tree order they form an aligned block of 2^j consecutive rows.  Only the first row of a block (the
  // representative) is computed; the others ("holes") are never written nor read.  After set_classes: fake_ids /
  // pair_fake_ids list representatives only, and
  //   w[v]      = 1 (real vertex), class size (representative), 0 (hole)             -- nullptr: no classes
  //   rep_of[v] = representative of v's class (v itself for real vertices and representatives)
  //   fake_wts  = class size per entry of fake_ids;  fake_tile_w = the sums of fake_wts over 128-entry tiles
  // Forward: statistics count a representative w times.  Backward: a representative carries the SUM of its class's
  // gradients (everything downstream is linear in it).
  float* w = nullptr;
  int* rep_of = nullptr;
  float* fake_wts = nullptr;
  float* fake_tile_w = nullptr;
  int n_fake_all = 0;             // fake vertices of the level (representatives + holes)
  int* live_ids = nullptr;        // [n_live]  real vertices and representatives, ascending: what the streaming passes walk
  int n_live = 0;
  int* live_pairs = nullptr;      // [n_live_pairs]  coarse vertices c with a live child (2c or 2c+1), ascending
  int n_live_pairs = 0;
};

// Row set of a kernel launch: logical row (b, i), i < n  ->  actual row b*V + ids[i]   (ids == nullptr: identity)
struct RowSet {
  const int* ids = nullptr;
  int n = 0;
  int V = 0;
};
// 1 = real vertices, 2 = fake vertices; 3 / 4 = the paired sets, over the V/2 rows of the next-coarser level
static inline bool row_set_valid(int which) { return which >= 1 && which <= 4; }
static inline RowSet row_set_of(const Graph& g, int which) {
  RowSet r;
  r.V = which >= 3 ? g.V / 2 : g.V;
  if (which == 1) { r.ids = g.real_ids; r.n = g.n_real; }
  else if (which == 2) { r.ids = g.fake_ids; r.n = g.n_fake; }
  else if (which == 3) { r.ids = g.pair_real_ids; r.n = g.n_pair_real; }
  else { r.ids = g.pair_fake_ids; r.n = g.n_pair_fake; }
  return r;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
#ifdef __HIPCC__
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
#endif

#ifdef __HIPCC__
// amax words (include/p2m.h, P2M_ARITH_F16X2): max |v| over a wave -> one atomic max on the word (the bits of non-negative
// floats order as unsigned integers; NaNs are skipped by fmaxf).  Every lane of the wave must get here.
// The word is read first (device-scope load: past the CU's vector cache, which atomics do not update) and the atomic only
// issued by a wave that raises it: a few per launch instead of one per wave - 220 000 same-address atomics made
// k_bn_act_fwd 6.5x slower.
__device__ __forceinline__ void amax_commit(unsigned* word, float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    const unsigned bits = __float_as_uint(m);
    if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
  }
}
// |v| as it enters an amax word: NaN and infinity do not count (a non-finite element then poisons its own row of the
// contraction, as in fp32, instead of the scale of the whole tensor)
__device__ __forceinline__ float amax_abs(float v) {
  const float a = fabsf(v);
  return a < __builtin_inff() ? a : 0.f;
}
__device__ __forceinline__ float amax4(float m, const float* o) {
  return fmaxf(fmaxf(m, fmaxf(amax_abs(o[0]), amax_abs(o[1]))), fmaxf(amax_abs(o[2]), amax_abs(o[3])));
}
#endif

#define P2M_CHECK_ARG(cond, msg)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      p2m::set_error("%s: %s", __func__, msg);         \
      return P2M_ERR_INVALID;                          \
    }                                                  \
  } while (0)

}  // namespace p2m
