// Sparse (Chebyshev-basis) stage of the Pose2Mesh GCN on gfx950 -- the HBM-bound part.
//
// Reference arithmetic: lib/models/backbones/cheby_graph_conv.py:16-34
//     x1 = L x0 ;  x2 = 2 L x1 - x0      (two cuSPARSE SpMMs + 4 layout shuffles per conv)
// Here: activations stay (B, V, F) row-major (F contiguous); per level the merged CSR of L and
// L2 = 2 L L - I is baked once (p2m_graph_create), and ONE gather pass produces both planes:
//     T1[r] = sum_j a_j x[col_j],  T2[r] = sum_j b_j x[col_j].
// Each row group of F/4 lanes streams whole feature rows as float4 (a wave reads 1 KiB
// contiguous per gather when F=256, 2 rows of 512 B when F=128): coalesced, every neighbour row
// is re-used from L2 (one sample's level is <= 6 MB).  A block owns a contiguous tile of rows of
// ONE sample, and block ids are swizzled so that consecutive tiles of a sample share an XCD (L2).
// Algorithmic HBM bytes: read X once, write T1 and T2 once.
// The row-per-wave kernels (k_basis_fwd / k_basis_bwd) are bound by the gathered volume through the texture path
// (~21 rows gathered per real output row); k_basis_tile stages the UNION of the neighbourhoods of 32 consecutive real
// rows in LDS (~4 rows loaded per output row) and is the default for the real rows of levels that have padding vertices.
#include <cstdlib>

#include "p2m_common.h"

namespace p2m {

constexpr int ROWS_PER_BLOCK = 4;   // rows per block (one per wave): small tiles keep an XCD's working window in its L2
                                    // (measured B=256,V=11776,F=128: bwd 1.5 -> 2.6 TB/s going from 16 to 4)

// samples a tile block walks (amortises the tile tables; measured 8 -> 4 -> 2 -> 1: 3 198 -> 3 112 -> 2 977 -> 2 631 GB/s)
static int basis_spb() { return 8; }

// P2M_BASIS_TILED=0 falls back to the row-per-wave gather kernel for the real rows of split levels (the independent
// kernel set of the parity tests)
static bool basis_tiled() {
  static int v = [] { const char* e = getenv("P2M_BASIS_TILED"); return e ? atoi(e) : 1; }();
  return v != 0;
}

__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  // observed dispatch: block b runs on XCD b % 8 -> give each XCD a contiguous range of logical ids
  return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
}

__device__ __forceinline__ void fma4(float4& acc, float s, const float4& x) {
  acc.x = fmaf(s, x.x, acc.x); acc.y = fmaf(s, x.y, acc.y);
  acc.z = fmaf(s, x.z, acc.z); acc.w = fmaf(s, x.w, acc.w);
}

// One wave = ONE vertex row x S = 64/LPR samples (LPR = F/4 lanes per sample).  The CSR row (col, a, b) is
// wave-uniform, so it is fetched with scalar loads and the vector memory pipe only carries the feature
// gathers; there is no divergence on the row length.
template <int LPR>
__global__ __launch_bounds__(256) void k_basis_fwd(Graph g, const float* __restrict__ X, float* __restrict__ T1,
                                                    float* __restrict__ T2, int B, int in_shift, int tiles_per_group,
                                                    const int* __restrict__ ids, int nset) {
  constexpr int F = LPR * 4;
  constexpr int S = 64 / LPR;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int group = lid / tiles_per_group;
  const int tile = lid - group * tiles_per_group;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int sample = group * S + lane / LPR;
  const int f4 = (lane % LPR) * 4;
  const bool active = sample < B;
  const float* Xb = X + (long)(active ? sample : 0) * (g.V >> in_shift) * F + f4;
  // ids != nullptr: only the listed (real) vertices are computed and the planes are written COMPACT, [B*nset, F]
  const long obase = (long)(active ? sample : 0) * nset * F + f4;
  int row_end = (tile + 1) * ROWS_PER_BLOCK;
  if (row_end > nset) row_end = nset;
  for (int lrow = tile * ROWS_PER_BLOCK + wave; lrow < row_end; lrow += 4) {
    const int row = ids ? ids[lrow] : lrow;
    const int s = g.rowptr[row], e = g.rowptr[row + 1];
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
    int j = s;
    for (; j + 8 <= e; j += 8) {       // 8 gathers in flight per lane: the kernel is bound by memory-level parallelism
      float4 x[8];
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = *reinterpret_cast<const float4*>(Xb + (long)(g.col[j + i] >> in_shift) * F);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        fma4(t1, g.a[j + i], x[i]);
        fma4(t2, g.b[j + i], x[i]);
      }
    }
    for (; j + 4 <= e; j += 4) {
      const int c0 = g.col[j], c1 = g.col[j + 1], c2 = g.col[j + 2], c3 = g.col[j + 3];
      const float4 x0 = *reinterpret_cast<const float4*>(Xb + (long)(c0 >> in_shift) * F);
      const float4 x1 = *reinterpret_cast<const float4*>(Xb + (long)(c1 >> in_shift) * F);
      const float4 x2 = *reinterpret_cast<const float4*>(Xb + (long)(c2 >> in_shift) * F);
      const float4 x3 = *reinterpret_cast<const float4*>(Xb + (long)(c3 >> in_shift) * F);
      fma4(t1, g.a[j], x0); fma4(t2, g.b[j], x0);
      fma4(t1, g.a[j + 1], x1); fma4(t2, g.b[j + 1], x1);
      fma4(t1, g.a[j + 2], x2); fma4(t2, g.b[j + 2], x2);
      fma4(t1, g.a[j + 3], x3); fma4(t2, g.b[j + 3], x3);
    }
    for (; j < e; j++) {
      const float4 x0 = *reinterpret_cast<const float4*>(Xb + (long)(g.col[j] >> in_shift) * F);
      fma4(t1, g.a[j], x0); fma4(t2, g.b[j], x0);
    }
    if (active) {
      *reinterpret_cast<float4*>(T1 + obase + (long)lrow * F) = t1;
      *reinterpret_cast<float4*>(T2 + obase + (long)lrow * F) = t2;
    }
  }
}

// any F (first conv F=5): one thread per (b, row, f)
__global__ void k_basis_fwd_generic(Graph g, const float* __restrict__ X, float* __restrict__ T1,
                                    float* __restrict__ T2, int B, int F, int in_shift) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * g.V * F;
  if (idx >= tot) return;
  int f = (int)(idx % F);
  long rv = idx / F;
  int row = (int)(rv % g.V);
  int b = (int)(rv / g.V);
  const float* Xb = X + (long)b * (g.V >> in_shift) * F + f;
  float t1 = 0.f, t2 = 0.f;
  for (int j = g.rowptr[row]; j < g.rowptr[row + 1]; j++) {
    float x = Xb[(long)(g.col[j] >> in_shift) * F];
    t1 = fmaf(g.a[j], x, t1);
    t2 = fmaf(g.b[j], x, t2);
  }
  T1[idx] = t1;
  T2[idx] = t2;
}

// dX[p] = sum_{children r of p} ( d0[r] + resid[r] + sum_j a_j d1[col_j] + b_j d2[col_j] )
// same wave mapping as the forward: one wave = one OUTPUT row x S samples.
template <int LPR>
__global__ __launch_bounds__(256) void k_basis_bwd(Graph g, const float* __restrict__ d0, const float* __restrict__ d1,
                                                    const float* __restrict__ d2, const float* __restrict__ resid,
                                                    float* __restrict__ dX, int B, int out_shift, int tiles_per_group) {
  constexpr int F = LPR * 4;
  constexpr int S = 64 / LPR;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int group = lid / tiles_per_group;
  const int tile = lid - group * tiles_per_group;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int sample = group * S + lane / LPR;
  const int f4 = (lane % LPR) * 4;
  const bool active = sample < B;
  const long ibase = (long)(active ? sample : 0) * g.V * F + f4;
  const int Vout = g.V >> out_shift;
  const long obase = (long)(active ? sample : 0) * Vout * F + f4;
  const int nchild = 1 << out_shift;
  int p_end = (tile + 1) * ROWS_PER_BLOCK;
  if (p_end > Vout) p_end = Vout;
  for (int p = tile * ROWS_PER_BLOCK + wave; p < p_end; p += 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ch = 0; ch < nchild; ch++) {
      const int row = (p << out_shift) + ch;
      const float4 v = *reinterpret_cast<const float4*>(d0 + ibase + (long)row * F);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (resid != nullptr) {
        const float4 q = *reinterpret_cast<const float4*>(resid + ibase + (long)row * F);
        acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
      }
      const int s = g.rowptr[row], e = g.rowptr[row + 1];
      int j = s;
      for (; j + 2 <= e; j += 2) {
        const int c0 = g.col[j], c1 = g.col[j + 1];
        const float4 u0 = *reinterpret_cast<const float4*>(d1 + ibase + (long)c0 * F);
        const float4 w0 = *reinterpret_cast<const float4*>(d2 + ibase + (long)c0 * F);
        const float4 u1 = *reinterpret_cast<const float4*>(d1 + ibase + (long)c1 * F);
        const float4 w1 = *reinterpret_cast<const float4*>(d2 + ibase + (long)c1 * F);
        fma4(acc, g.a[j], u0); fma4(acc, g.b[j], w0);
        fma4(acc, g.a[j + 1], u1); fma4(acc, g.b[j + 1], w1);
      }
      for (; j < e; j++) {
        const int c0 = g.col[j];
        const float4 u0 = *reinterpret_cast<const float4*>(d1 + ibase + (long)c0 * F);
        const float4 w0 = *reinterpret_cast<const float4*>(d2 + ibase + (long)c0 * F);
        fma4(acc, g.a[j], u0); fma4(acc, g.b[j], w0);
      }
    }
    if (active) *reinterpret_cast<float4*>(dX + obase + (long)p * F) = acc;
  }
}

__global__ void k_basis_bwd_generic(Graph g, const float* __restrict__ d0, const float* __restrict__ d1,
                                    const float* __restrict__ d2, const float* __restrict__ resid,
                                    float* __restrict__ dX, int B, int F, int out_shift) {
  const int Vout = g.V >> out_shift;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)B * Vout * F;
  if (idx >= tot) return;
  int f = (int)(idx % F);
  long rv = idx / F;
  int p = (int)(rv % Vout);
  int b = (int)(rv / Vout);
  const long ibase = (long)b * g.V * F + f;
  float acc = 0.f;
  for (int ch = 0; ch < (1 << out_shift); ch++) {
    int row = (p << out_shift) + ch;
    acc += d0[ibase + (long)row * F];
    if (resid) acc += resid[ibase + (long)row * F];
    for (int j = g.rowptr[row]; j < g.rowptr[row + 1]; j++) {
      long o = ibase + (long)g.col[j] * F;
      acc = fmaf(g.a[j], d1[o], acc);
      acc = fmaf(g.b[j], d2[o], acc);
    }
  }
  dX[idx] = acc;
}

// ---- narrow-output convolution by linearity (the final 64 -> 3 layer) --------------------------
//   y = [x | Lx | L2x] W  ==  P0 + L P1 + L2 P2   with  P = x [W0|W1|W2]  (a [M, 3*nc] GEMM first)
// so the sparse stage runs on nc = 3 columns instead of Fin = 64 (4.5x less HBM traffic than basis+GEMM).
// P rows are `ldp` floats wide: columns [0,nc) = P0, [nc,2nc) = P1, [2nc,3nc) = P2.
template <int NC>
__global__ __launch_bounds__(256) void k_combine_small(Graph g, const float* __restrict__ P, int ldp,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int B,
                                                        const int* __restrict__ ids, int nset,
                                                        const int* __restrict__ out_index, int out_rows, float scale,
                                                        int skip_real) {
  // ids != nullptr: only the listed (real) vertices are computed; out_index != nullptr: vertex v is stored at row
  // out_index[v] of a [B, out_rows, NC] tensor (mesh-model vertex order), scaled -- the Tester's / demo's
  // pred_mesh[:, graph_perm_reverse[:nv], :] * scale (lib/core/base.py:201-202) folded into the last conv's store
  const int n = ids ? nset : g.V;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * n) return;
  const int b = (int)(idx / n);
  const int i = (int)(idx - (long)b * n);
  const int row = ids ? ids[i] : i;
  // skip_real: the rows with neighbours were done by k_combine_small_tile; this launch covers the single-entry rows
  if (skip_real && g.rowptr[row + 1] - g.rowptr[row] > 1) return;
  const long base = (long)b * g.V * ldp;             // sample offset in P
  // classes (Graph::rep_of): a hole is never computed upstream; its output is its representative's, recomputed here
  const int src = g.rep_of ? g.rep_of[row] : row;
  const float* p0 = P + base + (long)src * ldp;
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) acc[c] = p0[c] + (bias ? bias[c] : 0.f);
  for (int j = g.rowptr[src]; j < g.rowptr[src + 1]; j++) {
    const float* q = P + base + (long)g.col[j] * ldp;
    const float a = g.a[j], bb = g.b[j];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = fmaf(bb, q[2 * NC + c], fmaf(a, q[NC + c], acc[c]));
  }
  long orow = row;
  int rows_out = g.V;
  if (out_index) {
    orow = out_index[row];
    rows_out = out_rows;
    if (orow < 0) return;
  }
  float* y = Y + ((long)b * rows_out + orow) * NC;
#pragma unroll
  for (int c = 0; c < NC; c++) y[c] = out_index ? acc[c] * scale : acc[c];
}

// E[r] = [ G[r] | (L G)[r] | (L2 G)[r] | 0 ... ]   (row width lde >= 3*NC): the basis of a narrow gradient
template <int NC>
__global__ __launch_bounds__(256) void k_expand_small(Graph g, const float* __restrict__ G, float* __restrict__ E,
                                                       int lde, int B, int skip_real) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * g.V) return;
  const int row = (int)(idx % g.V);
  if (skip_real && g.rowptr[row + 1] - g.rowptr[row] > 1) return;     // done by k_expand_small_tile
  const long base = (idx - row) * NC;
  float t1[NC], t2[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) t1[c] = t2[c] = 0.f;
  for (int j = g.rowptr[row]; j < g.rowptr[row + 1]; j++) {
    const float* q = G + base + (long)g.col[j] * NC;
    const float a = g.a[j], b = g.b[j];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      t1[c] = fmaf(a, q[c], t1[c]);
      t2[c] = fmaf(b, q[c], t2[c]);
    }
  }
  float* e = E + idx * lde;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    e[c] = G[idx * NC + c];
    e[NC + c] = t1[c];
    e[2 * NC + c] = t2[c];
  }
  for (int c = 3 * NC; c < lde; c++) e[c] = 0.f;
}

// LDS-staged forms of the two kernels above for the rows that HAVE neighbours, on levels with a tile plan (TilePlan,
// p2m_common.h): one block = one tile (<= 32 real rows) x 8 samples.  The 2 NC (combine) / NC (expand) floats each union
// row contributes are staged once per (tile, sample) instead of being fetched once per referencing row (~21x through
// L1/L2, a 128-byte line each): 537 -> ~170 us per launch at the finest level.  Entry order and fmaf chains are
// those of the row kernels: bitwise the same results.
constexpr int SMALL_SPB = 8;
template <int NC>
__global__ __launch_bounds__(256) void k_combine_small_tile(TilePlan pl, const int* __restrict__ real_ids, int V,
                                                             const float* __restrict__ P, int ldp,
                                                             const float* __restrict__ bias, float* __restrict__ Y, int B,
                                                             const int* __restrict__ out_index, int out_rows,
                                                             float scale) {
  __shared__ float4 ents[TILE_ECAP];
  __shared__ int rowoff[TILE_RMAX + 1];
  __shared__ float xs[SMALL_SPB][TILE_UCAP][2 * NC];
  const int tile = blockIdx.x % pl.ntiles, sg = blockIdx.x / pl.ntiles;
  const int t = threadIdx.x;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  const int e0 = pl.erow[r0], nE = pl.erow[r0 + R] - e0;
  for (int i = t; i < nE; i += 256) ents[i] = pl.ent[e0 + i];
  if (t <= R) rowoff[t] = pl.erow[r0 + t] - e0;
  const int b0 = sg * SMALL_SPB;
  for (int q = t; q < U * SMALL_SPB; q += 256) {
    const int u = q % U, sl = q / U, b = b0 + sl;
    if (b < B) {
      const float* src = P + ((long)b * V + pl.ucol[u0 + u]) * ldp + NC;
#pragma unroll
      for (int c = 0; c < 2 * NC; c++) xs[sl][u][c] = src[c];
    }
  }
  __syncthreads();
  const int i = t % TILE_RMAX, sl = t / TILE_RMAX, b = b0 + sl;
  if (i >= R || b >= B) return;
  const int row = real_ids[r0 + i];
  const float* p0 = P + ((long)b * V + row) * ldp;
  float acc[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) acc[c] = p0[c] + (bias ? bias[c] : 0.f);
  for (int j = rowoff[i]; j < rowoff[i + 1]; j++) {
    const float4 en = ents[j];
    const float* q = xs[sl][__float_as_int(en.z)];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = fmaf(en.y, q[NC + c], fmaf(en.x, q[c], acc[c]));
  }
  long orow = row;
  int rows_out = V;
  if (out_index) {
    orow = out_index[row];
    rows_out = out_rows;
    if (orow < 0) return;
  }
  float* y = Y + ((long)b * rows_out + orow) * NC;
#pragma unroll
  for (int c = 0; c < NC; c++) y[c] = out_index ? acc[c] * scale : acc[c];
}

template <int NC>
__global__ __launch_bounds__(256) void k_expand_small_tile(TilePlan pl, const int* __restrict__ real_ids, int V,
                                                            const float* __restrict__ G, float* __restrict__ E, int lde,
                                                            int B) {
  __shared__ float4 ents[TILE_ECAP];
  __shared__ int rowoff[TILE_RMAX + 1];
  __shared__ float xs[SMALL_SPB][TILE_UCAP][NC];
  const int tile = blockIdx.x % pl.ntiles, sg = blockIdx.x / pl.ntiles;
  const int t = threadIdx.x;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  const int e0 = pl.erow[r0], nE = pl.erow[r0 + R] - e0;
  for (int i = t; i < nE; i += 256) ents[i] = pl.ent[e0 + i];
  if (t <= R) rowoff[t] = pl.erow[r0 + t] - e0;
  const int b0 = sg * SMALL_SPB;
  for (int q = t; q < U * SMALL_SPB; q += 256) {
    const int u = q % U, sl = q / U, b = b0 + sl;
    if (b < B) {
      const float* src = G + ((long)b * V + pl.ucol[u0 + u]) * NC;
#pragma unroll
      for (int c = 0; c < NC; c++) xs[sl][u][c] = src[c];
    }
  }
  __syncthreads();
  const int i = t % TILE_RMAX, sl = t / TILE_RMAX, b = b0 + sl;
  if (i >= R || b >= B) return;
  const long ridx = (long)b * V + real_ids[r0 + i];
  float t1[NC], t2[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) t1[c] = t2[c] = 0.f;
  for (int j = rowoff[i]; j < rowoff[i + 1]; j++) {
    const float4 en = ents[j];
    const float* q = xs[sl][__float_as_int(en.z)];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      t1[c] = fmaf(en.x, q[c], t1[c]);
      t2[c] = fmaf(en.y, q[c], t2[c]);
    }
  }
  float* e = E + ridx * lde;
#pragma unroll
  for (int c = 0; c < NC; c++) {
    e[c] = G[ridx * NC + c];
    e[NC + c] = t1[c];
    e[2 * NC + c] = t2[c];
  }
  for (int c = 3 * NC; c < lde; c++) e[c] = 0.f;
}

}  // namespace p2m

using namespace p2m;

static int combine_small_launch(p2m_graph_t gh, const float* P, int32_t ldp, int32_t nc, const float* bias, float* Y,
                                int32_t B, int real_only, const int32_t* out_index, int32_t out_rows, float scale,
                                void* stream) {
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  const int* ids = real_only ? g.real_ids : nullptr;
  const int nset = real_only ? g.n_real : g.V;
  const long tot = (long)B * nset;
  if (tot == 0) return P2M_OK;
  hipStream_t s = (hipStream_t)stream;
  int skip_real = 0;
  // rows with neighbours through the tile plan (when the level has one); the row kernel then covers the rest
  const TilePlan& pl = g.plan[0];
  if (pl.ntiles > 0 && basis_tiled() && nc == 3) {
    hipLaunchKernelGGL(k_combine_small_tile<3>, dim3(pl.ntiles * cdiv(B, SMALL_SPB)), dim3(256), 0, s, pl, g.real_ids, g.V,
                       P, ldp, bias, Y, B, out_index, out_rows, scale);
    if (real_only) return check_launch("cheb_combine_small(tiled)");
    skip_real = 1;
  }
#define P2M_COMBINE(NCv) hipLaunchKernelGGL(k_combine_small<NCv>, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, P, ldp, bias, \
                                            Y, B, ids, nset, out_index, out_rows, scale, skip_real)
  switch (nc) {
    case 1: P2M_COMBINE(1); break;
    case 2: P2M_COMBINE(2); break;
    case 3: P2M_COMBINE(3); break;
    case 4: P2M_COMBINE(4); break;
    default: set_error("p2m_cheb_combine_small: nc must be 1..4 (got %d)", nc); return P2M_ERR_INVALID;
  }
#undef P2M_COMBINE
  return check_launch("cheb_combine_small");
}

extern "C" int p2m_cheb_combine_small(p2m_graph_t gh, const float* P, int32_t ldp, int32_t nc, const float* bias,
                                      float* Y, int32_t B, void* stream) {
  P2M_CHECK_ARG(gh && P && Y && ldp >= 3 * nc, "null pointer or ldp < 3*nc");
  if (B <= 0) return P2M_OK;
  return combine_small_launch(gh, P, ldp, nc, bias, Y, B, 0, nullptr, 0, 1.f, stream);
}

extern "C" int p2m_cheb_combine_small_real(p2m_graph_t gh, const float* P, int32_t ldp, int32_t nc, const float* bias,
                                           float* Y, int32_t B, const int32_t* out_index, int32_t out_rows,
                                           float scale, void* stream) {
  P2M_CHECK_ARG(gh && P && Y && ldp >= 3 * nc, "null pointer or ldp < 3*nc");
  P2M_CHECK_ARG(out_index == nullptr || out_rows > 0, "out_rows must be positive with out_index");
  if (B <= 0) return P2M_OK;
  return combine_small_launch(gh, P, ldp, nc, bias, Y, B, 1, out_index, out_rows, scale, stream);
}

extern "C" int p2m_cheb_expand_small(p2m_graph_t gh, const float* G, int32_t nc, float* E, int32_t lde, int32_t B,
                                     void* stream) {
  P2M_CHECK_ARG(gh && G && E && lde >= 3 * nc, "null pointer or lde < 3*nc");
  if (B <= 0) return P2M_OK;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  const long tot = (long)B * g.V;
  hipStream_t s = (hipStream_t)stream;
  int skip_real = 0;
  const TilePlan& pl = g.plan[0];
  if (pl.ntiles > 0 && basis_tiled() && nc == 3) {
    hipLaunchKernelGGL(k_expand_small_tile<3>, dim3(pl.ntiles * cdiv(B, SMALL_SPB)), dim3(256), 0, s, pl, g.real_ids, g.V,
                       G, E, lde, B);
    skip_real = 1;
  }
  switch (nc) {
    case 1: hipLaunchKernelGGL(k_expand_small<1>, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, G, E, lde, B, skip_real); break;
    case 2: hipLaunchKernelGGL(k_expand_small<2>, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, G, E, lde, B, skip_real); break;
    case 3: hipLaunchKernelGGL(k_expand_small<3>, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, G, E, lde, B, skip_real); break;
    case 4: hipLaunchKernelGGL(k_expand_small<4>, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, G, E, lde, B, skip_real); break;
    default: set_error("p2m_cheb_expand_small: nc must be 1..4 (got %d)", nc); return P2M_ERR_INVALID;
  }
  return check_launch("cheb_expand_small");
}


// ---------------------------------------------------------------------------------------------
// LDS-staged basis kernel for the real rows of a split level (TilePlan, p2m_common.h).
// One block (512 threads) = one tile (<= 32 consecutive real rows) x SPB samples x one 128-feature slice.  Per sample:
// the tile's union of source rows (~4 per output row instead of ~21 gathered rows) is loaded ONCE with whole-row
// coalesced float4 loads - into registers while the previous sample is being computed, then stored to LDS - and each
// lane group (LPR lanes = one row of FB = 4 LPR features) accumulates its output
// row from LDS: the per-entry {a, b, local index} is a broadcast ds_read_b128, the data a conflict-free ds_read_b128.
// Entry order = merged-CSR order and the same fmaf chain as k_basis_fwd: results are bitwise identical to it.
// ---------------------------------------------------------------------------------------------
namespace p2m {
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int LPR>
__global__ __launch_bounds__(512, 2) void k_basis_tile(TilePlan pl, const float* __restrict__ X,
                                                        float* __restrict__ T1, float* __restrict__ T2, int B, int F,
                                                        long x_rows, int nset, int spb,
                                                        const float* __restrict__ in_scale = nullptr,
                                                        const float* __restrict__ in_shift = nullptr) {
  constexpr int NT = 512;
  constexpr int FB = LPR * 4;                 // features per slice
  constexpr int RPI = NT / LPR;               // rows the block touches per instruction
  constexpr int NPF = (TILE_UCAP + RPI - 1) / RPI;   // union rows per lane group
  __shared__ __attribute__((aligned(16))) float xs[TILE_UCAP * FB];
  __shared__ __attribute__((aligned(16))) f32x4v ents[TILE_ECAP];
  __shared__ int rowoff[TILE_RMAX + 1];

  // block -> (feature slice, sample group, tile), tile fastest, through the XCD swizzle: the blocks resident on one
  // XCD then work on ADJACENT tiles of the same samples, whose unions overlap ~4x - the overlap is served by that
  // XCD's L2 instead of being fetched from HBM once per XCD
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int nsg = (B + spb - 1) / spb;
  if (lid >= pl.ntiles * nsg * (F / FB)) return;
  const int tile = lid % pl.ntiles;
  const int sgrp = (lid / pl.ntiles) % nsg;
  const int slice = lid / (pl.ntiles * nsg);
  const int t = threadIdx.x;
  const int grp = t / LPR;                    // row group of this lane
  const int lf = (t % LPR) * 4;
  const int l4 = lf + slice * FB;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  const int e0 = pl.erow[r0], nE = pl.erow[r0 + R] - e0;
  for (int i = t; i < nE; i += NT) ents[i] = *reinterpret_cast<const f32x4v*>(&pl.ent[e0 + i]);
  if (t <= R) rowoff[t] = pl.erow[r0 + t] - e0;
  // this lane group's union rows (the same for every sample): element offsets inside one sample of X
  long xoff[NPF];
#pragma unroll
  for (int q = 0; q < NPF; q++) {
    const int u = grp + q * RPI;
    xoff[q] = (long)pl.ucol[u0 + (u < U ? u : U - 1)] * F + l4;     // clamped: loads stay unconditional
  }

  const int b0 = sgrp * spb;
  int b1 = b0 + spb;
  if (b1 > B) b1 = B;
  // the union rows of sample b+1 are fetched into registers while sample b is computed from LDS
  f32x4v pf[NPF];
  auto issue = [&](int b) {
    const float* Xb = X + (long)b * x_rows * F;
#pragma unroll
    for (int q = 0; q < NPF; q++) pf[q] = *reinterpret_cast<const f32x4v*>(Xb + xoff[q]);
  };
  // activation on load (include/p2m.h): X holds a raw conv output y, the gathered operand is max(fma(y, scale, shift), 0) -
  // applied to the union rows on their way into LDS; this lane's four features are fixed for the whole kernel
  f32x4v asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
  if (in_scale != nullptr) {
    asc = *reinterpret_cast<const f32x4v*>(in_scale + l4);
    ash = *reinterpret_cast<const f32x4v*>(in_shift + l4);
  }
  issue(b0);
  for (int b = b0; b < b1; b++) {
    __syncthreads();                          // tables ready / the previous sample's LDS reads are done
#pragma unroll
    for (int q = 0; q < NPF; q++) {
      const int u = grp + q * RPI;
      f32x4v v = pf[q];
      if (in_scale != nullptr) {
#pragma unroll
        for (int c = 0; c < 4; c++) v[c] = fmaxf(fmaf(v[c], asc[c], ash[c]), 0.f);
      }
      if (u < U) *reinterpret_cast<f32x4v*>(&xs[u * FB + lf]) = v;
    }
    __syncthreads();
    issue(b + 1 < b1 ? b + 1 : b);            // unconditional (the last one re-reads sample b: harmless)
    for (int i = grp; i < R; i += RPI) {
      f32x4v t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
      const int s = rowoff[i], e = rowoff[i + 1];
      int j = s;
      for (; j + 4 <= e; j += 4) {
        f32x4v en[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) en[q] = ents[j + q];
#pragma unroll
        for (int q = 0; q < 4; q++)
          x[q] = *reinterpret_cast<const f32x4v*>(&xs[__float_as_int(en[q][2]) * FB + lf]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
#pragma unroll
          for (int c = 0; c < 4; c++) {
            t1[c] = fmaf(en[q][0], x[q][c], t1[c]);
            t2[c] = fmaf(en[q][1], x[q][c], t2[c]);
          }
        }
      }
      for (; j < e; j++) {
        const f32x4v en = ents[j];
        const f32x4v x = *reinterpret_cast<const f32x4v*>(&xs[__float_as_int(en[2]) * FB + lf]);
#pragma unroll
        for (int c = 0; c < 4; c++) {
          t1[c] = fmaf(en[0], x[c], t1[c]);
          t2[c] = fmaf(en[1], x[c], t2[c]);
        }
      }
      // streaming stores: the planes are read next by the GEMM, long after they have left L2 anyway - keeping them
      // out of L2 leaves it to the overlapping unions of neighbouring tiles
      const long o = ((long)b * nset + r0 + i) * F + l4;
      __builtin_nontemporal_store(t1, reinterpret_cast<f32x4v*>(T1 + o));
      __builtin_nontemporal_store(t2, reinterpret_cast<f32x4v*>(T2 + o));
    }
  }
}

}  // namespace p2m

static int basis_fwd_launch(p2m_graph_t gh, const float* X, float* T1, float* T2, int32_t B, int32_t F,
                            int32_t in_shift, int real_only, void* stream, const float* act_sc = nullptr,
                            const float* act_sh = nullptr) {
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  hipStream_t s = (hipStream_t)stream;
  const int* ids = real_only ? g.real_ids : nullptr;
  const int nset = real_only ? g.n_real : g.V;
  if (nset == 0) return P2M_OK;
  if (real_only && basis_tiled() && g.plan[in_shift].ntiles > 0 && (F == 32 || F == 64 || F % 128 == 0)) {
    const TilePlan& pl = g.plan[in_shift];
    const int spb = basis_spb();
    const long x_rows = g.V >> in_shift;
    const dim3 grid(cdiv((long)pl.ntiles * cdiv(B, spb) * (F >= 128 ? F / 128 : 1), 8) * 8);
    if (F == 32) hipLaunchKernelGGL(k_basis_tile<8>, grid, dim3(512), 0, s, pl, X, T1, T2, B, F, x_rows, nset, spb, act_sc, act_sh);
    else if (F == 64) hipLaunchKernelGGL(k_basis_tile<16>, grid, dim3(512), 0, s, pl, X, T1, T2, B, F, x_rows, nset, spb, act_sc, act_sh);
    else hipLaunchKernelGGL(k_basis_tile<32>, grid, dim3(512), 0, s, pl, X, T1, T2, B, F, x_rows, nset, spb, act_sc, act_sh);
    return check_launch("cheb_basis_fwd(tiled)");
  }
  if (act_sc != nullptr) {
    set_error("p2m_cheb_basis_fwd_real: activation on load exists in the tile-plan kernel only (this level / width / "
              "P2M_BASIS_TILED setting takes the row kernel)");
    return P2M_ERR_INVALID;
  }
  const int tps = cdiv(nset, ROWS_PER_BLOCK);
  auto grid = [&](int lpr) { return dim3(cdiv(B, 64 / lpr) * tps); };
  switch (F) {
    case 32:  hipLaunchKernelGGL(k_basis_fwd<8>,  grid(8),  dim3(256), 0, s, g, X, T1, T2, B, in_shift, tps, ids, nset); break;
    case 64:  hipLaunchKernelGGL(k_basis_fwd<16>, grid(16), dim3(256), 0, s, g, X, T1, T2, B, in_shift, tps, ids, nset); break;
    case 128: hipLaunchKernelGGL(k_basis_fwd<32>, grid(32), dim3(256), 0, s, g, X, T1, T2, B, in_shift, tps, ids, nset); break;
    case 256: hipLaunchKernelGGL(k_basis_fwd<64>, grid(64), dim3(256), 0, s, g, X, T1, T2, B, in_shift, tps, ids, nset); break;
    default:
      set_error("p2m_cheb_basis_fwd_real: feature width %d is not 32/64/128/256", F);
      return P2M_ERR_INVALID;
  }
  return check_launch("cheb_basis_fwd");
}

extern "C" int p2m_cheb_basis_pair(p2m_graph_t gh, const float* G, float* P1c, float* P2c, int32_t B, int32_t F,
                                   void* stream) {
  P2M_CHECK_ARG(gh && G && P1c && P2c && F > 0, "null pointer or empty shape");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  const TilePlan& pl = g.plan[2];
  P2M_CHECK_ARG(pl.ntiles > 0, "this level has no paired operator (p2m_graph_pair_info)");
  P2M_CHECK_ARG(F == 32 || F == 64 || F % 128 == 0, "feature width must be 32, 64 or a multiple of 128");
  if (B <= 0) return P2M_OK;
  hipStream_t s = (hipStream_t)stream;
  const int spb = basis_spb();
  const long x_rows = g.V;
  const int nset = g.n_pair_real;
  const dim3 grid(cdiv((long)pl.ntiles * cdiv(B, spb) * (F >= 128 ? F / 128 : 1), 8) * 8);
  if (F == 32) hipLaunchKernelGGL(k_basis_tile<8>, grid, dim3(512), 0, s, pl, G, P1c, P2c, B, F, x_rows, nset, spb);
  else if (F == 64) hipLaunchKernelGGL(k_basis_tile<16>, grid, dim3(512), 0, s, pl, G, P1c, P2c, B, F, x_rows, nset, spb);
  else hipLaunchKernelGGL(k_basis_tile<32>, grid, dim3(512), 0, s, pl, G, P1c, P2c, B, F, x_rows, nset, spb);
  return check_launch("cheb_basis_pair");
}

extern "C" int p2m_cheb_basis_fwd_real(p2m_graph_t gh, const float* X, float* T1c, float* T2c, int32_t B, int32_t F,
                                       int32_t in_shift, const float* act_scale, const float* act_shift, void* stream) {
  P2M_CHECK_ARG(gh && X && T1c && T2c && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(in_shift == 0 || in_shift == 1, "in_shift must be 0 or 1");
  P2M_CHECK_ARG((act_scale == nullptr) == (act_shift == nullptr), "act_scale / act_shift must both be given or both NULL");
  if (B <= 0) return P2M_OK;
  return basis_fwd_launch(gh, X, T1c, T2c, B, F, in_shift, 1, stream, act_scale, act_shift);
}

extern "C" int p2m_cheb_basis_fwd(p2m_graph_t gh, const float* X, float* T1, float* T2, int32_t B, int32_t F,
                                  int32_t in_shift, void* stream) {
  P2M_CHECK_ARG(gh && X && T1 && T2 && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(in_shift == 0 || in_shift == 1, "in_shift must be 0 or 1");
  if (B <= 0) return P2M_OK;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(in_shift == 0 || (g.V % 2 == 0), "virtual un-pool needs an even vertex count");
  hipStream_t s = (hipStream_t)stream;
  switch (F) {
    case 32: case 64: case 128: case 256:
      return basis_fwd_launch(gh, X, T1, T2, B, F, in_shift, 0, stream);
    default: {
      long tot = (long)B * g.V * F;
      hipLaunchKernelGGL(k_basis_fwd_generic, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, X, T1, T2, B, F, in_shift);
    }
  }
  return check_launch("cheb_basis_fwd");
}

extern "C" int p2m_cheb_basis_bwd(p2m_graph_t gh, const float* d0, const float* d1, const float* d2,
                                  const float* resid, float* dX, int32_t B, int32_t F, int32_t out_shift,
                                  void* stream) {
  P2M_CHECK_ARG(gh && d0 && d1 && d2 && dX && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(out_shift == 0 || out_shift == 1, "out_shift must be 0 or 1");
  if (B <= 0) return P2M_OK;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(out_shift == 0 || (g.V % 2 == 0), "pair-sum needs an even vertex count");
  hipStream_t s = (hipStream_t)stream;
  const int Vout = g.V >> out_shift;
  const int tps = cdiv(Vout, ROWS_PER_BLOCK);
  auto grid = [&](int lpr) { return dim3(cdiv(B, 64 / lpr) * tps); };
  switch (F) {
    case 32:  hipLaunchKernelGGL(k_basis_bwd<8>,  grid(8),  dim3(256), 0, s, g, d0, d1, d2, resid, dX, B, out_shift, tps); break;
    case 64:  hipLaunchKernelGGL(k_basis_bwd<16>, grid(16), dim3(256), 0, s, g, d0, d1, d2, resid, dX, B, out_shift, tps); break;
    case 128: hipLaunchKernelGGL(k_basis_bwd<32>, grid(32), dim3(256), 0, s, g, d0, d1, d2, resid, dX, B, out_shift, tps); break;
    case 256: hipLaunchKernelGGL(k_basis_bwd<64>, grid(64), dim3(256), 0, s, g, d0, d1, d2, resid, dX, B, out_shift, tps); break;
    default: {
      long tot = (long)B * Vout * F;
      hipLaunchKernelGGL(k_basis_bwd_generic, dim3(cdiv(tot, 256)), dim3(256), 0, s, g, d0, d1, d2, resid, dX, B, F, out_shift);
    }
  }
  return check_launch("cheb_basis_bwd");
}
