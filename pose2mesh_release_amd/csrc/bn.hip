// BatchNorm1d-over-(B*V)-rows, ReLU, feature-axis residual resize and un-pool helpers (gfx950).
//
// Reference arithmetic: nn.BatchNorm1d inside graph_conv_cheby (lib/models/backbones/cheby_graph_conv.py:39,
// statistics over ALL B*V rows, fake vertices included), F.relu (lib/models/meshnet.py:100),
// F.interpolate(mode='linear') along the feature axis + residual add (meshnet.py:109-110,114-115),
// nn.Upsample(scale_factor=2) (meshnet.py:71-78; here virtual: consumers index r>>1).
// All of these are streaming HBM-bound passes: float4 per lane, one row group of F/4 lanes per row.
#include <map>
#include <mutex>
#include <utility>

#include "p2m_common.h"

namespace p2m {

// ---- statistics finalize --------------------------------------------------------------------
// stats[tile][0][n] = sum, stats[tile][1][n] = sum (y - tile_mean)^2.  sum y^2 over a tile is
// M2 + sum^2/n; tile-centred partials make the double-precision E[y^2]-E[y]^2 benign.
// Two stages.  Stage 1 (k_bn_partial): a block owns 32 adjacent columns (one 128-byte run per tile row: coalesced) and
// one of FIN_SPLITS slices of the tile range; 8 row groups per block, reduced through LDS; double-precision partials.
// Stage 2 (k_bn_finalize): one thread per column adds the FIN_SPLITS partials in a fixed order and writes the
// coefficients.  Deterministic (no atomics).  The single-stage form read 4-byte words at a 2N*4-byte stride from
// N/8 blocks only: ~35 us per call for the fine levels, 40 calls per step.
// (Round 5 built both stages as ONE launch - the stage-1 block that draws the last ticket of its column group runs stage 2:
//  __threadfence / device-scope atomic / __threadfence.  Parity-green, 40 launches fewer, and 1.4-1.7 ms per step SLOWER
//  (44.1 vs 45.7 ms, same box): at agent scope a fence is buffer_wbl2 sc1 + buffer_inv sc1, which writes back and drops the
//  XCD's L2 - under the side-stream contraction whose working set lives there.  Two launches it stays.)
constexpr int FIN_COLS = 32;    // columns per stage-1 block
constexpr int FIN_RG = 8;       // tile-row groups per stage-1 block
constexpr int FIN_SPLITS = 48;  // slices of the tile range: the minimum ...
constexpr int FIN_SPLITS_MAX = 192;   // ... and what a long tile range is cut into (round 4: the stage-1 launches of the fine
                                      // levels read 56 MB from 4 x 48 = 192 blocks: 150 us; stage 2 walked its partials with
                                      // one dependent load per iteration: 25-30 us under load)
static inline int fin_splits(long nrows) {            // a function of the sizes only: results stay deterministic
  return nrows >= 16384 ? FIN_SPLITS_MAX : (nrows >= 4096 ? 96 : FIN_SPLITS);
}
// sum of the nsplits stage-1 partials of column n (layout part[(split * 2 + which) * N + n]) in a FIXED order, with the loads
// of 12 splits in flight at a time (independent accumulators, combined in order)
// Stage 2 runs 64 columns x FIN_Q quarter-ranges of the splits per block: thread (q, col) sums its quarter, the quarters are
// combined through LDS in the order 0..3 - one round of 12 loads in flight per thread at 48 splits instead of four.
constexpr int FIN_Q = 4;
__device__ __forceinline__ void fin_sum(const double* __restrict__ part, int nsplits, int N, int n, bool live, double& t1,
                                        double& t2) {
  __shared__ double q1[FIN_Q][64], q2[FIN_Q][64];
  const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int per = nsplits / FIN_Q;                     // nsplits is 48, 96 or 192: per is a multiple of 12
  t1 = 0.0;
  t2 = 0.0;
  if (live)
  for (int sp0 = q * per; sp0 < (q + 1) * per; sp0 += 12) {
    double a[12], b[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      a[k] = part[((long)(sp0 + k) * 2) * N + n];
      b[k] = part[((long)(sp0 + k) * 2 + 1) * N + n];
    }
#pragma unroll
    for (int k = 0; k < 12; k++) {
      t1 += a[k];
      t2 += b[k];
    }
  }
  q1[q][col] = t1;
  q2[q][col] = t2;
  __syncthreads();
  t1 = q1[0][col];
  t2 = q2[0][col];
#pragma unroll
  for (int k = 1; k < FIN_Q; k++) {
    t1 += q1[k][col];
    t2 += q2[k][col];
  }
}
// Up to two segments of partials (e.g. the real-vertex and the fake-vertex launch of one conv); inside a segment the
// tiles repeat with period `tps` over `seg_rows` rows (row-set launches tile every sample separately).
struct StatSeg {
  const float* stats;
  int ntiles, tps;
  long seg_rows;
  const float* tile_w;   // optional [tps]: total weight of each tile (weighted partials of class representatives)
};
__global__ __launch_bounds__(FIN_COLS * FIN_RG) void k_bn_partial(StatSeg sg0, StatSeg sg1, int tile_rows, int N,
                                                                  double* __restrict__ part, int nsplits) {
  __shared__ double s1[FIN_RG][FIN_COLS];
  __shared__ double s2[FIN_RG][FIN_COLS];
  const int c = threadIdx.x % FIN_COLS, rg = threadIdx.x / FIN_COLS;
  const int n = blockIdx.x * FIN_COLS + c;
  const int split = blockIdx.y;
  double a1 = 0.0, a2 = 0.0;
  if (n < N) {
    for (int sgi = 0; sgi < 2; sgi++) {
      const StatSeg sg = sgi == 0 ? sg0 : sg1;
      if (sg.stats == nullptr) continue;
      const int per = (sg.ntiles + nsplits - 1) / nsplits;
      const int i0 = split * per;
      int i1 = i0 + per;
      if (i1 > sg.ntiles) i1 = sg.ntiles;
      for (int i = i0 + rg; i < i1; i += FIN_RG) {
        long left = sg.seg_rows - (long)(i % sg.tps) * tile_rows;
        const double cnt = sg.tile_w ? (double)sg.tile_w[i % sg.tps] : (double)(left < tile_rows ? left : tile_rows);
        const double sm = (double)sg.stats[(long)i * 2 * N + n];
        const double m2 = (double)sg.stats[(long)i * 2 * N + N + n];
        a1 += sm;
        a2 += m2 + sm * sm / cnt;
      }
    }
  }
  s1[rg][c] = a1;
  s2[rg][c] = a2;
  __syncthreads();
  if (rg == 0 && n < N) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < FIN_RG; q++) {
      t1 += s1[q][c];
      t2 += s2[q][c];
    }
    part[((long)split * 2) * N + n] = t1;
    part[((long)split * 2 + 1) * N + n] = t2;
  }
}

__global__ __launch_bounds__(64 * FIN_Q) void k_bn_finalize(const double* __restrict__ part, long M,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* running_mean, float* running_var, float momentum, float eps,
                                                    float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int N,
                                                    int nsplits) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  double t1, t2;
  fin_sum(part, nsplits, N, n, n < N, t1, t2);
  if (n >= N || threadIdx.x >= 64) return;
  const double mean = t1 / (double)M;
  double var = t2 / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  mean_o[n] = meanf;
  invstd_o[n] = invstd;
  const float sc = gamma[n] * invstd;
  scale_o[n] = sc;
  shift_o[n] = beta[n] - meanf * sc;
  if (running_mean != nullptr) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * meanf;
    running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
  }
}

__global__ void k_bn_eval_coeffs(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int N) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float invstd = 1.0f / sqrtf(rv[n] + eps);
  float sc = gamma[n] * invstd;
  mean_o[n] = rm[n];
  invstd_o[n] = invstd;
  scale_o[n] = sc;
  shift_o[n] = beta[n] - rm[n] * sc;
}

// Row map of the streaming passes on a level with classes (include/p2m.h): they walk the LIVE rows only -- logical row
// j = b * n + i  ->  actual row b * V + ids[i] -- so every lane always has a row to move (predicating the holes away
// instead left a third of the loads in flight empty: 4.3 instead of 5+ TB/s).  ids == nullptr: identity.
struct RowMap {
  const float* w;      // per-vertex weight (1 real vertex, class size for a representative, 0 hole)
  const int* ids;      // [n] live vertices of the level, ascending
  unsigned n, V;       // ids == nullptr && w != nullptr (the apply pass with zero_holes): EVERY row is walked, vertex = row % V,
};                     // a hole is not loaded and gets zeros stored - the caller needs no memset of its outputs
struct RowPos { unsigned b, i; };
__device__ __forceinline__ RowPos row_pos(long j, const RowMap& m) {
  RowPos p;
  p.b = (unsigned)(j / m.n);
  p.i = (unsigned)(j - (long)p.b * m.n);
  return p;
}
__device__ __forceinline__ RowPos row_adv(RowPos p, unsigned step, const RowMap& m) {   // the division only at a wrap
  p.i += step;
  if (p.i >= m.n) {
    const unsigned q = p.i / m.n;
    p.b += q;
    p.i -= q * m.n;
  }
  return p;
}

// ---- forward activation ---------------------------------------------------------------------
__device__ __forceinline__ float lerp_feat(const float* __restrict__ row, int Fres, int F, int j) {
  // F.interpolate(mode='linear', align_corners=False) along an axis of length Fres -> F
  float src = ((float)j + 0.5f) * ((float)Fres / (float)F) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  int i1 = i0 + 1 < Fres ? i0 + 1 : Fres - 1;
  float w = src - (float)i0;
  return row[i0] * (1.f - w) + row[i1] * w;
}

// One float4 column group per thread, ACT_UNROLL rows per thread with all loads issued before the arithmetic (a
// single 16-byte load per thread leaves the memory pipe half empty: 3.3 TB/s -> see DESIGN.md).  Needs 256 % (F/4) == 0.
constexpr int ACT_UNROLL = 4;     // (stand-alone 5.0 - 5.3 TB/s; 8 x 2, 8 x 4, 4 x 8, 2 x 8 measure the same)
constexpr int ACT_PASSES = 4;     // passes of ACT_UNROLL row groups per block: one amax commit (a wave reduction + a load of
                                  // the word) per 16 rows of a thread instead of per 4
__global__ __launch_bounds__(256) void k_bn_act_fwd(const float* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu,
                                                     const float* __restrict__ resid, int Fres, int res_shift,
                                                     float* __restrict__ x, long M, int F, RowMap m,
                                                     unsigned* __restrict__ amax) {
  const int F4 = F >> 2;
  const int rpb = 256 / F4;                                  // rows per block pass
  float vmax = 0.f;                                          // max |x stored| -> the tensor's amax word
  const int f = (threadIdx.x % F4) * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (scale != nullptr) {
    sc = *reinterpret_cast<const float4*>(scale + f);
    sh = *reinterpret_cast<const float4*>(shift + f);
  }
  for (int pass = 0; pass < ACT_PASSES; pass++) {
    const long rbase = ((long)blockIdx.x * ACT_PASSES + pass) * rpb * ACT_UNROLL + threadIdx.x / F4;   // logical rows (M of them)
    if ((long)(blockIdx.x * ACT_PASSES + pass) * rpb * ACT_UNROLL >= M) break;                          // block-uniform
    float4 v[ACT_UNROLL], q[ACT_UNROLL];
    long row[ACT_UNROLL];
    const bool same = resid != nullptr && Fres == F;
    const bool mapped = m.ids != nullptr;
    RowPos p0 = {0u, 0u};
    if (mapped && rbase < M) p0 = row_pos(rbase, m);
#pragma unroll
    for (int u = 0; u < ACT_UNROLL; u++) {
      const long j = rbase + (long)u * rpb;
      long r = j < M ? j : M - 1;                               // clamped: keeps the loads unconditional
      if (mapped) {
        const RowPos pu = j < M ? row_adv(p0, (unsigned)(u * rpb), m) : row_pos(M - 1, m);
        r = (long)pu.b * m.V + m.ids[pu.i];
      }
      row[u] = r;
      v[u] = *reinterpret_cast<const float4*>(y + r * F + f);
      if (same) q[u] = *reinterpret_cast<const float4*>(resid + (r >> res_shift) * Fres + f);
    }
#pragma unroll
    for (int u = 0; u < ACT_UNROLL; u++) {
      if (rbase + (long)u * rpb >= M) break;
      const long r = row[u];
      float4 o = v[u];
      if (scale != nullptr) {
        o.x = fmaf(o.x, sc.x, sh.x); o.y = fmaf(o.y, sc.y, sh.y);
        o.z = fmaf(o.z, sc.z, sh.z); o.w = fmaf(o.w, sc.w, sh.w);
      }
      if (relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      if (same) {
        o.x += q[u].x; o.y += q[u].y; o.z += q[u].z; o.w += q[u].w;
      } else if (resid != nullptr) {
        const float* rr = resid + (r >> res_shift) * Fres;
        o.x += lerp_feat(rr, Fres, F, f);
        o.y += lerp_feat(rr, Fres, F, f + 1);
        o.z += lerp_feat(rr, Fres, F, f + 2);
        o.w += lerp_feat(rr, Fres, F, f + 3);
      }
      *reinterpret_cast<float4*>(x + r * F + f) = o;
      vmax = amax4(vmax, &o.x);
    }
  }
  if (amax != nullptr) amax_commit(amax, vmax);
}

// one float4 per thread: any F % 4 == 0
__global__ __launch_bounds__(256) void k_bn_act_fwd_v4(const float* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        const float* __restrict__ resid, int Fres, int res_shift,
                                                        float* __restrict__ x, long M, int F,
                                                        unsigned* __restrict__ amax) {
  const int F4 = F >> 2;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = M * F4;
  const bool live = idx < tot;
  if (!live) idx = tot - 1;                                  // clamped, not returned: every lane reaches amax_commit
  long r = idx / F4;
  int f = (int)(idx - r * F4) * 4;
  float4 v = *reinterpret_cast<const float4*>(y + r * F + f);
  if (scale != nullptr) {
    float4 sc = *reinterpret_cast<const float4*>(scale + f);
    float4 sh = *reinterpret_cast<const float4*>(shift + f);
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
    v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
  }
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  if (resid != nullptr) {
    const float* rr = resid + (r >> res_shift) * Fres;
    if (Fres == F) {
      float4 q = *reinterpret_cast<const float4*>(rr + f);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    } else {
      v.x += lerp_feat(rr, Fres, F, f);
      v.y += lerp_feat(rr, Fres, F, f + 1);
      v.z += lerp_feat(rr, Fres, F, f + 2);
      v.w += lerp_feat(rr, Fres, F, f + 3);
    }
  }
  if (live) *reinterpret_cast<float4*>(x + r * F + f) = v;
  if (amax != nullptr) amax_commit(amax, live ? amax4(0.f, &v.x) : 0.f);
}

// scalar version for F % 4 != 0
__global__ void k_bn_act_fwd_generic(const float* __restrict__ y, const float* __restrict__ scale,
                                     const float* __restrict__ shift, int relu, const float* __restrict__ resid,
                                     int Fres, int res_shift, float* __restrict__ x, long M, int F) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * F) return;
  long r = idx / F;
  int f = (int)(idx - r * F);
  float v = y[idx];
  if (scale) v = fmaf(v, scale[f], shift[f]);
  if (relu) v = fmaxf(v, 0.f);
  if (resid) v += lerp_feat(resid + (r >> res_shift) * Fres, Fres, F, f);
  x[idx] = v;
}

// ---- weighted statistics of class representatives ---------------------------------------------
// One block per (sample, tile of 128 entries of the id list): st[tile][0][n] = sum w y, st[tile][1][n] = sum w (y - m)^2
// with m the tile's weighted mean -- the same tile-centred form the contraction epilogue emits, a row counting w times.
__global__ __launch_bounds__(256) void k_stats_rows_w(const float* __restrict__ y, const int* __restrict__ ids,
                                                       const float* __restrict__ wts, int n, int V, int tps, int N,
                                                       float* __restrict__ st) {
  __shared__ double red[8][32];
  const int b = blockIdx.x / tps, tile = blockIdx.x - b * tps;
  const int i0 = tile * 128;
  int i1 = i0 + 128;
  if (i1 > n) i1 = n;
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  for (int n0 = 0; n0 < N; n0 += 32) {
    const int col = n0 + c;
    double sw = 0.0, s1 = 0.0;
    if (col < N)
      for (int i = i0 + rg; i < i1; i += 8) {
        const double wv = (double)wts[i];
        sw += wv;
        s1 += wv * (double)y[((long)b * V + ids[i]) * N + col];
      }
    red[rg][c] = s1;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) tot += red[q][c];
    __syncthreads();
    red[rg][c] = sw;
    __syncthreads();
    double wtot = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) wtot += red[q][c];
    __syncthreads();
    const double mean = wtot > 0.0 ? tot / wtot : 0.0;
    double m2 = 0.0;
    if (col < N)
      for (int i = i0 + rg; i < i1; i += 8) {
        const double d = (double)y[((long)b * V + ids[i]) * N + col] - mean;
        m2 += (double)wts[i] * d * d;
      }
    red[rg][c] = m2;
    __syncthreads();
    if (rg == 0 && col < N) {
      double t2 = 0.0;
#pragma unroll
      for (int q = 0; q < 8; q++) t2 += red[q][c];
      st[(long)blockIdx.x * 2 * N + col] = (float)tot;
      st[(long)blockIdx.x * 2 * N + N + col] = (float)t2;
    }
    __syncthreads();
  }
}

// float4 form (N a multiple of 4, N/4 lanes per row dividing 256): whole rows per lane group, both passes over the tile's
// <= 128 rows (the second one from L2)
__global__ __launch_bounds__(256) void k_stats_rows_w4(const float* __restrict__ y, const int* __restrict__ ids,
                                                        const float* __restrict__ wts, int n, int V, int tps, int N,
                                                        float* __restrict__ st) {
  __shared__ float red[256 * 4];
  __shared__ float wred[256];
  const int LPR = N >> 2, RP = 256 / LPR;
  const int b = blockIdx.x / tps, tile = blockIdx.x - b * tps;
  const int i0 = tile * 128;
  int i1 = i0 + 128;
  if (i1 > n) i1 = n;
  const int t = threadIdx.x, rg = t / LPR, f = (t - rg * LPR) * 4;
  const float* yb = y + (long)b * V * N;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float sw = 0.f;
  for (int i = i0 + rg; i < i1; i += RP) {
    const float wv = wts[i];
    const float4 v = *reinterpret_cast<const float4*>(yb + (long)ids[i] * N + f);
    s.x = fmaf(wv, v.x, s.x); s.y = fmaf(wv, v.y, s.y); s.z = fmaf(wv, v.z, s.z); s.w = fmaf(wv, v.w, s.w);
    sw += wv;
  }
  *reinterpret_cast<float4*>(&red[t * 4]) = s;
  wred[t] = sw;
  __syncthreads();
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
  float wtot = 0.f;
  for (int q = 0; q < RP; q++) {
    const float4 p = *reinterpret_cast<const float4*>(&red[(q * LPR + (t - rg * LPR)) * 4]);
    tot.x += p.x; tot.y += p.y; tot.z += p.z; tot.w += p.w;
    wtot += wred[q * LPR];
  }
  __syncthreads();
  const float inv = wtot > 0.f ? 1.f / wtot : 0.f;
  const float4 mean = make_float4(tot.x * inv, tot.y * inv, tot.z * inv, tot.w * inv);
  float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = i0 + rg; i < i1; i += RP) {
    const float wv = wts[i];
    const float4 v = *reinterpret_cast<const float4*>(yb + (long)ids[i] * N + f);
    float d;
    d = v.x - mean.x; m2.x = fmaf(wv * d, d, m2.x);
    d = v.y - mean.y; m2.y = fmaf(wv * d, d, m2.y);
    d = v.z - mean.z; m2.z = fmaf(wv * d, d, m2.z);
    d = v.w - mean.w; m2.w = fmaf(wv * d, d, m2.w);
  }
  *reinterpret_cast<float4*>(&red[t * 4]) = m2;
  __syncthreads();
  if (rg == 0) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < RP; q++) {
      const float4 p = *reinterpret_cast<const float4*>(&red[(q * LPR + t) * 4]);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    *reinterpret_cast<float4*>(st + (long)blockIdx.x * 2 * N + f) = tot;
    *reinterpret_cast<float4*>(st + (long)blockIdx.x * 2 * N + N + f) = a;
  }
}

// out[r] = sum of in over the class of r (representatives: w[v] consecutive rows starting at r), in[r] for real
// vertices, 0 for holes: the gradient a representative carries is the SUM over its class.
__global__ __launch_bounds__(256) void k_class_reduce(const float* __restrict__ in, const float* __restrict__ w,
                                                       float* __restrict__ out, long M, int V, int F) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * F) return;
  const long r = idx / F;
  const int f = (int)(idx - r * F);
  const int v = (int)(r % V);
  const int m = (int)w[v];
  float s = 0.f;
  for (int k = 0; k < m; k++) s += in[(r + k) * F + f];
  out[idx] = s;
}

// ---- backward ---------------------------------------------------------------------------------
// rows per block of the reduction: 512 for the big levels; fewer when that would leave the 256 CUs with a handful of
// blocks each (F = 256 at V <= 2944: 580-1100 blocks of 512 rows ran at 2.4 TB/s)
constexpr int BWD_ROWS_PER_BLOCK = 512;
static inline int bwd_rows_per_block(long M) {
  if (M >= (long)BWD_ROWS_PER_BLOCK * 4096) return BWD_ROWS_PER_BLOCK;
  long r = (M + 4095) / 4096;
  r = (r + 63) / 64 * 64;
  return (int)(r < 64 ? 64 : r);
}

template <int LPR>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ gx, const float* __restrict__ y,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int relu, float* __restrict__ part, long M, RowMap m,
                                                        int rows_per_block) {
  constexpr int F = LPR * 4;
  constexpr int RP = 256 / LPR;
  __shared__ float red[2][RP][F];
  const int t = threadIdx.x;
  const int rloc = t / LPR, f = (t % LPR) * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + f);
  const float4 sh = *reinterpret_cast<const float4*>(shift + f);
  const float4 mu = *reinterpret_cast<const float4*>(mean + f);
  const float4 is = *reinterpret_cast<const float4*>(invstd + f);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  const bool mapped = m.ids != nullptr;                     // M, r0, r1 count LOGICAL rows (the live ones) then
  RowPos pb = {0u, 0u};
  if (mapped && r0 + rloc < r1) pb = row_pos(r0 + rloc, m);
  for (long rb = r0 + rloc; rb < r1; rb += 4 * RP) {        // 4 rows per pass: 8 loads in flight per thread
    float4 gq[4], vq[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long j = rb + (long)u * RP;
      long r = j < r1 ? j : r1 - 1;
      if (mapped) {
        const RowPos pu = j < r1 ? row_adv(pb, (unsigned)(u * RP), m) : row_pos(r1 - 1, m);
        r = (long)pu.b * m.V + m.ids[pu.i];
      }
      gq[u] = *reinterpret_cast<const float4*>(gx + r * F + f);
      vq[u] = *reinterpret_cast<const float4*>(y + r * F + f);
    }
    if (mapped) pb = row_adv(pb, 4u * RP, m);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (rb + (long)u * RP >= r1) break;
      float4 g = gq[u];
      const float4 v = vq[u];
      if (relu) {
        if (fmaf(v.x, sc.x, sh.x) <= 0.f) g.x = 0.f;
        if (fmaf(v.y, sc.y, sh.y) <= 0.f) g.y = 0.f;
        if (fmaf(v.z, sc.z, sh.z) <= 0.f) g.z = 0.f;
        if (fmaf(v.w, sc.w, sh.w) <= 0.f) g.w = 0.f;
      }
      s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
      s1.x = fmaf(g.x, (v.x - mu.x) * is.x, s1.x);
      s1.y = fmaf(g.y, (v.y - mu.y) * is.y, s1.y);
      s1.z = fmaf(g.z, (v.z - mu.z) * is.z, s1.z);
      s1.w = fmaf(g.w, (v.w - mu.w) * is.w, s1.w);
    }
  }
  *reinterpret_cast<float4*>(&red[0][rloc][f]) = s0;
  *reinterpret_cast<float4*>(&red[1][rloc][f]) = s1;
  __syncthreads();
  for (int o = t; o < 2 * F; o += 256) {
    const int which = o / F, c = o - which * F;
    float s = 0.f;
#pragma unroll 4
    for (int q = 0; q < RP; q++) s += red[which][q][c];
    part[(long)blockIdx.x * 2 * F + o] = s;
  }
}

// any F: one block per BWD_ROWS_PER_BLOCK rows, threads stride over features (narrow / odd widths, e.g. Fout = 3 or 5)
__global__ __launch_bounds__(256) void k_bn_bwd_reduce_generic(const float* __restrict__ gx, const float* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu,
                                                                float* __restrict__ part, long M, int F, RowMap m,
                                                                int rows_per_block) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  for (int f = threadIdx.x; f < F; f += 256) {
    const float sc = scale[f], sh = shift[f], mu = mean[f], is = invstd[f];
    float s0 = 0.f, s1 = 0.f;
    for (long j = r0; j < r1; j++) {
      long r = j;
      if (m.ids != nullptr) {
        const RowPos pj = row_pos(j, m);
        r = (long)pj.b * m.V + m.ids[pj.i];
      }
      const float v = y[r * F + f];
      float g = gx[r * F + f];
      if (relu && fmaf(v, sc, sh) <= 0.f) g = 0.f;
      s0 += g;
      s1 = fmaf(g, (v - mu) * is, s1);
    }
    part[(long)blockIdx.x * 2 * F + f] = s0;
    part[(long)blockIdx.x * 2 * F + F + f] = s1;
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply_generic(const float* __restrict__ gx, const float* __restrict__ y,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ coef, int relu,
                                                               float* __restrict__ gy, long M, int F, RowMap m,
                                                               unsigned* __restrict__ amax) {
  long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < M * F;
  if (!live) idx = M * F - 1;                                // clamped, not returned: every lane reaches amax_commit
  const int f = (int)(idx % F);
  float wr = 1.f;
  if (m.ids != nullptr) {
    const RowPos pj = row_pos(idx / F, m);
    const unsigned vtx = (unsigned)m.ids[pj.i];
    wr = m.w[vtx];
    idx = ((long)pj.b * m.V + vtx) * F + f;
  } else if (m.w != nullptr) {                               // zero_holes: every row, a hole gets 0 (RowMap)
    wr = m.w[(unsigned)((idx / F) % (long)m.V)];
  }
  const float v = wr != 0.f ? y[idx] : 0.f;
  float go = wr != 0.f ? gx[idx] : 0.f;
  if (relu && fmaf(v, scale[f], shift[f]) <= 0.f) go = 0.f;
  const float k = gamma[f] * invstd[f];
  const float c0 = coef ? coef[f] : 0.f, c1 = coef ? coef[F + f] : 0.f;
  const float o = wr != 0.f ? fmaf(k, go, wr * fmaf(-k * c1 * invstd[f], v - mean[f], -k * c0)) : 0.f;
  if (live) gy[idx] = o;
  if (amax != nullptr) amax_commit(amax, live ? amax_abs(o) : 0.f);
}

// Two stages like the forward finalize.  Stage 1: block = 32 adjacent columns of BOTH partial kinds (128-byte coalesced
// reads) x 8 row groups, one of FIN_SPLITS slices of the nblk partial rows; stage 2: one thread per column.
__global__ __launch_bounds__(FIN_COLS * FIN_RG) void k_bn_bwd_partial(const float* __restrict__ part, int nblk, int F,
                                                                      double* __restrict__ out, int nsplits) {
  __shared__ double s0[FIN_RG][FIN_COLS];
  __shared__ double s1[FIN_RG][FIN_COLS];
  const int cc = threadIdx.x % FIN_COLS, rg = threadIdx.x / FIN_COLS;
  const int c = blockIdx.x * FIN_COLS + cc;
  const int split = blockIdx.y;
  const int per = (nblk + nsplits - 1) / nsplits;
  const int i0 = split * per;
  int i1 = i0 + per;
  if (i1 > nblk) i1 = nblk;
  double a0 = 0.0, a1 = 0.0;
  if (c < F) {
    for (int i = i0 + rg; i < i1; i += FIN_RG) {
      a0 += (double)part[(long)i * 2 * F + c];
      a1 += (double)part[(long)i * 2 * F + F + c];
    }
  }
  s0[rg][cc] = a0;
  s1[rg][cc] = a1;
  __syncthreads();
  if (rg == 0 && c < F) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int q = 0; q < FIN_RG; q++) {
      t0 += s0[q][cc];
      t1 += s1[q][cc];
    }
    out[((long)split * 2) * F + c] = t0;
    out[((long)split * 2 + 1) * F + c] = t1;
  }
}

__global__ __launch_bounds__(64 * FIN_Q) void k_bn_bwd_finalize(const double* __restrict__ part, long M, float* dgamma,
                                                        float* dbeta, float* coef, int accumulate, int F, int nsplits) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double t0, t1;
  fin_sum(part, nsplits, F, c, c < F, t0, t1);
  if (c >= F || threadIdx.x >= 64) return;
  const float db = (float)t0, dg = (float)t1;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + db : db;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  if (coef) {
    coef[c] = (float)(t0 / (double)M);
    coef[F + c] = (float)(t1 / (double)M);
  }
}

// gy = gamma*invstd*(go - c0 - yhat*c1): each thread owns ONE float4 column group (its coefficients live in
// registers) and walks rows, so the kernel is three streaming float4 accesses per element.
constexpr int APPLY_ROWS_PER_BLOCK = 256;
template <int LPR>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ gx, const float* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ coef,
                                                       int relu, float* __restrict__ gy, long M, RowMap m,
                                                       unsigned* __restrict__ amax) {
  constexpr int F = LPR * 4;
  constexpr int RP = 256 / LPR;
  const int t = threadIdx.x;
  const int rloc = t / LPR, f = (t % LPR) * 4;
  float vmax = 0.f;                                          // max |gy stored| -> the tensor's amax word
  float sc[4], sh[4], k[4], a0[4], a1[4];
  *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + f);
  *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + f);
  float mu[4], is[4], ga[4];
  *reinterpret_cast<float4*>(mu) = *reinterpret_cast<const float4*>(mean + f);
  *reinterpret_cast<float4*>(is) = *reinterpret_cast<const float4*>(invstd + f);
  *reinterpret_cast<float4*>(ga) = *reinterpret_cast<const float4*>(gamma + f);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    k[i] = ga[i] * is[i];
    // gy = k*go - k*c0 - k*c1*yhat,  yhat = (v - mu)*is   ->   gy = k*go + a0 + a1*(v - mu)
    const float c0 = coef ? coef[f + i] : 0.f, c1 = coef ? coef[F + f + i] : 0.f;
    a1[i] = -k[i] * c1 * is[i];
    a0[i] = -k[i] * c0;
  }
  const long r0 = (long)blockIdx.x * APPLY_ROWS_PER_BLOCK;
  long r1 = r0 + APPLY_ROWS_PER_BLOCK;
  if (r1 > M) r1 = M;
  const bool mapped = m.ids != nullptr;                     // M, r0, r1 count LOGICAL rows (the live ones) then
  const bool dense_w = !mapped && m.w != nullptr;           // every row, holes zeroed (RowMap)
  const bool weighted = mapped || dense_w;
  RowPos pb = {0u, 0u};
  if (mapped && r0 + rloc < r1) pb = row_pos(r0 + rloc, m);
  for (long rb = r0 + rloc; rb < r1; rb += 4 * RP) {        // 4 rows per pass: 8 loads in flight per thread
    float g[4][4], v[4][4], wq[4];
    long row[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long j = rb + (long)u * RP;
      long r = j < r1 ? j : r1 - 1;
      wq[u] = 1.f;
      if (mapped) {
        const RowPos pu = j < r1 ? row_adv(pb, (unsigned)(u * RP), m) : row_pos(r1 - 1, m);
        const unsigned vtx = (unsigned)m.ids[pu.i];
        wq[u] = m.w[vtx];
        r = (long)pu.b * m.V + vtx;
      } else if (dense_w) {
        wq[u] = m.w[(unsigned)(r % (long)m.V)];
      }
      row[u] = r;
      // (dense_w: a hole's row is loaded like any other - the memory is there, whatever it holds is dropped by the select
      //  below - so the eight loads of a pass stay unconditional)
      *reinterpret_cast<float4*>(g[u]) = *reinterpret_cast<const float4*>(gx + r * F + f);
      *reinterpret_cast<float4*>(v[u]) = *reinterpret_cast<const float4*>(y + r * F + f);
    }
    if (mapped) pb = row_adv(pb, 4u * RP, m);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (rb + (long)u * RP >= r1) break;
      // classes: the constant term enters once per class member (a representative carries the class sum)
      const float wr = wq[u];
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float go = g[u][i];
        if (relu && fmaf(v[u][i], sc[i], sh[i]) <= 0.f) go = 0.f;
        o[i] = weighted ? fmaf(k[i], go, wr * fmaf(a1[i], v[u][i] - mu[i], a0[i]))
                        : fmaf(k[i], go, fmaf(a1[i], v[u][i] - mu[i], a0[i]));
        if (wr == 0.f) o[i] = 0.f;                          // a hole (dense_w walk only): zeros, whatever a0 is
      }
      *reinterpret_cast<float4*>(gy + row[u] * F + f) = *reinterpret_cast<float4*>(o);
      vmax = amax4(vmax, o);
    }
  }
  if (amax != nullptr) amax_commit(amax, vmax);
}

// The same pass with each thread owning PAIRS of adjacent rows (2q, 2q+1), so that the pair-sums the backward of an
// un-pooled conv needs come out as by-products instead of separate passes: pair_gx[q] = gx[2q] + gx[2q+1] (the residual
// gradient handed to the coarser level) and / or pair_gy[q] = gy[2q] + gy[2q+1] (plane S g of the paired operator).
template <int LPR>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_pairs(const float* __restrict__ gx, const float* __restrict__ y,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ coef,
                                                             int relu, float* __restrict__ gy, float* __restrict__ pair_gx,
                                                             float* __restrict__ pair_gy, long Mp, RowMap m,
                                                             unsigned* __restrict__ amax) {
  constexpr int F = LPR * 4;
  constexpr int RP = 256 / LPR;
  const int t = threadIdx.x;
  const int rloc = t / LPR, f = (t % LPR) * 4;
  float vmax = 0.f;                                          // max |gy stored| (the pair sums are <= twice that)
  float sc[4], sh[4], k[4], a0[4], a1[4], mu[4], is[4], ga[4];
  *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + f);
  *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + f);
  *reinterpret_cast<float4*>(mu) = *reinterpret_cast<const float4*>(mean + f);
  *reinterpret_cast<float4*>(is) = *reinterpret_cast<const float4*>(invstd + f);
  *reinterpret_cast<float4*>(ga) = *reinterpret_cast<const float4*>(gamma + f);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    k[i] = ga[i] * is[i];
    const float c0 = coef ? coef[f + i] : 0.f, c1 = coef ? coef[F + f + i] : 0.f;
    a1[i] = -k[i] * c1 * is[i];
    a0[i] = -k[i] * c0;
  }
  const long p0 = (long)blockIdx.x * (APPLY_ROWS_PER_BLOCK / 2);
  long p1 = p0 + APPLY_ROWS_PER_BLOCK / 2;
  if (p1 > Mp) p1 = Mp;
  // with classes, m maps LOGICAL pairs (Mp of them: the coarse vertices with a live child, m.ids over m.V = V/2) to
  // actual pairs; m.w is the FINE level's weight table (children 2c, 2c+1): a hole child is neither loaded nor stored
  const bool mapped = m.ids != nullptr;
  const bool dense_w = !mapped && m.w != nullptr;           // every pair, holes and hole parents zeroed (RowMap; m.V = V/2)
  const bool weighted = mapped || dense_w;
  RowPos pp = {0u, 0u};
  if (mapped && p0 + rloc < p1) pp = row_pos(p0 + rloc, m);
  for (long pb = p0 + rloc; pb < p1; pb += 2 * RP) {        // 2 pairs = 4 rows per pass: 8 loads in flight per thread
    float g[4][4], v[4][4], wq[4];
    long pair[2];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long jq = pb + (long)(u >> 1) * RP;
      long q = jq < p1 ? jq : p1 - 1;
      wq[u] = 1.f;
      if (mapped) {
        const RowPos pu = jq < p1 ? row_adv(pp, (unsigned)((u >> 1) * RP), m) : row_pos(p1 - 1, m);
        const unsigned c = (unsigned)m.ids[pu.i];
        wq[u] = m.w[2 * c + (u & 1)];
        q = (long)pu.b * m.V + c;
      } else if (dense_w) {
        wq[u] = m.w[2u * (unsigned)(q % (long)m.V) + (u & 1)];
      }
      pair[u >> 1] = q;
      const long r = 2 * q + (u & 1);
#pragma unroll
      for (int i = 0; i < 4; i++) g[u][i] = v[u][i] = 0.f;
      if (wq[u] != 0.f) {
        *reinterpret_cast<float4*>(g[u]) = *reinterpret_cast<const float4*>(gx + r * F + f);
        *reinterpret_cast<float4*>(v[u]) = *reinterpret_cast<const float4*>(y + r * F + f);
      }
    }
    if (mapped) pp = row_adv(pp, 2u * RP, m);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (pb + (long)h * RP >= p1) break;
      const long q = pair[h];
      float o[2][4];
      const float wr[2] = {wq[2 * h], wq[2 * h + 1]};
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int u = 2 * h + c;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float go = g[u][i];
          if (relu && fmaf(v[u][i], sc[i], sh[i]) <= 0.f) go = 0.f;
          o[c][i] = weighted ? fmaf(k[i], go, wr[c] * fmaf(a1[i], v[u][i] - mu[i], a0[i]))
                             : fmaf(k[i], go, fmaf(a1[i], v[u][i] - mu[i], a0[i]));
          if (wr[c] == 0.f) o[c][i] = 0.f;                // a hole: no data (its g / v registers were zeroed above)
        }
        if (wr[c] != 0.f || dense_w) {
          *reinterpret_cast<float4*>(gy + (2 * q + c) * F + f) = *reinterpret_cast<float4*>(o[c]);
          vmax = amax4(vmax, o[c]);
        }
      }
      // the pair-sums leave the holes out; a pair of two holes (a hole parent) is not written at all
      if (wr[0] != 0.f || wr[1] != 0.f || dense_w) {
        if (pair_gx) {
          float sx[4];
#pragma unroll
          for (int i = 0; i < 4; i++) sx[i] = g[2 * h][i] + g[2 * h + 1][i];
          *reinterpret_cast<float4*>(pair_gx + q * F + f) = *reinterpret_cast<float4*>(sx);
        }
        if (pair_gy) {
          float sy[4];
#pragma unroll
          for (int i = 0; i < 4; i++) sy[i] = o[0][i] + o[1][i];
          *reinterpret_cast<float4*>(pair_gy + q * F + f) = *reinterpret_cast<float4*>(sy);
        }
      }
    }
  }
  if (amax != nullptr) amax_commit(amax, vmax);
}

__global__ __launch_bounds__(256) void k_pair_sum(const float* __restrict__ in, float* __restrict__ out, long Mout, int F,
                                                   const float* __restrict__ w, int V) {
  const int F4 = F >> 2;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Mout * F4) return;
  long p = idx / F4;
  int f = (int)(idx - p * F4) * 4;
  float4 a = *reinterpret_cast<const float4*>(in + (2 * p) * F + f);
  float4 b = *reinterpret_cast<const float4*>(in + (2 * p + 1) * F + f);
  if (w != nullptr) {                                 // classes: holes hold no data
    const unsigned v0 = (unsigned)(2 * p) % (unsigned)V;
    if (w[v0] == 0.f) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w[v0 + 1] == 0.f) b = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  *reinterpret_cast<float4*>(out + p * F + f) = a;
}

// dst[r][i] += sum_j w(j, i) * g[r][j]  -- deterministic gather form of the resize transpose
__global__ void k_lerp_bwd_add(const float* __restrict__ g, float* __restrict__ dst, long M, int F, int Fres) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Fres) return;
  long r = idx / Fres;
  int i = (int)(idx - r * Fres);
  const float* gr = g + r * F;
  float ratio = (float)Fres / (float)F;
  // candidate outputs j whose i0 or i1 can be i:  src(j) in (i-1, i+1)
  int jlo = (int)floorf(((float)i - 1.f + 0.5f) / ratio - 0.5f) - 1;
  int jhi = (int)ceilf(((float)i + 1.f + 0.5f) / ratio - 0.5f) + 1;
  if (jlo < 0) jlo = 0;
  if (jhi > F - 1) jhi = F - 1;
  float acc = 0.f;
  for (int j = jlo; j <= jhi; j++) {
    float src = ((float)j + 0.5f) * ratio - 0.5f;
    if (src < 0.f) src = 0.f;
    int i0 = (int)src;
    int i1 = i0 + 1 < Fres ? i0 + 1 : Fres - 1;
    float w = src - (float)i0;
    if (i0 == i) acc = fmaf(1.f - w, gr[j], acc);
    if (i1 == i) acc = fmaf(w, gr[j], acc);
  }
  dst[idx] += acc;
}

// Fres == 2 F (the 256 -> 128 block): the resize is the mean of channel pairs, src(j) = 2j + 0.5 exactly, so
// dst[i] += 0.5 * g[i >> 1] - the same single fmaf per element as the generic gather, vectorised
__global__ __launch_bounds__(256) void k_lerp_bwd_add_half(const float* __restrict__ g, float* __restrict__ dst,
                                                            long M, int F) {
  const int Q = F >> 1;                                     // float4 groups of dst per row = (2F)/4
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Q) return;
  const long r = idx / Q;
  const int q = (int)(idx - r * Q);
  const float2 gv = *reinterpret_cast<const float2*>(g + r * F + 2 * q);
  float4* d = reinterpret_cast<float4*>(dst + r * 2 * F + 4 * q);
  float4 v = *d;
  v.x += fmaf(0.5f, gv.x, 0.f); v.y += fmaf(0.5f, gv.x, 0.f);
  v.z += fmaf(0.5f, gv.y, 0.f); v.w += fmaf(0.5f, gv.y, 0.f);
  *d = v;
}

}  // namespace p2m

using namespace p2m;

// classes of a level (optional handle): the weight table and the vertex count the rows repeat with
static bool class_table(p2m_graph_t gh, int64_t M, const float** w, int* V) {
  *w = nullptr;
  *V = 1;
  if (gh == nullptr) return true;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (g.w == nullptr) return true;
  if (M % g.V != 0 || M >= (1LL << 32)) return false;
  *w = g.w;
  *V = g.V;
  return true;
}


// row map of a streaming pass (RowMap above): with classes the LIVE rows (or, pairs: the coarse vertices with a live
// child) of the level, else the identity; *Mlog = the logical rows the launch covers
static bool row_map_of(p2m_graph_t gh, int64_t M, bool pairs, RowMap* m, long* Mlog) {
  m->w = nullptr; m->ids = nullptr; m->n = 1u; m->V = 1u;
  *Mlog = pairs ? M / 2 : M;
  if (gh == nullptr) return true;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (g.w == nullptr) return true;
  if (M % g.V != 0 || M >= (1LL << 32) || (pairs && (g.V & 1))) return false;
  const long B = M / g.V;
  m->w = g.w;
  if (pairs) { m->ids = g.live_pairs; m->n = (unsigned)g.n_live_pairs; m->V = (unsigned)(g.V / 2); }
  else { m->ids = g.live_ids; m->n = (unsigned)g.n_live; m->V = (unsigned)g.V; }
  *Mlog = B * (long)m->n;
  return m->n > 0;
}

// scratch of the two-stage finalize kernels: FIN_SPLITS x 2 x N doubles, one buffer per (device, stream) - the calls of
// one stream are ordered, so stage 1 of the next call cannot overtake stage 2 of the previous one.  Internal to the
// library (never visible to the caller), allocated on first use, kept for the life of the process.
namespace {
constexpr int FIN_MAXN = 4096;
std::map<std::pair<int, void*>, double*> g_fin;
std::mutex g_fin_mu;
double* fin_scratch(int N, void* stream) {
  if (N > FIN_MAXN) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_fin_mu);
  double*& buf = g_fin[{dev, stream}];
  if (buf == nullptr && hipMalloc((void**)&buf, sizeof(double) * 2 * FIN_SPLITS_MAX * FIN_MAXN) != hipSuccess) buf = nullptr;
  return buf;
}
int finalize_launch(const StatSeg& a, const StatSeg& b, long M, int tile_rows, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                    float* scale, float* shift, int N, hipStream_t s, const char* what) {
  double* part = fin_scratch(N, (void*)s);
  if (part == nullptr) {
    set_error("%s: no scratch for %d columns (max %d) or hipMalloc failed", what, N, FIN_MAXN);
    return P2M_ERR_NOMEM;
  }
  const int nsplits = fin_splits((long)a.ntiles + (b.stats != nullptr ? b.ntiles : 0));
  hipLaunchKernelGGL(k_bn_partial, dim3(cdiv(N, FIN_COLS), nsplits), dim3(FIN_COLS * FIN_RG), 0, s, a, b, tile_rows, N,
                     part, nsplits);
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(N, 64)), dim3(64 * FIN_Q), 0, s, part, M, gamma, beta, running_mean, running_var,
                     momentum, eps, mean, invstd, scale, shift, N, nsplits);
  return check_launch(what);
}
}  // namespace

extern "C" int p2m_bn_finalize(const float* stats, int32_t ntiles, int64_t M, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* mean,
                               float* invstd, float* scale, float* shift, int32_t N, int32_t tile_rows, void* stream) {
  P2M_CHECK_ARG(stats && gamma && beta && mean && invstd && scale && shift && N > 0 && M > 0, "null pointer or empty shape");
  P2M_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
  P2M_CHECK_ARG(tile_rows > 0 && ntiles == cdiv(M, tile_rows), "ntiles does not match M / tile_rows");
  StatSeg a{stats, ntiles, ntiles, (long)M, nullptr}, b{nullptr, 0, 1, 0, nullptr};
  return finalize_launch(a, b, (long)M, tile_rows, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd,
                         scale, shift, N, (hipStream_t)stream, "bn_finalize");
}

static int finalize_rows(const float* stats_a, int32_t tps_a, int32_t rows_a, const float* tile_w_a,
                         const float* stats_b, int32_t tps_b, int32_t rows_b, const float* tile_w_b, int32_t B,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, float* mean, float* invstd, float* scale,
                         float* shift, int32_t N, void* stream) {
  P2M_CHECK_ARG(stats_a && gamma && beta && mean && invstd && scale && shift && N > 0 && B > 0, "null pointer or empty shape");
  P2M_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
  const int tile_rows = p2m_stats_tile_rows();
  StatSeg a{stats_a, B * tps_a, tps_a, (long)rows_a, tile_w_a};
  StatSeg b{stats_b, stats_b ? B * tps_b : 0, tps_b > 0 ? tps_b : 1, (long)rows_b, stats_b ? tile_w_b : nullptr};
  const long M = (long)B * ((long)rows_a + (stats_b ? (long)rows_b : 0));
  return finalize_launch(a, b, M, tile_rows, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale,
                         shift, N, (hipStream_t)stream, "bn_finalize_rows");
}

extern "C" int p2m_bn_finalize_rows(const float* stats_a, int32_t tps_a, int32_t rows_a, const float* stats_b,
                                    int32_t tps_b, int32_t rows_b, int32_t B, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                    float* invstd, float* scale, float* shift, int32_t N, void* stream) {
  return finalize_rows(stats_a, tps_a, rows_a, nullptr, stats_b, tps_b, rows_b, nullptr, B, gamma, beta, running_mean,
                       running_var, momentum, eps, mean, invstd, scale, shift, N, stream);
}

// the two launches of one split conv, sizes and (with classes) the tile weights of the representatives from the handle
extern "C" int p2m_bn_finalize_split(p2m_graph_t gh, const float* stats_real, const float* stats_fake, int32_t B,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, float* mean, float* invstd, float* scale,
                                     float* shift, int32_t N, void* stream) {
  P2M_CHECK_ARG(gh, "null graph");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  const int tile_rows = p2m_stats_tile_rows();
  const int rows_fake = g.w ? g.n_fake_all : g.n_fake;      // with classes: the statistics count every fake vertex
  return finalize_rows(stats_real, cdiv(g.n_real, tile_rows), g.n_real, nullptr, g.n_fake > 0 ? stats_fake : nullptr,
                       cdiv(g.n_fake, tile_rows), rows_fake, g.w ? g.fake_tile_w : nullptr, B, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift, N, stream);
}

// the same when the real-vertex partials come from p2m_cheb_tile_gemm: one (sum, M2) pair per (sample, tile) of the given
// tile plan of the level (0: level input, 1: un-pooled input), weight = the tile's row count
extern "C" int p2m_bn_finalize_tiles(p2m_graph_t gh, int32_t plan, const float* stats_real, const float* stats_fake,
                                     int32_t B, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float momentum, float eps, float* mean, float* invstd,
                                     float* scale, float* shift, int32_t N, void* stream) {
  P2M_CHECK_ARG(gh, "null graph");
  P2M_CHECK_ARG(plan == 0 || plan == 1, "plan must be 0 or 1");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(g.plan[plan].ntiles > 0, "this level has no such tile plan");
  const int tile_rows = p2m_stats_tile_rows();
  const int rows_fake = g.w ? g.n_fake_all : g.n_fake;
  return finalize_rows(stats_real, g.plan[plan].ntiles, g.n_real, g.plan[plan].tile_cnt,
                       g.n_fake > 0 ? stats_fake : nullptr, cdiv(g.n_fake, tile_rows), rows_fake,
                       g.w ? g.fake_tile_w : nullptr, B, gamma, beta, running_mean, running_var, momentum, eps, mean,
                       invstd, scale, shift, N, stream);
}

extern "C" int p2m_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* mean, float* invstd, float* scale,
                                  float* shift, int32_t N, void* stream) {
  P2M_CHECK_ARG(gamma && beta && running_mean && running_var && mean && invstd && scale && shift && N > 0,
                "null pointer or empty shape");
  hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, mean, invstd, scale, shift, N);
  return check_launch("bn_eval_coeffs");
}

extern "C" int p2m_bn_act_fwd(const float* y, const float* scale, const float* shift, int32_t relu,
                              const float* resid, int32_t Fres, int32_t res_shift, float* x, int64_t M, int32_t F,
                              p2m_graph_t classes, int32_t real_rows_only, void* amax_out, void* stream) {
  P2M_CHECK_ARG(y && x && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(!real_rows_only || classes != nullptr, "real_rows_only needs the level's graph handle");
  unsigned* amax = static_cast<unsigned*>(amax_out);
  P2M_CHECK_ARG((scale == nullptr) == (shift == nullptr), "scale/shift must both be given or both NULL");
  P2M_CHECK_ARG(resid == nullptr || Fres > 0, "Fres must be positive with a residual");
  P2M_CHECK_ARG(res_shift == 0 || res_shift == 1, "res_shift must be 0 or 1");
  if (M <= 0) return P2M_OK;
  hipStream_t s = (hipStream_t)stream;
  if (F % 4 == 0 && (resid == nullptr || Fres != F || Fres % 4 == 0)) {
    const int F4 = F / 4;
    if (F4 <= 256 && 256 % F4 == 0) {
      const long rows_per_block = (long)(256 / F4) * ACT_UNROLL * ACT_PASSES;
      RowMap m;
      long Mlog;
      if (real_rows_only) {               // inference on the real rows: the other rows of y hold no data
        const Graph& g = *reinterpret_cast<const Graph*>(classes);
        P2M_CHECK_ARG(M % g.V == 0 && M < (1LL << 32) && g.n_real > 0, "M is not a multiple of the level's vertex count");
        m.w = nullptr; m.ids = g.real_ids; m.n = (unsigned)g.n_real; m.V = (unsigned)g.V;
        Mlog = (M / g.V) * (long)g.n_real;
      } else {
        P2M_CHECK_ARG(row_map_of(classes, M, false, &m, &Mlog), "M is not a multiple of the level's vertex count (or too large)");
      }
      hipLaunchKernelGGL(k_bn_act_fwd, dim3(cdiv(Mlog, rows_per_block)), dim3(256), 0, s, y, scale, shift, relu, resid,
                         Fres, res_shift, x, Mlog, F, m, amax);
    } else {
      P2M_CHECK_ARG((classes == nullptr || amax == nullptr) && !real_rows_only, "row maps need 256 % (F / 4) == 0");
      long tot = M * F4;
      hipLaunchKernelGGL(k_bn_act_fwd_v4, dim3(cdiv(tot, 256)), dim3(256), 0, s, y, scale, shift, relu, resid, Fres,
                         res_shift, x, (long)M, F, amax);
    }
  } else {
    P2M_CHECK_ARG(amax == nullptr && !real_rows_only, "amax_out / real_rows_only need F % 4 == 0");
    long tot = M * F;
    hipLaunchKernelGGL(k_bn_act_fwd_generic, dim3(cdiv(tot, 256)), dim3(256), 0, s, y, scale, shift, relu, resid,
                       Fres, res_shift, x, (long)M, F);
  }
  return check_launch("bn_act_fwd");
}

// bound of max(fma(y, scale, shift), 0) given amax(y): one block, atomic max into the word (include/p2m.h)
__global__ __launch_bounds__(256) void k_act_bound(const float* __restrict__ scale, const float* __restrict__ shift, int N,
                                                   const unsigned* __restrict__ y_amax, unsigned* __restrict__ word) {
  const float A = __uint_as_float(*y_amax);
  float v = 0.f;
  for (int f = threadIdx.x; f < N; f += 256) v = fmaxf(v, fmaf(A, fabsf(scale[f]), fmaxf(shift[f], 0.f)));
  amax_commit(word, v);
}

extern "C" int p2m_act_bound(const float* scale, const float* shift, int32_t N, const void* y_amax, void* word,
                             void* stream) {
  P2M_CHECK_ARG(scale && shift && y_amax && word && N > 0, "null pointer or empty shape");
  hipLaunchKernelGGL(k_act_bound, dim3(1), dim3(256), 0, (hipStream_t)stream, scale, shift, N,
                     static_cast<const unsigned*>(y_amax), static_cast<unsigned*>(word));
  return check_launch("act_bound");
}

extern "C" int32_t p2m_bn_bwd_blocks(int64_t M, int32_t F) {
  (void)F;
  return cdiv(M, bwd_rows_per_block(M));
}
// the same when the reduction runs with `classes` (it walks the live rows only)
extern "C" int32_t p2m_bn_bwd_blocks_classes(p2m_graph_t classes, int64_t M, int32_t F) {
  RowMap m;
  long Mlog;
  if (!row_map_of(classes, M, false, &m, &Mlog)) return 0;
  return p2m_bn_bwd_blocks(Mlog, F);
}

extern "C" int p2m_bn_bwd_reduce(const float* gx, const float* y, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, int32_t relu, float* part, int64_t M,
                                 int32_t F, p2m_graph_t classes, void* stream) {
  P2M_CHECK_ARG(gx && y && scale && shift && mean && invstd && part && M > 0 && F > 0, "null pointer or empty shape");
  RowMap m;
  long Mlog;
  P2M_CHECK_ARG(row_map_of(classes, M, false, &m, &Mlog), "M is not a multiple of the level's vertex count (or too large)");
  hipStream_t s = (hipStream_t)stream;
  const int grid = p2m_bn_bwd_blocks(Mlog, F);
  const int rpb = bwd_rows_per_block(Mlog);
  switch (F) {
    case 32:  hipLaunchKernelGGL(k_bn_bwd_reduce<8>,  dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    case 64:  hipLaunchKernelGGL(k_bn_bwd_reduce<16>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    case 128: hipLaunchKernelGGL(k_bn_bwd_reduce<32>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    case 256: hipLaunchKernelGGL(k_bn_bwd_reduce<64>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    default:
      hipLaunchKernelGGL(k_bn_bwd_reduce_generic, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, F, m, rpb);
  }
  return check_launch("bn_bwd_reduce");
}

// The reduction over the FAKE-vertex rows of a level only (its fake_ids: every fake vertex, or - with classes - the
// representatives, which carry their class's summed gradient): the other half of a reduction whose real-vertex rows were
// summed by the kernel that produced gx (p2m_cheb_tile_gemm, bnr_* arguments).
static bool fake_map_of(p2m_graph_t gh, int32_t B, RowMap* m, long* Mlog) {
  if (gh == nullptr || B <= 0) return false;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (g.n_fake <= 0 || (long)B * g.V >= (1LL << 32)) return false;
  m->w = nullptr; m->ids = g.fake_ids; m->n = (unsigned)g.n_fake; m->V = (unsigned)g.V;
  *Mlog = (long)B * g.n_fake;
  return true;
}
extern "C" int32_t p2m_bn_bwd_blocks_fake(p2m_graph_t gh, int32_t B, int32_t F) {
  RowMap m;
  long Mlog;
  if (!fake_map_of(gh, B, &m, &Mlog)) return 0;
  return p2m_bn_bwd_blocks(Mlog, F);
}
extern "C" int p2m_bn_bwd_reduce_fake(p2m_graph_t gh, const float* gx, const float* y, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, int32_t relu,
                                      float* part, int32_t B, int32_t F, void* stream) {
  P2M_CHECK_ARG(gx && y && scale && shift && mean && invstd && part && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(F == 32 || F == 64 || F == 128 || F == 256, "F must be 32, 64, 128 or 256");
  RowMap m;
  long Mlog;
  P2M_CHECK_ARG(fake_map_of(gh, B, &m, &Mlog), "the level has no fake vertices (or B * V does not fit 32 bits)");
  hipStream_t s = (hipStream_t)stream;
  const int grid = p2m_bn_bwd_blocks(Mlog, F);
  const int rpb = bwd_rows_per_block(Mlog);
  switch (F) {
    case 32:  hipLaunchKernelGGL(k_bn_bwd_reduce<8>,  dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    case 64:  hipLaunchKernelGGL(k_bn_bwd_reduce<16>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    case 128: hipLaunchKernelGGL(k_bn_bwd_reduce<32>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
    default:  hipLaunchKernelGGL(k_bn_bwd_reduce<64>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, relu, part, Mlog, m, rpb); break;
  }
  return check_launch("bn_bwd_reduce_fake");
}

extern "C" int p2m_bn_bwd_finalize(const float* part, int32_t nblk, int64_t M, float* dgamma, float* dbeta,
                                   float* coef, int32_t accumulate, int32_t F, void* stream) {
  P2M_CHECK_ARG(part && nblk > 0 && M > 0 && F > 0, "null pointer or empty shape");
  hipStream_t s = (hipStream_t)stream;
  double* scratch = fin_scratch(F, stream);
  if (scratch == nullptr) {
    set_error("p2m_bn_bwd_finalize: no scratch for %d columns (max %d) or hipMalloc failed", F, FIN_MAXN);
    return P2M_ERR_NOMEM;
  }
  const int nsplits = fin_splits(nblk);
  hipLaunchKernelGGL(k_bn_bwd_partial, dim3(cdiv(F, FIN_COLS), nsplits), dim3(FIN_COLS * FIN_RG), 0, s, part, nblk, F,
                     scratch, nsplits);
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(cdiv(F, 64)), dim3(64 * FIN_Q), 0, s, scratch, (long)M, dgamma, dbeta, coef,
                     accumulate, F, nsplits);
  return check_launch("bn_bwd_finalize");
}

extern "C" int p2m_bn_bwd_apply(const float* gx, const float* y, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* gamma, const float* coef,
                                int32_t relu, float* gy, float* pair_gx, float* pair_gy, int64_t M, int32_t F,
                                p2m_graph_t classes, int32_t zero_holes, void* amax_out, void* stream) {
  P2M_CHECK_ARG(gx && y && scale && shift && mean && invstd && gamma && gy && M > 0, "null pointer or empty shape");
  unsigned* amax = static_cast<unsigned*>(amax_out);
  RowMap m;
  long Mlog;
  hipStream_t s = (hipStream_t)stream;
  // zero_holes: walk every row instead of the live ones (RowMap: ids = nullptr, w kept)
  auto dense = [&](bool pairs, long* Ml) {
    if (!zero_holes || m.ids == nullptr) return;
    m.ids = nullptr;
    m.n = m.V;
    *Ml = pairs ? M / 2 : M;
  };
  if (pair_gx || pair_gy) {
    P2M_CHECK_ARG(M % 2 == 0 && (F == 32 || F == 64 || F == 128 || F == 256),
                  "pair-sum by-products need an even row count and F in {32, 64, 128, 256}");
    long Mp;
    P2M_CHECK_ARG(row_map_of(classes, M, true, &m, &Mp), "M is not a multiple of the level's (even) vertex count");
    dense(true, &Mp);
    const int gridp = cdiv(Mp, APPLY_ROWS_PER_BLOCK / 2);
    switch (F) {
      case 32:  hipLaunchKernelGGL(k_bn_bwd_apply_pairs<8>,  dim3(gridp), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, pair_gx, pair_gy, Mp, m, amax); break;
      case 64:  hipLaunchKernelGGL(k_bn_bwd_apply_pairs<16>, dim3(gridp), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, pair_gx, pair_gy, Mp, m, amax); break;
      case 128: hipLaunchKernelGGL(k_bn_bwd_apply_pairs<32>, dim3(gridp), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, pair_gx, pair_gy, Mp, m, amax); break;
      default:  hipLaunchKernelGGL(k_bn_bwd_apply_pairs<64>, dim3(gridp), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, pair_gx, pair_gy, Mp, m, amax); break;
    }
    return check_launch("bn_bwd_apply(pairs)");
  }
  P2M_CHECK_ARG(row_map_of(classes, M, false, &m, &Mlog), "M is not a multiple of the level's vertex count (or too large)");
  dense(false, &Mlog);
  const int grid = cdiv(Mlog, APPLY_ROWS_PER_BLOCK);
  switch (F) {
    case 32:  hipLaunchKernelGGL(k_bn_bwd_apply<8>,  dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, Mlog, m, amax); break;
    case 64:  hipLaunchKernelGGL(k_bn_bwd_apply<16>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, Mlog, m, amax); break;
    case 128: hipLaunchKernelGGL(k_bn_bwd_apply<32>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, Mlog, m, amax); break;
    case 256: hipLaunchKernelGGL(k_bn_bwd_apply<64>, dim3(grid), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, Mlog, m, amax); break;
    default:
      hipLaunchKernelGGL(k_bn_bwd_apply_generic, dim3(cdiv(Mlog * F, 256)), dim3(256), 0, s, gx, y, scale, shift, mean, invstd, gamma, coef, relu, gy, Mlog, F, m, amax);
  }
  return check_launch("bn_bwd_apply");
}

extern "C" int p2m_stats_rows_w(p2m_graph_t gh, const float* y, int32_t B, int32_t N, float* stats, void* stream) {
  P2M_CHECK_ARG(gh && y && stats && N > 0, "null pointer or empty shape");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(g.w != nullptr, "the handle has no classes (p2m_graph_set_classes)");
  if (B <= 0 || g.n_fake == 0) return P2M_OK;
  const int tps = cdiv(g.n_fake, 128);
  if (N % 4 == 0 && N / 4 <= 256 && 256 % (N / 4) == 0)
    hipLaunchKernelGGL(k_stats_rows_w4, dim3(B * tps), dim3(256), 0, (hipStream_t)stream, y, g.fake_ids, g.fake_wts,
                       g.n_fake, g.V, tps, N, stats);
  else
    hipLaunchKernelGGL(k_stats_rows_w, dim3(B * tps), dim3(256), 0, (hipStream_t)stream, y, g.fake_ids, g.fake_wts,
                       g.n_fake, g.V, tps, N, stats);
  return check_launch("stats_rows_w");
}

extern "C" int p2m_class_reduce(p2m_graph_t gh, const float* in, float* out, int32_t B, int32_t F, void* stream) {
  P2M_CHECK_ARG(gh && in && out && F > 0, "null pointer or empty shape");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(g.w != nullptr, "the handle has no classes (p2m_graph_set_classes)");
  if (B <= 0) return P2M_OK;
  const long M = (long)B * g.V;
  hipLaunchKernelGGL(k_class_reduce, dim3(cdiv(M * F, 256)), dim3(256), 0, (hipStream_t)stream, in, g.w, out, M, g.V, F);
  return check_launch("class_reduce");
}

extern "C" int p2m_pair_sum(const float* in, float* out, int64_t Mout, int32_t F, p2m_graph_t classes, void* stream) {
  P2M_CHECK_ARG(in && out && F > 0 && F % 4 == 0, "null pointer or feature width not a multiple of 4");
  if (Mout <= 0) return P2M_OK;
  const float* w;
  int V;
  P2M_CHECK_ARG(class_table(classes, 2 * Mout, &w, &V) && (w == nullptr || V % 2 == 0),
                "row count is not a multiple of the level's (even) vertex count");
  long tot = Mout * (F / 4);
  hipLaunchKernelGGL(k_pair_sum, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long)Mout, F, w, V);
  return check_launch("pair_sum");
}

extern "C" int p2m_lerp_bwd_add(const float* g, float* dst, int64_t M, int32_t F, int32_t Fres, void* stream) {
  P2M_CHECK_ARG(g && dst && F > 0 && Fres > 0, "null pointer or empty shape");
  if (M <= 0) return P2M_OK;
  if (Fres == 2 * F && F % 2 == 0) {
    hipLaunchKernelGGL(k_lerp_bwd_add_half, dim3(cdiv(M * (F / 2), 256)), dim3(256), 0, (hipStream_t)stream, g, dst,
                       (long)M, F);
    return check_launch("lerp_bwd_add");
  }
  long tot = M * Fres;
  hipLaunchKernelGGL(k_lerp_bwd_add, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, g, dst, (long)M, F, Fres);
  return check_launch("lerp_bwd_add");
}
