// PoseNet, the 2D -> 3D pose lifter in front of MeshNet (lib/models/posenet.py:11-92): a 4096-wide residual MLP over B
// rows - Linear -> 2 x [BatchNorm1d -> ReLU -> Dropout -> Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear, + input]
// -> Linear.
//
// At a batch of a few hundred rows every Linear is a WEIGHT-STREAMING contraction (67 MB of fp32 weights against 4 MB of
// activations): all of them - forward, dX, dW - run on p2m_gemm_tn, the reduction-split contraction with both operands
// row-major over the reduction index (the shape the fc lift of lib/models/meshnet.py:105 already takes; dW straight into
// the parameter's .grad with p2m_gemm_tn_acc).  What is left between two contractions is a [B, F] tensor with B small, and
// a block that owns 32 COLUMNS (a thread: 4 of them, 16-byte accesses) owns all B rows of them - so everything BatchNorm1d needs (batch mean / variance over the B
// rows, lib/models/posenet.py:28,33) is block-local and ONE kernel does the whole elementwise stage:
//
//   k_pn_stage_fwd   z = sum of the contraction's partials + bias (+ residual)          posenet.py:31,36,38 / 79,85
//                    a = dropout(relu(batch_norm(z)))  (or a = z), stored row-major AND transposed (the next
//                    contraction reduces over the feature index: it wants a^T), batch statistics, running statistics
//   k_pn_stage_bwd   g_a = sum of the dX contraction's partials; backward of dropout / ReLU / BatchNorm1d (train: batch
//                    statistics; eval: running statistics) (+ the residual branch's gradient) -> g_z row-major and
//                    transposed, d gamma, d beta, and the bias gradient of the Linear that produced z
//
// Reference arithmetic: torch.nn.functional.batch_norm / relu / dropout / linear in fp32 (posenet.py:25-38,77-87); the
// dropout mask is drawn by the caller (a uniform [0, 1) tensor: keep where u >= p, scale 1 / (1 - p) - nn.Dropout's
// Bernoulli(1 - p) mask; which elements are dropped follows torch's device generator, as in the reference).
#include "p2m_common.h"

namespace p2m {

constexpr int PN_COLS = 32;          // columns per block: one 128-byte line per row
constexpr int PN_Q = PN_COLS / 4;    // a thread owns 4 adjacent columns (one 16-byte access per row) ...
constexpr int PN_RG = 256 / PN_Q;    // ... and the rows ty, ty + 32, ...: 32 row groups
constexpr int PN_TROWS = 256;        // rows per transposed-store pass (LDS tile 32 x 257 floats)
typedef float pn4 __attribute__((ext_vector_type(4)));

struct PnFwdArgs {
  const float* P;        // [nch][B][F] partials (nch >= 1), or the tensor itself (nch == 1)
  int nch;
  const float* bias;     // [F] or null
  const float* resid;    // [B][F] or null
  float* z;              // [B][F] out, or null (then P is read again instead - only legal when nch == 1 and no bias/resid)
  // BatchNorm1d + ReLU + dropout (has_bn != 0), else a = z
  int has_bn, training;
  const float* gamma;
  const float* beta;
  float* running_mean;   // updated when training and not null
  float* running_var;
  float momentum, eps;
  const float* rnd;      // [B][F] uniform [0, 1), or null (no dropout)
  float p_drop;
  float* a;              // [B][F] out or null
  float* aT;             // [F][B] out or null
  float* mean;           // [F] out (has_bn): the statistics used
  float* invstd;         // [F] out
  unsigned* amax_out;    // optional: max |a| (atomic max into a zeroed word)
  int B, F;
  int Br;                // rows that hold samples (<= B): rows >= Br are PADDING (batches the contractions cannot take as
                         // they are - B < 32 or B % 4 != 0 - are zero-padded by the caller): left out of the batch
                         // statistics, and their a / aT values are stored as 0 so that they stay zero through every Linear
};

__device__ __forceinline__ void pn_block_amax(unsigned* word, float m) {
  __shared__ float wmax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) {
      const unsigned bits = __float_as_uint(m);
      if (bits > __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(word, bits);
    }
  }
}

// column sums over the block's rows: every thread holds the partials of its 4 columns; result: the totals of the same 4
// columns in every thread (fixed summation order: deterministic)
__device__ __forceinline__ pn4 pn_colsum(pn4 v, float (*red)[PN_COLS]) {
  const int tq = threadIdx.x & (PN_Q - 1), ty = threadIdx.x / PN_Q;
  __syncthreads();                       // (red may still be read from the previous reduction)
  *reinterpret_cast<pn4*>(&red[ty][4 * tq]) = v;
  __syncthreads();
  pn4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int q = 0; q < PN_RG; q++) s += *reinterpret_cast<const pn4*>(&red[q][4 * tq]);
  return s;
}

// out^T[c0 + cc][r] = tile value: rows in passes of PN_TROWS through LDS so that the global stores run along r
__device__ __forceinline__ void pn_store_transposed(float* __restrict__ outT, float (*tile)[PN_TROWS + 1], int c0, int F,
                                                    int rbase, int nrows, int B) {
  __syncthreads();
  for (int cc = 0; cc < PN_COLS; cc++) {
    if (c0 + cc >= F) break;
    for (int rr = threadIdx.x; rr < nrows; rr += 256) outT[(long)(c0 + cc) * B + rbase + rr] = tile[cc][rr];
  }
  __syncthreads();
}

__device__ __forceinline__ pn4 pn_ld4(const float* p) { return *reinterpret_cast<const pn4*>(p); }
__device__ __forceinline__ pn4 pn_ld4_or(const float* p, int c, float dflt) {      // p == nullptr: the default
  return p != nullptr ? pn_ld4(p + c) : pn4{dflt, dflt, dflt, dflt};
}

__global__ __launch_bounds__(256) void k_pn_stage_fwd(PnFwdArgs g) {
  __shared__ __attribute__((aligned(16))) float red[PN_RG][PN_COLS];
  __shared__ float tile[PN_COLS][PN_TROWS + 1];
  const int tq = threadIdx.x & (PN_Q - 1), ty = threadIdx.x / PN_Q;
  const int c0 = blockIdx.x * PN_COLS, c = c0 + 4 * tq;
  const bool live = c < g.F;                                  // F % 4 == 0: a column quad is inside or outside
  const long BF = (long)g.B * g.F;
  const pn4 bias = live ? pn_ld4_or(g.bias, c, 0.f) : pn4{0.f, 0.f, 0.f, 0.f};
  // ---- pass 1: z = sum of partials + bias (+ residual); column sums
  const float* zsrc = g.z != nullptr ? g.z : g.P;
  pn4 s = {0.f, 0.f, 0.f, 0.f};
  if (live && (g.z != nullptr || g.has_bn)) {
    for (int r = ty; r < g.B; r += 4 * PN_RG) {            // 4 rows per pass: 4 x nch 16-byte loads in flight per thread
      pn4 v[4];
      long o[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int ru = r + u * PN_RG < g.B ? r + u * PN_RG : g.B - 1;      // clamped: the duplicate is not used
        o[u] = (long)ru * g.F + c;
        v[u] = g.z != nullptr ? bias : pn4{0.f, 0.f, 0.f, 0.f};
      }
      for (int ch = 0; ch < g.nch; ch++) {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] += pn_ld4(g.P + ch * BF + o[u]);
      }
      if (g.z != nullptr && g.resid != nullptr) {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] += pn_ld4(g.resid + o[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (r + u * PN_RG >= g.B) break;
        if (g.z != nullptr) *reinterpret_cast<pn4*>(g.z + o[u]) = v[u];
        if (r + u * PN_RG < g.Br) s += v[u];
      }
    }
  }
  pn4 mean = {0.f, 0.f, 0.f, 0.f}, invstd = {1.f, 1.f, 1.f, 1.f}, ga = invstd, be = mean;
  if (g.has_bn) {
    if (g.training) {
      const pn4 tot = pn_colsum(s, red);
      mean = tot / (float)g.Br;
      pn4 m2 = {0.f, 0.f, 0.f, 0.f};
      if (live)
        for (int r = ty; r < g.Br; r += PN_RG) {
          const pn4 d = pn_ld4(zsrc + (long)r * g.F + c) - mean;
          m2 += d * d;
        }
      const pn4 var = pn_colsum(m2, red) / (float)g.Br;               // biased: what normalises (F.batch_norm)
#pragma unroll
      for (int e = 0; e < 4; e++) invstd[e] = 1.0f / sqrtf(var[e] + g.eps);
      if (live && ty == 0 && g.running_mean != nullptr) {
        // nn.BatchNorm1d: running = (1 - momentum) * running + momentum * batch statistic, UNBIASED variance
        const float ub = g.Br > 1 ? (float)g.Br / (float)(g.Br - 1) : 1.f;
        const pn4 rm = pn_ld4(g.running_mean + c), rv = pn_ld4(g.running_var + c);
        *reinterpret_cast<pn4*>(g.running_mean + c) = (1.f - g.momentum) * rm + g.momentum * mean;
        *reinterpret_cast<pn4*>(g.running_var + c) = (1.f - g.momentum) * rv + g.momentum * (var * ub);
      }
    } else if (live) {
      mean = pn_ld4(g.running_mean + c);
      const pn4 rv = pn_ld4(g.running_var + c);
#pragma unroll
      for (int e = 0; e < 4; e++) invstd[e] = 1.0f / sqrtf(rv[e] + g.eps);
    }
    if (live) {
      ga = pn_ld4_or(g.gamma, c, 1.f);
      be = pn_ld4_or(g.beta, c, 0.f);
      if (ty == 0) {
        *reinterpret_cast<pn4*>(g.mean + c) = mean;
        *reinterpret_cast<pn4*>(g.invstd + c) = invstd;
      }
    }
  }
  if (g.a == nullptr && g.aT == nullptr) return;
  // ---- pass 2: a = dropout(relu(bn(z))) (or z), row-major and transposed
  const float keep_scale = g.rnd != nullptr ? 1.0f / (1.0f - g.p_drop) : 1.f;
  float vmax = 0.f;
  for (int rbase = 0; rbase < g.B; rbase += PN_TROWS) {
    const int nrows = g.B - rbase < PN_TROWS ? g.B - rbase : PN_TROWS;
    for (int rr = ty; rr < nrows; rr += PN_RG) {
      pn4 v = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        const long o = (long)(rbase + rr) * g.F + c;
        const bool pad = rbase + rr >= g.Br;
        if (!pad) v = pn_ld4(zsrc + o);
        if (g.has_bn && !pad) {
          pn4 u = {1.f, 1.f, 1.f, 1.f};
          if (g.rnd != nullptr) u = pn_ld4(g.rnd + o);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float w = fmaxf(fmaf((v[e] - mean[e]) * invstd[e], ga[e], be[e]), 0.f);
            if (g.rnd != nullptr) w = u[e] >= g.p_drop ? w * keep_scale : 0.f;
            v[e] = w;
          }
        }
        if (g.a != nullptr) *reinterpret_cast<pn4*>(g.a + o) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) vmax = fmaxf(vmax, amax_abs(v[e]));
      }
#pragma unroll
      for (int e = 0; e < 4; e++) tile[4 * tq + e][rr] = v[e];
    }
    if (g.aT != nullptr) pn_store_transposed(g.aT, tile, c0, g.F, rbase, nrows, g.B);
  }
  if (g.amax_out != nullptr) pn_block_amax(g.amax_out, vmax);
}

struct PnBwdArgs {
  const float* P;        // [nch][B][F] partials of g_a (the gradient w.r.t. the stage's OUTPUT a), nch >= 1
  int nch;
  const float* addend;   // [B][F] added to the result (the residual branch's gradient) or null
  // backward of dropout(relu(batch_norm(z))) (has_bn != 0), else g_z = g_a
  int has_bn, training;
  const float* z;        // [B][F]
  const float* mean;     // [F] the statistics the forward used
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* rnd;
  float p_drop;
  float* gz;             // [B][F] out (also scratch for the masked gradient between the passes)
  float* gzT;            // [F][B] out or null
  float* dgamma;         // [F] or null
  float* dbeta;
  float* dbias;          // [F] or null: sum over the rows of what is stored in gz (the producing Linear's bias gradient)
  int accumulate;        // dgamma / dbeta / dbias: += instead of =
  unsigned* amax_out;
  int B, F;
  int Br;                // rows that hold samples (see PnFwdArgs): the gradient of a padding row is 0 in and 0 out
};

__device__ __forceinline__ void pn_store_vec(float* dst, int c, pn4 v, int accumulate) {
  if (dst == nullptr) return;
  pn4* d = reinterpret_cast<pn4*>(dst + c);
  *d = accumulate ? *d + v : v;
}

__global__ __launch_bounds__(256) void k_pn_stage_bwd(PnBwdArgs g) {
  __shared__ __attribute__((aligned(16))) float red[PN_RG][PN_COLS];
  __shared__ float tile[PN_COLS][PN_TROWS + 1];
  const int tq = threadIdx.x & (PN_Q - 1), ty = threadIdx.x / PN_Q;
  const int c0 = blockIdx.x * PN_COLS, c = c0 + 4 * tq;
  const bool live = c < g.F;
  const long BF = (long)g.B * g.F;
  pn4 mean = {0.f, 0.f, 0.f, 0.f}, invstd = {1.f, 1.f, 1.f, 1.f}, ga = invstd, be = mean;
  if (g.has_bn && live) {
    mean = pn_ld4(g.mean + c);
    invstd = pn_ld4(g.invstd + c);
    ga = pn_ld4_or(g.gamma, c, 1.f);
    be = pn_ld4_or(g.beta, c, 0.f);
  }
  const float keep_scale = g.rnd != nullptr ? 1.0f / (1.0f - g.p_drop) : 1.f;
  // ---- pass 1: g_u = g_a * dropout mask * relu mask, kept in gz; s0 = sum g_u, s1 = sum g_u * xhat
  pn4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (live)
    for (int r = ty; r < g.B; r += 4 * PN_RG) {            // 4 rows per pass: 4 x nch 16-byte loads in flight per thread
      pn4 v[4];
      long o[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int ru = r + u * PN_RG < g.B ? r + u * PN_RG : g.B - 1;
        o[u] = (long)ru * g.F + c;
        v[u] = pn4{0.f, 0.f, 0.f, 0.f};
      }
      for (int ch = 0; ch < g.nch; ch++) {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] += pn_ld4(g.P + ch * BF + o[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (r + u * PN_RG >= g.B) break;
        pn4 w = v[u];
        if (r + u * PN_RG >= g.Br) w = pn4{0.f, 0.f, 0.f, 0.f};
        else if (g.has_bn) {
          const pn4 xh = (pn_ld4(g.z + o[u]) - mean) * invstd;
          pn4 rn = {1.f, 1.f, 1.f, 1.f};
          if (g.rnd != nullptr) rn = pn_ld4(g.rnd + o[u]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (g.rnd != nullptr) w[e] = rn[e] >= g.p_drop ? w[e] * keep_scale : 0.f;
            if (fmaf(xh[e], ga[e], be[e]) <= 0.f) w[e] = 0.f;
          }
          s0 += w;
          s1 += w * xh;
        }
        *reinterpret_cast<pn4*>(g.gz + o[u]) = w;
      }
    }
  pn4 c0m = {0.f, 0.f, 0.f, 0.f}, c1m = c0m;
  if (g.has_bn) {
    const pn4 t0 = pn_colsum(s0, red), t1 = pn_colsum(s1, red);
    if (live && ty == 0) {
      pn_store_vec(g.dbeta, c, t0, g.accumulate);
      pn_store_vec(g.dgamma, c, t1, g.accumulate);
    }
    if (g.training) {
      c0m = t0 / (float)g.Br;
      c1m = t1 / (float)g.Br;
    }
  }
  // ---- pass 2: g_z = gamma * invstd * (g_u - mean(g_u) - xhat * mean(g_u xhat))  (eval: gamma * invstd * g_u) (+ addend)
  const pn4 k = ga * invstd;
  float vmax = 0.f;
  pn4 sb = {0.f, 0.f, 0.f, 0.f};
  for (int rbase = 0; rbase < g.B; rbase += PN_TROWS) {
    const int nrows = g.B - rbase < PN_TROWS ? g.B - rbase : PN_TROWS;
    for (int rr = ty; rr < nrows; rr += PN_RG) {
      pn4 v = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        const long o = (long)(rbase + rr) * g.F + c;
        if (rbase + rr < g.Br) {
          v = pn_ld4(g.gz + o);                           // this thread's own store of pass 1
          if (g.has_bn) {
            const pn4 xh = (pn_ld4(g.z + o) - mean) * invstd;
            v = k * (v - c0m - xh * c1m);
          }
          if (g.addend != nullptr) v += pn_ld4(g.addend + o);
        }
        *reinterpret_cast<pn4*>(g.gz + o) = v;
        sb += v;
#pragma unroll
        for (int e = 0; e < 4; e++) vmax = fmaxf(vmax, amax_abs(v[e]));
      }
#pragma unroll
      for (int e = 0; e < 4; e++) tile[4 * tq + e][rr] = v[e];
    }
    if (g.gzT != nullptr) pn_store_transposed(g.gzT, tile, c0, g.F, rbase, nrows, g.B);
  }
  if (g.dbias != nullptr) {
    const pn4 t = pn_colsum(sb, red);
    if (live && ty == 0) pn_store_vec(g.dbias, c, t, g.accumulate);
  }
  if (g.amax_out != nullptr) pn_block_amax(g.amax_out, vmax);
}

}  // namespace p2m

using namespace p2m;

extern "C" int p2m_pn_stage_fwd(const float* P, int32_t nch, const float* bias, const float* resid, float* z,
                                int32_t has_bn, int32_t training, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, const float* rnd,
                                float p_drop, float* a, float* aT, float* mean, float* invstd, void* amax_out, int32_t B,
                                int32_t F, int32_t B_real, void* stream) {
  P2M_CHECK_ARG(P != nullptr && nch >= 1 && B > 0 && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(F % 4 == 0, "the feature count must be a multiple of 4 (16-byte accesses)");
  P2M_CHECK_ARG(B_real >= 0 && B_real <= B, "B_real must be 0 (= B) or 1 .. B");
  P2M_CHECK_ARG(z != nullptr || (nch == 1 && bias == nullptr && resid == nullptr),
                "without a z output P must be the tensor itself (one chunk, no bias, no residual)");
  P2M_CHECK_ARG(!has_bn || (mean != nullptr && invstd != nullptr), "BatchNorm needs the mean / invstd outputs");
  P2M_CHECK_ARG(!has_bn || training || (running_mean != nullptr && running_var != nullptr),
                "eval-mode BatchNorm needs the running statistics");
  P2M_CHECK_ARG(rnd == nullptr || (p_drop >= 0.f && p_drop < 1.f), "dropout probability must be in [0, 1)");
  PnFwdArgs g;
  g.P = P; g.nch = nch; g.bias = bias; g.resid = resid; g.z = z;
  g.has_bn = has_bn; g.training = training; g.gamma = gamma; g.beta = beta;
  g.running_mean = running_mean; g.running_var = running_var; g.momentum = momentum; g.eps = eps;
  g.rnd = (has_bn && rnd != nullptr && p_drop > 0.f) ? rnd : nullptr; g.p_drop = p_drop;
  g.a = a; g.aT = aT; g.mean = mean; g.invstd = invstd;
  g.amax_out = static_cast<unsigned*>(amax_out); g.B = B; g.F = F; g.Br = B_real > 0 ? B_real : B;
  hipLaunchKernelGGL(k_pn_stage_fwd, dim3(cdiv(F, PN_COLS)), dim3(256), 0, (hipStream_t)stream, g);
  return check_launch("pn_stage_fwd");
}

extern "C" int p2m_pn_stage_bwd(const float* P, int32_t nch, const float* addend, int32_t has_bn, int32_t training,
                                const float* z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, const float* rnd, float p_drop, float* gz, float* gzT, float* dgamma,
                                float* dbeta, float* dbias, int32_t accumulate, void* amax_out, int32_t B, int32_t F,
                                int32_t B_real, void* stream) {
  P2M_CHECK_ARG(P != nullptr && nch >= 1 && gz != nullptr && B > 0 && F > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(F % 4 == 0, "the feature count must be a multiple of 4 (16-byte accesses)");
  P2M_CHECK_ARG(B_real >= 0 && B_real <= B, "B_real must be 0 (= B) or 1 .. B");
  P2M_CHECK_ARG(!has_bn || (z != nullptr && mean != nullptr && invstd != nullptr),
                "the BatchNorm backward needs z and the statistics of the forward");
  P2M_CHECK_ARG(rnd == nullptr || (p_drop >= 0.f && p_drop < 1.f), "dropout probability must be in [0, 1)");
  PnBwdArgs g;
  g.P = P; g.nch = nch; g.addend = addend; g.has_bn = has_bn; g.training = training;
  g.z = z; g.mean = mean; g.invstd = invstd; g.gamma = gamma; g.beta = beta;
  g.rnd = (has_bn && rnd != nullptr && p_drop > 0.f) ? rnd : nullptr; g.p_drop = p_drop;
  g.gz = gz; g.gzT = gzT; g.dgamma = dgamma; g.dbeta = dbeta; g.dbias = dbias; g.accumulate = accumulate;
  g.amax_out = static_cast<unsigned*>(amax_out); g.B = B; g.F = F; g.Br = B_real > 0 ? B_real : B;
  hipLaunchKernelGGL(k_pn_stage_bwd, dim3(cdiv(F, PN_COLS)), dim3(256), 0, (hipStream_t)stream, g);
  return check_launch("pn_stage_bwd");
}
