// Dense contractions of the Pose2Mesh Chebyshev GCN on gfx950 matrix cores.
//
//   k_gemm_planes[_bx|_ws] : C = [A0|A1|A2] * Bm + bias  (+ addend / un-pool pair-sum, BatchNorm partials in the epilogue)
//   k_gemm_tn[_bx|_ws]     : P[chunk] = [A0|A1|A2]^T * G  over a chunk of rows (weight gradient)
//   k_naive_*              : scalar fall-backs for the odd shapes (Fin=5 first conv, Fout=3 last conv)
//
// Replaces nn.Linear inside graph_conv_cheby (lib/models/backbones/cheby_graph_conv.py:37), the
// fc lift (lib/models/meshnet.py:105) and their autograd backward.  1e-4 vertex parity needs fp32 products; gfx950
// has no TF32/xf32.  Three arithmetics compute the same fp32 contraction (include/p2m.h, P2M_ARITH_*):
//   plain    native f32 MFMA v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, 157 TFLOP/s peak);
//   _ws<3>   every operand cut exactly into three bf16 slices, six slice products per fp32 product on
//            v_mfma_f32_32x32x16_bf16 (2500 / 6 = 417 TFLOP/s peak), fp32 accumulation;
//   _ws<2>   every operand scaled by a power of two (from its amax word) and cut into two fp16 slices, three slice
//            products on v_mfma_f32_32x32x16_f16 (2500 / 3 = 833 TFLOP/s peak), 22-bit operands (p2m_split.h).
//            _ws = wave-specialised (4 MFMA waves + 4 staging waves per block).
//
// Tiling (64-wide waves), all variants: the MFMA waves are arranged 2(M) x 2(N); block tile 128 x BN (BN = 128 or
// 64); each wave owns 64 x BN/2 = (2 x BN/64) MFMA 32x32 tiles; K chunk 32 (f32) or 16 (bf16 slices) per barrier,
// global -> registers -> LDS staging with register prefetch, two blocks per CU.
#include <cstdlib>
#include <type_traits>

#include "p2m_split.h"

namespace p2m {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int AS_LD = BK + 1;  // [m][k] with stride 33: bank = (m + k) % 32 -> conflict-free frag reads

struct GemmArgs {
  const float* A[3];
  const float* Bm;
  const float* bias;
  const float* addend;   // optional [M, N] added to the result (single output plane)
  // optional fused activation (eval-mode BatchNorm + ReLU folded into the contraction that produces the tensor):
  //   v = fmaf(acc + bias, act_scale[n], act_shift[n]);  v = max(v, 0) if act_relu   -- the same two roundings as the
  //   separate p2m_bn_act_fwd pass, so the fused result is bitwise the unfused one
  const float* act_scale;
  const float* act_shift;
  int act_relu;
  int pair_out;          // write C[(row>>1)] = v(row) + v(row^1): backward of the x2 un-pool (single output plane)
  float* C[3];
  float* stats;
  long M;
  int nplanesA, Ka, a0_shift;
  int N, Nc;
  int ntm, ntn;
  // row-set mode (ROWS kernels): logical row (b, i), i < nset -> actual row b*V + ids[i]; tiles do not cross samples
  // (tps tiles per sample); planes 1,2 of A are COMPACT ([B*nset, Ka]) when `compact`
  const int* ids;
  int nset, V, tps, compact;
  // optional weights of the logical rows (aligned with ids): the BatchNorm partials count row i row_w[i] times (class
  // representatives stand for their whole class); NULL = every row once
  const float* row_w;
  // split-bf16 mode (k_gemm_planes_bx): Bx[k / 16][s][n][k % 16], s < 3, n < Npad, k < Ktot = nplanesA * Ka
  const unsigned short* Bx;
  int Npad, Ktot;
  // two-fp16-slice mode: amax words (p2m_split.h) of the A planes (+ a_bits binades of headroom) and of the weight
  const unsigned* a_amax;
  const unsigned* b_amax;
  int a_bits;
  unsigned* amax_out;    // optional: receives max |value stored| (atomic max; the caller zeroes it)
  // activation on load of A plane 0 (k_gemm_planes_ws, Ka <= 256): the operand is max(fma(A0, in_scale[k], in_shift[k]), 0)
  const float* in_scale;
  const float* in_shift;
};

// blockIdx -> (m tile, n tile).  Blocks b, b+8, b+16.. share an XCD (observed dispatch: b % 8);
// give the n-tiles of one m-tile consecutive slots of the SAME XCD so the A tile is re-read
// from that XCD's L2, not from HBM.
__device__ __forceinline__ bool tile_of_block(int bid, int ntm, int ntn, int& mt, int& nt) {
  int xcd = bid & 7;
  int slot = bid >> 3;
  nt = slot % ntn;
  mt = (slot / ntn) * 8 + xcd;
  return mt < ntm;
}

// Epilogue shared by the native-fp32 and the split-bf16 kernels: bias (+ addend), store (or pair-sum store), and the
// ROWS kernels keep, behind the BM row indices, the BM row weights and the two per-wave weight totals (threads 0..BM-1 =
// waves 0, 1 fill them; the k loop's barriers order the fill before the epilogue reads it)
constexpr int ROWTAB_WORDS = 2 * BM + 2;
__device__ __forceinline__ void rows_weights_fill(const GemmArgs& g, int* rowtab, int i, int t) {
  float* wtab = reinterpret_cast<float*>(rowtab + BM);
  const float w = (i < g.nset) ? g.row_w[i] : 0.f;
  wtab[t] = w;
  float s = w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) wtab[BM + (t >> 6)] = s;
}

// per-tile BatchNorm partials (sum, centred M2).  `red` is block LDS that is free once the k loop is over.
template <int BN, bool EXTRA, bool ROWS>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, floatx16 (&acc)[2][BN / 64], float* smem,
                                              const int* rowtab, int mt, long m0, int n0, int rs_i0, int wm, int wn,
                                              int l31, int lhi, int descale = 0) {
  constexpr int WTN = BN / 2;
  constexpr int TN = WTN / 32;
  constexpr int TM = 2;
  // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float bias_v[TN], sc_v[TN], sh_v[TN];
  int ncol[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) {
    ncol[j] = n0 + wn * WTN + j * 32 + l31;
    bias_v[j] = (g.bias != nullptr && ncol[j] < g.N) ? g.bias[ncol[j]] : 0.f;
    sc_v[j] = (g.act_scale != nullptr && ncol[j] < g.N) ? g.act_scale[ncol[j]] : 1.f;
    sh_v[j] = (g.act_scale != nullptr && ncol[j] < g.N) ? g.act_shift[ncol[j]] : 0.f;
  }
  float csum[TN];
  float vmax = 0.f;
  const bool wst = ROWS && g.row_w != nullptr && g.stats != nullptr;      // weighted partials (block-uniform)
  const float* wtab = reinterpret_cast<const float*>(rowtab + BM);
#pragma unroll
  for (int j = 0; j < TN; j++) csum[j] = 0.f;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = ncol[j];
      const int q = n / g.Nc;
      const int c = n - q * g.Nc;
      float* Cq = (n < g.N) ? g.C[q] : nullptr;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        long row = ROWS ? (long)rowtab[ml] : m0 + ml;
        const bool rok = ROWS ? (row >= 0) : (row < g.M);
        float v = __builtin_ldexpf(acc[i][j][r], descale) + bias_v[j];     // descale: exact (two-fp16-slice mode), else 0
        if (g.act_scale != nullptr) v = fmaf(v, sc_v[j], sh_v[j]);
        if (g.act_relu) v = fmaxf(v, 0.f);
        if (EXTRA && g.addend != nullptr && rok && Cq != nullptr) v += g.addend[row * g.N + n];
        acc[i][j][r] = rok ? v : 0.f;
        if (!(EXTRA && g.pair_out) && rok && Cq != nullptr) {
#ifdef P2M_PL_ABL_NOSTORE
          if (v == 12345.678f)
#endif
          Cq[row * g.Nc + c] = v;
          csum[j] += wst ? wtab[ml] * v : v;
          vmax = fmaxf(vmax, amax_abs(v));
        }
        // (round 6) the row addresses of four accumulator registers at a time: left alone, the scheduler forms all 64 store
        // addresses first - 150-170 registers for a kernel whose k loop needs 112, i.e. one block per CU instead of two
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      if (EXTRA && g.pair_out && Cq != nullptr) {   // rows (r, r+1), r even: the two children of one coarse vertex
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          long row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (row < g.M) {
            const float pv = acc[i][j][r] + acc[i][j][r + 1];
            Cq[(row >> 1) * g.Nc + c] = pv;
            vmax = fmaxf(vmax, amax_abs(pv));
          }
        }
      }
    }
  if (g.amax_out != nullptr) amax_commit(g.amax_out, vmax);
  if (g.stats == nullptr) return;

  // column sums over the 128-row tile: lane^32 holds the same column, the other wm wave the other 64 rows
  float* red = smem;  // [2 (wm)][BN]   (safe: all waves passed the last __syncthreads of the k loop)
  long rows_valid = ROWS ? (long)(g.nset - rs_i0) : g.M - m0;
  if (rows_valid > BM) rows_valid = BM;
#pragma unroll
  for (int j = 0; j < TN; j++) {
    csum[j] += __shfl_xor(csum[j], 32);
    if (lhi == 0) red[wm * BN + wn * WTN + j * 32 + l31] = csum[j];
  }
  __syncthreads();
  float cm2[TN];
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int cl = wn * WTN + j * 32 + l31;
    const float tot = red[cl] + red[BN + cl];
    const float mean = tot / (wst ? wtab[BM] + wtab[BM + 1] : (float)rows_valid);
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float d = acc[i][j][r] - mean;
        if (ROWS ? (rowtab[ml] >= 0) : (m0 + ml < g.M)) m2 += wst ? wtab[ml] * d * d : d * d;
      }
    m2 += __shfl_xor(m2, 32);
    cm2[j] = m2;
    csum[j] = tot;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TN; j++)
    if (lhi == 0) red[wm * BN + wn * WTN + j * 32 + l31] = cm2[j];
  __syncthreads();
  if (wm == 0 && lhi == 0) {
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int cl = wn * WTN + j * 32 + l31;
      const int n = n0 + cl;
      if (n < g.N) {
        float* st = g.stats + (long)mt * 2 * g.N;
        st[n] = csum[j];
        st[g.N + n] = red[cl] + red[BN + cl];
      }
    }
  }
}

// KB = K chunk staged per barrier (16: 34 KB LDS -> 3 blocks/CU; 32: 67 KB -> 2 blocks/CU);
// EXTRA compiles in the addend / pair-sum epilogue (kept out of the plain variant: it costs registers -> occupancy)
template <int BN, int KB, bool EXTRA, bool ROWS = false>
__global__ __launch_bounds__(256, 2) void k_gemm_planes(GemmArgs g) {
  constexpr int WTN = BN / 2;    // wave tile N
  constexpr int TN = WTN / 32;   // MFMA tiles along N per wave
  constexpr int TM = 2;          // wave tile M = 64
  constexpr int LDA = KB + 1;    // [m][k] row stride: bank = (m + k) % 32 -> conflict-free fragment reads
  constexpr int APASS = BM * KB / 4 / 256;        // float4 A loads per thread per chunk
  constexpr int AROWS = 256 / (KB / 4);           // rows of A covered per pass
  constexpr int BPASS = KB * BN / 4 / 256;        // float4 B loads per thread per chunk
  constexpr int BROWS = 256 / (BN / 4);           // rows of B covered per pass
  __shared__ float smem[2 * BM * LDA + 2 * KB * BN + (ROWS ? ROWTAB_WORDS : 0)];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDA;
  int* rowtab = reinterpret_cast<int*>(smem + 2 * BM * LDA + 2 * KB * BN);   // ROWS: actual row of each tile row, -1 = none

  int mt, nt;
  if (!tile_of_block(blockIdx.x, g.ntm, g.ntn, mt, nt)) return;
  const long m0 = (long)mt * BM;      // flat mode: first row of the tile; ROWS mode: unused for addressing
  const int n0 = nt * BN;
  const int t = threadIdx.x;
  // ROWS mode: tile -> (sample, tile within the sample's row set)
  const int rs_b = ROWS ? mt / g.tps : 0;
  const int rs_i0 = ROWS ? (mt - rs_b * g.tps) * BM : 0;
  if (ROWS && t < BM) {
    const int i = rs_i0 + t;
    rowtab[t] = (i < g.nset) ? rs_b * g.V + g.ids[i] : -1;
    if (g.row_w != nullptr) rows_weights_fill(g, rowtab, i, t);
  }
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int cpp = g.Ka / KB;              // chunks per plane
  const int nchunks = g.nplanesA * cpp;

  // staging registers
  float4 ra[APASS];
  float4 rb[BPASS];
  const int a_row = t / (KB / 4), a_k4 = (t % (KB / 4)) * 4;
  const int b_row = t / (BN / 4), b_c4 = (t % (BN / 4)) * 4;
  // ROWS mode: per-thread source rows of its APASS tile rows (plane 0: full layout, planes 1,2: compact)
  long rs_full[ROWS ? APASS : 1], rs_comp[ROWS ? APASS : 1];
  if (ROWS) {
#pragma unroll
    for (int ps = 0; ps < APASS; ps++) {
      const int i = rs_i0 + ps * AROWS + a_row;
      if (i < g.nset) {
        rs_full[ps] = (long)rs_b * g.V + g.ids[i];
        rs_comp[ps] = (long)rs_b * g.nset + i;
      } else {
        rs_full[ps] = -1;
        rs_comp[ps] = -1;
      }
    }
  }

  auto load_chunk = [&](int kc) {
    const int p = kc / cpp;
    const int k0 = (kc - p * cpp) * KB;
    const float* Ap = g.A[p];
    const int sh = (p == 0) ? g.a0_shift : 0;
#pragma unroll
    for (int ps = 0; ps < APASS; ps++) {
      if (ROWS) {
        const long r = (p == 0 || !g.compact) ? rs_full[ps] : rs_comp[ps];
        if (r >= 0)
          ra[ps] = *reinterpret_cast<const float4*>(Ap + (r >> sh) * g.Ka + k0 + a_k4);
        else
          ra[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        long r = m0 + ps * AROWS + a_row;
        if (r < g.M)
          ra[ps] = *reinterpret_cast<const float4*>(Ap + (r >> sh) * g.Ka + k0 + a_k4);
        else
          ra[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float* Bp = g.Bm + (long)(p * g.Ka + k0) * g.N;
#pragma unroll
    for (int ps = 0; ps < BPASS; ps++) {
      int kr = ps * BROWS + b_row;
      int n = n0 + b_c4;
      if (n < g.N)
        rb[ps] = *reinterpret_cast<const float4*>(Bp + (long)kr * g.N + n);
      else
        rb[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf) {
    float* as = As + buf * BM * LDA;
#pragma unroll
    for (int ps = 0; ps < APASS; ps++) {
      float* d = as + (ps * AROWS + a_row) * LDA + a_k4;
      d[0] = ra[ps].x; d[1] = ra[ps].y; d[2] = ra[ps].z; d[3] = ra[ps].w;
    }
    float* bs = Bs + buf * KB * BN;
#pragma unroll
    for (int ps = 0; ps < BPASS; ps++)
      *reinterpret_cast<float4*>(bs + (ps * BROWS + b_row) * BN + b_c4) = rb[ps];
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int kc = 0; kc < nchunks; kc++) {
    const int cur = kc & 1;
    if (kc + 1 < nchunks) load_chunk(kc + 1);
    const float* as = As + cur * BM * LDA + (wm * 64 + l31) * LDA + lhi;
    const float* bs = Bs + cur * KB * BN + lhi * BN + wn * WTN + l31;
#pragma unroll
    for (int ks = 0; ks < KB / 2; ks++) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = as[i * 32 * LDA + 2 * ks];
#pragma unroll
      for (int j = 0; j < TN; j++) b[j] = bs[2 * ks * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < nchunks) store_chunk(cur ^ 1);
    __syncthreads();
  }

  gemm_epilogue<BN, EXTRA, ROWS>(g, acc, smem, rowtab, mt, m0, n0, rs_i0, wm, wn, l31, lhi);
}

// ---------------------------------------------------------------------------------------------
// fp32 contraction on the BF16 matrix pipe (16x the rate of the f32 MFMA on gfx950).
//
// Every fp32 operand is cut into three bf16 slices by TRUNCATION: x = h + m + l exactly (8 + 8 + 8 significand bits;
// each remainder x - h, x - h - m is exact in fp32).  The product keeps the six slice pairs of weight >= 2^-16,
//     x*y ~= h h' + (h m' + m h') + (h l' + l h' + m m'),
// each pair exact inside v_mfma_f32_32x32x16_bf16 (8b x 8b significands) and accumulated in fp32; the dropped pairs
// (m l', l m', l l') are <= 2^-23 |x y| together - one fp32 rounding.  tests/test_gpu_ops.py checks the result against
// float64 with the same bound as the native f32 MFMA kernel.  Six bf16 MFMAs cost 6/16 of one f32 MFMA per flop.
//
// Same block/wave tiling, row addressing and epilogue as k_gemm_planes.  A is split while it is staged (fp32 global ->
// 3 bf16 planes in LDS, split3_pack4 in p2m_split.h); the weights arrive pre-split and k-contiguous (p2m_weight_split),
// so their staging is a straight 16-byte copy.  LDS rows hold 16 bf16 + 8 pad: the 48-byte row stride makes the 16-lane
// groups of ds_read_b128 hit all 64 banks once.
// ---------------------------------------------------------------------------------------------
// Wave-specialised: 512 threads = 4 consumer waves (fragment reads + MFMAs, the 2 x 2 wave tiling of k_gemm_planes) +
// 4 producer waves (global loads, slice arithmetic, LDS stores).  The hardware places waves i and i+4 of a block on the
// same SIMD, so each SIMD runs one MFMA stream and one VALU/memory stream side by side instead of one wave doing both
// in turn (a 4-wave form of the same kernel spent 24 % of a wave's phase in MFMAs).
//   * producers: two register stages of lead (they hold no accumulators), loads unconditional in the steady state,
//     LDS ring of two 16-wide chunks; during iteration kc they store chunk kc+1 and refill that stage with chunk kc+3;
//   * one LDS-only block barrier per chunk.  2 blocks (16 waves) per CU, 74 KB of LDS each (where the registers allow:
//     see the launch bound below).
// (Measured and removed in round 3: the 4-wave form, a 3-chunk ring with double-buffered fragments at 1 block per CU
//  (-1.4 % on the step), the BatchNorm-backward reduction in this kernel's epilogue (its 4-byte strided reads of the
//  layer's raw input cost the contraction +4.5 ms, the pass they replace 3.3 ms).)
// ---------------------------------------------------------------------------------------------
template <int BN, bool EXTRA, bool ROWS, int NS>
// Round 6: the ROW-SET instantiations are held to 128 registers (launch bound 4 waves per SIMD = TWO blocks per CU, as this
// header always assumed): hipcc gave the 128-wide ones 152-174 - one block per CU, one MFMA wave per SIMD and nothing to
// cover its fragment reads and barriers.  At 128 the epilogue spills 28-46 dwords per lane (none in the k loop) and
// k_gemm_planes_ws<128, true, true, 3> (the paired backward) runs 360 instead of 513 us, <128, false, true, 3> 94 instead of
// 103 us; the plain-row ones measured 4 % slower that way and keep their bound (same-box rocprofv3 --stats, 9 steps).
__global__ __launch_bounds__(512, (BN == 128 && ROWS) ? 4 : 2) void k_gemm_planes_ws(GemmArgs g) {
  constexpr int KB = 16;
  typedef typename SliceFrag<NS>::type frag_t;
  constexpr int NBUF = 2, NST = 2;    // LDS ring of two chunks, two register stages of lead (three: no change, round 6)
  constexpr int AHEAD = NBUF - 1;     // the producers store chunk kc + AHEAD during iteration kc
  constexpr int WTN = BN / 2;
  constexpr int TN = WTN / 32;
  constexpr int TM = 2;
  constexpr int LDX = KB + 8;
  constexpr int APASS = BM * KB / 4 / 256;
  constexpr int AROWS = 256 / (KB / 4);
  constexpr int A_BUF = NS * BM * LDX;
  constexpr int B_BUF = NS * BN * LDX;
  constexpr int SM_WORDS = NBUF * (A_BUF + B_BUF) / 2 + (ROWS ? ROWTAB_WORDS : 0);
  __shared__ __attribute__((aligned(16))) float smem[SM_WORDS];
  __shared__ __attribute__((aligned(16))) float actco[2 * 256];        // activation on load: scale | shift of plane 0's columns
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bs = As + NBUF * A_BUF;
  int* rowtab = reinterpret_cast<int*>(smem + NBUF * (A_BUF + B_BUF) / 2);

  int mt, nt;
  if (!tile_of_block(blockIdx.x, g.ntm, g.ntn, mt, nt)) return;
  const bool in_act = g.in_scale != nullptr;                             // block-uniform
  if (in_act) {
    if (threadIdx.x < g.Ka) {
      actco[threadIdx.x] = g.in_scale[threadIdx.x];
      actco[256 + threadIdx.x] = g.in_shift[threadIdx.x];
    }
    __syncthreads();            // (LDS, not a second global load per chunk: the staging waves' vmcnt bookkeeping stays as it is)
  }
  const long m0 = (long)mt * BM;
  const int n0 = nt * BN;
  const int t = threadIdx.x;
  const bool producer = t >= 256;          // wave-uniform
  const int pt = t & 255;
  const int rs_b = ROWS ? mt / g.tps : 0;
  const int rs_i0 = ROWS ? (mt - rs_b * g.tps) * BM : 0;
  if (ROWS && t < BM) {
    const int i = rs_i0 + t;
    rowtab[t] = (i < g.nset) ? rs_b * g.V + g.ids[i] : -1;
    if (g.row_w != nullptr) rows_weights_fill(g, rowtab, i, t);
  }
  const int lane = t & 63, wave = (t >> 6) & 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int cpp = g.Ka / KB;
  const int n = g.nplanesA * cpp;           // chunks
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // two-fp16-slice mode: the A planes are staged times 2^sa, the weight arrives times 2^sb (p2m_weight_split)
  float a_sc = 1.f;
  int descale = 0;
  if (NS == 2) {
    const int sa = slice_scale_exp(*g.a_amax, g.a_bits);
    a_sc = exp2_int(sa);
    descale = -(sa + slice_scale_exp(*g.b_amax, 0));
  }

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  if (producer) {
    // LDS stores without bank conflicts.  A row of a slice image is LDX = 24 bf16 = 12 dwords; a ds_write_b64 is serviced
    // in groups of 16 consecutive lanes (4 rows x 4 k-quads, an 8-dword window per row) over 32 banks, and rows r, r+3
    // overlap (12 * 3 = 36 = 4 mod 32): 2-way conflicts on every store, 33 % of all LDS cycles of the round-2 kernel
    // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Rows {0,2,4,6} (and {1,3,5,7}) of an 8-row group start at dwords
    // 0,24,16,8 (12,4,28,20) mod 32 and tile the 32 banks exactly, so each 16-lane group takes the even or the odd rows of
    // its 8-row group.  The global loads only permute rows inside a wave's 16 rows: same 64-byte segments, same coalescing.
    const int a_rl = (pt >> 2) & 7;
    const int a_row = ((pt >> 5) << 3) | ((a_rl & 3) << 1) | (a_rl >> 2), a_k4 = (pt % (KB / 4)) * 4;
    static_assert(KB == 16, "the conflict-free row permutation assumes 4 k-quads per row");
    // Addresses = wave-uniform 64-bit base (plane + first row of this block's sample / tile, + k0 per chunk: SALU) +
    // per-thread 32-BIT byte offset inside that sample / tile (fixed for the whole kernel): the loads take the
    // `global_load v, v_off, s[base]` form and the chunk loop carries no 64-bit vector address arithmetic - the
    // contraction is issue-bound, every VALU instruction of a staging wave delays an MFMA of the wave it shares a SIMD with
    unsigned voff0[APASS], voff12[APASS];
    long sbase0, sbase12;              // element offsets of the block's first row in plane 0 / planes 1,2
    if (ROWS) {
      sbase0 = (((long)rs_b * g.V) >> g.a0_shift) * g.Ka;        // V is even whenever a0_shift = 1
      sbase12 = (g.compact ? (long)rs_b * g.nset : (long)rs_b * g.V) * g.Ka;
    } else {
      sbase0 = (m0 >> g.a0_shift) * g.Ka;                         // m0 is a multiple of 128
      sbase12 = m0 * g.Ka;
    }
#pragma unroll
    for (int ps = 0; ps < APASS; ps++) {
      int lf, lc;                        // row relative to the block's base row, in plane 0 / planes 1,2
      if (ROWS) {
        int i = rs_i0 + ps * AROWS + a_row;
        if (i >= g.nset) i = g.nset - 1;
        lf = g.ids[i];
        lc = g.compact ? i : lf;
      } else {
        long rf = m0 + ps * AROWS + a_row;
        if (rf >= g.M) rf = g.M - 1;
        lf = (int)(rf - m0);
        lc = lf;
      }
      voff0[ps] = (unsigned)(((lf >> g.a0_shift) * g.Ka + a_k4) * 4);
      voff12[ps] = (unsigned)((lc * g.Ka + a_k4) * 4);
    }
    // same for the B image: a ds_write_b128 is serviced in groups of 8 consecutive lanes (4 columns x 2 halves)
    const int b_nl = (pt >> 1) & 7;
    const int b_n = ((((pt % (BN * 2)) >> 1) >> 3) << 3) | ((b_nl & 3) << 1) | (b_nl >> 2), b_half = (pt & 1) * 8;
    const unsigned bx_voff = (unsigned)((((n0 + b_n) * 16) + b_half) * 2);       // bytes inside one slice of one chunk
    const long bx_slice = (long)g.Npad * 16;

    f32x4 ra[NST][APASS];
    u32x4 rb[NST][NS];
    auto load_chunk = [&](int kc, f32x4 (&a)[APASS], u32x4 (&b)[NS]) {
      const int p = kc / cpp;
      const int k0 = (kc - p * cpp) * KB;
      const char* Ab = reinterpret_cast<const char*>(g.A[p] + (p == 0 ? sbase0 : sbase12) + k0);     // uniform
#pragma unroll
      for (int ps = 0; ps < APASS; ps++)
#ifdef P2M_PL_ABL_NOLOAD
        a[ps] = f32x4{__uint_as_float(voff0[ps]), 1.f, 2.f, (float)(long)Ab};
#else
        a[ps] = *reinterpret_cast<const f32x4*>(Ab + (p == 0 ? voff0[ps] : voff12[ps]));
#endif
      const char* src0 = reinterpret_cast<const char*>(g.Bx + (long)((p * g.Ka + k0) >> 4) * (NS * bx_slice));   // uniform
#pragma unroll
      for (int sl = 0; sl < NS; sl++) b[sl] = *reinterpret_cast<const u32x4*>(src0 + sl * 2 * bx_slice + bx_voff);
    };
    auto store_chunk = [&](int kc, f32x4 (&a)[APASS], const u32x4 (&b)[NS]) {
      const int buf = kc % NBUF;
      unsigned short* as = As + buf * A_BUF;
#pragma unroll
      for (int ps = 0; ps < APASS; ps++) asm volatile("" : "+v"(a[ps]));
      if (in_act && kc < cpp) {                  // plane 0 (chunks 0 .. cpp - 1): BatchNorm + ReLU of the previous conv on load
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&actco[kc * KB + a_k4]);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(&actco[256 + kc * KB + a_k4]);
#pragma unroll
        for (int ps = 0; ps < APASS; ps++)
#pragma unroll
          for (int e = 0; e < 4; e++) a[ps][e] = fmaxf(fmaf(a[ps][e], sc[e], sh[e]), 0.f);
      }
#pragma unroll
      for (int ps = 0; ps < APASS; ps++) {
        u32x2 sl[NS];
#ifdef P2M_PL_ABL_NOSLICE
        for (int q = 0; q < NS; q++) sl[q] = u32x2{__float_as_uint(a[ps][q & 1]), __float_as_uint(a[ps][2 + (q & 1)])};
#else
        split_pack4<NS>(a[ps][0], a[ps][1], a[ps][2], a[ps][3], a_sc, sl);
#endif
        unsigned short* d = as + (ps * AROWS + a_row) * LDX + a_k4;
#pragma unroll
        for (int q = 0; q < NS; q++) *reinterpret_cast<u32x2*>(d + q * BM * LDX) = sl[q];
      }
      unsigned short* d = Bs + buf * B_BUF + b_n * LDX + b_half;
#pragma unroll
      for (int q = 0; q < NS; q++) *reinterpret_cast<u32x4*>(d + q * BN * LDX) = b[q];
    };
    const int last = n - 1;
    // chunk c lives in register stage c % NST and LDS buffer c % NBUF.  Prologue: chunks 0..AHEAD-1 stored, the next NST
    // chunks in flight.
#pragma unroll
    for (int i = 0; i < NST; i++) load_chunk(i < last ? i : last, ra[i], rb[i]);
#pragma unroll
    for (int i = 0; i < AHEAD; i++) {
      if (i < n) store_chunk(i, ra[i], rb[i]);
      load_chunk(NST + i < last ? NST + i : last, ra[i], rb[i]);
    }
    lds_barrier();
    // iteration kc (consumers: MFMAs of chunk kc): store chunk kc+AHEAD, refill its stage with chunk kc+AHEAD+NST
    int kc = 0;
    for (; kc + (NST - 1) + AHEAD + NST < n; kc += NST) {     // steady state: every load below is in range
#pragma unroll
      for (int i = 0; i < NST; i++) {
        store_chunk(kc + i + AHEAD, ra[(i + AHEAD) % NST], rb[(i + AHEAD) % NST]);
        load_chunk(kc + i + AHEAD + NST, ra[(i + AHEAD) % NST], rb[(i + AHEAD) % NST]);
        lds_barrier();
      }
    }
    for (; kc < n; kc += NST) {                                // tail: same rotation, range-checked
#pragma unroll
      for (int i = 0; i < NST; i++) {
        if (kc + i < n) {
          if (kc + i + AHEAD < n) store_chunk(kc + i + AHEAD, ra[(i + AHEAD) % NST], rb[(i + AHEAD) % NST]);
          if (kc + i + AHEAD + NST < n) load_chunk(kc + i + AHEAD + NST, ra[(i + AHEAD) % NST], rb[(i + AHEAD) % NST]);
          lds_barrier();
        }
      }
    }
  } else {
    frag_t fa[NBUF - 1][NS][TM], fb[NBUF - 1][NS][TN];       // [register set][slice, high first][tile]
    auto read_frags = [&](int kc, frag_t (&a)[NS][TM], frag_t (&b)[NS][TN]) {
      const int buf = kc % NBUF;
      const unsigned short* as = As + buf * A_BUF + (wm * 64 + l31) * LDX + lhi * 8;
      const unsigned short* bs = Bs + buf * B_BUF + (wn * WTN + l31) * LDX + lhi * 8;
#pragma unroll
      for (int sl = 0; sl < NS; sl++) {
#pragma unroll
        for (int i = 0; i < TM; i++)
          a[sl][i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(as + (sl * BM + i * 32) * LDX));
#pragma unroll
        for (int j = 0; j < TN; j++)
          b[sl][j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(bs + (sl * BN + j * 32) * LDX));
      }
    };
    auto mfmas = [&](const frag_t (&a)[NS][TM], const frag_t (&b)[NS][TN]) {
#define P2M_PAIR(SA, SB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) \
      acc[i][j] = slice_mfma<NS>(a[SA][i], b[SB][j], acc[i][j]);
#ifdef P2M_PL_ABL_NOMFMA
      if constexpr (NS == 3) {
        P2M_PAIR(0, 0)
        for (int sl = 1; sl < NS; sl++)
          for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) { asm volatile("" :: "v"(a[sl][i])); asm volatile("" :: "v"(b[sl][j])); }
      } else
#endif
      if constexpr (NS == 3) {          // smallest products first
        P2M_PAIR(2, 0)
        P2M_PAIR(0, 2)
        P2M_PAIR(1, 1)
        P2M_PAIR(1, 0)
        P2M_PAIR(0, 1)
        P2M_PAIR(0, 0)
      } else {
        P2M_PAIR(1, 0)
        P2M_PAIR(0, 1)
        P2M_PAIR(0, 0)
      }
#undef P2M_PAIR
    };
    lds_barrier();                              // the first AHEAD chunks are in LDS
    for (int kc = 0; kc < n; kc++) {
      read_frags(kc, fa[0], fb[0]);
      mfmas(fa[0], fb[0]);
      lds_barrier();
    }
  }
  __syncthreads();   // staging buffers are free: the epilogue reuses them
  if (!producer) {
    gemm_epilogue<BN, EXTRA, ROWS>(g, acc, smem, rowtab, mt, m0, n0, rs_i0, wm, wn, l31, lhi, descale);
  } else {
    if (g.stats != nullptr) {                   // the three block barriers of the statistics reduction
      __syncthreads();
      __syncthreads();
      __syncthreads();
    }
  }
}

// Bx[k / 16][s][n][k % 16] = s-th bf16 slice of Bm[k][n] (zero for N <= n < Npad): the pre-split weight operand,
// chunk-major so that the [BN x 16] slice a block stages per chunk is one contiguous run
// NS = 2: fp16 slices of Bm 2^sb, sb from the bound 2^bits * (*amax_in) - the word trailing the image itself (bits = 0,
// written by k_amax_one_block just before) or the word of the tensor Bm was derived from; the trailer then receives
// that bound, so every consumer re-derives the same sb from the image alone
template <int NS>
__global__ void k_weight_split(const float* __restrict__ Bm, unsigned short* __restrict__ Bx, int K, int N, int Npad,
                               const unsigned* __restrict__ amax_in, int bits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)Npad * K) return;
  const int k = (int)(i / Npad), n = (int)(i - (long)k * Npad);      // n fastest: coalesced reads of Bm
  const long sl = (long)Npad * 16;
  unsigned short* d = Bx + (long)(k >> 4) * NS * sl + (long)n * 16 + (k & 15);
  if constexpr (NS == 3) {
    unsigned h = 0, m = 0, l = 0;
    if (n < N) split3(Bm[(long)k * N + n], h, m, l);
    d[0] = (unsigned short)(h >> 16);
    d[sl] = (unsigned short)(m >> 16);
    d[2 * sl] = (unsigned short)(l >> 16);
  } else {
    unsigned* trailer = reinterpret_cast<unsigned*>(Bx + (long)NS * Npad * K);
    const unsigned word = *amax_in;
    if (i == 0 && amax_in != trailer) *trailer = word == 0u ? 0u : __float_as_uint(__builtin_ldexpf(__uint_as_float(word), bits));
    const float y = n < N ? Bm[(long)k * N + n] * exp2_int(slice_scale_exp(word, bits)) : 0.f;
    const _Float16 h = (_Float16)y;
    const _Float16 l = (_Float16)(y - (float)h);
    d[0] = __builtin_bit_cast(unsigned short, h);
    d[sl] = __builtin_bit_cast(unsigned short, l);
  }
}

// amax words (p2m_split.h).  One block, overwrites the word: for small tensors (weights) - no zeroing, no atomics.
__global__ __launch_bounds__(1024) void k_amax_one_block(const float* __restrict__ x, long n, unsigned* __restrict__ word) {
  __shared__ float red[16];
  float m = 0.f;
  const long n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n >> 2 : 0;      // float4 body, 8 loads in flight per thread
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  long i = threadIdx.x;
  for (; i + 7 * 1024 < n4; i += 8 * 1024) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = x4[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; u++)
      m = fmaxf(fmaxf(m, fmaxf(amax_abs(v[u][0]), amax_abs(v[u][1]))), fmaxf(amax_abs(v[u][2]), amax_abs(v[u][3])));
  }
  for (; i < n4; i += 1024) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(amax_abs(v[0]), amax_abs(v[1]))), fmaxf(amax_abs(v[2]), amax_abs(v[3])));
  }
  for (long j = n4 * 4 + threadIdx.x; j < n; j += 1024) m = fmaxf(m, amax_abs(x[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) m = fmaxf(m, red[w]);
    *word = __float_as_uint(m);
  }
}
// Any size and alignment: atomic max into a word the caller has zeroed.  `head` scalars up to the first 16-byte boundary,
// n4 float4, `tail` scalars (block 0 takes the scalars).
// ONE commit per block (the block's maximum through LDS): the word starts at zero, so the first wave of every block sees a
// value it can raise - with a commit per wave a 6 MB tensor issued ~6 000 same-address atomics and took 95 us (round 4 trace:
// 64 GB/s); four loads in flight per thread.
__device__ __forceinline__ void amax_commit_block(unsigned* word, float m) {
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
__global__ __launch_bounds__(256) void k_amax(const float* __restrict__ x, int head, long n4, int tail,
                                              unsigned* __restrict__ word) {
  float m = 0.f;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x + head);
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const f32x4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    auto mx = [](const f32x4& v) {
      return fmaxf(fmaxf(amax_abs(v[0]), amax_abs(v[1])), fmaxf(amax_abs(v[2]), amax_abs(v[3])));
    };
    m = fmaxf(m, fmaxf(fmaxf(mx(v0), mx(v1)), fmaxf(mx(v2), mx(v3))));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(amax_abs(v[0]), amax_abs(v[1]))), fmaxf(amax_abs(v[2]), amax_abs(v[3])));
  }
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < head) m = fmaxf(m, amax_abs(x[threadIdx.x]));
    if ((int)threadIdx.x < tail) m = fmaxf(m, amax_abs(x[head + n4 * 4 + threadIdx.x]));
  }
  amax_commit_block(word, m);
}
// The rows of a row set only (the others may hold no data): row (b, i) -> b * V + ids[i], F % 4 == 0
__global__ __launch_bounds__(256) void k_amax_rows(const float* __restrict__ x, RowSet rs, int B, int F,
                                                   unsigned* __restrict__ word) {
  const int f4 = F >> 2;
  const long tot = (long)B * rs.n * f4;
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const long row = i / f4;
    const int c = (int)(i - row * f4);
    const int b = (int)(row / rs.n), r = (int)(row - (long)b * rs.n);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((long)b * rs.V + (rs.ids ? rs.ids[r] : r)) * F + c * 4);
    m = fmaxf(fmaxf(m, fmaxf(amax_abs(v[0]), amax_abs(v[1]))), fmaxf(amax_abs(v[2]), amax_abs(v[3])));
  }
  amax_commit_block(word, m);
}

// scalar fall-back (first conv Fin=5 -> K=15; last conv Fout=3): one thread per (row, n)
__global__ void k_naive_gemm_planes(GemmArgs g) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.M * g.N) return;
  long r = idx / g.N;
  int n = (int)(idx - r * g.N);
  float acc = 0.f;
  for (int p = 0; p < g.nplanesA; p++) {
    const float* a = g.A[p] + (r >> (p == 0 ? g.a0_shift : 0)) * g.Ka;
    const float* b = g.Bm + (long)p * g.Ka * g.N + n;
    for (int k = 0; k < g.Ka; k++) acc = fmaf(a[k], b[(long)k * g.N], acc);
  }
  if (g.bias) acc += g.bias[n];
  if (g.act_scale) acc = fmaf(acc, g.act_scale[n], g.act_shift[n]);
  if (g.act_relu) acc = fmaxf(acc, 0.f);
  if (g.addend) acc += g.addend[r * g.N + n];
  int q = n / g.Nc;
  g.C[q][r * g.Nc + (n - q * g.Nc)] = acc;
}

// per-128-row-tile statistics for the naive path (same partial format as the MFMA epilogue)
__global__ void k_naive_tile_stats(const float* __restrict__ Y, float* __restrict__ stats, long M, int N) {
  int tile = blockIdx.x;
  long r0 = (long)tile * BM;
  long r1 = r0 + BM < M ? r0 + BM : M;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    float s = 0.f;
    for (long r = r0; r < r1; r++) s += Y[r * N + n];
    float mean = s / (float)(r1 - r0);
    float m2 = 0.f;
    for (long r = r0; r < r1; r++) {
      float d = Y[r * N + n] - mean;
      m2 += d * d;
    }
    stats[(long)tile * 2 * N + n] = s;
    stats[(long)tile * 2 * N + N + n] = m2;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient: P[chunk][kk][n] = sum_{r in chunk} Z[r][kk] * G[r][n]
// ---------------------------------------------------------------------------------------------
struct TnArgs {
  const float* A[3];
  const float* G[3];   // nplanesG column planes of width Gc (N = nplanesG * Gc)
  int Gc;
  float* P;
  float* Pdb;
  long M, chunk_rows;
  int nplanesA, Ka, a0_shift, Ktot, N;
  int nkt, ntn;
  // row-set mode: chunk = (sample b, split s) over logical rows i < nset; A and G plane 0 are read at the actual row
  // b*V + ids[i], G planes 1,2 at the compact row b*nset + i when `compact`
  const int* ids;
  int nset, V, splits, compact;
  // k_gemm_tn_ws, row-set mode, splits = 1: a chunk is `spc` consecutive WHOLE samples (the last chunk: what is left of the B
  // samples) - fewer partials for the unpack to read where a sample has few rows and the gradient many tiles
  int spc = 1, B = 0;
  // two-fp16-slice mode: amax words of the A planes and of the G planes (+ binades of headroom), p2m_split.h
  const unsigned* a_amax;
  const unsigned* g_amax;
  int a_bits, g_bits;
  // activation on load of A plane 0 (k_gemm_tn_ws): the operand is max(fma(A, a_scale[k], a_shift[k]), 0); or null
  const float* a_scale;
  const float* a_shift;
  int nchunks;           // k_gemm_tn_ws: one-dimensional grid of ceil(nchunks / 8) * 8 * nkt * ntn blocks (XCD-aware mapping)
  int accum = 0;         // P += result instead of P = result (single chunk only: a plain Linear's weight gradient added
                         // straight into the parameter's .grad, p2m_gemm_tn_acc)
};

template <int BN, bool ROWS = false>
__global__ __launch_bounds__(256) void k_gemm_tn(TnArgs g) {
  constexpr int WTN = BN / 2;
  constexpr int TN = WTN / 32;
  constexpr int TM = 2;
  constexpr int RK = 32;                 // rows (reduction) per LDS stage
  constexpr int GPASS = BN / 32;         // float4 loads of G per thread per stage
  constexpr int GROWS = 256 / (BN / 4);
  __shared__ float smem[2 * RK * BM + 2 * RK * BN];
  float* As = smem;                      // [buf][r][kk]  (kk contiguous)
  float* Gs = smem + 2 * RK * BM;        // [buf][r][n]

  const int tile = blockIdx.x;
  const int kt = tile / g.ntn, nt = tile % g.ntn;
  const int chunk = blockIdx.y;
  const int kk0 = kt * BM, n0 = nt * BN;
  // flat mode: rows [r_begin, r_end) of the flattened (B*V) dimension; ROWS mode: logical rows of one sample
  const int rs_b = ROWS ? chunk / g.splits : 0;
  const long r_begin = ROWS ? (long)(chunk - rs_b * g.splits) * g.chunk_rows : (long)chunk * g.chunk_rows;
  long r_end = r_begin + g.chunk_rows;
  if (r_end > (ROWS ? (long)g.nset : g.M)) r_end = ROWS ? (long)g.nset : g.M;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // this thread's column of the A tile: fixed plane / offset for the whole kernel
  const int a_r = t >> 5, a_c4 = (t & 31) * 4;
  const int kk = kk0 + a_c4;
  const bool a_ok = kk < g.Ktot;
  const int ap = a_ok ? kk / g.Ka : 0;
  const int ak = kk - ap * g.Ka;
  const float* Ap = g.A[ap];
  const int ash = (ap == 0) ? g.a0_shift : 0;
  const int g_r = t / (BN / 4), g_c4 = (t % (BN / 4)) * 4;
  const bool g_ok = (n0 + g_c4) < g.N;
  const int gq = g_ok ? (n0 + g_c4) / g.Gc : 0;
  const int gcol = (n0 + g_c4) - gq * g.Gc;
  const float* Gp = g.G[gq];

  float4 ra[4], rg[GPASS];
  float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
  // ROWS: vertex ids of the rows this thread loads, fetched ONE STAGE AHEAD of the data loads that depend on them
  int ida[ROWS ? 4 : 1], idg[ROWS ? GPASS : 1];
  auto load_ids = [&](long r0) {
    if (!ROWS) return;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long r = r0 + ps * 8 + a_r;
      ida[ps] = (r < r_end) ? g.ids[r] : 0;
    }
#pragma unroll
    for (int ps = 0; ps < GPASS; ps++) {
      const long r = r0 + ps * GROWS + g_r;
      idg[ps] = (r < r_end) ? g.ids[r] : 0;
    }
  };
  auto load_stage = [&](long r0) {
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      long r = r0 + ps * 8 + a_r;
      if (a_ok && r < r_end) {
        if (ROWS) r = (long)rs_b * g.V + ida[ps];
        ra[ps] = *reinterpret_cast<const float4*>(Ap + (r >> ash) * g.Ka + ak);
      } else {
        ra[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ps = 0; ps < GPASS; ps++) {
      long r = r0 + ps * GROWS + g_r;
      if (g_ok && r < r_end) {
        if (ROWS) r = (gq == 0 || !g.compact) ? (long)rs_b * g.V + idg[ps] : (long)rs_b * g.nset + r;
        rg[ps] = *reinterpret_cast<const float4*>(Gp + r * g.Gc + gcol);
      } else {
        rg[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      dbs.x += rg[ps].x; dbs.y += rg[ps].y; dbs.z += rg[ps].z; dbs.w += rg[ps].w;
    }
    load_ids(r0 + RK);
  };
  auto store_stage = [&](int buf) {
    float* as = As + buf * RK * BM;
#pragma unroll
    for (int ps = 0; ps < 4; ps++)
      *reinterpret_cast<float4*>(as + (ps * 8 + a_r) * BM + a_c4) = ra[ps];
    float* gs = Gs + buf * RK * BN;
#pragma unroll
    for (int ps = 0; ps < GPASS; ps++)
      *reinterpret_cast<float4*>(gs + (ps * GROWS + g_r) * BN + g_c4) = rg[ps];
  };

  load_ids(r_begin);
  load_stage(r_begin);
  store_stage(0);
  __syncthreads();
  int it = 0;
  for (long r0 = r_begin; r0 < r_end; r0 += RK, it++) {
    const int cur = it & 1;
    const bool more = (r0 + RK) < r_end;
    if (more) load_stage(r0 + RK);
    const float* as = As + cur * RK * BM + lhi * BM + wm * 64 + l31;
    const float* gs = Gs + cur * RK * BN + lhi * BN + wn * WTN + l31;
#pragma unroll
    for (int ks = 0; ks < RK / 2; ks++) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = as[2 * ks * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; j++) b[j] = gs[2 * ks * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(cur ^ 1);
    __syncthreads();
  }

  float* Pc = g.P + (long)chunk * g.Ktot * g.N;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int krow = kk0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (krow < g.Ktot && n < g.N) {
          float* d = Pc + (long)krow * g.N + n;
          *d = g.accum ? *d + acc[i][j][r] : acc[i][j][r];
        }
      }
    }
  if (kt == 0 && g.Pdb != nullptr) {
    // bias gradient: column sums of G over this chunk (reduce the GROWS row groups through LDS)
    float* red = smem;  // [GROWS][BN]
    *reinterpret_cast<float4*>(red + g_r * BN + g_c4) = dbs;
    __syncthreads();
    if (t < BN) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < GROWS; q++) s += red[q * BN + t];
      if (n0 + t < g.N) g.Pdb[(long)chunk * g.N + n0 + t] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient on the BF16 matrix pipe (same 3-slice split as k_gemm_planes_ws).
//
// Both operands are activations stored row-major over the REDUCTION index r ([r][kk], [r][n]), while the MFMA wants
// each lane to hold 8 consecutive r of one kk / n.  The transpose happens in registers on the way into LDS: a staging
// thread owns a 4 (r) x 4 (kk) block (four float4 loads, rows r..r+3), cuts it into slices and writes, per kk, the
// four r of a slice as ONE 8-byte store into the kk-major image [slice][kk][16 r + 8 pad].  Staging waves 4,5 stage A,
// waves 6,7 stage G (wave-uniform roles, identical instruction stream); waves 0..3 run the MFMAs.
// Loads are unconditional (rows past the end are clamped and their values zeroed; columns past Ktot / N are clamped
// and never stored), with two register stages of lead and an LDS-only barrier, as in k_gemm_planes_ws.  In ROWS mode
// the vertex ids of a stage are fetched two phases before the data loads that depend on them (g.ids is zero-padded
// by 64 entries at bake time, so those 16-byte id loads need no bounds).
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef P2M_TN_TRACE
// Probe build only (tools/tn_trace.sh, tools/probes/tn_trace_probe.py): s_memtime stamps of every wave of one block at the
// phase boundaries of 32 steady-state stages.  [wave][stage - P2M_TN_TRACE_S0][event]; events: 0 past the barrier,
// 1 operands there (staging: global loads landed; MFMA: fragments read), 2 work issued (LDS stores / last MFMA), 3 at the barrier
__device__ unsigned long long g_tn_tr[8][32][4];
#ifndef P2M_TN_TRACE_S0
#define P2M_TN_TRACE_S0 40
#endif
#define P2M_TNT(s, ev)                                                                                          \
  do {                                                                                                          \
    if (tnt_on && (s) >= P2M_TN_TRACE_S0 && (s) < P2M_TN_TRACE_S0 + 32)                                         \
      g_tn_tr[threadIdx.x >> 6][(s) - P2M_TN_TRACE_S0][ev] = __builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define P2M_TNT(s, ev) do { } while (0)
#endif

// Wave-specialised (4 MFMA waves + 4 staging waves, one LDS-only barrier per 16-row stage, two register stages of lead);
// see k_gemm_planes_ws.
template <int BN, bool ROWS, int NS>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_ws(TnArgs g) {
  typedef typename SliceFrag<NS>::type frag_t;
  constexpr int RK = 16;                 // rows (reduction) per stage = one bf16 MFMA k-step
  constexpr int LDX = RK + 8;            // bf16 per LDS row: 48-byte stride, conflict-free ds_read_b128
  constexpr int WTN = BN / 2;
  constexpr int TN = WTN / 32;
  constexpr int TM = 2;
  constexpr int A_BUF = NS * BM * LDX;
  constexpr int G_BUF = NS * BN * LDX;
  __shared__ __attribute__((aligned(16))) float smem[(2 * A_BUF + 2 * G_BUF) / 2];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Gs = As + 2 * A_BUF;

  // The n-tiles of one row chunk read the SAME rows of A: give them consecutive slots of one XCD (observed dispatch: block
  // b runs on XCD b % 8), so that A comes from HBM once and from that XCD's L2 for the others.  With the chunk in
  // blockIdx.y the three n-tiles of a 128 x 384 weight gradient sat on three different XCDs (PMC: 1.9x the algorithmic
  // bytes per launch; 1 408 -> 884 MB on the finest level, 144 -> 131 GB per train step).  +0.4 % on the step.
  // Fewer than 8 chunks (a plain Linear's weight gradient over a batch of rows: ONE chunk, p2m_gemm_tn_acc): that mapping
  // would put every live block on the XCDs 0 .. nchunks - 1 (measured: one chunk of 1024 tiles ran on 32 of the 256 CUs,
  // 740 us for 8.6 GFLOP); consecutive blocks then simply take consecutive tiles.
  const int ntiles = g.nkt * g.ntn;
  int tile, chunk;
  if (g.nchunks >= 8) {
    const int slot = blockIdx.x >> 3;
    tile = slot % ntiles;
    chunk = (slot / ntiles) * 8 + (blockIdx.x & 7);
  } else {
    tile = blockIdx.x % ntiles;
    chunk = blockIdx.x / ntiles;
  }
  if (chunk >= g.nchunks) return;
  const int kt = tile / g.ntn, nt = tile % g.ntn;
  const int kk0 = kt * BM, n0 = nt * BN;
  const int spc = ROWS ? g.spc : 1;
  const int rs_b = ROWS ? (spc > 1 ? chunk * spc : chunk / g.splits) : 0;      // (first) sample of this chunk
  const int nsamp = (ROWS && spc > 1) ? (g.B - rs_b < spc ? g.B - rs_b : spc) : 1;
  const long r_begin = ROWS ? (spc > 1 ? 0L : (long)(chunk - rs_b * g.splits) * g.chunk_rows) : (long)chunk * g.chunk_rows;
  long r_end = r_begin + g.chunk_rows;
  if (r_end > (ROWS ? (long)g.nset : g.M)) r_end = ROWS ? (long)g.nset : g.M;
  const int nst = r_end > r_begin ? (int)((r_end - r_begin + RK - 1) / RK) : 0;

  const int t = threadIdx.x;
  const bool producer = t >= 256;          // waves 4..7 stage, waves 0..3 run the MFMAs (k_gemm_planes_ws)
#ifdef P2M_TN_TRACE
  const bool tnt_on = blockIdx.x == P2M_TN_TRACE && (t & 63) == 0 && BN == 128 && ROWS;
#endif
  const int pt = t & 255;
  const int lane = t & 63, wave = (t >> 6) & 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // staging role of this thread: a 4-row x 4-column block of the A tile (pt < 128) or of the G tile.  The role is
  // wave-uniform (waves 4,5 stage A, waves 6,7 stage G): readfirstlane tells the compiler so, and the role-dependent
  // code becomes scalar branches instead of exec-masked ones
  const bool is_a = __builtin_amdgcn_readfirstlane((int)(pt < 128)) != 0;
  const int st = pt & 127;
  const int rq = st & 3;                                   // row quad of the 16-row stage
  const int c4 = is_a ? (st >> 2) : (st >> 2) % (BN / 4);  // column quad (BN = 64: the upper threads duplicate)
  const float* src;       // plane base + column offset
  int pitch, shift;
  bool compact_rows;      // ROWS: read at the compact row b*nset + r instead of b*V + ids[r]
  unsigned short* dst;    // LDS image (buffer 0, slice 0) of this thread's first column, at its row quad
  int slice_stride;
  {
    if (is_a) {
      int kk = kk0 + c4 * 4;
      if (kk >= g.Ktot) kk = 0;                            // clamped: those rows of P are never stored
      const int ap = kk / g.Ka;
      src = g.A[ap] + (kk - ap * g.Ka);
      pitch = g.Ka;
      shift = (ap == 0) ? g.a0_shift : 0;
      compact_rows = false;
      dst = As + (c4 >> 2) * 16 * LDX + (c4 & 3) * 2 * LDX + rq * 4;
      slice_stride = BM * LDX;
    } else {
      int n = n0 + c4 * 4;
      if (n >= g.N) n = 0;
      const int gq = n / g.Gc;
      src = g.G[gq] + (n - gq * g.Gc);
      pitch = g.Gc;
      shift = 0;
      compact_rows = ROWS && gq != 0 && g.compact;
      dst = Gs + (c4 >> 2) * 16 * LDX + (c4 & 3) * 2 * LDX + rq * 4;
      slice_stride = BN * LDX;
    }
  }
  const int buf_stride = is_a ? A_BUF : G_BUF;
  // activation on load (A staging waves only; wave-uniform): the coefficients of this thread's four columns
  const bool a_act = __builtin_amdgcn_readfirstlane((int)(is_a && g.a_scale != nullptr)) != 0;
  f32x4 act_sc = {1.f, 1.f, 1.f, 1.f}, act_sh = {0.f, 0.f, 0.f, 0.f};
  if (a_act) {
    int kk = kk0 + c4 * 4;
    if (kk >= g.Ktot) kk = 0;
    act_sc = *reinterpret_cast<const f32x4*>(g.a_scale + kk);
    act_sh = *reinterpret_cast<const f32x4*>(g.a_shift + kk);
  }
  float my_sc = 1.f;                     // two-fp16-slice mode: this staging wave's operand is staged times 2^s
  int descale = 0;
  if (NS == 2) {
    const int sa = slice_scale_exp(*g.a_amax, g.a_bits), sg = slice_scale_exp(*g.g_amax, g.g_bits);
    my_sc = exp2_int(is_a ? sa : sg);
    descale = -(sa + sg);
  }
  const int* idp = ROWS ? g.ids + r_begin + rq * 4 : nullptr;
  // Addresses: per-thread 64-bit base (plane + column + the first row of this block's sample / chunk), fixed for the
  // whole kernel, plus a 32-bit byte offset per load = (row relative to that base >> shift) * pitch * 4: three 32-bit
  // VALU instructions and one 64-bit add per load instead of the 64-bit multiply-add chain.  base rows are even
  // (V is even whenever shift = 1; chunk_rows is a multiple of 16), so (base + rel) >> shift == (base >> shift) + (rel >> shift).
  const long base_row = ROWS ? (compact_rows ? (long)rs_b * g.nset + r_begin : (long)rs_b * g.V) : r_begin;
  const char* srcb = reinterpret_cast<const char*>(src + (base_row >> shift) * pitch);
  const long samp_bytes = ROWS ? (((long)(compact_rows ? g.nset : g.V) >> shift) * pitch) * 4 : 0;   // sample to sample
  const unsigned pitch4 = (unsigned)pitch * 4u;
  const int nrows_m1 = (int)(r_end - r_begin) - 1;

  f32x4 x0[4], x1[4];       // two register stages (native vector types: no scratch)
  i32x4 id0 = {0, 0, 0, 0}, id1 = {0, 0, 0, 0};
  f32x4 dbs = {0.f, 0.f, 0.f, 0.f};

  auto load_ids = [&](int kc, i32x4& id) {
    if (ROWS) id = *reinterpret_cast<const i32x4*>(idp + (long)kc * RK);
  };
  // CLAMP = false: every row of the stage exists (steady state); true: rows past the end are clamped to the last one
  auto load_stage = [&](int kc, f32x4 (&x)[4], const i32x4& id, auto clamp_tag) {
    constexpr bool CLAMP = decltype(clamp_tag)::value;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      int rel = kc * RK + rq * 4 + ps;                     // row relative to r_begin
      if (CLAMP && rel > nrows_m1) rel = nrows_m1;
      if (ROWS && !compact_rows) rel = id[ps];             // vertex id inside the sample (slack entries: vertex 0)
      const unsigned off = __umul24((unsigned)(rel >> shift), pitch4);
#ifdef P2M_TN_ABL_NOLOAD
      x[ps] = f32x4{__uint_as_float(off), 1.f, 2.f, 3.f};
#else                                   // (non-temporal loads - `nt` - of G, or of both operands: +6.5 % / +5 %, round 6)
      x[ps] = *reinterpret_cast<const f32x4*>(srcb + off);
#endif
    }
  };
  // TAIL = true: rows past the end of the chunk are zeroed before they are used
  auto store_stage = [&](int buf, int kc, f32x4 (&x)[4], auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) asm volatile("" : "+v"(x[ps]));     // see k_gemm_planes_bx
    P2M_TNT(kc, 1);
    if (a_act) {                                                        // (before the tail's zeroing: act(0) != 0)
#pragma unroll
      for (int ps = 0; ps < 4; ps++)
#pragma unroll
        for (int e = 0; e < 4; e++) x[ps][e] = fmaxf(fmaf(x[ps][e], act_sc[e], act_sh[e]), 0.f);
    }
    if (TAIL) {
      const int rlast = nrows_m1 + 1 - (kc * RK + rq * 4);             // rows of this quad that exist
#pragma unroll
      for (int ps = 0; ps < 4; ps++)
        if (ps >= rlast) x[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!is_a) {                                                        // bias gradient: column sums of G (scalar branch)
#pragma unroll
      for (int ps = 0; ps < 4; ps++) dbs += x[ps];
    }
    unsigned short* d = dst + buf * buf_stride;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      u32x2 sl[NS];
#ifdef P2M_TN_ABL_NOSLICE
      for (int q = 0; q < NS; q++) sl[q] = u32x2{__float_as_uint(x[q & 1][e]), __float_as_uint(x[2 + (q & 1)][e])};
#else
      split_pack4<NS>(x[0][e], x[1][e], x[2][e], x[3][e], my_sc, sl);
#endif
      // LDS rows are PERMUTED inside every block of 16: column 4 c + e of the block lives in row 2 c + (e & 1) + 8 (e >> 1).
      // The 16 lanes of a ds_write_b64 group are 4 column quads x 4 row quads: with the natural order their rows are 4
      // apart = 48 dwords = 16 banks, i.e. 2-way conflicts on every store (PMC: 33 % of the kernel's LDS cycles); rows
      // 2 apart are 24 dwords: four distinct 8-dword windows.  The MFMA waves read a block's 16 rows as a set either way.
#pragma unroll
      for (int q = 0; q < NS; q++)
        *reinterpret_cast<u32x2*>(d + ((e & 1) + 8 * (e >> 1)) * LDX + q * slice_stride) = sl[q];
    }
    P2M_TNT(kc, 2);
  };
  using std::false_type;
  using std::true_type;
  auto compute = [&](int cur, int kc) {
    // (row permutation of the staging stores: logical row j of a 16-block -> 2 (j >> 2) + (j & 1) + 8 ((j & 3) >> 1))
    const int lrow = (l31 & 16) + 2 * ((l31 & 15) >> 2) + (l31 & 1) + 8 * ((l31 & 3) >> 1);
    const unsigned short* as = As + cur * A_BUF + (wm * 64 + lrow) * LDX + lhi * 8;
    const unsigned short* gs = Gs + cur * G_BUF + (wn * WTN + lrow) * LDX + lhi * 8;
    frag_t fa[NS][TM], fb[NS][TN];       // [slice, high first][tile]
#pragma unroll
    for (int sl = 0; sl < NS; sl++) {
#pragma unroll
      for (int i = 0; i < TM; i++)
        fa[sl][i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(as + (sl * BM + i * 32) * LDX));
#pragma unroll
      for (int j = 0; j < TN; j++)
        fb[sl][j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(gs + (sl * BN + j * 32) * LDX));
    }
#ifdef P2M_TN_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    P2M_TNT(kc, 1);
#endif
#define P2M_PAIR(SA, SB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) \
      acc[i][j] = slice_mfma<NS>(fa[SA][i], fb[SB][j], acc[i][j]);
#ifdef P2M_TN_ABL_NOMFMA
    if constexpr (NS == 3) {
      P2M_PAIR(0, 0)
      for (int sl = 1; sl < NS; sl++)
        for (int i = 0; i < TM; i++) for (int j = 0; j < TN; j++) { asm volatile("" :: "v"(fa[sl][i])); asm volatile("" :: "v"(fb[sl][j])); }
    } else
#endif
    if constexpr (NS == 3) {            // smallest products first
      P2M_PAIR(2, 0)
      P2M_PAIR(0, 2)
      P2M_PAIR(1, 1)
      P2M_PAIR(1, 0)
      P2M_PAIR(0, 1)
      P2M_PAIR(0, 0)
    } else {
      P2M_PAIR(1, 0)
      P2M_PAIR(0, 1)
      P2M_PAIR(0, 0)
    }
#undef P2M_PAIR
    P2M_TNT(kc, 2);
    (void)kc;
  };
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  for (int sb = 0; sb < nsamp; sb++) {      // (one pass unless the chunk is several whole samples: same stages, next sample)
  if (nst > 0) {
    if (producer) {
      // stage s lives in register set s & 1 and LDS buffer s & 1; ids of stage s in id set s & 1.  During iteration
      // kc (consumers: MFMAs of stage kc) the producers store stage kc+1 and refill its set with stage kc+3.
      load_ids(0, id0);
      load_ids(1, id1);
      load_stage(0, x0, id0, true_type{});
      load_ids(2, id0);
      load_stage(nst > 1 ? 1 : 0, x1, id1, true_type{});
      load_ids(3, id1);
      store_stage(0, 0, x0, true_type{});
      load_stage(nst > 2 ? 2 : nst - 1, x0, id0, true_type{});
      load_ids(4, id0);
      lds_barrier();
      int kc = 0;
      // steady state: stages up to kc+4 are FULL stages (the possibly partial last stage nst-1 is left to the tail), so
      // neither the loads nor the stores carry clamps, zeroing or conditions
      for (; kc + 5 < nst; kc += 2) {
        P2M_TNT(kc + 1, 0);
        store_stage(1, kc + 1, x1, false_type{});
        load_stage(kc + 3, x1, id1, false_type{});
        load_ids(kc + 5, id1);
        P2M_TNT(kc + 1, 3);
        lds_barrier();
        P2M_TNT(kc + 2, 0);
        store_stage(0, kc + 2, x0, false_type{});
        load_stage(kc + 4, x0, id0, false_type{});
        load_ids(kc + 6, id0);
        P2M_TNT(kc + 2, 3);
        lds_barrier();
      }
      for (; kc < nst; kc += 2) {                // tail: same rotation, range-checked
        if (kc + 1 < nst) store_stage(1, kc + 1, x1, true_type{});
        if (kc + 3 < nst) {
          load_stage(kc + 3, x1, id1, true_type{});
          load_ids(kc + 5, id1);
        }
        lds_barrier();
        if (kc + 1 < nst) {
          if (kc + 2 < nst) store_stage(0, kc + 2, x0, true_type{});
          if (kc + 4 < nst) {
            load_stage(kc + 4, x0, id0, true_type{});
            load_ids(kc + 6, id0);
          }
          lds_barrier();
        }
      }
    } else {
      lds_barrier();                             // stage 0 is in LDS
      for (int kc = 0; kc < nst; kc++) {
        P2M_TNT(kc, 0);
        compute(kc & 1, kc);
        P2M_TNT(kc, 3);
        lds_barrier();
      }
    }
  }
  srcb += samp_bytes;
  }
  __syncthreads();

  float* Pc = g.P + (long)chunk * g.Ktot * g.N;
  if (!producer) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * WTN + j * 32 + l31;
        // accumulate form (p2m_gemm_tn_acc): the 16 old values of the tile first, all loads in flight, then add and store
        // (one load - add - store chain per element waited for every load in turn: 116 us for a 4096 x 4096 gradient)
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int krow = kk0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          old[r] = (g.accum && krow < g.Ktot && n < g.N) ? Pc[(long)krow * g.N + n] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int krow = kk0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (krow < g.Ktot && n < g.N) Pc[(long)krow * g.N + n] = __builtin_ldexpf(acc[i][j][r], descale) + old[r];
        }
      }
  }
  if (kt == 0 && g.Pdb != nullptr) {
    // bias gradient: column sums of G over this chunk; the four row quads of a column quad meet through LDS
    float* red = smem;  // [4 (rq)][BN]
    if (producer && !is_a && (st >> 2) < BN / 4) *reinterpret_cast<f32x4*>(red + rq * BN + c4 * 4) = dbs;
    __syncthreads();
    if (t < BN) {
      const float sum = red[t] + red[BN + t] + red[2 * BN + t] + red[3 * BN + t];
      if (n0 + t < g.N) g.Pdb[(long)chunk * g.N + n0 + t] = sum;
    }
  }
}

__global__ void k_naive_gemm_tn(TnArgs g) {
  // one thread per output (kk, n), one block row per chunk
  int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int chunk = blockIdx.y;
  const long r_begin = (long)chunk * g.chunk_rows;
  long r_end = r_begin + g.chunk_rows;
  if (r_end > g.M) r_end = g.M;
  const int nout = g.Ktot * g.N;
  if (o < nout) {
    int kk = o / g.N, n = o - kk * g.N;
    int p = kk / g.Ka, k = kk - p * g.Ka;
    const float* Ap = g.A[p];
    int sh = p == 0 ? g.a0_shift : 0;
    float acc = 0.f;
    const int q = n / g.Gc;
    const float* Gq = g.G[q] + (n - q * g.Gc);
    for (long r = r_begin; r < r_end; r++) acc = fmaf(Ap[(r >> sh) * g.Ka + k], Gq[r * g.Gc], acc);
    g.P[(long)chunk * nout + o] = g.accum ? g.P[(long)chunk * nout + o] + acc : acc;
  }
  if (g.Pdb != nullptr && o < g.N) {
    float s = 0.f;
    const int q = o / g.Gc;
    const float* Gq = g.G[q] + (o - q * g.Gc);
    for (long r = r_begin; r < r_end; r++) s += Gq[r * g.Gc];
    g.Pdb[(long)chunk * g.N + o] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// weight pack / gradient unpack
// ---------------------------------------------------------------------------------------------
__global__ void k_weight_pack(const float* __restrict__ W, float* __restrict__ Wt, float* __restrict__ W2,
                              float* __restrict__ W3, int Fout, int Fin, int K) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long tot = (long)Fout * Fin * K;
  if (idx >= tot) return;
  // idx enumerates Wt: [(k*Fin+fin)][fout]
  int fout = (int)(idx % Fout);
  long kk = idx / Fout;
  int k = (int)(kk / Fin), fin = (int)(kk % Fin);
  float v = W[(long)fout * Fin * K + (long)fin * K + k];
  Wt[idx] = v;
  if (W2) W2[(long)fout * Fin * K + kk] = v;
  if (W3) W3[((long)k * Fout + fout) * Fin + fin] = v;     // [k*Fout + fout][fin]: B operand of dX = [g|Lg|L2g] W3
}

// K = 1, Wt only: a plain transpose Wt[fin][fout] = W[fout][fin] (the fc lift of meshnet.py:105 and, round 5, the four
// 4096 x 4096 Linears of PoseNet: 67 MB each per optimizer step).  k_weight_pack reads W at a stride of Fin floats per lane:
// 100 us for 4096 x 4096 (1.3 TB/s) in the round-5 step trace.  64 x 64 tiles through LDS, 16-byte accesses on both sides.
__global__ __launch_bounds__(256) void k_transpose_tiled(const float* __restrict__ W, float* __restrict__ Wt, int R, int C) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tq = threadIdx.x & 15, ty = threadIdx.x >> 4;           // 16 column quads x 16 rows per pass
  const bool vec = (C & 3) == 0 && (R & 3) == 0;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int r = r0 + ty + 16 * p, c = c0 + 4 * tq;
    if (r < R) {
      if (vec && c + 3 < C) {
        const float4 v = *reinterpret_cast<const float4*>(W + (long)r * C + c);
        tile[ty + 16 * p][4 * tq] = v.x; tile[ty + 16 * p][4 * tq + 1] = v.y;
        tile[ty + 16 * p][4 * tq + 2] = v.z; tile[ty + 16 * p][4 * tq + 3] = v.w;
      } else {
        for (int e = 0; e < 4; e++)
          if (c + e < C) tile[ty + 16 * p][4 * tq + e] = W[(long)r * C + c + e];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int c = c0 + ty + 16 * p, r = r0 + 4 * tq;               // output row c, columns r .. r + 3
    if (c < C) {
      if (vec && r + 3 < R) {
        float4 v;
        v.x = tile[4 * tq][ty + 16 * p]; v.y = tile[4 * tq + 1][ty + 16 * p];
        v.z = tile[4 * tq + 2][ty + 16 * p]; v.w = tile[4 * tq + 3][ty + 16 * p];
        *reinterpret_cast<float4*>(Wt + (long)c * R + r) = v;
      } else {
        for (int e = 0; e < 4; e++)
          if (r + e < R) Wt[(long)c * R + r + e] = tile[4 * tq + e][ty + 16 * p];
      }
    }
  }
}

// Weff[k][n] = Wt[k][n] + a * Wt[Ka + k][n] + b * Wt[2 Ka + k][n]: the K = Fin weight seen by fake vertices
__global__ void k_weight_eff(const float* __restrict__ Wt, float* __restrict__ We, int Ka, int N, float a, float b) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Ka * N) return;
  const long plane = (long)Ka * N;
  We[idx] = Wt[idx] + a * Wt[plane + idx] + b * Wt[2 * plane + idx];
}

// ---- every slice image of a network's conv weights in two launches ----------------------------------------------------
// A split conv needs four pre-split operands of its weight W[Fout][Fin * 3 + k] per optimizer step: [3 Fin, Fout] (forward,
// real rows), its effective K = Fin form W0 + a W1 + b W2 (forward, padding rows), [3 Fout, Fin] (backward dX, real rows)
// and that one's effective form.  Through p2m_weight_pack / _eff / _split / p2m_amax that is 8 launches per layer, ~150
// per step, each a few microseconds long with a pipeline drain on either side - 1.5 ms of a 38 ms step.  Here: one block per
// layer for the parameters' amax words, then ONE launch that writes all images straight from W (blockIdx.y = layer).
struct ConvWeights {            // == p2m_conv_weights of include/p2m.h
  const float* W;
  int Fout, Fin;
  float fake_a, fake_b;
  int eff_bits, reserved;
  unsigned short *Bx_f, *Bx_ef, *Bx_b, *Bx_eb;
  unsigned* amax;
};
static_assert(sizeof(ConvWeights) == 72, "layout shared with the callers (include/p2m.h, ops.py)");

__global__ __launch_bounds__(1024) void k_conv_weights_amax(const ConvWeights* __restrict__ d) {
  __shared__ float red[16];
  const ConvWeights c = d[blockIdx.x];
  const long n = (long)c.Fout * c.Fin * 3;
  float m = 0.f;
  const long n4 = (reinterpret_cast<uintptr_t>(c.W) & 15) == 0 ? n >> 2 : 0;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(c.W);
  for (long i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(amax_abs(v[0]), amax_abs(v[1]))), fmaxf(amax_abs(v[2]), amax_abs(v[3])));
  }
  for (long j = n4 * 4 + threadIdx.x; j < n; j += 1024) m = fmaxf(m, amax_abs(c.W[j]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; w++) m = fmaxf(m, red[w]);
    *c.amax = __float_as_uint(m);
  }
}

// one image: Bx[k / 16][slice][n][k % 16] of value(k, n), n < Npad (zero beyond N), + the trailing amax word (NS = 2)
template <int NS, typename F>
__device__ __forceinline__ void write_weight_image(unsigned short* __restrict__ Bx, int K, int N, unsigned word, int bits,
                                                   F value) {
  if (Bx == nullptr) return;
  const int Npad = cdiv_dev(N, 128) * 128;
  const long sl = (long)Npad * 16, tot = (long)Npad * K;
  float sc = 1.f;
  if constexpr (NS == 2) {
    sc = exp2_int(slice_scale_exp(word, bits));
    if (blockIdx.x == 0 && threadIdx.x == 0)
      *reinterpret_cast<unsigned*>(Bx + (long)NS * Npad * K) =
          word == 0u ? 0u : __float_as_uint(__builtin_ldexpf(__uint_as_float(word), bits));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / Npad), n = (int)(i - (long)k * Npad);
    const float v = n < N ? value(k, n) : 0.f;
    unsigned short* dst = Bx + (long)(k >> 4) * NS * sl + (long)n * 16 + (k & 15);
    if constexpr (NS == 3) {
      unsigned h, m, l;
      split3(v, h, m, l);
      dst[0] = (unsigned short)(h >> 16);
      dst[sl] = (unsigned short)(m >> 16);
      dst[2 * sl] = (unsigned short)(l >> 16);
    } else {
      const float y = v * sc;
      const _Float16 h = (_Float16)y;
      const _Float16 l = (_Float16)(y - (float)h);
      dst[0] = __builtin_bit_cast(unsigned short, h);
      dst[sl] = __builtin_bit_cast(unsigned short, l);
    }
  }
}

template <int NS>
__global__ __launch_bounds__(256) void k_conv_weights_images(const ConvWeights* __restrict__ d) {
  const ConvWeights c = d[blockIdx.y];
  const float* __restrict__ W = c.W;
  const int Fout = c.Fout, Fin = c.Fin, ld = 3 * c.Fin;
  const float a = c.fake_a, b = c.fake_b;
  const unsigned word = NS == 2 ? *c.amax : 0u;
  // forward, real rows: Wt[kk * Fin + fin][fo] = W[fo][fin * 3 + kk]
  write_weight_image<NS>(c.Bx_f, 3 * Fin, Fout, word, 0, [&](int k, int n) {
    const int kk = k / Fin, fin = k - kk * Fin;
    return W[(long)n * ld + fin * 3 + kk];
  });
  // forward, padding rows: (W0 + a W1 + b W2)[fin][fo]   (the expression of k_weight_eff)
  write_weight_image<NS>(c.Bx_ef, Fin, Fout, word, c.eff_bits, [&](int k, int n) {
    const float* w = W + (long)n * ld + k * 3;
    return w[0] + a * w[1] + b * w[2];
  });
  // backward dX, real rows: W3[kk * Fout + fo][fin] = W[fo][fin * 3 + kk]
  write_weight_image<NS>(c.Bx_b, 3 * Fout, Fin, word, 0, [&](int k, int n) {
    const int kk = k / Fout, fo = k - kk * Fout;
    return W[(long)fo * ld + n * 3 + kk];
  });
  // backward dX, padding rows: (W3_0 + a W3_1 + b W3_2)[fo][fin]
  write_weight_image<NS>(c.Bx_eb, Fout, Fin, word, c.eff_bits, [&](int k, int n) {
    const float* w = W + (long)k * ld + n * 3;
    return w[0] + a * w[1] + b * w[2];
  });
}

// Block = 32 consecutive output elements x UNP_CG chunk groups: the partial buffers hold hundreds of chunks (one per
// sample in row-set mode), so the chunk loop is split UNP_CG ways (4 loads in flight each) and reduced through LDS.
// 8 groups (256 threads): alone on the GPU 32 groups are faster, but this kernel runs on the side stream next to the
// GEMM blocks that own most of the CUs' registers and LDS, and small blocks find a slot sooner (whole step, meshes/s:
// 2 groups 4213, 4: 4222-4230, 8: 4206-4253, 16: 4209, 32: 4174-4189).
constexpr int UNP_CG = 8;
__global__ __launch_bounds__(32 * UNP_CG) void k_weight_grad_unpack(
    const float* __restrict__ P, const float* __restrict__ Pdb, int nchunks, float* __restrict__ dW,
    float* __restrict__ db, int Fout, int Fin, int K, int accumulate, int layout, int pdb_stride,
    const float* __restrict__ P2, const float* __restrict__ Pdb2, int nchunks2, float fake_a, float fake_b) {
  __shared__ double red[UNP_CG][32];
  const int e = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + e;
  const long tot = (long)Fout * Fin * K;
  int fout = 0, k = 0, fin = 0;
  double s = 0.0;
  if (idx < tot) {
    if (layout == 0) {            // P[chunk][k*Fin + fin][fout]
      fout = (int)(idx % Fout);
      long kk = idx / Fout;
      k = (int)(kk / Fin); fin = (int)(kk % Fin);
    } else if (layout == 2) {     // P[chunk][fout][fin], K = 1: already the nn.Linear layout (a plain Linear's gradient
      fout = (int)(idx / Fin);    // taken as G^T X with the roles of the two operands swapped): reads AND writes coalesced
      fin = (int)(idx - (long)fout * Fin);
    } else {                      // P[chunk][fin][k*Fout + fout]   (weight gradient taken as X^T [g|Lg|L2g])
      long nn = idx % ((long)K * Fout);
      fin = (int)(idx / ((long)K * Fout));
      k = (int)(nn / Fout); fout = (int)(nn % Fout);
    }
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int c = cg;
    for (; c + 3 * UNP_CG < nchunks; c += 4 * UNP_CG) {
      t0 += P[(long)c * tot + idx];
      t1 += P[(long)(c + UNP_CG) * tot + idx];
      t2 += P[(long)(c + 2 * UNP_CG) * tot + idx];
      t3 += P[(long)(c + 3 * UNP_CG) * tot + idx];
    }
    for (; c < nchunks; c += UNP_CG) t0 += P[(long)c * tot + idx];
    s = ((double)t0 + (double)t1) + ((double)t2 + (double)t3);
    if (P2 != nullptr) {          // fake-vertex partials P2[chunk][fin][fout] enter plane k scaled by (1, a, b)[k]
      const long o2 = (long)fin * Fout + fout, st2 = (long)Fin * Fout;
      float q0 = 0.f, q1 = 0.f;
      int c2 = cg;
      for (; c2 + UNP_CG < nchunks2; c2 += 2 * UNP_CG) {
        q0 += P2[(long)c2 * st2 + o2];
        q1 += P2[(long)(c2 + UNP_CG) * st2 + o2];
      }
      for (; c2 < nchunks2; c2 += UNP_CG) q0 += P2[(long)c2 * st2 + o2];
      s += ((double)q0 + (double)q1) * (k == 0 ? 1.0 : (k == 1 ? (double)fake_a : (double)fake_b));
    }
  }
  red[cg][e] = s;
  __syncthreads();
  if (cg == 0 && idx < tot) {
    double r = 0.0;
#pragma unroll
    for (int q = 0; q < UNP_CG; q++) r += red[q][e];
    const long o = (long)fout * Fin * K + (long)fin * K + k;
    dW[o] = accumulate ? dW[o] + (float)r : (float)r;
  }
  // bias gradient: the first blocks also reduce Pdb (one output feature per lane, the chunks spread over the UNP_CG
  // lane groups and merged through LDS: a single thread walking all chunks was the kernel's long pole)
  if (db != nullptr && Pdb != nullptr && (long)blockIdx.x * 32 < Fout) {      // block-uniform
    __syncthreads();
    double r = 0.0;
    if (idx < Fout) {
      for (int c = cg; c < nchunks; c += UNP_CG) r += (double)Pdb[(long)c * pdb_stride + idx];
      if (Pdb2 != nullptr)
        for (int c2 = cg; c2 < nchunks2; c2 += UNP_CG) r += (double)Pdb2[(long)c2 * Fout + idx];
    }
    red[cg][e] = r;
    __syncthreads();
    if (cg == 0 && idx < Fout) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < UNP_CG; q++) t += red[q][e];
      db[idx] = accumulate ? db[idx] + (float)t : (float)t;
    }
  }
}

}  // namespace p2m

using namespace p2m;

extern "C" int32_t p2m_stats_tile_rows(void) { return BM; }


static inline int slices_of(int arith) { return arith == P2M_ARITH_F16X2 ? 2 : 3; }
// the amax word of a two-fp16-slice weight image trails its slices (8 elements = 16 bytes: keeps images 16-byte sized)
static inline const unsigned* weight_amax_word(const void* Bx, int arith, long Npad, long K) {
  return arith == P2M_ARITH_F16X2
             ? reinterpret_cast<const unsigned*>(static_cast<const unsigned short*>(Bx) + 2 * Npad * K)
             : nullptr;
}

extern "C" int64_t p2m_weight_split_elems(int32_t K, int32_t N, int32_t arith) {
  if (K <= 0 || N <= 0 || K % 16 != 0 || (arith != P2M_ARITH_BF16X3 && arith != P2M_ARITH_F16X2)) return 0;
  return (long)slices_of(arith) * (cdiv(N, 128) * 128ll) * K + (arith == P2M_ARITH_F16X2 ? 8 : 0);
}

extern "C" int p2m_weight_split(const float* Bm, int32_t K, int32_t N, int32_t arith, const void* amax_in,
                                int32_t amax_bits, void* Bx, void* stream) {
  P2M_CHECK_ARG(Bm && Bx && K > 0 && N > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(K % 16 == 0, "K must be a multiple of 16");
  P2M_CHECK_ARG(arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "arith must be P2M_ARITH_BF16X3 or P2M_ARITH_F16X2");
  const int Npad = cdiv(N, 128) * 128;
  const dim3 grid(cdiv((long)Npad * K, 256));
  unsigned short* bx = static_cast<unsigned short*>(Bx);
  if (arith == P2M_ARITH_F16X2) {
    const unsigned* word = static_cast<const unsigned*>(amax_in);
    P2M_CHECK_ARG(amax_bits >= 0 && amax_bits <= 30, "amax_bits out of range");
    if (word == nullptr) {              // no bound from the caller: the weight's own maximum (one small extra launch)
      unsigned* trailer = const_cast<unsigned*>(weight_amax_word(Bx, arith, Npad, K));
      hipLaunchKernelGGL(k_amax_one_block, dim3(1), dim3(1024), 0, (hipStream_t)stream, Bm, (long)K * N, trailer);
      word = trailer;
      amax_bits = 0;
    }
    hipLaunchKernelGGL(k_weight_split<2>, grid, dim3(256), 0, (hipStream_t)stream, Bm, bx, K, N, Npad, word, amax_bits);
  } else {
    hipLaunchKernelGGL(k_weight_split<3>, grid, dim3(256), 0, (hipStream_t)stream, Bm, bx, K, N, Npad, nullptr, 0);
  }
  return check_launch("weight_split");
}

extern "C" int p2m_conv_weights_prepare(const p2m_conv_weights* dev_desc, int32_t n, int32_t arith, void* stream) {
  static_assert(sizeof(p2m_conv_weights) == sizeof(ConvWeights), "descriptor layout");
  P2M_CHECK_ARG(dev_desc != nullptr && n > 0, "null descriptor array or no layers");
  P2M_CHECK_ARG(arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "arith must be P2M_ARITH_BF16X3 or P2M_ARITH_F16X2");
  const ConvWeights* d = reinterpret_cast<const ConvWeights*>(dev_desc);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(96, n);                    // grid-stride over each image: the largest (768 x 256) is 768 elements per thread
  if (arith == P2M_ARITH_F16X2) {
    hipLaunchKernelGGL(k_conv_weights_amax, dim3(n), dim3(1024), 0, s, d);
    hipLaunchKernelGGL(k_conv_weights_images<2>, grid, dim3(256), 0, s, d);
  } else {
    hipLaunchKernelGGL(k_conv_weights_images<3>, grid, dim3(256), 0, s, d);
  }
  return check_launch("conv_weights_prepare");
}

extern "C" int p2m_amax(const float* x, int64_t n, void* word, void* stream) {
  P2M_CHECK_ARG(x && word && n >= 0, "null pointer");
  P2M_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 3) == 0, "x must be 4-byte aligned");
  if (n == 0) return P2M_OK;
  long head = (long)((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4;
  if (head > n) head = n;
  const long n4 = (n - head) / 4;
  const int tail = (int)(n - head - 4 * n4);
  const int grid = (int)(n4 < 256l * 4 * 1024 ? (n4 > 0 ? cdiv(n4, 256 * 4) : 1) : 1024);    // >= 4 float4 per thread
  hipLaunchKernelGGL(k_amax, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (int)head, n4, tail,
                     static_cast<unsigned*>(word));
  return check_launch("amax");
}

extern "C" int p2m_amax_rows(p2m_graph_t gh, int32_t row_set, const float* x, int32_t B, int32_t F, void* word,
                             void* stream) {
  P2M_CHECK_ARG(gh && x && word, "null pointer");
  P2M_CHECK_ARG(row_set >= 0 && row_set <= 4, "row_set must be 0 (live rows) or 1..4");
  P2M_CHECK_ARG(F > 0 && F % 4 == 0, "F must be a positive multiple of 4");
  const Graph& gr = *reinterpret_cast<const Graph*>(gh);
  RowSet rs;
  if (row_set == 0) {                       // every row that holds data: all of them, or the live ones under classes
    rs.V = gr.V;
    rs.ids = gr.live_ids;
    rs.n = gr.live_ids ? gr.n_live : gr.V;
  } else {
    rs = row_set_of(gr, row_set);
  }
  if (B <= 0 || rs.n == 0) return P2M_OK;
  const long tot = (long)B * rs.n * (F / 4);
  const int grid = (int)(tot < 256l * 2048 ? cdiv(tot, 256) : 2048);       // (one load per thread and iteration: keep the grid wide)
  hipLaunchKernelGGL(k_amax_rows, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, rs, B, F, static_cast<unsigned*>(word));
  return check_launch("amax_rows");
}

// picks the instantiation: tile width from N, EXTRA epilogue, native f32 MFMA or slices (g.Bx != nullptr; two fp16
// slices when g.a_amax is set)
template <bool ROWS>
static void launch_gemm_planes(GemmArgs& g, bool extra, hipStream_t s) {
  const bool wide = (g.N % 128 == 0);
  g.ntn = wide ? g.N / 128 : cdiv(g.N, 64);
  const dim3 grid(cdiv(g.ntm, 8) * 8 * g.ntn);
  if (g.Bx != nullptr) {
#define P2M_LAUNCH_WS(BNv, EX)                                                                              \
  do {                                                                                                      \
    if (g.a_amax != nullptr) hipLaunchKernelGGL((k_gemm_planes_ws<BNv, EX, ROWS, 2>), grid, dim3(512), 0, s, g); \
    else hipLaunchKernelGGL((k_gemm_planes_ws<BNv, EX, ROWS, 3>), grid, dim3(512), 0, s, g);                 \
  } while (0)
    if (wide) { if (extra) P2M_LAUNCH_WS(128, true); else P2M_LAUNCH_WS(128, false); }
    else { if (extra) P2M_LAUNCH_WS(64, true); else P2M_LAUNCH_WS(64, false); }
#undef P2M_LAUNCH_WS
  } else {
#define P2M_LAUNCH(BNv, EX) hipLaunchKernelGGL((k_gemm_planes<BNv, 32, EX, ROWS>), grid, dim3(256), 0, s, g)
    if (wide) { if (extra) P2M_LAUNCH(128, true); else P2M_LAUNCH(128, false); }
    else { if (extra) P2M_LAUNCH(64, true); else P2M_LAUNCH(64, false); }
#undef P2M_LAUNCH
  }
}

extern "C" int p2m_gemm_planes(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                               int32_t a0_shift, const float* Bm, const void* Bsplit, int32_t arith,
                               const void* a_amax, int32_t a_bits, const float* bias,
                               const float* addend, float* C0, float* C1, float* C2, int32_t nplanesC, int32_t Nc,
                               int32_t pair_out, int64_t M, float* stats, const float* act_scale,
                               const float* act_shift, int32_t act_relu, void* amax_out, void* stream) {
  P2M_CHECK_ARG(nplanesA >= 1 && nplanesA <= 3 && nplanesC >= 1 && nplanesC <= 3, "plane count must be 1..3");
  P2M_CHECK_ARG(arith == P2M_ARITH_F32 || arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "unknown arithmetic");
  P2M_CHECK_ARG((act_scale == nullptr) == (act_shift == nullptr), "act_scale / act_shift must both be given or both NULL");
  P2M_CHECK_ARG(!((act_scale || act_relu) && (stats || pair_out)), "fused activation excludes stats and pair_out");
  P2M_CHECK_ARG(A0 && Bm && C0 && Ka > 0 && Nc > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(a0_shift == 0 || a0_shift == 1, "a0_shift must be 0 or 1");
  if (M <= 0) return P2M_OK;
  GemmArgs g;
  g.A[0] = A0; g.A[1] = A1; g.A[2] = A2;
  g.C[0] = C0; g.C[1] = C1; g.C[2] = C2;
  for (int p = 0; p < nplanesA; p++) P2M_CHECK_ARG(g.A[p] != nullptr, "missing A plane");
  for (int p = 0; p < nplanesC; p++) P2M_CHECK_ARG(g.C[p] != nullptr, "missing C plane");
  P2M_CHECK_ARG((addend == nullptr && !pair_out) || nplanesC == 1, "addend / pair_out need a single output plane");
  P2M_CHECK_ARG(!(pair_out && stats), "pair_out and stats are mutually exclusive");
  g.Bm = Bm; g.bias = bias; g.addend = addend; g.pair_out = pair_out; g.stats = stats; g.M = M;
  g.act_scale = act_scale; g.act_shift = act_shift; g.act_relu = act_relu;
  g.nplanesA = nplanesA; g.Ka = Ka; g.a0_shift = a0_shift;
  g.N = nplanesC * Nc; g.Nc = Nc;
  g.ids = nullptr; g.nset = 0; g.V = 0; g.tps = 0; g.compact = 0; g.row_w = nullptr;
  g.Bx = nullptr; g.Npad = 0; g.Ktot = 0;
  g.a_amax = g.b_amax = nullptr; g.a_bits = 0;
  g.amax_out = static_cast<unsigned*>(amax_out);
  g.in_scale = g.in_shift = nullptr;
  hipStream_t s = (hipStream_t)stream;
  const bool mfma_ok = (Ka % BK == 0) && (g.N % 32 == 0) && (Nc % 32 == 0);
  if (!mfma_ok) {
    P2M_CHECK_ARG(!pair_out, "pair_out needs the MFMA path (Ka % 32 == 0, N % 32 == 0)");
    P2M_CHECK_ARG(!amax_out, "amax_out needs the MFMA path (Ka % 32 == 0, N % 32 == 0)");
    long tot = M * g.N;
    g.ntm = g.ntn = 0;
    hipLaunchKernelGGL(k_naive_gemm_planes, dim3(cdiv(tot, 256)), dim3(256), 0, s, g);
    if (stats) {
      P2M_CHECK_ARG(nplanesC == 1, "stats need a single output plane");
      hipLaunchKernelGGL(k_naive_tile_stats, dim3(cdiv(M, BM)), dim3(64), 0, s, C0, stats, (long)M, g.N);
    }
    return check_launch("gemm_planes(naive)");
  }
  if (arith != P2M_ARITH_F32) {
    P2M_CHECK_ARG(Bsplit != nullptr, "the slice arithmetics need the pre-split weight (p2m_weight_split)");
    P2M_CHECK_ARG(arith != P2M_ARITH_F16X2 || a_amax != nullptr, "P2M_ARITH_F16X2 needs the amax word of the A planes");
    g.Bx = static_cast<const unsigned short*>(Bsplit);
    g.Npad = cdiv(g.N, 128) * 128;
    g.Ktot = nplanesA * Ka;
    if (arith == P2M_ARITH_F16X2) {
      g.a_amax = static_cast<const unsigned*>(a_amax);
      g.a_bits = a_bits;
      g.b_amax = weight_amax_word(Bsplit, arith, g.Npad, g.Ktot);
    }
  }
  g.ntm = cdiv(M, BM);
  launch_gemm_planes<false>(g, addend != nullptr || pair_out, s);
  return check_launch("gemm_planes");
}

extern "C" int p2m_gemm_planes_rows(p2m_graph_t gh, int32_t row_set, int32_t B, const float* A0, const float* A1,
                                    const float* A2, int32_t nplanesA, int32_t Ka, int32_t a0_shift,
                                    int32_t planes_compact, const float* Bm, const void* Bsplit, int32_t arith,
                                    const void* a_amax, int32_t a_bits, const float* bias,
                                    const float* addend, float* C, int32_t N, float* stats, const float* act_scale,
                                    const float* act_shift, int32_t act_relu, void* amax_out, const float* in_scale,
                                    const float* in_shift, void* stream) {
  P2M_CHECK_ARG(gh && A0 && Bm && C, "null pointer");
  P2M_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale / in_shift must both be given or both NULL");
  P2M_CHECK_ARG(in_scale == nullptr || (arith != P2M_ARITH_F32 && Ka <= 256),
                "activation on load exists in the slice arithmetics only, for Ka <= 256");
  P2M_CHECK_ARG(arith == P2M_ARITH_F32 || arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "unknown arithmetic");
  P2M_CHECK_ARG(arith == P2M_ARITH_F32 || Bsplit != nullptr, "the slice arithmetics need the pre-split weight (p2m_weight_split)");
  P2M_CHECK_ARG(arith != P2M_ARITH_F16X2 || a_amax != nullptr, "P2M_ARITH_F16X2 needs the amax word of the A planes");
  P2M_CHECK_ARG((act_scale == nullptr) == (act_shift == nullptr), "act_scale / act_shift must both be given or both NULL");
  P2M_CHECK_ARG(!((act_scale || act_relu) && stats), "fused activation excludes stats");
  P2M_CHECK_ARG(row_set_valid(row_set), "row_set must be 1 (real), 2 (fake), 3 (paired real) or 4 (paired fake)");
  P2M_CHECK_ARG(nplanesA >= 1 && nplanesA <= 3, "plane count must be 1..3");
  P2M_CHECK_ARG(Ka > 0 && Ka % BK == 0 && N > 0 && N % 32 == 0, "Ka and N must be positive multiples of 32");
  P2M_CHECK_ARG(a0_shift == 0 || a0_shift == 1, "a0_shift must be 0 or 1");
  const Graph& gr = *reinterpret_cast<const Graph*>(gh);
  const RowSet rs = row_set_of(gr, row_set);
  if (B <= 0 || rs.n == 0) return P2M_OK;
  GemmArgs g;
  g.A[0] = A0; g.A[1] = A1; g.A[2] = A2;
  for (int p = 0; p < nplanesA; p++) P2M_CHECK_ARG(g.A[p] != nullptr, "missing A plane");
  g.C[0] = C; g.C[1] = nullptr; g.C[2] = nullptr;
  g.Bm = Bm; g.bias = bias; g.addend = addend; g.pair_out = 0; g.stats = stats;
  g.act_scale = act_scale; g.act_shift = act_shift; g.act_relu = act_relu;
  g.nplanesA = nplanesA; g.Ka = Ka; g.a0_shift = a0_shift; g.N = N; g.Nc = N;
  g.ids = rs.ids; g.nset = rs.n; g.V = rs.V; g.tps = cdiv(rs.n, BM); g.compact = planes_compact;
  g.row_w = (stats != nullptr && row_set == 2) ? gr.fake_wts : nullptr;   // representatives count once per class member
  g.Bx = arith == P2M_ARITH_F32 ? nullptr : static_cast<const unsigned short*>(Bsplit);
  g.Npad = cdiv(N, 128) * 128;
  g.Ktot = nplanesA * Ka;
  g.a_amax = g.b_amax = nullptr; g.a_bits = 0;
  if (arith == P2M_ARITH_F16X2) {
    g.a_amax = static_cast<const unsigned*>(a_amax);
    g.a_bits = a_bits;
    g.b_amax = weight_amax_word(Bsplit, arith, g.Npad, g.Ktot);
  }
  g.amax_out = static_cast<unsigned*>(amax_out);
  g.in_scale = in_scale; g.in_shift = in_shift;
  g.M = (long)B * g.tps * BM;       // logical (padded) rows; validity comes from the row table
  g.ntm = B * g.tps;
  launch_gemm_planes<true>(g, addend != nullptr, (hipStream_t)stream);
  return check_launch("gemm_planes_rows");
}

// rows per sample tile count of a row set (for the BatchNorm finalize): tiles_per_sample = ceil(n / 128)
extern "C" int32_t p2m_rows_tiles_per_sample(p2m_graph_t gh, int32_t row_set) {
  if (!gh || !row_set_valid(row_set)) return 0;
  const Graph& gr = *reinterpret_cast<const Graph*>(gh);
  return cdiv(row_set_of(gr, row_set).n, BM);
}

static int gemm_tn_impl(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                        int32_t a0_shift, const float* G0, const float* G1, const float* G2, int32_t nplanesG,
                        int32_t Gc, int64_t M, int64_t chunk_rows, float* P, float* Pdb, int32_t arith,
                        const void* a_amax, int32_t a_bits, const void* g_amax, int32_t g_bits, int32_t accumulate,
                        void* stream);
extern "C" int p2m_gemm_tn(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                           int32_t a0_shift, const float* G0, const float* G1, const float* G2, int32_t nplanesG,
                           int32_t Gc, int64_t M, int64_t chunk_rows, float* P, float* Pdb, int32_t arith,
                           const void* a_amax, int32_t a_bits, const void* g_amax, int32_t g_bits, void* stream) {
  return gemm_tn_impl(A0, A1, A2, nplanesA, Ka, a0_shift, G0, G1, G2, nplanesG, Gc, M, chunk_rows, P, Pdb, arith, a_amax,
                      a_bits, g_amax, g_bits, 0, stream);
}
// P[k][n] += sum_r A[r][k] G[r][n] over ALL M rows in one chunk: the weight gradient of a plain Linear
// (lib/models/posenet.py:19,22,59,68) added straight into the parameter's .grad - no partial buffer, no unpack pass.
extern "C" int p2m_gemm_tn_acc(const float* A, int32_t Ka, const float* G, int32_t N, int64_t M, float* P, int32_t arith,
                               const void* a_amax, const void* g_amax, void* stream) {
  const int64_t chunk_rows = ((M + 31) / 32) * 32;
  return gemm_tn_impl(A, nullptr, nullptr, 1, Ka, 0, G, nullptr, nullptr, 1, N, M, chunk_rows > 0 ? chunk_rows : 32, P,
                      nullptr, arith, a_amax, 0, g_amax, 0, 1, stream);
}
static int gemm_tn_impl(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                        int32_t a0_shift, const float* G0, const float* G1, const float* G2, int32_t nplanesG,
                        int32_t Gc, int64_t M, int64_t chunk_rows, float* P, float* Pdb, int32_t arith,
                        const void* a_amax, int32_t a_bits, const void* g_amax, int32_t g_bits, int32_t accumulate,
                        void* stream) {
  P2M_CHECK_ARG(nplanesA >= 1 && nplanesA <= 3 && nplanesG >= 1 && nplanesG <= 3, "plane count must be 1..3");
  P2M_CHECK_ARG(arith == P2M_ARITH_F32 || arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "unknown arithmetic");
  P2M_CHECK_ARG(arith != P2M_ARITH_F16X2 || (a_amax && g_amax), "P2M_ARITH_F16X2 needs the amax words of both operands");
  const int N = nplanesG * Gc;
  P2M_CHECK_ARG(A0 && G0 && P && Ka > 0 && Gc > 0 && chunk_rows > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(a0_shift == 0 || a0_shift == 1, "a0_shift must be 0 or 1");
  if (M <= 0) return P2M_OK;
  TnArgs g;
  g.A[0] = A0; g.A[1] = A1; g.A[2] = A2;
  for (int p = 0; p < nplanesA; p++) P2M_CHECK_ARG(g.A[p] != nullptr, "missing A plane");
  g.G[0] = G0; g.G[1] = G1; g.G[2] = G2; g.Gc = Gc;
  for (int p = 0; p < nplanesG; p++) P2M_CHECK_ARG(g.G[p] != nullptr, "missing G plane");
  g.P = P; g.Pdb = Pdb; g.M = M; g.chunk_rows = chunk_rows;
  g.nplanesA = nplanesA; g.Ka = Ka; g.a0_shift = a0_shift; g.Ktot = nplanesA * Ka; g.N = N;
  g.ids = nullptr; g.nset = 0; g.V = 0; g.splits = 1; g.compact = 0;
  g.a_amax = static_cast<const unsigned*>(a_amax); g.g_amax = static_cast<const unsigned*>(g_amax);
  g.a_bits = a_bits; g.g_bits = g_bits;
  g.a_scale = g.a_shift = nullptr;
  const int nchunks = cdiv(M, chunk_rows);
  g.nchunks = nchunks;
  g.accum = accumulate;
  P2M_CHECK_ARG(!accumulate || nchunks == 1, "accumulation needs the whole reduction in one chunk");
  hipStream_t s = (hipStream_t)stream;
  const bool mfma_ok = (Ka % 4 == 0) && (N % 32 == 0) && (Gc % 4 == 0) && (g.Ktot >= 32);
  if (!mfma_ok) {
    g.nkt = g.ntn = 0;
    int nout = g.Ktot * N;
    if (nout < N) nout = N;
    hipLaunchKernelGGL(k_naive_gemm_tn, dim3(cdiv(nout, 256), nchunks), dim3(256), 0, s, g);
    return check_launch("gemm_tn(naive)");
  }
  g.nkt = cdiv(g.Ktot, BM);
  const bool bx = arith != P2M_ARITH_F32;
  // N = 192 (three planes of 64): two 128-wide tiles (the second half empty) stage A twice, three 64-wide tiles thrice
  if (N % 128 == 0 || (bx && N > 128)) {
    g.ntn = cdiv(N, 128);
    const dim3 grid(g.nkt * g.ntn, nchunks), grid_ws((nchunks >= 8 ? cdiv(nchunks, 8) * 8 : nchunks) * g.nkt * g.ntn);
    if (arith == P2M_ARITH_F16X2) hipLaunchKernelGGL((k_gemm_tn_ws<128, false, 2>), grid_ws, dim3(512), 0, s, g);
    else if (bx) hipLaunchKernelGGL((k_gemm_tn_ws<128, false, 3>), grid_ws, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_tn<128, false>), grid, dim3(256), 0, s, g);
  } else {
    g.ntn = cdiv(N, 64);
    const dim3 grid(g.nkt * g.ntn, nchunks), grid_ws((nchunks >= 8 ? cdiv(nchunks, 8) * 8 : nchunks) * g.nkt * g.ntn);
    if (arith == P2M_ARITH_F16X2) hipLaunchKernelGGL((k_gemm_tn_ws<64, false, 2>), grid_ws, dim3(512), 0, s, g);
    else if (bx) hipLaunchKernelGGL((k_gemm_tn_ws<64, false, 3>), grid_ws, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_tn<64, false>), grid, dim3(256), 0, s, g);
  }
  return check_launch("gemm_tn");
}

extern "C" int p2m_gemm_tn_rows(p2m_graph_t gh, int32_t row_set, int32_t B, const float* A, int32_t Ka,
                                int32_t a0_shift, const float* G0, const float* G1, const float* G2, int32_t nplanesG,
                                int32_t Gc, int32_t planes_compact, int32_t splits, float* P, float* Pdb,
                                int32_t arith, const void* a_amax, const void* g_amax, int32_t g_bits,
                                const float* a_scale, const float* a_shift, void* stream) {
  P2M_CHECK_ARG(gh && A && G0 && P, "null pointer");
  P2M_CHECK_ARG((a_scale == nullptr) == (a_shift == nullptr), "a_scale / a_shift must both be given or both NULL");
  P2M_CHECK_ARG(a_scale == nullptr || arith != P2M_ARITH_F32, "activation on load exists in the slice arithmetics only");
  P2M_CHECK_ARG(arith == P2M_ARITH_F32 || arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "unknown arithmetic");
  P2M_CHECK_ARG(arith != P2M_ARITH_F16X2 || (a_amax && g_amax), "P2M_ARITH_F16X2 needs the amax words of both operands");
  P2M_CHECK_ARG(row_set_valid(row_set), "row_set must be 1 (real), 2 (fake), 3 (paired real) or 4 (paired fake)");
  P2M_CHECK_ARG(nplanesG >= 1 && nplanesG <= 3 && (splits >= 1 || splits <= -2),
                "plane count must be 1..3; splits >= 1, or <= -2 (that many whole samples per chunk)");
  P2M_CHECK_ARG(splits >= 1 || arith != P2M_ARITH_F32, "several samples per chunk exist in the slice arithmetics only");
  P2M_CHECK_ARG(Ka % 32 == 0 && Gc % 32 == 0, "Ka and Gc must be multiples of 32");
  P2M_CHECK_ARG(a0_shift == 0 || a0_shift == 1, "a0_shift must be 0 or 1");
  const Graph& gr = *reinterpret_cast<const Graph*>(gh);
  const RowSet rs = row_set_of(gr, row_set);
  if (B <= 0 || rs.n == 0) return P2M_OK;
  TnArgs g;
  g.A[0] = A; g.A[1] = nullptr; g.A[2] = nullptr;
  g.G[0] = G0; g.G[1] = G1; g.G[2] = G2; g.Gc = Gc;
  for (int p = 0; p < nplanesG; p++) P2M_CHECK_ARG(g.G[p] != nullptr, "missing G plane");
  const int N = nplanesG * Gc;
  g.P = P; g.Pdb = Pdb; g.M = (long)B * rs.n;
  const int spc = splits < 0 ? -splits : 1;
  if (splits < 0) splits = 1;
  g.chunk_rows = cdiv(rs.n, splits);
  g.nplanesA = 1; g.Ka = Ka; g.a0_shift = a0_shift; g.Ktot = Ka; g.N = N;
  g.ids = rs.ids; g.nset = rs.n; g.V = rs.V; g.splits = splits; g.compact = planes_compact;
  g.spc = spc; g.B = B;
  g.a_amax = static_cast<const unsigned*>(a_amax); g.g_amax = static_cast<const unsigned*>(g_amax);
  g.a_bits = 0; g.g_bits = g_bits;
  g.a_scale = a_scale; g.a_shift = a_shift;
  const int nchunks = spc > 1 ? cdiv(B, spc) : B * splits;
  g.nchunks = nchunks;
  hipStream_t s = (hipStream_t)stream;
  g.nkt = cdiv(g.Ktot, BM);
  const bool bx = arith != P2M_ARITH_F32;
  if (bx) g.chunk_rows = cdiv(g.chunk_rows, 16) * 16;    // 16-byte aligned id loads; trailing splits may be empty
  if (N % 128 == 0 || (bx && N > 128)) {
    g.ntn = cdiv(N, 128);
    const dim3 grid(g.nkt * g.ntn, nchunks), grid_ws((nchunks >= 8 ? cdiv(nchunks, 8) * 8 : nchunks) * g.nkt * g.ntn);
    if (arith == P2M_ARITH_F16X2) hipLaunchKernelGGL((k_gemm_tn_ws<128, true, 2>), grid_ws, dim3(512), 0, s, g);
    else if (bx) hipLaunchKernelGGL((k_gemm_tn_ws<128, true, 3>), grid_ws, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_tn<128, true>), grid, dim3(256), 0, s, g);
  } else {
    g.ntn = cdiv(N, 64);
    const dim3 grid(g.nkt * g.ntn, nchunks), grid_ws((nchunks >= 8 ? cdiv(nchunks, 8) * 8 : nchunks) * g.nkt * g.ntn);
    if (arith == P2M_ARITH_F16X2) hipLaunchKernelGGL((k_gemm_tn_ws<64, true, 2>), grid_ws, dim3(512), 0, s, g);
    else if (bx) hipLaunchKernelGGL((k_gemm_tn_ws<64, true, 3>), grid_ws, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((k_gemm_tn<64, true>), grid, dim3(256), 0, s, g);
  }
  return check_launch("gemm_tn_rows");
}

extern "C" int p2m_weight_pack(const float* W, float* Wt, float* W2, float* W3, int32_t Fout, int32_t Fin, int32_t K,
                               void* stream) {
  P2M_CHECK_ARG(W && Wt && Fout > 0 && Fin > 0 && K > 0, "null pointer or empty shape");
  long tot = (long)Fout * Fin * K;
  if (K == 1 && W2 == nullptr && W3 == nullptr && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(Wt) & 15) == 0) {
    hipLaunchKernelGGL(k_transpose_tiled, dim3(cdiv(Fin, 64), cdiv(Fout, 64)), dim3(256), 0, (hipStream_t)stream, W, Wt, Fout,
                       Fin);
    return check_launch("weight_pack(transpose)");
  }
  hipLaunchKernelGGL(k_weight_pack, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, W, Wt, W2, W3, Fout, Fin, K);
  return check_launch("weight_pack");
}

extern "C" int p2m_weight_eff(const float* Wt, float* We, int32_t Ka, int32_t N, float a, float b, void* stream) {
  P2M_CHECK_ARG(Wt && We && Ka > 0 && N > 0, "null pointer or empty shape");
  hipLaunchKernelGGL(k_weight_eff, dim3(cdiv((long)Ka * N, 256)), dim3(256), 0, (hipStream_t)stream, Wt, We, Ka, N, a, b);
  return check_launch("weight_eff");
}

extern "C" int p2m_weight_grad_unpack2(const float* P, const float* Pdb, int32_t nchunks, const float* P2,
                                       const float* Pdb2, int32_t nchunks2, float s1, float s2, float* dW, float* db,
                                       int32_t Fout, int32_t Fin, int32_t K, int32_t accumulate, void* stream) {
  P2M_CHECK_ARG(P && P2 && dW && Fout > 0 && Fin > 0 && K == 3 && nchunks > 0 && nchunks2 > 0, "null pointer or bad shape");
  long tot = (long)Fout * Fin * K;
  hipLaunchKernelGGL(k_weight_grad_unpack, dim3(cdiv(tot, 32)), dim3(32 * UNP_CG), 0, (hipStream_t)stream, P, Pdb, nchunks,
                     dW, db, Fout, Fin, K, accumulate, 1, K * Fout, P2, Pdb2, nchunks2, s1, s2);
  return check_launch("weight_grad_unpack2");
}

extern "C" int p2m_weight_grad_unpack(const float* P, const float* Pdb, int32_t nchunks, float* dW, float* db,
                                      int32_t Fout, int32_t Fin, int32_t K, int32_t accumulate, int32_t layout,
                                      int32_t pdb_stride, void* stream) {
  P2M_CHECK_ARG(P && dW && Fout > 0 && Fin > 0 && K > 0 && nchunks > 0, "null pointer or empty shape");
  P2M_CHECK_ARG(layout >= 0 && layout <= 2 && (layout != 2 || K == 1), "layout must be 0, 1 or (K = 1 only) 2");
  long tot = (long)Fout * Fin * K;
  hipLaunchKernelGGL(k_weight_grad_unpack, dim3(cdiv(tot, 32)), dim3(32 * UNP_CG), 0, (hipStream_t)stream, P, Pdb, nchunks,
                     dW, db, Fout, Fin, K, accumulate, layout, pdb_stride, nullptr, nullptr, 0, 0.f, 0.f);
  return check_launch("weight_grad_unpack");
}

#ifdef P2M_TN_TRACE
extern "C" int p2m_tn_trace_dump(unsigned long long* out /* [8][32][4] host */) {
  if (hipDeviceSynchronize() != hipSuccess) return P2M_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2m::g_tn_tr), sizeof(unsigned long long) * 8 * 32 * 4) == hipSuccess ? P2M_OK
                                                                                                                : P2M_ERR_HIP;
}
#endif
