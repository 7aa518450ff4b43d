// Fused Chebyshev graph convolution on gfx950: the K=3 recurrence AND the dense contraction in ONE kernel.
//
//     C[r, :] = [ A[r] | (L A)[r] | (L2 A)[r] ] * Bm  (+ bias) (+ addend[r])       r = b*V + v
//
// Reference arithmetic: lib/models/backbones/cheby_graph_conv.py:16-37 (forward, A = x) and, through
// dX = [g | L g | L2 g] * W' (L symmetric, lib/coarsening.py:23), its autograd backward (A = dL/dy).
// The basis planes T1 = L A, T2 = L2 A never touch HBM: a conv forward moves Ka + N floats per row
// instead of 6 Ka + N (basis kernel + plane GEMM), which is what keeps the FP32 MFMA fed -- the unfused
// 128->128 GEMM sits at 48 FLOP/B, only 2x above the HBM ridge.
//
// Structure (1024 threads = 8 consumer + 8 producer waves, ONE persistent block per CU, ~140 KB LDS;
// two MFMA waves per SIMD cover each other's LDS/barrier bubbles, two gather waves per SIMD keep ~16 KB of
// neighbour rows in flight per CU):
//   waves 8..15 PRODUCERS  per stage (32 features): 8 lanes per row gather the ~12 neighbour rows of each of
//               the 128 tile rows as 128-byte lines (float4 per lane, L2-resident), accumulate the two
//               Chebyshev planes in registers, and write X|T1|T2 chunks into the LDS A tile [plane][m][33].
//               The CSR rows of a wave's 32 tile rows are cached in LDS once per tile (wave-private).
//   waves 0..7  CONSUMERS  WM x WN over the BM x BN tile, v_mfma_f32_32x32x2_f32; A fragments from LDS
//               (stride 33: conflict-free), B fragments straight from global/L2 into registers
//               (double-buffered per 32-deep plane chunk) -- LDS is spent on the double-buffered A tile.
//   one __syncthreads per stage (>= 12k MFMA cycles); producers gather stage s+1 while consumers multiply
//   stage s, ACROSS tile boundaries (persistent loop), so the epilogue's store burst overlaps the next
//   tile's gathers.  Tiles are dealt to XCDs in contiguous row ranges: an XCD's CUs walk neighbouring
//   vertex tiles of the same samples, so gathered rows hit that XCD's L2.
#include <cstdlib>
#include <type_traits>

#include "p2m_common.h"

namespace p2m {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FBK = 32;
constexpr int FLD = FBK + 1;
constexpr int CSR_CAP = 448;           // cached CSR entries per producer wave (<= 16 rows x ~21 entries, + slack)

struct FusedArgs {
  Graph g;
  const float* A;
  const float* Bm;
  const float* bias;
  const float* addend;
  float* C;
  float* stats;
  float* E1;
  float* E2;
  long M;
  int Ka, N, a_shift, pair_out;
  int ntm, ntn, tiles_per_xcd, blocks_per_xcd;
  int debug;   // bit 0: skip the neighbour gathers, bit 1: skip the MFMAs (profiling aids, P2M_FUSED_DEBUG)
};

// Stage barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): every stage would then wait
// for the consumers' prefetched B fragments and, worse, for the epilogue's 64 global stores per lane.
__device__ __forceinline__ void stage_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

constexpr int NPROD = 8;                       // producer waves
constexpr int NCONS = 8;                       // consumer waves

template <int FBM, int BN, int WM>
__global__ __launch_bounds__(1024) void k_cheb_gemm_fused(FusedArgs g) {
  constexpr int WN = NCONS / WM;              // consumer waves are arranged WM x WN over the tile
  constexpr int TM = FBM / WM / 32;           // MFMA tiles along M per consumer wave
  constexpr int TN = BN / WN / 32;            // MFMA tiles along N per consumer wave
  constexpr int RPW = FBM / NPROD;            // tile rows owned by one producer wave
  constexpr int PPW = RPW / 8;                // producer passes (8 rows per wave per pass)
  constexpr int KH = 8;                       // k-steps per B-fragment sub-chunk
  static_assert(TM >= 1 && TN >= 1 && PPW >= 1, "tile too small for 8+8 waves");
  constexpr int GS = TN;                      // tiles per B-fragment group
  constexpr int NG = 1;
  extern __shared__ float smem[];
  float* As = smem;                                         // [2][3][FBM*33]
  float* csr_a = smem + 2 * 3 * FBM * FLD;                  // [NPROD][CSR_CAP]
  float* csr_b = csr_a + NPROD * CSR_CAP;
  int* csr_c = reinterpret_cast<int*>(csr_b + NPROD * CSR_CAP);

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const bool producer = wave >= NCONS;
  const int V = g.g.V;
  const int Vs = V >> g.a_shift;
  const int cpp = g.Ka / FBK;

  // ---- persistent tile list of this block --------------------------------------------------
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt_begin = xcd * g.tiles_per_xcd;
  int mt_end = mt_begin + g.tiles_per_xcd;
  if (mt_end > g.ntm) mt_end = g.ntm;
  // sequence of (mt, nt): mt = mt_begin + slot + i*blocks_per_xcd, nt inner
  int cur_mt = mt_begin + slot, cur_nt = 0;
  if (cur_mt >= mt_end) return;

  // ---- producer state ----------------------------------------------------------------------
  const int pw = wave - NCONS;                // producer wave owns tile rows [pw*RPW, (pw+1)*RPW)
  const int grp = lane >> 3, l8 = lane & 7;
  int rs[PPW], re[PPW];                       // CSR range of my rows
  long arow[PPW], gbase[PPW];                     // source row of plane 0, sample base row for gathers
  int cbase = 0;                              // first cached CSR entry of this wave
  bool use_cache = false;                     // wave-uniform: all rows of this wave lie inside the cached range
  // ---- consumer state ----------------------------------------------------------------------
  const int wm = (wave / WN) % WM, wn = wave % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  floatx16 acc[TM][TN];
  float4 bc[KH / 4][GS], bn[KH / 4][GS];      // B fragments (4 k-steps per float4) of the current / next sub-chunk

  auto produce_setup = [&](int mt) {          // once per tile: row metadata + CSR cache (wave-private)
    const long r0 = (long)mt * FBM + pw * RPW;
#pragma unroll
    for (int ps = 0; ps < PPW; ps++) {
      const long r = r0 + ps * 8 + grp;
      if (r < g.M) {
        const int b = (int)(r / V);
        const int v = (int)(r - (long)b * V);
        rs[ps] = g.g.rowptr[v];
        re[ps] = g.g.rowptr[v + 1];
        arow[ps] = r >> g.a_shift;
        gbase[ps] = (long)b * Vs;
      } else {
        rs[ps] = re[ps] = 0;
        arow[ps] = -1;
        gbase[ps] = 0;
      }
    }
    long rf = r0 < g.M ? r0 : g.M - 1;
    const int vf = (int)(rf % V);
    cbase = __builtin_amdgcn_readfirstlane(g.g.rowptr[vf]);
    int cend = cbase + CSR_CAP;
    if (cend > g.g.nnz) cend = g.g.nnz;
    for (int j = cbase + lane; j < cend; j += 64) {
      csr_c[pw * CSR_CAP + j - cbase] = g.g.col[j];
      csr_a[pw * CSR_CAP + j - cbase] = g.g.a[j];
      csr_b[pw * CSR_CAP + j - cbase] = g.g.b[j];
    }
    // wave-uniform: every row of this wave inside [cbase, cbase + CSR_CAP)?  (false when the 32 rows wrap
    // into the next sample or are unusually dense -> that tile reads its CSR rows from global/L2)
    bool ok = true;
#pragma unroll
    for (int ps = 0; ps < PPW; ps++) ok = ok && (re[ps] == rs[ps] || (rs[ps] >= cbase && re[ps] <= cend));
    use_cache = __builtin_amdgcn_readfirstlane((int)__all((int)ok)) != 0;
    // wave-private data: LDS writes of this wave are visible to it after the lgkm wait the compiler inserts
    __builtin_amdgcn_wave_barrier();
  };

  auto produce = [&](int buf, int chunk, bool write_e) {
    const int k0 = chunk * FBK + l8 * 4;
    float* as = As + buf * 3 * FBM * FLD;
#pragma unroll
    for (int ps = 0; ps < PPW; ps++) {
      const int m = pw * RPW + ps * 8 + grp;
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = x0, t2 = x0;
      if (arow[ps] >= 0) {
        x0 = *reinterpret_cast<const float4*>(g.A + arow[ps] * g.Ka + k0);
        const float* Ab = g.A + gbase[ps] * g.Ka + k0;
        // 8 gathers in flight per lane, NO per-load predicates: hipcc branches around a conditional load and
        // drains vmcnt(0) per element; padded slots re-read the row's last entry with coefficient 0 instead.
        // The cached / uncached CSR variants are two separate loops (a per-element select would branch too).
        const int e = (g.debug & 1) ? rs[ps] : re[ps];
        auto batch = [&](int j, auto cached, auto width) {
          constexpr int UB = decltype(width)::value;
          int c[UB];
          float ca[UB], cb[UB];
          float4 u[UB];
#pragma unroll
          for (int i = 0; i < UB; i++) {
            const bool ok = (j + i) < e;
            const int jj = ok ? j + i : e - 1;
            if constexpr (decltype(cached)::value) {
              const int o = pw * CSR_CAP + jj - cbase;
              c[i] = csr_c[o]; ca[i] = csr_a[o]; cb[i] = csr_b[o];
            } else {
              c[i] = g.g.col[jj]; ca[i] = g.g.a[jj]; cb[i] = g.g.b[jj];
            }
            ca[i] = ok ? ca[i] : 0.f;
            cb[i] = ok ? cb[i] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < UB; i++) u[i] = *reinterpret_cast<const float4*>(Ab + (long)(c[i] >> g.a_shift) * g.Ka);
#pragma unroll
          for (int i = 0; i < UB; i++) {
            t1.x = fmaf(ca[i], u[i].x, t1.x); t1.y = fmaf(ca[i], u[i].y, t1.y);
            t1.z = fmaf(ca[i], u[i].z, t1.z); t1.w = fmaf(ca[i], u[i].w, t1.w);
            t2.x = fmaf(cb[i], u[i].x, t2.x); t2.y = fmaf(cb[i], u[i].y, t2.y);
            t2.z = fmaf(cb[i], u[i].z, t2.z); t2.w = fmaf(cb[i], u[i].w, t2.w);
          }
        };
        // full batches of 8, then a batch of 4 for the tail (fake vertices have a single entry)
        int j = rs[ps];
        if (use_cache) {
          for (; j + 8 <= e; j += 8) batch(j, std::true_type{}, std::integral_constant<int, 8>{});
          for (; j < e; j += 4) batch(j, std::true_type{}, std::integral_constant<int, 4>{});
        } else {
          for (; j + 8 <= e; j += 8) batch(j, std::false_type{}, std::integral_constant<int, 8>{});
          for (; j < e; j += 4) batch(j, std::false_type{}, std::integral_constant<int, 4>{});
        }
        if (write_e) {
          const long r = ((arow[ps] << g.a_shift));   // only used with a_shift == 0 (checked on the host)
          *reinterpret_cast<float4*>(g.E1 + r * g.Ka + k0) = t1;
          *reinterpret_cast<float4*>(g.E2 + r * g.Ka + k0) = t2;
        }
      }
      float* d = as + m * FLD + l8 * 4;
      d[0] = x0.x; d[1] = x0.y; d[2] = x0.z; d[3] = x0.w;
      d += FBM * FLD;
      d[0] = t1.x; d[1] = t1.y; d[2] = t1.z; d[3] = t1.w;
      d += FBM * FLD;
      d[0] = t2.x; d[1] = t2.y; d[2] = t2.z; d[3] = t2.w;
    }
  };

  // B fragment sub-chunk index q enumerates (plane p, half h): q = p*2 + h.  Bm is FRAGMENT-MAJOR packed
  // (p2m_frag_pack): [kc = k/32][nt = n/32][q4 = 0..3][lane][4] so that one coalesced dwordx4 per lane brings the
  // B operands of 4 consecutive k-steps -- per-lane dword loads cost 4x the TA cycles and made the kernel TA-bound.
  constexpr int NQ = 3 * 2;
  const int ntiles32 = g.N / 32;
  auto load_bfrag = [&](int nt, int chunk, int q) {       // into bn
    const int h = q & 1, p = q >> 1;
    const int kc = p * cpp + chunk;
    const float4* bp = reinterpret_cast<const float4*>(g.Bm) +
                       ((long)(kc * ntiles32 + (nt * BN + wn * (BN / WN)) / 32) * 4 + h * (KH / 4)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < GS; j++)
#pragma unroll
      for (int k4 = 0; k4 < KH / 4; k4++) bn[k4][j] = bp[(long)(j * 4 + k4) * 64];
  };

  auto mfma_chunk = [&](int buf, int q) {                  // uses bc
    const int h = q & 1, p = q >> 1;
    const float* as = As + (buf * 3 + p) * FBM * FLD + (wm * (FBM / WM) + l31) * FLD + lhi + h * 2 * KH;
#pragma unroll
    for (int ks = 0; ks < KH; ks++) {
      float a[TM];
#pragma unroll
      for (int i = 0; i < TM; i++) a[i] = as[i * 32 * FLD + 2 * ks];
#pragma unroll
      for (int j = 0; j < GS; j++) {
        const float4 b4 = bc[ks >> 2][j];
        const float bv = (ks & 3) == 0 ? b4.x : (ks & 3) == 1 ? b4.y : (ks & 3) == 2 ? b4.z : b4.w;
#pragma unroll
        for (int i = 0; i < TM; i++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv, acc[i][j], 0, 0, 0);
      }
    }
  };

  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  };

  auto epilogue = [&](int mt, int nt) {
    const long m0 = (long)mt * FBM;
    const int n0 = nt * BN;
    float csum[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) csum[j] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + wn * (BN / WN) + j * 32 + l31;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long row = m0 + wm * (FBM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          float v = acc[i][j][r] + bv;
          if (g.addend != nullptr && row < g.M) v += g.addend[row * g.N + n];
          acc[i][j][r] = v;
        }
        if (g.pair_out) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const long row = m0 + wm * (FBM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (row < g.M) g.C[(row >> 1) * g.N + n] = acc[i][j][r] + acc[i][j][r + 1];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const long row = m0 + wm * (FBM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (row < g.M) {
              g.C[row * g.N + n] = acc[i][j][r];
              csum[j] += acc[i][j][r];
            }
          }
        }
      }
    }
    if (g.stats == nullptr) return;
    // per-tile BatchNorm partials (sum, centred sum of squares).  Consumer-only exchange through the stats
    // buffer in global memory would need a barrier; instead every consumer wave writes its own FBM/WM-row
    // sub-tile: stats rows are [WM*mt + wm].
    long rows_valid = g.M - (m0 + wm * (FBM / WM));
    if (rows_valid > FBM / WM) rows_valid = FBM / WM;
    if (rows_valid <= 0) rows_valid = 1;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int n = n0 + wn * (BN / WN) + j * 32 + l31;
      float s = csum[j] + __shfl_xor(csum[j], 32);
      const float mean = s / (float)rows_valid;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const long row = m0 + wm * (FBM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const float d = acc[i][j][r] - mean;
          if (row < g.M) m2 += d * d;
        }
      m2 += __shfl_xor(m2, 32);
      if (lhi == 0) {
        float* st = g.stats + ((long)mt * WM + wm) * 2 * g.N;
        st[n] = s;
        st[g.N + n] = m2;
      }
    }
  };

  // ---- main persistent pipeline ---------------------------------------------------------------
  // Producers and consumers run SEPARATE copies of the stage loop (same trip counts, one s_barrier per stage
  // in each): the register allocator then sees max(producer, consumer) live values instead of their sum.
  const bool write_e_enabled = (g.E1 != nullptr);
  auto next_stage = [&](int mt, int nt, int chunk, int& n_mt, int& n_nt, int& n_chunk) {
    n_mt = mt; n_nt = nt; n_chunk = chunk + 1;
    if (n_chunk == cpp) {
      n_chunk = 0;
      n_nt++;
      if (n_nt == g.ntn) {
        n_nt = 0;
        n_mt += g.blocks_per_xcd;
      }
    }
  };
  if (producer) {
    produce_setup(cur_mt);
    produce(0, 0, write_e_enabled && cur_nt == 0);
    stage_barrier();
    int stage = 0, chunk = 0;
    while (cur_mt < mt_end) {
      int n_mt, n_nt, n_chunk;
      next_stage(cur_mt, cur_nt, chunk, n_mt, n_nt, n_chunk);
      if (n_mt < mt_end) {
        if (n_mt != cur_mt) produce_setup(n_mt);
        produce((stage & 1) ^ 1, n_chunk, write_e_enabled && n_nt == 0);
      }
      stage_barrier();
      cur_mt = n_mt; cur_nt = n_nt; chunk = n_chunk;
      stage++;
    }
  } else {
    zero_acc();
    load_bfrag(cur_nt, 0, 0);
    stage_barrier();
    int stage = 0, chunk = 0;
    while (cur_mt < mt_end) {
      int n_mt, n_nt, n_chunk;
      next_stage(cur_mt, cur_nt, chunk, n_mt, n_nt, n_chunk);
      const bool has_next = n_mt < mt_end;
      const int bufc = stage & 1;
#pragma unroll
      for (int q = 0; q < NQ; q++) {
#pragma unroll
        for (int k4 = 0; k4 < KH / 4; k4++)
#pragma unroll
          for (int j = 0; j < GS; j++) bc[k4][j] = bn[k4][j];
        if (q + 1 < NQ)
          load_bfrag(cur_nt, chunk, q + 1);
        else if (has_next)
          load_bfrag(n_nt, n_chunk, 0);
        if (!(g.debug & 2)) mfma_chunk(bufc, q);
      }
      if (chunk == cpp - 1) {
        epilogue(cur_mt, cur_nt);
        zero_acc();
      }
      stage_barrier();
      cur_mt = n_mt; cur_nt = n_nt; chunk = n_chunk;
      stage++;
    }
  }
}

// Bm [Ktot][N] row-major -> fragment-major [Ktot/32][N/32][4][64 lanes][4]:
//   element i of lane l in group q4 of (kc, nt) = Bm[kc*32 + (q4*4 + i)*2 + (l>>5)][nt*32 + (l&31)]
__global__ void k_frag_pack(const float* __restrict__ Bm, float* __restrict__ Bpk, int Ktot, int N) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Ktot * N) return;
  const int i = (int)(idx & 3);
  const int lane = (int)((idx >> 2) & 63);
  const int q4 = (int)((idx >> 8) & 3);
  const long t = idx >> 10;
  const int ntiles = N / 32;
  const int nt = (int)(t % ntiles), kc = (int)(t / ntiles);
  const int k = kc * 32 + (q4 * 4 + i) * 2 + (lane >> 5);
  const int n = nt * 32 + (lane & 31);
  Bpk[idx] = Bm[(long)k * N + n];
}

}  // namespace p2m

using namespace p2m;

extern "C" int p2m_frag_pack(const float* Bm, float* Bpk, int32_t Ktot, int32_t N, void* stream) {
  P2M_CHECK_ARG(Bm && Bpk && Ktot > 0 && N > 0 && Ktot % 32 == 0 && N % 32 == 0, "null pointer or shape not a multiple of 32");
  const long tot = (long)Ktot * N;
  hipLaunchKernelGGL(k_frag_pack, dim3(cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, Bm, Bpk, Ktot, N);
  return check_launch("frag_pack");
}

extern "C" int p2m_cheb_gemm_fused(p2m_graph_t gh, const float* A, int32_t Ka, int32_t a_shift, const float* Bm,
                                   const float* bias, const float* addend, float* C, int32_t N, int32_t pair_out,
                                   float* stats, float* E1, float* E2, int32_t B, void* stream) {
  P2M_CHECK_ARG(gh && A && Bm && C, "null pointer");
  P2M_CHECK_ARG(Ka > 0 && Ka % FBK == 0, "Ka must be a positive multiple of 32");
  P2M_CHECK_ARG(N == 64 || N == 128 || N == 256, "N must be 64, 128 or 256");
  P2M_CHECK_ARG(a_shift == 0 || a_shift == 1, "a_shift must be 0 or 1");
  P2M_CHECK_ARG((E1 == nullptr) == (E2 == nullptr), "E1/E2 must both be given or both NULL");
  P2M_CHECK_ARG(E1 == nullptr || a_shift == 0, "basis planes can only be written with a_shift == 0");
  P2M_CHECK_ARG(!(pair_out && stats), "pair_out and stats are mutually exclusive");
  if (B <= 0) return P2M_OK;
  FusedArgs f;
  f.g = *reinterpret_cast<const Graph*>(gh);
  P2M_CHECK_ARG(a_shift == 0 || f.g.V % 2 == 0, "virtual un-pool needs an even vertex count");
  P2M_CHECK_ARG(!pair_out || f.g.V % 2 == 0, "pair_out needs an even vertex count");
  f.A = A; f.Bm = Bm; f.bias = bias; f.addend = addend; f.C = C; f.stats = stats; f.E1 = E1; f.E2 = E2;
  f.M = (long)B * f.g.V;
  f.Ka = Ka; f.N = N; f.a_shift = a_shift; f.pair_out = pair_out;
  {
    const char* dbg = getenv("P2M_FUSED_DEBUG");
    f.debug = dbg ? atoi(dbg) : 0;
  }
  const int bm = (N == 256) ? 64 : 128;        // 64 x 256 / 128 x 128 / 128 x 64 tiles: <= 32 accumulator registers per lane
  f.ntm = cdiv(f.M, bm);
  f.ntn = 1;
  f.tiles_per_xcd = cdiv(f.ntm, 8);
  f.blocks_per_xcd = f.tiles_per_xcd < 32 ? f.tiles_per_xcd : 32;
  const int grid = 8 * f.blocks_per_xcd;
  const size_t lds = (size_t)(2 * 3 * bm * FLD + 3 * NPROD * CSR_CAP) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (N == 64) {
    e = hipFuncSetAttribute((const void*)k_cheb_gemm_fused<128, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((k_cheb_gemm_fused<128, 64, 4>), dim3(grid), dim3(1024), lds, s, f);
  } else if (N == 128) {
    e = hipFuncSetAttribute((const void*)k_cheb_gemm_fused<128, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((k_cheb_gemm_fused<128, 128, 2>), dim3(grid), dim3(1024), lds, s, f);
  } else {
    e = hipFuncSetAttribute((const void*)k_cheb_gemm_fused<64, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL((k_cheb_gemm_fused<64, 256, 2>), dim3(grid), dim3(1024), lds, s, f);
  }
  if (e != hipSuccess) {
    set_error("p2m_cheb_gemm_fused: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
    return P2M_ERR_HIP;
  }
  return check_launch("cheb_gemm_fused");
}

// rows per BatchNorm partial tile written by p2m_cheb_gemm_fused for output width N (stats has 2*ceil(M/(2*rows)) rows)
extern "C" int32_t p2m_fused_stats_tile_rows(int32_t N) { return N == 128 ? 64 : 32; }
