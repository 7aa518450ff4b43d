// Chebyshev basis INSIDE the contraction: the real rows of one graph convolution in ONE kernel, no T1 / T2 planes in HBM.
//
//     C[b, v, :] = [ A0[b, v] | (L X)[b, v] | (L2 X)[b, v] ] * W  (+ bias) (+ addend)        v = a real vertex of the level
//
// Reference arithmetic: lib/models/backbones/cheby_graph_conv.py:16-37 (x1 = L x0, x2 = 2 L x1 - x0, cat, nn.Linear) and,
// because L is symmetric (lib/coarsening.py:23), its autograd backward dX = [g | L g | L2 g] W3.  Round 2 ran this as
// k_basis_tile (writes T1, T2) + k_gemm_planes_ws (reads X, T1, T2): 4 of the ~7.5 row-widths a forward conv moves
// per real row were the two planes going out and coming back.  Here a block owns one tile of the level's TilePlan
// (<= 32 consecutive real rows, the sorted union of their source rows, per entry {a, b, local index}) and S = 4 samples
// = a 128-row M tile, and walks the features in chunks of 32 (one 128-byte line of every union row):
//
//   waves 4..7  PRODUCERS   per chunk: the union-row slices of the 4 samples (prefetched into registers a chunk ahead,
//               whole 128-byte lines) go to LDS xs[u][s][32]; every lane owns (row, sample, 4 features) and accumulates
//               T1 = sum a x, T2 = sum b x from LDS in the merged-CSR entry order (the fmaf chain of k_basis_fwd /
//               k_basis_tile: bitwise their planes), takes plane 0 from global, cuts the three planes into the exact
//               bf16 slices (p2m_split.h) IN REGISTERS, and - once the MFMA waves are done with the previous chunk - stores
//               them into the A operand image As[slice][s*32 + i][96 k] (k = plane*32 + feature);
//   waves 0..3  CONSUMERS   per chunk 6 k-steps of 16: A fragments from LDS (ds_read_b128, conflict-free: 208-byte rows),
//               B fragments (the pre-split weight, p2m_weight_split layout) straight from L2 into registers one step
//               ahead, 6 slice products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//   Two LDS-only block barriers per chunk (round 6; three before): B2 - the image of the previous chunk is free and every
//   producer is done with the union rows - and B1 - the image of this chunk and the union rows of the next are stored; the
//   producers' gather / split of chunk c+1 runs under the MFMAs of chunk c.  A block walks `gpb` sample groups of its tile
//   (tables loaded once); the tile rows are dealt to the producer waves by entry count (rowsel).
//
// HBM traffic per real row: the union rows once per tile (re-use across neighbouring tiles through L2, as in
// k_basis_tile), plane 0 once (L2-warm: the row is in its own union), C once; optionally the two gathered planes
// (backward: the weight gradient X^T [g | Lg | L2g] reads them).
// (Measured and removed: a double-buffered image in half chunks of 16 features - 2 x 43 KB, entries as {a, b} pairs +
// offsets, one barrier per half and no hand-over stall: 1.821 vs 1.793 ms on the finest 128 -> 128 conv, i.e. the same;
// what bounds the kernel is the VALU issue of the gather's fmafs and the slice split next to the MFMA stream, 11.4
// instructions per MFMA against 5.5 in the plain contraction, whose basis fmafs run in a separate HBM-bound kernel.)
// LDS: 80 640 (A image; 53 760 with two fp16 slices) + 61 440 (union rows) + 15 872 (entries, rows padded to 4) + tables
// = 158.5 KB (132 KB) -> one block per CU:
// 4 MFMA waves + 8 producer waves (two per SIMD: the gather is a chain of dependent LDS reads, a second wave fills its
// latencies) for N <= 128, 4 + 4 for N = 256 (the 128-register accumulator needs the 256-register budget).
#include "p2m_split.h"

#include <cstdlib>
#include <mutex>

namespace p2m {

constexpr int CT_S = 4;                       // samples per group (M tile = 4 x 32 rows)
constexpr int CT_CF = 32;                     // features per chunk
constexpr int CT_KC = 3 * CT_CF;              // k per chunk
constexpr int CT_LDA = CT_KC + 8;             // bf16 per row of the A image: 208 bytes = 52 dwords (13 x 16 B, 13 odd ->
                                              // the 16-lane groups of ds_read_b128 tile the 64 banks exactly)
constexpr int CT_SPAD = 32;                   // + 64 bytes per sample: the two samples of a 16-lane ds_write_b64 group
                                              // land on disjoint bank halves
constexpr int CT_SLICE = CT_S * 32 * CT_LDA + CT_S * CT_SPAD;     // bf16 per slice image
constexpr int CT_ECAP = TILE_ECAP + 3 * TILE_RMAX + 8;            // entries of a tile, every row padded to a multiple of 4,
                                                                  // + slack for the gather's read-ahead
constexpr int ct_as_bytes(int ns) { return ns * CT_SLICE * 2; }   // A image: ns slices (3 bf16 / 2 fp16, p2m_split.h)
constexpr int CT_XS_BYTES = TILE_UCAP * CT_S * CT_CF * 4;
constexpr int CT_ENT_BYTES = CT_ECAP * 16;
constexpr int CT_TAB_BYTES = 1024;            // rowoff[40], rowvid[32], rowlen[32], rawoff[40]
constexpr int ct_lds_bytes(int ns) { return ct_as_bytes(ns) + CT_XS_BYTES + CT_ENT_BYTES + CT_TAB_BYTES; }
static_assert(TILE_RMAX == 32, "one MFMA tile per (tile, sample)");
static_assert(ct_lds_bytes(3) <= 160 * 1024, "LDS budget of one CU");

struct TileGemmArgs {
  TilePlan pl;
  const int* row_ids;          // [nset] compact row -> vertex id of the OUTPUT level
  const float* X;              // gather source [B][x_rows][Ka]
  const float* A0;             // plane 0       [B][a0_rows][Ka], row = row_ids[i] >> a0_shift
  const unsigned short* Bx;    // pre-split weight Bx[k / 16][slice][n][k % 16], k = plane * Ka + feature (p2m_weight_split)
  const unsigned* x_amax;      // two-fp16-slice mode: amax words of X / A0 (+ x_bits binades for the planes) and of the weight
  const unsigned* b_amax;
  unsigned* amax_out;          // optional: atomic max of |value stored|
  int x_bits;
  const float* bias;           // [N] or null
  const float* addend;         // [B][c_rows][N] or null
  const float* act_scale;      // optional fused eval-mode BatchNorm (+ ReLU), the two roundings of p2m_bn_act_fwd
  const float* act_shift;
  const float* in_scale;       // optional activation ON LOAD (both kernels): X and A0 hold the RAW output y of the previous
  const float* in_shift;       // conv and the operand is x = max(fma(y, in_scale[f], in_shift[f]), 0) - its BatchNorm + ReLU,
                               // the two roundings of p2m_bn_act_fwd - so that x is never written to / read from HBM
  float* C;                    // [B][c_rows][N]
  float* stats;                // [B][ntiles][2][N] (sum, M2 about the tile mean) or null
  float* E1;                   // [B * nset][Ka] compact planes out, or null
  float* E2;
  long x_rows, a0_rows, c_rows;
  int act_relu, a0_shift, B, Ka, N, Npad, nset, gpb;
  int a0_in_x;                 // A0 is X and a row's own source row (row id >> a0_shift) is one of its union rows (plans 0, 1)
  // BatchNorm-backward REDUCTION fused into the store of C (round 6; k_cheb_tile_gemm with the LDS-staged epilogue only).
  // In the backward pass C is a gradient g = dL/dx with x = relu(batch_norm(y)) the input of this conv, i.e. the output of
  // the conv in front of it (lib/models/backbones/cheby_graph_conv.py:39 + lib/models/meshnet.py:100): the next thing the
  // backward does is sum_r g m and sum_r g m xhat over all rows (m = [y scale + shift > 0], xhat = (y - mean) invstd).  With
  // bnr_y (that conv's raw output, C's shape) and bnr_co ([4][N]: mean, invstd, scale, shift) given, every block adds up
  // its rows' terms while it copies the staged tile out (y comes in with the same 16-byte accesses) and writes
  // bnr_part[logical block][2][N]; p2m_bn_bwd_finalize merges them with the partials of the fake-vertex rows.
  const float* bnr_y;
  const float* bnr_co;
  float* bnr_part;
};

// MODE: what the epilogue does besides bias + store - compiled in, because an epilogue that tests addend / activation /
// statistics pointers per element is ~2 900 instructions with 250 branches, and nothing covers it (one block per CU)
enum { CT_PLAIN = 0, CT_STATS = 1, CT_ADDEND = 2, CT_ACT = 3 };
// The common case of the epilogue below - a full tile (32 rows), all samples of the group present, 2^descale a normal
// float - without its per-row / per-sample predication: ~400 instead of ~1 200 instructions per wave.  The epilogue runs
// in all four MFMA waves at once with the producer waves idle, once per sample group: cycle counters in the kernel put
// the general form at a fifth of the kernel's time.  Same values: fmaf(acc, 2^descale, bias) rounds once, like
// ldexp(acc, descale) + bias.  The stores take the `global_store v_off, v_data, s[base]` form (sample base in SGPRs).
template <int TM, int TN, int MODE, int SG = CT_S>
__device__ __forceinline__ void tile_epilogue_full(const TileGemmArgs& g, const TilePlan& pl, floatx16 (&acc)[TM][TN],
                                                   const int* rowvid, int grp, int tile, int wm, int wn, int l31, int lhi,
                                                   int descale) {
  const float dsc = exp2_int(descale);
  unsigned voff[16];                              // byte offset of (row, this lane's first column) inside one sample of C
#pragma unroll
  for (int r = 0; r < 16; r++)
    voff[r] = (unsigned)(rowvid[(r & 3) + 8 * (r >> 2) + 4 * lhi] * g.N + wn * TN * 32 + l31) * 4u;
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int n = wn * TN * 32 + j * 32 + l31;
    const float bias_v = g.bias != nullptr ? g.bias[n] : 0.f;
    float sc_v = 1.f, sh_v = 0.f;
    if (MODE == CT_ACT) { sc_v = g.act_scale[n]; sh_v = g.act_shift[n]; }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = fmaf(acc[i][j][r], dsc, bias_v);
        if (MODE == CT_ACT) {
          v = fmaf(v, sc_v, sh_v);
          if (g.act_relu) v = fmaxf(v, 0.f);
        }
        acc[i][j][r] = v;
      }
  }
  // wave-uniform sample index (readfirstlane tells the compiler so): 64-bit sample bases in SGPRs, 32-bit lane offsets
  const int b0s = grp * SG + __builtin_amdgcn_readfirstlane(wm) * TM;
#pragma unroll
  for (int i = 0; i < TM; i++) {
    char* Cs = reinterpret_cast<char*>(g.C + (long)(b0s + i) * g.c_rows * g.N);
    const char* As = MODE == CT_ADDEND ? reinterpret_cast<const char*>(g.addend + (long)(b0s + i) * g.c_rows * g.N) : nullptr;
#pragma unroll
    for (int r = 0; r < 16; r++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        if (MODE == CT_ADDEND) acc[i][j][r] += *reinterpret_cast<const float*>(As + voff[r] + j * 128);
        *reinterpret_cast<float*>(Cs + voff[r] + j * 128) = acc[i][j][r];
      }
  }
  if (g.amax_out != nullptr) {
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) vmax = fmaxf(vmax, amax_abs(acc[i][j][r]));
    amax_commit(g.amax_out, vmax);
  }
  if (MODE == CT_STATS) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        float csum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) csum += acc[i][j][r];
        csum += __shfl_xor(csum, 32);
        const float mean = csum * (1.f / 32.f);
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float d = acc[i][j][r] - mean;
          m2 = fmaf(d, d, m2);
        }
        m2 += __shfl_xor(m2, 32);
        if (lhi == 0) {
          float* st = g.stats + ((long)(b0s + i) * pl.ntiles + tile) * 2 * g.N;
          const int n = wn * TN * 32 + j * 32 + l31;
          st[n] = csum;
          st[g.N + n] = m2;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
}

// Epilogue of one sample group in the MFMA waves: bias (+ activation / addend), store, BatchNorm partials; then a fresh
// accumulator.  MODE is compiled in (see above).  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) +
// 8 (reg >> 2) + 4 (lane >> 5).
// STORE = false (round 5, the LDS-staged epilogue of k_cheb_tile_gemm): values and BatchNorm partials only - the accumulator
// keeps the finished values, nothing is stored or zeroed here (the block stages them in LDS and every wave copies rows out).
template <int TM, int TN, int MODE, int SG = CT_S, bool STORE = true>
__device__ __forceinline__ void tile_epilogue(const TileGemmArgs& g, const TilePlan& pl, floatx16 (&acc)[TM][TN],
                                              const int* rowvid, int grp, int tile, int R, int wm, int wn, int l31,
                                              int lhi, int descale) {
  int voff[16];                                   // element offset of this lane's 16 accumulator rows inside one
#pragma unroll                                          // sample of C (-1: no such row); < 2^31: V * N <= 3 M elements
  for (int r = 0; r < 16; r++) {
    const int vid = rowvid[(r & 3) + 8 * (r >> 2) + 4 * lhi];
    voff[r] = vid < 0 ? -1 : vid * g.N;
  }
  // pass 1: values, in registers (no branches)
#pragma unroll
  for (int j = 0; j < TN; j++) {
    const int n = wn * TN * 32 + j * 32 + l31;
    const float bias_v = g.bias != nullptr ? g.bias[n] : 0.f;
    float sc_v = 1.f, sh_v = 0.f;
    if (MODE == CT_ACT) { sc_v = g.act_scale[n]; sh_v = g.act_shift[n]; }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float v = __builtin_ldexpf(acc[i][j][r], descale) + bias_v;     // descale: two-fp16-slice mode, else 0
        if (MODE == CT_ACT) {
          v = fmaf(v, sc_v, sh_v);
          if (g.act_relu) v = fmaxf(v, 0.f);
        }
        acc[i][j][r] = v;
      }
  }
  // pass 2: stores, accumulator row by accumulator row: the row-validity mask is the same for every tile, a
  // sample's validity is wave-uniform
  const int b0s = grp * SG + wm * TM;
  float vmax = 0.f;
  float* Cb[TM][TN];
  const float* Ab[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++) {
    const long sbase = (long)(b0s + i < g.B ? b0s + i : 0) * g.c_rows * g.N;
#pragma unroll
    for (int j = 0; j < TN; j++) {
      Cb[i][j] = g.C + sbase + wn * TN * 32 + j * 32 + l31;
      Ab[i][j] = MODE == CT_ADDEND ? g.addend + sbase + wn * TN * 32 + j * 32 + l31 : nullptr;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) {
    if (STORE && voff[r] >= 0) {
      if (MODE == CT_ADDEND) {
        float ad[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) ad[i][j] = b0s + i < g.B ? Ab[i][j][voff[r]] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
          for (int j = 0; j < TN; j++) acc[i][j][r] += ad[i][j];
      }
#pragma unroll
      for (int i = 0; i < TM; i++) {
        if (b0s + i < g.B) {
#pragma unroll
          for (int j = 0; j < TN; j++) {
            Cb[i][j][voff[r]] = acc[i][j][r];
            vmax = fmaxf(vmax, amax_abs(acc[i][j][r]));
          }
        }
      }
    }
  }
  if (STORE && g.amax_out != nullptr) amax_commit(g.amax_out, vmax);
  if (MODE == CT_STATS) {
    // column sums over the tile's rows of each sample: the other 16 rows sit in lane ^ 32
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++) {
        float csum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) csum += voff[r] >= 0 ? acc[i][j][r] : 0.f;
        csum += __shfl_xor(csum, 32);
        const float mean = csum / (float)R;
        float m2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const float d = acc[i][j][r] - mean;
          m2 += voff[r] >= 0 ? d * d : 0.f;
        }
        m2 += __shfl_xor(m2, 32);
        if (lhi == 0 && b0s + i < g.B) {
          float* st = g.stats + ((long)(b0s + i) * pl.ntiles + tile) * 2 * g.N;
          const int n = wn * TN * 32 + j * 32 + l31;
          st[n] = csum;
          st[g.N + n] = m2;
        }
      }
  }
  if (STORE)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
}

#ifdef P2M_TILE_TRACE
// Probe build only (tools/tile_trace.sh, tools/probes/tile_trace_probe.py): s_memtime stamps of one producer wave and one MFMA
// wave of one block at the phase boundaries of every unit.  [role][unit][event]
__device__ unsigned long long g_tile_trc[2][40][8];
// ... and, per WAVE of that block, the stamps at the head of a unit and where the wave arrives at B2 (the end of its gather /
// of its k-steps): which wave is the block waiting for?   [wave][unit][0 head, 1 arrival at B2]
__device__ unsigned long long g_tile_trw[12][40][2];
#define P2M_TRC(role, w, ev)                                                                          \
  do {                                                                                                \
    if (trc_on && (w) < 40) g_tile_trc[role][w][ev] = __builtin_amdgcn_s_memtime();                   \
  } while (0)
#define P2M_TRW(w, ev)                                                                                \
  do {                                                                                                \
    if (trw_on && (w) < 40) g_tile_trw[threadIdx.x >> 6][w][ev] = __builtin_amdgcn_s_memtime();       \
  } while (0)
#else
#define P2M_TRC(role, w, ev) do { } while (0)
#define P2M_TRW(w, ev) do { } while (0)
#endif
// TM x TN: MFMA tiles (samples x 32-column tiles) per consumer wave; NPW: producer waves (4: 256 registers per wave, for
// the 128-register accumulator of N = 256; 8: two producer waves per SIMD cover each other's LDS latencies)
// BNR: the BatchNorm-backward reduction fused into the copy-out (TileGemmArgs::bnr_*) - compiled in only for the launches
// that ask for it: the kernel sits at its 168-register cap, and the first version (a run-time branch in every
// instantiation) made the FORWARD launches 13 % slower through spills they never needed (gpurun bench, round 6).
template <int TM, int TN, int NPW, int MODE, int NS, bool BNR = false>
__global__ __launch_bounds__(256 + 64 * NPW, (4 + NPW) / 4) void k_cheb_tile_gemm(TileGemmArgs g) {
  typedef typename SliceFrag<NS>::type frag_t;
  constexpr int CT_AS_BYTES = ct_as_bytes(NS);
  constexpr int NT = 256 + 64 * NPW;          // 4 consumer waves + NPW producer waves
  constexpr int WM = CT_S / TM;               // consumer waves along the samples
  constexpr int WN = 4 / WM;                  // ... along the columns
  constexpr int RPP = 2 * NPW;                // tile rows the producers gather per pass (2 per wave)
  constexpr int NRP = 32 / RPP;               // gather passes = rows per producer lane
  constexpr int NPU = (TILE_UCAP + RPP - 1) / RPP;      // union-row loads per producer lane
  // LDS-staged epilogue (round 5; three-bf16-slice launches with N <= 128): the in-kernel stamps put the epilogue - 64 (+ 64
  // addend) 4-byte accesses per lane in the four MFMA waves, everything else waiting - at 15 % (forward) / 26 % (addend, planes
  // out) of the kernel.  Here the MFMA waves drop the finished values of a sample group into the A image's LDS (free between
  // B2 and the next image store) and ALL waves of the block copy whole rows out with 16-byte accesses (6 per lane; the addend
  // comes in with 16-byte loads).  Same values, same BatchNorm partials (they come from the registers, as before).
  constexpr int NCOL = WN * TN * 32;                    // columns of the block's output tile (= N)
  constexpr bool LEPI = NS == 3 && TN == 1 && CT_AS_BYTES >= CT_S * 32 * NCOL * 4 + (NT / 64) * 2 * NCOL * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char ct_smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(ct_smem);
  unsigned char* xs = ct_smem + CT_AS_BYTES;
  f32x4* ents = reinterpret_cast<f32x4*>(ct_smem + CT_AS_BYTES + CT_XS_BYTES);
  int* rowoff = reinterpret_cast<int*>(ct_smem + CT_AS_BYTES + CT_XS_BYTES + CT_ENT_BYTES);
  int* rowvid = rowoff + 40;

  const TilePlan& pl = g.pl;
  const int ngroups = (g.B + CT_S - 1) / CT_S;
  const int nbg = (ngroups + g.gpb - 1) / g.gpb;
  const int lid = xcd_contiguous(blockIdx.x, gridDim.x);
  if (lid >= pl.ntiles * nbg) return;
  const int tile = lid % pl.ntiles;           // tile fastest: the blocks of one XCD work on adjacent tiles of the same
  const int bg = lid / pl.ntiles;             // samples, whose unions overlap (served by that XCD's L2)
  const int grp0 = bg * g.gpb;
  int grp1 = grp0 + g.gpb;
  if (grp1 > ngroups) grp1 = ngroups;
  const int nchunks = g.Ka / CT_CF;
  const int nunits = (grp1 - grp0) * nchunks;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  // Entry table of the tile in LDS, every row PADDED to a multiple of 4 entries with {0, 0, row 0}: the gather loop then
  // runs whole blocks of 4 with no tail and no masks (0 * x is exact for finite x).  local index -> byte offset in xs.
  int* rowlen = rowvid + 32;                  // [32] entries per row, [33] unpadded offsets
  int* rawoff = rowlen + 32;
  if (t < 32) {
    rowlen[t] = t < R ? pl.erow[r0 + t + 1] - pl.erow[r0 + t] : 0;
    rowvid[t] = t < R ? g.row_ids[r0 + t] : -1;
  }
  __syncthreads();
  if (t <= 32) {
    int po = 0, ro = 0;
    for (int r = 0; r < t; r++) {
      po += (rowlen[r] + 3) & ~3;
      ro += rowlen[r];
    }
    rowoff[t] = po;                           // rows >= R: empty
    rawoff[t] = ro;
  }
  // Round 6: BALANCED row assignment of the gather.  A producer wave gathers two rows at a time (its two half-waves), for as
  // many 4-entry blocks as the LONGER of the two has, and the block waits at B2 for its slowest wave: with the rows dealt out
  // in tile order the traced wave waited 2 200 of a unit's 12 800 cycles there (profiles/r06_tile_phase_trace_two_barriers.txt).
  // Rows sorted by length, neighbours of the sorted order paired (equal lengths share a wave pass), pairs dealt to the
  // waves in snake order (wave w: pairs w and 2 NPW - 1 - w, ...): every wave gets the same number of blocks to within one row.
  int* rowsel = rawoff + 40;                  // [32] slot (pass * RPP + wave * 2 + half) -> tile row
  if (t >= 64 && t < 96) {                    // 32 lanes of the second wave (the first is busy above): rank sort, one row each
    const int r = t - 64;
    const int len = rowlen[r];
    int rank = 0;
#pragma unroll 8
    for (int q = 0; q < 32; q++) {
      const int lq = rowlen[q];
      rank += (lq > len || (lq == len && q < r)) ? 1 : 0;
    }
    const int pr = rank >> 1;                 // pair pr = sorted rows 2 pr, 2 pr + 1 -> (pass, wave) in snake order
    const int ps = pr / NPW, k = pr % NPW;
    const int wv = (ps & 1) ? NPW - 1 - k : k;
    rowsel[ps * RPP + wv * 2 + (rank & 1)] = r;
  }
  __syncthreads();
  {
    const int e0 = pl.erow[r0];
    for (int r = t >> 6; r < R; r += NT / 64) {
      const int eb = e0 + rawoff[r], len = rowlen[r], o = rowoff[r];
      for (int k = lane; k < ((len + 3) & ~3); k += 64) {
        f32x4 en = {0.f, 0.f, 0.f, 0.f};
        if (k < len) {
          en = *reinterpret_cast<const f32x4*>(&pl.ent[eb + k]);
          en[2] = __int_as_float(__float_as_int(en[2]) * (CT_S * CT_CF * 4));
        }
        ents[o + k] = en;
      }
    }
  }
  __syncthreads();

  const bool producer = t >= 256;             // wave-uniform
#ifdef P2M_TILE_TRACE
  const bool trc_on = lid == P2M_TILE_TRACE && (t == 0 || t == 256);
  const bool trw_on = lid == P2M_TILE_TRACE && (t & 63) == 0 && (t >> 6) < 12;
#endif
  float x_sc = 1.f;                           // two-fp16-slice mode: the planes are staged times 2^sx
  int descale = 0;
  if (NS == 2) {
    const int sx = slice_scale_exp(*g.x_amax, g.x_bits);
    x_sc = exp2_int(sx);
    descale = -(sx + slice_scale_exp(*g.b_amax, 0));
  }

  // copy-out of the staged tile of sample group egrp by every thread of the block (LEPI): piece p = (sample, row, 16-byte column
  // group); a row of C is N * 4 contiguous bytes at (sample, vertex id of the tile row)
  float bnr_run = 0.f;                        // threads t < 2 * NCOL: running sum of one (which, column) over the block's groups
  auto copy_out = [&](int egrp) {
    constexpr int C4 = NCOL / 4, NPIECE = CT_S * 32 * C4, NIT = (NPIECE + NT - 1) / NT;
    static_assert(NT % C4 == 0, "a thread's column group is the same for all of its pieces");
    float* stg = reinterpret_cast<float*>(ct_smem);
    auto piece_off = [&](int k, int& lds_at) -> long {       // global element offset of piece k of this thread, or -1
      const int p = t + k * NT;
      const int c4 = p % C4, row = (p / C4) % 32, i = p / (C4 * 32);
      const int vid = p < NPIECE ? rowvid[row] : -1;
      const int b = egrp * CT_S + i;
      lds_at = (i * 32 + row) * NCOL + c4 * 4;
      return (vid >= 0 && b < g.B) ? ((long)b * g.c_rows + vid) * g.N + c4 * 4 : -1;
    };
    {
      f32x4 v[NIT], ad[NIT];
      long off[NIT];
      int at[NIT];
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        off[k] = piece_off(k, at[k]);
        if (off[k] >= 0) {
          v[k] = *reinterpret_cast<const f32x4*>(stg + at[k]);
          if (MODE == CT_ADDEND) ad[k] = *reinterpret_cast<const f32x4*>(g.addend + off[k]);
        }
      }
      float vmax = 0.f;
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        if (off[k] >= 0) {
          if (MODE == CT_ADDEND) {
            v[k] += ad[k];
            if (BNR) *reinterpret_cast<f32x4*>(stg + at[k]) = v[k];      // (the reduction pass below re-reads the FINAL value)
          }
          *reinterpret_cast<f32x4*>(g.C + off[k]) = v[k];
#pragma unroll
          for (int c = 0; c < 4; c++) vmax = fmaxf(vmax, amax_abs(v[k][c]));
        }
      }
      if (g.amax_out != nullptr) amax_commit(g.amax_out, vmax);
    }
    if constexpr (BNR) {
      // Second pass over the thread's own pieces (its own LDS words: no barrier in between): y comes in with the same
      // 16-byte accesses, g from the staged tile - kept apart from the store pass so that its registers (24 for y) are not
      // live beside v / addend / offsets in a kernel at its register cap.  The arithmetic of k_bn_bwd_reduce (csrc/bn.hip).
      f32x4 yq[NIT];
      int at[NIT];
      bool ok[NIT];
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        const long o = piece_off(k, at[k]);
        ok[k] = o >= 0;
        if (ok[k]) yq[k] = *reinterpret_cast<const f32x4*>(g.bnr_y + o);
      }
      const int c = (t % C4) * 4;
      const f32x4 bmu = *reinterpret_cast<const f32x4*>(g.bnr_co + c);
      const f32x4 bis = *reinterpret_cast<const f32x4*>(g.bnr_co + g.N + c);
      const f32x4 bsc = *reinterpret_cast<const f32x4*>(g.bnr_co + 2 * g.N + c);
      const f32x4 bsh = *reinterpret_cast<const f32x4*>(g.bnr_co + 3 * g.N + c);
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
      for (int k = 0; k < NIT; k++) {
        if (ok[k]) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(stg + at[k]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float gm = fmaf(yq[k][e], bsc[e], bsh[e]) <= 0.f ? 0.f : gv[e];
            s0[e] += gm;
            s1[e] = fmaf(gm, (yq[k][e] - bmu[e]) * bis[e], s1[e]);
          }
        }
      }
      // lanes that share a column group (lane % C4) first, then the 12 waves through the LDS behind the staged tile
      float* red = stg + CT_S * 32 * NCOL;                                     // [NT / 64][2][NCOL]
#pragma unroll
      for (int o = 32; o >= C4; o >>= 1)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          s0[e] += __shfl_xor(s0[e], o);
          s1[e] += __shfl_xor(s1[e], o);
        }
      if (lane < C4) {
        *reinterpret_cast<f32x4*>(red + ((t >> 6) * 2 + 0) * NCOL + lane * 4) = s0;
        *reinterpret_cast<f32x4*>(red + ((t >> 6) * 2 + 1) * NCOL + lane * 4) = s1;
      }
      lds_block_barrier();
      if (t < 2 * NCOL) {
        float a = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; wv++) a += red[wv * 2 * NCOL + t];
        bnr_run += a;
      }
    }
  };

  if (producer) {
    // ------------------------------------------------------------------------------------------------------------
    const int pt = t - 256;
    const int pw = pt >> 6;                             // producer wave
    const int q = pt & 7, s = (pt >> 3) & 3;            // this lane's 4 features (16 bytes) of sample s of the group
    const int lu = pt >> 5;                             // union-row loads: rows lu, lu + RPP, ...
    const int rlo = lane >> 5;                          // gather: rows ps * RPP + pw * 2 + rlo
    const unsigned lane_off = (unsigned)((pt & 31) * 16);                     // (s, q) inside a 512-byte xs row
    unsigned uoff[NPU];                                 // byte offset of this lane's union rows inside one sample of X
#pragma unroll
    for (int ps = 0; ps < NPU; ps++) {
      const int u = lu + ps * RPP;
      uoff[ps] = (unsigned)pl.ucol[u0 + (u < U ? u : U - 1)] * (unsigned)(g.Ka * 4);      // clamped: loads stay unconditional
    }
    int ri[NRP];
    unsigned a0off[NRP];
    // (Round 6, measured and removed: plane 0 from the union image - with A0 == X a row's own source row is one of the
    //  tile's union rows, already in LDS - instead of 16 of a unit's 80 global wave-loads: xs store + loads 2 492 -> 2 286
    //  cycles, but gather 5 962 -> 6 698 and image store 1 381 -> 2 046 (two more live registers per row in a kernel at its
    //  168-register cap): 18.2 vs 18.0 ms over the step's shapes.  profiles/r06_p0_from_lds_trace.txt.  k_cheb_tile_gemm_v2
    //  below keeps the idea.)
#pragma unroll
    for (int ps = 0; ps < NRP; ps++) {
      ri[ps] = rowsel[ps * RPP + pw * 2 + rlo];
      const int vid = rowvid[ri[ps]];
      a0off[ps] = (unsigned)((vid < 0 ? 0 : vid) >> g.a0_shift) * (unsigned)(g.Ka * 4);
    }
    f32x4 pf[NPU];                                      // union-row slices in flight (unit w + 1 while unit w is gathered)
    f32x4 p0[NRP];                                      // plane-0 values of the unit being gathered
    // unit w = (sample group grp0 + w / nchunks, feature chunk w % nchunks); the loaders are called with consecutive
    // units, so each keeps its own (group, chunk) counters instead of dividing
    auto sample_of = [&](int grp) {                     // clamped: a group's missing samples recompute the last one
      const int b = grp * CT_S + s;
      return b < g.B ? b : g.B - 1;
    };
    int ug = grp0, ufc = 0;                             // next unit of load_union
    auto load_union = [&]() {
      const char* base = reinterpret_cast<const char*>(g.X + ((long)sample_of(ug) * g.x_rows) * g.Ka + ufc * CT_CF + q * 4);
#pragma unroll
      for (int ps = 0; ps < NPU; ps++) pf[ps] = *reinterpret_cast<const f32x4*>(base + uoff[ps]);
      if (++ufc == nchunks) { ufc = 0; ug++; }
    };
    int pg = grp0, pfc = 0;                             // next unit of load_p0
    auto load_p0 = [&]() {
      const char* base = reinterpret_cast<const char*>(g.A0 + ((long)sample_of(pg) * g.a0_rows) * g.Ka + pfc * CT_CF + q * 4);
#pragma unroll
      for (int ps = 0; ps < NRP; ps++) p0[ps] = *reinterpret_cast<const f32x4*>(base + a0off[ps]);
      if (++pfc == nchunks) { pfc = 0; pg++; }
    };
    // activation on load (in_scale != nullptr; round 5: also in this kernel, i.e. in both slice arithmetics): X / A0 hold the
    // RAW output y of the previous conv; the union rows get max(fma(y, scale, shift), 0) on their way into LDS, plane 0 in
    // front of its split - the two roundings of p2m_bn_act_fwd.  store_xs walks the units 0, 1, 2, ... : own chunk counter.
    const bool in_act = g.in_scale != nullptr;          // block-uniform
    int sfc_x = 0;
    auto act_coeffs = [&](int fc, f32x4& sc, f32x4& sh) {
      sc = *reinterpret_cast<const f32x4*>(g.in_scale + fc * CT_CF + q * 4);
      sh = *reinterpret_cast<const f32x4*>(g.in_shift + fc * CT_CF + q * 4);
    };
    auto store_xs = [&]() {
      if (in_act) {
        f32x4 sc, sh;
        act_coeffs(sfc_x, sc, sh);
#pragma unroll
        for (int ps = 0; ps < NPU; ps++)
#pragma unroll
          for (int c = 0; c < 4; c++) pf[ps][c] = fmaxf(fmaf(pf[ps][c], sc[c], sh[c]), 0.f);
        if (++sfc_x == nchunks) sfc_x = 0;
      }
#pragma unroll
      for (int ps = 0; ps < NPU; ps++) {
        const int u = lu + ps * RPP;
        if (u < U) *reinterpret_cast<f32x4*>(xs + u * (CT_S * CT_CF * 4) + lane_off) = pf[ps];
      }
    };
    load_union();
    load_p0();
    store_xs();
    if (nunits > 1) load_union();
    lds_block_barrier();                                // B1(-1): xs(0) visible
    int grp = grp0, fc = 0;
    for (int w = 0; w < nunits; w++) {
      P2M_TRC(0, w, 0);
      P2M_TRW(w, 0);
      u32x2 sp[NRP][3][NS];                             // [row][plane][slice]: the A operand of this unit, held until the
                                                        // MFMA waves release the image
      if (in_act) {                                     // plane 0 of unit w = (grp, fc)
        f32x4 sc, sh;
        act_coeffs(fc, sc, sh);
#pragma unroll
        for (int ps = 0; ps < NRP; ps++)
#pragma unroll
          for (int c = 0; c < 4; c++) p0[ps][c] = fmaxf(fmaf(p0[ps][c], sc[c], sh[c]), 0.f);
      }
#pragma unroll
      for (int ps = 0; ps < NRP; ps++) {
        f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
        const int i = ri[ps];
        const int e = rowoff[i + 1];
        int j = rowoff[i];
        // (round 5, measured with the stamps above: this phase is ~6 000 of a unit's ~12 200 cycles and the MFMA waves wait
        //  ~3 500 for it - it sits at the rate of its ds_read_b128s, not at latency or VALU issue: two rows of a lane
        //  interleaved (two independent chains) gathered 7 % faster and made the kernel 2.5 % slower; a third fewer fmafs
        //  changed nothing.  profiles/r05_tile_phase_trace.txt)
        f32x4 en[4];                                    // the block of 4 entries being used; the next one is fetched under
#pragma unroll                                          // its FMAs (reading past the row's end is harmless: the table is
        for (int k = 0; k < 4; k++) en[k] = ents[j + k];   // contiguous and has 4 entries of slack)
        for (; j < e; j += 4) {
          f32x4 x[4], nn[4];
#pragma unroll
          for (int k = 0; k < 4; k++)
            x[k] = *reinterpret_cast<const f32x4*>(xs + (unsigned)__float_as_int(en[k][2]) + lane_off);
#pragma unroll
          for (int k = 0; k < 4; k++) nn[k] = ents[j + 4 + k];
#pragma unroll
          for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
              t1[c] = fmaf(en[k][0], x[k][c], t1[c]);
              t2[c] = fmaf(en[k][1], x[k][c], t2[c]);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) en[k] = nn[k];
        }
        if (g.E1 != nullptr && i < R) {
          const int b = grp * CT_S + s;
          if (b < g.B) {
            const long o = ((long)b * g.nset + r0 + i) * g.Ka + fc * CT_CF + q * 4;
            __builtin_nontemporal_store(t1, reinterpret_cast<f32x4*>(g.E1 + o));
            __builtin_nontemporal_store(t2, reinterpret_cast<f32x4*>(g.E2 + o));
          }
        }
        split_pack4<NS>(p0[ps][0], p0[ps][1], p0[ps][2], p0[ps][3], x_sc, sp[ps][0]);
        split_pack4<NS>(t1[0], t1[1], t1[2], t1[3], x_sc, sp[ps][1]);
        split_pack4<NS>(t2[0], t2[1], t2[2], t2[3], x_sc, sp[ps][2]);
      }
      P2M_TRC(0, w, 1);
      P2M_TRW(w, 1);
      lds_block_barrier();                              // B2(w): the MFMA waves are done with the image of unit w - 1,
                                                        //        every producer is done reading xs(w)
      P2M_TRC(0, w, 2);
      if (LEPI && w > 0 && fc == 0) {                   // the unit before closed a sample group: its tile is being staged
        lds_block_barrier();                            // E1: staged
        copy_out(grp - 1);
        lds_block_barrier();                            // E2: every wave has read its pieces - the image may overwrite them
      }
#pragma unroll
      for (int ps = 0; ps < NRP; ps++) {
        unsigned short* d = As + (s * 32 + ri[ps]) * CT_LDA + s * CT_SPAD + q * 4;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
          for (int sl = 0; sl < NS; sl++) *reinterpret_cast<u32x2*>(d + sl * CT_SLICE + p * CT_CF) = sp[ps][p][sl];
      }
      P2M_TRC(0, w, 3);
      if (w + 1 < nunits) {
        store_xs();                                     // xs(w + 1) from the registers loaded a unit ago (xs(w) is free: B2)
        load_p0();
        if (w + 2 < nunits) load_union();
      }
      if (++fc == nchunks) { fc = 0; grp++; }
      P2M_TRC(0, w, 4);
      lds_block_barrier();                              // B1(w): image of unit w AND xs(w + 1) visible
      P2M_TRC(0, w, 5);
      P2M_TRC(0, w, 6);
    }
    if (LEPI) {                                         // the last unit's tile
      lds_block_barrier();                              // F0: the MFMA waves are done with the last image
      lds_block_barrier();                              // E1
      copy_out(grp - 1);
    }
  } else {
    // ------------------------------------------------------------------------------------------------------------
    const int wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // B fragments: byte address = Bx + ((plane * Ka/16 + fc * 2 + half) * NS + slice) * Npad * 32 + (n * 16 + lhi * 8) * 2
    const long bx_slice = (long)g.Npad * 32;            // bytes per slice of one 16-wide k chunk
    const long bx_plane = (long)(g.Ka / 16) * NS * bx_slice;
    const char* bx_lane = reinterpret_cast<const char*>(g.Bx) + ((wn * TN * 32 + l31) * 16 + lhi * 8) * 2;
    constexpr int NB = TN == 1 ? 3 : 2;                 // ring of B fragment sets: the fragments of step st + NB - 1 are loaded
    frag_t fb[NB][NS][TN];                              // during step st (N = 256: no registers for a third set)
    auto load_b = [&](int fc, int st, frag_t (&b)[NS][TN]) {  // step st of chunk fc: plane st / 2, half st & 1
      const char* src = bx_lane + (st >> 1) * bx_plane + (long)(fc * 2 + (st & 1)) * NS * bx_slice;
#pragma unroll
      for (int sl = 0; sl < NS; sl++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          b[sl][j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(src + sl * bx_slice + j * (32 * 16 * 2)));
    };
    const unsigned short* a_lane = As + ((wm * TM) * 32 + l31) * CT_LDA + (wm * TM) * CT_SPAD + lhi * 8;
    auto read_a = [&](int sl, int st, frag_t (&a)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; i++)
        a[i] = __builtin_bit_cast(
            frag_t, *reinterpret_cast<const u32x4*>(a_lane + sl * CT_SLICE + i * (32 * CT_LDA + CT_SPAD) + st * 16));
    };
    auto stage_acc = [&]() {                            // LEPI: finished values -> LDS [sample][row][column], fresh accumulator
      float* stg = reinterpret_cast<float*>(ct_smem);
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            stg[((wm * TM + i) * 32 + row) * NCOL + wn * TN * 32 + j * 32 + l31] = acc[i][j][r];
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
    load_b(0, 0, fb[0]);
    if (NB == 3) load_b(0, 1, fb[1]);
    lds_block_barrier();                                // B1(-1)
    int grp = grp0, fc = 0;
    for (int w = 0; w < nunits; w++) {
      const int fcn = fc + 1 == nchunks ? 0 : fc + 1;
      P2M_TRC(1, w, 0);
      P2M_TRW(w, 1);                                    // (the wave arrives at B2(w) with the k-steps of unit w - 1 behind it)
      lds_block_barrier();                              // B2(w)
      P2M_TRC(1, w, 1);
      if (LEPI && w > 0 && fc == 0) {                   // (grp was advanced by the epilogue of the unit before)
        stage_acc();
        lds_block_barrier();                            // E1
        copy_out(grp - 1);
        zero_acc();
        lds_block_barrier();                            // E2
      }
      lds_block_barrier();                              // B1(w): image of unit w is in LDS
      P2M_TRC(1, w, 2);
      frag_t fl[TM];                                    // the low-slice A fragments: what the first MFMAs of a step read
      read_a(NS - 1, 0, fl);
#pragma unroll
      for (int st = 0; st < 6; st++) {
        // B fragments NB - 1 steps ahead (they do not depend on the producers; the last steps fetch the next unit's first).
        // sched_barrier: the loads must be ISSUED here - left alone, the scheduler sinks them to just above the first
        // MFMA that reads them and the L2 latency (longer than one step's MFMAs under load) is exposed at every step
        constexpr int AH = NB - 1;                      // steps of lead; 6 % NB == 0: the ring position of a step is st % NB
        if (st + AH < 6) load_b(fc, st + AH, fb[(st + AH) % NB]);
        else load_b(fcn, st + AH - 6, fb[(st + AH) % NB]);
        __builtin_amdgcn_sched_barrier(0);
        frag_t fh[TM], fm[TM];
        read_a(0, st, fh);
        if constexpr (NS == 3) read_a(1, st, fm);
#define P2M_PAIR(FA, SB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) \
      acc[i][j] = slice_mfma<NS>(FA[i], fb[st % NB][SB][j], acc[i][j]);
        P2M_PAIR(fl, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (st < 5) read_a(NS - 1, st + 1, fl);         // the next step's first operands, under this step's other products
        if constexpr (NS == 3) {
          P2M_PAIR(fh, 2)
          P2M_PAIR(fm, 1)
          P2M_PAIR(fm, 0)
          P2M_PAIR(fh, 1)
          P2M_PAIR(fh, 0)
        } else {
          P2M_PAIR(fh, 1)
          P2M_PAIR(fh, 0)
        }
#undef P2M_PAIR
        __builtin_amdgcn_sched_barrier(0);
      }
      P2M_TRC(1, w, 3);
      if (fc == nchunks - 1) {
        tile_epilogue<TM, TN, MODE, CT_S, !LEPI>(g, pl, acc, rowvid, grp, tile, R, wm, wn, l31, lhi, descale);
        grp++;
      }
      P2M_TRC(1, w, 4);
      fc = fcn;
    }
    if (LEPI) {
      lds_block_barrier();                              // F0
      stage_acc();
      lds_block_barrier();                              // E1
      copy_out(grp - 1);
      if (BNR && t < 2 * NCOL) g.bnr_part[(long)lid * 2 * g.N + t] = bnr_run;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_cheb_tile_gemm_v2 (round 6; N <= 128): the same unit of work - one tile x 4 samples x 32 features -, restructured
// around what the in-kernel phase trace of round 5 showed (profiles/r05_tile_phase_trace.txt): the unit's critical path ran
// through the PRODUCER waves - after the gather (6 000 cycles) and the image store (1 400) they spent 2 500 more issuing the
// next unit's 80 global loads (1 KB per wave-instruction, ~24 cycles each through the CU's vector-memory path) and storing
// the union rows into LDS, with the four MFMA waves idle 5 200 of the unit's 12 200 cycles (and a first attempt of this round,
// the same loads issued at the head of the gather by the producers themselves, made the gather 3 300 cycles longer).  Here:
//
//   * the union rows come in by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), 1 KB pieces = 4
//     union rows x 2 samples x 128 B, lane-linear in LDS, four pieces per producer wave and half, issued where a half
//     becomes free (first version: by the MFMA waves, the ones with slack - but every counted wait hipcc emits for their
//     weight-fragment loads then also waits for the pieces queued in front, and the first two k-steps of a unit took 4 200
//     cycles instead of 2 000);
//   * the union image is double-buffered IN THE SAMPLE DIMENSION at no extra LDS: two halves H0 / H1 of 120 rows x 2 samples
//     x 128 B (30 KB each; round 5: one image of 4 samples).  A producer lane owns (row, sample of the half, 4 features) -
//     512 lanes = 32 rows x 2 samples x 8 column quads - and gathers one row per half: h0 from H0, barrier MID, h1 from H1.
//     H0 is refilled (next unit) from MID on, H1 from B2 on - both under the other half's gather / the image store;
//   * plane 0 is read from the union image (a row is in its own union: the diagonal of L) instead of from global - 16 of a
//     unit's 80 KB; the un-paired plans only (plan 2 takes A0 from another tensor: global, as before);
//   * activation on load (in_scale) is applied where x is READ in the gather (the DMA cannot transform), the same two
//     roundings per element;
//   * three block barriers per unit (MID, B2, B1), none of them with a wave doing global-load issue behind it.
// Per unit and lane the producers now do: 2 x (gather of one row, three split_pack4, E1 / E2 out) + 18 ds_write_b64.
// Planes bitwise k_basis_tile's (same entry order, same fmaf chain); C as before (same k order in the MFMA waves).
// The LDS-DMA is inline asm (hipcc would drain vmcnt(0) at the next global-load use while one is in flight,
// /opt/skills/guides/cdna_hip_programming.md "Pipelining across barriers"): its completion is counted by hand - loads
// retire in order, every wave issues EXACTLY 4 pieces per half, so `vmcnt(4)` after the H1 burst proves the H0 pieces landed;
// the issuing waves hold no other global load in flight.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CT2_XH_BYTES = TILE_UCAP * 2 * CT_CF * 4;                // one half of the union image: 30 720 B
constexpr int CT2_ROWB = 2 * CT_CF * 4;                                // bytes per union row of a half (2 samples x 128 B)
constexpr int CT2_NPIECE = TILE_UCAP / 4;                              // 1 KB pieces per half (4 union rows each): 30
constexpr int CT2_TAB_BYTES = 1536 + 2048;    // rowoff[40], rowvid[32], rowlen[32], rawoff[40], diag[32], ucolb[120]; at 1536:
                                              // in_scale[256], in_shift[256] (activation on load: no global load in the loop)
constexpr int ct2_lds_bytes(int ns) { return ct_as_bytes(ns) + 2 * CT2_XH_BYTES + CT_ENT_BYTES + CT2_TAB_BYTES; }
static_assert(TILE_UCAP % 4 == 0 && CT2_NPIECE <= 32, "4 pieces per producer wave and half");
static_assert(ct2_lds_bytes(3) <= 160 * 1024, "LDS budget of one CU");

// one LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KB of LDS at lds_dst (wave-uniform byte address)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int TM, int TN, int MODE, int NS>
__global__ __launch_bounds__(768, 3) void k_cheb_tile_gemm_v2(TileGemmArgs g) {
  typedef typename SliceFrag<NS>::type frag_t;
  constexpr int CT_AS_BYTES = ct_as_bytes(NS);
  constexpr int NT = 768;                      // 4 MFMA waves + 8 producer waves
  constexpr int WM = CT_S / TM;
  constexpr int WN = 4 / WM;
  constexpr int NCOL = WN * TN * 32;
  constexpr bool LEPI = NS == 3 && TN == 1 && CT_AS_BYTES >= CT_S * 32 * NCOL * 4;
  constexpr int MIDST = 2;                     // the MFMA waves pass MID in front of this k-step
  extern __shared__ __attribute__((aligned(16))) unsigned char ct_smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(ct_smem);
  unsigned char* xs = ct_smem + CT_AS_BYTES;                   // H0 | H1
  f32x4* ents = reinterpret_cast<f32x4*>(ct_smem + CT_AS_BYTES + 2 * CT2_XH_BYTES);
  int* rowoff = reinterpret_cast<int*>(ct_smem + CT_AS_BYTES + 2 * CT2_XH_BYTES + CT_ENT_BYTES);
  int* rowvid = rowoff + 40;
  int* rowlen = rowvid + 32;
  int* rawoff = rowlen + 32;
  int* diag = rawoff + 40;                     // [32] local index of a row's own source row in the union (-1: none)
  unsigned* ucolb = reinterpret_cast<unsigned*>(diag + 32);    // [120] byte offset of union row u inside one sample of X
  float* actc = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(rowoff) + 1536);     // [2][256]

  const TilePlan& pl = g.pl;
  const int ngroups = (g.B + CT_S - 1) / CT_S;
  const int nbg = (ngroups + g.gpb - 1) / g.gpb;
  const int lid = xcd_contiguous(blockIdx.x, gridDim.x);
  if (lid >= pl.ntiles * nbg) return;
  const int tile = lid % pl.ntiles;
  const int bg = lid / pl.ntiles;
  const int grp0 = bg * g.gpb;
  int grp1 = grp0 + g.gpb;
  if (grp1 > ngroups) grp1 = ngroups;
  const int nchunks = g.Ka / CT_CF;
  const int nunits = (grp1 - grp0) * nchunks;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  if (t < 32) {
    rowlen[t] = t < R ? pl.erow[r0 + t + 1] - pl.erow[r0 + t] : 0;
    const int vid = t < R ? g.row_ids[r0 + t] : -1;
    rowvid[t] = vid;
    int dg = -1;
    if (vid >= 0 && g.a0_in_x) {               // the row's own source row (vid >> a0_shift) in the sorted union
      const int want = vid >> g.a0_shift;
      int lo = 0, hi = U - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1, c = pl.ucol[u0 + mid];
        if (c == want) { dg = mid; break; }
        if (c < want) lo = mid + 1; else hi = mid - 1;
      }
    }
    diag[t] = dg;
  }
  for (int u = t; u < TILE_UCAP; u += NT) ucolb[u] = (unsigned)pl.ucol[u0 + (u < U ? u : U - 1)] * (unsigned)(g.Ka * 4);
  if (g.in_scale != nullptr && t < g.Ka) {
    actc[t] = g.in_scale[t];
    actc[256 + t] = g.in_shift[t];
  }
  __syncthreads();
  if (t <= 32) {
    int po = 0, ro = 0;
    for (int r = 0; r < t; r++) {
      po += (rowlen[r] + 3) & ~3;
      ro += rowlen[r];
    }
    rowoff[t] = po;
    rawoff[t] = ro;
  }
  __syncthreads();
  {
    const int e0 = pl.erow[r0];
    for (int r = t >> 6; r < R; r += NT / 64) {
      const int eb = e0 + rawoff[r], len = rowlen[r], o = rowoff[r];
      for (int k = lane; k < ((len + 3) & ~3); k += 64) {
        f32x4 en = {0.f, 0.f, 0.f, 0.f};
        if (k < len) {
          en = *reinterpret_cast<const f32x4*>(&pl.ent[eb + k]);
          en[2] = __int_as_float(__float_as_int(en[2]) * CT2_ROWB);          // local index -> byte offset in a half
        }
        ents[o + k] = en;
      }
    }
  }
  __syncthreads();

  const bool producer = t >= 256;
#ifdef P2M_TILE_TRACE
  const bool trc_on = lid == P2M_TILE_TRACE && (t == 0 || t == 256);
  const bool trw_on = false;
  (void)trw_on;
#endif
  float x_sc = 1.f;
  int descale = 0;
  if (NS == 2) {
    const int sx = slice_scale_exp(*g.x_amax, g.x_bits);
    x_sc = exp2_int(sx);
    descale = -(sx + slice_scale_exp(*g.b_amax, 0));
  }

  auto copy_out = [&](int egrp) {
    constexpr int C4 = NCOL / 4, NPIECE = CT_S * 32 * C4, NIT = (NPIECE + NT - 1) / NT;
    const float* stg = reinterpret_cast<const float*>(ct_smem);
    f32x4 v[NIT], ad[NIT];
    long off[NIT];
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      const int p = t + k * NT;
      const int c4 = p % C4, row = (p / C4) % 32, i = p / (C4 * 32);
      const int vid = p < NPIECE ? rowvid[row] : -1;
      const int b = egrp * CT_S + i;
      off[k] = (vid >= 0 && b < g.B) ? ((long)b * g.c_rows + vid) * g.N + c4 * 4 : -1;
      if (off[k] >= 0) {
        v[k] = *reinterpret_cast<const f32x4*>(stg + (i * 32 + row) * NCOL + c4 * 4);
        if (MODE == CT_ADDEND) ad[k] = *reinterpret_cast<const f32x4*>(g.addend + off[k]);
      }
    }
    float vmax = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; k++) {
      if (off[k] >= 0) {
        if (MODE == CT_ADDEND) v[k] += ad[k];
        *reinterpret_cast<f32x4*>(g.C + off[k]) = v[k];
#pragma unroll
        for (int c = 0; c < 4; c++) vmax = fmaxf(vmax, amax_abs(v[k][c]));
      }
    }
    if (g.amax_out != nullptr) amax_commit(g.amax_out, vmax);
  };

  if (producer) {
    // ------------------------------------------------------------------------------------------------------------
    const int pt = t - 256;
    const int q = pt & 7, s2 = (pt >> 3) & 1;           // this lane's 4 features (16 bytes) of sample s2 of a half
    const int i = pt >> 4;                              // ... of tile row i: 4 rows per wave, 16 lanes per row
    const unsigned lane_off = (unsigned)(s2 * (CT_CF * 4) + q * 16);          // inside a 256-byte row of a half
    const int vid = rowvid[i];
    const int dg = diag[i];
    const int j0 = rowoff[i], je = rowoff[i + 1];
    const unsigned a0off = (unsigned)((vid < 0 ? 0 : vid) >> g.a0_shift) * (unsigned)(g.Ka * 4);
    const bool in_act = g.in_scale != nullptr;          // block-uniform
    // ---- LDS-DMA of the union rows: piece p of half h = union rows 4p .. 4p + 3, lane = (row >> 4, sample, quad).  Producer
    // wave pw issues pieces pw, pw + 8, pw + 16, pw + 24 (clamped: a constant FOUR per wave and half, so that the counted
    // waits below are exact).  These waves hold no other global load in flight in the loop (plane 0 and the activation
    // coefficients come from LDS), so nothing hipcc counts is queued behind a piece.
    const int pw = __builtin_amdgcn_readfirstlane(pt >> 6);
    const unsigned xs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)xs;
    const int du = lane >> 4;
    auto dma_half = [&](int h, int dgrp, int dfc) {
      int b = dgrp * CT_S + h * 2 + s2;
      b = b < g.B ? b : g.B - 1;
      const char* base = reinterpret_cast<const char*>(g.X) + ((long)b * g.x_rows) * (g.Ka * 4) + dfc * (CT_CF * 4) + q * 16;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int p = pw + 8 * k;
        p = p < CT2_NPIECE ? p : CT2_NPIECE - 1;
        glds16(base + ucolb[p * 4 + du], xs_lds + h * CT2_XH_BYTES + p * 1024);
      }
    };
    int d0g = grp0, d0f = 0, d1g = grp0, d1f = 0;       // next unit of the H0 / H1 stream
    auto adv = [&](int& dg_, int& df_) { if (++df_ == nchunks) { df_ = 0; dg_++; } };
    dma_half(0, d0g, d0f);
    dma_half(1, d1g, d1f);
    adv(d0g, d0f);
    adv(d1g, d1f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // both halves of unit 0 landed
    lds_block_barrier();                                // B1(-1)
    int grp = grp0, fc = 0;
    for (int w = 0; w < nunits; w++) {
      P2M_TRC(0, w, 0);
      u32x2 sp[2][3][NS];                               // [half][plane][slice]: this lane's part of the A image of unit w
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (in_act) {
        sc = *reinterpret_cast<const f32x4*>(actc + fc * CT_CF + q * 4);
        sh = *reinterpret_cast<const f32x4*>(actc + 256 + fc * CT_CF + q * 4);
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const unsigned char* xh = xs + h * CT2_XH_BYTES + lane_off;
        const int b = grp * CT_S + h * 2 + s2;
        f32x4 p0 = {0.f, 0.f, 0.f, 0.f};
        if (dg >= 0) p0 = *reinterpret_cast<const f32x4*>(xh + dg * CT2_ROWB);
        else if (vid >= 0)
          p0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(g.A0) +
                                               ((long)(b < g.B ? b : g.B - 1) * g.a0_rows) * (g.Ka * 4) + a0off +
                                               fc * (CT_CF * 4) + q * 16);
        f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
        int j = j0;
        f32x4 en[4];
#pragma unroll
        for (int k = 0; k < 4; k++) en[k] = ents[j + k];
        for (; j < je; j += 4) {
          f32x4 x[4], nn[4];
#pragma unroll
          for (int k = 0; k < 4; k++)
            x[k] = *reinterpret_cast<const f32x4*>(xh + (unsigned)__float_as_int(en[k][2]));
#pragma unroll
          for (int k = 0; k < 4; k++) nn[k] = ents[j + 4 + k];
          if (in_act) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
              for (int c = 0; c < 4; c++) x[k][c] = fmaxf(fmaf(x[k][c], sc[c], sh[c]), 0.f);
          }
#pragma unroll
          for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
              t1[c] = fmaf(en[k][0], x[k][c], t1[c]);
              t2[c] = fmaf(en[k][1], x[k][c], t2[c]);
            }
          }
#pragma unroll
          for (int k = 0; k < 4; k++) en[k] = nn[k];
        }
        auto planes_out = [&]() {
          if (g.E1 != nullptr && i < R && b < g.B) {
            const long o = ((long)b * g.nset + r0 + i) * g.Ka + fc * CT_CF + q * 4;
            __builtin_nontemporal_store(t1, reinterpret_cast<f32x4*>(g.E1 + o));
            __builtin_nontemporal_store(t2, reinterpret_cast<f32x4*>(g.E2 + o));
          }
        };
        if (h == 1) planes_out();                       // (half 0: behind MID - no store in flight at its counted wait)
        if (in_act) {
#pragma unroll
          for (int c = 0; c < 4; c++) p0[c] = fmaxf(fmaf(p0[c], sc[c], sh[c]), 0.f);
        }
        split_pack4<NS>(p0[0], p0[1], p0[2], p0[3], x_sc, sp[h][0]);
        split_pack4<NS>(t1[0], t1[1], t1[2], t1[3], x_sc, sp[h][1]);
        split_pack4<NS>(t2[0], t2[1], t2[2], t2[3], x_sc, sp[h][2]);
        if (h == 0) {
          P2M_TRC(0, w, 1);
          // H1(w) landed: its pieces (issued behind B2(w - 1)) are this wave's only loads in flight
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          lds_block_barrier();                          // MID(w): every producer is done with H0(w) - it may be refilled
          if (w + 1 < nunits) {
            dma_half(0, d0g, d0f);                      // H0(w + 1), under the gather of half 1 and the image store
            adv(d0g, d0f);
          }
          planes_out();
          P2M_TRC(0, w, 2);
        }
      }
      P2M_TRC(0, w, 3);
      lds_block_barrier();                              // B2(w): the MFMA waves are done with the image of unit w - 1,
                                                        //        every producer is done with H1(w)
      if (w + 1 < nunits) {
        dma_half(1, d1g, d1f);                          // H1(w + 1), under the image store and the next gather of half 0
        adv(d1g, d1f);
      }
      P2M_TRC(0, w, 4);
      if (LEPI && w > 0 && fc == 0) {
        lds_block_barrier();                            // E1: staged
        copy_out(grp - 1);
        lds_block_barrier();                            // E2
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int s = h * 2 + s2;
        unsigned short* d = As + (s * 32 + i) * CT_LDA + s * CT_SPAD + q * 4;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
          for (int sl = 0; sl < NS; sl++) *reinterpret_cast<u32x2*>(d + sl * CT_SLICE + p * CT_CF) = sp[h][p][sl];
      }
      if (++fc == nchunks) { fc = 0; grp++; }
      P2M_TRC(0, w, 5);
      // H0(w + 1) landed: loads retire in order and the 4 pieces of H1(w + 1) are the only younger ones
      if (w + 1 < nunits) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      lds_block_barrier();                              // B1(w): image of unit w visible; H0(w + 1) landed
      P2M_TRC(0, w, 6);
    }
    if (LEPI) {
      lds_block_barrier();                              // F0
      lds_block_barrier();                              // E1
      copy_out(grp - 1);
    }
  } else {
    // ------------------------------------------------------------------------------------------------------------
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const long bx_slice = (long)g.Npad * 32;
    const long bx_plane = (long)(g.Ka / 16) * NS * bx_slice;
    const char* bx_lane = reinterpret_cast<const char*>(g.Bx) + ((wn * TN * 32 + l31) * 16 + lhi * 8) * 2;
    constexpr int NB = TN == 1 ? 3 : 2;
    frag_t fb[NB][NS][TN];
    auto load_b = [&](int fc, int st, frag_t (&b)[NS][TN]) {
      const char* src = bx_lane + (st >> 1) * bx_plane + (long)(fc * 2 + (st & 1)) * NS * bx_slice;
#pragma unroll
      for (int sl = 0; sl < NS; sl++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          b[sl][j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(src + sl * bx_slice + j * (32 * 16 * 2)));
    };
    const unsigned short* a_lane = As + ((wm * TM) * 32 + l31) * CT_LDA + (wm * TM) * CT_SPAD + lhi * 8;
    auto read_a = [&](int sl, int st, frag_t (&a)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; i++)
        a[i] = __builtin_bit_cast(
            frag_t, *reinterpret_cast<const u32x4*>(a_lane + sl * CT_SLICE + i * (32 * CT_LDA + CT_SPAD) + st * 16));
    };
    auto stage_acc = [&]() {
      float* stg = reinterpret_cast<float*>(ct_smem);
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            stg[((wm * TM + i) * 32 + row) * NCOL + wn * TN * 32 + j * 32 + l31] = acc[i][j][r];
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
    load_b(0, 0, fb[0]);
    if (NB == 3) load_b(0, 1, fb[1]);
    lds_block_barrier();                                // B1(-1)
    // gather(0) runs now; nothing to multiply yet
    lds_block_barrier();                                // MID(0)
    int grp = grp0, fc = 0;
    for (int w = 0; w < nunits; w++) {
      const int fcn = fc + 1 == nchunks ? 0 : fc + 1;
      P2M_TRC(1, w, 0);
      lds_block_barrier();                              // B2(w): H1(w) is free
      P2M_TRC(1, w, 1);
      const bool more1 = w + 1 < nunits;                // block-uniform
      if (LEPI && w > 0 && fc == 0) {
        stage_acc();
        lds_block_barrier();                            // E1
        copy_out(grp - 1);
        zero_acc();
        lds_block_barrier();                            // E2
      }
      lds_block_barrier();                              // B1(w): image of unit w is in LDS
      P2M_TRC(1, w, 2);
      frag_t fl[TM];
      read_a(NS - 1, 0, fl);
#pragma unroll
      for (int st = 0; st < 6; st++) {
        if (st == MIDST && more1) lds_block_barrier();  // MID(w + 1) (the producers' hand-over of H0; block-wide barrier)
        constexpr int AH = NB - 1;
        if (st + AH < 6) load_b(fc, st + AH, fb[(st + AH) % NB]);
        else load_b(fcn, st + AH - 6, fb[(st + AH) % NB]);
        __builtin_amdgcn_sched_barrier(0);
        frag_t fh[TM], fm[TM];
        read_a(0, st, fh);
        if constexpr (NS == 3) read_a(1, st, fm);
#define P2M_PAIR(FA, SB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) \
      acc[i][j] = slice_mfma<NS>(FA[i], fb[st % NB][SB][j], acc[i][j]);
        P2M_PAIR(fl, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (st < 5) read_a(NS - 1, st + 1, fl);
        if constexpr (NS == 3) {
          P2M_PAIR(fh, 2)
          P2M_PAIR(fm, 1)
          P2M_PAIR(fm, 0)
          P2M_PAIR(fh, 1)
          P2M_PAIR(fh, 0)
        } else {
          P2M_PAIR(fh, 1)
          P2M_PAIR(fh, 0)
        }
#undef P2M_PAIR
        __builtin_amdgcn_sched_barrier(0);
      }
      P2M_TRC(1, w, 3);
      if (fc == nchunks - 1) {
        tile_epilogue<TM, TN, MODE, CT_S, !LEPI>(g, pl, acc, rowvid, grp, tile, R, wm, wn, l31, lhi, descale);
        grp++;
      }
      P2M_TRC(1, w, 4);
      fc = fcn;
    }
    if (LEPI) {
      lds_block_barrier();                              // F0
      stage_acc();
      lds_block_barrier();                              // E1
      copy_out(grp - 1);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same contraction with the GATHER ON THE MATRIX CORES (two-fp16-slice arithmetic only).
//
// k_cheb_tile_gemm above forms L X and L2 X with v_fma chains in its producer waves: 16 fmafs per merged-CSR entry and
// lane, 11 VALU instructions per MFMA - the kernel is bound by VALU issue, not by the matrix pipe or HBM.  Here the tile's
// operator is a DENSE block (TilePlan::ltx: the coefficients of the <= 32 output rows of both planes over the <= 128
// union columns, pre-cut into two fp16 slices at bake time) and the planes are a small matrix product per unit:
//
//   stage 1   E^T[(sample, feature), (plane, row)] = Xu^T[(sample, feature), u] * Lt^T[u, (plane, row)]
//             M = 128, N = 64, K = 128: 8 k-steps, each MFMA wave owns (two samples) x (one plane), three slice products
//             per step -> 48 MFMAs per wave and unit.  A operand: the union rows of the 4 samples, cut into fp16
//             slices and TRANSPOSED by the producer waves on the way into LDS (a lane owns 4 union rows x 4 features:
//             the 4 rows of one feature are one 8-byte store, as in k_gemm_tn_ws); B operand: the wave's 32 rows of the
//             dense block, copied to LDS once per block.
//   convert   the accumulators (fp32: E 2^(sx + lt_exp)) hold, per lane, 4 consecutive features of one (plane, row):
//             rescale, cut into fp16 slices, 8-byte stores into the A image of stage 2 - no cross-lane traffic.
//   stage 2   C = [A0 | E1 | E2] W: exactly the k loop of k_cheb_tile_gemm (72 MFMAs per wave and unit).
// The producers only move data: union rows and plane 0 from global (prefetched one unit ahead), split, store.  Two
// LDS-only block barriers per unit.  LDS: 69 632 (transposed union image) + 53 248 (A image) + 32 768 (dense block)
// = 152 KB.  120 MFMAs per wave and unit instead of 72, ~150 VALU instructions per lane instead of ~550.
// E1 / E2 differ from the fmaf chain of k_basis_tile by fp32 round-off (22-bit operands, fp32 accumulation).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MG_LDU = TILE_UPAD + 8;                  // 16-bit words per row of the transposed union image: 272 B = 17 x 16 B
constexpr int MG_LDK = 3 * CT_CF + 8;                  // 16-bit words per row of the A image: 208 B = 13 x 16 B
// per slice arithmetic (NS slices) and SG samples per unit
constexpr int mg_xu_slice(int sg) { return sg * CT_CF * MG_LDU; }     // rows (sample, feature as e * 8 + q for feature 4 q + e)
constexpr int mg_a_slice(int sg) { return sg * 32 * MG_LDK; }         // rows (sample, tile row)
constexpr int mg_lt_elems(int ns) { return (TILE_UPAD / 16) * ns * 64 * 16; }
constexpr int mg_lds_bytes(int ns, int sg) { return (ns * mg_xu_slice(sg) + ns * mg_a_slice(sg) + mg_lt_elems(ns)) * 2 + 256; }
static_assert(TILE_UCAP <= TILE_UPAD && TILE_UPAD == 128, "stage 1 walks 8 k-steps of 16 union rows");
static_assert(mg_lt_elems(2) == TILE_LTX_ELEMS && mg_lt_elems(3) == TILE_LTX3_ELEMS, "dense operator blocks of the plan");
static_assert(mg_lds_bytes(2, 4) <= 160 * 1024 && mg_lds_bytes(3, 2) <= 160 * 1024, "LDS budget of one CU");

// The matrix-core gather in BOTH slice arithmetics (round 5).  NS = 2 (two scaled fp16 slices): a unit is 4 samples x 32
// features, as described above.  NS = 3 (three exact bf16 slices - the arithmetic of the bench's headline): every image
// grows by half, so a unit is SG = 2 samples x 32 features (52 + 40 + 48 KB = 140 KB) - the same 120 MFMAs per wave and unit
// (stage 1: one sample x one plane per wave, 8 k-steps x 6 slice products; stage 2: 6 k-steps x 6 products x TN tiles),
// i.e. twice the units per row.  The dense operator block exists as a second image of three exact bf16 slices
// (TilePlan::ltx3, unscaled: bf16 has fp32's exponent range), X / plane 0 are cut with split3_pack4, nothing is scaled.
// TMS = samples per MFMA wave (SG = 2 TMS).
template <int TMS, int TN, int NPW, int MODE, int NS>
__global__ __launch_bounds__(256 + 64 * NPW, (4 + NPW) / 4) void k_cheb_mg_gemm(TileGemmArgs g) {
  typedef typename SliceFrag<NS>::type frag_t;
  constexpr int TM = TMS;
  constexpr int SG = 2 * TMS;                 // samples per unit
  constexpr int SB = SG == 4 ? 2 : 1;         // log2(SG)
  constexpr int XU_SLICE = mg_xu_slice(SG), A_SLICE = mg_a_slice(SG);
  constexpr int LT_BYTES = mg_lt_elems(NS) * 2;
  constexpr int NP = 64 * NPW;
  static_assert(NPW == 4 && (SG == 4 || SG == 2), "256 producer lanes; 4 or 2 samples per unit");
  constexpr int NUH = NP >> (4 + SB);         // u-quad pairs walked in parallel by the producer lanes
  constexpr int NXI = 16 / NUH;               // union items (4 rows x 4 features) per producer lane
  constexpr int NPR = NP >> (3 + SB);         // tile rows walked in parallel (plane 0): 8 (SG = 4) or 16 (SG = 2)
  constexpr int NPI = 32 / NPR;               // plane-0 items (1 row x 4 features) per producer lane
  extern __shared__ __attribute__((aligned(16))) unsigned char ct_smem[];
  unsigned short* Xu = reinterpret_cast<unsigned short*>(ct_smem);
  unsigned short* Ai = Xu + NS * XU_SLICE;
  unsigned short* Lt = Ai + NS * A_SLICE;
  int* rowvid = reinterpret_cast<int*>(ct_smem + (NS * XU_SLICE + NS * A_SLICE) * 2 + LT_BYTES);

  const TilePlan& pl = g.pl;
  const int ngroups = (g.B + SG - 1) / SG;
  const int nbg = (ngroups + g.gpb - 1) / g.gpb;
  const int lid = xcd_contiguous(blockIdx.x, gridDim.x);
  if (lid >= pl.ntiles * nbg) return;
  const int tile = lid % pl.ntiles;
  const int bg = lid / pl.ntiles;
  const int grp0 = bg * g.gpb;
  int grp1 = grp0 + g.gpb;
  if (grp1 > ngroups) grp1 = ngroups;
  const int nchunks = g.Ka / CT_CF;
  const int nunits = (grp1 - grp0) * nchunks;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int r0 = pl.tile_row[tile], R = pl.tile_row[tile + 1] - r0;
  const int u0 = pl.tile_u[tile], U = pl.tile_u[tile + 1] - u0;
  if (t < 32) rowvid[t] = t < R ? g.row_ids[r0 + t] : -1;
  {                                           // the tile's dense block: global -> LDS, once per block
    const unsigned short* img = NS == 2 ? pl.ltx + (size_t)tile * TILE_LTX_ELEMS : pl.ltx3 + (size_t)tile * TILE_LTX3_ELEMS;
    const u32x4* src = reinterpret_cast<const u32x4*>(img);
    u32x4* dst = reinterpret_cast<u32x4*>(Lt);
    for (int k = t; k < LT_BYTES / 16; k += 256 + 64 * NPW) dst[k] = src[k];
  }
  __syncthreads();
  int sx = 0, descale = 0, lt_exp = 0;        // NS = 3: nothing is scaled
  float x_sc = 1.f;
  if constexpr (NS == 2) {
    sx = slice_scale_exp(*g.x_amax, g.x_bits);
    x_sc = exp2_int(sx);
    descale = -(sx + slice_scale_exp(*g.b_amax, 0));
    lt_exp = pl.lt_exp;
  }

  if (t >= 256) {
    // ------------------------------------------------------------------------------------------------ producers
    const int pt = t - 256;
    const int q = pt & 7;                               // this lane's 4 features (16 bytes) of a 128-byte line
    // union image items: (u-quad, sample, q).  The 16 lanes of a store group are 8 q x 2 u-quads: rows e * 8 + q are
    // 4 dwords apart mod 32, the two u-quads 2 dwords -> every 8-byte store of a group has its own bank pair
    const int ulo = (pt >> 3) & 1, s = (pt >> 4) & (SG - 1), uhi = pt >> (4 + SB);
    unsigned uoff[NXI][4];
#pragma unroll
    for (int k = 0; k < NXI; k++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int u = ((((uhi + k * NUH) << 1) | ulo) << 2) + j;
        uoff[k][j] = (unsigned)pl.ucol[u0 + (u < U ? u : U - 1)] * (unsigned)(g.Ka * 4);   // clamped: their Lt columns are 0
      }
    // plane-0 items: (tile row, sample, q).  Round 5 (profiles/r05_f_lds_conflict_ablation.txt): the 16 lanes of a store
    // group were 8 q x 2 SAMPLES - 32 rows = 1 664 dwords = 0 mod 32 banks apart: a 2-way conflict on every store (8.0 M of
    // the kernel's 24.1 M SQ_LDS_BANK_CONFLICT cycles).  Now 8 q x 2 rows FOUR apart (208 dwords = 16 mod 32): the 8-byte
    // stores of a group tile the 32 banks.
    const int r4 = (pt >> 3) & 1, s0 = (pt >> 4) & (SG - 1), rh = pt >> (4 + SB);
    auto p0_row = [&](int k) { return (rh & 3) + 4 * r4 + 8 * ((rh >> 2) + (NPR / 8) * k); };
    unsigned a0off[NPI];
#pragma unroll
    for (int k = 0; k < NPI; k++) {
      const int vid = rowvid[p0_row(k)];
      a0off[k] = (unsigned)((vid < 0 ? 0 : vid) >> g.a0_shift) * (unsigned)(g.Ka * 4);
    }
    f32x4 xr[NXI][4], p0[NPI];
    auto sample_of = [&](int grp, int ss) {             // clamped: a group's missing samples recompute the last one
      const int b = grp * SG + ss;
      return b < g.B ? b : g.B - 1;
    };
    int lg = grp0, lfc = 0;                             // next unit of the loaders
    auto load_unit = [&]() {
      const char* bx = reinterpret_cast<const char*>(g.X + ((long)sample_of(lg, s) * g.x_rows) * g.Ka + lfc * CT_CF + q * 4);
#pragma unroll
      for (int k = 0; k < NXI; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) xr[k][j] = *reinterpret_cast<const f32x4*>(bx + uoff[k][j]);
      const char* ba = reinterpret_cast<const char*>(g.A0 + ((long)sample_of(lg, s0) * g.a0_rows) * g.Ka + lfc * CT_CF + q * 4);
#pragma unroll
      for (int k = 0; k < NPI; k++) p0[k] = *reinterpret_cast<const f32x4*>(ba + a0off[k]);
      if (++lfc == nchunks) { lfc = 0; lg++; }
    };
    // activation on load (in_scale != nullptr): applied to the registers of a unit when they are stored, a unit after their
    // loads were issued; the unit's feature chunk is tracked per store kind (both walk units 0, 1, 2, ...)
    int sfc_x = 0, sfc_p = 0;
    auto act_chunk = [&](int fc, f32x4& sc, f32x4& sh) {
      sc = *reinterpret_cast<const f32x4*>(g.in_scale + fc * CT_CF + q * 4);
      sh = *reinterpret_cast<const f32x4*>(g.in_shift + fc * CT_CF + q * 4);
    };
    auto act4 = [](f32x4& v, const f32x4& sc, const f32x4& sh) {
#pragma unroll
      for (int c = 0; c < 4; c++) v[c] = fmaxf(fmaf(v[c], sc[c], sh[c]), 0.f);
    };
    auto store_xu = [&]() {
      if (g.in_scale != nullptr) {
        f32x4 sc, sh;
        act_chunk(sfc_x, sc, sh);
#pragma unroll
        for (int k = 0; k < NXI; k++)
#pragma unroll
          for (int j = 0; j < 4; j++) act4(xr[k][j], sc, sh);
        if (++sfc_x == nchunks) sfc_x = 0;
      }
#pragma unroll
      for (int k = 0; k < NXI; k++) {
        const int uq = ((uhi + k * NUH) << 1) | ulo;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          u32x2 sl[NS];
          split_pack4<NS>(xr[k][0][e], xr[k][1][e], xr[k][2][e], xr[k][3][e], x_sc, sl);
          unsigned short* d = Xu + (s * 32 + e * 8 + q) * MG_LDU + uq * 4;
#ifndef P2M_ABL_NO_XU        // (profiling ablations, tools/lds_conflict_ablation.sh: which LDS access class conflicts)
#pragma unroll
          for (int z = 0; z < NS; z++) *reinterpret_cast<u32x2*>(d + z * XU_SLICE) = sl[z];
#else
          asm volatile("" :: "v"(sl[0]), "v"(sl[NS - 1]), "v"(d));
#endif
        }
      }
    };
    auto store_p0 = [&]() {
      if (g.in_scale != nullptr) {
        f32x4 sc, sh;
        act_chunk(sfc_p, sc, sh);
#pragma unroll
        for (int k = 0; k < NPI; k++) act4(p0[k], sc, sh);
        if (++sfc_p == nchunks) sfc_p = 0;
      }
#pragma unroll
      for (int k = 0; k < NPI; k++) {
        u32x2 sl[NS];
        split_pack4<NS>(p0[k][0], p0[k][1], p0[k][2], p0[k][3], x_sc, sl);
        unsigned short* d = Ai + (s0 * 32 + p0_row(k)) * MG_LDK + q * 4;
#ifndef P2M_ABL_NO_P0
#pragma unroll
        for (int z = 0; z < NS; z++) *reinterpret_cast<u32x2*>(d + z * A_SLICE) = sl[z];
#else
        asm volatile("" :: "v"(sl[0]), "v"(sl[NS - 1]), "v"(d));
#endif
      }
    };
    load_unit();
    store_p0();
    store_xu();
    if (nunits > 1) load_unit();
    lds_block_barrier();                                // X2(-1): unit 0 is staged
    for (int w = 0; w < nunits; w++) {
      lds_block_barrier();                              // X1(w): the MFMA waves are done with the union image of unit w
      if (w + 1 < nunits) store_xu();                   // ... under their stage 2
      lds_block_barrier();                              // X2(w): ... and with the A image of unit w
      if (w + 1 < nunits) {
        store_p0();                                     // plane 0 of unit w + 1, under its stage 1 (visible at X1(w + 1))
        if (w + 2 < nunits) load_unit();
      }
    }
  } else {
    // ------------------------------------------------------------------------------------------------ MFMA waves
    // 2 x 2 over (sample half, plane) in stage 1 and over (sample half, column half) in stage 2
    const int wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // stage 2: the pre-split weight, as in k_cheb_tile_gemm
    const long bx_slice = (long)g.Npad * 32;
    const long bx_plane = (long)(g.Ka / 16) * NS * bx_slice;
    const char* bx_lane = reinterpret_cast<const char*>(g.Bx) + ((wn * TN * 32 + l31) * 16 + lhi * 8) * 2;
    // ring of NB fragment sets: the set of step st is refilled, for step st + NB (of this unit or the next), as soon as
    // step st has used it.  Two fp16 slices - N = 64: one set per k-step, a whole unit of lead; N = 128: three sets (six
    // would spill).  Three bf16 slices (a set is half as large again, the accumulators half as many): three sets
    constexpr int NB = NS == 3 ? 3 : (TN == 1 ? 6 : 3);
    frag_t fb[NB][NS][TN];
    auto load_b = [&](int fc, int st, frag_t (&b)[NS][TN]) {
      const char* src = bx_lane + (st >> 1) * bx_plane + (long)(fc * 2 + (st & 1)) * NS * bx_slice;
#pragma unroll
      for (int sl = 0; sl < NS; sl++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          b[sl][j] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(src + sl * bx_slice + j * (32 * 16 * 2)));
    };
    const unsigned short* a_lane = Ai + ((wm * TM) * 32 + l31) * MG_LDK + lhi * 8;
    auto read_a = [&](int sl, int koff, frag_t (&a)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; i++)
        a[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(a_lane + sl * A_SLICE + i * 32 * MG_LDK + koff));
    };
    // stage 1: this wave forms plane wn + 1 of its TM samples.  A fragments = rows (sample, feature) of the union
    // image; B fragments = rows (plane wn, tile row) of the dense block (LDS, 16-byte units [k-step][slice][half][row])
    const unsigned short* xu_lane = Xu + ((wm * TM) * 32 + l31) * MG_LDU + lhi * 8;
    const unsigned short* lt_lane = Lt + (lhi * 64 + wn * 32 + l31) * 8;
    const float e_sc = exp2_int(-lt_exp);                // stage-1 accumulator = E 2^(sx + lt_exp)
    unsigned short* c_lane = Ai + ((wm * TM) * 32 + l31) * MG_LDK + (wn + 1) * CT_CF + 16 * lhi;
    float* Eout = wn == 0 ? g.E1 : g.E2;
#pragma unroll
    for (int st = 0; st < NB; st++) load_b(0, st, fb[st]);
    lds_block_barrier();                                // X2(-1)
    int grp = grp0, fc = 0;
    for (int w = 0; w < nunits; w++) {
      const int fcn = fc + 1 == nchunks ? 0 : fc + 1;
      // ---- stage 1.  Fragments of step ks + 1 are read in front of the MFMAs of step ks: issued behind them, their
      // latency would be a bubble of the matrix pipe at every step
      floatx16 e[TM];                                   // (the first product of a unit takes the literal 0 as its addend)
      frag_t xa[2][TM][NS], lt[2][NS];                  // [ring][sample][slice], [ring][slice]
#if defined(P2M_ABL_LINEAR_X)      // x fragments from a LINEAR (lane * 16 bytes) pattern: wrong values, conflict-free by construction
      const unsigned short* xu_lane = Xu + lane * 8;
#endif
#if defined(P2M_ABL_LINEAR_LT)
      const unsigned short* lt_lane = Lt + lane * 8;
#endif
      auto read_x = [&](int ks, frag_t (&x)[TM][NS], frag_t (&l)[NS]) {
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
#pragma unroll
          for (int i = 0; i < TM; i++)
            x[i][sl] = __builtin_bit_cast(
                frag_t, *reinterpret_cast<const u32x4*>(xu_lane + i * 32 * MG_LDU + sl * XU_SLICE + ks * 16));
          l[sl] = __builtin_bit_cast(frag_t, *reinterpret_cast<const u32x4*>(lt_lane + (ks * NS + sl) * 128 * 8));
        }
      };
      read_x(0, xa[0], lt[0]);
#pragma unroll
      for (int ks = 0; ks < TILE_UPAD / 16; ks++) {
        if (ks + 1 < TILE_UPAD / 16) read_x(ks + 1, xa[(ks + 1) & 1], lt[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const frag_t (&x)[TM][NS] = xa[ks & 1];
        const frag_t (&l)[NS] = lt[ks & 1];
        // slice products, smallest first (slice 0 = high)
#define P2M_S1(SX, SL, FIRST)                                                     \
  _Pragma("unroll") for (int i = 0; i < TM; i++)                                  \
      e[i] = slice_mfma<NS>(x[i][SX], l[SL], (FIRST) && ks == 0 ? floatx16{} : e[i]);
        if constexpr (NS == 3) {
          P2M_S1(2, 0, true)
          P2M_S1(0, 2, false)
          P2M_S1(1, 1, false)
          P2M_S1(1, 0, false)
          P2M_S1(0, 1, false)
          P2M_S1(0, 0, false)
        } else {
          P2M_S1(1, 0, true)
          P2M_S1(0, 1, false)
          P2M_S1(0, 0, false)
        }
#undef P2M_S1
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- the plane leaves the accumulators: lane = tile row l31, registers j + 4 q4 = features 16 lhi + 4 j + q4
      // Round 5: two feature quads per 16-BYTE store.  As 8-byte stores the 16 lanes of a store group were 16 tile rows of one
      // column - row stride 52 dwords = 20 mod 32 banks has period 8: a 2-way conflict on every store, 16.1 M of the kernel's
      // 24.1 M SQ_LDS_BANK_CONFLICT cycles (ablation: profiles/r05_f_lds_conflict_ablation.txt).  A ds_write_b128 is serviced
      // in groups of 8 lanes = 8 rows x 4 banks: they tile the 32 banks, and the stores are half as many.
#pragma unroll
      for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int jp = 0; jp < 2; jp++) {
          u32x2 sa[NS], sb[NS];
          const int j = 2 * jp;
          if constexpr (NS == 2) {
            split2_pack4(e[i][j], e[i][j + 4], e[i][j + 8], e[i][j + 12], e_sc, sa[0], sa[1]);
            split2_pack4(e[i][j + 1], e[i][j + 5], e[i][j + 9], e[i][j + 13], e_sc, sb[0], sb[1]);
          } else {
            split3_pack4(e[i][j], e[i][j + 4], e[i][j + 8], e[i][j + 12], sa[0], sa[1], sa[2]);
            split3_pack4(e[i][j + 1], e[i][j + 5], e[i][j + 9], e[i][j + 13], sb[0], sb[1], sb[2]);
          }
          unsigned short* d = c_lane + i * 32 * MG_LDK + 8 * jp;
#ifndef P2M_ABL_NO_CONV
#pragma unroll
          for (int z = 0; z < NS; z++) *reinterpret_cast<u32x4*>(d + z * A_SLICE) = u32x4{sa[z][0], sa[z][1], sb[z][0], sb[z][1]};
#else
          asm volatile("" :: "v"(sa[0]), "v"(sb[NS - 1]), "v"(d));
#endif
        }
      }
      lds_block_barrier();                              // X1(w): planes 1, 2 of unit w visible; union image released
      // ---- stage 2
#if defined(P2M_ABL_LINEAR_A)
      const unsigned short* a_lane = Ai + lane * 8;
#endif
      frag_t fa[2][NS][TM];                             // A fragments, one step ahead
#pragma unroll
      for (int sl = 0; sl < NS; sl++) read_a(sl, 0, fa[0][sl]);
#pragma unroll
      for (int st = 0; st < 6; st++) {
        if (st + 1 < 6) {
#pragma unroll
          for (int sl = 0; sl < NS; sl++) read_a(sl, (st + 1) * 16, fa[(st + 1) & 1][sl]);
        }
        __builtin_amdgcn_sched_barrier(0);
#define P2M_PAIR(SA, SB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < TM; i++) _Pragma("unroll") for (int j = 0; j < TN; j++) \
      acc[i][j] = slice_mfma<NS>(fa[st & 1][SA][i], fb[st % NB][SB][j], acc[i][j]);
        if constexpr (NS == 3) {
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
        __builtin_amdgcn_sched_barrier(0);
        if (st + NB < 6) load_b(fc, st + NB, fb[st % NB]);       // step st + NB of this unit, or of the next one (the last
        else load_b(fcn, st + NB - 6, fb[st % NB]);              // unit refetches chunk 0: unused)
        __builtin_amdgcn_sched_barrier(0);
      }
      // the plane as fp32, for the weight gradient.  AFTER stage 2: vmcnt counts loads and stores in order, so a store in
      // front of stage 2 makes each of its waits for weight fragments a wait for the store's HBM acknowledgement as well
      if (Eout != nullptr) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
          const int bsm = grp * SG + wm * TM + i;
          if (l31 < R && bsm < g.B) {
            float* dst = Eout + ((long)bsm * g.nset + r0 + l31) * g.Ka + fc * CT_CF + 16 * lhi;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              f32x4 o;
#pragma unroll
              for (int q4 = 0; q4 < 4; q4++)
                o[q4] = NS == 2 ? __builtin_ldexpf(e[i][j + 4 * q4], -(sx + lt_exp)) : e[i][j + 4 * q4];
              *reinterpret_cast<f32x4*>(dst + 4 * j) = o;          // (cached: L2 merges the 16-byte pieces of a line)
            }
          }
        }
      }
      if (fc == nchunks - 1) {
        if (R == 32 && (grp + 1) * SG <= g.B && descale >= -120 && descale <= 120)
          tile_epilogue_full<TM, TN, MODE, SG>(g, pl, acc, rowvid, grp, tile, wm, wn, l31, lhi, descale);
        else
          tile_epilogue<TM, TN, MODE, SG>(g, pl, acc, rowvid, grp, tile, R, wm, wn, l31, lhi, descale);
        grp++;
      }
      fc = fcn;
      lds_block_barrier();                              // X2(w): stage 2 is done with the A image
    }
  }
}

}  // namespace p2m

using namespace p2m;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute of a kernel: under nn.DataParallel (one host
// thread per GPU, lib/core/base.py:108) every device needs its own call, and two threads may get here at once.  One of
// these per kernel instantiation: a mutex-guarded bitmap over the devices of the process.
struct DeviceOnce {
  std::mutex mu;
  uint64_t done[4] = {0, 0, 0, 0};       // 256 devices
  template <class F>
  int run(F&& set_attr, const char* what, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) {
      set_error("%s: cannot identify the current device", what);
      return P2M_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev >> 6] >> (dev & 63) & 1) return P2M_OK;
    const hipError_t e = set_attr();
    if (e != hipSuccess) {
      set_error("%s: cannot reserve %d bytes of LDS on device %d: %s", what, lds_bytes, dev, hipGetErrorString(e));
      return P2M_ERR_HIP;
    }
    done[dev >> 6] |= uint64_t(1) << (dev & 63);
    return P2M_OK;
  }
};

// gpb: sample groups a block walks (tables loaded once per block).  At least ~8 blocks per CU so that the last round of
// blocks is well filled (measured over the 16 real-row shapes of a train step: gpb 16 / 8 / 4 / 2 -> 21.4 / 19.0 / 18.4 /
// 18.0 ms; the finest level alone prefers 8 by 2 %).
static int pick_gpb(int ntiles, int ngroups, int gpb_max = 8) {
  int gpb = gpb_max;
  while (gpb > 1 && (long)ntiles * cdiv(ngroups, gpb) < 8 * 256) gpb >>= 1;
  return gpb;
}

template <int TM, int TN, int NPW, int MODE, int NS, bool BNR = false>
static int launch_tile_gemm_mode(const TileGemmArgs& a, hipStream_t s) {
  constexpr int LDS_BYTES = ct_lds_bytes(NS);
  static DeviceOnce attr_set;       // once per DEVICE and instantiation (never inside a stream capture: the first call
                                    // of every shape happens in the eager warm-up steps)
  if (const int rc = attr_set.run([] {
        return hipFuncSetAttribute((const void*)k_cheb_tile_gemm<TM, TN, NPW, MODE, NS, BNR>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      }, "p2m_cheb_tile_gemm", LDS_BYTES))
    return rc;
  const int ngroups = cdiv(a.B, CT_S);
  const int nblocks = cdiv((long)a.pl.ntiles * cdiv(ngroups, a.gpb), 8) * 8;
  hipLaunchKernelGGL((k_cheb_tile_gemm<TM, TN, NPW, MODE, NS, BNR>), dim3(nblocks), dim3(256 + 64 * NPW), LDS_BYTES, s, a);
  return check_launch("cheb_tile_gemm");
}

template <int TM, int TN, int NPW, int NS>
static int launch_tile_gemm_ns(const TileGemmArgs& a, hipStream_t s) {
  if constexpr (NS == 3 && TN == 1) {       // the fused BatchNorm-backward reduction: backward launches (no stats, no activation)
    if (a.bnr_y != nullptr) {
      if (a.stats != nullptr || a.act_scale != nullptr || a.act_relu) {
        set_error("p2m_cheb_tile_gemm: bnr_* excludes stats and the fused activation");
        return P2M_ERR_INVALID;
      }
      if (a.addend != nullptr) return launch_tile_gemm_mode<TM, TN, NPW, CT_ADDEND, NS, true>(a, s);
      return launch_tile_gemm_mode<TM, TN, NPW, CT_PLAIN, NS, true>(a, s);
    }
  }
  if (a.stats != nullptr) return launch_tile_gemm_mode<TM, TN, NPW, CT_STATS, NS>(a, s);
  if (a.addend != nullptr) return launch_tile_gemm_mode<TM, TN, NPW, CT_ADDEND, NS>(a, s);
  if (a.act_scale != nullptr || a.act_relu) return launch_tile_gemm_mode<TM, TN, NPW, CT_ACT, NS>(a, s);
  return launch_tile_gemm_mode<TM, TN, NPW, CT_PLAIN, NS>(a, s);
}
// k_cheb_tile_gemm_v2 (N <= 128; round 6).  P2M_TILE_V2=0 (read once per process) keeps the round-5 kernel for A/B runs.
static bool tile_v2() {
  static const bool v = [] { const char* e = getenv("P2M_TILE_V2"); return e ? atoi(e) != 0 : false; }();
  return v;
}
template <int TM, int TN, int MODE, int NS>
static int launch_tile_gemm_v2_mode(const TileGemmArgs& a, hipStream_t s) {
  constexpr int LDS_BYTES = ct2_lds_bytes(NS);
  static DeviceOnce attr_set;
  if (const int rc = attr_set.run([] {
        return hipFuncSetAttribute((const void*)k_cheb_tile_gemm_v2<TM, TN, MODE, NS>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      }, "p2m_cheb_tile_gemm(v2)", LDS_BYTES))
    return rc;
  const int ngroups = cdiv(a.B, CT_S);
  const int nblocks = cdiv((long)a.pl.ntiles * cdiv(ngroups, a.gpb), 8) * 8;
  hipLaunchKernelGGL((k_cheb_tile_gemm_v2<TM, TN, MODE, NS>), dim3(nblocks), dim3(768), LDS_BYTES, s, a);
  return check_launch("cheb_tile_gemm(v2)");
}
template <int TM, int TN, int NS>
static int launch_tile_gemm_v2(const TileGemmArgs& a, hipStream_t s) {
  if (a.stats != nullptr) return launch_tile_gemm_v2_mode<TM, TN, CT_STATS, NS>(a, s);
  if (a.addend != nullptr) return launch_tile_gemm_v2_mode<TM, TN, CT_ADDEND, NS>(a, s);
  if (a.act_scale != nullptr || a.act_relu) return launch_tile_gemm_v2_mode<TM, TN, CT_ACT, NS>(a, s);
  return launch_tile_gemm_v2_mode<TM, TN, CT_PLAIN, NS>(a, s);
}

template <int TMS, int TN, int NPW, int MODE, int NS>
static int launch_mg_gemm_mode(const TileGemmArgs& a, hipStream_t s) {
  constexpr int SG = 2 * TMS;
  constexpr int LDS_BYTES = mg_lds_bytes(NS, SG);
  static DeviceOnce attr_set;
  if (const int rc = attr_set.run([] {
        return hipFuncSetAttribute((const void*)k_cheb_mg_gemm<TMS, TN, NPW, MODE, NS>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      }, "p2m_cheb_tile_gemm(matrix gather)", LDS_BYTES))
    return rc;
  TileGemmArgs b = a;
  const int ngroups = cdiv(a.B, SG);
  // (2-sample units: a block still walks up to 16 samples' worth of groups, so that the 48 KB operator block and the
  //  tables are loaded as rarely per row as with 4-sample units)
  b.gpb = pick_gpb(a.pl.ntiles, ngroups, SG == 2 ? 16 : 8);
  const int nblocks = cdiv((long)a.pl.ntiles * cdiv(ngroups, b.gpb), 8) * 8;
  hipLaunchKernelGGL((k_cheb_mg_gemm<TMS, TN, NPW, MODE, NS>), dim3(nblocks), dim3(256 + 64 * NPW), LDS_BYTES, s, b);
  return check_launch("cheb_tile_gemm(matrix gather)");
}
template <int TMS, int TN, int NPW, int NS>
static int launch_mg_gemm(const TileGemmArgs& a, hipStream_t s) {
  if (a.stats != nullptr) return launch_mg_gemm_mode<TMS, TN, NPW, CT_STATS, NS>(a, s);
  if (a.addend != nullptr) return launch_mg_gemm_mode<TMS, TN, NPW, CT_ADDEND, NS>(a, s);
  if (a.act_scale != nullptr || a.act_relu) return launch_mg_gemm_mode<TMS, TN, NPW, CT_ACT, NS>(a, s);
  return launch_mg_gemm_mode<TMS, TN, NPW, CT_PLAIN, NS>(a, s);
}

// P2M_MG_EXACT (read once per process): 1 = three-bf16-slice launches with N <= 128 take the
// matrix-core gather like the two-fp16-slice ones; 0 (default) = the VALU-gather kernel.  Measured over the real-row shapes of a
// train step at B = 256 (profiles/r05_d_probe_mg{0,1}.txt): forward 8.63 vs 8.51 ms, with the planes out 5.29 vs 5.21 ms - the
// exact matrix-core gather (240 MFMAs per 128 rows x 32 features in 2-sample units) only ties the VALU gather that six slice
// products per step already hide, so the kernel with bitwise planes stays the default.
static bool mg_exact() {
  static const bool v = [] { const char* e = getenv("P2M_MG_EXACT"); return e ? atoi(e) != 0 : false; }();
  return v;
}

template <int TM, int TN, int NPW>
static int launch_tile_gemm(const TileGemmArgs& a, hipStream_t s) {
  // N <= 128 takes the gather on the matrix cores (4 producer waves: they only move data, and the MFMA waves need the
  // 256-register budget for the dense-block rows and the accumulator sets); N = 256 (a 128-register accumulator) stays
  // with the VALU gather.  Two fp16 slices: 4 samples per unit; three bf16 slices: 2 samples per unit (LDS).
  if (a.x_amax == nullptr) {
    if constexpr (TN == 1) {
      if (mg_exact() && a.pl.ltx3 != nullptr)
        return a.N == 128 ? launch_mg_gemm<1, 2, 4, 3>(a, s) : launch_mg_gemm<1, 1, 4, 3>(a, s);
      if (tile_v2()) return launch_tile_gemm_v2<TM, TN, 3>(a, s);
    }
    return launch_tile_gemm_ns<TM, TN, NPW, 3>(a, s);
  }
  if constexpr (TN == 1) return a.N == 128 ? launch_mg_gemm<2, 2, 4, 2>(a, s) : launch_mg_gemm<2, 1, 4, 2>(a, s);
  else return launch_tile_gemm_ns<TM, TN, NPW, 2>(a, s);
}

extern "C" int32_t p2m_cheb_tile_gemm_mg(int32_t arith, int32_t N) {
  /* 1 when a p2m_cheb_tile_gemm launch of this arithmetic / width takes the matrix-core gather (planes out supported at
   * full speed, E1 / E2 to fp32 round-off instead of bitwise the basis kernel's) */
  if (N > 128) return 0;
  return arith == P2M_ARITH_F16X2 || (arith == P2M_ARITH_BF16X3 && mg_exact()) ? 1 : 0;
}

extern "C" int32_t p2m_cheb_tile_gemm_supported(p2m_graph_t gh, int32_t plan, int32_t Ka, int32_t N) {
  if (!gh || plan < 0 || plan > 2) return 0;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (g.plan[plan].ntiles <= 0) return 0;
  if (Ka < 32 || Ka % 32 != 0) return 0;
  return (N == 64 || N == 128 || N == 256) ? 1 : 0;
}

// Partial-sum slots a launch with the fused BatchNorm-backward reduction writes (bnr_part: [slots][2][N]), or 0 when this
// (arithmetic, width) does not take the kernel that has it (k_cheb_tile_gemm with the LDS-staged epilogue: three bf16
// slices, N <= 128, the VALU gather)
extern "C" int32_t p2m_cheb_tile_gemm_bnr_slots(p2m_graph_t gh, int32_t plan, int32_t arith, int32_t N, int32_t B) {
  if (!gh || plan < 0 || plan > 2 || B <= 0) return 0;
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (g.plan[plan].ntiles <= 0 || arith != P2M_ARITH_BF16X3 || (N != 64 && N != 128) || mg_exact() || tile_v2()) return 0;
  const int ngroups = cdiv(B, CT_S);
  return g.plan[plan].ntiles * cdiv(ngroups, pick_gpb(g.plan[plan].ntiles, ngroups));
}

extern "C" int p2m_cheb_tile_gemm(p2m_graph_t gh, int32_t plan, const float* X, const float* A0, int32_t Ka,
                                  const void* Bx, int32_t arith, const void* x_amax, const float* bias,
                                  const float* addend, float* C, int32_t N, float* stats, float* E1, float* E2,
                                  const float* act_scale, const float* act_shift, int32_t act_relu, void* amax_out,
                                  const float* in_scale, const float* in_shift, const float* bnr_y, const float* bnr_co,
                                  float* bnr_part, int32_t B, void* stream) {
  P2M_CHECK_ARG(gh && X && A0 && Bx && C, "null pointer");
  P2M_CHECK_ARG((bnr_y == nullptr) == (bnr_co == nullptr) && (bnr_y == nullptr) == (bnr_part == nullptr),
                "bnr_y / bnr_co / bnr_part must all be given or all NULL");
  P2M_CHECK_ARG(bnr_y == nullptr || p2m_cheb_tile_gemm_bnr_slots(gh, plan, arith, N, B) > 0,
                "the fused BatchNorm-backward reduction is not available for this launch (p2m_cheb_tile_gemm_bnr_slots)");
  P2M_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr), "in_scale / in_shift must both be given or both NULL");
  P2M_CHECK_ARG(in_scale == nullptr || E1 == nullptr, "activation on load excludes the planes out");
  P2M_CHECK_ARG(arith == P2M_ARITH_BF16X3 || arith == P2M_ARITH_F16X2, "arith must be P2M_ARITH_BF16X3 or P2M_ARITH_F16X2");
  P2M_CHECK_ARG(arith != P2M_ARITH_F16X2 || x_amax != nullptr, "P2M_ARITH_F16X2 needs the amax word of X / A0");
  P2M_CHECK_ARG(plan >= 0 && plan <= 2, "plan must be 0 (level), 1 (un-pooled input) or 2 (paired operator)");
  P2M_CHECK_ARG((E1 == nullptr) == (E2 == nullptr), "E1 / E2 must both be given or both NULL");
  P2M_CHECK_ARG((act_scale == nullptr) == (act_shift == nullptr), "act_scale / act_shift must both be given or both NULL");
  P2M_CHECK_ARG(!((act_scale || act_relu) && stats), "fused activation excludes stats");
  P2M_CHECK_ARG(!(addend && (stats || act_scale || act_relu)), "addend excludes stats and the fused activation");
  P2M_CHECK_ARG(!(act_relu && !act_scale), "act_relu needs act_scale / act_shift");
  const Graph& g = *reinterpret_cast<const Graph*>(gh);
  if (!p2m_cheb_tile_gemm_supported(gh, plan, Ka, N)) {
    set_error("p2m_cheb_tile_gemm: plan %d of this level / Ka = %d / N = %d is not supported (p2m_cheb_tile_gemm_supported)",
              plan, Ka, N);
    return P2M_ERR_INVALID;
  }
  if (B <= 0) return P2M_OK;
  TileGemmArgs a;
  a.pl = g.plan[plan];
  const bool paired = plan == 2;
  a.row_ids = paired ? g.pair_real_ids : g.real_ids;
  a.nset = paired ? g.n_pair_real : g.n_real;
  a.X = X;
  a.A0 = A0;
  a.Bx = reinterpret_cast<const unsigned short*>(Bx);
  a.bias = bias;
  a.addend = addend;
  a.act_scale = act_scale;
  a.act_shift = act_shift;
  a.act_relu = act_relu;
  a.in_scale = in_scale;
  a.in_shift = in_shift;
  a.C = C;
  a.stats = stats;
  a.E1 = E1;
  a.E2 = E2;
  a.x_rows = plan == 1 ? g.V / 2 : g.V;
  a.a0_rows = plan == 0 ? g.V : g.V / 2;
  a.c_rows = paired ? g.V / 2 : g.V;
  a.a0_shift = plan == 1 ? 1 : 0;
  a.a0_in_x = (A0 == X && !paired) ? 1 : 0;
  a.bnr_y = bnr_y;
  a.bnr_co = bnr_co;
  a.bnr_part = bnr_part;
  a.B = B;
  a.Ka = Ka;
  a.N = N;
  a.Npad = cdiv(N, 128) * 128;          // the layout p2m_weight_split writes
  a.x_amax = a.b_amax = nullptr;
  a.x_bits = 0;
  if (arith == P2M_ARITH_F16X2) {
    a.x_amax = static_cast<const unsigned*>(x_amax);
    a.x_bits = g.plane_bits + (paired ? 1 : 0);
    a.b_amax = reinterpret_cast<const unsigned*>(a.Bx + 2l * a.Npad * 3 * Ka);     // trails the slices (p2m_weight_split)
  }
  a.amax_out = static_cast<unsigned*>(amax_out);
  a.gpb = pick_gpb(a.pl.ntiles, cdiv(B, CT_S));
  hipStream_t s = (hipStream_t)stream;
  if (N == 256) return launch_tile_gemm<4, 2, 4>(a, s);
  return N == 128 ? launch_tile_gemm<4, 1, 8>(a, s) : launch_tile_gemm<2, 1, 8>(a, s);
}

#ifdef P2M_TILE_TRACE
extern "C" int p2m_tile_trace_dump_waves(unsigned long long* out /* [12][40][2] host */) {
  if (hipDeviceSynchronize() != hipSuccess) return P2M_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2m::g_tile_trw), sizeof(unsigned long long) * 12 * 40 * 2) == hipSuccess ? P2M_OK
                                                                                                                   : P2M_ERR_HIP;
}
extern "C" int p2m_tile_trace_dump(unsigned long long* out /* [2][40][8] host */) {
  if (hipDeviceSynchronize() != hipSuccess) return P2M_ERR_HIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(p2m::g_tile_trc), sizeof(unsigned long long) * 2 * 40 * 8) == hipSuccess ? P2M_OK
                                                                                                                   : P2M_ERR_HIP;
}
#endif
