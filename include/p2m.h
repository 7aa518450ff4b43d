/*
 * p2m.h -- C ABI of libp2m_hip.so: the MI355X (gfx950) implementation of the Pose2Mesh
 * coarse-to-fine Chebyshev graph-convolution hot path.
 *
 * The reference (hongsukchoi/Pose2Mesh_RELEASE) is pure PyTorch and has no FFI layer; the
 * "kernels" below replace the library ops that the reference dispatches to.  Every entry point
 * cites the reference lines whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain C symbols, device pointers + sizes + hipStream_t (passed as void*); the library never
 *    allocates caller-visible device memory except the immutable per-level graph handle (internally it keeps one small
 *    scratch buffer per (device, stream) for the two-stage BatchNorm finalize, allocated on first use);
 *  - all matrices fp32 row-major; a "row" is one (sample, vertex) pair, r = b*V + v;
 *  - return 0 on success, negative p2m_status on error; p2m_last_error_string() has the detail;
 *    nothing throws across the boundary;
 *  - thread-safe: no mutable global state besides the thread-local error string; the one kernel-variant knob of the
 *    library (P2M_BASIS_TILED) is read once from the environment and never changes;
 *  - `shift` arguments implement the reference's nearest x2 vertex un-pooling
 *    (lib/models/meshnet.py:71-78) *virtually*: a tensor stored at V/2 vertices is read as if it
 *    had V vertices through row index r>>1 (valid because V is even and r = b*V+v).
 */
#ifndef P2M_H_
#define P2M_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  P2M_OK = 0,
  P2M_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  P2M_ERR_HIP = -2,       /* HIP runtime error (launch, alloc, copy) */
  P2M_ERR_NOMEM = -3
} p2m_status;

/* Arithmetic of the dense contractions.  All compute the SAME fp32 contraction with fp32 accumulation:
 * F32     native f32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain;
 * BF16X3  every fp32 operand cut EXACTLY into three bf16 slices (8+8+8 significand bits), the six slice products of
 *         weight >= 2^-16 on v_mfma_f32_32x32x16_bf16; dropped terms <= 2^-23 |a b| (one fp32 rounding);
 * F16X2   every operand TENSOR multiplied by a power of two 2^s that brings an upper bound U of its magnitudes into
 *         [2^14, 2^15), then cut into two fp16 slices (11+11 significand bits, round to nearest); three slice products
 *         on v_mfma_f32_32x32x16_f16, the accumulator rescaled exactly (v_ldexp) in the epilogue.  An operand carries
 *         x (1 + d), |d| <= 2^-22, for |x| >= 2^-18 U and an absolute error <= 2^-40 U below; a product additionally
 *         drops <= 2^-22 |a b|.  Half the matrix-core work of BF16X3.  U comes from an AMAX WORD: a uint32 in device
 *         memory holding the bits of a non-negative float >= max |x| over the tensor, produced on the device by
 *         whoever wrote the tensor (the amax_out arguments, p2m_amax, p2m_amax_rows; NaN and infinity are left out,
 *         so a non-finite element poisons its own rows of a contraction, as in fp32, not the scale) - it only has to BOUND the
 *         magnitudes; `bits` arguments add binades of headroom for operands derived from the bounded tensor inside
 *         the same call (Chebyshev planes: p2m_graph_plane_bits).                                                  */
enum { P2M_ARITH_F32 = 0, P2M_ARITH_BF16X3 = 1, P2M_ARITH_F16X2 = 2 };

typedef struct p2m_graph* p2m_graph_t;

/* Thread-local description of the last error returned on this thread. */
const char* p2m_last_error_string(void);
/* Library version / build info (e.g. "p2m-hip 0.3 (gfx950; ...)"). */
const char* p2m_version(void);
/* Identity of the stream capture `stream` is recording into: *id_out = the runtime's capture id (hipStreamGetCaptureInfo),
 * 0 when the stream is not capturing.  Host-side only (no launch): the amax-word pool of the Python layer keys its chunks
 * on it, so that two captures taken back to back never share graph-private memory.  No reference counterpart (the
 * reference has no stream captures). */
int p2m_stream_capture_id(void* stream, unsigned long long* id_out);

/* ---- graph handle: one per coarsening level ------------------------------------------------
 * Replaces sparse_python_to_torch (lib/graph_utils.py:98-109) and the per-forward
 * `self.graph_L[i].cuda()` upload (lib/models/meshnet.py:81).  Input: host CSR of the rescaled
 * Laplacian L (fp32, V x V, symmetric).  The handle bakes, on the current device, the merged
 * CSR of L and L2 = 2*L*L - I (double accumulation on the host, rounded once to fp32) so the
 * K=3 Chebyshev recurrence T1 = L x, T2 = 2 L T1 - x (lib/models/backbones/cheby_graph_conv.py:25,28)
 * becomes ONE gather pass  T1 = L x, T2 = L2 x.  It also bakes the lists of real / isolated (padding) vertices
 * and, for levels that have both, the tile plans of the LDS-staged basis kernel (consecutive real rows grouped
 * by the union of their neighbourhoods: one plan per un-pool shift, one for the paired operator S L | S L2).
 * The handle is immutable afterwards, with one exception: p2m_graph_set_classes, to be called (once) before the
 * handle is used.                                                                                          */
int p2m_graph_create(const int32_t* row_ptr, const int32_t* col, const float* val,
                     int32_t V, int32_t nnz, p2m_graph_t* out);
int p2m_graph_destroy(p2m_graph_t g);
/* info[0]=V, info[1]=nnz(L), info[2]=nnz(merged L|L2), info[3]=max merged row length */
int p2m_graph_info(p2m_graph_t g, int32_t info[4]);

/* ---- Chebyshev basis (sparse stage, HBM-bound) ----------------------------------------------
 * cheby_graph_conv.py:16-34 without the permute/cat/permute shuffles.
 * X: (B, V>>in_shift, F)   T1, T2: (B, V, F).   T0 = X is never rewritten.
 * Algorithmic HBM bytes per call: 4*B*V*F*(1/(1<<in_shift) + 2).                               */
int p2m_cheb_basis_fwd(p2m_graph_t g, const float* X, float* T1, float* T2,
                       int32_t B, int32_t F, int32_t in_shift, void* stream);

/* Backward of the basis stage (autograd of cheby_graph_conv.py:25-28; L symmetric,
 * asserted at lib/coarsening.py:23):  dX = d0 + L d1 + L2 d2  (+ resid, same shape as d0).
 * d0,d1,d2,resid: (B, V, F).  dX: (B, V>>out_shift, F); with out_shift=1 the two children of a
 * coarse vertex are summed (backward of nn.Upsample nearest, meshnet.py:74).                  */
int p2m_cheb_basis_bwd(p2m_graph_t g, const float* d0, const float* d1, const float* d2,
                       const float* resid, float* dX,
                       int32_t B, int32_t F, int32_t out_shift, void* stream);

/* Narrow-output convolution by linearity (the final 64 -> 3 layer, cheby_graph_conv.py:25-37 re-associated):
 *   y = [x|Lx|L2x] W == P0 + L P1 + L2 P2 with P = x [W0|W1|W2] computed first by p2m_gemm_planes.
 * combine: Y[r, c] = bias[c] + P[r, c] + sum_j a_j P[col_j, nc+c] + b_j P[col_j, 2nc+c],  c < nc <= 4;
 *          P rows are ldp floats wide (ldp >= 3 nc), Y is [B*V, nc].
 * expand (its backward): E[r] = [ G[r] | (L G)[r] | (L2 G)[r] | 0.. ], G: [B*V, nc], E rows lde floats wide. */
int p2m_cheb_combine_small(p2m_graph_t g, const float* P, int32_t ldp, int32_t nc, const float* bias,
                           float* Y, int32_t B, void* stream);
int p2m_cheb_expand_small(p2m_graph_t g, const float* G, int32_t nc, float* E, int32_t lde, int32_t B,
                          void* stream);
/* Inference form of the combine: only the REAL vertices of the level are computed (padding vertices never reach a
 * caller: lib/core/base.py:201, demo/run.py:170 gather graph_perm_reverse[:nv]); P needs valid rows at real vertices
 * only.  out_index == NULL: Y is [B*V, nc], rows of padding vertices are left untouched.  out_index != NULL
 * ([V] int32, -1 = drop): vertex v goes to row out_index[v] of Y = [B, out_rows, nc], multiplied by scale -- the
 * perm-reverse gather (and the Tester's x1000) folded into the last conv's store.                                  */
int p2m_cheb_combine_small_real(p2m_graph_t g, const float* P, int32_t ldp, int32_t nc, const float* bias, float* Y,
                                int32_t B, const int32_t* out_index, int32_t out_rows, float scale, void* stream);

/* ---- weights ------------------------------------------------------------------------------
 * nn.Linear(Fin*K, Fout).weight is [Fout][fin*K + k] (cheby_graph_conv.py:32-37).  Packs it into
 *   Wt [k*Fin + fin][Fout]   (B operand of the forward contraction, K-major)
 *   W2 [Fout][k*Fin + fin]   (B operand of dZ = g W; may be NULL)
 *   W3 [k*Fout + fout][fin]  (B operand of the fused backward dX = [g|Lg|L2g] W3; may be NULL)
 * K=1 gives a plain transpose (used for fc, meshnet.py:36-37).                                */
int p2m_weight_pack(const float* W, float* Wt, float* W2, float* W3, int32_t Fout, int32_t Fin, int32_t K,
                    void* stream);
/* Sums `nchunks` partial gradients P[chunk][k*Fin+fin][Fout], Pdb[chunk][Fout] produced by
 * p2m_gemm_tn and writes dW in nn.Linear layout [Fout][fin*K+k] and db[Fout].
 * accumulate!=0 adds into dW/db instead of overwriting.  layout 1: P[chunk][fin][k*Fout+fout] (the
 * gradient taken as X^T [g|Lg|L2g]); layout 2 (K = 1): P[chunk][fout][fin] - a plain Linear's gradient taken as G^T X (the
 * operands of p2m_gemm_tn swapped), already in nn.Linear layout: coalesced on both sides (Pdb is then the wrong column sum:
 * pass NULL and reduce the bias gradient separately).  pdb_stride = row stride of Pdb (the N of the p2m_gemm_tn call).  */
int p2m_weight_grad_unpack(const float* P, const float* Pdb, int32_t nchunks, float* dW, float* db,
                           int32_t Fout, int32_t Fin, int32_t K, int32_t accumulate, int32_t layout,
                           int32_t pdb_stride, void* stream);

/* ---- dense contraction (fp32; on the BF16 or the f32 MFMA pipe, see P2M_ARITH_*) -------------
 * C[r, n] = sum_p sum_k A_p[r (>> a0_shift if p==0), k] * Bm[p*Ka + k, n]  + bias[n]
 * A_p: nplanesA row-major [M, Ka] matrices (the Chebyshev basis planes X|T1|T2, or one plane);
 * Bm: [nplanesA*Ka, N] row-major; C is split in nplanesC column planes of width Nc (N = nplanesC*Nc),
 * C_q: [M, Nc].  Replaces `cl(x)` (cheby_graph_conv.py:37), `self.fc` (meshnet.py:105) and the
 * autograd dZ = g W.  If stats != NULL it receives per-row-tile BatchNorm partials
 * stats[tile][0][n] = sum_r y, stats[tile][1][n] = sum_r (y - tile_mean)^2, tile = 128 rows
 * (cheby_graph_conv.py:39 batch statistics; fake vertices included).
 * Shapes with Ka%32!=0 or N%32!=0 take a scalar (VALU) path.
 * addend (optional, single output plane): [M, N] added to the result (the block residual's gradient);
 * pair_out (single output plane): C is [M/2, N] and receives the sum of the two children rows of every
 * coarse vertex -- backward of nn.Upsample (meshnet.py:74) in the epilogue.  The backward uses this entry
 * point in "forward form": dX = [g | L g | L2 g] W3 with the planes from p2m_cheb_basis_fwd(g).
 *
 * Arithmetic (P2M_ARITH_* above).  arith = P2M_ARITH_F32: native f32 MFMA (bitwise an fmaf chain), Bsplit ignored.
 * The slice arithmetics run the same fp32 contraction on the 16-bit matrix pipe and need Bsplit, made from the same Bm
 * by p2m_weight_split with the SAME arith; P2M_ARITH_F16X2 also needs a_amax (amax word bounding ALL the A planes after
 * a_bits binades of headroom).  amax_out (optional, MFMA path): atomic max of |value stored| into a zeroed amax word -
 * the bound for whoever consumes C next.
 *
 * Fused activation (optional; inference): act_scale/act_shift != NULL applies v = fmaf(acc + bias, act_scale[n],
 * act_shift[n]) and act_relu != 0 then v = max(v, 0) before the store -- eval-mode BatchNorm (p2m_bn_eval_coeffs) and
 * F.relu (cheby_graph_conv.py:39, meshnet.py:100) folded into the contraction with the SAME two roundings as the
 * separate p2m_bn_act_fwd pass (bitwise the unfused result).  Excludes stats and pair_out.                          */
int p2m_gemm_planes(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                    int32_t a0_shift, const float* Bm, const void* Bsplit, int32_t arith, const void* a_amax,
                    int32_t a_bits, const float* bias, const float* addend,
                    float* C0, float* C1, float* C2, int32_t nplanesC, int32_t Nc, int32_t pair_out,
                    int64_t M, float* stats, const float* act_scale, const float* act_shift, int32_t act_relu,
                    void* amax_out, void* stream);
/* Bx[k / 16][s][n][k % 16] (uint16 bit patterns; n < ceil(N/128)*128, zero padded; K % 16 == 0):
 *   P2M_ARITH_BF16X3  s < 3 bf16 slices of Bm[k][n], Bm = slice 0 + slice 1 + slice 2 exactly;
 *   P2M_ARITH_F16X2   s < 2 fp16 slices of Bm[k][n] 2^sb, followed by the amax word the scale came from, from which
 *                     every consumer re-derives sb.  amax_in = NULL: max |Bm|, computed here (one extra small launch);
 *                     else the word of a tensor that bounds Bm after amax_bits binades - e.g. the parameter Bm is a
 *                     permutation of (0 bits), or W0 + a W1 + b W2 (p2m_weight_eff: ceil(log2(1 + |a| + |b|)) bits) -
 *                     so that one amax pass per parameter serves all of its derived operands.
 * p2m_weight_split_elems = number of uint16 elements of Bx (0: unsupported shape / arith).                          */
int64_t p2m_weight_split_elems(int32_t K, int32_t N, int32_t arith);
int p2m_weight_split(const float* Bm, int32_t K, int32_t N, int32_t arith, const void* amax_in, int32_t amax_bits,
                     void* Bx, void* stream);
/* Every slice image of a network's conv weights in two launches (instead of p2m_weight_pack + p2m_weight_eff x 2 +
 * p2m_weight_split x 4 + p2m_amax per layer and optimizer step).  dev_desc: DEVICE array of n descriptors, one per conv
 * layer with weight W[Fout][Fin * 3 + k] (nn.Linear layout, cheby_graph_conv.py:37): the images (each sized by
 * p2m_weight_split_elems for its [K, N]; NULL = not wanted)
 *   Bx_f   [3 Fin, Fout]   forward, real rows            Bx_ef  [Fin, Fout]   forward, padding rows: W0 + a W1 + b W2
 *   Bx_b   [3 Fout, Fin]   backward dX, real rows        Bx_eb  [Fout, Fin]   backward dX, padding rows
 * and, for P2M_ARITH_F16X2, the parameter's amax word (written here; eff_bits = ceil(log2(1 + |a| + |b|)) is the headroom
 * of the two effective forms).  The buffers and the descriptor array can be allocated once and reused every step.      */
typedef struct p2m_conv_weights {
  const float* W;
  int32_t Fout, Fin;
  float fake_a, fake_b;
  int32_t eff_bits, reserved;
  void *Bx_f, *Bx_ef, *Bx_b, *Bx_eb;
  void* amax;
} p2m_conv_weights;
int p2m_conv_weights_prepare(const p2m_conv_weights* dev_desc, int32_t n, int32_t arith, void* stream);
/* amax words: atomic max of |x| into *word (uint32, device; the caller zeroes it before the first contribution).
 * p2m_amax: n contiguous floats.  p2m_amax_rows: rows of a [B, V, F] tensor of a level -
 * row_set 0 = every row that holds data (all rows; the live rows once classes are declared), 1..4 = that row set.   */
int p2m_amax(const float* x, int64_t n, void* word, void* stream);
int p2m_amax_rows(p2m_graph_t g, int32_t row_set, const float* x, int32_t B, int32_t F, void* word, void* stream);
/* rows per BatchNorm partial tile and the number of tiles for M rows */
int32_t p2m_stats_tile_rows(void);

/* Weight-gradient contraction (G = [G0|G1|G2], nplanesG column planes of width Gc, N = nplanesG*Gc):
 *   P[chunk][p*Ka + k][n] = sum_{r in chunk} A_p[r][k] * G[r][n],
 * Pdb[chunk][n] = sum_{r in chunk} G[r][n];  chunk c covers rows [c*chunk_rows, (c+1)*chunk_rows).
 * (autograd of cheby_graph_conv.py:37 / meshnet.py:105.)                                       */
int p2m_gemm_tn(const float* A0, const float* A1, const float* A2, int32_t nplanesA, int32_t Ka,
                int32_t a0_shift, const float* G0, const float* G1, const float* G2, int32_t nplanesG,
                int32_t Gc, int64_t M, int64_t chunk_rows, float* P, float* Pdb, int32_t arith,
                const void* a_amax, int32_t a_bits, const void* g_amax, int32_t g_bits, void* stream);

/* ---- BatchNorm1d over B*V rows + ReLU + residual (cheby_graph_conv.py:39, meshnet.py:100,108-115)
 * finalize: reduces the GEMM's partials to batch mean / biased var, writes
 *   mean[N], invstd[N], scale = gamma*invstd, shift = beta - mean*scale and updates
 *   running_mean/var (momentum, unbiased var) exactly like nn.BatchNorm1d in train().         */
int p2m_bn_finalize(const float* stats, int32_t ntiles, int64_t M, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* mean, float* invstd, float* scale, float* shift, int32_t N, int32_t tile_rows,
                    void* stream);
/* eval(): scale/shift/invstd from the running statistics. */
int p2m_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* mean, float* invstd, float* scale,
                       float* shift, int32_t N, void* stream);
/* x[r,f] = act(y[r,f]*scale[f] + shift[f]) + lerp_F(resid[r>>res_shift, 0..Fres))[f]
 * act = ReLU if relu!=0; scale/shift may be NULL (identity); resid may be NULL.
 * lerp_F is F.interpolate(mode='linear', align_corners=False) along the FEATURE axis
 * (meshnet.py:109,114).  Rows: all M; with `classes` (a level's handle) the live rows of a level with declared
 * classes (holes are skipped); with real_rows_only != 0 only the handle's real vertices (inference on the real rows:
 * the other rows of y hold no data and x's stay untouched).  amax_out (optional, F % 4 == 0): atomic max of |x stored|
 * into a zeroed amax word (P2M_ARITH_F16X2 above) - the bound for the contraction that consumes x.                  */
int p2m_bn_act_fwd(const float* y, const float* scale, const float* shift, int32_t relu,
                   const float* resid, int32_t Fres, int32_t res_shift,
                   float* x, int64_t M, int32_t F, p2m_graph_t classes /* or NULL */, int32_t real_rows_only,
                   void* amax_out, void* stream);
/* backward through ReLU + BatchNorm (train: batch statistics; eval: running statistics).
 *   go = gx * (y*scale+shift > 0 or !relu)
 *   reduce:   part[blk][0][f] = sum go, part[blk][1][f] = sum go * yhat       (nblk = p2m_bn_bwd_blocks(M,F))
 *   finalize: dgamma, dbeta (accumulate!=0 adds), coef[0][f]=dbeta/M, coef[1][f]=dgamma/M
 *   apply:    gy = gamma*invstd*(go - coef0 - yhat*coef1)   (training)   |   gamma*invstd*go (eval, coef NULL) */
int32_t p2m_bn_bwd_blocks(int64_t M, int32_t F);
int32_t p2m_bn_bwd_blocks_classes(p2m_graph_t classes, int64_t M, int32_t F);   /* blocks when `classes` is passed */
/* classes (optional graph handle with p2m_graph_set_classes; NULL = every row counts): the passes walk the live rows
 * of the level only (holes are neither read nor written; part then has p2m_bn_bwd_blocks_classes blocks), and the apply
 * pass adds the constant term of a representative once per class member.                                           */
int p2m_bn_bwd_reduce(const float* gx, const float* y, const float* scale, const float* shift,
                      const float* mean, const float* invstd, int32_t relu, float* part,
                      int64_t M, int32_t F, p2m_graph_t classes, void* stream);
int p2m_bn_bwd_finalize(const float* part, int32_t nblk, int64_t M, float* dgamma, float* dbeta,
                        float* coef, int32_t accumulate, int32_t F, void* stream);
/* The reduce pass over the FAKE-vertex rows of a level only (b * V + fake id; with classes: the representatives, which
 * carry their class's summed gradient) - the complement of a reduction whose real-vertex rows were summed by the kernel
 * that produced gx (p2m_cheb_tile_gemm, bnr_*).  part: [p2m_bn_bwd_blocks_fake(g, B, F)][2][F]; F in {32, 64, 128, 256}.  */
int32_t p2m_bn_bwd_blocks_fake(p2m_graph_t g, int32_t B, int32_t F);
int p2m_bn_bwd_reduce_fake(p2m_graph_t g, const float* gx, const float* y, const float* scale, const float* shift,
                           const float* mean, const float* invstd, int32_t relu, float* part, int32_t B, int32_t F,
                           void* stream);
/* pair_gx / pair_gy (optional, [M/2, F]; M even, F in {32,64,128,256}): by-products pair_gx[q] = gx[2q] + gx[2q+1] and
 * pair_gy[q] = gy[2q] + gy[2q+1] -- the pair-sums the backward of an un-pooled conv needs (residual gradient for the
 * coarser level; plane S g of the paired operator), produced while the rows are in registers anyway.
 * amax_out (optional): atomic max of |gy stored| into a zeroed amax word (the pair sums are bounded by twice it).
 * zero_holes (with classes): the pass walks EVERY row and stores zeros at the holes (and at pair sums of two holes) instead
 * of leaving them untouched - for levels whose other kernels read all rows; no memset of the outputs is needed.        */
int p2m_bn_bwd_apply(const float* gx, const float* y, const float* scale, const float* shift,
                     const float* mean, const float* invstd, const float* gamma, const float* coef,
                     int32_t relu, float* gy, float* pair_gx, float* pair_gy, int64_t M, int32_t F,
                     p2m_graph_t classes, int32_t zero_holes, void* amax_out, void* stream);

/* out[p, f] = in[2p, f] + in[2p+1, f]   (backward of the x2 nearest un-pool, meshnet.py:74) */
int p2m_pair_sum(const float* in, float* out, int64_t Mout, int32_t F, p2m_graph_t classes /* of `in`'s level, or NULL */,
                 void* stream);
/* dst[r, i] += sum_j w(j,i) g[r, j]: transpose of the feature-axis resize (meshnet.py:109,114);
 * g: [M, F], dst: [M, Fres].                                                                   */
int p2m_lerp_bwd_add(const float* g, float* dst, int64_t M, int32_t F, int32_t Fres, void* stream);

/* ---- fake-vertex split (row-set launches) ----------------------------------------------------------------
 * Padding ("fake") vertices of the coarsening tree are isolated (lib/coarsening.py:236-245): their merged
 * CSR row is the diagonal alone and identical for the whole level, so T1 = a x, T2 = b x and the contraction of
 * cheby_graph_conv.py:37 needs only K = Fin with W_eff = W0 + a W1 + b W2 -- 2/3 of the MFMA work on 30-40 % of the
 * rows disappears, and the basis planes are only formed (compactly, [B*n_real, F]) for real vertices.
 * Rows are selected by row_set: 1 = real vertices, 2 = fake vertices (sorted id lists baked in the handle; 3 / 4: the
 * paired sets below, over V/2 rows); tensors
 * keep their (B, V, F) layout, only the launches iterate over the subset.  BatchNorm statistics still cover ALL
 * rows (the reference includes fake vertices, cheby_graph_conv.py:39): p2m_bn_finalize_rows merges both launches. */
int p2m_graph_split_info(p2m_graph_t g, int32_t counts[2] /* n_real, n_fake */, float coef[2] /* a, b */);
int p2m_cheb_basis_fwd_real(p2m_graph_t g, const float* X, float* T1c, float* T2c, int32_t B, int32_t F,
                            int32_t in_shift, const float* act_scale /* or NULL */, const float* act_shift, void* stream);
/* ---- paired operator: the backward of an un-pooled conv at the COARSE resolution ------------------------------
 * A conv whose input was un-pooled x2 (meshnet.py:71-78,111) sees X_fine[r] = X_coarse[r >> 1], so with S = the
 * pair-sum (the un-pool's transpose) and L symmetric:
 *     dX_coarse = S ([g | L g | L2 g] W3) = [S g | S L g | S L2 g] W3,     dW = X_coarse^T [S g | S L g | S L2 g]
 * -- both contractions run over V/2 rows.  The handle of the FINE level bakes S L and S L2 as one more tile plan
 * (row c = merged row 2c + merged row 2c+1) and two more row sets over the coarse index space: row_set 3 = coarse
 * vertices with at least one real child (compact plane order), row_set 4 = both children fake (S L g = a S g,
 * S L2 g = b S g: effective weight, as for row_set 2).  counts = {0, 0}: no paired operator on this level.
 * p2m_cheb_basis_pair: P1c = S L g, P2c = S L2 g over row_set 3, compact [B*counts[0], F]; g: [B*V, F];
 * F = 32, 64 or a multiple of 128.  (S g itself is p2m_pair_sum.)                                                   */
int p2m_graph_pair_info(p2m_graph_t g, int32_t counts[2] /* n_pair_real, n_pair_fake */);
/* tiles of the LDS-staged basis kernel's plans: [in_shift 0, in_shift 1, paired]; 0 = no plan, the row kernel runs */
int p2m_graph_plan_info(p2m_graph_t g, int32_t ntiles[3]);
int p2m_cheb_basis_pair(p2m_graph_t g, const float* G, float* P1c, float* P2c, int32_t B, int32_t F, void* stream);
/* C[b*V + ids[i], :] = [A0[..] | A1 | A2] Bm + bias (+ addend): A0 is read at the actual row (>> a0_shift), A1/A2 at
 * the compact row b*n + i when planes_compact.  stats: [B * ceil(n/128)][2][N] per-sample tiles; for row_set 2 of a handle
 * with classes (p2m_graph_set_classes) every representative counts once per class member, as p2m_stats_rows_w would
 * count it (the tile weights for p2m_bn_finalize_split are the handle's).  in_scale / in_shift [Ka]
 * (optional; slice arithmetics, Ka <= 256): activation on load of PLANE 0, as in p2m_cheb_tile_gemm - A0 holds the raw
 * output y of the previous conv and the operand is max(fma(y, in_scale[k], in_shift[k]), 0) (A1 / A2 are then the planes
 * p2m_cheb_basis_fwd_real formed with the same act_scale / act_shift); a_amax must bound the activated operand.      */
int p2m_gemm_planes_rows(p2m_graph_t g, int32_t row_set, int32_t B, const float* A0, const float* A1,
                         const float* A2, int32_t nplanesA, int32_t Ka, int32_t a0_shift, int32_t planes_compact,
                         const float* Bm, const void* Bsplit, int32_t arith, const void* a_amax, int32_t a_bits,
                         const float* bias, const float* addend, float* C,
                         int32_t N, float* stats, const float* act_scale, const float* act_shift, int32_t act_relu,
                         void* amax_out, const float* in_scale /* or NULL */, const float* in_shift, void* stream);
int32_t p2m_rows_tiles_per_sample(p2m_graph_t g, int32_t row_set);
/* P[b*splits + s][k][n] = sum over the s-th slice of the row set of sample b of A[row][k] * G[row][n]
 * (G0 at the actual row, G1/G2 compact when planes_compact); Pdb likewise.  a_scale / a_shift [Ka] (optional, slice
 * arithmetics only): activation on load of A, as in p2m_cheb_tile_gemm - A holds the raw conv output y and the operand is
 * max(fma(y, a_scale[k], a_shift[k]), 0); a_amax then bounds that (p2m_act_bound).
 * splits <= -2 (round 6; slice arithmetics only): a chunk is -splits consecutive WHOLE samples instead of a slice of one -
 * P[c] (c < ceil(B / -splits)) sums the row sets of samples c * -splits ... : fewer partials where a sample has few rows.  */
int p2m_gemm_tn_rows(p2m_graph_t g, int32_t row_set, int32_t B, const float* A, int32_t Ka, int32_t a0_shift,
                     const float* G0, const float* G1, const float* G2, int32_t nplanesG, int32_t Gc,
                     int32_t planes_compact, int32_t splits, float* P, float* Pdb, int32_t arith,
                     const void* a_amax, const void* g_amax, int32_t g_bits, const float* a_scale,
                     const float* a_shift, void* stream);
/* We[k][n] = Wt[k][n] + a Wt[Ka+k][n] + b Wt[2Ka+k][n]  (Wt = [3Ka, N]) */
int p2m_weight_eff(const float* Wt, float* We, int32_t Ka, int32_t N, float a, float b, void* stream);
/* dW (nn.Linear layout) from the real-vertex partials P[c][fin][k*Fout+fo] plus the fake-vertex partials
 * P2[c][fin][fo] entering plane k with factor (1, s1, s2)[k]; db from both.  accumulate != 0 adds into dW/db (the
 * caller's gradient buffer, e.g. a slice of optim.FlatAdam.flat_grad) instead of overwriting.                      */
int p2m_weight_grad_unpack2(const float* P, const float* Pdb, int32_t nchunks, const float* P2, const float* Pdb2,
                            int32_t nchunks2, float s1, float s2, float* dW, float* db, int32_t Fout, int32_t Fin,
                            int32_t K, int32_t accumulate, void* stream);
int p2m_bn_finalize_rows(const float* stats_a, int32_t tps_a, int32_t rows_a, const float* stats_b, int32_t tps_b,
                         int32_t rows_b, int32_t B, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                         float* shift, int32_t N, void* stream);

/* Same, for the two launches of one split conv, with the sizes (and, with classes, the weights) taken from the handle. */
int p2m_bn_finalize_split(p2m_graph_t g, const float* stats_real, const float* stats_fake, int32_t B,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                          int32_t N, void* stream);

/* ---- Chebyshev basis inside the contraction (default on levels with a tile plan) ---------------------------------
 * The real rows of one graph convolution (lib/models/backbones/cheby_graph_conv.py:16-37) in ONE kernel - the planes
 * T1 = L X, T2 = (2 L L - I) X are formed per tile in LDS, cut into bf16 slices and contracted without touching HBM:
 *     C[b, v, :] = [ A0[b, v >> a0_shift] | (L X)[b, v] | (L2 X)[b, v] ] W (+ bias) (+ addend[b, v, :])
 * over the rows of the level's tile plan `plan` (p2m_graph_plan_info):
 *     0  real vertices v of the level;        X [B, V, Ka],   A0 [B, V, Ka]   (usually A0 = X),   C [B, V, N]
 *     1  the same with an un-pooled input:     X [B, V/2, Ka], A0 [B, V/2, Ka] read at v >> 1,     C [B, V, N]
 *     2  the paired operator (row set 3):      X [B, V, Ka],   A0 [B, V/2, Ka] = S X (pair-sums),  C [B, V/2, N],
 *        planes S L X, S L2 X - the backward of an un-pooled conv at the coarse resolution
 * Bx = p2m_weight_split of the [3 Ka, N] operand (rows k * Ka + fin).  N in {64, 128, 256}, Ka % 32 == 0
 * (p2m_cheb_tile_gemm_supported).  Rows of C outside the row set are not touched (fake vertices: p2m_gemm_planes_rows,
 * row set 2 / 4, with p2m_weight_eff).  Optional: stats[B * ntiles(plan)][2][N] BatchNorm partials per (sample, tile)
 * for p2m_bn_finalize_tiles; E1 / E2 [B * nset, Ka]: the two gathered planes, compact, for the weight gradient
 * p2m_gemm_tn_rows - bitwise those of p2m_cheb_basis_fwd_real / p2m_cheb_basis_pair where the kernel gathers with the
 * same fmaf chain (P2M_ARITH_BF16X3; N = 256), equal to fp32 round-off where the gather itself runs on the matrix cores
 * (P2M_ARITH_F16X2 with N <= 128: the tile's operator as a dense fp16-sliced block, baked with the plan); act_*: fused eval-mode
 * BatchNorm + ReLU as in p2m_gemm_planes (excludes stats).  arith: P2M_ARITH_BF16X3 or P2M_ARITH_F16X2 (Bx split with
 * the same arith; x_amax = amax word bounding X and A0, the kernel adds the level's p2m_graph_plane_bits for the planes
 * it forms).  amax_out as in p2m_gemm_planes.
 * in_scale / in_shift [Ka] (optional; both slice arithmetics and both gather forms since round 5; no planes out): ACTIVATION ON LOAD - X and A0 hold the
 * raw output y of the previous conv and the operand is x = max(fma(y, in_scale[f], in_shift[f]), 0), its BatchNorm + ReLU
 * (lib/models/backbones/cheby_graph_conv.py:39, lib/models/meshnet.py:100) with the two roundings of p2m_bn_act_fwd, applied in
 * the producer waves between the global load and the LDS image: the activated tensor never exists in HBM.  With
 * P2M_ARITH_F16X2 x_amax must then bound x (p2m_act_bound); P2M_ARITH_BF16X3 needs no bound.
 * bnr_y / bnr_co / bnr_part (optional, all or none; round 6): the BatchNorm-backward REDUCTION of the layer in front, fused
 * into the store of C.  In the backward pass C is g = dL/dx with x = relu(batch_norm(y)) the input of this conv
 * (lib/models/backbones/cheby_graph_conv.py:39, lib/models/meshnet.py:100), and the next step is p2m_bn_bwd_reduce over g
 * and y.  With bnr_y = y [B, c_rows, N] and bnr_co = [4][N] (mean, invstd, scale, shift of that BatchNorm) every block sums
 * g m and g m yhat (m = [y scale + shift > 0], yhat = (y - mean) invstd - the arithmetic of p2m_bn_bwd_reduce) over the
 * rows it stores and writes bnr_part[slot][2][N], slot < p2m_cheb_tile_gemm_bnr_slots (0: this arithmetic / width has no
 * fused reduction).  The fake-vertex rows of C come from another launch: p2m_bn_bwd_reduce_fake sums those, and
 * p2m_bn_bwd_finalize takes both partial sets as one array.  Saves the stand-alone pass over g and y for the real rows.  */
int32_t p2m_cheb_tile_gemm_supported(p2m_graph_t g, int32_t plan, int32_t Ka, int32_t N);
int32_t p2m_cheb_tile_gemm_bnr_slots(p2m_graph_t g, int32_t plan, int32_t arith, int32_t N, int32_t B);
/* 1 when a p2m_cheb_tile_gemm launch of this arithmetic and output width forms the planes with the gather ON THE MATRIX
 * CORES (the tile's operator as a dense pre-sliced block, TilePlan::ltx / ltx3): P2M_ARITH_F16X2 (two scaled fp16 slices,
 * 4 samples per unit) and - round 5 - P2M_ARITH_BF16X3 (three exact bf16 slices of operator and operand, 2 samples per
 * unit; OPT-IN, environment P2M_MG_EXACT=1 read once per process: it measured equal to the VALU gather), N <= 128.  The
 * optional planes E1 / E2 then agree with p2m_cheb_basis_fwd_real to fp32 round-off instead of bitwise.              */
int32_t p2m_cheb_tile_gemm_mg(int32_t arith, int32_t N);
int p2m_cheb_tile_gemm(p2m_graph_t g, int32_t plan, const float* X, const float* A0, int32_t Ka, const void* Bx,
                       int32_t arith, const void* x_amax, const float* bias, const float* addend, float* C, int32_t N,
                       float* stats, float* E1, float* E2, const float* act_scale, const float* act_shift,
                       int32_t act_relu, void* amax_out, const float* in_scale, const float* in_shift,
                       const float* bnr_y, const float* bnr_co, float* bnr_part, int32_t B, void* stream);
/* Bound of x = max(fma(y, scale[f], shift[f]), 0) over a tensor y whose amax word is y_amax:
 *     atomic max of max_f fma(amax(y), |scale[f]|, max(shift[f], 0)) into the amax word `word`
 * (a true bound in fp32: fma and max are monotone) - the x_amax / a_amax of a consumer that applies the activation on
 * load.  N <= 4096.                                                                                                        */
int p2m_act_bound(const float* scale, const float* shift, int32_t N, const void* y_amax, void* word, void* stream);
/* Binades of headroom that cover the Chebyshev planes of a level: ceil(log2(max(1, max_v sum_u |L_vu|, max_v sum_u
 * |L2_vu|))) for plan 0 / 1 (|L x|, |L2 x| <= 2^bits max |x|), one more for plan 2 (the pair sums).                */
int32_t p2m_graph_plane_bits(p2m_graph_t g, int32_t plan);
int p2m_bn_finalize_tiles(p2m_graph_t g, int32_t plan, const float* stats_real, const float* stats_fake, int32_t B,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                          int32_t N, void* stream);

/* ---- classes of identical fake rows ------------------------------------------------------------------------
 * Inside the coarse-to-fine stack every descendant of a fake vertex is fake, isolated and produced by the same per-row
 * arithmetic from the same un-pooled value (meshnet.py:71-78: both children copy the parent): in the tree order the
 * descendants of one fake vertex at a finer level are an aligned run of 2^j bitwise IDENTICAL rows.  The caller, who
 * knows how many un-pool steps lie above a level, declares these runs with rep_of[V] (host array: the first row of the
 * run for its members, v itself for everything else).  Afterwards the handle's fake row sets (2 and 4) list only the
 * representatives, the other members ("holes") are never written nor read:
 *   forward   BatchNorm statistics count a representative once per member (the `stats` of p2m_gemm_planes_rows, or
 *             p2m_stats_rows_w over a finished tensor; then p2m_bn_finalize_split);
 *             p2m_cheb_combine_small (the final conv) fills the holes of its OUTPUT with the representative's value;
 *   backward  a representative carries the SUM of its class's gradients -- everything downstream (BatchNorm backward,
 *             contractions, pair-sums, the weight gradient) is linear in it; p2m_class_reduce forms that sum from the
 *             incoming gradient, the BatchNorm-backward passes take the handle (`classes`) to skip holes and to add
 *             the constant term once per member.
 * Results are those of the full computation (fp32 round-off class: the statistics' summation order changes).       */
int p2m_graph_fake_ids(p2m_graph_t g, int32_t* out /* host, n_fake entries */);
/* The REAL (non-padding) vertices of the level in COMPACT ROW ORDER: row i of every compact plane ([B * n_real, F]:
 * p2m_cheb_basis_fwd_real, the planes out of p2m_cheb_tile_gemm, planes_compact of p2m_gemm_planes_rows / p2m_gemm_tn_rows)
 * belongs to vertex out[i].  A permutation of the real vertex ids - since round 5 a LOCALITY order (greedy patches over the
 * level's graph, so that the 32-row tiles of the tile plans have small neighbourhood unions), not the ascending one;
 * P2M_TILE_ORDER=tree (environment, read once) keeps the ascending coarsening-tree order.  out: [n_real] int32, host.  */
int p2m_graph_real_ids(p2m_graph_t g, int32_t* out);
int p2m_graph_set_classes(p2m_graph_t g, const int32_t* rep_of /* host, V entries */);
int p2m_graph_class_info(p2m_graph_t g, int32_t counts[3] /* has classes, representatives, all fake vertices */);
/* weighted BatchNorm partials of the representatives of y [B*V, N]: stats [B * ceil(n_rep/128)][2][N] */
int p2m_stats_rows_w(p2m_graph_t g, const float* y, int32_t B, int32_t N, float* stats, void* stream);
/* out[r] = sum of in over the class of r (representatives), in[r] (real vertices), 0 (holes); in, out: [B*V, F] */
int p2m_class_reduce(p2m_graph_t g, const float* in, float* out, int32_t B, int32_t F, void* stream);

/* ---- composite: one Chebyshev graph convolution (cheby_graph_conv.py:5-40, K=3) ------------
 * Y = [X|L X|L2 X] Wt + bias, BatchNorm partials in `stats` (may be NULL).  T1/T2 are caller
 * workspaces of B*V*Fin floats each and hold the basis planes afterwards (saved for backward).
 * Convenience entry: native f32 MFMA arithmetic, unsplit rows; the network path composes the pieces itself
 * (p2m_cheb_basis_fwd_real + p2m_weight_split + p2m_gemm_planes_rows).                                  */
int p2m_chebconv_fwd(p2m_graph_t g, const float* X, const float* Wt, const float* bias,
                     float* T1, float* T2, float* Y, float* stats,
                     int32_t B, int32_t Fin, int32_t Fout, int32_t in_shift, void* stream);

/* ---- mesh losses of the train step, value AND gradient (lib/core/base.py:130-143, lib/core/loss.py) --------
 * losses[0..3] = L1(pred_mesh, gt_mesh)*w_vertex, normal-vector loss*w_normal, edge-length loss*w_edge,
 * L1(J_regressor @ (pred_mesh*1000), gt_pose)*w_joint, where pred_mesh[b, v] = cam_mesh[b, perm[v]] (the
 * perm-reverse gather of base.py:130).  grad_cam (optional) receives d(sum of the four)/d cam_mesh, [B, V0, 3],
 * zero on fake vertices.  faces: [F,3] int32; vf_ptr/vf_idx: CSR vertex -> (face*3 + corner).  The joint regressor
 * (dense [J, nv] in the reference, 107 non-zeros of 117 130 for h36m) is passed sparse, by joint (jr_*: CSR, used for
 * the regression) and by vertex (vj_*: CSC, used for the gradient).  valid_mesh: [B, nv] or NULL, valid_pose: [B, J]
 * or NULL (the reference's masks are [B, nv, 1] / [B, J, 1], data/Human36M/dataset.py:392-394).  A weight of 0
 * switches the term AND its gradient off (the reference does not evaluate the edge loss before
 * cfg.TRAIN.edge_loss_start, base.py:141-143).  workspace: p2m_mesh_loss_workspace(...) floats.                */
int64_t p2m_mesh_loss_workspace(int32_t B, int32_t nv, int32_t F, int32_t J);
int p2m_mesh_loss(const float* cam_mesh, int32_t V0, const int32_t* perm, int32_t nv, const float* gt_mesh,
                  const float* valid_mesh, const int32_t* faces, int32_t F, const int32_t* vf_ptr,
                  const int32_t* vf_idx, const int32_t* jr_ptr, const int32_t* jr_idx, const float* jr_val,
                  const int32_t* vj_ptr, const int32_t* vj_idx, const float* vj_val, int32_t J,
                  const float* gt_pose, const float* valid_pose, float w_vertex, float w_normal, float w_edge,
                  float w_joint, float* workspace, float* losses, float* grad_cam, int32_t B, void* stream);

/* CoordLoss of a small tensor (lib/core/loss.py:10-23; the lifted-pose term of lib/core/base.py:128,139), value and gradient
 * in one launch:  loss[0] = w * mean |pred * valid - target * valid|,  grad[i] = w * sign(..) * valid / n   (grad optional).
 * valid (optional): one mask value per `per_mask` consecutive elements (per_mask = 3: the reference's [B, J, 1] masks).   */
int p2m_coord_loss(const float* pred, const float* target, const float* valid, int32_t per_mask, int64_t n, float w,
                   float* loss, float* grad, void* stream);

/* ---- test-step / demo epilogue (lib/core/base.py:200-204, demo/run.py:169-171) -------------------------------
 * mesh[b, i] = scale * cam_mesh[b, perm[i]], i < nv  (tree order incl. fake vertices -> mesh-model vertex order;
 * scale = 1000 in the Tester, 1 in the demo);  joints[b, j] = sum_k jr_val[k] * mesh[b, jr_idx[k]] over CSR row j.
 * mesh: [B, nv, 3] or NULL; joints: [B, J, 3] or NULL.                                                          */
int p2m_mesh_epilogue(const float* cam_mesh, int32_t V0, const int32_t* perm, int32_t nv, float scale,
                      const int32_t* jr_ptr, const int32_t* jr_idx, const float* jr_val, int32_t J,
                      float* mesh, float* joints, int32_t B, void* stream);

/* ---- PoseNet, the 2D -> 3D lifter in front of MeshNet (lib/models/posenet.py:11-92) ---------------------------------
 * A 4096-wide residual MLP over B rows.  Every Linear (posenet.py:19,22,59,68: F.linear and its autograd) is a
 * weight-streaming contraction at these batch sizes and runs on p2m_gemm_tn, the reduction-split contraction with both
 * operands row-major over the reduction index:
 *   forward  z = a W^T + b      P[ch][m][n] = sum_{k in ch} a^T[k][m] * W^T[k][n]      (A = a^T, G = p2m_weight_pack(W, K = 1))
 *   dX       g_a = g_z W        P[ch][m][k] = sum_{n in ch} g_z^T[n][m] * W[n][k]      (A = g_z^T, G = W as nn.Linear stores it)
 *   dW       W.grad += g_z^T a  p2m_gemm_tn_acc: the whole reduction (the B rows) in one chunk, added straight into .grad
 * p2m_gemm_tn_acc: P[k][n] += sum_{r < M} A[r][k] * G[r][n]   (A: [M, Ka], G: [M, N], P: [Ka, N]; same arithmetics).  */
int p2m_gemm_tn_acc(const float* A, int32_t Ka, const float* G, int32_t N, int64_t M, float* P, int32_t arith,
                    const void* a_amax, const void* g_amax, void* stream);
/* The elementwise stage between two contractions, forward (posenet.py:28-38,79-87).  A block owns 32 columns and ALL B
 * rows of them, so BatchNorm1d's batch statistics are block-local and the whole stage is one launch:
 *   z = sum_{ch < nch} P[ch] + bias (+ resid)                       written to z (z == NULL: P is the tensor itself)
 *   has_bn: a = dropout(relu(batch_norm(z)));  else a = z           written row-major to a and / or TRANSPOSED to aT [F, B]
 * batch_norm: training != 0 -> batch mean / biased variance over the B rows, running statistics updated with `momentum`
 * and the unbiased variance (nn.BatchNorm1d); else the running statistics.  mean / invstd receive the statistics used
 * (saved for the backward).  dropout: rnd = uniform [0, 1) numbers drawn by the caller, keep where rnd >= p_drop, scale
 * 1 / (1 - p_drop) (nn.Dropout); rnd == NULL or p_drop == 0: identity.  amax_out: atomic max of |a| (P2M_ARITH_F16X2).
 * B_real (round 6; 0 = B): the first B_real of the B rows hold samples, the rest are PADDING - the contractions want
 * B >= 32 and B % 4 == 0, so the caller zero-pads any other batch (B = 1 of demo/run.py:160 included).  Padding rows are
 * left out of the batch statistics (mean, variance and the unbiased-variance factor use B_real) and their a / aT values
 * are stored as 0, so they stay exactly zero through every Linear; in the backward their gradient is 0 in and 0 out. */
int p2m_pn_stage_fwd(const float* P, int32_t nch, const float* bias, const float* resid, float* z, int32_t has_bn,
                     int32_t training, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, const float* rnd, float p_drop, float* a, float* aT, float* mean,
                     float* invstd, void* amax_out, int32_t B, int32_t F, int32_t B_real, void* stream);
/* ... and backward (autograd of the same lines): g_a = sum_{ch < nch} P[ch] is the gradient w.r.t. the stage's output a;
 *   has_bn: g_u = g_a * dropout mask * [batch_norm(z) > 0];  dbeta = sum_r g_u;  dgamma = sum_r g_u xhat;
 *           g_z = gamma invstd (g_u - dbeta / B - xhat dgamma / B)  (training)   /   gamma invstd g_u  (running statistics)
 *   else    g_z = g_a;
 *   g_z += addend (the residual branch's gradient, posenet.py:38);  stored row-major (gz) and transposed (gzT, [F, B]);
 *   dbias = sum_r g_z: the bias gradient of the Linear that produced z.  accumulate != 0: dgamma / dbeta / dbias are
 *   added into (the parameters' .grad), else overwritten.  z, mean, invstd: what the forward saved.                  */
int p2m_pn_stage_bwd(const float* P, int32_t nch, const float* addend, int32_t has_bn, int32_t training, const float* z,
                     const float* mean, const float* invstd, const float* gamma, const float* beta, const float* rnd,
                     float p_drop, float* gz, float* gzT, float* dgamma, float* dbeta, float* dbias, int32_t accumulate,
                     void* amax_out, int32_t B, int32_t F, int32_t B_real, void* stream);

/* ---- optimizer step over a flat fp32 buffer ------------------------------------------------
 * torch.optim.Adam semantics (lib/funcs_utils.py:92-96, stepped at lib/core/base.py:148): one fused
 * launch for the whole model.  grad is multiplied by grad_scale first (1/world_size after a
 * sum all-reduce).  step counts from 1.                                                         */
int p2m_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                  void* stream);

/* torch.optim.RMSprop semantics with the defaults the reference's yaml recipes use (lib/funcs_utils.py:87-91,
 * the asset/yaml recipes, `optimizer: 'rmsprop'`: alpha 0.99, eps 1e-8, no momentum, not centered):
 *   v = alpha v + (1-alpha) g^2;  p -= lr g / (sqrt(v) + eps),  g = grad * grad_scale.                              */
int p2m_rmsprop_step(float* param, const float* grad, float* square_avg, int64_t n, float lr, float alpha,
                     float eps, float grad_scale, void* stream);
/* The same two steps with the step-dependent scalars in DEVICE memory, hp = {lr, 1 - beta1^t, sqrt(1 - beta2^t),
 * grad_scale} (RMSprop reads hp[0] and hp[3]): the launches are then identical from step to step and can be part of a
 * captured hipGraph (train.GraphedTrainStep); the host refreshes hp before each replay.                            */
int p2m_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hp,
                      float beta1, float beta2, float eps, void* stream);
int p2m_rmsprop_step_dev(float* param, const float* grad, float* square_avg, int64_t n, const float* hp, float alpha,
                         float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* P2M_H_ */
