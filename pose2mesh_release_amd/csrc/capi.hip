// Graph handles, error plumbing and the composite ChebConv entry point of libp2m_hip.so.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <map>
#include <utility>
#include <vector>

#include "p2m_common.h"

namespace p2m {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return P2M_ERR_HIP;
  }
  return P2M_OK;
}

static int upload(const void* src, size_t bytes, void** dst) {
  hipError_t e = hipMalloc(dst, bytes ? bytes : 4);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return P2M_ERR_NOMEM;
  }
  if (bytes) {
    e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
      return P2M_ERR_HIP;
    }
  }
  return P2M_OK;
}

// rows of a sparse operator, flattened: (source row, a, b) per entry
struct FlatRows {
  std::vector<int> rp{0}, src;
  std::vector<float> a, b;
  void push(int s, float x, float y) { src.push_back(s); a.push_back(x); b.push_back(y); }
  void end_row() { rp.push_back((int)src.size()); }
  int n() const { return (int)rp.size() - 1; }
};

// Greedy tiles of consecutive rows (<= TILE_RMAX rows, <= TILE_UCAP distinct source rows, <= TILE_ECAP entries) for
// k_basis_tile.  A row that does not fit a tile on its own leaves the plan empty (ntiles == 0: the row kernel stays in
// charge).  Entry order = the order of `fr` (the merged-CSR order): the tile kernel's fmaf chain is the row kernel's.
static inline unsigned short f16_bits(float v) {
  const _Float16 h = (_Float16)v;                    // round to nearest even
  unsigned short b;
  memcpy(&b, &h, sizeof(b));
  return b;
}

// Locality order of the real rows (round 5).  Tiles are runs of <= TILE_RMAX consecutive rows of the compact
// real-row order whose merged-CSR neighbourhoods have <= TILE_UCAP distinct source rows.  In the coarsening-tree order
// (lib/coarsening.py:214-258) 32 consecutive real rows touch ~130 rows, so the cap cuts the tiles at ~28 rows: 12 % of the
// 32-row MFMA tiles are padding and every output row fetches 4.05 union rows.  Here the order is rebuilt by greedy patch
// growing over the level's graph: a patch starts at the unassigned real vertex with the most assigned neighbours and keeps
// adding the frontier vertex whose merged row adds the FEWEST new union columns, until 32 rows or a cap.  Measured on the
// SMPL-like levels (tools/probes/tile_locality_probe.py, profiles/r05_tile_locality_probe.txt): 245 -> 230 tiles, 28.1 -> 30.0
// rows per tile, 4.05 -> 3.55 union rows per row at the finest level.  (The 2-ring of ANY 32-vertex patch of this degree-3..13
// mesh is ~100 rows: <= 3.0 rows per row is out of reach of a reordering.)  The order only permutes the compact row set
// (real_ids): storage order, r >> 1 un-pooling and the classes are untouched.  P2M_TILE_ORDER=tree keeps the tree order.
static std::vector<int> locality_order(const std::vector<int>& real_ids, int V, const int32_t* row_ptr, const int32_t* col,
                                       const std::vector<int>& rp, const std::vector<int>& mc) {
  const int n = (int)real_ids.size();
  std::vector<char> isreal(V, 0), assigned(V, 0), infront(V, 0), inbound(V, 0);
  for (int v : real_ids) isreal[v] = 1;
  std::vector<int> cnt(V, 0), stamp(V, -1), order, boundary, front;
  order.reserve(n);
  int done = 0, next_unassigned = 0, patch_id = 0;
  while (done < n) {
    // seed: the boundary vertex with the most assigned neighbours (ties: the smallest id), else the first unassigned one
    int seed = -1, best = -1;
    size_t keep = 0;
    for (size_t q = 0; q < boundary.size(); q++) {
      const int v = boundary[q];
      if (assigned[v]) continue;
      boundary[keep++] = v;
      if (cnt[v] > best || (cnt[v] == best && v < seed)) { best = cnt[v]; seed = v; }
    }
    boundary.resize(keep);
    if (seed < 0) {
      while (assigned[real_ids[next_unassigned]]) next_unassigned++;
      seed = real_ids[next_unassigned];
    }
    patch_id++;
    front.clear();
    int rows = 0, usize = 0, entries = 0;
    int cur = seed;
    while (true) {
      // add `cur` to the patch
      for (int j = rp[cur]; j < rp[cur + 1]; j++)
        if (stamp[mc[j]] != patch_id) { stamp[mc[j]] = patch_id; usize++; }
      entries += rp[cur + 1] - rp[cur];
      assigned[cur] = 1;
      order.push_back(cur);
      rows++;
      done++;
      for (int j = row_ptr[cur]; j < row_ptr[cur + 1]; j++) {
        const int w = col[j];
        cnt[w]++;
        if (isreal[w] && !assigned[w] && !infront[w]) { infront[w] = 1; front.push_back(w); }
      }
      if (rows >= TILE_RMAX) break;
      int pick = -1, pick_new = 1 << 30;
      for (int w : front) {
        if (assigned[w]) continue;
        int nw = 0;
        for (int j = rp[w]; j < rp[w + 1]; j++) nw += stamp[mc[j]] != patch_id;
        if (nw < pick_new || (nw == pick_new && w < pick)) { pick_new = nw; pick = w; }
      }
      if (pick < 0 || usize + pick_new > TILE_UCAP || entries + (rp[pick + 1] - rp[pick]) > TILE_ECAP) break;
      cur = pick;
    }
    for (int w : front) {                       // what is left of the frontier joins the boundary of the assigned region
      infront[w] = 0;
      if (!assigned[w] && !inbound[w]) { inbound[w] = 1; boundary.push_back(w); }
    }
  }
  return order;
}

// entry_bits: binades that bound the entries' magnitudes (|a|, |b| <= 2^entry_bits): the dense blocks are stored times
// 2^(14 - entry_bits)
static int build_tile_plan(const FlatRows& fr, int nsrc, TilePlan& pl, int entry_bits) {
  std::vector<int> tile_row{0}, tile_u{0}, ucol, erow{0};
  std::vector<float4> ent;
  std::vector<unsigned short> ltx, ltx3;
  std::vector<double> dense((size_t)64 * TILE_UPAD);
  const int lt_exp = 14 - entry_bits;
  std::vector<int> local(nsrc, -1), uni;
  const int n = fr.n();
  int i = 0;
  while (i < n) {
    uni.clear();
    int rows = 0, entries = 0;
    while (i + rows < n && rows < TILE_RMAX) {
      const int r = i + rows;
      const size_t before = uni.size();
      for (int j = fr.rp[r]; j < fr.rp[r + 1]; j++) {
        const int s = fr.src[j];
        if (local[s] < 0) { local[s] = 1; uni.push_back(s); }
      }
      const int len = fr.rp[r + 1] - fr.rp[r];
      if (rows > 0 && ((int)uni.size() > TILE_UCAP || entries + len > TILE_ECAP)) {
        for (size_t q = before; q < uni.size(); q++) local[uni[q]] = -1;   // undo this row
        uni.resize(before);
        break;
      }
      entries += len;
      rows++;
    }
    if ((int)uni.size() > TILE_UCAP || entries > TILE_ECAP) return P2M_OK;   // a single row too large: no plan
    std::sort(uni.begin(), uni.end());
    for (size_t q = 0; q < uni.size(); q++) local[uni[q]] = (int)q;
    const size_t lt0 = ltx.size(), lt30 = ltx3.size();
    ltx.resize(lt0 + TILE_LTX_ELEMS, 0);
    ltx3.resize(lt30 + TILE_LTX3_ELEMS, 0);
    std::fill(dense.begin(), dense.end(), 0.0);
    for (int r = i; r < i + rows; r++) {
      for (int j = fr.rp[r]; j < fr.rp[r + 1]; j++) {
        float4 e4;
        e4.x = fr.a[j]; e4.y = fr.b[j]; e4.w = 0.f;
        const int lc = local[fr.src[j]];
        memcpy(&e4.z, &lc, sizeof(int));
        ent.push_back(e4);
        // dense block: entries of one row that share a source column (the two children of an un-pooled input row) add up
        dense[(size_t)(r - i) * TILE_UPAD + lc] += (double)fr.a[j];
        dense[(size_t)(32 + r - i) * TILE_UPAD + lc] += (double)fr.b[j];
      }
      erow.push_back((int)ent.size());
    }
    for (int pi = 0; pi < 64; pi++)
      for (int u = 0; u < (int)uni.size(); u++) {
        const float y = (float)std::ldexp(dense[(size_t)pi * TILE_UPAD + u], lt_exp);
        const _Float16 h = (_Float16)y;
        const size_t at = lt0 + (((size_t)(u >> 4) * 2 * 2 + (size_t)((u >> 3) & 1)) * 64 + (size_t)pi) * 8 + (u & 7);
        ltx[at] = f16_bits((float)h);
        ltx[at + 2 * 64 * 8] = f16_bits(y - (float)h);
        // three exact bf16 slices of the fp32 coefficient (truncation: x = h + m + l, split3 of p2m_split.h), unscaled
        const float cf = (float)dense[(size_t)pi * TILE_UPAD + u];
        unsigned cb, hb, mb, lb;
        memcpy(&cb, &cf, 4);
        hb = cb & 0xFFFF0000u;
        float hf, r1, mf, r2;
        memcpy(&hf, &hb, 4);
        r1 = cf - hf;
        memcpy(&mb, &r1, 4);
        mb &= 0xFFFF0000u;
        memcpy(&mf, &mb, 4);
        r2 = r1 - mf;
        memcpy(&lb, &r2, 4);
        lb &= 0xFFFF0000u;
        const size_t at3 = lt30 + (((size_t)(u >> 4) * 3 * 2 + (size_t)((u >> 3) & 1)) * 64 + (size_t)pi) * 8 + (u & 7);
        ltx3[at3] = (unsigned short)(hb >> 16);
        ltx3[at3 + 2 * 64 * 8] = (unsigned short)(mb >> 16);
        ltx3[at3 + 2 * 2 * 64 * 8] = (unsigned short)(lb >> 16);
      }
    for (int c : uni) { ucol.push_back(c); local[c] = -1; }
    i += rows;
    tile_row.push_back(i);
    tile_u.push_back((int)ucol.size());
  }
  if (tile_row.size() < 2) return P2M_OK;
  std::vector<float> tile_cnt(tile_row.size() - 1);
  for (size_t q = 0; q + 1 < tile_row.size(); q++) tile_cnt[q] = (float)(tile_row[q + 1] - tile_row[q]);
  int rc;
  if ((rc = upload(tile_cnt.data(), sizeof(float) * tile_cnt.size(), (void**)&pl.tile_cnt)) != P2M_OK ||
      (rc = upload(tile_row.data(), sizeof(int) * tile_row.size(), (void**)&pl.tile_row)) != P2M_OK ||
      (rc = upload(tile_u.data(), sizeof(int) * tile_u.size(), (void**)&pl.tile_u)) != P2M_OK ||
      (rc = upload(ucol.data(), sizeof(int) * ucol.size(), (void**)&pl.ucol)) != P2M_OK ||
      (rc = upload(erow.data(), sizeof(int) * erow.size(), (void**)&pl.erow)) != P2M_OK ||
      (rc = upload(ent.data(), sizeof(float4) * ent.size(), (void**)&pl.ent)) != P2M_OK ||
      (rc = upload(ltx.data(), sizeof(unsigned short) * ltx.size(), (void**)&pl.ltx)) != P2M_OK ||
      (rc = upload(ltx3.data(), sizeof(unsigned short) * ltx3.size(), (void**)&pl.ltx3)) != P2M_OK)
    return rc;
  pl.lt_exp = lt_exp;
  pl.ntiles = (int)tile_row.size() - 1;
  return P2M_OK;
}

}  // namespace p2m

using namespace p2m;

extern "C" const char* p2m_last_error_string(void) { return g_err; }

extern "C" int p2m_stream_capture_id(void* stream, unsigned long long* id_out) {
  if (!id_out) { set_error("p2m_stream_capture_id: id_out is null"); return P2M_ERR_INVALID; }
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipError_t e = hipStreamGetCaptureInfo((hipStream_t)stream, &st, &id);
  if (e != hipSuccess) { set_error("hipStreamGetCaptureInfo failed: %s", hipGetErrorString(e)); return P2M_ERR_HIP; }
  *id_out = st == hipStreamCaptureStatusActive ? (id ? id : ~0ull) : 0ull;
  return 0;
}

extern "C" const char* p2m_version(void) { return "p2m-hip 0.7 (gfx950; PoseNet stages at any batch, locality-ordered tiles, two-barrier tile kernel, two-block plane contraction; fp32 contractions as 2 scaled fp16 slices or 3 exact bf16 slices on the matrix pipe, or on the f32 MFMA; basis inside the contraction, paired operator, fake-row classes, activation on load)"; }

// Host-side bake: merged CSR of L and L2 = 2*L*L - I (double accumulation, one rounding to fp32).
extern "C" int p2m_graph_create(const int32_t* row_ptr, const int32_t* col, const float* val, int32_t V, int32_t nnz,
                                p2m_graph_t* out) {
  P2M_CHECK_ARG(row_ptr && col && val && out && V > 0 && nnz >= 0, "null pointer or empty graph");
  P2M_CHECK_ARG(row_ptr[0] == 0 && row_ptr[V] == nnz, "row_ptr inconsistent with nnz");
  for (int i = 0; i < V; i++) P2M_CHECK_ARG(row_ptr[i] <= row_ptr[i + 1], "row_ptr not monotone");
  for (int j = 0; j < nnz; j++) P2M_CHECK_ARG(col[j] >= 0 && col[j] < V, "column index out of range");

  std::vector<int> rp(V + 1, 0), mc;
  std::vector<float> ma, mb;
  std::vector<double> accA(V, 0.0), accB(V, 0.0);
  std::vector<int> mark(V, -1), touched;
  int max_row = 0;
  for (int i = 0; i < V; i++) {
    touched.clear();
    auto touch = [&](int c) {
      if (mark[c] != i) {
        mark[c] = i;
        accA[c] = 0.0;
        accB[c] = 0.0;
        touched.push_back(c);
      }
    };
    for (int j = row_ptr[i]; j < row_ptr[i + 1]; j++) {
      const int k = col[j];
      const double lik = (double)val[j];
      touch(k);
      accA[k] += lik;  // duplicates in the input are summed (COO semantics of the reference tensor)
      for (int q = row_ptr[k]; q < row_ptr[k + 1]; q++) {
        touch(col[q]);
        accB[col[q]] += 2.0 * lik * (double)val[q];
      }
    }
    touch(i);
    accB[i] -= 1.0;
    std::sort(touched.begin(), touched.end());
    for (int c : touched) {
      mc.push_back(c);
      ma.push_back((float)accA[c]);
      mb.push_back((float)accB[c]);
    }
    rp[i + 1] = (int)mc.size();
    max_row = std::max(max_row, rp[i + 1] - rp[i]);
  }
  // |L x|, |L2 x| <= (max row sum of |L|, |L2|) max |x|: the headroom the two-fp16-slice contractions give the planes
  double row_sum_max = 1.0;
  for (int i = 0; i < V; i++) {
    double sa = 0.0, sb = 0.0;
    for (int j = rp[i]; j < rp[i + 1]; j++) { sa += std::fabs((double)ma[j]); sb += std::fabs((double)mb[j]); }
    row_sum_max = std::max(row_sum_max, std::max(sa, sb));
  }
  int plane_bits = 0;
  while (std::ldexp(1.0, plane_bits) < row_sum_max * (1.0 + 1e-6) && plane_bits < 30) plane_bits++;
  // fake (isolated) vertices: single-entry rows sharing the most common (a, b) pair
  std::vector<int> real_ids, fake_ids;
  float fa = 0.f, fb = 0.f;
  {
    std::map<std::pair<float, float>, int> count;
    for (int i = 0; i < V; i++)
      if (rp[i + 1] - rp[i] == 1 && mc[rp[i]] == i) count[{ma[rp[i]], mb[rp[i]]}]++;
    int best = 0;
    for (const auto& kv : count)
      if (kv.second > best) { best = kv.second; fa = kv.first.first; fb = kv.first.second; }
    for (int i = 0; i < V; i++) {
      const bool fake = best > 0 && rp[i + 1] - rp[i] == 1 && mc[rp[i]] == i && ma[rp[i]] == fa && mb[rp[i]] == fb;
      (fake ? fake_ids : real_ids).push_back(i);
    }
  }
  Graph* g = new Graph();
  g->V = V;
  g->n_real = (int)real_ids.size();
  g->n_fake = (int)fake_ids.size();
  // tile plans of the LDS-staged basis kernel (levels with a real/fake split only; small levels keep the row kernel)
  if (g->n_fake > 0 && g->n_real >= 256) {
    int rc2 = P2M_OK;
    {
      static const bool tree_order = [] { const char* e = getenv("P2M_TILE_ORDER"); return e && !strcmp(e, "tree"); }();
      if (!tree_order) real_ids = locality_order(real_ids, V, row_ptr, col, rp, mc);
    }
    for (int sh = 0; sh < 2 && rc2 == P2M_OK; sh++) {
      if (sh == 1 && (V & 1)) break;
      FlatRows fr;
      for (int v : real_ids) {
        for (int j = rp[v]; j < rp[v + 1]; j++) fr.push(mc[j] >> sh, ma[j], mb[j]);
        fr.end_row();
      }
      rc2 = build_tile_plan(fr, V >> sh, g->plan[sh], plane_bits);
    }
    // PAIRED operator (plan[2]): row c = (merged row 2c) + (merged row 2c+1), over the coarse vertices with at least one
    // real child.  S L g and S L2 g (S = the x2 un-pool's transpose, the pair-sum) in ONE pass over g: the backward
    // of a conv whose input was un-pooled then runs both of its contractions at the COARSE resolution (meshnet.py).
    if (rc2 == P2M_OK && !(V & 1)) {
      std::vector<char> is_fake(V, 0);
      for (int v : fake_ids) is_fake[v] = 1;
      std::vector<int> preal, pfake;
      FlatRows fr;
      for (int c = 0; c < V / 2; c++) {
        const int u = 2 * c, w = 2 * c + 1;
        if (is_fake[u] && is_fake[w]) { pfake.push_back(c); continue; }
        preal.push_back(c);
        int i = rp[u], j = rp[w];
        while (i < rp[u + 1] || j < rp[w + 1]) {                     // two sorted rows -> one, equal columns summed
          const int ci = i < rp[u + 1] ? mc[i] : V, cj = j < rp[w + 1] ? mc[j] : V;
          if (ci < cj) { fr.push(ci, ma[i], mb[i]); i++; }
          else if (cj < ci) { fr.push(cj, ma[j], mb[j]); j++; }
          else { fr.push(ci, (float)((double)ma[i] + (double)ma[j]), (float)((double)mb[i] + (double)mb[j])); i++; j++; }
        }
        fr.end_row();
      }
      if (preal.size() >= 128) {
        rc2 = build_tile_plan(fr, V, g->plan[2], plane_bits + 1);
        if (rc2 == P2M_OK && g->plan[2].ntiles > 0) {
          g->n_pair_real = (int)preal.size();
          g->n_pair_fake = (int)pfake.size();
          preal.resize(preal.size() + 64, 0);
          pfake.resize(pfake.size() + 64, 0);
          if ((rc2 = upload(preal.data(), sizeof(int) * preal.size(), (void**)&g->pair_real_ids)) == P2M_OK)
            rc2 = upload(pfake.data(), sizeof(int) * pfake.size(), (void**)&g->pair_fake_ids);
        }
      }
    }
    if (rc2 != P2M_OK) {
      p2m_graph_destroy(reinterpret_cast<p2m_graph_t>(g));
      return rc2;
    }
  }
  // 64 zero entries of slack: the pipelined kernels prefetch ids a few 16-row stages ahead without bounds checks
  real_ids.resize(real_ids.size() + 64, 0);
  fake_ids.resize(fake_ids.size() + 64, 0);
  g->fake_a = fa;
  g->fake_b = fb;
  g->nnz = (int)mc.size();
  g->nnz_L = nnz;
  g->max_row = max_row;
  g->plane_bits = plane_bits;
  int rc;
  if ((rc = upload(rp.data(), sizeof(int) * (V + 1), (void**)&g->rowptr)) != P2M_OK ||
      (rc = upload(mc.data(), sizeof(int) * mc.size(), (void**)&g->col)) != P2M_OK ||
      (rc = upload(ma.data(), sizeof(float) * ma.size(), (void**)&g->a)) != P2M_OK ||
      (rc = upload(mb.data(), sizeof(float) * mb.size(), (void**)&g->b)) != P2M_OK ||
      (rc = upload(real_ids.data(), sizeof(int) * real_ids.size(), (void**)&g->real_ids)) != P2M_OK ||
      (rc = upload(fake_ids.data(), sizeof(int) * fake_ids.size(), (void**)&g->fake_ids)) != P2M_OK) {
    p2m_graph_destroy(reinterpret_cast<p2m_graph_t>(g));
    return rc;
  }
  *out = reinterpret_cast<p2m_graph_t>(g);
  return P2M_OK;
}

extern "C" int p2m_graph_destroy(p2m_graph_t gh) {
  if (!gh) return P2M_OK;
  Graph* g = reinterpret_cast<Graph*>(gh);
  if (g->rowptr) (void)hipFree(g->rowptr);
  if (g->col) (void)hipFree(g->col);
  if (g->a) (void)hipFree(g->a);
  if (g->b) (void)hipFree(g->b);
  if (g->real_ids) (void)hipFree(g->real_ids);
  if (g->fake_ids) (void)hipFree(g->fake_ids);
  if (g->pair_real_ids) (void)hipFree(g->pair_real_ids);
  if (g->pair_fake_ids) (void)hipFree(g->pair_fake_ids);
  if (g->w) (void)hipFree(g->w);
  if (g->live_ids) (void)hipFree(g->live_ids);
  if (g->live_pairs) (void)hipFree(g->live_pairs);
  if (g->rep_of) (void)hipFree(g->rep_of);
  if (g->fake_wts) (void)hipFree(g->fake_wts);
  if (g->fake_tile_w) (void)hipFree(g->fake_tile_w);
  for (TilePlan& pl : g->plan) {
    if (pl.tile_row) (void)hipFree(pl.tile_row);
    if (pl.tile_u) (void)hipFree(pl.tile_u);
    if (pl.ucol) (void)hipFree(pl.ucol);
    if (pl.erow) (void)hipFree(pl.erow);
    if (pl.ent) (void)hipFree(pl.ent);
    if (pl.tile_cnt) (void)hipFree(pl.tile_cnt);
    if (pl.ltx) (void)hipFree(pl.ltx);
    if (pl.ltx3) (void)hipFree(pl.ltx3);
  }
  delete g;
  return P2M_OK;
}

extern "C" int p2m_graph_info(p2m_graph_t gh, int32_t info[4]) {
  P2M_CHECK_ARG(gh && info, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  info[0] = g->V;
  info[1] = g->nnz_L;
  info[2] = g->nnz;
  info[3] = g->max_row;
  return P2M_OK;
}

extern "C" int32_t p2m_graph_plane_bits(p2m_graph_t gh, int32_t plan) {
  if (!gh || plan < 0 || plan > 2) return 0;
  return reinterpret_cast<const Graph*>(gh)->plane_bits + (plan == 2 ? 1 : 0);
}

extern "C" int p2m_graph_split_info(p2m_graph_t gh, int32_t counts[2], float coef[2]) {
  P2M_CHECK_ARG(gh && counts && coef, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  counts[0] = g->n_real;
  counts[1] = g->n_fake;
  coef[0] = g->fake_a;
  coef[1] = g->fake_b;
  return P2M_OK;
}

extern "C" int p2m_graph_pair_info(p2m_graph_t gh, int32_t counts[2]) {
  P2M_CHECK_ARG(gh && counts, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  counts[0] = g->plan[2].ntiles > 0 ? g->n_pair_real : 0;
  counts[1] = g->plan[2].ntiles > 0 ? g->n_pair_fake : 0;
  return P2M_OK;
}

// ---- classes of identical fake rows ------------------------------------------------------------------------------
extern "C" int p2m_graph_fake_ids(p2m_graph_t gh, int32_t* out) {
  P2M_CHECK_ARG(gh && out, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  if (g->n_fake == 0) return P2M_OK;
  hipError_t e = hipMemcpy(out, g->fake_ids, sizeof(int) * g->n_fake, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); return P2M_ERR_HIP; }
  return P2M_OK;
}

extern "C" int p2m_graph_real_ids(p2m_graph_t gh, int32_t* out) {
  P2M_CHECK_ARG(gh && out, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  if (g->n_real == 0) return P2M_OK;
  hipError_t e = hipMemcpy(out, g->real_ids, sizeof(int) * g->n_real, hipMemcpyDeviceToHost);
  if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); return P2M_ERR_HIP; }
  return P2M_OK;
}

extern "C" int p2m_graph_set_classes(p2m_graph_t gh, const int32_t* rep_of) {
  P2M_CHECK_ARG(gh && rep_of, "null pointer");
  Graph* g = reinterpret_cast<Graph*>(gh);
  P2M_CHECK_ARG(g->w == nullptr, "classes already set on this handle");
  const int V = g->V;
  std::vector<int> fake(g->n_fake);
  if (g->n_fake) {
    const hipError_t e = hipMemcpy(fake.data(), g->fake_ids, sizeof(int) * g->n_fake, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("hipMemcpy D2H failed: %s", hipGetErrorString(e)); return P2M_ERR_HIP; }
  }
  std::vector<char> is_fake(V, 0);
  for (int v : fake) is_fake[v] = 1;
  std::vector<float> w(V, 1.f);
  std::vector<int> cnt(V, 0);
  for (int v = 0; v < V; v++) {
    const int r = rep_of[v];
    P2M_CHECK_ARG(r >= 0 && r <= v && rep_of[r] == r, "rep_of must point at the first row of a class");
    P2M_CHECK_ARG(r == v || (is_fake[v] && is_fake[r]), "only fake vertices can share a class");
    cnt[r]++;
  }
  for (int v = 0; v < V; v++)          // a class is a run of consecutive rows
    if (rep_of[v] != v) P2M_CHECK_ARG(rep_of[v - 1] == rep_of[v], "a class must be a run of consecutive rows");
  std::vector<int> reps, pfake;
  std::vector<float> wts;
  for (int v : fake) {
    if (rep_of[v] == v) { reps.push_back(v); wts.push_back((float)cnt[v]); w[v] = (float)cnt[v]; }
    else w[v] = 0.f;
  }
  if (!(V & 1))
    for (int c = 0; c < V / 2; c++)
      // 2c+1 is then either a hole of 2c's class or (on the coarsest level of the chain) a class of its own
      if (is_fake[2 * c] && is_fake[2 * c + 1] && rep_of[2 * c] == 2 * c) pfake.push_back(c);
  std::vector<int> live, live_pairs;
  for (int v = 0; v < V; v++)
    if (w[v] != 0.f) live.push_back(v);
  if (!(V & 1))
    for (int c = 0; c < V / 2; c++)
      if (w[2 * c] != 0.f || w[2 * c + 1] != 0.f) live_pairs.push_back(c);
  const int n_rep = (int)reps.size();
  std::vector<float> tile_w(cdiv(n_rep > 0 ? n_rep : 1, 128), 0.f);
  for (int i = 0; i < n_rep; i++) tile_w[i / 128] += wts[i];
  const int n_pf = (int)pfake.size();
  reps.resize(reps.size() + 64, 0);
  wts.resize(wts.size() + 64, 0.f);
  pfake.resize(pfake.size() + 64, 0);
  std::vector<int> rep_v(rep_of, rep_of + V);
  // upload everything into locals and commit to the handle only when all of it succeeded: on failure the handle is
  // exactly what it was ("no classes"), and a retry neither leaks nor sees half-set fields
  int *d_reps = nullptr, *d_pf = nullptr, *d_rep_of = nullptr, *d_live = nullptr, *d_live_pairs = nullptr;
  float *d_wts = nullptr, *d_tile_w = nullptr, *d_w = nullptr;
  int rc;
  if ((rc = upload(reps.data(), sizeof(int) * reps.size(), (void**)&d_reps)) != P2M_OK ||
      (rc = upload(pfake.data(), sizeof(int) * pfake.size(), (void**)&d_pf)) != P2M_OK ||
      (rc = upload(wts.data(), sizeof(float) * wts.size(), (void**)&d_wts)) != P2M_OK ||
      (rc = upload(tile_w.data(), sizeof(float) * tile_w.size(), (void**)&d_tile_w)) != P2M_OK ||
      (rc = upload(rep_v.data(), sizeof(int) * V, (void**)&d_rep_of)) != P2M_OK ||
      (rc = upload(live.data(), sizeof(int) * live.size(), (void**)&d_live)) != P2M_OK ||
      (rc = upload(live_pairs.data(), sizeof(int) * live_pairs.size(), (void**)&d_live_pairs)) != P2M_OK ||
      (rc = upload(w.data(), sizeof(float) * V, (void**)&d_w)) != P2M_OK) {
    for (void* q : {(void*)d_reps, (void*)d_pf, (void*)d_wts, (void*)d_tile_w, (void*)d_rep_of, (void*)d_live,
                    (void*)d_live_pairs, (void*)d_w})
      if (q) (void)hipFree(q);
    return rc;
  }
  g->n_live = (int)live.size();
  g->n_live_pairs = (int)live_pairs.size();
  g->n_fake_all = g->n_fake;
  g->fake_wts = d_wts;
  g->fake_tile_w = d_tile_w;
  g->rep_of = d_rep_of;
  g->live_ids = d_live;
  g->live_pairs = d_live_pairs;
  g->w = d_w;
  (void)hipFree(g->fake_ids);
  g->fake_ids = d_reps;
  g->n_fake = n_rep;
  if (g->pair_fake_ids) {
    (void)hipFree(g->pair_fake_ids);
    g->pair_fake_ids = d_pf;
    g->n_pair_fake = n_pf;
  } else {
    (void)hipFree(d_pf);
  }
  return P2M_OK;
}

extern "C" int p2m_graph_class_info(p2m_graph_t gh, int32_t counts[3]) {
  P2M_CHECK_ARG(gh && counts, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  counts[0] = g->w ? 1 : 0;
  counts[1] = g->n_fake;                                  // representatives (or all fake vertices without classes)
  counts[2] = g->w ? g->n_fake_all : g->n_fake;
  return P2M_OK;
}

extern "C" int p2m_graph_plan_info(p2m_graph_t gh, int32_t ntiles[3]) {
  P2M_CHECK_ARG(gh && ntiles, "null pointer");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  for (int i = 0; i < 3; i++) ntiles[i] = g->plan[i].ntiles;
  return P2M_OK;
}

extern "C" int p2m_chebconv_fwd(p2m_graph_t gh, const float* X, const float* Wt, const float* bias, float* T1,
                                float* T2, float* Y, float* stats, int32_t B, int32_t Fin, int32_t Fout,
                                int32_t in_shift, void* stream) {
  P2M_CHECK_ARG(gh, "null graph");
  const Graph* g = reinterpret_cast<const Graph*>(gh);
  int rc = p2m_cheb_basis_fwd(gh, X, T1, T2, B, Fin, in_shift, stream);
  if (rc != P2M_OK) return rc;
  return p2m_gemm_planes(X, T1, T2, 3, Fin, in_shift, Wt, nullptr, P2M_ARITH_F32, nullptr, 0, bias, nullptr, Y, nullptr,
                         nullptr, 1, Fout, 0, (int64_t)B * g->V, stats, nullptr, nullptr, 0, nullptr, stream);
}
