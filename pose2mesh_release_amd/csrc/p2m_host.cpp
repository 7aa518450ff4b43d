// libp2m_host.so -- host-side (CPU, no HIP) native helpers for the one-off graph preparation.
//
// Replaces the Python double loops of the reference's coarsening (lib/coarsening.py:153-211
// HEM_one_level, :214-258 compute_perm), which cost ~1.5 s per model/dataset construction.
// Results are bit-identical to the reference, including its quirks (documented below).
#include <cstdint>
#include <vector>

extern "C" {

// Greedy heavy-edge matching of one level (lib/coarsening.py:153-211).
//   rows (sorted ascending) / cols / vals : COO triplets of the symmetric weight matrix
//   order   : visiting order of the vertices (argsort of the weighted degree)
//   weights : per-vertex normaliser (degree)
//   cluster : out, length n_rows = rows[nnz-1] + 1
// returns the number of clusters, or -1 on bad input.
// Reference quirks reproduced exactly:
//   * the row table is filled with "count, then test for a new row": the first stored row owns
//     one extra entry (the first entry of the following row), the last stored row one fewer, and
//     rows are numbered by order of appearance (an empty row would shift all later ones);
//   * W_ii is taken to be the FIRST stored entry of the row, whatever its column;
//   * ties keep the earlier neighbour (strict >), candidates with score <= 0 never match.
int64_t p2m_hem_match(const int32_t* rows, const int32_t* cols, const double* vals, int64_t nnz,
                      const int64_t* order, int64_t n_order, const double* weights, int32_t* cluster) {
  if (nnz <= 0 || !rows || !cols || !vals || !order || !weights || !cluster) return -1;
  const int64_t n = (int64_t)rows[nnz - 1] + 1;
  std::vector<int64_t> first(n, 0), count(n, 0);
  {
    int64_t slot = 0;
    int32_t cur = rows[0];
    for (int64_t e = 0; e < nnz; ++e) {
      ++count[slot];
      if (rows[e] > cur) {
        cur = rows[e];
        if (slot + 1 >= n) return -1;
        first[++slot] = e;
      }
    }
  }
  std::vector<char> used(n, 0);
  for (int64_t i = 0; i < n; ++i) cluster[i] = 0;
  int32_t next_id = 0;
  const int64_t visits = n_order < n ? n_order : n;
  for (int64_t i = 0; i < visits; ++i) {
    const int64_t v = order[i];
    if (v < 0 || v >= n) return -1;
    if (used[v]) continue;
    used[v] = 1;
    int64_t mate = -1;
    double best = 0.0;
    const int64_t base = first[v];
    const double wvv = vals[base];
    for (int64_t q = 0; q < count[v]; ++q) {
      const int64_t u = cols[base + q];
      if (u < 0 || u >= n) return -1;
      double score = 0.0;
      if (!used[u]) score = (2. * vals[base + q] + wvv + vals[first[u]]) * 1. / (weights[v] + weights[u] + 1e-9);
      if (score > best) {
        best = score;
        mate = u;
      }
    }
    cluster[v] = next_id;
    if (mate >= 0) {
      cluster[mate] = next_id;
      used[mate] = 1;
    }
    ++next_id;
  }
  return next_id;
}

// One level of the binary-tree ordering (lib/coarsening.py:224-246): given the order of the
// coarse vertices and the parent id of each of the n fine vertices, emit for every coarse vertex
// its (up to two) children in ascending id order, padding with fresh fake ids n, n+1, ... .
// out has 2*m entries.  returns the number of fake ids used, or -1 if a parent has > 2 children.
int64_t p2m_tree_order_level(const int32_t* parent, int64_t n, const int64_t* coarse_order, int64_t m,
                             int64_t* out) {
  if (!parent || !coarse_order || !out || n < 0 || m < 0) return -1;
  int64_t ncluster = 0;
  for (int64_t i = 0; i < n; ++i)
    if (parent[i] + 1 > ncluster) ncluster = parent[i] + 1;
  std::vector<int64_t> kid0(ncluster, -1), kid1(ncluster, -1);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t p = parent[i];
    if (p < 0) return -1;
    if (kid0[p] < 0) kid0[p] = i;
    else if (kid1[p] < 0) kid1[p] = i;
    else return -1;
  }
  int64_t fake = n;
  for (int64_t j = 0; j < m; ++j) {
    const int64_t c = coarse_order[j];
    int64_t a = -1, b = -1;
    if (c >= 0 && c < ncluster) {
      a = kid0[c];
      b = kid1[c];
    }
    if (a < 0) a = fake++;   // childless (fake) parent: two fake children
    if (b < 0) b = fake++;   // singleton: one fake sibling
    out[2 * j] = a;
    out[2 * j + 1] = b;
  }
  return fake - n;
}

const char* p2m_host_version(void) { return "p2m-host 0.1"; }
}
