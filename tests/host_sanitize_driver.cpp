// Test infrastructure (tests/test_host_sanitizers.py): drives libp2m_host's two entry points under
// AddressSanitizer + UndefinedBehaviorSanitizer on a well-formed mesh-like graph and on malformed inputs.
// Exit code 0 = every call returned what it should and no sanitizer report was raised.
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

extern "C" {
int64_t p2m_hem_match(const int32_t* rows, const int32_t* cols, const double* vals, int64_t nnz, const int64_t* order,
                      int64_t n_order, const double* weights, int32_t* cluster);
int64_t p2m_tree_order_level(const int32_t* parent, int64_t n, const int64_t* coarse_order, int64_t m, int64_t* out);
const char* p2m_host_version(void);
}

#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main() {
  // a W x H grid graph with diagonal entries first in every row (as the reference's weight matrix has them)
  const int W = 37, H = 29, n = W * H;
  std::vector<int32_t> rows, cols;
  std::vector<double> vals, deg(n, 0.0);
  for (int v = 0; v < n; v++) {
    const int x = v % W, y = v / W;
    rows.push_back(v); cols.push_back(v); vals.push_back(0.0);
    const int nb[4][2] = {{x - 1, y}, {x + 1, y}, {x, y - 1}, {x, y + 1}};
    for (auto& q : nb)
      if (q[0] >= 0 && q[0] < W && q[1] >= 0 && q[1] < H) {
        rows.push_back(v); cols.push_back(q[1] * W + q[0]); vals.push_back(1.0 + 0.01 * ((v * 7 + q[0]) % 5));
        deg[v] += vals.back();
      }
  }
  const int64_t nnz = (int64_t)rows.size();
  std::vector<int64_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::vector<int32_t> cluster(n, -1);
  const int64_t nc = p2m_hem_match(rows.data(), cols.data(), vals.data(), nnz, order.data(), n, deg.data(), cluster.data());
  CHECK(nc > n / 2 - 1 && nc <= n);
  std::vector<int> members(nc, 0);
  for (int v = 0; v < n; v++) {
    CHECK(cluster[v] >= 0 && cluster[v] < nc);
    members[cluster[v]]++;
  }
  for (int64_t c = 0; c < nc; c++) CHECK(members[c] == 1 || members[c] == 2);
  // a shorter visiting order (the reference passes one entry per stored row) and the tree order of the result
  CHECK(p2m_hem_match(rows.data(), cols.data(), vals.data(), nnz, order.data(), n - 5, deg.data(), cluster.data()) > 0);
  CHECK(p2m_hem_match(rows.data(), cols.data(), vals.data(), nnz, order.data(), n, deg.data(), cluster.data()) == nc);
  std::vector<int64_t> corder(nc + 3);
  std::iota(corder.begin(), corder.end(), 0);                    // 3 ids beyond the clusters: childless (fake) parents
  std::vector<int64_t> out(2 * corder.size(), -7);
  const int64_t nfake = p2m_tree_order_level(cluster.data(), n, corder.data(), (int64_t)corder.size(), out.data());
  CHECK(nfake >= 6);
  std::vector<char> seen(n + nfake, 0);
  for (int64_t v : out) {
    CHECK(v >= 0 && v < n + nfake && !seen[v]);
    seen[v] = 1;
  }
  // malformed inputs: error codes, never a stray access
  CHECK(p2m_hem_match(nullptr, cols.data(), vals.data(), nnz, order.data(), n, deg.data(), cluster.data()) == -1);
  CHECK(p2m_hem_match(rows.data(), cols.data(), vals.data(), 0, order.data(), n, deg.data(), cluster.data()) == -1);
  std::vector<int64_t> bad_order(order);
  bad_order[3] = n + 100;
  CHECK(p2m_hem_match(rows.data(), cols.data(), vals.data(), nnz, bad_order.data(), n, deg.data(), cluster.data()) == -1);
  std::vector<int32_t> bad_cols(cols);
  bad_cols[1] = n + 1;                                          // an entry of the first visited row
  CHECK(p2m_hem_match(rows.data(), bad_cols.data(), vals.data(), nnz, order.data(), n, deg.data(), cluster.data()) == -1);
  std::vector<int32_t> three(n, 0);                               // every vertex in cluster 0: > 2 children
  CHECK(p2m_tree_order_level(three.data(), n, corder.data(), 1, out.data()) == -1);
  std::vector<int32_t> neg(n, -1);
  CHECK(p2m_tree_order_level(neg.data(), n, corder.data(), 1, out.data()) == -1);
  CHECK(p2m_tree_order_level(nullptr, n, corder.data(), 1, out.data()) == -1);
  CHECK(p2m_host_version() != nullptr);
  std::puts("host sanitizer driver ok");
  return 0;
}
