// Common host-side types of the MI355X-native RAS-apply library (libhpddm_hip.so).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace hpddm_hip {

typedef int32_t idx_t;

struct Error : public std::runtime_error {
  explicit Error(const std::string &s) : std::runtime_error(s) { }
};

#define HH_CHECK(cond, msg)                                                                      \
  do {                                                                                           \
    if (!(cond)) throw ::hpddm_hip::Error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
  } while (0)

// Symmetric-pattern graph of a sparse matrix (diagonal removed), CSR adjacency.
struct Graph {
  idx_t              n = 0;
  std::vector<idx_t> xadj;   // n+1
  std::vector<idx_t> adjncy; // xadj[n]
};

// Block (supernode) partition produced by the ordering: columns [blk_ptr[k], blk_ptr[k+1]) of the permuted matrix.
struct Ordering {
  std::vector<idx_t> perm;    // perm[new] = old
  std::vector<idx_t> iperm;   // iperm[old] = new
  std::vector<idx_t> blk_ptr; // nblk+1, ascending, blocks are in a topological (children-first) order
};

// Nested-dissection ordering of a symmetric-pattern graph (graph_nd.cpp).
//   leaf_size: stop dissecting below this many vertices (the leaf becomes one dense supernode)
void nested_dissection(const Graph &g, int leaf_size, Ordering &ord);

// Structure of the block factor (symbolic.cpp).  All indices are in the permuted numbering.
struct Symbolic {
  idx_t               n = 0, nblk = 0;
  std::vector<idx_t>  blk_ptr;   // nblk+1
  std::vector<idx_t>  parent;    // nblk, -1 for roots
  std::vector<int64_t> row_ptr;  // nblk+1: rows strictly below block k are rows[row_ptr[k] .. row_ptr[k+1])
  std::vector<idx_t>  rows;      // sorted ascending per block
  std::vector<idx_t>  height;    // nblk: 0 for leaves, 1 + max(children) otherwise
  int64_t             nnz_exact = 0; // structural nnz(L) of the scalar factor (incl. diagonal), no padding
  int64_t             nnz_stored = 0; // entries stored in the dense block layout (sum w*(w+1)/2 + nb*w)
  double              flops = 0;      // factorisation flops (Cholesky count)
};

void symbolic_factorization(const Graph &g, const Ordering &ord, Symbolic &sym);

void bind_thread_device(); // capi_subdomain.hip: hipSetDevice(the device of HpddmHipSetDevice) on the calling host thread
int host_thread_cap(); // OpenMP threads the host-side phases may use (numeric_host.cpp)

} // namespace hpddm_hip
