// Device-side helpers shared by the sweep kernels (sptrsv.hip: 1 / 2 / 4 / 8 real columns on the VALU with MFMA forward tiles;
// sptrsv16.hip: 16 real columns = 8 complex right-hand sides, every tile on the f64 MFMA pipe, interleaved vectors).
#pragma once
#include "device.hpp"

namespace hpddm_hip {

static constexpr int WG_THREADS  = 256;
static constexpr int NARROW      = 128;  // panels up to this padded width can be handled one wavefront per tile
static constexpr int WAVE_ROWS   = 256;  // a wavefront takes a whole supernode in the backward sweep up to this many rows

// Pointers read from a descriptor in memory lose their address space (the compiler falls back to FLAT instructions, which
// tie up the LDS counter as well): the kernels see the supernode through global-address-space pointers.
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(1))) *gcd_t;
typedef const dbl2 __attribute__((address_space(1)))   *gcd2_t;
typedef const int __attribute__((address_space(1)))    *gci_t;
typedef int int4v __attribute__((ext_vector_type(4)));
typedef const int4v __attribute__((address_space(1)))  *gci4_t;
struct SnView {
  gcd_t     F, G, dinv, FT, leaf;
  gci_t     rows, rel, cptr, crel;
  long long voff, soff, coff;
  int       n, c0, w, nb, ldw, wc, cs, s_in, nchild, s_out, ldh, tgs, nnzr, nnzc, c_in, c_out, t_r0, t_nr, t_rbeg, t_rend;
};
__device__ static inline SnView view(const SnDesc &d)
{
  SnView v;
  v.F = (gcd_t)d.F, v.G = (gcd_t)d.G, v.dinv = (gcd_t)d.dinv, v.FT = (gcd_t)d.FT, v.leaf = (gcd_t)d.leaf;
  v.ldh = d.ldh;
  v.rows = (gci_t)d.rows, v.rel = (gci_t)d.rel, v.cptr = (gci_t)d.cptr, v.crel = (gci_t)d.crel;
  v.voff = d.voff, v.soff = d.soff, v.coff = d.coff;
  v.n = d.n, v.c0 = d.c0, v.w = d.w, v.nb = d.nb, v.ldw = d.ldw, v.wc = d.wc, v.cs = d.cs, v.s_in = d.s_in, v.nchild = d.nchild, v.s_out = d.s_out;
  v.tgs = d.tgs, v.nnzr = d.nnzr, v.nnzc = d.nnzc, v.c_in = d.c_in, v.c_out = d.c_out;
  v.t_r0 = d.t_r0, v.t_nr = d.t_nr, v.t_rbeg = d.t_rbeg, v.t_rend = d.t_rend; // (the tile, in the per-tile copies of the descriptor)
  return v;
}
// the sections of a condensed leaf's blob (leaf_blob_layout of factor.hpp, in the address space of the kernels)
typedef const unsigned short __attribute__((address_space(1))) *gcu16_t;
struct LeafView {
  gcd_t   WT, srval, scval;
  gci_t   scrow;
  gcu16_t srptr, scptr, srcol;
};
__device__ static inline LeafView leaf_view(const SnView &d)
{
  typedef const char __attribute__((address_space(1))) *gcc_t;
  const LeafBlob lb = leaf_blob_layout(d.w, d.ldw, d.nb, d.nnzr, d.nnzc, d.cs);
  const gcc_t    p  = (gcc_t)d.leaf;
  LeafView       L;
  L.WT = (gcd_t)(p + lb.wt), L.srval = (gcd_t)(p + lb.srval), L.scval = (gcd_t)(p + lb.scval);
  L.scrow = (gci_t)(p + lb.scrow);
  L.srptr = (gcu16_t)(p + lb.srptr), L.scptr = (gcu16_t)(p + lb.scptr), L.srcol = (gcu16_t)(p + lb.srcol);
  return L;
}
// forward sweep: last column (scalar index) of the top block that row r holds an entry in -- r itself when the block is lower
// triangular (tgs = 0), the last column of r's diagonal tile when the LU factorisation swapped rows inside its tiles (SnDesc::tgs)
__device__ static inline int tri_last(int r, int tgs) { return (int)(((((long long)r >> tgs) + 1) << tgs) - 1); }
// LDS traffic between the lanes of ONE wavefront: make the writes land before the reads (no workgroup barrier)
__device__ static inline void wave_lds_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// The same without draining the global loads in flight (the workgroup-scope fences above wait for vmcnt(0) as well): the LDS
// instructions of one wavefront execute in program order, the counter wait makes the writes land, the barrier keeps the
// compiler from moving LDS accesses across
__device__ static inline void wave_lds_order()
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

typedef double v4f64 __attribute__((ext_vector_type(4)));

// the 16-column engine (sptrsv16.hip): one forward + backward sweep of the plan on 16 real columns of b / x (original numbering,
// batched layout of `mu` user right-hand sides per subdomain; complex factors: the 8 right-hand sides k0 .. k0 + 7, real factors:
// the 16 right-hand sides k0 .. k0 + 15)
void solve_block16(SolvePlan &P, const double *b, double *x, int mu, int k0, hipStream_t s);

} // namespace hpddm_hip
