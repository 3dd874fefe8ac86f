// msgpu_graph.hip -- first step of the graph clean-up that follows the overlap path, on the tables already resident in
// HBM: findContractionEdges (src/main.cpp:183-190, 416-463) with sanityCheck (libms/src/kernel/sc.cpp:29-90).
//
// The reference fans one job per edge over the ThreadPool; each job walks the neighbours of an order's start vertex
// through hash-map look-ups under shared locks and appends to a mutex-guarded map.  Here: the adjacency is a CSR built
// with two atomic passes, every contained & primary EdgeOrder becomes one wavefront whose lanes take the neighbours of
// the start vertex (the per-neighbour tests are independent and only ANDed), edges are found by binary search in the
// (v1, v2)-sorted edge table, and the per-edge "first sane order" is a final pass.  Integer + a few fp64 adds.
#include <hip/hip_runtime.h>

#include "msgpu.h"
#include "msgpu_internal.h"

namespace msgpu {

__global__ __launch_bounds__(256) void k_degree(const msgpu_edge *edges, uint64_t n_edges, uint32_t *deg) {
  const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
  if (e >= n_edges) return;
  atomicAdd(&deg[edges[e].v1], 1u);
  atomicAdd(&deg[edges[e].v2], 1u);
}

__global__ __launch_bounds__(256) void k_fill_adj(const msgpu_edge *edges, uint64_t n_edges, const uint64_t *adj_off,
                                                  uint32_t *cursor, uint32_t *adj) {
  const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
  if (e >= n_edges) return;
  const uint32_t a = edges[e].v1, b = edges[e].v2;
  adj[adj_off[a] + atomicAdd(&cursor[a], 1u)] = static_cast<uint32_t>(e);
  adj[adj_off[b] + atomicAdd(&cursor[b], 1u)] = static_cast<uint32_t>(e);
}

// candidates: contained & primary orders (main.cpp:422); appended wave-aggregated
__global__ __launch_bounds__(256) void k_mark_contained(const msgpu_order *orders, uint64_t n_orders, uint32_t *cand,
                                                        uint32_t *n_cand) {
  const uint64_t o    = blockIdx.x * 256ull + threadIdx.x;
  const bool     want = o < n_orders && (orders[o].flags & (MSGPU_ORD_CONTAINED | MSGPU_ORD_PRIMARY)) ==
                                        (MSGPU_ORD_CONTAINED | MSGPU_ORD_PRIMARY);
  const uint64_t m = __ballot(want);
  if (!m) return;
  const int      lane   = threadIdx.x & 63;
  const int      leader = __ffsll(static_cast<long long>(m)) - 1;
  uint32_t       base   = 0;
  if (lane == leader) base = atomicAdd(n_cand, static_cast<uint32_t>(__popcll(m)));
  base = __shfl(base, leader);
  if (want) cand[base + __popcll(m & ((1ull << lane) - 1))] = static_cast<uint32_t>(o);
}

__device__ __forceinline__ int64_t find_edge(const msgpu_edge *edges, uint64_t n_edges, uint32_t a, uint32_t b) {
  const uint32_t lo_v = a < b ? a : b, hi_v = a < b ? b : a;
  uint64_t       lo = 0, hi = n_edges;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    const uint2    v   = *reinterpret_cast<const uint2 *>(&edges[mid]); // (v1, v2)
    if (v.x < lo_v || (v.x == lo_v && v.y < hi_v)) lo = mid + 1;
    else hi = mid;
  }
  if (lo < n_edges && edges[lo].v1 == lo_v && edges[lo].v2 == hi_v) return static_cast<int64_t>(lo);
  return -1;
}

struct OrderLite { // the fields of an EdgeOrder sanityCheck reads
  double   left, right;
  uint32_t flags, start, end, base;
};
__device__ __forceinline__ OrderLite load_order(const msgpu_order *o) {
  OrderLite r;
  r.left  = o->left_offset;
  r.right = o->right_offset;
  r.flags = o->flags;
  r.start = o->start;
  r.end   = o->end;
  r.base  = o->base;
  return r;
}
__device__ __forceinline__ bool odir(const OrderLite &o) { return (o.flags & MSGPU_ORD_DIR) != 0; }
__device__ __forceinline__ bool ocont(const OrderLite &o) { return (o.flags & MSGPU_ORD_CONTAINED) != 0; }

// sanityCheck(graph, subnode, node, target, order, wiggleRoom), sc.cpp:29-90; checkOnEdge = (node, target),
// checkForEdge = (subnode, target)
__device__ bool sanity_check(const msgpu_edge *edges, const msgpu_order *orders, int64_t e_on, int64_t e_for,
                             uint32_t node, uint32_t target, const OrderLite &order, double wiggle) {
  const msgpu_edge on_e = edges[e_on], for_e = edges[e_for];
  for (uint32_t i = 0; i < on_e.order_cnt; ++i) {
    const OrderLite on = load_order(&orders[on_e.order_off + i]);
    for (uint32_t j = 0; j < for_e.order_cnt; ++j) {
      const OrderLite fr   = load_order(&orders[for_e.order_off + j]);
      bool            sane = (odir(order) == odir(on)) == odir(fr); // Toggle * Toggle is XNOR, :36
      if (ocont(fr) && ocont(on)) {                                  // :41-43
        sane = sane && (fr.start == target || fr.end == target) && on.start == target;
      } else if (ocont(fr) && !ocont(on)) { // :44-70
        if (fr.end != target) {
          bool l1 = false, l2 = false, l3 = false;
          if (on.end == target) { // both arms of the reference's condition (:50-52) reduce to this
            if (!odir(order)) l2 = true;
          } else {
            l1 = true;
            l3 = true;
            if (odir(order)) l2 = true;
          }
          if (!odir(order) && order.base != order.end) l1 = !l1;
          if (!odir(fr) && fr.base != fr.end) l2 = !l2;
          const double d1 = l1 ? order.left : order.right;
          const double d2 = l2 ? fr.left : fr.right;
          const double d3 = l3 ? on.left : on.right;
          sane = sane && (d1 + d2 + d3) < wiggle;
        }
      } else if (!ocont(fr) && ocont(on)) { // :71-72
        sane = sane && on.start == target;
      } else { // :73-82
        bool d1 = fr.start == target, d2 = on.start == target;
        if (!odir(fr) && fr.base == target) d1 = !d1;
        if (!odir(on) && on.base == target) d2 = !d2;
        if (!odir(order)) d1 = !d1;
        sane = sane && d1 == d2;
      }
      if (sane) return true;
    }
  }
  return false;
}

// one wavefront per candidate order; lanes take the neighbours of its start vertex
__global__ __launch_bounds__(256) void k_check_contraction(const msgpu_edge *edges, uint64_t n_edges,
                                                           const msgpu_order *orders, const uint64_t *adj_off,
                                                           const uint32_t *adj, const uint32_t *cand,
                                                           const uint32_t *n_cand, double wiggle, uint8_t *sane_out) {
  const uint32_t n     = *n_cand;
  const uint32_t waves = gridDim.x * 4;
  const int      lane  = threadIdx.x & 63;
  for (uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n; c += waves) { // every wave reaches the end of the list
    const uint32_t  oi    = cand[c];
    const OrderLite order = load_order(&orders[oi]);
    const uint64_t  a0 = adj_off[order.start], a1 = adj_off[order.start + 1];
    bool            ok = true;
    for (uint64_t a = a0 + lane; a < a1; a += 64) {
      const uint32_t   s      = adj[a];
      const msgpu_edge sub    = edges[s];
      const uint32_t   target = sub.v1 == order.start ? sub.v2 : sub.v1;
      if (target == order.end || sub.shadow) continue; // main.cpp:432-434
      const int64_t e_on = find_edge(edges, n_edges, order.end, target);
      if (e_on < 0) { // main.cpp:436
        ok = false;
        break;
      }
      if (!sanity_check(edges, orders, e_on, s, order.end, target, order, wiggle)) {
        ok = false;
        break;
      }
    }
    if (__all(ok) && lane == 0) sane_out[oi] = 1;
  }
}

// first sane order of each edge (the `break` of main.cpp:457)
__global__ __launch_bounds__(256) void k_pick_contraction(const msgpu_edge *edges, uint64_t n_edges,
                                                          const uint8_t *sane, int64_t *out) {
  const uint64_t e = blockIdx.x * 256ull + threadIdx.x;
  if (e >= n_edges) return;
  const msgpu_edge ed = edges[e];
  int64_t          r  = -1;
  for (uint32_t k = 0; k < ed.order_cnt; ++k)
    if (sane[ed.order_off + k]) {
      r = static_cast<int64_t>(ed.order_off + k);
      break;
    }
  out[e] = r;
}

// ---- MatchMap::getEdgeMatches for a list of edges (msgpu_get_edgematches) -------------------------------------------------
__global__ __launch_bounds__(256) void k_em_counts(const msgpu_edge *edges, const uint32_t *sel, uint64_t n, uint32_t *cnt) {
  const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
  if (i < n) cnt[i] = edges[sel[i]].em_cnt;
}
// one wavefront per listed edge: its EdgeMatches (32 B each, two 16-byte halves per lane) move to out[off[i] ...)
__global__ __launch_bounds__(256) void k_em_gather(const msgpu_edge *edges, const msgpu_edgematch *ems, const uint32_t *sel,
                                                   uint64_t n, const uint64_t *off, msgpu_edgematch *out) {
  const uint64_t i = blockIdx.x * 4ull + (threadIdx.x >> 6);
  if (i >= n) return;
  const int        lane = threadIdx.x & 63;
  const msgpu_edge e    = edges[sel[i]];
  const uint4     *src  = reinterpret_cast<const uint4 *>(ems + e.em_off);
  uint4           *dst  = reinterpret_cast<uint4 *>(out + off[i]);
  for (uint32_t k = lane; k < 2 * e.em_cnt; k += 64) dst[k] = src[k];
}

void launch_em_counts(hipStream_t st, const msgpu_edge *edges, const uint32_t *sel, uint64_t n, uint32_t *cnt) {
  if (n) hipLaunchKernelGGL(k_em_counts, dim3((n + 255) / 256), dim3(256), 0, st, edges, sel, n, cnt);
}
void launch_em_gather(hipStream_t st, const msgpu_edge *edges, const msgpu_edgematch *ems, const uint32_t *sel, uint64_t n,
                      const uint64_t *off, msgpu_edgematch *out) {
  if (n) hipLaunchKernelGGL(k_em_gather, dim3((n + 3) / 4), dim3(256), 0, st, edges, ems, sel, n, off, out);
}

void launch_degree(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, uint32_t *deg) {
  if (n_edges) hipLaunchKernelGGL(k_degree, dim3((n_edges + 255) / 256), dim3(256), 0, st, edges, n_edges, deg);
}
void launch_fill_adj(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const uint64_t *adj_off,
                     uint32_t *cursor, uint32_t *adj) {
  if (n_edges)
    hipLaunchKernelGGL(k_fill_adj, dim3((n_edges + 255) / 256), dim3(256), 0, st, edges, n_edges, adj_off, cursor, adj);
}
void launch_mark_contained(hipStream_t st, const msgpu_order *orders, uint64_t n_orders, uint32_t *cand,
                           uint32_t *n_cand) {
  if (n_orders)
    hipLaunchKernelGGL(k_mark_contained, dim3((n_orders + 255) / 256), dim3(256), 0, st, orders, n_orders, cand, n_cand);
}
void launch_check_contraction(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const msgpu_order *orders,
                              const uint64_t *adj_off, const uint32_t *adj, const uint32_t *cand,
                              const uint32_t *n_cand, uint64_t n_orders, double wiggle, uint8_t *sane) {
  if (!n_orders || !n_edges) return;
  // the candidate count stays on the device: a fixed grid strides over the list (at most one wave per order)
  uint64_t blocks = (n_orders + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_check_contraction, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, st, edges, n_edges, orders,
                     adj_off, adj, cand, n_cand, wiggle, sane);
}
void launch_pick_contraction(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const uint8_t *sane,
                             int64_t *out) {
  if (n_edges)
    hipLaunchKernelGGL(k_pick_contraction, dim3((n_edges + 255) / 256), dim3(256), 0, st, edges, n_edges, sane, out);
}

} // namespace msgpu
