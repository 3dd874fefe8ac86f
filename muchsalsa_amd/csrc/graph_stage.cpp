// graph_stage.cpp -- the host phases of MuCHSALSA between the overlap path and assemblePath, on the flat result tables:
//
//   graph clean-up         src/main.cpp:194-288, 465-618  contraction targets / roots, ContainElements, vertex and
//                                                         edge deletions, computeBitweight, decycle
//   getMaxSpanTree         libms/src/kernel/mst.cpp:34-111
//   getShortestPath        include/ms/graph/Graph.h:927-978
//   getConnectedComponents libms/src/kernel/cc.cpp:33-70
//   getDirectedGraph       libms/src/kernel/dg.cpp:35-121
//   linearizeGraph         libms/src/kernel/lg.cpp:41-629  sortReductionByWeight, findClusterWeights[Heuristic],
//                                                         findConservationPathAlt, extractPaths, path joins
//   assemblePaths          src/main.cpp:620-661           per component: directed graph -> paths -> msgpu_path_input
//
// The reference keeps shared_ptr vertices/edges in hash maps (Graph.h:404-407), copies whole adjacency maps per
// neighbour query (Graph.cpp:260-287) and fans jobs over a ThreadPool with one global mutex.  Here every graph is FLAT:
// vertices and edges are dense indices, adjacency is CSR (offsets + (neighbour, edge) pairs sorted by neighbour id),
// deletion is a tombstone byte, a "copy" of a DiGraph is a second tombstone array over the same CSR, per-vertex state
// lives in arrays indexed by id with round stamps instead of per-call hash maps.  findContractionEdges (the only part
// with per-edge independent arithmetic) runs on the GPU (msgpu_find_contraction_edges) and is an input.
//
// Restructurings that keep the reference's results while dropping its quadratic loops and its one-thread passes:
//  * decycle (main.cpp:575-618) asks for the tree path of ~every non-tree edge but only folds strand parities over it
//    unless the parity is odd: the span forest is rooted once (parent, depth, parity-to-root), so the fold is two array
//    reads and the path itself (climb to the common ancestor) is only walked for the conflicting edges.
//  * extractPaths (lg.cpp:371-407) re-sorts the WHOLE remaining graph and re-runs findConservationPathAlt for every
//    path it peels off.  Both are local to a weakly connected component: the stack-based topological order restricted
//    to a component is the order that component has on its own, and components are visited in descending id of their
//    zero-in-degree vertices.  So each component keeps its own best path, a heap picks the winner the global pass would
//    pick (longest; ties: the sink that comes first in the global order), and only the component a path was taken from
//    is re-solved.
//
//  * getDirectedGraph's walk (dg.cpp:44-118) pushes a vertex once per neighbour that meets it before its first pop; with
//    every alive edge holding a kept order (always, after the clean-up) that is a depth-first search with the arcs taken
//    last to first, one frame per vertex; the directed edges are made afterwards, on all threads, in the reference's order.
//  * getMaxSpanTree (mst.cpp:75-111) is Kruskal's loop over a TOTAL order (weight, then edge order), so its forest is the
//    one minimum spanning forest of that order: large graphs find it with Boruvka's rounds on all threads, and its sets
//    double as getConnectedComponents' result when decycle removed no edge.
//  * the cycle-free copy of a component's DiGraph (lg.cpp:351-356: everything but the shadow edges, a few per cent of the
//    edges) has an adjacency of its own instead of a tombstone array over the full one.
// Where the stage's time goes and what each of these bought: profiles/r5_08/README.md.
//
// Iteration orders the reference leaves to hash containers (or to pointer VALUES: lg.cpp:419, main.cpp:211) are fixed
// as in DESIGN.md section 2, "canonical order": vertices ascending id, edges in creation order (table order for the undirected graph),
// neighbours ascending id, std::sort ties stable, pointer-ordered containers ordered by vertex id.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <sys/mman.h>
#include <type_traits>
#include <vector>

#include "host_pool.h"
#include "msgpu.h"

namespace {

constexpr int8_t D_NONE = 0, D_POS = 1, D_NEG = -1;
constexpr double BASE_WEIGHT_MULTIPLICATOR = 1.1; // src/main.cpp:96
constexpr double MAX_WEIGHT_MULTIPLICATOR  = 0.8; // src/main.cpp:97
constexpr uint32_t NIL = 0xffffffffu;

struct GraphError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

void require(bool ok, const char *what) {
  if (!ok) throw GraphError(what);
}

// MSGPU_GRAPH_DEBUG=1: phase timings on stderr -- the phase's own time, the moment it ended (ms since the entry point began)
// and the component the calling worker is on (the components of msgpu_graph_linearize run side by side)
std::chrono::steady_clock::time_point g_tick_epoch = std::chrono::steady_clock::now();
thread_local int                      tl_tick_tag  = -1;
struct Tick {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  bool on = std::getenv("MSGPU_GRAPH_DEBUG") != nullptr;
  void operator()(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    char tag[16] = "";
    if (tl_tick_tag >= 0) snprintf(tag, sizeof(tag), " c%d", tl_tick_tag);
    fprintf(stderr, "[graph%s @%7.2f] %-28s %8.3f s\n", tag, 1e3 * std::chrono::duration<double>(n - g_tick_epoch).count(), what,
            std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// host threads for the loops over all edges / orders of the stage (results never depend on the count): the cores this
// process may use, at most 16; MSGPU_GRAPH_THREADS overrides
// a worker that runs beside other workers (one component each in msgpu_graph_linearize) limits the loops it starts to its
// share of the threads, so that the stage never runs more threads than it was given (a CPU quota punishes that twice)
thread_local unsigned tl_thread_share = 0; // 0 = no limit
unsigned stage_threads_total() {
  if (const char *e = std::getenv("MSGPU_GRAPH_THREADS")) {
    const int v = std::atoi(e);
    if (v > 0) return static_cast<unsigned>(v > 64 ? 64 : v);
  }
  static const unsigned hw = std::thread::hardware_concurrency();
  return hw == 0 ? 1u : hw > 16 ? 16u : hw;
}
unsigned stage_threads() {
  const unsigned n = stage_threads_total();
  return tl_thread_share && tl_thread_share < n ? tl_thread_share : n;
}
// below this many items a loop stays on the calling thread (MSGPU_GRAPH_PAR_MIN overrides: tests run the threaded code on small graphs)
size_t par_min() {
  if (const char *e = std::getenv("MSGPU_GRAPH_PAR_MIN")) {
    const long v = std::atol(e);
    if (v > 0) return static_cast<size_t>(v);
  }
  return size_t(1) << 16;
}
struct JoinAll { // a thread that could not be started must not leave the started ones unjoined behind the exception
  std::vector<std::thread> &t;
  ~JoinAll() {
    for (auto &x : t)
      if (x.joinable()) x.join();
  }
};
// The stage's loops run on the library's pool of parked threads (host_pool.h): a loop that started its own threads paid
// 0.3-0.5 ms for them, and the stage has some thirty such loops on its critical path.  Loops of several callers (the
// component workers of msgpu_graph_linearize) share the pool.
using StagePool = msgpu::HostPool;

// f(chunk, begin, end) over [0, n) cut into contiguous chunks, one per thread of the stage; chunk indices ascend with the
// range, so per-chunk results concatenated in chunk order are in index order.  The first exception of a chunk is rethrown here.
template <class F> unsigned parallel_chunks(size_t n, F f) {
  unsigned nt = stage_threads();
  if (n < par_min()) nt = 1;
  if (nt <= 1) {
    f(0u, size_t(0), n);
    return 1;
  }
  const size_t per = (n + nt - 1) / nt;
  StagePool::get().run(nt, nt, [&](size_t c) { f(static_cast<unsigned>(c), std::min(n, per * c), std::min(n, per * (c + 1))); });
  return nt;
}

// f(begin, end) over [0, n) in pieces of `grain` handed out by a counter: for loops whose cost per index is uneven
template <class F> void parallel_dynamic(size_t n, size_t grain, F f) {
  if (std::getenv("MSGPU_GRAPH_STATIC")) { // measurement switch: contiguous ranges instead
    parallel_chunks(n, [&](unsigned, size_t b, size_t e) { f(b, e); });
    return;
  }
  unsigned nt = stage_threads();
  if (n < 4 * grain) nt = 1;
  if (nt <= 1) {
    f(size_t(0), n);
    return;
  }
  StagePool::get().run(nt, (n + grain - 1) / grain, [&](size_t k) { f(k * grain, std::min(n, (k + 1) * grain)); });
}

// std::sort on the stage's threads: stretches sorted side by side, then merged pairwise (log2 rounds; the keys sorted here
// are unique, so the result is the one std::sort gives)
template <class It, class Cmp> void parallel_sort(It first, It last, Cmp cmp, size_t min_n = 0 /*0: par_min()*/) {
  const size_t n  = static_cast<size_t>(last - first);
  unsigned     nt = stage_threads();
  if (n < (min_n ? min_n : par_min()) || nt <= 1) {
    std::sort(first, last, cmp);
    return;
  }
  std::vector<size_t> cut(nt + 1);
  for (unsigned k = 0; k <= nt; ++k) cut[k] = n * k / nt;
  StagePool::get().run(nt, nt, [&](size_t k) { std::sort(first + static_cast<long>(cut[k]), first + static_cast<long>(cut[k + 1]), cmp); });
  for (unsigned width = 1; width < nt; width *= 2) {
    const unsigned pairs = (nt + 2 * width - 1) / (2 * width);
    StagePool::get().run(pairs, pairs, [&](size_t p) {
      const unsigned lo = static_cast<unsigned>(p) * 2 * width, mid = std::min(nt, lo + width), hi = std::min(nt, lo + 2 * width);
      if (mid < hi) std::inplace_merge(first + static_cast<long>(cut[lo]), first + static_cast<long>(cut[mid]), first + static_cast<long>(cut[hi]), cmp);
    });
  }
}

// ---- flat adjacency ----------------------------------------------------------------------------------------------------

struct Arc {
  uint32_t to; // neighbour
  uint32_t e;  // edge index
  Arc() {}     // (left as it is: a vector of arcs is sized first and filled by the stage's threads, which also touch its pages first)
  Arc(uint32_t to_, uint32_t e_) : to(to_), e(e_) {}
};
// std::vector whose resize() leaves new elements of a plain type as they are (no pass of zeroes on the calling thread
// over tables the stage's threads are about to fill)
template <class T> struct LeaveAlone {
  using value_type = T;
  LeaveAlone() = default;
  template <class U> LeaveAlone(const LeaveAlone<U> &) noexcept {}
  // a large table asks for transparent huge pages (where the kernel hands them out on request): the stage's threads touch
  // tens of megabytes for the first time in every phase, one page fault per 4 KB otherwise (MSGPU_GRAPH_THP=0: measurement switch)
  static bool huge(size_t n) {
    static const bool on = [] {
      const char *e = std::getenv("MSGPU_GRAPH_THP");
      return !(e && e[0] == '0');
    }();
    return on && n * sizeof(T) >= (size_t(4) << 20);
  }
  T *allocate(size_t n) {
    if (huge(n)) {
      constexpr size_t HP = size_t(2) << 20;
      const size_t     bytes = (n * sizeof(T) + HP - 1) & ~(HP - 1);
      void            *p = nullptr;
      if (posix_memalign(&p, HP, bytes) != 0) throw std::bad_alloc();
      madvise(p, bytes, MADV_HUGEPAGE); // (advice: a refusal changes nothing)
      return static_cast<T *>(p);
    }
    return std::allocator<T>().allocate(n);
  }
  void deallocate(T *p, size_t n) noexcept {
    if (huge(n)) std::free(p);
    else std::allocator<T>().deallocate(p, n);
  }
  template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
  template <class U> bool operator==(const LeaveAlone<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const LeaveAlone<U> &) const noexcept { return false; }
};
template <class T> using RawVec = std::vector<T, LeaveAlone<T>>;

struct Csr { // segment of vertex v: arcs[off[v] .. off[v+1]), ascending `to`
  std::vector<uint32_t> off;
  RawVec<Arc>           arcs;
  const Arc *begin(uint32_t v) const { return arcs.data() + off[v]; }
  const Arc *end(uint32_t v) const { return arcs.data() + off[v + 1]; }
  const Arc *find(uint32_t v, uint32_t to) const {
    const Arc *lo = begin(v), *hi = end(v);
    lo = std::lower_bound(lo, hi, to, [](const Arc &a, uint32_t t) { return a.to < t; });
    return lo != hi && lo->to == to ? lo : nullptr;
  }
};

// Graph::deleteVertex (Graph.cpp:158-185): the vertex goes, and with it every edge at it -- here: a tombstone per incident
// edge (`bury`), over the out-arcs and, for a DiGraph, the in-arcs.  Shared by msgpu_graph::delete_vertex (the clean-up's
// deletions, src/main.cpp:243,259,286) and msgpu_graph_bookkeeping (the reference's own test vectors for it).
template <class Bury> inline void bury_vertex(const Csr &out, const Csr *in, uint32_t v, Bury &&bury) {
  for (const Arc *t = out.begin(v); t != out.end(v); ++t) bury(t->e);
  if (in)
    for (const Arc *t = in->begin(v); t != in->end(v); ++t) bury(t->e);
}
// Graph::getNeighbors / DiGraph::getSuccessors / getPredecessors (Graph.cpp:232-287): the other ends of the vertex's living
// edges, ascending (the reference hands out a hash map: no order)
template <class Alive, class Emit> inline void living_neighbours(const Csr &c, uint32_t v, Alive &&alive, Emit &&emit) {
  for (const Arc *t = c.begin(v); t != c.end(v); ++t)
    if (alive(t->e)) emit(t->to);
}

// CSR of n vertices from m (from[i] -> to[i]) pairs, edge index i; both_ways = undirected.  Segments end up ascending
// in `to` (counting sort by source keeps input order; a segment that is not ascending already is sorted).
// The undirected adjacency of a large graph on all host threads, arc for arc what the serial builder below produces: every
// thread owns a contiguous stretch of the edge table and counts, per vertex, the edges of its stretch that have the vertex
// as their HIGHER end and as their LOWER end; a prefix over (vertex, thread) turns the counts into the first slot of every
// thread inside the vertex's "lower neighbours" and "higher neighbours" runs (both in edge order); the threads then drop
// their arcs without meeting each other.
bool build_csr_undirected_parallel(uint32_t n, const uint32_t *from, const uint32_t *to, size_t m, Csr &c) {
  const unsigned nt = stage_threads();
  if (std::getenv("MSGPU_GRAPH_SERIAL_CSR")) return false; // measurement switch
  if (nt < 2 || m < par_min() || m < 2 * static_cast<size_t>(n) || static_cast<size_t>(nt) * 2 * (static_cast<size_t>(n) + 1) > (size_t(1) << 26)) return false;
  const size_t          stride = static_cast<size_t>(n) + 1;
  std::unique_ptr<uint32_t[]> hist(new uint32_t[static_cast<size_t>(nt) * 2 * stride]); // [thread][hi | lo][vertex]; zeroed by its threads
  auto chunk = [&](unsigned t) { return std::make_pair(m * t / nt, m * (t + 1) / nt); };
  auto on_threads = [&](auto &&body) { // body(t) for t < nt
    StagePool::get().run(nt, nt, [&](size_t t) { body(static_cast<unsigned>(t)); });
  };
  on_threads([&](unsigned t) {
    uint32_t *hi = hist.get() + static_cast<size_t>(t) * 2 * stride, *lo = hi + stride;
    std::fill(hi, hi + 2 * stride, 0u);
    for (size_t i = chunk(t).first; i < chunk(t).second; ++i) {
      const uint32_t a = from[i], b = to[i];
      ++hi[a > b ? a : b];
      ++lo[a > b ? b : a];
    }
  });
  c.off.assign(stride, 0);
  on_threads([&](unsigned t) { // vertices [v0, v1): counts -> first slot of every thread inside the vertex's two runs
    const size_t v0 = stride * t / nt, v1 = stride * (t + 1) / nt;
    for (size_t v = v0; v < v1; ++v) {
      uint32_t run = 0;
      for (unsigned k = 0; k < nt; ++k) {
        uint32_t &h = hist[static_cast<size_t>(k) * 2 * stride + v];
        const uint32_t cnt = h;
        h                  = run;
        run += cnt;
      }
      for (unsigned k = 0; k < nt; ++k) {
        uint32_t &l = hist[static_cast<size_t>(k) * 2 * stride + stride + v];
        const uint32_t cnt = l;
        l                  = run;
        run += cnt;
      }
      if (v < n) c.off[v + 1] = run; // (slot n of the histograms is never counted into: vertex ids are < n)
    }
  });
  for (uint32_t v = 0; v < n; ++v) c.off[v + 1] += c.off[v];
  c.arcs.resize(c.off[n]);
  on_threads([&](unsigned t) {
    uint32_t *hi = hist.get() + static_cast<size_t>(t) * 2 * stride, *lo = hi + stride;
    for (size_t i = chunk(t).first; i < chunk(t).second; ++i) {
      const uint32_t a = from[i], b = to[i], h = a > b ? a : b, l = a > b ? b : a;
      c.arcs[c.off[h] + hi[h]++] = Arc{l, static_cast<uint32_t>(i)};
      c.arcs[c.off[l] + lo[l]++] = Arc{h, static_cast<uint32_t>(i)};
    }
  });
  on_threads([&](unsigned t) {
    for (size_t v = static_cast<size_t>(n) * t / nt; v < static_cast<size_t>(n) * (t + 1) / nt; ++v) {
      if (c.off[v + 1] - c.off[v] < 2) continue;
      Arc *b = c.arcs.data() + c.off[v], *e = c.arcs.data() + c.off[v + 1];
      bool sorted = true;
      for (Arc *p = b; p + 1 < e && sorted; ++p) sorted = p->to <= (p + 1)->to;
      if (!sorted) std::stable_sort(b, e, [](const Arc &x, const Arc &y) { return x.to < y.to; });
    }
  });
  return true;
}

// The same for a DIRECTED adjacency (the successor / predecessor lists of a component's DiGraph: on the critical path of the
// largest component, where the serial builder took a quarter of getDirectedGraph): every thread counts the sources of its
// stretch of the edge list, a prefix over (vertex, thread) gives every thread its first slot inside the vertex's segment (edge
// order is kept: stretches ascend with the threads), the threads drop their arcs, segments that are not ascending are sorted.
bool build_csr_directed_parallel(uint32_t n, const uint32_t *from, const uint32_t *to, size_t m, Csr &c) {
  const unsigned nt = stage_threads();
  if (std::getenv("MSGPU_GRAPH_SERIAL_CSR")) return false; // measurement switch
  if (nt < 2 || m < par_min() || static_cast<size_t>(nt) * (static_cast<size_t>(n) + 1) > (size_t(1) << 26)) return false;
  const size_t                stride = static_cast<size_t>(n) + 1;
  std::unique_ptr<uint32_t[]> hist(new uint32_t[static_cast<size_t>(nt) * stride]); // [thread][vertex]; zeroed by its threads
  auto chunk = [&](unsigned t) { return std::make_pair(m * t / nt, m * (t + 1) / nt); };
  auto on_threads = [&](auto &&body) { StagePool::get().run(nt, nt, [&](size_t t) { body(static_cast<unsigned>(t)); }); };
  on_threads([&](unsigned t) {
    uint32_t *h = hist.get() + static_cast<size_t>(t) * stride;
    std::fill(h, h + stride, 0u);
    for (size_t i = chunk(t).first; i < chunk(t).second; ++i) ++h[from[i]];
  });
  c.off.assign(stride, 0);
  on_threads([&](unsigned t) {
    const size_t v0 = stride * t / nt, v1 = stride * (t + 1) / nt;
    for (size_t v = v0; v < v1; ++v) {
      uint32_t run = 0;
      for (unsigned k = 0; k < nt; ++k) {
        uint32_t      &h   = hist[static_cast<size_t>(k) * stride + v];
        const uint32_t cnt = h;
        h                  = run;
        run += cnt;
      }
      if (v < n) c.off[v + 1] = run;
    }
  });
  for (uint32_t v = 0; v < n; ++v) c.off[v + 1] += c.off[v];
  c.arcs.resize(c.off[n]);
  on_threads([&](unsigned t) {
    uint32_t *h = hist.get() + static_cast<size_t>(t) * stride;
    for (size_t i = chunk(t).first; i < chunk(t).second; ++i) c.arcs[c.off[from[i]] + h[from[i]]++] = Arc{to[i], static_cast<uint32_t>(i)};
  });
  on_threads([&](unsigned t) {
    for (size_t v = static_cast<size_t>(n) * t / nt; v < static_cast<size_t>(n) * (t + 1) / nt; ++v) {
      if (c.off[v + 1] - c.off[v] < 2) continue;
      Arc *b = c.arcs.data() + c.off[v], *e = c.arcs.data() + c.off[v + 1];
      bool sorted = true;
      for (Arc *p = b; p + 1 < e && sorted; ++p) sorted = p->to <= (p + 1)->to;
      if (!sorted) std::stable_sort(b, e, [](const Arc &x, const Arc &y) { return x.to < y.to; });
    }
  });
  return true;
}

Csr build_csr(uint32_t n, const uint32_t *from, const uint32_t *to, size_t m, bool both_ways) {
  Csr c;
  if (both_ways && build_csr_undirected_parallel(n, from, to, m, c)) return c;
  if (!both_ways && build_csr_directed_parallel(n, from, to, m, c)) return c;
  c.off.assign(static_cast<size_t>(n) + 1, 0);
  for (size_t i = 0; i < m; ++i) {
    ++c.off[from[i] + 1];
    if (both_ways) ++c.off[to[i] + 1];
  }
  for (uint32_t v = 0; v < n; ++v) c.off[v + 1] += c.off[v];
  c.arcs.resize(c.off[n]);
  std::vector<uint32_t> cur(c.off.begin(), c.off.end() - 1);
  if (both_ways) { // lower neighbours first: with a (v1, v2)-sorted table every segment comes out ascending
    for (size_t i = 0; i < m; ++i) {
      if (from[i] > to[i]) c.arcs[cur[from[i]]++] = Arc{to[i], static_cast<uint32_t>(i)};
      else c.arcs[cur[to[i]]++] = Arc{from[i], static_cast<uint32_t>(i)};
    }
  }
  for (size_t i = 0; i < m; ++i) {
    if (!both_ways) c.arcs[cur[from[i]]++] = Arc{to[i], static_cast<uint32_t>(i)};
    else if (from[i] > to[i]) c.arcs[cur[to[i]]++] = Arc{from[i], static_cast<uint32_t>(i)};
    else c.arcs[cur[from[i]]++] = Arc{to[i], static_cast<uint32_t>(i)};
  }
  for (uint32_t v = 0; v < n; ++v) {
    if (c.off[v + 1] - c.off[v] < 2) continue;
    Arc *b = c.arcs.data() + c.off[v], *e = c.arcs.data() + c.off[v + 1];
    bool sorted = true;
    for (Arc *p = b; p + 1 < e && sorted; ++p) sorted = p->to <= (p + 1)->to;
    if (!sorted) std::stable_sort(b, e, [](const Arc &x, const Arc &y) { return x.to < y.to; });
  }
  return c;
}

// mst.cpp:35-73, including unify()'s use of the weights of the vertices rather than of their roots
struct UnionFind { // vertex ids are dense: vectors instead of the reference's two hash maps; weight 0 = "not seen yet"
  std::vector<uint32_t> parent;
  std::vector<uint64_t> weight;
  explicit UnionFind(uint32_t n) : parent(n), weight(n, 0) {}
  uint32_t find(uint32_t v) { // mst.cpp:40-60: walk to the vertex that is its own parent, then hang the whole path on it
    if (!weight[v]) {
      parent[v] = v;
      weight[v] = 1;
      return v;
    }
    uint32_t root = v;
    while (parent[root] != root) root = parent[root];
    for (uint32_t a = v; a != root;) {
      const uint32_t next = parent[a];
      parent[a]           = root;
      a                   = next;
    }
    return root;
  }
  // unify (mst.cpp:62-73) with the two roots its own find() calls return (the caller has just found them: a second
  // find() of a compressed path changes nothing)
  void unify(uint32_t v1, uint32_t v2, uint32_t first, uint32_t second) {
    if (weight[v2] > weight[v1]) std::swap(first, second);
    weight[first] += weight[second];
    parent[second] = first;
  }
};

// getMaxSpanTree (mst.cpp:75-111): Kruskal over the edges that have a consensus direction, heaviest first, ties in edge
// order (std::sort on a vector built in edge order; canonical = stable).  `cand` = candidate edge indices in edge order;
// ends(e) -> (a, b), weight(e).
template <class Ends, class Weight>
void max_span_tree(uint32_t nv, Ends ends, Weight weight, const std::vector<uint32_t> &cand, std::vector<uint8_t> &in_tree,
                   std::vector<uint32_t> *set_of = nullptr /*large graphs: per vertex, a vertex of its connected set*/) {
  // (weight, position): one sort key, no gathers in the compare.  Edges of weight 0 (every shadow edge: 97 % of the edges
  // of BASELINE.json configs[2]) come last whatever their number and keep their edge order among themselves: they are
  // not sorted at all.
  Tick                                                    tick;
  const size_t                                            nc = cand.size();
  std::vector<std::vector<std::pair<uint64_t, uint32_t>>> keyed_of(stage_threads() + 1);
  RawVec<uint8_t>                                         zero(nc);
  parallel_chunks(nc, [&](unsigned chunk, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const uint64_t w = weight(cand[i]);
      zero[i]          = w == 0;
      if (w) keyed_of[chunk].emplace_back(w, static_cast<uint32_t>(i));
    }
  });
  std::vector<std::pair<uint64_t, uint32_t>> keyed;
  for (auto &k : keyed_of) keyed.insert(keyed.end(), k.begin(), k.end());
  // (the weighted edges of configs[2] are 33 k: two milliseconds on one thread, and this function is on the stage's critical path)
  parallel_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, uint32_t> &x, const std::pair<uint64_t, uint32_t> &y) {
    return x.first != y.first ? x.first > y.first : x.second < y.second;
  }, nc >= par_min() ? size_t(4096) : size_t(0));
  tick("  mst: keys + sort");
  // Whether an edge joins the tree depends on one thing only: are its ends connected by the edges taken before it (which root
  // a set hangs on -- mst.cpp:62-73 goes by the weights of the two vertices, not of their roots -- decides nothing).
  if (nc < par_min() || stage_threads() < 2) {
    UnionFind uf(nv);
    auto      take = [&](uint32_t pos) {
      const uint32_t e  = cand[pos];
      const auto     ab = ends(e);
      const uint32_t ra = uf.find(ab.first), rb = uf.find(ab.second);
      if (ra != rb) {
        in_tree[e] = 1;
        uf.unify(ab.first, ab.second, ra, rb);
      }
    };
    for (auto &k : keyed) take(k.second);
    for (size_t i = 0; i < nc; ++i)
      if (zero[i]) take(static_cast<uint32_t>(i));
    return;
  }
  // A large graph.  Kruskal's order is a total order here (weight, then position in `cand`), so the forest it builds is THE
  // minimum spanning forest under that order, and any way of finding it finds the same edges.  The way that runs on all host
  // threads (Boruvka's): every set of vertices connected so far takes the FIRST edge of the order that leaves it -- by the
  // cut property an edge of the forest --, the sets are merged along those edges, and again, until no edge leaves a set:
  // at most log2(vertices) rounds of one pass over the edges that still join two sets.  (Kruskal's loop took a million turns
  // of two find() calls on one thread: two thirds of the clean-up's span-tree time on BASELINE configs[2].)
  require(nc < (size_t(1) << 31), "getMaxSpanTree: too many edges");
  const uint32_t   nk = static_cast<uint32_t>(keyed.size());
  RawVec<uint32_t> rank(nc), ea(nc), eb(nc); // place in Kruskal's order; the ends
  RawVec<uint8_t>  dead(nc);
  parallel_chunks(nc, [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const auto ab = ends(cand[i]);
      ea[i]         = ab.first;
      eb[i]         = ab.second;
      dead[i]       = 0;
      if (zero[i]) rank[i] = nk + static_cast<uint32_t>(i); // behind every weighted edge, in edge order
    }
  });
  parallel_chunks(nk, [&](unsigned, size_t b, size_t e) {
    for (size_t j = b; j < e; ++j) rank[keyed[j].second] = static_cast<uint32_t>(j);
  });
  constexpr uint64_t NONE = ~uint64_t(0);
  RawVec<uint32_t>   comp(nv), hook(nv);
  RawVec<uint64_t>   best(nv); // per set (at its representative): rank << 32 | position of the first edge that leaves it
  parallel_chunks(nv, [&](unsigned, size_t b, size_t e) {
    for (size_t v = b; v < e; ++v) comp[v] = static_cast<uint32_t>(v);
  });
  parallel_chunks(nv, [&](unsigned, size_t b, size_t e) {
    for (size_t v = b; v < e; ++v) {
      best[v] = NONE;
      hook[v] = static_cast<uint32_t>(v);
    }
  });
  RawVec<uint32_t> next_comp(nv);
  unsigned         n_rounds = 0;
  for (;;) {
    std::atomic<int> any{0};
    const bool       first_round = n_rounds == 0; // every vertex is its own set: nothing to look up
    parallel_dynamic(nc, 8192, [&](size_t b, size_t e) {
      bool mine = false;
      for (size_t i = b; i < e; ++i) {
        if (dead[i]) continue;
        const uint32_t ca = first_round ? ea[i] : comp[ea[i]], cb = first_round ? eb[i] : comp[eb[i]];
        if (ca == cb) { // inside a set from now on (also an edge from a vertex to itself)
          dead[i] = 1;
          continue;
        }
        mine                = true;
        const uint64_t key = (static_cast<uint64_t>(rank[i]) << 32) | static_cast<uint64_t>(i);
        for (const uint32_t c : {ca, cb}) {
          uint64_t seen = __atomic_load_n(&best[c], __ATOMIC_RELAXED);
          while (key < seen && !__atomic_compare_exchange_n(&best[c], &seen, key, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
          }
        }
      }
      if (mine) any.store(1, std::memory_order_relaxed);
    });
    if (!any.load()) break;
    ++n_rounds;
    // every set with an edge out takes it and hangs itself on the set at the other end; two sets that chose the same edge: the
    // one with the higher representative hangs on the other (the chosen edges have no other cycle: ranks are unique)
    parallel_chunks(nv, [&](unsigned, size_t b, size_t e) {
      for (size_t c = b; c < e; ++c) {
        if (best[c] == NONE) continue;
        const uint32_t i = static_cast<uint32_t>(best[c]);
        __atomic_store_n(&in_tree[cand[i]], uint8_t(1), __ATOMIC_RELAXED); // (both ends may choose the edge)
        const uint32_t ca = comp[ea[i]], cb = comp[eb[i]], other = ca == c ? cb : ca;
        if (static_cast<uint32_t>(best[other]) == i && best[other] != NONE && c < other) continue;
        hook[c] = other;
      }
    });
    parallel_chunks(nv, [&](unsigned, size_t b, size_t e) { // every vertex on the representative of its merged set
      for (size_t v = b; v < e; ++v) {
        uint32_t r = comp[v];
        while (hook[r] != r) r = hook[r];
        next_comp[v] = r; // (comp and hook are still being read by the other threads)
      }
    });
    parallel_chunks(nv, [&](unsigned, size_t b, size_t e) { // ... and the next round's clean slate
      for (size_t v = b; v < e; ++v) {
        comp[v] = next_comp[v];
        best[v] = NONE;
        hook[v] = static_cast<uint32_t>(v);
      }
    });
  }
  if (set_of) set_of->assign(comp.begin(), comp.end());
  if (std::getenv("MSGPU_GRAPH_DEBUG")) fprintf(stderr, "[graph]   mst: %u rounds (configs[2]: 10; a million edges leave the sets in the first four)\n", n_rounds);
  tick("  mst: Boruvka rounds");
}

// GraphUtil::getShortestPath (Graph.h:927-978): Dijkstra with unit weights whose queue is ordered by (distance,
// insertion counter) -- i.e. breadth-first search, a vertex keeping the path of whoever reached it first; neighbours
// ascending id (for a DiGraph: successors).  Empty result = unreachable.
std::vector<uint32_t> shortest_path(const Csr &adj, const uint8_t *arc_ok_by_edge, uint32_t n, uint32_t src, uint32_t dst) {
  std::vector<uint32_t> from(n, NIL), queue{src}, p;
  std::vector<uint8_t>  seen(n, 0);
  seen[src] = 1;
  bool found = src == dst;
  for (size_t h = 0; h < queue.size() && !found; ++h) {
    const uint32_t v = queue[h];
    for (const Arc *t = adj.begin(v); t != adj.end(v); ++t) {
      if ((arc_ok_by_edge && !arc_ok_by_edge[t->e]) || seen[t->to]) continue;
      seen[t->to] = 1;
      from[t->to] = v;
      if (t->to == dst) {
        found = true;
        break;
      }
      queue.push_back(t->to);
    }
  }
  if (!found) return p;
  for (uint32_t v = dst;; v = from[v]) {
    p.push_back(v);
    if (v == src) break;
  }
  std::reverse(p.begin(), p.end());
  return p;
}

// getConnectedComponents (cc.cpp:33-70): breadth-first over the edges that carry a consensus direction; vertices in
// ascending id, neighbours ascending id.  v_ok(v): vertex in the graph; e_ok(e): edge in the graph and with a consensus
// direction, asked per arc (const Arc *); set_comp(v, c) / get_comp(v): component number (NIL at the start).
template <class VOk, class EOk, class GetComp, class SetComp>
std::vector<std::vector<uint32_t>> connected_components(uint32_t nv, const Csr &adj, VOk v_ok, EOk e_ok, GetComp get_comp,
                                                        SetComp set_comp) {
  std::vector<std::vector<uint32_t>> comps;
  for (uint32_t s = 0; s < nv; ++s) {
    if (!v_ok(s) || get_comp(s) != NIL) continue;
    const uint32_t        cid = static_cast<uint32_t>(comps.size());
    std::vector<uint32_t> members{s};
    set_comp(s, cid);
    for (size_t h = 0; h < members.size(); ++h) {
      const uint32_t cur = members[h];
      for (const Arc *t = adj.begin(cur); t != adj.end(cur); ++t)
        if (get_comp(t->to) == NIL && e_ok(t)) {
          set_comp(t->to, cid);
          members.push_back(t->to);
        }
    }
    comps.push_back(std::move(members));
  }
  return comps;
}

// The same components for a large graph, on the stage's threads: lock-free union-find over the arcs e_ok accepts (the
// larger root goes under the smaller, so a tree's root is its lowest vertex), then the components are numbered in the order
// the scan above meets them -- by their first vertex v_ok accepts -- and every vertex whose tree has a number joins it,
// vertices ascending (the scan's members come in search order; nothing downstream reads that order).
template <class VOk, class EOk, class SetComp>
std::vector<std::vector<uint32_t>> connected_components_parallel(uint32_t nv, const Csr &adj, VOk v_ok, EOk e_ok, SetComp set_comp) {
  std::unique_ptr<std::atomic<uint32_t>[]> parent(new std::atomic<uint32_t>[nv]);
  parallel_chunks(nv, [&](unsigned, size_t b, size_t e) {
    for (size_t v = b; v < e; ++v) parent[v].store(static_cast<uint32_t>(v), std::memory_order_relaxed);
  });
  auto find = [&](uint32_t x) {
    for (;;) {
      uint32_t p = parent[x].load(std::memory_order_relaxed);
      if (p == x) return x;
      const uint32_t gp = parent[p].load(std::memory_order_relaxed);
      if (gp != p) parent[x].compare_exchange_weak(p, gp, std::memory_order_relaxed); // path halving (a lost race changes nothing)
      x = gp;
    }
  };
  parallel_dynamic(nv, 1024, [&](size_t b, size_t e) {
    for (size_t v = b; v < e; ++v)
      for (const Arc *t = adj.begin(static_cast<uint32_t>(v)); t != adj.end(static_cast<uint32_t>(v)); ++t) {
        if (t->to <= v || !e_ok(t)) continue; // (every undirected edge has an arc at both ends)
        uint32_t x = static_cast<uint32_t>(v), y = t->to;
        for (;;) {
          x = find(x);
          y = find(y);
          if (x == y) break;
          if (x < y) std::swap(x, y); // x: the larger root
          uint32_t expect = x;
          if (parent[x].compare_exchange_strong(expect, y, std::memory_order_relaxed)) break;
        }
      }
  });
  std::vector<uint32_t> cid(nv, NIL);
  uint32_t              n_comp = 0;
  for (uint32_t s = 0; s < nv; ++s) {
    if (!v_ok(s)) continue;
    const uint32_t r = find(s);
    if (cid[r] == NIL) cid[r] = n_comp++;
  }
  std::vector<std::vector<uint32_t>> comps(n_comp);
  for (uint32_t v = 0; v < nv; ++v) {
    const uint32_t c = cid[find(v)];
    if (c == NIL) continue;
    set_comp(v, c);
    comps[c].push_back(v);
  }
  return comps;
}

// DiGraph::sortTopologically (Graph.cpp:359-395): vertices without predecessors are collected in ascending id and
// taken from the BACK of that list; a vertex whose last predecessor has just been emitted goes on top.  Vertices on a
// cycle never appear.  `alive` (per arc's edge index, optional) and `valive` (per vertex, optional) are tombstones.
std::vector<uint32_t> sort_topologically(uint32_t n, const Csr &succ, const Csr &pred, const uint8_t *alive,
                                         const uint8_t *valive) {
  std::vector<int64_t>  deg(n, 0);
  std::vector<uint32_t> ready, result;
  for (uint32_t v = 0; v < n; ++v) {
    if (valive && !valive[v]) continue;
    for (const Arc *t = pred.begin(v); t != pred.end(v); ++t) deg[v] += !alive || alive[t->e];
    if (deg[v] == 0) ready.push_back(v);
  }
  result.reserve(n);
  while (!ready.empty()) {
    const uint32_t v = ready.back();
    ready.pop_back();
    for (const Arc *t = succ.begin(v); t != succ.end(v); ++t)
      if ((!alive || alive[t->e]) && --deg[t->to] == 0) ready.push_back(t->to);
    result.push_back(v);
  }
  return result;
}

} // namespace

// An array whose elements are constructed by whoever fills it (on the threads that fill it): a std::vector would
// construct -- and page in -- tens of megabytes on the calling thread first.
template <class T> struct RawArray {
  T     *p = nullptr;
  size_t n = 0;
  RawArray() = default;
  RawArray(const RawArray &) = delete;
  RawArray &operator=(const RawArray &) = delete;
  ~RawArray() { std::free(p); } // (T is trivially destructible)
  void allocate(size_t k) {
    static_assert(std::is_trivially_destructible<T>::value, "RawArray elements are never destroyed");
    std::free(p);
    p = static_cast<T *>(std::malloc((k ? k : 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    n = k;
  }
  T       &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t   size() const { return n; }
  T       *begin() { return p; }
  T       *end() { return p + n; }
};

struct msgpu_graph {
  // tables: the caller's (msgpu_graph_create_borrowed: valid until msgpu_graph_free) or copies held in own_* (msgpu_graph_create)
  std::vector<msgpu_edge>      own_edges;
  std::vector<msgpu_edgematch> own_ems;
  std::vector<msgpu_order>     own_orders;
  std::vector<uint32_t>        own_ids;
  bool                         ems_on_demand = false; // no EdgeMatch table: msgpu_graph_set_path_edgematches supplies the path edges'
  bool                         have_path_ems = false;
  std::vector<uint32_t>        path_edges;            // table index of the edge under every path step, paths concatenated
  const msgpu_edge      *t_edges = nullptr;
  const msgpu_edgematch *t_ems   = nullptr;
  const msgpu_order     *t_orders = nullptr;
  const uint32_t        *t_ids   = nullptr;
  uint64_t               n_edges = 0, n_ems = 0, n_orders = 0, n_ids = 0;
  uint32_t               nv = 0;
  // vertices and undirected edges (table order): one record each, so that a visit costs one cache line
  struct Vertex {
    int32_t  len   = 0;
    uint32_t meta0 = 0;
    uint32_t comp  = NIL; // connected component
    uint32_t loc   = NIL; // local id inside its component's DiGraph
    uint32_t seq   = NIL; // number of its first pop in getDirectedGraph's walk (NIL: never popped)
    int8_t   dir   = D_NONE;
    uint8_t  alive = 1, in_dg = 0;
  };
  struct Edge {
    uint32_t a = 0, b = 0;
    uint32_t ord_lo = 0;       // its orders: order table [ord_lo, ord_lo + ord_cnt), those with o_kept set
    uint32_t first  = NIL;     // first kept order, NIL = none
    uint32_t de_fwd = 0, de_bwd = 0; // its directed edges a -> b / b -> a in its component's DiGraph (+1; 0 = none)
    uint64_t weight = 0;
    uint16_t ord_cnt = 0;
    int8_t   cons    = D_NONE;
    uint8_t  alive = 1, shadow = 0;
  };
  std::vector<Vertex>  V;
  RawArray<Edge>       E; // (constructed in graph_create's parallel fill)
  std::vector<uint8_t> o_kept;  // per order: still on its edge (findDeletableEdges drops the contained ones)
  std::vector<uint32_t> walk_seq; // per vertex: its pop number in getDirectedGraph's walk (compact copy of Vertex::seq for the walk itself)
  std::vector<uint32_t> span_set; // per vertex, a vertex of its set in the span forest (large graphs; empty: not known) -- the
                                  // connected components of msgpu_graph_linearize when decycle() removed no edge
  std::vector<uint8_t> o_fwd;   // per kept order: it runs E.a -> E.b in its component's DiGraph (getDirectedGraph, pass 1)
  struct OLite { // what the walks over the graph read of an EdgeOrder (16 of its 64 bytes: the random accesses of
    uint32_t start, end, base, flags; // getDirectedGraph stay inside a quarter of the cache footprint)
  };
  RawArray<OLite> ol;
  Csr                  adj;
  // ContainElements (src/main.cpp:509-531) by host read: those of read v are contain[contain_first[v] .. contain_first[v + 1]),
  // in contraction order; their anchors lie in contain_anchor (one flat table instead of a map of vectors of vectors)
  struct Contain {
    uint32_t nano, direction;
    uint64_t anchors_off;
    uint32_t anchors_cnt;
  };
  std::vector<uint64_t> contain_first; // nv + 1 entries (empty before the clean-up)
  RawArray<Contain>     contain;
  std::vector<uint32_t> contain_anchor;
  bool cleaned = false, linearized = false;
  bool every_alive_edge_kept = false; // set with the arc flags in msgpu_graph_linearize
  uint32_t n_threads = 1;
  msgpu_graph_stats stats{};
  // paths + storage the msgpu_path_input views point into
  struct PathStore {
    std::vector<msgpu_path_read>    reads;
    std::vector<uint32_t>           order_off, em_off, contain_anchors, step_edge; // step_edge: edge-table index per step
    std::vector<msgpu_path_order>   orders;
    std::vector<msgpu_path_em>      ems;
    std::vector<msgpu_path_contain> contains;
  };
  std::vector<PathStore> paths;
  char err[256] = {0};

  bool odir(uint32_t o) const { return (ol[o].flags & MSGPU_ORD_DIR) != 0; }
  bool ocont(uint32_t o) const { return (ol[o].flags & MSGPU_ORD_CONTAINED) != 0; }
  int64_t edge_between(uint32_t a, uint32_t b) const {
    const Arc *t = adj.find(a, b);
    return t && E[t->e].alive ? static_cast<int64_t>(t->e) : -1;
  }
  void delete_edge(uint32_t e) { E[e].alive = 0; }             // Graph::deleteEdge
  void delete_vertex(uint32_t v) {                              // Graph::deleteVertex
    bury_vertex(adj, nullptr, v, [&](uint32_t e) { E[e].alive = 0; });
    V[v].alive = 0;
  }
};

namespace {

// ---- the DiGraph of one connected component (dg.cpp), flat ---------------------------------------------------------------

struct DiG {
  uint32_t              n = 0;
  std::vector<uint32_t> ids;          // local -> global vertex id, ascending
  RawVec<uint32_t>      ea, eb, src;  // directed edges in creation order: local endpoints, undirected edge of origin
  RawVec<uint8_t>       shadow;       // shared by the graph and its cycle-free copy (Graph.h:773-774: shallow copy)
  RawVec<uint64_t>      weight;
  std::vector<uint32_t> ord_off;      // EdgeOrders per directed edge, in appendOrder order
  RawVec<uint32_t>      ord;
  Csr                   succ, pred;
  Csr                   lsucc, lpred; // the arcs of the cycle-free copy: the edges that were no shadow edges when extract_paths began
  size_t m() const { return ea.size(); }
  int64_t get_edge(uint32_t a, uint32_t b) const {
    const Arc *t = succ.find(a, b);
    return t ? static_cast<int64_t>(t->e) : -1;
  }
};

// getDirectedGraph, dg.cpp:35-121; the component = the vertices with comp_of == cid (its sub-graph keeps every alive
// edge between those vertices, Graph.cpp:317-326).
//
// The reference walks the component with a stack and, for every edge it meets for the first time, turns the edge's
// EdgeOrders into directed edges.  Which endpoint meets an edge first, and with which toggle, is the only thing that
// depends on the walk: an edge with at least one kept order gets its directed edges when the first of its endpoints is
// popped for the first time (all of that vertex's edges are handled in that one pop; hasEdge() then makes every later
// visit skip it, including the visit's push), and the toggle of a first pop is the direction the vertex receives there.
// So the walk below only settles pop order, directions and reachability -- per arc it reads two flag bytes and the
// neighbour's 24-byte record, no edge record and no order -- and the directed edges are produced afterwards, on all host
// threads, in exactly the creation order of the reference: by pop number of the processing vertex, its arcs in
// ascending neighbour id, an edge's orders in table order, a direction's edge on its first order.
DiG get_directed_graph(msgpu_graph &g, uint32_t cid, const std::vector<uint32_t> &members, uint32_t start,
                       const RawVec<uint8_t> &arc_flags) {
  constexpr uint8_t AF_ALIVE = 1, AF_KEPT = 2, AF_CONS = 4, AF_POS = 8; // per arc: edge alive / has a kept order / consensus set / e_POS
  DiG                                    dg;
  Tick                                   tk;
  std::vector<uint32_t>                  pop_order;                 // vertices in the order of their FIRST pop
  std::vector<std::pair<uint32_t, bool>> stack{{start, true}};
  const Arc *const                       arcs0 = g.adj.arcs.data();
  if (g.every_alive_edge_kept) {
    // The reference's loop (kept below for graphs with an alive edge without a kept order) pushes a vertex once for every
    // neighbour that meets it before its first pop -- about ten entries per vertex on configs[2], nine of which are popped to no
    // effect.  With every alive edge kept, a pop other than the first does nothing, and the first pop of a vertex is the entry
    // pushed LAST for it: the one of the most recently popped neighbour -- its parent in a depth-first search that takes a
    // vertex's arcs from the last to the first and descends into a neighbour that has not been popped yet when its turn comes
    // (an entry pushed by an earlier neighbour lies deeper in the stack and finds the vertex popped).  So: that search, one
    // frame per vertex, with the toggle the parent's entry would have carried.  Every arc of a popped vertex is still looked
    // at once, so the in_dg marks end up the same.
    // What the search reads per arc: the flag byte beside the arc and ONE word of a table of pop numbers (400 KB for the
    // 100 k reads of configs[2]: it stays in the core's cache; the vertex records, 24 bytes each, do not).  A neighbour over an
    // alive edge with a consensus direction is a vertex of this component by the component's definition, so no record has
    // to say so.  Directions, pop numbers and marks go into the records afterwards, on all threads.
    struct Frame {
      uint32_t v, next; // next: one past the arc to look at next (arcs are taken in descending order)
      bool     toggle;
    };
    std::vector<Frame>   frames;
    std::vector<uint8_t> pop_toggle;
    uint32_t *const      seq = g.walk_seq.data(); // NIL for every vertex when msgpu_graph_linearize hands out the components
    const uint32_t *const off = g.adj.off.data();
    auto                 pop_first = [&](uint32_t v, bool toggle) {
      seq[v] = static_cast<uint32_t>(pop_order.size());
      pop_order.push_back(v);
      pop_toggle.push_back(toggle ? 1 : 0);
      frames.push_back(Frame{v, off[v + 1], toggle}); // (asking for the neighbours' arc rows ahead changed nothing: tools/experiments/graph_walk_ab.sh)
    };
    require(g.walk_seq.size() == g.V.size(), "getDirectedGraph: pop-number table not allocated");
    require(g.V[start].seq == NIL && g.V[start].dir == D_NONE && seq[start] == NIL, "getDirectedGraph: start vertex already directed");
    pop_order.reserve(members.size());
    pop_toggle.reserve(members.size());
    pop_first(start, true);
    while (!frames.empty()) {
      const uint32_t v = frames.back().v, first_arc = off[v];
      const bool     toggle = frames.back().toggle;
      uint32_t       q = frames.back().next;
      bool           descended = false;
      while (q > first_arc) {
        --q;
        const uint8_t af = arc_flags[q];
        if ((af & (AF_ALIVE | AF_CONS)) != (AF_ALIVE | AF_CONS)) continue; // (no consensus direction: not pushed)
        const uint32_t to = arcs0[q].to;
        if (seq[to] != NIL) continue; // popped already
        frames.back().next = q;
        pop_first(to, toggle == ((af & AF_POS) != 0)); // (frames may move: nothing of the old frame is used below)
        descended = true;
        break;
      }
      if (!descended) frames.pop_back();
    }
    const size_t n_popped = pop_order.size();
    parallel_chunks(n_popped, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        msgpu_graph::Vertex &vc = g.V[pop_order[i]];
        if (vc.comp != cid) throw GraphError("getDirectedGraph: the walk left its component");
        vc.seq   = static_cast<uint32_t>(i);
        vc.dir   = pop_toggle[i] ? D_POS : D_NEG; // (D_NONE until here: nothing else sets a direction)
        vc.in_dg = 1;
      }
    });
    // the reference marks every neighbour a popped vertex has over an alive edge inside the component; the component is
    // connected over such edges (cc.cpp:33-70 built it from them), so those are its vertices, all popped above.  Should a
    // vertex of it be left (it cannot), the marks are made the long way.
    if (n_popped != members.size())
      for (uint32_t v : pop_order)
        for (uint32_t q = off[v]; q < off[v + 1]; ++q)
          if ((arc_flags[q] & AF_ALIVE) && g.V[arcs0[q].to].comp == cid) g.V[arcs0[q].to].in_dg = 1;
    stack.clear();
  }
  while (!stack.empty()) {
    const uint32_t cur    = stack.back().first;
    const bool     toggle = stack.back().second;
    stack.pop_back();
    msgpu_graph::Vertex &vc    = g.V[cur];
    const bool           first = vc.seq == NIL;
    if (first) {
      vc.seq = static_cast<uint32_t>(pop_order.size());
      pop_order.push_back(cur);
    }
    vc.in_dg = 1;
    if (vc.dir == D_NONE) vc.dir = toggle ? D_POS : D_NEG;
    // A vertex is pushed once by every neighbour that sees it before its first pop, and every later pop walks its arcs
    // again -- to no effect when each alive edge has a kept order (always, after the clean-up): hasEdge() skips them all,
    // and the neighbours it would mark are marked since the first pop.  (About ten pops per vertex on configs[2].)
    if (!first && g.every_alive_edge_kept) continue;
    for (const Arc *n = g.adj.begin(cur); n != g.adj.end(cur); ++n) __builtin_prefetch(&g.V[n->to]);
    for (const Arc *n = g.adj.begin(cur); n != g.adj.end(cur); ++n) {
      const uint8_t af = arc_flags[n - arcs0];
      if (!(af & AF_ALIVE)) continue;
      msgpu_graph::Vertex &vn = g.V[n->to];
      if (vn.comp != cid) continue;
      const bool other_exists = vn.in_dg != 0 && vn.dir != D_NONE;
      if (!other_exists) vn.in_dg = 1;
      // hasEdge(a, b) || hasEdge(b, a): the edge got its directed edges at an earlier first pop of either end
      if ((af & AF_KEPT) && (!first || vn.seq != NIL)) continue;
      if (!(af & AF_CONS)) continue;
      const bool nxt = toggle == ((af & AF_POS) != 0);
      if (!other_exists) stack.emplace_back(n->to, nxt);
    }
  }
  tk("  dg: walk");
  // the directed edges: vertex i of the pop order handles its edges to the vertices popped after it (or never)
  const size_t          np = pop_order.size();
  std::vector<uint64_t> slot_base(np + 1, 0), ord_base(np + 1, 0);
  auto for_edges_of = [&](size_t i, auto &&body) { // body(undirected edge, neighbour, toggle of the processing vertex)
    const uint32_t v      = pop_order[i];
    const bool     toggle = g.V[v].dir == D_POS;
    // (the edge records and the neighbours' records are the random reads of both passes: asked for ahead of the loop that
    // waits for them -- a vertex has tens of arcs)
    for (const Arc *n = g.adj.begin(v); n != g.adj.end(v); ++n) {
      __builtin_prefetch(&g.E[n->e]);
      __builtin_prefetch(&g.V[n->to]);
    }
    for (const Arc *n = g.adj.begin(v); n != g.adj.end(v); ++n) __builtin_prefetch(&g.ol[g.E[n->e].ord_lo]); // (its first order: edges have one or two)
    for (const Arc *n = g.adj.begin(v); n != g.adj.end(v); ++n) {
      const uint8_t af = arc_flags[n - arcs0];
      if ((af & (AF_ALIVE | AF_KEPT)) != (AF_ALIVE | AF_KEPT)) continue;
      const msgpu_graph::Vertex &vn = g.V[n->to];
      if (vn.comp != cid || (vn.seq != NIL && vn.seq < i)) continue;
      body(n->e, n->to, toggle);
    }
  };
  auto direction_of = [&](uint32_t ue, uint32_t oi, uint32_t nb, bool toggle) -> bool { // true: the order runs E.a -> E.b
    const msgpu_graph::OLite &o = g.ol[oi];
    bool                      flip = false;
    if (!g.odir(oi) && o.base == nb) flip = !flip;
    if (!toggle) flip = !flip;
    const uint32_t s = flip ? o.end : o.start, t = flip ? o.start : o.end;
    if (!((s == g.E[ue].a && t == g.E[ue].b) || (s == g.E[ue].b && t == g.E[ue].a)))
      throw GraphError("getDirectedGraph: order between vertices outside the component");
    return s == g.E[ue].a;
  };
  // (a vertex popped early meets most of its neighbours unpopped, a late one hardly any: pieces by counter, not by range)
  // (vertices, each with tens of arcs: pieces of 2048; the tests' small graphs get pieces that still cut them)
  const size_t grain = par_min() >= (size_t(1) << 16) ? 2048 : std::max<size_t>(16, par_min() / 32);
  // the direction of every kept order, found once (pass 1) and read back by pass 2: an order belongs to one edge, an edge is
  // handled by one vertex, so no two threads write the same byte (pass 2 used to find each direction twice more: two more
  // random reads of the order and the edge record per order)
  require(g.o_fwd.size() >= g.o_kept.size(), "getDirectedGraph: direction table not allocated");
  uint8_t *const o_fwd = g.o_fwd.data();
  parallel_dynamic(np, grain, [&](size_t b, size_t e) { // pass 1: directed edges and orders per processing vertex
    for (size_t i = b; i < e; ++i) {
      uint64_t ns = 0, no = 0;
      for_edges_of(i, [&](uint32_t ue, uint32_t nb, bool toggle) {
        bool           fwd = false, bwd = false;
        const uint32_t lo = g.E[ue].ord_lo, hi = lo + g.E[ue].ord_cnt;
        for (uint32_t oi = lo; oi < hi; ++oi) {
          if (!g.o_kept[oi]) continue;
          const bool f = direction_of(ue, oi, nb, toggle);
          o_fwd[oi]    = f ? 1 : 0;
          (f ? fwd : bwd) = true;
          ++no;
        }
        ns += (fwd ? 1 : 0) + (bwd ? 1 : 0);
      });
      slot_base[i + 1] = ns;
      ord_base[i + 1]  = no;
    }
  });
  for (size_t i = 0; i < np; ++i) {
    slot_base[i + 1] += slot_base[i];
    ord_base[i + 1] += ord_base[i];
  }
  const size_t          m = slot_base[np];
  RawVec<uint32_t> ea(m), eb(m); // global endpoints
  dg.src.resize(m);
  dg.shadow.resize(m);
  dg.weight.resize(m);
  dg.ord_off.assign(m + 1, 0);
  dg.ord.resize(ord_base[np]);
  parallel_dynamic(np, grain, [&](size_t b, size_t e) { // pass 2: fill, every vertex into its own stretch
    for (size_t i = b; i < e; ++i) {
      uint64_t slot = slot_base[i], op = ord_base[i];
      for_edges_of(i, [&](uint32_t ue, uint32_t nb, bool toggle) {
        msgpu_graph::Edge &E = g.E[ue];
        const uint32_t     lo = E.ord_lo, hi = lo + E.ord_cnt;
        int64_t            s_fwd = -1, s_bwd = -1; // the two possible directed edges, numbered in order of first use
        uint32_t           n_fwd = 0, n_bwd = 0;
        (void)nb;
        (void)toggle;
        for (uint32_t oi = lo; oi < hi; ++oi) {
          if (!g.o_kept[oi]) continue;
          if (o_fwd[oi]) {
            if (s_fwd < 0) s_fwd = static_cast<int64_t>(slot++);
            ++n_fwd;
          } else {
            if (s_bwd < 0) s_bwd = static_cast<int64_t>(slot++);
            ++n_bwd;
          }
        }
        auto make = [&](int64_t sl, bool fwd, uint32_t cnt, uint64_t first_ord) {
          ea[sl]            = fwd ? E.a : E.b;
          eb[sl]            = fwd ? E.b : E.a;
          dg.src[sl]        = ue;
          dg.shadow[sl]     = E.shadow;
          dg.weight[sl]     = E.shadow ? 0 : E.weight;
          dg.ord_off[sl]    = static_cast<uint32_t>(first_ord); // (ord_off[m] is set below)
          (fwd ? E.de_fwd : E.de_bwd) = static_cast<uint32_t>(sl + 1);
          (void)cnt;
        };
        // a directed edge's orders are contiguous in dg.ord, the edges in slot order
        const bool     fwd_first = s_fwd >= 0 && (s_bwd < 0 || s_fwd < s_bwd);
        const uint64_t o_first = op, o_second = op + (fwd_first ? n_fwd : n_bwd);
        if (s_fwd >= 0) make(s_fwd, true, n_fwd, fwd_first ? o_first : o_second);
        if (s_bwd >= 0) make(s_bwd, false, n_bwd, fwd_first ? o_second : o_first);
        uint64_t pf = fwd_first ? o_first : o_second, pb = fwd_first ? o_second : o_first;
        for (uint32_t oi = lo; oi < hi; ++oi) {
          if (!g.o_kept[oi]) continue;
          if (o_fwd[oi]) dg.ord[pf++] = oi;
          else dg.ord[pb++] = oi;
        }
        op += n_fwd + n_bwd;
      });
    }
  });
  dg.ord_off[m] = static_cast<uint32_t>(ord_base[np]);
  tk("  dg: directed edges");
  // local ids in ascending global id
  for (uint32_t v : members)
    if (g.V[v].in_dg) dg.ids.push_back(v);
  std::sort(dg.ids.begin(), dg.ids.end());
  dg.n = static_cast<uint32_t>(dg.ids.size());
  parallel_chunks(dg.n, [&](unsigned, size_t b, size_t e) {
    for (size_t l = b; l < e; ++l) g.V[dg.ids[l]].loc = static_cast<uint32_t>(l);
  });
  dg.ea.resize(m);
  dg.eb.resize(m);
  parallel_chunks(m, [&](unsigned, size_t b, size_t e) { // (a million random reads of the vertex table: on all threads)
    for (size_t i = b; i < e; ++i) {
      dg.ea[i] = g.V[ea[i]].loc;
      dg.eb[i] = g.V[eb[i]].loc;
    }
  });
  tk("  dg: ids + order lists");
  // the two adjacencies, each on all the stage's threads (round 5: they were two serial builds side by side)
  dg.succ = build_csr(dg.n, dg.ea.data(), dg.eb.data(), m, false);
  dg.pred = build_csr(dg.n, dg.eb.data(), dg.ea.data(), m, false);
  tk("  dg: succ / pred CSR");
  return dg;
}

// ---- linearizeGraph, lg.cpp ------------------------------------------------------------------------------------------

// sortReductionByWeight (lg.cpp:418-520) on the cycle copy: `alive` = its edges; cut edges become shadow edges of BOTH
// graphs (shared Edge objects) and leave the copy.
void sort_reduction_by_weight(DiG &dg, std::vector<uint8_t> &alive) {
  const uint32_t        n = dg.n;
  std::vector<int64_t>  rem(n, 0);     // verticesWithNonNullInDegree[v]
  std::vector<uint8_t>  innn(n, 0), resolved(n, 0);
  std::vector<uint32_t> null;          // deque: consumed from `head`
  size_t                head = 0, nn = 0;
  for (uint32_t v = 0; v < n; ++v) {
    for (const Arc *t = dg.lpred.begin(v); t != dg.lpred.end(v); ++t) rem[v] += alive[t->e];
    if (rem[v] > 0) {
      innn[v] = 1;
      ++nn;
    } else {
      null.push_back(v);
    }
  }
  std::set<uint32_t> neighbors;
  if (nn)
    for (uint32_t v = 0; v < n; ++v)
      if (innn[v]) {
        neighbors.insert(v);
        break;
      }
  while (true) {
    while (head < null.size()) {
      const uint32_t v = null[head++];
      resolved[v] = 1;
      for (const Arc *s = dg.lsucc.begin(v); s != dg.lsucc.end(v); ++s) {
        if (!alive[s->e]) continue;
        if (!innn[s->to]) throw GraphError("sortReductionByWeight: in-degree map out of step");
        if (--rem[s->to] == 0) {
          null.push_back(s->to);
          innn[s->to] = 0;
          --nn;
          neighbors.erase(s->to);
        } else {
          neighbors.insert(s->to);
        }
      }
    }
    if (!nn) break;
    int64_t  min_edge = -1;
    uint32_t min_vertex = 0;
    uint64_t min_score = 0;
    auto     scan = [&](uint32_t cand) {
      for (const Arc *p = dg.lpred.begin(cand); p != dg.lpred.end(cand); ++p)
        if (alive[p->e] && !resolved[p->to] && (min_edge < 0 || dg.weight[p->e] < min_score)) {
          min_edge   = p->e;
          min_vertex = cand;
          min_score  = dg.weight[p->e];
        }
    };
    if (neighbors.empty()) {
      for (uint32_t v = 0; v < n; ++v)
        if (innn[v]) scan(v);
    } else {
      for (uint32_t v : neighbors) scan(v);
    }
    if (min_edge < 0) throw GraphError("sortReductionByWeight: no edge to cut");
    dg.shadow[min_edge] = 1;
    alive[min_edge]     = 0;
    if (--rem[min_vertex] == 0) {
      innn[min_vertex] = 0;
      --nn;
      null.push_back(min_vertex);
      neighbors.erase(min_vertex);
    }
  }
}

bool subset(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) { // a within b, both ascending
  return std::includes(b.begin(), b.end(), a.begin(), a.end());
}

// successors / predecessors of every vertex as ascending topological indices (the std::set<size_t> of lg.cpp:160-176)
struct TopoSets {
  std::vector<uint32_t> order, idx, soff, sidx, poff, pidx;
  TopoSets(const DiG &dg, const std::vector<uint8_t> &alive) {
    order = sort_topologically(dg.n, dg.lsucc, dg.lpred, alive.data(), nullptr);
    idx.assign(dg.n, NIL);
    for (uint32_t i = 0; i < order.size(); ++i) idx[order[i]] = i;
    auto fill = [&](const Csr &c, std::vector<uint32_t> &off, std::vector<uint32_t> &out) {
      off.assign(dg.n + 1, 0);
      for (uint32_t v = 0; v < dg.n; ++v) {
        off[v] = static_cast<uint32_t>(out.size());
        for (const Arc *t = c.begin(v); t != c.end(v); ++t)
          if (alive[t->e]) {
            if (idx[t->to] == NIL) throw GraphError("findClusterWeights: vertex missing from the topological order");
            out.push_back(idx[t->to]);
          }
        std::sort(out.begin() + off[v], out.end());
      }
      off[dg.n] = static_cast<uint32_t>(out.size());
    };
    fill(dg.lsucc, soff, sidx);
    fill(dg.lpred, poff, pidx);
  }
};

void add_path_weights(const DiG &dg, const std::vector<uint32_t> &order, const std::vector<uint32_t> &mv,
                      std::vector<uint64_t> &result) {
  size_t       c     = mv.size() - 1;
  const size_t limit = std::max<size_t>(mv.size(), 1) - 1;
  for (size_t i = 0; i < limit; ++i) {
    const int64_t e = dg.get_edge(order[mv[i]], order[mv[i + 1]]);
    require(e >= 0, "findClusterWeights: path edge missing");
    __atomic_fetch_add(&result[e], static_cast<uint64_t>(c), __ATOMIC_RELAXED); // (sums: the vertices of findClusterWeights run on several threads)
    c -= 1;
  }
}

std::vector<uint64_t> find_cluster_weights(const DiG &dg, const std::vector<uint8_t> &alive) { // lg.cpp:144-264
  const TopoSets        ts(dg, alive);
  std::vector<uint64_t> result(dg.m(), 0);
  struct Cand {
    std::vector<uint32_t> open, visited; // both ascending (a visited index is always larger than the one before it)
  };
  // every vertex of the order adds the weights of its own best paths to the edges: sums, so the vertices are handed to
  // the stage's threads in pieces (the result does not depend on who adds what when)
  const size_t grain = par_min() >= (size_t(1) << 16) ? 256 : std::max<size_t>(8, par_min() / 64);
  parallel_dynamic(ts.order.size(), grain, [&](size_t ob, size_t oe) {
    std::vector<Cand> cands, filtered;
    for (size_t oi = ob; oi < oe; ++oi) {
      const uint32_t  v  = ts.order[oi];
      const uint32_t *sb = ts.sidx.data() + ts.soff[v], *se = ts.sidx.data() + ts.soff[v + 1];
      if (sb == se) continue; // no successor: the only candidate is {v}, which adds nothing
      cands.clear();
      cands.push_back(Cand{std::vector<uint32_t>(sb, se), {ts.idx[v]}});
      for (const uint32_t *po = sb; po != se; ++po) {
        const uint32_t i_out = *po, active = ts.order[i_out];
        const uint32_t *ab = ts.sidx.data() + ts.soff[active], *ae = ts.sidx.data() + ts.soff[active + 1];
        for (uint32_t q = ts.poff[active]; q < ts.poff[active + 1]; ++q) {
          const uint32_t i_in = ts.pidx[q];
          for (size_t k = 0; k < cands.size(); ++k) // the bound grows with the emplace_back inside (:193)
            if (cands[k].visited.back() == i_in && std::binary_search(cands[k].open.begin(), cands[k].open.end(), i_out)) {
              Cand nc;
              std::set_intersection(cands[k].open.begin(), cands[k].open.end(), ab, ae, std::back_inserter(nc.open));
              nc.visited = cands[k].visited;
              nc.visited.push_back(i_out);
              cands.push_back(std::move(nc));
            }
        }
        filtered.clear();
        for (size_t a = 0; a < cands.size(); ++a) {
          bool dominated = false;
          for (size_t b = 0; b < cands.size() && !dominated; ++b)
            dominated = a != b && subset(cands[a].open, cands[b].open) && subset(cands[a].visited, cands[b].visited);
          if (!dominated) filtered.push_back(cands[a]);
        }
        cands.swap(filtered);
      }
      size_t best_len = 0;
      for (auto &c : cands) best_len = std::max(best_len, c.visited.size());
      for (auto &c : cands)
        if (c.visited.size() == best_len) add_path_weights(dg, ts.order, c.visited, result);
    }
  });
  return result;
}

std::vector<uint64_t> find_cluster_weights_heuristic(const DiG &dg, const std::vector<uint8_t> &alive) { // lg.cpp:72-141
  const TopoSets                     ts(dg, alive);
  std::vector<uint64_t>              result(dg.m(), 0);
  std::vector<uint32_t>              stamp(dg.n, 0), touched;
  std::vector<std::vector<uint32_t>> cand(dg.n); // candidates[vertex], valid when stamp == round
  uint32_t                           round = 0;
  for (uint32_t v : ts.order) {
    ++round;
    touched.assign(1, v);
    stamp[v] = round;
    cand[v].assign(1, ts.idx[v]);
    for (uint32_t q = ts.soff[v]; q < ts.soff[v + 1]; ++q) {
      const uint32_t               w = ts.order[ts.sidx[q]];
      const std::vector<uint32_t> *best = nullptr;
      for (const Arc *p = dg.lpred.begin(w); p != dg.lpred.end(w); ++p) // ascending id = the canonical order of :122
        if (alive[p->e] && stamp[p->to] == round && cand[p->to].size() > (best ? best->size() : 0)) best = &cand[p->to];
      if (stamp[w] == round) continue; // emplace: an existing entry stays
      std::vector<uint32_t> np = best ? *best : std::vector<uint32_t>();
      np.push_back(ts.idx[w]);
      cand[w]  = std::move(np);
      stamp[w] = round;
      touched.push_back(w);
    }
    std::sort(touched.begin(), touched.end()); // std::max_element over the map: first longest, ascending id
    const std::vector<uint32_t> *best = nullptr;
    for (uint32_t t : touched)
      if (!best || cand[t].size() > best->size()) best = &cand[t];
    add_path_weights(dg, ts.order, *best, result);
  }
  return result;
}

// extractPaths (lg.cpp:347-414).  The repeated findConservationPathAlt (lg.cpp:267-344) runs per weakly connected
// component of the remaining cycle-free graph; see the header of this file for why that gives the same sequence.
struct PathPeeler {
  const DiG                   &dg;
  const std::vector<uint64_t> &cw;
  std::vector<uint8_t>         alive, valive; // edges / vertices still in diGraphCycle
  size_t                       n_arcs = 0;
  // per-solve scratch (round-stamped)
  struct VState { // one record per vertex (a solve touches all of these for every vertex it meets: one cache line, not seven)
    uint64_t oscore;
    int32_t  deg, ohead;
    uint32_t ostamp, olen, seg, lab;
  };
  std::vector<VState> vs;
  uint32_t              round = 0, lab_round = 0;
  struct Node {
    uint32_t v;
    int32_t  parent;
  };
  struct Scratch { // a solve's own work space, kept from solve to solve (a peeling takes thousands of rounds of a few vertices each)
    std::vector<Node>     nodes;
    std::vector<uint32_t> order, stack, max_outs;
  };
  Scratch               scratch;     // of the calling thread's solves
  std::vector<uint32_t> members_tmp; // split()'s search
  struct Comp {
    std::vector<uint32_t> members, path; // members ascending
    uint32_t              segroot = 0;
  };
  std::vector<Comp> comps;
  struct Key {
    uint32_t size, segroot, comp;
    bool operator<(const Key &o) const { return size != o.size ? size < o.size : segroot < o.segroot; }
  };
  std::priority_queue<Key>      heap;
  std::priority_queue<uint32_t> iso; // vertices without edges: each is a root and a sink of its own

  PathPeeler(const DiG &d, const std::vector<uint8_t> &a, const std::vector<uint64_t> &w)
      : dg(d), cw(w), alive(a), valive(d.n, 1), vs(d.n, VState{0, 0, 0, 0, 0, 0, 0}) {
    for (uint8_t x : alive) n_arcs += x;
  }

  bool has_arcs(uint32_t v) const {
    for (const Arc *t = dg.lsucc.begin(v); t != dg.lsucc.end(v); ++t)
      if (alive[t->e]) return true;
    for (const Arc *t = dg.lpred.begin(v); t != dg.lpred.end(v); ++t)
      if (alive[t->e]) return true;
    return false;
  }

  // split `pool` (alive vertices, ascending) into weakly connected components; lone vertices go to `iso`.  The components
  // are independent (a solve reads and writes the scratch of its own vertices only): those of a large pool -- the first
  // split of a large component, thousands of them -- are solved on the stage's threads and enter the heap in the order a
  // single thread would have found them.
  void split(const std::vector<uint32_t> &pool) {
    ++lab_round;
    std::vector<uint32_t> &members = members_tmp;
    std::vector<Comp>      found;
    const bool            fan_out = pool.size() >= par_min() / 4 && stage_threads() > 1;
    for (uint32_t s : pool) {
      if (!valive[s] || vs[s].lab == lab_round) continue;
      members.assign(1, s);
      vs[s].lab = lab_round;
      for (size_t h = 0; h < members.size(); ++h) {
        const uint32_t v = members[h];
        for (const Csr *c : {&dg.lsucc, &dg.lpred})
          for (const Arc *t = c->begin(v); t != c->end(v); ++t)
            if (alive[t->e] && vs[t->to].lab != lab_round) {
              vs[t->to].lab = lab_round;
              members.push_back(t->to);
            }
      }
      if (members.size() == 1) {
        iso.push(s);
        continue;
      }
      Comp c;
      c.members = members;
      if (fan_out) {
        found.push_back(std::move(c));
        continue;
      }
      std::sort(c.members.begin(), c.members.end());
      solve(c, ++round, scratch);
      heap.push(Key{static_cast<uint32_t>(c.path.size()), c.segroot, static_cast<uint32_t>(comps.size())});
      comps.push_back(std::move(c));
    }
    if (found.empty()) return;
    const uint32_t first_round = round + 1; // every solve has a round of its own
    round += static_cast<uint32_t>(found.size());
    parallel_dynamic(found.size(), 16, [&](size_t b, size_t e) {
      Scratch mine;
      for (size_t k = b; k < e; ++k) {
        std::sort(found[k].members.begin(), found[k].members.end());
        solve(found[k], first_round + static_cast<uint32_t>(k), mine);
      }
    });
    for (Comp &c : found) {
      heap.push(Key{static_cast<uint32_t>(c.path.size()), c.segroot, static_cast<uint32_t>(comps.size())});
      comps.push_back(std::move(c));
    }
  }

  // sortTopologically + findConservationPathAlt on one component
  void solve(Comp &c, const uint32_t round, Scratch &sc) { // (round, sc: this solve's own -- solves may run side by side)
    std::vector<Node>     &nodes = sc.nodes;
    std::vector<uint32_t> &order = sc.order, &stack = sc.stack, &max_outs = sc.max_outs;
    auto node = [&](uint32_t v, int32_t parent) {
      nodes.push_back(Node{v, parent});
      return static_cast<int32_t>(nodes.size() - 1);
    };
    nodes.clear();
    order.clear();
    order.reserve(c.members.size());
    for (uint32_t v : c.members) {
      vs[v].deg = 0;
      for (const Arc *t = dg.lpred.begin(v); t != dg.lpred.end(v); ++t) vs[v].deg += alive[t->e];
    }
    for (size_t i = c.members.size(); i-- > 0;) { // zero-in-degree vertices, highest id first
      const uint32_t r = c.members[i];
      bool           is_root = true;
      for (const Arc *t = dg.lpred.begin(r); t != dg.lpred.end(r) && is_root; ++t) is_root = !alive[t->e];
      if (!is_root) continue;
      stack.assign(1, r);
      while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        vs[v].seg = r;
        order.push_back(v);
        for (const Arc *t = dg.lsucc.begin(v); t != dg.lsucc.end(v); ++t)
          if (alive[t->e] && --vs[t->to].deg == 0) stack.push_back(t->to);
      }
    }
    auto has_open = [&](uint32_t v) { return vs[v].ostamp == round; };
    auto touch    = [&](uint32_t v) { // operator[] of the reference: default-constructs (0, {})
      if (vs[v].ostamp != round) {
        vs[v].ostamp = round;
        vs[v].oscore = 0;
        vs[v].olen   = 0;
        vs[v].ohead  = -1;
      }
    };
    uint32_t final_len = 0, final_single = NIL, final_seg = 0;
    int32_t  final_head = -1;
    for (uint32_t v : order) {
      uint64_t max_out = 0;
      max_outs.clear();
      bool sink = true;
      for (const Arc *t = dg.lsucc.begin(v); t != dg.lsucc.end(v); ++t) {
        if (!alive[t->e]) continue;
        sink = false;
        const uint64_t w = cw[t->e];
        if (w > max_out) {
          max_out = w;
          max_outs.assign(1, t->to);
        } else if (w == max_out) {
          max_outs.push_back(t->to);
        }
      }
      if (sink) {
        if (!has_open(v)) {
          if (final_len == 0) {
            final_len    = 1;
            final_single = v;
            final_head   = -1;
            final_seg    = vs[v].seg;
          }
        } else {
          if (vs[v].olen > final_len) {
            final_len    = vs[v].olen;
            final_head   = vs[v].ohead;
            final_single = NIL;
            final_seg    = vs[v].seg;
          }
          vs[v].olen  = 0;
          vs[v].ohead = -1;
        }
        continue;
      }
      for (uint32_t nxt : max_outs) {
        if (has_open(nxt)) {
          bool take;
          if (vs[nxt].oscore < max_out) take = true;
          else if (vs[nxt].oscore == max_out) {
            touch(v);
            take = vs[nxt].olen < vs[v].olen + 1;
          } else take = false;
          if (take) {
            touch(v);
            vs[nxt].ohead  = node(nxt, vs[v].ohead);
            vs[nxt].olen   = vs[v].olen + 1;
            vs[nxt].oscore = max_out;
          }
        } else if (has_open(v)) {
          touch(nxt);
          vs[nxt].ohead  = node(nxt, vs[v].ohead);
          vs[nxt].olen   = vs[v].olen + 1;
          vs[nxt].oscore = max_out;
        } else {
          touch(nxt);
          vs[nxt].ohead  = node(nxt, node(v, -1));
          vs[nxt].olen   = 2;
          vs[nxt].oscore = max_out;
        }
      }
      touch(v);
      vs[v].olen  = 0;
      vs[v].ohead = -1;
    }
    c.path.clear();
    c.segroot = final_seg;
    if (final_single != NIL) {
      c.path.push_back(final_single);
    } else {
      for (int32_t p = final_head; p >= 0; p = nodes[p].parent) c.path.push_back(nodes[p].v);
      std::reverse(c.path.begin(), c.path.end());
    }
    if (c.path.empty()) throw GraphError("extractPaths: empty path");
  }

  void delete_vertex(uint32_t v) {
    for (const Csr *c : {&dg.lsucc, &dg.lpred})
      for (const Arc *t = c->begin(v); t != c->end(v); ++t)
        if (alive[t->e]) {
          alive[t->e] = 0;
          --n_arcs;
        }
    valive[v] = 0;
  }
};

std::vector<std::vector<uint32_t>> extract_paths(DiG &dg) { // local vertex ids
  Tick                 tick;
  std::vector<uint8_t>               alive(dg.m());
  std::vector<std::vector<uint32_t>> le_of(stage_threads() + 1);
  parallel_chunks(dg.m(), [&](unsigned chunk, size_t b, size_t e_end) { // the copy without its shadow edges (:351-356)
    for (size_t e = b; e < e_end; ++e) {
      alive[e] = !dg.shadow[e];
      if (alive[e]) le_of[chunk].push_back(static_cast<uint32_t>(e));
    }
  });
  // (the copy's own adjacency: 3 % of the arcs on BASELINE configs[2]; every pass below still asks `alive`, which only loses edges)
  {
    std::vector<uint32_t> le, la, lb;
    for (auto &v : le_of) le.insert(le.end(), v.begin(), v.end()); // (chunk order = edge order)
    for (uint32_t e : le) {
      la.push_back(dg.ea[e]);
      lb.push_back(dg.eb[e]);
    }
    dg.lsucc = build_csr(dg.n, la.data(), lb.data(), le.size(), false);
    dg.lpred = build_csr(dg.n, lb.data(), la.data(), le.size(), false);
    for (Csr *c : {&dg.lsucc, &dg.lpred})
      for (Arc &t : c->arcs) t.e = le[t.e]; // (build_csr numbers the edges it is given)
  }
  tick("copy + drop shadow edges");
  sort_reduction_by_weight(dg, alive);
  tick("sortReductionByWeight");
  const std::vector<uint64_t> cw = dg.n < 150000 ? find_cluster_weights(dg, alive) : find_cluster_weights_heuristic(dg, alive);
  tick("findClusterWeights");
  std::vector<std::vector<uint32_t>> paths;
  std::vector<uint8_t>               visited(dg.n, 0);
  PathPeeler                         pp(dg, alive, cw);
  {
    std::vector<uint32_t> all(dg.n);
    for (uint32_t v = 0; v < dg.n; ++v) all[v] = v;
    pp.split(all);
  }
  tick("conservation paths: first split");
  std::vector<uint32_t> longest;
  size_t                n_rounds = 0;
  while (pp.n_arcs > 0) { // diGraphCycle.getSize() > 0
    ++n_rounds;
    require(!pp.heap.empty(), "extractPaths: edges left but no component");
    const PathPeeler::Key top = pp.heap.top();
    bool                  lone = false;
    if (top.size == 1) { // only then can a vertex without edges come first in the topological order and win
      while (!pp.iso.empty() && !pp.valive[pp.iso.top()]) pp.iso.pop();
      lone = !pp.iso.empty() && pp.iso.top() > top.segroot;
    }
    std::vector<uint32_t> pool;
    if (lone) {
      longest.assign(1, pp.iso.top());
      pp.iso.pop();
    } else {
      pp.heap.pop();
      longest.swap(pp.comps[top.comp].path); // (its component is done with it)
      pool.swap(pp.comps[top.comp].members);
      pp.comps[top.comp].path.clear();
    }
    if (longest.size() < 10) {
      bool in_visit = false, out_visit = false;
      for (const Arc *p = dg.pred.begin(longest.front()); p != dg.pred.end(longest.front()); ++p)
        in_visit = in_visit || visited[p->to];
      for (const Arc *q = dg.succ.begin(longest.back()); q != dg.succ.end(longest.back()); ++q)
        out_visit = out_visit || visited[q->to];
      if ((!in_visit && !out_visit) || ((in_visit || out_visit) && longest.size() > 5)) paths.push_back(longest);
    } else {
      paths.push_back(longest);
    }
    for (uint32_t v : longest) {
      visited[v] = 1;
      pp.delete_vertex(v);
    }
    pp.split(pool);
  }
  if (std::getenv("MSGPU_GRAPH_DEBUG")) fprintf(stderr, "[graph] %zu peeling rounds, %zu paths kept, %zu solves\n", n_rounds, paths.size(), static_cast<size_t>(pp.round));
  tick("conservation paths loop");
  // lg.cpp:409-411 appends a path of one vertex for every vertex left over.  linearizeGraph (the only caller) joins no such
  // path to anything -- it has nothing in front of its vertex and nothing behind it (the test of lg.cpp:560) -- and drops every
  // path of one vertex at its end (lg.cpp:626); no vertex left over is on another path.  So they are not made: tens of thousands
  // of one-element vectors on a large component.
  return paths;
}

std::vector<std::vector<uint32_t>> linearize_graph(DiG &dg) { // lg.cpp:522-629
  std::vector<std::vector<uint32_t>> paths = extract_paths(dg);
  std::vector<size_t>                color_corr(paths.size()), color_len(paths.size());
  std::vector<uint32_t> v2idx(dg.n, NIL), v2pos(dg.n, 0); // v2pos: what std::find returns on the still unjoined paths
  for (size_t i = 0; i < paths.size(); ++i) {
    for (size_t k = 0; k < paths[i].size(); ++k)
      if (v2idx[paths[i][k]] == NIL) {
        v2idx[paths[i][k]] = static_cast<uint32_t>(i);
        v2pos[paths[i][k]] = static_cast<uint32_t>(k);
      }
    color_corr[i] = i;
    color_len[i]  = paths[i].size();
  }
  auto index_in = [](const std::vector<uint32_t> &p, uint32_t v) {
    return static_cast<size_t>(std::find(p.begin(), p.end(), v) - p.begin());
  };
  std::vector<std::pair<size_t, uint32_t>> joins;
  for (size_t e = 0; e < dg.m(); ++e) {
    if (!dg.shadow[e]) continue;
    const uint32_t a = dg.ea[e], b = dg.eb[e];
    if (v2idx[a] == NIL || v2idx[b] == NIL) continue;
    const size_t i1 = v2idx[a], i2 = v2idx[b], s1 = v2pos[a], s2 = v2pos[b];
    const size_t l1_end = color_len[i1] - s1 - 1, l2_end = color_len[i2] - s2 - 1;
    if (i1 != i2 && l1_end < s1 && s2 < l2_end) joins.emplace_back(l1_end + s2, static_cast<uint32_t>(e));
  }
  std::stable_sort(joins.begin(), joins.end(),
                   [](const std::pair<size_t, uint32_t> &x, const std::pair<size_t, uint32_t> &y) { return x.first < y.first; });
  for (auto &j : joins) {
    const size_t dist = j.first;
    if (dist > 3) break;
    auto color = [&](size_t i) {
      while (color_corr[i] != i) i = color_corr[i];
      return i;
    };
    const uint32_t a = dg.ea[j.second], b = dg.eb[j.second];
    const size_t   c1 = color(v2idx[a]), c2 = color(v2idx[b]);
    if (c1 == c2) continue;
    const size_t i1 = index_in(paths[c1], a), i2 = index_in(paths[c2], b);
    if (i1 == paths[c1].size() || i2 == paths[c2].size()) continue;
    if (color_len[c1] - i1 - 1 + i2 != dist) continue;
    paths[c1].erase(paths[c1].begin() + static_cast<long>(i1 + 1), paths[c1].end());
    paths[c1].insert(paths[c1].end(), paths[c2].begin() + static_cast<long>(i2), paths[c2].end());
    paths[c2].clear();
    color_corr[c2] = color_corr[c1];
    color_len[c1]  = paths[c1].size();
    color_len[c2]  = 0;
  }
  paths.erase(std::remove_if(paths.begin(), paths.end(), [](const std::vector<uint32_t> &p) { return p.size() <= 1; }),
              paths.end());
  return paths;
}

// one connected component: getDirectedGraph + linearizeGraph + the assemblePath inputs of its paths (main.cpp:620-661)
std::vector<msgpu_graph::PathStore> component_paths(msgpu_graph *g, uint32_t cid, const std::vector<uint32_t> &comp,
                                                    const RawVec<uint8_t> &arc_flags) {
  uint32_t start = NIL;
  {
    std::vector<uint32_t> sorted(comp);
    std::sort(sorted.begin(), sorted.end());
    for (uint32_t v : sorted) // std::max_element: the first of the longest, vertices ascending
      if (start == NIL || g->V[v].len > g->V[start].len) start = v;
  }
  Tick tk;
  DiG  dg = get_directed_graph(*g, cid, comp, start, arc_flags);
  tk("getDirectedGraph");
  const std::vector<std::vector<uint32_t>> lin = linearize_graph(dg);
  tk("linearizeGraph");
  // (a path's store is made of its own vertices and edges alone: the paths of a large component on the stage's threads)
  std::vector<msgpu_graph::PathStore> out(lin.size());
  parallel_dynamic(lin.size(), lin.size() >= 256 ? 16 : std::max<size_t>(1, lin.size()), [&](size_t p_begin, size_t p_end) {
  for (size_t pi = p_begin; pi < p_end; ++pi) {
    const std::vector<uint32_t> &p = lin[pi];
    msgpu_graph::PathStore ps;
    ps.order_off.push_back(0);
    ps.em_off.push_back(0);
    for (uint32_t l : p) {
      const uint32_t  v = dg.ids[l];
      msgpu_path_read r{};
      r.read_id         = v;
      r.direction       = g->V[v].dir == D_POS ? 1u : g->V[v].dir == D_NEG ? 0u : 2u;
      r.nanopore_length = static_cast<uint64_t>(g->V[v].len);
      ps.reads.push_back(r);
    }
    for (size_t i = 0; i + 1 < p.size(); ++i) {
      const int64_t de = dg.get_edge(p[i], p[i + 1]);
      require(de >= 0, "path edge missing in the directed graph");
      for (uint32_t q = dg.ord_off[de]; q < dg.ord_off[de + 1]; ++q) {
        const msgpu_order &o = g->t_orders[dg.ord[q]];
        msgpu_path_order   po{};
        po.score     = o.score;
        po.base_read = o.base;
        po.ids_off   = static_cast<uint32_t>(o.ids_off);
        po.ids_cnt   = o.ids_cnt;
        ps.orders.push_back(po);
      }
      const msgpu_edge &te = g->t_edges[dg.src[de]];
      ps.step_edge.push_back(dg.src[de]);
      if (!g->ems_on_demand)
        for (uint32_t k = 0; k < te.em_cnt; ++k) {
          const msgpu_edgematch &m = g->t_ems[te.em_off + k];
          ps.ems.push_back(msgpu_path_em{m.anchor_id, m.ov_lo, m.ov_hi});
        }
      ps.order_off.push_back(static_cast<uint32_t>(ps.orders.size()));
      ps.em_off.push_back(static_cast<uint32_t>(ps.ems.size()));
    }
    for (uint32_t l : p) {
      const uint32_t v = dg.ids[l];
      if (g->contain_first.empty()) continue;
      for (uint64_t q = g->contain_first[v]; q < g->contain_first[v + 1]; ++q) {
        const msgpu_graph::Contain &ce = g->contain[q];
        msgpu_path_contain          pc{};
        pc.host_read   = v;
        pc.nano        = ce.nano;
        pc.direction   = ce.direction;
        pc.anchors_off = static_cast<uint32_t>(ps.contain_anchors.size());
        pc.anchors_cnt = ce.anchors_cnt;
        ps.contain_anchors.insert(ps.contain_anchors.end(), g->contain_anchor.begin() + static_cast<long>(ce.anchors_off),
                                  g->contain_anchor.begin() + static_cast<long>(ce.anchors_off + ce.anchors_cnt));
        ps.contains.push_back(pc);
      }
    }
    out[pi] = std::move(ps);
  }
  });
  tk("path stores");
  return out;
}

} // namespace

extern "C" {

static int graph_create(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                        const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                        const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads, bool copy,
                        msgpu_graph **out) {
  // ems == NULL (with edges that carry em_off / em_cnt): the EdgeMatch table stays where it is (HBM); the EdgeMatches of
  // the path edges arrive later through msgpu_graph_set_path_edgematches
  const bool on_demand = ems == nullptr;
  if (!out || (n_edges && !edges) || (n_orders && !orders) || (n_ids && !ids) ||
      (n_reads && (!read_len || !read_first_line)) || n_edges >= 0x7ffffff0ull || n_orders >= 0xfffffff0ull ||
      n_ids >= 0xfffffff0ull)
    return MSGPU_E_ARG;
  *out = nullptr;
  try {
    Tick                         tick;
    std::unique_ptr<msgpu_graph> g(new msgpu_graph());
    if (copy) {
      g->own_edges.assign(edges, edges + n_edges);
      g->own_orders.assign(orders, orders + n_orders);
      g->own_ids.assign(ids, ids + n_ids);
      if (!on_demand) g->own_ems.assign(ems, ems + n_ems);
      edges  = g->own_edges.data();
      orders = g->own_orders.data();
      ids    = g->own_ids.data();
      ems    = on_demand ? nullptr : g->own_ems.data();
    }
    g->ems_on_demand = on_demand;
    g->t_edges  = edges;
    g->t_ems    = ems;
    g->t_orders = orders;
    g->t_ids    = ids;
    g->n_edges  = n_edges;
    g->n_ems    = n_ems;
    g->n_orders = n_orders;
    g->n_ids    = n_ids;
    g->nv       = n_reads;
    g->V.resize(n_reads);
    for (uint32_t v = 0; v < n_reads; ++v) {
      g->V[v].len   = read_len[v];
      g->V[v].meta0 = read_first_line[v];
    }
    tick("  create: copies + vertices");
    g->E.allocate(n_edges);
    g->o_kept.assign(n_orders, 0);
    g->o_fwd.assign(n_orders, 0); // (sized here: the component workers of msgpu_graph_linearize write it side by side)
    g->ol.allocate(n_orders);
    tick("  create: table allocation");
    std::atomic<int> bad{0};
    parallel_chunks(n_edges, [&](unsigned, size_t b, size_t e_end) {
      for (size_t i = b; i < e_end; ++i) {
        const msgpu_edge &e = edges[i];
        if (e.v1 >= n_reads || e.v2 >= n_reads || e.v1 == e.v2 || e.order_off + e.order_cnt > n_orders ||
            (!on_demand && e.em_off + e.em_cnt > n_ems)) {
          bad = 1;
          return;
        }
        msgpu_graph::Edge &u = *new (&g->E[i]) msgpu_graph::Edge();
        u.a       = e.v1;
        u.b       = e.v2;
        u.shadow  = e.shadow != 0;
        u.ord_lo  = static_cast<uint32_t>(e.order_off);
        u.ord_cnt = e.order_cnt;
        for (uint32_t k = 0; k < e.order_cnt; ++k) g->o_kept[e.order_off + k] = 1;
        if (e.order_cnt) u.first = u.ord_lo;
      }
    });
    if (bad) return MSGPU_E_ARG;
    {
      // an order on two edges: the edges' order ranges would cover fewer orders than their lengths add up to
      std::vector<uint64_t> sums(stage_threads() + 1, 0), kept(stage_threads() + 1, 0);
      parallel_chunks(n_edges, [&](unsigned c, size_t b, size_t e_end) {
        uint64_t s = 0;
        for (size_t i = b; i < e_end; ++i) s += edges[i].order_cnt;
        sums[c] = s;
      });
      parallel_chunks(n_orders, [&](unsigned c, size_t b, size_t e_end) {
        uint64_t s = 0;
        for (size_t k = b; k < e_end; ++k) {
          const msgpu_order &o = orders[k];
          if (o.start >= n_reads || o.end >= n_reads || o.base >= n_reads || o.ids_off + o.ids_cnt > n_ids) {
            bad = 1;
            return;
          }
          g->ol[k] = msgpu_graph::OLite{o.start, o.end, o.base, o.flags};
          s += g->o_kept[k];
        }
        kept[c] = s;
      });
      if (bad) return MSGPU_E_ARG;
      uint64_t a = 0, b = 0;
      for (uint64_t x : sums) a += x;
      for (uint64_t x : kept) b += x;
      if (a != b) return MSGPU_E_ARG;
    }
    tick("  create: edge + order records");
    {
      std::unique_ptr<uint32_t[]> ea(new uint32_t[n_edges ? n_edges : 1]), eb(new uint32_t[n_edges ? n_edges : 1]); // (not zero-filled)
      parallel_chunks(n_edges, [&](unsigned, size_t b, size_t e_end) {
        for (size_t i = b; i < e_end; ++i) {
          ea[i] = edges[i].v1;
          eb[i] = edges[i].v2;
        }
      });
      g->adj = build_csr(n_reads, ea.get(), eb.get(), n_edges, true);
    }
    tick("  create: adjacency");
    parallel_chunks(n_reads, [&](unsigned, size_t b, size_t e_end) { // duplicate edge
      for (size_t v = b; v < e_end; ++v)
        for (uint32_t q = g->adj.off[v]; q + 1 < g->adj.off[v + 1]; ++q)
          if (g->adj.arcs[q].to == g->adj.arcs[q + 1].to) bad = 1;
    });
    if (bad) return MSGPU_E_ARG;
    g->stats.n_vertices_in = n_reads;
    g->stats.n_edges_in    = n_edges;
    tick("graph create");
    *out = g.release();
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

int msgpu_graph_create(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                       const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                       const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads, msgpu_graph **out) {
  return graph_create(edges, n_edges, ems, n_ems, orders, n_orders, ids, n_ids, read_len, read_first_line, n_reads, true, out);
}
int msgpu_graph_create_borrowed(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                                const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                                const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads,
                                msgpu_graph **out) {
  return graph_create(edges, n_edges, ems, n_ems, orders, n_orders, ids, n_ids, read_len, read_first_line, n_reads, false, out);
}

void        msgpu_graph_free(msgpu_graph *g) { delete g; }
const char *msgpu_graph_last_error(const msgpu_graph *g) { return g ? g->err : "null graph"; }

// src/main.cpp:194-288 after findContractionEdges
int msgpu_graph_clean_up(msgpu_graph *g, const int64_t *contraction_order, const msgpu_row *rows, size_t n_rows) {
  if (!g || (g->n_edges && !contraction_order) || (n_rows && !rows)) return MSGPU_E_ARG;
  if (g->cleaned) return MSGPU_E_STATE;
  g->err[0] = 0;
  try {
    g_tick_epoch = std::chrono::steady_clock::now();
    Tick                  tick;
    const uint32_t        nv = g->nv;
    const size_t          ne = g->n_edges;
    std::vector<uint32_t> contraction; // order indices, in edge order
    for (size_t e = 0; e < ne; ++e) {
      const int64_t k = contraction_order[e];
      if (k < 0) continue;
      if (static_cast<uint64_t>(k) >= g->n_orders || g->t_orders[k].edge_idx != e) return MSGPU_E_ARG;
      contraction.push_back(static_cast<uint32_t>(k));
    }
    g->stats.n_contraction_edges = contraction.size();
    tick("  contraction: its orders");
    // MatchMap::getVertexMatch(read, anchor) != nullptr for the anchors of the contraction orders.  A PAF is grouped by
    // its query and the Registry numbers the anchors as they come, so the rows of anchor a are rows[first[a] .. first[a + 1])
    // as they stand (checked): a look-up walks the handful of rows of one anchor.  Rows in any other order are keyed and
    // sorted instead (the tests' shuffled tables).
    std::vector<uint64_t> has_vm;    // sorted (read, anchor) keys -- only when the rows are not grouped by ascending anchor
    std::vector<uint64_t> first_row; // first_row[a] .. first_row[a + 1]: the rows of anchor a -- when they are
    if (!contraction.empty() && rows) {
      std::atomic<int> unordered{0};
      parallel_chunks(n_rows, [&](unsigned, size_t b, size_t e) {
        for (size_t i = std::max<size_t>(b, 1); i < e; ++i)
          if (rows[i].anchor_id < rows[i - 1].anchor_id) {
            unordered = 1;
            return;
          }
      });
      if (!unordered && n_rows) {
        const uint32_t n_anchors = rows[n_rows - 1].anchor_id + 1;
        first_row.assign(static_cast<size_t>(n_anchors) + 1, 0);
        parallel_chunks(n_rows, [&](unsigned, size_t b, size_t e) { // first_row[a] for every anchor a that BEGINS in [b, e), and for the empty ones in front of it
          for (size_t i = b; i < e; ++i) {
            const uint32_t a = rows[i].anchor_id, before = i ? rows[i - 1].anchor_id : 0;
            if (i == 0) {
              for (uint32_t x = 0; x <= a; ++x) first_row[x] = 0;
            } else if (a != before) {
              for (uint32_t x = before + 1; x <= a; ++x) first_row[x] = i;
            }
          }
        });
        first_row[n_anchors] = n_rows;
      } else {
        has_vm.resize(n_rows);
        for (size_t i = 0; i < n_rows; ++i) has_vm[i] = (static_cast<uint64_t>(rows[i].read_id) << 32) | rows[i].anchor_id;
        std::sort(has_vm.begin(), has_vm.end());
      }
    }
    tick("  contraction: rows by anchor");
    auto has_vertex_match = [&](uint32_t read, uint32_t anchor) {
      if (!rows) return true;
      if (!first_row.empty()) {
        if (static_cast<size_t>(anchor) + 1 >= first_row.size()) return false;
        for (uint64_t i = first_row[anchor]; i < first_row[anchor + 1]; ++i)
          if (rows[i].read_id == read) return true;
        return false;
      }
      return std::binary_search(has_vm.begin(), has_vm.end(), (static_cast<uint64_t>(read) << 32) | anchor);
    };
    std::vector<uint32_t> targets(nv);
    for (uint32_t v = 0; v < nv; ++v) targets[v] = v; // :194-197
    for (uint32_t k : contraction) {                  // findContractionTargets, :465-482
      const msgpu_order &o  = g->t_orders[k];
      const uint32_t     to = targets[o.end];
      if (targets[o.start] == o.start || g->V[targets[o.start]].meta0 > g->V[to].meta0) targets[o.start] = to;
    }
    std::vector<uint8_t> deletable(nv, 0), roots(nv, 0);
    for (uint32_t k : contraction) { // findDeletableVertices, :484-507
      const msgpu_order &o = g->t_orders[k];
      deletable[o.start]      = 1;
      roots[targets[o.start]] = 1;
      roots[o.start]          = 0;
    }
    tick("  contraction: targets + roots");
    // contract, :509-531.  Which anchors of a contraction order have a VertexMatch on the contained read is looked up on
    // the stage's threads (two passes: count, fill); the ContainElements enter the map in contraction order.
    std::vector<uint64_t> keep_off(contraction.size() + 1, 0);
    parallel_chunks(contraction.size(), [&](unsigned, size_t b, size_t e) {
      for (size_t j = b; j < e; ++j) {
        const msgpu_order &o = g->t_orders[contraction[j]];
        uint64_t           n = 0;
        if (roots[o.end])
          for (uint32_t i = 0; i < o.ids_cnt; ++i) n += has_vertex_match(o.start, g->t_ids[o.ids_off + i]);
        keep_off[j + 1] = n;
      }
    });
    for (size_t j = 0; j < contraction.size(); ++j) keep_off[j + 1] += keep_off[j];
    g->contain_anchor.resize(keep_off[contraction.size()]);
    uint32_t *const kept_anchor = g->contain_anchor.data();
    parallel_chunks(contraction.size(), [&](unsigned, size_t b, size_t e) {
      for (size_t j = b; j < e; ++j) {
        const msgpu_order &o = g->t_orders[contraction[j]];
        if (!roots[o.end]) continue;
        uint64_t at = keep_off[j];
        for (uint32_t i = 0; i < o.ids_cnt; ++i) {
          const uint32_t a = g->t_ids[o.ids_off + i];
          if (has_vertex_match(o.start, a)) kept_anchor[at++] = a;
        }
      }
    });
    tick("  contraction: anchors with a VertexMatch");
    // the ContainElements by host read, those of one read in contraction order (a counting sort by host)
    g->contain_first.assign(static_cast<size_t>(nv) + 1, 0);
    for (uint32_t k : contraction) {
      const msgpu_order &o = g->t_orders[k];
      if (roots[o.end]) ++g->contain_first[o.end + 1];
    }
    for (uint32_t v = 0; v < nv; ++v) g->contain_first[v + 1] += g->contain_first[v];
    g->contain.allocate(g->contain_first[nv]);
    {
      std::vector<uint64_t> cur(g->contain_first.begin(), g->contain_first.end() - 1);
      for (size_t j = 0; j < contraction.size(); ++j) {
        const uint32_t     k = contraction[j];
        const msgpu_order &o = g->t_orders[k];
        if (!roots[o.end]) continue;
        g->contain[cur[o.end]++] = msgpu_graph::Contain{o.start, g->odir(k) ? 1u : 0u, keep_off[j], static_cast<uint32_t>(keep_off[j + 1] - keep_off[j])};
        ++g->stats.n_contain_elements;
      }
    }
    tick("contraction bookkeeping");
    for (uint32_t v = 0; v < nv; ++v) // :242-244
      if (deletable[v]) {
        g->delete_vertex(v);
        ++g->stats.n_deleted_vertices;
      }
    // findDeletableEdges (:534-549, + deletion :258-260) and computeBitweight (:551-573): per-edge independent, on all host
    // threads; the candidate list of the span tree is the per-chunk lists in chunk order = edge order (:264, mst.cpp:79-86)
    std::vector<std::vector<uint32_t>> cand_of(stage_threads() + 1);
    parallel_chunks(ne, [&](unsigned chunk, size_t e_begin, size_t e_end) {
      std::vector<uint32_t> &mine = cand_of[chunk];
      for (size_t e = e_begin; e < e_end; ++e) {
        if (!g->E[e].alive) continue;
        const uint32_t lo = g->E[e].ord_lo, hi = lo + g->E[e].ord_cnt;
        uint32_t       first = NIL;
        for (uint32_t oi = lo; oi < hi; ++oi) {
          if (g->ocont(oi)) g->o_kept[oi] = 0;
          else if (first == NIL) first = oi;
        }
        g->E[e].first = first;
        if (first == NIL) {
          g->delete_edge(static_cast<uint32_t>(e));
          continue;
        }
        const bool d0 = g->odir(first);
        if (g->E[e].shadow) {
          bool other = false;
          for (uint32_t oi = first; oi < hi; ++oi) other = other || (g->o_kept[oi] && g->odir(oi) != d0);
          if (!other) g->E[e].cons = d0 ? D_POS : D_NEG;
        } else {
          g->E[e].weight = g->t_orders[first].score;
          g->E[e].cons   = d0 ? D_POS : D_NEG;
        }
        if (g->E[e].cons != D_NONE) mine.push_back(static_cast<uint32_t>(e));
      }
    });
    std::vector<uint32_t> cand; // alive edges with a consensus direction, in edge order
    for (auto &v : cand_of) cand.insert(cand.end(), v.begin(), v.end());
    tick("deletions + bitweight");
    std::vector<uint8_t> in_tree(ne, 0); // getMaxSpanTree, mst.cpp:75-111
    g->span_set.clear();
    max_span_tree(
        nv, [&](uint32_t e) { return std::make_pair(g->E[e].a, g->E[e].b); }, [&](uint32_t e) { return g->E[e].weight; }, cand,
        in_tree, &g->span_set);
    tick("span tree");
    // the span forest, rooted: parent / edge to parent / depth / number of e_NEG edges to the root (mod 2)
    std::vector<uint32_t> parent(nv, NIL), pedge(nv, NIL), depth(nv, 0);
    std::vector<uint8_t>  negpar(nv, 0);
    {
      // (over an adjacency of the tree edges alone -- one edge in ten here --, not over every arc of the graph; how the
      // forest is rooted does not matter to decycle: the tree path between two vertices is the same under any root)
      std::vector<uint32_t>              te, ta, tb;
      std::vector<std::vector<uint32_t>> te_of(stage_threads() + 1);
      parallel_chunks(cand.size(), [&](unsigned chunk, size_t b, size_t e_end) {
        for (size_t i = b; i < e_end; ++i)
          if (in_tree[cand[i]]) te_of[chunk].push_back(cand[i]);
      });
      for (auto &v : te_of) te.insert(te.end(), v.begin(), v.end()); // (chunk order = edge order)
      ta.resize(te.size());
      tb.resize(te.size());
      std::vector<uint8_t> tneg(te.size()); // the tree edge's consensus direction is e_NEG (read here, not record by record in the walk below)
      parallel_chunks(te.size(), [&](unsigned, size_t b, size_t e_end) {
        for (size_t k = b; k < e_end; ++k) {
          const msgpu_graph::Edge &E = g->E[te[k]];
          ta[k]   = E.a;
          tb[k]   = E.b;
          tneg[k] = E.cons == D_NEG;
        }
      });
      const Csr             tree = build_csr(nv, ta.data(), tb.data(), te.size(), true);
      std::vector<uint32_t> queue;
      for (uint32_t r = 0; r < nv; ++r) {
        if (parent[r] != NIL) continue;
        parent[r] = r;
        queue.assign(1, r);
        for (size_t h = 0; h < queue.size(); ++h) {
          const uint32_t v = queue[h];
          for (const Arc *t = tree.begin(v); t != tree.end(v); ++t)
            if (parent[t->to] == NIL) {
              parent[t->to] = v;
              pedge[t->to]  = te[t->e];
              depth[t->to]  = depth[v] + 1;
              negpar[t->to] = negpar[v] ^ tneg[t->e];
              queue.push_back(t->to);
            }
        }
      }
    }
    tick("span forest rooted");
    std::vector<uint8_t> dele(ne, 0);
    // decycle, :575-618: every non-tree edge is judged against the finished span forest on its own (dele is only ever set
    // to 1), so the candidates are cut over the host threads
    parallel_chunks(cand.size(), [&](unsigned, size_t c_begin, size_t c_end) {
      std::vector<uint32_t> left, right;
      for (size_t ci = c_begin; ci < c_end; ++ci) {
        const uint32_t e = cand[ci];
        if (in_tree[e]) continue;
        const uint32_t a = g->E[e].a, b = g->E[e].b;
        // direction folded over the tree path a..b (:590-603): XNOR chain = parity of the e_NEG edges on it
        const bool direction = !((g->E[e].cons == D_NEG) ^ negpar[a] ^ negpar[b]);
        if (direction) continue;
        left.clear();
        right.clear();
        uint32_t ua = a, ub = b;
        while (depth[ua] > depth[ub]) {
          left.push_back(pedge[ua]);
          ua = parent[ua];
        }
        while (depth[ub] > depth[ua]) {
          right.push_back(pedge[ub]);
          ub = parent[ub];
        }
        while (ua != ub) {
          require(parent[ua] != ua && parent[ub] != ub, "decycle: the span tree does not connect the ends of an edge");
          left.push_back(pedge[ua]);
          ua = parent[ua];
          right.push_back(pedge[ub]);
          ub = parent[ub];
        }
        left.insert(left.end(), right.rbegin(), right.rend()); // tree edges in path order a -> b
        if (left.empty()) continue;
        size_t lo = 0;
        double wlo = static_cast<double>(g->E[left[0]].weight), whi = wlo;
        for (size_t i = 1; i < left.size(); ++i) {
          const double w = static_cast<double>(g->E[left[i]].weight);
          if (w < wlo) { // std::min_element: the first minimum
            wlo = w;
            lo  = i;
          }
          if (w > whi) whi = w;
        }
        const double base = static_cast<double>(g->E[e].weight);
        if (wlo < base || (base * BASE_WEIGHT_MULTIPLICATOR >= wlo && wlo < whi * MAX_WEIGHT_MULTIPLICATOR))
          __atomic_store_n(&dele[left[lo]], uint8_t(1), __ATOMIC_RELAXED); // (several edges may name the same tree edge)
        __atomic_store_n(&dele[e], uint8_t(1), __ATOMIC_RELAXED);
      }
    });
    tick("decycle");
    std::atomic<uint64_t> n_dele{0}, n_alive{0}; // :285-287, and the edges that are left (one pass over the records, on all threads)
    parallel_chunks(ne, [&](unsigned, size_t e_begin, size_t e_end) {
      uint64_t d = 0, a = 0;
      for (size_t e = e_begin; e < e_end; ++e) {
        if (dele[e]) {
          g->delete_edge(static_cast<uint32_t>(e));
          ++d;
        }
        a += g->E[e].alive;
      }
      n_dele += d;
      n_alive += a;
    });
    g->stats.n_decycled_edges += n_dele;
    if (n_dele != 0) g->span_set.clear(); // (the sets of the span forest are the components only while decycle took no edge away)
    uint64_t nv_alive = 0;
    for (auto &x : g->V) nv_alive += x.alive;
    g->stats.n_vertices = nv_alive;
    g->stats.n_edges    = n_alive;
    g->cleaned          = true;
  } catch (std::bad_alloc const &) {
    return MSGPU_E_NOMEM;
  } catch (std::exception const &e) { // GraphError and anything a container throws
    snprintf(g->err, sizeof(g->err), "%s", e.what());
    return MSGPU_E_LAYOUT;
  }
  return MSGPU_OK;
}

// host threads for the per-component work of msgpu_graph_linearize (default 1; the reference runs one assemblePaths job
// per component on its ThreadPool, src/main.cpp:300-310)
int msgpu_graph_set_threads(msgpu_graph *g, uint32_t n_threads) {
  if (!g) return MSGPU_E_ARG;
  g->n_threads = n_threads ? n_threads : 1;
  return MSGPU_OK;
}

// getConnectedComponents (cc.cpp:33-70) + per component getDirectedGraph + linearizeGraph (src/main.cpp:300-310, 620-661)
int msgpu_graph_linearize(msgpu_graph *g) {
  if (!g) return MSGPU_E_ARG;
  if (!g->cleaned || g->linearized) return MSGPU_E_STATE;
  g->err[0] = 0;
  try {
    g_tick_epoch = std::chrono::steady_clock::now();
    Tick tick;
    // what the walks read of an edge, next to the arc (a byte of flags instead of a 40-byte record, and no random access)
    RawVec<uint8_t> arc_flags(g->adj.arcs.size());
    std::atomic<int>     unkept{0};
    parallel_chunks(arc_flags.size(), [&](unsigned, size_t b, size_t e_end) {
      bool any = false;
      for (size_t q = b; q < e_end; ++q) {
        if (q + 24 < e_end) __builtin_prefetch(&g->E[g->adj.arcs[q + 24].e]); // (a random record per arc)
        const msgpu_graph::Edge &E = g->E[g->adj.arcs[q].e];
        arc_flags[q] = static_cast<uint8_t>((E.alive ? 1 : 0) | (E.first != NIL ? 2 : 0) | (E.cons != D_NONE ? 4 : 0) |
                                            (E.cons == D_POS ? 8 : 0));
        any = any || (E.alive && E.first == NIL);
      }
      if (any) unkept = 1;
    });
    g->every_alive_edge_kept = unkept == 0;
    tick("arc flags");
    const Arc *const arcs0 = g->adj.arcs.data();
    auto v_ok = [&](uint32_t v) { return g->V[v].alive != 0; };
    auto e_ok = [&](const Arc *t) { return (arc_flags[t - arcs0] & 5) == 5; }; // alive and with a consensus direction
    auto set_comp = [&](uint32_t v, uint32_t c) { g->V[v].comp = c; };
    std::vector<std::vector<uint32_t>> comps;
    if (g->span_set.size() == g->nv && g->nv) {
      // the span forest was grown over exactly these edges (alive, with a consensus direction; decycle() removed none): its
      // sets are the components.  Numbered as the scans below number them: by their first vertex in the graph, members ascending.
      std::vector<uint32_t> cid(g->nv, NIL);
      uint32_t              n_comp = 0;
      for (uint32_t s = 0; s < g->nv; ++s)
        if (v_ok(s) && cid[g->span_set[s]] == NIL) cid[g->span_set[s]] = n_comp++;
      comps.resize(n_comp);
      for (uint32_t v = 0; v < g->nv; ++v) {
        const uint32_t c = cid[g->span_set[v]];
        if (c == NIL) continue;
        set_comp(v, c);
        comps[c].push_back(v);
      }
    } else {
      comps = g->nv >= par_min() && stage_threads() > 1
                  ? connected_components_parallel(g->nv, g->adj, v_ok, e_ok, set_comp)
                  : connected_components(g->nv, g->adj, v_ok, e_ok, [&](uint32_t v) { return g->V[v].comp; }, set_comp);
    }
    g->stats.n_components = comps.size();
    g->walk_seq.assign(g->nv, NIL);
    tick("components");
    // Components are independent (a component only orients and reads its own vertices and edges): largest first on the
    // worker threads, results appended in component order -- the order a single-threaded reference run assembles them in.
    std::vector<std::vector<msgpu_graph::PathStore>> per(comps.size());
    std::vector<std::string>                         errs(comps.size());
    std::vector<int>                                 rcs(comps.size(), MSGPU_OK);
    std::vector<size_t>                              by_size(comps.size());
    for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = i;
    std::stable_sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return comps[x].size() > comps[y].size(); });
    // the workers' own loops (directed edges, adjacency builds, cluster weights) share the stage's threads: a large
    // component gets its share of them, a small one runs on its worker alone
    // (by the SQUARE of its size: the largest component is the stage's critical path -- its walk and its linearizeGraph
    // are serial -- and the smaller ones have the time its serial parts take to finish their own parallel loops on fewer threads)
    double large_weight = 0;
    for (const auto &c : comps)
      if (c.size() >= par_min() / 8) large_weight += static_cast<double>(c.size()) * static_cast<double>(c.size());
    std::atomic<size_t> next{0};
    auto                work = [&]() {
      for (size_t k = next.fetch_add(1); k < by_size.size(); k = next.fetch_add(1)) {
        const size_t i = by_size[k];
        // (its share of the vertices of the large components, at least one thread)
        const double mine = static_cast<double>(comps[i].size()) * static_cast<double>(comps[i].size());
        tl_thread_share   = comps[i].size() >= par_min() / 8 && large_weight > 0
                                ? std::max<unsigned>(1, static_cast<unsigned>(stage_threads_total() * mine / large_weight + 0.5))
                                : 1;
        tl_tick_tag = static_cast<int>(i);
        if (std::getenv("MSGPU_GRAPH_DEBUG")) fprintf(stderr, "[graph c%zu] %zu vertices, %u threads\n", i, comps[i].size(), tl_thread_share);
        try {
          per[i] = component_paths(g, static_cast<uint32_t>(i), comps[i], arc_flags);
        } catch (std::bad_alloc const &) { rcs[i] = MSGPU_E_NOMEM; } catch (std::exception const &e) {
          rcs[i]  = MSGPU_E_LAYOUT;
          errs[i] = e.what();
        }
        tl_thread_share = 0;
        tl_tick_tag     = -1;
      }
    };
    uint32_t nt = g->n_threads;
    if (nt > comps.size()) nt = static_cast<uint32_t>(comps.size() ? comps.size() : 1);
    std::vector<std::thread> pool;
    JoinAll                  join_all{pool};
    for (uint32_t t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < comps.size(); ++i)
      if (rcs[i] != MSGPU_OK) {
        snprintf(g->err, sizeof(g->err), "%s", errs[i].c_str());
        return rcs[i];
      }
    for (auto &v : per)
      for (auto &ps : v) {
        g->stats.n_path_reads += ps.reads.size();
        g->paths.push_back(std::move(ps));
      }
    g->stats.n_paths = g->paths.size();
    g->path_edges.clear();
    for (const auto &ps : g->paths) g->path_edges.insert(g->path_edges.end(), ps.step_edge.begin(), ps.step_edge.end());
    g->linearized    = true;
  } catch (std::bad_alloc const &) {
    return MSGPU_E_NOMEM;
  } catch (std::exception const &e) {
    snprintf(g->err, sizeof(g->err), "%s", e.what());
    return MSGPU_E_LAYOUT;
  }
  return MSGPU_OK;
}

int msgpu_graph_get_stats(const msgpu_graph *g, msgpu_graph_stats *out) {
  if (!g || !out) return MSGPU_E_ARG;
  *out = g->stats;
  return MSGPU_OK;
}

uint32_t msgpu_graph_path_count(const msgpu_graph *g) { return g ? static_cast<uint32_t>(g->paths.size()) : 0; }

// path i as the input of msgpu_assembly_add_path(s); the pointers stay valid until msgpu_graph_free.  asm_idx = i
// (the reference numbers assemblies 0, 1, ... in the order assemblePathsSub runs, src/main.cpp:300,673)
int msgpu_graph_path_edges(const msgpu_graph *g, const uint32_t **edge_idx, size_t *n) {
  if (!g || !edge_idx || !n) return MSGPU_E_ARG;
  if (!g->linearized) return MSGPU_E_STATE;
  *edge_idx = g->path_edges.data();
  *n        = g->path_edges.size();
  return MSGPU_OK;
}

// the EdgeMatches of msgpu_graph_path_edges' list (what msgpu_get_edgematches returns for it), copied into the path inputs
int msgpu_graph_set_path_edgematches(msgpu_graph *g, const uint64_t *em_off, const msgpu_edgematch *ems) {
  if (!g || (!g->path_edges.empty() && !em_off)) return MSGPU_E_ARG;
  if (!g->linearized) return MSGPU_E_STATE;
  try {
    size_t step = 0;
    for (auto &ps : g->paths) {
      ps.ems.clear();
      ps.em_off.assign(1, 0);
      for (size_t k = 0; k < ps.step_edge.size(); ++k, ++step) {
        const msgpu_edge &te = g->t_edges[ps.step_edge[k]];
        if (em_off[step + 1] - em_off[step] != te.em_cnt || (te.em_cnt && !ems)) {
          snprintf(g->err, sizeof(g->err), "EdgeMatch list of path step %zu has %llu entries, its edge %u", step,
                   (unsigned long long)(em_off[step + 1] - em_off[step]), te.em_cnt);
          return MSGPU_E_ARG;
        }
        for (uint64_t q = em_off[step]; q < em_off[step + 1]; ++q)
          ps.ems.push_back(msgpu_path_em{ems[q].anchor_id, ems[q].ov_lo, ems[q].ov_hi});
        ps.em_off.push_back(static_cast<uint32_t>(ps.ems.size()));
      }
    }
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  g->have_path_ems = true;
  return MSGPU_OK;
}

int msgpu_graph_path_input(const msgpu_graph *g, uint32_t i, msgpu_path_input *out) {
  if (!g || !out || i >= g->paths.size()) return MSGPU_E_ARG;
  if (g->ems_on_demand && !g->have_path_ems) return MSGPU_E_STATE; // msgpu_graph_set_path_edgematches first
  const msgpu_graph::PathStore &p = g->paths[i];
  std::memset(out, 0, sizeof(*out));
  out->reads           = p.reads.data();
  out->n_reads         = static_cast<uint32_t>(p.reads.size());
  out->asm_idx         = static_cast<int32_t>(i);
  out->order_off       = p.order_off.data();
  out->orders          = p.orders.data();
  out->ids             = g->t_ids;
  out->em_off          = p.em_off.data();
  out->ems             = p.ems.data();
  out->contains        = p.contains.data();
  out->n_contains      = static_cast<uint32_t>(p.contains.size());
  out->contain_anchors = p.contain_anchors.data();
  return MSGPU_OK;
}

// The assemblePaths fan-out (src/main.cpp:620-677) straight from the graph: every linearised path goes to
// msgpu_assembly_add_paths (n_threads layout threads; status: msgpu_graph_path_count(g) entries or NULL) without the caller
// collecting the path inputs one by one.
int msgpu_assembly_add_graph_paths(msgpu_assembly *a, const msgpu_graph *g, uint32_t n_threads, int *status) {
  if (!a || !g) return MSGPU_E_ARG;
  try {
    std::vector<msgpu_path_input> in(g->paths.size());
    for (uint32_t i = 0; i < in.size(); ++i) {
      const int rc = msgpu_graph_path_input(g, i, &in[i]);
      if (rc != MSGPU_OK) return rc;
    }
    return msgpu_assembly_add_paths(a, in.data(), in.size(), n_threads, status);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
}

// alive[v] (n_reads entries, optional) = vertex still in the graph; edge_alive[e] (optional) likewise; direction[v]
// (optional) = Vertex::getVertexDirection() as 1 / 0 / 2 (e_POS / e_NEG / e_NONE)
int msgpu_graph_state(const msgpu_graph *g, uint8_t *vertex_alive, uint8_t *vertex_direction, uint8_t *edge_alive,
                      uint8_t *edge_consensus, uint64_t *edge_weight) {
  if (!g) return MSGPU_E_ARG;
  for (size_t v = 0; v < g->nv; ++v) {
    if (vertex_alive) vertex_alive[v] = g->V[v].alive;
    if (vertex_direction) vertex_direction[v] = g->V[v].dir == D_POS ? 1 : g->V[v].dir == D_NEG ? 0 : 2;
  }
  for (size_t e = 0; e < g->n_edges; ++e) {
    if (edge_alive) edge_alive[e] = g->E[e].alive;
    if (edge_consensus) edge_consensus[e] = g->E[e].cons == D_POS ? 1 : g->E[e].cons == D_NEG ? 0 : 2;
    if (edge_weight) edge_weight[e] = g->E[e].weight;
  }
  return MSGPU_OK;
}

// ---- the graph primitives of the stage on caller-supplied graphs ---------------------------------------------------------
// Same code as above (max_span_tree, connected_components, shortest_path, sort_topologically); exposed so that the vectors
// the reference's own unit tests hold for them (libms/tests/MST_test.cpp, CC_test.cpp, Graph_test.cpp) can be replayed
// through the C-ABI.  consensus: 1 e_POS, 0 e_NEG, 2 e_NONE.

static bool edges_ok(uint32_t n, const uint32_t *a, const uint32_t *b, uint64_t m) {
  if (m && (!a || !b)) return false;
  if (m >= 0x7ffffff0ull) return false;
  for (uint64_t i = 0; i < m; ++i)
    if (a[i] >= n || b[i] >= n) return false;
  return true;
}

int msgpu_graph_max_span_tree(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, const uint64_t *weight,
                              const uint8_t *consensus, uint64_t n_edges, uint8_t *in_tree) {
  if (!edges_ok(n_vertices, a, b, n_edges) || (n_edges && (!weight || !consensus || !in_tree))) return MSGPU_E_ARG;
  try {
    std::vector<uint32_t> cand;
    for (uint64_t e = 0; e < n_edges; ++e)
      if (consensus[e] != 2) cand.push_back(static_cast<uint32_t>(e));
    std::vector<uint8_t> t(n_edges, 0);
    max_span_tree(
        n_vertices, [&](uint32_t e) { return std::make_pair(a[e], b[e]); }, [&](uint32_t e) { return weight[e]; }, cand, t);
    std::copy(t.begin(), t.end(), in_tree);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

int msgpu_graph_connected_components(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, const uint8_t *consensus,
                                     uint64_t n_edges, uint32_t *component, uint32_t *n_components) {
  if (!edges_ok(n_vertices, a, b, n_edges) || (n_edges && !consensus) || (n_vertices && !component) || !n_components)
    return MSGPU_E_ARG;
  try {
    const Csr           adj = build_csr(n_vertices, a, b, n_edges, true);
    std::vector<uint32_t> comp(n_vertices, NIL);
    const auto            comps = connected_components(
        n_vertices, adj, [](uint32_t) { return true; }, [&](const Arc *t) { return consensus[t->e] != 2; },
        [&](uint32_t v) { return comp[v]; }, [&](uint32_t v, uint32_t c) { comp[v] = c; });
    std::copy(comp.begin(), comp.end(), component);
    *n_components = static_cast<uint32_t>(comps.size());
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

int msgpu_graph_shortest_path(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges, int directed,
                              uint32_t src, uint32_t dst, uint32_t *path, uint32_t *n_path) {
  if (!edges_ok(n_vertices, a, b, n_edges) || src >= n_vertices || dst >= n_vertices || !n_path || (*n_path && !path))
    return MSGPU_E_ARG;
  try {
    const Csr                   adj = build_csr(n_vertices, a, b, n_edges, !directed);
    const std::vector<uint32_t> p   = shortest_path(adj, nullptr, n_vertices, src, dst);
    const uint32_t              cap = *n_path;
    *n_path                         = static_cast<uint32_t>(p.size());
    if (p.size() > cap) return MSGPU_E_ARG;
    std::copy(p.begin(), p.end(), path);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

// Graph / DiGraph bookkeeping on its own (include/msgpu.h): a script of deletions and queries over a flat graph built by the
// stage's CSR builder, tombstones as the stage keeps them.
int msgpu_graph_bookkeeping(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges, int directed,
                            const msgpu_graph_op *ops, size_t n_ops, uint32_t *out, size_t out_capacity, size_t *n_out) {
  if (!edges_ok(n_vertices, a, b, n_edges) || (n_ops && !ops) || !n_out || (out_capacity && !out)) return MSGPU_E_ARG;
  try {
    // Graph::addEdge keeps ONE edge per vertex pair (per direction in a DiGraph): Graph.cpp:212-230 (hasEdge first)
    std::vector<uint32_t> ea, eb;
    {
      std::vector<std::pair<uint32_t, uint32_t>> seen;
      for (uint64_t i = 0; i < n_edges; ++i) {
        std::pair<uint32_t, uint32_t> key(a[i], b[i]);
        if (!directed && key.first > key.second) std::swap(key.first, key.second);
        if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
        seen.push_back(key);
        ea.push_back(a[i]);
        eb.push_back(b[i]);
      }
    }
    const size_t         m    = ea.size();
    const Csr            outc = build_csr(n_vertices, ea.data(), eb.data(), m, !directed);
    const Csr            inc  = directed ? build_csr(n_vertices, eb.data(), ea.data(), m, false) : Csr();
    std::vector<uint8_t> ealive(m, 1), valive(n_vertices, 1);
    size_t               w = 0;
    auto push = [&](uint32_t x) {
      if (w < out_capacity) out[w] = x;
      ++w;
    };
    auto edge_of = [&](uint32_t x, uint32_t y) -> int64_t { // Graph::getEdge: undirected either way round, directed x -> y
      const Arc *t = outc.find(x, y);
      return t && ealive[t->e] ? static_cast<int64_t>(t->e) : -1;
    };
    auto alive_e = [&](uint32_t e) { return ealive[e] != 0; };
    for (size_t k = 0; k < n_ops; ++k) {
      const msgpu_graph_op &op = ops[k];
      if (op.op != MSGPU_GOP_ORDER && op.op != MSGPU_GOP_SIZE && op.op != MSGPU_GOP_SUBGRAPH &&
          (op.x >= n_vertices || ((op.op == MSGPU_GOP_DELETE_EDGE || op.op == MSGPU_GOP_HAS_EDGE) && op.y >= n_vertices)))
        return MSGPU_E_ARG;
      switch (op.op) {
      case MSGPU_GOP_DELETE_EDGE: { // Graph::deleteEdge (Graph.cpp:187-210)
        const int64_t e = edge_of(op.x, op.y);
        if (e >= 0) ealive[static_cast<size_t>(e)] = 0;
        break;
      }
      case MSGPU_GOP_DELETE_VERTEX: // Graph::deleteVertex
        if (valive[op.x]) {
          bury_vertex(outc, directed ? &inc : nullptr, op.x, [&](uint32_t e) { ealive[e] = 0; });
          valive[op.x] = 0;
        }
        break;
      case MSGPU_GOP_ORDER: push(static_cast<uint32_t>(std::count(valive.begin(), valive.end(), uint8_t(1)))); break;
      case MSGPU_GOP_SIZE: push(static_cast<uint32_t>(std::count(ealive.begin(), ealive.end(), uint8_t(1)))); break;
      case MSGPU_GOP_HAS_EDGE: push(edge_of(op.x, op.y) >= 0 ? 1u : 0u); break;
      case MSGPU_GOP_NEIGHBORS: // getNeighbors (undirected) / getSuccessors (directed)
      case MSGPU_GOP_PREDECESSORS: {
        if (op.op == MSGPU_GOP_PREDECESSORS && !directed) return MSGPU_E_ARG;
        const size_t at = w;
        push(0);
        uint32_t cnt = 0;
        living_neighbours(op.op == MSGPU_GOP_PREDECESSORS ? inc : outc, op.x, alive_e, [&](uint32_t v) {
          push(v);
          ++cnt;
        });
        if (at < out_capacity) out[at] = cnt;
        break;
      }
      case MSGPU_GOP_IN_DEGREE:
      case MSGPU_GOP_OUT_DEGREE: { // DiGraph::getInDegrees / getOutDegrees of one living vertex
        if (!directed) return MSGPU_E_ARG;
        uint32_t cnt = 0;
        living_neighbours(op.op == MSGPU_GOP_IN_DEGREE ? inc : outc, op.x, alive_e, [&](uint32_t) { ++cnt; });
        push(valive[op.x] ? cnt : 0xffffffffu); // (a deleted vertex has no entry in the reference's degree maps)
        break;
      }
      case MSGPU_GOP_SUBGRAPH: {
        // Graph::getSubgraph (Graph.cpp:317-326): the listed vertices and every living edge between two of them -- the rule
        // by which a connected component becomes its own graph (getDirectedGraph's walk keeps to comp_of == cid).  x = first
        // entry of the vertex list inside `ops` (entries with op MSGPU_GOP_ARG, x = vertex), y = their number.  Emits the
        // order, the size, then the edges as (a, b) pairs in table order.
        if (static_cast<size_t>(op.x) + op.y > n_ops) return MSGPU_E_ARG;
        std::vector<uint8_t> member(n_vertices, 0);
        uint32_t             order = 0;
        for (uint32_t q = 0; q < op.y; ++q) {
          const msgpu_graph_op &arg = ops[op.x + q];
          if (arg.op != MSGPU_GOP_ARG || arg.x >= n_vertices) return MSGPU_E_ARG;
          if (valive[arg.x] && !member[arg.x]) {
            member[arg.x] = 1;
            ++order;
          }
        }
        push(order);
        const size_t at = w;
        push(0);
        uint32_t cnt = 0;
        for (size_t e = 0; e < m; ++e)
          if (ealive[e] && member[ea[e]] && member[eb[e]]) {
            push(ea[e]);
            push(eb[e]);
            ++cnt;
          }
        if (at < out_capacity) out[at] = cnt;
        break;
      }
      case MSGPU_GOP_ARG: break; // (data of a MSGPU_GOP_SUBGRAPH entry)
      default: return MSGPU_E_ARG;
      }
    }
    *n_out = w;
    if (w > out_capacity) return MSGPU_E_ARG;
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

int msgpu_graph_sort_topologically(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges,
                                   uint32_t *order, uint32_t *n_order) {
  if (!edges_ok(n_vertices, a, b, n_edges) || (n_vertices && !order) || !n_order) return MSGPU_E_ARG;
  try {
    const Csr                   succ = build_csr(n_vertices, a, b, n_edges, false);
    const Csr                   pred = build_csr(n_vertices, b, a, n_edges, false);
    const std::vector<uint32_t> o    = sort_topologically(n_vertices, succ, pred, nullptr, nullptr);
    std::copy(o.begin(), o.end(), order);
    *n_order = static_cast<uint32_t>(o.size());
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

} // extern "C"
