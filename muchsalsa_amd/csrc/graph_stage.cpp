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
// The reference keeps shared_ptr vertices/edges in hash maps and fans jobs over a ThreadPool with one global mutex.
// Here the undirected graph is a table of edges plus ordered adjacency maps, built once from the tables that
// msgpu_copy_tables returns; findContractionEdges (the only part with per-edge independent arithmetic) runs on the
// GPU (msgpu_find_contraction_edges) and is an input.  Serial, branchy, pointer-chasing work: host code by design.
//
// Iteration orders the reference leaves to hash containers (or to pointer VALUES: lg.cpp:419, main.cpp:211) are fixed
// as in DESIGN.md section 9: vertices ascending id, edges in creation order ((v1, v2) table order for the undirected
// graph), neighbours ascending id, std::sort ties stable, pointer-ordered containers ordered by vertex id.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <queue>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "msgpu.h"

namespace {

constexpr int    D_NONE = 0, D_POS = 1, D_NEG = -1;
constexpr double BASE_WEIGHT_MULTIPLICATOR = 1.1; // src/main.cpp:96
constexpr double MAX_WEIGHT_MULTIPLICATOR  = 0.8; // src/main.cpp:97

struct GraphError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Tick { // MSGPU_GRAPH_DEBUG=1: phase timings on stderr
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  bool on = std::getenv("MSGPU_GRAPH_DEBUG") != nullptr;
  void operator()(const char *what) {
    if (!on) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[graph] %-28s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

struct Vertex {
  int32_t  length = 0;
  uint32_t meta0  = 0;
  int      direction = D_NONE;
  bool     alive = false;
};

struct UEdge { // an Edge of the undirected Graph
  uint32_t              a = 0, b = 0;
  std::vector<uint32_t> orders; // indices into the order table
  bool                  shadow = false, alive = true;
  uint64_t              weight = 0;
  int                   consensus = D_NONE;
};

struct DEdge { // an Edge of a DiGraph; shared between a DiGraph and its copies (Graph.h:773-774)
  uint32_t              a, b;
  std::vector<uint32_t> orders;
  bool                  shadow = false;
  uint64_t              weight = 0;
  uint32_t              src = 0; // undirected edge whose EdgeMatches it carries (dg.cpp:97-99)
  uint64_t              seq = 0; // creation order
};
using DEdgeP = std::shared_ptr<DEdge>;

struct DiGraph {
  std::set<uint32_t>                               vertices;
  std::map<uint32_t, std::map<uint32_t, DEdgeP>>   succ, pred; // ascending ids
  std::map<uint64_t, DEdgeP>                       edges;      // creation order
  std::map<uint32_t, int64_t>                      indeg, outdeg;
  uint64_t                                         next_seq = 0;

  void add_vertex(uint32_t v) {
    vertices.insert(v);
    indeg.emplace(v, 0);
    outdeg.emplace(v, 0);
  }
  bool   has_vertex(uint32_t v) const { return vertices.count(v) != 0; }
  DEdgeP get_edge(uint32_t a, uint32_t b) const {
    auto it = succ.find(a);
    if (it == succ.end()) return nullptr;
    auto jt = it->second.find(b);
    return jt == it->second.end() ? nullptr : jt->second;
  }
  bool   has_edge(uint32_t a, uint32_t b) const { return get_edge(a, b) != nullptr; }
  DEdgeP add_edge(uint32_t a, uint32_t b) { // GraphBase::_addEdge + DiGraph::_onEdgeAdded
    if (!has_vertex(a) || !has_vertex(b)) return nullptr;
    if (DEdgeP e = get_edge(a, b)) return e;
    DEdgeP e = std::make_shared<DEdge>();
    e->a   = a;
    e->b   = b;
    e->seq = next_seq++;
    succ[a][b] = e;
    pred[b][a] = e;
    edges[e->seq] = e;
    ++outdeg[a];
    ++indeg[b];
    return e;
  }
  void delete_edge(const DEdgeP &e) { // GraphBase::_deleteEdge(.., false)
    auto it = succ.find(e->a);
    if (it != succ.end() && it->second.erase(e->b)) {
      pred[e->b].erase(e->a);
      auto o = outdeg.find(e->a);
      if (o != outdeg.end()) --o->second;
      auto i = indeg.find(e->b);
      if (i != indeg.end()) --i->second;
    }
    edges.erase(e->seq);
  }
  void delete_vertex(uint32_t v) { // GraphBase::_deleteVertex(.., false) + DiGraph::deleteVertex
    std::vector<DEdgeP> gone;
    auto                s = succ.find(v);
    if (s != succ.end())
      for (auto &t : s->second) gone.push_back(t.second);
    auto p = pred.find(v);
    if (p != pred.end())
      for (auto &t : p->second) gone.push_back(t.second);
    for (auto &e : gone) delete_edge(e);
    succ.erase(v);
    pred.erase(v);
    vertices.erase(v);
    indeg.erase(v);
    outdeg.erase(v);
  }
  std::vector<uint32_t> sort_topologically() const { // Graph.cpp:359-395
    std::map<uint32_t, int64_t> nonnull;
    std::vector<uint32_t>       ready, result;
    for (auto &d : indeg) {
      if (d.second > 0) nonnull[d.first] = d.second;
      else ready.push_back(d.first);
    }
    while (!ready.empty()) {
      const uint32_t v = ready.back();
      ready.pop_back();
      auto s = succ.find(v);
      if (s != succ.end())
        for (auto &t : s->second) {
          int64_t &d = nonnull[t.first];
          d -= 1;
          if (d == 0) {
            ready.push_back(t.first);
            nonnull.erase(t.first);
          }
        }
      result.push_back(v);
    }
    return result;
  }
  const std::map<uint32_t, DEdgeP> &successors(uint32_t v) const {
    static const std::map<uint32_t, DEdgeP> none;
    auto                                    it = succ.find(v);
    return it == succ.end() ? none : it->second;
  }
  const std::map<uint32_t, DEdgeP> &predecessors(uint32_t v) const {
    static const std::map<uint32_t, DEdgeP> none;
    auto                                    it = pred.find(v);
    return it == pred.end() ? none : it->second;
  }
};

} // namespace

struct msgpu_graph {
  // tables (copied)
  std::vector<msgpu_edge>      t_edges;
  std::vector<msgpu_edgematch> t_ems;
  std::vector<msgpu_order>     t_orders;
  std::vector<uint32_t>        t_ids;
  // undirected graph
  std::vector<Vertex>                         V;
  std::vector<UEdge>                          E;   // creation order = table order
  std::vector<std::map<uint32_t, uint32_t>>   adj; // neighbour -> edge index
  struct Contain {
    uint32_t nano, direction;
    std::vector<uint32_t> anchors;
  };
  std::map<uint32_t, std::vector<Contain>> contain;
  bool cleaned = false, linearized = false;
  uint32_t n_threads = 1;
  msgpu_graph_stats stats{};
  // paths + storage the msgpu_path_input views point into
  struct PathStore {
    std::vector<msgpu_path_read>    reads;
    std::vector<uint32_t>           order_off, em_off, contain_anchors;
    std::vector<msgpu_path_order>   orders;
    std::vector<msgpu_path_em>      ems;
    std::vector<msgpu_path_contain> contains;
  };
  std::vector<PathStore> paths;
  char err[256] = {0};

  bool odir(uint32_t o) const { return (t_orders[o].flags & MSGPU_ORD_DIR) != 0; }
  bool ocont(uint32_t o) const { return (t_orders[o].flags & MSGPU_ORD_CONTAINED) != 0; }

  int64_t edge_between(uint32_t a, uint32_t b) const {
    auto it = adj[a].find(b);
    return it == adj[a].end() ? -1 : static_cast<int64_t>(it->second);
  }
  void delete_edge(uint32_t e) { // Graph::deleteEdge
    if (!E[e].alive) return;
    adj[E[e].a].erase(E[e].b);
    adj[E[e].b].erase(E[e].a);
    E[e].alive = false;
  }
  void delete_vertex(uint32_t v) { // Graph::deleteVertex
    std::vector<uint32_t> gone;
    for (auto &n : adj[v]) gone.push_back(n.second);
    for (uint32_t e : gone) delete_edge(e);
    V[v].alive = false;
  }
};

namespace {

// mst.cpp:35-73, including unify()'s use of the weights of the vertices rather than of their roots
struct UnionFind { // vertex ids are dense: vectors instead of the reference's two hash maps; weight 0 = "not seen yet"
  std::vector<uint32_t> parent;
  std::vector<uint64_t> weight;
  std::vector<uint32_t> path;
  explicit UnionFind(uint32_t n) : parent(n), weight(n, 0) {}
  uint32_t find(uint32_t v) {
    if (!weight[v]) {
      parent[v] = v;
      weight[v] = 1;
      return v;
    }
    path.assign(1, v);
    uint32_t root = parent[v];
    while (root != path.back()) {
      path.push_back(root);
      root = parent[root];
    }
    for (uint32_t a : path) parent[a] = root;
    return root;
  }
  void unify(uint32_t v1, uint32_t v2) {
    uint32_t first = find(v1), second = find(v2);
    if (weight[v2] > weight[v1]) std::swap(first, second);
    weight[first] += weight[second];
    parent[second] = first;
  }
};

using TreeAdj = std::vector<std::map<uint32_t, uint32_t>>;

// GraphUtil::getShortestPath (Graph.h:927-978: Dijkstra with unit weights, ties by insertion order) on the span tree.
// In a forest the path between two vertices is unique, and (distance, insertion counter) order is breadth-first order,
// so this is a BFS with dense, stamp-reset scratch instead of three hash maps per call.
struct TreeSearch {
  std::vector<uint32_t> from, stamp, queue;
  uint32_t              round = 0;
  explicit TreeSearch(uint32_t n) : from(n), stamp(n, 0) {}
  std::vector<uint32_t> path(const TreeAdj &tree, uint32_t src, uint32_t dst) {
    ++round;
    queue.assign(1, src);
    stamp[src] = round;
    bool found = src == dst;
    for (size_t h = 0; h < queue.size() && !found; ++h) {
      const uint32_t v = queue[h];
      for (auto &n : tree[v]) {
        if (stamp[n.first] == round) continue;
        stamp[n.first] = round;
        from[n.first]  = v;
        if (n.first == dst) {
          found = true;
          break;
        }
        queue.push_back(n.first);
      }
    }
    std::vector<uint32_t> p;
    if (!found) return p;
    for (uint32_t v = dst;; v = from[v]) {
      p.push_back(v);
      if (v == src) break;
    }
    std::reverse(p.begin(), p.end());
    return p;
  }
};

// ---- linearizeGraph, lg.cpp ------------------------------------------------------------------------------------------

void sort_reduction_by_weight(DiGraph &dg) { // lg.cpp:418-520
  std::map<uint32_t, int64_t> nonnull;
  std::deque<uint32_t>        null;
  for (auto &d : dg.indeg) {
    if (d.second > 0) nonnull[d.first] = d.second;
    else null.push_back(d.first);
  }
  std::set<uint32_t> resolved, neighbors;
  if (!nonnull.empty()) neighbors.insert(nonnull.begin()->first);
  while (true) {
    while (!null.empty()) {
      const uint32_t v = null.front();
      null.pop_front();
      resolved.insert(v);
      for (auto &s : dg.successors(v)) {
        auto it = nonnull.find(s.first);
        if (it == nonnull.end()) throw GraphError("sortReductionByWeight: in-degree map out of step");
        if (--it->second == 0) {
          null.push_back(s.first);
          nonnull.erase(it);
          neighbors.erase(s.first);
        } else {
          neighbors.insert(s.first);
        }
      }
    }
    if (nonnull.empty()) break;
    DEdgeP   min_edge;
    uint32_t min_vertex = 0;
    uint64_t min_score  = 0;
    auto     scan       = [&](uint32_t cand) {
      for (auto &p : dg.predecessors(cand))
        if (!resolved.count(p.first) && (!min_edge || p.second->weight < min_score)) {
          min_edge   = p.second;
          min_vertex = cand;
          min_score  = p.second->weight;
        }
    };
    if (neighbors.empty())
      for (auto &kv : nonnull) scan(kv.first);
    else
      for (uint32_t n : neighbors) scan(n);
    if (!min_edge) throw GraphError("sortReductionByWeight: no edge to cut");
    min_edge->shadow = true;
    dg.delete_edge(min_edge);
    auto it = nonnull.find(min_vertex);
    if (--it->second == 0) {
      nonnull.erase(it);
      null.push_back(min_vertex);
      neighbors.erase(min_vertex);
    }
  }
}

using ClusterWeights = std::unordered_map<const DEdge *, uint64_t>;

bool subset(const std::set<size_t> &a, const std::set<size_t> &b) {
  return std::includes(b.begin(), b.end(), a.begin(), a.end());
}

ClusterWeights find_cluster_weights(const DiGraph &dg) { // lg.cpp:144-264
  const std::vector<uint32_t>          order = dg.sort_topologically();
  std::unordered_map<uint32_t, size_t> idx;
  for (size_t i = 0; i < order.size(); ++i) idx[order[i]] = i;
  ClusterWeights result;
  for (auto &e : dg.edges) result[e.second.get()] = 0;
  std::unordered_map<uint32_t, std::set<size_t>> succ, pred;
  for (uint32_t v : order) {
    auto &s = succ[v];
    for (auto &t : dg.successors(v)) s.insert(idx.at(t.first));
    auto &p = pred[v];
    for (auto &t : dg.predecessors(v)) p.insert(idx.at(t.first));
  }
  struct Cand {
    std::set<size_t>    open;
    std::vector<size_t> visited;
  };
  for (uint32_t v : order) {
    std::vector<Cand> cands{Cand{succ.at(v), {idx.at(v)}}};
    for (size_t i_out : succ.at(v)) {
      const uint32_t active = order[i_out];
      for (size_t i_in : pred.at(active))
        for (size_t k = 0; k < cands.size(); ++k) // the bound grows with the emplace_back inside (:193)
          if (cands[k].visited.back() == i_in && cands[k].open.count(i_out)) {
            Cand n;
            std::set_intersection(cands[k].open.begin(), cands[k].open.end(), succ.at(active).begin(),
                                  succ.at(active).end(), std::inserter(n.open, n.open.end()));
            n.visited = cands[k].visited;
            n.visited.push_back(i_out);
            cands.push_back(std::move(n));
          }
      std::vector<Cand> filtered;
      for (size_t a = 0; a < cands.size(); ++a) {
        bool                   dominated = false;
        const std::set<size_t> va(cands[a].visited.begin(), cands[a].visited.end());
        for (size_t b = 0; b < cands.size() && !dominated; ++b) {
          if (a == b || !subset(cands[a].open, cands[b].open)) continue;
          const std::set<size_t> vb(cands[b].visited.begin(), cands[b].visited.end());
          dominated = subset(va, vb);
        }
        if (!dominated) filtered.push_back(cands[a]);
      }
      cands = std::move(filtered);
    }
    std::vector<const std::vector<size_t> *> best;
    size_t                                   best_len = 0;
    for (auto &c : cands) {
      if (c.visited.size() > best_len) {
        best     = {&c.visited};
        best_len = c.visited.size();
      } else if (c.visited.size() == best_len) {
        best.push_back(&c.visited);
      }
    }
    for (auto *mv : best) {
      size_t       c     = mv->size() - 1;
      const size_t limit = std::max<size_t>(mv->size(), 1) - 1;
      for (size_t i = 0; i < limit; ++i) {
        result[dg.get_edge(order[(*mv)[i]], order[(*mv)[i + 1]]).get()] += c;
        c -= 1;
      }
    }
  }
  return result;
}

ClusterWeights find_cluster_weights_heuristic(const DiGraph &dg) { // lg.cpp:72-141
  const std::vector<uint32_t>          order = dg.sort_topologically();
  std::unordered_map<uint32_t, size_t> idx;
  for (size_t i = 0; i < order.size(); ++i) idx[order[i]] = i;
  ClusterWeights result;
  for (auto &e : dg.edges) result[e.second.get()] = 0;
  for (uint32_t v : order) {
    std::set<size_t> sorted_succ;
    for (auto &t : dg.successors(v)) sorted_succ.insert(idx.at(t.first));
    std::map<uint32_t, std::vector<size_t>> cands; // ascending id = the canonical iteration order of :122
    cands[v] = {idx.at(v)};
    for (size_t sid : sorted_succ) {
      const uint32_t      w = order[sid];
      std::vector<size_t> best;
      for (auto &p : dg.predecessors(w)) {
        auto c = cands.find(p.first);
        if (c != cands.end() && c->second.size() > best.size()) best = c->second;
      }
      best.push_back(idx.at(w));
      cands.emplace(w, std::move(best));
    }
    const std::vector<size_t> *best = nullptr;
    for (auto &c : cands)
      if (!best || c.second.size() > best->size()) best = &c.second;
    size_t       c     = best->size() - 1;
    const size_t limit = std::max<size_t>(best->size(), 1) - 1;
    for (size_t i = 0; i < limit; ++i) {
      result[dg.get_edge(order[(*best)[i]], order[(*best)[i + 1]]).get()] += c;
      c -= 1;
    }
  }
  return result;
}

std::vector<std::vector<uint32_t>> extract_paths(DiGraph &dg) { // lg.cpp:347-414
  Tick tick;
  DiGraph cyc = dg;                                              // shallow: edges are shared
  {
    std::vector<DEdgeP> sh;
    for (auto &e : cyc.edges)
      if (e.second->shadow) sh.push_back(e.second);
    for (auto &e : sh) cyc.delete_edge(e);
  }
  tick("copy + drop shadow edges");
  sort_reduction_by_weight(cyc);
  tick("sortReductionByWeight");
  const ClusterWeights cw =
      cyc.vertices.size() < 150000 ? find_cluster_weights(cyc) : find_cluster_weights_heuristic(cyc);
  tick("findClusterWeights");
  std::vector<std::vector<uint32_t>> paths;
  std::unordered_set<uint32_t>       visited;
  // The reference re-sorts the whole remaining graph topologically and re-runs findConservationPathAlt for every path
  // it peels off (lg.cpp:371-407), O(paths x (V + E)) on hash maps.  Same loop here on a dense mirror of `cyc` that
  // only holds vertices which still have an edge: vertices without edges keep their relative place in the stack-based
  // topological order, can only ever offer a one-vertex path, and a path of >= 2 vertices exists while edges exist,
  // so dropping them changes neither the path found nor the tie-breaks among the others.
  struct Arc {
    uint32_t     to;
    const DEdge *e;
  };
  std::vector<uint32_t>                ids; // local -> global, ascending
  std::unordered_map<uint32_t, uint32_t> loc;
  for (uint32_t v : cyc.vertices)
    if (!cyc.successors(v).empty() || !cyc.predecessors(v).empty()) {
      loc.emplace(v, static_cast<uint32_t>(ids.size()));
      ids.push_back(v);
    }
  const uint32_t                n = static_cast<uint32_t>(ids.size());
  std::vector<std::vector<Arc>> succ(n), pred(n); // ascending neighbour (local ids keep the global order)
  size_t                        n_arcs = 0;
  for (uint32_t l = 0; l < n; ++l) {
    for (auto &t : cyc.successors(ids[l])) succ[l].push_back(Arc{loc.at(t.first), t.second.get()});
    for (auto &t : cyc.predecessors(ids[l])) pred[l].push_back(Arc{loc.at(t.first), t.second.get()});
    n_arcs += succ[l].size();
  }
  std::vector<uint32_t> active(n);
  for (uint32_t l = 0; l < n; ++l) active[l] = l;
  std::vector<int64_t>  deg(n);
  std::vector<uint32_t> order, ready, stamp(n, 0);
  std::vector<std::pair<uint64_t, std::vector<uint32_t>>> open(n);
  uint32_t round = 0;
  while (n_arcs > 0) {
    // DiGraph::sortTopologically (Graph.cpp:359-395) over the active vertices
    order.clear();
    ready.clear();
    for (uint32_t v : active) {
      deg[v] = static_cast<int64_t>(pred[v].size());
      if (!deg[v]) ready.push_back(v);
    }
    while (!ready.empty()) {
      const uint32_t v = ready.back();
      ready.pop_back();
      for (const Arc &t : succ[v])
        if (--deg[t.to] == 0) ready.push_back(t.to);
      order.push_back(v);
    }
    // findConservationPathAlt (lg.cpp:267-344); open[] entries are valid when stamp[] == round
    ++round;
    std::vector<uint32_t> final_path;
    auto has_open = [&](uint32_t v) { return stamp[v] == round; };
    auto touch    = [&](uint32_t v) -> std::pair<uint64_t, std::vector<uint32_t>> & { // operator[] of the reference
      if (stamp[v] != round) {
        stamp[v]      = round;
        open[v].first = 0;
        open[v].second.clear();
      }
      return open[v];
    };
    std::vector<std::pair<uint32_t, uint32_t>> max_outs;
    for (uint32_t v : order) {
      if (succ[v].empty()) {
        if (!has_open(v)) {
          if (final_path.empty()) final_path = {v};
        } else {
          if (open[v].second.size() > final_path.size()) final_path = std::move(open[v].second);
          open[v].second.clear();
        }
        continue;
      }
      max_outs.clear();
      uint64_t max_out = 0;
      for (const Arc &t : succ[v]) {
        const uint64_t w = cw.at(t.e);
        if (w > max_out) {
          max_out = w;
          max_outs.clear();
          max_outs.emplace_back(v, t.to);
        } else if (w == max_out) {
          max_outs.emplace_back(v, t.to);
        }
      }
      for (auto &edge : max_outs) {
        const uint32_t nxt = edge.second;
        if (has_open(nxt)) {
          bool take;
          if (open[nxt].first < max_out) take = true;
          else if (open[nxt].first == max_out) take = open[nxt].second.size() < touch(v).second.size() + 1;
          else take = false;
          if (take) {
            std::vector<uint32_t> tmp = touch(v).second;
            tmp.push_back(nxt);
            open[nxt] = {max_out, std::move(tmp)};
          }
        } else if (has_open(v)) {
          std::vector<uint32_t> tmp = open[v].second;
          tmp.push_back(nxt);
          touch(nxt) = {max_out, std::move(tmp)};
        } else {
          touch(nxt) = {max_out, {edge.first, edge.second}};
        }
      }
      touch(v).second.clear();
    }
    if (final_path.empty()) throw GraphError("extractPaths: empty path");
    std::vector<uint32_t> longest;
    for (uint32_t l : final_path) longest.push_back(ids[l]);
    if (longest.size() < 10) {
      bool in_visit = false, out_visit = false;
      for (auto &p : dg.predecessors(longest.front())) in_visit = in_visit || visited.count(p.first);
      for (auto &q : dg.successors(longest.back())) out_visit = out_visit || visited.count(q.first);
      if ((!in_visit && !out_visit) || ((in_visit || out_visit) && longest.size() > 5)) paths.push_back(longest);
    } else {
      paths.push_back(longest);
    }
    for (uint32_t l : final_path) { // diGraphCycle.deleteVertex
      visited.insert(ids[l]);
      for (const Arc &t : succ[l]) {
        auto &pv = pred[t.to];
        pv.erase(std::find_if(pv.begin(), pv.end(), [&](const Arc &a) { return a.to == l; }));
      }
      for (const Arc &t : pred[l]) {
        auto &sv = succ[t.to];
        sv.erase(std::find_if(sv.begin(), sv.end(), [&](const Arc &a) { return a.to == l; }));
        --n_arcs;
      }
      n_arcs -= succ[l].size();
      succ[l].clear();
      pred[l].clear();
    }
    active.erase(std::remove_if(active.begin(), active.end(),
                                [&](uint32_t v) { return succ[v].empty() && pred[v].empty(); }),
                 active.end());
  }
  {
    std::vector<uint32_t> left;
    for (uint32_t v : cyc.vertices)
      if (!visited.count(v)) left.push_back(v);
    tick("conservation paths loop");
    for (uint32_t v : left) paths.push_back({v});
    return paths;
  }
}

std::vector<std::vector<uint32_t>> linearize_graph(DiGraph &dg) { // lg.cpp:522-629
  std::vector<std::vector<uint32_t>>   paths = extract_paths(dg);
  std::vector<size_t>                  color_corr(paths.size()), color_len(paths.size());
  std::unordered_map<uint32_t, size_t> v2idx, v2pos; // v2pos: what std::find returns on the still unjoined paths (:553-556)
  for (size_t i = 0; i < paths.size(); ++i) {
    for (size_t k = 0; k < paths[i].size(); ++k) {
      if (v2idx.emplace(paths[i][k], i).second) v2pos.emplace(paths[i][k], k);
    }
    color_corr[i] = i;
    color_len[i]  = paths[i].size();
  }
  auto index_in = [](const std::vector<uint32_t> &p, uint32_t v) {
    return static_cast<size_t>(std::find(p.begin(), p.end(), v) - p.begin());
  };
  std::vector<std::pair<size_t, DEdgeP>> joins;
  for (auto &kv : dg.edges) {
    const DEdgeP &e = kv.second;
    if (!e->shadow) continue;
    auto i1 = v2idx.find(e->a), i2 = v2idx.find(e->b);
    if (i1 == v2idx.end() || i2 == v2idx.end()) continue;
    const size_t s1 = v2pos.at(e->a), s2 = v2pos.at(e->b);
    const size_t l1_end = color_len[i1->second] - s1 - 1, l2_end = color_len[i2->second] - s2 - 1;
    if (i1->second != i2->second && l1_end < s1 && s2 < l2_end) joins.emplace_back(l1_end + s2, e);
  }
  std::stable_sort(joins.begin(), joins.end(),
                   [](const std::pair<size_t, DEdgeP> &x, const std::pair<size_t, DEdgeP> &y) { return x.first < y.first; });
  for (auto &j : joins) {
    const size_t dist = j.first;
    if (dist > 3) break;
    auto color = [&](size_t i) {
      while (color_corr[i] != i) i = color_corr[i];
      return i;
    };
    const size_t c1 = color(v2idx.at(j.second->a)), c2 = color(v2idx.at(j.second->b));
    if (c1 == c2) continue;
    const size_t i1 = index_in(paths[c1], j.second->a), i2 = index_in(paths[c2], j.second->b);
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

// getDirectedGraph, dg.cpp:35-121; `component` = vertex set of the connected component (its sub-graph keeps every
// alive edge between those vertices, Graph.cpp:317-326)
DiGraph get_directed_graph(msgpu_graph &g, const std::set<uint32_t> &component, uint32_t start) {
  DiGraph                                 dg;
  std::vector<std::pair<uint32_t, bool>> stack{{start, true}};
  while (!stack.empty()) {
    const uint32_t cur    = stack.back().first;
    const bool     toggle = stack.back().second;
    stack.pop_back();
    if (!dg.has_vertex(cur)) dg.add_vertex(cur);
    if (g.V[cur].direction == D_NONE) g.V[cur].direction = toggle ? D_POS : D_NEG;
    for (auto &n : g.adj[cur]) {
      const uint32_t nb = n.first;
      if (!component.count(nb)) continue;
      const UEdge &ne           = g.E[n.second];
      bool         other_exists = dg.has_vertex(nb);
      if (other_exists) other_exists = g.V[nb].direction != D_NONE;
      if (!other_exists) dg.add_vertex(nb);
      if (dg.has_edge(ne.a, ne.b) || dg.has_edge(ne.b, ne.a)) continue;
      for (uint32_t oi : ne.orders) {
        const msgpu_order &o    = g.t_orders[oi];
        bool               flip = false;
        if (!g.odir(oi) && o.base == nb) flip = !flip;
        if (!toggle) flip = !flip;
        const uint32_t s = flip ? o.end : o.start, t = flip ? o.start : o.end;
        DEdgeP         de = dg.get_edge(s, t);
        if (!de) {
          de = dg.add_edge(s, t);
          if (!de) throw GraphError("getDirectedGraph: order between vertices outside the component");
          de->shadow = ne.shadow;
          if (!ne.shadow) de->weight = ne.weight;
          de->src = n.second;
        }
        de->orders.push_back(oi);
      }
      if (ne.consensus == D_NONE) continue;
      const bool nxt = toggle == (ne.consensus == D_POS);
      if (!other_exists) stack.emplace_back(nb, nxt);
    }
  }
  return dg;
}

void require(bool ok, const char *what) {
  if (!ok) throw GraphError(what);
}

} // namespace

extern "C" {

int msgpu_graph_create(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                       const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                       const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads, msgpu_graph **out) {
  if (!out || (n_edges && !edges) || (n_ems && !ems) || (n_orders && !orders) || (n_ids && !ids) ||
      (n_reads && (!read_len || !read_first_line)) || n_edges >= 0xfffffff0ull || n_orders >= 0xfffffff0ull ||
      n_ids >= 0xfffffff0ull)
    return MSGPU_E_ARG;
  *out = nullptr;
  try {
    std::unique_ptr<msgpu_graph> g(new msgpu_graph());
    g->t_edges.assign(edges, edges + n_edges);
    g->t_ems.assign(ems, ems + n_ems);
    g->t_orders.assign(orders, orders + n_orders);
    g->t_ids.assign(ids, ids + n_ids);
    g->V.resize(n_reads);
    g->adj.resize(n_reads);
    for (uint32_t v = 0; v < n_reads; ++v) {
      g->V[v].length = read_len[v];
      g->V[v].meta0  = read_first_line[v];
      g->V[v].alive  = true;
    }
    g->E.resize(n_edges);
    for (uint64_t i = 0; i < n_edges; ++i) {
      const msgpu_edge &e = edges[i];
      if (e.v1 >= n_reads || e.v2 >= n_reads || e.v1 == e.v2 || e.order_off + e.order_cnt > n_orders ||
          e.em_off + e.em_cnt > n_ems)
        return MSGPU_E_ARG;
      UEdge &u = g->E[i];
      u.a      = e.v1;
      u.b      = e.v2;
      u.shadow = e.shadow != 0;
      for (uint32_t k = 0; k < e.order_cnt; ++k) u.orders.push_back(static_cast<uint32_t>(e.order_off + k));
      if (!g->adj[e.v1].emplace(e.v2, static_cast<uint32_t>(i)).second) return MSGPU_E_ARG; // duplicate edge
      g->adj[e.v2].emplace(e.v1, static_cast<uint32_t>(i));
    }
    for (uint64_t k = 0; k < n_orders; ++k) {
      const msgpu_order &o = orders[k];
      if (o.start >= n_reads || o.end >= n_reads || o.base >= n_reads || o.ids_off + o.ids_cnt > n_ids) return MSGPU_E_ARG;
    }
    g->stats.n_vertices_in = n_reads;
    g->stats.n_edges_in    = n_edges;
    *out                   = g.release();
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

void        msgpu_graph_free(msgpu_graph *g) { delete g; }
const char *msgpu_graph_last_error(const msgpu_graph *g) { return g ? g->err : "null graph"; }

// src/main.cpp:194-288 after findContractionEdges
int msgpu_graph_clean_up(msgpu_graph *g, const int64_t *contraction_order, const msgpu_row *rows, size_t n_rows) {
  if (!g || (!g->E.empty() && !contraction_order) || (n_rows && !rows)) return MSGPU_E_ARG;
  if (g->cleaned) return MSGPU_E_STATE;
  g->err[0] = 0;
  try {
    Tick tick;
    std::vector<uint64_t> has_vm(n_rows); // MatchMap::getVertexMatch(read, anchor) != nullptr: sorted (read, anchor) keys
    for (size_t i = 0; i < n_rows; ++i) has_vm[i] = (static_cast<uint64_t>(rows[i].read_id) << 32) | rows[i].anchor_id;
    std::sort(has_vm.begin(), has_vm.end());
    std::vector<uint32_t> contraction; // order indices, in edge order
    for (size_t e = 0; e < g->E.size(); ++e) {
      const int64_t k = contraction_order[e];
      if (k < 0) continue;
      if (static_cast<uint64_t>(k) >= g->t_orders.size() || g->t_orders[k].edge_idx != e) return MSGPU_E_ARG;
      contraction.push_back(static_cast<uint32_t>(k));
    }
    g->stats.n_contraction_edges = contraction.size();
    const uint32_t        nv = static_cast<uint32_t>(g->V.size());
    std::vector<uint32_t> targets(nv);
    for (uint32_t v = 0; v < nv; ++v) targets[v] = v; // :194-197
    for (uint32_t k : contraction) {                  // findContractionTargets, :465-482
      const msgpu_order &o  = g->t_orders[k];
      const uint32_t     to = targets[o.end];
      if (targets[o.start] == o.start || g->V[targets[o.start]].meta0 > g->V[to].meta0) targets[o.start] = to;
    }
    std::set<uint32_t> deletable, roots;
    for (uint32_t k : contraction) { // findDeletableVertices, :484-507
      const msgpu_order &o = g->t_orders[k];
      deletable.insert(o.start);
      roots.insert(targets[o.start]);
      roots.erase(o.start);
    }
    for (uint32_t k : contraction) { // contract, :509-531
      const msgpu_order &o = g->t_orders[k];
      if (!roots.count(o.end)) continue;
      msgpu_graph::Contain c;
      c.nano      = o.start;
      c.direction = g->odir(k) ? 1u : 0u;
      for (uint32_t i = 0; i < o.ids_cnt; ++i) {
        const uint32_t a = g->t_ids[o.ids_off + i];
        if (!rows || std::binary_search(has_vm.begin(), has_vm.end(), (static_cast<uint64_t>(o.start) << 32) | a))
          c.anchors.push_back(a);
      }
      g->contain[o.end].push_back(std::move(c));
      ++g->stats.n_contain_elements;
    }
    tick("contraction bookkeeping");
    for (uint32_t v : deletable) g->delete_vertex(v); // :242-244
    g->stats.n_deleted_vertices = deletable.size();
    for (uint32_t e = 0; e < g->E.size(); ++e) { // findDeletableEdges, :534-549 (+ deletion :258-260)
      UEdge &u = g->E[e];
      if (!u.alive) continue;
      std::vector<uint32_t> kept;
      for (uint32_t oi : u.orders)
        if (!g->ocont(oi)) kept.push_back(oi);
      u.orders = std::move(kept);
      if (u.orders.empty()) g->delete_edge(e);
    }
    std::vector<uint32_t> edges; // :264
    for (uint32_t e = 0; e < g->E.size(); ++e)
      if (g->E[e].alive) edges.push_back(e);
    for (uint32_t e : edges) { // computeBitweight, :551-573
      UEdge &u = g->E[e];
      if (u.orders.empty()) continue;
      const bool d0 = g->odir(u.orders[0]);
      if (u.shadow) {
        bool other = false;
        for (uint32_t oi : u.orders) other = other || g->odir(oi) != d0;
        if (!other) u.consensus = d0 ? D_POS : D_NEG;
      } else {
        u.weight    = g->t_orders[u.orders[0]].score;
        u.consensus = d0 ? D_POS : D_NEG;
      }
    }
    tick("deletions + bitweight");
    // getMaxSpanTree, mst.cpp:75-111
    std::vector<uint32_t> cand;
    for (uint32_t e : edges)
      if (g->E[e].consensus != D_NONE) cand.push_back(e);
    std::stable_sort(cand.begin(), cand.end(), [&](uint32_t x, uint32_t y) { return g->E[x].weight > g->E[y].weight; });
    UnionFind uf(nv);
    TreeAdj   tree(nv);
    for (uint32_t e : cand) {
      const UEdge &u = g->E[e];
      if (uf.find(u.a) != uf.find(u.b)) {
        tree[u.a][u.b] = e;
        tree[u.b][u.a] = e;
        uf.unify(u.a, u.b);
      }
    }
    tick("span tree");
    std::set<uint32_t> dele;
    TreeSearch         search(nv);
    for (uint32_t e : edges) { // decycle, :575-618
      const UEdge &u = g->E[e];
      if (u.consensus == D_NONE || tree[u.a].count(u.b)) continue;
      const std::vector<uint32_t> path = search.path(tree, u.a, u.b);
      require(!path.empty(), "decycle: the span tree does not connect the ends of an edge");
      bool                direction = u.consensus == D_POS;
      std::vector<double> weights;
      for (size_t i = 0; i + 1 < path.size(); ++i) {
        const int64_t pe = g->edge_between(path[i], path[i + 1]);
        require(pe >= 0, "decycle: tree edge missing from the graph");
        direction = direction == (g->E[pe].consensus == D_POS);
        weights.push_back(static_cast<double>(g->E[pe].weight));
      }
      if (!direction && !weights.empty()) {
        const auto   lo = std::min_element(weights.begin(), weights.end());
        const auto   hi = std::max_element(weights.begin(), weights.end());
        const double base = static_cast<double>(u.weight);
        if (*lo < base || (base * BASE_WEIGHT_MULTIPLICATOR >= *lo && *lo < *hi * MAX_WEIGHT_MULTIPLICATOR)) {
          const size_t i = static_cast<size_t>(lo - weights.begin());
          dele.insert(static_cast<uint32_t>(g->edge_between(path[i], path[i + 1])));
        }
        dele.insert(e);
      }
    }
    tick("decycle");
    for (uint32_t e : dele) g->delete_edge(e); // :285-287
    g->stats.n_decycled_edges = dele.size();
    uint64_t nv_alive = 0, ne_alive = 0;
    for (auto &v : g->V) nv_alive += v.alive;
    for (auto &e : g->E) ne_alive += e.alive;
    g->stats.n_vertices = nv_alive;
    g->stats.n_edges    = ne_alive;
    g->cleaned          = true;
  } catch (std::bad_alloc const &) {
    return MSGPU_E_NOMEM;
  } catch (std::exception const &e) { // GraphError and anything a container throws
    snprintf(g->err, sizeof(g->err), "%s", e.what());
    return MSGPU_E_LAYOUT;
  }
  return MSGPU_OK;
}

// getConnectedComponents (cc.cpp:33-70) + per component getDirectedGraph + linearizeGraph (src/main.cpp:300-310, 620-661)
} // extern "C"

namespace {

// one connected component: getDirectedGraph + linearizeGraph + the assemblePath inputs of its paths (main.cpp:620-661)
std::vector<msgpu_graph::PathStore> component_paths(msgpu_graph *g, const std::vector<uint32_t> &comp) {
  const std::set<uint32_t> cset(comp.begin(), comp.end());
  uint32_t                 start = *cset.begin();
  for (uint32_t v : cset) // std::max_element: the first of the longest, vertices ascending
    if (g->V[v].length > g->V[start].length) start = v;
  Tick    tk;
  DiGraph dg = get_directed_graph(*g, cset, start);
  tk("getDirectedGraph");
  const std::vector<std::vector<uint32_t>> lin = linearize_graph(dg);
  tk("linearizeGraph");
  std::vector<msgpu_graph::PathStore> out;
  for (const std::vector<uint32_t> &p : lin) {
    msgpu_graph::PathStore ps;
    ps.order_off.push_back(0);
    ps.em_off.push_back(0);
    for (uint32_t v : p) {
      msgpu_path_read r{};
      r.read_id         = v;
      r.direction       = g->V[v].direction == D_POS ? 1u : g->V[v].direction == D_NEG ? 0u : 2u;
      r.nanopore_length = static_cast<uint64_t>(g->V[v].length);
      ps.reads.push_back(r);
    }
    for (size_t i = 0; i + 1 < p.size(); ++i) {
      const DEdgeP de = dg.get_edge(p[i], p[i + 1]);
      require(de != nullptr, "path edge missing in the directed graph");
      for (uint32_t oi : de->orders) {
        const msgpu_order &o = g->t_orders[oi];
        msgpu_path_order   po{};
        po.score     = o.score;
        po.base_read = o.base;
        po.ids_off   = static_cast<uint32_t>(o.ids_off);
        po.ids_cnt   = o.ids_cnt;
        ps.orders.push_back(po);
      }
      const msgpu_edge &te = g->t_edges[de->src];
      for (uint32_t k = 0; k < te.em_cnt; ++k) {
        const msgpu_edgematch &m = g->t_ems[te.em_off + k];
        ps.ems.push_back(msgpu_path_em{m.anchor_id, m.ov_lo, m.ov_hi});
      }
      ps.order_off.push_back(static_cast<uint32_t>(ps.orders.size()));
      ps.em_off.push_back(static_cast<uint32_t>(ps.ems.size()));
    }
    for (uint32_t v : p) {
      auto c = g->contain.find(v);
      if (c == g->contain.end()) continue;
      for (const msgpu_graph::Contain &ce : c->second) {
        msgpu_path_contain pc{};
        pc.host_read   = v;
        pc.nano        = ce.nano;
        pc.direction   = ce.direction;
        pc.anchors_off = static_cast<uint32_t>(ps.contain_anchors.size());
        pc.anchors_cnt = static_cast<uint32_t>(ce.anchors.size());
        ps.contain_anchors.insert(ps.contain_anchors.end(), ce.anchors.begin(), ce.anchors.end());
        ps.contains.push_back(pc);
      }
    }
    out.push_back(std::move(ps));
  }
  return out;
}

} // namespace

extern "C" {

// host threads for the per-component work of msgpu_graph_linearize (default 1; the reference runs one assemblePaths job
// per component on its ThreadPool, src/main.cpp:300-310)
int msgpu_graph_set_threads(msgpu_graph *g, uint32_t n_threads) {
  if (!g) return MSGPU_E_ARG;
  g->n_threads = n_threads ? n_threads : 1;
  return MSGPU_OK;
}

int msgpu_graph_linearize(msgpu_graph *g) {
  if (!g) return MSGPU_E_ARG;
  if (!g->cleaned || g->linearized) return MSGPU_E_STATE;
  g->err[0] = 0;
  try {
    // getConnectedComponents, cc.cpp:33-70
    const uint32_t                     nv = static_cast<uint32_t>(g->V.size());
    std::vector<bool>                  visited(nv, false);
    std::vector<std::vector<uint32_t>> comps;
    for (uint32_t s = 0; s < nv; ++s) {
      if (!g->V[s].alive || visited[s]) continue;
      std::vector<uint32_t> comp{s};
      std::deque<uint32_t>  queue{s};
      visited[s] = true;
      while (!queue.empty()) {
        const uint32_t cur = queue.front();
        queue.pop_front();
        for (auto &n : g->adj[cur])
          if (!visited[n.first] && g->E[n.second].consensus != D_NONE) {
            comp.push_back(n.first);
            queue.push_back(n.first);
            visited[n.first] = true;
          }
      }
      comps.push_back(std::move(comp));
    }
    g->stats.n_components = comps.size();
    // Components are independent (a component only orients and reads its own vertices): largest first on the worker
    // threads, results appended in component order -- the order a single-threaded reference run assembles them in.
    std::vector<std::vector<msgpu_graph::PathStore>> per(comps.size());
    std::vector<std::string>                         errs(comps.size());
    std::vector<int>                                 rcs(comps.size(), MSGPU_OK);
    std::vector<size_t>                              by_size(comps.size());
    for (size_t i = 0; i < by_size.size(); ++i) by_size[i] = i;
    std::stable_sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return comps[x].size() > comps[y].size(); });
    std::atomic<size_t> next{0};
    auto                work = [&]() {
      for (size_t k = next.fetch_add(1); k < by_size.size(); k = next.fetch_add(1)) {
        const size_t i = by_size[k];
        try {
          per[i] = component_paths(g, comps[i]);
        } catch (std::bad_alloc const &) { rcs[i] = MSGPU_E_NOMEM; } catch (std::exception const &e) {
          rcs[i]  = MSGPU_E_LAYOUT;
          errs[i] = e.what();
        }
      }
    };
    uint32_t nt = g->n_threads;
    if (nt > comps.size()) nt = static_cast<uint32_t>(comps.size() ? comps.size() : 1);
    std::vector<std::thread> pool;
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
int msgpu_graph_path_input(const msgpu_graph *g, uint32_t i, msgpu_path_input *out) {
  if (!g || !out || i >= g->paths.size()) return MSGPU_E_ARG;
  const msgpu_graph::PathStore &p = g->paths[i];
  std::memset(out, 0, sizeof(*out));
  out->reads           = p.reads.data();
  out->n_reads         = static_cast<uint32_t>(p.reads.size());
  out->asm_idx         = static_cast<int32_t>(i);
  out->order_off       = p.order_off.data();
  out->orders          = p.orders.data();
  out->ids             = g->t_ids.data();
  out->em_off          = p.em_off.data();
  out->ems             = p.ems.data();
  out->contains        = p.contains.data();
  out->n_contains      = static_cast<uint32_t>(p.contains.size());
  out->contain_anchors = p.contain_anchors.data();
  return MSGPU_OK;
}

// alive[v] (n_reads entries, optional) = vertex still in the graph; edge_alive[e] (optional) likewise; direction[v]
// (optional) = Vertex::getVertexDirection() as 1 / 0 / 2 (e_POS / e_NEG / e_NONE)
int msgpu_graph_state(const msgpu_graph *g, uint8_t *vertex_alive, uint8_t *vertex_direction, uint8_t *edge_alive,
                      uint8_t *edge_consensus, uint64_t *edge_weight) {
  if (!g) return MSGPU_E_ARG;
  for (size_t v = 0; v < g->V.size(); ++v) {
    if (vertex_alive) vertex_alive[v] = g->V[v].alive;
    if (vertex_direction) vertex_direction[v] = g->V[v].direction == D_POS ? 1 : g->V[v].direction == D_NEG ? 0 : 2;
  }
  for (size_t e = 0; e < g->E.size(); ++e) {
    if (edge_alive) edge_alive[e] = g->E[e].alive;
    if (edge_consensus) edge_consensus[e] = g->E[e].consensus == D_POS ? 1 : g->E[e].consensus == D_NEG ? 0 : 2;
    if (edge_weight) edge_weight[e] = g->E[e].weight;
  }
  return MSGPU_OK;
}

} // extern "C"
