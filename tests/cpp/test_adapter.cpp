// Compiles include/msgpu_adapter.hpp against MOCKS of the reference's Graph / MatchMap / Registry interface (same
// member names and argument shapes as include/ms/graph/Graph.h, include/ms/matching/MatchMap.h, include/ms/Registry.h)
// and drives the reference call sequence of src/main.cpp:153-178 through it.
//   test_adapter <paf>          -> prints one line of counts (needs a GPU)
//   test_adapter --errors       -> checks the exception behaviour that needs no GPU
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "msgpu_adapter.hpp"

namespace mock {
struct Vertex {
  Vertex(unsigned id, std::size_t len, std::size_t line) : id(id), len(len), line(line) {}
  unsigned    id;
  std::size_t len, line;
};
struct EdgeOrder {
  Vertex const *startVertex, *endVertex;
  double        leftOffset, rightOffset;
  bool          isContained;
  Vertex const *baseVertex;
  std::size_t   score;
  std::vector<unsigned int> ids;
  bool          direction, isPrimary;
};
struct Edge {
  std::vector<EdgeOrder> orders;
  bool                   shadow = false;
  void appendOrder(EdgeOrder &&o) { orders.push_back(std::move(o)); }
  void setShadow(bool s) { shadow = s; }
};
struct Graph {
  std::unordered_map<unsigned, std::shared_ptr<Vertex>> vertices;
  std::map<std::pair<unsigned, unsigned>, std::unique_ptr<Edge>> edges;
  void addVertex(std::shared_ptr<Vertex> &&v) { vertices.emplace(v->id, std::move(v)); } // first wins (Graph.cpp:148)
  Vertex *getVertex(unsigned id) const { return vertices.at(id).get(); }
  void addEdge(std::pair<Vertex *, Vertex *> const &p) {
    auto &e = edges[{p.first->id, p.second->id}];
    if (!e) e = std::make_unique<Edge>();
  }
  Edge *getEdge(std::pair<Vertex *, Vertex *> const &p) const { return edges.at({p.first->id, p.second->id}).get(); }
};
struct VertexMatch {
  std::pair<int, int> nanoporeRange, illuminaRange;
  double              rRatio;
  bool                direction;
  std::size_t         score;
  bool                isPrimary;
  std::size_t         lineNumber;
};
struct EdgeMatch {
  std::pair<int, int> overlap;
  bool                direction;
  double              score;
  bool                isPrimary;
  std::size_t         lineNumber;
};
struct MatchMap {
  std::size_t nVertexMatches = 0, nEdgeMatches = 0;
  std::map<std::pair<unsigned, unsigned>, std::shared_ptr<VertexMatch>> vm;
  std::map<Edge const *, std::vector<std::pair<unsigned, std::shared_ptr<EdgeMatch>>>> em; // per edge, in insertion order
  void addVertexMatch(unsigned n, unsigned i, std::shared_ptr<VertexMatch> const &m) {
    auto it = vm.find({n, i});
    if (it == vm.end() || it->second->lineNumber > m->lineNumber) { // lowest line wins (MatchMap.cpp:64-80)
      if (it == vm.end()) ++nVertexMatches;
      vm[{n, i}] = m;
    }
  }
  void addEdgeMatch(Edge const *e, unsigned anchor, std::shared_ptr<EdgeMatch> const &m) {
    ++nEdgeMatches;
    em[e].emplace_back(anchor, m);
  }
};
struct Registry {
  std::unordered_map<std::string, unsigned> ids;
  unsigned next = 0;
  unsigned const &operator[](std::string const &s) {
    auto it = ids.find(s);
    if (it == ids.end()) it = ids.emplace(s, next++).first;
    return it->second;
  }
};
} // namespace mock

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (std::strcmp(argv[1], "--errors") == 0) {
    // no GPU needed: construction either works (GPU box) or throws the loud "no HIP device" error
    try {
      msgpu::OverlapCore core(0);
      try {
        core.read("/nonexistent/file.paf");
        std::puts("FAIL: missing file accepted");
        return 1;
      } catch (std::runtime_error const &e) {
        if (std::string(e.what()) != "Can't open blast file.") {
          std::printf("FAIL: %s\n", e.what());
          return 1;
        }
      }
      std::puts("ok gpu");
    } catch (std::runtime_error const &e) {
      if (std::string(e.what()).find("no HIP device") == std::string::npos) {
        std::printf("FAIL: %s\n", e.what());
        return 1;
      }
      std::puts("ok nogpu");
    }
    return 0;
  }
  if (argc >= 6 && std::string(argv[1]) == "--assemble") { // <paf> <unitigs> <nanopore> <outdir>: the whole main()
    try {
      auto const n = msgpu::assemble(argv[2], argv[3], argv[4], argv[5], 4, 300, 0);
      std::printf("{\"rows\": %llu, \"edges\": %llu, \"contraction_edges\": %llu, \"paths\": %llu, \"paths_skipped\": %llu, "
                  "\"contigs\": %llu, \"target_bases\": %llu, \"queries\": %llu}\n",
                  (unsigned long long)n.rows, (unsigned long long)n.edges, (unsigned long long)n.contractionEdges,
                  (unsigned long long)n.paths, (unsigned long long)n.pathsSkipped, (unsigned long long)n.contigs,
                  (unsigned long long)n.targetBases, (unsigned long long)n.queries);
      return 0;
    } catch (std::exception const &e) {
      std::printf("FAIL: %s\n", e.what());
      return 1;
    }
  }
  msgpu::OverlapCore core(0, 300);
  core.read(argv[1]);
  core.calculateEdges();
  core.chainingAndOverlaps();
  auto const t = core.tables();
  { // the same phases as one batched call (msgpu_overlap_batched) must hand back the same tables
    auto const b = core.overlapBatched(3);
    auto same = [](auto const &x, auto const &y) {
      return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0);
    };
    if (!same(b.edges, t.edges) || !same(b.ems, t.ems) || !same(b.orders, t.orders) || !same(b.ids, t.ids) ||
        !same(b.readLength, t.readLength) || !same(b.readFirstLine, t.readFirstLine)) {
      std::puts("FAIL: overlapBatched tables differ from the single-pass tables");
      return 1;
    }
    core.calculateEdges(); // the context is left loaded: rebuild its own tables for what follows
    core.chainingAndOverlaps();
  }
  mock::Graph    graph;
  mock::MatchMap matchMap;
  mock::Registry rn, ri;
  msgpu::fillReferenceObjects<mock::Vertex, mock::VertexMatch, mock::EdgeMatch, mock::EdgeOrder>(core, t, graph, matchMap,
                                                                                                rn, ri);
  if (argc >= 3) { // every object the adapter filled, one line each, for the comparison with the oracle's tables
    auto bits = [](double d) {
      unsigned long long u;
      std::memcpy(&u, &d, 8);
      return u;
    };
    std::FILE *f = std::fopen(argv[2], "w");
    if (!f) return 3;
    for (auto const &kv : graph.vertices) std::fprintf(f, "V %u %zu %zu\n", kv.second->id, kv.second->len, kv.second->line);
    for (auto const &kv : matchMap.vm) {
      auto const &m = *kv.second;
      std::fprintf(f, "VM %u %u %d %d %d %d %016llx %d %zu %d %zu\n", kv.first.first, kv.first.second, m.nanoporeRange.first,
                   m.nanoporeRange.second, m.illuminaRange.first, m.illuminaRange.second, bits(m.rRatio), int(m.direction), m.score,
                   int(m.isPrimary), m.lineNumber);
    }
    for (auto const &kv : graph.edges) {
      std::fprintf(f, "E %u %u %d %zu\n", kv.first.first, kv.first.second, int(kv.second->shadow), kv.second->orders.size());
      auto const it = matchMap.em.find(kv.second.get());
      if (it != matchMap.em.end())
        for (auto const &am : it->second)
          std::fprintf(f, "EM %u %u %u %d %d %d %016llx %d %zu\n", kv.first.first, kv.first.second, am.first, am.second->overlap.first,
                       am.second->overlap.second, int(am.second->direction), bits(am.second->score), int(am.second->isPrimary),
                       am.second->lineNumber);
      std::size_t k = 0;
      for (auto const &o : kv.second->orders) {
        std::fprintf(f, "O %u %u %zu %u %u %u %016llx %016llx %d %zu %d %d", kv.first.first, kv.first.second, k++, o.startVertex->id,
                     o.endVertex->id, o.baseVertex->id, bits(o.leftOffset), bits(o.rightOffset), int(o.isContained), o.score,
                     int(o.direction), int(o.isPrimary));
        for (unsigned id : o.ids) std::fprintf(f, " %u", id);
        std::fputc('\n', f);
      }
    }
    std::fclose(f);
    // the same job on a group of one device (msgpu_group_overlap through RCCL): the merged tables are these tables
    msgpu_group_tables const gt = core.overlapOnDevices({0});
    if (gt.n_edges != t.edges.size() || gt.n_orders != t.orders.size() || gt.n_ids != t.ids.size() ||
        std::memcmp(gt.edges, t.edges.data(), t.edges.size() * sizeof(msgpu_edge)) != 0 ||
        std::memcmp(gt.orders, t.orders.data(), t.orders.size() * sizeof(msgpu_order)) != 0 ||
        std::memcmp(gt.ids, t.ids.data(), t.ids.size() * 4) != 0) {
      std::puts("FAIL: overlapOnDevices({0}) differs from the single-context tables");
      return 1;
    }
  }
  std::size_t orders = 0, shadows = 0, ids = 0;
  for (auto const &kv : graph.edges) {
    orders += kv.second->orders.size();
    shadows += kv.second->shadow;
    for (auto const &o : kv.second->orders) ids += o.ids.size();
  }
  std::printf("{\"vertices\": %zu, \"edges\": %zu, \"vertexmatches\": %zu, \"edgematches\": %zu, \"orders\": %zu, "
              "\"shadows\": %zu, \"ids\": %zu}\n",
              graph.vertices.size(), graph.edges.size(), matchMap.nVertexMatches, matchMap.nEdgeMatches, orders, shadows,
              ids);
  return 0;
}
