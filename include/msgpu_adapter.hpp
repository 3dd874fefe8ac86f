// msgpu_adapter.hpp -- C++ host side above the C-ABI (include/msgpu.h), shaped like the reference's own interface.
//
// Two layers:
//   1. msgpu::OverlapCore       RAII + exceptions around the C-ABI, method names = the reference entry points they
//                               replace (BlastFileReader::read, MatchMap::calculateEdges, chainingAndOverlaps).
//                               Errors are std::runtime_error with the reference's messages
//                               ("Can't open blast file.", "Invalid BLAST file.", "Unexpected nullptr.").
//   2. msgpu::fillReferenceObjects(...)   header-only template that replays the result tables into the reference's
//                               Graph / MatchMap / Registry objects (Graph::addVertex, MatchMap::addVertexMatch,
//                               Graph::addEdge, MatchMap::addEdgeMatch, Edge::appendOrder, Edge::setShadow), so the
//                               phases after src/main.cpp:178 run unchanged.  It is a template over the reference
//                               types, so this header has no dependency on the reference tree (or on GSL).
//
// INTEGRATION.md shows the patch of src/main.cpp that uses it.
#ifndef MSGPU_ADAPTER_HPP
#define MSGPU_ADAPTER_HPP

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "msgpu.h"

namespace msgpu {

// Result tables on the host (see include/msgpu.h for the record semantics and orders).
struct OverlapTables {
  std::vector<msgpu_edge>      edges;
  std::vector<msgpu_edgematch> ems;
  std::vector<msgpu_order>     orders;
  std::vector<std::uint32_t>   ids;
  std::vector<std::int32_t>    readLength;    // Vertex::getNanoporeLength(), by read id
  std::vector<std::uint32_t>   readFirstLine; // Vertex::getMetaDatum<std::size_t>(0), by read id
};

class OverlapCore {
public:
  // Replaces ThreadPool(threadCount) + Graph + MatchMap construction (src/main.cpp:143-148).
  explicit OverlapCore(int device = 0, std::size_t wiggleRoom = 300) {
    msgpu_default_params(&m_params);
    m_params.wiggle_room = wiggleRoom; // Application::getWiggleRoom()
    check(msgpu_create(device, &m_params, &m_ctx), nullptr);
  }
  ~OverlapCore() {
    if (m_paf) msgpu_paf_free(m_paf);
    if (m_ctx) msgpu_destroy(m_ctx);
  }
  OverlapCore(OverlapCore const &)            = delete;
  OverlapCore &operator=(OverlapCore const &) = delete;

  // BlastFileAccessor(path) + BlastFileReader::read() (src/main.cpp:153-156): parse, filter, register ids, then fill the
  // device-resident vertex / VertexMatch store.
  void read(std::string const &pafPath) {
    if (m_paf) {
      msgpu_paf_free(m_paf);
      m_paf = nullptr;
    }
    check(msgpu_parse_paf(pafPath.c_str(), &m_params, &m_paf), nullptr);
    std::size_t      n    = 0;
    msgpu_row const *rows = msgpu_paf_rows(m_paf, &n);
    check(msgpu_load_rows(m_ctx, rows, n), m_ctx);
  }
  // Same from rows the caller already holds (Registry ids in first-line order).
  void addRows(msgpu_row const *rows, std::size_t n) { check(msgpu_load_rows(m_ctx, rows, n), m_ctx); }

  // MatchMap::calculateEdges() (src/main.cpp:157)
  void calculateEdges() { check(msgpu_calculate_edges(m_ctx), m_ctx); }

  // the whole "for edge: Job(chainingAndOverlaps); wg.wait()" phase (src/main.cpp:170-178)
  void chainingAndOverlaps() { check(msgpu_chaining_and_overlaps(m_ctx), m_ctx); }

  // Graph::getOrder() / Graph::getSize() as TRACEd at src/main.cpp:159
  msgpu_counts counts() const {
    msgpu_counts c;
    check(msgpu_get_counts(m_ctx, &c), m_ctx);
    return c;
  }

  OverlapTables tables() const {
    msgpu_counts  c = counts();
    OverlapTables t;
    t.edges.resize(c.n_edges);
    t.ems.resize(c.n_ems);
    t.orders.resize(c.n_orders);
    t.ids.resize(c.n_ids);
    t.readLength.resize(c.n_reads);
    t.readFirstLine.resize(c.n_reads);
    check(msgpu_copy_tables(m_ctx, t.edges.data(), t.ems.data(), t.orders.data(), t.ids.data()), m_ctx);
    check(msgpu_copy_reads(m_ctx, t.readLength.data(), t.readFirstLine.data()), m_ctx);
    return t;
  }

  // Registry reverse look-ups (valid after read())
  char const *readName(std::uint32_t id) const { return msgpu_paf_read_name(m_paf, id); }
  char const *anchorName(std::uint32_t id) const { return msgpu_paf_anchor_name(m_paf, id); }
  msgpu_row const *rows(std::size_t *n) const { return msgpu_paf_rows(m_paf, n); }

  msgpu_ctx *handle() const { return m_ctx; }

private:
  static void check(int rc, msgpu_ctx const *ctx) {
    if (rc == MSGPU_OK) return;
    switch (rc) { // the reference's own exception texts where it has one
    case MSGPU_E_IO: throw std::runtime_error("Can't open blast file.");   // BlastFileAccessor.cpp:44
    case MSGPU_E_FORMAT: throw std::runtime_error("Invalid BLAST file.");  // BlastFileReader.cpp:98
    case MSGPU_E_NUMBER: throw std::invalid_argument("stoi");              // what std::stoi throws
    case MSGPU_E_ARG: throw std::runtime_error("Unexpected nullptr.");     // MatchMap.cpp:56
    default: {
      std::string msg = msgpu_strerror(rc);
      if (ctx && msgpu_last_error(ctx)[0]) msg += std::string(": ") + msgpu_last_error(ctx);
      throw std::runtime_error(msg);
    }
    }
  }

  msgpu_params m_params{};
  msgpu_ctx   *m_ctx = nullptr;
  msgpu_paf   *m_paf = nullptr;
};

// Replay rows + result tables into the reference's containers.  Template parameters are the reference types
// (muchsalsa::graph::Graph, ::Vertex, ::EdgeOrder, muchsalsa::matching::MatchMap, ::VertexMatch, ::EdgeMatch,
// muchsalsa::Registry); nothing here names them, so the header compiles without the reference tree.
//
// Afterwards the reference objects are in the state main() expects at src/main.cpp:180 -- vertices, VertexMatches,
// edges, EdgeMatches, EdgeOrders and shadow flags -- except for hash-map iteration order, which the reference does
// not define either.
template <class Vertex, class VertexMatch, class EdgeMatch, class EdgeOrder, class Graph, class MatchMap, class Registry>
void fillReferenceObjects(OverlapCore const &core, OverlapTables const &t, Graph &graph, MatchMap &matchMap,
                          Registry &registryNanopore, Registry &registryIllumina) {
  std::size_t      nRows = 0;
  msgpu_row const *rows  = core.rows(&nRows);
  // BlastFileReader::parseLine tail (BlastFileReader.cpp:110-126), in line order
  for (std::size_t i = 0; i < nRows; ++i) {
    msgpu_row const &r   = rows[i];
    auto const       nid = registryNanopore[core.readName(r.read_id)];
    auto const       iid = registryIllumina[core.anchorName(r.anchor_id)];
    graph.addVertex(std::make_shared<Vertex>(nid, static_cast<std::size_t>(r.read_len), static_cast<std::size_t>(r.line)));
    auto const rRatio = static_cast<double>(r.i_hi - r.i_lo + 1) / static_cast<double>(r.n_hi - r.n_lo + 1);
    matchMap.addVertexMatch(nid, iid,
                            std::make_shared<VertexMatch>(VertexMatch{
                                std::make_pair(r.n_lo, r.n_hi), std::make_pair(r.i_lo, r.i_hi), rRatio,
                                (r.flags & MSGPU_ROW_DIR) != 0, static_cast<std::size_t>(r.score),
                                (r.flags & MSGPU_ROW_PRIMARY) != 0, static_cast<std::size_t>(r.line)}));
  }
  // MatchMap::processScaffold's effects (MatchMap.cpp:214-218) + chainingAndOverlaps' effects (main.cpp:389-411)
  for (msgpu_edge const &e : t.edges) {
    auto *const v1 = graph.getVertex(e.v1);
    auto *const v2 = graph.getVertex(e.v2);
    graph.addEdge(std::make_pair(v1, v2));
    auto *const pEdge = graph.getEdge(std::make_pair(v1, v2));
    for (std::uint64_t k = e.em_off; k < e.em_off + e.em_cnt; ++k) {
      msgpu_edgematch const &m = t.ems[k];
      matchMap.addEdgeMatch(pEdge, m.anchor_id,
                            std::make_shared<EdgeMatch>(EdgeMatch{std::make_pair(m.ov_lo, m.ov_hi), (m.flags & 1u) != 0,
                                                                  m.score, (m.flags & 2u) != 0,
                                                                  static_cast<std::size_t>(m.line)}));
    }
    pEdge->setShadow(e.shadow != 0);
    for (std::uint64_t k = e.order_off; k < e.order_off + e.order_cnt; ++k) {
      msgpu_order const &o = t.orders[k];
      pEdge->appendOrder(EdgeOrder{graph.getVertex(o.start), graph.getVertex(o.end), o.left_offset, o.right_offset,
                                   (o.flags & MSGPU_ORD_CONTAINED) != 0, graph.getVertex(o.base),
                                   static_cast<std::size_t>(o.score),
                                   std::vector<unsigned int>(t.ids.begin() + static_cast<std::ptrdiff_t>(o.ids_off),
                                                             t.ids.begin() + static_cast<std::ptrdiff_t>(o.ids_off + o.ids_cnt)),
                                   (o.flags & MSGPU_ORD_DIR) != 0, (o.flags & MSGPU_ORD_PRIMARY) != 0});
    }
  }
}

} // namespace msgpu

#endif // MSGPU_ADAPTER_HPP
