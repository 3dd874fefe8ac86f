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
//   3. msgpu::assemble(...)     the whole of main() (src/main.cpp:130-322) over the C-ABI: PAF + unitigs + long reads ->
//                               the three output files, for builds that drop the reference's own phases altogether.
//
// INTEGRATION.md shows the patch of src/main.cpp that uses it.
#ifndef MSGPU_ADAPTER_HPP
#define MSGPU_ADAPTER_HPP

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <unordered_map>
#include <exception>
#include <stdexcept>
#include <string>
#include <future>
#include <thread>
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
    if (m_group) msgpu_group_destroy(m_group);
    if (m_ctx) msgpu_destroy(m_ctx);
  }
  OverlapCore(OverlapCore const &)            = delete;
  OverlapCore &operator=(OverlapCore const &) = delete;

  // BlastFileAccessor(path) + BlastFileReader::read() (src/main.cpp:153-156): parse, filter, register ids, then fill the
  // device-resident vertex / VertexMatch store.
  void read(std::string const &pafPath) {
    parse(pafPath);
    std::size_t      n    = 0;
    msgpu_row const *rows = msgpu_paf_rows(m_paf, &n);
    check(msgpu_load_rows(m_ctx, rows, n), m_ctx);
  }
  // The host half of read() alone (parse, filter, Registry ids): for callers that go on with overlapBatched() /
  // overlapResident(), which carry the rows to HBM themselves.
  void parse(std::string const &pafPath) {
    if (m_paf) {
      msgpu_paf_free(m_paf);
      m_paf = nullptr;
    }
    check(msgpu_parse_paf(pafPath.c_str(), &m_params, &m_paf), nullptr);
    check(msgpu_set_id_space(m_ctx, msgpu_paf_read_count(m_paf), msgpu_paf_anchor_count(m_paf)), m_ctx);
  }
  // Same from rows the caller already holds (Registry ids in first-line order).
  void addRows(msgpu_row const *rows, std::size_t n) {
    check(msgpu_set_id_space(m_ctx, 0, 0), m_ctx);
    check(msgpu_load_rows(m_ctx, rows, n), m_ctx);
  }

  // MatchMap::calculateEdges() (src/main.cpp:157)
  void calculateEdges() { check(msgpu_calculate_edges(m_ctx), m_ctx); }

  // the whole "for edge: Job(chainingAndOverlaps); wg.wait()" phase (src/main.cpp:170-178)
  void chainingAndOverlaps() { check(msgpu_chaining_and_overlaps(m_ctx), m_ctx); }

  // read() must have run.  MatchMap::calculateEdges() + the chainingAndOverlaps fan-out as ONE msgpu_overlap_batched
  // call (the ThreadPool / WaitGroup phases of src/main.cpp:153-178 as windows of owner reads on two HIP streams): the
  // tables arrive in host memory while later windows still compute.  Same tables as calculateEdges() +
  // chainingAndOverlaps() + tables(), bit for bit.
  OverlapTables overlapBatched(unsigned nBatches = 0) {
    std::size_t      n    = 0;
    msgpu_row const *rows = msgpu_paf_rows(m_paf, &n);
    msgpu_host_tables h;
    check(msgpu_overlap_batched(m_ctx, rows, n, nBatches, &h), m_ctx);
    OverlapTables t;
    t.edges.assign(h.edges, h.edges + h.n_edges);
    t.ems.assign(h.ems, h.ems + h.n_ems);
    t.orders.assign(h.orders, h.orders + h.n_orders);
    t.ids.assign(h.ids, h.ids + h.n_ids);
    t.readLength.assign(h.read_len, h.read_len + h.n_reads);
    t.readFirstLine.assign(h.read_first_line, h.read_first_line + h.n_reads);
    return t;
  }

  // The same with the job's tables kept whole in HBM and the EdgeMatch table left there (msgpu_overlap_batched_ex,
  // MSGPU_BATCH_NO_EDGEMATCHES): edges / orders / ids / Vertex facts arrive as VIEWS of the context's pinned result memory
  // (valid until the next overlap call on this object); findContractionEdges() and edgeMatchesOf() then work on the
  // resident tables.  h.ems is null.
  msgpu_host_tables overlapResident(unsigned nBatches = 0) {
    std::size_t      n    = 0;
    msgpu_row const *rows = msgpu_paf_rows(m_paf, &n);
    msgpu_host_tables h;
    check(msgpu_overlap_batched_ex(m_ctx, rows, n, nBatches, MSGPU_BATCH_NO_EDGEMATCHES, &h), m_ctx);
    return h;
  }
  // MatchMap::getEdgeMatches(edge) for a list of edge-table indices, from the EdgeMatch table in HBM (views, valid until the
  // next call): the EdgeMatches of edgeIdx[i] are ems[off[i] .. off[i + 1])
  void edgeMatchesOf(std::uint32_t const *edgeIdx, std::size_t n, std::uint64_t const **off, msgpu_edgematch const **ems) {
    check(msgpu_get_edgematches(m_ctx, edgeIdx, n, off, ems), m_ctx);
  }

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
  // findContractionEdges fan-out (src/main.cpp:183-190): per edge the index of its contraction order, or -1
  std::vector<std::int64_t> findContractionEdges() {
    std::vector<std::int64_t> out(counts().n_edges, -1);
    check(msgpu_find_contraction_edges(m_ctx, nullptr, 0, nullptr, 0, 0, out.data()), m_ctx);
    return out;
  }
  // SequenceAccessor::_build*Idx (SequenceAccessor.cpp:143-231): Registry::operator[] per record of a parsed sequence file,
  // on the registries read() / parse() filled -- names the PAF registered keep their id, unknown names take the next free
  // ids in file order.  kind 0 = reads, 1 = unitigs.
  std::vector<std::uint32_t> registerSequences(int kind, msgpu_seqfile const *f, std::uint32_t *idSpace) {
    std::vector<std::uint32_t> ids(msgpu_seq_count(f));
    check(msgpu_paf_register_sequences(m_paf, kind, f, ids.data(), idSpace), nullptr);
    return ids;
  }
  std::uint32_t readCount() const { return msgpu_paf_read_count(m_paf); }
  std::uint32_t anchorCount() const { return msgpu_paf_anchor_count(m_paf); }
  char const *readName(std::uint32_t id) const { return msgpu_paf_read_name(m_paf, id); }
  char const *anchorName(std::uint32_t id) const { return msgpu_paf_anchor_name(m_paf, id); }
  msgpu_row const *rows(std::size_t *n) const { return msgpu_paf_rows(m_paf, n); }

  msgpu_ctx *handle() const { return m_ctx; }

  // The node's GPUs behind the same call site (src/main.cpp:143-178: one process, the phase closed by WaitGroup::wait()):
  // parse() must have run.  Every device in `devices` takes the row table, builds the index and computes the edges with
  // v1 % n == its position; ONE grouped RCCL all-gather over xGMI + the merge on every device give the job's merged edge
  // list (msgpu_group_overlap).  Views of the group's pinned result memory, valid until the next call; the EdgeMatches of an
  // edge stay with its owner: msgpu_get_edgematches(msgpu_group_ctx(group(), v1 % n), ...).
  msgpu_group_tables overlapOnDevices(std::vector<int> const &devices) {
    if (!m_group || m_groupDevices != devices) {
      if (m_group) msgpu_group_destroy(m_group);
      m_group = nullptr;
      check(msgpu_group_create(devices.data(), static_cast<int>(devices.size()), &m_params, &m_group), nullptr);
      m_groupDevices = devices;
    }
    std::size_t        n    = 0;
    msgpu_row const   *rows = msgpu_paf_rows(m_paf, &n);
    msgpu_group_tables t;
    int const          rc = msgpu_group_overlap(m_group, rows, n, &t);
    if (rc != MSGPU_OK) throw std::runtime_error(std::string(msgpu_strerror(rc)) + ": " + msgpu_group_last_error(m_group));
    return t;
  }
  msgpu_group *group() const { return m_group; }

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
  msgpu_group *m_group = nullptr;
  std::vector<int> m_groupDevices;
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

// ---- 3. the whole executable ------------------------------------------------------------------------------------------

struct AssemblyCounts {
  std::uint64_t rows = 0, reads = 0, edges = 0, orders = 0, contractionEdges = 0, paths = 0, pathsSkipped = 0,
                contigs = 0, targetBases = 0, queries = 0;
};

namespace detail {
inline void require(int rc, char const *what, char const *detail = nullptr) {
  if (rc == MSGPU_OK) return;
  std::string msg = std::string(what) + ": " + msgpu_strerror(rc);
  if (detail && *detail) msg += std::string(" (") + detail + ")";
  throw std::runtime_error(msg);
}
} // namespace detail

// main() of the reference (src/main.cpp:130-322): contigs PAF, unitig FASTA, long-read FASTA/FASTQ -> outDir/temp_1.*
inline AssemblyCounts assemble(std::string const &contigsPaf, std::string const &unitigsPath,
                               std::string const &nanoporePath, std::string const &outDir, unsigned threads = 1,
                               std::size_t wiggleRoom = 300, int device = 0) {
  AssemblyCounts n;
  OverlapCore    core(device, wiggleRoom);

  struct Seq { // :161-163 -- the two sequence files need nothing from the PAF: one thread per file, started before the PAF
               // is read, beside the parser, the GPU and the graph stage.  A file's bytes travel to HBM while it is parsed
               // (msgpu_seq_parse_upload: page-locked ring, the host never holds the bases; one stream per kind) and are
               // converted to the 2-bit form there; the Registry ids of the records follow once the PAF is read.
    msgpu_seqctx   *ctx = nullptr;
    msgpu_seqfile  *fn = nullptr, *fi = nullptr;
    msgpu_assembly *as = nullptr;
    ~Seq() {
      msgpu_assembly_free(as);
      msgpu_seq_free(fn);
      msgpu_seq_free(fi);
      msgpu_seq_destroy(ctx);
    }
  } s;
  detail::require(msgpu_seq_create(device, &s.ctx), "msgpu_seq_create");
  std::exception_ptr loadError[2];
  auto               load = [&](int kind, std::string const &path, int isFastq, msgpu_seqfile **f, char const *what) {
    try {
      detail::require(msgpu_seq_parse_upload(s.ctx, kind, path.c_str(), isFastq, f), what, msgpu_seq_last_error(s.ctx));
      detail::require(msgpu_seq_pack_store(s.ctx, kind), "msgpu_seq_pack_store", msgpu_seq_last_error(s.ctx)); // 2 bits per base in HBM
    } catch (...) { loadError[kind] = std::current_exception(); }
  };
  // Registry::operator[] for the records of both files (SequenceAccessor.cpp:171,215) needs the PAF's registries: a third
  // thread waits for the two loaders and for the PAF, then registers the records and hands the ids to the stores -- beside
  // the overlap and graph stages, which read neither.
  std::promise<bool> pafParsed;
  std::exception_ptr regError;
  std::thread        loaders[2] = {std::thread([&]() { load(0, nanoporePath, -1, &s.fn, "nanopore file"); }),
                                   std::thread([&]() { load(1, unitigsPath, 0, &s.fi, "unitig file"); })};
  std::thread        registrar([&, parsed = pafParsed.get_future()]() mutable {
    for (auto &t : loaders) t.join();
    try {
      if (!parsed.get()) return; // (the PAF could not be read: nothing to register)
      for (auto &e : loadError)
        if (e) std::rethrow_exception(e);
      std::uint32_t                    readSpace = 0, anchorSpace = 0;
      std::vector<std::uint32_t> const readIds = core.registerSequences(0, s.fn, &readSpace), anchorIds = core.registerSequences(1, s.fi, &anchorSpace);
      detail::require(msgpu_seq_set_ids(s.ctx, 0, s.fn, readIds.data(), readSpace), "ids of the reads", msgpu_seq_last_error(s.ctx));
      detail::require(msgpu_seq_set_ids(s.ctx, 1, s.fi, anchorIds.data(), anchorSpace), "ids of the unitigs",
                      msgpu_seq_last_error(s.ctx));
    } catch (...) { regError = std::current_exception(); }
  });
  struct Joiner { // (whatever happens below, the three threads are over before their captures go)
    std::promise<bool> &parsed;
    std::thread        &t;
    bool                told = false;
    void tell(bool ok) {
      if (!told) parsed.set_value(ok);
      told = true;
    }
    ~Joiner() {
      tell(false);
      if (t.joinable()) t.join();
    }
  } joiner{pafParsed, registrar};

  core.parse(contigsPaf); // :153-156 (the rows travel to HBM inside overlapResident below)
  joiner.tell(true);

  // :157 + :170-178 -- calculateEdges and the chainingAndOverlaps fan-out as windows of owner reads on two HIP streams (the
  // ThreadPool replacement); the tables arrive in pinned host memory while later windows compute, the EdgeMatch table
  // stays in HBM
  msgpu_host_tables const t = core.overlapResident();
  auto const contraction    = core.findContractionEdges(); // :183-190, on the resident tables
  std::size_t      nRows = 0;
  msgpu_row const *rows  = core.rows(&nRows);
  n.rows = nRows, n.reads = t.n_reads, n.edges = t.n_edges, n.orders = t.n_orders;
  for (auto c : contraction) n.contractionEdges += c >= 0;

  struct Graph { // :194-310
    msgpu_graph *g = nullptr;
    ~Graph() { msgpu_graph_free(g); }
  } graph;
  detail::require(msgpu_graph_create_borrowed(t.edges, t.n_edges, nullptr, 0, t.orders, t.n_orders, t.ids, t.n_ids,
                                              t.read_len, t.read_first_line, t.n_reads, &graph.g),
                  "msgpu_graph_create_borrowed"); // (views of core's pinned memory: core outlives graph)
  detail::require(msgpu_graph_clean_up(graph.g, contraction.data(), rows, nRows), "msgpu_graph_clean_up",
                  msgpu_graph_last_error(graph.g));
  msgpu_graph_set_threads(graph.g, threads ? threads : 1);
  detail::require(msgpu_graph_linearize(graph.g), "msgpu_graph_linearize", msgpu_graph_last_error(graph.g));
  { // MatchMap::getEdgeMatches of the path edges (dg.cpp:99-101): the only EdgeMatches anything downstream reads
    std::uint32_t const *pathEdges = nullptr;
    std::size_t          nPathEdges = 0;
    detail::require(msgpu_graph_path_edges(graph.g, &pathEdges, &nPathEdges), "msgpu_graph_path_edges");
    std::uint64_t const   *emOff = nullptr;
    msgpu_edgematch const *ems   = nullptr;
    core.edgeMatchesOf(pathEdges, nPathEdges, &emOff, &ems);
    detail::require(msgpu_graph_set_path_edgematches(graph.g, emOff, ems), "msgpu_graph_set_path_edgematches",
                    msgpu_graph_last_error(graph.g));
  }

  registrar.join();
  if (regError) std::rethrow_exception(regError);

  detail::require(msgpu_assembly_create(s.ctx, &s.as), "msgpu_assembly_create"); // :300-310, 620-677
  detail::require(msgpu_assembly_borrow_rows(s.as, rows, nRows), "msgpu_assembly_borrow_rows"); // (the loader's table: core outlives s.as)
  std::vector<int> status(msgpu_graph_path_count(graph.g), 0);
  detail::require(msgpu_assembly_add_graph_paths(s.as, graph.g, threads ? threads : 1, status.data()),
                  "msgpu_assembly_add_graph_paths", msgpu_assembly_last_error(s.as));
  detail::require(msgpu_assembly_finish(s.as, nullptr), "msgpu_assembly_finish", msgpu_seq_last_error(s.ctx));
  n.paths = status.size();
  for (int st : status) n.pathsSkipped += st != MSGPU_OK;
  n.contigs = msgpu_assembly_path_count(s.as);
  n.queries = msgpu_assembly_query_count(s.as);
  for (std::uint32_t i = 0; i < n.contigs; ++i) {
    msgpu_path_info pi;
    msgpu_assembly_path_info(s.as, i, &pi);
    n.targetBases += pi.target_len;
  }
  char const *const names[3] = {"/temp_1.target.fa", "/temp_1.query.fa", "/temp_1.align.paf"}; // :294-296
  std::string       writeError[3];
  auto              write = [&](int w) { // (straight from the library's buffers, the three files side by side)
    std::uint64_t len  = 0;
    char const   *text = msgpu_assembly_text(s.as, w, &len);
    std::FILE    *f    = std::fopen((outDir + names[w]).c_str(), "wb");
    if (!f || (len && std::fwrite(text, 1, len, f) != len)) writeError[w] = "cannot write " + outDir + names[w];
    if (f && std::fclose(f) != 0) writeError[w] = "cannot write " + outDir + names[w];
  };
  {
    std::thread a([&] { write(0); }), b([&] { write(1); });
    write(2);
    a.join();
    b.join();
  }
  for (auto const &e : writeError)
    if (!e.empty()) throw std::runtime_error(e);
  return n;
}

} // namespace msgpu

#endif // MSGPU_ADAPTER_HPP
