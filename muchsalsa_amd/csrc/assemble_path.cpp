// assemble_path.cpp -- the layout half of muchsalsa::assemblePath (libms/src/kernel/ap.cpp:615-1362) on copy pieces.
//
// The reference turns one linearised path of reads into a target contig, query segments and a PAF by building
// std::strings as it goes (whole-record reads from disk, slices, reverse complements, O(L^2) re-copies of the growing
// contig).  Here the same decisions are taken on the host over integers only -- which EdgeOrder per path edge
// (:621-706), anchor cliques and their common overlaps (:91-189, :708-719), the per-read anchor order (:721-777), the
// anchor DAG (:779-853), anchor distances (:581-611), placement (:231-349, :865-1010), flanks (:1012-1032) and the
// contained reads (:1227-1361) -- and every sequence the reference would have built is recorded as a short list of
// copy pieces.  No base is read here; msgpu_assembly_finish (msgpu_seq.hip) produces all of them in one gather launch.
//
// Iteration orders the reference leaves to std::unordered_* are fixed as: vertices ascending id, anchor-DAG edges in
// creation order, successors / predecessors ascending id, tap entries ascending id, std::sort ties stable.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <pthread.h>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "asm_internal.h"
#include "msgpu.h"

namespace msgpu {
std::string target_header(int32_t asm_idx) { return ">muchsalsa_" + std::to_string(asm_idx) + "\n"; }
std::string query_header(uint32_t kind, int32_t asm_idx, uint32_t query_idx) {
  static const char *const names[] = {">Middle.", ">Left.", ">Right.", ">Contain_Illumina_Match.", ">Contain_Nano_Middle."};
  return std::string(names[kind]) + std::to_string(asm_idx) + "." + std::to_string(query_idx) + "\n";
}
} // namespace msgpu

namespace {

constexpr uint64_t TH_SEQUENCE_LENGTH = 200; // ap.cpp:53
constexpr int      NANO = 0, ILLU = 1;

struct LayoutError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct ApiError {
  int code;
};

struct Seg { // a sequence the reference would hold in a std::string
  std::vector<msgpu_copy> p;
  uint64_t                len = 0;
};

using Key = std::pair<uint32_t, uint32_t>; // (illumina id, clique index): the key of Id2OverlapMap
using Ov  = std::pair<int, int>;
struct Match { // tuple<tuple<id, clique>, modifier>, ap.cpp:721-723
  Key      key;
  uint32_t mod;
  bool     operator==(const Match &o) const { return key == o.key && mod == o.mod; }
};
struct Info { // one entry of vertexInfo
  Ov    nr;
  Match m;
};

// (sequence, borderLeft, borderRight) of updateConsensusBase
struct Base {
  msgpu_consensus *c   = msgpu_consensus_new();
  bool             has = false;
  Base() {
    if (!c) throw std::bad_alloc();
  }
  Base(const Base &o) : c(msgpu_consensus_clone(o.c)), has(o.has) {
    if (!c) throw std::bad_alloc();
  }
  Base &operator=(const Base &) = delete;
  ~Base() { msgpu_consensus_free(c); }
  void update(const Seg &s, int lo, int hi) {
    const int rc = msgpu_consensus_update(c, s.p.data(), static_cast<uint32_t>(s.p.size()), lo, hi);
    if (rc != MSGPU_OK) throw ApiError{rc};
    has = true;
  }
  void reset(const Seg &s, int lo, int hi) { // "sequence = anchorSequences.at(v); pos1 = 0; pos2 = ..."
    msgpu_consensus_free(c);
    c = msgpu_consensus_new();
    if (!c) throw std::bad_alloc();
    has = false;
    update(s, lo, hi);
  }
  int lo() const {
    int32_t l = 0;
    msgpu_consensus_borders(c, &l, nullptr, nullptr);
    return has ? l : 0;
  }
  int hi() const {
    int32_t h = 0;
    msgpu_consensus_borders(c, nullptr, &h, nullptr);
    return has ? h : 0;
  }
  Seg seg() const {
    Seg s;
    s.p.resize(msgpu_consensus_pieces(c, 0, nullptr, 0));
    msgpu_consensus_pieces(c, 0, s.p.data(), s.p.size());
    for (const auto &q : s.p) s.len += q.len;
    return s;
  }
};

struct Adg { // the anchor DiGraph of one path; vertices are registry ids 0..n-1
  std::vector<std::map<uint32_t, uint32_t>> succ, pred; // neighbour -> edge index, ascending
  std::vector<std::pair<uint32_t, uint32_t>> edges;     // creation order
  uint32_t add_vertex() {
    succ.emplace_back();
    pred.emplace_back();
    return static_cast<uint32_t>(succ.size() - 1);
  }
  uint32_t add_edge(uint32_t u, uint32_t v) { // GraphBase::_addEdgeInternal keeps an existing edge, Graph.cpp:291-311
    auto it = succ[u].find(v);
    if (it != succ[u].end()) return it->second;
    const uint32_t e = static_cast<uint32_t>(edges.size());
    edges.emplace_back(u, v);
    succ[u][v] = e;
    pred[v][u] = e;
    return e;
  }
  std::vector<uint32_t> sort_topologically() const { // DiGraph::sortTopologically, Graph.cpp:359-395
    const uint32_t        n = static_cast<uint32_t>(succ.size());
    std::vector<uint32_t> indeg(n), ready, result;
    for (uint32_t v = 0; v < n; ++v) {
      indeg[v] = static_cast<uint32_t>(pred[v].size());
      if (!indeg[v]) ready.push_back(v);
    }
    while (!ready.empty()) {
      const uint32_t v = ready.back();
      ready.pop_back();
      for (const auto &t : succ[v])
        if (--indeg[t.first] == 0) ready.push_back(t.first);
      result.push_back(v);
    }
    return result;
  }
};

// The same recursion on bit masks for clusters of up to 64 path edges (node i = i-th edge of the cluster, ascending):
// vs keeps its ascending order through the recursion exactly as the vectors below do, and only the MEMBERS of the
// returned clique are used by the caller, so a mask carries everything.
uint64_t ramsey_mask(const uint64_t *adj, uint64_t vs) {
  if (!vs) return 0;
  const int      first = __builtin_ctzll(vs);
  const uint64_t rest  = vs & (vs - 1);
  const uint64_t cn    = ramsey_mask(adj, rest & adj[first]) | (1ull << first);
  const uint64_t cnn   = ramsey_mask(adj, rest & ~adj[first]);
  return __builtin_popcountll(cn) >= __builtin_popcountll(cnn) ? cn : cnn;
}

// ramseyR2 / getAnchorCliques, ap.cpp:91-138, on the small interval-intersection graph of one anchor
std::vector<uint32_t> ramsey(const std::map<uint32_t, std::set<uint32_t>> &adj, const std::vector<uint32_t> &vs) {
  if (vs.empty()) return {};
  const uint32_t        first = vs[0];
  std::vector<uint32_t> nb, non;
  const auto           &a = adj.at(first);
  for (size_t i = 1; i < vs.size(); ++i) (a.count(vs[i]) ? nb : non).push_back(vs[i]);
  std::vector<uint32_t> cn  = ramsey(adj, nb);
  std::vector<uint32_t> cnn = ramsey(adj, non);
  cn.push_back(first);
  return cn.size() >= cnn.size() ? cn : cnn;
}

struct PathLayout {
  msgpu_seqctx           *ctx;
  const msgpu_path_input &in;
  const msgpu_assembly   *shared = nullptr; // rows installed once with msgpu_assembly_set_rows
  std::unordered_map<uint64_t, const msgpu_row *> vm;
  std::unordered_map<uint32_t, uint32_t>          dir_of; // read id -> direction
  std::vector<std::map<uint32_t, Ov>>             em;     // per path edge: anchor -> overlap

  PathLayout(msgpu_seqctx *c, const msgpu_path_input &i) : ctx(c), in(i) {}

  const msgpu_row *row(uint32_t read, uint32_t anchor) const {
    const uint64_t key = (static_cast<uint64_t>(read) << 32) | anchor;
    auto           it  = vm.find(key);
    if (it != vm.end()) return it->second;
    if (shared && static_cast<size_t>(anchor) + 1 < shared->anchor_start.size()) { // the anchor's own scaffold only
      const msgpu_row *b = shared->rows_view + shared->anchor_start[anchor];
      const msgpu_row *e = shared->rows_view + shared->anchor_start[static_cast<size_t>(anchor) + 1];
      // scaffolds in read-id order (what msgpu_parse_paf's output grouped by query and target is): O(log n) whatever the
      // coverage; otherwise the table was only accepted with short scaffolds (install_rows) and a scan is as good
      if (shared->anchor_sorted_by_read)
        b = std::lower_bound(b, e, read, [](const msgpu_row &m, uint32_t r) { return m.read_id < r; });
      const msgpu_row *best = nullptr;
      for (const msgpu_row *m = b; m < e; ++m) {
        if (shared->anchor_sorted_by_read && m->read_id != read) break;
        if (m->read_id == read && (!best || m->line < best->line)) best = m; // of equal keys the lowest line
      }
      if (best) return best;
    } else if (shared && static_cast<size_t>(read) + 1 < shared->row_start.size()) { // the read's own ~50 rows only
      const auto b  = shared->row_recs.begin() + static_cast<long>(shared->row_start[read]);
      const auto e  = shared->row_recs.begin() + static_cast<long>(shared->row_start[read + 1]);
      const auto lo = std::lower_bound(b, e, key, [](const msgpu_assembly::RowRec &r, uint64_t k) { return r.key < k; });
      if (lo != e && lo->key == key) return &shared->rows_view[lo->idx]; // the first of equal keys = the lowest line
    }
    throw LayoutError("no VertexMatch for read " + std::to_string(read) + " on anchor " + std::to_string(anchor));
  }
  static bool mdir(const msgpu_row *m) { return (m->flags & MSGPU_ROW_DIR) != 0; }
  static void check(int rc) {
    if (rc != MSGPU_OK) throw ApiError{rc};
  }
  static void close(Seg &s, uint32_t n, uint64_t len) {
    s.p.resize(n);
    s.len = len;
  }

  Seg anchor_seq(uint32_t read, uint32_t anchor, Ov ov, bool pos) const {
    Seg      s;
    uint32_t n = 0;
    s.p.resize(1);
    check(msgpu_seg_anchor(ctx, row(read, anchor), ov.first, ov.second, pos, s.p.data(), &n, &s.len));
    close(s, n, s.len);
    return s;
  }
  Seg flank(bool left, const msgpu_path_read &r, uint32_t anchor, Ov ov) const {
    Seg      s;
    uint32_t n = 0;
    s.p.resize(2);
    check((left ? msgpu_seg_left_of_anchor : msgpu_seg_right_of_anchor)(
        ctx, row(r.read_id, anchor), r.nanopore_length, ov.first, ov.second, r.direction == 1, s.p.data(), &n, &s.len));
    close(s, n, s.len);
    return s;
  }
  // one get{Illumina,Nanopore}Sequence(id, l, r, d) as a segment
  Seg slice(int kind, uint32_t id, int l, int r, bool d) const {
    Seg s;
    s.p.resize(1);
    check(msgpu_seq_resolve(ctx, kind, id, l, r, d, s.p.data()));
    s.p[0].dst_off = 0;
    s.len          = s.p[0].len;
    return s;
  }
};

struct Candidate { // ap.cpp:621-629 (edges[i] is always path edge i)
  std::vector<uint32_t>              open, visited; // the reference's id sets, kept as sorted unique vectors
  uint64_t                           score = 0, kinks = 0;
  std::vector<uint32_t>              orders; // index into in.orders
  std::vector<std::vector<uint32_t>> modifiers;
};

void find_best(const std::vector<Candidate> &cs, bool &any, uint64_t &min_kinks, uint64_t &max_score) { // :633-642
  any = false;
  for (const auto &c : cs)
    if (!any || c.kinks < min_kinks || (c.kinks == min_kinks && c.score > max_score)) {
      any       = true;
      min_kinks = c.kinks;
      max_score = c.score;
    }
}

struct Record { // one output sequence of the path
  uint32_t kind; // MSGPU_QUERY_* or ~0u for the target
  Seg      seg;
  int64_t  lb = 0, rb = 0;
};

// visitOrdered, ap.cpp:231-349
Base visit_ordered(std::map<uint32_t, bool> &visited, std::map<uint32_t, Ov> &tap, const Adg &adg,
                   const std::vector<Key> &reg2id, const std::vector<uint32_t> &pos_of,
                   const std::vector<uint32_t> &order, const std::vector<int> &distances,
                   const std::vector<std::vector<Seg>> &sequences, const std::vector<Seg> &anchor_seq,
                   const std::map<Key, Ov> &id2ov, uint32_t start) {
  Base base;
  auto cmp = [](const std::pair<size_t, int> &l, const std::pair<size_t, int> &r) { // :244-250
    return std::make_pair(l.first, -l.second) < std::make_pair(r.first, -r.second);
  };
  std::set<std::pair<size_t, int>, decltype(cmp)> queue_edges(cmp);
  std::set<size_t>                                queue_vertices;
  queue_vertices.insert(pos_of[start]);
  while (!queue_vertices.empty()) {
    const size_t   idx = *queue_vertices.begin();
    const uint32_t v   = order[idx];
    queue_vertices.erase(queue_vertices.begin());
    if (!visited[v]) {
      visited[v] = true;
      for (const auto &t : adg.succ[v]) {
        queue_edges.emplace(pos_of[t.first], static_cast<int>(idx));
        queue_vertices.insert(pos_of[t.first]);
      }
      while (!queue_edges.empty() && queue_edges.begin()->first == idx) {
        const uint32_t left = order[static_cast<size_t>(queue_edges.begin()->second)], right = order[queue_edges.begin()->first];
        const bool     has_l = tap.count(left) != 0, has_r = tap.count(right) != 0;
        const Ov       ov_l = id2ov.at(reg2id[left]), ov_r = id2ov.at(reg2id[right]);
        const uint32_t e      = adg.succ[left].at(right);
        const int      offset = distances[e];
        const int      len_l = ov_l.second - ov_l.first + 1, len_r = ov_r.second - ov_r.first + 1;
        if (has_l && !has_r) { // :295-307
          const int pos = tap.at(left).second;
          tap[right]    = Ov(pos + offset + 1, pos + offset + len_r);
          if (offset > 0) base.update(sequences[e].front(), pos + 1, pos + offset);
          base.update(anchor_seq[right], tap[right].first, tap[right].second);
        } else if (!has_l && has_r) { // :308-320
          const int pos = tap.at(right).first;
          tap[left]     = Ov(pos - offset - len_l, pos - offset - 1);
          if (offset > 0) base.update(sequences[e].front(), pos - offset, pos);
          base.update(anchor_seq[left], tap[left].first, tap[left].second);
        } else if (!has_l && !has_r) { // :321-337
          tap[left]  = Ov(0, len_l - 1);
          tap[right] = Ov(len_l + offset, len_l + offset + len_r - 1);
          if (offset > 0) base.update(sequences[e].front(), len_l, len_l + offset - 1);
          base.update(anchor_seq[left], tap[left].first, tap[left].second);
          base.update(anchor_seq[right], tap[right].first, tap[right].second);
        }
        queue_edges.erase(queue_edges.begin());
      }
    } else {
      while (!queue_edges.empty() && queue_edges.begin()->first == idx) queue_edges.erase(queue_edges.begin());
    }
  }
  return base;
}

struct PathResult {
  Seg                 target;
  std::vector<Record> queries;
  msgpu_path_info     info{};
  int                 left_most = 0;
  std::string         paf; // the path's temp_1.align.paf lines, formatted by the layout thread
};

// MSGPU_ASM_DEBUG=1: nanoseconds per layout section, summed over all paths and threads, printed by msgpu_assembly_free
std::atomic<uint64_t> g_sec_ns[10]; // 0..7 layout sections, 8 = wall time of the layout fan-out, 9 = commits
const bool            g_sec_on = [] {
  const char *e = getenv("MSGPU_ASM_DEBUG");
  return e && e[0] == '1';
}();
struct SecTimer {
  std::chrono::steady_clock::time_point t;
  SecTimer() {
    if (g_sec_on) t = std::chrono::steady_clock::now();
  }
  void mark(int i) {
    if (!g_sec_on) return;
    const auto n = std::chrono::steady_clock::now();
    g_sec_ns[i] += static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count());
    t = n;
  }
};

PathResult layout_path(const msgpu_assembly *a, const msgpu_path_input &in) {
  SecTimer sec;
  msgpu_seqctx *ctx = a->ctx;
  PathLayout    L(ctx, in);
  L.shared = a;
  const uint32_t n_reads = in.n_reads, n_edges = n_reads - 1;
  for (size_t i = 0; i < in.n_rows; ++i) { // MatchMap::addVertexMatch keeps the lowest line, MatchMap.cpp:64-80
    const uint64_t k = (static_cast<uint64_t>(in.rows[i].read_id) << 32) | in.rows[i].anchor_id;
    auto           it = L.vm.find(k);
    if (it == L.vm.end() || in.rows[i].line < it->second->line) L.vm[k] = &in.rows[i];
  }
  for (uint32_t i = 0; i < n_reads; ++i) L.dir_of[in.reads[i].read_id] = in.reads[i].direction;
  L.em.resize(n_edges);
  for (uint32_t i = 0; i < n_edges; ++i)
    for (uint32_t k = in.em_off[i]; k < in.em_off[i + 1]; ++k)
      L.em[i].emplace(in.ems[k].anchor_id, Ov(in.ems[k].ov_lo, in.ems[k].ov_hi));
  auto order_ids = [&](uint32_t oi) { // ids of an order, reversed when its base vertex is e_NEG (:658-661, :731-735)
    const msgpu_path_order &o = in.orders[oi];
    std::vector<uint32_t>   ids(in.ids + o.ids_off, in.ids + o.ids_off + o.ids_cnt);
    auto                    d = L.dir_of.find(o.base_read);
    if (d == L.dir_of.end()) throw LayoutError("EdgeOrder based at a read that is not on the path");
    if (d->second == 0) std::reverse(ids.begin(), ids.end()); // getVertexDirection() == e_NEG
    return ids;
  };
  auto em_of = [&](uint32_t edge, uint32_t anchor) {
    auto it = L.em[edge].find(anchor);
    if (it == L.em[edge].end())
      throw LayoutError("no EdgeMatch for anchor " + std::to_string(anchor) + " on path edge " + std::to_string(edge));
    return it->second;
  };

  sec.mark(0);
  // ---- which EdgeOrder per path edge, ap.cpp:631-706 -----------------------------------------------------------------
  std::vector<Candidate> candidates(1);
  for (uint32_t i = 0; i < n_edges; ++i) {
    std::vector<Candidate> next;
    for (uint32_t oi = in.order_off[i]; oi < in.order_off[i + 1]; ++oi) {
      const std::vector<uint32_t> ids = order_ids(oi);
      std::vector<Candidate>      sub;
      for (const Candidate &c : candidates) {
        Candidate n;
        std::vector<uint32_t> mods;
        for (uint32_t id : ids)
          if (!std::binary_search(c.open.begin(), c.open.end(), id) && std::binary_search(c.visited.begin(), c.visited.end(), id))
            mods.push_back(id);
        n.open = ids;
        std::sort(n.open.begin(), n.open.end());
        n.open.erase(std::unique(n.open.begin(), n.open.end()), n.open.end());
        n.visited.resize(c.visited.size() + n.open.size());
        n.visited.erase(std::set_union(c.visited.begin(), c.visited.end(), n.open.begin(), n.open.end(), n.visited.begin()),
                        n.visited.end());
        n.score = c.score + in.orders[oi].score;
        n.kinks = c.kinks + mods.size();
        n.orders = c.orders;
        n.orders.push_back(oi);
        n.modifiers = c.modifiers;
        n.modifiers.push_back(std::move(mods));
        sub.push_back(std::move(n));
      }
      bool     any;
      uint64_t mk = 0, ms = 0;
      find_best(sub, any, mk, ms);
      for (auto &c : sub)
        if (any && c.kinks == mk && c.score == ms) next.push_back(std::move(c));
    }
    candidates = std::move(next);
  }
  if (candidates.empty()) throw LayoutError("a path edge has no EdgeOrder");
  bool     any;
  uint64_t mk = 0, ms = 0;
  find_best(candidates, any, mk, ms);
  const Candidate &best =
      *std::find_if(candidates.begin(), candidates.end(), [&](const Candidate &c) { return c.kinks == mk && c.score == ms; });

  sec.mark(1);
  // ---- anchor clusters -> cliques -> common overlaps, :708-719 + getClusterAnchors :140-189 ----------------------------
  std::map<uint32_t, std::vector<uint32_t>> clusters;
  for (uint32_t idx = 0; idx < n_edges; ++idx) {
    const msgpu_path_order &o = in.orders[best.orders[idx]];
    for (uint32_t k = 0; k < o.ids_cnt; ++k) clusters[in.ids[o.ids_off + k]].push_back(idx);
  }
  std::map<Key, Ov>                              id2ov;
  std::vector<std::unordered_map<uint32_t, uint32_t>> cluster_modifier(n_edges);
  for (const auto &cl : clusters) {
    const uint32_t                         anchor = cl.first;
    if (cl.second.size() == 1) { // one path edge: one clique, its overlap (what the general code below yields)
      cluster_modifier[cl.second[0]][anchor] = 0;
      id2ov[Key(anchor, 0)]                  = em_of(cl.second[0], anchor);
      continue;
    }
    const size_t m = cl.second.size();
    if (m <= 64 && std::adjacent_find(cl.second.begin(), cl.second.end(), std::greater_equal<uint32_t>()) == cl.second.end()) {
      // strictly ascending edge list (no anchor listed twice by one order): the same cliques on bit masks
      Ov       ovs[64];
      uint64_t madj[64];
      for (size_t i = 0; i < m; ++i) {
        ovs[i]  = em_of(cl.second[i], anchor);
        madj[i] = 0;
      }
      for (size_t i = 0; i < m; ++i)
        for (size_t j = 0; j < i; ++j)
          if (std::max(ovs[i].first, ovs[j].first) <= std::min(ovs[i].second, ovs[j].second)) {
            madj[i] |= 1ull << j;
            madj[j] |= 1ull << i;
          }
      uint64_t left = m == 64 ? ~0ull : ((1ull << m) - 1);
      uint32_t idx  = 0;
      while (left) {
        const uint64_t clique = ramsey_mask(madj, left); // never empty while nodes are left
        left &= ~clique;
        Ov   common;
        bool have = false;
        for (uint64_t c = clique; c; c &= c - 1) {
          const int i = __builtin_ctzll(c);
          cluster_modifier[cl.second[static_cast<size_t>(i)]][anchor] = idx;
          common = have ? Ov(std::max(common.first, ovs[i].first), std::min(common.second, ovs[i].second)) : ovs[i];
          have   = true;
        }
        id2ov[Key(anchor, idx)] = common;
        ++idx;
      }
      continue;
    }
    std::map<uint32_t, std::set<uint32_t>> adj;
    for (uint32_t e1 : cl.second) {
      adj[e1];
      for (uint32_t e2 : cl.second) {
        if (e1 == e2) break;
        const Ov o1 = em_of(e1, anchor), o2 = em_of(e2, anchor);
        if (std::max(o1.first, o2.first) <= std::min(o1.second, o2.second)) {
          adj[e2].insert(e1);
          adj[e1].insert(e2);
        }
      }
    }
    std::set<uint32_t> left;
    for (const auto &a : adj) left.insert(a.first);
    std::vector<std::vector<uint32_t>> cliques;
    std::vector<uint32_t> current = ramsey(adj, std::vector<uint32_t>(left.begin(), left.end()));
    cliques.push_back(current);
    while (!left.empty()) {
      for (uint32_t v : current) left.erase(v);
      current = ramsey(adj, std::vector<uint32_t>(left.begin(), left.end()));
      if (!current.empty()) cliques.push_back(current);
    }
    for (uint32_t idx = 0; idx < cliques.size(); ++idx) {
      bool have = false;
      Ov   common;
      for (uint32_t v : cliques[idx]) {
        cluster_modifier[v][anchor] = idx;
        const Ov ov                 = em_of(v, anchor);
        common = have ? Ov(std::max(common.first, ov.first), std::min(common.second, ov.second)) : ov;
        have   = true;
      }
      if (!have) throw LayoutError("empty anchor clique"); // commonOverlap.value() would throw
      id2ov[Key(anchor, idx)] = common;
    }
  }

  sec.mark(2);
  // ---- anchors per read in read order, :721-752 --------------------------------------------------------------------------
  std::vector<std::vector<Info>>          vertex_info(n_edges + 1);
  std::unordered_map<uint32_t, uint32_t> match_modifiers;
  for (uint32_t idx = 0; idx < n_edges; ++idx) {
    for (uint32_t m : best.modifiers[idx]) ++match_modifiers[m];
    const uint32_t ra = in.reads[idx].read_id, rb = in.reads[idx + 1].read_id;
    for (uint32_t id : order_ids(best.orders[idx])) {
      auto        mm = match_modifiers.find(id);
      const Match m{Key(id, cluster_modifier[idx][id]), mm == match_modifiers.end() ? 0u : mm->second};
      const msgpu_row *a = L.row(ra, id), *b = L.row(rb, id);
      vertex_info[idx].push_back(Info{Ov(a->n_lo, a->n_hi), m});
      vertex_info[idx + 1].push_back(Info{Ov(b->n_lo, b->n_hi), m});
    }
  }

  sec.mark(3);
  // ---- anchor DAG, anchor sequences, flanks, :754-853 ----------------------------------------------------------------------
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> registry; // tupleTuple2Id -> Registry id
  Adg                                                          adg;
  std::vector<Key>                                             reg2id;
  std::vector<Seg>                                             anchor_seq;
  std::vector<std::vector<uint32_t>>                           nanopores; // per DAG edge: path positions of its reads
  std::map<uint32_t, std::vector<Seg>>                         pre, post;
  for (uint32_t idx = 0; idx <= n_edges; ++idx) {
    const msgpu_path_read &r   = in.reads[idx];
    const bool             pos = r.direction == 1, neg = r.direction == 0; // e_POS / e_NEG; 2 = e_NONE is neither
    auto                  &info = vertex_info[idx];
    std::stable_sort(info.begin(), info.end(), [&](const Info &l, const Info &rr) { // :760-770
      if (l.nr == rr.nr) {
        const Ov lo = id2ov.at(l.m.key), ro = id2ov.at(rr.m.key);
        if (!PathLayout::mdir(L.row(r.read_id, l.m.key.first))) return ro < lo;
        return lo < ro;
      }
      return l.nr < rr.nr;
    });
    if (neg) std::reverse(info.begin(), info.end());
    if (info.empty()) continue;
    auto ensure = [&](const Match &m) { // :787-793, :800-806
      const auto key = std::make_tuple(m.key.first, m.key.second, m.mod);
      auto       it  = registry.find(key);
      if (it != registry.end()) return it->second;
      const uint32_t v = adg.add_vertex();
      registry.emplace(key, v);
      anchor_seq.push_back(L.anchor_seq(r.read_id, m.key.first, id2ov.at(m.key), pos));
      reg2id.push_back(m.key);
      return v;
    };
    Ov    last_nr    = info.front().nr;
    Match last_match = info.front().m;
    for (const Info &it : info) {
      const uint32_t v = ensure(it.m);
      if (it.m == last_match) continue;
      const uint32_t vl   = ensure(last_match);
      bool           flip = false;
      if ((last_nr.second > it.nr.second && last_nr.first < it.nr.first) ||
          (last_nr.second < it.nr.second && last_nr.first > it.nr.first)) { // :809-822, getCorrectedNanoporeRange :191-203
        auto corrected = [&](const Match &m, double &first, double &second) {
          const msgpu_row *row = L.row(r.read_id, m.key.first);
          const Ov         ov  = id2ov.at(m.key);
          const double ratio = static_cast<double>(row->i_hi - row->i_lo + 1) / static_cast<double>(row->n_hi - row->n_lo + 1);
          double       cl = (ov.first - row->i_lo) / ratio, cr = (row->i_hi - ov.second) / ratio;
          if (!PathLayout::mdir(row)) std::swap(cl, cr);
          first  = row->n_lo + cl;
          second = row->n_hi - cr;
        };
        double lf, ls, rf, rs;
        corrected(last_match, lf, ls);
        corrected(it.m, rf, rs);
        flip = (pos && (lf > rf || (lf == rf && ls > rs))) || (neg && (lf < rf || (lf == rf && ls < rs)));
      }
      const uint32_t e = flip ? adg.add_edge(v, vl) : adg.add_edge(vl, v);
      if (e >= nanopores.size()) nanopores.resize(e + 1);
      nanopores[e].push_back(idx);
      last_match = it.m;
      last_nr    = it.nr;
    }
    const Match &first = info.front().m, &second = info.back().m;
    pre[registry.at(std::make_tuple(first.key.first, first.key.second, first.mod))].push_back(
        L.flank(true, r, first.key.first, id2ov.at(first.key)));
    post[registry.at(std::make_tuple(second.key.first, second.key.second, second.mod))].push_back(
        L.flank(false, r, second.key.first, id2ov.at(second.key)));
  }
  const uint32_t n_anchors = static_cast<uint32_t>(adg.succ.size());

  sec.mark(4);
  // ---- sequences between neighbouring anchors, :855-863 + alignAnchorRegion :581-611 ---------------------------------------
  std::vector<int>              distances(adg.edges.size());
  std::vector<std::vector<Seg>> sequences(adg.edges.size());
  for (uint32_t e = 0; e < adg.edges.size(); ++e) {
    const Key  kl = reg2id[adg.edges[e].first], kr = reg2id[adg.edges[e].second];
    const Ov   ovl = id2ov.at(kl), ovr = id2ov.at(kr);
    bool       have = false;
    for (uint32_t pi : nanopores[e]) {
      const msgpu_path_read &r = in.reads[pi];
      Seg                    s;
      uint32_t               n = 0;
      int32_t                dist = 0;
      int                    has  = 0;
      s.p.resize(3);
      PathLayout::check(msgpu_seg_between_anchors(ctx, L.row(r.read_id, kl.first), L.row(r.read_id, kr.first), ovl.first,
                                                  ovl.second, ovr.first, ovr.second, r.direction == 1, s.p.data(), &n,
                                                  &dist, &has));
      if (has) {
        PathLayout::close(s, n, static_cast<uint64_t>(dist));
        sequences[e].push_back(std::move(s));
      }
      if (!have) {
        distances[e] = dist;
        have         = true;
      }
    }
  }

  sec.mark(5);
  // ---- placement, :865-1010 -----------------------------------------------------------------------------------------------------
  const std::vector<uint32_t> order = adg.sort_topologically();
  if (order.empty() || order.size() != n_anchors) throw LayoutError("the anchor graph of the path has a cycle");
  std::vector<uint32_t> pos_of(n_anchors);
  for (uint32_t i = 0; i < n_anchors; ++i) pos_of[order[i]] = i;
  std::map<uint32_t, bool> visited;
  std::map<uint32_t, Ov>   tap;
  Base glob = visit_ordered(visited, tap, adg, reg2id, pos_of, order, distances, sequences, anchor_seq, id2ov, order[0]);
  if (n_anchors == 1) { // :886-895
    const Ov ov = id2ov.at(reg2id[0]);
    tap[0]      = Ov(0, ov.second - ov.first);
    glob.reset(anchor_seq[0], 0, ov.second - ov.first);
  }
  struct Extra {
    Base                   base;
    std::map<uint32_t, Ov> tap;
    bool                   added;
  };
  std::vector<Extra> additional;
  for (size_t i = 1; i < order.size(); ++i) { // :897-925
    const uint32_t v = order[i];
    if (visited.count(v)) continue;
    std::map<uint32_t, Ov> ltap;
    Base loc = visit_ordered(visited, ltap, adg, reg2id, pos_of, order, distances, sequences, anchor_seq, id2ov, v);
    if (ltap.empty()) {
      const Ov ov = id2ov.at(reg2id[v]);
      ltap[v]     = Ov(0, ov.second - ov.first);
      loc.reset(anchor_seq[v], 0, ov.second - ov.first);
    }
    additional.push_back(Extra{loc, std::move(ltap), false});
  }
  for (bool loop = true; loop;) { // :927-1010
    loop          = false;
    bool progress = false;
    for (Extra &x : additional) {
      if (x.added) continue;
      Base loc(x.base);
      int  group_offset = 0;
      bool found        = false;
      for (const auto &m : x.tap) {
        found = false;
        for (const auto &t : adg.succ[m.first]) {
          auto tp = tap.find(t.first);
          if (tp == tap.end()) continue;
          const uint32_t e = t.second;
          group_offset     = tp->second.first - distances[e] - m.second.second - 1;
          if (!sequences[e].empty()) loc.update(sequences[e].front(), m.second.second + 1, m.second.second + distances[e]);
          found = true;
          break;
        }
        if (found) break;
        for (const auto &t : adg.pred[m.first]) {
          auto tp = tap.find(t.first);
          if (tp == tap.end()) continue;
          const uint32_t e = t.second;
          group_offset     = tp->second.second + distances[e] + 1 - m.second.first + 1;
          if (!sequences[e].empty()) loc.update(sequences[e].front(), m.second.first - distances[e], m.second.first - 1);
          found = true;
          break;
        }
        if (found) break;
      }
      if (!found) {
        loop = true;
        continue;
      }
      x.added  = true;
      progress = true;
      for (const auto &m : x.tap) tap[m.first] = Ov(m.second.first + group_offset, m.second.second + group_offset);
      if (!loc.has) throw LayoutError("anchor group without a sequence"); // localSequence.value()
      glob.update(loc.seg(), loc.lo() + group_offset, loc.hi() + group_offset);
    }
    if (loop && !progress) throw LayoutError("a group of anchors never connects to the contig");
  }
  if (!glob.has) throw LayoutError("path without a sequence"); // globalSequence.value()

  auto longest = [](const std::vector<Seg> &v) { // std::max_element: the first of the longest
    return &*std::max_element(v.begin(), v.end(), [](const Seg &a, const Seg &b) { return a.len < b.len; });
  };
  auto tap_at = [&](uint32_t v) {
    auto it = tap.find(v);
    if (it == tap.end()) throw LayoutError("anchor without a position"); // tap.at
    return it->second;
  };
  for (uint32_t v = 0; v < n_anchors; ++v) { // :1012-1032
    auto p = pre.find(v);
    if (p != pre.end()) {
      const Seg *s   = longest(p->second);
      const int  len = static_cast<int>(s->len);
      glob.update(*s, tap_at(v).first - len, tap_at(v).first - 1);
    }
    p = post.find(v);
    if (p != post.end()) {
      const Seg *s   = longest(p->second);
      const int  len = static_cast<int>(s->len);
      glob.update(*s, tap_at(v).second + 1, tap_at(v).second + len);
    }
  }

  sec.mark(6);
  // ---- output records, :1034-1361 ----------------------------------------------------------------------------------------------
  PathResult res;
  res.target    = glob.seg();
  res.left_most = -glob.lo();
  const int lm  = res.left_most;
  for (uint32_t e = 0; e < adg.edges.size(); ++e) // :1052-1109
    for (const Seg &s : sequences[e]) {
      if (!s.len) continue;
      res.queries.push_back(Record{MSGPU_QUERY_MIDDLE, s, tap_at(adg.edges[e].first).second + 1 + lm,
                                   tap_at(adg.edges[e].second).first - 1 + lm});
    }
  for (uint32_t v = 0; v < n_anchors; ++v) { // :1111-1225
    auto p = pre.find(v);
    if (p != pre.end())
      for (const Seg &s : p->second) {
        if (s.len < TH_SEQUENCE_LENGTH) continue;
        const int64_t rb = tap_at(v).first - 1 + lm;
        res.queries.push_back(Record{MSGPU_QUERY_LEFT, s, rb - static_cast<int>(s.len) + 1, rb});
      }
    p = post.find(v);
    if (p != post.end())
      for (const Seg &s : p->second) {
        if (s.len < TH_SEQUENCE_LENGTH) continue;
        const int64_t lb = tap_at(v).second + 1 + lm;
        res.queries.push_back(Record{MSGPU_QUERY_RIGHT, s, lb, lb + static_cast<int>(s.len) - 1});
      }
  }
  for (uint32_t idx = 0; idx <= n_edges; ++idx) { // contained reads, :1227-1361
    const msgpu_path_read            &r = in.reads[idx];
    const bool                        pos = r.direction == 1;
    std::unordered_map<uint32_t, Match> id2anchor;
    for (const Info &i : vertex_info[idx]) id2anchor[i.m.key.first] = i.m;
    for (uint32_t ci = 0; ci < in.n_contains; ++ci) {
      const msgpu_path_contain &ce = in.contains[ci];
      if (ce.host_read != r.read_id) continue;
      std::vector<std::pair<Ov, uint32_t>> cinfo;
      for (uint32_t k = 0; k < ce.anchors_cnt; ++k) {
        const uint32_t a = in.contain_anchors[ce.anchors_off + k];
        if (!id2anchor.count(a)) continue;
        const msgpu_row *cm = L.row(ce.nano, a);
        cinfo.emplace_back(Ov(cm->n_lo, cm->n_hi), a);
      }
      if (cinfo.empty()) continue;
      std::sort(cinfo.begin(), cinfo.end());
      const bool direction = msgpu::toggle_mul(ce.direction != 0, pos);
      if (!direction) std::reverse(cinfo.begin(), cinfo.end());
      std::vector<std::pair<int, int>> ranges;
      for (const auto &ci2 : cinfo) {
        const uint32_t   a       = ci2.second;
        const Match     &tap_id  = id2anchor.at(a);
        const bool       tap_dir = PathLayout::mdir(L.row(r.read_id, a)) == pos;
        const Ov         ov      = id2ov.at(tap_id.key);
        const int        illumina_ref = tap_dir ? ov.second : ov.first;
        const int        total_ref =
            tap_at(registry.at(std::make_tuple(tap_id.key.first, tap_id.key.second, tap_id.mod))).second + lm;
        const msgpu_row *cm       = L.row(ce.nano, a);
        const bool       cont_dir = PathLayout::mdir(cm) == direction;
        if (!cont_dir) {
          const int off = cm->i_lo - illumina_ref;
          ranges.emplace_back(total_ref - off - (cm->i_hi - cm->i_lo), total_ref - off);
        } else {
          const int off = cm->i_hi - illumina_ref;
          ranges.emplace_back(total_ref + off - (cm->i_hi - cm->i_lo), total_ref + off);
        }
      }
      std::vector<Record> to_write;
      for (size_t k = 0; k < cinfo.size(); ++k) {
        const uint32_t   a  = cinfo[k].second;
        const msgpu_row *cm = L.row(ce.nano, a);
        to_write.push_back(Record{MSGPU_QUERY_CONTAIN_ILLUMINA,
                                  L.slice(ILLU, a, cm->i_lo, cm->i_hi, PathLayout::mdir(cm) == direction), ranges[k].first,
                                  ranges[k].second});
        if (k == 0) continue;
        const Ov pre_n = cinfo[k - 1].first;
        to_write.push_back(Record{MSGPU_QUERY_CONTAIN_NANO, L.slice(NANO, ce.nano, pre_n.second + 1, cm->n_lo - 1, direction),
                                  ranges[k - 1].second + 1, ranges[k].first - 1});
      }
      for (Record &w : to_write) {
        if (w.seg.len < TH_SEQUENCE_LENGTH) continue;
        res.queries.push_back(std::move(w));
      }
    }
  }
  res.info.target_len     = res.target.len;
  res.info.n_anchors      = n_anchors;
  res.info.n_anchor_edges = static_cast<uint32_t>(adg.edges.size());
  res.info.border_lo      = glob.lo();
  res.info.border_hi      = glob.hi();
  res.info.asm_idx        = in.asm_idx;
  sec.mark(7);
  return res;
}

uint64_t place(msgpu_assembly *a, const Seg &s) { // append a record's pieces to the raw layout, 16-B aligned
  const uint64_t off = (a->raw_bytes + 15) & ~15ull;
  for (msgpu_copy p : s.p) {
    if (!p.len) continue;
    p.dst_off += off;
    a->pieces.push_back(p);
  }
  a->raw_bytes = off + s.len;
  return off;
}

} // namespace

extern "C" {

int msgpu_assembly_create(msgpu_seqctx *ctx, msgpu_assembly **out) {
  if (!ctx || !out) return MSGPU_E_ARG;
  *out = new (std::nothrow) msgpu_assembly();
  if (!*out) return MSGPU_E_NOMEM;
  (*out)->ctx = ctx;
  return MSGPU_OK;
}

void msgpu_assembly_free(msgpu_assembly *a) {
  if (a && g_sec_on) {
    static const char *const names[10] = {"inputs", "edge orders", "cliques", "anchors per read", "anchor DAG + flanks",
                                          "between anchors", "placement", "records", "add_paths: layout (wall)",
                                          "add_paths: commit (wall)"};
    for (int i = 0; i < 10; ++i)
      fprintf(stderr, "[msgpu asm] %-22s %10.3f ms\n", names[i], static_cast<double>(g_sec_ns[i].exchange(0)) * 1e-6);
  }
  if (a && a->release) a->release(a);
  delete a;
}
const char *msgpu_assembly_last_error(const msgpu_assembly *a) { return a ? a->err : "null assembly"; }

namespace {

int check_input(const msgpu_path_input *in) {
  if (!in || !in->reads || !in->order_off || !in->em_off || (in->n_rows && !in->rows)) return MSGPU_E_ARG;
  if (in->n_reads < 2) return MSGPU_E_LAYOUT;
  const uint32_t ne = in->n_reads - 1;
  if ((in->order_off[ne] && (!in->orders || !in->ids)) || (in->em_off[ne] && !in->ems) ||
      (in->n_contains && (!in->contains || !in->contain_anchors)))
    return MSGPU_E_ARG;
  return MSGPU_OK;
}

// layout of one path; never throws.  msg receives the reason when the result is not MSGPU_OK.
int try_layout(const msgpu_assembly *a, const msgpu_path_input *in, PathResult &r, std::string &msg) {
  int rc = check_input(in);
  if (rc == MSGPU_E_LAYOUT) msg = "a path needs at least two reads";
  if (rc != MSGPU_OK) return rc;
  try {
    r = layout_path(a, *in);
  } catch (LayoutError const &e) {
    msg = e.what();
    return MSGPU_E_LAYOUT;
  } catch (ApiError const &e) {
    msg = std::string(msgpu_strerror(e.code)) + " (" + msgpu_seq_last_error(a->ctx) + ")";
    return e.code;
  } catch (std::out_of_range const &e) {
    msg = std::string("missing map entry (") + e.what() + ")";
    return MSGPU_E_LAYOUT;
  } catch (std::bad_alloc const &) {
    return MSGPU_E_NOMEM;
  } catch (std::exception const &e) { // nothing may leave a worker thread or cross the C-ABI
    msg = e.what();
    return MSGPU_E_LAYOUT;
  }
  return MSGPU_OK;
}

// append a laid-out path to the assembly: raw layout, records, PAF text (OutputWriter order = path order)
// the path's lines of temp_1.align.paf (ap.cpp:1041-1056 and the four other record sites): one line per query
void format_paf(PathResult &r, int32_t asm_idx) {
  const std::string tname = "muchsalsa_" + std::to_string(asm_idx);
  uint32_t          qi    = 0;
  for (const Record &q : r.queries) {
    std::string name = msgpu::query_header(q.kind, asm_idx, qi++);
    name             = name.substr(1, name.size() - 2); // without '>' and '\n'
    char line[512];
    const long long len = static_cast<long long>(q.seg.len), span = static_cast<long long>(q.rb - q.lb + 1);
    snprintf(line, sizeof(line), "%s\t%lld\t0\t%lld\t+\t%s\t%llu\t%lld\t%lld\t%lld\t%lld\t255\n", name.c_str(), len, len,
             tname.c_str(), static_cast<unsigned long long>(r.target.len), static_cast<long long>(q.lb),
             static_cast<long long>(q.rb), span, span);
    r.paf += line;
  }
}

int commit(msgpu_assembly *a, PathResult &r, int32_t asm_idx) {
  try {
    r.info.target_raw_off = place(a, r.target);
    r.info.query_begin    = static_cast<uint32_t>(a->queries.size());
    for (const Record &q : r.queries) {
      msgpu_query_info info{};
      info.len     = q.seg.len;
      info.raw_off = place(a, q.seg);
      info.lb      = q.lb;
      info.rb      = q.rb;
      info.kind    = q.kind;
      info.path    = static_cast<uint32_t>(a->paths.size());
      a->queries.push_back(info);
    }
    if (r.paf.empty() && !r.queries.empty()) format_paf(r, asm_idx);
    a->paf += r.paf;
    r.info.query_end = static_cast<uint32_t>(a->queries.size());
    a->paths.push_back(r.info);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; }
  return MSGPU_OK;
}

// Host threads of msgpu_assembly_add_paths.  Starting 15 std::threads costs about as much as laying out a hundred
// paths, so they are started once per process and parked on a condition variable between calls (one fan-out at a
// time; the pool is leaked at exit on purpose -- parked threads need no clean-up -- and forgotten in a forked child).
class LayoutPool {
 public:
  ~LayoutPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : th_) t.join();
  }
  // runs job() on `helpers` pool threads and on the caller; returns when ALL of them are done -- also when one of them
  // threw (bad_alloc in a body): the helpers work on the caller's stack frame, so nobody leaves before the last one has
  // finished, and the first exception is rethrown on the caller afterwards
  void run(uint32_t helpers, const std::function<void()> &job) {
    while (th_.size() < helpers) th_.emplace_back([this, idx = static_cast<uint32_t>(th_.size())] { loop(idx); });
    {
      std::lock_guard<std::mutex> g(m_);
      job_     = &job;
      helpers_ = helpers;
      pending_ = helpers;
      ++gen_;
    }
    cv_work_.notify_all();
    std::exception_ptr mine;
    try {
      job();
    } catch (...) { mine = std::current_exception(); }
    std::unique_lock<std::mutex> g(m_);
    cv_done_.wait(g, [this] { return pending_ == 0; });
    job_ = nullptr;
    std::exception_ptr err = mine ? mine : err_;
    err_                   = nullptr;
    g.unlock();
    if (err) std::rethrow_exception(err);
  }

 private:
  void loop(uint32_t idx) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void()> *job = nullptr;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_work_.wait(g, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        if (idx >= helpers_) continue; // this call asked for fewer threads
        job = job_;
      }
      std::exception_ptr err;
      try {
        (*job)();
      } catch (...) { err = std::current_exception(); } // (an exception leaving a thread function is std::terminate)
      {
        std::lock_guard<std::mutex> g(m_);
        if (err && !err_) err_ = err;
        --pending_;
      }
      cv_done_.notify_one();
    }
  }

 public:
  std::mutex run_lock; // one fan-out at a time

 private:
  std::vector<std::thread>     th_;
  std::mutex                   m_;
  std::condition_variable      cv_work_, cv_done_;
  const std::function<void()> *job_     = nullptr;
  uint32_t                     helpers_ = 0, pending_ = 0;
  uint64_t                     gen_     = 0;
  bool                         stop_    = false;
  std::exception_ptr           err_;    // first exception of a helper in the current fan-out
};

std::atomic<LayoutPool *> g_pool{nullptr};
std::once_flag            g_pool_atfork;
LayoutPool               *layout_pool() {
  std::call_once(g_pool_atfork, [] { pthread_atfork(nullptr, nullptr, [] { g_pool.store(nullptr); }); });
  LayoutPool *p = g_pool.load();
  if (!p) {
    LayoutPool *fresh = new LayoutPool();
    if (g_pool.compare_exchange_strong(p, fresh))
      p = fresh;
    else
      delete fresh; // another thread was first (it has no workers yet, so this is cheap)
  }
  return p;
}

} // namespace

// MatchMap::getVertexMatch for every later path: one record per row, grouped by read (dense Registry ids) and, inside a
// read, ordered by (anchor, line) -- lowest line first, MatchMap.cpp:64-80.  5 M rows are a job of their own, and a plain
// counting sort by read scatters every record to a random place (one cache miss per row).  So it runs in two levels on
// the layout threads, every pass either sequential or inside a cache-sized block:
//   1. the chunks of the input count their rows per PARTITION (a few hundred consecutive read ids);
//   2. a prefix over (partition, chunk) hands every chunk its slots; the chunks copy their rows (input order) and append
//      a record {read << 32 | anchor, line, row number} to each partition -- a few hundred sequential write streams;
//   3. every partition (tens of thousands of records, cache resident) is counting-sorted by read into its final place
//      and its reads are put in (anchor, line) order (a PAF grouped by query is already in order there).
static int install_rows(msgpu_assembly *a, const msgpu_row *rows, size_t n_rows, bool copy) {
  if (!a || (n_rows && !rows)) return MSGPU_E_ARG;
  if (n_rows >= 0xffffffffull) return MSGPU_E_ARG;
  try {
    typedef msgpu_assembly::RowRec Rec;
    unsigned nt = std::thread::hardware_concurrency();
    nt          = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
    if (n_rows < (1u << 16)) nt = 1;
    LayoutPool                  *pool = layout_pool();
    std::lock_guard<std::mutex> one(pool->run_lock);
    auto chunk = [&](unsigned t) { return std::make_pair(n_rows * t / nt, n_rows * (t + 1) / nt); };
    std::atomic<size_t> next{0};
    auto                fan = [&](size_t n_items, const std::function<void(size_t)> &body) { // body(i), i < n_items, on the pool
      next.store(0);
      const std::function<void()> job = [&] {
        for (size_t i = next.fetch_add(1); i < n_items; i = next.fetch_add(1)) body(i);
      };
      if (nt == 1) job();
      else pool->run(nt - 1, job);
    };
    std::vector<uint32_t> cmax(nt, 0);
    std::vector<char>     asc(nt, 1); // chunk t's anchor ids never decrease (its first row compared with the row before it)
    std::vector<char>     rasc(nt, 1); // ... and inside an anchor its read ids never decrease
    a->rows.resize(copy ? n_rows : 0);
    a->rows_view = copy ? a->rows.data() : rows;
    a->row_recs.resize(0);
    a->row_start.clear();
    a->anchor_start.clear();
    // (the same pass notes where every anchor's rows start, as if the table were grouped by ascending anchor id -- it is, when
    // it comes from a PAF --: a chunk fills the starts of the anchors that begin inside it and of the absent ids before them,
    // for as long as its ids ascend; the result is used only if every chunk's did.  One pass over the 200 MB of configs[2]'s
    // table instead of two.)
    const size_t n_anchors_guess = n_rows ? static_cast<size_t>(rows[n_rows - 1].anchor_id) + 1 : 0;
    const bool   ahead = n_anchors_guess <= 2 * n_rows + 1024; // (a table in another order may end on any id: no table sized by it then)
    if (ahead) a->anchor_start.resize(n_anchors_guess + 1);
    uint64_t *st = a->anchor_start.data();
    fan(nt, [&](size_t t) {
      uint32_t m = 0;
      bool     up = true, rup = true;
      const size_t b = chunk(t).first, e = chunk(t).second;
      uint32_t     prev = b ? rows[b - 1].anchor_id : 0, prev_read = b ? rows[b - 1].read_id : 0;
      for (size_t i = b; i < e; ++i) {
        const uint32_t an = rows[i].anchor_id;
        m = std::max(m, rows[i].read_id);
        up &= prev <= an;
        rup &= prev != an || prev_read <= rows[i].read_id || i == 0;
        if (ahead && up && an < n_anchors_guess && (i == 0 || prev != an))
          for (size_t id = i == 0 ? 0 : static_cast<size_t>(prev) + 1; id <= an; ++id)
            __atomic_store_n(&st[id], static_cast<uint64_t>(i), __ATOMIC_RELAXED); // (chunks of a table in another order may meet on an id: the table is dropped then)
        prev      = an;
        prev_read = rows[i].read_id;
      }
      if (copy && e > b) memcpy(a->rows.data() + b, rows + b, (e - b) * sizeof(msgpu_row));
      cmax[t] = m;
      asc[t]  = up;
      rasc[t] = rup;
    });
    a->anchor_sorted_by_read = false;
    const size_t n_reads = n_rows ? static_cast<size_t>(*std::max_element(cmax.begin(), cmax.end())) + 1 : 0;
    if (n_rows && std::find(asc.begin(), asc.end(), 0) == asc.end()) {
      // Grouped by ascending anchor id -- what a PAF is (grouped by query, ids handed out in first-seen order): the
      // table needs no sort, only where each anchor's rows start.  Every chunk fills the starts of the anchors that
      // begin inside it (and of the absent ids before them): disjoint stretches of anchor_start.
      const size_t n_anchors = n_anchors_guess; // (ascending: the last row holds the highest id)
      if (!ahead) { // sparse ids: the starts in a pass of their own
        a->anchor_start.resize(n_anchors + 1);
        st = a->anchor_start.data();
        fan(nt, [&](size_t t) {
          for (size_t i = chunk(t).first; i < chunk(t).second; ++i) {
            const size_t from = i == 0 ? 0 : static_cast<size_t>(rows[i - 1].anchor_id) + 1;
            for (size_t id = from; id <= rows[i].anchor_id; ++id) st[id] = i;
          }
        });
      }
      st[n_anchors] = n_rows;
      a->anchor_sorted_by_read = std::find(rasc.begin(), rasc.end(), 0) == rasc.end();
      // Look-ups search an anchor's rows by read id (binary when they are in read order).  A table whose scaffolds are in
      // no order is kept in this form only while every scaffold is short (a scan of <= 64 rows); a deep one (repeat
      // anchors, high coverage: thousands of rows) takes the per-read table below, as any other order does.
      uint64_t longest = 0;
      if (!a->anchor_sorted_by_read)
        for (size_t id = 0; id < n_anchors; ++id) longest = std::max(longest, st[id + 1] - st[id]);
      if (longest <= 64) return MSGPU_OK;
    }
    a->anchor_start.clear();
    // partitions of 2^shift consecutive read ids, at most 1024 of them
    unsigned shift = 8;
    while ((n_reads >> shift) >= 1024) ++shift;
    const size_t          n_part = (n_reads >> shift) + 1;
    std::vector<uint64_t> cell(static_cast<size_t>(nt) * n_part, 0); // rows of chunk t in partition p, then their first slot
    fan(nt, [&](size_t t) {
      uint64_t *h = cell.data() + t * n_part;
      for (size_t i = chunk(t).first; i < chunk(t).second; ++i) ++h[rows[i].read_id >> shift];
    });
    std::vector<uint64_t> part_start(n_part + 1, 0);
    for (size_t p = 0; p < n_part; ++p) {
      uint64_t run = part_start[p];
      for (unsigned t = 0; t < nt; ++t) {
        uint64_t &c = cell[static_cast<size_t>(t) * n_part + p];
        const uint64_t k = c;
        c                = run;
        run += k;
      }
      part_start[p + 1] = run;
    }
    a->row_recs.resize(n_rows);
    msgpu_assembly::RawBuf<Rec> staged;
    staged.resize(n_rows);
    fan(nt, [&](size_t t) {
      uint64_t *h = cell.data() + t * n_part;
      for (size_t i = chunk(t).first; i < chunk(t).second; ++i) {
        const msgpu_row &row = rows[i];
        staged[h[row.read_id >> shift]++] =
            Rec{(static_cast<uint64_t>(row.read_id) << 32) | row.anchor_id, row.line, static_cast<uint32_t>(i)};
      }
    });
    std::vector<uint64_t> start(n_reads + 1, 0);
    Rec                  *recs = a->row_recs.data();
    auto less = [](const Rec &x, const Rec &y) { return x.key != y.key ? x.key < y.key : x.line < y.line; };
    fan(n_part, [&](size_t p) {
      const size_t r0 = p << shift, r1 = std::min(n_reads, (p + 1) << shift);
      if (r0 >= r1) return;
      const Rec *b = staged.data() + part_start[p], *e = staged.data() + part_start[p + 1];
      std::vector<uint64_t> cur(r1 - r0 + 1, 0);
      for (const Rec *q = b; q != e; ++q) ++cur[(q->key >> 32) - r0 + 1];
      cur[0] = part_start[p];
      for (size_t r = 0; r + 1 < cur.size(); ++r) cur[r + 1] += cur[r];
      for (size_t r = r0; r < r1; ++r) start[r] = cur[r - r0]; // partitions are consecutive in read order
      for (const Rec *q = b; q != e; ++q) recs[cur[(q->key >> 32) - r0]++] = *q; // input order inside a read is kept
      for (size_t r = r0; r < r1; ++r) {
        Rec *rb = recs + start[r], *re = recs + cur[r - r0];
        if (!std::is_sorted(rb, re, less)) std::sort(rb, re, less);
      }
    });
    start[n_reads] = n_rows;
    a->row_start   = std::move(start);
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; } catch (std::system_error const &) {
    return MSGPU_E_NOMEM; // could not start a thread
  }
  return MSGPU_OK;
}

int msgpu_assembly_set_rows(msgpu_assembly *a, const msgpu_row *rows, size_t n_rows) { return install_rows(a, rows, n_rows, true); }
int msgpu_assembly_borrow_rows(msgpu_assembly *a, const msgpu_row *rows, size_t n_rows) { return install_rows(a, rows, n_rows, false); }

int msgpu_assembly_add_path(msgpu_assembly *a, const msgpu_path_input *in) {
  if (!a) return MSGPU_E_ARG;
  if (a->finished) return MSGPU_E_STATE;
  a->err[0] = 0;
  PathResult  r;
  std::string msg;
  int         rc = try_layout(a, in, r, msg);
  if (rc == MSGPU_OK) rc = commit(a, r, in->asm_idx);
  if (rc != MSGPU_OK) snprintf(a->err, sizeof(a->err), "%s", msg.c_str());
  return rc;
}

// The assemblePaths fan-out (src/main.cpp:620-677: one job per path on the ThreadPool): the layouts of n paths are
// computed by n_threads host threads, then appended in input order.  status[i] (optional) = result for path i;
// paths with MSGPU_E_LAYOUT are skipped.  Returns the first status that is neither MSGPU_OK nor MSGPU_E_LAYOUT.
int msgpu_assembly_add_paths(msgpu_assembly *a, const msgpu_path_input *in, size_t n, uint32_t n_threads, int *status) {
  if (!a || (n && !in)) return MSGPU_E_ARG;
  if (a->finished) return MSGPU_E_STATE;
  a->err[0] = 0;
  if (!n_threads) n_threads = 1;
  if (n_threads > n) n_threads = static_cast<uint32_t>(n ? n : 1);
  std::vector<PathResult>  res;
  std::vector<std::string> msg;
  std::vector<int>         rc(n, MSGPU_OK);
  SecTimer                 wall;
  const auto               t_enter = std::chrono::steady_clock::now();
  try {
    res.resize(n);
    msg.resize(n);
    // longest paths first: a path of hundreds of reads laid out last would leave the other threads idle behind it
    std::vector<uint32_t> by_size(n);
    for (size_t i = 0; i < n; ++i) by_size[i] = static_cast<uint32_t>(i);
    std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return in[x].n_reads > in[y].n_reads; });
    std::atomic<size_t>         next{0};
    const std::function<void()> work = [&]() {
      for (size_t k = next.fetch_add(1); k < n; k = next.fetch_add(1)) {
        const size_t i = by_size[k];
        rc[i] = try_layout(a, &in[i], res[i], msg[i]);
        if (rc[i] == MSGPU_OK) {
          try {
            format_paf(res[i], in[i].asm_idx);
          } catch (std::bad_alloc const &) { rc[i] = MSGPU_E_NOMEM; }
        }
      }
    };
    if (n_threads > 1) {
      LayoutPool                 &pool = *layout_pool();
      std::lock_guard<std::mutex> one(pool.run_lock);
      pool.run(n_threads - 1, work);
    } else {
      work();
    }
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; } catch (std::system_error const &) {
    return MSGPU_E_NOMEM;
  }
  wall.mark(8);
  // Append the laid-out paths in input order (what commit() does for one path).  A path's records sit at 16-byte aligned
  // offsets behind a 16-byte aligned start, so its internal layout does not depend on what precedes it: sizes per path on
  // the layout threads, a prefix over the paths, then every path writes its pieces / query records / PAF lines into its
  // own stretch of the assembly's tables (for deep layouts -- 643 k query records, 1.5 M pieces on the tiled workload --
  // the appends of a serial loop were half of this call).
  int first_bad = MSGPU_OK;
  try {
    struct Span {
      uint64_t raw = 0, pieces = 0, queries = 0, paf = 0; // sizes, then (after the prefix) the path's first slot in each table
      uint32_t path = 0;
    };
    std::vector<Span>           span(n);
    std::atomic<size_t>         next{0};
    auto fan = [&](const std::function<void(size_t)> &body) {
      next.store(0);
      const std::function<void()> job = [&] {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1))
          if (rc[i] == MSGPU_OK) body(i);
      };
      if (n_threads > 1) {
        LayoutPool                 &pool = *layout_pool();
        std::lock_guard<std::mutex> one(pool.run_lock);
        pool.run(n_threads - 1, job);
      } else {
        job();
      }
    };
    auto count = [](const Seg &sg) {
      uint64_t k = 0;
      for (const msgpu_copy &p : sg.p) k += p.len != 0;
      return k;
    };
    fan([&](size_t i) {
      const PathResult &r = res[i];
      Span             &sp = span[i];
      sp.raw    = r.target.len;
      sp.pieces = count(r.target);
      for (const Record &q : r.queries) {
        sp.raw = ((sp.raw + 15) & ~15ull) + q.seg.len;
        sp.pieces += count(q.seg);
      }
      sp.queries = r.queries.size();
      sp.paf     = r.paf.size();
    });
    uint64_t raw = a->raw_bytes, np = a->pieces.size(), nq = a->queries.size(), nf = a->paf.size();
    uint32_t npath = static_cast<uint32_t>(a->paths.size());
    for (size_t i = 0; i < n; ++i) {
      if (rc[i] != MSGPU_OK) continue;
      Span          &sp = span[i];
      const uint64_t base = (raw + 15) & ~15ull;
      raw                 = base + sp.raw;
      const Span sizes    = sp;
      sp.raw = base, sp.pieces = np, sp.queries = nq, sp.paf = nf, sp.path = npath++;
      np += sizes.pieces, nq += sizes.queries, nf += sizes.paf;
    }
    a->pieces.resize(np);
    a->queries.resize(nq);
    a->paf.resize(nf);
    a->paths.resize(npath);
    fan([&](size_t i) {
      PathResult &r  = res[i];
      const Span &sp = span[i];
      uint64_t    at = sp.raw, pi = sp.pieces, qi = sp.queries;
      auto        put = [&](const Seg &sg) { // place(): the record's pieces, 16-byte aligned
        const uint64_t off = (at + 15) & ~15ull;
        for (msgpu_copy p : sg.p) {
          if (!p.len) continue;
          p.dst_off += off;
          a->pieces[pi++] = p;
        }
        at = off + sg.len;
        return off;
      };
      r.info.target_raw_off = put(r.target);
      r.info.query_begin    = static_cast<uint32_t>(sp.queries);
      for (const Record &q : r.queries) {
        msgpu_query_info info{};
        info.len     = q.seg.len;
        info.raw_off = put(q.seg);
        info.lb      = q.lb;
        info.rb      = q.rb;
        info.kind    = q.kind;
        info.path    = sp.path;
        a->queries[qi++] = info;
      }
      if (!r.paf.empty()) memcpy(&a->paf[sp.paf], r.paf.data(), r.paf.size());
      r.info.query_end  = static_cast<uint32_t>(qi);
      a->paths[sp.path] = r.info;
      // give the path's layout back here, on this thread: hundreds of thousands of small vectors freed one after the other
      // at the end of the call took as long as laying the paths out
      std::vector<Record>().swap(r.queries);
      Seg().p.swap(r.target.p);
      std::string().swap(r.paf);
    });
    a->raw_bytes = raw;
  } catch (std::bad_alloc const &) { return MSGPU_E_NOMEM; } catch (std::system_error const &) {
    return MSGPU_E_NOMEM;
  }
  for (size_t i = 0; i < n; ++i) {
    if (status) status[i] = rc[i];
    if (rc[i] != MSGPU_OK && a->err[0] == 0) snprintf(a->err, sizeof(a->err), "path %zu: %s", i, msg[i].c_str());
    if (rc[i] != MSGPU_OK && rc[i] != MSGPU_E_LAYOUT && first_bad == MSGPU_OK) first_bad = rc[i];
  }
  wall.mark(9);
  if (g_sec_on)
    fprintf(stderr, "[msgpu asm] add_paths(%zu paths, %u threads) returned after %.3f ms\n", n, n_threads,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count());
  return first_bad;
}

uint32_t msgpu_assembly_path_count(const msgpu_assembly *a) { return a ? static_cast<uint32_t>(a->paths.size()) : 0; }
uint32_t msgpu_assembly_query_count(const msgpu_assembly *a) { return a ? static_cast<uint32_t>(a->queries.size()) : 0; }
int      msgpu_assembly_path_info(const msgpu_assembly *a, uint32_t path, msgpu_path_info *out) {
  if (!a || !out || path >= a->paths.size()) return MSGPU_E_ARG;
  *out = a->paths[path];
  return MSGPU_OK;
}
int msgpu_assembly_query_info(const msgpu_assembly *a, uint32_t query, msgpu_query_info *out) {
  if (!a || !out || query >= a->queries.size()) return MSGPU_E_ARG;
  *out = a->queries[query];
  return MSGPU_OK;
}
size_t msgpu_assembly_pieces(const msgpu_assembly *a, msgpu_copy *out, size_t cap) {
  if (!a) return 0;
  if (out) std::copy_n(a->pieces.begin(), std::min(cap, a->pieces.size()), out);
  return a->pieces.size();
}
uint64_t msgpu_assembly_raw_bytes(const msgpu_assembly *a) { return a ? a->raw_bytes : 0; }

const char *msgpu_assembly_text(const msgpu_assembly *a, int which, uint64_t *len) {
  if (!a || which < 0 || which > 2 || (which < 2 && !a->finished)) {
    if (len) *len = 0;
    return nullptr;
  }
  if (which == 2) {
    if (len) *len = a->paf.size();
    return a->paf.data();
  }
  if (len) *len = which == 0 ? a->target_fa_len : a->query_fa_len;
  return a->text + (which == 0 ? 0 : a->query_fa_off);
}

} // extern "C"
