// wire_host.cpp -- the exchange's wire form (msgpu_kernels.hip, "the exchange's wire form") back into records, on host threads.
//
// The dispatcher (msgpu_overlap_batched_ex) sends a window's edge / order / id tables over the host link in the form the
// multi-GPU exchange uses -- 17 + 33 bytes per edge + order instead of 32 + 64, three bytes per anchor id -- because from the
// first window on the link is what the job waits for; this is the receiving end: the same arithmetic as k_merge_wire for one
// slab (counts from CSR differences, start / end / base from the flags and the edge's vertices), plus the window's bases.
#include <emmintrin.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "asm_internal.h"
#include "host_pool.h"
#include "msgpu_internal.h"

namespace msgpu {

void unpack_wire_host(const uint8_t *w_edges, const uint8_t *w_orders, const uint32_t *w_ids, uint32_t id_bytes, uint64_t n_edges,
                      uint64_t n_orders, uint64_t n_ids, uint64_t base_edges, uint64_t base_ems, uint64_t base_orders,
                      uint64_t base_ids, msgpu_edge *edges, msgpu_order *orders, uint32_t *ids, unsigned threads, unsigned tables) {
  // the column views of WireEdges / WireOrders
  const uint32_t *e_em = reinterpret_cast<const uint32_t *>(w_edges), *e_or = e_em + (n_edges + 1), *e_v1 = e_or + (n_edges + 1),
                 *e_v2 = e_v1 + n_edges;
  const uint8_t  *e_sh = reinterpret_cast<const uint8_t *>(e_v2 + n_edges);
  const double   *o_l = reinterpret_cast<const double *>(w_orders), *o_r = o_l + n_orders;
  const uint64_t *o_sc = reinterpret_cast<const uint64_t *>(o_r + n_orders);
  const uint32_t *o_io = reinterpret_cast<const uint32_t *>(o_sc + n_orders), *o_ei = o_io + (n_orders + 1);
  const uint8_t  *o_fl = reinterpret_cast<const uint8_t *>(o_ei + n_orders);

  // records leave through streaming stores where the table is 16-byte aligned (the dispatcher's are): the tables are written
  // once and read much later, and a store that first reads the line it overwrites costs the host memory twice
  const bool nt_e = (reinterpret_cast<uintptr_t>(edges) & 15) == 0, nt_o = (reinterpret_cast<uintptr_t>(orders) & 15) == 0;
  auto put = [](void *dst, const void *rec, int quads, bool nt) {
    if (nt) {
      for (int q = 0; q < quads; ++q)
        _mm_stream_si128(static_cast<__m128i *>(dst) + q, _mm_loadu_si128(static_cast<const __m128i *>(rec) + q));
    } else {
      memcpy(dst, rec, size_t(quads) * 16);
    }
  };
  constexpr uint64_t PIECE = 1u << 15; // records per task (ids: four times as many)
  // (tables: 1 edges, 2 orders, 4 ids -- a caller whose blocks arrive one after the other turns each into records as it lands;
  // the orders need the edge BLOCK for their vertices, not the edge records)
  const uint64_t     pe = (tables & 1u) ? (n_edges + PIECE - 1) / PIECE : 0, po = (tables & 2u) ? (n_orders + PIECE - 1) / PIECE : 0,
                     pi = (tables & 4u) ? (n_ids + 4 * PIECE - 1) / (4 * PIECE) : 0;
  auto task = [&](size_t t) {
    if (t < pe) {
      const uint64_t lo = t * PIECE, hi = std::min(n_edges, lo + PIECE);
      for (uint64_t i = lo; i < hi; ++i) {
        msgpu_edge e;
        e.v1        = e_v1[i];
        e.v2        = e_v2[i];
        e.em_off    = base_ems + e_em[i];
        e.order_off = base_orders + e_or[i];
        e.em_cnt    = e_em[i + 1] - e_em[i];
        e.order_cnt = static_cast<uint16_t>(e_or[i + 1] - e_or[i]);
        e.shadow    = e_sh[i];
        e.pad       = 0;
        put(edges + i, &e, 2, nt_e);
      }
      _mm_sfence();
    } else if (t < pe + po) {
      const uint64_t lo = (t - pe) * PIECE, hi = std::min(n_orders, lo + PIECE);
      for (uint64_t i = lo; i < hi; ++i) {
        const uint32_t ei = o_ei[i], fl = o_fl[i], v1 = e_v1[ei], v2 = e_v2[ei];
        msgpu_order    o;
        o.edge_idx     = static_cast<uint32_t>(ei + base_edges);
        o.flags        = fl;
        o.left_offset  = o_l[i];
        o.right_offset = o_r[i];
        o.score        = o_sc[i];
        o.ids_off      = base_ids + o_io[i];
        o.ids_cnt      = o_io[i + 1] - o_io[i];
        o.start        = (fl & MSGPU_ORD_START_V1) ? v1 : v2;
        o.end          = (fl & MSGPU_ORD_START_V1) ? v2 : v1;
        o.base         = v1;
        o.pad[0]       = 0;
        o.pad[1]       = 0;
        put(orders + i, &o, 4, nt_o);
      }
      _mm_sfence();
    } else {
      const uint64_t lo = (t - pe - po) * 4 * PIECE, hi = std::min(n_ids, lo + 4 * PIECE); // (lo is a multiple of four)
      if (id_bytes == 4) {
        memcpy(ids + lo, w_ids + lo, (hi - lo) * 4);
      } else {
        uint64_t i = lo;
        for (; i + 4 <= hi; i += 4) { // four ids in three words
          const uint32_t *w = w_ids + 3 * (i >> 2);
          const uint32_t  a = w[0], b = w[1], c = w[2];
          ids[i]            = a & 0xffffffu;
          ids[i + 1]        = (a >> 24) | ((b & 0xffffu) << 8);
          ids[i + 2]        = (b >> 16) | ((c & 0xffu) << 16);
          ids[i + 3]        = c >> 8;
        }
        if (i < hi) { // the last group reaches only into the words its ids need
          const uint32_t *w = w_ids + 3 * (i >> 2);
          const uint64_t  rem = hi - i;
          const uint32_t  a = w[0], b = rem > 1 ? w[1] : 0, c = rem > 2 ? w[2] : 0;
          ids[i] = a & 0xffffffu;
          if (rem > 1) ids[i + 1] = (a >> 24) | ((b & 0xffffu) << 8);
          if (rem > 2) ids[i + 2] = (b >> 16) | ((c & 0xffu) << 16);
        }
      }
    }
  };
  HostPool::get().run(threads, pe + po + pi, task);
}

} // namespace msgpu

// ---- the row table's 28-byte form for the host link (include/msgpu.h, msgpu_row28 / msgpu_packed_rows) ---------------------------
// BlastFileReader.cpp:101-126 is what a row must hold; what the link need not carry is derived in HBM by k_expand_rows.
extern "C" int msgpu_pack_rows(const msgpu_row *rows, size_t n_rows, uint32_t n_reads, msgpu_packed_rows *out) {
  using namespace msgpu;
  if (!out || (n_rows && !rows) || n_rows >= (1ull << 32)) return MSGPU_E_ARG;
  memset(out, 0, sizeof(*out));
  // pass 1 (threads): what does not pack; the places where line - row index changes; first row of every read
  const unsigned nt = 16;
  const size_t   per = (n_rows + nt - 1) / nt;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> runs(nt); // (row, delta) where the delta differs from the row before
  std::vector<int>                                        bad(nt, 0);
  HostPool::get().run(nt, nt, [&](size_t t) {
    const size_t lo = std::min(n_rows, t * per), hi = std::min(n_rows, lo + per);
    for (size_t i = lo; i < hi; ++i) {
      const msgpu_row &r = rows[i];
      if (r.score >= (1u << 30) || r.read_id >= n_reads || r.line < i || (r.flags & ~3u)) {
        bad[t] = 1;
        return;
      }
      const uint32_t d = r.line - static_cast<uint32_t>(i);
      if (i == lo || d != rows[i - 1].line - static_cast<uint32_t>(i - 1)) runs[t].emplace_back(static_cast<uint32_t>(i), d);
    }
  });
  for (int b : bad)
    if (b) return MSGPU_E_ARG;
  // (a piece's first row always opens a run: merge it into the piece before where the delta did not change)
  std::vector<std::pair<uint32_t, uint32_t>> all;
  for (unsigned t = 0; t < nt; ++t)
    for (const auto &rn : runs[t])
      if (all.empty() || all.back().second != rn.second) all.push_back(rn);
  if (all.empty()) all.emplace_back(0u, 0u);
  const size_t n_runs = all.size();
  const size_t off_len = (n_rows * sizeof(msgpu_row28) + 63) / 64 * 64, off_rs = off_len + (size_t(n_reads) * 4 + 63) / 64 * 64,
               off_rd = off_rs + (n_runs * 4 + 63) / 64 * 64, total = off_rd + n_runs * 4 + 64;
  char *blk = static_cast<char *>(pinned_block_alloc(total));
  if (!blk) return MSGPU_E_NOMEM;
  msgpu_row28 *pr  = reinterpret_cast<msgpu_row28 *>(blk);
  int32_t     *len = reinterpret_cast<int32_t *>(blk + off_len);
  uint32_t    *rs = reinterpret_cast<uint32_t *>(blk + off_rs), *rd = reinterpret_cast<uint32_t *>(blk + off_rd);
  for (size_t k = 0; k < n_runs; ++k) {
    rs[k] = all[k].first;
    rd[k] = all[k].second;
  }
  HostPool::get().run(nt, nt, [&](size_t t) {
    const size_t lo = std::min(n_rows, t * per), hi = std::min(n_rows, lo + per);
    for (size_t i = lo; i < hi; ++i) {
      const msgpu_row &r = rows[i];
      pr[i] = msgpu_row28{r.anchor_id, r.read_id, r.i_lo, r.i_hi, r.n_lo, r.n_hi, r.score | ((r.flags & 1u) << 30) | ((r.flags & 2u) << 30)};
    }
  });
  // the Vertex' length is its FIRST line's (Graph.cpp:148): lowest line number among the read's rows
  std::vector<uint32_t> first_line(n_reads, 0xffffffffu);
  for (size_t i = 0; i < n_rows; ++i) {
    const msgpu_row &r = rows[i];
    if (r.line < first_line[r.read_id]) {
      first_line[r.read_id] = r.line;
      len[r.read_id]        = r.read_len;
    }
  }
  for (uint32_t v = 0; v < n_reads; ++v)
    if (first_line[v] == 0xffffffffu) len[v] = 0;
  out->rows      = pr;
  out->n_rows    = n_rows;
  out->read_len  = len;
  out->n_reads   = n_reads;
  out->n_runs    = static_cast<uint32_t>(n_runs);
  out->run_start = rs;
  out->run_delta = rd;
  out->owner     = blk;
  return MSGPU_OK;
}
extern "C" void msgpu_packed_rows_free(msgpu_packed_rows *p) {
  if (p && p->owner) msgpu::pinned_block_free(p->owner);
  if (p) memset(p, 0, sizeof(*p));
}

