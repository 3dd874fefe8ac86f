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
