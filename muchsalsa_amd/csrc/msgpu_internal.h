// msgpu_internal.h -- types shared by the kernels (msgpu_kernels.hip) and the C-ABI host layer (msgpu_api.hip).
#ifndef MSGPU_INTERNAL_H
#define MSGPU_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msgpu.h"

namespace msgpu {

// 32-byte row of the two device tables (two 16-byte vector loads per row).
//   by_read  : rows of one read sorted by (n_lo, n_hi, anchor) -- `other` = anchor id, pf = flags | its place in by_anchor
//   by_anchor: rows of one anchor (scaffold) sorted by read id -- `other` = read id,   pf = flags | rank in its read
struct IRow {
  int32_t  n_lo, n_hi, i_lo, i_hi;
  uint32_t score, line, other, pf;
};
static_assert(sizeof(IRow) == 32, "IRow must be 32 bytes");
constexpr uint32_t PF_POS_MASK = 0x3fffffffu;
constexpr uint32_t PF_DIR      = 1u << 30;
constexpr uint32_t PF_PRIM     = 1u << 31;

// one owner read of an LDS class, as its candidate workgroup needs it (written by k_classify_reads)
struct CandDesc {
  uint32_t r, rb, n1, bound; // read id, first row in by_read, rows, scaffold rows to visit
  uint64_t co, pad;          // first slot of its candidate / edge scratch
};
static_assert(sizeof(CandDesc) == 32, "CandDesc must be 32 bytes");

struct CandArgs {
  const uint32_t *read_off, *read_cnt, *anchor_off;
  const IRow     *by_read, *by_anchor;
  const uint4    *vis;  // per by_read row:   {i_lo, i_hi, first scaffold row behind it, rows behind it}
  const uint64_t *cand_off;     // per read: first slot of its candidate / edge scratch (exclusive scan of bound)
  uint32_t       *cand_j;       // sorted candidates: rank of the anchor in v1's row list
  uint32_t       *cand_t;       // sorted candidates: row of v2 in by_anchor
  uint32_t       *edge_scr_v2;  // per-read edge scratch: v2
  uint32_t       *edge_scr_start; // per-read edge scratch: first candidate of the edge
  uint32_t       *n_cand, *n_edge; // per read
  unsigned long long *big_stats; // [0] edges with > 64 EdgeMatches, [1] their EdgeMatches (this launch set)
  uint32_t        th_overlap;
};
constexpr uint32_t CAND_CHUNK = 256;     // read ids per chunk of the candidate stage's sums = per workgroup of k_emit_edges
constexpr uint32_t SC_PUBLISH_MAX = 64;  // scalar words a fused read-back can publish (one wavefront)
// what k_classify_reads zeroes for the candidate kernels (they ADD to it)
struct CandZero {
  uint32_t *n_cand, *n_edge;   // per read, V + 1 entries
  uint32_t *scalar_words;      // big-edge statistics / cursors / class counts in the scalar block
  uint32_t  n_scalar_words;
};
// k_emit_edges: the candidate stage's closing launch (scans, edge table, size-sorted edge list, read-back)
struct EmitArgs {
  const uint32_t *n_edge, *n_cand;
  const uint64_t *cand_off;
  const uint32_t *edge_scr_v2, *edge_scr_start;
  uint32_t        V;
  const uint32_t *hist;
  const unsigned long long *chunk_sums;
  uint32_t        n_chunks;             // ceil(V / CAND_CHUNK)
  msgpu_edge     *edges;
  uint64_t       *edge_cand;
  uint32_t       *list;                 // the edges of <= 64 EdgeMatches by size, largest first
  uint32_t       *big_list;
  uint64_t       *big_off;
  unsigned long long *big_cursor;       // [2]
  const unsigned long long *big_stats;  // [2]
  uint64_t        cap_edges, cap_big;
  unsigned long long *chain_chunk_sums; // zeroed here for the chain kernels
  uint32_t        n_chain_chunk_words;
  uint64_t       *scalars, *host_scalars; // the read-back (see CompactArgs)
  uint32_t        slot_ems, slot_edges, slot_cls, n_scalars;
  uint64_t        seq;
  unsigned long long *nlists;           // the four list cursors of k_classify_reads (2 words): zero at rest, zeroed here
};

struct ChainArgs {
  msgpu_edge      *edges;
  const uint64_t  *edge_cand;
  uint64_t         n_edges;
  const uint32_t  *cand_j, *cand_t;
  const uint32_t  *read_off, *read_cnt;
  const int32_t   *read_len;
  const IRow      *by_read, *by_anchor;
  msgpu_edgematch *ems;
  msgpu_order     *order_scr; // slot em_off + i for the i-th order of an edge
  uint32_t        *ids_scr;   // slots em_off .. em_off + em_cnt of an edge
  uint32_t        *edge_norders, *edge_nids;
  uint32_t        *err;
  const uint32_t  *pair_tab; // four tables (sweep width 64, 32, 16, 8) of PAIR_TAB_STRIDE entries: k | l << 8 | run << 16
  const uint32_t  *pair_tab64; // the width-64 table with every field k_chain's sweep needs ready to use (k_fill_pair_tab)
  const uint32_t  *pair_tab_sub; // the width-32 / 16 / 8 tables, 8 bytes per pair, in the form k_chain_sub's sweep consumes
  unsigned long long *chunk_sums; // per chunk of COMPACT_CHUNK edges: {shortcut edges | orders << 32, ids} (see chunk_add)
  int              fast_path; // 0 disables the shortcut (every edge takes the full pair sweep)
  double           wiggle, ratio_pct, alt_frac;
  uint32_t         out_edge_base; // added to EdgeMatch::edge_idx: position of this batch's first edge in the job's table
};

constexpr uint32_t COMPACT_CHUNK = 1024; // edges per workgroup of k_compact = per pair of chunk sums of the chain kernels
struct CompactArgs {
  msgpu_edge        *edges;
  uint64_t           n_edges;
  const uint32_t    *edge_norders, *edge_nids;
  const unsigned long long *chunk_sums; // [2 n_chunks], left by the chain kernels
  uint32_t           n_chunks;          // ceil(n_edges / COMPACT_CHUNK)
  // the read-back of the table sizes rides on workgroup 0: the scalar block (device), its mapped host mirror (null: no
  // publication), the slots of the three sizes, the words to publish and the sequence number the host polls for (0: none)
  uint64_t          *scalars, *host_scalars;
  uint32_t           slot_orders, slot_ids, slot_fast, n_scalars;
  uint64_t           seq;
  const msgpu_order *order_scr;
  const uint32_t    *ids_scr;
  msgpu_order       *orders;
  uint32_t          *ids;
  // a batch of a larger job: what precedes this batch in the job's tables (0 for a whole-job run)
  uint64_t           out_em_base, out_order_base, out_ids_base;
  uint32_t           out_edge_base;
  uint64_t           cap_orders, cap_ids; // records `orders` / `ids` can hold (the launch may precede the size read-back)
};

constexpr uint32_t MAX_WORLD = 64;
struct MergeBase {
  uint64_t edges, orders, ids;
  uint32_t read_id, anchor_id; // added to the read ids (v1, v2, start, end, base) / anchor ids (id pool) of this rank's records
};
struct MergeArgs {
  const uint8_t *gathered;
  uint64_t       slab_bytes, off_edges, off_orders, off_ids;
  uint32_t       world;
  MergeBase      base[MAX_WORLD + 1]; // exclusive prefix over ranks; base[world] = totals
  msgpu_edge    *edges;
  msgpu_order   *orders;
  uint32_t      *ids;
};

// the exchange's wire form (include/msgpu.h, "wire form"): columns, 32-bit CSR offsets, nothing the receiver can derive
struct PackWireArgs {
  const msgpu_edge  *edges;
  const msgpu_order *orders;
  const uint32_t    *ids;      // non-null: the id block as 3-byte ids (four ids = three words), n_ids of them
  uint64_t           n_edges, n_orders, n_ids;
  uint8_t           *w_edges, *w_orders;
  uint32_t          *w_ids;
  // what the records' cross references carry beyond this table set (a window of the dispatcher: the job's records before it);
  // the wire offsets are relative to the set, so they fit 32 bits whatever came before
  uint64_t           base_ems = 0, base_orders = 0, base_ids = 0;
  uint32_t           base_edges = 0;
};
inline uint64_t wire_edges_bytes(uint64_t n) { return 17 * n + 8; }
inline uint64_t wire_orders_bytes(uint64_t n) { return 33 * n + 4; }
inline uint64_t wire_ids_bytes(uint64_t n, uint32_t id_bytes) { return id_bytes == 3 ? (3 * n + 3) / 4 * 4 : 4 * n; }
struct WindowCutArgs {
  uint32_t n;         // cuts wanted (windows - 1, at most 255)
  float    frac[255]; // cumulative share of the work in front of cut j
};
void launch_window_cuts(hipStream_t st, const uint64_t *cum, uint32_t V, const WindowCutArgs &a, uint64_t *out);
void launch_pack_wire(hipStream_t st, const PackWireArgs &a);
// the wire form back into records on the host (wire_host.cpp): what k_merge_wire does for one slab, with the bases added
void unpack_wire_host(const uint8_t *w_edges, const uint8_t *w_orders, const uint32_t *w_ids, uint32_t id_bytes, uint64_t n_edges,
                      uint64_t n_orders, uint64_t n_ids, uint64_t base_edges, uint64_t base_ems, uint64_t base_orders,
                      uint64_t base_ids, msgpu_edge *edges, msgpu_order *orders, uint32_t *ids, unsigned threads, unsigned tables = 7);
void launch_merge_wire(hipStream_t st, const MergeArgs &a, bool ids3);

template <class T> void exclusive_scan(hipStream_t st, const uint32_t *in, uint64_t n, T *out, T *block_sums, T *d_total);
uint32_t scan_blocks(uint64_t n);

void launch_max_ids(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t *max_ids);
void launch_expand_rows(hipStream_t st, const void *pk, uint64_t n, const int32_t *read_len, uint32_t n_reads, const uint32_t *run_start,
                        const uint32_t *run_delta, uint32_t n_runs, msgpu_row *out, uint32_t *err);
// index flags (device word): why the fast by_anchor path cannot be used
constexpr uint32_t IXF_UNSORTED = 1u; // rows are not strictly ascending in (anchor id, line)
constexpr uint32_t IXF_SPARSE   = 2u; // an anchor id has no row
constexpr uint32_t IXF_DUPS     = 4u; // a (read, anchor) pair occurs more than once
constexpr uint32_t IXF_FORCE    = 8u; // host asked for the generic path
constexpr uint32_t IXF_OVERFLOW = 16u; // one-pass build: a read has more rows than a bucket holds (host rebuilds in two passes)
constexpr uint32_t IXF_BIGSCAF  = 32u; // a scaffold longer than the context pass 1 sorts it in
constexpr uint32_t IXF_BINFAIL  = 64u; // bin path (msgpu_index.hip): a bucket beyond its capacity, a duplicate pair or a read of > 256 rows
// the bin path: rows binned by coarse bucket (consecutive read ids) without a global atomic per row, then a workgroup per bucket
constexpr uint32_t BIN_NB_MAX    = 8192; // coarse buckets per pass (k_index_bin's LDS histogram)
constexpr uint32_t BIN_RPB_SHIFT = 4;    // 16 read ids per coarse bucket
constexpr uint32_t BIN_PASSES_MAX = 8;   // passes over the row table (BIN_NB_MAX * 16 = 131,072 reads each); more reads: the atomic path
// what the last workgroup of k_index_bin needs to turn the bucket counts into bucket starts (the former k_bin_scan), and the
// two zero-at-rest counters of the launch
struct BinTail {
  uint32_t *bin_start;     // [nb + 1]
  uint32_t *row_base;      // by_read rows of the passes before this one (device word, updated for the next pass)
  uint32_t *read_off_end;  // &read_off[V] in the last pass, else null
  unsigned long long *done_heads; // low half: workgroups that have finished (the last one runs the tail and resets it); high
                                  // half: scaffolds that begin (first pass) -- k_index_epilogue compares it with the anchor count
};
// k_index_epilogue: everything element-wise behind the sort of a bin-path build in ONE launch -- the Registry-order check over
// the reads, the scaffold offsets over the anchors, the scan of the per-read visit counts (carries from the buckets' sums),
// the classification of the owner reads for the candidate kernels, and the read-back of flags and sizes by the last workgroup
struct IndexEpilogueArgs {
  const uint32_t *read_first;
  uint32_t        V, A, n_rows;
  uint32_t       *err, *flags;
  const uint32_t *anchor_first, *anchor_off_gen;
  uint32_t       *anchor_off, *n_alive;
  uint32_t       *heads, *row_base;       // consumed here (IXF_SPARSE), both zeroed for the next build
  const uint32_t *visits, *bucket_visits;
  uint32_t        n_buckets;
  uint64_t       *cand_off;               // [V + 1] out
  uint64_t       *total;                  // sum of all visits
  int             classify;               // 0: offsets and checks only (a window job classifies per window)
  const uint32_t *read_off, *read_cnt;
  uint32_t        shard, nshards;
  CandDesc       *list0, *list1, *list2;
  uint32_t       *list3, *n_lists;
  unsigned long long *own_total;          // visits of the reads classified (= total without shards)
  CandZero        z;
  uint32_t       *done;                   // zero at rest
  uint64_t       *scalars, *host_scalars;
  uint32_t        n_scalars;
  uint64_t        seq;
  uint64_t        zero_mask;              // scalar words zeroed behind the publication (bit = slot): error bits, flags, counters
};
void launch_index_epilogue(hipStream_t st, const IndexEpilogueArgs &a);
uint32_t bin_capacity(uint64_t n, uint32_t V); // rows a coarse bucket can hold; 0 = the bin path does not apply
bool index_sort_bin_prepare(uint32_t cap);   // k_index_sort_bin can be launched with such buckets on the current device (asks for > 64 KB of LDS where needed)
void launch_index_bin(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t V, uint32_t A, uint32_t *flags, uint32_t *err,
                      uint32_t *anchor_first, uint32_t *cursor, uint4 *bin_rec, uint32_t rd_lo, uint32_t nb, uint32_t cap,
                      const BinTail &tail);
void launch_index_sort_bin(hipStream_t st, uint32_t *cursor, const uint32_t *bin_start, uint32_t V, uint32_t rd_lo, uint32_t nb,
                           uint32_t cap, const uint4 *bin_rec, IRow *by_read, IRow *by_anchor, uint4 *vis, uint32_t *read_off,
                           uint32_t *read_cnt, int32_t *read_len, uint32_t *read_first, uint32_t *visits, const msgpu_row *rows,
                           uint32_t *flags, uint32_t *err, uint32_t *bucket_visits);
void launch_publish_scalars(hipStream_t st, const uint64_t *src, uint64_t *dst_host, uint32_t n, uint64_t seq);
void launch_index_init(hipStream_t st, uint32_t *const zero[4], const uint32_t n_zero[4], uint32_t *const ones[2],
                       const uint32_t n_ones[2]);
void launch_index_init8(hipStream_t st, uint32_t *const zero[8], const uint32_t n_zero[8], uint32_t *const ones[2],
                        const uint32_t n_ones[2]);
void launch_index_pass1(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t *cnt_read, uint32_t *anchor_first,
                        uint32_t V, uint32_t A, uint32_t *flags, uint32_t *err, IRow *bkt_row, uint32_t cap, uint2 *spos);
void launch_scatter_read(hipStream_t st, const msgpu_row *rows, uint64_t n, const uint32_t *read_off, uint32_t *cursor,
                         IRow *bkt_row);
void launch_sort_read(hipStream_t st, const uint32_t *read_off, const uint32_t *cnt_read, uint32_t V, const IRow *bkt_row,
                      IRow *by_read, uint32_t *read_cnt, uint32_t *alive_rank,
                      uint32_t *anchor_cnt, uint8_t *bkt_dead, uint32_t *flags, IRow *by_anchor, uint32_t cap,
                      const msgpu_row *rows, int32_t *read_len, uint32_t *read_first, uint32_t *err,
                      const uint2 *spos, uint4 *vis, uint32_t *visits);
void launch_index_finish(hipStream_t st, const uint32_t *read_first, uint32_t V, uint32_t *err, const uint32_t *flags,
                         const uint32_t *fast_off, const uint32_t *gen_off, uint32_t A, uint32_t *anchor_off, uint32_t *d_n_alive,
                         uint32_t n_rows);
void launch_select_anchor_off(hipStream_t st, const uint32_t *flags, const uint32_t *fast_off, const uint32_t *gen_off,
                              uint32_t A, uint32_t *anchor_off, uint32_t *d_n_alive, uint32_t n_rows);
void launch_scatter_anchor(hipStream_t st, const msgpu_row *rows, uint64_t n, const uint32_t *alive_rank,
                           const uint32_t *anchor_off, uint32_t *cursor, uint32_t *bkt_idx, uint32_t *bkt_line,
                           const uint32_t *flags);
void launch_rank_anchor(hipStream_t st, const uint32_t *anchor_off, uint64_t n_rows, const uint32_t *d_n_alive,
                        const uint32_t *bkt_idx, const uint32_t *bkt_line, const msgpu_row *rows,
                        const uint32_t *alive_rank, IRow *by_anchor, const uint32_t *flags, const uint32_t *read_off,
                        IRow *by_read, uint4 *vis);
void launch_bound(hipStream_t st, const uint32_t *read_off, const uint32_t *read_cnt, const uint4 *vis, uint32_t V,
                  uint32_t shard, uint32_t nshards, uint32_t lo, uint32_t hi, uint32_t *bound);
void launch_classify_reads(hipStream_t st, const uint32_t *read_off, const uint32_t *read_cnt, const uint32_t *bound,
                           const uint64_t *cand_off, uint32_t V, uint32_t shard, uint32_t nshards, uint32_t lo,
                           uint32_t hi, CandDesc *l0, CandDesc *l1, CandDesc *l2, uint32_t *l3, uint32_t *n_lists, const CandZero &z,
                           unsigned long long *own_total = nullptr);
void launch_candidates(hipStream_t st, const CandArgs &a, int cls, const CandDesc *list, uint32_t n_list);
void launch_candidates_big(hipStream_t st, const CandArgs &a, const uint32_t *list, uint32_t n_list, uint64_t *big_key,
                           uint32_t *big_t, uint32_t *big_r2s, uint32_t *big_pfx);
void launch_emit_edges(hipStream_t st, const EmitArgs &a, bool reduce); // reduce: k_cand_reduce in front (a repeat after a reallocation needs none)
constexpr uint32_t PAIR_TAB_STRIDE = 2016 + 128; // pairs k < l < 64 + padding read by lanes past the last pair
void launch_fill_pair_tab(hipStream_t st, uint32_t *tab);
void launch_chain(hipStream_t st, const ChainArgs &a, const uint32_t *list, uint32_t n_list);
void launch_chain_sub(hipStream_t st, const ChainArgs &a, int width, const uint32_t *list, uint32_t n_list);
struct ChainSubLists { // the three sub-wavefront classes of k_chain_sub_all: edge lists, their lengths, workgroups of the first two
  const uint32_t *list32, *list16, *list8;
  uint32_t        n32, n16, n8, nb32, nb16;
};
void launch_chain_sub_all(hipStream_t st, const ChainArgs &a, const uint32_t *l32, uint32_t n32, const uint32_t *l16, uint32_t n16,
                          const uint32_t *l8, uint32_t n8);
size_t big_elem_bytes();
size_t big_path_bytes();
void launch_chain_big(hipStream_t st, const ChainArgs &a, const uint32_t *big_list, const uint64_t *big_off,
                      uint32_t n_big, void *elems, void *paths);
void launch_merge_gathered(hipStream_t st, const MergeArgs &a);
// msgpu_graph.hip: findContractionEdges + sanityCheck
void launch_degree(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, uint32_t *deg);
void launch_fill_adj(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const uint64_t *adj_off,
                     uint32_t *cursor, uint32_t *adj);
void launch_mark_contained(hipStream_t st, const msgpu_order *orders, uint64_t n_orders, uint32_t *cand,
                           uint32_t *n_cand);
void launch_check_contraction(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const msgpu_order *orders,
                              const uint64_t *adj_off, const uint32_t *adj, const uint32_t *cand,
                              const uint32_t *n_cand, uint64_t n_orders, double wiggle, uint8_t *sane);
void launch_pick_contraction(hipStream_t st, const msgpu_edge *edges, uint64_t n_edges, const uint8_t *sane,
                             int64_t *out);
void launch_compact(hipStream_t st, const CompactArgs &a);
// msgpu_get_edgematches: the EdgeMatches of a list of edges out of the resident table
void launch_em_counts(hipStream_t st, const msgpu_edge *edges, const uint32_t *sel, uint64_t n, uint32_t *cnt);
void launch_em_gather(hipStream_t st, const msgpu_edge *edges, const msgpu_edgematch *ems, const uint32_t *sel, uint64_t n,
                      const uint64_t *off, msgpu_edgematch *out);

} // namespace msgpu

#endif
