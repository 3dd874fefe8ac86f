/*
 * msgpu.h -- C-ABI of the MI355X-native overlap-and-consensus core for MuCHSALSA (libmsgpu.so).
 *
 * Drop-in boundary for the reference's hot path.  Plain C, opaque contexts, caller-owned input buffers,
 * int status codes (0 = ok) + msgpu_*_last_error(); no exception crosses it.  Each entry point names the
 * reference interface it replaces (paths relative to the reference tree).
 *
 *   reference call site (src/main.cpp)                   replaced by
 *   ---------------------------------------------------  --------------------------------------------
 *   ThreadPool(threadCount)                      :143    msgpu_create          (HIP stream dispatcher)
 *   BlastFileAccessor + BlastFileReader::read()  :153-156 msgpu_parse_paf + msgpu_load_rows
 *   MatchMap::calculateEdges()                   :157    msgpu_calculate_edges
 *   SequenceAccessor + buildIndex()              :161-163 msgpu_seq_parse + msgpu_seq_upload (+ msgpu_seq_pack)
 *   for edge: Job(chainingAndOverlaps)           :170-178 msgpu_chaining_and_overlaps
 *   for edge: Job(findContractionEdges)          :183-190 msgpu_find_contraction_edges
 *   contraction ... decycle                      :194-288 msgpu_graph_create + msgpu_graph_clean_up
 *   getConnectedComponents + assemblePaths       :300-310, 620-661  msgpu_graph_linearize + msgpu_graph_path_input
 *   assemblePath per path + OutputWriter         :663-677 msgpu_assembly_add_paths + msgpu_assembly_finish / _text
 *   graph.getEdges()/Edge::getEdgeOrders()/...           msgpu_get_counts + msgpu_copy_tables
 *
 * Semantics are the single-thread reference's, bit for bit: int32 coordinates, IEEE fp64 scores/offsets
 * evaluated in the reference's expression order (kernels are built with -ffp-contract=off).
 */
#ifndef MSGPU_H
#define MSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSGPU_VERSION 1

/* ---- status codes ------------------------------------------------------------------------------------------- */
#define MSGPU_OK 0
#define MSGPU_E_IO (-1)       /* "Can't open blast file."   BlastFileAccessor.cpp:43-45                        */
#define MSGPU_E_FORMAT (-2)   /* "Invalid BLAST file."      BlastFileReader.cpp:97-99                          */
#define MSGPU_E_NUMBER (-3)   /* a field std::stoi would reject (BlastFileReader.cpp:101-116)                  */
#define MSGPU_E_NOMEM (-4)
#define MSGPU_E_ARG (-5)      /* "Unexpected nullptr." & friends (MatchMap.cpp:55-57)                          */
#define MSGPU_E_HIP (-6)      /* a HIP runtime call failed; text in msgpu_last_error                           */
#define MSGPU_E_STATE (-7)    /* entry points called out of order                                              */
#define MSGPU_E_IDS (-8)      /* read ids are not in first-line (Registry) order, Registry.cpp:36-45           */
#define MSGPU_E_NODEVICE (-9) /* no HIP device: the product has NO CPU fallback                                */
#define MSGPU_E_LAYOUT (-10)  /* assemblePath input the reference itself cannot assemble (it would terminate, hang
                                 or read past a container): text in msgpu_assembly_last_error                     */
#define MSGPU_E_TIMEOUT (-11) /* a deadline (msgpu_set_deadline, msgpu_group_set_timeout) passed with device work
                                 still queued: the reference's one catch at src/main.cpp:313-315 sees an error
                                 instead of a process that never returns                                          */

/* ---- records (byte-identical to oracle/ms_oracle.h) ----------------------------------------------------------- */

/* One ACCEPTED PAF line = what BlastFileReader::parseLine hands to Graph::addVertex and
 * MatchMap::addVertexMatch (BlastFileReader.cpp:101-126).  40 bytes. */
typedef struct msgpu_row {
  uint32_t anchor_id; /* illumina/unitig id, Registry first-seen order (registered second, :111)             */
  uint32_t read_id;   /* nanopore id, Registry first-seen order (registered first, :110)                     */
  int32_t  read_len;  /* col 6                                                                                */
  int32_t  i_lo, i_hi; /* VertexMatch::illuminaRange = (col2, col3-1)                                          */
  int32_t  n_lo, n_hi; /* VertexMatch::nanoporeRange = (col7, col8-1)                                          */
  uint32_t score;     /* VertexMatch::score = col 9                                                           */
  uint32_t line;      /* VertexMatch::lineNumber                                                              */
  uint32_t flags;     /* bit0 VertexMatch::direction, bit1 VertexMatch::isPrimary                             */
} msgpu_row;
#define MSGPU_ROW_DIR 1u
#define MSGPU_ROW_PRIMARY 2u

/* The same table as it crosses the host link: 28 bytes per row instead of 40.  What a loader-produced table carries that the
 * link need not (BlastFileReader.cpp:101-126 is what a row must hold): `read_len` once per READ instead of once per row (the
 * Vertex takes it from the read's first line, Graph.cpp:148); `line` as RUNS -- accepted lines are consecutive file lines
 * except where a line was rejected, so line = row index + a delta that changes at few places; the two flag bits in the top
 * bits of the score (a score of 2^30 or more does not pack: msgpu_pack_rows says so and the caller keeps the 40-byte form).
 * msgpu_load_rows_packed expands it in HBM (one kernel, ~0.1 ms for 5 M rows) to exactly the rows msgpu_load_rows would have
 * been given: every table downstream is the same, bit for bit. */
typedef struct msgpu_row28 {
  uint32_t anchor_id, read_id;
  int32_t  i_lo, i_hi, n_lo, n_hi;
  uint32_t score_flags; /* score | direction << 30 | isPrimary << 31 */
} msgpu_row28;
typedef struct msgpu_packed_rows {
  const msgpu_row28 *rows;      /* n_rows of them, in the order of the msgpu_row table                                  */
  uint64_t           n_rows;
  const int32_t     *read_len;  /* n_reads entries: msgpu_row::read_len of the read's FIRST row                         */
  uint32_t           n_reads, n_runs;
  const uint32_t    *run_start; /* n_runs ascending row indices, run_start[0] = 0: rows [run_start[k], run_start[k+1])  */
  const uint32_t    *run_delta; /* ... have line = row index + run_delta[k]                                             */
  void              *owner;     /* msgpu_pack_rows: the one page-locked block everything above lives in                 */
} msgpu_packed_rows;
/* Host code (a few threads): the packed form of a row table whose read ids are < n_reads.  Page-locked memory, to be given
 * back with msgpu_packed_rows_free.  MSGPU_E_ARG when the table does not pack (a score >= 2^30, a read id >= n_reads, a line
 * below its row index -- lines ascend with the rows in a loader's table) -- the 40-byte form takes every input. */
int  msgpu_pack_rows(const msgpu_row *rows, size_t n_rows, uint32_t n_reads, msgpu_packed_rows *out);
void msgpu_packed_rows_free(msgpu_packed_rows *p);

/* graph::Edge (Edge.h:212-218) as a table row.  32 bytes.  Table order: ascending (v1, v2). */
typedef struct msgpu_edge {
  uint32_t v1, v2;    /* Edge::getVertices(): v1 = read with the lower first line (MatchMap.cpp:204-213)      */
  uint64_t em_off;    /* this edge's EdgeMatches are ems[em_off .. em_off+em_cnt)                             */
  uint64_t order_off; /* this edge's EdgeOrders are orders[order_off .. order_off+order_cnt)                  */
  uint32_t em_cnt;
  uint16_t order_cnt;
  uint8_t  shadow;    /* Edge::isShadow (src/main.cpp:389-395)                                                */
  uint8_t  pad;
} msgpu_edge;

/* matching::EdgeMatch (MatchMap.h:68-74).  32 bytes.  Within an edge: ascending
 * (nanoporeRange on v1, anchor_id) = the vStart order of mpp.cpp:164-172. */
typedef struct msgpu_edgematch {
  int32_t  ov_lo, ov_hi; /* EdgeMatch::overlap                                                                */
  double   score;        /* EdgeMatch::score (MatchMap.cpp:200-202)                                           */
  uint32_t anchor_id;
  uint32_t line;         /* EdgeMatch::lineNumber = outerMatch->lineNumber (MatchMap.cpp:218)                 */
  uint32_t flags;        /* bit0 direction, bit1 isPrimary                                                    */
  uint32_t edge_idx;
} msgpu_edgematch;

/* graph::EdgeOrder (Edge.h:49-60).  64 bytes.  Within an edge: emission order of src/main.cpp:397-411
 * (minus-direction paths first, then plus). */
typedef struct msgpu_order {
  uint32_t edge_idx;
  uint32_t flags;        /* MSGPU_ORD_* */
  double   left_offset;
  double   right_offset;
  uint64_t score;        /* path score truncated like path_t's std::size_t (mpp.cpp:34,221,244)               */
  uint64_t ids_off;      /* EdgeOrder::ids = ids[ids_off .. ids_off+ids_cnt)                                  */
  uint32_t ids_cnt;
  uint32_t start, end, base; /* startVertex / endVertex / baseVertex as read ids                              */
  uint32_t pad[2];
} msgpu_order;
#define MSGPU_ORD_START_V1 1u  /* startVertex == v1 (else startVertex == v2, endVertex == v1)                 */
#define MSGPU_ORD_CONTAINED 2u /* EdgeOrder::isContained                                                      */
#define MSGPU_ORD_DIR 4u       /* EdgeOrder::direction                                                        */
#define MSGPU_ORD_PRIMARY 8u   /* EdgeOrder::isPrimary                                                        */

/* compile-time constants of the reference, exposed as parameters ("Config / flags") */
typedef struct msgpu_params {
  uint32_t min_matches; /* 400  MINIMUM_MATCHES  BlastFileReader.cpp:48 */
  uint32_t th_length;   /* 500  TH_LENGTH        BlastFileReader.cpp:49 */
  uint32_t th_matches;  /* 500  TH_MATCHES       BlastFileReader.cpp:50 */
  uint32_t th_overlap;  /* 100  TH_OVERLAP       MatchMap.cpp:41        */
  uint64_t wiggle_room; /* 300  Application::getWiggleRoom, Application.h:132 */
  double   ratio_pct;   /* 15   mpp.cpp:136 */
  double   alt_frac;    /* 0.75 mpp.cpp:223 */
} msgpu_params;

typedef struct msgpu_counts {
  uint64_t n_rows_in;    /* rows handed to msgpu_load_rows                                                    */
  uint64_t n_rows_alive; /* after the (read, anchor) lowest-line rule of MatchMap::addVertexMatch             */
  uint32_t n_reads;      /* Graph::getOrder()                                                                 */
  uint32_t n_anchors;    /* anchor id space                                                                   */
  uint64_t n_edges;      /* Graph::getSize() (this shard)                                                     */
  uint64_t n_ems;        /* EdgeMatches (this shard)                                                          */
  uint64_t n_orders;     /* EdgeOrders (this shard)                                                           */
  uint64_t n_ids;        /* sum of |EdgeOrder::ids| (this shard)                                              */
  uint64_t n_pairs_scanned; /* scaffold rows visited while looking for pairs (both directions of each pair)   */
  uint64_t n_edges_fastpath; /* edges whose pairs were all proven compatible without the O(n^2) sweep (this shard) */
  uint64_t n_lost_publications; /* size read-backs of this context whose publication into mapped host memory never arrived
                                   although the stream finished (the values were re-read by copy): 0 on a healthy system */
  uint64_t index_path;   /* how the last msgpu_load_rows built its index: MSGPU_INDEX_* (same tables whichever way)    */
} msgpu_counts;
#define MSGPU_INDEX_BIN 0u      /* rows binned by coarse read-id bucket, no global atomic per row: rows grouped by anchor with
                                   ascending lines and no duplicate (read, anchor) pair -- what msgpu_parse_paf produces      */
#define MSGPU_INDEX_ATOMIC 1u   /* one counter atomic per row, fixed-size bucket per read (rounds 1-3; MSGPU_NO_BIN=1 forces it) */
#define MSGPU_INDEX_TWO_PASS 2u /* count, scan, scatter: a read with more rows than a fixed bucket holds                      */
#define MSGPU_INDEX_GENERIC 4u  /* OR-ed in: scaffolds built by the generic per-anchor pass (any row order, duplicates)        */

/* Device time of the last run of each stage, milliseconds, measured with HIP events on the context's stream. */
typedef struct msgpu_timings {
  float index_ms;      /* msgpu_load_rows: MatchMap-equivalent index build (device part)                      */
  float candidates_ms; /* msgpu_calculate_edges: pair scan + group by edge                                    */
  float chain_ms;      /* msgpu_chaining_and_overlaps: EdgeMatch + chaining DP + overlap kernel (dominant)    */
  float compact_ms;    /* msgpu_chaining_and_overlaps: order/id compaction                                    */
  float chain_kernel_ms; /* the chain kernels alone (HIP events directly around their launches): mean over the   */
  uint32_t chain_kernel_launches; /* ... msgpu_chaining_and_overlaps calls since the last msgpu_get_timings (<= 256)    */
  uint32_t pad;
} msgpu_timings;

typedef struct msgpu_ctx msgpu_ctx;
typedef struct msgpu_paf msgpu_paf;

/* ---- life cycle ------------------------------------------------------------------------------------------------ */

void msgpu_default_params(msgpu_params *p);
const char *msgpu_strerror(int code);

/* Replaces ThreadPool(threadCount) + Graph + MatchMap construction (src/main.cpp:143-148).
 * `device` is a HIP device ordinal.  Fails with MSGPU_E_NODEVICE when there is none: there is no CPU path. */
int  msgpu_create(int device, const msgpu_params *params, msgpu_ctx **out);
void msgpu_destroy(msgpu_ctx *ctx);
const char *msgpu_last_error(const msgpu_ctx *ctx);

/* ---- STREAM AND THREAD CONTRACT -- who orders what at this boundary ---------------------------------------------------
 * (The reference states its notification contract where it has one: MatchMap observes the Graph's deletions,
 * include/ms/matching/MatchMap.h:200-207.  This boundary's equivalent is stream order.)
 *
 * 1. A context queues all its work on ONE stream: its own (created hipStreamNonBlocking: it does NOT synchronise with the
 *    legacy default stream, nor with any other stream, e.g. the one a tensor library fills buffers on) or the one given to
 *    msgpu_set_stream.  Entry points that take a `hip_stream` argument queue on that stream instead (NULL = the context's).
 * 2. HOST buffers (msgpu_load_rows, msgpu_copy_tables, msgpu_copy_reads, msgpu_find_contraction_edges' result,
 *    msgpu_overlap_batched[_ex], msgpu_get_edgematches): the call returns when the bytes are there / have been consumed.
 *    Nothing to order.
 * 3. DEVICE buffers of the caller -- read: msgpu_load_rows_device (d_rows, until the next load or msgpu_destroy),
 *    msgpu_overlap_batched_ex with MSGPU_BATCH_ROWS_ON_DEVICE, msgpu_merge_gathered[_ex] / msgpu_merge_wire (d_gathered),
 *    msgpu_find_contraction_edges (d_edges, d_orders); written: msgpu_copy_tables_device, msgpu_pack_wire,
 *    msgpu_merge_gathered[_ex] / msgpu_merge_wire (d_edges, d_orders, d_ids) -- are touched ASYNCHRONOUSLY, in the order of
 *    the stream of rule 1 and in no order with anything else.  The CALLER must therefore
 *      (a) have finished -- in stream order -- whatever it queued on those buffers before the call (a fill, a memset, the
 *          all-gather that produced d_gathered): otherwise the library's kernel and the caller's run concurrently and the
 *          buffer ends up a mixture of both;
 *      (b) not read, overwrite or free them before the library's work has finished -- in stream order.
 *    Three ways to meet (a) and (b), cheapest first:
 *      - msgpu_set_stream(ctx, the stream the caller works on): everything is one stream's order;
 *      - msgpu_stream_wait(ctx, s) before the call -- the context's stream waits for what is queued on s so far -- and
 *        msgpu_stream_release(ctx, s) after it -- s waits for what the context has queued so far (two event operations,
 *        no host wait);
 *      - a host wait: the caller synchronises its stream before the call and calls msgpu_synchronize(ctx) after it.
 *    A sequence store (msgpu_seqctx, below) is a context of its own with a stream of its own, and the same rule holds for
 *    the caller's device buffers it touches: msgpu_seq_upload_device (d_bases, read), msgpu_gather_run (d_out, written),
 *    msgpu_fasta_format (d_raw read, d_text written), msgpu_edit_distance (d_a, d_b read); its entry points take the stream to
 *    queue on as `hip_stream` (NULL = the store's own), msgpu_seq_synchronize is its host wait.
 * 4. msgpu_last_error is meaningful only after a call returned a non-zero code.
 * 5. A context is NOT thread-safe: one host thread at a time.  One exception, what an exchange thread needs:
 *    msgpu_merge_gathered_ex / msgpu_merge_wire with a hip_stream of the caller's may run on a second thread beside any
 *    other call (they touch no state of the context except, on failure, the error text). */

/* Run all work of this context on an existing HIP stream (hipStream_t), e.g. the caller's current stream.
 * NULL restores the context's own stream.  Waits for the stream used so far. */
int msgpu_set_stream(msgpu_ctx *ctx, void *hip_stream);
/* The hipStream_t the context queues on at the moment (its own unless msgpu_set_stream changed it). */
void *msgpu_get_stream(const msgpu_ctx *ctx);
/* Rule 3: the context's stream waits for everything queued so far on `hip_stream` (NULL = the legacy default stream);
 * `hip_stream` waits for everything the context has queued so far.  Neither blocks the host. */
int msgpu_stream_wait(msgpu_ctx *ctx, void *hip_stream);
int msgpu_stream_release(msgpu_ctx *ctx, void *hip_stream);

/* Multi-GPU: this context owns the edges whose v1 satisfies v1 % n_shards == shard (default 0 of 1).
 * Every shard loads the full row table; edges/orders of different shards are disjoint and their union is the
 * 1-GPU result.  Must be called before msgpu_calculate_edges. */
int msgpu_set_shard(msgpu_ctx *ctx, uint32_t shard, uint32_t n_shards);

/* Optional: declare the id spaces of the rows that will be loaded (n_reads = highest read id + 1, n_anchors likewise;
 * msgpu_paf_read_count / msgpu_paf_anchor_count of the loader, i.e. Registry::m_ui32Size, Registry.cpp:36-45).  The
 * index build then skips its own pass over the table and one read-back.  Stays in force for later loads; (0, 0)
 * returns to discovery.  A row with an id outside the declared space makes the load fail with MSGPU_E_IDS. */
int msgpu_set_id_space(msgpu_ctx *ctx, uint32_t n_reads, uint32_t n_anchors);

/* ---- A1: PAF loader (host) ------------------------------------------------------------------------------------- */

/* Replaces BlastFileAccessor::_buildIndex (BlastFileAccessor.cpp:77-91) + BlastFileReader::read/parseLine
 * (BlastFileReader.cpp:72-130): indexes all lines, parses all but the LAST one (:76), keeps a line iff
 * col9 >= min_matches and col3-col2 >= min_matches, assigns Registry ids in first-seen order. */
int  msgpu_parse_paf(const char *path, const msgpu_params *params, msgpu_paf **out);
void msgpu_paf_free(msgpu_paf *paf);
const msgpu_row *msgpu_paf_rows(const msgpu_paf *paf, size_t *n_rows);
size_t      msgpu_paf_line_count(const msgpu_paf *paf);
uint32_t    msgpu_paf_read_count(const msgpu_paf *paf);
uint32_t    msgpu_paf_anchor_count(const msgpu_paf *paf);
const char *msgpu_paf_read_name(const msgpu_paf *paf, uint32_t read_id);     /* Registry reverse lookup */
const char *msgpu_paf_anchor_name(const msgpu_paf *paf, uint32_t anchor_id);

/* Registry::operator[] for every record of a parsed sequence file, on the registries of `paf` (the reference's
 * SequenceAccessor calls the very Registry objects BlastFileReader filled: SequenceAccessor.cpp:171,215, src/main.cpp:
 * 149-163): a name the PAF registered keeps its id, an unknown name takes the next free id in file order (the registry
 * grows).  kind 0 = reads (nanopore registry), 1 = unitigs (illumina registry).  ids: msgpu_seq_count(f) entries, what
 * msgpu_seq_upload takes; *id_space (optional) = the registry's size afterwards. */
struct msgpu_seqfile;
int msgpu_paf_register_sequences(msgpu_paf *paf, int kind, const struct msgpu_seqfile *f, uint32_t *ids, uint32_t *id_space);

/* Host utilities of libms that the loader is made of, on their own (the reference's unit tests hold vectors for them:
 * libms/tests/IO_test.cpp:12-35, Registry_test.cpp:5-14, Toggle_test.cpp:5-25; replayed from tests/golden/ref_tests/).
 * msgpu_index_lines: the line index of BlastFileAccessor::_buildIndex (BlastFileAccessor.cpp:77-91) over readline
 * (IO.cpp:54-97): every '\n' ends a line and belongs to it, a non-empty tail without '\n' is a line.  offsets (capacity
 * entries, may be NULL with capacity 0 to count) receives the start of every line and, if it fits, the file size after
 * the last one.  msgpu_registry_*: Registry (Registry.cpp:36-52), dense ids in first-seen order; clear() restarts at 0.
 * msgpu_toggle_mul: Toggle::operator* (Toggle.h:127-153) = XNOR. */
int msgpu_index_lines(const char *path, uint64_t *offsets, size_t capacity, size_t *n_lines);
typedef struct msgpu_registry msgpu_registry;
msgpu_registry *msgpu_registry_new(void);
void     msgpu_registry_free(msgpu_registry *r);
uint32_t msgpu_registry_id(msgpu_registry *r, const char *name); /* operator[]; 0xffffffff on error */
uint32_t msgpu_registry_size(const msgpu_registry *r);
void     msgpu_registry_clear(msgpu_registry *r);
int      msgpu_toggle_mul(int a, int b);

/* ---- A1 tail: fill the device-resident MatchMap/Graph-vertex equivalent ---------------------------------------- */

/* Replaces the effect of BlastFileReader::read() on Graph (addVertex: first line wins, Graph.cpp:148) and
 * MatchMap (addVertexMatch: lowest line per (read, anchor) wins, MatchMap.cpp:52-81).
 * `rows` may be in any order and may contain (read, anchor) duplicates.  Read ids must follow first-line
 * order (what msgpu_parse_paf produces), else MSGPU_E_IDS.  Host buffer; copied to HBM. */
int msgpu_load_rows(msgpu_ctx *ctx, const msgpu_row *rows, size_t n_rows);
/* msgpu_load_rows for the 28-byte form: packed rows over the link (page-locked: msgpu_pack_rows), expanded in HBM. */
int msgpu_load_rows_packed(msgpu_ctx *ctx, const msgpu_packed_rows *packed);
/* Same, rows already resident in HBM (device pointer, n_rows * 40 bytes).  The buffer is only read -- by this call and by
 * every later stage until the next load (STREAM CONTRACT rule 3: the rows must be complete in the context's stream order). */
int msgpu_load_rows_device(msgpu_ctx *ctx, const void *d_rows, size_t n_rows);

/* ---- A2/A3: MatchMap::calculateEdges (MatchMap.cpp:161-224) ----------------------------------------------------- */

/* All pairs of reads sharing an anchor, overlap test (> th_overlap), grouping by edge.  After it returns
 * msgpu_get_counts().n_edges / .n_ems are valid (Graph::getSize()). */
int msgpu_calculate_edges(msgpu_ctx *ctx);

/* ---- A4..A7: the chainingAndOverlaps fan-out (src/main.cpp:170-178, 328-414) ------------------------------------ */

/* EdgeMatch scores, getMaxPairwisePaths (mpp.cpp:145-305) for both directions, the primary/multi filters,
 * Edge::setShadow, getOverlap (ol.cpp:53-101), Edge::appendOrder -- for every edge of this shard. */
int msgpu_chaining_and_overlaps(msgpu_ctx *ctx);

/* ---- results ----------------------------------------------------------------------------------------------------- */

int msgpu_get_counts(msgpu_ctx *ctx, msgpu_counts *out);
int msgpu_get_timings(msgpu_ctx *ctx, msgpu_timings *out);
/* The stage boundaries (index / candidates / chain / compact) are marked with HIP events on the context's stream; every
 * marker costs a few microseconds of command-processor time.  on = 0 drops them: msgpu_get_timings then reports only
 * chain_kernel_ms (the two events around the chain kernels stay) and zeros for the stages.  Default: on. */
int msgpu_set_stage_events(msgpu_ctx *ctx, int on);

/* Copy result tables to HOST buffers sized from msgpu_get_counts (any pointer may be NULL to skip). */
int msgpu_copy_tables(msgpu_ctx *ctx, msgpu_edge *edges, msgpu_edgematch *ems, msgpu_order *orders, uint32_t *ids);
/* Same into DEVICE buffers (device-to-device on the context's stream; used in front of the RCCL all-gather).
 * Asynchronous: STREAM CONTRACT rule 3 applies to the destination buffers. */
int msgpu_copy_tables_device(msgpu_ctx *ctx, void *d_edges, void *d_ems, void *d_orders, void *d_ids);
/* Per-read Vertex facts: Vertex::getNanoporeLength() and metaDatum(0) (first line), n_reads entries each (host). */
int msgpu_copy_reads(msgpu_ctx *ctx, int32_t *read_len, uint32_t *read_first_line);

/* Multi-GPU merge of the edge list (north_star: "a single RCCL all-gather over xGMI to merge the edge list").
 * Input: the result of ONE all-gather of equally sized slabs, slab r (from rank r) at d_gathered + r*slab_bytes,
 * holding that rank's edge / order / id tables at off_edges / off_orders / off_ids (padding after each is ignored).
 * counts = world x {n_edges, n_orders, n_ids} (host).  Output (device): dense rank-major tables with order_off,
 * edge_idx and ids_off re-based to the merged tables; em_off stays rank-local (EdgeMatch tables are not gathered).
 * The reference has no counterpart (single process); consumed like graph.getEdges() + Edge::getEdgeOrders().
 * Asynchronous: STREAM CONTRACT rule 3 applies to d_gathered (read) and to d_edges / d_orders / d_ids (written). */
int msgpu_merge_gathered(msgpu_ctx *ctx, const void *d_gathered, uint32_t world, const uint64_t *counts,
                         uint64_t slab_bytes, uint64_t off_edges, uint64_t off_orders, uint64_t off_ids, void *d_edges,
                         void *d_orders, void *d_ids);

/* The same with (a) id bases: id_base = world x {read id base, anchor id base} (NULL = zeros) is added to the read ids
 * (v1, v2, start, end, base) and to the anchor ids of rank r's records -- the ranks hold PARTITIONS of a larger job
 * (disjoint sets of reads and anchors, e.g. chromosomes: no edge crosses a partition), each with ids from 0, and the merged
 * list is the larger job's, (v1, v2)-sorted when the bases ascend; (b) the stream the merge kernel runs on (NULL = the
 * context's), so that an exchange can run on its own stream beside the next batch's compute. */
int msgpu_merge_gathered_ex(msgpu_ctx *ctx, const void *d_gathered, uint32_t world, const uint64_t *counts,
                            uint64_t slab_bytes, uint64_t off_edges, uint64_t off_orders, uint64_t off_ids,
                            const uint32_t *id_base, void *d_edges, void *d_orders, void *d_ids, void *hip_stream);

/* The exchange's WIRE FORM: the same merge with about 30 % fewer bytes over xGMI.  What a receiver can derive is not sent:
 * a rank's tables are dense (every record's slice of the next table starts where the one before ends), so the 64-bit
 * offsets travel as 32-bit CSR columns of n + 1 entries and the counts are their differences; an EdgeOrder's start / end /
 * base vertices follow from its flags and its edge (src/main.cpp:397-411: base = the edge's first vertex); padding is not
 * sent.  Columns, in this order, every block densely packed for the rank's OWN record count n:
 *   edge block  (msgpu_wire_edges_bytes(n)  = 17 n + 8): em_off[n+1] u32 | order_off[n+1] u32 | v1[n] | v2[n] | shadow[n] u8
 *   order block (msgpu_wire_orders_bytes(n) = 33 n + 4): left[n] f64 | right[n] f64 | score[n] u64 | ids_off[n+1] u32 |
 *                                                        edge_idx[n] u32 | flags[n] u8
 *   id block    (msgpu_wire_ids_bytes(n, id_bytes)): id_bytes = 4: as in the tables (4 n); id_bytes = 3, while the job's
 *               anchor ids fit 24 bits: four ids in three words (3 n, rounded up to a word) -- the ids are half of the
 *               wire slab, so this takes another 13 % off it.  Every rank of an exchange must use the same id_bytes.
 * msgpu_pack_wire writes the context's tables (after msgpu_chaining_and_overlaps) in that form into three DEVICE blocks
 * (edge and id blocks 4-byte, order block 8-byte aligned) on the context's stream; MSGPU_E_ARG when a table has more than
 * 2^32 - 1 EdgeMatches, orders or ids (exchange such tables whole) or, with id_bytes = 3, an anchor id space beyond 2^24.  msgpu_merge_wire = msgpu_merge_gathered_ex over
 * slabs whose three blocks are in wire form: the merged tables are the same, byte for byte (d_edges and d_orders 16-byte
 * aligned: the records leave as whole lines).
 * Both are asynchronous: STREAM CONTRACT rule 3 applies to the three blocks msgpu_pack_wire writes (a fill of those
 * buffers queued on another stream races with the pack kernel) and to msgpu_merge_wire's input and output buffers.
 * A table without records still sends its closing CSR entries (zeros): 8 bytes of the edge block, 4 of the order block. */
uint64_t msgpu_wire_edges_bytes(uint64_t n_edges);
uint64_t msgpu_wire_orders_bytes(uint64_t n_orders);
uint64_t msgpu_wire_ids_bytes(uint64_t n_ids, uint32_t id_bytes);
int msgpu_pack_wire(msgpu_ctx *ctx, void *d_wire_edges, void *d_wire_orders, void *d_ids, uint32_t id_bytes);
int msgpu_merge_wire(msgpu_ctx *ctx, const void *d_gathered, uint32_t world, const uint64_t *counts, uint64_t slab_bytes,
                     uint64_t off_edges, uint64_t off_orders, uint64_t off_ids, uint32_t id_bytes, const uint32_t *id_base,
                     void *d_edges, void *d_orders, void *d_ids, void *hip_stream);
/* The receiving end on the HOST: one set of wire blocks (host memory) back into records, on up to `threads` host threads (0 = 16).
 * base[4] = what precedes the set in the tables its records point into {edges, EdgeMatches, orders, ids}: added to edge_idx,
 * em_off, order_off, ids_off (NULL = zeros: the records of msgpu_copy_tables for the same context).  tables: which of them to
 * write -- 1 edges, 2 orders, 4 ids, 0 = all; a caller whose blocks arrive one after the other unpacks each as it lands (the
 * orders read the edge BLOCK for their vertices, not the edge records).  msgpu_overlap_batched_ex uses it for its windows
 * when the EdgeMatch table stays in HBM, msgpu_group_overlap for every member's slab (the tables travel over the host link in
 * wire form).  No GPU, no context: plain host code. */
int msgpu_unpack_wire_host(const void *wire_edges, const void *wire_orders, const void *wire_ids, uint32_t id_bytes,
                           uint64_t n_edges, uint64_t n_orders, uint64_t n_ids, const uint64_t *base, msgpu_edge *edges,
                           msgpu_order *orders, uint32_t *ids, uint32_t threads, uint32_t tables);

/* ---- one process, the node's GPUs: a GROUP of contexts behind the same call site -------------------------------------------
 * The reference is ONE process that fans jobs over its workers and closes each phase with a barrier (src/main.cpp:143-178,
 * libms/src/threading/ThreadPool.cpp:38-129, WaitGroup.cpp:62-72).  Its multi-GPU equivalent: one process, one context per
 * device, one host thread per device for the duration of a call, and ONE collective on the results (with several members a
 * second, input-side one completes the row table in every HBM).  msgpu_group_overlap =
 *   rows (host) -> a 1/n-th over each device's own link, completed in every HBM by a grouped in-place all-gather over xGMI
 *   (MSGPU_GROUP_ROWS=replicate at creation, a group of one, or fewer than 1024 rows: the whole table over every link) ->
 *   index build on every device (replicated: a member needs
 *   the rank of every row inside its read) -> device i computes the edges with v1 % n == i (msgpu_set_shard) -> its edge /
 *   order / id tables in WIRE FORM into its slab (msgpu_pack_wire) -> ONE grouped RCCL all-gather over xGMI
 *   (ncclGroupStart / n x ncclAllGather / ncclGroupEnd, each on its member's stream) -> msgpu_merge_wire on every device:
 *   the merged edge list of the job in every HBM, and in host memory (every member sends its OWN slab, still in wire form,
 *   over its own link beside the exchange; host threads turn the slabs into the merged records: msgpu_unpack_wire_host with
 *   the member's bases); WaitGroup::wait() = the join of the member threads.
 * The merged tables are the ones msgpu_merge_wire defines: rank-major (member 0's edges in (v1, v2) order, then member 1's ...),
 * order_off / edge_idx / ids_off re-based to the merged tables, em_off local to the owning member (EdgeMatch tables are not
 * gathered: msgpu_get_edgematches on msgpu_group_ctx(g, v1 % n)).  With n = 1 they are the single-context tables bit for bit.
 * RCCL is loaded on first use (dlopen of librccl.so.1: the process's own copy where a framework already brought one); a
 * process that never creates a group never maps it.  MSGPU_E_NODEVICE when a device is missing, MSGPU_E_HIP with the RCCL
 * error text when a collective fails.  A group is driven by one host thread at a time (STREAM AND THREAD CONTRACT rule 5);
 * the members' contexts must not be used by the caller while a group call runs.
 * Rehearsal on a box with fewer GPUs than members: with MSGPU_GROUP_TRANSPORT=copy in the environment when the group is
 * created, the all-gather is carried by device-to-device copies of this process instead of RCCL and members may share a device;
 * shards, threads, slab layout, pack and merge are the same code.  For tests; never the default. */
typedef struct msgpu_group msgpu_group;
typedef struct msgpu_group_tables {
  const msgpu_edge  *edges;   /* host (pinned, owned by the group, valid until its next call): the merged edge list     */
  const msgpu_order *orders;
  const uint32_t    *ids;
  const int32_t     *read_len;        /* Vertex::getNanoporeLength(), n_reads entries                                  */
  const uint32_t    *read_first_line; /* Vertex::metaDatum(0)                                                          */
  uint64_t n_edges, n_orders, n_ids, n_ems; /* n_ems: EdgeMatches over all members (the tables stay in their HBMs)     */
  uint32_t n_reads, n_anchors, n_members, id_bytes; /* id_bytes: 3 or 4, how anchor ids travelled                      */
  uint64_t slab_bytes;        /* bytes every member sent (the largest member's wire blocks)                            */
  float wall_ms;              /* host clock: call entry -> merged tables in host memory                                */
  float compute_ms;           /*   slowest member: rows in HBM + index + its shard                                     */
  float exchange_ms;          /*   slowest member: pack + all-gather + merge (device time, HIP events)                 */
  uint32_t rows_sliced;       /* 1: every member took a 1/n-th of the rows over its link, an all-gather did the rest   */
} msgpu_group_tables;
int  msgpu_group_create(const int *devices, int n, const msgpu_params *params, msgpu_group **out);
void msgpu_group_destroy(msgpu_group *g);
const char *msgpu_group_last_error(const msgpu_group *g);
int  msgpu_group_size(const msgpu_group *g);
msgpu_ctx *msgpu_group_ctx(msgpu_group *g, int member); /* member i's context: its counts, its EdgeMatch table, its merged device tables */
int  msgpu_group_overlap(msgpu_group *g, const msgpu_row *rows, size_t n_rows, msgpu_group_tables *out);
/* A way out of a collective that never completes.  timeout_ms > 0: msgpu_group_overlap gives up that long after its entry --
 * every host wait of the call (the members' table sizes, the slabs landing in host memory, the closing barrier) polls instead of
 * blocking -- then aborts the members' communicators (ncclCommAbort: the collective's kernels leave the streams), waits at most
 * the same time again for the streams to drain, and returns MSGPU_E_TIMEOUT with the text saying whether they did.  The next call
 * builds fresh communicators.  0 = wait for ever (the default; MSGPU_GROUP_TIMEOUT_MS in the environment at creation sets
 * another).  The one-catch error model of src/main.cpp:313-315: the caller sees a code, not a process that hangs.
 * On ANY error return the group has synchronised what it queued where that was possible (nothing of the call reads `rows` or
 * writes the group's host tables any more unless the text says the streams did not drain); the calling thread's current HIP
 * device is restored on every return. */
int  msgpu_group_set_timeout(msgpu_group *g, uint32_t timeout_ms);
/* The merged tables as member `member` holds them in ITS HBM (device pointers, valid until the group's next call): what
 * msgpu_find_contraction_edges(msgpu_group_ctx(g, member), d_edges, n_edges, d_orders, n_orders, n_reads, ...) takes. */
int  msgpu_group_device_tables(msgpu_group *g, int member, const void **d_edges, const void **d_orders, const void **d_ids);

/* ---- the ThreadPool replacement: the whole overlap path, host memory to host memory, as batches on two HIP streams ----
 * Replaces the phases of src/main.cpp:153-178 that the reference fans over its ThreadPool (one Job per PAF line, per
 * anchor, per edge; libms/src/threading/ThreadPool.cpp:38-129) and closes with WaitGroup::wait() (WaitGroup.cpp:62-72):
 *   rows -> HBM and index build once (msgpu_load_rows), then n_batches windows of owner reads, each one
 *   msgpu_calculate_edges + msgpu_chaining_and_overlaps on the compute stream, while the previous window's edge /
 *   EdgeMatch / order / id tables are copied to pinned host memory on a second stream (two table sets in HBM).
 * The call returns when every batch is done (the phase barrier).  The host tables are the single-pass tables of
 * msgpu_copy_tables bit for bit -- canonical order, cross references (em_off, order_off, edge_idx, ids_off) into the
 * whole tables -- owned by the context and valid until its next msgpu_overlap_batched / msgpu_destroy.  HBM holds one
 * window's tables at a time instead of the job's.  With msgpu_set_shard the windows cut this shard's reads.
 * The windows are cut by measured work (the index build counts, per read, the scaffold rows it visits as an owner), so they
 * hold the shares wanted whatever the read ids have to do with genome position.
 * n_batches 0 = 8 (3 with MSGPU_BATCH_NO_EDGEMATCHES).  `rows` should be pinned (msgpu_pinned_alloc) for the copy to run at link speed. */
typedef struct msgpu_host_tables {
  const msgpu_edge      *edges;
  const msgpu_edgematch *ems;
  const msgpu_order     *orders;
  const uint32_t        *ids;
  const int32_t         *read_len;        /* Vertex::getNanoporeLength(), n_reads entries */
  const uint32_t        *read_first_line; /* Vertex::metaDatum(0) */
  uint64_t n_edges, n_ems, n_orders, n_ids;
  uint32_t n_reads, n_anchors, n_batches, pad;
  float wall_ms;         /* host clock: call entry -> all tables in host memory */
  float load_ms;         /*   of which rows -> HBM + index build */
  float first_batch_ms;  /*   first window computed (its copy starts here) */
  float compute_done_ms; /*   last window computed (what remains is copy) */
} msgpu_host_tables;
int msgpu_overlap_batched(msgpu_ctx *ctx, const msgpu_row *rows, size_t n_rows, uint32_t n_batches,
                          msgpu_host_tables *out);
/* The same with options (flags = 0: msgpu_overlap_batched).
 *   MSGPU_BATCH_RESIDENT        the job's four tables stay WHOLE in HBM (window k writes behind window k-1; 288 GB of HBM
 *                               hold the 1 GB of BASELINE.json configs[2] many times over): no second table set, the
 *                               compute stream never waits for a copy, and afterwards the context is in the state
 *                               msgpu_chaining_and_overlaps leaves -- msgpu_find_contraction_edges, msgpu_copy_tables
 *                               [_device], msgpu_get_edgematches and msgpu_get_counts work on the job's tables.
 *   MSGPU_BATCH_NO_EDGEMATCHES  (implies RESIDENT) the EdgeMatch table is NOT copied to the host: out->ems = NULL,
 *                               out->n_ems is still its size.  Downstream only assemblePath reads EdgeMatches, and only
 *                               those of path edges (dg.cpp:99-101 -> ap.cpp:631-706): fetch them with
 *                               msgpu_get_edgematches.  The other three tables cross the host link in the exchange's
 *                               wire form (msgpu_pack_wire per window; 93 MB instead of 154 MB -- or 971 MB with the
 *                               EdgeMatches -- on configs[2]) and a host thread of the call turns every window back
 *                               into records (msgpu_unpack_wire_host) while the next one computes: the tables handed
 *                               out are the same records.  MSGPU_NO_WIRE_COPY=1 in the environment of msgpu_create:
 *                               whole records over the link (A/B switch).
 *   MSGPU_BATCH_ROWS_ON_DEVICE  `rows` is a DEVICE pointer (msgpu_load_rows_device): the table is in HBM already, e.g.
 *                               all-gathered over xGMI from the 1/N slices the ranks of a node uploaded over their own links.
 *   MSGPU_BATCH_ROWS_PACKED     `rows` points to a msgpu_packed_rows (below) and n_rows is its n_rows: 28 bytes per row over
 *                               the host link instead of 40 (msgpu_load_rows_packed).
 */
#define MSGPU_BATCH_RESIDENT 1u
#define MSGPU_BATCH_NO_EDGEMATCHES 2u
#define MSGPU_BATCH_ROWS_ON_DEVICE 4u
#define MSGPU_BATCH_ROWS_PACKED 8u
int msgpu_overlap_batched_ex(msgpu_ctx *ctx, const msgpu_row *rows, size_t n_rows, uint32_t n_batches, uint32_t flags,
                             msgpu_host_tables *out);
/* MatchMap::getEdgeMatches(edge) (libms/src/matching/MatchMap.cpp:136-159) for a LIST of edges, from the EdgeMatch table
 * resident in HBM (after msgpu_chaining_and_overlaps or a resident msgpu_overlap_batched_ex): one gather kernel + one
 * copy.  edge_idx[i] = index in the edge table.  *em_off (n + 1 entries) / *ems: the EdgeMatches of edge_idx[i] are
 * (*ems)[(*em_off)[i] .. (*em_off)[i+1]), in table order (edge_idx of each record is unchanged).  Both arrays are pinned
 * host memory owned by the context, valid until its next msgpu_get_edgematches / msgpu_destroy. */
int msgpu_get_edgematches(msgpu_ctx *ctx, const uint32_t *edge_idx, size_t n, const uint64_t **em_off,
                          const msgpu_edgematch **ems);
/* page-locked host memory for rows handed to msgpu_load_rows / msgpu_overlap_batched (NULL when out of memory) */
void *msgpu_pinned_alloc(size_t bytes);
void  msgpu_pinned_free(void *p);

/* findContractionEdges (src/main.cpp:183-190, 416-463) with sanityCheck (libms/src/kernel/sc.cpp:29-90) -- the step
 * that follows the chaining fan-out -- on an edge/order table resident in HBM.  contraction_order (host, n_edges
 * entries): for every edge the index in the order table of its first contained & primary EdgeOrder that is sane against
 * every non-shadow neighbour of the order's start vertex (what the reference inserts into `contractionEdges`), or -1.
 * d_edges = d_orders = NULL: the context's own tables (single GPU, after msgpu_chaining_and_overlaps); otherwise any
 * (v1, v2)-sorted edge table + its order table, e.g. the merged list of msgpu_merge_gathered; n_reads = max id + 1.
 * Uses msgpu_params.wiggle_room.  Synchronous for the host; caller tables are read in the context's stream order
 * (STREAM CONTRACT rule 3 (a): they must be complete there). */
int msgpu_find_contraction_edges(msgpu_ctx *ctx, const void *d_edges, uint64_t n_edges, const void *d_orders,
                                 uint64_t n_orders, uint32_t n_reads, int64_t *contraction_order);

/* Block the host until everything queued on the context's stream has finished. */
int msgpu_synchronize(msgpu_ctx *ctx);
/* A deadline for the host waits of this context: from now on, and until the next msgpu_set_deadline, every wait a call of this
 * context makes for its stream (table sizes coming back in msgpu_load_rows* / msgpu_calculate_edges /
 * msgpu_chaining_and_overlaps, msgpu_synchronize, msgpu_copy_reads) gives up `timeout_ms` after THIS call and returns
 * MSGPU_E_TIMEOUT; 0 = no deadline (the default: the runtime's blocking waits).  Nothing is cancelled: the work stays queued,
 * the caller removes what holds the stream up (e.g. aborts its collective) and synchronises, or destroys the context.  Exists
 * for contexts whose stream also carries somebody else's collective (msgpu_group_overlap sets it from the group's timeout). */
int msgpu_set_deadline(msgpu_ctx *ctx, uint32_t timeout_ms);
/* A gate for a caller's OTHER thread that has device work of its own to place (an exchange of the previous job's tables, say):
 * msgpu_chain_launches = how often msgpu_chaining_and_overlaps has launched its chain kernels so far; msgpu_wait_chain_launch
 * blocks the calling thread until that count reaches `count` (0) or `timeout_us` has passed (1).  The chain stage is bound by
 * instruction issue, the stages before it by memory: work enqueued on another stream when the wait returns runs beside the
 * stage that has bandwidth to spare (bench.py --gpus N: the merge of step k's gathered slabs beside the chain stage of step
 * k + 1; the all-gather itself, bound by the links, is not held).  The two calls may be made from any thread while another thread drives the context (the one exception to rule 5 of
 * the STREAM AND THREAD CONTRACT besides msgpu_merge_*_ex); they touch nothing but the counter. */
uint64_t msgpu_chain_launches(msgpu_ctx *ctx);
int msgpu_wait_chain_launch(msgpu_ctx *ctx, uint64_t count, uint32_t timeout_us);

/* ==== sequence store + slice / reverse-complement / stitch kernel: device half of the "consensus" stage (A9) ========
 *
 * assemblePath (libms/src/kernel/ap.cpp:615-1362) decides where every piece of sequence goes; the bytes come from
 *   SequenceAccessor::buildIndex / get{Nanopore,Illumina}Sequence(id)     SequenceAccessor.cpp:54-69,114-231
 *   strSlice, getReverseComplement, get{Illumina,Nanopore}Sequence(l,r,d)  SequenceUtils.cpp:27-85
 *   updateConsensusBase (prepend / append the uncovered part)              ap.cpp:205-229
 * Here: the files are parsed once on the host (msgpu_seq_parse), every record lives whitespace-free in HBM
 * (msgpu_seq_upload), and a whole batch of pieces is produced by ONE kernel launch (msgpu_gather_run). */

typedef struct msgpu_seqfile msgpu_seqfile;     /* a parsed FASTA/FASTQ file on the host              */
typedef struct msgpu_seqctx msgpu_seqctx;       /* the two sequence stores (nanopore, illumina) in HBM */
typedef struct msgpu_gather_plan msgpu_gather_plan;

/* Replaces SequenceAccessor::_buildNanoporeIdx/_buildIlluminaIdx + getSequenceFromFile.  is_fastq: 1, 0, or -1 to
 * decide from the extension like isFastQ (SequenceAccessor.cpp:71-80: FASTQ unless ".fa"/".fasta"). */
int         msgpu_seq_parse(const char *path, int is_fastq, msgpu_seqfile **out);
void        msgpu_seq_free(msgpu_seqfile *f);
uint32_t    msgpu_seq_count(const msgpu_seqfile *f);
const char *msgpu_seq_name(const msgpu_seqfile *f, uint32_t record);   /* cleaned id (cut at the first whitespace) */
uint64_t    msgpu_seq_length(const msgpu_seqfile *f, uint32_t record);
const char *msgpu_seq_bases(const msgpu_seqfile *f, uint32_t record);  /* not NUL-terminated                       */
/* the one buffer every record's bytes lie in (msgpu_seq_bases points into it; records need not touch each other) and
 * its used size: what msgpu_seq_upload copies to HBM */
const char *msgpu_seq_buffer(const msgpu_seqfile *f, uint64_t *bytes);
uint64_t    msgpu_seq_offset(const msgpu_seqfile *f, uint32_t record);  /* of the record's bytes inside that buffer */

/* strSlice (SequenceUtils.cpp:27-38) as (returned offset, *len): Python-like indices, INCLUSIVE clipped end. */
uint64_t msgpu_str_slice(uint64_t size, int32_t start, int32_t end, uint64_t *len);

/* device = HIP ordinal (MSGPU_E_NODEVICE without a GPU), or -1 for a layout-only context: the slice arithmetic and the
 * segment composers below work on it, but nothing can be uploaded to or gathered on a device. */
int         msgpu_seq_create(int device, msgpu_seqctx **out);
void        msgpu_seq_destroy(msgpu_seqctx *ctx);
const char *msgpu_seq_last_error(const msgpu_seqctx *ctx);
/* kind 0 = nanopore reads, 1 = illumina unitigs.  ids[record] = Registry id of that record (0xffffffff = skip);
 * NULL = record order.  n_ids = size of the id space. */
int msgpu_seq_upload(msgpu_seqctx *ctx, int kind, const msgpu_seqfile *f, const uint32_t *ids, uint32_t n_ids);
/* The same in two steps, for a caller that parses the sequence files beside the PAF (SequenceAccessor::buildIndex needs
 * the Registry only for the ids, SequenceAccessor.cpp:171,215): the bytes as soon as a file is parsed -- the two kinds may
 * be sent (and converted, msgpu_seq_pack_store) from two host threads at the same time --, the ids of the SAME file once
 * the registries exist (MSGPU_E_STATE when the store holds another file's bytes). */
int msgpu_seq_upload_bases(msgpu_seqctx *ctx, int kind, const msgpu_seqfile *f);
/* msgpu_seq_parse + msgpu_seq_upload_bases in one pass over the file: the parser's threads strip the records into a ring
 * of page-locked slots that travel to the store while the file is still being read; the host never holds the bases.  The
 * msgpu_seqfile that comes back has names, lengths and offsets (msgpu_seq_set_ids and msgpu_paf_register_sequences take
 * it) but no bytes: msgpu_seq_bases and msgpu_seq_buffer return NULL for it. */
int msgpu_seq_parse_upload(msgpu_seqctx *ctx, int kind, const char *path, int is_fastq, msgpu_seqfile **out);
int msgpu_seq_set_ids(msgpu_seqctx *ctx, int kind, const msgpu_seqfile *f, const uint32_t *ids, uint32_t n_ids);

/* Same from a device buffer that already holds the whitespace-free bases (copied device-to-device into the store):
 * off[id] / len[id] = position of sequence `id` inside it (off = ~0 for an id without sequence). */
int msgpu_seq_upload_device(msgpu_seqctx *ctx, int kind, const void *d_bases, uint64_t n_bases, const uint64_t *off,
                            const uint64_t *len, uint32_t n_ids);

/* Convert both resident stores to 2 bits per base (A C G T) plus a sorted list of the positions holding any other byte
 * (N, lower case, IUPAC: reproduced verbatim), and free the byte-per-base copies: a quarter of the HBM footprint and
 * 1.25 instead of 2 bytes of traffic per gathered base.  Results of every later gather are unchanged.  A new upload
 * returns the store to the byte form. */
int msgpu_seq_pack(msgpu_seqctx *ctx);
int msgpu_seq_pack_store(msgpu_seqctx *ctx, int kind); /* one store only */

/* One piece of output: `len` bases starting at `src_off` of a store, as they are or reverse-complemented, written
 * at dst_off.  24 bytes. */
typedef struct msgpu_copy {
  uint64_t src_off; /* first source base, offset inside the store                     */
  uint64_t dst_off; /* first output byte                                               */
  uint32_t len;
  uint32_t flags;   /* MSGPU_COPY_*                                                    */
} msgpu_copy;
#define MSGPU_COPY_ILLUMINA 1u /* source store: illumina (else nanopore)                */
#define MSGPU_COPY_REVCOMP 2u  /* reverse complement (direction == false)               */

/* get{Nanopore,Illumina}Sequence(seq_id, left, right, direction) (SequenceUtils.cpp:63-85) as a piece: fills src_off,
 * len and flags of *out (dst_off is the caller's layout decision). */
int msgpu_seq_resolve(msgpu_seqctx *ctx, int kind, uint32_t seq_id, int32_t left, int32_t right, int direction,
                      msgpu_copy *out);

/* The segment builders of assemblePath as piece composers (libms/src/kernel/ap.cpp:191-203, 352-579).  m / ml / mr =
 * the read's VertexMatch rows on the anchor(s) (read_id, anchor_id, ranges, direction bit); ov* = the anchor's
 * overlap from Id2OverlapMap; direction = the read's orientation in the layout (Vertex::getVertexDirection() == e_POS).
 * `out` receives the pieces in output order with dst_off relative to the start of the segment (up to 1 / 2 / 2 / 3
 * pieces); add the segment's position in the output to every dst_off before planning the gather. */
int msgpu_seg_anchor(msgpu_seqctx *ctx, const msgpu_row *m, int32_t ov_lo, int32_t ov_hi, int direction,
                     msgpu_copy *out, uint32_t *n_out, uint64_t *len);                       /* getAnchorSequence        */
int msgpu_seg_left_of_anchor(msgpu_seqctx *ctx, const msgpu_row *m, uint64_t nanopore_length, int32_t ov_lo,
                             int32_t ov_hi, int direction, msgpu_copy *out, uint32_t *n_out,
                             uint64_t *len);                                                  /* getSequenceLeftOfAnchor  */
int msgpu_seg_right_of_anchor(msgpu_seqctx *ctx, const msgpu_row *m, uint64_t nanopore_length, int32_t ov_lo,
                              int32_t ov_hi, int direction, msgpu_copy *out, uint32_t *n_out,
                              uint64_t *len);                                                 /* getSequenceRightOfAnchor */
/* getSequenceBetweenAnchors: *has_sequence = 0 mirrors std::nullopt; *distance = std::get<0> of its result. */
int msgpu_seg_between_anchors(msgpu_seqctx *ctx, const msgpu_row *ml, const msgpu_row *mr, int32_t ovl_lo,
                              int32_t ovl_hi, int32_t ovr_lo, int32_t ovr_hi, int direction, msgpu_copy *out,
                              uint32_t *n_out, int32_t *distance, int *has_sequence);

/* updateConsensusBase (ap.cpp:205-229) on piece lists: the growing contig of visitOrdered as (pieces, borderLeft,
 * borderRight).  An update takes the new sequence as a segment (pieces with segment-relative dst_off, e.g. what the
 * msgpu_seg_* composers return) plus its borders and prepends / appends the part the contig does not cover yet. */
typedef struct msgpu_consensus msgpu_consensus;
msgpu_consensus *msgpu_consensus_new(void);
void             msgpu_consensus_free(msgpu_consensus *c);
int msgpu_consensus_update(msgpu_consensus *c, const msgpu_copy *seg, uint32_t n, int32_t new_lo, int32_t new_hi);
int msgpu_consensus_borders(const msgpu_consensus *c, int32_t *lo, int32_t *hi, uint64_t *length);
/* The contig as pieces laid out from dst_off = base; returns the piece count (call with out = NULL to size). */
size_t msgpu_consensus_pieces(const msgpu_consensus *c, uint64_t base, msgpu_copy *out, size_t cap);
msgpu_consensus *msgpu_consensus_clone(const msgpu_consensus *c);

/* Upload a batch of pieces (+ its work partition) once; run it any number of times. */
int      msgpu_gather_plan_create(msgpu_seqctx *ctx, const msgpu_copy *pieces, size_t n, msgpu_gather_plan **out);
void     msgpu_gather_plan_free(msgpu_gather_plan *plan);
uint64_t msgpu_gather_plan_out_bytes(const msgpu_gather_plan *plan); /* max(dst_off + len)  */
uint64_t msgpu_gather_plan_bases(const msgpu_gather_plan *plan);     /* sum(len)            */
/* d_out: device buffer of out_capacity >= out_bytes.  hip_stream NULL = the context's stream. Asynchronous. */
int msgpu_gather_run(msgpu_seqctx *ctx, const msgpu_gather_plan *plan, void *d_out, uint64_t out_capacity,
                     void *hip_stream);
int msgpu_seq_synchronize(msgpu_seqctx *ctx);

/* ---- assemblePath (libms/src/kernel/ap.cpp:615-1362; caller assemblePathsSub, src/main.cpp:663-677) ---------------
 * One msgpu_assembly collects any number of paths.  msgpu_assembly_add_path runs the reference's layout decisions on
 * the host for one path (candidate EdgeOrders, anchor cliques, the anchor DAG, placement, flanks, contained reads) and
 * records every output sequence as copy pieces; it reads no base.  msgpu_assembly_finish then produces all bases of
 * all paths with ONE gather launch and wraps them into FASTA text on the device (60 columns, ap.cpp:52,61-76):
 * the texts are what OutputWriter::writeTarget / writeQuery / writePaf (OutputWriter.cpp:49-62) receive, in path order.
 *
 * Hash-order note: where ap.cpp iterates std::unordered_map / unordered_set (graph vertices, edges, successors, tap
 * entries) the reference's order is unspecified; this library uses ascending ids / creation order ("canonical order"). */
typedef struct msgpu_path_read {
  uint32_t read_id;          /* Vertex::getId()                                         */
  uint32_t direction;        /* Vertex::getVertexDirection(): 1 = e_POS, 0 = e_NEG, 2 = e_NONE */
  uint64_t nanopore_length;  /* Vertex::getNanoporeLength()                             */
} msgpu_path_read;
typedef struct msgpu_path_order { /* an EdgeOrder (include/ms/graph/Edge.h:49-60) of a directed path edge */
  uint64_t score;
  uint32_t base_read; /* baseVertex id                                                 */
  uint32_t ids_off;   /* its ids = ids[ids_off .. ids_off + ids_cnt)                   */
  uint32_t ids_cnt;
  uint32_t pad;
} msgpu_path_order;
typedef struct msgpu_path_em { /* EdgeMatch::overlap of a path edge on one anchor, MatchMap.h:68-74 */
  uint32_t anchor_id;
  int32_t  ov_lo, ov_hi;
} msgpu_path_em;
typedef struct msgpu_path_contain { /* a ContainElement (MatchMap.h:80-87) attached to a read of the path */
  uint32_t host_read;   /* the path read that contains it                              */
  uint32_t nano;        /* the contained read                                          */
  uint32_t direction;   /* ContainElement::direction                                   */
  uint32_t anchors_off; /* keys of ContainElement::matches = contain_anchors[off .. off + cnt); the VertexMatch of */
  uint32_t anchors_cnt; /* (nano, anchor) is looked up in `rows`                       */
} msgpu_path_contain;
typedef struct msgpu_path_input {
  const msgpu_path_read  *reads;     /* the path, n_reads >= 2                                              */
  uint32_t                n_reads;
  int32_t                 asm_idx;   /* asmIdx: names ">muchsalsa_<asmIdx>", ">Middle.<asmIdx>.<n>" ...     */
  const uint32_t         *order_off; /* n_reads entries: path edge i = reads[i] -> reads[i+1] owns orders   */
  const msgpu_path_order *orders;    /*   [order_off[i], order_off[i+1]) (diGraph.getEdge(..)->getEdgeOrders()) */
  const uint32_t         *ids;       /* id pool of the orders                                               */
  const uint32_t         *em_off;    /* n_reads entries: EdgeMatches of path edge i                         */
  const msgpu_path_em    *ems;
  const msgpu_row        *rows;      /* MatchMap::getVertexMatch source: the (read, anchor) rows of the path's reads */
  size_t                  n_rows;    /*   and of the contained reads (any order); optional after msgpu_assembly_set_rows */
  const msgpu_path_contain *contains;
  uint32_t                n_contains;
  uint32_t                pad;
  const uint32_t         *contain_anchors;
} msgpu_path_input;

typedef struct msgpu_assembly msgpu_assembly;
typedef struct msgpu_path_info {
  uint64_t target_len;     /* bases of the contig                                        */
  uint64_t target_raw_off; /* where its bases start in the raw (unwrapped) buffer        */
  uint32_t query_begin, query_end; /* its query records                                  */
  uint32_t n_anchors, n_anchor_edges;
  int32_t  border_lo, border_hi;   /* globalPos1, globalPos2 (ap.cpp:880-881)            */
  int32_t  asm_idx;
  uint32_t pad;
} msgpu_path_info;
#define MSGPU_QUERY_MIDDLE 0u
#define MSGPU_QUERY_LEFT 1u
#define MSGPU_QUERY_RIGHT 2u
#define MSGPU_QUERY_CONTAIN_ILLUMINA 3u
#define MSGPU_QUERY_CONTAIN_NANO 4u
typedef struct msgpu_query_info {
  uint64_t len;
  uint64_t raw_off;
  int64_t  lb, rb; /* PAF columns 8 and 9 as the reference prints them                   */
  uint32_t kind;   /* MSGPU_QUERY_*                                                      */
  uint32_t path;
} msgpu_query_info;

/* The assembly uses `ctx` until msgpu_assembly_finish / _validate have returned; freeing it never touches `ctx`. */
int         msgpu_assembly_create(msgpu_seqctx *ctx, msgpu_assembly **out);
void        msgpu_assembly_free(msgpu_assembly *a);
const char *msgpu_assembly_last_error(const msgpu_assembly *a);
/* Install the VertexMatch table once (MatchMap::getVertexMatch for every later path; copied, n_rows < 2^32).  A path's
 * own msgpu_path_input.rows, when given, are looked up first. */
int msgpu_assembly_set_rows(msgpu_assembly *a, const msgpu_row *rows, size_t n_rows);
/* The same without the copy: `rows` (e.g. msgpu_paf_rows of a live msgpu_paf) stays the caller's and must outlive the
 * assembly's last msgpu_assembly_add_path(s).  The reference's MatchMap hands out pointers into its own table the same way. */
int msgpu_assembly_borrow_rows(msgpu_assembly *a, const msgpu_row *rows, size_t n_rows);
/* MSGPU_E_LAYOUT leaves the assembly unchanged (the path is skipped). */
int      msgpu_assembly_add_path(msgpu_assembly *a, const msgpu_path_input *in);
/* The assemblePaths fan-out (src/main.cpp:620-677): n paths laid out by n_threads host threads, appended in input
 * order.  status (optional, n entries) receives each path's result; MSGPU_E_LAYOUT paths are skipped.  Returns the
 * first status that is neither MSGPU_OK nor MSGPU_E_LAYOUT, else MSGPU_OK. */
int msgpu_assembly_add_paths(msgpu_assembly *a, const msgpu_path_input *in, size_t n, uint32_t n_threads, int *status);
uint32_t msgpu_assembly_path_count(const msgpu_assembly *a);
uint32_t msgpu_assembly_query_count(const msgpu_assembly *a);
int      msgpu_assembly_path_info(const msgpu_assembly *a, uint32_t path, msgpu_path_info *out);
int      msgpu_assembly_query_info(const msgpu_assembly *a, uint32_t query, msgpu_query_info *out);
/* every copy piece of every record, dst_off = position in the raw buffer; returns the count (out = NULL to size) */
size_t   msgpu_assembly_pieces(const msgpu_assembly *a, msgpu_copy *out, size_t cap);
uint64_t msgpu_assembly_raw_bytes(const msgpu_assembly *a);
/* gather + FASTA wrapping on the device, texts copied to host memory owned by the assembly.  hip_stream NULL = the
 * context's stream.  Synchronous.  MSGPU_E_NODEVICE on a layout-only context. */
int msgpu_assembly_finish(msgpu_assembly *a, void *hip_stream);
/* After finish: banded Levenshtein distance (msgpu_edit_distance semantics, band <= 127) of every query record against
 * the stretch of its contig that its PAF line names, clipped to the contig; distance[i] for query i (host, query_count
 * entries): the distance when <= band, else band + 1.  dp_cells (optional): DP cells inside the band that were evaluated.
 * No reference counterpart: it is the meter of how well the query records agree with the contig. */
int msgpu_assembly_validate(msgpu_assembly *a, uint32_t band, uint32_t *distance, uint64_t *dp_cells);
/* which: 0 = temp_1.target.fa, 1 = temp_1.query.fa (both after finish), 2 = temp_1.align.paf (after add_path) */
const char *msgpu_assembly_text(const msgpu_assembly *a, int which, uint64_t *len);

/* FASTA wrapping on the device: record r = header bytes, then `len` bases from d_raw + raw_off in lines of 60
 * (limitLength, ap.cpp:61-76), then '\n'; written at d_text + text_off.  Asynchronous on hip_stream / the ctx stream. */
typedef struct msgpu_fasta_record {
  uint64_t raw_off, text_off;
  uint32_t len;
  uint32_t header_off, header_len; /* header = headers[header_off .. +header_len), e.g. ">muchsalsa_1\n" */
  uint32_t pad;
} msgpu_fasta_record;
uint64_t msgpu_fasta_text_bytes(uint32_t header_len, uint64_t len);
int msgpu_fasta_format(msgpu_seqctx *ctx, const void *d_raw, const msgpu_fasta_record *records, size_t n,
                       const char *headers, size_t headers_bytes, void *d_text, uint64_t text_capacity,
                       void *hip_stream);

/* ---- between the overlap path and assemblePath (host) -----------------------------
 * graph clean-up (src/main.cpp:194-288, 465-618: contraction targets and roots, ContainElements, deletions,
 * computeBitweight, getMaxSpanTree mst.cpp:34-111, decycle), getConnectedComponents (cc.cpp:33-70) and, per component,
 * getDirectedGraph (dg.cpp:35-121) + linearizeGraph (lg.cpp:41-629) as the assemblePaths job does (main.cpp:620-661).
 * Input: the tables of msgpu_copy_tables / msgpu_copy_reads (copied) and the result of msgpu_find_contraction_edges.
 * Output: one msgpu_path_input per linearised path, ready for msgpu_assembly_add_paths.  Iteration orders the
 * reference leaves to hash containers follow DESIGN.md section 2, "canonical order".  MSGPU_E_LAYOUT = the reference would terminate. */
typedef struct msgpu_graph msgpu_graph;
typedef struct msgpu_graph_stats {
  uint64_t n_vertices_in, n_edges_in;
  uint64_t n_contraction_edges, n_deleted_vertices, n_contain_elements, n_decycled_edges;
  uint64_t n_vertices, n_edges; /* after the clean-up */
  uint64_t n_components, n_paths, n_path_reads;
} msgpu_graph_stats;
/* The four tables are COPIED (read_len / read_first_line always are).  ems = NULL (n_ems ignored): no EdgeMatch table on
 * the host -- the graph stage itself never reads one; the EdgeMatches of the path edges, which assemblePath needs
 * (dg.cpp:99-101), are supplied after msgpu_graph_linearize through msgpu_graph_path_edges + msgpu_get_edgematches +
 * msgpu_graph_set_path_edgematches. */
int  msgpu_graph_create(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                        const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                        const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads, msgpu_graph **out);
/* The same WITHOUT the copy: the four tables are borrowed (e.g. the pinned result of msgpu_overlap_batched; the EdgeMatch
 * table alone is 0.8 GB on BASELINE.json configs[2]) and must stay valid and unchanged until msgpu_graph_free. */
int  msgpu_graph_create_borrowed(const msgpu_edge *edges, uint64_t n_edges, const msgpu_edgematch *ems, uint64_t n_ems,
                                 const msgpu_order *orders, uint64_t n_orders, const uint32_t *ids, uint64_t n_ids,
                                 const int32_t *read_len, const uint32_t *read_first_line, uint32_t n_reads,
                                 msgpu_graph **out);
void msgpu_graph_free(msgpu_graph *g);
const char *msgpu_graph_last_error(const msgpu_graph *g);
/* rows (optional): the VertexMatch table, for contract()'s "getVertexMatch(start, id) != nullptr" (main.cpp:514-519);
 * NULL = every id of an order has one (true for tables produced by this library). */
int msgpu_graph_clean_up(msgpu_graph *g, const int64_t *contraction_order, const msgpu_row *rows, size_t n_rows);
/* host threads for msgpu_graph_linearize (components are independent, cf. one assemblePaths job per component,
 * src/main.cpp:300-310); default 1; the result does not depend on it */
int msgpu_graph_set_threads(msgpu_graph *g, uint32_t n_threads);
int msgpu_graph_linearize(msgpu_graph *g);
int msgpu_graph_get_stats(const msgpu_graph *g, msgpu_graph_stats *out);
uint32_t msgpu_graph_path_count(const msgpu_graph *g);
/* After msgpu_graph_linearize: the edge-table index of the edge under every path step, paths concatenated in path order
 * (path i contributes n_reads_i - 1 entries); owned by the graph.  With a graph created without an EdgeMatch table, hand
 * this list to msgpu_get_edgematches and its result to msgpu_graph_set_path_edgematches (copied) before
 * msgpu_graph_path_input, which fails with MSGPU_E_STATE until then. */
int msgpu_graph_path_edges(const msgpu_graph *g, const uint32_t **edge_idx, size_t *n);
int msgpu_graph_set_path_edgematches(msgpu_graph *g, const uint64_t *em_off, const msgpu_edgematch *ems);
/* path i (asm_idx = i) as assemblePath input; pointers are owned by the graph and valid until msgpu_graph_free;
 * rows / n_rows are left empty (use msgpu_assembly_set_rows). */
int msgpu_graph_path_input(const msgpu_graph *g, uint32_t i, msgpu_path_input *out);
/* msgpu_graph_path_input for every path + msgpu_assembly_add_paths in one call (status: msgpu_graph_path_count(g) entries
 * or NULL): the assemblePaths fan-out of src/main.cpp:620-677 over the paths of this graph.  The graph must outlive the
 * call (the layouts read its tables), not the assembly. */
int msgpu_assembly_add_graph_paths(msgpu_assembly *a, const msgpu_graph *g, uint32_t n_threads, int *status);
/* inspection (all optional): per vertex alive flag and direction (1 e_POS / 0 e_NEG / 2 e_NONE); per edge (table
 * order) alive flag, consensus direction (same coding) and weight */
int msgpu_graph_state(const msgpu_graph *g, uint8_t *vertex_alive, uint8_t *vertex_direction, uint8_t *edge_alive,
                      uint8_t *edge_consensus, uint64_t *edge_weight);

/* The graph primitives of the stage on caller-supplied graphs (vertices 0 .. n_vertices-1, edge i = (a[i], b[i])): the
 * same code msgpu_graph_clean_up / msgpu_graph_linearize run, exposed so that the vectors the reference's own unit tests
 * hold for them can be replayed through this boundary (tests/golden/ref_tests/).  consensus: 1 e_POS, 0 e_NEG, 2 e_NONE.
 *   getMaxSpanTree          libms/src/kernel/mst.cpp:75-111   (libms/tests/MST_test.cpp:8-60)
 *   getConnectedComponents  libms/src/kernel/cc.cpp:33-70     (libms/tests/CC_test.cpp:11-90)
 *   GraphUtil::getShortestPath  include/ms/graph/Graph.h:927-978  (libms/tests/Graph_test.cpp:279-331); directed != 0:
 *                           a DiGraph (edges a -> b, neighbours = successors)
 *   DiGraph::sortTopologically  libms/src/graph/Graph.cpp:359-395 (libms/tests/Graph_test.cpp:393-423) */
int msgpu_graph_max_span_tree(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, const uint64_t *weight,
                              const uint8_t *consensus, uint64_t n_edges, uint8_t *in_tree /* n_edges */);
/* Graph / DiGraph bookkeeping on its own -- what the clean-up's deletions (src/main.cpp:243,259,286) and the component split
 * (src/main.cpp:625 getSubgraph) rest on; the reference's tests hold vectors for it (libms/tests/Graph_test.cpp:81-277,
 * 333-391: EdgeDeletion, VertexDeletion, Neighboor, Subgraph, Degree).  A flat graph of n_vertices and the pairs (a[i], b[i])
 * -- undirected, or a -> b when directed != 0; a pair given twice is ONE edge, as Graph::addEdge has it (Graph.cpp:212-230) --
 * is built by the stage's CSR builder; `ops` is run over it in order, with tombstones as the stage keeps them; every query
 * appends to `out`:
 *   MSGPU_GOP_DELETE_EDGE x y    Graph::deleteEdge (Graph.cpp:187-210); nothing when there is no such edge
 *   MSGPU_GOP_DELETE_VERTEX x    Graph::deleteVertex (:158-185): the vertex and every edge at it
 *   MSGPU_GOP_ORDER / _SIZE      getOrder() / getSize()                                   -> 1 word
 *   MSGPU_GOP_HAS_EDGE x y       hasEdge (undirected: either way round)                   -> 0 | 1
 *   MSGPU_GOP_NEIGHBORS x        getNeighbors (undirected) / getSuccessors (directed)     -> count, then the ids ascending
 *   MSGPU_GOP_PREDECESSORS x     getPredecessors (directed only)                          -> count, ids
 *   MSGPU_GOP_IN_DEGREE x / _OUT_DEGREE x   getInDegrees().at(x) / getOutDegrees().at(x)  -> 1 word (0xffffffff: x was deleted)
 *   MSGPU_GOP_SUBGRAPH x y       getSubgraph of the y vertices listed in ops[x .. x + y) (entries MSGPU_GOP_ARG, .x = vertex)
 *                                                                                         -> order, size, then (a, b) per edge
 * *n_out = words the script produces; MSGPU_E_ARG when they do not fit out_capacity (call again with room) or an operand is
 * out of range. */
typedef struct msgpu_graph_op {
  uint32_t op, x, y;
} msgpu_graph_op;
#define MSGPU_GOP_DELETE_EDGE 1u
#define MSGPU_GOP_DELETE_VERTEX 2u
#define MSGPU_GOP_ORDER 3u
#define MSGPU_GOP_SIZE 4u
#define MSGPU_GOP_HAS_EDGE 5u
#define MSGPU_GOP_NEIGHBORS 6u
#define MSGPU_GOP_PREDECESSORS 7u
#define MSGPU_GOP_IN_DEGREE 8u
#define MSGPU_GOP_OUT_DEGREE 9u
#define MSGPU_GOP_SUBGRAPH 10u
#define MSGPU_GOP_ARG 11u
int msgpu_graph_bookkeeping(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges, int directed,
                            const msgpu_graph_op *ops, size_t n_ops, uint32_t *out, size_t out_capacity, size_t *n_out);
int msgpu_graph_connected_components(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, const uint8_t *consensus,
                                     uint64_t n_edges, uint32_t *component /* n_vertices */, uint32_t *n_components);
/* *n_path: in = capacity of path, out = vertices on the path (0 = unreachable); MSGPU_E_ARG when it does not fit */
int msgpu_graph_shortest_path(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges, int directed,
                              uint32_t src, uint32_t dst, uint32_t *path, uint32_t *n_path);
/* order: n_vertices entries of room; *n_order < n_vertices when the graph has a cycle (its vertices never appear) */
int msgpu_graph_sort_topologically(uint32_t n_vertices, const uint32_t *a, const uint32_t *b, uint64_t n_edges,
                                   uint32_t *order, uint32_t *n_order);

/* ---- banded edit distance (no reference counterpart) ---------------------------------
 * The meter for north_star's "consensus sequences within a stated edit-distance tolerance": Levenshtein distance
 * (unit costs, global) of n pairs a[a_off .. a_off+a_len) vs b[b_off .. b_off+b_len) taken from two DEVICE buffers,
 * inside the band |j - i| <= band (band <= 127).  out[p] (host) = the distance when it is <= band, else band + 1
 * (= min(distance, band + 1): a path of d <= band edits never leaves the band).  Sequences shorter than 2^30 bytes.
 * Computed by furthest-reaching points per (edits, diagonal); MSGPU_ED_DP=1 in the environment selects the banded
 * anti-diagonal DP kernel instead (same numbers, for cross-checks). */
typedef struct msgpu_align_pair {
  uint64_t a_off, b_off;
  uint32_t a_len, b_len;
} msgpu_align_pair;
int msgpu_edit_distance(msgpu_seqctx *ctx, const void *d_a, const void *d_b, const msgpu_align_pair *pairs, size_t n,
                        uint32_t band, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* MSGPU_H */
