// msgpu_api.hip -- C-ABI host layer of libmsgpu: context, HBM arena, stage orchestration on one HIP stream.
//
// This is the replacement of the reference's ThreadPool/WaitGroup fan-out (libms/src/threading/*): every reference
// phase "one Job per line/anchor/edge + WaitGroup::wait()" becomes a handful of kernel launches on the context's
// stream; the phase barrier is stream order.  The host blocks only where a table size is needed to allocate.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <algorithm>
#include <sys/mman.h>
#include <unistd.h>

#include "asm_internal.h"
#include "host_pool.h"
#include "msgpu.h"
#include "msgpu_internal.h"

using namespace msgpu;

// ---- page-locked host memory ---------------------------------------------------------------------------------------------
// hipHostMalloc spends its time touching the block's pages one by one on the calling thread (30 ms per 200 MB, and 17 ms
// to give them back): tables that are written once and read once never earn that back.  Here a block is an anonymous
// mapping on 2 MiB pages where the kernel grants them, its pages are touched by several threads (or by the threads that
// fill it), and hipHostRegister locks what is there (0.6 ms per 200 MB on 2 MiB pages, 5 ms on 4 KiB pages); copies run at
// the same 54 GB/s (tools/experiments/pin_timing.cpp).
namespace msgpu {
namespace {
constexpr size_t   PB_HEADER = 64;                // keeps the 64-byte alignment of the payload
constexpr size_t   PB_HUGE   = size_t(2) << 20;
constexpr uint64_t PB_MAGIC  = 0x6d7367707550494eull;
struct BlockHeader {
  uint64_t magic;
  void    *raw;        // what mmap returned
  size_t   raw_len;
  size_t   span;       // header + payload, rounded up to pages: what is (to be) registered
  uint32_t registered; // hipHostRegister succeeded
  uint32_t hip_alloc;  // fallback: the block came from hipHostMalloc
};
static_assert(sizeof(BlockHeader) <= PB_HEADER, "header");
bool have_device() {
  static const bool yes = [] {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
  }();
  return yes;
}
BlockHeader *header_of(void *payload) { return reinterpret_cast<BlockHeader *>(static_cast<char *>(payload) - PB_HEADER); }
void *block_map(size_t bytes) noexcept { // -> payload, untouched and unregistered; nullptr when the mapping fails
  const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
  const size_t span = (PB_HEADER + bytes + page - 1) / page * page;
  const size_t len  = span + PB_HUGE;
  void        *raw  = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == MAP_FAILED) return nullptr;
  char *base = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + PB_HUGE - 1) & ~(uintptr_t(PB_HUGE) - 1));
  madvise(base, span, MADV_HUGEPAGE);
  auto *h = reinterpret_cast<BlockHeader *>(base);
  *h      = BlockHeader{PB_MAGIC, raw, len, span, 0, 0};
  return base + PB_HEADER;
}
void block_touch(void *payload) noexcept { // first touch of every page, on a few threads when the block is large
  BlockHeader *h    = header_of(payload);
  char        *base = reinterpret_cast<char *>(h);
  const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
  auto         touch = [&](size_t b, size_t e) {
    for (size_t o = b; o < e; o += page) static_cast<volatile char *>(base)[o < PB_HEADER ? PB_HEADER : o] = 0;
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt          = nt > 16 ? 16 : (nt ? nt : 1);
  if (h->span < (size_t(8) << 20)) nt = 1;
  const size_t per = ((h->span + nt - 1) / nt + PB_HUGE - 1) / PB_HUGE * PB_HUGE;
  const size_t n_pieces = (h->span + per - 1) / per;
  try {
    msgpu::HostPool::get().run(nt, n_pieces, [&](size_t k) { touch(per * k, std::min(h->span, per * (k + 1))); });
  } catch (...) { // (cannot happen: touching memory does not throw)
    touch(0, h->span);
  }
}
void block_register(void *payload) noexcept {
  BlockHeader *h = header_of(payload);
  if (h->registered || !have_device()) return;
  if (hipHostRegister(h, h->span, hipHostRegisterDefault) == hipSuccess) h->registered = 1;
  else (void)hipGetLastError(); // (the block stays pageable: copies still work, through the runtime's staging)
}
void block_unmap(void *payload) noexcept {
  BlockHeader *h = header_of(payload);
  if (h->hip_alloc) {
    (void)hipHostFree(h);
    return;
  }
  if (h->registered) (void)hipHostUnregister(h);
  munmap(h->raw, h->raw_len);
}
} // namespace

void *pinned_block_alloc(size_t bytes) noexcept {
  void *p = block_map(bytes ? bytes : 1);
  if (p) {
    block_touch(p);
    block_register(p);
    if (header_of(p)->registered || !have_device()) return p;
    block_unmap(p); // (registration refused, e.g. a locked-memory limit: let the runtime allocate)
  }
  void *q = nullptr;
  if (hipHostMalloc(&q, PB_HEADER + (bytes ? bytes : 1), hipHostMallocDefault) != hipSuccess || !q) {
    (void)hipGetLastError();
    return nullptr;
  }
  *static_cast<BlockHeader *>(q) = BlockHeader{PB_MAGIC, q, 0, 0, 0, 1};
  return static_cast<char *>(q) + PB_HEADER;
}
void pinned_block_free(void *p) noexcept {
  if (p) block_unmap(p);
}

// the PAF loader's row table: mapped here, touched by the loader's threads as they write the rows, locked by host_table_pin
void *host_table_alloc(size_t bytes) {
  void *p = block_map(bytes ? bytes : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void host_table_free(void *q) noexcept {
  if (q) block_unmap(q);
}
void host_table_pin(void *q) noexcept {
  if (q) block_register(q);
}
} // namespace msgpu


namespace {

struct DevBuf {
  void  *p   = nullptr; // the allocation
  size_t cap = 0;       // its size in bytes
  // A buffer can be used as a VIEW that starts `off` bytes into the allocation: the job-wide result tables of a
  // resident batched run (msgpu_overlap_batched_ex) -- window k writes behind what the windows before it left, and a
  // reallocation keeps those first `off` bytes.  off = 0 everywhere else.
  size_t off  = 0;
  size_t hint = 0; // expected final size of the allocation (bytes): a reallocation asks for at least this much
  // While a batched job runs its first windows, every (re)allocation on this thread asks for the size the job's LARGEST
  // window will need, not the current one's (msgpu_overlap_batched_ex sets this: windows grow, and a scratch table that
  // grows with them is freed and allocated again -- a device-wide synchronisation -- in every window of a first call)
  static inline thread_local double grow_by = 1.0;
  double own_grow_by = 0.0; // the same for this buffer alone (a job-wide result table while the job's FIRST window fills it)
  hipError_t ensure(size_t bytes) { // room for `bytes` behind `off`
    if (off + bytes <= cap) return hipSuccess;
    const size_t need = off + bytes;
    size_t       want = need + need / 8 + 256; // slack so slowly growing inputs do not reallocate each run
    const double by = own_grow_by > grow_by ? own_grow_by : grow_by;
    if (by > 1.0) want = off + static_cast<size_t>(double(want - off) * by);
    if (hint > want) want = hint;
    static const bool dbg_alloc = std::getenv("MSGPU_ALLOC_DEBUG") != nullptr;
    if (dbg_alloc) fprintf(stderr, "[alloc] %zu -> %zu bytes (need %zu, kept %zu, hint %zu, grow_by %.2f)\n", cap, want, need, off, hint, grow_by);
    void      *np = nullptr;
    hipError_t e;
    if (p && !off) { // nothing to keep: give the old block back first
      e   = hipFree(p);
      p   = nullptr;
      cap = 0;
      if (e != hipSuccess) return e;
    }
    e = hipMalloc(&np, want);
    if (e != hipSuccess) return e;
    // MSGPU_POISON=1 (debugging): fresh device memory is filled with 0xA5 so that a kernel which relies on zeroed scratch
    // fails every time instead of only when the allocator hands back recycled pages
    static const bool poison = std::getenv("MSGPU_POISON") != nullptr;
    if (poison) {
      e = hipMemset(np, 0xA5, want); // null stream, may return before it ran: wait, the context's stream does not
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) {
        (void)hipFree(np);
        return e;
      }
    }
    if (p) { // a view that outgrew its allocation: everything in flight on the old block first, then its kept part moves
      e = hipDeviceSynchronize();
      if (e == hipSuccess) e = hipMemcpy(np, p, off, hipMemcpyDeviceToDevice);
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) {
        (void)hipFree(np);
        return e;
      }
      (void)hipFree(p);
    }
    p   = np;
    cap = want;
    return hipSuccess;
  }
  void release() {
    if (p) (void)hipFree(p);
    p   = nullptr;
    cap = 0;
    off = hint = 0;
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(static_cast<char *>(p) + off); }
  void  *at() const { return static_cast<char *>(p) + off; }
  size_t room() const { return cap > off ? cap - off : 0; } // bytes behind `off`
};

enum State { ST_CREATED = 0, ST_LOADED = 1, ST_EDGES = 2, ST_CHAINED = 3,
             ST_RESULT = 4 /* the whole job's tables, built window by window (msgpu_overlap_batched_ex, resident): no per-edge scratch */ };

} // namespace

struct msgpu_ctx {
  int          device     = 0;
  hipStream_t  stream     = nullptr;
  hipStream_t  own_stream = nullptr;
  msgpu_params p;
  char         err[512]   = {0};
  State        state      = ST_CREATED;
  uint32_t     shard = 0, nshards = 1;
  uint32_t     win_lo = 0, win_hi = 0xffffffffu; // owner-read window of the current batch (msgpu_overlap_batched)
  uint64_t     base_edges = 0, base_ems = 0, base_orders = 0, base_ids = 0; // what precedes it in the job's tables
  uint64_t    *h_scalars = nullptr; // pinned, device-mapped mirror of `scalars` (+ one word: the read-back sequence number)
  uint64_t    *h_scalars_dev = nullptr; // the same memory as the device sees it
  bool         stage_events = true; // the stage boundaries are marked with events (msgpu_set_stage_events)
  bool         chain_zeroed = false; // msgpu_calculate_edges zeroed the chain stage's per-edge counters
  bool         no_prologue = false; // msgpu_overlap_batched with several windows: every window has its own opening
  bool         prologue_ok = false; // the candidate stage's opening ran with the index build (whole table, fast index)
  bool         scalars_clean = false; // the scalar block is zero where a build counts from nothing (k_index_epilogue's publisher left it so)
  bool         bin_clean = false;     // the bin path's bucket cursors are zero (k_index_sort_bin leaves them so)
  size_t       bin_zero_words = 0;    // ... as far as an init launch has ever zeroed them (a larger job needs words beyond that)
  uint32_t     prologue_shard = 0, prologue_nshards = 1; // the shard the index build's classification was made for
  uint64_t     prologue_own = 0;      // ... and the visits of its owner reads
  bool         full_scan_ok = false;  // cand_off holds the exclusive scan of ALL reads' visit counts (k_index_epilogue), index_total its sum
  uint64_t     index_total = 0, own_accum = 0; // own_accum: the SC_OWN sum the host has accounted for (windows add to it)
  bool         nlists_clean = false; // the four list cursors of k_classify_reads are zero (see msgpu_calculate_edges)
  bool         cand_zeroed = false; // ... including the zeroing of the candidate kernels' counters (used up by the next msgpu_calculate_edges)
  uint64_t     prologue_bound = 0;
  uint32_t     prologue_lists[4] = {0, 0, 0, 0};
  uint64_t     readback_seq = 0;
  uint64_t     lost_publications = 0; // read-backs whose publication never arrived (wait_scalars fell back to a copy)
  bool         readback_polled = false;
  hipEvent_t   ev_readback = nullptr; // the synchronising read-back path waits for the copy only
  hipEvent_t   ev_order = nullptr;    // msgpu_stream_wait / msgpu_stream_release
  uint32_t     decl_V = 0, decl_A = 0; // msgpu_set_id_space: id counts declared by the caller (0 = find them)
  // msgpu_set_deadline: the host waits of this context's calls give up at this point in time (a collective of the caller's
  // that never completes in front of our work on the stream must not hold a libms caller for ever)
  bool                                  has_deadline = false;
  std::chrono::steady_clock::time_point deadline{};

  // loaded rows
  uint64_t n_rows = 0, n_alive = 0;
  uint32_t V = 0, A = 0;
  const msgpu_row *d_rows = nullptr; // either rows_in.p or the caller's device buffer

  // results
  uint64_t n_edges = 0, n_ems = 0, n_orders = 0, n_ids = 0, n_visit = 0, total_bound = 0;
  uint32_t n_list[4] = {0, 0, 0, 0};

  // arena
  DevBuf rows_in, rows_pk, cnt_read, read_off, cursor, bkt_key, bkt_dead, by_read, read_cnt, alive_rank,
      anchor_cnt, anchor_off, anchor_first, anchor_off_gen, bkt2_idx, bkt2_line, by_anchor, read_len, read_first, scalars,
      scan_tmp, vis16, spos2, visits, bin_cursor, bin_start;
  DevBuf bound, cand_off, cand_j, cand_t, scr_v2, scr_start, n_cand, n_edge, lists, edges, edge_cand;
  DevBuf big_key, big_t, big_r2s, big_pfx, pair_tab, chain_chunks, big_off, cand_sums, bucket_visits;
  hipStream_t side_stream = nullptr, side_stream2 = nullptr;
  hipEvent_t  ev_side[2]  = {nullptr, nullptr}, ev_side2 = nullptr;
  uint64_t    n_big_edges = 0, n_big_ems = 0;
  bool   fast_path = true;
  bool   sub_wave  = true; // short edges share a wavefront (k_chain_sub); MSGPU_NO_SUBWAVE=1 sends them all to k_chain
  uint32_t n_cls[4] = {0, 0, 0, 0}; // edges of 9..16, 17..32, 33..64 and <= 8 EdgeMatches
  uint64_t n_edges_fast = 0;
  DevBuf ems, order_scr, ids_scr, edge_norders, edge_nids, orders, ids, big_list, cls_list, big_elems,
      big_paths;
  DevBuf g_deg, g_off, g_adj, g_cand, g_sane, g_out; // findContractionEdges
  DevBuf sel_idx, sel_cnt, sel_off, sel_ems;          // msgpu_get_edgematches

  // batched execution (msgpu_overlap_batched): second set of output tables, copy stream, pinned result arena
  DevBuf      alt_edges, alt_ems, alt_orders, alt_ids;
  hipStream_t copy_stream = nullptr;
  hipEvent_t  ev_done[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr}, ev_wall[2] = {nullptr, nullptr};
  struct HostBuf {
    void  *p   = nullptr;
    size_t cap = 0;
  } h_edges, h_ems, h_orders, h_ids, h_read_len, h_read_first, h_sel_off, h_sel_ems, h_wire[2];
  hipEvent_t ev_part[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; // a window's wire blocks, copied one by one
  // how often the chain kernels have been launched: another thread can wait for the next launch (msgpu_wait_chain_launch) to
  // put its own device work beside the chain stage instead of beside the memory-bound stages before it
  std::mutex              gate_m;
  std::condition_variable gate_cv;
  uint64_t                chain_launches = 0;
  DevBuf win_cuts;        // the dispatcher's window cuts by measured work (k_window_cuts)
  DevBuf wire_dev[2];     // a window's edge / order / id tables in wire form, on their way to the host (two sets: see the dispatcher)
  bool   wire_copy = true; // MSGPU_NO_WIRE_COPY=1: windows always leave as whole records (A/B switch)

  // timing
  hipEvent_t ev[10] = {nullptr};
  // the two events around the chain kernels, one pair per msgpu_chaining_and_overlaps call in a ring: msgpu_get_timings
  // averages over the calls since the last msgpu_get_timings without having synchronised after each of them
  static constexpr int CK_RING = 256;
  hipEvent_t ck_ev[CK_RING][2] = {};
  uint32_t   ck_head = 0, ck_count = 0; // next slot; pairs recorded since the last msgpu_get_timings
  bool       have_index_t = false, have_cand_t = false, have_chain_t = false, index_fast = false;
  bool   have_stage_t = false;
  uint32_t index_path = 0;  // MSGPU_INDEX_* of the last index build
  bool     index_binned = false; // the last build_index_once ran the bin path's kernels
  bool     use_bin = true;  // try the bin path first (MSGPU_NO_BIN=1: never)
};

namespace {

// scalar slots in ctx->scalars (uint64 each)
enum { SC_MAXIDS = 0 /*2 x u32*/, SC_ERR = 1, SC_TOTAL_A = 2, SC_TOTAL_B = 3, SC_TOTAL_C = 4, SC_NLISTS = 5 /*4 x u32, spans 5..6*/,
       SC_NBIG = 7, SC_NALIVE = 8, SC_IXFLAGS = 9, SC_BIGSTATS = 10 /*2 x u64*/, SC_BIGCUR = 12 /*2 x u64*/,
       SC_CLS = 14 /*4 x u32: edges per width class, spans 14..15*/,
       SC_HEADS = 16 /*k_index_bin: finished workgroups (low half) | scaffolds that begin (high half)*/, SC_DONE = 17 /*u32: finished workgroups of k_index_epilogue*/,
       SC_OWN = 18 /*visits of the owner reads classified by k_index_epilogue*/, SC_COUNT = 19 };

int fail(msgpu_ctx *c, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(c->err, sizeof(c->err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(c, expr)                                                                                                \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess)                                                                                              \
      return fail((c), _e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "%s failed: %s (%s:%d)", #expr,        \
                  hipGetErrorString(_e), __FILE__, __LINE__);                                                          \
  } while (0)

#define ENSURE(c, buf, bytes) HIPCHK(c, (c)->buf.ensure(bytes))

template <class T> T *scalar(msgpu_ctx *c, int slot) { return reinterpret_cast<T *>(c->scalars.as<uint64_t>() + slot); }
// host value of a scalar slot after read_scalars()
template <class T> const T *host_scalar(const msgpu_ctx *c, int slot) { return reinterpret_cast<const T *>(c->h_scalars + slot); }
// The candidate stage's per-chunk block (cand_sums): [chunks][2] 64-bit sums, then [chunks][64] 32-bit size histograms.
static uint32_t cand_chunks(uint32_t V) { return (V + CAND_CHUNK - 1) / CAND_CHUNK; }
static size_t   cand_sums_bytes(uint32_t V) { return static_cast<size_t>(cand_chunks(V) + 1) * (16 + 256); }
static unsigned long long *cand_chunk_sums(msgpu_ctx *c) { return c->cand_sums.as<unsigned long long>(); }
static uint32_t *cand_hist(msgpu_ctx *c, uint32_t V) { return reinterpret_cast<uint32_t *>(c->cand_sums.as<unsigned long long>() + 2 * static_cast<size_t>(cand_chunks(V) + 1)); }
// what k_classify_reads zeroes in front of the candidate kernels (buffers must exist)
static CandZero cand_zero(msgpu_ctx *c, uint32_t V) {
  static_assert(SC_BIGCUR == SC_BIGSTATS + 2 && SC_CLS == SC_BIGCUR + 2 && SC_HEADS == SC_CLS + 2, "adjacent scalar slots");
  static_assert(SC_COUNT <= SC_PUBLISH_MAX, "a fused read-back publishes the whole scalar block");
  CandZero z;
  z.n_cand         = c->n_cand.as<uint32_t>();
  z.n_edge         = c->n_edge.as<uint32_t>();
  z.scalar_words   = scalar<uint32_t>(c, SC_BIGSTATS);
  z.n_scalar_words = 12; // big-edge statistics (2 x u64), big-edge cursors (2 x u64), class counts (4 x u32)
  return z;
}

// One copy of the whole scalar block into pinned memory + a stream synchronisation.  (Separate 4-byte copies into
// pageable host variables cost ~20 us each on this stack; there were up to three per read-back.)
// Default: no copy and no stream synchronisation -- a one-wavefront kernel writes the block into the (mapped) pinned
// mirror and publishes a sequence number; the host polls for it (about half the latency of copy + synchronise, and
// the host is back on the stream sooner).  A stream that stops making progress (a failed launch) is noticed by
// hipStreamQuery and handled by the synchronising path, which is also what MSGPU_SYNC_READBACK=1 selects.
// publish_scalars() enqueues the publication, wait_scalars() polls for it: work that does not depend on the values can
// be enqueued in between and keeps the GPU busy while the host turns around.
bool past_deadline(const msgpu_ctx *c) { return c->has_deadline && std::chrono::steady_clock::now() >= c->deadline; }
// A host wait for a stream (or an event) that honours the context's deadline: without one the runtime's blocking wait, with
// one a poll that gives up with MSGPU_E_TIMEOUT and leaves the work queued (the caller aborts what blocks it, or destroys).
int host_sync(msgpu_ctx *c, hipStream_t st, hipEvent_t ev = nullptr) {
  if (!c->has_deadline) {
    if (ev) HIPCHK(c, hipEventSynchronize(ev));
    else HIPCHK(c, hipStreamSynchronize(st));
    return MSGPU_OK;
  }
  for (uint32_t spins = 0;; ++spins) {
    const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(st);
    if (q == hipSuccess) return MSGPU_OK;
    if (q != hipErrorNotReady) return fail(c, MSGPU_E_HIP, "%s failed: %s", ev ? "hipEventQuery" : "hipStreamQuery", hipGetErrorString(q));
    if (past_deadline(c))
      return fail(c, MSGPU_E_TIMEOUT, "the context's stream did not drain before the deadline (msgpu_set_deadline): work is still queued");
    if (spins < 2000) __builtin_ia32_pause();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

int publish_scalars(msgpu_ctx *c, hipEvent_t mark = nullptr) {
  static const bool sync_path = getenv("MSGPU_SYNC_READBACK") != nullptr;
  c->readback_polled = !sync_path && c->h_scalars_dev;
  if (c->readback_polled) {
    launch_publish_scalars(c->stream, c->scalars.as<uint64_t>(), c->h_scalars_dev, SC_COUNT, ++c->readback_seq);
  } else {
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, SC_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_readback, c->stream));
  }
  if (mark) HIPCHK(c, hipEventRecord(mark, c->stream));
  return MSGPU_OK;
}
int wait_scalars(msgpu_ctx *c) {
  if (c->readback_polled) {
    const uint64_t     seq  = c->readback_seq;
    volatile uint64_t *flag = c->h_scalars + SC_COUNT;
    for (uint64_t spins = 1;; ++spins) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return MSGPU_OK;
      __builtin_ia32_pause();
      if ((spins & 0xffff) == 0) { // every ~1 ms: is the stream still alive?
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) { // everything ran: the flag is there, or something is badly wrong
          if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return MSGPU_OK;
          break;
        }
        if (q != hipErrorNotReady) break;
        if (past_deadline(c))
          return fail(c, MSGPU_E_TIMEOUT, "table sizes did not come back before the deadline (msgpu_set_deadline): the stream is "
                                          "held up by work queued in front of ours");
      }
    }
    // The stream stopped making progress, or finished without the publication arriving in mapped memory.  Take the values
    // the slow way, surface a stream error if there is one, and leave a trace either way: a lost publication is counted
    // (msgpu_counts.n_lost_publications).  The error text is NOT touched on a call that goes on to succeed (msgpu.h,
    // STREAM AND THREAD CONTRACT rule 4: the text belongs to a non-zero return code).
    ++c->lost_publications;
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, SC_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return host_sync(c, c->stream);
  }
  return host_sync(c, nullptr, c->ev_readback); // the copy, not whatever was enqueued behind it
}
int read_scalars(msgpu_ctx *c, hipEvent_t mark = nullptr) {
  if (int rc = publish_scalars(c, mark)) return rc;
  return wait_scalars(c);
}

void release_all(msgpu_ctx *c) {
  DevBuf *all[] = {&c->sel_idx, &c->sel_cnt, &c->sel_off, &c->sel_ems, &c->g_deg, &c->g_off, &c->g_adj, &c->g_cand, &c->g_sane, &c->g_out, &c->rows_in, &c->rows_pk, &c->cnt_read, &c->read_off, &c->cursor, &c->bkt_key,
                   &c->bkt_dead, &c->by_read, &c->read_cnt, &c->alive_rank, &c->anchor_cnt, &c->anchor_off,
                   &c->anchor_first, &c->anchor_off_gen,
                   &c->bkt2_idx, &c->bkt2_line, &c->by_anchor, &c->read_len, &c->read_first, &c->scalars, &c->scan_tmp,
                   &c->bound, &c->cand_off, &c->cand_j, &c->cand_t, &c->scr_v2, &c->scr_start, &c->n_cand, &c->n_edge,
                   &c->lists, &c->edges, &c->edge_cand,
                   &c->big_key, &c->big_t, &c->big_r2s, &c->big_pfx, &c->pair_tab, &c->chain_chunks, &c->big_off, &c->cand_sums, &c->bucket_visits, &c->ems, &c->order_scr, &c->ids_scr,
                   &c->edge_norders, &c->edge_nids, &c->orders, &c->ids, &c->big_list, &c->cls_list, &c->big_elems, &c->big_paths, &c->alt_edges, &c->alt_ems, &c->alt_orders, &c->alt_ids, &c->vis16, &c->visits,
                   &c->spos2, &c->bin_cursor, &c->bin_start, &c->wire_dev[0], &c->wire_dev[1], &c->win_cuts};
  for (DevBuf *b : all) b->release();
}

int build_index_once(msgpu_ctx *c, bool force_generic, bool two_pass, bool bin, uint32_t *ix_flags_out) {
  hipStream_t st = c->stream;
  const uint64_t n = c->n_rows;
  {
    const void *before = c->scalars.p;
    ENSURE(c, scalars, SC_COUNT * sizeof(uint64_t));
    if (c->scalars.p != before) c->scalars_clean = false;
  }
  // The scalar block is zero at rest where a build counts from nothing (k_index_epilogue's last workgroup leaves it so, behind
  // its publication): a build that follows one of those needs no memset.  Anything else -- a first build, a build after an
  // error, the synchronising read-back path, ids to be discovered -- zeroes the block here.
  bool scalars_zeroed = false;
  auto zero_scalars = [&]() -> int {
    if (scalars_zeroed) return MSGPU_OK;
    HIPCHK(c, hipMemsetAsync(c->scalars.p, 0, SC_COUNT * sizeof(uint64_t), st));
    scalars_zeroed  = true;
    c->nlists_clean = true;
    return MSGPU_OK;
  };

  // id spaces: declared by the caller (the parser knows them), else one pass over the rows and a read-back
  if (c->decl_V && c->decl_A) {
    c->V = c->decl_V;
    c->A = c->decl_A;
  } else {
    if (int rc = zero_scalars()) return rc;
    launch_max_ids(st, c->d_rows, n, scalar<uint32_t>(c, SC_MAXIDS));
    if (int rc = read_scalars(c)) return rc;
    c->V = host_scalar<uint32_t>(c, SC_MAXIDS)[0];
    c->A = host_scalar<uint32_t>(c, SC_MAXIDS)[1];
  }
  const uint32_t V = c->V, A = c->A;

  const size_t nz = n ? n : 1;
  const size_t mva = size_t(V > A ? V : A) + 2;
  // One-pass bucketing: every read owns BUCKET_CAP slots, so the rows can be dropped into their read's bucket by the
  // same pass that counts them (no offsets needed yet).  Used while that fits (reads x 128 x 32 B <= 4 GiB); a read
  // with more rows raises IXF_OVERFLOW and build_index() comes back here with two_pass = true.
  constexpr uint32_t BUCKET_CAP = 128;
  // The bin path (msgpu_index.hip): no global atomic per row.  Rows go to coarse buckets of 16 / 32 / 64 consecutive read ids
  // (k_index_bin), a workgroup per bucket groups and ranks them (k_index_sort_bin).  Covers what msgpu_parse_paf hands
  // over; anything else raises a flag and build_index() comes back with bin = false.
  const uint32_t reads_per_pass = BIN_NB_MAX << BIN_RPB_SHIFT;
  const uint32_t bpasses = (bin && !force_generic && !two_pass && n && V) ? static_cast<uint32_t>((size_t(V) + reads_per_pass - 1) / reads_per_pass) : 0;
  uint32_t       bcap    = (bpasses && bpasses <= BIN_PASSES_MAX) ? bin_capacity(n, V) : 0;
  if (bcap && !index_sort_bin_prepare(bcap)) bcap = 0; // (the device does not grant the bucket's LDS: the atomic path from the start)
  const uint32_t bshift  = bcap ? BIN_RPB_SHIFT : 0; // != 0: this build takes the bin path
  const uint32_t nb      = bshift ? static_cast<uint32_t>((size_t(std::min(V, reads_per_pass)) + (1u << bshift) - 1) >> bshift) : 0; // buckets of a (full) pass
  const uint32_t     cap = (!bshift && !two_pass && V && size_t(V) * BUCKET_CAP * sizeof(IRow) <= (size_t(4) << 30)) ? BUCKET_CAP : 0;
  ENSURE(c, cnt_read, (size_t(V) + 1) * 4);
  ENSURE(c, read_off, (size_t(V) + 2) * 4);
  ENSURE(c, cursor, mva * 4);
  if (bshift) {
    const void *before = c->bin_cursor.p;
    ENSURE(c, bin_cursor, (size_t(bpasses) * (nb + 1) + 1) * 4); // per pass: the bucket cursors; last word: by_read rows of the passes so far
    if (c->bin_cursor.p != before) {
      c->bin_clean      = false;
      c->bin_zero_words = 0;
    }
    if (size_t(bpasses) * (nb + 1) + 1 > c->bin_zero_words) c->bin_clean = false; // (words no init launch has reached yet)
    ENSURE(c, bin_start, (size_t(nb) + 2) * 4);
    ENSURE(c, bucket_visits, ((size_t(V) >> BIN_RPB_SHIFT) + 2) * 4);
  }
  static const bool sync_path = getenv("MSGPU_SYNC_READBACK") != nullptr;
  const bool fused_readback = bshift != 0 && !sync_path && c->h_scalars_dev; // k_index_epilogue publishes (and zeroes behind it)
  if (!(bshift && c->scalars_clean && fused_readback))
    if (int rc = zero_scalars()) return rc;
  c->scalars_clean = false; // (until this build's publisher has left it so again)
  ENSURE(c, bkt_key, (bshift ? size_t(nb) * bcap * 2 : cap ? size_t(V) * cap : nz) * sizeof(IRow)); // (bin path: 64-byte records)
  ENSURE(c, bkt_dead, nz);
  ENSURE(c, by_read, nz * sizeof(IRow));
  ENSURE(c, read_cnt, (size_t(V) + 1) * 4);
  ENSURE(c, alive_rank, nz * 4);
  ENSURE(c, anchor_cnt, (size_t(A) + 1) * 4);
  ENSURE(c, anchor_first, (size_t(A) + 2) * 4);
  ENSURE(c, anchor_off_gen, (size_t(A) + 2) * 4);
  ENSURE(c, anchor_off, (size_t(A) + 2) * 4);
  ENSURE(c, bkt2_idx, nz * 4);
  ENSURE(c, bkt2_line, nz * 4);
  ENSURE(c, by_anchor, nz * sizeof(IRow));
  ENSURE(c, vis16, nz * 16);
  ENSURE(c, spos2, (bshift ? 1 : cap ? size_t(V) * cap : nz) * 8); // one-pass build: by bucket slot, else by source row (bin path: inside the record)
  ENSURE(c, visits, (size_t(V) + 1) * 4);
  ENSURE(c, read_len, (size_t(V) + 1) * 4);
  ENSURE(c, read_first, (size_t(V) + 1) * 4);
  {
    uint64_t m = n;
    if (V > m) m = V;
    if (A > m) m = A;
    ENSURE(c, scan_tmp, 3 * (size_t(scan_blocks(m)) + 1) * 8);
  }

  // The opening of msgpu_calculate_edges for the whole table (scratch offsets from the visit counts the sort leaves, owner
  // reads classified by LDS footprint, counters zeroed) runs inside the index build, so that its two numbers come back with
  // the index flags instead of costing a read-back of their own.  Valid for a fast index of an unsharded context; anything
  // else redoes it there.
  c->prologue_ok           = false;
  static const bool env_no_prologue = getenv("MSGPU_NO_PROLOGUE") != nullptr; // measurement switch
  // (the bin path classifies for the context's shard; the atomic path's prologue is the unsharded one)
  const bool want_prologue = !env_no_prologue && !force_generic && !c->no_prologue && V != 0 && (c->nshards == 1 || bshift) && c->win_lo == 0 && c->win_hi >= V;
  if (want_prologue) {
    ENSURE(c, n_cand, (size_t(V) + 1) * 4);
    ENSURE(c, n_edge, (size_t(V) + 1) * 4);
    ENSURE(c, cand_sums, cand_sums_bytes(V));
  }
  size_t zero_words_known = 0;
  {
    // (the bin path counts per bucket, not per read: its cursors take the per-read counters' place in the zero list.  What the
    // candidate kernels add to is zeroed by k_classify_reads, the launch in front of them.)
    uint32_t *const zero[8]   = {bshift ? c->bin_cursor.as<uint32_t>() : c->cnt_read.as<uint32_t>(), c->cursor.as<uint32_t>(),
                                 c->read_cnt.as<uint32_t>(), c->anchor_cnt.as<uint32_t>(), nullptr, nullptr, nullptr, nullptr};
    const uint32_t  n_zero[8] = {bshift ? bpasses * (nb + 1) + 1 : V + 1, static_cast<uint32_t>(mva), V + 1, A + 1, 0, 0, 0, 0};
    uint32_t *const ones[2]   = {c->anchor_first.as<uint32_t>(), nullptr};
    const uint32_t  n_ones[2] = {A + 2, 0};
    // the bin path needs its bucket cursors zero and nothing else of this list (the cursors are zero at rest: k_index_sort_bin
    // leaves them so; sparse anchor ids are found by counting scaffolds, not by looking into a filled anchor_first)
    if (!bshift) launch_index_init8(st, zero, n_zero, ones, n_ones);
    else if (!c->bin_clean) {
      c->bin_zero_words     = std::max<size_t>(c->bin_zero_words, n_zero[0]);
      uint32_t *const z4[4] = {zero[0], nullptr, nullptr, nullptr};
      const uint32_t  n4[4] = {n_zero[0], 0, 0, 0};
      uint32_t *const o2[2] = {nullptr, nullptr};
      const uint32_t  no2[2] = {0, 0};
      launch_index_init(st, z4, n4, o2, no2);
    }
    c->bin_clean = false;
    if (bshift) { // while a build is in flight nothing is known to be zero: it is again once the build has come back clean
      zero_words_known  = c->bin_zero_words;
      c->bin_zero_words = 0;
    }
  }
  uint32_t *d_flags = scalar<uint32_t>(c, SC_IXFLAGS);
  if (force_generic) {
    const uint32_t f = IXF_FORCE;
    HIPCHK(c, hipMemcpyAsync(d_flags, &f, 4, hipMemcpyHostToDevice, st));
  }

  if (bshift) {
    uint32_t *row_base = c->bin_cursor.as<uint32_t>() + size_t(bpasses) * (nb + 1);
    for (uint32_t p = 0; p < bpasses; ++p) { // one pass per 131,072 reads (BASELINE.json configs[2]: one)
      const uint32_t rd_lo = p * reads_per_pass;
      const uint32_t nb_p  = static_cast<uint32_t>((size_t(std::min(V - rd_lo, reads_per_pass)) + (1u << bshift) - 1) >> bshift);
      uint32_t      *cur   = c->bin_cursor.as<uint32_t>() + size_t(p) * (nb + 1);
      BinTail        tail;
      tail.bin_start    = c->bin_start.as<uint32_t>();
      tail.row_base     = row_base;
      tail.read_off_end = p + 1 == bpasses ? c->read_off.as<uint32_t>() + V : nullptr;
      tail.done_heads   = scalar<unsigned long long>(c, SC_HEADS);
      launch_index_bin(st, c->d_rows, n, V, A, d_flags, scalar<uint32_t>(c, SC_ERR), c->anchor_first.as<uint32_t>(), cur,
                       c->bkt_key.as<uint4>(), rd_lo, nb_p, bcap, tail);
      launch_index_sort_bin(st, cur, c->bin_start.as<uint32_t>(), V, rd_lo, nb_p, bcap, c->bkt_key.as<uint4>(),
                            c->by_read.as<IRow>(), c->by_anchor.as<IRow>(), c->vis16.as<uint4>(), c->read_off.as<uint32_t>(),
                            c->read_cnt.as<uint32_t>(), c->read_len.as<int32_t>(), c->read_first.as<uint32_t>(),
                            c->visits.as<uint32_t>(), c->d_rows, d_flags, scalar<uint32_t>(c, SC_ERR), c->bucket_visits.as<uint32_t>());
    }
  } else {
  launch_index_pass1(st, c->d_rows, n, c->cnt_read.as<uint32_t>(), c->anchor_first.as<uint32_t>(), V, A, d_flags,
                     scalar<uint32_t>(c, SC_ERR), c->bkt_key.as<IRow>(), cap, c->spos2.as<uint2>());
  exclusive_scan<uint32_t>(st, c->cnt_read.as<uint32_t>(), V, c->read_off.as<uint32_t>(), c->scan_tmp.as<uint32_t>(),
                           scalar<uint32_t>(c, SC_TOTAL_A));
  if (!cap) launch_scatter_read(st, c->d_rows, n, c->read_off.as<uint32_t>(), c->cursor.as<uint32_t>(), c->bkt_key.as<IRow>());
  launch_sort_read(st, c->read_off.as<uint32_t>(), c->cnt_read.as<uint32_t>(), V, c->bkt_key.as<IRow>(),
                   c->by_read.as<IRow>(), c->read_cnt.as<uint32_t>(),
                   c->alive_rank.as<uint32_t>(), c->anchor_cnt.as<uint32_t>(), c->bkt_dead.as<uint8_t>(), d_flags,
                   c->by_anchor.as<IRow>(), cap, c->d_rows, c->read_len.as<int32_t>(), c->read_first.as<uint32_t>(),
                   scalar<uint32_t>(c, SC_ERR), c->spos2.as<uint2>(), c->vis16.as<uint4>(), c->visits.as<uint32_t>()); // fast mode: the sort writes the scaffold rows too (at
                                                                              // the places pass 1 left in spos2); always: the Vertex facts
  }
  if (bshift) {
    // ONE launch closes a bin-path build (k_index_epilogue): the Registry-order check, the scaffold offsets, the scan of the
    // visit counts, the owner reads classified for the candidate kernels (for this context's shard; a window job classifies
    // per window), and the read-back of flags and sizes by its last workgroup -- which also leaves the scalar block zero where
    // the next build counts from nothing.
    ENSURE(c, cand_off, (size_t(V) + 2) * 8);
    ENSURE(c, lists, (size_t(V) + 1) * (3 * sizeof(CandDesc) + 4));
    CandDesc *l0 = c->lists.as<CandDesc>(), *l1 = l0 + V + 1, *l2 = l1 + V + 1;
    if (want_prologue && !c->nlists_clean) {
      HIPCHK(c, hipMemsetAsync(scalar<uint32_t>(c, SC_NLISTS), 0, 16, st));
      c->nlists_clean = true;
    }
    IndexEpilogueArgs k;
    k.read_first     = c->read_first.as<uint32_t>();
    k.V              = V;
    k.A              = A;
    k.n_rows         = static_cast<uint32_t>(n);
    k.err            = scalar<uint32_t>(c, SC_ERR);
    k.flags          = d_flags;
    k.anchor_first   = c->anchor_first.as<uint32_t>();
    k.anchor_off_gen = c->anchor_off_gen.as<uint32_t>();
    k.anchor_off     = c->anchor_off.as<uint32_t>();
    k.n_alive        = scalar<uint32_t>(c, SC_NALIVE);
    k.heads          = scalar<uint32_t>(c, SC_HEADS) + 1; // (the high half)
    k.row_base       = c->bin_cursor.as<uint32_t>() + size_t(bpasses) * (nb + 1);
    k.visits         = c->visits.as<uint32_t>();
    k.bucket_visits  = c->bucket_visits.as<uint32_t>();
    k.n_buckets      = static_cast<uint32_t>((size_t(V) + (1u << BIN_RPB_SHIFT) - 1) >> BIN_RPB_SHIFT);
    k.cand_off       = c->cand_off.as<uint64_t>();
    k.total          = scalar<uint64_t>(c, SC_TOTAL_A);
    k.classify       = want_prologue ? 1 : 0;
    k.read_off       = c->read_off.as<uint32_t>();
    k.read_cnt       = c->read_cnt.as<uint32_t>();
    k.shard          = c->shard;
    k.nshards        = c->nshards;
    k.list0          = l0;
    k.list1          = l1;
    k.list2          = l2;
    k.list3          = reinterpret_cast<uint32_t *>(l2 + V + 1);
    k.n_lists        = scalar<uint32_t>(c, SC_NLISTS);
    k.own_total      = scalar<unsigned long long>(c, SC_OWN);
    k.z              = want_prologue ? cand_zero(c, V) : CandZero{};
    k.done           = scalar<uint32_t>(c, SC_DONE);
    k.scalars        = c->scalars.as<uint64_t>();
    k.host_scalars   = fused_readback ? c->h_scalars_dev : nullptr;
    k.n_scalars      = SC_COUNT;
    c->readback_polled = fused_readback;
    k.seq            = fused_readback ? ++c->readback_seq : 0;
    // zeroed behind the publication: error bits, index flags, the scaffold count, the finished-workgroup counters, the owner
    // reads' visits.  (Not with the synchronising read-back: its copy comes after the kernel.)
    k.zero_mask      = fused_readback ? ((1ull << SC_ERR) | (1ull << SC_IXFLAGS) | (1ull << SC_HEADS) | (1ull << SC_DONE) | (1ull << SC_OWN)) : 0ull;
    launch_index_epilogue(st, k);
    if (want_prologue) c->nlists_clean = false;
    HIPCHK(c, hipGetLastError());
    if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[1], st));
    if (!fused_readback) {
      HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, SC_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIPCHK(c, hipEventRecord(c->ev_readback, st));
    }
    if (int rc = wait_scalars(c)) return rc;
  } else {
  // the Registry-order check on the first lines the sort found and, in the same launch, the scaffold offsets: fast mode
  // (input grouped by anchor, ascending lines: what the PAF loader hands over) is finished here but for them, and they are
  // the speculative ones of pass 1; the flags come back with the read-back below and only an input that is not in that
  // form pays for the generic scaffold build (a second read-back).
  launch_index_finish(st, c->read_first.as<uint32_t>(), V, scalar<uint32_t>(c, SC_ERR), d_flags, c->anchor_first.as<uint32_t>(),
                      c->anchor_off_gen.as<uint32_t>(), A, c->anchor_off.as<uint32_t>(), scalar<uint32_t>(c, SC_NALIVE),
                      static_cast<uint32_t>(n));
  HIPCHK(c, hipGetLastError());
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[1], st));
  if (want_prologue) { // (see the top of this function)
    ENSURE(c, cand_off, (size_t(V) + 2) * 8);
    ENSURE(c, lists, (size_t(V) + 1) * (3 * sizeof(CandDesc) + 4));
    CandDesc *l0 = c->lists.as<CandDesc>(), *l1 = l0 + V + 1, *l2 = l1 + V + 1;
    exclusive_scan<uint64_t>(st, c->visits.as<uint32_t>(), V, c->cand_off.as<uint64_t>(), c->scan_tmp.as<uint64_t>(),
                             scalar<uint64_t>(c, SC_TOTAL_A));
    launch_classify_reads(st, c->read_off.as<uint32_t>(), c->read_cnt.as<uint32_t>(), c->visits.as<uint32_t>(),
                          c->cand_off.as<uint64_t>(), V, 0, 1, 0, 0xffffffffu, l0, l1, l2,
                          reinterpret_cast<uint32_t *>(l2 + V + 1), scalar<uint32_t>(c, SC_NLISTS), cand_zero(c, V));
    c->nlists_clean = false;
    HIPCHK(c, hipGetLastError());
  }

  if (int rc = read_scalars(c)) return rc;
  }
  const uint32_t err = *host_scalar<uint32_t>(c, SC_ERR), ixf = *host_scalar<uint32_t>(c, SC_IXFLAGS);
  c->cand_zeroed = false;
  c->full_scan_ok = bshift && ixf == 0 && err == 0;
  c->index_total  = c->full_scan_ok ? *host_scalar<uint64_t>(c, SC_TOTAL_A) : 0;
  c->own_accum    = (bshift && fused_readback) ? 0 : *host_scalar<uint64_t>(c, SC_OWN); // (the publisher zeroed it, or it stands)
  if (bshift && fused_readback && ixf == 0 && err == 0) { // (zero at rest, see above)
    c->scalars_clean = c->bin_clean = true;
    c->bin_zero_words = zero_words_known;
  }
  if (want_prologue && ixf == 0 && err == 0) {
    c->prologue_ok      = true;
    c->cand_zeroed      = true;
    c->prologue_bound   = *host_scalar<uint64_t>(c, SC_TOTAL_A);
    c->prologue_shard   = bshift ? c->shard : 0;
    c->prologue_nshards = bshift ? c->nshards : 1;
    c->prologue_own     = (bshift && c->nshards > 1) ? *host_scalar<uint64_t>(c, SC_OWN) : c->prologue_bound;
    for (int k = 0; k < 4; ++k) c->prologue_lists[k] = host_scalar<uint32_t>(c, SC_NLISTS)[k];
  }
  uint32_t       n_alive = *host_scalar<uint32_t>(c, SC_NALIVE);
  if (err & 2u)
    return fail(c, MSGPU_E_IDS, "a row has an id outside the declared id space (%u reads, %u anchors)", V, A);
  c->index_binned = bshift != 0;
  if (bshift && ixf != 0) { // the bin path covers the loader's form only: nothing of this build is kept, the atomic path decides
    c->prologue_ok = c->cand_zeroed = false;
    *ix_flags_out  = ixf | IXF_BINFAIL;
    return MSGPU_OK;
  }
  if (ixf & IXF_OVERFLOW) { // a read did not fit its bucket: nothing of this build is kept
    *ix_flags_out = ixf;
    return MSGPU_OK;
  }
  if ((ixf & ~IXF_DUPS) != 0 && !(err & 1u)) {
    // generic scaffold build: count, scan, bucket by anchor, rank by line (MatchMap.cpp:178-183)
    exclusive_scan<uint32_t>(st, c->anchor_cnt.as<uint32_t>(), A, c->anchor_off_gen.as<uint32_t>(),
                             c->scan_tmp.as<uint32_t>(), scalar<uint32_t>(c, SC_NALIVE));
    launch_select_anchor_off(st, d_flags, c->anchor_first.as<uint32_t>(), c->anchor_off_gen.as<uint32_t>(), A,
                             c->anchor_off.as<uint32_t>(), scalar<uint32_t>(c, SC_NALIVE), static_cast<uint32_t>(n));
    HIPCHK(c, hipMemsetAsync(c->cursor.p, 0, mva * 4, st));
    launch_scatter_anchor(st, c->d_rows, n, c->alive_rank.as<uint32_t>(), c->anchor_off.as<uint32_t>(),
                          c->cursor.as<uint32_t>(), c->bkt2_idx.as<uint32_t>(), c->bkt2_line.as<uint32_t>(), d_flags);
    launch_rank_anchor(st, c->anchor_off.as<uint32_t>(), n, scalar<uint32_t>(c, SC_NALIVE), c->bkt2_idx.as<uint32_t>(),
                       c->bkt2_line.as<uint32_t>(), c->d_rows, c->alive_rank.as<uint32_t>(), c->by_anchor.as<IRow>(),
                       d_flags, c->read_off.as<uint32_t>(), c->by_read.as<IRow>(), c->vis16.as<uint4>());
    HIPCHK(c, hipGetLastError());
    if (int rc = read_scalars(c, c->ev[1])) return rc;
    n_alive = *host_scalar<uint32_t>(c, SC_NALIVE);
  }
  c->n_alive    = n_alive;
  *ix_flags_out = ixf;
  if (err & 1u)
    return fail(c, MSGPU_E_IDS,
                "read ids are not dense Registry ids in first-line order (Registry.cpp:36-45): use msgpu_parse_paf ids");
  return MSGPU_OK;
}

int build_index(msgpu_ctx *c) {
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  uint32_t ixf = 0;
  bool     two_pass = false;
  int      rc  = build_index_once(c, false, two_pass, c->use_bin, &ixf);
  if (rc != MSGPU_OK) return rc;
  const bool binned = c->use_bin && !(ixf & IXF_BINFAIL) && c->index_binned;
  if (ixf & IXF_BINFAIL) { // not the loader's form (or beyond the bin path's capacities): the atomic path covers every input
    ixf = 0;
    rc  = build_index_once(c, false, two_pass, false, &ixf);
    if (rc != MSGPU_OK) return rc;
  }
  if (ixf & IXF_OVERFLOW) { // a read with more rows than a one-pass bucket holds: count, scan, scatter instead
    two_pass = true;
    rc       = build_index_once(c, false, two_pass, false, &ixf);
    if (rc != MSGPU_OK) return rc;
  }
  // the fast by_anchor path assumed no duplicate (read, anchor) pair; if one turned up, rebuild generically
  if ((ixf & IXF_DUPS) && (ixf & ~IXF_DUPS) == 0) {
    rc = build_index_once(c, true, two_pass, false, &ixf);
    if (rc != MSGPU_OK) return rc;
  }
  c->index_fast   = (ixf & ~IXF_DUPS) == 0;
  c->index_path   = (binned ? MSGPU_INDEX_BIN : two_pass ? MSGPU_INDEX_TWO_PASS : MSGPU_INDEX_ATOMIC) | (c->index_fast ? 0u : MSGPU_INDEX_GENERIC);
  c->have_index_t = c->stage_events;
  c->state        = ST_LOADED;
  return MSGPU_OK;
}

} // namespace

extern "C" {

void msgpu_default_params(msgpu_params *p) {
  if (!p) return;
  p->min_matches = 400;
  p->th_length   = 500;
  p->th_matches  = 500;
  p->th_overlap  = 100;
  p->wiggle_room = 300;
  p->ratio_pct   = 15;
  p->alt_frac    = 0.75;
}

const char *msgpu_strerror(int code) {
  switch (code) {
  case MSGPU_OK: return "ok";
  case MSGPU_E_IO: return "can't open blast file";
  case MSGPU_E_FORMAT: return "invalid BLAST file";
  case MSGPU_E_NUMBER: return "invalid integer field in BLAST file";
  case MSGPU_E_NOMEM: return "out of memory";
  case MSGPU_E_ARG: return "invalid argument (unexpected nullptr)";
  case MSGPU_E_HIP: return "HIP runtime error";
  case MSGPU_E_STATE: return "entry points called out of order";
  case MSGPU_E_IDS: return "read ids are not in Registry (first-line) order";
  case MSGPU_E_NODEVICE: return "no HIP device (libmsgpu has no CPU fallback)";
  case MSGPU_E_LAYOUT: return "path cannot be assembled (the reference would terminate or hang on it)";
  case MSGPU_E_TIMEOUT: return "a deadline passed while device work was still queued";
  default: return "unknown error";
  }
}

int msgpu_create(int device, const msgpu_params *params, msgpu_ctx **out) {
  if (!out) return MSGPU_E_ARG;
  *out = nullptr;
  int        ndev = 0;
  hipError_t e    = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return MSGPU_E_NODEVICE;
  if (device < 0 || device >= ndev) return MSGPU_E_ARG;
  msgpu_ctx *c = new (std::nothrow) msgpu_ctx();
  if (!c) return MSGPU_E_NOMEM;
  c->device = device;
  if (params)
    c->p = *params;
  else
    msgpu_default_params(&c->p);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return MSGPU_E_HIP;
  }
  c->stream = c->own_stream;
  if (hipHostMalloc(reinterpret_cast<void **>(&c->h_scalars), (SC_COUNT + 1) * sizeof(uint64_t),
                    hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    c->h_scalars = nullptr;
    msgpu_destroy(c);
    return MSGPU_E_NOMEM;
  }
  memset(c->h_scalars, 0, (SC_COUNT + 1) * sizeof(uint64_t));
  if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_scalars_dev), c->h_scalars, 0) != hipSuccess) c->h_scalars_dev = nullptr;
  {
    const char *nf = getenv("MSGPU_NO_FASTPATH"); // test hook: force the full pair sweep on every edge
    c->fast_path   = !(nf && nf[0] == '1');
    const char *ns = getenv("MSGPU_NO_SUBWAVE"); // test hook: one edge per wavefront whatever its size
    c->sub_wave    = !(ns && ns[0] == '1');
    const char *nw = getenv("MSGPU_NO_WIRE_COPY");
    c->wire_copy   = !(nw && nw[0] == '1');
    const char *nbin = getenv("MSGPU_NO_BIN"); // A/B switch: the index build's atomic path (rounds 1-3) for every input
    c->use_bin       = !(nbin && nbin[0] == '1');
  }
  for (auto &ev : c->ev)
    if (hipEventCreate(&ev) != hipSuccess) {
      msgpu_destroy(c);
      return MSGPU_E_HIP;
    }
  // the side stream carries the few heavy workgroups that run beside a stage's main kernel (large LDS classes, edges with
  // more than 64 EdgeMatches): highest priority, so that they finish first and the join never waits for them
  int prio_least = 0, prio_greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_least = prio_greatest = 0;
  if (hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipStreamCreateWithPriority(&c->side_stream2, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_side2, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[0][0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[0][1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[0][2], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[1][0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[1][1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_part[1][2], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_side[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_side[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_readback, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming) != hipSuccess) {
    msgpu_destroy(c);
    return MSGPU_E_HIP;
  }
  if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    msgpu_destroy(c);
    return MSGPU_E_HIP;
  }
  for (int k = 0; k < 2; ++k)
    if (hipEventCreateWithFlags(&c->ev_done[k], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copied[k], hipEventDisableTiming) != hipSuccess) {
      msgpu_destroy(c);
      return MSGPU_E_HIP;
    }
  *out = c;
  return MSGPU_OK;
}

void msgpu_destroy(msgpu_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  release_all(c);
  for (auto &ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto &ev : c->ev_side)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_readback) (void)hipEventDestroy(c->ev_readback);
  if (c->ev_order) (void)hipEventDestroy(c->ev_order);
  if (c->ev_side2) (void)hipEventDestroy(c->ev_side2);
  for (auto &per_set : c->ev_part)
    for (hipEvent_t e : per_set)
      if (e) (void)hipEventDestroy(e);
  if (c->side_stream2) {
    (void)hipStreamSynchronize(c->side_stream2);
    (void)hipStreamDestroy(c->side_stream2);
  }
  for (auto &pair : c->ck_ev)
    for (auto &ev : pair)
      if (ev) (void)hipEventDestroy(ev);
  if (c->side_stream) {
    (void)hipStreamSynchronize(c->side_stream);
    (void)hipStreamDestroy(c->side_stream);
  }
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  for (int k = 0; k < 2; ++k) {
    if (c->ev_done[k]) (void)hipEventDestroy(c->ev_done[k]);
    if (c->ev_copied[k]) (void)hipEventDestroy(c->ev_copied[k]);
  }
  for (msgpu_ctx::HostBuf *h : {&c->h_edges, &c->h_ems, &c->h_orders, &c->h_ids, &c->h_read_len, &c->h_read_first, &c->h_sel_off,
                               &c->h_sel_ems, &c->h_wire[0], &c->h_wire[1]})
    if (h->p) pinned_block_free(h->p);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  delete c;
}

const char *msgpu_last_error(const msgpu_ctx *c) { return c ? c->err : "null context"; }

int msgpu_set_stream(msgpu_ctx *c, void *hip_stream) {
  if (!c) return MSGPU_E_ARG;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return MSGPU_OK;
}

void *msgpu_get_stream(const msgpu_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

// STREAM CONTRACT rule 3 (include/msgpu.h): order the context's stream behind / in front of a stream of the caller's
int msgpu_stream_wait(msgpu_ctx *c, void *hip_stream) {
  if (!c) return MSGPU_E_ARG;
  hipStream_t other = static_cast<hipStream_t>(hip_stream);
  if (other == c->stream) return MSGPU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev_order, other));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_order, 0));
  return MSGPU_OK;
}
int msgpu_stream_release(msgpu_ctx *c, void *hip_stream) {
  if (!c) return MSGPU_E_ARG;
  hipStream_t other = static_cast<hipStream_t>(hip_stream);
  if (other == c->stream) return MSGPU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->ev_order, c->stream));
  HIPCHK(c, hipStreamWaitEvent(other, c->ev_order, 0));
  return MSGPU_OK;
}

int msgpu_set_shard(msgpu_ctx *c, uint32_t shard, uint32_t n_shards) {
  if (!c) return MSGPU_E_ARG;
  if (n_shards == 0 || shard >= n_shards) return fail(c, MSGPU_E_ARG, "shard %u of %u is out of range", shard, n_shards);
  c->shard   = shard;
  c->nshards = n_shards;
  if (c->state > ST_LOADED) c->state = ST_LOADED;
  return MSGPU_OK;
}

int msgpu_set_id_space(msgpu_ctx *c, uint32_t n_reads, uint32_t n_anchors) {
  if (!c) return MSGPU_E_ARG;
  if ((n_reads == 0) != (n_anchors == 0)) return fail(c, MSGPU_E_ARG, "declare both id counts, or 0 and 0 to find them");
  c->decl_V = n_reads;
  c->decl_A = n_anchors;
  return MSGPU_OK;
}

int msgpu_load_rows(msgpu_ctx *c, const msgpu_row *rows, size_t n_rows) {
  if (!c) return MSGPU_E_ARG;
  if (n_rows && !rows) return fail(c, MSGPU_E_ARG, "Unexpected nullptr.");
  if (n_rows > (1ull << 30)) return fail(c, MSGPU_E_ARG, "row table too large (%zu rows, at most 2^30)", n_rows);
  HIPCHK(c, hipSetDevice(c->device));
  c->state = ST_CREATED;
  ENSURE(c, rows_in, (n_rows ? n_rows : 1) * sizeof(msgpu_row));
  if (n_rows)
    HIPCHK(c, hipMemcpyAsync(c->rows_in.p, rows, n_rows * sizeof(msgpu_row), hipMemcpyHostToDevice, c->stream));
  c->d_rows = c->rows_in.as<msgpu_row>();
  c->n_rows = n_rows;
  return build_index(c);
}

int msgpu_load_rows_packed(msgpu_ctx *c, const msgpu_packed_rows *p) {
  if (!c) return MSGPU_E_ARG;
  if (!p || (p->n_rows && (!p->rows || !p->run_start || !p->run_delta || !p->n_runs)) || (p->n_reads && !p->read_len))
    return fail(c, MSGPU_E_ARG, "Unexpected nullptr.");
  if (p->n_rows > (1ull << 30)) return fail(c, MSGPU_E_ARG, "row table too large (%llu rows, at most 2^30)", (unsigned long long)p->n_rows);
  HIPCHK(c, hipSetDevice(c->device));
  c->state = ST_CREATED;
  const size_t n = static_cast<size_t>(p->n_rows);
  // 28 bytes per row + the per-read lengths + the line runs cross the link; the 40-byte rows are made in HBM
  const size_t off_len = (n * sizeof(msgpu_row28) + 255) / 256 * 256, off_rs = off_len + (size_t(p->n_reads) * 4 + 255) / 256 * 256,
               off_rd = off_rs + (size_t(p->n_runs) * 4 + 255) / 256 * 256, total = off_rd + size_t(p->n_runs) * 4 + 256;
  ENSURE(c, rows_pk, total);
  ENSURE(c, rows_in, (n ? n : 1) * sizeof(msgpu_row));
  char *d = c->rows_pk.as<char>();
  if (n) {
    HIPCHK(c, hipMemcpyAsync(d, p->rows, n * sizeof(msgpu_row28), hipMemcpyHostToDevice, c->stream));
    if (p->n_reads) HIPCHK(c, hipMemcpyAsync(d + off_len, p->read_len, size_t(p->n_reads) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + off_rs, p->run_start, size_t(p->n_runs) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + off_rd, p->run_delta, size_t(p->n_runs) * 4, hipMemcpyHostToDevice, c->stream));
    // (a read id beyond the per-read table -- msgpu_pack_rows makes no such table -- gets length 0 and a note in a scratch
    // word; the index build reports ids outside its id space itself)
    launch_expand_rows(c->stream, d, n, reinterpret_cast<const int32_t *>(d + off_len), p->n_reads, reinterpret_cast<const uint32_t *>(d + off_rs),
                       reinterpret_cast<const uint32_t *>(d + off_rd), p->n_runs, c->rows_in.as<msgpu_row>(),
                       reinterpret_cast<uint32_t *>(d + off_rd + size_t(p->n_runs) * 4));
    HIPCHK(c, hipGetLastError());
  }
  c->d_rows = c->rows_in.as<msgpu_row>();
  c->n_rows = n;
  return build_index(c);
}

int msgpu_load_rows_device(msgpu_ctx *c, const void *d_rows, size_t n_rows) {
  if (!c) return MSGPU_E_ARG;
  if (n_rows && !d_rows) return fail(c, MSGPU_E_ARG, "Unexpected nullptr.");
  if (n_rows > (1ull << 30)) return fail(c, MSGPU_E_ARG, "row table too large (%zu rows, at most 2^30)", n_rows);
  HIPCHK(c, hipSetDevice(c->device));
  c->state  = ST_CREATED;
  c->d_rows = static_cast<const msgpu_row *>(d_rows);
  c->n_rows = n_rows;
  return build_index(c);
}

// words of the chain stage's chunk sums (two 64-bit words per chunk of COMPACT_CHUNK edges) for a table of up to n edges
static size_t chunk_words(uint64_t n_edges) { return 2 * static_cast<size_t>(n_edges / COMPACT_CHUNK + 2); }

int msgpu_calculate_edges(msgpu_ctx *c) {
  if (!c) return MSGPU_E_ARG;
  if (c->state < ST_LOADED) return fail(c, MSGPU_E_STATE, "msgpu_calculate_edges before msgpu_load_rows");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t    st = c->stream;
  const uint32_t V  = c->V;
  c->state          = ST_LOADED;
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[2], st));

  ENSURE(c, bound, (size_t(V) + 1) * 4);
  ENSURE(c, cand_off, (size_t(V) + 2) * 8);
  ENSURE(c, lists, (size_t(V) + 1) * (3 * sizeof(CandDesc) + 4));
  ENSURE(c, n_cand, (size_t(V) + 1) * 4);
  ENSURE(c, n_edge, (size_t(V) + 1) * 4);
  ENSURE(c, cand_sums, cand_sums_bytes(V));
  CandDesc *l0 = c->lists.as<CandDesc>(), *l1 = l0 + V + 1, *l2 = l1 + V + 1;
  uint32_t *l3 = reinterpret_cast<uint32_t *>(l2 + V + 1);

  const bool all_reads = c->index_fast && c->nshards == 1 && c->win_lo == 0 && c->win_hi >= V;
  // the index build classified the owner reads already: for this shard, the whole table
  const bool use_prologue = c->index_fast && c->win_lo == 0 && c->win_hi >= V && c->prologue_ok && c->prologue_shard == c->shard &&
                            c->prologue_nshards == c->nshards;
  // What the candidate kernels add to (per-read counts, chunk sums, size histograms, big-edge statistics) is zeroed by
  // k_classify_reads, the launch in front of them: by the index build's prologue, or here.
  const bool zeroed    = use_prologue && c->cand_zeroed; // the index build's prologue did it, and nothing has used it up
  c->cand_zeroed       = false;
  // the scaffold rows each owner read visits: the index build's sort left the sum per read (fast index); a shard or a
  // window of owner reads, or a generically built index (scan view patched after the sort), counts them here
  const uint32_t *bound     = all_reads ? c->visits.as<uint32_t>() : c->bound.as<uint32_t>();
  uint64_t total_bound, own_bound;
  if (use_prologue) { // done with the index build; the numbers came back with its flags
    total_bound = c->prologue_bound; // (the candidate scratch is laid out for ALL reads: the scan is the whole table's)
    own_bound   = c->prologue_own;
    for (int k = 0; k < 4; ++k) c->n_list[k] = c->prologue_lists[k];
    if (!zeroed) { // a second msgpu_calculate_edges on the same index: the lists stand, the sums start over (one launch, rare)
      const CandZero  z         = cand_zero(c, V);
      uint32_t *const zero[4]   = {z.n_cand, z.n_edge, z.scalar_words, nullptr};
      const uint32_t  n_zero[4] = {V + 1, V + 1, z.n_scalar_words, 0};
      uint32_t *const ones[2]   = {nullptr, nullptr};
      const uint32_t  n_ones[2] = {0, 0};
      launch_index_init(st, zero, n_zero, ones, n_ones);
    }
  } else if (c->full_scan_ok && c->index_fast && c->nshards == 1) {
    // a window of the dispatcher on a bin-path index: the scan of ALL reads' visit counts stands since the build (the scratch is
    // laid out for the whole table), so the window needs its reads classified and nothing else -- no per-window count of the
    // visits, no scan of its own (three launches per window fewer)
    if (!c->nlists_clean) HIPCHK(c, hipMemsetAsync(scalar<uint32_t>(c, SC_NLISTS), 0, 16, st));
    c->nlists_clean = false;
    launch_classify_reads(st, c->read_off.as<uint32_t>(), c->read_cnt.as<uint32_t>(), c->visits.as<uint32_t>(),
                          c->cand_off.as<uint64_t>(), V, 0, 1, c->win_lo, c->win_hi, l0, l1, l2, l3,
                          scalar<uint32_t>(c, SC_NLISTS), cand_zero(c, V), scalar<unsigned long long>(c, SC_OWN));
    HIPCHK(c, hipGetLastError());
    if (int rc = read_scalars(c)) return rc;
    total_bound  = c->index_total;
    const uint64_t own_now = *host_scalar<uint64_t>(c, SC_OWN);
    own_bound    = own_now - c->own_accum;
    c->own_accum = own_now;
    for (int k = 0; k < 4; ++k) c->n_list[k] = host_scalar<uint32_t>(c, SC_NLISTS)[k];
  } else {
    // the four list cursors are zero at rest (the scalar block's memset of the index build, k_emit_edges afterwards); a call
    // that ended between k_classify_reads and k_emit_edges left them dirty
    if (!c->nlists_clean) HIPCHK(c, hipMemsetAsync(scalar<uint32_t>(c, SC_NLISTS), 0, 16, st));
    c->full_scan_ok = false; // (cand_off is rewritten below for this subset of the reads)
    if (!all_reads)
      launch_bound(st, c->read_off.as<uint32_t>(), c->read_cnt.as<uint32_t>(), c->vis16.as<uint4>(), V, c->shard,
                   c->nshards, c->win_lo, c->win_hi, c->bound.as<uint32_t>());
    exclusive_scan<uint64_t>(st, bound, V, c->cand_off.as<uint64_t>(), c->scan_tmp.as<uint64_t>(),
                             scalar<uint64_t>(c, SC_TOTAL_A));
    c->nlists_clean = false;
    launch_classify_reads(st, c->read_off.as<uint32_t>(), c->read_cnt.as<uint32_t>(), bound,
                          c->cand_off.as<uint64_t>(), V, c->shard, c->nshards, c->win_lo, c->win_hi, l0, l1, l2, l3,
                          scalar<uint32_t>(c, SC_NLISTS), cand_zero(c, V));
    HIPCHK(c, hipGetLastError());
    if (int rc = read_scalars(c)) return rc; // sizes of the candidate scratch
    total_bound = own_bound = *host_scalar<uint64_t>(c, SC_TOTAL_A);
    for (int k = 0; k < 4; ++k) c->n_list[k] = host_scalar<uint32_t>(c, SC_NLISTS)[k];
  }
  c->total_bound = own_bound; // the scaffold rows this context's owner reads visit

  const size_t tb = total_bound ? total_bound : 1;
  ENSURE(c, cand_j, tb * 4);
  ENSURE(c, cand_t, tb * 4);
  ENSURE(c, scr_v2, tb * 4);
  ENSURE(c, scr_start, tb * 4);

  CandArgs a;
  a.read_off       = c->read_off.as<uint32_t>();
  a.read_cnt       = c->read_cnt.as<uint32_t>();
  a.anchor_off     = c->anchor_off.as<uint32_t>();
  a.by_read        = c->by_read.as<IRow>();
  a.by_anchor      = c->by_anchor.as<IRow>();
  a.vis            = c->vis16.as<uint4>();
  a.cand_off       = c->cand_off.as<uint64_t>();
  a.cand_j         = c->cand_j.as<uint32_t>();
  a.cand_t         = c->cand_t.as<uint32_t>();
  a.edge_scr_v2    = c->scr_v2.as<uint32_t>();
  a.edge_scr_start = c->scr_start.as<uint32_t>();
  a.n_cand         = c->n_cand.as<uint32_t>();
  a.n_edge         = c->n_edge.as<uint32_t>();
  a.th_overlap     = c->p.th_overlap;
  a.big_stats      = scalar<unsigned long long>(c, SC_BIGSTATS);
  // The LDS classes run side by side, each on a stream of its own, and the two heavy ones are launched FIRST: a
  // workgroup of class 2 needs 112 KB of LDS and one of class 1 28 KB, a CU that is full of class-0 workgroups (8 x 16 KB)
  // never has that much free while class 0's grid keeps refilling it -- launched behind class 0 the nine class-2
  // workgroups of BASELINE.json configs[2] waited for the END of class 0 (240 us for a few microseconds of work,
  // profiles/r2_08/timeline_one_step.txt) and class 1 queued behind them.  Launched first they take their CUs while
  // those are empty and class 0 fills the rest.
  static const bool no_fork = getenv("MSGPU_NO_FORK") != nullptr; // measurement switch: the classes one after the other
  const bool fork = (c->n_list[1] || c->n_list[2]) && c->n_list[0] && !no_fork;
  if (fork) {
    // class 1 beside class 0 on a stream of its own, and the handful of class-2 workgroups on the other side stream (idle until
    // the chain stage), launched FIRST: alone in front of class 0 on the main stream they cost 17 us of every step (one or two
    // workgroups at the latency of a whole kernel -- a fifth of a shard-of-eight's candidate stage, profiles/r5_05)
    // (in front of class 0's launch only what must be there: every runtime call is a few microseconds of the host's turn-around)
    HIPCHK(c, hipEventRecord(c->ev_side[0], st));
    if (c->n_list[2]) {
      HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_side[0], 0));
      launch_candidates(c->side_stream, a, 2, l2, c->n_list[2]);
    }
    launch_candidates(st, a, 0, l0, c->n_list[0]);
    if (c->n_list[1]) {
      HIPCHK(c, hipStreamWaitEvent(c->side_stream2, c->ev_side[0], 0));
      launch_candidates(c->side_stream2, a, 1, l1, c->n_list[1]);
      HIPCHK(c, hipEventRecord(c->ev_side2, c->side_stream2));
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_side2, 0));
    }
    if (c->n_list[2]) {
      HIPCHK(c, hipEventRecord(c->ev_side[1], c->side_stream));
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_side[1], 0));
    }
  } else {
    launch_candidates(st, a, 2, l2, c->n_list[2]);
    launch_candidates(st, a, 1, l1, c->n_list[1]);
    launch_candidates(st, a, 0, l0, c->n_list[0]);
  }
  if (c->n_list[3]) {
    ENSURE(c, big_key, tb * 8);
    ENSURE(c, big_t, tb * 4);
    ENSURE(c, big_r2s, tb * 4);
    ENSURE(c, big_pfx, (c->n_rows ? c->n_rows : 1) * 4);
    launch_candidates_big(st, a, l3, c->n_list[3], c->big_key.as<uint64_t>(), c->big_t.as<uint32_t>(),
                          c->big_r2s.as<uint32_t>(), c->big_pfx.as<uint32_t>());
  }
  HIPCHK(c, hipGetLastError());
  // ONE launch closes the stage (k_emit_edges): the scans of the per-read counts, the edge table, the list of big edges, the
  // size-sorted edge list with its class sizes -- and the read-back: its first workgroup publishes the table sizes before the
  // tables are written, so the GPU is busy while the host turns around.  The launch goes into whatever the tables hold from
  // earlier calls; kernel and host compare the same counts with the same capacities: if something does not fit the kernel
  // has written nothing and the host allocates and launches again (the first call of a context always does).
  static const bool sync_path = getenv("MSGPU_SYNC_READBACK") != nullptr;
  c->readback_polled = !sync_path && c->h_scalars_dev;
  auto capacities = [&](uint64_t *cap_edges, uint64_t *cap_big) {
    uint64_t ce = c->edges.room() / sizeof(msgpu_edge);
    if (c->edge_cand.cap / 8 < ce) ce = c->edge_cand.cap / 8;
    if (c->cls_list.cap / 4 < ce) ce = c->cls_list.cap / 4;
    uint64_t cb = c->big_list.cap / 4;
    if (c->big_off.cap / 8 < cb) cb = c->big_off.cap / 8;
    if (c->chain_chunks.cap / 8 < chunk_words(ce)) ce = 0; // (sized with the edge table below: never the limit once both exist)
    *cap_edges = ce;
    *cap_big   = cb;
  };
  auto emit = [&](uint64_t cap_edges, uint64_t cap_big, bool publish) {
    EmitArgs k;
    k.n_edge         = c->n_edge.as<uint32_t>();
    k.n_cand         = c->n_cand.as<uint32_t>();
    k.cand_off       = c->cand_off.as<uint64_t>();
    k.edge_scr_v2    = c->scr_v2.as<uint32_t>();
    k.edge_scr_start = c->scr_start.as<uint32_t>();
    k.V              = V;
    k.hist           = cand_hist(c, V);
    k.chunk_sums     = cand_chunk_sums(c);
    k.n_chunks       = cand_chunks(V);
    k.edges          = c->edges.as<msgpu_edge>();
    k.edge_cand      = c->edge_cand.as<uint64_t>();
    k.list           = c->cls_list.as<uint32_t>();
    k.big_list       = c->big_list.as<uint32_t>();
    k.big_off        = c->big_off.as<uint64_t>();
    k.big_cursor     = scalar<unsigned long long>(c, SC_BIGCUR);
    k.big_stats      = scalar<unsigned long long>(c, SC_BIGSTATS);
    k.cap_edges      = cap_edges;
    k.cap_big        = cap_big;
    k.chain_chunk_sums    = c->chain_chunks.as<unsigned long long>();
    k.n_chain_chunk_words = cap_edges ? static_cast<uint32_t>(chunk_words(cap_edges)) : 0u;
    k.scalars        = c->scalars.as<uint64_t>();
    k.host_scalars   = (publish && c->readback_polled) ? c->h_scalars_dev : nullptr;
    k.slot_ems       = SC_TOTAL_A;
    k.slot_edges     = SC_TOTAL_B;
    k.slot_cls       = SC_CLS;
    k.n_scalars      = SC_COUNT;
    k.seq            = (publish && c->readback_polled) ? ++c->readback_seq : 0;
    k.nlists         = scalar<unsigned long long>(c, SC_NLISTS);
    launch_emit_edges(st, k, publish); // (the first launch of a call: with k_cand_reduce in front)
  };
  uint64_t cap_edges = 0, cap_big = 0;
  capacities(&cap_edges, &cap_big);
  emit(cap_edges, cap_big, true);
  c->nlists_clean = true; // (its first workgroup zeroes the list cursors behind the publication)
  HIPCHK(c, hipGetLastError());
  if (!c->readback_polled) { // the synchronising read-back path: a copy behind the kernel
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, SC_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_readback, st));
  }
  if (int rc = wait_scalars(c)) return rc;
  const uint64_t *tot = host_scalar<uint64_t>(c, SC_TOTAL_A), *big = host_scalar<uint64_t>(c, SC_BIGSTATS);
  c->n_big_edges = big[0];
  c->n_big_ems   = big[1];
  for (int k = 0; k < 4; ++k) c->n_cls[k] = host_scalar<uint32_t>(c, SC_CLS)[k];
  c->n_ems   = tot[0];
  c->n_edges = tot[1];
  c->n_visit = c->total_bound; // the scaffold rows visited = the bound (scaffolds in read-id order: only owned partners)
  if (c->n_edges >= 0xfffffff0ull) return fail(c, MSGPU_E_ARG, "edge table too large (%llu)", (unsigned long long)c->n_edges);
  if (c->n_edges > cap_edges || c->n_big_edges >= cap_big) { // (the kernel's own test, see k_emit_edges)
    ENSURE(c, edges, (c->n_edges ? c->n_edges : 1) * sizeof(msgpu_edge));
    ENSURE(c, edge_cand, (c->n_edges ? c->n_edges : 1) * 8);
    ENSURE(c, cls_list, (c->n_edges + 1) * 4);
    // the edges with more than 64 EdgeMatches (counted by the candidate kernels) are listed as they are emitted
    ENSURE(c, big_list, (c->n_big_edges + 1) * 4);
    ENSURE(c, big_off, (c->n_big_edges + 1) * 8);
    ENSURE(c, chain_chunks, chunk_words(c->edges.room() / sizeof(msgpu_edge)) * 8);
    capacities(&cap_edges, &cap_big);
    emit(cap_edges, cap_big, false);
  }
  // the chain stage's chunk sums were zeroed by k_emit_edges: msgpu_chaining_and_overlaps starts with its kernels
  c->chain_zeroed = true;
  HIPCHK(c, hipGetLastError());
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[3], st));
  c->have_cand_t = c->stage_events;
  c->n_orders    = 0;
  c->n_ids       = 0;
  c->state       = ST_EDGES;
  return MSGPU_OK;
}

int msgpu_chaining_and_overlaps(msgpu_ctx *c) {
  if (!c) return MSGPU_E_ARG;
  if (c->state < ST_EDGES || c->state == ST_RESULT)
    return fail(c, MSGPU_E_STATE, "msgpu_chaining_and_overlaps before msgpu_calculate_edges");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t    st = c->stream;
  const uint64_t E = c->n_edges, M = c->n_ems;
  for (int k = 0; k < 2; ++k)
    if (!c->ck_ev[c->ck_head][k]) HIPCHK(c, hipEventCreate(&c->ck_ev[c->ck_head][k]));
  const hipEvent_t ck_begin = c->ck_ev[c->ck_head][0], ck_end = c->ck_ev[c->ck_head][1];
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[4], st));

  ENSURE(c, ems, (M ? M : 1) * sizeof(msgpu_edgematch));
  ENSURE(c, order_scr, (M ? M : 1) * sizeof(msgpu_order));
  ENSURE(c, ids_scr, (M ? M : 1) * 4);
  ENSURE(c, edge_norders, (E + 4) * 4);
  ENSURE(c, edge_nids, (E + 4) * 4);
  ENSURE(c, chain_chunks, chunk_words(E) * 8);

  ChainArgs a;
  a.edges        = c->edges.as<msgpu_edge>();
  a.edge_cand    = c->edge_cand.as<uint64_t>();
  a.n_edges      = E;
  a.cand_j       = c->cand_j.as<uint32_t>();
  a.cand_t       = c->cand_t.as<uint32_t>();
  a.read_off     = c->read_off.as<uint32_t>();
  a.read_cnt     = c->read_cnt.as<uint32_t>();
  a.read_len     = c->read_len.as<int32_t>();
  a.by_read      = c->by_read.as<IRow>();
  a.by_anchor    = c->by_anchor.as<IRow>();
  a.ems          = c->ems.as<msgpu_edgematch>();
  a.order_scr    = c->order_scr.as<msgpu_order>();
  a.ids_scr      = c->ids_scr.as<uint32_t>();
  a.edge_norders = c->edge_norders.as<uint32_t>();
  a.edge_nids    = c->edge_nids.as<uint32_t>();
  a.err          = scalar<uint32_t>(c, SC_ERR);
  if (!c->pair_tab.p) {
    ENSURE(c, pair_tab, 4 * PAIR_TAB_STRIDE * sizeof(uint32_t) + PAIR_TAB_STRIDE * sizeof(uint2) + 3 * PAIR_TAB_STRIDE * 8);
    launch_fill_pair_tab(st, c->pair_tab.as<uint32_t>());
  }
  a.pair_tab     = c->pair_tab.as<uint32_t>();
  a.pair_tab64   = c->pair_tab.as<uint32_t>() + 4 * PAIR_TAB_STRIDE;
  a.pair_tab_sub = c->pair_tab.as<uint32_t>() + 4 * PAIR_TAB_STRIDE + 2 * PAIR_TAB_STRIDE;
  if (!c->chain_zeroed) // (msgpu_calculate_edges' size sort left the chunk sums zeroed; a second chaining pass, or a run without the sort, zeroes them here)
    HIPCHK(c, hipMemsetAsync(c->chain_chunks.p, 0, chunk_words(E) * 8, st));
  c->chain_zeroed = false;
  a.chunk_sums   = c->chain_chunks.as<unsigned long long>();
  a.fast_path    = c->fast_path ? 1 : 0;
  a.wiggle       = static_cast<double>(c->p.wiggle_room);
  a.ratio_pct    = c->p.ratio_pct;
  a.alt_frac     = c->p.alt_frac;
  a.out_edge_base = static_cast<uint32_t>(c->base_edges);

  // Edges with more than 64 EdgeMatches (counted by the candidate kernels, so the host already knows how many there
  // are and how much scratch they need) run in k_chain_big on the side stream, concurrently with k_chain.
  const uint32_t n_big = static_cast<uint32_t>(c->n_big_edges);
  if (n_big) {
    ENSURE(c, big_elems, (c->n_big_ems ? c->n_big_ems : 1) * big_elem_bytes());
    ENSURE(c, big_paths, (c->n_big_ems ? c->n_big_ems : 1) * 2 * big_path_bytes());
  }
  // Two events mark "the candidate stage is done" on the main stream: the first releases k_chain_big on its side stream, the
  // second opens the chain kernels' timing window and releases the sub-wavefront classes on theirs (a third one for those cost the
  // main stream one more packet in front of k_chain).  k_chain_big keeps an event of its own, recorded FIRST: released by the
  // same event as the others it started behind k_chain, its thousand long-lived wavefronts then sat beside the others for the whole
  // stage instead of its first fifth, and the stage took 30 us longer (gpurun_out/r5_45 against r5_38).
  if (n_big) HIPCHK(c, hipEventRecord(c->ev_side[0], st)); // (the list and the scratch offsets: k_emit_edges)
  auto launch_big = [&]() -> int {
    if (!n_big) return MSGPU_OK;
    HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_side[0], 0));
    launch_chain_big(c->side_stream, a, c->big_list.as<uint32_t>(), c->big_off.as<uint64_t>(), n_big,
                     c->big_elems.p, c->big_paths.p);
    HIPCHK(c, hipEventRecord(c->ev_side[1], c->side_stream));
    return MSGPU_OK;
  };
  {
    std::lock_guard<std::mutex> g(c->gate_m);
    ++c->chain_launches;
  }
  c->gate_cv.notify_all();
  if (int rc = launch_big()) return rc; // (first: its few long-lived wavefronts get their registers before k_chain fills the device)
  HIPCHK(c, hipEventRecord(ck_begin, st));
  if (c->sub_wave && E) {
    // the size-sorted edge list and the class sizes are there since msgpu_calculate_edges
    const uint32_t *list = c->cls_list.as<uint32_t>();
    const uint32_t *l64 = list, *l32 = l64 + c->n_cls[2], *l16 = l32 + c->n_cls[1], *l8 = l16 + c->n_cls[0]; // sizes descending
    // The three sub-wavefront classes go out as ONE launch (k_chain_sub_all: its workgroups take the 32-, 16- and 8-wide class
    // in turn) on the side stream the candidate stage used, beside k_chain: a kernel of a few thousand wavefronts lasts as long
    // as ONE of its wavefronts (45 us each for the 16- and 8-wide classes of a shard of eight, one after the other behind
    // k_chain and k_chain_sub<32>: a third of that shard's chain stage, profiles/r5_05); side by side they fill what the long
    // class leaves.  On the whole job the kernels' work is the same either way.
    const bool serial = getenv("MSGPU_CHAIN_SERIAL") != nullptr; // A/B switch (read per call: a test flips it): the four classes one after the other, a launch each
    const bool any_sub = c->n_cls[0] || c->n_cls[1] || c->n_cls[3];
    const bool beside  = !serial && any_sub && c->n_cls[2];
    if (beside) HIPCHK(c, hipStreamWaitEvent(c->side_stream2, ck_begin, 0));
    launch_chain(st, a, l64, c->n_cls[2]); // the long ones first: the short classes fill the tail
    if (serial) {
      launch_chain_sub(st, a, 32, l32, c->n_cls[1]);
      launch_chain_sub(st, a, 16, l16, c->n_cls[0]);
      launch_chain_sub(st, a, 8, l8, c->n_cls[3]);
    } else if (any_sub) {
      launch_chain_sub_all(beside ? c->side_stream2 : st, a, l32, c->n_cls[1], l16, c->n_cls[0], l8, c->n_cls[3]);
    }
    if (beside) {
      HIPCHK(c, hipEventRecord(c->ev_side2, c->side_stream2));
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_side2, 0));
    }
  } else {
    launch_chain(st, a, nullptr, 0);
  }
  HIPCHK(c, hipEventRecord(ck_end, st));
  c->ck_head = (c->ck_head + 1) % msgpu_ctx::CK_RING;
  if (c->ck_count < msgpu_ctx::CK_RING) ++c->ck_count;
  HIPCHK(c, hipGetLastError());
  if (n_big) HIPCHK(c, hipStreamWaitEvent(st, c->ev_side[1], 0));

  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[7], st));
  // ONE launch closes the stage: k_compact sums the chunk sums the chain kernels left (the scan), its first workgroup writes the
  // table sizes into the scalar block and publishes it to the host, and the move into the dense tables follows in the same
  // kernel -- into what the two tables hold from earlier calls (see msgpu_calculate_edges): the GPU is busy while the host
  // turns around; if the tables turn out too small the kernel has written nothing and is launched again behind the allocation.
  static const bool sync_path = getenv("MSGPU_SYNC_READBACK") != nullptr;
  c->readback_polled = !sync_path && c->h_scalars_dev;
  auto compact = [&](bool publish) {
    CompactArgs k;
    k.edges        = c->edges.as<msgpu_edge>();
    k.n_edges      = E;
    k.edge_norders = c->edge_norders.as<uint32_t>();
    k.edge_nids    = c->edge_nids.as<uint32_t>();
    k.chunk_sums   = c->chain_chunks.as<unsigned long long>();
    k.n_chunks     = static_cast<uint32_t>((E + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
    k.scalars      = c->scalars.as<uint64_t>();
    k.host_scalars = (publish && c->readback_polled) ? c->h_scalars_dev : nullptr;
    k.slot_orders  = SC_TOTAL_A;
    k.slot_ids     = SC_TOTAL_B;
    k.slot_fast    = SC_TOTAL_C;
    k.n_scalars    = SC_COUNT;
    k.seq          = (publish && c->readback_polled) ? ++c->readback_seq : 0;
    k.order_scr    = c->order_scr.as<msgpu_order>();
    k.ids_scr      = c->ids_scr.as<uint32_t>();
    k.orders       = c->orders.as<msgpu_order>();
    k.ids          = c->ids.as<uint32_t>();
    k.out_em_base    = c->base_ems;
    k.out_order_base = c->base_orders; // (a batched run: the earlier batches' counts, known to the caller)
    k.out_ids_base   = c->base_ids;
    k.out_edge_base  = static_cast<uint32_t>(c->base_edges);
    k.cap_orders     = c->orders.room() / sizeof(msgpu_order);
    k.cap_ids        = c->ids.room() / 4;
    launch_compact(st, k);
  };
  const uint64_t cap_orders = c->orders.room() / sizeof(msgpu_order), cap_ids = c->ids.room() / 4;
  compact(true);
  HIPCHK(c, hipGetLastError());
  if (!c->readback_polled) { // the synchronising read-back path (MSGPU_SYNC_READBACK=1, or no mapped mirror): a copy behind the kernel
    HIPCHK(c, hipMemcpyAsync(c->h_scalars, c->scalars.p, SC_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->ev_readback, st));
  }
  if (int rc = wait_scalars(c)) return rc;
  c->n_edges_fast     = *host_scalar<uint64_t>(c, SC_TOTAL_C);
  const uint64_t *tot = host_scalar<uint64_t>(c, SC_TOTAL_A);
  c->n_orders = tot[0];
  c->n_ids    = tot[1];
  if (c->n_orders > cap_orders || c->n_ids > cap_ids) { // (the kernel's own test)
    ENSURE(c, orders, (c->n_orders ? c->n_orders : 1) * sizeof(msgpu_order));
    ENSURE(c, ids, (c->n_ids ? c->n_ids : 1) * 4);
    compact(false);
  }
  HIPCHK(c, hipGetLastError());
  if (c->stage_events) HIPCHK(c, hipEventRecord(c->ev[8], st));
  c->have_chain_t = true;
  c->have_stage_t = c->stage_events;
  c->state        = ST_CHAINED;
  return MSGPU_OK;
}

int msgpu_set_stage_events(msgpu_ctx *c, int on) {
  if (!c) return MSGPU_E_ARG;
  c->stage_events = on != 0;
  return MSGPU_OK;
}

int msgpu_get_counts(msgpu_ctx *c, msgpu_counts *out) {
  if (!c || !out) return MSGPU_E_ARG;
  memset(out, 0, sizeof(*out));
  out->n_rows_in       = c->n_rows;
  out->n_rows_alive    = c->n_alive;
  out->n_reads         = c->V;
  out->n_anchors       = c->A;
  out->n_edges         = c->state >= ST_EDGES ? c->n_edges : 0;
  out->n_ems           = c->state >= ST_EDGES ? c->n_ems : 0;
  out->n_orders        = c->state >= ST_CHAINED ? c->n_orders : 0;
  out->n_ids           = c->state >= ST_CHAINED ? c->n_ids : 0;
  out->n_pairs_scanned = c->state >= ST_EDGES ? c->n_visit : 0;
  out->n_edges_fastpath = c->state >= ST_CHAINED ? c->n_edges_fast : 0;
  out->n_lost_publications = c->lost_publications;
  out->index_path          = c->index_path;
  return MSGPU_OK;
}

int msgpu_get_timings(msgpu_ctx *c, msgpu_timings *out) {
  if (!c || !out) return MSGPU_E_ARG;
  memset(out, 0, sizeof(*out));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->have_index_t) HIPCHK(c, hipEventElapsedTime(&out->index_ms, c->ev[0], c->ev[1]));
  if (c->have_cand_t) HIPCHK(c, hipEventElapsedTime(&out->candidates_ms, c->ev[2], c->ev[3]));
  if (c->have_chain_t) {
    if (c->have_stage_t) {
      HIPCHK(c, hipEventElapsedTime(&out->chain_ms, c->ev[4], c->ev[7]));
      HIPCHK(c, hipEventElapsedTime(&out->compact_ms, c->ev[7], c->ev[8]));
    }
    // mean over the msgpu_chaining_and_overlaps calls since the last msgpu_get_timings (at most CK_RING of them)
    double   sum = 0;
    uint32_t n   = 0;
    for (uint32_t i = 0; i < c->ck_count; ++i) {
      const uint32_t slot = (c->ck_head + msgpu_ctx::CK_RING - 1 - i) % msgpu_ctx::CK_RING;
      float          ms   = 0;
      HIPCHK(c, hipEventElapsedTime(&ms, c->ck_ev[slot][0], c->ck_ev[slot][1]));
      sum += ms;
      ++n;
    }
    out->chain_kernel_ms       = n ? static_cast<float>(sum / n) : 0.f;
    out->chain_kernel_launches = n;
    c->ck_count                = 0;
  }
  return MSGPU_OK;
}

static int copy_tables(msgpu_ctx *c, void *edges, void *ems, void *orders, void *ids, hipMemcpyKind kind) {
  if (!c) return MSGPU_E_ARG;
  if (c->state < ST_EDGES) return fail(c, MSGPU_E_STATE, "no tables yet");
  if ((ems || orders || ids) && c->state < ST_CHAINED)
    return fail(c, MSGPU_E_STATE, "EdgeMatch/order tables exist only after msgpu_chaining_and_overlaps");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (edges && c->n_edges) HIPCHK(c, hipMemcpyAsync(edges, c->edges.at(), c->n_edges * sizeof(msgpu_edge), kind, st));
  if (ems && c->n_ems) HIPCHK(c, hipMemcpyAsync(ems, c->ems.at(), c->n_ems * sizeof(msgpu_edgematch), kind, st));
  if (orders && c->n_orders) HIPCHK(c, hipMemcpyAsync(orders, c->orders.at(), c->n_orders * sizeof(msgpu_order), kind, st));
  if (ids && c->n_ids) HIPCHK(c, hipMemcpyAsync(ids, c->ids.at(), c->n_ids * 4, kind, st));
  if (kind == hipMemcpyDeviceToHost) HIPCHK(c, hipStreamSynchronize(st));
  return MSGPU_OK;
}

int msgpu_copy_tables(msgpu_ctx *c, msgpu_edge *edges, msgpu_edgematch *ems, msgpu_order *orders, uint32_t *ids) {
  return copy_tables(c, edges, ems, orders, ids, hipMemcpyDeviceToHost);
}
int msgpu_copy_tables_device(msgpu_ctx *c, void *d_edges, void *d_ems, void *d_orders, void *d_ids) {
  return copy_tables(c, d_edges, d_ems, d_orders, d_ids, hipMemcpyDeviceToDevice);
}

int msgpu_copy_reads(msgpu_ctx *c, int32_t *read_len, uint32_t *read_first_line) {
  if (!c) return MSGPU_E_ARG;
  if (c->state < ST_LOADED) return fail(c, MSGPU_E_STATE, "no rows loaded");
  HIPCHK(c, hipSetDevice(c->device));
  if (read_len && c->V)
    HIPCHK(c, hipMemcpyAsync(read_len, c->read_len.p, size_t(c->V) * 4, hipMemcpyDeviceToHost, c->stream));
  if (read_first_line && c->V)
    HIPCHK(c, hipMemcpyAsync(read_first_line, c->read_first.p, size_t(c->V) * 4, hipMemcpyDeviceToHost, c->stream));
  return host_sync(c, c->stream);
}

// findContractionEdges (src/main.cpp:183-190, 416-463) + sanityCheck (sc.cpp:29-90) over a resident edge/order table
int msgpu_find_contraction_edges(msgpu_ctx *c, const void *d_edges, uint64_t n_edges, const void *d_orders,
                                 uint64_t n_orders, uint32_t n_reads, int64_t *contraction_order) {
  if (!c) return MSGPU_E_ARG;
  if ((d_edges == nullptr) != (d_orders == nullptr)) return fail(c, MSGPU_E_ARG, "pass both tables or neither");
  if (!d_edges) {
    if (c->state < ST_CHAINED) return fail(c, MSGPU_E_STATE, "msgpu_find_contraction_edges before msgpu_chaining_and_overlaps");
    if (c->nshards > 1)
      return fail(c, MSGPU_E_STATE, "a shard holds only its own edges: pass the merged tables (msgpu_merge_gathered)");
    d_edges  = c->edges.at();
    d_orders = c->orders.at();
    n_edges  = c->n_edges;
    n_orders = c->n_orders;
    n_reads  = c->V;
  }
  if (n_edges && !contraction_order) return MSGPU_E_ARG;
  if (n_edges >= 0x7ffffff0ull || n_orders >= 0xfffffff0ull) return fail(c, MSGPU_E_ARG, "tables too large");
  if (!n_edges) return MSGPU_OK;
  HIPCHK(c, hipSetDevice(c->device));
  const auto *edges  = static_cast<const msgpu_edge *>(d_edges);
  const auto *orders = static_cast<const msgpu_order *>(d_orders);
  const size_t V = n_reads;
  ENSURE(c, g_deg, (2 * V + 2) * 4);                 // degree | cursor
  ENSURE(c, g_off, (V + 2) * 8);
  ENSURE(c, g_adj, 2 * n_edges * 4);
  ENSURE(c, g_cand, (n_orders + 2) * 4);            // [0] = count, then the list
  ENSURE(c, g_sane, n_orders + 1);
  ENSURE(c, g_out, n_edges * 8);
  ENSURE(c, scan_tmp, (size_t(scan_blocks(V + 1)) + 1) * 8);
  ENSURE(c, scalars, SC_COUNT * 8);
  uint32_t *deg = c->g_deg.as<uint32_t>(), *cursor = deg + V + 1;
  uint64_t *off = c->g_off.as<uint64_t>();
  uint32_t *cand = c->g_cand.as<uint32_t>();
  HIPCHK(c, hipMemsetAsync(deg, 0, (2 * V + 2) * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(cand, 0, 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->g_sane.p, 0, n_orders + 1, c->stream));
  launch_degree(c->stream, edges, n_edges, deg);
  exclusive_scan<uint64_t>(c->stream, deg, V + 1, off, c->scan_tmp.as<uint64_t>(), scalar<uint64_t>(c, SC_TOTAL_A));
  launch_fill_adj(c->stream, edges, n_edges, off, cursor, c->g_adj.as<uint32_t>());
  launch_mark_contained(c->stream, orders, n_orders, cand + 1, cand);
  launch_check_contraction(c->stream, edges, n_edges, orders, off, c->g_adj.as<uint32_t>(), cand + 1, cand, n_orders,
                           static_cast<double>(c->p.wiggle_room), c->g_sane.as<uint8_t>());
  launch_pick_contraction(c->stream, edges, n_edges, c->g_sane.as<uint8_t>(), c->g_out.as<int64_t>());
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(contraction_order, c->g_out.p, n_edges * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MSGPU_OK;
}

int msgpu_merge_gathered(msgpu_ctx *c, const void *d_gathered, uint32_t world, const uint64_t *counts,
                         uint64_t slab_bytes, uint64_t off_edges, uint64_t off_orders, uint64_t off_ids, void *d_edges,
                         void *d_orders, void *d_ids) {
  return msgpu_merge_gathered_ex(c, d_gathered, world, counts, slab_bytes, off_edges, off_orders, off_ids, nullptr, d_edges,
                                 d_orders, d_ids, nullptr);
}

int msgpu_merge_gathered_ex(msgpu_ctx *c, const void *d_gathered, uint32_t world, const uint64_t *counts,
                            uint64_t slab_bytes, uint64_t off_edges, uint64_t off_orders, uint64_t off_ids,
                            const uint32_t *id_base, void *d_edges, void *d_orders, void *d_ids, void *hip_stream) {
  if (!c) return MSGPU_E_ARG;
  if (!d_gathered || !counts || world == 0 || world > MAX_WORLD) return fail(c, MSGPU_E_ARG, "bad merge arguments");
  HIPCHK(c, hipSetDevice(c->device));
  MergeArgs a;
  a.gathered   = static_cast<const uint8_t *>(d_gathered);
  a.slab_bytes = slab_bytes;
  a.off_edges  = off_edges;
  a.off_orders = off_orders;
  a.off_ids    = off_ids;
  a.world      = world;
  a.base[0]    = MergeBase{0, 0, 0, 0, 0};
  for (uint32_t r = 0; r < world; ++r) {
    a.base[r + 1].edges  = a.base[r].edges + counts[3 * r + 0];
    a.base[r + 1].orders = a.base[r].orders + counts[3 * r + 1];
    a.base[r + 1].ids    = a.base[r].ids + counts[3 * r + 2];
    a.base[r].read_id    = id_base ? id_base[2 * r + 0] : 0;
    a.base[r].anchor_id  = id_base ? id_base[2 * r + 1] : 0;
  }
  a.base[world].read_id = a.base[world].anchor_id = 0;
  if (a.base[world].edges >= 0xfffffff0ull) return fail(c, MSGPU_E_ARG, "merged edge table too large");
  a.edges  = static_cast<msgpu_edge *>(d_edges);
  a.orders = static_cast<msgpu_order *>(d_orders);
  a.ids    = static_cast<uint32_t *>(d_ids);
  launch_merge_gathered(hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream, a);
  HIPCHK(c, hipGetLastError());
  return MSGPU_OK;
}

// ---- the exchange's wire form -------------------------------------------------------------------------------------------

uint64_t msgpu_wire_edges_bytes(uint64_t n_edges) { return wire_edges_bytes(n_edges); }
uint64_t msgpu_wire_orders_bytes(uint64_t n_orders) { return wire_orders_bytes(n_orders); }
uint64_t msgpu_wire_ids_bytes(uint64_t n_ids, uint32_t id_bytes) { return wire_ids_bytes(n_ids, id_bytes); }

int msgpu_pack_wire(msgpu_ctx *c, void *d_wire_edges, void *d_wire_orders, void *d_ids, uint32_t id_bytes) {
  if (!c) return MSGPU_E_ARG;
  if (id_bytes != 3 && id_bytes != 4) return fail(c, MSGPU_E_ARG, "msgpu_pack_wire: id_bytes is 3 or 4");
  if (id_bytes == 3 && c->A > (1u << 24))
    return fail(c, MSGPU_E_ARG, "msgpu_pack_wire: anchor ids beyond 24 bits need id_bytes = 4");
  if (c->state < ST_CHAINED) return fail(c, MSGPU_E_STATE, "msgpu_pack_wire before msgpu_chaining_and_overlaps");
  if (!d_wire_edges || !d_wire_orders || !d_ids) return fail(c, MSGPU_E_ARG, "msgpu_pack_wire: a null block");
  if (c->n_ems > 0xffffffffull || c->n_orders > 0xffffffffull || c->n_ids > 0xffffffffull)
    return fail(c, MSGPU_E_ARG, "tables beyond the wire form's 32-bit offsets: exchange them whole "
                                "(msgpu_copy_tables_device + msgpu_merge_gathered)");
  if ((reinterpret_cast<uintptr_t>(d_wire_edges) & 3) || (reinterpret_cast<uintptr_t>(d_wire_orders) & 7) ||
      (reinterpret_cast<uintptr_t>(d_ids) & 3))
    return fail(c, MSGPU_E_ARG, "msgpu_pack_wire: the edge and id blocks need 4-byte, the order block 8-byte alignment");
  HIPCHK(c, hipSetDevice(c->device));
  PackWireArgs a;
  a.edges    = static_cast<const msgpu_edge *>(c->edges.at());
  a.orders   = static_cast<const msgpu_order *>(c->orders.at());
  a.n_edges  = c->n_edges;
  a.n_orders = c->n_orders;
  a.w_edges  = static_cast<uint8_t *>(d_wire_edges);
  a.w_orders = static_cast<uint8_t *>(d_wire_orders);
  a.ids      = id_bytes == 3 ? static_cast<const uint32_t *>(c->ids.at()) : nullptr;
  a.n_ids    = c->n_ids;
  a.w_ids    = static_cast<uint32_t *>(d_ids);
  launch_pack_wire(c->stream, a);
  HIPCHK(c, hipGetLastError());
  if (id_bytes == 4 && c->n_ids)
    HIPCHK(c, hipMemcpyAsync(d_ids, c->ids.at(), c->n_ids * 4, hipMemcpyDeviceToDevice, c->stream));
  return MSGPU_OK;
}

int msgpu_unpack_wire_host(const void *wire_edges, const void *wire_orders, const void *wire_ids, uint32_t id_bytes, uint64_t n_edges,
                           uint64_t n_orders, uint64_t n_ids, const uint64_t *base, msgpu_edge *edges, msgpu_order *orders,
                           uint32_t *ids, uint32_t threads, uint32_t tables) {
  if (!tables) tables = 7;
  if ((id_bytes != 3 && id_bytes != 4) || tables > 7 || !wire_edges || ((tables & 2u) && !wire_orders) || ((tables & 4u) && n_ids && !wire_ids) ||
      ((tables & 1u) && n_edges && !edges) || ((tables & 2u) && n_orders && !orders) || ((tables & 4u) && n_ids && !ids))
    return MSGPU_E_ARG;
  if ((reinterpret_cast<uintptr_t>(wire_edges) & 3) || (reinterpret_cast<uintptr_t>(wire_orders) & 7) ||
      (reinterpret_cast<uintptr_t>(wire_ids) & 3))
    return MSGPU_E_ARG;
  try {
    unpack_wire_host(static_cast<const uint8_t *>(wire_edges), static_cast<const uint8_t *>(wire_orders),
                     static_cast<const uint32_t *>(wire_ids), id_bytes, n_edges, n_orders, n_ids, base ? base[0] : 0, base ? base[1] : 0,
                     base ? base[2] : 0, base ? base[3] : 0, edges, orders, ids, threads ? threads : 16, tables);
  } catch (...) {
    return MSGPU_E_NOMEM;
  }
  return MSGPU_OK;
}

int msgpu_merge_wire(msgpu_ctx *c, const void *d_gathered, uint32_t world, const uint64_t *counts, uint64_t slab_bytes,
                     uint64_t off_edges, uint64_t off_orders, uint64_t off_ids, uint32_t id_bytes, const uint32_t *id_base,
                     void *d_edges, void *d_orders, void *d_ids, void *hip_stream) {
  if (!c) return MSGPU_E_ARG;
  if (id_bytes != 3 && id_bytes != 4) return fail(c, MSGPU_E_ARG, "msgpu_merge_wire: id_bytes is 3 or 4");
  if (!d_gathered || !counts || world == 0 || world > MAX_WORLD) return fail(c, MSGPU_E_ARG, "bad merge arguments");
  if ((reinterpret_cast<uintptr_t>(d_gathered) & 7) || (slab_bytes & 7) || (off_edges & 3) || (off_orders & 7) || (off_ids & 3))
    return fail(c, MSGPU_E_ARG, "msgpu_merge_wire: slabs and order blocks need 8-byte, edge and id blocks 4-byte alignment");
  if ((reinterpret_cast<uintptr_t>(d_edges) & 15) || (reinterpret_cast<uintptr_t>(d_orders) & 15))
    return fail(c, MSGPU_E_ARG, "msgpu_merge_wire: the merged edge and order tables need 16-byte alignment (whole-line stores)");
  HIPCHK(c, hipSetDevice(c->device));
  MergeArgs a;
  a.gathered   = static_cast<const uint8_t *>(d_gathered);
  a.slab_bytes = slab_bytes;
  a.off_edges  = off_edges;
  a.off_orders = off_orders;
  a.off_ids    = off_ids;
  a.world      = world;
  a.base[0]    = MergeBase{0, 0, 0, 0, 0};
  for (uint32_t r = 0; r < world; ++r) {
    if (counts[3 * r + 1] > 0xffffffffull || counts[3 * r + 2] > 0xffffffffull)
      return fail(c, MSGPU_E_ARG, "a rank's tables are beyond the wire form's 32-bit offsets");
    a.base[r + 1].edges  = a.base[r].edges + counts[3 * r + 0];
    a.base[r + 1].orders = a.base[r].orders + counts[3 * r + 1];
    a.base[r + 1].ids    = a.base[r].ids + counts[3 * r + 2];
    a.base[r].read_id    = id_base ? id_base[2 * r + 0] : 0;
    a.base[r].anchor_id  = id_base ? id_base[2 * r + 1] : 0;
  }
  a.base[world].read_id = a.base[world].anchor_id = 0;
  if (a.base[world].edges >= 0xfffffff0ull) return fail(c, MSGPU_E_ARG, "merged edge table too large");
  a.edges  = static_cast<msgpu_edge *>(d_edges);
  a.orders = static_cast<msgpu_order *>(d_orders);
  a.ids    = static_cast<uint32_t *>(d_ids);
  launch_merge_wire(hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream, a, id_bytes == 3);
  HIPCHK(c, hipGetLastError());
  return MSGPU_OK;
}

// ---- the ThreadPool replacement: the whole overlap path as owner-read batches on two HIP streams -------------------------
//
// The reference fans one Job per PAF line / anchor / edge over ThreadPool workers and blocks in WaitGroup::wait() at the
// end of each phase (libms/src/threading/ThreadPool.cpp:38-129, WaitGroup.cpp:36-72, src/main.cpp:143-178).  Here the
// unit of dispatch is a BATCH of owner reads (the edges whose first vertex lies in a window of read ids): its candidate
// scan, chaining and compaction are a few dozen kernel launches on the compute stream, and while batch k+1 computes,
// batch k's four tables travel to pinned host memory on the copy stream (two sets of output tables in HBM, guarded by
// events).  The phase barrier -- all batches done -- is the final wait on the copy stream.

namespace {

// pinned host block of at least `need` bytes that keeps its first `valid` bytes (the copy stream must be idle when it
// moves); `hint` = expected final size
int ensure_host(msgpu_ctx *c, msgpu_ctx::HostBuf &h, size_t need, size_t valid, size_t hint) {
  if (need <= h.cap) return MSGPU_OK;
  size_t want = need + need / 4 + 4096;
  if (hint > want) want = hint;
  void *np = pinned_block_alloc(want);
  if (!np) return fail(c, MSGPU_E_NOMEM, "page-locked host table of %zu bytes", want);
  if (h.p) {
    if (valid) {
      hipError_t e = hipStreamSynchronize(c->copy_stream);
      if (e != hipSuccess) {
        pinned_block_free(np);
        HIPCHK(c, e);
      }
      // (on the host threads: one thread moves 100 MB in 10 ms, and the GPU waits for the result tables meanwhile)
      const size_t piece = size_t(4) << 20, n_pieces = (valid + piece - 1) / piece;
      char        *dst = static_cast<char *>(np);
      const char  *src = static_cast<const char *>(h.p);
      auto         move = [&](size_t k) { memcpy(dst + k * piece, src + k * piece, std::min(piece, valid - k * piece)); };
      try {
        msgpu::HostPool::get().run(n_pieces > 1 ? 16 : 1, n_pieces, move);
      } catch (...) { // (no thread to be had: copying does not throw)
        memcpy(np, h.p, valid);
      }
    }
    pinned_block_free(h.p);
  }
  h.p   = np;
  h.cap = want;
  return MSGPU_OK;
}

// The receiving end of a dispatcher window that left in wire form: a thread that waits for the window's copy and turns the
// blocks into records (unpack_wire_host on the host pool) while the stream thread goes on with the next window.
struct WindowUnpacker {
  struct Job {
    hipEvent_t      copied;
    const uint8_t  *w_edges, *w_orders;
    const uint32_t *w_ids; // null: the ids travelled as they are, straight into the table
    uint64_t        n_edges, n_orders, n_ids, base_edges, base_ems, base_orders, base_ids;
    msgpu_edge     *edges;
    msgpu_order    *orders;
    uint32_t       *ids;
    unsigned        tables = 7; // 1 edges, 2 orders, 4 ids: which of them this job turns into records
  };
  int                     device = 0;
  bool                    debug = false; // MSGPU_BATCH_DEBUG: when every window's copy was seen and when its records stood
  std::chrono::steady_clock::time_point t0;
  std::mutex              m;
  std::condition_variable cv;
  std::deque<Job>         q;
  uint64_t                pushed = 0, done = 0;
  bool                    stop = false;
  hipError_t              err = hipSuccess;
  std::thread             th;
  bool start(int dev) {
    device = dev;
    try {
      th = std::thread([this] { loop(); });
    } catch (...) {
      return false;
    }
    return true;
  }
  void loop() {
    (void)hipSetDevice(device);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        j = q.front();
        q.pop_front();
      }
      hipError_t e = hipEventSynchronize(j.copied);
      const auto t_copied = std::chrono::steady_clock::now();
      if (e == hipSuccess) {
        try {
          unpack_wire_host(j.w_edges, j.w_orders, j.w_ids, j.w_ids ? 3 : 4, j.n_edges, j.n_orders, j.w_ids ? j.n_ids : 0, j.base_edges,
                           j.base_ems, j.base_orders, j.base_ids, j.edges, j.orders, j.ids, 16, j.tables);
        } catch (...) {
          e = hipErrorOutOfMemory;
        }
      }
      if (debug)
        fprintf(stderr, "[batched] unpacker: tables %u of the copy of %llu edges / %llu orders / %llu ids seen at %.2f ms, records at %.2f ms\n",
                j.tables, (unsigned long long)j.n_edges, (unsigned long long)j.n_orders, (unsigned long long)j.n_ids,
                std::chrono::duration<double, std::milli>(t_copied - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      std::lock_guard<std::mutex> g(m);
      if (e != hipSuccess && err == hipSuccess) err = e;
      ++done;
      cv.notify_all();
    }
  }
  void push(const Job &j) {
    std::lock_guard<std::mutex> g(m);
    q.push_back(j);
    ++pushed;
    cv.notify_all();
  }
  void wait_done(uint64_t upto) { // the first `upto` windows pushed are records in the host tables
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done >= upto; });
  }
  void drain() { wait_done(pushed); }
  hipError_t finish() {
    if (th.joinable()) {
      {
        std::lock_guard<std::mutex> g(m);
        stop = true;
        cv.notify_all();
      }
      th.join(); // (the loop empties the queue before it looks at `stop`)
    }
    return err;
  }
  ~WindowUnpacker() { (void)finish(); }
};

} // namespace

int msgpu_overlap_batched(msgpu_ctx *c, const msgpu_row *rows, size_t n_rows, uint32_t n_batches, msgpu_host_tables *out) {
  return msgpu_overlap_batched_ex(c, rows, n_rows, n_batches, 0, out);
}

static int overlap_batched_impl(msgpu_ctx *c, const msgpu_row *rows, size_t n_rows, uint32_t n_batches, uint32_t flags,
                                msgpu_host_tables *out);
int msgpu_overlap_batched_ex(msgpu_ctx *c, const msgpu_row *rows, size_t n_rows, uint32_t n_batches, uint32_t flags,
                             msgpu_host_tables *out) {
  if (!c || !out) return MSGPU_E_ARG;
  try { // (the dispatcher keeps a few small host containers and starts a thread: nothing C++ may leave through the C boundary)
    return overlap_batched_impl(c, rows, n_rows, n_batches, flags, out);
  } catch (const std::bad_alloc &) {
    return fail(c, MSGPU_E_NOMEM, "host memory for the dispatcher's bookkeeping");
  } catch (...) {
    return fail(c, MSGPU_E_HIP, "unexpected C++ exception in the dispatcher");
  }
}
static int overlap_batched_impl(msgpu_ctx *c, const msgpu_row *rows, size_t n_rows, uint32_t n_batches, uint32_t flags,
                                msgpu_host_tables *out) {
  memset(out, 0, sizeof(*out));
  if (flags & ~(MSGPU_BATCH_RESIDENT | MSGPU_BATCH_NO_EDGEMATCHES | MSGPU_BATCH_ROWS_ON_DEVICE | MSGPU_BATCH_ROWS_PACKED))
    return fail(c, MSGPU_E_ARG, "unknown flags %#x", flags);
  if ((flags & MSGPU_BATCH_ROWS_PACKED) && (flags & MSGPU_BATCH_ROWS_ON_DEVICE))
    return fail(c, MSGPU_E_ARG, "MSGPU_BATCH_ROWS_PACKED is a host form: not with MSGPU_BATCH_ROWS_ON_DEVICE");
  if (flags & MSGPU_BATCH_NO_EDGEMATCHES) flags |= MSGPU_BATCH_RESIDENT; // (an EdgeMatch that is not copied must stay)
  const bool resident = (flags & MSGPU_BATCH_RESIDENT) != 0, copy_ems = !(flags & MSGPU_BATCH_NO_EDGEMATCHES);
  const auto t_start = std::chrono::steady_clock::now();
  auto       ms_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  DevBuf *const tabs[4] = {&c->edges, &c->ems, &c->orders, &c->ids};
  auto reset_views = [&]() {
    for (DevBuf *b : tabs) {
      b->off = b->hint = 0;
      b->own_grow_by   = 0.0;
    }
  };
  c->win_lo = 0;
  c->win_hi = 0xffffffffu;
  c->base_edges = c->base_ems = c->base_orders = c->base_ids = 0;
  reset_views();
  // windows: the caller's choice, else 8 when all four tables travel (the link is the bound: many windows hide the first
  // one's compute) and 3 when the EdgeMatch table stays (compute and copies weigh about the same: few, large windows)
  if (!n_batches) n_batches = copy_ems ? 8 : 3;
  c->no_prologue = n_batches > 1; // several windows: each has its own scratch offsets and read lists
  // rows host -> HBM once (unless they are there already), index build once
  const int rc_load = (flags & MSGPU_BATCH_ROWS_PACKED)      ? msgpu_load_rows_packed(c, reinterpret_cast<const msgpu_packed_rows *>(rows))
                      : (flags & MSGPU_BATCH_ROWS_ON_DEVICE) ? msgpu_load_rows_device(c, rows, n_rows)
                                                             : msgpu_load_rows(c, rows, n_rows);
  c->no_prologue    = false;
  if (rc_load) return rc_load;
  out->load_ms = ms_since(t_start);
  const uint32_t V = c->V;
  uint32_t       B = n_batches;
  if (B > 256) B = 256;
  if (B > V) B = V ? V : 1;
  hipStream_t st = c->stream, cs = c->copy_stream;

  // Vertex facts (Vertex::getNanoporeLength, metaDatum(0)) go first on the copy stream
  if (int rc = ensure_host(c, c->h_read_len, (size_t(V) + 1) * 4, 0, 0)) return rc;
  if (int rc = ensure_host(c, c->h_read_first, (size_t(V) + 1) * 4, 0, 0)) return rc;
  HIPCHK(c, hipEventRecord(c->ev_done[0], st));
  HIPCHK(c, hipStreamWaitEvent(cs, c->ev_done[0], 0));
  if (V) {
    HIPCHK(c, hipMemcpyAsync(c->h_read_len.p, c->read_len.p, size_t(V) * 4, hipMemcpyDeviceToHost, cs));
    HIPCHK(c, hipMemcpyAsync(c->h_read_first.p, c->read_first.p, size_t(V) * 4, hipMemcpyDeviceToHost, cs));
  }

  // Windows of owner reads, cut by measured work: the index build left, per read, the scaffold rows it visits as an owner
  // (`visits`: what its candidate scan costs and, closely, what it yields); a prefix sum over them and one binary search per
  // cut give windows that hold the wanted shares of the job whatever the read ids have to do with genome position.  Shares:
  // equal -- unless the windows leave in wire form (below): then the job waits for the compute, and what is left when the last
  // window is done is that window's copy and its unpacking, so the windows shrink towards the end (9 : 7 : 4 for three).
  // Without visit counts (a generic index; nothing to visit) the cuts assume read ids unrelated to genome position: read r owns
  // about (V - r) / V of its pairs, windows that end at V (1 - sqrt(1 - k/B)) hold equal shares.
  const bool wire_wanted = !copy_ems && c->wire_copy && V != 0;
  std::vector<uint32_t> cuts(size_t(B) + 1, V);
  std::vector<double>   cum_share(size_t(B) + 1, 1.0); // share of the job's work in front of every cut (for the size hints)
  cuts[0]      = 0;
  cum_share[0] = 0.0;
  bool measured = false;
  if (B > 1 && c->index_fast) {
    WindowCutArgs wa;
    wa.n = B - 1;
    double acc = 0, sum = 0;
    auto   weight = [&](uint32_t k) { return wire_wanted ? 1.0 + 1.25 * double(B - 1 - k) / double(B - 1) : 1.0; };
    for (uint32_t k = 0; k < B; ++k) sum += weight(k);
    for (uint32_t k = 0; k + 1 < B; ++k) {
      acc += weight(k) / sum;
      wa.frac[k] = static_cast<float>(acc);
    }
    ENSURE(c, cand_off, (size_t(V) + 2) * 8);
    ENSURE(c, win_cuts, (2 * size_t(B) + 1) * 8);
    if (!c->full_scan_ok) // (a bin-path build left exactly this scan in cand_off, its total in cand_off[V]: k_index_epilogue)
      exclusive_scan<uint64_t>(st, c->visits.as<uint32_t>(), V, c->cand_off.as<uint64_t>(), c->scan_tmp.as<uint64_t>(),
                               c->cand_off.as<uint64_t>() + V);
    launch_window_cuts(st, c->cand_off.as<uint64_t>(), V, wa, c->win_cuts.as<uint64_t>());
    HIPCHK(c, hipGetLastError());
    std::vector<uint64_t> got(2 * size_t(B) - 1);
    HIPCHK(c, hipMemcpyAsync(got.data(), c->win_cuts.p, got.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const uint64_t total = got[2 * size_t(B) - 2];
    if (total) {
      measured = true;
      for (uint32_t k = 1; k < B; ++k) {
        cuts[k]      = static_cast<uint32_t>(std::min<uint64_t>(got[k - 1], V));
        cum_share[k] = double(got[B - 1 + k - 1]) / double(total);
      }
    }
  }
  if (!measured)
    for (uint32_t k = 1; k < B; ++k) {
      const double x = 1.0 - std::sqrt(1.0 - double(k) / double(B));
      cuts[k]        = static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(x * double(V)), V));
      // (the smaller of the two shares a cut can hold, see below: ids unrelated to position / ids that follow the genome)
      cum_share[k] = double(cuts[k]) / double(V ? V : 1);
    }
  for (uint32_t k = 1; k < B; ++k) cuts[k] = std::max(cuts[k], cuts[k - 1] + 1);            // no empty window (B <= V)
  for (uint32_t k = B - 1; k >= 1; --k) cuts[k] = std::min(cuts[k], V - (B - k));
  auto cut = [&](uint32_t k) -> uint32_t { return k >= B ? V : cuts[k]; };
  uint64_t tot_e = 0, tot_m = 0, tot_o = 0, tot_i = 0, tot_fast = 0;
  int      rc = MSGPU_OK;
  // The EdgeMatch table stays in HBM: from the first window on the job waits for the host link, so the other three tables
  // cross it in the exchange's wire form (17 + 33 bytes per edge + order instead of 32 + 64, three bytes per anchor id while
  // the ids fit 24 bits) and a host thread turns every window back into records while the next one computes.
  WindowUnpacker unpacker;
  unpacker.debug = std::getenv("MSGPU_BATCH_DEBUG") != nullptr;
  unpacker.t0    = t_start;
  bool           wire = !copy_ems && c->wire_copy && V != 0 && unpacker.start(c->device);
  const bool     wire_ids3 = c->A <= (1u << 24);
  uint64_t       job_of_set[2] = {0, 0}; // the unpacker's job count after the last window that used this set's host block
  const bool dbg = std::getenv("MSGPU_BATCH_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[batched] rows loaded + index at %.2f ms\n", ms_since(t_start));
  struct GrowGuard { // (back to 1 however the loop is left)
    ~GrowGuard() { DevBuf::grow_by = 1.0; }
  } grow_guard;
  for (uint32_t k = 0; k < B && rc == MSGPU_OK; ++k) {
    const int set = static_cast<int>(k & 1);
    c->win_lo      = cut(k);
    c->win_hi      = cut(k + 1);
    { // the largest window still to come, in reads, against this one (records per read are about even)
      uint32_t most = 0;
      for (uint32_t j = k + 1; j < B; ++j) most = std::max(most, cut(j + 1) - cut(j));
      const uint32_t mine = c->win_hi - c->win_lo;
      DevBuf::grow_by     = mine && most > mine ? std::min(6.0, double(most) / double(mine)) : 1.0;
    }
    c->base_edges  = tot_e;
    c->base_ems    = tot_m;
    c->base_orders = tot_o;
    c->base_ids    = tot_i;
    // the share of the job's records the windows up to and including this one hold (used to extrapolate sizes): measured by
    // the visit counts the cuts were made with, or -- without them -- at least: with read ids unrelated to genome position a
    // read owns its pairs with every later read and the windows hold 1 - (1 - hi/V)^2 of the records; with ids that follow the
    // genome (a PAF whose anchors come in assembly order) every read owns about half of its pairs and the share is hi/V.
    // Real inputs lie between, so sizes are then extrapolated with the smaller share: too large a table costs address space,
    // too small a one an allocation and a move in the middle of the job.
    const double share        = cum_share[k + 1];
    const double share_before = cum_share[k]; // ... the windows before this one hold
    auto         hint_of = [&](uint64_t have, size_t rec, double sh) {
      return static_cast<size_t>(double(have) / (sh > 0.02 ? sh : 0.02) * 1.15) * rec;
    };
    auto hint = [&](uint64_t have, size_t rec) { return hint_of(have, rec, share); };
    if (resident) {
      // the job's tables stay whole in HBM: this window writes behind the earlier ones
      c->edges.off  = tot_e * sizeof(msgpu_edge);
      c->ems.off    = tot_m * sizeof(msgpu_edgematch);
      c->orders.off = tot_o * sizeof(msgpu_order);
      c->ids.off    = tot_i * 4;
      // the first window cannot be extrapolated from: its tables are asked to hold the job at this window's share of it
      for (DevBuf *b : tabs) b->own_grow_by = k == 0 && share > 0 ? std::min(8.0, 1.15 / share) : 0.0;
      if (k) { // (what the windows so far produced, extrapolated to the job by THEIR share of it)
        c->edges.hint  = hint_of(tot_e, sizeof(msgpu_edge), share_before);
        c->ems.hint    = hint_of(tot_m, sizeof(msgpu_edgematch), share_before);
        c->orders.hint = hint_of(tot_o, sizeof(msgpu_order), share_before);
        c->ids.hint    = hint_of(tot_i, 4, share_before);
      }
    } else if (k >= 2) {
      // this set of output tables was last read by the copy of batch k - 2
      rc = hipStreamWaitEvent(st, c->ev_copied[set], 0) == hipSuccess ? MSGPU_OK : fail(c, MSGPU_E_HIP, "hipStreamWaitEvent failed");
    }
    if (rc == MSGPU_OK) rc = msgpu_calculate_edges(c);
    if (rc == MSGPU_OK) rc = msgpu_chaining_and_overlaps(c);
    if (rc != MSGPU_OK) break;
    if (tot_e + c->n_edges >= 0xfffffff0ull) {
      rc = fail(c, MSGPU_E_ARG, "edge table too large");
      break;
    }
    if (k == 0) out->first_batch_ms = ms_since(t_start);
    auto guarded = [&](hipError_t e, const char *what) {
      if (e != hipSuccess && rc == MSGPU_OK) rc = fail(c, MSGPU_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    // this window in wire form?  (32-bit offsets inside a window; the staging block of this set was last read by the copy of
    // window k - 2)
    const bool     w_this = wire && c->n_ems <= 0xffffffffull && c->n_orders <= 0xffffffffull && c->n_ids <= 0xffffffffull;
    const uint64_t w_off_o = (wire_edges_bytes(c->n_edges) + 255) & ~uint64_t(255),
                   w_off_i = (w_off_o + wire_orders_bytes(c->n_orders) + 255) & ~uint64_t(255),
                   w_total = wire_ids3 ? ((w_off_i + wire_ids_bytes(c->n_ids, 3) + 255) & ~uint64_t(255)) : w_off_i;
    PackWireArgs pa;
    auto         pack_into = [&](void *block, hipStream_t on) {
      pa.edges       = static_cast<const msgpu_edge *>(c->edges.at());
      pa.orders      = static_cast<const msgpu_order *>(c->orders.at());
      pa.n_edges     = c->n_edges;
      pa.n_orders    = c->n_orders;
      pa.w_edges     = static_cast<uint8_t *>(block);
      pa.w_orders    = pa.w_edges + w_off_o;
      pa.ids         = wire_ids3 ? static_cast<const uint32_t *>(c->ids.at()) : nullptr;
      pa.n_ids       = c->n_ids;
      pa.w_ids       = reinterpret_cast<uint32_t *>(pa.w_edges + w_off_i);
      pa.base_edges  = static_cast<uint32_t>(tot_e);
      pa.base_ems    = tot_m;
      pa.base_orders = tot_o;
      pa.base_ids    = tot_i;
      launch_pack_wire(on, pa);
      guarded(hipGetLastError(), "the pack kernel");
    };
    if (w_this) {
      guarded(c->wire_dev[set].ensure(w_total), "device block of the wire form");
      if (rc == MSGPU_OK && k >= 2) guarded(hipStreamWaitEvent(st, c->ev_copied[set], 0), "hipStreamWaitEvent");
      if (rc != MSGPU_OK) break;
      pack_into(c->wire_dev[set].p, st);
    }
    guarded(hipEventRecord(c->ev_done[set], st), "hipEventRecord");
    if (dbg) fprintf(stderr, "[batched] window %u: launched at %.2f ms (edges %llu, orders %llu, ids %llu%s)\n", k, ms_since(t_start),
                     (unsigned long long)c->n_edges, (unsigned long long)c->n_orders, (unsigned long long)c->n_ids, w_this ? ", wire form" : "");
    if (wire) { // a host table that has to grow moves: every window before this one must be in it first
      const bool grows = (tot_e + c->n_edges + 1) * sizeof(msgpu_edge) > c->h_edges.cap || (tot_o + c->n_orders + 1) * sizeof(msgpu_order) > c->h_orders.cap ||
                         (tot_i + c->n_ids + 1) * 4 > c->h_ids.cap;
      if (grows) unpacker.drain();
      if (w_this) unpacker.wait_done(job_of_set[set]); // the host block of this set is free again
      if (rc == MSGPU_OK && w_this) rc = ensure_host(c, c->h_wire[set], w_total, 0, 0);
    }
    // room in the pinned result tables; the expected job size is extrapolated from what the windows so far produced
    if (rc == MSGPU_OK) rc = ensure_host(c, c->h_edges, (tot_e + c->n_edges + 1) * sizeof(msgpu_edge), tot_e * sizeof(msgpu_edge), hint(tot_e + c->n_edges, sizeof(msgpu_edge)));
    if (rc == MSGPU_OK && copy_ems) rc = ensure_host(c, c->h_ems, (tot_m + c->n_ems + 1) * sizeof(msgpu_edgematch), tot_m * sizeof(msgpu_edgematch), hint(tot_m + c->n_ems, sizeof(msgpu_edgematch)));
    if (rc == MSGPU_OK) rc = ensure_host(c, c->h_orders, (tot_o + c->n_orders + 1) * sizeof(msgpu_order), tot_o * sizeof(msgpu_order), hint(tot_o + c->n_orders, sizeof(msgpu_order)));
    if (rc == MSGPU_OK) rc = ensure_host(c, c->h_ids, (tot_i + c->n_ids + 1) * 4, tot_i * 4, hint(tot_i + c->n_ids, 4));
    if (rc != MSGPU_OK) break;
    if (dbg) fprintf(stderr, "[batched] window %u: host tables ready at %.2f ms (caps %zu %zu %zu)\n", k, ms_since(t_start),
                     c->h_edges.cap, c->h_orders.cap, c->h_ids.cap);
    guarded(hipStreamWaitEvent(cs, c->ev_done[set], 0), "hipStreamWaitEvent");
    // (Device-to-host copies are shader blits on this stack -- __amd_rocclr_copyBuffer in a kernel trace -- and a compute
    // kernel beside one makes next to no progress until the copy is over (profiles/r4_06/window_timeline_*.txt: the 16 us
    // kernel that opens the next window takes as long as the copy beside it; k_chain beside a copy takes its own time plus
    // the copy's).  So windows buy little overlap, the job is the sum of compute and link time, and fewer bytes are what
    // helps: few, large windows, wire form.  Tried and dropped: copies in 1 MB pieces (30.6 instead of 22.0 ms for the full
    // tables); a window's copy held back until the next window's long kernels start; a copy kernel of our own with 8..256
    // workgroups writing mapped host memory (every size slower: 8.1-9.1 against 7.9 ms -- it is the host writes, not the
    // occupied wave slots, that stall the others); the pack kernel writing host memory itself (8.9 ms); host tables from
    // hipHostMalloc instead of registered blocks; the runtime's copy-engine switches.)
    if (w_this) {
      // three blocks, three copies, three unpacking jobs: the edge block first (the orders take their vertices from it), and what
      // is left to do when the last copy lands is one table
      const uint8_t *hw = static_cast<const uint8_t *>(c->h_wire[set].p);
      WindowUnpacker::Job j;
      j.w_edges     = hw;
      j.w_orders    = hw + w_off_o;
      j.w_ids       = wire_ids3 ? reinterpret_cast<const uint32_t *>(hw + w_off_i) : nullptr;
      j.n_edges     = c->n_edges;
      j.n_orders    = c->n_orders;
      j.n_ids       = c->n_ids;
      j.base_edges  = tot_e;
      j.base_ems    = tot_m;
      j.base_orders = tot_o;
      j.base_ids    = tot_i;
      j.edges       = static_cast<msgpu_edge *>(c->h_edges.p) + tot_e;
      j.orders      = static_cast<msgpu_order *>(c->h_orders.p) + tot_o;
      j.ids         = static_cast<uint32_t *>(c->h_ids.p) + tot_i;
      const uint64_t lo[3] = {0, w_off_o, w_off_i}, hi[3] = {w_off_o, w_off_i, w_total};
      for (int part = 0; part < 3 && rc == MSGPU_OK; ++part) {
        if (hi[part] > lo[part])
          guarded(hipMemcpyAsync(static_cast<char *>(c->h_wire[set].p) + lo[part], static_cast<const char *>(c->wire_dev[set].p) + lo[part],
                                 hi[part] - lo[part], hipMemcpyDeviceToHost, cs), "copy of a wire block");
        if (part == 2 && !wire_ids3 && c->n_ids) // 4-byte ids are not packed: they go straight into the id table
          guarded(hipMemcpyAsync(static_cast<uint32_t *>(c->h_ids.p) + tot_i, c->ids.at(), c->n_ids * 4, hipMemcpyDeviceToHost, cs), "copy of the id table");
        guarded(hipEventRecord(c->ev_part[set][part], cs), "hipEventRecord");
        if (rc != MSGPU_OK) break;
        j.copied = c->ev_part[set][part];
        j.tables = 1u << part;
        unpacker.push(j);
      }
      guarded(hipEventRecord(c->ev_copied[set], cs), "hipEventRecord");
      job_of_set[set] = unpacker.pushed; // (only this thread pushes)
    } else {
      if (c->n_edges) guarded(hipMemcpyAsync(static_cast<msgpu_edge *>(c->h_edges.p) + tot_e, c->edges.at(), c->n_edges * sizeof(msgpu_edge), hipMemcpyDeviceToHost, cs), "copy of the edge table");
      if (c->n_ems && copy_ems) guarded(hipMemcpyAsync(static_cast<msgpu_edgematch *>(c->h_ems.p) + tot_m, c->ems.at(), c->n_ems * sizeof(msgpu_edgematch), hipMemcpyDeviceToHost, cs), "copy of the EdgeMatch table");
      if (c->n_orders) guarded(hipMemcpyAsync(static_cast<msgpu_order *>(c->h_orders.p) + tot_o, c->orders.at(), c->n_orders * sizeof(msgpu_order), hipMemcpyDeviceToHost, cs), "copy of the order table");
      if (c->n_ids) guarded(hipMemcpyAsync(static_cast<uint32_t *>(c->h_ids.p) + tot_i, c->ids.at(), c->n_ids * 4, hipMemcpyDeviceToHost, cs), "copy of the id table");
    }
    if (!w_this) guarded(hipEventRecord(c->ev_copied[set], cs), "hipEventRecord");
    tot_e += c->n_edges;
    tot_m += c->n_ems;
    tot_o += c->n_orders;
    tot_i += c->n_ids;
    tot_fast += c->n_edges_fast;
    if (!resident) { // the next batch writes the other set
      std::swap(c->edges, c->alt_edges);
      std::swap(c->ems, c->alt_ems);
      std::swap(c->orders, c->alt_orders);
      std::swap(c->ids, c->alt_ids);
    }
  }
  out->compute_done_ms = ms_since(t_start);
  // WaitGroup::wait(): every batch's tables are in host memory
  const hipError_t e0 = unpacker.finish(); // (its windows are records now)
  hipError_t e1 = hipStreamSynchronize(cs), e2 = hipStreamSynchronize(st);
  if (dbg) fprintf(stderr, "[batched] loop left at %.2f ms, streams idle at %.2f ms\n", out->compute_done_ms, ms_since(t_start));
  c->win_lo = 0;
  c->win_hi = 0xffffffffu;
  c->base_edges = c->base_ems = c->base_orders = c->base_ids = 0;
  reset_views();
  c->state = ST_LOADED; // the context's own tables hold one batch only: results are the host tables
  if (rc != MSGPU_OK) return rc;
  HIPCHK(c, e0);
  HIPCHK(c, e1);
  HIPCHK(c, e2);
  if (resident) { // ... or the whole job: the state msgpu_chaining_and_overlaps leaves, minus the per-edge scratch
    c->n_edges      = tot_e;
    c->n_ems        = tot_m;
    c->n_orders     = tot_o;
    c->n_ids        = tot_i;
    c->n_edges_fast = tot_fast;
    c->state        = ST_RESULT;
  }
  out->edges           = static_cast<const msgpu_edge *>(c->h_edges.p);
  out->ems             = copy_ems ? static_cast<const msgpu_edgematch *>(c->h_ems.p) : nullptr;
  out->orders          = static_cast<const msgpu_order *>(c->h_orders.p);
  out->ids             = static_cast<const uint32_t *>(c->h_ids.p);
  out->read_len        = static_cast<const int32_t *>(c->h_read_len.p);
  out->read_first_line = static_cast<const uint32_t *>(c->h_read_first.p);
  out->n_edges         = tot_e;
  out->n_ems           = tot_m;
  out->n_orders        = tot_o;
  out->n_ids           = tot_i;
  out->n_reads         = V;
  out->n_anchors       = c->A;
  out->n_batches       = B;
  out->wall_ms         = ms_since(t_start);
  return MSGPU_OK;
}

// MatchMap::getEdgeMatches(edge) (libms/src/matching/MatchMap.cpp:136-159) for a list of edges, out of the EdgeMatch table
// resident in HBM: what assemblePath reads of it (ap.cpp:631-706 via dg.cpp:99-101) is the EdgeMatches of the path edges.
int msgpu_get_edgematches(msgpu_ctx *c, const uint32_t *edge_idx, size_t n, const uint64_t **em_off,
                          const msgpu_edgematch **ems) {
  if (!c || !em_off || !ems || (n && !edge_idx)) return MSGPU_E_ARG;
  *em_off = nullptr;
  *ems    = nullptr;
  if (c->state < ST_CHAINED) return fail(c, MSGPU_E_STATE, "no EdgeMatch table resident (run the chaining stage first)");
  if (n >= 0xfffffff0ull) return fail(c, MSGPU_E_ARG, "edge list too long");
  for (size_t i = 0; i < n; ++i)
    if (edge_idx[i] >= c->n_edges) return fail(c, MSGPU_E_ARG, "edge index %u out of range (%llu edges)", edge_idx[i], (unsigned long long)c->n_edges);
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  if (int rc = ensure_host(c, c->h_sel_off, (n + 1) * 8, 0, 0)) return rc;
  uint64_t *h_off = static_cast<uint64_t *>(c->h_sel_off.p);
  h_off[0]        = 0;
  if (n) {
    ENSURE(c, sel_idx, n * 4);
    ENSURE(c, sel_cnt, (n + 1) * 4);
    ENSURE(c, sel_off, (n + 2) * 8);
    ENSURE(c, scan_tmp, 3 * (size_t(scan_blocks(n)) + 1) * 8);
    HIPCHK(c, hipMemcpyAsync(c->sel_idx.p, edge_idx, n * 4, hipMemcpyHostToDevice, st));
    launch_em_counts(st, c->edges.as<msgpu_edge>(), c->sel_idx.as<uint32_t>(), n, c->sel_cnt.as<uint32_t>());
    exclusive_scan<uint64_t>(st, c->sel_cnt.as<uint32_t>(), n, c->sel_off.as<uint64_t>(), c->scan_tmp.as<uint64_t>(),
                             c->sel_off.as<uint64_t>() + n + 1); // (out[n] = the total closes the offset list)
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_off, c->sel_off.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  const uint64_t total = h_off[n];
  if (int rc = ensure_host(c, c->h_sel_ems, (total + 1) * sizeof(msgpu_edgematch), 0, 0)) return rc;
  if (total) {
    ENSURE(c, sel_ems, total * sizeof(msgpu_edgematch));
    launch_em_gather(st, c->edges.as<msgpu_edge>(), c->ems.as<msgpu_edgematch>(), c->sel_idx.as<uint32_t>(), n,
                     c->sel_off.as<uint64_t>(), c->sel_ems.as<msgpu_edgematch>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_sel_ems.p, c->sel_ems.p, total * sizeof(msgpu_edgematch), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  *em_off = h_off;
  *ems    = static_cast<const msgpu_edgematch *>(c->h_sel_ems.p);
  return MSGPU_OK;
}

void *msgpu_pinned_alloc(size_t bytes) { return pinned_block_alloc(bytes); }
void  msgpu_pinned_free(void *p) { pinned_block_free(p); }

uint64_t msgpu_chain_launches(msgpu_ctx *c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> g(c->gate_m);
  return c->chain_launches;
}
int msgpu_wait_chain_launch(msgpu_ctx *c, uint64_t count, uint32_t timeout_us) {
  if (!c) return MSGPU_E_ARG;
  std::unique_lock<std::mutex> lk(c->gate_m);
  return c->gate_cv.wait_for(lk, std::chrono::microseconds(timeout_us), [&] { return c->chain_launches >= count; }) ? MSGPU_OK : 1;
}

int msgpu_synchronize(msgpu_ctx *c) {
  if (!c) return MSGPU_E_ARG;
  return host_sync(c, c->stream);
}

int msgpu_set_deadline(msgpu_ctx *c, uint32_t timeout_ms) {
  if (!c) return MSGPU_E_ARG;
  c->has_deadline = timeout_ms != 0;
  if (timeout_ms) c->deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  return MSGPU_OK;
}

} // extern "C"
