// msgpu_group.cpp -- one process, the node's GPUs: a group of contexts behind the reference's call site.
//
// The reference is one process whose phases fan jobs over ThreadPool workers and end in WaitGroup::wait()
// (src/main.cpp:143-178, libms/src/threading/ThreadPool.cpp:38-129, WaitGroup.cpp:62-72).  Here the workers are devices: one
// msgpu_ctx per GPU, one host thread per GPU while a call runs (a context's calls wait for table sizes, so n devices need
// n callers), device i owns the edges with v1 % n == i, and the phase ends with ONE grouped RCCL all-gather over xGMI of the
// members' wire-form slabs + msgpu_merge_wire on every device.  Everything goes through the public C-ABI of include/msgpu.h;
// this file adds no kernel.  RCCL is resolved at run time (dlopen): a process that never makes a group never maps it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "msgpu.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclCommInitAll)    CommInitAll    = nullptr;
  decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
  decltype(&ncclCommAbort)      CommAbort      = nullptr;
  decltype(&ncclAllGather)      AllGather      = nullptr;
  decltype(&ncclGroupStart)     GroupStart     = nullptr;
  decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  char why[256] = {0};
};

// the process's RCCL: the copy already loaded under its soname if there is one (a framework's), else the ROCm installation's
Rccl *rccl() {
  static Rccl       r;
  static std::mutex m;
  std::lock_guard<std::mutex> lock(m);
  if (r.handle || r.why[0]) return &r;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    snprintf(r.why, sizeof(r.why), "librccl.so.1 not found: %s", dlerror());
    return &r;
  }
#define RESOLVE(field, symbol)                                                                                          \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, symbol));                                               \
  if (!r.field) snprintf(r.why, sizeof(r.why), "librccl lacks %s", symbol)
  RESOLVE(CommInitAll, "ncclCommInitAll");
  RESOLVE(CommDestroy, "ncclCommDestroy");
  RESOLVE(CommAbort, "ncclCommAbort");
  RESOLVE(AllGather, "ncclAllGather");
  RESOLVE(GroupStart, "ncclGroupStart");
  RESOLVE(GroupEnd, "ncclGroupEnd");
  RESOLVE(GetErrorString, "ncclGetErrorString");
#undef RESOLVE
  return &r;
}

struct DevBlock { // a device buffer that only grows
  void  *p   = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p   = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    hipError_t   e    = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p   = nullptr;
    cap = 0;
  }
};
struct HostBlock { // pinned, grows only
  void  *p   = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    if (p) msgpu_pinned_free(p);
    cap = 0;
    p   = msgpu_pinned_alloc(bytes + bytes / 8 + 4096);
    if (p) cap = bytes + bytes / 8 + 4096;
    return p != nullptr;
  }
  void release() {
    if (p) msgpu_pinned_free(p);
    p   = nullptr;
    cap = 0;
  }
};

struct Member {
  int          device = 0;
  msgpu_ctx   *ctx    = nullptr;
  ncclComm_t   comm   = nullptr;
  DevBlock     slab, gathered, m_edges, m_orders, m_ids, rows;
  hipEvent_t   ev0 = nullptr, ev1 = nullptr;
  // the way out to the host: the member's OWN slab, block by block, on a stream of its own beside the exchange
  hipStream_t  out_stream = nullptr;
  hipEvent_t   ev_packed = nullptr, ev_out[3] = {nullptr, nullptr, nullptr};
  HostBlock    h_slab;
  msgpu_counts counts{};
  int          rc = MSGPU_OK;
  char         err[512] = {0};
  double       compute_ms = 0;
};

constexpr uint64_t ALIGN = 256; // blocks inside a slab (muchsalsa_amd/distributed.py uses the same layout)
uint64_t round_up(uint64_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

} // namespace

struct msgpu_group {
  std::vector<Member> m;
  msgpu_params        p;
  char                err[640] = {0};
  HostBlock           h_edges, h_orders, h_ids, h_read_len, h_read_first;
  bool                comms_ok = false;
  // Rehearsal transport (MSGPU_GROUP_TRANSPORT=copy, read at creation): the all-gather as device-to-device copies issued by
  // this process instead of RCCL, so that a group of SEVERAL members can run on a box with fewer GPUs (members may then share
  // a device).  Everything else -- shards, threads, slab layout, pack, merge -- is the path RCCL carries.  Never the default.
  bool                copy_transport = false;
  // MSGPU_GROUP_ROWS=replicate (read at creation): every member takes the whole row table over its own link, as a group of
  // one does.  Default with several members: a 1/n-th each, all-gathered over xGMI (see msgpu_group_overlap).
  bool                replicate_rows = false;
  // msgpu_group_set_timeout: 0 = the blocking waits of the runtime, else every host wait of a call polls against one deadline
  uint32_t            timeout_ms = 0;
  bool                dirty = false; // an earlier call gave up with work still queued on the members' streams
};

namespace {

int gfail(msgpu_group *g, int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g->err, sizeof(g->err), fmt, ap);
  va_end(ap);
  return code;
}
#define GHIP(g, expr)                                                                                                  \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess)                                                                                              \
      return gfail((g), _e == hipErrorOutOfMemory ? MSGPU_E_NOMEM : MSGPU_E_HIP, "%s failed: %s (%s:%d)", #expr,       \
                   hipGetErrorString(_e), __FILE__, __LINE__);                                                         \
  } while (0)
// the one deadline of a msgpu_group_overlap call
struct Deadline {
  bool                                  on = false;
  std::chrono::steady_clock::time_point at{};
  bool passed() const { return on && std::chrono::steady_clock::now() >= at; }
};
Deadline deadline_in(uint32_t ms) {
  Deadline d;
  d.on = ms != 0;
  if (ms) d.at = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
  return d;
}
// hipStreamSynchronize / hipEventSynchronize that give up at the deadline (0 = done, 1 = deadline passed, else a HIP error)
int wait_for(hipStream_t st, hipEvent_t ev, const Deadline &dl, hipError_t *err) {
  *err = hipSuccess;
  if (!dl.on) {
    *err = ev ? hipEventSynchronize(ev) : hipStreamSynchronize(st);
    return *err == hipSuccess ? 0 : 2;
  }
  for (uint32_t spins = 0;; ++spins) {
    const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) {
      *err = q;
      return 2;
    }
    if (dl.passed()) return 1;
    if (spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
#define GWAIT(g, st, ev, dl, what)                                                                                     \
  do {                                                                                                                 \
    hipError_t _e;                                                                                                     \
    const int  _w = wait_for((st), (ev), (dl), &_e);                                                                   \
    if (_w == 1) return gfail((g), MSGPU_E_TIMEOUT, "%s did not finish within %u ms (msgpu_group_set_timeout)", (what), (g)->timeout_ms); \
    if (_w == 2) return gfail((g), MSGPU_E_HIP, "waiting for %s: %s (%s:%d)", (what), hipGetErrorString(_e), __FILE__, __LINE__);        \
  } while (0)

// test hook (MSGPU_GROUP_TEST_STALL_MS / _AT, read per call): a host function that sleeps on a member's stream -- what a
// collective that does not complete looks like to everything queued behind it
void stall_fn(void *ms) { std::this_thread::sleep_for(std::chrono::milliseconds(reinterpret_cast<uintptr_t>(ms))); }

#define GNCCL(g, expr)                                                                                                 \
  do {                                                                                                                 \
    ncclResult_t _r = (expr);                                                                                          \
    if (_r != ncclSuccess)                                                                                             \
      return gfail((g), MSGPU_E_HIP, "%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__);  \
  } while (0)

} // namespace

extern "C" {

int msgpu_group_create(const int *devices, int n, const msgpu_params *params, msgpu_group **out) {
  if (!out) return MSGPU_E_ARG;
  *out = nullptr;
  if (!devices || n <= 0 || n > 64) return MSGPU_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MSGPU_E_NODEVICE;
  const char *tr   = getenv("MSGPU_GROUP_TRANSPORT");
  const bool  copy = tr && strcmp(tr, "copy") == 0;
  for (int i = 0; i < n; ++i) {
    if (devices[i] < 0 || devices[i] >= ndev) return MSGPU_E_NODEVICE;
    for (int j = 0; j < i; ++j)
      if (devices[j] == devices[i] && !copy) return MSGPU_E_ARG; // one member per device (RCCL: one rank per GPU)
  }
  msgpu_group *g = new (std::nothrow) msgpu_group();
  if (!g) return MSGPU_E_NOMEM;
  int caller_dev = -1;
  if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
  struct Restore { // the calling thread's current device is the caller's business
    int d;
    ~Restore() {
      if (d >= 0) (void)hipSetDevice(d);
    }
  } restore{caller_dev};
  g->copy_transport = copy;
  if (const char *to = getenv("MSGPU_GROUP_TIMEOUT_MS")) g->timeout_ms = static_cast<uint32_t>(strtoul(to, nullptr, 10));
  const char *rw    = getenv("MSGPU_GROUP_ROWS");
  g->replicate_rows = rw && strcmp(rw, "replicate") == 0;
  if (params)
    g->p = *params;
  else
    msgpu_default_params(&g->p);
  g->m.resize(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    Member &mb = g->m[static_cast<size_t>(i)];
    mb.device  = devices[i];
    int rc     = msgpu_create(devices[i], &g->p, &mb.ctx);
    if (rc == MSGPU_OK) rc = msgpu_set_shard(mb.ctx, static_cast<uint32_t>(i), static_cast<uint32_t>(n));
    if (rc == MSGPU_OK && (hipSetDevice(devices[i]) != hipSuccess || hipEventCreate(&mb.ev0) != hipSuccess ||
                           hipEventCreate(&mb.ev1) != hipSuccess || hipStreamCreateWithFlags(&mb.out_stream, hipStreamNonBlocking) != hipSuccess ||
                           hipEventCreateWithFlags(&mb.ev_packed, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&mb.ev_out[0], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&mb.ev_out[1], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&mb.ev_out[2], hipEventDisableTiming) != hipSuccess))
      rc = MSGPU_E_HIP;
    if (rc != MSGPU_OK) {
      msgpu_group_destroy(g);
      return rc;
    }
  }
  *out = g; // (the communicators are made by the first msgpu_group_overlap: creating a group costs no RCCL start-up)
  return MSGPU_OK;
}

void msgpu_group_destroy(msgpu_group *g) {
  if (!g) return;
  for (Member &mb : g->m) {
    (void)hipSetDevice(mb.device);
    if (mb.ctx) (void)msgpu_synchronize(mb.ctx);
    if (mb.comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(mb.comm);
    for (DevBlock *b : {&mb.slab, &mb.gathered, &mb.m_edges, &mb.m_orders, &mb.m_ids, &mb.rows}) b->release();
    if (mb.ev0) (void)hipEventDestroy(mb.ev0);
    if (mb.ev1) (void)hipEventDestroy(mb.ev1);
    if (mb.out_stream) {
      (void)hipStreamSynchronize(mb.out_stream);
      (void)hipStreamDestroy(mb.out_stream);
    }
    for (hipEvent_t e : {mb.ev_packed, mb.ev_out[0], mb.ev_out[1], mb.ev_out[2]})
      if (e) (void)hipEventDestroy(e);
    mb.h_slab.release();
    if (mb.ctx) msgpu_destroy(mb.ctx);
  }
  for (HostBlock *h : {&g->h_edges, &g->h_orders, &g->h_ids, &g->h_read_len, &g->h_read_first}) h->release();
  delete g;
}

const char *msgpu_group_last_error(const msgpu_group *g) { return g ? g->err : "null group"; }
int         msgpu_group_size(const msgpu_group *g) { return g ? static_cast<int>(g->m.size()) : 0; }
msgpu_ctx  *msgpu_group_ctx(msgpu_group *g, int member) {
  return (g && member >= 0 && static_cast<size_t>(member) < g->m.size()) ? g->m[static_cast<size_t>(member)].ctx : nullptr;
}

int msgpu_group_device_tables(msgpu_group *g, int member, const void **d_edges, const void **d_orders, const void **d_ids) {
  if (!g || member < 0 || static_cast<size_t>(member) >= g->m.size()) return MSGPU_E_ARG;
  const Member &mb = g->m[static_cast<size_t>(member)];
  if (d_edges) *d_edges = mb.m_edges.p;
  if (d_orders) *d_orders = mb.m_orders.p;
  if (d_ids) *d_ids = mb.m_ids.p;
  return MSGPU_OK;
}

static int group_overlap_inner(msgpu_group *g, const msgpu_row *rows, size_t n_rows, msgpu_group_tables *out, const Deadline &dl) {
  const auto   t_start = std::chrono::steady_clock::now();
  const size_t n       = g->m.size();
  uint32_t     stall_ms = 0, stall_at = 0;
  if (const char *sm = getenv("MSGPU_GROUP_TEST_STALL_MS")) stall_ms = static_cast<uint32_t>(strtoul(sm, nullptr, 10));
  if (const char *sa = getenv("MSGPU_GROUP_TEST_STALL_AT")) stall_at = static_cast<uint32_t>(strtoul(sa, nullptr, 10));
  auto stall = [&](uint32_t where) -> hipError_t { // on the LAST member's stream
    if (!stall_ms || stall_at != where) return hipSuccess;
    Member &mb = g->m[n - 1];
    hipError_t e = hipSetDevice(mb.device);
    if (e == hipSuccess)
      e = hipLaunchHostFunc(static_cast<hipStream_t>(msgpu_get_stream(mb.ctx)), stall_fn, reinterpret_cast<void *>(static_cast<uintptr_t>(stall_ms)));
    return e;
  };
  Rccl *nc = g->copy_transport ? nullptr : rccl();
  if (nc && nc->why[0]) return gfail(g, MSGPU_E_HIP, "RCCL is not available: %s", nc->why);
  if (nc && !g->comms_ok) { // one communicator per member, all in this process
    std::vector<ncclComm_t> comms(n);
    std::vector<int>        devs(n);
    for (size_t i = 0; i < n; ++i) devs[i] = g->m[i].device;
    GNCCL(g, nc->CommInitAll(comms.data(), static_cast<int>(n), devs.data()));
    for (size_t i = 0; i < n; ++i) g->m[i].comm = comms[i];
    g->comms_ok = true;
  }

  // ---- the rows: n links carry a 1/n-th each, xGMI carries the rest ------------------------------------------------------
  // Every member needs the whole row table (the index is replicated).  n copies of it over n PCIe links would all come out of
  // the same host memory; instead member i takes rows [i per, (i + 1) per) over its link and ONE grouped in-place all-gather
  // on the members' streams completes every copy -- the links move the table once, the fabric (7 x the bandwidth of a link
  // per device) the other n - 1 times.  The index build of a member is stream-ordered behind its all-gather.
  const bool   sliced = n > 1 && !g->replicate_rows && n_rows >= 1024;
  const size_t per    = sliced ? (n_rows + n - 1) / n : 0;
  GHIP(g, stall(0));
  if (sliced) {
    for (size_t i = 0; i < n; ++i) {
      Member &mb = g->m[i];
      GHIP(g, hipSetDevice(mb.device));
      GHIP(g, mb.rows.ensure(n * per * sizeof(msgpu_row)));
      const size_t lo = std::min(i * per, n_rows), hi = std::min(lo + per, n_rows);
      if (hi > lo)
        GHIP(g, hipMemcpyAsync(static_cast<msgpu_row *>(mb.rows.p) + lo, rows + lo, (hi - lo) * sizeof(msgpu_row), hipMemcpyHostToDevice,
                               static_cast<hipStream_t>(msgpu_get_stream(mb.ctx))));
    }
    if (nc) {
      GNCCL(g, nc->GroupStart());
      ncclResult_t first_bad = ncclSuccess;
      for (size_t i = 0; i < n && first_bad == ncclSuccess; ++i) {
        Member &mb = g->m[i];
        first_bad  = nc->AllGather(static_cast<msgpu_row *>(mb.rows.p) + i * per, mb.rows.p, per * sizeof(msgpu_row), ncclChar, mb.comm,
                                   static_cast<hipStream_t>(msgpu_get_stream(mb.ctx)));
      }
      const ncclResult_t closed = nc->GroupEnd();
      GNCCL(g, first_bad);
      GNCCL(g, closed);
    } else { // rehearsal transport: every slice has landed, then member i fetches the other n - 1
      for (size_t i = 0; i < n; ++i) {
        GHIP(g, hipSetDevice(g->m[i].device));
        GWAIT(g, static_cast<hipStream_t>(msgpu_get_stream(g->m[i].ctx)), nullptr, dl, "a member's slice of the rows");
      }
      for (size_t i = 0; i < n; ++i) {
        Member &mb = g->m[i];
        GHIP(g, hipSetDevice(mb.device));
        for (size_t r = 0; r < n; ++r) {
          const size_t lo = std::min(r * per, n_rows), hi = std::min(lo + per, n_rows);
          if (r != i && hi > lo)
            GHIP(g, hipMemcpyPeerAsync(static_cast<msgpu_row *>(mb.rows.p) + lo, mb.device, static_cast<msgpu_row *>(g->m[r].rows.p) + lo,
                                       g->m[r].device, (hi - lo) * sizeof(msgpu_row), static_cast<hipStream_t>(msgpu_get_stream(mb.ctx))));
        }
      }
    }
  }

  // ---- the fan-out: every member builds the index and computes its shard ------------------------------------------------
  auto work = [&](size_t i) {
    Member    &mb = g->m[i];
    const auto t0 = std::chrono::steady_clock::now();
    mb.rc         = hipSetDevice(mb.device) == hipSuccess ? MSGPU_OK : MSGPU_E_HIP;
    if (mb.rc == MSGPU_OK) mb.rc = sliced ? msgpu_load_rows_device(mb.ctx, mb.rows.p, n_rows) : msgpu_load_rows(mb.ctx, rows, n_rows);
    if (mb.rc == MSGPU_OK) mb.rc = msgpu_calculate_edges(mb.ctx);
    if (mb.rc == MSGPU_OK) mb.rc = msgpu_chaining_and_overlaps(mb.ctx);
    if (mb.rc == MSGPU_OK) mb.rc = msgpu_get_counts(mb.ctx, &mb.counts);
    if (mb.rc == MSGPU_OK) mb.rc = msgpu_synchronize(mb.ctx);
    if (mb.rc != MSGPU_OK) snprintf(mb.err, sizeof(mb.err), "member %zu (device %d): %s", i, mb.device, msgpu_last_error(mb.ctx));
    mb.compute_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  };
  {
    std::vector<std::thread> th;
    try {
      for (size_t i = 1; i < n; ++i) th.emplace_back(work, i);
    } catch (...) { // no thread to be had: the members that have none run here, one after the other
      for (size_t i = th.size() + 1; i < n; ++i) work(i);
    }
    work(0);
    for (std::thread &t : th) t.join(); // WaitGroup::wait()
  }
  for (size_t i = 0; i < n; ++i)
    if (g->m[i].rc != MSGPU_OK) return gfail(g, g->m[i].rc, "%s", g->m[i].err);

  GHIP(g, stall(1));
  // ---- the one exchange: wire-form slabs, one grouped all-gather, the merge on every device ---------------------------
  std::vector<uint64_t> counts(3 * n);
  uint64_t              mx[3] = {0, 0, 0}, tot[3] = {0, 0, 0}, n_ems = 0;
  for (size_t i = 0; i < n; ++i) {
    const msgpu_counts &c = g->m[i].counts;
    const uint64_t      v[3] = {c.n_edges, c.n_orders, c.n_ids};
    for (int k = 0; k < 3; ++k) {
      counts[3 * i + k] = v[k];
      mx[k]             = std::max(mx[k], v[k]);
      tot[k] += v[k];
    }
    n_ems += c.n_ems;
  }
  const uint32_t id_bytes = g->m[0].counts.n_anchors <= (1u << 24) ? 3u : 4u; // every member holds the same rows: same id space
  const uint64_t off_e = 0, off_o = round_up(off_e + msgpu_wire_edges_bytes(mx[0])),
                 off_i = round_up(off_o + msgpu_wire_orders_bytes(mx[1])),
                 slab_bytes = std::max<uint64_t>(round_up(off_i + msgpu_wire_ids_bytes(mx[2], id_bytes)), ALIGN);
  // the merged list in host memory is put together from the members' OWN slabs -- wire form over every member's link (a 1/n-th
  // of the job each, 40 % fewer bytes than records), beside the exchange instead of behind it -- and unpacked by host threads
  // with the member's bases (msgpu_unpack_wire_host): the records msgpu_merge_wire writes in HBM, byte for byte
  const uint32_t V = g->m[0].counts.n_reads;
  if (!g->h_edges.ensure(std::max<uint64_t>(tot[0], 1) * sizeof(msgpu_edge)) ||
      !g->h_orders.ensure(std::max<uint64_t>(tot[1], 1) * sizeof(msgpu_order)) || !g->h_ids.ensure(std::max<uint64_t>(tot[2], 1) * 4) ||
      !g->h_read_len.ensure((size_t(V) + 1) * 4) || !g->h_read_first.ensure((size_t(V) + 1) * 4))
    return gfail(g, MSGPU_E_NOMEM, "page-locked host tables of the merged edge list");
  for (size_t i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    GHIP(g, hipSetDevice(mb.device));
    if (!mb.h_slab.ensure(slab_bytes)) return gfail(g, MSGPU_E_NOMEM, "page-locked slab of member %zu", i);
    GHIP(g, mb.slab.ensure(slab_bytes));
    GHIP(g, mb.gathered.ensure(n * slab_bytes));
    GHIP(g, mb.m_edges.ensure(std::max<uint64_t>(tot[0], 1) * sizeof(msgpu_edge)));
    GHIP(g, mb.m_orders.ensure(std::max<uint64_t>(tot[1], 1) * sizeof(msgpu_order)));
    GHIP(g, mb.m_ids.ensure(std::max<uint64_t>(tot[2], 1) * 4));
    hipStream_t st = static_cast<hipStream_t>(msgpu_get_stream(mb.ctx));
    GHIP(g, hipEventRecord(mb.ev0, st));
    char *slab = static_cast<char *>(mb.slab.p);
    if (int rc = msgpu_pack_wire(mb.ctx, slab + off_e, slab + off_o, slab + off_i, id_bytes))
      return gfail(g, rc, "member %zu: %s", i, msgpu_last_error(mb.ctx));
    GHIP(g, hipEventRecord(mb.ev_packed, st));
    GHIP(g, hipStreamWaitEvent(mb.out_stream, mb.ev_packed, 0));
    const uint64_t lo[3] = {off_e, off_o, off_i},
                   len[3] = {msgpu_wire_edges_bytes(counts[3 * i]), msgpu_wire_orders_bytes(counts[3 * i + 1]),
                             msgpu_wire_ids_bytes(counts[3 * i + 2], id_bytes)};
    for (int part = 0; part < 3; ++part) { // edge block first: the orders take their vertices from it
      if (len[part])
        GHIP(g, hipMemcpyAsync(static_cast<char *>(mb.h_slab.p) + lo[part], slab + lo[part], len[part], hipMemcpyDeviceToHost, mb.out_stream));
      GHIP(g, hipEventRecord(mb.ev_out[part], mb.out_stream));
    }
  }
  if (nc) {
    GNCCL(g, nc->GroupStart()); // one thread drives n communicators: the n calls are one collective
    ncclResult_t first_bad = ncclSuccess;
    for (size_t i = 0; i < n && first_bad == ncclSuccess; ++i) {
      Member &mb = g->m[i];
      first_bad  = nc->AllGather(mb.slab.p, mb.gathered.p, slab_bytes, ncclChar, mb.comm, static_cast<hipStream_t>(msgpu_get_stream(mb.ctx)));
    }
    const ncclResult_t closed = nc->GroupEnd(); // (closed whatever happened inside: an open group would swallow the next call)
    GNCCL(g, first_bad);
    GNCCL(g, closed);
  } else { // rehearsal transport: every slab is complete (its stream drained), then member i copies all n slabs
    for (size_t i = 0; i < n; ++i) {
      GHIP(g, hipSetDevice(g->m[i].device));
      GWAIT(g, static_cast<hipStream_t>(msgpu_get_stream(g->m[i].ctx)), nullptr, dl, "a member's slab");
    }
    for (size_t i = 0; i < n; ++i) {
      Member &mb = g->m[i];
      GHIP(g, hipSetDevice(mb.device));
      for (size_t r = 0; r < n; ++r)
        GHIP(g, hipMemcpyPeerAsync(static_cast<char *>(mb.gathered.p) + r * slab_bytes, mb.device, g->m[r].slab.p, g->m[r].device, slab_bytes,
                                   static_cast<hipStream_t>(msgpu_get_stream(mb.ctx))));
    }
  }
  for (size_t i = 0; i < n; ++i) {
    Member &mb = g->m[i];
    GHIP(g, hipSetDevice(mb.device));
    if (int rc = msgpu_merge_wire(mb.ctx, mb.gathered.p, static_cast<uint32_t>(n), counts.data(), slab_bytes, off_e, off_o, off_i,
                                  id_bytes, nullptr, mb.m_edges.p, mb.m_orders.p, mb.m_ids.p, nullptr))
      return gfail(g, rc, "member %zu: %s", i, msgpu_last_error(mb.ctx));
    GHIP(g, hipEventRecord(mb.ev1, static_cast<hipStream_t>(msgpu_get_stream(mb.ctx))));
  }
  // ---- the Vertex facts; the members' slabs into records as their blocks land ---------------------------------------------
  {
    Member &mb = g->m[0];
    GHIP(g, hipSetDevice(mb.device));
    if (V)
      if (int rc = msgpu_copy_reads(mb.ctx, static_cast<int32_t *>(g->h_read_len.p), static_cast<uint32_t *>(g->h_read_first.p)))
        return gfail(g, rc, "member 0: %s", msgpu_last_error(mb.ctx));
  }
  {
    uint64_t base[4] = {0, 0, 0, 0}; // {edges, EdgeMatches (local to the owner: 0), orders, ids} in front of member i
    for (size_t i = 0; i < n; ++i) {
      Member     &mb = g->m[i];
      const char *hs = static_cast<const char *>(mb.h_slab.p);
      GHIP(g, hipSetDevice(mb.device));
      for (int part = 0; part < 3; ++part) {
        GWAIT(g, nullptr, mb.ev_out[part], dl, "a member's slab on its way to the host");
        if (int rc = msgpu_unpack_wire_host(hs + off_e, hs + off_o, hs + off_i, id_bytes, counts[3 * i], counts[3 * i + 1], counts[3 * i + 2], base,
                                            static_cast<msgpu_edge *>(g->h_edges.p) + base[0], static_cast<msgpu_order *>(g->h_orders.p) + base[2],
                                            static_cast<uint32_t *>(g->h_ids.p) + base[3], 16, 1u << part))
          return gfail(g, rc, "unpacking the slab of member %zu", i);
      }
      base[0] += counts[3 * i];
      base[2] += counts[3 * i + 1];
      base[3] += counts[3 * i + 2];
    }
  }
  float exchange_ms = 0;
  for (size_t i = 0; i < n; ++i) { // the phase barrier: every member's stream has drained
    Member &mb = g->m[i];
    GHIP(g, hipSetDevice(mb.device));
    GWAIT(g, static_cast<hipStream_t>(msgpu_get_stream(mb.ctx)), nullptr, dl, "the exchange (all-gather + merge) on a member's stream");
    float ms = 0;
    GHIP(g, hipEventElapsedTime(&ms, mb.ev0, mb.ev1));
    exchange_ms = std::max(exchange_ms, ms);
  }
  out->edges           = static_cast<const msgpu_edge *>(g->h_edges.p);
  out->orders          = static_cast<const msgpu_order *>(g->h_orders.p);
  out->ids             = static_cast<const uint32_t *>(g->h_ids.p);
  out->read_len        = static_cast<const int32_t *>(g->h_read_len.p);
  out->read_first_line = static_cast<const uint32_t *>(g->h_read_first.p);
  out->n_edges         = tot[0];
  out->n_orders        = tot[1];
  out->n_ids           = tot[2];
  out->n_ems           = n_ems;
  out->n_reads         = V;
  out->n_anchors       = g->m[0].counts.n_anchors;
  out->n_members       = static_cast<uint32_t>(n);
  out->id_bytes        = id_bytes;
  out->slab_bytes      = slab_bytes;
  out->rows_sliced     = sliced ? 1u : 0u;
  double cm = 0;
  for (const Member &mb : g->m) cm = std::max(cm, mb.compute_ms);
  out->compute_ms  = static_cast<float>(cm);
  out->exchange_ms = exchange_ms;
  out->wall_ms     = static_cast<float>(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
  g->err[0]        = 0;
  return MSGPU_OK;
}

// every member's stream and copy-out stream has drained (true), or the deadline passed first / a wait failed (false)
static bool drain_members(msgpu_group *g, const Deadline &dl) {
  bool ok = true;
  for (Member &mb : g->m) {
    if (hipSetDevice(mb.device) != hipSuccess) {
      ok = false;
      continue;
    }
    hipError_t e;
    if (mb.ctx && wait_for(static_cast<hipStream_t>(msgpu_get_stream(mb.ctx)), nullptr, dl, &e) != 0) ok = false;
    if (mb.out_stream && wait_for(mb.out_stream, nullptr, dl, &e) != 0) ok = false;
  }
  (void)hipGetLastError();
  return ok;
}

int msgpu_group_set_timeout(msgpu_group *g, uint32_t timeout_ms) {
  if (!g) return MSGPU_E_ARG;
  g->timeout_ms = timeout_ms;
  return MSGPU_OK;
}

int msgpu_group_overlap(msgpu_group *g, const msgpu_row *rows, size_t n_rows, msgpu_group_tables *out) {
  if (!g || !out) return MSGPU_E_ARG;
  memset(out, 0, sizeof(*out));
  if (n_rows && !rows) return gfail(g, MSGPU_E_ARG, "Unexpected nullptr.");
  int caller_dev = -1;
  if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
  const Deadline dl = deadline_in(g->timeout_ms);
  for (Member &mb : g->m) (void)msgpu_set_deadline(mb.ctx, g->timeout_ms); // the members' own waits (table sizes) share it
  int rc = MSGPU_OK;
  if (g->dirty) { // an earlier call gave up with work still queued: it has to be gone before buffers are written again
    g->dirty = !drain_members(g, dl);
    if (g->dirty) rc = gfail(g, MSGPU_E_TIMEOUT, "the members' streams still hold work of an earlier call that timed out");
  }
  if (rc == MSGPU_OK) rc = group_overlap_inner(g, rows, n_rows, out, dl);
  if (rc != MSGPU_OK) {
    // No early return leaves work behind that reads the caller's rows or writes the group's host tables: abort what can hang
    // (a timed-out collective), then wait for everything the call queued -- for ever without a timeout, else once more as long.
    memset(out, 0, sizeof(*out));
    Rccl *nc = g->copy_transport ? nullptr : rccl();
    if (rc == MSGPU_E_TIMEOUT && nc && g->comms_ok) {
      for (Member &mb : g->m) {
        if (mb.comm && nc->CommAbort) (void)nc->CommAbort(mb.comm);
        mb.comm = nullptr;
      }
      g->comms_ok = false; // the next call builds fresh communicators
    }
    g->dirty = !drain_members(g, deadline_in(g->timeout_ms));
    const size_t len = strlen(g->err);
    snprintf(g->err + len, sizeof(g->err) - len, "%s", g->dirty ? "; the members' streams have NOT drained: keep `rows` alive until the next call or msgpu_group_destroy"
                                                                : "; everything the call queued has finished");
  }
  for (Member &mb : g->m) (void)msgpu_set_deadline(mb.ctx, 0);
  if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
  return rc;
}

} // extern "C"
