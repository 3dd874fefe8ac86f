// msgpu_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the MuCHSALSA overlap core.
//
// Pipeline (SURVEY.md section 8, rows A1-tail .. A7):
//
//   index build   rows (40 B, HBM) -> two 32-B row tables:  by_read  [read ][nanoporeRange, anchor]   (m_vertexMatches)
//                                                           by_anchor[anchor][line]                    (m_scaffolds)
//   candidates    one workgroup per owner read v1: scan the scaffolds of v1's anchors, keep rows of reads v2 > v1
//                 whose anchor interval overlaps by > TH_OVERLAP, sort (v2, j) in LDS, cut into edges
//                 (= MatchMap::processScaffold, MatchMap.cpp:175-224, turned inside out so that every edge is born
//                 grouped and in vStart order -- no global sort of EdgeMatches)
//   chain         one wavefront per edge, one lane per EdgeMatch: EdgeMatch score, corrected ranges, the O(n^2)
//                 chaining DP with readlane broadcast + per-lane path bitmasks, alternatives, demotion, filters,
//                 shadow, overhangs -> EdgeOrders   (mpp.cpp:38-305, ol.cpp:31-101, src/main.cpp:328-414)
//   compact       per-edge order slots -> dense canonical order / id tables
//
// Everything is integer or IEEE fp64 VALU work (no MFMA: there is no contraction on this path).  The file is built
// with -ffp-contract=off: the reference's fp64 expressions are evaluated in its operation order, bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "msgpu.h"
#include "msgpu_internal.h"
#include "msgpu_device.h"

namespace msgpu {

// ---------------------------------------------------------------------------------------------------------------------
// generic exclusive scan (u32 in, T out), two launches; total written to *d_total
// ---------------------------------------------------------------------------------------------------------------------

constexpr int SCAN_ITEMS = 8; // per thread -> 2048 per block
uint32_t      scan_blocks(uint64_t n);

template <class T> __global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *in, uint64_t n, T *block_sums) {
  __shared__ T s[4];
  uint64_t     base = static_cast<uint64_t>(blockIdx.x) * 256 * SCAN_ITEMS;
  T            sum  = 0;
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    uint64_t idx = base + static_cast<uint64_t>(i) * 256 + threadIdx.x;
    if (idx < n) sum += in[idx];
  }
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <class T>
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *in, uint64_t n, const T *block_sums, T *out,
                                                    T *d_total) {
  // out has n+1 entries; out[n] = *d_total = total is written by the last block.  The carry of a block is the sum of
  // the (raw) sums of the blocks before it: a few thousand words at most, cheaper than a launch of its own.
  __shared__ T s_w[4];
  __shared__ T s_carry;
  uint64_t     base = static_cast<uint64_t>(blockIdx.x) * 256 * SCAN_ITEMS;
  const int    lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    T c = 0;
    for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 256) c += block_sums[i];
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    uint64_t idx = base + static_cast<uint64_t>(it) * 256 + threadIdx.x;
    T        v   = idx < n ? in[idx] : 0;
    T        inc = v;
    for (int d = 1; d < 64; d <<= 1) {
      T t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    T b = s_carry;
    for (int w = 0; w < wave; ++w) b += s_w[w];
    if (idx < n) out[idx] = b + inc - v;
    if (idx == n - 1) {
      out[n]   = b + inc;
      *d_total = b + inc;
    }
    __syncthreads();
    if (threadIdx.x == 255) s_carry = b + inc;
    __syncthreads();
  }
}

template <class T> void exclusive_scan(hipStream_t st, const uint32_t *in, uint64_t n, T *out, T *block_sums, T *d_total) {
  if (n == 0) {
    (void)hipMemsetAsync(out, 0, sizeof(T), st);
    (void)hipMemsetAsync(d_total, 0, sizeof(T), st);
    return;
  }
  uint32_t nb = static_cast<uint32_t>((n + 256 * SCAN_ITEMS - 1) / (256 * SCAN_ITEMS));
  hipLaunchKernelGGL(k_scan_reduce<T>, dim3(nb), dim3(256), 0, st, in, n, block_sums);
  hipLaunchKernelGGL(k_scan_apply<T>, dim3(nb), dim3(256), 0, st, in, n, block_sums, out, d_total);
}
template void exclusive_scan<uint32_t>(hipStream_t, const uint32_t *, uint64_t, uint32_t *, uint32_t *, uint32_t *);
template void exclusive_scan<uint64_t>(hipStream_t, const uint32_t *, uint64_t, uint64_t *, uint64_t *, uint64_t *);

// read-back without a copy engine and without a stream synchronisation: one wavefront writes the scalar block into
// mapped host memory, makes it visible system-wide and then publishes a sequence number the host is polling for
__global__ __launch_bounds__(64) void k_publish_scalars(const uint64_t *src, uint64_t *dst_host, uint32_t n, uint64_t seq) {
  if (threadIdx.x < n) dst_host[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&dst_host[n], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_publish_scalars(hipStream_t st, const uint64_t *src, uint64_t *dst_host, uint32_t n, uint64_t seq) {
  hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, st, src, dst_host, n, seq);
}

// Window cuts of the dispatcher by measured work: cum[r] = scaffold rows the owner reads before r visit (exclusive prefix of
// the index build's `visits`, cum[V] = all of them); thread j finds the first read whose prefix reaches target j.
__global__ __launch_bounds__(256) void k_window_cuts(const uint64_t *cum, uint32_t V, WindowCutArgs a, uint64_t *out /*[2 n + 1]*/) {
  const uint64_t total = cum[V];
  if (threadIdx.x == 0) out[2 * a.n] = total;
  if (threadIdx.x >= a.n) return;
  const uint64_t t  = static_cast<uint64_t>(static_cast<double>(a.frac[threadIdx.x]) * static_cast<double>(total));
  uint32_t       lo = 0, hi = V; // first index in [0, V] with cum[index] >= t
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (cum[mid] >= t) hi = mid;
    else lo = mid + 1;
  }
  out[threadIdx.x]       = lo;
  out[a.n + threadIdx.x] = cum[lo];
}
void launch_window_cuts(hipStream_t st, const uint64_t *cum, uint32_t V, const WindowCutArgs &a, uint64_t *out) {
  hipLaunchKernelGGL(k_window_cuts, dim3(1), dim3(256), 0, st, cum, V, a, out);
}

uint32_t scan_blocks(uint64_t n) { return static_cast<uint32_t>((n + 256 * SCAN_ITEMS - 1) / (256 * SCAN_ITEMS)); }

// ---------------------------------------------------------------------------------------------------------------------
// index build: the device-resident equivalent of Graph vertices + MatchMap::m_vertexMatches + m_scaffolds
// (Graph.cpp:148 first line wins; MatchMap.cpp:52-81 lowest line per (read, anchor) wins)
// ---------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_max_ids(const msgpu_row *rows, uint64_t n, uint32_t *max_ids /*[2]*/) {
  // grid-stride, one atomic pair per workgroup (a single word takes only ~88 atomics/us on MI355X)
  __shared__ uint32_t s_r[4], s_a[4];
  uint32_t            mr = 0, ma = 0;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * 256) {
    mr = max(mr, rows[i].read_id + 1);
    ma = max(ma, rows[i].anchor_id + 1);
  }
  for (int d = 32; d > 0; d >>= 1) {
    mr = max(mr, __shfl_down(mr, d));
    ma = max(ma, __shfl_down(ma, d));
  }
  if ((threadIdx.x & 63) == 0) {
    s_r[threadIdx.x >> 6] = mr;
    s_a[threadIdx.x >> 6] = ma;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&max_ids[0], max(max(s_r[0], s_r[1]), max(s_r[2], s_r[3])));
    atomicMax(&max_ids[1], max(max(s_a[0], s_a[1]), max(s_a[2], s_a[3])));
  }
}

// The row table's 28-byte link form (include/msgpu.h: msgpu_row28 / msgpu_packed_rows) back into the 40-byte rows the index
// build reads: read_len from the per-read table, line = row index + the delta of the row's run, the two flag bits out of the
// score's top bits.  256 rows per workgroup, both sides through LDS so that global loads and stores are whole dwords / 16-byte
// pieces of consecutive addresses (a 28-byte record is not a vector the memory system knows).
__global__ __launch_bounds__(256) void k_expand_rows(const uint32_t *pk, uint64_t n, const int32_t *read_len, uint32_t n_reads,
                                                     const uint32_t *run_start, const uint32_t *run_delta, uint32_t n_runs,
                                                     msgpu_row *out, uint32_t *err) {
  __shared__ uint32_t s_in[256 * 7];
  __shared__ __attribute__((aligned(16))) uint32_t s_out[256 * 10];
  __shared__ uint32_t s_k0;
  const uint64_t i0  = static_cast<uint64_t>(blockIdx.x) * 256;
  const uint32_t cnt = static_cast<uint32_t>(min(static_cast<uint64_t>(256), n - i0));
  for (uint32_t w = threadIdx.x; w < cnt * 7; w += 256) s_in[w] = pk[i0 * 7 + w];
  if (threadIdx.x == 0) { // the run of the workgroup's first row: last k with run_start[k] <= i0 (run_start[0] = 0)
    uint32_t lo = 0, hi = n_runs;
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (run_start[mid] <= i0) lo = mid;
      else hi = mid;
    }
    s_k0 = lo;
  }
  __syncthreads();
  if (threadIdx.x < cnt) {
    const uint64_t  i = i0 + threadIdx.x;
    const uint32_t *r = s_in + threadIdx.x * 7;
    uint32_t lo = s_k0, hi = min(n_runs, s_k0 + 257u); // at most 256 runs begin among the workgroup's other rows
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (run_start[mid] <= i) lo = mid;
      else hi = mid;
    }
    const uint32_t rd = r[1], sf = r[6];
    int32_t        len = 0;
    if (rd < n_reads) len = read_len[rd];
    else atomicOr(err, 2u); // (the index build reports it: an id outside the declared space)
    uint32_t *o = s_out + threadIdx.x * 10;
    o[0] = r[0];
    o[1] = rd;
    o[2] = static_cast<uint32_t>(len);
    o[3] = r[2];
    o[4] = r[3];
    o[5] = r[4];
    o[6] = r[5];
    o[7] = sf & 0x3fffffffu;
    o[8] = static_cast<uint32_t>(i) + run_delta[lo];
    o[9] = sf >> 30;
  }
  __syncthreads();
  static_assert(sizeof(msgpu_row) == 40, "ten dwords per row");
  uint4       *dst = reinterpret_cast<uint4 *>(out + i0); // (i0 is a multiple of 256: 16-byte aligned)
  const uint4 *src = reinterpret_cast<const uint4 *>(s_out);
  const uint32_t quads = cnt * 10 / 4, tail = cnt * 10 % 4;
  for (uint32_t q = threadIdx.x; q < quads; q += 256) dst[q] = src[q];
  if (threadIdx.x < tail) reinterpret_cast<uint32_t *>(out + i0)[quads * 4 + threadIdx.x] = s_out[quads * 4 + threadIdx.x];
}
void launch_expand_rows(hipStream_t st, const void *pk, uint64_t n, const int32_t *read_len, uint32_t n_reads, const uint32_t *run_start,
                        const uint32_t *run_delta, uint32_t n_runs, msgpu_row *out, uint32_t *err) {
  if (n)
    hipLaunchKernelGGL(k_expand_rows, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, st, static_cast<const uint32_t *>(pk), n,
                       read_len, n_reads, run_start, run_delta, n_runs, out, err);
}

// one launch instead of six memsets: zero / all-ones fill of the per-read and per-anchor tables of the index build
struct IndexInitArgs {
  uint32_t *zero[8];
  uint32_t  n_zero[8];
  uint32_t *ones[2];
  uint32_t  n_ones[2];
};
__global__ __launch_bounds__(256) void k_index_init(IndexInitArgs a) {
  const uint32_t stride = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    for (uint32_t i = t0; i < a.n_zero[k]; i += stride) a.zero[k][i] = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    for (uint32_t i = t0; i < a.n_ones[k]; i += stride) a.ones[k][i] = 0xffffffffu;
}

// pass 1 over the rows: per-read row counts, first line per read, and -- speculatively -- the scaffold offsets that
// hold when the table is already grouped by anchor with ascending lines (what a PAF from minimap2 looks like).
// With cap != 0 the same pass also buckets the rows by read: every read owns `cap` slots (bkt_row[rd * cap ..]), a row
// takes the slot its count atomic returns -- no offsets are needed, so the scan and the second pass over the table
// (k_scatter_read) fall away.  A read with more than cap rows raises IXF_OVERFLOW and the host rebuilds in two passes.
// Scaffolds are kept sorted by READ ID: read r then finds the partners it owns (read id > r, MatchMap.cpp:204-213) as the
// stretch behind its own row, and never looks at the others.  With the input grouped by anchor (fast mode) the rows of a
// scaffold sit next to each other: the scaffold starts `before` rows in front of row i, and row i belongs at the start
// plus the number of the scaffold's rows with a lower read id.  Both are counted in an LDS tile with SCAF_HALO rows of
// context on either side; a scaffold that does not fit the context raises IXF_BIGSCAF and the generic scaffold build
// runs instead.
#ifndef MSGPU_SCAF_HALO
#define MSGPU_SCAF_HALO 128
#endif
constexpr int SCAF_HALO = MSGPU_SCAF_HALO;
__global__ __launch_bounds__(256) void k_index_pass1(const msgpu_row *rows, uint64_t n, uint32_t *cnt_read,
                                                     uint32_t *anchor_first, uint32_t *flags, uint32_t V, uint32_t A,
                                                     uint32_t *err, IRow *bkt_row, uint32_t cap, uint2 *spos) {
  __shared__ uint32_t s_an[256 + 2 * SCAF_HALO], s_rd[256 + 2 * SCAF_HALO];
  const uint64_t      i0 = static_cast<uint64_t>(blockIdx.x) * 256;
  const uint64_t      i  = i0 + threadIdx.x;
  msgpu_row           row{};
  if (i < n) row = rows[i];
  // tile = (anchor, read) of this workgroup's rows plus SCAF_HALO rows of context on either side
  s_an[SCAF_HALO + threadIdx.x] = i < n ? row.anchor_id : 0xffffffffu; // no valid anchor id (ids are < A <= 2^32 - 1)
  s_rd[SCAF_HALO + threadIdx.x] = row.read_id;
  for (int t = threadIdx.x; t < 2 * SCAF_HALO; t += 256) {
    const int       slot = t < SCAF_HALO ? t : 256 + t;
    const long long g    = static_cast<long long>(i0) - SCAF_HALO + slot;
    uint2           ar   = make_uint2(0xffffffffu, 0u);
    if (g >= 0 && static_cast<uint64_t>(g) < n) ar = *reinterpret_cast<const uint2 *>(&rows[g]); // anchor_id, read_id
    s_an[slot] = ar.x;
    s_rd[slot] = ar.y;
  }
  __syncthreads();
  if (i >= n) return;
  const uint32_t rd = row.read_id, an = row.anchor_id, ln = row.line;
  if (rd >= V || an >= A) { // only possible when the host declared the id space (msgpu_set_id_space)
    atomicOr(err, 2u);
    return;
  }
  uint2 sp; // place in the scaffold, scaffold rows behind it
  {
    const int c = static_cast<int>(threadIdx.x) + SCAF_HALO;
    uint32_t  before = 0, lower = 0;
    int       d;
    for (d = 1; d <= SCAF_HALO && s_an[c - d] == an; ++d) {
      ++before;
      lower += s_rd[c - d] < rd ? 1u : 0u;
    }
    bool     big = d > SCAF_HALO;
    uint32_t after = 0;
    for (d = 1; d <= SCAF_HALO && s_an[c + d] == an; ++d) {
      ++after;
      lower += s_rd[c + d] < rd ? 1u : 0u;
    }
    big |= d > SCAF_HALO;
    if (big) atomicOr(flags, IXF_BIGSCAF);
    // place in the scaffold, and the number of scaffold rows behind it (= partners with a higher read id)
    sp = make_uint2(static_cast<uint32_t>(i) - before + lower, before + after - lower);
  }
  if (cap) {
    // one-pass build: the pair travels with the row, at the row's bucket slot -- the sort reads it next to the row
    // (a contiguous read per bucket) instead of gathering 8 bytes per row by source index
    const uint32_t pos = atomicAdd(&cnt_read[rd], 1u);
    if (pos < cap) {
      const uint64_t slot = static_cast<uint64_t>(rd) * cap + pos;
      store_irow(&bkt_row[slot], make_irow(row, an, static_cast<uint32_t>(i)));
      spos[slot] = sp;
    } else {
      atomicOr(flags, IXF_OVERFLOW);
    }
  } else {
    spos[i] = sp; // two-pass build: by source row
    atomicAdd(&cnt_read[rd], 1u);
  }
  if (i == 0) {
    anchor_first[an] = 0;
  } else {
    const uint32_t pa = s_an[threadIdx.x + SCAF_HALO - 1], pl = rows[i - 1].line;
    if (pa > an || (pa == an && pl >= ln)) atomicOr(flags, IXF_UNSORTED);
    if (pa != an) anchor_first[an] = static_cast<uint32_t>(i);
  }
}

__global__ __launch_bounds__(256) void k_check_anchor_first(uint32_t *anchor_first, uint32_t A, uint32_t n, uint32_t *flags) {
  uint32_t a = blockIdx.x * 256 + threadIdx.x;
  if (a > A) return;
  if (a == A) {
    anchor_first[A] = n;
    return;
  }
  if (anchor_first[a] == 0xffffffffu) atomicOr(flags, IXF_SPARSE);
}

// bucket rows by read: the whole row goes into the bucket (a fire-and-forget 32 B scatter), so the sort kernel reads
// its bucket with contiguous loads instead of gathering 40 B rows.  IRow.other = anchor, IRow.pf = flags | index of
// the source row (the 30 position bits are free until the sort writes the rank there; the loaders cap the table at
// 2^30 rows).
__global__ __launch_bounds__(256) void k_scatter_read(const msgpu_row *rows, uint64_t n, const uint32_t *read_off,
                                                      uint32_t *cursor, IRow *bkt_row) {
  uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const msgpu_row row = rows[i];
  const uint32_t  pos = read_off[row.read_id] + atomicAdd(&cursor[row.read_id], 1u);
  store_irow(&bkt_row[pos], make_irow(row, row.anchor_id, static_cast<uint32_t>(i)));
}


// Reads with 65..64*K rows: every lane keeps K rows in registers and every row is broadcast once (readlane), so a row
// costs 3 scalar reads + K compares per lane instead of a pass over the bucket in global memory.  Returns false (and
// writes nothing) when a (read, anchor) pair occurs twice: the caller then takes the generic path.
template <int K>
__device__ __forceinline__ bool sort_read_in_registers(uint32_t r, uint64_t bs, uint32_t b, uint32_t n, int lane, bool fast,
                                                       const IRow *bkt_row, IRow *by_read,
                                                       uint32_t *read_cnt, uint32_t *alive_rank, uint32_t *anchor_cnt,
                                                       IRow *by_anchor, const msgpu_row *rows, int32_t *read_len,
                                                       uint32_t *read_first, const uint2 *spos, uint4 *vis,
                                                       uint32_t *visits, bool by_slot) {
  IRow     row[K];
  uint32_t idx[K], man[K], less[K];
  int      mlo[K], mhi[K];
  bool     dup = false;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t e = static_cast<uint32_t>(k) * 64 + lane;
    row[k]           = IRow{};
    idx[k]           = 0xffffffffu;
    if (e < n) {
      row[k] = load_irow(&bkt_row[bs + e]);
      idx[k] = row[k].pf & PF_POS_MASK;
    }
    mlo[k]  = e < n ? row[k].n_lo : 0x7fffffff;
    mhi[k]  = e < n ? row[k].n_hi : 0x7fffffff;
    man[k]  = e < n ? row[k].other : 0xffffffffu;
    less[k] = 0;
  }
#pragma unroll
  for (int s = 0; s < K; ++s) {
    const int cnt = min(64, static_cast<int>(n) - 64 * s); // wave-uniform
    for (int t = 0; t < cnt; ++t) {
      const int      olo = rl_i32(mlo[s], t), ohi = rl_i32(mhi[s], t);
      const uint32_t oan = rl_u32(man[s], t);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        less[k] += key_less(olo, ohi, oan, mlo[k], mhi[k], man[k]) ? 1u : 0u;
        dup |= ((s != k) | (t != lane)) & (oan == man[k]);
      }
    }
  }
  {
    unsigned long long fk = ~0ull;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (static_cast<uint32_t>(k) * 64 + lane < n) fk = min(fk, (static_cast<unsigned long long>(row[k].line) << 32) | idx[k]);
    note_first_row(lane, fk, r, rows, read_len, read_first);
  }
  if (__ballot(dup)) return false; // sentinel rows (anchor 0xffffffff) are never broadcast, so they cannot match
  uint32_t behind = 0; // scaffold rows behind this read's own rows = the visits of its candidate scan
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (static_cast<uint32_t>(k) * 64 + lane < n) {
      row[k].pf = (row[k].pf & ~PF_POS_MASK) | less[k];
      if (fast) {
        IRow           w  = row[k];
        const uint2    sc = spos[by_slot ? bs + static_cast<uint32_t>(k) * 64 + lane : idx[k]];
        const uint32_t sp = sc.x;
        w.other           = r;
        store_irow(&by_anchor[sp], w);
        row[k].pf = (row[k].pf & ~PF_POS_MASK) | sp; // by_read rows carry their place in the scaffold
        vis[b + less[k]] = make_uint4(static_cast<uint32_t>(row[k].i_lo), static_cast<uint32_t>(row[k].i_hi), sp + 1, sc.y);
        behind += sc.y;
      }
      store_irow(&by_read[b + less[k]], row[k]);
      if (!fast) {
        alive_rank[idx[k]] = less[k];
        atomicAdd(&anchor_cnt[row[k].other], 1u);
      }
    }
  }
  behind = wave_sum(behind);
  if (lane == 0) {
    read_cnt[r] = n;
    visits[r]   = behind;
  }
  return true;
}

// One wavefront per read: MatchMap::addVertexMatch's lowest-line rule (MatchMap.cpp:64-80) + the rank of every alive
// row by (nanoporeRange, anchor id) = the order of mpp.cpp:164-172 / :259-267.  Reads with <= 256 rows sort in
// registers (readlane broadcast); longer ones loop over the bucket in global memory.
// Output: by_read rows (rank order), read_cnt, alive_rank[source row] (rank, or 0xffffffff for a dead row) and, in
// generic mode, the per-anchor alive counts.
__global__ __launch_bounds__(256) void k_sort_read(const uint32_t *read_off, const uint32_t *cnt_read, uint32_t V,
                                                   const IRow *bkt_row, IRow *by_read,
                                                   uint32_t *read_cnt, uint32_t *alive_rank, uint32_t *anchor_cnt,
                                                   uint8_t *bkt_dead, uint32_t *flags, IRow *by_anchor, uint32_t cap,
                                                   const msgpu_row *rows, int32_t *read_len, uint32_t *read_first,
                                                   uint32_t *err, const uint2 *spos, uint4 *vis, uint32_t *visits) {
  const int      lane = threadIdx.x & 63;
  const uint32_t r    = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (r >= V) return;
  // bucket of read r: dense at read_off[r] (two-pass build) or slot r of `cap` rows (one-pass build; a read with more
  // rows has raised IXF_OVERFLOW and everything written here is thrown away -- only stay inside the bucket)
  const uint32_t b  = read_off[r];
  const uint64_t bs = cap ? static_cast<uint64_t>(r) * cap : b;
  const uint32_t n  = cap ? min(cnt_read[r], cap) : cnt_read[r];
  if (n == 0) { // an id without any row: ids are not Registry-dense
    if (lane == 0) {
      read_len[r]   = 0;
      read_first[r] = 0xffffffffu;
      visits[r]     = 0;
      atomicOr(err, 1u);
    }
    return;
  }
  const bool fast = (*flags & ~IXF_DUPS) == 0; // decided by pass 1; a duplicate found later is reported to the host
  if (n <= 64) {
    const bool have = lane < static_cast<int>(n);
    IRow       row{};
    uint32_t   idx = 0xffffffffu;
    if (have) {
      row = load_irow(&bkt_row[bs + lane]);
      idx = row.pf & PF_POS_MASK;
    }
    note_first_row(lane, have ? (static_cast<unsigned long long>(row.line) << 32) | idx : ~0ull, r, rows, read_len, read_first);
    const int      mlo = have ? row.n_lo : 0x7fffffff, mhi = have ? row.n_hi : 0x7fffffff;
    const uint32_t man = have ? row.other : 0xffffffffu;
    // rank = rows with a smaller (n_lo, n_hi, anchor).  Branch-free, every predicate a wavefront mask: (n_lo, n_hi) as
    // one order-preserving 64-bit key (one compare for "less", one for "equal"), the anchor decides ties, the count goes
    // up through the carry; "same anchor on another row" collects in a scalar mask.
    const uint32_t klo = static_cast<uint32_t>(mhi) ^ 0x80000000u, khi = static_cast<uint32_t>(mlo) ^ 0x80000000u;
    const uint64_t mkey = (static_cast<uint64_t>(khi) << 32) | klo;
    uint32_t           less = 0;
    unsigned long long dupm = 0;
    for (int t = 0; t < static_cast<int>(n); ++t) {
      const uint64_t okey = (static_cast<uint64_t>(rl_u32(khi, t)) << 32) | rl_u32(klo, t);
      const uint32_t oan  = rl_u32(man, t);
      const unsigned long long lt = __ballot(okey < mkey), eq = __ballot(okey == mkey);
      const unsigned long long al = __ballot(oan < man), ae = __ballot(oan == man);
      // less += 1 in the lanes of the mask: the mask goes in as the carry of an add-with-carry (one instruction)
      asm("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(less) : "s"(lt | (eq & al)) : "vcc");
      dupm |= ae & ~(1ull << t);
    }
    const bool dup = (dupm >> lane) & 1ull;
    bool     alive  = have;
    uint32_t behind = 0; // scaffold rows behind this row = its visits in the candidate scan (fast mode)
    if (__ballot(dup && have)) { // rare: a (read, anchor) pair occurs more than once -- lowest line wins
      bool dead = false;
      for (int t = 0; t < static_cast<int>(n); ++t) {
        const uint32_t oan = rl_u32(man, t), oln = rl_u32(row.line, t), oix = rl_u32(idx, t);
        dead |= (t != lane) & (oan == man) & (oln < row.line || (oln == row.line && oix < idx));
      }
      alive = have && !dead;
      less  = 0;
      for (unsigned long long rem = __ballot(alive); rem; rem &= rem - 1) {
        const int t = __builtin_ctzll(rem);
        less += key_less(rl_i32(mlo, t), rl_i32(mhi, t), rl_u32(man, t), mlo, mhi, man) ? 1u : 0u;
      }
      if (lane == 0) atomicOr(flags, IXF_DUPS);
    }
    if (alive) {
      row.pf = (row.pf & ~PF_POS_MASK) | less;
      if (fast) {
        // input already grouped by anchor with ascending lines: the scaffold's rows are the input's, so this row's
        // by_anchor entry (same 32 bytes, `other` = the read, rank in the read attached) goes straight to the place
        // pass 1 worked out for it (scaffolds in read-id order).  A duplicate (read, anchor) pair found anywhere voids
        // the fast table: the host rebuilds generically.
        IRow           w  = row;
        const uint2    sc = spos[cap ? bs + lane : idx];
        const uint32_t sp = sc.x;
        w.other           = r;
        store_irow(&by_anchor[sp], w);
        row.pf = (row.pf & ~PF_POS_MASK) | sp; // by_read rows carry their place in the scaffold
        // what the candidate scan reads of this row: anchor interval + the stretch of the scaffold behind it (the
        // partners with a higher read id)
        vis[b + less] = make_uint4(static_cast<uint32_t>(row.i_lo), static_cast<uint32_t>(row.i_hi), sp + 1, sc.y);
        behind = sc.y;
      }
      store_irow(&by_read[b + less], row);
      if (!fast) {
        alive_rank[idx] = less;
        atomicAdd(&anchor_cnt[row.other], 1u);
      }
    } else if (have) {
      alive_rank[idx] = 0xffffffffu;
    }
    const unsigned long long alive_mask = __ballot(alive); // all lanes vote (not inside the lane-0 branch)
    behind = wave_sum(behind);
    if (lane == 0) {
      read_cnt[r] = static_cast<uint32_t>(__popcll(alive_mask));
      visits[r]   = behind;
    }
    return;
  }
  if (n <= 128) {
    if (sort_read_in_registers<2>(r, bs, b, n, lane, fast, bkt_row, by_read, read_cnt, alive_rank, anchor_cnt,
                                  by_anchor, rows, read_len, read_first, spos, vis, visits, cap != 0))
      return;
  } else if (n <= 256) {
    if (sort_read_in_registers<4>(r, bs, b, n, lane, fast, bkt_row, by_read, read_cnt, alive_rank, anchor_cnt,
                                  by_anchor, rows, read_len, read_first, spos, vis, visits, cap != 0))
      return;
  }
  // very long read, or one with a duplicated (read, anchor) pair: the bucket stays in global memory
  uint32_t           n_alive = 0, n_behind = 0;
  unsigned long long fk      = ~0ull;
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e = e0 + lane;
    bool           dead = false;
    if (e < n) {
      const IRow     k   = load_irow(&bkt_row[bs + e]);
      const uint32_t kix = k.pf & PF_POS_MASK;
      fk                 = min(fk, (static_cast<unsigned long long>(k.line) << 32) | kix);
      for (uint32_t q = 0; q < n; ++q) {
        const IRow o = load_irow(&bkt_row[bs + q]);
        if (q != e && o.other == k.other) dead |= o.line < k.line || (o.line == k.line && (o.pf & PF_POS_MASK) < kix);
      }
      bkt_dead[b + e] = dead ? 1 : 0;
      if (dead) alive_rank[kix] = 0xffffffffu;
    }
    if (__ballot(e < n && dead) && lane == 0) atomicOr(flags, IXF_DUPS);
  }
  note_first_row(lane, fk, r, rows, read_len, read_first);
  __threadfence(); // the dead flags are read back by other lanes of this wave
  for (uint32_t e0 = 0; e0 < n; e0 += 64) {
    const uint32_t e     = e0 + lane;
    const bool     alive = e < n && !bkt_dead[b + e];
    if (alive) {
      IRow           k    = load_irow(&bkt_row[bs + e]);
      const uint32_t kix  = k.pf & PF_POS_MASK;
      uint32_t       less = 0;
      for (uint32_t q = 0; q < n; ++q) {
        const IRow o = load_irow(&bkt_row[bs + q]);
        if (!bkt_dead[b + q]) less += key_less(o.n_lo, o.n_hi, o.other, k.n_lo, k.n_hi, k.other) ? 1u : 0u;
      }
      k.pf = (k.pf & ~PF_POS_MASK) | less;
      if (fast) {
        IRow           w  = k;
        const uint2    sc = spos[cap ? bs + e : kix];
        const uint32_t sp = sc.x;
        w.other           = r;
        store_irow(&by_anchor[sp], w);
        k.pf = (k.pf & ~PF_POS_MASK) | sp;
        vis[b + less] = make_uint4(static_cast<uint32_t>(k.i_lo), static_cast<uint32_t>(k.i_hi), sp + 1, sc.y);
        n_behind += sc.y;
      }
      store_irow(&by_read[b + less], k);
      if (!fast) {
        alive_rank[kix] = less;
        atomicAdd(&anchor_cnt[k.other], 1u);
      }
    }
    n_alive += static_cast<uint32_t>(__popcll(__ballot(alive)));
  }
  n_behind = wave_sum(n_behind);
  if (lane == 0) {
    read_cnt[r] = n_alive;
    visits[r]   = n_behind;
  }
}


// scaffold offsets: the speculative ones of pass 1 (fast) or the scan of the alive counts (generic)
__global__ __launch_bounds__(256) void k_select_anchor_off(const uint32_t *flags, const uint32_t *fast_off,
                                                           const uint32_t *gen_off, uint32_t A, uint32_t *anchor_off,
                                                           uint32_t *d_n_alive, uint32_t n_rows) {
  uint32_t   a    = blockIdx.x * 256 + threadIdx.x;
  const bool fast = (*flags & ~IXF_DUPS) == 0;
  if (a <= A) anchor_off[a] = fast ? fast_off[a] : gen_off[a];
  if (a == 0 && fast) *d_n_alive = n_rows;
}

// the two element-wise closings of an index build in one launch.  Over the reads: the Registry-order check on the first lines
// the sort found -- ids must follow first-line order (Registry.cpp:36-45); an id without any row (marked 0xffffffff by the
// sort) means the ids are not dense.  Over the anchors: k_select_anchor_off.
__global__ __launch_bounds__(256) void k_index_finish(const uint32_t *read_first, uint32_t V, uint32_t *err, const uint32_t *flags,
                                                      const uint32_t *fast_off, const uint32_t *gen_off, uint32_t A,
                                                      uint32_t *anchor_off, uint32_t *d_n_alive, uint32_t n_rows) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < V) {
    const uint32_t f0 = read_first[i], f1 = read_first[i + 1];
    if (f0 != 0xffffffffu && f1 != 0xffffffffu && f1 <= f0) atomicOr(err, 1u);
  }
  const bool fast = (*flags & ~IXF_DUPS) == 0;
  if (i <= A) anchor_off[i] = fast ? fast_off[i] : gen_off[i];
  if (i == 0 && fast) *d_n_alive = n_rows;
}

__global__ __launch_bounds__(256) void k_scatter_anchor(const msgpu_row *rows, uint64_t n, const uint32_t *alive_rank,
                                                        const uint32_t *anchor_off, uint32_t *cursor, uint32_t *bkt_idx,
                                                        uint32_t *bkt_line, const uint32_t *flags) {
  if ((*flags & ~IXF_DUPS) == 0) return; // fast mode: by_anchor already written
  uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  if (alive_rank[i] == 0xffffffffu) return;
  uint32_t a    = rows[i].anchor_id;
  uint32_t pos  = anchor_off[a] + atomicAdd(&cursor[a], 1u);
  bkt_idx[pos]  = static_cast<uint32_t>(i);
  bkt_line[pos] = rows[i].read_id; // the scaffold's sort key
}

// scaffold of each anchor sorted by read id (alive rows: a read occurs once per scaffold).  The reference sorts by line
// (MatchMap.cpp:178-183) only to name the outer match of a pair, which k_chain takes from the two rows' lines.  The
// by_read copy of the row learns its place in the scaffold.
__global__ __launch_bounds__(256) void k_rank_anchor(const uint32_t *anchor_off, const uint32_t *d_n_alive, const uint32_t *bkt_idx,
                                                     const uint32_t *bkt_key, const msgpu_row *rows,
                                                     const uint32_t *alive_rank, IRow *by_anchor, const uint32_t *flags,
                                                     const uint32_t *read_off, IRow *by_read, uint4 *vis) {
  if ((*flags & ~IXF_DUPS) == 0) return;
  uint64_t p = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (p >= *d_n_alive) return;
  uint32_t  idx = bkt_idx[p];
  msgpu_row row = rows[idx];
  uint32_t  b = anchor_off[row.anchor_id], e = anchor_off[row.anchor_id + 1];
  uint32_t  rank = 0;
  for (uint32_t q = b; q < e; ++q) rank += bkt_key[q] < row.read_id ? 1u : 0u;
  const uint32_t ar = alive_rank[idx], sp = b + rank;
  store_irow(&by_anchor[sp], make_irow(row, row.read_id, ar));
  const uint32_t at = read_off[row.read_id] + ar;
  uint32_t      *pf = &by_read[at].pf;
  *pf               = (*pf & ~PF_POS_MASK) | sp;
  vis[at] = make_uint4(static_cast<uint32_t>(row.i_lo), static_cast<uint32_t>(row.i_hi), sp + 1, e - (sp + 1));
}

// the scaffold rows a read has to visit = sum over its anchors of the rows behind its own (scaffolds are in read-id order)
__global__ __launch_bounds__(256) void k_bound(const uint32_t *read_off, const uint32_t *read_cnt, const uint4 *vis,
                                               uint32_t V, uint32_t shard, uint32_t nshards, uint32_t lo, uint32_t hi,
                                               uint32_t *bound) {
  // 16 lanes per read: neighbouring lanes read neighbouring rows
  const uint32_t r   = blockIdx.x * 16 + (threadIdx.x >> 4);
  const uint32_t sub = threadIdx.x & 15;
  uint32_t       s   = 0;
  if (r < V && r >= lo && r < hi && r % nshards == shard) { // owner reads of this shard / batch
    const uint32_t b = read_off[r], n = read_cnt[r];
    for (uint32_t j = sub; j < n; j += 16) s += vis[b + j].w; // the rows behind this read's own row in each scaffold
  }
  for (int d = 8; d > 0; d >>= 1) s += __shfl_xor(s, d); // every lane takes part
  if (r < V && sub == 0) bound[r] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// candidates: MatchMap::calculateEdges / processScaffold (MatchMap.cpp:161-224) from the owner read's point of view
// ---------------------------------------------------------------------------------------------------------------------
//
// The reference walks each anchor's scaffold and tests all pairs (inner < outer by line).  A pair of reads (x, y) that
// passes becomes an EdgeMatch of the edge whose first vertex is the read with the lower first line = the lower
// Registry id.  Here the workgroup of read v1 visits the scaffolds of v1's own anchors and keeps exactly the pairs
// (v1, v2) with v2 > v1 -- every pair is produced once, by its edge's first vertex, so all EdgeMatches of an edge are
// born in one workgroup and can be grouped in LDS.  The list is sorted by (v2, j) where j is the rank of the anchor
// in v1's (nanoporeRange, anchor) order, i.e. each edge comes out in the vStart order of mpp.cpp:164-172.

// NT threads per workgroup (= per owner read): 128 for the reads with few scaffold rows to visit -- twice as many reads in
// flight per CU, and the kernel waits on memory for most of its cycles -- 256 otherwise.
template <int R1MAX, int CMAX, int NT>
__global__ __launch_bounds__(NT) void k_candidates(CandArgs a, const CandDesc *desc, uint32_t n_list) {
  constexpr int      HSZ   = CMAX; // hash slots >= candidates: insertion always terminates
  constexpr int      HBITS = CMAX == 512 ? 9 : CMAX == 1024 ? 10 : CMAX == 2048 ? 11 : CMAX == 4096 ? 12 : 13;
  static_assert((1 << HBITS) == HSZ, "CMAX must be 512, 1024, 2048, 4096 or 8192");
  constexpr uint32_t EMPTY = 0xffffffffu;
  // LDS, with the tables of the later phases laid over those of the earlier ones (lifetimes separated by the
  // __syncthreads between the phases): fewer bytes per workgroup = more reads in flight per CU (the kernel waits on
  // memory for most of its cycles).
  //   region A: s_row (phases a-b)                            later g_rank   (phase c2 on)
  //   region B: g_slot, g_off (phase c2 on)
  static_assert(4 * R1MAX * 4 + 16 >= 2 * HSZ, "region A must hold g_rank");
  __shared__ __attribute__((aligned(16))) unsigned char s_regA[4 * R1MAX * 4 + 16];
  __shared__ __attribute__((aligned(16))) unsigned char s_regB[4 * CMAX + 16];
  // per row j of v1: {i_lo, i_hi, -, -}, the anchor interval on v1's side
  uint4 *const    s_row  = reinterpret_cast<uint4 *>(s_regA);
  uint16_t *const g_rank = reinterpret_cast<uint16_t *>(s_regA); // HSZ entries
  uint16_t *const g_slot = reinterpret_cast<uint16_t *>(s_regB);
  uint16_t *const g_off  = g_slot + CMAX; // CMAX + 1 entries
  __shared__ uint32_t s_t[CMAX];                // by_anchor row of candidate slot c, later (staging) of position pos
  __shared__ uint16_t s_j[CMAX], s_g[CMAX];     // row j of candidate slot c, later of staging position pos; group of pos
  __shared__ uint32_t h_key[HSZ], h_cnt[HSZ];   // open-addressing table v2 -> group; members per group
  __shared__ uint32_t s_bm[HSZ];                // per group (in v2 order): bitmap of the rows j it holds, R1MAX bits
  __shared__ uint32_t s_wave[NT / 64], s_ng;    // s_ng: groups created so far
  uint16_t *const     g_list = s_g;             // slots of the groups in creation order (s_g is idle until the staging path)

  if (blockIdx.x >= n_list) return;
  // everything the workgroup needs to know about its read in one 32-byte scalar load (k_classify_reads wrote it): the
  // read's row range and scratch offset used to be a second level of dependent loads behind the list entry
  const CandDesc dsc = desc[blockIdx.x];
  const uint32_t r = dsc.r, rb = dsc.rb, n1 = dsc.n1;
  const uint64_t co = dsc.co;
  const int      tid = threadIdx.x;
  for (int h = tid; h < HSZ; h += NT) {
    h_key[h] = EMPTY;
    h_cnt[h] = 0;
    s_bm[h]  = 0;
  }
  if (tid == 0) s_ng = 0;

  // (a) v1's rows (already in vStart order) and, per row, the stretch of its anchor's scaffold behind v1's own row
  constexpr int JPT = R1MAX / NT; // rows per thread, consecutive
  uint4         vr[JPT];
  uint32_t      tsum = 0;
#pragma unroll
  for (int q = 0; q < JPT; ++q) {
    const uint32_t j = tid * JPT + q;
    // anchor interval, first scaffold row behind v1's own (= first partner with a higher id), their number
    vr[q] = j < n1 ? a.vis[rb + j] : make_uint4(0, 0, 0, 0);
    tsum += vr[q].w;
  }
  // every visit x in [0, T) is one candidate slot: s_j[x] = the row of v1 it belongs to and s_t[x] = the scaffold row
  // to load (filled here, stretch by stretch), so the scan below is one scaffold row per lane with no search, no
  // per-row lane groups and no compaction (nearly every visit passes the overlap test now that only the owned partners
  // are visited; the few that fail keep an empty slot that the later phases skip).  A read of up to 64 rows (the rule)
  // is scanned by its first wavefront alone: no workgroup scan, no barrier before the one that publishes the slots.
  const bool one_wave = JPT == 1 && n1 <= 64; // workgroup-uniform
  uint32_t   T = 0, ex = 0;
  if (one_wave) {
    if (tid < 64) {
      const uint32_t inc = wave_incl_scan(tsum);
      ex                 = inc - tsum;
      if (tid == 63) s_wave[0] = inc;
    }
  } else {
    ex = block_excl_scan<NT>(tsum, s_wave, &T);
  }
#pragma unroll
  for (int q = 0; q < JPT; ++q) {
    const uint32_t j = tid * JPT + q;
    if (j < n1) {
      s_row[j] = make_uint4(vr[q].x, vr[q].y, 0u, 0u);
      for (uint32_t k = 0; k < vr[q].w; ++k) {
        s_j[ex + k] = static_cast<uint16_t>(j);
        s_t[ex + k] = vr[q].z + k;
      }
    }
    ex += vr[q].w;
  }
  __syncthreads();
  if (one_wave) T = s_wave[0];
  const uint32_t nc = T; // candidate slots (empty ones included)

  // (b) one scaffold row per lane: consecutive lanes read consecutive rows of a scaffold.  Slot c = tid + NT * q stays
  // with its thread through phase (c): v2, the scaffold row and j never leave the registers.
  constexpr int CPT = CMAX / NT;
  uint16_t      c_slot[CPT], c_li[CPT], c_j[CPT];
  uint32_t      c_t[CPT], c_v2[CPT];
  bool          c_ok[CPT]; // slot c holds a candidate (a visit that passed the overlap test)
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const uint32_t c = tid + NT * q;
    c_slot[q] = c_li[q] = c_j[q] = 0;
    c_t[q] = c_v2[q] = 0;
    c_ok[q]          = false;
    if (c < nc) {
      const uint32_t j  = s_j[c];
      const uint32_t vm = s_t[c];
      const IRow     o  = load_irow(&a.by_anchor[vm]);
      const uint4    rw = s_row[j]; // the anchor interval on v1's side: needed only once the scaffold row is here
      const int      ovlo = max(o.i_lo, static_cast<int>(rw.x)), ovhi = min(o.i_hi, static_cast<int>(rw.y));
      // overlap test (MatchMap.cpp:192); the owner rule (v2 > v1 <=> v1 has the lower first line, :204-213) is the
      // choice of the rows visited: scaffolds are in read-id order and the walk starts behind v1's own row
      c_ok[q] = ovlo <= ovhi && (ovhi - ovlo) > static_cast<int>(a.th_overlap);
      c_v2[q] = o.other;
      c_t[q]  = vm;
      c_j[q]  = static_cast<uint16_t>(j);
    }
  }

  // (c1) group by v2: open-addressing insert; the thread that creates a group numbers it (creation order, any);
  // li = arrival number inside the group (any order)
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    if (c_ok[q]) {
      const uint32_t v2 = c_v2[q];
      uint32_t       h  = (v2 * 2654435761u) >> (32 - HBITS);
      while (true) {
        const uint32_t old = atomicCAS(&h_key[h], EMPTY, v2);
        if (old == EMPTY) g_list[atomicAdd(&s_ng, 1u)] = static_cast<uint16_t>(h);
        if (old == EMPTY || old == v2) break;
        h = (h + 1) & (HSZ - 1);
      }
      c_slot[q] = static_cast<uint16_t>(h);
      c_li[q]   = static_cast<uint16_t>(atomicAdd(&h_cnt[h], 1u));
    }
  }
  __syncthreads();

  // (c2/c3) the groups (= edges) ranked by v2, and the first staging position of every group in that order.  A read
  // has a handful of partners: up to 64 groups are ranked by ONE wavefront with the keys and counts in registers
  // (v_readlane in a loop over the groups; rank and offset come out of the same loop), no scan and no LDS round trips
  // in the loop -- this phase used to be two workgroup scans and a loop of dependent LDS reads.
  const uint32_t ng = s_ng;
  if (ng <= 64) {
    if (tid < 64) { // wave 0
      const bool     have = static_cast<uint32_t>(tid) < ng;
      const uint32_t slot = have ? g_list[tid] : 0u;
      const uint32_t key = have ? h_key[slot] : EMPTY, cnt = have ? h_cnt[slot] : 0u;
      uint32_t       rank = 0, off = 0;
      for (uint32_t q = 0; q < ng; ++q) { // wave-uniform
        const uint32_t kq = rl_u32(key, static_cast<int>(q)), cq = rl_u32(cnt, static_cast<int>(q));
        const bool     lower = kq < key;
        rank += lower ? 1u : 0u;
        off += lower ? cq : 0u;
      }
      if (have) {
        g_rank[slot] = static_cast<uint16_t>(rank);
        g_slot[rank] = static_cast<uint16_t>(slot);
        g_off[rank]  = static_cast<uint16_t>(off);
        if (rank + 1 == ng) g_off[ng] = static_cast<uint16_t>(off + cnt); // candidates of this read
      }
    }
  } else { // many partners: the same ranking with the keys read from LDS
    for (uint32_t g = tid; g < ng; g += NT) {
      const uint32_t slot = g_list[g], key = h_key[slot];
      uint32_t       rank = 0, off = 0;
      for (uint32_t q = 0; q < ng; ++q) {
        const uint32_t sq = g_list[q];
        const bool     lower = h_key[sq] < key;
        rank += lower ? 1u : 0u;
        off += lower ? h_cnt[sq] : 0u;
      }
      g_rank[slot] = static_cast<uint16_t>(rank);
      g_slot[rank] = static_cast<uint16_t>(slot);
      g_off[rank]  = static_cast<uint16_t>(off);
      if (rank + 1 == ng) g_off[ng] = static_cast<uint16_t>(off + h_cnt[slot]);
    }
  }
  if (ng == 0 && tid == 0) g_off[0] = 0;
  __syncthreads();
  const uint32_t tot = g_off[ng];

  // (c4/c5) inside a group the candidates go in the order of j = the vStart order of mpp.cpp:164-172 (j is unique
  // inside a group).  One bitmap of R1MAX bits per group: every candidate sets
  // bit j of its group, then its position is the number of set bits below j -- a popcount or two per candidate
  // instead of a walk over the group's members (that walk was a quarter of this kernel's time).
  constexpr uint32_t WPG = R1MAX / 32; // bitmap words per group
  if (ng <= static_cast<uint32_t>(HSZ) / WPG) {
    uint32_t *const bm = s_bm; // zeroed with the hash table
    uint32_t        c_rk[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      c_rk[q] = 0;
      if (c_ok[q]) {
        c_rk[q] = g_rank[c_slot[q]];
        atomicOr(&bm[c_rk[q] * WPG + (c_j[q] >> 5)], 1u << (c_j[q] & 31));
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      if (c_ok[q]) {
        const uint32_t *g  = bm + c_rk[q] * WPG;
        const uint32_t  jw = c_j[q] >> 5;
        uint32_t        rr = static_cast<uint32_t>(__popc(g[jw] & ((1u << (c_j[q] & 31)) - 1u)));
        for (uint32_t w = 0; w < jw; ++w) rr += static_cast<uint32_t>(__popc(g[w]));
        const uint64_t dst = co + g_off[c_rk[q]] + rr;
        a.cand_j[dst]      = c_j[q];
        a.cand_t[dst]      = c_t[q];
      }
    }
  } else { // more groups than bitmaps fit: stage by group, rank by comparison
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      if (c_ok[q]) {
        const uint32_t rk  = g_rank[c_slot[q]];
        const uint32_t pos = g_off[rk] + c_li[q];
        s_j[pos]           = c_j[q];
        s_t[pos]           = c_t[q];
        s_g[pos]           = static_cast<uint16_t>(rk);
      }
    }
    __syncthreads();
    for (uint32_t pos = tid; pos < tot; pos += NT) {
      const uint32_t rk = s_g[pos], gs = g_off[rk], ge = g_off[rk + 1];
      const uint16_t mj = s_j[pos];
      uint32_t       rr = 0;
      for (uint32_t q = gs; q < ge; ++q) rr += (s_j[q] < mj) ? 1u : 0u;
      a.cand_j[co + gs + rr] = mj;
      a.cand_t[co + gs + rr] = s_t[pos];
    }
  }
  // (d) one edge per group (edges with more than 64 EdgeMatches are counted: they take k_chain_big)
  for (uint32_t g = tid; g < ng; g += NT) {
    a.edge_scr_v2[co + g]    = h_key[g_slot[g]];
    a.edge_scr_start[co + g] = g_off[g];
    const uint32_t cnt = static_cast<uint32_t>(g_off[g + 1]) - g_off[g];
    if (cnt > 64) {
      atomicAdd(&a.big_stats[0], 1ull);
      atomicAdd(&a.big_stats[1], static_cast<unsigned long long>(cnt));
    }
  }
  if (tid == 0) {
    a.n_cand[r] = tot;
    a.n_edge[r] = ng;
  }
}

template __global__ void k_candidates<256, 512, 256>(CandArgs, const CandDesc *, uint32_t);
template __global__ void k_candidates<256, 1024, 256>(CandArgs, const CandDesc *, uint32_t);
template __global__ void k_candidates<1024, 4096, 256>(CandArgs, const CandDesc *, uint32_t);

// classify reads of this shard by the LDS footprint their candidate scan needs
// classes by LDS footprint: 0 = <256 rows, 512 candidates> (half the LDS of class 1, so twice as many reads per CU),
// 1 = <256, 1024>, 2 = <1024, 4096>, 3 = global-scratch kernel
__global__ __launch_bounds__(1024) void k_classify_reads(const uint32_t *read_off, const uint32_t *read_cnt,
                                                         const uint32_t *bound, const uint64_t *cand_off, uint32_t V,
                                                         uint32_t shard, uint32_t nshards, uint32_t lo, uint32_t hi,
                                                         CandDesc *list0, CandDesc *list1, CandDesc *list2,
                                                         uint32_t *list3, uint32_t *n_lists /*[4]*/, CandZero z,
                                                         unsigned long long *own_total /*null, or: += visits of the reads classified*/) {
  // the four list cursors are single words (~88 atomics/us each): count inside the workgroup in LDS first, then one
  // global atomic per workgroup and class
  __shared__ uint32_t s_cnt[4], s_base[4];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t r   = blockIdx.x * 1024 + threadIdx.x;
  // What the candidate kernels ADD to starts from zero here -- this launch is in front of them on every path: the per-read
  // counts of the reads no workgroup will take (not owned, nothing to visit) and the block of big-edge statistics / cursors /
  // class counts in the scalars.
  if (r <= V) z.n_cand[r] = z.n_edge[r] = 0;
  if (blockIdx.x == 0 && threadIdx.x < z.n_scalar_words) z.scalar_words[threadIdx.x] = 0;
  int      cls = -1;
  uint32_t n1 = 0, bd = 0;
  if (r < V && r >= lo && r < hi && r % nshards == shard) {
    n1 = read_cnt[r];
    bd = bound[r];
    if (n1 != 0 && bd != 0)
      cls = (n1 <= 256 && bd <= 512) ? 0 : (n1 <= 256 && bd <= 1024) ? 1 : (n1 <= 1024 && bd <= 4096) ? 2 : 3;
  }
  const int lane  = threadIdx.x & 63;
  uint32_t  local = 0;
  if (own_total) { // (one atomic per workgroup: a single word takes ~88 atomics/us)
    __shared__ unsigned long long s_own[16];
    unsigned long long            own = cls >= 0 ? bd : 0u;
    for (int d = 32; d > 0; d >>= 1) own += __shfl_xor(own, d);
    if (lane == 0) s_own[threadIdx.x >> 6] = own;
    __syncthreads();
    if (threadIdx.x == 0) {
      own = 0;
      for (int w = 0; w < 16; ++w) own += s_own[w];
      if (own) atomicAdd(own_total, own);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned long long m = __ballot(cls == k);
    if (!m) continue; // wave-uniform
    uint32_t base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(&s_cnt[k], static_cast<uint32_t>(__popcll(m)));
    base = rl_u32(base, __builtin_ctzll(m));
    if (cls == k) local = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1)));
  }
  __syncthreads();
  if (threadIdx.x < 4 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&n_lists[threadIdx.x], s_cnt[threadIdx.x]);
  __syncthreads();
  if (cls == 3) {
    list3[s_base[3] + local] = r;
  } else if (cls >= 0) {
    CandDesc d;
    d.r     = r;
    d.rb    = read_off[r];
    d.n1    = n1;
    d.bound = bd;
    d.co    = cand_off[r];
    d.pad   = 0;
    (cls == 0 ? list0 : cls == 1 ? list1 : list2)[s_base[cls] + local] = d;
  }
}

// The closing launch of a bin-path index build (see IndexEpilogueArgs).  Thread i takes read i and anchor i.
//   reads:   ids must follow first-line order (Registry.cpp:36-45); an id without any row (0xffffffff from the sort) = not dense
//   anchors: the scaffold offsets are the speculative ones of pass 1; IXF_SPARSE when fewer scaffolds began than there are ids
//   scan:    cand_off = exclusive prefix of the visit counts -- the carry of a workgroup's 1024 reads is the sum of the buckets
//            in front of them (k_index_sort_bin left one sum per bucket of 16 reads), the rest a block scan: no scan launches
//   classes: k_classify_reads' body, with the offset straight out of the scan
//   read-back: the LAST workgroup to finish publishes the scalar block (flags, error bits, sizes, list lengths) and zeroes
//            what the next build counts from nothing
__global__ __launch_bounds__(1024) void k_index_epilogue(IndexEpilogueArgs a) {
  __shared__ uint32_t           s_cnt[4], s_base[4], s_wave[16], s_last;
  __shared__ unsigned long long s_carry[33];
  const uint32_t i    = blockIdx.x * 1024 + threadIdx.x;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  // What the LAST workgroup publishes must be done when a workgroup takes its ticket.  No fence for that (an agent-scope release
  // writes the L2 back: 40 us of this kernel's 49 in profiles/r5_04): every scalar is written with an atomic whose old value
  // comes BACK (so it has been performed where every CU sees it), collected in `dep`, which the ticket waits for.
  uint32_t dep = 0;
  // ---- reads: Registry order -------------------------------------------------------------------------------------------------
  if (i + 1 < a.V) {
    const uint32_t f0 = a.read_first[i], f1 = a.read_first[i + 1];
    if (f0 != 0xffffffffu && f1 != 0xffffffffu && f1 <= f0) dep |= atomicOr(a.err, 1u);
  }
  // ---- anchors: scaffold offsets (speculative / generic), the closing entry, sparse ids ------------------------------------------
  const uint32_t heads = *a.heads;
  if (i == 0 && heads != a.A) dep |= atomicOr(a.flags, IXF_SPARSE);
  const bool fast = (*a.flags & ~IXF_DUPS) == 0 && heads == a.A;
  if (i <= a.A) a.anchor_off[i] = fast ? (i == a.A ? a.n_rows : a.anchor_first[i]) : a.anchor_off_gen[i];
  if (i == 0 && fast) dep |= atomicExch(a.n_alive, a.n_rows);
  // ---- scan of the visit counts ------------------------------------------------------------------------------------------------
  unsigned long long carry = 0, all = 0;
  {
    constexpr uint32_t BPW = 1024u >> BIN_RPB_SHIFT; // buckets per workgroup
    const uint32_t     before = min(a.n_buckets, blockIdx.x * BPW);
    for (uint32_t b = threadIdx.x; b < a.n_buckets; b += 1024) {
      const unsigned long long v = a.bucket_visits[b];
      carry += b < before ? v : 0ull;
      all += v;
    }
    for (int d = 32; d > 0; d >>= 1) {
      carry += __shfl_xor(carry, d);
      all += __shfl_xor(all, d);
    }
    if (lane == 0) {
      s_carry[wave]      = carry;
      s_carry[16 + wave] = all;
    }
    __syncthreads();
    // (the sixteen partial sums are added by the first wavefront and handed on as one word: sixteen 64-bit words live in
    // every thread kept the kernel above 64 registers, one workgroup to a CU)
    if (wave == 0) {
      unsigned long long v = lane < 32 ? s_carry[lane] : 0ull; // lanes 0..15: carries, 16..31: totals
      for (int d = 8; d > 0; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_carry[32] = v;
      if (lane == 16 && blockIdx.x == 0) dep |= static_cast<uint32_t>(atomicExch(reinterpret_cast<unsigned long long *>(a.total), v));
    }
    __syncthreads();
    carry = s_carry[32];
  }
  const uint32_t vis = i < a.V ? a.visits[i] : 0u;
  uint32_t       blk_total;
  const uint32_t ex = block_excl_scan<1024>(vis, s_wave, &blk_total);
  const uint64_t co = carry + ex;
  if (i <= a.V) a.cand_off[i] = co;
  // ---- classification (k_classify_reads) --------------------------------------------------------------------------------------
  if (a.classify) {
    if (i <= a.V) a.z.n_cand[i] = a.z.n_edge[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < a.z.n_scalar_words) a.z.scalar_words[threadIdx.x] = 0;
    int      cls = -1;
    uint32_t n1  = 0;
    if (i < a.V && i % a.nshards == a.shard) {
      n1 = a.read_cnt[i];
      if (n1 != 0 && vis != 0) cls = (n1 <= 256 && vis <= 512) ? 0 : (n1 <= 256 && vis <= 1024) ? 1 : (n1 <= 1024 && vis <= 4096) ? 2 : 3;
    }
    uint32_t local = 0;
    if (a.nshards > 1) { // the visits of THIS shard's owner reads (without shards: the total, the host knows that)
      // one atomic per workgroup: a single word takes ~88 atomics/us, one per wavefront made this kernel five times as long
      unsigned long long own = cls >= 0 ? vis : 0u;
      for (int d = 32; d > 0; d >>= 1) own += __shfl_xor(own, d);
      if (lane == 0) s_carry[wave] = own;
      __syncthreads();
      if (threadIdx.x == 0) {
        own = 0;
        for (int w = 0; w < 16; ++w) own += s_carry[w];
        if (own) dep |= static_cast<uint32_t>(atomicAdd(a.own_total, own));
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long m = __ballot(cls == k);
      if (!m) continue; // wave-uniform
      uint32_t base = 0;
      if (lane == __builtin_ctzll(m)) base = atomicAdd(&s_cnt[k], static_cast<uint32_t>(__popcll(m)));
      base = rl_u32(base, __builtin_ctzll(m));
      if (cls == k) local = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1)));
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&a.n_lists[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if (cls == 3) {
      a.list3[s_base[3] + local] = i;
    } else if (cls >= 0) {
      CandDesc d;
      d.r     = i;
      d.rb    = a.read_off[i];
      d.n1    = n1;
      d.bound = vis;
      d.co    = co;
      d.pad   = 0;
      (cls == 0 ? a.list0 : cls == 1 ? a.list1 : a.list2)[s_base[cls] + local] = d;
    }
  }
  // ---- the last workgroup publishes --------------------------------------------------------------------------------------------
  asm volatile("" ::"v"(dep)); // (the atomics' old values have arrived: they are performed)
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(a.done, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < 64) {
    uint64_t v = 0;
    if (static_cast<uint32_t>(lane) < a.n_scalars) v = __hip_atomic_load(&a.scalars[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.host_scalars && a.seq) {
      if (static_cast<uint32_t>(lane) < a.n_scalars) a.host_scalars[lane] = v;
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&a.host_scalars[a.n_scalars], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // zero at rest: the error bits, the index flags and the counters of this build (the host has them, or reads the block with a
    // copy BEFORE this point never: the synchronising read-back path does not use this kernel's zeroing, see the host side)
    if ((a.zero_mask >> lane) & 1ull) a.scalars[lane] = 0;
    if (lane == 0) *a.row_base = 0;
  }
}
void launch_index_epilogue(hipStream_t st, const IndexEpilogueArgs &a) {
  const uint32_t m = (a.V > a.A ? a.V : a.A) + 1;
  hipLaunchKernelGGL(k_index_epilogue, dim3((m + 1023) / 1024), dim3(1024), 0, st, a);
}

// big reads: same algorithm with every staging array in global memory (slow path, any size)
__global__ __launch_bounds__(256) void k_candidates_big(CandArgs a, const uint32_t *read_list, uint32_t n_list,
                                                        uint64_t *big_key, uint32_t *big_t, uint32_t *big_r2s,
                                                        uint32_t *big_pfx) {
  __shared__ uint32_t s_wave[4], s_nc, s_carry;
  if (blockIdx.x >= n_list) return;
  const uint32_t r  = read_list[blockIdx.x];
  const uint32_t rb = a.read_off[r], n1 = a.read_cnt[r];
  const uint64_t co = a.cand_off[r];
  const uint64_t bo = co; // staging arrays share the candidate-scratch index space
  const int      tid = threadIdx.x;
  uint32_t      *pfx = big_pfx + rb; // n1 entries; the total stays in LDS
  if (tid == 0) {
    s_nc    = 0;
    s_carry = 0;
  }
  __syncthreads();
  // (a) exclusive prefix of scaffold sizes over v1's rows, chunks of 256
  for (uint32_t j0 = 0; j0 < n1; j0 += 256) {
    uint32_t j = j0 + tid, c = 0;
    if (j < n1) {
      const uint32_t an = a.by_read[rb + j].other, sp = a.by_read[rb + j].pf & PF_POS_MASK;
      c                 = a.anchor_off[an + 1] - (sp + 1);
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(c, s_wave, &tot);
    if (j < n1) pfx[j] = s_carry + ex;
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  const uint32_t T = s_carry;
  __threadfence_block();
  __syncthreads();
  // (b)
  for (uint32_t x0 = 0; x0 < T; x0 += 256) {
    uint32_t x    = x0 + tid;
    bool     pass = false;
    uint32_t j = 0, vm = 0, r2 = 0;
    if (x < T) {
      uint32_t lo = 0, hi = n1;
      while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (pfx[mid] <= x)
          lo = mid;
        else
          hi = mid;
      }
      j        = lo;
      IRow me  = load_irow(&a.by_read[rb + j]);
      vm       = (me.pf & PF_POS_MASK) + 1 + (x - pfx[j]);
      IRow o   = load_irow(&a.by_anchor[vm]);
      r2       = o.other;
      int ovlo = max(o.i_lo, me.i_lo), ovhi = min(o.i_hi, me.i_hi);
      pass     = ovlo <= ovhi && (ovhi - ovlo) > static_cast<int>(a.th_overlap);
    }
    unsigned long long m = __ballot(pass);
    if (m) {
      uint32_t base = 0;
      int      lane = tid & 63;
      if (lane == 0) base = atomicAdd(&s_nc, static_cast<uint32_t>(__popcll(m)));
      base = __shfl(base, 0);
      if (pass) {
        uint32_t c       = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1)));
        big_key[bo + c]  = (static_cast<uint64_t>(r2) << 32) | j;
        big_t[bo + c]    = vm;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  const uint32_t nc = s_nc;
  // (c)
  for (uint32_t c = tid; c < nc; c += 256) {
    uint64_t k    = big_key[bo + c];
    uint32_t rank = 0;
    for (uint32_t q = 0; q < nc; ++q) rank += (big_key[bo + q] < k) ? 1u : 0u;
    big_r2s[bo + rank]  = static_cast<uint32_t>(k >> 32);
    a.cand_j[co + rank] = static_cast<uint32_t>(k);
    a.cand_t[co + rank] = big_t[bo + c];
  }
  __threadfence_block();
  __syncthreads();
  // (d)
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < nc; i0 += 256) {
    uint32_t i  = i0 + tid;
    uint32_t fl = (i < nc && (i == 0 || big_r2s[bo + i] != big_r2s[bo + i - 1])) ? 1u : 0u;
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(fl, s_wave, &tot);
    if (fl) {
      a.edge_scr_v2[co + s_carry + ex]    = big_r2s[bo + i];
      a.edge_scr_start[co + s_carry + ex] = i;
    }
    __syncthreads();
    if (tid == 0) s_carry += tot;
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  {
    const uint32_t ne = s_carry;
    for (uint32_t g = tid; g < ne; g += 256) {
      const uint32_t st  = a.edge_scr_start[co + g];
      const uint32_t en  = g + 1 < ne ? a.edge_scr_start[co + g + 1] : nc;
      if (en - st > 64) {
        atomicAdd(&a.big_stats[0], 1ull);
        atomicAdd(&a.big_stats[1], static_cast<unsigned long long>(en - st));
      }
    }
  }
  if (tid == 0) {
    a.n_cand[r] = nc;
    a.n_edge[r] = s_carry;
  }
}

// Per chunk of CAND_CHUNK read ids: the sums of the reads' candidate and edge counts and a histogram of the chunk's edges by
// size (1..64 EdgeMatches; larger ones take k_chain_big and are counted by the candidate kernels), written as plain rows -- what
// k_emit_edges needs from ALL chunks before it can place anything.  (The candidate kernels adding to such rows with atomics, one
// per edge, cost them 50 us on BASELINE.json configs[2]: 1.2 M memory-side operations beside a kernel that is bound by exactly
// that pipe, profiles/r5_03.  This pass reads 4 bytes per edge out of L2 instead.)  16 lanes per read, as in k_emit_edges.
__global__ __launch_bounds__(1024) void k_cand_reduce(const uint32_t *n_edge, const uint32_t *n_cand, const uint64_t *cand_off,
                                                      const uint32_t *edge_scr_start, uint32_t V, unsigned long long *chunk_sums,
                                                      uint32_t *hist) {
  __shared__ uint32_t           s_h[64];
  __shared__ unsigned long long s_sum[16][2];
  const uint32_t chunk = blockIdx.x;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x & 15;
  if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long sc = 0, se = 0;
#pragma unroll
  for (uint32_t round = 0; round < CAND_CHUNK / 64; ++round) { // 64 reads per round, every load of a round in flight together
    const uint32_t r = chunk * CAND_CHUNK + round * 64 + (threadIdx.x >> 4);
    if (r >= V) continue;
    const uint32_t ne = n_edge[r];
    if (ne == 0) continue;
    const uint32_t nc = n_cand[r];
    const uint64_t co = cand_off[r];
    if (sub == 0) {
      sc += nc;
      se += ne;
    }
    for (uint32_t e = sub; e < ne; e += 16) {
      const uint32_t st = edge_scr_start[co + e], en = (e + 1 < ne) ? edge_scr_start[co + e + 1] : nc, cnt = en - st;
      if (cnt >= 1 && cnt <= 64) atomicAdd(&s_h[cnt - 1], 1u);
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    sc += __shfl_xor(sc, d);
    se += __shfl_xor(se, d);
  }
  if (lane == 0) {
    s_sum[wave][0] = sc;
    s_sum[wave][1] = se;
  }
  __syncthreads();
  if (threadIdx.x < 64) hist[static_cast<size_t>(chunk) * 64 + threadIdx.x] = s_h[threadIdx.x];
  if (threadIdx.x == 0) {
    unsigned long long tc = 0, te = 0;
    for (int w = 0; w < 16; ++w) {
      tc += s_sum[w][0];
      te += s_sum[w][1];
    }
    chunk_sums[2 * chunk]     = tc;
    chunk_sums[2 * chunk + 1] = te;
  }
}

// The candidate stage closes in ONE more launch: the scans of the per-read candidate / edge counts, the dense edge table (ascending
// (v1, v2)), the list of big edges, the edges of <= 64 EdgeMatches listed by size (largest first: the width classes of the
// chain kernels are contiguous stretches [64..33 | 32..17 | 16..9 | 8..1], the edges that share a wavefront in k_chain_sub have
// (nearly) the same size, the longest edges of a launch start first), the class sizes, and the read-back of all the sizes.
// A workgroup per chunk of CAND_CHUNK reads.  Every workgroup sums the rows k_cand_reduce left per chunk: the sums of the chunks before its own are its bases -- first edge, first EdgeMatch, and per size the first
// list position of the chunk's edges of that size -- the sums over all chunks the table sizes; workgroup 0 writes those into the
// scalar block and publishes it to the host at once (k_publish_scalars' protocol), the host turns around while the tables are
// written.  The launch may be speculative (into whatever the tables hold from earlier calls): if the edges or the list of big
// edges do not fit nothing is written; the host, which compares the same numbers, allocates and launches again.
__global__ __launch_bounds__(1024) void k_emit_edges(EmitArgs a) {
  __shared__ unsigned long long s_red[16][4], s_tot[4];
  __shared__ uint32_t           s_hb[16][64], s_ht[16][64], s_pos[64], s_w[2][4];
  __shared__ uint32_t           s_ne[CAND_CHUNK], s_nc[CAND_CHUNK], s_eb[CAND_CHUNK], s_mb[CAND_CHUNK];
  static_assert(CAND_CHUNK == 256, "the first four wavefronts hold a read per lane");
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t chunk = blockIdx.x;
  // (this thread's read, for the prefix inside the chunk: asked for now, used behind the sums below)
  uint32_t ne_r = 0, nc_r = 0;
  if (threadIdx.x < CAND_CHUNK) {
    const uint32_t r = chunk * CAND_CHUNK + threadIdx.x;
    ne_r             = r < a.V ? a.n_edge[r] : 0u;
    nc_r             = r < a.V ? a.n_cand[r] : 0u;
  }
  // the chain stage's chunk sums start from zero (this is the last launch in front of the chain kernels)
  for (uint32_t i = blockIdx.x * 1024 + threadIdx.x; i < a.n_chain_chunk_words; i += gridDim.x * 1024) a.chain_chunk_sums[i] = 0;
  // ---- sums over the chunks: candidates and edges before this chunk / in all chunks ---------------------------------------
  unsigned long long t[4] = {0, 0, 0, 0};
  for (uint32_t c = threadIdx.x; c < a.n_chunks; c += 1024) {
    const unsigned long long nc = a.chunk_sums[2 * c], ne = a.chunk_sums[2 * c + 1];
    if (c < chunk) {
      t[0] += nc;
      t[1] += ne;
    }
    t[2] += nc;
    t[3] += ne;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int d = 32; d > 0; d >>= 1) t[k] += __shfl_xor(t[k], d);
    if (lane == 0) s_red[wave][k] = t[k];
  }
  // ---- the size histograms: per size, the edges of the chunks before this one / of all chunks (a wavefront takes every
  // sixteenth chunk, a lane one size) -------------------------------------------------------------------------------------
  {
    uint32_t hb = 0, ht = 0;
    for (uint32_t c0 = wave; c0 < a.n_chunks; c0 += 64) { // four rows in flight per trip
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t c = c0 + 16 * u;
        v[u]             = c < a.n_chunks ? a.hist[static_cast<size_t>(c) * 64 + lane] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        hb += c0 + 16 * u < chunk ? v[u] : 0u;
        ht += v[u];
      }
    }
    s_hb[wave][lane] = hb;
    s_ht[wave][lane] = ht;
  }
  __syncthreads();
  // (the sixteen partial sums are added by the second wavefront -- the first one has the histogram below -- and handed to
  // everybody as four words: every thread adding all sixty-four of them itself cost the kernel 128 registers and spills)
  if (wave == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned long long v = lane < 16 ? s_red[lane][k] : 0ull;
      for (int d = 8; d > 0; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_tot[k] = v;
    }
  }
  uint32_t c8 = 0, c16 = 0, c32 = 0, c64 = 0;
  if (wave == 0) { // lane = size - 1: first list position of this chunk's edges of the size, sizes DESCENDING
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      before += s_hb[w][lane];
      total += s_ht[w][lane];
    }
    uint32_t inc = total;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_down(inc, d);
      if (lane + d < 64) inc += o;
    }
    s_pos[lane] = inc - total + before;
    c8 = lane < 8 ? total : 0, c16 = (lane >= 8 && lane < 16) ? total : 0, c32 = (lane >= 16 && lane < 32) ? total : 0, c64 = lane >= 32 ? total : 0;
    for (int d = 32; d > 0; d >>= 1) {
      c8 += __shfl_xor(c8, d);
      c16 += __shfl_xor(c16, d);
      c32 += __shfl_xor(c32, d);
      c64 += __shfl_xor(c64, d);
    }
  }
  __syncthreads();
  const unsigned long long base_m = s_tot[0], base_e = s_tot[1], tot_m = s_tot[2], tot_e = s_tot[3];
  if (chunk == 0 && threadIdx.x < 64) { // (wave 0: the class sizes are in its registers)
    // counts[0..3] = edges of 9..16, 17..32, 33..64, <= 8 EdgeMatches
    const unsigned long long cls_lo = static_cast<unsigned long long>(c16) | (static_cast<unsigned long long>(c32) << 32);
    const unsigned long long cls_hi = static_cast<unsigned long long>(c64) | (static_cast<unsigned long long>(c8) << 32);
    if (lane == 0) {
      a.scalars[a.slot_ems]     = tot_m;
      a.scalars[a.slot_edges]   = tot_e;
      a.scalars[a.slot_cls]     = cls_lo;
      a.scalars[a.slot_cls + 1] = cls_hi;
    }
    if (a.host_scalars && a.seq) {
      if (static_cast<uint32_t>(lane) < a.n_scalars) {
        const uint32_t k = lane;
        a.host_scalars[k] = k == a.slot_ems ? tot_m : k == a.slot_edges ? tot_e : k == a.slot_cls ? cls_lo : k == a.slot_cls + 1 ? cls_hi : a.scalars[k];
      }
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&a.host_scalars[a.n_scalars], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the list cursors of k_classify_reads are used up (the host read them before it launched the candidate kernels): zero
    // at rest for the next classification
    if (lane == 0 && a.nlists) a.nlists[0] = a.nlists[1] = 0;
  }
  static_assert(SC_PUBLISH_MAX <= 64, "one wavefront publishes the scalar block");
  if (tot_e > a.cap_edges || a.big_stats[0] >= a.cap_big) return;
  // ---- prefix inside the chunk: a read per thread of the first four wavefronts ------------------------------------------------
  uint32_t ie = 0, im = 0;
  if (threadIdx.x < CAND_CHUNK) {
    ie               = wave_incl_scan(ne_r);
    im               = wave_incl_scan(nc_r);
    if (lane == 63) {
      s_w[0][wave] = ie;
      s_w[1][wave] = im;
    }
  }
  __syncthreads();
  if (threadIdx.x < CAND_CHUNK) {
    uint32_t be = ie - ne_r, bm = im - nc_r;
    for (int w = 0; w < wave; ++w) {
      be += s_w[0][w];
      bm += s_w[1][w];
    }
    s_ne[threadIdx.x] = ne_r;
    s_nc[threadIdx.x] = nc_r;
    s_eb[threadIdx.x] = be;
    s_mb[threadIdx.x] = bm;
  }
  __syncthreads();
  // ---- the edges: 16 lanes per read (a read has ~24 edges), 64 reads per round, four rounds ---------------------------------------
  // The kernel waits on memory -- read's scratch offset -> its edges' (start, v2) -> the stores -- and a workgroup that took its
  // four rounds one after the other waited four times as long: every level of that chain is asked for in all four rounds
  // before the next level is touched (the first sixteen edges of a read; the few reads with more finish in the loop behind).
  constexpr int ROUNDS = CAND_CHUNK / 64;
  const int     sub = threadIdx.x & 15;
  uint32_t      ne_r4[ROUNDS], nc_r4[ROUNDS], st_r[ROUNDS], en_r[ROUNDS], v2_r[ROUNDS];
  uint64_t      co_r[ROUNDS];
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t lr = round * 64 + (threadIdx.x >> 4), r = chunk * CAND_CHUNK + lr;
    ne_r4[round] = s_ne[lr];
    nc_r4[round] = s_nc[lr];
    co_r[round]  = ne_r4[round] ? a.cand_off[r] : 0ull;
  }
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t e = sub;
    st_r[round] = en_r[round] = v2_r[round] = 0;
    if (e < ne_r4[round]) {
      st_r[round] = a.edge_scr_start[co_r[round] + e];
      en_r[round] = (e + 1 < ne_r4[round]) ? a.edge_scr_start[co_r[round] + e + 1] : nc_r4[round];
      v2_r[round] = a.edge_scr_v2[co_r[round] + e];
    }
  }
#pragma unroll
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t lr = round * 64 + (threadIdx.x >> 4), r = chunk * CAND_CHUNK + lr;
    const uint32_t ne = ne_r4[round];
    if (ne == 0) continue;
    const uint32_t nc = nc_r4[round];
    const uint64_t eb = base_e + s_eb[lr], mb = base_m + s_mb[lr], co = co_r[round];
    for (uint32_t e = sub; e < ne; e += 16) {
      const bool     first = e == static_cast<uint32_t>(sub);
      const uint32_t st = first ? st_r[round] : a.edge_scr_start[co + e];
      const uint32_t en = first ? en_r[round] : ((e + 1 < ne) ? a.edge_scr_start[co + e + 1] : nc);
      msgpu_edge     ed;
      ed.v1        = r;
      ed.v2        = first ? v2_r[round] : a.edge_scr_v2[co + e];
      ed.em_off    = mb + st;
      ed.order_off = 0;
      ed.em_cnt    = en - st;
      ed.order_cnt = 0;
      ed.shadow    = 0;
      ed.pad       = 0;
      a.edges[eb + e]     = ed;
      a.edge_cand[eb + e] = co + st;
      if (ed.em_cnt > 64) { // the few edges k_chain_big takes: listed here, with the first scratch element of each
        const unsigned long long i = atomicAdd(&a.big_cursor[0], 1ull);
        a.big_list[i] = static_cast<uint32_t>(eb + e);
        a.big_off[i]  = atomicAdd(&a.big_cursor[1], static_cast<unsigned long long>(ed.em_cnt));
      } else if (ed.em_cnt) {
        a.list[atomicAdd(&s_pos[ed.em_cnt - 1], 1u)] = static_cast<uint32_t>(eb + e);
      }
    }
  }
}
void launch_emit_edges(hipStream_t st, const EmitArgs &a, bool reduce) {
  if (reduce && a.n_chunks)
    hipLaunchKernelGGL(k_cand_reduce, dim3(a.n_chunks), dim3(1024), 0, st, a.n_edge, a.n_cand, a.cand_off, a.edge_scr_start, a.V,
                       const_cast<unsigned long long *>(a.chunk_sums), const_cast<uint32_t *>(a.hist));
  hipLaunchKernelGGL(k_emit_edges, dim3(a.n_chunks ? a.n_chunks : 1), dim3(1024), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// chain: EdgeMatch + getMaxPairwisePaths + chainingAndOverlaps filters + getOverlap, one wavefront per edge
// ---------------------------------------------------------------------------------------------------------------------

// The scan behind the chain kernels without a scan launch: every edge adds its order / id counts (and whether the shortcut
// took it) to the sums of its CHUNK of COMPACT_CHUNK consecutive edges -- two fire-and-forget atomics per edge on ~10^3
// counters (1 M edges: nothing beside the kernels' own work) -- and k_compact takes a workgroup per chunk: the chunk's base is
// the sum of the chunks before it (a few hundred words, summed by every workgroup), the prefix inside the chunk a block scan.
//   chunk_sums[2 c]     += shortcut | orders << 32        chunk_sums[2 c + 1] += ids
__device__ __forceinline__ void chunk_add(unsigned long long *chunk_sums, uint64_t e, bool clean, uint32_t n_orders, uint32_t n_ids) {
  unsigned long long *c = chunk_sums + 2 * (e / COMPACT_CHUNK);
  __hip_atomic_fetch_add(c, (clean ? 1ull : 0ull) | (static_cast<unsigned long long>(n_orders) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(c + 1, static_cast<unsigned long long>(n_ids), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PathRec {
  uint64_t mask;    // lanes (= EdgeMatches of the edge, vStart order) on the path
  uint64_t score;   // truncated like path_t's std::size_t
  uint32_t primary;
  uint32_t pad;
};

// Element of an edge as the compatibility test needs it: corrected + raw nanopore ranges on both vertices. 48 bytes,
// read from LDS with three 16-byte loads.
struct __attribute__((aligned(16))) ChainElem {
  double clo1, chi1;
  double clo2, chi2;
  int    rlo1, rhi1, rlo2, rhi2;
};

// nanoCheck (mpp.cpp:40-118) for one vertex: "1" = K (illuminaId1), "2" = L.  Returns abort; orientation/diff by ref.
// Branch-free restatement: the four (orientation, diff) cases of :67-91 are mutually exclusive, and
//   diff = (c1.second - c2.first) + 1   (o = +2)      (c2.first - c1.second) + 1 = -(c1.second - c2.first) + 1  (o = +1)
//   diff = (c2.second - c1.first) + 1   (o = -2)      (c1.first - c2.second) + 1 = -(c2.second - c1.first) + 1  (o = -1)
// (negation is exact in IEEE arithmetic, so the two subtractions below reproduce all four differences bit for bit).
// nanoCheck of checkCompatibility (mpp.cpp:67-109) for one vertex, for all pairs of a sweep step at once: every
// predicate is a wavefront mask in scalar registers (a v_cmp writes it directly, the logic is s_and / s_or) and only the
// selects of the difference use it as a lane condition again (inverse ballot = free).  pos = orientation +1 / +2,
// neg = -1 / -2, ovl = |orientation| is 2 (or the ranges overlap without an order: pos = neg = 0).
struct NanoMasks {
  unsigned long long pos, neg, ovl, abort_;
};
// WF ("well formed", checked once per edge, practically always true): no NaN and lo <= hi in every corrected range of
// the edge.  Then the overlap predicate is the sign of the one difference the diff needs anyway
//   k_clo < l_clo: k_clo < l_clo <= l_chi, so ovl = (l_clo <= k_chi) = (x >= 0);   else l_clo <= k_clo <= k_chi, ovl = (y >= 0)
// and the diff is |difference| + 1 (ovl keeps a non-negative difference, no-ovl negates a negative one): one compare
// and two selects less per vertex, bit for bit the same masks and the same diff.
// SORTED (vertex 1 only): the pairs run k < l in v1's (n_lo, n_hi, anchor) order, so k_rlo <= l_rlo <= l_rhi: the first
// half of the raw overlap test is true and "k behind l" (um2) is false -- three raw compares instead of six.
template <bool WF, int SORTED /*0: any order; 1: k_rlo <= l_rlo (and lo <= hi); 2: k_rlo >= l_rlo (and lo <= hi)*/>
__device__ __forceinline__ NanoMasks nano_check(double k_clo, double k_chi, double l_clo, double l_chi, int k_rlo,
                                                int k_rhi, int l_rlo, int l_rhi, double &d) {
  typedef unsigned long long M;
  const bool lt_lo_b = k_clo < l_clo;
  const M    lt_lo = __ballot(lt_lo_b), lt_hi = __ballot(k_chi < l_chi);
  const M    gt_lo = __ballot(k_clo > l_clo), gt_hi = __ballot(k_chi > l_chi);
  NanoMasks  f;
  // diff (:70-91): fwd -> k_chi - l_clo, bwd -> l_chi - k_clo, negated when the ranges do not overlap, + 1.  lt_lo
  // selects the right difference in every case that has an order.
  const double x = k_chi - l_clo;
  const double y = l_chi - k_clo;
  double       t = lt_lo_b ? x : y;
  if (WF) {
    f.ovl = __ballot(t >= 0.0);
    d     = fabs(t) + 1;
  } else {
    f.ovl = __ballot(k_clo <= l_chi) & __ballot(l_clo <= k_chi);
    t     = __builtin_amdgcn_inverse_ballot_w64(f.ovl) ? t : -t;
    // (the reference leaves diff = 0 when the ranges overlap without a strict order, orientation 0; checkCompatibility
    // reads the differences only when BOTH orientations are non-zero, :133-138, so that case needs no select here)
    d = t + 1;
  }
  // fwd2 | fwd1 = (ovl & lt_lo & lt_hi) | (~ovl & lt_lo),  bwd2 | bwd1 = (ovl & gt_lo & gt_hi) | (~ovl & ~lt_lo), written
  // with few scalar instructions (the loop keeps the scalar unit about half busy, the vector unit at 80-85 %)
  if (WF) {
    // lo <= hi on both sides: ranges that do not overlap and k starts first means k ends before l starts (k_chi < l_clo <=
    // l_chi), so lt_hi holds without asking for the overlap -- and the same for gt: pos = lt_lo & lt_hi, neg = gt_lo & gt_hi
    // cover orientation +-1 and +-2 alike (the scalar unit is as busy as the vector unit in this loop: four instructions less)
    f.pos = lt_lo & lt_hi;
    f.neg = gt_lo & gt_hi;
  } else {
    f.pos = lt_lo & (lt_hi | ~f.ovl);
    f.neg = (f.ovl & gt_lo & gt_hi) | ~(f.ovl | lt_lo);
  }
  // abort when the raw ranges overlap and their order contradicts the corrected orientation (:93-109)
  if (SORTED == 1) {
    const M rovl = __ballot(l_rlo <= k_rhi);
    const M u2   = __ballot(k_rlo < l_rlo) & __ballot(k_rhi < l_rhi);
    f.abort_     = rovl & (f.neg | (f.pos & ~u2));
  } else if (SORTED == 2) {
    // the mirror image (a reverse edge's second vertex): l_rlo <= k_rlo <= k_rhi, so the second half of the raw overlap test
    // is true and "k in front of l" (u2) is false
    const M rovl = __ballot(k_rlo <= l_rhi);
    const M um2  = __ballot(k_rlo > l_rlo) & __ballot(k_rhi > l_rhi);
    f.abort_     = rovl & (f.pos | (f.neg & ~um2));
  } else {
    const M rovl = __ballot(k_rlo <= l_rhi) & __ballot(l_rlo <= k_rhi);
    const M u2   = __ballot(k_rlo < l_rlo) & __ballot(k_rhi < l_rhi);
    const M um2  = __ballot(k_rlo > l_rlo) & __ballot(k_rhi > l_rhi);
    f.abort_     = rovl & ((f.neg & ~um2) | (f.pos & ~u2));
  }
  return f;
}

// Post-DP part of getMaxPairwisePaths (mpp.cpp:201-302) for the lanes `act` of one direction.
__device__ __forceinline__ int paths_of_direction(unsigned long long act, bool direction, int lane, double pop,
                                                  uint64_t pm, bool em_prim, uint32_t j1, uint32_t q2, uint32_t n1,
                                                  uint32_t n2, double alt_frac, PathRec *paths) {
  if (act == 0) return 0; // :150-152
  const bool mine = (act >> lane) & 1ull;
  // argmax, :201-210: strict > starting from 0.0, first maximum wins, iterator starts at begin().  The maximum alone is
  // reduced over the wavefront (one v_max_f64 per butterfly step; the scores are finite, so the hardware maximum is the
  // comparison's); the FIRST lane that holds it comes out of one ballot -- the index used to travel through the butterfly
  // with two compares and three selects per step.
  double best = mine ? pop : -1.0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) best = __builtin_fmax(best, __shfl_xor(best, d));
  double maxv;
  int    maxi;
  if (best > 0.0) {
    maxv = best;
    maxi = __builtin_ctzll(__ballot(mine && pop == best));
  } else {
    maxv = 0.0;
    maxi = __builtin_ctzll(act);
  }
  maxi = __builtin_amdgcn_readfirstlane(maxi);
  const unsigned long long prim_lanes = __ballot(mine && em_prim);
  int                      np         = 0;
  const uint64_t           m          = rl_u64(pm, maxi);
  bool                     hp         = (m & prim_lanes) != 0; // :217-219
  hp |= __popcll(m) > 2;                                       // :220
  if (lane == 0) {
    paths[0].mask    = m;
    paths[0].score   = static_cast<uint64_t>(maxv);
    paths[0].primary = hp;
  }
  np = 1;
  // alternatives, :223-249: population order, score > 0.75*max, id-disjoint from every accepted path
  const double       thr  = maxv * alt_frac;
  uint64_t           used = m;
  unsigned long long cand = __ballot(mine && pop > thr);
  while (true) {
    cand &= __ballot((pm & used) == 0); // a path that touches a used anchor can never become disjoint again
    if (!cand) break;
    const int      p  = __builtin_ctzll(cand);
    const uint64_t mp = rl_u64(pm, p);
    const double   sp = rl_f64(pop, p);
    if (lane == 0) {
      paths[np].mask    = mp;
      paths[np].score   = static_cast<uint64_t>(sp);
      paths[np].primary = (mp & prim_lanes) != 0;
    }
    ++np;
    used |= mp;
    cand &= ~(1ull << p);
  }
  // single primary result, :251-302
  if (np == 1 && hp) {
    const int first = __builtin_ctzll(m), last = 63 - __builtin_clzll(m);
    // position of each path anchor in v1's list is j1, in v2's (reversed when !direction) list is qe
    const uint32_t qe = direction ? q2 : (n2 - 1 - q2);
    const uint32_t jf = rl_u32(j1, first), jl = rl_u32(j1, last), qf = rl_u32(qe, first), ql = rl_u32(qe, last);
    bool           demote;
    if ((jf != 0 && qf != 0) || (jl != n1 - 1 && ql != n2 - 1)) {
      demote = true; // :272-274
    } else {
      // :280-296, all path anchors at once.  The scan walks the path in order; for anchor t it looks for t's position
      // in v1's list from i (always found: the path follows v1's order) and in v2's list from jj, and calls t
      // "interleaved" when BOTH searches skipped at least one foreign anchor; it stops at the first such t, so the
      // outcome is "any t interleaved" with i/jj evolving as if it never stopped:
      //   i_t  = j1(prev)+1                          -> inter1(t) = j1(t) > j1(prev)+1      (first: j1(t) > 0)
      //   jj_t = qe(prev)+1 while v2's order agrees  -> inter2(t) = qe(t) > qe(prev)+1      (first: qe(t) > 0)
      //   at the first t whose qe(t) < jj_t std::find_if runs off the end: re = n2, inter2 = n2 > jj_t, and from then
      //   on jj = n2+1 so no later anchor can be interleaved.
      const bool     on   = (m >> lane) & 1ull;
      const uint64_t below = m & ((1ull << lane) - 1);
      const int      prev  = below ? 63 - __builtin_clzll(below) : lane; // previous path anchor (self for the first)
      const uint32_t pj = __shfl(j1, prev), pq = __shfl(qe, prev);
      const bool     firstt = below == 0;
      const uint32_t i_t    = firstt ? 0u : pj + 1u;
      const uint32_t jj_t   = firstt ? 0u : pq + 1u;
      const bool     viol   = on && qe < jj_t; // v2's order disagrees here
      const unsigned long long vmask = __ballot(viol);
      const int      fv   = vmask ? __builtin_ctzll(vmask) : 64;
      const bool     i1   = j1 > i_t;
      const bool     i2   = lane < fv ? (qe > jj_t) : (lane == fv ? (n2 > jj_t) : false);
      demote              = __ballot(on && i1 && i2) != 0;
    }
    if (demote && lane == 0) paths[0].primary = 0;
  }
  return np;
}

// a ChainElem at an absolute LDS address (three 16-byte reads)
__device__ __forceinline__ ChainElem lds_elem(uint32_t addr) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const u32x4 lds_u4;
  lds_u4     *q = reinterpret_cast<lds_u4 *>(static_cast<uintptr_t>(addr));
  const u32x4 a = q[0], b = q[1], c = q[2];
  ChainElem   e;
  e.clo1 = __hiloint2double(static_cast<int>(a.y), static_cast<int>(a.x));
  e.chi1 = __hiloint2double(static_cast<int>(a.w), static_cast<int>(a.z));
  e.clo2 = __hiloint2double(static_cast<int>(b.y), static_cast<int>(b.x));
  e.chi2 = __hiloint2double(static_cast<int>(b.w), static_cast<int>(b.z));
  e.rlo1 = static_cast<int>(c.x);
  e.rhi1 = static_cast<int>(c.y);
  e.rlo2 = static_cast<int>(c.z);
  e.rhi2 = static_cast<int>(c.w);
  return e;
}
constexpr uint32_t CHAIN_LDS_EL = 0, CHAIN_LDS_CM = sizeof(ChainElem) * 64, CHAIN_LDS_BYTES = CHAIN_LDS_CM + 34 * 8;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_chain(ChainArgs a, const uint32_t *list, uint32_t n_list) {
  // ONE wavefront per workgroup, and no static LDS: the dynamic block starts at LDS address 0, so the element table's
  // address is a compile-time constant and the pair table can hold the LDS offsets of a pair's two elements ready to use --
  // the two address additions per sweep step (5 % of the loop's vector instructions) are gone.
  //   [0, 3072)     the 64 ChainElem of the sweep, later (the elements are dead once the compatibility masks exist) the path
  //                 lists of both directions in the same bytes
  //   [3072, 3344)  the verdicts of the pair sweep, one 64-bit wavefront mask per step (pairs p0 .. p0 + 63): at most 32 steps
  //                 (+ 2 words that the row extraction may touch past the last one)
  static_assert(sizeof(ChainElem) * 64 == sizeof(PathRec) * 2 * 64, "the path lists overlay the element table");
  extern __shared__ __attribute__((aligned(16))) unsigned char s_chain_lds[];
  const int      lane = threadIdx.x;
  // list == nullptr: wave i takes edge i; else the edges of the list (the 33..64 class of the size-sorted list)
  const uint32_t slot = blockIdx.x;
  if (list ? slot >= n_list : slot >= a.n_edges) return;
  const uint64_t e = list ? __builtin_amdgcn_readfirstlane(list[slot]) : slot;
  const msgpu_edge ed = a.edges[e];
  const uint32_t   n  = ed.em_cnt;
  if (n > 64) return; // handled by k_chain_big
  const uint64_t cp  = a.edge_cand[e];
  const bool     act = lane < static_cast<int>(n);
  const uint32_t v1 = ed.v1, v2 = ed.v2;
  const uint32_t n1 = a.read_cnt[v1], n2 = a.read_cnt[v2];
  const int      len1 = a.read_len[v1], len2 = a.read_len[v2];
  ChainElem     *el = reinterpret_cast<ChainElem *>(s_chain_lds + CHAIN_LDS_EL);
  uint64_t      *cm = reinterpret_cast<uint64_t *>(s_chain_lds + CHAIN_LDS_CM);

  // ---- per-lane element: VertexMatch on v1 (row j of v1), VertexMatch on v2 (scaffold row t), EdgeMatch ----------
  uint32_t j1 = 0, q2 = 0, anchor = 0;
  double   clo1 = 0, clo2 = 0, ovr1 = 0, ovr2 = 0, em_score = 0;
  bool     em_dir = false, em_prim = false;
  ChainElem x{};
  if (act) {
    j1               = a.cand_j[cp + lane];
    const uint32_t t = a.cand_t[cp + lane];
    const IRow m1 = load_irow(&a.by_read[a.read_off[v1] + j1]);
    const IRow m2 = load_irow(&a.by_anchor[t]);
    anchor        = m1.other;
    q2            = m2.pf & PF_POS_MASK;
    // EdgeMatch, MatchMap.cpp:188-202,218.  outer = the row with the higher line number.
    const int  ov_lo = max(m1.i_lo, m2.i_lo), ov_hi = min(m1.i_hi, m2.i_hi);
    const bool d1 = (m1.pf & PF_DIR) != 0, d2 = (m2.pf & PF_DIR) != 0;
    em_dir           = d1 == d2;
    em_prim          = (m1.pf & PF_PRIM) && (m2.pf & PF_PRIM);
    const bool   o1  = m1.line > m2.line;
    const IRow  &om = o1 ? m1 : m2, &im = o1 ? m2 : m1;
    const double ol  = static_cast<double>(om.i_hi - om.i_lo + 1);
    const double il  = static_cast<double>(im.i_hi - im.i_lo + 1);
    const double cl  = static_cast<double>(ov_hi - ov_lo + 1);
    const double os  = static_cast<double>(om.score) * cl / ol;
    const double is_ = static_cast<double>(im.score) * cl / il;
    em_score         = os + is_;
    {
      uint4 *q = reinterpret_cast<uint4 *>(&a.ems[ed.em_off + lane]);
      q[0]     = make_uint4(static_cast<uint32_t>(ov_lo), static_cast<uint32_t>(ov_hi),
                            static_cast<uint32_t>(__double_as_longlong(em_score)),
                            static_cast<uint32_t>(__double_as_longlong(em_score) >> 32));
      q[1]     = make_uint4(anchor, om.line, (em_dir ? 1u : 0u) | (em_prim ? 2u : 0u), static_cast<uint32_t>(e) + a.out_edge_base);
    }
    // corrected nanopore ranges (mpp.cpp:48-65) and overhangs (ol.cpp:37-47) on both vertices.  ov = [max lo, min hi]:
    // on each side at most ONE of the two rows has a non-zero correction numerator (the other one is 0 / rRatio = +0),
    // so one division per side serves both rows: (|difference of the anchor coordinates|) / (rRatio of the row that
    // reaches further out).  Six fp64 divisions per EdgeMatch instead of eight, bit for bit the same quotients.
    {
      const double rr1 = static_cast<double>(m1.i_hi - m1.i_lo + 1) / static_cast<double>(m1.n_hi - m1.n_lo + 1);
      const double rr2 = static_cast<double>(m2.i_hi - m2.i_lo + 1) / static_cast<double>(m2.n_hi - m2.n_lo + 1);
      const bool   l1  = m1.i_lo < m2.i_lo, l2 = m2.i_lo < m1.i_lo; // whose start lies left of the overlap
      const bool   h1  = m1.i_hi > m2.i_hi, h2 = m2.i_hi > m1.i_hi; // whose end lies right of the overlap
      const double qL  = static_cast<double>(l1 ? m2.i_lo - m1.i_lo : m1.i_lo - m2.i_lo) / (l1 ? rr1 : rr2);
      const double qR  = static_cast<double>(h1 ? m1.i_hi - m2.i_hi : m2.i_hi - m1.i_hi) / (h1 ? rr1 : rr2);
      double ncl1 = l1 ? qL : 0.0, ncr1 = h1 ? qR : 0.0, ncl2 = l2 ? qL : 0.0, ncr2 = h2 ? qR : 0.0;
      if (!d1) {
        const double tmp = ncl1;
        ncl1             = ncr1;
        ncr1             = tmp;
      }
      if (!d2) {
        const double tmp = ncl2;
        ncl2             = ncr2;
        ncr2             = tmp;
      }
      x.rlo1 = m1.n_lo;
      x.rhi1 = m1.n_hi;
      x.clo1 = static_cast<double>(m1.n_lo) + ncl1; // == overhangLeft
      x.chi1 = static_cast<double>(m1.n_hi) - ncr1;
      ovr1   = static_cast<double>(len1 - m1.n_hi) + ncr1;
      x.rlo2 = m2.n_lo;
      x.rhi2 = m2.n_hi;
      x.clo2 = static_cast<double>(m2.n_lo) + ncl2;
      x.chi2 = static_cast<double>(m2.n_hi) - ncr2;
      ovr2   = static_cast<double>(len2 - m2.n_hi) + ncr2;
    }
    clo1     = x.clo1;
    clo2     = x.clo2;
    el[lane] = x;
  }
  const unsigned long long m_plus = __ballot(act && em_dir), m_minus = __ballot(act && !em_dir);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- all-pairs-compatible shortcut -----------------------------------------------------------------------------
  // A true overlap is a strictly monotone chain of anchors on both reads.  If, for a one-direction edge,
  //   (1) corrected AND raw nanopore ranges are strictly increasing in lo and hi along v1's order, and strictly
  //       increasing (plus) / decreasing (minus) on v2  -> every pair has orientation +1/+2 on v1 and, after the flip
  //       of mpp.cpp:131, +1/+2 on v2, and nanoCheck never aborts (uco = +-2 whenever raw ranges overlap);
  //   (2) with the signed gaps g1 = c1lo(l) - c1hi(k), g2 = c2lo(l) - c2hi(k) (plus) or c2lo(k) - c2hi(l) (minus):
  //       diff = |g| + 1 in every case, so "same orientation" pairs need |g1 - g2| <= wiggle and mixed pairs
  //       (overlap on one read, gap on the other) need d1 + d2 = |g1 - g2| + 2 <= wiggle (:133-138); and
  //       g1 - g2 = a(l) - b(k) with a = c1lo -+ c2lo/c2hi, b = c1hi -+ c2hi/c2lo, so max a - min b and
  //       max b - min a bound every pair;
  // then checkCompatibility is true for ALL pairs, and with positive scores the DP of :185-199 takes k = l-1 at every
  // l: population[l] = population[l-1] + score(l), path = all anchors up to l.  The bound is checked in integers with
  // a margin of 3 (> every rounding error of the reference's fp64 expressions), so a "clean" verdict is exact; any
  // edge that is not provably clean takes the full pair sweep below.
  bool clean = false;
  if ((m_plus == 0 || m_minus == 0) && n >= 2 && a.fast_path) {
    const bool      plus = m_minus == 0;
    const ChainElem Pv   = el[lane > 0 ? lane - 1 : 0];
    bool            good = true;
    if (act && lane > 0) {
      good = (Pv.clo1 < x.clo1) & (Pv.chi1 < x.chi1) & (Pv.rlo1 < x.rlo1) & (Pv.rhi1 < x.rhi1);
      good &= plus ? ((Pv.clo2 < x.clo2) & (Pv.chi2 < x.chi2) & (Pv.rlo2 < x.rlo2) & (Pv.rhi2 < x.rhi2))
                   : ((Pv.clo2 > x.clo2) & (Pv.chi2 > x.chi2) & (Pv.rlo2 > x.rlo2) & (Pv.rhi2 > x.rhi2));
    }
    const double av = plus ? x.clo1 - x.clo2 : x.clo1 + x.chi2;
    const double bv = plus ? x.chi1 - x.chi2 : x.chi1 + x.clo2;
    good &= (av > -1.0e9) & (av < 1.0e9) & (bv > -1.0e9) & (bv < 1.0e9) & (em_score > 1.0e-6);
    if (__ballot(act && !good) == 0) { // the cheap monotonicity test first: most non-clean edges stop here
      int a_hi = act ? static_cast<int>(ceil(av)) : -2147483647 - 1;
      int a_lo = act ? static_cast<int>(floor(av)) : 2147483647;
      int b_hi = act ? static_cast<int>(ceil(bv)) : -2147483647 - 1;
      int b_lo = act ? static_cast<int>(floor(bv)) : 2147483647;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        a_hi = max(a_hi, __shfl_xor(a_hi, d));
        a_lo = min(a_lo, __shfl_xor(a_lo, d));
        b_hi = max(b_hi, __shfl_xor(b_hi, d));
        b_lo = min(b_lo, __shfl_xor(b_lo, d));
      }
      const long long W = static_cast<long long>(a.wiggle) - 3;
      clean = static_cast<long long>(a_hi) - b_lo <= W && static_cast<long long>(b_hi) - a_lo <= W;
    }
  }

  // ---- checkCompatibility (mpp.cpp:38-142) for every pair k < l of one direction, all 64 lanes busy ---------------
  // pair p = l(l-1)/2 + k, row-major over l; a lane handles p = it*64 + lane
  const int P = __builtin_amdgcn_readfirstlane(clean ? 0 : static_cast<int>(n * (n - 1) / 2)); // scalar loop control
  // Pair p: k | l << 8 | run << 16 | (64 - run - k) << 24 from a table that is the same for every edge (run = length of
  // the stretch of row l that starts at this lane of a 64-wide step, 0 if none starts here).  The table is padded with
  // (0, 1, 0) beyond the last pair and every (k, l) in it is < 64, so lanes past P read without a clamp or a branch and
  // are masked out.
  static_assert(sizeof(ChainElem) == 48, "the pair table holds byte offsets of 48-byte elements");
  // the pair table through a buffer descriptor: scalar offset (the step) + constant lane offset, no vector address
  // arithmetic in the loop
  const __amdgpu_buffer_rsrc_t tab_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.pair_tab64), 0, PAIR_TAB_STRIDE * 8, 0x00020000);
  const uint32_t tab_lane = static_cast<uint32_t>(lane) * 8u;
  typedef uint32_t pair_t __attribute__((ext_vector_type(2))); // {LDS offset of element k, of element l}
  auto           load_pairs = [&](int p0) __attribute__((always_inline)) -> pair_t {
    return __builtin_amdgcn_raw_buffer_load_b64(tab_rsrc, tab_lane, p0 * 8, 0);
  };
  // The scalar registers of this loop are all taken by pair masks; a loop-invariant scalar operand would be spilled and
  // re-read (v_readlane) in every step.  The one constant the common path compares with lives in a vector register.
  double wiggle = a.wiggle;
  asm volatile("" : "+v"(wiggle));
  // every element well formed (see nano_check) and v1's raw ranges in list order: the lean pair test; else the general one
  bool wf_lane = true;
  if (act) {
    const int prev_rlo1 = el[lane > 0 ? lane - 1 : 0].rlo1;
    wf_lane = (x.clo1 <= x.chi1) & (x.clo2 <= x.chi2) & (x.rlo1 <= x.rhi1) & (lane == 0 || prev_rlo1 <= x.rlo1);
  }
  const bool wf = __ballot(!wf_lane) == 0;
  // (The lean raw test for the SECOND vertex too -- its raw ranges in list order on forward edges, in reverse order on reverse
  // ones, nano_check<.., 1 / 2> -- was built and measured in round 5: on BASELINE.json's reads the +-15 bp jitter of the
  // alignment coordinates puts some neighbouring anchors of nearly every edge out of order on the second read, the instances
  // fired on next to no edge, and four more copies of the loop are four more loops in the instruction cache.  Not kept.)
  // DIR: 0 = every EdgeMatch forward, 1 = every EdgeMatch reverse (the flip of mpp.cpp:131 is then the same for all
  // pairs and costs nothing), 2 = both directions present (pairs of one direction only, flip per pair).  The loop is
  // bound by SCALAR issue (the mask algebra), so everything wave-uniform is decided outside it: six instances.
  // LDS address of the verdict word of the current PAIR of steps, kept in a vector register (a scalar one would be copied into a
  // vector register for every store)
  uint32_t cm_addr = CHAIN_LDS_CM;
  asm volatile("" : "+v"(cm_addr));
  auto sweep_step = [&](const int step, const pair_t kl, auto wft, auto dirt, auto s2t) __attribute__((always_inline)) {
    typedef decltype(wft)  WFT;
    typedef decltype(dirt) DIRT;
    typedef decltype(s2t)  S2T;
    constexpr int          DIR = DIRT::value;
    unsigned long long     bits = 0; // checkCompatibility(k, l) of the 64 pairs of this step
    { // every lane evaluates a pair: no divergence, all masks are wave-uniform
      typedef unsigned long long M;
      // lanes past the last pair evaluate the padding pair (0, 1) and have run = 0: their bits are never stored, so
      // one-direction edges need no "valid" mask at all
      M valid = ~0ull, KD = 0;
      if (DIR == 2) { // pairs of one direction only
        const int  k = static_cast<int>((kl.x - CHAIN_LDS_EL) / 48u), l = static_cast<int>((kl.y - CHAIN_LDS_EL) / 48u);
        const bool kd = (m_plus >> k) & 1ull, ld = (m_plus >> l) & 1ull;
        valid = __ballot(kd == ld);
        KD    = __ballot(kd);
      }
      // (the table holds LDS ADDRESSES -- the block starts at LDS address 0 -- used as they are: through the array's symbol the
      // compiler would add its link-time address, "+ 0", in every step)
      const ChainElem K = lds_elem(kl.x), L = lds_elem(kl.y);
      double          d1, d2;
      const NanoMasks f1 = nano_check<WFT::value, WFT::value ? 1 : 0>(K.clo1, K.chi1, L.clo1, L.chi1, K.rlo1, K.rhi1, L.rlo1, L.rhi1, d1);
      const NanoMasks f2 = nano_check<WFT::value, S2T::value>(K.clo2, K.chi2, L.clo2, L.chi2, K.rlo2, K.rhi2, L.rlo2, L.rhi2, d2);
      M p2, n2; // :131 flip by EdgeMatch(k).direction
      if (DIR == 0) {
        p2 = f2.pos;
        n2 = f2.neg;
      } else if (DIR == 1) {
        p2 = f2.neg;
        n2 = f2.pos;
      } else {
        p2 = (KD & f2.pos) | (~KD & f2.neg);
        n2 = (KD & f2.neg) | (~KD & f2.pos);
      }
      const M codir = (f1.pos & p2) | (f1.neg & n2); // :137 same sign
      const M mixed = f1.ovl ^ f2.ovl;               // :133 "equal and non-zero" fails: one overlaps, one does not
      M       cl    = codir & ~(f1.abort_ | f2.abort_);
      if (DIR == 2) cl &= valid;
      // :133-138 in one comparison: a pair whose orientations are equal needs |d1 - d2| <= wiggle (or the 15 % rule), a
      // mixed pair d1 + d2 <= wiggle -- so the second difference is negated where the pair is not mixed (exact: a - b is
      // a + (-b) bit for bit) and ONE sum serves both: |d1 +- d2| <= wiggle (d1 + d2 >= 2 is its own magnitude).  Three
      // scalar instructions and a compare less per step than testing both and choosing by mask: scalar and vector issue bound
      // this loop about equally.
      double val = d1 - d2;
      {
        M saved; // (the sum for the mixed pairs, written over the difference under their mask as EXEC: one vector instruction)
        asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                     "v_add_f64 %[val], %[x], %[y]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [val] "+v"(val), [sv] "=&s"(saved)
                     : [m] "s"(mixed), [x] "v"(d1), [y] "v"(d2)
                     : "scc");
      }
      const M pass = __ballot(fabs(val) <= wiggle);
      M            ok   = cl & pass;
      // the fp64 division of :136 only where the first test failed (rare for true overlaps): std::max(d1, d2) -
      // std::min(d1, d2) = |d1 - d2| bit for bit (a - b and b - a round to the same magnitude); the maximum itself is only
      // needed by the division
      const M need_div = (cl & ~mixed) & ~pass;
      if (need_div) {
        bool passd = false;
        if (__builtin_amdgcn_inverse_ballot_w64(need_div)) passd = fabs(val) * 100 / fmax(d1, d2) <= a.ratio_pct;
        ok |= __ballot(passd);
      }
      bits = ok;
    }
    // Every lane parks the step's 64 verdicts in the same word (no EXEC round trip for "lane 0 only": two scalar instructions
    // less); every lane cuts its own row out of the parked words after the sweep.  (The stretch of every row used to be shifted
    // into place and OR-ed into the row's word by its first lane in every step: eight vector instructions per step instead of
    // three.  The same change makes the sub-wavefront kernels slower -- their registers are the tighter resource -- so they keep
    // the per-step store.)  Lanes past the last pair evaluate the padding pair: their bits lie beyond every row.
    {
      typedef __attribute__((address_space(3))) unsigned long long lds_u64;
      *reinterpret_cast<lds_u64 *>(static_cast<uintptr_t>(cm_addr + (step & 1) * 8u)) = bits;
    }
  };
  auto sweep = [&](auto wft, auto dirt, auto s2t) __attribute__((always_inline)) {
    // two steps per trip, each with its own registers for the table entries: the next step's pairs are on their way
    // while this one computes, and nothing is copied from "next" to "current"
    pair_t ka = load_pairs(0);
    for (int p0 = 0; p0 < P; p0 += 128) {
      const pair_t kb = load_pairs(p0 + 64);
      sweep_step(p0 >> 6, ka, wft, dirt, s2t);
      if (p0 + 64 >= P) break;
      ka = load_pairs(p0 + 128);
      sweep_step((p0 >> 6) + 1, kb, wft, dirt, s2t);
      cm_addr += 16;
    }
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  // six instances: {well formed, general} x {forward, reverse, both directions}
  if (wf) {
    if (m_minus == 0) sweep(std::true_type{}, I0{}, I0{});
    else if (m_plus == 0) sweep(std::true_type{}, I1{}, I0{});
    else sweep(std::true_type{}, I2{}, I0{});
  } else {
    if (m_minus == 0) sweep(std::false_type{}, I0{}, I0{});
    else if (m_plus == 0) sweep(std::false_type{}, I1{}, I0{});
    else sweep(std::false_type{}, I2{}, I0{});
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // bit k of mycm: checkCompatibility(k, lane) for k < lane of the same direction = pair lane (lane - 1) / 2 + k: the row
  // starts at bit b of word w and may run on into word w + 1 (bits past the row, and words past the last step, are
  // masked off: they may hold anything)
  uint64_t mycm = 0;
  if (act && lane > 0 && P != 0) {
    const int      pr = lane * (lane - 1) / 2, w = pr >> 6, b = pr & 63;
    const uint64_t lo = cm[w] >> b, hi = (cm[w + 1] << 1) << (63 - b);
    mycm              = (lo | hi) & ((1ull << lane) - 1ull);
  }

  // ---- chaining DP (mpp.cpp:181-199), both directions at once: they never share a compatible pair -----------------
  double   pop = em_score;      // population[l].score
  uint64_t pm  = 1ull << lane;  // path of population[l] incl. l itself (self index appended at :203)
  if (clean) {
    // every pair compatible, scores positive: population[l] = population[l-1] + score(l), summed in the reference's
    // left-to-right order; path(l) = {0..l}
    for (int l = 1; l < static_cast<int>(n); ++l) {
      const double prev = rl_f64(pop, l - 1);
      if (lane == l) pop = prev + em_score;
    }
    pm = lane < 63 ? ((2ull << lane) - 1) : ~0ull;
  } else {
    // The loop carries only the score and the predecessor (9 vector instructions per step instead of 15 with the
    // 64-bit path mask): "k compatible with me" is the sign bit of the bit-reversed mask, shifted left once per step.
    uint32_t pred = static_cast<uint32_t>(lane);
    // bit k of mycm -> the sign bit after k shifts, in two 32-bit halves (k < 32, k >= 32): 32-bit compare and shift
    uint32_t rev = __builtin_bitreverse32(static_cast<uint32_t>(mycm));
    const int n1 = static_cast<int>(n) - 1, nlo = n1 < 32 ? n1 : 32;
    // Seven vector instructions per step (was ten): the compatibility bit and the shift are ONE add with carry-out (rev + rev:
    // the carry is the old sign bit, as a wavefront mask), and the three conditional moves (score: two, predecessor: one, the
    // step number copied into a vector register for it: one) are two plain moves under the condition as EXEC mask.
    auto dp_step = [&](int k) __attribute__((always_inline)) {
      const double       k_pop = rl_f64(pop, k);
      const double       cand  = k_pop + em_score; // :189
      unsigned long long comp, saved;
      asm volatile("v_add_co_u32 %0, %1, %0, %0" : "+v"(rev), "=s"(comp));
      const unsigned long long upd = comp & __ballot(cand > pop); // :190-197
      asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                   "v_mov_b64 %[pop], %[cand]\n\t"
                   "v_mov_b32 %[pred], %[k]\n\t"
                   "s_mov_b64 exec, %[sv]"
                   : [pop] "+v"(pop), [pred] "+v"(pred), [sv] "=&s"(saved)
                   : [m] "s"(upd), [cand] "v"(cand), [k] "s"(k)
                   : "scc"); // (s_and_saveexec writes SCC: without the clobber the loop's compare may sit in front of this block
                             // and its branch behind it -- an endless loop, gpurun call 535)
    };
    for (int k = 0; k < nlo; ++k) dp_step(k);
    rev = __builtin_bitreverse32(static_cast<uint32_t>(mycm >> 32));
    for (int k = 32; k < n1; ++k) dp_step(k);
    // population[l].path = population[pred].path + {l} with pred's path already final when l took it (pred < l): the
    // path masks are the closure of the predecessor pointers, built by pointer doubling in ceil(log2 n) rounds.
    uint32_t ptr = pred;
    for (uint32_t span = 1; span < n; span <<= 1) {
      const uint32_t lo = __shfl(static_cast<uint32_t>(pm), static_cast<int>(ptr));
      const uint32_t hi = __shfl(static_cast<uint32_t>(pm >> 32), static_cast<int>(ptr));
      const uint32_t p2 = __shfl(ptr, static_cast<int>(ptr));
      pm |= (static_cast<uint64_t>(hi) << 32) | lo;
      ptr = p2;
    }
  }

  // src/main.cpp:341-353: split by EdgeMatch direction, minus then plus
  PathRec  *pmn = reinterpret_cast<PathRec *>(s_chain_lds + CHAIN_LDS_EL), *ppl = pmn + 64; // el[] is dead from here on
  const int n_m = paths_of_direction(m_minus, false, lane, pop, pm, em_prim, j1, q2, n1, n2, a.alt_frac, pmn);
  const int n_p = paths_of_direction(m_plus, true, lane, pop, pm, em_prim, j1, q2, n1, n2, a.alt_frac, ppl);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // filters of src/main.cpp:355-387 on the union of both lists: lane p holds path p of each list
  bool     k_m = false, k_p = false;
  uint32_t prim_m = 0, prim_p = 0, len_m = 0, len_p = 0;
  if (lane < n_m) {
    prim_m = pmn[lane].primary;
    len_m  = static_cast<uint32_t>(__popcll(pmn[lane].mask));
    k_m    = true;
  }
  if (lane < n_p) {
    prim_p = ppl[lane].primary;
    len_p  = static_cast<uint32_t>(__popcll(ppl[lane].mask));
    k_p    = true;
  }
  const bool has_primary = __ballot((k_m && prim_m) || (k_p && prim_p)) != 0;
  if (has_primary) {
    k_m = k_m && prim_m;
    k_p = k_p && prim_p;
  }
  const bool has_multi = __ballot((k_m && len_m > 1) || (k_p && len_p > 1)) != 0;
  if (has_multi) {
    k_m = k_m && len_m > 1;
    k_p = k_p && len_p > 1;
  }
  const unsigned long long keep_m = __ballot(k_m), keep_p = __ballot(k_p);
  const int                combined = __popcll(keep_m) + __popcll(keep_p);
  bool                     shadow;
  if (combined > 1) {
    shadow = true; // :389-391
  } else {
    // the single remaining path: minus list first (:393)
    uint32_t pr = keep_m ? rl_u32(prim_m, __builtin_ctzll(keep_m)) : rl_u32(prim_p, __builtin_ctzll(keep_p));
    shadow      = !pr;
  }

  // getOverlap (ol.cpp:53-101) per kept path, minus first; orders into this edge's slots, ids into its id slots
  uint32_t n_orders = 0, n_ids = 0;
  for (int pass = 0; pass < 2; ++pass) {
    unsigned long long keep = pass == 0 ? keep_m : keep_p;
    const PathRec     *pv   = pass == 0 ? pmn : ppl;
    const bool         dir  = pass == 1;
    while (keep) {
      const int pi = __builtin_ctzll(keep);
      keep &= keep - 1;
      const uint64_t mask = pv[pi].mask;
      const int      f = __builtin_ctzll(mask), l = 63 - __builtin_clzll(mask);
      const double   L1 = rl_f64(clo1, f), R1 = rl_f64(ovr1, l);
      double         L2 = rl_f64(clo2, f), R2 = rl_f64(ovr2, l);
      if (!dir) { // :73-76
        L2 = rl_f64(ovr2, f);
        R2 = rl_f64(clo2, l);
      }
      bool     have = false;
      uint32_t fl   = 0;
      double   lo = 0, ro = 0;
      if (L1 <= L2 && R1 <= R2) {
        have = true;
        fl   = MSGPU_ORD_START_V1 | MSGPU_ORD_CONTAINED;
        lo   = L2 - L1;
        ro   = R2 - R1;
      } else if (L1 >= L2 && R1 >= R2) {
        have = true;
        fl   = MSGPU_ORD_CONTAINED;
        lo   = L1 - L2;
        ro   = R1 - R2;
      } else if (L1 > L2 && R1 < R2) {
        have = true;
        fl   = MSGPU_ORD_START_V1;
        lo   = L1 - L2;
        ro   = R2 - R1;
      } else if (L1 < L2 && R1 > R2) {
        have = true;
        fl   = 0;
        lo   = L2 - L1;
        ro   = R1 - R2;
      }
      if (!have) continue;
      const uint32_t cnt = static_cast<uint32_t>(__popcll(mask));
      // ids in path order = ascending lane
      if ((mask >> lane) & 1ull) {
        const uint32_t pos = static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1)));
        a.ids_scr[ed.em_off + n_ids + pos] = anchor;
      }
      if (lane == 0) {
        msgpu_order o;
        o.edge_idx     = static_cast<uint32_t>(e);
        o.flags        = fl | (dir ? MSGPU_ORD_DIR : 0u) | (pv[pi].primary ? MSGPU_ORD_PRIMARY : 0u);
        o.left_offset  = lo;
        o.right_offset = ro;
        o.score        = pv[pi].score;
        o.ids_off      = n_ids; // relative to the edge's id slots; made absolute by k_compact
        o.ids_cnt      = cnt;
        o.start        = (fl & MSGPU_ORD_START_V1) ? v1 : v2;
        o.end          = (fl & MSGPU_ORD_START_V1) ? v2 : v1;
        o.base         = v1;
        o.pad[0]       = 0;
        o.pad[1]       = 0;
        a.order_scr[ed.em_off + n_orders] = o;
      }
      ++n_orders;
      n_ids += cnt;
    }
  }
  if (lane == 0) {
    a.edge_norders[e] = n_orders;
    a.edge_nids[e]    = n_ids;
    a.edges[e].shadow = shadow ? 1 : 0;
    chunk_add(a.chunk_sums, e, clean, n_orders, n_ids);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// chain, short edges: G = 64 / W edges share a wavefront, W = 16 or 32 lanes each (k_chain_sub<W>).
//
// k_chain spends a fixed ~550 vector instructions per edge outside the pair sweep and the DP (the element phase with
// its six fp64 divisions, the path lists, the filters, getOverlap) whatever the edge's size, on 64 lanes of which an
// edge with n EdgeMatches uses n.  Here every quantity that is wave-uniform in k_chain (the edge, its masks, counters,
// the path being emitted) is uniform per GROUP of W lanes and lives in vector registers; wavefront masks (__ballot) are
// cut to the group's W bits, broadcasts are ds_bpermute shuffles inside the group, and loops whose trip count depends
// on the edge run to the largest count among the groups with the finished groups predicated off.  Same arithmetic in
// the same order as k_chain, so the same bits.  Edges are assigned through a list built by k_list_edges_by_size.
// ---------------------------------------------------------------------------------------------------------------------

struct __attribute__((aligned(16))) SubPath { // a path of one direction; mask bit i = the group's i-th EdgeMatch
  uint32_t mask, primary;
  uint64_t score;
};

__device__ __forceinline__ double shfl_f64(double v, int src) {
  const long long b  = __double_as_longlong(v);
  const uint32_t  lo = __shfl(static_cast<uint32_t>(b), src), hi = __shfl(static_cast<uint32_t>(b >> 32), src);
  return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo));
}

// the W bits of a wavefront mask that belong to this lane's group
template <int W> __device__ __forceinline__ uint32_t group_bits(unsigned long long m, int gbase) {
  return static_cast<uint32_t>(m >> gbase) & (W == 32 ? 0xffffffffu : ((1u << W) - 1u));
}

// paths_of_direction for a group of W lanes; returns the group's path count (group-uniform, 0 for an empty direction)
template <int W>
__device__ __forceinline__ uint32_t paths_of_direction_sub(uint32_t act, bool direction, int sl, int gbase, double pop,
                                                           uint32_t pm, bool em_prim, uint32_t j1, uint32_t q2,
                                                           uint32_t n1, uint32_t n2, double alt_frac, SubPath *paths) {
  const bool live = act != 0; // :150-152
  const bool mine = (act >> sl) & 1u;
  double     best = mine ? pop : -1.0; // argmax, :201-210 (as in paths_of_direction: the maximum by v_max_f64, its first lane by a ballot)
#pragma unroll
  for (int d = W / 2; d > 0; d >>= 1) best = __builtin_fmax(best, __shfl_xor(best, d));
  double maxv = 0.0;
  int    maxi = live ? __builtin_ctz(act) : 0;
  if (best > 0.0) {
    maxv = best;
    const uint32_t at = group_bits<W>(__ballot(mine && pop == best), gbase);
    maxi = at ? __builtin_ctz(at) : 0;
  }
  const uint32_t prim_lanes = group_bits<W>(__ballot(mine && em_prim), gbase);
  const uint32_t m          = __shfl(pm, gbase + maxi);
  const bool     hp         = ((m & prim_lanes) != 0) | (__popc(m) > 2); // :217-220
  if (live && sl == 0) paths[0] = SubPath{m, hp ? 1u : 0u, static_cast<uint64_t>(maxv)};
  uint32_t np = live ? 1u : 0u;
  // alternatives, :223-249
  const double thr  = maxv * alt_frac;
  uint32_t     used = m;
  uint32_t     cand = group_bits<W>(__ballot(mine && pop > thr), gbase);
  while (true) {
    cand &= group_bits<W>(__ballot((pm & used) == 0), gbase);
    if (__ballot(cand != 0) == 0) break; // no group has a candidate left
    const bool     has = cand != 0;
    const int      p   = has ? __builtin_ctz(cand) : 0;
    const uint32_t mp  = __shfl(pm, gbase + p);
    const double   sp  = shfl_f64(pop, gbase + p);
    if (has && sl == 0) paths[np] = SubPath{mp, (mp & prim_lanes) != 0 ? 1u : 0u, static_cast<uint64_t>(sp)};
    if (has) {
      ++np;
      used |= mp;
      cand &= ~(1u << p);
    }
  }
  // single primary result, :251-302 (evaluated by every group, applied where it holds)
  const bool     single = live && np == 1 && hp;
  const uint32_t ms     = m ? m : 1u;
  const int      first = __builtin_ctz(ms), last = 31 - __builtin_clz(ms);
  const uint32_t qe = direction ? q2 : (n2 - 1 - q2);
  const uint32_t jf = __shfl(j1, gbase + first), jl = __shfl(j1, gbase + last);
  const uint32_t qf = __shfl(qe, gbase + first), ql = __shfl(qe, gbase + last);
  const bool     ends   = (jf != 0 && qf != 0) || (jl != n1 - 1 && ql != n2 - 1); // :272-274
  const bool     on     = (m >> sl) & 1u;
  const uint32_t below  = m & ((1u << sl) - 1u);
  const int      prev   = below ? 31 - __builtin_clz(below) : sl;
  const uint32_t pj = __shfl(j1, gbase + prev), pq = __shfl(qe, gbase + prev);
  const bool     firstt = below == 0;
  const uint32_t i_t = firstt ? 0u : pj + 1u, jj_t = firstt ? 0u : pq + 1u;
  const bool     viol  = on && qe < jj_t;
  const uint32_t vmask = group_bits<W>(__ballot(viol), gbase);
  const int      fv    = vmask ? __builtin_ctz(vmask) : 64;
  const bool     i1    = j1 > i_t;
  const bool     i2    = sl < fv ? (qe > jj_t) : (sl == fv ? (n2 > jj_t) : false);
  const bool     inter = group_bits<W>(__ballot(on && i1 && i2), gbase) != 0;
  if (single && (ends || inter) && sl == 0) paths[0].primary = 0;
  return np;
}

// The body of k_chain_sub<W> for workgroup `block` of its class (the kernels below supply the LDS: the same two arrays for
// every width).
template <int W>
__device__ __forceinline__ void chain_sub_body(const ChainArgs &a, const uint32_t *list, uint32_t n_list, uint32_t block,
                                               unsigned char (*s_wavebuf)[sizeof(ChainElem) * 64], uint32_t (*s_cm)[64],
                                               double (*s_keep)[64][4], uint32_t (*s_anchor)[64]) {
  constexpr int G = 64 / W;
  static_assert(W == 8 || W == 16 || W == 32, "group width");
  static_assert(sizeof(ChainElem) * 64 >= sizeof(SubPath) * 2 * 64, "the path lists overlay the element table");
  const int      wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int      g = lane / W, sl = lane % W, gbase = g * W;
  const uint32_t slot    = (block * 4 + wave) * G + g;
  const bool     valid_g = slot < n_list;
  if (__ballot(valid_g) == 0) return;
  const uint32_t   e  = valid_g ? list[slot] : list[0];
  const msgpu_edge ed = a.edges[e];
  const uint32_t   n  = (valid_g && ed.em_cnt <= static_cast<uint32_t>(W)) ? ed.em_cnt : 0u;
  const uint64_t   cp = a.edge_cand[e];
  const bool       act = sl < static_cast<int>(n);
  const uint32_t   v1 = ed.v1, v2 = ed.v2;
  const uint32_t   n1 = a.read_cnt[v1], n2 = a.read_cnt[v2];
  const int        len1 = a.read_len[v1], len2 = a.read_len[v2];
  ChainElem       *el = reinterpret_cast<ChainElem *>(s_wavebuf[wave]);
  uint32_t        *cm = s_cm[wave];

  // ---- per-lane element (as in k_chain) ------------------------------------------------------------------------------
  uint32_t  j1 = 0, q2 = 0, anchor = 0;
  double    clo1 = 0, clo2 = 0, ovr1 = 0, ovr2 = 0, em_score = 0;
  bool      em_dir = false, em_prim = false;
  ChainElem x{};
  cm[lane] = 0;
  if (act) {
    j1               = a.cand_j[cp + sl];
    const uint32_t t = a.cand_t[cp + sl];
    const IRow m1 = load_irow(&a.by_read[a.read_off[v1] + j1]);
    const IRow m2 = load_irow(&a.by_anchor[t]);
    anchor        = m1.other;
    q2            = m2.pf & PF_POS_MASK;
    const int  ov_lo = max(m1.i_lo, m2.i_lo), ov_hi = min(m1.i_hi, m2.i_hi);
    const bool d1 = (m1.pf & PF_DIR) != 0, d2 = (m2.pf & PF_DIR) != 0;
    em_dir           = d1 == d2;
    em_prim          = (m1.pf & PF_PRIM) && (m2.pf & PF_PRIM);
    const bool   o1  = m1.line > m2.line;
    const IRow  &om = o1 ? m1 : m2, &im = o1 ? m2 : m1;
    const double ol  = static_cast<double>(om.i_hi - om.i_lo + 1);
    const double il  = static_cast<double>(im.i_hi - im.i_lo + 1);
    const double cl  = static_cast<double>(ov_hi - ov_lo + 1);
    const double os  = static_cast<double>(om.score) * cl / ol;
    const double is_ = static_cast<double>(im.score) * cl / il;
    em_score         = os + is_;
    {
      uint4 *q = reinterpret_cast<uint4 *>(&a.ems[ed.em_off + sl]);
      q[0]     = make_uint4(static_cast<uint32_t>(ov_lo), static_cast<uint32_t>(ov_hi),
                            static_cast<uint32_t>(__double_as_longlong(em_score)),
                            static_cast<uint32_t>(__double_as_longlong(em_score) >> 32));
      q[1]     = make_uint4(anchor, om.line, (em_dir ? 1u : 0u) | (em_prim ? 2u : 0u), e + a.out_edge_base);
    }
    {
      const double rr1 = static_cast<double>(m1.i_hi - m1.i_lo + 1) / static_cast<double>(m1.n_hi - m1.n_lo + 1);
      const double rr2 = static_cast<double>(m2.i_hi - m2.i_lo + 1) / static_cast<double>(m2.n_hi - m2.n_lo + 1);
      const bool   l1  = m1.i_lo < m2.i_lo, l2 = m2.i_lo < m1.i_lo;
      const bool   h1  = m1.i_hi > m2.i_hi, h2 = m2.i_hi > m1.i_hi;
      const double qL  = static_cast<double>(l1 ? m2.i_lo - m1.i_lo : m1.i_lo - m2.i_lo) / (l1 ? rr1 : rr2);
      const double qR  = static_cast<double>(h1 ? m1.i_hi - m2.i_hi : m2.i_hi - m1.i_hi) / (h1 ? rr1 : rr2);
      double ncl1 = l1 ? qL : 0.0, ncr1 = h1 ? qR : 0.0, ncl2 = l2 ? qL : 0.0, ncr2 = h2 ? qR : 0.0;
      if (!d1) {
        const double tmp = ncl1;
        ncl1             = ncr1;
        ncr1             = tmp;
      }
      if (!d2) {
        const double tmp = ncl2;
        ncl2             = ncr2;
        ncr2             = tmp;
      }
      x.rlo1 = m1.n_lo;
      x.rhi1 = m1.n_hi;
      x.clo1 = static_cast<double>(m1.n_lo) + ncl1;
      x.chi1 = static_cast<double>(m1.n_hi) - ncr1;
      ovr1   = static_cast<double>(len1 - m1.n_hi) + ncr1;
      x.rlo2 = m2.n_lo;
      x.rhi2 = m2.n_hi;
      x.clo2 = static_cast<double>(m2.n_lo) + ncl2;
      x.chi2 = static_cast<double>(m2.n_hi) - ncr2;
      ovr2   = static_cast<double>(len2 - m2.n_hi) + ncr2;
    }
    clo1     = x.clo1;
    clo2     = x.clo2;
    el[lane] = x;
  }
  // the four numbers the overhangs at the very end are made of wait in LDS, not in eight registers across the sweep and the DP
  // (the element table itself is overwritten by the path lists before then)
  {
    double *kp = s_keep[wave][lane];
    kp[0] = clo1, kp[1] = clo2, kp[2] = ovr1, kp[3] = ovr2;
    s_anchor[wave][lane] = anchor; // (the ids at the very end)
  }
  const uint32_t m_plus  = group_bits<W>(__ballot(act && em_dir), gbase);
  const uint32_t m_minus = group_bits<W>(__ballot(act && !em_dir), gbase);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- all-pairs-compatible shortcut (see k_chain) -------------------------------------------------------------------
  bool clean = false;
  {
    const bool      try_clean = (m_plus == 0 || m_minus == 0) && n >= 2 && a.fast_path;
    const bool      plus      = m_minus == 0;
    const ChainElem Pv        = el[sl > 0 ? lane - 1 : lane];
    bool            good      = true;
    if (act && sl > 0) {
      good = (Pv.clo1 < x.clo1) & (Pv.chi1 < x.chi1) & (Pv.rlo1 < x.rlo1) & (Pv.rhi1 < x.rhi1);
      good &= plus ? ((Pv.clo2 < x.clo2) & (Pv.chi2 < x.chi2) & (Pv.rlo2 < x.rlo2) & (Pv.rhi2 < x.rhi2))
                   : ((Pv.clo2 > x.clo2) & (Pv.chi2 > x.chi2) & (Pv.rlo2 > x.rlo2) & (Pv.rhi2 > x.rhi2));
    }
    const double av = plus ? x.clo1 - x.clo2 : x.clo1 + x.chi2;
    const double bv = plus ? x.chi1 - x.chi2 : x.chi1 + x.clo2;
    good &= (av > -1.0e9) & (av < 1.0e9) & (bv > -1.0e9) & (bv < 1.0e9) & (em_score > 1.0e-6);
    const bool maybe = try_clean && group_bits<W>(__ballot(act && !good), gbase) == 0;
    if (__ballot(maybe)) { // wave-uniform: some group passed the monotonicity test
      int a_hi = act ? static_cast<int>(ceil(av)) : -2147483647 - 1;
      int a_lo = act ? static_cast<int>(floor(av)) : 2147483647;
      int b_hi = act ? static_cast<int>(ceil(bv)) : -2147483647 - 1;
      int b_lo = act ? static_cast<int>(floor(bv)) : 2147483647;
#pragma unroll
      for (int d = W / 2; d > 0; d >>= 1) {
        a_hi = max(a_hi, __shfl_xor(a_hi, d));
        a_lo = min(a_lo, __shfl_xor(a_lo, d));
        b_hi = max(b_hi, __shfl_xor(b_hi, d));
        b_lo = min(b_lo, __shfl_xor(b_lo, d));
      }
      const long long Wg = static_cast<long long>(a.wiggle) - 3;
      clean = maybe && static_cast<long long>(a_hi) - b_lo <= Wg && static_cast<long long>(b_hi) - a_lo <= Wg;
    }
  }

  // ---- checkCompatibility for every pair k < l: a lane takes pair p0 + sl of ITS edge, W pairs per edge and step ------
  const int P = clean ? 0 : static_cast<int>(n * (n - 1) / 2);
  int       Pmax = P;
#pragma unroll
  for (int d = 32; d >= W; d >>= 1) Pmax = max(Pmax, __shfl_xor(Pmax, d));
  Pmax = __builtin_amdgcn_readfirstlane(Pmax);
  const bool               one_dir = __ballot(m_plus != 0 && m_minus != 0) == 0; // every edge of the wave has one direction
  const unsigned long long KD_one  = __ballot(m_minus == 0);
  // the table of this width through a buffer descriptor (scalar step offset + constant lane offset).  Lanes past their
  // edge's last pair read on into pairs the edge does not have: k < l < W keeps them inside the group's elements, and
  // they store nothing (the store tests the pair number)
  static_assert(sizeof(ChainElem) == 48, "the pair table holds byte offsets of 48-byte elements");
  const __amdgpu_buffer_rsrc_t tab_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t *>(a.pair_tab_sub) + (W == 32 ? 0 : W == 16 ? 1 : 2) * PAIR_TAB_STRIDE * 2, 0, PAIR_TAB_STRIDE * 8, 0x00020000);
  const uint32_t tab_lane = static_cast<uint32_t>(sl) * 8u;
  auto           load_pairs = [&](int p0) __attribute__((always_inline)) -> uint2 {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i   v = __builtin_amdgcn_raw_buffer_load_b64(tab_rsrc, tab_lane, p0 * 8, 0);
    return make_uint2(static_cast<uint32_t>(v.x), static_cast<uint32_t>(v.y));
  };
  const int            Pl  = P - sl; // pair p0 + sl exists iff p0 < Pl
  const unsigned char *elg = reinterpret_cast<const unsigned char *>(el + gbase);
  unsigned char       *cmg = reinterpret_cast<unsigned char *>(cm + gbase);
  double          wiggle  = a.wiggle; // in a vector register: see k_chain
  asm volatile("" : "+v"(wiggle));
  // every element of every group well formed (see nano_check) and v1's raw ranges in list order: the lean pair test
  bool wf_lane = true;
  if (act) {
    const int prev_rlo1 = el[sl > 0 ? lane - 1 : lane].rlo1;
    wf_lane = (x.clo1 <= x.chi1) & (x.clo2 <= x.chi2) & (x.rlo1 <= x.rhi1) & (sl == 0 || prev_rlo1 <= x.rlo1);
  }
  const bool wf = __ballot(!wf_lane) == 0;
  // DIR (see k_chain: the loop is bound by scalar issue): 0 = every EdgeMatch of every edge of the wavefront forward,
  // 1 = all reverse, 3 = one direction per edge (the flip mask is a constant of the wavefront), 2 = an edge with both
  // directions (pairs of one direction only, flip per pair)
  auto sweep_step = [&](int p0, const uint2 kl, auto wft, auto dirt) __attribute__((always_inline)) {
    typedef decltype(wft)  WFT;
    typedef decltype(dirt) DIRT;
    constexpr int          DIR = DIRT::value;
    unsigned long long     bits;
    const bool             in_range = p0 < Pl; // this lane's pair exists
    {
      typedef unsigned long long M;
      // lanes past their edge's last pair evaluate pairs the edge does not have (elements nobody wrote): they store
      // nothing, and they are masked out so that they never send the wavefront into the division
      M valid = __ballot(in_range), KD = KD_one;
      if (DIR == 2) {
        const int  k = static_cast<int>(kl.y & 0xffu), l = static_cast<int>(kl.y >> 18);
        const bool kd = (m_plus >> k) & 1u, ld = (m_plus >> l) & 1u;
        valid &= __ballot(kd == ld);
        KD    = __ballot(kd);
      }
      const ChainElem K = *reinterpret_cast<const ChainElem *>(elg + (kl.x & 0xffffu));
      const ChainElem L = *reinterpret_cast<const ChainElem *>(elg + (kl.x >> 16));
      double          d1, d2;
      const NanoMasks f1 = nano_check<WFT::value, WFT::value>(K.clo1, K.chi1, L.clo1, L.chi1, K.rlo1, K.rhi1, L.rlo1, L.rhi1, d1);
      const NanoMasks f2 = nano_check<WFT::value, false>(K.clo2, K.chi2, L.clo2, L.chi2, K.rlo2, K.rhi2, L.rlo2, L.rhi2, d2);
      M p2, n2m; // :131 flip by EdgeMatch(k).direction
      if (DIR == 0) {
        p2  = f2.pos;
        n2m = f2.neg;
      } else if (DIR == 1) {
        p2  = f2.neg;
        n2m = f2.pos;
      } else {
        const M fl = KD & (f2.pos ^ f2.neg); // where the direction is forward, the two swap
        p2         = f2.neg ^ fl;
        n2m        = f2.pos ^ fl;
      }
      const M codir = (f1.pos & p2) | (f1.neg & n2m);
      const M mixed = f1.ovl ^ f2.ovl;
      M       cl    = codir & ~(f1.abort_ | f2.abort_);
      cl &= valid;
      const double df = fabs(d1 - d2); // = std::max - std::min, see k_chain
      const M near_ = __ballot(df <= wiggle), sum_ok = __ballot(d1 + d2 <= wiggle);
      M       ok    = cl & ((near_ & ~mixed) | (sum_ok & mixed));
      const M need_div = cl & ~(mixed | near_);
      if (need_div) {
        bool pass = false;
        if (__builtin_amdgcn_inverse_ballot_w64(need_div)) pass = df * 100 / fmax(d1, d2) <= a.ratio_pct;
        ok |= __ballot(pass);
      }
      bits = ok;
    }
    // the pairs of row l are consecutive lanes of the group; the first lane of each run stores the run's bits: `run`
    // bits from bit `lane` on, put at bit k (run <= 31; the shift count k is the low five bits of its table word)
    const uint32_t run = (kl.y >> 8) & 0xffu;
    if (run != 0 && in_range) {
      uint32_t *row = reinterpret_cast<uint32_t *>(cmg + (kl.y >> 16));
      *row |= __builtin_amdgcn_ubfe(static_cast<uint32_t>(bits >> lane), 0u, run) << (kl.y & 31u);
    }
  };
  auto sweep = [&](auto wft, auto dirt) __attribute__((always_inline)) {
    // two steps per trip, each with its own registers for the table entries (see k_chain)
    uint2 ka = load_pairs(0);
    for (int p0 = 0; p0 < Pmax; p0 += 2 * W) {
      const uint2 kb = load_pairs(p0 + W);
      sweep_step(p0, ka, wft, dirt);
      if (p0 + W >= Pmax) break;
      ka = load_pairs(p0 + 2 * W);
      sweep_step(p0 + W, kb, wft, dirt);
    }
  };
  auto sweep_dir = [&](auto wft) __attribute__((always_inline)) {
    if (!one_dir) sweep(wft, std::integral_constant<int, 2>{});
    else if (KD_one == ~0ull) sweep(wft, std::integral_constant<int, 0>{});
    else if (KD_one == 0) sweep(wft, std::integral_constant<int, 1>{});
    else sweep(wft, std::integral_constant<int, 3>{});
  };
  if (wf) sweep_dir(std::true_type{});
  else sweep_dir(std::false_type{});
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // bit k: checkCompatibility(k, sl) for k < sl; a clean edge has them all (and then the DP below adds the scores up
  // left to right, exactly the sum k_chain's shortcut forms)
  const uint32_t mycm = clean ? (act ? (1u << sl) - 1u : 0u) : cm[lane];

  // ---- chaining DP (mpp.cpp:181-199) ---------------------------------------------------------------------------------
  int nmax = static_cast<int>(n);
#pragma unroll
  for (int d = 32; d >= W; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d));
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  double   pop = em_score;
  uint32_t pm  = 1u << sl;
  {
    uint32_t pred = static_cast<uint32_t>(sl);
    uint32_t rev  = __builtin_bitreverse32(mycm); // bit k of mycm -> bit 31 - k
    const uint32_t base4 = static_cast<uint32_t>(gbase) * 4u; // ds_bpermute takes a byte address: lane * 4
    for (int k = 0; k + 1 < nmax; ++k) {
      // population[k] of this lane's group: one address addition per step (__shfl adds, masks and shifts for every call)
      const int       src = static_cast<int>(base4 + static_cast<uint32_t>(k) * 4u);
      const long long pb  = __double_as_longlong(pop);
      const uint32_t  plo = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(static_cast<uint32_t>(pb))));
      const uint32_t  phi = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(static_cast<uint32_t>(pb >> 32))));
      const double    k_pop = __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(phi) << 32) | plo));
      const double    cand  = k_pop + em_score; // :189
      // (as in k_chain: bit test and shift are one add with carry-out, the three conditional moves two moves under EXEC)
      unsigned long long comp, saved;
      asm volatile("v_add_co_u32 %0, %1, %0, %0" : "+v"(rev), "=s"(comp));
      const unsigned long long upd = comp & __ballot(cand > pop); // :190-197
      asm volatile("s_and_saveexec_b64 %[sv], %[m]\n\t"
                   "v_mov_b64 %[pop], %[cand]\n\t"
                   "v_mov_b32 %[pred], %[k]\n\t"
                   "s_mov_b64 exec, %[sv]"
                   : [pop] "+v"(pop), [pred] "+v"(pred), [sv] "=&s"(saved)
                   : [m] "s"(upd), [cand] "v"(cand), [k] "s"(k)
                   : "scc");
    }
    uint32_t ptr = pred;
    for (int span = 1; span < nmax; span <<= 1) {
      const uint32_t o  = __shfl(pm, gbase + static_cast<int>(ptr));
      const uint32_t p2 = __shfl(ptr, gbase + static_cast<int>(ptr));
      pm |= o;
      ptr = p2;
    }
  }

  // src/main.cpp:341-353: split by EdgeMatch direction, minus then plus
  SubPath *pmn = reinterpret_cast<SubPath *>(s_wavebuf[wave]) + gbase, *ppl = pmn + 64; // el[] is dead from here on
  const uint32_t n_m = paths_of_direction_sub<W>(m_minus, false, sl, gbase, pop, pm, em_prim, j1, q2, n1, n2, a.alt_frac, pmn);
  const uint32_t n_p = paths_of_direction_sub<W>(m_plus, true, sl, gbase, pop, pm, em_prim, j1, q2, n1, n2, a.alt_frac, ppl);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // filters of src/main.cpp:355-387
  bool     k_m = false, k_p = false;
  uint32_t prim_m = 0, prim_p = 0, len_m = 0, len_p = 0;
  if (sl < static_cast<int>(n_m)) {
    prim_m = pmn[sl].primary;
    len_m  = static_cast<uint32_t>(__popc(pmn[sl].mask));
    k_m    = true;
  }
  if (sl < static_cast<int>(n_p)) {
    prim_p = ppl[sl].primary;
    len_p  = static_cast<uint32_t>(__popc(ppl[sl].mask));
    k_p    = true;
  }
  const bool has_primary = group_bits<W>(__ballot((k_m && prim_m) || (k_p && prim_p)), gbase) != 0;
  if (has_primary) {
    k_m = k_m && prim_m;
    k_p = k_p && prim_p;
  }
  const bool has_multi = group_bits<W>(__ballot((k_m && len_m > 1) || (k_p && len_p > 1)), gbase) != 0;
  if (has_multi) {
    k_m = k_m && len_m > 1;
    k_p = k_p && len_p > 1;
  }
  const uint32_t keep_m = group_bits<W>(__ballot(k_m), gbase), keep_p = group_bits<W>(__ballot(k_p), gbase);
  const int      combined = __popc(keep_m) + __popc(keep_p);
  bool           shadow   = true; // :389-391
  if (combined == 1) shadow = !(keep_m ? pmn[__builtin_ctz(keep_m)].primary : ppl[__builtin_ctz(keep_p)].primary);

  // getOverlap (ol.cpp:53-101) per kept path, minus first
  uint32_t n_orders = 0, n_ids = 0;
  for (int pass = 0; pass < 2; ++pass) {
    uint32_t       keep = pass == 0 ? keep_m : keep_p;
    const SubPath *pv   = pass == 0 ? pmn : ppl;
    const bool     dir  = pass == 1;
    while (__ballot(keep != 0)) {
      const bool has = keep != 0;
      const int  pi  = has ? __builtin_ctz(keep) : 0;
      keep &= keep - 1u;
      const SubPath  rec  = pv[pi];
      const uint32_t mask = (has && rec.mask) ? rec.mask : 1u;
      const int      f = __builtin_ctz(mask), l = 31 - __builtin_clz(mask);
      const double  *kf = s_keep[wave][gbase + f], *kl = s_keep[wave][gbase + l];
      const double   L1 = kf[0], R1 = kl[2];
      const double   L2 = dir ? kf[1] : kf[3], R2 = dir ? kl[3] : kl[1]; // :73-76
      bool     have = false;
      uint32_t fl   = 0;
      double   lo = 0, ro = 0;
      if (L1 <= L2 && R1 <= R2) {
        have = true;
        fl   = MSGPU_ORD_START_V1 | MSGPU_ORD_CONTAINED;
        lo   = L2 - L1;
        ro   = R2 - R1;
      } else if (L1 >= L2 && R1 >= R2) {
        have = true;
        fl   = MSGPU_ORD_CONTAINED;
        lo   = L1 - L2;
        ro   = R1 - R2;
      } else if (L1 > L2 && R1 < R2) {
        have = true;
        fl   = MSGPU_ORD_START_V1;
        lo   = L1 - L2;
        ro   = R2 - R1;
      } else if (L1 < L2 && R1 > R2) {
        have = true;
        fl   = 0;
        lo   = L2 - L1;
        ro   = R1 - R2;
      }
      const bool emit = has && have;
      if (emit) {
        const uint32_t cnt = static_cast<uint32_t>(__popc(mask));
        if ((mask >> sl) & 1u) a.ids_scr[ed.em_off + n_ids + static_cast<uint32_t>(__popc(mask & ((1u << sl) - 1u)))] = s_anchor[wave][lane];
        if (sl == 0) {
          msgpu_order o;
          o.edge_idx     = e;
          o.flags        = fl | (dir ? MSGPU_ORD_DIR : 0u) | (rec.primary ? MSGPU_ORD_PRIMARY : 0u);
          o.left_offset  = lo;
          o.right_offset = ro;
          o.score        = rec.score;
          o.ids_off      = n_ids;
          o.ids_cnt      = cnt;
          o.start        = (fl & MSGPU_ORD_START_V1) ? v1 : v2;
          o.end          = (fl & MSGPU_ORD_START_V1) ? v2 : v1;
          o.base         = v1;
          o.pad[0]       = 0;
          o.pad[1]       = 0;
          a.order_scr[ed.em_off + n_orders] = o;
        }
        ++n_orders;
        n_ids += cnt;
      }
    }
  }
  if (n && sl == 0) {
    a.edge_norders[e] = n_orders;
    a.edge_nids[e]    = n_ids;
    a.edges[e].shadow = shadow ? 1 : 0;
    chunk_add(a.chunk_sums, e, clean, n_orders, n_ids);
  }
}

template <int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W == 32 ? 7 : 6, W == 32 ? 7 : 6))) void k_chain_sub(ChainArgs a, const uint32_t *list, uint32_t n_list) {
  __shared__ __attribute__((aligned(16))) unsigned char s_wavebuf[4][sizeof(ChainElem) * 64];
  __shared__ uint32_t                                  s_cm[4][64];
  __shared__ __attribute__((aligned(16))) double       s_keep[4][64][4];
  __shared__ uint32_t                                  s_anchor[4][64];
  chain_sub_body<W>(a, list, n_list, blockIdx.x, s_wavebuf, s_cm, s_keep, s_anchor);
}
// The three sub-wavefront classes in ONE launch (the step's launches: 14 -> 12): workgroups [0, nb32) take the 32-wide class,
// the next nb16 the 16-wide one, the rest the 8-wide one -- the longest-lived first.  (One kernel = one register budget for the three bodies.)
// Wavefronts per SIMD: 7 since the sub-wavefront body parks what only its last lines need -- the four numbers of the overhangs,
// the anchor id -- in LDS (9 KB a workgroup: seven workgroups still fit a CU) instead of nine registers across the sweep and the
// DP: 71 registers, six spilled outside every loop.  Six wavefronts (73 registers, none spilled) are 20 us slower, eight spill
// inside the loops (profiles/r5_08/README.md).
#ifndef MSGPU_SUB_WAVES
#define MSGPU_SUB_WAVES 7
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MSGPU_SUB_WAVES, MSGPU_SUB_WAVES))) void k_chain_sub_all(ChainArgs a, ChainSubLists l) {
  __shared__ __attribute__((aligned(16))) unsigned char s_wavebuf[4][sizeof(ChainElem) * 64];
  __shared__ uint32_t                                  s_cm[4][64];
  __shared__ __attribute__((aligned(16))) double       s_keep[4][64][4];
  __shared__ uint32_t                                  s_anchor[4][64];
  uint32_t b = blockIdx.x; // (workgroup-uniform branches)
  if (b < l.nb32) {
    chain_sub_body<32>(a, l.list32, l.n32, b, s_wavebuf, s_cm, s_keep, s_anchor);
  } else if ((b -= l.nb32) < l.nb16) {
    chain_sub_body<16>(a, l.list16, l.n16, b, s_wavebuf, s_cm, s_keep, s_anchor);
  } else {
    chain_sub_body<8>(a, l.list8, l.n8, b - l.nb16, s_wavebuf, s_cm, s_keep, s_anchor);
  }
}
template __global__ void k_chain_sub<8>(ChainArgs, const uint32_t *, uint32_t);
template __global__ void k_chain_sub<16>(ChainArgs, const uint32_t *, uint32_t);
template __global__ void k_chain_sub<32>(ChainArgs, const uint32_t *, uint32_t);

// ---------------------------------------------------------------------------------------------------------------------
// chain, big edges (> 64 EdgeMatches): same algorithm, one wavefront per edge, element state in global scratch.
// The O(n^2) compatibility sweep is lane-parallel; the (short) sequential tails run on lane 0.  Slow path, any size.
// ---------------------------------------------------------------------------------------------------------------------

struct BigElem { // 96 bytes per EdgeMatch
  int      rlo1, rhi1, rlo2, rhi2;
  double   clo1, chi1, clo2, chi2, ovr1, ovr2, score;
  uint32_t anchor, j1, q2, flags; // flags bit0 dir, bit1 prim
  double   pop;
  uint32_t pred, used;
};

struct BigPath {
  uint32_t end, first, len, primary;
  uint64_t score;
};

// the fields of a BigElem checkCompatibility looks at
struct BigHot {
  int    rlo1, rhi1, rlo2, rhi2;
  double clo1, chi1, clo2, chi2;
};
template <class EK, class EL>
__device__ __forceinline__ bool big_compat(const EK &K, const EL &L, bool direction, double wiggle, double ratio_pct) {
  int    o1 = 0, o2 = 0;
  double d1 = 0, d2 = 0;
  bool   abort_ = false;
  {
    if (K.clo1 <= L.chi1 && L.clo1 <= K.chi1) {
      if (K.clo1 < L.clo1 && K.chi1 < L.chi1) {
        o1 = 2;
        d1 = K.chi1 - L.clo1 + 1;
      }
      if (K.clo1 > L.clo1 && K.chi1 > L.chi1) {
        o1 = -2;
        d1 = L.chi1 - K.clo1 + 1;
      }
    } else if (K.clo1 < L.clo1) {
      o1 = 1;
      d1 = L.clo1 - K.chi1 + 1;
    } else {
      o1 = -1;
      d1 = K.clo1 - L.chi1 + 1;
    }
    int uco = 0;
    if (K.rlo1 <= L.rhi1 && L.rlo1 <= K.rhi1) {
      if (K.rlo1 < L.rlo1 && K.rhi1 < L.rhi1) uco = 2;
      if (K.rlo1 > L.rlo1 && K.rhi1 > L.rhi1) uco = -2;
      if ((o1 < 0 && uco >= 0) || (o1 > 0 && uco <= 0)) abort_ = true;
    }
  }
  {
    if (K.clo2 <= L.chi2 && L.clo2 <= K.chi2) {
      if (K.clo2 < L.clo2 && K.chi2 < L.chi2) {
        o2 = 2;
        d2 = K.chi2 - L.clo2 + 1;
      }
      if (K.clo2 > L.clo2 && K.chi2 > L.chi2) {
        o2 = -2;
        d2 = L.chi2 - K.clo2 + 1;
      }
    } else if (K.clo2 < L.clo2) {
      o2 = 1;
      d2 = L.clo2 - K.chi2 + 1;
    } else {
      o2 = -1;
      d2 = K.clo2 - L.chi2 + 1;
    }
    int uco = 0;
    if (K.rlo2 <= L.rhi2 && L.rlo2 <= K.rhi2) {
      if (K.rlo2 < L.rlo2 && K.rhi2 < L.rhi2) uco = 2;
      if (K.rlo2 > L.rlo2 && K.rhi2 > L.rhi2) uco = -2;
      if ((o2 < 0 && uco >= 0) || (o2 > 0 && uco <= 0)) abort_ = true;
    }
  }
  if (abort_) return false;
  if (!direction) o2 = -o2;
  if (o1 == o2 && o1 != 0) {
    const double mx = std_max(d1, d2);
    const double df = mx - std_min(d1, d2);
    return (df <= wiggle) || (df * 100 / mx <= ratio_pct);
  }
  if ((o1 < 0 && o2 < 0) || (o1 > 0 && o2 > 0)) return d1 + d2 <= wiggle;
  return false;
}

constexpr uint32_t BIG_REG_CAP = 256; // four elements per lane
__global__ __launch_bounds__(64) void k_chain_big(ChainArgs a, const uint32_t *big_list, const uint64_t *big_off,
                                                  uint32_t n_big, BigElem *elems,
                                                  BigPath *paths /* 2 slots per EdgeMatch: minus then plus */) {
  if (blockIdx.x >= n_big) return;
  const int        lane = threadIdx.x;
  const uint64_t   e    = big_list[blockIdx.x];
  const msgpu_edge ed   = a.edges[e];
  const uint32_t   n    = ed.em_cnt;
  const uint64_t   cp   = a.edge_cand[e];
  const uint32_t   v1 = ed.v1, v2 = ed.v2;
  const uint32_t   n1 = a.read_cnt[v1], n2 = a.read_cnt[v2];
  const int        len1 = a.read_len[v1], len2 = a.read_len[v2];
  BigElem         *E  = elems + big_off[blockIdx.x];
  BigPath         *Pm = paths + 2 * big_off[blockIdx.x], *Pp = Pm + n;
  // Up to BIG_REG_CAP EdgeMatches (nearly every edge that comes here) the chaining DP keeps its state in registers -- lane l
  // holds elements l, l + 64, ... -- and element k reaches the others by readlane, as in k_chain; what the sequential tails
  // chase afterwards (score, predecessor, "used", direction) lies in 5 KB of LDS.  With that state in the global scratch every
  // one of an edge's ~2 n DP steps and every hop of a path walk was a dependent trip to memory: 0.3-0.4 ms for the slowest
  // edge whatever the number of edges -- hidden beside k_chain on a whole job, but the longest thing in a dispatcher window or
  // in the shard of a group member.  Larger edges take the old way.
  __shared__ double   s_pop[BIG_REG_CAP];
  __shared__ uint32_t s_pred[BIG_REG_CAP], s_used[BIG_REG_CAP], s_flags[BIG_REG_CAP];
  const bool fast = n <= BIG_REG_CAP;
  auto POP   = [&](uint32_t i) -> double & { return fast ? s_pop[i] : E[i].pop; };
  auto PRED  = [&](uint32_t i) -> uint32_t & { return fast ? s_pred[i] : E[i].pred; };
  auto USED  = [&](uint32_t i) -> uint32_t & { return fast ? s_used[i] : E[i].used; };
  auto FLAGS = [&](uint32_t i) -> uint32_t { return fast ? s_flags[i] : E[i].flags; };

  // elements + EdgeMatch table
  for (uint32_t i = lane; i < n; i += 64) {
    const uint32_t j1 = a.cand_j[cp + i], t = a.cand_t[cp + i];
    const IRow     m1 = load_irow(&a.by_read[a.read_off[v1] + j1]);
    const IRow     m2 = load_irow(&a.by_anchor[t]);
    const int      ov_lo = max(m1.i_lo, m2.i_lo), ov_hi = min(m1.i_hi, m2.i_hi);
    const bool     d1 = (m1.pf & PF_DIR) != 0, d2 = (m2.pf & PF_DIR) != 0;
    const bool     em_dir = d1 == d2, em_prim = (m1.pf & PF_PRIM) && (m2.pf & PF_PRIM);
    const bool     o1 = m1.line > m2.line;
    const IRow    &om = o1 ? m1 : m2, &im = o1 ? m2 : m1;
    const double   ol = static_cast<double>(om.i_hi - om.i_lo + 1);
    const double   il = static_cast<double>(im.i_hi - im.i_lo + 1);
    const double   cl = static_cast<double>(ov_hi - ov_lo + 1);
    const double   os = static_cast<double>(om.score) * cl / ol;
    const double   is_ = static_cast<double>(im.score) * cl / il;
    BigElem        x;
    x.score  = os + is_;
    x.anchor = m1.other;
    x.j1     = j1;
    x.q2     = m2.pf & PF_POS_MASK;
    x.flags  = (em_dir ? 1u : 0u) | (em_prim ? 2u : 0u);
    msgpu_edgematch em;
    em.ov_lo     = ov_lo;
    em.ov_hi     = ov_hi;
    em.score     = x.score;
    em.anchor_id = x.anchor;
    em.line      = om.line;
    em.flags     = x.flags;
    em.edge_idx  = static_cast<uint32_t>(e) + a.out_edge_base;
    a.ems[ed.em_off + i] = em;
    {
      const double rr  = static_cast<double>(m1.i_hi - m1.i_lo + 1) / static_cast<double>(m1.n_hi - m1.n_lo + 1);
      double       ncl = static_cast<double>(ov_lo - m1.i_lo) / rr;
      double       ncr = static_cast<double>(m1.i_hi - ov_hi) / rr;
      if (!d1) {
        double tmp = ncl;
        ncl        = ncr;
        ncr        = tmp;
      }
      x.rlo1 = m1.n_lo;
      x.rhi1 = m1.n_hi;
      x.clo1 = static_cast<double>(m1.n_lo) + ncl;
      x.chi1 = static_cast<double>(m1.n_hi) - ncr;
      x.ovr1 = static_cast<double>(len1 - m1.n_hi) + ncr;
    }
    {
      const double rr  = static_cast<double>(m2.i_hi - m2.i_lo + 1) / static_cast<double>(m2.n_hi - m2.n_lo + 1);
      double       ncl = static_cast<double>(ov_lo - m2.i_lo) / rr;
      double       ncr = static_cast<double>(m2.i_hi - ov_hi) / rr;
      if (!d2) {
        double tmp = ncl;
        ncl        = ncr;
        ncr        = tmp;
      }
      x.rlo2 = m2.n_lo;
      x.rhi2 = m2.n_hi;
      x.clo2 = static_cast<double>(m2.n_lo) + ncl;
      x.chi2 = static_cast<double>(m2.n_hi) - ncr;
      x.ovr2 = static_cast<double>(len2 - m2.n_hi) + ncr;
    }
    x.pop  = x.score;
    x.pred = 0xffffffffu;
    x.used = 0;
    E[i]   = x;
  }
  __threadfence_block();
  __syncthreads();

  if (fast) {
    // both directions in one loop: a pair is only looked at when its two EdgeMatches have the same direction, and element k's
    // score is final once the steps before k are done (mpp.cpp:185-199 per direction, src/main.cpp:341-353)
    BigHot   Lh[4];
    double   sc[4], pp[4];
    uint32_t pr[4], fl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t l = static_cast<uint32_t>(lane) + 64u * r;
      Lh[r] = BigHot{0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0};
      sc[r] = pp[r] = 0.0;
      pr[r] = 0xffffffffu;
      fl[r] = 0;
      if (l < n) {
        const BigElem x = E[l];
        Lh[r] = BigHot{x.rlo1, x.rhi1, x.rlo2, x.rhi2, x.clo1, x.chi1, x.clo2, x.chi2};
        sc[r] = pp[r] = x.score;
        fl[r] = x.flags;
      }
    }
    for (uint32_t k = 0; k + 1 < n; ++k) {
      const int src = static_cast<int>(k & 63u);
      BigHot    K;
      double    k_pop;
      uint32_t  k_fl;
      auto take = [&](int r) { // element k from the lane that holds it (r is wave-uniform)
        K.rlo1 = rl_i32(Lh[r].rlo1, src);
        K.rhi1 = rl_i32(Lh[r].rhi1, src);
        K.rlo2 = rl_i32(Lh[r].rlo2, src);
        K.rhi2 = rl_i32(Lh[r].rhi2, src);
        K.clo1 = rl_f64(Lh[r].clo1, src);
        K.chi1 = rl_f64(Lh[r].chi1, src);
        K.clo2 = rl_f64(Lh[r].clo2, src);
        K.chi2 = rl_f64(Lh[r].chi2, src);
        k_pop  = rl_f64(pp[r], src);
        k_fl   = rl_u32(fl[r], src);
      };
      switch (k >> 6) {
        case 0: take(0); break;
        case 1: take(1); break;
        case 2: take(2); break;
        default: take(3); break;
      }
      const bool direction = (k_fl & 1u) != 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (64u * r + 63u <= k || 64u * r >= n) continue; // (wave-uniform: nothing of this slot lies behind k, or at all)
        const uint32_t l = static_cast<uint32_t>(lane) + 64u * r;
        if (l > k && l < n && ((fl[r] ^ k_fl) & 1u) == 0) {
          const bool   ok   = big_compat(K, Lh[r], direction, a.wiggle, a.ratio_pct);
          const double cand = k_pop + sc[r];
          if (ok && cand > pp[r]) {
            pp[r] = cand;
            pr[r] = k;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t l = static_cast<uint32_t>(lane) + 64u * r;
      if (l < n) {
        s_pop[l]   = pp[r];
        s_pred[l]  = pr[r];
        s_used[l]  = 0;
        s_flags[l] = fl[r];
      }
    }
    __syncthreads();
  }

  uint32_t np_dir[2] = {0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    const bool     direction = pass == 1;
    const uint32_t want      = direction ? 1u : 0u;
    BigPath       *P         = direction ? Pp : Pm;
    // DP (an edge beyond the register path: state in the global scratch)
    for (uint32_t k = 0; !fast && k + 1 < n; ++k) {
      const BigElem K = E[k];
      if ((K.flags & 1u) != want) continue;
      for (uint32_t l = k + 1 + lane; l < n; l += 64) {
        BigElem L = E[l];
        if ((L.flags & 1u) != want) continue;
        const bool   ok   = big_compat(K, L, direction, a.wiggle, a.ratio_pct);
        const double cand = K.pop + L.score;
        if (ok && cand > L.pop) {
          E[l].pop  = cand;
          E[l].pred = k;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    // sequential tail on lane 0
    uint32_t np = 0;
    if (lane == 0) {
      double   maxv = 0.0;
      uint32_t maxi = 0xffffffffu, firsti = 0xffffffffu;
      for (uint32_t i = 0; i < n; ++i) {
        if ((FLAGS(i) & 1u) != want) continue;
        if (firsti == 0xffffffffu) firsti = i;
        if (POP(i) > maxv) {
          maxv = POP(i);
          maxi = i;
        }
      }
      if (firsti != 0xffffffffu) {
        if (maxi == 0xffffffffu) maxi = firsti;
        // best path
        {
          uint32_t len = 0, first = maxi;
          bool     prim = false;
          for (uint32_t c = maxi; c != 0xffffffffu; c = PRED(c)) {
            USED(c) = 1;
            prim |= (FLAGS(c) & 2u) != 0;
            first = c;
            ++len;
          }
          P[0].end     = maxi;
          P[0].first   = first;
          P[0].len     = len;
          P[0].primary = (prim || len > 2) ? 1u : 0u;
          P[0].score   = static_cast<uint64_t>(maxv);
          np           = 1;
        }
        const double thr = maxv * a.alt_frac;
        for (uint32_t p = 0; p < n; ++p) {
          if ((FLAGS(p) & 1u) != want || !(POP(p) > thr)) continue;
          bool     disjoint = true;
          uint32_t len = 0, first = p;
          bool     prim = false;
          for (uint32_t c = p; c != 0xffffffffu; c = PRED(c)) {
            if (USED(c)) {
              disjoint = false;
              break;
            }
            prim |= (FLAGS(c) & 2u) != 0;
            first = c;
            ++len;
          }
          if (!disjoint) continue;
          for (uint32_t c = p; c != 0xffffffffu; c = PRED(c)) USED(c) = 1;
          P[np].end     = p;
          P[np].first   = first;
          P[np].len     = len;
          P[np].primary = prim ? 1u : 0u;
          P[np].score   = static_cast<uint64_t>(POP(p));
          ++np;
        }
        if (np == 1 && P[0].primary) {
          const uint32_t f = P[0].first, l = P[0].end;
          auto           qe = [&](uint32_t i) { return direction ? E[i].q2 : (n2 - 1 - E[i].q2); };
          bool           demote;
          if ((E[f].j1 != 0 && qe(f) != 0) || (E[l].j1 != n1 - 1 && qe(l) != n2 - 1)) {
            demote = true;
          } else {
            // ascending walk: reverse the pred chain into the id scratch of this edge (free at this point)
            uint32_t *tmp = a.ids_scr + ed.em_off;
            uint32_t  len = P[0].len, w = len;
            for (uint32_t c = l; c != 0xffffffffu; c = PRED(c)) tmp[--w] = c;
            long long i = 0, jj = 0;
            bool      is_shadow = false;
            for (uint32_t t = 0; t < len && !is_shadow; ++t) {
              const uint32_t  c  = tmp[t];
              const long long rs = static_cast<long long>(E[c].j1);
              bool            inter = rs > i;
              i                     = rs + 1;
              long long re          = static_cast<long long>(qe(c));
              if (re < jj) re = static_cast<long long>(n2);
              inter &= re > jj;
              jj        = re + 1;
              is_shadow = inter;
            }
            demote = is_shadow;
          }
          if (demote) P[0].primary = 0;
        }
      }
    }
    np_dir[pass] = np; // valid on lane 0
    __threadfence_block();
    __syncthreads();
  }

  if (lane == 0) {
    const uint32_t n_m = np_dir[0], n_p = np_dir[1];
    bool           has_primary = false;
    for (uint32_t i = 0; i < n_p; ++i) has_primary |= Pp[i].primary != 0;
    for (uint32_t i = 0; i < n_m; ++i) has_primary |= Pm[i].primary != 0;
    // keep flag in bit 31 of .primary's companion: reuse .first's top bit is unsafe; recompute on the fly instead
    auto keep1 = [&](const BigPath &p) { return !has_primary || p.primary; };
    bool has_multi = false;
    for (uint32_t i = 0; i < n_p; ++i) has_multi |= keep1(Pp[i]) && Pp[i].len > 1;
    for (uint32_t i = 0; i < n_m; ++i) has_multi |= keep1(Pm[i]) && Pm[i].len > 1;
    auto     keep = [&](const BigPath &p) { return keep1(p) && (!has_multi || p.len > 1); };
    uint32_t combined = 0;
    for (uint32_t i = 0; i < n_m; ++i) combined += keep(Pm[i]) ? 1u : 0u;
    for (uint32_t i = 0; i < n_p; ++i) combined += keep(Pp[i]) ? 1u : 0u;
    bool shadow;
    if (combined > 1) {
      shadow = true;
    } else {
      const BigPath *first = nullptr;
      for (uint32_t i = 0; i < n_m && !first; ++i)
        if (keep(Pm[i])) first = &Pm[i];
      for (uint32_t i = 0; i < n_p && !first; ++i)
        if (keep(Pp[i])) first = &Pp[i];
      shadow = first ? !first->primary : false;
    }
    uint32_t n_orders = 0, n_ids = 0;
    for (int pass = 0; pass < 2; ++pass) {
      const BigPath *pv  = pass == 0 ? Pm : Pp;
      const uint32_t npv = pass == 0 ? n_m : n_p;
      const bool     dir = pass == 1;
      for (uint32_t pi = 0; pi < npv; ++pi) {
        if (!keep(pv[pi])) continue;
        const uint32_t f = pv[pi].first, l = pv[pi].end;
        const double   L1 = E[f].clo1, R1 = E[l].ovr1;
        double         L2 = E[f].clo2, R2 = E[l].ovr2;
        if (!dir) {
          L2 = E[f].ovr2;
          R2 = E[l].clo2;
        }
        bool     have = false;
        uint32_t fl   = 0;
        double   lo = 0, ro = 0;
        if (L1 <= L2 && R1 <= R2) {
          have = true;
          fl   = MSGPU_ORD_START_V1 | MSGPU_ORD_CONTAINED;
          lo   = L2 - L1;
          ro   = R2 - R1;
        } else if (L1 >= L2 && R1 >= R2) {
          have = true;
          fl   = MSGPU_ORD_CONTAINED;
          lo   = L1 - L2;
          ro   = R1 - R2;
        } else if (L1 > L2 && R1 < R2) {
          have = true;
          fl   = MSGPU_ORD_START_V1;
          lo   = L1 - L2;
          ro   = R2 - R1;
        } else if (L1 < L2 && R1 > R2) {
          have = true;
          fl   = 0;
          lo   = L2 - L1;
          ro   = R1 - R2;
        }
        if (!have) continue;
        const uint32_t cnt = pv[pi].len;
        uint32_t       w   = cnt;
        for (uint32_t c = l; c != 0xffffffffu; c = PRED(c)) a.ids_scr[ed.em_off + n_ids + --w] = E[c].anchor;
        msgpu_order o;
        o.edge_idx     = static_cast<uint32_t>(e);
        o.flags        = fl | (dir ? MSGPU_ORD_DIR : 0u) | (pv[pi].primary ? MSGPU_ORD_PRIMARY : 0u);
        o.left_offset  = lo;
        o.right_offset = ro;
        o.score        = pv[pi].score;
        o.ids_off      = n_ids;
        o.ids_cnt      = cnt;
        o.start        = (fl & MSGPU_ORD_START_V1) ? v1 : v2;
        o.end          = (fl & MSGPU_ORD_START_V1) ? v2 : v1;
        o.base         = v1;
        o.pad[0]       = 0;
        o.pad[1]       = 0;
        a.order_scr[ed.em_off + n_orders] = o;
        ++n_orders;
        n_ids += cnt;
      }
    }
    a.edge_norders[e] = n_orders;
    a.edge_nids[e]    = n_ids;
    a.edges[e].shadow = shadow ? 1 : 0;
    chunk_add(a.chunk_sums, e, false, n_orders, n_ids);
  }
}

// Pair tables: entry p = k | l << 8 | run << 16 | (64 - run - k) << 24 for the flattened pair index p = l(l-1)/2 + k, k < l < 64, where run
// is the length of the stretch of row l that starts at lane p % W of a W-wide sweep step (0 if no stretch starts
// there).  Four tables (W = 64, 32, 16, 8) of PAIR_TAB_STRIDE entries, each padded with (0, 1, 0).
// The width-64 table once more, 4 bytes per pair, in the form k_chain's sweep consumes it (byte offsets that an SDWA add
// takes as they are): byte offset of element k in the wavefront's element array | byte offset of element l << 16
// (k and l themselves, which only mixed-direction edges need, are the offsets / 48)
__device__ __forceinline__ void tab64_entry(uint32_t *tab, int p, int k, int l, int run) {
  (void)run;
  // k_chain: the LDS addresses of the pair's two elements (its element table starts at LDS address CHAIN_LDS_EL = 0)
  tab[4 * PAIR_TAB_STRIDE + 2 * p]     = static_cast<uint32_t>(k * 48);
  tab[4 * PAIR_TAB_STRIDE + 2 * p + 1] = static_cast<uint32_t>(l * 48);
}
// The tables of the sub-wavefront kernels (W = 32, 16, 8) once more, 8 bytes per pair, in the form their sweep consumes:
//   x = byte offset of element k in the group's elements | byte offset of element l << 16
//   y = k | run << 8 | byte offset of row l in the group's compatibility rows << 16   (k: low five bits = shift count;
//       run = 0: this lane stores nothing)
__device__ __forceinline__ void tabsub_entry(uint32_t *tab, int t, int p, int k, int l, int run) {
  uint32_t *e = tab + 4 * PAIR_TAB_STRIDE + 2 * PAIR_TAB_STRIDE + (static_cast<size_t>(t - 1) * PAIR_TAB_STRIDE + p) * 2;
  e[0]        = static_cast<uint32_t>(k * 48) | (static_cast<uint32_t>(l * 48) << 16);
  e[1]        = static_cast<uint32_t>(k) | (static_cast<uint32_t>(run) << 8) | (static_cast<uint32_t>(l * 4) << 16);
}
__global__ __launch_bounds__(256) void k_fill_pair_tab(uint32_t *tab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * static_cast<int>(PAIR_TAB_STRIDE)) return;
  const int t = i / static_cast<int>(PAIR_TAB_STRIDE), p = i % static_cast<int>(PAIR_TAB_STRIDE), W = 64 >> t;
  if (p >= 2016) {
    tab[i] = 1u << 8;
    if (t == 0) tab64_entry(tab, p, 0, 1, 0);
    else tabsub_entry(tab, t, p, 0, 1, 0);
    return;
  }
  int l = static_cast<int>((1.0f + __fsqrt_rn(1.0f + 8.0f * static_cast<float>(p))) * 0.5f);
  if (l * (l - 1) / 2 > p) --l;
  if ((l + 1) * l / 2 <= p) ++l;
  const int k = p - l * (l - 1) / 2, lane = p % W;
  const int run = (k == 0 || lane == 0) ? min(l - k, W - lane) : 0;
  tab[i] = static_cast<uint32_t>(k | (l << 8) | (run << 16)) | (static_cast<uint32_t>(64 - run - k) << 24);
  if (t == 0) tab64_entry(tab, p, k, l, run);
  else tabsub_entry(tab, t, p, k, l, run);
}

// dense, canonical order + id tables -- and the scan that places them, and the read-back of their sizes, in the same launch.
// A workgroup per chunk of COMPACT_CHUNK edges.  Every workgroup sums the chunk sums the chain kernels left (chunk_add): the
// ones before its own are its base, all of them the table sizes.  Workgroup 0 writes the sizes into the scalar block and
// publishes the block to the host AT ONCE (k_publish_scalars' protocol): the host turns around while the tables are still being
// written.  Then the prefix inside the chunk (a block scan over the per-edge counts), then the move.
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_compact(CompactArgs a) {
  __shared__ unsigned long long s_red[16][5], s_tot[5], s_em[COMPACT_CHUNK];
  __shared__ uint32_t           s_ob[COMPACT_CHUNK], s_ib[COMPACT_CHUNK], s_no[COMPACT_CHUNK], s_w[2][16];
  static_assert(COMPACT_CHUNK == 1024, "an edge per thread");
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t chunk = blockIdx.x;
  // (this thread's edge: asked for now, used behind the sums below -- one wait instead of two in a row)
  const uint64_t my_e  = static_cast<uint64_t>(chunk) * COMPACT_CHUNK + threadIdx.x;
  const uint32_t my_no = my_e < a.n_edges ? a.edge_norders[my_e] : 0u, my_ni = my_e < a.n_edges ? a.edge_nids[my_e] : 0u;
  const uint64_t my_em = my_e < a.n_edges ? a.edges[my_e].em_off : 0ull;
  unsigned long long t[5] = {0, 0, 0, 0, 0}; // orders, ids of the chunks before this one | orders, ids, shortcut edges of all chunks
  for (uint32_t c = threadIdx.x; c < a.n_chunks; c += 1024) {
    const unsigned long long w0 = a.chunk_sums[2 * c], w1 = a.chunk_sums[2 * c + 1];
    const unsigned long long no = w0 >> 32, nf = w0 & 0xffffffffull;
    if (c < chunk) {
      t[0] += no;
      t[1] += w1;
    }
    t[2] += no;
    t[3] += w1;
    t[4] += nf;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    for (int d = 32; d > 0; d >>= 1) t[k] += __shfl_xor(t[k], d);
    if (lane == 0) s_red[wave][k] = t[k];
  }
  __syncthreads();
  // (the sixteen partial sums are added by the first wavefront and handed to everybody as five words: every thread adding all
  // eighty of them held them in registers at once, and the kernel must stay within 64 to have two workgroups on a CU)
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      unsigned long long v = lane < 16 ? s_red[lane][k] : 0ull;
      for (int d = 8; d > 0; d >>= 1) v += __shfl_xor(v, d);
      if (lane == 0) s_tot[k] = v;
    }
  }
  __syncthreads();
  const unsigned long long base_o = s_tot[0], base_i = s_tot[1], tot_o = s_tot[2], tot_i = s_tot[3], tot_f = s_tot[4];
  if (chunk == 0 && a.scalars && threadIdx.x < 64) {
    // (every launch writes the sizes; only the first launch of a call publishes -- a repeat after a reallocation has seq = 0)
    if (lane == 0) {
      a.scalars[a.slot_orders] = tot_o;
      a.scalars[a.slot_ids]    = tot_i;
      a.scalars[a.slot_fast]   = tot_f;
    }
    if (a.host_scalars && a.seq) {
      if (static_cast<uint32_t>(lane) < a.n_scalars) {
        const uint32_t k = lane;
        a.host_scalars[k] = k == a.slot_orders ? tot_o : k == a.slot_ids ? tot_i : k == a.slot_fast ? tot_f : a.scalars[k];
      }
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&a.host_scalars[a.n_scalars], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // (may be launched before the host knows the table sizes, into whatever the tables hold from earlier calls: if they do not
  // fit nothing is written; the host, which compares the same numbers, allocates and launches again)
  if (tot_o > a.cap_orders || tot_i > a.cap_ids) return;
  const uint64_t e0 = static_cast<uint64_t>(chunk) * COMPACT_CHUNK;
  if (e0 >= a.n_edges) return;
  {
    // prefix inside the chunk: an edge per thread.  What the move below needs of the edge -- its order count and where its
    // scratch begins -- is read here too, beside the counts, and handed over through LDS: the move's chain of dependent reads
    // (edge -> order record -> ids) is one read shorter, and the four rounds of it start together.
    const uint32_t no = my_no, ni = my_ni;
    const uint64_t em = my_em;
    const uint32_t io = wave_incl_scan(no), ii = wave_incl_scan(ni);
    if (lane == 63) {
      s_w[0][wave] = io;
      s_w[1][wave] = ii;
    }
    __syncthreads();
    uint32_t bo = io - no, bi = ii - ni;
    for (int w = 0; w < wave; ++w) {
      bo += s_w[0][w];
      bi += s_w[1][w];
    }
    s_ob[threadIdx.x] = bo;
    s_ib[threadIdx.x] = bi;
    s_no[threadIdx.x] = no;
    s_em[threadIdx.x] = em;
  }
  __syncthreads();
  // Four lanes per edge, sixteen edges per wavefront, four rounds per workgroup: a lane moves one 16-byte quarter of an
  // order record and every fourth id.  (Most edges have one order; the dependent chain "order record -> id count ->
  // ids" is then as long as a single edge's, not sixteen of them in a row.)  The kernel waits on memory: what counts is how
  // many such chains a CU has in flight -- two workgroups (64 registers a lane: the rounds are NOT unrolled into each other,
  // which took 96 and left room for one).
  constexpr int ROUNDS = COMPACT_CHUNK / 256;
  const int     sub = lane & 3;
#pragma unroll 1
  for (int round = 0; round < ROUNDS; ++round) {
    const uint32_t le   = round * 256 + wave * 16 + (lane >> 2);
    const uint64_t e    = e0 + le;
    const bool     have = e < a.n_edges;
    const uint32_t no   = have ? s_no[le] : 0u;
    const uint64_t em_off = s_em[le];
    uint64_t       oo = 0, io = 0;
    if (have) {
      oo = base_o + s_ob[le];
      io = base_i + s_ib[le];
      if (sub == 0) { // cross references leave as positions in the whole job's tables (a batch / shard adds its bases)
        a.edges[e].order_off = oo + a.out_order_base;
        a.edges[e].order_cnt = static_cast<uint16_t>(no);
        a.edges[e].em_off    = em_off + a.out_em_base;
      }
    }
    uint32_t max_no = no; // every lane runs the same number of rounds (shuffles inside)
    for (int d = 32; d >= 4; d >>= 1) max_no = max(max_no, static_cast<uint32_t>(__shfl_xor(static_cast<int>(max_no), d)));
    for (uint32_t i = 0; i < max_no; ++i) {
      const bool on = i < no;
      uint4      w  = make_uint4(0, 0, 0, 0);
      if (on) w = reinterpret_cast<const uint4 *>(&a.order_scr[em_off + i])[sub];
      // msgpu_order as 16 dwords: [8,9] = ids_off (edge-relative in the scratch), [10] = ids_cnt -- the third quarter
      const int      q2  = (lane & ~3) + 2;
      const uint32_t rel = static_cast<uint32_t>(__shfl(static_cast<int>(w.x), q2));
      const uint32_t cnt = static_cast<uint32_t>(__shfl(static_cast<int>(w.z), q2));
      if (sub == 2) {
        const uint64_t gio = io + a.out_ids_base;
        w.x = static_cast<uint32_t>(gio);
        w.y = static_cast<uint32_t>(gio >> 32);
      }
      if (sub == 0) w.x += a.out_edge_base; // dword 0 = edge_idx
      if (on) {
        reinterpret_cast<uint4 *>(&a.orders[oo + i])[sub] = w;
        for (uint32_t q = sub; q < cnt; q += 4) a.ids[io + q] = a.ids_scr[em_off + rel + q];
        io += cnt;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// multi-GPU merge: after the all-gather every rank holds `world` padded slabs (edges | orders | ids); compact them into
// dense rank-major tables and re-base the cross references (edge -> orders, order -> edge, order -> ids).
// EdgeMatch offsets (em_off) stay rank-local: EdgeMatch tables are not gathered.
// ---------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_merge_gathered(MergeArgs a) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  const uint64_t nE = a.base[a.world].edges, nO = a.base[a.world].orders, nI = a.base[a.world].ids;
  if (i < nE) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].edges) ++r;
    const uint64_t    k  = i - a.base[r].edges;
    const msgpu_edge *src = reinterpret_cast<const msgpu_edge *>(a.gathered + r * a.slab_bytes + a.off_edges);
    msgpu_edge        e  = src[k];
    e.order_off += a.base[r].orders;
    e.v1 += a.base[r].read_id; // (partitions of a larger job: the rank's read ids start at its base)
    e.v2 += a.base[r].read_id;
    a.edges[i] = e;
  }
  if (i < nO) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].orders) ++r;
    const uint64_t     k   = i - a.base[r].orders;
    const msgpu_order *src = reinterpret_cast<const msgpu_order *>(a.gathered + r * a.slab_bytes + a.off_orders);
    msgpu_order        o   = src[k];
    o.edge_idx += static_cast<uint32_t>(a.base[r].edges);
    o.ids_off += a.base[r].ids;
    o.start += a.base[r].read_id;
    o.end += a.base[r].read_id;
    o.base += a.base[r].read_id;
    a.orders[i] = o;
  }
  if (i < nI) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].ids) ++r;
    const uint64_t  k   = i - a.base[r].ids;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.gathered + r * a.slab_bytes + a.off_ids);
    a.ids[i]            = src[k] + a.base[r].anchor_id;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the exchange's wire form: what travels over xGMI is what the receiver cannot derive.  Per rank
//   edges  part (17 n + 8 bytes): em_off[n + 1] u32 | order_off[n + 1] u32 | v1[n] | v2[n] | shadow[n] u8
//   orders part (33 n + 4 bytes): left[n] f64 | right[n] f64 | score[n] u64 | ids_off[n + 1] u32 | edge_idx[n] u32 | flags[n] u8
//   ids    part: 4 n bytes, or -- while every anchor id fits 24 bits -- 3 n bytes (four ids in three words)
// The counts are differences of the CSR offsets (a rank's tables are dense: every record's slice starts where the one
// before ends), start / end / base follow from the flags and the edge's vertices (k_chain's emission: base = v1).
// ---------------------------------------------------------------------------------------------------------------------

struct WireEdges {
  const uint32_t *em_off, *order_off, *v1, *v2;
  const uint8_t  *shadow;
  __device__ WireEdges(const uint8_t *p, uint64_t n) {
    em_off    = reinterpret_cast<const uint32_t *>(p);
    order_off = em_off + (n + 1);
    v1        = order_off + (n + 1);
    v2        = v1 + n;
    shadow    = reinterpret_cast<const uint8_t *>(v2 + n);
  }
};
struct WireOrders {
  const double   *left, *right;
  const uint64_t *score;
  const uint32_t *ids_off, *edge_idx;
  const uint8_t  *flags;
  __device__ WireOrders(const uint8_t *p, uint64_t n) {
    left     = reinterpret_cast<const double *>(p);
    right    = left + n;
    score    = reinterpret_cast<const uint64_t *>(right + n);
    ids_off  = reinterpret_cast<const uint32_t *>(score + n);
    edge_idx = ids_off + (n + 1);
    flags    = reinterpret_cast<const uint8_t *>(edge_idx + n);
  }
};

__global__ __launch_bounds__(256) void k_pack_wire(PackWireArgs a) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < a.n_edges) {
    const WireEdges  w(a.w_edges, a.n_edges);
    const msgpu_edge e = a.edges[i];
    const_cast<uint32_t *>(w.em_off)[i]    = static_cast<uint32_t>(e.em_off - a.base_ems);
    const_cast<uint32_t *>(w.order_off)[i] = static_cast<uint32_t>(e.order_off - a.base_orders);
    const_cast<uint32_t *>(w.v1)[i]        = e.v1;
    const_cast<uint32_t *>(w.v2)[i]        = e.v2;
    const_cast<uint8_t *>(w.shadow)[i]     = e.shadow;
    if (i + 1 == a.n_edges) {
      const_cast<uint32_t *>(w.em_off)[i + 1]    = static_cast<uint32_t>(e.em_off - a.base_ems + e.em_cnt);
      const_cast<uint32_t *>(w.order_off)[i + 1] = static_cast<uint32_t>(e.order_off - a.base_orders + e.order_cnt);
    }
  }
  if (i < a.n_orders) {
    const WireOrders  w(a.w_orders, a.n_orders);
    const msgpu_order o = a.orders[i];
    const_cast<double *>(w.left)[i]       = o.left_offset;
    const_cast<double *>(w.right)[i]      = o.right_offset;
    const_cast<uint64_t *>(w.score)[i]    = o.score;
    const_cast<uint32_t *>(w.ids_off)[i]  = static_cast<uint32_t>(o.ids_off - a.base_ids);
    const_cast<uint32_t *>(w.edge_idx)[i] = o.edge_idx - a.base_edges;
    const_cast<uint8_t *>(w.flags)[i]     = static_cast<uint8_t>(o.flags);
    if (i + 1 == a.n_orders) const_cast<uint32_t *>(w.ids_off)[i + 1] = static_cast<uint32_t>(o.ids_off - a.base_ids + o.ids_cnt);
  }
  if (i == 0) { // an empty table still has its closing CSR entries on the wire (the host statement writes zeros there)
    if (a.n_edges == 0) {
      const WireEdges w(a.w_edges, 0);
      const_cast<uint32_t *>(w.em_off)[0]    = 0;
      const_cast<uint32_t *>(w.order_off)[0] = 0;
    }
    if (a.n_orders == 0) const_cast<uint32_t *>(WireOrders(a.w_orders, 0).ids_off)[0] = 0;
  }
  if (a.ids && 4 * i < a.n_ids) { // 3-byte ids: four ids -> three words (the last group writes the words its ids reach into)
    const uint64_t rem = a.n_ids - 4 * i;
    const uint32_t i0 = a.ids[4 * i], i1 = rem > 1 ? a.ids[4 * i + 1] : 0, i2 = rem > 2 ? a.ids[4 * i + 2] : 0,
                   i3 = rem > 3 ? a.ids[4 * i + 3] : 0;
    uint32_t *w = a.w_ids + 3 * i;
    w[0] = i0 | (i1 << 24);
    if (rem > 1) w[1] = (i1 >> 8) | (i2 << 16);
    if (rem > 2) w[2] = (i2 >> 16) | (i3 << 8);
  }
}

// the merge of k_merge_gathered over slabs in wire form: the same dense rank-major tables, byte for byte
template <bool IDS3> __global__ __launch_bounds__(256) void k_merge_wire(MergeArgs a) {
  // A thread builds one edge and one order record from the wire columns; the records leave through LDS so that every store
  // instruction of a wavefront writes 1 KB of consecutive bytes (a record per thread stored as it stands touches 64 sectors with
  // 16 bytes each, four times over: the kernel ran at a third of what the same bytes take as whole lines).
  __shared__ uint4 s_rec[256 * 4]; // 256 orders of 64 bytes; the 256 edges of 32 bytes go through the first half before them
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x, i0 = static_cast<uint64_t>(blockIdx.x) * 256;
  const uint64_t nE = a.base[a.world].edges, nO = a.base[a.world].orders, nI = a.base[a.world].ids;
  if (i < nE) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].edges) ++r;
    const uint64_t  k = i - a.base[r].edges;
    const WireEdges w(a.gathered + r * a.slab_bytes + a.off_edges, a.base[r + 1].edges - a.base[r].edges);
    const uint32_t  m0 = w.em_off[k], q0 = w.order_off[k];
    msgpu_edge      e;
    e.v1        = w.v1[k] + a.base[r].read_id;
    e.v2        = w.v2[k] + a.base[r].read_id;
    e.em_off    = m0;
    e.order_off = q0 + a.base[r].orders;
    e.em_cnt    = w.em_off[k + 1] - m0;
    e.order_cnt = static_cast<uint16_t>(w.order_off[k + 1] - q0);
    e.shadow    = w.shadow[k];
    e.pad       = 0;
    *reinterpret_cast<msgpu_edge *>(&s_rec[threadIdx.x * 2]) = e;
  }
  if (i0 < nE) { // (block-uniform)
    __syncthreads();
    const uint64_t quads = min<uint64_t>(256, nE - i0) * 2; // 16-byte pieces of this block's edges
    uint4         *dst   = reinterpret_cast<uint4 *>(a.edges + i0);
    for (uint32_t q = threadIdx.x; q < quads; q += 256) dst[q] = s_rec[q];
    __syncthreads();
  }
  if (i < nO) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].orders) ++r;
    const uint64_t   k = i - a.base[r].orders;
    const WireOrders w(a.gathered + r * a.slab_bytes + a.off_orders, a.base[r + 1].orders - a.base[r].orders);
    const WireEdges  we(a.gathered + r * a.slab_bytes + a.off_edges, a.base[r + 1].edges - a.base[r].edges);
    const uint32_t   ei = w.edge_idx[k], fl = w.flags[k], j0 = w.ids_off[k];
    const uint32_t   v1 = we.v1[ei] + a.base[r].read_id, v2 = we.v2[ei] + a.base[r].read_id;
    msgpu_order      o;
    o.edge_idx     = ei + static_cast<uint32_t>(a.base[r].edges);
    o.flags        = fl;
    o.left_offset  = w.left[k];
    o.right_offset = w.right[k];
    o.score        = w.score[k];
    o.ids_off      = j0 + a.base[r].ids;
    o.ids_cnt      = w.ids_off[k + 1] - j0;
    o.start        = (fl & MSGPU_ORD_START_V1) ? v1 : v2;
    o.end          = (fl & MSGPU_ORD_START_V1) ? v2 : v1;
    o.base         = v1;
    o.pad[0]       = 0;
    o.pad[1]       = 0;
    *reinterpret_cast<msgpu_order *>(&s_rec[threadIdx.x * 4]) = o;
  }
  if (i0 < nO) {
    __syncthreads();
    const uint64_t quads = min<uint64_t>(256, nO - i0) * 4;
    uint4         *dst   = reinterpret_cast<uint4 *>(a.orders + i0);
    for (uint32_t q = threadIdx.x; q < quads; q += 256) dst[q] = s_rec[q];
  }
  if (i < nI) {
    uint32_t r = 0;
    while (i >= a.base[r + 1].ids) ++r;
    const uint64_t  k   = i - a.base[r].ids;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.gathered + r * a.slab_bytes + a.off_ids);
    uint32_t        id;
    if (IDS3) {
      const uint32_t *w = src + 3 * (k >> 2);
      const uint32_t  j = static_cast<uint32_t>(k & 3);
      id = j == 0 ? (w[0] & 0xffffffu) : j == 1 ? ((w[0] >> 24) | ((w[1] & 0xffffu) << 8))
           : j == 2 ? ((w[1] >> 16) | ((w[2] & 0xffu) << 16)) : (w[2] >> 8);
    } else {
      id = src[k];
    }
    a.ids[i] = id + a.base[r].anchor_id;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------------------------------------------------

static inline dim3 grid1(uint64_t n, uint32_t per_block) { return dim3(static_cast<uint32_t>((n + per_block - 1) / per_block)); }

void launch_max_ids(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t *max_ids) {
  if (n) {
    uint64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_max_ids, dim3(static_cast<uint32_t>(nb < 2048 ? nb : 2048)), dim3(256), 0, st, rows, n, max_ids);
  }
}
void launch_index_init(hipStream_t st, uint32_t *const zero[4], const uint32_t n_zero[4], uint32_t *const ones[2],
                       const uint32_t n_ones[2]) {
  uint32_t *const z8[8] = {zero[0], zero[1], zero[2], zero[3], nullptr, nullptr, nullptr, nullptr};
  const uint32_t  n8[8] = {n_zero[0], n_zero[1], n_zero[2], n_zero[3], 0, 0, 0, 0};
  launch_index_init8(st, z8, n8, ones, n_ones);
}
void launch_index_init8(hipStream_t st, uint32_t *const zero[8], const uint32_t n_zero[8], uint32_t *const ones[2],
                        const uint32_t n_ones[2]) {
  IndexInitArgs a;
  uint32_t      most = 1;
  for (int k = 0; k < 8; ++k) {
    a.zero[k]   = zero[k];
    a.n_zero[k] = zero[k] ? n_zero[k] : 0;
    most        = a.n_zero[k] > most ? a.n_zero[k] : most;
  }
  for (int k = 0; k < 2; ++k) {
    a.ones[k]   = ones[k];
    a.n_ones[k] = n_ones[k];
    most        = n_ones[k] > most ? n_ones[k] : most;
  }
  uint32_t nb = (most + 255) / 256;
  hipLaunchKernelGGL(k_index_init, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, st, a);
}
void launch_index_pass1(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t *cnt_read, uint32_t *anchor_first,
                        uint32_t V, uint32_t A, uint32_t *flags, uint32_t *err, IRow *bkt_row, uint32_t cap,
                        uint2 *spos) {
  if (n)
    hipLaunchKernelGGL(k_index_pass1, grid1(n, 256), dim3(256), 0, st, rows, n, cnt_read, anchor_first, flags, V, A, err,
                       bkt_row, cap, spos);
  hipLaunchKernelGGL(k_check_anchor_first, grid1(static_cast<uint64_t>(A) + 1, 256), dim3(256), 0, st, anchor_first, A,
                     static_cast<uint32_t>(n), flags);
}
void launch_scatter_read(hipStream_t st, const msgpu_row *rows, uint64_t n, const uint32_t *read_off, uint32_t *cursor,
                         IRow *bkt_row) {
  if (n)
    hipLaunchKernelGGL(k_scatter_read, grid1(n, 256), dim3(256), 0, st, rows, n, read_off, cursor, bkt_row);
}
void launch_sort_read(hipStream_t st, const uint32_t *read_off, const uint32_t *cnt_read, uint32_t V, const IRow *bkt_row,
                      IRow *by_read, uint32_t *read_cnt, uint32_t *alive_rank,
                      uint32_t *anchor_cnt, uint8_t *bkt_dead, uint32_t *flags, IRow *by_anchor, uint32_t cap,
                      const msgpu_row *rows, int32_t *read_len, uint32_t *read_first, uint32_t *err,
                      const uint2 *spos, uint4 *vis, uint32_t *visits) {
  if (V)
    hipLaunchKernelGGL(k_sort_read, grid1(V, 4), dim3(256), 0, st, read_off, cnt_read, V, bkt_row, by_read,
                       read_cnt, alive_rank, anchor_cnt, bkt_dead, flags, by_anchor, cap, rows, read_len, read_first, err,
                       spos, vis, visits);
}
void launch_index_finish(hipStream_t st, const uint32_t *read_first, uint32_t V, uint32_t *err, const uint32_t *flags,
                         const uint32_t *fast_off, const uint32_t *gen_off, uint32_t A, uint32_t *anchor_off, uint32_t *d_n_alive,
                         uint32_t n_rows) {
  const uint64_t most = V > static_cast<uint64_t>(A) + 1 ? V : static_cast<uint64_t>(A) + 1;
  hipLaunchKernelGGL(k_index_finish, grid1(most, 256), dim3(256), 0, st, read_first, V, err, flags, fast_off, gen_off, A, anchor_off,
                     d_n_alive, n_rows);
}
void launch_select_anchor_off(hipStream_t st, const uint32_t *flags, const uint32_t *fast_off, const uint32_t *gen_off,
                              uint32_t A, uint32_t *anchor_off, uint32_t *d_n_alive, uint32_t n_rows) {
  hipLaunchKernelGGL(k_select_anchor_off, grid1(static_cast<uint64_t>(A) + 1, 256), dim3(256), 0, st, flags, fast_off,
                     gen_off, A, anchor_off, d_n_alive, n_rows);
}
void launch_scatter_anchor(hipStream_t st, const msgpu_row *rows, uint64_t n, const uint32_t *alive_rank,
                           const uint32_t *anchor_off, uint32_t *cursor, uint32_t *bkt_idx, uint32_t *bkt_line,
                           const uint32_t *flags) {
  if (n)
    hipLaunchKernelGGL(k_scatter_anchor, grid1(n, 256), dim3(256), 0, st, rows, n, alive_rank, anchor_off, cursor,
                       bkt_idx, bkt_line, flags);
}
void launch_rank_anchor(hipStream_t st, const uint32_t *anchor_off, uint64_t n_rows, const uint32_t *d_n_alive,
                        const uint32_t *bkt_idx, const uint32_t *bkt_line, const msgpu_row *rows,
                        const uint32_t *alive_rank, IRow *by_anchor, const uint32_t *flags, const uint32_t *read_off,
                        IRow *by_read, uint4 *vis) {
  if (n_rows)
    hipLaunchKernelGGL(k_rank_anchor, grid1(n_rows, 256), dim3(256), 0, st, anchor_off, d_n_alive, bkt_idx, bkt_line,
                       rows, alive_rank, by_anchor, flags, read_off, by_read, vis);
}
void launch_bound(hipStream_t st, const uint32_t *read_off, const uint32_t *read_cnt, const uint4 *vis, uint32_t V,
                  uint32_t shard, uint32_t nshards, uint32_t lo, uint32_t hi, uint32_t *bound) {
  if (V)
    hipLaunchKernelGGL(k_bound, grid1(V, 16), dim3(256), 0, st, read_off, read_cnt, vis, V, shard, nshards, lo, hi,
                       bound);
}
void launch_classify_reads(hipStream_t st, const uint32_t *read_off, const uint32_t *read_cnt, const uint32_t *bound,
                           const uint64_t *cand_off, uint32_t V, uint32_t shard, uint32_t nshards, uint32_t lo,
                           uint32_t hi, CandDesc *l0, CandDesc *l1, CandDesc *l2, uint32_t *l3, uint32_t *n_lists, const CandZero &z,
                           unsigned long long *own_total) {
  if (V)
    hipLaunchKernelGGL(k_classify_reads, grid1(V, 1024), dim3(1024), 0, st, read_off, read_cnt, bound, cand_off, V, shard,
                       nshards, lo, hi, l0, l1, l2, l3, n_lists, z, own_total);
}
void launch_candidates(hipStream_t st, const CandArgs &a, int cls, const CandDesc *list, uint32_t n_list) {
  if (!n_list) return;
  if (cls == 0)
    hipLaunchKernelGGL((k_candidates<256, 512, 256>), dim3(n_list), dim3(256), 0, st, a, list, n_list);
  else if (cls == 1)
    hipLaunchKernelGGL((k_candidates<256, 1024, 256>), dim3(n_list), dim3(256), 0, st, a, list, n_list);
  else
    hipLaunchKernelGGL((k_candidates<1024, 4096, 256>), dim3(n_list), dim3(256), 0, st, a, list, n_list);
}
void launch_candidates_big(hipStream_t st, const CandArgs &a, const uint32_t *list, uint32_t n_list, uint64_t *big_key,
                           uint32_t *big_t, uint32_t *big_r2s, uint32_t *big_pfx) {
  if (!n_list) return;
  hipLaunchKernelGGL(k_candidates_big, dim3(n_list), dim3(256), 0, st, a, list, n_list, big_key, big_t, big_r2s,
                     big_pfx);
}
void launch_fill_pair_tab(hipStream_t st, uint32_t *tab) {
  hipLaunchKernelGGL(k_fill_pair_tab, dim3((4 * PAIR_TAB_STRIDE + 255) / 256), dim3(256), 0, st, tab);
}
void launch_chain(hipStream_t st, const ChainArgs &a, const uint32_t *list, uint32_t n_list) {
  const uint64_t n = list ? n_list : a.n_edges;
  if (n) hipLaunchKernelGGL(k_chain, dim3(static_cast<uint32_t>(n)), dim3(64), CHAIN_LDS_BYTES, st, a, list, n_list); // a wavefront per workgroup
}
void launch_chain_sub(hipStream_t st, const ChainArgs &a, int width, const uint32_t *list, uint32_t n_list) {
  if (!n_list) return;
  if (width == 8)
    hipLaunchKernelGGL(k_chain_sub<8>, grid1(n_list, 32), dim3(256), 0, st, a, list, n_list);
  else if (width == 16)
    hipLaunchKernelGGL(k_chain_sub<16>, grid1(n_list, 16), dim3(256), 0, st, a, list, n_list);
  else
    hipLaunchKernelGGL(k_chain_sub<32>, grid1(n_list, 8), dim3(256), 0, st, a, list, n_list);
}
void launch_chain_sub_all(hipStream_t st, const ChainArgs &a, const uint32_t *l32, uint32_t n32, const uint32_t *l16, uint32_t n16,
                          const uint32_t *l8, uint32_t n8) {
  ChainSubLists l;
  l.list32 = l32, l.list16 = l16, l.list8 = l8;
  l.n32 = n32, l.n16 = n16, l.n8 = n8;
  l.nb32 = n32 ? grid1(n32, 8).x : 0u, l.nb16 = n16 ? grid1(n16, 16).x : 0u;
  const uint32_t nb8 = n8 ? grid1(n8, 32).x : 0u, nb = l.nb32 + l.nb16 + nb8;
  if (nb) hipLaunchKernelGGL(k_chain_sub_all, dim3(nb), dim3(256), 0, st, a, l);
}
size_t big_elem_bytes() { return sizeof(BigElem); }
size_t big_path_bytes() { return sizeof(BigPath); }
void launch_chain_big(hipStream_t st, const ChainArgs &a, const uint32_t *big_list, const uint64_t *big_off,
                      uint32_t n_big, void *elems, void *paths) {
  if (n_big)
    hipLaunchKernelGGL(k_chain_big, dim3(n_big), dim3(64), 0, st, a, big_list, big_off, n_big,
                       static_cast<BigElem *>(elems), static_cast<BigPath *>(paths));
}
void launch_merge_gathered(hipStream_t st, const MergeArgs &a) {
  uint64_t n = a.base[a.world].edges;
  if (a.base[a.world].orders > n) n = a.base[a.world].orders;
  if (a.base[a.world].ids > n) n = a.base[a.world].ids;
  if (n) hipLaunchKernelGGL(k_merge_gathered, grid1(n, 256), dim3(256), 0, st, a);
}
void launch_merge_wire(hipStream_t st, const MergeArgs &a, bool ids3) {
  uint64_t n = a.base[a.world].edges;
  if (a.base[a.world].orders > n) n = a.base[a.world].orders;
  if (a.base[a.world].ids > n) n = a.base[a.world].ids;
  if (!n) return;
  if (ids3)
    hipLaunchKernelGGL(k_merge_wire<true>, grid1(n, 256), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_merge_wire<false>, grid1(n, 256), dim3(256), 0, st, a);
}
void launch_pack_wire(hipStream_t st, const PackWireArgs &a) {
  uint64_t n = a.n_edges > a.n_orders ? a.n_edges : a.n_orders;
  if (a.ids && (a.n_ids + 3) / 4 > n) n = (a.n_ids + 3) / 4;
  if (!n) n = 1; // empty tables: thread 0 writes the closing CSR entries
  hipLaunchKernelGGL(k_pack_wire, grid1(n, 256), dim3(256), 0, st, a);
}
void launch_compact(hipStream_t st, const CompactArgs &a) { // a workgroup per chunk; one even without edges: it publishes the sizes
  hipLaunchKernelGGL(k_compact, dim3(a.n_chunks ? a.n_chunks : 1), dim3(1024), 0, st, a);
}

} // namespace msgpu
