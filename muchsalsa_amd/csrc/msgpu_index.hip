// msgpu_index.hip -- the index build's bin path (gfx950 / CDNA4, wave64): rows (40 B, grouped by anchor with ascending
// lines: what msgpu_parse_paf hands over) -> by_read, by_anchor, the scan view and the per-read Vertex facts WITHOUT a
// global atomic per row.
//
// Why (tools/micro/index_limits.hip, MI355X, the 5.06 M rows of BASELINE.json configs[2]): a global atomic on a counter
// chosen by the row's read id -- read ids are unrelated to the row's place in the file -- runs at 24 G atomics/s whether it
// returns a value or not (every lane of a wave instruction is a memory-side request of its own): 206 us, which was
// k_index_pass1's 220 us through round 3.  Here a 2048-row tile (BIN_TILE) is binned by COARSE bucket (16 consecutive read ids) with
// LDS atomics, the tile's rows of a bucket are reserved with ONE global atomic per (tile, bucket) on adjacent counters (a
// wave instruction covers 64 neighbouring counters: merged requests), and a workgroup per bucket stages its ~10^3 rows in
// LDS -- read once, coalesced -- grouped by read, then ranks every read's rows in registers as k_sort_read does (rank by
// (nanoporeRange, anchor): mpp.cpp:164-172 / :259-267).  read_off comes from a scan of the bucket counts, not from per-read
// counters.  What is left per row is what the tables themselves cost: one scattered whole-sector store into the bucket, one
// scattered 32-byte store into by_anchor (a scattered store of <= 16 bytes costs 60 us per 5 M rows, of 32 bytes 80 us: a
// read-modify-write of a sector, whatever the lanes do -- same micro-benchmark).
//
// Same tables as the atomic path, bit for bit (the sort keys are unique; the arrival order inside a bucket is no input of
// anything).  What the path does not cover -- rows not grouped by anchor, a duplicate (read, anchor) pair (MatchMap.cpp:64-80:
// lowest line wins), a read with more than 256 rows, a bucket beyond its capacity -- raises IXF_BINFAIL (or one of pass 1's
// flags) and the host rebuilds with the atomic path, which covers every input.  More than BIN_NB_MAX * 16 reads: several
// passes over the row table, one per range of read ids.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "msgpu.h"
#include "msgpu_internal.h"
#include "msgpu_device.h"

namespace msgpu {

// Diagnostic build only (tools/micro/index_stamps.hip compiles this file with -DMSGPU_STAMPS): s_memtime at the phase
// boundaries of the two kernels, one row of stamps per workgroup, into a buffer nothing else reads.  The product build has
// no stamp in it.
#ifdef MSGPU_STAMPS
__device__ unsigned long long *g_stamps = nullptr; // [workgroups][8]
#define STAMP(k)                                                                                                       \
  do {                                                                                                                 \
    if (threadIdx.x == 0 && g_stamps) {                                                                                \
      unsigned long long t_;                                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                       \
      g_stamps[static_cast<size_t>(blockIdx.x) * 8 + (k)] = t_;                                                        \
    }                                                                                                                  \
  } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

constexpr int      BIN_PT    = 4;                  // rows per thread of k_index_bin
constexpr int      BIN_NT    = 512;
constexpr int      BIN_TILE  = BIN_PT * BIN_NT;    // 2048 rows per workgroup
constexpr int      BIN_HALO  = 128;                // rows of context on either side (= SCAF_HALO of the atomic path)
constexpr uint32_t BIN_PACK_BEHIND_SHIFT = 9;      // record word 9 = (lower - before + 128) | behind << 9
constexpr int      BIN_SORT_NT = 512;              // threads of k_index_sort_bin: 8 wavefronts share a bucket's reads

// A bucket record is ONE 64-byte sector: [0] n_lo n_hi i_lo i_hi | [1] score line anchor flags|source row | [2] read id,
// packed scaffold place / rows behind, 0, 0 | [3] zeros.  A scattered store of a whole sector costs the memory system one
// write; a 32-byte row plus 8 bytes elsewhere are two read-modify-writes (tools/micro/index_limits.hip: 100 us against
// 130 + 60 for the 5.06 M rows) -- provided ONE wave instruction writes the whole sector: four neighbouring lanes store its
// four 16-byte pieces (the pieces travel through a per-wavefront LDS tile); four stores of one lane are four requests (177 us).

// ---- pass 1: scaffold places (as k_index_pass1), input-order checks, rows binned by coarse bucket ---------------------
// Reads [rd_lo, rd_lo + nb * 16) are binned by this launch; the checks on the input's order run in the launch with
// rd_lo == 0 only.
__global__ __launch_bounds__(BIN_NT) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_index_bin(const msgpu_row *rows, uint64_t n, uint32_t V, uint32_t A, uint32_t *flags, uint32_t *err, uint32_t *anchor_first,
                 uint32_t *cursor /*[nb]*/, uint4 *bin_rec, uint32_t rd_lo, uint32_t nb, uint32_t cap, BinTail tail) {
  __shared__ uint32_t s_an[BIN_TILE + 2 * BIN_HALO], s_rd[BIN_TILE + 2 * BIN_HALO];
  __shared__ uint32_t s_heads, s_last;
  __shared__ uint32_t s_cnt[BIN_NB_MAX];
  __shared__ unsigned long long s_head[(BIN_TILE + 2 * BIN_HALO) / 64]; // bit t: position t begins a scaffold (its anchor differs from t - 1's)
  __shared__ uint4    s_stage[BIN_NT / 64][64][3];
  __shared__ uint32_t s_slot[BIN_NT / 64][64];
  const uint64_t i0    = static_cast<uint64_t>(blockIdx.x) * BIN_TILE;
  const uint32_t rd_hi = min(V, rd_lo + (nb << BIN_RPB_SHIFT));
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t b = threadIdx.x; b < nb; b += BIN_NT) s_cnt[b] = 0;
  if (threadIdx.x == 0) s_heads = 0;
  STAMP(0);
  // the tile's rows: every load is in flight before the first one is used (a row is read once)
  msgpu_row row[BIN_PT];
#pragma unroll
  for (int k = 0; k < BIN_PT; ++k) {
    const uint64_t i = i0 + static_cast<uint64_t>(k) * BIN_NT + threadIdx.x;
    row[k]           = msgpu_row{};
    row[k].anchor_id = 0xffffffffu; // no valid anchor id (ids are < A <= 2^32 - 1)
    if (i < n) row[k] = rows[i];
  }
  uint2 halo = make_uint2(0xffffffffu, 0u); // (anchor, read) of BIN_HALO rows of context on either side
  if (threadIdx.x < 2 * BIN_HALO) {
    const int       slot = threadIdx.x < BIN_HALO ? static_cast<int>(threadIdx.x) : BIN_TILE + static_cast<int>(threadIdx.x);
    const long long g    = static_cast<long long>(i0) - BIN_HALO + slot;
    if (g >= 0 && static_cast<uint64_t>(g) < n) halo = *reinterpret_cast<const uint2 *>(&rows[g]); // anchor_id, read_id
  }
#pragma unroll
  for (int k = 0; k < BIN_PT; ++k) {
    s_an[BIN_HALO + k * BIN_NT + threadIdx.x] = row[k].anchor_id;
    s_rd[BIN_HALO + k * BIN_NT + threadIdx.x] = row[k].read_id;
  }
  if (threadIdx.x < 2 * BIN_HALO) {
    const int slot = threadIdx.x < BIN_HALO ? static_cast<int>(threadIdx.x) : BIN_TILE + static_cast<int>(threadIdx.x);
    s_an[slot] = halo.x;
    s_rd[slot] = halo.y;
  }
  __syncthreads();
  STAMP(1);
  // where scaffolds begin, as bit masks (a wavefront covers 64 consecutive positions: one ballot per word).  Position 0 counts
  // as a beginning: a scaffold that reaches it is longer than the context and is reported as such below.
  static_assert((BIN_TILE + 2 * BIN_HALO) % 64 == 0 && BIN_HALO % 64 == 0, "whole words");
  uint32_t my_heads = 0; // scaffolds that begin among this tile's own rows (lane 0 of a wavefront counts its words)
  for (int t = threadIdx.x; t < BIN_TILE + 2 * BIN_HALO; t += BIN_NT) {
    const bool               head = t == 0 || s_an[t] != s_an[t - 1];
    const unsigned long long m    = __ballot(head);
    if (lane == 0) s_head[t >> 6] = m;
    const bool own = t >= BIN_HALO && t < BIN_HALO + BIN_TILE && i0 + static_cast<uint64_t>(t - BIN_HALO) < n;
    my_heads += static_cast<uint32_t>(__popcll(__ballot(head && own)));
  }
  // IXF_SPARSE without a filled table to look into: the rows are grouped by anchor with ascending ids (else IXF_UNSORTED), so
  // the scaffolds that begin are the distinct anchor ids -- all A of them have a row iff A scaffolds begin (k_index_epilogue)
  if (lane == 0 && rd_lo == 0 && my_heads) atomicAdd(&s_heads, my_heads);
  uint32_t bk[BIN_PT], lr[BIN_PT];
#pragma unroll
  for (int k = 0; k < BIN_PT; ++k) {
    const uint64_t i = i0 + static_cast<uint64_t>(k) * BIN_NT + threadIdx.x;
    bk[k] = 0xffffffffu;
    lr[k] = 0;
    if (i >= n) continue;
    const uint32_t rd = row[k].read_id, an = row[k].anchor_id;
    if (rd >= V || an >= A) { // only possible when the host declared the id space (msgpu_set_id_space)
      if (rd_lo == 0) atomicOr(err, 2u);
      continue;
    }
    if (rd < rd_lo || rd >= rd_hi) continue; // another pass's read
    bk[k] = (rd - rd_lo) >> BIN_RPB_SHIFT;
    lr[k] = atomicAdd(&s_cnt[bk[k]], 1u); // LDS: rank inside (tile, bucket)
  }
  __syncthreads();
  STAMP(2);
  // one global atomic per (tile, non-empty bucket); a wave instruction covers 64 adjacent counters (merged requests)
  // (all of a thread's atomics are in flight together: a returning atomic takes microseconds at the memory side)
  {
    constexpr int NA = BIN_NB_MAX / BIN_NT;
    uint32_t      cc[NA], base[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const uint32_t b = q * BIN_NT + threadIdx.x;
      cc[q]            = b < nb ? s_cnt[b] : 0u;
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      base[q] = 0;
      if (cc[q]) base[q] = atomicAdd(&cursor[q * BIN_NT + threadIdx.x], cc[q]);
    }
#pragma unroll
    for (int q = 0; q < NA; ++q)
      if (cc[q]) s_cnt[q * BIN_NT + threadIdx.x] = base[q];
  }
  __syncthreads();
  STAMP(3);
#pragma unroll
  for (int k = 0; k < BIN_PT; ++k) {
    const uint64_t i = i0 + static_cast<uint64_t>(k) * BIN_NT + threadIdx.x; // (every lane runs the iteration: the stores below are the wavefront's)
    const int      c = BIN_HALO + k * BIN_NT + static_cast<int>(threadIdx.x);
    uint32_t       slot = 0xffffffffu;
    if (i < n) {
      const uint32_t an = row[k].anchor_id, rd = row[k].read_id;
      if (rd_lo == 0 && an < A && rd < V) { // the input's order (once per build)
        if (i == 0) {
          anchor_first[an] = 0;
        } else {
          const uint32_t pa = s_an[c - 1];
          if (pa > an || (pa == an && rows[i - 1].line >= row[k].line)) atomicOr(flags, IXF_UNSORTED);
          if (pa != an) anchor_first[an] = static_cast<uint32_t>(i);
        }
      }
      if (bk[k] != 0xffffffffu) {
        // the scaffold of this row = [first, last]: the beginning at or in front of c, the one behind c (bit scans over at
        // most three words each: BIN_HALO = 128 positions); then ONE loop over the scaffold counts the lower read ids
        const int w = c >> 6, bit = c & 63;
        int       first, last;
        {
          unsigned long long m = s_head[w] & (~0ull >> (63 - bit));
          int                ww = w;
          if (!m && ww > 0) m = s_head[--ww];
          if (!m && ww > 0) m = s_head[--ww];
          first = m ? ww * 64 + 63 - __builtin_clzll(m) : 0;
          m  = bit == 63 ? 0ull : s_head[w] & (~0ull << (bit + 1));
          ww = w;
          constexpr int NW = (BIN_TILE + 2 * BIN_HALO) / 64;
          if (!m && ww + 1 < NW) m = s_head[++ww];
          if (!m && ww + 1 < NW) m = s_head[++ww];
          last = (m ? ww * 64 + __builtin_ctzll(m) : BIN_TILE + 2 * BIN_HALO) - 1;
        }
        const uint32_t before = static_cast<uint32_t>(c - first), after = static_cast<uint32_t>(last - c);
        if (before >= BIN_HALO || after >= BIN_HALO) atomicOr(flags, IXF_BIGSCAF); // (= the atomic path's test: 128 equal anchors on a side)
        uint32_t lower = 0;
        bool     twin  = false;
        for (int p = max(first, c - BIN_HALO); p <= min(last, c + BIN_HALO); ++p) {
          const uint32_t o = s_rd[p];
          lower += o < rd ? 1u : 0u;
          twin |= (o == rd) & (p != c);
        }
        if (twin) atomicOr(flags, IXF_DUPS); // a second row of this (read, anchor) pair (MatchMap.cpp:64-80): the atomic path sorts that out
        // place in the (read-id-sorted) scaffold = i - before + lower; scaffold rows behind it = before + after - lower
        const uint32_t packed = (lower + BIN_HALO - before) | ((before + after - lower) << BIN_PACK_BEHIND_SHIFT);
        const uint32_t pos    = s_cnt[bk[k]] + lr[k];
        if (pos < cap) {
          slot                    = bk[k] * cap + pos; // (nb * cap records: below 2^32, checked by the host)
          const IRow r            = make_irow(row[k], an, static_cast<uint32_t>(i));
          s_stage[wave][lane][0] = make_uint4(static_cast<uint32_t>(r.n_lo), static_cast<uint32_t>(r.n_hi),
                                              static_cast<uint32_t>(r.i_lo), static_cast<uint32_t>(r.i_hi));
          s_stage[wave][lane][1] = make_uint4(r.score, r.line, r.other, r.pf);
          s_stage[wave][lane][2] = make_uint4(rd, packed, 0u, 0u);
        } else {
          atomicOr(flags, IXF_BINFAIL);
        }
      }
    }
    s_slot[wave][lane] = slot;
    // the wavefront's 64 records leave as whole sectors: lane l stores piece (l & 3) of the record of lane 16 j + (l >> 2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int      src = 16 * j + (lane >> 2), piece = lane & 3;
      const uint32_t sl  = s_slot[wave][src];
      if (sl != 0xffffffffu)
        bin_rec[static_cast<uint64_t>(sl) * 4 + piece] = piece < 3 ? s_stage[wave][src][piece] : make_uint4(0u, 0u, 0u, 0u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the tile's reads are done before the next iteration's writes
    __builtin_amdgcn_wave_barrier();
  }
#ifdef MSGPU_STAMPS
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(4);
#endif
  // ---- the LAST workgroup to get here turns the bucket counts into the first by_read row of every bucket (what used to be
  // k_bin_scan's launch): every workgroup's counter updates are out (they returned values) and ordered in front of its ticket
  // (No fence: an agent-scope release writes the L2 back on this part -- it doubled the kernel's time, profiles/r5_04.  None
  // is needed: the counters were updated with atomics whose values came back, so they are done; the ticket is one more atomic
  // behind them in program order; the tail reads the counters with agent-scope loads.)
  __syncthreads();
  if (threadIdx.x == 0) {
    // one word: finished workgroups in the low half, scaffolds that began (first pass) in the high half
    const unsigned long long t = atomicAdd(tail.done_heads, (static_cast<unsigned long long>(rd_lo == 0 ? s_heads : 0u) << 32) | 1ull);
    s_last = static_cast<uint32_t>(t) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  {
    uint32_t *s_wave = s_slot[0]; // (idle by now)
    constexpr int PT = BIN_NB_MAX / BIN_NT;
    uint32_t      c[PT], sum = 0;
    bool          over = false;
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const uint32_t b = threadIdx.x * PT + k;
      c[k]             = b < nb ? __hip_atomic_load(&cursor[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      over |= c[k] > cap;
      sum += c[k];
    }
    uint32_t       total;
    const uint32_t rb   = *tail.row_base;
    uint32_t       base = block_excl_scan<BIN_NT>(sum, s_wave, &total) + rb;
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const uint32_t b = threadIdx.x * PT + k;
      if (b < nb) tail.bin_start[b] = base;
      base += c[k];
    }
    if (threadIdx.x == 0) {
      const uint32_t end = rb + total;
      tail.bin_start[nb] = end;
      *tail.row_base     = end;
      if (tail.read_off_end) *tail.read_off_end = end;
      *reinterpret_cast<uint32_t *>(tail.done_heads) = 0; // the low half: zero at rest, the next launch counts from nothing
    }
    if (over) atomicOr(flags, IXF_BINFAIL);
  }
}

// ---- pass 2: one workgroup per coarse bucket -----------------------------------------------------------------------------
// The bucket's records are read once (whole sectors, 16 bytes per lane) into LDS; an index grouped by read is built there; then
// a wavefront per read ranks its rows in registers (K rows per lane, every row broadcast once by readlane: the fast branch
// of k_sort_read / sort_read_in_registers).
struct BinLds { // views into the dynamic LDS block, [cap] each
  uint4    *row_a, *row_b; // the two halves of the 32-byte rows, in arrival order
  uint32_t *aux;           // packed scaffold place / rows behind (18 bits) | the read's number inside the bucket << 18
  uint16_t *idx;           // arrival positions grouped by read
};

// value of lane ^ J: quad permutes for 1 and 2, ds_swizzle (bit mode: and 0x1f, xor J; no LDS memory touched) for 4, 8, 16,
// a shuffle for 32
template <int J> __device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if (J == 1) return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xf, 0xf, false)); // quad_perm:[1,0,3,2]
  if (J == 2) return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xf, 0xf, false)); // quad_perm:[2,3,0,1]
  if (J == 4) return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x101f));
  if (J == 8) return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x201f));
  if (J == 16) return static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x401f));
  return static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), 32));
}
// lanes that keep the SMALLER key of their pair in the stage (block size KB, distance J): the lower lane of a pair inside an
// ascending block (bit KB of the lane clear; the last merge, KB = 64, is ascending throughout), the upper lane inside a
// descending one
constexpr unsigned long long bitonic_min_mask(int KB, int J) {
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    const bool asc = KB == 64 || (l & KB) == 0, lower = (l & J) == 0;
    if (lower == asc) m |= 1ull << l;
  }
  return m;
}
template <int KB, int J>
__device__ __forceinline__ void bitonic_stage(uint32_t &khi, uint32_t &klo, uint32_t &org) {
  constexpr unsigned long long MIN = bitonic_min_mask(KB, J);
  const uint32_t ph = lane_xor<J>(khi), pl = lane_xor<J>(klo), po = lane_xor<J>(org);
  const uint64_t mine = (static_cast<uint64_t>(khi) << 32) | klo, other = (static_cast<uint64_t>(ph) << 32) | pl;
  // both lanes of a pair reach the same verdict: the one that keeps the minimum takes a smaller partner, the other one a larger
  // partner; equal keys stay where they are (and are found afterwards)
  const unsigned long long take = (__ballot(other < mine) & MIN) | (__ballot(mine < other) & ~MIN);
  const bool               t    = __builtin_amdgcn_inverse_ballot_w64(take);
  khi = t ? ph : khi;
  klo = t ? pl : klo;
  org = t ? po : org;
}
__device__ __forceinline__ void bitonic_sort64(uint32_t &khi, uint32_t &klo, uint32_t &org, int) {
  bitonic_stage<2, 1>(khi, klo, org);
  bitonic_stage<4, 2>(khi, klo, org);
  bitonic_stage<4, 1>(khi, klo, org);
  bitonic_stage<8, 4>(khi, klo, org);
  bitonic_stage<8, 2>(khi, klo, org);
  bitonic_stage<8, 1>(khi, klo, org);
  bitonic_stage<16, 8>(khi, klo, org);
  bitonic_stage<16, 4>(khi, klo, org);
  bitonic_stage<16, 2>(khi, klo, org);
  bitonic_stage<16, 1>(khi, klo, org);
  bitonic_stage<32, 16>(khi, klo, org);
  bitonic_stage<32, 8>(khi, klo, org);
  bitonic_stage<32, 4>(khi, klo, org);
  bitonic_stage<32, 2>(khi, klo, org);
  bitonic_stage<32, 1>(khi, klo, org);
  bitonic_stage<64, 32>(khi, klo, org);
  bitonic_stage<64, 16>(khi, klo, org);
  bitonic_stage<64, 8>(khi, klo, org);
  bitonic_stage<64, 4>(khi, klo, org);
  bitonic_stage<64, 2>(khi, klo, org);
  bitonic_stage<64, 1>(khi, klo, org);
}

template <int K>
__device__ __forceinline__ bool sort_bin_read(uint32_t r, uint32_t n, uint32_t b /*first by_read row*/, uint32_t off /*first LDS row*/,
                                              int lane, const BinLds &s, IRow *by_read, IRow *by_anchor, uint4 *vis,
                                              unsigned long long *first_key /*LDS*/, uint32_t *read_cnt, uint32_t *visits) {
  IRow     row[K];
  uint32_t pk[K], man[K], less[K];
  int      mlo[K], mhi[K];
  bool     dup = false;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const uint32_t e = static_cast<uint32_t>(k) * 64 + lane;
    row[k]           = IRow{};
    pk[k]            = 0;
    if (e < n) {
      const uint32_t at = s.idx[off + e];
      const uint4    a = s.row_a[at], c = s.row_b[at];
      row[k].n_lo   = static_cast<int>(a.x);
      row[k].n_hi   = static_cast<int>(a.y);
      row[k].i_lo   = static_cast<int>(a.z);
      row[k].i_hi   = static_cast<int>(a.w);
      row[k].score  = c.x;
      row[k].line   = c.y;
      row[k].other  = c.z;
      row[k].pf     = c.w;
      pk[k]         = s.aux[at] & 0x3ffffu;
    }
    mlo[k]  = e < n ? row[k].n_lo : 0x7fffffff;
    mhi[k]  = e < n ? row[k].n_hi : 0x7fffffff;
    man[k]  = e < n ? row[k].other : 0xffffffffu;
    less[k] = 0;
  }
  bool ranked = false;
  if (K == 1) {
    // One row per lane: SORT (nanopore range as one order-preserving 64-bit key, original lane as payload) with a bitonic
    // network over the wavefront -- 21 compare-exchange stages of two compares and three selects, partners through DPP /
    // ds_swizzle -- instead of broadcasting every row to every lane (n x (2 readlanes + 2 compares + add)): the broadcast
    // loop was what bounded this kernel (about 1,000 instructions per read).  Lanes without a row hold the largest key and
    // sort to the end.  Two rows with the same range (the anchor decides then) show up as equal neighbours afterwards: the
    // broadcast loop below ranks such a read.  Duplicate (read, anchor) pairs are pass 1's to find.
    uint32_t khi = static_cast<uint32_t>(mlo[0]) ^ 0x80000000u, klo = static_cast<uint32_t>(mhi[0]) ^ 0x80000000u;
    if (static_cast<uint32_t>(lane) >= n) khi = klo = 0xffffffffu;
    uint32_t org = static_cast<uint32_t>(lane);
    bitonic_sort64(khi, klo, org, lane);
    const uint32_t nhi = static_cast<uint32_t>(__shfl_down(static_cast<int>(khi), 1)), nlo = static_cast<uint32_t>(__shfl_down(static_cast<int>(klo), 1));
    const bool     tie = lane < 63 && static_cast<uint32_t>(lane) < n && nhi == khi && nlo == klo; // (lane n - 1 against a lane without a row: a row with the largest key)
    ranked = __ballot(tie) == 0;
    // lane p holds the row that came from lane org: that row's rank is p
    if (ranked) less[0] = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>(org << 2), lane));
  }
  if (!ranked)
#pragma unroll
  for (int sx = 0; sx < K; ++sx) {
    const int cnt = min(64, static_cast<int>(n) - 64 * sx); // wave-uniform
    for (int t = 0; t < cnt; ++t) {
      const int      olo = rl_i32(mlo[sx], t), ohi = rl_i32(mhi[sx], t);
      const uint32_t oan = rl_u32(man[sx], t);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        less[k] += key_less(olo, ohi, oan, mlo[k], mhi[k], man[k]) ? 1u : 0u;
        dup |= ((sx != k) | (t != lane)) & (oan == man[k]);
      }
    }
  }
  if (__ballot(dup)) return false; // sentinel rows (anchor 0xffffffff) are never broadcast, so they cannot match
  {
    unsigned long long fk = ~0ull;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (static_cast<uint32_t>(k) * 64 + lane < n)
        fk = min(fk, (static_cast<unsigned long long>(row[k].line) << 32) | (row[k].pf & PF_POS_MASK));
    // the read's first line (Graph.cpp:148: the Vertex is made there).  Its length is fetched for all the bucket's reads
    // together at the end (a dependent global load here would stall the wavefront in front of its next read).
    for (int d = 32; d > 0; d >>= 1) {
      const unsigned long long o =
          (static_cast<unsigned long long>(static_cast<uint32_t>(__shfl_xor(static_cast<int>(fk >> 32), d))) << 32) |
          static_cast<uint32_t>(__shfl_xor(static_cast<int>(fk), d));
      fk = o < fk ? o : fk;
    }
    if (lane == 0) *first_key = fk;
  }
  uint32_t behind = 0; // scaffold rows behind this read's own rows = the visits of its candidate scan
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (static_cast<uint32_t>(k) * 64 + lane < n) {
      const uint32_t idx = row[k].pf & PF_POS_MASK; // source row
      const uint32_t sp  = idx + (pk[k] & ((1u << BIN_PACK_BEHIND_SHIFT) - 1)) - BIN_HALO, bh = pk[k] >> BIN_PACK_BEHIND_SHIFT;
      IRow           w   = row[k];
      w.other            = r;
      w.pf               = (row[k].pf & ~PF_POS_MASK) | less[k]; // scaffold rows carry the rank inside their read
      store_irow(&by_anchor[sp], w);
      row[k].pf = (row[k].pf & ~PF_POS_MASK) | sp; // by_read rows carry their place in the scaffold
      vis[b + less[k]] = make_uint4(static_cast<uint32_t>(row[k].i_lo), static_cast<uint32_t>(row[k].i_hi), sp + 1, bh);
      store_irow(&by_read[b + less[k]], row[k]);
      behind += bh;
    }
  }
  behind = wave_sum(behind);
  if (lane == 0) {
    read_cnt[r] = n;
    visits[r]   = behind;
  }
  return true;
}

__global__ __launch_bounds__(BIN_SORT_NT) void k_index_sort_bin(uint32_t *cursor, const uint32_t *bin_start, uint32_t V,
                                                                uint32_t rd_lo, uint32_t cap, const uint4 *bin_rec, IRow *by_read,
                                                                IRow *by_anchor, uint4 *vis, uint32_t *read_off, uint32_t *read_cnt,
                                                                int32_t *read_len, uint32_t *read_first, uint32_t *visits,
                                                                const msgpu_row *rows, uint32_t *flags, uint32_t *err,
                                                                uint32_t *bucket_visits /*[all buckets of the job]*/) {
  constexpr uint32_t RPB = 1u << BIN_RPB_SHIFT;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  __shared__ uint32_t s_rcnt[RPB], s_roff[RPB + 1], s_fill[RPB], s_stop;
  __shared__ unsigned long long s_first[RPB]; // per read: line << 32 | source row of its first line
  // pass 1 refused the input, or a bucket ran over: the host rebuilds.  (One thread reads the word for the workgroup: other
  // workgroups of this launch may be raising IXF_BINFAIL at this moment, and the barriers below need every thread.)
  if (threadIdx.x == 0) s_stop = (*flags & ~IXF_DUPS) != 0 ? 1u : 0u;
  if (threadIdx.x < RPB) s_rcnt[threadIdx.x] = 0;
  __syncthreads();
  if (s_stop) return;
  STAMP(0);
  BinLds s;
  s.row_a = reinterpret_cast<uint4 *>(s_dyn);
  s.row_b = s.row_a + cap;
  s.aux   = reinterpret_cast<uint32_t *>(s.row_b + cap);
  s.idx   = reinterpret_cast<uint16_t *>(s.aux + cap);
  const uint32_t b    = blockIdx.x;
  const uint32_t r0   = rd_lo + (b << BIN_RPB_SHIFT);
  const uint32_t nr   = min(RPB, V - r0);
  const uint32_t n_b  = min(cursor[b], cap);
  const uint4   *rec  = bin_rec + static_cast<uint64_t>(b) * cap * 4;
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the bucket: one coalesced pass over its sectors, eight loads per thread in flight (a thread always meets the same piece:
  // BIN_SORT_NT is a multiple of 4)
  {
    const uint32_t piece = threadIdx.x & 3;
    for (uint32_t q0 = 0; q0 < 4 * n_b; q0 += 8 * BIN_SORT_NT) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t q = q0 + u * BIN_SORT_NT + threadIdx.x;
        if (piece != 3 && q < 4 * n_b) v[u] = rec[q];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t q = q0 + u * BIN_SORT_NT + threadIdx.x, e = q >> 2;
        if (piece == 3 || q >= 4 * n_b) continue;
        if (piece == 0) {
          s.row_a[e] = v[u];
        } else if (piece == 1) {
          s.row_b[e] = v[u];
        } else {
          s.aux[e] = v[u].y | ((v[u].x - r0) << 18);
          atomicAdd(&s_rcnt[v[u].x - r0], 1u);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) cursor[b] = 0; // zero at rest: every thread has read the count; the next build bins from nothing
  STAMP(1);
  if (threadIdx.x < 64) { // RPB <= 64 reads: one wavefront scans the counts
    const uint32_t c   = lane < static_cast<int>(nr) ? s_rcnt[lane] : 0u;
    const uint32_t inc = wave_incl_scan(c);
    if (lane < static_cast<int>(RPB)) {
      s_roff[lane] = inc - c;
      s_fill[lane] = inc - c;
    }
    if (lane == static_cast<int>(RPB) - 1) s_roff[RPB] = inc;
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < n_b; e += BIN_SORT_NT) s.idx[atomicAdd(&s_fill[s.aux[e] >> 18], 1u)] = static_cast<uint16_t>(e);
  __syncthreads();
  STAMP(2);
  const uint32_t start = bin_start[b];
  for (uint32_t q = wave; q < nr; q += BIN_SORT_NT / 64) { // a wavefront per read
    const uint32_t r = r0 + q, n = s_rcnt[q], off = s_roff[q];
    if (lane == 0) read_off[r] = start + off;
    if (n == 0) { // an id without any row: ids are not Registry-dense (Registry.cpp:36-45)
      if (lane == 0) {
        s_first[q]  = ~0ull;
        read_cnt[r] = 0;
        visits[r]   = 0;
        atomicOr(err, 1u);
      }
      continue;
    }
    bool ok;
    if (n <= 64)
      ok = sort_bin_read<1>(r, n, start + off, off, lane, s, by_read, by_anchor, vis, &s_first[q], read_cnt, visits);
    else if (n <= 128)
      ok = sort_bin_read<2>(r, n, start + off, off, lane, s, by_read, by_anchor, vis, &s_first[q], read_cnt, visits);
    else if (n <= 256)
      ok = sort_bin_read<4>(r, n, start + off, off, lane, s, by_read, by_anchor, vis, &s_first[q], read_cnt, visits);
    else
      ok = false;
    if (!ok && lane == 0) {
      s_first[q] = ~0ull;
      atomicOr(flags, IXF_BINFAIL); // a duplicate (read, anchor) pair or a very long read: the atomic path covers them
    }
  }
  __syncthreads();
  STAMP(3);
  if (threadIdx.x < nr) { // the Vertex facts of the bucket's reads: first line, and the read length that line states
    const unsigned long long fk = s_first[threadIdx.x];
    read_first[r0 + threadIdx.x] = static_cast<uint32_t>(fk >> 32);
    read_len[r0 + threadIdx.x]   = fk == ~0ull ? 0 : rows[static_cast<uint32_t>(fk)].read_len;
  }
  // the bucket's share of the candidate scan's visits: k_index_epilogue scans the per-read counts with these as its carries.
  // (`visits` of this bucket's reads were written by this workgroup's wavefronts in front of the barrier above: an agent-scope
  // load sees them whatever cache the stores went through.)
  if (threadIdx.x < 64) {
    uint32_t v = threadIdx.x < nr ? __hip_atomic_load(&visits[r0 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
    v          = wave_sum(v);
    if (threadIdx.x == 0) bucket_visits[r0 >> BIN_RPB_SHIFT] = v;
  }
#ifdef MSGPU_STAMPS
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(4);
#endif
}

// ---- launchers -----------------------------------------------------------------------------------------------------------

// rows a coarse bucket can hold: the mean with a quarter of slack + a few standard deviations of a sum of per-read counts;
// 0 = more than k_index_sort_bin can stage in LDS (96 KB: 2340 rows) or address: the bin path does not apply
constexpr uint32_t BIN_LDS_PER_ROW = 38; // 32 (row) + 4 (side data) + 2 (index)
uint32_t bin_capacity(uint64_t n, uint32_t V) {
  const uint64_t n_buckets = (static_cast<uint64_t>(V) + (1u << BIN_RPB_SHIFT) - 1) >> BIN_RPB_SHIFT;
  const uint64_t mean      = n_buckets ? (n + n_buckets - 1) / n_buckets : 0;
  uint64_t       root      = 1;
  while (root * root < mean) ++root;
  // mean + 1/8 + six standard deviations of a Poisson-like sum (+ room for tiny jobs).  Read lengths that vary by more than
  // that between neighbouring buckets overflow a bucket: IXF_BINFAIL, the atomic path takes the job.
  const uint64_t cap = (mean + mean / 8 + 6 * root + 48 + 7) & ~7ull; // (a multiple of 8: the LDS views stay 16-byte aligned)
  if (cap * BIN_LDS_PER_ROW > (96u << 10) || cap * std::min<uint64_t>(n_buckets, BIN_NB_MAX) >= 0xffffffffull) return 0;
  return static_cast<uint32_t>(cap);
}
// Can k_index_sort_bin be launched with buckets of `cap` rows on the CURRENT device?  More than 64 KB of dynamic LDS needs
// the function attribute -- per device (a group drives several devices from one process), so it is asked for by every build
// that needs it, BEFORE anything of the bin path is launched: a device (or a runtime) that refuses sends the build down the
// atomic path, which covers every input, instead of into a launch that fails.
bool index_sort_bin_prepare(uint32_t cap) {
  const size_t lds = static_cast<size_t>(cap) * BIN_LDS_PER_ROW;
  if (lds > (96u << 10)) return false;
  if (lds <= (64u << 10)) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_index_sort_bin), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return true;
}
void launch_index_bin(hipStream_t st, const msgpu_row *rows, uint64_t n, uint32_t V, uint32_t A, uint32_t *flags, uint32_t *err,
                      uint32_t *anchor_first, uint32_t *cursor, uint4 *bin_rec, uint32_t rd_lo, uint32_t nb, uint32_t cap,
                      const BinTail &tail) {
  // (n == 0: one workgroup without rows still runs the tail -- bin_start and the closing read_off entry are written)
  hipLaunchKernelGGL(k_index_bin, dim3(n ? static_cast<uint32_t>((n + BIN_TILE - 1) / BIN_TILE) : 1), dim3(BIN_NT), 0, st, rows, n, V, A,
                     flags, err, anchor_first, cursor, bin_rec, rd_lo, nb, cap, tail);
}
void launch_index_sort_bin(hipStream_t st, uint32_t *cursor, const uint32_t *bin_start, uint32_t V, uint32_t rd_lo, uint32_t nb,
                           uint32_t cap, const uint4 *bin_rec, IRow *by_read, IRow *by_anchor, uint4 *vis, uint32_t *read_off,
                           uint32_t *read_cnt, int32_t *read_len, uint32_t *read_first, uint32_t *visits, const msgpu_row *rows,
                           uint32_t *flags, uint32_t *err, uint32_t *bucket_visits) {
  if (!nb) return;
  const size_t lds = static_cast<size_t>(cap) * BIN_LDS_PER_ROW; // (granted by index_sort_bin_prepare, which the host asks first)
  hipLaunchKernelGGL(k_index_sort_bin, dim3(nb), dim3(BIN_SORT_NT), lds, st, cursor, bin_start, V,
                     rd_lo, cap, bin_rec, by_read, by_anchor, vis, read_off, read_cnt, read_len, read_first, visits, rows, flags, err,
                     bucket_visits);
}

} // namespace msgpu
