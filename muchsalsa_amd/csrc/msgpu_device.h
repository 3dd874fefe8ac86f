// msgpu_device.h -- device-side helpers shared by the kernel files (msgpu_kernels.hip, msgpu_index.hip): 32-byte row
// loads / stores, readlane of wide types, wavefront and workgroup scans.  gfx950, wave64.
#ifndef MSGPU_DEVICE_H
#define MSGPU_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msgpu.h"
#include "msgpu_internal.h"

namespace msgpu {

__device__ __forceinline__ IRow load_irow(const IRow *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  uint4        a = q[0], b = q[1];
  IRow         r;
  r.n_lo  = static_cast<int>(a.x);
  r.n_hi  = static_cast<int>(a.y);
  r.i_lo  = static_cast<int>(a.z);
  r.i_hi  = static_cast<int>(a.w);
  r.score = b.x;
  r.line  = b.y;
  r.other = b.z;
  r.pf    = b.w;
  return r;
}
__device__ __forceinline__ void store_irow(IRow *p, const IRow &r) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
  q[0]     = make_uint4(static_cast<uint32_t>(r.n_lo), static_cast<uint32_t>(r.n_hi), static_cast<uint32_t>(r.i_lo),
                        static_cast<uint32_t>(r.i_hi));
  q[1]     = make_uint4(r.score, r.line, r.other, r.pf);
}

__device__ __forceinline__ IRow make_irow(const msgpu_row &row, uint32_t other, uint32_t rank) {
  IRow out;
  out.n_lo  = row.n_lo;
  out.n_hi  = row.n_hi;
  out.i_lo  = row.i_lo;
  out.i_hi  = row.i_hi;
  out.score = row.score;
  out.line  = row.line;
  out.other = other;
  out.pf    = ((row.flags & MSGPU_ROW_DIR) ? PF_DIR : 0u) | ((row.flags & MSGPU_ROW_PRIMARY) ? PF_PRIM : 0u) |
           (rank & PF_POS_MASK);
  return out;
}

// readlane of wider types (lane index must be wave-uniform)
__device__ __forceinline__ int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int lane) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int lane) {
  uint32_t lo = rl_u32(static_cast<uint32_t>(v), lane), hi = rl_u32(static_cast<uint32_t>(v >> 32), lane);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ double rl_f64(double v, int lane) {
  return __longlong_as_double(static_cast<long long>(rl_u64(static_cast<uint64_t>(__double_as_longlong(v)), lane)));
}

// std::max / std::min on doubles with the library's tie and NaN behaviour (NOT fmax/fmin)
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }

// block-wide exclusive scan of one value per thread, NT threads; returns exclusive prefix, total through *total
// inclusive scan over the wavefront with data-parallel-primitive moves (vector-ALU latency; __shfl_up is a ds_bpermute, one
// LDS round trip per step): Hillis-Steele inside the rows of 16 lanes (lanes without a source add 0), then lane 15 of
// rows 0 / 2 into rows 1 / 3 (row_bcast:15, row mask 0xa) and lane 31 into rows 2 and 3 (row_bcast:31, row mask 0xc)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false)); // row_shr:1
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false)); // row_shr:2
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false)); // row_shr:4
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false)); // row_shr:8
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false)); // row_bcast:15
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false)); // row_bcast:31
  return v;
}

// the same scan of 32-bit items with a 64-bit result
__device__ __forceinline__ uint64_t wave_incl_scan64(uint32_t v) {
  uint64_t inc = v;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  return inc;
}
// sum over the wavefront, the same value in every lane
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(wave_incl_scan(v)), 63));
}

template <int NT> __device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_wave /*[NT / 64]*/, uint32_t *total) {
  const int      lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc  = wave_incl_scan(v);
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t x = s_wave[w];
    base += w < wave ? x : 0u;
    tot += x;
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *s_wave /*[4]*/, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t  inc  = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
  uint32_t base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
  *total        = w0 + w1 + w2 + w3;
  __syncthreads();
  return base + inc - v;
}

// The Vertex of a read is made at its first line (Graph.cpp:148: nanoporeLength and metaDatum(0) of that line).  key =
// line << 32 | source row index of a lane's row (all ones for a lane without one); every lane of the wavefront calls.
__device__ __forceinline__ void note_first_row(int lane, unsigned long long key, uint32_t r, const msgpu_row *rows,
                                               int32_t *read_len, uint32_t *read_first) {
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long o =
        (static_cast<unsigned long long>(static_cast<uint32_t>(__shfl_xor(static_cast<int>(key >> 32), d))) << 32) |
        static_cast<uint32_t>(__shfl_xor(static_cast<int>(key), d));
    key = o < key ? o : key;
  }
  if (lane == 0) {
    read_first[r] = static_cast<uint32_t>(key >> 32);
    read_len[r]   = rows[static_cast<uint32_t>(key)].read_len;
  }
}

__device__ __forceinline__ bool key_less(int alo, int ahi, uint32_t aan, int blo, int bhi, uint32_t ban) {
  return alo < blo || (alo == blo && (ahi < bhi || (ahi == bhi && aan < ban)));
}

} // namespace msgpu

#endif
