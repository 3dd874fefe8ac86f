// micro-benchmark behind the index build's design: what bounds "bucket 5 M rows by read"?
//   (a) one returning global atomic per row on a per-read counter (what k_index_pass1 did through round 3)
//   (b) the same without a return value
//   (c) the 32-byte row scattered to a random slot of a per-read bucket (no atomic)
//   (d) (a) + (c) together
//   (e) coarse binning: an 8192-row tile is binned by coarse bucket in LDS, one global atomic per (tile, non-empty bucket)
//       on ADJACENT counters (a wave instruction covers 64 neighbouring counters), rows scattered in runs
// Shapes of BASELINE.json configs[2]: R = 5.06 M rows, V = 100 k reads, read ids unrelated to the row's position.
// build: hipcc --offload-arch=gfx950 -O3 -o index_limits tools/micro/index_limits.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <utility>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Row40 { uint32_t w[10]; };
struct Row32 { uint4 a, b; };

__global__ __launch_bounds__(256) void k_atomic_ret(const Row40 *rows, uint32_t n, uint32_t *cnt, uint32_t *sink) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t rd = rows[i].w[1];
  const uint32_t pos = atomicAdd(&cnt[rd], 1u);
  if (pos == 0xffffffffu) sink[0] = i;
}
__global__ __launch_bounds__(256) void k_atomic_noret(const Row40 *rows, uint32_t n, uint32_t *cnt) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  atomicAdd(&cnt[rows[i].w[1]], 1u);
}
__global__ __launch_bounds__(256) void k_scatter(const Row40 *rows, uint32_t n, Row32 *bkt, uint32_t cap) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Row40 r = rows[i];
  const uint32_t pos = (i * 2654435761u) >> 26; // some slot of the read's bucket (0..63)
  Row32 o;
  o.a = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  o.b = make_uint4(r.w[6], r.w[7], r.w[0], i);
  bkt[static_cast<uint64_t>(r.w[1]) * cap + pos] = o;
}
__global__ __launch_bounds__(256) void k_scatter_mod(const Row40 *rows, uint32_t n, Row32 *bkt, uint32_t cap) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Row40 r = rows[i];
  const uint32_t pos = ((i * 2654435761u) >> 26) % cap;
  Row32 o;
  o.a = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  o.b = make_uint4(r.w[6], r.w[7], r.w[0], i);
  bkt[static_cast<uint64_t>(r.w[1]) * cap + pos] = o;
}
__global__ __launch_bounds__(256) void k_both(const Row40 *rows, uint32_t n, uint32_t *cnt, Row32 *bkt, uint32_t cap) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Row40 r = rows[i];
  const uint32_t pos = atomicAdd(&cnt[r.w[1]], 1u) & (cap - 1);
  Row32 o;
  o.a = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  o.b = make_uint4(r.w[6], r.w[7], r.w[0], i);
  bkt[static_cast<uint64_t>(r.w[1]) * cap + pos] = o;
}

// (f) / (g): reading the 40-byte rows alone -- one row per lane (three loads at a 40-byte stride) against 16 bytes per lane
// over the tile's bytes (what a copy does)
__global__ __launch_bounds__(256) void k_read_struct(const Row40 *rows, uint32_t n, uint32_t *sink) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Row40 r = rows[i];
  uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) x ^= r.w[k];
  if (x == 0x12345678u) sink[0] = x;
}
__global__ __launch_bounds__(256) void k_read_flat(const uint4 *p, uint32_t n16, uint32_t *sink) {
  uint32_t x = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
    const uint4 v = p[i];
    x ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (x == 0x12345678u) sink[0] = x;
}

// (h) the scatter of (c) with LANE PAIRS: a row's two 16-byte halves are stored by two neighbouring lanes in ONE wave
// instruction (32 rows per instruction), so the memory system sees one 32-byte request per row instead of two of 16.
// The halves travel through a per-wave LDS tile.  (i): 48-byte records (row + 8 bytes of side data + padding) by lane triples.
__global__ __launch_bounds__(256) void k_scatter_pairs(const Row40 *rows, uint32_t n, Row32 *bkt, uint32_t cap) {
  __shared__ uint4 s_half[4][64][2];
  __shared__ uint32_t s_slot[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  Row40 r{};
  if (i < n) r = rows[i];
  const uint32_t pos = (i * 2654435761u) >> 26;
  s_half[w][lane][0] = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  s_half[w][lane][1] = make_uint4(r.w[6], r.w[7], r.w[0], i);
  s_slot[w][lane]    = i < n ? r.w[1] * cap + pos : 0xffffffffu;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int src = 32 * j + (lane >> 1);
    const uint32_t slot = s_slot[w][src];
    if (slot != 0xffffffffu) reinterpret_cast<uint4 *>(bkt + slot)[lane & 1] = s_half[w][src][lane & 1];
  }
}
struct Rec48 { uint4 a, b, c; };
__global__ __launch_bounds__(256) void k_scatter_triples(const Row40 *rows, uint32_t n, Rec48 *bkt, uint32_t cap) {
  __shared__ uint4 s_part[4][64][3];
  __shared__ uint32_t s_slot[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  Row40 r{};
  if (i < n) r = rows[i];
  const uint32_t pos = (i * 2654435761u) >> 26;
  s_part[w][lane][0] = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  s_part[w][lane][1] = make_uint4(r.w[6], r.w[7], r.w[0], i);
  s_part[w][lane][2] = make_uint4(r.w[1], r.w[8], 0, 0);
  s_slot[w][lane]    = i < n ? r.w[1] * cap + pos : 0xffffffffu;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int q = 64 * j + lane, src = q / 3, part = q - 3 * src; // 192 parts of 64 records over three instructions
    const uint32_t slot = s_slot[w][src];
    if (slot != 0xffffffffu) reinterpret_cast<uint4 *>(bkt + slot)[part] = s_part[w][src][part];
  }
}

// (j) / (k) / (l): what plan B of the bin path moves per row -- a 16-byte key record scattered into the buckets, a 4-byte
// patch scattered into a 162 MB row table, a 32-byte row gathered from that table
__global__ __launch_bounds__(256) void k_scatter16(const Row40 *rows, uint32_t n, uint4 *bkt, uint32_t cap) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint2 ar = *reinterpret_cast<const uint2 *>(&rows[i]);
  const uint32_t pos = ((i * 2654435761u) >> 26) % cap;
  bkt[static_cast<uint64_t>(ar.y) * cap + pos] = make_uint4(ar.x, i, ar.y, pos);
}
__global__ __launch_bounds__(256) void k_patch4(const uint32_t *perm, uint32_t n, Row32 *table) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  reinterpret_cast<uint32_t *>(table + perm[i])[7] = i;
}
__global__ __launch_bounds__(256) void k_gather32(const uint32_t *perm, uint32_t n, const Row32 *table, Row32 *out) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = table[perm[i]];
}

// (m): the scatter of (c) with records that are WHOLE 64-byte sectors (row + side data + padding, 64-byte aligned): four
// 16-byte stores per lane; (n): by lane quads (one wave instruction writes 16 whole sectors)
__global__ __launch_bounds__(256) void k_scatter64(const Row40 *rows, uint32_t n, uint4 *bkt, uint32_t cap) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Row40 r = rows[i];
  const uint32_t pos = ((i * 2654435761u) >> 26) % cap;
  uint4 *o = bkt + (static_cast<uint64_t>(r.w[1]) * cap + pos) * 4;
  o[0] = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  o[1] = make_uint4(r.w[6], r.w[7], r.w[0], i);
  o[2] = make_uint4(r.w[1], r.w[8], r.w[9], 0);
  o[3] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_scatter64_quads(const Row40 *rows, uint32_t n, uint4 *bkt, uint32_t cap) {
  __shared__ uint4 s_part[4][64][4];
  __shared__ uint32_t s_slot[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  Row40 r{};
  if (i < n) r = rows[i];
  const uint32_t pos = ((i * 2654435761u) >> 26) % cap;
  s_part[w][lane][0] = make_uint4(r.w[2], r.w[3], r.w[4], r.w[5]);
  s_part[w][lane][1] = make_uint4(r.w[6], r.w[7], r.w[0], i);
  s_part[w][lane][2] = make_uint4(r.w[1], r.w[8], r.w[9], 0);
  s_part[w][lane][3] = make_uint4(0, 0, 0, 0);
  s_slot[w][lane]    = i < n ? r.w[1] * cap + pos : 0xffffffffu;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int src = 16 * j + (lane >> 2);
    const uint32_t slot = s_slot[w][src];
    if (slot != 0xffffffffu) bkt[static_cast<uint64_t>(slot) * 4 + (lane & 3)] = s_part[w][src][lane & 3];
  }
}

// (e) coarse binning.  NB buckets (a power of two), bucket = read id * NB / V (contiguous read-id ranges).
template <int NB, int PER_THREAD>
__global__ __launch_bounds__(1024) void k_bin(const Row40 *rows, uint32_t n, uint32_t V, uint32_t *cursor /*[NB]*/, Row32 *out,
                                              uint32_t cap /*rows per coarse bucket*/) {
  __shared__ uint32_t s_cnt[NB];
  const uint32_t tile = 1024 * PER_THREAD, i0 = blockIdx.x * tile;
  for (int b = threadIdx.x; b < NB; b += 1024) s_cnt[b] = 0;
  __syncthreads();
  Row40    r[PER_THREAD];
  uint32_t bk[PER_THREAD], lr[PER_THREAD];
#pragma unroll
  for (int k = 0; k < PER_THREAD; ++k) {
    const uint32_t i = i0 + k * 1024 + threadIdx.x;
    bk[k] = 0xffffffffu;
    if (i < n) {
      r[k]  = rows[i];
      bk[k] = static_cast<uint32_t>((static_cast<uint64_t>(r[k].w[1]) * NB) / V);
    }
  }
#pragma unroll
  for (int k = 0; k < PER_THREAD; ++k)
    if (bk[k] != 0xffffffffu) lr[k] = atomicAdd(&s_cnt[bk[k]], 1u); // LDS atomic: rank inside (tile, bucket)
  __syncthreads();
  for (int b = threadIdx.x; b < NB; b += 1024) { // adjacent counters per wave instruction: merged requests
    const uint32_t c = s_cnt[b];
    s_cnt[b] = c ? atomicAdd(&cursor[b], c) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER_THREAD; ++k)
    if (bk[k] != 0xffffffffu) {
      const uint32_t pos = s_cnt[bk[k]] + lr[k];
      if (pos < cap) {
        Row32 o;
        o.a = make_uint4(r[k].w[2], r[k].w[3], r[k].w[4], r[k].w[5]);
        o.b = make_uint4(r[k].w[6], r[k].w[7], r[k].w[0], r[k].w[1]);
        out[static_cast<uint64_t>(bk[k]) * cap + pos] = o;
      }
    }
}

int main() {
  const uint32_t R = 5060000, V = 100000, CAP = 128;
  std::vector<Row40> h(R);
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < R; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    for (int k = 0; k < 10; ++k) h[i].w[k] = static_cast<uint32_t>(s >> (k * 3));
    h[i].w[0] = i / 10;                              // anchor: rows grouped by anchor
    h[i].w[1] = static_cast<uint32_t>(s % V);        // read: unrelated to the position
  }
  Row40 *d_rows; uint32_t *d_cnt, *d_sink, *d_cursor; Row32 *d_bkt, *d_out;
  CK(hipMalloc(&d_rows, sizeof(Row40) * R)); CK(hipMalloc(&d_cnt, 4 * V)); CK(hipMalloc(&d_sink, 64));
  CK(hipMalloc(&d_bkt, size_t(V) * CAP * 32)); CK(hipMalloc(&d_cursor, 4 * 8192));
  const uint32_t coarse_cap = 2 * (R / 1024 + 1024);
  CK(hipMalloc(&d_out, size_t(8192) * (R / 4096 + 1024) * 2 * 32 > size_t(1024) * coarse_cap * 32 ? size_t(8192) * (R / 4096 + 1024) * 2 * 32 : size_t(1024) * coarse_cap * 32));
  CK(hipMemcpy(d_rows, h.data(), sizeof(Row40) * R, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 g((R + 255) / 256), b(256);
  auto time = [&](const char *name, auto launch, double bytes) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipMemsetAsync(d_cnt, 0, 4 * V, 0); (void)hipMemsetAsync(d_cursor, 0, 4 * 8192, 0);
      (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("%-58s %8.1f us  %6.2f G rows/s  %6.2f TB/s of algorithmic bytes\n", name, best * 1e3, R / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    return 0;
  };
  time("(a) returning atomic per row (random counter of 100 k)", [&] { hipLaunchKernelGGL(k_atomic_ret, g, b, 0, 0, d_rows, R, d_cnt, d_sink); }, 40.0 * R);
  time("(b) non-returning atomic per row", [&] { hipLaunchKernelGGL(k_atomic_noret, g, b, 0, 0, d_rows, R, d_cnt); }, 40.0 * R);
  time("(c) 32-byte row scattered into per-read buckets, no atomic", [&] { hipLaunchKernelGGL(k_scatter, g, b, 0, 0, d_rows, R, d_bkt, CAP); }, 72.0 * R);
  time("(d) (a) + (c)", [&] { hipLaunchKernelGGL(k_both, g, b, 0, 0, d_rows, R, d_cnt, d_bkt, CAP); }, 72.0 * R);
  for (uint32_t capx : {128u, 96u, 64u, 52u, 40u}) { // the same scatter into a smaller and smaller bucket region: does it matter that the region fits the 256 MiB Infinity Cache?
    char name[96];
    snprintf(name, sizeof(name), "(c') scatter, bucket region %4.0f MB (%u slots of %u per read used)", V * double(capx) * 32 / 1e6, capx < 64 ? capx : 64, capx);
    time(name, [&] { hipLaunchKernelGGL(k_scatter_mod, g, b, 0, 0, d_rows, R, d_bkt, capx); }, 72.0 * R);
  }
  {
    std::vector<uint32_t> perm(R);
    for (uint32_t i = 0; i < R; ++i) perm[i] = i;
    for (uint32_t i = R - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(perm[i], perm[s % (i + 1)]); }
    uint32_t *d_perm; CK(hipMalloc(&d_perm, 4 * size_t(R))); CK(hipMemcpy(d_perm, perm.data(), 4 * size_t(R), hipMemcpyHostToDevice));
    Row32 *table = d_bkt, *outp = d_bkt + R;
    time("(j) 16-byte key record scattered into per-read buckets (52 slots)", [&] { hipLaunchKernelGGL(k_scatter16, g, b, 0, 0, d_rows, R, reinterpret_cast<uint4 *>(d_bkt), 52u); }, 24.0 * R);
    time("(k) 4-byte patch scattered into a 162 MB row table", [&] { hipLaunchKernelGGL(k_patch4, g, b, 0, 0, d_perm, R, table); }, 8.0 * R);
    time("(l) 32-byte row gathered from a 162 MB table, written in order", [&] { hipLaunchKernelGGL(k_gather32, g, b, 0, 0, d_perm, R, table, outp); }, 68.0 * R);
  }
  time("(m) 64-byte aligned record (a whole sector) scattered, 52 slots", [&] { hipLaunchKernelGGL(k_scatter64, g, b, 0, 0, d_rows, R, reinterpret_cast<uint4 *>(d_bkt), 52u); }, 104.0 * R);
  time("(n) the same by lane quads (16 sectors per wave instruction)", [&] { hipLaunchKernelGGL(k_scatter64_quads, g, b, 0, 0, d_rows, R, reinterpret_cast<uint4 *>(d_bkt), 52u); }, 104.0 * R);
  time("(h) 32-byte row scattered by LANE PAIRS (one request per row)", [&] { hipLaunchKernelGGL(k_scatter_pairs, g, b, 0, 0, d_rows, R, d_bkt, CAP); }, 72.0 * R);
  time("(i) 48-byte record scattered by lane triples", [&] { hipLaunchKernelGGL(k_scatter_triples, g, b, 0, 0, d_rows, R, reinterpret_cast<Rec48 *>(d_bkt), CAP * 2 / 3); }, 88.0 * R);
  time("(f) read the rows, one 40-byte row per lane", [&] { hipLaunchKernelGGL(k_read_struct, g, b, 0, 0, d_rows, R, d_sink); }, 40.0 * R);
  time("(g) read the rows, 16 bytes per lane, grid-stride", [&] { hipLaunchKernelGGL(k_read_flat, dim3(4096), b, 0, 0, reinterpret_cast<const uint4 *>(d_rows), R / 2 * 5, d_sink); }, 40.0 * R);
  time("(e) coarse bins in LDS, 1024 buckets, tile 8192", [&] { hipLaunchKernelGGL((k_bin<1024, 8>), dim3((R + 8191) / 8192), dim3(1024), 0, 0, d_rows, R, V, d_cursor, d_out, coarse_cap); }, 72.0 * R);
  time("(e) coarse bins in LDS, 4096 buckets, tile 8192", [&] { hipLaunchKernelGGL((k_bin<4096, 8>), dim3((R + 8191) / 8192), dim3(1024), 0, 0, d_rows, R, V, d_cursor, d_out, 2 * (R / 4096 + 1024)); }, 72.0 * R);
  time("(e) coarse bins in LDS, 4096 buckets, tile 4096", [&] { hipLaunchKernelGGL((k_bin<4096, 4>), dim3((R + 4095) / 4096), dim3(1024), 0, 0, d_rows, R, V, d_cursor, d_out, 2 * (R / 4096 + 1024)); }, 72.0 * R);
  time("(e) coarse bins in LDS, 8192 buckets, tile 8192", [&] { hipLaunchKernelGGL((k_bin<8192, 8>), dim3((R + 8191) / 8192), dim3(1024), 0, 0, d_rows, R, V, d_cursor, d_out, 2 * (R / 8192 + 1024)); }, 72.0 * R);
  return 0;
}
