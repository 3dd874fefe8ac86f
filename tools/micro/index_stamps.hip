// Diagnostic build of the index build's bin path (muchsalsa_amd/csrc/msgpu_index.hip compiled with -DMSGPU_STAMPS): where a
// workgroup of k_index_bin / k_index_sort_bin spends its life.  Rows shaped like BASELINE.json configs[2]: 5.06 M rows grouped
// by anchor (about ten per anchor, ascending lines), read ids unrelated to the position, 100 k reads.  Prints the median
// cycles between the stamps over all workgroups.  A stamped build forbids overlaps the product has: read the SHARES.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMSGPU_STAMPS -Iinclude -Imuchsalsa_amd/csrc -o tools/micro/index_stamps tools/micro/index_stamps.hip
#include "../../muchsalsa_amd/csrc/msgpu_index.hip"

#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace msgpu;

static void report(const char *name, const std::vector<unsigned long long> &st, size_t wgs, const char *const *labels, int n) {
  printf("%s: %zu workgroups; median cycles per phase (100 MHz s_memtime ticks x 24 at 2.4 GHz are NOT applied: raw ticks)\n", name, wgs);
  for (int k = 0; k + 1 < n; ++k) {
    std::vector<unsigned long long> d;
    for (size_t w = 0; w < wgs; ++w)
      if (st[w * 8 + k + 1] > st[w * 8 + k]) d.push_back(st[w * 8 + k + 1] - st[w * 8 + k]);
    if (d.empty()) continue;
    std::sort(d.begin(), d.end());
    printf("  %-52s median %8llu   p90 %8llu\n", labels[k], d[d.size() / 2], d[d.size() * 9 / 10]);
  }
  std::vector<unsigned long long> tot;
  for (size_t w = 0; w < wgs; ++w) tot.push_back(st[w * 8 + n - 1] - st[w * 8]);
  std::sort(tot.begin(), tot.end());
  printf("  %-52s median %8llu   p90 %8llu\n", "whole workgroup", tot[tot.size() / 2], tot[tot.size() * 9 / 10]);
}

int main() {
  const uint32_t V = 100000, A = 506000;
  const uint64_t R = 5060000;
  std::vector<msgpu_row> h(R);
  uint64_t s = 88172645463325252ull;
  for (uint64_t i = 0; i < R; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    msgpu_row &r = h[i];
    r.anchor_id = static_cast<uint32_t>(i / 10);
    r.read_id   = static_cast<uint32_t>(s % V);
    r.read_len  = 10000;
    r.i_lo = 0; r.i_hi = 800;
    r.n_lo = static_cast<int32_t>((s >> 20) % 9000); r.n_hi = r.n_lo + 800;
    r.score = 700; r.line = static_cast<uint32_t>(i); r.flags = 3;
  }
  // (a read id twice inside one anchor would be a duplicate pair: nudge it)
  for (uint64_t i = 1; i < R; ++i)
    for (uint64_t j = i - i % 10; j < i; ++j)
      if (h[j].read_id == h[i].read_id) h[i].read_id = (h[i].read_id + 7919) % V;
  const uint32_t nb = (V + 15) / 16, cap = bin_capacity(R, V);
  printf("rows %llu, reads %u, buckets %u, capacity %u rows (%u KB of LDS per sort workgroup)\n", (unsigned long long)R, V, nb, cap, cap * 38 / 1024);
  msgpu_row *d_rows; uint32_t *d_flags, *d_err, *d_first, *d_cursor, *d_start, *d_base, *d_off, *d_cnt, *d_rfirst, *d_vis;
  int32_t *d_len; uint4 *d_rec, *d_vis16; IRow *d_by_read, *d_by_anchor;
  CK(hipMalloc(&d_rows, R * sizeof(msgpu_row))); CK(hipMemcpy(d_rows, h.data(), R * sizeof(msgpu_row), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_flags, 64)); CK(hipMalloc(&d_err, 64)); CK(hipMalloc(&d_first, (A + 2) * 4ull)); CK(hipMalloc(&d_cursor, (nb + 2) * 4ull));
  CK(hipMalloc(&d_start, (nb + 2) * 4ull)); CK(hipMalloc(&d_base, 64)); CK(hipMalloc(&d_off, (V + 2) * 4ull)); CK(hipMalloc(&d_cnt, (V + 2) * 4ull));
  CK(hipMalloc(&d_rfirst, (V + 2) * 4ull)); CK(hipMalloc(&d_vis, (V + 2) * 4ull)); CK(hipMalloc(&d_len, (V + 2) * 4ull));
  CK(hipMalloc(&d_rec, size_t(nb) * cap * 64)); CK(hipMalloc(&d_vis16, R * 16)); CK(hipMalloc(&d_by_read, R * 32)); CK(hipMalloc(&d_by_anchor, R * 32));
  const size_t wg1 = (R + 2047) / 2048, wg3 = nb;
  unsigned long long *d_st; CK(hipMalloc(&d_st, std::max(wg1, wg3) * 64));
  std::vector<unsigned long long> st(std::max(wg1, wg3) * 8);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(d_flags, 0, 64)); CK(hipMemset(d_err, 0, 64)); CK(hipMemset(d_cursor, 0, (nb + 2) * 4ull)); CK(hipMemset(d_base, 0, 64));
    CK(hipMemset(d_first, 0xff, (A + 2) * 4ull)); CK(hipMemset(d_st, 0, std::max(wg1, wg3) * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    CK(hipEventRecord(e0, 0));
    launch_index_bin(0, d_rows, R, V, A, d_flags, d_err, d_first, d_cursor, d_rec, 0, nb, cap, d_start, d_base, d_off + V, true);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st.data(), d_st, wg1 * 64, hipMemcpyDeviceToHost));
    static const char *const l1[] = {"rows loaded, (anchor, read) in LDS [barrier]", "scaffold-start masks + LDS atomics [barrier]", "global atomics, bases in LDS [barrier]", "scaffold places, staging, sector stores (drained)"};
    if (rep) report("k_index_bin", st, wg1, l1, 5);
    CK(hipMemset(d_st, 0, std::max(wg1, wg3) * 64));
    CK(hipEventRecord(e1, 0));
    launch_index_sort_bin(0, d_cursor, d_start, V, 0, nb, cap, d_rec, d_by_read, d_by_anchor, d_vis16, d_off, d_cnt, d_len, d_rfirst, d_vis, d_rows, d_flags, d_err);
    CK(hipEventRecord(e2, 0));
    CK(hipDeviceSynchronize());
    uint32_t fl = 0; CK(hipMemcpy(&fl, d_flags, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st.data(), d_st, wg3 * 64, hipMemcpyDeviceToHost));
    static const char *const l3[] = {"bucket records read into LDS, per-read counts [barrier]", "count scan + index grouped by read [2 barriers]", "per-read ranking + stores, 2 reads per wave [barrier]", "first lines: read_len gathered (drained)"};
    if (rep) { report("k_index_sort_bin", st, wg3, l3, 5); float ms; CK(hipEventElapsedTime(&ms, e1, e2)); printf("flags %u; k_index_sort_bin (stamped) %.1f us\n", fl, ms * 1e3); }
  }
  return 0;
}
