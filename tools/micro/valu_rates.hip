// micro-benchmark: issue cost of the vector instructions k_chain's pair sweep is made of, per wave instruction and SIMD, with
// eight wavefronts per SIMD (the kernel's occupancy): fp64 compare / add against 64-bit and 32-bit integer compares.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/micro/valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, UNROLL = 16;
#define REP16(s) s s s s s s s s s s s s s s s s
template <int OP> __global__ __launch_bounds__(256) void k(const double *in, uint64_t *out) {
  double   a = in[threadIdx.x], b = in[threadIdx.x + 256], c = a;
  uint64_t ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  uint32_t x = static_cast<uint32_t>(ua), y = static_cast<uint32_t>(ub);
  for (int i = 0; i < ITER; ++i) {
    if (OP == 0) asm volatile(REP16("v_cmp_lt_f64 s[20:21], %0, %1\n") ::"v"(a), "v"(b) : "s20", "s21");
    if (OP == 1) asm volatile(REP16("v_cmp_lt_u64 s[20:21], %0, %1\n") ::"v"(ua), "v"(ub) : "s20", "s21");
    if (OP == 2) asm volatile(REP16("v_cmp_lt_u32 s[20:21], %0, %1\n") ::"v"(x), "v"(y) : "s20", "s21");
    if (OP == 3) asm volatile(REP16("v_add_f64 %0, %1, %2\n") : "=v"(c) : "v"(a), "v"(b));
    if (OP == 4) asm volatile(REP16("v_add_u32 %0, %1, %2\n") : "=v"(x) : "v"(x), "v"(y));
    if (OP == 5) asm volatile(REP16("v_cndmask_b32 %0, %1, %2, s[20:21]\n") : "=v"(x) : "v"(x), "v"(y) : "s20", "s21");
    if (OP == 6) asm volatile(REP16("s_and_b64 s[20:21], s[22:23], s[24:25]\n") ::: "s20", "s21", "scc");
    if (OP == 7) asm volatile(REP16("v_cmp_lt_i64 s[20:21], %0, %1\n") ::"v"(ua), "v"(ub) : "s20", "s21");
    if (OP == 8) asm volatile(REP16("v_cmp_class_f64 s[20:21], %0, %1\n") ::"v"(a), "v"(x) : "s20", "s21");
    if (OP == 9) asm volatile(REP16("v_max_f64 %0, %1, %2\n") : "=v"(c) : "v"(a), "v"(b));
  }
  if (c == 12345.678 || x == 0xdeadbeef) out[0] = x + static_cast<uint64_t>(c);
}
int main() {
  double *d_in; uint64_t *d_out;
  CK(hipMalloc(&d_in, 4096)); CK(hipMalloc(&d_out, 64)); CK(hipMemset(d_in, 0x3f, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[] = {"v_cmp_lt_f64", "v_cmp_lt_u64", "v_cmp_lt_u32", "v_add_f64", "v_add_u32", "v_cndmask_b32", "s_and_b64", "v_cmp_lt_i64", "v_cmp_class_f64", "v_max_f64"};
  const int blocks = 256 * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  auto run = [&](int op) {
    switch (op) {
      case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 7: hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 8: hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
      case 9: hipLaunchKernelGGL(k<9>, dim3(blocks), dim3(256), 0, 0, d_in, d_out); break;
    }
  };
  for (int op = 0; op < 10; ++op) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0, 0); run(op); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double insts_per_simd = double(blocks) * 4 / 1024 * ITER * UNROLL; // wave instructions issued per SIMD
    printf("%-18s %8.3f ms  %6.2f ns per wave-instruction and SIMD (= %.2f cycles at 2.4 GHz)\n", names[op], best, best * 1e6 / insts_per_simd, best * 1e6 / insts_per_simd * 2.4);
  }
  return 0;
}
