// micro-benchmark: what does a 128-byte size read-back cost?  (a) hipMemcpyAsync D2H + hipStreamSynchronize,
// (b) a one-wave kernel that copies the block to host-mapped memory and raises a sequence flag the host spins on.
// build: hipcc --offload-arch=gfx950 -O2 -o readback tools/micro/readback.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(uint64_t *s, int n) { // a little real work in front of the read-back
  uint64_t a = threadIdx.x;
  for (int i = 0; i < n; ++i) a = a * 6364136223846793005ull + 1442695040888963407ull;
  if (threadIdx.x < 16) s[threadIdx.x] = a + blockIdx.x;
}
__global__ void k_publish(const uint64_t *s, volatile uint64_t *host, volatile uint64_t *flag, uint64_t seq) {
  if (threadIdx.x < 16) host[threadIdx.x] = s[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *flag = seq;
}
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint64_t *d, *h, *hm; CK(hipMalloc(&d, 128)); CK(hipHostMalloc(&h, 128, hipHostMallocDefault));
  CK(hipHostMalloc(&hm, 256, hipHostMallocMapped | hipHostMallocCoherent));
  uint64_t *hm_dev; CK(hipHostGetDevicePointer((void **)&hm_dev, hm, 0));
  volatile uint64_t *flag = hm + 16;
  const int N = 2000;
  for (int work : {0, 2000}) {
    for (int rep = 0; rep < 2; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (work) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, work);
        CK(hipMemcpyAsync(h, d, 128, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      }
      double a = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      *flag = 0;
      t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (work) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, work);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, d, hm_dev, hm_dev + 16, (uint64_t)(i + 1));
        while (*flag != (uint64_t)(i + 1)) { }
      }
      double b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (work) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, work);
        CK(hipStreamSynchronize(st));
      }
      double c = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      printf("work=%d rep=%d: memcpyAsync+sync %.2f us | publish kernel + spin %.2f us | (launch+)sync only %.2f us\n", work, rep, a, b, c);
    }
  }
  return 0;
}
