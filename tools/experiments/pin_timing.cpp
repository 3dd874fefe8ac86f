// How long does it take to get 200 MB of page-locked host memory, and how fast do copies from it run?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pin_timing tools/experiments/pin_timing.cpp && /tmp/pin_timing
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                                                   \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

int main() {
  const size_t n = size_t(200) << 20;
  void        *d = nullptr;
  CK(hipMalloc(&d, n));
  CK(hipMemset(d, 0, n));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now();
    void  *h = nullptr;
    CK(hipHostMalloc(&h, n, hipHostMallocDefault));
    double t1 = now();
    memset(h, 1, n);
    double t2 = now();
    CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    double t3 = now();
    CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    double t4 = now();
    CK(hipHostFree(h));
    double t5 = now();
    printf("hipHostMalloc %.1f ms, fill %.1f, H2D %.1f, D2H %.1f, hipHostFree %.1f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);

    for (int huge = 0; huge < 2; ++huge) {
      t0 = now();
      const size_t H = size_t(2) << 20;
      char *raw = static_cast<char *>(mmap(nullptr, n + H, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
      char *p   = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + H - 1) & ~(uintptr_t(H) - 1));
      if (huge) madvise(p, n, MADV_HUGEPAGE);
      memset(p, 1, n);
      t1 = now();
      CK(hipHostRegister(p, n, hipHostRegisterDefault));
      t2 = now();
      CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice));
      t3 = now();
      CK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost));
      t4 = now();
      CK(hipHostUnregister(p));
      t5 = now();
      munmap(raw, n + H);
      double t6 = now();
      printf("  mmap%s + touch %.1f ms, hipHostRegister %.1f, H2D %.1f, D2H %.1f, unregister %.1f, munmap %.1f\n",
             huge ? " (MADV_HUGEPAGE)" : "", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5);
    }
    {
      t0 = now();
      char *p = static_cast<char *>(malloc(n));
      memset(p, 1, n);
      t1 = now();
      CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice));
      t2 = now();
      CK(hipMemcpy(d, p, n, hipMemcpyHostToDevice));
      t3 = now();
      CK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost));
      t4 = now();
      free(p);
      t5 = now();
      printf("  malloc + touch %.1f ms, pageable H2D %.1f, again %.1f, pageable D2H %.1f, free %.1f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
    }
  }
  return 0;
}
