// cost of hipMalloc / hipFree / hipMallocAsync by size (decides how the device-side AMG set-up gets its memory)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { using namespace std::chrono; return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreate(&s);
  void *w; hipMalloc(&w, 1 << 20); hipFree(w);
  for (size_t mb : {1, 4, 16, 64, 256}) {
    std::vector<void *> p(8);
    double t0 = now();
    for (auto &q : p) hipMalloc(&q, mb << 20);
    double t1 = now();
    for (auto &q : p) hipMemsetAsync(q, 0, mb << 20, s);
    hipStreamSynchronize(s);
    double t2 = now();
    for (auto &q : p) hipFree(q);
    double t3 = now();
    printf("hipMalloc %4zu MB: malloc %.1f us, first-touch memset %.1f us, free %.1f us (each)\n", mb, (t1 - t0) / 8, (t2 - t1) / 8, (t3 - t2) / 8);
  }
  hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
  uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  for (int rep = 0; rep < 2; rep++)
    for (size_t mb : {1, 4, 16, 64, 256}) {
      std::vector<void *> p(8);
      double t0 = now();
      for (auto &q : p) hipMallocAsync(&q, mb << 20, s);
      hipStreamSynchronize(s);
      double t1 = now();
      for (auto &q : p) hipFreeAsync(q, s);
      hipStreamSynchronize(s);
      double t2 = now();
      printf("hipMallocAsync rep %d %4zu MB: malloc %.1f us, free %.1f us (each)\n", rep, mb, (t1 - t0) / 8, (t2 - t1) / 8);
    }
  return 0;
}
