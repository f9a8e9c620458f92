// Developer micro-benchmark: cost of a grid-wide barrier (cooperative groups) on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o gridsync gridsync.hip && ./gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void k(int iters, double *buf) {
  cg::grid_group g = cg::this_grid();
  double a = 0;
  for (int i = 0; i < iters; i++) {
    a += buf[(blockIdx.x * blockDim.x + threadIdx.x + i) & 1023];
    g.sync();
  }
  if (a == 123.456) buf[0] = a;
}
__global__ void empty(double *buf) { if (buf[0] == 123.456) buf[1] = 0; }
int main() {
  double *buf; hipMalloc(&buf, 8192); hipMemset(buf, 0, 8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nb : {64, 256, 512, 1024}) {
    for (int iters : {1, 101}) {
      void *args[] = {&iters, &buf};
      hipError_t r = hipLaunchCooperativeKernel((void *)k, dim3(nb), dim3(256), args, 0, 0);
      if (r != hipSuccess) { printf("nb %d: launch failed %s\n", nb, hipGetErrorString(r)); continue; }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; rep++) hipLaunchCooperativeKernel((void *)k, dim3(nb), dim3(256), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d iters %3d: %.2f us per launch\n", nb, iters, 1e3 * ms / 20);
    }
  }
  hipEventRecord(e0);
  for (int rep = 0; rep < 200; rep++) hipLaunchKernelGGL(empty, dim3(256), dim3(256), 0, 0, buf);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("plain empty kernel back-to-back: %.2f us per launch\n", 1e3 * ms / 200);
  return 0;
}
