// launch_bench.hip -- how long an (almost) empty launch takes as a function of its shape: the floor under
// every launch-per-ply kernel.  Each thread reads one dword and, if it is a value that never occurs, writes.
// Usage: tools/launch_bench   (prints us per launch, median of 2000 back-to-back launches, per shape)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ void k_empty(const int* in, int* out, int lds_words) {
  extern __shared__ int sm[];
  if (lds_words) sm[threadIdx.x % lds_words] = 0;
  if (in[blockIdx.x] == 0x7fffffff) out[blockIdx.x] = sm[0];
}

int main() {
  int *in, *out;
  hipMalloc(&in, 1 << 20);
  hipMalloc(&out, 1 << 20);
  hipMemset(in, 0, 1 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int shapes[][3] = {{1024, 64, 0}, {1024, 128, 0}, {1024, 192, 0}, {1024, 192, 1536}, {512, 384, 3072},
                           {256, 768, 6144}, {342, 576, 4608}, {2048, 64, 0}, {3072, 64, 0}, {8192, 192, 1536},
                           {4096, 384, 3072}, {256, 64, 0}, {1, 64, 0}};
  for (auto& s : shapes) {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(s[0]), dim3(s[1]), s[2] * 4, 0, in, out, s[2]);
    hipDeviceSynchronize();
    const int N = 2000;
    hipEventRecord(e0, 0);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(s[0]), dim3(s[1]), s[2] * 4, 0, in, out, s[2]);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("grid %5d x %4d threads, %5d B LDS: %7.2f us per launch (back to back)\n", s[0], s[1], s[2] * 4, ms * 1e3 / N);
  }
  return 0;
}
