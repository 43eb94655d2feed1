// Diagnostic microbenchmark: how fast can one wave per board write its 9025-float feature block
// under different store shapes?  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define NP 361
#define FB 9025

__global__ __launch_bounds__(64) void p0_kmajor(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x;
  float* o = out + (size_t)b * FB;
  for (int k = 0; k < 6; ++k) { int q = lane + 64 * k; if (q < NP) {
#pragma unroll
      for (int pl = 0; pl < 25; ++pl) o[pl * NP + q] = (float)(pl + k); } }
}
__global__ __launch_bounds__(64) void p1_planemajor(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x;
  float* o = out + (size_t)b * FB;
#pragma unroll
  for (int pl = 0; pl < 25; ++pl) {
#pragma unroll
    for (int k = 0; k < 6; ++k) { int q = lane + 64 * k; if (q < NP) o[pl * NP + q] = (float)(pl + k); } }
}
typedef float f4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) uf4 { float x, y, z, w; };
__global__ __launch_bounds__(64) void p2_planemajor_x4(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x;
  float* o = out + (size_t)b * FB;
#pragma unroll
  for (int pl = 0; pl < 25; ++pl) {
    float v = (float)pl;
    uf4 val = {v, v + 1, v + 2, v + 3};
    *(uf4*)(o + pl * NP + 4 * lane) = val;
    if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
    if (lane == 26) o[pl * NP + 360] = v;
  }
}
// whole block as an aligned flat stream: head (<=3 floats), 16-B body, tail
__global__ __launch_bounds__(64) void p3_flat_aligned(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x;
  size_t s = (size_t)b * FB, e = s + FB;
  size_t a = (s + 3) & ~(size_t)3, z = e & ~(size_t)3;
  if (lane < (int)(a - s)) out[s + lane] = 1.f;
  if (lane < (int)(e - z)) out[z + lane] = 2.f;
  f4 val = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = a + 4 * lane; i < z; i += 256) *(f4*)(out + i) = val;
}
// p2 with the non-temporal (streaming) hint on every 16-byte store
__device__ __forceinline__ void st_nt(float* p, f4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
__global__ __launch_bounds__(64) void p4_planemajor_x4_nt(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x;
  float* o = out + (size_t)b * FB;
#pragma unroll
  for (int pl = 0; pl < 25; ++pl) {
    float v = (float)pl;
    f4 val = {v, v + 1, v + 2, v + 3};
    st_nt(o + pl * NP + 4 * lane, val);
    if (lane < 26) st_nt(o + pl * NP + 256 + 4 * lane, val);
    if (lane == 26) o[pl * NP + 360] = v;
  }
}
// p2 launched as 3 waves per board (each wave a third of the planes), like the product kernel
__global__ __launch_bounds__(192) void p5_three_waves(float* out, int n) {
  int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* o = out + (size_t)b * FB;
  for (int pl = w; pl < 25; pl += 3) {
    float v = (float)pl;
    uf4 val = {v, v + 1, v + 2, v + 3};
    *(uf4*)(o + pl * NP + 4 * lane) = val;
    if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
    if (lane == 26) o[pl * NP + 360] = v;
  }
}
// persistent variants: every board (one wave, or three) writes its 36,100-B block for T plies in a
// row, as the generator kernel does.  lay 0: ring[t][b] (slot-major), lay 1: ring[b][t] (a board's
// plies contiguous).
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void p_persist(float* out, int n, int T, int lay) {
  int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int t = 0; t < T; ++t) {
    float* o = out + (lay ? ((size_t)b * T + t) : ((size_t)t * n + b)) * FB;
    for (int pl = w; pl < 25; pl += NWV) {
      float v = (float)(pl + t);
      uf4 val = {v, v + 1, v + 2, v + 3};
      *(uf4*)(o + pl * NP + 4 * lane) = val;
      if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
      if (lane == 26) o[pl * NP + 360] = v;
    }
  }
}
static void run_persist(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwv = 1; nwv <= 3; nwv += 2)
    for (int lay = 0; lay < 2; ++lay) {
      for (int r = 0; r < reps + 1; ++r) {
        if (r == 1) hipEventRecord(e0);
        if (nwv == 1) hipLaunchKernelGGL(p_persist<1>, dim3(n), dim3(64), 0, 0, out, n, T, lay);
        else hipLaunchKernelGGL(p_persist<3>, dim3(n), dim3(192), 0, 0, out, n, T, lay);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("n=%d T=%d persistent %d wave(s)/board, %s  %8.3f us per ply  %7.1f GB/s\n", n, T, nwv,
             lay ? "ring[b][t]" : "ring[t][b]", ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
    }
}
// persistent, one wave per board, with at most VM plane stores of the wave in flight (does a
// narrower window of outstanding writes help DRAM locality, as the plain fills suggest?)
template <int VM>
__global__ __launch_bounds__(64) void p_persist_throttled(float* out, int n, int T) {
  int b = blockIdx.x, lane = threadIdx.x;
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * FB;
    for (int pl = 0; pl < 25; ++pl) {
      float v = (float)(pl + t);
      uf4 val = {v, v + 1, v + 2, v + 3};
      *(uf4*)(o + pl * NP + 4 * lane) = val;
      if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
      if (lane == 26) o[pl * NP + 360] = v;
      if (VM == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (VM == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (VM == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      if (VM == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
  }
}
static void run_throttled(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int vms[5] = {0, 3, 6, 12, 63};
  for (int vi = 0; vi < 5; ++vi) {
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) hipEventRecord(e0);
      switch (vms[vi]) {
        case 0: hipLaunchKernelGGL(p_persist_throttled<0>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 3: hipLaunchKernelGGL(p_persist_throttled<3>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 6: hipLaunchKernelGGL(p_persist_throttled<6>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 12: hipLaunchKernelGGL(p_persist_throttled<12>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        default: hipLaunchKernelGGL(p_persist_throttled<63>, dim3(n), dim3(64), 0, 0, out, n, T); break;
      }
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d persistent 1 wave/board, at most %2d store groups in flight  %8.3f us per ply  %7.1f GB/s\n", n, T,
           vms[vi], ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}
// ceiling probes: plain aligned fills of the same byte count (no board structure at all)
__global__ __launch_bounds__(256) void fill_gridstride(f4* out, size_t n4) {
  f4 val = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = val;
}
__global__ __launch_bounds__(256) void fill_chunked(f4* out, size_t n4) {   // each workgroup one contiguous chunk
  f4 val = {1.f, 2.f, 3.f, 4.f};
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) out[i] = val;
}
static void run_fills(float* out, int n, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n4 = (size_t)n * FB / 4;
  const int grids[5] = {256, 1024, 2048, 8192, 65536};
  for (int v = 0; v < 2; ++v)
    for (int gi = 0; gi < 5; ++gi) {
      for (int r = 0; r < reps + 1; ++r) {
        if (r == 1) hipEventRecord(e0);
        if (v == 0) hipLaunchKernelGGL(fill_gridstride, dim3(grids[gi]), dim3(256), 0, 0, (f4*)out, n4);
        else hipLaunchKernelGGL(fill_chunked, dim3(grids[gi]), dim3(256), 0, 0, (f4*)out, n4);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("fill %-11s %6d workgroups x 256  %9.3f us  %7.1f GB/s\n", v ? "chunked" : "grid-stride", grids[gi], ms * 1e3,
             (double)n4 * 16 / (ms * 1e-3) / 1e9);
    }
  hipMemset(out, 0, n4 * 16); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipMemsetAsync(out, 0, n4 * 16, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("hipMemsetAsync %9.3f us  %7.1f GB/s\n", ms * 1e3, (double)n4 * 16 / (ms * 1e-3) / 1e9);
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 65536, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int T = argc > 3 ? atoi(argv[3]) : 0;   // > 0: the persistent variants only
  float* out;
  if (hipMalloc(&out, (size_t)n * (T > 0 ? T : 1) * FB * 4 + 64) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  if (T > 0) { run_persist(out, n, T, reps); run_throttled(out, n, T, reps); return 0; }
  if (T < 0) { run_fills(out, n, reps); return 0; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  void (*ks[6])(float*, int) = {p0_kmajor, p1_planemajor, p2_planemajor_x4, p3_flat_aligned, p4_planemajor_x4_nt, p5_three_waves};
  const char* nm[6] = {"p0 k-major dword", "p1 plane-major dword", "p2 plane-major x4 (4B aligned)", "p3 flat aligned x4",
                       "p4 plane-major x4 nt", "p5 plane-major x4, 3 waves/board"};
  const int bs[6] = {64, 64, 64, 64, 64, 192};
  for (int round = 0; round < 2; ++round)
    for (int v = 0; v < 6; ++v) {
      hipLaunchKernelGGL(ks[v], dim3(n), dim3(bs[v]), 0, 0, out, n);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ks[v], dim3(n), dim3(bs[v]), 0, 0, out, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      if (round) printf("n=%d %-34s %8.3f us  %7.1f GB/s\n", n, nm[v], ms * 1e3, (double)n * FB * 4 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
