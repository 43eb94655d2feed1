// Diagnostic microbenchmark: how fast can one wave per board write its 9025-float feature block
// under different store shapes?  Not part of the product.
// Usage: store_bench <boards> <reps> [plies [mode]]      (results under profiles/r01_store_bench*.txt)
//   plies = 0 (default)  one launch per ply, store shapes p0..p5
//   plies < 0            plain fills of the same byte count (grid-stride / chunked / hipMemsetAsync)
//   plies > 0, mode 0    persistent stores into ring[t][b] / ring[b][t], 1 and 3 waves per board, throttled variants
//   mode 1  expanders with a chip-wide write front (fill_front, expand_board, expand_flat)
//   mode 2  board <-> workgroup maps (identity, XCD-contiguous, CU-contiguous, shuffled)
//   mode 3  plane splits between two waves of a board      mode 4  one storing wave per group of boards
//   mode 5  stream expanders from packed samples           mode 6  alignment of the wave-stores
//   mode 7  cache-policy bits of the stores (sc0 / sc1 / nt)     mode 8  samples written in address order in time
//   mode 10 paced stores: rows padded to whole cache lines / written as aligned 1-KB stores
//   mode 9  paced stores: rows per XCD / per CU / ring[b][t] / half samples / two waves
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
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

// ---- two-stage probes: a rules kernel would leave one packed dword per point (1,472 B per board-ply) and a
// separate "expander" would write the 36,100-B sample.  How fast can such an expander write, as a function of
// how wide the chip-wide write front is?
// fill_front: G workgroups in teams of g; team i writes boards i, i+G/g, ...; member m of a team writes the m-th
// slice of the board.  g = 1: every workgroup whole boards (front = G boards wide).
__global__ __launch_bounds__(256) void fill_front(float* out, int nbp, int g) {
  const int team = blockIdx.x / g, mem = blockIdx.x % g, nteams = gridDim.x / g;
  const int Q = (((FB + g - 1) / g) + 3) & ~3;
  const int lo = mem * Q, hi = lo + Q < FB ? lo + Q : FB;
  for (int bp = team; bp < nbp; bp += nteams) {
    float* o = out + (size_t)bp * FB;
    const int body = lo + ((hi - lo) & ~3);
    for (int i = lo + 4 * (int)threadIdx.x; i < body; i += 1024) { uf4 v = {1.f, 2.f, 3.f, (float)bp}; *(uf4*)(o + i) = v; }
    if ((int)threadIdx.x < hi - body) o[body + threadIdx.x] = 5.f;
  }
}
#define CW 368   // dwords per packed board-ply: 361 points + header
__global__ void init_comp(uint32_t* comp, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)(i * 2654435761u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const int q = (int)(i % CW);
    uint32_t col = h % 3, cls = col ? 1 + (h >> 4) % 3 : 0;
    comp[i] = q < NP ? (col | cls << 2 | ((h >> 8) % 97 == 0) << 4 | ((h >> 10) % 300) << 5 | ((h >> 20) % 37) << 16 | ((h >> 26) % 37) << 24)
                     : (1 + ((i / CW) & 1)) | 3u << 2;
  }
}
__device__ __forceinline__ float plane_val(int pl, uint32_t W, uint32_t our, uint32_t has, const float* lut) {
  const uint32_t col = W & 3u, cls = (W >> 2) & 3u, opp = 3u - our;
  const bool mine = col == our, theirs = col == opp;
  switch (pl) {
    case 0: return (mine && cls == 1) ? 1.f : 0.f;
    case 1: return (mine && cls == 2) ? 1.f : 0.f;
    case 2: return (mine && cls != 1 && cls != 2) ? 1.f : 0.f;
    case 3: return (theirs && cls == 1) ? 1.f : 0.f;
    case 4: return (theirs && cls == 2) ? 1.f : 0.f;
    case 5: return (theirs && cls != 1 && cls != 2) ? 1.f : 0.f;
    case 6: return (W & 16u) ? 1.f : 0.f;
    case 7: return mine ? 1.f : 0.f;
    case 8: return theirs ? 1.f : 0.f;
    case 9: return col == 0 ? 1.f : 0.f;
    case 10: return mine ? lut[(W >> 5) & 0x7ffu] : 0.f;
    case 11: return theirs ? lut[(W >> 5) & 0x7ffu] : 0.f;
    case 14: return ((has >> (our - 1)) & 1u) ? (float)((W >> (our == 1 ? 16 : 24)) & 0xffu) : 10000.f;
    case 15: return ((has >> (opp - 1)) & 1u) ? (float)((W >> (opp == 1 ? 16 : 24)) & 0xffu) : 10000.f;
    default: return 0.f;
  }
}
// expand_board: one workgroup (4 waves) per board-ply, boards dealt round-robin (front = gridDim boards wide).
// Thread t owns points 4t..4t+3 (t < 90) and thread 90 points 357..360; wave w stores planes w, w+4, ...
// restricted to its own points: plane-major over the workgroup, so two wave-stores per plane as in the product.
__global__ __launch_bounds__(256) void expand_board(const uint32_t* __restrict__ comp, const float* __restrict__ lut,
                                                    float* __restrict__ out, int nbp) {
  __shared__ float slut[2048];
  __shared__ uint32_t rec[2][CW];
  for (int i = threadIdx.x; i < 2048; i += 256) slut[i] = lut[i];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int qa = 4 * lane, qb = lane < 26 ? 256 + 4 * lane : NP - 4;
  const bool act_b = lane <= 26;
  int bp = blockIdx.x, cur = 0;
  uint32_t pre0 = 0, pre1 = 0;
  if (bp < nbp) { pre0 = comp[(size_t)bp * CW + threadIdx.x]; if (threadIdx.x < CW - 256) pre1 = comp[(size_t)bp * CW + 256 + threadIdx.x]; }
  for (; bp < nbp; bp += gridDim.x, cur ^= 1) {
    rec[cur][threadIdx.x] = pre0; if (threadIdx.x < CW - 256) rec[cur][256 + threadIdx.x] = pre1;
    __syncthreads();
    const int nx = bp + gridDim.x;
    if (nx < nbp) { pre0 = comp[(size_t)nx * CW + threadIdx.x]; if (threadIdx.x < CW - 256) pre1 = comp[(size_t)nx * CW + 256 + threadIdx.x]; }
    const uint32_t hdr = rec[cur][NP], our = hdr & 3u, has = hdr >> 2;
    uint32_t d[8];
    for (int i = 0; i < 4; ++i) { d[i] = rec[cur][qa + i]; d[4 + i] = rec[cur][qb + i]; }
    float* o = out + (size_t)bp * FB;
#pragma unroll
    for (int pl = 0; pl < 25; ++pl) {
      if ((pl & 3) != w) continue;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = plane_val(pl, d[i], our, has, slut);
      uf4 va = {v[0], v[1], v[2], v[3]}, vb = {v[4], v[5], v[6], v[7]};
      *(uf4*)(o + pl * NP + qa) = va;
      if (act_b) *(uf4*)(o + pl * NP + qb) = vb;
    }
  }
}
// expand_flat: XCD-aware narrow front.  Workgroup w runs on XCD w%8 (round-robin dispatch); XCD x owns the
// board-plies bp = 8k + x and its gridDim/8 workgroups walk them as ONE flat float stream, 4 floats per thread.
__global__ __launch_bounds__(256) void expand_flat(const uint32_t* __restrict__ comp, const float* __restrict__ lut,
                                                   float* __restrict__ out, int nbp, int compute) {
  __shared__ float slut[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) slut[i] = lut[i];
  __syncthreads();
  const int x = blockIdx.x & 7, s = blockIdx.x >> 3, S = gridDim.x >> 3;
  const long long nk = (nbp - x + 7) >> 3, total = nk * FB;
  for (long long base = ((long long)s * 256 + threadIdx.x) * 4; base < total; base += (long long)S * 1024) {
    int k = (int)(base / FB), r = (int)(base - (long long)k * FB);
    float v[4];
    size_t bp = (size_t)8 * k + x;
    float* o = out + bp * FB + r;
    if (!compute) { uf4 vv = {1.f, 2.f, 3.f, (float)k}; if (r <= FB - 4) *(uf4*)o = vv; else for (int j = 0; j < FB - r; ++j) o[j] = 1.f; continue; }
    int pl = r / NP, q = r - pl * NP;
    const uint32_t* c = comp + bp * CW;
    const uint32_t hdr = c[NP], our = hdr & 3u, has = hdr >> 2;
    const int nfl = r <= FB - 4 ? 4 : FB - r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool zero = pl == 12 || pl == 13 || pl >= 16;
      v[j] = zero ? 0.f : plane_val(pl, c[q], our, has, slut);
      if (++q == NP) { q = 0; ++pl; }
    }
    if (nfl == 4) { uf4 vv = {v[0], v[1], v[2], v[3]}; *(uf4*)o = vv; }
    else {
      for (int j = 0; j < nfl; ++j) o[j] = v[j];
      // the floats past this board's end belong to the XCD's next board (8 boards on in memory)
      if ((long long)(k + 1) < nk) {
        float* o2 = out + (bp + 8) * FB; const uint32_t* c2 = comp + (bp + 8) * CW;
        const uint32_t h2 = c2[NP];
        for (int j = nfl; j < 4; ++j) o2[j - nfl] = plane_val(0, c2[j - nfl], h2 & 3u, h2 >> 2, slut);
      }
    }
  }
}

// board <-> workgroup permutations of the persistent one-wave-per-board store (ring[t][b]): does it matter
// which XCD writes which board?  map 0: board = workgroup; 1: XCD x (= workgroup % 8) owns the contiguous
// boards [x n/8, (x+1) n/8); 2: CU-contiguous (workgroups that share a CU own neighbouring boards, assuming
// round-robin over 8 XCDs x 32 CUs); 3: a fixed odd-multiplier shuffle (no locality at all).
__global__ __launch_bounds__(64) void p_persist_map(float* out, int n, int T, int map, int nper) {
  const int w = blockIdx.x, lane = threadIdx.x;
  int b = w;
  if (map == 1) b = (w & 7) * (n >> 3) + (w >> 3);
  else if (map == 2) { const int cu = w & 255, k = w >> 8; b = cu * (n >> 8) + k; }
  else if (map == 3) b = (int)(((unsigned)w * 40503u) % (unsigned)n);
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * FB;
    for (int pl = 0; pl < 25; ++pl) {
      float v = (float)(pl + t);
      uf4 val = {v, v + 1, v + 2, v + 3};
      *(uf4*)(o + pl * NP + 4 * lane) = val;
      if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
      if (lane == 26) o[pl * NP + 360] = v;
      if (nper && (pl % nper) == nper - 1) __builtin_amdgcn_s_sleep(8);
    }
  }
}
static void run_map(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int map = 0; map < 4; ++map) {
    for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(p_persist_map, dim3(n), dim3(64), 0, 0, out, n, T, map, 0); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d persistent 1 wave/board ring[t][b], board map %d  %8.3f us per ply  %7.1f GB/s\n", n, T, map,
           ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}

// two waves per board (as the generator at <= 4,096 boards), XCD-contiguous boards, one barrier per ply:
// how should the 25 planes be split between the waves?  split 0: wave 0 planes 0..11, wave 1 planes 12..24;
// 1: even / odd planes; 2: the generator's (wave 1: 0..11, wave 0: 12, 13, 16..24, then 14 | 15 after a
// barrier); 3: wave 0 stores everything, wave 1 idles; 4: as 0 without the barrier.
__global__ __launch_bounds__(128) void p_persist2(float* out, int n, int T, int split) {
  const int w0 = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = (w0 & 7) * (n >> 3) + (w0 >> 3);
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * FB;
#define PL(pl) do { float v = (float)((pl) + t); uf4 val = {v, v + 1, v + 2, v + 3}; *(uf4*)(o + (pl) * NP + 4 * lane) = val; \
    if (lane < 26) *(uf4*)(o + (pl) * NP + 256 + 4 * lane) = val; if (lane == 26) o[(pl) * NP + 360] = v; } while (0)
    if (split == 0 || split == 4) { if (w == 0) for (int pl = 0; pl < 12; ++pl) PL(pl); else for (int pl = 12; pl < 25; ++pl) PL(pl); }
    else if (split == 1) { for (int pl = w; pl < 25; pl += 2) PL(pl); }
    else if (split == 2) {
      if (w == 1) for (int pl = 0; pl < 12; ++pl) PL(pl);
      else { PL(12); PL(13); for (int pl = 16; pl < 25; ++pl) PL(pl); }
      __syncthreads();
      if (w == 0) PL(14); else PL(15);
    } else if (split == 3) { if (w == 0) for (int pl = 0; pl < 25; ++pl) PL(pl); }
#undef PL
    if (split != 4) __syncthreads();
  }
}
static void run_split(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int sp = 0; sp < 5; ++sp) {
    for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(p_persist2, dim3(n), dim3(128), 0, 0, out, n, T, sp); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d persistent 2 waves/board ring[t][b], XCD-contiguous, plane split %d  %8.3f us per ply  %7.1f GB/s\n", n, T, sp,
           ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}

// how many concurrent sequential streams does the memory side like?  n boards, one storing wave per GROUP of
// bpw boards (XCD-contiguous groups); the wave writes its boards one after the other, every ply.
__global__ __launch_bounds__(64) void p_persist_group(float* out, int n, int T, int bpw) {
  const int w = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  const int grp = (w & 7) * (G >> 3) + (w >> 3);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < bpw; ++i) {
      float* o = out + ((size_t)t * n + (size_t)grp * bpw + i) * FB;
      for (int pl = 0; pl < 25; ++pl) {
        float v = (float)(pl + t);
        uf4 val = {v, v + 1, v + 2, v + 3};
        *(uf4*)(o + pl * NP + 4 * lane) = val;
        if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
        if (lane == 26) o[pl * NP + 360] = v;
      }
    }
}
static void run_group(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpw = 1; bpw <= 16 && n / bpw >= 64; bpw *= 2) {
    for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(p_persist_group, dim3(n / bpw), dim3(64), 0, 0, out, n, T, bpw); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d one storing wave per %2d boards (%5d waves)  %8.3f us per ply  %7.1f GB/s\n", n, T, bpw, n / bpw,
           ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}

// expander candidates with ONE sequential stream per board (the shape that reaches 5.46 TB/s as a pure store):
// board b of ply t from the packed record comp[t][b][368].  Variant 0: one wave per board, next record
// prefetched into registers; variant 1: two waves per board, wave 0 only loads records into an LDS ring two
// plies ahead (its vmcnt sees loads only), wave 1 expands and stores.
template <int VAR>
__global__ __launch_bounds__(VAR ? 128 : 64) void expand_stream(const uint32_t* __restrict__ comp, const float* __restrict__ lut,
                                                                float* __restrict__ out, int n, int T) {
  __shared__ float slut[2048];
  __shared__ uint32_t ring[VAR ? 3 : 1][CW];
  const int nthr = VAR ? 128 : 64;
  for (int i = threadIdx.x; i < 2048; i += nthr) slut[i] = lut[i];
  const int w0 = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (w0 & 7) * (n >> 3) + (w0 >> 3);
  const int qa = 4 * lane, qb = lane < 26 ? 256 + 4 * lane : NP - 4;
  const bool act_b = lane <= 26;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(4))) uu4 { uint32_t x, y, z, w; };
#define EXPAND_PLANES(o, d, hdr)                                                                          \
  do {                                                                                                    \
    const uint32_t our = (hdr) & 3u, has = (hdr) >> 2;                                                    \
    _Pragma("unroll") for (int pl = 0; pl < 25; ++pl) {                                                   \
      float v[8];                                                                                         \
      const bool zero = pl == 12 || pl == 13 || pl >= 16;                                                 \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) v[i] = zero ? 0.f : plane_val(pl, d[i], our, has, slut); \
      uf4 va = {v[0], v[1], v[2], v[3]}, vb = {v[4], v[5], v[6], v[7]};                                   \
      *(uf4*)((o) + pl * NP + qa) = va;                                                                   \
      if (act_b) *(uf4*)((o) + pl * NP + qb) = vb;                                                        \
    }                                                                                                     \
  } while (0)
  if (VAR == 0) {
    __syncthreads();
    uint32_t d[8], nd[8], hdr, nhdr;
    const uint32_t* c = comp + (size_t)b * CW;
    { uu4 a = *(const uu4*)(c + qa), bb = *(const uu4*)(c + qb); nd[0] = a.x; nd[1] = a.y; nd[2] = a.z; nd[3] = a.w; nd[4] = bb.x; nd[5] = bb.y; nd[6] = bb.z; nd[7] = bb.w; nhdr = c[NP]; }
    for (int t = 0; t < T; ++t) {
      for (int i = 0; i < 8; ++i) d[i] = nd[i];
      hdr = nhdr;
      if (t + 1 < T) {
        const uint32_t* c2 = comp + ((size_t)(t + 1) * n + b) * CW;
        uu4 a = *(const uu4*)(c2 + qa), bb = *(const uu4*)(c2 + qb);
        nd[0] = a.x; nd[1] = a.y; nd[2] = a.z; nd[3] = a.w; nd[4] = bb.x; nd[5] = bb.y; nd[6] = bb.z; nd[7] = bb.w; nhdr = c2[NP];
      }
      float* o = out + ((size_t)t * n + b) * FB;
      EXPAND_PLANES(o, d, hdr);
    }
  } else {
    // wave 0: loader, two plies ahead; wave 1: storer.  One barrier per ply.
    if (wave == 0) {
      for (int pre = 0; pre < 2 && pre < T; ++pre) {
        const uint32_t* c = comp + ((size_t)pre * n + b) * CW;
        for (int i = lane; i < CW; i += 64) ring[pre][i] = c[i];
      }
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      if (wave == 0) {
        if (t + 2 < T) {
          const uint32_t* c = comp + ((size_t)(t + 2) * n + b) * CW;
          uint32_t* r = ring[(t + 2) % 3];
          for (int i = lane; i < CW; i += 64) r[i] = c[i];
        }
      } else {
        const uint32_t* r = ring[t % 3];
        uint32_t d[8];
        for (int i = 0; i < 4; ++i) { d[i] = r[qa + i]; d[4 + i] = r[qb + i]; }
        const uint32_t hdr = r[NP];
        float* o = out + ((size_t)t * n + b) * FB;
        EXPAND_PLANES(o, d, hdr);
      }
      __syncthreads();
    }
  }
#undef EXPAND_PLANES
}
static void run_expand_stream(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nbp = n * T;
  uint32_t* comp; float* lut;
  hipMalloc(&comp, (size_t)(nbp + 16) * CW * 4); hipMalloc(&lut, 2048 * 4);
  hipLaunchKernelGGL(init_comp, dim3(1024), dim3(256), 0, 0, comp, (size_t)(nbp + 16) * CW);
  hipMemset(lut, 0, 2048 * 4); hipDeviceSynchronize();
  const double bytes = (double)nbp * FB * 4;
  float ms;
  for (int var = 0; var < 2; ++var) {
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) hipEventRecord(e0);
      if (var == 0) hipLaunchKernelGGL(expand_stream<0>, dim3(n), dim3(64), 0, 0, comp, lut, out, n, T);
      else hipLaunchKernelGGL(expand_stream<1>, dim3(n), dim3(128), 0, 0, comp, lut, out, n, T);
    }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d expand_stream variant %d (%s)  %8.3f us per ply  %7.1f GB/s of planes\n", n, T, var,
           var ? "loader wave + storing wave" : "one wave, register prefetch", ms * 1e3 / T, bytes / (ms * 1e-3) / 1e9);
  }
}

// does line alignment of the wave-stores matter?  One wave per board (XCD-contiguous), the board's 36,100 B as a
// flat stream: align 0 = plane-major as the product (planes start every 1,444 B); 16 / 128 / 1024 = a head of
// single floats up to that alignment, then dwordx4 wave-stores of 1 KB each at that alignment, then a tail.
__global__ __launch_bounds__(64) void p_persist_flat(float* out, int n, int T, int align) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const int b = (w & 7) * (n >> 3) + (w >> 3);
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * FB;
    if (align == 0) {
      for (int pl = 0; pl < 25; ++pl) {
        float v = (float)(pl + t);
        uf4 val = {v, v + 1, v + 2, v + 3};
        *(uf4*)(o + pl * NP + 4 * lane) = val;
        if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
        if (lane == 26) o[pl * NP + 360] = v;
      }
    } else {
      const size_t a0 = (size_t)o, a1 = (a0 + align - 1) / align * align;
      const int head = (int)((a1 - a0) >> 2);              // floats before the aligned body
      if (lane < head) o[lane] = 1.f;
      if (lane + 64 < head) o[lane + 64] = 1.f;
      if (lane + 128 < head) o[lane + 128] = 1.f;
      if (lane + 192 < head) o[lane + 192] = 1.f;
      const int body = (FB - head) & ~255;                  // whole 1-KB wave-stores
      f4 val = {1.f, 2.f, 3.f, (float)t};
      for (int i = 0; i < body; i += 256) *(f4*)(o + head + i + 4 * lane) = val;
      for (int i = head + body + lane; i < FB; i += 64) o[i] = 2.f;
    }
  }
}
static void run_flat(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int al[4] = {0, 16, 128, 1024};
  for (int a = 0; a < 4; ++a) {
    for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(p_persist_flat, dim3(n), dim3(64), 0, 0, out, n, T, al[a]); }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d persistent 1 wave/board, flat stream aligned to %4d B  %8.3f us per ply  %7.1f GB/s\n", n, T, al[a],
           ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}

// cache-policy bits of the plane stores (gfx950: sc0, sc1, nt), persistent one wave per board, XCD-contiguous
template <int POL>
__device__ __forceinline__ void st_pol(float* p, f4 v) {
  if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int POL>
__global__ __launch_bounds__(64) void p_persist_pol(float* out, int n, int T) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const int b = (w & 7) * (n >> 3) + (w >> 3);
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * FB;
    for (int pl = 0; pl < 25; ++pl) {
      float v = (float)(pl + t);
      f4 val = {v, v + 1, v + 2, v + 3};
      st_pol<POL>(o + pl * NP + 4 * lane, val);
      if (lane < 26) st_pol<POL>(o + pl * NP + 256 + 4 * lane, val);
      if (lane == 26) o[pl * NP + 360] = v;
    }
  }
}
static void run_pol(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* nm[8] = {"(none)", "sc0", "sc1", "sc0 sc1", "nt", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
  for (int pol = 0; pol < 8; ++pol) {
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) hipEventRecord(e0);
      switch (pol) {
        case 0: hipLaunchKernelGGL(p_persist_pol<0>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 1: hipLaunchKernelGGL(p_persist_pol<1>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 2: hipLaunchKernelGGL(p_persist_pol<2>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 3: hipLaunchKernelGGL(p_persist_pol<3>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 4: hipLaunchKernelGGL(p_persist_pol<4>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 5: hipLaunchKernelGGL(p_persist_pol<5>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        case 6: hipLaunchKernelGGL(p_persist_pol<6>, dim3(n), dim3(64), 0, 0, out, n, T); break;
        default: hipLaunchKernelGGL(p_persist_pol<7>, dim3(n), dim3(64), 0, 0, out, n, T); break;
      }
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("n=%d T=%d persistent 1 wave/board, store policy %-11s %8.3f us per ply  %7.1f GB/s\n", n, T, nm[pol],
           ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
  }
}
// mode 8: does the memory side take the same bytes faster when the boards' samples are written IN ADDRESS ORDER IN TIME?
// One wave per board, persistent; board b starts its ply-t sample at T0 + t P + phase(b) P (100-MHz wall clock, common to
// the chip), so that with phase(b) = b / n the samples being written at any instant are a narrow, contiguous, moving window
// of the slot -- a write front made of whole samples.  phase mode 0: free-running (no clock); 1: b / n; 2: all boards at once;
// 3: hashed.  Reported: the time per ply actually achieved (>= P when paced).
__global__ __launch_bounds__(64) void p_persist_stagger(float* out, int n, int T, int map, int phase, int period_ns,
                                                        unsigned long long* t0p) {
  const int w = blockIdx.x, lane = threadIdx.x;
  int b = w;
  if (map == 1) b = (w & 7) * (n >> 3) + (w >> 3);
  unsigned long long t0 = 0;
  long long off_ns = 0;
  if (phase) {
    const unsigned long long now = wall_clock64();
    if (lane == 0) atomicMin(t0p, now);
    __builtin_amdgcn_s_sleep(127);
    t0 = __hip_atomic_load(t0p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 3000;   // 30 us after the first wave started
    t0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(t0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
    if (phase == 1) off_ns = (long long)b * period_ns / n;
    else if (phase == 3) off_ns = (long long)(((unsigned)b * 40503u) % (unsigned)n) * period_ns / n;
  }
  for (int t = 0; t < T; ++t) {
    if (phase) {
      const unsigned long long target = t0 + (unsigned long long)(((long long)t * period_ns + off_ns) / 10);
      for (int spin = 0; spin < (1 << 20) && wall_clock64() < target; ++spin) __builtin_amdgcn_s_sleep(1);
    }
    float* o = out + ((size_t)t * n + b) * FB;
    for (int pl = 0; pl < 25; ++pl) {
      float v = (float)(pl + t);
      uf4 val = {v, v + 1, v + 2, v + 3};
      *(uf4*)(o + pl * NP + 4 * lane) = val;
      if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
      if (lane == 26) o[pl * NP + 360] = v;
    }
  }
}
static void run_stagger(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long* t0p; hipMalloc(&t0p, 8);
  const int periods[6] = {5000, 5500, 6000, 6500, 7000, 8000};
  for (int map = 0; map < 2; ++map)
    for (int phase = 0; phase < 4; ++phase)
      for (int pi = 0; pi < (phase ? 6 : 1); ++pi) {
        float tot = 0;
        for (int r = 0; r < reps + 1; ++r) {
          hipMemsetAsync(t0p, 0xff, 8, 0);
          hipEventRecord(e0);
          hipLaunchKernelGGL(p_persist_stagger, dim3(n), dim3(64), 0, 0, out, n, T, map, phase, periods[pi], t0p);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (r) tot += ms;
        }
        float ms = tot / reps - (phase ? 0.030f : 0.f);   // the 30-us lead-in of the paced variants
        printf("n=%d T=%d map %d phase %d period %4d ns  %8.3f us per ply  %7.1f GB/s\n", n, T, map, phase, phase ? periods[pi] : 0,
               ms * 1e3 / T, (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
        fflush(stdout);
      }
}
// mode 9: paced stores (mode 8, hashed phases), which pattern does the memory side take fastest when it is NOT overdriven?
// var 0: rows per XCD (the product)   1: CU-contiguous rows   2: ring[b][t] instead of ring[t][b]
// var 3: a sample in two halves, each at half the period   4: two waves per board, a contiguous half each
// var 5: board = workgroup (rows interleaved over the XCDs)
__global__ __launch_bounds__(128) void p_paced(float* out, int n, int T, int var, int period_ns, unsigned long long* t0p) {
  const int w = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (var != 4 && wave) return;
  int b = (w & 7) * (n >> 3) + (w >> 3);
  if (var == 1) { const int cu = w & 255, k = w >> 8; b = cu * (n >> 8) + k; }
  if (var == 5) b = w;
  const unsigned long long now = wall_clock64();
  if (lane == 0) atomicMin(t0p, now);
  __builtin_amdgcn_s_sleep(127);
  unsigned long long t0 = __hip_atomic_load(t0p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 3000;
  t0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(t0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
  const long long off_ns = (long long)(((unsigned)b * 40503u) % (unsigned)n) * period_ns / n;
  const int halves = var == 3 ? 2 : 1;
  for (int t = 0; t < T; ++t) {
    float* o = out + (var == 2 ? ((size_t)b * T + t) : ((size_t)t * n + b)) * FB;
    for (int h = 0; h < halves; ++h) {
      const unsigned long long target = t0 + (unsigned long long)(((long long)t * period_ns + off_ns + (long long)h * period_ns / 2) / 10);
      for (int spin = 0; spin < (1 << 20) && wall_clock64() < target; ++spin) __builtin_amdgcn_s_sleep(1);
      int p0 = 0, p1 = 25;
      if (var == 3) { p0 = h ? 13 : 0; p1 = h ? 25 : 13; }
      if (var == 4) { p0 = wave ? 13 : 0; p1 = wave ? 25 : 13; }
      for (int pl = p0; pl < p1; ++pl) {
        float v = (float)(pl + t);
        uf4 val = {v, v + 1, v + 2, v + 3};
        *(uf4*)(o + pl * NP + 4 * lane) = val;
        if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
        if (lane == 26) o[pl * NP + 360] = v;
      }
    }
  }
}
static void run_paced(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long* t0p; hipMalloc(&t0p, 8);
  const int periods[7] = {5600, 5800, 5900, 6000, 6100, 6200, 6400};
  for (int var = 0; var < 6; ++var)
    for (int pi = 0; pi < 7; ++pi) {
      const int period = (int)((long long)periods[pi] * n / 1024);
      float tot = 0;
      for (int r = 0; r < reps + 1; ++r) {
        hipMemsetAsync(t0p, 0xff, 8, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(p_paced, dim3(n), dim3(128), 0, 0, out, n, T, var, period, t0p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) tot += ms;
      }
      float ms = tot / reps - 0.030f;
      printf("n=%d T=%d var %d period %5d ns  %8.3f us per ply  %7.1f GB/s\n", n, T, var, period, ms * 1e3 / T,
             (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
}
// mode 10: do partial cache lines matter?  A sample is 36,100 B, so rows of a ring slot start at every offset mod 128 and the line at
// each row boundary is written by two workgroups.  Paced stores as in mode 9 var 0 (rows per XCD, hashed phases) with
// var 0: the product's layout (row stride 9,025 floats)            1: rows padded to 9,056 floats (283 whole lines), same stores
// var 2: padded rows, written as a flat run of aligned 1-KB stores  3: the product's stride, flat run of 1-KB stores (unaligned)
__global__ __launch_bounds__(64) void p_paced_align(float* out, int n, int T, int var, int period_ns, unsigned long long* t0p) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const int b = (w & 7) * (n >> 3) + (w >> 3);
  const size_t stride = (var == 1 || var == 2) ? 9056 : FB;
  const unsigned long long now = wall_clock64();
  if (lane == 0) atomicMin(t0p, now);
  __builtin_amdgcn_s_sleep(127);
  unsigned long long t0 = __hip_atomic_load(t0p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 3000;
  t0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(t0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t0);
  const long long off_ns = (long long)(((unsigned)b * 40503u) % (unsigned)n) * period_ns / n;
  for (int t = 0; t < T; ++t) {
    float* o = out + ((size_t)t * n + b) * stride;
    const unsigned long long target = t0 + (unsigned long long)(((long long)t * period_ns + off_ns) / 10);
    for (int spin = 0; spin < (1 << 20) && wall_clock64() < target; ++spin) __builtin_amdgcn_s_sleep(1);
    if (var < 2) {
      for (int pl = 0; pl < 25; ++pl) {
        float v = (float)(pl + t);
        uf4 val = {v, v + 1, v + 2, v + 3};
        *(uf4*)(o + pl * NP + 4 * lane) = val;
        if (lane < 26) *(uf4*)(o + pl * NP + 256 + 4 * lane) = val;
        if (lane == 26) o[pl * NP + 360] = v;
      }
    } else {
      for (int k = 0; k < 35; ++k) {             // 35 x 256 floats = 8,960, then the tail of 65 floats
        float v = (float)(k + t);
        uf4 val = {v, v + 1, v + 2, v + 3};
        *(uf4*)(o + k * 256 + 4 * lane) = val;
      }
      if (lane < 16) { uf4 val = {1.f, 2.f, 3.f, 4.f}; *(uf4*)(o + 8960 + 4 * lane) = val; }
      if (lane == 16) o[9024] = 5.f;
    }
  }
}
static void run_paced_align(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long* t0p; hipMalloc(&t0p, 8);
  const int periods[7] = {5600, 5800, 5900, 6000, 6100, 6200, 6400};
  for (int var = 0; var < 4; ++var)
    for (int pi = 0; pi < 7; ++pi) {
      const int period = (int)((long long)periods[pi] * n / 1024);
      float tot = 0;
      for (int r = 0; r < reps + 1; ++r) {
        hipMemsetAsync(t0p, 0xff, 8, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(p_paced_align, dim3(n), dim3(64), 0, 0, out, n, T, var, period, t0p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) tot += ms;
      }
      float ms = tot / reps - 0.030f;
      printf("n=%d T=%d align var %d period %5d ns  %8.3f us per ply  %7.1f GB/s (of 36,100-B samples)\n", n, T, var, period, ms * 1e3 / T,
             (double)n * T * FB * 4 / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
}
static void run_front(float* out, int n, int T, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nbp = n * T;
  uint32_t* comp; float* lut;
  hipMalloc(&comp, (size_t)(nbp + 16) * CW * 4); hipMalloc(&lut, 2048 * 4);
  hipLaunchKernelGGL(init_comp, dim3(1024), dim3(256), 0, 0, comp, (size_t)(nbp + 16) * CW);
  hipMemset(lut, 0, 2048 * 4); hipDeviceSynchronize();
  const double bytes = (double)nbp * FB * 4;
  float ms;
  const int gs[6] = {1, 2, 4, 8, 32, 256};
  const int Gs[4] = {256, 512, 1024, 2048};
  for (int Gi = 0; Gi < 4; ++Gi)
    for (int gi = 0; gi < 6; ++gi) {
      if (Gi > 0 && gi > 1) continue;
      for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(fill_front, dim3(Gs[Gi]), dim3(256), 0, 0, out, nbp, gs[gi]); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("fill_front   %5d workgroups, teams of %3d  %9.3f us  %7.1f GB/s\n", Gs[Gi], gs[gi], ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    }
  for (int Gi = 0; Gi < 4; ++Gi) {
    for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(expand_board, dim3(Gs[Gi]), dim3(256), 0, 0, comp, lut, out, nbp); }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("expand_board %5d workgroups  %9.3f us  %7.1f GB/s\n", Gs[Gi], ms * 1e3, bytes / (ms * 1e-3) / 1e9);
  }
  for (int c = 0; c < 2; ++c)
    for (int Gi = 0; Gi < 4; ++Gi) {
      for (int r = 0; r < reps + 1; ++r) { if (r == 1) hipEventRecord(e0); hipLaunchKernelGGL(expand_flat, dim3(Gs[Gi]), dim3(256), 0, 0, comp, lut, out, nbp, c); }
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("expand_flat  %5d workgroups, %s  %9.3f us  %7.1f GB/s\n", Gs[Gi], c ? "expanding" : "fill only", ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    }
}
int main(int argc, char** argv) {
  int n = argc > 1 ? atoi(argv[1]) : 65536, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int T = argc > 3 ? atoi(argv[3]) : 0;   // > 0: the persistent variants only
  const int front = argc > 4 ? atoi(argv[4]) : 0; // 1: the two-stage (expander) probes over n*T board-plies
  float* out;
  if (hipMalloc(&out, (size_t)n * (T > 0 ? T : 1) * (front == 10 ? 9056 : FB) * 4 + 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  if (front == 10) { run_paced_align(out, n, T, reps); return 0; }
  if (front == 9) { run_paced(out, n, T, reps); return 0; }
  if (front == 8) { run_stagger(out, n, T, reps); return 0; }
  if (front == 7) { run_pol(out, n, T, reps); return 0; }
  if (front == 6) { run_flat(out, n, T, reps); return 0; }
  if (front == 5) { run_expand_stream(out, n, T, reps); return 0; }
  if (front == 4) { run_group(out, n, T, reps); return 0; }
  if (front == 3) { run_split(out, n, T, reps); return 0; }
  if (front == 2) { run_map(out, n, T, reps); return 0; }
  if (front) { run_front(out, n, T, reps); return 0; }
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
