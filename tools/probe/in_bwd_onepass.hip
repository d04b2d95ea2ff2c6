// Feasibility probe: a ONE-pass InstanceNorm backward.  A 1024-thread block owns (image, 16-channel chunk) of a
// 64x64x256 bf16 map, keeps x and g (2 x 128 KB) in REGISTERS (64 per thread), reduces the two per-channel sums through LDS and writes dx.
// Global accesses are 32 B per pixel at a 512 B stride: the four blocks that share a 128-byte line get consecutive logical ids on one XCD.
// Prints the time per launch and the implied HBM rate over the 3 tensors; compare with the two-pass kernels (62 us at B = 32).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
constexpr int HW = 4096, C = 256, CH = 16, NT = 1024, PPT = HW / NT;   // 4 pixels per thread

template <int XCD_AWARE>
__global__ __launch_bounds__(NT) void onepass(const char* __restrict__ x, const char* __restrict__ g, char* __restrict__ dx, const float* __restrict__ stats) {
  __shared__ float red[16][CH][2];
  __shared__ float coef[CH][3];
  const int G = gridDim.x;
  int L = XCD_AWARE && (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int img = L / (C / CH), ck = L % (C / CH);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t base = ((size_t)img * HW) * C * 2 + ck * CH * 2;
  u32x4 xv[PPT][2], gv[PPT][2];
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const size_t o = base + (size_t)(tid + NT * r) * C * 2;
    xv[r][0] = *(const u32x4*)(x + o); xv[r][1] = *(const u32x4*)(x + o + 16);
    gv[r][0] = *(const u32x4*)(g + o); gv[r][1] = *(const u32x4*)(g + o + 16);
  }
  float s1[CH], s2[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) s1[c] = s2[c] = 0.f;
#pragma unroll
  for (int r = 0; r < PPT; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = lo(xv[r][h][e]), x1 = hi(xv[r][h][e]), g0 = lo(gv[r][h][e]), g1 = hi(gv[r][h][e]);
        const int c = h * 8 + e * 2;
        s1[c] += g0; s2[c] += g0 * x0; s1[c + 1] += g1; s2[c + 1] += g1 * x1;
      }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1[c] += __shfl_xor(s1[c], o, 64); s2[c] += __shfl_xor(s2[c], o, 64); }
  }
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < CH; ++c) { red[wave][c][0] = s1[c]; red[wave][c][1] = s2[c]; }
  __syncthreads();
  if (tid < CH) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < 16; ++w) { a += red[w][tid][0]; b += red[w][tid][1]; }
    const float mean = stats[(img * C + ck * CH + tid) * 2], rstd = stats[(img * C + ck * CH + tid) * 2 + 1];
    const float S1 = a, S2 = (b - mean * a) * rstd;      // sum g, sum g * xhat
    coef[tid][0] = rstd;                                   // A
    coef[tid][1] = -rstd * rstd * S2 / HW;                 // Bc (multiplies x - mean, folded below)
    coef[tid][2] = -rstd * S1 / HW;                        // Cc
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    u32x4 o4[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = h * 8 + e * 2;
        const float x0 = lo(xv[r][h][e]), x1 = hi(xv[r][h][e]), g0 = lo(gv[r][h][e]), g1 = hi(gv[r][h][e]);
        const float d0 = g0 * coef[c][0] + x0 * coef[c][1] + coef[c][2], d1 = g1 * coef[c + 1][0] + x1 * coef[c + 1][1] + coef[c + 1][2];
        o4[h][e] = pk(d0, d1);
      }
    const size_t o = base + (size_t)(tid + NT * r) * C * 2;
    *(u32x4*)(dx + o) = o4[0]; *(u32x4*)(dx + o + 16) = o4[1];
  }
}

__global__ void copy3(const u32x4* __restrict__ x, const u32x4* __restrict__ g, u32x4* __restrict__ d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 a = x[i], b = g[i];
    d[i] = a ^ b;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  const size_t bytes = (size_t)B * HW * C * 2;
  char *x, *g, *dx; float* st;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&g, bytes)); CK(hipMalloc(&dx, bytes)); CK(hipMalloc(&st, (size_t)B * C * 2 * 4));
  CK(hipMemset(x, 0x3c, bytes)); CK(hipMemset(g, 0x3d, bytes)); CK(hipMemset(st, 0, (size_t)B * C * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = B * (C / CH);
  for (int variant = 0; variant < 3; ++variant) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) {
        if (variant == 0) hipLaunchKernelGGL(onepass<1>, dim3(grid), dim3(NT), 0, 0, x, g, dx, st);
        else if (variant == 1) hipLaunchKernelGGL(onepass<0>, dim3(grid), dim3(NT), 0, 0, x, g, dx, st);
        else hipLaunchKernelGGL(copy3, dim3(2048), dim3(256), 0, 0, (const u32x4*)x, (const u32x4*)g, (u32x4*)dx, bytes / 16);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double us = ms / 20 * 1e3;
    printf("B=%d %-28s %7.1f us per launch  %.2f TB/s over x + g + dx (%.0f MB)\n", B,
           variant == 0 ? "one-pass, XCD-aware ids" : variant == 1 ? "one-pass, plain ids" : "streaming copy x^g -> dx", us, 3.0 * bytes / us / 1e6, 3.0 * bytes / 1e6);
  }
  return 0;
}
