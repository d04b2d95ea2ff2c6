// Probe: the 3x3 256->256 forward as HALF-CU blocks -- 256 threads = one wave per SIMD, tile 128 pixels x 256 channels (a wave: 128 x 64 = 8 x 4
// fragments, 128 accumulators), 32-channel slabs (64 B per pixel: a fragment read is 1 KB contiguous, no swizzle) staged by LDS-DMA into a
// double buffer of 2 x 20 KB, weights fragment-major from L2 and activations from LDS both one k-step ahead in a second register set.
// Two such blocks share a CU: the two waves of a SIMD then belong to DIFFERENT blocks (no common barrier, no lockstep).
//   hipcc --offload-arch=gfx950 -O3 conv_half.hip -o conv_half && ./conv_half [B]
// RESULT (round 3): correct (bf16-level agreement with a host reference), 252 registers, no scratch, two blocks per CU -- and SLOWER than the shipped
// 8-wave 256 x 256 tile: 160 us against 110 us on 32 images (76 against 60 on 16).  With the weight fetches compiled out it runs 118 us, without the
// LDS-DMA staging 142: a 128-pixel tile re-fetches every weight fragment for half as many pixels, and the L2 -> L1 path (the same 16 KB per k-step for
// all 256 CUs) is what limits it.  Independent half-CU blocks need a tile that keeps the weight bytes per pixel of the 256-pixel tile (256 pixels x
// 128 channels, staging the slab twice) -- i.e. they give back what the 256-channel tile gained.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

constexpr int FI = 8, FJ = 4, BM = 128, ROWS = 320, SLAB = ROWS * 64, NT = 9;

struct Args {
  const char* in; const char* w; char* out;
  int B, H, W, Cin, Hp, Wp, tiles_img, tiles, KB, w_bytes;
  int tap[NT];
};

__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

__global__ __launch_bounds__(256, 2) void conv_half(Args a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];   // [2][SLAB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  int tau = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tau >= a.tiles) return;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const int lane16 = lane * 16;
  const uint32_t pixb = (uint32_t)a.Cin * 2u;
  // staging role: DMA instruction q of this wave covers slab rows 16 * (wave * 5 + q) .. + 15 (lane -> row lane >> 2, 16-byte chunk lane & 3)
  auto stage = [&](int P0, int c, int buf) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int row = 16 * (wave * 5 + q) + (lane >> 2);
      const uint32_t off = (uint32_t)(P0 + row) * pixb + (uint32_t)(c * 64 + (lane & 3) * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.in + off),
                                       (__attribute__((address_space(3))) void*)(lds + buf * SLAB + (wave * 5 + q) * 1024), 16, 0, 0);
    }
  };
  auto geo = [&](int t, int& b, int& m0, int& P0) {
    b = t / a.tiles_img; m0 = (t - b * a.tiles_img) * BM;
    const int ho = m0 / a.W, wo = m0 - ho * a.W;
    P0 = (b * a.Hp + ho) * a.Wp + wo;
  };
  int b, m0, P0;
  geo(tau, b, m0, P0);
  stage(P0, 0, 0);
  u32x4_t Wr[2][FJ], Xr[2][FI];
  auto w_load = [&](int kb, u32x4_t (&f)[FJ]) {
    const int base = __builtin_amdgcn_readfirstlane(((wave * FJ) * a.KB + kb) * 1024);
#pragma unroll
    for (int j = 0; j < FJ; ++j) f[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * a.KB * 1024, 0));
  };
  const int cin32 = a.Cin >> 5;
  w_load(0, Wr[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  while (true) {
    uint32_t lb[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int d = i * 16 + (lane & 15);
      lb[i] = (uint32_t)(((d / a.W) * a.Wp + (d % a.W)) * 64 + (lane >> 4) * 16);
    }
    const int tau_next = tau + G;
    const bool has_next = tau_next < a.tiles;
    int bn = b, m0n = m0, P0n = P0;
    if (has_next) geo(tau_next, bn, m0n, P0n);
    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cin32; ++c) {
      const char* sb = lds + buf * SLAB;
      const bool last = c + 1 == cin32;
      if (!last) stage(P0, c + 1, buf ^ 1);
      else if (has_next) stage(P0n, 0, buf ^ 1);
      auto x_load = [&](int t, u32x4_t (&f)[FI]) {
        const uint32_t to = (uint32_t)(a.tap[t] * 64);
#pragma unroll
        for (int i = 0; i < FI; ++i) f[i] = *reinterpret_cast<const u32x4_t*>(sb + lb[i] + to);
      };
      x_load(0, Xr[0]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int cur = t & 1;
        // next k-step's operands: next tap of this slab, else tap 0 of the next slab (weights only: its activations wait for the barrier)
        const int kbn = t + 1 < NT ? (t + 1) * cin32 + c : (last ? 0 : c + 1);
        w_load(kbn, Wr[cur ^ 1]);
        if (t + 1 < NT) x_load(t + 1, Xr[cur ^ 1]);
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int j = 0; j < FJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, Wr[cur][j]), __builtin_bit_cast(bf16x8_t, Xr[cur][i]), acc[i][j], 0, 0, 0);
      }
      // NT is odd: the weights fetched by the last tap sit in set 1; the next slab starts from set 0
#pragma unroll
      for (int j = 0; j < FJ; ++j) Wr[0][j] = Wr[1][j];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next slab have landed
      __syncthreads();
      buf ^= 1;
    }
    // epilogue: D[n = (lane >> 4) * 4 + e][pixel = lane & 15] -> bf16 quads
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int m = m0 + i * 16 + (lane & 15);
      char* o = a.out + ((size_t)(b * a.H * a.W + m) * 256 + wave * 64 + (lane >> 4) * 4) * 2;
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        u32x2_t pk;
        pk[0] = (uint32_t)f2bf(acc[i][j][0]) | ((uint32_t)f2bf(acc[i][j][1]) << 16);
        pk[1] = (uint32_t)f2bf(acc[i][j][2]) | ((uint32_t)f2bf(acc[i][j][3]) << 16);
        *reinterpret_cast<u32x2_t*>(o + j * 32) = pk;
      }
    }
    if (!has_next) break;
    tau = tau_next; b = bn; m0 = m0n; P0 = P0n;
  }
}

static inline uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float h_bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, H = 64, W = 64, C = 256, Hp = H + 2, Wp = W + 2, KB = 9 * C / 32;
  std::vector<uint16_t> hin((size_t)B * Hp * Wp * C), hw((size_t)C * 9 * C), hwf((size_t)C * 9 * C);
  srand(1);
  for (auto& v : hin) v = h_f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : hw) v = h_f2bf((rand() % 2001 - 1000) / 20000.f);
  // fragment-major: element (n, k = t * C + c) at (((n / 16) * KB + k / 32) * 64 + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8
  for (int n = 0; n < C; ++n)
    for (int k = 0; k < 9 * C; ++k)
      hwf[(((size_t)(n / 16) * KB + k / 32) * 64 + ((k % 32) / 8) * 16 + n % 16) * 8 + k % 8] = hw[(size_t)n * 9 * C + k];
  char *din, *dw, *dout;
  hipMalloc(&din, hin.size() * 2 + 65536); hipMalloc(&dw, hwf.size() * 2); hipMalloc(&dout, (size_t)B * H * W * C * 2);
  hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hwf.data(), hwf.size() * 2, hipMemcpyHostToDevice);
  Args a;
  a.in = din; a.w = dw; a.out = dout; a.B = B; a.H = H; a.W = W; a.Cin = C; a.Hp = Hp; a.Wp = Wp;
  a.tiles_img = H * W / BM; a.tiles = B * a.tiles_img; a.KB = KB; a.w_bytes = (int)(hwf.size() * 2);
  for (int t = 0; t < 9; ++t) a.tap[t] = (t / 3) * Wp + t % 3;
  const int grid = a.tiles < 512 ? a.tiles : 512;
  hipFuncSetAttribute((const void*)conv_half, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(conv_half, dim3(grid), dim3(256), 2 * SLAB, 0, a);
  hipEventRecord(e0);
  const int iters = 20;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(conv_half, dim3(grid), dim3(256), 2 * SLAB, 0, a);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * B * H * W * C * C * 9;
  printf("conv_half B=%d: %.1f us per launch, %.1f TFLOP/s (err %s)\n", B, ms / iters * 1e3, flop / (ms / iters * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
  // spot check against a host reference
  std::vector<uint16_t> hout((size_t)B * H * W * C);
  hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int s = 0; s < 200; ++s) {
    const int b = rand() % B, y = rand() % H, x = rand() % W, n = rand() % C;
    double ref = 0;
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < C; ++c)
        ref += (double)h_bf2f(hin[((size_t)(b * Hp + y + t / 3) * Wp + x + t % 3) * C + c]) * h_bf2f(hw[(size_t)n * 9 * C + t * C + c]);
    const double got = h_bf2f(hout[((size_t)(b * H + y) * W + x) * C + n]);
    maxerr = fmax(maxerr, fabs(got - ref) / (fabs(ref) + 0.05));
  }
  printf("max relative error over 200 samples: %.4f\n", maxerr);
  return 0;
}
