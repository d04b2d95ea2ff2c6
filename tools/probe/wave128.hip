// Probe: ONE wave per SIMD (256-thread block, 512 registers per lane) with a 128 x 128 wave tile -- 64 v_mfma_f32_16x16x32_bf16 per k-step
// against 8 + 8 operand fragments (activations from LDS, weights from L2 through a buffer descriptor, both one k-step ahead in a second
// register set).  Measures cycles per MFMA in the steady loop (ideal: 16).   hipcc --offload-arch=gfx950 -O3 wave128.hip -o wave128
// RESULT (round 3, ROCm 7.2): not reachable through hipcc.  With the MFMA builtin the 256 accumulators are split between VGPRs and AGPRs
// and shuffled (792 v_accvgpr moves, 148 B of scratch per lane); with the accumulators pinned to AGPRs by inline-asm MFMAs ("+a") the 64
// MFMAs of a k-step issue back to back, but the second operand set is spilled through scratch behind vmcnt(0) waits: 39.8 cycles per
// MFMA (869 TFLOP/s) against 20.9 in the shipped 2-waves-per-SIMD kernel.  This tile needs a hand-written (assembly) loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

constexpr int FI = 8, FJ = 8, KSTEPS = 18 * 4 * 2;   // two tiles of four slabs of 18 k-steps

__global__ __launch_bounds__(256) void probe(const char* w, int w_bytes, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 57344 / 16; i += 256) reinterpret_cast<u32x4_t*>(lds)[i] = u32x4_t{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, w_bytes, 0x00020000);
  const int lane16 = lane * 16;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4_t acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t W[2][FJ], X[2][FI];
  uint32_t xaddr[FI];
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int prow = wm * 128 + i * 16 + (lane & 15);
    xaddr[i] = (uint32_t)(prow * 128 + (((lane >> 4) ^ (prow & 7)) << 4));
  }
  auto w_load = [&](int ks, u32x4_t (&f)[FJ]) {
    const int base = __builtin_amdgcn_readfirstlane(((wn * FJ) * 162 + (ks % 162)) * 1024);
#pragma unroll
    for (int j = 0; j < FJ; ++j) f[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * 162 * 1024, 0));
  };
  auto x_load = [&](int ks, u32x4_t (&f)[FI]) {
    const uint32_t sh = (uint32_t)((ks % 9) * 130 * 128 % 16384) ^ ((ks & 1) ? 64u : 0u);
#pragma unroll
    for (int i = 0; i < FI; ++i) f[i] = *reinterpret_cast<const u32x4_t*>(lds + ((xaddr[i] + sh) & 0xffff));
  };
  w_load(0, W[0]);
  x_load(0, X[0]);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int ks = 0; ks < KSTEPS; ks += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      w_load(ks + h + 1, W[h ^ 1]);
      x_load(ks + h + 1, X[h ^ 1]);
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(W[h][j]), "v"(X[h][i]));
    }
    if ((ks % 18) == 16) __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int w_bytes = 16 * 162 * 1024;   // 256 n x (9 taps x 256 ch) bf16, fragment-major: 16 n-tiles x 162 k-blocks x 1 KB
  char* w; float* out; unsigned long long* cyc;
  hipMalloc(&w, w_bytes); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<uint16_t> hw(w_bytes / 2);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff);   // random-ish bf16 around 0.01
  hipMemcpy(w, hw.data(), w_bytes, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, w, w_bytes, out, cyc);
  hipEventRecord(e0);
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, w, w_bytes, out, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(256);
  hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double med = 0; for (auto c : hc) med += c; med /= 256;
  const double mfma = (double)KSTEPS * FI * FJ;
  printf("wave128: %.1f us per launch, %.0f cycles in the loop, %.2f cycles per MFMA (ideal 16), %.1f TFLOP/s\n", ms / 20 * 1e3, med, med / mfma,
         256.0 * 4 * mfma * 16384 / (ms / 20 * 1e-3) / 1e12);
  return 0;
}
