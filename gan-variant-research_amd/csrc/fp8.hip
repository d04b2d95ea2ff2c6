// e4m3 operand copies for the fp8 convolution path (BASELINE.json configs[4]: "CUT 512x512, fp8 MFMA conv path").
//
// The bottleneck 3x3 256->256 convolutions (GAN_Variant1/models/generator_resnet_attn.py:33,48) read e4m3 copies of their
// activation / output-gradient and weight operands on v_mfma_scale_f32_16x16x128_f8f6f4 (conv_patch.hip); everything else --
// results, InstanceNorm, master weights, optimiser -- keeps its precision.  This file makes the copies:
//   gan_quantize_fp8        activation / gradient buffer -> e4m3 buffer of the same geometry (halo included), unit scale or a
//                           per-image scale from max|x| (gan_in_bwd_amax);
//   gan_weight_scale_batch  per-tensor weight scale max|W| / 448 for gan_pack_weight(_batch) with dtype GAN_FP8.
// HBM-bound, 16-byte accesses.
#include "common.h"

namespace {

// src chunk pair (2 x 16 B of bf16 = 16 elements) -> one 16-byte e4m3 chunk
__global__ __launch_bounds__(256) void quantize_bf16_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int64_t nchunk16, int64_t per_image16,
                                                           const float* __restrict__ amax, float* __restrict__ scale_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk16; i += (int64_t)gridDim.x * blockDim.x) {
    float inv = 1.f;
    if (amax) {
      const int b = (int)(i / per_image16);
      const float am = amax[b], sc = am > 0.f ? am * (1.f / 448.f) : 1.f;
      inv = 1.f / sc;
      if (i == (int64_t)b * per_image16) scale_out[b] = sc;
    }
    const u32x4_t a = src[2 * i], c = src[2 * i + 1];
    float v[16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __builtin_bit_cast(float, a[k] << 16) * inv; v[2 * k + 1] = __builtin_bit_cast(float, a[k] & 0xffff0000u) * inv;
      v[8 + 2 * k] = __builtin_bit_cast(float, c[k] << 16) * inv; v[9 + 2 * k] = __builtin_bit_cast(float, c[k] & 0xffff0000u) * inv;
    }
    u32x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = f2e4m3x4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    dst[i] = o;
  }
}
__global__ __launch_bounds__(256) void quantize_f32_kernel(const f32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, int64_t nchunk16, int64_t per_image16,
                                                          const float* __restrict__ amax, float* __restrict__ scale_out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nchunk16; i += (int64_t)gridDim.x * blockDim.x) {
    float inv = 1.f;
    if (amax) {
      const int b = (int)(i / per_image16);
      const float am = amax[b], sc = am > 0.f ? am * (1.f / 448.f) : 1.f;
      inv = 1.f / sc;
      if (i == (int64_t)b * per_image16) scale_out[b] = sc;
    }
    u32x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const f32x4_t t = src[4 * i + k]; o[k] = f2e4m3x4(t[0] * inv, t[1] * inv, t[2] * inv, t[3] * inv); }
    dst[i] = o;
  }
}

// one block per pack descriptor: *scale = max|W| / 448
__global__ __launch_bounds__(1024) void weight_scale_kernel(const gan_pack_desc* __restrict__ descs) {
  const gan_pack_desc D = descs[blockIdx.x];
  if (D.dtype != GAN_FP8 || !D.scale) return;
  const int64_t n = (int64_t)(D.swap ? D.C_real : D.N_real) * D.I2 * D.KK;
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(D.src[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
    *D.scale = m > 0.f ? m * (1.f / 448.f) : 1.f;
  }
}

}  // namespace

extern "C" int gan_quantize_fp8(const gan_view* src, const gan_view* dst, const float* amax, float* scale_out, void* stream) {
  if (gan_check_view(src, "quantize_fp8.src") || gan_check_view(dst, "quantize_fp8.dst")) return -1;
  GAN_CHECK((src->dtype == GAN_BF16 || src->dtype == GAN_F32) && dst->dtype == GAN_FP8, "quantize_fp8: src must be bf16/fp32 and dst fp8");
  GAN_CHECK(src->B == dst->B && src->Hp == dst->Hp && src->Wp == dst->Wp && src->C == dst->C && src->y0 == dst->y0 && src->x0 == dst->x0 &&
            src->H == dst->H && src->W == dst->W, "quantize_fp8: src and dst geometry differ");
  GAN_CHECK((amax == nullptr) == (scale_out == nullptr), "quantize_fp8: amax and scale_out go together");
  const int64_t per_image16 = (int64_t)src->Hp * src->Wp * src->C / 16, n16 = per_image16 * src->B;
  const int grid = (int)((n16 + 255) / 256 < 8192 ? (n16 + 255) / 256 : 8192);
  if (src->dtype == GAN_BF16)
    hipLaunchKernelGGL(quantize_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4_t*)src->ptr, (u32x4_t*)dst->ptr, n16, per_image16, amax, scale_out);
  else
    hipLaunchKernelGGL(quantize_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4_t*)src->ptr, (u32x4_t*)dst->ptr, n16, per_image16, amax, scale_out);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_weight_scale_batch(const gan_pack_desc* descs, int n, void* stream) {
  GAN_CHECK(descs && n > 0, "weight_scale_batch: bad arguments");
  hipLaunchKernelGGL(weight_scale_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, descs);
  GAN_LAUNCH_CHECK();
  return 0;
}
