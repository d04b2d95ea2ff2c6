// Shared device/host helpers for libmi355x_gan (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mi355x_gan.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

int gan_set_error(int code, const char* fmt, ...);

#define GAN_CHECK(cond, ...)                                  \
  do {                                                        \
    if (!(cond)) return gan_set_error(-1, __VA_ARGS__);       \
  } while (0)

#define GAN_LAUNCH_CHECK()                                                              \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) return gan_set_error(-2, "launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// OCP e4m3 (GAN_FP8): v_cvt_pk_fp8_f32 rounds to nearest even and turns out-of-range values into NaN, so operands are clamped to +-448
__device__ __forceinline__ uint32_t f2e4m3x4(float a, float b, float c, float d) {
  auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };
  uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), 0u, false);
  return __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), r, true);
}

// 16-byte chunk of an activation: 4 fp32 or 8 bf16, handled as floats in registers.
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* v) {
    f32x4_t t = *reinterpret_cast<const f32x4_t*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    f32x4_t t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4_t*>(p) = t;
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
    u32x4_t t = *reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __builtin_bit_cast(float, t[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, t[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
    u32x4_t t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4_t*>(p) = t;
  }
};

// The same chunk as a raw 16-byte register quad (Raw<T>: f32x4 for fp32, u32x4 for bf16 -- each type is loaded as what it is): a
// kernel that keeps U pixels in flight issues its U loads back to back with ldraw() and converts with cvtraw() only when it
// computes, so that nothing between the loads waits for memory.
template <typename T> struct RawOf;
template <> struct RawOf<float> { typedef f32x4_t type; };
template <> struct RawOf<bf16_t> { typedef u32x4_t type; };
template <typename T> using Raw = typename RawOf<T>::type;
template <typename T> __device__ __forceinline__ Raw<T> ldraw(const T* p) { return *reinterpret_cast<const Raw<T>*>(p); }
template <typename T> __device__ __forceinline__ void cvtraw(const Raw<T>& t, float* v);
template <> __device__ __forceinline__ void cvtraw<float>(const f32x4_t& t, float* v) {
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void cvtraw<bf16_t>(const u32x4_t& t, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __builtin_bit_cast(float, t[i] << 16);
    v[2 * i + 1] = __builtin_bit_cast(float, t[i] & 0xffff0000u);
  }
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == GAN_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == GAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  if (act == GAN_ACT_TANH) return tanhf(v);
  return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
  if (act == GAN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == GAN_ACT_LRELU) return y > 0.f ? 1.f : 0.2f;
  if (act == GAN_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 16 floats */) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// reflect index into [0, n)
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// device-side copy of a view (kernel argument by value)
struct DView {
  char* ptr;
  int B, Hp, Wp, C, y0, x0, H, W;
  __device__ __forceinline__ int64_t pix(int b, int y, int x) const {  // element offset of logical pixel (y,x)
    return ((int64_t)(b * Hp + y + y0) * Wp + (x + x0)) * C;
  }
  __device__ __forceinline__ int64_t pixp(int b, int yp, int xp) const {  // padded coordinates
    return ((int64_t)(b * Hp + yp) * Wp + xp) * C;
  }
};
static inline DView to_dview(const gan_view* v) {
  DView d;
  d.ptr = (char*)v->ptr; d.B = v->B; d.Hp = v->Hp; d.Wp = v->Wp; d.C = v->C;
  d.y0 = v->y0; d.x0 = v->x0; d.H = v->H; d.W = v->W;
  return d;
}
static inline DView null_dview() { DView d = {}; return d; }
int gan_check_view(const gan_view* v, const char* name);

#define GAN_DISPATCH_DTYPE(dt, ...)                     \
  if ((dt) == GAN_F32) { typedef float T; __VA_ARGS__ } \
  else { typedef bf16_t T; __VA_ARGS__ }
