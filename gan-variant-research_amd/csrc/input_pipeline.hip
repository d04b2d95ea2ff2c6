// Device-side input pipeline: decoded uint8 RGB images -> the (B,3,S,S) fp32 batch in [-1,1] the trainers consume.
//
// Replaces, per image, the PIL chain the reference's datasets run on CPU workers
//   GAN_Variant1/dataio/transforms.py:10-49   RandomCropResize (crop + BICUBIC resize), RandomHorizontalFlip,
//                                             ColorJitter(0.05, 0.05, 0.05, 0.02), ToTensor, Normalize(0.5, 0.5); eval: Resize
//   Basic_GAN/src/data.py:8-26                Resize(load_size, BICUBIC), RandomCrop / CenterCrop, flip, ToTensor, Normalize
// bit for bit: Pillow's resize is integer (22-bit fixed-point taps, uint8 between the two passes), its ImageEnhance blends are
// float32 with truncation, its RGB<->HSV conversions mix float32 variables with double constants.  The arithmetic below follows
// Pillow 12.2 operation by operation (oracle/input_ref.py is the numpy restatement, pinned against Pillow itself), so floating-point
// contraction is off for this file and every rounding is the one the C code performs.
// All kernels are HBM-/latency-bound byte work: one pixel per lane, one image per blockIdx.y.
#include "common.h"
#include <cmath>

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---- horizontal pass: source rows of the crop box -> tmp [B][tmp_rows][S][4] (uint8 RGBX), only the window's columns
__global__ __launch_bounds__(256) void input_resize_h_kernel(const gan_input_job* __restrict__ jobs, const int32_t* __restrict__ tables,
                                                             uint8_t* __restrict__ tmp, int tmp_rows, int S) {
  const gan_input_job j = jobs[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= j.crop_h * S) return;
  const int y = idx / S, x = idx - y * S;
  const int32_t* b = tables + j.hb_off + 2 * (j.win_x + x);
  const int32_t* k = tables + j.hk_off + (int64_t)(j.win_x + x) * j.hksize;
  const int xmin = b[0], n = b[1];
  const uint8_t* row = j.src + (int64_t)(j.crop_y + y) * j.src_stride + (int64_t)(j.crop_x + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    s0 += row[3 * t] * w; s1 += row[3 * t + 1] * w; s2 += row[3 * t + 2] * w;
  }
  uchar4 o = make_uchar4((unsigned char)clip8(s0 >> PRECISION_BITS), (unsigned char)clip8(s1 >> PRECISION_BITS), (unsigned char)clip8(s2 >> PRECISION_BITS), 255);
  reinterpret_cast<uchar4*>(tmp)[((int64_t)blockIdx.y * tmp_rows + y) * S + x] = o;
}

// ---- vertical pass: tmp -> img [B][S][S][4], only the window's rows
__global__ __launch_bounds__(256) void input_resize_v_kernel(const gan_input_job* __restrict__ jobs, const int32_t* __restrict__ tables,
                                                             const uint8_t* __restrict__ tmp, int tmp_rows, int S, uint8_t* __restrict__ img) {
  const gan_input_job j = jobs[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * S) return;
  const int y = idx / S, x = idx - y * S;
  const int32_t* b = tables + j.vb_off + 2 * (j.win_y + y);
  const int32_t* k = tables + j.vk_off + (int64_t)(j.win_y + y) * j.vksize;
  const int ymin = b[0], n = b[1];
  const uchar4* col = reinterpret_cast<const uchar4*>(tmp) + ((int64_t)blockIdx.y * tmp_rows + ymin) * S + x;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    const uchar4 p = col[(int64_t)t * S];
    s0 += p.x * w; s1 += p.y * w; s2 += p.z * w;
  }
  uchar4 o = make_uchar4((unsigned char)clip8(s0 >> PRECISION_BITS), (unsigned char)clip8(s1 >> PRECISION_BITS), (unsigned char)clip8(s2 >> PRECISION_BITS), 255);
  reinterpret_cast<uchar4*>(img)[((int64_t)blockIdx.y * S + y) * S + x] = o;
}

// Convert.c rgb2l
__device__ __forceinline__ int luma(uchar4 p) { return (p.x * 19595 + p.y * 38470 + p.z * 7471 + 0x8000) >> 16; }

// ---- per-image grey mean of the current image state: int(sum / n + 0.5) (ImageEnhance.Contrast); one block per image
__global__ __launch_bounds__(1024) void input_gray_mean_kernel(const uint8_t* __restrict__ img, int S, int32_t* __restrict__ mean) {
  __shared__ unsigned long long sh[16];
  const uchar4* p = reinterpret_cast<const uchar4*>(img) + (int64_t)blockIdx.x * S * S;
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < S * S; i += 1024) s += (unsigned)luma(p[i]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < 16; ++i) t += sh[i];
    const unsigned long long n = (unsigned long long)S * S;
    mean[blockIdx.x] = (int32_t)((2 * t + n) / (2 * n));       // == int(t / n + 0.5): the quotient is never within 1/(2n) of a half below it
  }
}

// Blend.c ImagingBlend: float32 in1 + alpha * (in2 - in1); alpha in [0,1] truncates, otherwise clip first
__device__ __forceinline__ int blend1(int in1, int in2, float alpha, bool inside) {
  const float t = __fadd_rn((float)in1, __fmul_rn(alpha, (float)(in2 - in1)));
  if (inside) return (int)t & 255;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// Convert.c rgb2hsv_row / hsv2rgb with the H channel shifted in between (torchvision _functional_pil.adjust_hue)
__device__ __forceinline__ uchar4 hue_rotate(uchar4 px, int shift) {
  const int r = px.x, g = px.y, b = px.z;
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)((2.0 + (double)rc) - (double)bc);
    else h = (float)((4.0 + (double)gc) - (double)rc);
    const double t = (double)h / 6.0 + 1.0;
    h = (float)(t - trunc(t));                                   // fmod(t, 1.0), t > 0
    uh = clip8((int)((double)h * 255.0));
    us = clip8((int)((double)s * 255.0));
  }
  uh = (uh + shift) & 255;
  if (us == 0) return make_uchar4((unsigned char)uv, (unsigned char)uv, (unsigned char)uv, px.w);
  const double hf = (double)(float)uh * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const double f = (double)(float)(hf - (double)(float)i);
  const double fs = (double)(float)((double)(float)us / 255.0);
  const double vf = (double)(float)uv;
  const int p = clip8((int)round(vf * (1.0 - fs)));
  const int q = clip8((int)round(vf * (1.0 - fs * f)));
  const int t2 = clip8((int)round(vf * (1.0 - fs * (1.0 - f))));
  int R, G, B;
  switch (i % 6) {
    case 0: R = uv; G = t2; B = p; break;
    case 1: R = q; G = uv; B = p; break;
    case 2: R = p; G = uv; B = t2; break;
    case 3: R = p; G = q; B = uv; break;
    case 4: R = t2; G = p; B = uv; break;
    default: R = uv; G = p; B = q; break;
  }
  return make_uchar4((unsigned char)R, (unsigned char)G, (unsigned char)B, px.w);
}

// ---- one ColorJitter slot: image b applies its op order[slot] in place
__global__ __launch_bounds__(256) void input_jitter_kernel(const gan_input_job* __restrict__ jobs, uint8_t* __restrict__ img, int S, int slot,
                                                           const int32_t* __restrict__ mean) {
  const gan_input_job& j = jobs[blockIdx.y];
  const int op = j.order[slot];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (op < 0 || idx >= S * S) return;
  uchar4* p = reinterpret_cast<uchar4*>(img) + (int64_t)blockIdx.y * S * S + idx;
  uchar4 v = *p;
  if (op == 3) { *p = hue_rotate(v, j.hue_shift); return; }
  const float a = j.factor[op];
  const bool inside = a >= 0.f && a <= 1.f;
  int d0, d1, d2;                       // the degenerate image: black, mean grey, own grey
  if (op == 0) d0 = d1 = d2 = 0;
  else if (op == 1) d0 = d1 = d2 = mean[blockIdx.y];
  else d0 = d1 = d2 = luma(v);
  v.x = (unsigned char)blend1(d0, v.x, a, inside);
  v.y = (unsigned char)blend1(d1, v.y, a, inside);
  v.z = (unsigned char)blend1(d2, v.z, a, inside);
  *p = v;
}

// ---- flip + ToTensor + Normalize(0.5, 0.5): img [B][S][S][4] -> out fp32 [B][3][S][S]
__global__ __launch_bounds__(256) void input_finish_kernel(const gan_input_job* __restrict__ jobs, const uint8_t* __restrict__ img, int S, float* __restrict__ out) {
  const int flip = jobs[blockIdx.y].flip;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * S) return;
  const int y = idx / S, x = idx - y * S;
  const uchar4 v = reinterpret_cast<const uchar4*>(img)[((int64_t)blockIdx.y * S + y) * S + (flip ? S - 1 - x : x)];
  float* o = out + (int64_t)blockIdx.y * 3 * S * S + idx;
  const int c[3] = {v.x, v.y, v.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) o[(int64_t)k * S * S] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)c[k], 255.f), 0.5f), 0.5f);
}
}  // namespace

// ---- host: Pillow's bicubic taps (Resample.c precompute_coeffs + normalize_coeffs_8bpc), double arithmetic, no GPU involved
static double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

extern "C" int gan_resize_ksize(int in_size, int out_size) {
  if (in_size <= 0 || out_size <= 0) return gan_set_error(-1, "resize_ksize: sizes must be positive (%d -> %d)", in_size, out_size);
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(2.0 * filterscale) * 2 + 1;
}

extern "C" int gan_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize) {
  GAN_CHECK(in_size > 0 && out_size > 0 && bounds && kk, "resize_coeffs: bad arguments (%d -> %d)", in_size, out_size);
  GAN_CHECK(ksize == gan_resize_ksize(in_size, out_size), "resize_coeffs: ksize must be gan_resize_ksize(in, out) = %d, got %d",
            gan_resize_ksize(in_size, out_size), ksize);
  const double in0 = 0.0, in1 = (double)in_size;
  double scale = (in1 - in0) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
  double w[1024];
  GAN_CHECK(ksize <= 1024, "resize_coeffs: downscale factor too large (ksize %d)", ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) { w[x] = bicubic_filter((x + xmin - center + 0.5) * ss); ww += w[x]; }
    int32_t* k = kk + (int64_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
      if (x >= xmax) { k[x] = 0; continue; }
      const double p = ww != 0.0 ? w[x] / ww : w[x];
      k[x] = p < 0 ? (int32_t)(-0.5 + p * (1 << PRECISION_BITS)) : (int32_t)(0.5 + p * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
  }
  return 0;
}

extern "C" int gan_input_pipeline(const gan_input_job* jobs_dev, const gan_input_job* jobs_host, int B, const int32_t* tables_dev, int S,
                                  uint8_t* tmp, int tmp_rows, uint8_t* img, int32_t* mean_ws, float* out, void* stream) {
  GAN_CHECK(jobs_dev && jobs_host && tables_dev && tmp && img && mean_ws && out && B > 0 && S > 0, "input_pipeline: null pointer or empty batch");
  int max_rows = 0;
  bool slot_used[4] = {false, false, false, false}, slot_mean[4] = {false, false, false, false};
  for (int b = 0; b < B; ++b) {
    const gan_input_job& j = jobs_host[b];
    GAN_CHECK(j.src && j.crop_h > 0 && j.crop_w > 0 && j.crop_y >= 0 && j.crop_x >= 0 && j.src_stride >= 3 * (j.crop_x + j.crop_w),
              "input_pipeline: image %d: crop box (%d,%d,%d,%d) outside a row of %d bytes", b, j.crop_y, j.crop_x, j.crop_h, j.crop_w, j.src_stride);
    GAN_CHECK(j.win_y >= 0 && j.win_x >= 0 && j.win_y + S <= j.res_h && j.win_x + S <= j.res_w,
              "input_pipeline: image %d: %dx%d window at (%d,%d) outside the %dx%d resized image", b, S, S, j.win_y, j.win_x, j.res_h, j.res_w);
    GAN_CHECK(j.hksize == gan_resize_ksize(j.crop_w, j.res_w) && j.vksize == gan_resize_ksize(j.crop_h, j.res_h),
              "input_pipeline: image %d: tap counts do not match its sizes", b);
    GAN_CHECK(j.crop_h <= tmp_rows, "input_pipeline: image %d: %d source rows > tmp_rows %d", b, j.crop_h, tmp_rows);
    if (j.crop_h > max_rows) max_rows = j.crop_h;
    for (int s = 0; s < 4; ++s) {
      GAN_CHECK(j.order[s] >= -1 && j.order[s] <= 3, "input_pipeline: image %d: jitter op %d", b, j.order[s]);
      if (j.order[s] >= 0) slot_used[s] = true;
      if (j.order[s] == 1) slot_mean[s] = true;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(input_resize_h_kernel, dim3((max_rows * S + 255) / 256, B), dim3(256), 0, s, jobs_dev, tables_dev, tmp, tmp_rows, S);
  const int gp = (S * S + 255) / 256;
  hipLaunchKernelGGL(input_resize_v_kernel, dim3(gp, B), dim3(256), 0, s, jobs_dev, tables_dev, tmp, tmp_rows, S, img);
  for (int slot = 0; slot < 4; ++slot) {
    if (!slot_used[slot]) continue;
    if (slot_mean[slot]) hipLaunchKernelGGL(input_gray_mean_kernel, dim3(B), dim3(1024), 0, s, img, S, mean_ws);
    hipLaunchKernelGGL(input_jitter_kernel, dim3(gp, B), dim3(256), 0, s, jobs_dev, img, S, slot, mean_ws);
  }
  hipLaunchKernelGGL(input_finish_kernel, dim3(gp, B), dim3(256), 0, s, jobs_dev, img, S, out);
  GAN_LAUNCH_CHECK();
  return 0;
}
