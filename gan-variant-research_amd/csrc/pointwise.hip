// DiffAugment forward/backward, patch losses (hinge / LSGAN / BCE), L1 identity & cycle losses, R1 reduction.
//
// Replaces GAN_Variant1/training/diffaugment.py:6-60,94-106 (brightness, saturation, contrast, translation,
// cutout fused into one gather pass each way), GAN_Variant1/losses/adv_hinge.py:6-62,
// GAN_Variant1/losses/identity_l1.py:18-20, Basic_GAN/src/losses.py:5-30 and the reduction of
// r1_regularization (GAN_Variant1/training/train_cutpp.py:201).  All HBM-bound, one 16-byte chunk per lane.
#include "common.h"

namespace {

// store the first chunk of a C=8 pixel and zero the rest of the pixel (fp32 pixels are two chunks)
template <typename T> __device__ __forceinline__ void store_px8(T* p, const float* v) {
  Chunk<T>::store(p, v);
  if (Chunk<T>::N == 4) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; Chunk<T>::store(p + 4, z); }
}

struct AugP { float br, sat, con; int tx, ty, lo_h, hi_h, lo_w, hi_w; };
__device__ __forceinline__ AugP load_aug(const float* prm, int b) {
  const float* p = prm + b * 12;
  AugP a;
  a.br = p[0]; a.sat = p[1]; a.con = p[2];
  a.tx = (int)p[3]; a.ty = (int)p[4];
  a.lo_h = (int)p[5]; a.hi_h = (int)p[6]; a.lo_w = (int)p[7]; a.hi_w = (int)p[8];
  return a;
}

// per-image sum over real channels and logical pixels; one block per image
template <typename T>
__global__ __launch_bounds__(1024) void image_sum_kernel(DView x, int C, float* __restrict__ out) {
  constexpr int N = Chunk<T>::N;
  const int b = blockIdx.x, HW = x.H * x.W;
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    float v[N];
    Chunk<T>::load(xp + x.pix(b, p / x.W, p % x.W), v);
#pragma unroll
    for (int e = 0; e < N; ++e) if (e < C) s += v[e];
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[b] = s;
}

// y[b,h,w,:] = cutout(h,w) * inrange(h+tx,w+ty) * contrast(saturation(brightness(x[b,h+tx,w+ty,:])))
template <typename T>
__global__ void diffaug_fwd_kernel(DView x, int C, const float* __restrict__ prm, const float* __restrict__ sums, DView y) {
  constexpr int N = Chunk<T>::N;
  const int64_t total = (int64_t)x.B * x.H * x.W;
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  T* yp = reinterpret_cast<T*>(y.ptr);
  const float invn = 1.f / (float)(C * x.H * x.W);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % x.W);
    const int h = (int)((i / x.W) % x.H);
    const int b = (int)(i / ((int64_t)x.W * x.H));
    const AugP a = load_aug(prm, b);
    const int sh_ = h + a.tx, sw = w + a.ty;
    const bool cut = h >= a.lo_h && h <= a.hi_h && w >= a.lo_w && w <= a.hi_w;
    float v[N];
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = 0.f;
    if (!cut && sh_ >= 0 && sh_ < x.H && sw >= 0 && sw < x.W) {
      float t[N];
      Chunk<T>::load(xp + x.pix(b, sh_, sw), t);
      float mc = 0.f;
#pragma unroll
      for (int e = 0; e < N; ++e) if (e < C) { t[e] += a.br; mc += t[e]; }
      mc /= (float)C;
      const float mu = sums[b] * invn + a.br;  // mean over (c,h,w) after brightness; saturation keeps it
#pragma unroll
      for (int e = 0; e < N; ++e) if (e < C) {
        const float s = (t[e] - mc) * a.sat + mc;
        v[e] = (s - mu) * a.con + mu;
      }
    }
    store_px8<T>(yp + y.pix(b, h, w), v);
  }
}

// per-image sum of gy over the output pixels that took a value from x (valid source, not cut)
template <typename T>
__global__ __launch_bounds__(1024) void diffaug_gsum_kernel(DView gy, int C, const float* __restrict__ prm, float* __restrict__ out) {
  constexpr int N = Chunk<T>::N;
  const int b = blockIdx.x, HW = gy.H * gy.W;
  const AugP a = load_aug(prm, b);
  const T* gp = reinterpret_cast<const T*>(gy.ptr);
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const int h = p / gy.W, w = p % gy.W;
    const int sh_ = h + a.tx, sw = w + a.ty;
    const bool cut = h >= a.lo_h && h <= a.hi_h && w >= a.lo_w && w <= a.hi_w;
    if (cut || sh_ < 0 || sh_ >= gy.H || sw < 0 || sw >= gy.W) continue;
    float v[N];
    Chunk<T>::load(gp + gy.pix(b, h, w), v);
#pragma unroll
    for (int e = 0; e < N; ++e) if (e < C) s += v[e];
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[b] = s;
}

template <typename T>
__global__ void diffaug_bwd_kernel(DView gy, int C, const float* __restrict__ prm, const float* __restrict__ gsum, DView gx) {
  constexpr int N = Chunk<T>::N;
  const int64_t total = (int64_t)gx.B * gx.H * gx.W;
  const T* gp = reinterpret_cast<const T*>(gy.ptr);
  T* xp = reinterpret_cast<T*>(gx.ptr);
  const float invn = 1.f / (float)(C * gx.H * gx.W);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % gx.W);
    const int h = (int)((i / gx.W) % gx.H);
    const int b = (int)(i / ((int64_t)gx.W * gx.H));
    const AugP a = load_aug(prm, b);
    const int oh = h - a.tx, ow = w - a.ty;  // the output pixel that read source pixel (h,w)
    float g[N];
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = 0.f;
    if (oh >= 0 && oh < gx.H && ow >= 0 && ow < gx.W && !(oh >= a.lo_h && oh <= a.hi_h && ow >= a.lo_w && ow <= a.hi_w))
      Chunk<T>::load(gp + gy.pix(b, oh, ow), g);
    const float gm = gsum[b] * invn;
    float mc = 0.f;
#pragma unroll
    for (int e = 0; e < N; ++e) if (e < C) { g[e] = a.con * g[e] + (1.f - a.con) * gm; mc += g[e]; }
    mc /= (float)C;
    float v[N];
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = e < C ? a.sat * g[e] + (1.f - a.sat) * mc : 0.f;
    store_px8<T>(xp + gx.pix(b, h, w), v);
  }
}

// ---- patch losses on channel 0 of the logits view; single block (B*H*W is a few thousand)
template <typename T>
__global__ __launch_bounds__(1024) void patch_loss_kernel(DView x, int mode, float target, float scale, float* __restrict__ loss, DView g,
                                                         int has_g) {
  constexpr int N = Chunk<T>::N;
  const int n = x.B * x.H * x.W;
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  T* gp = reinterpret_cast<T*>(g.ptr);
  const float inv = scale / (float)n;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int w = i % x.W, h = (i / x.W) % x.H, b = i / (x.W * x.H);
    const float v = ld1<T>(xp + x.pix(b, h, w));
    float f, d;
    if (mode == 0) { f = fmaxf(1.f - v, 0.f); d = v < 1.f ? -1.f : 0.f; }
    else if (mode == 1) { f = fmaxf(1.f + v, 0.f); d = v > -1.f ? 1.f : 0.f; }
    else if (mode == 2) { f = -v; d = -1.f; }
    else if (mode == 3) { f = (v - target) * (v - target); d = 2.f * (v - target); }
    else { f = fmaxf(v, 0.f) - v * target + log1pf(expf(-fabsf(v))); d = 1.f / (1.f + expf(-v)) - target; }
    s += f;
    if (has_g) {
      float o[N];
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = 0.f;
      o[0] = d * inv;
      store_px8<T>(gp + g.pix(b, h, w), o);
    }
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *loss = s * inv;
}

// ---- L1: loss = scale * mean|x - t|, grad = scale * sign(x - t) / n.  stage 1 partials -> ws, stage 2 sums.
template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(DView x, int C, const float* __restrict__ tgt, float gscale, const float* __restrict__ dev_scale,
                                                DView g, int has_g, float* __restrict__ ws) {
  if (dev_scale) gscale *= *dev_scale;
  constexpr int N = Chunk<T>::N;
  const int64_t total = (int64_t)x.B * x.H * x.W;
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  T* gp = reinterpret_cast<T*>(g.ptr);
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % x.W), h = (int)((i / x.W) % x.H), b = (int)(i / ((int64_t)x.W * x.H));
    float v[N], o[N];
    Chunk<T>::load(xp + x.pix(b, h, w), v);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      o[e] = 0.f;
      if (e < C) {
        const float d = v[e] - tgt[(((int64_t)b * C + e) * x.H + h) * x.W + w];
        s += fabsf(d);
        o[e] = d > 0.f ? gscale : (d < 0.f ? -gscale : 0.f);
      }
    }
    if (has_g) store_px8<T>(gp + g.pix(b, h, w), o);
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ void sum_scale_kernel(const float* __restrict__ ws, int n, float scale, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += ws[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *out = s * scale;
}

// ---- R1: partial sums of g^2 and u = uscale * g
template <typename T>
__global__ __launch_bounds__(256) void r1_kernel(DView g, int C, float uscale, DView u, int has_u, float* __restrict__ ws) {
  constexpr int N = Chunk<T>::N;
  const int64_t total = (int64_t)g.B * g.H * g.W;
  const T* gp = reinterpret_cast<const T*>(g.ptr);
  T* up = reinterpret_cast<T*>(u.ptr);
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % g.W), h = (int)((i / g.W) % g.H), b = (int)(i / ((int64_t)g.W * g.H));
    float v[N], o[N];
    Chunk<T>::load(gp + g.pix(b, h, w), v);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      o[e] = 0.f;
      if (e < C) { s += v[e] * v[e]; o[e] = uscale * v[e]; }
    }
    if (has_u) store_px8<T>(up + u.pix(b, h, w), o);
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

}  // namespace

#define VCHK(v, name) do { if (gan_check_view(v, name)) return -1; } while (0)

extern "C" int gan_diffaug_fwd(const gan_view* x, int C, const float* prm, const gan_view* y, float* ws, void* stream) {
  VCHK(x, "diffaug_fwd.x"); VCHK(y, "diffaug_fwd.y");
  GAN_CHECK(x->C == 8 && y->C == 8 && C > 0 && C <= 4 && x->dtype == y->dtype, "diffaug: views must have C=8 and the same dtype (C real <= 4)");
  GAN_CHECK(x->B == y->B && x->H == y->H && x->W == y->W && prm && ws, "diffaug: shape mismatch");
  hipStream_t s = (hipStream_t)stream;
  DView vx = to_dview(x), vy = to_dview(y);
  const int64_t total = (int64_t)x->B * x->H * x->W;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  GAN_DISPATCH_DTYPE(x->dtype,
    hipLaunchKernelGGL((image_sum_kernel<T>), dim3(x->B), dim3(1024), 0, s, vx, C, ws);
    hipLaunchKernelGGL((diffaug_fwd_kernel<T>), dim3(grid), dim3(256), 0, s, vx, C, prm, ws, vy);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_diffaug_bwd(const gan_view* gy, int C, const float* prm, const gan_view* gx, float* ws, void* stream) {
  VCHK(gy, "diffaug_bwd.gy"); VCHK(gx, "diffaug_bwd.gx");
  GAN_CHECK(gy->C == 8 && gx->C == 8 && C > 0 && C <= 4 && gy->dtype == gx->dtype, "diffaug: views must have C=8 and the same dtype");
  GAN_CHECK(gx->B == gy->B && gx->H == gy->H && gx->W == gy->W && prm && ws, "diffaug: shape mismatch");
  hipStream_t s = (hipStream_t)stream;
  DView vg = to_dview(gy), vx = to_dview(gx);
  const int64_t total = (int64_t)gx->B * gx->H * gx->W;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  GAN_DISPATCH_DTYPE(gy->dtype,
    hipLaunchKernelGGL((diffaug_gsum_kernel<T>), dim3(gy->B), dim3(1024), 0, s, vg, C, prm, ws);
    hipLaunchKernelGGL((diffaug_bwd_kernel<T>), dim3(grid), dim3(256), 0, s, vg, C, prm, ws, vx);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_patch_loss(const gan_view* logits, int mode, float target, float scale, float* loss, const gan_view* grad, void* stream) {
  VCHK(logits, "patch_loss.logits");
  GAN_CHECK(mode >= 0 && mode <= 4 && loss, "patch_loss: bad mode %d", mode);
  if (grad) {
    VCHK(grad, "patch_loss.grad");
    GAN_CHECK(grad->B == logits->B && grad->H == logits->H && grad->W == logits->W && grad->C == 8 && grad->dtype == logits->dtype,
              "patch_loss: grad view must match the logits (C=8)");
  }
  GAN_CHECK((int64_t)logits->B * logits->H * logits->W < (1 << 24), "patch_loss: too many logits for the single-block reduction");
  DView vx = to_dview(logits), vg = grad ? to_dview(grad) : null_dview();
  GAN_DISPATCH_DTYPE(logits->dtype, hipLaunchKernelGGL((patch_loss_kernel<T>), dim3(1), dim3(1024), 0, (hipStream_t)stream, vx, mode, target,
                                                       scale, loss, vg, grad ? 1 : 0);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_l1_loss(const gan_view* x, int C, const float* target_nchw, float scale, const float* dev_grad_scale, float* loss,
                           const gan_view* grad, float* ws, void* stream) {
  VCHK(x, "l1_loss.x");
  GAN_CHECK(x->C == 8 && C > 0 && C <= 4 && target_nchw && loss && ws, "l1_loss: x must have C=8 (<= 4 real channels)");
  if (grad) {
    VCHK(grad, "l1_loss.grad");
    GAN_CHECK(grad->B == x->B && grad->H == x->H && grad->W == x->W && grad->C == 8 && grad->dtype == x->dtype, "l1_loss: grad view mismatch");
  }
  const int64_t n = (int64_t)x->B * C * x->H * x->W;
  const int nblk = 512;
  DView vx = to_dview(x), vg = grad ? to_dview(grad) : null_dview();
  hipStream_t s = (hipStream_t)stream;
  GAN_DISPATCH_DTYPE(x->dtype, hipLaunchKernelGGL((l1_kernel<T>), dim3(nblk), dim3(256), 0, s, vx, C, target_nchw, scale / (float)n, dev_grad_scale, vg,
                                                  grad ? 1 : 0, ws);)
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, ws, nblk, scale / (float)n, loss);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_r1_reduce(const gan_view* g, int C, float scale, float* loss, const gan_view* u, float* ws, void* stream) {
  VCHK(g, "r1_reduce.g");
  GAN_CHECK(g->C == 8 && C > 0 && C <= 4 && loss && ws, "r1_reduce: g must have C=8 (<= 4 real channels)");
  if (u) {
    VCHK(u, "r1_reduce.u");
    GAN_CHECK(u->B == g->B && u->H == g->H && u->W == g->W && u->C == 8 && u->dtype == g->dtype, "r1_reduce: u view mismatch");
  }
  const int nblk = 512;
  DView vg = to_dview(g), vu = u ? to_dview(u) : null_dview();
  hipStream_t s = (hipStream_t)stream;
  GAN_DISPATCH_DTYPE(g->dtype, hipLaunchKernelGGL((r1_kernel<T>), dim3(nblk), dim3(256), 0, s, vg, C, scale * 2.f / (float)g->B, vu, u ? 1 : 0, ws);)
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, s, ws, nblk, 1.f / (float)g->B, loss);
  GAN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ AvgPool2d(3, stride 2, padding 1, count_include_pad=False)
// The downsampling between the scales of MultiscaleDiscriminator (GAN_Variant1/models/discriminator_patchgan.py:100, 110-112):
// every output pixel is the mean of the taps of its 3x3 window that fall inside the image.  One 16-byte chunk per lane.
namespace {
template <typename T>
__global__ void avgpool_fwd_kernel(DView x, DView y) {
  constexpr int N = Chunk<T>::N;
  const int nck = y.C / N;
  const int64_t total = (int64_t)y.B * y.H * y.W * nck;
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  T* yp = reinterpret_cast<T*>(y.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % nck);
    int64_t r = i / nck;
    const int ox = (int)(r % y.W); r /= y.W;
    const int oy = (int)(r % y.H);
    const int b = (int)(r / y.H);
    float s[N];
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = 0.f;
    int cnt = 0;
    for (int dy = -1; dy <= 1; ++dy) {
      const int iy = 2 * oy + dy;
      if (iy < 0 || iy >= x.H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int ix = 2 * ox + dx;
        if (ix < 0 || ix >= x.W) continue;
        float v[N];
        Chunk<T>::load(xp + x.pix(b, iy, ix) + ck * N, v);
#pragma unroll
        for (int e = 0; e < N; ++e) s[e] += v[e];
        ++cnt;
      }
    }
    const float d = (float)cnt;
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] = s[e] / d;
    Chunk<T>::store(yp + y.pix(b, oy, ox) + ck * N, s);
  }
}

__device__ __forceinline__ int pool_taps(int o, int n) {   // taps of output index o that lie in [0, n)
  return (2 * o - 1 >= 0 ? 1 : 0) + 1 + (2 * o + 1 < n ? 1 : 0);
}

template <typename T>
__global__ void avgpool_bwd_kernel(DView gy, DView gx, int accumulate) {
  constexpr int N = Chunk<T>::N;
  const int nck = gx.C / N;
  const int64_t total = (int64_t)gx.B * gx.H * gx.W * nck;
  const T* gp = reinterpret_cast<const T*>(gy.ptr);
  T* xp = reinterpret_cast<T*>(gx.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % nck);
    int64_t r = i / nck;
    const int ix = (int)(r % gx.W); r /= gx.W;
    const int iy = (int)(r % gx.H);
    const int b = (int)(r / gx.H);
    float s[N];
    T* dst = xp + gx.pix(b, iy, ix) + ck * N;
    if (accumulate) Chunk<T>::load(dst, s);
    else {
#pragma unroll
      for (int e = 0; e < N; ++e) s[e] = 0.f;
    }
    // outputs whose window covers input pixel i: 2o-1 <= i <= 2o+1  <=>  o in {ceil((i-1)/2) .. floor((i+1)/2)}
    for (int oy = iy >> 1; oy <= (iy + 1) >> 1; ++oy) {
      if (oy >= gy.H) continue;
      const int ny = pool_taps(oy, gx.H);
      for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ++ox) {
        if (ox >= gy.W) continue;
        const float inv = 1.f / (float)(ny * pool_taps(ox, gx.W));
        float v[N];
        Chunk<T>::load(gp + gy.pix(b, oy, ox) + ck * N, v);
#pragma unroll
        for (int e = 0; e < N; ++e) s[e] += v[e] * inv;
      }
    }
    Chunk<T>::store(dst, s);
  }
}
}  // namespace

static int avgpool_check(const gan_view* big, const gan_view* small, const char* what) {
  GAN_CHECK(big->B == small->B && big->C == small->C && big->dtype == small->dtype, "%s: batch / channel / dtype mismatch", what);
  GAN_CHECK(small->H == (big->H - 1) / 2 + 1 && small->W == (big->W - 1) / 2 + 1, "%s: pooled size must be ((H-1)/2+1, (W-1)/2+1) = (%d,%d), got (%d,%d)",
            what, (big->H - 1) / 2 + 1, (big->W - 1) / 2 + 1, small->H, small->W);
  return 0;
}

extern "C" int gan_avgpool_fwd(const gan_view* x, const gan_view* y, void* stream) {
  VCHK(x, "avgpool_fwd.x"); VCHK(y, "avgpool_fwd.y");
  if (avgpool_check(x, y, "avgpool_fwd")) return -1;
  const int epc = y->dtype == GAN_F32 ? 4 : 8;
  const int64_t total = (int64_t)y->B * y->H * y->W * (y->C / epc);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView vx = to_dview(x), vy = to_dview(y);
  GAN_DISPATCH_DTYPE(y->dtype, hipLaunchKernelGGL((avgpool_fwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, vx, vy);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_avgpool_bwd(const gan_view* gy, const gan_view* gx, int accumulate, void* stream) {
  VCHK(gy, "avgpool_bwd.gy"); VCHK(gx, "avgpool_bwd.gx");
  if (avgpool_check(gx, gy, "avgpool_bwd")) return -1;
  const int epc = gx->dtype == GAN_F32 ? 4 : 8;
  const int64_t total = (int64_t)gx->B * gx->H * gx->W * (gx->C / epc);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView vg = to_dview(gy), vx = to_dview(gx);
  GAN_DISPATCH_DTYPE(gx->dtype, hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, vg, vx, accumulate);)
  GAN_LAUNCH_CHECK();
  return 0;
}
