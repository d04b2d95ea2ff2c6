// InstanceNorm2d forward/backward fused with activation, residual add, reflection/zero halo fill and the
// reflection-pad gradient fold; plus the NCHW <-> halo-NHWC boundary conversions.
//
// Replaces nn.InstanceNorm2d (aten::native_batch_norm on a (1,B*C,H,W) view), nn.ReLU / nn.LeakyReLU,
// nn.ReflectionPad2d and the residual add of the reference (GAN_Variant1/models/generator_resnet_attn.py:
// 25,43,56,64,71,111,114-115,126-127,150-151,158; Basic_GAN/src/models.py:10-18,30-31,38-39,52-53,91-92,99-100).
// All kernels are HBM-bound: every access is one 16-byte channel chunk per lane, consecutive lanes on
// consecutive chunks of a pixel (halo-NHWC rows are contiguous), statistics accumulate in fp32 and are
// combined in fp64.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NTHR = 256;
constexpr int MAXCH = 96;  // max row-chunks per image for the two-stage statistics

// thread -> (chunk lane cl in [0,CL), row lane rl in [0,RL)), CL = C/EPC (power of two <= 256)
template <typename T> struct Lanes {
  static constexpr int EPC = Chunk<T>::N;
  int CL, RL, cl, rl;
  __device__ Lanes(int C) { CL = C / EPC; RL = NTHR / CL; cl = threadIdx.x % CL; rl = threadIdx.x / CL; }
};

__device__ __forceinline__ void pixel_yx(int p, int W, int& y, int& x) { y = p / W; x = p - y * W; }
// (y, x) of pixel p + step without a division per pixel; returns the number of row wraps
__device__ __forceinline__ int pixel_step(int step, int W, int& y, int& x) {
  x += step;
  int wraps = 0;
  while (x >= W) { x -= W; ++y; ++wraps; }
  return wraps;
}
// element offset of a view's current pixel, advanced with the walk (these kernels were VALU-bound on 64-bit address
// arithmetic: ~100 instructions per 3 memory operations)
struct Off {
  int64_t o, dstep, dwrap;
  __device__ __forceinline__ Off(const DView& v, int b, int y, int x, int step, int W) : o(v.pix(b, y, x)), dstep((int64_t)step * v.C), dwrap((int64_t)(v.Wp - W) * v.C) {}
  __device__ __forceinline__ void advance(int wraps) { o += dstep + wraps * dwrap; }
};

// sum of the padded-domain gradient over the reflect pre-images of logical pixel (y,x); pad = g.y0
template <typename T>
__device__ __forceinline__ void load_folded(const DView& g, int fold, int b, int y, int x, int64_t off, int cofs, float* v) {
  constexpr int N = Chunk<T>::N;
  const T* p = reinterpret_cast<const T*>(g.ptr);
  Chunk<T>::load(p + off + cofs, v);   // off = g.pix(b, y, x), maintained incrementally by the caller
  if (!fold) return;
  // mirror partners in the padded domain (-1: none).  Nearly every pixel has none: one compare pair, no extra loads.
  const int py = g.y0, px = g.x0;
  int y2 = -1, x2 = -1;
  if (y >= 1 && y <= py) y2 = py - y;
  else if (y >= g.H - 1 - py && y <= g.H - 2) y2 = 2 * (g.H - 1) - y + py;
  if (x >= 1 && x <= px) x2 = px - x;
  else if (x >= g.W - 1 - px && x <= g.W - 2) x2 = 2 * (g.W - 1) - x + px;
  if ((y2 & x2) < 0 && (y2 | x2) < 0) return;   // both -1
  float t[N];
  if (x2 >= 0) {
    Chunk<T>::load(p + g.pixp(b, y + py, x2) + cofs, t);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += t[e];
  }
  if (y2 >= 0) {
    Chunk<T>::load(p + g.pixp(b, y2, x + px) + cofs, t);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += t[e];
    if (x2 >= 0) {
      Chunk<T>::load(p + g.pixp(b, y2, x2) + cofs, t);
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] += t[e];
    }
  }
}

// ------------------------------------------------------------------ forward statistics
// ws[((b*nch + ch)*C + c)*2 + {0,1}] = partial (sum, sum of squares)
template <typename T>
__global__ __launch_bounds__(NTHR) void in_partial_kernel(DView x, int nch, float* __restrict__ ws) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, ch = blockIdx.x, HW = x.H * x.W;
  const int per = (HW + nch - 1) / nch, p0 = ch * per, p1 = min(HW, p0 + per);
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  float s[N], q[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = q[e] = 0.f;
  int y, xx; pixel_yx(p0 + L.rl, x.W, y, xx);
  Off ox(x, b, y, xx, L.RL, x.W);
  for (int p = p0 + L.rl; p < p1; p += L.RL, ox.advance(pixel_step(L.RL, x.W, y, xx))) {
    float v[N];
    Chunk<T>::load(xp + ox.o + L.cl * N, v);
#pragma unroll
    for (int e = 0; e < N; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
  }
  __shared__ float sh[NTHR * 16];
#pragma unroll
  for (int e = 0; e < N; ++e) { sh[(threadIdx.x * N + e) * 2] = s[e]; sh[(threadIdx.x * N + e) * 2 + 1] = q[e]; }
  __syncthreads();
  if (L.rl == 0) {
    for (int r = 1; r < L.RL; ++r)
#pragma unroll
      for (int e = 0; e < N; ++e) {
        s[e] += sh[((r * L.CL + L.cl) * N + e) * 2];
        q[e] += sh[((r * L.CL + L.cl) * N + e) * 2 + 1];
      }
    float* o = ws + ((int64_t)(b * nch + ch) * x.C + L.cl * N) * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { o[2 * e] = s[e]; o[2 * e + 1] = q[e]; }
  }
}
// stats[(b*C+c)*2] = mean, +1 = rstd.  One block per 32 (b,c) pairs, 8 partial lanes each.
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ ws, int nch, int C, int BC, int HW, float eps, float* __restrict__ stats) {
  const int i = blockIdx.x * 32 + (threadIdx.x & 31), k0 = threadIdx.x >> 5;
  double s = 0, q = 0;
  if (i < BC) {
    const int b = i / C, c = i - b * C;
    for (int k = k0; k < nch; k += 8) {
      const float* p = ws + ((int64_t)(b * nch + k) * C + c) * 2;
      s += p[0]; q += p[1];
    }
  }
  __shared__ double sh[512];
  sh[threadIdx.x * 2] = s; sh[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (k0 == 0 && i < BC) {
    for (int k = 1; k < 8; ++k) { s += sh[(k * 32 + threadIdx.x) * 2]; q += sh[(k * 32 + threadIdx.x) * 2 + 1]; }
    const double mean = s / HW;
    double var = q / HW - mean * mean;
    if (var < 0) var = 0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__global__ void in_finalize_inplace_kernel(float* __restrict__ stats, int BC, int HW, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BC) return;
  const double mean = (double)stats[2 * i] / HW;
  double var = (double)stats[2 * i + 1] / HW - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ------------------------------------------------------------------ forward apply (+act, +residual, +halo)
// iterates the padded domain when halo_mode == REFLECT, else the interior
template <typename T>
__global__ __launch_bounds__(NTHR) void in_apply_kernel(DView x, const float* __restrict__ stats, int act, DView res, int has_res, DView y,
                                                       int halo_mode, int nblk) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(x.C);
  const int b = blockIdx.y;
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  const int DH = padded ? y.H + 2 * y.y0 : y.H, DW = padded ? y.W + 2 * y.x0 : y.W, total = DH * DW;
  const int per = (total + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(total, p0 + per);
  float mean[N], rstd[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    mean[e] = stats[((int64_t)b * x.C + L.cl * N + e) * 2];
    rstd[e] = stats[((int64_t)b * x.C + L.cl * N + e) * 2 + 1];
  }
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  const T* rp = reinterpret_cast<const T*>(res.ptr);
  T* yp = reinterpret_cast<T*>(y.ptr);
  int dy, dx; pixel_yx(p0 + L.rl, DW, dy, dx);
  for (int p = p0 + L.rl; p < p1; p += L.RL, pixel_step(L.RL, DW, dy, dx)) {
    int sy = dy, sx = dx;
    if (padded) { sy = reflect_idx(dy - y.y0, y.H); sx = reflect_idx(dx - y.x0, y.W); }
    float v[N];
    Chunk<T>::load(xp + x.pix(b, sy, sx) + L.cl * N, v);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = act_apply((v[e] - mean[e]) * rstd[e], act);
    if (has_res) {
      float r[N];
      Chunk<T>::load(rp + res.pix(b, sy, sx) + L.cl * N, r);
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] += r[e];
    }
    const int64_t o = padded ? y.pixp(b, dy, dx) : y.pix(b, dy, dx);
    Chunk<T>::store(yp + o + L.cl * N, v);
  }
}

// ------------------------------------------------------------------ backward
template <typename T>
__device__ __forceinline__ void in_bwd_g(const DView& x, const float* mean, const float* rstd, int act, const DView& gy, int fold,
                                         const DView& g2, int has_g2, int b, int yy, int xx, int cofs, int64_t offx, int64_t offg,
                                         int64_t offg2, float* g, float* xh) {
  constexpr int N = Chunk<T>::N;
  Chunk<T>::load(reinterpret_cast<const T*>(x.ptr) + offx + cofs, xh);
  load_folded<T>(gy, fold, b, yy, xx, offg, cofs, g);
  if (has_g2) {
    float t[N];
    Chunk<T>::load(reinterpret_cast<const T*>(g2.ptr) + offg2 + cofs, t);
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] += t[e];
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    xh[e] = (xh[e] - mean[e]) * rstd[e];
    if (act == GAN_ACT_RELU) g[e] = xh[e] > 0.f ? g[e] : 0.f;
    else if (act == GAN_ACT_LRELU) g[e] = xh[e] > 0.f ? g[e] : 0.2f * g[e];
  }
}

template <typename T>
__global__ __launch_bounds__(NTHR) void in_bwd_partial_kernel(DView x, const float* __restrict__ stats, int act, DView gy, int fold, DView g2,
                                                             int has_g2, int nch, float* __restrict__ ws) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, ch = blockIdx.x, HW = x.H * x.W;
  const int per = (HW + nch - 1) / nch, p0 = ch * per, p1 = min(HW, p0 + per);
  float mean[N], rstd[N], s1[N], s2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    mean[e] = stats[((int64_t)b * x.C + L.cl * N + e) * 2];
    rstd[e] = stats[((int64_t)b * x.C + L.cl * N + e) * 2 + 1];
    s1[e] = s2[e] = 0.f;
  }
  int yy, xx; pixel_yx(p0 + L.rl, x.W, yy, xx);
  Off ox(x, b, yy, xx, L.RL, x.W), og(gy, b, yy, xx, L.RL, x.W), o2(has_g2 ? g2 : x, b, yy, xx, L.RL, x.W);
  for (int p = p0 + L.rl; p < p1; p += L.RL) {
    float g[N], xh[N];
    in_bwd_g<T>(x, mean, rstd, act, gy, fold, g2, has_g2, b, yy, xx, L.cl * N, ox.o, og.o, o2.o, g, xh);
#pragma unroll
    for (int e = 0; e < N; ++e) { s1[e] += g[e]; s2[e] += g[e] * xh[e]; }
    const int wr = pixel_step(L.RL, x.W, yy, xx);
    ox.advance(wr); og.advance(wr); o2.advance(wr);
  }
  __shared__ float sh[NTHR * 16];
#pragma unroll
  for (int e = 0; e < N; ++e) { sh[(threadIdx.x * N + e) * 2] = s1[e]; sh[(threadIdx.x * N + e) * 2 + 1] = s2[e]; }
  __syncthreads();
  if (L.rl == 0) {
    for (int r = 1; r < L.RL; ++r)
#pragma unroll
      for (int e = 0; e < N; ++e) {
        s1[e] += sh[((r * L.CL + L.cl) * N + e) * 2];
        s2[e] += sh[((r * L.CL + L.cl) * N + e) * 2 + 1];
      }
    float* o = ws + ((int64_t)(b * nch + ch) * x.C + L.cl * N) * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
  }
}
// ws2[(b*C+c)*2] = mean(g), +1 = mean(g*xhat)
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const float* __restrict__ ws, int nch, int C, int BC, int HW, float* __restrict__ ws2) {
  const int i = blockIdx.x * 32 + (threadIdx.x & 31), k0 = threadIdx.x >> 5;
  double s = 0, q = 0;
  if (i < BC) {
    const int b = i / C, c = i - b * C;
    for (int k = k0; k < nch; k += 8) {
      const float* p = ws + ((int64_t)(b * nch + k) * C + c) * 2;
      s += p[0]; q += p[1];
    }
  }
  __shared__ double sh[512];
  sh[threadIdx.x * 2] = s; sh[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (k0 == 0 && i < BC) {
    for (int k = 1; k < 8; ++k) { s += sh[(k * 32 + threadIdx.x) * 2]; q += sh[(k * 32 + threadIdx.x) * 2 + 1]; }
    ws2[2 * i] = (float)(s / HW);
    ws2[2 * i + 1] = (float)(q / HW);
  }
}
template <typename T>
__global__ __launch_bounds__(NTHR) void in_bwd_apply_kernel(DView x, const float* __restrict__ stats, int act, DView gy, int fold, DView g2,
                                                           int has_g2, const float* __restrict__ ws2, DView dx, int nblk,
                                                           float* __restrict__ bias_part) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, HW = x.H * x.W;
  const int per = (HW + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  float mean[N], rstd[N], m1[N], m2[N], bs[N];
#pragma unroll
  for (int e = 0; e < N; ++e) bs[e] = 0.f;
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int64_t i = (int64_t)b * x.C + L.cl * N + e;
    mean[e] = stats[2 * i]; rstd[e] = stats[2 * i + 1]; m1[e] = ws2[2 * i]; m2[e] = ws2[2 * i + 1];
  }
  T* dp = reinterpret_cast<T*>(dx.ptr);
  int yy, xx; pixel_yx(p0 + L.rl, x.W, yy, xx);
  Off ox(x, b, yy, xx, L.RL, x.W), og(gy, b, yy, xx, L.RL, x.W), o2(has_g2 ? g2 : x, b, yy, xx, L.RL, x.W), od(dx, b, yy, xx, L.RL, x.W);
  for (int p = p0 + L.rl; p < p1; p += L.RL) {
    float g[N], xh[N];
    in_bwd_g<T>(x, mean, rstd, act, gy, fold, g2, has_g2, b, yy, xx, L.cl * N, ox.o, og.o, o2.o, g, xh);
#pragma unroll
    for (int e = 0; e < N; ++e) { g[e] = rstd[e] * (g[e] - m1[e] - xh[e] * m2[e]); bs[e] += g[e]; }
    Chunk<T>::store(dp + od.o + L.cl * N, g);
    const int wr = pixel_step(L.RL, x.W, yy, xx);
    ox.advance(wr); og.advance(wr); o2.advance(wr); od.advance(wr);
  }
  if (bias_part) {   // column sums of dx = gradient of the conv bias in front of this norm: partial per block
    __shared__ float sh[NTHR * 8];
#pragma unroll
    for (int e = 0; e < N; ++e) sh[threadIdx.x * N + e] = bs[e];
    __syncthreads();
    if (L.rl == 0) {
      for (int r = 1; r < L.RL; ++r)
#pragma unroll
        for (int e = 0; e < N; ++e) bs[e] += sh[(r * L.CL + L.cl) * N + e];
      float* o = bias_part + (int64_t)(b * nblk + blockIdx.x) * x.C + L.cl * N;
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = bs[e];
    }
  }
}
// out[seg][c] (+)= sum over this segment's blocks of part[blk][c]; gridDim.y segments (two-level reduction: many partials,
// few channels -> the first level spreads the partial list over gridDim.y blocks per 32 channels)
__global__ __launch_bounds__(256) void bias_part_finalize_kernel(const float* __restrict__ part, int nparts, int C, int N_real, float* __restrict__ out,
                                                                int out_stride, int accumulate) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), k0 = threadIdx.x >> 5;
  const int per = (nparts + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
  float s = 0.f;
  if (c < C)
    for (int k = p0 + k0; k < p1; k += 8) s += part[(int64_t)k * C + c];
  __shared__ float sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (k0 == 0 && c < N_real) {
    for (int k = 1; k < 8; ++k) s += sh[k * 32 + (threadIdx.x & 31)];
    float* o = out + (int64_t)blockIdx.y * out_stride + c;
    *o = accumulate ? *o + s : s;
  }
}

// the same sum for many layers in ONE launch: block -> (descriptor, 32-channel group) through first_block
__global__ __launch_bounds__(1024) void bias_finalize_batch_kernel(const gan_bias_part_desc* __restrict__ descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const gan_bias_part_desc d = descs[lo];
  const int c = ((int)blockIdx.x - d.first_block) * 32 + (threadIdx.x & 31), k0 = threadIdx.x >> 5;   // 32 partial rows in flight per channel
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < d.C) {
    const float* p = d.part + c;
    int k = k0;
    for (; k + 96 < d.nparts; k += 128) {
      s0 += p[(int64_t)k * d.C]; s1 += p[(int64_t)(k + 32) * d.C]; s2 += p[(int64_t)(k + 64) * d.C]; s3 += p[(int64_t)(k + 96) * d.C];
    }
    for (; k < d.nparts; k += 32) s0 += p[(int64_t)k * d.C];
  }
  __shared__ float sh[1024];
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (k0 == 0 && c < d.N_real) {
    float s = 0.f;
    for (int k = 0; k < 32; ++k) s += sh[k * 32 + (threadIdx.x & 31)];
    d.grad[c] = d.accumulate ? d.grad[c] + s : s;
  }
}

// out = a + fold(b)   /   dx = (fold(g) + g2) * act'(y)
template <typename T>
__global__ __launch_bounds__(NTHR) void fold_add_kernel(DView a, int has_a, DView g, int fold, DView y, int act, DView out, int nblk) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(out.C);
  const int b = blockIdx.y, HW = out.H * out.W;
  const int per = (HW + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  T* op = reinterpret_cast<T*>(out.ptr);
  int yy, xx; pixel_yx(p0 + L.rl, out.W, yy, xx);
  Off og(g, b, yy, xx, L.RL, out.W), oa(has_a ? a : g, b, yy, xx, L.RL, out.W), oy(act != GAN_ACT_NONE ? y : g, b, yy, xx, L.RL, out.W),
      oo(out, b, yy, xx, L.RL, out.W);
  for (int p = p0 + L.rl; p < p1; p += L.RL) {
    float v[N];
    load_folded<T>(g, fold, b, yy, xx, og.o, L.cl * N, v);
    if (has_a) {
      float t[N];
      Chunk<T>::load(reinterpret_cast<const T*>(a.ptr) + oa.o + L.cl * N, t);
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] += t[e];
    }
    if (act != GAN_ACT_NONE) {
      float t[N];
      Chunk<T>::load(reinterpret_cast<const T*>(y.ptr) + oy.o + L.cl * N, t);
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] *= act_grad_from_out(t[e], act);
    }
    Chunk<T>::store(op + oo.o + L.cl * N, v);
    const int wr = pixel_step(L.RL, out.W, yy, xx);
    og.advance(wr); oa.advance(wr); oy.advance(wr); oo.advance(wr);
  }
}

// ------------------------------------------------------------------ layout boundary
template <typename T>
__global__ void nchw_to_view_kernel(const float* __restrict__ src, int C, DView dst, int halo_mode) {
  constexpr int N = Chunk<T>::N;
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  const int DH = padded ? dst.H + 2 * dst.y0 : dst.H, DW = padded ? dst.W + 2 * dst.x0 : dst.W;
  const int nck = dst.C / N;
  const int64_t total = (int64_t)dst.B * DH * DW * nck;
  T* dp = reinterpret_cast<T*>(dst.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % nck);
    int64_t r = i / nck;
    const int dx = (int)(r % DW); r /= DW;
    const int dy = (int)(r % DH);
    const int b = (int)(r / DH);
    int sy = dy, sx = dx;
    if (padded) { sy = reflect_idx(dy - dst.y0, dst.H); sx = reflect_idx(dx - dst.x0, dst.W); }
    float v[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int c = ck * N + e;
      v[e] = c < C ? src[(((int64_t)b * C + c) * dst.H + sy) * dst.W + sx] : 0.f;
    }
    const int64_t o = padded ? dst.pixp(b, dy, dx) : dst.pix(b, dy, dx);
    Chunk<T>::store(dp + o + ck * N, v);
  }
}
template <typename T>
__global__ void view_to_nchw_kernel(DView src, int C, float* __restrict__ dst) {
  const int64_t total = (int64_t)src.B * C * src.H * src.W;
  const T* sp = reinterpret_cast<const T*>(src.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % src.W);
    int64_t r = i / src.W;
    const int y = (int)(r % src.H); r /= src.H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    dst[i] = ld1<T>(sp + src.pix(b, y, x) + c);
  }
}
template <typename T>
__global__ void view_copy_kernel(DView src, DView dst, int halo_mode) {
  constexpr int N = Chunk<T>::N;
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  const int DH = padded ? dst.H + 2 * dst.y0 : dst.H, DW = padded ? dst.W + 2 * dst.x0 : dst.W;
  const int nck = dst.C / N;
  const int64_t total = (int64_t)dst.B * DH * DW * nck;
  const T* sp = reinterpret_cast<const T*>(src.ptr);
  T* dp = reinterpret_cast<T*>(dst.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % nck);
    int64_t r = i / nck;
    const int dx = (int)(r % DW); r /= DW;
    const int dy = (int)(r % DH);
    const int b = (int)(r / DH);
    int sy = dy, sx = dx;
    if (padded) { sy = reflect_idx(dy - dst.y0, dst.H); sx = reflect_idx(dx - dst.x0, dst.W); }
    float v[N];
    Chunk<T>::load(sp + src.pix(b, sy, sx) + ck * N, v);
    const int64_t o = padded ? dst.pixp(b, dy, dx) : dst.pix(b, dy, dx);
    Chunk<T>::store(dp + o + ck * N, v);
  }
}

int check_lanes(const gan_view* v, const char* what) {
  const int epc = v->dtype == GAN_F32 ? 4 : 8;
  const int cl = v->C / epc;
  if (v->C % epc != 0 || cl > NTHR || (cl & (cl - 1)) != 0) return gan_set_error(-1, "%s: C=%d unsupported (C/%d must be a power of two <= 256)", what, v->C, epc);
  return 0;
}
// row-chunks per image for the statistics passes: ~2048 16-byte loads per block, at most MAXCH (workspace bound)
int work_per_block() {   // 16-byte loads per block; GAN_NORM_WORK overrides (tuning aid)
  static int w = 0;
  if (!w) { const char* e = getenv("GAN_NORM_WORK"); w = e ? atoi(e) : 2048; if (w < 256) w = 256; }
  return w;
}
int nchunks_for(int HW, int cl) {
  const int W = work_per_block();
  int64_t n = ((int64_t)HW * cl + W - 1) / W;
  if (n < 1) n = 1;
  if (n > MAXCH) n = MAXCH;
  return (int)n;
}
// blocks per image for the apply passes (no workspace bound)
int nblocks_for(int pixels, int cl) {
  const int W = work_per_block();
  int64_t n = ((int64_t)pixels * cl + W - 1) / W;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}
int lanes_of(const gan_view* v) { return v->C / (v->dtype == GAN_F32 ? 4 : 8); }
int fold_ok(const gan_view* g, int fold) {
  if (!fold) return 0;
  if (g->y0 < 1 || g->x0 < 1 || g->H < 2 * g->y0 + 2 || g->W < 2 * g->x0 + 2 || g->y0 + g->H + g->y0 > g->Hp || g->x0 + g->W + g->x0 > g->Wp)
    return gan_set_error(-1, "fold: view must carry a symmetric halo of y0/x0 pixels and H >= 2*pad+2");
  return 0;
}

}  // namespace

#define VCHK(v, name) do { if (gan_check_view(v, name)) return -1; } while (0)
#define SAME_SHAPE(a, b, what) GAN_CHECK((a)->B == (b)->B && (a)->H == (b)->H && (a)->W == (b)->W && (a)->C == (b)->C && (a)->dtype == (b)->dtype, what ": shape/dtype mismatch")

// ws: fp32, >= B*MAXCH*C*2 floats
extern "C" int gan_in_stats(const gan_view* x, float eps, float* stats, float* ws, void* stream) {
  VCHK(x, "in_stats.x");
  if (check_lanes(x, "in_stats")) return -1;
  GAN_CHECK(stats && ws, "in_stats: null pointer");
  const int HW = x->H * x->W, nch = nchunks_for(HW, lanes_of(x)), BC = x->B * x->C;
  DView dx = to_dview(x);
  hipStream_t s = (hipStream_t)stream;
  GAN_DISPATCH_DTYPE(x->dtype, hipLaunchKernelGGL((in_partial_kernel<T>), dim3(nch, x->B), dim3(NTHR), 0, s, dx, nch, ws);)
  hipLaunchKernelGGL(in_finalize_kernel, dim3((BC + 31) / 32), dim3(256), 0, s, ws, nch, x->C, BC, HW, eps, stats);
  GAN_LAUNCH_CHECK();
  return 0;
}

// (mean, rstd) from per-tile partials written by a convolution epilogue (gan_conv_desc.stats): parts = fp32 [B][nparts][C][2]
extern "C" int gan_in_stats_from_parts(const float* parts, int nparts, int B, int C, int HW, float eps, float* stats, void* stream) {
  GAN_CHECK(parts && stats && nparts > 0 && B > 0 && C > 0 && HW > 0, "in_stats_from_parts: bad arguments");
  const int BC = B * C;
  hipLaunchKernelGGL(in_finalize_kernel, dim3((BC + 31) / 32), dim3(256), 0, (hipStream_t)stream, parts, nparts, C, BC, HW, eps, stats);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_in_finalize(float* stats, int BC, int HW, float eps, void* stream) {
  GAN_CHECK(stats && BC > 0 && HW > 0, "in_finalize: bad arguments");
  hipLaunchKernelGGL(in_finalize_inplace_kernel, dim3((BC + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, BC, HW, eps);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_in_apply(const gan_view* x, const float* stats, int act, const gan_view* residual, const gan_view* y, int halo_mode,
                            void* stream) {
  VCHK(x, "in_apply.x"); VCHK(y, "in_apply.y");
  if (check_lanes(x, "in_apply")) return -1;
  SAME_SHAPE(x, y, "in_apply(x,y)");
  if (residual) { VCHK(residual, "in_apply.residual"); SAME_SHAPE(x, residual, "in_apply(x,residual)"); }
  GAN_CHECK(stats, "in_apply: null stats");
  if (halo_mode == GAN_HALO_REFLECT)
    GAN_CHECK(y->y0 < y->H && y->x0 < y->W && 2 * y->y0 + y->H <= y->Hp && 2 * y->x0 + y->W <= y->Wp, "in_apply: reflect halo does not fit");
  const int DH = halo_mode == GAN_HALO_REFLECT ? y->H + 2 * y->y0 : y->H, DW = halo_mode == GAN_HALO_REFLECT ? y->W + 2 * y->x0 : y->W;
  const int nblk = nblocks_for(DH * DW, lanes_of(x));
  DView dx = to_dview(x), dy = to_dview(y), dr = residual ? to_dview(residual) : null_dview();
  GAN_DISPATCH_DTYPE(x->dtype, hipLaunchKernelGGL((in_apply_kernel<T>), dim3(nblk, x->B), dim3(NTHR), 0, (hipStream_t)stream, dx, stats, act,
                                                  dr, residual ? 1 : 0, dy, halo_mode, nblk);)
  GAN_LAUNCH_CHECK();
  return 0;
}

// ws: fp32, >= B*MAXCH*C*2 + B*C*2 floats
static int in_bwd_impl(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2, const gan_view* dx,
                       float* ws, float* bias_grad, int bias_n, int bias_acc, void* stream, float* bias_part = nullptr);

extern "C" int gan_in_bwd(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
                          const gan_view* dx, float* ws, void* stream) {
  return in_bwd_impl(x, stats, act, gy, fold, g2, dx, ws, nullptr, 0, 0, stream);
}

// same, and additionally bias_grad[n] (+)= sum over pixels of dx[...,n], n < bias_n (the conv bias in front of the norm).
// ws: fp32 >= B*96*C*2 + B*C*2 + (B*1024 + 32)*C floats
extern "C" int gan_in_bwd_bias(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
                               const gan_view* dx, float* ws, float* bias_grad, int bias_n, int bias_accumulate, void* stream) {
  GAN_CHECK(bias_grad && bias_n > 0 && bias_n <= x->C, "in_bwd_bias: bad bias arguments");
  return in_bwd_impl(x, stats, act, gy, fold, g2, dx, ws, bias_grad, bias_n, bias_accumulate, stream);
}

static int in_bwd_impl(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2, const gan_view* dx,
                       float* ws, float* bias_grad, int bias_n, int bias_acc, void* stream, float* bias_part) {
  VCHK(x, "in_bwd.x"); VCHK(gy, "in_bwd.gy"); VCHK(dx, "in_bwd.dx");
  if (check_lanes(x, "in_bwd")) return -1;
  SAME_SHAPE(x, gy, "in_bwd(x,gy)"); SAME_SHAPE(x, dx, "in_bwd(x,dx)");
  if (g2) { VCHK(g2, "in_bwd.g2"); SAME_SHAPE(x, g2, "in_bwd(x,g2)"); }
  if (fold_ok(gy, fold)) return -1;
  GAN_CHECK(stats && ws, "in_bwd: null pointer");
  GAN_CHECK(act == GAN_ACT_NONE || act == GAN_ACT_RELU || act == GAN_ACT_LRELU, "in_bwd: unsupported activation %d", act);
  const int HW = x->H * x->W, nch = nchunks_for(HW, lanes_of(x)), BC = x->B * x->C;
  float* ws2 = ws + (int64_t)x->B * MAXCH * x->C * 2;
  DView vx = to_dview(x), vg = to_dview(gy), v2 = g2 ? to_dview(g2) : null_dview(), vd = to_dview(dx);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = nblocks_for(HW, lanes_of(x));
  GAN_DISPATCH_DTYPE(x->dtype,
    hipLaunchKernelGGL((in_bwd_partial_kernel<T>), dim3(nch, x->B), dim3(NTHR), 0, s, vx, stats, act, vg, fold, v2, g2 ? 1 : 0, nch, ws);
    hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3((BC + 31) / 32), dim3(256), 0, s, ws, nch, x->C, BC, HW, ws2);
    hipLaunchKernelGGL((in_bwd_apply_kernel<T>), dim3(nblk, x->B), dim3(NTHR), 0, s, vx, stats, act, vg, fold, v2, g2 ? 1 : 0, ws2, vd, nblk,
                       bias_part ? bias_part : (bias_grad ? ws2 + (int64_t)BC * 2 : nullptr));)
  if (bias_grad && !bias_part) {
    float* part = ws2 + (int64_t)BC * 2;
    const int nparts = x->B * nblk;
    if (nparts > 64) {   // two levels: 32 segments -> scratch behind the partials, then the final 32 -> grad
      float* seg = part + (int64_t)nparts * x->C;
      hipLaunchKernelGGL(bias_part_finalize_kernel, dim3((x->C + 31) / 32, 32), dim3(256), 0, s, part, nparts, x->C, x->C, seg, x->C, 0);
      hipLaunchKernelGGL(bias_part_finalize_kernel, dim3((x->C + 31) / 32, 1), dim3(256), 0, s, seg, 32, x->C, bias_n, bias_grad, 0, bias_acc);
    } else {
      hipLaunchKernelGGL(bias_part_finalize_kernel, dim3((x->C + 31) / 32, 1), dim3(256), 0, s, part, nparts, x->C, bias_n, bias_grad, 0, bias_acc);
    }
  }
  GAN_LAUNCH_CHECK();
  return 0;
}

// The bias-gradient partials of gan_in_bwd_bias go to a caller-owned buffer ([gan_in_bwd_bias_parts(x)][x->C] floats) and are summed
// later, for all layers of a backward pass at once, by gan_bias_finalize_batch: two tiny launches per layer leave the backward chain.
extern "C" int gan_in_bwd_bias_parts(const gan_view* x) {
  if (gan_check_view(x, "in_bwd_bias_parts.x")) return -1;
  return x->B * nblocks_for(x->H * x->W, lanes_of(x));
}

extern "C" int gan_in_bwd_bias_deferred(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
                                        const gan_view* dx, float* ws, float* bias_part, void* stream) {
  GAN_CHECK(bias_part, "in_bwd_bias_deferred: null partial buffer");
  return in_bwd_impl(x, stats, act, gy, fold, g2, dx, ws, nullptr, 0, 0, stream, bias_part);
}

extern "C" int gan_bias_finalize_batch(const gan_bias_part_desc* descs, int n, int total_blocks, void* stream) {
  GAN_CHECK(descs && n > 0 && total_blocks > 0, "bias_finalize_batch: empty batch");
  hipLaunchKernelGGL(bias_finalize_batch_kernel, dim3(total_blocks), dim3(1024), 0, (hipStream_t)stream, descs, n);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_fold_add(const gan_view* a, const gan_view* b, int fold, const gan_view* out, void* stream) {
  VCHK(b, "fold_add.b"); VCHK(out, "fold_add.out");
  if (check_lanes(out, "fold_add")) return -1;
  SAME_SHAPE(b, out, "fold_add(b,out)");
  if (a) { VCHK(a, "fold_add.a"); SAME_SHAPE(a, out, "fold_add(a,out)"); }
  if (fold_ok(b, fold)) return -1;
  const int HW = out->H * out->W;
  const int nblk = nblocks_for(HW, lanes_of(out));
  DView va = a ? to_dview(a) : null_dview(), vb = to_dview(b), vo = to_dview(out);
  GAN_DISPATCH_DTYPE(out->dtype, hipLaunchKernelGGL((fold_add_kernel<T>), dim3(nblk, out->B), dim3(NTHR), 0, (hipStream_t)stream, va, a ? 1 : 0,
                                                    vb, fold, null_dview(), GAN_ACT_NONE, vo, nblk);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_act_bwd(const gan_view* y, int act, const gan_view* g, int fold, const gan_view* g2, const gan_view* dx, void* stream) {
  VCHK(y, "act_bwd.y"); VCHK(g, "act_bwd.g"); VCHK(dx, "act_bwd.dx");
  if (check_lanes(dx, "act_bwd")) return -1;
  SAME_SHAPE(y, g, "act_bwd(y,g)"); SAME_SHAPE(y, dx, "act_bwd(y,dx)");
  if (g2) { VCHK(g2, "act_bwd.g2"); SAME_SHAPE(y, g2, "act_bwd(y,g2)"); }
  if (fold_ok(g, fold)) return -1;
  const int HW = dx->H * dx->W;
  const int nblk = nblocks_for(HW, lanes_of(dx));
  DView vy = to_dview(y), vg = to_dview(g), v2 = g2 ? to_dview(g2) : null_dview(), vd = to_dview(dx);
  GAN_DISPATCH_DTYPE(dx->dtype, hipLaunchKernelGGL((fold_add_kernel<T>), dim3(nblk, dx->B), dim3(NTHR), 0, (hipStream_t)stream, v2, g2 ? 1 : 0,
                                                   vg, fold, vy, act, vd, nblk);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_nchw_to_view(const float* src, int C, const gan_view* dst, int halo_mode, void* stream) {
  VCHK(dst, "nchw_to_view.dst");
  GAN_CHECK(src && C > 0 && C <= dst->C, "nchw_to_view: bad C=%d", C);
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  if (padded) GAN_CHECK(dst->y0 < dst->H && dst->x0 < dst->W && 2 * dst->y0 + dst->H <= dst->Hp && 2 * dst->x0 + dst->W <= dst->Wp, "nchw_to_view: reflect halo does not fit");
  const int epc = dst->dtype == GAN_F32 ? 4 : 8;
  const int64_t total = (int64_t)dst->B * (padded ? dst->H + 2 * dst->y0 : dst->H) * (padded ? dst->W + 2 * dst->x0 : dst->W) * (dst->C / epc);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView dv = to_dview(dst);
  GAN_DISPATCH_DTYPE(dst->dtype, hipLaunchKernelGGL((nchw_to_view_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, C, dv, halo_mode);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_view_to_nchw(const gan_view* src, int C, float* dst, void* stream) {
  VCHK(src, "view_to_nchw.src");
  GAN_CHECK(dst && C > 0 && C <= src->C, "view_to_nchw: bad C=%d", C);
  const int64_t total = (int64_t)src->B * C * src->H * src->W;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView sv = to_dview(src);
  GAN_DISPATCH_DTYPE(src->dtype, hipLaunchKernelGGL((view_to_nchw_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, sv, C, dst);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_view_copy(const gan_view* src, const gan_view* dst, int halo_mode, void* stream) {
  VCHK(src, "view_copy.src"); VCHK(dst, "view_copy.dst");
  SAME_SHAPE(src, dst, "view_copy");
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  if (padded) GAN_CHECK(dst->y0 < dst->H && dst->x0 < dst->W && 2 * dst->y0 + dst->H <= dst->Hp && 2 * dst->x0 + dst->W <= dst->Wp, "view_copy: reflect halo does not fit");
  const int epc = dst->dtype == GAN_F32 ? 4 : 8;
  const int64_t total = (int64_t)dst->B * (padded ? dst->H + 2 * dst->y0 : dst->H) * (padded ? dst->W + 2 * dst->x0 : dst->W) * (dst->C / epc);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView sv = to_dview(src), dv = to_dview(dst);
  GAN_DISPATCH_DTYPE(dst->dtype, hipLaunchKernelGGL((view_copy_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, sv, dv, halo_mode);)
  GAN_LAUNCH_CHECK();
  return 0;
}
