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
#include <type_traits>
#include "common.h"

namespace {

constexpr int NTHR = 256;
constexpr int MAXPARTS = 16;  // partials per image that the apply passes sum themselves (no finalize launch)
// UNR (template parameter of the streaming kernels): pixels in flight per thread.  The HBM-bound passes keep UNR x (operands) x 16 B per
// thread outstanding, so that they still stream when an MFMA-bound kernel on another stream leaves them only one block per CU.
constexpr int MAXCH = 96;  // max row-chunks per image for the two-stage statistics (gan_in_stats: partial + finalize launch)

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

// ------------------------------------------------------------------ forward statistics
// ws[((b*nch + ch)*C + c)*2 + {0,1}] = partial (sum, sum of squares)
template <typename T, int UNR>
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
  for (int p = p0 + L.rl; p < p1; p += 2 * UNR * L.RL) {
    Raw<T> xr[2 * UNR];
#pragma unroll
    for (int u = 0; u < 2 * UNR; ++u) {
      if (p + u * L.RL < p1) xr[u] = ldraw(xp + ox.o + L.cl * N);
      ox.advance(pixel_step(L.RL, x.W, y, xx));
    }
#pragma unroll
    for (int u = 0; u < 2 * UNR; ++u) {
      if (p + u * L.RL < p1) {
        float v[N];
        cvtraw<T>(xr[u], v);
#pragma unroll
        for (int e = 0; e < N; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
      }
    }
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

// ------------------------------------------------------------------ per-image sums from partials (fused "finalize")
// A consumer block turns the per-(image, chunk) partial pairs [B][nparts][C][2] into per-channel values itself: thread t adds the
// nparts pairs of channels t, t + NTHR, ... in fp64 and in chunk order (deterministic; coalesced, every pair read once per block),
// `fin` maps the two totals to two floats, and the block's lanes pick their N channels up from LDS.  With <= MAXPARTS partials per
// image this replaces the separate 5 us finalize launch (+ its launch boundary) in front of every apply pass.
constexpr int SUMS_MAXC = 1024;   // channels the LDS exchange holds (2 floats each)
template <int N, typename Fin>
__device__ __forceinline__ void block_sums(const float* __restrict__ parts, int nparts, int C, int b, int c0, float* sh, float* a, float* bb, Fin fin) {
  for (int c = threadIdx.x; c < C; c += NTHR) {
    const float2* p = reinterpret_cast<const float2*>(parts + ((int64_t)b * nparts * C + c) * 2);
    double s = 0.0, q = 0.0;
#pragma unroll 8
    for (int k = 0; k < nparts; ++k) { const float2 t = p[(int64_t)k * C]; s += t.x; q += t.y; }
    float2 r;
    fin(c, s, q, r.x, r.y);
    reinterpret_cast<float2*>(sh)[c] = r;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < N; ++e) { a[e] = sh[2 * (c0 + e)]; bb[e] = sh[2 * (c0 + e) + 1]; }
  __syncthreads();   // sh is reused by the caller
}

// ------------------------------------------------------------------ forward apply (+act, +residual, +halo)
// iterates the padded domain when halo_mode == REFLECT, else the interior.  Statistics: `stats` (mean, rstd) as given, or -- when
// `parts` is set -- computed here from the partial sums (sum, sum of squares) and written to `stats` by the image's first block
// (the backward reads them).
// y8 (optional): an e4m3 copy of y with the same geometry (unit scale, clamped to +-448) written by the same pass -- the operand of the
// next convolution on the fp8 path, at the price of one extra 8-byte store per chunk instead of a separate quantisation pass.
template <typename T, int UNR>
__global__ __launch_bounds__(NTHR) void in_apply_kernel(DView x, float* __restrict__ stats, const float* __restrict__ parts, int nparts, float eps,
                                                       int act, DView res, int has_res, DView y, int halo_mode, int nblk, uint8_t* __restrict__ y8 = nullptr) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, cofs = L.cl * N;
  const bool padded = halo_mode == GAN_HALO_REFLECT;
  const int DH = padded ? y.H + 2 * y.y0 : y.H, DW = padded ? y.W + 2 * y.x0 : y.W, total = DH * DW;
  const int per = (total + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(total, p0 + per);
  float mean[N], rstd[N];
  if (parts) {
    __shared__ float sh[2 * SUMS_MAXC];
    const double HW = (double)(x.H * x.W);
    const bool first = blockIdx.x == 0;
    block_sums<N>(parts, nparts, x.C, b, cofs, sh, mean, rstd, [&](int c, double s, double q, float& m_out, float& r_out) {
      const double m = s / HW;
      double var = q / HW - m * m;
      if (var < 0) var = 0;
      m_out = (float)m;
      r_out = (float)(1.0 / sqrt(var + (double)eps));
      if (first) { stats[((int64_t)b * x.C + c) * 2] = m_out; stats[((int64_t)b * x.C + c) * 2 + 1] = r_out; }
    });
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      mean[e] = stats[((int64_t)b * x.C + cofs + e) * 2];
      rstd[e] = stats[((int64_t)b * x.C + cofs + e) * 2 + 1];
    }
  }
  const T* xp = reinterpret_cast<const T*>(x.ptr);
  const T* rp = reinterpret_cast<const T*>(res.ptr);
  T* yp = reinterpret_cast<T*>(y.ptr);
  int dy, dx; pixel_yx(p0 + L.rl, DW, dy, dx);
  for (int p = p0 + L.rl; p < p1; p += UNR * L.RL) {
    Raw<T> xr[UNR], rr[UNR];
    int64_t oo[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * L.RL < p1) {
        int sy = dy, sx = dx;
        if (padded) { sy = reflect_idx(dy - y.y0, y.H); sx = reflect_idx(dx - y.x0, y.W); }
        xr[u] = ldraw(xp + x.pix(b, sy, sx) + cofs);
        if (has_res) rr[u] = ldraw(rp + res.pix(b, sy, sx) + cofs);
        oo[u] = padded ? y.pixp(b, dy, dx) : y.pix(b, dy, dx);
      }
      pixel_step(L.RL, DW, dy, dx);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * L.RL < p1) {
        float v[N];
        cvtraw<T>(xr[u], v);
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = act_apply((v[e] - mean[e]) * rstd[e], act);
        if (has_res) {
          float r[N];
          cvtraw<T>(rr[u], r);
#pragma unroll
          for (int e = 0; e < N; ++e) v[e] += r[e];
        }
        Chunk<T>::store(yp + oo[u] + cofs, v);
        if (y8) {
          uint32_t q[N / 4];
#pragma unroll
          for (int k = 0; k < N / 4; ++k) q[k] = f2e4m3x4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
          if (N == 8) *reinterpret_cast<u32x2_t*>(y8 + oo[u] + cofs) = u32x2_t{q[0], q[N / 4 - 1]};
          else *reinterpret_cast<uint32_t*>(y8 + oo[u] + cofs) = q[0];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ backward
// The backward passes run beside the weight-gradient kernels of the second stream, whose persistent blocks leave a CU 80 registers
// per SIMD lane and 16 KB of LDS: these kernels are written to fit into that (one 256-thread block per CU then streams with
// UNR x 2 x 16 B per thread in flight).  Hence: 32-bit offsets relative to the image, statistics folded into per-channel
// coefficients (no xhat in the loop), activation / fold as template parameters, 8 KB of LDS.
struct Off32 {   // element offset of the walk's current pixel inside ONE image of a view
  int o, dstep, dwrap;
  __device__ __forceinline__ Off32(const DView& v, int y, int x, int step, int W)
      : o(((y + v.y0) * v.Wp + x + v.x0) * v.C), dstep(step * v.C), dwrap((v.Wp - W) * v.C) {}
  __device__ __forceinline__ void advance(int wraps) { o += dstep + wraps * dwrap; }
};
template <typename T> __device__ __forceinline__ const T* image_ptr(const DView& v, int b, int cofs) {
  return reinterpret_cast<const T*>(v.ptr) + (int64_t)b * v.Hp * v.Wp * v.C + cofs;
}

// adds the reflect pre-images of logical pixel (y,x) (pad = g.y0) to v: the fold of a padded-domain gradient.  gp = image_ptr of g.
// Nearly every pixel has none: one compare pair, no loads.
template <typename T>
__device__ __forceinline__ void fold_extra(const DView& g, const T* gp, int y, int x, float* v) {
  constexpr int N = Chunk<T>::N;
  const int py = g.y0, px = g.x0;
  int y2 = -1, x2 = -1;
  if (y >= 1 && y <= py) y2 = py - y;
  else if (y >= g.H - 1 - py && y <= g.H - 2) y2 = 2 * (g.H - 1) - y + py;
  if (x >= 1 && x <= px) x2 = px - x;
  else if (x >= g.W - 1 - px && x <= g.W - 2) x2 = 2 * (g.W - 1) - x + px;
  if ((y2 & x2) < 0 && (y2 | x2) < 0) return;   // both -1
  float t[N];
  if (x2 >= 0) {
    Chunk<T>::load(gp + ((y + py) * g.Wp + x2) * g.C, t);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += t[e];
  }
  if (y2 >= 0) {
    Chunk<T>::load(gp + (y2 * g.Wp + x + px) * g.C, t);
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] += t[e];
    if (x2 >= 0) {
      Chunk<T>::load(gp + (y2 * g.Wp + x2) * g.C, t);
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] += t[e];
    }
  }
}
template <int ACT> __device__ __forceinline__ float act_mask(float g, float x, float mean) {   // g * act'(xhat), sign(xhat) = sign(x - mean)
  if (ACT == GAN_ACT_RELU) return x > mean ? g : 0.f;
  if (ACT == GAN_ACT_LRELU) return x > mean ? g : 0.2f * g;
  return g;
}
// sum of v[e] over the row lanes (threads with equal cl) of the block, through 8 KB of LDS; valid in the rl == 0 threads
template <int N, typename LanesT>
__device__ __forceinline__ void row_lane_sum(const LanesT& L, float* sh, float* v) {
  __syncthreads();
#pragma unroll
  for (int e = 0; e < N; ++e) sh[threadIdx.x * N + e] = v[e];
  __syncthreads();
  if (L.rl == 0)
    for (int r = 1; r < L.RL; ++r)
#pragma unroll
      for (int e = 0; e < N; ++e) v[e] += sh[(r * L.CL + L.cl) * N + e];
}

// element pairs (2j, 2j+1) of a raw chunk, read and written in place: the passes below walk a chunk pair by pair so that only the
// pair's coefficients and two temporaries are live beside the UNR raw chunks
template <typename T> struct Pairs;
template <> struct Pairs<bf16_t> {
  static constexpr int NP = 4;
  static __device__ __forceinline__ void get(const u32x4_t& r, int j, float& a, float& b) {
    a = __builtin_bit_cast(float, r[j] << 16); b = __builtin_bit_cast(float, r[j] & 0xffff0000u);
  }
  static __device__ __forceinline__ void set(u32x4_t& r, int j, float a, float b) { r[j] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }
};
template <> struct Pairs<float> {
  static constexpr int NP = 2;
  static __device__ __forceinline__ void get(const f32x4_t& r, int j, float& a, float& b) { a = r[2 * j]; b = r[2 * j + 1]; }
  static __device__ __forceinline__ void set(f32x4_t& r, int j, float a, float b) { r[2 * j] = a; r[2 * j + 1] = b; }
};
// fold_extra on a raw chunk: the pre-images are added pair by pair (fp32 adds; the sum goes back to the chunk's storage type, i.e. a
// border pixel's folded gradient is rounded to bf16 once more in bf16 mode -- 6 % of the pixels, 2^-9 relative)
template <typename T>
__device__ __forceinline__ void raw_add(Raw<T>& r, const Raw<T>& t) {
#pragma unroll
  for (int j = 0; j < Pairs<T>::NP; ++j) {
    float a, b, c, d;
    Pairs<T>::get(r, j, a, b);
    Pairs<T>::get(t, j, c, d);
    Pairs<T>::set(r, j, a + c, b + d);
  }
}
template <typename T>
__device__ __forceinline__ void fold_extra_raw(const DView& g, const T* gp, int y, int x, Raw<T>& r) {
  const int py = g.y0, px = g.x0;
  int y2 = -1, x2 = -1;
  if (y >= 1 && y <= py) y2 = py - y;
  else if (y >= g.H - 1 - py && y <= g.H - 2) y2 = 2 * (g.H - 1) - y + py;
  if (x >= 1 && x <= px) x2 = px - x;
  else if (x >= g.W - 1 - px && x <= g.W - 2) x2 = 2 * (g.W - 1) - x + px;
  if (x2 >= 0) raw_add<T>(r, ldraw(gp + ((y + py) * g.Wp + x2) * g.C));
  if (y2 >= 0) {
    raw_add<T>(r, ldraw(gp + (y2 * g.Wp + x + px) * g.C));
    if (x2 >= 0) raw_add<T>(r, ldraw(gp + (y2 * g.Wp + x2) * g.C));
  }
}
// does logical pixel (y,x) of a view with halo (py,px) collect reflect pre-images (see fold_extra)?
__device__ __forceinline__ bool fold_needed(const DView& g, int y, int x) {
  const int py = g.y0, px = g.x0;
  return (y >= 1 && y <= py) || (y >= g.H - 1 - py && y <= g.H - 2) || (x >= 1 && x <= px) || (x >= g.W - 1 - px && x <= g.W - 2);
}

// ws[((b*nch + ch)*C + c)*2 + {0,1}] = partial (sum g', sum g'*x) of chunk ch over its pixels, g' = act'-masked (folded) gradient and
// x the RAW norm input: the apply pass turns the totals into mean(g') and mean(g'*xhat) = rstd * (S2 - mean * S1) / HW in fp64.
// Registers: UNR raw chunk pairs + 2N accumulators; the per-channel means (activation masks) are re-read from LDS per group.
template <typename T, int UNR, int ACT, bool FOLD>
__global__ __launch_bounds__(NTHR, 6) void in_bwd_partial_kernel(DView x, const float* __restrict__ stats, DView gy, int nch, float* __restrict__ ws) {
  constexpr int N = Chunk<T>::N, NP = Pairs<T>::NP;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, ch = blockIdx.x, HW = x.H * x.W, cofs = L.cl * N;
  const int per = (HW + nch - 1) / nch, p0 = ch * per, p1 = min(HW, p0 + per);
  __shared__ float sh[NTHR * 8];
  float s1[N], s2[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s1[e] = s2[e] = 0.f;
  if (ACT != GAN_ACT_NONE) {
    for (int c = threadIdx.x; c < x.C; c += NTHR) sh[c] = stats[((int64_t)b * x.C + c) * 2];
    __syncthreads();
  }
  const float* mean = sh + cofs;
  const T *xp = image_ptr<T>(x, b, cofs), *gp = image_ptr<T>(gy, b, cofs);
  int yy, xx; pixel_yx(p0 + L.rl, x.W, yy, xx);
  Off32 ox(x, yy, xx, L.RL, x.W), og(gy, yy, xx, L.RL, x.W);
  for (int p = p0 + L.rl; p < p1; p += UNR * L.RL) {
    Raw<T> xr[UNR], gr[UNR];
    unsigned fmask = 0;              // bit u: pixel u collects reflect pre-images (border pixels only)
    asm volatile("" ::: "memory");   // the LDS reads below stay inside the loop (hoisted, they would cost N registers)
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * L.RL < p1) {
        xr[u] = ldraw(xp + ox.o); gr[u] = ldraw(gp + og.o);
        if (FOLD && fold_needed(gy, yy, xx)) fmask |= 1u << u;
      } else { xr[u] = Raw<T>{}; gr[u] = Raw<T>{}; }          // zero gradient: contributes nothing
      const int wr = pixel_step(L.RL, x.W, yy, xx);
      ox.advance(wr); og.advance(wr);
    }
    if (FOLD && fmask) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if ((fmask >> u) & 1) {    // border pixel (rare): add the pre-images to the raw chunk
          const int pu = p + u * L.RL, fy = pu / x.W, fx = pu - fy * x.W;
          fold_extra_raw<T>(gy, gp, fy, fx, gr[u]);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const float ma = ACT == GAN_ACT_NONE ? 0.f : mean[2 * j], mb = ACT == GAN_ACT_NONE ? 0.f : mean[2 * j + 1];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float xa, xb, ga, gb;
        Pairs<T>::get(xr[u], j, xa, xb);
        Pairs<T>::get(gr[u], j, ga, gb);
        ga = act_mask<ACT>(ga, xa, ma); gb = act_mask<ACT>(gb, xb, mb);
        s1[2 * j] += ga; s2[2 * j] += ga * xa; s1[2 * j + 1] += gb; s2[2 * j + 1] += gb * xb;
      }
    }
  }
  row_lane_sum<N>(L, sh, s1);
  row_lane_sum<N>(L, sh, s2);
  if (L.rl == 0) {
    float* o = ws + ((int64_t)(b * nch + ch) * x.C + cofs) * 2;
#pragma unroll
    for (int e = 0; e < N; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
  }
}
// dx = rstd * (g' - mean(g') - xhat * mean(g'*xhat)) = g' * A + x * Bc + Cc with per-channel A = rstd, Bc = -rstd^2 m2, Cc = -rstd m1 +
// rstd^2 m2 mean, computed once per block (fp64) from the nch partials of in_bwd_partial_kernel and kept in LDS as (mean, A, Bc, Cc):
// a group of UNR pixels is finished pair of channels by pair of channels, so only two coefficient quads are live at a time.
// bias_part (optional): this block's row of the bias-gradient partials.  The gradient of a bias in front of a non-affine InstanceNorm is
// the column sum of dx, which is IDENTICALLY zero: sum_p dx = A S1 + Bc HW mean + HW Cc = rstd S1 - HW rstd (S1 / HW).  The reference's
// autograd value is the rounding noise of that sum (|g| ~ 1e-9); here the sum is evaluated in closed form from the
// same totals (fp64) by the image's first block -- the other blocks' rows are zero -- instead of being re-accumulated per element.
// AMAX: additionally amax[b] = max(amax[b], max |dx| of this block) (atomic max on the bit patterns of non-negative floats: order-independent).
template <typename T, int UNR, int ACT, bool FOLD, bool AMAX = false>
__global__ __launch_bounds__(NTHR, 6) void in_bwd_apply_kernel(DView x, const float* __restrict__ stats, DView gy, const float* __restrict__ ws, int nch,
                                                           DView dx, int nblk, float* __restrict__ bias_part, float* __restrict__ amax = nullptr, int s2_xhat = 0) {
  constexpr int N = Chunk<T>::N, NP = Pairs<T>::NP;
  Lanes<T> L(x.C);
  const int b = blockIdx.y, HW = x.H * x.W, cofs = L.cl * N;
  const int per = (HW + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  __shared__ float sh[NTHR * 8];
  f32x4_t* sh4 = reinterpret_cast<f32x4_t*>(sh);        // [C] x (mean, A, Bc, Cc); C <= NTHR * 8 / 4
  for (int c = threadIdx.x; c < x.C; c += NTHR) {
    const float2* pp = reinterpret_cast<const float2*>(ws + ((int64_t)b * nch * x.C + c) * 2);
    double S1 = 0.0, S2 = 0.0;
#pragma unroll 8
    for (int k = 0; k < nch; ++k) { const float2 t = pp[(int64_t)k * x.C]; S1 += t.x; S2 += t.y; }
    const double mu = stats[((int64_t)b * x.C + c) * 2], rs = stats[((int64_t)b * x.C + c) * 2 + 1];
    // s2_xhat: the second sum was taken against xhat itself (an input-gradient epilogue summing g * relu(xhat), gan_conv_desc.stats_mode 1)
    const double m1 = S1 / HW, m2 = s2_xhat ? S2 / HW : rs * (S2 - mu * S1) / HW;
    const double bcd = -rs * rs * m2, ccd = -rs * m1 + rs * rs * m2 * mu;
    const f32x4_t r = {(float)mu, (float)rs, (float)bcd, (float)ccd};
    sh4[c] = r;
    if (bias_part) bias_part[(int64_t)(b * nblk + blockIdx.x) * x.C + c] = blockIdx.x == 0 ? (float)(rs * S1 + bcd * (HW * mu) + HW * ccd) : 0.f;
  }
  __syncthreads();
  const f32x4_t* cf = sh4 + cofs;
  const T *xp = image_ptr<T>(x, b, cofs), *gp = image_ptr<T>(gy, b, cofs);
  T* dp = const_cast<T*>(image_ptr<T>(dx, b, cofs));
  int yy, xx; pixel_yx(p0 + L.rl, x.W, yy, xx);
  Off32 ox(x, yy, xx, L.RL, x.W), og(gy, yy, xx, L.RL, x.W), od(dx, yy, xx, L.RL, x.W);
  float mx = 0.f;
  for (int p = p0 + L.rl; p < p1; p += UNR * L.RL) {
    Raw<T> xr[UNR], gr[UNR];
    int ods[UNR];
    unsigned fmask = 0, live = 0;    // bit u: pixel u collects reflect pre-images (border pixels only) / lies inside this block's range
    asm volatile("" ::: "memory");   // the coefficient reads stay inside the loop (hoisted, they would cost 4N registers)
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      ods[u] = od.o;
      if (p + u * L.RL < p1) {
        xr[u] = ldraw(xp + ox.o); gr[u] = ldraw(gp + og.o);
        live |= 1u << u;
        if (FOLD && fold_needed(gy, yy, xx)) fmask |= 1u << u;
      }
      const int wr = pixel_step(L.RL, x.W, yy, xx);
      ox.advance(wr); og.advance(wr); od.advance(wr);
    }
    if (FOLD && fmask) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if ((fmask >> u) & 1) {    // border pixel (rare): add the pre-images to the raw chunk
          const int pu = p + u * L.RL, fy = pu / x.W, fx = pu - fy * x.W;
          fold_extra_raw<T>(gy, gp, fy, fx, gr[u]);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const f32x4_t ca = cf[2 * j], cb = cf[2 * j + 1];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float xa, xb, ga, gb;
        Pairs<T>::get(xr[u], j, xa, xb);
        Pairs<T>::get(gr[u], j, ga, gb);
        ga = act_mask<ACT>(ga, xa, ca[0]) * ca[1] + (xa * ca[2] + ca[3]);
        gb = act_mask<ACT>(gb, xb, cb[0]) * cb[1] + (xb * cb[2] + cb[3]);
        if (AMAX && ((live >> u) & 1)) mx = fmaxf(mx, fmaxf(fabsf(ga), fabsf(gb)));
        Pairs<T>::set(gr[u], j, ga, gb);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if ((live >> u) & 1) *reinterpret_cast<Raw<T>*>(dp + ods[u]) = gr[u];
  }
  if (AMAX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(amax) + b, __builtin_bit_cast(unsigned, mx));
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
template <typename T, int UNR, bool FOLD>
__global__ __launch_bounds__(NTHR, 6) void fold_add_kernel(DView a, int has_a, DView g, DView y, int act, DView out, int nblk) {
  constexpr int N = Chunk<T>::N;
  Lanes<T> L(out.C);
  const int b = blockIdx.y, HW = out.H * out.W, cofs = L.cl * N;
  const int per = (HW + nblk - 1) / nblk, p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  const bool has_y = act != GAN_ACT_NONE;
  const T *gp = image_ptr<T>(g, b, cofs), *ap = image_ptr<T>(has_a ? a : g, b, cofs), *yp = image_ptr<T>(has_y ? y : g, b, cofs);
  T* op = const_cast<T*>(image_ptr<T>(out, b, cofs));
  int yy, xx; pixel_yx(p0 + L.rl, out.W, yy, xx);
  Off32 og(g, yy, xx, L.RL, out.W), oa(has_a ? a : g, yy, xx, L.RL, out.W), oy(has_y ? y : g, yy, xx, L.RL, out.W), oo(out, yy, xx, L.RL, out.W);
  for (int p = p0 + L.rl; p < p1; p += UNR * L.RL) {
    Raw<T> gr[UNR], ar[UNR];
    int ys[UNR], xs[UNR], os[UNR], oys[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (FOLD) { ys[u] = yy; xs[u] = xx; }
      os[u] = oo.o; oys[u] = oy.o;
      if (p + u * L.RL < p1) {
        gr[u] = ldraw(gp + og.o);
        if (has_a) ar[u] = ldraw(ap + oa.o);
      }
      const int wr = pixel_step(L.RL, out.W, yy, xx);
      og.advance(wr); oa.advance(wr); oy.advance(wr); oo.advance(wr);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * L.RL < p1) {
        float v[N], t[N];
        cvtraw<T>(gr[u], v);
        if (FOLD) fold_extra<T>(g, gp, ys[u], xs[u], v);
        if (has_a) {
          cvtraw<T>(ar[u], t);
#pragma unroll
          for (int e = 0; e < N; ++e) v[e] += t[e];
        }
        if (has_y) {
          Chunk<T>::load(yp + oys[u], t);
#pragma unroll
          for (int e = 0; e < N; ++e) v[e] *= act_grad_from_out(t[e], act);
        }
        Chunk<T>::store(op + os[u], v);
      }
    }
  }
}

// ------------------------------------------------------------------ layout boundary
// gradient of a padding layer (gan_pad_fold): every interior pixel adds up the padded positions the padding copied from it
template <typename T>
__global__ void pad_fold_kernel(DView g, int mode, DView out) {
  constexpr int N = Chunk<T>::N;
  const int nck = out.C / N, py = g.y0, px = g.x0;
  const int64_t total = (int64_t)out.B * out.H * out.W * nck;
  const T* gp = reinterpret_cast<const T*>(g.ptr);
  T* op = reinterpret_cast<T*>(out.ptr);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ck = (int)(i % nck);
    int64_t r = i / nck;
    const int x = (int)(r % out.W); r /= out.W;
    const int y = (int)(r % out.H);
    const int b = (int)(r / out.H);
    float acc[N];
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    // candidates: the pixel's own padded position and the 2*pad halo rows / columns (at most (2*pad+1)^2 positions, most rejected)
    for (int jy = 0; jy <= 2 * py; ++jy) {
      const int yp = jy == 2 * py ? y + py : (jy < py ? jy : g.H + jy);
      const int sy = mode == GAN_HALO_REFLECT ? reflect_idx(yp - py, g.H) : min(max(yp - py, 0), g.H - 1);
      if (sy != y) continue;
      for (int jx = 0; jx <= 2 * px; ++jx) {
        const int xp = jx == 2 * px ? x + px : (jx < px ? jx : g.W + jx);
        const int sx = mode == GAN_HALO_REFLECT ? reflect_idx(xp - px, g.W) : min(max(xp - px, 0), g.W - 1);
        if (sx != x) continue;
        float t[N];
        Chunk<T>::load(gp + g.pixp(b, yp, xp) + ck * N, t);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] += t[e];
      }
    }
    Chunk<T>::store(op + out.pix(b, y, x) + ck * N, acc);
  }
}

template <typename T>
__global__ void nchw_to_view_kernel(const float* __restrict__ src, int C, DView dst, int halo_mode) {
  constexpr int N = Chunk<T>::N;
  const bool padded = halo_mode == GAN_HALO_REFLECT || halo_mode == GAN_HALO_REPLICATE;
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
    if (halo_mode == GAN_HALO_REPLICATE) { sy = min(max(dy - dst.y0, 0), dst.H - 1); sx = min(max(dx - dst.x0, 0), dst.W - 1); }
    else if (padded) { sy = reflect_idx(dy - dst.y0, dst.H); sx = reflect_idx(dx - dst.x0, dst.W); }
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
  if (v->dtype != GAN_F32 && v->dtype != GAN_BF16) return gan_set_error(-1, "%s: dtype %d (these passes run on fp32 / bf16 buffers; fp8 is an operand-copy format)", what, v->dtype);
  const int epc = v->dtype == GAN_F32 ? 4 : 8;
  const int cl = v->C / epc;
  if (v->C % epc != 0 || cl > NTHR || (cl & (cl - 1)) != 0) return gan_set_error(-1, "%s: C=%d unsupported (C/%d must be a power of two <= 256)", what, v->C, epc);
  return 0;
}
// row-chunks per image for the statistics passes: ~2048 16-byte loads per block, at most MAXCH (workspace bound)
int work_per_block() {   // 16-byte loads per block and operand; GAN_NORM_WORK overrides (tuning aid)
  static int w = 0;
  if (!w) { const char* e = getenv("GAN_NORM_WORK"); w = e ? atoi(e) : 4096; if (w < 256) w = 256; }
  return w;
}
// partials per image for the passes whose consumer sums them itself (<= MAXPARTS): as many as keep ~2 blocks per CU busy
int nparts_for(int B, int HW, int cl) {
  int64_t n = ((int64_t)HW * cl + 2047) / 2048;       // at least 2048 loads per block
  int64_t want = (512 + B - 1) / B;                    // ~512 blocks per launch
  if (n > want) n = want;
  if (n > MAXPARTS) n = MAXPARTS;
  if (n < 1) n = 1;
  return (int)n;
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

// pixels in flight per thread of the streaming kernels: 2 for the backward family (fits beside a resident weight-gradient block: <= 80
// registers), 4 for the forward passes, which run alone.  GAN_NORM_UNR / GAN_NORM_UNR_FWD = 2, 3 or 4 override (tuning aid, read once).
static int norm_unroll(bool fwd) {
  static int u[2] = {0, 0};
  if (!u[fwd]) { const char* e = getenv(fwd ? "GAN_NORM_UNR_FWD" : "GAN_NORM_UNR"); const int v = e ? atoi(e) : (fwd ? 4 : 2); u[fwd] = v >= 2 && v <= 4 ? v : (fwd ? 4 : 2); }
  return u[fwd];
}
#define GAN_DISPATCH_NORM_U(dt, fwd, ...)                                            \
  {                                                                                  \
    const int unr__ = norm_unroll(fwd);                                              \
    if ((dt) == GAN_F32) { typedef float T;                                          \
      if (unr__ == 2) { constexpr int U = 2; __VA_ARGS__ } else if (unr__ == 3) { constexpr int U = 3; __VA_ARGS__ } else { constexpr int U = 4; __VA_ARGS__ } } \
    else { typedef bf16_t T;                                                         \
      if (unr__ == 2) { constexpr int U = 2; __VA_ARGS__ } else if (unr__ == 3) { constexpr int U = 3; __VA_ARGS__ } else { constexpr int U = 4; __VA_ARGS__ } } \
  }
#define GAN_DISPATCH_NORM(dt, ...) GAN_DISPATCH_NORM_U(dt, false, __VA_ARGS__)
#define GAN_DISPATCH_NORM_FWD(dt, ...) GAN_DISPATCH_NORM_U(dt, true, __VA_ARGS__)

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
  GAN_DISPATCH_NORM_FWD(x->dtype, hipLaunchKernelGGL((in_partial_kernel<T, U>), dim3(nch, x->B), dim3(NTHR), 0, s, dx, nch, ws);)
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
  GAN_DISPATCH_NORM_FWD(x->dtype, hipLaunchKernelGGL((in_apply_kernel<T, U>), dim3(nblk, x->B), dim3(NTHR), 0, (hipStream_t)stream, dx,
                                                  const_cast<float*>(stats), (const float*)nullptr, 0, 0.f, act, dr, residual ? 1 : 0, dy, halo_mode, nblk, (uint8_t*)nullptr);)
  GAN_LAUNCH_CHECK();
  return 0;
}

// Statistics partials of x for gan_in_apply_parts: parts = fp32 [B][gan_in_partial_count(x)][C][2] (sum, sum of squares per chunk of pixels)
extern "C" int gan_in_partial_count(const gan_view* x) {
  if (gan_check_view(x, "in_partial_count.x")) return -1;
  return nparts_for(x->B, x->H * x->W, lanes_of(x));
}
extern "C" int gan_in_partial(const gan_view* x, float* parts, void* stream) {
  VCHK(x, "in_partial.x");
  if (check_lanes(x, "in_partial")) return -1;
  GAN_CHECK(parts, "in_partial: null pointer");
  const int nch = nparts_for(x->B, x->H * x->W, lanes_of(x));
  DView dx = to_dview(x);
  GAN_DISPATCH_NORM_FWD(x->dtype, hipLaunchKernelGGL((in_partial_kernel<T, U>), dim3(nch, x->B), dim3(NTHR), 0, (hipStream_t)stream, dx, nch, parts);)
  GAN_LAUNCH_CHECK();
  return 0;
}

// gan_in_apply with the statistics taken from per-chunk partials (gan_in_partial, or a convolution epilogue's gan_conv_desc.stats):
// every block sums the nparts (<= 16) partial pairs of its image itself, so no finalize launch precedes the pass; the image's first
// block also writes (mean, rstd) to `stats` for the backward pass.
static int in_apply_parts_impl(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                               const gan_view* y, const gan_view* y8, int halo_mode, void* stream);
extern "C" int gan_in_apply_parts(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                                  const gan_view* y, int halo_mode, void* stream) {
  return in_apply_parts_impl(x, parts, nparts, eps, stats, act, residual, y, nullptr, halo_mode, stream);
}
// the same pass writing, besides y, an e4m3 copy y8 of it (GAN_FP8 view of y's geometry; unit scale): gan_quantize_fp8(y, y8) for free
extern "C" int gan_in_apply_parts_fp8(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                                      const gan_view* y, const gan_view* y8, int halo_mode, void* stream) {
  if (gan_check_view(y8, "in_apply_parts_fp8.y8") || gan_check_view(y, "in_apply_parts_fp8.y")) return -1;
  GAN_CHECK(y8->dtype == GAN_FP8 && y8->B == y->B && y8->Hp == y->Hp && y8->Wp == y->Wp && y8->C == y->C && y8->y0 == y->y0 && y8->x0 == y->x0,
            "in_apply_parts_fp8: y8 must be an fp8 view of y's geometry");
  return in_apply_parts_impl(x, parts, nparts, eps, stats, act, residual, y, y8, halo_mode, stream);
}
static int in_apply_parts_impl(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                               const gan_view* y, const gan_view* y8, int halo_mode, void* stream) {
  VCHK(x, "in_apply_parts.x"); VCHK(y, "in_apply_parts.y");
  if (check_lanes(x, "in_apply_parts")) return -1;
  SAME_SHAPE(x, y, "in_apply_parts(x,y)");
  if (residual) { VCHK(residual, "in_apply_parts.residual"); SAME_SHAPE(x, residual, "in_apply_parts(x,residual)"); }
  GAN_CHECK(parts && stats, "in_apply_parts: null pointer");
  GAN_CHECK(nparts >= 1 && nparts <= MAXPARTS, "in_apply_parts: nparts=%d outside 1..%d (use gan_in_stats_from_parts + gan_in_apply)", nparts, MAXPARTS);
  GAN_CHECK(x->C <= SUMS_MAXC, "in_apply_parts: C=%d > %d", x->C, SUMS_MAXC);
  if (halo_mode == GAN_HALO_REFLECT)
    GAN_CHECK(y->y0 < y->H && y->x0 < y->W && 2 * y->y0 + y->H <= y->Hp && 2 * y->x0 + y->W <= y->Wp, "in_apply_parts: reflect halo does not fit");
  const int DH = halo_mode == GAN_HALO_REFLECT ? y->H + 2 * y->y0 : y->H, DW = halo_mode == GAN_HALO_REFLECT ? y->W + 2 * y->x0 : y->W;
  const int nblk = nblocks_for(DH * DW, lanes_of(x));
  DView dx = to_dview(x), dy = to_dview(y), dr = residual ? to_dview(residual) : null_dview();
  uint8_t* p8 = y8 ? (uint8_t*)y8->ptr : nullptr;
  GAN_DISPATCH_NORM_FWD(x->dtype, hipLaunchKernelGGL((in_apply_kernel<T, U>), dim3(nblk, x->B), dim3(NTHR), 0, (hipStream_t)stream, dx, stats, parts, nparts,
                                                  eps, act, dr, residual ? 1 : 0, dy, halo_mode, nblk, p8);)
  GAN_LAUNCH_CHECK();
  return 0;
}

// ws: fp32, >= B*MAXCH*C*2 + B*C*2 floats
static int in_bwd_impl(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2, const gan_view* dx,
                       float* ws, float* bias_grad, int bias_n, int bias_acc, void* stream, float* bias_part = nullptr, float* amax = nullptr);

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
                       float* ws, float* bias_grad, int bias_n, int bias_acc, void* stream, float* bias_part, float* amax) {
  VCHK(x, "in_bwd.x"); VCHK(gy, "in_bwd.gy"); VCHK(dx, "in_bwd.dx");
  if (check_lanes(x, "in_bwd")) return -1;
  SAME_SHAPE(x, gy, "in_bwd(x,gy)"); SAME_SHAPE(x, dx, "in_bwd(x,dx)");
  if (g2) { VCHK(g2, "in_bwd.g2"); SAME_SHAPE(x, g2, "in_bwd(x,g2)"); }
  if (fold_ok(gy, fold)) return -1;
  GAN_CHECK(stats && ws, "in_bwd: null pointer");
  GAN_CHECK(act == GAN_ACT_NONE || act == GAN_ACT_RELU || act == GAN_ACT_LRELU, "in_bwd: unsupported activation %d", act);
  GAN_CHECK(x->C <= NTHR * 8 / 4, "in_bwd: C=%d > %d", x->C, NTHR * 8 / 4);
  // two launches: partial sums (sum g, sum g*xhat) per (image, chunk), then the apply pass, whose blocks add the <= 16 chunks up themselves
  const int HW = x->H * x->W, nch = nparts_for(x->B, HW, lanes_of(x)), BC = x->B * x->C;
  float* ws2 = ws + (int64_t)x->B * MAXCH * x->C * 2;
  DView vx = to_dview(x), vg = to_dview(gy), v2 = g2 ? to_dview(g2) : null_dview(), vd = to_dview(dx);
  hipStream_t s = (hipStream_t)stream;
  const int nblk = nblocks_for(HW, lanes_of(x));
  float* bp = bias_part ? bias_part : (bias_grad ? ws2 + (int64_t)BC * 2 : nullptr);
  if (g2) {   // second addend (rare: the trainers never pass one): dx <- fold(gy) + g2 first, then the norm backward in place on dx
    GAN_DISPATCH_NORM(x->dtype,
      if (fold) hipLaunchKernelGGL((fold_add_kernel<T, U, true>), dim3(nblk, x->B), dim3(NTHR), 0, s, v2, 1, vg, null_dview(), GAN_ACT_NONE, vd, nblk);
      else hipLaunchKernelGGL((fold_add_kernel<T, U, false>), dim3(nblk, x->B), dim3(NTHR), 0, s, v2, 1, vg, null_dview(), GAN_ACT_NONE, vd, nblk);)
    vg = vd;
    fold = 0;
  }
  GAN_DISPATCH_NORM(x->dtype,
    auto go = [&](auto act_c, auto fold_c) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool FOLD = decltype(fold_c)::value;
      hipLaunchKernelGGL((in_bwd_partial_kernel<T, U, ACT, FOLD>), dim3(nch, x->B), dim3(NTHR), 0, s, vx, stats, vg, nch, ws);
      if (amax) hipLaunchKernelGGL((in_bwd_apply_kernel<T, U, ACT, FOLD, true>), dim3(nblk, x->B), dim3(NTHR), 0, s, vx, stats, vg, ws, nch, vd, nblk, bp, amax);
      else hipLaunchKernelGGL((in_bwd_apply_kernel<T, U, ACT, FOLD>), dim3(nblk, x->B), dim3(NTHR), 0, s, vx, stats, vg, ws, nch, vd, nblk, bp, (float*)nullptr);
    };
    using std::integral_constant;
    if (fold) {
      if (act == GAN_ACT_RELU) go(integral_constant<int, GAN_ACT_RELU>{}, integral_constant<bool, true>{});
      else if (act == GAN_ACT_LRELU) go(integral_constant<int, GAN_ACT_LRELU>{}, integral_constant<bool, true>{});
      else go(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, true>{});
    } else {
      if (act == GAN_ACT_RELU) go(integral_constant<int, GAN_ACT_RELU>{}, integral_constant<bool, false>{});
      else if (act == GAN_ACT_LRELU) go(integral_constant<int, GAN_ACT_LRELU>{}, integral_constant<bool, false>{});
      else go(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, false>{});
    })
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

extern "C" int gan_in_bwd_amax(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* dx, float* ws,
                               float* bias_part, float* amax, void* stream) {
  GAN_CHECK(amax, "in_bwd_amax: null amax");
  if (hipMemsetAsync(amax, 0, sizeof(float) * (x ? x->B : 0), (hipStream_t)stream) != hipSuccess) return gan_set_error(-2, "in_bwd_amax: memset failed");
  return in_bwd_impl(x, stats, act, gy, fold, nullptr, dx, ws, nullptr, 0, 0, stream, bias_part, amax);
}

// The apply pass alone: the two sums per (image, channel) come as partials from the launch that produced gy (an input-gradient epilogue,
// gan_conv_desc.stats_mode 1 | 2), so x and gy are read once instead of twice.
extern "C" int gan_in_bwd_parts(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* dx, const float* parts,
                                int nparts, int parts_mode, float* bias_part, void* stream) {
  VCHK(x, "in_bwd_parts.x"); VCHK(gy, "in_bwd_parts.gy"); VCHK(dx, "in_bwd_parts.dx");
  if (check_lanes(x, "in_bwd_parts")) return -1;
  SAME_SHAPE(x, gy, "in_bwd_parts(x,gy)"); SAME_SHAPE(x, dx, "in_bwd_parts(x,dx)");
  if (fold_ok(gy, fold)) return -1;
  GAN_CHECK(stats && parts, "in_bwd_parts: null pointer");
  GAN_CHECK(nparts >= 1 && nparts <= MAXCH, "in_bwd_parts: nparts=%d outside 1..%d", nparts, MAXCH);
  GAN_CHECK(parts_mode == 1 || parts_mode == 2, "in_bwd_parts: parts_mode %d (1: sums against xhat, 2: against the raw x)", parts_mode);
  GAN_CHECK(act == GAN_ACT_NONE || act == GAN_ACT_RELU || act == GAN_ACT_LRELU, "in_bwd_parts: unsupported activation %d", act);
  GAN_CHECK(parts_mode != 1 || act == GAN_ACT_RELU, "in_bwd_parts: sums against relu(xhat) belong to a ReLU'd norm");
  GAN_CHECK(x->C <= NTHR * 8 / 4, "in_bwd_parts: C=%d > %d", x->C, NTHR * 8 / 4);
  const int HW = x->H * x->W, nblk = nblocks_for(HW, lanes_of(x));
  DView vx = to_dview(x), vg = to_dview(gy), vd = to_dview(dx);
  hipStream_t s = (hipStream_t)stream;
  const int xh = parts_mode == 1 ? 1 : 0;
  GAN_DISPATCH_NORM(x->dtype,
    auto go = [&](auto act_c, auto fold_c) {
      constexpr int ACT = decltype(act_c)::value;
      constexpr bool FOLD = decltype(fold_c)::value;
      hipLaunchKernelGGL((in_bwd_apply_kernel<T, U, ACT, FOLD>), dim3(nblk, x->B), dim3(NTHR), 0, s, vx, stats, vg, parts, nparts, vd, nblk, bias_part, (float*)nullptr, xh);
    };
    using std::integral_constant;
    if (fold) {
      if (act == GAN_ACT_RELU) go(integral_constant<int, GAN_ACT_RELU>{}, integral_constant<bool, true>{});
      else if (act == GAN_ACT_LRELU) go(integral_constant<int, GAN_ACT_LRELU>{}, integral_constant<bool, true>{});
      else go(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, true>{});
    } else {
      if (act == GAN_ACT_RELU) go(integral_constant<int, GAN_ACT_RELU>{}, integral_constant<bool, false>{});
      else if (act == GAN_ACT_LRELU) go(integral_constant<int, GAN_ACT_LRELU>{}, integral_constant<bool, false>{});
      else go(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, false>{});
    })
  GAN_LAUNCH_CHECK();
  return 0;
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
  GAN_DISPATCH_NORM(out->dtype,
    if (fold) hipLaunchKernelGGL((fold_add_kernel<T, U, true>), dim3(nblk, out->B), dim3(NTHR), 0, (hipStream_t)stream, va, a ? 1 : 0, vb, null_dview(), GAN_ACT_NONE, vo, nblk);
    else hipLaunchKernelGGL((fold_add_kernel<T, U, false>), dim3(nblk, out->B), dim3(NTHR), 0, (hipStream_t)stream, va, a ? 1 : 0, vb, null_dview(), GAN_ACT_NONE, vo, nblk);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_pad_fold(const gan_view* g, int mode, const gan_view* out, void* stream) {
  VCHK(g, "pad_fold.g"); VCHK(out, "pad_fold.out");
  if (check_lanes(out, "pad_fold")) return -1;
  SAME_SHAPE(g, out, "pad_fold(g,out)");
  GAN_CHECK(mode == GAN_HALO_REPLICATE || mode == GAN_HALO_REFLECT, "pad_fold: mode %d", mode);
  GAN_CHECK(g->y0 + g->H + g->y0 <= g->Hp && g->x0 + g->W + g->x0 <= g->Wp && (mode == GAN_HALO_REPLICATE || (g->y0 < g->H && g->x0 < g->W)),
            "pad_fold: g must carry a symmetric halo of y0/x0 pixels");
  const int epc = out->dtype == GAN_F32 ? 4 : 8;
  const int64_t total = (int64_t)out->B * out->H * out->W * (out->C / epc);
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  DView vg = to_dview(g), vo = to_dview(out);
  GAN_DISPATCH_DTYPE(out->dtype, hipLaunchKernelGGL((pad_fold_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, vg, mode, vo);)
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
  GAN_DISPATCH_NORM(dx->dtype,
    if (fold) hipLaunchKernelGGL((fold_add_kernel<T, U, true>), dim3(nblk, dx->B), dim3(NTHR), 0, (hipStream_t)stream, v2, g2 ? 1 : 0, vg, vy, act, vd, nblk);
    else hipLaunchKernelGGL((fold_add_kernel<T, U, false>), dim3(nblk, dx->B), dim3(NTHR), 0, (hipStream_t)stream, v2, g2 ? 1 : 0, vg, vy, act, vd, nblk);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_nchw_to_view(const float* src, int C, const gan_view* dst, int halo_mode, void* stream) {
  VCHK(dst, "nchw_to_view.dst");
  GAN_CHECK(src && C > 0 && C <= dst->C, "nchw_to_view: bad C=%d", C);
  const bool padded = halo_mode == GAN_HALO_REFLECT || halo_mode == GAN_HALO_REPLICATE;
  if (padded) GAN_CHECK((halo_mode == GAN_HALO_REPLICATE || (dst->y0 < dst->H && dst->x0 < dst->W)) && 2 * dst->y0 + dst->H <= dst->Hp && 2 * dst->x0 + dst->W <= dst->Wp, "nchw_to_view: halo does not fit");
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
