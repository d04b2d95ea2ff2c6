// Spectral normalisation of a convolution weight, forward and backward.
//
// Replaces torch.nn.utils.spectral_norm as the reference applies it to every discriminator convolution when
// `use_spectral_norm` is set (GAN_Variant1/models/discriminator_patchgan.py:21-23; Basic_GAN/src/models.py:68-69):
// W is weight_orig viewed as an h x w matrix (h = Cout, w = Cin*kh*kw; OIHW is already that matrix, row-major),
//   training-mode forward:  v <- normalize(W^T u),  u <- normalize(W v)      (one power iteration, in place, no gradient)
//   always:                 sigma = u . (W v),  W_sn = W / sigma
//   backward (u, v constants):  dL/dW = (G - <G, W_sn> u v^T) / sigma,  G = dL/dW_sn
// The matrices are small (<= 512 x 8192 fp32); every kernel is one pass over W at HBM/L2 speed, reductions are fixed-order
// (no atomics), so results are deterministic.
#include "common.h"

namespace {

// t[j] = sum_i W[i][j] * u[i]      (one thread per column, coalesced across j)
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ W, int h, int w, const float* __restrict__ u, float* __restrict__ t) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= w) return;
  float s = 0.f;
  for (int i = 0; i < h; ++i) s += W[(int64_t)i * w + j] * u[i];
  t[j] = s;
}

// out = x / max(||x||, eps), single block
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* __restrict__ x, int n, float eps, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i] * x[i];
  s = block_sum(s, sh);
  const float d = fmaxf(sqrtf(s), eps);
  for (int i = threadIdx.x; i < n; i += 1024) out[i] = x[i] / d;
}

// s[i] = sum_j W[i][j] * v[j]      (one block per row)
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W, int w, const float* __restrict__ v, float* __restrict__ s) {
  __shared__ float sh[16];
  const float* row = W + (int64_t)blockIdx.x * w;
  float a = 0.f;
  for (int j = threadIdx.x; j < w; j += 256) a += row[j] * v[j];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) s[blockIdx.x] = a;
}

// power_iter: u <- s / max(||s||, eps); then sigma = u . s      (s = W v), single block
__global__ __launch_bounds__(1024) void sn_sigma_kernel(const float* __restrict__ s, int h, int power_iter, float eps, float* __restrict__ u, float* __restrict__ sigma) {
  __shared__ float sh[16];
  float d = 1.f;
  if (power_iter) {
    float q = 0.f;
    for (int i = threadIdx.x; i < h; i += 1024) q += s[i] * s[i];
    q = block_sum(q, sh);
    d = fmaxf(sqrtf(q), eps);
  }
  float a = 0.f;
  for (int i = threadIdx.x; i < h; i += 1024) {
    float ui = u[i];
    if (power_iter) { ui = s[i] / d; u[i] = ui; }
    a += ui * s[i];
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) sigma[0] = a;
}

__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ W, int64_t n, const float* __restrict__ sigma, float* __restrict__ out) {
  const float sg = sigma[0];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = W[i] / sg;
}

// part[b] = sum over a fixed slice of G .* Wsn
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ G, const float* __restrict__ Wsn, int64_t n, float* __restrict__ part) {
  __shared__ float sh[16];
  float a = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += G[i] * Wsn[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// dW[i][j] = (G[i][j] - dot * u[i] * v[j]) / sigma,  dot = sum(part[0..np))  (every block re-reduces the np partials: np <= 256)
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ G, const float* __restrict__ part, int np, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma, int h, int w, float* __restrict__ dW) {
  __shared__ float sh[16];
  float a = threadIdx.x < np ? part[threadIdx.x] : 0.f;
  const float dot = block_sum(a, sh);
  const float sg = sigma[0];
  const int64_t n = (int64_t)h * w;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / w), c = (int)(i - (int64_t)r * w);
    dW[i] = (G[i] - dot * u[r] * v[c]) / sg;
  }
}
}  // namespace

extern "C" int64_t gan_spectral_norm_ws_floats(int h, int w) { return (int64_t)h + w + 256 + 16; }

extern "C" int gan_spectral_norm_fwd(const float* W, int h, int w, float* u, float* v, int power_iter, float eps, float* sigma, float* Wsn,
                                     float* ws, void* stream) {
  GAN_CHECK(W && u && v && sigma && Wsn && ws && h > 0 && w > 0, "spectral_norm_fwd: null pointer or empty matrix (h=%d, w=%d)", h, w);
  hipStream_t s = (hipStream_t)stream;
  float* t = ws;          // [w]
  float* sv = ws + w;     // [h]
  if (power_iter) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((w + 255) / 256), dim3(256), 0, s, W, h, w, u, t);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, t, w, eps, v);
  }
  hipLaunchKernelGGL(sn_w_v_kernel, dim3(h), dim3(256), 0, s, W, w, v, sv);
  hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(1024), 0, s, sv, h, power_iter, eps, u, sigma);
  const int64_t n = (int64_t)h * w;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(grid), dim3(256), 0, s, W, n, sigma, Wsn);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_spectral_norm_bwd(const float* G, const float* Wsn, const float* u, const float* v, const float* sigma, int h, int w, float* dW,
                                     float* ws, void* stream) {
  GAN_CHECK(G && Wsn && u && v && sigma && dW && ws && h > 0 && w > 0, "spectral_norm_bwd: null pointer or empty matrix (h=%d, w=%d)", h, w);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)h * w;
  const int np = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  float* part = ws + h + w;   // [256]
  hipLaunchKernelGGL(sn_dot_kernel, dim3(np), dim3(256), 0, s, G, Wsn, n, part);
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid), dim3(256), 0, s, G, part, np, u, v, sigma, h, w, dW);
  GAN_LAUNCH_CHECK();
  return 0;
}
