// Fused clip_grad_norm_ + Adam + EMA over a whole parameter list in three launches
// (sum of squares -> update -> step counters), independent of the number of tensors.
//
// Replaces, per optimiser step of the reference: torch.nn.utils.clip_grad_norm_ and GradScaler.step ->
// torch.optim.Adam.step (GAN_Variant1/utils/amp_utils.py:29-41; GAN_Variant1/training/sched_optim.py:5-27;
// Basic_GAN/src/train.py:45-50,96,105,114) and EMA.update (GAN_Variant1/utils/io_ckpt.py:23-29), which on a GPU
// are ~200 tiny foreach launches.  Maths follows torch/optim/adam.py:528-546 (single-tensor, no amsgrad,
// weight_decay 0): m.lerp_(g, 1-b1); v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps), with
// per-tensor step counts (a tensor whose grad is None is skipped entirely, exactly as torch does).
// Bias corrections are evaluated in fp64 on device from the device-resident step counters, so a captured
// hipGraph replays correctly.  HBM-bound: 28 B/param (+8 B/param with EMA).
#include "common.h"

namespace {

constexpr int CHUNK = 16384;  // elements per block

__global__ __launch_bounds__(256) void adam_sumsq_kernel(const gan_adam_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off, float grad_scale, const float* __restrict__ inv_scale_dev,
                                                        float* __restrict__ ws) {
  const gan_adam_tensor t = table[chunk_tensor[blockIdx.x]];
  if (inv_scale_dev) grad_scale *= *inv_scale_dev;      // GradScaler.unscale_: the loss scale's reciprocal lives on the device
  float s = 0.f;
  if (t.g) {
    const int64_t off = chunk_off[blockIdx.x];
    const int64_t n = t.numel - off < CHUNK ? t.numel - off : CHUNK;
    const float* g = t.g + off;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const float v = g[i] * grad_scale;
      s += v * v;
    }
  }
  __shared__ float sh[16];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void adam_apply_kernel(const gan_adam_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off, int nchunks, float lr, float beta1, float beta2,
                                                        float eps, float max_norm, float grad_scale, float ema_decay,
                                                        const float* __restrict__ lr_dev, const float* __restrict__ inv_scale_dev, int skip_nonfinite,
                                                        float* __restrict__ norm_out, const float* __restrict__ ws) {
  __shared__ float sh[16];
  __shared__ float s_bc[2];
  // every block re-derives the global norm from the per-chunk partials (a few thousand floats, L2-resident)
  float s = 0.f;
  for (int i = threadIdx.x; i < nchunks; i += 256) s += ws[i];
  s = block_sum(s, sh);
  const float total = sqrtf(s);
  const float coef = max_norm > 0.f ? fminf(1.f, max_norm / (total + 1e-6f)) : 1.f;
  // GradScaler.step: an inf / nan anywhere in the (unscaled) gradients shows in the global norm; the whole step is then skipped
  const bool found_inf = !(total <= 3.4028234e38f);
  if (blockIdx.x == 0 && threadIdx.x == 0) { norm_out[0] = total; norm_out[1] = coef; norm_out[2] = found_inf ? 1.f : 0.f; }
  if (skip_nonfinite && found_inf) return;
  if (lr_dev) lr = *lr_dev;                             // LambdaLR: the host rewrites one device float, the prebuilt launch stays
  if (inv_scale_dev) grad_scale *= *inv_scale_dev;
  const gan_adam_tensor t = table[chunk_tensor[blockIdx.x]];
  if (!t.g) return;
  if (threadIdx.x == 0) {
    const double step = (double)(*t.step + 1);
    s_bc[0] = (float)(1.0 - pow((double)beta1, step));
    s_bc[1] = (float)sqrt(1.0 - pow((double)beta2, step));
  }
  __syncthreads();
  const float step_size = lr / s_bc[0], bc2s = s_bc[1];
  const float gs = grad_scale * coef, w1 = 1.f - beta1, w2 = 1.f - beta2, we = 1.f - ema_decay;
  const int64_t off = chunk_off[blockIdx.x];
  const int64_t n = t.numel - off < CHUNK ? t.numel - off : CHUNK;
  float* p = t.p + off; float* m = t.m + off; float* v = t.v + off;
  const float* g = t.g + off;
  float* ema = t.ema ? t.ema + off : nullptr;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float gi = g[i] * gs;
    float mi = m[i], vi = v[i];
    mi = w1 < 0.5f ? mi + w1 * (gi - mi) : gi - (gi - mi) * (1.f - w1);  // at::lerp
    vi = vi * beta2 + w2 * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    const float pi = p[i] - step_size * (mi / denom);
    m[i] = mi; v[i] = vi; p[i] = pi;
    if (ema) ema[i] = we * pi + ema_decay * ema[i];
  }
}

__global__ void adam_bump_kernel(const gan_adam_tensor* __restrict__ table, int ntensors, int skip_nonfinite, const float* __restrict__ norm_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (skip_nonfinite && norm_out[2] != 0.f) return;     // a skipped step does not count (torch: optimizer.step is not called)
  if (i < ntensors && table[i].g) *table[i].step += 1;
}

// torch.amp.GradScaler.update (amp_utils.py:22,41 call it after every step): backoff on an overflow, growth after `interval` clean steps
__global__ void scaler_update_kernel(float* scale, float* inv_scale, int32_t* tracker, const float* found_inf, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float sc = *scale;
  if (*found_inf != 0.f) { sc *= backoff; *tracker = 0; }
  else if (*tracker + 1 >= interval) { sc *= growth; *tracker = 0; }
  else *tracker += 1;
  *scale = sc;
  *inv_scale = 1.f / sc;
}

}  // namespace

// ws: fp32 >= nchunks floats.  Chunks are CHUNK=16384-element slices: chunk_tensor[k], chunk_off[k].
extern "C" int gan_adam_step(const gan_adam_tensor* table, int ntensors, const int32_t* chunk_tensor, const int64_t* chunk_off, int nchunks,
                             float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale, float ema_decay,
                             const float* lr_dev, const float* inv_scale_dev, int skip_nonfinite, float* norm_out, float* ws, void* stream) {
  GAN_CHECK(table && chunk_tensor && chunk_off && norm_out && ws && ntensors > 0 && nchunks > 0, "adam: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_sumsq_kernel, dim3(nchunks), dim3(256), 0, s, table, chunk_tensor, chunk_off, grad_scale, inv_scale_dev, ws);
  hipLaunchKernelGGL(adam_apply_kernel, dim3(nchunks), dim3(256), 0, s, table, chunk_tensor, chunk_off, nchunks, lr, beta1, beta2, eps, max_norm,
                     grad_scale, ema_decay, lr_dev, inv_scale_dev, skip_nonfinite, norm_out, ws);
  hipLaunchKernelGGL(adam_bump_kernel, dim3((ntensors + 63) / 64), dim3(64), 0, s, table, ntensors, skip_nonfinite, norm_out);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_scaler_update(float* scale, float* inv_scale, int32_t* growth_tracker, const float* found_inf, float growth_factor,
                                 float backoff_factor, int growth_interval, void* stream) {
  GAN_CHECK(scale && inv_scale && growth_tracker && found_inf && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
            "scaler_update: bad arguments");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale, inv_scale, growth_tracker, found_inf, growth_factor, backoff_factor,
                     growth_interval);
  GAN_LAUNCH_CHECK();
  return 0;
}
