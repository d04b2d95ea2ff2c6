// Error reporting, view validation and small fp32 helpers of libmi355x_gan.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

int gan_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* gan_last_error(void) { return g_err; }
extern "C" int gan_version(void) { return 100; }

int gan_check_view(const gan_view* v, const char* name) {
  if (!v || !v->ptr) return gan_set_error(-1, "%s: null view", name);
  if (v->dtype != GAN_F32 && v->dtype != GAN_BF16 && v->dtype != GAN_FP8) return gan_set_error(-1, "%s: bad dtype %d", name, v->dtype);
  if (v->C <= 0 || v->C % (v->dtype == GAN_FP8 ? 16 : 8) != 0) return gan_set_error(-1, "%s: C=%d must be a positive multiple of %d", name, v->C, v->dtype == GAN_FP8 ? 16 : 8);
  if (v->B <= 0 || v->H <= 0 || v->W <= 0 || v->y0 < 0 || v->x0 < 0 || v->y0 + v->H > v->Hp || v->x0 + v->W > v->Wp)
    return gan_set_error(-1, "%s: logical window %dx%d at (%d,%d) outside allocation %dx%d", name, v->H, v->W, v->y0, v->x0, v->Hp, v->Wp);
  if ((uintptr_t)v->ptr % 16 != 0) return gan_set_error(-1, "%s: pointer must be 16-byte aligned", name);
  return 0;
}

namespace {
__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void axpy_kernel(float* y, const float* x, float a, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += a * x[i];
}
}  // namespace

extern "C" int gan_fill_f32(float* p, int64_t n, float v, void* stream) {
  GAN_CHECK(p && n >= 0, "fill: bad arguments");
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, n, v);
  GAN_LAUNCH_CHECK();
  return 0;
}
extern "C" int gan_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream) {
  GAN_CHECK(y && x && n >= 0, "axpy: bad arguments");
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  GAN_LAUNCH_CHECK();
  return 0;
}
