// Probe: operand lane layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 data, the meaning of scale 0x7f, and what
// v_cvt_pk_fp8_f32 produces on gfx950 (OCP e4m3? saturation?).  Build: hipcc --offload-arch=gfx950 fp8_mfma.hip -o bin/fp8_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

// OCP e4m3fn decode
static float e4m3_dec(uint8_t b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
static uint8_t e4m3_enc(float f) {   // nearest (ties to even by brute force), saturating
  uint8_t best = 0; float bd = 1e30f;
  for (int b = 0; b < 256; ++b) { float v = e4m3_dec((uint8_t)b); if (isnan(v)) continue; float d = fabsf(v - f); if (d < bd || (d == bd && !(b & 1))) { bd = d; best = (uint8_t)b; } }
  return best;
}

__global__ void k_mfma(const v8i* a, const v8i* b, v4f* c, int sa, int sb) {
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa, 0, sb);
  c[threadIdx.x] = acc;
}
__global__ void k_cvt(const float* f, uint8_t* o, int n) {
  int i = threadIdx.x;
  if (2 * i + 1 < n) {
    unsigned r = __builtin_amdgcn_cvt_pk_fp8_f32(f[2 * i], f[2 * i + 1], 0u, false);
    o[2 * i] = r & 0xff; o[2 * i + 1] = (r >> 8) & 0xff;
  }
}

int main() {
  // A[16][128], B[128][16] small exact values
  std::vector<float> A(16 * 128), B(128 * 16);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i * 128 + k] = (float)(((i * 7 + k * 3) % 9) - 4) * 0.5f;
  for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)(((k * 5 + j * 11) % 7) - 3);
  std::vector<double> C(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 128; ++k) s += (double)A[i * 128 + k] * B[k * 16 + j]; C[i * 16 + j] = s; }
  // hypothesis H: lane l holds row/col l%16, k = 32*(l/16) + byte index
  std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 32; ++q) {
    int r = l % 16, k = 32 * (l / 16) + q;
    ha[l * 32 + q] = e4m3_enc(A[r * 128 + k]);
    hb[l * 32 + q] = e4m3_enc(B[k * 16 + r]);
  }
  v8i *da, *db; v4f* dc;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16);
  hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
  for (int sc : {0x7f7f7f7f, 0x7f, 0x80, 0}) {
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, sc, 0x7f7f7f7f);
    float hc[256];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    double maxerr = 0, ratio = 0; int cnt = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      int row = (l >> 4) * 4 + r, col = l & 15;
      double want = C[row * 16 + col], got = hc[l * 4 + r];
      maxerr = fmax(maxerr, fabs(want - got));
      if (fabs(want) > 1) { ratio += got / want; ++cnt; }
    }
    printf("scale_a=0x%08x: max |C - ref| = %g, mean got/want = %g\n", sc, maxerr, ratio / cnt);
  }
  // conversions
  float tf[] = {0.f, 1.f, -1.f, 0.0625f, 0.001953125f, 0.0009765625f, 1.0625f, 1.125f, 1.1875f, 447.f, 448.f, 449.f, 480.f, 1000.f, -1000.f, 1e-8f, 3.3f, 100.f};
  const int n = sizeof(tf) / sizeof(float);
  float* df; uint8_t* dob;
  hipMalloc(&df, sizeof(tf)); hipMalloc(&dob, n);
  hipMemcpy(df, tf, sizeof(tf), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, df, dob, n);
  uint8_t ob[64];
  hipMemcpy(ob, dob, n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt %12g -> 0x%02x = %g   (host OCP e4m3 nearest: 0x%02x = %g)\n", tf[i], ob[i], e4m3_dec(ob[i]), e4m3_enc(tf[i]), e4m3_dec(e4m3_enc(tf[i])));
  return 0;
}
