#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* A, const float* B, float* D) {   // A[16][4], B[4][16], D[16][16]
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int e = 0; e < 4; ++e) D[((l >> 4) * 4 + e) * 16 + (l & 15)] = c[e];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = (float)(i * 7 + k * 3 + 1);
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = (float)((k + 1) * 5 - j * 2);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD; hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256; ++i) if (hD[i] != ref[i]) ++bad;
  printf("mfma_f32_16x16x4: mismatches %d of 256; D[1][2]=%g ref %g; D[5][9]=%g ref %g\n", bad, hD[18], ref[18], hD[89], ref[89]);
  return 0;
}
