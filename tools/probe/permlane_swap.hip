// Probe: lane mapping of v_permlane16_swap_b32 (gfx950).  r = swap(a, b): which lane's a / b ends up in r[0] / r[1]?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 4) printf("lane %2d: r0 %3u r1 %3u\n", l, h[l], h[64 + l]);
  return 0;
}
