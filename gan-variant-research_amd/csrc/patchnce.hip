// PatchNCE for one feature layer, forward and backward, with no host synchronisation.
//
// Replaces PatchNCELoss._compute_nce_loss (GAN_Variant1/losses/patchnce_cut.py:42-110): the strided row gather
// feat.view(B,C,HW).permute(0,2,1)[b, ids, :] is a contiguous C-vector per patch in halo-NHWC, F.normalize
// (eps 1e-6) is fused with the gather, the per-image 256xC . Cx256 torch.mm / clamp(+-50) / cross-entropy run
// as one tiled kernel per (image, 16 target rows), and the reference's 4*B+4 host-side isfinite() branches
// (:97,:106) become a device flag per image.  FLOPs are negligible (92 MFLOP per image for all four layers),
// so this is plain fp32 FMA (exact fp32 semantics in both precisions), latency-bound by design.
//
// Workspace (floats): Sn[B][P][C] | Tn[B][P][C] | tnorm[B][P] | lse[B][P] | rowloss[B][P] | flag[B] | dX[B][P][C]
#include "common.h"

namespace {

struct NceWs {
  float *Sn, *Tn, *tnorm, *lse, *rowloss, *flag, *dX;
};
__host__ __device__ inline NceWs carve(float* ws, int B, int P, int C) {
  NceWs w;
  const int64_t bpc = (int64_t)B * P * C, bp = (int64_t)B * P;
  w.Sn = ws; w.Tn = w.Sn + bpc; w.tnorm = w.Tn + bpc; w.lse = w.tnorm + bp; w.rowloss = w.lse + bp;
  w.flag = w.rowloss + bp; w.dX = w.flag + ((B + 3) / 4) * 4;
  return w;
}

// one wave per (b, patch): gather the C-vector, L2-normalise (x / max(|x|, eps))
template <typename T>
__global__ __launch_bounds__(256) void nce_gather_kernel(DView src, DView tgt, int has_src, const int32_t* __restrict__ ids, int P, int C,
                                                        NceWs w) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= tgt.B * P) return;
  const int b = row / P, i = row - b * P;
  const int id = ids[i], y = id / tgt.W, x = id - y * tgt.W;
  for (int which = has_src ? 0 : 1; which < 2; ++which) {
    const DView& v = which ? tgt : src;
    const T* p = reinterpret_cast<const T*>(v.ptr) + v.pix(b, y, x);
    float vals[8];  // C <= 512
    float ss = 0.f;
    int k = 0;
    for (int c = lane; c < C; c += 64, ++k) { vals[k] = ld1<T>(p + c); ss += vals[k] * vals[k]; }
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-6f);
    float* o = (which ? w.Tn : w.Sn) + (int64_t)row * C;
    k = 0;
    for (int c = lane; c < C; c += 64, ++k) o[c] = vals[k] / nrm;
    if (which && lane == 0) w.tnorm[row] = nrm;
  }
}

constexpr int TI = 16;  // target rows per block

// computes, for rows i0..i0+15 of image b, the clamped logits against every source patch j (thread j).
// lg[r] = logit(i0+r, j), raw[r] = the unclamped value.  S is streamed through LDS in 32-channel slabs.
__device__ __forceinline__ void nce_logits(const float* __restrict__ Sn_b, const float* __restrict__ Tn_b, int i0, int P, int C, float inv_t,
                                           float* tsh /* [TI][C] */, float* ssh /* [256][33] */, float* lg, float* raw) {
  const int j = threadIdx.x;
  for (int k = threadIdx.x; k < TI * C; k += 256) {
    const int r = k / C, c = k - r * C;
    tsh[k] = (i0 + r < P) ? Tn_b[(int64_t)(i0 + r) * C + c] : 0.f;
  }
  float acc[TI];
#pragma unroll
  for (int r = 0; r < TI; ++r) acc[r] = 0.f;
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();
    for (int k = threadIdx.x; k < P * 32; k += 256) {
      const int jj = k >> 5, cc = k & 31;
      ssh[jj * 33 + cc] = (c0 + cc < C) ? Sn_b[(int64_t)jj * C + c0 + cc] : 0.f;
    }
    __syncthreads();
    if (j < P) {
      for (int cc = 0; cc < 32 && c0 + cc < C; ++cc) {
        const float s = ssh[j * 33 + cc];
#pragma unroll
        for (int r = 0; r < TI; ++r) acc[r] += tsh[r * C + c0 + cc] * s;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < TI; ++r) {
    raw[r] = acc[r] * inv_t;
    lg[r] = fminf(fmaxf(raw[r], -50.f), 50.f);
  }
}

__device__ __forceinline__ float block_max(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void nce_fwd_kernel(int P, int C, float inv_t, NceWs w) {
  extern __shared__ float dyn[];
  float* tsh = dyn;            // TI*C
  float* ssh = tsh + TI * C;   // 256*33
  __shared__ float red[16];
  const int b = blockIdx.y, i0 = blockIdx.x * TI, j = threadIdx.x;
  float lg[TI], raw[TI];
  nce_logits(w.Sn + (int64_t)b * P * C, w.Tn + (int64_t)b * P * C, i0, P, C, inv_t, tsh, ssh, lg, raw);
  for (int r = 0; r < TI; ++r) {
    const int i = i0 + r;
    if (i >= P) break;  // uniform
    const float mx = block_max(j < P ? lg[r] : -1e30f, red);
    const float se = block_sum(j < P ? expf(lg[r] - mx) : 0.f, red);
    const float lse = mx + logf(se);
    if (j == i) { w.lse[(int64_t)b * P + i] = lse; w.rowloss[(int64_t)b * P + i] = lse - lg[r]; }
  }
}

// per image: mean of row losses; non-finite -> 0 with flag 0 (patchnce_cut.py:97-99); *loss += weight * mean_b
__global__ __launch_bounds__(256) void nce_finalize_kernel(int B, int P, float weight, NceWs w, float* __restrict__ loss) {
  __shared__ float sh[16];
  float tot = 0.f;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += w.rowloss[(int64_t)b * P + i];
    s = block_sum(s, sh) / (float)P;
    const bool ok = isfinite(s);
    if (threadIdx.x == 0) w.flag[b] = ok ? 1.f : 0.f;
    tot += ok ? s : 0.f;
  }
  tot /= (float)B;
  if (threadIdx.x == 0) {
    if (!isfinite(tot)) tot = 0.f;  // :106-108
    *loss += weight * tot;
  }
}

// dX[b][i][:] = d loss / d tgt_row, through softmax-CE, clamp, 1/T, and the normalisation
__global__ __launch_bounds__(256) void nce_bwd_kernel(int B, int P, int C, float inv_t, float weight, NceWs w) {
  extern __shared__ float dyn[];
  float* tsh = dyn;               // TI*C
  float* ssh = tsh + TI * C;      // 256*33
  __shared__ float dl[TI * 256];
  __shared__ float red[16];
  const int b = blockIdx.y, i0 = blockIdx.x * TI, j = threadIdx.x;
  const float* Sn_b = w.Sn + (int64_t)b * P * C;
  float lg[TI], raw[TI];
  nce_logits(Sn_b, w.Tn + (int64_t)b * P * C, i0, P, C, inv_t, tsh, ssh, lg, raw);
  const float scale = weight * w.flag[b] * inv_t / ((float)P * (float)B);
#pragma unroll
  for (int r = 0; r < TI; ++r) {
    const int i = i0 + r;
    float d = 0.f;
    if (i < P && j < P) {
      d = expf(lg[r] - w.lse[(int64_t)b * P + i]) - (j == i ? 1.f : 0.f);
      if (raw[r] < -50.f || raw[r] > 50.f) d = 0.f;  // clamp passes no gradient outside [-50, 50]
      d *= scale;
    }
    dl[r * 256 + j] = d;
  }
  __syncthreads();
  // thread <-> channel(s): dTn[r][c] = sum_j dl[r][j] * Sn[j][c]
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc[TI];
#pragma unroll
    for (int r = 0; r < TI; ++r) acc[r] = 0.f;
    for (int jj = 0; jj < P; ++jj) {
      const float s = Sn_b[(int64_t)jj * C + c];
#pragma unroll
      for (int r = 0; r < TI; ++r) acc[r] += dl[r * 256 + jj] * s;
    }
#pragma unroll
    for (int r = 0; r < TI; ++r) ssh[r * C + c] = acc[r];  // reuse ssh as dTn[TI][C] (TI*C <= 256*33)
  }
  __syncthreads();
  for (int r = 0; r < TI; ++r) {
    const int i = i0 + r;
    if (i >= P) break;
    float dot = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) dot += tsh[r * C + c] * ssh[r * C + c];
    dot = block_sum(dot, red);
    const float nrm = w.tnorm[(int64_t)b * P + i];
    for (int c = threadIdx.x; c < C; c += 256) {
      // x/max(|x|,eps): for |x| >= eps the Jacobian is (I - t t^T)/|x|; below eps it is I/eps
      const float g = ssh[r * C + c];
      w.dX[((int64_t)b * P + i) * C + c] = nrm > 1e-6f ? (g - tsh[r * C + c] * dot) / nrm : g / nrm;
    }
  }
}

// one wave per (b, patch): the first occurrence of each id adds the summed rows of all its duplicates
template <typename T>
__global__ __launch_bounds__(256) void nce_scatter_kernel(DView gt, const int32_t* __restrict__ ids, int P, int C, NceWs w) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= gt.B * P) return;
  const int b = row / P, i = row - b * P;
  const int id = ids[i];
  for (int k = 0; k < i; ++k) if (ids[k] == id) return;  // not the leader (wave-uniform)
  const int y = id / gt.W, x = id - y * gt.W;
  T* p = reinterpret_cast<T*>(gt.ptr) + gt.pix(b, y, x);
  for (int c = lane; c < C; c += 64) {
    float s = 0.f;
    for (int k = i; k < P; ++k) if (ids[k] == id) s += w.dX[((int64_t)b * P + k) * C + c];
    st1<T>(p + c, ld1<T>(p + c) + s);
  }
}

}  // namespace

#define VCHK(v, name) do { if (gan_check_view(v, name)) return -1; } while (0)

extern "C" int64_t gan_patchnce_ws_floats(int B, int P, int C) {
  return 3ll * B * P * C + 3ll * B * P + ((B + 3) / 4) * 4 + 64;
}

static int nce_check(const gan_view* t, int P, int C) {
  GAN_CHECK(P > 0 && P <= 256, "patchnce: P=%d must be in 1..256", P);
  GAN_CHECK(C > 0 && C <= 512 && C <= t->C, "patchnce: C=%d unsupported", C);
  GAN_CHECK(TI * C <= 256 * 33, "patchnce: C too large for the LDS tile");
  return 0;
}

extern "C" int gan_patchnce_fwd(const gan_view* src, const gan_view* tgt, const int32_t* ids, int P, int C, float temperature, float weight,
                                float* loss, float* ws, void* stream) {
  VCHK(src, "patchnce.src"); VCHK(tgt, "patchnce.tgt");
  GAN_CHECK(src->B == tgt->B && src->H == tgt->H && src->W == tgt->W && src->dtype == tgt->dtype && ids && loss && ws, "patchnce: src/tgt mismatch");
  if (nce_check(tgt, P, C)) return -1;
  const int B = tgt->B;
  NceWs w = carve(ws, B, P, C);
  hipStream_t s = (hipStream_t)stream;
  DView vs = to_dview(src), vt = to_dview(tgt);
  GAN_DISPATCH_DTYPE(tgt->dtype, hipLaunchKernelGGL((nce_gather_kernel<T>), dim3((B * P + 3) / 4), dim3(256), 0, s, vs, vt, 1, ids, P, C, w);)
  const size_t shm = (size_t)(TI * C + 256 * 33) * sizeof(float);
  hipLaunchKernelGGL(nce_fwd_kernel, dim3((P + TI - 1) / TI, B), dim3(256), shm, s, P, C, 1.f / temperature, w);
  hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(256), 0, s, B, P, weight, w, loss);
  GAN_LAUNCH_CHECK();
  return 0;
}

// must follow gan_patchnce_fwd with the same ws (uses Sn, Tn, tnorm, lse, flag)
extern "C" int gan_patchnce_bwd(const gan_view* tgt, const int32_t* ids, int P, int C, float temperature, float weight, const gan_view* gtgt,
                                float* ws, void* stream) {
  VCHK(tgt, "patchnce.tgt"); VCHK(gtgt, "patchnce.gtgt");
  GAN_CHECK(gtgt->B == tgt->B && gtgt->H == tgt->H && gtgt->W == tgt->W && gtgt->dtype == tgt->dtype && ids && ws, "patchnce: gtgt mismatch");
  if (nce_check(tgt, P, C)) return -1;
  GAN_CHECK(C <= gtgt->C, "patchnce: gtgt has fewer channels than C");
  const int B = tgt->B;
  NceWs w = carve(ws, B, P, C);
  hipStream_t s = (hipStream_t)stream;
  const size_t shm = (size_t)(TI * C + 256 * 33) * sizeof(float);
  hipLaunchKernelGGL(nce_bwd_kernel, dim3((P + TI - 1) / TI, B), dim3(256), shm, s, B, P, C, 1.f / temperature, weight, w);
  DView vg = to_dview(gtgt);
  GAN_DISPATCH_DTYPE(gtgt->dtype, hipLaunchKernelGGL((nce_scatter_kernel<T>), dim3((B * P + 3) / 4), dim3(256), 0, s, vg, ids, P, C, w);)
  GAN_LAUNCH_CHECK();
  return 0;
}
