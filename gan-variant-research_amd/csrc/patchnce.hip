// PatchNCE for one feature layer, forward and backward, with no host synchronisation.
//
// Replaces PatchNCELoss._compute_nce_loss (GAN_Variant1/losses/patchnce_cut.py:42-110): the strided row gather
// feat.view(B,C,HW).permute(0,2,1)[b, ids, :] is a contiguous C-vector per patch in halo-NHWC, F.normalize
// (eps 1e-6) is fused with the gather, the per-image 256xC . Cx256 torch.mm / clamp(+-50) / cross-entropy run
// as one tiled kernel per (image, 16 target rows), and the reference's 4*B+4 host-side isfinite() branches
// (:97,:106) become a device flag per image.  FLOPs are negligible (92 MFLOP per image for all four layers).
// Both contractions (logits = Tn . Sn^T and dTn = dLogits . Sn) run on v_mfma_f32_16x16x4_f32 -- exact fp32 products with
// fp32 accumulation in both precision modes -- with operands streamed from L2 straight into MFMA registers (k index
// permuted identically on both sides so every lane reads 16 contiguous bytes); shapes the MFMA tiling does not cover
// (C not a multiple of 64, P not a multiple of 16) take the scalar-FMA kernels below.
//
// Workspace (floats): Sn[B][P][C] | Tn[B][P][C] | tnorm[B][P] | lse[B][P] | rowloss[B][P] | flag[B] | dX[B][P][C]
#include <stdlib.h>
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
  // one wave per image (a wave reduction, no block-wide barrier per image: 16 images took 14 us of barriers); the images' means are
  // then added in image order by one thread, as before
  __shared__ float sb[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float tot = 0.f;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int nb = min(256, B - b0);
    for (int k = wave; k < nb; k += 4) {
      float s = 0.f;
      for (int i = lane; i < P; i += 64) s += w.rowloss[(int64_t)(b0 + k) * P + i];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
      if (lane == 0) sb[k] = s / (float)P;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 0; k < nb; ++k) {
        const bool ok = isfinite(sb[k]);
        w.flag[b0 + k] = ok ? 1.f : 0.f;
        tot += ok ? sb[k] : 0.f;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tot /= (float)B;
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

// ---------------------------------------------------------------------------------------------- MFMA variants
// Raw similarity sums of target rows i0..i0+15 against the 64 source patches 64*wave .. 64*wave+63 of image b.
// acc[t][r] belongs to row (lane>>4)*4 + r and source column 64*wave + 16*t + (lane&15).  MFMA k-step (q, e) covers channels
// 16q + 4*(lane>>4) + e, so each lane reads a float4 per operand and 16 channels.
__device__ __forceinline__ void nce_logits_mfma(const float* __restrict__ Sn_b, const float* __restrict__ Tn_b, int i0, int P, int C,
                                                f32x4_t (&acc)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const float* pa = Tn_b + (int64_t)min(i0 + fr, P - 1) * C + 4 * fg;
  const float* pb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    pb[t] = Sn_b + (int64_t)min(64 * wave + 16 * t + fr, P - 1) * C + 4 * fg;
    acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // C % 64 == 0 (nce_mfma_ok): four k-steps per trip, all their loads issued before the first MFMA (a rolled loop waits for every step's
  // loads: a chain of C / 16 dependent L2 round trips)
  for (int q = 0; q < C; q += 64) {
    f32x4_t a4[4], b4[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a4[u] = *reinterpret_cast<const f32x4_t*>(pa + q + 16 * u);
#pragma unroll
      for (int t = 0; t < 4; ++t) b4[u][t] = *reinterpret_cast<const f32x4_t*>(pb[t] + q + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][e], b4[u][t][e], acc[t], 0, 0, 0);
  }
}

// row-wise reduction over the 256 columns held as [4 tiles][16 lanes of a lane group][4 waves]; red: [4 waves][16 rows]
template <bool MAX>
__device__ __forceinline__ void nce_row_reduce(float (&v)[4], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const float u = __shfl_xor(v[r], o, 64); v[r] = MAX ? fmaxf(v[r], u) : v[r] + u; }
  __syncthreads();   // red may still be read from the previous reduction
  if (fr == 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 16 + fg * 4 + r] = v[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* q = red + fg * 4 + r;
    v[r] = MAX ? fmaxf(fmaxf(q[0], q[16]), fmaxf(q[32], q[48])) : (q[0] + q[16]) + (q[32] + q[48]);
  }
}

__global__ __launch_bounds__(256) void nce_fwd_mfma_kernel(int P, int C, float inv_t, NceWs w) {
  __shared__ float red[64];
  const int b = blockIdx.y, i0 = blockIdx.x * TI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  f32x4_t acc[4];
  nce_logits_mfma(w.Sn + (int64_t)b * P * C, w.Tn + (int64_t)b * P * C, i0, P, C, acc);
  float lg[4][4], mx[4], se[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) mx[r] = -1e30f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool ok = 64 * wave + 16 * t + fr < P;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lg[t][r] = ok ? fminf(fmaxf(acc[t][r] * inv_t, -50.f), 50.f) : -1e30f;
      mx[r] = fmaxf(mx[r], lg[t][r]);
    }
  }
  nce_row_reduce<true>(mx, red);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    se[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) se[r] += (64 * wave + 16 * t + fr < P) ? expf(lg[t][r] - mx[r]) : 0.f;
  }
  nce_row_reduce<false>(se, red);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + fg * 4 + r;   // the diagonal logit of row i lives in wave i/64, tile (i%64)/16, lane group fg, lane i%16
    if (i < P && (i >> 6) == wave && (i & 15) == fr) {
      const int t = (i & 63) >> 4;
      const float d = t == 0 ? lg[0][r] : t == 1 ? lg[1][r] : t == 2 ? lg[2][r] : lg[3][r];
      const float lse = mx[r] + logf(se[r]);
      w.lse[(int64_t)b * P + i] = lse;
      w.rowloss[(int64_t)b * P + i] = lse - d;
    }
  }
}

constexpr int DP = 260;   // pitch (floats) of the dLogits tile in LDS

__global__ __launch_bounds__(256) void nce_bwd_mfma_kernel(int B, int P, int C, float inv_t, float weight, NceWs w) {
  extern __shared__ float dyn[];
  float* dsh = dyn;              // [TI][DP]  dLogits
  float* tsh = dsh + TI * DP;    // [TI][C]   normalised target rows
  float* gsh = tsh + TI * C;     // [TI][C]   dTn
  const int b = blockIdx.y, i0 = blockIdx.x * TI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const float* Sn_b = w.Sn + (int64_t)b * P * C;
  const float* Tn_b = w.Tn + (int64_t)b * P * C;
  for (int k = threadIdx.x; k < TI * C; k += 256) {
    const int r = k / C, c = k - r * C;
    tsh[k] = (i0 + r < P) ? Tn_b[(int64_t)(i0 + r) * C + c] : 0.f;
  }
  f32x4_t acc[4];
  nce_logits_mfma(Sn_b, Tn_b, i0, P, C, acc);
  const float scale = weight * w.flag[b] * inv_t / ((float)P * (float)B);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = 64 * wave + 16 * t + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + fg * 4 + r;
      const float raw = acc[t][r] * inv_t;
      float d = 0.f;
      if (i < P && j < P) {
        d = expf(fminf(fmaxf(raw, -50.f), 50.f) - w.lse[(int64_t)b * P + i]) - (j == i ? 1.f : 0.f);
        if (raw < -50.f || raw > 50.f) d = 0.f;   // clamp passes no gradient outside [-50, 50]
        d *= scale;
      }
      dsh[(fg * 4 + r) * DP + j] = d;
    }
  }
  __syncthreads();
  // dTn[i][c] = sum_j dLogits[i][j] * Sn[j][c]: wave owns channels wave*C/4 ..., 16 at a time; k = j, same permutation
  const int cw = C >> 2, nct = cw >> 4;          // C % 64 == 0, C <= 256: one to four 16-channel blocks per wave
  {
    f32x4_t g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float* pb = Sn_b + (int64_t)(4 * fg) * C + wave * cw + fr;
    // four k-steps of all the wave's channel blocks per trip, their strided loads (up to 64) issued before the first MFMA: the rolled
    // loops were a chain of (C / 64) x (P / 16) dependent L2 round trips (59 us per launch for 0.27 GFLOP; now 4 trips)
    for (int q = 0; q < P; q += 64) {
      float bv[4][4][4];
      f32x4_t a4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int qq = min(q + 16 * u, P - 16);       // P % 16 == 0; a step past the end re-reads the last one and is not multiplied
        a4[u] = *reinterpret_cast<const f32x4_t*>(dsh + fr * DP + qq + 4 * fg);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < nct) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[k][u][e] = pb[(int64_t)(qq + e) * C + 16 * k];
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < nct) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (q + 16 * u < P) {
#pragma unroll
              for (int e = 0; e < 4; ++e) g[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][e], bv[k][u][e], g[k], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gsh[(fg * 4 + r) * C + wave * cw + 16 * k + fr] = g[k][r];
      }
  }
  __syncthreads();
  // normalisation backward, one wave per four rows (TI = 16 rows, 4 waves): the dot product is a wave reduction -- no block-wide
  // barrier per row (16 x 2 barriers were a third of the launch)
  static_assert(TI == 16, "four rows per wave");
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr, i = i0 + r;
    if (i >= P) break;                      // wave-uniform
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += tsh[r * C + c] * gsh[r * C + c];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 64);
    const float nrm = w.tnorm[(int64_t)b * P + i];
    for (int c = lane; c < C; c += 64) {
      const float gg = gsh[r * C + c];
      w.dX[((int64_t)b * P + i) * C + c] = nrm > 1e-6f ? (gg - tsh[r * C + c] * dot) / nrm : gg / nrm;
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
  // positions holding the same id, found 64 at a time (P <= 256): the first one is the leader and adds all of them, in
  // ascending order of position
  unsigned long long m[4];
  int first = -1;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const int k = ch * 64 + lane;
    m[ch] = __ballot(k < P && ids[k < P ? k : 0] == id);
    if (first < 0 && m[ch]) first = ch * 64 + __builtin_ctzll(m[ch]);
  }
  if (first != i) return;   // wave-uniform
  const int y = id / gt.W, x = id - y * gt.W;
  T* p = reinterpret_cast<T*>(gt.ptr) + gt.pix(b, y, x);
  for (int c = lane; c < C; c += 64) {
    float s = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
      for (unsigned long long mm = m[ch]; mm; mm &= mm - 1) s += w.dX[((int64_t)b * P + ch * 64 + __builtin_ctzll(mm)) * C + c];
    st1<T>(p + c, ld1<T>(p + c) + s);
  }
}

}  // namespace

#define VCHK(v, name) do { if (gan_check_view(v, name)) return -1; } while (0)

extern "C" int64_t gan_patchnce_ws_floats(int B, int P, int C) {
  return 3ll * B * P * C + 3ll * B * P + ((B + 3) / 4) * 4 + 64;
}

// the MFMA tiling: 16-channel k chunks per load, 4 waves x C/4 output channels in 16-wide tiles, 16-patch k chunks
static bool nce_mfma_ok(int P, int C) {
  static int off = -1;
  if (off < 0) { const char* e = getenv("GAN_NO_NCE_MFMA"); off = (e && atoi(e)) ? 1 : 0; }
  return !off && C % 64 == 0 && C <= 256 && P % 16 == 0 && P <= 256;
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
  if (nce_mfma_ok(P, C)) hipLaunchKernelGGL(nce_fwd_mfma_kernel, dim3((P + TI - 1) / TI, B), dim3(256), 0, s, P, C, 1.f / temperature, w);
  else hipLaunchKernelGGL(nce_fwd_kernel, dim3((P + TI - 1) / TI, B), dim3(256), shm, s, P, C, 1.f / temperature, w);
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
  if (nce_mfma_ok(P, C))
    hipLaunchKernelGGL(nce_bwd_mfma_kernel, dim3((P + TI - 1) / TI, B), dim3(256), (size_t)(TI * DP + 2 * TI * C) * sizeof(float), s, B, P, C,
                       1.f / temperature, weight, w);
  else hipLaunchKernelGGL(nce_bwd_kernel, dim3((P + TI - 1) / TI, B), dim3(256), shm, s, B, P, C, 1.f / temperature, weight, w);
  DView vg = to_dview(gtgt);
  GAN_DISPATCH_DTYPE(gtgt->dtype, hipLaunchKernelGGL((nce_scatter_kernel<T>), dim3((B * P + 3) / 4), dim3(256), 0, s, vg, ids, P, C, w);)
  GAN_LAUNCH_CHECK();
  return 0;
}
