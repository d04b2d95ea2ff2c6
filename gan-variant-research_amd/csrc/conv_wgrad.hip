// Weight gradient of the generalised tap convolution: part[s][n][t][c] = sum_m g[m][n] * x[row(m,t)][c].
//
// Replaces aten::convolution_backward's weight branch for every nn.Conv2d / nn.ConvTranspose2d of the reference
// (GAN_Variant1/models/generator_resnet_attn.py:33,48,113,125,146-149,160; discriminator_patchgan.py:27-51).
// The reduction runs over GEMM rows m (pixels), i.e. over the ROW index of both halo-NHWC operands, so both
// MFMA operands are "k-major" in memory.  Tiles are staged row-major by LDS-DMA (global_load_lds_dwordx4) and
// transposed on the way to registers with ds_read_b64_tr_b16 (bf16) -- no transposed copy ever exists in HBM.
// The 16-byte chunk index of each LDS row is XOR-swizzled by ((row&3) | ((row>>3)&1)<<2) << 1 so that the eight
// rows a half-wave's transposed read touches fall in distinct bank groups.  Split-K over pixel ranges writes
// fp32 slabs; gan_wgrad_reduce sums them deterministically into the OIHW (or IOHW) gradient.
#include "common.h"

namespace {

struct WgArgs {
  const char* x; const char* g; const int32_t* tapoff; float* part;
  int M, HoWo, Ho, Wo, Ms;  // Ms = rows per split (multiple of 64)
  int Cx, lgCx, ntaps, Ktot, N;
  int x_Hp, x_Wp, x_y0, x_x0, x_sy, x_sx;
  int g_Hp, g_Wp, g_C, g_y0, g_x0, g_sy, g_sx;
  int JTILES, NTILES;
  int fast;   // every 64-pixel stage lies inside one image and starts at a fixed position of its image rows: uniform stage base + per-thread constants
};

__device__ __forceinline__ void glds16w(const char* gbase, uint32_t goff, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff),
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
__device__ __forceinline__ int swz_h(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

// GCH: 16-byte chunks per LDS row of the g tile (16 = full 256-byte rows, 2/4 = skinny 16-channel tile)
template <typename T, int GCH>
__global__ __launch_bounds__(256) void wgrad_kernel(WgArgs a) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int KM = 64;                       // pixels per K-step
  constexpr int XROWB = 256, GROWB = GCH * 16;  // LDS row bytes
  constexpr int JT = 16 * EPC;                  // columns (t,c) per block tile
  constexpr int NTILE = GCH * EPC;              // g channels per block tile
  constexpr int WAVES_N = NTILE >= 64 ? 2 : 1, WAVES_J = 4 / WAVES_N;
  constexpr int TN = NTILE / 16 / WAVES_N, TJ = JT / 16 / WAVES_J;  // 16x16 tiles per wave
  constexpr int STAGE = KM * (XROWB + GROWB);
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware ids: workgroups are dealt round-robin over the 8 XCDs; the JTILES*NTILES blocks of one pixel split read the
  // same x / dY ranges, so consecutive logical ids go to ONE XCD (x of the 7x7 layers was fetched from HBM 11x over).
  const int grid_q = gridDim.x >> 3, grid_r = gridDim.x & 7, xcd = blockIdx.x & 7;
  const int bid = xcd * grid_q + min(xcd, grid_r) + (int)(blockIdx.x >> 3);
  const int jt = bid % a.JTILES, nt = (bid / a.JTILES) % a.NTILES, sp = bid / (a.JTILES * a.NTILES);
  const int j0 = jt * JT, n0 = nt * NTILE;
  const int m_begin = sp * a.Ms, m_end = min(a.M, m_begin + a.Ms);
  const int nk = (m_end - m_begin + KM - 1) / KM;

  // ---- x tile staging: thread -> (row xr + 16 i, position xp); source chunk = xp ^ swz_h(row)
  const int xr = tid >> 4, xp = tid & 15;
  // rows xr + 16*i have (row&3) and ((row>>3)&1) independent of i, so the swizzle is fixed per thread
  const int xq = xp ^ swz_h(xr);
  int xj = j0 + xq * EPC;
  int xtap = xj >> a.lgCx;
  if (xtap >= a.ntaps) { xtap = a.ntaps - 1; xj = xtap * a.Cx; }  // columns past Ktot: valid address, never stored
  const uint32_t xk = (uint32_t)(a.tapoff[xtap] + (xj & (a.Cx - 1))) * (uint32_t)sizeof(T);
  // ---- g tile staging
  constexpr int GTHR = KM * GCH;  // chunks per stage
  constexpr int GI = (GTHR + 255) / 256;
  const int gr = tid / GCH, gp = tid % GCH;
  constexpr int GRSTEP = 256 / GCH;  // rows covered by one staging instruction of the block for g
  const int gq = (GCH == 16) ? (gp ^ swz_h(gr)) : gp;
  int gn = n0 + gq * EPC;
  if (gn >= a.N) gn = 0;  // clamp to a valid chunk; those output rows are never stored
  const uint32_t gk = (uint32_t)gn * (uint32_t)sizeof(T);

  // row decomposition state for this thread's first row; rows +16*i are derived by stepping
  auto decomp = [&](int m, int& b, int& ho, int& wo) {
    m = m < a.M ? m : a.M - 1;
    b = m / a.HoWo; int r2 = m - b * a.HoWo; ho = r2 / a.Wo; wo = r2 - ho * a.Wo;
  };
  auto advance = [&](int& b, int& ho, int& wo, int d) {
    wo += d;
    while (wo >= a.Wo) { wo -= a.Wo; ++ho; }
    while (ho >= a.Ho) { ho -= a.Ho; ++b; }
    if (b >= (a.M / a.HoWo)) { b = a.M / a.HoWo - 1; ho = a.Ho - 1; wo = a.Wo - 1; }  // clamp past the end
  };
  int xb, xho, xwo, gb, gho, gwo;
  decomp(m_begin + xr, xb, xho, xwo);
  decomp(m_begin + gr, gb, gho, gwo);

  // FAST addressing (a.fast: HoWo % 64 == 0 and Wo a multiple or a divisor of 64 -- every map of the 256^2 / 512^2 networks): a stage's 64
  // pixels lie inside one image at the same place of its rows for every stage, so a thread's row is the wave-UNIFORM position of the
  // stage (scalar registers, stepped by 64 pixels) plus an offset that never changes.  The general path below re-derives (image, row,
  // column) of 4 + GI rows per stage with wrap loops: as many vector instructions as the stage's 32 MFMAs take cycles.
  uint32_t xoff[4], goff[GI];
  int sb = 0, sho = 0, swo = 0;      // uniform: position of the next stage to be issued
  if (a.fast) {
    auto rel = [&](int rho, int sy, int sx, int Wp, int C) {
      const int dho = a.Wo >= KM ? 0 : rho / a.Wo, dwo = rho - dho * a.Wo;
      return (uint32_t)((dho * sy * Wp + dwo * sx) * C) * (uint32_t)sizeof(T);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) xoff[i] = rel(xr + 16 * i, a.x_sy, a.x_sx, a.x_Wp, a.Cx);
#pragma unroll
    for (int i = 0; i < GI; ++i) goff[i] = rel(gr + GRSTEP * i, a.g_sy, a.g_sx, a.g_Wp, a.g_C);
    sb = m_begin / a.HoWo;
    const int r = m_begin - sb * a.HoWo;
    sho = r / a.Wo; swo = r - sho * a.Wo;
  }
  auto stage = [&](int ks, int buf) {
    char* sx = lds + buf * STAGE;
    char* sg = sx + KM * XROWB;
    if (a.fast) {
      const uint32_t xbase = (uint32_t)(((sb * a.x_Hp + sho * a.x_sy + a.x_y0) * a.x_Wp + swo * a.x_sx + a.x_x0) * a.Cx) * (uint32_t)sizeof(T);
      const uint32_t gbase = (uint32_t)(((sb * a.g_Hp + sho * a.g_sy + a.g_y0) * a.g_Wp + swo * a.g_sx + a.g_x0) * a.g_C) * (uint32_t)sizeof(T);
#pragma unroll
      for (int i = 0; i < 4; ++i) glds16w(a.x, xbase + xoff[i] + xk, sx + wave * 1024 + i * 4096);
#pragma unroll
      for (int i = 0; i < GI; ++i)
        if (GTHR >= 256 || tid < GTHR) glds16w(a.g, gbase + goff[i] + gk, sg + wave * 1024 + i * 4096);
      swo += KM;
      while (swo >= a.Wo) { swo -= a.Wo; ++sho; }
      while (sho >= a.Ho) { sho -= a.Ho; ++sb; }
      return;
    }
    int b = xb, ho = xho, wo = xwo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t off = (uint32_t)(((b * a.x_Hp + ho * a.x_sy + a.x_y0) * a.x_Wp + wo * a.x_sx + a.x_x0) * a.Cx) * (uint32_t)sizeof(T);
      glds16w(a.x, off + xk, sx + wave * 1024 + i * 4096);
      advance(b, ho, wo, 16);
    }
    advance(xb, xho, xwo, KM);
    b = gb; ho = gho; wo = gwo;
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      if (GTHR >= 256 || tid < GTHR) {
        const uint32_t off = (uint32_t)(((b * a.g_Hp + ho * a.g_sy + a.g_y0) * a.g_Wp + wo * a.g_sx + a.g_x0) * a.g_C) * (uint32_t)sizeof(T);
        glds16w(a.g, off + gk, sg + wave * 1024 + i * 4096);
      }
      advance(b, ho, wo, GRSTEP);
    }
    advance(gb, gho, gwo, KM);
  };

  const int wn = wave / WAVES_J, wj = wave % WAVES_J;
  const int fi = lane & 15, fg = lane >> 4;
  f32x4_t acc[TN][TJ];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    __syncthreads();
    char* sbuf = lds + (ks & 1) * STAGE;
    const int mrow0 = m_begin + ks * KM;
    if (mrow0 + KM > m_end) {  // tail: rows past the split end must not contribute -> zero them in the g tile
      for (int c = tid; c < GTHR; c += 256) {
        const int row = c / GCH;
        if (mrow0 + row >= m_end) *reinterpret_cast<u32x4_t*>(sbuf + KM * XROWB + c * 16) = u32x4_t{0, 0, 0, 0};
      }
      __syncthreads();
    }
    if (ks + 1 < nk) stage(ks + 1, (ks + 1) & 1);
    const char* sx = sbuf;
    const char* sg = sbuf + KM * XROWB;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kq = 0; kq < KM / 32; ++kq) {
        // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block
        const int q = fi >> 2, p = fi & 3;
        s16x4_t gf[TN][2], xf[TJ][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = kq * 32 + fg * 8 + h * 4 + q;
#pragma unroll
          for (int i = 0; i < TN; ++i) {
            const int col = (wn * TN + i) * 16 + p * 4;  // element column inside the tile
            const int ch = col >> 3;
            const int pos = (GCH == 16) ? (ch ^ swz_h(row)) : ch;
            gf[i][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(sg + row * GROWB + pos * 16 + (col & 7) * 2));
          }
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const int col = (wj * TJ + j) * 16 + p * 4;
            const int pos = (col >> 3) ^ swz_h(row);
            xf[j][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(sx + row * XROWB + pos * 16 + (col & 7) * 2));
          }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            bf16x8_t av = {gf[i][0][0], gf[i][0][1], gf[i][0][2], gf[i][0][3], gf[i][1][0], gf[i][1][1], gf[i][1][2], gf[i][1][3]};
            bf16x8_t bv = {xf[j][0][0], xf[j][0][1], xf[j][0][2], xf[j][0][3], xf[j][1][0], xf[j][1][1], xf[j][1][2], xf[j][1][3]};
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i][j], 0, 0, 0);
          }
      }
    } else {
#pragma unroll 4
      for (int kq = 0; kq < KM / 4; ++kq) {
        const int row = kq * 4 + fg;
        float gf[TN], xf[TJ];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int col = (wn * TN + i) * 16 + fi;
          const int ch = col >> 2;
          const int pos = (GCH == 16) ? (ch ^ swz_h(row)) : ch;
          gf[i] = *reinterpret_cast<const float*>(sg + row * GROWB + pos * 16 + (col & 3) * 4);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = (wj * TJ + j) * 16 + fi;
          const int pos = (col >> 2) ^ swz_h(row);
          xf[j] = *reinterpret_cast<const float*>(sx + row * XROWB + pos * 16 + (col & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // D[row = n (fg*4+e)][col = j (fi)]
  float* part = a.part + (int64_t)sp * a.N * a.Ktot;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int jc = j0 + (wj * TJ + j) * 16 + fi;
      if (jc >= a.Ktot) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + (wn * TN + i) * 16 + fg * 4 + e;
        if (n < a.N) part[(int64_t)n * a.Ktot + jc] = acc[i][j][e];
      }
    }
}

// One thread sums 4 consecutive channels c of one (n, t) over all slabs with 16-byte loads (the slabs are the traffic:
// nsplit x the gradient), then scatters the 4 results into the reference's OIHW / IOHW layout.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, int N, int ntaps, int Cx, int N_real, int C_real,
                                                          int swap, int I2, int KK, const int32_t* __restrict__ khw, float* __restrict__ grad,
                                                          int accumulate) {
  const int c4n = Cx >> 2;
  const int64_t total = (int64_t)N_real * ntaps * c4n;
  const int64_t slab = (int64_t)N * ntaps * Cx;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const int t = (int)((i / c4n) % ntaps);
    const int n = (int)(i / ((int64_t)c4n * ntaps));
    const int k = khw[t];
    if (k < 0 || c >= C_real) continue;
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    const float* p = part + ((int64_t)n * ntaps + t) * Cx + c;
    for (int sp = 0; sp < nsplit; ++sp) s += *reinterpret_cast<const f32x4_t*>(p + sp * slab);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e >= C_real) break;
      const int64_t o = swap ? ((int64_t)(c + e) * I2 + n) * KK + k : ((int64_t)n * I2 + c + e) * KK + k;
      grad[o] = accumulate ? grad[o] + s[e] : s[e];
    }
  }
}

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int Nw, int ntaps, int Cin, int N_real, int C_real,
                                   int swap, int I2, int KK, const int32_t* __restrict__ khw, int layout) {
  const int64_t total = (int64_t)Nw * ntaps * Cin;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int t = (int)((i / Cin) % ntaps);
    const int n = (int)(i / ((int64_t)Cin * ntaps));
    float v = 0.f;
    const int k = khw[t];
    if (n < N_real && c < C_real && k >= 0) v = src[swap ? ((int64_t)c * I2 + n) * KK + k : ((int64_t)n * I2 + c) * KK + k];
    int64_t o = i;
    if (layout == 1) {
      const int kk = t * Cin + c, KB = ntaps * Cin / 32;
      o = ((((int64_t)(n >> 4) * KB + (kk >> 5)) * 64) + ((kk & 31) >> 3) * 16 + (n & 15)) * 8 + (kk & 7);
    }
    st1<T>(dst + o, v);
  }
}

// all operand copies of a network in ONE launch: block -> descriptor by binary search over first_block (a CUT generator has
// ~70 copies of 10^2..10^6 elements each; one launch per copy cost 7.6 us apiece, 0.6 ms per step)
template <typename T>
__device__ __forceinline__ void pack_one(const gan_pack_desc& D, int lb) {
  const float* __restrict__ src = D.src;
  T* __restrict__ dst = reinterpret_cast<T*>(D.dst);
  const int64_t total = (int64_t)D.Nw * D.ntaps * D.Cin;
  for (int64_t i = lb * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)D.nblocks * blockDim.x) {
    const int c = (int)(i % D.Cin);
    const int t = (int)((i / D.Cin) % D.ntaps);
    const int n = (int)(i / ((int64_t)D.Cin * D.ntaps));
    float v = 0.f;
    const int k = D.khw[t];
    if (n < D.N_real && c < D.C_real && k >= 0) v = src[D.swap ? ((int64_t)c * D.I2 + n) * D.KK + k : ((int64_t)n * D.I2 + c) * D.KK + k];
    int64_t o = i;
    if (D.layout == 1) {
      const int kk = t * D.Cin + c, KB = D.ntaps * D.Cin / 32;
      o = ((((int64_t)(n >> 4) * KB + (kk >> 5)) * 64) + ((kk & 31) >> 3) * 16 + (n & 15)) * 8 + (kk & 7);
    }
    st1<T>(dst + o, v);
  }
}
// e4m3 operand copy (dtype GAN_FP8), dst = e4m3(src / *scale).  Layout 1: the fragment-major image of the bf16 kernel with TWO fp8
// channels in every 2-byte slot -- element (n, k) lives at byte 2 * slot(n, k / 2) + (k & 1), slot() as for bf16 over Cin / 2
// "pseudo channels" -- so the range-patch kernel moves exactly the bytes it moves for bf16 and feeds them to the 128-deep fp8 MFMA.
__device__ __forceinline__ void pack_one_fp8(const gan_pack_desc& D, int lb) {
  const float* __restrict__ src = D.src;
  uint8_t* __restrict__ dst = reinterpret_cast<uint8_t*>(D.dst);
  const float inv = 1.f / *D.scale;
  const int64_t total4 = (int64_t)D.Nw * D.ntaps * D.Cin / 4;          // four consecutive channels (one dword) per thread step
  for (int64_t i4 = lb * (int64_t)blockDim.x + threadIdx.x; i4 < total4; i4 += (int64_t)D.nblocks * blockDim.x) {
    const int64_t i = i4 * 4;
    const int c = (int)(i % D.Cin);
    const int t = (int)((i / D.Cin) % D.ntaps);
    const int n = (int)(i / ((int64_t)D.Cin * D.ntaps));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int k = D.khw[t];
    if (n < D.N_real && k >= 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < D.C_real) v[e] = src[D.swap ? ((int64_t)(c + e) * D.I2 + n) * D.KK + k : ((int64_t)n * D.I2 + c + e) * D.KK + k] * inv;
    }
    int64_t o = i;
    if (D.layout == 1) {
      const int Cp = D.Cin / 2, kk = t * Cp + c / 2, KB = D.ntaps * Cp / 32;     // c is a multiple of 4: two whole pseudo slots
      o = 2 * (((((int64_t)(n >> 4) * KB + (kk >> 5)) * 64) + ((kk & 31) >> 3) * 16 + (n & 15)) * 8 + (kk & 7));
    }
    *reinterpret_cast<uint32_t*>(dst + o) = f2e4m3x4(v[0], v[1], v[2], v[3]);
  }
}
// bf16 copies, the common case (every operand copy of the bf16 step): one thread per (row n, 8 consecutive channels, tap); the taps of a group are
// neighbouring threads, so a wave's loads fall into a few source lines, and a thread's result leaves as ONE 16-byte store (8 consecutive kk share
// a fragment lane).  The element-wise loop above stores 2 bytes at a time.
__device__ __forceinline__ void pack_one_bf16x8(const gan_pack_desc& D, int lb) {
  const float* __restrict__ src = D.src;
  bf16_t* __restrict__ dst = reinterpret_cast<bf16_t*>(D.dst);
  const int C8 = D.Cin >> 3;
  const int64_t units = (int64_t)D.Nw * C8 * D.ntaps;
  const int KB = D.ntaps * D.Cin / 32;
  for (int64_t u = lb * (int64_t)blockDim.x + threadIdx.x; u < units; u += (int64_t)D.nblocks * blockDim.x) {
    const int t = (int)(u % D.ntaps);                    // neighbouring threads: the taps of one (n, 8 channels) group -- they share its source lines
    const int64_t r = u / D.ntaps;
    const int c = (int)(r % C8) * 8, n = (int)(r / C8);
    const int k = D.khw[t];
    uint32_t pk[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ce = c + 2 * h + q;
        const bool ok = k >= 0 && n < D.N_real && ce < D.C_real;
        v[q] = ok ? src[(D.swap ? ((int64_t)ce * D.I2 + n) : ((int64_t)n * D.I2 + ce)) * D.KK + k] : 0.f;
      }
      pk[h] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    }
    int64_t o = ((int64_t)n * D.ntaps + t) * D.Cin + c;
    if (D.layout == 1) {
      const int kk = t * D.Cin + c;
      o = ((((int64_t)(n >> 4) * KB + (kk >> 5)) * 64) + ((kk & 31) >> 3) * 16 + (n & 15)) * 8;
    }
    *reinterpret_cast<u32x4_t*>(dst + o) = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
  }
}
__global__ __launch_bounds__(256) void pack_batch_kernel(const gan_pack_desc* __restrict__ d, int n) {
  int lo = 0, hi = n - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  const gan_pack_desc D = d[lo];
  const int lb = blk - D.first_block;
  if (lb >= D.nblocks) return;
  if (D.dtype == GAN_FP8) pack_one_fp8(D, lb);
  else if (D.dtype == GAN_BF16 && (D.Cin & 7) == 0 && ((uintptr_t)D.dst & 15) == 0) pack_one_bf16x8(D, lb);
  else if (D.dtype == GAN_BF16) pack_one<bf16_t>(D, lb);
  else pack_one<float>(D, lb);
}

// column sums of g over logical pixels: stage 1 -> ws[block][C] (16-byte chunk loads, one chunk lane per 4/8 channels),
// stage 2 -> grad (one block per 32 channels, 8 partial lanes each)
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_stage1(DView g, int npix, int per, float* __restrict__ ws) {
  constexpr int N = Chunk<T>::N;
  const int CL = g.C / N, RL = 256 / CL, cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const T* p = reinterpret_cast<const T*>(g.ptr);
  const int HW = g.H * g.W;
  const int p0 = blockIdx.x * per, p1 = min(npix, p0 + per);
  float s[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = 0.f;
  for (int i = p0 + rl; i < p1; i += RL) {
    const int b = i / HW, r = i - b * HW, y = r / g.W, x = r - y * g.W;
    float v[N];
    Chunk<T>::load(p + g.pix(b, y, x) + cl * N, v);
#pragma unroll
    for (int e = 0; e < N; ++e) s[e] += v[e];
  }
  __shared__ float sh[256 * 8];
#pragma unroll
  for (int e = 0; e < N; ++e) sh[threadIdx.x * N + e] = s[e];
  __syncthreads();
  if (rl == 0) {
    for (int r = 1; r < RL; ++r)
#pragma unroll
      for (int e = 0; e < N; ++e) s[e] += sh[(r * CL + cl) * N + e];
#pragma unroll
    for (int e = 0; e < N; ++e) ws[(int64_t)blockIdx.x * g.C + cl * N + e] = s[e];
  }
}
__global__ __launch_bounds__(256) void bias_grad_stage2(const float* __restrict__ ws, int nblk, int C, int N_real, float* __restrict__ grad, int accumulate) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), k0 = threadIdx.x >> 5;
  float s = 0.f;
  if (c < C)
    for (int k = k0; k < nblk; k += 8) s += ws[(int64_t)k * C + c];
  __shared__ float sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (k0 == 0 && c < N_real) {
    for (int k = 1; k < 8; ++k) s += sh[k * 32 + (threadIdx.x & 31)];
    grad[c] = accumulate ? grad[c] + s : s;
  }
}

}  // namespace

int gan_wgrad_patch_launch(const gan_wgrad_desc* d, hipStream_t s);
int gan_wgrad_win7_launch(const gan_wgrad_desc* d, hipStream_t s);

extern "C" int gan_conv_wgrad(const gan_wgrad_desc* d, void* stream) {
  GAN_CHECK(d, "wgrad: null descriptor");
  if (d->variant == 1) return gan_wgrad_patch_launch(d, (hipStream_t)stream);
  if (d->variant == 2) return gan_wgrad_win7_launch(d, (hipStream_t)stream);
  GAN_CHECK(d->dtype == GAN_F32 || d->dtype == GAN_BF16, "wgrad: bad dtype");
  const int es = d->dtype == GAN_F32 ? 4 : 2, epc = 16 / es;
  GAN_CHECK(d->Cx >= 8 && (d->Cx & (d->Cx - 1)) == 0, "wgrad: Cx=%d must be a power of two >= 8", d->Cx);
  GAN_CHECK(d->N >= 8 && d->N % 8 == 0 && d->N <= d->g_C, "wgrad: N=%d g_C=%d", d->N, d->g_C);
  GAN_CHECK(d->ntaps > 0 && d->ntaps <= 128 && d->nsplit > 0, "wgrad: ntaps=%d nsplit=%d", d->ntaps, d->nsplit);
  GAN_CHECK(d->x && d->g && d->part && d->tapoff, "wgrad: null pointer");
  GAN_CHECK(((uintptr_t)d->x % 16) == 0 && ((uintptr_t)d->g % 16) == 0, "wgrad: pointers must be 16-byte aligned");
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  GAN_CHECK(M < (1ll << 31), "wgrad: M too large");
  GAN_CHECK((int64_t)d->B * d->x_Hp * d->x_Wp * d->Cx * es < (1ll << 32) && (int64_t)d->B * d->g_Hp * d->g_Wp * d->g_C * es < (1ll << 32),
            "wgrad: operand exceeds 4 GiB addressing");
  WgArgs a;
  a.x = (const char*)d->x; a.g = (const char*)d->g; a.tapoff = d->tapoff; a.part = d->part;
  a.M = (int)M; a.HoWo = d->Ho * d->Wo; a.Ho = d->Ho; a.Wo = d->Wo;
  int64_t ms = (M + d->nsplit - 1) / d->nsplit;
  ms = (ms + 63) / 64 * 64;
  a.Ms = (int)ms;
  GAN_CHECK((int64_t)(d->nsplit - 1) * ms < M, "wgrad: nsplit=%d leaves empty splits for M=%lld", d->nsplit, (long long)M);
  a.Cx = d->Cx; a.lgCx = __builtin_ctz(d->Cx); a.ntaps = d->ntaps; a.Ktot = d->ntaps * d->Cx; a.N = d->N;
  a.x_Hp = d->x_Hp; a.x_Wp = d->x_Wp; a.x_y0 = d->x_y0; a.x_x0 = d->x_x0; a.x_sy = d->x_sy; a.x_sx = d->x_sx;
  a.g_Hp = d->g_Hp; a.g_Wp = d->g_Wp; a.g_C = d->g_C; a.g_y0 = d->g_y0; a.g_x0 = d->g_x0; a.g_sy = d->g_sy; a.g_sx = d->g_sx;
  static const bool no_fast = [] { const char* e = getenv("GAN_WGRAD_SLOW_ADDR"); return e && atoi(e); }();     // A/B switch
  a.fast = (!no_fast && a.HoWo % 64 == 0 && (a.Wo % 64 == 0 || 64 % a.Wo == 0)) ? 1 : 0;
  const int JT = 16 * epc;
  a.JTILES = (a.Ktot + JT - 1) / JT;
  hipStream_t s = (hipStream_t)stream;
  const bool skinny = d->N <= 16;
  if (d->dtype == GAN_BF16) {
    if (skinny) { a.NTILES = (d->N + 15) / 16; hipLaunchKernelGGL((wgrad_kernel<bf16_t, 2>), dim3(a.JTILES * a.NTILES * d->nsplit), dim3(256), 0, s, a); }
    else { a.NTILES = (d->N + 127) / 128; hipLaunchKernelGGL((wgrad_kernel<bf16_t, 16>), dim3(a.JTILES * a.NTILES * d->nsplit), dim3(256), 0, s, a); }
  } else {
    if (skinny) { a.NTILES = (d->N + 15) / 16; hipLaunchKernelGGL((wgrad_kernel<float, 4>), dim3(a.JTILES * a.NTILES * d->nsplit), dim3(256), 0, s, a); }
    else { a.NTILES = (d->N + 63) / 64; hipLaunchKernelGGL((wgrad_kernel<float, 16>), dim3(a.JTILES * a.NTILES * d->nsplit), dim3(256), 0, s, a); }
  }
  GAN_LAUNCH_CHECK();
  return 0;
}

// Few outputs, many slabs (the 7x7 C = 3 layers: 6 k quads x 256 slabs; the stride-2 layers' 100 slabs): one thread per quad walked its slabs
// one dependent 16-byte load after the other -- 64 us for 26 MB at the END of the backward, where the optimiser waits for it (and 208 us
// beside other work).  Here G = 2, 4 ... 32 neighbouring lanes share a quad: lane l adds slabs l, l + G, ... in ascending order, the lanes are
// combined by an xor butterfly (a fixed association: deterministic, the same on every run).
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_coop_kernel(const float* __restrict__ part, int nsplit, int N, int ntaps, int Cx, int N_real,
                                                               int C_real, int swap, int I2, int KK, const int32_t* __restrict__ khw,
                                                               float* __restrict__ grad, int accumulate) {
  const int c4n = Cx >> 2;
  const int64_t total = (int64_t)N_real * ntaps * c4n;
  const int64_t slab = (int64_t)N * ntaps * Cx;
  const int l = threadIdx.x & (G - 1);
  // the G lanes of a group (aligned, inside one wave) share i: they enter and leave the loop together, which the shuffles below need
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G; i < total; i += (int64_t)gridDim.x * blockDim.x / G) {
    const int c = (int)(i % c4n) * 4;
    const int t = (int)((i / c4n) % ntaps);
    const int n = (int)(i / ((int64_t)c4n * ntaps));
    const int k = khw[t];
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
    if (k >= 0 && c < C_real) {
      const float* p = part + ((int64_t)n * ntaps + t) * Cx + c;
      for (int sp = l; sp < nsplit; sp += G) s += *reinterpret_cast<const f32x4_t*>(p + sp * slab);
    }
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], o, 64);
    }
    if (l == 0 && k >= 0 && c < C_real) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c + e >= C_real) break;
        const int64_t o = swap ? ((int64_t)(c + e) * I2 + n) * KK + k : ((int64_t)n * I2 + c + e) * KK + k;
        grad[o] = accumulate ? grad[o] + s[e] : s[e];
      }
    }
  }
}

extern "C" int gan_wgrad_reduce(const float* part, int nsplit, int N, int ntaps, int Cx, int N_real, int C_real, int swap, int I2,
                                int KK, const int32_t* khw, float* grad, int accumulate, void* stream) {
  GAN_CHECK(part && khw && grad && nsplit > 0 && N_real <= N && C_real <= Cx, "wgrad_reduce: bad arguments");
  GAN_CHECK(Cx % 4 == 0 && ((uintptr_t)part % 16) == 0, "wgrad_reduce: Cx must be a multiple of 4 and part 16-byte aligned");
  const int64_t total = (int64_t)N_real * ntaps * (Cx / 4);
  // lanes per output quad: enough threads to cover the load latency (~64 k), at most one lane per 4 slabs
  static const int coop_env = [] { const char* e = getenv("GAN_WGRAD_REDUCE_COOP"); return e ? atoi(e) : -1; }();    // A/B: 0 = off, else forced G
  int G = 1;
  while (G < 32 && total * G < 65536 && G * 8 <= nsplit) G *= 2;
  if (coop_env >= 0) G = coop_env <= 1 ? 1 : coop_env >= 32 ? 32 : coop_env >= 16 ? 16 : coop_env >= 8 ? 8 : coop_env >= 4 ? 4 : 2;
  const int64_t thr = total * G;
  const int grid = (int)((thr + 255) / 256 < 8192 ? (thr + 255) / 256 : 8192);
#define GAN_REDUCE_COOP(GG) hipLaunchKernelGGL(wgrad_reduce_coop_kernel<GG>, dim3(grid), dim3(256), 0, (hipStream_t)stream, part, nsplit, N, ntaps, Cx, \
                                               N_real, C_real, swap, I2, KK, khw, grad, accumulate)
  if (G == 1) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, part, nsplit, N, ntaps, Cx, N_real, C_real, swap,
                                 I2, KK, khw, grad, accumulate);
  else if (G == 2) GAN_REDUCE_COOP(2);
  else if (G == 4) GAN_REDUCE_COOP(4);
  else if (G == 8) GAN_REDUCE_COOP(8);
  else if (G == 16) GAN_REDUCE_COOP(16);
  else GAN_REDUCE_COOP(32);
#undef GAN_REDUCE_COOP
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_pack_weight(const float* src, void* dst, int dtype, int Nw, int ntaps, int Cin, int N_real, int C_real, int swap,
                               int I2, int KK, const int32_t* khw, int layout, void* stream) {
  GAN_CHECK(src && dst && khw && N_real <= Nw && C_real <= Cin, "pack_weight: bad arguments");
  GAN_CHECK(dtype == GAN_F32 || dtype == GAN_BF16, "pack_weight: dtype %d (fp8 copies need a scale: gan_pack_weight_batch)", dtype);
  GAN_CHECK(layout == 0 || (layout == 1 && Nw % 16 == 0 && (ntaps * Cin) % 32 == 0), "pack_weight: bad layout %d", layout);
  const int64_t total = (int64_t)Nw * ntaps * Cin;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  GAN_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, Nw,
                                               ntaps, Cin, N_real, C_real, swap, I2, KK, khw, layout);)
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_pack_weight_batch(const gan_pack_desc* descs, int n, int total_blocks, void* stream) {
  GAN_CHECK(descs && n > 0 && total_blocks > 0, "pack_weight_batch: bad arguments");
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs, n);
  GAN_LAUNCH_CHECK();
  return 0;
}

extern "C" int gan_bias_grad(const gan_view* g, int N_real, float* grad, int accumulate, float* ws, void* stream) {
  GAN_CHECK(gan_check_view(g, "bias_grad.g") == 0, "%s", gan_last_error());
  const int epc = g->dtype == GAN_F32 ? 4 : 8, cl = g->C / epc;
  GAN_CHECK(cl <= 256 && (cl & (cl - 1)) == 0 && N_real <= g->C && ws && grad, "bias_grad: unsupported C=%d", g->C);
  const int npix = g->B * g->H * g->W;
  // ws: fp32 >= 256*max(C,256) floats -> at most min(1024, 65536/C) blocks
  int nblk = (int)(((int64_t)npix * cl + 2047) / 2048);
  const int cap = 65536 / g->C < 1024 ? 65536 / g->C : 1024;
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  const int per = (npix + nblk - 1) / nblk;
  DView dv = to_dview(g);
  GAN_DISPATCH_DTYPE(g->dtype, hipLaunchKernelGGL((bias_grad_stage1<T>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, dv, npix, per, ws);)
  hipLaunchKernelGGL(bias_grad_stage2, dim3((g->C + 31) / 32), dim3(256), 0, (hipStream_t)stream, ws, nblk, g->C, N_real, grad, accumulate);
  GAN_LAUNCH_CHECK();
  return 0;
}
