// "Range-patch" weight gradient for stride-1 windows with <= 9 taps on narrow maps (the 18 residual 3x3 256->256
// convolutions: 54 of the 76 weight-gradient launches of a CUT step).
//
//   part[s][n][t][c] = sum over the pixels m of split s of  g[m][n] * x[pix(m) + tapoff[t]][c]
//
// Same lesson as conv_patch.hip: the generic kernel (conv_wgrad.hip) spends its time in one barrier per 64-pixel K-step
// and re-stages x once per tap.  Here a block owns an output tile of 128 n x 64 c x ALL taps and walks its pixel range in
// stages of 128 pixels: the g tile (128 x 128 n) and ONE contiguous pixel range of x (128 + window span pixels x 64 c)
// are staged by LDS-DMA, then every tap reads its x fragments from that same range at a shifted row -- one barrier per
// 144 MFMA per wave.  Both operands are reduction-major in memory, so fragments come from ds_read_b64_tr_b16; the MFMA
// k index is permuted (k = 16h + 4*lanegroup + q, identically for both operands) so that the eight rows a half-wave
// reads are CONSECUTIVE pixels: with x rows stored at a 160-byte stride (and the g tile's 32-byte pieces XOR-ed by row & 7) the
// transposed reads are bank-conflict free at ANY tap shift, and a tap costs one address add per read.  4 waves (one per SIMD, 512-register budget) = 2 (n) x 2 (c); a
// wave owns 4 n-tiles x 2 c-tiles x 9 taps = 72 accumulator tiles and prefetches x fragments four taps ahead of their MFMAs.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int KM = 128;            // pixels per stage
constexpr int NB = 128, CB = 64;   // output tile: g channels x x channels (x all taps)
constexpr int RX = 288;            // x pixel rows per stage buffer (>= KM + window span)
constexpr int XROW = 160;          // x rows padded 128 -> 160 bytes: 8 consecutive rows at one 32-byte column hit 8 distinct bank groups
constexpr int GT_BYTES = KM * 256, XP_BYTES = RX * XROW, STAGE_BYTES = GT_BYTES + XP_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES + 256;
constexpr int MAXT = 9;

struct WpArgs {
  const char* x; const char* g; const int32_t* tapoff; float* part;
  int B, HoWo, Wo, spi, per;       // spi: splits per image, per: pixels per split (multiple of KM)
  int Cx, ntaps, N;
  int x_Hp, x_Wp, x_y0, x_x0, x_pix;
  int g_Hp, g_Wp, g_C, g_y0, g_x0;
  int NBLK, CBLK;
};

__device__ __forceinline__ void glds16q(const char* gbase, uint32_t goff, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff),
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// g tile (256-byte rows): a half-wave's transposed read touches 8 consecutive rows at one 32-byte column -> XOR the 32-byte
// piece index (= 16-byte chunk index >> 1) with row & 7
__device__ __forceinline__ int swz_g(int row) { return (row & 7) << 1; }

template <int NT>
__global__ __launch_bounds__(256) void wgrad_patch_kernel(WpArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  int bid = blockIdx.x;
  const int cb = bid % a.CBLK; bid /= a.CBLK;
  const int nb = bid % a.NBLK; bid /= a.NBLK;
  const int sp = bid;                                  // split index: image b = sp / spi, sub-range sp % spi
  const int b = sp / a.spi, sub = sp - b * a.spi;
  const int m_begin = sub * a.per, m_end = min(a.HoWo, m_begin + a.per);
  const int nstage = (m_end - m_begin + KM - 1) / KM;
  const int n0 = nb * NB, c0 = cb * CB;

  // tap offsets as wave-uniform scalars, in bytes of the padded x image
  int toffb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) toffb[t] = __builtin_amdgcn_readfirstlane(a.tapoff[t] / a.Cx) * XROW;

  auto pix_x = [&](int m) { const int ho = m / a.Wo, wo = m - ho * a.Wo; return (b * a.x_Hp + ho + a.x_y0) * a.x_Wp + wo + a.x_x0; };
  auto pix_g = [&](int m) { const int ho = m / a.Wo, wo = m - ho * a.Wo; return (b * a.g_Hp + ho + a.g_y0) * a.g_Wp + wo + a.g_x0; };

  // ---- staging roles
  // g tile by LDS-DMA: row gr + 16*i (i<8), LDS position gp (16-byte chunk of the 256-byte row) holds source chunk gp ^ swz_g(row)
  const int gr = tid >> 4, gp = tid & 15;
  const uint32_t gsrc0 = (uint32_t)((n0 + ((gp ^ swz_g(gr)) << 3)) * 2), gsrc1 = (uint32_t)((n0 + ((gp ^ swz_g(gr + 16)) << 3)) * 2);
  (void)gsrc1;   // rows gr+16*i: (row & 7) == (gr & 7) for every i, so one source chunk serves all eight rows
  const uint32_t x_pixb = (uint32_t)a.Cx * 2u, g_pixb = (uint32_t)a.g_C * 2u;
  // x range through registers into the padded image: row xr + 32*i (i<9), 16-byte chunk xp
  const int xr = tid >> 3, xp = tid & 7;   // 32 rows per pass
  const uint32_t xsrc = (uint32_t)((c0 + xp * 8) * 2);

  auto stage_g = [&](int st, int buf) {
    char* gt = lds + buf * STAGE_BYTES;
    const int m0 = m_begin + st * KM;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int m = m0 + gr + 16 * i;
      m = m < a.HoWo ? m : a.HoWo - 1;   // rows past the split end are zeroed after landing
      glds16q(a.g, (uint32_t)pix_g(m) * g_pixb + gsrc0, gt + wave * 1024 + i * 4096);
    }
  };
  constexpr int XI = RX / 32;   // 9 rows per thread
  auto x_fetch = [&](int P0, int i0, int cnt, u32x4_t (&v)[XI]) {
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (i >= i0 && i < i0 + cnt) {
        int pix = P0 + xr + 32 * i;
        pix = pix < a.x_pix ? pix : a.x_pix - 1;
        v[i] = *reinterpret_cast<const u32x4_t*>(a.x + (size_t)((uint32_t)pix * x_pixb + xsrc));
      }
  };
  auto x_commit = [&](int buf, int i0, int cnt, const u32x4_t (&v)[XI]) {
    char* xpb = lds + buf * STAGE_BYTES + GT_BYTES;
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (i >= i0 && i < i0 + cnt) *reinterpret_cast<u32x4_t*>(xpb + (xr + 32 * i) * XROW + xp * 16) = v[i];
  };

  const int wn = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, p4 = fr & 3;
  // transposed-read geometry: instruction h of k-step ks reads, for lane group fg, pixel rows ks*32 + 16h + 4fg + q
  uint32_t gcolx[4];   // swizzle-ready g column offsets (chunk bits and intra-chunk bits)
#pragma unroll
  for (int i = 0; i < 4; ++i) gcolx[i] = (uint32_t)(((wn * 4 + i) * 16 + p4 * 4) * 2);
  const uint32_t xcol = (uint32_t)((wc * 32 + p4 * 4) * 2);   // first of this wave's two c-tiles; the second is +32 bytes

  f32x4_t acc[4][2][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[i][j][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  u32x4_t nx[XI];
  {   // prologue: stage 0
    x_fetch(pix_x(m_begin), 0, XI, nx);
    stage_g(0, 0);
    x_commit(0, 0, XI, nx);
  }
  for (int st = 0; st < nstage; ++st) {
    __syncthreads();   // stage st is complete in LDS (LDS-DMA drained, x rows written); everyone is done with the other buffer
    char* gt = lds + (st & 1) * STAGE_BYTES;
    const char* xpb = gt + GT_BYTES;
    const int m0 = m_begin + st * KM;
    if (m0 + KM > m_end) {   // tail: pixels past the split end must not contribute -> zero their g rows
      for (int c = tid; c < KM * 16; c += 256)
        if (m0 + (c >> 4) >= m_end) *reinterpret_cast<u32x4_t*>(gt + c * 16) = u32x4_t{0, 0, 0, 0};
      __syncthreads();
    }
    const bool more = st + 1 < nstage;
    const int Pn = more ? pix_x(m_begin + (st + 1) * KM) : 0;
    if (more) { stage_g(st + 1, (st + 1) & 1); x_fetch(Pn, 0, XI, nx); }   // lands while this stage computes; written at k-step 2

    // this lane's first pixel of the stage: (ho, wo) once, then 16-pixel steps without divisions
    const int P0 = pix_x(m0);
    int m = m0 + 4 * fg + q;
    int ho = m / a.Wo, wo = m - ho * a.Wo;
#pragma unroll
    for (int ks = 0; ks < KM / 32; ++ks) {
      uint32_t xa[2];   // byte address of this lane's x row for h = 0 / 1 (tap 0)
      s16x4_t gf[4][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int hoc = ho < a.HoWo / a.Wo ? ho : a.HoWo / a.Wo - 1;   // clamp rows past the image (their g rows are zero)
        const int pix = (b * a.x_Hp + hoc + a.x_y0) * a.x_Wp + wo + a.x_x0;
        xa[h] = (uint32_t)((pix - P0) * XROW) + xcol;
        const int row = ks * 32 + 16 * h + 4 * fg + q;
        const uint32_t rb = (uint32_t)(row * 256), sw = (uint32_t)(swz_g(row) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          gf[i][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(gt + rb + (gcolx[i] ^ sw)));
        wo += 16;
        if (wo >= a.Wo) { wo -= a.Wo; ++ho; }
        if (wo >= a.Wo) { wo -= a.Wo; ++ho; }   // maps narrower than 16 pixels are not routed here (Wo >= 8 checked on the host)
      }
      bf16x8_t av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        av[i] = bf16x8_t{gf[i][0][0], gf[i][0][1], gf[i][0][2], gf[i][0][3], gf[i][1][0], gf[i][1][1], gf[i][1][2], gf[i][1][3]};
      // taps: a ring of x fragments (2 c-tiles each), read RING-1 taps ahead of the MFMAs that consume them
      constexpr int RING = 5;
      s16x4_t xf[RING][2][2];
      auto x_read = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            xf[t % RING][j][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4_t*)(xpb + xa[h] + (uint32_t)(toffb[t] + j * 32)));
      };
#pragma unroll
      for (int t = 0; t < RING - 1 && t < NT; ++t) x_read(t);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t + RING - 1 < NT) x_read(t + RING - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const s16x4_t(&f)[2] = xf[t % RING][j];
          const bf16x8_t bv = {f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv, acc[i][j][t], 0, 0, 0);
        }
      }
      if (more && ks == 2) x_commit((st + 1) & 1, 0, XI, nx);
    }
  }

  // D[row = n (fg*4+e)][col = c (fr)]
  float* part = a.part + (int64_t)sp * a.N * a.ntaps * a.Cx;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + (wn * 4 + i) * 16 + fg * 4 + e;
          part[((int64_t)n * a.ntaps + t) * a.Cx + c0 + (wc * 2 + j) * 16 + fr] = acc[i][j][t][e];
        }
}

}  // namespace

// splits per image the range-patch weight-gradient kernel wants for this problem (0 = descriptor does not qualify).
// The planner sizes `part` for B * spi slabs and sets nsplit = B * spi, variant = 1.
extern "C" int gan_wgrad_patch_splits(const gan_wgrad_desc* d) {
  // EXPERIMENTAL (round 1): correct (parity-tested) but slower than the generic kernel because hipcc spills accumulators in
  // the MFMA loop; opt-in with GAN_WPATCH=1 until the register allocation is fixed.
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("GAN_WPATCH"); enabled = (e && atoi(e)) ? 1 : 0; }
  if (!enabled || !d) return 0;
  if (d->dtype != GAN_BF16 || d->ntaps != MAXT || d->Cx % CB != 0 || d->N % NB != 0 || d->N != d->g_C) return 0;
  if (d->x_sy != 1 || d->x_sx != 1 || d->g_sy != 1 || d->g_sx != 1 || d->max_tapoff <= 0) return 0;
  const int HoWo = d->Ho * d->Wo;
  if (HoWo < KM || d->Wo < 8) return 0;
  const int wraps = (KM - 1) / d->Wo + 1;
  const int jump = d->x_Wp - d->Wo;
  const int span = (KM - 1) + wraps * (jump > 0 ? jump : 0) + d->max_tapoff / d->Cx + 1;
  if (span > RX) return 0;
  const int blocks_per_split = (d->N / NB) * (d->Cx / CB);
  int spi = (256 + d->B * blocks_per_split - 1) / (d->B * blocks_per_split);   // ~one block per CU
  const int max_spi = HoWo / (2 * KM) > 0 ? HoWo / (2 * KM) : 1;
  if (spi > max_spi) spi = max_spi;
  if (spi < 1) spi = 1;
  return spi;
}

int gan_wgrad_patch_launch(const gan_wgrad_desc* d, hipStream_t s) {
  const int spi_want = gan_wgrad_patch_splits(d);
  GAN_CHECK(spi_want > 0 && d->nsplit % d->B == 0, "wgrad: variant=1 but the descriptor does not qualify for the range-patch kernel");
  WpArgs a;
  a.x = (const char*)d->x; a.g = (const char*)d->g; a.tapoff = d->tapoff; a.part = d->part;
  a.B = d->B; a.HoWo = d->Ho * d->Wo; a.Wo = d->Wo; a.spi = d->nsplit / d->B;
  int per = (a.HoWo + a.spi - 1) / a.spi;
  per = (per + KM - 1) / KM * KM;
  a.per = per;
  GAN_CHECK((a.spi - 1) * per < a.HoWo, "wgrad_patch: nsplit=%d leaves empty splits", d->nsplit);
  a.Cx = d->Cx; a.ntaps = d->ntaps; a.N = d->N;
  a.x_Hp = d->x_Hp; a.x_Wp = d->x_Wp; a.x_y0 = d->x_y0; a.x_x0 = d->x_x0; a.x_pix = d->B * d->x_Hp * d->x_Wp;
  a.g_Hp = d->g_Hp; a.g_Wp = d->g_Wp; a.g_C = d->g_C; a.g_y0 = d->g_y0; a.g_x0 = d->g_x0;
  a.NBLK = d->N / NB; a.CBLK = d->Cx / CB;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_patch_kernel<MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
      return gan_set_error(-2, "wgrad_patch: cannot raise the dynamic LDS limit to %d bytes", LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_patch_kernel<MAXT>, dim3(a.NBLK * a.CBLK * d->nsplit), dim3(256), LDS_BYTES, s, a);
  if (hipGetLastError() != hipSuccess) return gan_set_error(-2, "wgrad_patch: launch failed");
  return 0;
}
