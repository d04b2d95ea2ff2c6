// "Range-patch" weight gradient for stride-1 3x3 windows on maps whose width divides 128 (the 18 residual 3x3 256->256
// convolutions: 54 of the 76 weight-gradient launches of a CUT step).
//
//   part[s][n][t][c] = sum over the pixels m of split s of  g[m][n] * x[pix(m) + tapoff[t]][c]
//
// Same lesson as conv_patch.hip: the generic kernel (conv_wgrad.hip) pays one block-wide barrier per 64-pixel K-step and
// re-stages x once per tap.  Here a block owns an output tile of 128 n x 64 c x ALL 9 taps and walks its pixel range in
// stages of 128 pixels (= 128/Wo whole image rows): the g tile (128 pixels x 128 n) and the x window (128/Wo + 2 image rows
// x 64 c) are staged once by LDS-DMA and every tap reads its x fragments from that window at a shifted position -- one
// barrier per 144 MFMA per wave.  Both operands are reduction-major in memory, so fragments come from ds_read_b64_tr_b16.
//  * The MFMA k index is permuted (k = 16h + 4*lanegroup + q, identically for both operands) so that the eight rows a
//    half-wave reads are CONSECUTIVE pixels; the 32-byte piece index is XOR-ed by row bits (g: row&7, x: (row>>1)&3), which
//    makes the transposed reads bank-conflict free at ANY pixel shift.
//  * The x window is stored as a 2-D image whose row pitch is padded to a multiple of 8 pixels: the vertical part of a tap
//    offset then never changes the swizzle bits and becomes an immediate offset of the ds_read; only the three horizontal
//    shifts need their own (precomputed) swizzled address -> ~1 vector ALU op per transposed read.
//  * 8 waves = 2 (n) x 4 (c); a wave owns 4 n-tiles x 1 c-tile x 9 taps = 36 accumulator tiles (144 registers).
#include <stdlib.h>
#include <atomic>
#include "common.h"

namespace {

constexpr int KM = 128;            // pixels per stage
constexpr int NB = 128, CB = 64;   // output tile: g channels x x channels (x all taps)
constexpr int RX = 320;            // x window rows (pixels incl. pitch padding) per stage buffer
constexpr int GT_BYTES = KM * 256, XP_BYTES = RX * 128, STAGE_BYTES = GT_BYTES + XP_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
constexpr int NT = 9;
// RING variant (maps exactly KM = 128 pixels wide: the residual layers of 512x512 images): a stage is ONE image row, whose window
// (3 padded rows of 136 pixels) would not fit twice; consecutive stages share two of their three rows, so the rows live in a ring of
// four slots (image row y -> slot y & 3) and every stage fetches only its newest row -- a third of the window traffic, 135 KB of LDS.
constexpr int RING_SLOTS = 4, RING_PITCH = 136;
constexpr int LDS_BYTES_RING = 2 * GT_BYTES + RING_SLOTS * RING_PITCH * 128;

struct WpArgs {
  const char* x; const char* g; float* part;
  int B, HoWo, Wo, lgWo, spi, per;   // spi: splits per image, per: pixels per split (multiple of KM)
  int ipb;                           // images per split (> 1: small maps, spi = 1 -- a block accumulates over ipb whole images)
  int Cx, N, pitch, nrows;           // pitch: padded window row pitch (pixels, multiple of 8); nrows: image rows per window
  int x_Hp, x_Wp, x_y0, x_x0;        // x_y0/x_x0: position of tap (0,0) of output pixel (0,0) inside the padded image
  int g_Hp, g_Wp, g_C, g_y0, g_x0;
  int NBLK, CBLK;
};

__device__ __forceinline__ void glds16q(const char* gbase, uint32_t goff, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff),
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <bool RING>
__global__ __launch_bounds__(512) void wgrad_patch_kernel(WpArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware: the NBLK*CBLK blocks of one pixel split share its dY tile / x window; workgroups b and b+8 share an XCD, so
  // hand consecutive ids to one XCD (the split's operands are then fetched into one L2 once)
  int bid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int cb = bid % a.CBLK; bid /= a.CBLK;
  const int nb = bid % a.NBLK; bid /= a.NBLK;
  const int sp = bid;                                  // split index: image b0 = sp / spi, sub-range sp % spi -- or ipb whole images from sp * ipb
  const int b0 = a.ipb > 1 ? sp * a.ipb : sp / a.spi, sub = a.ipb > 1 ? 0 : sp - b0 * a.spi;
  const int m_begin = sub * a.per, m_end = min(a.HoWo, m_begin + a.per);
  const int nst_img = (m_end - m_begin + KM - 1) / KM;   // stages per image
  const int nstage = nst_img * a.ipb;                    // the stage pipeline runs across the images of the split
  const int n0 = nb * NB, c0 = cb * CB;

  // ---- staging roles (LDS-DMA, lane-linear images, XOR applied on the SOURCE chunk)
  // g tile: 128 rows x 256 B; row gr + 32*i (i<4); position gp holds source chunk gp ^ ((row & 7) << 1)
  const int gr = tid >> 4, gp = tid & 15;
  const uint32_t gsrc = (uint32_t)((n0 + ((gp ^ ((gr & 7) << 1)) << 3)) * 2);
  // x window: RX rows x 128 B; row xr + 64*i (i<5); position xp holds source chunk with piece (xp>>1) ^ ((row>>1)&3)
  const int xr = tid >> 3, xp = tid & 7;
  const uint32_t xsrc = (uint32_t)((c0 + ((((xp >> 1) ^ ((xr >> 1) & 3)) << 1 | (xp & 1)) << 3)) * 2);
  const uint32_t x_pixb = (uint32_t)a.Cx * 2u, g_pixb = (uint32_t)a.g_C * 2u;

  char* const ring = lds + 2 * GT_BYTES;                 // RING: [RING_SLOTS][pitch][128 B] behind the two g tiles
  const uint32_t ring_rowb = (uint32_t)(a.pitch * 128);
  const int b = b0;                                      // RING variant: one image per split
  auto load_row = [&](int iy) {                          // RING: one padded image row -> its slot
    char* slot = ring + (uint32_t)(iy & (RING_SLOTS - 1)) * ring_rowb;
    const int iyc = iy < a.x_Hp ? iy : a.x_Hp - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int row = xr + 64 * i;
      if (row < a.pitch) {                                 // pitch = 136: the third instruction runs in wave 0 only (rows 128..135)
        int ix = row + a.x_x0;
        ix = ix < a.x_Wp ? ix : a.x_Wp - 1;
        glds16q(a.x, (uint32_t)((b * a.x_Hp + iyc) * a.x_Wp + ix) * x_pixb + xsrc, slot + wave * 1024 + i * 8192);
      }
    }
  };
  auto stage = [&](int st, int buf) {
    char* gt = lds + buf * (RING ? GT_BYTES : STAGE_BYTES);
    char* xw = gt + GT_BYTES;
    const int img = st / nst_img, b = b0 + img;
    const int m0 = m_begin + (st - img * nst_img) * KM, ho0 = m0 >> a.lgWo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + gr + 32 * i;
      m = m < a.HoWo ? m : a.HoWo - 1;   // rows past the split end are zeroed after landing
      const int ho = m >> a.lgWo, wo = m & (a.Wo - 1);
      glds16q(a.g, (uint32_t)((b * a.g_Hp + ho + a.g_y0) * a.g_Wp + wo + a.g_x0) * g_pixb + gsrc, gt + wave * 1024 + i * 8192);
    }
    if constexpr (RING) { load_row(ho0 + 2 + a.x_y0); return; }     // the newest of the stage's three rows
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = xr + 64 * i;                      // window row = image row wr, column wcol
      if (row < a.nrows * a.pitch) {
        const int wr = row / a.pitch, wcol = row - wr * a.pitch;
        int iy = ho0 + wr + a.x_y0, ix = wcol + a.x_x0;   // pitch padding and rows past the image read a valid (unused) pixel
        iy = iy < a.x_Hp ? iy : a.x_Hp - 1;
        ix = ix < a.x_Wp ? ix : a.x_Wp - 1;
        glds16q(a.x, (uint32_t)((b * a.x_Hp + iy) * a.x_Wp + ix) * x_pixb + xsrc, xw + wave * 1024 + i * 8192);
      }
    }
  };

  const int wn = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4, q = fr >> 2, p4 = fr & 3;
  // transposed-read geometry: instruction h of k-step ks reads, for lane group fg, pixels ks*32 + 16h + 4fg + q
  uint32_t gcol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) gcol[i] = (uint32_t)(((wn * 4 + i) * 16 + p4 * 4) * 2);
  const uint32_t xcol = (uint32_t)((wc * 16 + p4 * 4) * 2);
  const uint32_t rowb = (uint32_t)(a.pitch * 128);   // bytes per window image row: a multiple of 1024, never touches the swizzle bits
  // Per-lane base addresses (k-step 0, h = 0).  Every other (k-step, h) adds a wave-uniform byte offset: 16h + 32ks is a
  // multiple of 8 pixels (swizzle bits unchanged) and, for Wo >= 16, never carries into the lane's 4fg+q part.
  const int kl = 4 * fg + q;
  uint32_t gbase[4], xbase[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) gbase[i] = (uint32_t)(kl * 256) + (gcol[i] ^ (uint32_t)((kl & 7) << 5));
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int r = kl + dx;
    xbase[dx] = (uint32_t)(r * 128) + (xcol ^ (uint32_t)((r & 6) << 4));
  }

  f32x4_t acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if constexpr (RING) { const int iy0 = (m_begin >> a.lgWo) + a.x_y0; load_row(iy0); load_row(iy0 + 1); }
  stage(0, 0);
  for (int st = 0; st < nstage; ++st) {
    __syncthreads();   // stage st landed (LDS-DMA drained + barrier); everyone is done with the other buffer
    char* gt = lds + (st & 1) * (RING ? GT_BYTES : STAGE_BYTES);
    const char* xw = RING ? ring : gt + GT_BYTES;
    const int m0 = m_begin + (st % nst_img) * KM;
    uint32_t so[3] = {0u, 0u, 0u};                        // RING: byte offset of the slot of tap row dy
    if constexpr (RING) {
      const int iy0 = (m0 >> a.lgWo) + a.x_y0;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) so[dy] = (uint32_t)((iy0 + dy) & (RING_SLOTS - 1)) * ring_rowb;
    }
    if (m0 + KM > m_end) {   // tail: pixels past the split end must not contribute -> zero their g rows
      for (int c = tid; c < KM * 16; c += 512)
        if (m0 + (c >> 4) >= m_end) *reinterpret_cast<u32x4_t*>(gt + c * 16) = u32x4_t{0, 0, 0, 0};
      __syncthreads();
    }
    if (st + 1 < nstage) stage(st + 1, (st + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < KM / 32; ++ks) {
      // keep the seven base addresses opaque so that the 72 derived addresses are recomputed (1 add each), not hoisted and spilled
      asm volatile("" : "+v"(gbase[0]), "+v"(gbase[1]), "+v"(gbase[2]), "+v"(gbase[3]), "+v"(xbase[0]), "+v"(xbase[1]), "+v"(xbase[2]));
      s16x4_t gf[4][2];
      uint32_t xh[2][3];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ku = ks * 32 + 16 * h;                                                    // wave-uniform part of the pixel index
        const uint32_t goff = (uint32_t)(ku * 256);
        const uint32_t xoff = RING ? (uint32_t)(ku * 128) : (uint32_t)(((ku >> a.lgWo) * a.pitch + (ku & (a.Wo - 1))) * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          gf[i][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(gt + gbase[i] + goff));
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xh[h][dx] = xbase[dx] + xoff;
      }
      bf16x8_t av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        av[i] = bf16x8_t{gf[i][0][0], gf[i][0][1], gf[i][0][2], gf[i][0][3], gf[i][1][0], gf[i][1][1], gf[i][1][2], gf[i][1][3]};
      // 9 taps: a 3-deep ring of x fragments, read two taps ahead of their MFMAs; tap t = (dy, dx) = (t / 3, t % 3)
      s16x4_t xf[3][2];
      auto x_read = [&](int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          xf[t % 3][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(xw + xh[h][t % 3] + (RING ? so[t / 3] : (uint32_t)(t / 3) * rowb)));
      };
      x_read(0);
      x_read(1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t + 2 < NT) x_read(t + 2);
        const bf16x8_t bv = {xf[t % 3][0][0], xf[t % 3][0][1], xf[t % 3][0][2], xf[t % 3][0][3],
                             xf[t % 3][1][0], xf[t % 3][1][1], xf[t % 3][1][2], xf[t % 3][1][3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv, acc[i][t], 0, 0, 0);
      }
    }
  }

  // D[row = n (fg*4+e)][col = c (fr)]
  float* part = a.part + (int64_t)sp * a.N * NT * a.Cx;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + (wn * 4 + i) * 16 + fg * 4 + e;
        part[((int64_t)n * NT + t) * a.Cx + c0 + wc * 16 + fr] = acc[i][t][e];
      }
}

}  // namespace

// splits per image the range-patch weight-gradient kernel wants for this problem (0 = descriptor does not qualify).
// The planner sizes `part` for B * spi slabs and sets nsplit = B * spi, variant = 1.
extern "C" int gan_wgrad_patch_splits(const gan_wgrad_desc* d) {
  static int disabled = -1;
  if (disabled < 0) { const char* e = getenv("GAN_NO_WPATCH"); disabled = (e && atoi(e)) ? 1 : 0; }
  if (disabled || !d) return 0;
  if (d->dtype != GAN_BF16 || d->ntaps != NT || d->Cx % CB != 0 || d->N % NB != 0 || d->N != d->g_C) return 0;
  if (d->x_sy != 1 || d->x_sx != 1 || d->g_sy != 1 || d->g_sx != 1) return 0;
  // 3x3 window in row-major tap order over a map whose width is a power of two dividing the stage
  if (d->Wo < 16 || (d->Wo & (d->Wo - 1)) != 0 || KM % d->Wo != 0 || d->max_tapoff != (2 * d->x_Wp + 2) * d->Cx) return 0;
  const int HoWo = d->Ho * d->Wo;
  if (HoWo < KM) return 0;
  const int pitch = (d->Wo + 2 + 7) / 8 * 8, nrows = KM / d->Wo + 2;
  if (nrows * pitch > RX && !(d->Wo == KM && pitch == RING_PITCH)) return 0;     // 128-wide maps: the row-ring variant
  const int blocks_per_split = (d->N / NB) * (d->Cx / CB);
  // Many small images (Basic_GAN: 16x16 maps at batch 256): with one split per image a block runs two stages between a full prologue and
  // a 300 KB partial store, and the reduction reads B slabs (0.77x of the generic kernel, measured).  A split then covers SEVERAL whole
  // images -- the negative return value: -(images per split), the largest divisor of B that still leaves one block per CU.
  if (HoWo < 8 * KM && d->B * blocks_per_split > 256 && HoWo % KM == 0 && nrows * pitch <= RX) {
    int ipb = d->B * blocks_per_split / 256;
    while (ipb > 1 && d->B % ipb != 0) --ipb;
    if (ipb > 1) return -ipb;
  }
  if (HoWo < 8 * KM && d->B > 64) return 0;
  int spi = (256 + d->B * blocks_per_split - 1) / (d->B * blocks_per_split);   // ~one block per CU
  const int max_spi = HoWo / (2 * KM) > 0 ? HoWo / (2 * KM) : 1;
  if (spi > max_spi) spi = max_spi;
  if (spi < 1) spi = 1;
  return spi;
}

int gan_wgrad_patch_launch(const gan_wgrad_desc* d, hipStream_t s) {
  const int spi_want = gan_wgrad_patch_splits(d);
  GAN_CHECK(spi_want != 0 && d->nsplit > 0 && (d->nsplit % d->B == 0 || (d->B % d->nsplit == 0 && (d->Ho * d->Wo) % KM == 0)),
            "wgrad: variant=1 but the descriptor does not qualify for the range-patch kernel");
  WpArgs a;
  a.x = (const char*)d->x; a.g = (const char*)d->g; a.part = d->part;
  a.B = d->B; a.HoWo = d->Ho * d->Wo; a.Wo = d->Wo; a.lgWo = __builtin_ctz(d->Wo);
  a.ipb = d->nsplit < d->B ? d->B / d->nsplit : 1;      // fewer splits than images: whole images per split
  a.spi = a.ipb > 1 ? 1 : d->nsplit / d->B;
  int per = (a.HoWo + a.spi - 1) / a.spi;
  per = (per + KM - 1) / KM * KM;
  a.per = per;
  GAN_CHECK((a.spi - 1) * per < a.HoWo, "wgrad_patch: nsplit=%d leaves empty splits", d->nsplit);
  a.Cx = d->Cx; a.N = d->N; a.pitch = (d->Wo + 2 + 7) / 8 * 8; a.nrows = KM / d->Wo + 2;
  a.x_Hp = d->x_Hp; a.x_Wp = d->x_Wp; a.x_y0 = d->x_y0; a.x_x0 = d->x_x0;
  a.g_Hp = d->g_Hp; a.g_Wp = d->g_Wp; a.g_C = d->g_C; a.g_y0 = d->g_y0; a.g_x0 = d->g_x0;
  a.NBLK = d->N / NB; a.CBLK = d->Cx / CB;
  // the planner's split count must be the one gan_wgrad_patch_splits answered (the kernels index partial slabs and images by it), and the
  // row-ring variant walks ONE image per split
  const bool ring = a.nrows * a.pitch > RX;
  GAN_CHECK(spi_want > 0 ? d->nsplit == d->B * spi_want : d->nsplit * -spi_want == d->B, "wgrad_patch: nsplit=%d is not what gan_wgrad_patch_splits implies (%d)",
            d->nsplit, spi_want);
  GAN_CHECK(!(ring && a.ipb > 1), "wgrad_patch: the row-ring variant (128-pixel-wide maps) takes one image per split");
  // the staging addresses are 32-bit byte offsets from the tensor bases
  GAN_CHECK((int64_t)d->B * d->x_Hp * d->x_Wp * d->Cx * 2 < (1ll << 32) && (int64_t)d->B * d->g_Hp * d->g_Wp * d->g_C * 2 < (1ll << 32),
            "wgrad_patch: an operand tensor exceeds the kernel's 32-bit byte offsets (4 GiB): split the batch");
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gan_set_error(-2, "wgrad_patch: hipGetDevice failed");
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
    if (hipFuncSetAttribute((const void*)wgrad_patch_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)wgrad_patch_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_RING) != hipSuccess)
      return gan_set_error(-2, "wgrad_patch: cannot raise the dynamic LDS limit to %d bytes", LDS_BYTES);
    attr_devs.fetch_or(dev_bit, std::memory_order_release);
  }
  if (a.nrows * a.pitch > RX) hipLaunchKernelGGL(wgrad_patch_kernel<true>, dim3(a.NBLK * a.CBLK * d->nsplit), dim3(512), LDS_BYTES_RING, s, a);
  else hipLaunchKernelGGL(wgrad_patch_kernel<false>, dim3(a.NBLK * a.CBLK * d->nsplit), dim3(512), LDS_BYTES, s, a);
  if (hipGetLastError() != hipSuccess) return gan_set_error(-2, "wgrad_patch: launch failed");
  return 0;
}
