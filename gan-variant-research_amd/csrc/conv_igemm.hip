// Generalised tap convolution as an implicit GEMM on CDNA4 MFMA (gfx950).
//
// Replaces the ATen convolution the reference reaches through nn.Conv2d / nn.ConvTranspose2d
// (GAN_Variant1/models/generator_resnet_attn.py:33,48,113,125,146-149,160; discriminator_patchgan.py:27-51;
// Basic_GAN/src/models.py:12-103): forward, input gradient and the transposed-conv sub-pixel phases are all
// the same kernel with different tap tables (see gan_conv_desc in include/mi355x_gan.h).
//
// Structure (MI355X-first, not a cuDNN/CUTLASS translation):
//  * activations are halo-NHWC, so GEMM row m of tap t is one contiguous 128-byte run of channels: the A tile
//    is gathered by `global_load_lds_dwordx4` (16 B per lane, LDS-DMA, no VGPR staging), 8 lanes per row;
//  * LDS rows are 128 B with the 16-byte chunk index XOR-swizzled by (row & 7) on the SOURCE address (the LDS
//    image of an LDS-DMA is lane-linear), making the ds_read_b128 fragment reads bank-conflict free;
//  * weights are the MFMA "A" operand and activations the "B" operand, so each lane ends up with 4 consecutive
//    output channels of one pixel -> one 8/16-byte store per 16x16 tile, bias/activation fused;
//  * bf16 path: v_mfma_f32_16x16x32_bf16; fp32 (parity) path: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain);
//  * block id -> tile mapping keeps all N-tiles of one M-tile on one XCD (shared A tile served from that L2).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

struct ConvArgs {
  const char* in; const char* w; const float* bias; char* out; const char* mask; const int32_t* tapoff; float* stats;
  int M, HoWo, Wo;
  uint32_t wo_magic;   // ceil(2^32 / Wo): r / Wo == umulhi(r, wo_magic) for every pixel index r < HoWo of one image (host-checked)
  int Cin, lgCin, ntaps, Ktot, nk;
  int in_Hp, in_Wp, in_y0, in_x0, in_sy, in_sx;
  int out_Hp, out_Wp, out_C, out_y0, out_x0, out_sy, out_sx;
  int Nst, act, MT, NTILES;
  int mask_Hp, mask_Wp, mask_y0, mask_x0;
  int dbg;  // timing-only ablations (GAN_CONV_DEBUG): 1 skip A staging after step 0, 2 skip B staging, 4 skip MFMA
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int EPC = 8;  // elements per 16-byte chunk
  static __device__ __forceinline__ void run(const u32x4_t& wa, const u32x4_t& xb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int EPC = 4;
  static __device__ __forceinline__ void run(const u32x4_t& wa, const u32x4_t& xb, f32x4_t& acc) {
    // lane group g = lane>>4 owns chunk g of this 4-chunk group; step s multiplies element s of every lane's chunk
    const f32x4_t wv = __builtin_bit_cast(f32x4_t, wa), xv = __builtin_bit_cast(f32x4_t, xb);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const char* gbase, uint32_t goff, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + goff),
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <typename T, int WM, int WN, int NT>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(ConvArgs a) {
  constexpr int BM = WM * 64, BN = WN * NT * 16, NTHR = WM * WN * 64;
  constexpr int EPC = Mma<T>::EPC, BKE = 8 * EPC;  // elements per 128-byte K-step
  constexpr int AI = BM / (NTHR / 8), BI = (BN + NTHR / 8 - 1) / (NTHR / 8);
  constexpr int STAGE = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE + 512];
  int32_t* taptab = reinterpret_cast<int32_t*>(lds + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // PERSISTENT blocks (round 3): a block walks virtual block ids L = blockIdx.x, + gridDim.x, ... and stages the first K-step of its next
  // tile before the epilogue of the current one.  Measured on the 3x3 stride-2 64->128 layer (9 K-steps per tile, 32 images): 121 of the
  // 199 us were prologue + first stage + epilogue of 4096 one-tile blocks (GAN_CONV_DEBUG=8); the MFMAs were 9 us of it.
  // XCD-aware tile mapping: virtual blocks L and L+8 share an XCD (gridDim.x is a multiple of 8); give them the same M-tile.
  const int nvirt = ((a.MT + 7) / 8) * 8 * a.NTILES;
  auto decode = [&](int L, int& m0, int& n0) {
    const int grp = L / (8 * a.NTILES), rem = L % (8 * a.NTILES);
    const int ntile = rem >> 3, mtile = grp * 8 + (rem & 7);
    m0 = mtile * BM; n0 = ntile * BN;
    return mtile < a.MT;
  };
  auto next_valid = [&](int L, int& m0, int& n0) {      // first virtual id >= L of this block's sequence that is a real tile, or -1
    for (; L < nvirt; L += gridDim.x)
      if (decode(L, m0, n0)) return L;
    return -1;
  };
  int m0, n0;
  int L = next_valid(blockIdx.x, m0, n0);
  if (L < 0) return;

  for (int i = tid; i < a.ntaps; i += NTHR) taptab[i] = a.tapoff[i];

  // ---- per-thread staging assignment: LDS position p of row r holds source chunk c = p ^ (r & 7)
  const int rr = tid >> 3, pp = tid & 7, cc = pp ^ (rr & 7);
  constexpr int RSTEP = NTHR / 8;  // rows covered per staging instruction of the block
  // pixel index -> (image, row, column) with ONE integer division per tile (of its first pixel; the rows of a tile are at most BM - 1
  // further on): the 16 divisions per tile and lane of the first version were as many issue slots as the MFMAs of a 9-step tile
  auto split = [&](int mbase, int bbase, int rbase, int m, int& b, int& ho, int& wo) {
    int r = rbase + (m - mbase);
    b = bbase;
    if (a.HoWo >= BM) { if (r >= a.HoWo) { r -= a.HoWo; ++b; } }
    else { const int q = r / a.HoWo; r -= q * a.HoWo; b += q; }        // maps smaller than a tile: several images per tile (rare, small)
    ho = a.wo_magic ? (int)__umulhi((uint32_t)r, a.wo_magic) : r;      // magic 0: one-pixel-wide maps
    wo = r - ho * a.Wo;
  };
  auto rows = [&](int m0, int n0, uint32_t (&a_row)[AI], uint32_t (&b_row)[BI]) {
    const int b0 = m0 / a.HoWo, r0 = m0 - b0 * a.HoWo;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int m = m0 + rr + RSTEP * i;
      m = m < a.M ? m : a.M - 1;
      int b, ho, wo;
      split(m0, b0, r0, m, b, ho, wo);
      a_row[i] = (uint32_t)(((b * a.in_Hp + ho * a.in_sy + a.in_y0) * a.in_Wp + wo * a.in_sx + a.in_x0) * a.Cin) * (uint32_t)sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) b_row[i] = (uint32_t)((n0 + rr + RSTEP * i) * a.Ktot + cc * EPC) * (uint32_t)sizeof(T);
  };
  uint32_t a_row[AI], b_row[BI];
  rows(m0, n0, a_row, b_row);
  const uint32_t lds_thr = (uint32_t)(wave * 1024);  // lane*16 is added by the hardware

  __syncthreads();  // tap table visible

  auto stage = [&](const uint32_t (&a_row)[AI], const uint32_t (&b_row)[BI], int ks, int buf) {
    char* sa = lds + buf * STAGE;
    char* sb = sa + BM * 128;
    const int kk = ks * BKE + cc * EPC;
    const uint32_t koff = (uint32_t)(taptab[kk >> a.lgCin] + (kk & (a.Cin - 1))) * (uint32_t)sizeof(T);
    if (!((a.dbg & 1) && ks > 0)) {
#pragma unroll
      for (int i = 0; i < AI; ++i) glds16(a.in, a_row[i] + koff, sa + lds_thr + i * (RSTEP * 128));
    }
    const uint32_t kb = (uint32_t)(ks * BKE) * (uint32_t)sizeof(T);
    if (!((a.dbg & 2) && ks > 0)) {
#pragma unroll
      for (int i = 0; i < BI; ++i)
        if (BN >= RSTEP * (i + 1) || rr + RSTEP * i < BN) glds16(a.w, b_row[i] + kb, sb + lds_thr + i * (RSTEP * 128));
    }
  };

  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;

  // fragment byte offsets inside a stage (row & 7 == fr & 7 for every tile row this lane reads)
  const int sw = fr & 7;
  uint32_t xa[4], wb[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = (uint32_t)((wm * 64 + i * 16 + fr) * 128);
#pragma unroll
  for (int j = 0; j < NT; ++j) wb[j] = (uint32_t)(BM * 128 + (wn * NT * 16 + j * 16 + fr) * 128);

  const int nk = (a.dbg & 8) ? 1 : a.nk;  // dbg 8: prologue + one K-step + epilogue only
  T* out = reinterpret_cast<T*>(a.out);
  const T* mask = reinterpret_cast<const T*>(a.mask);
  int buf = 0;
  stage(a_row, b_row, 0, 0);
  while (true) {
    // the tile after this one (its staging rows are computed while this tile's first stage is in flight)
    int m0n, n0n;
    const int Ln = next_valid(L + gridDim.x, m0n, n0n);
    uint32_t a_rown[AI], b_rown[BI];
    if (Ln >= 0) rows(m0n, n0n, a_rown, b_rown);

    f32x4_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int ks = 0; ks < nk; ++ks) {
      __syncthreads();  // stage ks landed (vmcnt(0) + barrier); everyone is done reading the other buffer
      if (ks + 1 < nk) stage(a_row, b_row, ks + 1, buf ^ 1);
      else if (Ln >= 0) stage(a_rown, b_rown, 0, buf ^ 1);     // the next tile's first stage rides under this tile's last K-step and epilogue
      const char* sbuf = lds + buf * STAGE;
      buf ^= 1;
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        const uint32_t co = (uint32_t)(((fg + 4 * kq) ^ sw) << 4);
        u32x4_t xf[4], wf[NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4_t*>(sbuf + xa[i] + co);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4_t*>(sbuf + wb[j] + co);
        if (a.dbg & 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(xf[i]));
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(wf[j]));
          continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) Mma<T>::run(wf[j], xf[i], acc[i][j]);
      }
    }

    // ---- epilogue: lane holds, per tile, pixel m = ..+fr and channels n = ..+fg*4 .. +3.  One bias quad per channel tile (not per value),
    // the activation switch outside the value loops, one division per tile
    {
      const int eb0 = m0 / a.HoWo, er0 = m0 - eb0 * a.HoWo;
      f32x4_t bq[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * NT * 16 + j * 16 + fg * 4;
        bq[j] = (a.bias && n < a.Nst) ? *reinterpret_cast<const f32x4_t*>(a.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      auto body = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + wm * 64 + i * 16 + fr;
          if (m >= a.M) continue;
          int b, ho, wo;
          split(m0, eb0, er0, m, b, ho, wo);
          const int64_t ob = ((int64_t)(b * a.out_Hp + ho * a.out_sy + a.out_y0) * a.out_Wp + wo * a.out_sx + a.out_x0) * a.out_C;
          // bf16, Nst % 8 == 0: lanes 16 apart hold the same pixel and adjacent channel quads; of each pair of channel tiles even fg keeps
          // the first and odd fg the second (v_permlane16_swap, as in conv_patch.hip) -> one 16-byte store per lane and tile pair instead of
          // two 8-byte ones (the 8-byte stores of the 3x3 s2 64->128 layer were 43 of its 135 us)
          const bool vec16 = sizeof(T) == 2 && NT % 2 == 0 && (a.Nst & 7) == 0 && (a.out_C & 7) == 0 && !(a.dbg & 16);
          u32x2_t pkv[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = n0 + wn * NT * 16 + j * 16 + fg * 4;
            pkv[j] = u32x2_t{0, 0};
            if (n >= a.Nst) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = acc[i][j][e] + bq[j][e];
              v[e] = ACT == GAN_ACT_RELU ? fmaxf(t, 0.f) : ACT == GAN_ACT_LRELU ? (t > 0.f ? t : 0.2f * t) : ACT == GAN_ACT_TANH ? tanhf(t) : t;
            }
            if (mask) {
              const int64_t mb = ((int64_t)(b * a.mask_Hp + ho * a.out_sy + a.mask_y0) * a.mask_Wp + wo * a.out_sx + a.mask_x0) * a.out_C;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= (ld1<T>(mask + mb + n + e) > 0.f ? 1.f : 0.2f);
            }
            if (a.dbg & 16) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); continue; }
            if constexpr (sizeof(T) == 4) {
              *reinterpret_cast<f32x4_t*>(out + ob + n) = f32x4_t{v[0], v[1], v[2], v[3]};
            } else {
              pkv[j][0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
              pkv[j][1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
              if (!vec16) *reinterpret_cast<u32x2_t*>(out + ob + n) = pkv[j];
            }
          }
          if constexpr (sizeof(T) == 2 && NT % 2 == 0) {
            if (vec16) {
#pragma unroll
              for (int jp = 0; jp < NT / 2; ++jp) {
                const auto ra = __builtin_amdgcn_permlane16_swap(pkv[2 * jp][0], pkv[2 * jp + 1][0], false, false);
                const auto rb = __builtin_amdgcn_permlane16_swap(pkv[2 * jp][1], pkv[2 * jp + 1][1], false, false);
                const int nst = n0 + wn * NT * 16 + (2 * jp + (fg & 1)) * 16 + (fg & 2) * 4;
                if (nst < a.Nst) *reinterpret_cast<u32x4_t*>(out + ob + nst) = u32x4_t{ra[0], rb[0], ra[1], rb[1]};
              }
            }
          }
        }
      };
      if (a.act == GAN_ACT_NONE) body(std::integral_constant<int, GAN_ACT_NONE>{});
      else if (a.act == GAN_ACT_RELU) body(std::integral_constant<int, GAN_ACT_RELU>{});
      else if (a.act == GAN_ACT_LRELU) body(std::integral_constant<int, GAN_ACT_LRELU>{});
      else body(std::integral_constant<int, GAN_ACT_TANH>{});
    }
    if (Ln < 0) break;
    L = Ln; m0 = m0n; n0 = n0n;
#pragma unroll
    for (int i = 0; i < AI; ++i) a_row[i] = a_rown[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) b_row[i] = b_rown[i];
  }
}

template <typename T, int WM, int WN, int NT>
int launch_conv(const ConvArgs& a, hipStream_t s) {
  constexpr int BM = WM * 64, BN = WN * NT * 16;
  ConvArgs k = a;
  k.MT = (a.M + BM - 1) / BM;
  k.NTILES = (a.Nst + BN - 1) / BN;
  const int nvirt = ((k.MT + 7) / 8) * 8 * k.NTILES;
  // persistent: as many blocks as the chip holds at once (LDS: two 2 x 32 KB blocks per CU on 256 CUs), a multiple of 8 (XCD mapping)
  static const int resident = [] { const char* e = getenv("GAN_IGEMM_BLOCKS"); const int v = e ? atoi(e) : 512; return v >= 8 ? (v / 8) * 8 : 512; }();
  const int grid = nvirt < resident ? nvirt : resident;
  hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, NT>), dim3(grid), dim3(WM * WN * 64), 0, s, k);
  GAN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int gan_conv_patch_launch(const gan_conv_desc* d, hipStream_t s);
int gan_conv_win7_launch(const gan_conv_desc* d, hipStream_t s);

extern "C" int gan_conv_igemm(const gan_conv_desc* d, void* stream) {
  GAN_CHECK(d, "conv: null descriptor");
  GAN_CHECK(d->dtype == GAN_F32 || d->dtype == GAN_BF16 || d->dtype == GAN_FP8, "conv: bad dtype %d", d->dtype);
  GAN_CHECK(d->dtype != GAN_FP8 || d->w_layout == 1, "conv: fp8 operands run on the range-patch kernel only (w_layout 1, gan_conv_patch_ok)");
  const int es = d->dtype == GAN_F32 ? 4 : d->dtype == GAN_FP8 ? 1 : 2, bke = 128 / es;
  GAN_CHECK(d->B > 0 && d->Ho > 0 && d->Wo > 0, "conv: empty problem");
  GAN_CHECK(d->Cin >= 8 && (d->Cin & (d->Cin - 1)) == 0, "conv: Cin=%d must be a power of two >= 8", d->Cin);
  GAN_CHECK(d->ntaps > 0 && d->ntaps <= 128 && ((int64_t)d->ntaps * d->Cin) % bke == 0, "conv: ntaps*Cin=%d*%d not a multiple of %d",
            d->ntaps, d->Cin, bke);
  GAN_CHECK(d->Nst > 0 && d->Nst % 4 == 0 && d->Nst <= d->out_C && d->out_C % 4 == 0, "conv: Nst=%d out_C=%d", d->Nst, d->out_C);
  GAN_CHECK(d->Nw >= d->Nst && d->Nw % 16 == 0, "conv: Nw=%d < Nst=%d or not a multiple of 16", d->Nw, d->Nst);
  GAN_CHECK(d->in && d->w && d->out && d->tapoff, "conv: null pointer");
  GAN_CHECK(((uintptr_t)d->in % 16) == 0 && ((uintptr_t)d->w % 16) == 0 && ((uintptr_t)d->out % 16) == 0, "conv: pointers must be 16-byte aligned");
  GAN_CHECK(d->stats == nullptr || d->w_layout == 1 || d->w_layout == 2, "conv: fused statistics exist only on the range-patch and 7x7 window paths (gan_conv_stats_parts)");
  GAN_CHECK(d->stats_mode == 0 || d->w_layout == 1, "conv: stats_mode 1 exists only on the range-patch path (gan_conv_patch_ok)");
  const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
  GAN_CHECK(M < (1ll << 31), "conv: M too large");
  const int64_t in_bytes = (int64_t)d->B * d->in_Hp * d->in_Wp * d->Cin * es;
  GAN_CHECK(in_bytes < (1ll << 32) && (int64_t)d->Nw * d->ntaps * d->Cin * es < (1ll << 32), "conv: operand exceeds 4 GiB addressing");
  // the last row/tap read must stay inside the allocation
  GAN_CHECK(d->in_y0 >= 0 && d->in_x0 >= 0 && (d->Ho - 1) * d->in_sy + d->in_y0 < d->in_Hp && (d->Wo - 1) * d->in_sx + d->in_x0 < d->in_Wp,
            "conv: input window outside the allocation");
  GAN_CHECK(d->out_y0 >= 0 && d->out_x0 >= 0 && (d->Ho - 1) * d->out_sy + d->out_y0 < d->out_Hp && (d->Wo - 1) * d->out_sx + d->out_x0 < d->out_Wp,
            "conv: output window outside the allocation");
  if (d->w_layout == 1) return gan_conv_patch_launch(d, (hipStream_t)stream);
  if (d->w_layout == 2) return gan_conv_win7_launch(d, (hipStream_t)stream);
  GAN_CHECK(d->w_layout == 0, "conv: bad w_layout %d", d->w_layout);
  ConvArgs a;
  a.in = (const char*)d->in; a.w = (const char*)d->w; a.bias = d->bias; a.out = (char*)d->out; a.mask = (const char*)d->mask;
  a.tapoff = d->tapoff; a.stats = d->stats;
  a.M = (int)M; a.HoWo = d->Ho * d->Wo; a.Wo = d->Wo;
  // r / Wo by multiplication: with magic = ceil(2^32 / Wo) the quotient is exact while r * Wo < 2^32; r < Ho * Wo (+ one tile)
  GAN_CHECK((int64_t)(a.HoWo + 256) * d->Wo < (1ll << 32), "conv: map too large for the index arithmetic");
  a.wo_magic = d->Wo == 1 ? 0u : (uint32_t)(((1ull << 32) + d->Wo - 1) / d->Wo);
  a.Cin = d->Cin; a.lgCin = __builtin_ctz(d->Cin); a.ntaps = d->ntaps; a.Ktot = d->ntaps * d->Cin; a.nk = a.Ktot / bke;
  a.in_Hp = d->in_Hp; a.in_Wp = d->in_Wp; a.in_y0 = d->in_y0; a.in_x0 = d->in_x0; a.in_sy = d->in_sy; a.in_sx = d->in_sx;
  a.out_Hp = d->out_Hp; a.out_Wp = d->out_Wp; a.out_C = d->out_C; a.out_y0 = d->out_y0; a.out_x0 = d->out_x0;
  a.out_sy = d->out_sy; a.out_sx = d->out_sx;
  a.Nst = d->Nst; a.act = d->act; a.MT = 0; a.NTILES = 0;
  { const char* e = getenv("GAN_CONV_DEBUG"); a.dbg = e ? atoi(e) : 0; }
  a.mask_Hp = d->mask_Hp; a.mask_Wp = d->mask_Wp; a.mask_y0 = d->mask_y0; a.mask_x0 = d->mask_x0;
  hipStream_t s = (hipStream_t)stream;
  // tile choice by the packed weight height (the packer pads Nw to the tile the launcher will use)
  if (d->Nw % 128 == 0) {
    GAN_DISPATCH_DTYPE(d->dtype, return launch_conv<T, 2, 2, 4>(a, s);)
  } else if (d->Nw % 64 == 0) {
    GAN_DISPATCH_DTYPE(d->dtype, return launch_conv<T, 2, 2, 2>(a, s);)
  } else {
    GAN_CHECK(d->Nw == 16, "conv: Nw=%d must be 16 or a multiple of 64", d->Nw);
    // few rows and a long reduction (the discriminator's 512 -> 1 4x4 convolution: 113 tiles of 256 rows x 128 K-steps -- 85 us for 31 MB of input on
    // less than half of the CUs): 64-row tiles, one wave per block, four times the blocks
    if ((M + 255) / 256 < 256) { GAN_DISPATCH_DTYPE(d->dtype, return launch_conv<T, 1, 1, 1>(a, s);) }
    GAN_DISPATCH_DTYPE(d->dtype, return launch_conv<T, 4, 1, 1>(a, s);)
  }
}
