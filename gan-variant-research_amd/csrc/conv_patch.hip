// "Range-patch" implicit-GEMM convolution: the fast path of gan_conv_igemm for windows on narrow maps with >= 64 input
// channels (the 18 residual 3x3 256->256 convolutions and their input gradients -- 40 % of the CUT step -- the
// transposed-conv phases and the last discriminator layers).
//
// Measurements on gfx950 (profiles/) showed the generic kernel (conv_igemm.hip) is neither bandwidth- nor
// bank-conflict-bound: its K-loop is LATENCY-bound by one block-wide barrier per 128-byte K-step around small dependent
// LDS read batches, and a third of its time is per-tile prologue/epilogue.  This kernel is built around removing
// synchronisation instead:
//  * A operand: the pixels a 256-row tile needs for ALL taps form one contiguous range of the halo-NHWC image (GEMM rows
//    are consecutive pixels, taps are constant pixel offsets).  One 64-channel slab of that range (<= 448 pixels x 128 B)
//    is staged into LDS once and every tap reads its fragments from it at a shifted row -> ONE barrier per slab (288 MFMA
//    per wave) instead of one per tap-step, and ~6x less global->LDS traffic for 3x3 windows;
//  * W operand: never touches LDS.  The weights are packed "fragment-major" ([n/16][k/32][lane][8 bf16], see
//    gan_pack_weight layout 1), so a wave fetches each MFMA A-fragment with one fully coalesced 1 KB global_load_dwordx4
//    straight from L2/L1 into operand registers, prefetched one MFMA k-step ahead in a second register set;
//  * one persistent 512-thread block per CU walks its tiles as one software pipeline: the next slab (or the next tile's
//    first slab) is fetched to registers and written to the other LDS buffer while the current one computes, so tile
//    boundaries cost no load latency.  Tile 256x128 (8 waves = 4(M) x 2(N), each 64x64 of v_mfma_f32_16x16x32_bf16) or
//    288x128 (2 x 4 waves of 144x32), whichever needs fewer CU-rounds x rows; consecutive tiles of one pixel tile go to one XCD;
//  * for exactly 9 taps the 256-row tile has a static schedule (NT = 9): unrolled tap loop, all fragment addresses of a tile
//    precomputed, LDS buffer index folded into the ds_read immediate;
//  * all global loads are plain VGPR loads (no LDS-DMA), so hipcc's counted s_waitcnt vmcnt(N) keeps the prefetch in flight;
//  * the epilogue pairs lanes 16 apart (same pixel, adjacent channel quads) and writes 16-byte stores; optionally it also
//    emits the InstanceNorm statistics of its tile (gan_conv_desc.stats) with DPP row reductions.
#include <stdlib.h>
#include <type_traits>
#include <array>
#include <atomic>
#include "common.h"

namespace {

struct PatchArgs {
  const char* in; const char* w; const float* bias; char* out; const char* mask; const int32_t* tapoff;
  int B, M_img, Wo, MT_img, NTILES, tiles;
  uint32_t wo_magic;            // ceil(2^32 / Wo): m / Wo == umulhi(m, wo_magic) for every output pixel index m of one image (host-checked)
  int Cin, nchunk, ntaps, KB;                             // KB = Ktot/32 fragment blocks per 16-row weight tile
  int nslice;                                             // 64-pixel slices of the patch buffer a tile really spans (<= NS)
  int in_Hp, in_Wp, in_y0, in_x0, in_sy, in_sx, in_pix;   // in_pix: pixels in the whole input tensor
  int out_Hp, out_Wp, out_C, out_y0, out_x0, out_sy, out_sx;
  int Nst, act;
  int mask_Hp, mask_Wp, mask_y0, mask_x0;
  int w_bytes;
  uint32_t out_bytes, mask_bytes;   // whole-tensor sizes: the epilogue addresses `out` / `mask` through buffer descriptors with 32-bit offsets
  int tapdiv;                   // tapoff[] / tapdiv = pixel offset of a tap (the operand's REAL channel count; Cin counts 2-byte slots)
  const float* w_scale; const float* in_scale;   // FP8: dequantisation scales (weights: one float; input: per image or NULL)
  float* stats;                 // optional per-tile InstanceNorm partials [B][MT_img][out_C][2] (sum, sum of squares), plain stores
  int smode;                    // statistics mode (gan_conv_desc.stats_mode)
  unsigned long long* stamps;   // diagnostic build only (GAN_PATCH_STAMPS): [block][32] s_memtime stamps of wave 0
};

constexpr int NTHR = 512;   // the tile is a template parameter: BM = 256 (4 x 2 waves of 64 x 64) or 288 (2 x 4 waves of 144 x 32) rows by BN = 128 channels,
                            // or BM x 256 with 2 x 4 waves of (BM/2) x 64 (the whole Cout of the residual layers in one tile; see the BN == 256 loop)
// 8 waves = (8/WGN) pixel groups x WGN channel groups.  Measured on the 3x3 256->256 layer (s_memtime stamps, cycles per
// 64-channel slab): 4 x 2 (64 x 64 per wave) 15.5 k, 2 x 4 (128 x 32 per wave, half the weight bytes through the vector L1,
// twice the LDS reads) 16.0 k; MFMA alone would be 9.2 k.  Neither operand path is the limiter: with two waves per SIMD each
// 16x16x32 MFMA holds the SIMD's issue port for 8 of its 16 cycles, so the 30-odd non-MFMA instructions of a k-step do not
// hide.  The 4 x 2 split is kept (shorter first slab, fewer address registers).
// Tile 288 x 128 exists for tile-count quantisation: the 66x66 input-gradient domain is 4356 pixels per image = 18 tiles of 256
// (576 tiles = 2.25 waves on 256 CUs -> 3 rounds) but 16 tiles of 288 (512 tiles -> 2 rounds of 1.125x the work: -25 %).
// pixels per patch buffer: 7 slices of 64 (maps up to 64 pixels wide: a 256-pixel tile of a 3x3 window spans 6 padded rows), or 9 slices
// for maps up to 128 wide (512x512 images: 4 padded rows of 130) -- template parameter NS; two buffers of 9 slices are 147 KB of LDS
constexpr int RMAX = 448, RMAX_WIDE = 576;
constexpr int STATS_LDS = 8 * 128 * 2 * 4;    // per-wave (sum, sumsq) of up to 128 channels, combined across the pixel-split waves
constexpr int lds_bytes(int ns) { return 2 * ns * 64 * 128 + 512 + STATS_LDS; }

struct TileGeo { int b, m0, n0, P0; };

// sum over the 16 lanes of a DPP row (every lane ends with the total): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});
  v += dpp(v, std::integral_constant<int, 0x4E>{});
  v += dpp(v, std::integral_constant<int, 0x141>{});
  v += dpp(v, std::integral_constant<int, 0x140>{});
  return v;
}

__device__ __forceinline__ int pixbase(const PatchArgs& a, int b, int m) {
  const int ho = (int)__umulhi((uint32_t)m, a.wo_magic), wo = m - ho * a.Wo;
  return (b * a.in_Hp + ho * a.in_sy + a.in_y0) * a.in_Wp + wo * a.in_sx + a.in_x0;
}
template <int BM, int BN>
__device__ __forceinline__ TileGeo tile_geo(const PatchArgs& a, int tau) {
  TileGeo g;
  const int mt = tau / a.NTILES;
  g.n0 = (tau - mt * a.NTILES) * BN;
  g.b = mt / a.MT_img;
  g.m0 = (mt - g.b * a.MT_img) * BM;
  g.P0 = pixbase(a, g.b, g.m0);
  return g;
}

// NT > 0: static schedule for exactly NT taps and an even number of 64-channel slabs -- the tap loop is unrolled, tap offsets
// live in scalar registers, the LDS buffer index is a compile-time constant (folded into the ds_read immediate) and, where
// registers allow (FI <= 4), the swizzled fragment addresses of all taps are computed once per tile.  NT = 0: any tap count.
// FP8: `in` and `w` hold e4m3 bytes, two channels per 2-byte slot of the bf16 image (Cin = real channels / 2): the kernel moves exactly
// the bytes it moves for bf16; the two 16-byte fragments a lane reads for the two 32-slot k-steps of a tap are the low / high half of
// its 32-byte operand of v_mfma_scale_f32_16x16x128_f8f6f4 (lane l: row l % 16, k block l / 16 -- probed with exact integer data,
// tools/probe/fp8_mfma.hip; both operands permute the 128 channels of a k-step identically, so the product is unchanged).  Per byte
// moved the fp8 MFMA takes the cycles of the two bf16 MFMAs it replaces: twice the FLOPs per byte, HBM and LDS traffic halved.
typedef __attribute__((ext_vector_type(8))) int v8i_t;
typedef __attribute__((ext_vector_type(4))) int v4i_t;
// EPI: 0 = the forward / plain input-gradient epilogues; 1 = the backward-chain epilogue (gan_conv_desc.stats_mode 1) -- kernels of their
// own, so that it does not cost the other instantiations a register.
template <int BM, int WGN, int NT, bool FP8 = false, int NS = 7, int BN = 128, int EPI = 0>
__device__ __forceinline__ void conv_patch_body(const PatchArgs& a) {
  static_assert(EPI == 0 || (!FP8 && NT == 0), "backward-chain epilogues: bf16, generic tap loop");
  static_assert(!FP8 || NT == 0, "the fp8 path uses the generic tap loop");
  static_assert(BN == 128 || (BN == 256 && WGN == 4 && NT == 0 && !FP8), "256-channel tiles: 2 x 4 waves, generic tap loop, bf16");
  static_assert(NS == 7 || BM == 256, "the 9-slice buffers exist for the 256-row tile only");
  constexpr int NSLICE = NS, PATCHB = NS * 64 * 128;
  constexpr int FI = BM / (8 / WGN) / 16, FJ = BN / WGN / 16;   // fragments per wave: FI pixel groups x FJ channel groups
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  char* pbuf = lds;                       // [2][PATCHB]
  int32_t* taptab = reinterpret_cast<int32_t*>(lds + 2 * PATCHB);
  float* stsh = reinterpret_cast<float*>(lds + 2 * PATCHB + 512);   // [8 waves][16*FJ channels][2]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave roles are scalars: nothing derived from them costs a VGPR
  for (int i = tid; i < a.ntaps; i += NTHR) taptab[i] = a.tapoff[i] / a.tapdiv;   // pixel offsets
  int stoff[NT > 0 ? NT : 1];   // static schedule: pixel offset of every tap, wave-uniform
  if constexpr (NT > 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) stoff[t] = __builtin_amdgcn_readfirstlane(a.tapoff[t] / a.tapdiv);
  }
  const int G = gridDim.x;
  // XCD-aware start tile: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), and the NTILES channel tiles
  // of one pixel tile (consecutive tau) read the same input patch -> give consecutive tau to workgroups of ONE XCD so the
  // patch is fetched into one L2 once (PMC: 97 MB fetched per launch against 37 MB of input)
  int tau = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (tau >= a.tiles) return;

  // slab staging role: patch row r = 64*j + wave*8 + (lane>>3); LDS position p = lane&7 holds source chunk p ^ (r&7)
  const int sr = wave * 8 + (lane >> 3), sp = lane & 7, sc = sp ^ (sr & 7);
  const uint32_t pix_bytes = (uint32_t)a.Cin * 2u;
  const uint32_t st_lds = (uint32_t)(sr * 128 + sp * 16);
  auto slab_load = [&](const TileGeo& g, int chunk, int j) -> u32x4_t {
    int srow = sr;
    asm volatile("" : "+v"(srow));   // keeps the (unrolled) per-slice source addresses from being hoisted and spilled
    int pix = g.P0 + 64 * j + srow;
    pix = pix < a.in_pix ? pix : a.in_pix - 1;
    return *reinterpret_cast<const u32x4_t*>(a.in + (size_t)((uint32_t)pix * pix_bytes + (uint32_t)(chunk * 128 + sc * 16)));
  };
  auto slab_store = [&](int buf, int j, const u32x4_t& v) {
    *reinterpret_cast<u32x4_t*>(pbuf + buf * PATCHB + j * 8192 + st_lds) = v;
  };

  const int wm = wave / WGN, wn = wave % WGN;
  const int fr = lane & 15, fg = lane >> 4;
  // weights, fragment-major: byte offset of (n16, kb, lane) = ((n16*KB + kb)*64 + lane)*16
  // weights through a buffer descriptor: voffset = lane*16 (constant), everything else is a wave-uniform scalar offset,
  // so a weight fetch costs no vector ALU work at all
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
  const int lane16 = lane * 16;
  const int wn_u = __builtin_amdgcn_readfirstlane(wn);
  auto w_load = [&](int n0_tile, int kb, u32x4_t (&f)[FJ]) {   // one MFMA k-step (32 channels) of this wave's weight rows
    const int base = __builtin_amdgcn_readfirstlane((((n0_tile + wn_u * (16 * FJ)) >> 4) * a.KB + kb) * 1024);   // provably wave-uniform: no waterfall
#pragma unroll
    for (int j = 0; j < FJ; ++j)
      f[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * a.KB * 1024, 0));
  };
  const bool dbg_w0 = a.stamps && ((uintptr_t)a.stamps & 2);   // diagnostic: every weight fetch reads block 0 (L1-resident)
  auto kb_of = [&](int c, int t) { return dbg_w0 ? 0 : (t * a.Cin + c * 64) >> 5; };

  int nstamp = 0;
  auto stamp = [&]() {
    if (a.stamps && wave == 0 && nstamp < 30) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (lane == 0) {
        unsigned long long* sb = (unsigned long long*)((uintptr_t)a.stamps & ~(uintptr_t)7) + blockIdx.x * 32;
        sb[nstamp] = t;
        // slots 30 / 31: the 100 MHz wall counter at the first / latest stamp (in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz)
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime();
        if (nstamp == 0) sb[30] = rt;
        sb[31] = rt;
      }
      ++nstamp;
    }
  };
  stamp();
  TileGeo g = tile_geo<BM, BN>(a, tau);
  // prologue: first slab -> LDS buffer 0, first tap's weights -> registers
  {
    u32x4_t tmp[NSLICE];
#pragma unroll
    for (int j = 0; j < NSLICE; ++j) tmp[j] = slab_load(g, 0, j < a.nslice ? j : 0);     // slices past the tile's span: a harmless re-read
#pragma unroll
    for (int j = 0; j < NSLICE; ++j) slab_store(0, j, tmp[j]);
  }
  u32x4_t Wa[FJ], Wb[FJ], Xa[FI], Xb[FI];   // operand fragments of the even / odd k-step of a tap
  // FP8: 32-byte operands of a whole tap (both k-steps), two sets: the tap being multiplied and the next one in flight
  v8i_t W8[2][FP8 ? FJ : 1], X8[1][FP8 ? FI : 1];
  auto w_load8 = [&](int n0_tile, int kb, v8i_t (&f)[FP8 ? FJ : 1]) {
    const int base = __builtin_amdgcn_readfirstlane((((n0_tile + wn_u * (16 * FJ)) >> 4) * a.KB + kb) * 1024);
#pragma unroll
    for (int j = 0; j < (FP8 ? FJ : 1); ++j) {
      const u32x4_t lo = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * a.KB * 1024, 0));
      const u32x4_t hi = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * a.KB * 1024 + 1024, 0));
      f[j] = __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  if constexpr (FP8) w_load8(g.n0, kb_of(0, 0), W8[0]);
  else w_load(g.n0, kb_of(0, 0), Wa);

  __syncthreads();   // tap table + slab 0 visible
  stamp();

  int pcur = 0;
  while (true) {
    int lbase[FI];
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      int m = g.m0 + wm * (16 * FI) + i * 16 + fr;
      m = m < a.M_img ? m : a.M_img - 1;
      lbase[i] = pixbase(a, g.b, m) - g.P0;
    }
    const int tau_next = tau + G;
    const bool has_next = tau_next < a.tiles;
    TileGeo gn = g;
    if (has_next) gn = tile_geo<BM, BN>(a, tau_next);

    f32x4_t acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // 16 MFMAs of one k-step in two parts, so that the next k-step's fetches can be issued AFTER the waits that guard this
    // k-step's operands (a wait placed before the first MFMA would otherwise also cover the fetches just issued)
    auto mma_part = [&](const u32x4_t (&wf)[FJ], const u32x4_t (&xf)[FI], int i0, int i1) {
#pragma unroll
      for (int i = 0; i < FI; ++i)
        if (i >= i0 && i < i1) {
#pragma unroll
          for (int j = 0; j < FJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xf[i]), acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NT > 0) {
      constexpr bool PRE = FI <= 4 && NT <= 9;       // 16 taps x 4 fragment addresses would be 64 registers
      uint32_t xa[PRE ? NT : 1][FI];
      if constexpr (PRE) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int prow = lbase[i] + stoff[t];
            xa[t][i] = (uint32_t)(prow * 128 + ((fg ^ (prow & 7)) << 4));
          }
      }
      const int cin32 = a.Cin >> 5;
      auto static_slab = [&](auto pc_tag, int c) {
        constexpr int PC = decltype(pc_tag)::value;
        const bool last_chunk = c + 1 == a.nchunk;
        const bool stage_next = !last_chunk || has_next;
        const TileGeo gs = last_chunk ? gn : g;
        const int cs = last_chunk ? 0 : c + 1;
        const char* pb = pbuf + PC * PATCHB;
        // slices of the next slab moved per tap: a tap writes what the previous tap fetched and fetches its own share, so that the last tap's
        // writes complete the slab (few-tap layers -- the sub-pixel phases have 1, 2 or 4 taps -- move several slices per tap; one-tap layers
        // flush after the tap).  The tap loop stays branch-free in its loads: hipcc's s_waitcnt insertion turns conservative at every
        // control-flow join (vmcnt(0): every k-step of the generic loop waits for the slab loads it has just issued -- 1.4-1.7 k cycles
        // per k-step of 0.5 k cycles of MFMAs on the 4-tap phase launches)
        constexpr int SP = NT > 1 ? (NSLICE + NT - 2) / (NT - 1) : NSLICE;
        u32x4_t stg[SP];
#pragma unroll
        for (int u = 0; u < SP; ++u) stg[u] = u32x4_t{0, 0, 0, 0};
        // hipcc hoists loop-invariant per-lane addresses out of the (unrolled) loops and then spills them: the values that
        // feed the address arithmetic are made opaque where they are used
        auto x_load = [&](int t, int kq, u32x4_t (&xf)[FI]) {
          uint32_t flip = kq ? 64u : 0u;
          if (kq) asm volatile("" : "+v"(flip));
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            uint32_t ad;
            if constexpr (PRE) ad = xa[t][i];
            else {
              int lb = lbase[i];
              asm volatile("" : "+v"(lb));
              const int prow = lb + stoff[t];
              ad = (uint32_t)(prow * 128 + ((fg ^ (prow & 7)) << 4));
            }
            xf[i] = *reinterpret_cast<const u32x4_t*>(pb + (ad ^ flip));
          }
        };
        x_load(0, 0, Xa);   // Wa was fetched by the previous slab's last step (or the prologue)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int kb_cur = t * cin32 + 2 * c;
          __builtin_amdgcn_sched_barrier(0);
          mma_part(Wa, Xa, 0, FI / 4);
          __builtin_amdgcn_sched_barrier(0);
          w_load(g.n0, kb_cur + 1, Wb);
          x_load(t, 1, Xb);
          if (stage_next) {
#pragma unroll
            for (int u = 0; u < SP; ++u) {
              if (t >= 1 && (t - 1) * SP + u < NSLICE) slab_store(PC ^ 1, (t - 1) * SP + u, stg[u]);
              // slices past the tile's span are never read: re-read slice 0 instead (an L1 / L2 hit) -- these launches are bound by the bytes they stage
              if (t * SP + u < NSLICE) stg[u] = slab_load(gs, cs, t * SP + u < a.nslice ? t * SP + u : 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          mma_part(Wa, Xa, FI / 4, FI);
          __builtin_amdgcn_sched_barrier(0);
          mma_part(Wb, Xb, 0, FI / 4);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < NT) {
            w_load(g.n0, kb_cur + cin32, Wa);
            x_load(t + 1, 0, Xa);
          } else {   // first tap of the next slab, or of the next tile (after the very last tap: a harmless re-read)
            w_load(last_chunk ? gn.n0 : g.n0, last_chunk ? 0 : 2 * (c + 1), Wa);
          }
          __builtin_amdgcn_sched_barrier(0);
          mma_part(Wb, Xb, FI / 4, FI);
        }
        if constexpr ((NT - 1) * SP < NSLICE) {      // one-tap layers: what the only tap fetched is written after it (an exposed round trip)
          if (stage_next) {
#pragma unroll
            for (int u = 0; u < SP; ++u)
              if ((NT - 1) * SP + u < NSLICE) slab_store(PC ^ 1, (NT - 1) * SP + u, stg[u]);
          }
        }
        __syncthreads();   // every wave is done with this slab; the other buffer is completely written
        stamp();
      };
      for (int c = 0; c < a.nchunk; c += 2) {   // nchunk is even: every tile starts on buffer 0
        static_slab(std::integral_constant<int, 0>{}, c);
        static_slab(std::integral_constant<int, 1>{}, c + 1);
      }
    } else if constexpr (FP8) {
      // One k-step = one tap over the whole 128-byte slab row.  Registers are what limits this variant (the bf16 kernels leave 80 per
      // SIMD lane to the kernels of the other streams; so must this one): weights are double-buffered (L2 latency), the activation
      // fragments are ONE set that is re-read for the next tap fragment by fragment, each right after the four MFMAs that consume it
      // (pixel-group-major MFMA order): its LDS latency hides behind the remaining MFMAs of the tap.
      for (int c = 0; c < a.nchunk; ++c) {
        const bool last_chunk = c + 1 == a.nchunk;
        const bool stage_next = !last_chunk || has_next;
        const TileGeo gs = last_chunk ? gn : g;
        const int cs = last_chunk ? 0 : c + 1;
        const char* pb = pbuf + pcur * PATCHB;
        u32x4_t stg = {0, 0, 0, 0};
        int sj = 0;
        uint32_t xaddr[FI];
        auto x_addr = [&](int toff) {
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int prow = lbase[i] + toff;
            xaddr[i] = (uint32_t)(prow * 128 + ((fg ^ (prow & 7)) << 4));
          }
        };
        auto x_read = [&](int i) -> v8i_t {
          const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(pb + xaddr[i]), hi = *reinterpret_cast<const u32x4_t*>(pb + (xaddr[i] ^ 64u));
          return __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto step = [&](auto par_tag, int t) {
          constexpr int P = decltype(par_tag)::value;
          const bool last_tap = t + 1 == a.ntaps;
          const int nt = last_tap ? 0 : t + 1;
          const int nc = last_tap ? (last_chunk ? 0 : c + 1) : c;
          const int nn0 = (last_tap && last_chunk) ? gn.n0 : g.n0;
          const int kb_next = kb_of(nc, nt);
          x_addr(taptab[last_tap ? t : t + 1]);        // after the last tap: a harmless re-read (the next slab is read after the barrier)
          __builtin_amdgcn_sched_barrier(0);
          w_load8(nn0, kb_next, W8[P ^ 1]);            // next tap, else first tap of the next slab / tile
#pragma unroll
          for (int i = 0; i < (FP8 ? FI : 1); ++i) {
#pragma unroll
            for (int j = 0; j < (FP8 ? FJ : 1); ++j)
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W8[P][j], X8[0][i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            __builtin_amdgcn_sched_barrier(0);
            X8[0][i] = x_read(i);
            if (i == 0 && stage_next) {
              if (sj > 0 && sj <= NSLICE) slab_store(pcur ^ 1, sj - 1, stg);
              if (sj < NSLICE) stg = slab_load(gs, cs, sj);
              ++sj;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        x_addr(taptab[0]);
#pragma unroll
        for (int i = 0; i < (FP8 ? FI : 1); ++i) X8[0][i] = x_read(i);
        int t = 0;
        for (; t + 1 < a.ntaps; t += 2) {
          step(std::integral_constant<int, 0>{}, t);
          step(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < a.ntaps) {     // odd tap count: the prefetched weights change sets
          step(std::integral_constant<int, 0>{}, t);
#pragma unroll
          for (int j = 0; j < (FP8 ? FJ : 1); ++j) W8[0][j] = W8[1][j];
        }
        if (stage_next) {
          while (sj <= NSLICE) {
            if (sj > 0) slab_store(pcur ^ 1, sj - 1, stg);
            if (sj < NSLICE) stg = slab_load(gs, cs, sj);
            ++sj;
          }
        }
        __syncthreads();
        pcur ^= 1;
        stamp();
      }
    } else if constexpr (BN == 256) {
      // 256-channel tile: a wave owns (BM/2) x 64 outputs = 8-9 pixel fragments x 4 channel fragments, 32-36 MFMAs per k-step against
      // 12-13 operand fetches (the 128-channel tiles: 16 against 8), and the slab is staged once for the whole Cout.  An MFMA holds the
      // SIMD's vector issue for 8 of its 16 cycles, so what limits the 128-channel tiles is the ~30 other instructions per 16 MFMAs of
      // two waves sharing one issue port; here there are ~20 per 36.  Registers: 128-144 accumulators, ONE weight set and ONE set of
      // activation fragments: a k-step runs channel-fragment-major, a fragment's weights are refetched for the next k-step as soon as
      // its FI MFMAs are issued (27 MFMAs of cover for the L2 latency) and the activation fragments are re-read during the last
      // channel fragment, each right after the MFMA that consumes it (8 MFMAs of cover for the LDS latency).
      // The k-steps are branch-free (hipcc's s_waitcnt insertion turns conservative -- vmcnt(0) right behind the weight fetches -- at
      // every join): the 7 slices of the next slab ride on the first 8 k-steps, which are unrolled with their slice numbers as
      // constants (the descriptor has >= 4 taps); after the block's last slab the staging re-reads the current tile into the idle buffer.
      // Measured on the 3x3 256->256 layer, B = 16 (s_memtime, cycles per 64-channel slab; MFMA alone 18.4 k / 20.7 k): 256 rows 24.1 k,
      // 288 rows 27.3 k = 76 % (the 128-channel tiles: 69-73 %) and half the epilogues; forward 65.6 -> 58.4 us, input gradient 75.7 -> 64.5 us.
      // Where the other 5.8 k cycles of a slab go (each fetch class compiled out in turn, results wrong, timing only): activation
      // re-reads 2.8 k (LDS latency: 8 MFMAs of cover, a second register set does not fit), weight fetches 1.8 k, staging 0.5 k,
      // barrier + first reads of the slab 0.7 k -- no single limiter is left.  (A second activation set for the 256-row tile, filled during the
      // first channel fragment of the previous k-step -- 24 MFMAs of cover -- measured SLOWER: 24.1 k -> 25.0 k cycles per slab.)
      constexpr int STAGED_TAPS = (NSLICE + 2) / 2;    // taps whose k-steps carry the NSLICE + 1 staging slots (fetch of slice s, then its LDS write)
      for (int c = 0; c < a.nchunk; ++c) {
        const bool last_chunk = c + 1 == a.nchunk;
        const TileGeo gs = last_chunk ? gn : g;
        const int cs = last_chunk ? 0 : c + 1;
        const char* pb = pbuf + pcur * PATCHB;
        u32x4_t stg = {0, 0, 0, 0};
        uint32_t xaddr[FI];
        auto x_addr = [&](int toff) {
#pragma unroll
          for (int i = 0; i < FI; ++i) {
            const int prow = lbase[i] + toff;
            xaddr[i] = (uint32_t)(prow * 128 + ((fg ^ (prow & 7)) << 4));
          }
        };
        // one k-step: multiplies (w, Xa), refetches w for the k-step (n0_next, kb_next), re-reads Xa[i] at xaddr[i] ^ flip, and moves
        // one slice of the next slab: SS = slice written to LDS (fetched by the previous k-step), LS = slice fetched; -1: none
        auto kstep = [&](auto ss_tag, auto ls_tag, u32x4_t (&w)[FJ], int n0_next, int kb_next, uint32_t flip) {
          constexpr int SS = decltype(ss_tag)::value, LS = decltype(ls_tag)::value;
          const int base = __builtin_amdgcn_readfirstlane((((n0_next + wn_u * (16 * FJ)) >> 4) * a.KB + kb_next) * 1024);
#pragma unroll
          for (int j = 0; j < FJ; ++j) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < FI; ++i) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[j]), __builtin_bit_cast(bf16x8_t, Xa[i]), acc[i][j], 0, 0, 0);
              if (j == FJ - 1) {
                __builtin_amdgcn_sched_barrier(0);
                Xa[i] = *reinterpret_cast<const u32x4_t*>(pb + (xaddr[i] ^ flip));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            w[j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane16, base + j * a.KB * 1024, 0));
            if (j == 0) {
              if constexpr (SS >= 0) slab_store(pcur ^ 1, SS, stg);
              if constexpr (LS >= 0) stg = slab_load(gs, cs, LS);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        auto tap = [&](auto s0, auto l0, auto s1, auto l1, int t) {
          const bool last_tap = t + 1 == a.ntaps;
          const int toff_next = taptab[last_tap ? t : t + 1];     // read a k-step before its use (after the last tap: a harmless re-read)
          const int nt = last_tap ? 0 : t + 1;
          const int nc = last_tap ? (last_chunk ? 0 : c + 1) : c;
          const int nn0 = (last_tap && last_chunk) ? gn.n0 : g.n0;
          const int kb_cur = kb_of(c, t), kb_next = kb_of(nc, nt);
          kstep(s0, l0, Wa, g.n0, kb_cur + 1, 64u);              // first 32 channels of the tap; re-read: the other half of the row
          x_addr(toff_next);
          kstep(s1, l1, Wa, nn0, kb_next, 0u);                   // second half; re-read: the next tap
        };
        x_addr(taptab[0]);
#pragma unroll
        for (int i = 0; i < FI; ++i) Xa[i] = *reinterpret_cast<const u32x4_t*>(pb + xaddr[i]);   // Wa: fetched by the previous slab's last k-step (or the prologue)
        using IC = integral_constant<int, -1>;
        // k-step ks (= 2 * tap + half) writes slice ks - 1 and fetches slice ks, while they exist
        auto staged_tap = [&](auto t_tag) {
          constexpr int T = decltype(t_tag)::value;
          constexpr int S0 = 2 * T - 1, L0 = 2 * T, S1 = 2 * T, L1 = 2 * T + 1;
          tap(integral_constant<int, (S0 >= 0 && S0 < NSLICE) ? S0 : -1>{}, integral_constant<int, L0 < NSLICE ? L0 : -1>{},
              integral_constant<int, S1 < NSLICE ? S1 : -1>{}, integral_constant<int, L1 < NSLICE ? L1 : -1>{}, T);
        };
        staged_tap(integral_constant<int, 0>{});
        staged_tap(integral_constant<int, 1>{});
        staged_tap(integral_constant<int, 2>{});
        staged_tap(integral_constant<int, 3>{});
        if constexpr (STAGED_TAPS > 4) staged_tap(integral_constant<int, 4>{});
        static_assert(STAGED_TAPS <= 5, "at most 9 slices");
        for (int t = STAGED_TAPS; t < a.ntaps; ++t) tap(IC{}, IC{}, IC{}, IC{}, t);
        __syncthreads();
        pcur ^= 1;
        stamp();
      }
    } else
    for (int c = 0; c < a.nchunk; ++c) {
      const bool last_chunk = c + 1 == a.nchunk;
      const bool stage_next = !last_chunk || has_next;
      const TileGeo gs = last_chunk ? gn : g;   // owner of the next slab
      const int cs = last_chunk ? 0 : c + 1;
      const char* pb = pbuf + pcur * PATCHB;
      // Staging of the next slab: two slots per tap (one behind each weight fetch); a slot fetches `sps` slices (1, 2 or 4: few-tap
      // layers -- the sub-pixel phases of the transposed convolutions have 1, 2 or 4 taps -- would otherwise leave most of the slab to
      // a serial flush after the taps: measured 5 exposed fetch -> write round trips per slab on a 2-tap layer) and writes the slices
      // the previous slot fetched; only the a.nslice slices the tile's span touches are moved
      u32x4_t stg[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      const int nsl = a.nslice;
      const int sps = 2 * a.ntaps - 1 >= nsl ? 1 : (2 * (2 * a.ntaps - 1) >= nsl ? 2 : 4);
      int slot = 0;
      auto stage_slot = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < sps) {
            const int sl_prev = (slot - 1) * sps + u, sl = slot * sps + u;
            if (slot > 0 && sl_prev < nsl) slab_store(pcur ^ 1, sl_prev, stg[u]);
            if (sl < nsl) stg[u] = slab_load(gs, cs, sl);
          }
        }
        ++slot;
      };

      // A fragments of one k-step of tap t: 4 x ds_read_b128 from the slab at the tap's row shift.  The second k-step's
      // chunk index differs by 4, i.e. its swizzled address is the first one's XOR 64.
      uint32_t xaddr[FI];
      auto x_addr = [&](int toff) {
#pragma unroll
        for (int i = 0; i < FI; ++i) {
          const int prow = lbase[i] + toff;
          xaddr[i] = (uint32_t)(prow * 128 + ((fg ^ (prow & 7)) << 4));
        }
      };
      auto x_load = [&](int kq, u32x4_t (&xf)[FI]) {
#pragma unroll
        for (int i = 0; i < FI; ++i) xf[i] = *reinterpret_cast<const u32x4_t*>(pb + (xaddr[i] ^ (kq ? 64u : 0u)));
      };
      x_addr(taptab[0]);
      x_load(0, Xa);   // Wa was fetched by the previous slab's last step (or the prologue)
      for (int t = 0; t < a.ntaps; ++t) {
        // where the NEXT tap's weights live: next tap, else first tap of the next slab, else of the next tile (branch-free scalars;
        // after the very last tap this re-reads the current tile's first block, harmlessly)
        const bool last_tap = t + 1 == a.ntaps;
        const int nt = last_tap ? 0 : t + 1;
        const int nc = last_tap ? (last_chunk ? 0 : c + 1) : c;
        const int nn0 = (last_tap && last_chunk) ? gn.n0 : g.n0;
        const int kb_cur = kb_of(c, t), kb_next = kb_of(nc, nt);

        __builtin_amdgcn_sched_barrier(0);
        mma_part(Wa, Xa, 0, FI / 4);                 // waits for Wa / Xa (issued one k-step ago) land here
        __builtin_amdgcn_sched_barrier(0);
        w_load(g.n0, kb_cur + 1, Wb);
        x_load(1, Xb);
        const int toff_next = taptab[last_tap ? t : t + 1];   // fetched a half step before x_addr needs it
        if (stage_next) stage_slot();
        __builtin_amdgcn_sched_barrier(0);
        mma_part(Wa, Xa, FI / 4, FI);
        __builtin_amdgcn_sched_barrier(0);
        mma_part(Wb, Xb, 0, FI / 4);
        __builtin_amdgcn_sched_barrier(0);
        w_load(nn0, kb_next, Wa);
        if (!last_tap) { x_addr(toff_next); x_load(0, Xa); }   // the next slab's activations wait for the barrier
        if (stage_next) stage_slot();
        __builtin_amdgcn_sched_barrier(0);
        mma_part(Wb, Xb, FI / 4, FI);
      }
      // write what the last slot fetched (and, on one-tap layers, fetch and write the rest), then hand the buffer over
      if (stage_next) {
        while ((slot - 1) * sps < nsl) stage_slot();
      }
      __syncthreads();   // every wave is done with slab `pcur`; slab `pcur^1` is completely written
      pcur ^= 1;
      stamp();
    }

    // ---- epilogue (the next tile's slab is in LDS and its first weights are in flight).  Specialised on the activation at
    // compile time: with a run-time switch per value the 64 results per lane made this phase VALU-bound (8.4k cycles).
    auto epilogue = [&](auto act_tag, auto mask_tag, auto stats_tag) {
      constexpr int ACT = decltype(act_tag)::value;
      constexpr bool MASK = decltype(mask_tag)::value;
      constexpr bool STATS = decltype(stats_tag)::value;
      // lane roles re-derived from an opaque copy: otherwise hipcc computes the epilogue's per-lane offsets once per kernel, finds no free
      // register across the tap loop of the 256-channel tiles and spills them -- and every scratch reload waits with vmcnt(0), which
      // also drains the stores in flight (measured: 21 k cycles of epilogue instead of 9 k)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int fr = lane_o & 15, fg = lane_o >> 4;
      // results leave through a buffer descriptor: 32-bit offsets (the 64-bit address arithmetic per store was a fifth of the epilogue's
      // instructions and its temporaries spilled beside 144 accumulators), and a store outside the tile's real rows / channels is sent
      // to offset 2^32 - 16, which the range check drops -- no branch around the stores
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, 0, a.mask_bytes, 0x00020000);
      float oscale = 1.f;
      if constexpr (FP8) oscale = a.w_scale[0] * (a.in_scale ? a.in_scale[g.b] : 1.f);
      // The 256-channel tiles walk the pixel groups once per PAIR of channel fragments (JG passes): with 128-144 accumulators live, the
      // bias quads and statistics of all four fragments at once (48 registers) do not fit beside the next tile's weights in flight
      // Statistics are summed over groups of 64 (256-row tiles) or 144 (288-row tiles) pixels at either tile width -- HS groups per wave --
      // and combined in group order: the partials, and with them every result downstream, do not depend on the tile width the
      // planner picks (it depends on the batch size: G(x)[i] stays bit-identical whatever else is in the batch)
      constexpr int JG = BN == 256 ? FJ / 2 : 1, JW = FJ / JG;
      constexpr int HS = (BN == 256 && BM == 256) ? 2 : 1, FH = FI / HS;
#pragma unroll
      for (int jg = 0; jg < JG; ++jg) {
        const int j0 = jg * JW;
        f32x4_t bq[JW];   // bias of this lane's channel quads
#pragma unroll
        for (int j = 0; j < JW; ++j) {
          const int n = g.n0 + wn * (16 * FJ) + (j0 + j) * 16 + fg * 4;
          bq[j] = (a.bias && n < a.Nst) ? *reinterpret_cast<const f32x4_t*>(a.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int h = 0; h < HS; ++h) {
        float ssum[STATS ? 4 * JW : 1], ssq[STATS ? 4 * JW : 1];
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 4 * JW; ++q) ssum[q] = ssq[q] = 0.f;
        }
#pragma unroll
        for (int i = h * FH; i < (h + 1) * FH; ++i) {
          __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from overlapping pixel groups (registers: it spills on the wide tiles, and the narrow ones must leave room for the other streams)
          int lf = lane_o;
          if constexpr (JG > 1) asm volatile("" : "+v"(lf));   // every pass recomputes its addresses from the lane id (shared, they would be spilled between the passes)
          const int m = g.m0 + wm * (16 * FI) + i * 16 + (lf & 15);
          const bool mok = m < a.M_img;
          const int mm = mok ? m : a.M_img - 1;
          const int ho = (int)__umulhi((uint32_t)mm, a.wo_magic), wo = mm - ho * a.Wo;
          const uint32_t ob = (uint32_t)(((g.b * a.out_Hp + ho * a.out_sy + a.out_y0) * a.out_Wp + wo * a.out_sx + a.out_x0) * a.out_C);   // element offset
          u32x2_t pk[JW];
#pragma unroll
          for (int j = 0; j < JW; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = FP8 ? acc[i][j0 + j][e] * oscale + bq[j][e] : acc[i][j0 + j][e] + bq[j][e];
              if constexpr (STATS) { const float tm = mok ? t : 0.f; ssum[4 * j + e] += tm; ssq[4 * j + e] += tm * tm; }
              v[e] = ACT == GAN_ACT_RELU ? fmaxf(t, 0.f) : ACT == GAN_ACT_LRELU ? (t > 0.f ? t : 0.2f * t) : ACT == GAN_ACT_TANH ? tanhf(t) : t;
            }
            if (MASK) {
              const int n = g.n0 + wn * (16 * FJ) + (j0 + j) * 16 + fg * 4;
              if (n < a.Nst) {
                const uint32_t mb = (uint32_t)(((g.b * a.mask_Hp + ho * a.out_sy + a.mask_y0) * a.mask_Wp + wo * a.out_sx + a.mask_x0) * a.out_C);
                const u32x2_t mv = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(mrsrc, (int)((mb + (uint32_t)n) * 2u), 0, 0));
                v[0] *= (bf2f((bf16_t)(mv[0] & 0xffff)) > 0.f ? 1.f : 0.2f); v[1] *= (bf2f((bf16_t)(mv[0] >> 16)) > 0.f ? 1.f : 0.2f);
                v[2] *= (bf2f((bf16_t)(mv[1] & 0xffff)) > 0.f ? 1.f : 0.2f); v[3] *= (bf2f((bf16_t)(mv[1] >> 16)) > 0.f ? 1.f : 0.2f);
              }
            }
            pk[j][0] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk[j][1] = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          }
          // lanes 16 apart hold the same pixel and adjacent channel quads: of each pair of channel tiles even fg keeps the first
          // and odd fg the second; each sends the quad of the tile it does not keep -> one 16-byte run per lane and tile pair
#pragma unroll
          for (int jp = 0; jp < JW / 2; ++jp) {
            // v_permlane16_swap(X, Y): lanes of the even 16-lane rows end with (own X, partner's X), lanes of the odd rows with (partner's Y,
            // own Y) -- the exchange and the selection in two vector instructions (probed: tools/probe/permlane_swap.hip); the former
            // __shfl_xor pair was two ds_bpermute round trips through the LDS pipe plus four selects
            const auto ra = __builtin_amdgcn_permlane16_swap(pk[2 * jp][0], pk[2 * jp + 1][0], false, false);
            const auto rb = __builtin_amdgcn_permlane16_swap(pk[2 * jp][1], pk[2 * jp + 1][1], false, false);
            const u32x4_t st = {ra[0], rb[0], ra[1], rb[1]};
            const int fgl = JG > 1 ? lf >> 4 : fg;   // 256-channel tiles: from the opaque lane id (a hoisted 64-bit store base would be spilled and reloaded per store)
            const int nst = g.n0 + wn * (16 * FJ) + (j0 + 2 * jp + (fgl & 1)) * 16 + (fgl & 2) * 4;
            const uint32_t off = (mok && nst < a.Nst) ? (ob + (uint32_t)nst) * 2u : 0xfffffff0u;
            __builtin_amdgcn_raw_buffer_store_b128(st, orsrc, (int)off, 0, 0);
          }
        }
        if constexpr (STATS) {
          // InstanceNorm partials of this tile: over the 16 pixel lanes of a lane group, then over the pixel-split waves in a
          // fixed order (deterministic: no atomics), one float2 per channel to stats[b][m-tile][n]
          // row (16-lane) all-reduce with DPP modifiers on the adds -- vector ALU only; __shfl_xor lowers to ds_bpermute and 256 of
          // those per wave cost 8 us per launch, as much as the statistics pass they replace
#pragma unroll
          for (int q = 0; q < 4 * JW; ++q) { ssum[q] = row16_sum(ssum[q]); ssq[q] = row16_sum(ssq[q]); }
          if (fr == 0) {
#pragma unroll
            for (int q = 0; q < 4 * JW; ++q) {
              const int ch = (j0 + (q >> 2)) * 16 + fg * 4 + (q & 3);           // channel inside this wave's 16*FJ
              *reinterpret_cast<float2*>(stsh + ((((wm * HS + h) * WGN + wn) * (16 * FJ) + ch) << 1)) = make_float2(ssum[q], ssq[q]);
            }
          }
        }
        }
      }
      if constexpr (STATS) {
        __syncthreads();
        constexpr int WM = 8 / WGN * HS;
        if (tid < BN) {
          const int cwn = tid / (16 * FJ), cch = tid % (16 * FJ);
          float2 tot = make_float2(0.f, 0.f);
#pragma unroll
          for (int m = 0; m < WM; ++m) {
            const float2 v2 = *reinterpret_cast<const float2*>(stsh + (((m * WGN + cwn) * (16 * FJ) + cch) << 1));
            tot.x += v2.x; tot.y += v2.y;
          }
          const int n = g.n0 + tid;
          if (n < a.Nst)
            *reinterpret_cast<float2*>(a.stats + (((int64_t)g.b * a.MT_img + g.m0 / BM) * a.out_C + n) * 2) = tot;
        }
        __syncthreads();   // stsh is rewritten by the next tile
      }
    };
    // ---- backward-chain epilogue (EPI == 1; gan_conv_desc.stats_mode 1): besides the padded-domain input gradient it leaves the two sums the
    // InstanceNorm backward behind a ReLU needs, sum g [y > 0] and sum g y, y = the saved activation at the output pixel's own position.
    // y comes from HBM (written a whole forward pass ago): fetched two pixel groups ahead of its use the loads were latency-bound (49 k cycles
    // per tile against 8 k for the plain epilogue).  So the epilogue runs in two passes: pass A rounds a pixel group's accumulators to bf16
    // (16 registers become 8), stores them, and issues the group's y quads into the 8 registers that freed -- after it ALL of the tile's y
    // loads are in flight at no extra register; pass B walks the groups again and sums from the rounded gradient (what the consumer will
    // read) and y.  The per-tile partials are summed over the same pixel groups in the same order at either tile width.
    auto epilogue_chain = [&]() {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int fr = lane_o & 15, fg = lane_o >> 4;
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask, 0, a.mask_bytes, 0x00020000);
      constexpr int HS = (BN == 256 && BM == 256) ? 2 : 1, FH = FI / HS;
      u32x2_t pk[FI][FJ], yq[FI][FJ];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const int m = g.m0 + wm * (16 * FI) + i * 16 + fr;
        const bool mok = m < a.M_img;
        const int mm = mok ? m : a.M_img - 1;
        const int ho = (int)__umulhi((uint32_t)mm, a.wo_magic), wo = mm - ho * a.Wo;
        uint32_t ob = (uint32_t)(((g.b * a.out_Hp + ho * a.out_sy + a.out_y0) * a.out_Wp + wo * a.out_sx + a.out_x0) * a.out_C);
        uint32_t mb = (uint32_t)(((g.b * a.mask_Hp + ho * a.out_sy + a.mask_y0) * a.mask_Wp + wo * a.out_sx + a.mask_x0) * a.out_C);
        asm volatile("" : "+v"(ob), "+v"(mb));      // computed here, not inside a branch on `mok` (a join costs a conservative wait)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          pk[i][j][0] = (uint32_t)f2bf(acc[i][j][0]) | ((uint32_t)f2bf(acc[i][j][1]) << 16);
          pk[i][j][1] = (uint32_t)f2bf(acc[i][j][2]) | ((uint32_t)f2bf(acc[i][j][3]) << 16);
        }
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          const int n = g.n0 + wn * (16 * FJ) + j * 16 + fg * 4;
          yq[i][j] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(mrsrc, n < a.Nst ? (int)((mb + (uint32_t)n) * 2u) : (int)0xfffffff0u, 0, 0));
        }
#pragma unroll
        for (int jp = 0; jp < FJ / 2; ++jp) {
          const auto ra = __builtin_amdgcn_permlane16_swap(pk[i][2 * jp][0], pk[i][2 * jp + 1][0], false, false);
          const auto rb = __builtin_amdgcn_permlane16_swap(pk[i][2 * jp][1], pk[i][2 * jp + 1][1], false, false);
          const u32x4_t st = {ra[0], rb[0], ra[1], rb[1]};
          const int nst = g.n0 + wn * (16 * FJ) + (2 * jp + (fg & 1)) * 16 + (fg & 2) * 4;
          const uint32_t off = (mok && nst < a.Nst) ? (ob + (uint32_t)nst) * 2u : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(st, orsrc, (int)off, 0, 0);
        }
      }
#pragma unroll
      for (int h = 0; h < HS; ++h) {
        float ssum[4 * FJ], ssq[4 * FJ];
#pragma unroll
        for (int q = 0; q < 4 * FJ; ++q) ssum[q] = ssq[q] = 0.f;
#pragma unroll
        for (int i = h * FH; i < (h + 1) * FH; ++i) {
          __builtin_amdgcn_sched_barrier(0);
          const bool mok = g.m0 + wm * (16 * FI) + i * 16 + fr < a.M_img;
#pragma unroll
          for (int j = 0; j < FJ; ++j) {
            const u32x2_t gq = pk[i][j], y2 = yq[i][j];
            const float gv[4] = {__builtin_bit_cast(float, gq[0] << 16), __builtin_bit_cast(float, gq[0] & 0xffff0000u),
                                 __builtin_bit_cast(float, gq[1] << 16), __builtin_bit_cast(float, gq[1] & 0xffff0000u)};
            const float yv[4] = {__builtin_bit_cast(float, y2[0] << 16), __builtin_bit_cast(float, y2[0] & 0xffff0000u),
                                 __builtin_bit_cast(float, y2[1] << 16), __builtin_bit_cast(float, y2[1] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float tm = mok ? gv[e] : 0.f;
              ssum[4 * j + e] += yv[e] > 0.f ? tm : 0.f;
              ssq[4 * j + e] += tm * yv[e];
            }
          }
#pragma unroll
          for (int q = 0; q < 4 * FJ; ++q) asm volatile("" : "+v"(ssum[q]), "+v"(ssq[q]));   // pinned: hipcc otherwise sinks the sums below the loop and keeps every group alive
        }
#pragma unroll
        for (int q = 0; q < 4 * FJ; ++q) { ssum[q] = row16_sum(ssum[q]); ssq[q] = row16_sum(ssq[q]); }
        if (fr == 0) {
#pragma unroll
          for (int q = 0; q < 4 * FJ; ++q) {
            const int ch = (q >> 2) * 16 + fg * 4 + (q & 3);           // channel inside this wave's 16*FJ
            *reinterpret_cast<float2*>(stsh + ((((wm * HS + h) * WGN + wn) * (16 * FJ) + ch) << 1)) = make_float2(ssum[q], ssq[q]);
          }
        }
      }
      __syncthreads();
      constexpr int WM = 8 / WGN * HS;
      if (tid < BN) {
        const int cwn = tid / (16 * FJ), cch = tid % (16 * FJ);
        float2 tot = make_float2(0.f, 0.f);
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          const float2 v2 = *reinterpret_cast<const float2*>(stsh + (((m * WGN + cwn) * (16 * FJ) + cch) << 1));
          tot.x += v2.x; tot.y += v2.y;
        }
        const int n = g.n0 + tid;
        if (n < a.Nst)
          *reinterpret_cast<float2*>(a.stats + (((int64_t)g.b * a.MT_img + g.m0 / BM) * a.out_C + n) * 2) = tot;
      }
      __syncthreads();   // stsh is rewritten by the next tile
    };
    using std::integral_constant;
    if constexpr (EPI == 1) epilogue_chain();
    else {
      if (a.mask) epilogue(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, true>{}, integral_constant<bool, false>{});   // LeakyReLU' masks only follow plain dgrads
      else if (a.act == GAN_ACT_NONE && a.stats) epilogue(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{});
      else if (a.act == GAN_ACT_NONE) epilogue(integral_constant<int, GAN_ACT_NONE>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{});
      else if (a.act == GAN_ACT_LRELU) epilogue(integral_constant<int, GAN_ACT_LRELU>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{});
      else if (a.act == GAN_ACT_RELU) epilogue(integral_constant<int, GAN_ACT_RELU>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{});
      else epilogue(integral_constant<int, GAN_ACT_TANH>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{});
    }

    stamp();
    if (!has_next) break;
    tau = tau_next;
    g = gn;
  }
}

template <int BM, int WGN, int NT, bool FP8 = false, int NS = 7, int BN = 128>
__global__ __launch_bounds__(NTHR) void conv_patch_kernel(PatchArgs a) { conv_patch_body<BM, WGN, NT, FP8, NS, BN>(a); }
template <int BM, int WGN, int NS, int BN>
__global__ __launch_bounds__(NTHR) void conv_patch_bwdchain_kernel(PatchArgs a) { conv_patch_body<BM, WGN, 0, false, NS, BN, 1>(a); }
// The fp8 variants take all 256 registers (two 32-byte weight sets, one activation set, 64 accumulators and the compiler's scheduling
// slack), so nothing of another stream runs beside them.  Capping them (`amdgpu_num_vgpr(108)`: on gfx90a+ the request is doubled, a
// request of 216 is silently dropped) was measured: 216 registers = 304 bytes of scratch per lane inside the tap loop, 3x slower.
__global__ __launch_bounds__(NTHR) void conv_patch_fp8_kernel(PatchArgs a) { conv_patch_body<256, 2, 0, true, 7>(a); }
__global__ __launch_bounds__(NTHR) void conv_patch_fp8_wide_kernel(PatchArgs a) { conv_patch_body<256, 2, 0, true, 9>(a); }

}  // namespace

// pixels of the contiguous input range a tile of BM output pixels reads (all taps)
static int patch_span(const gan_conv_desc* d, int BM) {
  const int M_img = d->Ho * d->Wo;
  const int maxtap = d->max_tapoff / d->Cin;
  const int rows = BM < M_img ? BM : M_img;
  const int wraps = (rows - 1) / d->Wo + 1;
  const int jump = d->in_Wp * d->in_sy - d->Wo * d->in_sx;
  return (rows - 1) * d->in_sx + wraps * (jump > 0 ? jump : 0) + maxtap + 1;
}

// tile width for a tile height: the planner's choice if the descriptor carries one; otherwise 256 channels when the layer's Cout is a
// multiple of it and the 256-wide tiles still fill the chip (bf16, >= 4 taps, maps up to 64 pixels wide: the 7-slice buffers), else 128.
// GAN_PATCH_BN = 128 | 256 forces one wherever the layer is eligible (a tuning / test aid that only the planning call reads).
static int patch_bn(const gan_conv_desc* d, int BM, bool planning = false) {
  // >= 4 taps (5 on maps wider than 64 pixels, whose 9 slices take 10 k-steps): the staging schedule is unrolled over the first taps
  const int span = patch_span(d, BM);
  const bool eligible = d->dtype == GAN_BF16 && d->Nw % 256 == 0 && d->Nst % 256 == 0 &&
                        ((span <= RMAX && d->ntaps >= 4) || (BM == 256 && span <= RMAX_WIDE && d->ntaps >= 5));
  if (!eligible) return 128;
  if (!planning && (d->tile_cols == 128 || d->tile_cols == 256)) return d->tile_cols;
  if (planning) {
    const char* e = getenv("GAN_PATCH_BN");
    const int forced = e ? atoi(e) : 0;
    if (forced == 128 || forced == 256) return forced;
  }
  const int64_t tiles = (int64_t)d->B * ((d->Ho * d->Wo + BM - 1) / BM) * (d->Nst / 256);
  return tiles >= 192 ? 256 : 128;
}

// tile height: the planner's choice if the descriptor carries one, otherwise the one that needs the fewest CU-rounds x tile area
// (GAN_PATCH_BM forces one; a tuning aid that only the planning call gan_conv_patch_tile_rows reads)
static int patch_tile_rows(const gan_conv_desc* d, bool planning = false) {
  constexpr int BN = 128;
  const int M_img = d->Ho * d->Wo, ncu = 256;
  int BM = 0, forced = 0;
  int64_t best = 0;
  if (d->dtype == GAN_FP8) return 256;      // only 256-row e4m3 kernels exist: a descriptor's tile_rows cannot ask for another
  if (!planning && (d->tile_rows == 256 || d->tile_rows == 288) && patch_span(d, d->tile_rows) <= (d->tile_rows == 256 ? RMAX_WIDE : RMAX)) return d->tile_rows;
  if (planning) { const char* e = getenv("GAN_PATCH_BM"); forced = e ? atoi(e) : 0; }
  for (int cand : {256, 288}) {
    const int lim = cand == 256 ? RMAX_WIDE : RMAX;      // the 9-slice buffers exist for the 256-row tile only
    if (patch_span(d, cand) > lim || (forced && forced != cand && patch_span(d, forced) <= (forced == 256 ? RMAX_WIDE : RMAX))) continue;
    const int bn = patch_bn(d, cand, planning);
    const int64_t tiles = (int64_t)d->B * ((M_img + cand - 1) / cand) * ((d->Nst + bn - 1) / bn);
    const int64_t cost = ((tiles + ncu - 1) / ncu) * cand * (bn / BN);
    if (!BM || cost < best) { BM = cand; best = cost; }
  }
  return BM;
}

// Pure predicate (no device access): does this descriptor qualify for the range-patch kernel?  The planner asks at
// build time because qualifying calls need the fragment-major weight packing (gan_pack_weight layout 1).
extern "C" int gan_conv_patch_ok(const gan_conv_desc* d) {
  static int disabled = -1;
  if (disabled < 0) { const char* e = getenv("GAN_NO_PATCH"); disabled = (e && atoi(e)) ? 1 : 0; }
  if (disabled || !d) return 0;
  constexpr int BN = 128;
  if (d->mask && d->act != GAN_ACT_NONE) return 0;   // the masked epilogue is specialised for act = none
  const bool fp8 = d->dtype == GAN_FP8;
  const int slots = fp8 ? d->Cin / 2 : d->Cin;             // 2-byte slots per pixel (fp8: two channels per slot)
  if ((d->dtype != GAN_BF16 && !fp8) || slots < 64 || slots % 64 != 0 || d->Nw % BN != 0 || d->Nst % 8 != 0 || d->out_C % 8 != 0) return 0;
  if (fp8 && (d->mask || !d->w_scale || patch_span(d, 256) > RMAX_WIDE)) return 0;   // fp8: 256-row tile, plain epilogue
  if (d->max_tapoff <= 0 || d->ntaps < 1 || d->Wo < 2) return 0;     // one-pixel-wide maps: m / Wo has no 32-bit magic (umulhi(m, 2^32 - 1) = m - 1)
  // 32-bit byte offsets into the input, output and mask tensors
  const int64_t lim = (1ll << 32) - 4096;
  if ((int64_t)d->B * d->in_Hp * d->in_Wp * d->Cin * 2 >= lim || (int64_t)d->B * d->out_Hp * d->out_Wp * d->out_C * 2 >= lim ||
      (d->mask && (int64_t)d->B * d->mask_Hp * d->mask_Wp * d->out_C * 2 >= lim)) return 0;
  if (!(patch_span(d, 256) <= RMAX_WIDE || patch_span(d, 288) <= RMAX)) return 0;
  // tile utilisation: a map of 324 pixels (18x18 input-gradient domain of a 16x16 layer) fills 63 % of two 256-row tiles -- the
  // generic kernel's 128-row tiles waste less there (Basic_GAN at 64x64: +5 % with it)
  const int M_img = d->Ho * d->Wo;
  const int t256 = (M_img + 255) / 256 * 256, t288 = (M_img + 287) / 288 * 288;
  const int rows = t256 < t288 ? t256 : t288;
  if (d->B * ((M_img + 255) / 256) <= 128) return 1;   // few tiles: the CUs are not full either way
  return 4 * M_img >= 3 * rows ? 1 : 0;
}

extern "C" int gan_conv_patch_tile_rows(const gan_conv_desc* d) {
  return gan_conv_patch_ok(d) ? patch_tile_rows(d, true) : 0;
}

extern "C" int gan_conv_patch_tile_cols(const gan_conv_desc* d) {
  return gan_conv_patch_ok(d) ? patch_bn(d, patch_tile_rows(d), true) : 0;
}

// InstanceNorm partials per image the range-patch kernel writes to d->stats ([B][parts][out_C][2]); 0: this descriptor cannot fuse them
int gan_conv_win7_stats_parts(const gan_conv_desc* d);
extern "C" int gan_conv_stats_parts(const gan_conv_desc* d) {
  if (d && d->w_layout == 2) return gan_conv_win7_stats_parts(d);
  if (!gan_conv_patch_ok(d) || d->act != GAN_ACT_NONE || (d->mask && d->stats_mode == 0) || d->out_sy != 1 || d->out_sx != 1) return 0;
  if (d->stats_mode != 0 && (d->dtype != GAN_BF16 || !d->mask || d->stats_mode != 1 || d->bias)) return 0;
  const int BM = patch_tile_rows(d);
  return (d->Ho * d->Wo + BM - 1) / BM;
}

// Which instantiation a qualifying descriptor runs on (the launch and the planner's query share this decision):
// tile rows | tile columns << 12 | LDS slices (7: maps up to 64 pixels wide, 9: up to 128) << 24 | e4m3 operands << 28 | static 3x3 schedule << 29
static int patch_variant(const gan_conv_desc* d) {
  static const bool static_off = [] { const char* e = getenv("GAN_PATCH_STATIC"); return e && !atoi(e); }();
  const int BM = patch_tile_rows(d);
  const int BN = patch_bn(d, BM);
  const bool fp8 = d->dtype == GAN_FP8;
  const int slots = fp8 ? d->Cin / 2 : d->Cin;
  const bool wide = BM == 256 && patch_span(d, 256) > RMAX;      // needs the 9-slice buffers (maps wider than 64 pixels)
  const bool chain = d->stats_mode != 0;      // backward-chain epilogue: generic tap loop
  const bool st_ok = !fp8 && !chain && BN == 128 && BM == 256 && (slots / 64) % 2 == 0 && !static_off;
  const bool st9 = st_ok && !wide && d->ntaps == 9;
  // static schedules for the other tap counts of the two networks (bits 30-31: 1 = 4 taps, 2 = 2 taps, 3 = 16 taps): the sub-pixel phases of the
  // transposed convolutions / strided input gradients (2 and 4 taps, also on the 9-slice buffers) and the discriminator's 4x4 windows
  const int stx = !st_ok ? 0 : d->ntaps == 4 ? 1 : d->ntaps == 2 ? 2 : (d->ntaps == 16 && !wide) ? 3 : 0;
  return BM | (BN << 12) | ((wide ? 9 : 7) << 24) | ((fp8 ? 1 : 0) << 28) | ((st9 ? 1 : 0) << 29) | (stx << 30);
}
extern "C" int gan_conv_patch_variant(const gan_conv_desc* d) { return gan_conv_patch_ok(d) ? patch_variant(d) : 0; }

int gan_conv_patch_launch(const gan_conv_desc* d, hipStream_t s) {
  if (!gan_conv_patch_ok(d)) return gan_set_error(-1, "conv: w_layout=1 (fragment-major weights) but the descriptor does not qualify for the range-patch kernel");
  PatchArgs a;
  const int M_img = d->Ho * d->Wo;
  const int ncu = 256;
  const int variant = patch_variant(d);
  const int BM = variant & 0xfff, BN = (variant >> 12) & 0xfff;
  a.in = (const char*)d->in; a.w = (const char*)d->w; a.bias = d->bias; a.out = (char*)d->out; a.mask = (const char*)d->mask; a.tapoff = d->tapoff;
  // m / Wo by multiplication: with magic = ceil(2^32 / Wo) the quotient is exact while m * (magic * Wo - 2^32) < 2^32, i.e. for m * Wo < 2^32
  if ((int64_t)(M_img + 288) * d->Wo >= (1ll << 32) || d->Wo < 1) return gan_set_error(-1, "conv_patch: map too large for the index arithmetic");
  a.wo_magic = (uint32_t)(((1ull << 32) + d->Wo - 1) / d->Wo);      // Wo >= 2 (gan_conv_patch_ok): for Wo = 1 no 32-bit magic exists
  a.B = d->B; a.M_img = M_img; a.Wo = d->Wo; a.MT_img = (M_img + BM - 1) / BM; a.NTILES = (d->Nst + BN - 1) / BN;
  a.tiles = a.B * a.MT_img * a.NTILES;
  const bool fp8 = d->dtype == GAN_FP8;
  a.Cin = fp8 ? d->Cin / 2 : d->Cin;          // 2-byte slots per pixel and tap
  a.tapdiv = d->Cin; a.w_scale = d->w_scale; a.in_scale = d->in_scale;
  a.nchunk = a.Cin / 64; a.ntaps = d->ntaps; a.KB = d->ntaps * a.Cin / 32;
  a.nslice = (patch_span(d, BM) + 63) / 64;
  a.w_bytes = d->Nw * d->ntaps * a.Cin * 2;
  // 32-bit byte offsets everywhere (buffer descriptors; slab addresses): every operand tensor must stay below 4 GiB - 16 (a dropped store is
  // sent to offset 2^32 - 16).  At 2 bytes x 256 channels that is ~490 images of a 130 x 130 padded map: far above any batch of this path.
  const int64_t lim = (1ll << 32) - 16;
  if ((int64_t)d->B * d->out_Hp * d->out_Wp * d->out_C * 2 > lim || (int64_t)d->B * d->in_Hp * d->in_Wp * (fp8 ? d->Cin : d->Cin * 2) > lim ||
      (d->mask && (int64_t)d->B * d->mask_Hp * d->mask_Wp * d->out_C * 2 > lim))
    return gan_set_error(-1, "conv_patch: an operand tensor exceeds the kernel's 32-bit byte offsets (4 GiB): split the batch");
  a.out_bytes = (uint32_t)((int64_t)d->B * d->out_Hp * d->out_Wp * d->out_C * 2);
  a.mask_bytes = d->mask ? (uint32_t)((int64_t)d->B * d->mask_Hp * d->mask_Wp * d->out_C * 2) : 0;
  a.in_Hp = d->in_Hp; a.in_Wp = d->in_Wp; a.in_y0 = d->in_y0; a.in_x0 = d->in_x0; a.in_sy = d->in_sy; a.in_sx = d->in_sx;
  a.in_pix = d->B * d->in_Hp * d->in_Wp;
  a.out_Hp = d->out_Hp; a.out_Wp = d->out_Wp; a.out_C = d->out_C; a.out_y0 = d->out_y0; a.out_x0 = d->out_x0; a.out_sy = d->out_sy; a.out_sx = d->out_sx;
  a.Nst = d->Nst; a.act = d->act; a.stats = d->stats;
  if (d->stats && (d->act != GAN_ACT_NONE || (d->mask && d->stats_mode == 0))) return gan_set_error(-1, "conv: fused statistics need act = none and no mask");
  // backward-chain epilogue (gan_conv_desc.stats_mode 1): bf16 operands, plain result of an input gradient (no bias)
  a.smode = d->stats ? d->stats_mode : 0;
  if (d->stats_mode != 0) {
    if (d->stats_mode != 1) return gan_set_error(-1, "conv: stats_mode %d (0 | 1)", d->stats_mode);
    if (d->dtype != GAN_BF16 || d->act != GAN_ACT_NONE || d->out_sy != 1 || d->out_sx != 1 || d->bias)
      return gan_set_error(-1, "conv: stats_mode 1 needs bf16 operands, act = none, no bias and a dense output");
    if (!d->stats || !d->mask) return gan_set_error(-1, "conv: stats_mode 1 needs stats and its operand (mask)");
    if (d->mask_y0 + d->Ho > d->mask_Hp || d->mask_x0 + d->Wo > d->mask_Wp) return gan_set_error(-1, "conv: stats_mode 1: the operand does not cover the output domain");
  }
  a.mask_Hp = d->mask_Hp; a.mask_Wp = d->mask_Wp; a.mask_y0 = d->mask_y0; a.mask_x0 = d->mask_x0;
  // diagnostic environment (stamp buffer, static-schedule switch): read once per process, not per launch
  static unsigned long long* const stamps_env = [] { const char* e = getenv("GAN_PATCH_STAMPS"); return e ? (unsigned long long*)strtoull(e, nullptr, 0) : nullptr; }();
  // GAN_PATCH_STAMPS_SEL="rows,cols,batch,chain,taps": only launches of that tile, batch and epilogue stamp (tools/step_clock.py reads one kernel
  // of the running step)
  static const std::array<int, 5> stamps_sel = [] {
    std::array<int, 5> v{0, 0, 0, 0, 0};
    if (const char* e = getenv("GAN_PATCH_STAMPS_SEL")) sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
    return v;
  }();
  a.stamps = (stamps_sel[0] == 0 || (stamps_sel[0] == BM && stamps_sel[1] == BN && stamps_sel[2] == d->B && stamps_sel[3] == (a.smode != 0) && stamps_sel[4] == d->ntaps)) ? stamps_env : nullptr;
  const int grid = a.tiles < ncu ? a.tiles : ncu;
  // the dynamic-LDS limit is a per-device function attribute: one bit per device, set on that device's first launch
  static std::atomic<uint64_t> attr_devs{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return gan_set_error(-2, "conv_patch: hipGetDevice failed");
  const uint64_t dev_bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & dev_bit)) {
    auto raise = [](const void* f, int bytes) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess; };
    if (!raise((const void*)conv_patch_kernel<256, 2, 0>, lds_bytes(7)) || !raise((const void*)conv_patch_kernel<288, 4, 0>, lds_bytes(7)) ||
        !raise((const void*)conv_patch_kernel<256, 2, 9>, lds_bytes(7)) || !raise((const void*)conv_patch_fp8_kernel, lds_bytes(7)) ||
        !raise((const void*)conv_patch_kernel<256, 2, 4>, lds_bytes(7)) || !raise((const void*)conv_patch_kernel<256, 2, 2>, lds_bytes(7)) ||
        !raise((const void*)conv_patch_kernel<256, 2, 16>, lds_bytes(7)) || !raise((const void*)conv_patch_kernel<256, 2, 4, false, 9>, lds_bytes(9)) ||
        !raise((const void*)conv_patch_kernel<256, 2, 2, false, 9>, lds_bytes(9)) ||
        !raise((const void*)conv_patch_kernel<256, 2, 0, false, 9>, lds_bytes(9)) || !raise((const void*)conv_patch_fp8_wide_kernel, lds_bytes(9)) ||
        !raise((const void*)conv_patch_kernel<256, 4, 0, false, 7, 256>, lds_bytes(7)) || !raise((const void*)conv_patch_kernel<288, 4, 0, false, 7, 256>, lds_bytes(7)) ||
        !raise((const void*)conv_patch_kernel<256, 4, 0, false, 9, 256>, lds_bytes(9)) ||
        !raise((const void*)conv_patch_bwdchain_kernel<256, 2, 7, 128>, lds_bytes(7)) || !raise((const void*)conv_patch_bwdchain_kernel<288, 4, 7, 128>, lds_bytes(7)) ||
        !raise((const void*)conv_patch_bwdchain_kernel<256, 2, 9, 128>, lds_bytes(9)) || !raise((const void*)conv_patch_bwdchain_kernel<256, 4, 7, 256>, lds_bytes(7)) ||
        !raise((const void*)conv_patch_bwdchain_kernel<288, 4, 7, 256>, lds_bytes(7)) || !raise((const void*)conv_patch_bwdchain_kernel<256, 4, 9, 256>, lds_bytes(9)))
      return gan_set_error(-2, "conv_patch: cannot raise the dynamic LDS limit to %d bytes", lds_bytes(9));
    attr_devs.fetch_or(dev_bit, std::memory_order_release);
  }
  // The static 3x3 schedule, 256-row tile only.  Measured (s_memtime): 15.5 k -> 13.0 k cycles per slab, but the denser issue
  // stream clocks lower (2.04 -> 1.88 GHz), so the forward gains 3 % wall (69.5 -> 67.2 us = 1.15 PFLOP/s); on the 288-row tile,
  // whose 9 fragment addresses per tap do not fit in registers, it lost 9 % and is not instantiated.
  const bool wide = ((variant >> 24) & 0xf) == 9, st9 = (variant >> 29) & 1;
  const int stx = (variant >> 30) & 3;
  if (a.smode != 0) {
    if (BN == 256) {
      if (BM == 256 && wide) hipLaunchKernelGGL((conv_patch_bwdchain_kernel<256, 4, 9, 256>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
      else if (BM == 256) hipLaunchKernelGGL((conv_patch_bwdchain_kernel<256, 4, 7, 256>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
      else hipLaunchKernelGGL((conv_patch_bwdchain_kernel<288, 4, 7, 256>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    } else if (BM == 256) {
      if (wide) hipLaunchKernelGGL((conv_patch_bwdchain_kernel<256, 2, 9, 128>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
      else hipLaunchKernelGGL((conv_patch_bwdchain_kernel<256, 2, 7, 128>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    } else {
      hipLaunchKernelGGL((conv_patch_bwdchain_kernel<288, 4, 7, 128>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    }
  } else if (fp8) {
    if (wide) hipLaunchKernelGGL(conv_patch_fp8_wide_kernel, dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
    else hipLaunchKernelGGL(conv_patch_fp8_kernel, dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
  } else if (BN == 256) {
    if (BM == 256 && wide) hipLaunchKernelGGL((conv_patch_kernel<256, 4, 0, false, 9, 256>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
    else if (BM == 256) hipLaunchKernelGGL((conv_patch_kernel<256, 4, 0, false, 7, 256>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<288, 4, 0, false, 7, 256>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
  } else if (BM == 256) {
    if (stx == 1 && wide) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 4, false, 9>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
    else if (stx == 2 && wide) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 2, false, 9>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
    else if (stx == 1) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 4>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    else if (stx == 2) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 2>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    else if (stx == 3) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 16>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    else if (wide) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 0, false, 9>), dim3(grid), dim3(NTHR), lds_bytes(9), s, a);
    else if (st9) hipLaunchKernelGGL((conv_patch_kernel<256, 2, 9>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
    else hipLaunchKernelGGL((conv_patch_kernel<256, 2, 0>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
  } else {
    hipLaunchKernelGGL((conv_patch_kernel<288, 4, 0>), dim3(grid), dim3(NTHR), lds_bytes(7), s, a);
  }
  if (hipGetLastError() != hipSuccess) return gan_set_error(-2, "conv_patch: launch failed");
  return 0;
}
