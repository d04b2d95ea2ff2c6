// 7x7, stride 1, 64 channels -> <= 8 channels: the generator's output convolution (generator_resnet_attn.py:157-162, 64 -> 3, tanh)
// and the input gradient of its first convolution (:110-116, 64 -> 3), bf16 operands, fp32 accumulation.
//
// The generic kernel stages the activation tile once per TAP (49 times); here a block stages the 22x22-pixel window of its 16x16
// output tile ONCE (62 KB, pixel pitch 144 B so that the 16 lanes of a ds_read_b128 group hit all 64 banks) and every tap is an LDS
// offset.  N is only 3 (padded to the MFMA's 16).  The 14 (column shift, half-channel group) pairs are dealt to the four waves; a wave
// keeps the weight fragments of its pairs for all 7 tap rows in registers (read once from the generic [Nw=16][49][64] packing), reads
// each activation fragment once and uses it for every tap row, accumulates all 16 rows of the tile, and the four partial tiles are
// summed through LDS before bias / tanh / store.  Two blocks fit a CU (2 x 70 KB LDS, <= 256 VGPRs).
#include "common.h"

namespace {

constexpr int TS = 16, PW = TS + 6, PITCH = 144, PATCH_BYTES = PW * PW * PITCH;   // 69,696 B
constexpr int NJ = 4;

struct Win7Args {
  const char* in; const char* w; const float* bias; char* out;
  int B, Ho, Wo, tiles_x, tiles_y;
  int in_Hp, in_Wp, in_y0, in_x0, ty0, tx0;      // first tap of output (0,0) reads padded pixel (in_y0 + ty0, in_x0 + tx0)
  int out_Hp, out_Wp, out_y0, out_x0;
  int act;
  float* stats;                                  // 3 -> 64 kernel only: InstanceNorm partials [B][tiles per image][64][2], or null
};

__global__ __launch_bounds__(256, 2) void conv_win7_kernel(Win7Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // consecutive logical tiles on one XCD (its L2 then serves the halo overlap between neighbouring windows)
  const int G = gridDim.x;
  const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int tile = xcd < r ? xcd * (q + 1) + k : r * (q + 1) + (xcd - r) * q + k;
  const int per_img = a.tiles_x * a.tiles_y;
  const int b = tile / per_img, t2 = tile - b * per_img;
  const int oy0 = (t2 / a.tiles_x) * TS, ox0 = (t2 % a.tiles_x) * TS;

  // ---- work split: the 14 (column shift kx, half-channel group c) pairs are dealt to the waves (4,4,3,3); a wave holds the weight
  //      fragments of its pairs for all 7 tap rows (A operand: row n = lane&15, k = 8*(lane>>4)..+7)
  u32x4_t bw[NJ][7];
  {
    const char* wl = a.w + (lane & 15) * (49 * 128) + (lane >> 4) * 16;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int id = wave + 4 * j, kx = min(id >> 1, 6), c = id & 1;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(wl + (ky * 7 + kx) * 128 + c * 64);
        if (id >= 14) v = u32x4_t{0u, 0u, 0u, 0u};
        bw[j][ky] = v;
      }
    }
  }
  // ---- stage the window: 484 pixels x 8 chunks of 16 B; pixels outside the allocation read as zero (they only feed masked outputs).
  //      All 16 loads of a lane are issued before the first LDS write: with two blocks (8 waves) per CU nothing else hides their latency.
  {
    const int ay0 = oy0 + a.in_y0 + a.ty0, ax0 = ox0 + a.in_x0 + a.tx0;
    const char* img = a.in + (int64_t)b * a.in_Hp * a.in_Wp * 128;
    constexpr int NIT = (PW * PW * 8 + 255) / 256;      // 16
    u32x4_t st[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int p = i >> 3, ck = i & 7;
      const int py = p / PW, px = p - py * PW;
      const int ay = ay0 + py, ax = ax0 + px;
      const bool ok = i < PW * PW * 8 && ay >= 0 && ay < a.in_Hp && ax >= 0 && ax < a.in_Wp;
      const int64_t off = ok ? ((int64_t)ay * a.in_Wp + ax) * 128 + ck * 16 : 0;      // a safe address for the lanes that are masked
      u32x4_t v = *reinterpret_cast<const u32x4_t*>(img + off);
      if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
      st[it] = v;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      if (i < PW * PW * 8) *reinterpret_cast<u32x4_t*>(lds + (i >> 3) * PITCH + (i & 7) * 16) = st[it];
    }
  }
  __syncthreads();

  // ---- main loop: B operand = activations, column j = pixel x = lane&15 of row m, k = 8*(lane>>4)..+7
  f32x4_t acc[TS];
#pragma unroll
  for (int m = 0; m < TS; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // an activation fragment -- window row R, shift kx, group c -- is read once and feeds every tap row ky with output row m = R - ky:
  // 22 reads and up to 7 x 16 MFMAs per pair instead of one read per MFMA (the loop was LDS-bound)
  const char* la = lds + (lane & 15) * PITCH + (lane >> 4) * 16;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int id = wave + 4 * j, kx = min(id >> 1, 6), c = id & 1;
    const char* lj = la + kx * PITCH + c * 64;
#pragma unroll
    for (int R = 0; R < PW; ++R) {
      const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(lj + R * (PW * PITCH));
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int m = R - ky;
        if (m >= 0 && m < TS)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[j][ky]), __builtin_bit_cast(bf16x8_t, xa), acc[m], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                       // every wave is done with the window: its space becomes the reduction buffer
  // acc[m]: row n = 4*(lane>>4)+r, column x = lane&15  ->  red[wave][m][x][n]
  {
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int m = 0; m < TS; ++m)
      *reinterpret_cast<f32x4_t*>(red + (((wave * TS + m) * TS + (lane & 15)) * 16 + (lane >> 4) * 4)) = acc[m];
  }
  __syncthreads();
  // ---- one output pixel per thread: sum the four partial tiles, bias, activation, 8 bf16 channels = 16 bytes
  {
    const float* red = reinterpret_cast<const float*>(lds);
    const int m = tid >> 4, x = tid & 15;
    float v[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(red + (((w2 * TS + m) * TS + x) * 16));
      const f32x4_t p1 = *reinterpret_cast<const f32x4_t*>(red + (((w2 * TS + m) * TS + x) * 16 + 4));
      v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    }
    const int oy = oy0 + m, ox = ox0 + x;
    if (oy < a.Ho && ox < a.Wo) {
#pragma unroll
      for (int n = 0; n < 8; ++n) v[n] = act_apply(v[n], a.act);
      bf16_t* o = reinterpret_cast<bf16_t*>(a.out) + (((int64_t)b * a.out_Hp + oy + a.out_y0) * a.out_Wp + ox + a.out_x0) * 8;
      Chunk<bf16_t>::store(o, v);
    }
  }
}
// sum over the 16 lanes of a DPP row (conv_patch.hip: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum_w7(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});
  v += dpp(v, std::integral_constant<int, 0x4E>{});
  v += dpp(v, std::integral_constant<int, 0x141>{});
  v += dpp(v, std::integral_constant<int, 0x140>{});
  return v;
}

// ---- the mirror case, 3 (padded 8) -> 64 channels: the generator's first convolution (generator_resnet_attn.py:110-116) and the input
// gradient of its output convolution.  A pixel is one 16-byte chunk, so a K-step of 32 is FOUR taps (lane group g = lane>>4 reads tap
// 4*ks+g); 49 taps = 13 K-steps against the generic kernel's 56 tap slots, and the 22x22 window is 7.7 KB.  Wave w owns output channels
// 16w..16w+15 for all 16 rows of the tile (13 weight fragments in registers); results go straight to HBM, 8 bytes per lane.
constexpr int KS3 = 13;

__global__ __launch_bounds__(256, 2) void conv_win7_from3_kernel(Win7Args a, int wtaps) {
  __shared__ __attribute__((aligned(16))) char lds[(PW * PW + 16) * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int q = G >> 3, r = G & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int tile = xcd < r ? xcd * (q + 1) + k : r * (q + 1) + (xcd - r) * q + k;
  const int per_img = a.tiles_x * a.tiles_y;
  const int b = tile / per_img, t2 = tile - b * per_img;
  const int oy0 = (t2 / a.tiles_x) * TS, ox0 = (t2 % a.tiles_x) * TS;
  // weights: row n = 16*wave + (lane&15) of [64][wtaps][8]; K-step ks, lane group g -> tap 4*ks+g (taps >= 49 are zero in the packing)
  u32x4_t bw[KS3];
  {
    const char* wl = a.w + ((int64_t)(wave * 16 + (lane & 15)) * wtaps + (lane >> 4)) * 16;
#pragma unroll
    for (int ks = 0; ks < KS3; ++ks) bw[ks] = *reinterpret_cast<const u32x4_t*>(wl + ks * 64);
  }
  {
    const int ay0 = oy0 + a.in_y0 + a.ty0, ax0 = ox0 + a.in_x0 + a.tx0;
    const char* img = a.in + (int64_t)b * a.in_Hp * a.in_Wp * 16;
    u32x4_t st[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int p = tid + it * 256;
      const int py = p / PW, px = p - py * PW;
      const int ay = ay0 + py, ax = ax0 + px;
      const bool ok = p < PW * PW && ay >= 0 && ay < a.in_Hp && ax >= 0 && ax < a.in_Wp;
      u32x4_t v = *reinterpret_cast<const u32x4_t*>(img + (ok ? ((int64_t)ay * a.in_Wp + ax) * 16 : 0));
      if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
      st[it] = v;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
      if (tid + it * 256 < PW * PW + 16) *reinterpret_cast<u32x4_t*>(lds + (tid + it * 256) * 16) = st[it];   // +16: slack read by the padding taps
  }
  __syncthreads();
  f32x4_t acc[TS];
#pragma unroll
  for (int m = 0; m < TS; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS3; ++ks) {
    const int t = min(4 * ks + g, 48);                      // taps 49..51: zero weights, any in-window address will do
    const int ky = t / 7, kx = t - ky * 7;
    const char* lj = lds + ((ky * PW + kx) + (lane & 15)) * 16;
#pragma unroll
    for (int m = 0; m < TS; ++m) {
      const u32x4_t xa = *reinterpret_cast<const u32x4_t*>(lj + m * (PW * 16));
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bw[ks]), __builtin_bit_cast(bf16x8_t, xa), acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);      // keep one K-step's 16 fragment reads in flight, not all 13 K-steps' (register pressure)
  }
  // acc[m]: channel n = 16*wave + 4*g + r, pixel x = lane&15 of row m
  const int n0 = wave * 16 + g * 4, x = lane & 15;
  float bias[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bias[i] = a.bias ? a.bias[n0 + i] : 0.f;
  bf16_t* dst = reinterpret_cast<bf16_t*>(a.out) + (((int64_t)b * a.out_Hp + oy0 + a.out_y0) * a.out_Wp + ox0 + x + a.out_x0) * 64 + n0;
  const bool xok = ox0 + x < a.Wo;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < TS; ++m) {
    const float v0 = acc[m].x + bias[0], v1 = acc[m].y + bias[1], v2 = acc[m].z + bias[2], v3 = acc[m].w + bias[3];
    uint2 o;                                                 // no activation on this path (gan_conv_win7_ok)
    o.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
    o.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
    if (xok && oy0 + m < a.Ho) {
      *reinterpret_cast<uint2*>(dst + (int64_t)m * a.out_Wp * 64) = o;
      ssum[0] += v0; ssum[1] += v1; ssum[2] += v2; ssum[3] += v3;          // statistics of the fp32 result, before its rounding
      ssq[0] += v0 * v0; ssq[1] += v1 * v1; ssq[2] += v2 * v2; ssq[3] += v3 * v3;
    }
  }
  if (a.stats) {      // per (image, tile, channel) sum and sum of squares: lanes of one DPP row hold the 16 pixels of a row for 4 channels
#pragma unroll
    for (int i = 0; i < 4; ++i) { ssum[i] = row16_sum_w7(ssum[i]); ssq[i] = row16_sum_w7(ssq[i]); }
    if (x == 0) {
      float* sp = a.stats + (((int64_t)b * per_img + t2) * 64 + n0) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float2*>(sp + 2 * i) = make_float2(ssum[i], ssq[i]);
    }
  }
}
}  // namespace

// Host predicate: the descriptor is a 7x7 window convolution this kernel covers.  win_ty0 / win_tx0 give the first tap's position
// (tapoff[t] must be ((win_ty0 + t/7) * in_Wp + win_tx0 + t%7) * Cin, which max_tapoff lets the library cross-check).
extern "C" int gan_conv_win7_ok(const gan_conv_desc* d) {
  if (!d || d->dtype != GAN_BF16) return 0;
  const bool to3 = d->Cin == 64 && d->ntaps == 49 && d->Nw == 16 && d->Nst == 8 && d->out_C == 8;
  const bool from3 = d->Cin == 8 && d->ntaps >= 52 && d->Nw == 64 && d->Nst == 64 && d->out_C == 64;
  if (!to3 && !from3) return 0;
  if (d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1 || d->mask || (d->stats && !from3)) return 0;
  if (d->act != GAN_ACT_NONE && !(to3 && d->act == GAN_ACT_TANH)) return 0;
  if (d->max_tapoff != ((d->win_ty0 + 6) * d->in_Wp + d->win_tx0 + 6) * d->Cin) return 0;
  { const char* e = getenv("GAN_NO_WIN7"); if (e && atoi(e)) return 0; }
  return 1;
}

// tiles per image for which the 3 -> 64 kernel writes InstanceNorm partials to d->stats (0: this descriptor cannot)
int gan_conv_win7_stats_parts(const gan_conv_desc* d) {
  if (d->w_layout != 2 || d->Cin != 8 || d->Nst != 64 || d->act != GAN_ACT_NONE) return 0;
  return ((d->Wo + TS - 1) / TS) * ((d->Ho + TS - 1) / TS);
}

int gan_conv_win7_launch(const gan_conv_desc* d, hipStream_t s) {
  GAN_CHECK(gan_conv_win7_ok(d), "conv: w_layout 2 set on a descriptor the 7x7 window kernel does not cover");
  // every tap of every real output pixel must lie inside the allocation (the planner's halos guarantee it; checked here)
  GAN_CHECK(d->in_y0 + d->win_ty0 >= 0 && d->in_x0 + d->win_tx0 >= 0 && d->Ho - 1 + d->in_y0 + d->win_ty0 + 6 < d->in_Hp &&
                d->Wo - 1 + d->in_x0 + d->win_tx0 + 6 < d->in_Wp, "conv(7x7 window): taps reach outside the input allocation");
  Win7Args a;
  a.in = (const char*)d->in; a.w = (const char*)d->w; a.bias = d->bias; a.out = (char*)d->out;
  a.B = d->B; a.Ho = d->Ho; a.Wo = d->Wo; a.tiles_x = (d->Wo + TS - 1) / TS; a.tiles_y = (d->Ho + TS - 1) / TS;
  a.in_Hp = d->in_Hp; a.in_Wp = d->in_Wp; a.in_y0 = d->in_y0; a.in_x0 = d->in_x0; a.ty0 = d->win_ty0; a.tx0 = d->win_tx0;
  a.out_Hp = d->out_Hp; a.out_Wp = d->out_Wp; a.out_y0 = d->out_y0; a.out_x0 = d->out_x0;
  a.act = d->act; a.stats = d->stats;
  const int64_t blocks = (int64_t)d->B * a.tiles_x * a.tiles_y;
  GAN_CHECK(blocks < (1ll << 31), "conv(7x7 window): too many tiles");
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_win7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PATCH_BYTES + 64) != hipSuccess)
      return gan_set_error(-2, "conv(7x7 window): cannot raise the dynamic LDS limit to %d bytes", PATCH_BYTES + 64);
    attr_set = true;
  }
  if (d->Cin == 8) {
    hipLaunchKernelGGL(conv_win7_from3_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, d->ntaps);
    GAN_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(conv_win7_kernel, dim3((unsigned)blocks), dim3(256), PATCH_BYTES + 64, s, a);
  GAN_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient, 64 -> 3 channel layer
// dW[co][ky][kx][ci] = sum over pixels of dY[pixel][co] * X[pixel + (ky,kx)][ci] for the generator's output convolution
// (generator_resnet_attn.py:157-162): 49 taps x 64 channels x 3 (padded 8) outputs, reduced over B*H*W pixels.  The generic kernel
// re-stages the X tile per tap; here a block walks 16x16 pixel tiles, stages each tile's 22x22 window once (the forward kernel's layout)
// and keeps the WHOLE gradient in registers: wave w owns input channels 16w..16w+15 for all 49 taps = 49 accumulator tiles.
// Pixels are the K dimension, so both operands are read through ds_read_b64_tr_b16 (channel-major fragments from pixel-major LDS).
// Every block writes one partial [8][49][64] slab; gan_wgrad_reduce sums the slabs (deterministic, no atomics).
namespace {

struct Wg7Args {
  const char* x; const char* g; float* part;
  int B, Ho, Wo, tiles_x, tiles_y, ntiles;
  int x_Hp, x_Wp, x_y0, x_x0;       // tap (0,0) of pixel (0,0) reads padded x pixel (x_y0, x_x0)
  int g_Hp, g_Wp, g_y0, g_x0;
};

constexpr int GP = 32;                                   // bytes per pixel of the dY tile in LDS: 8 real + 8 zero channel slots
// window pitch 160 B here: a transposing read's 32 lanes touch 8 consecutive pixels x 32 B, and 40 banks per pixel spreads their
// starts over 0,40,16,56,32,8,48,24 -- every bank once (the forward kernel's 16-lane b128 groups want 144 instead)
constexpr int WPITCH = 160, WPATCH = PW * PW * WPITCH;
constexpr int WG7_LDS = WPATCH + 64 + TS * TS * GP;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__global__ __launch_bounds__(256, 1) void wgrad_win7_kernel(Wg7Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* lx = lds;
  char* lg = lds + WPATCH + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4, q = fi >> 2, p = fi & 3;
  f32x4_t acc[49];
#pragma unroll
  for (int t = 0; t < 49; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int per_img = a.tiles_x * a.tiles_y;
  // lane (q, p) of a 16-lane group addresses pixel q of a 4-pixel block, channels 4p..4p+3.  The K index 8*fg + 4*h + q of a K-step
  // (32 pixels = a row pair) is mapped to the pixel (row fg>>1, x = 8*h + 4*(fg&1) + q) -- any bijection does, as long as both operands
  // use it -- so that one read instruction's lane groups fg = 0,1 cover 8 CONSECUTIVE pixels (bank-conflict free at pitch 160)
  int xoff[2], goff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = fg >> 1, xx = h * 8 + (fg & 1) * 4 + q;
    xoff[h] = (r * PW + xx) * WPITCH + (wave * 16 + p * 4) * 2;
    goff[h] = (r * TS + xx) * GP + p * 8;
  }
  // The next tile's window and dY chunk are fetched into registers while the current tile is multiplied (one block per CU: nothing
  // else would hide the global latency), and written to LDS when the current tile's reads are done.
  constexpr int NIT = (PW * PW * 8 + 255) / 256;
  u32x4_t st[NIT], gv;
  auto fetch = [&](int tile) {
    const int b = tile / per_img, t2 = tile - b * per_img;
    const int oy0 = (t2 / a.tiles_x) * TS, ox0 = (t2 % a.tiles_x) * TS;
    const int ay0 = oy0 + a.x_y0, ax0 = ox0 + a.x_x0;
    const char* img = a.x + (int64_t)b * a.x_Hp * a.x_Wp * 128;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int px_ = i >> 3, ck = i & 7;
      const int py = px_ / PW, px = px_ - py * PW;
      const int ay = ay0 + py, ax = ax0 + px;
      const bool ok = i < PW * PW * 8 && ay >= 0 && ay < a.x_Hp && ax >= 0 && ax < a.x_Wp;
      u32x4_t v = *reinterpret_cast<const u32x4_t*>(img + (ok ? ((int64_t)ay * a.x_Wp + ax) * 128 + ck * 16 : 0));
      if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
      st[it] = v;
    }
    // dY tile: pixels outside the image contribute nothing (zero)
    const int gy = oy0 + (tid >> 4), gx = ox0 + (tid & 15);
    gv = u32x4_t{0u, 0u, 0u, 0u};
    if (gy < a.Ho && gx < a.Wo) gv = *reinterpret_cast<const u32x4_t*>(a.g + (((int64_t)b * a.g_Hp + gy + a.g_y0) * a.g_Wp + gx + a.g_x0) * 16);
  };
  if ((int)blockIdx.x < a.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's LDS reads are done
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      if (i < PW * PW * 8) *reinterpret_cast<u32x4_t*>(lx + (i >> 3) * WPITCH + (i & 7) * 16) = st[it];
    }
    *reinterpret_cast<u32x4_t*>(lg + tid * GP) = gv;
    *reinterpret_cast<u32x4_t*>(lg + tid * GP + 16) = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);
    // dY fragments of the 8 K-steps (row pairs 2ks, 2ks+1) stay in registers; an X fragment -- window row pair R, column shift kx -- is
    // read ONCE and feeds every (tap row ky, K-step ks) with 2ks + ky == R: 147 fragment reads per tile instead of 392
    bf16x8_t gfr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const s16x4_t g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lg + ks * (2 * TS * GP) + goff[0]));
      const s16x4_t g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lg + ks * (2 * TS * GP) + goff[1]));
      gfr[ks] = bf16x8_t{g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    }
#pragma unroll
    for (int R = 0; R <= 20; ++R) {
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int tb = (R * PW + kx) * WPITCH;
        const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lx + tb + xoff[0]));
        const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lx + tb + xoff[1]));
        const bf16x8_t xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int ky = R & 1; ky < 7; ky += 2) {
          const int ks = (R - ky) / 2;
          if (R - ky >= 0 && ks < 8) acc[ky * 7 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xv, gfr[ks], acc[ky * 7 + kx], 0, 0, 0);   // D[channel][co]
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // acc[t]: row i = channel 16*wave + 4*fg + r, column j = co = fi  ->  part[block][co][t][channel]
  if (fi < 8) {
    float* dst = a.part + ((int64_t)blockIdx.x * 8 + fi) * (49 * 64) + wave * 16 + fg * 4;
#pragma unroll
    for (int t = 0; t < 49; ++t) *reinterpret_cast<f32x4_t*>(dst + t * 64) = acc[t];
  }
}
}  // namespace

// ---- the mirror case: weight gradient of the generator's FIRST convolution (3, padded 8, -> 64 channels).  Rows of the MFMA are the 64
// output channels (wave w owns 16 of them), columns are (tap, input channel): a pixel of X is one 16-byte chunk, so the 16 columns of a
// transposing read are the 8 channels of pixel P and of pixel P+1 = two horizontally adjacent taps.  28 accumulator tiles per wave
// (7 tap rows x 4 column pairs, the 8th column of the last pair is padding); X fragments are shared across tap rows as above.
namespace {
constexpr int GP64 = 160;                                 // dY pixel pitch in LDS (64 channels = 128 B, padded: see WPITCH)
constexpr int XW_BYTES = (PW * PW + 8) * 16;
constexpr int WG7B_LDS = XW_BYTES + TS * TS * GP64;

__global__ __launch_bounds__(256, 2) void wgrad_win7_from3_kernel(Wg7Args a) {
  __shared__ __attribute__((aligned(16))) char lds[WG7B_LDS];
  char* lx = lds;
  char* lg = lds + XW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fg = lane >> 4, q = fi >> 2, p = fi & 3;
  f32x4_t acc[7][4];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) acc[ky][pr] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int per_img = a.tiles_x * a.tiles_y;
  int xoff[2], goff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = fg >> 1, xx = h * 8 + (fg & 1) * 4 + q;           // the K-index -> pixel bijection of wgrad_win7_kernel
    xoff[h] = (r * PW + xx) * 16 + p * 8;
    goff[h] = (r * TS + xx) * GP64 + (wave * 16 + p * 4) * 2;
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / per_img, t2 = tile - b * per_img;
    const int oy0 = (t2 / a.tiles_x) * TS, ox0 = (t2 % a.tiles_x) * TS;
    __syncthreads();
    {
      const int ay0 = oy0 + a.x_y0, ax0 = ox0 + a.x_x0;
      const char* img = a.x + (int64_t)b * a.x_Hp * a.x_Wp * 16;
      u32x4_t sx[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int px_ = tid + it * 256;
        const int py = px_ / PW, px = px_ - py * PW;
        const int ay = ay0 + py, ax = ax0 + px;
        const bool ok = px_ < PW * PW && ay >= 0 && ay < a.x_Hp && ax >= 0 && ax < a.x_Wp;
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(img + (ok ? ((int64_t)ay * a.x_Wp + ax) * 16 : 0));
        if (!ok) v = u32x4_t{0u, 0u, 0u, 0u};
        sx[it] = v;
      }
      // dY tile: 256 pixels x 8 chunks; one pixel per thread, 8 loads in flight
      const int gy = oy0 + (tid >> 4), gx = ox0 + (tid & 15);
      const bool gok = gy < a.Ho && gx < a.Wo;
      const char* gp = a.g + (gok ? (((int64_t)b * a.g_Hp + gy + a.g_y0) * a.g_Wp + gx + a.g_x0) * 128 : 0);
      u32x4_t sg[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { sg[c] = *reinterpret_cast<const u32x4_t*>(gp + c * 16); if (!gok) sg[c] = u32x4_t{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int it = 0; it < 2; ++it)
        if (tid + it * 256 < PW * PW + 8) *reinterpret_cast<u32x4_t*>(lx + (tid + it * 256) * 16) = sx[it];
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4_t*>(lg + tid * GP64 + c * 16) = sg[c];
    }
    __syncthreads();
    bf16x8_t gfr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const s16x4_t g0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lg + ks * (2 * TS * GP64) + goff[0]));
      const s16x4_t g1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lg + ks * (2 * TS * GP64) + goff[1]));
      gfr[ks] = bf16x8_t{g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
    }
#pragma unroll
    for (int R = 0; R <= 20; ++R) {
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const int tb = (R * PW + 2 * pr) * 16;
        const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lx + tb + xoff[0]));
        const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lx + tb + xoff[1]));
        const bf16x8_t xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
        for (int ky = R & 1; ky < 7; ky += 2) {
          const int ks = (R - ky) / 2;
          if (R - ky >= 0 && ks < 8) acc[ky][pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr[ks], xv, acc[ky][pr], 0, 0, 0);   // D[co][(tap, ci)]
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // acc[ky][pr]: row co = 16*wave + 4*fg + r, column fi -> tap (ky, 2*pr + (fi>>3)), input channel fi&7  ->  part[block][co][tap][ci]
  const int kxo = fi >> 3, ci = fi & 7;
  float* dst = a.part + ((int64_t)blockIdx.x * 64 + wave * 16 + fg * 4) * (49 * 8) + ci;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const int kx = 2 * pr + kxo;
      if (kx < 7) {
        float* o = dst + (ky * 7 + kx) * 8;
        o[0] = acc[ky][pr].x; o[49 * 8] = acc[ky][pr].y; o[2 * 49 * 8] = acc[ky][pr].z; o[3 * 49 * 8] = acc[ky][pr].w;
      }
    }
}
}  // namespace

// slabs (= blocks) the window weight-gradient kernel writes for this descriptor; 0: it does not qualify
extern "C" int gan_wgrad_win7_splits(const gan_wgrad_desc* d) {
  if (!d || d->dtype != GAN_BF16 || d->ntaps != 49) return 0;
  const bool to3 = d->Cx == 64 && d->N == 8 && d->g_C == 8, from3 = d->Cx == 8 && d->N == 64 && d->g_C == 64;
  if (!to3 && !from3) return 0;
  if (d->x_sy != 1 || d->x_sx != 1 || d->g_sy != 1 || d->g_sx != 1) return 0;
  if (d->max_tapoff != (6 * d->x_Wp + 6) * d->Cx) return 0;           // 49 row-major taps from (x_y0, x_x0)
  { const char* e = getenv("GAN_NO_WIN7"); if (e && atoi(e)) return 0; }
  const int64_t tiles = (int64_t)d->B * ((d->Ho + TS - 1) / TS) * ((d->Wo + TS - 1) / TS);
  return (int)(tiles < 256 ? tiles : 256);
}

int gan_wgrad_win7_launch(const gan_wgrad_desc* d, hipStream_t s) {
  const int ns = gan_wgrad_win7_splits(d);
  GAN_CHECK(ns > 0 && d->nsplit == ns, "wgrad: variant 2 needs nsplit = gan_wgrad_win7_splits() = %d, got %d", ns, d->nsplit);
  GAN_CHECK(d->x && d->g && d->part, "wgrad(7x7 window): null pointer");
  GAN_CHECK(d->x_y0 >= 0 && d->x_x0 >= 0 && d->Ho - 1 + d->x_y0 + 6 < d->x_Hp && d->Wo - 1 + d->x_x0 + 6 < d->x_Wp,
            "wgrad(7x7 window): taps reach outside the x allocation");
  Wg7Args a;
  a.x = (const char*)d->x; a.g = (const char*)d->g; a.part = d->part;
  a.B = d->B; a.Ho = d->Ho; a.Wo = d->Wo; a.tiles_x = (d->Wo + TS - 1) / TS; a.tiles_y = (d->Ho + TS - 1) / TS;
  a.ntiles = d->B * a.tiles_x * a.tiles_y;
  a.x_Hp = d->x_Hp; a.x_Wp = d->x_Wp; a.x_y0 = d->x_y0; a.x_x0 = d->x_x0;
  a.g_Hp = d->g_Hp; a.g_Wp = d->g_Wp; a.g_y0 = d->g_y0; a.g_x0 = d->g_x0;
  if (d->Cx == 8) {
    hipLaunchKernelGGL(wgrad_win7_from3_kernel, dim3(ns), dim3(256), 0, s, a);
    GAN_LAUNCH_CHECK();
    return 0;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)wgrad_win7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WG7_LDS) != hipSuccess)
      return gan_set_error(-2, "wgrad(7x7 window): cannot raise the dynamic LDS limit to %d bytes", WG7_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_win7_kernel, dim3(ns), dim3(256), WG7_LDS, s, a);
  GAN_LAUNCH_CHECK();
  return 0;
}
