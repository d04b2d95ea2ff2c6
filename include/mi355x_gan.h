/*
 * mi355x_gan.h -- C ABI of libmi355x_gan.so: the MI355X (gfx950) GAN training inner loop.
 *
 * The reference (Cameronr11/GAN-Variant-Research) has no FFI of its own: its operator API is
 * torch.nn -> ATen.  Each entry point below replaces the ATen ops one reference call site dispatches
 * (cited as file:line relative to the reference root).  All pointers are DEVICE pointers unless a
 * parameter says "host"; `stream` is a hipStream_t passed as void* (NULL = default stream).  The library
 * never allocates, frees or synchronises; workspaces are supplied by the caller.  Every function returns
 * 0 on success or a negative error code; gan_last_error() describes the last failure of this thread.
 *
 * Data layout in HBM ("halo-NHWC"): an activation is [B][Hp][Wp][C] with C a multiple of 8 (zero-filled
 * pad channels) and an optional spatial halo already materialised by the producer (reflect or zero), so
 * every convolution is a bounds-check-free "valid" implicit GEMM over 16-byte channel chunks.
 */
#ifndef MI355X_GAN_H
#define MI355X_GAN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* GAN_FP8: OCP e4m3 ("e4m3fn": no infinities, max 448), one byte per element.  Only operand COPIES of the bottleneck convolutions are
 * fp8 (BASELINE.json configs[4]): gan_quantize_fp8 writes activations / output gradients, gan_pack_weight the weights; every result,
 * statistic and master weight stays bf16 / fp32.  An fp8 view has C % 16 == 0. */
enum { GAN_F32 = 0, GAN_BF16 = 1, GAN_FP8 = 2 };
enum { GAN_ACT_NONE = 0, GAN_ACT_RELU = 1, GAN_ACT_LRELU = 2, GAN_ACT_TANH = 3 };
/* GAN_HALO_REPLICATE (nn.ReplicationPad2d, generator_resnet_attn.py:26-27,45-46: an option the shipped configs do not use): accepted by
 * gan_nchw_to_view only; its gradient is gan_pad_fold */
enum { GAN_HALO_NONE = 0, GAN_HALO_ZERO = 1, GAN_HALO_REFLECT = 2, GAN_HALO_REPLICATE = 3 };

/* A halo-NHWC activation: `ptr` addresses element [0][0][0][0] of the allocation; the logical HxW image
 * starts at (y0,x0).  dtype: GAN_F32 or GAN_BF16. */
typedef struct gan_view {
  void* ptr;
  int32_t B, Hp, Wp, C;
  int32_t y0, x0, H, W;
  int32_t dtype;
  int32_t _pad;
} gan_view;

/* Generalised tap convolution (implicit GEMM on MFMA).  Row m=(b,ho,wo) of the GEMM reads, for tap t and
 * channel c, in[((b*in_Hp + ho*in_sy + in_y0)*in_Wp + wo*in_sx + in_x0)*Cin + tapoff[t] + c] and the result
 * for output channel n is act(sum + bias[n]) [* lrelu'(mask)] stored at
 * out[((b*out_Hp + ho*out_sy + out_y0)*out_Wp + wo*out_sx + out_x0)*out_C + n].
 * One descriptor covers nn.Conv2d forward, its input-gradient (flipped taps over a zero-haloed dY), the
 * four sub-pixel phases of nn.ConvTranspose2d and their gradients. */
typedef struct gan_conv_desc {
  int32_t dtype;                 /* operand/out dtype */
  int32_t B, Ho, Wo;             /* GEMM rows */
  int32_t Cin;                   /* padded channels per tap: power of two >= 8 */
  int32_t ntaps;                 /* padded so that ntaps*Cin is a multiple of 128 bytes of operand */
  int32_t Nw;                    /* rows of the packed weight [Nw][ntaps][Cin]; multiple of the N tile */
  int32_t Nst;                   /* channels stored (<= out_C, multiple of 4) */
  const void* in;
  int32_t in_Hp, in_Wp, in_y0, in_x0, in_sy, in_sx;
  const int32_t* tapoff;         /* device [ntaps]: (dy*in_Wp + dx)*Cin */
  const void* w;
  const float* bias;             /* device fp32 [>= Nst] or NULL */
  void* out;
  int32_t out_Hp, out_Wp, out_C, out_y0, out_x0, out_sy, out_sx;
  int32_t act;                   /* GAN_ACT_* applied after bias */
  const void* mask;              /* optional LeakyReLU-derivative mask: result *= (mask>0 ? 1 : 0.2); element
                                    ((b*mask_Hp + ho*out_sy + mask_y0)*mask_Wp + wo*out_sx + mask_x0)*out_C + n */
  int32_t mask_Hp, mask_Wp, mask_y0, mask_x0;
  float* stats;                  /* optional InstanceNorm partials fp32 [B][P][out_C][2], P = gan_conv_stats_parts(desc) > 0: per
                                    (image, pixel tile, channel) sum and sum of squares of the result (bias included, before its
                                    rounding to the output type), written with plain stores (deterministic); act must be none */
  int32_t max_tapoff;            /* largest value in tapoff[] (needed by the range-patch kernel's span check) */
  int32_t w_layout;              /* 0: w is [Nw][ntaps][Cin] (generic kernel); 1: fragment-major [Nw/16][ntaps*Cin/32][64][8]
                                    for the range-patch kernel (the descriptor must satisfy gan_conv_patch_ok); 2: w as for 0, run
                                    by the 7x7 window kernel (the descriptor must satisfy gan_conv_win7_ok) */
  int32_t win_ty0, win_tx0;      /* w_layout 2: tapoff[t] = ((win_ty0 + t/7) * in_Wp + win_tx0 + t%7) * Cin, t = 0..48 */
  int32_t tile_rows;             /* w_layout 1: output pixels per tile (256 or 288), fixed by the planner with gan_conv_patch_tile_rows so
                                    that the launch and the partial count of `stats` (gan_conv_stats_parts) agree; 0: chosen at launch */
  int32_t tile_cols;             /* w_layout 1: output channels per tile (128 or 256), fixed by the planner with gan_conv_patch_tile_cols
                                    AFTER tile_rows is set; 0: chosen at launch */
  /* dtype GAN_FP8 (range-patch kernel only, w_layout 1): `in` and `w` hold e4m3 bytes, `out` / `mask` / `bias` are as for GAN_BF16
   * (the result is bf16).  result = act(acc * w_scale[0] * (in_scale ? in_scale[b] : 1) + bias): the dequantisation scales of the
   * weight copy (gan_weight_scale_batch) and of image b of the input copy (gan_quantize_fp8), both device pointers. */
  const float* w_scale;
  const float* in_scale;
  /* Backward chain of a residual block (range-patch kernel only, w_layout 1, bf16, act none, no bias; generator_resnet_attn.py:56,64):
   * stats_mode 0: `stats` receives (sum t, sum t^2), the forward statistics of the result, and `mask` is the LeakyReLU' mask (as before);
   * stats_mode 1: `stats` receives (sum t [m > 0], sum t m) -- the two sums the InstanceNorm backward behind a ReLU needs of its incoming
   *   gradient t (this launch's result, an input gradient on the reflect-padded domain) -- with m = relu(xhat) = the activation the
   *   forward saved WITH its reflect halo, named by `mask` and the mask_* geometry and read at the output pixel's own position; the mask
   *   is then NOT applied to the result.  Summing on the padded domain equals summing the folded gradient because the halo of m is a copy
   *   of its pre-image.  Same partial layout as mode 0 ([B][P][out_C][2], P = gan_conv_stats_parts); consumer: gan_in_bwd_parts. */
  int32_t stats_mode;
  int32_t _pad2;
} gan_conv_desc;

/* Weight-gradient GEMM: part[s][n][t][c] = sum over the rows m of split s of
 * g[g_off(m) + n] * x[x_off(m) + tapoff[t] + c], with g_off/x_off as in gan_conv_desc.  nsplit slabs. */
typedef struct gan_wgrad_desc {
  int32_t dtype;
  int32_t B, Ho, Wo;
  int32_t Cx;                    /* channels of x per tap (multiple of 8) */
  int32_t ntaps;                 /* real taps */
  int32_t N;                     /* channels of g used (multiple of 8) */
  int32_t nsplit;
  const void* x;
  int32_t x_Hp, x_Wp, x_y0, x_x0, x_sy, x_sx;
  const int32_t* tapoff;         /* device [ntaps]: (dy*x_Wp + dx)*Cx */
  const void* g;
  int32_t g_Hp, g_Wp, g_C, g_y0, g_x0, g_sy, g_sx;
  float* part;                   /* device fp32 [nsplit][N][ntaps][Cx] */
  int32_t max_tapoff;            /* largest value in tapoff[] (range-patch variant's span check) */
  int32_t variant;               /* 0: generic kernel, any nsplit; 1: range-patch kernel, nsplit = B * gan_wgrad_patch_splits() (or B / -that);
                                    2: 7x7 window kernel, nsplit = gan_wgrad_win7_splits() */
} gan_wgrad_desc;

const char* gan_last_error(void);
int gan_version(void);

/* ---- convolution family: replaces nn.Conv2d / nn.ConvTranspose2d forward+backward
 *      (GAN_Variant1/models/generator_resnet_attn.py:33,48,113,125,146-149,160; discriminator_patchgan.py:27,38,45,51;
 *       Basic_GAN/src/models.py:12,16,29,37,50-51,59,81,88,96,103) */
int gan_conv_igemm(const gan_conv_desc* d, void* stream);
/* 1 if the descriptor qualifies for the range-patch kernel (bf16, Cin % 64 == 0, Nw % 128 == 0, one tile's pixel span fits
 * the LDS slab); pure host-side predicate used by the planner to choose the weight layout */
int gan_conv_patch_ok(const gan_conv_desc* d);
/* pixels per tile the range-patch kernel would choose for this descriptor (256 or 288: the one with the fewest CU-rounds x rows; the
 * tuning variable GAN_PATCH_BM is read HERE, at planning time, never at launch); 0 if the descriptor does not qualify */
int gan_conv_patch_tile_rows(const gan_conv_desc* d);
/* output channels per tile for the descriptor's tile_rows: 256 (the whole Cout of the residual 256 -> 256 layers: one slab staging and
 * one epilogue per pixel tile, 32-36 MFMAs per k-step and wave) when Nst % 256 == 0, the layer has >= 4 taps, its maps are at most 64
 * pixels wide and the 256-wide tiles still fill the chip (>= 192 of them), else 128.  GAN_PATCH_BN = 128 | 256 (read here, at planning
 * time) forces one wherever the layer is eligible; 0 if the descriptor does not qualify for the range-patch kernel */
int gan_conv_patch_tile_cols(const gan_conv_desc* d);
/* the instantiation gan_conv_igemm runs a qualifying descriptor on (0: it does not qualify): tile rows | tile columns << 12 | LDS
 * slices (7 or 9: pixels a tile's taps span, / 64) << 24 | e4m3 operands << 28 | static 3x3 schedule << 29 | other static tap
 * schedules << 30 (1: 4 taps, 2: 2 taps, 3: 16 taps -- the sub-pixel phases and the discriminator's 4x4 windows).  The value uses bit 31:
 * read it as unsigned.  Pure host-side query: tests assert with it that a case really reaches the kernel it is meant to cover. */
int gan_conv_patch_variant(const gan_conv_desc* d);
/* 1 if the descriptor qualifies for a 7x7 window kernel (bf16, stride 1, 49 row-major taps located by win_ty0/win_tx0, act none or
 * tanh, no mask / stats): Cin = 64, Nw = 16, Nst = out_C = 8 (the 64 -> 3 channel layers) or Cin = 8, Nw = Nst = out_C = 64 with the
 * tap list padded to >= 52 (the 3 -> 64 channel layers) */
int gan_conv_win7_ok(const gan_conv_desc* d);
/* pixel tiles per image for which the descriptor's launch writes InstanceNorm partials to d->stats; 0: it cannot (then use gan_in_stats) */
int gan_conv_stats_parts(const gan_conv_desc* d);
int gan_conv_wgrad(const gan_wgrad_desc* d, void* stream);
/* splits per image the range-patch weight-gradient kernel wants (0: the descriptor does not qualify: bf16, 9 taps, stride 1,
 * Cx % 64 == 0, N % 128 == 0, one 128-pixel stage's window span fits LDS); pure host-side predicate for the planner.
 * A NEGATIVE value -k means k whole images per split (many small maps, e.g. 16x16 at batch 256): nsplit = B / k. */
int gan_wgrad_patch_splits(const gan_wgrad_desc* d);
/* grad[(a*I2 + b)*KK + khw[t]] (+)= sum_s part[s][n][t][c], (a,b) = swap ? (c,n) : (n,c), for n<N_real, c<C_real, khw[t]>=0 */
/* slabs the 7x7 window weight-gradient kernels write (0: the descriptor does not qualify: bf16, 49 row-major taps, stride 1, and
 * Cx = 64, N = g_C = 8 -- the generator's 64 -> 3 channel output convolution -- or Cx = 8, N = g_C = 64 -- its 3 -> 64 first one) */
int gan_wgrad_win7_splits(const gan_wgrad_desc* d);
int gan_wgrad_reduce(const float* part, int nsplit, int N, int ntaps, int Cx, int N_real, int C_real, int swap, int I2,
                     int KK, const int32_t* khw, float* grad, int accumulate, void* stream);
/* dst[n][t][c] = src[(a*I2 + b)*KK + khw[t]] (0 where n>=N_real, c>=C_real or khw[t]<0); dst dtype GAN_*.
 * layout 0: row-major [Nw][ntaps][Cin]; layout 1: fragment-major, element (n, k=t*Cin+c) at
 * (((n/16)*(ntaps*Cin/32) + k/32)*64 + ((k%32)/8)*16 + n%16)*8 + k%8 (one MFMA operand fragment = 1 KB contiguous) */
int gan_pack_weight(const float* src, void* dst, int dtype, int Nw, int ntaps, int Cin, int N_real, int C_real, int swap,
                    int I2, int KK, const int32_t* khw, int layout, void* stream);
/* gan_pack_weight for many operand copies in one launch.  `descs` is a DEVICE array of n descriptors (fields as the arguments
 * of gan_pack_weight); the caller assigns each a contiguous block range: first_block = running sum of nblocks (256 threads per
 * block, any nblocks >= 1), total_blocks = their sum.  Validation of each descriptor is the caller's (same rules). */
typedef struct gan_pack_desc {
  const float* src; void* dst; const int32_t* khw;
  int32_t dtype, Nw, ntaps, Cin, N_real, C_real, swap, I2, KK, layout, first_block, nblocks;
  float* scale;                  /* dtype GAN_FP8: device float, dst = e4m3(src / *scale); written by gan_weight_scale_batch; else unused */
} gan_pack_desc;
int gan_pack_weight_batch(const gan_pack_desc* descs, int n, int total_blocks, void* stream);
/* For every descriptor with dtype GAN_FP8: *scale = max|src| / 448 over the whole master weight ((swap ? C_real : N_real) * I2 * KK
 * floats), 1 if the weight is all zero -- the per-tensor dequantisation scale of the e4m3 operand copy.  One launch for the batch
 * (same DEVICE descriptor array as gan_pack_weight_batch, which it precedes). */
int gan_weight_scale_batch(const gan_pack_desc* descs, int n, void* stream);
/* e4m3 operand copy of an activation / gradient buffer: dst (GAN_FP8) has exactly src's (GAN_BF16 / GAN_F32) geometry (B, Hp, Wp, C,
 * halo) and the WHOLE allocation is converted, halo included (the producer already materialised it).  amax == NULL: dst = e4m3(src)
 * (unit scale: InstanceNorm outputs are O(1)).  amax != NULL: device float[B] holding max|src| per image (gan_in_bwd_amax); then
 * scale_out[b] = amax[b] / 448 (1 if zero) and dst = e4m3(src / scale_out[b]).  Values are clamped to +-448 before conversion. */
int gan_quantize_fp8(const gan_view* src, const gan_view* dst, const float* amax, float* scale_out, void* stream);
/* bias gradient: grad[n] (+)= sum over logical pixels of g[...,n], n < N_real (column sums of dY) */
int gan_bias_grad(const gan_view* g, int N_real, float* grad, int accumulate, float* ws, void* stream);

/* ---- InstanceNorm2d (+ReLU/LeakyReLU, + residual add, + halo fill): replaces nn.InstanceNorm2d, nn.ReLU,
 *      nn.ReflectionPad2d and the residual add (generator_resnet_attn.py:25,43,56,64,71,111,114-115,126-127,150-151,158;
 *      Basic_GAN/src/models.py:10-18,30-31,38-39,52-53,91-92,99-100).  stats = fp32 [B][C][2] (mean, rstd). */
int gan_in_stats(const gan_view* x, float eps, float* stats, float* ws, void* stream);
/* (mean, rstd) from the per-tile partials a convolution epilogue wrote to gan_conv_desc.stats (parts = [B][nparts][C][2]) */
int gan_in_stats_from_parts(const float* parts, int nparts, int B, int C, int HW, float eps, float* stats, void* stream);
/* turns whole-image sums (sum, sum of squares) into (mean, rstd) in place */
int gan_in_finalize(float* stats, int BC, int HW, float eps, void* stream);
int gan_in_apply(const gan_view* x, const float* stats, int act, const gan_view* residual, const gan_view* y,
                 int halo_mode, void* stream);
/* The same pass with the statistics taken from per-chunk partial sums, parts = fp32 [B][nparts][C][2] (sum, sum of squares),
 * 1 <= nparts <= 16: each block adds the partials of its image up itself (fp64, fixed order), so no separate statistics launch sits
 * between the producing convolution and this pass; (mean, rstd) are also written to `stats` for the backward pass.  The partials come
 * from a convolution epilogue (gan_conv_desc.stats with gan_conv_stats_parts(desc) <= 16) or from gan_in_partial, which writes
 * gan_in_partial_count(x) of them per image. */
int gan_in_partial_count(const gan_view* x);
int gan_in_partial(const gan_view* x, float* parts, void* stream);
int gan_in_apply_parts(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                       const gan_view* y, int halo_mode, void* stream);
/* gan_in_apply_parts that also writes y8, an e4m3 copy of y (GAN_FP8 view of y's geometry, unit scale, halo included): the operand of the
 * next convolution on the fp8 path without a separate gan_quantize_fp8 pass */
int gan_in_apply_parts_fp8(const gan_view* x, const float* parts, int nparts, float eps, float* stats, int act, const gan_view* residual,
                           const gan_view* y, const gan_view* y8, int halo_mode, void* stream);
/* backward: g = (fold of `gy` over its reflect halo if fold) [+ g2], masked by act'(xhat) (relu / lrelu);
 * dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) written to the interior of `dx` (halo untouched).
 * ws: fp32 >= B*96*C*2 + B*C*2 floats (gan_in_stats: B*96*C*2). */
int gan_in_bwd(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
               const gan_view* dx, float* ws, void* stream);
/* gan_in_bwd that also produces the gradient of the convolution bias in front of the norm (column sums of dx) in the same
 * pass: bias_grad[n] (+)= sum_pixels dx[..,n], n < bias_n.  ws: fp32 >= B*96*C*2 + B*C*2 + (B*1024+32)*C floats. */
int gan_in_bwd_bias(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
                    const gan_view* dx, float* ws, float* bias_grad, int bias_n, int bias_accumulate, void* stream);
/* out = a + fold(b): gradient of a residual block input (skip path + reflect-padded conv path) */
/* gan_in_bwd_bias with the bias-gradient sum deferred: the per-block column sums go to `bias_part`
 * ([gan_in_bwd_bias_parts(x)][x->C] floats, caller-owned) and gan_bias_finalize_batch adds them up for many layers in one launch
 * (descs on the device; first_block = running sum of ceil(C/32); total_blocks = that sum over all descriptors). */
typedef struct gan_bias_part_desc {
  const float* part; float* grad;
  int32_t nparts, C, N_real, accumulate, first_block, _pad;
} gan_bias_part_desc;
int gan_in_bwd_bias_parts(const gan_view* x);
/* The apply half of gan_in_bwd_bias_deferred alone, for a gradient whose two per-(image, channel) sums were already produced by the launch
 * that wrote it: parts = fp32 [B][nparts][C][2], 1 <= nparts <= 96, summed here in fp64 and in part order.
 * parts_mode 1: (sum g', sum g' xhat), second sum in normalised units (gan_conv_desc.stats_mode 1: m = relu(xhat); act must be relu);
 * parts_mode 2: (sum g', sum g' x) against the raw x (what gan_in_bwd's own first pass computes: for a producer that has x at hand).
 * bias_part may be NULL. */
int gan_in_bwd_parts(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* dx, const float* parts,
                     int nparts, int parts_mode, float* bias_part, void* stream);
int gan_in_bwd_bias_deferred(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* g2,
                             const gan_view* dx, float* ws, float* bias_part, void* stream);
int gan_bias_finalize_batch(const gan_bias_part_desc* descs, int n, int total_blocks, void* stream);
/* gan_in_bwd_bias_deferred (bias_part may be NULL: no bias gradient) that also leaves max|dx| per image in amax[B] (device floats,
 * combined with atomic max on the bit patterns -- order-independent, hence deterministic): the scale of dx's e4m3 copy */
int gan_in_bwd_amax(const gan_view* x, const float* stats, int act, const gan_view* gy, int fold, const gan_view* dx, float* ws,
                    float* bias_part, float* amax, void* stream);
int gan_fold_add(const gan_view* a, const gan_view* b, int fold, const gan_view* out, void* stream);
/* gradient of a padding layer: out[b][y][x] = sum of g over the padded positions (g carries a halo of y0/x0 pixels) that the padding
 * copies from (y,x); mode GAN_HALO_REPLICATE (nn.ReplicationPad2d: an edge pixel collects its whole halo run) or GAN_HALO_REFLECT */
int gan_pad_fold(const gan_view* g, int mode, const gan_view* out, void* stream);
/* dx = g * act'(y) (tanh: 1-y^2, lrelu: y>0?1:0.2), g optionally folded; written to the interior of dx */
int gan_act_bwd(const gan_view* y, int act, const gan_view* g, int fold, const gan_view* g2, const gan_view* dx, void* stream);

/* ---- layout boundary: NCHW fp32 (the reference's tensors) <-> halo-NHWC */
int gan_nchw_to_view(const float* src, int C, const gan_view* dst, int halo_mode, void* stream);
int gan_view_to_nchw(const gan_view* src, int C, float* dst, void* stream);
int gan_view_copy(const gan_view* src, const gan_view* dst, int halo_mode, void* stream);   /* interior copy + halo fill */

/* ---- AvgPool2d(kernel 3, stride 2, padding 1, count_include_pad=False): the downsampling between the scales of
 *      MultiscaleDiscriminator (GAN_Variant1/models/discriminator_patchgan.py:100, 110-112; get_intermediate_features :125-127).
 *      y is ((H-1)/2+1) x ((W-1)/2+1); only interiors are written (zero halos stay zero).  bwd: gx (+)= pool^T gy. */
int gan_avgpool_fwd(const gan_view* x, const gan_view* y, void* stream);
int gan_avgpool_bwd(const gan_view* gy, const gan_view* gx, int accumulate, void* stream);

/* ---- Spectral normalisation of a convolution weight: torch.nn.utils.spectral_norm as applied by
 *      GAN_Variant1/models/discriminator_patchgan.py:21-23 and Basic_GAN/src/models.py:68-69 (n_power_iterations 1, eps 1e-12).
 *      W = weight_orig as an h x w row-major matrix (h = Cout, w = Cin*kh*kw: the OIHW tensor itself); u [h], v [w] are the
 *      module's weight_u / weight_v buffers.  fwd: if power_iter, v <- normalize(W^T u), u <- normalize(W v) in place; then
 *      *sigma = u . (W v) and Wsn = W / sigma.  bwd: dW = (G - <G, Wsn> u v^T) / sigma with G = dL/dWsn (u, v, sigma: the values
 *      the forward left).  ws: fp32, >= gan_spectral_norm_ws_floats(h, w). */
int64_t gan_spectral_norm_ws_floats(int h, int w);
int gan_spectral_norm_fwd(const float* W, int h, int w, float* u, float* v, int power_iter, float eps, float* sigma, float* Wsn,
                          float* ws, void* stream);
int gan_spectral_norm_bwd(const float* G, const float* Wsn, const float* u, const float* v, const float* sigma, int h, int w,
                          float* dW, float* ws, void* stream);

/* ---- DiffAugment (GAN_Variant1/training/diffaugment.py:6-60,94-106), per-sample parameters injected.
 *      prm = device fp32 [B][12]: brightness add, saturation factor, contrast factor, tx, ty,
 *      cut_lo_h, cut_hi_h, cut_lo_w, cut_hi_w (inclusive; lo>hi = no cutout), 3 spare.  C = real channels (3). */
int gan_diffaug_fwd(const gan_view* x, int C, const float* prm, const gan_view* y, float* ws, void* stream);
int gan_diffaug_bwd(const gan_view* gy, int C, const float* prm, const gan_view* gx, float* ws, void* stream);

/* ---- Device-side input pipeline: what the reference's dataset workers do per image with PIL, bit for bit --
 *      GAN_Variant1/dataio/transforms.py:10-49 (RandomCropResize = crop + BICUBIC resize, RandomHorizontalFlip,
 *      ColorJitter(0.05,0.05,0.05,0.02), ToTensor, Normalize(0.5,0.5); eval: Resize) and Basic_GAN/src/data.py:8-26
 *      (Resize(load_size, BICUBIC), RandomCrop/CenterCrop, flip, ToTensor, Normalize).  One job per image; the random draws are
 *      made by the caller (dataio.py reproduces the reference's draw order) so the kernels are deterministic. */
typedef struct gan_input_job {
  const uint8_t* src;            /* device: decoded RGB, HWC, 3 bytes per pixel */
  int32_t src_stride;            /* bytes per source row */
  int32_t crop_y, crop_x, crop_h, crop_w;   /* TF.crop box in the source (whole image: 0,0,H,W) */
  int32_t res_h, res_w;          /* size the box is resized to (Image.resize, BICUBIC) */
  int32_t win_y, win_x;          /* S x S window taken from the resized image (RandomCrop / CenterCrop; 0,0 when res == S) */
  int32_t flip;                  /* RandomHorizontalFlip */
  int32_t order[4];              /* ColorJitter: op applied in slot 0..3 (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none) */
  float factor[4];               /* brightness, contrast, saturation factors (indexed by op); [3] unused */
  int32_t hue_shift;             /* uint8(hue_factor * 255): added to the H channel with wrap-around */
  int32_t hb_off, hk_off, hksize;  /* horizontal taps in the tables block (int32 units): bounds [res_w][2], taps [res_w][hksize] */
  int32_t vb_off, vk_off, vksize;  /* vertical taps: bounds [res_h][2], taps [res_h][vksize] */
} gan_input_job;
/* Pillow's bicubic taps for resizing in_size -> out_size (Resample.c precompute_coeffs, 22-bit fixed point).  Host-only, no GPU. */
int gan_resize_ksize(int in_size, int out_size);
int gan_resize_coeffs(int in_size, int out_size, int32_t* bounds /* [out][2]: first index, count */, int32_t* kk /* [out][ksize] */, int ksize);
/* jobs_dev/tables_dev: device copies; jobs_host: the same jobs readable by the host (validation, launch shapes).
 * tmp: >= B*tmp_rows*S*4 bytes (tmp_rows >= max crop_h); img: B*S*S*4 bytes; mean_ws: B int32; out: fp32 [B][3][S][S] in [-1,1]. */
int gan_input_pipeline(const gan_input_job* jobs_dev, const gan_input_job* jobs_host, int B, const int32_t* tables_dev, int S,
                       uint8_t* tmp, int tmp_rows, uint8_t* img, int32_t* mean_ws, float* out, void* stream);

/* ---- losses.  Every loss writes its value to *loss (device fp32, overwritten) and the gradient wrt its input.
 *      hinge: adv_hinge.py:6-62 (mode 0: mean relu(1-x), 1: mean relu(1+x), 2: -mean x), scaled by `scale`;
 *      lsgan/bce: Basic_GAN/src/losses.py:5-22 (mode 3: mse vs target, 4: bce-with-logits vs target in {0,1});
 *      l1: identity_l1.py:18-20 and Basic_GAN/src/losses.py:24-30 (target given as NCHW fp32). */
int gan_patch_loss(const gan_view* logits, int mode, float target, float scale, float* loss, const gan_view* grad, void* stream);
int gan_l1_loss(const gan_view* x, int C, const float* target_nchw, float scale, const float* dev_grad_scale, float* loss,
                const gan_view* grad, float* ws, void* stream);   /* grad additionally * (*dev_grad_scale) if non-NULL */
/* r1 = (1/B) sum_b sum_chw g^2 (train_cutpp.py:201) and u = scale * 2 g / B into `u` (interior) */
int gan_r1_reduce(const gan_view* g, int C, float scale, float* loss, const gan_view* u, float* ws, void* stream);

/* ---- PatchNCE (GAN_Variant1/losses/patchnce_cut.py:42-110) for one feature layer.
 *      ids: device int32 [P] positions in [0,H*W).  ws: fp32 workspace >= gan_patchnce_ws_floats(B,P,C).
 *      fwd: *loss += weight * mean_b CE (non-finite per-image losses count as 0).  bwd: grad rows of tgt
 *      (scaled by weight) are ADDED into `gtgt` (duplicates in ids accumulate). */
int64_t gan_patchnce_ws_floats(int B, int P, int C);
int gan_patchnce_fwd(const gan_view* src, const gan_view* tgt, const int32_t* ids, int P, int C, float temperature,
                     float weight, float* loss, float* ws, void* stream);
int gan_patchnce_bwd(const gan_view* tgt, const int32_t* ids, int P, int C, float temperature, float weight,
                     const gan_view* gtgt, float* ws, void* stream);

/* ---- fused clip_grad_norm_ + Adam + EMA (amp_utils.py:29-41, sched_optim.py:5-27, io_ckpt.py:23-29) over a
 *      tensor list.  The table is a device array of gan_adam_tensor; tensors with g == NULL are skipped. */
typedef struct gan_adam_tensor {
  float* p; const float* g; float* m; float* v; float* ema;   /* ema may be NULL */
  int64_t numel;
  int32_t* step;                                               /* device per-tensor step counter (incremented) */
  int64_t _pad;
} gan_adam_tensor;
/* norm_out: device fp32 [3] = (total L2 norm before clipping, clip coefficient, found_inf).  max_norm <= 0: no clipping.
 * grad_scale multiplies every gradient first (1/world_size after a sum all-reduce).
 * lr_dev (optional device float): the learning rate is read from it instead of `lr` -- a scheduler (Basic_GAN/src/train.py:27-31,54-58,125:
 *   LambdaLR with lambda_rule) rewrites one device float and the prebuilt launch stays valid.
 * inv_scale_dev (optional device float) and skip_nonfinite: torch.amp.GradScaler's unscale_ / step (amp_utils.py:29-41): gradients are
 *   multiplied by *inv_scale_dev, and with skip_nonfinite a non-finite total norm skips the update AND the step counters; found_inf is
 *   written to norm_out[2] either way (gan_scaler_update consumes it). */
int gan_adam_step(const gan_adam_tensor* table, int ntensors, const int32_t* chunk_tensor, const int64_t* chunk_off,
                  int nchunks, float lr, float beta1, float beta2, float eps, float max_norm, float grad_scale,
                  float ema_decay, const float* lr_dev, const float* inv_scale_dev, int skip_nonfinite, float* norm_out, float* ws, void* stream);
/* torch.amp.GradScaler.update on the device (amp_utils.py:22,41): scale *= backoff_factor after an overflow (found_inf != 0, e.g.
 * norm_out + 2 of gan_adam_step), *= growth_factor after growth_interval clean steps; inv_scale = 1 / scale; no host synchronisation. */
int gan_scaler_update(float* scale, float* inv_scale, int32_t* growth_tracker, const float* found_inf, float growth_factor,
                      float backoff_factor, int growth_interval, void* stream);

/* small helpers */
int gan_fill_f32(float* p, int64_t n, float v, void* stream);
int gan_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream);   /* y += a*x */

#ifdef __cplusplus
}
#endif
#endif
