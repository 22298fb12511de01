/*
 * tfc_gan.h -- C ABI of libtfcgan_hip.so: the MI355X (gfx950) kernels of the TFC-GAN PATCH-16 training hot path.
 *
 * The reference (nudro/TFC-GAN) has NO native or FFI layer: its hot path is the PyTorch op sequence inside
 * TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py ("P16" below), lines 545-638.  Each entry point here replaces one
 * stock-op sequence of that script; the host-side mirror of the reference's Python surface (GeneratorUNet,
 * Discriminator1, make_16_patches, ...) lives in tfc-gan_amd/ and binds these symbols with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named *_host; `stream` is a hipStream_t passed as void* (0 = null stream);
 *   - activations are NHWC with an explicit pixel pitch (elements), dtype `dt` (TFC_DT_BF16 storage / fp32 accumulate,
 *     or TFC_DT_F32 = exact-fp32 parity mode using v_mfma_f32_32x32x2_f32); channel counts that are gathered or
 *     produced by MFMA tiles are padded to a multiple of 8 by the caller (3 -> 8, 6 -> 8, 1 -> 8);
 *   - weights / gradients / statistics / losses are fp32 in the torch layouts of the reference's state_dict;
 *   - return value 0 = success, otherwise a negative code; tfc_last_error() gives the thread-local message.
 *     No entry point allocates, frees or synchronises (all are hipGraph-capturable).
 *   - DETERMINISM: no entry point of the training path adds floats with atomics. Sums that cross workgroups (InstanceNorm statistics, their
 *     backward reductions, bias gradients, spectral-norm products, split-K weight gradients) leave every workgroup as a partial in a
 *     FIXED slot of a scratch buffer and are added in a fixed order behind the kernel: the same inputs give the same bits on every run.
 *     Entry points that reduce therefore take `part_ws`: tfc_part_ws_floats() floats (32 MiB) of device scratch that only launches
 *     of ONE stream use at a time (the Python side keeps one per (device, stream)); contents are undefined between calls.
 */
#ifndef TFC_GAN_H
#define TFC_GAN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TFC_DT_BF16 0
#define TFC_DT_F32 1

/* convolution ops of the path */
#define TFC_OP_CONV 0     /* nn.Conv2d(k4,s1,p1)                              P16:105 (UNetDown), :189 (Discriminator1 blocks) */
#define TFC_OP_PADCONV 1  /* nn.ZeroPad2d((1,0,1,0)) + nn.Conv2d(k4,p1)       P16:201-202 (PatchGAN head)                    */
#define TFC_OP_CONVT 2    /* nn.ConvTranspose2d(k4,s2,p1)                     P16:122 (UNetUp)                               */
#define TFC_OP_UPCONV 3   /* nn.Upsample(x2)+nn.ZeroPad2d((1,0,1,0))+nn.Conv2d(k4,p1) [+Tanh]   P16:153-158 (generator head)  */
#define TFC_OP_CONV3 4    /* nn.Conv2d(k3,s1,p1): VGG16 features of criterion_lpips (P16:70-73, :598). Weights are passed zero-padded to
                             [Cout][Cin][4][4] (filter in rows / columns 0..2); forward and input gradient only (frozen network)          */

/* epilogue flags of tfc_conv_fwd / tfc_conv_dgrad */
#define TFC_EP_BIAS 1       /* + bias[Cout]                                                                                   */
#define TFC_EP_STATS 2      /* stats[N][Cout][2] += (sum, sum of squares) of the result: nn.InstanceNorm2d statistics, P16:107 (needs part_ws) */
#define TFC_EP_ACCUM 4      /* result += existing output (skip-connection gradient accumulation, torch.cat backward P16:133)  */
#define TFC_EP_TANH_NCHW 8  /* nn.Tanh (P16:157) and store fp32 NCHW to `out_nchw`                                            */
#define TFC_EP_LEAKY 16     /* nn.LeakyReLU(0.2) of Discriminator1 (P16:190) applied to the result before it is stored          */
#define TFC_EP_RELU 32      /* nn.ReLU applied to the result before it is stored (VGG16 feature stack of LPIPS)                    */

const char* tfc_last_error(void);
int tfc_abi_version(void);
size_t tfc_part_ws_floats(void);   /* size (floats) of the per-stream partial-sum scratch `part_ws` */

/* ---- weights ------------------------------------------------------------------------------------------------- */
/* pass: 0 = forward operand stream, 1 = dgrad operand stream.  w is the torch-layout fp32 weight
 * (Conv2d [Cout][Cin][4][4]; ConvTranspose2d [Cin][Cout][4][4]); *scale (nullable device scalar) multiplies every
 * element (1/sigma of spectral_norm, P16:188).  Replaces the implicit weight cast of torch.cuda.amp.autocast. */
size_t tfc_conv_packed_bytes(int dt, int op, int pass, int Cin, int Cout);
int tfc_conv_pack(void* stream, int dt, int op, int pass, const float* w, const float* scale, void* packed, int Cin, int Cout);

/* Planned packing: all operand streams of a network in ONE launch. Build the job table once on the host (geometry, weight and
 * stream pointers are fixed for the lifetime of an engine), copy it to the device, then call tfc_conv_pack_planned per update. */
size_t tfc_pack_plan_bytes(int nlayers);
int tfc_pack_plan_build(int dt, int nlayers, const int* ops_host, const int* passes_host, const float* const* w_host,
                        void* const* packed_host, const int* Cin_host, const int* Cout_host, void* plan_host, int* nblocks_host);
int tfc_conv_pack_planned(void* stream, int dt, const void* plan_dev, int njobs, int nblocks);

/* ---- forward: y = oscale * op(x) + bias ; x: [N][H][W][x_pitch], y: [N][OH][OW][y_pitch] (OH = H-1 | H | 2H | 2H) ---
 * oscale: nullable DEVICE scalar multiplying the accumulator before the bias -- 1/sigma of spectral_norm (P16:188), so the
 * discriminator's operand streams are packed once per weight update and not once per forward. */
int tfc_conv_fwd(void* stream, int dt, int op, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout,
                 const void* packed, void* y, int y_pitch, const float* bias, float* stats, float* out_nchw, const float* oscale, int flags,
                 float* part_ws /* nullable unless TFC_EP_STATS */);
/* ---- the FIRST convolution of either network (UNetDown(channels, 64, normalize=False) P16:140; discriminator_block(2 * channels, 64) P16:194): bf16,
 * Cin <= 8 (x: NHWC8), Cout == 64, weights-stationary kernel. Same result as tfc_conv_fwd(TFC_OP_CONV) with flags in {TFC_EP_BIAS, TFC_EP_LEAKY}; in
 * addition sign_mask (nullable, 8-byte aligned): uint8 [N][H-1][W-1][8], bit c of a pixel's 64-bit word = (stored y[c] > 0) -- all that the backward of
 * the block needs of y when only its weight / bias gradient is wanted (tfc_first_block_bwd_wgrad reads 8 bytes per pixel instead of 128). */
int tfc_conv_first_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const void* packed, void* y, int y_pitch,
                       const float* bias, const float* oscale, int flags, uint8_t* sign_mask);
/* the whole first block forward in ONE kernel: BlurPool(stride 2)(LeakyReLU(slope)(oscale * conv(x) + bias)) -> out [N][Ho][Wo] (out_pitch), Ho = (H-2)/2+1,
 * for discriminator_block(2 * channels, 64) (P16:187-196; act_after_rounding = 0: the activation runs before the bf16 rounding, as in the conv epilogue of
 * the unfused chain) and UNetDown(channels, 64, normalize=False) (P16:140; act_after_rounding = 1: the raw conv output is rounded, the activation runs in
 * fp32 inside the pooling). The 266 MB conv output is never written; sign_mask (nullable) receives its sign words as tfc_conv_first_fwd would leave them.
 * Same numbers as tfc_conv_first_fwd + tfc_act_fwd(pool = 2) up to the fp32 summation order of the 16 blur taps (<= 1 bf16 ulp on rare elements). */
int tfc_first_block_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const void* packed, const float* bias,
                        const float* oscale, float slope, int act_after_rounding, void* out, int out_pitch, uint8_t* sign_mask);
/* ---- PatchGAN head forward, P16:201-202: ZeroPad2d((1,0,1,0)) + Conv2d(C,1,k4,p1,no bias) as a wave-per-pixel dot product
 * (w: torch-layout fp32 [1][C][4][4], y: [N][H][W][y_pitch] channel 0). Same result as tfc_conv_fwd(TFC_OP_PADCONV, Cout=1). */
int tfc_patchgan_head_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int C, const float* w,
                          void* y, int y_pitch);
/* generator head, P16:150-157: nn.Upsample(2) -> nn.ZeroPad2d((1,0,1,0)) -> nn.Conv2d(128, Cout <= 4, 4, padding=1) -> nn.Tanh, forward,
 * bf16 only: x [N][H][W][x_pitch] (128 channels), w torch-layout fp32 [Cout][128][4][4], out fp32 NCHW [N][Cout][2H][2W].
 * Same result as tfc_conv_fwd(TFC_OP_UPCONV, TFC_EP_BIAS | TFC_EP_TANH_NCHW); the four sub-pixel phases share one 16-wide MFMA tile. */
int tfc_upconv_head_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const float* w,
                        const float* bias, float* out_nchw);
/* its input gradient: dy NHWC8 [N][2H][2W][8] (Cout <= 8 real channels) -> dx NHWC [N][H][W] channels [0,128) at dx_pitch, overwritten.
 * w: the fp32 torch-layout filter [Cout][128][4][4] (the collapsed taps are summed in fp32 and rounded to bf16 once, as in the forward). */
int tfc_upconv_head_dgrad(void* stream, int dt, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout, const float* w, void* dx, int dx_pitch);

/* input gradient of the FIRST discriminator convolution (nn.Conv2d(Cin, 64, 4, stride 1, padding 1), P16:188) w.r.t. its first `nch`
 * (<= 4) input channels -- the generated image inside cat(img_A, img_B), P16:205 -- written as fp32 NCHW [N][nch][H][W]; bf16 only.
 * dy: [N][H-1][W-1][dy_pitch] (64 channels), w: torch-layout fp32 [64][Cin][4][4], oscale: nullable device scalar (1/sigma).
 * Same numbers as tfc_conv_dgrad(TFC_OP_CONV) + tfc_unpack_nchw restricted to those channels. */
int tfc_conv_dgrad_image(void* stream, int dt, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout, const float* w,
                         const float* oscale, int nch, float* dx_nchw);

/* ---- input gradient: dx = oscale * op^T(dy) (flags: TFC_EP_ACCUM) ------------------------------------------------ */
int tfc_conv_dgrad(void* stream, int dt, int op, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout,
                   const void* packed, void* dx, int dx_pitch, const float* oscale, int flags);
/* ---- weight gradient: dw (torch layout, fp32) = or += x (*) dy ; ws: tfc_conv_wgrad_ws_bytes() of scratch = [64 MiB of split-K
 *      slabs | fp32 accumulator]. Zero it once after allocation: the accumulator part must be ALL ZERO on entry and is left all
 *      zero on return (the slab part is scratch). One buffer sized for the largest layer may be shared by every layer. ------- */
size_t tfc_conv_wgrad_ws_bytes(int op, int Cin, int Cout);
int tfc_conv_wgrad(void* stream, int dt, int op, const void* x, int x_pitch, const void* dy, int dy_pitch, int N, int H, int W,
                   int Cin, int Cout, void* ws, float* dw, int accumulate);

/* ---- fused normalisation / activation / anti-aliased pooling ------------------------------------------------------
 * forward  : y = Dropout( Blur_{pool}( Act_{slope}( norm ? InstanceNorm(x; stats) : x ) ) )      P16:106-111, :123-128, :191-192
 *   pool: 0 none, 1 antialiased_cnns.BlurPool(stride=1), 2 BlurPool(stride=2) (reflect pad 1/2, [1,3,3,1]^2/64)
 *   slope: 0.2 LeakyReLU, 0 ReLU, 1 identity; stats: [N][C][2] (sum, sumsq) of x over H*W; eps 1e-5, biased variance
 *   stats_out (nullable): [N][C][2] += (sum, sumsq) of y  (InstanceNorm that FOLLOWS the blur in UNetUp)
 *   dropout: drop_p in [0,1): counter-based mask of (seed, element index), identical in forward and backward
 * backward : mode 0: dx = g'            (norm == 0); rstats (nullable) = float[N][C] += per-image column sums of dx = the bias gradient of the
 *                     convolution that produced x (Discriminator1 blocks, P16:189)
 *            mode 1: rstats[N][C][2] += (sum g', sum g' * xhat)          (InstanceNorm backward, reduction phase)
 *            mode 2: dx = rstd * (g' - mean g' - xhat * mean(g' xhat))   (apply phase)
 *            g' = Blur^T(dropmask * dy) * Act'(xhat) ; x == NULL => Act' = 1
 *   part_ws: needed whenever stats_out (forward) or a reduction into rstats (backward mode 1, mode 0 with rstats) is requested
 */
int tfc_act_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int C, const float* stats, int norm,
                float slope, int pool, float drop_p, uint32_t seed, void* y, int y_pitch, float* stats_out, float* part_ws);
int tfc_act_bwd(void* stream, int dt, int mode, const void* dy, int dy_pitch, const void* x, int x_pitch, int N, int H, int W, int C,
                const float* stats, int norm, float slope, int pool, float drop_p, uint32_t seed, float* rstats, void* dx, int dx_pitch, float* part_ws);
/* tfc_act_bwd(mode 0, pool 2, norm 0) for a 64-channel bf16 tensor x that was never stored: sign_mask = its sign words (uint8 [N][H][W][8], bit c of the
 * 64-bit word of a pixel = (x[c] > 0), as tfc_first_block_fwd / tfc_conv_first_fwd leave them). Same dx bits as tfc_act_bwd on the stored tensor. */
int tfc_act_bwd_signs(void* stream, int dt, const void* dy, int dy_pitch, const unsigned char* sign_mask, int N, int H, int W, int C, float slope,
                      float* rstats, void* dx, int dx_pitch, float* part_ws);
int tfc_dropout_mask(void* stream, uint8_t* keep, long long n, float drop_p, uint32_t seed);   /* test hook: the mask itself */

/* ---- layout plumbing at the NCHW fp32 module boundary -------------------------------------------------------------- */
int tfc_pack_nhwc8(void* stream, int dt, const float* a, int Ca, const float* b, int Cb, void* out, int N, int H, int W);   /* torch.cat((a,b),1), P16:207 */
int tfc_unpack_nchw(void* stream, int dt, const void* in, int pitch, int c0, int C, float* out, int N, int H, int W, float alpha, float beta);
/* nn.Tanh backward of the generator head (P16:157) + NHWC8 packing: dyraw = g * (1 - y^2); dbias (nullable, needs part_ws): float[C] += column sums */
int tfc_tanh_bwd_pack(void* stream, int dt, const float* g, const float* y, void* dyraw, float* dbias, int N, int C, int H, int W, float* part_ws);
/* bias gradient: out[C] += column sums of x [rows][pitch]; part_ws (nullable): lets many rows be split over workgroups */
int tfc_colsum(void* stream, int dt, const void* x, long long rows, int pitch, int C, float* out, float* part_ws);
int tfc_cast(void* stream, int dt, int to_f32, const void* x, void* y, long long n);
int tfc_axpby(void* stream, float* out, const float* x, const float* y, long long n, float a, float b);

/* ---- spectral norm: torch.nn.utils.parametrizations.spectral_norm, P16:188 ---------------------------------------- */
/* W: [R][K] fp32; u[R], v[K] updated in place when power_iter != 0 (u <- norm(W v), v <- norm(W^T u)); sigma2 = {sigma, 1/sigma} with
 * sigma = u . (W v), evaluated as (W^T u) . v after a power iteration (the same number up to fp32 round-off, one pass over W less);
 * ws: tfc_spectral_norm_batched_ws_floats(1, &R, &K) floats */
int tfc_spectral_norm_step(void* stream, const float* W, float* u, float* v, float* sigma2, float* ws, int R, int K, int power_iter);
/* the same for up to 4 layers in 3 launches (host arrays of device pointers / sizes); u_snap / v_snap (nullable arrays of
 * nullable pointers) receive copies of the updated u, v for the backward of THIS forward call; ws: ..._ws_floats() floats
 * (W^T u is formed from per-row-block partials that are added in a fixed order: no atomics) */
size_t tfc_spectral_norm_batched_ws_floats(int nlayers, const int* R_host, const int* K_host);
int tfc_spectral_norm_step_batched(void* stream, int nlayers, const float* const* W_host, float* const* u_host, float* const* v_host,
                                   float* const* sigma2_host, float* const* u_snap_host, float* const* v_snap_host,
                                   const int* R_host, const int* K_host, float* ws, int power_iter);
/* gW_orig (=/+=) (G - <G, W/sigma> u v^T) / sigma ; ws: 256 floats, 8-byte aligned (per-workgroup partials of <G, W>, added in a fixed order) */
int tfc_spectral_norm_bwd(void* stream, const float* G, const float* W, const float* u, const float* v, const float* sigma2,
                          float* ws, float* gout, int R, int K, int accumulate);

/* ---- loss heads --------------------------------------------------------------------------------------------------- */
/* 16-patch triplet ("contrastive") head: make_16_patches P16:227-253 + nn.TripletMarginLoss(margin=1,p=2) P16:75 applied
 * 16x with negatives real-patch[neg_idx_host[k]] P16:562-583.  fake/real: fp32 NCHW [N][C][256][256].
 * loss[0] = (1/16) sum_k mean(...);  dfake (nullable) = gscale * d loss / d fake. */
int tfc_patch16_triplet(void* stream, const float* fake, const float* real, const int* neg_idx_host, int N, int C,
                        float* loss, float* dfake, float gscale);
/* spectra: ToPILImage -> convert("L") -> np.fft.rfft2 -> fftshift -> abs / arctan2, P16:271-319.
 * img: fp32 [N][C][rows][rs] window grid: S in {64,256}; windows per image = wins_x*wins_y tiles of S x S starting at the
 * image origin; amp/pha: [N*wins][S][S/2+1] fp32; shift != 0 applies np.fft.fftshift to both axes. */
int tfc_fft_spectrum(void* stream, const float* img, long long batch_stride, long long chan_stride, int row_stride, int C, int S,
                     int wins_x, int wins_y, int N, float* amp, float* pha, int shift, void* ws);
/* ws: tfc_fft_spectrum_ws_bytes(S, N * wins_x * wins_y) bytes of scratch (row-transformed half spectra) -> radix-4 FFT in LDS (rows, then
 * columns); ws == NULL -> direct DFT (S^2 work per output row: fine for one 64 x 64 window, 1 ms per call for 32 whole 256 x 256 images). */
size_t tfc_fft_spectrum_ws_bytes(int S, int nwin);
/* evaluation metric of TFC-GAN-FFT/Devcom_MagMSE.py:91-118 (mse_spec): per window MSE(log|fft2(a)|, log|fft2(b)|) over the FULL S x S
 * spectrum, computed from the half spectra amp_a / amp_b [nwin][S][S/2+1] of tfc_fft_spectrum; out[nwin] */
int tfc_logmag_mse(void* stream, const float* amp_a, const float* amp_b, int S, int nwin, float* out);
/* the companion metric of TFC-GAN-FFT/eval/Eurecom/Eurecom_MagOther.py:90-118 (other_spec): mean_absolute_error of the same two log-magnitude spectra */
int tfc_logmag_mae(void* stream, const float* amp_a, const float* amp_b, int S, int nwin, float* out);
/* temperature head, P16:255-268 (vectorize_temps) over TFC-GAN-FFT/datasets_temp.py:14-35 (TempVector_PyTorch): the red channel of
 * ToPILImage(img[n]) = (uint8) trunc(x*255) (wraps mod 256) looked up in lut256 (float32(np.linspace(24,38,256)), device).
 * img: fp32 NCHW, channel 0 is read ([n*batch_stride + y*row_stride + x]); out: [N][H][W] fp32 temperatures in Celsius. */
int tfc_vectorize_temps(void* stream, const float* img, long long batch_stride, int row_stride, int N, int H, int W,
                        const float* lut256, float* out);
/* nn.TripletMarginLoss(margin, p=2, eps=1e-6) on row-major [rows][W] fp32 operands (distance over the last dim, mean over rows):
 * criterion_temp of P16:80, :595.  loss[0] = mean_rows max(margin + ||a-p+eps|| - ||a-n+eps||, 0).  Forward only (the
 * reference's temperature term carries no gradient: tensor -> PIL -> numpy, P16:263-265). */
int tfc_row_triplet(void* stream, const float* anchor, const float* positive, const float* negative, long long rows, int W,
                    float margin, float* loss);
/* out[0] (=/+=) scale * sum |a-b| : nn.L1Loss pieces of calculate_ffts, P16:323-375 */
int tfc_l1_sum(void* stream, const float* a, const float* b, long long n, float scale, float* out, int zero_first);
/* relativistic BCEWithLogits, P16:554 (mode 0) and P16:628-630 (mode 1); a,b: n logits in dt, `stride` elements apart
 * (the PatchGAN head stores its single channel at pixel pitch 8); loss fp32 scalar; da/db (nullable) = gscale * dloss. With stride == 8 the
 * WHOLE 8-channel pixel of da / db is written (gradient, seven zeros): the head's input-gradient pass reads all eight, no pre-zeroing needed */
int tfc_bce_relativistic(void* stream, int dt, const void* a, const void* b, int n, int stride, float t1, float t2, int mode,
                         float* loss, void* da, void* db, float gscale);
/* torch.optim.Adam step (P16:461-462) on flat fp32 buffers; step >= 1; gscale multiplies the gradient (1/world size) */
int tfc_adam_step(void* stream, float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                  int step, float gscale);

/* ---- STN21 configuration: TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py ("STN") ---------------- */
/* Net.forward, STN:228-229: F.affine_grid(theta, size, align_corners=True) + F.grid_sample(src, grid, mode='bicubic', padding_mode='border',
 * align_corners=True), fused (the grid is never stored). src / out: fp32 NCHW [N][C][H][W]; theta: [N][2][3] (identity already added, STN:208-211). */
int tfc_affine_warp_fwd(void* stream, const float* src, const float* theta, float* out, int N, int C, int H, int W);
/* backward: dtheta[N][6] = d loss / d theta (the trainable path: theta comes from the localiser; per-workgroup partials in part_ws, added in a
 * fixed order); dsrc (nullable) = d loss / d src (a scatter: float atomics, the one order-dependent output of the library -- the STN21 step
 * never asks for it). Both are overwritten (zeroed inside). gout: d loss / d out. */
int tfc_affine_warp_bwd(void* stream, const float* src, const float* theta, const float* gout, float* dtheta, float* dsrc, int N, int C, int H, int W,
                        float* part_ws);
/* morph_triplet, STN:444-449: kornia.morphology.gradient(x, [[0,1,0],[1,1,1],[0,1,0]]) = dilation - erosion with geodesic borders (neighbours
 * outside the image never win). x / out: fp32, `planes` images of H x W; arg (nullable): per pixel arg-max | arg-min << 4 for the backward. */
int tfc_morph_grad_fwd(void* stream, const float* x, float* out, uint8_t* arg, long long planes, int H, int W);
int tfc_morph_grad_bwd(void* stream, const float* gout, const uint8_t* arg, float* dx, long long planes, int H, int W);
/* criterion_morph = nn.TripletMarginLoss(margin=1.0, p=2) (STN:99, :457) over the last dim with its gradient w.r.t. the anchor:
 * loss[0] = mean_rows max(margin + ||a-p+eps|| - ||a-n+eps||, 0); danchor (nullable) = gscale * d loss / d anchor. */
int tfc_row_triplet_grad(void* stream, const float* anchor, const float* positive, const float* negative, long long rows, int W, float margin,
                         float gscale, float* loss, float* danchor);

/* ---- first block, backward, fused: UNetDown(channels, 64, normalize=False) (P16:133) and discriminator_block(2 * channels, 64) (P16:183-196) are
 * conv -> LeakyReLU -> BlurPool(stride 2) with no normalisation. When the gradient of the convolution OUTPUT is needed by nothing but the weight
 * (and bias) gradient, it is never written: = tfc_act_bwd(mode 0, pool 2) + tfc_conv_wgrad(TFC_OP_CONV) in one kernel, same bits.
 * x: NHWC8 input image [N][H][W][8]; y: the stored conv output [N][H-1][W-1] (its sign is all that is read: pre- or post-activation);
 * dy_pooled: gradient of the pooled output [N][Ho][Wo], Ho = (H-2)/2+1; dw: torch-layout gradient [Cout][Cin][4][4] (=/+=);
 * bias_sums (nullable, needs part_ws): float[N][Cout] += per-image sums of the conv-output gradient; ws: tfc_conv_wgrad_ws_bytes() of scratch (zeroed once).
 * sign_mask (nullable): the word tfc_conv_first_fwd left; when given, y is not read (and may be NULL). The transposed blur runs as a GEMM against the
 * tile's tap matrix on the matrix core: d_raw may differ from the unfused chain's by 1 bf16 ulp on rare elements (different fp32 summation order). ---- */
int tfc_first_block_bwd_supported(int dt, int Cin, int Cout);
int tfc_first_block_bwd_wgrad(void* stream, int dt, const void* x, int x_pitch, const void* y, int y_pitch, const void* dy_pooled, int dyp_pitch, int N, int H,
                              int W, int Cin, int Cout, float slope, void* ws, float* dw, int accumulate, float* bias_sums, float* part_ws,
                              const uint8_t* sign_mask);

/* ---- input pipeline: ImageDataset.__getitem__, TFC-GAN-FFT/datasets_temp.py:38-123 --------------------------
 * A decoded file is one RGB uint8 image [H][W][3] with the visible image A in columns [0, xsplit) and the thermal image B in [xsplit, W),
 * xsplit = round-half-even(W / 2) as Image.crop((0, 0, w / 2, h)) does (:54-55). Each half is resized to out x out with PIL's BICUBIC
 * resampler (:63-66; restated bit-exactly: antialiased separable convolution, 22-bit fixed-point taps, uint8 after each pass), then
 * ToTensor + Normalize(0.5, 0.5) (P16:479-482) -> fp32 NCHW, and T_B = linspace(24, 38, 256)[B red channel] (:41-42, :69-70).
 * The plan (tap tables of one file geometry) is built on the host and uploaded by the caller; all files of a batch share (H, W). */
size_t tfc_resize_plan_bytes(int H, int W, int out);
int tfc_resize_plan_build(int H, int W, int out, void* plan_host);
size_t tfc_pair_resize_ws_bytes(int N, int H, int out);
/* src: device uint8, image n at src + n * img_stride, rows row_stride bytes apart. plan_host / plan_dev: the same plan bytes on the host
 * (header) and on the device (tables). A, B: fp32 [N][3][out][out]; TB (nullable): fp32 [N][out][out]; A8 / B8 (nullable): the resized
 * uint8 images [N][out][out][3] (what PIL's resize returns, for callers that keep 8-bit copies). */
int tfc_pair_resize_normalize(void* stream, const uint8_t* src, long long img_stride, int row_stride, int N, const void* plan_host,
                              const void* plan_dev, void* ws, const float* lut256, float* A, float* B, float* TB, uint8_t* A8, uint8_t* B8);

/* ---- LPIPS term of loss_G: criterion_lpips = lpips.LPIPS(net_type='vgg', version='0.1'), P16:70-73, used at
 * P16:598 inside loss_G. lpips_pytorch is a pip dependency that is absent from the reference tree; its published algorithm is restated
 * (PARITY UNPINNED): z-score the inputs, VGG16 features at relu1_2 / 2_2 / 3_3 / 4_3 / 5_3 (3x3 convolutions = TFC_OP_CONV3 with
 * TFC_EP_RELU), channel-normalise, squared difference, 1x1 "lin" head, spatial mean, sum over layers (and over the batch: the package
 * returns torch.sum(torch.cat(res, 0), 0, True)). Activations NHWC in dt with pitch == C. ---- */
/* (x - shift[c]) / scale[c], fp32 NCHW [N][C][H][W] -> NHWC channels [0,8) of a `pitch`-channel image (C <= 8; the rest zero / untouched) */
int tfc_lpips_input_fwd(void* stream, int dt, const float* x, const float* shift, const float* scale, void* out, int N, int C, int H, int W, int pitch);
/* dx fp32 NCHW (=/+=) alpha * g[..c] / scale[c] */
int tfc_lpips_input_bwd(void* stream, int dt, const void* g, const float* scale, float* dx, int N, int C, int H, int W, int pitch, float alpha,
                        int accumulate);
/* nn.MaxPool2d(2, 2) of torchvision's vgg16.features; backward routes to the first maximum of each window */
int tfc_maxpool2_fwd(void* stream, int dt, const void* x, void* y, int N, int H, int W, int C);
int tfc_maxpool2_bwd(void* stream, int dt, const void* x, const void* dy, void* dx, int N, int H, int W, int C);
/* dz = (y > 0) ? dy (+ extra, nullable) : 0 over n elements (n % 8 == 0); dz may alias dy */
int tfc_relu_bwd(void* stream, int dt, const void* dy, const void* y, const void* extra, void* dz, long long n);
/* one LPIPS layer: out[n] += mean_pixels sum_c w[c] (fx/(|fx|+1e-10) - fy/(|fy|+1e-10))_c^2 ; dfx (nullable) = gscale * d out[n] / d fx */
int tfc_lpips_head(void* stream, int dt, const void* fx, const void* fy, const float* w, float* out, void* dfx, int N, int H, int W, int C,
                   float gscale);

/* ---- measurement -------------------------------------------------------------------------------------------------- */
/* When enabled every gather-GEMM / wgrad launch is bracketed by hipEvents on its own stream; tfc_prof_collect()
 * (call after synchronising) sums them per kernel class: 0 = tfc_igemm_kernel family, 1 = tfc_wgrad family (+ slab reduce),
 * 2 = the wgrad finish pass, 3 = the fused first-block backward (transposed blur + weight gradient in one launch). */
int tfc_prof_enable(int on);
int tfc_prof_collect(int kclass, double* total_ms, double* algorithmic_flop, long long* launches);
/* per-call detail of the records gathered on this thread since the last collect (kclass 2 = the wgrad finish pass): arrays of max_records entries,
 * meta7 = {op, pass (0 fwd, 1 dgrad, 2 wgrad, 3 wgrad finish), N, H, W, Cin, Cout} per record; returns the number of records, < 0 on error.
 * The profiling state is per THREAD: enable, launch and collect from the same thread. */
int tfc_prof_records(int max_records, int* kclass, double* ms, double* algorithmic_flop, int* meta7);

/* ---- test hooks (host side, no GPU): the table-driven gather model evaluated on the CPU with the SAME descriptors and
 * the SAME packed operand stream the kernels consume.  Never called by the product path. */
int tfc_host_emulate_conv(int op, int pass, int elem_size, const float* x_host, const float* w_host, float* y_host,
                          int N, int H, int W, int Cin, int Cout);   /* pass 2 = wgrad: w_host is dy, y_host is dw */
/* force the gather-GEMM workgroup tile (0: 128 px x 128 ch, 1: x64, 2: x32; -1: heuristic) so the tests reach every variant */
int tfc_debug_set_igemm_config(int cfg);
/* device probe of the MFMA / transposing-read lane maps the kernels rely on (writes 3*64*16 floats) */
int tfc_probe_mfma(void* stream, float* out);

#ifdef __cplusplus
}
#endif
#endif
