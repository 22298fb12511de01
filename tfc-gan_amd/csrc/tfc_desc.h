// Shared host/device descriptors of the TFC-GAN gather-GEMM engine (gfx950).
//
// Every convolution-shaped op of the PATCH-16 hot path (reference:
// TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py:102-211 -- Conv2d k4 s1 p1, ConvTranspose2d k4 s2 p1,
// Upsample+ZeroPad+Conv2d, ZeroPad+Conv2d, and all of their dgrad / wgrad passes) is expressed as one
// table-driven "gather GEMM":
//
//     out[img][a*OS+OOY][b*OS+OOX][n] = sum over planes pl, taps t in pl, channels c of
//         in[img][(a + dy0[pl] + dy[t])*SS + py[pl]][(b + dx0[pl] + dx[t])*SS + px[pl]][c]
//           * W[n][slot(pl,t)][c]                       (zero outside the source image)
//
// (a,b) runs over a GH x GW "gather grid"; a workgroup owns an 8 x 16 tile of that grid, stages the
// (8+span) x (16+span) halo of the source once per 64-byte channel chunk in LDS and replays it for
// every tap, so the im2col matrix is never materialised and each source byte crosses HBM/L2 once per
// chunk instead of once per tap.
#pragma once
#include <stdint.h>

#define TFC_TILE_H 8
#define TFC_TILE_W 16
#define TFC_LDS_P 24          // halo row pitch in pixels (== 8 mod 16: conflict-free ds_read_b128, see igemm.hip)
#define TFC_MAX_HH 11         // max halo rows (8 + 3)
#define TFC_MAX_HW 19         // max halo cols (16 + 3)
#define TFC_MAX_TAPS 16
#define TFC_MAX_PLANES 4
#define TFC_WPAD 16           // slack k-substeps at the end of a packed weight stream (prefetch distance headroom)
#define TFC_PART_WS_FLOATS (8 << 20)   // floats of the per-stream partial-sum scratch every reducing entry point takes (32 MiB; include/tfc_gan.h)
#define TFC_SN_BWD_PARTS 128           // workgroup partials (doubles) of the spectral-norm backward's dot product

#define TFC_DT_BF16 0
#define TFC_DT_F32 1

// epilogue flags
#define TFC_EP_BIAS 1
#define TFC_EP_STATS 2        // accumulate per-(img,channel) sum / sum of squares (InstanceNorm statistics)
#define TFC_EP_ACCUM 4        // out += result (skip-connection gradient accumulation)
#define TFC_EP_TANH_NCHW 8    // final layer: tanh, store fp32 NCHW
#define TFC_EP_LEAKY 16       // LeakyReLU(0.2) on the result (discriminator blocks: no normalisation between conv and activation)
#define TFC_EP_RELU 32        // ReLU on the result (the VGG feature stack of the LPIPS term)

struct TfcPlane {
  int dy0, dx0;               // halo origin relative to the tile origin, in plane coordinates
  int py, px;                 // source = plane*SS + (py,px)
  int hh, hw;                 // halo rows / cols actually used (<= TFC_MAX_HH / TFC_MAX_HW)
  int ntaps;
  int pad_;
  int tap_dy[TFC_MAX_TAPS];   // tap position inside the halo (>= 0)
  int tap_dx[TFC_MAX_TAPS];
  int tap_mask[TFC_MAX_TAPS]; // bit (ky*4+kx) set for every 4x4 filter tap of the torch weight this gather tap multiplies
                              // (one bit normally; several when duplicated taps of an upsampled input are collapsed: their weights add)
};

struct TfcGather {
  int IH, IW, in_pitch, SS;   // source tensor (NHWC, pitch in elements), source stride 1 or 2
  int Cin_pad;                // gathered channels, multiple of 8
  int GH, GW;                 // gather grid per image
  int tiles_y, tiles_x, nimg;
  int nplanes;
  int OH, OW, OS, OOY, OOX;   // output tensor dims, output stride / offset (sub-pixel phases)
  int out_pitch, Nout;        // output pixel pitch (elements), real output channels
  int ph_n;                   // sub-pixel phases folded into ONE launch (1 or 4): phase (py,px) = (ph >> 1, ph & 1)
  int ph_d0, ph_oo;           // per phase bit: shift of the halo origin (dy0/dx0) and of the output offset (OOY/OOX)
  struct TfcPlane plane[TFC_MAX_PLANES];
};

// Derived constants (host and device agree through these helpers).
static inline __host__ __device__ int tfc_pb(int cin_pad, int es) { int b = cin_pad * es; return b < 64 ? b : 64; }  // chunk bytes per pixel
static inline __host__ __device__ int tfc_ps(int pb) { return pb == 16 ? 16 : pb + 16; }                             // LDS pixel stride (bytes)
static inline __host__ __device__ int tfc_nsub(int ntaps, int pb) { return ntaps * (pb >> 4) / 2; }                  // k-substeps of one plane per chunk

// fused normalisation / activation / blur-pool kernels (elementwise.hip)
struct ActParams {
  int N, H, W, C;            // pre-pool tensor dims (x)
  int Ho, Wo;                // post-pool dims (== H,W when pool == 0 or 1)
  int x_pitch, o_pitch;      // pixel pitches (elements) of x / of the pooled-side tensor
  int pool;                  // 0 none, 1 blur stride 1, 2 blur stride 2
  int norm;                  // apply InstanceNorm using stats
  float slope;               // LeakyReLU slope (0 => ReLU, 1 => identity)
  float eps;
  unsigned drop_thresh24;    // 0 => no dropout
  unsigned seed;
  float drop_scale;
};

#ifdef __cplusplus
// Host-side hooks shared by the translation units of the library. All per-THREAD (the reference calls forward from several Python
// threads; nothing in the library is process-global mutable state).
extern thread_local int g_tfc_force_cfg;          // test hook (tfc_debug_set_igemm_config): -1 = heuristic tile choice
extern thread_local long long g_tfc_launch_count; // kernel launches issued by the conv-class launchers on this thread (profiling join key)
#define TFC_LAUNCH(...) do { ++g_tfc_launch_count; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// fixed-order sum of per-workgroup partials (elementwise.hip): out[g][j] += sum_p part[(g * nparts + p) * L + j]
hipError_t tfc_launch_part_reduce(const float* part, float* out, int G, int nparts, int L, hipStream_t st);

// torch-layout destination of a weight gradient, handed to the wgrad launchers: when a launch can reduce its split-K slabs straight into it
// (tfc_wgrad_reduce_fin_kernel) it sets `done` and the caller skips the separate finish pass
struct TfcWgradFin { float* grad; long long sn, sc; int accumulate; bool done; };

// ---- input pipeline (input.hip): resampling plan of one (H, W) file geometry -> 2 x (out x out). The plan buffer starts with this header,
// followed by the int tables the offsets (in ints from the start of the buffer) point at: bounds[out][2] = {first source index, tap count},
// coef[out][ksize] = taps in 22-bit fixed point.
struct TfcResizeAxis { int ksize; int bounds_off; int coef_off; int in_size; };
struct TfcResizePlan { TfcResizeAxis hA, hB, v; int out; int xsplit; int H; int W; };
#endif
