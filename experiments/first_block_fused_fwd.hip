// First block of both networks, fused (reference P16:102-127 UNetDown(channels, 64, normalize=False) = Conv2d(k4, s1, p1, bias=False) ->
// LeakyReLU(0.2) -> BlurPool(64, stride=2); P16:183-196 discriminator_block(2 * channels, 64) = SN-Conv2d(k4, s1, p1) -> LeakyReLU(0.2) ->
// BlurPool(64, stride=2)).
//
// At 256 x 256 x 32 images the 255 x 255 x 64 activation between the convolution and the blur-pool is 266 MB per call: the unfused pair
// (tfc_conv_c8_kernel + tfc_act_pool2_fwd_kernel) writes and re-reads it (68 + 80 us, both HBM-bound), and the backward
// (tfc_act_pool2_bwd_kernel + tfc_wgrad_kernel) re-reads it for the activation sign, writes the 266 MB gradient and reads that back
// (150 + 131 us). The fused kernels never materialise either tensor:
//
//   forward   x8 (33 MB) -> conv on MFMA (K = 8 channels x 16 taps = 128) -> LeakyReLU -> bf16 in LDS -> 4 x 4 binomial blur, stride 2,
//             reflect pad (1,2,1,2) -> pooled output (67 MB).                                       HBM: 33 + 67 MB instead of 33 + 266 + 266 + 67.
//   backward  pooled gradient (67 MB) + x8 (33 MB) -> transposed blur -> RECOMPUTED conv gives the activation sign -> d_raw bf16 in LDS ->
//             weight-gradient GEMM on MFMA (M = 64 outputs, N = 128 = taps x channels, K = pixels) + bias gradient.  HBM: 100 MB instead of 950.
//
// Arithmetic is kept where the unfused pair rounds: the activation is rounded to bf16 exactly where the unfused convolution stored it and
// the blur accumulates in fp32 in the same order, so the fused forward is bit-identical to the unfused pair; d_raw is rounded to bf16 where
// the unfused backward stored it.
#include "common.h"

namespace {
constexpr int FB_PH = 4, FB_PW = 15;                             // pooled outputs per tile
constexpr int FB_AR = 2 * FB_PH + 2, FB_AC = 2 * FB_PW + 2;       // activation region of a tile: 10 x 32
constexpr int FB_HR = FB_AR + 3, FB_HC = FB_AC + 3;               // input halo: 13 x 35 pixels of 8 channels (16 B)
constexpr int FB_PIN = 40;                                       // halo LDS pitch in pixels (two tile rows land 32 banks apart)
constexpr int FB_HB = FB_HR * FB_PIN * 16;                        // one halo buffer
constexpr int FB_ROWP = 64 * 2 + 16;                             // activation row in LDS: 64 channels bf16 + pad
__device__ __forceinline__ int fb_reflect(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }
__device__ __forceinline__ float fb_blur_w(int k) { return (k == 0 || k == 3) ? 0.125f : 0.375f; }
}  // namespace

// One workgroup = 4 waves; wave = (wm: activation column block of 16, wn: output-channel half). Weights-stationary (8 B fragments = 32 VGPRs
// per wave for the whole launch), persistent over tiles, the next tile's halo is requested before this tile's arithmetic.
__global__ void __launch_bounds__(256, 2)
tfc_c8_block_fwd_kernel(const bf16_t* __restrict__ in, int S, int in_pitch, const uint4* __restrict__ wp, int NB32, const float* __restrict__ bias,
                        const float* __restrict__ oscale, int leaky_pre, float slope_post, bf16_t* __restrict__ out, int out_pitch, int Po,
                        int tiles_y, int tiles_x, int nwork) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FB_HB + FB_AR * FB_AC * FB_ROWP];
  unsigned char* act = smem + 2 * FB_HB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r = lane & 31;
  const int G = gridDim.x;
  const int Ha = S - 1;                                          // activation size (k4 s1 p1)

  uint4 bw[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) bw[s] = wp[((size_t)s * NB32 + wn) * 64 + lane];
  const int n = wn * 32 + r;
  const float bv = bias ? bias[n] : 0.f;
  const float osc = oscale ? *oscale : 1.f;

  auto decode = [&](int w, int& img, int& py0, int& px0) {
    int tile = tfc_xcd_remap(w, nwork);
    const int tx = tile % tiles_x; tile /= tiles_x;
    const int ty = tile % tiles_y;
    img = tile / tiles_y;
    py0 = min(ty * FB_PH, Po - FB_PH);                           // the last tile of a row / column is anchored at the border (it recomputes
    px0 = min(tx * FB_PW, Po - FB_PW);                           // a few outputs of its neighbour): the reflect aliases stay inside its region
  };
  // halo: pixel hp of the 13 x 35 window <-> threads tid, tid + 256
  int hoff[2], hy[2], hx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int hp = tid + i * 256;
    hy[i] = hp / FB_HC; hx[i] = hp - hy[i] * FB_HC;
    hoff[i] = hp < FB_HR * FB_HC ? (hy[i] * FB_PIN + hx[i]) * 16 : -1;
  }
  uint4 hv[2];
  auto halo_load = [&](int img, int py0, int px0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = 2 * py0 - 2 + hy[i], x = 2 * px0 - 2 + hx[i];
      hv[i] = make_uint4(0, 0, 0, 0);
      if (hoff[i] >= 0 && y >= 0 && y < S && x >= 0 && x < S) hv[i] = *reinterpret_cast<const uint4*>(in + ((size_t)(img * S + y) * S + x) * in_pitch);
    }
  };
  auto halo_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (hoff[i] >= 0) *reinterpret_cast<uint4*>(buf + hoff[i]) = hv[i];
  };
  const int laneBase = ((r & 1) * FB_PIN + 16 * wm + (r >> 1) + h) * 16;   // lane half h takes the odd tap of a k-substep

  int w = blockIdx.x;
  if (w >= nwork) return;
  int img, py0, px0;
  decode(w, img, py0, px0);
  halo_load(img, py0, px0);
  halo_store(smem);
  __syncthreads();
  for (int k = 0;; ++k) {
    const bool more = w + G < nwork;
    int n_img = 0, n_py0 = 0, n_px0 = 0;
    if (more) { decode(w + G, n_img, n_py0, n_px0); halo_load(n_img, n_py0, n_px0); }
    const unsigned char* buf = smem + (k & 1) * FB_HB + laneBase;
    // ---- convolution of the 10 x 32 activation region: this wave's column block, five row pairs ----
#pragma unroll 1
    for (int q = 0; q < FB_AR / 2; ++q) {
      f32x16_t acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {                              // k-substep s = taps 2s (h = 0), 2s + 1 (h = 1) of the 4 x 4 raster
        const uint4 a = *reinterpret_cast<const uint4*>(buf + ((2 * q + (s >> 1)) * FB_PIN + 2 * (s & 1)) * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[s]), acc, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;          // pixel of the 32-pixel m-tile: (row & 1) = region row parity, row >> 1 = column
        const int rb = 2 * q + (row & 1), cb = 16 * wm + (row >> 1);
        float v = acc[j] * osc + bv;
        if (leaky_pre) v = fmaxf(v, 0.2f * v);
        *reinterpret_cast<bf16_t*>(act + (rb * FB_AC + cb) * FB_ROWP + n * 2) = f32_to_bf16(v);
      }
    }
    if (more) halo_store(smem + ((k + 1) & 1) * FB_HB);
    __syncthreads();
    // ---- blur-pool from LDS: unit = (pooled pixel, 8 channels) ----
    for (int uidx = tid; uidx < FB_PH * FB_PW * 8; uidx += 256) {
      const int u = uidx & 7, pp = uidx >> 3;
      const int i = pp / FB_PW, jx = pp - i * FB_PW;
      const int py = py0 + i, px = px0 + jx;
      int cbs[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) cbs[t] = fb_reflect(2 * px - 1 + t, Ha) - (2 * px0 - 1);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll 1
      for (int ti = 0; ti < 4; ++ti) {
        const int rb = fb_reflect(2 * py - 1 + ti, Ha) - (2 * py0 - 1);
        float hsum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) hsum[e] = 0.f;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          float v[8];
          unpack16<bf16_t>(*reinterpret_cast<const uint4*>(act + (rb * FB_AC + cbs[tj]) * FB_ROWP + u * 16), v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float t = v[e];
            if (!leaky_pre) t = t > 0.f ? t : t * slope_post;
            hsum[e] += fb_blur_w(tj) * t;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += fb_blur_w(ti) * hsum[e];
      }
      store_stream16(out + ((size_t)(img * Po + py) * Po + px) * out_pitch + u * 8, pack16<bf16_t>(o));
    }
    if (!more) break;
    __syncthreads();                                             // the activation region is free again
    w += G; img = n_img; py0 = n_py0; px0 = n_px0;
  }
}

hipError_t tfc_launch_c8_block_fwd(const void* in, int N, int S, int in_pitch, const void* wp, int NB32, const float* bias, const float* oscale,
                                   int leaky_pre, float slope_post, void* out, int out_pitch, hipStream_t st) {
  const int Ha = S - 1, Po = (Ha - 1) / 2 + 1;
  const int tiles_y = (Po + FB_PH - 1) / FB_PH, tiles_x = (Po + FB_PW - 1) / FB_PW;
  const int nwork = N * tiles_y * tiles_x;
  static int grid_cap = 0;
  if (!grid_cap) {
    int occ = 0, dev = 0, ncu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_c8_block_fwd_kernel, 256, 0);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    grid_cap = (occ < 1 ? 1 : occ) * ncu;
  }
  hipLaunchKernelGGL(tfc_c8_block_fwd_kernel, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), 0, st, (const bf16_t*)in, S, in_pitch,
                     (const uint4*)wp, NB32, bias, oscale, leaky_pre, slope_post, (bf16_t*)out, out_pitch, Po, tiles_y, tiles_x, nwork);
  return hipGetLastError();
}
