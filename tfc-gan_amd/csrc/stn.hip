// Kernels of the STN21 configuration (reference TFC-STN/TFCGAN_STN21_Original_NewModel3_Official.py, "STN"):
//
//   tfc_affine_warp_fwd / _bwd : F.affine_grid(theta, size, align_corners=True) + F.grid_sample(src, grid, mode='bicubic',
//                                padding_mode='border', align_corners=True)  (STN:228-229) fused: the sampling grid is never stored.
//                                backward: gradient w.r.t. theta (the trainable path: theta comes from the localiser) and, optionally, w.r.t. src.
//   tfc_morph_grad_fwd / _bwd  : kornia.morphology.gradient(x, cross 3x3) = dilation - erosion (STN:444-449) with geodesic borders
//                                (out-of-image neighbours never win), and its backward (the gradient goes to the arg-max / arg-min pixel).
//   tfc_row_triplet_grad       : nn.TripletMarginLoss(margin, p=2) over the last dim (criterion_morph, STN:99, :457) with d loss / d anchor.
//
// All three are HBM-bound fp32 NCHW elementwise / gather kernels (3-channel 256 x 256 images): one thread per pixel, coalesced along x.
#include "common.h"

namespace {
constexpr float kA = -0.75f;                                      // cubic convolution coefficient of torch's bicubic

__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
  const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
  w[0] = ((kA * x0 - 5.f * kA) * x0 + 8.f * kA) * x0 - 4.f * kA;
  w[1] = ((kA + 2.f) * t - (kA + 3.f)) * t * t + 1.f;
  w[2] = ((kA + 2.f) * x2 - (kA + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((kA * x3 - 5.f * kA) * x3 + 8.f * kA) * x3 - 4.f * kA;
}
// MINUS d w / d t, in the form torch's get_cubic_coefficients_grad writes it (its caller subtracts: gix -= value * gOut * coeff)
__device__ __forceinline__ void cubic_dw(float t, float (&g)[4]) {
  float x;
  x = -1.f - t; g[0] = (-3.f * kA * x - 10.f * kA) * x - 8.f * kA;
  x = -t;       g[1] = (-3.f * (kA + 2.f) * x - 2.f * (kA + 3.f)) * x;
  x = 1.f - t;  g[2] = (3.f * (kA + 2.f) * x - 2.f * (kA + 3.f)) * x;
  x = 2.f - t;  g[3] = (3.f * kA * x - 10.f * kA) * x + 8.f * kA;
}
__device__ __forceinline__ int clip_border(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }

// source coordinate of output pixel (i, j) under theta (align_corners = True): base grid x_j = -1 + 2 j / (W-1), then unnormalise
__device__ __forceinline__ void src_coord(const float* th, int i, int j, int H, int W, float& ix, float& iy, float& bx, float& by) {
  bx = W > 1 ? -1.f + 2.f * (float)j / (float)(W - 1) : 0.f;
  by = H > 1 ? -1.f + 2.f * (float)i / (float)(H - 1) : 0.f;
  const float gx = th[0] * bx + th[1] * by + th[2];
  const float gy = th[3] * bx + th[4] * by + th[5];
  ix = (gx + 1.f) * 0.5f * (float)(W - 1);
  iy = (gy + 1.f) * 0.5f * (float)(H - 1);
}
}  // namespace

__global__ void __launch_bounds__(256)
tfc_affine_warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ theta, float* __restrict__ out, int N, int C, int H, int W) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * H * W) return;
  const int j = (int)(idx % W), i = (int)((idx / W) % H), n = (int)(idx / ((long long)W * H));
  float ix, iy, bx, by;
  src_coord(theta + n * 6, i, j, H, W, ix, iy, bx, by);
  const float fx = floorf(ix), fy = floorf(iy);
  float wx[4], wy[4];
  cubic_w(ix - fx, wx);
  cubic_w(iy - fy, wy);
  int xs[4], ys[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { xs[k] = clip_border((int)fx - 1 + k, W); ys[k] = clip_border((int)fy - 1 + k, H); }
  for (int c = 0; c < C; ++c) {
    const float* p = src + ((size_t)n * C + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += p[(size_t)ys[a] * W + xs[b]] * wx[b];
      acc += row * wy[a];
    }
    out[((size_t)n * C + c) * H * W + (size_t)i * W + j] = acc;
  }
}

// gout: d loss / d out.  dtheta = part[n][blockIdx.x][6] (one slot per workgroup, added in a fixed order by tfc_part_reduce_kernel);
// dsrc (nullable) [N][C][H][W] += ... (a scatter through float atomics; the launcher zeroes it)
__global__ void __launch_bounds__(256)
tfc_affine_warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ theta, const float* __restrict__ gout, float* dtheta,
                           float* dsrc, int N, int C, int H, int W) {
  __shared__ float red[4][6];
  const int n = blockIdx.y;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (idx < (long long)H * W) {
    const int j = (int)(idx % W), i = (int)(idx / W);
    float ix, iy, bx, by;
    src_coord(theta + n * 6, i, j, H, W, ix, iy, bx, by);
    const float fx = floorf(ix), fy = floorf(iy);
    float wx[4], wy[4], dx[4], dy[4];
    cubic_w(ix - fx, wx); cubic_w(iy - fy, wy);
    cubic_dw(ix - fx, dx); cubic_dw(iy - fy, dy);
    int xs[4], ys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xs[k] = clip_border((int)fx - 1 + k, W); ys[k] = clip_border((int)fy - 1 + k, H); }
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < C; ++c) {
      const size_t plane = ((size_t)n * C + c) * H * W;
      const float go = gout[plane + (size_t)i * W + j];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const size_t si = plane + (size_t)ys[a] * W + xs[b];
          const float v = src[si];
          gix -= go * v * dx[b] * wy[a];
          giy -= go * v * wx[b] * dy[a];
          if (dsrc) atomicAdd(&dsrc[si], go * wx[b] * wy[a]);
        }
    }
    gix *= 0.5f * (float)(W - 1);                                   // d ix / d gx (align_corners = True)
    giy *= 0.5f * (float)(H - 1);
    g6[0] = gix * bx; g6[1] = gix * by; g6[2] = gix;
    g6[3] = giy * bx; g6[4] = giy * by; g6[5] = giy;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) g6[k] = wave_sum(g6[k]);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 6; ++k) red[threadIdx.x >> 6][k] = g6[k];
  __syncthreads();
  if (threadIdx.x < 6)
    dtheta[((size_t)n * gridDim.x + blockIdx.x) * 6 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// morphological gradient with the 3 x 3 cross {(0,0), (-1,0), (1,0), (0,-1), (0,1)}: out = max - min over the in-image neighbours.
// amax / amin (nullable): offset code 0..4 of the arg-max / arg-min (first maximum in the order centre, up, down, left, right) for the backward.
__global__ void __launch_bounds__(256)
tfc_morph_grad_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, unsigned char* __restrict__ arg, long long planes, int H, int W) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= planes * H * W) return;
  const int j = (int)(idx % W), i = (int)((idx / W) % H);
  const float* p = x + idx;
  float mx = p[0], mn = p[0];
  int amx = 0, amn = 0;
  const int di[4] = {-1, 1, 0, 0}, dj[4] = {0, 0, -1, 1};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ii = i + di[k], jj = j + dj[k];
    if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
    const float v = p[di[k] * W + dj[k]];
    if (v > mx) { mx = v; amx = k + 1; }
    if (v < mn) { mn = v; amn = k + 1; }
  }
  out[idx] = mx - mn;
  if (arg) arg[idx] = (unsigned char)(amx | (amn << 4));
}
// backward as a GATHER (deterministic; round 2 scattered with float atomics): pixel p collects +g[q] from every in-image neighbour q (and itself)
// whose arg-max code points at p, and -g[q] where the arg-min code does; code k <-> offset {0, -W, +W, -1, +1}[k], so q = p - offset[k]
__global__ void __launch_bounds__(256)
tfc_morph_grad_bwd_kernel(const float* __restrict__ gout, const unsigned char* __restrict__ arg, float* __restrict__ dx, long long planes, int H, int W) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= planes * H * W) return;
  const int j = (int)(idx % W), i = (int)((idx / W) % H);
  const int off[5] = {0, -W, W, -1, 1};
  const bool ok[5] = {true, i + 1 < H, i >= 1, j + 1 < W, j >= 1};   // q = p - off[k] inside the image
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (!ok[k]) continue;
    const long long q = idx - off[k];
    const int a = arg[q];
    const float g = gout[q];
    if ((a & 15) == k) s += g;
    if ((a >> 4) == k) s -= g;
  }
  dx[idx] = s;
}

// triplet margin loss over rows of width W (p = 2, eps added to the difference like F.pairwise_distance), mean over rows, with d / d anchor.
static __device__ TfcRedSlot g_rowtrip2_slot;
__global__ void __launch_bounds__(256)
tfc_row_triplet_grad_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ ng, long long rows, int W, float margin,
                            float eps, float gscale, float* loss, float* da) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float lsum = 0.f;
  for (long long r = (long long)blockIdx.x * 4 + wv; r < rows; r += (long long)gridDim.x * 4) {
    const float* ra = a + r * W; const float* rp = p + r * W; const float* rn = ng + r * W;
    float sp = 0.f, sn = 0.f;
    for (int x = lane; x < W; x += 64) {
      const float dp = ra[x] - rp[x] + eps, dn = ra[x] - rn[x] + eps;
      sp += dp * dp; sn += dn * dn;
    }
    sp = wave_sum(sp); sn = wave_sum(sn);
    const float dap = sqrtf(sp), dan = sqrtf(sn);
    const float v = margin + dap - dan;
    if (lane == 0 && v > 0.f) lsum += v;
    if (da) {
      const float s = gscale / (float)rows;
      for (int x = lane; x < W; x += 64) {
        float g = 0.f;
        if (v > 0.f) g = (dap > 0.f ? (ra[x] - rp[x] + eps) / dap : 0.f) - (dan > 0.f ? (ra[x] - rn[x] + eps) / dan : 0.f);
        da[r * W + x] = g * s;
      }
    }
  }
  if (lane == 0) red[wv] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) tfc_block_commit(&g_rowtrip2_slot, ((double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]) / (double)rows, loss);
}

hipError_t tfc_launch_affine_warp_fwd(const float* src, const float* theta, float* out, int N, int C, int H, int W, hipStream_t st) {
  const long long tot = (long long)N * H * W;
  hipLaunchKernelGGL(tfc_affine_warp_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, theta, out, N, C, H, W);
  return hipGetLastError();
}
hipError_t tfc_launch_affine_warp_bwd(const float* src, const float* theta, const float* gout, float* dtheta, float* dsrc, float* part_ws, int N, int C,
                                      int H, int W, hipStream_t st) {
  const unsigned nbx = (unsigned)(((long long)H * W + 255) / 256);
  if (!part_ws || (long long)nbx * N * 6 > (long long)TFC_PART_WS_FLOATS) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(dtheta, 0, sizeof(float) * 6 * N, st);
  if (e != hipSuccess) return e;
  if (dsrc && (e = hipMemsetAsync(dsrc, 0, sizeof(float) * (size_t)N * C * H * W, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(tfc_affine_warp_bwd_kernel, dim3(nbx, N), dim3(256), 0, st, src, theta, gout, part_ws, dsrc, N, C, H, W);
  return tfc_launch_part_reduce(part_ws, dtheta, N, (int)nbx, 6, st);
}
hipError_t tfc_launch_morph_grad_fwd(const float* x, float* out, unsigned char* arg, long long planes, int H, int W, hipStream_t st) {
  const long long tot = planes * H * W;
  hipLaunchKernelGGL(tfc_morph_grad_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, out, arg, planes, H, W);
  return hipGetLastError();
}
hipError_t tfc_launch_morph_grad_bwd(const float* gout, const unsigned char* arg, float* dx, long long planes, int H, int W, hipStream_t st) {
  const long long tot = planes * H * W;
  hipLaunchKernelGGL(tfc_morph_grad_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gout, arg, dx, planes, H, W);
  return hipGetLastError();
}
hipError_t tfc_launch_row_triplet_grad(const float* a, const float* p, const float* ng, long long rows, int W, float margin, float eps, float gscale,
                                       float* loss, float* da, hipStream_t st) {
  hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  long long nb = (rows + 3) / 4;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(tfc_row_triplet_grad_kernel, dim3((unsigned)nb), dim3(256), 0, st, a, p, ng, rows, W, margin, eps, gscale, loss, da);
  return hipGetLastError();
}
