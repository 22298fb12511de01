// Input pipeline of the training scripts (reference TFC-GAN-FFT/datasets_temp.py:38-123, ImageDataset.__getitem__) on the GPU:
//   one decoded file = an RGB uint8 image holding the visible image A (left half) and the thermal image B (right half)   (:52-55)
//   each half -> PIL Image.resize((256, 256), BICUBIC)                                                                  (:61-66)
//   T_B = LUT[B red channel], LUT = linspace(24, 38, 256)                                                              (:41-42, :69-70, :31-36)
//   ToTensor + Normalize(0.5, 0.5): (v / 255 - 0.5) / 0.5, CHW fp32                                                     (P16:479-482, :103-104)
//   B1..B4 = the 128 x 128 quadrants of B                                                                              (:80-110; views on the host side)
// PIL's resampler is restated exactly (Pillow src/libImaging/Resample.c, 8-bit path): separable, horizontal pass first, the filter support
// scaled by the reduction factor, coefficients normalised in double precision and quantised to 22 fractional bits, each pass accumulated in
// int32 from 1 << 21 and clipped to uint8. The coefficient tables are built on the host (api.hip: tfc_resize_plan_build) with the same double
// arithmetic, so the kernels are integer-only and BIT-EXACT against PIL (which is importable here and on the GPU box: tests pin against the
// very function the reference calls).
// Both kernels are tiny and HBM/latency-bound: 32 pairs of 640 x 480 are 59 MB in, 23 MB of uint8 intermediate, 50 MB out.
#include "common.h"

#define TFC_RS_PRECISION 22

__device__ __forceinline__ int clip8(int v) {
  v >>= TFC_RS_PRECISION;                                        // arithmetic shift, as Pillow's clip8() table index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// pass 1: horizontal. tmp[n][half][y][xo][c] uint8. One thread per (n, half, y, xo).
__global__ void __launch_bounds__(256)
tfc_resize_h_kernel(const uint8_t* __restrict__ src, long long img_stride, int row_stride, TfcResizePlan p, const int* __restrict__ plan,
                    uint8_t* __restrict__ tmp, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xo = (int)(idx % p.out);
  const int y = (int)((idx / p.out) % p.H);
  const int half = (int)((idx / ((long long)p.out * p.H)) & 1);
  const long long n = idx / ((long long)p.out * p.H * 2);
  const TfcResizeAxis ax = half ? p.hB : p.hA;
  const int xmin = plan[ax.bounds_off + 2 * xo], cnt = plan[ax.bounds_off + 2 * xo + 1];
  const int* k = plan + ax.coef_off + xo * ax.ksize;
  const uint8_t* row = src + n * img_stride + (long long)y * row_stride + (long long)((half ? p.xsplit : 0) + xmin) * 3;
  int s0 = 1 << (TFC_RS_PRECISION - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < cnt; ++t) {
    const int kt = k[t];
    s0 += row[3 * t + 0] * kt;
    s1 += row[3 * t + 1] * kt;
    s2 += row[3 * t + 2] * kt;
  }
  uint8_t* o = tmp + idx * 3;
  o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// pass 2: vertical + ToTensor + Normalize (+ temperature LUT for B). One thread per (n, half, yo, xo); A / B fp32 NCHW [N][3][out][out].
__global__ void __launch_bounds__(256)
tfc_resize_v_kernel(const uint8_t* __restrict__ tmp, TfcResizePlan p, const int* __restrict__ plan, const float* __restrict__ lut,
                    float* __restrict__ A, float* __restrict__ B, float* __restrict__ TB, uint8_t* __restrict__ A8, uint8_t* __restrict__ B8,
                    long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int xo = (int)(idx % p.out);
  const int yo = (int)((idx / p.out) % p.out);
  const int half = (int)((idx / ((long long)p.out * p.out)) & 1);
  const long long n = idx / ((long long)p.out * p.out * 2);
  const int ymin = plan[p.v.bounds_off + 2 * yo], cnt = plan[p.v.bounds_off + 2 * yo + 1];
  const int* k = plan + p.v.coef_off + yo * p.v.ksize;
  const uint8_t* col = tmp + (((n * 2 + half) * p.H + ymin) * (long long)p.out + xo) * 3;
  int s0 = 1 << (TFC_RS_PRECISION - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < cnt; ++t) {
    const int kt = k[t];
    const uint8_t* q = col + (long long)t * p.out * 3;
    s0 += q[0] * kt;
    s1 += q[1] * kt;
    s2 += q[2] * kt;
  }
  const int v[3] = {clip8(s0), clip8(s1), clip8(s2)};
  float* dst = half ? B : A;
  uint8_t* dst8 = half ? B8 : A8;
  const long long plane = (long long)p.out * p.out;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = (float)v[c] / 255.0f;                        // transforms.ToTensor(): byte -> float, div(255)
    dst[(n * 3 + c) * plane + (long long)yo * p.out + xo] = (t - 0.5f) / 0.5f;     // transforms.Normalize((0.5,)*3, (0.5,)*3)
    if (dst8) dst8[((n * p.out + yo) * (long long)p.out + xo) * 3 + c] = (uint8_t)v[c];
  }
  if (half && TB) TB[n * plane + (long long)yo * p.out + xo] = lut[v[0]];
}

hipError_t tfc_launch_pair_resize(const uint8_t* src, long long img_stride, int row_stride, int N, const TfcResizePlan& p, const int* plan,
                                  uint8_t* tmp, const float* lut, float* A, float* B, float* TB, uint8_t* A8, uint8_t* B8, hipStream_t st) {
  const long long t1 = (long long)N * 2 * p.H * p.out;
  hipLaunchKernelGGL(tfc_resize_h_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, st, src, img_stride, row_stride, p, plan, tmp, t1);
  const long long t2 = (long long)N * 2 * p.out * p.out;
  hipLaunchKernelGGL(tfc_resize_v_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, st, tmp, p, plan, lut, A, B, TB, A8, B8, t2);
  return hipGetLastError();
}
