// Elementwise / reduction kernels of the LPIPS term of loss_G (reference P16:70-73, :598: lpips_pytorch.LPIPS(net_type='vgg', version='0.1')).
// The package itself is absent from the reference tree (pip dependency, weights fetched from the network): its published algorithm is restated
// here -- PARITY UNPINNED; the tests compare against a torch-CPU restatement with random weights.
//
//   z-score input   : x' = (x - shift) / scale per channel, fp32 NCHW -> NHWC (C padded to `pitch` channels, zeros)
//   conv3x3 + ReLU  : TFC_OP_CONV3 on the gather GEMM (igemm.hip), bias + TFC_EP_RELU in its epilogue
//   max-pool 2x2    : forward, and backward routing the gradient to the first maximum of the window (torch's rule)
//   ReLU backward   : dz = (y > 0) ? dy [+ extra] : 0        (extra = the gradient a tap layer receives from its LPIPS head)
//   head            : per pixel  n = f / (||f||_2 + 1e-10) over channels for both images, d = n_x - n_y, value = sum_c w_c d_c^2;
//                     per image  mean over pixels; forward value and d value / d f_x in ONE pass (the head's upstream gradient is a constant)
// All HBM-bound; activations NHWC with pitch == C (a multiple of 8), 16-byte units.
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256)
tfc_lpips_input_kernel(const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ scale, T* __restrict__ out,
                       int C, long long HW, long long total, int pitch) {
  constexpr int UE = ElemTraits<T>::UE;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;          // pixel index over N*H*W
  if (idx >= total) return;
  const long long n = idx / HW, p = idx % HW;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) v[c] = (x[(n * C + c) * HW + p] - shift[c]) / scale[c];
  T* po = out + idx * pitch;
  *reinterpret_cast<uint4*>(po) = pack16<T>(v);
  if (UE == 4) *reinterpret_cast<uint4*>(po + 4) = pack16<T>(v + 4);
}
// dx (fp32 NCHW) (=/+=) alpha * g[.., c] / scale[c]
template <typename T>
__global__ void __launch_bounds__(256)
tfc_lpips_input_bwd_kernel(const T* __restrict__ g, const float* __restrict__ scale, float* dx, int C, long long HW, long long total, int pitch,
                           float alpha, int accumulate) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long n = idx / HW, p = idx % HW;
  for (int c = 0; c < C; ++c) {
    const float v = alpha * ElemTraits<T>::ld(g + idx * pitch + c) / scale[c];
    float* o = dx + (n * C + c) * HW + p;
    *o = accumulate ? *o + v : v;
  }
}

// 2x2 / stride 2 max pooling, NHWC, one thread per (output pixel, 16-byte unit)
template <typename T>
__global__ void __launch_bounds__(256)
tfc_maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, long long total) {
  constexpr int UE = ElemTraits<T>::UE;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cu = C / UE, u = (int)(idx % cu);
  const long long op = idx / cu;
  const int Wo = W / 2, Ho = H / 2;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho);
  const long long n = op / ((long long)Wo * Ho);
  const T* p = x + ((n * H + 2 * oy) * W + 2 * ox) * (long long)C + u * UE;
  float a[UE], m[UE];
  unpack16<T>(*reinterpret_cast<const uint4*>(p), m);
  const long long offs[3] = {C, (long long)W * C, (long long)W * C + C};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    unpack16<T>(*reinterpret_cast<const uint4*>(p + offs[k]), a);
#pragma unroll
    for (int e = 0; e < UE; ++e) m[e] = fmaxf(m[e], a[e]);
  }
  *reinterpret_cast<uint4*>(y + op * C + u * UE) = pack16<T>(m);
}
// dx = dy where x is the FIRST maximum of its window (row-major scan), else 0
template <typename T>
__global__ void __launch_bounds__(256)
tfc_maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, long long total) {
  constexpr int UE = ElemTraits<T>::UE;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cu = C / UE, u = (int)(idx % cu);
  const long long op = idx / cu;
  const int Wo = W / 2, Ho = H / 2;
  const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho);
  const long long n = op / ((long long)Wo * Ho);
  const long long base = ((n * H + 2 * oy) * W + 2 * ox) * (long long)C + u * UE;
  const long long offs[4] = {0, C, (long long)W * C, (long long)W * C + C};
  float v[4][UE], g[UE], m[UE];
#pragma unroll
  for (int k = 0; k < 4; ++k) unpack16<T>(*reinterpret_cast<const uint4*>(x + base + offs[k]), v[k]);
  unpack16<T>(*reinterpret_cast<const uint4*>(dy + op * C + u * UE), g);
#pragma unroll
  for (int e = 0; e < UE; ++e) m[e] = fmaxf(fmaxf(v[0][e], v[1][e]), fmaxf(v[2][e], v[3][e]));
  bool taken[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) taken[e] = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float o[UE];
#pragma unroll
    for (int e = 0; e < UE; ++e) {
      const bool hit = !taken[e] && v[k][e] == m[e];
      o[e] = hit ? g[e] : 0.f;
      taken[e] = taken[e] || hit;
    }
    *reinterpret_cast<uint4*>(dx + base + offs[k]) = pack16<T>(o);
  }
}

// dz = (y > 0) ? dy (+ extra) : 0
template <typename T>
__global__ void __launch_bounds__(256)
tfc_relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ extra, T* __restrict__ dz, long long units) {
  constexpr int UE = ElemTraits<T>::UE;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < units; i += (long long)gridDim.x * 256) {
    float g[UE], a[UE], x[UE];
    unpack16<T>(load_stream16(dy + i * UE), g);
    unpack16<T>(load_stream16(y + i * UE), a);
    if (extra) {
      unpack16<T>(load_stream16(extra + i * UE), x);
#pragma unroll
      for (int e = 0; e < UE; ++e) g[e] += x[e];
    }
#pragma unroll
    for (int e = 0; e < UE; ++e) g[e] = a[e] > 0.f ? g[e] : 0.f;
    store_stream16(dz + i * UE, pack16<T>(g));
  }
}

// LPIPS head of one tap layer. A group of GS lanes (GS = power of two >= min(C / UE, 64)) owns one pixel, so a wave handles 64 / GS pixels at
// once and every lane holds 16-byte units u = gl, gl + GS, ... of its pixel. out[n] += (1 / HW) * sum_c w_c (nx_c - ny_c)^2 ;
// dfx (nullable) = gscale * d out[n] / d fx.
__device__ __forceinline__ float group_sum(float v, int gs) {
  for (int o = gs >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T>
__global__ void __launch_bounds__(256)
tfc_lpips_head_kernel(const T* __restrict__ fx, const T* __restrict__ fy, const float* __restrict__ w, float* out, T* __restrict__ dfx,
                      int C, long long HW, long long npix, float gscale, int GS) {
  constexpr int UE = ElemTraits<T>::UE;
  // Same-address float atomics from all 8 XCDs resolve at the memory side and cost ~0.3 us EACH (measured: 8192 of them per image made this
  // kernel 16 ms): a workgroup owns a CONTIGUOUS pixel range (so it touches few images), sums per image in LDS and issues one global atomic
  // per image it touched.
  constexpr int SLOTS = 64;
  __shared__ float simg[SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = lane & (GS - 1), grp = lane / GS, ppw = 64 / GS;          // lane in group, group in wave, pixels per wave
  const int units = C / UE;
  const long long chunk = (npix + gridDim.x - 1) / gridDim.x;
  const long long p0 = (long long)blockIdx.x * chunk, p1 = (p0 + chunk < npix) ? p0 + chunk : npix;
  if (p0 >= npix) return;
  const long long img0 = p0 / HW;
  if (threadIdx.x < SLOTS) simg[threadIdx.x] = 0.f;
  __syncthreads();
  float local = 0.f;
  long long img_prev = -1;
  auto commit = [&](long long img, float v) {
    const long long s = img - img0;
    if (s < SLOTS) atomicAdd(&simg[s], v / (float)HW);
    else atomicAdd(&out[img], v / (float)HW);
  };
  const long long stride = 4 * ppw;
  // every lane of a wave runs the same number of iterations (the shuffles need all lanes): out-of-range groups compute on pixel p1-1, masked
  for (long long base = p0 + (long long)wv * ppw; base < p1; base += stride) {
    const long long pix_raw = base + grp;
    const bool live = pix_raw < p1;
    const long long pix = live ? pix_raw : p1 - 1;
    const T* px = fx + pix * C;
    const T* py = fy + pix * C;
    float sx = 0.f, sy = 0.f;
    for (int u = gl; u < units; u += GS) {
      float a[UE], b[UE];
      unpack16<T>(*reinterpret_cast<const uint4*>(px + u * UE), a);
      unpack16<T>(*reinterpret_cast<const uint4*>(py + u * UE), b);
#pragma unroll
      for (int e = 0; e < UE; ++e) { sx += a[e] * a[e]; sy += b[e] * b[e]; }
    }
    sx = group_sum(sx, GS); sy = group_sum(sy, GS);
    const float rx = sqrtf(sx), ry = sqrtf(sy);
    const float ax = rx + 1e-10f, ay = ry + 1e-10f;
    float val = 0.f, dot = 0.f;                                    // dot = sum_c g_c x_c with g_c = 2 w_c d_c
    for (int u = gl; u < units; u += GS) {
      float a[UE], b[UE];
      unpack16<T>(*reinterpret_cast<const uint4*>(px + u * UE), a);
      unpack16<T>(*reinterpret_cast<const uint4*>(py + u * UE), b);
#pragma unroll
      for (int e = 0; e < UE; ++e) {
        const float d = a[e] / ax - b[e] / ay, wc = w[u * UE + e];
        val += wc * d * d;
        dot += 2.f * wc * d * a[e];
      }
    }
    val = group_sum(val, GS); dot = group_sum(dot, GS);
    if (dfx && live) {
      const float s = gscale / (float)HW;
      const float k2 = rx > 0.f ? dot / (rx * ax * ax) : 0.f;     // d n_c / d x_j = delta_cj / a - x_c x_j / (r a^2)
      for (int u = gl; u < units; u += GS) {
        float a[UE], b[UE], g[UE];
        unpack16<T>(*reinterpret_cast<const uint4*>(px + u * UE), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(py + u * UE), b);
#pragma unroll
        for (int e = 0; e < UE; ++e) {
          const float d = a[e] / ax - b[e] / ay, wc = w[u * UE + e];
          g[e] = s * (2.f * wc * d / ax - a[e] * k2);
        }
        *reinterpret_cast<uint4*>(dfx + pix * C + u * UE) = pack16<T>(g);
      }
    }
    if (gl == 0 && live) {                                         // group leader: running sum, committed when the image changes
      const long long img = pix / HW;
      if (img_prev >= 0 && img != img_prev) { commit(img_prev, local); local = 0.f; }
      local += val;
      img_prev = img;
    }
  }
  if (gl == 0 && img_prev >= 0) commit(img_prev, local);
  __syncthreads();
  if (threadIdx.x < SLOTS && img0 + threadIdx.x <= (p1 - 1) / HW) {
    const float v = simg[threadIdx.x];
    if (v != 0.f) atomicAdd(&out[img0 + threadIdx.x], v);
  }
}

#define TFC_DISPATCH_T(dt, CALL_BF16, CALL_F32) do { if ((dt) == TFC_DT_BF16) { CALL_BF16; } else { CALL_F32; } } while (0)

hipError_t tfc_launch_lpips_input(int dt, const float* x, const float* shift, const float* scale, void* out, int N, int C, long long HW, int pitch,
                                  hipStream_t st) {
  const long long tot = (long long)N * HW;
  const dim3 g((unsigned)((tot + 255) / 256));
  TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_lpips_input_kernel<bf16_t>, g, dim3(256), 0, st, x, shift, scale, (bf16_t*)out, C, HW, tot, pitch),
                 hipLaunchKernelGGL(tfc_lpips_input_kernel<float>, g, dim3(256), 0, st, x, shift, scale, (float*)out, C, HW, tot, pitch));
  return hipGetLastError();
}
hipError_t tfc_launch_lpips_input_bwd(int dt, const void* gr, const float* scale, float* dx, int N, int C, long long HW, int pitch, float alpha,
                                      int accumulate, hipStream_t st) {
  const long long tot = (long long)N * HW;
  const dim3 g((unsigned)((tot + 255) / 256));
  TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_lpips_input_bwd_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)gr, scale, dx, C, HW, tot, pitch, alpha, accumulate),
                 hipLaunchKernelGGL(tfc_lpips_input_bwd_kernel<float>, g, dim3(256), 0, st, (const float*)gr, scale, dx, C, HW, tot, pitch, alpha, accumulate));
  return hipGetLastError();
}
hipError_t tfc_launch_maxpool2(int dt, int bwd, const void* x, const void* dy, void* out, int N, int H, int W, int C, hipStream_t st) {
  const int ue = dt == TFC_DT_BF16 ? 8 : 4;
  const long long tot = (long long)N * (H / 2) * (W / 2) * (C / ue);
  const dim3 g((unsigned)((tot + 255) / 256));
  if (!bwd)
    TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_maxpool2_fwd_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, H, W, C, tot),
                   hipLaunchKernelGGL(tfc_maxpool2_fwd_kernel<float>, g, dim3(256), 0, st, (const float*)x, (float*)out, H, W, C, tot));
  else
    TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_maxpool2_bwd_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, H, W, C, tot),
                   hipLaunchKernelGGL(tfc_maxpool2_bwd_kernel<float>, g, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, H, W, C, tot));
  return hipGetLastError();
}
hipError_t tfc_launch_relu_bwd(int dt, const void* dy, const void* y, const void* extra, void* dz, long long n, hipStream_t st) {
  const int ue = dt == TFC_DT_BF16 ? 8 : 4;
  const long long units = n / ue;
  long long nb = (units + 255) / 256;
  if (nb > 8192) nb = 8192;
  TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_relu_bwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)extra, (bf16_t*)dz, units),
                 hipLaunchKernelGGL(tfc_relu_bwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)extra, (float*)dz, units));
  return hipGetLastError();
}
hipError_t tfc_launch_lpips_head(int dt, const void* fx, const void* fy, const float* w, float* out, void* dfx, int N, long long HW, int C, float gscale,
                                 hipStream_t st) {
  const long long npix = (long long)N * HW;
  const int units = C / (dt == TFC_DT_BF16 ? 8 : 4);
  int GS = 1;
  while (GS < units && GS < 64) GS <<= 1;
  const int ppb = 4 * (64 / GS);                                    // pixels per workgroup iteration
  long long nb = (npix + ppb - 1) / ppb;
  if (nb > 2048) nb = 2048;                                         // 8 workgroups per CU; each owns a contiguous pixel range
  TFC_DISPATCH_T(dt, hipLaunchKernelGGL(tfc_lpips_head_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)fx, (const bf16_t*)fy, w, out, (bf16_t*)dfx, C, HW, npix, gscale, GS),
                 hipLaunchKernelGGL(tfc_lpips_head_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)fx, (const float*)fy, w, out, (float*)dfx, C, HW, npix, gscale, GS));
  return hipGetLastError();
}
