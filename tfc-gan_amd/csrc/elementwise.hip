// HBM-bound fused kernels of the TFC-GAN hot path (gfx950): everything between the convolutions.
// All activation tensors are NHWC with an explicit pixel pitch; each lane moves 16 bytes (8 bf16 / 4 fp32).
//
//   act_fwd   : [InstanceNorm] -> LeakyReLU/ReLU/identity -> [BlurPool stride 2 | blur stride 1] -> [Dropout]
//               (+ optional InstanceNorm statistics of the result)   reference: UNetDown / UNetUp bodies,
//               TFC-GAN-FFT/TFCGAN_multigpu_patchFFT_16P.py:102-134, Discriminator1 blocks :187-200
//   act_bwd   : the exact transpose, in reduce / apply phases for the InstanceNorm backward
//   pack/unpack NCHW fp32 <-> NHWC8, tanh backward, column sums (bias grads), spectral-norm power iteration,
//   relativistic BCE-with-logits, Adam, axpby, dropout-mask export.
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// Deterministic cross-workgroup sums (round 3). Every kernel that used to add its workgroup's partial into a small fp32 vector with
// memory-side float atomics (InstanceNorm sums, their backward reductions, bias gradients) now STORES it,
//     part[(g * nparts + p) * L + j]        g: image (or 0), p: the workgroup's FIXED slot, j: element,
// and this kernel adds the slots in ONE fixed order: eight part-lanes take p = q, q + 8, ... ascending, then lanes 0..7 in order. Same inputs
// => same bits, whatever order the producing workgroups ran in (float atomics gave run-to-run cos 0.991 on bf16 generator gradients).
// out[g][j] += total (callers hand over zeroed or running sums, as they did to the atomics).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
tfc_part_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int L) {
  __shared__ float red[8][32];
  const int jl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl, g = blockIdx.y;
  float s = 0.f;
  if (j < L) {
    const float* p0 = part + (size_t)g * nparts * L + j;
#pragma unroll 4
    for (int p = q; p < nparts; p += 8) s += p0[(size_t)p * L];
  }
  red[q][jl] = s;
  __syncthreads();
  if (q == 0 && j < L) {
    float t = red[0][jl];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][jl];
    out[(size_t)g * L + j] += t;
  }
}
// out[j] += sum_p part[p * stride + j], j < L <= 8 (the generator head's bias gradient: a handful of columns, up to 1024 slots): 32 part-lanes per column
__global__ void __launch_bounds__(256)
tfc_part_reduce_strided_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, int stride, int L) {
  __shared__ float red[8][32];
  const int q = threadIdx.x & 31, j = threadIdx.x >> 5;
  float s = 0.f;
  if (j < L)
    for (int p = q; p < nparts; p += 32) s += part[(size_t)p * stride + j];
  red[j][q] = s;
  __syncthreads();
  if (q == 0 && j < L) {
    float t = 0.f;
    for (int i = 0; i < 32; ++i) t += red[j][i];
    out[j] += t;
  }
}
hipError_t tfc_launch_part_reduce(const float* part, float* out, int G, int nparts, int L, hipStream_t st) {
  hipLaunchKernelGGL(tfc_part_reduce_kernel, dim3((L + 31) / 32, G), dim3(256), 0, st, part, out, nparts, L);
  return hipGetLastError();
}
static inline bool part_fits(long long floats) { return floats <= (long long)TFC_PART_WS_FLOATS; }

// BlurPool taps: [1,3,3,1]/8 per dimension (antialiased_cnns.BlurPool, filt_size 4), reflect pad (1,2).
__device__ __forceinline__ float blur_w(int k) { return (k == 0 || k == 3) ? 0.125f : 0.375f; }
__device__ __forceinline__ int reflect_idx(int p, int n) { return p < 0 ? -p : (p >= n ? 2 * n - 2 - p : p); }

// block = 256 threads = PPB pixels x CV channel vectors; grid = (pixel blocks, N); grid-stride over pixels
template <typename T, bool STATS_OUT>
__global__ void __launch_bounds__(256)
tfc_act_fwd_kernel(const ActParams p, const T* __restrict__ x, const float* __restrict__ stats, T* __restrict__ out,
                   float* stats_out) {
  constexpr int UE = ElemTraits<T>::UE;
  __shared__ float red[2][256 * 8 / 8 * 8];                      // [2][256*UE] worst case UE=8 -> 2048 floats each
  const int CV = p.C / UE;
  const int PPB = 256 / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const int n = blockIdx.y;
  const int npix = p.Ho * p.Wo;
  float mean[UE], rstd[UE];
  if (p.norm) {
    const float inv = 1.f / (float)(p.H * p.W);
#pragma unroll
    for (int e = 0; e < UE; ++e) {
      const float s1 = stats[((size_t)n * p.C + cv * UE + e) * 2 + 0];
      const float s2 = stats[((size_t)n * p.C + cv * UE + e) * 2 + 1];
      const float m = s1 * inv;
      const float var = fmaxf(s2 * inv - m * m, 0.f);
      mean[e] = m;
      rstd[e] = rsqrtf(var + p.eps);
    }
  }
  float a1[UE], a2[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { a1[e] = 0.f; a2[e] = 0.f; }

  const T* xn = x + (size_t)n * p.H * p.W * p.x_pitch;
  T* on = out + (size_t)n * npix * p.o_pitch;
  if (pl < PPB) {
    for (int pix = blockIdx.x * PPB + pl; pix < npix; pix += gridDim.x * PPB) {
      const int oy = pix / p.Wo, ox = pix - oy * p.Wo;
      float r[UE];
#pragma unroll
      for (int e = 0; e < UE; ++e) r[e] = 0.f;
      if (p.pool == 0) {
        float v[UE];
        unpack16<T>(*reinterpret_cast<const uint4*>(xn + (size_t)(oy * p.W + ox) * p.x_pitch + cv * UE), v);
#pragma unroll
        for (int e = 0; e < UE; ++e) {
          float t = p.norm ? (v[e] - mean[e]) * rstd[e] : v[e];
          r[e] = t > 0.f ? t : t * p.slope;
        }
      } else {
        const int s = p.pool;                                     // 1 or 2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int yy = reflect_idx(oy * s - 1 + i, p.H);
          const float wy = blur_w(i);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int xx = reflect_idx(ox * s - 1 + j, p.W);
            const float w = wy * blur_w(j);
            float v[UE];
            unpack16<T>(*reinterpret_cast<const uint4*>(xn + (size_t)(yy * p.W + xx) * p.x_pitch + cv * UE), v);
#pragma unroll
            for (int e = 0; e < UE; ++e) {
              float t = p.norm ? (v[e] - mean[e]) * rstd[e] : v[e];
              t = t > 0.f ? t : t * p.slope;
              r[e] += w * t;
            }
          }
        }
      }
      if (p.drop_thresh24) {
        const uint32_t base = (uint32_t)(((size_t)n * npix + pix) * p.C + cv * UE);
#pragma unroll
        for (int e = 0; e < UE; ++e) r[e] = tfc_keep(p.seed, base + e, p.drop_thresh24) ? r[e] * p.drop_scale : 0.f;
      }
      const uint4 pk = pack16<T>(r);
      if (STATS_OUT) {                                             // InstanceNorm sums of the values actually STORED (the tensor the normaliser reads)
        float rs[UE];
        unpack16<T>(pk, rs);
#pragma unroll
        for (int e = 0; e < UE; ++e) { a1[e] += rs[e]; a2[e] += rs[e] * rs[e]; }
      }
      *reinterpret_cast<uint4*>(on + (size_t)pix * p.o_pitch + cv * UE) = pk;
    }
  }
  if (STATS_OUT) {
    // block reduction over the PPB pixel lanes (fixed order), then this workgroup's slot of the partial buffer: stats_out = part[n][blockIdx.x][C][2]
#pragma unroll
    for (int e = 0; e < UE; ++e) { red[0][threadIdx.x * UE + e] = a1[e]; red[1][threadIdx.x * UE + e] = a2[e]; }
    __syncthreads();
    const int nch = CV * UE;                                      // == C
    float2* slot = reinterpret_cast<float2*>(stats_out) + ((size_t)n * gridDim.x + blockIdx.x) * p.C;
    for (int c = threadIdx.x; c < nch; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      const int ccv = c / UE, ce = c % UE;
      for (int q = 0; q < PPB; ++q) { s1 += red[0][(q * CV + ccv) * UE + ce]; s2 += red[1][(q * CV + ccv) * UE + ce]; }
      slot[c] = make_float2(s1, s2);
    }
  }
}

// Stride-2 BlurPool forward, separable and blocked: a thread produces a 2x2 block of outputs from its 6x6 input window --
// 36 loads / activations per 4 outputs instead of 64 -- horizontal [1,3,3,1]/8 pass per input row, then the vertical pass.
// ACT = false: no normalisation and identity activation (discriminator blocks, whose LeakyReLU already ran in the conv epilogue):
// a pure anti-aliased down-sampling, without the 2.25x-redundant activation arithmetic.
template <typename T, bool ACT>
__global__ void __launch_bounds__(256)
tfc_act_pool2_fwd_kernel(const ActParams p, const T* __restrict__ x, const float* __restrict__ stats, T* __restrict__ out) {
  constexpr int UE = ElemTraits<T>::UE;
  const int CV = p.C / UE;
  const int PPB = 256 / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const int n = blockIdx.y;
  const int bw = (p.Wo + 1) >> 1, bh = (p.Ho + 1) >> 1;
  const int nob = bw * bh;
  const int npix = p.Ho * p.Wo;
  float mean[UE], rstd[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { mean[e] = 0.f; rstd[e] = 1.f; }
  if (p.norm) {
    const float inv = 1.f / (float)(p.H * p.W);
#pragma unroll
    for (int e = 0; e < UE; ++e) {
      const float s1 = stats[((size_t)n * p.C + cv * UE + e) * 2 + 0];
      const float s2 = stats[((size_t)n * p.C + cv * UE + e) * 2 + 1];
      const float m = s1 * inv;
      mean[e] = m;
      rstd[e] = rsqrtf(fmaxf(s2 * inv - m * m, 0.f) + p.eps);
    }
  }
  const T* xn = x + (size_t)n * p.H * p.W * p.x_pitch;
  T* on = out + (size_t)n * npix * p.o_pitch;
  if (pl >= PPB) return;
  for (int ob = blockIdx.x * PPB + pl; ob < nob; ob += gridDim.x * PPB) {
    const int oby = ob / bw, obx = ob - oby * bw;
    int cx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) cx[j] = min(max(reflect_idx(4 * obx - 1 + j, p.W), 0), p.W - 1);
    float o00[UE], o01[UE], o10[UE], o11[UE];
#pragma unroll
    for (int e = 0; e < UE; ++e) { o00[e] = 0.f; o01[e] = 0.f; o10[e] = 0.f; o11[e] = 0.f; }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {                                // one input row at a time: 6 loads in flight, not 36
      const int yy = min(max(reflect_idx(4 * oby - 1 + i, p.H), 0), p.H - 1);
      const T* row = xn + (size_t)yy * p.W * p.x_pitch + cv * UE;
      float h0[UE], h1[UE];
#pragma unroll
      for (int e = 0; e < UE; ++e) { h0[e] = 0.f; h1[e] = 0.f; }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float v[UE];
        unpack16<T>(*reinterpret_cast<const uint4*>(row + (size_t)cx[j] * p.x_pitch), v);
#pragma unroll
        for (int e = 0; e < UE; ++e) {
          float t = v[e];
          if (ACT) {
            t = p.norm ? (t - mean[e]) * rstd[e] : t;
            t = t > 0.f ? t : t * p.slope;
          }
          if (j < 4) h0[e] += blur_w(j) * t;
          if (j >= 2) h1[e] += blur_w(j - 2) * t;
        }
      }
      const float wa = i < 4 ? blur_w(i) : 0.f, wb = i >= 2 ? blur_w(i - 2) : 0.f;
#pragma unroll
      for (int e = 0; e < UE; ++e) {
        o00[e] += wa * h0[e]; o01[e] += wa * h1[e];
        o10[e] += wb * h0[e]; o11[e] += wb * h1[e];
      }
    }
    auto store = [&](int oy, int ox, float* r) {
      if (oy >= p.Ho || ox >= p.Wo) return;
      const int pix = oy * p.Wo + ox;
      if (p.drop_thresh24) {
        const uint32_t base = (uint32_t)(((size_t)n * npix + pix) * p.C + cv * UE);
#pragma unroll
        for (int e = 0; e < UE; ++e) r[e] = tfc_keep(p.seed, base + e, p.drop_thresh24) ? r[e] * p.drop_scale : 0.f;
      }
      *reinterpret_cast<uint4*>(on + (size_t)pix * p.o_pitch + cv * UE) = pack16<T>(r);
    };
    store(2 * oby, 2 * obx, o00);
    store(2 * oby, 2 * obx + 1, o01);
    store(2 * oby + 1, 2 * obx, o10);
    store(2 * oby + 1, 2 * obx + 1, o11);
  }
}

// backward. mode 0: dx = g' (no norm)   mode 1: rstats += (sum g', sum g'*xhat)   mode 2: dx = rstd*(g' - mg - xhat*mgx)
//   g' = [blur^T](dropmask * dout) * act'(xhat);  xhat = norm ? (x-mean)*rstd : x;  use_x == 0 => act' = 1
template <typename T, int MODE, int POOL>
__global__ void __launch_bounds__(256)
tfc_act_bwd_kernel(const ActParams p, const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ stats,
                   float* rstats, T* __restrict__ dx, int use_x, int dx_pitch) {
  constexpr int UE = ElemTraits<T>::UE;
  __shared__ float red[2][2048];
  const int CV = p.C / UE;
  const int PPB = 256 / CV;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const int n = blockIdx.y;
  const int npix = p.H * p.W;
  const int nopix = p.Ho * p.Wo;
  float mean[UE], rstd[UE], mg[UE], mgx[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { mean[e] = 0.f; rstd[e] = 1.f; mg[e] = 0.f; mgx[e] = 0.f; }
  if (p.norm) {
    const float inv = 1.f / (float)npix;
#pragma unroll
    for (int e = 0; e < UE; ++e) {
      const size_t si = ((size_t)n * p.C + cv * UE + e) * 2;
      const float m = stats[si] * inv;
      const float var = fmaxf(stats[si + 1] * inv - m * m, 0.f);
      mean[e] = m;
      rstd[e] = rsqrtf(var + p.eps);
      if (MODE == 2) { mg[e] = rstats[si] * inv; mgx[e] = rstats[si + 1] * inv; }
    }
  }
  float a1[UE], a2[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  const T* xn = x + (size_t)n * npix * p.x_pitch;
  const T* dn = dout + (size_t)n * nopix * p.o_pitch;
  T* dxn = dx + (size_t)n * npix * dx_pitch;
  if (pl < PPB) {
    for (int pix = blockIdx.x * PPB + pl; pix < npix; pix += gridDim.x * PPB) {
      const int y = pix / p.W, xq = pix - y * p.W;
      float g[UE];
#pragma unroll
      for (int e = 0; e < UE; ++e) g[e] = 0.f;
      if (POOL == 0) {
        unpack16<T>(*reinterpret_cast<const uint4*>(dn + (size_t)pix * p.o_pitch + cv * UE), g);
        if (p.drop_thresh24) {
          const uint32_t base = (uint32_t)(((size_t)n * nopix + pix) * p.C + cv * UE);
#pragma unroll
          for (int e = 0; e < UE; ++e) g[e] = tfc_keep(p.seed, base + e, p.drop_thresh24) ? g[e] * p.drop_scale : 0.f;
        }
      } else {
        auto tap = [&](int oyy, int oxx, float w) {
          const int opix = oyy * p.Wo + oxx;
          float v[UE];
          unpack16<T>(*reinterpret_cast<const uint4*>(dn + (size_t)opix * p.o_pitch + cv * UE), v);
          if (p.drop_thresh24) {
            const uint32_t base = (uint32_t)(((size_t)n * nopix + opix) * p.C + cv * UE);
#pragma unroll
            for (int e = 0; e < UE; ++e) v[e] = tfc_keep(p.seed, base + e, p.drop_thresh24) ? v[e] * p.drop_scale : 0.f;
          }
#pragma unroll
          for (int e = 0; e < UE; ++e) g[e] += w * v[e];
        };
        const bool yin = y >= 2 && y <= p.H - 4, xin = xq >= 2 && xq <= p.W - 4;
        if (POOL == 2 && yin && xin) {
          // interior of a stride-2 BlurPool: exactly two outputs per dimension read this input, with taps {3/8, 1/8}
          const int oy0 = (y + 1) >> 1, ox0 = (xq + 1) >> 1;
          const float wy0 = (y & 1) ? 0.125f : 0.375f, wx0 = (xq & 1) ? 0.125f : 0.375f;
          const float wy1 = 0.5f - wy0, wx1 = 0.5f - wx0;
          tap(oy0, ox0, wy0 * wx0);
          tap(oy0, ox0 - 1, wy0 * wx1);
          tap(oy0 - 1, ox0, wy1 * wx0);
          tap(oy0 - 1, ox0 - 1, wy1 * wx1);
        } else if (POOL == 1 && yin && xin) {
          // interior of the stride-1 blur: the 4x4 outputs (y+1-ky, x+1-kx) read this input with taps [1,3,3,1]^2/64
#pragma unroll 1
          for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) tap(y + 1 - ky, xq + 1 - kx, blur_w(ky) * blur_w(kx));
        } else {
          // border (reflect-pad aliases): enumerate (alias, tap) pairs; alias a: 0 -> p = y, 1 -> p = -1 (y == 1),
          // 2 -> p = n (y == n-2), 3 -> p = n+1 (y == n-3)
          constexpr int st = POOL == 0 ? 1 : POOL;
          for (int ay = 0; ay < 4; ++ay) {
            if ((ay == 1 && y != 1) || (ay == 2 && y != p.H - 2) || (ay == 3 && y != p.H - 3)) continue;
            const int py = ay == 0 ? y : (ay == 1 ? -1 : (ay == 2 ? p.H : p.H + 1));
            for (int ky = 0; ky < 4; ++ky) {
              const int ty = py + 1 - ky;
              if (ty < 0 || (st == 2 && (ty & 1))) continue;
              const int oyy = st == 2 ? (ty >> 1) : ty;
              if (oyy >= p.Ho) continue;
              for (int ax = 0; ax < 4; ++ax) {
                if ((ax == 1 && xq != 1) || (ax == 2 && xq != p.W - 2) || (ax == 3 && xq != p.W - 3)) continue;
                const int px = ax == 0 ? xq : (ax == 1 ? -1 : (ax == 2 ? p.W : p.W + 1));
                for (int kx = 0; kx < 4; ++kx) {
                  const int tx = px + 1 - kx;
                  if (tx < 0 || (st == 2 && (tx & 1))) continue;
                  const int oxx = st == 2 ? (tx >> 1) : tx;
                  if (oxx >= p.Wo) continue;
                  tap(oyy, oxx, blur_w(ky) * blur_w(kx));
                }
              }
            }
          }
        }
      }
      float xh[UE];
      if (use_x) {
        float v[UE];
        unpack16<T>(*reinterpret_cast<const uint4*>(xn + (size_t)pix * p.x_pitch + cv * UE), v);
#pragma unroll
        for (int e = 0; e < UE; ++e) {
          xh[e] = p.norm ? (v[e] - mean[e]) * rstd[e] : v[e];
          g[e] *= (xh[e] > 0.f ? 1.f : p.slope);
        }
      } else {
#pragma unroll
        for (int e = 0; e < UE; ++e) xh[e] = 0.f;
      }
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < UE; ++e) { a1[e] += g[e]; a2[e] += g[e] * xh[e]; }
      } else {
        if (MODE == 0 && rstats) {
#pragma unroll
          for (int e = 0; e < UE; ++e) a1[e] += g[e];             // bias gradient of the convolution that produced x
        }
        float r[UE];
#pragma unroll
        for (int e = 0; e < UE; ++e) r[e] = (MODE == 2) ? rstd[e] * (g[e] - mg[e] - xh[e] * mgx[e]) : g[e];
        *reinterpret_cast<uint4*>(dxn + (size_t)pix * dx_pitch + cv * UE) = pack16<T>(r);
      }
    }
  }
  if (MODE == 1) {
#pragma unroll
    for (int e = 0; e < UE; ++e) { red[0][threadIdx.x * UE + e] = a1[e]; red[1][threadIdx.x * UE + e] = a2[e]; }
    __syncthreads();
    const int nch = CV * UE;
    for (int c = threadIdx.x; c < nch; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      const int ccv = c / UE, ce = c % UE;
      for (int q = 0; q < PPB; ++q) { s1 += red[0][(q * CV + ccv) * UE + ce]; s2 += red[1][(q * CV + ccv) * UE + ce]; }
      reinterpret_cast<float2*>(rstats)[((size_t)n * gridDim.x + blockIdx.x) * p.C + c] = make_float2(s1, s2);   // rstats = part[n][blockIdx.x][C][2]
    }
  } else if (MODE == 0 && rstats) {                              // rstats = part[n][blockIdx.x][C]: per-image column sums of dx (bias gradient)
#pragma unroll
    for (int e = 0; e < UE; ++e) red[0][threadIdx.x * UE + e] = a1[e];
    __syncthreads();
    const int nch = CV * UE;
    for (int c = threadIdx.x; c < nch; c += 256) {
      float s1 = 0.f;
      const int ccv = c / UE, ce = c % UE;
      for (int q = 0; q < PPB; ++q) s1 += red[0][(q * CV + ccv) * UE + ce];
      rstats[((size_t)n * gridDim.x + blockIdx.x) * p.C + c] = s1;
    }
  }
}

// Stride-2 BlurPool backward, tiled: a workgroup owns a 16 x 32 tile of INPUT pixels and a 64-channel slice. The (10 x 18)-pixel
// window of the pooled gradient that the tile's interior pixels read is staged ONCE in LDS (with the dropout mask applied once
// per element, not once per tap); each pixel then takes its 2 x 2 taps (3 x 3 next to the bottom / right reflect border) from
// LDS with weights tabulated per tile row / column, so interior and border pixels run the same short code.
// MODE as in tfc_act_bwd_kernel. SIGNW (mode 0, 64 channels): `x` is not the activation but its sign words (8 bytes per pixel, bit c = (x[c] > 0), as
// tfc_first_block_fwd / tfc_conv_first_fwd leave them) -- a thread reads the ONE byte of its eight channels instead of 16 bytes.
template <typename T, int MODE, bool NORM, bool SIGNW = false>
__global__ void __launch_bounds__(256)
tfc_act_pool2_bwd_kernel(const ActParams p, const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ stats,
                         float* rstats, T* __restrict__ dx, int dx_pitch, int tiles_x, int cslice) {
  constexpr int UE = ElemTraits<T>::UE;
  constexpr int TH = 16, TW = 32, WH = TH / 2 + 3, WW = TW / 2 + 3;   // window 11 x 19 output pixels (covers the reflect aliases too)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int CVS = cslice / UE;                                    // channel vectors of this slice (<= 16)
  const int PL = 256 / CVS;                                       // pixel lanes
  const int cvl = threadIdx.x % CVS, pl = threadIdx.x / CVS;
  const int n = blockIdx.y;
  const int c0 = blockIdx.z * cslice;
  const int cvg = c0 / UE + cvl;                                  // global channel-vector index
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const int oyb = ty0 / 2 - 1, oxb = tx0 / 2 - 1;                 // window origin in pooled coordinates
  const int npix = p.H * p.W, nopix = p.Ho * p.Wo;
  const T* dn = dout + (size_t)n * nopix * p.o_pitch;
  const T* xn = x + (size_t)n * npix * p.x_pitch;
  const unsigned char* sgn = reinterpret_cast<const unsigned char*>(x) + (size_t)n * npix * 8;   // SIGNW
  T* dxn = dx + (size_t)n * npix * dx_pitch;
  uint4* win = reinterpret_cast<uint4*>(smem_raw);                // [WH][WW][CVS] 16-byte units
  float4* wrow = reinterpret_cast<float4*>(smem_raw + (size_t)WH * WW * CVS * 16);   // [TH] tap weights of pooled rows oy0-1, oy0, oy0+1
  float4* wcol = wrow + TH;                                       // [TW] same for columns
  float* red = reinterpret_cast<float*>(smem_raw);                // [2][256*UE] MODE 0/1 reductions: reuses the window after the main loop

  // Per input row y the transposed blur touches pooled rows oy0-1, oy0, oy0+1 with oy0 = (y+1)>>1 (the third one only through the
  // reflect aliases next to the bottom / right border). The weights -- reflect aliases merged -- are tabulated once per tile, so
  // the per-pixel code is the same short sequence for interior and border pixels.
  if (threadIdx.x < TH + TW) {
    const bool isrow = threadIdx.x < TH;
    const int q = isrow ? ty0 + threadIdx.x : tx0 + (threadIdx.x - TH);
    const int L = isrow ? p.H : p.W, Lo = isrow ? p.Ho : p.Wo;
    float w3[3] = {0.f, 0.f, 0.f};
    if (q < L) {
      const int o0 = (q + 1) >> 1;
      for (int a = 0; a < 4; ++a) {
        if ((a == 1 && q != 1) || (a == 2 && q != L - 2) || (a == 3 && q != L - 3)) continue;
        const int pq = a == 0 ? q : (a == 1 ? -1 : (a == 2 ? L : L + 1));
        for (int k = 0; k < 4; ++k) {
          const int t = pq + 1 - k;
          if (t < 0 || (t & 1) || (t >> 1) >= Lo) continue;
          const int d = (t >> 1) - o0 + 1;
          if (d == 0) w3[0] += blur_w(k); else if (d == 1) w3[1] += blur_w(k); else if (d == 2) w3[2] += blur_w(k);
        }
      }
    }
    (isrow ? wrow[threadIdx.x] : wcol[threadIdx.x - TH]) = make_float4(w3[0], w3[1], w3[2], 0.f);
  }
  // window staging, seven 16-byte units per thread and batch: all loads of a batch are requested (clamped address) before the first is stored -- one
  // dependent load per loop iteration left a single request in flight per lane (the pattern that cost tfc_head_fwd_kernel half its time)
  for (int b0 = 0; b0 < WH * WW * CVS; b0 += 7 * 256) {
    uint4 wv[7];
    int wopix[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int i = b0 + u * 256 + threadIdx.x;
      const int cv = i % CVS, wp = i / CVS;
      const int oy = oyb + wp / WW, ox = oxb + wp % WW;
      const bool ok = i < WH * WW * CVS && oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
      wopix[u] = ok ? oy * p.Wo + ox : -1;
      wv[u] = *reinterpret_cast<const uint4*>(dn + (size_t)(ok ? wopix[u] : 0) * p.o_pitch + (c0 / UE + cv) * UE);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int i = b0 + u * 256 + threadIdx.x;
      if (i >= WH * WW * CVS) continue;
      uint4 v = wopix[u] >= 0 ? wv[u] : make_uint4(0, 0, 0, 0);
      if (p.drop_thresh24 && wopix[u] >= 0) {
        float f[UE];
        unpack16<T>(v, f);
        const uint32_t base = (uint32_t)(((size_t)n * nopix + wopix[u]) * p.C + c0 + (i % CVS) * UE);
#pragma unroll
        for (int e = 0; e < UE; ++e) f[e] = tfc_keep(p.seed, base + e, p.drop_thresh24) ? f[e] * p.drop_scale : 0.f;
        v = pack16<T>(f);
      }
      win[i] = v;
    }
  }
  float mean[UE], rstd[UE], mg[UE], mgx[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { mean[e] = 0.f; rstd[e] = 1.f; mg[e] = 0.f; mgx[e] = 0.f; }
  if (NORM) {
    const float inv = 1.f / (float)npix;
#pragma unroll
    for (int e = 0; e < UE; ++e) {
      const size_t si = ((size_t)n * p.C + cvg * UE + e) * 2;
      const float m = stats[si] * inv;
      mean[e] = m;
      rstd[e] = rsqrtf(fmaxf(stats[si + 1] * inv - m * m, 0.f) + p.eps);
      if (MODE == 2) { mg[e] = rstats[si] * inv; mgx[e] = rstats[si + 1] * inv; }
    }
  }
  __syncthreads();
  float a1[UE], a2[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  auto process = [&](int y, int xq, const uint4& xraw) -> uint4 {
    float g[UE];
#pragma unroll
    for (int e = 0; e < UE; ++e) g[e] = 0.f;
    const float4 wr = wrow[y - ty0], wc = wcol[xq - tx0];
    const int wy = ((y + 1) >> 1) - oyb, wx = ((xq + 1) >> 1) - oxb;   // window coords of the (oy0, ox0) tap
    auto tap = [&](int dy, int dx, float w) {
      float v[UE];
      unpack16<T>(win[((wy + dy) * WW + wx + dx) * CVS + cvl], v);
#pragma unroll
      for (int e = 0; e < UE; ++e) g[e] += w * v[e];
    };
    tap(-1, -1, wr.x * wc.x); tap(-1, 0, wr.x * wc.y); tap(0, -1, wr.y * wc.x); tap(0, 0, wr.y * wc.y);
    if (wr.z != 0.f || wc.z != 0.f) {                              // only next to the bottom / right reflect border
      tap(-1, 1, wr.x * wc.z); tap(0, 1, wr.y * wc.z);
      tap(1, -1, wr.z * wc.x); tap(1, 0, wr.z * wc.y); tap(1, 1, wr.z * wc.z);
    }
    float xv[UE], xh[UE];
    if (SIGNW) {
#pragma unroll
      for (int e = 0; e < UE; ++e) { xh[e] = 0.f; g[e] = ((xraw.x >> e) & 1u) ? g[e] : g[e] * p.slope; }
    } else {
      unpack16<T>(xraw, xv);
#pragma unroll
      for (int e = 0; e < UE; ++e) {
        xh[e] = NORM ? (xv[e] - mean[e]) * rstd[e] : xv[e];
        g[e] = xh[e] > 0.f ? g[e] : g[e] * p.slope;
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < UE; ++e) { a1[e] += g[e]; a2[e] += g[e] * xh[e]; }
      return make_uint4(0, 0, 0, 0);
    }
    if (MODE == 0 && rstats) {
#pragma unroll
      for (int e = 0; e < UE; ++e) a1[e] += g[e];
    }
    float r[UE];
#pragma unroll
    for (int e = 0; e < UE; ++e) r[e] = (MODE == 2) ? rstd[e] * (g[e] - mg[e] - xh[e] * mgx[e]) : g[e];
    return pack16<T>(r);
  };
  // four pixels per iteration with their x loads issued up front: one dependent load per iteration would leave a single
  // 16-byte request in flight per lane and starve the memory pipeline
  for (int tp0 = pl; tp0 < TH * TW; tp0 += 4 * PL) {
    uint4 xr[4];
    int yy[4], xx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tp = tp0 + u * PL;
      yy[u] = ty0 + tp / TW; xx[u] = tx0 + tp % TW;
      const bool ok = tp < TH * TW && yy[u] < p.H && xx[u] < p.W;
      if (!ok) yy[u] = -1;
      if (SIGNW) xr[u] = make_uint4(ok ? sgn[(size_t)(yy[u] * p.W + xx[u]) * 8 + cvg] : 0u, 0, 0, 0);
      else xr[u] = ok ? *reinterpret_cast<const uint4*>(xn + (size_t)(yy[u] * p.W + xx[u]) * p.x_pitch + cvg * UE) : make_uint4(0, 0, 0, 0);
    }
    // the compute phase touches LDS only; the four results stay in four separate register quads and are stored together at the
    // end: on this target a store's data VGPRs may not be rewritten before the store completes (hipcc waits vmcnt for it)
    uint4 res[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) res[u] = yy[u] >= 0 ? process(yy[u], xx[u], xr[u]) : make_uint4(0, 0, 0, 0);
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (yy[u] >= 0) *reinterpret_cast<uint4*>(dxn + (size_t)(yy[u] * p.W + xx[u]) * dx_pitch + cvg * UE) = res[u];
    }
  }
  if (MODE == 1 || (MODE == 0 && rstats)) {
    __syncthreads();                                               // the window is dead: its LDS becomes the reduction scratch
#pragma unroll
    for (int e = 0; e < UE; ++e) { red[threadIdx.x * UE + e] = a1[e]; red[256 * UE + threadIdx.x * UE + e] = a2[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < cslice; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      const int ccv = c / UE, ce = c % UE;
      for (int q = 0; q < PL; ++q) { s1 += red[(q * CVS + ccv) * UE + ce]; s2 += red[256 * UE + (q * CVS + ccv) * UE + ce]; }
      const size_t slot = ((size_t)n * gridDim.x + blockIdx.x) * p.C + c0 + c;   // rstats = part[n][tile][C]([2]): summed in a fixed order by tfc_part_reduce_kernel
      if (MODE == 1) reinterpret_cast<float2*>(rstats)[slot] = make_float2(s1, s2);
      else rstats[slot] = s1;
    }
  }
}

// Stride-1 blur of the up path (UNetUp: ConvTranspose2d -> BlurPool(stride 1) -> InstanceNorm, reference :118-134), forward and
// transpose in one tiled kernel: no activation sits in front of this blur, so it is a pure depthwise [1,3,3,1]^2/64 filter with
// reflect padding (1,2). A workgroup owns an 8 x 32 tile x (8 channel vectors): the (12 x 36)-pixel source window is staged once
// in LDS (forward: reflect indexing at staging time; transpose: zero fill), and each thread runs the separable filter down one
// column -- a horizontal pass per window row kept in a 5-deep register ring, then the vertical pass. Tap weights are tabulated
// per tile row / column at offsets -2..+2 (forward uses -1..+2; the transpose -2..+1, plus +2 through the reflect aliases of the
// bottom / right border), so border and interior pixels run the same code. STATS: InstanceNorm sums of the result.
template <typename T, bool STATS>
__global__ void __launch_bounds__(256)
tfc_blur1_kernel(const ActParams p, const T* __restrict__ src, int src_pitch, T* __restrict__ dst, int dst_pitch, float* stats_out,
                 int transpose, int tiles_x) {
  constexpr int UE = ElemTraits<T>::UE;
  constexpr int TH = 8, TW = 32, WH = TH + 4, WW = TW + 4, CVS = 8, PL = 256 / CVS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* win = reinterpret_cast<uint4*>(smem_raw);                // [WH][WW][CVS]
  float* wrow = reinterpret_cast<float*>(smem_raw + (size_t)WH * WW * CVS * 16);   // [TH][8]
  float* wcol = wrow + TH * 8;                                    // [TW][8]
  float* red = reinterpret_cast<float*>(smem_raw);                // [2][256*UE] after the main loop
  const int cvl = threadIdx.x % CVS, pl = threadIdx.x / CVS;
  const int n = blockIdx.y;
  const int c0 = blockIdx.z * (CVS * UE);
  const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
  const T* sn = src + (size_t)n * p.H * p.W * src_pitch + c0;
  T* dn = dst + (size_t)n * p.H * p.W * dst_pitch + c0;
  if (threadIdx.x < TH + TW) {
    const bool isrow = threadIdx.x < TH;
    const int q = isrow ? ty0 + threadIdx.x : tx0 + (threadIdx.x - TH);
    const int L = isrow ? p.H : p.W;
    float w5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (q < L) {
      if (!transpose) {
        w5[1] = 0.125f; w5[2] = 0.375f; w5[3] = 0.375f; w5[4] = 0.125f;
      } else {
        for (int a = 0; a < 4; ++a) {
          if ((a == 1 && q != 1) || (a == 2 && q != L - 2) || (a == 3 && q != L - 3)) continue;
          const int pq = a == 0 ? q : (a == 1 ? -1 : (a == 2 ? L : L + 1));
          for (int k = 0; k < 4; ++k) {
            const int o = pq + 1 - k;
            if (o < 0 || o >= L) continue;
            const int d = o - q + 2;
            for (int j = 0; j < 5; ++j) if (j == d) w5[j] += blur_w(k);
          }
        }
      }
    }
    float* t = isrow ? wrow + threadIdx.x * 8 : wcol + (threadIdx.x - TH) * 8;
    for (int j = 0; j < 5; ++j) t[j] = w5[j];
  }
  // window staging in batches of seven 16-byte units per thread, all loads of a batch requested (clamped address) before the first LDS store
  // (as in tfc_act_pool2_bwd_kernel: one dependent load per loop iteration left a single request in flight per lane)
  for (int b0 = 0; b0 < WH * WW * CVS; b0 += 7 * 256) {
    uint4 wv[7];
    bool wok[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int i = b0 + u * 256 + threadIdx.x;
      const int cv = i % CVS, wp = i / CVS;
      int y = ty0 - 2 + wp / WW, x = tx0 - 2 + wp % WW;
      if (!transpose) { y = reflect_idx(y, p.H); x = reflect_idx(x, p.W); }   // forward: reflect (1,2); positions beyond that carry zero weight
      wok[u] = i < WH * WW * CVS && y >= 0 && y < p.H && x >= 0 && x < p.W;
      wv[u] = *reinterpret_cast<const uint4*>(sn + (size_t)(wok[u] ? y * p.W + x : 0) * src_pitch + cv * UE);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int i = b0 + u * 256 + threadIdx.x;
      if (i < WH * WW * CVS) win[i] = wok[u] ? wv[u] : make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  float a1[UE], a2[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
#pragma unroll 1
  for (int lx = pl; lx < TW; lx += PL) {
    const float4 wc03 = *reinterpret_cast<const float4*>(wcol + lx * 8);
    const float wc4 = wcol[lx * 8 + 4];
    const float wc[5] = {wc03.x, wc03.y, wc03.z, wc03.w, wc4};
    const bool xin = tx0 + lx < p.W;
    float ring[5][UE];
#pragma unroll
    for (int r = 0; r < WH; ++r) {
      float h[UE];
#pragma unroll
      for (int e = 0; e < UE; ++e) h[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (wc[j] != 0.f) {
          float v[UE];
          unpack16<T>(win[(r * WW + lx + j) * CVS + cvl], v);
#pragma unroll
          for (int e = 0; e < UE; ++e) h[e] += wc[j] * v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < UE; ++e) ring[r % 5][e] = h[e];
      if (r >= 4) {
        const int ly = r - 4;
        const float4 wr03 = *reinterpret_cast<const float4*>(wrow + ly * 8);
        const float wr4 = wrow[ly * 8 + 4];
        float o[UE];
#pragma unroll
        for (int e = 0; e < UE; ++e)
          o[e] = wr03.x * ring[(ly + 0) % 5][e] + wr03.y * ring[(ly + 1) % 5][e] + wr03.z * ring[(ly + 2) % 5][e] +
                 wr03.w * ring[(ly + 3) % 5][e] + wr4 * ring[(ly + 4) % 5][e];
        if (xin && ty0 + ly < p.H) {
          const uint4 pk = pack16<T>(o);
          if (STATS) {                                             // InstanceNorm sums of the values actually STORED (bf16-rounded): the tensor
            float os[UE];                                          // nn.InstanceNorm2d normalises; sums of the fp32 results differ from it by the
            unpack16<T>(pk, os);                                   // rounding noise, which an 8 x 8 plane's backward amplifies to ~1e-2
#pragma unroll
            for (int e = 0; e < UE; ++e) { a1[e] += os[e]; a2[e] += os[e] * os[e]; }
          }
          *reinterpret_cast<uint4*>(dn + (size_t)((ty0 + ly) * p.W + tx0 + lx) * dst_pitch + cvl * UE) = pk;
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < UE; ++e) { red[threadIdx.x * UE + e] = a1[e]; red[256 * UE + threadIdx.x * UE + e] = a2[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < CVS * UE; c += 256) {
      float s1 = 0.f, s2 = 0.f;
      const int ccv = c / UE, ce = c % UE;
      for (int q = 0; q < PL; ++q) { s1 += red[(q * CVS + ccv) * UE + ce]; s2 += red[256 * UE + (q * CVS + ccv) * UE + ce]; }
      reinterpret_cast<float2*>(stats_out)[((size_t)n * gridDim.x + blockIdx.x) * p.C + c0 + c] = make_float2(s1, s2);   // part[n][tile][C][2]
    }
  }
}
template <typename T>
static bool blur1_launch(const ActParams& p, const void* src, int src_pitch, void* dst, int dst_pitch, float* stats_out, float* part_ws, int transpose,
                         hipStream_t st) {
  constexpr int UE = ElemTraits<T>::UE;
  if (p.C % (8 * UE) != 0 || p.H < 4 || p.W < 4) return false;
  const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + 7) / 8;
  const dim3 grid(tiles_x * tiles_y, p.N, p.C / (8 * UE));
  const size_t lds = (size_t)12 * 36 * 8 * 16 + (8 + 32) * 8 * sizeof(float);
  if (stats_out) {
    if (!part_ws || !part_fits((long long)grid.x * p.N * p.C * 2)) return false;   // the generic kernel's launcher reports the error
    hipLaunchKernelGGL((tfc_blur1_kernel<T, true>), grid, dim3(256), lds, st, p, (const T*)src, src_pitch, (T*)dst, dst_pitch, part_ws, transpose, tiles_x);
    tfc_launch_part_reduce(part_ws, stats_out, p.N, grid.x, 2 * p.C, st);
  } else
    hipLaunchKernelGGL((tfc_blur1_kernel<T, false>), grid, dim3(256), lds, st, p, (const T*)src, src_pitch, (T*)dst, dst_pitch, stats_out, transpose, tiles_x);
  return true;
}

// column sums: out[c] += sum over rows of x[row][c]   (bias gradients). Deterministic: a workgroup owns 8 channel vectors, its 32 row-lanes take rows
// r, r + 32, ... ascending and meet in LDS in lane order; with many rows (and a partial buffer) the rows are first split over blockIdx.y and the
// slots summed by tfc_part_reduce_kernel.
template <typename T>
__global__ void __launch_bounds__(256)
tfc_colsum_kernel(const T* __restrict__ x, long long rows, int pitch, int C, float* out, int to_part) {
  constexpr int UE = ElemTraits<T>::UE;
  __shared__ float red[32][8 * UE];
  const int CV = C / UE;
  const int cvl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int cv = blockIdx.x * 8 + cvl;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = (long long)blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float a[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) a[e] = 0.f;
  if (cv < CV)
    for (long long r = r0 + rl; r < r1; r += 32) {
      float v[UE];
      unpack16<T>(*reinterpret_cast<const uint4*>(x + r * pitch + cv * UE), v);
#pragma unroll
      for (int e = 0; e < UE; ++e) a[e] += v[e];
    }
#pragma unroll
  for (int e = 0; e < UE; ++e) red[rl][cvl * UE + e] = a[e];
  __syncthreads();
  if (threadIdx.x < 8 * UE) {
    const int c = blockIdx.x * 8 * UE + threadIdx.x;
    if (c < C) {
      float s = 0.f;
#pragma unroll 8
      for (int q = 0; q < 32; ++q) s += red[q][threadIdx.x];
      if (to_part) out[(size_t)blockIdx.y * C + c] = s;           // out = part[blockIdx.y][C]
      else out[c] += s;
    }
  }
}

// NCHW fp32 (a: Ca channels, b: Cb channels) -> NHWC with 8 channels (zero padded)
template <typename T>
__global__ void __launch_bounds__(256)
tfc_pack_nhwc8_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb, T* __restrict__ out,
                      int N, int HW) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * HW) return;
  const int n = (int)(idx / HW), pix = (int)(idx - (long long)n * HW);
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float t = 0.f;
    if (c < Ca) t = a[((size_t)n * Ca + c) * HW + pix];
    else if (c < Ca + Cb) t = b[((size_t)n * Cb + (c - Ca)) * HW + pix];
    v[c] = t;
  }
  T* o = out + idx * 8;
  if (sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(o) = pack16<T>(v);
  } else {
    *reinterpret_cast<uint4*>(o) = pack16<T>(v);
    *reinterpret_cast<uint4*>(o + 4) = pack16<T>(v + 4);
  }
}

// NHWC (pitch) channels [c0, c0+C) -> NCHW fp32: out = alpha * in + beta * out
template <typename T>
__global__ void __launch_bounds__(256)
tfc_unpack_nchw_kernel(const T* __restrict__ in, int pitch, int c0, int C, float* __restrict__ out, int N, int HW,
                       float alpha, float beta) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * HW) return;
  const int n = (int)(idx / HW), pix = (int)(idx - (long long)n * HW);
  for (int c = 0; c < C; ++c) {
    const float v = ElemTraits<T>::ld(in + idx * pitch + c0 + c) * alpha;
    float* o = out + ((size_t)n * C + c) * HW + pix;
    *o = beta != 0.f ? v + beta * (*o) : v;
  }
}

// generator head backward: dyraw[n,pix,c<3] = g * (1 - y^2)  (NHWC8), bias grad dbias[c] += sum
template <typename T>
__global__ void __launch_bounds__(256)
tfc_tanh_bwd_pack_kernel(const float* __restrict__ g, const float* __restrict__ y, T* __restrict__ out, float* dbias,
                         int N, int C, int HW) {
  __shared__ float red[4][256];
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  const long long total = (long long)N * HW;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
    const int n = (int)(idx / HW), pix = (int)(idx - (long long)n * HW);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) {
        const size_t o = ((size_t)n * C + c) * HW + pix;
        const float yy = y[o];
        v[c] = g[o] * (1.f - yy * yy);
        bs[c] += v[c];
      }
    T* o = out + idx * 8;
    *reinterpret_cast<uint4*>(o) = pack16<T>(v);
    if (sizeof(T) == 4) *reinterpret_cast<uint4*>(o + 4) = pack16<T>(v + 4);
  }
  if (dbias) {                                                   // dbias = part[blockIdx.x][4]: summed in a fixed order by tfc_part_reduce_kernel
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = bs[c];
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float sred = red[c][threadIdx.x] + red[c][threadIdx.x + 64] + red[c][threadIdx.x + 128] + red[c][threadIdx.x + 192];
        sred = wave_sum(sred);
        if (threadIdx.x == 0) dbias[(size_t)blockIdx.x * 4 + c] = c < C ? sred : 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// spectral norm (torch.nn.utils.parametrizations.spectral_norm, 1 power iteration per training forward):
//   u <- normalize(W v);  v <- normalize(W^T u);  sigma = u . (W v)        W: [R][K] fp32 row-major
// ---------------------------------------------------------------------------------------------------
// ---- batched form: all spectral-normed layers of one Discriminator1 forward in 3 launches (blockIdx.y = layer) ----
struct SnBatch {
  const float* W[4];
  float* u[4]; float* v[4]; float* sigma2[4];
  float* us[4]; float* vs[4];                                    // per-call snapshots of u, v for the backward (nullable)
  float* s[4]; float* t[4];                                      // scratch
  int R[4], K[4];
  int n;
};

__device__ __forceinline__ float block_sum_256(float a, float* red4) {
  a = wave_sum(a);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = a;
  __syncthreads();
  return red4[0] + red4[1] + red4[2] + red4[3];
}
// s = W v (one wave per row)
__global__ void __launch_bounds__(256) tfc_snb_mv_kernel(const SnBatch b) {
  const int L = blockIdx.y, R = b.R[L], K = b.K[L];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* W = b.W[L] + (size_t)row * K;
  const float* v = b.v[L];
  float a = 0.f;
  if ((K & 3) == 0) {                                            // 16-byte loads (K = Cin * 16; every row start is 16-byte aligned)
    const float4* W4 = reinterpret_cast<const float4*>(W);
    const float4* v4 = reinterpret_cast<const float4*>(v);
#pragma unroll 8                                                   // eight row fragments in flight per lane (the adds stay in k order)
    for (int k = lane; k < (K >> 2); k += 64) { const float4 w = W4[k], x = v4[k]; a += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w; }
  } else {
    for (int k = lane; k < K; k += 64) a += W[k] * v[k];
  }
  a = wave_sum(a);
  if (lane == 0) b.s[L][row] = a;
}
// u = s / max(|s|, eps) (norm recomputed per workgroup: R <= 4096); t_part[z] = W^T u over this workgroup's 32 rows (z = blockIdx.z) -- plain stores,
// summed over z in a fixed order by tfc_snb_vnorm_kernel (the float atomics this replaces made sigma, and with it every discriminator activation,
// differ from run to run)
__global__ void __launch_bounds__(256) tfc_snb_mtv_kernel(const SnBatch b, float eps) {
  __shared__ float red4[4];
  __shared__ float ush[32];
  const int L = blockIdx.y, R = b.R[L], K = b.K[L];
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.z * 32;
  if (blockIdx.x * 256 >= K || r0 >= R) return;                  // workgroup-uniform
  float a = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) { const float x = b.s[L][i]; a += x * x; }
  const float inv = 1.f / fmaxf(sqrtf(block_sum_256(a, red4)), eps);
  if (threadIdx.x < 32) {
    const int r = r0 + threadIdx.x;
    const float uv = r < R ? b.s[L][r] * inv : 0.f;
    ush[threadIdx.x] = uv;
    if (blockIdx.x == 0 && r < R) { b.u[L][r] = uv; if (b.us[L]) b.us[L][r] = uv; }
  }
  __syncthreads();
  if (k >= K) return;
  const float* W = b.W[L];
  const int r1 = min(R, r0 + 32);
  float acc = 0.f;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) acc += W[(size_t)r * K + k] * ush[r - r0];
  b.t[L][(size_t)blockIdx.z * K + k] = acc;
}
// one workgroup per layer: t = sum_z t_part[z] (z ascending), v = t / max(|t|, eps), sigma = t . v  (= u . (W v) with the u, v just computed:
// u^T W v = (W^T u)^T v; torch evaluates the left form, the two agree to fp32 round-off and this one needs no third pass over W)
__global__ void __launch_bounds__(1024) tfc_snb_vnorm_kernel(const SnBatch b, float eps) {
  __shared__ float red[16];
  const int L = blockIdx.x, R = b.R[L], K = b.K[L];
  const int nz = (R + 31) / 32;
  const float* tp = b.t[L];
  // this thread's columns k = tid, tid + 1024, ...: the first four stay in registers (K <= 4096 on the path), so the partials are read once; the z loop
  // is unrolled so that its loads are in flight together (the adds stay in z order)
  float tv[4] = {0.f, 0.f, 0.f, 0.f};
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = threadIdx.x + i * 1024;
    if (k < K) {
      float t = 0.f;
#pragma unroll 8
      for (int z = 0; z < nz; ++z) t += tp[(size_t)z * K + k];
      tv[i] = t;
      a += t * t;
    }
  }
  for (int k = threadIdx.x + 4096; k < K; k += 1024) {
    float t = 0.f;
    for (int z = 0; z < nz; ++z) t += tp[(size_t)z * K + k];
    a += t * t;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) tot += red[i];
  const float inv = 1.f / fmaxf(sqrtf(tot), eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = threadIdx.x + i * 1024;
    if (k < K) {
      const float x = tv[i] * inv;
      b.v[L][k] = x;
      if (b.vs[L]) b.vs[L][k] = x;
    }
  }
  for (int k = threadIdx.x + 4096; k < K; k += 1024) {
    float t = 0.f;
    for (int z = 0; z < nz; ++z) t += tp[(size_t)z * K + k];
    const float x = t * inv;
    b.v[L][k] = x;
    if (b.vs[L]) b.vs[L][k] = x;
  }
  if (threadIdx.x == 0) {
    const float sg = tot * inv;
    b.sigma2[L][0] = sg;
    b.sigma2[L][1] = 1.f / sg;
  }
}
// no power iteration (eval-mode forward): sigma = u . s with s = W v from tfc_snb_mv_kernel, summed in a fixed order; snapshots of the stored u, v
__global__ void __launch_bounds__(1024) tfc_snb_sigma_kernel(const SnBatch b) {
  __shared__ float red[16];
  const int L = blockIdx.x, R = b.R[L], K = b.K[L];
  float a = 0.f;
  for (int i = threadIdx.x; i < R; i += 1024) a += b.u[L][i] * b.s[L][i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];
    b.sigma2[L][0] = tot;
    b.sigma2[L][1] = 1.f / tot;
  }
  for (int i = threadIdx.x; i < K; i += 1024) if (b.vs[L]) b.vs[L][i] = b.v[L][i];
  for (int i = threadIdx.x; i < R; i += 1024) if (b.us[L]) b.us[L][i] = b.u[L][i];
}

hipError_t tfc_launch_sn_step_batched(const SnBatch& b, int power_iter, float eps, hipStream_t st) {
  int maxR = 0, maxK = 0;
  for (int i = 0; i < b.n; ++i) { maxR = b.R[i] > maxR ? b.R[i] : maxR; maxK = b.K[i] > maxK ? b.K[i] : maxK; }
  hipLaunchKernelGGL(tfc_snb_mv_kernel, dim3((maxR + 3) / 4, b.n), dim3(256), 0, st, b);
  if (power_iter) {
    hipLaunchKernelGGL(tfc_snb_mtv_kernel, dim3((maxK + 255) / 256, b.n, (maxR + 31) / 32), dim3(256), 0, st, b, eps);
    hipLaunchKernelGGL(tfc_snb_vnorm_kernel, dim3(b.n), dim3(1024), 0, st, b, eps);
  } else {
    hipLaunchKernelGGL(tfc_snb_sigma_kernel, dim3(b.n), dim3(1024), 0, st, b);
  }
  return hipGetLastError();
}

// spectral-norm backward: gw = (G - (sum G*Wsn) u v^T) / sigma, Wsn = W/sigma.  pass 1: dot += sum G*W ; pass 2: apply
// dot = sum a*b: per-workgroup partials (doubles) to ws, then ONE workgroup adds them in slot order (tfc_dot_fin_kernel) -- the scalar feeds every
// element of the weight gradient, so it must not depend on the order in which workgroups finish
__global__ void __launch_bounds__(256) tfc_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, double* part) {
  __shared__ float red[4];
  float s = 0.f;
  if ((n & 3) == 0) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += (long long)gridDim.x * 256) {
      const float4 x = a4[i], y = b4[i];
      s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += a[i] * b[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
}
__global__ void __launch_bounds__(256) tfc_sn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ u, const float* __restrict__ v,
                                                                const float* __restrict__ sigma2, const double* __restrict__ dot_part, int nparts,
                                                                float* gout, int R, int K, int accumulate) {
  __shared__ float dsh;
  if (threadIdx.x < 64) {                                        // the dot product: <= 128 partials, two per lane, then a fixed butterfly
    double d = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) d += dot_part[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (threadIdx.x == 0) dsh = (float)d;
  }
  __syncthreads();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)R * K) return;
  const int r = (int)(i / K), k = (int)(i - (long long)r * K);
  const float inv = sigma2[1];
  const float dotws = dsh * inv;                                 // sum G * Wsn = (sum G*W)/sigma
  const float val = (G[i] - dotws * u[r] * v[k]) * inv;
  gout[i] = accumulate ? gout[i] + val : val;
}

// ---------------------------------------------------------------------------------------------------
// relativistic BCE-with-logits (reference :554, :628-630).  x = a - b
//   mode 0 (G): loss = mean BCE(x, t1);                da = dL/dx
//   mode 1 (D): loss = 0.5*[mean BCE(x,t1) + mean BCE(-x,t2)];  da = dL/dx, db = -dL/dx
// ---------------------------------------------------------------------------------------------------
static __device__ TfcRedSlot g_bce_slot;
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
template <typename T>
__global__ void __launch_bounds__(256)
tfc_bce_rel_kernel(const T* __restrict__ a, const T* __restrict__ b, int n, int stride, float t1, float t2, int mode, float* loss,
                   T* da, T* db, float gscale) {
  __shared__ float red[4];
  float l = 0.f;
  const float invn = 1.f / (float)n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float x = ElemTraits<T>::ld(a + (size_t)i * stride) - ElemTraits<T>::ld(b + (size_t)i * stride);
    float gx;
    if (mode == 0) {
      l += bce_logits(x, t1);
      gx = (sigmoidf_(x) - t1) * invn;
    } else {
      l += 0.5f * (bce_logits(x, t1) + bce_logits(-x, t2));
      gx = 0.5f * ((sigmoidf_(x) - t1) - (sigmoidf_(-x) - t2)) * invn;
    }
    // the logit lives in channel 0 of an 8-channel pixel (pitch `stride`) whose other channels the consumer (the head's input-gradient GEMM)
    // reads as zeros: write the whole pixel, so the caller need not zero the buffer first (three fill launches per step)
    auto put = [&](T* dst, float v) {
      constexpr int UE = ElemTraits<T>::UE;
      if (stride == 8 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        float z[8] = {v, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<uint4*>(dst) = pack16<T>(z);
        if (UE == 4) *reinterpret_cast<uint4*>(dst + 4) = pack16<T>(z + 4);
      } else {
        ElemTraits<T>::st(dst, v);
      }
    };
    if (da) put(da + (size_t)i * stride, gx * gscale);
    if (db) put(db + (size_t)i * stride, -gx * gscale);
  }
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) tfc_block_commit(&g_bce_slot, ((double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]) / (double)n, loss, true);
}

// Adam (torch.optim.Adam defaults: no weight decay, no amsgrad; reference :461-462)
__global__ void __launch_bounds__(256)
tfc_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

__global__ void __launch_bounds__(256)
tfc_axpby_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ y, long long n, float a, float b) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

__global__ void __launch_bounds__(256)
tfc_dropout_mask_kernel(unsigned char* __restrict__ out, long long n, unsigned seed, unsigned thresh24) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = tfc_keep(seed, (uint32_t)i, thresh24) ? 1 : 0;
}

// fp32 <-> compute dtype casts of flat buffers (test plumbing and fft inputs)
template <typename T>
__global__ void __launch_bounds__(256) tfc_cast_from_f32_kernel(const float* __restrict__ x, T* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) ElemTraits<T>::st(y + i, x[i]);
}
template <typename T>
__global__ void __launch_bounds__(256) tfc_cast_to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = ElemTraits<T>::ld(x + i);
}

// ---------------------------------------------------------------------------------------------------
// PatchGAN head forward: ZeroPad2d((1,0,1,0)) + Conv2d(C -> 1, k4, p1, no bias)   (reference :201-202).
// One output channel is a poor fit for an MFMA tile (N = 1 of 32) and, at 16x16 outputs, for the tile grid (64 workgroups
// x 512 serial k-steps): here a wave owns one output pixel, lanes split the C channels (16 B each), the 16 taps are walked
// with the filter held transposed in LDS, and a wave shuffle finishes the dot product.
// y[n][oy][ox] = sum_{ky,kx,c} x[n][oy+ky-2][ox+kx-2][c] * w[c][ky][kx]
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
tfc_head_fwd_kernel(const T* __restrict__ x, int x_pitch, const float* __restrict__ w, T* __restrict__ y, int y_pitch,
                    int N, int H, int W, int C) {
  constexpr int UE = ElemTraits<T>::UE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int npix = N * H * W;
  const int CV = C / UE;                                          // 16-byte units per pixel
  if (CV <= 64) {
    // one unit per lane: the lane keeps ITS channels' 16 filter taps in registers for the whole launch (torch layout [c][16]: UE * 16
    // contiguous floats per lane, streamed once) -- no LDS, no per-workgroup transpose of the 32 KB filter
    float wr[UE][16];
    const bool act = lane < CV;
#pragma unroll
    for (int e = 0; e < UE; ++e)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 f = act ? *reinterpret_cast<const float4*>(w + ((size_t)(lane * UE + e)) * 16 + q4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        wr[e][q4 * 4 + 0] = f.x; wr[e][q4 * 4 + 1] = f.y; wr[e][q4 * 4 + 2] = f.z; wr[e][q4 * 4 + 3] = f.w;
      }
    for (int pix = blockIdx.x * 4 + wave; pix < npix; pix += gridDim.x * 4) {
      const int n = pix / (H * W), rem = pix - n * H * W;
      const int oy = rem / W, ox = rem - oy * W;
      float acc = 0.f;
      // all 16 taps are requested before the first is used (clamped address; a tap outside the plane is skipped below, wave-uniformly): behind a
      // per-tap bounds branch the loads went out one at a time and the kernel ran at 16 memory latencies per pixel (36.7 -> see DESIGN 3.1)
      uint4 xv[16];
      const int lc = act ? lane : 0;                               // idle lanes (C < 64 units) read lane 0's unit; their weights are zero
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        int iy = oy + (t >> 2) - 2, ix = ox + (t & 3) - 2;
        iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        xv[t] = *reinterpret_cast<const uint4*>(x + ((size_t)(n * H + iy) * W + ix) * x_pitch + lc * UE);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int iy = oy + (t >> 2) - 2, ix = ox + (t & 3) - 2;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;      // wave-uniform
        float v[UE];
        unpack16<T>(xv[t], v);
#pragma unroll
        for (int e = 0; e < UE; ++e) acc += v[e] * wr[e][t];
      }
      acc = wave_sum(acc);
      if (lane == 0) ElemTraits<T>::st(y + (size_t)pix * y_pitch, acc);
    }
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [16 taps][C]  (wide inputs: filter transposed in LDS)
  for (int i = threadIdx.x; i < 16 * C; i += 256) {
    const int c = i >> 4, t = i & 15;                             // torch layout index c*16 + t
    wl[t * C + c] = w[i];
  }
  __syncthreads();
  for (int pix = blockIdx.x * 4 + wave; pix < npix; pix += gridDim.x * 4) {
    const int n = pix / (H * W), rem = pix - n * H * W;
    const int oy = rem / W, ox = rem - oy * W;
    float acc = 0.f;
    for (int t = 0; t < 16; ++t) {
      const int iy = oy + (t >> 2) - 2, ix = ox + (t & 3) - 2;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;       // wave-uniform
      const T* px = x + ((size_t)(n * H + iy) * W + ix) * x_pitch;
      for (int u = lane; u < CV; u += 64) {
        float v[UE];
        unpack16<T>(*reinterpret_cast<const uint4*>(px + u * UE), v);
        const float* wt = wl + t * C + u * UE;
#pragma unroll
        for (int e = 0; e < UE; ++e) acc += v[e] * wt[e];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) ElemTraits<T>::st(y + (size_t)pix * y_pitch, acc);
  }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline dim3 act_grid(int npix, int C, int ue, int N) {
  const int cv = C / ue;
  const int ppb = 256 / cv;
  int nb = (npix + ppb - 1) / ppb;
#ifndef TFC_ACT_CAP
#define TFC_ACT_CAP 4096                                           // ~16 grid-stride workgroups per CU; 8192 and more measured 0.4-0.6 % slower per step
#endif
  int cap = TFC_ACT_CAP / (N > 0 ? N : 1);
  if (cap < 1) cap = 1;
  if (nb > cap) nb = cap;
  return dim3(nb, N);
}

template <typename T>
static hipError_t act_fwd_t(const ActParams& p, const void* x, const float* stats, void* out, float* stats_out, float* part_ws, hipStream_t st) {
  if (p.pool == 2 && !stats_out) {
    const dim3 g2 = act_grid(((p.Ho + 1) / 2) * ((p.Wo + 1) / 2), p.C, ElemTraits<T>::UE, p.N);
    if (!p.norm && p.slope == 1.f)
      hipLaunchKernelGGL((tfc_act_pool2_fwd_kernel<T, false>), g2, dim3(256), 0, st, p, (const T*)x, stats, (T*)out);
    else
      hipLaunchKernelGGL((tfc_act_pool2_fwd_kernel<T, true>), g2, dim3(256), 0, st, p, (const T*)x, stats, (T*)out);
    return hipGetLastError();
  }
  if (p.pool == 1 && !p.norm && p.slope == 1.f && !p.drop_thresh24 &&
      blur1_launch<T>(p, x, p.x_pitch, out, p.o_pitch, stats_out, part_ws, 0, st))        // pure blur of the up path: tiled kernel
    return hipGetLastError();
  const dim3 grid = act_grid(p.Ho * p.Wo, p.C, ElemTraits<T>::UE, p.N);
  if (stats_out) {
    if (!part_ws || !part_fits((long long)grid.x * p.N * p.C * 2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((tfc_act_fwd_kernel<T, true>), grid, dim3(256), 0, st, p, (const T*)x, stats, (T*)out, part_ws);
    return tfc_launch_part_reduce(part_ws, stats_out, p.N, grid.x, 2 * p.C, st);
  } else
    hipLaunchKernelGGL((tfc_act_fwd_kernel<T, false>), grid, dim3(256), 0, st, p, (const T*)x, stats, (T*)out, stats_out);
  return hipGetLastError();
}
hipError_t tfc_launch_act_fwd(int dt, const ActParams& p, const void* x, const float* stats, void* out, float* stats_out, float* part_ws, hipStream_t st) {
  return dt == TFC_DT_BF16 ? act_fwd_t<bf16_t>(p, x, stats, out, stats_out, part_ws, st) : act_fwd_t<float>(p, x, stats, out, stats_out, part_ws, st);
}
template <typename T, int MODE>
static void act_bwd_launch(const dim3& grid, const ActParams& p, const void* dout, const void* x, const float* stats, float* rstats,
                           void* dx, int use_x, int dx_pitch, hipStream_t st) {
  if (p.pool == 0)
    hipLaunchKernelGGL((tfc_act_bwd_kernel<T, MODE, 0>), grid, dim3(256), 0, st, p, (const T*)dout, (const T*)x, stats, rstats, (T*)dx, use_x, dx_pitch);
  else if (p.pool == 1)
    hipLaunchKernelGGL((tfc_act_bwd_kernel<T, MODE, 1>), grid, dim3(256), 0, st, p, (const T*)dout, (const T*)x, stats, rstats, (T*)dx, use_x, dx_pitch);
  else
    hipLaunchKernelGGL((tfc_act_bwd_kernel<T, MODE, 2>), grid, dim3(256), 0, st, p, (const T*)dout, (const T*)x, stats, rstats, (T*)dx, use_x, dx_pitch);
}
template <typename T>
static hipError_t act_bwd_t(int mode, const ActParams& p, const void* dout, const void* x, const float* stats, float* rstats,
                            void* dx, int use_x, int dx_pitch, float* part_ws, hipStream_t st) {
  // reductions (mode 1: the two InstanceNorm-backward sums per (image, channel); mode 0 with rstats: per-image bias-gradient sums) leave each workgroup
  // as one slot of part_ws and are added in a fixed order behind the kernel
  const int red_l = mode == 1 ? 2 * p.C : ((mode == 0 && rstats) ? p.C : 0);
  if (red_l && !part_ws) return hipErrorInvalidValue;
  if (p.pool == 2 && use_x) {                                     // tiled stride-2 backward (16 x 32 input pixels x 64 channels per workgroup)
    constexpr int UE = ElemTraits<T>::UE;
    const int cslice = p.C < 64 ? p.C : 64;
    if (p.C % cslice == 0 && 256 % (cslice / UE) == 0) {
      const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + 15) / 16;
      const dim3 grid(tiles_x * tiles_y, p.N, p.C / cslice);
      size_t lds = (size_t)11 * 19 * (cslice / UE) * 16;        // pooled-gradient window, reused as reduction scratch ...
      if (lds < 2 * 256 * UE * sizeof(float)) lds = 2 * 256 * UE * sizeof(float);
      lds += (16 + 32) * 16;                                      // ... + the per-row / per-column tap-weight tables
      if (red_l && !part_fits((long long)grid.x * p.N * red_l)) return hipErrorInvalidValue;
      if (mode == 0)                                              // mode 0 is only valid without normalisation, modes 1/2 only with it
        hipLaunchKernelGGL((tfc_act_pool2_bwd_kernel<T, 0, false>), grid, dim3(256), lds, st, p, (const T*)dout, (const T*)x, stats, rstats ? part_ws : nullptr, (T*)dx, dx_pitch, tiles_x, cslice);
      else if (mode == 1)
        hipLaunchKernelGGL((tfc_act_pool2_bwd_kernel<T, 1, true>), grid, dim3(256), lds, st, p, (const T*)dout, (const T*)x, stats, part_ws, (T*)dx, dx_pitch, tiles_x, cslice);
      else
        hipLaunchKernelGGL((tfc_act_pool2_bwd_kernel<T, 2, true>), grid, dim3(256), lds, st, p, (const T*)dout, (const T*)x, stats, rstats, (T*)dx, dx_pitch, tiles_x, cslice);
      if (red_l) return tfc_launch_part_reduce(part_ws, rstats, p.N, grid.x, red_l, st);
      return hipGetLastError();
    }
  }
  if (p.pool == 1 && mode == 0 && !use_x && !rstats && !p.drop_thresh24 &&
      blur1_launch<T>(p, dout, p.o_pitch, dx, dx_pitch, nullptr, nullptr, 1, st))          // transpose of the pure blur
    return hipGetLastError();
  const dim3 grid = act_grid(p.H * p.W, p.C, ElemTraits<T>::UE, p.N);
  if (red_l && !part_fits((long long)grid.x * p.N * red_l)) return hipErrorInvalidValue;
  if (mode == 0) act_bwd_launch<T, 0>(grid, p, dout, x, stats, rstats ? part_ws : nullptr, dx, use_x, dx_pitch, st);
  else if (mode == 1) act_bwd_launch<T, 1>(grid, p, dout, x, stats, part_ws, dx, use_x, dx_pitch, st);
  else act_bwd_launch<T, 2>(grid, p, dout, x, stats, rstats, dx, use_x, dx_pitch, st);
  if (red_l) return tfc_launch_part_reduce(part_ws, rstats, p.N, grid.x, red_l, st);
  return hipGetLastError();
}
// mode 0, stride-2 BlurPool, 64 bf16 channels, LeakyReLU' from sign words (the conv output was never stored: tfc_first_block_fwd)
hipError_t tfc_launch_act_pool2_bwd_signs(const ActParams& p, const void* dout, const unsigned char* sign_mask, float* rstats, void* dx, int dx_pitch,
                                          float* part_ws, hipStream_t st) {
  if (p.C != 64 || p.pool != 2 || (rstats && !part_ws)) return hipErrorInvalidValue;
  const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + 15) / 16;
  const dim3 grid(tiles_x * tiles_y, p.N, 1);
  size_t lds = (size_t)11 * 19 * 8 * 16;
  if (lds < 2 * 256 * 8 * sizeof(float)) lds = 2 * 256 * 8 * sizeof(float);
  lds += (16 + 32) * 16;
  if (rstats && !part_fits((long long)grid.x * p.N * p.C)) return hipErrorInvalidValue;
  hipLaunchKernelGGL((tfc_act_pool2_bwd_kernel<bf16_t, 0, false, true>), grid, dim3(256), lds, st, p, (const bf16_t*)dout, (const bf16_t*)sign_mask, nullptr,
                     rstats ? part_ws : nullptr, (bf16_t*)dx, dx_pitch, tiles_x, 64);
  if (rstats) return tfc_launch_part_reduce(part_ws, rstats, p.N, grid.x, p.C, st);
  return hipGetLastError();
}
hipError_t tfc_launch_act_bwd(int dt, int mode, const ActParams& p, const void* dout, const void* x, const float* stats,
                              float* rstats, void* dx, int use_x, int dx_pitch, float* part_ws, hipStream_t st) {
  return dt == TFC_DT_BF16 ? act_bwd_t<bf16_t>(mode, p, dout, x, stats, rstats, dx, use_x, dx_pitch, part_ws, st)
                           : act_bwd_t<float>(mode, p, dout, x, stats, rstats, dx, use_x, dx_pitch, part_ws, st);
}
hipError_t tfc_launch_colsum(int dt, const void* x, long long rows, int pitch, int C, float* out, float* part_ws, hipStream_t st) {
  const int ue = dt == TFC_DT_BF16 ? 8 : 4;
  const int nbx = (C / ue + 7) / 8;
  int ny = 1;                                                     // row split: only when there are enough rows to matter and a partial buffer to split into
  if (part_ws && rows >= 4096) {
    ny = (int)((rows + 1023) / 1024);
    const int cap = 2048 / nbx > 1 ? 2048 / nbx : 1;
    if (ny > cap) ny = cap;
    if (!part_fits((long long)ny * C)) ny = 1;
  }
  float* dst = ny > 1 ? part_ws : out;
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_colsum_kernel<bf16_t>), dim3(nbx, ny), dim3(256), 0, st, (const bf16_t*)x, rows, pitch, C, dst, ny > 1);
  else hipLaunchKernelGGL((tfc_colsum_kernel<float>), dim3(nbx, ny), dim3(256), 0, st, (const float*)x, rows, pitch, C, dst, ny > 1);
  if (ny > 1) return tfc_launch_part_reduce(part_ws, out, 1, ny, C, st);
  return hipGetLastError();
}
hipError_t tfc_launch_pack_nhwc8(int dt, const float* a, int Ca, const float* b, int Cb, void* out, int N, int HW, hipStream_t st) {
  const long long tot = (long long)N * HW;
  const dim3 grid((unsigned)((tot + 255) / 256));
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_pack_nhwc8_kernel<bf16_t>), grid, dim3(256), 0, st, a, Ca, b, Cb, (bf16_t*)out, N, HW);
  else hipLaunchKernelGGL((tfc_pack_nhwc8_kernel<float>), grid, dim3(256), 0, st, a, Ca, b, Cb, (float*)out, N, HW);
  return hipGetLastError();
}
hipError_t tfc_launch_unpack_nchw(int dt, const void* in, int pitch, int c0, int C, float* out, int N, int HW, float alpha, float beta, hipStream_t st) {
  const long long tot = (long long)N * HW;
  const dim3 grid((unsigned)((tot + 255) / 256));
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_unpack_nchw_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, pitch, c0, C, out, N, HW, alpha, beta);
  else hipLaunchKernelGGL((tfc_unpack_nchw_kernel<float>), grid, dim3(256), 0, st, (const float*)in, pitch, c0, C, out, N, HW, alpha, beta);
  return hipGetLastError();
}
hipError_t tfc_launch_tanh_bwd_pack(int dt, const float* g, const float* y, void* out, float* dbias, float* part_ws, int N, int C, int HW, hipStream_t st) {
  const long long tot = (long long)N * HW;
  long long nbk = (tot + 255) / 256;
  if (nbk > 1024) nbk = 1024;
  const dim3 grid((unsigned)nbk);
  if (dbias && !part_ws) return hipErrorInvalidValue;
  float* part = dbias ? part_ws : nullptr;                        // part[workgroup][4]
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_tanh_bwd_pack_kernel<bf16_t>), grid, dim3(256), 0, st, g, y, (bf16_t*)out, part, N, C, HW);
  else hipLaunchKernelGGL((tfc_tanh_bwd_pack_kernel<float>), grid, dim3(256), 0, st, g, y, (float*)out, part, N, C, HW);
  if (dbias) {                                                    // dbias[c] += sum over the workgroup slots, c < C (slot stride 4)
    hipLaunchKernelGGL(tfc_part_reduce_strided_kernel, dim3(1), dim3(256), 0, st, part, dbias, (int)nbk, 4, C);
  }
  return hipGetLastError();
}
// gout = (G - (sum G*W)/sigma * u v^T)/sigma ; dot_ws: 2 * TFC_SN_BWD_PARTS floats of scratch (the dot product's per-workgroup partials, as doubles)
hipError_t tfc_launch_sn_bwd(const float* G, const float* W, const float* u, const float* v, const float* sigma2, float* dot_ws,
                             float* gout, int R, int K, int accumulate, hipStream_t st) {
  const long long n = (long long)R * K;
  int nb = (int)((n + 1023) / 1024);
  if (nb > TFC_SN_BWD_PARTS) nb = TFC_SN_BWD_PARTS;
  double* part = reinterpret_cast<double*>(dot_ws);
  hipLaunchKernelGGL(tfc_dot_kernel, dim3(nb), dim3(256), 0, st, G, W, n, part);
  hipLaunchKernelGGL(tfc_sn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, u, v, sigma2, part, nb, gout, R, K, accumulate);
  return hipGetLastError();
}
hipError_t tfc_launch_bce_rel(int dt, const void* a, const void* b, int n, int stride, float t1, float t2, int mode, float* loss, void* da, void* db, float gscale, hipStream_t st) {
  int nb = (n + 255) / 256;
  if (nb > 256) nb = 256;
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_bce_rel_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, n, stride, t1, t2, mode, loss, (bf16_t*)da, (bf16_t*)db, gscale);
  else hipLaunchKernelGGL((tfc_bce_rel_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)a, (const float*)b, n, stride, t1, t2, mode, loss, (float*)da, (float*)db, gscale);
  return hipGetLastError();
}
hipError_t tfc_launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                           float bc1, float bc2_sqrt, float gscale, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(tfc_adam_kernel, dim3((int)nb), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
  return hipGetLastError();
}
hipError_t tfc_launch_axpby(float* out, const float* x, const float* y, long long n, float a, float b, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(tfc_axpby_kernel, dim3((int)nb), dim3(256), 0, st, out, x, y, n, a, b);
  return hipGetLastError();
}
hipError_t tfc_launch_dropout_mask(unsigned char* out, long long n, unsigned seed, unsigned thresh24, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(tfc_dropout_mask_kernel, dim3((int)nb), dim3(256), 0, st, out, n, seed, thresh24);
  return hipGetLastError();
}
hipError_t tfc_launch_cast(int dt, int to_f32, const void* x, void* y, long long n, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (dt == TFC_DT_BF16) {
    if (to_f32) hipLaunchKernelGGL((tfc_cast_to_f32_kernel<bf16_t>), dim3((int)nb), dim3(256), 0, st, (const bf16_t*)x, (float*)y, n);
    else hipLaunchKernelGGL((tfc_cast_from_f32_kernel<bf16_t>), dim3((int)nb), dim3(256), 0, st, (const float*)x, (bf16_t*)y, n);
  } else {
    if (to_f32) hipLaunchKernelGGL((tfc_cast_to_f32_kernel<float>), dim3((int)nb), dim3(256), 0, st, (const float*)x, (float*)y, n);
    else hipLaunchKernelGGL((tfc_cast_from_f32_kernel<float>), dim3((int)nb), dim3(256), 0, st, (const float*)x, (float*)y, n);
  }
  return hipGetLastError();
}

#ifndef TFC_HEAD_NB
#define TFC_HEAD_NB 512
#endif
hipError_t tfc_launch_head_fwd(int dt, const void* x, int x_pitch, const float* w, void* y, int y_pitch, int N, int H, int W, int C, hipStream_t st) {
  const int npix = N * H * W;
  int nb = (npix + 3) / 4;
  if (nb > TFC_HEAD_NB) nb = TFC_HEAD_NB;                        // the filter is loaded once per wave: few, long-lived workgroups
  const int ue = dt == TFC_DT_BF16 ? 8 : 4;
  const size_t lds = (C / ue <= 64) ? 0 : (size_t)16 * C * sizeof(float);
  if (dt == TFC_DT_BF16) hipLaunchKernelGGL((tfc_head_fwd_kernel<bf16_t>), dim3(nb), dim3(256), lds, st, (const bf16_t*)x, x_pitch, w, (bf16_t*)y, y_pitch, N, H, W, C);
  else hipLaunchKernelGGL((tfc_head_fwd_kernel<float>), dim3(nb), dim3(256), lds, st, (const float*)x, x_pitch, w, (float*)y, y_pitch, N, H, W, C);
  return hipGetLastError();
}
