// Loss heads of the PATCH-16 training step (gfx950).
//   triplet16 : the 16-patch "contrastive" head  (reference TFCGAN_multigpu_patchFFT_16P.py:75, :558-583)
//   spectrum  : ToPILImage -> convert("L") -> np.fft.rfft2 -> fftshift -> |.|, atan2  (reference :271-319)
//   l1 mean   : nn.L1Loss over amplitude / phase arrays (reference :323-375)
#include "common.h"

// ---------------------------------------------------------------------------------------------------
// 16-patch triplet.  Patch k (0-based) = rows 64*(k/4).., cols 64*(k%4)..  (make_16_patches, reference :227-253;
// first flat NCHW index of patch k = 64*(k%4) + 16384*(k/4)).  F.triplet_margin_loss(margin 1, p 2, eps 1e-6):
//   d(x,y) = || x - y + eps ||_2 over the LAST dim (one 64-pixel patch row); loss_k = mean_{n,c,row} max(1 + d_ap - d_an, 0)
//   total = (1/16) sum_k loss_k.  One wave per patch row: lane = pixel, wave-shuffle reductions.
// grad wrt anchor (fake):  [hinge>0] * ((a-p+eps)/d_ap - (a-n+eps)/d_an) * scale
// ---------------------------------------------------------------------------------------------------
struct NegIdx { int r[16]; };
static __device__ TfcRedSlot g_trip_slot, g_l1_slot;

__global__ void __launch_bounds__(256)
tfc_triplet16_kernel(const float* __restrict__ fake, const float* __restrict__ real, const NegIdx neg, int N, int C,
                     float margin, float eps, float* loss, float* dfake, float gscale) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const long long nrows = (long long)N * C * 256 * 4;
  const float scale = 1.f / (16.f * (float)N * (float)C * 64.f);
  float lsum = 0.f;
  // four rows per iteration: their loads are issued together (one dependent load -> reduce -> store chain per row leaves the
  // memory pipeline idle most of the time)
  constexpr int R = 4;
  const long long stride = (long long)gridDim.x * 4;
  for (long long row0 = (long long)blockIdx.x * 4 + w; row0 < nrows; row0 += stride * R) {
    float a[R], p[R], ng[R];
    size_t ia[R];
    bool ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long row = row0 + r * stride;
      ok[r] = row < nrows;
      const long long rw = ok[r] ? row : row0;
      const int kx = (int)(rw & 3);
      const long long r2 = rw >> 2;
      const int y = (int)(r2 & 255);
      const long long nc = r2 >> 8;                              // n*C + c
      const int k = (y >> 6) * 4 + kx;
      const int rk = neg.r[k];
      const size_t plane = (size_t)nc * 65536;
      ia[r] = plane + (size_t)y * 256 + kx * 64 + lane;
      const size_t in_ = plane + (size_t)((rk >> 2) * 64 + (y & 63)) * 256 + (rk & 3) * 64 + lane;
      a[r] = fake[ia[r]]; p[r] = real[ia[r]]; ng[r] = real[in_];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!ok[r]) continue;                                      // wave-uniform
      const float dp = a[r] - p[r] + eps, dn = a[r] - ng[r] + eps;
      const float sp = wave_sum(dp * dp), sn = wave_sum(dn * dn);
      const float dap = sqrtf(sp), dan = sqrtf(sn);
      const float hinge = margin + dap - dan;
      if (hinge > 0.f) {
        lsum += hinge;                                           // identical on all lanes
        if (dfake) {
          const float g = (dap > 0.f ? dp / dap : 0.f) - (dan > 0.f ? dn / dan : 0.f);
          dfake[ia[r]] = g * scale * gscale;
        }
      } else if (dfake) {
        dfake[ia[r]] = 0.f;
      }
    }
  }
  if (lane == 0) red[w] = lsum;
  __syncthreads();
  if (threadIdx.x == 0)
    tfc_block_commit(&g_trip_slot, ((double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]) / (16.0 * N * C * 64.0), loss, true);
}

// ---------------------------------------------------------------------------------------------------
// Spectrum of an S x S window of an NCHW fp32 image in [-1,1]:
//   u8 = (uint8) trunc(x*255)  (wraps mod 256 exactly like tensor.mul(255).byte());  L = (19595 R + 38470 G + 7471 B + 32768) >> 16
//   F = rfft2(L)  (S x (S/2+1)),  amp = |F|, pha = atan2(Im, Re); optional fftshift of both axes on store.
// Direct DFT in LDS with an exact sincospi twiddle table: rows (real input) then columns. A workgroup owns one window
// and a group of KG output columns, so S=256 (GLO-16) fits LDS as well as S=64 (PATCH-16).
// The four self-conjugate bins have Im forced to +0 (numpy's pocketfft yields exact zeros there).
// window w -> (n = w / wins_per_img, k = w % wins_per_img), origin row (k / wins_x)*S, col (k % wins_x)*S.
// ---------------------------------------------------------------------------------------------------
template <int S, int KG>
__global__ void __launch_bounds__(256)
tfc_spectrum_kernel(const float* __restrict__ img, long long bs, long long cs, int rs, int C, int wins_x, int wins_per_img,
                    float* __restrict__ amp, float* __restrict__ pha, int shift) {
  constexpr int NB = S / 2 + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lum = smem;                                     // S*S bytes
  float* tw = reinterpret_cast<float*>(smem + S * S);            // 2*S floats (cos, sin)
  float* R = tw + 2 * S;                                         // S*KG*2 floats
  const int w = blockIdx.x;
  const int kx0 = blockIdx.y * KG;
  const int n = w / wins_per_img, k = w % wins_per_img;
  const int y0 = (k / wins_x) * S, x0 = (k % wins_x) * S;
  const float* base = img + (size_t)n * bs + (size_t)y0 * rs + x0;
  for (int i = threadIdx.x; i < S * S; i += 256) {
    const int y = i / S, x = i % S;
    int q[3];
    for (int c = 0; c < 3; ++c) {
      const float v = base[(size_t)(C == 1 ? 0 : c) * cs + (size_t)y * rs + x] * 255.f;
      q[c] = ((int)v) & 255;
    }
    lum[i] = (unsigned char)((19595 * q[0] + 38470 * q[1] + 7471 * q[2] + 32768) >> 16);
  }
  for (int i = threadIdx.x; i < S; i += 256) {
    float sn, cn;
    sincospif(2.f * (float)i / (float)S, &sn, &cn);
    tw[2 * i] = cn; tw[2 * i + 1] = sn;
  }
  __syncthreads();
  // rows: R[y][kk] = sum_x L[y][x] * exp(-2 pi i x kx / S)
  for (int o = threadIdx.x; o < S * KG; o += 256) {
    const int y = o / KG, kk = o % KG, kx = kx0 + kk;
    float re = 0.f, im = 0.f;
    if (kx < NB) {
      for (int x = 0; x < S; ++x) {
        const int t = (x * kx) & (S - 1);
        const float v = (float)lum[y * S + x];
        re += v * tw[2 * t];
        im -= v * tw[2 * t + 1];
      }
    }
    R[2 * o] = re; R[2 * o + 1] = im;
  }
  __syncthreads();
  // columns: F[ky][kx] = sum_y R[y][kx] * exp(-2 pi i y ky / S)
  for (int o = threadIdx.x; o < S * KG; o += 256) {
    const int ky = o / KG, kk = o % KG, kx = kx0 + kk;
    if (kx >= NB) continue;
    float re = 0.f, im = 0.f;
    for (int y = 0; y < S; ++y) {
      const int t = (y * ky) & (S - 1);
      const float c = tw[2 * t], s = tw[2 * t + 1];
      const float rr = R[2 * (y * KG + kk)], ri = R[2 * (y * KG + kk) + 1];
      re += rr * c + ri * s;                                     // (rr + i ri) * (c - i s)
      im += ri * c - rr * s;
    }
    if ((kx == 0 || kx == S / 2) && (ky == 0 || ky == S / 2)) im = 0.f;
    int oy = ky, ox = kx;
    if (shift) { oy = (ky + S / 2) % S; ox = (kx + NB / 2) % NB; }
    const size_t oi = ((size_t)w * S + oy) * NB + ox;
    amp[oi] = sqrtf(re * re + im * im);
    pha[oi] = atan2f(im, re);
  }
}

// out[0] += scale * sum |a - b|
__global__ void __launch_bounds__(256)
tfc_l1_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, float scale, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += fabsf(a[i] - b[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tfc_block_commit(&g_l1_slot, ((double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]) * (double)scale, out);
}

// Full-spectrum log-magnitude MSE of the evaluation scripts (Devcom_MagMSE.py:91-118: mean_squared_error(log|fftshift(fft2(a))|,
// log|fftshift(fft2(b))|)) from the HALF spectra amp_a / amp_b [W][S][S/2+1]: |F[ky][kx]| = |F[-ky][-kx]| for real input, so the columns
// 1..S/2-1 count twice and columns 0 and S/2 once. out[w] += sum / S^2. One workgroup per (window, row block).
template <bool ABS>
__global__ void __launch_bounds__(256)
tfc_logmag_err_kernel(const float* __restrict__ a, const float* __restrict__ b, int S, float* out) {
  __shared__ float red[4];
  const int NB = S / 2 + 1;
  const int w = blockIdx.x;                                       // one workgroup per window, fixed summation order: no atomics, no memset
  const size_t base = (size_t)w * S * NB;
  float acc = 0.f;
  for (int i = threadIdx.x; i < S * NB; i += 256) {
    const int kx = i % NB;
    const float d = logf(a[base + i]) - logf(b[base + i]);
    acc += ((kx == 0 || kx == S / 2) ? 1.f : 2.f) * (ABS ? fabsf(d) : d * d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[w] = (((red[0] + red[1]) + red[2]) + red[3]) / ((float)S * (float)S);
}
hipError_t tfc_launch_logmag_mse(const float* a, const float* b, int S, int nwin, float* out, int absolute, hipStream_t st) {
  if (absolute) hipLaunchKernelGGL(tfc_logmag_err_kernel<true>, dim3(nwin), dim3(256), 0, st, a, b, S, out);
  else hipLaunchKernelGGL(tfc_logmag_err_kernel<false>, dim3(nwin), dim3(256), 0, st, a, b, S, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Temperature head (reference :255-268 vectorize_temps + datasets_temp.py:14-35 TempVector_PyTorch, loss :587-595).
//   vectorize : red channel of ToPILImage(x) = (uint8) trunc(x*255) (wraps mod 256) -> lut[u8]  (lut = float32(np.linspace(24,38,256)))
//   row triplet: F.triplet_margin_loss on [N,1,H,W] tensors: d(x,y) = ||x - y + eps||_2 over the last dim (one image row),
//                loss = mean_rows max(margin + d(a,p) - d(a,n), 0).  One wave per row.  No gradient (the reference detaches via PIL).
// ---------------------------------------------------------------------------------------------------
static __device__ TfcRedSlot g_rowtrip_slot;

__global__ void __launch_bounds__(256)
tfc_vectorize_temps_kernel(const float* __restrict__ x, long long bs, int rs, int H, int W, long long total,
                           const float* __restrict__ lut, float* __restrict__ out) {
  __shared__ float sl[256];
  sl[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xw = (int)(i % W);
    const long long r = i / W;
    const int y = (int)(r % H);
    const long long n = r / H;
    const float v = x[(size_t)n * bs + (size_t)y * rs + xw] * 255.f;
    out[i] = sl[((int)v) & 255];
  }
}

__global__ void __launch_bounds__(256)
tfc_row_triplet_kernel(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ ng, long long rows, int W,
                       float margin, float eps, float* loss) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  float lsum = 0.f;
  for (long long row = (long long)blockIdx.x * 4 + w; row < rows; row += (long long)gridDim.x * 4) {
    const size_t o = (size_t)row * W;
    float sp = 0.f, sn = 0.f;
    for (int xw = lane; xw < W; xw += 64) {
      const float av = a[o + xw];
      const float dp = av - p[o + xw] + eps, dn = av - ng[o + xw] + eps;
      sp += dp * dp; sn += dn * dn;
    }
    sp = wave_sum(sp); sn = wave_sum(sn);
    const float hinge = margin + sqrtf(sp) - sqrtf(sn);
    if (hinge > 0.f) lsum += hinge;
  }
  if (lane == 0) red[w] = lsum;
  __syncthreads();
  if (threadIdx.x == 0)
    tfc_block_commit(&g_rowtrip_slot, ((double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]) / (double)rows, loss, true);
}
hipError_t tfc_launch_vectorize_temps(const float* x, long long bs, int rs, int N, int H, int W, const float* lut, float* out, hipStream_t st) {
  const long long total = (long long)N * H * W;
  long long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(tfc_vectorize_temps_kernel, dim3((int)nb), dim3(256), 0, st, x, bs, rs, H, W, total, lut, out);
  return hipGetLastError();
}
hipError_t tfc_launch_row_triplet(const float* a, const float* p, const float* ng, long long rows, int W, float margin, float eps,
                                  float* loss, hipStream_t st) {
  long long nb = (rows + 3) / 4;                                  // the last workgroup to arrive STORES the mean (tfc_block_commit set): no memset launch
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(tfc_row_triplet_kernel, dim3((int)nb), dim3(256), 0, st, a, p, ng, rows, W, margin, eps, loss);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
hipError_t tfc_launch_triplet16(const float* fake, const float* real, const int* neg_idx, int N, int C, float margin, float eps,
                                float* loss, float* dfake, float gscale, hipStream_t st) {
  NegIdx ni;
  for (int i = 0; i < 16; ++i) ni.r[i] = neg_idx[i];
  long long nrows = (long long)N * C * 1024;
  long long nb = (nrows + 3) / 4;
  if (nb > 512) nb = 512;                                        // one double atomic + one ticket per workgroup on a single address
  hipLaunchKernelGGL(tfc_triplet16_kernel, dim3((int)nb), dim3(256), 0, st, fake, real, ni, N, C, margin, eps, loss, dfake, gscale);
  return hipGetLastError();
}

// S in {64, 256}; windows = N * wins_per_img; amp/pha: [windows][S][S/2+1]

// ---------------------------------------------------------------------------------------------------
// The same spectra by FFT (S = 64 or 256 = 4^3 / 4^4): radix-4 Stockham autosort passes in LDS, no bit reversal, exact sincospi twiddle table.
// The direct DFT above costs S^2 MACs per output row; at S = 256 (GLO-16, G16:294-313) that was 1.04 ms per call = 15 % of the GLO-16 step.
//   pass 1 (rows)   : a workgroup owns 32 consecutive rows of one window; two REAL rows are packed into one complex transform
//                     (z = row0 + i row1;  R0[k] = (Z[k] + conj Z[S-k]) / 2,  R1[k] = (Z[k] - conj Z[S-k]) / 2i); the half spectra go through an
//                     LDS tile to the scratch  T[window][kx][y]  (transposed, so that pass 2 reads whole columns as contiguous runs);
//   pass 2 (columns): complex transforms of CB columns per workgroup; amp = |F|, pha = atan2(Im, Re) (Im forced to +0 at the four self-conjugate
//                     bins, as the direct kernel does), staged in LDS and stored with the optional fftshift of both axes.
// One transform is carried by S/4 lanes (one radix-4 butterfly each per pass): a wave runs one 256-point or four 64-point transforms at a time.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(const float2 a, const float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// N-point forward transform of b0 (natural order in, natural order out); j = this lane's butterfly index 0 .. N/4-1; tw[k] = exp(-2 pi i k / N).
// The N/4 lanes of a transform belong to ONE wave: LDS accesses of a wave execute in program order, a compiler-level fence is all that is needed.
template <int N>
__device__ __forceinline__ float2* tfc_fft_r4(float2* b0, float2* b1, const float2* __restrict__ tw, int j) {
#pragma unroll
  for (int Ns = 1; Ns < N; Ns *= 4) {
    const int k = j & (Ns - 1);
    const int ts = k * (N / (4 * Ns));
    float2 v0 = b0[j], v1 = b0[j + N / 4], v2 = b0[j + N / 2], v3 = b0[j + 3 * N / 4];
    if (Ns > 1) { v1 = cmul(v1, tw[ts]); v2 = cmul(v2, tw[2 * ts]); v3 = cmul(v3, tw[3 * ts]); }
    const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
    const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y), a3 = make_float2(v1.y - v3.y, v3.x - v1.x);   // -i (v1 - v3)
    const int j0 = ((j - k) << 2) + k;
    b1[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
    b1[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
    b1[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
    b1[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float2* t = b0; b0 = b1; b1 = t;
  }
  return b0;
}

template <int S>
__global__ void __launch_bounds__(256)
tfc_fft_rows_kernel(const float* __restrict__ img, long long bs, long long cs, int rs, int C, int wins_x, int wins_per_img, float2* __restrict__ T) {
  constexpr int NB = S / 2 + 1, G = S / 4, FPW = 64 / G, RPB = 32;          // lanes per transform, transforms per wave at a time, rows per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);                   // [S]
  float2* wbuf = tw + S;                                          // [4 waves][2][FPW][S]
  float2* tile = wbuf + 4 * 2 * FPW * S;                          // [NB][RPB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = lane / G, j = lane % G;
  const int w = blockIdx.x / (S / RPB), y0 = (blockIdx.x % (S / RPB)) * RPB;
  const int n = w / wins_per_img, kw = w % wins_per_img;
  const float* base = img + (size_t)n * bs + (size_t)((kw / wins_x) * S) * rs + (kw % wins_x) * S;
  for (int i = tid; i < S; i += 256) {
    float sn, cn;
    sincospif(2.f * (float)i / (float)S, &sn, &cn);
    tw[i] = make_float2(cn, -sn);
  }
  __syncthreads();
  float2* b0 = wbuf + ((wave * 2 + 0) * FPW + f) * S;
  float2* b1 = wbuf + ((wave * 2 + 1) * FPW + f) * S;
  auto luma = [&](int y, int x) -> float {
    int q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = base[(size_t)(C == 1 ? 0 : c) * cs + (size_t)y * rs + x] * 255.f;
      q[c] = ((int)v) & 255;
    }
    return (float)((19595 * q[0] + 38470 * q[1] + 7471 * q[2] + 32768) >> 16);
  };
  for (int it = 0; it < RPB / 2 / (4 * FPW); ++it) {
    const int pr = (it * 4 + wave) * FPW + f;                     // row pair of this workgroup (0 .. 15)
    const int ya = y0 + 2 * pr;
#pragma unroll
    for (int r = 0; r < 4; ++r) b0[j + r * G] = make_float2(luma(ya, j + r * G), luma(ya + 1, j + r * G));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float2* Z = tfc_fft_r4<S>(b0, b1, tw, j);
    for (int k = j; k < NB; k += G) {
      const float2 a = Z[k], b = Z[(S - k) & (S - 1)];
      tile[k * RPB + 2 * pr] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
      tile[k * RPB + 2 * pr + 1] = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  float2* Tw = T + (size_t)w * NB * S;
  for (int i = tid; i < NB * RPB; i += 256) {
    const int kx = i / RPB, yy = i % RPB;
    Tw[(size_t)kx * S + y0 + yy] = tile[i];
  }
}

template <int S, int CB>
__global__ void __launch_bounds__(256)
tfc_fft_cols_kernel(const float2* __restrict__ T, float* __restrict__ amp, float* __restrict__ pha, int shift) {
  constexpr int NB = S / 2 + 1, G = S / 4, FPW = 64 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tw = reinterpret_cast<float2*>(smem);                   // [S]
  float2* wbuf = tw + S;                                          // [4 waves][2][FPW][S]
  float* ta = reinterpret_cast<float*>(wbuf + 4 * 2 * FPW * S);   // [S][CB] amplitude
  float* tp = ta + S * CB;                                        // [S][CB] phase
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = lane / G, j = lane % G;
  constexpr int NCB = (NB + CB - 1) / CB;
  const int w = blockIdx.x / NCB, c0 = (blockIdx.x % NCB) * CB;
  for (int i = tid; i < S; i += 256) {
    float sn, cn;
    sincospif(2.f * (float)i / (float)S, &sn, &cn);
    tw[i] = make_float2(cn, -sn);
  }
  __syncthreads();
  float2* b0 = wbuf + ((wave * 2 + 0) * FPW + f) * S;
  float2* b1 = wbuf + ((wave * 2 + 1) * FPW + f) * S;
  for (int it = 0; it < CB / (4 * FPW); ++it) {
    const int ci = (it * 4 + wave) * FPW + f;
    const int c = c0 + ci;
    if (c < NB) {                                                 // uniform per group of G lanes (a whole wave when FPW == 1)
      const float2* src = T + ((size_t)w * NB + c) * S;
#pragma unroll
      for (int r = 0; r < 4; ++r) b0[j + r * G] = src[j + r * G];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float2* F = tfc_fft_r4<S>(b0, b1, tw, j);               // columns past the half spectrum transform stale LDS: results unused
    if (c < NB)
      for (int ky = j; ky < S; ky += G) {
        float re = F[ky].x, im = F[ky].y;
        if ((c == 0 || c == S / 2) && (ky == 0 || ky == S / 2)) im = 0.f;
        ta[ky * CB + ci] = sqrtf(re * re + im * im);
        tp[ky * CB + ci] = atan2f(im, re);
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int i = tid; i < S * CB; i += 256) {
    const int ky = i / CB, ci = i % CB, c = c0 + ci;
    if (c >= NB) continue;
    int oy = ky, ox = c;
    if (shift) { oy = (ky + S / 2) % S; ox = (c + NB / 2) % NB; }
    const size_t oi = ((size_t)w * S + oy) * NB + ox;
    amp[oi] = ta[i];
    pha[oi] = tp[i];
  }
}

size_t tfc_fft_ws_bytes(int S, int nwin) { return (size_t)nwin * (S / 2 + 1) * S * sizeof(float2); }
template <int S, int CB>
static hipError_t launch_fft_t(const float* img, long long bs, long long cs, int rs, int C, int wins_x, int wins_per_img, int nwin, float* amp,
                               float* pha, int shift, void* ws, hipStream_t st) {
  constexpr int NB = S / 2 + 1, FPW = 64 / (S / 4);
  const size_t lds_r = (size_t)(S + 4 * 2 * FPW * S + NB * 32) * sizeof(float2);
  const size_t lds_c = (size_t)(S + 4 * 2 * FPW * S) * sizeof(float2) + (size_t)2 * S * CB * sizeof(float);
  hipLaunchKernelGGL((tfc_fft_rows_kernel<S>), dim3(nwin * (S / 32)), dim3(256), lds_r, st, img, bs, cs, rs, C, wins_x, wins_per_img, (float2*)ws);
  hipLaunchKernelGGL((tfc_fft_cols_kernel<S, CB>), dim3(nwin * ((NB + CB - 1) / CB)), dim3(256), lds_c, st, (const float2*)ws, amp, pha, shift);
  return hipGetLastError();
}
hipError_t tfc_launch_spectrum(const float* img, long long bs, long long cs, int rs, int C, int S, int wins_x, int wins_per_img,
                               int nwin, float* amp, float* pha, int shift, void* ws, hipStream_t st) {
  if (ws && S == 64) return launch_fft_t<64, 16>(img, bs, cs, rs, C, wins_x, wins_per_img, nwin, amp, pha, shift, ws, st);
  if (ws && S == 256) return launch_fft_t<256, 8>(img, bs, cs, rs, C, wins_x, wins_per_img, nwin, amp, pha, shift, ws, st);
  if (S == 64) {                                                  // no scratch given: direct DFT (also the independent cross-check of the FFT path)
    constexpr int KG = 33;
    const size_t lds = 64 * 64 + 2 * 64 * 4 + 64 * KG * 2 * 4;
    hipLaunchKernelGGL((tfc_spectrum_kernel<64, KG>), dim3(nwin, 1), dim3(256), lds, st, img, bs, cs, rs, C, wins_x, wins_per_img, amp, pha, shift);
  } else if (S == 256) {
    constexpr int KG = 16;
    const size_t lds = 256 * 256 + 2 * 256 * 4 + 256 * KG * 2 * 4;
    static bool attr_set = false;                                 // > 64 KiB of dynamic LDS needs an explicit opt-in (once)
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tfc_spectrum_kernel<256, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_set = true;
    }
    hipLaunchKernelGGL((tfc_spectrum_kernel<256, KG>), dim3(nwin, (129 + KG - 1) / KG), dim3(256), lds, st, img, bs, cs, rs, C, wins_x, wins_per_img, amp, pha, shift);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t tfc_launch_l1_sum(const float* a, const float* b, long long n, float scale, float* out, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(tfc_l1_sum_kernel, dim3((int)nb), dim3(256), 0, st, a, b, n, scale, out);
  return hipGetLastError();
}
