// Diagnostic: cycles per v_mfma_f32_32x32x16_bf16 / 16x16x32 in a bare loop, and the shader clock (s_memtime vs s_memrealtime), by grid size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 1) k(const u32x4* src, float* sink, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 256], b0 = src[threadIdx.x + 512], b1 = src[threadIdx.x + 768];
  f32x16_t c0 = {}, c1 = {}, c2 = {}, c3 = {};
  f32x4_t d[16] = {};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (SHAPE == 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b0), __builtin_bit_cast(bf16x8_t, a0), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b1), __builtin_bit_cast(bf16x8_t, a0), c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b0), __builtin_bit_cast(bf16x8_t, a1), c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, b1), __builtin_bit_cast(bf16x8_t, a1), c3, 0, 0, 0);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
          d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, (j & 1) ? b1 : b0), __builtin_bit_cast(bf16x8_t, (j & 2) ? a1 : a0), d[j], 0, 0, 0);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j] + d[j][0] + d[j][1] + d[j][2] + d[j][3];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) { stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

int main() {
  u32x4* src; float* sink; unsigned long long* st;
  hipMalloc(&src, 1024 * 16); hipMalloc(&sink, 4); hipMalloc(&st, 4096 * 4 * 16);
  std::vector<unsigned> h(4096);
  for (auto& v : h) { unsigned short lo = 0x3f80 ^ (rand() & 0x80ff), hi = 0x3f80 ^ (rand() & 0x80ff); v = lo | ((unsigned)hi << 16); }   // random bf16 in +-[1,2)
  hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int shape : {32, 16}) for (int wgs : {1, 8, 64, 256, 512, 1024}) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(wgs), dim3(256), 0, 0, src, sink, st, iters);
    else hipLaunchKernelGGL(k<16>, dim3(wgs), dim3(256), 0, 0, src, sink, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(wgs * 8);
    hipMemcpy(hs.data(), st, wgs * 64, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < wgs * 4; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
    cyc /= wgs * 4; rt /= wgs * 4;
    const double nm = (double)iters * 32;      // MFMAs per wave (both shapes: 32 per iteration)
    const double flop = nm * (shape == 32 ? 32768.0 : 16384.0) * wgs * 4;
    if (rep == 1) printf("shape %2d  wgs %4d: %.1f ms  memtime/MFMA %.1f  memtime/realtime %.2f (x100MHz)  -> %.0f TFLOP/s\n", shape, wgs, ms, cyc / nm, cyc / rt, flop / ms / 1e9);
  }
  return 0;
}
