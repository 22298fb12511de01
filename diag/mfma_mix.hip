// Diagnostic: cycles per v_mfma_f32_32x32x16_bf16 when each group of 4 MFMAs carries the memory instructions of one k-substep of the gather GEMM
// (2 ds_read_b128 + 2 global_load_dwordx4 of L2-resident data + counted waits), by variant and by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ void __launch_bounds__(256, 2) k(const u32x4* __restrict__ src, float* sink, unsigned long long* stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<u32x4*>(smem)[i] = src[i & 1023];
  __syncthreads();
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + lane * 16;
  const u32x4* gp = src + lane + (blockIdx.x & 7) * 64;
  u32x4 a0 = src[threadIdx.x], a1 = src[threadIdx.x + 256], n0 = a0, n1 = a1;
  u32x4 b[4][2];
  for (int i = 0; i < 4; ++i) { b[i][0] = src[threadIdx.x + 64 * i]; b[i][1] = src[threadIdx.x + 64 * i + 32]; }
  f32x16_t c0 = {}, c1 = {}, c2 = {}, c3 = {};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                  // one k-substep: 4 MFMAs
      if (V >= 1) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1)); }
      if (V >= 2) { asm volatile("s_waitcnt vmcnt(6)" : "+v"(b[u & 3][0]), "+v"(b[u & 3][1])); }
      MF(c0, b[u & 3][0], a0); SB;
      if (V >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(n0) : "v"(lds), "n"(0) : "memory");
      SB; MF(c1, b[u & 3][0], a1); SB;
      if (V >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(n1) : "v"(lds), "n"(4096) : "memory");
      if (V >= 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[u & 3][0]) : "v"(gp) : "memory");
      SB; MF(c2, b[u & 3][1], a0); SB;
      SB; MF(c3, b[u & 3][1], a1); SB;
      if (V >= 2) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(b[u & 3][1]) : "v"(gp) : "memory");
      SB;
      if (V >= 1) { u32x4 t = a0; a0 = n0; n0 = t; t = a1; a1 = n1; n1 = t; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  u32x4* src; float* sink; unsigned long long* st;
  (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&sink, 4); (void)hipMalloc(&st, 4096 * 4 * 8);
  std::vector<unsigned> h(1 << 18);
  for (auto& v : h) { unsigned short lo = 0x3f80 ^ (rand() & 0x80ff), hi = 0x3f80 ^ (rand() & 0x80ff); v = lo | ((unsigned)hi << 16); }
  (void)hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int v = 0; v < 3; ++v) for (int wgs : {1, 256, 512}) for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    if (v == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 65536, 0, src, sink, st, iters);
    if (v == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 65536, 0, src, sink, st, iters);
    if (v == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 65536, 0, src, sink, st, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hs(wgs * 4);
    (void)hipMemcpy(hs.data(), st, wgs * 32, hipMemcpyDeviceToHost);
    double cyc = 0; for (auto x : hs) cyc += x; cyc /= hs.size();
    const double nm = (double)iters * 32;
    if (rep == 1) printf("variant %d (%s)  wgs %3d (%d per CU): cycles per MFMA per wave %.1f   %.0f TFLOP/s\n", v, v == 0 ? "MFMA only" : v == 1 ? "+ A ds_reads" : "+ A ds_reads + B global loads",
                         wgs, wgs > 256 ? 2 : 1, cyc / nm, nm * 32768.0 * wgs * 4 / ms / 1e9);
  }
  return 0;
}
