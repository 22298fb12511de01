// Device probe of the lane maps the MFMA kernels rely on (run once by the GPU test-suite before anything else):
//   [0,1024)    v_mfma_f32_32x32x16_bf16 : lane l holds A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31]; D[(j&3)+8(j>>2)+4(l>>5)][l&31]
//   [1024,2048) v_mfma_f32_32x32x2_f32   : lane l holds A[l&31][l>>5], B[l>>5][l&31]; same D map
//   [2048,2560) ds_read_b64_tr_b16 pair  : lane l, element e = M[8(l>>5)+e][16((l>>4)&1) + (l&15)] of a [16][32] LDS tile
// Integer-valued operands, so every result is exact.
#include "common.h"

__global__ void __launch_bounds__(64) tfc_probe_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[16 * 32];
  const int l = threadIdx.x;
  const int r = l & 31, h = l >> 5;
  // bf16 32x32x16
  {
    uint4 a, b;
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * h + j;
      av[j] = (float)((3 * r + k) % 7);                        // A[i=r][k]
      bv[j] = (float)((k + 5 * r) % 5);                        // B[k][j=r]
    }
    a = pack16<bf16_t>(av);
    b = pack16<bf16_t>(bv);
    f32x16_t c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    for (int j = 0; j < 16; ++j) out[l * 16 + j] = c[j];
  }
  // fp32 32x32x2
  {
    const float a = (float)((3 * r + h) % 7), b = (float)((h + 5 * r) % 5);
    f32x16_t c;
    for (int j = 0; j < 16; ++j) c[j] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    for (int j = 0; j < 16; ++j) out[1024 + l * 16 + j] = c[j];
  }
  // transposing LDS read
  {
    for (int i = l; i < 16 * 32; i += 64) tile[i] = (unsigned short)i;   // M[row][col] = row*32 + col
    __syncthreads();
    const int grp = l >> 4, li = l & 15;
    const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(tile) + (8 * hk + q) * 64 + cb16 * 32 + p * 8;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, base));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, base + 4 * 64));
    for (int e = 0; e < 4; ++e) {
      out[2048 + l * 8 + e] = (float)(unsigned short)lo[e];
      out[2048 + l * 8 + 4 + e] = (float)(unsigned short)hi[e];
    }
  }
}

hipError_t tfc_launch_probe(float* out, hipStream_t st) {
  hipLaunchKernelGGL(tfc_probe_kernel, dim3(1), dim3(64), 0, st, out);
  return hipGetLastError();
}
