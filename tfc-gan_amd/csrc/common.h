// Device-side helpers shared by all TFC-GAN HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tfc_desc.h"

typedef unsigned short bf16_t;                                   // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;      // one MFMA 32x32x16 operand fragment
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;                                           // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return __builtin_bit_cast(unsigned short, b);
}

// Streaming (non-temporal) 16-byte store / load: activation tensors are written once and read once by the NEXT kernel, hundreds
// of MB later -- keeping them out of the L2 write-allocate path leaves the L2 to the operands that are re-used (weights, halos).
// Measured on the implicit-GEMM epilogue: -20..28 % kernel time.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream16(void* p, const uint4& v) {
#ifdef TFC_NO_STREAM_STORES                                       // A/B build (scripts/build_variant.sh): plain stores
  *reinterpret_cast<uint4*>(p) = v;
#else
  u32x4_t vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<u32x4_t*>(p));
#endif
}
__device__ __forceinline__ uint4 load_stream16(const void* p) {
  const u32x4_t vv = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(vv.x, vv.y, vv.z, vv.w);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> {
  static constexpr int UE = 8;                                    // elements per 16-byte unit
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};
template <> struct ElemTraits<float> {
  static constexpr int UE = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// 16-byte vector <-> UE floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& u, float* v);
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& u, float* v) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack16<float>(const uint4& u, float* v) {
  v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* v);
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> one dword of two bf16 (RNE): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* v) {
  uint4 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  u.z = pack_bf16x2(v[4], v[5]);
  u.w = pack_bf16x2(v[6], v[7]);
  return u;
}
template <> __device__ __forceinline__ uint4 pack16<float>(const float* v) {
  uint4 u; u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
  return u;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// counter-based dropout RNG: keep-mask bit for element `idx` of stream `seed` (murmur3 finaliser of a 64-bit key)
__host__ __device__ __forceinline__ uint32_t tfc_hash32(uint32_t seed, uint32_t idx) {
  uint32_t x = idx * 0x9E3779B1u + seed;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x += seed * 0x27D4EB2Fu; x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
// p_keep threshold compare on the top 24 bits
__host__ __device__ __forceinline__ bool tfc_keep(uint32_t seed, uint32_t idx, uint32_t thresh24) {
  return (tfc_hash32(seed, idx) >> 8) >= thresh24;                // drop with probability thresh24 / 2^24
}

// XCD-aware bijective block remap: blocks b and b+8 share an XCD (observed round-robin dispatch), so give each
// XCD a contiguous range of logical ids; neighbouring tiles then hit the same 4 MiB L2. Speed only, never correctness.
__device__ __forceinline__ int tfc_xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, w = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + w;
}

// Scalar reduction across workgroups without fp32 atomic round-off: every workgroup adds its partial as a DOUBLE atomic
// (memory-side, device scope) and takes an arrival ticket; the last arriver reads the total, adds it to out[0] and re-arms
// the slot.  One thread per workgroup calls this.  Used for LOGGED LOSS SCALARS only (triplet, L1 / FFT, BCE, row triplet): nothing a gradient is
// computed from goes through it (those sums use fixed-order partials, tfc_part_reduce_kernel).  Slots are per-kernel __device__ globals: two launches
// of the SAME kernel must not overlap in time.  The engine runs two streams (nets.py), so the invariant is by kernel: BCE runs on the caller's stream
// only, the triplet / L1 / row-triplet heads on the side stream only (engine.step: pixel_losses), and a pluggable extra_loss_G -- which also runs on
// the side stream -- must not call tfc_bce_relativistic (it would share g_bce_slot with the main stream's call).
struct TfcRedSlot { double acc; unsigned cnt; unsigned pad; };
// `set`: the last arriver STORES the total (out[0] = total) instead of adding it, so the caller needs no memset launch in front of the kernel
__device__ __forceinline__ void tfc_block_commit(TfcRedSlot* slot, double partial, float* out, bool set = false) {
  atomicAdd(&slot->acc, partial);
  __threadfence();
  const unsigned t = atomicAdd(&slot->cnt, 1u);
  if (t == gridDim.x * gridDim.y * gridDim.z - 1) {
    __threadfence();
    const double tot = atomicAdd(&slot->acc, 0.0);
    atomicExch(reinterpret_cast<unsigned long long*>(&slot->acc), 0ull);
    atomicExch(&slot->cnt, 0u);
    out[0] = set ? (float)tot : out[0] + (float)tot;
  }
}
