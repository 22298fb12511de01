// Halo-staged implicit-GEMM convolution kernels for gfx950 (MI355X), bf16 and fp32.
//
//   tfc_igemm_kernel   : forward-type gather GEMM (conv fwd, conv dgrad, convT fwd phases, convT dgrad,
//                        upsample+pad+conv, pad+conv) -- see tfc_desc.h for the abstract model.
//   tfc_wgrad_kernel   : weight-gradient GEMM (K = pixels) for the same gather descriptors.
//   tfc_pack_w_kernel  : torch-layout fp32 weights -> MFMA-fragment-ordered operand stream.
//   tfc_wgrad_finish_kernel : fp32 accumulator [slot][n][c] -> torch-layout gradient.
//
// Design (MI355X-first, not a cuDNN-style im2col):
//   * a workgroup (4 waves) owns an 8x16 tile of output pixels of one image; per 64-byte channel chunk it stages
//     the (8+3)x(16+3) input halo ONCE in LDS and replays it for all 16 taps, so the K loop over taps is
//     barrier-free: A fragments are ds_read_b128 from the halo at tap-constant offsets, B fragments are
//     1-KiB fully coalesced global_load_dwordx4 from a weight stream pre-packed in MFMA fragment order
//     (L2-resident), software-pipelined in registers;
//   * LDS image: pixel stride = chunk+16 B, row pitch 24 pixels and rows interleaved over lane parity, which
//     makes every ds_read_b128 lane group hit 16 distinct 16-byte slots (conflict-free; derivation below);
//   * bf16 uses v_mfma_f32_32x32x16_bf16, fp32 (parity mode) uses v_mfma_f32_32x32x2_f32 on the same
//     byte geometry (a 16-byte unit = 8 bf16 = 4 fp32 channels);
//   * epilogue fuses bias, InstanceNorm statistics (wave-shuffle + fp32 atomics), skip-gradient
//     accumulation, and tanh + NCHW store for the generator head.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "pack_math.h"

// ---------------------------------------------------------------------------------------------------
// MFMA wrappers: one "k-substep" consumes one 16-byte unit per lane of A and of B.
// bf16: 1 x 32x32x16 (lane (r,h) holds k = 8h..8h+7).  fp32: 4 x 32x32x2 (lane (r,h), element q is k = 4h+q
// of the 8-channel substep; both operands use the same (h,q) -> channel map so any consistent order is exact).
// ---------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// Tile pixel of MFMA row `row` (0..31) of M-subtile `ms` (0..3):  ty = 2*ms + (row & 1), tx = row >> 1.
// With row pitch P = 24 pixels the pixel index (ty*P + tx) of the 16 lanes of every ds_read_b128 lane group
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) is distinct mod 16, and the pixel stride is an odd multiple of
// 16 B (80 / 48 / 16), so the 16 lanes land on 16 distinct 16-byte slots of the 256-byte bank row.

// ---------------------------------------------------------------------------------------------------
// forward-type gather GEMM
//   PAT: compile-time tap pattern of every plane (needs 64-byte chunks, i.e. UPP == 4), 0 = generic run-time tables
//     1: 16 taps, 4x4 raster (dy = t>>2, dx = t&3)      conv / pad+conv, forward and dgrad
//     2:  4 taps, dy = 1-(t>>1), dx = 1-(t&1)            transposed-conv forward phases
//     3:  4 taps, dy = t>>1, dx = t&1                    transposed-conv dgrad parity planes
//   With PAT != 0 the per-stage body is fully unrolled: every A read is ds_read_b128 base+immediate, the weight stream is
//   prefetched BD k-substeps ahead through a statically indexed register ring, and nothing scalar is loaded in the loop.
// ---------------------------------------------------------------------------------------------------
template <int PAT> struct TapPat;     // taps enumerated row-major: t = row * COLS + col; REV mirrors both axes
template <int R, int C, bool REV> struct TapPatRC {
  static constexpr int ROWS = R, COLS = C;
  static constexpr int dy(int r) { return REV ? R - 1 - r : r; }
  static constexpr int dx(int c) { return REV ? C - 1 - c : c; }
};
template <> struct TapPat<1> : TapPatRC<4, 4, false> {};   // conv / pad+conv, forward and dgrad
template <> struct TapPat<2> : TapPatRC<2, 2, true> {};    // transposed-conv forward phases
template <> struct TapPat<3> : TapPatRC<2, 2, false> {};   // transposed-conv dgrad parity planes, upsample-conv phase (0,0)
template <> struct TapPat<6> : TapPatRC<3, 3, false> {};   // upsample-conv: every phase on the 3 x 3 source-offset grid

#ifndef TFC_BD
#define TFC_BD 4
#endif
// Diagnostic build only (-DTFC_STAMP, scripts/stamp_igemm.py): s_memtime stamps of the kernel's phases go to `out_nchw` (a buffer of their
// own, never an output element); no stamp executes in the shipped library.
#ifdef TFC_STAMP
#define TFC_STAMP_AT(slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0 && out_nchw && !(flags & TFC_EP_TANH_NCHW)) reinterpret_cast<unsigned long long*>(out_nchw)[((size_t)blockIdx.x * 4 + wave) * 16 + (slot)] = t_; } while (0)
#else
#define TFC_STAMP_AT(slot) do { } while (0)
#endif
#ifndef TFC_MINW
#define TFC_MINW 3
#endif
#ifndef TFC_BD2
#define TFC_BD2 4          // persistent kernel: weight-ring depth (k-substeps); 3 x 3 tap pattern (18 k-substeps per stage): 2, 3 or 6
#endif
#ifndef TFC_BD2_PAT6
#define TFC_BD2_PAT6 6
#endif
template <typename T, int MT, int NT, int WM, int WN, int PAT>
__global__ void __launch_bounds__(256, TFC_MINW)
tfc_igemm_kernel(const TfcGather d, const T* __restrict__ in, const uint4* __restrict__ wp, T* out,
                 const float* __restrict__ bias, float* stats, float* out_nchw, const float* __restrict__ oscale,
                 int flags, int NB32, int nblkN, int buf_bytes, long long phase_wbytes) {
  static_assert(WM * WN == 4, "4 waves");
  static_assert(WM * MT == 4, "tile is 4 M-subtiles (128 pixels)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  constexpr int UE = 16 / ES;
  constexpr int P = TFC_LDS_P;
  constexpr int BD = (PAT == 4 || PAT == 6) ? 2 : TFC_BD;         // weight-stream prefetch distance (k-substeps), PAT != 0

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, r = lane & 31;
  TFC_STAMP_AT(0);

  const int PB = PAT ? 64 : tfc_pb(d.Cin_pad, ES);
  const int UPP = PB >> 4;
  const int upp_shift = (UPP == 4) ? 2 : (UPP == 2 ? 1 : 0);
  const int PS = PAT ? 80 : tfc_ps(PB);
  const int CK = PB / ES;
  const int nchunks = (d.Cin_pad * ES) / PB;
  const int nst = nchunks * d.nplanes;

  const int bid = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int nb_blk = bid % nblkN;
  int tile = bid / nblkN;
  const int txb = tile % d.tiles_x; tile /= d.tiles_x;
  const int tyb = tile % d.tiles_y; tile /= d.tiles_y;
  const int img = tile % d.nimg;
  const int phase = tile / d.nimg;                              // sub-pixel phase folded into the grid (transposed conv)
  const int phy = phase >> 1, phx = phase & 1;
  const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
  const int nb0 = (nb_blk * WN + wn) * NT;                     // first 32-channel block of this wave

  // A-operand lane base inside a halo buffer
  const int laneBase = ((2 * wm * MT + (r & 1)) * P + (r >> 1)) * PS + (PAT ? h * 16 : 0);
  const int MSTRIDE = 2 * P * PS;

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][nt][j] = 0.f;

  // ---- halo staging helpers (register-staged: zero fill for padding, padded LDS pixel stride) ----
  const T* in_img = in + (size_t)img * d.IH * d.IW * d.in_pitch;
  uint4 hv[4];
  int hoff[4];
  auto halo_load = [&](int st) {
    const int cc = st / d.nplanes, pl = st - cc * d.nplanes;
    const TfcPlane& pd = d.plane[pl];
    const int nunits = pd.hh * pd.hw * UPP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      hv[i] = make_uint4(0, 0, 0, 0);
      hoff[i] = -1;
      if (idx < nunits) {
        const int pix = idx >> upp_shift, g = idx & (UPP - 1);
        const int hy = pix / pd.hw, hx = pix - hy * pd.hw;
        hoff[i] = (hy * P + hx) * PS + g * 16;
        const int y = (a0 + pd.dy0 + phy * d.ph_d0 + hy) * d.SS + pd.py;
        const int x = (b0 + pd.dx0 + phx * d.ph_d0 + hx) * d.SS + pd.px;
        if (y >= 0 && y < d.IH && x >= 0 && x < d.IW)
          hv[i] = *reinterpret_cast<const uint4*>(in_img + ((size_t)(y * d.IW + x)) * d.in_pitch + cc * CK + g * UE);
      }
    }
  };
  auto halo_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (hoff[i] >= 0) *reinterpret_cast<uint4*>(buf + hoff[i]) = hv[i];
  };

  // ---- weight stream: wave-uniform base (scalar registers) + constant per-lane offset; the stream carries TFC_WPAD
  //      k-substeps of slack after its last real one, so the prefetch never needs a bounds check ----
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(wp) + (size_t)phase * (size_t)phase_wbytes;
  const unsigned laneoff = (unsigned)(nb0 * 64 + lane) * 16u;
  const size_t wstep_b = (size_t)NB32 * 1024;
  auto loadB = [&](int gs, uint4 (&b)[NT]) {
    const unsigned char* pw = wbase + (size_t)gs * wstep_b;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const uint4*>(pw + laneoff + nt * 1024);
  };

  // epilogue scalars, requested NOW: a load issued at the start of the epilogue would stall every workgroup for a full memory round trip
  const float osc = oscale ? *oscale : 1.f;                      // spectral norm: conv(x, W / sigma) = conv(x, W) / sigma
  float bvs[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (nb0 + nt) * 32 + r;
    bvs[nt] = ((flags & TFC_EP_BIAS) && n < d.Nout) ? bias[n] : 0.f;
  }

  if constexpr (PAT != 0) {
    constexpr int NSR = TapPat<PAT>::COLS * 2;                   // k-substeps per filter row (2 per tap: 4 units of 16 B)
    static_assert(NSR % BD == 0, "register ring must realign every filter row");
    uint4 br[BD][NT];
#pragma unroll
    for (int i = 0; i < BD; ++i) loadB(i, br[i]);
    halo_load(0);
    halo_store(smem);
    __syncthreads();
    TFC_STAMP_AT(1);
    int gs = 0;
    for (int st = 0; st < nst; ++st) {
      const bool more = (st + 1) < nst;
      if (more) halo_load(st + 1);
      const unsigned char* buf = smem + (st & 1) * buf_bytes + laneBase;
#pragma unroll 1
      for (int row = 0; row < TapPat<PAT>::ROWS; ++row) {
        const unsigned char* rbuf = buf + (PAT == 2 ? (1 - row) : row) * (P * 80);   // == TapPat::dy(row)
        uint4 a[2][MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) a[0][mi] = *reinterpret_cast<const uint4*>(rbuf + TapPat<PAT>::dx(0) * 80 + mi * (2 * P * 80));
#pragma unroll
        for (int s = 0; s < NSR; ++s) {
          if (s + 1 < NSR) {                                     // A fragments one k-substep ahead
            const int off = TapPat<PAT>::dx((s + 1) >> 1) * 80 + ((s + 1) & 1) * 32;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) a[(s + 1) & 1][mi] = *reinterpret_cast<const uint4*>(rbuf + off + mi * (2 * P * 80));
          }
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(a[s & 1][mi], br[s % BD][nt], acc[mi][nt]);
          loadB(gs + s + BD, br[s % BD]);                        // weights BD k-substeps ahead
          asm volatile("" ::: "memory");                         // keep the issue order: hipcc otherwise sinks the prefetch to its use
        }
        gs += NSR;
      }
      if (more) halo_store(smem + ((st + 1) & 1) * buf_bytes);
      __syncthreads();
    }
  } else {
    uint4 b0r[NT], b1r[NT];
    loadB(0, b0r);
    loadB(1, b1r);
    halo_load(0);
    halo_store(smem);
    __syncthreads();
    int gs = 0;
    for (int st = 0; st < nst; ++st) {
      const bool more = (st + 1) < nst;
      if (more) halo_load(st + 1);
      const unsigned char* buf = smem + (st & 1) * buf_bytes + laneBase;
      const int pl = st % d.nplanes;
      const TfcPlane& pd = d.plane[pl];
      const int nsub = tfc_nsub(pd.ntaps, PB);                    // even by construction
      for (int s = 0; s < nsub; s += 2) {
        {
          const int u0 = 2 * s, u1 = u0 + 1;
          const int t0 = u0 >> upp_shift, t1 = u1 >> upp_shift;
          const int off0 = (pd.tap_dy[t0] * P + pd.tap_dx[t0]) * PS + (u0 & (UPP - 1)) * 16;
          const int off1 = (pd.tap_dy[t1] * P + pd.tap_dx[t1]) * PS + (u1 & (UPP - 1)) * 16;
          const int off = h ? off1 : off0;
          uint4 a[MT];
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) a[mi] = *reinterpret_cast<const uint4*>(buf + off + mi * MSTRIDE);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(a[mi], b0r[nt], acc[mi][nt]);
          loadB(gs + 2, b0r);
        }
        {
          const int u0 = 2 * s + 2, u1 = u0 + 1;
          const int t0 = u0 >> upp_shift, t1 = u1 >> upp_shift;
          const int off0 = (pd.tap_dy[t0] * P + pd.tap_dx[t0]) * PS + (u0 & (UPP - 1)) * 16;
          const int off1 = (pd.tap_dy[t1] * P + pd.tap_dx[t1]) * PS + (u1 & (UPP - 1)) * 16;
          const int off = h ? off1 : off0;
          uint4 a[MT];
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) a[mi] = *reinterpret_cast<const uint4*>(buf + off + mi * MSTRIDE);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(a[mi], b1r[nt], acc[mi][nt]);
          loadB(gs + 3, b1r);
        }
        gs += 2;
      }
      if (more) halo_store(smem + ((st + 1) & 1) * buf_bytes);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  TFC_STAMP_AT(2);
  constexpr bool STAGED = (ES == 2);                             // bf16: transpose through LDS, store whole 16-byte units
  constexpr int BN = 32 * NT * WN;
  constexpr int ROWP = BN * ES + 16;                             // LDS bytes per pixel row of the staged tile
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = (nb0 + nt) * 32 + r;
    const bool nok = n < d.Nout;
    const float bv = bvs[nt];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
        const int ty = 2 * (wm * MT + mi) + (row & 1), tx = row >> 1;
        const int a = a0 + ty, b = b0 + tx;
        const bool ok = nok && a < d.GH && b < d.GW;
        float v = acc[mi][nt][j] * osc + bv;
        if (flags & TFC_EP_LEAKY) v = fmaxf(v, 0.2f * v);
        if (flags & TFC_EP_RELU) v = fmaxf(v, 0.f);
        if (STAGED && !(flags & TFC_EP_TANH_NCHW)) {
          if (ok) { s1 += v; s2 += v * v; }
          *reinterpret_cast<bf16_t*>(smem + (ty * TFC_TILE_W + tx) * ROWP + ((wn * NT + nt) * 32 + r) * 2) = f32_to_bf16(v);
        } else if (ok) {
          const int oy = a * d.OS + d.OOY + phy * d.ph_oo, ox = b * d.OS + d.OOX + phx * d.ph_oo;
          if (flags & TFC_EP_TANH_NCHW) {
            out_nchw[(((size_t)img * d.Nout + n) * d.OH + oy) * d.OW + ox] = tanhf(v);
          } else {
            T* po = out + ((size_t)(img * d.OH + oy) * d.OW + ox) * d.out_pitch + n;
            if (flags & TFC_EP_ACCUM) v += ElemTraits<T>::ld(po);
            ElemTraits<T>::st(po, v);
          }
          s1 += v; s2 += v * v;
        }
      }
    }
    if (flags & TFC_EP_STATS) {                                  // stats = part[img][tile * WM + wm][Nout][2] (plain stores; summed in a fixed order afterwards)
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0 && nok) {
        const int tpi = d.tiles_y * d.tiles_x;                   // folded sub-pixel phases: every phase has its own tile slots
        const size_t slot = (((size_t)img * (d.ph_n > 1 ? d.ph_n : 1) + phase) * tpi + (tyb * d.tiles_x + txb)) * WM + wm;
        reinterpret_cast<float2*>(stats)[slot * d.Nout + n] = make_float2(s1, s2);
      }
    }
  }
  TFC_STAMP_AT(3);
  if (STAGED && !(flags & TFC_EP_TANH_NCHW)) {
    __syncthreads();
    TFC_STAMP_AT(4);
    constexpr int UPR = BN / 8;                                  // 16-byte units per pixel row of the tile
    const int nbase = nb_blk * BN;
#pragma unroll 2
    for (int idx = tid; idx < 128 * UPR; idx += 256) {
      const int pix = idx / UPR, u = idx - pix * UPR;
      const int ty = pix >> 4, tx = pix & 15;
      const int a = a0 + ty, b = b0 + tx;
      const int n0 = nbase + u * 8;
      if (a < d.GH && b < d.GW && n0 < d.Nout) {
        const int oy = a * d.OS + d.OOY + phy * d.ph_oo, ox = b * d.OS + d.OOX + phx * d.ph_oo;
        T* po = out + ((size_t)(img * d.OH + oy) * d.OW + ox) * d.out_pitch + n0;
        uint4 v = *reinterpret_cast<const uint4*>(smem + pix * ROWP + u * 16);
        if (n0 + 8 <= d.Nout) {
          if (flags & TFC_EP_ACCUM) {
            float f[8], g[8];
            unpack16<bf16_t>(v, f);
            unpack16<bf16_t>(*reinterpret_cast<const uint4*>(po), g);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += g[e];
            v = pack16<bf16_t>(f);
          }
          store_stream16(po, v);
        } else {                                                 // ragged tail (Nout not a multiple of 8): element stores
          float f[8];
          unpack16<bf16_t>(v, f);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n0 + e < d.Nout) {
              float fe = f[e];
              if (flags & TFC_EP_ACCUM) fe += ElemTraits<T>::ld(po + e);
              ElemTraits<T>::st(po + e, fe);
            }
        }
      }
    }
  }
#ifdef TFC_STAMP
  TFC_STAMP_AT(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFC_STAMP_AT(6);
  if (lane == 0 && out_nchw && !(flags & TFC_EP_TANH_NCHW)) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    reinterpret_cast<unsigned long long*>(out_nchw)[((size_t)blockIdx.x * 4 + wave) * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// tfc_igemm2_kernel -- the bf16 production form of the gather GEMM above: PERSISTENT workgroups, two per CU.
//
// What the s_memtime stamps of the one-tile-per-workgroup kernel showed at the shape of down2 (profiles/r02_stamps_igemm_v1.md): a wave spends
// 16 % of its life in the prologue (first halo + weight fragments: a cold round trip per tile), 19 % transposing its accumulators through
// LDS with 64 two-byte ds_write per lane (all twelve waves of a CU contend for the LDS store path), and inside the K loop it already runs at
// the MFMA-bound rate; the 768-slot grid also quantises badly (5.33 / 2.67 / 1.33 rounds for down2 / down3 / down4).  Hence:
//   * grid = 2 workgroups per CU (512): a workgroup walks work items  b, b + G, ...  (8 / 4 / 2 whole rounds for the three big shapes);
//   * the stream of (tile, stage) is ONE software pipeline: the halo of the next tile's first chunk is requested at the start of the current
//     tile's last stage and stored behind it, the weight ring runs on into the next tile's stream during the last filter row, and both are
//     issued BEFORE the epilogue's stores (vmcnt retires in order) -- a tile never starts cold;
//   * operands are SWAPPED in the MFMA (A = weight fragment, B = pixel fragment): the accumulator then holds  row = channel, column (lane) =
//     pixel, i.e. four consecutive CHANNELS of one pixel per four registers.  Two v_cvt_pk + one v_permlane32_swap pair turn them into whole
//     16-byte units (8 channels) and the tile reaches LDS with 8 ds_write_b128 per lane instead of 64 ds_write_b16 (conflict-free: rows in
//     MFMA order, row pitch = 4 banks mod 32);
//   * bias lives in LDS per tile (the channel now varies with the register, not with the lane); InstanceNorm statistics are taken in the
//     store pass from the bf16 values actually stored (the values the normaliser will read), reduced over the lanes that share a unit column.
// Weight fragments still stream from L2 per wave (lane-linear 1-KiB loads): cfg "no redundant B" measured +3 % only, the prologue/epilogue
// were the loss.
// ---------------------------------------------------------------------------------------------------
template <int OFF> __device__ __forceinline__ void lds_rd(u32x4_t& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF) : "memory");
}
// compile-time loop: f(std::integral_constant<int, I>{}) for I = B .. E-1 (immediates of the inline-asm statements must be constants)
template <int B, int E, typename F> __device__ __forceinline__ void tfc_static_for(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); tfc_static_for<B + 1, E>(f); }
}
struct Tile2 {
  int img, a0, b0, phy, phx, nb_blk;
  const unsigned char* wbase;                                    // this tile's weight stream (phase, n-block), wave part excluded
};

// HALVES == 2 (round 3): ONE 512-thread workgroup per CU whose two 4-wave halves each do what a workgroup of the HALVES == 1 form does (own tiles, own LDS
// region of half_bytes) but share the workgroup barrier, and half 1 runs `shift` barrier slots BEHIND half 0. Why: two independent workgroups on a CU
// phase-lock -- a workgroup whose SIMD partner is in its epilogue has the matrix pipe to itself, finishes its K loop early and catches up, so both end up
// in their K loops together (sharing the pipe) and in their MFMA-free epilogues together (pipe idle: 8.2 k MFMA cycles per wave and tile inside a 27-32 k
// cycle tile period, SQ_VALU_MFMA_BUSY 50-60 %; a start stagger re-locks within a few tiles, measured neutral in round 2). With a common barrier
// sequence the offset cannot drift: per tile a half executes nst + 1 barriers (one per stage, one inside the epilogue), so with half 1 shifted by
// (nst + 1) / 2 slots one half's epilogue (accumulators -> LDS -> global: no MFMA) always lies beside the other half's K-loop stages.
// Both halves execute the SAME total number of barriers (dummy barriers pad the shorter sequence), so every wave reaches the end.
template <int MT, int NT, int WM, int WN, int PAT, int HALVES = 1>
__global__ void __launch_bounds__(256 * HALVES, 2)
tfc_igemm2_kernel(const TfcGather d, const bf16_t* __restrict__ in, const uint4* __restrict__ wp, bf16_t* out,
                  const float* __restrict__ bias, float* stats, float* dbg, const float* __restrict__ oscale,
                  int flags, int NB32, int nblkN, int buf_bytes, long long phase_wbytes, int nwork, int half_bytes, int shift_opts) {
  const int shift = shift_opts & 255;
  const bool prio = (shift_opts >> 8) & 1;                        // (A/B knob TFC_IGEMM_PRIO) raise the wave priority inside the K loop
  static_assert(WM * WN == 4 && WM * MT == 4, "4 waves, 128-pixel tile");
  static_assert(PAT != 0, "compile-time tap patterns only");
  static_assert(HALVES == 1 || HALVES == 2, "one or two 4-wave halves");
  static_assert(HALVES == 1 || 32 * NT * WN <= 128, "the two-half form has no 256-channel tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int half = HALVES == 2 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  unsigned char* const smem = smem_all + half * half_bytes;
  int nbar = 0;                                                  // workgroup barriers executed so far by this wave (HALVES == 2: padded to a common total)
#define TFC_BAR() do { __builtin_amdgcn_s_barrier(); ++nbar; } while (0)
#define TFC_SYNC() do { __syncthreads(); ++nbar; } while (0)
  typedef bf16_t T;
  constexpr int P = TFC_LDS_P;
  constexpr int NSR = TapPat<PAT>::COLS * 2;
  constexpr int ROWS = TapPat<PAT>::ROWS;
  // weight-ring depth in k-substeps. A wave has BD * NT one-KiB fragment loads in flight; the deep layers' operand streams (8-17 MB) miss the 4 MB
  // L2 of an XCD, so their K loop ran at (fragments in flight) / (memory latency): 4 per ~500 ns = 230 cycles per k-substep instead of 64.
  // One-fragment-per-substep tiles (NT == 1: the configurations those layers use) therefore keep 8 substeps in flight (32 VGPRs).
#ifndef TFC_BD2_NT1
#define TFC_BD2_NT1 8
#endif
  constexpr int BDW = (NT == 1) ? TFC_BD2_NT1 : TFC_BD2;
  constexpr int BD = (PAT == 6) ? (NT == 1 ? 6 : TFC_BD2_PAT6) : (BDW <= ROWS * NSR ? BDW : ROWS * NSR);
  static_assert(BD <= TFC_WPAD, "the packed stream carries TFC_WPAD slack k-substeps for the ring's read-ahead");
  static_assert((ROWS * NSR) % BD == 0, "register ring must realign every stage");
  constexpr int BN = 32 * NT * WN;
  // 256-channel workgroup tile (a wave owns 128 pixels x 64 channels: half the LDS operand traffic per MFMA of <4,1,1,4>): wave wn owns the
  // 32-channel blocks {wn, wn + WN, ...} (INTERLEAVED, so that every wave takes part in each 128-channel pass of the epilogue) and the
  // staged tile holds one 128-channel pass at a time (77 KB of LDS per workgroup: two workgroups per CU)
  constexpr bool IL = BN > 128;
  constexpr int NPASS = IL ? NT : 1;                             // epilogue passes of BNS channels
  constexpr int BNS = IL ? 32 * WN : BN;
  constexpr int ROWP = BNS * 2 + 16;                             // staged tile: bytes per pixel row (pitch = 4 banks mod 32)
  constexpr int UPR = BNS / 8;                                   // 16-byte units per pixel row
  constexpr int BSTEP = IL ? WN * 1024 : 1024;                   // byte distance of a wave's consecutive weight fragments in the stream
  float* out_nchw = dbg;                                         // TFC_STAMP_AT writes here in the diagnostic build
  (void)out_nchw;

  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, r = lane & 31;
  const int G = gridDim.x * HALVES;
  TFC_STAMP_AT(0);
#ifdef TFC_STAMP
#define TFC_NOW(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
  unsigned long long tk0 = 0, tk1 = 0, acc_main = 0, acc_ep1 = 0, acc_bar = 0, acc_store = 0, ntile = 0, ts0 = 0, ts1 = 0, acc_sync = 0, tq0 = 0, tq1 = 0, tq2 = 0, acc_h = 0, acc_f = 0;
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime(), mt_begin = __builtin_amdgcn_s_memtime();
#else
#define TFC_NOW(v) do { } while (0)
#endif

  const int nchunks = (d.Cin_pad * 2) / 64;
  const int nst = nchunks * d.nplanes;
  // LDS map.  Narrow tiles: [halo 0 | halo 1 | staged tile | statistics | bias].  256-channel tile: [halo 0 | X | halo 1 | statistics | bias] -- the
  // staged tile of the epilogue OVERLAYS the halo buffer the K loop has just consumed plus the gap X (the other one already holds the next
  // tile's first chunk): [halo 0 | X] or [X | halo 1], 62 KB per workgroup instead of 83 KB, i.e. two workgroups per CU.
  const int xtra = IL ? (128 * ROWP > buf_bytes ? 128 * ROWP - buf_bytes : 0) : 0;
  const int bstride = buf_bytes + xtra;                          // distance of the two halo buffers
  unsigned char* stage_fix = smem + 2 * buf_bytes;
  float* sstat = reinterpret_cast<float*>(IL ? smem + 2 * buf_bytes + xtra : stage_fix + 128 * ROWP);   // [4 waves][BN][2] statistics partials of the tile just stored (EP_STATS only)
  float* sbias = sstat + ((flags & TFC_EP_STATS) ? 8 * BN : 0);  // bias of EVERY output channel of the layer (nblkN * BN floats), loaded once (EP_BIAS only)

  const int laneBase = ((2 * wm * MT + (r & 1)) * P + (r >> 1)) * 80 + h * 16;
  const unsigned lanepart = (unsigned)((IL ? wn : wn * NT) * 64 + lane) * 16u;
  const size_t wstep_b = (size_t)NB32 * 1024;
  const float osc = oscale ? *oscale : 1.f;

  auto decode = [&](int bid, Tile2& t) {                          // bid: logical work item (n-block fastest, then tile, image, phase)
    t.nb_blk = bid % nblkN;
    int tile = bid / nblkN;
    const int txb = tile % d.tiles_x; tile /= d.tiles_x;
    const int tyb = tile % d.tiles_y; tile /= d.tiles_y;
    t.img = tile % d.nimg;
    const int phase = tile / d.nimg;
    t.phy = phase >> 1; t.phx = phase & 1;
    t.a0 = tyb * TFC_TILE_H; t.b0 = txb * TFC_TILE_W;
    t.wbase = reinterpret_cast<const unsigned char*>(wp) + (size_t)phase * (size_t)phase_wbytes + (size_t)(t.nb_blk * WN * NT) * 1024;
  };

  // ---- memory operations of the K loop: inline asm with hand-counted waits.  hipcc (ROCm 7.2) schedules the compiler-visible form badly
  //      (the ISA it emitted: s_waitcnt vmcnt(0) in front of every conditional halo load -- five drains of the weight ring per stage -- and
  //      each ds_read_b128 sunk to its MFMA and followed by lgkmcnt(0)); an asm load is invisible to its s_waitcnt bookkeeping, so EVERY
  //      wait below is counted by hand and the counts are STATIC: the four halo loads and the NT weight loads of a k-substep are always
  //      issued (clamped addresses instead of branches).  vmcnt retires in order:  wait vmcnt(N) guarantees a load once at most N younger
  //      vector-memory operations exist; assuming FEWER younger operations than really were issued only waits longer, never too short.
  u32x4_t hv[4];
  int hoff[4];                                                   // LDS byte offset of this thread's halo unit i (-1: none)
  int hyS[4], hxS[4], hrel[4];                                   // tile-invariant source geometry: row / column step and element offset of the unit
  unsigned hvalid = 0;                                           // bit i: halo unit i is inside the source image (else: zero padding)
  {
    // every plane of a descriptor has the same halo extent (api.hip: build_desc), so (hy, hx) of a unit -- two integer divisions -- are
    // computed ONCE per launch; a stage only adds its plane's origin (the per-stage divisions and 64-bit multiplies this replaces cost a lone
    // wave ~2,000 cycles per stage: half as much again as the stage's 128 MFMAs)
    const int hw0 = d.plane[0].hw, nunits = d.plane[0].hh * d.plane[0].hw * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int pix = idx >> 2, g = idx & 3;
      const int hy = pix / hw0, hx = pix - hy * hw0;
      hoff[i] = idx < nunits ? (hy * P + hx) * 80 + g * 16 : -1;
      hyS[i] = hy * d.SS; hxS[i] = hx * d.SS;
      hrel[i] = (hyS[i] * d.IW + hxS[i]) * d.in_pitch + g * 8;
    }
  }
  auto halo_load = [&](const Tile2& t, int st) {
    const int pl = st & (d.nplanes - 1), cc = d.nplanes == 4 ? (st >> 2) : st;      // nplanes is 1 or 4
    const TfcPlane& pd = d.plane[pl];
    const int Y0 = (t.a0 + pd.dy0 + t.phy * d.ph_d0) * d.SS + pd.py, X0 = (t.b0 + pd.dx0 + t.phx * d.ph_d0) * d.SS + pd.px;
    const bf16_t* base = in + ((long long)t.img * d.IH * d.IW + (long long)Y0 * d.IW + X0) * d.in_pitch + cc * 32;   // wave-uniform
    hvalid = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = hoff[i] >= 0 && (unsigned)(Y0 + hyS[i]) < (unsigned)d.IH && (unsigned)(X0 + hxS[i]) < (unsigned)d.IW;
      const bf16_t* src = ok ? base + hrel[i] : in;
      hvalid |= ok ? (1u << i) : 0u;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[i]) : "v"(src) : "memory");
    }
  };
  auto halo_store = [&](unsigned char* buf) {                    // caller has waited for the four loads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("" : "+v"(hv[i]));
      const u32x4_t z = {0u, 0u, 0u, 0u};
      const u32x4_t v = (hvalid >> i) & 1u ? hv[i] : z;
      if (hoff[i] >= 0) *reinterpret_cast<u32x4_t*>(buf + hoff[i]) = v;
    }
  };
  auto loadB = [&](const unsigned char* p, u32x4_t (&b)[NT]) {    // p: this lane's address of the fragment of n-block 0
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (IL) {
        const unsigned char* pn = p + nt * BSTEP;                 // 4096 does not fit the instruction's 13-bit signed offset
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[nt]) : "v"(pn) : "memory");
      } else {
        if (nt == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b[0]) : "v"(p) : "memory");
        if (nt == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(b[1]) : "v"(p) : "memory");
        if (nt == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(b[2]) : "v"(p) : "memory");
        if (nt == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(b[3]) : "v"(p) : "memory");
      }
    }
  };
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(unsigned char, smem);   // LDS byte address of the dynamic region

  // store pass of the epilogue: thread -> (unit column su, staged rows rho = tid / UPR + kk * 256 / UPR); element offset of (row, unit) inside a tile's
  // output window is tile-invariant
  constexpr int NSK = (128 * UPR) / 256;
  const int su = tid % UPR;
  int srel[NSK];
#pragma unroll
  for (int kk = 0; kk < NSK; ++kk) {
    const int rho = tid / UPR + kk * (256 / UPR);
    const int rr = rho & 31;
    const int ty = 2 * (rho >> 5) + (rr & 1), tx = rr >> 1;
    srel[kk] = (ty * d.OS * d.OW + tx * d.OS) * d.out_pitch + su * 8;
  }

  // ---- work distribution: static round-robin, work item of round k = k * G + r(blockIdx) (r: XCD-aware bijection, the workgroups of one XCD
  //      hold neighbouring tiles of a round).  Measured alternatives at the three big shapes, all within +-2 % of this one and therefore not
  //      kept: pulling items from per-CU-pair counters (balances the pair -- the first-dispatched workgroup of a CU wins every arbitration and
  //      finishes its equal share ~25 % earlier -- but each returning atomic sits on the in-order vmcnt queue), a half-period start stagger of the
  //      second workgroup, s_setprio schemes.
  // HALVES == 2: worker id = 2 * r(blockIdx) + half -- the two halves of a workgroup hold neighbouring tiles (shared halo columns in one L2 / L1)
  const int worker2 = 2 * tfc_xcd_remap(blockIdx.x, gridDim.x) + half;
  auto item = [&](int k) {
    if constexpr (HALVES == 2) return k * G + worker2 < nwork ? k * G + worker2 : -1;
    const int cnt = (nwork - k * G) < G ? (nwork - k * G) : G;   // the last round may be partial
    return (int)blockIdx.x < cnt ? k * G + tfc_xcd_remap(blockIdx.x, cnt) : -1;
  };
  int nbar_total = 0;
  if constexpr (HALVES == 2) {
    // barriers each half will execute: [shift dummies (half 1)] + 1 (prologue) + (nst + 1) per tile + 1 (last statistics flush); pad to the larger
    const int S = nst + 1, wa = worker2 - half, wb = wa + 1;
    const int na = wa < nwork ? (nwork - wa + G - 1) / G : 0, nb = wb < nwork ? (nwork - wb + G - 1) / G : 0;
    const int ta = na ? 2 + na * S : 0, tb = nb ? shift + 2 + nb * S : 0;
    nbar_total = ta > tb ? ta : tb;
    if ((half ? nb : na) == 0) {                                  // nothing for this half (a ragged last round): keep the other half's barriers company
      for (; nbar < nbar_total; ++nbar) __builtin_amdgcn_s_barrier();
      return;
    }
    if (half) for (int i = 0; i < shift; ++i) TFC_BAR();
  }
  int round = 0;
  int w = item(0), w_next = item(1);
  Tile2 cur, nxt;
  decode(w, cur);
  nxt = cur;
  u32x4_t br[BD][NT];
#pragma unroll
  for (int i = 0; i < BD; ++i) loadB(cur.wbase + lanepart + (size_t)i * wstep_b, br[i]);
  const unsigned char* bptr = cur.wbase + lanepart + (size_t)BD * wstep_b;   // next fragment to request (per lane)
  if (flags & TFC_EP_BIAS)
    for (int n = tid; n < nblkN * BN; n += 256) sbias[n] = n < d.Nout ? bias[n] : 0.f;
  halo_load(cur, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  halo_store(smem);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  TFC_BAR();
  TFC_STAMP_AT(1);
  int sc = 0;                                                    // running stage counter: halo buffer parity across tiles

  // InstanceNorm statistics. Deterministic (round 3): the four waves' sums of a tile go to FOUR LDS slots with plain stores (the LDS float atomics they
  // replace added in arrival order), wave 0 adds the slots in wave order and STORES the tile's 2 * BN sums into the tile's own slot of the partial buffer
  // stats = part[img][tile][Nout][2]; tfc_part_reduce_kernel adds the tiles in a fixed order behind this launch. The flush still happens one tile LATE,
  // right after the next tile's K loop: every workgroup barrier of that loop lies between the slot writes and these reads, and the stores leave the
  // wave's in-order vmcnt queue long before the epilogue drains it (the memory-side atomics of round 2 sat there for ~3,000 cycles each).
  float* stat_prev = nullptr;
  int stat_lim = 0, tcount = 0;
  auto stat_flush = [&]() {                                       // wave 0
    if (stat_prev && wave == 0) {
#pragma unroll
      for (int i = lane; i < 2 * BN; i += 64) {
        const float v = ((sstat[i] + sstat[2 * BN + i]) + sstat[4 * BN + i]) + sstat[6 * BN + i];
        if (i < stat_lim) stat_prev[i] = v;
      }
    }
  };
  for (;;) {
    const bool has_next = w_next >= 0;
    if (has_next) decode(w_next, nxt);


    f32x16_t acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mi][nt][j] = 0.f;

    TFC_NOW(tk0);
    if (prio) asm volatile("s_setprio 2");                        // K loop: this wave's MFMAs go ahead of the partner wave's epilogue VALU / LDS work
    for (int st = 0; st < nst; ++st) {
      const bool last = (st + 1) == nst;
      const bool more = !last || has_next;
      // always FOUR loads (static vmcnt counts): the next stage's halo, or -- nothing follows -- a harmless re-read that is never stored
#ifdef TFC_STAMP
      TFC_NOW(tq0);
#endif
      halo_load((last && has_next) ? nxt : cur, last ? 0 : st + 1);
#ifdef TFC_STAMP
      TFC_NOW(tq1); acc_h += tq1 - tq0;
#endif
      const unsigned abase = lds0 + (sc & 1) * bstride + laneBase;
      constexpr int Q = ROWS * NSR;                               // k-substeps of one stage, one straight-line pipeline
      u32x4_t a[IL ? 1 : 2][MT];                                  // IL: ONE set, each fragment re-requested right after its last MFMA (16 VGPRs less)
      tfc_static_for<0, MT>([&](auto mic) {
        constexpr int mi = decltype(mic)::value;
        lds_rd<(TapPat<PAT>::dy(0) * P + TapPat<PAT>::dx(0)) * 80 + mi * (2 * P * 80)>(a[0][mi], abase);
      });
#ifdef TFC_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TFC_NOW(tq2); acc_f += tq2 - tq1;
#endif
      tfc_static_for<0, Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (IL) {
          // 256-channel tile: pixel-fragment-major MFMA order (mi, nt) -- a[mi] is free after its NT MFMAs and is re-requested for k-substep
          // q + 1 into the same registers, six MFMAs ahead of its next use.  LDS returns in order: before the MFMAs of fragment mi exactly the
          // MT - 1 younger requests may be outstanding (fewer in the last k-substep of a stage, which requests nothing).
          if constexpr (q < BD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((BD - 1) * NT + 4) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((BD - 1) * NT) : "memory");
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(br[q % BD][nt]));
          if constexpr (q + BD == Q) { if (last && has_next) bptr = nxt.wbase + lanepart; }
          tfc_static_for<0, MT>([&](auto mc) {
            constexpr int mi = decltype(mc)::value;
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(q + 1 < Q ? MT - 1 : MT - 1 - mi) : "memory");
            asm volatile("" : "+v"(a[0][mi]));
            tfc_static_for<0, NT>([&](auto nc) {
              constexpr int nt = decltype(nc)::value;
              acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, br[q % BD][nt]),
                                                                    __builtin_bit_cast(bf16x8_t, a[0][mi]), acc[mi][nt], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (nt == NT - 1 && q + 1 < Q) {
                constexpr int r1 = (q + 1) / NSR, s1 = (q + 1) % NSR;
                constexpr int off = (TapPat<PAT>::dy(r1) * P + TapPat<PAT>::dx(s1 >> 1)) * 80 + (s1 & 1) * 32;
                lds_rd<off + mi * (2 * P * 80)>(a[0][mi], abase);
              }
              if constexpr (mi == MT - 1) {                       // br[q % BD][nt] has fed its last MFMA: request the fragment BD k-substeps ahead
                const unsigned char* pn = bptr + nt * BSTEP;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(br[q % BD][nt]) : "v"(pn) : "memory");
              }
              __builtin_amdgcn_sched_barrier(0);
            });
          });
          bptr += wstep_b;
        } else {
        // operands of this k-substep.  A(q) was requested during k-substep q - 1 (nothing younger on the LDS queue); of the weight loads
        // (BD - 1) * NT are younger than B(q), plus -- during the first BD k-substeps of a stage -- the four halo loads issued at its start
#ifdef TFC_ABL_LGKM0
        constexpr bool CNT_LGKM = false;
#else
        constexpr bool CNT_LGKM = NT == 1;                        // counted per-fragment LDS waits (below); measured worse for the two-fragment tiles
#endif
        if constexpr (!CNT_LGKM) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) asm volatile("" : "+v"(a[q & 1][mi]));
        }
        if constexpr (q < BD) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((BD - 1) * NT + 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((BD - 1) * NT) : "memory");
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(br[q % BD][nt]));
        if constexpr (q + BD == Q) { if (last && has_next) bptr = nxt.wbase + lanepart; }
#ifndef TFC_ABL_HS_END
        // the next stage's halo goes to LDS HERE, not at the end of the stage: the four halo loads are older than B(q) for q >= BD, so the wait above has
        // landed them; the buffer they go to was last read a stage ago (behind that stage's barrier).  The end of the stage is then a bare s_barrier instead of
        // wait -> 4 x ds_write_b128 -> lgkmcnt(0) -> barrier in series (~0.6 k cycles per stage, 15 % of a 128-MFMA stage).
        if constexpr (Q > BD && q == BD) { if (more) halo_store(smem + ((sc + 1) & 1) * bstride); }
#endif
        // One wave issues in order: memory instructions bunched behind the last MFMA of a k-substep all land in ONE 32-cycle MFMA gap and overrun
        // it (measured: ~60 idle matrix-pipe cycles per k-substep for a wave alone on its SIMD).  They are therefore dealt out over the gaps,
        // at most two per gap, and the order is pinned with sched_barrier (the MFMAs are builtins: nothing else keeps hipcc from re-bunching):
        //   after MFMA i (n-block-major order): A fragment i of the NEXT k-substep (other register set); after the last MFMA of n-block nt:
        //   the weight fragment of n-block nt BD k-substeps ahead (its register is free from here on)
        tfc_static_for<0, MT * NT>([&](auto ic) {
          constexpr int i = decltype(ic)::value, nt = i / MT, mi = i % MT;
          // COUNTED LDS wait: fragment mi of this k-substep was requested one k-substep ago (behind MFMA mi of k-substep q - 1) and LDS returns in order;
          // behind it exactly MT - 1 younger fragment reads exist (the rest of its own k-substep + the first mi of the next one) -- fewer in the last
          // k-substep of a stage, which requests nothing.  (lgkmcnt(0) at the top of the k-substep waited for the fragment requested a few cycles
          // earlier: a full LDS round trip exposed per k-substep.)  Other LDS traffic (the mid-stage halo store) is younger and only lengthens the wait.
          if constexpr (CNT_LGKM && nt == 0) {
            constexpr int per0 = (MT + MT * NT - 1) / (MT * NT);                         // fragment reads per gap (see below)
            constexpr int younger = q + 1 < Q ? (per0 == 1 ? MT - 1 : MT - per0) : MT - 1 - mi;
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(younger < 0 ? 0 : younger) : "memory");
            asm volatile("" : "+v"(a[q & 1][mi]));
          }
          acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, br[q % BD][nt]),      // swapped operands: rows =
                                                                __builtin_bit_cast(bf16x8_t, a[q & 1][mi]), acc[mi][nt], 0, 0, 0);   // channels, lanes = pixels
          __builtin_amdgcn_sched_barrier(0);
#ifndef TFC_ABL_NOA
          if constexpr (q + 1 < Q) {
            constexpr int r1 = (q + 1) / NSR, s1 = (q + 1) % NSR;
            constexpr int off = (TapPat<PAT>::dy(r1) * P + TapPat<PAT>::dx(s1 >> 1)) * 80 + (s1 & 1) * 32;
            constexpr int per = (MT + MT * NT - 1) / (MT * NT);    // A reads per gap (1 unless there are fewer MFMAs than A fragments)
            tfc_static_for<i * per, (i * per + per < MT ? i * per + per : MT)>([&](auto mc) {
              constexpr int m2 = decltype(mc)::value;
              lds_rd<off + m2 * (2 * P * 80)>(a[(q + 1) & 1][m2], abase);
            });
          }
#endif
#ifndef TFC_ABL_NOB
          if constexpr (mi == MT - 1) {
            if constexpr (nt == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(br[q % BD][0]) : "v"(bptr) : "memory");
            if constexpr (nt == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(br[q % BD][1]) : "v"(bptr) : "memory");
            if constexpr (nt == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(br[q % BD][2]) : "v"(bptr) : "memory");
            if constexpr (nt == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(br[q % BD][3]) : "v"(bptr) : "memory");
          }
#endif
          if constexpr (i == MT * NT - 1) bptr += wstep_b;
          __builtin_amdgcn_sched_barrier(0);
        });
        }
      });
      // the halo loads are older than every weight load of this stage: with at most BD * NT vector-memory operations left they have landed
#ifdef TFC_STAMP
      TFC_NOW(ts0);
#endif
#ifndef TFC_ABL_HS_END
      constexpr bool HS_MID = !IL && Q > BD;                      // halo already stored inside the K loop (see there)
#else
      constexpr bool HS_MID = false;
#endif
      if constexpr (!HS_MID) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(BD * NT) : "memory");
        if (more) halo_store(smem + ((sc + 1) & 1) * bstride);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TFC_BAR();
#ifdef TFC_STAMP
      TFC_NOW(ts1); acc_sync += ts1 - ts0;
#endif
      ++sc;
    }

    if (prio) asm volatile("s_setprio 0");
    if (flags & TFC_EP_STATS) stat_flush();
    // ---- epilogue: accumulators (channel rows x pixel lanes) -> 16-byte units -> staged tile in LDS ----
    TFC_STAMP_AT(2);
#ifdef TFC_STAMP
    TFC_NOW(tk1); acc_main += tk1 - tk0; tk0 = tk1; ++ntile;
#endif
    unsigned char* stage = IL ? smem + ((sc & 1) ? 0 : buf_bytes) : stage_fix;   // IL: beside the live halo buffer (parity sc & 1 after the last stage)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {                          // one pass per BNS staged channels (a single pass unless the tile is 256 wide)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (IL && nt != ps) continue;
      const int blk = IL ? nt * WN + wn : wn * NT + nt;           // this accumulator's 32-channel block inside the n-block ...
      const int col = IL ? wn : wn * NT + nt;                     // ... and inside the staged pass
      float4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bq[q] = (flags & TFC_EP_BIAS) ? *reinterpret_cast<const float4*>(sbias + cur.nb_blk * BN + blk * 32 + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        uint32_t pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v0 = acc[mi][nt][4 * q + 0] * osc + bq[q].x, v1 = acc[mi][nt][4 * q + 1] * osc + bq[q].y;
          float v2 = acc[mi][nt][4 * q + 2] * osc + bq[q].z, v3 = acc[mi][nt][4 * q + 3] * osc + bq[q].w;
          if (flags & TFC_EP_LEAKY) { v0 = fmaxf(v0, 0.2f * v0); v1 = fmaxf(v1, 0.2f * v1); v2 = fmaxf(v2, 0.2f * v2); v3 = fmaxf(v3, 0.2f * v3); }
          if (flags & TFC_EP_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          pk[q][0] = pack_bf16x2(v0, v1);
          pk[q][1] = pack_bf16x2(v2, v3);
        }
        const int rho = (wm * MT + mi) * 32 + r;                  // LDS row = MFMA column order (pixel: ty = 2*ms + (r & 1), tx = r >> 1)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          // lanes < 32 hold channels 8q+0..3 (q = 2pr) and want 8q+4..7 from their partner lane + 32, which in turn wants this lane's group 2pr+1
          auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
          const uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          *reinterpret_cast<uint4*>(stage + rho * ROWP + (col * 32 + pr * 16 + h * 8) * 2) = o;
        }
      }
    }
    if (ps == 0) {
      TFC_STAMP_AT(3);
#ifdef TFC_STAMP
      TFC_NOW(tk1); acc_ep1 += tk1 - tk0; tk0 = tk1;
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next tile's first weight fragments (requested ~2.5k cycles ago) land BEFORE the stores below
#pragma unroll
      for (int i = 0; i < BD; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(br[i][nt]));
    }
    TFC_SYNC();
    if (ps == 0) {
      TFC_STAMP_AT(4);
#ifdef TFC_STAMP
      TFC_NOW(tk1); acc_bar += tk1 - tk0; tk0 = tk1;
#endif
    }
    {
      const int n0 = cur.nb_blk * BN + ps * BNS + su * 8;
      const int ylim = d.GH - cur.a0, xlim = d.GW - cur.b0;
      // wave-uniform part of the output address of this tile (the per-thread part srel[k] is tile-invariant)
      bf16_t* obase = out + ((long long)(cur.img * d.OH + cur.a0 * d.OS + d.OOY + cur.phy * d.ph_oo) * d.OW + cur.b0 * d.OS + d.OOX + cur.phx * d.ph_oo) * d.out_pitch +
                      cur.nb_blk * BN + ps * BNS;
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < NSK; ++kk) {
        const int rho = tid / UPR + kk * (256 / UPR);
        const int rr = rho & 31;
        const int ty = 2 * (rho >> 5) + (rr & 1), tx = rr >> 1;
        if (ty < ylim && tx < xlim && n0 < d.Nout) {
          T* po = obase + srel[kk];
          uint4 v = *reinterpret_cast<const uint4*>(stage + rho * ROWP + su * 16);
          if (flags & (TFC_EP_ACCUM | TFC_EP_STATS)) {
            float f[8];
            unpack16<bf16_t>(v, f);
            if (flags & TFC_EP_ACCUM) {
              float g[8];
              unpack16<bf16_t>(*reinterpret_cast<const uint4*>(po), g);
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] += g[e];
              v = pack16<bf16_t>(f);
              if (flags & TFC_EP_STATS) unpack16<bf16_t>(v, f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
          }
          store_stream16(po, v);
        }
      }
      if (flags & TFC_EP_STATS) {
        // lanes with equal (lane % UPR) hold partial sums of the same 8 channels: butterfly over them (a fixed tree), then each wave stores its sums
        // into its own LDS slot; wave 0 adds the four slots in wave order one tile later (stat_flush) and writes the tile's 2 * BN sums as whole
        // 256-byte runs of part[img][tile][n][2]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 32; o >= UPR; o >>= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        if (lane < UPR) {
          float2* sp = reinterpret_cast<float2*>(sstat + wave * 2 * BN + ps * 2 * BNS) + su * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) sp[e] = make_float2(s1[e], s2[e]);
        }
        const int tpi = d.tiles_y * d.tiles_x;
        const int tile_id = (cur.phy * 2 + cur.phx) * tpi + (cur.a0 / TFC_TILE_H) * d.tiles_x + cur.b0 / TFC_TILE_W;
        stat_prev = stats + (((size_t)cur.img * (tpi * (d.ph_n > 1 ? d.ph_n : 1)) + tile_id) * d.Nout + cur.nb_blk * BN) * 2;
        stat_lim = 2 * (d.Nout - cur.nb_blk * BN);                // floats of this n-block that exist
      }
    }
    if (IL) TFC_SYNC();                                           // the staged tile is rewritten by the next pass / by the next tile's first halo store
    }
    ++tcount;
    TFC_STAMP_AT(5);
#ifdef TFC_STAMP
    TFC_NOW(tk1); acc_store += tk1 - tk0;
#endif
    if (!has_next) break;
    cur = nxt;
    ++round;
    w = w_next;
    w_next = item(round + 1);
  }
  if (HALVES == 2 || (flags & TFC_EP_STATS)) TFC_SYNC();          // (HALVES == 2: always, so that the barrier count does not depend on the flags)
  if (flags & TFC_EP_STATS) stat_flush();                         // the last tile's sums
  if constexpr (HALVES == 2)
    for (; nbar < nbar_total; ++nbar) __builtin_amdgcn_s_barrier();
#ifdef TFC_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFC_STAMP_AT(6);
  if (lane == 0 && dbg) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(dbg) + ((size_t)blockIdx.x * 4 + wave) * 16;
    o[2] = acc_main; o[3] = acc_ep1; o[4] = acc_bar; o[5] = acc_store; o[7] = ntile; o[1] = acc_sync; o[8] = acc_h; o[9] = acc_f;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[6] = ((unsigned long long)xcc << 32) | hwid;
    o[0] = __builtin_amdgcn_s_memtime() - mt_begin;               // wave lifetime in shader cycles ...
    o[6] = __builtin_amdgcn_s_memrealtime() - rt_begin;            // ... and in 100 MHz ticks (replaces the placement word)
  }
#endif
#undef TFC_NOW
#undef TFC_BAR
#undef TFC_SYNC
}

// ---------------------------------------------------------------------------------------------------
// First-layer convolution (8 padded input channels x 16 taps: K = 128 = 8 k-substeps; G down1 3->64 and D block 1 6->64 at
// 256 x 256): the generic kernel spends its life in prologue / epilogue here -- per 16 KB of output it re-reads 32 KB of weight
// fragments and pays a workgroup launch. This variant is weights-stationary and persistent: a wave keeps its eight B fragments in
// 32 VGPRs for the whole launch, a workgroup walks tiles  b, b + grid, ...; the halo of the tile after next is requested
// BEFORE the current tile's stores are issued (vmcnt retires in order: a load issued behind stores would have to wait for them),
// so nothing in the steady state ever waits for a store. Bound: the 266 MB output stream (HBM).
// ---------------------------------------------------------------------------------------------------
template <int LANE> __device__ __forceinline__ unsigned tfc_writelane(unsigned v, unsigned old) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(LANE));   // lane LANE of the result <- the wave-uniform v (one SGPR operand: constant bus)
  return old;
#else
  (void)v; return old;
#endif
}
// MASK: also leave one sign bit per stored value (value > 0) -- 8 bytes per pixel, all that the fused backward of the block reads of this 266 MB tensor
// (tfc_wgrad_c8_fusedm_kernel). A lane of the accumulator holds ONE channel of 16 pixels, so the comparison of a register across the wave IS the sign
// word: v_cmp writes a 64-bit lane mask = the 32 channels of this wave for two pixels; v_writelane files it under the pixel's lane, one 4-byte LDS
// store per lane and tile, and 128 threads of the store pass write the tile's 1 KB of sign words. (Deriving the bits in the store pass -- unpack,
// eight compares, a byte or a butterfly per unit -- cost 26-40 us per launch: the kernel lives at 128 VGPRs and four workgroups per CU.)
template <bool MASK>
__global__ void __launch_bounds__(256, 4)
tfc_conv_c8_kernel(const TfcGather d, const bf16_t* __restrict__ in, const uint4* __restrict__ wp, bf16_t* __restrict__ out,
                   const float* __restrict__ bias, const float* __restrict__ oscale, int leaky, int NB32, int nwork, unsigned char* __restrict__ sign_mask) {
  constexpr int P = TFC_LDS_P, PS = 16, MT = 2;
  constexpr int HB = TFC_MAX_HH * P * PS;                        // one halo buffer (4224 B)
  constexpr int ROWP = 64 * 2 + 16;                              // staged output tile: 64 channels per pixel row + pad
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HB + 128 * ROWP + (MASK ? 128 * 8 : 0)];
  unsigned char* stage = smem + 2 * HB;
  unsigned* smask = reinterpret_cast<unsigned*>(smem + 2 * HB + 128 * ROWP);   // [128 pixels][2 channel halves]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r = lane & 31;
  const TfcPlane& pd = d.plane[0];
  const int G = gridDim.x;

  uint4 bw[8];                                                   // this wave's 32 output channels x K = 128, for the whole launch
#pragma unroll
  for (int s = 0; s < 8; ++s) bw[s] = wp[((size_t)s * NB32 + wn) * 64 + lane];
  const int n = wn * 32 + r;
  const float bv = (bias && n < d.Nout) ? bias[n] : 0.f;
  const float osc = oscale ? *oscale : 1.f;

  auto decode = [&](int w, int& img, int& a0, int& b0) {
    int tile = tfc_xcd_remap(w, nwork);
    const int txb = tile % d.tiles_x; tile /= d.tiles_x;
    const int tyb = tile % d.tiles_y;
    img = tile / d.tiles_y;
    a0 = tyb * TFC_TILE_H; b0 = txb * TFC_TILE_W;
  };
  const int hpix = tid;                                          // one 16-byte halo unit (= one pixel) per thread
  const int hy = hpix / pd.hw, hx = hpix - hy * pd.hw;
  const bool hact = hpix < pd.hh * pd.hw;
  const int hoff = (hy * P + hx) * PS;
  auto halo_load = [&](int img, int a0, int b0) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    const int y = a0 + pd.dy0 + hy, x = b0 + pd.dx0 + hx;
    if (hact && y >= 0 && y < d.IH && x >= 0 && x < d.IW)
      v = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch);
    return v;
  };
  const int laneBase = ((2 * wm * MT + (r & 1)) * P + (r >> 1)) * PS + h * PS;   // lane half h takes the odd tap of a k-substep
  constexpr int MSTRIDE = 2 * P * PS;

  int w = blockIdx.x;
  int img, a0, b0, n_img = 0, n_a0 = 0, n_b0 = 0;
  decode(w, img, a0, b0);
  uint4 hv = halo_load(img, a0, b0);
  if (hact) *reinterpret_cast<uint4*>(smem + hoff) = hv;
  if (w + G < nwork) { decode(w + G, n_img, n_a0, n_b0); hv = halo_load(n_img, n_a0, n_b0); }
  __syncthreads();

  for (int k = 0;; ++k) {
    const unsigned char* buf = smem + (k & 1) * HB + laneBase;
    f32x16_t acc[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][j] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {                                // k-substep s = taps 2s (h = 0) and 2s + 1 (h = 1) of the 4 x 4 raster
      const int off = ((s >> 1) * P + 2 * (s & 1)) * PS;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const uint4 a = *reinterpret_cast<const uint4*>(buf + off + mi * MSTRIDE);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[s]), acc[mi], 0, 0, 0);
      }
    }
    unsigned mw = 0;                                              // MASK: lane (mi * 32 + row) <- sign word of this wave's 32 channels for that pixel
    tfc_static_for<0, MT>([&](auto mic) {
      constexpr int mi = decltype(mic)::value;
      tfc_static_for<0, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
        const int ty = 2 * (wm * MT + mi) + (row & 1), tx = row >> 1;
        float v = acc[mi][j] * osc + bv;
        if (leaky) v = fmaxf(v, 0.2f * v);
        const bf16_t hb = f32_to_bf16(v);
        *reinterpret_cast<bf16_t*>(stage + (ty * TFC_TILE_W + tx) * ROWP + n * 2) = hb;
        if constexpr (MASK) {
          const unsigned long long bal = __ballot((short)hb > 0);  // bf16 is sign-magnitude: > 0 <=> the bits as a signed integer > 0
          constexpr int r0 = (j & 3) + 8 * (j >> 2);               // the pixel rows of the two half-waves: r0 (lanes 0..31), r0 + 4 (lanes 32..63)
          mw = tfc_writelane<mi * 32 + r0>((unsigned)bal, mw);
          mw = tfc_writelane<mi * 32 + r0 + 4>((unsigned)(bal >> 32), mw);
        }
      });
    });
    if constexpr (MASK) {
      const int mrow = lane & 31;
      const int mty = 2 * (wm * MT + (lane >> 5)) + (mrow & 1), mtx = mrow >> 1;
      smask[(mty * TFC_TILE_W + mtx) * 2 + wn] = mw;
    }
    const bool has1 = (w + G) < nwork;
    if (has1 && hact) *reinterpret_cast<uint4*>(smem + ((k + 1) & 1) * HB + hoff) = hv;
    int nn_img = 0, nn_a0 = 0, nn_b0 = 0;
    if (w + 2 * G < nwork) { decode(w + 2 * G, nn_img, nn_a0, nn_b0); hv = halo_load(nn_img, nn_a0, nn_b0); }   // before the stores below
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int pix = idx >> 3, u = idx & 7;
      const int a = a0 + (pix >> 4), b = b0 + (pix & 15);
      if (a < d.GH && b < d.GW && u * 8 < d.Nout) {
        const uint4 v = *reinterpret_cast<const uint4*>(stage + pix * ROWP + u * 16);
        store_stream16(out + ((size_t)(img * d.OH + a + d.OOY) * d.OW + b + d.OOX) * d.out_pitch + u * 8, v);
      }
    }
    if constexpr (MASK) {
      if (tid < 128) {
        const int a = a0 + (tid >> 4), b = b0 + (tid & 15);
        if (a < d.GH && b < d.GW)
          *reinterpret_cast<uint2*>(sign_mask + ((size_t)(img * d.OH + a) * d.OW + b) * 8) = *reinterpret_cast<const uint2*>(smask + tid * 2);
      }
    }
    if (!has1) break;
    __syncthreads();                                             // the staged tile is free again
    w += G; img = n_img; a0 = n_a0; b0 = n_b0; n_img = nn_img; n_a0 = nn_a0; n_b0 = nn_b0;
  }
}

// ---------------------------------------------------------------------------------------------------
// Generator head (P16:150-157: Upsample(x2 nearest) -> ZeroPad2d(1,0,1,0) -> Conv2d(128, C<=4, 4, padding=1) -> Tanh), forward, bf16.
// The gather-GEMM spends a 32-wide MFMA tile on 3 output channels, once per sub-pixel phase. Here the four phases of an INPUT pixel
// share its 3 x 3 neighbourhood, so they become COLUMNS of one 16-wide tile:  col = phase * 4 + oc  (12 of 16 used),
//   out[2a+py][2b+px][oc] = tanh(bias[oc] + sum_{r,c,ci} x[a-1+r][b-1+c][ci] * Wc[py,px][r][c][ci][oc]),
// Wc = the filter taps that collapse onto source offset (r,c) in that phase, summed (zero where a phase lacks the offset).
// K = 9 offsets x 128 channels = 36 steps of v_mfma_f32_16x16x32_bf16, split over the workgroup's four waves BY CHANNEL CHUNK: wave w
// owns channels 32w..32w+31 for the whole 8 x 16 tile, so its nine B fragments (built once from the fp32 filter, weights-stationary,
// persistent workgroups) cost 36 VGPRs, and its eight 16 x 16 accumulators are partial sums that meet in LDS. The whole
// 10 x 18 x 128 halo of a tile is staged in LDS (pixel stride 256 + 16 B); the next tile's halo is requested before this tile's
// stores are issued, so the wait for it never includes them (vmcnt retires in order). Output: tanh, fp32 NCHW, 8-byte stores that
// form whole 128-byte row segments per 16 lanes.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_upconv_head_kernel(const bf16_t* __restrict__ x, int IH, int IW, int x_pitch, const float* __restrict__ w, const float* __restrict__ bias,
                       int Cout, float* __restrict__ out, int nimg, int nwork) {
  constexpr int HH = TFC_TILE_H + 2, HW = TFC_TILE_W + 2, NPIX = HH * HW;      // 10 x 18 halo
  constexpr int PS = 256 + 16;                                   // LDS bytes per halo pixel: 128 channels + pad (conflict-free 16-lane reads)
  constexpr int NHV = (NPIX * 16 + 255) / 256;                   // 16-byte halo units per thread (12)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* part = reinterpret_cast<float*>(smem);                  // [4 waves][8 ty][16 tx][16 cols] partial sums: reuses the halo bytes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int ph = col >> 2, oc = col & 3, py = ph >> 1, px = ph & 1;
  const int G = gridDim.x;
  const int tiles_y = (IH + TFC_TILE_H - 1) / TFC_TILE_H, tiles_x = (IW + TFC_TILE_W - 1) / TFC_TILE_W;
  const int OH = 2 * IH, OW = 2 * IW;

  // ---- B fragments of this wave's channel chunk: bw[r * 3 + c], element j <-> ci = wave * 32 + kg * 8 + j ----
  uint4 bw[9];
  {
    const bool real = oc < Cout;
    float f[9][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = wave * 32 + kg * 8 + j;
      float t[16];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 v = real ? *reinterpret_cast<const float4*>(w + ((size_t)oc * 128 + ci) * 16 + q4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        t[q4 * 4 + 0] = v.x; t[q4 * 4 + 1] = v.y; t[q4 * 4 + 2] = v.z; t[q4 * 4 + 3] = v.w;
      }
      // filter rows ky -> source row r: py = 0: {0,1} -> 0, {2,3} -> 1;  py = 1: {0} -> 0, {1,2} -> 1, {3} -> 2   (same for columns)
      float rr[3][4];
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        rr[0][kx] = py ? t[kx] : t[kx] + t[4 + kx];
        rr[1][kx] = py ? t[4 + kx] + t[8 + kx] : t[8 + kx] + t[12 + kx];
        rr[2][kx] = py ? t[12 + kx] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        f[r * 3 + 0][j] = px ? rr[r][0] : rr[r][0] + rr[r][1];
        f[r * 3 + 1][j] = px ? rr[r][1] + rr[r][2] : rr[r][2] + rr[r][3];
        f[r * 3 + 2][j] = px ? rr[r][3] : 0.f;
      }
    }
#pragma unroll
    for (int o = 0; o < 9; ++o) bw[o] = pack16<bf16_t>(f[o]);
  }

  auto decode = [&](int wk, int& img, int& a0, int& b0) {
    int tile = tfc_xcd_remap(wk, nwork);
    const int txb = tile % tiles_x; tile /= tiles_x;
    const int tyb = tile % tiles_y;
    img = tile / tiles_y;
    a0 = tyb * TFC_TILE_H; b0 = txb * TFC_TILE_W;
  };
  uint4 hv[NHV];
  auto halo_load = [&](int img, int a0, int b0) {
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int idx = tid + i * 256;                             // pixel * 16 + unit
      const int pix = idx >> 4, u = idx & 15;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int y = a0 - 1 + hy, xx = b0 - 1 + hx;
      hv[i] = make_uint4(0, 0, 0, 0);
      if (pix < NPIX && y >= 0 && y < IH && xx >= 0 && xx < IW)
        hv[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(img * IH + y) * IW + xx) * x_pitch + u * 8);
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int idx = tid + i * 256;
      if (idx < NPIX * 16) *reinterpret_cast<uint4*>(smem + (idx >> 4) * PS + (idx & 15) * 16) = hv[i];
    }
  };
  // reduction / store role of this thread: output pixel pair (2*rty + rh, 2*rtx .. 2*rtx+1), i.e. columns rh*8 .. rh*8+7 = phases (rh,0), (rh,1)
  const int rty = tid >> 5, rtx = (tid >> 1) & 15, rh = tid & 1;
  float bz[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bz[c] = (bias && c < Cout) ? bias[c] : 0.f;

  int wk = blockIdx.x;
  int img, a0, b0;
  decode(wk, img, a0, b0);
  halo_load(img, a0, b0);
  halo_store();
  __syncthreads();

  for (;;) {
    f32x4_t acc[8];
#pragma unroll
    for (int ty = 0; ty < 8; ++ty) acc[ty] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // A fragment: lane (row = lane & 15 -> tile column tx, kg) reads 16 B of pixel (ty + r, tx + c), channels wave*32 + kg*8 ..
    const unsigned char* abase = smem + col * PS + wave * 64 + kg * 16;
#pragma unroll
    for (int ty = 0; ty < 8; ++ty)
#pragma unroll
      for (int o = 0; o < 9; ++o) {
        const uint4 a = *reinterpret_cast<const uint4*>(abase + ((ty + o / 3) * HW + o % 3) * PS);
        acc[ty] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[o]), acc[ty], 0, 0, 0);
      }
    const bool has1 = (wk + G) < nwork;
    __syncthreads();                                             // every wave is done with this tile's halo: its bytes become the partial sums
    int n_img = 0, n_a0 = 0, n_b0 = 0;
    if (has1) { decode(wk + G, n_img, n_a0, n_b0); halo_load(n_img, n_a0, n_b0); }   // requested BEFORE the stores below (vmcnt is in order)
    // D layout: col = lane & 15 = (phase, oc); row = 4 * (lane >> 4) + j = tx
#pragma unroll
    for (int ty = 0; ty < 8; ++ty)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[((wave * 8 + ty) * 16 + 4 * kg + j) * 16 + col] = acc[ty][j];
    __syncthreads();
    {
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const float4* pp = reinterpret_cast<const float4*>(part + ((wv * 8 + rty) * 16 + rtx) * 16 + rh * 8);
        const float4 u0 = pp[0], u1 = pp[1];
        s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
        s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
      }
      const float v0[4] = {s0.x, s0.y, s0.z, s0.w}, v1[4] = {s1.x, s1.y, s1.z, s1.w};   // px = 0 / px = 1, channel oc
      const int oy = 2 * (a0 + rty) + rh, ox = 2 * (b0 + rtx);
      if (oy < OH && ox < OW) {
        for (int c = 0; c < Cout; ++c) {
          float* dst = out + (((size_t)img * Cout + c) * OH + oy) * OW + ox;
          const float e0 = tanhf(v0[c] + bz[c]), e1 = tanhf(v1[c] + bz[c]);
          if (ox + 1 < OW && (OW & 1) == 0) *reinterpret_cast<float2*>(dst) = make_float2(e0, e1);
          else { dst[0] = e0; if (ox + 1 < OW) dst[1] = e1; }
        }
      }
    }
    if (!has1) break;
    __syncthreads();                                             // partial sums consumed: the bytes become the next halo
    halo_store();
    __syncthreads();
    wk += G; img = n_img; a0 = n_a0; b0 = n_b0;
  }
}
// ---------------------------------------------------------------------------------------------------
// Input gradient of discriminator block 1 (P16:188: SN-Conv2d(6, 64, 4, 1, 1)) w.r.t. the generated image, bf16:
//   dx[iy][ix][ci] = (1/sigma) * sum_{ky,kx,co} dy[iy+1-ky][ix+1-kx][co] * W[co][ci][ky][kx]      for ci < NC <= 4
// -- a GEMM with 3 useful columns. Four consecutive OUTPUT ROWS are packed into the columns of one 16-wide MFMA tile:
//   col = delta * 4 + ci,  row = 16 consecutive ix,  K = (7 row offsets x 4 column offsets) x 64 co = 56 steps of 16x16x32,
// where offset r' of the seven uses filter row ky = 3 - (r' - delta) (zero outside 0..3). The 56 k-steps are split over the four
// waves (co chunk x offset half): 14 register-resident B fragments per wave, built once from the fp32 filter; partial sums meet in
// LDS. Tile = 8 rows x 32 columns (4 MFMA subtiles), halo 11 x 35 pixels x 64 co in LDS; persistent workgroups, next halo requested
// before the stores. Output: fp32 NCHW, NC channels (the layout the loss gradients are added in -- no NHWC round trip).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_dgrad_rows4_kernel(const bf16_t* __restrict__ dy, int OHs, int OWs, int dy_pitch, const float* __restrict__ w, int Cin, const float* __restrict__ oscale,
                       int NC, float* __restrict__ dx, int IH, int IW, int nimg, int nwork) {
  constexpr int TR = 8, TC = 32, HH = TR + 3, HW = TC + 3, NPIX = HH * HW;      // 11 x 35 halo of dy
  constexpr int PS = 128 + 16;                                   // LDS bytes per halo pixel: 64 co + pad (conflict-free 16-lane reads)
  constexpr int NHV = (NPIX * 8 + 255) / 256;                    // 16-byte halo units per thread (13)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* part = reinterpret_cast<float*>(smem);                  // [4 waves][4 subtiles][16 rows][16 cols]: reuses the halo bytes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int delta = col >> 2, ci = col & 3;
  const int chunk = wave & 1, ohalf = wave >> 1;                 // this wave: co = chunk*32 .., offsets ohalf*14 .. ohalf*14+13
  const int G = gridDim.x;
  const int tiles_y = (IH + TR - 1) / TR, tiles_x = (IW + TC - 1) / TC;

  uint4 bw[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int o = ohalf * 14 + i;                                // offset (r', s) = (o / 4, o % 4)
    const int rp = o >> 2, sx = o & 3;
    const int ky = 3 - (rp - delta), kx = 3 - sx;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = chunk * 32 + kg * 8 + j;
      f[j] = (ky >= 0 && ky <= 3 && ci < NC) ? w[((size_t)co * Cin + ci) * 16 + ky * 4 + kx] : 0.f;
    }
    bw[i] = pack16<bf16_t>(f);
  }
  const float osc = oscale ? *oscale : 1.f;

  auto decode = [&](int wk, int& img, int& a0, int& b0) {
    int tile = tfc_xcd_remap(wk, nwork);
    const int txb = tile % tiles_x; tile /= tiles_x;
    const int tyb = tile % tiles_y;
    img = tile / tiles_y;
    a0 = tyb * TR; b0 = txb * TC;
  };
  uint4 hv[NHV];
  auto halo_load = [&](int img, int a0, int b0) {
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int idx = tid + i * 256;                             // pixel * 8 + unit
      const int pix = idx >> 3, u = idx & 7;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int y = a0 - 2 + hy, xx = b0 - 2 + hx;
      hv[i] = make_uint4(0, 0, 0, 0);
      if (pix < NPIX && y >= 0 && y < OHs && xx >= 0 && xx < OWs)
        hv[i] = *reinterpret_cast<const uint4*>(dy + ((size_t)(img * OHs + y) * OWs + xx) * dy_pitch + u * 8);
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int idx = tid + i * 256;
      if (idx < NPIX * 8) *reinterpret_cast<uint4*>(smem + (idx >> 3) * PS + (idx & 7) * 16) = hv[i];
    }
  };
  // reduction / store role: output pixel (row rrow of 8, column rcol of 32) of the tile, all NC channels
  const int rrow = tid >> 5, rcol = tid & 31;

  int wk = blockIdx.x;
  int img, a0, b0;
  decode(wk, img, a0, b0);
  halo_load(img, a0, b0);
  halo_store();
  __syncthreads();

  for (;;) {
    f32x4_t acc[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) acc[st] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // subtile st = (qy, xg): output rows 4*qy .. 4*qy+3, columns 16*xg .. +15; A fragment of offset (r', s): halo pixel (4*qy + r', 16*xg + row + s)
    const unsigned char* abase = smem + col * PS + chunk * 64 + kg * 16;
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        const int hp = (4 * (st >> 1)) * HW + 16 * (st & 1);     // compile-time part; the offset part depends on the wave's half
        const int o = ohalf * 14 + i;
        const uint4 a = *reinterpret_cast<const uint4*>(abase + (hp + (o >> 2) * HW + (o & 3)) * PS);
        acc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[i]), acc[st], 0, 0, 0);
      }
    const bool has1 = (wk + G) < nwork;
    __syncthreads();                                             // halo consumed: its bytes become the partial sums
    int n_img = 0, n_a0 = 0, n_b0 = 0;
    if (has1) { decode(wk + G, n_img, n_a0, n_b0); halo_load(n_img, n_a0, n_b0); }   // requested BEFORE the stores below (vmcnt is in order)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[((wave * 4 + st) * 16 + 4 * kg + j) * 16 + col] = acc[st][j];
    __syncthreads();
    {
      const int st = (rrow >> 2) * 2 + (rcol >> 4);
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        const float4 u = *reinterpret_cast<const float4*>(part + ((wv * 4 + st) * 16 + (rcol & 15)) * 16 + (rrow & 3) * 4);
        sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w;
      }
      const float v[4] = {sum.x, sum.y, sum.z, sum.w};
      const int iy = a0 + rrow, ix = b0 + rcol;
      if (iy < IH && ix < IW)
        for (int c = 0; c < NC; ++c) dx[(((size_t)img * NC + c) * IH + iy) * IW + ix] = v[c] * osc;
    }
    if (!has1) break;
    __syncthreads();                                             // partial sums consumed: the bytes become the next halo
    halo_store();
    __syncthreads();
    wk += G; img = n_img; a0 = n_a0; b0 = n_b0;
  }
}
hipError_t tfc_launch_dgrad_rows4(const void* dy, int dy_pitch, int N, int H, int W, const float* w, int Cin, const float* oscale, int NC, float* dx,
                                  hipStream_t st) {
  const int lds = 11 * 35 * (128 + 16);                          // 55,440 B >= the 16 KiB of partial sums that reuse it
  static int grid_cap = 0;
  if (!grid_cap) {
    int occ = 0, dev = 0, ncu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_dgrad_rows4_kernel, 256, (size_t)lds);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    grid_cap = (occ < 1 ? 1 : occ) * ncu;
  }
  const int nwork = N * ((H + 7) / 8) * ((W + 31) / 32);
  TFC_LAUNCH(tfc_dgrad_rows4_kernel, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), lds, st, (const bf16_t*)dy, H - 1, W - 1, dy_pitch, w,
                     Cin, oscale, NC, dx, H, W, N, nwork);
  return hipGetLastError();
}

hipError_t tfc_launch_upconv_head(const void* x, int x_pitch, int N, int H, int W, const float* w, const float* bias, int Cout, float* out,
                                  hipStream_t st) {
  const int lds = (TFC_TILE_H + 2) * (TFC_TILE_W + 2) * (256 + 16);   // 48,960 B >= the 32 KiB of partial sums that reuse it
  static int grid_cap = 0;
  if (!grid_cap) {
    int occ = 0, dev = 0, ncu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_upconv_head_kernel, 256, (size_t)lds);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    grid_cap = (occ < 1 ? 1 : occ) * ncu;
  }
  const int nwork = N * ((H + TFC_TILE_H - 1) / TFC_TILE_H) * ((W + TFC_TILE_W - 1) / TFC_TILE_W);
  TFC_LAUNCH(tfc_upconv_head_kernel, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), lds, st, (const bf16_t*)x, H, W, x_pitch, w, bias,
                     Cout, out, N, nwork);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Input gradient of the generator head (Upsample x2 -> ZeroPad(1,0,1,0) -> Conv2d(128, C <= 8, 4, padding=1); P16:150-157), bf16:
//   dx[a][b][ci] = sum_{m,n = 0..4} sum_oc dy[2a-1+m][2b-1+n][oc] * Wd[m][n][oc][ci],   Wd[m][n] = sum_{ky in S(m), kx in S(n)} w[oc][ci][ky][kx],
//   S = {3}, {2,3}, {1,2}, {0,1}, {0}   (the filter taps whose up-sampled source row is a, seen from dy row 2a-1+m).
// The generic kernel ran this on 16-byte channel chunks without a compile-time tap pattern (114 us); its bound is the 134 MB dx stream (~40 us).
// Here: K = 25 positions x 8 padded channels = 13 k-substeps of two positions; the collapsed filter is built ONCE per wave from the fp32
// weights as register-resident B fragments (weights-stationary, persistent workgroups, like tfc_conv_c8_kernel); the (2*8+3) x (2*16+3) halo
// of dy is staged in LDS (16 B per pixel) and read at stride-2 pixel addresses; 128 pixels x 128 channels per workgroup, wave = 128 pixels x 32 channels.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_dgrad_head_kernel(const bf16_t* __restrict__ dy, int dy_pitch, const float* __restrict__ w, int Cout, bf16_t* __restrict__ dx, int dx_pitch,
                      int IH, int IW, int nimg, int nwork) {
  constexpr int HR = 2 * TFC_TILE_H + 3, HC = 2 * TFC_TILE_W + 3, P = 36;   // 19 x 35 halo of dy, LDS pitch 36 pixels
  constexpr int HB = HR * P * 16;
  constexpr int ROWP = 128 * 2 + 16;                             // staged output tile: 128 channels per pixel row + pad
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HB + 128 * ROWP];
  unsigned char* stage = smem + 2 * HB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;                                           // this wave: input channels 32 wn .. 32 wn + 31, all 128 pixels (4 m-tiles)
  const int h = lane >> 5, r = lane & 31;
  const int G = gridDim.x;
  const int OH = 2 * IH, OW = 2 * IW;
  const int tiles_y = (IH + TFC_TILE_H - 1) / TFC_TILE_H, tiles_x = (IW + TFC_TILE_W - 1) / TFC_TILE_W;

  // ---- B fragments: bw[s][nt], element e = oc of position t = 2s + h, column ci = (wn * 2 + nt) * 32 + r ----
  uint4 bw[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    const int t = 2 * s + h, m = t / 5, n = t - 5 * m;
    // S(m): filter rows ky in [ylo, yhi]
    const int ylo = m == 0 ? 3 : (m == 4 ? 0 : 3 - m), yhi = m == 0 ? 3 : (m == 4 ? 0 : 4 - m);
    const int xlo = n == 0 ? 3 : (n == 4 ? 0 : 3 - n), xhi = n == 0 ? 3 : (n == 4 ? 0 : 4 - n);
    {
      const int ci = wn * 32 + r;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float acc = 0.f;
        if (e < Cout && t < 25)
          for (int ky = ylo; ky <= yhi; ++ky)
            for (int kx = xlo; kx <= xhi; ++kx) acc += w[((size_t)e * 128 + ci) * 16 + ky * 4 + kx];
        f[e] = acc;
      }
      bw[s] = pack16<bf16_t>(f);
    }
  }
  // per-lane LDS offset of position t = 2s + h relative to the lane's pixel (compile-time per (s, h); selected once)
  int offs[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    const int t0 = 2 * s, t1 = 2 * s + 1 < 25 ? 2 * s + 1 : 24;  // slot 25 carries zero weights: any valid address
    const int o0 = ((t0 / 5) * P + t0 % 5) * 16, o1 = ((t1 / 5) * P + t1 % 5) * 16;
    offs[s] = h ? o1 : o0;
  }
  auto decode = [&](int wk, int& img, int& a0, int& b0) {
    int tile = tfc_xcd_remap(wk, nwork);
    const int txb = tile % tiles_x; tile /= tiles_x;
    const int tyb = tile % tiles_y;
    img = tile / tiles_y;
    a0 = tyb * TFC_TILE_H; b0 = txb * TFC_TILE_W;
  };
  constexpr int NHV = (HR * HC + 255) / 256;                     // halo pixels per thread (3)
  int hoff[NHV], hy[NHV], hx[NHV];
#pragma unroll
  for (int i = 0; i < NHV; ++i) {
    const int hp = tid + i * 256;
    hy[i] = hp / HC; hx[i] = hp - hy[i] * HC;
    hoff[i] = hp < HR * HC ? (hy[i] * P + hx[i]) * 16 : -1;
  }
  uint4 hv[NHV];
  auto halo_load = [&](int img, int a0, int b0) {
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
      const int y = 2 * a0 - 1 + hy[i], x = 2 * b0 - 1 + hx[i];
      hv[i] = make_uint4(0, 0, 0, 0);
      if (hoff[i] >= 0 && y >= 0 && y < OH && x >= 0 && x < OW) hv[i] = *reinterpret_cast<const uint4*>(dy + ((size_t)(img * OH + y) * OW + x) * dy_pitch);
    }
  };
  auto halo_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NHV; ++i)
      if (hoff[i] >= 0) *reinterpret_cast<uint4*>(buf + hoff[i]) = hv[i];
  };
  // lane pixel of m-tile ms: ty = 2 * ms + (r & 1), tx = r >> 1 -> halo pixel (2 ty, 2 tx)
  const int laneBase = ((2 * (r & 1)) * P + 2 * (r >> 1)) * 16;
  constexpr int MSTRIDE = 4 * P * 16;                            // next m-tile: ty + 2 -> 4 halo rows

  int wk = blockIdx.x;
  if (wk >= nwork) return;
  int img, a0, b0;
  decode(wk, img, a0, b0);
  halo_load(img, a0, b0);
  halo_store(smem);
  __syncthreads();
  for (int k = 0;; ++k) {
    const bool more = wk + G < nwork;
    int n_img = 0, n_a0 = 0, n_b0 = 0;
    if (more) { decode(wk + G, n_img, n_a0, n_b0); halo_load(n_img, n_a0, n_b0); }
    const unsigned char* buf = smem + (k & 1) * HB + laneBase;
    f32x16_t acc[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][j] = 0.f;
#pragma unroll
    for (int s = 0; s < 13; ++s) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const uint4 a = *reinterpret_cast<const uint4*>(buf + offs[s] + mi * MSTRIDE);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw[s]), acc[mi], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
        const int ty = 2 * mi + (row & 1), tx = row >> 1;
        *reinterpret_cast<bf16_t*>(stage + (ty * TFC_TILE_W + tx) * ROWP + (wn * 32 + r) * 2) = f32_to_bf16(acc[mi][j]);
      }
    if (more) halo_store(smem + ((k + 1) & 1) * HB);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + i * 256;
      const int pix = idx >> 4, u = idx & 15;
      const int a = a0 + (pix >> 4), b = b0 + (pix & 15);
      if (a < IH && b < IW) {
        const uint4 v = *reinterpret_cast<const uint4*>(stage + pix * ROWP + u * 16);
        store_stream16(dx + ((size_t)(img * IH + a) * IW + b) * dx_pitch + u * 8, v);
      }
    }
    if (!more) break;
    __syncthreads();
    wk += G; img = n_img; a0 = n_a0; b0 = n_b0;
  }
}
hipError_t tfc_launch_dgrad_head(const void* dy, int dy_pitch, int N, int H, int W, const float* w, int Cout, void* dx, int dx_pitch, hipStream_t st) {
  static int grid_cap = 0;
  if (!grid_cap) {
    int occ = 0, dev = 0, ncu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_dgrad_head_kernel, 256, 0);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    grid_cap = (occ < 1 ? 1 : occ) * ncu;
  }
  const int nwork = N * ((H + TFC_TILE_H - 1) / TFC_TILE_H) * ((W + TFC_TILE_W - 1) / TFC_TILE_W);
  TFC_LAUNCH(tfc_dgrad_head_kernel, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), 0, st, (const bf16_t*)dy, dy_pitch, w, Cout, (bf16_t*)dx, dx_pitch, H, W, N,
             nwork);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// weight-gradient GEMM for 2 x 2-tap planes (bf16): the sub-pixel phases of the transposed convolution and phase (0,0) of the
// upsample-conv. With only four taps the generic kernel gives each wave ONE tap (3 transposing LDS reads per MFMA); here a
// workgroup owns 64 n x 64 c and wave (nh, ch) owns the 32 n x 32 c quadrant for ALL four taps: the B fragments of halo rows
// kt and kt+1 (column shifts 0 / 1) form a sliding register window, so a k-step costs 1 A + 2 new B fragments for 4 MFMAs
// (1.5 reads per MFMA). Accumulator index = tap_dy * 2 + tap_dx (geometry); the flush maps it back to the filter taps.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, 2)
tfc_wgrad22_kernel(const TfcGather d, const T* __restrict__ dO, const T* __restrict__ in, float* dwacc, float4* slab,
                   int Nn_pad, int Nn_real, int Cw_real, int nbw, int ncb2, int nsplit) {
  static_assert(sizeof(T) == 2, "bf16 only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWB = 64;                                       // 32 channels x 2 bytes per LDS pixel row
  constexpr int DO_BYTES = 2 * 128 * ROWB;
  constexpr int NDO = 4, NHA = 5;                                // 16-byte units per thread: dO 1024, halo <= 2 * 9 * 17 * 4 = 1224
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nh = wave & 1, ch = wave >> 1;
  const TfcPlane& pd = d.plane[0];
  const int HP = pd.hh * pd.hw * ROWB;                           // bytes of one 32-channel halo plane
  const int BUF_BYTES = DO_BYTES + 2 * HP;

  const int bid = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % (nbw * ncb2);
  const int sp = bid / (nbw * ncb2);
  const int cb = pair % ncb2, nb = pair / ncb2;
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x;

  f32x16_t acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;

  uint4 vdo[NDO], vha[NHA];
  const int nunits = pd.hh * pd.hw * 4;                          // per halo plane
  auto tile_load = [&](int tl) {
    int t = tl;
    const int txb = t % d.tiles_x; t /= d.tiles_x;
    const int tyb = t % d.tiles_y;
    const int img = t / d.tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < NDO; ++i) {
      const int idx = tid + i * 256;
      const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
      const int a = a0 + (px >> 4), b = b0 + (px & 15);
      const int n0 = nb * 64 + ni * 32 + g * 8;
      vdo[i] = make_uint4(0, 0, 0, 0);
      if (a < d.GH && b < d.GW && n0 < Nn_pad) {
        const int oy = a * d.OS + d.OOY, ox = b * d.OS + d.OOX;
        vdo[i] = *reinterpret_cast<const uint4*>(dO + ((size_t)(img * d.OH + oy) * d.OW + ox) * d.out_pitch + n0);
      }
    }
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;
      vha[i] = make_uint4(0, 0, 0, 0);
      if (idx < 2 * nunits) {
        const int half = idx >= nunits ? 1 : 0;
        const int u = idx - half * nunits;
        const int g = u & 3, pix = u >> 2;
        const int hy = pix / pd.hw, hx = pix - hy * pd.hw;
        const int y = (a0 + pd.dy0 + hy) * d.SS + pd.py;
        const int x = (b0 + pd.dx0 + hx) * d.SS + pd.px;
        const int c0 = (cb * 2 + half) * 32 + g * 8;
        if (y >= 0 && y < d.IH && x >= 0 && x < d.IW && c0 < d.Cin_pad)
          vha[i] = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch + c0);
      }
    }
  };
  auto tile_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NDO; ++i) *reinterpret_cast<uint4*>(buf + (tid + i * 256) * 16) = vdo[i];   // [ni][px][32 n]
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;                             // == (half * hh*hw + pix) * 4 + g : the LDS image order
      if (idx < 2 * nunits) *reinterpret_cast<uint4*>(buf + DO_BYTES + idx * 16) = vha[i];
    }
  };

  const bool active = (Nn_pad - nb * 64) > nh * 32 && (cb * 2 + ch) * 32 < d.Cin_pad;   // quadrant outside the real tensor: no MFMAs
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trLane = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + 4 * ROWB));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  auto compute = [&](const unsigned char* buf) {
    if (!active) return;
    const unsigned char* acol = buf + nh * 128 * ROWB + trLane;
    const unsigned char* hcol = buf + DO_BYTES + ch * HP + trLane;
    const int rowb = pd.hw * ROWB;
    uint4 b0[2], b1[2];
    b0[0] = tr16(hcol);
    b0[1] = tr16(hcol + ROWB);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      b1[0] = tr16(hcol + (kt + 1) * rowb);
      b1[1] = tr16(hcol + (kt + 1) * rowb + ROWB);
      const uint4 a = tr16(acol + kt * 16 * ROWB);
      const bf16x8_t av = __builtin_bit_cast(bf16x8_t, a);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b0[0]), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b0[1]), acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b1[0]), acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, b1[1]), acc[3], 0, 0, 0);
      b0[0] = b1[0]; b0[1] = b1[1];
    }
  };

  int tl = sp;
  int cur = 0;
  if (tl < ntiles) { tile_load(tl); tile_store(smem); }
  __syncthreads();
  for (; tl < ntiles; tl += nsplit) {
    const bool more = (tl + nsplit) < ntiles;
    if (more) tile_load(tl + nsplit);
    compute(smem + cur * BUF_BYTES);
    if (more) tile_store(smem + (cur ^ 1) * BUF_BYTES);
    __syncthreads();
    cur ^= 1;
  }

  if (!active) return;
  if (slab) {                                                    // split-K partial -> this workgroup's slab, register order (see tfc_wgrad_reduce_kernel)
    float4* ps = slab + ((size_t)bid * 4 + wave) * (4 * 4 * 64) + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        ps[(a * 4 + q4) * 64] = make_float4(acc[a][4 * q4], acc[a][4 * q4 + 1], acc[a][4 * q4 + 2], acc[a][4 * q4 + 3]);
    return;
  }
  const int c = (cb * 2 + ch) * 32 + (lane & 31);
  const int hrow = lane >> 5;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int mask = 0;
    for (int t = 0; t < 4; ++t)
      if (pd.tap_dy[t] == (a >> 1) && pd.tap_dx[t] == (a & 1)) mask = pd.tap_mask[t];
    for (int m = mask; m; m &= m - 1) {
      const int slot = __ffs(m) - 1;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int n = nb * 64 + nh * 32 + (j & 3) + 8 * (j >> 2) + 4 * hrow;
        if (n < Nn_real && c < Cw_real) atomicAdd(&dwacc[((size_t)slot * Nn_real + n) * Cw_real + c], acc[a][j]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// weight-gradient GEMM:  dWacc[slot][n][c] += sum_pixels dO[pixel][n] * in[src(pixel, tap)][c]
//   rows = n (64 per workgroup = 2 MFMA row blocks), cols = c (32 per workgroup), K = the 128 pixels of a tile;
//   wave w owns taps {w*TPW .. w*TPW+TPW-1}; a workgroup walks a strided subset of the pixel tiles (split-K) and
//   flushes with fp32 atomics (each wave-instruction adds two contiguous 128-B row segments).
//   bf16: both operands are K(pixel)-strided in LDS, fetched with ds_read_b64_tr_b16 (hardware transpose);
//   fp32: 32x32x2 operands are one dword per lane, plain ds_read_b32.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct WgradFrag;

//   RASTER (bf16, 16 taps in 4x4 raster order): wave w owns filter COLUMN kx = w (taps ky*4 + w). The B fragment of (k-step kt,
//   filter row ky) is the halo row kt + ky at column shift w, i.e. it only depends on kt + ky: a 4-deep sliding register
//   window needs ONE new fragment per k-step instead of four.
template <typename T, int TPW, bool RASTER>
__global__ void __launch_bounds__(256, 2)
tfc_wgrad_kernel(const TfcGather d, const T* __restrict__ dO, const T* __restrict__ in, float* dwacc, float4* slab,
                 int Nn_pad, int Nn_real, int Cw_real, int nbw, int ncb, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  constexpr int UE = 16 / ES;
  constexpr int UPN = 32 / UE;                                   // 16-B units per 32 channels
  constexpr int ROWB = 32 * ES;                                  // bytes per LDS row (32 channels)
  constexpr int DO_BYTES = 2 * 128 * ROWB;
  constexpr int HALO_BYTES = TFC_MAX_HH * TFC_MAX_HW * ROWB;
  constexpr int BUF_BYTES = DO_BYTES + HALO_BYTES;
  constexpr bool PREFETCH = (ES == 2);
  constexpr int NDO = (2 * 128 * UPN) / 256;                     // dO units per thread (4 bf16 / 8 fp32)
  constexpr int NHA = (TFC_MAX_HH * TFC_MAX_HW * UPN + 255) / 256;  // halo units per thread (4 / 7)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TfcPlane& pd = d.plane[0];

  const int bid = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % (nbw * ncb);
  const int sp = bid / (nbw * ncb);
  const int cb = pair % ncb, nb = pair / ncb;
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x;

  f32x16_t acc[TPW][2];
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[ti][ni][j] = 0.f;

  uint4 vdo[NDO], vha[NHA];
  int hyq[NHA], hxq[NHA], hcq[NHA];
  bool hok[NHA];
  {
    const int nunits = pd.hh * pd.hw * UPN;
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;
      const int g = idx % UPN, pix = idx / UPN;
      hyq[i] = pix / pd.hw; hxq[i] = pix - hyq[i] * pd.hw;
      hcq[i] = cb * 32 + g * UE;
      hok[i] = idx < nunits && hcq[i] < d.Cin_pad;
    }
  }
  auto tile_load = [&](int tl) {
#ifdef TFC_ABL_WG_SAMETILE
    tl = sp;                                                     // ablation: every load hits the workgroup's first (cache-hot) tile
#endif
    int t = tl;
    const int txb = t % d.tiles_x; t /= d.tiles_x;
    const int tyb = t % d.tiles_y;
    const int img = t / d.tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < NDO; ++i) {
      const int idx = tid + i * 256;
      const int g = idx % UPN, px = (idx / UPN) & 127, ni = idx / (UPN * 128);
      const int a = a0 + (px >> 4), b = b0 + (px & 15);
      const int n0 = nb * 64 + ni * 32 + g * UE;
      vdo[i] = make_uint4(0, 0, 0, 0);
      if (a < d.GH && b < d.GW && n0 < Nn_pad) {
        const int oy = a * d.OS + d.OOY, ox = b * d.OS + d.OOX;
        vdo[i] = *reinterpret_cast<const uint4*>(dO + ((size_t)(img * d.OH + oy) * d.OW + ox) * d.out_pitch + n0);
      }
    }
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      vha[i] = make_uint4(0, 0, 0, 0);
      if (hok[i]) {                                               // (hy, hx, channel) of a thread's halo units are tile-invariant: computed once per launch
        const int y = (a0 + pd.dy0 + hyq[i]) * d.SS + pd.py;
        const int x = (b0 + pd.dx0 + hxq[i]) * d.SS + pd.px;
        if (y >= 0 && y < d.IH && x >= 0 && x < d.IW)
          vha[i] = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch + hcq[i]);
      }
    }
  };
  auto tile_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NDO; ++i) {
      const int idx = tid + i * 256;                             // == (ni*128 + px)*UPN + g : the LDS image order
      *reinterpret_cast<uint4*>(buf + idx * 16) = vdo[i];
    }
    const int nunits = pd.hh * pd.hw * UPN;
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;                             // == pix*UPN + g
      if (idx < nunits) *reinterpret_cast<uint4*>(buf + DO_BYTES + idx * 16) = vha[i];
    }
  };

  const int nni = (Nn_pad - nb * 64) > 32 ? 2 : 1;               // second 32-row block empty (e.g. the 3-channel head)? skip its MFMAs
  // lane decode for the transposing reads (bf16) / dword reads (fp32)
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trLane = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;    // bf16 only

  auto compute = [&](const unsigned char* buf) {
    const unsigned char* dob = buf;
    const unsigned char* hab = buf + DO_BYTES;
    if constexpr (ES == 2 && RASTER) {
      auto tr16 = [&](const unsigned char* p0) {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + 4 * ROWB));
        uint4 r;
        r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
        r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
        r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
        r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
        return r;
      };
      const unsigned char* hcol = hab + wave * ROWB + trLane;     // halo column shift kx = wave
      const int rowb = pd.hw * ROWB;
      uint4 bw[4];
      bw[0] = tr16(hcol);
      bw[1] = tr16(hcol + rowb);
      bw[2] = tr16(hcol + 2 * rowb);
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        bw[(kt + 3) & 3] = tr16(hcol + (kt + 3) * rowb);
        uint4 a[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) a[ni] = tr16(dob + ni * 128 * ROWB + kt * 16 * ROWB + trLane);
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            if (ni < nni)
              acc[ky][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ni]),
                                                                     __builtin_bit_cast(bf16x8_t, bw[(kt + ky) & 3]), acc[ky][ni], 0, 0, 0);
      }
    } else if constexpr (ES == 2) {
#pragma unroll 2
      for (int kt = 0; kt < 8; ++kt) {
        uint4 a[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const unsigned char* pa = dob + ni * 128 * ROWB + kt * 16 * ROWB + trLane;
          s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, pa));
          s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, pa + 4 * ROWB));
          a[ni].x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
          a[ni].y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
          a[ni].z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
          a[ni].w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
        }
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
          const int tap = ti * 4 + wave;
          if (tap < pd.ntaps) {                                  // wave-uniform
            const unsigned char* pb = hab + ((kt + pd.tap_dy[tap]) * pd.hw + pd.tap_dx[tap]) * ROWB + trLane;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, pb));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, pb + 4 * ROWB));
            uint4 b;
            b.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
            b.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
            b.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
            b.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              if (ni < nni)
                acc[ti][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ni]),
                                                                       __builtin_bit_cast(bf16x8_t, b), acc[ti][ni], 0, 0, 0);
          }
        }
      }
    } else {
      const int r = lane & 31, kh = lane >> 5;
      for (int kk = 0; kk < 64; ++kk) {
        const int px = 2 * kk + kh;
        float a[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) a[ni] = *reinterpret_cast<const float*>(dob + (ni * 128 + px) * ROWB + r * 4);
        const int ty = px >> 4, tx = px & 15;
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
          const int tap = ti * 4 + wave;
          if (tap < pd.ntaps) {
            const float b = *reinterpret_cast<const float*>(hab + ((ty + pd.tap_dy[tap]) * pd.hw + tx + pd.tap_dx[tap]) * ROWB + r * 4);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              if (ni < nni) acc[ti][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ni], b, acc[ti][ni], 0, 0, 0);
          }
        }
      }
    }
  };

  if constexpr (PREFETCH) {
    int tl = sp;
    int cur = 0;
    if (tl < ntiles) { tile_load(tl); tile_store(smem); }
    __syncthreads();
    for (; tl < ntiles; tl += nsplit) {
      const bool more = (tl + nsplit) < ntiles;
      if (more) tile_load(tl + nsplit);
      compute(smem + cur * BUF_BYTES);
      if (more) tile_store(smem + (cur ^ 1) * BUF_BYTES);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    for (int tl = sp; tl < ntiles; tl += nsplit) {
      tile_load(tl);
      tile_store(smem);
      __syncthreads();
      compute(smem);
      __syncthreads();
    }
  }

  // ---- flush ----
  if (slab) {                                                    // split-K partial -> this workgroup's slab, register order (see tfc_wgrad_reduce_kernel)
    float4* ps = slab + ((size_t)bid * 4 + wave) * (TPW * 2 * 4 * 64) + lane;
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
          ps[((ti * 2 + ni) * 4 + q4) * 64] = make_float4(acc[ti][ni][4 * q4], acc[ti][ni][4 * q4 + 1], acc[ti][ni][4 * q4 + 2], acc[ti][ni][4 * q4 + 3]);
    return;
  }
  const int c = cb * 32 + (lane & 31);
  const int hrow = lane >> 5;
#pragma unroll
  for (int ti = 0; ti < TPW; ++ti) {
    const int tap = ti * 4 + wave;
    if (tap < pd.ntaps) {
      for (int m = pd.tap_mask[tap]; m; m &= m - 1) {              // a collapsed tap feeds every filter tap it stands for
        const int slot = __ffs(m) - 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int n = nb * 64 + ni * 32 + (j & 3) + 8 * (j >> 2) + 4 * hrow;
            if (ni < nni && n < Nn_real && c < Cw_real)
              atomicAdd(&dwacc[((size_t)slot * Nn_real + n) * Cw_real + c], acc[ti][ni][j]);
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of the FIRST layer (8 padded input channels, 4 x 4 raster taps, <= 64 outputs; bf16): G down1 3 -> 64 and D block 1 6 -> 64 at
// 256 x 256. The generic kernel pads the 8 channels to a 32-wide MFMA tile per tap -- 4x the arithmetic, 127 us for 12.8 GFLOP. Here the four
// taps of a filter ROW share one tile: the halo is kept at 16 bytes per pixel, so column (kx, c) of pixel x lives at  x * 16 + (kx * 8 + c) * 2
// bytes -- a K(pixel)-strided matrix whose rows overlap (row pitch 16 B, row length 64 B), which the transposing LDS read takes as it is.
// Wave w owns filter row ky = w: per k-step (16 pixels of a tile row) 2 A fragments (dO^T, 2 x 32 outputs), 1 B fragment, 2 MFMAs -- a quarter
// of the generic kernel's MFMAs; the bound becomes the 266 MB dO stream. Split-K partials go to slabs [workgroup][ky][ni][q][lane] (32 KB per
// workgroup), summed by tfc_wgrad_c8_reduce_kernel into the fp32 accumulator [slot][n][c] that tfc_wgrad_finish_kernel lays out.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_wgrad_c8_kernel(const TfcGather d, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ in, float4* __restrict__ slab, int Nn_pad, int nsplit) {
  constexpr int ROWB = 64;                                       // dO: 32 outputs x 2 bytes per LDS pixel row
  constexpr int DO_BYTES = 2 * 128 * ROWB;
  constexpr int HALO_BYTES = (TFC_MAX_HH * TFC_MAX_HW * 16 + 255) & ~255;
  constexpr int BUF_BYTES = DO_BYTES + HALO_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TfcPlane& pd = d.plane[0];
  const int sp = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x;
  f32x16_t acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[ni][j] = 0.f;

  uint4 vdo[4], vha;
  auto tile_load = [&](int tl) {
    int t = tl;
    const int txb = t % d.tiles_x; t /= d.tiles_x;
    const int tyb = t % d.tiles_y;
    const int img = t / d.tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;                             // (ni * 128 + px) * 4 + g
      const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
      const int a = a0 + (px >> 4), b = b0 + (px & 15);
      const int n0 = ni * 32 + g * 8;
      vdo[i] = make_uint4(0, 0, 0, 0);
      if (a < d.GH && b < d.GW && n0 < Nn_pad)
        vdo[i] = load_stream16(dO + ((size_t)(img * d.OH + a + d.OOY) * d.OW + b + d.OOX) * d.out_pitch + n0);
    }
    vha = make_uint4(0, 0, 0, 0);
    if (tid < pd.hh * pd.hw) {
      const int hy = tid / pd.hw, hx = tid - hy * pd.hw;
      const int y = a0 + pd.dy0 + hy, x = b0 + pd.dx0 + hx;
      if (y >= 0 && y < d.IH && x >= 0 && x < d.IW) vha = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch);
    }
  };
  auto tile_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(buf + (tid + i * 256) * 16) = vdo[i];
    if (tid < TFC_MAX_HH * TFC_MAX_HW) *reinterpret_cast<uint4*>(buf + DO_BYTES + tid * 16) = vha;
  };
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trA = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;        // dO: rows = pixels (64 B), columns = outputs
  const int trB = (8 * hk + q) * 16 + cb16 * 32 + p * 8;          // halo: rows = pixels (16 B pitch, 64 B long: overlapping), columns = (kx, c)
  auto tr16 = [&](const unsigned char* p0, int rowb4) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + rowb4));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  auto compute = [&](const unsigned char* buf) {
    const unsigned char* hrow = buf + DO_BYTES + wave * pd.hw * 16 + trB;   // halo row kt + ky, ky = wave
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const uint4 b = tr16(hrow + kt * pd.hw * 16, 4 * 16);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const uint4 a = tr16(buf + ni * 128 * ROWB + kt * 16 * ROWB + trA, 4 * ROWB);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[ni], 0, 0, 0);
      }
    }
  };
  int tl = sp, cur = 0;
  if (tl < ntiles) { tile_load(tl); tile_store(smem); }
  __syncthreads();
  for (; tl < ntiles; tl += nsplit) {
    const bool more = (tl + nsplit) < ntiles;
    if (more) tile_load(tl + nsplit);
    compute(smem + cur * BUF_BYTES);
    if (more) tile_store(smem + (cur ^ 1) * BUF_BYTES);
    __syncthreads();
    cur ^= 1;
  }
  float4* ps = slab + ((size_t)sp * 4 + wave) * (2 * 4 * 64) + lane;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      ps[(ni * 4 + q4) * 64] = make_float4(acc[ni][4 * q4], acc[ni][4 * q4 + 1], acc[ni][4 * q4 + 2], acc[ni][4 * q4 + 3]);
}
// ---------------------------------------------------------------------------------------------------
// First block, backward, fused: [BlurPool(stride 2)]^T -> LeakyReLU' -> weight gradient (+ bias gradient), bf16. The unfused chain
// (tfc_act_pool2_bwd_kernel, then tfc_wgrad_c8_kernel) is bound by WRITING the 255 x 255 x 64 gradient of the convolution output (266 MB at
// batch 32; HBM writes run at about half the read rate) and reading it back. When nothing else needs that tensor -- every backward of the first
// block except the one generator-step pass that continues to the image gradient -- it never has to exist: this kernel is tfc_wgrad_c8_kernel
// with a PRODUCER in front of its LDS tile. Per 8 x 16 tile of convolution outputs: the 7 x 11 window of the pooled gradient is staged in LDS,
// every thread turns its four 16-byte units of the stored activation (sign only) into d_raw = (sum of its 2 x 2 -- next to the reflect borders
// 3 x 3 -- window taps) * (y > 0 ? 1 : slope), rounds to bf16 where the unfused kernel stored it, and writes the dO tile the MFMA loop reads.
// Reads 266 + 67 + 33 MB, writes 32 KB of slabs per workgroup. Tap weights with the reflect aliases merged are tabulated per tile row / column
// exactly as in tfc_act_pool2_bwd_kernel, and the arithmetic order is the same, so d_raw -- and with it the weight gradient -- has the same bits.
// Tiles are dealt out CONTIGUOUSLY inside ONE image per workgroup (wpi workgroups per image), so the bias-gradient sums leave as one 64-float slot
// rstats = part[img][workgroup of the image][64], added in a fixed order by tfc_part_reduce_kernel (round 2 used LDS + memory-side float atomics).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fbw(int k) { return (k == 0 || k == 3) ? 0.125f : 0.375f; }
__global__ void __launch_bounds__(256, 2)
tfc_wgrad_c8_fused_kernel(const TfcGather d, const bf16_t* __restrict__ yact, int y_pitch, const bf16_t* __restrict__ dyp, int dyp_pitch, int Ho, int Wo,
                          const bf16_t* __restrict__ in, float4* __restrict__ slab, float* rstats, float slope, int wpi, int per) {
  constexpr int ROWB = 64;
  constexpr int DO_BYTES = 2 * 128 * ROWB;
  constexpr int HALO_BYTES = (TFC_MAX_HH * TFC_MAX_HW * 16 + 255) & ~255;
  constexpr int WH = 7, WW = 11, WIN_BYTES = WH * WW * 8 * 16;     // pooled-gradient window: 77 pixels x 64 channels
  __shared__ __attribute__((aligned(16))) unsigned char smem[DO_BYTES + 2 * HALO_BYTES + WIN_BYTES + 24 * 16 + 4 * 64 * 4];
  unsigned char* halo0 = smem + DO_BYTES;
  uint4* win = reinterpret_cast<uint4*>(smem + DO_BYTES + 2 * HALO_BYTES);
  float4* wrow = reinterpret_cast<float4*>(smem + DO_BYTES + 2 * HALO_BYTES + WIN_BYTES);   // [8] tap weights of pooled rows o0-1, o0, o0+1
  float4* wcol = wrow + 8;                                                                  // [16]
  float* sbias = reinterpret_cast<float*>(wcol + 16);                                       // [4 waves][64] bias-gradient sums of this workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TfcPlane& pd = d.plane[0];
  const int sp = blockIdx.x;
  const int tpi = d.tiles_y * d.tiles_x;                          // tiles per image; this workgroup: tiles [t0, t1) of image sp / wpi
  const int wimg = sp / wpi, wj = sp - wimg * wpi;
  const int t0 = wimg * tpi + wj * per, t1 = (wj * per + per) < tpi ? (t0 + per) : (wimg + 1) * tpi;
  f32x16_t acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[ni][j] = 0.f;

  auto decode = [&](int tl, int& img, int& a0, int& b0) {
    int t = tl;
    const int txb = t % d.tiles_x; t /= d.tiles_x;
    const int tyb = t % d.tiles_y;
    img = t / d.tiles_y;
    a0 = tyb * TFC_TILE_H; b0 = txb * TFC_TILE_W;
  };
  uint4 vy[4], vw[3], vha;
  auto tile_load = [&](int tl) {
    int img, a0, b0;
    decode(tl, img, a0, b0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;                             // (ni * 128 + px) * 4 + g
      const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
      const int a = a0 + (px >> 4), b = b0 + (px & 15);
      vy[i] = make_uint4(0, 0, 0, 0);
      if (a < d.GH && b < d.GW) vy[i] = load_stream16(yact + ((size_t)(img * d.OH + a) * d.OW + b) * y_pitch + ni * 32 + g * 8);
    }
    const int oyb = a0 / 2 - 1, oxb = b0 / 2 - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;                             // window pixel * 8 + unit
      const int u = idx & 7, wp = idx >> 3;
      const int oy = oyb + wp / WW, ox = oxb + wp % WW;
      vw[i] = make_uint4(0, 0, 0, 0);
      if (wp < WH * WW && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) vw[i] = *reinterpret_cast<const uint4*>(dyp + ((size_t)(img * Ho + oy) * Wo + ox) * dyp_pitch + u * 8);
    }
    vha = make_uint4(0, 0, 0, 0);
    if (tid < pd.hh * pd.hw) {
      const int hy = tid / pd.hw, hx = tid - hy * pd.hw;
      const int y = a0 + pd.dy0 + hy, x = b0 + pd.dx0 + hx;
      if (y >= 0 && y < d.IH && x >= 0 && x < d.IW) vha = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch);
    }
  };
  // tap weights of one act row / column q of length L (Lo pooled): pooled rows o0-1, o0, o0+1 with o0 = (q+1) >> 1, reflect aliases merged
  auto taps3 = [&](int q, int L, int Lo) {
    float w3[3] = {0.f, 0.f, 0.f};
    if (q < L) {
      const int o0 = (q + 1) >> 1;
      for (int a = 0; a < 4; ++a) {
        if ((a == 1 && q != 1) || (a == 2 && q != L - 2) || (a == 3 && q != L - 3)) continue;
        const int pq = a == 0 ? q : (a == 1 ? -1 : (a == 2 ? L : L + 1));
        for (int k = 0; k < 4; ++k) {
          const int t = pq + 1 - k;
          if (t < 0 || (t & 1) || (t >> 1) >= Lo) continue;
          const int dd = (t >> 1) - o0 + 1;
          if (dd == 0) w3[0] += fbw(k); else if (dd == 1) w3[1] += fbw(k); else if (dd == 2) w3[2] += fbw(k);
        }
      }
    }
    return make_float4(w3[0], w3[1], w3[2], 0.f);
  };
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trA = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  const int trB = (8 * hk + q) * 16 + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0, int rowb4) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + rowb4));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  float bsum[2][8];                                              // this thread's channels: ni = 0 / 1, unit g = tid & 3
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[h2][e] = 0.f;
  if (t0 < t1) tile_load(t0);
  int cur = 0;
  for (int tl = t0; tl < t1; ++tl) {
    int img, a0, b0;
    decode(tl, img, a0, b0);
    // A. window, halo, tap tables of this tile -> LDS; the activation units stay in registers
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int idx = tid + i * 256; if (idx < WH * WW * 8) win[idx] = vw[i]; }
    if (tid < TFC_MAX_HH * TFC_MAX_HW) *reinterpret_cast<uint4*>(halo0 + cur * HALO_BYTES + tid * 16) = vha;
    if (tid < 8) wrow[tid] = taps3(a0 + tid, d.GH, Ho);
    else if (tid < 24) wcol[tid - 8] = taps3(b0 + tid - 8, d.GW, Wo);
    uint4 ycur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ycur[i] = vy[i];
    __syncthreads();
    // C. next tile's loads
    if (tl + 1 < t1) tile_load(tl + 1);
    // D. d_raw of this tile -> the dO image the MFMA loop reads
    const int oyb = a0 / 2 - 1, oxb = b0 / 2 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
      const int ty = px >> 4, tx = px & 15;
      const int y = a0 + ty, x = b0 + tx;
      uint4 res = make_uint4(0, 0, 0, 0);
      if (y < d.GH && x < d.GW) {
        float gsum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gsum[e] = 0.f;
        const float4 wr = wrow[ty], wc = wcol[tx];
        const int wy = ((y + 1) >> 1) - oyb, wx = ((x + 1) >> 1) - oxb;
        const int u = ni * 4 + g;
        auto tap = [&](int dy, int dx, float w) {
          float v[8];
          unpack16<bf16_t>(win[((wy + dy) * WW + wx + dx) * 8 + u], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) gsum[e] += w * v[e];
        };
        tap(-1, -1, wr.x * wc.x); tap(-1, 0, wr.x * wc.y); tap(0, -1, wr.y * wc.x); tap(0, 0, wr.y * wc.y);
        if (wr.z != 0.f || wc.z != 0.f) {                          // only next to the bottom / right reflect border
          tap(-1, 1, wr.x * wc.z); tap(0, 1, wr.y * wc.z);
          tap(1, -1, wr.z * wc.x); tap(1, 0, wr.z * wc.y); tap(1, 1, wr.z * wc.z);
        }
        float yv[8];
        unpack16<bf16_t>(ycur[i], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) gsum[e] = yv[e] > 0.f ? gsum[e] : gsum[e] * slope;
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i >> 1][e] += gsum[e];
        res = pack16<bf16_t>(gsum);
      }
      *reinterpret_cast<uint4*>(smem + idx * 16) = res;
    }
    __syncthreads();
    // F. the weight-gradient MFMAs of tfc_wgrad_c8_kernel
    {
      const unsigned char* buf = smem;
      const unsigned char* hrow = halo0 + cur * HALO_BYTES + wave * pd.hw * 16 + trB;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const uint4 b = tr16(hrow + kt * pd.hw * 16, 4 * 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const uint4 a = tr16(buf + ni * 128 * ROWB + kt * 16 * ROWB + trA, 4 * ROWB);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[ni], 0, 0, 0);
        }
      }
    }
    cur ^= 1;
  }
  if (rstats) {
    // bias-gradient sums of this workgroup, in a fixed order: the lanes that share (lane & 3) hold the same 16 channels -> butterfly over lane bits
    // 2..5, the four waves meet in four LDS slots, 64 threads add the slots in wave order
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bsum[h2][e];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (lane < 4) sbias[wave * 64 + h2 * 32 + lane * 8 + e] = v;
      }
    __syncthreads();
    if (tid < 64) rstats[(size_t)sp * 64 + tid] = ((sbias[tid] + sbias[64 + tid]) + sbias[128 + tid]) + sbias[192 + tid];   // part[img][wj][64], sp = img * wpi + wj
  }
  float4* ps = slab + ((size_t)sp * 4 + wave) * (2 * 4 * 64) + lane;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      ps[(ni * 4 + q4) * 64] = make_float4(acc[ni][4 * q4], acc[ni][4 * q4 + 1], acc[ni][4 * q4 + 2], acc[ni][4 * q4 + 3]);
}
// ---------------------------------------------------------------------------------------------------
// The same fused first-block backward with the transposed blur ON THE MATRIX CORE (round 3). The kernel above spends ~1,200 VALU instructions per
// thread and tile on the 2 x 2 (3 x 3 at the reflect borders) taps -- 5.5 k cycles per tile against 0.5 k for its 16 weight-gradient MFMAs: VALU-bound
// at 170 us where its 366 MB of HBM traffic need ~80. The transposed blur of a tile is a LINEAR map from the 7 x 11 window of the pooled gradient
// to the tile's 8 x 16 pixels, the same for every channel:   g[p][c] = sum_q T[p][q] * win[q][c],   T[p][q] = wrow[ty][dy] * wcol[tx][dx].
// Its entries are products of sums of {1/8, 3/8} -- exactly representable in bf16 -- and win is bf16, so the products are exact in fp32 and only the
// ORDER of the <= 9 additions differs from the VALU form (fp32 round-off; 1 bf16 ulp on rare elements after rounding). As a GEMM per tile:
// 128 pixels x 80 (77 + 3 zero) window positions x 64 channels = 40 MFMAs 32x32x16 -- 10 per wave, 320 cycles. Operands: the window is staged as two
// 32-channel planes [q][32] of 64-byte rows and fetched with the transposing LDS read the weight-gradient loop already uses (A, rows = channels);
// T lives in LDS as Tt[pixel][q] (176-byte rows: conflict-free 16-byte reads) and is rebuilt only when the tile's border class changes (first /
// interior / last tile rows and columns; tiles are dealt contiguously); lanes hold 16 channels of ONE pixel, so the LeakyReLU' mask comes from a
// 64-bit sign word per pixel that the threads build from their (coalesced, natural-layout) loads of the stored activation.
// ---------------------------------------------------------------------------------------------------
template <bool MASK>                                              // MASK: the forward pass left sign words (8 bytes per pixel are read instead of 128)
__global__ void __launch_bounds__(256, 2)
tfc_wgrad_c8_fusedm_kernel(const TfcGather d, const bf16_t* __restrict__ yact, int y_pitch, const bf16_t* __restrict__ dyp, int dyp_pitch, int Ho, int Wo,
                           const bf16_t* __restrict__ in, float4* __restrict__ slab, float* rstats, float slope, int wpi, int per,
                           const unsigned char* __restrict__ sign_mask) {
  constexpr int ROWB = 64;
  constexpr int DO_BYTES = 2 * 128 * ROWB;
  constexpr int HALO_BYTES = (TFC_MAX_HH * TFC_MAX_HW * 16 + 255) & ~255;
  constexpr int WH = 7, WW = 11, NQ = 80;                          // pooled-gradient window: 77 pixels (+ 3 zero rows: K = 5 x 16)
  constexpr int PLANE = NQ * ROWB, WIN_BYTES = 2 * PLANE;          // two 32-channel planes [q][32 channels]
  constexpr int TP = 176, TT_BYTES = 128 * TP;                     // Tt[pixel][88 bf16]
  __shared__ __attribute__((aligned(16))) unsigned char smem[DO_BYTES + 2 * HALO_BYTES + WIN_BYTES + TT_BYTES + 128 * 8 + 24 * 16 + 4 * 64 * 4];
  unsigned char* halo0 = smem + DO_BYTES;
  unsigned char* winb = halo0 + 2 * HALO_BYTES;
  unsigned char* ttb = winb + WIN_BYTES;
  unsigned char* maskb = ttb + TT_BYTES;                           // [128 pixels][8 bytes]: bit c of pixel p = (stored activation of channel c > 0)
  float4* wrow = reinterpret_cast<float4*>(maskb + 128 * 8);       // [8] tap weights of pooled rows o0-1, o0, o0+1
  float4* wcol = wrow + 8;                                         // [16]
  float* sbias = reinterpret_cast<float*>(wcol + 16);              // [4 waves][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TfcPlane& pd = d.plane[0];
  const int sp = blockIdx.x;
  const int tpi = d.tiles_y * d.tiles_x;
  const int wimg = sp / wpi, wj = sp - wimg * wpi;
  const int t0 = wimg * tpi + wj * per, t1 = (wj * per + per) < tpi ? (t0 + per) : (wimg + 1) * tpi;
  f32x16_t acc[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[ni][j] = 0.f;
  for (int i = tid; i < 2 * 3 * 4; i += 256) {                     // window rows 77..79 of both planes stay zero for the whole launch
    const int pl = i / 12, r = i % 12;
    *reinterpret_cast<uint4*>(winb + pl * PLANE + 77 * ROWB + r * 16) = make_uint4(0, 0, 0, 0);
  }
  // a workgroup's tiles are consecutive tiles of ONE image: (tile row, tile column) advance incrementally and every per-thread index that does not
  // depend on the tile is computed once (the integer divisions of a per-tile decode were a third of the VALU work of a tile)
  const int img = wimg;
  int wrow_i[3], wcol_i[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { const int wp = (tid + i * 256) >> 3; wrow_i[i] = wp / WW; wcol_i[i] = wp - wrow_i[i] * WW; }
  const int hy_t = tid / pd.hw, hx_t = tid - hy_t * pd.hw;
  // Loads run DEPTH tiles ahead of their use, in DEPTH register stages (a tile is a few microseconds of work and its loads are scattered rows of three
  // tensors: one tile of read-ahead left every tile waiting for most of a memory round trip). With sign words a stage is 18 VGPRs: two stages; with the
  // stored activation (64 more bytes per thread) one.
  constexpr int DEPTH = MASK ? 2 : 1;
  struct Stage { uint4 vy[4], vw[3], vha; uint2 vm; };
  Stage S0, S1;
  auto tile_load = [&](Stage& S, int a0, int b0) {
    uint4 (&vy)[4] = S.vy; uint4 (&vw)[3] = S.vw; uint4& vha = S.vha; uint2& vm = S.vm;
    if constexpr (MASK) {                                         // 8 bytes per pixel instead of the 128 bytes of the stored activation
      vm = make_uint2(0, 0);
      if (tid < 128) {
        const int a = a0 + (tid >> 4), b = b0 + (tid & 15);
        if (a < d.GH && b < d.GW) vm = *reinterpret_cast<const uint2*>(sign_mask + ((size_t)(img * d.OH + a) * d.OW + b) * 8);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;                           // (ni * 128 + px) * 4 + g
        const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
        const int a = a0 + (px >> 4), b = b0 + (px & 15);
        vy[i] = make_uint4(0, 0, 0, 0);
        if (a < d.GH && b < d.GW) vy[i] = load_stream16(yact + ((size_t)(img * d.OH + a) * d.OW + b) * y_pitch + ni * 32 + g * 8);
      }
    }
    const int oyb = a0 / 2 - 1, oxb = b0 / 2 - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;                             // window pixel * 8 + unit
      const int u = idx & 7;
      const int oy = oyb + wrow_i[i], ox = oxb + wcol_i[i];
      vw[i] = make_uint4(0, 0, 0, 0);
      if (wrow_i[i] < WH && oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) vw[i] = *reinterpret_cast<const uint4*>(dyp + ((size_t)(img * Ho + oy) * Wo + ox) * dyp_pitch + u * 8);
    }
    vha = make_uint4(0, 0, 0, 0);
    if (tid < pd.hh * pd.hw) {
      const int hy = hy_t, hx = hx_t;
      const int y = a0 + pd.dy0 + hy, x = b0 + pd.dx0 + hx;
      if (y >= 0 && y < d.IH && x >= 0 && x < d.IW) vha = *reinterpret_cast<const uint4*>(in + ((size_t)(img * d.IH + y) * d.IW + x) * d.in_pitch);
    }
  };
  auto taps3 = [&](int q, int L, int Lo) {                         // as in the VALU form: tap weights of pooled rows o0-1, o0, o0+1, reflect aliases merged
    float w3[3] = {0.f, 0.f, 0.f};
    if (q < L) {
      const int o0 = (q + 1) >> 1;
      for (int a = 0; a < 4; ++a) {
        if ((a == 1 && q != 1) || (a == 2 && q != L - 2) || (a == 3 && q != L - 3)) continue;
        const int pq = a == 0 ? q : (a == 1 ? -1 : (a == 2 ? L : L + 1));
        for (int k = 0; k < 4; ++k) {
          const int t = pq + 1 - k;
          if (t < 0 || (t & 1) || (t >> 1) >= Lo) continue;
          const int dd = (t >> 1) - o0 + 1;
          if (dd == 0) w3[0] += fbw(k); else if (dd == 1) w3[1] += fbw(k); else if (dd == 2) w3[2] += fbw(k);
        }
      }
    }
    return make_float4(w3[0], w3[1], w3[2], 0.f);
  };
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trA = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  const int trB = (8 * hk + q) * 16 + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0, int rowb4) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + rowb4));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  float bsum[2][16];                                              // bias-gradient sums of this lane's 2 x 16 channels (channel = cb*32 + (j&3) + 8(j>>2) + 4h)
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < 16; ++j) bsum[cb][j] = 0.f;
  const int hh = lane >> 5, pl = 32 * wave + (lane & 31);          // this lane's pixel of the tile in the blur GEMM (wave = 32-pixel block = 2 tile rows)
  int key_a = -2, key_b = -2;                                      // border class of the tile Tt was built for

  int cur = 0;
  auto body = [&](Stage& S, int a0, int b0, bool prefetch, int pa0, int pb0) {
    uint4 (&vy)[4] = S.vy; uint4 (&vw)[3] = S.vw; uint4& vha = S.vha; uint2& vm = S.vm;
    // A. window planes, halo, sign words of this tile -> LDS
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;
      if (idx < WH * WW * 8) { const int u = idx & 7, wp = idx >> 3; *reinterpret_cast<uint4*>(winb + (u >> 2) * PLANE + wp * ROWB + (u & 3) * 16) = vw[i]; }
    }
    if (tid < TFC_MAX_HH * TFC_MAX_HW) *reinterpret_cast<uint4*>(halo0 + cur * HALO_BYTES + tid * 16) = vha;
    // the tap tables (and the tap matrix built from them) change only with the tile's border class: interior tiles share one set
    const int ka = (a0 == 0 || a0 + TFC_TILE_H > d.GH - 3) ? a0 : -1, kb = (b0 == 0 || b0 + TFC_TILE_W > d.GW - 3) ? b0 : -1;
    const bool rebuild = ka != key_a || kb != key_b;
    if (rebuild) {
      if (tid < 8) wrow[tid] = taps3(a0 + tid, d.GH, Ho);
      else if (tid < 24) wcol[tid - 8] = taps3(b0 + tid - 8, d.GW, Wo);
    }
    if constexpr (MASK) {
      if (tid < 128) *reinterpret_cast<uint2*>(maskb + tid * 8) = vm;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * 256;
        const int g = idx & 3, px = (idx >> 2) & 127, ni = idx >> 9;
        float yv[8];
        unpack16<bf16_t>(vy[i], yv);
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (yv[e] > 0.f ? 1u : 0u) << e;
        maskb[px * 8 + ni * 4 + g] = (unsigned char)bits;
      }
    }
    __syncthreads();
    // B. the tap matrix of this tile's border class (rows / columns next to a reflect border or beyond the image differ from the interior ones)
    if (rebuild) {
      key_a = ka; key_b = kb;
      for (int i = tid; i < TT_BYTES / 16; i += 256) *reinterpret_cast<uint4*>(ttb + i * 16) = make_uint4(0, 0, 0, 0);
      __syncthreads();
      if (tid < 128) {
        const int ty = tid >> 4, tx = tid & 15;
        const float4 wr = wrow[ty], wc = wcol[tx];
        const float wrv[3] = {wr.x, wr.y, wr.z}, wcv[3] = {wc.x, wc.y, wc.z};
        const int wy = ((ty + 1) >> 1) + 1, wx = ((tx + 1) >> 1) + 1;      // window position of the (oy0, ox0) tap: tile-invariant (a0, b0 are multiples of 8 / 16)
        bf16_t* trow = reinterpret_cast<bf16_t*>(ttb + tid * TP);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float w = wrv[dy] * wcv[dx];
            if (w != 0.f) trow[(wy + dy - 1) * WW + wx + dx - 1] = f32_to_bf16(w);
          }
      }
      __syncthreads();
    }
    // C. this stage's registers are free again: the loads of the tile DEPTH ahead
    if (prefetch) tile_load(S, pa0, pb0);
    // D. transposed blur of the tile: g[channel][pixel] = window^T (A, transposing read) x Tt^T (B, 16 contiguous bytes per lane)
    f32x16_t gacc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int j = 0; j < 16; ++j) gacc[cb][j] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NQ / 16; ++ks) {
      const uint4 b = *reinterpret_cast<const uint4*>(ttb + pl * TP + (ks * 16 + 8 * hh) * 2);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const uint4 a = tr16(winb + cb * PLANE + ks * 16 * ROWB + trA, 4 * ROWB);
        gacc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), gacc[cb], 0, 0, 0);
      }
    }
    // E. LeakyReLU' from the sign word, bias sums, bf16, into the dO tile [ni][pixel][32 channels] the weight-gradient loop reads
    {
      const uint2 m64 = *reinterpret_cast<const uint2*>(maskb + pl * 8);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const unsigned mlo = (cb ? m64.y : m64.x) >> (4 * hh);     // bit 8 q4 + e of mlo = this lane's channel 8 q4 + 4 hh + e of the block
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gv = gacc[cb][4 * q4 + e];
            v[e] = (mlo & (1u << (8 * q4 + e))) ? gv : gv * slope;
            bsum[cb][4 * q4 + e] += v[e];
          }
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(smem + (cb * 128 + pl) * ROWB + (8 * q4 + 4 * hh) * 2) = o;
        }
      }
    }
    __syncthreads();
    // F. the weight-gradient MFMAs of tfc_wgrad_c8_kernel
    {
      const unsigned char* hrow = halo0 + cur * HALO_BYTES + wave * pd.hw * 16 + trB;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        const uint4 b = tr16(hrow + kt * pd.hw * 16, 4 * 16);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const uint4 a = tr16(smem + ni * 128 * ROWB + kt * 16 * ROWB + trA, 4 * ROWB);
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[ni], 0, 0, 0);
        }
      }
    }
    cur ^= 1;
  };
  // the workgroup's tiles in order; (tyb, txb): the tile being processed, (pyb, pxb): the tile DEPTH ahead
  int txb = (t0 - wimg * tpi) % d.tiles_x, tyb = (t0 - wimg * tpi) / d.tiles_x;
  int pxb = txb, pyb = tyb;
  auto step = [&](int& xb, int& yb) { if (++xb == d.tiles_x) { xb = 0; ++yb; } };
  if (t0 < t1) { tile_load(S0, pyb * TFC_TILE_H, pxb * TFC_TILE_W); step(pxb, pyb); }
  if (DEPTH == 2 && t0 + 1 < t1) { tile_load(S1, pyb * TFC_TILE_H, pxb * TFC_TILE_W); step(pxb, pyb); }
  for (int tl = t0; tl < t1; tl += DEPTH) {
    body(S0, tyb * TFC_TILE_H, txb * TFC_TILE_W, tl + DEPTH < t1, pyb * TFC_TILE_H, pxb * TFC_TILE_W);
    step(txb, tyb); step(pxb, pyb);
    if (DEPTH == 2 && tl + 1 < t1) {
      body(S1, tyb * TFC_TILE_H, txb * TFC_TILE_W, tl + 1 + DEPTH < t1, pyb * TFC_TILE_H, pxb * TFC_TILE_W);
      step(txb, tyb); step(pxb, pyb);
    }
  }
  if (rstats) {
    // bias-gradient sums in a fixed order: the 32 lanes of a half-wave hold the same channels for different pixels -> butterfly over lane bits 0..4,
    // the four waves meet in four LDS slots, 64 threads add the slots in wave order
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float v = bsum[cb][j];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        if ((lane & 31) == 0) sbias[wave * 64 + cb * 32 + (j & 3) + 8 * (j >> 2) + 4 * hh] = v;
      }
    __syncthreads();
    if (tid < 64) rstats[(size_t)sp * 64 + tid] = ((sbias[tid] + sbias[64 + tid]) + sbias[128 + tid]) + sbias[192 + tid];
  }
  float4* ps = slab + ((size_t)sp * 4 + wave) * (2 * 4 * 64) + lane;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      ps[(ni * 4 + q4) * 64] = make_float4(acc[ni][4 * q4], acc[ni][4 * q4 + 1], acc[ni][4 * q4 + 2], acc[ni][4 * q4 + 3]);
}

// position = ((ky * 2 + ni) * 4 + q4) * 64 + lane of the 2048 float4 of a workgroup slab. A block owns 16 positions; its 16 slab-lanes take the slabs
// sp = l, l + 16, ... ascending and meet in LDS in lane order: a fixed summation order and ONE owner per accumulator element (no atomics).
// grad != nullptr: the sums go straight into the torch-layout gradient grad[n * sn + c * sc + ky * 4 + kx] (=/+=) and the accumulator is not touched
// (one launch instead of reduce + tfc_wgrad_finish_kernel; the gradient is 64 x 8 x 16 floats)
__global__ void __launch_bounds__(256)
tfc_wgrad_c8_reduce_kernel(const float4* __restrict__ slab, float* acc, int nsplit, int Nn_real, int Cw_real, float* grad, long long sn, long long sc,
                           int accumulate) {
  __shared__ float4 red[16][16];
  const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int pos = blockIdx.x * 16 + pl;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sp = sl; sp < nsplit; sp += 16) {
    const float4 v = slab[(size_t)sp * 2048 + pos];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[sl][pl] = s;
  __syncthreads();
  if (sl != 0) return;
#pragma unroll
  for (int i = 1; i < 16; ++i) { const float4 v = red[i][pl]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const int lane = pos & 63, q4 = (pos >> 6) & 3, ni = (pos >> 8) & 1, ky = pos >> 9;
  const int col = lane & 31, kx = col >> 3, c = col & 7;
  const int n0 = ni * 32 + 8 * q4 + 4 * (lane >> 5);
  if (c >= Cw_real) return;
  const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n0 + e < Nn_real) {
      if (grad) {
        float* g = grad + (long long)(n0 + e) * sn + (long long)c * sc + ky * 4 + kx;
        *g = accumulate ? *g + sv[e] : sv[e];
      } else {
        acc[((size_t)(ky * 4 + kx) * Nn_real + n0 + e) * Cw_real + c] += sv[e];
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// weight packing: torch fp32 weight -> operand stream  wp[((gs*NB32 + nb)*64 + lane)] (16 bytes each)
//   gs enumerates (chunk, plane, k-substep); lane = (rn, h); unit u = 2*s + h; tap = u / UPP; g = u % UPP;
//   element e <-> channel c = chunk*CK + g*UE + e;  value = W[n*sn + c*sc + slot(tap)] * (*scale)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
tfc_pack_w_kernel(const TfcGather d, const float* __restrict__ w, const float* __restrict__ scale_ptr,
                  uint4* __restrict__ wp, int NB32, int Nreal, int Creal, long long sn, long long sc, int total_units) {
  constexpr int ES = sizeof(T);
  constexpr int UE = 16 / ES;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total_units) return;
  int n, mask, c0;
  tfc_pack_locate(d, ES, NB32, idx, &n, &mask, &c0);
  const float scale = scale_ptr ? *scale_ptr : 1.f;
  float v[UE];
#pragma unroll
  for (int e = 0; e < UE; ++e) {
    const int c = c0 + e;
    float a = 0.f;
    if (n < Nreal && c < Creal)
      for (int m = mask; m; m &= m - 1) a += w[(long long)n * sn + (long long)c * sc + (__ffs(m) - 1)];
    v[e] = a * scale;
  }
  wp[idx] = pack16<T>(v);
}

// One launch packs every operand stream of a network: a device-resident plan (built once by the host for fixed geometry and
// fixed buffers) lists the jobs; a workgroup finds its job from the prefix sums of the unit counts.
struct TfcPackJob {
  TfcGather d;
  const float* w;
  uint4* wp;
  long long sn, sc;
  int NB32, Nreal, Creal, units;                                  // units = 16-byte units of this job
  int first_block, threads;                                       // first 256-thread workgroup of this job in the planned grid; its thread count
};
// One thread per (output channel n, 16-byte channel unit o): it reads the 16 filter taps of its UE (n, c) pairs ONCE -- 64 contiguous
// bytes each, both torch layouts keep the taps innermost -- and emits one packed unit per gather tap (collapsed taps: the sum of
// their filter taps). n runs fastest over the threads, so a wave's stores are 512-byte runs of the stream (lane field = n & 31).
// The unit-per-thread gather this replaces fetched every 64-byte tap row sixteen times through L2 (195 us for the generator).
template <typename T>
__global__ void __launch_bounds__(256)
tfc_pack_planned_kernel(const TfcPackJob* __restrict__ jobs, int njobs) {
  constexpr int ES = sizeof(T);
  constexpr int UE = 16 / ES;
  int lo = 0, hi = njobs - 1;                                     // last job with first_block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TfcPackJob& j = jobs[lo];
  const int t = ((int)blockIdx.x - j.first_block) * 256 + threadIdx.x;
  if (t >= j.threads) return;
  const int Npad = j.NB32 * 32;
  const int n = t % Npad, o = t / Npad;
  const int PB = tfc_pb(j.d.Cin_pad, ES), UPP = PB >> 4;
  const int cc = o / UPP, g = o - cc * UPP;
  const int c0 = o * UE;
  float row[UE][16];
#pragma unroll
  for (int e = 0; e < UE; ++e) {
    const bool ok = n < j.Nreal && c0 + e < j.Creal;
    const float4* pr = reinterpret_cast<const float4*>(j.w + (long long)n * j.sn + (long long)(c0 + e) * j.sc);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = ok ? pr[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
      row[e][q4 * 4 + 0] = v.x; row[e][q4 * 4 + 1] = v.y; row[e][q4 * 4 + 2] = v.z; row[e][q4 * 4 + 3] = v.w;
    }
  }
  int per_chunk = 0;
  for (int pl = 0; pl < j.d.nplanes; ++pl) per_chunk += tfc_nsub(j.d.plane[pl].ntaps, PB);
  int gs_base = cc * per_chunk;
  const int lane_lo = n & 31, nb = n >> 5;
  // Fast path: every tap stands for exactly ONE filter slot (all layers of the path except the upsample-conv head, whose taps are sums): walk the
  // 16 slots with compile-time register indices and send each to the tap that owns it -- the general path below selects each tap's value out of
  // the 16 with a mask scan, 2,048 select-adds per thread (the kernel was VALU-bound at 2.5x its HBM time).
  bool single = true;
  for (int pl = 0; pl < j.d.nplanes && single; ++pl)
    for (int tap = 0; tap < j.d.plane[pl].ntaps && single; ++tap) single = __popc(j.d.plane[pl].tap_mask[tap]) == 1;
  if (single) {
    int base = gs_base;
    for (int pl = 0; pl < j.d.nplanes; ++pl) {
      const TfcPlane& p = j.d.plane[pl];
      for (int tap = 0; tap < p.ntaps; ++tap) {
        const int slot = __ffs(p.tap_mask[tap]) - 1, u = tap * UPP + g;
        uint4* dst = reinterpret_cast<uint4*>(j.wp) + ((size_t)(base + (u >> 1)) * j.NB32 + nb) * 64 + lane_lo + 32 * (u & 1);
        float v[UE];
        switch (slot) {                                          // compile-time register index per case: no dynamic indexing, no scan
#define TFC_SLOT(S) case S: { _Pragma("unroll") for (int e = 0; e < UE; ++e) v[e] = row[e][S]; } break;
          TFC_SLOT(0) TFC_SLOT(1) TFC_SLOT(2) TFC_SLOT(3) TFC_SLOT(4) TFC_SLOT(5) TFC_SLOT(6) TFC_SLOT(7)
          TFC_SLOT(8) TFC_SLOT(9) TFC_SLOT(10) TFC_SLOT(11) TFC_SLOT(12) TFC_SLOT(13) TFC_SLOT(14) default: TFC_SLOT(15)
#undef TFC_SLOT
        }
        *dst = pack16<T>(v);
      }
      base += tfc_nsub(p.ntaps, PB);
    }
    return;
  }
  for (int pl = 0; pl < j.d.nplanes; ++pl) {
    const TfcPlane& p = j.d.plane[pl];
    for (int tap = 0; tap < p.ntaps; ++tap) {
      const int u = tap * UPP + g;
      const int mask = p.tap_mask[tap];
      float v[UE];
#pragma unroll
      for (int e = 0; e < UE; ++e) {
        float a = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) a += ((mask >> sl) & 1) ? row[e][sl] : 0.f;
        v[e] = a;
      }
      j.wp[((size_t)(gs_base + (u >> 1)) * j.NB32 + nb) * 64 + lane_lo + 32 * (u & 1)] = pack16<T>(v);
    }
    gs_base += tfc_nsub(p.ntaps, PB);
  }
}
hipError_t tfc_launch_pack_planned(int dt, const void* plan_dev, int njobs, int nblocks, hipStream_t st) {
  if (dt == TFC_DT_BF16) TFC_LAUNCH((tfc_pack_planned_kernel<bf16_t>), dim3(nblocks), dim3(256), 0, st, (const TfcPackJob*)plan_dev, njobs);
  else TFC_LAUNCH((tfc_pack_planned_kernel<float>), dim3(nblocks), dim3(256), 0, st, (const TfcPackJob*)plan_dev, njobs);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Transposed-convolution weight gradient, all four sub-pixel phases in ONE launch (bf16).
//   dW[ci][co][ky][kx] = sum_{a,b} x[a][b][ci] * dy[2a-1+ky][2b-1+kx][co]
// Phase (py,px) of the output grid owns the taps ky = 1-py+2jy, kx = 1-px+2jx and reads x at (a+py-jy, b+px-jx). The per-phase
// launches re-read the input halo four times and moved 118 FLOP per byte; here a workgroup owns 32 n (= co) x 32 c (= ci), stages
// the (8+2) x (16+2) halo of x ONCE plus the four parity sub-grids of the 16 x 32 patch of dy, and wave w = phase (py,px) runs its
// 2 x 2 taps with the sliding two-row window of tfc_wgrad22_kernel (1 A + 2 new B fragments per 4 MFMAs): 380 FLOP per byte, one
// launch and one slab set per layer instead of four. Split-K partials go to slabs (tfc_wgrad_reduce_kernel kind 2).
// ---------------------------------------------------------------------------------------------------
// UP = true: the same structure for the upsample(2x nearest) + ZeroPad(1,0,1,0) + conv head (generator `final`, P16:150-157):
//   dW[co][ci][ky][kx] = sum dy[2a+py][2b+px][co] * x[a + off(py,ky)][b + off(px,kx)][ci],  off(0,.) = {-1,-1,0,0}, off(1,.) = {-1,0,0,1}
// -- phase (py,px) has (2+py) x (2+px) distinct source offsets (collapsed taps; the reduce kernel adds each to all the filter taps it
// stands for), read from the same 10 x 18 halo.
template <bool UP>
__global__ void __launch_bounds__(256, UP ? 2 : 3)
tfc_wgradT_kernel(const bf16_t* __restrict__ x, int IH, int IW, int x_pitch, int Cin_pad, const bf16_t* __restrict__ dy, int dy_pitch,
                  int Nn_pad, int nimg, float4* slab, int nbw, int ncb, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWB = 64;                                       // 32 channels x 2 bytes per LDS pixel row
  constexpr int HH = TFC_TILE_H + 2, HW = TFC_TILE_W + 2;
  constexpr int DO_BYTES = 4 * 128 * ROWB;                       // [phase][px][32 n]
  constexpr int NDO = DO_BYTES / 16 / 256, NHA = (HH * HW * 4 + 255) / 256;   // 8 and 3 units per thread
  constexpr int NCM = UP ? 3 : 2, NA = NCM * NCM;                // accumulator tiles per wave: (row offset, column offset)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int OH = 2 * IH, OW = 2 * IW;
  const int tiles_y = (IH + TFC_TILE_H - 1) / TFC_TILE_H, tiles_x = (IW + TFC_TILE_W - 1) / TFC_TILE_W;
  const int ntiles = nimg * tiles_y * tiles_x;

  const int bid = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % (nbw * ncb);
  const int sp = bid / (nbw * ncb);
  const int cb = pair % ncb, nb = pair / ncb;

  f32x16_t acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;

  uint4 vdo[NDO], vha[NHA];
  auto tile_load = [&](int tl) {
    int t = tl;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y;
    const int img = t / tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < NDO; ++i) {
      const int idx = tid + i * 256;                             // ((phase * 128) + pixel) * 4 + g
      const int g = idx & 3, pxl = (idx >> 2) & 127, ph = idx >> 9;
      const int a = a0 + (pxl >> 4), b = b0 + (pxl & 15);
      const int n0 = nb * 32 + g * 8;
      vdo[i] = make_uint4(0, 0, 0, 0);
      if (a < IH && b < IW && n0 < Nn_pad)
        vdo[i] = *reinterpret_cast<const uint4*>(dy + ((size_t)(img * OH + 2 * a + (ph >> 1)) * OW + 2 * b + (ph & 1)) * dy_pitch + n0);
    }
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;                             // pixel * 4 + g
      vha[i] = make_uint4(0, 0, 0, 0);
      if (idx < HH * HW * 4) {
        const int g = idx & 3, pix = idx >> 2;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int y = a0 - 1 + hy, xx = b0 - 1 + hx;
        const int c0 = cb * 32 + g * 8;
        if (y >= 0 && y < IH && xx >= 0 && xx < IW && c0 < Cin_pad)
          vha[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(img * IH + y) * IW + xx) * x_pitch + c0);
      }
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < NDO; ++i) *reinterpret_cast<uint4*>(smem + (tid + i * 256) * 16) = vdo[i];
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;
      if (idx < HH * HW * 4) *reinterpret_cast<uint4*>(smem + DO_BYTES + idx * 16) = vha[i];
    }
  };

  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trLane = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + 4 * ROWB));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  // Accumulator (r, c): halo row kt + rb + r, column shift cs + c, with a sliding window of NR rows (wave-uniform NR, NC in {2, 3}).
  //   transposed conv: rb = py, cs = px, 2 x 2 (filter tap jy = 1 - r, jx = 1 - c);  upsample conv: rb = cs = 0, (2+py) x (2+px)
  const int rb = UP ? 0 : py, cs = UP ? 0 : px;
  constexpr int rowb = HW * ROWB;
  auto compute_v = [&](auto nr_tag, auto nc_tag) {
    constexpr int NR = decltype(nr_tag)::value, NC = decltype(nc_tag)::value;
    const unsigned char* acol = smem + wave * 128 * ROWB + trLane;
    const unsigned char* hcol = smem + DO_BYTES + (rb * HW + cs) * ROWB + trLane;
    uint4 rw[NR][NC];
#pragma unroll
    for (int r = 0; r < NR - 1; ++r)
#pragma unroll
      for (int c = 0; c < NC; ++c) rw[r][c] = tr16(hcol + r * rowb + c * ROWB);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int c = 0; c < NC; ++c) rw[NR - 1][c] = tr16(hcol + (kt + NR - 1) * rowb + c * ROWB);
      const bf16x8_t av = __builtin_bit_cast(bf16x8_t, tr16(acol + kt * 16 * ROWB));
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[r * NCM + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, rw[r][c]), acc[r * NCM + c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < NR - 1; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) rw[r][c] = rw[r + 1][c];
    }
  };
  // upsample conv: every wave runs the full 3 x 3 offset grid; the offsets a phase does not have (row 2 for py = 0, column 2 for
  // px = 0) only fill accumulators that the reduce kernel ignores -- cheaper than four differently shaped code paths
  auto compute = [&]() { compute_v(std::integral_constant<int, NCM>{}, std::integral_constant<int, NCM>{}); };

  int tl = sp;
  if (tl < ntiles) { tile_load(tl); tile_store(); }
  __syncthreads();
  for (; tl < ntiles; tl += nsplit) {
    const bool more = (tl + nsplit) < ntiles;
    if (!UP && more) tile_load(tl + nsplit);                     // next tile -> registers while this one is multiplied
    compute();
    __syncthreads();                                             // single LDS image: everyone is done reading it
    if (UP && more) tile_load(tl + nsplit);                      // nine accumulator tiles leave no room for the staging registers: load late
    if (more) tile_store();
    __syncthreads();
  }

  float4* ps = slab + ((size_t)bid * 4 + wave) * (NA * 4 * 64) + lane;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      ps[(a * 4 + q4) * 64] = make_float4(acc[a][4 * q4], acc[a][4 * q4 + 1], acc[a][4 * q4 + 2], acc[a][4 * q4 + 3]);
}

// ---------------------------------------------------------------------------------------------------
// Round 3: the same transposed-convolution weight gradient with a 32 n x 64 c workgroup tile. tfc_wgradT_kernel<false> is bound by operand re-streaming:
// a 32 x 32 tile re-reads dy Cin/32 times and x Cout/32 times from L2 (670 MB per launch at 256 -> 64 @ 64^2, 6.7 TB/s at 100 us) and issues six
// transposing LDS reads per four MFMAs. Here wave w = phase still owns its 2 x 2 taps, but for TWO 32-channel halves of x: eight accumulator tiles per wave
// (128 VGPRs), one A fragment serves eight MFMAs, dy is streamed Cin/64 times. Two workgroups per CU (55 KB of LDS). The slabs keep the logical numbering
// of the 32 x 32 kernel (pair = nb * ncb + cb, cb = 2 cb2 + half), so the reduce kernels are unchanged.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_wgradT2_kernel(const bf16_t* __restrict__ x, int IH, int IW, int x_pitch, int Cin_pad, const bf16_t* __restrict__ dy, int dy_pitch,
                   int Nn_pad, int nimg, float4* slab, int nbw, int ncb, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWB = 64;
  constexpr int HH = TFC_TILE_H + 2, HW = TFC_TILE_W + 2;
  constexpr int DO_BYTES = 4 * 128 * ROWB;                       // [phase][px][32 n]
  constexpr int HPLANE = HH * HW * ROWB;                         // one 32-channel half of the halo
  constexpr int NDO = DO_BYTES / 16 / 256, NHA = (HH * HW * 8 + 255) / 256;   // 8 and 6 units per thread
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int OH = 2 * IH, OW = 2 * IW;
  const int tiles_y = (IH + TFC_TILE_H - 1) / TFC_TILE_H, tiles_x = (IW + TFC_TILE_W - 1) / TFC_TILE_W;
  const int ntiles = nimg * tiles_y * tiles_x;
  const int ncb2 = ncb >> 1;

  const int bid = tfc_xcd_remap(blockIdx.x, gridDim.x);
  const int pair2 = bid % (nbw * ncb2);
  const int sp = bid / (nbw * ncb2);
  const int cb2 = pair2 % ncb2, nb = pair2 / ncb2;

  f32x16_t acc[8];                                               // [half][row offset][column offset]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;

  uint4 vdo[NDO], vha[NHA];
  auto tile_load = [&](int tl) {
    int t = tl;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y;
    const int img = t / tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < NDO; ++i) {
      const int idx = tid + i * 256;                             // ((phase * 128) + pixel) * 4 + g
      const int g = idx & 3, pxl = (idx >> 2) & 127, ph = idx >> 9;
      const int a = a0 + (pxl >> 4), b = b0 + (pxl & 15);
      const int n0 = nb * 32 + g * 8;
      vdo[i] = make_uint4(0, 0, 0, 0);
      if (a < IH && b < IW && n0 < Nn_pad)
        vdo[i] = *reinterpret_cast<const uint4*>(dy + ((size_t)(img * OH + 2 * a + (ph >> 1)) * OW + 2 * b + (ph & 1)) * dy_pitch + n0);
    }
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;                             // (pixel * 2 + half) * 4 + g: a pixel's 128 bytes are one contiguous run
      vha[i] = make_uint4(0, 0, 0, 0);
      if (idx < HH * HW * 8) {
        const int g = idx & 3, hc = (idx >> 2) & 1, pix = idx >> 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int y = a0 - 1 + hy, xx = b0 - 1 + hx;
        const int c0 = (2 * cb2 + hc) * 32 + g * 8;
        if (y >= 0 && y < IH && xx >= 0 && xx < IW && c0 < Cin_pad)
          vha[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(img * IH + y) * IW + xx) * x_pitch + c0);
      }
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < NDO; ++i) *reinterpret_cast<uint4*>(smem + (tid + i * 256) * 16) = vdo[i];
#pragma unroll
    for (int i = 0; i < NHA; ++i) {
      const int idx = tid + i * 256;
      if (idx < HH * HW * 8) {
        const int g = idx & 3, hc = (idx >> 2) & 1, pix = idx >> 3;
        *reinterpret_cast<uint4*>(smem + DO_BYTES + hc * HPLANE + (pix * 4 + g) * 16) = vha[i];
      }
    }
  };

  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trLane = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + 4 * ROWB));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  constexpr int rowb = HW * ROWB;
  auto compute = [&]() {
    const unsigned char* acol = smem + wave * 128 * ROWB + trLane;
    const unsigned char* hcol = smem + DO_BYTES + (py * HW + px) * ROWB + trLane;    // halo row kt + py + r, column shift px + c (filter tap jy = 1 - r, jx = 1 - c)
    uint4 rw[2][2][2];                                           // [half][window row][column]
#pragma unroll
    for (int hc = 0; hc < 2; ++hc)
#pragma unroll
      for (int c = 0; c < 2; ++c) rw[hc][0][c] = tr16(hcol + hc * HPLANE + c * ROWB);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int hc = 0; hc < 2; ++hc)
#pragma unroll
        for (int c = 0; c < 2; ++c) rw[hc][1][c] = tr16(hcol + hc * HPLANE + (kt + 1) * rowb + c * ROWB);
      const bf16x8_t av = __builtin_bit_cast(bf16x8_t, tr16(acol + kt * 16 * ROWB));
#pragma unroll
      for (int hc = 0; hc < 2; ++hc)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[hc * 4 + r * 2 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, rw[hc][r][c]), acc[hc * 4 + r * 2 + c], 0, 0, 0);
#pragma unroll
      for (int hc = 0; hc < 2; ++hc)
#pragma unroll
        for (int c = 0; c < 2; ++c) rw[hc][0][c] = rw[hc][1][c];
    }
  };

  int tl = sp;
  if (tl < ntiles) { tile_load(tl); tile_store(); }
  __syncthreads();
  for (; tl < ntiles; tl += nsplit) {
    const bool more = (tl + nsplit) < ntiles;
    if (more) tile_load(tl + nsplit);                            // next tile -> registers while this one is multiplied
    compute();
    __syncthreads();                                             // single LDS image: everyone is done reading it
    if (more) tile_store();
    __syncthreads();
  }
  // slabs in the numbering of the 32 x 32 kernel: logical workgroup (sp, nb, cb = 2 cb2 + half)
#pragma unroll
  for (int hc = 0; hc < 2; ++hc) {
    const int lbid = sp * (nbw * ncb) + nb * ncb + 2 * cb2 + hc;
    float4* ps = slab + ((size_t)lbid * 4 + wave) * (4 * 4 * 64) + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        ps[(a * 4 + q4) * 64] = make_float4(acc[hc * 4 + a][4 * q4], acc[hc * 4 + a][4 * q4 + 1], acc[hc * 4 + a][4 * q4 + 2], acc[hc * 4 + a][4 * q4 + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------
// Weight gradient of the generator head (Upsample x2 -> ZeroPad -> Conv2d(128, C <= 8, 4); P16:150-157), bf16. tfc_wgradT_kernel<true> spends a
// 32-row MFMA tile on the 3 (padded: 8) output channels of ONE sub-pixel phase per wave: 4.7 M MFMAs, 147 us, MFMA-bound on padding. Here the four
// phases x 8 padded channels ARE the 32 rows (row = phase * 8 + oc): dy of a tile is staged as [input-grid pixel][py][px][oc8] = 64 bytes per
// pixel, so the transposing read that the other wgrad kernels use yields the merged A fragment directly; wave w owns input channels 32w..32w+31
// and slides the 3 x 3 source-offset window down the 10 x 18 x 128-channel halo (3 new B fragments + 1 A fragment per 9 MFMAs). A quarter of the
// MFMAs and dy is read once instead of four times: the bound becomes the 134 MB x stream. Accumulators of (phase, offset) pairs a phase does
// not have are ignored by the reduce pass (as for tfc_wgradT_kernel<true>).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
tfc_wgrad_head_kernel(const bf16_t* __restrict__ x, int IH, int IW, int x_pitch, const bf16_t* __restrict__ dy, int dy_pitch, int nimg,
                      float4* __restrict__ slab, int nsplit) {
  constexpr int HH = TFC_TILE_H + 2, HW = TFC_TILE_W + 2;          // 10 x 18 halo of x, 128 channels = 256 B per pixel
  constexpr int DY_BYTES = 128 * 64;                             // [pixel][py][px][oc8]
  constexpr int HB = 256, ROWB = 64;
  constexpr int NHA = (HH * HW * 16 + 255) / 256;                // 12 halo units per thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[DY_BYTES + HH * HW * HB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int OH = 2 * IH, OW = 2 * IW;
  const int tiles_y = (IH + TFC_TILE_H - 1) / TFC_TILE_H, tiles_x = (IW + TFC_TILE_W - 1) / TFC_TILE_W;
  const int ntiles = nimg * tiles_y * tiles_x;
  const int sp = tfc_xcd_remap(blockIdx.x, gridDim.x);
  f32x16_t acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;

  auto tile_stage = [&](int tl) {                                 // global -> LDS (through registers, a few units at a time: 144 accumulator VGPRs are live)
    int t = tl;
    const int txb = t % tiles_x; t /= tiles_x;
    const int tyb = t % tiles_y;
    const int img = t / tiles_y;
    const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;                             // pixel * 4 + phase
      const int ph = idx & 3, pxl = idx >> 2;
      const int a = a0 + (pxl >> 4), b = b0 + (pxl & 15);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a < IH && b < IW) v = *reinterpret_cast<const uint4*>(dy + ((size_t)(img * OH + 2 * a + (ph >> 1)) * OW + 2 * b + (ph & 1)) * dy_pitch);
      *reinterpret_cast<uint4*>(smem + idx * 16) = v;
    }
#pragma unroll
    for (int i0 = 0; i0 < NHA; i0 += 4) {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + (i0 + i) * 256;                    // pixel * 16 + unit
        v[i] = make_uint4(0, 0, 0, 0);
        if (i0 + i < NHA && idx < HH * HW * 16) {
          const int u = idx & 15, pix = idx >> 4;
          const int hy = pix / HW, hx = pix - hy * HW;
          const int y = a0 - 1 + hy, xx = b0 - 1 + hx;
          if (y >= 0 && y < IH && xx >= 0 && xx < IW) v[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(img * IH + y) * IW + xx) * x_pitch + u * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + (i0 + i) * 256;
        if (i0 + i < NHA && idx < HH * HW * 16) *reinterpret_cast<uint4*>(smem + DY_BYTES + idx * 16) = v[i];
      }
    }
  };
  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, p = li & 3;
  const int trA = (8 * hk + q) * ROWB + cb16 * 32 + p * 8;
  const int trB = (8 * hk + q) * HB + wave * 64 + cb16 * 32 + p * 8;
  auto tr16 = [&](const unsigned char* p0, int rowb4) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + rowb4));
    uint4 r;
    r.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  };
  auto compute = [&]() {
    const unsigned char* acol = smem + trA;
    const unsigned char* hcol = smem + DY_BYTES + trB;
    constexpr int rowb = HW * HB;
    uint4 rw[3][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) rw[r][c] = tr16(hcol + r * rowb + c * HB, 4 * HB);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rw[2][c] = tr16(hcol + (kt + 2) * rowb + c * HB, 4 * HB);
      const bf16x8_t av = __builtin_bit_cast(bf16x8_t, tr16(acol + kt * 16 * ROWB, 4 * ROWB));
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          acc[r * 3 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8_t, rw[r][c]), acc[r * 3 + c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) rw[r][c] = rw[r + 1][c];
    }
  };
  for (int tl = sp; tl < ntiles; tl += nsplit) {
    tile_stage(tl);
    __syncthreads();
    compute();
    __syncthreads();
  }
  float4* ps = slab + ((size_t)sp * 4 + wave) * (9 * 4 * 64) + lane;
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
      ps[(a * 4 + q4) * 64] = make_float4(acc[a][4 * q4], acc[a][4 * q4 + 1], acc[a][4 * q4 + 2], acc[a][4 * q4 + 3]);
}
// position = ((w * 9 + a) * 4 + q4) * 64 + lane of the 9216 float4 of a workgroup slab. Stage 1 sums the slabs position by position (16 positions x 16
// slab-lanes per block, slab-lanes take sp = l, l + 16, ... ascending and meet in LDS in lane order) and leaves the total IN PLACE in slab 0.
__global__ void __launch_bounds__(256)
tfc_wgrad_head_reduce_kernel(float4* __restrict__ slab, int nsplit) {
  __shared__ float4 red[16][16];
  const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int pos = blockIdx.x * 16 + pl;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sp = sl; sp < nsplit; sp += 16) {
    const float4 v = slab[(size_t)sp * 9216 + pos];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[sl][pl] = s;
  __syncthreads();
  if (sl != 0) return;
#pragma unroll
  for (int i = 1; i < 16; ++i) { const float4 v = red[i][pl]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  slab[pos] = s;
}
// Stage 2 (replaces tfc_wgrad_finish_kernel for the head): one thread per (oc, c) GATHERS its 16 filter taps -- row (phase * 8 + oc) of source offset
// a = ir * 3 + ic feeds every filter tap that collapses onto that offset in that phase, so a tap is the sum of four rows (one per phase, phase order);
// round 2 scattered them with float atomics.
__global__ void __launch_bounds__(256)
tfc_wgrad_head_finish_kernel(const float4* __restrict__ S, float* __restrict__ grad, int Nn_real, int Cw_real, long long sn, long long sc, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Nn_real * Cw_real) return;
  const int c = idx % Cw_real, oc = idx / Cw_real;
  const int w = c >> 5, lane = (oc >> 2) * 32 + (c & 31), e = oc & 3;
  float v[16];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      float t = 0.f;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int wpy = ph >> 1, wpx = ph & 1;
        const int sy = wpy ? (ky == 0 ? 0 : (ky == 3 ? 2 : 1)) : (ky >> 1);   // source row offset + 1 of filter row ky in this phase
        const int sx = wpx ? (kx == 0 ? 0 : (kx == 3 ? 2 : 1)) : (kx >> 1);
        const float4 q = S[((size_t)(w * 9 + sy * 3 + sx) * 4 + ph) * 64 + lane];
        t += e == 0 ? q.x : (e == 1 ? q.y : (e == 2 ? q.z : q.w));
      }
      v[ky * 4 + kx] = t;
    }
  float4* g = reinterpret_cast<float4*>(grad + (long long)oc * sn + (long long)c * sc);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    float4 o = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
    if (accumulate) { const float4 p = g[q4]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    g[q4] = o;
  }
}

// Split-K reduction of the weight-gradient slabs. fp32 atomics run at ~1.3 TB/s on this chip (they execute at the memory side) and a
// one-round wgrad launch flushes the WHOLE chip's accumulator state (~67 MB): ~50 us of a ~130 us kernel. Instead every workgroup
// stores its accumulators, in register order, with plain 16-byte stores (~6 TB/s): slab[workgroup = sp*npairs + pair][wave][tile a][q][lane]
// = float4 of accumulator registers 4q..4q+3.  This kernel sums the nsplit slabs of one 64-lane row (one workgroup per row, its four
// waves take every fourth split, LDS combine) and adds the result to acc[slot][n][c] -- one thread per element, no atomics.
//   kind 0 (tfc_wgrad_kernel):   wave = tap % 4, tile a = (tap / 4) * 2 + ni;     n = nb*64 + ni*32 + row, c = cb*32 + (lane & 31)
//   kind 1 (tfc_wgrad22_kernel): wave = ch*2 + nh, tile a = tap_dy*2 + tap_dx;     n = nb*64 + nh*32 + row, c = (cb*2+ch)*32 + (lane & 31)
//   kind 2 (tfc_wgradT_kernel):  wave = phase, tile a = dyg*2 + dxg -> filter tap (1-py+2(1-dyg), 1-px+2(1-dxg)); n = nb*32 + row
//   row of register j = 4q+e in lane l:  e + 8q + 4*(l >> 5)
__global__ void __launch_bounds__(256)
tfc_wgrad_reduce_kernel(const float4* __restrict__ slab, float* acc, const TfcPlane pd, int kind, int T, int nsplit, int npairs, int ncbx,
                        int Nn_real, int Cw_real, int wave_only) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  int rowid = blockIdx.x;                                        // ((pair * 4 + wave) * T + a) * 4 + q   (wave_only >= 0: (pair * T + a) * 4 + q)
  const int q = rowid & 3; rowid >>= 2;
  const int a = rowid % T; rowid /= T;
  const int wave = wave_only >= 0 ? wave_only : (rowid & 3);
  const int pair = wave_only >= 0 ? rowid : (rowid >> 2);
  const size_t blk_units = (size_t)4 * T * 4 * 64;
  const float4* p0 = slab + ((size_t)pair * 4 + wave) * (T * 4 * 64) + (a * 4 + q) * 64 + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sp = k; sp < nsplit; sp += 4) {
    const float4 v = p0[(size_t)sp * npairs * blk_units];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  part[k][lane] = s;
  __syncthreads();
  if (k != 0) return;
#pragma unroll
  for (int i = 1; i < 4; ++i) { const float4 v = part[i][lane]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const int cbx = pair % ncbx, nb = pair / ncbx;
  int mask = 0, n0, c;
  if (kind == 3) {                                               // tfc_wgradT_kernel<true>: wave = phase, a = ir*3 + ic (collapsed taps of the upsample conv)
    const int wpy = wave >> 1, wpx = wave & 1, ir = a / 3, ic = a % 3;
    if (ir < 2 + wpy && ic < 2 + wpx)
      for (int ky = 0; ky < 4; ++ky)
        for (int kx = 0; kx < 4; ++kx) {
          const int sy = wpy ? (ky == 0 ? 0 : (ky == 3 ? 2 : 1)) : (ky >> 1);   // source row offset + 1 of filter row ky in this phase
          const int sx = wpx ? (kx == 0 ? 0 : (kx == 3 ? 2 : 1)) : (kx >> 1);
          if (sy == ir && sx == ic) mask |= 1 << (ky * 4 + kx);
        }
    n0 = nb * 32;
    c = cbx * 32 + (lane & 31);
  } else if (kind == 2) {                                        // tfc_wgradT_kernel: wave = phase (py,px), a = dyg*2 + dxg, 32 n per workgroup
    const int jy = 1 - (a >> 1), jx = 1 - (a & 1);
    mask = 1 << ((1 - (wave >> 1) + 2 * jy) * 4 + (1 - (wave & 1) + 2 * jx));
    n0 = nb * 32;
    c = cbx * 32 + (lane & 31);
  } else if (kind == 0) {
    const int tap = (a >> 1) * 4 + wave;
    if (tap < pd.ntaps) mask = pd.tap_mask[tap];
    n0 = nb * 64 + (a & 1) * 32;
    c = cbx * 32 + (lane & 31);
  } else {
    for (int t = 0; t < 4; ++t)
      if (pd.tap_dy[t] == (a >> 1) && pd.tap_dx[t] == (a & 1)) mask = pd.tap_mask[t];
    n0 = nb * 64 + (wave & 1) * 32;
    c = (cbx * 2 + (wave >> 1)) * 32 + (lane & 31);
  }
  n0 += 8 * q + 4 * (lane >> 5);
  if (c >= Cw_real) return;
  const float sv[4] = {s.x, s.y, s.z, s.w};
  for (int m = mask; m; m &= m - 1) {
    const int slot = __ffs(m) - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n0 + e < Nn_real) acc[((size_t)slot * Nn_real + n0 + e) * Cw_real + c] += sv[e];
  }
}

// fp32 accumulator [16 slots][Nn][Cw] -> torch-layout gradient  grad[n*sn + c*sc + slot].  One thread per (n, c): its 16 reads are
// coalesced across the wave slot by slot (c runs over the lanes), its 16 results are 64 contiguous bytes (both torch layouts keep
// the taps innermost), and the accumulator is re-zeroed with the same coalesced pattern.
__global__ void __launch_bounds__(256)
tfc_wgrad_finish_kernel(float* __restrict__ acc, float* __restrict__ grad, int Nn, int Cw,
                        long long sn, long long sc, int accumulate, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                // n * Cw + c
  if (idx >= total) return;
  const int c = idx % Cw, n = idx / Cw;
  float v[16];
#pragma unroll
  for (int slot = 0; slot < 16; ++slot) {
    float* pa = acc + (size_t)slot * total + idx;
    v[slot] = *pa;
    *pa = 0.f;                                                    // leave the accumulator zeroed for the next wgrad
  }
  float4* g = reinterpret_cast<float4*>(grad + (long long)n * sn + (long long)c * sc);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    float4 o = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
    if (accumulate) { const float4 p = g[q4]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    g[q4] = o;
  }
}

// Split-K reduction AND layout in one pass (kinds 0 and 2 with identity tap masks, i.e. the plain 4 x 4 convolution and the phase-fused
// transposed convolution): the reduce kernel above adds into the fp32 accumulator [slot][n][c] (read-modify-write) and tfc_wgrad_finish_kernel
// then reads that accumulator back, transposes it to the torch layout and re-zeroes it -- three passes over the gradient, 0.58 ms per step. Here a
// workgroup owns the four filter taps (ky = g, kx = 0..3) of a 64-lane row group: its eight waves split the nsplit slabs, combine in LDS,
// and wave 0 writes each (n, c) as ONE float4 = 16 contiguous bytes of the torch-layout gradient (both layouts keep the taps innermost).
// The accumulator is never touched (it stays all-zero, as the other paths leave it).
__global__ void __launch_bounds__(512)
tfc_wgrad_reduce_fin_kernel(const float4* __restrict__ slab, float* __restrict__ grad, int kind, int T, int nsplit, int npairs, int ncbx,
                            int Nn_real, int Cw_real, long long sn, long long sc, int accumulate) {
  __shared__ float4 part[7][4][64];
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  int b = blockIdx.x;
  const int g = b & 3; b >>= 2;
  const int q = b & 3; b >>= 2;
  int ni = 0;
  if (kind == 0) { ni = b & 1; b >>= 1; }
  const int pair = b;
  const size_t blk_units = (size_t)4 * T * 4 * 64;
  size_t row[4];
#pragma unroll
  for (int kx = 0; kx < 4; ++kx) {
    int wv, a;
    if (kind == 0) { wv = kx; a = g * 2 + ni; }                    // raster: tap = ti * 4 + wave, slab tile a = ti * 2 + ni
    else {                                                       // phase-fused convT: ky = 1 - py + 2 jy, kx = 1 - px + 2 jx; wave = phase, a = (1-jy)*2 + (1-jx)
      const int py = 1 - (g & 1), jy = g >> 1, px = 1 - (kx & 1), jx = kx >> 1;
      wv = py * 2 + px; a = (1 - jy) * 2 + (1 - jx);
    }
    row[kx] = (size_t)pair * blk_units + ((size_t)wv * T + a) * 256 + q * 64 + lane;
  }
  float4 s[4];
#pragma unroll
  for (int kx = 0; kx < 4; ++kx) s[kx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
  for (int sp = k; sp < nsplit; sp += 8) {
    const size_t o = (size_t)sp * npairs * blk_units;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      const float4 v = slab[o + row[kx]];
      s[kx].x += v.x; s[kx].y += v.y; s[kx].z += v.z; s[kx].w += v.w;
    }
  }
  if (k > 0) {
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) part[k - 1][kx][lane] = s[kx];
  }
  __syncthreads();
  if (k != 0) return;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) { const float4 v = part[i][kx][lane]; s[kx].x += v.x; s[kx].y += v.y; s[kx].z += v.z; s[kx].w += v.w; }
  const int cbx = pair % ncbx, nb = pair / ncbx;
  const int c = cbx * 32 + (lane & 31);
  const int n0 = (kind == 0 ? nb * 64 + ni * 32 : nb * 32) + 8 * q + 4 * (lane >> 5);
  if (c >= Cw_real) return;
  const float e0[4] = {s[0].x, s[1].x, s[2].x, s[3].x}, e1[4] = {s[0].y, s[1].y, s[2].y, s[3].y};
  const float e2[4] = {s[0].z, s[1].z, s[2].z, s[3].z}, e3[4] = {s[0].w, s[1].w, s[2].w, s[3].w};
  const float* ev[4] = {e0, e1, e2, e3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n0 + e >= Nn_real) continue;
    float4* gp = reinterpret_cast<float4*>(grad + (long long)(n0 + e) * sn + (long long)c * sc + g * 4);
    float4 o = make_float4(ev[e][0], ev[e][1], ev[e][2], ev[e][3]);
    if (accumulate) { const float4 p0 = *gp; o.x += p0.x; o.y += p0.y; o.z += p0.z; o.w += p0.w; }
    *gp = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// host launchers (internal C++ API used by api.hip)
// ---------------------------------------------------------------------------------------------------
int tfc_total_substeps(const TfcGather& d, int es) {
  const int PB = tfc_pb(d.Cin_pad, es);
  int per_chunk = 0;
  for (int pl = 0; pl < d.nplanes; ++pl) per_chunk += tfc_nsub(d.plane[pl].ntaps, PB);
  return per_chunk * ((d.Cin_pad * es) / PB);
}
int tfc_nb32(int nout) { return (nout + 31) / 32; }
// NB32 of the packed stream is rounded up to the widest workgroup N extent in use (4 blocks) so every config can read it
int tfc_nb32_padded(int nout) { int nb = tfc_nb32(nout); return nb <= 1 ? 1 : (nb <= 2 ? 2 : (nb + 3) / 4 * 4); }
size_t tfc_packed_bytes(const TfcGather& d, int es) {
  return (size_t)(tfc_total_substeps(d, es) + TFC_WPAD) * tfc_nb32_padded(d.Nout) * 64 * 16;
}

template <typename T>
static hipError_t launch_pack_t(const TfcGather& d, const float* w, const float* scale, void* wp, int Nreal, int Creal,
                                long long sn, long long sc, hipStream_t st) {
  const int NB32 = tfc_nb32_padded(d.Nout);
  const int total = tfc_total_substeps(d, sizeof(T)) * NB32 * 64;
  TFC_LAUNCH((tfc_pack_w_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, st, d, w, scale, (uint4*)wp, NB32,
                     Nreal, Creal, sn, sc, total);
  return hipGetLastError();
}
hipError_t tfc_launch_pack(int dt, const TfcGather& d, const float* w, const float* scale, void* wp, int Nreal, int Creal,
                           long long sn, long long sc, hipStream_t st) {
  return dt == TFC_DT_BF16 ? launch_pack_t<bf16_t>(d, w, scale, wp, Nreal, Creal, sn, sc, st)
                           : launch_pack_t<float>(d, w, scale, wp, Nreal, Creal, sn, sc, st);
}

// which compile-time tap pattern (if any) a descriptor matches (every plane must match the same one)
static int plane_pattern(const TfcPlane& p) {
  static const int R[7] = {0, 4, 2, 2, 2, 3, 3}, C[7] = {0, 4, 2, 2, 3, 2, 3}, REV[7] = {0, 0, 1, 0, 0, 0, 0};
  for (int pat = 1; pat <= 6; ++pat) {
    if (pat == 4 || pat == 5) continue;                           // 2x3 / 3x2 grids are no longer produced (the upsample conv uses 3x3 for every phase)
    if (p.ntaps != R[pat] * C[pat]) continue;
    bool ok = true;
    for (int t = 0; t < p.ntaps && ok; ++t) {
      const int r = t / C[pat], c = t % C[pat];
      ok = p.tap_dy[t] == (REV[pat] ? R[pat] - 1 - r : r) && p.tap_dx[t] == (REV[pat] ? C[pat] - 1 - c : c);
    }
    if (ok) return pat;
  }
  return 0;
}
static int match_pattern(const TfcGather& d, int es) {
  if (tfc_pb(d.Cin_pad, es) != 64) return 0;
  int pat = -1;
  for (int pl = 0; pl < d.nplanes; ++pl) {
    const int m = plane_pattern(d.plane[pl]);
    if (m == 0 || (pat != -1 && pat != m)) return 0;
    pat = m;
  }
  return pat < 0 ? 0 : pat;
}

template <typename T, int MT, int NT, int WM, int WN, int PAT>
static hipError_t launch_igemm_pat(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                                   float* stats, float* part_ws, float* out_nchw, const float* oscale, int flags, hipStream_t st) {
  constexpr int ES = sizeof(T);
  const int NB32 = tfc_nb32_padded(d.Nout);
  const int per_blk = NT * WN;
  const int nblkN = (tfc_nb32(d.Nout) + per_blk - 1) / per_blk;
  int maxhh = 0;
  for (int pl = 0; pl < d.nplanes; ++pl) maxhh = d.plane[pl].hh > maxhh ? d.plane[pl].hh : maxhh;
  const int PS = tfc_ps(tfc_pb(d.Cin_pad, ES));
  const int buf_bytes = maxhh * TFC_LDS_P * PS;
  int lds = 2 * buf_bytes;
  if (ES == 2) {                                                 // staged epilogue tile
    const int ep = 128 * (32 * NT * WN * ES + 16);
    lds = lds > ep ? lds : ep;
  }
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x * (d.ph_n > 1 ? d.ph_n : 1);
  const long long phase_wbytes = (long long)tfc_packed_bytes(d, ES);
  // InstanceNorm sums: every (tile, M-wave) stores its partial into part_ws[img][tile * WM + wm][Nout][2]; a fixed-order pass adds them to stats
  const int nparts = d.tiles_y * d.tiles_x * WM * (d.ph_n > 1 ? d.ph_n : 1);
  if (flags & TFC_EP_STATS) {
    if (!part_ws || (long long)d.nimg * nparts * d.Nout * 2 > (long long)TFC_PART_WS_FLOATS) return hipErrorInvalidValue;
  }
  TFC_LAUNCH((tfc_igemm_kernel<T, MT, NT, WM, WN, PAT>), dim3(ntiles * nblkN), dim3(256), lds, st, d,
                     (const T*)in, (const uint4*)wp, (T*)out, bias, part_ws, out_nchw, oscale, flags, NB32, nblkN, buf_bytes, phase_wbytes);
  if (flags & TFC_EP_STATS) return tfc_launch_part_reduce(part_ws, stats, d.nimg, nparts, 2 * d.Nout, st);
  return hipGetLastError();
}

// persistent bf16 kernel: 2 workgroups per CU walk the work items (tile x n-block x phase)
static int tfc_num_cus() {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  return ncu;
}
template <int MT, int NT, int WM, int WN, int PAT>
static hipError_t launch_igemm2_pat(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                                    float* stats, float* part_ws, float* dbg, const float* oscale, int flags, hipStream_t st) {
  const int NB32 = tfc_nb32_padded(d.Nout);
  const int per_blk = NT * WN;
  const int nblkN = (tfc_nb32(d.Nout) + per_blk - 1) / per_blk;
  int maxhh = 0;
  for (int pl = 0; pl < d.nplanes; ++pl) maxhh = d.plane[pl].hh > maxhh ? d.plane[pl].hh : maxhh;
  const int buf_bytes = maxhh * TFC_LDS_P * 80;
  constexpr int BN = 32 * NT * WN;
  constexpr int BNS = BN > 128 ? 32 * WN : BN;                    // channels of one staged epilogue pass
  constexpr int STG = 128 * (BNS * 2 + 16);
  // 256-channel tile: the staged tile overlays a consumed halo buffer (see the kernel's LDS map)
  // halo x 2 | staged tile | per-wave statistics slots (EP_STATS) | bias table (EP_BIAS)
  const int lds = (BN > 128 ? 2 * buf_bytes + (STG > buf_bytes ? STG - buf_bytes : 0) : 2 * buf_bytes + STG) + ((flags & TFC_EP_STATS) ? 8 * BN * 4 : 0) +
                  ((flags & TFC_EP_BIAS) ? nblkN * BN * 4 : 0);
  const int nparts = d.tiles_y * d.tiles_x * (d.ph_n > 1 ? d.ph_n : 1);   // InstanceNorm partial slots per image: part_ws[img][phase][tile][Nout][2]
  if (flags & TFC_EP_STATS) {
    if (!part_ws || (long long)d.nimg * nparts * d.Nout * 2 > (long long)TFC_PART_WS_FLOATS) return hipErrorInvalidValue;
  }
  const int nwork = d.nimg * d.tiles_y * d.tiles_x * (d.ph_n > 1 ? d.ph_n : 1) * nblkN;
  // resident workgroups per CU of THIS instantiation (2 for the 128-channel tile: 77 KB of LDS, ~195 VGPRs; 3 for the narrower tiles); a persistent
  // grid never depends on co-residency for correctness (no inter-workgroup waits), so the occupancy query only sizes the grid
  static thread_local int occ_cache = 0, occ_lds = -1;
  if (!occ_cache || occ_lds != lds) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_igemm2_kernel<MT, NT, WM, WN, PAT, 1>, 256, (size_t)lds) != hipSuccess || occ < 1) occ = 2;
    occ_cache = occ > 3 ? 3 : occ;
    occ_lds = lds;
  }
  const int cap = occ_cache * tfc_num_cus();
  const long long phase_wbytes = (long long)tfc_packed_bytes(d, 2);
  if constexpr (BN <= 128) {
    // two-half form (see the kernel): one 512-thread workgroup per CU, halves (nst + 1) / 2 barrier slots apart. Needs two LDS regions (<= 160 KiB) and
    // enough work for both halves of every CU. MEASURED NEUTRAL (scripts/ab_halves.py, DESIGN 3.1: conv shapes x0.98-1.02, transposed x0.87-0.93, with or
    // without s_setprio in the K loop or a deeper weight ring), so it is OFF by default: TFC_IGEMM_HALVES=2 or test config | 32 select it
    static const int halves_env = [] { const char* e = getenv("TFC_IGEMM_HALVES"); return e ? atoi(e) : 1; }();
    const int nst = ((d.Cin_pad * 2) / 64) * d.nplanes;
    const int half_bytes = (lds + 255) & ~255;
    if (halves_env == 2 && 2 * half_bytes <= 160 * 1024 && nwork >= 2 * tfc_num_cus() && (g_tfc_force_cfg < 0 || (g_tfc_force_cfg & 32))) {   // test hook: forced tiles run the two-workgroup form unless bit 5 is set
      static const int shift_env = [] { const char* e = getenv("TFC_IGEMM_SHIFT"); return e ? atoi(e) : -1; }();
      static const int prio_env = [] { const char* e = getenv("TFC_IGEMM_PRIO"); return e ? atoi(e) : 0; }();
      const int shift = (shift_env >= 0 ? shift_env : (nst + 1) / 2) | (prio_env ? 256 : 0);
      TFC_LAUNCH((tfc_igemm2_kernel<MT, NT, WM, WN, PAT, 2>), dim3(tfc_num_cus()), dim3(512), 2 * half_bytes, st, d, (const bf16_t*)in, (const uint4*)wp,
                 (bf16_t*)out, bias, part_ws, dbg, oscale, flags, NB32, nblkN, buf_bytes, phase_wbytes, nwork, half_bytes, shift);
      if (flags & TFC_EP_STATS) return tfc_launch_part_reduce(part_ws, stats, d.nimg, nparts, 2 * d.Nout, st);
      return hipGetLastError();
    }
  }
  TFC_LAUNCH((tfc_igemm2_kernel<MT, NT, WM, WN, PAT>), dim3(nwork < cap ? nwork : cap), dim3(256), lds, st, d, (const bf16_t*)in, (const uint4*)wp,
             (bf16_t*)out, bias, part_ws, dbg, oscale, flags, NB32, nblkN, buf_bytes, phase_wbytes, nwork, 0,
             [] { static const int p = [] { const char* e = getenv("TFC_IGEMM_PRIO"); return e ? atoi(e) : 0; }(); return p ? 256 : 0; }());
  if (flags & TFC_EP_STATS) return tfc_launch_part_reduce(part_ws, stats, d.nimg, nparts, 2 * d.Nout, st);
  return hipGetLastError();
}
template <int MT, int NT, int WM, int WN>
static hipError_t launch_igemm2_cfg(int pat, const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                                    float* stats, float* part_ws, float* dbg, const float* oscale, int flags, hipStream_t st) {
  switch (pat) {
    case 1: return launch_igemm2_pat<MT, NT, WM, WN, 1>(d, in, wp, out, bias, stats, part_ws, dbg, oscale, flags, st);
    case 2: return launch_igemm2_pat<MT, NT, WM, WN, 2>(d, in, wp, out, bias, stats, part_ws, dbg, oscale, flags, st);
    case 3: return launch_igemm2_pat<MT, NT, WM, WN, 3>(d, in, wp, out, bias, stats, part_ws, dbg, oscale, flags, st);
    default: return launch_igemm2_pat<MT, NT, WM, WN, 6>(d, in, wp, out, bias, stats, part_ws, dbg, oscale, flags, st);
  }
}

template <typename T, int MT, int NT, int WM, int WN>
static hipError_t launch_igemm_cfg(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                                   float* stats, float* part_ws, float* out_nchw, const float* oscale, int flags, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    // bf16, compile-time tap pattern, whole 16-byte output units, NHWC output: the persistent kernel (test hook: config | 16 = one tile per workgroup)
    const int pat = match_pattern(d, 2);
    static const bool legacy_env = [] { const char* e = getenv("TFC_LEGACY_IGEMM"); return e && atoi(e) != 0; }();   // A/B knob for profiling
    if (pat != 0 && d.Nout % 8 == 0 && !(flags & TFC_EP_TANH_NCHW) && !(g_tfc_force_cfg >= 0 && (g_tfc_force_cfg & 16)) && !legacy_env)
      return launch_igemm2_cfg<MT, NT, WM, WN>(pat, d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  }
  switch (match_pattern(d, sizeof(T))) {
    case 1: return launch_igemm_pat<T, MT, NT, WM, WN, 1>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
    case 2: return launch_igemm_pat<T, MT, NT, WM, WN, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
    case 3: return launch_igemm_pat<T, MT, NT, WM, WN, 3>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
    case 6: return launch_igemm_pat<T, MT, NT, WM, WN, 6>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
    default: return launch_igemm_pat<T, MT, NT, WM, WN, 0>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  }
}

// Tile-shape choice: 128 pixels x {128, 64, 32} channels. Wider N tiles reuse each A fragment more, but deep / up-path layers
// have so few pixel tiles that a 128-wide tile leaves most of the 256 CUs idle -> narrow the tile until the grid has >= ~2
// workgroups per CU (or the narrowest tile is reached).
thread_local int g_tfc_force_cfg = -1;                           // test hook (tfc_debug_set_igemm_config): -1 = heuristic; per thread
thread_local long long g_tfc_launch_count = 0;

// ---------------------------------------------------------------------------------------------------
// First block, FORWARD, fused (round 3): conv(8 padded channels -> 64, k4 s1 p1) -> [+bias, x 1/sigma] -> LeakyReLU(0.2) -> BlurPool(stride 2) in one
// kernel. The unfused pair (tfc_conv_c8_kernel, tfc_act_pool2_fwd_kernel) writes the 255 x 255 x 64 conv output (266 MB at batch 32) and reads it
// back -- 532 MB of HBM traffic for a 67 MB result whose backward needs only the SIGN of that tensor (sign words, 16.6 MB). Here it never exists:
//   * tile = 4 x 15 pooled pixels; their 10 x 32 conv pixels (5 row pairs x 2 column halves = 10 MFMA M-subtiles) are computed from a 13 x 35 input halo
//     with the weights-stationary fragments of tfc_conv_c8_kernel (wave (wm, wn): 5 subtiles x 32 channels), rounded to bf16 exactly where the unfused
//     chain stores them, and staged in LDS as two 32-channel planes [conv pixel][32] of 64-byte rows;
//   * the BlurPool is LINEAR: pooled[p][c] = sum_k P[p][k] * y[k][c] with P = products of [1,3,3,1]/8 taps (reflect aliases merged; exact in bf16).
//     A block of 2 pooled rows x 16 columns (32 = one MFMA N extent) depends on 6 conv rows x 32 columns = 192 positions: 12 MFMAs 32x32x16 per
//     (block, 32-channel plane), one such pair per wave; the plane is the A operand through the transposing LDS read, P rows are 16-byte reads;
//   * form D (discriminator block 1: LeakyReLU runs BEFORE the bf16 rounding, in the conv epilogue) blurs the stored bf16 activation; form G (generator
//     down1: the raw conv output is rounded, LeakyReLU runs in fp32 inside the pooling pass) uses leaky(y) = max(y, 0) + 0.2 min(y, 0): both parts are
//     bf16-exact (v_pk_max_i16 / v_pk_min_i16 on the sign-magnitude bits), blurred separately and combined in fp32 -- the same numbers as the fp32 blur of
//     the unfused kernel up to the order of <= 16 exact-product additions;
//   * sign words of the would-be-stored tensor leave through the accumulator comparison (v_cmp lane mask -> v_writelane), as in tfc_conv_c8_kernel<true>.
// Each conv pixel is computed by the one or two tiles that need it (1.33x the conv MFMAs: the layer has 25.6 GFLOP, the matrix cores idle anyway).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_max0(unsigned w) {          // per 16-bit half: max(x, 0) as signed integers (v_pk_max_i16)
  const s16x2_t z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, w), z));
}
template <bool GFORM>
__global__ void __launch_bounds__(256, 2)
tfc_first_block_fwd_kernel(const bf16_t* __restrict__ in, int IH, int IW, const uint4* __restrict__ wp, int NB32, const float* __restrict__ bias,
                           const float* __restrict__ oscale, float slope, bf16_t* __restrict__ out, int o_pitch, int Ho, int Wo,
                           unsigned char* __restrict__ sign_mask, int nimg, int tiles_y, int tiles_x, int per) {
  constexpr int P = 40, PS = 16;                                 // halo row pitch (pixels; == 8 mod 16: conflict-free fragment reads), bytes per halo pixel
  constexpr int HHH = 13, HHW = 35;                              // input halo of the 10 x 32 conv region
  constexpr int HB = HHH * P * PS;                               // 8320
  constexpr int NK = 320, PLANE = NK * 64;                       // conv pixels of a tile; one 32-channel plane [k][32 ch]
  constexpr int PP = 400, PB = 32 * PP;                          // tap matrix of a block: [32 pooled px][192 k] bf16, 400-byte rows (conflict-free 16-byte reads)
  __shared__ __attribute__((aligned(16))) unsigned char smem[HB + 2 * PLANE + 2 * PB + NK * 16];
  unsigned char* planes = smem + HB;
  unsigned char* pmat = planes + 2 * PLANE;
  unsigned* smask = reinterpret_cast<unsigned*>(pmat + 2 * PB);  // [320 conv px][2 channel halves][2 lane halves]: partial sign words, OR-ed by the reader
#ifdef TFC_STAMP
  long long* stamp_out = reinterpret_cast<long long*>(sign_mask);   // diagnostic build: phase totals per wave instead of the sign words
  sign_mask = nullptr;
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};                    // [7]: tap-matrix builds
  unsigned long long tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev) :: "memory");
#define FB_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ph[i] += (long long)(t_ - tprev); tprev = t_; } while (0)
#else
#define FB_STAMP(i) do { } while (0)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, r = lane & 31;
  const int CH = IH - 1, CW = IW - 1;                             // conv output size

  uint4 bw[8];                                                   // this wave's 32 output channels x K = 128, for the whole launch
#pragma unroll
  for (int s = 0; s < 8; ++s) bw[s] = wp[((size_t)s * NB32 + wn) * 64 + lane];
  // swapped MFMA operands (weights as A): a lane holds 16 channels {8 q4 + 4 h + e} of ONE conv pixel -> 8-byte LDS stores, sign bits by packed integer ops
  float bvv[16];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 t = bias ? *reinterpret_cast<const float4*>(bias + wn * 32 + 8 * q4 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    bvv[4 * q4] = t.x; bvv[4 * q4 + 1] = t.y; bvv[4 * q4 + 2] = t.z; bvv[4 * q4 + 3] = t.w;
  }
  const float osc = oscale ? *oscale : 1.f;
  // plane rows are 64 bytes (eight 8-byte chunks); chunk c of row k lives at c ^ sw(k), sw(k) = ((k >> 2) & 3) | (((k >> 5) & 1) << 2): the epilogue's
  // 8-byte stores (lane = pixel: k steps by 1 per two lanes, by 32 between them) and the transposing reads of the pooling GEMM are both conflict-free
  const int swl = ((r >> 3) & 3) | ((r & 1) << 2);

  const int tpi = tiles_y * tiles_x;
  const int ntiles = nimg * tpi;
  const int t0 = blockIdx.x * per, t1 = (t0 + per) < ntiles ? (t0 + per) : ntiles;
  if (t0 >= t1) return;
  // halo staging: 13 x 35 pixels of 16 bytes, two per thread
  int hy_[2], hx_[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int idx = tid + i * 256; hy_[i] = idx / HHW; hx_[i] = idx - hy_[i] * HHW; }
  uint4 hv[2];
  // conv row / column of region pixel (0, 0): 2 oy0 - 1 -- one less when the tile's FIRST pooled row (column) is the image's last one and the conv size is
  // odd: its tap 2 oy + 2 = CH + 1 then reflects to CH - 3 = 2 oy0 - 2, in front of the usual origin (the rows behind are unused in that tile)
  auto origin = [&](int o0, int Lo, int L) { return 2 * o0 - 1 - ((o0 == Lo - 1 && L == 2 * o0 + 1) ? 1 : 0); };
  bool hok[2];
  // always two loads per thread (clamped address + validity flag applied at the LDS store): straight-line code, so that the compiler's wait counts are exact
  // tiles are walked column-major inside an image (the border class -- tap matrices -- changes 3 times per column of tiles); positions advance
  // incrementally (the divisions of a decode cost a wave ~450 cycles per tile)
  struct Pos { int img, txb, tyb; };
  auto decode = [&](int tl) { Pos q; q.img = tl / tpi; const int rem = tl - q.img * tpi; q.txb = rem / tiles_y; q.tyb = rem - q.txb * tiles_y; return q; };
  auto advance = [&](Pos& q) {
    if (++q.tyb == tiles_y) { q.tyb = 0; if (++q.txb == tiles_x) { q.txb = 0; ++q.img; } }
  };
  auto halo_load = [&](const Pos& q) {
    const int img = q.img, txb = q.txb, tyb = q.tyb;
    const int y0 = origin(4 * tyb, Ho, CH) - 1, x0 = origin(15 * txb, Wo, CW) - 1;   // input row / column of halo pixel (0, 0): conv row c reads input rows from c - 1
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = y0 + hy_[i], x = x0 + hx_[i];
      hok[i] = hy_[i] < HHH && y >= 0 && y < IH && x >= 0 && x < IW;
      const int yc = hok[i] ? y : 0, xc = hok[i] ? x : 0;
      hv[i] = *reinterpret_cast<const uint4*>(in + ((size_t)(img * IH + yc) * IW + xc) * 8);
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (hy_[i] < HHH) *reinterpret_cast<uint4*>(smem + (hy_[i] * P + hx_[i]) * PS) = hok[i] ? hv[i] : make_uint4(0, 0, 0, 0);
  };
  auto refl = [](int v, int L) { return v < 0 ? -v : (v >= L ? 2 * L - 2 - v : v); };
  // first region row of the 6-row K window of block b (pooled rows oy0 + 2b, + 1): the smallest conv row its valid pooled rows touch (reflect aliases
  // included), at most 4 so that the window stays inside the 10 region rows
  auto win_start = [&](int b, int oy0, int cr0) {
    int lo = 1 << 30;
    for (int pr = 0; pr < 2; ++pr) {
      const int oy = oy0 + 2 * b + pr;
      if (oy >= Ho) continue;
      for (int i = 0; i < 4; ++i) { const int y = refl(2 * oy - 1 + i, CH); lo = y < lo ? y : lo; }
    }
    const int w = lo == (1 << 30) ? 4 * b : lo - cr0;
    return w < 0 ? 0 : (w > 4 ? 4 : w);
  };
  // the tap matrix of block b (pooled rows 2b, 2b + 1 of the tile), P[p = pr * 16 + pc][k = rr * 32 + cc] = R[pr][rr] * C[pc][cc]: the per-row and per-column
  // tap sums (reflect aliases merged; multiples of 1/8) are tabulated first, the products (multiples of 1/64 <= 1: exact in bf16) written 8 at a time.
  // The tables overlay the sign-word buffer, which is idle here (written by the epilogue after the build's last barrier, read before the next build).
  auto pmat_build = [&](int b, int oy0, int ox0) {
    unsigned char* pm = pmat + b * PB;
    float* Ct = reinterpret_cast<float*>(smask);                  // [16][32]
    float* Rt = Ct + 16 * 32;                                     // [2][8]
    const int rg0 = origin(oy0, Ho, CH);
    const int cr0 = rg0 + win_start(b, oy0, rg0), cc0 = origin(ox0, Wo, CW);   // conv row / column of k = 0 of this block
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256, pc = e >> 5, cc = e & 31;
      const int ox = ox0 + pc;
      float w = 0.f;
      if (pc < 15 && ox < Wo)
        for (int j = 0; j < 4; ++j)
          if (refl(2 * ox - 1 + j, CW) - cc0 == cc) w += (j == 0 || j == 3) ? 0.125f : 0.375f;
      Ct[e] = w;
    }
    if (tid < 16) {
      const int pr = tid >> 3, rr = tid & 7;
      const int oy = oy0 + 2 * b + pr;
      float w = 0.f;
      if (rr < 6 && oy < Ho)
        for (int i = 0; i < 4; ++i)
          if (refl(2 * oy - 1 + i, CH) - cr0 == rr) w += (i == 0 || i == 3) ? 0.125f : 0.375f;
      Rt[tid] = w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int u = tid + i * 256;                                // 768 units of 8 entries
      const int pp = u / 24, k0 = (u - pp * 24) * 8;
      const float rw = Rt[(pp >> 4) * 8 + (k0 >> 5)];
      const float4 c0 = *reinterpret_cast<const float4*>(Ct + (pp & 15) * 32 + (k0 & 31));
      const float4 c1 = *reinterpret_cast<const float4*>(Ct + (pp & 15) * 32 + (k0 & 31) + 4);
      uint4 o;
      o.x = pack_bf16x2(rw * c0.x, rw * c0.y);
      o.y = pack_bf16x2(rw * c0.z, rw * c0.w);
      o.z = pack_bf16x2(rw * c1.x, rw * c1.y);
      o.w = pack_bf16x2(rw * c1.z, rw * c1.w);
      *reinterpret_cast<uint4*>(pm + pp * PP + k0 * 2) = o;
    }
    __syncthreads();
  };
  int key_y[2] = {-(1 << 30), -(1 << 30)}, key_x = -(1 << 30);           // border class the two tap matrices were built for

  const int grp = lane >> 4, li = lane & 15;
  const int cb16 = grp & 1, hk = grp >> 1, q = li >> 2, pq = li & 3;
  const int c8 = cb16 * 4 + pq;
  // [row parity of the 32-row group][low / high four rows]: byte offset of this lane's 8-byte chunk in a 16-row k-step
  const int trLo0 = (8 * hk + q) * 64 + ((c8 ^ (2 * hk)) * 8), trHi0 = (8 * hk + q + 4) * 64 + ((c8 ^ (2 * hk + 1)) * 8);
  const int trLo1 = (8 * hk + q) * 64 + ((c8 ^ (2 * hk) ^ 4) * 8), trHi1 = (8 * hk + q + 4) * 64 + ((c8 ^ (2 * hk + 1) ^ 4) * 8);
  auto tr16 = [&](const unsigned char* p0, int olo, int ohi) {
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + olo));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p0 + ohi));
    uint4 o;
    o.x = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    o.y = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    o.z = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    o.w = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return o;
  };

  Pos cur = decode(t0), pre = cur;                                // pre: the tile whose halo is requested next (at most t1 - 1)
  int pre_tl = t0;
  halo_load(pre);
  halo_store();
  __syncthreads();
  if (pre_tl + 1 < t1) { ++pre_tl; advance(pre); }
  halo_load(pre);                                                 // two tiles ahead from here on: the loads of tile t + 2 are issued in front of tile t's output stores
  for (int tl = t0; tl < t1; ++tl, advance(cur)) {
    const int img = cur.img, txb = cur.txb, tyb = cur.tyb;
    const int oy0 = 4 * tyb, ox0 = 15 * txb;
    const bool more = tl + 1 < t1;
    FB_STAMP(6);
    // the tap matrices of this tile (rebuilt only when the border class of a block changes; here, while the sign-word buffer they borrow is idle)
    {
      const int kx = (ox0 == 0 || 2 * (ox0 + 15) + 1 >= CW) ? ox0 : -1;
      bool rb[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oyb = oy0 + 2 * b;
        const int ky = (oyb == 0 || 2 * (oyb + 1) + 2 >= CH || oyb + 1 >= Ho) ? oyb : -1;
        rb[b] = ky != key_y[b] || kx != key_x;
        key_y[b] = ky;
      }
      key_x = kx;
      if (rb[0]) pmat_build(0, oy0, ox0);
      if (rb[1]) pmat_build(1, oy0, ox0);
    }
    FB_STAMP(7);
    // ---- 1. convolution of the 10 x 32 region: subtile m = (row pair rp = m >> 1, column half m & 1); wave (wm, wn): m = 5 wm .. 5 wm + 4 ----
    f32x16_t acc[5];
#pragma unroll
    for (int mi = 0; mi < 5; ++mi)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[mi][j] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 5; ++mi) {
      const int m = 5 * wm + mi;
      const unsigned char* buf = smem + ((2 * (m >> 1) + (r & 1)) * P + 16 * (m & 1) + (r >> 1)) * PS + h * PS;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 a = *reinterpret_cast<const uint4*>(buf + ((s >> 1) * P + 2 * (s & 1)) * PS);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw[s]), __builtin_bit_cast(bf16x8_t, a), acc[mi], 0, 0, 0);
      }
    }
    FB_STAMP(0);
    __syncthreads();                                              // halo consumed
    FB_STAMP(1);
    if (more) halo_store();                                       // tile tl + 1 (requested a tile ago)
    if (pre_tl + 1 < t1) { ++pre_tl; advance(pre); }
    halo_load(pre);                                               // (the last tiles re-request the last halo: no branch around a load)
    // ---- 2. epilogue: the value the unfused chain would store (bf16), into the planes; its sign into the sign words ----
    tfc_static_for<0, 5>([&](auto mic) {
      constexpr int mi = decltype(mic)::value;
      const int m = 5 * wm + mi;
      const int kpx = (2 * (m >> 1) + (r & 1)) * 32 + 16 * (m & 1) + (r >> 1);   // this lane's conv pixel
      unsigned char* prow = planes + wn * PLANE + kpx * 64;
      unsigned slo = 0, shi = 0;
      tfc_static_for<0, 4>([&](auto qc) {
        constexpr int q4 = decltype(qc)::value;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[mi][4 * q4 + e] * osc + bvv[4 * q4 + e];
          if (!GFORM) v[e] = fmaxf(v[e], slope * v[e]);             // form D: LeakyReLU before the rounding (conv epilogue of the unfused chain)
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(prow + (((2 * q4 + h) ^ swl) * 8)) = o;
        if (sign_mask) {
          // (short)x > 0 per half: clamp to [0, 1] ([0, 2] in the high half) as signed 16-bit integers, then one dot2 puts both bits at 8 q4 + 2 i (+ 1)
          constexpr unsigned one_two = 0x00020001u;
          const s16x2_t lim = __builtin_bit_cast(s16x2_t, one_two), z = {0, 0};
          const u16x2_t c0 = __builtin_bit_cast(u16x2_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2_t, o.x), z), lim));
          const u16x2_t c1 = __builtin_bit_cast(u16x2_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2_t, o.y), z), lim));
          constexpr unsigned short w0 = 1u << (8 * (q4 & 1)), w1 = 1u << (8 * (q4 & 1) + 2);
          const u16x2_t g0 = {w0, w0}, g1 = {w1, w1};
          if (q4 < 2) {
            slo = __builtin_amdgcn_udot2(c0, g0, slo, false);
            slo = __builtin_amdgcn_udot2(c1, g1, slo, false);
          } else {
            shi = __builtin_amdgcn_udot2(c0, g0, shi, false);
            shi = __builtin_amdgcn_udot2(c1, g1, shi, false);
          }
        }
      });
      if (sign_mask) smask[kpx * 4 + wn * 2 + h] = (slo | (shi << 16)) << (4 * h);
    });
    FB_STAMP(2);
    __syncthreads();                                              // planes and sign words visible
    FB_STAMP(3);
    // ---- 3. sign words of the conv pixels this tile owns: region rows 1..8, columns 1..30 ----
    if (sign_mask && tid < 240) {
      const int y = 2 * oy0 + tid / 30, x = 2 * ox0 + tid % 30;    // the tile owns conv rows 2 oy0 .. + 7, columns 2 ox0 .. + 29
      const int rr = y - origin(oy0, Ho, CH), cc = x - origin(ox0, Wo, CW);
      if (y < CH && x < CW && rr < 10 && cc < 32)
      {
        const uint4 pw = *reinterpret_cast<const uint4*>(smask + (rr * 32 + cc) * 4);
        *reinterpret_cast<uint2*>(sign_mask + ((size_t)(img * CH + y) * CW + x) * 8) = make_uint2(pw.x | pw.y, pw.z | pw.w);
      }
    }
    // ---- 4. BlurPool as a GEMM: wave = (block b, 32-channel plane cb) ----
    {
      const int b = wave >> 1, cb = wave & 1;
      f32x16_t gp, gn;
#pragma unroll
      for (int j = 0; j < 16; ++j) { gp[j] = 0.f; gn[j] = 0.f; }
      const int wst = win_start(b, oy0, origin(oy0, Ho, CH));
      const unsigned char* pa = planes + cb * PLANE + (32 * wst) * 64;
      const bool wodd = wst & 1;                                      // parity of the first 32-row group of the window (wave-uniform)
      const int oL0 = wodd ? trLo1 : trLo0, oH0 = wodd ? trHi1 : trHi0, oL1 = wodd ? trLo0 : trLo1, oH1 = wodd ? trHi0 : trHi1;
      const unsigned char* pb = pmat + b * PB + r * PP + 8 * h * 2;
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        const uint4 a = ((ks >> 1) & 1) ? tr16(pa + ks * 16 * 64, oL1, oH1) : tr16(pa + ks * 16 * 64, oL0, oH0);
        const uint4 bm = *reinterpret_cast<const uint4*>(pb + ks * 32);
        if constexpr (GFORM) {
          uint4 ap, an;                                            // max(y, 0) and min(y, 0) on the sign-magnitude bits of the bf16 pairs
          ap.x = pk_max0(a.x);
          ap.y = pk_max0(a.y);
          ap.z = pk_max0(a.z);
          ap.w = pk_max0(a.w);
          an.x = a.x ^ ap.x; an.y = a.y ^ ap.y; an.z = a.z ^ ap.z; an.w = a.w ^ ap.w;      // y = y+ or y-: the other part is all-zero bits
          gp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ap), __builtin_bit_cast(bf16x8_t, bm), gp, 0, 0, 0);
          gn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, an), __builtin_bit_cast(bf16x8_t, bm), gn, 0, 0, 0);
        } else {
          gp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bm), gp, 0, 0, 0);
        }
      }
      FB_STAMP(4);
      // lane (pooled pixel r of the block, half h): 16 channels {8 q4 + 4 h + e} of plane cb
      const int pr = r >> 4, pc = r & 15;
      const int oy = oy0 + 2 * b + pr, ox = ox0 + pc;
      // (the lane swaps below are whole-wave operations: outside the bounds test)
      unsigned pk[4][2];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = GFORM ? gp[4 * q4 + e] + slope * gn[4 * q4 + e] : gp[4 * q4 + e];
        pk[q4][0] = pack_bf16x2(v[0], v[1]);
        pk[q4][1] = pack_bf16x2(v[2], v[3]);
      }
#pragma unroll
      for (int pr2 = 0; pr2 < 2; ++pr2) {
        // lanes < 32 hold channels 8q+0..3 (q = 2 pr2) and want 8q+4..7 from lane + 32, which wants this lane's group 2 pr2 + 1: 16 bytes per lane
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr2][0], pk[2 * pr2 + 1][0], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr2][1], pk[2 * pr2 + 1][1], false, false);
        if (pc < 15 && oy < Ho && ox < Wo)
          *reinterpret_cast<uint4*>(out + ((size_t)(img * Ho + oy) * Wo + ox) * o_pitch + cb * 32 + pr2 * 16 + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    }
    FB_STAMP(5);
    __syncthreads();                                              // planes / sign words free for the next tile
  }
#ifdef TFC_STAMP
  if (stamp_out && lane == 0)
    for (int i = 0; i < 8; ++i) stamp_out[(blockIdx.x * 4 + wave) * 8 + i] = ph[i];
#endif
#undef FB_STAMP
}

// first-layer shape (8 padded input channels, 4 x 4 raster taps, <= 64 output channels, bias / scale / LeakyReLU epilogue only)?
bool tfc_conv_c8_eligible(const TfcGather& d, int flags) {
  return d.Cin_pad == 8 && d.nplanes == 1 && d.ph_n <= 1 && d.SS == 1 && d.OS == 1 && plane_pattern(d.plane[0]) == 1 && d.Nout <= 64 && d.Nout % 8 == 0 &&
         (flags & ~(TFC_EP_BIAS | TFC_EP_LEAKY)) == 0;
}
hipError_t tfc_launch_conv_c8(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias, const float* oscale, int flags,
                              unsigned char* sign_mask, hipStream_t st) {
  static int grid_cap = 0;
  if (!grid_cap) {
    int occ = 0, dev = 0, ncu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tfc_conv_c8_kernel<true>, 256, 0);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    grid_cap = (occ < 1 ? 1 : occ) * ncu;
  }
  const int nwork = d.nimg * d.tiles_y * d.tiles_x;
  if (sign_mask)
    TFC_LAUNCH(tfc_conv_c8_kernel<true>, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), 0, st, d, (const bf16_t*)in, (const uint4*)wp,
               (bf16_t*)out, (flags & TFC_EP_BIAS) ? bias : nullptr, oscale, (flags & TFC_EP_LEAKY) ? 1 : 0, tfc_nb32_padded(d.Nout), nwork, sign_mask);
  else
    TFC_LAUNCH(tfc_conv_c8_kernel<false>, dim3(nwork < grid_cap ? nwork : grid_cap), dim3(256), 0, st, d, (const bf16_t*)in, (const uint4*)wp,
               (bf16_t*)out, (flags & TFC_EP_BIAS) ? bias : nullptr, oscale, (flags & TFC_EP_LEAKY) ? 1 : 0, tfc_nb32_padded(d.Nout), nwork, sign_mask);
  return hipGetLastError();
}

// fused first block forward: x NHWC8 [N][IH][IW][8] -> pooled [N][Ho][Wo] (o_pitch); gform: LeakyReLU after the bf16 rounding of the raw conv output
hipError_t tfc_launch_first_block_fwd(const void* in, int N, int IH, int IW, const void* wp, const float* bias, const float* oscale, float slope, int gform,
                                      void* out, int o_pitch, unsigned char* sign_mask, hipStream_t st) {
  const int CH = IH - 1, CW = IW - 1;
  const int Ho = (CH - 1) / 2 + 1, Wo = (CW - 1) / 2 + 1;
  const int tiles_y = (Ho + 3) / 4, tiles_x = (Wo + 14) / 15;
  const int ntiles = N * tiles_y * tiles_x;
  int nwg = 2 * tfc_num_cus();
  if (nwg > ntiles) nwg = ntiles;
  const int per = (ntiles + nwg - 1) / nwg;
  nwg = (ntiles + per - 1) / per;
  if (gform)
    TFC_LAUNCH(tfc_first_block_fwd_kernel<true>, dim3(nwg), dim3(256), 0, st, (const bf16_t*)in, IH, IW, (const uint4*)wp, tfc_nb32_padded(64), bias, oscale, slope,
               (bf16_t*)out, o_pitch, Ho, Wo, sign_mask, N, tiles_y, tiles_x, per);
  else
    TFC_LAUNCH(tfc_first_block_fwd_kernel<false>, dim3(nwg), dim3(256), 0, st, (const bf16_t*)in, IH, IW, (const uint4*)wp, tfc_nb32_padded(64), bias, oscale, slope,
               (bf16_t*)out, o_pitch, Ho, Wo, sign_mask, N, tiles_y, tiles_x, per);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_igemm_t(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                                 float* stats, float* part_ws, float* out_nchw, const float* oscale, int flags, hipStream_t st) {
  const int nb = tfc_nb32(d.Nout);
  const int fcfg = g_tfc_force_cfg < 0 ? -1 : (g_tfc_force_cfg & 15);     // bit 4 of the test hook selects the one-tile-per-workgroup kernel
  if constexpr (sizeof(T) == 2) {
    if (g_tfc_force_cfg < 0 && tfc_conv_c8_eligible(d, flags)) return tfc_launch_conv_c8(d, in, wp, out, bias, oscale, flags, nullptr, st);
  }
#ifdef TFC_PROBE_W64
  if constexpr (sizeof(T) == 2) {
    if (fcfg == 4 && nb >= 8) return launch_igemm2_cfg<4, 2, 1, 4>(match_pattern(d, 2), d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  }
#endif
  if (fcfg == 3 && nb >= 4) return launch_igemm_cfg<T, 4, 1, 1, 4>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if (fcfg == 0 && nb >= 4) return launch_igemm_cfg<T, 2, 2, 2, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if ((fcfg == 0 || fcfg == 1) && nb >= 2) return launch_igemm_cfg<T, 2, 1, 2, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if (fcfg >= 0 && fcfg != 15) return launch_igemm_cfg<T, 1, 1, 4, 1>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x;
  const int target = 512;
  if constexpr (sizeof(T) == 2) {
    // Few tiles, many output channels (the deep 16 x 16 ... 4 x 4 layers): the weight stream dominates. One 128-channel tile per workgroup with
    // each wave owning 32 channels for all 128 pixels (no weight fragment is fetched by two waves) beats two 64-channel workgroups as soon as
    // it still gives every CU a workgroup -- measured at batch 32: conv 16x16 512->512 70 -> 66 us, convT 16x16 1024->256 86 -> 67 us,
    // convT 8x8 1024->512 84 -> 66 us, convT 4x4 512->512 44 -> 32 us, convT 32x32 512->128 100 -> 69 us (conv 8x8 512->512 would leave half
    // the CUs idle: 50 -> 57 us, excluded; the four sub-pixel phases of a transposed convolution count as tiles).
    static const bool old_rule = [] { const char* e = getenv("TFC_TILE_RULE_OLD"); return e && atoi(e) != 0; }();   // A/B knob for profiling
    const int w128 = ntiles * (d.ph_n > 1 ? d.ph_n : 1) * ((nb + 3) / 4);
    if (!old_rule && nb >= 4 && d.Cin_pad >= 64 && w128 >= tfc_num_cus() && (w128 < 2048 || d.Cin_pad >= 128))   // (128 -> 256 at 64 x 64: 130 -> 124 us; 64 -> 128 at 128 x 128 stays on <2,2,2,2>: 132 vs 157)   // (also 1-3 % ahead of <2,2,2,2> at 32 x 32 256->512; the two are equal beyond)
      return launch_igemm_cfg<T, 4, 1, 1, 4>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  }
  if (nb >= 4 && ntiles * ((nb + 3) / 4) >= target) return launch_igemm_cfg<T, 2, 2, 2, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if (nb >= 2 && (ntiles * ((nb + 1) / 2) >= target || nb < 4)) return launch_igemm_cfg<T, 2, 1, 2, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if (nb >= 4 && ntiles * nb < target / 2) return launch_igemm_cfg<T, 1, 1, 4, 1>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  if (nb >= 2) return launch_igemm_cfg<T, 2, 1, 2, 2>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
  return launch_igemm_cfg<T, 1, 1, 4, 1>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
}
hipError_t tfc_launch_igemm(int dt, const TfcGather& d, const void* in, const void* wp, void* out, const float* bias,
                            float* stats, float* part_ws, float* out_nchw, const float* oscale, int flags, hipStream_t st) {
  return dt == TFC_DT_BF16 ? launch_igemm_t<bf16_t>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st)
                           : launch_igemm_t<float>(d, in, wp, out, bias, stats, part_ws, out_nchw, oscale, flags, st);
}

// fin (nullable): torch-layout destination of the gradient. When the launch can reduce its slabs straight into it (tfc_wgrad_reduce_fin_kernel),
// fin->done is set and the caller skips the finish pass.
static bool tfc_fin_eligible(const TfcWgradFin* fin, int npairs, int sel) {
  static const bool off = [] { const char* e = getenv("TFC_WGRAD_NO_FIN"); return e && atoi(e) != 0; }();   // A/B knob for profiling
  return fin && fin->grad && !off && fin->sn % 16 == 0 && fin->sc % 16 == 0 && npairs * sel * 4 >= 256;     // enough workgroups to fill the chip
}
template <typename T>
static hipError_t launch_wgrad_t(const TfcGather& d, const void* dO, const void* in, float* dwacc, float4* slab, int Nn_pad, int Nn_real,
                                 int Cw_real, hipStream_t st, TfcWgradFin* fin) {
  constexpr int ES = sizeof(T);
  const int nbw = (Nn_pad + 63) / 64, ncb = (d.Cin_pad + 31) / 32;
  const int ntiles = d.nimg * d.tiles_y * d.tiles_x;
  // split-K over pixel tiles: the kernel runs 2 workgroups per CU (208 VGPRs, 60 KB LDS), so aim at exactly 512 of them --
  // one full round, no half-empty tail, and the fewest atomic flushes
  int nsplit = 512 / (nbw * ncb);
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) nsplit = 1;
  // slab budget (api.hip reserves 512 workgroups x 128 KiB): a layer with more than 512 (n-block, c-block) pairs (e.g. 2048 x 1024) cannot keep
  // one slab per workgroup -- such a layer flushes with fp32 atomics instead (slab = nullptr), it never writes past the region
  if (slab && (long long)nbw * ncb * nsplit > 512) slab = nullptr;
  const int lds = (2 * 128 * 32 * ES + TFC_MAX_HH * TFC_MAX_HW * 32 * ES) * (ES == 2 ? 2 : 1);
  const dim3 grid(nbw * ncb * nsplit);
  if constexpr (ES == 2) {
    bool t22 = d.plane[0].ntaps == 4 && (g_tfc_force_cfg < 0 || (g_tfc_force_cfg & 15) != 2);
    int seen = 0;
    for (int t = 0; t < 4 && t22; ++t) {
      t22 = (unsigned)d.plane[0].tap_dy[t] < 2u && (unsigned)d.plane[0].tap_dx[t] < 2u;
      seen |= 1 << (d.plane[0].tap_dy[t] * 2 + d.plane[0].tap_dx[t]);
    }
    if (t22 && seen == 15) {                                     // 2 x 2-tap plane: quadrant-per-wave kernel (64 n x 64 c per workgroup)
      const int ncb2 = (d.Cin_pad + 63) / 64;
      int ns = 512 / (nbw * ncb2);
      if (ns > ntiles) ns = ntiles;
      if (ns < 1) ns = 1;
      if (slab && (long long)nbw * ncb2 * ns > 512) slab = nullptr;   // same slab budget as below
      const int lds22 = 2 * (2 * 128 * 64 + 2 * d.plane[0].hh * d.plane[0].hw * 64);
      TFC_LAUNCH((tfc_wgrad22_kernel<T>), dim3(nbw * ncb2 * ns), dim3(256), lds22, st, d, (const T*)dO, (const T*)in, dwacc, slab,
                         Nn_pad, Nn_real, Cw_real, nbw, ncb2, ns);
      if (slab)
        TFC_LAUNCH(tfc_wgrad_reduce_kernel, dim3(nbw * ncb2 * 4 * 4 * 4), dim3(256), 0, st, slab, dwacc, d.plane[0], 1, 4, ns, nbw * ncb2,
                           ncb2, Nn_real, Cw_real, -1);
      return hipGetLastError();
    }
  }
  const int tpw = (d.plane[0].ntaps + 3) / 4;                    // taps per wave (tap t belongs to wave t % 4)
  bool raster = (ES == 2) && d.plane[0].ntaps == 16;
  for (int t = 0; t < 16 && raster; ++t) raster = d.plane[0].tap_dy[t] == (t >> 2) && d.plane[0].tap_dx[t] == (t & 3);
  if constexpr (ES == 2) {
    static const bool c8_off = [] { const char* e = getenv("TFC_WGRAD_NO_C8"); return e && atoi(e) != 0; }();   // A/B knob for profiling
    bool ident = raster;
    for (int t = 0; t < 16 && ident; ++t) ident = d.plane[0].tap_mask[t] == (1 << t);
    if (slab && ident && !c8_off && d.Cin_pad == 8 && d.in_pitch == 8 && Nn_pad <= 64 && d.ph_n <= 1 && d.SS == 1 && d.OS == 1 &&
        d.plane[0].hh <= TFC_MAX_HH && d.plane[0].hw <= TFC_MAX_HW && (g_tfc_force_cfg < 0 || (g_tfc_force_cfg & 15) == 15)) {
      const int ns = ntiles < 512 ? ntiles : 512;                 // 2 workgroups per CU, 32 KB of slab each
      TFC_LAUNCH(tfc_wgrad_c8_kernel, dim3(ns), dim3(256), 0, st, d, (const bf16_t*)dO, (const bf16_t*)in, slab, Nn_pad, ns);
      const bool direct = fin && fin->grad;
      TFC_LAUNCH(tfc_wgrad_c8_reduce_kernel, dim3(128), dim3(256), 0, st, slab, dwacc, ns, Nn_real, Cw_real, direct ? fin->grad : nullptr,
                 direct ? fin->sn : 0, direct ? fin->sc : 0, direct ? fin->accumulate : 0);
      if (direct) fin->done = true;
      return hipGetLastError();
    }
  }
#define TFC_WG(TPW_, R_) TFC_LAUNCH((tfc_wgrad_kernel<T, TPW_, R_>), grid, dim3(256), lds, st, d, (const T*)dO, (const T*)in, dwacc, slab, \
                                            Nn_pad, Nn_real, Cw_real, nbw, ncb, nsplit)
  int tw = 4;
  if (raster) TFC_WG(4, true);
  else if (tpw <= 1) { tw = 1; TFC_WG(1, false); } else if (tpw == 2) { tw = 2; TFC_WG(2, false); } else if (tpw == 3) { tw = 3; TFC_WG(3, false); } else TFC_WG(4, false);
#undef TFC_WG
  bool ident = raster;                                           // identity tap masks: tap t IS filter slot t
  for (int t = 0; t < 16 && ident; ++t) ident = d.plane[0].tap_mask[t] == (1 << t);
  if (slab && ident && d.ph_n <= 1 && tfc_fin_eligible(fin, nbw * ncb, 8)) {
    TFC_LAUNCH(tfc_wgrad_reduce_fin_kernel, dim3(nbw * ncb * 8 * 4), dim3(512), 0, st, slab, fin->grad, 0, 8, nsplit, nbw * ncb, ncb, Nn_real,
               Cw_real, fin->sn, fin->sc, fin->accumulate);
    fin->done = true;
  } else if (slab)
    TFC_LAUNCH(tfc_wgrad_reduce_kernel, dim3(nbw * ncb * 4 * (tw * 2) * 4), dim3(256), 0, st, slab, dwacc, d.plane[0], 0, tw * 2, nsplit,
                       nbw * ncb, ncb, Nn_real, Cw_real, -1);
  return hipGetLastError();
}
// fused first-block backward (tfc_wgrad_c8_fused_kernel): d = the TFC_OP_CONV pass-2 descriptor of the layer (8 padded input channels, 64 outputs)
hipError_t tfc_launch_first_block_bwd(const TfcGather& d, const void* yact, int y_pitch, const void* dyp, int dyp_pitch, int Ho, int Wo, const void* in,
                                      void* slab, float* dwacc, float* rstats, float* part_ws, float slope, int Nn_real, int Cw_real,
                                      const unsigned char* sign_mask, TfcWgradFin* fin, hipStream_t st) {
  // wpi workgroups per image, each with `per` consecutive tiles of that image: about 512 workgroups in all (2 per CU), at most 2048 (32 KB of slab each)
  const int tpi = d.tiles_y * d.tiles_x;
  int wpi = 512 / d.nimg;
  if (wpi < 1) wpi = 1;
  if (wpi > tpi) wpi = tpi;
  const int per = (tpi + wpi - 1) / wpi;
  wpi = (tpi + per - 1) / per;
  const int ns = d.nimg * wpi;
  if (ns > 2048 || (rstats && (!part_ws || (long long)ns * 64 > (long long)TFC_PART_WS_FLOATS))) return hipErrorInvalidValue;
  const char* valu_env = getenv("TFC_FIRST_BWD_VALU");            // A/B knob (read per call: the tests compare the two forms in one process)
  if (valu_env && atoi(valu_env) != 0 && yact)
    TFC_LAUNCH(tfc_wgrad_c8_fused_kernel, dim3(ns), dim3(256), 0, st, d, (const bf16_t*)yact, y_pitch, (const bf16_t*)dyp, dyp_pitch, Ho, Wo,
               (const bf16_t*)in, (float4*)slab, rstats ? part_ws : nullptr, slope, wpi, per);
  else if (sign_mask)                                             // transposed blur on the matrix core, signs from the forward pass's sign words
    TFC_LAUNCH(tfc_wgrad_c8_fusedm_kernel<true>, dim3(ns), dim3(256), 0, st, d, (const bf16_t*)yact, y_pitch, (const bf16_t*)dyp, dyp_pitch, Ho, Wo,
               (const bf16_t*)in, (float4*)slab, rstats ? part_ws : nullptr, slope, wpi, per, sign_mask);
  else
    TFC_LAUNCH(tfc_wgrad_c8_fusedm_kernel<false>, dim3(ns), dim3(256), 0, st, d, (const bf16_t*)yact, y_pitch, (const bf16_t*)dyp, dyp_pitch, Ho, Wo,
               (const bf16_t*)in, (float4*)slab, rstats ? part_ws : nullptr, slope, wpi, per, sign_mask);
  TFC_LAUNCH(tfc_wgrad_c8_reduce_kernel, dim3(128), dim3(256), 0, st, (const float4*)slab, dwacc, ns, Nn_real, Cw_real, fin->grad, fin->sn, fin->sc,
             fin->accumulate);
  fin->done = true;
  if (rstats) return tfc_launch_part_reduce(part_ws, rstats, d.nimg, wpi, 64, st);   // rstats[img][64] += the image's workgroup slots, in order
  return hipGetLastError();
}
// transposed convolution / upsample conv, bf16: all four phases in one launch; false = not applicable (caller falls back to per-phase launches)
bool tfc_launch_wgrad_phases_fused(int up, const void* x, int N, int IH, int IW, int x_pitch, int Cin_pad, const void* dy, int dy_pitch, int Cout,
                                   int Cin, float* dwacc, void* slab, hipStream_t st, hipError_t* err, TfcWgradFin* fin) {
  if (g_tfc_force_cfg >= 0 && (g_tfc_force_cfg & 15) == 2) return false;   // tests: keep the per-phase kernels reachable
  const int Nn_pad = (Cout + 7) / 8 * 8;
  const int nbw = (Nn_pad + 31) / 32, ncb = (Cin_pad + 31) / 32;
  const int ntiles = N * ((IH + TFC_TILE_H - 1) / TFC_TILE_H) * ((IW + TFC_TILE_W - 1) / TFC_TILE_W);
  const int T = up ? 9 : 4;                                       // accumulator tiles per wave
  const size_t blk_bytes = (size_t)4 * T * 4 * 64 * 16;           // slab bytes per workgroup
  int nsplit = (up ? 512 : 768) / (nbw * ncb);                    // 2 / 3 workgroups per CU (44 KB LDS; <= 256 / 168 VGPRs)
  const size_t budget = (size_t)512 * 4 * 8 * 4 * 64 * 16;        // slab bytes api.hip reserves
  if ((size_t)nbw * ncb * blk_bytes > budget) return false;
  if ((size_t)nbw * ncb * nsplit * blk_bytes > budget) nsplit = (int)(budget / ((size_t)nbw * ncb * blk_bytes));
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) nsplit = 1;
  const int lds = 4 * 128 * 64 + (TFC_TILE_H + 2) * (TFC_TILE_W + 2) * 64;
  const dim3 grid(nbw * ncb * nsplit);
  TfcPlane none{};
  static const bool head_off = [] { const char* e = getenv("TFC_WGRAD_NO_HEAD"); return e && atoi(e) != 0; }();   // A/B knob for profiling
  if (up && !head_off && Cin_pad == 128 && Nn_pad == 8 && x_pitch >= 128) {   // the generator head: phases x padded channels = one 32-row tile
    const size_t wg_bytes = (size_t)9216 * 16;
    int ns = (int)(budget / wg_bytes);
    if (ns > 512) ns = 512;
    if (ns > ntiles) ns = ntiles;
    if (!fin || !fin->grad || fin->sn % 4 != 0 || fin->sc % 4 != 0) return false;   // the head's own finish pass writes the torch-layout gradient
    TFC_LAUNCH(tfc_wgrad_head_kernel, dim3(ns), dim3(256), 0, st, (const bf16_t*)x, IH, IW, x_pitch, (const bf16_t*)dy, dy_pitch, N, (float4*)slab, ns);
    TFC_LAUNCH(tfc_wgrad_head_reduce_kernel, dim3(576), dim3(256), 0, st, (float4*)slab, ns);
    TFC_LAUNCH(tfc_wgrad_head_finish_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, st, (const float4*)slab, fin->grad, Cout, Cin, fin->sn, fin->sc,
               fin->accumulate);
    fin->done = true;
    *err = hipGetLastError();
    return true;
  }
  if (up) {
    TFC_LAUNCH(tfc_wgradT_kernel<true>, grid, dim3(256), lds, st, (const bf16_t*)x, IH, IW, x_pitch, Cin_pad, (const bf16_t*)dy, dy_pitch,
                       Nn_pad, N, (float4*)slab, nbw, ncb, nsplit);
    for (int wv = 0; wv < 4; ++wv)                                // the phases overlap on the filter taps: one reduce pass per phase, in order
      TFC_LAUNCH(tfc_wgrad_reduce_kernel, dim3(nbw * ncb * T * 4), dim3(256), 0, st, (const float4*)slab, dwacc, none, 3, T, nsplit,
                         nbw * ncb, ncb, Cout, Cin, wv);
  } else {
    static const bool narrow = [] { const char* e = getenv("TFC_WGRADT_NARROW"); return e && atoi(e) != 0; }();   // A/B knob: the 32 x 32 workgroup tile
    if (!narrow && ncb % 2 == 0 && ncb >= 2) {                    // 32 n x 64 c tiles, 2 workgroups per CU
      int ns2 = 512 / (nbw * (ncb / 2));                           // (256 / 384 workgroups measured 8-9 % slower)
      if ((size_t)nbw * ncb * ns2 * blk_bytes > budget) ns2 = (int)(budget / ((size_t)nbw * ncb * blk_bytes));
      if (ns2 > ntiles) ns2 = ntiles;
      if (ns2 < 1) ns2 = 1;
      nsplit = ns2;
      TFC_LAUNCH(tfc_wgradT2_kernel, dim3(nbw * (ncb / 2) * nsplit), dim3(256), 4 * 128 * 64 + 2 * (TFC_TILE_H + 2) * (TFC_TILE_W + 2) * 64, st, (const bf16_t*)x, IH, IW,
                 x_pitch, Cin_pad, (const bf16_t*)dy, dy_pitch, Nn_pad, N, (float4*)slab, nbw, ncb, nsplit);
    } else
    TFC_LAUNCH(tfc_wgradT_kernel<false>, grid, dim3(256), lds, st, (const bf16_t*)x, IH, IW, x_pitch, Cin_pad, (const bf16_t*)dy, dy_pitch,
                       Nn_pad, N, (float4*)slab, nbw, ncb, nsplit);
    if (tfc_fin_eligible(fin, nbw * ncb, 4)) {
      TFC_LAUNCH(tfc_wgrad_reduce_fin_kernel, dim3(nbw * ncb * 4 * 4), dim3(512), 0, st, (const float4*)slab, fin->grad, 2, T, nsplit, nbw * ncb, ncb,
                 Cout, Cin, fin->sn, fin->sc, fin->accumulate);
      fin->done = true;
    } else
      TFC_LAUNCH(tfc_wgrad_reduce_kernel, dim3(nbw * ncb * 4 * T * 4), dim3(256), 0, st, (const float4*)slab, dwacc, none, 2, T, nsplit,
                         nbw * ncb, ncb, Cout, Cin, -1);
  }
  *err = hipGetLastError();
  return true;
}
// slab: >= TFC_WGRAD_SLAB_BYTES of scratch for the split-K partials (bf16 path); nullptr = flush with fp32 atomics (fp32 parity mode)
hipError_t tfc_launch_wgrad(int dt, const TfcGather& d, const void* dO, const void* in, float* dwacc, void* slab, int Nn_pad,
                            int Nn_real, int Cw_real, hipStream_t st, TfcWgradFin* fin) {
  return dt == TFC_DT_BF16 ? launch_wgrad_t<bf16_t>(d, dO, in, dwacc, (float4*)slab, Nn_pad, Nn_real, Cw_real, st, fin)
                           : launch_wgrad_t<float>(d, dO, in, dwacc, (float4*)slab, Nn_pad, Nn_real, Cw_real, st, nullptr);   // fp32 too: slabs, not atomics (deterministic)
}
hipError_t tfc_launch_wgrad_finish(float* acc, float* grad, int Nn, int Cw, long long sn, long long sc,
                                   int accumulate, hipStream_t st) {
  const int total = Nn * Cw;                                      // one thread per (n, c): all 16 taps
  TFC_LAUNCH(tfc_wgrad_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, st, acc, grad, Nn, Cw, sn, sc,
                     accumulate, total);
  return hipGetLastError();
}
