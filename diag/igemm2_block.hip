
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
struct Tile2 {
  int img, a0, b0, phy, phx, nb_blk;
  const unsigned char* wbase;                                    // this tile's weight stream (phase, n-block), wave part excluded
};

template <int MT, int NT, int WM, int WN, int PAT>
__global__ void __launch_bounds__(256, 2)
tfc_igemm2_kernel(const TfcGather d, const bf16_t* __restrict__ in, const uint4* __restrict__ wp, bf16_t* out,
                  const float* __restrict__ bias, float* stats, float* dbg, const float* __restrict__ oscale,
                  int flags, int NB32, int nblkN, int buf_bytes, long long phase_wbytes, int nwork) {
  static_assert(WM * WN == 4 && WM * MT == 4, "4 waves, 128-pixel tile");
  static_assert(PAT != 0, "compile-time tap patterns only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef bf16_t T;
  constexpr int P = TFC_LDS_P;
  constexpr int BD = (PAT == 6) ? 2 : TFC_BD;
  constexpr int NSR = TapPat<PAT>::COLS * 2;
  constexpr int ROWS = TapPat<PAT>::ROWS;
  static_assert(NSR % BD == 0, "register ring must realign every filter row");
  constexpr int BN = 32 * NT * WN;
  constexpr int ROWP = BN * 2 + 16;                              // staged tile: bytes per pixel row (pitch = 4 banks mod 32)
  constexpr int UPR = BN / 8;                                    // 16-byte units per pixel row
  float* out_nchw = dbg;                                         // TFC_STAMP_AT writes here in the diagnostic build
  (void)out_nchw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, r = lane & 31;
  const int G = gridDim.x;
  TFC_STAMP_AT(0);

  const int nchunks = (d.Cin_pad * 2) / 64;
  const int nst = nchunks * d.nplanes;
  unsigned char* stage = smem + 2 * buf_bytes;
  float* sbias = reinterpret_cast<float*>(stage + 128 * ROWP);

  const int laneBase = ((2 * wm * MT + (r & 1)) * P + (r >> 1)) * 80 + h * 16;
  const unsigned lanepart = (unsigned)((wn * NT) * 64 + lane) * 16u;
  const size_t wstep_b = (size_t)NB32 * 1024;
  const float osc = oscale ? *oscale : 1.f;

  auto decode = [&](int w, Tile2& t) {
    const int k = w / G;
    const int cnt = (nwork - k * G) < G ? (nwork - k * G) : G;   // the last round may be partial
    int bid = k * G + tfc_xcd_remap(w - k * G, cnt);
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

  uint4 hv[4];
  int hoff[4];
  auto halo_load = [&](const Tile2& t, int st) {
    const int cc = st / d.nplanes, pl = st - cc * d.nplanes;
    const TfcPlane& pd = d.plane[pl];
    const int nunits = pd.hh * pd.hw * 4;
    const bf16_t* in_img = in + (size_t)t.img * d.IH * d.IW * d.in_pitch;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      hv[i] = make_uint4(0, 0, 0, 0);
      hoff[i] = -1;
      if (idx < nunits) {
        const int pix = idx >> 2, g = idx & 3;
        const int hy = pix / pd.hw, hx = pix - hy * pd.hw;
        hoff[i] = (hy * P + hx) * 80 + g * 16;
        const int y = (t.a0 + pd.dy0 + t.phy * d.ph_d0 + hy) * d.SS + pd.py;
        const int x = (t.b0 + pd.dx0 + t.phx * d.ph_d0 + hx) * d.SS + pd.px;
        if (y >= 0 && y < d.IH && x >= 0 && x < d.IW)
          hv[i] = *reinterpret_cast<const uint4*>(in_img + ((size_t)(y * d.IW + x)) * d.in_pitch + cc * 32 + g * 8);
      }
    }
  };
  auto halo_store = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (hoff[i] >= 0) *reinterpret_cast<uint4*>(buf + hoff[i]) = hv[i];
  };
  auto loadB = [&](const unsigned char* pw, uint4 (&b)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const uint4*>(pw + lanepart + nt * 1024);
  };

  int w = blockIdx.x;
  Tile2 cur, nxt;
  decode(w, cur);
  nxt = cur;
  uint4 br[BD][NT];
#pragma unroll
  for (int i = 0; i < BD; ++i) loadB(cur.wbase + (size_t)i * wstep_b, br[i]);
  float bnext = 0.f;                                             // this thread's bias element of the tile about to start
  if ((flags & TFC_EP_BIAS) && tid < BN) { const int n = cur.nb_blk * BN + tid; bnext = n < d.Nout ? bias[n] : 0.f; }
  halo_load(cur, 0);
  halo_store(smem);
  __syncthreads();
  TFC_STAMP_AT(1);
  int sc = 0;                                                    // running stage counter: halo buffer parity across tiles

  for (;;) {
    const bool has_next = (w + G) < nwork;
    if (has_next) decode(w + G, nxt);
    if ((flags & TFC_EP_BIAS) && tid < BN) sbias[tid] = bnext;   // the previous epilogue's readers are behind its barrier
    if ((flags & TFC_EP_BIAS) && has_next && tid < BN) { const int n = nxt.nb_blk * BN + tid; bnext = n < d.Nout ? bias[n] : 0.f; }

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mi][nt][j] = 0.f;

    const unsigned char* prow = cur.wbase;
    for (int st = 0; st < nst; ++st) {
      const bool last = (st + 1) == nst;
      const bool more = !last || has_next;
      if (!last) halo_load(cur, st + 1);
      else if (has_next) halo_load(nxt, 0);
      const unsigned char* buf = smem + (sc & 1) * buf_bytes + laneBase;
#pragma unroll 1
      for (int row = 0; row < ROWS; ++row) {
        // prefetches that run past this filter row land in the next row of the stream -- or, in the tile's last row, at the START of the
        // next tile's stream (without a next tile: the stream's slack substeps)
        const unsigned char* pnext = (last && row == ROWS - 1 && has_next) ? nxt.wbase : prow + (size_t)NSR * wstep_b;
        const unsigned char* rbuf = buf + (PAT == 2 ? (1 - row) : row) * (P * 80);   // == TapPat::dy(row)
        uint4 a[2][MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) a[0][mi] = *reinterpret_cast<const uint4*>(rbuf + TapPat<PAT>::dx(0) * 80 + mi * (2 * P * 80));
#pragma unroll
        for (int s = 0; s < NSR; ++s) {
          if (s + 1 < NSR) {
            const int off = TapPat<PAT>::dx((s + 1) >> 1) * 80 + ((s + 1) & 1) * 32;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) a[(s + 1) & 1][mi] = *reinterpret_cast<const uint4*>(rbuf + off + mi * (2 * P * 80));
          }
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)                       // swapped operands: rows = channels, columns (lanes) = pixels
              acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, br[s % BD][nt]),
                                                                    __builtin_bit_cast(bf16x8_t, a[s & 1][mi]), acc[mi][nt], 0, 0, 0);
          if (s + BD < NSR) loadB(prow + (size_t)(s + BD) * wstep_b, br[s % BD]);
          else loadB(pnext + (size_t)(s + BD - NSR) * wstep_b, br[s % BD]);
          asm volatile("" ::: "memory");
        }
        prow += (size_t)NSR * wstep_b;
      }
      if (more) halo_store(smem + ((sc + 1) & 1) * buf_bytes);
      __syncthreads();
      ++sc;
    }

    // ---- epilogue: accumulators (channel rows x pixel lanes) -> 16-byte units -> staged tile in LDS ----
    TFC_STAMP_AT(2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        bq[q] = (flags & TFC_EP_BIAS) ? *reinterpret_cast<const float4*>(sbias + (wn * NT + nt) * 32 + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        uint32_t pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v0 = acc[mi][nt][4 * q + 0] * osc + bq[q].x, v1 = acc[mi][nt][4 * q + 1] * osc + bq[q].y;
          float v2 = acc[mi][nt][4 * q + 2] * osc + bq[q].z, v3 = acc[mi][nt][4 * q + 3] * osc + bq[q].w;
          if (flags & TFC_EP_LEAKY) { v0 = fmaxf(v0, 0.2f * v0); v1 = fmaxf(v1, 0.2f * v1); v2 = fmaxf(v2, 0.2f * v2); v3 = fmaxf(v3, 0.2f * v3); }
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
          *reinterpret_cast<uint4*>(stage + rho * ROWP + ((wn * NT + nt) * 32 + pr * 16 + h * 8) * 2) = o;
        }
      }
    }
    TFC_STAMP_AT(3);
    __syncthreads();
    TFC_STAMP_AT(4);
    {
      const int nbase = cur.nb_blk * BN;
      const int u = tid % UPR;
      const int n0 = nbase + u * 8;
      float s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll 2
      for (int k = 0; k < (128 * UPR) / 256; ++k) {
        const int rho = tid / UPR + k * (256 / UPR);
        const int rr = rho & 31;
        const int ty = 2 * (rho >> 5) + (rr & 1), tx = rr >> 1;
        const int a = cur.a0 + ty, b = cur.b0 + tx;
        if (a < d.GH && b < d.GW && n0 < d.Nout) {
          const int oy = a * d.OS + d.OOY + cur.phy * d.ph_oo, ox = b * d.OS + d.OOX + cur.phx * d.ph_oo;
          T* po = out + ((size_t)(cur.img * d.OH + oy) * d.OW + ox) * d.out_pitch + n0;
          uint4 v = *reinterpret_cast<const uint4*>(stage + rho * ROWP + u * 16);
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
      if (flags & TFC_EP_STATS) {                                // lanes with equal (lane % UPR) hold partial sums of the same 8 channels
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 32; o >= UPR; o >>= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        }
        if (lane < UPR && n0 < d.Nout) {
          float* ps = stats + ((size_t)cur.img * d.Nout + n0) * 2;
#pragma unroll
          for (int e = 0; e < 8; ++e) { atomicAdd(ps + 2 * e, s1[e]); atomicAdd(ps + 2 * e + 1, s2[e]); }
        }
      }
    }
    TFC_STAMP_AT(5);
    if (!has_next) break;
    cur = nxt;
    w += G;
  }
#ifdef TFC_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFC_STAMP_AT(6);
#endif
}
