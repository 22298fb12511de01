// extern "C" surface of libtfcgan_hip.so (declared in include/tfc_gan.h): argument checking, gather-descriptor
// construction for every convolution op / pass of the PATCH-16 path, launch, per-kernel timing hooks, and the
// host-side emulator of the gather model used by the CPU tests.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/tfc_gan.h"
#undef TFC_DT_BF16
#undef TFC_DT_F32
#undef TFC_EP_BIAS
#undef TFC_EP_STATS
#undef TFC_EP_ACCUM
#undef TFC_EP_TANH_NCHW
#undef TFC_EP_LEAKY
#undef TFC_EP_RELU
#undef TFC_OP_CONV
#undef TFC_OP_PADCONV
#undef TFC_OP_CONVT
#undef TFC_OP_UPCONV
#undef TFC_OP_CONV3
#include "tfc_desc.h"
#include "pack_math.h"
#define TFC_OP_CONV 0
#define TFC_OP_PADCONV 1
#define TFC_OP_CONVT 2
#define TFC_OP_UPCONV 3
#define TFC_OP_CONV3 4

// ---- internal launchers (igemm.hip / elementwise.hip / losses.hip) ------------------------------------------
int tfc_total_substeps(const TfcGather& d, int es);
int tfc_nb32(int nout);
int tfc_nb32_padded(int nout);
size_t tfc_packed_bytes(const TfcGather& d, int es);
hipError_t tfc_launch_pack(int dt, const TfcGather& d, const float* w, const float* scale, void* wp, int Nreal, int Creal, long long sn, long long sc, hipStream_t st);
hipError_t tfc_launch_igemm(int dt, const TfcGather& d, const void* in, const void* wp, void* out, const float* bias, float* stats, float* part_ws, float* out_nchw, const float* oscale, int flags, hipStream_t st);
hipError_t tfc_launch_first_block_fwd(const void* in, int N, int IH, int IW, const void* wp, const float* bias, const float* oscale, float slope, int gform,
                                      void* out, int o_pitch, unsigned char* sign_mask, hipStream_t st);
bool tfc_conv_c8_eligible(const TfcGather& d, int flags);
hipError_t tfc_launch_conv_c8(const TfcGather& d, const void* in, const void* wp, void* out, const float* bias, const float* oscale, int flags,
                              unsigned char* sign_mask, hipStream_t st);
hipError_t tfc_launch_wgrad(int dt, const TfcGather& d, const void* dO, const void* in, float* dwacc, void* slab, int Nn_pad, int Nn_real, int Cw_real, hipStream_t st, TfcWgradFin* fin);
hipError_t tfc_launch_dgrad_rows4(const void* dy, int dy_pitch, int N, int H, int W, const float* w, int Cin, const float* oscale, int NC, float* dx, hipStream_t st);
hipError_t tfc_launch_upconv_head(const void* x, int x_pitch, int N, int H, int W, const float* w, const float* bias, int Cout, float* out, hipStream_t st);
bool tfc_launch_wgrad_phases_fused(int up, const void* x, int N, int IH, int IW, int x_pitch, int Cin_pad, const void* dy, int dy_pitch, int Cout,
                                   int Cin, float* dwacc, void* slab, hipStream_t st, hipError_t* err, TfcWgradFin* fin);
hipError_t tfc_launch_wgrad_finish(float* acc, float* grad, int Nn, int Cw, long long sn, long long sc, int accumulate, hipStream_t st);
hipError_t tfc_launch_pack_planned(int dt, const void* plan_dev, int njobs, int nblocks, hipStream_t st);
hipError_t tfc_launch_act_fwd(int dt, const ActParams& p, const void* x, const float* stats, void* out, float* stats_out, float* part_ws, hipStream_t st);
hipError_t tfc_launch_act_bwd(int dt, int mode, const ActParams& p, const void* dout, const void* x, const float* stats, float* rstats, void* dx, int use_x, int dx_pitch, float* part_ws, hipStream_t st);
hipError_t tfc_launch_act_pool2_bwd_signs(const ActParams& p, const void* dout, const unsigned char* sign_mask, float* rstats, void* dx, int dx_pitch, float* part_ws, hipStream_t st);
hipError_t tfc_launch_colsum(int dt, const void* x, long long rows, int pitch, int C, float* out, float* part_ws, hipStream_t st);
hipError_t tfc_launch_pack_nhwc8(int dt, const float* a, int Ca, const float* b, int Cb, void* out, int N, int HW, hipStream_t st);
hipError_t tfc_launch_unpack_nchw(int dt, const void* in, int pitch, int c0, int C, float* out, int N, int HW, float alpha, float beta, hipStream_t st);
hipError_t tfc_launch_tanh_bwd_pack(int dt, const float* g, const float* y, void* out, float* dbias, float* part_ws, int N, int C, int HW, hipStream_t st);
hipError_t tfc_launch_sn_bwd(const float* G, const float* W, const float* u, const float* v, const float* sigma2, float* dot_ws, float* gout, int R, int K, int accumulate, hipStream_t st);
hipError_t tfc_launch_bce_rel(int dt, const void* a, const void* b, int n, int stride, float t1, float t2, int mode, float* loss, void* da, void* db, float gscale, hipStream_t st);
hipError_t tfc_launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale, hipStream_t st);
hipError_t tfc_launch_axpby(float* out, const float* x, const float* y, long long n, float a, float b, hipStream_t st);
hipError_t tfc_launch_dropout_mask(unsigned char* out, long long n, unsigned seed, unsigned thresh24, hipStream_t st);
hipError_t tfc_launch_cast(int dt, int to_f32, const void* x, void* y, long long n, hipStream_t st);
hipError_t tfc_launch_triplet16(const float* fake, const float* real, const int* neg_idx, int N, int C, float margin, float eps, float* loss, float* dfake, float gscale, hipStream_t st);
hipError_t tfc_launch_spectrum(const float* img, long long bs, long long cs, int rs, int C, int S, int wins_x, int wins_per_img, int nwin, float* amp, float* pha, int shift, void* ws, hipStream_t st);
size_t tfc_fft_ws_bytes(int S, int nwin);
hipError_t tfc_launch_l1_sum(const float* a, const float* b, long long n, float scale, float* out, hipStream_t st);
hipError_t tfc_launch_probe(float* out, hipStream_t st);
hipError_t tfc_launch_logmag_mse(const float* a, const float* b, int S, int nwin, float* out, int absolute, hipStream_t st);
hipError_t tfc_launch_vectorize_temps(const float* x, long long bs, int rs, int N, int H, int W, const float* lut, float* out, hipStream_t st);
hipError_t tfc_launch_row_triplet(const float* a, const float* p, const float* ng, long long rows, int W, float margin, float eps,
                                  float* loss, hipStream_t st);
hipError_t tfc_launch_head_fwd(int dt, const void* x, int x_pitch, const float* w, void* y, int y_pitch, int N, int H, int W, int C, hipStream_t st);
hipError_t tfc_launch_affine_warp_fwd(const float* src, const float* theta, float* out, int N, int C, int H, int W, hipStream_t st);
hipError_t tfc_launch_affine_warp_bwd(const float* src, const float* theta, const float* gout, float* dtheta, float* dsrc, float* part_ws, int N, int C, int H, int W, hipStream_t st);
hipError_t tfc_launch_morph_grad_fwd(const float* x, float* out, unsigned char* arg, long long planes, int H, int W, hipStream_t st);
hipError_t tfc_launch_morph_grad_bwd(const float* gout, const unsigned char* arg, float* dx, long long planes, int H, int W, hipStream_t st);
hipError_t tfc_launch_row_triplet_grad(const float* a, const float* p, const float* ng, long long rows, int W, float margin, float eps, float gscale,
                                       float* loss, float* da, hipStream_t st);
struct SnBatch {
  const float* W[4];
  float* u[4]; float* v[4]; float* sigma2[4];
  float* us[4]; float* vs[4];
  float* s[4]; float* t[4];
  int R[4], K[4];
  int n;
};
hipError_t tfc_launch_sn_step_batched(const SnBatch& b, int power_iter, float eps, hipStream_t st);

// ---- error handling --------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
static int hipfail(hipError_t e, const char* where) { return fail(-100 - (int)e, "%s: %s", where, hipGetErrorString(e)); }
#define CHECK_HIP(expr, where) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hipfail(e_, where); } while (0)
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(-1, __VA_ARGS__); } while (0)

extern "C" const char* tfc_last_error(void) { return g_err.c_str(); }
extern "C" int tfc_abi_version(void) { return 2; }              // 2: deterministic reductions (part_ws arguments)
extern "C" size_t tfc_part_ws_floats(void) { return (size_t)TFC_PART_WS_FLOATS; }

static inline int pad8(int c) { return (c + 7) / 8 * 8; }
static inline int es_of(int dt) { return dt == TFC_DT_BF16 ? 2 : 4; }

// ---- profiling: hipEvents around the two MFMA kernel classes ---------------------------------------------------
// Per-THREAD state (the thread that enables profiling is the thread that launches and collects). Each record keeps the call's shape,
// so tfc_prof_records() gives a per-layer table; TFC_LAUNCH_LOG=<file> (read once) additionally appends one line per conv-class call
// "kclass op pass N H W Cin Cout flop nlaunch" -- the join key scripts/per_layer.py uses to split a rocprofv3 kernel trace by layer.
namespace {
struct ProfRec { hipEvent_t a, b; int kclass; double flop; int meta[7]; };
thread_local bool g_prof_on = false;
thread_local std::vector<ProfRec> g_prof;
thread_local std::vector<hipEvent_t> g_event_pool;
FILE* launch_log() {
  static FILE* f = [] { const char* p = getenv("TFC_LAUNCH_LOG"); return (p && *p) ? fopen(p, "a") : (FILE*)nullptr; }();
  return f;
}
hipEvent_t get_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
struct ProfScope {
  bool on;
  hipStream_t st;
  ProfRec rec;
  long long launches0;
  ProfScope(int kclass, double flop, hipStream_t s, int op, int pass, int N, int H, int W, int Cin, int Cout) : st(s) {
    on = g_prof_on;
    rec.kclass = kclass; rec.flop = flop;
    const int m[7] = {op, pass, N, H, W, Cin, Cout};
    memcpy(rec.meta, m, sizeof m);
    launches0 = g_tfc_launch_count;
    if (on) { rec.a = get_event(); rec.b = get_event(); hipEventRecord(rec.a, st); }
  }
  ~ProfScope() {
    if (on) { hipEventRecord(rec.b, st); g_prof.push_back(rec); }
    if (FILE* f = launch_log()) {
      fprintf(f, "%d %d %d %d %d %d %d %d %.0f %lld\n", rec.kclass, rec.meta[0], rec.meta[1], rec.meta[2], rec.meta[3], rec.meta[4], rec.meta[5],
              rec.meta[6], rec.flop, g_tfc_launch_count - launches0);
      fflush(f);
    }
  }
};
}  // namespace

extern "C" int tfc_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}
extern "C" int tfc_prof_collect(int kclass, double* total_ms, double* flop, long long* launches) {
  double ms = 0, fl = 0;
  long long n = 0;
  std::vector<ProfRec> keep;
  for (auto& r : g_prof) {
    if (r.kclass != kclass) { keep.push_back(r); continue; }
    float t = 0.f;
    hipError_t e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hipfail(e, "tfc_prof_collect (synchronise before collecting)");
    ms += t; fl += r.flop; ++n;
    g_event_pool.push_back(r.a);
    g_event_pool.push_back(r.b);
  }
  g_prof.swap(keep);
  if (total_ms) *total_ms = ms;
  if (flop) *flop = fl;
  if (launches) *launches = n;
  return 0;
}
// per-call detail of the records gathered so far on this thread (does not consume them): returns the number of records written
extern "C" int tfc_prof_records(int max_records, int* kclass, double* ms, double* flop, int* meta7) {
  int n = 0;
  for (auto& r : g_prof) {
    if (n >= max_records) break;
    float t = 0.f;
    hipError_t e = hipEventElapsedTime(&t, r.a, r.b);
    if (e != hipSuccess) return hipfail(e, "tfc_prof_records (synchronise before collecting)");
    if (kclass) kclass[n] = r.kclass;
    if (ms) ms[n] = t;
    if (flop) flop[n] = r.flop;
    if (meta7) memcpy(meta7 + 7 * n, r.meta, sizeof r.meta);
    ++n;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------
// descriptor construction.  (N,H,W,Cin,Cout) always describe the op's FORWARD input.
//   pass 0 = forward, 1 = dgrad (source = dy, result = dx), 2 = wgrad (forward geometry, K = pixels)
// ---------------------------------------------------------------------------------------------------------------
struct WeightMap { int Nreal, Creal; long long sn, sc; };

static int out_hw(int op, int h) { return op == TFC_OP_CONV ? h - 1 : ((op == TFC_OP_PADCONV || op == TFC_OP_CONV3) ? h : 2 * h); }
static int num_phases(int op, int pass) { return ((op == TFC_OP_CONVT || op == TFC_OP_UPCONV) && pass != 1) ? 4 : 1; }

static void set_tiles(TfcGather& d) {
  d.tiles_y = (d.GH + TFC_TILE_H - 1) / TFC_TILE_H;
  d.tiles_x = (d.GW + TFC_TILE_W - 1) / TFC_TILE_W;
}

// UPCONV: out[2a+p] reads source row a + off(p,k), off(0,.) = {-1,-1,0,0}, off(1,.) = {-1,0,0,1}
static const int kUpOff[2][4] = {{-1, -1, 0, 0}, {-1, 0, 0, 1}};

static int build_desc(int op, int pass, int phase, int N, int H, int W, int Cin, int Cout, int in_pitch, int out_pitch,
                      TfcGather* dp, WeightMap* wm) {
  TfcGather& d = *dp;
  memset(&d, 0, sizeof d);
  const int OH = out_hw(op, H), OW = out_hw(op, W);
  d.nimg = N;
  d.in_pitch = in_pitch;
  d.out_pitch = out_pitch;
  d.SS = 1; d.OS = 1; d.OOY = 0; d.OOX = 0;
  d.nplanes = 1;
  d.ph_n = 1; d.ph_d0 = 0; d.ph_oo = 0;
  const int py = phase >> 1, px = phase & 1;
  const bool fwd_geom = (pass != 1);
  if (fwd_geom) {
    d.IH = H; d.IW = W; d.Cin_pad = pad8(Cin);
    d.OH = OH; d.OW = OW; d.Nout = Cout;
    TfcPlane& p = d.plane[0];
    switch (op) {
      case TFC_OP_CONV:
      case TFC_OP_PADCONV: {
        const int o0 = (op == TFC_OP_CONV) ? -1 : -2;
        d.GH = OH; d.GW = OW;
        p.dy0 = o0; p.dx0 = o0; p.hh = 11; p.hw = 19; p.ntaps = 16;
        for (int ky = 0; ky < 4; ++ky)
          for (int kx = 0; kx < 4; ++kx) { const int t = ky * 4 + kx; p.tap_dy[t] = ky; p.tap_dx[t] = kx; p.tap_mask[t] = 1 << t; }
        if (wm) *wm = {Cout, Cin, (long long)Cin * 16, 16};
        break;
      }
      case TFC_OP_CONVT: {
        d.GH = H; d.GW = W; d.OS = 2; d.OOY = py; d.OOX = px;
        p.dy0 = py - 1; p.dx0 = px - 1; p.hh = 9; p.hw = 17; p.ntaps = 4;
        for (int jy = 0; jy < 2; ++jy)
          for (int jx = 0; jx < 2; ++jx) {
            const int t = jy * 2 + jx;
            p.tap_dy[t] = 1 - jy; p.tap_dx[t] = 1 - jx;
            p.tap_mask[t] = 1 << ((1 - py + 2 * jy) * 4 + (1 - px + 2 * jx));
          }
        if (wm) *wm = {Cout, Cin, 16, (long long)Cout * 16};   // ConvTranspose2d weight [Cin][Cout][4][4]
        break;
      }
      case TFC_OP_UPCONV: {
        // out[2a+p] reads source row a + off(p,k): the 4 filter rows hit only 2 (p=0) or 3 (p=1) distinct source rows, so the
        // duplicated taps are collapsed (their weights add): 2x2 / 2x3 / 3x2 / 3x3 distinct source offsets instead of 16 taps per
        // phase. Every phase is laid out on the SAME 3 x 3 offset grid (an offset a phase lacks gets mask 0 = zero weights): the four
        // phases then share one tap pattern and one halo geometry and fold into ONE launch like the transposed convolution.
        d.GH = H; d.GW = W; d.OS = 2; d.OOY = py; d.OOX = px;
        const int ny = 3, nx = 3;
        p.dy0 = -1; p.dx0 = -1; p.hh = 8 + ny - 1; p.hw = 16 + nx - 1; p.ntaps = ny * nx;
        for (int iy = 0; iy < ny; ++iy)
          for (int ix = 0; ix < nx; ++ix) {
            const int t = iy * nx + ix;
            int mask = 0;
            for (int ky = 0; ky < 4; ++ky)
              for (int kx = 0; kx < 4; ++kx)
                if (kUpOff[py][ky] + 1 == iy && kUpOff[px][kx] + 1 == ix) mask |= 1 << (ky * 4 + kx);
            p.tap_dy[t] = iy; p.tap_dx[t] = ix; p.tap_mask[t] = mask;
          }
        if (wm) *wm = {Cout, Cin, (long long)Cin * 16, 16};
        break;
      }
      case TFC_OP_CONV3: {
        // nn.Conv2d(k3, s1, p1) (VGG16 feature stack of LPIPS): the 3 x 3 filter is handed over zero-padded to the 4 x 4 torch layout of the other
        // ops ([Cout][Cin][4][4], taps (ky, kx) with ky, kx <= 2 used), so packing shares every code path; 3 x 3 raster = tap pattern 6
        d.GH = OH; d.GW = OW;
        p.dy0 = -1; p.dx0 = -1; p.hh = 10; p.hw = 18; p.ntaps = 9;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) { const int t = ky * 3 + kx; p.tap_dy[t] = ky; p.tap_dx[t] = kx; p.tap_mask[t] = 1 << (ky * 4 + kx); }
        if (wm) *wm = {Cout, Cin, (long long)Cin * 16, 16};
        break;
      }
      default: return fail(-1, "unknown op %d", op);
    }
  } else {
    // dgrad: source = dy [N][OH][OW][Cout], result = dx [N][H][W][Cin]
    d.IH = OH; d.IW = OW; d.Cin_pad = pad8(Cout);
    d.OH = H; d.OW = W; d.Nout = Cin;
    d.GH = H; d.GW = W;
    switch (op) {
      case TFC_OP_CONV:
      case TFC_OP_PADCONV: {
        TfcPlane& p = d.plane[0];
        const int o0 = (op == TFC_OP_CONV) ? -2 : -1;
        p.dy0 = o0; p.dx0 = o0; p.hh = 11; p.hw = 19; p.ntaps = 16;
        for (int ky = 0; ky < 4; ++ky)
          for (int kx = 0; kx < 4; ++kx) { const int t = ky * 4 + kx; p.tap_dy[t] = ky; p.tap_dx[t] = kx; p.tap_mask[t] = 1 << ((3 - ky) * 4 + (3 - kx)); }
        if (wm) *wm = {Cin, Cout, 16, (long long)Cin * 16};    // Conv2d weight [Cout][Cin][4][4], n = ci, c = co
        break;
      }
      case TFC_OP_CONV3: {
        TfcPlane& p = d.plane[0];                                 // dx[i] = sum_k dy[i + 1 - k] W[k], k = 0..2: plane offsets 0..2 from i - 1, flipped taps
        p.dy0 = -1; p.dx0 = -1; p.hh = 10; p.hw = 18; p.ntaps = 9;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) { const int t = ky * 3 + kx; p.tap_dy[t] = ky; p.tap_dx[t] = kx; p.tap_mask[t] = 1 << ((2 - ky) * 4 + (2 - kx)); }
        if (wm) *wm = {Cin, Cout, 16, (long long)Cin * 16};
        break;
      }
      case TFC_OP_CONVT: {
        // dx[i] = sum_k dy[2i-1+k] W[k]: parity plane 0 holds k in {1,3} at plane rows {i, i+1}; plane 1 holds k in {0,2} at {i-1, i}
        d.SS = 2; d.nplanes = 4;
        static const int kOf[2][2] = {{1, 3}, {0, 2}};
        for (int ppy = 0; ppy < 2; ++ppy)
          for (int ppx = 0; ppx < 2; ++ppx) {
            TfcPlane& p = d.plane[ppy * 2 + ppx];
            p.py = ppy; p.px = ppx; p.dy0 = ppy ? -1 : 0; p.dx0 = ppx ? -1 : 0; p.hh = 9; p.hw = 17; p.ntaps = 4;
            for (int jy = 0; jy < 2; ++jy)
              for (int jx = 0; jx < 2; ++jx) {
                const int t = jy * 2 + jx;
                p.tap_dy[t] = jy; p.tap_dx[t] = jx; p.tap_mask[t] = 1 << (kOf[ppy][jy] * 4 + kOf[ppx][jx]);
              }
          }
        if (wm) *wm = {Cin, Cout, (long long)Cout * 16, 16};   // W[ci][co][k]: n = ci, c = co
        break;
      }
      case TFC_OP_UPCONV: {
        // du[a] = sum over parity planes p of dy, taps k with a' = a - off(p,k):  plane row a' = a + (-off)
        d.SS = 2; d.nplanes = 4;
        for (int ppy = 0; ppy < 2; ++ppy)
          for (int ppx = 0; ppx < 2; ++ppx) {
            TfcPlane& p = d.plane[ppy * 2 + ppx];
            p.py = ppy; p.px = ppx; p.dy0 = ppy ? -1 : 0; p.dx0 = ppx ? -1 : 0; p.hh = 10; p.hw = 18; p.ntaps = 16;
            for (int ky = 0; ky < 4; ++ky)
              for (int kx = 0; kx < 4; ++kx) {
                const int t = ky * 4 + kx;
                p.tap_dy[t] = -kUpOff[ppy][ky] - p.dy0; p.tap_dx[t] = -kUpOff[ppx][kx] - p.dx0; p.tap_mask[t] = 1 << t;
              }
          }
        if (wm) *wm = {Cin, Cout, 16, (long long)Cin * 16};    // Conv2d weight [Cout][Cin][4][4]: n = ci, c = co
        break;
      }
      default: return fail(-1, "unknown op %d", op);
    }
  }
  set_tiles(d);
  // sanity: every tap must stay inside the staged halo
  for (int pl = 0; pl < d.nplanes; ++pl) {
    const TfcPlane& p = d.plane[pl];
    if (p.hh > TFC_MAX_HH || p.hw > TFC_MAX_HW) return fail(-1, "halo too large");
    for (int t = 0; t < p.ntaps; ++t)
      if (p.tap_dy[t] < 0 || p.tap_dx[t] < 0 || p.tap_dy[t] + TFC_TILE_H > p.hh || p.tap_dx[t] + TFC_TILE_W > p.hw)
        return fail(-1, "tap %d of plane %d leaves the halo (op %d pass %d)", t, pl, op, pass);
  }
  return 0;
}

// the gather-GEMM main loop consumes k-substeps in pairs
static int check_desc(const TfcGather& d, int dt) {
  for (int pl = 0; pl < d.nplanes; ++pl)
    REQUIRE(tfc_nsub(d.plane[pl].ntaps, tfc_pb(d.Cin_pad, es_of(dt))) % 2 == 0,
            "unsupported shape: %d taps with %d-byte channel chunks give an odd number of k-substeps", d.plane[pl].ntaps, tfc_pb(d.Cin_pad, es_of(dt)));
  return 0;
}

static int check_common(int dt, int op, int N, int H, int W, int Cin, int Cout) {
  REQUIRE(dt == TFC_DT_BF16 || dt == TFC_DT_F32, "bad dtype %d", dt);
  REQUIRE(op >= 0 && op <= 4, "bad op %d", op);
  REQUIRE(N > 0 && H > 1 && W > 1 && Cin > 0 && Cout > 0, "bad dims N=%d H=%d W=%d Cin=%d Cout=%d", N, H, W, Cin, Cout);
  REQUIRE((long long)N * (2 * H) * (2 * W) * (long long)(pad8(Cin) > pad8(Cout) ? pad8(Cin) : pad8(Cout)) < 2147483647LL,
          "tensor exceeds 2^31 elements");
  const int es = es_of(dt);
  REQUIRE(pad8(Cin) * es <= 64 || (pad8(Cin) * es) % 64 == 0, "Cin=%d: padded channel bytes must be <= 64 or a multiple of 64", Cin);
  REQUIRE(pad8(Cout) * es <= 64 || (pad8(Cout) * es) % 64 == 0, "Cout=%d: padded channel bytes must be <= 64 or a multiple of 64", Cout);
  // 9 taps do not fill whole 64-byte K chunks with narrow channels: the caller pads the image to 64 bytes of channels (lpips.py does)
  REQUIRE(op != TFC_OP_CONV3 || ((pad8(Cin) * es) % 64 == 0 && (pad8(Cout) * es) % 64 == 0), "TFC_OP_CONV3 needs channel bytes in multiples of 64 (Cin=%d Cout=%d)", Cin, Cout);
  return 0;
}
static int check_pitch(int dt, int pitch, int cpad, const char* what) {
  const int ue = 16 / es_of(dt);
  REQUIRE(pitch >= cpad && pitch % ue == 0, "%s pitch %d must be >= %d and a multiple of %d", what, pitch, cpad, ue);
  return 0;
}
static int check_dt(int dt) {
  REQUIRE(dt == TFC_DT_BF16 || dt == TFC_DT_F32, "bad dtype %d", dt);
  return 0;
}
static int check_ptr16(const void* p, const char* what) {
  REQUIRE(p != nullptr && (((uintptr_t)p) & 15) == 0, "%s must be a 16-byte aligned device pointer", what);
  return 0;
}

// byte offset of phase `ph` inside the packed stream of (op, pass); ph == num_phases gives the total size
static size_t phase_packed_offset(int dt, int op, int pass, int Cin, int Cout, int ph) {
  size_t off = 0;
  for (int q = 0; q < ph; ++q) {
    TfcGather d;
    if (build_desc(op, pass, q, 1, 16, 16, Cin, Cout, pad8(Cin), pad8(Cout), &d, nullptr)) return 0;
    off += tfc_packed_bytes(d, es_of(dt));
  }
  return off;
}

extern "C" size_t tfc_conv_packed_bytes(int dt, int op, int pass, int Cin, int Cout) {
  if (pass < 0 || pass > 1 || op < 0 || op > 4) return 0;
  return phase_packed_offset(dt, op, pass, Cin, Cout, num_phases(op, pass));
}

extern "C" int tfc_conv_pack(void* stream, int dt, int op, int pass, const float* w, const float* scale, void* packed, int Cin, int Cout) {
  if (int e = check_common(dt, op, 1, 16, 16, Cin, Cout)) return e;
  REQUIRE(pass == 0 || pass == 1, "pass must be 0 (fwd) or 1 (dgrad)");
  if (int e = check_ptr16(packed, "packed")) return e;
  REQUIRE(w != nullptr, "w is null");
  for (int ph = 0; ph < num_phases(op, pass); ++ph) {
    TfcGather d;
    WeightMap wm;
    if (int e = build_desc(op, pass, ph, 1, 16, 16, Cin, Cout, pad8(Cin), pad8(Cout), &d, &wm)) return e;
    CHECK_HIP(tfc_launch_pack(dt, d, w, scale, (char*)packed + phase_packed_offset(dt, op, pass, Cin, Cout, ph), wm.Nreal, wm.Creal, wm.sn, wm.sc, (hipStream_t)stream), "tfc_conv_pack");
  }
  return 0;
}

extern "C" int tfc_first_block_bwd_supported(int dt, int Cin, int Cout);
static double conv_flop(int op, int N, int H, int W, int Cin, int Cout) {
  const double oh = out_hw(op, H), ow = out_hw(op, W);
  const double taps = (op == TFC_OP_CONVT) ? 4.0 : (op == TFC_OP_CONV3 ? 9.0 : 16.0);
  return 2.0 * N * oh * ow * (double)Cin * Cout * taps;
}

extern "C" int tfc_conv_fwd(void* stream, int dt, int op, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout,
                            const void* packed, void* y, int y_pitch, const float* bias, float* stats, float* out_nchw, const float* oscale, int flags,
                            float* part_ws) {
  if (int e = check_common(dt, op, N, H, W, Cin, Cout)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(packed, "packed")) return e;
  if (int e = check_pitch(dt, x_pitch, pad8(Cin), "x")) return e;
  if (flags & TFC_EP_TANH_NCHW) { REQUIRE(out_nchw != nullptr, "out_nchw is null"); }
  else {
    REQUIRE(y != nullptr, "y is null");
    REQUIRE(y_pitch >= Cout, "y pitch %d < Cout %d", y_pitch, Cout);
    if (dt == TFC_DT_BF16 && Cout >= 8) REQUIRE(y_pitch % 8 == 0 && (((uintptr_t)y) & 15) == 0, "bf16 outputs are stored in 16-byte units: y must be 16-byte aligned with pitch %% 8 == 0");
  }
  if (flags & TFC_EP_BIAS) REQUIRE(bias != nullptr, "bias is null");
  if (flags & TFC_EP_STATS) REQUIRE(stats != nullptr && part_ws != nullptr, "TFC_EP_STATS needs stats and part_ws");
  int nph = num_phases(op, 0);
  const bool fold = (op == TFC_OP_CONVT || op == TFC_OP_UPCONV);   // equal-shaped phases: fold all four into the grid of one launch
  if (fold) nph = 1;
  ProfScope prof(0, conv_flop(op, N, H, W, Cin, Cout), (hipStream_t)stream, op, 0, N, H, W, Cin, Cout);
  for (int ph = 0; ph < nph; ++ph) {
    TfcGather d;
    if (int e = build_desc(op, 0, ph, N, H, W, Cin, Cout, x_pitch, y_pitch, &d, nullptr)) return e;
    if (fold) { d.ph_n = 4; d.ph_d0 = (op == TFC_OP_CONVT) ? 1 : 0; d.ph_oo = 1; }   // phase 0 descriptor + per-phase shifts (convT: dy0 = py-1; both: OOY = py)
    if (int e = check_desc(d, dt)) return e;
    CHECK_HIP(tfc_launch_igemm(dt, d, x, (const char*)packed + phase_packed_offset(dt, op, 0, Cin, Cout, ph), y, bias, stats, part_ws, out_nchw, oscale, flags, (hipStream_t)stream), "tfc_conv_fwd");
  }
  return 0;
}

extern "C" int tfc_conv_first_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const void* packed, void* y,
                                  int y_pitch, const float* bias, const float* oscale, int flags, uint8_t* sign_mask) {
  REQUIRE(dt == TFC_DT_BF16 && tfc_first_block_bwd_supported(dt, Cin, Cout), "tfc_conv_first_fwd: bf16, Cin <= 8, Cout == 64 (use tfc_conv_fwd otherwise)");
  if (int e = check_common(dt, TFC_OP_CONV, N, H, W, Cin, Cout)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(packed, "packed")) return e;
  if (int e = check_ptr16(y, "y")) return e;
  REQUIRE(x_pitch == 8 && y_pitch >= Cout && y_pitch % 8 == 0, "pitches: x %d (must be 8), y %d", x_pitch, y_pitch);
  REQUIRE((flags & ~(TFC_EP_BIAS | TFC_EP_LEAKY)) == 0 && (!(flags & TFC_EP_BIAS) || bias), "flags: TFC_EP_BIAS (with bias) | TFC_EP_LEAKY only");
  REQUIRE(!sign_mask || (((uintptr_t)sign_mask) & 7) == 0, "sign_mask must be 8-byte aligned");
  TfcGather d;
  if (int e = build_desc(TFC_OP_CONV, 0, 0, N, H, W, Cin, Cout, x_pitch, y_pitch, &d, nullptr)) return e;
  REQUIRE(tfc_conv_c8_eligible(d, flags), "unexpected descriptor");
  ProfScope prof(0, conv_flop(TFC_OP_CONV, N, H, W, Cin, Cout), (hipStream_t)stream, TFC_OP_CONV, 0, N, H, W, Cin, Cout);
  CHECK_HIP(tfc_launch_conv_c8(d, x, packed, y, bias, oscale, flags, sign_mask, (hipStream_t)stream), "tfc_conv_first_fwd");
  return 0;
}

extern "C" int tfc_first_block_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const void* packed, const float* bias,
                                   const float* oscale, float slope, int act_after_rounding, void* out, int out_pitch, uint8_t* sign_mask) {
  REQUIRE(dt == TFC_DT_BF16 && tfc_first_block_bwd_supported(dt, Cin, Cout), "tfc_first_block_fwd: bf16, Cin <= 8, Cout == 64");
  if (int e = check_common(dt, TFC_OP_CONV, N, H, W, Cin, Cout)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(packed, "packed")) return e;
  REQUIRE(out && (((uintptr_t)out) & 7) == 0 && out_pitch >= 64 && out_pitch % 4 == 0, "out must be 8-byte aligned with pitch >= 64 and a multiple of 4");
  REQUIRE(x_pitch == 8 && H >= 6 && W >= 6, "x pitch %d (must be 8); the reflect padding of the pooling needs a 5 x 5 convolution output at least", x_pitch);
  REQUIRE(!sign_mask || (((uintptr_t)sign_mask) & 7) == 0, "sign_mask must be 8-byte aligned");
  // 2 x (the convolution on the 1.33x overlapping regions) is what the kernel executes; the algorithmic work of the block is the convolution's
  ProfScope prof(0, conv_flop(TFC_OP_CONV, N, H, W, Cin, Cout), (hipStream_t)stream, TFC_OP_CONV, 0, N, H, W, Cin, Cout);
  CHECK_HIP(tfc_launch_first_block_fwd(x, N, H, W, packed, bias, oscale, slope, act_after_rounding, out, out_pitch, sign_mask, (hipStream_t)stream),
            "tfc_first_block_fwd");
  return 0;
}

extern "C" int tfc_conv_dgrad_image(void* stream, int dt, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout, const float* w,
                                   const float* oscale, int nch, float* dx_nchw) {
  REQUIRE(dt == TFC_DT_BF16 && Cout == 64 && Cin >= 1 && nch >= 1 && nch <= 4 && nch <= Cin, "tfc_conv_dgrad_image: bf16, Cout == 64, nch <= min(4, Cin) only");
  REQUIRE(dy && w && dx_nchw && N > 0 && H > 4 && W > 4 && dy_pitch >= 64 && dy_pitch % 8 == 0, "bad args");
  if (int e = check_ptr16(dy, "dy")) return e;
  ProfScope prof(0, 2.0 * N * (H - 1.0) * (W - 1.0) * nch * Cout * 16.0, (hipStream_t)stream, TFC_OP_CONV, 1, N, H, W, nch, Cout);
  CHECK_HIP(tfc_launch_dgrad_rows4(dy, dy_pitch, N, H, W, w, Cin, oscale, nch, dx_nchw, (hipStream_t)stream), "tfc_conv_dgrad_image");
  return 0;
}

hipError_t tfc_launch_dgrad_head(const void* dy, int dy_pitch, int N, int H, int W, const float* w, int Cout, void* dx, int dx_pitch, hipStream_t st);
extern "C" int tfc_upconv_head_dgrad(void* stream, int dt, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout, const float* w, void* dx,
                                     int dx_pitch) {
  REQUIRE(dt == TFC_DT_BF16 && Cin == 128 && Cout >= 1 && Cout <= 8, "tfc_upconv_head_dgrad: bf16, Cin == 128, Cout <= 8 only (use tfc_conv_dgrad(TFC_OP_UPCONV) otherwise)");
  REQUIRE(dy && w && dx && N > 0 && H > 1 && W > 1 && dy_pitch >= 8 && dy_pitch % 8 == 0 && dx_pitch >= 128 && dx_pitch % 8 == 0, "bad args");
  REQUIRE((long long)N * H * W * dx_pitch < 2147483647LL, "tensor exceeds 2^31 elements");
  if (int e = check_ptr16(dy, "dy")) return e;
  if (int e = check_ptr16(dx, "dx")) return e;
  if (int e = check_ptr16(w, "w")) return e;
  ProfScope prof(0, conv_flop(TFC_OP_UPCONV, N, H, W, Cin, Cout), (hipStream_t)stream, TFC_OP_UPCONV, 1, N, H, W, Cin, Cout);
  CHECK_HIP(tfc_launch_dgrad_head(dy, dy_pitch, N, H, W, w, Cout, dx, dx_pitch, (hipStream_t)stream), "tfc_upconv_head_dgrad");
  return 0;
}

extern "C" int tfc_upconv_head_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int Cin, int Cout, const float* w,
                                  const float* bias, float* out_nchw) {
  REQUIRE(dt == TFC_DT_BF16 && Cin == 128 && Cout >= 1 && Cout <= 4, "tfc_upconv_head_fwd: bf16, Cin == 128, Cout <= 4 only (use tfc_conv_fwd(TFC_OP_UPCONV) otherwise)");
  REQUIRE(x && w && out_nchw && N > 0 && H > 1 && W > 1 && x_pitch >= 128 && x_pitch % 8 == 0, "bad args");
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(w, "w")) return e;
  ProfScope prof(0, conv_flop(TFC_OP_UPCONV, N, H, W, Cin, Cout), (hipStream_t)stream, TFC_OP_UPCONV, 0, N, H, W, Cin, Cout);
  CHECK_HIP(tfc_launch_upconv_head(x, x_pitch, N, H, W, w, bias, Cout, out_nchw, (hipStream_t)stream), "tfc_upconv_head_fwd");
  return 0;
}

extern "C" int tfc_conv_dgrad(void* stream, int dt, int op, const void* dy, int dy_pitch, int N, int H, int W, int Cin, int Cout,
                              const void* packed, void* dx, int dx_pitch, const float* oscale, int flags) {
  if (int e = check_common(dt, op, N, H, W, Cin, Cout)) return e;
  if (int e = check_ptr16(dy, "dy")) return e;
  if (int e = check_ptr16(packed, "packed")) return e;
  if (int e = check_pitch(dt, dy_pitch, pad8(Cout), "dy")) return e;
  REQUIRE(dx != nullptr && dx_pitch >= Cin, "dx null or pitch %d < Cin %d", dx_pitch, Cin);
  if (dt == TFC_DT_BF16 && Cin >= 8) REQUIRE(dx_pitch % 8 == 0 && (((uintptr_t)dx) & 15) == 0, "bf16 outputs are stored in 16-byte units: dx must be 16-byte aligned with pitch %% 8 == 0");
  REQUIRE((flags & ~TFC_EP_ACCUM) == 0, "dgrad supports only TFC_EP_ACCUM");
  TfcGather d;
  if (int e = build_desc(op, 1, 0, N, H, W, Cin, Cout, dy_pitch, dx_pitch, &d, nullptr)) return e;
  if (int e = check_desc(d, dt)) return e;
  ProfScope prof(0, conv_flop(op, N, H, W, Cin, Cout), (hipStream_t)stream, op, 1, N, H, W, Cin, Cout);
  CHECK_HIP(tfc_launch_igemm(dt, d, dy, packed, dx, nullptr, nullptr, nullptr, nullptr, oscale, flags, (hipStream_t)stream), "tfc_conv_dgrad");
  return 0;
}

// ---- planned packing: host builds a job table once (geometry and buffers fixed), the caller keeps a device copy ----
struct TfcPackJob {
  TfcGather d;
  const float* w;
  void* wp;
  long long sn, sc;
  int NB32, Nreal, Creal, units;
  int first_block, threads;
};
extern "C" size_t tfc_pack_plan_bytes(int nlayers) { return (size_t)nlayers * 4 * sizeof(TfcPackJob); }
// fills plan_host with one job per (layer, pass, phase); returns the number of jobs (<0 on error) and the grid size in *nblocks
extern "C" int tfc_pack_plan_build(int dt, int nlayers, const int* ops, const int* passes, const float* const* w, void* const* packed,
                                   const int* Cin, const int* Cout, void* plan_host, int* nblocks) {
  REQUIRE(nlayers > 0 && ops && passes && w && packed && Cin && Cout && plan_host && nblocks, "bad args");
  TfcPackJob* jobs = (TfcPackJob*)plan_host;
  int nj = 0, blk = 0;
  for (int l = 0; l < nlayers; ++l) {
    if (int e = check_common(dt, ops[l], 1, 16, 16, Cin[l], Cout[l])) return e;
    REQUIRE(passes[l] == 0 || passes[l] == 1, "pass must be 0 or 1");
    for (int ph = 0; ph < num_phases(ops[l], passes[l]); ++ph) {
      TfcPackJob& j = jobs[nj];
      WeightMap wm;
      if (int e = build_desc(ops[l], passes[l], ph, 1, 16, 16, Cin[l], Cout[l], pad8(Cin[l]), pad8(Cout[l]), &j.d, &wm)) return e;
      j.w = w[l];
      j.wp = (char*)packed[l] + phase_packed_offset(dt, ops[l], passes[l], Cin[l], Cout[l], ph);
      j.sn = wm.sn; j.sc = wm.sc; j.Nreal = wm.Nreal; j.Creal = wm.Creal;
      j.NB32 = tfc_nb32_padded(j.d.Nout);
      j.units = tfc_total_substeps(j.d, es_of(dt)) * j.NB32 * 64;
      j.threads = j.NB32 * 32 * ((j.d.Cin_pad * es_of(dt)) / 16);   // one thread per (output channel, 16-byte channel unit)
      j.first_block = blk;
      blk += (j.threads + 255) / 256;
      ++nj;
    }
  }
  *nblocks = blk;
  return nj;
}
extern "C" int tfc_conv_pack_planned(void* stream, int dt, const void* plan_dev, int njobs, int nblocks) {
  REQUIRE(plan_dev && njobs > 0 && nblocks > 0, "bad args");
  CHECK_HIP(tfc_launch_pack_planned(dt, plan_dev, njobs, nblocks, (hipStream_t)stream), "tfc_conv_pack_planned");
  return 0;
}

extern "C" int tfc_patchgan_head_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int C, const float* w,
                                     void* y, int y_pitch) {
  REQUIRE(dt == TFC_DT_BF16 || dt == TFC_DT_F32, "bad dtype");
  if (int e = check_ptr16(x, "x")) return e;
  const int ue = 16 / es_of(dt);
  REQUIRE(w && y && N > 0 && H > 0 && W > 0 && C > 0 && C % ue == 0 && C <= 2048 && x_pitch >= C && x_pitch % ue == 0 && y_pitch >= 1, "bad args");
  CHECK_HIP(tfc_launch_head_fwd(dt, x, x_pitch, w, y, y_pitch, N, H, W, C, (hipStream_t)stream), "tfc_patchgan_head_fwd");
  return 0;
}

// wgrad scratch = [split-K slabs | fp32 accumulator 16 x Cout x Cin]: at most 512 workgroups x 4 waves x 8 tiles x 4 KiB = 64 MiB of slabs.
// The slabs come FIRST so that the accumulator (which must stay all-zero between calls) starts at the same offset for every layer
// that shares one scratch buffer.
static size_t wgrad_acc_bytes(int Cin, int Cout) { return (((size_t)16 * Cin * Cout * sizeof(float)) + 255) & ~(size_t)255; }
static const size_t kWgradSlabBytes = (size_t)512 * 4 * 8 * 4 * 64 * 16;
extern "C" size_t tfc_conv_wgrad_ws_bytes(int op, int Cin, int Cout) { (void)op; return wgrad_acc_bytes(Cin, Cout) + kWgradSlabBytes; }

extern "C" int tfc_conv_wgrad(void* stream, int dt, int op, const void* x, int x_pitch, const void* dy, int dy_pitch, int N, int H, int W,
                              int Cin, int Cout, void* ws, float* dw, int accumulate) {
  if (int e = check_common(dt, op, N, H, W, Cin, Cout)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(dy, "dy")) return e;
  if (int e = check_pitch(dt, x_pitch, pad8(Cin), "x")) return e;
  if (int e = check_pitch(dt, dy_pitch, pad8(Cout), "dy")) return e;
  REQUIRE(ws != nullptr && dw != nullptr, "ws / dw null");
  REQUIRE(op != TFC_OP_CONV3, "TFC_OP_CONV3 (frozen VGG features of the LPIPS term) has no weight-gradient pass");
  hipStream_t st = (hipStream_t)stream;
  WeightMap wm{};                                                // the accumulator part of ws is all-zero on entry (caller zeroes it ONCE) and again on exit
  TfcWgradFin fin{dw, 0, 0, accumulate, false};
  {
    ProfScope prof(1, conv_flop(op, N, H, W, Cin, Cout), st, op, 2, N, H, W, Cin, Cout);
    bool fused = false;
    if ((op == TFC_OP_CONVT || op == TFC_OP_UPCONV) && dt == TFC_DT_BF16) {   // all four sub-pixel phases in one launch
      TfcGather d0;
      if (int e = build_desc(op, 2, 0, N, H, W, Cin, Cout, x_pitch, dy_pitch, &d0, &wm)) return e;   // wm: the weight layout map
      fin.sn = wm.sn; fin.sc = wm.sc;
      hipError_t herr = hipSuccess;
      fused = tfc_launch_wgrad_phases_fused(op == TFC_OP_UPCONV, x, N, H, W, x_pitch, pad8(Cin), dy, dy_pitch, Cout, Cin,
                                            (float*)((char*)ws + kWgradSlabBytes), ws, st, &herr, &fin);
      if (fused) CHECK_HIP(herr, "tfc_conv_wgrad (phase-fused)");
    }
    const int nph = num_phases(op, 2);
    for (int ph = 0; ph < nph && !fused; ++ph) {
      TfcGather d;
      if (int e = build_desc(op, 2, ph, N, H, W, Cin, Cout, x_pitch, dy_pitch, &d, &wm)) return e;
      fin.sn = wm.sn; fin.sc = wm.sc;
      CHECK_HIP(tfc_launch_wgrad(dt, d, dy, x, (float*)((char*)ws + kWgradSlabBytes), ws, pad8(Cout), Cout, Cin, st, nph == 1 ? &fin : nullptr), "tfc_conv_wgrad");
    }
  }
  if (!fin.done) {
    ProfScope prof(2, 0.0, st, op, 3, N, H, W, Cin, Cout);     // class 2: the finish pass (layout + re-zero), no algorithmic FLOP of its own
    CHECK_HIP(tfc_launch_wgrad_finish((float*)((char*)ws + kWgradSlabBytes), dw, Cout, Cin, wm.sn, wm.sc, accumulate, st), "tfc_conv_wgrad finish");
  }
  return 0;
}

// ---- fused first-block backward --------------------------------------------------------------------------------------------------------
hipError_t tfc_launch_first_block_bwd(const TfcGather& d, const void* yact, int y_pitch, const void* dyp, int dyp_pitch, int Ho, int Wo, const void* in,
                                      void* slab, float* dwacc, float* rstats, float* part_ws, float slope, int Nn_real, int Cw_real,
                                      const unsigned char* sign_mask, TfcWgradFin* fin, hipStream_t st);

extern "C" int tfc_first_block_bwd_supported(int dt, int Cin, int Cout) { return dt == TFC_DT_BF16 && Cin > 0 && Cin <= 8 && Cout == 64 ? 1 : 0; }
extern "C" int tfc_first_block_bwd_wgrad(void* stream, int dt, const void* x, int x_pitch, const void* y, int y_pitch, const void* dy_pooled, int dyp_pitch,
                                         int N, int H, int W, int Cin, int Cout, float slope, void* ws, float* dw, int accumulate, float* bias_sums,
                                         float* part_ws, const uint8_t* sign_mask) {
  REQUIRE(tfc_first_block_bwd_supported(dt, Cin, Cout), "fused first-block backward: bf16, Cin <= 8, Cout == 64 (dt=%d Cin=%d Cout=%d)", dt, Cin, Cout);
  if (int e = check_common(dt, TFC_OP_CONV, N, H, W, Cin, Cout)) return e;
  REQUIRE(x && (y || sign_mask) && dy_pooled && ws && dw, "null argument (y may be null only when sign_mask is given)");
  REQUIRE(!bias_sums || part_ws, "bias_sums needs part_ws");
  REQUIRE(x_pitch == 8 && (!y || (y_pitch >= 64 && y_pitch % 8 == 0)) && dyp_pitch >= 64 && dyp_pitch % 8 == 0, "pitches: x %d (must be 8), y %d, dy %d", x_pitch, y_pitch, dyp_pitch);
  REQUIRE(H >= 4 && W >= 4, "reflect padding of the blur needs a 3 x 3 activation at least");
  if (int e = check_ptr16(x, "x")) return e;
  if (y) { if (int e = check_ptr16(y, "y")) return e; }
  if (sign_mask) REQUIRE((((uintptr_t)sign_mask) & 7) == 0, "sign_mask must be 8-byte aligned");
  if (int e = check_ptr16(dy_pooled, "dy_pooled")) return e;
  hipStream_t st = (hipStream_t)stream;
  TfcGather d;
  WeightMap wm{};
  if (int e = build_desc(TFC_OP_CONV, 2, 0, N, H, W, Cin, Cout, x_pitch, y_pitch, &d, &wm)) return e;
  REQUIRE(d.plane[0].ntaps == 16 && d.plane[0].hh <= TFC_MAX_HH && d.plane[0].hw <= TFC_MAX_HW, "unexpected descriptor");
  const int Ho = (H - 2) / 2 + 1, Wo = (W - 2) / 2 + 1;          // pooled size of the (H-1) x (W-1) activation
  {
    // class 3, not 1: this launch also carries the transposed blur that used to be an elementwise pass of its own -- keeping it out of the
    // weight-gradient class keeps that class comparable across rounds. The slab reduction writes the torch-layout gradient itself (no finish pass).
    ProfScope prof(3, conv_flop(TFC_OP_CONV, N, H, W, Cin, Cout), st, TFC_OP_CONV, 2, N, H, W, Cin, Cout);
    TfcWgradFin fin{dw, wm.sn, wm.sc, accumulate, false};
    CHECK_HIP(tfc_launch_first_block_bwd(d, y, y_pitch, dy_pooled, dyp_pitch, Ho, Wo, x, ws, (float*)((char*)ws + kWgradSlabBytes), bias_sums, part_ws, slope, Cout, Cin,
                                         sign_mask, &fin, st), "tfc_first_block_bwd_wgrad");
  }
  return 0;
}

// ---- fused activation family ------------------------------------------------------------------------------------
static int fill_act(ActParams& p, int dt, int N, int H, int W, int C, int x_pitch, int o_pitch, int norm, float slope, int pool,
                    float drop_p, uint32_t seed) {
  REQUIRE(dt == TFC_DT_BF16 || dt == TFC_DT_F32, "bad dtype");
  const int ue = 16 / es_of(dt);
  REQUIRE(C % ue == 0, "C=%d must be a multiple of %d", C, ue);
  const int cv = C / ue;
  REQUIRE(cv <= 256 && (256 % cv) == 0, "C/%d = %d must divide 256", ue, cv);
  REQUIRE(pool >= 0 && pool <= 2, "bad pool %d", pool);
  REQUIRE(pool == 0 || (H >= 3 && W >= 3), "reflect pad needs H,W >= 3");
  REQUIRE(x_pitch % ue == 0 && o_pitch % ue == 0, "pitches must be multiples of %d", ue);
  REQUIRE(drop_p >= 0.f && drop_p < 1.f, "drop_p out of range");
  p.N = N; p.H = H; p.W = W; p.C = C;
  p.Ho = pool == 2 ? (H - 1) / 2 + 1 : H;
  p.Wo = pool == 2 ? (W - 1) / 2 + 1 : W;
  p.x_pitch = x_pitch; p.o_pitch = o_pitch; p.pool = pool; p.norm = norm; p.slope = slope; p.eps = 1e-5f;
  p.drop_thresh24 = (unsigned)lrintf(drop_p * 16777216.f);
  p.seed = seed;
  p.drop_scale = 1.f / (1.f - drop_p);
  return 0;
}

extern "C" int tfc_act_fwd(void* stream, int dt, const void* x, int x_pitch, int N, int H, int W, int C, const float* stats, int norm,
                           float slope, int pool, float drop_p, uint32_t seed, void* y, int y_pitch, float* stats_out, float* part_ws) {
  ActParams p;
  if (int e = fill_act(p, dt, N, H, W, C, x_pitch, y_pitch, norm, slope, pool, drop_p, seed)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(y, "y")) return e;
  REQUIRE(!norm || stats, "stats is null");
  REQUIRE(!stats_out || part_ws, "stats_out needs part_ws");
  CHECK_HIP(tfc_launch_act_fwd(dt, p, x, stats, y, stats_out, part_ws, (hipStream_t)stream), "tfc_act_fwd");
  return 0;
}

extern "C" int tfc_act_bwd(void* stream, int dt, int mode, const void* dy, int dy_pitch, const void* x, int x_pitch, int N, int H, int W, int C,
                           const float* stats, int norm, float slope, int pool, float drop_p, uint32_t seed, float* rstats, void* dx, int dx_pitch,
                           float* part_ws) {
  ActParams p;
  if (int e = fill_act(p, dt, N, H, W, C, x ? x_pitch : C, dy_pitch, norm, slope, pool, drop_p, seed)) return e;
  REQUIRE(mode >= 0 && mode <= 2, "bad mode");
  if (int e = check_ptr16(dy, "dy")) return e;
  REQUIRE(!norm || (stats && x), "norm needs stats and x");
  REQUIRE(mode == 0 || (norm && rstats), "modes 1/2 need norm and rstats");   // mode 0: rstats (nullable) = float[C] += column sums of dx
  REQUIRE(mode == 1 || dx, "dx is null");
  REQUIRE(!(mode == 1 || (mode == 0 && rstats)) || part_ws, "a reduction into rstats needs part_ws");
  const void* xx = x ? x : dy;
  CHECK_HIP(tfc_launch_act_bwd(dt, mode, p, dy, xx, stats, rstats, dx ? dx : (void*)dy, x ? 1 : 0, dx_pitch, part_ws, (hipStream_t)stream), "tfc_act_bwd");
  return 0;
}

extern "C" int tfc_act_bwd_signs(void* stream, int dt, const void* dy, int dy_pitch, const unsigned char* sign_mask, int N, int H, int W, int C, float slope,
                                 float* rstats, void* dx, int dx_pitch, float* part_ws) {
  ActParams p;
  REQUIRE(dt == TFC_DT_BF16 && C == 64, "tfc_act_bwd_signs: bf16, 64 channels");
  if (int e = fill_act(p, dt, N, H, W, C, C, dy_pitch, 0, slope, 2, 0.f, 0)) return e;
  if (int e = check_ptr16(dy, "dy")) return e;
  REQUIRE(sign_mask && dx, "sign_mask / dx is null");
  REQUIRE(!rstats || part_ws, "a reduction into rstats needs part_ws");
  CHECK_HIP(tfc_launch_act_pool2_bwd_signs(p, dy, sign_mask, rstats, dx, dx_pitch, part_ws, (hipStream_t)stream), "tfc_act_bwd_signs");
  return 0;
}

extern "C" int tfc_dropout_mask(void* stream, uint8_t* keep, long long n, float drop_p, uint32_t seed) {
  REQUIRE(keep && n > 0, "bad args");
  CHECK_HIP(tfc_launch_dropout_mask(keep, n, seed, (unsigned)lrintf(drop_p * 16777216.f), (hipStream_t)stream), "tfc_dropout_mask");
  return 0;
}

extern "C" int tfc_pack_nhwc8(void* stream, int dt, const float* a, int Ca, const float* b, int Cb, void* out, int N, int H, int W) {
  REQUIRE(a && out && Ca > 0 && Ca + Cb <= 8 && (Cb == 0 || b), "bad args");
  CHECK_HIP(tfc_launch_pack_nhwc8(dt, a, Ca, b, Cb, out, N, H * W, (hipStream_t)stream), "tfc_pack_nhwc8");
  return 0;
}
extern "C" int tfc_unpack_nchw(void* stream, int dt, const void* in, int pitch, int c0, int C, float* out, int N, int H, int W, float alpha, float beta) {
  REQUIRE(in && out && C > 0 && c0 >= 0 && c0 + C <= pitch, "bad args");
  CHECK_HIP(tfc_launch_unpack_nchw(dt, in, pitch, c0, C, out, N, H * W, alpha, beta, (hipStream_t)stream), "tfc_unpack_nchw");
  return 0;
}
extern "C" int tfc_tanh_bwd_pack(void* stream, int dt, const float* g, const float* y, void* dyraw, float* dbias, int N, int C, int H, int W, float* part_ws) {
  REQUIRE(g && y && dyraw && C > 0 && C <= 4, "bad args (C <= 4)");
  REQUIRE(!dbias || part_ws, "dbias needs part_ws");
  CHECK_HIP(tfc_launch_tanh_bwd_pack(dt, g, y, dyraw, dbias, part_ws, N, C, H * W, (hipStream_t)stream), "tfc_tanh_bwd_pack");
  return 0;
}
extern "C" int tfc_colsum(void* stream, int dt, const void* x, long long rows, int pitch, int C, float* out, float* part_ws) {
  const int ue = 16 / es_of(dt);
  REQUIRE(x && out && rows > 0 && C % ue == 0, "bad args");
  CHECK_HIP(tfc_launch_colsum(dt, x, rows, pitch, C, out, part_ws, (hipStream_t)stream), "tfc_colsum");
  return 0;
}
extern "C" int tfc_cast(void* stream, int dt, int to_f32, const void* x, void* y, long long n) {
  REQUIRE(x && y && n > 0, "bad args");
  CHECK_HIP(tfc_launch_cast(dt, to_f32, x, y, n, (hipStream_t)stream), "tfc_cast");
  return 0;
}
extern "C" int tfc_axpby(void* stream, float* out, const float* x, const float* y, long long n, float a, float b) {
  REQUIRE(out && x && n > 0, "bad args");
  CHECK_HIP(tfc_launch_axpby(out, x, y, n, a, b, (hipStream_t)stream), "tfc_axpby");
  return 0;
}

extern "C" size_t tfc_spectral_norm_batched_ws_floats(int nlayers, const int* R, const int* K) {
  size_t n = 0;                                                   // per layer: (R / 32 row-block partials of W^T u) x K  +  s = W v (R)
  for (int i = 0; i < nlayers; ++i) n += (size_t)((R[i] + 31) / 32) * K[i] + R[i];
  return n;
}
extern "C" int tfc_spectral_norm_step_batched(void* stream, int nlayers, const float* const* W, float* const* u, float* const* v,
                                              float* const* sigma2, float* const* u_snap, float* const* v_snap, const int* R,
                                              const int* K, float* ws, int power_iter) {
  REQUIRE(nlayers >= 1 && nlayers <= 4 && W && u && v && sigma2 && R && K && ws, "bad args (1..4 layers)");
  SnBatch b{};
  b.n = nlayers;
  float* wp = ws;                                                // [t partials of layer 0 | s_0 | t partials of layer 1 | s_1 | ...]
  for (int i = 0; i < nlayers; ++i) {
    REQUIRE(W[i] && u[i] && v[i] && sigma2[i] && R[i] > 0 && K[i] > 0 && R[i] <= 4096, "bad layer %d", i);
    b.W[i] = W[i]; b.u[i] = u[i]; b.v[i] = v[i]; b.sigma2[i] = sigma2[i];
    b.us[i] = u_snap ? u_snap[i] : nullptr;
    b.vs[i] = v_snap ? v_snap[i] : nullptr;
    b.R[i] = R[i]; b.K[i] = K[i];
    b.t[i] = wp; wp += (size_t)((R[i] + 31) / 32) * K[i];
    b.s[i] = wp; wp += R[i];
  }
  CHECK_HIP(tfc_launch_sn_step_batched(b, power_iter, 1e-12f, (hipStream_t)stream), "tfc_spectral_norm_step_batched");
  return 0;
}
extern "C" int tfc_spectral_norm_step(void* stream, const float* W, float* u, float* v, float* sigma2, float* ws, int R, int K, int power_iter) {
  REQUIRE(W && u && v && sigma2 && ws && R > 0 && K > 0, "bad args");
  return tfc_spectral_norm_step_batched(stream, 1, &W, &u, &v, &sigma2, nullptr, nullptr, &R, &K, ws, power_iter);
}
extern "C" int tfc_spectral_norm_bwd(void* stream, const float* G, const float* W, const float* u, const float* v, const float* sigma2,
                                     float* ws, float* gout, int R, int K, int accumulate) {
  REQUIRE(G && W && u && v && sigma2 && ws && gout && (((uintptr_t)ws) & 7) == 0, "bad args (ws: 256 floats, 8-byte aligned)");
  CHECK_HIP(tfc_launch_sn_bwd(G, W, u, v, sigma2, ws, gout, R, K, accumulate, (hipStream_t)stream), "tfc_spectral_norm_bwd");
  return 0;
}

extern "C" int tfc_patch16_triplet(void* stream, const float* fake, const float* real, const int* neg_idx_host, int N, int C,
                                   float* loss, float* dfake, float gscale) {
  REQUIRE(fake && real && neg_idx_host && loss && N > 0 && C > 0, "bad args");
  for (int i = 0; i < 16; ++i) REQUIRE(neg_idx_host[i] >= 0 && neg_idx_host[i] < 16, "neg_idx[%d]=%d out of range", i, neg_idx_host[i]);
  CHECK_HIP(tfc_launch_triplet16(fake, real, neg_idx_host, N, C, 1.0f, 1e-6f, loss, dfake, gscale, (hipStream_t)stream), "tfc_patch16_triplet");
  return 0;
}
extern "C" size_t tfc_fft_spectrum_ws_bytes(int S, int nwin) { return (S == 64 || S == 256) && nwin > 0 ? tfc_fft_ws_bytes(S, nwin) : 0; }
extern "C" int tfc_fft_spectrum(void* stream, const float* img, long long batch_stride, long long chan_stride, int row_stride, int C, int S,
                                int wins_x, int wins_y, int N, float* amp, float* pha, int shift, void* ws) {
  REQUIRE(img && amp && pha && (S == 64 || S == 256) && (C == 1 || C == 3) && wins_x > 0 && wins_y > 0 && N > 0, "bad args");
  if (ws) { if (int e = check_ptr16(ws, "ws")) return e; }
  CHECK_HIP(tfc_launch_spectrum(img, batch_stride, chan_stride, row_stride, C, S, wins_x, wins_x * wins_y, N * wins_x * wins_y, amp, pha, shift, ws, (hipStream_t)stream), "tfc_fft_spectrum");
  return 0;
}
extern "C" int tfc_logmag_mse(void* stream, const float* amp_a, const float* amp_b, int S, int nwin, float* out) {
  REQUIRE(amp_a && amp_b && out && (S == 64 || S == 256) && nwin > 0, "bad args");
  CHECK_HIP(tfc_launch_logmag_mse(amp_a, amp_b, S, nwin, out, 0, (hipStream_t)stream), "tfc_logmag_mse");
  return 0;
}
extern "C" int tfc_logmag_mae(void* stream, const float* amp_a, const float* amp_b, int S, int nwin, float* out) {
  REQUIRE(amp_a && amp_b && out && (S == 64 || S == 256) && nwin > 0, "bad args");
  CHECK_HIP(tfc_launch_logmag_mse(amp_a, amp_b, S, nwin, out, 1, (hipStream_t)stream), "tfc_logmag_mae");
  return 0;
}
extern "C" int tfc_vectorize_temps(void* stream, const float* img, long long batch_stride, int row_stride, int N, int H, int W,
                                  const float* lut256, float* out) {
  REQUIRE(img && lut256 && out && N > 0 && H > 0 && W > 0 && row_stride >= W && batch_stride >= (long long)H * row_stride, "bad args");
  CHECK_HIP(tfc_launch_vectorize_temps(img, batch_stride, row_stride, N, H, W, lut256, out, (hipStream_t)stream), "tfc_vectorize_temps");
  return 0;
}
extern "C" int tfc_row_triplet(void* stream, const float* anchor, const float* positive, const float* negative, long long rows, int W,
                              float margin, float* loss) {
  REQUIRE(anchor && positive && negative && loss && rows > 0 && W > 0, "bad args");
  CHECK_HIP(tfc_launch_row_triplet(anchor, positive, negative, rows, W, margin, 1e-6f, loss, (hipStream_t)stream), "tfc_row_triplet");
  return 0;
}
extern "C" int tfc_affine_warp_fwd(void* stream, const float* src, const float* theta, float* out, int N, int C, int H, int W) {
  REQUIRE(src && theta && out && N > 0 && C > 0 && H > 1 && W > 1, "bad args");
  CHECK_HIP(tfc_launch_affine_warp_fwd(src, theta, out, N, C, H, W, (hipStream_t)stream), "tfc_affine_warp_fwd");
  return 0;
}
extern "C" int tfc_affine_warp_bwd(void* stream, const float* src, const float* theta, const float* gout, float* dtheta, float* dsrc, int N, int C,
                                   int H, int W, float* part_ws) {
  REQUIRE(src && theta && gout && dtheta && part_ws && N > 0 && C > 0 && H > 1 && W > 1, "bad args");
  CHECK_HIP(tfc_launch_affine_warp_bwd(src, theta, gout, dtheta, dsrc, part_ws, N, C, H, W, (hipStream_t)stream), "tfc_affine_warp_bwd");
  return 0;
}
extern "C" int tfc_morph_grad_fwd(void* stream, const float* x, float* out, uint8_t* arg, long long planes, int H, int W) {
  REQUIRE(x && out && planes > 0 && H > 0 && W > 0, "bad args");
  CHECK_HIP(tfc_launch_morph_grad_fwd(x, out, arg, planes, H, W, (hipStream_t)stream), "tfc_morph_grad_fwd");
  return 0;
}
extern "C" int tfc_morph_grad_bwd(void* stream, const float* gout, const uint8_t* arg, float* dx, long long planes, int H, int W) {
  REQUIRE(gout && arg && dx && planes > 0 && H > 0 && W > 0, "bad args");
  CHECK_HIP(tfc_launch_morph_grad_bwd(gout, arg, dx, planes, H, W, (hipStream_t)stream), "tfc_morph_grad_bwd");
  return 0;
}
extern "C" int tfc_row_triplet_grad(void* stream, const float* anchor, const float* positive, const float* negative, long long rows, int W,
                                   float margin, float gscale, float* loss, float* danchor) {
  REQUIRE(anchor && positive && negative && loss && rows > 0 && W > 0, "bad args");
  CHECK_HIP(tfc_launch_row_triplet_grad(anchor, positive, negative, rows, W, margin, 1e-6f, gscale, loss, danchor, (hipStream_t)stream), "tfc_row_triplet_grad");
  return 0;
}
// ---- input pipeline (input.hip) ---------------------------------------------------------------------------------------------------
hipError_t tfc_launch_pair_resize(const uint8_t* src, long long img_stride, int row_stride, int N, const TfcResizePlan& p, const int* plan,
                                  uint8_t* tmp, const float* lut, float* A, float* B, float* TB, uint8_t* A8, uint8_t* B8, hipStream_t st);
namespace {
// Pillow's bicubic kernel (a = -0.5) and coefficient precomputation (src/libImaging/Resample.c: bicubic_filter, precompute_coeffs,
// normalize_coeffs_8bpc), restated: the SAME double arithmetic in the same order, so the quantised taps are identical to PIL's.
double pil_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
int pil_ksize(int in_size, int out_size) {
  double fs = (double)in_size / out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(2.0 * fs) * 2 + 1;
}
// bounds[out][2], coef[out][ksize]
void pil_coeffs(int in_size, int out_size, int ksize, int* bounds, int* coef) {
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = pil_bicubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x) {
      const double v = x < xmax ? k[x] : 0.0;
      coef[xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << 22)) : (int)(0.5 + v * (1 << 22));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}
int pil_round_half_even(int w) {                                   // Image.crop rounds its float box with Python's round(): w / 2 for odd w
  if (w % 2 == 0) return w / 2;
  const int k = w / 2;                                             // w / 2 = k + 0.5
  return (k % 2 == 0) ? k : k + 1;
}
size_t resize_plan_ints(int H, int W, int out) {
  const int xs = pil_round_half_even(W);
  const int kA = pil_ksize(xs, out), kB = pil_ksize(W - xs, out), kV = pil_ksize(H, out);
  return (sizeof(TfcResizePlan) + 3) / 4 + (size_t)out * (6 + kA + kB + kV);
}
}  // namespace

extern "C" size_t tfc_resize_plan_bytes(int H, int W, int out) { return (H > 0 && W > 1 && out > 0) ? 4 * resize_plan_ints(H, W, out) : 0; }
extern "C" int tfc_resize_plan_build(int H, int W, int out, void* plan_host) {
  REQUIRE(H > 0 && W > 1 && out > 0 && plan_host, "bad args");
  REQUIRE((long long)H * W < (1 << 28), "image too large");
  TfcResizePlan p;
  memset(&p, 0, sizeof p);
  p.out = out; p.H = H; p.W = W; p.xsplit = pil_round_half_even(W);
  int* base = (int*)plan_host;
  int off = (int)((sizeof(TfcResizePlan) + 3) / 4);
  TfcResizeAxis* axes[3] = {&p.hA, &p.hB, &p.v};
  const int sizes[3] = {p.xsplit, W - p.xsplit, H};
  for (int i = 0; i < 3; ++i) {
    TfcResizeAxis& a = *axes[i];
    a.in_size = sizes[i];
    a.ksize = pil_ksize(sizes[i], out);
    a.bounds_off = off; off += 2 * out;
    a.coef_off = off; off += a.ksize * out;
    pil_coeffs(sizes[i], out, a.ksize, base + a.bounds_off, base + a.coef_off);
  }
  memcpy(plan_host, &p, sizeof p);
  return 0;
}
extern "C" size_t tfc_pair_resize_ws_bytes(int N, int H, int out) { return (N > 0 && H > 0 && out > 0) ? (size_t)N * 2 * H * out * 3 : 0; }
extern "C" int tfc_pair_resize_normalize(void* stream, const uint8_t* src, long long img_stride, int row_stride, int N, const void* plan_host,
                                         const void* plan_dev, void* ws, const float* lut256, float* A, float* B, float* TB, uint8_t* A8, uint8_t* B8) {
  REQUIRE(src && plan_host && plan_dev && ws && A && B && N > 0, "bad args");
  REQUIRE(TB == nullptr || lut256 != nullptr, "T_B needs the 256-entry temperature table");
  TfcResizePlan p;
  memcpy(&p, plan_host, sizeof p);
  REQUIRE(p.out > 0 && p.H > 0 && p.W > 1 && p.xsplit > 0 && p.xsplit < p.W, "corrupt plan header");
  REQUIRE(row_stride >= 3 * p.W && img_stride >= (long long)row_stride * p.H, "strides smaller than the %d x %d RGB image of the plan", p.W, p.H);
  CHECK_HIP(tfc_launch_pair_resize(src, img_stride, row_stride, N, p, (const int*)plan_dev, (uint8_t*)ws, lut256, A, B, TB, A8, B8, (hipStream_t)stream),
            "tfc_pair_resize_normalize");
  return 0;
}

// ---- LPIPS term (lpips.hip) ------------------------------------------------------------------------------------------------------
hipError_t tfc_launch_lpips_input(int dt, const float* x, const float* shift, const float* scale, void* out, int N, int C, long long HW, int pitch, hipStream_t st);
hipError_t tfc_launch_lpips_input_bwd(int dt, const void* g, const float* scale, float* dx, int N, int C, long long HW, int pitch, float alpha, int accumulate, hipStream_t st);
hipError_t tfc_launch_maxpool2(int dt, int bwd, const void* x, const void* dy, void* out, int N, int H, int W, int C, hipStream_t st);
hipError_t tfc_launch_relu_bwd(int dt, const void* dy, const void* y, const void* extra, void* dz, long long n, hipStream_t st);
hipError_t tfc_launch_lpips_head(int dt, const void* fx, const void* fy, const float* w, float* out, void* dfx, int N, long long HW, int C, float gscale, hipStream_t st);

extern "C" int tfc_lpips_input_fwd(void* stream, int dt, const float* x, const float* shift, const float* scale, void* out, int N, int C, int H, int W,
                                   int pitch) {
  REQUIRE(x && shift && scale && out && N > 0 && C > 0 && C <= 8 && H > 0 && W > 0 && pitch >= 8 && pitch % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  if (int e = check_ptr16(out, "out")) return e;
  // the kernel writes channels [0, 8) of every pixel; channels [8, pitch) keep what the caller put there (zeros)
  CHECK_HIP(tfc_launch_lpips_input(dt, x, shift, scale, out, N, C, (long long)H * W, pitch, (hipStream_t)stream), "tfc_lpips_input_fwd");
  return 0;
}
extern "C" int tfc_lpips_input_bwd(void* stream, int dt, const void* g, const float* scale, float* dx, int N, int C, int H, int W, int pitch, float alpha,
                                   int accumulate) {
  REQUIRE(g && scale && dx && N > 0 && C > 0 && C <= 8 && H > 0 && W > 0 && pitch >= 8 && pitch % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  CHECK_HIP(tfc_launch_lpips_input_bwd(dt, g, scale, dx, N, C, (long long)H * W, pitch, alpha, accumulate, (hipStream_t)stream), "tfc_lpips_input_bwd");
  return 0;
}
extern "C" int tfc_maxpool2_fwd(void* stream, int dt, const void* x, void* y, int N, int H, int W, int C) {
  REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(y, "y")) return e;
  CHECK_HIP(tfc_launch_maxpool2(dt, 0, x, nullptr, y, N, H, W, C, (hipStream_t)stream), "tfc_maxpool2_fwd");
  return 0;
}
extern "C" int tfc_maxpool2_bwd(void* stream, int dt, const void* x, const void* dy, void* dx, int N, int H, int W, int C) {
  REQUIRE(x && dy && dx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  if (int e = check_ptr16(x, "x")) return e;
  if (int e = check_ptr16(dy, "dy")) return e;
  if (int e = check_ptr16(dx, "dx")) return e;
  CHECK_HIP(tfc_launch_maxpool2(dt, 1, x, dy, dx, N, H, W, C, (hipStream_t)stream), "tfc_maxpool2_bwd");
  return 0;
}
extern "C" int tfc_relu_bwd(void* stream, int dt, const void* dy, const void* y, const void* extra, void* dz, long long n) {
  REQUIRE(dy && y && dz && n > 0 && n % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  if (int e = check_ptr16(dy, "dy")) return e;
  if (int e = check_ptr16(y, "y")) return e;
  if (int e = check_ptr16(dz, "dz")) return e;
  if (extra) { if (int e = check_ptr16(extra, "extra")) return e; }
  CHECK_HIP(tfc_launch_relu_bwd(dt, dy, y, extra, dz, n, (hipStream_t)stream), "tfc_relu_bwd");
  return 0;
}
extern "C" int tfc_lpips_head(void* stream, int dt, const void* fx, const void* fy, const float* w, float* out, void* dfx, int N, int H, int W, int C,
                              float gscale) {
  REQUIRE(fx && fy && w && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bad args");
  if (int e = check_dt(dt)) return e;
  if (int e = check_ptr16(fx, "fx")) return e;
  if (int e = check_ptr16(fy, "fy")) return e;
  if (dfx) { if (int e = check_ptr16(dfx, "dfx")) return e; }
  CHECK_HIP(tfc_launch_lpips_head(dt, fx, fy, w, out, dfx, N, (long long)H * W, C, gscale, (hipStream_t)stream), "tfc_lpips_head");
  return 0;
}
extern "C" int tfc_l1_sum(void* stream, const float* a, const float* b, long long n, float scale, float* out, int zero_first) {
  REQUIRE(a && b && out && n > 0, "bad args");
  if (zero_first) CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream), "tfc_l1_sum memset");
  CHECK_HIP(tfc_launch_l1_sum(a, b, n, scale, out, (hipStream_t)stream), "tfc_l1_sum");
  return 0;
}
extern "C" int tfc_bce_relativistic(void* stream, int dt, const void* a, const void* b, int n, int stride, float t1, float t2, int mode,
                                    float* loss, void* da, void* db, float gscale) {
  REQUIRE(a && b && loss && n > 0 && stride > 0 && (mode == 0 || mode == 1), "bad args");
  CHECK_HIP(tfc_launch_bce_rel(dt, a, b, n, stride, t1, t2, mode, loss, da, db, gscale, (hipStream_t)stream), "tfc_bce_relativistic");
  return 0;
}
extern "C" int tfc_adam_step(void* stream, float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                             int step, float gscale) {
  REQUIRE(p && g && m && v && n > 0 && step >= 1, "bad args");
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2 = 1.f - powf(b2, (float)step);
  CHECK_HIP(tfc_launch_adam(p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2), gscale, (hipStream_t)stream), "tfc_adam_step");
  return 0;
}
extern "C" int tfc_debug_set_igemm_config(int cfg) {
  REQUIRE(cfg >= -1 && cfg <= 63, "cfg must be -1 (heuristic) or tile 0 (128x128) | 1 (128x64) | 2 (128x32) | 3 (128x128, 4 n-waves) | 15 (heuristic tile), "
          "optionally + 16 = the one-tile-per-workgroup gather GEMM instead of the persistent one, + 32 = the persistent kernel may take its two-half form");
  g_tfc_force_cfg = cfg;
  return 0;
}
extern "C" int tfc_probe_mfma(void* stream, float* out) {
  REQUIRE(out, "out is null");
  CHECK_HIP(tfc_launch_probe(out, (hipStream_t)stream), "tfc_probe_mfma");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// host emulator of the gather model (CPU tests only): consumes the SAME descriptors and the SAME packed operand
// stream layout as the kernels (fp32 units), so descriptor / packing / phase bugs show up without a GPU.
//   pass 0: x [N][H][W][pad8(Cin)]    -> y [N][OH][OW][pad8(Cout)]
//   pass 1: x = dy [N][OH][OW][pad8(Cout)] -> y = dx [N][H][W][pad8(Cin)]
//   pass 2: x [N][H][W][pad8(Cin)], w_host = dy [N][OH][OW][pad8(Cout)] -> y = dw (torch layout)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int tfc_host_emulate_conv(int op, int pass, int es, const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout) {
  REQUIRE(pass >= 0 && pass <= 2 && op >= 0 && op <= 4 && (es == 2 || es == 4), "bad op/pass/es");
  const int UE = 16 / es;                                        // elements per 16-byte unit in the emulated geometry
  const int OH = out_hw(op, H), OW = out_hw(op, W);
  const int inC = pass == 1 ? pad8(Cout) : pad8(Cin);
  const int outC = pass == 1 ? pad8(Cin) : pad8(Cout);
  if (pass == 2) {
    std::vector<double> acc((size_t)16 * Cout * Cin, 0.0);
    WeightMap wm{};
    for (int ph = 0; ph < num_phases(op, 2); ++ph) {
      TfcGather d;
      if (int e = build_desc(op, 2, ph, N, H, W, Cin, Cout, inC, outC, &d, &wm)) return e;
      const TfcPlane& p = d.plane[0];
      for (int img = 0; img < N; ++img)
        for (int a = 0; a < d.GH; ++a)
          for (int b = 0; b < d.GW; ++b) {
            const int oy = a * d.OS + d.OOY, ox = b * d.OS + d.OOX;
            const float* dyp = w + ((size_t)(img * d.OH + oy) * d.OW + ox) * outC;
            const int ta = a % TFC_TILE_H, tb = b % TFC_TILE_W, a0 = a - ta, b0 = b - tb;
            for (int t = 0; t < p.ntaps; ++t) {
              const int sy = (a0 + p.dy0 + ta + p.tap_dy[t]) * d.SS + p.py, sx = (b0 + p.dx0 + tb + p.tap_dx[t]) * d.SS + p.px;
              if (sy < 0 || sy >= d.IH || sx < 0 || sx >= d.IW) continue;
              const float* xp = x + ((size_t)(img * d.IH + sy) * d.IW + sx) * inC;
              for (int n = 0; n < Cout; ++n)
                for (int c = 0; c < Cin; ++c)
                  for (int m = p.tap_mask[t]; m; m &= m - 1)
                    acc[((size_t)__builtin_ctz(m) * Cout + n) * Cin + c] += (double)dyp[n] * xp[c];
            }
          }
    }
    for (int n = 0; n < Cout; ++n)
      for (int c = 0; c < Cin; ++c)
        for (int s = 0; s < 16; ++s) y[(long long)n * wm.sn + (long long)c * wm.sc + s] = (float)acc[((size_t)s * Cout + n) * Cin + c];
    return 0;
  }
  for (int ph = 0; ph < num_phases(op, pass); ++ph) {
    TfcGather d;
    WeightMap wm;
    if (int e = build_desc(op, pass, ph, N, H, W, Cin, Cout, inC, outC, &d, &wm)) return e;
    // pack on the host with the kernel's index math
    const int NB32 = tfc_nb32_padded(d.Nout);
    const int total_sub = tfc_total_substeps(d, es);
    const int total_units = total_sub * NB32 * 64;
    std::vector<float> wp((size_t)total_units * UE);
    for (int idx = 0; idx < total_units; ++idx) {
      int n, mask, c0;
      tfc_pack_locate(d, es, NB32, idx, &n, &mask, &c0);
      for (int e = 0; e < UE; ++e) {
        const int c = c0 + e;
        float a = 0.f;
        if (n < wm.Nreal && c < wm.Creal)
          for (int m = mask; m; m &= m - 1) a += w[(long long)n * wm.sn + (long long)c * wm.sc + __builtin_ctz(m)];
        wp[(size_t)idx * UE + e] = a;
      }
    }
    const int PB = tfc_pb(d.Cin_pad, es), UPP = PB >> 4, CK = PB / es;
    const int nchunks = d.Cin_pad * es / PB;
    for (int img = 0; img < N; ++img)
      for (int tyb = 0; tyb < d.tiles_y; ++tyb)
        for (int txb = 0; txb < d.tiles_x; ++txb) {
          const int a0 = tyb * TFC_TILE_H, b0 = txb * TFC_TILE_W;
          for (int ty = 0; ty < TFC_TILE_H; ++ty)
            for (int tx = 0; tx < TFC_TILE_W; ++tx) {
              const int a = a0 + ty, b = b0 + tx;
              if (a >= d.GH || b >= d.GW) continue;
              const int oy = a * d.OS + d.OOY, ox = b * d.OS + d.OOX;
              for (int n = 0; n < d.Nout; ++n) {
                const int nb = n >> 5, rn = n & 31;
                double acc = 0.0;
                int gs = 0;
                for (int cc = 0; cc < nchunks; ++cc)
                  for (int pl = 0; pl < d.nplanes; ++pl) {
                    const TfcPlane& p = d.plane[pl];
                    const int nsub = tfc_nsub(p.ntaps, PB);
                    for (int s = 0; s < nsub; ++s, ++gs)
                      for (int h = 0; h < 2; ++h) {
                        const int u = 2 * s + h, tap = u / UPP, g = u % UPP;
                        const int sy = (a0 + p.dy0 + ty + p.tap_dy[tap]) * d.SS + p.py;
                        const int sx = (b0 + p.dx0 + tx + p.tap_dx[tap]) * d.SS + p.px;
                        if (sy < 0 || sy >= d.IH || sx < 0 || sx >= d.IW) continue;
                        const float* xp = x + ((size_t)(img * d.IH + sy) * d.IW + sx) * inC + cc * CK + g * UE;
                        const float* wq = &wp[((size_t)(gs * NB32 + nb) * 64 + (h * 32 + rn)) * UE];
                        for (int e = 0; e < UE; ++e) acc += (double)xp[e] * wq[e];
                      }
                  }
                y[((size_t)(img * d.OH + oy) * d.OW + ox) * outC + n] = (float)acc;
              }
            }
        }
  }
  return 0;
}
