// nint_common.h -- shared device/host helpers for the gfx950 ConvLSTM kernels.
// Written for CDNA4 only: 64-wide waves, MFMA 16x16 tiles, 160 KiB LDS, no other targets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nint.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define NINT_CHECK_HIP(expr)                      \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

#define NINT_LAUNCH_CHECK()                       \
  do {                                            \
    hipError_t _e = hipGetLastError();            \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

// Element-type traits. One K-step of the implicit GEMMs always covers 64 BYTES of channels
// per pixel: 32 bf16 (one v_mfma_f32_16x16x32_bf16) or 16 f32 (four v_mfma_f32_16x16x4_f32).
template <int DT> struct Elem;
template <> struct Elem<NINT_F32> {
  typedef float type;
  static constexpr int ES = 4;    // bytes per element
  static constexpr int KC = 16;   // channels per K-step
  static constexpr int EPL = 4;   // elements per lane per K-step (16 bytes)
};
template <> struct Elem<NINT_BF16> {
  typedef __bf16 type;
  static constexpr int ES = 2;
  static constexpr int KC = 32;
  static constexpr int EPL = 8;
};

__host__ __device__ inline int nint_round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline int nint_cdiv(int a, int b) { return (a + b - 1) / b; }

// float -> bf16 round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// and keeps NaN a NaN (MI355X_MICROARCH.md, correctness boundaries).
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
// two floats -> one dword of two bf16 (low half = first): a single v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }

template <int DT> __device__ __forceinline__ float load_elem(const void* p, size_t i);
template <> __device__ __forceinline__ float load_elem<NINT_F32>(const void* p, size_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float load_elem<NINT_BF16>(const void* p, size_t i) { return bf2f(((const uint16_t*)p)[i]); }
template <int DT> __device__ __forceinline__ void store_elem(void* p, size_t i, float v);
template <> __device__ __forceinline__ void store_elem<NINT_F32>(void* p, size_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void store_elem<NINT_BF16>(void* p, size_t i, float v) { ((uint16_t*)p)[i] = f2bf(v); }

// 4 consecutive elements (16-byte f32 / 8-byte bf16 vector store); i must be a multiple of 4
template <int DT> __device__ __forceinline__ void store_vec4(void* p, size_t i, f32x4_t v);
template <> __device__ __forceinline__ void store_vec4<NINT_F32>(void* p, size_t i, f32x4_t v) {
  *(f32x4_t*)((float*)p + i) = v;
}
template <> __device__ __forceinline__ void store_vec4<NINT_BF16>(void* p, size_t i, f32x4_t v) {
  u32x2_t w;
  w[0] = pack_bf16x2(v[0], v[1]);
  w[1] = pack_bf16x2(v[2], v[3]);
  *(u32x2_t*)((uint16_t*)p + i) = w;
}

// 4 consecutive elements as f32 (16-byte f32 / 8-byte bf16 vector load); i must be a multiple of 4
template <int DT> __device__ __forceinline__ f32x4_t load_vec4(const void* p, size_t i);
template <> __device__ __forceinline__ f32x4_t load_vec4<NINT_F32>(const void* p, size_t i) { return *(const f32x4_t*)((const float*)p + i); }
template <> __device__ __forceinline__ f32x4_t load_vec4<NINT_BF16>(const void* p, size_t i) {
  const u32x2_t w = *(const u32x2_t*)((const uint16_t*)p + i);
  return (f32x4_t){__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                   __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
}

// One K-step of D += A*B on a 16x16 tile from two 16-byte fragments.
//   bf16: lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15], j=0..7
//   f32 : four MFMAs; in MFMA j lane l supplies A[row l&15][k = l>>4] = element j of its fragment,
//         i.e. channel 4*(l>>4)+j of the K-step -- the same bytes-per-lane geometry as bf16.
template <int DT> __device__ __forceinline__ f32x4_t mma_step(u32x4_t a, u32x4_t b, f32x4_t c);
template <> __device__ __forceinline__ f32x4_t mma_step<NINT_BF16>(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mma_step<NINT_F32>(u32x4_t a, u32x4_t b, f32x4_t c) {
  f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], c, 0, 0, 0);
  return c;
}

// Reciprocal by v_rcp_f32 (1 ulp): an IEEE division costs ~10 VALU instructions (div_scale x2, rcp, 4 fma,
// div_fmas, div_fixup) and the LSTM epilogue does five per element -- it was half of the epilogue's VALU work.
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
// tanh(x) = 2*sigmoid(2x) - 1: five instructions (mul, exp, add, rcp, fma); saturates cleanly (exp overflow ->
// inf -> rcp 0 -> -1; underflow -> 0 -> rcp(1) -> 1); absolute error ~1e-7 (cancellation near 0 is absolute, not relative)
__device__ __forceinline__ float tanhf_(float x) { return fmaf(2.0f, rcpf_(1.0f + __expf(-2.0f * x)), -1.0f); }

// LSTM pointwise backward (pointwise.hip: lstm_bwd_pointwise_kernel; also a problem of conv_bwd_multi_kernel, nint_seq.wave = 4):
// block `blk` of `nblk` 256-thread blocks, grid-stride over (pixel, 4 consecutive hidden channels).
struct PwArgs {
  const void* gates; const float* c_prev; const float* c_new; const void* dh; const void* dh2; float* dc; void* dG;
  int N, H, W, P, Hh, Wh, Ch16, Chp, dc_zero;
};
template <int DT, int U = 1>
__device__ __forceinline__ void lstm_bwd_pointwise_body(const PwArgs& a, size_t blk, size_t nblk) {
  // U items per thread and loop turn, all loads of a turn issued before the first store (U = 2 inside conv_bwd_multi_kernel, where
  // the pass runs at that kernel's occupancy -- 4 workgroups per CU instead of 8 -- and needs the loads in flight per thread instead)
  const int nq = a.Ch16 >> 2;
  const size_t total = (size_t)a.N * a.H * a.W * nq;
  const int Gc = 4 * a.Ch16;
  const size_t stride = nblk * 256;
  for (size_t i0 = blk * 256 + threadIdx.x; i0 < total; i0 += stride * U) {
    f32x4_t gi[U], gf[U], gg[U], go[U], cp[U], cn[U], dhv[U], dcv[U];
    size_t ci[U], ob[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * stride;
      on[u] = i < total;
      if (!on[u]) continue;
      const int q = i % nq;
      const size_t pix = i / nq;
      const int x = pix % a.W;
      size_t r = pix / a.W;
      const int y = r % a.H;
      const int n = r / a.H;
      const int ch = 4 * q;
      const int cblock = ch >> 4, col = ch & 15;
      const size_t gb = pix * Gc + (size_t)cblock * 64 + col;
      gi[u] = load_vec4<DT>(a.gates, gb);
      gf[u] = load_vec4<DT>(a.gates, gb + 16);
      gg[u] = load_vec4<DT>(a.gates, gb + 32);
      go[u] = load_vec4<DT>(a.gates, gb + 48);
      ci[u] = pix * a.Chp + ch;
      cp[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (a.c_prev) cp[u] = *(const f32x4_t*)(a.c_prev + ci[u]);
      cn[u] = *(const f32x4_t*)(a.c_new + ci[u]);
      dhv[u] = load_vec4<DT>(a.dh, ci[u]);
      if (a.dh2) dhv[u] += load_vec4<DT>(a.dh2, ci[u]);  // d/dh in two pieces (nint_seq.wave = 4: the x columns of the layer above + the layer's own h columns)
      dcv[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (!a.dc_zero) dcv[u] = *(const f32x4_t*)(a.dc + ci[u]);
      ob[u] = ((((size_t)n * a.Hh) + (y + a.P)) * a.Wh + (x + a.P)) * Gc + (size_t)cblock * 64 + col;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!on[u]) continue;
      f32x4_t o_i, o_f, o_g, o_o, dcp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float tc = tanhf_(cn[u][e]);
        const float dct = dcv[u][e] + dhv[u][e] * go[u][e] * (1.f - tc * tc);
        const float d_o = dhv[u][e] * tc;
        o_i[e] = dct * gg[u][e] * gi[u][e] * (1.f - gi[u][e]);
        o_f[e] = dct * cp[u][e] * gf[u][e] * (1.f - gf[u][e]);
        o_g[e] = dct * gi[u][e] * (1.f - gg[u][e] * gg[u][e]);
        o_o[e] = d_o * go[u][e] * (1.f - go[u][e]);
        dcp[e] = dct * gf[u][e];
      }
      store_vec4<DT>(a.dG, ob[u], o_i);
      store_vec4<DT>(a.dG, ob[u] + 16, o_f);
      store_vec4<DT>(a.dG, ob[u] + 32, o_g);
      store_vec4<DT>(a.dG, ob[u] + 48, o_o);
      *(f32x4_t*)(a.dc + ci[u]) = dcp;
    }
  }
}

// launcher-side descriptors -----------------------------------------------------------------
struct ConvArgs {
  const char* src0;      // halo slab (x for fwd, dG for dgrad)
  const char* src1;      // second halo slab (h_prev) or nullptr
  int nchunk0, nchunk1;  // 64-byte channel chunks taken from src0 / src1
  long img_stride0, img_stride1;  // bytes per image
  int pix_stride0, pix_stride1;   // bytes per pixel
  const char* Bp;        // packed weights [S][NTt][64 lanes][16 B]
  int NTt;               // n-tiles per K-step in Bp
  int nt_begin;          // first n-tile this launch computes
  int k, p, taps;
  int kx0;               // horizontal taps of src0: k, or 1 for a horizontally folded x source (nint_layer.xfold)
  int H, W, P, Hh, Wh;
  int tiles_x, tiles_y;
  int tiles_full_y, tiles_x2, n_full;   // full tile rows, merged tiles per image (0: none), number of full tiles in the launch
  int nhp_pad2;                         // halo-tile pixels of a merged tile (MT/2 rows x 32 pixels), rounded up to 16
  unsigned magic_nhpp2, magic_hwt2;
  int tile_rows;         // 0 = per launch shape, 4 / 8 = forced tile height
  int cpf;               // channel chunks per LDS A fill
  int a_bytes;           // bytes reserved for the A image
  int nhp_pad;           // halo-tile pixels rounded up to 16 (one g-plane of the A image)
  unsigned magic_nhpp, magic_hwt;   // ceil(2^32 / nhp_pad), ceil(2^32 / (16 + 2p)): division by multiply-high in the fill
  // LSTM epilogue
  const float* bias;     // [4*Ch16] permuted
  const float* c_prev;   // compact [N][H][W][Chp] or nullptr (= 0)
  float* c_out;
  char* h_out;           // halo slab, ET
  char* gates_out;       // stash [N][H][W][4*Ch16] ET or nullptr
  int Chp, Ch16;
  // DGRAD epilogue
  char* out0;            // += columns [0, C0p)  (ET compact)
  char* out1;            // =  columns [C0p, C0p+C1p)  (ET compact)
  int C0p, C1p;
  int out0_overwrite;    // 1: columns [0, C0p) are stored, not accumulated (the destination is known to be zero)
  // DGRAD_PW epilogue: the pointwise LSTM backward of the previous time step on the h columns (Chp / Ch16 as above)
  const char* pw_gates;  // gate stash of that step [N][H][W][4*Ch16] ET
  const float* pw_c_prev;// c_{t-1} compact f32 or nullptr (= 0)
  const float* pw_c_new; // c_t
  float* pw_dc;          // d/dc, read and overwritten in place
  const char* pw_old;    // the x columns the layer above left for this step (ET compact [N][H][W][Chp]) or nullptr
  char* pw_dG;           // dG halo slab of that step
  // ... and of the layer BELOW at this launch's own time step, on the x columns (lo_gates == nullptr: x columns stored)
  const char* lo_gates; const float* lo_c_prev; const float* lo_c_new; float* lo_dc; char* lo_dG;
  int lo_Ch16, lo_dc_zero;
};

// In-step timing probe (nint_seq.probe): a one-thread launch that writes {tag, s_memrealtime} into the caller's buffer.
// Bracketing a launch with two of them costs two ordinary kernel boundaries (no event / barrier packets, which measured
// +8-10 us per bracketed launch); the back-to-back calibration pair at the start of each pass prices those boundaries.
struct Probe {
  unsigned long long* buf; int cap; int n; unsigned mask; hipStream_t st;
  void stamp(unsigned kind, int layer, int t, int end);                    // no-op unless bit `kind` of mask is set (kind 0: always)
};

// internal entry points shared between translation units (not part of the C ABI)
int nint_internal_cell_bwd_pointwise(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* gates,
                                     const float* c_prev, const float* c_new, const void* dh, float* dc, void* dG,
                                     bool dc_zero, void* stream, const void* dh2 = nullptr,    // dh2: a second piece of d/dh, added
                                     PwArgs* plan = nullptr);                                  // plan: nothing enqueued, *plan describes the launch
struct WgJob {           // one layer's weight / bias gradient
  const nint_layer* ly; int N;
  const void* dG; const void* x_slab; const void* h_slab;
  float* dW; float* db;
  int h_skip;                                // leading images whose h source is identically zero
};
int nint_internal_conv_wgrad_multi(const WgJob* jobs, int njobs, const nint_geom* g, int dtype, float* partial,
                                   size_t partial_bytes, int n_cu, void* stream, Probe* probe = nullptr);
// stencil.hip: the gate step of tiny hidden widths (Ch <= 8, 3x3) on the vector ALU.  Its weight image -- one row of 32 f32 per
// (tap, channel) in the kernel's iteration order -- sits behind the MFMA images in the Wf buffer (nint_pack_weights).
__host__ __device__ inline bool nint_stencil_shape(int Cx, int Ch, int k, int xfold) {
  return k == 3 && Ch <= 8 && (xfold ? 3 * Cx <= 64 : Cx <= 16);
}
__host__ __device__ inline int nint_stencil_rows(int Cx, int Ch, int xfold) {     // rows per vertical tap: x part + h part
  return (xfold ? 4 * nint_cdiv(3 * Cx, 4) : 12 * nint_cdiv(Cx, 4)) + 12 * nint_cdiv(Ch, 4);
}
__host__ __device__ inline size_t nint_internal_stencil_offset(int Cxp, int Chp, int Ch16, int k, int dtype) {
  const size_t img = (size_t)(Cxp + Chp) * 4 * Ch16 * k * k * (dtype == NINT_BF16 ? 2 : 4);    // both MFMA images fit in this
  return (img + 255) / 256 * 256;
}
// tiny_gemm.hip: the same layers on the matrix pipe with a DENSE K: K is a list of 16-byte groups (8 bf16 / 4 f32 channels of one
// halo pixel of one source), four groups per K-step.  Group order: the x source's groups (folded: per vertical tap the centre
// pixel's xg groups; plain: per tap (ky, kx) the pixel's xg groups), then the h source's (per tap, hg groups).
constexpr int NINT_TINY_MAXSTEPS = 12, NINT_TINY_HW = 34;      // K-steps a wave keeps in registers; halo tile width (32 + 2)
__host__ __device__ inline int nint_tiny_xg(int Cx, int xfold, int dtype) { return nint_cdiv((xfold ? 3 * Cx : Cx) * (dtype == NINT_BF16 ? 2 : 4), 16); }
__host__ __device__ inline int nint_tiny_hg(int Ch, int dtype) { return nint_cdiv(Ch * (dtype == NINT_BF16 ? 2 : 4), 16); }
__host__ __device__ inline int nint_tiny_ngx(int Cx, int xfold, int dtype) { return (xfold ? 3 : 9) * nint_tiny_xg(Cx, xfold, dtype); }
__host__ __device__ inline bool nint_tiny_shape(int Cx, int Ch, int k, int xfold, int dtype) {
  return nint_stencil_shape(Cx, Ch, k, xfold) &&
         nint_cdiv(nint_tiny_ngx(Cx, xfold, dtype) + 9 * nint_tiny_hg(Ch, dtype), 4) <= NINT_TINY_MAXSTEPS;
}
__host__ __device__ inline size_t nint_tiny_bytes() { return (size_t)NINT_TINY_MAXSTEPS * 2 * 1024 + 4 * NINT_TINY_MAXSTEPS * sizeof(int) + 64; }
__host__ __device__ inline size_t nint_internal_tiny_offset(int Cx, int Cxp, int Ch, int Chp, int Ch16, int k, int xfold, int dtype) {
  const size_t o = nint_internal_stencil_offset(Cxp, Chp, Ch16, k, dtype) + ((size_t)3 * nint_stencil_rows(Cx, Ch, xfold) + 1) * 128;
  return (o + 255) / 256 * 256;
}
int nint_internal_tiny_lstm(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* x_slab,
                            const void* h_prev, const float* c_prev, void* h_out, float* c_out, void* gates_out,
                            void* stream);
bool nint_internal_stencil_holds(const nint_layer* ly);
int nint_internal_stencil_lstm(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* x_slab,
                               const void* h_prev, const float* c_prev, void* h_out, float* c_out, void* gates_out,
                               void* stream);
#define NINT_MULTI_MAX 4  // problems per merged grid (conv_lstm_multi_kernel / conv_bwd_multi_kernel)
struct CellFwdJob {      // one gate launch (nint_cell_fwd's arguments)
  const nint_layer* ly; const void* x_slab; const void* h_prev; const float* c_prev; void* h_out; float* c_out; void* gates_out;
};
// A conv_igemm launch that is planned but not enqueued: the sequence drivers collect independent ones and enqueue them as ONE
// grid (nint_internal_conv_multi; NINT_E_SHAPE = not possible for these shapes, nothing enqueued).
struct ConvPlan {
  ConvArgs a; int gx, gy; size_t lds; int variant;   // grid, dynamic LDS, kernel shape (EPI, WN, WK, NTW, MT)
};
int nint_internal_cell_fwd_plan(const CellFwdJob* j, const nint_geom* g, int dtype, int N, ConvPlan* plan);
int nint_internal_conv_multi(const ConvPlan* plans, int n, int dtype, void* stream, const PwArgs* pw = nullptr,    // pw: one more problem, a pointwise backward pass
                             bool dry_run = false);                                                          // dry_run: NINT_OK / NINT_E_SHAPE, nothing enqueued
struct DgradPw {         // fused pointwise backward of the previous time step (EPI_DGRAD_PW)
  const void* gates; const float* c_prev; const float* c_new; float* dc; const void* old; void* dG_out;
  // optional: the layer below's pointwise backward of THIS time step, run on the x columns (the layer's dh buffer is only read)
  const void* lo_gates; const float* lo_c_prev; const float* lo_c_new; float* lo_dc; void* lo_dG_out; int lo_Ch16; bool lo_dc_zero;
  int tile_rows;          // 0 = the layer's choice, 4 / 8 = this launch's tile height
};
int nint_internal_conv_dgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* dG, void* dx_accum,
                             void* dh_prev, bool overwrite_dx, const DgradPw* pw, void* stream, ConvPlan* plan = nullptr);
