// stencil.hip -- the ConvLSTM gate step for TINY hidden widths on the vector ALU (gfx950).
//
//   gates = W (*) cat[x,h] + b, sigmoid/tanh, c/h update      (reference model.py:216-231)
//
// With 4*Ch <= 32 gate columns the convolution is no dense contraction: on the MFMA path of conv_igemm.hip BASELINE
// configs[0] (4 input channels, 8 hidden, 3x3) keeps 14 % of its multiply-accumulates and reads 8 x padded channel bytes.
// This kernel is the stencil form north_star names for that case:
//   - one LANE per pixel, all 4*Ch (<= 32) gate pre-activations of the pixel in 32 accumulator registers;
//   - a workgroup (4 waves) owns 8 rows x 32 pixels, a wave 4 rows x 16 pixels, so a 16-lane DPP row is 16 consecutive
//     longitudes of one image row; the (8+2) x (32+2) halo tile of both sources is staged once in LDS by coalesced
//     16-byte loads along the longitude stride (only the REAL channels travel, not the 64-byte K-chunk padding);
//   - vertical taps are LDS reads of the lane's own column; HORIZONTAL taps are wave shuffles: the left / right
//     neighbour's channel quad comes by DPP row_shr:1 / row_shl:1, only the two edge lanes of a row read the halo column
//     from LDS (a horizontally folded x source -- nint_layer.xfold -- carries its horizontal taps in its channels);
//   - weights are wave-uniform: one row of 32 f32 per (tap, channel) in iteration order (nint_pack_weights writes that
//     image behind the MFMA images), fetched by scalar loads and fed to the FMAs as SGPR operands;
//   - the LSTM epilogue is the one of conv_igemm.hip (same sigmoid / tanh / fmaf association), per lane, vector stores.
// Run by nint_cell_fwd for nint_layer.tile_rows == 1 on layers with Ch <= 8, k = 3 and a thin input (nint_stencil_holds).
// BPTT of such layers stays on the MFMA kernels.
#include <type_traits>
#include "nint_common.h"

#ifndef NINT_ST_FENCE
#define NINT_ST_FENCE 0       // scalar weight loads fenced per row pair (1: 52 us on the full grid, bf16) or left to the scheduler per channel quad (0: 44 us)
#endif

struct StencilArgs {
  const char* xs; const char* hs;          // halo slabs (hs == nullptr: zero state, model.py:259-262)
  int x_pix, h_pix;                        // bytes per slab pixel
  long x_img, h_img;                       // bytes per slab image
  int xq, hq;                              // channel quads read per pixel (x: ceil(k*Cx / 4) folded, ceil(Cx / 4) plain)
  int xfold;
  const float* Ws;                         // stencil weight image: rows of 32 f32 in iteration order
  const float* bias;                       // gate-stash order [4*Ch16]
  const float* c_prev; float* c_out; char* h_out; char* gates_out;
  int Ch, Chp, Ch16;
  int H, W, P, Hh, Wh, tiles_x, tiles_y;
};

constexpr int ST_ROWS = 8, ST_COLS = 32, ST_HW = ST_COLS + 2, ST_HH = ST_ROWS + 2;

// rows of the stencil weight image (shared with the packer in pointwise.hip through nint_common.h)
template <int DT>
__global__ __launch_bounds__(256) void stencil_lstm_kernel(StencilArgs a_) {
  const StencilArgs& a0 = a_;
  typedef Elem<DT> E;
  constexpr int QB = 4 * E::ES;                       // bytes of a channel quad
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx = t % a0.tiles_x; t /= a0.tiles_x;
  const int ty = t % a0.tiles_y;
  const int img = t / a0.tiles_y;
  const int y0 = ty * ST_ROWS, x0 = tx * ST_COLS;
  // ---- stage the halo tile of both sources: [halo pixel][quads] per source, quads of the real channels only
  const int xrow = a0.xq * QB, hrow = a0.hq * QB;       // LDS bytes per halo pixel
  char* lx = smem;
  char* lh = smem + ST_HH * ST_HW * xrow;
  {
    const char* gx = a0.xs + (long)img * a0.x_img + ((long)(y0 + a0.P - 1) * a0.Wh + (x0 + a0.P - 1)) * a0.x_pix;
    const int nx = ST_HH * ST_HW * a0.xq;
    for (int u = tid; u < nx; u += 256) {
      const int hp = u / a0.xq, q = u - hp * a0.xq;
      const int hy = hp / ST_HW, hx = hp - hy * ST_HW;
      const char* src = gx + ((long)hy * a0.Wh + hx) * a0.x_pix + q * QB;
      if constexpr (DT == NINT_BF16) *(u32x2_t*)(lx + u * QB) = *(const u32x2_t*)src;
      else *(u32x4_t*)(lx + u * QB) = *(const u32x4_t*)src;
    }
    if (a0.hs) {
      const char* gh = a0.hs + (long)img * a0.h_img + ((long)(y0 + a0.P - 1) * a0.Wh + (x0 + a0.P - 1)) * a0.h_pix;
      const int nh = ST_HH * ST_HW * a0.hq;
      for (int u = tid; u < nh; u += 256) {
        const int hp = u / a0.hq, q = u - hp * a0.hq;
        const int hy = hp / ST_HW, hx = hp - hy * ST_HW;
        const char* src = gh + ((long)hy * a0.Wh + hx) * a0.h_pix + q * QB;
        if constexpr (DT == NINT_BF16) *(u32x2_t*)(lh + u * QB) = *(const u32x2_t*)src;
        else *(u32x4_t*)(lh + u * QB) = *(const u32x4_t*)src;
      }
    }
  }
  __syncthreads();

  // lane -> pixel: wave (wy, wx) owns rows 4*wy .. 4*wy+3, columns 16*wx .. 16*wx+15; lane = 16*row + column
  const int col = lane & 15, row = lane >> 4;
  const int py = (wave >> 1) * 4 + row, px = (wave & 1) * 16 + col;       // inside the tile
  float acc[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) acc[o] = a0.bias[(o >> 3) * 16 + (o & 7)];   // o = gate*8 + ch; bias_p is [cblock 0][gate][16]

  typedef typename std::conditional<DT == NINT_BF16, u32x2_t, u32x4_t>::type quad_t;
  auto widen = [](quad_t v) __attribute__((always_inline)) {
    if constexpr (DT == NINT_BF16)
      return (f32x4_t){__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000u),
                       __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000u)};
    else
      return __builtin_bit_cast(f32x4_t, v);
  };
  const float* wrow = a0.Ws;                           // wave-uniform: advances by 32 floats per (tap, channel)
  auto fma_quad = [&](f32x4_t v) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = 0; o < 32; ++o) acc[o] = fmaf(wrow[e * 32 + o], v[e], acc[o]);
      // (two rows = 64 scalar weights in flight; without the fence the scheduler hoists all 128 loads of the quad and
      // spills SGPRs into VGPR lanes)
#if NINT_ST_FENCE
      if (e & 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    wrow += 128;
  };
  // neighbour of a packed channel quad inside the 16-lane row; the edge lane keeps `edge` (bound_ctrl off: old value)
  auto shuffle = [&](quad_t v, quad_t edge, bool right) __attribute__((always_inline)) {
    quad_t r;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(quad_t) / 4); ++i)
      r[i] = right ? (unsigned)__builtin_amdgcn_update_dpp((int)edge[i], (int)v[i], 0x101, 0xf, 0xf, false)    // row_shl:1: lane i <- lane i+1
                   : (unsigned)__builtin_amdgcn_update_dpp((int)edge[i], (int)v[i], 0x111, 0xf, 0xf, false);   // row_shr:1: lane i <- lane i-1
    return r;
  };
  auto source = [&](const char* img_lds, int rowb, int nq, bool folded, int ky) __attribute__((always_inline)) {
    const char* pc = img_lds + ((py + ky) * ST_HW + (px + 1)) * rowb;        // the lane's own column, halo row py + ky
    for (int q = 0; q < nq; ++q) {
      const quad_t c = *(const quad_t*)(pc + q * QB);
      if (folded) {                                   // horizontal taps live in the channels (kx*Cx + c)
        fma_quad(widen(c));
      } else {
        // what the row's edge lanes need: the halo columns.  Branch-free (every lane reads; interior lanes re-read their own
        // quad, which the shuffle overwrites): a divergent branch here makes the compiler park the weight rows in flight
        const quad_t el = *(const quad_t*)(pc + (col == 0 ? -rowb : 0) + q * QB);
        const quad_t er = *(const quad_t*)(pc + (col == 15 ? rowb : 0) + q * QB);
        fma_quad(widen(shuffle(c, el, false)));       // kx = 0: pixel x - 1
        fma_quad(widen(c));                           // kx = 1
        fma_quad(widen(shuffle(c, er, true)));        // kx = 2: pixel x + 1
      }
    }
  };
#pragma unroll 1
  for (int ky = 0; ky < 3; ++ky) {
    source(lx, xrow, a0.xq, a0.xfold != 0, ky);
    if (a0.hs) source(lh, hrow, a0.hq, false, ky);
    else wrow += 3 * a0.hq * 128;
  }

  // ---- LSTM epilogue (model.py:223-229), this lane's pixel.  Its arguments are read from the kernarg segment HERE, through a
  // pointer the compiler cannot see through: fetched at kernel entry they would sit in ~20 SGPRs across the main loop, which
  // wants them for weight rows (SGPRs spill into VGPR lanes, one v_readlane per use).
  const StencilArgs* kp = (const StencilArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  const StencilArgs& a = *kp;
  const int y = y0 + py, x = x0 + px;
  if (y >= a.H || x >= a.W) return;
  const size_t pix = ((size_t)img * a.H + y) * a.W + x;
  const int Ch = a.Ch;
  float* co = a.c_out + pix * a.Chp;
  char* ho = a.h_out + ((((size_t)img * a.Hh) + (y + a.P)) * a.Wh + (x + a.P)) * a.Chp * E::ES;
  char* gs = a.gates_out ? a.gates_out + pix * 4 * a.Ch16 * E::ES : nullptr;        // column (cblock 0 * 4 + gate) * 16 + ch
#pragma unroll
  for (int c4 = 0; c4 < 8; c4 += 4) {
    // (Ch < 8: a quad's padding channels have zero weights and zero bias: gates 0.5 / 0.5 / 0 / 0.5, c = h = 0 -- what the MFMA
    // path stores there too; a quad with no real channel only gets its stash columns written)
    f32x4_t cp = {0.f, 0.f, 0.f, 0.f};
    if (a.c_prev && c4 < Ch) cp = *(const f32x4_t*)(a.c_prev + pix * a.Chp + c4);
    f32x4_t gi, gf, gg, go, cn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gi[e] = sigmoidf_(acc[c4 + e]);
      gf[e] = sigmoidf_(acc[8 + c4 + e]);
      gg[e] = tanhf_(acc[16 + c4 + e]);
      go[e] = sigmoidf_(acc[24 + c4 + e]);
      cn[e] = fmaf(cp[e], gf[e], gi[e] * gg[e]);       // model.py:228 (same association as conv_igemm.hip)
      hn[e] = go[e] * tanhf_(cn[e]);                   // model.py:229
    }
    if (c4 < Ch) {
      *(f32x4_t*)(co + c4) = cn;
      store_vec4<DT>(ho, c4, hn);
    }
    if (gs) {
      store_vec4<DT>(gs, 0 + c4, gi);
      store_vec4<DT>(gs, 16 + c4, gf);
      store_vec4<DT>(gs, 32 + c4, gg);
      store_vec4<DT>(gs, 48 + c4, go);
      // columns 8 .. 15 of every gate block: the values of a channel with zero weights, so that the backward kernels read
      // finite numbers there (the stash is not pre-initialised)
      store_vec4<DT>(gs, 0 + 8 + c4, (f32x4_t){0.5f, 0.5f, 0.5f, 0.5f});
      store_vec4<DT>(gs, 16 + 8 + c4, (f32x4_t){0.5f, 0.5f, 0.5f, 0.5f});
      store_vec4<DT>(gs, 32 + 8 + c4, (f32x4_t){0.f, 0.f, 0.f, 0.f});
      store_vec4<DT>(gs, 48 + 8 + c4, (f32x4_t){0.5f, 0.5f, 0.5f, 0.5f});
    }
  }
}

// host side -----------------------------------------------------------------------------------------------------------
bool nint_internal_stencil_holds(const nint_layer* ly) {
  return ly && nint_stencil_shape(ly->Cx, ly->Ch, ly->k, ly->xfold);
}

extern "C" int nint_stencil_holds(const nint_layer* ly) { return nint_internal_stencil_holds(ly) ? 1 : 0; }

int nint_internal_stencil_lstm(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* x_slab,
                               const void* h_prev, const float* c_prev, void* h_out, float* c_out, void* gates_out,
                               void* stream) {
  if (!nint_internal_stencil_holds(ly) || g->P < 1) return NINT_E_SHAPE;
  const int es = dtype == NINT_BF16 ? 2 : 4;
  StencilArgs a = {};
  a.xs = (const char*)x_slab; a.hs = (const char*)h_prev;
  a.x_pix = ly->Cxp * es; a.h_pix = ly->Chp * es;
  a.x_img = (long)g->Hh * g->Wh * a.x_pix; a.h_img = (long)g->Hh * g->Wh * a.h_pix;
  a.xfold = ly->xfold;
  a.xq = nint_cdiv(ly->xfold ? 3 * ly->Cx : ly->Cx, 4);
  a.hq = nint_cdiv(ly->Ch, 4);
  a.Ws = (const float*)((const char*)ly->Wf + nint_internal_stencil_offset(ly->Cxp, ly->Chp, ly->Ch16, ly->k, dtype));
  a.bias = ly->bias_p;
  a.c_prev = c_prev; a.c_out = c_out; a.h_out = (char*)h_out; a.gates_out = (char*)gates_out;
  a.Ch = ly->Ch; a.Chp = ly->Chp; a.Ch16 = ly->Ch16;
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.tiles_x = nint_cdiv(g->W, ST_COLS); a.tiles_y = nint_cdiv(g->H, ST_ROWS);
  const int qb = 4 * es;
  const size_t lds = (size_t)ST_HH * ST_HW * (a.xq + a.hq) * qb;
  dim3 grid(N * a.tiles_x * a.tiles_y), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == NINT_BF16) hipLaunchKernelGGL(stencil_lstm_kernel<NINT_BF16>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(stencil_lstm_kernel<NINT_F32>, grid, block, lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}
