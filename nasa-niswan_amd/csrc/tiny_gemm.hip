// tiny_gemm.hip -- the ConvLSTM gate step for TINY hidden widths on the matrix pipe with a DENSE K (gfx950).
//
//   gates = W (*) cat[x,h] + b, sigmoid/tanh, c/h update      (reference model.py:216-231)
//
// With Ch <= 8 the implicit GEMM of conv_igemm.hip is mostly padding: BASELINE configs[0] (4 input channels, 8 hidden, 3x3)
// runs 12 K-steps of 32 channels for 108 real ones and 4 column tiles for 32 real gate columns (14 % useful MACs).  The
// stencil kernel (stencil.hip) removes the padding but has to broadcast its weights through the scalar path, a latency chain
// that loses to the padded tiles.  This kernel keeps the matrix pipe -- the chip's broadcast engine -- and makes the
// contraction dense instead:
//   - K is a list of 16-BYTE GROUPS (8 bf16 / 4 f32 channels of ONE halo pixel of ONE source): per vertical tap the folded x
//     source's ceil(3 Cx ES / 16) groups (a plain x source: per tap), per tap the h source's ceil(Ch ES / 16) groups.  A
//     K-step is four groups -- one per 16-lane group of the pixel fragment -- so configs[0] is 15 groups = 4 K-steps (bf16)
//     instead of 12, and the four lane groups of ONE ds_read_b128 gather four different (tap, channel group) pieces of the
//     staged halo tile: per-lane address = group table entry + the lane's pixel offset.  No im2col copy exists.
//   - N is 32 columns = 2 MFMA column tiles in the order row = 4 c + gate, channel = 2 c + tile: with the operands fed
//     swapped (D = [column][pixel]) a lane owns the i, f, g, o of channels 2 g' and 2 g' + 1 of one pixel: the LSTM epilogue
//     needs no cross-lane traffic and stores adjacent channels.
//   - the weights (K-steps x 2 tiles x 1 KiB, fragment order, written by nint_pack_weights behind the other images) are
//     loaded ONCE per wave into registers; the K loop is 4-7 steps of 2 x 4 MFMAs with no memory instruction but the
//     fragment reads.
// A workgroup (4 waves) owns 8 rows x 32 pixels (16 row tiles, four per wave).  The kernel is fill + epilogue latency: the
// point is that it does a sixth of the padded kernel's matrix work and reads only the real channels.
// Selected by nint_cell_fwd for nint_layer.tile_rows == 0 on layers nint_stencil_holds() (same shapes as the stencil kernel).
#include "nint_common.h"

struct TinyArgs {
  const char* xs; const char* hs;          // halo slabs (hs == nullptr: zero state: its groups are skipped)
  int x_pix, h_pix;                        // bytes per slab pixel
  long x_img, h_img;
  int xg, hg;                              // 16-byte groups staged per pixel (x: folded or plain channels; h)
  int ngx, ngroups;                        // K groups: [0, ngx) read the x image, [ngx, ngroups) the h image
  const char* Wt;                          // dense-K weight fragments [K-step][2][64 lanes][16 B] + the group table behind them
  const int* table;                        // ngroups_padded ints: byte offset of group gi inside its source's LDS image (tap + channel part)
  const float* bias;
  const float* c_prev; float* c_out; char* h_out; char* gates_out;
  int Ch, Chp, Ch16;
  int H, W, P, Hh, Wh, tiles_x, tiles_y;
};

constexpr int TG_ROWS = 8, TG_COLS = 32, TG_HW = NINT_TINY_HW, TG_HH = TG_ROWS + 2, TG_MAXSTEPS = NINT_TINY_MAXSTEPS;
static_assert(TG_HW == TG_COLS + 2, "the packer's group table uses this halo width");

// two adjacent channels (even index) in one store: 8 bytes f32 / 4 bytes bf16
template <int DT> __device__ __forceinline__ void store_pair(void* p, size_t i, float v0, float v1) {
  if constexpr (DT == NINT_BF16) *(uint32_t*)((uint16_t*)p + i) = pack_bf16x2(v0, v1);
  else *(float2*)((float*)p + i) = make_float2(v0, v1);
}

template <int DT>
__global__ __launch_bounds__(256) void tiny_lstm_kernel(TinyArgs a) {
  typedef Elem<DT> E;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int img = t / a.tiles_y;
  const int y0 = ty * TG_ROWS, x0 = tx * TG_COLS;
  const bool has_h = a.hs != nullptr;
  const int ng = has_h ? a.ngroups : a.ngx;                  // K groups of this launch (zero state: the x groups only)
  const int nsteps = (ng + 3) / 4;
  // ---- weights of this wave: all K-steps of both column tiles, resident in registers
  u32x4_t wq[TG_MAXSTEPS][2];
#pragma unroll
  for (int s = 0; s < TG_MAXSTEPS; ++s) {
    if (s < nsteps) {
      wq[s][0] = *(const u32x4_t*)(a.Wt + ((size_t)(s * 2 + 0) * 64 + lane) * 16);
      wq[s][1] = *(const u32x4_t*)(a.Wt + ((size_t)(s * 2 + 1) * 64 + lane) * 16);
    }
  }
  // ---- stage the halo tile of both sources: [halo pixel][groups of 16 B], the real channels only; one zero group at the end
  const int xrow = a.xg * 16, hrow = a.hg * 16;
  char* lx = smem;
  char* lh = smem + TG_HH * TG_HW * xrow;
  char* lz = lh + TG_HH * TG_HW * hrow;                      // 16 zero bytes: where padding groups point
  {
    const char* gx = a.xs + (long)img * a.x_img + ((long)(y0 + a.P - 1) * a.Wh + (x0 + a.P - 1)) * a.x_pix;
    const int nx = TG_HH * TG_HW * a.xg;
    for (int u = tid; u < nx; u += 256) {
      const int hp = u / a.xg, q = u - hp * a.xg;
      const int hy = hp / TG_HW, hx = hp - hy * TG_HW;
      *(u32x4_t*)(lx + u * 16) = *(const u32x4_t*)(gx + ((long)hy * a.Wh + hx) * a.x_pix + q * 16);
    }
    if (has_h) {
      const char* gh = a.hs + (long)img * a.h_img + ((long)(y0 + a.P - 1) * a.Wh + (x0 + a.P - 1)) * a.h_pix;
      const int nh = TG_HH * TG_HW * a.hg;
      for (int u = tid; u < nh; u += 256) {
        const int hp = u / a.hg, q = u - hp * a.hg;
        const int hy = hp / TG_HW, hx = hp - hy * TG_HW;
        *(u32x4_t*)(lh + u * 16) = *(const u32x4_t*)(gh + ((long)hy * a.Wh + hx) * a.h_pix + q * 16);
      }
    }
    if (tid < 4) ((unsigned*)lz)[tid] = 0u;
  }
  // ---- this lane's group of every K-step: LDS byte offset of (tap, channel group) relative to the lane's pixel
  const int g = lane >> 4, pxl = lane & 15;
  int goff[TG_MAXSTEPS];       // offset into smem of the group for halo pixel (0, 0) of its source image; < 0: padding group
  int gstr[TG_MAXSTEPS];       // bytes per halo pixel of that group's source
#pragma unroll
  for (int s = 0; s < TG_MAXSTEPS; ++s) {
    const int gi = 4 * s + g;
    int off = -1, str = 0;
    if (s < nsteps && gi < ng) {
      const bool isx = gi < a.ngx;
      off = a.table[gi] + (isx ? 0 : (int)(lh - smem));
      str = isx ? xrow : hrow;
    }
    goff[s] = off; gstr[s] = str;
  }
  __syncthreads();

  // row tile r of this wave: r = 0..3 -> tile row 2 * wave + (r >> 1), columns 16 * (r & 1) .. + 15
  f32x4_t acc[4][2];
  {
    // accumulators start at the gate bias: register j of column tile t is gate j of channel 2 g + t (column order of this kernel)
    f32x4_t b0, b1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      b0[j] = a.bias[j * 16 + 2 * g];                          // bias_p is [cblock 0][gate][16 channels]
      b1[j] = a.bias[j * 16 + 2 * g + 1];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[r][0] = b0; acc[r][1] = b1; }
  }
  int pixoff[4];               // halo-pixel index of this lane's pixel for row tile r (tap (0,0) = the pixel itself; the table adds the tap)
#pragma unroll
  for (int r = 0; r < 4; ++r) pixoff[r] = (2 * wave + (r >> 1)) * TG_HW + 16 * (r & 1) + pxl;
#pragma unroll
  for (int s = 0; s < TG_MAXSTEPS; ++s) {
    if (s < nsteps) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* ad = goff[s] >= 0 ? smem + goff[s] + pixoff[r] * gstr[s] : lz;
        const u32x4_t px = *(const u32x4_t*)ad;
        acc[r][0] = mma_step<DT>(wq[s][0], px, acc[r][0]);     // swapped operands: D[column][pixel]
        acc[r][1] = mma_step<DT>(wq[s][1], px, acc[r][1]);
      }
    }
  }

  // ---- LSTM epilogue (model.py:223-229): this lane holds i, f, g, o of channels 2 g and 2 g + 1 of pixel pxl of each row tile
  const int Ch = a.Ch, ch0 = 2 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = y0 + 2 * wave + (r >> 1), x = x0 + 16 * (r & 1) + pxl;
    if (y >= a.H || x >= a.W) continue;
    const size_t pix = ((size_t)img * a.H + y) * a.W + x;
    float cp0 = 0.f, cp1 = 0.f;
    if (a.c_prev && ch0 < Ch) {
      const float2 cpp = *(const float2*)(a.c_prev + pix * a.Chp + ch0);   // (channel ch0 + 1 may be padding: the slab holds a zero there)
      cp0 = cpp.x; cp1 = cpp.y;
    }
    float gi[2], gf[2], gg[2], go[2], cn[2], hn[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      gi[t2] = sigmoidf_(acc[r][t2][0]);
      gf[t2] = sigmoidf_(acc[r][t2][1]);
      gg[t2] = tanhf_(acc[r][t2][2]);
      go[t2] = sigmoidf_(acc[r][t2][3]);
      cn[t2] = fmaf(t2 ? cp1 : cp0, gf[t2], gi[t2] * gg[t2]);   // model.py:228 (same association as conv_igemm.hip)
      hn[t2] = go[t2] * tanhf_(cn[t2]);                        // model.py:229
    }
    if (ch0 < Ch) {                                            // (a half-empty pair: its padding channel is an exact zero)
      *(float2*)(a.c_out + pix * a.Chp + ch0) = make_float2(cn[0], cn[1]);
      char* ho = a.h_out + ((((size_t)img * a.Hh) + (y + a.P)) * a.Wh + (x + a.P)) * a.Chp * E::ES;
      store_pair<DT>(ho, ch0, hn[0], hn[1]);
    }
    if (a.gates_out) {
      char* gs = a.gates_out + pix * 4 * a.Ch16 * E::ES;       // column (cblock 0 * 4 + gate) * 16 + ch
      store_pair<DT>(gs, 0 + ch0, gi[0], gi[1]);
      store_pair<DT>(gs, 16 + ch0, gf[0], gf[1]);
      store_pair<DT>(gs, 32 + ch0, gg[0], gg[1]);
      store_pair<DT>(gs, 48 + ch0, go[0], go[1]);
      // columns 8 .. 15 of every gate block: the values of a channel with zero weights (the stash is not pre-initialised)
      store_pair<DT>(gs, 0 + 8 + ch0, 0.5f, 0.5f);
      store_pair<DT>(gs, 16 + 8 + ch0, 0.5f, 0.5f);
      store_pair<DT>(gs, 32 + 8 + ch0, 0.f, 0.f);
      store_pair<DT>(gs, 48 + 8 + ch0, 0.5f, 0.5f);
    }
  }
}

// host side -----------------------------------------------------------------------------------------------------------
int nint_internal_tiny_lstm(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* x_slab,
                            const void* h_prev, const float* c_prev, void* h_out, float* c_out, void* gates_out,
                            void* stream) {
  if (!ly || !nint_tiny_shape(ly->Cx, ly->Ch, ly->k, ly->xfold, dtype) || g->P < 1) return NINT_E_SHAPE;
  const int es = dtype == NINT_BF16 ? 2 : 4;
  TinyArgs a = {};
  a.xs = (const char*)x_slab; a.hs = (const char*)h_prev;
  a.x_pix = ly->Cxp * es; a.h_pix = ly->Chp * es;
  a.x_img = (long)g->Hh * g->Wh * a.x_pix; a.h_img = (long)g->Hh * g->Wh * a.h_pix;
  a.xg = nint_tiny_xg(ly->Cx, ly->xfold, dtype); a.hg = nint_tiny_hg(ly->Ch, dtype);
  a.ngx = nint_tiny_ngx(ly->Cx, ly->xfold, dtype);
  a.ngroups = a.ngx + 9 * a.hg;
  const size_t off = nint_internal_tiny_offset(ly->Cx, ly->Cxp, ly->Ch, ly->Chp, ly->Ch16, ly->k, ly->xfold, dtype);
  a.Wt = (const char*)ly->Wf + off;
  a.table = (const int*)(a.Wt + (size_t)TG_MAXSTEPS * 2 * 1024);
  a.bias = ly->bias_p;
  a.c_prev = c_prev; a.c_out = c_out; a.h_out = (char*)h_out; a.gates_out = (char*)gates_out;
  a.Ch = ly->Ch; a.Chp = ly->Chp; a.Ch16 = ly->Ch16;
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.tiles_x = nint_cdiv(g->W, TG_COLS); a.tiles_y = nint_cdiv(g->H, TG_ROWS);
  const size_t lds = (size_t)TG_HH * TG_HW * (a.xg + a.hg) * 16 + 16;
  dim3 grid(N * a.tiles_x * a.tiles_y), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == NINT_BF16) hipLaunchKernelGGL(tiny_lstm_kernel<NINT_BF16>, grid, block, lds, st, a);
  else hipLaunchKernelGGL(tiny_lstm_kernel<NINT_F32>, grid, block, lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}
