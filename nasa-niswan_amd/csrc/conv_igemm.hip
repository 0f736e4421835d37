// conv_igemm.hip -- implicit-GEMM convolution on MFMA 16x16 tiles for gfx950.
//
// One kernel template serves both convolutions of the ConvLSTM cell:
//   EPI_LSTM  : gates = W (*) cat[x,h] + b, sigmoid/tanh, c/h update      (reference model.py:219-229)
//   EPI_DGRAD : d cat[x,h] = W^T (*) dG                                    (autograd of model.py:220)
//
// GEMM view: M = pixels (16 consecutive x per MFMA row tile), N = output columns
// (gate channels / cat channels), K = (64-byte channel chunk, tap).  The zero padding of
// nn.Conv2d is physical: sources are halo slabs whose border is kept zero, so no tap is ever
// predicated.  A workgroup owns TH=8 rows x 16 columns of pixels:
//   - its (8+2p) x (16+2p) halo tile is staged in LDS one or more channel chunks at a time,
//     laid out [chunk][halo pixel][64 B] so that every A fragment read is one contiguous
//     1 KiB ds_read_b128 (conflict free), re-used by all k*k taps and all N waves;
//   - weights are pre-packed in MFMA fragment order, streamed L2 -> registers -> LDS in
//     double-buffered groups (one barrier per group), shared by the M waves;
//   - the i,f,g,o tiles of one hidden channel sit in the same lane (column order
//     n' = (cblock*4+gate)*16+col), so the LSTM epilogue needs no cross-lane traffic.
#include "nint_common.h"

enum { EPI_LSTM = 0, EPI_DGRAD = 1 };

template <int DT, int EPI, int WM, int WN, int MT, int NTW>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr int NTH = 64 * WM * WN;
  constexpr int TH = WM * MT;
  constexpr int NTWG = WN * NTW;                       // n-tiles per workgroup
  constexpr int KG = (4 * WM + NTW - 1) / NTW;         // K-steps per weight group (~4 KiB per wave)
  constexpr int BU = (KG * NTWG * 64 + NTH - 1) / NTH; // 16-byte units of a weight group per thread
  constexpr int BG_BYTES = KG * NTWG * 1024;
  static_assert(EPI != EPI_LSTM || NTW % 4 == 0, "LSTM epilogue needs the 4 gate tiles in one wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Abuf = smem;
  char* Bbuf = smem + a.a_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile = blockIdx.x;
  const int tx = tile % a.tiles_x; tile /= a.tiles_x;
  const int ty = tile % a.tiles_y;
  const int img = tile / a.tiles_y;
  const int nt0 = a.nt_begin + blockIdx.y * NTWG;
  const int y0 = ty * TH, x0 = tx * 16;
  const int p = a.p, k = a.k, taps = a.taps;
  const int HWt = 16 + 2 * p;            // halo tile width
  const int NHP = (TH + 2 * p) * HWt;    // halo tile pixels

  f32x4_t acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nchunks = a.nchunk0 + a.nchunk1;
  const char* base0 = a.src0 + (long)img * a.img_stride0 +
                      ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride0;
  const char* base1 = a.src1 ? a.src1 + (long)img * a.img_stride1 +
                                   ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride1
                             : nullptr;
  const int a_lane_off = (lane & 15) * 64 + (lane >> 4) * 16;

  for (int c_begin = 0; c_begin < nchunks; c_begin += a.cpf) {
    const int c_cnt = min(a.cpf, nchunks - c_begin);
    // ---- stage the halo tile for chunks [c_begin, c_begin+c_cnt): global -> regs -> LDS ----
    // (the barrier that ended the previous fill's last group already freed Abuf and Bbuf)
    const int a_units = c_cnt * NHP * 4;
    for (int u = tid; u < a_units; u += NTH) {
      const int q = u & 3;
      const int hpc = u >> 2;
      const int cl = hpc / NHP;
      const int hp = hpc - cl * NHP;
      const int hy = hp / HWt;
      const int hx = hp - hy * HWt;
      const int c = c_begin + cl;
      const char* src = (c < a.nchunk0)
          ? base0 + ((long)hy * a.Wh + hx) * a.pix_stride0 + c * 64 + q * 16
          : base1 + ((long)hy * a.Wh + hx) * a.pix_stride1 + (c - a.nchunk0) * 64 + q * 16;
      *(u32x4_t*)(Abuf + (size_t)u * 16) = *(const u32x4_t*)src;
    }
    const int nsteps = c_cnt * taps;
    const int ngroups = (nsteps + KG - 1) / KG;
    const char* Bsrc = a.Bp + ((size_t)(c_begin * taps) * a.NTt + nt0) * 1024;
    u32x4_t breg[BU];
    // ---- weight group 0 ----
    {
      const int units = min(KG, nsteps) * NTWG * 64;
#pragma unroll
      for (int i = 0; i < BU; ++i) {
        const int u = tid + i * NTH;
        if (u < units) {
          const int ks = u / (NTWG * 64), r = u - ks * (NTWG * 64);
          breg[i] = *(const u32x4_t*)(Bsrc + (size_t)ks * a.NTt * 1024 + r * 16);
        }
      }
#pragma unroll
      for (int i = 0; i < BU; ++i) {
        const int u = tid + i * NTH;
        if (u < units) *(u32x4_t*)(Bbuf + (size_t)u * 16) = breg[i];
      }
    }
    __syncthreads();
    for (int grp = 0; grp < ngroups; ++grp) {
      const int s0 = grp * KG;
      const bool more = grp + 1 < ngroups;
      const int units_next = more ? min(KG, nsteps - (s0 + KG)) * NTWG * 64 : 0;
      if (more) {
#pragma unroll
        for (int i = 0; i < BU; ++i) {
          const int u = tid + i * NTH;
          if (u < units_next) {
            const int ks = u / (NTWG * 64), r = u - ks * (NTWG * 64);
            breg[i] = *(const u32x4_t*)(Bsrc + (size_t)(s0 + KG + ks) * a.NTt * 1024 + r * 16);
          }
        }
      }
      const char* Bcur = Bbuf + (grp & 1) * BG_BYTES;
      const int nks = min(KG, nsteps - s0);
      for (int ks = 0; ks < nks; ++ks) {
        const int sl = s0 + ks;
        const int cl = sl / taps;
        const int tap = sl - cl * taps;
        const int tyy = tap / k;
        const int txx = tap - tyy * k;
        const char* Ab = Abuf + ((size_t)(cl * NHP + (wm * MT + tyy) * HWt + txx)) * 64 + a_lane_off;
        const char* Bb = Bcur + (size_t)(ks * NTWG + wn * NTW) * 1024 + lane * 16;
        u32x4_t af[MT], bf[NTW];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *(const u32x4_t*)(Ab + (size_t)i * HWt * 64);
#pragma unroll
        for (int j = 0; j < NTW; ++j) bf[j] = *(const u32x4_t*)(Bb + j * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<DT>(af[i], bf[j], acc[i][j]);
      }
      if (more) {
        char* Bnext = Bbuf + ((grp + 1) & 1) * BG_BYTES;
#pragma unroll
        for (int i = 0; i < BU; ++i) {
          const int u = tid + i * NTH;
          if (u < units_next) *(u32x4_t*)(Bnext + (size_t)u * 16) = breg[i];
        }
      }
      __syncthreads();
    }
  }

  // ------------------------------------------------------------------ epilogue
  // C/D map of the 16x16 MFMA: column = lane&15, row = 4*(lane>>4) + reg.
  const int col = lane & 15;
  const int rbase = 4 * (lane >> 4);
  if constexpr (EPI == EPI_LSTM) {
    typedef Elem<DT> E;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int y = y0 + wm * MT + i;
#pragma unroll
      for (int cb = 0; cb < NTW / 4; ++cb) {
        const int cblock = (nt0 + wn * NTW) / 4 + cb;
        const int ch = cblock * 16 + col;
        const float bi = a.bias[(cblock * 4 + 0) * 16 + col];
        const float bf_ = a.bias[(cblock * 4 + 1) * 16 + col];
        const float bg = a.bias[(cblock * 4 + 2) * 16 + col];
        const float bo = a.bias[(cblock * 4 + 3) * 16 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = x0 + rbase + r;
          if (y < a.H && x < a.W) {
            const float gi = sigmoidf_(acc[i][cb * 4 + 0][r] + bi);
            const float gf = sigmoidf_(acc[i][cb * 4 + 1][r] + bf_);
            const float gg = tanhf_(acc[i][cb * 4 + 2][r] + bg);
            const float go = sigmoidf_(acc[i][cb * 4 + 3][r] + bo);
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
            const float cp = a.c_prev ? a.c_prev[pix * a.Chp + ch] : 0.f;
            const float cn = cp * gf + gi * gg;          // model.py:228
            const float hn = go * tanhf_(cn);            // model.py:229
            a.c_out[pix * a.Chp + ch] = cn;
            const size_t hpix = ((size_t)img * a.Hh + (y + a.P)) * a.Wh + (x + a.P);
            store_elem<DT>(a.h_out, hpix * a.Chp + ch, hn);
            if (a.gates_out) {
              const size_t gb = pix * (size_t)(4 * a.Ch16) + (size_t)cblock * 64 + col;
              store_elem<DT>(a.gates_out, gb + 0, gi);
              store_elem<DT>(a.gates_out, gb + 16, gf);
              store_elem<DT>(a.gates_out, gb + 32, gg);
              store_elem<DT>(a.gates_out, gb + 48, go);
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int y = y0 + wm * MT + i;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = (nt0 + wn * NTW + j) * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = x0 + rbase + r;
          if (y < a.H && x < a.W) {
            const size_t pix = ((size_t)img * a.H + y) * a.W + x;
            const float v = acc[i][j][r];
            if (n < a.C0p) {
              if (a.out0) a.out0[pix * a.C0p + n] += v;
            } else if (a.out1) {
              a.out1[pix * a.C1p + (n - a.C0p)] = v;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ host side
template <int DT, int EPI, int WM, int WN, int MT, int NTW>
static int launch_cfg(ConvArgs& a, int N, int ngroups_y, hipStream_t st) {
  constexpr int KG = (4 * WM + NTW - 1) / NTW;
  constexpr int NTWG = WN * NTW;
  constexpr int TH = WM * MT;
  static_assert(TH == 8, "slab slack rows assume 8-row tiles");
  const int NHP = (TH + 2 * a.p) * (16 + 2 * a.p);
  const int nchunks = a.nchunk0 + a.nchunk1;
  const int b_bytes = 2 * KG * NTWG * 1024;
  // as many channel chunks per fill as fit in ~64 KiB (two workgroups per CU stay resident)
  int cpf = (64 * 1024 - b_bytes) / (NHP * 64);
  if (cpf < 1) cpf = 1;
  if (cpf > nchunks) cpf = nchunks;
  a.cpf = cpf;
  a.a_bytes = cpf * NHP * 64;
  const size_t lds = (size_t)a.a_bytes + b_bytes;
  if (lds > 160 * 1024) return NINT_E_LDS;
  a.tiles_x = nint_cdiv(a.W, 16);
  a.tiles_y = nint_cdiv(a.H, TH);
  auto kern = conv_igemm_kernel<DT, EPI, WM, WN, MT, NTW>;
  if (lds > 64 * 1024) {
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  dim3 grid(a.tiles_x * a.tiles_y * N, ngroups_y), block(64 * WM * WN);
  hipLaunchKernelGGL(kern, grid, block, lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

template <int DT, int EPI>
static int launch_conv(ConvArgs& a, int N, int ntiles, hipStream_t st) {
  if (ntiles <= 0) return NINT_OK;
  if constexpr (EPI == EPI_LSTM) {
    if (ntiles % 4) return NINT_E_SHAPE;
    const int cbs = ntiles / 4;
    if (cbs % 4 == 0) return launch_cfg<DT, EPI, 1, 4, 8, 4>(a, N, cbs / 4, st);
    if (cbs % 2 == 0) return launch_cfg<DT, EPI, 2, 2, 4, 4>(a, N, cbs / 2, st);
    return launch_cfg<DT, EPI, 4, 1, 2, 4>(a, N, cbs, st);
  } else {
    // pick (WN, NTW) with WN*NTW dividing the tile count, widest first
    if (ntiles % 16 == 0) return launch_cfg<DT, EPI, 1, 4, 8, 4>(a, N, ntiles / 16, st);
    if (ntiles % 12 == 0) return launch_cfg<DT, EPI, 1, 4, 8, 3>(a, N, ntiles / 12, st);
    if (ntiles % 8 == 0) return launch_cfg<DT, EPI, 2, 2, 4, 4>(a, N, ntiles / 8, st);
    if (ntiles % 6 == 0) return launch_cfg<DT, EPI, 2, 2, 4, 3>(a, N, ntiles / 6, st);
    if (ntiles % 4 == 0) return launch_cfg<DT, EPI, 4, 1, 2, 4>(a, N, ntiles / 4, st);
    if (ntiles % 3 == 0) return launch_cfg<DT, EPI, 4, 1, 2, 3>(a, N, ntiles / 3, st);
    if (ntiles % 2 == 0) return launch_cfg<DT, EPI, 4, 1, 2, 2>(a, N, ntiles / 2, st);
    return launch_cfg<DT, EPI, 4, 1, 2, 1>(a, N, ntiles, st);
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int nint_cell_fwd(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                             const void* x_slab, const void* h_prev, const float* c_prev,
                             void* h_out, float* c_out, void* gates_out, void* stream) {
  if (!ly || !g || !x_slab || !h_out || !c_out || N <= 0) return NINT_E_ARG;
  if (dtype != NINT_F32 && dtype != NINT_BF16) return NINT_E_ARG;
  if (!(ly->k & 1) || ly->k / 2 > g->P) return NINT_E_ARG;
  if (!aligned16(x_slab) || !aligned16(h_prev) || !aligned16(ly->Wf) || !aligned16(h_out)) return NINT_E_ALIGN;
  const int es = dtype == NINT_BF16 ? 2 : 4, kc = dtype == NINT_BF16 ? 32 : 16;
  if (ly->Cxp % kc || ly->Chp % kc || ly->Ch16 % 16 || ly->Chp < ly->Ch16) return NINT_E_ARG;
  ConvArgs a = {};
  a.src0 = (const char*)x_slab;
  a.src1 = (const char*)h_prev;
  a.nchunk0 = ly->Cxp / kc;
  a.nchunk1 = h_prev ? ly->Chp / kc : 0;
  a.pix_stride0 = ly->Cxp * es;
  a.pix_stride1 = ly->Chp * es;
  a.img_stride0 = (long)g->Hh * g->Wh * a.pix_stride0;
  a.img_stride1 = (long)g->Hh * g->Wh * a.pix_stride1;
  a.Bp = (const char*)ly->Wf;
  a.NTt = 4 * ly->Ch16 / 16;
  a.nt_begin = 0;
  a.k = ly->k; a.p = ly->k / 2; a.taps = ly->k * ly->k;
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.bias = ly->bias_p;
  a.c_prev = c_prev; a.c_out = c_out; a.h_out = (char*)h_out; a.gates_out = (char*)gates_out;
  a.Chp = ly->Chp; a.Ch16 = ly->Ch16;
  hipStream_t st = (hipStream_t)stream;
  return dtype == NINT_BF16 ? launch_conv<NINT_BF16, EPI_LSTM>(a, N, a.NTt, st)
                            : launch_conv<NINT_F32, EPI_LSTM>(a, N, a.NTt, st);
}

extern "C" int nint_conv_dgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                               const void* dG, float* dx_accum, float* dh_prev, void* stream) {
  if (!ly || !g || !dG || N <= 0) return NINT_E_ARG;
  if (dtype != NINT_F32 && dtype != NINT_BF16) return NINT_E_ARG;
  if (!dx_accum && !dh_prev) return NINT_OK;
  if (!aligned16(dG) || !aligned16(ly->Wd)) return NINT_E_ALIGN;
  const int es = dtype == NINT_BF16 ? 2 : 4, kc = dtype == NINT_BF16 ? 32 : 16;
  const int Gc = 4 * ly->Ch16;
  ConvArgs a = {};
  a.src0 = (const char*)dG;
  a.src1 = nullptr;
  a.nchunk0 = Gc / kc;
  a.nchunk1 = 0;
  a.pix_stride0 = Gc * es;
  a.img_stride0 = (long)g->Hh * g->Wh * a.pix_stride0;
  a.Bp = (const char*)ly->Wd;
  a.NTt = (ly->Cxp + ly->Chp) / 16;
  a.k = ly->k; a.p = ly->k / 2; a.taps = ly->k * ly->k;
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.out0 = dx_accum; a.out1 = dh_prev;
  a.C0p = ly->Cxp; a.C1p = ly->Chp;
  // only the n-tiles whose destination exists are computed
  const int nt_x = ly->Cxp / 16, nt_h = ly->Chp / 16;
  a.nt_begin = dx_accum ? 0 : nt_x;
  const int ntiles = (dx_accum ? nt_x : 0) + (dh_prev ? nt_h : 0);
  hipStream_t st = (hipStream_t)stream;
  return dtype == NINT_BF16 ? launch_conv<NINT_BF16, EPI_DGRAD>(a, N, ntiles, st)
                            : launch_conv<NINT_F32, EPI_DGRAD>(a, N, ntiles, st);
}
