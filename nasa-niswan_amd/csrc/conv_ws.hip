// conv_ws.hip -- weight-STATIONARY persistent gate kernel for the narrow ConvLSTM layers (bf16, gfx950).
//
// EXPERIMENT BUILD ONLY (build.py --exp, tools/kbench.py --exp --dbg 0x1000): parity-green (bit-identical to the streaming
// kernel through the whole model suite), measured, NOT faster, not shipped -- layer 2 23.1 against 22.6 us, layer 1 53.0
// against 44.4 us.  Ablations: with one or two persistent workgroups per CU the K loop, the K-slice
// exchange, the epilogue and its stores run strictly one after the other (layer 2: 5.8 + 2.6 + 5.8 us on a 12.6 us
// skeleton), whereas the streaming kernel's four workgroups per CU overlap each other's phases.  What it would need is a
// second wave group per workgroup that runs the previous tile's exchange / epilogue under this tile's K loop.
//
// The streaming kernel of conv_igemm.hip re-reads a layer's packed weights once per pixel tile: for the narrow layers
// (hidden 16 / 32: 72 / 216 KiB of gate weights, 18 / 27 K-steps) a 64-pixel tile is 2000 workgroups per launch whose
// phases -- halo fill (HBM burst), K loop (weights through L1, MFMA), epilogue (HBM burst) -- run in lockstep and add up:
// 47 us for 22 us of matrix work.  Here the weights never move:
//   - a workgroup is PERSISTENT (one or two per CU) and walks a contiguous range of 4 x 16-pixel tiles;
//   - its waves split the gate columns (WN groups of one hidden-channel block = 4 column tiles) and the K-steps (WK
//     slices); a wave's K-slice of the weights -- at most SL K-steps x 4 column tiles = SL * 16 VGPRs -- is loaded ONCE
//     and stays in registers for every tile;
//   - so the K loop has no vector-memory traffic at all, and the in-order vmcnt is free for the NEXT tile: its halo
//     image is LDS-DMA'd into the second buffer and its c_{t-1} vectors are fetched while this tile computes;
//   - K-slices are summed through LDS and every wave runs the fused LSTM epilogue of its row (same arithmetic, same
//     order as the streaming kernel: results are bit-identical).
// Same data layout, same packed weights, same epilogue stores as conv_igemm.hip (reference model.py:219-229).
#include "nint_common.h"

namespace {

constexpr int MT = 4;          // rows of a pixel tile
constexpr int NTW = 4;         // column tiles per wave: the i, f, g, o tiles of one 16-channel block

template <int WN, int WK, int SL>
__global__ __launch_bounds__(64 * WN * WK, WN * WK <= 4 ? 2 : 1) void conv_ws_lstm_kernel(ConvArgs a, int ntiles) {
  constexpr int NW = WN * WK, NTH = 64 * NW;
  constexpr int Q = MT / WK;                    // rows whose epilogue a wave runs
  constexpr int FR = MT - Q;                    // rows it hands to the other K-slices
  constexpr int XR = 1, NTX = NTW / XR;         // exchange rounds, column tiles per round
  constexpr int NPMAX = 8;                      // DMA pieces (64 x 16 bytes) of a halo image per wave, at most
  static_assert(MT % WK == 0 && WK >= 2, "K-slice waves split the tile rows");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wk = wave / WN;
  const int nt0 = wn * NTW;
  const int p = a.p, k = a.k, taps = a.taps;
  const int HWt = 16 + 2 * p, NHP = (MT + 2 * p) * HWt, NHPp = a.nhp_pad;
  const int chunk_bytes = 4 * NHPp * 16;
  const int nchunks = a.nchunk0 + a.nchunk1;
  const int abuf_bytes = nchunks * chunk_bytes;
  char* const xbuf = smem + 2 * abuf_bytes;

  // ---- this workgroup's tiles: XCD x (= blockIdx % 8, each with its own L2) takes a contiguous tile range, dealt
  // round-robin to the XCD's workgroups
  int t_first, t_step, t_end;
  {
    const int nb = gridDim.x, b = blockIdx.x, x = b % 8, i = b / 8;
    const int q8 = ntiles / 8, r8 = ntiles % 8;
    const int lo = x * q8 + (x < r8 ? x : r8), len = q8 + (x < r8 ? 1 : 0);
    t_step = (nb - x + 7) / 8;                  // workgroups of this XCD
    t_first = lo + i;
    t_end = lo + len;
  }
  if (t_first >= t_end) return;                 // (whole workgroup: no barrier is skipped by part of it)

  // ---- this wave's K-slice: steps [s_lo, s_lo + cnt) of S = nchunks * taps, weights resident in registers
  const int S = nchunks * taps;
  const int s_lo = (S * wk) / WK, cnt = (S * (wk + 1)) / WK - s_lo;          // cnt <= SL (host-checked)
  u32x4_t bw[SL][NTW];
  int soff[SL];                                 // LDS offset of the step's A fragment (row 0, lane 0): wave-uniform
  {
    const char* Bw = a.Bp + (size_t)nt0 * 1024 + lane * 16;
    const size_t bstep = (size_t)a.NTt * 1024;
#pragma unroll
    for (int d = 0; d < SL; ++d) {
      const bool on = d < cnt;
      const int s = on ? s_lo + d : s_lo;       // steps past the slice: zero weights on a valid address
      const int cl = s / taps, tap = s - cl * taps;
      const int tyy = tap / k, txx = tap - tyy * k;
      soff[d] = __builtin_amdgcn_readfirstlane(cl * chunk_bytes + (tyy * HWt + txx) * 16);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        bw[d][j] = (u32x4_t){0u, 0u, 0u, 0u};
        if (on) bw[d][j] = *(const u32x4_t*)(Bw + (size_t)s * bstep + j * 1024);
      }
    }
  }
  // per-lane A addresses of the wave's rows: local row i is tile row (i + wk*Q) % MT, so the rows a wave owns after the
  // K-slice exchange are its local rows 0..Q-1
  int a_row[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_row[i] = (lane >> 4) * (NHPp * 16) + (lane & 15) * 16 + ((i + wk * Q) % MT) * HWt * 16;
  // bias: accumulators start at the gate bias in K-slice 0 (register r of column tile j = channel 4*(lane>>4)+r)
  f32x4_t bias[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
    bias[j] = wk == 0 ? *(const f32x4_t*)(a.bias + (nt0 + j) * 16 + 4 * (lane >> 4)) : (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // ---- halo-image staging by LDS-DMA: piece t of the image = 64 consecutive 16-byte units; the per-lane source
  // offset of a piece does not depend on the tile
  const int npieces = nchunks * 4 * NHPp / 64;
  unsigned doff[NPMAX];
#pragma unroll
  for (int i = 0; i < NPMAX; ++i) {
    const int t = wave + i * NW;
    unsigned o = 0;
    if (t < npieces) {
      const int u = t * 64 + lane;
      const int cq = u / NHPp;                  // chunk*4 + q
      int hp = u - cq * NHPp;
      hp = hp < NHP ? hp : 0;                   // pad units re-read pixel 0 (their LDS slots are never used)
      const int q = cq & 3, c = cq >> 2;
      const int hy = hp / HWt, hx = hp - hy * HWt;
      o = c < a.nchunk0 ? (unsigned)((hy * a.Wh + hx) * a.pix_stride0 + c * 64 + q * 16)
                        : (unsigned)((hy * a.Wh + hx) * a.pix_stride1 + (c - a.nchunk0) * 64 + q * 16);
    }
    doff[i] = o;
  }
  const int pieces_per_chunk = 4 * NHPp / 64;
  auto tile_coords = [&](int tile, int& img, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % a.tiles_x;
    const int r = tile / a.tiles_x;
    const int ty = r % a.tiles_y;
    img = r / a.tiles_y;
    y0 = ty * MT; x0 = tx * 16;
  };
  auto issue_fill = [&](int tile, char* buf) __attribute__((always_inline)) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const char* base0 = a.src0 + (long)img * a.img_stride0 + ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride0;
    const char* base1 = a.src1 ? a.src1 + (long)img * a.img_stride1 + ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride1
                               : nullptr;
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
      const int t = wave + i * NW;              // wave-uniform
      if (t < npieces) {
        const char* src = (t / pieces_per_chunk < a.nchunk0 ? base0 : base1) + doff[i];
        // inline asm: invisible to the compiler's vmcnt bookkeeping (it would otherwise drain the queue ahead of every
        // LDS read); the wait is ours -- s_waitcnt vmcnt(0) at the top of the tile loop.  M0 = LDS destination base.
        unsigned keep;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(buf + t * 1024));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
      }
    }
  };
  // c_{t-1} of the rows this wave finishes (rows (i + wk*Q) % MT), one 4-channel vector per row
  auto load_cprev = [&](int tile, f32x4_t* cpv) __attribute__((always_inline)) {
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int y = y0 + (i + wk * Q) % MT, xq = x0 + (lane & 15);
      cpv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (a.c_prev && y < a.H && xq < a.W)
        cpv[i] = *(const f32x4_t*)(a.c_prev + (((size_t)img * a.H + y) * a.W + xq) * a.Chp + wn * 16 + 4 * (lane >> 4));
    }
  };

#ifdef NINT_EXPERIMENT
  const bool abl_nostore = a.dbg & 0x100, abl_noxchg = a.dbg & 0x200, abl_nok = a.dbg & 0x400, abl_nofill = a.dbg & 0x800;
#else
  constexpr bool abl_nostore = false, abl_noxchg = false, abl_nok = false, abl_nofill = false;
#endif
  f32x4_t cp_next[Q];
  issue_fill(t_first, smem);
  load_cprev(t_first, cp_next);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int it = 0;
  for (int tile = t_first; tile < t_end; tile += t_step, ++it) {
    char* const abuf = smem + (it & 1) * abuf_bytes;
    f32x4_t cpv[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) cpv[i] = cp_next[i];
    if (tile + t_step < t_end) {                // the next tile's operands travel under this tile's matrix work
      load_cprev(tile + t_step, cp_next);
      if (!abl_nofill) issue_fill(tile + t_step, smem + ((it + 1) & 1) * abuf_bytes);
    }
    // ---- K loop: resident weights, A fragments from the image; no vector memory
    f32x4_t acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = bias[j];
    if (!abl_nok)
#pragma unroll
    for (int d = 0; d < SL; ++d) {
      u32x4_t af[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *(const u32x4_t*)(abuf + soff[d] + a_row[i]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<NINT_BF16>(bw[d][j], af[i], acc[i][j]);   // swapped: D[channel][pixel]
    }
    // ---- K-slice reduction through LDS (as conv_igemm.hip): park the foreign rows, add the partials of the own rows
    if (!abl_noxchg)
#pragma unroll
    for (int xr = 0; xr < XR; ++xr) {
      if (xr > 0) __syncthreads();              // the previous round's partials are consumed
      char* mine = xbuf + (size_t)((wn * WK + wk) * FR * NTX) * 1024 + lane * 16;
#pragma unroll
      for (int ii = Q; ii < MT; ++ii)
#pragma unroll
        for (int j = 0; j < NTX; ++j) *(f32x4_t*)(mine + ((ii - Q) * NTX + j) * 1024) = acc[ii][xr * NTX + j];
      __syncthreads();
#pragma unroll
      for (int d = 1; d < WK; ++d) {
        const int src = (wk + d) % WK;
        const int shift = ((wk - src + WK) % WK) * Q;
        const char* theirs = xbuf + (size_t)((wn * WK + src) * FR * NTX) * 1024 + lane * 16;
#pragma unroll
        for (int ii = 0; ii < Q; ++ii)
#pragma unroll
          for (int j = 0; j < NTX; ++j)
            acc[ii][xr * NTX + j] += *(const f32x4_t*)(theirs + ((ii + shift - Q) * NTX + j) * 1024);
      }
    }
    // The next tile's image and c_{t-1} have landed (they were issued a K loop ago; the wait does not cover this tile's
    // stores, which are issued below and drain under the next K loop) -- for every wave; this image and the exchange
    // buffer are no longer read.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- LSTM epilogue (model.py:222-229), the streaming kernel's arithmetic and stores
    int img, y0, x0;
    tile_coords(tile, img, y0, x0);
    const int c4 = 4 * (lane >> 4), x = x0 + (lane & 15);
    const int cblock = wn, ch = cblock * 16 + c4;
    const int Gc = 4 * a.Ch16;
    const int odd = (lane >> 4) & 1, chb = (lane >> 5) * 8;
    const unsigned lo_c = (unsigned)(x * a.Chp + ch);
    const unsigned lo_h = (unsigned)((x + a.P) * a.Chp + ch);
    const unsigned lo_g = (unsigned)(x * Gc + cblock * 64) + (unsigned)(chb + (odd ? 32 : 0));
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int y = y0 + (i + wk * Q) % MT;
      const bool ok = y < a.H && x < a.W && !abl_nostore;   // (the lane exchange below needs every lane: no divergent block)
      const size_t rowpix = ((size_t)img * a.H + y) * a.W;
      const f32x4_t cp = cpv[i];
      f32x4_t gi, gf, gg, go, cn, hn;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoidf_(acc[i][0][r]);        // (bias already in the accumulator)
        gf[r] = sigmoidf_(acc[i][1][r]);
        gg[r] = tanhf_(acc[i][2][r]);
        go[r] = sigmoidf_(acc[i][3][r]);
        cn[r] = cp[r] * gf[r] + gi[r] * gg[r];  // model.py:228
        hn[r] = go[r] * tanhf_(cn[r]);          // model.py:229
      }
      if (ok) {
        *(f32x4_t*)(a.c_out + rowpix * a.Chp + lo_c) = cn;
        char* hrow = a.h_out + (((size_t)img * a.Hh + (y + a.P)) * a.Wh) * a.Chp * 2;
        store_vec4<NINT_BF16>(hrow, lo_h, hn);
      }
      if (a.gates_out) {
        // 16-byte gate-stash stores: lane rows 2r / 2r+1 trade halves (see conv_igemm.hip)
        typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
        const u2_t ig0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gi[0], gi[1]), pack_bf16x2(gg[0], gg[1]), false, false);
        const u2_t ig1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gi[2], gi[3]), pack_bf16x2(gg[2], gg[3]), false, false);
        const u2_t fo0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gf[0], gf[1]), pack_bf16x2(go[0], go[1]), false, false);
        const u2_t fo1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gf[2], gf[3]), pack_bf16x2(go[2], go[3]), false, false);
        if (ok) {
          uint16_t* grow = (uint16_t*)a.gates_out + rowpix * Gc;
          *(u32x4_t*)(grow + lo_g) = (u32x4_t){ig0[0], ig1[0], ig0[1], ig1[1]};        // gate i (even row) / g (odd row)
          *(u32x4_t*)(grow + lo_g + 16) = (u32x4_t){fo0[0], fo1[0], fo0[1], fo1[1]};   // gate f / o
        }
      }
    }
  }
}

template <int WN, int WK, int SL>
int launch_ws(ConvArgs& a, int ntiles, int n_cu, size_t lds, hipStream_t st) {
  auto kern = conv_ws_lstm_kernel<WN, WK, SL>;
  if (lds > 64 * 1024)
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = WN * WK <= 4 ? 2 : 1;
  if (per_cu * lds > 160 * 1024) per_cu = 1;
  int grid = n_cu * per_cu;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WN * WK), lds, st, a, ntiles);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

}  // namespace

// Returns NINT_E_SHAPE when the launch is not one this kernel serves (the caller then takes the streaming kernel).
int nint_internal_conv_ws_lstm(ConvArgs& a, int N, void* stream) {
  if (a.kx0 != a.k || a.tile_rows == 8) return NINT_E_SHAPE;
  if (a.NTt != 4 && a.NTt != 8) return NINT_E_SHAPE;                      // hidden 16 / 32
  const int S = (a.nchunk0 + a.nchunk1) * a.taps;
  const int SLn = nint_cdiv(S, 4);
  if (SLn > 7) return NINT_E_SHAPE;
  const int HWt = 16 + 2 * a.p, NHP = (MT + 2 * a.p) * HWt;
  a.nhp_pad = nint_round_up(NHP, 16);
  a.tiles_x = nint_cdiv(a.W, 16);
  a.tiles_y = nint_cdiv(a.H, MT);
  const int ntiles = N * a.tiles_x * a.tiles_y;
  const int WN = a.NTt / 4, WK = 4, NW = WN * WK;
  const int abuf = (a.nchunk0 + a.nchunk1) * 4 * a.nhp_pad * 16;
  if (abuf / 1024 > 8 * NW) return NINT_E_SHAPE;                          // DMA pieces per wave
  const size_t lds = 2 * (size_t)abuf + (size_t)NW * (MT - MT / WK) * NTW * 1024;     // two images + the K-slice exchange
  if (lds > 160 * 1024) return NINT_E_SHAPE;
  int dev = 0, n_cu = 0;
  NINT_CHECK_HIP(hipGetDevice(&dev));
  NINT_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (n_cu <= 0) return NINT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (WN == 1) return SLn <= 5 ? launch_ws<1, 4, 5>(a, ntiles, n_cu, lds, st) : launch_ws<1, 4, 7>(a, ntiles, n_cu, lds, st);
  return SLn <= 5 ? launch_ws<2, 4, 5>(a, ntiles, n_cu, lds, st) : launch_ws<2, 4, 7>(a, ntiles, n_cu, lds, st);
}
