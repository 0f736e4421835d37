// conv_wide.hip -- the gate convolution of the WIDE ConvLSTM layers as a persistent 8-wave implicit GEMM whose
// weight tiles go through LDS once per workgroup (gfx950, bf16).            reference: model.py:219-229
//
// conv_igemm.hip's 4-wave kernel streams every wave's own weight fragments L2 -> L1 -> VGPR: for the reference's layer 0
// (Conv2d(62+64 -> 256, k=5)) that is 1.66 GB of on-chip weight traffic per launch through the texture path (TD busy
// 72 % of the launch, profiles/r02_b_pmc_memory_path_counters.txt), and its 1000 workgroups run their fill / K loop /
// epilogue phases in two chip-wide rounds.  This kernel restructures the same arithmetic:
//
//   * a workgroup = 8 waves = PS pixel slices x CG column groups (PS*CG = 8) owns a 256-PIXEL tile (16 row tiles of 16
//     pixels: R rows x Cb blocks, R*Cb = 16) and 64*CG gate columns; a wave computes 16/PS row tiles x 4 column tiles
//     (one hidden-channel block: its i,f,g,o tiles share lanes, as in conv_igemm.hip).
//   * B: the K-step's weight tile (4*CG KiB, already in MFMA fragment order in global memory) is copied ONCE per
//     workgroup by LDS-DMA (global_load_lds_dwordx4, every wave issues its share) into a D-slot ring, D-1 steps ahead;
//     all PS pixel slices read their fragments from there (ds_read_b128, lane-linear: conflict-free).
//   * A: the halo tile is staged per 64-byte channel CHUNK into an R-slot ring: while the K loop runs the taps of chunk c,
//     the pieces of chunk c+R-1 -- of this tile or of the workgroup's NEXT tile -- arrive by LDS-DMA.  The workgroup is
//     PERSISTENT (one per CU, tiles dealt in XCD-contiguous ranges), so only its very first chunks are waited for; every
//     other fill, and the weights of the next tile's first steps, travel under MFMA work.
//   * synchronisation: raw s_barrier + counted s_waitcnt vmcnt(N), never 0 inside the loop.  All waves run the same
//     program, {P1: issue DMA, read fragments | barrier | P2: MFMAs, counted wait | barrier} per K-step, but waves 4-7
//     (the second wave of every SIMD) run ONE BARRIER BEHIND waves 0-3: while one wave of a SIMD holds the matrix pipe the
//     other one reads LDS and issues DMA (MI355X_MICROARCH.md, two waves per SIMD, item 9).
//   * the LSTM epilogue is conv_igemm.hip's (D = [channel][pixel], 16 / 8-byte vectors, bias in the accumulator init);
//     c_{t-1} is fetched one K-step ahead of it.
//
// Hazards, by barrier count (beta_n = the n-th barrier; group 0 = waves 0-3 runs P1(s) before beta_2s and P2(s) before
// beta_2s+1; group 1 = waves 4-7 one barrier later):
//   RAW weights of step s: read by group 0 after beta_2s-1.  Every wave waits for ITS pieces of step s before that
//       barrier: group 0 in P2(s-1) (N = (D-2) steps' pieces may stay in flight), group 1 in P2(s-2) (N = D-3 steps').
//   WAR weight slot of step s (reused by step s+D): last read by group 1 in P1(s), before beta_2s+1; step s+D is issued
//       in P1(s+1), after beta_2s+1 (group 0) / beta_2s+2 (group 1).
//   RAW chunk g+R-1: its pieces are issued in the first steps of chunk period g, BEFORE the weight pieces of the same
//       step, so they are older than every weight piece that is waited for at the start of period g+R-1 (the host checks
//       that the issue steps end D steps before that period).  WAR: the slot held chunk g-1, whose last read (group 1) is
//       before the first barrier of period g.
// hipcc does not know about the inline-asm DMA (it counts neither their vmcnt nor their LDS writes); its own counted waits
// for the epilogue's loads stay correct because vmcnt retires in order (extra older / younger operations only make a
// compiler-computed wait more conservative).
#include "nint_common.h"

struct WideArgs {
  const char* src0; const char* src1;          // halo slabs: x (or the layer below's h) and h_{t-1} (or nullptr)
  int nchunk0, nchunk1;
  long img_stride0, img_stride1;
  int pix_stride0, pix_stride1;
  const char* Bp; int NTt;                     // packed weights [K-step][NTt n-tiles][64 lanes][16 B]
  int k, p, taps;
  int H, W, P, Hh, Wh;
  // two tile classes: 0 = full tiles (R[0] rows x Cb[0] blocks of 16 pixels), 1 = the leftover strip (R[1] = 0: none)
  int R[2], Cb[2], HWt[2], NHP[2];
  unsigned magic_hwt[2];
  int tiles_x[2], tiles_y0, tiles_img, ntiles;
  int nhpp;                                    // pixels per g-plane of a chunk slot (>= NHP of both classes, multiple of 16)
  unsigned magic_nhpp;
  int npc, pps;                                // 1-KiB pieces per chunk; pieces per wave per K-step while a chunk is being fetched
  int spt;                                     // K-steps per tile
  const float* bias; const float* c_prev; float* c_out; char* h_out; char* gates_out;
  int Chp, Ch16;
};

struct WTile { int img, y0, x0, cls; };

__device__ __forceinline__ WTile wide_tile(const WideArgs& a, int id) {
  WTile t;
  t.img = id / a.tiles_img;
  int r = id - t.img * a.tiles_img;
  const int n0 = a.tiles_x[0] * a.tiles_y0;
  if (r < n0) {
    const int ty = r / a.tiles_x[0], tx = r - ty * a.tiles_x[0];
    t.cls = 0; t.y0 = ty * a.R[0]; t.x0 = tx * 16 * a.Cb[0];
  } else {
    r -= n0;
    t.cls = 1; t.y0 = a.tiles_y0 * a.R[0]; t.x0 = r * 16 * a.Cb[1];
  }
  return t;
}

// one 1-KiB LDS-DMA piece: lane l copies 16 bytes from gsrc (per lane) to lds_dst + 16*l (wave-uniform base in M0)
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N < 0, "add the immediate");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int PS, int D, int R, int KS>
__global__ __launch_bounds__(512, 2) void conv_wide_lstm_kernel(WideArgs a) {
  constexpr int CG = 8 / PS, RPW = 16 / PS, NTW = 4;
  constexpr int k = KS, taps = KS * KS, p = KS / 2;
  constexpr int WT_BYTES = 4 * CG * 1024;            // weight tile of one K-step
  constexpr int PPWB = CG >= 2 ? CG / 2 : 1;         // weight pieces per wave per K-step (CG = 1: waves 0-3 only)
  static_assert(D >= 4, "group 1 waits D-3 steps ahead");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, ps = wave / CG, cg = wave % CG;
  const int plane = a.nhpp * 16, chunk_bytes = 4 * plane;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wring = lds0 + R * chunk_bytes;
  const int nchunks = a.nchunk0 + a.nchunk1, spt = a.spt;

  // ---- this workgroup's tiles.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one L2), so XCD x
  // takes the contiguous tile range x and its workgroups walk it with the stride of their number: tiles that are
  // neighbours in space (shared halo pixels, the same images) are neighbours in time on one L2.  Any bijection is correct.
  int t_first, t_cnt, t_stride;
  {
    const int G = gridDim.x, b = blockIdx.x, x = b % 8, i = b / 8;
    const int nx = G / 8 + (x < G % 8 ? 1 : 0);
    const int q8 = a.ntiles / 8, r8 = a.ntiles % 8;
    const int lo = x * q8 + (x < r8 ? x : r8), sz = q8 + (x < r8 ? 1 : 0);
    t_first = lo + i; t_stride = nx;
    t_cnt = i < sz ? (sz - i + nx - 1) / nx : 0;
  }
  if (t_cnt == 0) return;
  const int S = t_cnt * spt;                          // K-steps of this workgroup's whole run
  const int nt0 = blockIdx.y * 4 * CG + cg * 4;       // first n-tile of this wave

  // ---- DMA issue ------------------------------------------------------------------------------------------------
  // weights: piece j of step ws comes from Bp + ((ws * NTt + blockIdx.y * 4 * CG + j) << 10), lane-linear on both sides
  const char* Bwg = a.Bp + (size_t)blockIdx.y * WT_BYTES + lane * 16;
  const size_t bstep = (size_t)a.NTt * 1024;
  auto issue_weights = [&](int ws, int slot) __attribute__((always_inline)) {
    if (CG == 1 && wave >= 4) return;
#pragma unroll
    for (int i = 0; i < PPWB; ++i) {
      const int j = wave * PPWB + i;
      glds16(Bwg + (size_t)ws * bstep + j * 1024, wring + slot * WT_BYTES + j * 1024);
    }
  };
  // halo chunk: piece q = 64 consecutive 16-byte units of the slot image [g][halo pixel]; unit u -> (g, halo pixel) ->
  // source address.  Pad units (hp >= NHP) and pixels beyond the slab (tiles that overhang the slack of the halo slab;
  // only masked output pixels read them) re-read the tile's first pixel.
  // The chunk being fetched is described by scalars that change once per chunk (fetch descriptor): first byte of the
  // tile's first halo pixel in that chunk, the source's pixel stride, the tile class and the slab's remaining extent.
  const char* f_base; int f_pixs, f_nhp, f_hwt, f_hymax, f_hxmax; unsigned f_magic;
  auto issue_chunk_piece = [&](int slot, int q) __attribute__((always_inline)) {
    const int u = q * 64 + lane;
    const int g = (int)__umulhi((unsigned)u, a.magic_nhpp);
    int hp = u - g * a.nhpp;
    hp = hp < f_nhp ? hp : 0;
    const int hy = (int)__umulhi((unsigned)hp, f_magic);
    const int hx = hp - hy * f_hwt;
    const bool inside = hy < f_hymax && hx < f_hxmax;
    const int off = inside ? (hy * a.Wh + hx) * f_pixs : 0;
    glds16(f_base + off + g * 16, lds0 + slot * chunk_bytes + q * 1024);
  };

  // chunk-fetch cursor: global chunk number d_gc (tile d_j, chunk d_c of it)
  int d_gc = 0, d_c = 0, d_j = 0;
  WTile d_tile = wide_tile(a, t_first);
  auto set_fetch = [&]() __attribute__((always_inline)) {
    const int ay0 = d_tile.y0 + a.P - p, ax0 = d_tile.x0 + a.P - p;
    const long pix = (long)ay0 * a.Wh + ax0;
    if (d_c < a.nchunk0) { f_pixs = a.pix_stride0; f_base = a.src0 + (long)d_tile.img * a.img_stride0 + pix * a.pix_stride0 + d_c * 64; }
    else { f_pixs = a.pix_stride1; f_base = a.src1 + (long)d_tile.img * a.img_stride1 + pix * a.pix_stride1 + (d_c - a.nchunk0) * 64; }
    f_nhp = a.NHP[d_tile.cls]; f_hwt = a.HWt[d_tile.cls]; f_magic = a.magic_hwt[d_tile.cls];
    f_hymax = a.Hh - ay0; f_hxmax = a.Wh - ax0;
  };
  set_fetch();
  auto advance_chunk_cursor = [&]() __attribute__((always_inline)) {
    ++d_gc;
    if (++d_c == nchunks) {
      d_c = 0; ++d_j;
      if (d_j < t_cnt) d_tile = wide_tile(a, t_first + d_j * t_stride);
    }
    if (d_j < t_cnt) set_fetch();
  };

  // ---- prologue: chunks 0 .. R-2 whole, weights of steps 0 .. D-2; the only full drain of the run
  for (int r = 0; r < R - 1 && d_j < t_cnt; ++r) {
    for (int q = wave; q < a.npc; q += 8) issue_chunk_piece(d_gc % R, q);
    advance_chunk_cursor();
  }
  int w_s = 0, w_ws = 0;                              // next weight step to issue: global index, index inside its tile
  for (; w_s < D - 1 && w_s < S; ++w_s) {
    issue_weights(w_ws, w_s % D);
    if (++w_ws == spt) w_ws = 0;
  }
  wait_vm<0>();
  wg_barrier();
  if (grp == 1) wg_barrier();                         // the second wave of every SIMD runs one barrier behind the first

  // ---- compute state
  f32x4_t acc[RPW][NTW];
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const f32x4_t b0 = *(const f32x4_t*)(a.bias + (nt0 + j) * 16 + 4 * (lane >> 4));
#pragma unroll
      for (int i = 0; i < RPW; ++i) acc[i][j] = b0;
    }
  };
  init_acc();
  WTile ct = wide_tile(a, t_first);                   // the tile being computed
  int c_j = 0, c_ws = 0;                              // its number, K-step inside it
  int c_gc = 0, c_c = 0, c_tap = 0, tyy = 0, txx = 0; // chunk (global number, index in tile), tap inside the chunk
  const int a_lane_off = (lane >> 4) * plane + (lane & 15) * 16;
  const unsigned blane = (unsigned)(cg * 4 * 1024 + lane * 16);
  int rowoff[RPW], Rc, HWtc;
  auto set_rowoff = [&]() __attribute__((always_inline)) {
    Rc = a.R[ct.cls]; HWtc = a.HWt[ct.cls];
    const int nrt = Rc * a.Cb[ct.cls];                // row tiles the class really has (narrow grids: fewer than 16)
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rt = ps * RPW + i;
      rowoff[i] = rt < nrt ? ((rt % Rc) * HWtc + 16 * (rt / Rc)) * 16 : 0;   // (idle row tiles read row tile 0; their pixels lie beyond W: never stored)
    }
  };
  set_rowoff();
  f32x4_t cpv[RPW];

  for (int s = 0; s < S; ++s) {
    // ============================== P1(s): DMA issue, fragment reads ==============================
    // pieces of chunk c_gc + R - 1 (cursor d_*) during the first steps of this chunk period, ahead of the weights
    if (d_j < t_cnt) {
      for (int e = 0; e < a.pps; ++e) {
        const int q = wave + 8 * (c_tap * a.pps + e);
        if (q < a.npc) issue_chunk_piece(d_gc % R, q);
      }
    }
    if (w_s < S) {
      issue_weights(w_ws, w_s % D);
      ++w_s;
      if (++w_ws == spt) w_ws = 0;
    }
    u32x4_t bq[NTW], ax[RPW];
    {
      const char* Bs = smem + R * chunk_bytes + (s % D) * WT_BYTES + blane;
#pragma unroll
      for (int j = 0; j < NTW; ++j) bq[j] = *(const u32x4_t*)(Bs + j * 1024);
      const char* As = smem + (c_gc % R) * chunk_bytes + (tyy * HWtc + txx) * 16 + a_lane_off;
#pragma unroll
      for (int i = 0; i < RPW; ++i) ax[i] = *(const u32x4_t*)(As + rowoff[i]);
    }
    const bool last = c_ws + 1 == spt;                // last K-step of the tile
    if (last) {                                       // c_{t-1} of the tile's rows, one K-step ahead of the epilogue
      if (a.c_prev) {                                 // (branch-free per row: pixels outside the grid read pixel (0, 0) of the image)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const int rt = ps * RPW + i;
          const int y = ct.y0 + rt % Rc, xq = ct.x0 + 16 * (rt / Rc) + (lane & 15);
          const bool in = y < a.H && xq < a.W;
          const float* crow = a.c_prev + ((size_t)ct.img * a.H + (y < a.H ? y : 0)) * a.W * a.Chp;     // wave-uniform row base
          cpv[i] = *(const f32x4_t*)(crow + (unsigned)((in ? xq : 0) * a.Chp + (nt0 / 4) * 16 + 4 * (lane >> 4)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < RPW; ++i) cpv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
    wg_barrier();
    // ============================== P2(s): MFMAs, counted wait ==============================
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<NINT_BF16>(bq[j], ax[i], acc[i][j]);   // swapped: D[channel][pixel]
    __builtin_amdgcn_s_setprio(0);
    // weights of step s+1 (group 0) / s+2 (group 1) have landed for this wave; near the end of the run fewer steps are in flight
    if (s + D >= S) wait_vm<0>();
    else if (grp == 0) wait_vm<(D - 2) * PPWB>();
    else wait_vm<(D - 3) * PPWB>();
    wg_barrier();

    // ============================== advance the cursors ==============================
    ++c_ws;
    if (++txx == k) { txx = 0; ++tyy; }
    if (++c_tap == taps) {                            // chunk period over: the fetch cursor moves on with it
      c_tap = 0; tyy = 0; ++c_gc; ++c_c;
      if (d_j < t_cnt) advance_chunk_cursor();
    }
    if (last) {
      // ---------------------------------------------------------------- LSTM epilogue (model.py:221-229)
      const int cblock = nt0 / 4, c4 = 4 * (lane >> 4);
      const int ch = cblock * 16 + c4;
      const int Gc = 4 * a.Ch16;
      const int odd = (lane >> 4) & 1, chb = (lane >> 5) * 8;
      const int px = lane & 15;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int rt = ps * RPW + i;
        const int y = ct.y0 + rt % Rc, x = ct.x0 + 16 * (rt / Rc) + px;
        const bool ok = y < a.H && x < a.W;           // (the lane exchange below needs every lane: no divergent block)
        const size_t rowpix = ((size_t)ct.img * a.H + y) * a.W;
        const f32x4_t cp = cpv[i];
        f32x4_t gi, gf, gg, go, cn, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gi[r] = sigmoidf_(acc[i][0][r]);            // (bias already in the accumulator)
          gf[r] = sigmoidf_(acc[i][1][r]);
          gg[r] = tanhf_(acc[i][2][r]);
          go[r] = sigmoidf_(acc[i][3][r]);
          cn[r] = cp[r] * gf[r] + gi[r] * gg[r];      // model.py:228
          hn[r] = go[r] * tanhf_(cn[r]);              // model.py:229
        }
        if (ok) {
          *(f32x4_t*)(a.c_out + rowpix * a.Chp + (unsigned)(x * a.Chp + ch)) = cn;
          char* hrow = a.h_out + (((size_t)ct.img * a.Hh + (y + a.P)) * a.Wh) * a.Chp * 2;
          store_vec4<NINT_BF16>(hrow, (unsigned)((x + a.P) * a.Chp + ch), hn);
        }
        if (a.gates_out) {
          // 16-byte stash stores: lane rows 2r / 2r+1 trade halves (v_permlane16_swap), as in conv_igemm.hip
          typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
          const u2_t ig0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gi[0], gi[1]), pack_bf16x2(gg[0], gg[1]), false, false);
          const u2_t ig1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gi[2], gi[3]), pack_bf16x2(gg[2], gg[3]), false, false);
          const u2_t fo0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gf[0], gf[1]), pack_bf16x2(go[0], go[1]), false, false);
          const u2_t fo1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(gf[2], gf[3]), pack_bf16x2(go[2], go[3]), false, false);
          if (ok) {
            uint16_t* grow = (uint16_t*)a.gates_out + rowpix * Gc;
            const unsigned lo_g = (unsigned)(x * Gc + cblock * 64 + chb + (odd ? 32 : 0));
            *(u32x4_t*)(grow + lo_g) = (u32x4_t){ig0[0], ig1[0], ig0[1], ig1[1]};        // gate i (even row) / g (odd row)
            *(u32x4_t*)(grow + lo_g + 16) = (u32x4_t){fo0[0], fo1[0], fo0[1], fo1[1]};   // gate f / o
          }
        }
      }
      // ---------------------------------------------------------------- next tile
      c_ws = 0; c_c = 0;
      if (++c_j < t_cnt) {
        ct = wide_tile(a, t_first + c_j * t_stride);
        set_rowoff();
        init_acc();
      }
    }
  }
  if (grp == 0) wg_barrier();                         // pairs with group 1's extra barrier at the start
}

// ------------------------------------------------------------------------------ host side
static unsigned magic_of(int d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

template <int PS, int D, int KS>
static int launch_wide(WideArgs& a, int ny, int n_cu, int R, hipStream_t st) {
  constexpr int CG = 8 / PS;
  const size_t lds = (size_t)R * a.nhpp * 64 + (size_t)D * 4 * CG * 1024;
  if (lds > 160 * 1024) return NINT_E_LDS;
  const int nwg = a.ntiles < n_cu ? a.ntiles : n_cu;
#define NINT_WIDE_LAUNCH(R_)                                                                                          \
  {                                                                                                                   \
    auto kern = conv_wide_lstm_kernel<PS, D, R_, KS>;                                                                     \
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(kern, dim3(nwg, ny), dim3(512), lds, st, a);                                                   \
  }
  if (R == 2) NINT_WIDE_LAUNCH(2) else if (R == 3) NINT_WIDE_LAUNCH(3) else if (R == 4) NINT_WIDE_LAUNCH(4) else return NINT_E_ARG;
#undef NINT_WIDE_LAUNCH
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

// Serves: bf16, LSTM epilogue, k x k taps on both sources (no horizontal fold), k = 3 or 5, gate columns a multiple of 256 (PS = 2),
// enough 256-pixel tiles to give every CU one.  NINT_E_SHAPE = not served (the caller takes conv_igemm.hip's kernel).
int nint_internal_conv_wide_lstm(const ConvArgs& c, int N, int force, void* stream) {
  if (c.kx0 != c.k || (c.k != 3 && c.k != 5) || c.nchunk0 + c.nchunk1 < 1) return NINT_E_SHAPE;
  if (c.NTt % 16 != 0) return NINT_E_SHAPE;
  constexpr int D = 5;
  int dev = 0, n_cu = 0;
  NINT_CHECK_HIP(hipGetDevice(&dev));
  NINT_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  WideArgs a = {};
  a.src0 = c.src0; a.src1 = c.src1; a.nchunk0 = c.nchunk0; a.nchunk1 = c.src1 ? c.nchunk1 : 0;
  a.img_stride0 = c.img_stride0; a.img_stride1 = c.img_stride1; a.pix_stride0 = c.pix_stride0; a.pix_stride1 = c.pix_stride1;
  a.Bp = c.Bp; a.NTt = c.NTt; a.k = c.k; a.p = c.p; a.taps = c.taps;
  a.H = c.H; a.W = c.W; a.P = c.P; a.Hh = c.Hh; a.Wh = c.Wh;
  a.bias = c.bias; a.c_prev = c.c_prev; a.c_out = c.c_out; a.h_out = c.h_out; a.gates_out = c.gates_out;
  a.Chp = c.Chp; a.Ch16 = c.Ch16;
  // tile shape: R rows x (16 / R) blocks of 16 pixels; leftover rows (<= R/2 of them) as a strip of flatter tiles.
  // Fewest tiles wins, then the smaller halo.
  int best_tiles = 1 << 30, best_halo = 1 << 30;
  for (int R0 = 16; R0 >= 1; R0 >>= 1) {
    const int bx = nint_cdiv(a.W, 16);               // 16-pixel blocks the grid is wide: a tile is never wider
    const int Cb0 = 16 / R0 < bx ? 16 / R0 : bx;
    int ty0 = a.H / R0, left = a.H % R0, R1 = 0;
    if (left) {
      R1 = 1;
      while (R1 < left) R1 <<= 1;
      if (R1 >= R0) { R1 = 0; ++ty0; }               // more than half a tile of rows left: one more row of full tiles
    }
    const int Cb1 = R1 ? (16 / R1 < bx ? 16 / R1 : bx) : 0;
    const int tx0 = nint_cdiv(a.W, 16 * Cb0), tx1 = R1 ? nint_cdiv(a.W, 16 * Cb1) : 0;
    const int tiles = ty0 * tx0 + tx1;
    const int halo = (R0 + 2 * a.p) * (16 * Cb0 + 2 * a.p);
    if (tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
      best_tiles = tiles; best_halo = halo;
      a.R[0] = R0; a.Cb[0] = Cb0; a.R[1] = R1; a.Cb[1] = Cb1;
      a.tiles_x[0] = tx0; a.tiles_x[1] = tx1; a.tiles_y0 = ty0;
    }
  }
  a.tiles_img = best_tiles;
  a.ntiles = N * best_tiles;
  if (!force && a.ntiles * 16 < n_cu * 15) return NINT_E_SHAPE;      // fewer tiles than CUs: the 4-wave kernel's small tiles fill the chip better
  int nhp_max = 0;
  for (int q = 0; q < 2; ++q) {
    if (q == 1 && !a.R[1]) { a.R[1] = a.R[0]; a.Cb[1] = a.Cb[0]; a.HWt[1] = a.HWt[0]; a.NHP[1] = a.NHP[0]; a.magic_hwt[1] = a.magic_hwt[0]; break; }
    a.HWt[q] = 16 * a.Cb[q] + 2 * a.p;
    a.NHP[q] = (a.R[q] + 2 * a.p) * a.HWt[q];
    a.magic_hwt[q] = magic_of(a.HWt[q]);
    if (a.NHP[q] > nhp_max) nhp_max = a.NHP[q];
  }
  a.nhpp = nint_round_up(nhp_max, 16);
  if (4 * a.nhpp >= 65536) return NINT_E_SHAPE;                       // (multiply-high division is exact below 2^16)
  a.magic_nhpp = magic_of(a.nhpp);
  a.npc = a.nhpp / 16;
  a.spt = (a.nchunk0 + a.nchunk1) * a.taps;
  // chunk ring depth and issue rate: the pieces of chunk g+R-1 go out during the first steps of period g and must be
  // older than every weight piece waited for at the start of period g+R-1 (header comment): last issue step <= (R-1)*taps - D
  const int wave_pieces = nint_cdiv(a.npc, 8);
  int R = 0;
  for (int r = 2; r <= 4 && !R; ++r) {
    for (int pps = 1; pps <= 4; ++pps) {
      const int issue_steps = nint_cdiv(wave_pieces, pps);
      if (issue_steps <= a.taps && issue_steps - 1 <= (r - 1) * a.taps - D &&
          (size_t)r * a.nhpp * 64 + (size_t)D * 16 * 1024 <= 160 * 1024) { R = r; a.pps = pps; break; }
    }
  }
  if (!R) return NINT_E_SHAPE;
  return a.k == 5 ? launch_wide<2, D, 5>(a, a.NTt / 16, n_cu, R, (hipStream_t)stream)
                  : launch_wide<2, D, 3>(a, a.NTt / 16, n_cu, R, (hipStream_t)stream);
}
