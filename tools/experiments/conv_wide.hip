// conv_wide.hip -- the gate convolution of the WIDE ConvLSTM layers as a persistent 8-wave implicit GEMM whose
// weight tiles go through LDS once per workgroup (gfx950, bf16).            reference: model.py:219-229
//
// OPT-IN (nint_layer.wide >= 2): bit-identical to conv_igemm.hip's 4-wave kernel, a third less traffic on the texture path,
// and 10-17 % SLOWER at the bench shape -- DESIGN.md 4.4, profiles/HISTORY.md and profiles/r03_b_* hold the stamp / PMC record of why.  Kept
// for that record's tools (tools/wideprobe.py, tools/wide_stress.py) and tests (tests/test_gpu_wide.py).
//
// conv_igemm.hip's 4-wave kernel streams every wave's own weight fragments L2 -> L1 -> VGPR: for the reference's layer 0
// (Conv2d(62+64 -> 256, k=5)) that is 1.66 GB of on-chip weight traffic per launch through the texture path (TD busy
// 72 % of the launch, profiles/r02_b_pmc_memory_path_counters.txt), and its 1000 workgroups run their fill / K loop /
// epilogue phases in two chip-wide rounds.  This kernel restructures the same arithmetic:
//
//   * a workgroup = 8 waves = PS pixel slices x CG column groups (PS*CG = 8); a wave computes 8 row tiles of 16 pixels x
//     4 column tiles (one hidden-channel block: its i,f,g,o tiles share lanes, as in conv_igemm.hip).  A UNIT of work =
//     one pixel tile of TRT = 8*PS row tiles (256 or 512 pixels: R rows x Cb blocks of 16, R a power of two) x one set of
//     64*CG gate columns; the column-group sets of a pixel tile are consecutive units.
//   * B: the K-step's weight tile (4*CG KiB, already in MFMA fragment order in global memory) is copied ONCE per
//     workgroup by LDS-DMA (global_load_lds_dwordx4, every wave issues its share) into a D-slot ring, D-1 steps ahead;
//     all PS pixel slices read their fragments from there (ds_read_b128, lane-linear: conflict-free).
//   * A: the halo tile is staged per 64-byte channel CHUNK into an R-slot ring: while the K loop runs the taps of chunk c,
//     the pieces of chunk c+R-1 -- of this unit or of the workgroup's NEXT unit -- arrive by LDS-DMA.  The workgroup is
//     PERSISTENT (one per CU, units dealt in XCD-contiguous ranges), so only its very first chunks are waited for.
//   * synchronisation: raw s_barrier + counted s_waitcnt vmcnt(N).  All waves run the same program,
//     {P1: issue DMA, read fragments | barrier | P2: MFMAs, counted wait | barrier} per K-step, but waves 4-7
//     (the second wave of every SIMD) run ONE BARRIER BEHIND waves 0-3: while one wave of a SIMD holds the matrix pipe the
//     other one reads LDS and issues DMA (MI355X_MICROARCH.md, two waves per SIMD, item 9).
//   * all per-step bookkeeping is incremental scalar state: no division, multiplication or kernel-argument reload in the loop.
//   * the LSTM epilogue is conv_igemm.hip's (D = [channel][pixel], 16 / 8-byte vectors, bias in the accumulator init);
//     c_{t-1} is fetched one K-step ahead of it.
//
// Hazards, by barrier count (beta_n = the n-th barrier; group 0 = waves 0-3 runs P1(s) before beta_2s and P2(s) before
// beta_2s+1; group 1 = waves 4-7 one barrier later):
//   RAW weights of step s: read by group 0 after beta_2s-1.  Every wave waits for ITS pieces of step s before that
//       barrier: group 0 in P2(s-1) (N = (D-2) steps' pieces may stay in flight), group 1 in P2(s-2) (N = D-3 steps').
//   WAR weight slot of step s (reused by step s+D): last read by group 1 in P1(s), before beta_2s+1; step s+D is issued
//       in P1(s+1), after beta_2s+1 (group 0) / beta_2s+2 (group 1).  (The fragment reads of P1 are consumed by the MFMAs of
//       P2, so they may still be in flight one interval later; a DMA piece issued then lands hundreds of cycles after them.)
//   RAW chunk g+R-1: its pieces are issued in the first steps of chunk period g, BEFORE the weight pieces of the same
//       step, so they are older than every weight piece that is waited for at the start of period g+R-1 (the host checks
//       that the issue steps end D steps before that period).  WAR: the slot held chunk g-1, whose last read (group 1) is
//       before the first barrier of period g.
//   The counted waits are exact only among LDS-DMA pieces: a register-staged variant of the weight path (plain loads +
//   ds_write, removed) showed that a DMA piece can complete, and decrement vmcnt, before an OLDER plain load.  Around a unit's
//   end this wave also has plain loads (c_{t-1}, the next unit's bias) and the epilogue's stores in flight, so there -- the
//   unit's last step and the D steps behind the epilogue -- and at the end of the run the wait is a full drain.
// hipcc does not know about the inline-asm DMA (it counts neither their vmcnt nor their LDS writes); its own waits for the
// epilogue's loads only ever become more conservative by operations it does not see.
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
  int ny, nunits;                              // column-group sets (of 64*CG gate columns) per pixel tile; units = ntiles * ny
  int nhpp;                                    // pixels per g-plane of a chunk slot (>= NHP of both classes, multiple of 16)
  unsigned magic_nhpp;
  int npc, pps;                                // 1-KiB pieces per chunk; pieces per wave per K-step while a chunk is being fetched
  int spt;                                     // K-steps per unit
  int rot;                                     // 1: every workgroup starts the taps of a chunk at its own tap (blockIdx.x % taps)
  const float* bias; const float* c_prev; float* c_out; char* h_out; char* gates_out;
  int Chp, Ch16;
};

struct WTile { int img, y0, x0, cls, ch; };

#ifdef NINT_STAMP
// Diagnostic build only (tools/wideprobe.py): per-wave cycle totals of the K loop's segments.  The values go to a buffer of
// their own that no kernel reads; the shipped library is built without NINT_STAMP.
#define WIDE_STAMP_WGS 512
__device__ unsigned long long g_wide_stamp[WIDE_STAMP_WGS * 8 * 16];
extern "C" int nint_debug_read_wide_stamps(unsigned long long* host, int n_wgs) {
  if (!host || n_wgs <= 0 || n_wgs > WIDE_STAMP_WGS) return NINT_E_ARG;
  NINT_CHECK_HIP(hipDeviceSynchronize());
  NINT_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wide_stamp), (size_t)n_wgs * 8 * 16 * sizeof(unsigned long long)));
  return NINT_OK;
}
#if NINT_STAMP >= 2      // full: every segment of every K-step (the stamps themselves cost ~130 cycles per pair and drain lgkmcnt)
#define WSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#define WACC(slot, a_, b_) st_acc[slot] += (b_) - (a_);
#else                    // light: whole-run totals only (prologue, epilogues; the K loop is undisturbed)
#define WSTAMP(var)
#define WACC(slot, a_, b_)
#endif
#define WSTAMP1(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#define WACC1(slot, a_, b_) st_acc[slot] += (b_) - (a_);
#else
#define WSTAMP1(var)
#define WACC1(slot, a_, b_)
#define WSTAMP(var)
#define WACC(slot, a_, b_)
#endif

// unit u = (pixel tile u / ny, column-group set u % ny): the sets of one pixel tile are consecutive units, so the workgroup
// (or its XCD neighbours) that runs the next set finds the tile's halo pixels in L2
__device__ __forceinline__ WTile wide_tile(const WideArgs& a, int u) {
  WTile t;
  const int id = u / a.ny;
  t.ch = u - id * a.ny;
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

// One 1-KiB LDS-DMA piece: lane l copies 16 bytes from gsrc (per lane) to lds_dst + 16*l (wave-uniform base in M0).
// M0 is written and NOT restored: nothing else in this translation unit's kernels uses M0 (gfx950 LDS instructions do not;
// checked in the ISA: no other m0 operand).
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
// two consecutive pieces (global +1024, LDS +1024: the instruction offset applies to both addresses) behind ONE M0 write
__device__ __forceinline__ void glds16x2(const char* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
               : : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 12, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int PS, int D, int R, int KS>
__global__ __launch_bounds__(512, 2) void conv_wide_lstm_kernel(WideArgs a) {
  constexpr int CG = 8 / PS, RPW = 8, NTW = 4;       // 8 row tiles x 4 column tiles per wave: 128 accumulator registers
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

  // ---- this workgroup's units.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one L2), so XCD x
  // takes the contiguous unit range x and its workgroups walk it with the stride of their number: tiles that are
  // neighbours in space (shared halo pixels, the same images) are neighbours in time on one L2.  Any bijection is correct.
  int t_first, t_cnt, t_stride;
  {
    const int G = gridDim.x, b = blockIdx.x, x = b % 8, i = b / 8;
    const int nx = G / 8 + (x < G % 8 ? 1 : 0);
    const int q8 = a.nunits / 8, r8 = a.nunits % 8;
    const int lo = x * q8 + (x < r8 ? x : r8), sz = q8 + (x < r8 ? 1 : 0);
    t_first = lo + i; t_stride = nx;
    t_cnt = i < sz ? (sz - i + nx - 1) / nx : 0;
  }
  if (t_cnt == 0) return;
#ifdef NINT_STAMP
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int S = t_cnt * spt;                          // K-steps of this workgroup's whole run
  int nt0;                                            // first n-tile of this wave in the unit being computed

  // ---- DMA issue ------------------------------------------------------------------------------------------------
  // weights: piece j of K-step ws of column-group set ch comes from Bp + ((ws * NTt + ch * 4 * CG + j) << 10), lane-linear on both sides
  const unsigned bstep = (unsigned)a.NTt * 1024;
  const int ny = a.ny;
  int w_ch = t_first % ny;                            // column-group set of the unit whose weights are being fetched
  const int w_chstep = t_stride % ny;                 // ... and how it moves from one of this workgroup's units to the next
  // K order: chunk by chunk, and inside a chunk the k*k taps CYCLICALLY from this workgroup's own first tap t0 (a.rot).  In
  // plain order all persistent workgroups walk the packed weights in lockstep, i.e. every CU of an XCD asks its L2 for the
  // same tile at the same moment, step after step; rotated, the requests of a moment spread over k*k weight tiles
  // (measured: no gain, 2186-2387 vs 2377 cycles per K-step; results then differ from the plain order in the last bits).
  const int t0 = a.rot ? (int)(blockIdx.x % taps) : 0;
  const int ty0 = t0 / k, tx0 = t0 - ty0 * k;
  unsigned w_off = (unsigned)w_ch * WT_BYTES + (unsigned)t0 * bstep;   // byte offset of the next step to issue inside the packed weights
  int w_left = S;                                     // steps not yet issued
  int w_ws = 0;                                       // ... its index inside its unit
  int w_tapi = t0, w_cnt = 0;                         // ... its tap, and how many steps of its chunk period have been issued
  const bool w_issuer = !(CG == 1 && wave >= 4);
  const char* Bw_lane = a.Bp + (size_t)(wave * PPWB) * 1024 + lane * 16;   // this wave's first piece of step 0, set 0
  unsigned w_lds = wring + (unsigned)(wave * PPWB) * 1024;                 // LDS address of that piece in the next ring slot
  int w_slot = 0;
  auto issue_weights = [&]() __attribute__((always_inline)) {
    if (w_issuer) {
      if constexpr (PPWB == 2) glds16x2(Bw_lane + w_off, w_lds);
      else glds16(Bw_lane + w_off, w_lds);
    }
    --w_left;
    w_off += bstep; ++w_ws; ++w_tapi; ++w_cnt;
    if (w_tapi == taps) { w_tapi = 0; w_off -= taps * bstep; }        // tap k*k-1 -> tap 0 of the same chunk
    if (w_cnt == taps) { w_cnt = 0; w_off += taps * bstep; }          // period over (the tap is back at t0): next chunk
    if (w_ws == spt) {                                // next unit of this workgroup: (u + t_stride) % ny
      w_ws = 0;
      w_ch += w_chstep;
      if (w_ch >= ny) w_ch -= ny;
      w_off = (unsigned)w_ch * WT_BYTES + (unsigned)t0 * bstep;
    }
    w_lds += WT_BYTES; ++w_slot;
    if (w_slot == D) { w_slot = 0; w_lds -= D * WT_BYTES; }
  };
  // halo chunk: piece q = 64 consecutive 16-byte units of the slot image [g][halo pixel]; unit u -> (g, halo pixel) ->
  // source address.  Pad units (hp >= NHP) and pixels beyond the slab (tiles that overhang the slack of the halo slab;
  // only masked output pixels read them) re-read the tile's first pixel.
  // The chunk being fetched is described by scalars that change once per chunk (fetch descriptor): first byte of the
  // tile's first halo pixel in that chunk, the source's pixel stride, the tile class and the slab's remaining extent.
  const char* f_base; int f_pixs, f_nhp, f_hwt, f_hymax, f_hxmax; unsigned f_magic;
  unsigned f_lds = lds0;                              // LDS address of the slot being filled
  int f_slot = 0;
  const int nhpp = a.nhpp, Wh = a.Wh, npc = a.npc, pps = a.pps;
  const unsigned magic_nhpp = a.magic_nhpp;
  auto issue_chunk_piece = [&](int q) __attribute__((always_inline)) {
    const int u = q * 64 + lane;
    const int g = (int)__umulhi((unsigned)u, magic_nhpp);
    int hp = u - g * nhpp;
    hp = hp < f_nhp ? hp : 0;
    const int hy = (int)__umulhi((unsigned)hp, f_magic);
    const int hx = hp - hy * f_hwt;
    const bool inside = hy < f_hymax && hx < f_hxmax;
    const int off = inside ? (hy * Wh + hx) * f_pixs : 0;
    glds16(f_base + off + g * 16, f_lds + q * 1024);
  };

  // chunk-fetch cursor: unit d_j, chunk d_c of it
  int d_c = 0, d_j = 0;
  WTile d_tile = wide_tile(a, t_first);
  auto set_fetch = [&]() __attribute__((always_inline)) {
    const int ay0 = d_tile.y0 + a.P - p, ax0 = d_tile.x0 + a.P - p;
    const long pix = (long)ay0 * Wh + ax0;
    if (d_c < a.nchunk0) { f_pixs = a.pix_stride0; f_base = a.src0 + (long)d_tile.img * a.img_stride0 + pix * a.pix_stride0 + d_c * 64; }
    else { f_pixs = a.pix_stride1; f_base = a.src1 + (long)d_tile.img * a.img_stride1 + pix * a.pix_stride1 + (d_c - a.nchunk0) * 64; }
    f_nhp = a.NHP[d_tile.cls]; f_hwt = a.HWt[d_tile.cls]; f_magic = a.magic_hwt[d_tile.cls];
    f_hymax = a.Hh - ay0; f_hxmax = Wh - ax0;
  };
  set_fetch();
  auto advance_chunk_cursor = [&]() __attribute__((always_inline)) {
    f_lds += chunk_bytes; ++f_slot;
    if (f_slot == R) { f_slot = 0; f_lds = lds0; }
    if (++d_c == nchunks) {
      d_c = 0; ++d_j;
      if (d_j < t_cnt) d_tile = wide_tile(a, t_first + d_j * t_stride);
    }
    if (d_j < t_cnt) set_fetch();
  };

  // ---- prologue: chunks 0 .. R-2 whole, weights of steps 0 .. D-2; one full drain
  for (int r = 0; r < R - 1 && d_j < t_cnt; ++r) {
    for (int q = wave; q < npc; q += 8) issue_chunk_piece(q);
    advance_chunk_cursor();
  }
  for (int i = 0; i < D - 1 && w_left > 0; ++i) issue_weights();
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
  WTile ct = wide_tile(a, t_first);                   // the unit being computed
  nt0 = ct.ch * 4 * CG + cg * 4;
  init_acc();
  int c_j = 0, c_ws = 0;                              // its number, K-step inside it
  int c_tap = 0, tyy = ty0, txx = tx0;                // step inside the chunk period; the tap it computes
  int a_soff = 0, a_slot = 0;                         // byte offset of (chunk slot, tap) inside the A ring
  int b_soff = R * chunk_bytes, b_slot = 0;           // byte offset of this step's weight slot
  const int a_lane_off = (lane >> 4) * plane + (lane & 15) * 16;
  const unsigned blane = (unsigned)(cg * 4 * 1024 + lane * 16);
  // row tile i of this wave is tile row tile rt = ps*RPW + i at (rt & (Rc-1), 16 * (rt >> lgR)) -- tile heights are powers of two
  int arow[RPW], lgR, Rm, HWtc, d_rowwrap, nrt;
  auto set_tile = [&]() __attribute__((always_inline)) {
    const int Rc = a.R[ct.cls];
    lgR = 31 - __builtin_clz(Rc); Rm = Rc - 1; HWtc = a.HWt[ct.cls];
    d_rowwrap = (HWtc - k) * 16;                      // tap (ty, k-1) -> (ty+1, 0), on top of the +16 of every step
    nrt = Rc * a.Cb[ct.cls];                          // row tiles the class really has (tiles narrower than TRT / R blocks: fewer)
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rt = ps * RPW + i;
      // (idle row tiles read row tile 0 and store nothing)
      arow[i] = a_lane_off + (rt < nrt ? ((rt & Rm) * HWtc + 16 * (rt >> lgR)) * 16 : 0);
    }
  };
  set_tile();
  a_soff = (ty0 * HWtc + tx0) * 16;
  f32x4_t cpv[RPW];

  WSTAMP1(st_loop0)
  WACC1(6, st_t0, st_loop0)
  for (int s = 0; s < S; ++s) {
    WSTAMP(st_a)
    // ============================== P1(s): DMA issue, fragment reads ==============================
    // pieces of the chunk R-1 periods ahead (fetch cursor) during the first steps of this chunk period, ahead of the weights
    if (d_j < t_cnt) {
      for (int e = 0; e < pps; ++e) {
        const int q = wave + 8 * (c_tap * pps + e);
        if (q < npc) issue_chunk_piece(q);
      }
    }
    if (w_left > 0) issue_weights();
    u32x4_t bq[NTW], ax[RPW];
    {
      const char* Bs = smem + b_soff + blane;
#pragma unroll
      for (int j = 0; j < NTW; ++j) bq[j] = *(const u32x4_t*)(Bs + j * 1024);
#pragma unroll
      for (int i = 0; i < RPW; ++i) ax[i] = *(const u32x4_t*)(smem + (a_soff + arow[i]));
    }
    const bool last = c_ws + 1 == spt;                // last K-step of the unit
    if (last) {                                       // c_{t-1} of the tile's rows, one K-step ahead of the epilogue
      if (a.c_prev) {                                 // (branch-free per row: pixels outside the grid read pixel (0, 0) of the image)
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          const int rt = ps * RPW + i;
          const int y = ct.y0 + (rt & Rm), xq = ct.x0 + 16 * (rt >> lgR) + (lane & 15);
          const bool in = y < a.H && xq < a.W;
          const float* crow = a.c_prev + ((size_t)ct.img * a.H + (y < a.H ? y : 0)) * a.W * a.Chp;     // wave-uniform row base
          cpv[i] = *(const f32x4_t*)(crow + (unsigned)((in ? xq : 0) * a.Chp + (nt0 / 4) * 16 + 4 * (lane >> 4)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < RPW; ++i) cpv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      }
    }
    WSTAMP(st_b)
    wg_barrier();
    WSTAMP(st_c)
    // ============================== P2(s): MFMAs, counted wait ==============================
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<NINT_BF16>(bq[j], ax[i], acc[i][j]);   // swapped: D[channel][pixel]
    __builtin_amdgcn_s_setprio(0);
    WSTAMP(st_d)
    // weights of step s+1 (group 0) / s+2 (group 1) have landed for this wave; a full drain where other kinds of operations
    // are in flight (header) and near the end of the run
    if (s + D >= S || last || (c_j > 0 && c_ws < D) || !w_issuer) wait_vm<0>();
    else if (grp == 0) wait_vm<(D - 2) * PPWB>();
    else wait_vm<(D - 3) * PPWB>();
    WSTAMP(st_e)
    wg_barrier();
    WSTAMP(st_f)
    WACC(0, st_a, st_b) WACC(1, st_b, st_c) WACC(2, st_c, st_d) WACC(3, st_d, st_e) WACC(4, st_e, st_f)

    // ============================== advance the cursors ==============================
    ++c_ws;
    b_soff += WT_BYTES; ++b_slot;
    if (b_slot == D) { b_slot = 0; b_soff -= D * WT_BYTES; }
    a_soff += 16; ++txx;
    if (txx == k) {
      txx = 0; a_soff += d_rowwrap; ++tyy;
      if (tyy == k) { tyy = 0; a_soff -= k * HWtc * 16; }             // tap (k-1, k-1) -> (0, 0)
    }
    bool new_period = false;
    if (++c_tap == taps) {                            // chunk period over (the tap is back at t0): next slot of the A ring; the fetch cursor moves on with it
      c_tap = 0;
      ++a_slot;
      if (a_slot == R) a_slot = 0;
      new_period = true;
      if (d_j < t_cnt) advance_chunk_cursor();
    }
    if (last) {
      WSTAMP1(st_ep0)
      // ---------------------------------------------------------------- LSTM epilogue (model.py:221-229)
      const int cblock = nt0 / 4, c4 = 4 * (lane >> 4);
      const int ch = cblock * 16 + c4;
      const int Gc = 4 * a.Ch16;
      const int odd = (lane >> 4) & 1, chb = (lane >> 5) * 8;
      const int px = lane & 15;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int rt = ps * RPW + i;
        const int y = ct.y0 + (rt & Rm), x = ct.x0 + 16 * (rt >> lgR) + px;
        const bool ok = rt < nrt && y < a.H && x < a.W;   // (the lane exchange below needs every lane: no divergent block)
        const size_t rowpix = ((size_t)ct.img * a.H + y) * a.W;
        const f32x4_t cp = cpv[i];
        f32x4_t gi, gf, gg, go, cn, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gi[r] = sigmoidf_(acc[i][0][r]);            // (bias already in the accumulator)
          gf[r] = sigmoidf_(acc[i][1][r]);
          gg[r] = tanhf_(acc[i][2][r]);
          go[r] = sigmoidf_(acc[i][3][r]);
          cn[r] = fmaf(cp[r], gf[r], gi[r] * gg[r]);  // model.py:228 (association pinned as in conv_igemm.hip)
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
      // ---------------------------------------------------------------- next unit
      c_ws = 0;
      if (++c_j < t_cnt) {
        ct = wide_tile(a, t_first + c_j * t_stride);
        nt0 = ct.ch * 4 * CG + cg * 4;
        set_tile();
        init_acc();
      }
      WSTAMP1(st_g)
      WACC1(5, st_ep0, st_g)
    }
    if (new_period) a_soff = a_slot * chunk_bytes + (ty0 * HWtc + tx0) * 16;   // (after a unit switch: the new tile's halo width)
  }
  if (grp == 0) wg_barrier();                         // pairs with group 1's extra barrier at the start
#ifdef NINT_STAMP
  if (lane == 0 && blockIdx.x < WIDE_STAMP_WGS) {
    unsigned long long* o = g_wide_stamp + ((size_t)blockIdx.x * 8 + wave) * 16;
    for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    o[8] = __builtin_amdgcn_s_memtime() - st_t0; o[9] = __builtin_amdgcn_s_memrealtime() - st_r0; o[10] = S; o[11] = t_cnt;
    o[12] = st_r0; o[13] = st_acc[8]; o[14] = st_acc[9]; o[15] = st_acc[10];
  }
#endif
}

// ------------------------------------------------------------------------------ host side
static unsigned magic_of(int d) { return (unsigned)(((1ull << 32) + d - 1) / d); }

template <int PS, int D, int KS>
static int launch_wide(WideArgs& a, int n_cu, int R, hipStream_t st) {
  constexpr int CG = 8 / PS;
  const size_t lds = (size_t)R * a.nhpp * 64 + (size_t)D * 4 * CG * 1024;
  if (lds > 160 * 1024) return NINT_E_LDS;
  const int nwg = a.nunits < n_cu ? a.nunits : n_cu;
#define NINT_WIDE_LAUNCH(R_)                                                                                          \
  {                                                                                                                   \
    auto kern = conv_wide_lstm_kernel<PS, D, R_, KS>;                                                                 \
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, a);                                                       \
  }
  if (R == 2) NINT_WIDE_LAUNCH(2) else if (R == 3) NINT_WIDE_LAUNCH(3) else if (R == 4) NINT_WIDE_LAUNCH(4) else return NINT_E_ARG;
#undef NINT_WIDE_LAUNCH
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

// Plan of one tile size (PS pixel slices of 8 row tiles: 256 or 512 pixels per workgroup tile, 64*8/PS gate columns per unit).
template <int PS, int D>
static int plan_wide(WideArgs& a, int N, int* R_out) {
  constexpr int CG = 8 / PS, TRT = PS * 8;
  if (a.NTt % (4 * CG)) return NINT_E_SHAPE;
  a.ny = a.NTt / (4 * CG);
  // tile shape: R rows x (TRT / R) blocks of 16 pixels; leftover rows (<= R/2 of them) as a strip of flatter tiles.
  // Fewest tiles wins, then the smaller halo.
  int best_tiles = 1 << 30, best_halo = 1 << 30;
  const int bx = nint_cdiv(a.W, 16);                 // 16-pixel blocks the grid is wide: a tile is never wider
  for (int R0 = TRT; R0 >= 1; R0 >>= 1) {
    // (widths balanced over the tiles of a row: 154 pixels = 10 blocks in tiles of at most 8 -> 2 tiles of 5, not 8 + 2)
    const int Cm0 = TRT / R0 < bx ? TRT / R0 : bx;
    const int Cb0 = nint_cdiv(bx, nint_cdiv(bx, Cm0));
    int ty0 = a.H / R0, left = a.H % R0, R1 = 0;
    if (left) {
      R1 = 1;
      while (R1 < left) R1 <<= 1;
      if (R1 >= R0) { R1 = 0; ++ty0; }               // more than half a tile of rows left: one more row of full tiles
    }
    const int Cm1 = R1 ? (TRT / R1 < bx ? TRT / R1 : bx) : 0;
    const int Cb1 = R1 ? nint_cdiv(bx, nint_cdiv(bx, Cm1)) : 0;
    const int tx0 = nint_cdiv(a.W, 16 * Cb0), tx1 = R1 ? nint_cdiv(a.W, 16 * Cb1) : 0;
    const int tiles = ty0 * tx0 + tx1;
    int halo = (R0 + 2 * a.p) * (16 * Cb0 + 2 * a.p);
    if (R1 && (R1 + 2 * a.p) * (16 * Cb1 + 2 * a.p) > halo) halo = (R1 + 2 * a.p) * (16 * Cb1 + 2 * a.p);
    if (tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
      best_tiles = tiles; best_halo = halo;
      a.R[0] = R0; a.Cb[0] = Cb0; a.R[1] = R1; a.Cb[1] = Cb1;
      a.tiles_x[0] = tx0; a.tiles_x[1] = tx1; a.tiles_y0 = ty0;
    }
  }
  a.tiles_img = best_tiles;
  a.ntiles = N * best_tiles;
  a.nunits = a.ntiles * a.ny;
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
  // chunk ring depth and issue rate: the pieces of chunk g+R-1 go out during the first steps of period g and must be older
  // than every weight piece waited for at the start of period g+R-1 (header comment): last issue step <= (R-1)*taps - D
  const int wave_pieces = nint_cdiv(a.npc, 8);
  for (int r = 2; r <= 4; ++r) {
    for (int pps = 1; pps <= 8; ++pps) {
      const int issue_steps = nint_cdiv(wave_pieces, pps);
      const bool in_time = issue_steps <= a.taps && issue_steps - 1 <= (r - 1) * a.taps - D;
      if (in_time && (size_t)r * a.nhpp * 64 + (size_t)D * 4 * CG * 1024 <= 160 * 1024) { *R_out = r; a.pps = pps; return NINT_OK; }
    }
  }
  return NINT_E_LDS;
}

// Serves: bf16, LSTM epilogue, k x k taps on both sources (no horizontal fold), k = 3 or 5, gate columns a multiple of 128.
// NINT_E_SHAPE = not served (the caller takes conv_igemm.hip's kernel).
// force (nint_layer.wide): 0 = the library's choice -- never this kernel, see below; 2 = wherever it is instantiated; 3 / 4 = that,
// with 256- / 512-pixel tiles; + 8 = taps in plain order in every workgroup (results then equal the 4-wave kernel's bit for bit).
int nint_internal_conv_wide_lstm(const ConvArgs& c, int N, int force_, void* stream) {
  const int force = force_ & 7, rot = (force_ & 8) ? 0 : 1;
  // Measured on MI355X at the bench shape (B = 8, 100 x 154, 62 + 64 -> 256, k = 5): 187-200 us per launch in the step against
  // 167-171 us for the 4-wave kernel -- the K loop runs ~1950 cycles per K-step instead of the ~1000 it reaches with the DMA
  // issue ablated (profiles/HISTORY.md: an LDS-DMA piece costs its issuing wave ~200 cycles, a halo gather piece far more, and the
  // barrier-coupled loop makes all eight waves wait for it).
  if (!force) return NINT_E_SHAPE;
  if (c.kx0 != c.k || (c.k != 3 && c.k != 5) || c.nchunk0 + c.nchunk1 < 1) return NINT_E_SHAPE;
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
  // 512-pixel tiles (half the weight bytes per MFMA) when their units fill the CUs about as evenly as the 256-pixel ones;
  // efficiency = units / (rounds * CUs)
  WideArgs a4 = a, a2 = a;
  int R4 = 0, R2 = 0;
  const int rc4 = force == 3 ? NINT_E_SHAPE : plan_wide<4, D>(a4, N, &R4);
  const int rc2 = force == 4 ? NINT_E_SHAPE : plan_wide<2, D>(a2, N, &R2);
  auto eff = [&](const WideArgs& w) { return (double)w.nunits / ((double)nint_cdiv(w.nunits, n_cu) * n_cu); };
  const bool use4 = rc4 == NINT_OK && (rc2 != NINT_OK || eff(a4) >= 0.9 * eff(a2));
  if (!use4 && rc2 != NINT_OK) return NINT_E_SHAPE;
  WideArgs& w = use4 ? a4 : a2;
  w.rot = rot;
  hipStream_t st = (hipStream_t)stream;
  if (use4) return a.k == 5 ? launch_wide<4, D, 5>(w, n_cu, R4, st) : launch_wide<4, D, 3>(w, n_cu, R4, st);
  return a.k == 5 ? launch_wide<2, D, 5>(w, n_cu, R2, st) : launch_wide<2, D, 3>(w, n_cu, R2, st);
}
