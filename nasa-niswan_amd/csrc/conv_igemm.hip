// conv_igemm.hip -- implicit-GEMM convolution on MFMA 16x16 tiles for gfx950.
//
// One kernel template serves both convolutions of the ConvLSTM cell:
//   EPI_LSTM  : gates = W (*) cat[x,h] + b, sigmoid/tanh, c/h update      (reference model.py:219-229)
//   EPI_DGRAD : d cat[x,h] = W^T (*) dG                                    (autograd of model.py:220)
//   EPI_DGRAD_PW : the same, with the pointwise LSTM backward of the PREVIOUS time step (autograd of model.py:222-229)
//                  run in the epilogue on the h columns while they are still accumulators
//
// GEMM view: M = pixels (16 consecutive x per MFMA row tile), N = output columns
// (gate channels / cat channels), K = (64-byte channel chunk, tap).  The zero padding of
// nn.Conv2d is physical: sources are halo slabs whose border is kept zero, so no tap is ever
// predicated.  A workgroup (4 waves) owns MT (8 or 4) rows x 16 columns of pixels:
//   - A: its (MT+2p) x (16+2p) halo tile is staged in LDS, several channel chunks per fill, laid
//     out [chunk][g][halo pixel][16 B] (g = the lane group that consumes those 16 bytes), so
//     every A fragment read is a lane-linear ds_read_b128 (no bank conflicts) and a tap is a
//     constant address offset.  The image is read-only between fills: NO barrier in the K loop.
//   - B: weights are pre-packed in MFMA fragment order, so a wave's B fragment is one fully
//     coalesced 1 KiB global load straight into VGPRs (L2-resident, 3-deep register ring).
//     Waves never share B: the 4 waves split the work as WN column groups x WK K-slices, every
//     wave computing MT row tiles x NTW column tiles.  K-slices (narrow-N launches) are summed
//     through ONE LDS exchange at the end, after which every wave owns MT/WK rows of the epilogue.
//   - operands are fed to the MFMA swapped (A := weights, B := pixels), so D = [channel][pixel]: a
//     lane owns 4 consecutive channels of one pixel -> 16-byte / 8-byte vector epilogue; the
//     i,f,g,o tiles of one hidden channel sit in the same lane (column order
//     n' = (cblock*4+gate)*16+col), so the LSTM epilogue needs no cross-lane traffic.
#include "nint_common.h"

enum { EPI_LSTM = 0, EPI_DGRAD = 1, EPI_DGRAD_PW = 2 };

// The library's own choice between the stencil kernel and the padded MFMA tiles for layers both hold (tile_rows == 0).
// Measured on the full 100 x 154 grid, B = 8, configs[0]'s layer (4 -> 8, 3x3): see DESIGN.md section 6.
// ... and between the dense-K MFMA kernel (csrc/tiny_gemm.hip) and the padded tiles: measured on the same launch (configs[0]'s layer,
// full grid, B = 8, profiles/r04_d_tiny_layer_families.txt): f32 34.0 against 45.7 us, bf16 23.9 against 22.0 us -- both bf16 forms
// sit on the slab layout's channel padding (the epilogue's partial-line stores), not on their matrix work.
#ifndef NINT_TINY_AUTO
#define NINT_TINY_AUTO(dtype) ((dtype) == NINT_F32)
#endif
#ifndef NINT_STENCIL_AUTO
#define NINT_STENCIL_AUTO(dtype) false
#endif

// MT (template) = row tiles per wave = rows of the pixel tile: 8 for the wide launches (one weight
// stream per 128 pixels), 4 for the short-K launches of the narrow layers, whose time is all halo
// fill + epilogue latency: smaller tiles put 3-4 workgroups per CU in flight instead of 2.
#ifndef NINT_BD
#define NINT_BD 3
#endif
constexpr int BD_WIDE = NINT_BD;   // depth of the per-wave weight-fragment ring, 8-row tiles (two workgroups per CU)
constexpr int BD_NARROW = 2;       // 4-row tiles: three workgroups per CU hide more latency; measured -2.5 % vs depth 3
#ifndef NINT_AG
#define NINT_AG 2
#endif
#ifndef NINT_ABUF
#define NINT_ABUF 2
#endif

#ifdef NINT_STAMP
// Diagnostic build only (tools/clockprobe.py): per-workgroup phase stamps.  The values go to a buffer of
// their own that no kernel reads; the shipped library is built without NINT_STAMP.
#define NINT_STAMP_WGS 4096
__device__ unsigned long long g_stamp[NINT_STAMP_WGS * 8];
__device__ unsigned g_hwid[NINT_STAMP_WGS * 2];      // HW_ID (wave / SIMD / CU / SH / SE ids) and XCC_ID of each workgroup's first wave
#define NINT_STAMP_AT(slot)                                                              \
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < NINT_STAMP_WGS) {                      \
    g_stamp[blockIdx.x * 8 + 2 * (slot)] = __builtin_amdgcn_s_memtime();                 \
    g_stamp[blockIdx.x * 8 + 2 * (slot) + 1] = __builtin_amdgcn_s_memrealtime();         \
    if ((slot) == 0) {                                                                   \
      g_hwid[blockIdx.x * 2] = __builtin_amdgcn_s_getreg(4 | (31 << 11));                \
      g_hwid[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg(20 | (3 << 11));            \
    }                                                                                    \
  }
extern "C" int nint_debug_read_hwid(unsigned* host, int n_wgs) {
  if (!host || n_wgs <= 0 || n_wgs > NINT_STAMP_WGS) return NINT_E_ARG;
  NINT_CHECK_HIP(hipDeviceSynchronize());
  NINT_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_hwid), (size_t)n_wgs * 2 * sizeof(unsigned)));
  return NINT_OK;
}
extern "C" int nint_debug_read_stamps(unsigned long long* host, int n_wgs) {
  if (!host || n_wgs <= 0 || n_wgs > NINT_STAMP_WGS) return NINT_E_ARG;
  NINT_CHECK_HIP(hipDeviceSynchronize());
  NINT_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), (size_t)n_wgs * 8 * sizeof(unsigned long long)));
  return NINT_OK;
}
#else
#define NINT_STAMP_AT(slot)
#endif

// 4 consecutive elements kept PACKED until they are used (bf16: 2 registers instead of 4)
template <int DT> struct Pk4;
template <> struct Pk4<NINT_F32> {
  typedef f32x4_t T;
  static __device__ __forceinline__ T ld(const void* p, size_t i) { return *(const f32x4_t*)((const float*)p + i); }
  static __device__ __forceinline__ f32x4_t up(T v) { return v; }
  static __device__ __forceinline__ T zero() { return (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
};
template <> struct Pk4<NINT_BF16> {
  typedef u32x2_t T;
  static __device__ __forceinline__ T ld(const void* p, size_t i) { return *(const u32x2_t*)((const uint16_t*)p + i); }
  static __device__ __forceinline__ f32x4_t up(T w) {
    return (f32x4_t){__builtin_bit_cast(float, w[0] << 16), __builtin_bit_cast(float, w[0] & 0xffff0000u),
                     __builtin_bit_cast(float, w[1] << 16), __builtin_bit_cast(float, w[1] & 0xffff0000u)};
  }
  static __device__ __forceinline__ T zero() { return (u32x2_t){0u, 0u}; }
};

// Rounds of the K-slice exchange.  Four K-slices of a 4-row tile park 3/4 of the accumulators: in ONE round that
// buffer (12 KiB per wave, 48 KiB) is the workgroup's whole LDS footprint for the narrow layers and caps them at 3
// workgroups per CU; in two rounds of half the column tiles it is 24 KiB (under the halo image) and a fourth fits.
// Measured (same device, rocprofv3): the layer-2 gate kernel -6 %.  8-row tiles with four K-slices need four rounds to
// stay at two workgroups per CU (96 KiB otherwise).
constexpr int xchg_rounds(int EPI, int WK, int NTW, int MT) {
  return (WK == 4 && MT == 4 && NTW % 2 == 0) ? 2 : ((WK == 4 && MT == 8 && NTW == 4) ? 4 : 1);
}

// The workgroup's work as a device function of (launch arguments, LDS base, workgroup coordinates): conv_igemm_kernel runs
// it for blockIdx; conv_lstm_multi_kernel runs the gate launches of several layers (one wavefront step) in ONE grid.
template <int DT, int EPI, int WN, int WK, int NTW, int MT>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, char* smem, const int bx_, const int by_, const int nbx_) {
  static_assert(WN * WK == 4, "four waves per workgroup");
  constexpr int BD = MT >= 8 ? BD_WIDE : BD_NARROW;
  static_assert(EPI != EPI_LSTM || NTW % 4 == 0, "LSTM epilogue needs the 4 gate tiles in one wave");
  constexpr int NTH = 256;
  constexpr int NTWG = WN * NTW;   // n-tiles per workgroup
  constexpr int Q = MT / WK;       // rows whose epilogue this wave runs (K-slice waves split the rows)
  static_assert(MT % WK == 0, "K-slice waves split the tile rows evenly");

  const int tid = threadIdx.x, lane = tid & 63;
  NINT_STAMP_AT(0)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wk = wave / WN;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), each
  // with its own L2, so XCD x takes the CONTIGUOUS tile range x: neighbouring tiles (which share halo
  // pixels) and, at B = 8, whole images then stay inside one L2.  Any bijection is correct.
  int tile;
  {
    const int nb = nbx_, b = bx_, q8 = nb / 8, r8 = nb % 8, x = b % 8, i = b / 8;
    tile = x * q8 + (x < r8 ? x : r8) + i;     // ranges of q8 (+1 for the first r8 XCDs) tiles
  }
  // Two tile classes.  FULL tiles: MT rows x 16 pixels.  When the grid leaves 1..MT/2 rows over (100 = 12*8 + 4), the
  // leftover strip is covered by MERGED tiles: MT/2 rows x 32 pixels, i.e. the MT row tiles of a workgroup are the
  // rows of two adjacent 16-pixel columns.  No row tile computes padding rows, and at B = 8 the bench's gate launch
  // is 8 * (12*10 + 5) = 1000 workgroups instead of 1040: they fit the chip's 512 slots in two rounds, without the
  // 16-workgroup third round that kept 15 of 16 CUs idle for the last 30 us (tools/clockprobe.py).
  const bool mg = MT >= 8 && tile >= a.n_full;                 // workgroup-uniform
  int tx, ty, img;
  if (!mg) {
    tx = tile % a.tiles_x; tile /= a.tiles_x;
    ty = tile % a.tiles_full_y;
    img = tile / a.tiles_full_y;
  } else {
    tile -= a.n_full;
    tx = tile % a.tiles_x2;
    img = tile / a.tiles_x2;
    ty = a.tiles_full_y;
  }
  constexpr int RHM = MT >= 8 ? MT / 2 : MT;                   // rows of a merged tile
  const int nt0 = a.nt_begin + by_ * NTWG + wn * NTW;   // first n-tile of this wave
  const int y0 = ty * MT, x0 = mg ? tx * 32 : tx * 16;
  // row tile r of the workgroup sits at (y0 + rdy(r), x0 + rdx(r))
  auto rdy = [&](int r) __attribute__((always_inline)) { return mg ? r % RHM : r; };
  auto rdx = [&](int r) __attribute__((always_inline)) { return mg ? 16 * (r / RHM) : 0; };
  const int p = a.p, k = a.k, taps = a.taps;
  const int HWt = (mg ? 32 : 16) + 2 * p;      // halo tile width
  const int NHP = ((mg ? RHM : MT) + 2 * p) * HWt;   // halo tile pixels
  const int NHPp = mg ? a.nhp_pad2 : a.nhp_pad;      // padded to a multiple of 16 pixels (planes stay bank aligned)
  const unsigned magic_nhpp = mg ? a.magic_nhpp2 : a.magic_nhpp, magic_hwt = mg ? a.magic_hwt2 : a.magic_hwt;
  const int plane = NHPp * 16;                 // bytes of one g-plane
  const int chunk_bytes = 4 * plane;

  // accumulators start at the gate bias (LSTM epilogue; the tile is D[channel][pixel], so register r of n-tile j
  // is channel 4*(lane>>4)+r of column block nt0+j) -- only in K-slice 0, whose partial every row sums once
  f32x4_t acc[MT][NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    f32x4_t b0 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_LSTM) {
      if (wk == 0) b0 = *(const f32x4_t*)(a.bias + (nt0 + j) * 16 + 4 * (lane >> 4));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i][j] = b0;
  }

  const int nchunks = a.nchunk0 + a.nchunk1;
  const char* base0 = a.src0 + (long)img * a.img_stride0 +
                      ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride0;
  const char* base1 = a.src1 ? a.src1 + (long)img * a.img_stride1 +
                                   ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * a.pix_stride1
                             : nullptr;
  const int a_lane_off = (lane >> 4) * plane + (lane & 15) * 16;
  // local accumulator row i of K-slice wk is tile row (i + wk*Q) % MT: the rows a wave owns after the
  // K-slice exchange are then always its local rows 0..Q-1 (static register indexing)
  int rowoff[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) rowoff[i] = (rdy((i + wk * Q) % MT) * HWt + rdx((i + wk * Q) % MT)) * 16;
  const char* Bwave = a.Bp + (size_t)nt0 * 1024;   // wave-uniform (scalar) base; the lane part is a 32-bit offset
  const unsigned blane = lane * 16;
  const size_t bstep = (size_t)a.NTt * 1024;   // bytes between consecutive K-steps in Bp

  // c_{t-1} of the rows this wave finishes in the epilogue (rows (i + wk*Q) % MT, see the K-slice exchange below).
  // 4-row tiles (the narrow layers, whose time is all memory latency) fetch it HERE, ahead of the halo fill: the loads
  // complete under the fill wait instead of stalling the epilogue on an HBM round trip (Q*NTW/4 vectors, 4-8 VGPRs).
  // (The dgrad epilogue's read-modify-write operands stay late: held across the K loop they cost 20 registers and
  // with them the fourth workgroup per CU.)
  // 8-row tiles have no registers to spare and fetch it at the top of the epilogue, before the first store: vmcnt
  // retires in order and counts stores too, so a load issued between the epilogue's stores would make every row wait
  // for the previous row's stores to be acknowledged (measured: 22 us of epilogue per round).
  constexpr bool EARLY = MT <= 4;
  f32x4_t cpv[EPI == EPI_LSTM ? Q : 1][EPI == EPI_LSTM ? NTW / 4 : 1];
  auto load_cprev = [&]() __attribute__((always_inline)) {
    if constexpr (EPI == EPI_LSTM) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int r = (i + wk * Q) % MT;
        const int y = y0 + rdy(r), xq = x0 + rdx(r) + (lane & 15);
        const float* crow = a.c_prev + ((size_t)img * a.H + y) * a.W * a.Chp;     // wave-uniform row base
#pragma unroll
        for (int cb = 0; cb < NTW / 4; ++cb) {
          cpv[i][cb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (a.c_prev && y < a.H && xq < a.W)
            cpv[i][cb] = *(const f32x4_t*)(crow + (unsigned)(xq * a.Chp + (nt0 / 4 + cb) * 16 + 4 * (lane >> 4)));
        }
      }
    }
  };
  // the read-modify-write operands of the dgrad epilogue (x columns accumulate into the layer below's dh), same rule
  constexpr bool HOIST = EPI == EPI_DGRAD && Q * NTW <= 16;      // (register budget: the widest tiles keep the load inside the row loop)
  f32x4_t old[HOIST ? Q : 1][HOIST ? NTW : 1];
  auto load_old = [&]() __attribute__((always_inline)) {
    if constexpr (HOIST) {
      const bool rmw = a.out0 && !a.out0_overwrite;
      const int c4 = 4 * (lane >> 4);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int r = (i + wk * Q) % MT;
        const int y = y0 + rdy(r), x = x0 + rdx(r) + (lane & 15);
        const size_t rowpix = ((size_t)img * a.H + y) * a.W;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int n = (nt0 + j) * 16 + c4;
          old[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (rmw && n < a.C0p && y < a.H && x < a.W)
            old[i][j] = load_vec4<DT>(a.out0 + rowpix * a.C0p * Elem<DT>::ES, (unsigned)(x * a.C0p + n));
        }
      }
    }
  };
  if constexpr (EARLY) load_cprev();
  for (int c_begin = 0; c_begin < nchunks; c_begin += a.cpf) {
    const int c_cnt = min(a.cpf, nchunks - c_begin);
    if (c_begin > 0) __syncthreads();          // every wave is done reading the previous image
    // ---- stage the halo tile for chunks [c_begin, c_begin+c_cnt): global -> LDS ----
    const int a_units = c_cnt * 4 * NHPp;
    // LDS-DMA fill: one global_load_lds_dwordx4 per wave moves 64 consecutive 16-byte units of the image
    // (wave-uniform LDS base + lane*16; the SOURCE address is per lane), no VGPR round trip, so the whole
    // image is in flight at once instead of FB loads per thread.  Units of the pad pixels re-read pixel 0
    // (their LDS slots are never used); a_units is a multiple of 64.  The barrier below drains the DMA.
    for (int ub = wave * 64; ub < a_units; ub += 4 * 64) {
      const int u = ub + lane;
      // u / NHPp and hp / HWt by multiply-high with host-side magic numbers (exact for u < 65536, divisor <= 4096):
      // an integer division by a run-time value costs ~25 VALU instructions, and there are two per unit
      const int cq = (int)__umulhi((unsigned)u, magic_nhpp);   // cl*4 + q
      int hp = u - cq * NHPp;
      hp = hp < NHP ? hp : 0;
      const int q = cq & 3, cl = cq >> 2;
      const int hy = (int)__umulhi((unsigned)hp, magic_hwt);
      const int hx = hp - hy * HWt;
      const int c = c_begin + cl;
      const char* src = (c < a.nchunk0)
          ? base0 + ((long)hy * a.Wh + hx) * a.pix_stride0 + c * 64 + q * 16
          : base1 + ((long)hy * a.Wh + hx) * a.pix_stride1 + (c - a.nchunk0) * 64 + q * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + (size_t)ub * 16), 16, 0, 0);
    // (default cache policy: neighbouring tiles re-read the halo pixels; a non-temporal fill measured -8 % on the step)
    }
    // K-steps of this fill in two SEGMENTS: chunks of a horizontally folded source 0 (nint_layer.xfold: channel =
    // (kx, c), so only the k vertical taps of the halo tile's centre column remain: kx0 = 1) and all other chunks
    // (k x k taps).  Without folding segment 0 is empty and everything below is the one-segment loop.
    const int n0f = a.kx0 == k ? 0 : min(max(a.nchunk0 - c_begin, 0), c_cnt);   // folded chunks in this fill
    const int steps_before = min(c_begin, a.nchunk0) * k * a.kx0 + max(c_begin - a.nchunk0, 0) * taps;   // K-steps of chunks [0, c_begin)
    bool synced = false;
    for (int seg = 0; seg < 2; ++seg) {
    const int cs_lo = seg == 0 ? 0 : n0f, cs_n = seg == 0 ? n0f : c_cnt - n0f;
    if (cs_n == 0) continue;
    const int kx = seg == 0 ? a.kx0 : k;       // horizontal taps of this segment
    const int xoff = kx == k ? 0 : p;          // folded: the centre column
    const int tps = k * kx;
    // ---- this wave's K-slice of the segment: steps [s_lo, s_hi) of cs_n*tps ----
    const int nsteps = cs_n * tps;
    const int s_lo = (nsteps * wk) / WK, s_hi = (nsteps * (wk + 1)) / WK;
    int cl = s_lo / tps;
    int tap = s_lo - cl * tps;
    cl += cs_lo;
    int tyy = tap / kx;
    int txx = tap - tyy * kx;
    const char* Bs = Bwave + (size_t)(steps_before + (seg == 0 ? 0 : n0f * k * a.kx0) + s_lo) * bstep;
    // B ring: BD K-steps of weight fragments in flight per wave (L2 latency under load is several
    // K-steps long and only two waves share a SIMD, so one step of prefetch is not enough)
    // The ring loop is kept BRANCH-FREE (reloads are unconditional; past the slice's last step the pointer
    // stops advancing and the last step is re-read) so that the compiler can count outstanding loads and emits
    // s_waitcnt vmcnt((BD-1)*NTW) instead of draining the ring with vmcnt(0) at every step.
    u32x4_t bq[BD][NTW];
    int b_rem = s_hi - 1 - s_lo;                                    // steps the load pointer may still advance
    const char* Bn = Bs;                                            // next step to load (wave-uniform)
#define NINT_B_ADVANCE() { const bool adv_ = b_rem > 0; b_rem -= adv_ ? 1 : 0; Bn += adv_ ? bstep : 0; }
    if (s_lo < s_hi) {
#pragma unroll
      for (int d = 0; d < BD; ++d) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) bq[d][j] = *(const u32x4_t*)(Bn + blane + j * 1024);
        NINT_B_ADVANCE()
      }
    }
    if (!synced) {
      __syncthreads();                         // image visible to all waves
      synced = true;
      if (c_begin == 0) { NINT_STAMP_AT(1) }
    }
    // MT >= 8 (two workgroups per CU): the A fragments are software-pipelined in groups of AG rows through
    // two register groups -- while the AG*NTW MFMAs of one group run, the reads of the group after next
    // are in flight -- so that a wave alone on its SIMD keeps the matrix pipe busy.  (Left to itself the
    // compiler serialises read-wait-MFMA every two rows to save registers: 66 % MFMA occupancy for a lone
    // wave, which is what a workgroup sees whenever its CU partner is in its fill or epilogue.)
    // MT = 4 (three workgroups per CU, 168-register budget) keeps the plain loop.
    constexpr int AG = MT >= 8 ? NINT_AG : 0;
    if constexpr (AG > 0) {
      constexpr int NG = MT / AG;              // row groups per K-step
      constexpr int NBUF = NINT_ABUF;          // register groups; the reads run NBUF-1 groups ahead of the MFMAs
      constexpr int LA = NBUF - 1;
      static_assert(MT % AG == 0 && LA >= 1 && LA <= NG && (NG * BD) % NBUF == 0, "group rotation must close over the unrolled ring");
      u32x4_t ax[NBUF * AG];
      int va = (int)(cl * chunk_bytes + (tyy * HWt + txx + xoff) * 16) + a_lane_off;   // LDS byte offset of this step's fragment (row 0)
      const int d_row = (HWt - (kx - 1)) * 16;                                 // tap (ty, kx-1) -> (ty+1, 0)
      const int d_chunk = chunk_bytes - ((k - 1) * HWt + (kx - 1)) * 16;       // tap (k-1, kx-1) -> next chunk, tap (0, 0)
      if (s_lo < s_hi) {
#pragma unroll
        for (int i = 0; i < LA * AG; ++i) ax[i] = *(const u32x4_t*)(smem + va + rowoff[i]);
      }
      int s = s_lo;
      // K-step number D (static, position in the unrolled ring) on ring slot BQ; MORE = another step follows in
      // this slice (else the look-ahead re-reads this step)
#define NINT_K_STEP(BQ, MORE, D)                                                                           \
      {                                                                                                    \
        /* branch-free tap advance: the next step's base is this one plus one of three constants */       \
        const bool m_ = (MORE);                                                                            \
        const bool wx_ = txx + 1 == kx;                                                                    \
        const bool wy_ = wx_ && (tyy + 1 == k);                                                            \
        const int dl_ = wy_ ? d_chunk : (wx_ ? d_row : 16);                                                \
        const int vn = va + (m_ ? dl_ : 0);    /* (this step's again at the slice end) */                  \
        tyy = m_ ? (wy_ ? 0 : (wx_ ? tyy + 1 : tyy)) : tyy;                                                \
        txx = m_ ? (wx_ ? 0 : txx + 1) : txx;                                                              \
        _Pragma("unroll") for (int q = 0; q < NG; ++q) {                                                   \
          const int g_ = NG * (D) + q;         /* group counter inside the unrolled ring */                \
          const int ql = q + LA;               /* the group whose reads are issued now */                  \
          const int base = ql < NG ? va : vn;                                                              \
          const int row = AG * (ql % NG);                                                                  \
          const int bl = ((g_ + LA) % NBUF) * AG, bm = (g_ % NBUF) * AG;                                   \
          _Pragma("unroll") for (int r = 0; r < AG; ++r)                                                   \
            ax[bl + r] = *(const u32x4_t*)(smem + base + rowoff[row + r]);                                 \
          _Pragma("unroll") for (int r = 0; r < AG; ++r)                                                   \
            _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                \
              acc[AG * q + r][j] = mma_step<DT>(BQ[j], ax[bm + r], acc[AG * q + r][j]);                    \
          __builtin_amdgcn_sched_group_barrier(0x100, AG, 0);                                              \
          __builtin_amdgcn_sched_group_barrier(0x008, AG * NTW * (DT == NINT_BF16 ? 1 : 4), 0);            \
        }                                                                                                  \
        va = vn;                                                                                           \
      }
      for (; s + BD <= s_hi; s += BD) {
#pragma unroll
        for (int d = 0; d < BD; ++d) {
          NINT_K_STEP(bq[d], d + 1 < BD || s + BD < s_hi, d)
#pragma unroll
          for (int j = 0; j < NTW; ++j) bq[d][j] = *(const u32x4_t*)(Bn + blane + j * 1024);
          NINT_B_ADVANCE()
          __builtin_amdgcn_sched_group_barrier(0x020, NTW, 0);   // this slot's reloads go out before the next step starts
        }
      }
      // remainder (< BD steps): ring slots 0.. already hold exactly these steps
#pragma unroll
      for (int d = 0; d < BD - 1; ++d) {
        if (s + d < s_hi) NINT_K_STEP(bq[d], s + d + 1 < s_hi, d)
      }
#undef NINT_K_STEP
    } else {
  int s = s_lo;
      for (; s + BD <= s_hi; s += BD) {
#pragma unroll
        for (int d = 0; d < BD; ++d) {
          const char* Ab = smem + (size_t)cl * chunk_bytes + (tyy * HWt + txx + xoff) * 16 + a_lane_off;
          u32x4_t af[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = *(const u32x4_t*)(Ab + rowoff[i]);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<DT>(bq[d][j], af[i], acc[i][j]);   // swapped: D[channel][pixel]
#pragma unroll
          for (int j = 0; j < NTW; ++j) bq[d][j] = *(const u32x4_t*)(Bn + blane + j * 1024);
          NINT_B_ADVANCE()
          if (++txx == kx) { txx = 0; if (++tyy == k) { tyy = 0; ++cl; } }
        }
      }
      // remainder (< BD steps): ring slots 0.. already hold exactly these steps
#pragma unroll
      for (int d = 0; d < BD - 1; ++d) {
        if (s + d < s_hi) {
          const char* Ab = smem + (size_t)cl * chunk_bytes + (tyy * HWt + txx + xoff) * 16 + a_lane_off;
          u32x4_t af[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = *(const u32x4_t*)(Ab + rowoff[i]);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = mma_step<DT>(bq[d][j], af[i], acc[i][j]);   // swapped: D[channel][pixel]
          if (++txx == kx) { txx = 0; if (++tyy == k) { tyy = 0; ++cl; } }
        }
      }
    }
    }   // segment
  }

#undef NINT_B_ADVANCE
  NINT_STAMP_AT(2)
  // ------------------------------------------------------------------ K-slice reduction
  // Every wave OWNS Q = MT/WK of the rows: its local accumulators 0..Q-1 (local row ii of slice
  // wk is tile row (ii + wk*Q) % MT, see rowoff above), so the epilogue is spread over all four
  // waves.  One exchange through LDS: each wave parks its MT-Q foreign rows (tile-major,
  // lane-linear 1 KiB tiles), barrier, each wave adds the WK-1 foreign partials of its own rows.
  if constexpr (WK > 1) {
    constexpr int FR = MT - Q;                 // foreign rows per wave
    constexpr int XR = xchg_rounds(EPI, WK, NTW, MT), NTX = NTW / XR;   // column tiles exchanged per round
    __syncthreads();                           // every wave is done reading the A image
#pragma unroll
    for (int xr = 0; xr < XR; ++xr) {
      if (xr > 0) __syncthreads();             // the previous round's partials are consumed
      char* mine = smem + (size_t)((wn * WK + wk) * FR * NTX) * 1024 + lane * 16;
#pragma unroll
      for (int ii = Q; ii < MT; ++ii)
#pragma unroll
        for (int j = 0; j < NTX; ++j) *(f32x4_t*)(mine + ((ii - Q) * NTX + j) * 1024) = acc[ii][xr * NTX + j];
      __syncthreads();
#pragma unroll
      for (int d = 1; d < WK; ++d) {
        const int src = (wk + d) % WK;         // slice whose partials are added now
        // my local row ii is tile row (ii + wk*Q) % MT = local row (ii + (wk-src)*Q) mod MT of `src`
        const int shift = ((wk - src + WK) % WK) * Q;      // in [Q, MT): always a foreign row there
        const char* theirs = smem + (size_t)((wn * WK + src) * FR * NTX) * 1024 + lane * 16;
#pragma unroll
        for (int ii = 0; ii < Q; ++ii)
#pragma unroll
          for (int j = 0; j < NTX; ++j)
            acc[ii][xr * NTX + j] += *(const f32x4_t*)(theirs + ((ii + shift - Q) * NTX + j) * 1024);
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  // Operands are fed SWAPPED to the MFMA (A := weight fragment, B := pixel fragment), so the
  // 16x16 result tile is D[channel][pixel]: column = lane&15 = pixel, row = 4*(lane>>4)+reg =
  // channel.  A lane therefore owns 4 CONSECUTIVE channels of one pixel and every epilogue
  // access is a 16-byte (f32) / 8-byte (bf16) vector on channels-last memory.
  // c_{t-1} of the rows this wave finishes in the epilogue is fetched up front, before the first store: vmcnt
  // retires in order and counts stores too, so a load issued between the epilogue's stores would make every
  // row wait for the previous row's stores to be acknowledged (measured: 22 us of epilogue per round).
  if constexpr (!EARLY) load_cprev();
  load_old();
  const int px = lane & 15;
  const int c4 = 4 * (lane >> 4);
  const int x = x0 + px;
  if constexpr (EPI == EPI_LSTM) {
#pragma unroll
    for (int cb = 0; cb < NTW / 4; ++cb) {
      const int cblock = nt0 / 4 + cb;
      const int ch = cblock * 16 + c4;
      // addresses = wave-uniform row base (scalar arithmetic) + a 32-bit lane offset that is the same for every row
      const int Gc = 4 * a.Ch16;
      const int odd = (lane >> 4) & 1, chb = (lane >> 5) * 8;   // (bf16 gate stash: see below)
      const unsigned lo_c = (unsigned)(x * a.Chp + ch);
      const unsigned lo_h = (unsigned)((x + a.P) * a.Chp + ch);
      const unsigned lo_g = (unsigned)(x * Gc + cblock * 64) + (DT == NINT_BF16 ? (unsigned)(chb + (odd ? 32 : 0)) : (unsigned)c4);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const int rt = (i + wk * Q) % MT;        // row tile of the workgroup
        const int y = y0 + rdy(rt), xo = rdx(rt);
        const bool ok = y < a.H && x + xo < a.W;   // (the lane exchange below needs every lane: no divergent block)
        const size_t rowpix = ((size_t)img * a.H + y) * a.W + xo;   // (a merged tile's second column: 16 pixels on)
        const f32x4_t cp = cpv[i][cb];
        f32x4_t gi, gf, gg, go, cn, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gi[r] = sigmoidf_(acc[i][cb * 4 + 0][r]);      // (bias already in the accumulator)
          gf[r] = sigmoidf_(acc[i][cb * 4 + 1][r]);
          gg[r] = tanhf_(acc[i][cb * 4 + 2][r]);
          go[r] = sigmoidf_(acc[i][cb * 4 + 3][r]);
          cn[r] = fmaf(cp[r], gf[r], gi[r] * gg[r]);   // model.py:228 (the association is pinned: every kernel family agrees bit for bit)
          hn[r] = go[r] * tanhf_(cn[r]);               // model.py:229
        }
        if (ok) {
          *(f32x4_t*)(a.c_out + rowpix * a.Chp + lo_c) = cn;
          char* hrow = a.h_out + (((size_t)img * a.Hh + (y + a.P)) * a.Wh + xo) * a.Chp * Elem<DT>::ES;
          store_vec4<DT>(hrow, lo_h, hn);
        }
        if (a.gates_out) {
          if constexpr (DT == NINT_BF16) {
            // The epilogue is store-ISSUE-bound (8-byte stores): lane rows 2r and 2r+1 (channel quads 8r..8r+3 and
            // 8r+4..8r+7 of the same pixel) trade halves with v_permlane16_swap so that the even row stores gates
            // i and f, the odd row g and o, each as ONE 16-byte vector of 8 channels: 2 stores instead of 4.
            auto pk = [](float lo, float hi) __attribute__((always_inline)) { return pack_bf16x2(lo, hi); };
            typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
            const u2_t ig0 = __builtin_amdgcn_permlane16_swap(pk(gi[0], gi[1]), pk(gg[0], gg[1]), false, false);
            const u2_t ig1 = __builtin_amdgcn_permlane16_swap(pk(gi[2], gi[3]), pk(gg[2], gg[3]), false, false);
            const u2_t fo0 = __builtin_amdgcn_permlane16_swap(pk(gf[0], gf[1]), pk(go[0], go[1]), false, false);
            const u2_t fo1 = __builtin_amdgcn_permlane16_swap(pk(gf[2], gf[3]), pk(go[2], go[3]), false, false);
            // even row: (own, partner) of the first operand = gate i / f; odd row: (partner, own) of the second = g / o
            if (ok) {
              uint16_t* grow = (uint16_t*)a.gates_out + rowpix * Gc;
              // NON-TEMPORAL: the stash (512 of the epilogue's 896 bytes per pixel) is next read in the backward pass; written
              // through the caches it evicts the weights / halo pixels / h that the following launches re-read
              // (measured in the bench step, same device, alternating: 946.8 / 953.5 -> 973.1 / 968.8 samples/s)
              __builtin_nontemporal_store((u32x4_t){ig0[0], ig1[0], ig0[1], ig1[1]}, (u32x4_t*)(grow + lo_g));        // gate i (even row) / g (odd row)
              __builtin_nontemporal_store((u32x4_t){fo0[0], fo1[0], fo0[1], fo1[1]}, (u32x4_t*)(grow + lo_g + 16));   // gate f / o
            }
          } else if (ok) {
            char* grow = a.gates_out + rowpix * Gc * Elem<DT>::ES;
            store_vec4<DT>(grow, lo_g + 0, gi);
            store_vec4<DT>(grow, lo_g + 16, gf);
            store_vec4<DT>(grow, lo_g + 32, gg);
            store_vec4<DT>(grow, lo_g + 48, go);
          }
        }
      }
    }
  } else if constexpr (EPI == EPI_DGRAD_PW) {
    // Fused BPTT step.  This launch is dgrad(t+1) of the layer; its h columns are d/dh_t from the layer's own
    // recurrence -- the last contribution to dh_t (the layer above already left its x columns of time t in pw_old) --
    // so the pointwise backward of time t runs HERE, on the accumulators: dh_t never goes to memory, and the
    // HBM-bound pass (34 bytes per channel-pixel) hides behind the matrix work of the CU's other workgroup.
    //   dct = dc + dh*o*(1 - tanh(c_t)^2);  dG = (dct*g*i(1-i), dct*c_{t-1}*f(1-f), dct*i*(1-g^2), dh*tanh(c_t)*o(1-o))
    //   dc <- dct*f                                                     (autograd of model.py:222-229)
    // x columns are stored, or accumulated onto what a classic layer below keeps in its dh buffer (out0_overwrite).
    // All loads of a batch of rows are issued before its first store (vmcnt retires in order and counts stores).
    // With lo_gates set the x columns get the same treatment: they are the LAST contribution to d/dh of the layer BELOW at
    // this launch's own time step (that layer's h columns of the next time step are already in out0), so its pointwise
    // backward runs here too and its stand-alone launch disappears.
    typedef Pk4<DT> PK;
    constexpr int RB0 = NTW >= 4 ? 1 : 4 / NTW;                  // rows per batch: about 4 column tiles (88 registers) in flight
    constexpr int RB = RB0 >= Q ? Q : (Q % RB0 == 0 ? RB0 : 1);
    // per column tile (wave-uniform): 0 = x columns stored / accumulated, 1 = pointwise backward of this layer (h columns),
    // 2 = pointwise backward of the layer below (x columns), -1 = padding
    auto tile_kind = [&](int j, int& ch0) __attribute__((always_inline)) {
      const int n0 = (nt0 + j) * 16;
      if (n0 >= a.C0p) {
        ch0 = n0 - a.C0p;
        if (!a.pw_gates) return (a.out1 && ch0 < a.C1p) ? 3 : -1;      // 3 = h columns stored (a classic layer's dgrad)
        return ch0 < a.Ch16 ? 1 : -1;
      }
      ch0 = n0;
      return a.lo_gates ? (ch0 < a.lo_Ch16 ? 2 : -1) : 0;
    };
#pragma unroll
    for (int ib = 0; ib < Q; ib += RB) {
      typename PK::T gq[RB][NTW][4], oldv[RB][NTW];
      f32x4_t cpq[RB][NTW], cnq[RB][NTW], dcq[RB][NTW];
#pragma unroll
      for (int ii = 0; ii < RB; ++ii) {
        const int rt = (ib + ii + wk * Q) % MT;
        const int y = y0 + rdy(rt), xo = rdx(rt);
        const bool okp = y < a.H && x + xo < a.W;
        const size_t rowpix = ((size_t)img * a.H + y) * a.W + xo;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          int ch0;
          const int kind = tile_kind(j, ch0);
#pragma unroll
          for (int q = 0; q < 4; ++q) gq[ii][j][q] = PK::zero();
          oldv[ii][j] = PK::zero();
          cpq[ii][j] = cnq[ii][j] = dcq[ii][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          if (!okp || kind < 0) continue;
          const size_t pix = rowpix + x;
          if (kind == 3) continue;
          if (kind != 1) {                                         // x columns: what a classic layer below keeps in its dh buffer
            if (a.out0 && !a.out0_overwrite) oldv[ii][j] = PK::ld(a.out0, pix * a.C0p + ch0 + c4);
          } else if (a.pw_old) {
            oldv[ii][j] = PK::ld(a.pw_old, pix * a.Chp + ch0 + c4);
          }
          if (kind == 0 || kind == 3) continue;
          const bool lo = kind == 2;
          const int Gc = 4 * (lo ? a.lo_Ch16 : a.Ch16), Cp = lo ? a.C0p : a.Chp;
          const char* gp = lo ? a.lo_gates : a.pw_gates;
          const float* cpp = lo ? a.lo_c_prev : a.pw_c_prev;
          const float* cnp = lo ? a.lo_c_new : a.pw_c_new;
          const float* dcp_ = lo ? (a.lo_dc_zero ? nullptr : a.lo_dc) : a.pw_dc;
          const size_t gb = pix * Gc + (size_t)(ch0 >> 4) * 64 + c4;
#pragma unroll
          for (int q = 0; q < 4; ++q) gq[ii][j][q] = PK::ld(gp, gb + 16 * q);
          const size_t ci = pix * Cp + ch0 + c4;
          if (cpp) cpq[ii][j] = *(const f32x4_t*)(cpp + ci);
          cnq[ii][j] = *(const f32x4_t*)(cnp + ci);
          if (dcp_) dcq[ii][j] = *(const f32x4_t*)(dcp_ + ci);
        }
      }
#pragma unroll
      for (int ii = 0; ii < RB; ++ii) {
        const int i = ib + ii;
        const int rt = (i + wk * Q) % MT;
        const int y = y0 + rdy(rt), xo = rdx(rt);
        const bool okp = y < a.H && x + xo < a.W;
        const size_t rowpix = ((size_t)img * a.H + y) * a.W + xo;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          int ch0;
          const int kind = tile_kind(j, ch0);
          if (!okp || kind < 0) continue;
          const f32x4_t dhv = acc[i][j] + PK::up(oldv[ii][j]);
          if (kind == 0) {
            if (a.out0) store_vec4<DT>(a.out0 + rowpix * a.C0p * Elem<DT>::ES, (unsigned)(x * a.C0p + ch0 + c4), dhv);
            continue;
          }
          if (kind == 3) {
            store_vec4<DT>(a.out1 + rowpix * a.C1p * Elem<DT>::ES, (unsigned)(x * a.C1p + ch0 + c4), acc[i][j]);
            continue;
          }
          const bool lo = kind == 2;
          const int Gc = 4 * (lo ? a.lo_Ch16 : a.Ch16), Cp = lo ? a.C0p : a.Chp;
          const f32x4_t gi = PK::up(gq[ii][j][0]), gf = PK::up(gq[ii][j][1]), gg = PK::up(gq[ii][j][2]), go = PK::up(gq[ii][j][3]);
          const f32x4_t cp = cpq[ii][j], cn = cnq[ii][j], dcv = dcq[ii][j];
          f32x4_t o_i, o_f, o_g, o_o, dcp;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float tc = tanhf_(cn[e]);
            const float dct = dcv[e] + dhv[e] * go[e] * (1.f - tc * tc);
            const float d_o = dhv[e] * tc;
            o_i[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
            o_f[e] = dct * cp[e] * gf[e] * (1.f - gf[e]);
            o_g[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
            o_o[e] = d_o * go[e] * (1.f - go[e]);
            dcp[e] = dct * gf[e];
          }
          char* grow = (lo ? a.lo_dG : a.pw_dG) + ((((size_t)img * a.Hh) + (y + a.P)) * a.Wh + (xo + a.P)) * Gc * Elem<DT>::ES;
          const unsigned ob = (unsigned)(x * Gc + (ch0 >> 4) * 64 + c4);
          store_vec4<DT>(grow, ob, o_i);
          store_vec4<DT>(grow, ob + 16, o_f);
          store_vec4<DT>(grow, ob + 32, o_g);
          store_vec4<DT>(grow, ob + 48, o_o);
          *(f32x4_t*)((lo ? a.lo_dc : a.pw_dc) + (rowpix + x) * Cp + ch0 + c4) = dcp;
        }
      }
    }
  } else {
    // x columns accumulate into (or overwrite) the layer below's dh / dx, h columns are stored to dh_prev.
    // Addresses = wave-uniform row base + row-invariant lane offset; the read-modify-write loads of ALL rows are
    // issued before the first store (vmcnt retires in order and counts stores: a load between stores would wait
    // for the previous row's stores to be acknowledged).
    const bool rmw = a.out0 && !a.out0_overwrite;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const int rt = (i + wk * Q) % MT;
      const int y = y0 + rdy(rt), xo = rdx(rt);
      if (y < a.H && x + xo < a.W) {
        const size_t rowpix = ((size_t)img * a.H + y) * a.W + xo;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int n = (nt0 + j) * 16 + c4;
          if (n < a.C0p) {
            if (a.out0) {
              char* row0 = a.out0 + rowpix * a.C0p * Elem<DT>::ES;
              const unsigned o = (unsigned)(x * a.C0p + n);
              f32x4_t prev;
              if constexpr (HOIST) prev = old[i][j];
              else prev = rmw ? load_vec4<DT>(row0, o) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
              store_vec4<DT>(row0, o, prev + acc[i][j]);
            }
          } else if (a.out1) {
            store_vec4<DT>(a.out1 + rowpix * a.C1p * Elem<DT>::ES, (unsigned)(x * a.C1p + (n - a.C0p)), acc[i][j]);
          }
        }
      }
    }
  }
  NINT_STAMP_AT(3)
}

template <int DT, int EPI, int WN, int WK, int NTW, int MT>
__global__ __launch_bounds__(256, MT >= 8 ? 2 : (xchg_rounds(EPI, WK, NTW, MT) == 2 && DT == NINT_BF16 && EPI != EPI_DGRAD_PW ? 4 : 3)) void conv_igemm_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv_igemm_body<DT, EPI, WN, WK, NTW, MT>(a, smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// Independent launches in ONE grid.  Forward: one wavefront step -- gate(0, t+1), gate(1, t), gate(2, t-1), ... are
// independent (model.py:265-271); backward: the bottom layer's dgrad of time u and the top layer's fused step of time u-1
// (adjacent in the BPTT order, and they share no buffer in a stack of three or more layers).  At small batches (the
// strong-scaling shape) none of these fills the chip -- run back to back each pays its own fill / epilogue latency on
// half-empty CUs.  Workgroup ranges [begin[i], begin[i+1]) belong to problem i (each range starts at a multiple of 8, so a
// problem's tile -> XCD mapping is the one of its own launch); every problem runs the body of the 4-row-tile shape its own
// launch would have taken (ConvPlan::variant): same instructions on the same data, bit-identical results.  Registers / LDS
// are those of the widest variant in the kernel.
struct ConvMulti {
  ConvArgs a[NINT_MULTI_MAX];
  int n;
  int begin[NINT_MULTI_MAX + 1];     // first workgroup of each problem (multiples of 8); begin[n] = grid size
  int nbx[NINT_MULTI_MAX];           // pixel-tile workgroups of each problem (its launch's gridDim.x); column group = (b - begin) / nbx
  int nwg[NINT_MULTI_MAX];           // nbx * column groups: workgroups past it are padding
  int variant[NINT_MULTI_MAX];
  PwArgs pw;                         // conv_bwd_multi_kernel only: workgroups [begin[n], begin[n] + pw_blocks) run a pointwise LSTM backward pass
  int pw_blocks;
};
constexpr int conv_variant(int EPI, int WN, int WK, int NTW, int MT) { return EPI * 10000 + WN * 1000 + WK * 100 + NTW * 10 + MT / 4; }
#define NINT_MULTI_PROLOGUE                                                                                     \
  extern __shared__ __attribute__((aligned(16))) char smem[];                                                   \
  int i = 0;                                                                                                    \
  _Pragma("unroll") for (int q = 1; q < NINT_MULTI_MAX; ++q) i += (q < m.n && (int)blockIdx.x >= m.begin[q]) ? 1 : 0; \
  const int b = blockIdx.x - m.begin[i];                                                                        \
  if (b >= m.nwg[i]) return;                   /* (padding up to the next multiple of 8) */                     \
  const int nbx = m.nbx[i], by = b / nbx, bx = b - by * nbx;                                                    \
  const ConvArgs& a = m.a[i];
#define NINT_MULTI_CASE(EPI_, WN_, WK_, NTW_) \
  case conv_variant(EPI_, WN_, WK_, NTW_, 4): conv_igemm_body<DT, EPI_, WN_, WK_, NTW_, 4>(a, smem, bx, by, nbx); break;
template <int DT>
__global__ __launch_bounds__(256, 3) void conv_lstm_multi_kernel(ConvMulti m) {
  NINT_MULTI_PROLOGUE
  switch (m.variant[i]) {
    NINT_MULTI_CASE(EPI_LSTM, 4, 1, 4)
    NINT_MULTI_CASE(EPI_LSTM, 2, 2, 4)
    NINT_MULTI_CASE(EPI_LSTM, 1, 4, 4)
    default: break;
  }
}
// ... with the first layer on its 8-row tiles (mid-size batches; 256 registers: two workgroups per CU for every problem of
// the grid).  B = 4 at 100 x 154: 966 -> 989 samples/s, steady; B = 8: bimodal from process to process (1021-1032 or 995-1000
// against a steady 1022-1025), so the engine's rule stops below it.
#define NINT_MULTI_CASE8(EPI_, WN_, WK_, NTW_) \
  case conv_variant(EPI_, WN_, WK_, NTW_, 8): conv_igemm_body<DT, EPI_, WN_, WK_, NTW_, 8>(a, smem, bx, by, nbx); break;
template <int DT>
__global__ __launch_bounds__(256, 2) void conv_lstm_multi8_kernel(ConvMulti m) {
  NINT_MULTI_PROLOGUE
  switch (m.variant[i]) {
    NINT_MULTI_CASE8(EPI_LSTM, 4, 1, 4)
    NINT_MULTI_CASE8(EPI_LSTM, 2, 2, 4)
    NINT_MULTI_CASE8(EPI_LSTM, 1, 4, 4)
    NINT_MULTI_CASE(EPI_LSTM, 4, 1, 4)
    NINT_MULTI_CASE(EPI_LSTM, 2, 2, 4)
    NINT_MULTI_CASE(EPI_LSTM, 1, 4, 4)
    default: break;
  }
}
template <int DT>
__global__ __launch_bounds__(256, 3) void conv_bwd_multi_kernel(ConvMulti m) {
  if (m.pw_blocks && (int)blockIdx.x >= m.begin[m.n]) {      // (nint_seq.wave = 4: the bottom layer's pointwise backward behind the top layer's fused step)
#ifndef NINT_PW_U
#define NINT_PW_U 2
#endif
    lstm_bwd_pointwise_body<DT, NINT_PW_U>(m.pw, blockIdx.x - m.begin[m.n], m.pw_blocks);
    return;
  }
  NINT_MULTI_PROLOGUE
  switch (m.variant[i]) {
    NINT_MULTI_CASE(EPI_DGRAD, 1, 4, 4)
    NINT_MULTI_CASE(EPI_DGRAD, 1, 4, 2)
    NINT_MULTI_CASE(EPI_DGRAD, 2, 2, 3)
    NINT_MULTI_CASE(EPI_DGRAD, 1, 4, 3)
    NINT_MULTI_CASE(EPI_DGRAD_PW, 1, 4, 3)
    default: break;                            // (the register-heavy fused shapes -- 4 column tiles per wave -- would spill at 168 VGPRs)
  }
}
template <int DT>
__global__ __launch_bounds__(256, 2) void conv_bwd_multi8_kernel(ConvMulti m) {
  NINT_MULTI_PROLOGUE
  switch (m.variant[i]) {
    NINT_MULTI_CASE8(EPI_DGRAD, 1, 4, 4)
    NINT_MULTI_CASE8(EPI_DGRAD_PW, 1, 4, 3)
    NINT_MULTI_CASE(EPI_DGRAD, 1, 4, 4)
    NINT_MULTI_CASE(EPI_DGRAD, 1, 4, 2)
    NINT_MULTI_CASE(EPI_DGRAD_PW, 1, 4, 3)
    default: break;
  }
}
// the dgrad launches of two neighbouring layers (nint_seq.wave = 4: the bottom layer's of time u+1 with the next layer's of time
// u), both on 8-row tiles.  A kernel of its own: as two more cases of the kernel above the register allocation of ALL its
// bodies collapsed (1036 spilled registers).
template <int DT>
__global__ __launch_bounds__(256, 2) void conv_dgrad_multi8_kernel(ConvMulti m) {
  NINT_MULTI_PROLOGUE
  switch (m.variant[i]) {
    NINT_MULTI_CASE8(EPI_DGRAD, 1, 4, 4)
    NINT_MULTI_CASE8(EPI_DGRAD, 2, 2, 3)
    default: break;
  }
}
static bool multi_holds(int variant) {
  switch (variant) {
    case conv_variant(EPI_LSTM, 4, 1, 4, 8): case conv_variant(EPI_DGRAD, 1, 4, 4, 8):      // (the *_multi8 kernels)
    case conv_variant(EPI_LSTM, 2, 2, 4, 8): case conv_variant(EPI_LSTM, 1, 4, 4, 8): case conv_variant(EPI_DGRAD_PW, 1, 4, 3, 8):
    case conv_variant(EPI_LSTM, 4, 1, 4, 4): case conv_variant(EPI_LSTM, 2, 2, 4, 4): case conv_variant(EPI_LSTM, 1, 4, 4, 4):
    case conv_variant(EPI_DGRAD, 1, 4, 4, 4): case conv_variant(EPI_DGRAD, 1, 4, 2, 4): case conv_variant(EPI_DGRAD_PW, 1, 4, 3, 4):
    case conv_variant(EPI_DGRAD, 2, 2, 3, 8): case conv_variant(EPI_DGRAD, 2, 2, 3, 4):      // (8-row: conv_dgrad_multi8_kernel)
    case conv_variant(EPI_DGRAD, 1, 4, 3, 4):
      return true;
    default: return false;
  }
}

// ------------------------------------------------------------------------------ host side
// (ConvPlan, nint_common.h: a launch that is planned but not enqueued; the sequence drivers collect independent ones and
// enqueue them as one grid)
template <int DT, int EPI, int WN, int WK, int NTW, int MT>
static int launch_cfg(ConvArgs& a, int N, int ngroups_y, hipStream_t st, ConvPlan* plan = nullptr) {
  const int NHP = (MT + 2 * a.p) * (16 + 2 * a.p);
  a.nhp_pad = nint_round_up(NHP, 16);
  a.magic_nhpp = (unsigned)(((1ull << 32) + a.nhp_pad - 1) / a.nhp_pad);
  a.magic_hwt = (unsigned)(((1ull << 32) + (16 + 2 * a.p) - 1) / (16 + 2 * a.p));
  a.tiles_x = nint_cdiv(a.W, 16);
  a.tiles_y = nint_cdiv(a.H, MT);
  // leftover strip of 1..MT/2 rows (8-row tiles): merged tiles of MT/2 rows x 32 pixels instead of half-empty ones
  const int left = a.H % MT;
  const bool merge = MT >= 8 && left >= 1 && left <= MT / 2 && a.tiles_x >= 2;
  a.tiles_full_y = merge ? a.H / MT : a.tiles_y;
  a.tiles_x2 = merge ? nint_cdiv(a.tiles_x, 2) : 0;
  a.n_full = N * a.tiles_x * a.tiles_full_y;
  int nhp_max = a.nhp_pad;
  if (merge) {
    const int NHP2 = (MT / 2 + 2 * a.p) * (32 + 2 * a.p);
    a.nhp_pad2 = nint_round_up(NHP2, 16);
    a.magic_nhpp2 = (unsigned)(((1ull << 32) + a.nhp_pad2 - 1) / a.nhp_pad2);
    a.magic_hwt2 = (unsigned)(((1ull << 32) + (32 + 2 * a.p) - 1) / (32 + 2 * a.p));
    if (a.nhp_pad2 > nhp_max) nhp_max = a.nhp_pad2;
  }
  const int chunk_bytes = 4 * nhp_max * 16;    // (the larger of the two tile images)
  const int nchunks = a.nchunk0 + a.nchunk1;
  const int red_bytes = WK > 1 ? WN * WK * (MT - MT / WK) * (NTW / xchg_rounds(EPI, WK, NTW, MT)) * 1024 : 0;   // K-slice exchange buffer
  // as many channel chunks per fill as fit in ~72 KiB (two workgroups per CU stay resident)
  int cpf = (72 * 1024) / chunk_bytes;
  if (cpf < 1) cpf = 1;
  if (cpf > nchunks) cpf = nchunks;
  // even out the fills (e.g. 5 chunks with room for 4 -> 3 + 2)
  const int nfill = nint_cdiv(nchunks, cpf);
  cpf = nint_cdiv(nchunks, nfill);
  a.cpf = cpf;
  a.a_bytes = cpf * chunk_bytes;
  size_t lds = (size_t)a.a_bytes;
  if ((size_t)red_bytes > lds) lds = red_bytes;
  if (lds > 160 * 1024) return NINT_E_LDS;
  if (plan) {
    plan->a = a; plan->gx = a.n_full + N * a.tiles_x2; plan->gy = ngroups_y; plan->lds = lds;
    plan->variant = conv_variant(EPI, WN, WK, NTW, MT);
    return NINT_OK;
  }
  auto kern = conv_igemm_kernel<DT, EPI, WN, WK, NTW, MT>;
  if (lds > 64 * 1024)
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(a.n_full + N * a.tiles_x2, ngroups_y), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

#ifndef NINT_SPLIT_NUM
#define NINT_SPLIT_NUM 3        // small batches: split the columns while workgroups < NINT_SPLIT_NUM / 2 per CU
#endif
template <int DT, int EPI>
static int launch_conv(ConvArgs& a, int N, int ntiles, hipStream_t st, ConvPlan* plan = nullptr) {
  if (ntiles <= 0) return NINT_OK;             // (a plan keeps gx == 0)
  // short-K launches (narrow layers) take 4-row tiles; nint_layer.tile_rows = 4 | 8 overrides (tests run both
  // heights on every shape)
  const int ksteps = a.nchunk0 * a.k * a.kx0 + a.nchunk1 * a.taps;
  // (measured with the leftover strip merged, B = 8, 100x154: dgrad layer 0 -- 200 steps, 4 column tiles -- 94.7 us with
  // 8-row tiles against 104.0; dgrad layer 1 -- 36 steps -- 40.2 against 41.5; the 18-27-step launches prefer 4 rows)
  bool mt4 = a.tile_rows ? a.tile_rows == 4
                         : (EPI != EPI_LSTM ? ksteps <= 32 : (ksteps <= 48 || ntiles <= 4));
  int n_cu = 256;
  { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); }
  // few images (B = 1 or 2 per GPU): 8-row tiles would leave CUs without a workgroup; 4-row tiles double the count
  if (!a.tile_rows && 2 * N * nint_cdiv(a.W, 16) * nint_cdiv(a.H, 8) < 3 * n_cu) mt4 = true;
  if constexpr (EPI == EPI_LSTM) {
    if (ntiles % 4) return NINT_E_SHAPE;
    const int cbs = ntiles / 4;
    // Small batches (the strong-scaling shape: B = 1 per GPU is 125 8-row tiles for 256 CUs): when the pixel tiles alone
    // leave CUs empty, the gate columns are split over more workgroups (column groups on blockIdx.y: the waves that no
    // longer have columns of their own slice K instead) -- each then re-stages the halo tile, which an idle CU does for free.
    const int ptiles = N * nint_cdiv(a.W, 16) * nint_cdiv(a.H, mt4 ? 4 : 8);
    // (an explicit nint_layer.tile_rows pins the launch shape: no split)
    const bool few4 = !a.tile_rows && 2 * ptiles * (cbs / 4 > 0 ? cbs / 4 : 1) < NINT_SPLIT_NUM * n_cu;       // with 4 column blocks per workgroup
    const bool few2 = !a.tile_rows && 2 * ptiles * (cbs / 2 > 0 ? cbs / 2 : 1) < NINT_SPLIT_NUM * n_cu;       // with 2
    if (cbs % 4 == 0 && !few4) return mt4 ? launch_cfg<DT, EPI, 4, 1, 4, 4>(a, N, cbs / 4, st, plan) : launch_cfg<DT, EPI, 4, 1, 4, 8>(a, N, cbs / 4, st, plan);
    if (cbs % 2 == 0 && !few2) return mt4 ? launch_cfg<DT, EPI, 2, 2, 4, 4>(a, N, cbs / 2, st, plan) : launch_cfg<DT, EPI, 2, 2, 4, 8>(a, N, cbs / 2, st, plan);
    return mt4 ? launch_cfg<DT, EPI, 1, 4, 4, 4>(a, N, cbs, st, plan) : launch_cfg<DT, EPI, 1, 4, 4, 8>(a, N, cbs, st, plan);
  } else {
    // (WN, NTW) with WN*NTW dividing the tile count, widest first; leftover waves split K.  Small batches: a shape whose
    // launch would leave CUs without a workgroup is passed over for the next narrower one (more column groups on
    // blockIdx.y; the same rule as the gate launches above)
    const int ptiles = N * nint_cdiv(a.W, 16) * nint_cdiv(a.H, mt4 ? 4 : 8);
    auto few = [&](int cols) { return !a.tile_rows && EPI == EPI_DGRAD && 2 * ptiles * (ntiles / cols) < NINT_SPLIT_NUM * n_cu; };
    if (ntiles % 16 == 0 && !few(16)) return mt4 ? launch_cfg<DT, EPI, 4, 1, 4, 4>(a, N, ntiles / 16, st, plan) : launch_cfg<DT, EPI, 4, 1, 4, 8>(a, N, ntiles / 16, st, plan);
    if (ntiles % 12 == 0 && !few(12)) return mt4 ? launch_cfg<DT, EPI, 4, 1, 3, 4>(a, N, ntiles / 12, st, plan) : launch_cfg<DT, EPI, 4, 1, 3, 8>(a, N, ntiles / 12, st, plan);
    if (ntiles % 8 == 0 && !few(8)) return mt4 ? launch_cfg<DT, EPI, 2, 2, 4, 4>(a, N, ntiles / 8, st, plan) : launch_cfg<DT, EPI, 2, 2, 4, 8>(a, N, ntiles / 8, st, plan);
    if (ntiles % 6 == 0 && !few(6)) return mt4 ? launch_cfg<DT, EPI, 2, 2, 3, 4>(a, N, ntiles / 6, st, plan) : launch_cfg<DT, EPI, 2, 2, 3, 8>(a, N, ntiles / 6, st, plan);
    if (ntiles % 4 == 0 && !few(4)) return mt4 ? launch_cfg<DT, EPI, 1, 4, 4, 4>(a, N, ntiles / 4, st, plan) : launch_cfg<DT, EPI, 1, 4, 4, 8>(a, N, ntiles / 4, st, plan);
    if (ntiles % 3 == 0 && !few(3)) return mt4 ? launch_cfg<DT, EPI, 1, 4, 3, 4>(a, N, ntiles / 3, st, plan) : launch_cfg<DT, EPI, 1, 4, 3, 8>(a, N, ntiles / 3, st, plan);
    if (ntiles % 2 == 0 && !few(2)) return mt4 ? launch_cfg<DT, EPI, 1, 4, 2, 4>(a, N, ntiles / 2, st, plan) : launch_cfg<DT, EPI, 1, 4, 2, 8>(a, N, ntiles / 2, st, plan);
    return mt4 ? launch_cfg<DT, EPI, 1, 4, 1, 4>(a, N, ntiles, st, plan) : launch_cfg<DT, EPI, 1, 4, 1, 8>(a, N, ntiles, st, plan);
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static int cell_fwd(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                    const void* x_slab, const void* h_prev, const float* c_prev,
                    void* h_out, float* c_out, void* gates_out, void* stream, ConvPlan* plan) {
  if (!ly || !g || !x_slab || !h_out || !c_out || N <= 0) return NINT_E_ARG;
  if (dtype != NINT_F32 && dtype != NINT_BF16) return NINT_E_ARG;
  if (!(ly->k & 1) || ly->k / 2 > g->P) return NINT_E_ARG;
  if (ly->tile_rows != 0 && ly->tile_rows != 1 && ly->tile_rows != 2 && ly->tile_rows != 4 && ly->tile_rows != 8) return NINT_E_ARG;
  if (!aligned16(x_slab) || !aligned16(h_prev) || !aligned16(ly->Wf) || !aligned16(h_out)) return NINT_E_ALIGN;
  const int es = dtype == NINT_BF16 ? 2 : 4, kc = dtype == NINT_BF16 ? 32 : 16;
  if (ly->Cxp % kc || ly->Chp % kc || ly->Ch16 % 16 || ly->Chp < ly->Ch16) return NINT_E_ARG;
  if ((ly->xfold != 0 && ly->xfold != 1) || (ly->xfold && ly->Cxp < ly->k * ly->Cx)) return NINT_E_ARG;
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
  a.kx0 = ly->xfold ? 1 : ly->k;              // horizontally folded x source: vertical taps only
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.bias = ly->bias_p;
  a.c_prev = c_prev; a.c_out = c_out; a.h_out = (char*)h_out; a.gates_out = (char*)gates_out;
  a.Chp = ly->Chp; a.Ch16 = ly->Ch16;
  a.tile_rows = ly->tile_rows <= 2 ? 0 : ly->tile_rows;
  hipStream_t st = (hipStream_t)stream;
  if (ly->wide < 0 || ly->wide > 2) return NINT_E_ARG;   // (the weight-gradient family switch: nothing to do with this launch)
  // tiny hidden widths (4*Ch <= 32 gate columns: no dense contraction): the VALU stencil kernel (csrc/stencil.hip) -- on request
  // (nint_layer.tile_rows == 1: "one pixel per lane"), or where it measured faster than the padded MFMA tiles (NINT_STENCIL_AUTO).
  // A planned launch (merged grids) has no stencil form: the caller then enqueues it by itself.
  if (nint_internal_stencil_holds(ly) && (ly->tile_rows == 1 || (ly->tile_rows == 0 && NINT_STENCIL_AUTO(dtype)))) {
    if (plan) return NINT_E_SHAPE;
    return nint_internal_stencil_lstm(ly, g, dtype, N, x_slab, h_prev, c_prev, h_out, c_out, gates_out, stream);
  }
  // ... and the library's own choice for such layers: the matrix pipe with a DENSE K (csrc/tiny_gemm.hip)
  if ((ly->tile_rows == 2 || (ly->tile_rows == 0 && NINT_TINY_AUTO(dtype))) && nint_tiny_shape(ly->Cx, ly->Ch, ly->k, ly->xfold, dtype)) {
    if (plan) return NINT_E_SHAPE;
    return nint_internal_tiny_lstm(ly, g, dtype, N, x_slab, h_prev, c_prev, h_out, c_out, gates_out, stream);
  }
  return dtype == NINT_BF16 ? launch_conv<NINT_BF16, EPI_LSTM>(a, N, a.NTt, st, plan)
                            : launch_conv<NINT_F32, EPI_LSTM>(a, N, a.NTt, st, plan);
}

extern "C" int nint_cell_fwd(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                             const void* x_slab, const void* h_prev, const float* c_prev,
                             void* h_out, float* c_out, void* gates_out, void* stream) {
  return cell_fwd(ly, g, dtype, N, x_slab, h_prev, c_prev, h_out, c_out, gates_out, stream, nullptr);
}

int nint_internal_cell_fwd_plan(const CellFwdJob* j, const nint_geom* g, int dtype, int N, ConvPlan* plan) {
  return cell_fwd(j->ly, g, dtype, N, j->x_slab, j->h_prev, j->c_prev, j->h_out, j->c_out, j->gates_out, nullptr, plan);
}

// Up to NINT_MULTI_MAX planned launches that do not depend on each other, as ONE grid.  NINT_E_SHAPE (nothing enqueued): one
// of them has a shape the merged kernels do not hold, or they are of both kinds -- the caller then enqueues them one by one.
int nint_internal_conv_multi(const ConvPlan* plans, int n, int dtype, void* stream, const PwArgs* pw, bool dry_run) {
  if (!plans || n < 1 || n > NINT_MULTI_MAX) return NINT_E_SHAPE;
  ConvMulti m = {};
  size_t lds = 0;
  int b = 0, nfwd = 0;
  for (int i = 0; i < n; ++i) {
    const ConvPlan& pl = plans[i];
    if (!multi_holds(pl.variant)) return NINT_E_SHAPE;
    nfwd += pl.variant / 10000 == EPI_LSTM ? 1 : 0;
    m.a[i] = pl.a; m.variant[i] = pl.variant; m.nbx[i] = pl.gx; m.nwg[i] = pl.gx * pl.gy;
    m.begin[i] = b;
    b += nint_round_up(pl.gx * pl.gy, 8);
    if (pl.lds > lds) lds = pl.lds;
  }
  if (nfwd != 0 && nfwd != n) return NINT_E_SHAPE;
  if (pw && nfwd) return NINT_E_SHAPE;
  m.n = n;
  for (int i = n; i <= NINT_MULTI_MAX; ++i) m.begin[i] = b;
  hipStream_t st = (hipStream_t)stream;
#define NINT_MULTI_LAUNCH(KERN_)                                                                                                      \
  { auto kern = KERN_;                                                                                                                \
    if (lds > 64 * 1024) NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(kern, dim3(b), dim3(256), lds, st, m); }
  bool rows8 = false, dpair = false;
  for (int i = 0; i < n; ++i) rows8 = rows8 || plans[i].variant % 10 == 2;
  for (int i = 0; i < n; ++i) dpair = dpair || plans[i].variant == conv_variant(EPI_DGRAD, 2, 2, 3, 8);
  for (int i = 0; i < n; ++i)     // (the 4-row form of that shape is a case of conv_bwd_multi_kernel only, not of the 8-row kernel)
    if (rows8 && (plans[i].variant == conv_variant(EPI_DGRAD, 2, 2, 3, 4) || plans[i].variant == conv_variant(EPI_DGRAD, 1, 4, 3, 4))) return NINT_E_SHAPE;
  if (dpair)        // (conv_dgrad_multi8_kernel holds these two shapes only)
    for (int i = 0; i < n; ++i)
      if (plans[i].variant != conv_variant(EPI_DGRAD, 2, 2, 3, 8) && plans[i].variant != conv_variant(EPI_DGRAD, 1, 4, 4, 8)) return NINT_E_SHAPE;
  if (pw) {                        // (a case of the 4-row backward kernel only)
    if (rows8 || dpair) return NINT_E_SHAPE;
    const size_t total = (size_t)pw->N * pw->H * pw->W * (pw->Ch16 / 4);
    size_t pb = (total + 255) / 256;
    // as many blocks as the conv problems have workgroups, within [1, 4] items per thread
    const size_t want = (size_t)b;
    if (pb > want) pb = want < (pb + 3) / 4 ? (pb + 3) / 4 : want;
    m.pw = *pw; m.pw_blocks = (int)pb;
    b += (int)pb;
  }
  if (dry_run) return NINT_OK;     // (the caller only asked whether these launches go into one grid)
  if (dpair) { if (dtype == NINT_BF16) NINT_MULTI_LAUNCH(conv_dgrad_multi8_kernel<NINT_BF16>) else NINT_MULTI_LAUNCH(conv_dgrad_multi8_kernel<NINT_F32>) }
  else if (nfwd && rows8) { if (dtype == NINT_BF16) NINT_MULTI_LAUNCH(conv_lstm_multi8_kernel<NINT_BF16>) else NINT_MULTI_LAUNCH(conv_lstm_multi8_kernel<NINT_F32>) }
  else if (nfwd) { if (dtype == NINT_BF16) NINT_MULTI_LAUNCH(conv_lstm_multi_kernel<NINT_BF16>) else NINT_MULTI_LAUNCH(conv_lstm_multi_kernel<NINT_F32>) }
  else if (rows8) { if (dtype == NINT_BF16) NINT_MULTI_LAUNCH(conv_bwd_multi8_kernel<NINT_BF16>) else NINT_MULTI_LAUNCH(conv_bwd_multi8_kernel<NINT_F32>) }
  else { if (dtype == NINT_BF16) NINT_MULTI_LAUNCH(conv_bwd_multi_kernel<NINT_BF16>) else NINT_MULTI_LAUNCH(conv_bwd_multi_kernel<NINT_F32>) }
#undef NINT_MULTI_LAUNCH
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

extern "C" int nint_conv_dgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                               const void* dG, void* dx_accum, void* dh_prev, void* stream) {
  return nint_internal_conv_dgrad(ly, g, dtype, N, dG, dx_accum, dh_prev, false, nullptr, stream);
}

// plan != nullptr: nothing is enqueued, *plan describes the launch (plan->gx == 0: there is nothing to launch)
int nint_internal_conv_dgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* dG, void* dx_accum,
                             void* dh_prev, bool overwrite_dx, const DgradPw* pw, void* stream, ConvPlan* plan) {
  if (plan) plan->gx = 0;
  if (!ly || !g || !dG || N <= 0) return NINT_E_ARG;
  if (dtype != NINT_F32 && dtype != NINT_BF16) return NINT_E_ARG;
  // fused forms: pw->gates set = this layer's pointwise backward on the h columns (dh_prev is then not stored);
  // pw->gates NULL + pw->lo_gates = a classic dgrad (h columns stored) that runs the LOWER layer's pointwise backward
  if (pw && pw->gates && (!pw->c_new || !pw->dc || !pw->dG_out || dh_prev)) return NINT_E_ARG;
  if (pw && !pw->gates && !pw->lo_gates) return NINT_E_ARG;
  if (!dx_accum && !dh_prev && !pw) return NINT_OK;
  if (!aligned16(dG) || !aligned16(ly->Wd)) return NINT_E_ALIGN;
  if (ly->tile_rows != 0 && ly->tile_rows != 1 && ly->tile_rows != 2 && ly->tile_rows != 4 && ly->tile_rows != 8) return NINT_E_ARG;
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
  a.kx0 = ly->k;                              // the dG source always has all k x k taps (a folded x only zeroes weights)
  a.H = g->H; a.W = g->W; a.P = g->P; a.Hh = g->Hh; a.Wh = g->Wh;
  a.out0 = (char*)dx_accum; a.out1 = (char*)dh_prev;
  a.out0_overwrite = overwrite_dx ? 1 : 0;
  a.C0p = ly->Cxp; a.C1p = ly->Chp;
  a.Chp = ly->Chp; a.Ch16 = ly->Ch16;
  a.tile_rows = ly->tile_rows <= 2 ? 0 : ly->tile_rows;    // (1 / 2 = the stencil / dense-K GATE kernels: the backward launches take their own choice)
  // only the n-tiles whose destination exists are computed (fused: the Ch16 real hidden columns, not their padding)
  const int nt_x = ly->Cxp / 16, nt_h = (pw && pw->gates) ? ly->Ch16 / 16 : ly->Chp / 16;
  a.nt_begin = dx_accum ? 0 : nt_x;
  const int ntiles = (dx_accum ? nt_x : 0) + ((dh_prev || (pw && pw->gates)) ? nt_h : 0);
  hipStream_t st = (hipStream_t)stream;
  if (pw) {
    a.pw_gates = (const char*)pw->gates; a.pw_c_prev = pw->c_prev; a.pw_c_new = pw->c_new; a.pw_dc = pw->dc;
    a.pw_old = (const char*)pw->old; a.pw_dG = (char*)pw->dG_out;
    if (pw->lo_gates) {                        // the layer below's pointwise backward of this time step, on the x columns
      if (!dx_accum || !pw->lo_c_new || !pw->lo_dc || !pw->lo_dG_out || pw->lo_Ch16 <= 0 || pw->lo_Ch16 > ly->Cxp) return NINT_E_ARG;
      a.lo_gates = (const char*)pw->lo_gates; a.lo_c_prev = pw->lo_c_prev; a.lo_c_new = pw->lo_c_new; a.lo_dc = pw->lo_dc;
      a.lo_dG = (char*)pw->lo_dG_out; a.lo_Ch16 = pw->lo_Ch16; a.lo_dc_zero = pw->lo_dc_zero ? 1 : 0;
    }
    if (pw->tile_rows) a.tile_rows = pw->tile_rows;
    return dtype == NINT_BF16 ? launch_conv<NINT_BF16, EPI_DGRAD_PW>(a, N, ntiles, st, plan)
                              : launch_conv<NINT_F32, EPI_DGRAD_PW>(a, N, ntiles, st, plan);
  }
  return dtype == NINT_BF16 ? launch_conv<NINT_BF16, EPI_DGRAD>(a, N, ntiles, st, plan)
                            : launch_conv<NINT_F32, EPI_DGRAD>(a, N, ntiles, st, plan);
}

extern "C" int nint_cell_bwd_fused(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* dG_next, void* dx,
                                   const void* gates, const float* c_prev, const float* c_new, const void* dh_above,
                                   float* dc, void* dG, void* stream) {
  DgradPw pw = {};
  pw.gates = gates; pw.c_prev = c_prev; pw.c_new = c_new; pw.dc = dc; pw.old = dh_above; pw.dG_out = dG;
  return nint_internal_conv_dgrad(ly, g, dtype, N, dG_next, dx, nullptr, true, &pw, stream);
}
