// wgrad.hip -- conv backward-weight for the ConvLSTM gate convolution on gfx950 MFMA.
//
//   dW[o][c][ky][kx] = sum_{n,y,x} dG[n,y,x,o] * cat[n, y+ky-p, x+kx-p, c]      (autograd of model.py:220)
//
// All T time steps are reduced in ONE launch per source (x / h): the BPTT loop only stores
// dG_t, so the reduction dimension is K = T*B*H*W pixels and the grid is (split-K, output block).
// GEMM view: M = 64 gate columns n' (4 MFMA row tiles), N = 16*NTC cat channels x all k*k taps,
// K = pixels.  The MFMA K index runs over pixels, but the slabs are channels-last, so both
// operands are read TRANSPOSED from their LDS images:
//   bf16: ds_read_b64_tr_b16 -- a 16-lane group turns a 4-pixel x 16-channel block into
//         "lane = channel, 4 pixels per lane"; two reads give the 8 K-slots of a 16x16x32 MFMA.
//         A (dG) and B (cat, shifted by the tap) use the SAME pixel->K-slot assignment, which is
//         all a reduction needs.
//   f32 : plain ds_read_b32 (lane = channel) feeding v_mfma_f32_16x16x4_f32.
// Taps are free: the cat image is a halo tile and a tap is a constant LDS address offset.
// The bias gradient db[o] = sum_pixels dG[.,o] rides along: the dG fragments are in registers anyway, so ONE wave of the
// workgroups that own channel block 0 of the x source multiplies them with an all-ones fragment (4 extra MFMAs per
// pixel row on the wave with the fewest columns); the result is one more column of the partial slab, folded by the
// same reduction launch.  No separate pass over dG, no per-step partial rows in the BPTT kernels.
// Each workgroup keeps its 64 x (16*NTC*taps) output block in accumulators for its whole pixel
// range and writes ONE partial slab; a second kernel folds the slabs in fixed order
// (bitwise reproducible, no float atomics) into the OIHW gradient.
#include <type_traits>
#include "nint_common.h"

struct WgSrc {           // one source (x or h) of a launch
  const char* dG;        // first image of this source's reduction (the h source may skip the zero-state time step)
  const char* src; long src_img_stride; int src_pix_stride;
  float* partial;
  int CB, JG;            // channel blocks; columns per block slab (the x source's slabs carry the bias-gradient column)
  int nblk;              // workgroup columns of this source: NB * CB * TG
  int ntiles, tiles_per_split;
  int want_db;           // 1: the slab's last column (JG-1) carries the bias gradient (column sums of dG), see below
};
struct WgradArgs {
  long dG_img_stride; int dG_pix_stride;
  WgSrc s[2];
  int nparts;            // 1, or 2: BOTH sources in one launch (same kernel shape), their workgroups interleaved per gate block so
                         // that the x and h workgroups of a pixel range run together and share the dG tiles in L2
  int NTC, J;            // channel tiles per block, (tap, channel-tile) columns in all
  int TG;                // column groups of 4*JW columns (49 taps of a 7x7 kernel: 2)
  int k, p, taps;
  int P, Wh;
  int tiles_x, tiles_y;
};

template <int DT> struct WgTile;
template <> struct WgTile<NINT_BF16> { static constexpr int PR = 4, RA = 160; static constexpr int rb(int ntc) { return ntc == 1 ? 32 : 96; } };
template <> struct WgTile<NINT_F32> { static constexpr int PR = 2, RA = 320; static constexpr int rb(int ntc) { return ntc == 1 ? 64 : 192; } };

// NS = gate-column groups among the 4 waves: NS=1 -> every wave owns all 4 row tiles and a quarter of
// the (tap, channel-tile) columns; NS=2 -> (2 row tiles) x (half of the columns): 25 taps split 13+12
// instead of 7+7+7+4, at the price of more fragment reads per MFMA.
// KS (kernel size) and NTCT (channel tiles per column block) are template parameters so that every LDS
// fragment address is `per-lane base VGPR + compile-time immediate`: the transposed reads then cost no
// VALU instruction at all (the MFMAs leave only ~8 issue cycles each to the rest of the wave).
// KX = horizontal taps of the source: KS, or 1 for a horizontally folded x source (nint_layer.xfold: its columns are
// (vertical tap, folded channel tile) and every tap reads the centre column of the halo tile).
template <int DT, int JW, int NS, int KS, int NTCT, int KX>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs a) {
  typedef Elem<DT> E;
  typedef WgTile<DT> TT;
  constexpr int PR = TT::PR, RA = TT::RA;
  constexpr int NTN = 4 / NS;
  constexpr int A_UNITS_PIX = 64 * E::ES / 16;           // 16-byte data units per dG pixel row (64 gate columns)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int p = KS / 2, k = KS;
  constexpr int HWt = 32 + 2 * p, HHt = PR + 2 * p;
  constexpr int RB = TT::rb(NTCT);
  constexpr int b_units_pix = 16 * NTCT * E::ES / 16;    // data units per cat halo pixel
  // The two LDS images seen as 16-byte units INCLUDING their row padding: a tile is staged by LDS-DMA
  // (global_load_lds_dwordx4: 64 consecutive units per wave instruction, per-lane source address), no VGPR
  // round trip, no ds_write pass; lanes that land on padding re-read the tile's first unit.
  constexpr int UA_ROW = RA / 16, NA_U = PR * 32 * UA_ROW;
  constexpr int UB_PIX = RB / 16, NB_U = HHt * HWt * UB_PIX, NB_U_PAD = (NB_U + 63) / 64 * 64;
  constexpr int a_bytes = PR * 32 * RA;
  constexpr int b_bytes = NB_U_PAD * 16;
  constexpr int buf_bytes = a_bytes + b_bytes;
  static_assert(NA_U % 64 == 0 && a_bytes % 1024 == 0, "whole DMA pieces");
  constexpr int NI_A = NA_U / 64, NI_B = NB_U_PAD / 64, NI = NI_A + NI_B;

  // (integer division runs on the vector ALU: readfirstlane puts the workgroup-uniform results back into scalars)
  int by = blockIdx.y, part = 0;
  if (a.nparts == 2) {                        // per gate block: the x source's channel blocks, then the h source's (TG == 1)
    const int per_nb = a.s[0].CB + a.s[1].CB;
    const int nbq = by / per_nb, r = by - nbq * per_nb;
    part = r >= a.s[0].CB ? 1 : 0;
    by = part ? nbq * a.s[1].CB + (r - a.s[0].CB) : nbq * a.s[0].CB + r;
  }
  part = __builtin_amdgcn_readfirstlane(part);
  by = __builtin_amdgcn_readfirstlane(by);
  // (this source's fields, read once into scalars: the tile loop must not index the argument table)
  struct { const char* dG; const char* src; long src_img_stride; int src_pix_stride; float* partial; int CB, JG, nblk, ntiles, tiles_per_split, want_db; } S;
  {
    const WgSrc& T = a.s[part];
    S.dG = T.dG; S.src = T.src; S.src_img_stride = T.src_img_stride; S.src_pix_stride = T.src_pix_stride; S.partial = T.partial;
    S.CB = T.CB; S.JG = T.JG; S.nblk = T.nblk; S.ntiles = T.ntiles; S.tiles_per_split = T.tiles_per_split; S.want_db = T.want_db;
  }
  const int tg = __builtin_amdgcn_readfirstlane(by % a.TG);   // column group
  const int bc = __builtin_amdgcn_readfirstlane(by / a.TG);   // (gate block, channel block)
  const int nb = bc / S.CB, cb = bc % S.CB;
  const int jb = tg * 4 * JW;                 // first column of this workgroup
  const int t_begin = blockIdx.x * S.tiles_per_split;
  const int t_end = min(S.ntiles, t_begin + S.tiles_per_split);
  const int i0 = (wave % NS) * NTN;           // first row tile (16 gate columns each) of this wave
  const int j0 = (wave / NS) * JW;

  f32x4_t acc[NTN][JW];
#pragma unroll
  for (int i = 0; i < NTN; ++i)
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // bias gradient: channel block 0 / LAST column group of the source that carries it, on the last wave -- which has
  // fewer than JW real columns there (host-checked), so its accumulator column JW-1 is free: no extra registers
  const bool do_db = __builtin_amdgcn_readfirstlane((S.want_db && cb == 0 && tg == a.TG - 1 && wave == 3 && NS == 1) ? 1 : 0);

  // DMA pieces of a tile are dealt to the waves in CONTIGUOUS ranges sized to even out each wave's work per tile:
  // a wave with fewer real columns (25 taps over 4 waves = 7, 7, 7, 4) has idle issue slots that the staging of the
  // next tile can use, so it takes more pieces: n_w = (L - nv_w * CM) / CD with the common level L such that the n_w
  // sum to NI (CM = matrix-pipe cycles per column and tile, CD ~ issue cost of one piece).  Wave-uniform integer math.
  constexpr int NI_WM = (NI + 1) / 2;                    // most pieces one wave may take
  int p_begin, p_cnt;
  {
    constexpr int CM = (DT == NINT_BF16 ? 16 : 32 * 4) * NTN * PR, CD = 100;
    int nv[4], n16[4], tot = 0, sum_nv = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { nv[w] = min(JW, max(0, a.J - jb - (w / NS) * JW)); sum_nv += nv[w]; }
    const int L = (NI * CD + CM * sum_nv) / 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) { n16[w] = max(0, (L - nv[w] * CM) * 16 / CD); tot += n16[w]; }
    // scale to NI pieces (rounded prefix sums), cap every wave at NI_WM and hand the excess on (4 * NI_WM >= 2 * NI)
    int cnt[4], prev = 0, acc16 = 0, excess = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      acc16 += n16[w];
      const int e = w == 3 ? NI : (tot > 0 ? min(NI, (acc16 * NI + tot / 2) / tot) : (w + 1) * NI / 4);
      cnt[w] = e - prev;
      prev = e;
      if (cnt[w] > NI_WM) { excess += cnt[w] - NI_WM; cnt[w] = NI_WM; }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int take = min(NI_WM - cnt[w], excess);
      cnt[w] += take;
      excess -= take;
    }
    const int s1 = cnt[0], s2 = s1 + cnt[1], s3 = s2 + cnt[2];
    // (the divisions above run on the vector ALU: back into scalars, the LDS destination of a DMA piece is an SGPR)
    p_begin = __builtin_amdgcn_readfirstlane(wave == 0 ? 0 : (wave == 1 ? s1 : (wave == 2 ? s2 : s3)));
    p_cnt = __builtin_amdgcn_readfirstlane(wave == 0 ? cnt[0] : (wave == 1 ? cnt[1] : (wave == 2 ? cnt[2] : cnt[3])));
  }
  // Per-lane source offsets of this wave's DMA pieces (piece t = p_begin + i), relative to the wave-uniform tile
  // base: they do not depend on the tile, so the tile loop spends no VALU on staging addresses.
  unsigned doff[NI_WM];
#pragma unroll
  for (int i = 0; i < NI_WM; ++i) {
    const int t = p_begin + i;
    unsigned o = 0;
    if (t < NI_A) {
      const int u = t * 64 + lane;
      const int pix = u / UA_ROW, q = u - pix * UA_ROW;
      if (q < A_UNITS_PIX) o = (unsigned)(((pix >> 5) * a.Wh + (pix & 31)) * a.dG_pix_stride + q * 16);
    } else if (t < NI) {
      const int u = (t - NI_A) * 64 + lane;
      const int hp = u / UB_PIX, q = u - hp * UB_PIX;
      const int hy = hp / HWt, hx = hp - hy * HWt;
      if (hp < HHt * HWt && q < b_units_pix) o = (unsigned)((hy * a.Wh + hx) * S.src_pix_stride + q * 16);
    }
    doff[i] = o;
  }
  // tile coordinates of the next tile to load, kept incrementally (tiles of a split are consecutive)
  int ld_tx, ld_ty, ld_img;
  {
    const int r = t_begin / a.tiles_x;
    ld_tx = t_begin - r * a.tiles_x;
    ld_img = r / a.tiles_y;
    ld_ty = r - ld_img * a.tiles_y;
  }
  const char* const ga0 = S.dG + nb * 64 * E::ES;
  const char* const gb0 = S.src + cb * 16 * NTCT * E::ES;

  auto issue_dma = [&](char* buf) {          // stages tile (ld_img, ld_ty, ld_tx) into buf, then steps to the next tile
    const int y0 = ld_ty * PR, x0 = ld_tx * 32;
    const char* ga = ga0 + (long)ld_img * a.dG_img_stride + ((long)(y0 + a.P) * a.Wh + (x0 + a.P)) * a.dG_pix_stride;
    const char* gb = gb0 + (long)ld_img * S.src_img_stride + ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * S.src_pix_stride;
#pragma unroll
    for (int i = 0; i < NI_WM; ++i) {
      const int t = p_begin + i;               // wave-uniform
      if (i < p_cnt) {
        const char* src = (t < NI_A ? ga : gb) + doff[i];
        char* dst = buf + (t < NI_A ? t * 1024 : a_bytes + (t - NI_A) * 1024);
        // issued as inline asm ON PURPOSE: hipcc would count the builtin as an LDS write and wait vmcnt(0) ahead
        // of the next ds_read (it cannot tell the two buffers apart), which makes the staging synchronous; an asm
        // DMA is invisible to its bookkeeping, so the wait is ours: s_waitcnt vmcnt(0) ahead of the tile's barrier.
        // (M0 = LDS destination base; saved and restored in the same statement.)
        unsigned keep;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
      }
    }
    const bool wx = ld_tx + 1 == a.tiles_x;
    const bool wy = wx && (ld_ty + 1 == a.tiles_y);
    ld_tx = wx ? 0 : ld_tx + 1;
    ld_ty = wy ? 0 : (wx ? ld_ty + 1 : ld_ty);
    ld_img += wy ? 1 : 0;
  };

  if (t_begin < t_end) issue_dma(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // lane constants of the transposed reads
  const int g = lane >> 4, i16 = lane & 15;
  // Per-lane base offsets (VGPRs, computed once).  Column jj of this wave is (tap, channel tile)
  // j0+jj; columns past J are clamped duplicates whose accumulators are never flushed, which keeps
  // the K loop branch-free.
  int vB[JW];
  int vA;
  const int nvalid = __builtin_amdgcn_readfirstlane(min(JW, max(0, a.J - jb - j0)));   // real columns of this wave
  constexpr int JT = KS * KX * NTCT;                                              // = a.J (host-checked)
  constexpr int XO = KX == KS ? 0 : p;                                            // folded: the centre column
  constexpr int NVL = JT % JW == 0 ? JW : JT % JW;                                // columns of the last, partial group
  if constexpr (DT == NINT_BF16) {
    const int q = i16 >> 2, p8 = (i16 & 3) * 8;
    vA = (4 * g + q) * RA + i0 * 32 + p8;             // pixel 4g+q of a 16-pixel half segment, 8 bytes of 4 channels
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const int j = min(jb + j0 + jj, a.J - 1);
      const int tap = j / NTCT, ct = j - tap * NTCT;
      const int tyy = tap / KX, txx = tap - tyy * KX + XO;
      vB[jj] = a_bytes + (tyy * HWt + txx + 4 * g + q) * RB + ct * 32 + p8;
    }
  } else {
    vA = g * RA + (i0 * 16 + i16) * 4;                // MFMA m of a 16-pixel K-step takes pixel 4m+g
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
      const int j = min(jb + j0 + jj, a.J - 1);
      const int tap = j / NTCT, ct = j - tap * NTCT;
      const int tyy = tap / KX, txx = tap - tyy * KX + XO;
      vB[jj] = a_bytes + (tyy * HWt + txx + g) * RB + (ct * 16 + i16) * 4;
    }
  }
  // The tile loop exists once per column count a wave can have (JW, the NVL columns left for the last group, or none):
  // inside it the column count is a compile-time constant, so a tile is ONE basic block -- the compiler is free to
  // hoist the transposed reads of pixel row pr+1 above the MFMAs of row pr -- and no MFMA is predicated.  Every variant
  // issues the same DMA pieces and the same barriers.
  auto run_tiles = [&](auto nvc, auto dbc) __attribute__((always_inline)) {
    constexpr int NV = decltype(nvc)::value;
    constexpr bool DB = decltype(dbc)::value;
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int cur = (tile - t_begin) & 1;
      const bool more = tile + 1 < t_end;
      if (more) issue_dma(smem + (cur ^ 1) * buf_bytes);   // the other buffer was last read before the previous barrier
      const char* Ab = smem + cur * buf_bytes + vA;      // one add per base and tile; everything below is base + immediate
      const char* Bb[JW];
#pragma unroll
      for (int jj = 0; jj < JW; ++jj) Bb[jj] = smem + cur * buf_bytes + vB[jj];
      if constexpr ((NV > 0 || DB) && DT == NINT_BF16) {
#pragma unroll
        for (int pr = 0; pr < PR; ++pr) {
          // pixel -> K-slot: read rd covers pixels rd*16 + 4*g + q of the 32-pixel row segment
          auto read_tr = [&](const char* ad, int half) __attribute__((always_inline)) {
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(ad));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(ad + half));
            u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            return (u32x4_t){l2[0], l2[1], h2[0], h2[1]};
          };
          u32x4_t af[NTN], bf[NV > 0 ? NV : 1];
#pragma unroll
          for (int i = 0; i < NTN; ++i) af[i] = read_tr(Ab + pr * 32 * RA + i * 32, 16 * RA);
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) bf[jj] = read_tr(Bb[jj] + pr * HWt * RB, 16 * RB);
#pragma unroll
          for (int jj = 0; jj < NV; ++jj)
#pragma unroll
            for (int i = 0; i < NTN; ++i) acc[i][jj] = mma_step<NINT_BF16>(af[i], bf[jj], acc[i][jj]);
          if constexpr (DB) {                  // D[gate column][.] += sum over the 32 pixels of this row segment
            const u32x4_t ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
            for (int i = 0; i < NTN; ++i) acc[i][JW - 1] = mma_step<NINT_BF16>(af[i], ones, acc[i][JW - 1]);
          }
        }
      } else if constexpr (NV > 0 || DB) {
#pragma unroll
        for (int pr = 0; pr < PR; ++pr) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            // 16 pixels per K-step: MFMA m takes pixel 4*m + g of the segment as its K index g
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              float af[NTN];
#pragma unroll
              for (int i = 0; i < NTN; ++i)
                af[i] = *(const float*)(Ab + (pr * 32 + ks * 16 + 4 * m) * RA + i * 64);
#pragma unroll
              for (int jj = 0; jj < NV; ++jj) {
                const float bf = *(const float*)(Bb[jj] + (pr * HWt + ks * 16 + 4 * m) * RB);
#pragma unroll
                for (int i = 0; i < NTN; ++i)
                  acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf, acc[i][jj], 0, 0, 0);
              }
              if constexpr (DB) {
#pragma unroll
                for (int i = 0; i < NTN; ++i) acc[i][JW - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], 1.0f, acc[i][JW - 1], 0, 0, 0);
              }
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile has landed ...
      __syncthreads();                           // ... for every wave, and this one is fully read
    }
  };
  if (do_db) {                                            // (wave 3 of the last group: NVL < JW columns, or none)
    if (NVL < JW && nvalid == NVL) run_tiles(std::integral_constant<int, NVL < JW ? NVL : 0>{}, std::true_type{});
    else run_tiles(std::integral_constant<int, 0>{}, std::true_type{});
  } else {
    if (nvalid == JW) run_tiles(std::integral_constant<int, JW>{}, std::false_type{});
    else if (nvalid == NVL) run_tiles(std::integral_constant<int, NVL>{}, std::false_type{});
    else run_tiles(std::integral_constant<int, 0>{}, std::false_type{});      // a wave without real columns only stages and synchronises
  }

  // ---- flush: partial[split][blockIdx.y][j local][n'loc 64][c 16]
  float* out = S.partial + ((size_t)blockIdx.x * S.nblk + by) * S.JG * 1024;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const int j = j0 + jj;
    if (jb + j < a.J) {
#pragma unroll
      for (int i = 0; i < NTN; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[(size_t)j * 1024 + ((i0 + i) * 16 + 4 * g + r) * 16 + i16] = acc[i][jj][r];
    }
  }
  if (do_db) {                                            // slab column JG-1: all 16 "channel" columns hold the same sum
#pragma unroll
    for (int i = 0; i < NTN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)(S.JG - 1) * 1024 + ((i0 + i) * 16 + 4 * g + r) * 16 + i16] = acc[i][JW - 1][r];
  }
}

// ---------------------------------------------------------------------------------------------- the 128-column kernel
// Wide layers (round 4).  The 4-wave kernel above stages a 64-column dG tile and a 16/32-channel cat tile per 64 x (16*NTC*taps)
// output block: 62 (5x5) to 137 (3x3) staged bytes per MFMA, and with every CU streaming two such workgroups the L2 -> LDS
// staging rate itself (~25 GB/s per CU by LDS-DMA) is what the matrix pipe waits for; its waves also re-read the dG fragments
// four times.  This kernel gives an 8-wave workgroup (one per CU, two waves per SIMD) a 128-gate-column x NCT-channel-tile
// output block: wave (gh, ct, half) owns the 64 gate columns gh of channel tile ct for a contiguous range of taps -- 4 row tiles
// x 9 columns (3x3: 4 channel tiles), 4 x 13 (5x5: 2 channel tiles, taps 13 + 12), 4 x 13 (7x7: 1 channel tile) -- so a dG tile
// serves NCT channel tiles and a cat tile both gate halves: 35 (5x5) / 59 (3x3) staged bytes per MFMA and 0.65 / 0.72 instead
// of 0.79 / 0.9 transposed reads per MFMA.  Tap ranges are compile-time per wave role (switch on `half`), so every fragment
// address is `per-lane base + immediate`, as above.  Partials leave in the layout of the 4-wave kernel with NTC = 1, so the
// fold kernel and the workspace arithmetic are shared.  bf16 storage only (f32 mode is not the throughput mode).
template <int KS, int NCT> struct WgWide {
  static constexpr int taps = KS * KS;
  static constexpr int HG = KS == 3 ? 1 : (KS == 5 ? 2 : 4);         // column groups per channel tile
  static constexpr int JW = (taps + HG - 1) / HG;                   // taps per wave: 9, 13, 13
  static constexpr int NVL = taps - (HG - 1) * JW;                  // taps of the last group: 9, 12, 10
  static constexpr int JWA = JW + (NVL == JW ? 1 : 0);              // accumulator columns (the last group keeps one free for db)
  static constexpr int NW = 2 * NCT * HG;                           // waves
  static constexpr int PR = 4, RA = 288;                            // pixel rows per tile; bytes per dG pixel row (256 + 32 pad)
  static constexpr int RB = NCT == 4 ? 160 : (NCT == 2 ? 96 : 32);  // bytes per cat halo pixel (conflict-free strides)
  static constexpr int p = KS / 2, HWt = 32 + 2 * p, HHt = PR + 2 * p;
  static constexpr int UA_ROW = RA / 16, NA_U = PR * 32 * UA_ROW;
  static constexpr int UB_PIX = RB / 16, NB_U = HHt * HWt * UB_PIX, NB_U_PAD = (NB_U + 63) / 64 * 64;
  static constexpr int a_bytes = PR * 32 * RA, b_bytes = NB_U_PAD * 16, buf_bytes = a_bytes + b_bytes;
  static constexpr int NI_A = NA_U / 64, NI_B = NB_U_PAD / 64, NI = NI_A + NI_B;
  static constexpr int NI_W = (NI + NW - 1) / NW;                   // DMA pieces per wave and tile
  static_assert(NA_U % 64 == 0, "whole DMA pieces");
  static_assert(NW == 8, "eight waves: two per SIMD");
};

template <int KS, int NCT>
__global__ __launch_bounds__(512, 2) void wgrad_wide_kernel(WgradArgs a) {
  typedef WgWide<KS, NCT> G;
  constexpr int PR = G::PR, RA = G::RA, RB = G::RB, HWt = G::HWt, HHt = G::HHt, p = G::p, JW = G::JW, JWA = G::JWA, HG = G::HG;
  constexpr int a_bytes = G::a_bytes, buf_bytes = G::buf_bytes, NI_A = G::NI_A, NI = G::NI, NI_W = G::NI_W;
#ifndef NINT_WG_STREAM
#define NINT_WG_STREAM (JW > 9)
#endif
  constexpr bool STREAM = NINT_WG_STREAM;     // 4 x 13 tiles: explicit two-deep fragment stream; 4 x 9: the compiler's schedule
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gh = wave & 1, cgi = wave >> 1, ct = cgi / HG, half = cgi % HG;
  const WgSrc& T = a.s[0];
  const int CBW = T.CB / NCT;                               // channel groups of NCT tiles (T.CB counts 16-channel tiles)
  const int by = blockIdx.y;
  const int nbw = __builtin_amdgcn_readfirstlane(by / CBW), cbw = __builtin_amdgcn_readfirstlane(by % CBW);
  const int t_begin = blockIdx.x * T.tiles_per_split;
  const int t_end = min(T.ntiles, t_begin + T.tiles_per_split);
  const unsigned src_pix_stride = (unsigned)T.src_pix_stride;

  f32x4_t acc[4][JWA];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JWA; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // bias gradient: channel tile 0 of the source that carries it, on the waves of the last tap group (free accumulator column)
  const bool do_db = __builtin_amdgcn_readfirstlane((T.want_db && cbw == 0 && ct == 0 && half == HG - 1) ? 1 : 0);

  // DMA pieces of a tile: piece t = wave + NW * i (dealt round-robin: every wave has the same matrix work here).  The per-lane
  // source offset of a piece is recomputed when it is issued (a dozen VALU instructions between MFMAs): the 4 x 13 accumulator
  // tile leaves no registers to keep NI_W of them.
  auto piece_off = [&](int t) __attribute__((always_inline)) {
    unsigned o = 0;
    if (t < NI_A) {
      const unsigned u = (unsigned)(t * 64 + lane);
      const unsigned pix = __umulhi(u, (unsigned)(((1ull << 32) + G::UA_ROW - 1) / G::UA_ROW)), q = u - pix * G::UA_ROW;
      o = q < 16 ? ((pix >> 5) * a.Wh + (pix & 31)) * a.dG_pix_stride + q * 16 : 0u;
    } else {
      const unsigned u = (unsigned)((t - NI_A) * 64 + lane);
      const unsigned hp = __umulhi(u, (unsigned)(((1ull << 32) + G::UB_PIX - 1) / G::UB_PIX)), q = u - hp * G::UB_PIX;
      const unsigned hy = __umulhi(hp, (unsigned)(((1ull << 32) + HWt - 1) / HWt)), hx = hp - hy * HWt;
      o = (hp < (unsigned)(HHt * HWt) && q < 2u * NCT) ? (hy * a.Wh + hx) * src_pix_stride + q * 16 : 0u;
    }
    return o;
  };
  int ld_tx, ld_ty, ld_img;
  {
    const int r = t_begin / a.tiles_x;
    ld_tx = t_begin - r * a.tiles_x;
    ld_img = r / a.tiles_y;
    ld_ty = r - ld_img * a.tiles_y;
  }
  const char* const ga0 = T.dG + nbw * 128 * 2;
  const char* const gb0 = T.src + cbw * 16 * NCT * 2;
  const long src_img_stride = T.src_img_stride;
  auto issue_dma = [&](char* buf) {
    const int y0 = ld_ty * PR, x0 = ld_tx * 32;
    const char* ga = ga0 + (long)ld_img * a.dG_img_stride + ((long)(y0 + a.P) * a.Wh + (x0 + a.P)) * a.dG_pix_stride;
    const char* gb = gb0 + (long)ld_img * src_img_stride + ((long)(y0 + a.P - p) * a.Wh + (x0 + a.P - p)) * src_pix_stride;
#pragma unroll
    for (int i = 0; i < NI_W; ++i) {
      const int t = wave + G::NW * i;              // wave-uniform
      if (t < NI) {
        const char* src = (t < NI_A ? ga : gb) + piece_off(t);
        char* dst = buf + (t < NI_A ? t * 1024 : a_bytes + (t - NI_A) * 1024);
        // (inline asm for the reason given in wgrad_kernel: the wait is ours, ahead of the tile's barrier)
        unsigned keep;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dst);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
      }
    }
    const bool wx = ld_tx + 1 == a.tiles_x;
    const bool wy = wx && (ld_ty + 1 == a.tiles_y);
    ld_tx = wx ? 0 : ld_tx + 1;
    ld_ty = wy ? 0 : (wx ? ld_ty + 1 : ld_ty);
    ld_img += wy ? 1 : 0;
  };
  if (t_begin < t_end) issue_dma(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p8 = (i16 & 3) * 8;
  const int vA = (4 * g + q4) * RA + gh * 128 + p8;                   // row tile i: + i * 32
  const int vB = a_bytes + (4 * g + q4) * RB + ct * 32 + p8;          // tap (ty, tx): + (ty * HWt + tx) * RB

  auto run_tiles = [&](auto tap0c, auto nvc, auto dbc) __attribute__((always_inline)) {
    constexpr int TAP0 = decltype(tap0c)::value, NV = decltype(nvc)::value;
    constexpr bool DB = decltype(dbc)::value;
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int cur = (tile - t_begin) & 1;
      if (tile + 1 < t_end) issue_dma(smem + (cur ^ 1) * buf_bytes);   // the other buffer was last read before the previous barrier
      const char* Ab = smem + cur * buf_bytes + vA;
      const char* Bb = smem + cur * buf_bytes + vB;
      auto read_tr = [&](const char* ad, int half_off) __attribute__((always_inline)) {
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(ad));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(ad + half_off));
        u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        return (u32x4_t){l2[0], l2[1], h2[0], h2[1]};
      };
#pragma unroll
      for (int pr = 0; pr < PR; ++pr) {
        u32x4_t af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = read_tr(Ab + pr * 32 * RA + i * 32, 16 * RA);
        auto colptr = [&](int jj) __attribute__((always_inline)) {
          const int tap = TAP0 + jj, tyy = tap / KS, txx = tap - tyy * KS;
          return Bb + (tyy * HWt + txx) * RB + pr * HWt * RB;
        };
        if constexpr (STREAM) {
          // cat fragments STREAMED: two live at a time, the next column's reads issued ahead of this column's MFMAs (the
          // compiler's own schedule hoists a whole row of them: 52 registers the 4 x 13 tile does not have)
          u32x4_t bfq[2];
          if constexpr (NV > 0) bfq[0] = read_tr(colptr(0), 16 * RB);
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) {
            if (jj + 1 < NV) bfq[(jj + 1) & 1] = read_tr(colptr(jj + 1), 16 * RB);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][jj] = mma_step<NINT_BF16>(af[i], bfq[jj & 1], acc[i][jj]);
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) {
            const u32x4_t bf = read_tr(colptr(jj), 16 * RB);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][jj] = mma_step<NINT_BF16>(af[i], bf, acc[i][jj]);
          }
        }
        if constexpr (DB) {
          const u32x4_t ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][JWA - 1] = mma_step<NINT_BF16>(af[i], ones, acc[i][JWA - 1]);
        }
        if constexpr (STREAM) {
          __builtin_amdgcn_sched_group_barrier(0x100, NV > 0 ? 10 : 8, 0);
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) {
            if (jj + 1 < NV) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          }
          if constexpr (DB) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile has landed ...
      __syncthreads();                                   // ... for every wave, and this one is fully read
    }
  };
  // wave roles: the tap range is a compile-time constant inside each instance of the tile loop
  auto role = [&](auto hc) __attribute__((always_inline)) {
    constexpr int H_ = decltype(hc)::value;
    constexpr int TAP0 = H_ * JW, NV = H_ == HG - 1 ? G::NVL : JW;
    if (H_ == HG - 1 && do_db) run_tiles(std::integral_constant<int, TAP0>{}, std::integral_constant<int, NV>{}, std::true_type{});
    else run_tiles(std::integral_constant<int, TAP0>{}, std::integral_constant<int, NV>{}, std::false_type{});
  };
  if constexpr (HG == 1) role(std::integral_constant<int, 0>{});
  else if constexpr (HG == 2) { if (half == 0) role(std::integral_constant<int, 0>{}); else role(std::integral_constant<int, 1>{}); }
  else { if (half == 0) role(std::integral_constant<int, 0>{}); else if (half == 1) role(std::integral_constant<int, 1>{});
         else if (half == 2) role(std::integral_constant<int, 2>{}); else role(std::integral_constant<int, 3>{}); }

  // ---- flush in the 4-wave kernel's slab layout with NTC = 1: partial[split][(nb, cb)][tap][n'loc 64][c 16]
  const int nb = 2 * nbw + gh, cb = cbw * NCT + ct;
  float* out = T.partial + ((size_t)blockIdx.x * T.nblk + (size_t)nb * T.CB + cb) * T.JG * 1024;
  const int tap0 = half * JW, nv = half == HG - 1 ? G::NVL : JW;
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    if (jj < nv) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(tap0 + jj) * 1024 + (i * 16 + 4 * g + r) * 16 + i16] = acc[i][jj][r];
    }
  }
  if (do_db) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)(T.JG - 1) * 1024 + (i * 16 + 4 * g + r) * 16 + i16] = acc[i][JWA - 1][r];
  }
}

// Fold the split-K slabs of ONE source (x or h part) into dW (OIHW f32).  A workgroup owns 64 consecutive
// elements of the slab layout [block][j][n'loc 64][c 16] (one 256-byte line per split, fully coalesced) and
// spreads the splits over its blockDim/64 waves; the per-wave sums are folded through LDS.  The order is a
// fixed function of the launch shape (bitwise reproducible); only the single write per weight is scattered.
struct ReduceEntry {
  const float* part; float* dW;
  int Cx, Ch, Ch16, k, NB, CB, NTC, J, splits, is_h, xfold;
  int TG, JG;                               // column groups per block column, columns per block slab (incl. the db column)
  float* db;                                // non-NULL: slab column JG-1 of (channel block 0, last group) is the bias gradient
  int waves;                                // waves that share the splits of one 64-element line (the others exit)
  unsigned blk_begin;                       // first workgroup of this (layer, source) in the merged launch
};
struct ReduceTable { ReduceEntry e[2 * NINT_MAX_LAYERS]; int n; };

__global__ void wgrad_reduce_kernel(ReduceTable t) {
  __shared__ f32x4_t red[1024];
  int ei = 0;
  for (int q = 1; q < t.n; ++q) ei = blockIdx.x >= t.e[q].blk_begin ? q : ei;     // entries are in launch order
  const ReduceEntry& E = t.e[ei];
  const int Cx = E.Cx, Ch = E.Ch, k = E.k, CB = E.CB, NTC = E.NTC, J = E.J, splits = E.splits, is_h = E.is_h;
  const int TG = E.TG, JG = E.JG;
  const float* __restrict__ part = E.part;
  float* __restrict__ dW = E.dW;
  const int taps = k * k, Ctot = Cx + Ch;
  const size_t slab = (size_t)E.NB * CB * TG * JG * 1024;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, G = E.waves;   // waves >= G idle through the barrier
  // a workgroup owns 256 consecutive slab elements, 4 per lane (16-byte loads: 1 KiB per wave and split)
  const size_t i = ((size_t)(blockIdx.x - E.blk_begin) * 64 + lane) * 4;  // slab is a multiple of 1024: no tail
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  if (grp < G) {
#pragma unroll 8
    // NON-TEMPORAL: the slabs are read exactly once (in the step: 76 -> 66 us at B = 8, 76 -> 56 us at B = 1, where the fold is 5 % of
    // the step; the same hint on the weight-gradient kernels' slab STORES costs them more than it saves here: profiles/r04_h_fold_nt.txt)
    for (int sp = grp; sp < splits; sp += G) s += __builtin_nontemporal_load((const f32x4_t*)(part + i + (size_t)sp * slab));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (grp != 0) return;
  for (int q = 1; q < G; ++q) s += red[q * 64 + lane];
  const int c16 = i & 15, nloc = (i >> 4) & 63;       // c16 is a multiple of 4: the lane's elements are channels c16 .. c16+3
  size_t r = i >> 10;
  const int jl = r % JG; r /= JG;
  const int tgi = r % TG; r /= TG;
  const int cb = r % CB;
  const int nb = r / CB;
  const int np = nb * 64 + nloc;                      // gate column n' = (cblock*4+gate)*16+col
  const int ch = (np >> 6) * 16 + (np & 15), gate = (np >> 4) & 3;
  const int JR = JG - (E.db ? 1 : 0);                 // real columns per group
  if (jl >= JR) {                                     // the bias-gradient column (all 16 "channels" hold the same sum)
    if (cb == 0 && tgi == TG - 1 && c16 == 0 && ch < Ch) E.db[gate * Ch + ch] = s[0];
    return;
  }
  const int j = tgi * JR + jl;
  if (j >= J || ch >= Ch) return;                     // unused slots of the last column group / padding rows
  const int tap0 = j / NTC;
  const int ct = j - tap0 * NTC;
  const int Csrc = is_h ? Ch : Cx;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int cc = (cb * NTC + ct) * 16 + c16 + e;          // channel inside this source
    int tap = tap0;
    if (!is_h && E.xfold) {                           // folded x source: column (ky, kx*Cx + c) -> W[.][c][ky][kx]
      if (cc >= k * Cx) continue;
      tap = tap0 * k + cc / Cx;
      cc = cc % Cx;
    }
    if (cc >= Csrc) continue;                         // padding columns
    const int ic = is_h ? Cx + cc : cc;
    dW[(((size_t)(gate * Ch + ch)) * Ctot + ic) * taps + tap] = s[e];
  }
}

// ------------------------------------------------------------------------------ host side
struct WgPart { int NTC, J, JW, KX, CB, TG, JG, splits;   // one source (x or h) of the reduction
                int wide_nct, wcols; };                     // > 0: the 8-wave 128-column kernel with this many channel tiles per workgroup (wgrad_wide_kernel); its workgroup columns
struct WgPlan {
  WgPart part[2];
  int NB, tiles_x, tiles_y, ntiles;
  size_t off_h, total_floats;
  bool merged;            // both sources in ONE launch (narrow layers: the launches are bound by re-reading dG, and the x and h
                          // workgroups of a pixel range then share its tiles in L2)
};

// The 128-column kernel holds a SOURCE (x or h) of a layer when it is an unfolded bf16 slab, the gate columns come in pairs of
// 64-column blocks and the source's channels in whole channel groups (3x3: 64 channels, 5x5: 32, 7x7: 16) -- the reference
// stack's second layer (64 -> 32, 3x3) takes it for its x source only.  nint_layer.wide: 1 = never, 2 / 0 = wherever held.
static int wg_wide_nct(const nint_layer* ly, int dtype, int q) {
  if (dtype != NINT_BF16 || ly->wide == 1 || (q == 0 && ly->xfold)) return 0;
  const int nct = ly->k == 3 ? 4 : (ly->k == 5 ? 2 : (ly->k == 7 ? 1 : 0));
  if (!nct || (4 * ly->Ch16) % 128 || (q == 0 ? ly->Cxp : ly->Chp) % (16 * nct)) return 0;
  return nct;
}

#ifndef NINT_WG_MIN_TILES
#define NINT_WG_MIN_TILES 8
#endif
static int wg_plan(const nint_layer* ly, int dtype, int n_cu, int N, const nint_geom* g, WgPlan* pl) {
  if (ly->k != 1 && ly->k != 3 && ly->k != 5 && ly->k != 7) return NINT_E_SHAPE;
  pl->NB = 4 * ly->Ch16 / 64;
  const int PR = dtype == NINT_BF16 ? 4 : 2;
  pl->tiles_x = g ? nint_cdiv(g->W, 32) : 1;
  pl->tiles_y = g ? nint_cdiv(g->H, PR) : 1;
  pl->ntiles = g ? N * pl->tiles_x * pl->tiles_y : (1 << 30);
  if (n_cu <= 0) n_cu = 256;
  size_t floats[2];
  for (int q = 0; q < 2; ++q) {
    WgPart& w = pl->part[q];
    const int Cp = q == 0 ? ly->Cxp : ly->Chp;
    w.wide_nct = wg_wide_nct(ly, dtype, q);
    w.wcols = 0;
    if (w.wide_nct) {
      // one 8-wave workgroup per CU; partial slabs in the 4-wave kernel's layout with one channel tile per block (NTC = 1)
      w.KX = ly->k; w.NTC = 1; w.J = ly->k * ly->k; w.JW = 0; w.TG = 1;
      w.JG = w.J + (q == 0 ? 1 : 0);
      w.CB = Cp / 16;
      w.wcols = (pl->NB / 2) * (w.CB / w.wide_nct);
      int s = nint_cdiv(n_cu, w.wcols);
      if (s > pl->ntiles / NINT_WG_MIN_TILES) s = pl->ntiles / NINT_WG_MIN_TILES;
      if (s < 1) s = 1;
      s = nint_cdiv(pl->ntiles, nint_cdiv(pl->ntiles, s));
      if (s >= 8) s -= s % 8;             // all workgroup columns of a split on one XCD (they share its dG / cat tiles in L2)
      w.splits = s;
      floats[q] = (size_t)w.splits * pl->NB * w.CB * w.JG * 1024;
      continue;
    }
    w.KX = (q == 0 && ly->xfold) ? 1 : ly->k;           // folded x source: vertical taps only
    const int taps = ly->k * w.KX;
    w.NTC = (taps * 2 <= 20 && Cp % 32 == 0) ? 2 : 1;   // channel tiles per column block
    w.J = taps * w.NTC;
    // every wave owns all 4 row tiles (64 gate columns) and a quarter of the (tap, channel-tile) columns
    w.JW = w.J <= 20 ? 5 : 7;
    // more than 4*JW columns (the 49 taps of a 7x7 kernel): column groups on extra workgroups; only whole waves of
    // columns there (the tile loop exists for JW, J % JW and 0 columns per wave)
    w.TG = nint_cdiv(w.J, 4 * w.JW);
    w.JG = (w.TG == 1 ? w.J : 4 * w.JW) + (q == 0 ? 1 : 0);     // the x part's slabs carry the bias-gradient column
    if (w.TG > 1 && w.J % w.JW) return NINT_E_SHAPE;
    if (q == 0 && w.J - (w.TG - 1) * 4 * w.JW > 3 * w.JW + (w.JW - 1)) return NINT_E_SHAPE;   // the db column needs wave 3's last accumulator column free
    const int CW = 16 * w.NTC;
    if (Cp % CW) return NINT_E_SHAPE;
    w.CB = Cp / CW;
    // two workgroups per CU in flight, but never fewer than NINT_WG_MIN_TILES pixel tiles per split: the
    // accumulator flush (J KiB-tiles per workgroup) must stay small against the K work.  (32 until round 3; at B = 1-2
    // per GPU that left the narrow layers' launches at 92-276 workgroups: 8 measured +2.9 % on the B = 1 step, +0.9 % at
    // B = 2, nothing at B >= 4 where the workgroup cap binds first; 4 and 2 equal 8.)
#ifndef NINT_WG_MIN_TILES
#define NINT_WG_MIN_TILES 8
#endif
    int s = nint_cdiv(2 * n_cu, pl->NB * w.CB * w.TG);
    if (s > pl->ntiles / NINT_WG_MIN_TILES) s = pl->ntiles / NINT_WG_MIN_TILES;
    if (s < 1) s = 1;
    // re-derive the split count so that no split is empty
    w.splits = nint_cdiv(pl->ntiles, nint_cdiv(pl->ntiles, s));
    floats[q] = (size_t)w.splits * pl->NB * w.CB * w.TG * w.JG * 1024;
  }
  // same kernel shape for both sources and few workgroup columns: one launch, one split count
  const WgPart &wx = pl->part[0], &wh = pl->part[1];
  pl->merged = !wx.wide_nct && !wh.wide_nct && wx.NTC == wh.NTC && wx.JW == wh.JW && wx.KX == wh.KX && wx.TG == 1 && wh.TG == 1 && pl->NB * (wx.CB + wh.CB) <= 8;
  if (pl->merged) {
    int s = nint_cdiv(2 * n_cu, pl->NB * (wx.CB + wh.CB));
    if (s > pl->ntiles / NINT_WG_MIN_TILES) s = pl->ntiles / NINT_WG_MIN_TILES;
    if (s < 1) s = 1;
    s = nint_cdiv(pl->ntiles, nint_cdiv(pl->ntiles, s));
    // a multiple of 8 splits: workgroup (split, column) sits on XCD (split + splits * column) % 8, so all columns of a split
    // -- the workgroups that read the same dG tiles -- then share one L2 (86 splits for layer 1: 534 us; 80: see DESIGN.md)
    if (s >= 8) s -= s % 8;
    for (int q = 0; q < 2; ++q) {
      WgPart& w = pl->part[q];
      w.splits = s;
      floats[q] = (size_t)w.splits * pl->NB * w.CB * w.TG * w.JG * 1024;
    }
  }
  pl->off_h = floats[0];
  pl->total_floats = floats[0] + floats[1];
  return NINT_OK;
}

extern "C" size_t nint_wgrad_workspace_bytes(const nint_layer* ly, int dtype, int n_cu) {
  if (!ly) return 0;
  WgPlan pl;
  // upper bound independent of N: splits are capped by 2*n_cu / blocks
  if (wg_plan(ly, dtype, n_cu, 1, nullptr, &pl) != NINT_OK) return 0;
  return pl.total_floats * sizeof(float);
}

template <int DT, int JW, int KS, int NTCT, int KX>
static int launch_wgrad(WgradArgs& a, int splits, int nblk, hipStream_t st) {
  typedef WgTile<DT> TT;
  const int p = a.p;
  const int a_bytes = TT::PR * 32 * TT::RA;
  const int b_bytes = nint_round_up((TT::PR + 2 * p) * (32 + 2 * p) * TT::rb(a.NTC), 1024);   // whole 1-KiB DMA pieces
  const size_t lds = 2 * (size_t)(a_bytes + b_bytes);
  if (lds > 160 * 1024) return NINT_E_LDS;
  if (a.k != KS || a.NTC != NTCT || a.J != KS * KX * NTCT) return NINT_E_ARG;
  auto kern = wgrad_kernel<DT, JW, 1, KS, NTCT, KX>;
  if (lds > 64 * 1024)
    NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(splits, nblk), dim3(256), lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

template <int KS, int NCT>
static int launch_wgrad_wide(WgradArgs& a, int splits, int cols, hipStream_t st) {
  typedef WgWide<KS, NCT> G;
  const size_t lds = 2 * (size_t)G::buf_bytes;
  if (lds > 160 * 1024) return NINT_E_LDS;
  auto kern = wgrad_wide_kernel<KS, NCT>;
  NINT_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(splits, cols), dim3(512), lds, st, a);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

// instantiated (kernel size, channel tiles, columns per wave, horizontal taps) combinations
template <int DT>
static int dispatch_wgrad(WgradArgs& a, const WgPart& w, int nblk, hipStream_t st) {
  const int key = a.k * 1000 + w.NTC * 100 + w.JW * 10 + w.KX;
  switch (key) {
    case 5 * 1000 + 1 * 100 + 7 * 10 + 5: return launch_wgrad<DT, 7, 5, 1, 5>(a, w.splits, nblk, st);
    case 3 * 1000 + 2 * 100 + 5 * 10 + 3: return launch_wgrad<DT, 5, 3, 2, 3>(a, w.splits, nblk, st);
    case 3 * 1000 + 1 * 100 + 5 * 10 + 3: return launch_wgrad<DT, 5, 3, 1, 3>(a, w.splits, nblk, st);
    case 1 * 1000 + 2 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 1, 2, 1>(a, w.splits, nblk, st);
    case 1 * 1000 + 1 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 1, 1, 1>(a, w.splits, nblk, st);
    // horizontally folded x source (thin first-layer inputs)
    case 5 * 1000 + 2 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 5, 2, 1>(a, w.splits, nblk, st);
    case 5 * 1000 + 1 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 5, 1, 1>(a, w.splits, nblk, st);
    case 3 * 1000 + 2 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 3, 2, 1>(a, w.splits, nblk, st);
    case 3 * 1000 + 1 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 3, 1, 1>(a, w.splits, nblk, st);
    // 7x7 kernels: 49 taps in two column groups; folded thin inputs 7 or 14 columns
    case 7 * 1000 + 1 * 100 + 7 * 10 + 7: return launch_wgrad<DT, 7, 7, 1, 7>(a, w.splits, nblk, st);
    case 7 * 1000 + 2 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 7, 2, 1>(a, w.splits, nblk, st);
    case 7 * 1000 + 1 * 100 + 5 * 10 + 1: return launch_wgrad<DT, 5, 7, 1, 1>(a, w.splits, nblk, st);
    default: return NINT_E_SHAPE;
  }
}

// Weight / bias gradients of several layers ("jobs") in one go: two MFMA launches per layer (x and h source) into
// consecutive regions of ONE workspace, then ONE launch folds every split-K slab of every layer into its dW and db
// (the x part's slabs carry the bias-gradient column).
// h_skip: the first h_skip images have an identically zero h source (h_{-1} = 0 of a sequence that starts from
// the zero state, model.py:259-262): the h part skips them -- 1/T of its work.
int nint_internal_conv_wgrad_multi(const WgJob* jobs, int njobs, const nint_geom* g, int dtype, float* partial,
                                   size_t partial_bytes, int n_cu, void* stream, Probe* probe) {
  if (!jobs || njobs < 1 || njobs > NINT_MAX_LAYERS || !g || !partial) return NINT_E_ARG;
  if (dtype != NINT_F32 && dtype != NINT_BF16) return NINT_E_ARG;
  const int es = dtype == NINT_BF16 ? 2 : 4;
  hipStream_t st = (hipStream_t)stream;
  ReduceTable rt = {};
  size_t off = 0;
  unsigned blk = 0;
  int red_threads = 256;
  for (int q = 0; q < njobs; ++q) {
    const WgJob& jb = jobs[q];
    const nint_layer* ly = jb.ly;
    if (!ly || !jb.dG || !jb.x_slab || !jb.h_slab || !jb.dW || !jb.db || jb.N <= 0) return NINT_E_ARG;
    if (jb.h_skip < 0 || jb.h_skip > jb.N) return NINT_E_ARG;
    WgPlan pl;
    int rc = wg_plan(ly, dtype, n_cu, jb.N, g, &pl);
    if (rc != NINT_OK) return rc;
    if ((off + pl.total_floats) * sizeof(float) > partial_bytes) return NINT_E_ARG;
    float* base = partial + off;
    off += (pl.total_floats + 63) / 64 * 64;
    const int Gc = 4 * ly->Ch16;
    WgradArgs a = {};
    a.dG_pix_stride = Gc * es;
    a.dG_img_stride = (long)g->Hh * g->Wh * a.dG_pix_stride;
    a.k = ly->k; a.p = ly->k / 2;
    a.P = g->P; a.Wh = g->Wh;
    a.tiles_x = pl.tiles_x; a.tiles_y = pl.tiles_y;
    if (probe) probe->stamp(NINT_PROBE_WGRAD, q, 0, 0);
    for (int part = 0; part < 2; ++part) {
      const WgPart& w = pl.part[part];
      WgSrc& S = a.s[pl.merged ? part : 0];
      const int skip = part == 1 ? jb.h_skip : 0;
      S.dG = (const char*)jb.dG + (size_t)skip * a.dG_img_stride;
      const int Cp = part == 0 ? ly->Cxp : ly->Chp;
      S.src_pix_stride = Cp * es;
      S.src_img_stride = (long)g->Hh * g->Wh * S.src_pix_stride;
      S.src = (const char*)(part == 0 ? jb.x_slab : jb.h_slab) + (size_t)skip * S.src_img_stride;
      S.partial = base + (part == 0 ? 0 : pl.off_h);
      S.CB = w.CB; S.JG = w.JG;
      S.nblk = pl.NB * w.CB * w.TG;
      S.want_db = part == 0 ? 1 : 0;           // the x part sees every image (the h part may skip the first time step)
      S.ntiles = (jb.N - skip) * pl.tiles_x * pl.tiles_y;  // empty splits flush zeros
      S.tiles_per_split = S.ntiles > 0 ? nint_cdiv(S.ntiles, w.splits) : 1;   // spread what is there evenly over the planned splits
      a.NTC = w.NTC; a.J = w.J; a.TG = w.TG;
      a.taps = ly->k * w.KX;
      if (w.wide_nct) {                        // the 8-wave 128-column kernel, one launch per source
        a.nparts = 1;
        rc = ly->k == 3 ? launch_wgrad_wide<3, 4>(a, w.splits, w.wcols, st)
           : (ly->k == 5 ? launch_wgrad_wide<5, 2>(a, w.splits, w.wcols, st) : launch_wgrad_wide<7, 1>(a, w.splits, w.wcols, st));
        if (rc != NINT_OK) return rc;
      } else if (!pl.merged || part == 1) {    // separate launches per source, or both sources in one
        a.nparts = pl.merged ? 2 : 1;
        const int nblk = pl.merged ? a.s[0].nblk + a.s[1].nblk : S.nblk;
        rc = dtype == NINT_BF16 ? dispatch_wgrad<NINT_BF16>(a, w, nblk, st) : dispatch_wgrad<NINT_F32>(a, w, nblk, st);
        if (rc != NINT_OK) return rc;
      }
      ReduceEntry& E = rt.e[rt.n++];
      E.part = S.partial; E.dW = jb.dW; E.db = part == 0 ? jb.db : nullptr;
      E.Cx = ly->Cx; E.Ch = ly->Ch; E.Ch16 = ly->Ch16; E.k = ly->k; E.NB = pl.NB; E.CB = w.CB; E.NTC = w.NTC; E.J = w.J;
      E.splits = w.splits; E.is_h = part; E.xfold = ly->xfold; E.TG = w.TG; E.JG = w.JG;
      E.blk_begin = blk;
      blk += (unsigned)((size_t)pl.NB * w.CB * w.TG * w.JG * 1024 / 256);
      E.waves = 4;                               // 4 waves share the splits of a 256-element line (16-wave workgroups for the
                                                 // many-split entries made every entry's workgroup 1024 threads: 46 -> 40 us)
      if (E.waves * 64 > red_threads) red_threads = E.waves * 64;
    }
    if (probe) probe->stamp(NINT_PROBE_WGRAD, q, 0, 1);
  }
  if (probe) probe->stamp(NINT_PROBE_FOLD, 0, 0, 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blk), dim3(red_threads), 0, st, rt);
  NINT_LAUNCH_CHECK();
  if (probe) probe->stamp(NINT_PROBE_FOLD, 0, 0, 1);
  return NINT_OK;
}

extern "C" int nint_conv_wgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* dG,
                               const void* x_slab, const void* h_slab, float* dW, float* db, float* partial,
                               size_t partial_bytes, int n_cu, void* stream) {
  const WgJob jb = {ly, N, dG, x_slab, h_slab, dW, db, 0};
  return nint_internal_conv_wgrad_multi(&jb, 1, g, dtype, partial, partial_bytes, n_cu, stream);
}
