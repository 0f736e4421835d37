// seq.hip -- library entry points that are not kernels themselves: version / device info,
// the hardware self-test, and the whole-sequence drivers that enqueue every launch of a
// ConvLSTM forward (model.py:253-274) or its BPTT from C++ on one HIP stream, so the Python
// side pays one ctypes call per pass instead of one per kernel.
#include <string.h>
#include "nint_common.h"

extern "C" int nint_version(void) { return NINT_VERSION; }

extern "C" const char* nint_error_string(int code) {
  switch (code) {
    case NINT_OK: return "ok";
    case NINT_E_ARG: return "nint: invalid argument";
    case NINT_E_SHAPE: return "nint: shape not supported by any kernel instantiation";
    case NINT_E_LDS: return "nint: tile does not fit in LDS";
    case NINT_E_ALIGN: return "nint: pointer not 16-byte aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "nint: unknown error";
  }
}

extern "C" int nint_device_info(int* n_cu, int* lds_bytes_per_cu, int* wave_size, char* name, int name_len) {
  int dev = 0;
  NINT_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  NINT_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (wave_size) *wave_size = prop.warpSize;
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  return NINT_OK;
}

extern "C" int nint_geom_make(nint_geom* g, int H, int W, int P) {
  if (!g || H <= 0 || W <= 0 || P < 0) return NINT_E_ARG;
  g->H = H; g->W = W; g->P = P;
  g->Hh = nint_round_up(H, 8) + 2 * P;
  g->Wh = nint_round_up(W, 32) + 2 * P;
  return NINT_OK;
}

// ------------------------------------------------------------------------------ self-test
// out[0..255]     : D of mfma_f32_16x16x32_bf16 with A[m][k] = m + 1 (k == 3 only), B[k][n] = 32 + n (k == 3 only),
//                   i.e. D[m][n] = (m+1)*(32+n) (asymmetric),
//                   stored as out[lane*4 + r]   -> pins the C/D register map and the A/B k-slot pairing
// out[1024..2047] : same for mfma_f32_16x16x4f32 with the one-hot k = 2
// out[2048..2303] : ds_read_b64_tr_b16 of an LDS image img[row][col] = 64*row + col (16 columns,
//                   32-byte rows), lane 4q+p of each 16-lane group addressing row 4*g+q, cols 4p..4p+3
//                   stored as out[2048 + lane*4 + e]
__global__ void selftest_kernel(float* out) {
  const int lane = threadIdx.x;
  const int g = lane >> 4, i16 = lane & 15;
  {
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) {
      const int kk = 8 * g + j;
      a[j] = (__bf16)(kk == 3 ? (float)(i16 + 1) : 0.f);
      b[j] = (__bf16)(kk == 3 ? (float)(32 + i16) : 0.f);
    }
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
  }
  {
    const float a = (g == 2) ? (float)(i16 + 1) : 0.f;
    const float b = (g == 2) ? (float)(32 + i16) : 0.f;
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[1024 + lane * 4 + r] = c[r];
  }
  {
    __shared__ __attribute__((aligned(16))) uint16_t img[16 * 16];
    for (int i = lane; i < 256; i += 64) img[i] = (uint16_t)(64 * (i / 16) + (i % 16));
    __syncthreads();
    const int q = i16 >> 2, p = i16 & 3;
    const char* ad = (const char*)img + (4 * g + q) * 32 + p * 8;
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)ad);
    for (int e = 0; e < 4; ++e) out[2048 + lane * 4 + e] = (float)(uint16_t)v[e];
  }
}

extern "C" int nint_selftest(float* out, void* stream) {
  if (!out) return NINT_E_ARG;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  NINT_LAUNCH_CHECK();
  return NINT_OK;
}

// ------------------------------------------------------------------------------ sequence drivers
// Every launch of a pass is enqueued from C++ on the CALLER's stream, in dependency order.  The library owns no
// streams, events or other state.  At the bench's batch size a (t, layer) wavefront on side streams and weight-gradient
// reductions overlapped with the BPTT chain were both at or below this order (co-resident
// MFMA-bound kernels evict each other's LDS / register budget; every cross-stream edge is a marker on the first layer's
// chain).  For small batches the forward wavefront runs as ONE grid per step instead (nint_seq.wave, conv_lstm_multi_kernel).
static inline size_t esize(int dtype) { return dtype == NINT_BF16 ? 2 : 4; }

__global__ void probe_stamp_kernel(unsigned long long* slot, unsigned long long tag) {
  if (threadIdx.x == 0) { slot[0] = tag; slot[1] = __builtin_amdgcn_s_memrealtime(); }
}
void Probe::stamp(unsigned kind, int layer, int t, int end) {
  if (!buf || n >= cap || (kind != NINT_PROBE_CAL && !((mask >> kind) & 1))) return;
  const unsigned long long tag = kind | ((unsigned long long)layer << 8) | ((unsigned long long)t << 16) | ((unsigned long long)end << 31) | (1ull << 63);
  hipLaunchKernelGGL(probe_stamp_kernel, dim3(1), dim3(64), 0, st, buf + 2 * (size_t)n, tag);
  ++n;
}
static Probe make_probe(const nint_seq* s, bool bwd, void* stream) {
  Probe p = {};
  if (s->probe && s->probe_mask && s->probe_slots >= 8) {
    const int half = s->probe_slots / 2;
    p.buf = s->probe + (bwd ? 2 * (size_t)half : 0);
    p.cap = half; p.mask = (unsigned)s->probe_mask; p.st = (hipStream_t)stream;
    p.stamp(NINT_PROBE_CAL, 0, 0, 0);          // two back-to-back stamps: the price of the brackets themselves
    p.stamp(NINT_PROBE_CAL, 0, 0, 1);
  }
  return p;
}

static int seq_check(const nint_seq* s) {
  if (!s || s->L < 1 || s->L > NINT_MAX_LAYERS || s->B < 1 || s->T < 1) return NINT_E_ARG;
  if (s->dtype != NINT_F32 && s->dtype != NINT_BF16) return NINT_E_ARG;
  if (!s->xs) return NINT_E_ARG;
  for (int l = 0; l < s->L; ++l) {
    if (!s->h[l] || !s->c[l]) return NINT_E_ARG;
    if (l > 0 && s->layer[l].Cxp != s->layer[l - 1].Chp) return NINT_E_ARG;
  }
  return NINT_OK;
}

extern "C" int nint_seq_fwd(const nint_seq* s, void* stream) {
  int rc = seq_check(s);
  if (rc != NINT_OK) return rc;
  const nint_geom* g = &s->g;
  const size_t es = esize(s->dtype);
  const size_t halo_px = (size_t)g->Hh * g->Wh, comp_px = (size_t)g->H * g->W;
  const int B = s->B, L = s->L;
  Probe probe = make_probe(s, false, stream);
  auto job = [&](int l, int t) {
    const nint_layer* ly = &s->layer[l];
    const char* x_slab = (l == 0)
        ? (const char*)s->xs + (size_t)t * B * halo_px * ly->Cxp * es            // x[:, t]  (model.py:266)
        : (const char*)s->h[l - 1] + (size_t)(t + 1) * B * halo_px * ly->Cxp * es;  // h of the layer below (model.py:271)
    const size_t hs = (size_t)B * halo_px * ly->Chp * es;
    const size_t cs = (size_t)B * comp_px * ly->Chp;
    const bool zero_state = (t == 0 && !s->has_init_state);     // model.py:259-262: zeros -> skip the h half of K
    CellFwdJob j;
    j.ly = ly; j.x_slab = x_slab;
    j.h_prev = zero_state ? nullptr : (const char*)s->h[l] + (size_t)t * hs;
    j.c_prev = zero_state ? nullptr : s->c[l] + (size_t)t * cs;
    j.h_out = (char*)s->h[l] + (size_t)(t + 1) * hs;
    j.c_out = s->c[l] + (size_t)(t + 1) * cs;
    j.gates_out = s->gates[l] ? (char*)s->gates[l] + (size_t)t * B * comp_px * 4 * ly->Ch16 * es : nullptr;
    return j;
  };
  // wave = 2 / 3: every gate launch of the pass on 8-row tiles (the merged grids AND the lone launches at the ends of the
  // wavefront, so that the pass equals the time-major order with tile_rows pinned to 8 bit for bit)
  const bool rows8 = (s->wave == 2 || s->wave == 3 || s->wave == 4) && L > 1 && L <= NINT_MULTI_MAX;
  auto launch = [&](int l, int t) {
    CellFwdJob j = job(l, t);
    nint_layer l8;
    if (rows8 && j.ly->tile_rows == 0) { l8 = *j.ly; l8.tile_rows = 8; j.ly = &l8; }
    probe.stamp(NINT_PROBE_GATE, l, t, 0);
    const int r = nint_cell_fwd(j.ly, g, s->dtype, B, j.x_slab, j.h_prev, j.c_prev, j.h_out, j.c_out, j.gates_out, stream);
    probe.stamp(NINT_PROBE_GATE, l, t, 1);
    return r;
  };
  if (s->wave && L > 1 && L <= NINT_MULTI_MAX) {
    // (t, layer) WAVEFRONT: step w runs gate(l, w - l) of every layer -- each needs gate(l-1, w-l) and gate(l, w-l-1), both
    // of step w-1 -- as ONE grid (conv_lstm_multi_kernel).  T + L - 1 launches instead of T * L; the same workgroups
    // on the same data, so the results are those of the time-major order bit for bit.
    for (int w = 0; w < s->T + L - 1; ++w) {
      ConvPlan plans[NINT_MULTI_MAX];
      int lt[NINT_MULTI_MAX][2], n = 0;
      for (int l = 0; l < L; ++l) {
        const int t = w - l;
        if (t < 0 || t >= s->T) continue;
        lt[n][0] = l; lt[n][1] = t; ++n;
      }
      rc = n > 1 ? NINT_OK : NINT_E_SHAPE;
      nint_layer ly8[NINT_MULTI_MAX];
      for (int q = 0; q < n && rc == NINT_OK; ++q) {
        CellFwdJob j = job(lt[q][0], lt[q][1]);
        if (rows8 && j.ly->tile_rows == 0) {
          // mid-size batches: the first layer's 8-row tiles make the merged grid a 256-register kernel at two workgroups per CU,
          // where the narrow layers' 4-row tiles lose what they were chosen for (a third and fourth workgroup per CU): every
          // problem of the grid takes 8-row tiles (half the weight bytes per MFMA).  Measured at B = 8, three fresh-process
          // pairs: forward 2.84-2.86 -> 2.71-2.72 ms, step 1018-1024 -> 1037-1042 samples/s (profiles/r04_d_wave_rows8.txt).
          // = the time-major order with tile_rows pinned to 8, bit for bit; against the default order (4-row narrow tiles) the
          // four K-slice partials of a pixel are summed in another order: f32 rounding.
          ly8[q] = *j.ly; ly8[q].tile_rows = 8; j.ly = &ly8[q];
        }
        rc = nint_internal_cell_fwd_plan(&j, g, s->dtype, B, &plans[q]);
      }
      if (rc == NINT_OK) {
        probe.stamp(NINT_PROBE_WAVE, n, w, 0);
        rc = nint_internal_conv_multi(plans, n, s->dtype, stream);
        probe.stamp(NINT_PROBE_WAVE, n, w, 1);         // (a shape the merged grid does not hold: an empty bracket, then the launches one by one)
      }
      if (rc == NINT_E_SHAPE) {                // a shape the merged grid does not hold (or a single launch): one by one
        for (int q = 0; q < n; ++q) {
          rc = launch(lt[q][0], lt[q][1]);
          if (rc != NINT_OK) return rc;
        }
      } else if (rc != NINT_OK) {
        return rc;
      }
    }
    return NINT_OK;
  }
  for (int t = 0; t < s->T; ++t) {                               // model.py:265
    for (int l = 0; l < L; ++l) {                                // model.py:267
      rc = launch(l, t);
      if (rc != NINT_OK) return rc;
    }
  }
  return NINT_OK;
}

// default of the "lower layer's pointwise backward on the fused layer's x columns" option: on (measured inside the bench
// step, five alternations on one device: 962.2 -> 963.6 samples/s, every pair positive; profiles/HISTORY.md)
#ifndef NINT_AUTO_LO
#define NINT_AUTO_LO true
#endif

extern "C" int nint_seq_bwd(const nint_seq* s, void* stream) {
  int rc = seq_check(s);
  if (rc != NINT_OK) return rc;
  const nint_geom* g = &s->g;
  const size_t es = esize(s->dtype);
  const size_t halo_px = (size_t)g->Hh * g->Wh, comp_px = (size_t)g->H * g->W;
  const int B = s->B, L = s->L;
  for (int l = 0; l < L; ++l)
    if (!s->gates[l] || !s->dG[l] || !s->dh[l] || !s->dc[l] || !s->dW[l] || !s->db[l]) return NINT_E_ARG;
  if (s->need_dx && !s->dx) return NINT_E_ARG;
  if (!s->wg_partial) return NINT_E_ARG;
  if (s->fuse_bwd < 0 || (s->fuse_bwd > 2 && !(s->fuse_bwd & 0x40000000))) return NINT_E_ARG;
  if (s->bwd_parts < 0 || s->bwd_parts > 2) return NINT_E_ARG;

  // BPTT.  A layer runs either the CLASSIC step (pointwise backward of time u, then conv backward-data of time u) or the
  // FUSED step X[u] = conv backward-data of time u with the pointwise backward of time u-1 in its epilogue
  // (nint_cell_bwd_fused: d/dh_{u-1} never goes to memory).  Chosen per layer by the K-steps of its dgrad launch:
  // measured inside the bench step on two devices (bench.py --fuse-bwd 0x40000000|masks), fusing the 18-step top layer alone gives
  // +0.7 ... +1.0 %, the 36-step layer -0.9 %, the 200-step layer -1.7 % (its workgroups run their phases in lockstep, so
  // the heavier epilogue adds its full HBM time instead of hiding behind the other workgroup's matrix work).
  // A fused layer consumes the x columns that the layer above produced for time u-1, so it runs ONE time step behind the
  // layer above: at outer step s layer l works on time u_l = s + off_l, off_l = number of fused layers among l..L-1.
  // Fused layer: u = T -> pointwise backward of T-1 alone (d/dh_{T-1} comes from the head / the caller),
  // 1 <= u <= T-1 -> X[u], u = 0 -> plain conv backward-data of time 0.
  // A fused layer above a classic one can ALSO run that layer's pointwise backward, on its x columns (they are the last
  // contribution to the lower layer's d/dh of the same time step): lo[l] -- the lower layer then only launches its dgrad.
  bool fused[NINT_MAX_LAYERS], lo[NINT_MAX_LAYERS], loc[NINT_MAX_LAYERS];
  int off[NINT_MAX_LAYERS], pw_done[NINT_MAX_LAYERS];
  const bool explicit_mask = (s->fuse_bwd & 0x40000000) != 0;
  for (int l = L - 1; l >= 0; --l) {
    const nint_layer* ly = &s->layer[l];
    const int ksteps = (4 * ly->Ch16 / (s->dtype == NINT_BF16 ? 32 : 16)) * ly->k * ly->k;   // K-steps of the layer's dgrad launch
    fused[l] = explicit_mask ? ((s->fuse_bwd >> l) & 1) != 0 : (s->fuse_bwd == 2 || (s->fuse_bwd == 0 && ksteps <= 24));
    off[l] = (l == L - 1 ? 0 : off[l + 1]) + (fused[l] ? 1 : 0);
    pw_done[l] = -1;
  }
  for (int l = 0; l < L; ++l) {
    lo[l] = fused[l] && l > 0 && !fused[l - 1] && (explicit_mask ? ((s->fuse_bwd >> (8 + l)) & 1) != 0 : NINT_AUTO_LO);
    // ... and a CLASSIC layer can do the same for the classic layer below it (its dgrad then runs the fused kernel with the
    // h columns stored; 4-row tiles, whose many small workgroups overlap the added HBM traffic)
    loc[l] = !fused[l] && l > 0 && !fused[l - 1] && explicit_mask && ((s->fuse_bwd >> (16 + l)) & 1) != 0;
  }
  const int T = s->T;
  Probe probe = make_probe(s, true, stream);
  // Small batches (nint_seq.wave): the bottom layer's dgrad of one outer step and the top layer's fused step of the next are
  // ADJACENT launches that share no buffer when the stack has three or more layers (the top layer's step touches its own
  // state and layer L-2's; the bottom dgrad reads dG[0] and writes dh[0] / dx): they go out as ONE grid
  // (nint_internal_conv_multi).  The bottom dgrad is held back (`pend`) until the next launch is known.
  const int wv = s->wave;
  const bool merge = (wv == 1 || wv == 3) && L >= 3 && fused[L - 1] && !fused[0] && !loc[0];   // (wave == 2: the forward wavefront only)
  // Mid-size batches (wave = 4): the bottom layer's dgrad of time u+1 waits for the dgrad of the layer above of time u instead and
  // the two go out as one grid, the wide one first (the narrow layer's workgroups fill its last round: the forward wavefront's
  // effect).  Both produce a piece of the bottom layer's d/dh of time u, so each stores its own -- the layer above into dh[0],
  // the bottom layer into the head of the split-K scratch, which is idle until the weight gradients -- and the bottom layer's
  // pointwise backward adds the two (f32: the same sum as the read-modify-write of the time-major order, bit for bit; bf16: each
  // piece is rounded to bf16 before the f32 add instead of the running sum after it).
  const size_t dh0_bytes = (size_t)B * comp_px * s->layer[0].Chp * es;
  const bool merge_d = (wv == 4 || wv == 5) && L >= 2 && !fused[0] && !fused[1] && !loc[0] && !loc[1] && s->wg_partial_bytes >= dh0_bytes;
  void* const dh0_own = merge_d ? s->wg_partial : s->dh[0];
  struct { bool on; ConvPlan plan; const void* dG; void* dx; void* dh_prev; bool ow; int u; } pend = {};
  auto flush = [&]() {                           // the held-back dgrad as a launch of its own
    if (!pend.on) return (int)NINT_OK;
    pend.on = false;
    probe.stamp(NINT_PROBE_DGRAD, 0, pend.u, 0);
    const int r = nint_internal_conv_dgrad(&s->layer[0], g, s->dtype, B, pend.dG, pend.dx, pend.dh_prev, pend.ow, nullptr, stream);
    probe.stamp(NINT_PROBE_DGRAD, 0, pend.u, 1);
    return r;
  };
  // ... and the bottom layer's pointwise backward of time u waits for the top layer's fused step of time u-1, the next launch in
  // this order and independent of it (it touches layers >= 1 only): one grid, the fused step's workgroups first (conv_bwd_multi_kernel
  // with a pointwise problem; the same arithmetic: bit-identical).  B = 2 / 4 / 8: another +1.3 / +0.6 / +0.25 % (profiles/r04_f_wave4.txt).
  const bool merge_p = merge_d && L >= 3 && fused[L - 1];
  struct { bool on; int t; } pend_pw = {};
  auto p0_launch = [&](int t, PwArgs* plan) {
    const nint_layer* l0 = &s->layer[0];
    const size_t cs0 = (size_t)B * comp_px * l0->Chp, Gc0 = 4 * (size_t)l0->Ch16;
    return nint_internal_cell_bwd_pointwise(l0, g, s->dtype, B, (const char*)s->gates[0] + (size_t)t * B * comp_px * Gc0 * es, s->c[0] + (size_t)t * cs0,
                                            s->c[0] + (size_t)(t + 1) * cs0, s->dh[0], s->dc[0], (char*)s->dG[0] + (size_t)t * B * halo_px * Gc0 * es,
                                            t == T - 1 && (s->zero_dstate & 1), stream, merge_d && t < T - 1 ? dh0_own : nullptr, plan);
  };
  auto flush_pw = [&]() {
    if (!pend_pw.on) return (int)NINT_OK;
    pend_pw.on = false;
    probe.stamp(NINT_PROBE_POINTWISE, 0, pend_pw.t, 0);
    const int r = p0_launch(pend_pw.t, nullptr);
    probe.stamp(NINT_PROBE_POINTWISE, 0, pend_pw.t, 1);
    return r;
  };
  for (int so = T - 1; so >= -off[0] && s->bwd_parts != 2; --so) {      // (part 2: the chain ran in the part-1 call)
    for (int l = L - 1; l >= 0; --l) {
      const int u = so + off[l];
      if (u < 0 || u > (fused[l] ? T : T - 1)) continue;
      if (pend_pw.on && !(l == L - 1 && u >= 1 && u < T)) {     // (anything but the fused step it waits for)
        rc = flush_pw();
        if (rc != NINT_OK) return rc;
      }
      const nint_layer* ly = &s->layer[l];
      const size_t cs = (size_t)B * comp_px * ly->Chp;
      const size_t Gc = 4 * (size_t)ly->Ch16;
      const size_t gs = (size_t)B * comp_px * Gc * es, dgs = (size_t)B * halo_px * Gc * es;
      // c[l][0] is the (zero or given) initial state, so c_prev is always a valid pointer
      auto pointwise = [&](int t) {      // consumes dh[l] / dc[l] of time t, writes dG of time t
        // first BPTT step: state gradients flagged all-zero are neither read (dc) nor accumulated into (dh below)
        if (!merge_d || l == 0) { const int rf = flush(); if (rf != NINT_OK) return rf; }    // (wave = 4: the held-back launch touches layer 0 only)
        if (merge_p && l == 0 && so + off[L - 1] >= 2) {     // the next outer step opens with a fused step of the top layer
          pend_pw.on = true; pend_pw.t = t;
          return (int)NINT_OK;
        }
        probe.stamp(NINT_PROBE_POINTWISE, l, t, 0);
        const int r = nint_internal_cell_bwd_pointwise(ly, g, s->dtype, B, (const char*)s->gates[l] + (size_t)t * gs, s->c[l] + (size_t)t * cs,
                                                       s->c[l] + (size_t)(t + 1) * cs, s->dh[l], s->dc[l], (char*)s->dG[l] + (size_t)t * dgs,
                                                       t == T - 1 && ((s->zero_dstate >> (2 * l)) & 1), stream,
                                                       merge_d && l == 0 && t < T - 1 ? dh0_own : nullptr);
        probe.stamp(NINT_PROBE_POINTWISE, l, t, 1);
        return r;
      };
      // destination of the x columns of time t: the layer below's dh, or this time step's dx slab (written once)
      void* dx_dst = (l > 0) ? s->dh[l - 1]
                             : (s->need_dx ? (void*)((char*)s->dx + (size_t)u * B * comp_px * ly->Cxp * es) : nullptr);
      // ... stored where nothing else is there: dx; a dh flagged zero at the first step; a FUSED layer below (its dh
      // buffer only ever carries these columns).  Accumulated onto the h columns a classic layer below stored.
      const bool ow = l == 0 ? true : (u == T - 1 ? (((s->zero_dstate >> (2 * (l - 1) + 1)) & 1) != 0) : (fused[l - 1] || (merge_d && l == 1)));
      // at time 0 with a zero initial state nobody consumes d/dh_{-1}
      void* dh_prev = (u == 0 && !s->has_init_state) ? nullptr : (l == 0 && u > 0 ? dh0_own : s->dh[l]);   // (time 0: the caller reads d/dh_{-1} from dh[0])
      if (!fused[l]) {
        if (pw_done[l] != u) {                 // (else: the fused layer above already ran this pointwise backward)
          rc = pointwise(u);
          if (rc != NINT_OK) return rc;
        }
        DgradPw pw = {};
        if (loc[l]) {
          const nint_layer* lb = &s->layer[l - 1];
          const size_t cs_b = (size_t)B * comp_px * lb->Chp, Gc_b = 4 * (size_t)lb->Ch16;
          pw.lo_gates = (const char*)s->gates[l - 1] + (size_t)u * B * comp_px * Gc_b * es;
          pw.lo_c_prev = s->c[l - 1] + (size_t)u * cs_b; pw.lo_c_new = s->c[l - 1] + (size_t)(u + 1) * cs_b;
          pw.lo_dc = s->dc[l - 1]; pw.lo_dG_out = (char*)s->dG[l - 1] + (size_t)u * B * halo_px * Gc_b * es;
          pw.lo_Ch16 = lb->Ch16;
          pw.lo_dc_zero = u == T - 1 && ((s->zero_dstate >> (2 * (l - 1))) & 1);
          pw.tile_rows = 4;
          pw_done[l - 1] = u;
        }
        if (merge_d && l == 1 && pend.on) {       // the bottom layer's dgrad of the step before + this one: one grid
          ConvPlan pl[2];
          pl[0] = pend.plan;
          rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l] + (size_t)u * dgs, dx_dst, dh_prev, ow, nullptr, stream, &pl[1]);
          if (rc != NINT_OK) return rc;
          rc = pl[1].gx > 0 ? nint_internal_conv_multi(pl, 2, s->dtype, stream, nullptr, true) : NINT_E_SHAPE;     // (dry run: is there such a grid?)
          if (rc == NINT_OK) {
            probe.stamp(NINT_PROBE_BWD_PAIR, l, u, 0);
            rc = nint_internal_conv_multi(pl, 2, s->dtype, stream);
            probe.stamp(NINT_PROBE_BWD_PAIR, l, u, 1);
          }
          if (rc == NINT_OK) { pend.on = false; continue; }
          if (rc != NINT_E_SHAPE) return rc;
        }
        if (!merge_d || l <= 1) {
          rc = flush();
          if (rc != NINT_OK) return rc;
        }
        if ((merge || merge_d) && l == 0 && so > -off[0]) {  // (not the very last launch: there is a top-layer step to pair it with)
          pend.dG = (const char*)s->dG[l] + (size_t)u * dgs; pend.dx = dx_dst; pend.dh_prev = dh_prev; pend.ow = ow; pend.u = u;
          rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, pend.dG, pend.dx, pend.dh_prev, pend.ow, nullptr, stream, &pend.plan);
          if (rc != NINT_OK) return rc;
          pend.on = pend.plan.gx > 0;
          continue;
        }
        // (time 0 of the bottom layer from a zero state without an input gradient: nothing to launch -- and nothing to bracket: rounds
        // 3-4 stamped this empty call, one zero among the 12 layer-0 dgrad durations of a step)
        const bool nop = !dx_dst && !dh_prev && !loc[l];
        if (!nop) probe.stamp(NINT_PROBE_DGRAD, l, u, 0);
        rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l] + (size_t)u * dgs, dx_dst, dh_prev, ow, loc[l] ? &pw : nullptr, stream);
        if (!nop) probe.stamp(NINT_PROBE_DGRAD, l, u, 1);
      } else if (u == T) {
        rc = pointwise(T - 1);
      } else if (u >= 1) {
        DgradPw pw = {};
        pw.gates = (const char*)s->gates[l] + (size_t)(u - 1) * gs; pw.c_prev = s->c[l] + (size_t)(u - 1) * cs;
        pw.c_new = s->c[l] + (size_t)u * cs; pw.dc = s->dc[l]; pw.old = l < L - 1 ? s->dh[l] : nullptr;
        pw.dG_out = (char*)s->dG[l] + (size_t)(u - 1) * dgs;
        if (lo[l]) {                           // the classic layer below: its pointwise backward of time u rides on the x columns
          const nint_layer* lb = &s->layer[l - 1];
          const size_t cs_b = (size_t)B * comp_px * lb->Chp, Gc_b = 4 * (size_t)lb->Ch16;
          pw.lo_gates = (const char*)s->gates[l - 1] + (size_t)u * B * comp_px * Gc_b * es;
          pw.lo_c_prev = s->c[l - 1] + (size_t)u * cs_b; pw.lo_c_new = s->c[l - 1] + (size_t)(u + 1) * cs_b;
          pw.lo_dc = s->dc[l - 1]; pw.lo_dG_out = (char*)s->dG[l - 1] + (size_t)u * B * halo_px * Gc_b * es;
          pw.lo_Ch16 = lb->Ch16;
          pw.lo_dc_zero = u == T - 1 && ((s->zero_dstate >> (2 * (l - 1))) & 1);
          pw_done[l - 1] = u;
        }
        if (merge && pend.on && l == L - 1) {    // the top layer's step right behind the held-back bottom dgrad: one grid
          ConvPlan pl[2];
          pl[0] = pend.plan;
          if (s->wave == 3) pw.tile_rows = 8;    // (experiment: the fused step on 8-row tiles inside the two-workgroups-per-CU grid)
          rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l] + (size_t)u * dgs, dx_dst, nullptr, ow, &pw, stream, &pl[1]);
          if (rc != NINT_OK) return rc;
          rc = pl[1].gx > 0 ? nint_internal_conv_multi(pl, 2, s->dtype, stream, nullptr, true) : NINT_E_SHAPE;     // (dry run: is there such a grid?)
          if (rc == NINT_OK) {
            probe.stamp(NINT_PROBE_BWD_PAIR, l, u, 0);
            rc = nint_internal_conv_multi(pl, 2, s->dtype, stream);
            probe.stamp(NINT_PROBE_BWD_PAIR, l, u, 1);
          }
          if (rc == NINT_OK) { pend.on = false; continue; }
          if (rc != NINT_E_SHAPE) return rc;
        }
        if (!merge_d) {
          rc = flush();
          if (rc != NINT_OK) return rc;
        }
        if (pend_pw.on) {                        // this fused step and the bottom layer's pointwise backward of the step before: one grid
          ConvPlan pl;
          PwArgs pa;
          rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l] + (size_t)u * dgs, dx_dst, nullptr, ow, &pw, stream, &pl);
          if (rc == NINT_OK) rc = p0_launch(pend_pw.t, &pa);
          if (rc != NINT_OK) return rc;
          rc = pl.gx > 0 ? nint_internal_conv_multi(&pl, 1, s->dtype, stream, &pa, true) : NINT_E_SHAPE;
          if (rc == NINT_OK) {
            probe.stamp(NINT_PROBE_BWD_PW, l, u, 0);
            rc = nint_internal_conv_multi(&pl, 1, s->dtype, stream, &pa);
            probe.stamp(NINT_PROBE_BWD_PW, l, u, 1);
          }
          if (rc == NINT_OK) { pend_pw.on = false; continue; }
          if (rc != NINT_E_SHAPE) return rc;
          rc = flush_pw();
          if (rc != NINT_OK) return rc;
        }
        probe.stamp(NINT_PROBE_FUSED, l, u, 0);
        rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l] + (size_t)u * dgs, dx_dst, nullptr, ow, &pw, stream);
        probe.stamp(NINT_PROBE_FUSED, l, u, 1);
      } else {
        if (!merge_d) {
          rc = flush();
          if (rc != NINT_OK) return rc;
        }
        const bool nop = !dx_dst && !dh_prev;
        if (!nop) probe.stamp(NINT_PROBE_DGRAD, l, 0, 0);
        rc = nint_internal_conv_dgrad(ly, g, s->dtype, B, (const char*)s->dG[l], dx_dst, dh_prev, ow, nullptr, stream);
        if (!nop) probe.stamp(NINT_PROBE_DGRAD, l, 0, 1);
      }
      if (rc != NINT_OK) return rc;
    }
  }
  rc = flush_pw();
  if (rc != NINT_OK) return rc;
  rc = flush();
  if (rc != NINT_OK) return rc;
  // weight / bias gradients: ONE reduction over all T time steps per layer and source, all layers' folds merged
  WgJob jobs[NINT_MAX_LAYERS];
  for (int l = 0; l < L; ++l) {
    const nint_layer* ly = &s->layer[l];
    const char* x_all = (l == 0) ? (const char*)s->xs
                                 : (const char*)s->h[l - 1] + (size_t)B * halo_px * ly->Cxp * es;  // h^{l-1}_t = slab t+1
    // h_{-1} = 0 for a sequence from the zero state: the h part of the reduction skips time step 0
    jobs[l] = WgJob{ly, s->T * B, s->dG[l], x_all, s->h[l] /* h_{t-1} = slab t */, s->dW[l], s->db[l],
                    s->has_init_state ? 0 : B};
  }
  // (bwd_parts: layers >= 1 in the first call, layer 0 in the second; each call folds what it reduced)
  const int j0 = s->bwd_parts == 1 ? 1 : 0, j1 = s->bwd_parts == 2 ? 1 : L;
  if (j1 > j0) {
    rc = nint_internal_conv_wgrad_multi(jobs + j0, j1 - j0, g, s->dtype, s->wg_partial, s->wg_partial_bytes, s->n_cu, stream,
                                        probe.buf ? &probe : nullptr);
    if (rc != NINT_OK) return rc;
  }
  return NINT_OK;
}
