/*
 * nint.h -- C ABI of the MI355X-native ConvLSTM hot path for Smart-NINT (nasa-niswan).
 *
 * The reference (smhassanerfani/nasa-niswan) has NO native/FFI interface: its boundary for
 * this path is the Python surface `model.py:ConvLSTMCell/ConvLSTM` + the `train.py` batch
 * step.  This header is the boundary the drop-in Python modules in `nasa-niswan_amd/` bind
 * with ctypes; every entry point cites the reference lines whose ATen op sequence it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the parameter says "host".
 *   - every entry takes an explicit `stream` (hipStream_t as void*) and never synchronises.
 *   - every entry returns 0 on success, a negative NINT_E_* code for bad arguments or a
 *     positive hipError_t value; nothing throws or aborts.
 *   - the caller owns all memory (the Python side allocates through torch's caching
 *     allocator); the library keeps no state between calls, reads no environment variables and owns
 *     no streams or events: every launch goes to the caller's stream.
 *
 * Internal data layout ("slabs"), chosen for gfx950 rather than inherited from NCHW:
 *   halo slab     ET [N][Hh][Wh][Cp]   channels-last, ET = float or bf16, physical zero halo P
 *                                      on every side plus zero slack up to a multiple of 8 rows /
 *                                      32 columns, Cp = channels rounded up to KC (16 f32 / 32 bf16)
 *   compact slab  [N][H][W][Cp]        channels-last, no halo: f32 for the cell state c and dc, ET for the
 *                                      transient gradients dh, dx (written once, read once per BPTT step)
 *   gate stash    ET [N][H][W][Gc]     Gc = 4*Ch16, column n' = (cblock*4 + gate)*16 + col
 *   image index   n = t*B + b          (time-major so that one launch can span all T)
 */
#ifndef NINT_H
#define NINT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NINT_VERSION 111

enum { NINT_F32 = 0, NINT_BF16 = 1 };

enum {
  NINT_OK = 0,
  NINT_E_ARG = -1,      /* inconsistent / unsupported argument */
  NINT_E_SHAPE = -2,    /* shape not supported by any kernel instantiation */
  NINT_E_LDS = -3,      /* tile does not fit in the 160 KiB LDS */
  NINT_E_ALIGN = -4     /* pointer not 16-byte aligned */
};

/* Spatial geometry shared by all slabs of one model instance. */
typedef struct nint_geom {
  int32_t H, W;    /* grid the model runs on (100x154 = 90x144 + halo 5 in the reference, launcher.sh:24) */
  int32_t P;       /* physical zero halo of halo slabs = max_l(k_l/2) */
  int32_t Hh, Wh;  /* halo-slab rows / cols: roundup(H,8)+2P, roundup(W,32)+2P */
} nint_geom;

/* One ConvLSTMCell (reference model.py:196-231) in packed form. */
typedef struct nint_layer {
  int32_t Cx, Cxp;        /* input channels / padded to KC */
  int32_t Ch, Ch16, Chp;  /* hidden channels / padded to 16 / padded to KC */
  int32_t k;              /* odd kernel size, padding k/2 (model.py:204) */
  int32_t tile_rows;      /* rows of the gate / dgrad kernels' pixel tile: 0 = chosen per launch shape, or 4 / 8.  Tiny layers
                           * (nint_stencil_holds()) have two more gate kernels: 1 = one pixel per LANE, the vector-ALU stencil
                           * kernel (csrc/stencil.hip); 2 = the matrix pipe with a dense K (csrc/tiny_gemm.hip; also the library's
                           * choice for such layers in f32 storage).  On other layers 1 / 2 mean 0. */
  int32_t xfold;          /* 1: the x source is HORIZONTALLY FOLDED (thin inputs, first layer only): slab channel
                           * kx*Cx + c holds x[.., x + kx - k/2][c] (0 outside the image), Cxp = roundup(k*Cx, KC), and
                           * the x part of K is k vertical taps x k*Cx channels instead of k*k taps x Cx channels padded
                           * to KC each (reference layer 0: Conv2d(5+64 -> 256, k=5), model.py:207-211: 5 x-steps, not 25) */
  int32_t wide;           /* weight-gradient kernel family (csrc/wgrad.hip): 0 = the library's choice (the 8-wave kernel with a
                           * 128-gate-column output block wherever it is instantiated and the layer is wide enough, else
                           * the 4-wave 64-column kernel); 1 = always the 4-wave kernel; 2 = the 8-wave kernel wherever it is
                           * instantiated (bf16, k = 3 / 5 / 7, unfolded sources, gate columns a multiple of 128, channel
                           * counts a multiple of its channel group), whatever the layer's width.  Same sums in another f32
                           * order.  (Rounds 2-3 used this field for an 8-wave GATE kernel that measured slower everywhere
                           * and left the library in round 4: tools/experiments/.) */
  const void* Wf;         /* fwd weights, MFMA-fragment order, ET   (nint_pack_weights) */
  const void* Wd;         /* dgrad weights (transposed + flipped), ET */
  const float* bias_p;    /* bias permuted to gate-stash column order [4*Ch16] */
} nint_layer;

/* Everything one forward/backward over a (B,T) batch needs.  All pointers are device
 * workspaces allocated by the caller; sizes follow from nint_seq_workspace_sizes(). */
#define NINT_MAX_LAYERS 8
typedef struct nint_seq {
  int32_t dtype;            /* NINT_F32 | NINT_BF16: storage type ET of halo slabs / stash / weights */
  int32_t B, T, L;
  int32_t need_dx;          /* backward also produces d/dx of the input sequence */
  int32_t has_init_state;   /* 0: h0=c0=0 (reference ConvLSTM, model.py:259-262); 1: h[l][0], c0[l] given */
  int32_t n_cu;             /* CU count used to size split-K grids */
  int32_t zero_dstate;      /* backward: bit 2l = dc[l] is all-zero at entry, bit 2l+1 = dh[l] is all-zero at entry -- the first
                             * BPTT step then neither reads nor needs them zero-filled (reference: zero state grads) */
  nint_geom g;
  nint_layer layer[NINT_MAX_LAYERS];
  const void* xs;                      /* ET halo slab [T*B][Hh][Wh][Cxp0]: packed input sequence */
  void* h[NINT_MAX_LAYERS];            /* ET halo slab [(T+1)*B][Hh][Wh][Chp]: h[0]=initial, h[t+1]=h_t */
  float* c[NINT_MAX_LAYERS];           /* f32 compact [(T+1)*B][H][W][Chp]: c[0]=initial, c[t+1]=c_t */
  void* gates[NINT_MAX_LAYERS];        /* ET stash [T*B][H][W][4*Ch16] post-activation i,f,g,o (training only; may be NULL) */
  void* dG[NINT_MAX_LAYERS];           /* ET halo slab [T*B][Hh][Wh][4*Ch16] pre-activation gate grads */
  void* dh[NINT_MAX_LAYERS];           /* ET compact [B][H][W][Chp] running dL/dh_t (in: dL/dh_{T-1}, out: dL/dh_init) */
  float* dc[NINT_MAX_LAYERS];          /* f32 compact [B][H][W][Chp] running dL/dc_t */
  void* dx;                            /* ET compact [T*B][H][W][Cxp0] (need_dx only) */
  float* dW[NINT_MAX_LAYERS];          /* f32 OIHW (4Ch, Cx+Ch, k, k) gradient, overwritten */
  float* db[NINT_MAX_LAYERS];          /* f32 (4Ch) gradient, overwritten */
  float* wg_partial;                   /* f32 split-K slabs for wgrad: the SUM over the layers of nint_wgrad_workspace_bytes
                                        * (each rounded up to 256 bytes) -- the layers' slabs sit side by side */
  size_t wg_partial_bytes;
  int32_t fuse_bwd;                    /* backward schedule: 0 = per layer (fused BPTT step for the short-K layers), 1 = never
                                        * fused, 2 = every layer fused, 0x40000000 | masks = explicit: bit l: layer l runs the
                                        * fused step; bit 8+l: the fused layer l also runs the pointwise backward of the
                                        * classic layer l-1 on its x columns; bit 16+l: the CLASSIC layer l does that for the
                                        * classic layer l-1 (nint_cell_bwd_fused; same results up to the bf16 rounding of the
                                        * intermediate dh, which the fused step skips) */
  int32_t probe_mask;                  /* in-step timing probes (diagnostic; 0 = none): bit k brackets every launch of kind k
                                        * (NINT_PROBE_*) of nint_seq_fwd / nint_seq_bwd with two one-thread stamp launches */
  unsigned long long* probe;           /* device buffer of probe_slots {tag, s_memrealtime (100 MHz)} pairs, or NULL.  nint_seq_fwd
                                        * fills slots from 0, nint_seq_bwd from probe_slots / 2; each starts with two back-to-back
                                        * calibration stamps (kind 0).  tag = kind | layer << 8 | t << 16 | end << 31 */
  int32_t probe_slots;
  /* Independent launches as ONE grid (csrc/conv_igemm.hip: conv_lstm_multi[8]_kernel, conv_bwd_multi[8]_kernel,
   * conv_dgrad_multi8_kernel; every problem of a grid runs the body its own launch would have run).
   *   0  every launch by itself, time-major order (for t: for layer).
   *   1  forward: the (t, layer) wavefront (model.py:265-271: gate(l, t) needs gate(l-1, t) and gate(l, t-1) only, so gate(0, t+1),
   *      gate(1, t), gate(2, t-1) are independent) -- each wavefront step is one grid, T + L - 1 launches instead of T * L, every
   *      layer on the tiles its own launch takes; BPTT: the bottom layer's dgrad of one step with the top layer's fused step of
   *      the next (adjacent launches that share no buffer in a stack of three or more layers).  Bit-identical to 0.
   *   2  the forward wavefront only, every layer of a grid on 8-row tiles (the first layer's tile makes the grid a two-workgroups-
   *      per-CU kernel anyway; half the weight bytes per MFMA for the narrow layers: forward pass -5 % at B = 8,
   *      profiles/r04_c_wave_repeats.txt, r04_d_wave_rows8.txt).  = the time-major order with tile_rows pinned to 8 bit for bit, the
   *      default time-major order (4-row narrow tiles: another order of the K-slice partials) to f32 rounding.
   *   3  2 + the BPTT pair of 1 with the fused step on 8-row tiles (measured +0.2 ... +0.5 %; not used).
   *   4  the forward pass of 2; BPTT as TWO grids per step (profiles/r04_f_wave4.txt):
   *      (a) the bottom layer's dgrad of time u+1 waits for layer 1's dgrad of time u: one grid, the wide launch first, so that
   *          the narrow layer's workgroups fill its last round.  Both produce a piece of the bottom layer's d/dh of time u; each
   *          stores its own (layer 1 into dh[0], the bottom layer into the head of wg_partial, idle until the weight gradients)
   *          and the bottom layer's pointwise backward adds the two: f32 = the time-major order bit for bit (the same f32 sum);
   *          bf16: each piece is rounded before the f32 add instead of the running sum after it (layer 0's gradients move by
   *          ~1e-3 relative).  Needs wg_partial_bytes >= B*H*W*Chp[0]*es and classic (unfused) steps in layers 0 and 1;
   *      (b) with a fused top layer in a stack of three or more, the bottom layer's pointwise backward of time u waits for the
   *          top layer's fused step of time u-1 (the next launch, touching layers >= 1 only): one grid, the pointwise pass as a
   *          problem of the conv kernel (the same arithmetic: bit-identical).
   *      B = 2 / 4 / 8 at 100 x 154: +4.5 / +1.8 / +0.65 % over 2; B = 12 / 16 / 32: +1 ... +2 % over 0.
   *   5  the forward pass of 1 with the BPTT of 4 (B = 1 at 100 x 154: 661 against 647 samples/s of 1 and 622 of 4).
   * A shape the merged kernels do not hold (register-heavy fused shapes, more than 4 layers, launch shapes without a case) goes out
   * as separate launches.  Probes do not change the schedule: merged grids are bracketed as such (NINT_PROBE_WAVE,
   * NINT_PROBE_BWD_PAIR, NINT_PROBE_BWD_PW).  SeqEngine picks 5 for the smallest batches and 4 above (engine.py:_set_wave). */
  int32_t wave;
  /* nint_seq_bwd in two calls, for the data-parallel exchange: 0 = everything in one call; 1 = the BPTT chain and
   * the weight / bias gradients of layers >= 1 (their fold included); 2 = the weight / bias gradient of layer 0 only (dG[0] of
   * a preceding part-1 call on the same stream is its input).  Between the two the caller starts the all-reduce of everything
   * but layer 0's slice of the gradient bucket, which then runs under layer 0's weight gradient -- the largest launches of the
   * step -- instead of after them.  Same launches, same order inside each layer: bit-identical gradients. */
  int32_t bwd_parts;
} nint_seq;

/* launch kinds for nint_seq.probe_mask / the probe tags */
enum { NINT_PROBE_CAL = 0, NINT_PROBE_GATE = 1, NINT_PROBE_POINTWISE = 2, NINT_PROBE_DGRAD = 3, NINT_PROBE_FUSED = 4,
       NINT_PROBE_WGRAD = 5, NINT_PROBE_FOLD = 6,
       NINT_PROBE_WAVE = 7, /* a merged forward grid (nint_seq.wave): tag layer = number of gate launches in it, t = wavefront step */
       NINT_PROBE_BWD_PAIR = 8, /* a merged BPTT grid of two conv launches (wave = 4 / 5: the dgrad launches of layers 0 and 1; wave = 1 / 3:
                                 * the bottom dgrad with the top layer's fused step): tag layer = the second launch's layer, t = its time step */
       NINT_PROBE_BWD_PW = 9    /* wave = 4 / 5: the top layer's fused step with the bottom layer's pointwise backward: tag layer = the
                                 * fused layer, t = its time step */ };

/* ---- library / device ---------------------------------------------------------------- */
int nint_version(void);
const char* nint_error_string(int code);
/* host out-params */
int nint_device_info(int* n_cu, int* lds_bytes_per_cu, int* wave_size, char* name, int name_len);
/* Hardware self-test used by tests: MFMA fragment maps and ds_read_b64_tr_b16 semantics.
 * out: device buffer of >= 4096 floats. */
int nint_selftest(float* out, void* stream);

/* channel padding rule: KC = 16 (f32) or 32 (bf16) channels = 64 bytes per pixel per K-step */
int nint_kc(int dtype);
int nint_geom_make(nint_geom* g /*host*/, int H, int W, int P);

/* ---- layout conversion at the boundary ------------------------------------------------- */
/* x (B,T,C,H,W) f32 contiguous (model.py:255) -> halo slab image t*B+b.  Replaces the
 * `x[:, t]` slicing of model.py:266 and torch.cat of model.py:219 (never materialised). */
int nint_pack_btchw(const float* src, void* dst, int B, int T, int C, int Cp, const nint_geom* g,
                    int dtype, void* stream);
/* the same into a HORIZONTALLY FOLDED slab (nint_layer.xfold): channel kx*C + c of pixel x holds src[.., x + kx - k/2]
 * (zero outside the image: the convolution's own zero padding), Cp >= k*C */
int nint_pack_btchw_xfold(const float* src, void* dst, int B, int T, int C, int k, int Cp, const nint_geom* g,
                          int dtype, void* stream);
/* backward of that fold: compact ET slab [N][H][W][Cp] of d/d(folded x) -> d/dx (N,C,H,W) f32,
 * dx[c][x] = sum_kx dfold[x - kx + k/2][kx*C + c] */
int nint_unfold_dx(const void* src, float* dst, int N, int C, int k, int Cp, int H, int W, int dtype, void* stream);
/* halo slab images [n0, n0+N) -> (N,C,H,W) f32 */
int nint_unpack_halo(const void* src, float* dst, int n0, int N, int C, int Cp, const nint_geom* g,
                     int dtype, void* stream);
/* (N,C,H,W) f32 <-> compact f32 slab [N][H][W][Cp] */
int nint_pack_compact(const float* src, void* dst, int N, int C, int Cp, int H, int W, int dtype, void* stream);   /* dtype: element type of the compact slab */
int nint_unpack_compact(const void* src, float* dst, int N, int C, int Cp, int H, int W, int dtype, void* stream);

/* ---- weights ----------------------------------------------------------------------------- */
/* 1 when horizontally folding the x source of a (first) layer lowers its number of K-steps:
 * ceil(k*Cx / KC) * k < ceil(Cx / KC) * k * k  (see nint_layer.xfold); pure host arithmetic */
int nint_xfold_pays(int Cx, int k, int dtype);
/* bytes to reserve for EACH of the packed fwd / dgrad weight images of one layer */
size_t nint_packed_weight_bytes(int Cx, int Ch, int k, int dtype, int xfold);
/* W (4Ch, Cx+Ch, k, k) f32 OIHW + bias (4Ch) as nn.Conv2d stores them (model.py:207-211)
 * -> Wf, Wd (fragment order, ET) and bias_p.  Must be re-run after every optimiser step. */
int nint_pack_weights(const float* W, const float* bias, void* Wf, void* Wd, float* bias_p,
                      int Cx, int Ch, int k, int xfold, int dtype, void* stream);

/* the same for every layer of a model in ONE launch: W[l] / bias[l] (host arrays of device pointers; bias[l] may be
 * NULL) into layers[l].Wf / .Wd / .bias_p */
int nint_pack_weights_layers(const float* const* W /*host*/, const float* const* bias /*host*/,
                             const struct nint_layer* layers /*host*/, int L, int dtype, void* stream);

/* ---- the hot path: one cell step ----------------------------------------------------------- */
/* ConvLSTMCell.forward (model.py:216-231): gates = conv(cat[x,h]) ; sigmoid/tanh ; c,h update,
 * fused.  x_slab/h_prev halo slabs (image n0x.. / n0h..), c_prev/c_out compact, h_out halo slab,
 * gates_out stash (NULL in inference).  h_prev == NULL means h_prev = 0 (skips that half of K). */
int nint_cell_fwd(const nint_layer* ly /*host*/, const nint_geom* g /*host*/, int dtype, int N,
                  const void* x_slab, const void* h_prev, const float* c_prev,
                  void* h_out, float* c_out, void* gates_out, void* stream);

/* 1 when the vector-ALU STENCIL kernel (csrc/stencil.hip) holds this layer's gate step: tiny hidden widths (Ch <= 8: at most
 * 32 gate columns, no dense contraction), k = 3, thin input (<= 16 channels, or a folded first-layer input of <= 64 folded
 * channels).  nint_cell_fwd runs it for nint_layer.tile_rows == 1.  One lane per pixel, LDS-staged halo tile, DPP row
 * shifts for the horizontal taps, scalar-operand weights, the same LSTM epilogue.  Pure host arithmetic. */
int nint_stencil_holds(const nint_layer* ly /*host*/);

/* autograd backward of model.py:223-229 (pointwise part): consumes dh, dc (in place -> dc_prev),
 * the stashed gates and c_prev / c_new; writes pre-activation gate grads into the dG halo slab. */
int nint_cell_bwd_pointwise(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                            const void* gates, const float* c_prev, const float* c_new,
                            const void* dh, float* dc, void* dG, void* stream);

/* conv backward-data of model.py:220: d cat[x,h] = W^T (*) dG.  h columns are STORED to dh_prev,
 * x columns are ACCUMULATED (+=) into dx_accum (the layer below's dh, or dx); either may be NULL. */
int nint_conv_dgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                    const void* dG, void* dx_accum, void* dh_prev, void* stream);

/* One fused BPTT step of a cell: conv backward-data of time t+1 AND the pointwise backward of time t.
 *   d cat[x,h]_{t+1} = W^T (*) dG_next;  x columns are STORED to dx (compact, may be NULL: not computed);
 *   d/dh_t = (h columns) + dh_above  never goes to memory: it feeds the pointwise backward of time t (gates / c_prev /
 *   c_new = the stash of time t; c_prev == NULL means c_{t-1} = 0), which overwrites dc in place (-> d/dc_{t-1}) and
 *   writes the dG halo slab of time t.  dh_above: d/dh_t from the layer above (its x columns) or the head, compact
 *   [N][H][W][Chp] ET; NULL = 0.  Same results as nint_conv_dgrad + nint_cell_bwd_pointwise up to the bf16 rounding of
 *   the intermediate dh, which this entry skips. */
int nint_cell_bwd_fused(const nint_layer* ly, const nint_geom* g, int dtype, int N, const void* dG_next, void* dx,
                        const void* gates, const float* c_prev, const float* c_new, const void* dh_above,
                        float* dc, void* dG, void* stream);

/* conv backward-weight of model.py:220 over N = T*B images in ONE launch (all time steps):
 * dW[o][c][ky][kx] = sum_n,y,x dG[n,y,x,o] * cat[n,y+ky-p,x+kx-p,c] ; db[o] = sum dG.
 * partial: split-K workspace of nint_wgrad_workspace_bytes(). */
size_t nint_wgrad_workspace_bytes(const nint_layer* ly, int dtype, int n_cu);
int nint_conv_wgrad(const nint_layer* ly, const nint_geom* g, int dtype, int N,
                    const void* dG, const void* x_slab, const void* h_slab,
                    float* dW, float* db, float* partial, size_t partial_bytes, int n_cu, void* stream);
/* db rides along with the x source's weight gradient: the dG fragments are multiplied with an all-ones fragment on the
 * matrix pipe (one extra column of the split-K slab), so there is no separate pass over dG. */

/* ---- whole-sequence drivers (model.py:253-274 and its BPTT), all launches from C++ ---------- */
int nint_seq_fwd(const nint_seq* s /*host*/, void* stream);
int nint_seq_bwd(const nint_seq* s /*host*/, void* stream);

/* ---- 1x1 head (model.py:251,274) ------------------------------------------------------------ */
/* pred (N,O,H,W) f32 = w (O,Ch) . h + b  from halo-slab images [n0, n0+N) */
int nint_head_fwd(const void* h_slab, int n0, int N, int Ch, int Chp, int O, const float* w, const float* b,
                  float* pred, const nint_geom* g, int dtype, void* stream);
/* dh (compact ET [N][H][W][Chp], overwritten) ; dw (O,Ch), db (O) overwritten */
int nint_head_bwd(const void* h_slab, int n0, int N, int Ch, int Chp, int O, const float* w,
                  const float* dpred, void* dh, float* dw, float* db, const nint_geom* g, int dtype,
                  float* scratch, size_t scratch_bytes, void* stream);
/* scratch (may be NULL): >= 256*O*(Ch+1) floats enables the tiled two-stage weight-gradient path. */

/* ---- loss (train.py:102,105) ------------------------------------------------------------------ */
/* pred (N,O,H,W) f32, y (N,O,Hc,Wc) f32; crop window [oy,oy+Hc) x [ox,ox+Wc).
 * loss_out: NINT_LOSS_SCRATCH_FLOATS floats, 8-byte aligned; [0] = mean((y-p)^2) + mean(|y-p|), the
 * rest is reduction scratch (4 doubles per workgroup of the partial-sum launch, up to 1024 workgroups); dpred (N,O,H,W) = d loss / d pred (0 outside the
 * crop), may be NULL; stats (NINT_LOSS_STATS doubles, accumulated, caller zeroes): [0..4] pooled sums
 * sum (y-p)^2, sum |y-p|, sum y, sum y^2, count; [5..7] the reference's per-batch statistics: sum over calls
 * of the loss, sum over calls of sklearn-style r2_score(y, pred) of that call, number of calls -- the
 * device-side accumulators replacing loss.item() / r2_score(...cpu()) of train.py:113-117, utils.py:73-75. */
#define NINT_LOSS_SCRATCH_FLOATS 8194
#define NINT_LOSS_STATS 8
int nint_loss_mse_l1_crop(const float* pred, const float* y, float* dpred, float* loss_out, double* stats,
                          int N, int O, int H, int W, int oy, int ox, int Hc, int Wc, void* stream);

/* Training fast path: head forward + crop + loss + d loss/d pred + head backward-data in ONE pass over the pixels
 * (train.py:96-109 around model.py:274); the same arithmetic in the same order as nint_head_fwd ->
 * nint_loss_mse_l1_crop -> nint_head_bwd(dh).  pred is not materialised; dpred (N,O,H,W) is written for
 * nint_head_bwd(dh = NULL) to form dw / db.  Chp <= 64, else NINT_E_SHAPE (use the three separate entries). */
int nint_head_loss_fused(const void* h_slab, int n0, int N, int Ch, int Chp, int O, const float* w, const float* b,
                         const float* y, float* dpred, void* dh, float* loss_out, double* stats, const nint_geom* g,
                         int oy, int ox, int Hc, int Wc, int dtype, void* stream);

/* ---- optimiser (train.py:71,110) ---------------------------------------------------------------- */
/* torch.optim.Adam (eps 1e-8, no weight decay / amsgrad) on one flat f32 buffer.
 * grad_scale multiplies g first (1/world_size after the RCCL all-reduce). step is 1-based. */
int nint_adam_flat(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1,
                   double beta2, double eps, int step, float grad_scale, void* stream);

/* ---- preproc (dataset.py:520-536, 61-98) --------------------------------------------------------- */
/* Fuse on the channel axis, z-score with mean/std (C = sum lev floats, device), cyclic-lon + lat halo pad.
 * mode 0 = the committed reference behaviour (np.fliplr on the channel axis, dataset.py:96),
 * mode 1 = true latitude reflect (dataset.py:51 semantics).
 * srcs: host array of nsrc device pointers; lev: host array of levels per source (u,v,omega at L levels,
 * prec and emission at 1).
 *
 * nint_preproc_fuse_pad      : ONE sample; srcs[i] points at the window's first time step, (T, lev_i, H, W) f32
 *                              -> out (T, C, Hp, Wp) f32   (the Dataset.__getitem__ result, dataset.py:538-539).
 * nint_preproc_fuse_pad_batch: B samples in one launch; srcs[i] is the whole RECORD (n_steps, lev_i, H, W) and
 *                              sample b reads the time steps [t0[b], t0[b]+T) (the sliding window of
 *                              dataset.py:614-616 as a pointer offset) -> out (B, T, C, Hp, Wp) f32.
 * nint_preproc_fuse_pad_slab : the same B windows written STRAIGHT into the model's input halo slab
 *                              (image t*B+b, ET = dtype, channels-last, channel padding zeroed) on the padded
 *                              grid g->H x g->W = Hp x Wp: the f32 NCHW tensor of dataset.py:538 and the
 *                              nint_pack_btchw pass never exist.  Values are the f32 result rounded once to ET.
 * t0: host array of B non-negative window starts. */
#define NINT_PRE_MAX_B 64   /* samples per launch (larger batches are split internally) */
int nint_preproc_fuse_pad(const float* const* srcs /*host*/, const int* lev /*host*/, int nsrc,
                          const float* mean, const float* std, float* out, int T, int H, int W,
                          int Hp, int Wp, int mode, void* stream);
int nint_preproc_fuse_pad_batch(const float* const* srcs /*host*/, const int* lev /*host*/, int nsrc,
                                const float* mean, const float* std, const int* t0 /*host*/, int B, float* out,
                                int T, int H, int W, int Hp, int Wp, int mode, void* stream);
int nint_preproc_fuse_pad_slab(const float* const* srcs /*host*/, const int* lev /*host*/, int nsrc,
                               const float* mean, const float* std, const int* t0 /*host*/, int B, void* xs_slab,
                               int Cxp, int xfold_k /* 0: plain slab; k: horizontally folded for kernel size k */,
                               int T, int H, int W, const nint_geom* g /*host*/, int mode, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NINT_H */
