/*
 * radmmm_hip.h -- C ABI of libradmmm_hip.so: hand-written gfx950 (MI355X) kernels for
 * the RAD-MMM flow-decoder training step.
 *
 * The reference (NVIDIA/RAD-MMM) is pure Python/PyTorch and has NO C/FFI boundary
 * (SURVEY.md §8b): its plug-in mechanism is jsonargparse `class_path`
 * (configs/RADTTS_model_config.yaml:16-48, tts_main.py:64-65).  The drop-in is
 * therefore the Python package `rad_mmm_amd` (same ctor kwargs / forward signature /
 * state_dict names as decoders.RADMMMFlow and loss.RADMMMLoss); THIS header is the
 * boundary underneath it, i.e. what a binding for the reference's operators would
 * bind.  Each entry point cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; radmmm_last_error() returns a
 *    thread-local message for the last failure on the calling thread.
 *  - all pointers are DEVICE pointers into caller-owned memory (no ownership
 *    transfer, no allocation inside the library, no internal threads, no global
 *    mutable state apart from the error string).
 *  - every call takes the hipStream_t to launch on (as void*) and is asynchronous.
 *  - activations are CHANNELS-LAST fp32: a [B, C, T] tensor of the reference is held
 *    as a row-major matrix [B*T rows][ld floats] with row r = b*T + t.  ld % 4 == 0
 *    and 16-byte aligned bases are required wherever a matrix feeds a GEMM.
 *  - `lens` is an int32 device array [B] of valid frames per item (already divided by
 *    n_group_size); NULL means all T frames valid.
 */
#ifndef RADMMM_HIP_H
#define RADMMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RADMMM_ABI_VERSION 4

typedef void* radmmm_stream_t; /* hipStream_t */

const char* radmmm_last_error(void);
int radmmm_abi_version(void);

/* activation codes (forward) / derivative-from-output codes (backward) */
enum { RADMMM_ACT_NONE = 0, RADMMM_ACT_SOFTPLUS = 1, RADMMM_ACT_RELU = 2, RADMMM_ACT_LEAKY = 3 };
/* scaling functions of the affine coupling, common.py:1127-1140 */
enum { RADMMM_SCALE_TANH = 0, RADMMM_SCALE_EXP = 1, RADMMM_SCALE_SIGMOID = 2, RADMMM_SCALE_TRANSLATE = 3 };

/* ------------------------------------------------------------------------------------
 * Split formats of the row-major operand copies (one K-contiguous row per frame / output channel):
 *   RADMMM_SPLIT_F16  two half arrays [rows][ld]: hi = fp16(t), lo = fp16(t - hi), t = scale * x clamped to +-60000
 *   RADMMM_SPLIT_X8A  hi as above + the 8-bit CROSS array in place of lo, same row pitch (2*ld bytes), ld % 32 == 0:
 *                     per 32 columns 64 bytes [ e4m3(t * 2^e) x 32 | e4m3((t - hi) * 2^(11+e)) x 32 ]  (A role)
 *   RADMMM_SPLIT_X8B  the same with the two 32-byte halves swapped [ lo8 | hi8 ]                     (B role, weights)
 * e4m3 = OCP FP8 E4M3, round to nearest even, saturated at +-448.  A block-scaled FP8 MFMA of an X8A fragment with an
 * X8B fragment yields Ah.Bl + Al.Bh (radmmm_rowgemm_h3 with nprod = 2).
 * ------------------------------------------------------------------------------------ */
#define RADMMM_SPLIT_F16 0
#define RADMMM_SPLIT_X8A 1
#define RADMMM_SPLIT_X8B 2
/* options of a split producer: format, exponent e of the 8-bit parts, optional saturation flag (device int, OR-ed with
 * bit 0 when |scale * x| > 60000 was clamped; with bit 1 when an element's 8-bit parts left e4m3's range (|scale * x| *
 * 2^e > 448: that element keeps single-product accuracy) and then also, one-hot, with bit 1 + L, L = ceil(log2(|scale * x|
 * * 2^e / 448)) clamped to 1 .. 6: by how many powers of two e was too large) */
typedef struct { int fmt; int x8_exp; int32_t* sat_flag;
                 void* lo16; /* optional, 8-bit formats only: the fp16 lo part fp16(s*x - hi) as well, same pitch as the hi
                                array (radmmm_wn_input_fwd, radmmm_dact_mul_transposed): feeds radmmm_wgrad_rm */
} radmmm_split_opts;

/* ------------------------------------------------------------------------------------
 * Row GEMM with taps: the Conv1d family in channels-last form, fp32 MFMA
 * (v_mfma_f32_32x32x2_f32, exact fp32).
 *
 *   acc[r, n] = sum_{tap} sum_{k<K} Am[r + shift(tap), k] * Bt[tap](k, n)
 *   shift(tap) = sign * (tap - taps/2) * dil
 *   Am[row, :] = A[row, :] if the shifted frame lies in the same item and inside
 *                [0, T) (a_mask_mode 0) or [0, lens[b]) (a_mask_mode 1), else 0
 *   Bt[tap](k, n) = B[tap*b_tap_stride + n*ldb + k]   (b_layout 0: [N][K], forward)
 *                 = B[tap*b_tap_stride + k*ldb + n]   (b_layout 1: [K][N], data-grad)
 *
 * Epilogue, applied in this order to v = acc:
 *   pconv     : v *= taps_r / (cnt + 1e-6), cnt = #taps of a (ratio_taps, ratio_dil)
 *               window centred on r whose frame is < lens[b]  (0 if cnt == 0)
 *   premask   : v *= [t < lens[b]]
 *   bias      : v += bias[n]
 *   add       : v += add[r*ldadd + n]
 *   postmask  : v *= [t < lens[b]]
 *   dact      : v *= act'(y) expressed from the saved OUTPUT y = dact_src[r*lddact+n]
 *               (softplus: 1 - exp(-y); relu: y > 0; leaky: y > 0 ? 1 : 0.01)
 *   rowscale  : 1: v *= mask ; 2: v *= mask * ratio (same ratio definition as pconv)
 *   act       : v = act(v)
 *   C[r*ldc+n] = v ;  if C2: C2[r*ldc2+n] = (c2_accum ? C2[...] : 0) + v
 *
 * Replaces: F.conv1d / PartialConv1d / ConvNorm / weight-normed 1x1 convs and their
 * autograd data-gradients (common.py:179-191, 816-835; partialconv1d.py:58-94), and
 * the invertible 1x1 channel mix (common.py:546, 615).
 * ------------------------------------------------------------------------------------ */
typedef struct {
  const float* A; int lda;
  int64_t a_item_stride; /* 0: row r at A + r*lda; else item b, frame t at A + b*a_item_stride + t*lda
                            (rows of one item may then overlap: lda < K is allowed, STFT framing) */
  const float* B; int ldb; int64_t b_tap_stride; int b_layout;
  float* C; int ldc;      /* radmmm_rowgemm_h3: may be NULL when Ch is given (no fp32 copy of the result is written) */
  int M, N, K;
  int taps, dil, sign;
  int T;                 /* frames per item; M must be a multiple of T */
  const int32_t* lens;   /* [M/T] or NULL */
  int a_mask_mode;
  const float* bias;
  int pconv, premask, postmask;
  int ratio_taps, ratio_dil;
  const float* add; int ldadd;
  const float* dact_src; int lddact; int dact;
  int rowscale;
  int act;
  float* C2; int ldc2; int c2_accum;
  /* optional split-fp16 copies of the outputs (operands of the next split-f16 GEMM):
   * Ch/Cl [M][ldch] halves receive hi/lo of ch_scale * C, C2h/C2l of c2h_scale * C2 (ld % 4 == 0) */
  void* Ch; void* Cl; int ldch; float ch_scale;
  void* C2h; void* C2l; int ldc2h; float c2h_scale;
  /* split format of Ch/Cl and C2h/C2l (RADMMM_SPLIT_*, see "split formats" below) and the exponents of their 8-bit parts */
  int split_fmt; int ch_x8_exp; int c2h_x8_exp;
  int32_t* sat_flag;     /* optional device int: saturation report of the split outputs, bits as in radmmm_split_opts.sat_flag */
  void* Clo;             /* optional, with Ch and an 8-bit split_fmt: [M][ldch] halves, the fp16 lo part fp16(ch_scale*v - Ch)
                            as well (the row-major pair Ch / Clo is what radmmm_wgrad_rm contracts) */
  /* optional (radmmm_rowgemm_h3 only): colsum_out[n] = sum over the rows r inside their utterance's length (all rows when
   * the launch has no row scale) of the epilogue's value BEFORE its row scale -- x_r[n] of step (6) below, after the
   * act' factor.  With rowscale = 2 this is the bias gradient of the partial conv whose data gradient the launch computes
   * (common.py:179-191 backward), i.e. radmmm_colsum(C, row_weight 2) without the pass over C: the kernels with the direct
   * epilogue sum it from the accumulators (one partial row per row tile in colsum_scratch, added up in a fixed order:
   * deterministic), every other path runs radmmm_colsum on C afterwards.  colsum_scratch: device scratch of
   * radmmm_rowgemm_h3_colsum_scratch_floats(M, N) floats. */
  float* colsum_out; float* colsum_scratch;
  /* optional (radmmm_rowgemm_h3; ABI 3): the dact step reads the saved OUTPUT y from its row-major 8-bit split copy instead
   * of an fp32 array -- dact_h [M][lddact_h] fp16 hi, dact_x the RADMMM_SPLIT_X8A cross array of the same tensor, both
   * written with scale 1 and exponent dact_x8_exp:  y = hi + e4m3_lo * 2^-(11 + dact_x8_exp)   (within 2^-15 |y| of the
   * fp32 value the pair was split from; the softplus derivative 1 - exp(-y) then carries <= 1.2e-5 absolute).  dact_src
   * must be NULL then.  With it a producer need not keep an fp32 copy of an activation beside its split pair:
   * radmmm_rowgemm_h3 accepts C == NULL when Ch is given (the split copy alone carries the result). */
  const void* dact_h; const void* dact_x; int lddact_h; int dact_x8_exp;
  /* optional (ABI 4): the value that goes to C2 / C2h is  ((c2_src[0] + c2_src[1]) + c2_src[2]) + y  instead of C2 + y --
   * n_c2_src = 1 .. 3 fp32 arrays [M][ldc2] (0: the read-modify-write of C2 as before; c2_accum is ignored otherwise).  The
   * LAST res/skip layer of a WN adds the earlier layers' outputs itself (same association as the running sum, so the same
   * bits), and the earlier layers write their own output only: the skip sum is neither read nor written four times.  C2
   * may then be NULL when C2h is given (only the split copy of the sum is kept). */
  const float* c2_src[3]; int n_c2_src;
} radmmm_rowgemm_desc;

int radmmm_rowgemm_f32(const radmmm_rowgemm_desc* d, radmmm_stream_t stream);

/* Same operation on the f16 matrix cores with fp32-class accuracy by operand splitting
 * (x*s = hi + lo in fp16;  A.B ~= Ah.Bh + Ah.Bl + Al.Bh, fp32 accumulate; measured 2.2e-6 max
 * rel. error at K=5120 vs 3.8e-6 for the fp32 MFMA, at 2.3x its rate).  d->A / d->B are ignored;
 * the operands are the split copies below, both K-CONTIGUOUS:
 *   Ah/Al [M][lda_h] halves (rows = frames, same shift/masking rules as radmmm_rowgemm_f32)
 *   Bh/Bl [taps][N][ldb_h] halves (tap stride b_tap_stride_h halves); b_layout must be 0 --
 *   the data-gradient uses a transposed split copy of the weights instead of layout 1.
 * acc is multiplied by acc_scale (= 1/(a_scale*b_scale)) before the epilogue.
 * Requires K % 32 == 0, lda_h/ldb_h % 8 == 0. */
typedef struct {
  radmmm_rowgemm_desc base;
  const void* Ah; const void* Al; int lda_h;
  const void* Bh; const void* Bl; int ldb_h; int64_t b_tap_stride_h;
  float acc_scale;
  int nprod;                  /* 0 or 3: split products Ah.Bh + Ah.Bl + Al.Bh (fp32-class accuracy);
                                 1: Ah.Bh only = plain fp16 operands, fp32 accumulate (16-bit throughput mode);
                                 2: Ah.Bh on the f16 pipe + both cross terms in ONE block-scaled FP8 MFMA per 32-deep k
                                    step: Al / Bl are then the 8-bit cross arrays RADMMM_SPLIT_X8A / RADMMM_SPLIT_X8B */
  int a8_exp, b8_exp;         /* nprod 2: exponents e the 8-bit parts of A / B were written with (values * 2^e) */
  /* optional EXTRA K segment after the taps: one more "tap" with shift 0 whose A rows start extra_a_rows rows below row 0
   * of Ah/Al (a second activation matrix stored behind the first in the same allocation, same lda_h) and whose weights
   * are tap index `taps` of Bh/Bl: acc += A2 . B[taps].  Sums two GEMMs that share their output (the data gradient of a
   * k-tap conv and of a 1x1 conv reaching the same tensor) in one launch, without the intermediate tensor. */
  int extra_tap; int extra_a_rows;
} radmmm_rowgemm_h3_desc;

int radmmm_rowgemm_h3(const radmmm_rowgemm_h3_desc* d, radmmm_stream_t stream);
int64_t radmmm_rowgemm_h3_colsum_scratch_floats(int M, int N);
/* Deferred column sums (ABI 3): a launch with colsum_scratch set and colsum_out == NULL leaves its per-row-tile partial rows
 * in colsum_scratch ([rows][N] floats, rows = the value returned here) and the caller adds them up later -- several launches'
 * finals in one radmmm_colsum_final_multi call.  Returns 0 when this descriptor would take a kernel that cannot leave
 * partials (the generic epilogue, the narrow kernel): give it colsum_out then. */
int radmmm_rowgemm_h3_colsum_rows(const radmmm_rowgemm_h3_desc* d);
typedef struct { const float* part; float* out; int nparts; int cols; } radmmm_cs_item;
/* out[c] = sum_p part[p * cols + c] for every item (fixed order per item: deterministic), any number of items */
int radmmm_colsum_final_multi(const radmmm_cs_item* items, int n, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Weight-gradient GEMM (contraction over frames), fp32 MFMA:
 *   P[split][tap][m][n] = sum_{r in split} GY[r, m] * Xm[r + shift(tap), n]
 * m < Mc (out channels), n < Nc (in channels); Xm masked as in radmmm_rowgemm_f32.
 * `splits` partial slabs are written (deterministic; the consumer sums them),
 * P slab layout [taps][Mc][ldp].  Replaces conv weight-gradients of autograd
 * (aten::convolution_backward) for common.py:816-835.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  const float* GY; int ldgy;
  const float* X; int ldx;
  float* P; int ldp; int64_t split_stride;  /* floats between slabs */
  int R;                 /* total rows (B*T) */
  int Mc, Nc;
  int taps, dil;
  int T; const int32_t* lens; int x_mask_mode;
  int splits;
} radmmm_wgrad_desc;

int radmmm_wgrad_f32(const radmmm_wgrad_desc* d, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Weight norm fold / unfold  (torch weight_norm dim=0; common.py:173-174,791,813)
 *   forward : W[tap][co][col(ci)] = g[co] * v[co][ci][tap] / ||v[co]||_2 ,
 *             inv_norm[co] = 1/||v[co]||
 *             col(ci) = ci < perm_split ? ci + off_lo : ci - perm_split + off_hi
 *             (columns not hit by col() are left untouched: zero them once)
 *   backward: from `splits` slabs of dL/dW (layout of W) -> dL/dv [co][ci][tap], dL/dg [co] (+ *poison)
 * v is in the reference's checkpoint layout [Cout][Cin][taps].
 * ------------------------------------------------------------------------------------ */
int radmmm_weightnorm_fwd(const float* v, const float* g, float* W, float* inv_norm,
                          int Cout, int Cin, int taps, int ldw,
                          int perm_split, int off_lo, int off_hi, radmmm_stream_t stream);
int radmmm_weightnorm_bwd(const float* v, const float* g, const float* inv_norm,
                          const float* dW, int splits, int64_t split_stride,
                          float* dv, float* dg,
                          int Cout, int Cin, int taps, int ldw,
                          int perm_split, int off_lo, int off_hi,
                          const float* poison /* optional device scalar added to every dg element: 0, or NaN when the
                                                 pass's incoming gradient was not finite (the split operands clamp NaN / Inf
                                                 to finite values; this puts the non-finite result back where an AMP
                                                 GradScaler or a gradient-norm clip looks for it) */,
                          radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * WN input assembly (cat((z0, context), 1), common.py:819) in channels-last, K-padded:
 *   X0[r, 0:D] = ctx[r, 0:D] ; X0[r, D:D+h] = z[r, 0:h] ; X0[r, D+h:ldx0] = 0
 * and its gradient scatter:
 *   gctx[r, 0:D] (+)= gX0[r, 0:D] ; gz[r, 0:h] += gX0[r, D:D+h]
 * ------------------------------------------------------------------------------------ */
int radmmm_wn_input_fwd(const float* ctx, int ldctx, const float* z, int ldz,
                        float* X0 /* may be NULL when the split copy is written (ABI 3) */, int ldx0, int rows, int D, int h,
                        void* X0h, void* X0l /* optional split copy, pitch ldx0, may be NULL */,
                        const radmmm_split_opts* so, radmmm_stream_t stream);
int radmmm_wn_input_bwd(const float* gX0, int ldx0, float* gctx, int ldctx, int ctx_accum,
                        float* gz, int ldz, int rows, int D, int h, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Squeeze = nn.Unfold(kernel=(g,1), stride=g) of the reference (decoders.py:118-122,178; models/radmmm.py:114-120)
 * straight from the reference layout into channels-last rows of a wider matrix:
 *   out[(b*T' + t') * ld + col0 + c*g + k] = in[(b*C + c)*T + t'*g + k],  T' = T / g (frames beyond g*T' are dropped)
 * so the flow variable and the context LSTM's input are assembled without a permuted copy or a concatenation.
 * radmmm_unsqueeze_rows is its gradient: gin[b, c, t] = gout[row(t / g)][col0 + c*g + t % g] for t < g*T', else 0.
 * g must divide 64.
 * ------------------------------------------------------------------------------------ */
int radmmm_squeeze_rows(const float* in /* [B, C, T] */, float* out, int B, int C, int T, int g, int ld, int col0,
                        radmmm_stream_t stream);
int radmmm_unsqueeze_rows(const float* gout, float* gin /* [B, C, T] */, int B, int C, int T, int g, int ld, int col0,
                          radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Affine coupling (AffineTransformationLayer.forward, common.py:1163-1185):
 *   su = O[r, 0:h], b = O[r, h:2h]; (s, log_s) = scaling(su)
 *   zout[r, 0:h] = z[r, 0:h]; zout[r, h:2h] = s * z[r, h:2h] + b; zout[r, 2h:ldz] = z[...]
 *   log_s[r, 0:h] stored with row stride h.
 * backward: given gzout, glog_s (may be NULL), O, z:
 *   gO[r, 0:h] = (gzout1 * z1 + glog_s * dlog_s/ds) * ds/dsu ; gO[r, h:2h] = gzout1
 *   gz[r, 0:h] = gzout[r, 0:h]; gz[r, h:2h] = gzout1 * s ; gz[r, 2h:ldz] = gzout[...]
 * ------------------------------------------------------------------------------------ */
int radmmm_affine_coupling_fwd(const float* O, int ldo, const float* z, int ldz,
                               float* zout, float* log_s, int rows, int h, int scaling,
                               radmmm_stream_t stream);
int radmmm_affine_coupling_bwd(const float* O, int ldo, const float* z, int ldz,
                               const float* gzout, const float* glog_s,
                               float* gO, float* gz, int rows, int h, int scaling,
                               radmmm_stream_t stream);

/* y[r, c] = g[r, c] * act'(saved[r, c]) * w(r), c < cols: derivative of the activation from its
 * saved OUTPUT (dact 0: none, saved may be NULL) times a row weight (rowscale 0: 1; 1: mask;
 * 2: mask * partial-conv ratio of a (taps, dil) window) -- the step from dL/d(act output) to
 * dL/d(conv accumulator) of ConvNorm/PartialConv1d (common.py:179-191) */
int radmmm_dact_mul(const float* g, int ldg, const float* saved, int lds, float* y, int ldy,
                    int rows, int cols, int dact, int rowscale, int T, const int32_t* lens,
                    int taps, int dil,
                    void* yh, void* yl, int ldyh, float yscale /* optional split-fp16 copy of yscale*y */,
                    const radmmm_split_opts* so /* NULL: f16 pair, no flag */, radmmm_stream_t stream);

/* out[c] = sum_r w(r) * f(X[r, c]), f = identity or square;  row_weight 0: 1 ; 1: [t < lens[b]] ;
 * 2: (cnt+1e-6)/taps over valid rows (undoes the partial-conv ratio: bias gradient of
 * PartialConv1d).  Also the masked first/second moments of MaskedBatchNorm1d
 * (maskedbatchnorm1d.py:83-84).                                                          */
int radmmm_colsum(const float* X, int ldx, float* out, float* scratch, int rows, int cols,
                  int row_weight, int T, const int32_t* lens, int taps, int dil, int square,
                  radmmm_stream_t stream);
int64_t radmmm_colsum_scratch_floats(int rows, int cols);

/* ------------------------------------------------------------------------------------
 * Flow NLL reductions (compute_flow_loss, loss.py:85-110) on arbitrary-stride
 * [B, C, T] views:  mode 0: sum(x*m), mode 1: sum((x*m)^2);   m = [t < lens[b]].
 * Deterministic two-stage reduction; `out` receives one float.
 * backward (elementwise): gx[b,c,t] = coef[0] * m * (mode 1 ? 2*x : 1); gx uses x's strides
 * ------------------------------------------------------------------------------------ */
int radmmm_masked_reduce(const float* x, int B, int C, int T, int64_t sb, int64_t sc, int64_t st,
                         const int32_t* lens, int mode, float* out, float* scratch,
                         radmmm_stream_t stream);
int64_t radmmm_masked_reduce_scratch_floats(int B, int C, int T);
int radmmm_masked_reduce_bwd(const float* x, int B, int C, int T, int64_t sb, int64_t sc,
                             int64_t st, const int32_t* lens, int mode, const float* coef,
                             float* gx, radmmm_stream_t stream);

/* fused_add_tanh_sigmoid_multiply (common.py:66-73): y[r, c] = tanh(a+b)[r, c] *
 * sigmoid(a+b)[r, n + c], c < n.  Not on any config's path (WaveNetOriginal only). */
int radmmm_fused_add_tanh_sigmoid_multiply(const float* a, const float* b, int ld, float* y,
                                           int ldy, int rows, int n, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Piecewise-quadratic spline coupling transform (splines.py:241-339, forward branch)
 * on x [rows, h] (already normalised to [0,1) by (z+bound)/(2 bound)), q [rows, h*(2K+1)]
 * channels-last with channel index c*(2K+1)+j (first K: widths, last K+1: heights).
 *   y [rows, h], logj_sum[0:rows] = sum_c logj; logj_sum must have room for rows + rows*h
 *   floats (the tail receives the per-element log-jacobians the sums are reduced from)
 * backward: gy [rows,h], glogj [rows] -> gx [rows,h], gq [rows, h*(2K+1)]
 * ------------------------------------------------------------------------------------ */
int radmmm_pq_spline_fwd(const float* x, int ldx, const float* q, int ldq, float* y, int ldy,
                         float* logj_sum, int rows, int h, int K, radmmm_stream_t stream);
/* INDEX accounting of the forward bin search (splines.py:300-306 `searchsorted`), for parity tests: bins[r*h + c] = the bin
 * radmmm_pq_spline_fwd picks for element (r, c), or -1 for an element outside [0, 1) (passed through); edge_l / edge_r = the
 * two edges of that bin as the search compared them (the kernel's running sum of the softmax widths; 1 for the last bin). */
int radmmm_pq_spline_bins(const float* x, int ldx, const float* q, int ldq, int32_t* bins, float* edge_l, float* edge_r,
                          int rows, int h, int K, radmmm_stream_t stream);
/* inverse direction of the same transform (splines.py:327-339; no log-jacobian): x from y, both in [0,1) units */
int radmmm_pq_spline_inv(const float* y, int ldy, const float* q, int ldq, float* x, int ldx, int rows, int h, int K,
                         radmmm_stream_t stream);
int radmmm_pq_spline_bwd(const float* x, int ldx, const float* q, int ldq, const float* gy,
                         int ldgy, const float* glogj, float* gx, int ldgx, float* gq, int ldgq,
                         int rows, int h, int K, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * FiLM residual block tail (FiLMResBlock.forward, common.py:728-735) with the training-mode
 * masked batch-norm (maskedbatchnorm1d.py:53-118) fused in, channels-last [rows, C]:
 *   y = use_bn ? (h2 - mean) * invstd * w + b : h2 ;  t = y * (c1[:, 0:C] + 1) + c1[:, C:2C]
 *   out = 0.5 * (leaky_relu(t) + x1r)
 * mean/invstd are the caller's masked statistics (radmmm_colsum with square = 0/1).
 * backward: gout -> gh2, gc1 [rows, 2C], gx1r, and dL/dw, dL/db of the batch-norm affine;
 * n_valid = number of unmasked frames; scratch: radmmm_film_bwd_scratch_floats(rows, C).
 * ------------------------------------------------------------------------------------ */
int radmmm_film_fwd(const float* h2, int ldh, const float* c1, int ldc, const float* x1r, int ldx,
                    const float* mean, const float* invstd, const float* w, const float* b,
                    float* out, int ldo, int rows, int C, int use_bn, radmmm_stream_t stream);
int radmmm_film_bwd(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                    const float* mean, const float* invstd, const float* w, const float* b,
                    float n_valid, int T, const int32_t* lens, float* gh2, int ldgh, float* gc1,
                    int ldgc, float* gx1r, int ldgx, float* gw, float* gb, float* scratch, int rows,
                    int C, int use_bn, radmmm_stream_t stream);
int64_t radmmm_film_bwd_scratch_floats(int rows, int C);
/* radmmm_film_bwd in two halves, for synchronised masked batch-norm statistics under data parallelism
 * (maskedbatchnorm1d.py:88-95; MaskedBatchNorm1d.distributed_sync, tts_lightning_modules.py:238-243): _sums leaves this
 * rank's S = [sum g_y | sum g_y xhat] ([2][C]) at scratch + ceil(rows/64)*2*C ... (radmmm_film_bwd_scratch_floats - 2*C)
 * and copies it to gb / gw; the caller all-reduces S in place; _apply uses it with n_valid = the global frame count. */
int radmmm_film_bwd_sums(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                         const float* mean, const float* invstd, const float* w, const float* b, float* gw, float* gb,
                         float* scratch, int rows, int C, radmmm_stream_t stream);
int radmmm_film_bwd_apply(const float* h2, int ldh, const float* c1, int ldc, const float* gout, int ldg,
                          const float* mean, const float* invstd, const float* w, const float* b, float n_valid, int T,
                          const int32_t* lens, float* gh2, int ldgh, float* gc1, int ldgc, float* gx1r, int ldgx,
                          const float* scratch, int rows, int C, radmmm_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Alignment attention (ConvAttention.forward, common.py:1262-1277), from projected
 * queries Q [B, T1, Ca] and keys Kx [B, T2, Ca] (channels-last):
 *   d[b,t,s] = -temp * sum_c (Q[b,t,c]-K[b,s,c])^2
 *   logprob = log_softmax_s(d) + log(prior + 1e-8)   (prior NULL: logprob = d)
 *   attn = softmax_s(logprob masked to s < in_lens[b])
 * outputs attn, logprob [B, T1, T2].  backward: gattn, glogprob -> gQ, gK.
 * ------------------------------------------------------------------------------------ */
int radmmm_attn_fwd(const float* Q, const float* Kx, const float* prior, const int32_t* in_lens,
                    float* attn, float* logprob, int B, int T1, int T2, int Ca, float temp,
                    radmmm_stream_t stream);
int radmmm_attn_bwd(const float* Q, const float* Kx, const float* prior, const int32_t* in_lens,
                    const float* attn, const float* logprob, const float* gattn,
                    const float* glogprob, float* gQ, float* gK, float* gd_scratch /* B*T1*T2 */,
                    int B, int T1, int T2, int Ca, float temp, radmmm_stream_t stream);

/* Monotonic alignment search, width 1 (alignment.py:31-59; INTEGER/INDEX path, bit-exact
 * given identical log inputs).  logp [B, T1, T2] = log(attn) (caller's log), per item the
 * [out_lens[b], in_lens[b]] corner is aligned; hard [B, T1, T2] receives 0/1 (zeroed
 * outside).  scratch: radmmm_mas_scratch_bytes(B, T1, T2). */
int radmmm_mas_width1(const float* logp, const int32_t* in_lens, const int32_t* out_lens,
                      float* hard, void* scratch, int B, int T1, int T2, radmmm_stream_t stream);
/* Same search on PROBABILITIES attn [B, T1, T2] (TTSModel.binarize_attention, tts_lightning_modules.py:270-284,
 * hands alignment.py:31 the soft attention and the log is taken inside, alignment.py:36): the kernel takes the log
 * itself, correctly rounded to fp32 (a first launch over the whole chip, into `scratch`; the search itself is a chain
 * of T1 dependent steps per utterance and keeps nothing but an LDS row exchange and a ballot in each). */
int radmmm_mas_width1_prob(const float* attn, const int32_t* in_lens, const int32_t* out_lens,
                           float* hard, void* scratch, int B, int T1, int T2, radmmm_stream_t stream);
int64_t radmmm_mas_scratch_bytes(int B, int T1, int T2);

/* STFT magnitude -> mel -> log-clamp (audio_processing.py:137-154, 227-255).
 * audio [B, S]; basis [2*(n_fft/2+1)][n_fft] windowed DFT rows (re then im);
 * mel_basis [n_mel][n_fft/2+1]; out mel [B, n_mel, 1 + S/hop] (reference layout). */
int radmmm_stft_mel(const float* audio, const float* basis, const float* mel_basis, float* mel,
                    float* scratch, int B, int S, int n_fft, int hop, int n_mel,
                    float clip, radmmm_stream_t stream);
int64_t radmmm_stft_mel_scratch_floats(int B, int S, int n_fft, int hop, int n_mel);

/* ------------------------------------------------------------------------------------
 * Split-f16 support (radmmm_rowgemm_h3's operands): fp32-class GEMM on the f16 matrix
 * cores by operand splitting (x*scale = hi + lo in fp16; A.B ~= Ah.Bh + Ah.Bl + Al.Bh with fp32
 * accumulate).  radmmm_split_f16 writes hi/lo [rows][ldh] halves (ldh % 8 == 0, zero padded).
 * See DESIGN.md §4.2 for the measured rate and error.
 * ------------------------------------------------------------------------------------ */
int radmmm_split_f16(const float* x, int ld, void* hi, void* lo, int ldh, int rows, int cols,
                     float scale, const radmmm_split_opts* so, radmmm_stream_t stream);
/* weight-norm fold (see radmmm_weightnorm_fwd) straight into split packed weights
 * W{h,l}[tap][Cout][ldk] = split(scale * g v/||v||); g == NULL: plain weights (inv_norm unused).
 * so->fmt RADMMM_SPLIT_X8B: Wl is the B-role cross array (ldk % 32 == 0) */
int radmmm_weightnorm_fwd_h3(const float* v, const float* g, void* Wh, void* Wl, float* inv_norm,
                             int Cout, int Cin, int taps, int ldk, int perm_split, int off_lo,
                             int off_hi, float scale, const radmmm_split_opts* so, radmmm_stream_t stream);
/* dst[b][c][r] = src[b][r][c] for both members of a split pair (batches of [rows][cols]); fmt RADMMM_SPLIT_X8B:
 * src_l / dst_l are B-role cross arrays (their 8-bit hi parts are re-derived from the fp16 hi, the 8-bit lo parts
 * move with the transposition; ld_src, ld_dst % 32 == 0, x8_exp as the source was written) */
int radmmm_transpose_f16_pair(const void* src_h, const void* src_l, int ld_src, int64_t src_batch,
                              void* dst_h, void* dst_l, int ld_dst, int64_t dst_batch, int batches,
                              int rows, int cols, int fmt, int x8_exp, radmmm_stream_t stream);
/* The two above for up to 16 tensors in ONE launch each: the weights of a flow step are mostly 4 - 9 MB, where a launch of its
 * own is two memory round trips long rather than bandwidth-bound.  Same arithmetic per tensor, same results. */
typedef struct radmmm_wn_item {
  const float* v;
  const float* g;       /* NULL: plain weights */
  void* Wh;
  void* Wl;
  float* inv_norm;      /* may be NULL when g is NULL */
  int Cout, Cin, taps, ldk, perm_split, off_lo, off_hi;
} radmmm_wn_item;
int radmmm_weightnorm_fwd_h3_multi(const radmmm_wn_item* items, int n, float scale, const radmmm_split_opts* so,
                                   radmmm_stream_t stream);
typedef struct radmmm_tp_item {
  const void* src_h;
  const void* src_l;
  void* dst_h;
  void* dst_l;
  int64_t src_batch, dst_batch;
  int ld_src, ld_dst, batches, rows, cols;
} radmmm_tp_item;
/* (8-bit B-role pairs only: fmt RADMMM_SPLIT_X8B) */
int radmmm_transpose_f16_pair_multi(const radmmm_tp_item* items, int n, int fmt, int x8_exp, radmmm_stream_t stream);
/* Workgroup slots the GEMM grids are sized for: the device's CU count, or RADMMM_GEMM_CUS (32 .. CUs; read once
 * per process) when data-parallel runs leave CUs to RCCL's channel kernels (Lightning `strategy: ddp`,
 * configs/RADMMM_train_config.yaml:28; rad_mmm_amd/ddp.py reserve_collective_cus). */
int radmmm_gemm_cu_slots(void);
/* ------------------------------------------------------------------------------------
 * CTC loss of the alignment attention (loss.py:112-141: torch.nn.CTCLoss, blank 0, zero_infinity, with the targets
 * 1, 2, .., L_b -- every text position once, in order) for a whole batch, value and gradient in one call (two launches, no host read).
 *   lp [B][T][C] log-probabilities (log-softmax outputs, class 0 = blank), lens_txt[b] = L_b <= C - 1, lens_mel[b] = T_b <= T
 *   nll [B]: -log p(targets | frames 0 .. T_b - 1), or 0 where no alignment exists (T_b < L_b)
 *   grad [B][T][C]: d nll[b] / d lp as torch's ctc_loss_backward defines it (exp(lp) - exp(log-sum alpha*beta + nll - lp);
 *   0 for frames >= T_b and for utterances without alignment)
 *   scratch: radmmm_ctc_monotonic_scratch_floats(B, T, C) floats.  At most 511 text positions.
 * ------------------------------------------------------------------------------------ */
int64_t radmmm_ctc_monotonic_scratch_floats(int B, int T, int C);
int radmmm_ctc_monotonic(const float* lp, const int32_t* lens_txt, const int32_t* lens_mel, float* nll, float* grad,
                         float* scratch, int B, int T, int C, radmmm_stream_t stream);

/* A HIP stream whose kernels may use only `enabled_cus` of the device's CUs (hipExtStreamCreateWithCUMask); *out receives
 * the hipStream_t (wrap it, e.g. torch.cuda.ExternalStream).  radmmm_stream_destroy releases it. */
int radmmm_stream_create_masked(int enabled_cus, void** out);
int radmmm_stream_destroy(void* stream);

/* Weight gradient on the split-f16 path.  Operands are transposed, time-contiguous, zero-gapped
 * split copies made by radmmm_transpose_split_act from channels-last fp32 [B*T][ld]:
 *   out[c][front + b*Tp + t] = split(scale * x[b*T+t][c]),  t < (mask_mode ? lens[b] : T), else 0
 * (Tp >= T + max|shift|, front >= max|shift|, ldk % 8 == 0, ldk >= front + B*Tp; the front columns
 * and the row tail must be zero: allocate zeroed once, the gaps are rewritten on every call).
 * o1h/o1l (optional) receive the copy advanced by one column, used for odd tap shifts.
 *   P[split][tap][m][n] = acc_scale * sum_k GYt[m][k] * Xt[n][k + (tap - taps/2)*dil]
 * over the Kt (% 32 == 0) columns k in [k0, k0 + Kt), k0 = front; split-K into `splits` slabs as
 * radmmm_wgrad_f32.  ldk >= k0 + Kt + max|shift|. */
int radmmm_transpose_split_act(const float* x, int ld, int C, int B, int T, int Tp, int front,
                               const int32_t* lens, int mask_mode, float scale, void* oh, void* ol,
                               void* o1h, void* o1l, int ldk, radmmm_stream_t stream);
/* radmmm_transpose_split_act + the weighted column sums of x (bias gradient) from the same pass:
 * part [B * ceil(Tp / 64)][C] partial rows, to be added by radmmm_colsum_final(part, out, nparts, C).
 * sum_weight / sum_taps / sum_dil as radmmm_colsum's row_weight / taps / dil. */
int radmmm_transpose_split_act_colsum(const float* x, int ld, int C, int B, int T, int Tp, int front,
                                      const int32_t* lens, int mask_mode, float scale, void* oh, void* ol, void* o1h,
                                      void* o1l, int ldk, float* part, int sum_weight, int sum_taps, int sum_dil,
                                      radmmm_stream_t stream);
int radmmm_colsum_final(const float* part, float* out, int nparts, int cols, radmmm_stream_t stream);
/* y = g * act'(saved) (radmmm_dact_mul without row weights) written three ways in one pass, with no fp32 copy of y:
 * the row-major split pair yh/yl [rows][ldyh] of scale*y in so's format (feeds the data-gradient GEMM), the transposed
 * zero-gapped split-f16 pair oh/ol [C][ldk] of scale*y (radmmm_transpose_split_act's layout: feeds radmmm_wgrad_h3), and
 * the column sums of y as part [B * ceil(Tp / 64)][C] (radmmm_colsum_final adds them: the bias gradient).
 * C, ldg, lds multiples of 4; yh may be NULL. */
int radmmm_dact_mul_transposed(const float* g, int ldg, const float* saved, int lds, int C, int B, int T, int Tp,
                               int front, int dact, float scale, void* yh, void* yl, int ldyh,
                               const radmmm_split_opts* so, void* oh, void* ol, int ldk, float* part,
                               radmmm_stream_t stream);
/* radmmm_dact_mul for a backward that needs no fp32 copy of y: y = g * act'(saved) * row factor (rowscale as radmmm_dact_mul:
 * 0 none, 1 the length mask, 2 mask x partial-conv ratio, partialconv1d.py:75-81) written as the row-major split pair
 * yh/yl [B*T][ldyh] of scale*y only, and the BIAS-gradient sums in the same pass: part [B * ceil(T / 64)][C] partial rows of
 * sum_t g * act' * [t < len] (= radmmm_colsum(y, row_weight = rowscale): the ratio and its inverse weight cancel), to be
 * added by radmmm_colsum_final.  C, ldg, lds multiples of 4, 16-byte aligned g / saved. */
int radmmm_dact_mul_rows(const float* g, int ldg, const float* saved, int lds, int C, int B, int T, int dact,
                         int rowscale, const int32_t* lens, int taps, int dil, float scale, void* yh, void* yl,
                         int ldyh, const radmmm_split_opts* so, float* part, radmmm_stream_t stream);
/* y_j = g * act'(saved_j), j < n <= 4, in ONE pass over g (the four res/skip layers of a WN all start from the gradient of the
 * skip sum, common.py:816-835): per item the row-major split pair yh / yl (pitch ldyh, format / exponent / saturation flag from
 * `so`; ylo16: the optional fp16 lo part beside an 8-bit cross array) and B * ceil(T / 64) rows of column-sum partials (pitch
 * C) for radmmm_colsum_final(_multi) -- element for element what n calls of radmmm_dact_mul_rows with rowscale 0 write.
 * All saved tensors share the pitch lds; so->lo16 is ignored (per item). */
typedef struct radmmm_dact_item {
  const float* saved;
  void* yh;
  void* yl;
  void* ylo16; /* or NULL */
  float* part;
} radmmm_dact_item;
int radmmm_dact_mul_rows_multi(const float* g, int ldg, const radmmm_dact_item* items, int n, int lds, int C, int B, int T,
                               int dact, float scale, int ldyh, const radmmm_split_opts* so, radmmm_stream_t stream);
/* number of workgroup tiles radmmm_wgrad_h3 launches per split (the caller picks `splits` so that
 * tiles * splits fills whole rounds of the CUs: one workgroup per CU) */
int radmmm_wgrad_h3_tiles(int Mc, int Nc, int taps);
int radmmm_wgrad_h3(const void* GYh, const void* GYl, const void* Xh, const void* Xl, const void* X1h,
                    const void* X1l, int ldk, int k0, int Kt, float* P, int ldp, int64_t split_stride, int Mc,
                    int Nc, int taps, int dil, int splits, float acc_scale, int nprod /* 3 or 1, as radmmm_rowgemm_h3_desc */,
                    radmmm_stream_t stream);

/* The same weight gradient on ROW-MAJOR split operands (csrc/wgrad_rm.hip): GYh/GYl [R][ldg] and Xh/Xl [R][ldx] are the
 * [frames][channels] fp16 hi/lo pairs the GEMM epilogues write (R = B*T rows, utterance-major), contracted over the frames
 * with the transposition done in the LDS read; no transposed copies.
 *   P[split][tap][m][n] = acc_scale * sum_f GY[f][m] * X[f + s][n],  s = (tap - taps/2)*dil, for f + s in f's utterance.
 * x_mask: X rows at frames >= lens[b] count as zeros (partial padding: the conv's input is x * mask).  ldg, ldx %% 8 == 0,
 * 16-byte aligned operands, T >= 32, at most 1024 utterances; ldp >= Nc; slabs `split_stride` floats apart;
 * radmmm_wgrad_rm_tiles as radmmm_wgrad_h3_tiles. */
int radmmm_wgrad_rm_tiles(int Mc, int Nc, int taps);
int radmmm_wgrad_rm(const void* GYh, const void* GYl, int ldg, const void* Xh, const void* Xl, int ldx, int R, int T,
                    const int32_t* lens, int x_mask, float* P, int ldp, int64_t split_stride, int Mc, int Nc, int taps,
                    int dil, int splits, float acc_scale, radmmm_stream_t stream);

/* The same weight gradient under the FP8-cross scheme (csrc/wgrad_rm8.hip; common.py:816-835 backward,
 * partialconv1d.py:82): GYh.Xh on the f16 matrix pipe + both cross terms GYh.Xl + GYl.Xh in one block-scaled FP8 MFMA per
 * 32-frame K step -- two thirds of radmmm_wgrad_rm's MFMA work.  GYx / Xx are the tensors' 8-bit cross arrays
 * (RADMMM_SPLIT_X8A, written with exponents g8_exp / x8_exp); only their lo8 halves are read, the hi8 parts are derived
 * from the fp16 hi planes in registers.  No fp16 lo array is involved.  ldg, ldx multiples of 32. */
int radmmm_wgrad_rm8(const void* GYh, const void* GYx, int ldg, int g8_exp, const void* Xh, const void* Xx, int ldx, int x8_exp,
                     int R, int T, const int32_t* lens, int x_mask, float* P, int ldp, int64_t split_stride, int Mc, int Nc,
                     int taps, int dil, int splits, float acc_scale, radmmm_stream_t stream);

/* Bidirectional single-layer LSTM, recurrent part (reference: the decoder's context LSTM,
 * models/radmmm.py:141-146 = torch.nn.LSTM(bidirectional, batch_first) on a packed batch; gate order
 * i, f, g, o; frames t >= lens[b] produce h = c = 0 as pad_packed_sequence does).
 *   G  [B*T][8H]  in: x W_ih^T + b_ih + b_hh (direction d in columns d*4H ..); out: gate activations
 *   W_hh [2][4H][H];  y, c [B*T][2H];  scratch sizes: radmmm_lstm_scratch_bytes(B, H, which) with
 *   which = 0 wsplit, 1 hsplit (fwd), 2 wtpack, 3 P, 4 dcbuf (bwd).
 * radmmm_lstm_bwd overwrites G with the pre-activation gradients dG; the input / weight / bias
 * gradients are plain GEMMs of dG (dW_ih = dG^T x, dx = dG W_ih, db = colsum dG, dW_hh[d] = dG_d^T h_prev).
 * gscale: device scalar, power of two bringing dG into fp16 range (e.g. 2^floor(log2(64 / max|dy|))).
 * All T steps run in ONE launch when the grid (ceil(H/8) x 2 x ceil(B/32) workgroups) fits one workgroup per CU: the
 * workgroups hand h / the partial recurrent gradients from step to step through memory, the data being its own ready
 * flag (csrc/lstm.hip).  The forward pass then needs hseq, radmmm_lstm_hseq_bytes(B, T, H) bytes of scratch (one operand
 * slot per step); that function returns 0, and hseq may be null, when the dimensions take the launch-per-step path. */
int64_t radmmm_lstm_scratch_bytes(int B, int H, int which);
int64_t radmmm_lstm_hseq_bytes(int B, int T, int H);
int radmmm_lstm_fwd(float* G, const float* W_hh, float* y, float* c, const int32_t* lens, void* wsplit, void* hsplit,
                    void* hseq, int B, int T, int H, radmmm_stream_t stream);
int radmmm_lstm_bwd(float* G, const float* c, const float* dy, const float* W_hh, const int32_t* lens, void* wtpack,
                    float* P, float* dcbuf, int B, int T, int H, const float* gscale, radmmm_stream_t stream);

/* Channel-mix matrix of the LUS invertible 1x1 conv (common.py:507-548, Invertible1x1ConvLUS.forward's
 * W = P (L U) with L = tril(lower,-1) + diag(lower_diag), U = triu(upper,1) + diag(upper_diag), and
 * log|det W| = sum log|upper_diag|) in one launch each way instead of ~16 small stock launches.
 * P, lower, upper: row-major fp32 [c][c]; c <= 256.  fwd writes the whole zero-padded [ldw][ldw] matrix W
 * with the c x c block at rows [0,c), columns [col_offset, col_offset+c) (what the flow step's first GEMM
 * reads; an early exit is a column offset) and the scalar logdet (may be NULL).  bwd reads the same
 * block of gW [ldw][ldw] and the device scalar g_logdet (may be NULL) and writes the full [c][c]
 * g_lower (strictly lower, zeros elsewhere), g_upper (strictly upper) and g_upper_diag [c]. */
int radmmm_lu_weight_fwd(const float* P, const float* lower, const float* lower_diag, const float* upper,
                         const float* upper_diag, int c, float* W, int ldw, int col_offset, float* logdet,
                         radmmm_stream_t stream);
int radmmm_lu_weight_bwd(const float* P, const float* lower, const float* lower_diag, const float* upper,
                         const float* upper_diag, int c, const float* gW, int ldw, int col_offset,
                         const float* g_logdet, float* g_lower, float* g_upper, float* g_upper_diag,
                         radmmm_stream_t stream);

/* Masked InstanceNorm1d (+ ReLU) on channels-last rows (next-row f1, the text encoder's
 * nn.InstanceNorm1d(affine=True) applied per utterance, common.py:439-441,476-484): statistics over the
 * frames t < lens[b] of item b (biased variance, eps as torch), zeros at frames >= lens[b].
 * x, y, gy, gx [B*T][ld]; mean, rstd [B][C]; dw_part, db_part [B][C] (sum over B = parameter gradients). */
int radmmm_instnorm_fwd(const float* x, int ldx, const float* weight, const float* bias, float* y, int ldy, float* mean,
                        float* rstd, const int32_t* lens, int B, int T, int C, float eps, int relu, radmmm_stream_t stream);
int radmmm_instnorm_bwd(const float* gy, int ldg, const float* x, int ldx, const float* y, int ldy, const float* weight,
                        const float* mean, const float* rstd, float* gx, int ldgx, float* dw_part, float* db_part,
                        const int32_t* lens, int B, int T, int C, int relu, radmmm_stream_t stream);

/* On-device data path (next-row f4): what the reference's CPU dataset workers compute per utterance.
 * radmmm_betabinom_prior: data.py:90-102 (beta_binomial_prior_distribution) -> out [M][P] float64, row i the pmf
 *   of BetaBinomial(P-1, scaling*(i+1), scaling*(M-i)).
 * radmmm_prior_zoom_batch: BetaBinomialInterpolator.__call__ (data.py:76-88: scipy.ndimage.zoom order=1 of the
 *   anchor prior at the rounded sizes, rows renormalised) for a whole batch, zero-padded as DataCollate does
 *   (data.py:678-679,737-741).  items: DEVICE array [B][5] of int64 {anchor pointer (float64 [bh][bw]), bh, bw,
 *   n_frames, n_tokens}; out [B][Tmax][Nmax] fp32, fully written.
 * radmmm_energy_average: data.py:363-366 (+ :339-342 when scaled): mel [B][n_mel][T] -> out [B][T]. */
int radmmm_betabinom_prior(int P, int M, double scaling, double* out, radmmm_stream_t stream);
int radmmm_prior_zoom_batch(const int64_t* items, int B, float* out, int Tmax, int Nmax, radmmm_stream_t stream);
int radmmm_energy_average(const float* mel, float* out, int B, int n_mel, int T, int scaled, radmmm_stream_t stream);

/* Optimizer step on flat fp32 buffers (next-row f3): the reference's vendored RAdam (radam.py:63-142)
 * with the global-norm clip of configs/RADMMM_train_config.yaml:7-8 folded in as a device scalar.
 * step_size / use_denom (N_sma >= 5) come from the host's step count as radam.py:101-123. */
int64_t radmmm_sumsq_scratch_floats(void);
int radmmm_sumsq(const float* x, int64_t n, float* partial, radmmm_stream_t stream);
int radmmm_radam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* clip_coef, float beta1,
                      float beta2, float eps, float step_size, float wd_lr, int use_denom, radmmm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADMMM_HIP_H */
