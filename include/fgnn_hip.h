/*
 * fgnn_hip.h — C ABI of libfgnn_hip.so, the MI355X (gfx950) implementation of the
 * FGNN Variable->Factor / Factor->Variable message operator.
 *
 * The reference (zzhang1987/Factor-Graph-Neural-Network) has no FFI for this path: the
 * operator is a Python nn.Module, `mp_conv_v2.forward(x, nn_idx, etype)`
 * (lib/model/mpnn/mp_nn.py:115-175), reached through `mp_conv_residual.forward`
 * (lib/model/mpnn/mp_nn_residual.py:39-56).  This header is the boundary a maintainer of
 * the reference would bind (ctypes stub in INTEGRATION.md): plain pointers, sizes and a
 * hipStream_t — no torch types.  Every entry point returns 0 on success and a negative
 * FGNN_E* code otherwise; fgnn_last_error() gives the message (thread-local).
 *
 * Tensor conventions (all strides are in ELEMENTS, so both NCHW and channels-last views of
 * the reference's [B,C,N,1] tensors can be passed without a copy):
 *   x      [B, nin, N]      float32 or bf16     element (b,c,n) at x + b*x_sb + c*x_sc + n*x_sn
 *   nn_idx [B, M, k]        int64 (as the reference passes it), values in [0,N);
 *                           idx_sb == 0 means "one graph shared by the whole batch"
 *   etype  [B, net, M, k]   same dtype as x; et_sb == 0 allowed (shared edge weights)
 *   filters[R, nou*net]     float32, R = nin (NO_EXTENSION) or 2*nin; column = o*net + e
 *                           (the reference's `filters` parameter, mp_nn.py:41-49)
 *   y      [B, nou, M]      same dtype as x
 */
#ifndef FGNN_HIP_H
#define FGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fgnn_stream_t; /* a hipStream_t; NULL = the null stream */

enum { FGNN_OK = 0, FGNN_EINVAL = -1, FGNN_ELAUNCH = -2, FGNN_EUNSUPPORTED = -3 };

/* mp_conv_type (mp_nn.py:7-10) */
enum { FGNN_EXT_NONE = 0, FGNN_EXT_NEIGHBOR = 1, FGNN_EXT_DIFF = 2 };
/* aggregator strings 'max' / 'softmax' (= (1/3) logsumexp(3 x)) / 'mean' (mp_nn.py:68-90) */
enum { FGNN_AGG_MAX = 0, FGNN_AGG_LSE = 1, FGNN_AGG_MEAN = 2 };
enum { FGNN_F32 = 0, FGNN_BF16 = 1 };

typedef struct fgnn_mpconv_desc {
    int32_t B, nin, nou, net, N, M, k;
    int32_t ext;    /* FGNN_EXT_*  */
    int32_t agg;    /* FGNN_AGG_*  */
    int32_t dtype;  /* FGNN_F32 / FGNN_BF16: storage type of x, etype, y (accumulation is f32) */
    int32_t relu;   /* apply max(.,0) last (forward only) */
    int32_t reserved; /* forward: FGNN_DESC_IDENTITY_LIST (below); backward: bits 0-15 = largest in-degree of the shared neighbour table (0 = unknown),
                         FGNN_DESC_GETYPE_REDUCED = getype is the batch-summed [net, M, k] gradient (see below) */
    int64_t x_sb, x_sc, x_sn;
    int64_t idx_sb, idx_sm, idx_sk;
    int64_t et_sb, et_se, et_sm, et_sk;
    int64_t y_sb, y_sc, y_sm;
} fgnn_mpconv_desc;

/*
 * Batch-statistics BatchNorm, forward finalisation (torch.nn.BatchNorm2d semantics: the reference's conv1 / conv2 BatchNorms,
 * mp_nn_residual.py:25-35, mp_conv_v2.bn, mp_nn.py:57-58,170-173, iid_mapping_bn, base_model.py:62-79): what to compute from the
 * per-channel sums a kernel has formed over `count` rows.  All vectors float32 [C].  Entry points that take a `fgnn_bn_final` leave
 * the BatchNorm FINALISED: by a small launch of their own behind the producing kernel (default), or — fgnn_set_inkernel_finalisers(1),
 * with a `fold_scratch` — by the producer's last workgroup, which folds the per-workgroup partial rows in a fixed order
 * (csrc/fgnn_gridfold.h).  Either way the caller launches nothing else.
 *   fold_scratch: FGNN_FOLD_SCRATCH_BYTES of device memory, ZERO when first handed over and only ever passed to kernels of one
 *   stream at a time (ticket counters, reset by their last user, and second-level rows).
 */
typedef struct fgnn_bn_final {
    const float* gamma;            /* or NULL = 1 */
    const float* beta;             /* or NULL = 0 */
    float* running_mean;           /* updated in place with `momentum`, or NULL */
    float* running_var;            /* ... with the UNBIASED variance over `population` rows */
    int64_t* num_batches_tracked;  /* += 1, or NULL */
    float* mean; float* invstd; float* scale; float* shift;   /* outputs: scale = gamma invstd, shift = beta - mean scale */
    const float* shift_k;          /* per-channel constant the producer left out of its sums (a bias added later), or NULL */
    int64_t count;                 /* rows the sums run over */
    int64_t population;            /* rows the statistics stand for in the unbiased running variance (0 = count): a tensor that is ONE row
                                      broadcast over m nodes has count rows but count * m of them in the reference's BatchNorm */
    float momentum, eps;
} fgnn_bn_final;
#define FGNN_FOLD_SCRATCH_BYTES (512 + 64 * 512 * 8)
/* Who finalises: 0 (default) = a small finaliser launch behind the producer (rounds 1-4; measured faster on MI355X: the in-kernel
 * fold is six serialised memory-side round trips at the producer's tail), 1 = the producer's last workgroup.  Returns the previous
 * setting.  Also FGNN_INKERNEL_FINALISERS=1 in the environment. */
int fgnn_set_inkernel_finalisers(int32_t on);

/*
 * Forward: y = act( post_scale * (agg_j sum_e etype[e,m,j] * msg[m,j,:,e] + bias) + post_shift )
 * Replaces mp_conv_v2.forward steps a-k: gather -> matmul(filters) -> bmm(etype)
 * -> aggregate -> +bias -> (eval-mode BatchNorm folded into post_scale/post_shift) -> ReLU.
 *   bias, post_scale, post_shift : float32 [nou] or NULL.
 *   argmax : uint8 [B, nou, M] with the SAME element strides as y, or NULL; for FGNN_AGG_MAX it
 *            receives the winning neighbour slot (first occurrence on ties, as torch.max on CPU).
 */
int fgnn_mpconv_forward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                        const void* etype, const float* filters, const float* bias,
                        const float* post_scale, const float* post_shift, void* y,
                        uint8_t* argmax, fgnn_stream_t stream);

/*
 * fgnn_mpconv_forward (inference form: post_scale / post_shift given, d->relu) with up to three ADDENDS of y's layout — the
 * layer's running sum, residual and skip terms, which /root/reference/lib/model/mpnn/factor_mpnn_sp.py:139-168 adds to the
 * operator's activated output — folded into the kernel's epilogue where the kernel family has one for them (the bf16
 * parity-check kernels of csrc/mpconv_fwd_ws.hip).  Returns 1: y = act(...) + addend0 (+ addend1 + addend2); 0: y was
 * computed WITHOUT the addends (the caller adds them); < 0: error.  addend0 first; NULL = absent.
 */
int fgnn_mpconv_forward_addends(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                                const float* filters, const float* bias, const float* post_scale, const float* post_shift,
                                const void* addend0, const void* addend1, const void* addend2, void* y, fgnn_stream_t stream);

/*
 * Backward of z = agg(...) + bias w.r.t. x, etype, filters, bias (mp_conv_v2 is trained through
 * autograd in the reference; this is the hand-written counterpart).
 *   gz      [B, nou, M]  upstream gradient w.r.t. the pre-BN output z (strides y_s* of d)
 *   z       reserved, pass NULL (the softmax weights are recomputed from x)
 *   argmax  as written by the forward (FGNN_AGG_MAX; NULL otherwise)
 *   gx      [B, nin, N]  same dtype and ELEMENT strides as x; fully written
 *   getype  [B, net, M, k] contiguous, same dtype as etype, fully written; NULL = not wanted (the
 *            caller's etype is a constant: skips the edge-weight gradient and unlocks the
 *            hyper-edge kernels of mpconv_bwd_hyper.hip)
 *   gfilters[R, nou*net] float32, ACCUMULATED into (caller zero-fills)
 *   gbias   [nou] float32 or NULL, ACCUMULATED into
 *   workspace / workspace_bytes : REQUIRED device scratch of fgnn_mpconv_backward_workspace_bytes(d) bytes (ABI >= 6: every backward —
 *                                 the shape-generic one included — accumulates dW / dbias in per-workgroup slabs there and folds them
 *                                 in a fixed order; no float atomics).  NULL or too small: FGNN_EINVAL.  Device memory, contents undefined on
 *                                 entry and exit.
 */
int fgnn_mpconv_backward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx,
                         const void* etype, const float* filters, const void* gz,
                         const void* z, const uint8_t* argmax, void* gx, void* getype,
                         float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                         fgnn_stream_t stream);

int64_t fgnn_mpconv_backward_workspace_bytes(const fgnn_mpconv_desc* d);
/*
 * The graph of a training run is static (the reference hands the same nn_idx every step, train_ldpc.py:209-216): the table-driven
 * backward of the bf16 parity shapes (64 -> 64 / 64 -> 128 channels, degree 3 / 6, one table shared by the batch) can take the
 * transposed incidence it works from — which source node feeds which (destination, slot) — PRE-BUILT:
 *   fgnn_mpconv_backward_tables_bytes(d)  bytes of the tables for this descriptor (0 = this shape does not use any);
 *   fgnn_mpconv_backward_tables(...)      builds them (one small launch; d->reserved carries the in-degree bound as for the backward);
 *   fgnn_mpconv_backward_with_tables(...) = fgnn_mpconv_backward with `tables` (NULL = none, identical results either way): without
 *                                         them every workgroup of every launch builds its own (17 k cycles, which — measured — hide
 *                                         under the first samples' loads: the tables buy nothing on MI355X and are off by default).
 * The tables depend on nn_idx's contents, N, M and k only.
 */
int64_t fgnn_mpconv_backward_tables_bytes(const fgnn_mpconv_desc* d);
int fgnn_mpconv_backward_tables(const fgnn_mpconv_desc* d, const int64_t* nn_idx, void* tables, fgnn_stream_t stream);
int fgnn_mpconv_backward_with_tables(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                                     const float* filters, const void* gz, const void* z, const uint8_t* argmax, void* gx,
                                     void* getype, float* gfilters, float* gbias, void* workspace, int64_t workspace_bytes,
                                     const void* tables, fgnn_stream_t stream);

/*
 * Edge weights shared by the batch (et_sb == 0; every reference script builds them from one [1, ., M, k] feature table,
 * train_syn_hop_factor.py:284-295): what autograd needs is the gradient SUMMED over the batch.  Where
 * fgnn_mpconv_backward_reduces_getype(d) returns 1, setting FGNN_DESC_GETYPE_REDUCED in d->reserved makes
 * fgnn_mpconv_backward write getype as [net, M, k] (float32, contiguous, fully written, summed over the batch in a fixed
 * order) instead of [B, net, M, k]; with the flag set on any other descriptor the call fails with FGNN_EUNSUPPORTED.
 */
#define FGNN_DESC_GETYPE_REDUCED 0x10000
/* forward only, in fgnn_mpconv_desc.reserved: the caller vouches that the (batch-shared) neighbour table of a one-destination call
 * (M == 1, k == N) is the identity list idx[j] == j — the LDPC hyper-factor's (train_ldpc.py:40-46).  The fan-in kernel then never reads
 * the table and reduces over the nodes in the matrix-core accumulators (the projected values stay f32: the general kernel rounds them
 * to bf16 before the edge weight — results agree to bf16 rounding).  Unset = the general kernel. */
#define FGNN_DESC_IDENTITY_LIST 0x20000
int fgnn_mpconv_backward_reduces_getype(const fgnn_mpconv_desc* d);

/* Bytes of dynamic LDS the forward will request for this descriptor (diagnostics / tests). */
/*
 * Training form of fgnn_mpconv_forward (no post-affine, no ReLU) whose epilogue also leaves the batch statistics of
 * the stored output for the BatchNorm that follows the operator (mp_nn.py:170): per-workgroup partials
 * [rows][2][nou] (sum, sum of squares) in fgnn_bn_finalize's layout, so that BatchNorm needs no pass of its own over z.
 * fgnn_mpconv_forward_stats_partials(d) = the number of partial rows the launch writes (<= 1024), or 0 when this shape
 * has no statistics epilogue (bf16 one-pass parity shapes with <= 64 output channels have one).
 */
int fgnn_mpconv_forward_stats_partials(const fgnn_mpconv_desc* d);
int fgnn_mpconv_forward_stats(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                              const float* filters, const float* bias, void* y, uint8_t* argmax, float* stats_partials,
                              const fgnn_bn_final* fin, void* fold_scratch, fgnn_stream_t stream);
/* fin (with fold_scratch; or NULL): the BatchNorm behind the operator is finalised by this launch (fgnn_bn_final: count = B * M). */

int64_t fgnn_mpconv_forward_lds_bytes(const fgnn_mpconv_desc* d);

/* Algorithmic HBM bytes of one forward call (SURVEY §8d formula; used by bench.py's roofline). */
int64_t fgnn_mpconv_algorithmic_bytes(const fgnn_mpconv_desc* d);

/*
 * Forward of the node-wise (1x1) maps, `Conv2d(cin, cout, 1)` of mp_conv_residual / iid_mapping*
 * (reference mp_nn_residual.py:25-35, base_model.py:43-90), as a streaming bf16 GEMM: y[R][Cout] = x[R][Cin] W^T + b
 * with bf16 x / y, f32 W [Cout][Cin] / b, Cin and Cout multiples of 64 up to 256 (FGNN_EUNSUPPORTED otherwise).
 * stats_partials: NULL, or fgnn_linear_forward_partials(R, Cin, Cout) * 2 * Cout floats of device scratch that
 * receive per-workgroup (sum y, sum y^2) for fgnn_bn_finalize.  w_transposed != 0: W is [Cin][Cout] in memory
 * (y = x W) — the grad-input product gy W of a map whose own weight is stored [cout'][cin'].
 */
int fgnn_linear_forward(const void* x, const float* W, const float* bias, void* y, int64_t R, int32_t Cin,
                        int32_t Cout, float* stats_partials, const fgnn_bn_final* fin, void* fold_scratch, int32_t w_transposed,
                        fgnn_stream_t stream);
/* fin (with stats_partials and fold_scratch; or NULL): the BatchNorm behind the map is finalised by this launch (see fgnn_bn_final). */

/*
 * The node-wise map FOLLOWED BY InstanceNorm (+ ReLU) in one pass — `iid_mapping_in`, /root/reference/lib/model/mpnn/base_model.py:82-90
 * (Conv2d(cin, cout, 1) -> InstanceNorm2d(cout) -> ReLU: FactorNN's v2v / f2f maps, factor_mpnn_sp.py:77,140) — for bf16
 * channel-fastest rows: y[b, n, :] = act((z[b, n, :] - mean_b) * rstd_b), z = x W^T + bias, statistics per (sample, channel) over the
 * sample's N nodes (biased variance, eps as given), formed from the bf16-rounded z.  z (or NULL): [B * N][Cout] receives the
 * pre-norm output for the backward (fgnn_instnorm_backward reads it); it is never read here.  N in {48, 96}, Cin / Cout
 * multiples of 64 up to 256; FGNN_EUNSUPPORTED otherwise (run fgnn_linear_forward + fgnn_instnorm_forward).
 */
int fgnn_linear_instnorm_forward(const void* x, const float* W, const float* bias, void* z, void* y, int32_t B, int32_t N,
                                 int32_t Cin, int32_t Cout, int32_t relu, float eps, fgnn_stream_t stream);

/*
 * Several node-wise maps into ONE tensor: y[r][:] = sum_s x_s[r][:] W_s + addend_0[r][:] + addend_1[r][:] + addend_2[r][:] — the
 * gradient of a FactorNN state that feeds several consumers (factor_mpnn_sp.py:136-168: the v2v / f2f map and one block per factor
 * type, each contributing `gz_s W_s`, plus the residual / skip gradients that arrive unchanged) as one K-concatenated product
 * instead of one [R][Cout] tensor per consumer and an n-input sum.  x_s [R][K[s]] bf16 dense, W_s [K[s]][Cout] f32 row-major (a
 * map's [cout'][cin'] weight as it lies in memory), addends / y [R][Cout] bf16; three slots each (K[s] == 0 / NULL = absent).
 * K[0], K[2] in {0, 64}, K[1] in {0, 64, 128, 256} (K[0] == 0 only with K[2] == 0), Cout in {64, 128, 256}: fgnn_linear_multi_supported;
 * FGNN_EUNSUPPORTED otherwise.  f32 accumulation, one rounding of the sum.
 */
int fgnn_linear_multi_supported(int64_t R, const int32_t* K, int32_t Cout);
int fgnn_linear_multi_forward(const void* const* x, const int32_t* K, const float* const* W, const void* const* addend, void* y,
                              int64_t R, int32_t Cout, fgnn_stream_t stream);
int fgnn_linear_forward_partials(int64_t R, int32_t Cin, int32_t Cout);

/*
 * Parameter-gradient folds as ONE launch per backward pass.  Every gradient entry point that sums over the batch rows (fgnn_mpconv_backward*,
 * fgnn_linear_wgrad) writes per-workgroup partial slabs into the caller's workspace and folds them into gfilters / gbias / gW / gb with
 * a small launch of its own.  After fgnn_fold_defer(1) those entry points RECORD their fold instead (process-wide list; the caller
 * must then give each call a workspace of its own and keep it alive and untouched), and fgnn_fold_flush(stream) — `stream` ordered
 * behind every producer — folds everything recorded with one launch per 48 jobs (jobs that accumulate into overlapping targets go to
 * separate launches, in recording order).  A fixed per-element summation order (bit-reproducible), not the immediate folds' one: the two
 * agree to f32 rounding.  The
 * accumulators are complete only after the flush.  fgnn_fold_defer returns the previous setting; fgnn_fold_pending the number of
 * recorded jobs; fgnn_fold_discard forgets them (a pass that died).
 */
int fgnn_fold_defer(int32_t on);
int fgnn_fold_pending(void);
void fgnn_fold_discard(void);
int fgnn_fold_flush(fgnn_stream_t stream);

/*
 * Weight / bias gradient of a node-wise linear map y[r,:] = W x[r,:] + b over R = B*N rows
 * (the 1x1 convolutions around the operator: mp_nn_residual.py:25-35, base_model.py:43-90):
 *   gW[o][c] += sum_r gy[r][o] x[r][c],  gb[o] += sum_r gy[r][o]      (f32, ACCUMULATED into)
 * x [R][Cin], gy [R][Cout]: dense row-major, dtype FGNN_F32 / FGNN_BF16.  gb may be NULL.
 * workspace: fgnn_linear_wgrad_workspace_bytes(R, Cin, Cout) bytes of device scratch.
 */
int fgnn_linear_wgrad(const void* x, const void* gy, int64_t R, int32_t Cin, int32_t Cout, int32_t dtype,
                      float* gW, float* gb, void* workspace, int64_t workspace_bytes, fgnn_stream_t stream);
int64_t fgnn_linear_wgrad_workspace_bytes(int64_t R, int32_t Cin, int32_t Cout);

/*
 * The same gradients for nsrc <= 3 maps that read the SAME rows x [R][Cin] — the consumers of one layer state in FactorNN's layer body
 * (factor_mpnn_sp.py:136-168: the state's own v2v / f2f map and conv1 of each block that starts from it, mp_nn_residual.py:25-29) —
 * in ONE pass: gW[s] [couts[s]][Cin] += gy[s]^T x, gb[s] [couts[s]] += column sums of gy[s] (gb, or any gb[s], may be NULL).  x is read
 * once instead of nsrc times.  bf16 only; Cin and every couts[s] multiples of 64 up to 256 with sum_s (couts[s] / 64) (Cin / 64) <= 16
 * (fgnn_linear_wgrad_multi_workspace_bytes returns -1 outside that family; the call itself FGNN_EUNSUPPORTED).  Sums and rounding per
 * map are those of fgnn_linear_wgrad on the same grid.  ABI >= 10.
 */
int fgnn_linear_wgrad_multi(const void* x, int64_t R, int32_t Cin, int32_t nsrc, const void* const* gy, const int32_t* couts,
                            float* const* gW, float* const* gb, void* workspace, int64_t workspace_bytes, fgnn_stream_t stream);
int64_t fgnn_linear_wgrad_multi_workspace_bytes(int64_t R, int32_t Cin, int32_t nsrc, const int32_t* couts);

/*
 * InstanceNorm2d(affine=False, eps 1e-5, biased variance) over the node axis, optionally fused with ReLU,
 * on dense channel-fastest activations x[B][N][C] (the norm of iid_mapping_in, base_model.py:82-90).
 * backward: gx = d/dx of act(norm(x)) given gy; only x is needed (statistics are recomputed).
 */
int fgnn_instnorm_forward(const void* x, void* y, int32_t B, int32_t N, int32_t C, int32_t dtype, int32_t relu,
                          fgnn_stream_t stream);
int fgnn_instnorm_backward(const void* x, const void* gy, void* gx, int32_t B, int32_t N, int32_t C,
                           int32_t dtype, int32_t relu, fgnn_stream_t stream);

/*
 * The classifier's closing pair as one pass: InstanceNorm2d + ReLU + the one-column map behind it
 * (/root/reference/lib/model/mpnn/factor_mpnn_sp.py:104-108: ... InstanceNorm2d -> ReLU -> Dropout/Identity -> Conv2d(128, 1, 1),
 * the decoder's logit per variable) on dense channel-fastest x[B][N][128]:
 *     out[b][n] = bias + sum_c w[c] * relu(instnorm(x)[b][n][c])          (out: dtype of x; w [128], bias [1] or NULL: f32)
 * backward, given gout[B][N] (dtype of x): gx[B][N][128]; gw[128] += d/dw, gbias[0] += sum gout (f32, ACCUMULATED into; gbias may
 * be NULL).  Only x is needed (statistics and the normalised tensor are recomputed; neither it nor its gradient ever exists in
 * memory).  workspace: fgnn_instnorm_dot_workspace_bytes(B) bytes.  C == 128 and 2 <= N <= 128, x / gx 16-byte aligned; anything
 * else: FGNN_EUNSUPPORTED (run fgnn_instnorm_forward + a linear map).
 */
int fgnn_instnorm_dot_forward(const void* x, const float* w, const float* bias, void* out, int32_t B, int32_t N, int32_t C,
                              int32_t dtype, fgnn_stream_t stream);
int fgnn_instnorm_dot_backward(const void* x, const float* w, const void* gout, void* gx, float* gw, float* gbias,
                               int32_t B, int32_t N, int32_t C, int32_t dtype, void* workspace, int64_t workspace_bytes,
                               fgnn_stream_t stream);
int64_t fgnn_instnorm_dot_workspace_bytes(int32_t B);

/*
 * TRAINING-mode fusion of the tail of `mp_conv_residual` (reference mp_nn.py:165-175 = the operator's
 * BatchNorm + ReLU, mp_nn_residual.py:31-35,49-51 = conv2 + BatchNorm + LeakyReLU): behind the message operator's
 * 64-channel output e [R][64] (bf16, R = B * M destination rows)
 *     a2 = act2(e * scale2 + shift2);   z3 = a2 W2^T + b2  (W2 [Cout][64] f32, Cout in {64, 128, 256});
 *     out = act3(z3 * scale3 + shift3) + addend0 + addend1 + addend2
 * with batch-statistics BatchNorms, WITHOUT storing the Cout-wide z3: every pass recomputes it from e on the matrix cores.
 *   fgnn_block_tail_stats   : BatchNorm3's batch statistics -> *fin (count = R; the kernel sums z3 - b2 and finalises with
 *                             K = b2 itself: pass fin->shift_k = NULL); partials: fgnn_block_tail_partials(R, Cout) rows of
 *                             [2][Cout] floats of scratch.
 *   fgnn_block_tail_apply   : out [R][Cout] bf16 (addends: bf16 [R][Cout], or [R / period][Cout] with addend_period, or NULL);
 *                             a2_out (or NULL) receives a2 [R][64] bf16 for the weight-gradient kernel of the backward.
 *   fgnn_block_tail_backward: from gout [R][Cout]: BatchNorm3's parameter gradients (ACCUMULATED into gweight3 / gbias3, may
 *                             be NULL), gz3 [R][Cout] = the gradient of z3 (bf16; what fgnn_linear_wgrad contracts with a2)
 *                             and ga2 [R][64] = gz3 W2 (the gradient of a2, before act2' / BatchNorm2).
 *                             workspace: >= (2048 * Cout + 2 * Cout + 1024 * 128) * 4 bytes.
 * slope: LeakyReLU slope of the activation (0 = ReLU, 1 = none).  FGNN_EUNSUPPORTED outside this family.
 */
int fgnn_block_tail_partials(int64_t R, int32_t Cout);
int fgnn_block_tail_stats(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                          const float* b2, int64_t R, int32_t Cout, float* partials, const fgnn_bn_final* fin,
                          void* fold_scratch, fgnn_stream_t stream);
/* addend_period (NULL = {1,1,1}): see fgnn_bn_apply. */
int fgnn_block_tail_apply(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                          const float* b2, const float* scale3, const float* shift3, float slope3, const void* addend0,
                          const void* addend1, const void* addend2, const int32_t* addend_period, void* out, void* a2_out,
                          int64_t R, int32_t Cout, fgnn_stream_t stream);
/* bn2_dsum (or NULL, with mean2 / invstd2; gweight2 / gbias2 ACCUMULATED into, may be NULL): [2][64] floats receiving BatchNorm2's
 * backward sums (dbeta, dgamma) for fgnn_bn_backward_apply — the 64-channel BatchNorm then needs no reduction pass of its own.
 * Two launches (reduce, grad), each finalising its sums in its last workgroup. */
int fgnn_block_tail_backward(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                             const float* b2, const float* mean3, const float* invstd3, const float* gamma3,
                             const float* scale3, const float* shift3, float slope3, const void* gout, void* gz3, void* ga2,
                             float* gweight3, float* gbias3, const float* mean2, const float* invstd2, float* gweight2,
                             float* gbias2, float* bn2_dsum, int64_t R, int32_t Cout, void* workspace, int64_t workspace_bytes,
                             void* fold_scratch, fgnn_stream_t stream);
int fgnn_block_tail_backward_partials(int64_t R, int32_t Cout);   /* workgroups (= BatchNorm2 partial rows) of the backward's grad launch */
/* conv2's WEIGHT gradient without a stored gz3 / a2 (ABI 12).  gz3 = scale3 g' + A (z3 - b2) + (A b2 + Bc) is affine in what the reduce
 * pass of the backward already holds, so  gW2 = diag(scale3) M1 + diag(A) (bf16(W2) Gram) + (A b2 + Bc) v^T  with the moments
 * M1 = sum_rows g' a2^T [Cout][64], Gram = sum_rows a2 a2^T [64][64], v = sum_rows a2 [64].  fgnn_block_tail_backward_moments is
 * fgnn_block_tail_backward whose reduce launch also accumulates the moments into `moments` (fgnn_block_tail_moments_bytes(R, Cout)
 * bytes, 16-byte aligned, owned by the caller until the finish call has run; it also keeps A and Bc); gz3 may then be NULL (not stored).
 * fgnn_block_tail_wgrad_finish folds the moments and ADDS conv2's weight gradient to gW2 [Cout][64] f32 (two short launches, to be
 * issued any time before the optimizer).  conv2's bias gradient in front of a batch-statistics BatchNorm is identically zero.
 * Replaces fgnn_linear_wgrad(a2, gz3) behind /root/reference/lib/model/mpnn/mp_nn_residual.py:31-35 (autograd of self.conv2). */
int64_t fgnn_block_tail_moments_bytes(int64_t R, int32_t Cout);
int fgnn_block_tail_backward_moments(const void* e, const float* scale2, const float* shift2, float slope2, const float* W2,
                                     const float* b2, const float* mean3, const float* invstd3, const float* gamma3,
                                     const float* scale3, const float* shift3, float slope3, const void* gout, void* gz3, void* ga2,
                                     float* gweight3, float* gbias3, const float* mean2, const float* invstd2, float* gweight2,
                                     float* gbias2, float* bn2_dsum, int64_t R, int32_t Cout, void* workspace, int64_t workspace_bytes,
                                     void* fold_scratch, void* moments, int64_t moments_bytes, fgnn_stream_t stream);
int fgnn_block_tail_wgrad_finish(void* moments, int64_t moments_bytes, int64_t R, int32_t Cout, const float* W2, const float* b2,
                                 const float* scale3, float* gW2, fgnn_stream_t stream);

/*
 * HEAD of a training-mode `mp_conv_residual`, backward: autograd through `self.conv1` = Conv2d(nin, nmed, 1) -> BatchNorm2d ->
 * LeakyReLU (/root/reference/lib/model/mpnn/mp_nn_residual.py:25-29,42-44) for nmed = 64, bf16 channel-fastest rows.
 *   z1 [R][64]   conv1's output (BatchNorm1's input), ga1 [R][64] the gradient of the activated output (what the operator's
 *   backward returns for its x), mean / invstd / gamma / beta [64] BatchNorm1's batch statistics and parameters, W1 [64][Cin] f32.
 *   Writes gz1 [R][64] (BatchNorm1's input gradient: conv1's weight-gradient kernel reads it) and gx [R][Cin] = gz1 W1 (conv1's
 *   input gradient) in ONE element pass — gz1 is not read back — after BatchNorm1's reduction pass; gweight / gbias [64]
 *   (BatchNorm1's parameter gradients) are ACCUMULATED into.  Cin in {64, 128, 256}; workspace: fgnn_bn_workspace_bytes(R, 64).
 *   gx may be NULL: only gz1 is formed (the caller multiplies it by W1 together with the state's other gradients:
 *   fgnn_linear_multi_forward).
 */
int fgnn_block_head_backward(const void* z1, const void* ga1, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float slope, const float* W1, void* gz1, void* gx, float* gweight,
                             float* gbias, int64_t R, int32_t Cin, void* workspace, int64_t workspace_bytes,
                             void* fold_scratch, fgnn_stream_t stream);

/*
 * Train-mode BatchNorm fused with the LeakyReLU(slope) behind it (slope 0 = ReLU, 1 = none) on dense
 * channel-fastest x[R][C] (conv1/conv2 blocks mp_nn_residual.py:25-35, mp_conv_v2.bn mp_nn.py:57-58,170-173,
 * iid_mapping_bn base_model.py:62-79).  torch.nn.BatchNorm2d semantics: biased variance normalises, running
 * statistics take the unbiased one with `momentum`.  All per-channel vectors are float32 [C].
 */
int fgnn_bn_supported(int64_t R, int32_t C, int32_t dtype);
int64_t fgnn_bn_workspace_bytes(int64_t R, int32_t C);
int fgnn_bn_stats(const void* x, int64_t R, int32_t C, int32_t dtype, const fgnn_bn_final* fin, void* workspace,
                  int64_t workspace_bytes, void* fold_scratch, fgnn_stream_t stream);
/* The same outputs from per-workgroup partials [npartials][2][C] of (sum (y - K), sum (y - K)^2), K = fin->shift_k or 0, formed
 * OUTSIDE this library (its own producers — fgnn_linear_forward, fgnn_mpconv_forward_stats, fgnn_block_tail_stats — take the
 * fgnn_bn_final themselves). */
int fgnn_bn_finalize(const float* partials, int32_t npartials, int32_t C, const fgnn_bn_final* fin, fgnn_stream_t stream);
/* y = act(x*scale + shift) + addend + addend2 + addend3: up to three tensors of y's layout (NULL = absent) ride in
 * the apply pass — the `acc + block(x) + residual + skip` sums of factor_mpnn_sp.py:139-170.  addend_period (NULL = {1,1,1}):
 * addend a has ONE row per addend_period[a] consecutive rows of y — a per-sample vector broadcast over the sample's nodes (the
 * hyper-factor's message to the variables, train_ldpc.py:40-46,82-88: one source, hetype == 1, the same row for all 96 nodes). */
int fgnn_bn_apply(const void* x, void* y, int64_t R, int32_t C, int32_t dtype, const float* scale,
                  const float* shift, float slope, const void* addend, const void* addend2, const void* addend3,
                  const int32_t* addend_period, fgnn_stream_t stream);
int fgnn_bn_backward(const void* x, const void* gy, void* gx, int64_t R, int32_t C, int32_t dtype,
                     const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                     float* gweight, float* gbias, void* workspace, int64_t workspace_bytes, void* fold_scratch,
                     fgnn_stream_t stream);
/* The element-wise half of fgnn_bn_backward alone, from sums somebody else finalised: dsum [2][C] = (dbeta, dgamma), as
 * fgnn_block_tail_backward leaves them for the 64-channel BatchNorm in front of conv2. */
int fgnn_bn_backward_apply(const void* x, const void* gy, void* gx, int64_t R, int32_t C, int32_t dtype, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float slope, const float* dsum,
                           fgnn_stream_t stream);

/*
 * Inference forward of a whole mp_conv_residual block (mp_nn_residual.py:39-56) in one kernel:
 *   a1 = LeakyReLU(s1 * (x W1^T) + t1)  ->  z = max_j sum_e etype * (a1 filters)[idx]  ->  a2 = ReLU(s2 * z + t2)
 *   y  = LeakyReLU(s3 * (a2 W2^T) + t3) (+ addend + addend1 + addend2)
 * with every eval-mode BatchNorm and bias folded into the per-channel float32 affines (s, t).  d describes the inner
 * message operator (nin = nou = 64, net = 4, max, NO_EXTENSION, bf16, edge-type-fastest etype, k in {3, 6}) but its
 * x / y strides those of the BLOCK's channel-fastest input x [B,N,nin] and output y [B,M,nout], nin / nout in
 * {64,128,256}; W1 [64][nin] and W2 [nout][64] are float32 conv weights [out][in]; s1,t1,s2,t2 are [64], s3,t3
 * [nout]; addend, addend1, addend2 (each or NULL; addend first) have y's layout and are summed in f32 — the layer's running
 * sum, residual and skip terms (factor_mpnn_sp.py:139-168) without a separate sum pass.  FGNN_EUNSUPPORTED outside that family.
 */
int fgnn_mpconv_block_forward(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                              const float* W1, const float* s1, const float* t1, const float* filters,
                              const float* s2, const float* t2, const float* W2, const float* s3, const float* t3,
                              float slope, int32_t nin, int32_t nout, const void* addend, const void* addend1, const void* addend2, void* y,
                              fgnn_stream_t stream);

/* fgnn_mpconv_block_forward with per-sample ROW addends (ABI 11): bit a of addend_row_mask marks addend a as [B][nout] — one row per
 * sample, added to every destination.  That is the LDPC hyper-factor's message to the variables (/root/reference/train_ldpc.py:40-46,
 * 82-88: one source node, identical single edges): fgnn_mpconv_block_forward_fanout run with M = 1 forms it once per codeword and
 * the 96 identical rows the reference materialises are neither written nor read. */
int fgnn_mpconv_block_forward_rows(const fgnn_mpconv_desc* d, const void* x, const int64_t* nn_idx, const void* etype,
                                   const float* W1, const float* s1, const float* t1, const float* filters,
                                   const float* s2, const float* t2, const float* W2, const float* s3, const float* t3,
                                   float slope, int32_t nin, int32_t nout, const void* addend, const void* addend1, const void* addend2,
                                   int32_t addend_row_mask, void* y, fgnn_stream_t stream);

/* The same block around the hyper-factor FAN-OUT call: inner operator with N = 1 source, k = 1, one edge type
 * (filters [64][64]); x is the block's input [B, nin], y its output [B, M, nout]. */
int fgnn_mpconv_block_forward_fanout(const fgnn_mpconv_desc* d, const void* x, const void* etype, const float* W1,
                                     const float* s1, const float* t1, const float* filters, const float* s2,
                                     const float* t2, const float* W2, const float* s3, const float* t3, float slope,
                                     int32_t nin, int32_t nout, const void* addend, const void* addend1, const void* addend2, void* y, fgnn_stream_t stream);
/* Fan-in form (variables -> one factor listening to ALL N nodes in order): d has M = 1, k = N, net = 1, nin = nou = 64
 * and the neighbour list must be the identity (the kernel does not read nn_idx; the caller checks); etype [B][k]
 * (et_sb / et_sk strides, et_sb may be 0); x [B][N][nin] channel-fastest; y / addend [B][nout]. */
int fgnn_mpconv_block_forward_fanin(const fgnn_mpconv_desc* d, const void* x, const void* etype, const float* W1,
                                     const float* s1, const float* t1, const float* filters, const float* s2,
                                     const float* t2, const float* W2, const float* s3, const float* t3, float slope,
                                     int32_t nin, int32_t nout, const void* addend, const void* addend1, const void* addend2, void* y, fgnn_stream_t stream);

/*
 * §8f-3 — ONE WHOLE 64 -> 64 `FactorNN` layer of the LDPC model in one kernel, inference
 * (/root/reference/lib/model/mpnn/factor_mpnn_sp.py:136-168: v2v / f2f `iid_mapping_in` maps, the F->V and V->F
 * `mp_conv_residual` blocks of the parity checks and of the hyper-factor, residual and skip-link sums):
 *     var'  = ReLU(IN(Wvv var))  + f2v_parity(fac0) + f2v_hyper(fac1) + [residual] var  + skip_var
 *     fac0' = ReLU(IN(Wff fac0)) + v2f_parity(var)                    + [residual] fac0 + skip_fac0
 *     fac1' =                      v2f_hyper(var)                     + [residual] fac1 + skip_fac1
 * States are bf16 channel-fastest: var [B][96][64], fac0 [B][48][64], fac1 [B][64]; skip_* (NULL = none) and out_* have the
 * same layouts.  idx_v2f [48][6] / idx_f2v [96][3]: the neighbour tables, SHARED by the batch (strides in elements);
 * et_* the parity edge types [B][M][k][4] (edge-type-fastest) with batch strides in elements; het_* the hyper-factor's
 * per-edge weights ([96] bf16, shared by the batch; NULL = ones, what train_ldpc.py:60-75 passes).
 * params: fgnn_factor_layer_param_count() float32 values — Wvv [64][64], Wff [64][64] ([out][in]), then for the blocks
 * V->F parity, F->V parity, V->F hyper, F->V hyper: W1 [64][64], s1 [64], t1 [64], filters [64][64 * net], s2, t2, W2
 * [64][64], s3, t3, with conv / operator biases and the eval-mode BatchNorms folded into the (s, t) affines as for
 * fgnn_mpconv_block_forward.  slope: the blocks' LeakyReLU slope.
 */
int64_t fgnn_factor_layer_param_count(void);
int fgnn_factor_layer_forward(int32_t B, const void* var, const void* fac0, const void* fac1, const void* skip_var,
                              const void* skip_fac0, const void* skip_fac1, const int64_t* idx_v2f, int32_t idx_v2f_sm,
                              int32_t idx_v2f_sk, const int64_t* idx_f2v, int32_t idx_f2v_sm, int32_t idx_f2v_sk,
                              const void* et_v2f, int64_t et_v2f_sb, const void* et_f2v, int64_t et_f2v_sb,
                              const void* het_v2f, const void* het_f2v, const float* params, int32_t residual, float slope,
                              void* out_var, void* out_fac0, void* out_fac1, fgnn_stream_t stream);

/*
 * out = inputs[0] + ... + inputs[n-1] (n <= 8) over dense arrays of `numel` elements in one pass — the gradient of
 * a state that fans out into several consumers (factor_mpnn_sp.py:139-170) instead of autograd's pairwise adds.
 */
int fgnn_sum_n(const void* const* inputs, int32_t n, int64_t numel, int32_t dtype, void* out, fgnn_stream_t stream);

/* The node-axis concatenation of two channel-fastest activations whose ROWS may be strided (a channel slice of a wider activation — what
 * torch.cat's backward hands on): out[s][r] = r < rows_a ? a[s][r] : b[s][r - rows_a], rows of row_bytes, dense result.
 * /root/reference/lib/model/mpnn/factor_mpnn.py:104-112 (the stacking of variables and factors, and autograd's gradient of the two
 * slices).  Multiples of 16 bytes, 16-byte aligned pointers.  ABI >= 13. */
int fgnn_concat_rows(const void* a, const void* b, void* out, int64_t samples, int64_t rows_a, int64_t rows_b, int64_t row_bytes,
                     int64_t a_sample_stride_bytes, int64_t a_row_stride_bytes, int64_t b_sample_stride_bytes,
                     int64_t b_row_stride_bytes, fgnn_stream_t stream);

/* out[s][n] = [ a[s][n] | b[s][n] ], s < samples, n < inner: two arrays of chunks interleaved chunk by chunk in one pass, dense result —
 * torch.cat of two channel-fastest activations along the node axis (inner = 1, chunk = a sample's nodes x channels;
 * /root/reference/lib/model/mpnn/factor_mpnn.py:104-107: the variables and one factor type stacked for a block) or along the channel axis
 * (inner = nodes, chunk = channels; factor_mpnn.py:116: the blocks' messages in front of the merge map).  The inputs may be slices of
 * larger activations: per-sample and per-chunk strides in bytes.  Sizes and strides are multiples of 16 bytes, pointers 16-byte aligned;
 * any element type.  ABI >= 13. */
int fgnn_concat_pair(const void* a, const void* b, void* out, int64_t samples, int64_t inner, int64_t chunk_a_bytes, int64_t chunk_b_bytes,
                     int64_t a_sample_stride_bytes, int64_t a_chunk_stride_bytes, int64_t b_sample_stride_bytes,
                     int64_t b_chunk_stride_bytes, fgnn_stream_t stream);
/*
 * out [B][C] = sum over the M rows of each sample of g [B][M][C] (dense channel-fastest rows): the backward of a per-sample row
 * broadcast over the sample's nodes — the LDPC hyper-factor's message to the variables (train_ldpc.py:40-46,82-88: one source node,
 * hetype == 1) is carried as [B][C] and added by the consumer with `addend_period` (fgnn_block_tail_apply, fgnn_bn_apply).
 */
int fgnn_node_sum(const void* g, void* out, int64_t B, int32_t M, int32_t C, int32_t dtype, fgnn_stream_t stream);

/*
 * One Adam step (torch.optim.Adam's rule, no amsgrad — the optimizer of the reference's training scripts, e.g.
 * train_ldpc.py:160-166) over a flat f32 parameter buffer and its flat gradient in a single pass:
 *   g' = grad * grad_scale + weight_decay * param;  exp_avg, exp_avg_sq updated in place;  param updated in place.
 * bf16_mirror (or NULL): a bf16 copy of the parameters refreshed in the same pass (what the bf16 GEMMs read).
 * grad_scale = 1 / world_size turns the all-reduced SUM of the data-parallel ranks into their mean.  step >= 1.
 */
int fgnn_flat_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_mirror, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int64_t step,
                   fgnn_stream_t stream);

/*
 * The same Adam step with nothing step-dependent among the launch arguments, so that it can be recorded into a hipGraph
 * (together with the gradient all-reduce in front of it) and replayed: *step_dev (int64 in device memory, starts at 0, advanced
 * by one per call) replaces `step`, *lr_dev (f32 in device memory; a scheduler overwrites it between replays) replaces `lr`,
 * coef_dev is two floats of device scratch.  Two launches (a one-thread kernel forms the bias-corrected coefficients).
 */
int fgnn_flat_adam_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_mirror, int64_t n,
                       const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                       int64_t* step_dev, float* coef_dev, fgnn_stream_t stream);

/*
 * The edge-type MLP in front of the operator, etype = W2 ReLU(W1 efeature + b1) + b2 with Cin <= 8 -> 64 -> net <= 4
 * (`emodel_f2v / emodel_v2f`, /root/reference/train_ldpc.py:32-38,68-69), without the 64-channel hidden tensor ever
 * reaching memory.  x is bf16 with element (b, c, r) at b*x_sb + c*x_sc + r*x_sr (r one of the E = M*k edge rows of
 * sample b); y is [B][E][net] bf16 (edge-type-fastest, what the operator kernels read); parameters are f32.  The
 * backward recomputes the hidden units and ACCUMULATES the parameter gradients (any may be NULL); gy element (b, e, r)
 * sits at b*gy_sb + e*gy_se + r*gy_sr so either layout of the operator's getype is read in place.  Edge features
 * take no gradient.
 */
int fgnn_edge_mlp_forward(const void* x, int64_t x_sb, int64_t x_sc, int64_t x_sr, const float* W1, const float* b1,
                          const float* W2, const float* b2, void* y, int64_t B, int32_t E, int32_t Cin, int32_t net,
                          fgnn_stream_t stream);
int64_t fgnn_edge_mlp_workspace_bytes(int64_t B, int32_t E);
int fgnn_edge_mlp_backward(const void* x, int64_t x_sb, int64_t x_sc, int64_t x_sr, const void* gy, int64_t gy_sb,
                           int64_t gy_se, int64_t gy_sr, const float* W1, const float* b1, const float* W2, int64_t B, int32_t E, int32_t Cin, int32_t net, float* gW1,
                           float* gb1, float* gW2, float* gb2, void* workspace, int64_t workspace_bytes,
                           fgnn_stream_t stream);

/*
 * The LDPC data path in front of the decoder (the reference's only native code, lib/data/MNC, plus the numpy feature
 * construction of lib/data/ldpc_dataset.py:92-106,222-236), per batch instead of per codeword:
 *   fgnn_ldpc_encode            cw [B][K+P] = [s | G s mod 2] for messages s [B][K] (bytes 0/1) — what
 *                               `s2t(s, K, P, Gfile, smn=True)` returns (MNC_py.cpp:22-83); gmask[r] is row r of G
 *                               packed over the K <= 64 message bits.
 *   fgnn_ldpc_channel_features  `t2y` (MNC_py.cpp:86-102) with its random draws as inputs (z1, z2 ~ N(0,1), u ~ U[0,1)):
 *                               y = 2 gcx (cw - 1/2) + z1, plus gcx sigma_b z2 where sigma_b >= 1e-20 and u < rho,
 *                               gcx = 10^(snr_db/20); then the model inputs gathered from y along the incidence lists
 *                               var_to_factors [nvar][dv] / factor_to_vars [nchk][dc]: node [B][2][nvar] (y, snr_db),
 *                               hop [B][dc][nchk], ef_f2v [B][dc+1][nvar][dv], ef_v2f [B][dc+1][nchk][dc] in `dtype`.
 */
int fgnn_ldpc_encode(const uint8_t* s, const uint64_t* gmask, int64_t B, int32_t K, int32_t P, uint8_t* cw,
                     fgnn_stream_t stream);
int fgnn_ldpc_channel_features(const uint8_t* cw, const float* snr_db, const float* sigma_b, float rho, const float* z1,
                               const float* u, const float* z2, const int32_t* var_to_factors,
                               const int32_t* factor_to_vars, int64_t B, int32_t nvar, int32_t nchk, int32_t dv,
                               int32_t dc, int32_t dtype, float* y, void* node, void* hop, void* ef_f2v, void* ef_v2f,
                               fgnn_stream_t stream);
/* fgnn_ldpc_channel_features with t2y's draws (`xt::random::randn` / `rand`, MNC_py.cpp:89,94-97) made inside the kernel by a
 * counter-based generator — Philox4x32-10, key = seed, counter = (index of the codeword bit in the batch, offset) — so no noise
 * tensor exists in HBM and a batch is reproducible from (seed, offset) whatever the launch geometry. */
int fgnn_ldpc_channel_features_rng(const uint8_t* cw, const float* snr_db, const float* sigma_b, float rho, uint64_t seed,
                                   uint64_t offset, const int32_t* var_to_factors, const int32_t* factor_to_vars, int64_t B,
                                   int32_t nvar, int32_t nchk, int32_t dv, int32_t dc, int32_t dtype, float* y, void* node,
                                   void* hop, void* ef_f2v, void* ef_v2f, fgnn_stream_t stream);

/*
 * The training loss behind the decoder (/root/reference/train_ldpc.py:222-227), two short launches forward and one backward:
 *     loss[0] = mean_{b,j} BCEWithLogits(logits[b][j], label[b][j]) + mse_weight * mean_b (pred[b] - 10^(sigma_b[b] / 20))^2
* logits [B][n] dense, `dtype` FGNN_F32 / FGNN_BF16; label [B][n], pred [B], sigma_b [B], loss [1]: f32.  Summed in double, per
 * workgroup and then over the workgroups in order (bit-reproducible); workspace: fgnn_ldpc_loss_workspace_bytes() bytes, 8-byte aligned.  backward: glogits [B][n] (dtype of logits) and gpred [B] (f32) = gloss[0] * d loss.
 */
int fgnn_ldpc_loss_forward(const void* logits, const float* label, const float* pred, const float* sigma_b, int64_t B, int32_t n,
                           int32_t dtype, float mse_weight, float* loss, void* workspace, int64_t workspace_bytes, fgnn_stream_t stream);
int64_t fgnn_ldpc_loss_workspace_bytes(void);
int fgnn_ldpc_loss_backward(const void* logits, const float* label, const float* pred, const float* sigma_b, const float* gloss,
                            int64_t B, int32_t n, int32_t dtype, float mse_weight, void* glogits, float* gpred, fgnn_stream_t stream);

/*
 * The reference's classical baseline: MacKay's sum-product decoder `zb2x(z, k, n, Afile, 1, loops)` ->
 * `bndecode` (lib/data/MNC/MNC_py.cpp:110-183, bnd/bnd.cpp:150-371; defaults clip 0.9999999999, tinydiv 1e-40, target
 * syndrome 0) for B words at once, float64, reference operation order: results equal the compiled reference's bit for
 * bit.  bias [B][N] = P(bit = 1) (`y2b`).  Incidence as device tables: col_ptr [N+1] (edge col_ptr[n]+u = variable n's
 * u-th check in the alist file's order), row_ptr [M+1], row_edge / row_var [E] (a check's edges / variables in
 * increasing variable order).  <= 16 edges per variable and per check, N, E <= 1024.  Outputs: x [B][N] hard
 * decisions, q1 [B][N] pseudo-posteriors (or NULL), viol [B] violated checks at exit, iters [B] iterations run.
 */
int fgnn_ldpc_decode(const double* bias, const int32_t* col_ptr, const int32_t* row_ptr, const int32_t* row_edge,
                     const int32_t* row_var, int64_t B, int32_t N, int32_t M, int32_t E, int32_t loops, uint8_t* x,
                     double* q1, int32_t* viol, int32_t* iters, fgnn_stream_t stream);

const char* fgnn_last_error(void);
/* Name (as rocprofv3 prints it) of the kernel the calling thread's last forward/backward dispatched to. */
const char* fgnn_last_kernel(void);
/* Bumped whenever an entry point is added or an argument / descriptor field changes meaning.  The host binding
 * (fgnn_amd/_hip.py: ABI_VERSION) checks it BEFORE binding symbols, so a stale library is reported as a version
 * mismatch and not as a missing symbol or a misread field.  4: round-2 additions (flat_adam, factor_layer_*,
 * ldpc_channel_features_rng, backward_reduces_getype, desc.reserved = in-degree | GETYPE_REDUCED); 5: fgnn_block_tail_*;
 * 6: fgnn_block_head_backward.  11: fgnn_mpconv_block_forward_rows.  12: fgnn_block_tail_backward_moments,
 * fgnn_block_tail_wgrad_finish, fgnn_block_tail_moments_bytes. */
#define FGNN_ABI_VERSION 13
/* Arithmetic of the f32 synthetic-PGM operator's BACKWARD (16 edge types, ORIG_WITH_NEIGHBOR / ORIG_WITH_DIFF, 64 -> 64, max: the
 * autograd of /root/reference/lib/model/mpnn/mp_nn.py:136-175 as train_syn_*.py reaches it): 2 (default) = every f32 operand of the three
 * GEMMs as two bf16 pieces on the bf16 matrix cores (gradients within 5e-6 of the exact kernel's), 3 = three pieces (4e-7), 0 = f32 matrix
 * cores (exact f32 products).  Process-wide; the environment variable FGNN_EXT_BWD_PIECES gives the initial value.  Returns the previous
 * setting; other values leave it unchanged.  No reference counterpart. */
int fgnn_set_ext_backward_pieces(int pieces);

/* Diagnostic: a one-thread kernel on `stream` writes the device's constant 100 MHz clock to *dst (uint64).  Inside a captured step
 * it tells when that point of the stream is reached in a replay without a profiler attached.  No reference counterpart. */
int fgnn_stamp(void* dst, void* stream);

/* Diagnostic: one thread on `stream` spins for `ticks` of the 100 MHz device clock (<= 1e8).  At the head of a captured step it lets
 * the host enqueue the rest of the graph before the device starts on it (a kernel trace taken under a profiler then shows the graph's
 * own schedule, not the profiler-slowed host's enqueue order).  No reference counterpart. */
int fgnn_spin(int64_t ticks, void* stream);

int fgnn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FGNN_HIP_H */
