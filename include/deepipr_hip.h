/* deepipr_hip.h -- C ABI of the MI355X (gfx950) passport-layer kernels.
 *
 * Drop-in boundary of the DeepIPR hot path (reference kamwoh/DeepIPR, a pure-Python/PyTorch
 * code base with no FFI of its own).  Each entry point replaces a run of stock
 * ATen ops inside one reference method; the file:line after "replaces" is that run.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory (HBM) unless named host_*
 *   - all tensors fp32, dense, NCHW / OIHW; `HW` = H*W of the activation plane
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it: no
 *     allocation, no synchronisation, no global mutable state -> re-entrant and capturable in a
 *     hipGraph.  Scratch memory is passed in by the caller (`*_workspace_bytes` gives its size).
 *   - return value: DEEPIPR_OK (0) or a negative DEEPIPR_E* code; deepipr_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 *   - inputs are never written; outputs are fully overwritten (the one accumulate-into entry point is
 *     deepipr_gamma_beta_bwd_acc, named for it).
 *   - reductions are fixed-order (no float atomics): results are bit-reproducible run to run.
 */
#ifndef DEEPIPR_HIP_H
#define DEEPIPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEEPIPR_OK 0
#define DEEPIPR_EINVAL (-1)   /* bad shape / null pointer / misaligned pointer / workspace too small */
#define DEEPIPR_ELAUNCH (-2)  /* hipLaunchKernel reported an error */
#define DEEPIPR_EUNSUPPORTED (-3)  /* shape outside the fused form (nothing was enqueued): use the unfused entry points */

#define DEEPIPR_ABI_VERSION 12

int deepipr_abi_version(void);
const char *deepipr_last_error(void);

/* ------------------------------------------------------------------ external events of a captured step
 * A train step captured into a hipGraph cannot call back into the host, yet a data-parallel step wants to start the
 * all-reduce of a gradient bucket the moment backward has finished it (the reference gathers gradients only after
 * backward, experiments/trainer.py:92-93).  deepipr_event_record on a CAPTURING stream adds an event-record node
 * behind everything captured so far (hipGraphAddEventRecordNode on the capture's graph, the capture continues behind
 * it): each launch of the graph records the event when execution
 * reaches that point; after the launch call the host makes another stream wait for it (deepipr_stream_wait_event) and
 * enqueues the collective there -- no collective is captured, the graph is not split.  On a stream that is not
 * capturing, deepipr_event_record is a plain hipEventRecord.  Events are created without timing. */
int deepipr_event_create(void **event);
int deepipr_event_destroy(void *event);
int deepipr_event_record(void *event, void *stream);
int deepipr_stream_wait_event(void *stream, void *event);
/* Block the calling HOST thread until the event's latest record has completed (hipEventSynchronize). */
int deepipr_event_synchronize(void *event);

/* Opt-in in-situ timing (the one piece of process-global state, off by default): while enabled every
 * kernel below is dispatched through hipExtLaunchKernelGGL with a start and a stop hipEvent attached to
 * its own dispatch packet on the launch stream, so the elapsed time is the kernel's execution time (what
 * rocprofv3's kernel trace reports), free of event-record overhead.  read() waits for the events and
 * returns the accumulated milliseconds / launch count of one kernel since the last enable(1).  Must be
 * off during hipGraph capture. */
#define DEEPIPR_K_POOLED_PATCH_MEAN 0
#define DEEPIPR_K_GAMMA_BETA_FWD 1
#define DEEPIPR_K_GAMMA_BETA_BWD 2
#define DEEPIPR_K_AFFINE_FWD 3
#define DEEPIPR_K_AFFINE_BWD 4
#define DEEPIPR_K_REDUCE_PARTIALS 5
#define DEEPIPR_K_PASSPORT_BWD_FINISH 6
#define DEEPIPR_K_SIGN_LOSS_FWD 7
#define DEEPIPR_K_SIGN_LOSS_BWD 8
#define DEEPIPR_K_DKEY 9
#define DEEPIPR_K_RESERVED 10
#define DEEPIPR_K_BN_STATS 11
#define DEEPIPR_K_BN_AFFINE_FWD 12
#define DEEPIPR_K_BN_BWD_REDUCE 13
#define DEEPIPR_K_BN_AFFINE_BWD 14
#define DEEPIPR_K_SGD 15
#define DEEPIPR_K_ADD_RELU 16
#define DEEPIPR_K_BN_RES_FWD 17
#define DEEPIPR_K_BN_RES_BWD 18
#define DEEPIPR_K_GN_FWD 19
#define DEEPIPR_K_GN_BWD 20
#define DEEPIPR_K_CONV_WGRAD 21         /* `bytes` of this slot are FLOPs: its roofline is the fp32 MFMA peak */
#define DEEPIPR_K_CONV_WGRAD_REDUCE 22
#define DEEPIPR_K_CONV_FWD 23           /* FLOPs, like DEEPIPR_K_CONV_WGRAD */
#define DEEPIPR_K_CONV_DGRAD 24
#define DEEPIPR_K_CONV_WGRAD_B3 25      /* the bf16x3 weight gradient: ALGORITHMIC FLOPs (it issues six times as many on the bf16 MFMA) */
#define DEEPIPR_K_CONV_SPLIT_SUM 26     /* sum of the split-K output slabs of deepipr_conv_fwd_ws / _dgrad_ws (bytes) */
#define DEEPIPR_K_CONV_WINO_FWD 27      /* Winograd F(2x2, 3x3) forward: EXECUTED FLOPs (the direct sum's / 2.25) */
#define DEEPIPR_K_CONV_WINO_DGRAD 28
#define DEEPIPR_K_CONV_WINO_WGRAD 29     /* Winograd F(3x3, 2x2) weight gradient: EXECUTED FLOPs */
#define DEEPIPR_K_CONV_WINO_WEIGHTS 30   /* the Winograd weight transform of the pre-transformed form (bytes: 36 in + 66 out per filter and direction) */
#define DEEPIPR_K_CONV1X1_WGRAD 31       /* 1x1 stride-1 weight gradient (deepipr_conv_1x1.inc): FLOPs */
#define DEEPIPR_K_MAXPOOL 32              /* 3x3 stride-2 max-pool forward / backward (bytes) */
#define DEEPIPR_K_RESAMPLE2 33            /* the stride-2 pixel gather / zero-interleaving scatter around a 1x1 stride-2 convolution (bytes) */
#define DEEPIPR_K_CONV1X1_FWD 34          /* 1x1 stride-1 forward / backward-data GEMM over NCHW (deepipr_conv_1x1.inc): FLOPs */
#define DEEPIPR_K_CONV1X1_DGRAD 35
#define DEEPIPR_K_HEAD 36                 /* avg-pool + Linear of the CIFAR-geometry classifier, forward / backward (bytes) */
#define DEEPIPR_PROFILE_KERNELS 37
int deepipr_profile_enable(int on);   /* 1 = reset counters and enable, 2 = resume without reset, 0 = pause */
int deepipr_profile_read(int kernel, double *total_ms, long long *launches);
/* algorithmic HBM bytes of the launches timed so far, for the streaming kernels (0 for the others) */
int deepipr_profile_read_bytes(int kernel, double *total_bytes);
/* Launches issued between deepipr_profile_scope(1) and deepipr_profile_scope(0) are ALSO accounted in a second set of
 * counters (the host opens the scope around the calls that serve passport layers, so that bench.py can report the
 * passport-affine kernels apart from the plain norm layers that share them: `roofline_passport`). */
int deepipr_profile_scope(int scope);
int deepipr_profile_read_scope(int kernel, double *total_ms, long long *launches, double *total_bytes);

/* ------------------------------------------------------------------ passport conv -> global pool
 * m[k] = mean over (b, oh, ow) of im2col(key)[b, k, (oh,ow)], k = (ci*kh + r)*kw + q, kept in f64,
 * for `nkeys` passport tensors of identical shape laid out back to back.  Because conv and the two
 * means are linear, gamma = W_mat . m (see below).  Keys are constant buffers during training, so
 * the host caches `m` per key version.
 * replaces: the im2col half of self.conv(skey) / self.conv(key),
 *           models/layers/passportconv2d.py:148,169 (private twin :146,167).
 * keys  [nkeys][B][Ci][H][W]   m_out [nkeys][Ci*kh*kw] (double) */
int deepipr_pooled_patch_mean(const float *keys, int nkeys, int B, int Ci, int H, int W,
                              int kh, int kw, int stride, int pad, double *m_out, void *stream);

/* gamma[co] = sum_k W[co,k]*m_scale[k],  beta[co] likewise with m_bias; f64 accumulation, one
 * rounding to f32 at the end.
 * replaces: conv -> view(b,c,-1).mean(2) -> mean(0), passportconv2d.py:148-152 (get_scale) and
 *           :169-173 (get_bias); private twin passportconv2d_private.py:146-150,167-171.
 * W [Co][K]   m [2][K] double (scale key first)   gamma, beta [Co] */
int deepipr_gamma_beta_fwd(const float *W, const double *m, int Co, int K,
                           float *gamma, float *beta, void *stream);

/* dW[co,k] = dgamma[co]*m_scale[k] + dbeta[co]*m_bias[k]: the passport branch's
 * contribution to the shared conv weight's gradient (autograd adds the data conv's wgrad).
 * replaces: the two convolution_backward(wgrad) + mean backward chains of passportconv2d.py:148-152,
 *           169-173.   dW [Co][K] */
int deepipr_gamma_beta_bwd(const float *dgamma, const float *dbeta, const double *m,
                           int Co, int K, float *dW, void *stream);
/* The same update ADDED to dW, which already holds the data convolution's wgrad of the shared weight
 * (passportconv2d.py:148,169,218 use one W three times): dW[co,k] += dgamma[co]*m_scale[k] + dbeta[co]*m_bias[k].
 * 8 B per weight in one launch instead of a fresh 4 B write plus autograd's 12 B add kernel. */
int deepipr_gamma_beta_bwd_acc(const float *dgamma, const float *dbeta, const double *m,
                               int Co, int K, float *dW, void *stream);

/* The same three operations for SEVERAL layers in ONE launch each (a net's passport layers: ResNet18's five layer4
 * weights are 33.6 MB, and one launch over all of them streams W at the bandwidth a single 9.4 MB launch cannot
 * reach -- round 2 measured 0.17 of the HBM roofline per layer, latency-bound).  `layers` is a HOST array of n <=
 * DEEPIPR_GEMV_MAX_LAYERS descriptors of device pointers; it is read during the call only (the descriptors travel in
 * the kernel arguments).  The single-layer entry points above are the n = 1 case of these.
 * replaces: the get_scale() / get_bias() passport convs of every passport layer of a forward,
 *           models/layers/passportconv2d.py:148-152,169-173, and their backward. */
#define DEEPIPR_GEMV_MAX_LAYERS 16
typedef struct DeepiprGemvLayer {
    const float *W;          /* [Co][K] */
    const double *m;         /* [2][K] pooled patches, scale key first */
    float *gamma, *beta;     /* [Co] out */
    int Co, K;
} DeepiprGemvLayer;
typedef struct DeepiprRank2Layer {
    const float *dgamma, *dbeta;   /* [Co] */
    const double *m;               /* [2][K] */
    float *dW;                     /* [Co][K]; accumulate != 0: holds the data convolution's wgrad and is added to */
    int Co, K;
} DeepiprRank2Layer;
int deepipr_gamma_beta_fwd_multi(const DeepiprGemvLayer *layers, int n, void *stream);
int deepipr_gamma_beta_bwd_multi(const DeepiprRank2Layer *layers, int n, int accumulate, void *stream);

/* Gradient w.r.t. the passport tensors themselves (keys made nn.Parameters by
 * passport_attack_3.py:232-243).  dkeys[j][b][ci][ih][iw] = sum over patches covering (ih,iw) of
 * (sum_co d[j][co]*W[co,ci,r,q]) * inv_n,  d[0]=dgamma (scale key), d[1]=dbeta (bias key).
 * workspace: deepipr_gamma_beta_dkey_workspace_bytes().   dkeys [2][B][Ci][H][W] */
size_t deepipr_gamma_beta_dkey_workspace_bytes(int Ci, int kh, int kw);
int deepipr_gamma_beta_dkey(const float *dgamma, const float *dbeta, const float *W, int Co,
                            int B, int Ci, int H, int Wd, int kh, int kw, int stride, int pad,
                            float *dkeys, void *workspace, void *stream);

/* ------------------------------------------------------------------ passport affine (+ReLU)
 * y = gamma[c]*xhat + beta[c] (separate mul and add roundings, as ATen's mul, add), ReLU if relu!=0.
 * replaces: `scale * x + bias` and relu_, passportconv2d.py:220-222 (private :217-218).
 * One pass: 8 B / element of HBM traffic.  xhat, y [N][C][HW] */
int deepipr_affine_relu_fwd(const float *xhat, const float *gamma, const float *beta, float *y,
                            int N, int C, int HW, int relu, void *stream);

/* dxhat = dy*mask*gamma, dgamma[c] = sum(dy*mask*xhat), dbeta[c] = sum(dy*mask); the ReLU mask is
 * recomputed as gamma*xhat+beta > 0 (relu'(0)=0).  One pass over dy and xhat (12 B / element) plus a
 * fixed-order finish over the per-split partial sums.
 * replaces: threshold_backward, mul/sum backward of passportconv2d.py:220-222.
 * workspace: deepipr_affine_relu_bwd_workspace_bytes(N, C, HW) bytes. */
size_t deepipr_affine_relu_bwd_workspace_bytes(int N, int C, int HW);
int deepipr_affine_relu_bwd(const float *dy, const float *xhat, const float *gamma, const float *beta,
                            float *dxhat, float *dgamma, float *dbeta, int N, int C, int HW, int relu,
                            void *workspace, void *stream);

/* ------------------------------------------------------------------ hinge sign loss on gamma
 * loss = sum_c alpha*relu(-b*gamma + margin) + l2*sum_c gamma^2;  acc = mean(sign(b) == sign(gamma));
 * bits[c] = sign(gamma[c]) in {-1,0,+1}.
 * replaces: SignLoss.add / get_loss / get_acc, models/losses/sign_loss.py:20,27,52-54 and the
 *           signature read-out experiments/trainer_private.py:50,57.
 * gamma, b [C]; loss, acc: one float each; bits [C] int8 (may be NULL). */
int deepipr_sign_loss_fwd(const float *gamma, const float *b, float alpha, float margin, float l2,
                          int C, float *loss, float *acc, int8_t *bits, void *stream);

/* dgamma[c] = dloss * ( -alpha*b*[ -b*gamma+margin > 0 ] + 2*l2*gamma ).  `dloss` is a device scalar
 * (the upstream gradient), so no host synchronisation is needed. */
int deepipr_sign_loss_bwd(const float *dloss, const float *gamma, const float *b, float alpha,
                          float margin, float l2, int C, float *dgamma, void *stream);

/* ------------------------------------------------------------------ fused layer entry points
 * Forward of one passport layer after the norm: gamma/beta from the pooled sums, sign loss, affine.
 *   launch 1: gamma_beta (one workgroup per output channel row)
 *   launch 2: affine+ReLU over xhat; workgroup 0 also emits loss / acc / bits.
 * loss/acc/bits may be NULL when the layer has no sign loss (alpha == 0, passportconv2d.py:45-48). */
int deepipr_passport_fwd(const float *xhat, const float *W, const double *m,
                         const float *b, float alpha, float margin, float l2,
                         int N, int C, int HW, int K, int relu,
                         float *y, float *gamma, float *beta, float *loss, float *acc, int8_t *bits,
                         void *stream);

/* Backward of the same: launch 1 = affine backward (dxhat + per-split partial sums), launch 2 =
 * finish the partial sums, add dgamma_extra / dbeta_extra (gradients arriving at gamma / beta from
 * elsewhere, may be NULL) and the sign-loss gradient (scaled by the device scalar *dloss, NULL = no sign loss), write
 * dgamma/dbeta and the rank-2 update dW.  dW == NULL: dgamma / dbeta only (sign-loss gradient and extras included);
 * the caller then adds the rank-2 update to the data convolution's wgrad with deepipr_gamma_beta_bwd_acc (same for
 * deepipr_passport_gn_bwd).  workspace: deepipr_passport_bwd_workspace_bytes(). */
size_t deepipr_passport_bwd_workspace_bytes(int N, int C, int HW);
int deepipr_passport_bwd(const float *dy, const float *xhat, const float *gamma, const float *beta,
                         const double *m, const float *b, float alpha, float margin, float l2,
                         const float *dloss, const float *dgamma_extra, const float *dbeta_extra,
                         int N, int C, int HW, int K, int relu,
                         float *dxhat, float *dW, float *dgamma, float *dbeta,
                         void *workspace, void *stream);

/* ------------------------------------------------------------------ BatchNorm-fused passport layer
 * The passport layer's norm is BatchNorm2d(o, affine=False) by default (passportconv2d.py:57-58).  These two
 * entry points take the CONV OUTPUT x and do norm + passport affine + ReLU (+ sign loss) without ever
 * materialising the normalised activation:
 *   forward  (3 launches): per-channel sums of x, x^2  ->  gamma/beta GEMV + statistics finish (mean, invstd,
 *            running-stat update with `momentum`, unbiased variance)  ->  y = relu(gamma*((x-mean)*invstd)+beta)
 *   backward (3 launches): sums of dz*xhat and dz  ->  dgamma/dbeta/dW finish  ->
 *            dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))
 * replaces: self.bn(x) (native_batch_norm + native_batch_norm_backward, passportconv2d.py:219) and everything
 *           deepipr_passport_fwd / _bwd replace.
 * table, table_out: [C][8] floats {mean, invstd, gamma, beta, c2, c3, -, -}; `table` from forward is the only
 * per-channel state backward needs.  W == NULL selects the public branch of PassportPrivateBlock: gamma_in /
 * beta_in are the learnable scale / bias (passportconv2d_private.py:140-141,162-163), dW is not produced.
 * training == 0 uses the running statistics (eval mode) and backward treats them as constants.
 * running_mean / running_var / num_batches_tracked may be NULL (track_running_stats=False) when training.
 * workspace: deepipr_passport_bn_workspace_bytes(N, C, HW) bytes (forward needs it only when training).
 *
 * Register-resident single pass.  When the layer's activations fit in the register file of the device (all of
 * the CIFAR-shape nets at batch <= 128 per GPU) and HW % 4 == 0, both directions run as ONE kernel that loads
 * x (backward: dy and x) once, forms the channel sums, and produces y (dx) from registers: 8 / 12 B per element
 * instead of 12 / 20, one launch instead of three (+ the gamma/beta GEMV, resp. the dW update, for layers with W).
 * Layers with fewer channels than the device has CUs split every channel over several workgroups, which exchange
 * their partial sums inside the launch through `sync`: DEEPIPR_SYNC_WORDS 32-bit words (8-byte {payload, tag}
 * granules, the tag being a per-slot call counter) that are zero before their first use, are owned by this library
 * from then on (never reset, also not across hipGraph replays) and are not shared by calls that can run
 * concurrently.  That form needs all its workgroups co-resident, so pass
 * sync == NULL whenever another kernel may occupy CUs of the device at the same time (e.g. a collective on a
 * second stream); channel-owning layers (C >= CUs) then still take the single pass, the others the 3-launch
 * form.  Maps too large for the register file at once run as channel-range passes of the same kernels
 * (deepipr_passport_bn_passes).  If a bounded in-kernel wait ever expires, word [DEEPIPR_SYNC_TIMEOUT_WORD] becomes non-zero AND the
 * affected channels' statistics are poisoned with NaN (so y / dx, the loss and every gradient turn NaN): the
 * failure cannot pass silently.  The host layer checks the word once per epoch and raises.
 * deepipr_set_resident(0) disables the single-pass kernels process-wide (testing), (1) restores the default.
 *
 * Fused residual tail (single-pass form only).  The last layer of a residual block is followed by
 * `out = relu(layer(x) + shortcut)` (models/resnet_passport.py:77-84).  forward: residual != NULL makes the kernel
 * write that `out` into y directly (reads x and the shortcut, 12 B/element, the layer's own output is never
 * written).  backward: tail_out = that `out`, dy (+ optional dy2: `out` has two consumers, see
 * deepipr_relu_bwd2) the gradient w.r.t. it; the kernel forms d = (dy + dy2) * [tail_out > 0], writes it to dres
 * (the shortcut's gradient) and continues with d as the layer's upstream gradient (20-24 B/element instead of
 * 28 in two kernels).  dy2 without tail_out (single-pass form only): the layer's output simply has two consumers
 * (the stem feeds layer1's first conv and its identity shortcut); the upstream gradient is dy + dy2, summed in the
 * kernel (16 B/element) instead of by a separate add pass.  deepipr_passport_bn_resident(N, C, HW, have_sync) -> bit 0: forward, bit 1: backward take
 * the single-pass form for this shape; with a residual / tail_out outside it the entry points return
 * DEEPIPR_EUNSUPPORTED and enqueue nothing. */
#define DEEPIPR_SYNC_WORDS (2 * (256 * 30 * 4 + 4096 + 63 * 2048) + 16)   /* 8-byte granules: 256 channels x (2+4+8+16) slices x 4, two retired regions, one region of 2 048 per slice count 2 .. 64 of the channel-range form (ABI v11), + flags: 1.3 MB */
#define DEEPIPR_SYNC_TIMEOUT_WORD (2 * (256 * 30 * 4 + 4096 + 63 * 2048))
int deepipr_set_resident(int mode);
#ifdef DEEPIPR_TEST_HOOKS
/* MEASUREMENT / TEST BUILD ONLY (libdeepipr_hip_trace.so, `make -C deepipr_amd/csrc trace`): the production library
 * exports neither symbol and its kernels carry neither the knobs nor the hooks.
 * deepipr_debug_tune: planning knobs of the single-pass kernels, process-wide: "split_full" (default 1: split
 * channels over workgroups whenever they do not fill the chip; 0: only below half), "xcd_map" (default 0; 1: a
 * channel's slices on workgroups with equal index % 8), "small_t"; and the time-out test hooks "exchange_spin" (bound
 * of the in-launch wait, <= 0 restores the default) and "exchange_drop" (slice that never publishes its partial sums:
 * forces the time-out path; -1 = none). */
int deepipr_debug_tune(const char *key, int value);
/* Phase tracing of the single-pass kernels: while device_buffer != NULL, thread 0 of every workgroup of
 * k_bn_res_fwd / _bwd writes five 100 MHz wall-clock stamps to device_buffer[block][8]: entry, loads consumed +
 * workgroup sums formed, exchange done, channel table ready, all stores issued.  The buffer needs 8 * 8 bytes per
 * workgroup (<= 2 * CUs + 1 of them); pass NULL to switch tracing off. */
int deepipr_debug_trace(unsigned long long *device_buffer);
/* The same for the Winograd convolution kernels (k_conv_wino): thread 0 of every workgroup writes {100 MHz wall clock,
 * shader clock} pairs to device_buffer[block][32][2]: 0 entry, 1 first chunk staged, 2 MFMA loop done, 3 transformed
 * sums in LDS, 4 stores issued, then for the first nine chunks: step top, LDS writes drained, barrier passed
 * (tools/wino_trace.py).  64 * 8 bytes per workgroup; NULL switches it off. */
int deepipr_debug_wino_trace(unsigned long long *device_buffer);
#endif
int deepipr_passport_bn_resident(int N, int C, int HW, int have_sync);
/* Workgroups per channel the single-pass kernels would use for this shape when exchange words are passed (the larger
 * of forward and backward; 1 = no in-launch exchange, also when the shape is not single-pass).  Host-side planning
 * only: lets a driver that overlaps collectives with compute know WHICH layer calls depend on co-residency
 * (deepipr_amd/experiments/staged.py keeps collectives away from exactly those stages). */
int deepipr_passport_bn_slices(int N, int C, int HW);
/* Launches the single-pass form takes for this shape when exchange words are passed: 1 = the whole layer in one
 * launch; > 1 = channel-range passes of a map too large for the register file (ImageNet-size maps: each pass holds as
 * many channels as fill the chip once every channel is split over up to 64 workgroups; still 8 / 12 B per element);
 * 0 = not single-pass (three launches). */
int deepipr_passport_bn_passes(int N, int C, int HW, int backward);
/* A projection block's LAST TWO norm layers and its tail in one launch per direction
 * (models/resnet_passport.py:67-85: out = relu(block_2(h) + block_s(x)), both ConvBlocks = conv -> BatchNorm2d(affine)
 * -> ReLU, models/layers/conv2d.py:5-36; relu_a / relu_b say whether the inner ReLUs exist -- the reference builds
 * both with one).  xa = conv_2's output, xb = the shortcut conv's, same
 * shape [N, C, HW]; gamma_* / beta_* the norms' learnable weight / bias; batch statistics (training), running
 * statistics updated as in deepipr_passport_bn_fwd; table_a / table_b [C][8] receive {mean, invstd, gamma, beta}
 * for backward.  forward reads xa, xb and writes `out` (12 B/element; the two layers' own outputs are never
 * written); backward forms d = (dy + dy2) * [out > 0] in registers (dy2 may be NULL), masks it with each layer's own
 * ReLU and writes dxa, dxb, dgamma_* and dbeta_* (28 B/element against 36 of the two separate launches).  Bit-identical to deepipr_passport_bn_fwd /_bwd
 * called for the shortcut layer and then, with residual / tail_out, for the other.  Single-pass shapes only:
 * deepipr_bn_dual_tail_supported(N, C, HW, have_sync) -> 1, otherwise the entry points return DEEPIPR_EUNSUPPORTED
 * and enqueue nothing.  `sync` as for deepipr_passport_bn_fwd. */
int deepipr_bn_dual_tail_supported(int N, int C, int HW, int have_sync);
int deepipr_bn_dual_tail_fwd(const float *xa, const float *xb, const float *gamma_a, const float *beta_a,
                             const float *gamma_b, const float *beta_b, float *running_mean_a, float *running_var_a,
                             long long *num_batches_tracked_a, float *running_mean_b, float *running_var_b,
                             long long *num_batches_tracked_b, float momentum_a, float momentum_b, float eps_a,
                             float eps_b, int relu_a, int relu_b, int N, int C, int HW, float *out, float *table_a,
                             float *table_b, unsigned int *sync, void *stream);
int deepipr_bn_dual_tail_bwd(const float *dy, const float *dy2, const float *out, const float *xa, const float *xb,
                             const float *table_a, const float *table_b, float *dxa, float *dxb, float *dgamma_a,
                             float *dbeta_a, float *dgamma_b, float *dbeta_b, int relu_a, int relu_b, int N, int C,
                             int HW, unsigned int *sync, void *stream);
size_t deepipr_passport_bn_workspace_bytes(int N, int C, int HW);
int deepipr_passport_bn_fwd(const float *x, const float *W, const double *m, const float *gamma_in,
                            const float *beta_in, const float *b, float alpha, float margin, float l2,
                            float *running_mean, float *running_var, long long *num_batches_tracked,
                            float momentum, float eps, int training, int N, int C, int HW, int K, int relu,
                            float *y, float *table, float *gamma, float *beta, float *loss, float *acc,
                            int8_t *bits, const float *residual, void *workspace, unsigned int *sync,
                            void *stream);
int deepipr_passport_bn_bwd(const float *dy, const float *x, const float *table, const double *m, const float *b,
                            float alpha, float margin, float l2, const float *dloss, const float *dgamma_extra,
                            const float *dbeta_extra, int training, int N, int C, int HW, int K, int relu,
                            float *dx, float *dW, float *dgamma, float *dbeta, float *table_out, void *workspace,
                            unsigned int *sync, const float *dy2, const float *tail_out, float *dres,
                            void *stream);

/* ------------------------------------------------------------------ GroupNorm / InstanceNorm-fused passport layer
 * The passport layer's other norms are GroupNorm(o // 16, o, affine=False) and InstanceNorm2d(o)
 * (passportconv2d.py:59-62; InstanceNorm = one group per channel).  Their statistics live on one (sample, group)
 * chunk of cpg = C / groups adjacent channels, so norm + passport affine + ReLU is a single register-resident
 * kernel per direction with no cross-workgroup dependency: forward reads x and writes y (8 B/element), backward
 * reads dy and x and writes dx (12 B/element) plus per-sample partial sums of dgamma / dbeta that the finish
 * launch reduces over the batch in fixed order (and turns into dW for layers with W).
 *   forward : [gamma/beta GEMV when W]  ->  k_gn_fwd  ->  [sign loss]
 *   backward: k_gn_bwd  ->  finish (dgamma, dbeta[, dW])
 * replaces: self.bn(x) for norm_type 'gn' / 'in' (native_group_norm / instance_norm and their backward) and
 *           everything deepipr_passport_fwd / _bwd replace.
 * stats: [N * groups][2] floats {mean, invstd}, the only state backward needs besides x, gamma, beta.
 * gamma_in / beta_in (W == NULL): learnable scale / bias of the public branch, or the norm's own affine weights
 * of a plain ConvBlock; both may be NULL (= 1 and 0: InstanceNorm2d without affine).  Backward takes the gamma /
 * beta actually used (NULL likewise).  Needs HW % 4 == 0 and cpg * HW <= 24576 floats: ask
 * deepipr_passport_gn_supported() first; the entry points return DEEPIPR_EUNSUPPORTED without enqueuing anything
 * otherwise.  workspace (backward): deepipr_passport_gn_workspace_bytes(N, C, HW) bytes. */
int deepipr_passport_gn_supported(int N, int C, int HW, int groups);
size_t deepipr_passport_gn_workspace_bytes(int N, int C, int HW);
int deepipr_passport_gn_fwd(const float *x, const float *W, const double *m, const float *gamma_in,
                            const float *beta_in, const float *b, float alpha, float margin, float l2, int groups,
                            float eps, int N, int C, int HW, int K, int relu, float *y, float *stats, float *gamma,
                            float *beta, float *loss, float *acc, int8_t *bits, void *stream);
int deepipr_passport_gn_bwd(const float *dy, const float *x, const float *stats, const float *gamma,
                            const float *beta, const double *m, const float *b, float alpha, float margin, float l2,
                            const float *dloss, const float *dgamma_extra, const float *dbeta_extra, int groups,
                            int N, int C, int HW, int K, int relu, float *dx, float *dW, float *dgamma, float *dbeta,
                            void *workspace, void *stream);

/* ------------------------------------------------------------------ optimiser step on flat buffers
 * SGD with momentum and weight decay, torch.optim.SGD semantics (dampening 0, no Nesterov):
 *     g = grad_scale*grad + weight_decay*param;  buf = momentum*buf + g;  param -= lr*buf
 * over ONE flat buffer of n floats holding every parameter (20 B of traffic per parameter, one launch,
 * instead of a multi-tensor loop over ~60 tensors).  grad_scale = 1/world_size folds the averaging of a summed
 * all-reduce into the pass.  buf must be zero before the first step (torch then sets buf = g, the same value).
 * replaces: optimizer.step() of experiments/trainer.py:145 with optim.SGD(lr, momentum=0.9,
 *           weight_decay=1e-4) from experiments/classification.py:47-50. */
int deepipr_sgd_momentum_step(float *param, const float *grad, float *momentum_buf, size_t n, float lr,
                              float momentum, float weight_decay, float grad_scale, void *stream);
/* The same with the four hyper-parameters read from DEVICE memory: hyper = {lr, momentum, weight_decay,
 * grad_scale}.  Launch arguments are frozen when a step is captured into a hipGraph; with this form a replayed
 * step follows a learning-rate schedule (MultiStepLR of experiments/classification.py:52-56, lr_configs/<name>.json):
 * the host rewrites the floats between replays. */
int deepipr_sgd_momentum_step_dev(float *param, const float *grad, float *momentum_buf, size_t n,
                                  const float *hyper, void *stream);
/* The same update with the gradients read where autograd left them -- no packing pass into a flat gradient buffer.
 * `table` (device): `entries` rows of three 64-bit integers {address of a gradient chunk, offset (in floats) of the
 * matching chunk in param / momentum_buf, element count <= deepipr_sgd_momentum_chunk()}; one workgroup per row.
 * Meant for a step captured into a hipGraph on one GPU (the gradient addresses are then stable and the table is built
 * once); with a gradient all-reduce the gradients have to be contiguous anyway and deepipr_sgd_momentum_step_dev is
 * used.  total_elements is only used for the byte accounting of deepipr_profile_read_bytes. */
int deepipr_sgd_momentum_chunk(void);
int deepipr_sgd_momentum_step_multi(float *param, float *momentum_buf, const long long *table, int entries,
                                    size_t total_elements, const float *hyper, void *stream);

/* ------------------------------------------------------------------ head of the train step
 * loss = mean_n( logsumexp(logits[n]) - logits[n][target[n]] ),  top1_pct = 100 * mean_n( argmax(logits[n]) ==
 * target[n] ) (ties: the lowest class index), lse[n] = logsumexp(logits[n]) saved for backward; two launches (a
 * wavefront per row, then a fixed-order sum over the rows).  workspace: deepipr_ce_top1_workspace_bytes(N) bytes.
 * dlogits[n][c] = dloss/N * (exp(logits[n][c] - lse[n]) - [c == target[n]]); `dloss` is a device scalar.
 * replaces: F.cross_entropy(pred, target) and accuracy(pred, target)[0] of experiments/trainer.py:136,149
 *           (trainer_private.py:161-166) -- log_softmax, nll_loss, topk, eq, sum, mul_ and their backward.
 * target: int64 class indices in [0, C); no ignore_index.  A label outside the range (ATen would raise a device
 * assert; F.cross_entropy's default ignore_index = -100 lands here too) is never dereferenced: its row's loss term
 * is NaN -- so the mean loss is NaN -- and its dlogits row is NaN: the failure cannot pass silently.  Up to 2^20 logits
 * (deepipr_ce_top1_supported): larger problems return DEEPIPR_EUNSUPPORTED without enqueuing anything.
 * logits, dlogits [N][C] f32; loss, top1_pct: one float each; lse [N]. */
int deepipr_ce_top1_supported(int N, int C);
size_t deepipr_ce_top1_workspace_bytes(int N);
int deepipr_ce_top1_fwd(const float *logits, const long long *target, int N, int C, float *loss, float *top1_pct,
                        float *lse, void *workspace, void *stream);
int deepipr_ce_bwd(const float *dloss, const float *logits, const long long *target, const float *lse, int N, int C,
                   float *dlogits, void *stream);

/* ------------------------------------------------------------------ the classifier of the CIFAR-geometry nets (ABI v12)
 * logits[n][k] = b[k] + sum_c W[k][c] * mean_{hw} x[n][c][hw]        -- avg-pool to 1x1, flatten, nn.Linear -- in one launch,
 * and its backward in one: dx[n][c][hw] = (sum_k dlogits[n][k] W[k][c]) / HW, dW[k][c] = sum_n dlogits[n][k] pooled[n][c],
 * db[k] = sum_n dlogits[n][k].  `pooled` [N][C] is written by the forward and read by the backward.  fp32 sums in a fixed
 * order: bit-reproducible.  b / db may be null (a Linear without bias).
 * Supported (deepipr_pooled_linear_supported): C a multiple of 64 up to 4096, HW a multiple of 4 up to 256, K <= 128 classes;
 * anything else (the ImageNet heads: 1000 classes, 7x7 maps) returns DEEPIPR_EUNSUPPORTED without enqueuing anything.
 * replaces: F.avg_pool2d(out, 4) + view + self.linear(out) and their backward (models/resnet_passport.py:127-129 of the
 *           reference): a mean reduce, three BLAS GEMMs, a broadcast-divide and a bias reduce -- 65 us of launch latency around 4 MB. */
int deepipr_pooled_linear_supported(int N, int C, int HW, int K);
int deepipr_pooled_linear_fwd(const float *x, const float *W, const float *b, float *pooled, float *logits, int N, int C, int HW,
                              int K, void *stream);
int deepipr_pooled_linear_bwd(const float *dlogits, const float *W, const float *pooled, float *dx, float *dW, float *db, int N,
                              int C, int HW, int K, void *stream);

/* The scalar bookkeeping of a train step in one launch (ABI v10): out[0] = terms[0] + ... + terms[n_a-1], out[1] = the sum of the
 * next n_b terms (both left to right -- the chain of one-element aten::add launches it replaces, bit for bit), out[2] = out[0] +
 * out[1].  terms: HOST array of n_a + n_b (<= 48) device pointers to single floats (copied into the kernel arguments).
 * replaces: `loss = ce (+ ce_private)`, `sign_loss += m.loss` over the SignLoss modules, `(loss + sign_loss).backward()`
 *           (experiments/trainer.py:140-145, experiments/trainer_private.py:163-173). */
int deepipr_scalar_sums(const float *const *terms, int n_a, int n_b, float *out, void *stream);

/* ------------------------------------------------------------------ residual tail of a block
 * out = relu(a + b) in one pass (12 B/element), and its backward d = dy * [out > 0] (the same gradient goes
 * to both inputs).  All pointers 16-byte aligned, n floats.
 * replaces: `out = out + shortcut; out = F.relu(out)`, models/resnet_passport.py:77-84 (private twin :79-86). */
int deepipr_add_relu_fwd(const float *a, const float *b, float *out, size_t n, void *stream);
int deepipr_relu_bwd(const float *dy, const float *out, float *dx, size_t n, void *stream);
/* The same with the incoming gradient in two pieces, d = (dy + dy2) * [out > 0] (16 B/element): `out` feeds two
 * consumers (the next block's first conv and its shortcut), whose gradients autograd would otherwise add in a
 * separate kernel (12 B/element more).  dy2 == NULL is deepipr_relu_bwd. */
int deepipr_relu_bwd2(const float *dy, const float *dy2, const float *out, float *dx, size_t n, void *stream);

/* ------------------------------------------------------------------ 3x3 stride-2 pad-1 max-pool (the ImageNet stem)
 * y[plane][oh][ow] = max over x[plane][2 oh - 1 .. 2 oh + 1][2 ow - 1 .. 2 ow + 1] inside the map, OH = (H - 1) / 2 + 1 (OW alike);
 * slot[plane][oh][ow] = 3 a + b of the maximum's window position (first maximum in scan order, NaN wins -- ATen's rule); the
 * backward adds dy into dx at the recorded positions, windows in ATen's order (bit-identical dx), no atomics.  planes = N * C;
 * y / dy: planes * OH * OW floats; slot: as many BYTES (the caller's buffer; ATen keeps int64 indices).
 * replaces: nn.MaxPool2d(3, 2, 1) and its backward, models/resnet_passport.py:94-98 (the 224 x 224 stem). */
int deepipr_maxpool3x3s2_fwd(const float *x, float *y, unsigned char *slot, size_t planes, int H, int W, void *stream);
int deepipr_maxpool3x3s2_bwd(const float *dy, const unsigned char *slot, float *dx, size_t planes, int H, int W, void *stream);
/* 2x2 stride-2 pad-0 max-pool (the CIFAR AlexNet's nn.MaxPool2d(2, 2), models/alexnet_passport.py:30-38; ABI v12): the same
 * contract with window slots 0 .. 3; even H, W a multiple of 4 -- anything else returns DEEPIPR_EUNSUPPORTED (keep the library's).
 * x, dx [planes][H][W]; y, dy, slot [planes][H/2][W/2].  Bit-identical to ATen's forward values and backward. */
int deepipr_maxpool2x2s2_fwd(const float *x, float *y, unsigned char *slot, size_t planes, int H, int W, void *stream);
int deepipr_maxpool2x2s2_bwd(const float *dy, const unsigned char *slot, float *dx, size_t planes, int H, int W, void *stream);

/* ------------------------------------------------------------------ 1x1 stride-2 convolution = pixel gather + stride-1 GEMM
 * deepipr_subsample2: y[plane][oh][ow] = x[plane][2 oh][2 ow] (H, W even; planes = N * C).  deepipr_upsample2_zero: its adjoint,
 * dx[plane][2 oh][2 ow] = dy[plane][oh][ow], zero elsewhere (every element of dx is written: no separate fill).  With them a
 * 1x1 stride-2 pad-0 convolution is conv1x1(subsample2(x)) in all three directions -- forward a plain GEMM, weight gradient
 * deepipr_conv_wgrad(.., k = 1, stride 1) on the gathered input, backward-data the transposed GEMM followed by the scatter.
 * replaces: the NHWC implicit-GEMM solvers + layout transposes + zero fill the vendor library runs for the projection shortcuts,
 *           models/resnet_normal.py:41-42 / models/resnet_passport.py:33-36 at ImageNet map widths. */
int deepipr_subsample2(const float *x, float *y, size_t planes, int H, int W, void *stream);
int deepipr_upsample2_zero(const float *dy, float *dx, size_t planes, int H, int W, void *stream);

/* ------------------------------------------------------------------ data convolution: weight gradient
 * dW[co][ci][r][s] = sum_{n,oh,ow} dy[n][co][oh][ow] * x[n][ci][oh*stride + r - pad][ow*stride + s - pad]
 * on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation in a fixed order --
 * bit-reproducible, no atomics), straight from the NCHW tensors: no layout conversion of x / dy / dW, no zero fill.
 * Split K over workgroups leaves partial tiles in `workspace`; a second launch sums them in split order and writes dW
 * (OIHW).  With dgamma / dbeta / m (all three or none) that launch also adds the passport branch's term,
 * dW[co][k] += dgamma[co] * m[0][k] + dbeta[co] * m[1][k]  (deepipr_gamma_beta_bwd_acc's arithmetic), so the shared
 * weight's three-way gradient is complete when it is first written.
 * Supported (H, W: the INPUT map; dy is [N][Co][H / stride][W / stride]; Ci, Co multiples of 64 unless said otherwise):
 *   3x3 pad 1 stride 1   with the default algorithm (deepipr_conv_set_algo(1): Winograd F(3x3, 2x2), fp32 arithmetic) maps 4 / 8 /
 *                        16 / 32 and 14 / 28 / 56 wide of any even height (whole tile-row bands: H / 2 a multiple of 2 / 4 / 2 / 1
 *                        on the first four) and 7 x 7 maps, Ci a multiple of 32, any N (ragged image groups are masked), the
 *                        rank-2 term included (deepipr_conv_wgrad_workspace_bytes is the authority);
 *                        with the direct algorithm (deepipr_conv_set_algo(0)) or bf16x3 arithmetic: output maps 4 / 8 / 16 / 32
 *                        wide, H a multiple of the row band 4 / 8 / 4 / 2, N even on 4-wide maps;
 *   3x3 pad 1 stride 2   output maps 4 / 8 / 16 wide, output height a multiple of 4 / 4 / 2;
 *   1x1 pad 0 stride 1   planes of H * W = a multiple of 64, of 56, of 28, or exactly 49 positions (the Bottleneck's
 *                        convolutions at 56 / 28 / 14 / 7-wide maps, models/resnet_normal.py:30-49; any N; ABI v11);
 *   1x1 pad 0 stride 2   output maps 4 / 8 / 16 wide, output height a multiple of 4 / 8 / 4, N even on 4-wide maps
 *                        (the projection shortcuts; models/resnet_passport.py:33-36);
 *   3x3 pad 1 stride 1 with Ci = 3 on 32-wide maps, H a multiple of 4 (the CIFAR stem): NO fused rank-2 term -- with dgamma /
 *                        dbeta / m the call returns DEEPIPR_EUNSUPPORTED (add it with deepipr_gamma_beta_bwd_acc).
 *   7x7 pad 3 stride 2 with Ci = 3, Co = 64 on 224-wide images, H even (the ImageNet stem, models/resnet_passport.py:94-98;
 *                        deepipr_conv_stem7.inc): no fused rank-2 term either; replaces igemm_wrw_gtcx35_nhwc + its transposes.
 * Anything else: deepipr_conv_wgrad_workspace_bytes returns 0 and deepipr_conv_wgrad returns DEEPIPR_EUNSUPPORTED without
 * enqueuing anything -- the caller keeps the library's wgrad.
 * replaces: the weight half of aten::convolution_backward behind `self.conv(x)`,
 *           models/layers/passportconv2d.py:218 (private twin :215), models/layers/conv2d.py:31 -- MIOpen's
 *           igemm_wrw_gtcx35_nhwc + batched_transpose_* + SubTensorOpWithScalar1d (profiles/r03_steady_state.md).
 * x [N][Ci][H][W]   dy [N][Co][H/stride][W/stride]   dW [Co][Ci][k][k]   m [2][Ci*k*k] double   all pointers 16-byte aligned
 *
 * Arithmetic (ABI v8).  Default: fp32 in, fp32 MFMA, fp32 out, as described above.  Opt-in -- deepipr_conv_set_arith(1),
 * or DEEPIPR_CONV_ARITH=bf16x3 in the environment at load time -- the 3x3 stride-1 instances on maps 8 / 16 / 32 wide run
 * on the bf16 matrix cores with every fp32 operand split EXACTLY into three bf16 words (x = h + m + l) and six of the nine
 * cross products accumulated in fp32 ("bf16x3": v_mfma_f32_32x32x16_bf16 multiplies at 16x the rate of the fp32 MFMA; the
 * three products left out are below 2^-24 |x y|, so a product carries the error of ONE fp32 rounding; still a fixed
 * summation order, still bit-reproducible; one-hot inputs still give exact results; operands below about 2^-110 lose the
 * third word to bf16's subnormal range and are then carried with 16 significant bits).  Measured against a float64
 * convolution the result is as accurate as the fp32-MFMA kernel's and closer than the vendor library's fp32 solvers
 * (tests/test_conv_wgrad_gpu.py, profiles/r04_wgrad_bench_bf16x3.json); it is opt-in because it is not the fp32
 * multiply the reference's `self.conv(x)` names.  deepipr_conv_get_arith reports the mode.  The setting is process-wide
 * and read when a call is planned: do not change it between a workspace query and its call. */
int deepipr_conv_set_arith(int mode);          /* 0: fp32 MFMA, 1: bf16x3 */
int deepipr_conv_get_arith(void);
size_t deepipr_conv_wgrad_workspace_bytes(int N, int Ci, int Co, int H, int W, int kh, int kw, int stride, int pad);
int deepipr_conv_wgrad(const float *x, const float *dy, float *dW, int N, int Ci, int Co, int H, int W, int kh, int kw,
                       int stride, int pad, const float *dgamma, const float *dbeta, const double *m, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ data convolution: forward / backward-data
 * y[n][co][oh][ow] = sum_{ci,r,s} w[co][ci][r][s] * x[n][ci][oh*stride + r - pad][ow*stride + s - pad]      (conv_fwd)
 * dx[n][ci][ih][iw] = sum_{co,r,s} w[co][ci][r][s] * dy[n][co][(ih + pad - r) / stride][(iw + pad - s) / stride]
 *                                                   over the (r, s) for which the divisions are exact      (conv_dgrad)
 * as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), NCHW in and out: no layout conversion, no
 * workspace, bit-reproducible (fixed summation order).  Backward-data gathers the transposed (stride 1: and flipped)
 * weights while it stages them -- there is no transposed copy of w -- and runs a stride-2 convolution as its four output
 * parity classes in one launch (no multiplication by inserted zeros; the 1x1 case writes the zero pixels itself).
 * Supported (deepipr_conv_supported; direction 0 = forward, 1 = backward-data): Ci, Co multiples of 64;
 *   3x3 pad 1 stride 1 on output maps 4 / 8 / 16 / 32 wide (both directions),
 *   3x3 pad 1 stride 2 and 1x1 pad 0 stride 2 on output maps 4 / 8 / 16 wide (both directions; 3x3 backward-data also 32);
 *   whole row bands: the output height a multiple of 8 / 8 / 4 / 2 (32-, 16-, 8-, 4-wide maps), N a multiple of 4 on 4-wide maps;
 *   1x1 pad 0 stride 1 on any map (both directions; ABI v11): one GEMM over the flattened (image, pixel) positions, the
 *   Bottleneck's convolutions (models/resnet_normal.py:30-49); N * H * W * max(Ci, Co) below 2^31.
 *   3x3 pad 1 stride 1, Ci = 3, Co = 64, W = 32, H a multiple of 8 (forward only): the CIFAR stem (models/resnet_passport.py:99-101);
 *   7x7 pad 3 stride 2, Ci = 3, Co = 64, W = 224, H a multiple of 8 (forward only -- the image has no gradient): the ImageNet stem (models/resnet_passport.py:
 *   94-98; replaces miopenSp3AsmConv_*_f3x2_stride2 + its CNHW transposes), deepipr_conv_stem7.inc.
 * Anything else returns DEEPIPR_EUNSUPPORTED without enqueuing anything: the caller keeps the library's convolution.
 * replaces: aten::convolution / the data half of aten::convolution_backward behind `self.conv(x)`,
 *           models/layers/passportconv2d.py:218 (private twin :215), models/layers/conv2d.py:31 -- for the stride-2
 *           and 1x1 convolutions MIOpen's igemm_{fwd,bwd}_gtcx35_nhwc + batched_transpose_* + SubTensorOpWithScalar1d.
 * H, W: the convolution's INPUT map (x) in both calls.   x [N][Ci][H][W]   w [Co][Ci][k][k]   y, dy [N][Co][H/stride][W/stride] */
int deepipr_conv_supported(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction);
int deepipr_conv_fwd(const float *x, const float *w, float *y, int N, int Ci, int Co, int H, int W, int k, int stride, int pad,
                     void *stream);
int deepipr_conv_dgrad(const float *dy, const float *w, float *dx, int N, int Ci, int Co, int H, int W, int k, int stride,
                       int pad, void *stream);
/* The same two with a workspace (ABI v8).  Where the 64 x 64 output tiles alone would leave the chip short of two
 * workgroups per CU -- the 4x4 / 8x8 maps of the deep layers, small batches -- K = Cin * 9 is split over workgroups: each
 * writes a partial output into its slab of the workspace, a second launch adds the slabs in split order (fixed order, no
 * atomics: bit-reproducible like the plain form).  deepipr_conv_workspace_bytes returns what the call needs (0: the
 * plain form runs; stride-2 backward-data never splits); with no or too small a workspace the plain form runs.
 * 16-byte aligned workspace. */
size_t deepipr_conv_workspace_bytes(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction);
/* Algorithm of the 3x3 stride-1 pad-1 convolutions, both directions (ABI v9).  1 (default): Winograd F(2x2, 3x3) around the
 * fp32 MFMA -- the 3x3 filters are transformed (G g G^T) while they are staged, every wavefront transforms its row of the
 * 4x4 input tiles (B^T d B) straight out of the staged NCHW band, sixteen batched 32 x 32 x 2 MFMA GEMMs over channel pairs,
 * the output transform in the epilogue: 16 instead of 36 multiplies per 2x2 outputs and channel, NCHW in and out, fixed
 * summation order (bit-reproducible), exact on small-integer operands; against float64 a few 1e-7 of the output scale (the
 * algorithm the vendor library runs for these layers on the vector ALUs, miopenSp3AsmConv_*_f2x3).  Any N and any H
 * (ragged image groups, ragged row bands and the half-empty last tiles of odd maps are masked), Ci a multiple of 8 and Co
 * of 32 (backward-data: the other way round), maps 4 / 8 / 16 / 32 wide (CIFAR geometry) and 7 / 14 / 28 / 56 wide
 * (ImageNet geometry: 28 of a workgroup's 32 tile columns busy).
 * 0: the direct implicit GEMM above (also what every other shape takes).  DEEPIPR_CONV_ALGO=direct|winograd in the
 * environment at load time sets the default.  Process-wide, read when a call is planned: do not change it between
 * deepipr_conv_supported / _workspace_bytes and the call.  deepipr_conv_algo_of: the algorithm a call of this shape takes. */
int deepipr_conv_set_algo(int algo);
int deepipr_conv_get_algo(void);
int deepipr_conv_algo_of(int N, int Ci, int Co, int H, int W, int k, int stride, int pad, int direction);
int deepipr_conv_fwd_ws(const float *x, const float *w, float *y, int N, int Ci, int Co, int H, int W, int k, int stride, int pad,
                        void *workspace, size_t workspace_bytes, void *stream);
int deepipr_conv_dgrad_ws(const float *dy, const float *w, float *dx, int N, int Ci, int Co, int H, int W, int k, int stride,
                          int pad, void *workspace, size_t workspace_bytes, void *stream);

/* Pre-transformed form of the Winograd forward / backward-data convolution (ABI v10).  In deepipr_conv_fwd_ws / _dgrad_ws every
 * workgroup transforms the filters it stages (G g G^T: the same arithmetic 1 000 times per launch of a 64-channel layer at
 * batch 128, and the costliest part of the main loop beside the MFMAs).  Here the transform runs ONCE per step:
 *   deepipr_conv_wino_transform_multi  writes, for up to deepipr_conv_wino_max_layers() weights [Co][Ci][3][3] per launch, the
 *       image Uf (forward: rows = co) and / or Ud (backward-data: rows = ci, filters rotated by 180 degrees) -- each
 *       deepipr_conv_wino_image_bytes(Co, Ci) bytes (66 B per filter; 0: Co or Ci is not a multiple of 32, no such form), laid
 *       out as the kernels' LDS stage: [chunk of 8 input channels][row / 32][32 rows][U[row][xi][c][nu], 128 floats + 4 of pitch];
 *   deepipr_conv_fwd_pre / _dgrad_pre  are deepipr_conv_fwd_ws / _dgrad_ws of a 3x3 stride-1 pad-1 convolution with the image
 *       in place of w: a chunk's rows go global -> LDS by global_load_lds_dwordx4 (no registers, no vector instructions, no LDS
 *       stores for the weights).  Same shapes, same workspace (deepipr_conv_workspace_bytes), same planner; results BIT-IDENTICAL
 *       to the _ws entry points (one definition of the transform's arithmetic).  DEEPIPR_EUNSUPPORTED where the call of this shape
 *       would not take the Winograd kernel (deepipr_conv_algo_of) or the weight has no image.
 * The images belong to the caller and must be rewritten whenever the weights change (the train step: once, before the
 * forward pass; the backward pass of the same step reads Ud).
 * replaces: nothing of the reference's own -- `self.conv(x)` (models/layers/passportconv2d.py:218, models/layers/conv2d.py:31)
 *           is one ATen call there; this is the same convolution with the filter transform hoisted out of the launch. */
#define DEEPIPR_WINO_MAX_LAYERS 24
typedef struct DeepiprWinoLayer {
    const float *W;          /* [Co][Ci][3][3] */
    float *Uf, *Ud;          /* images out; either may be NULL */
    int Co, Ci;
} DeepiprWinoLayer;
size_t deepipr_conv_wino_image_bytes(int Co, int Ci);
int deepipr_conv_wino_max_layers(void);
int deepipr_conv_wino_transform_multi(const DeepiprWinoLayer *layers, int n, void *stream);
int deepipr_conv_fwd_pre(const float *x, const float *image, float *y, int N, int Ci, int Co, int H, int W, void *workspace,
                         size_t workspace_bytes, void *stream);
int deepipr_conv_dgrad_pre(const float *dy, const float *image, float *dx, int N, int Ci, int Co, int H, int W, void *workspace,
                           size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPIPR_HIP_H */
