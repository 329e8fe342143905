/* dpft_hip.h -- C-ABI of libdpft_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the DPFT hot path.  Every entry point is a
 * plain C function over raw device pointers + sizes + a hipStream_t; no torch types.  Rules:
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing is allocated inside;
 *   - no hidden synchronisation: every call only enqueues work on `stream` (graph-capturable);
 *   - re-entrant and stream-ordered;
 *   - return 0 on success, a negative code otherwise; dpft_last_error() returns the message of
 *     the last failure on the calling thread (the Python side raises RuntimeError from it).
 *
 * Reference interfaces replaced (paths relative to the TUMFTM/DPFT checkout):
 *   dpft_msda_fwd_f32 / dpft_msda_bwd_f32
 *       MSDA.ms_deform_attn_forward / ms_deform_attn_backward, the only native seam the
 *       reference has: src/dprt/models/layers/ms_deform_attn.py:24,32-39,58-66.
 *   dpft_xattn_fwd_f32 / dpft_xattn_bwd_f32
 *       the flatten+cat -> value_proj -> MSDA core chain of MLFusion.forward_cross_attn /
 *       MSDeformAttn.forward: src/dprt/models/fusers/mpfusion.py:150-208 and
 *       src/dprt/models/layers/ms_deform_attn.py:172-213 ("sample-then-project", reads the
 *       NHWC FPN levels in place).
 *   dpft_conv2d_nhwc_{fwd,dgrad,wgrad}_f32, dpft_bn_*, dpft_maxpool_*, dpft_bn_add_relu_*
 *       the cuDNN kernels torch dispatches for the torchvision ResNet body and FPN:
 *       src/dprt/models/backbones/resnet.py:47-55,80-107, src/dprt/models/necks/fpn.py:39-43,70-83.
 *   dpft_fpn_topdown_*, dpft_add_pos_*
 *       F.interpolate(nearest)+add inside torchvision FPN, and the in-place sinusoidal
 *       embedding src/dprt/models/embeddings/sinusoidal.py:63-110.
 *   dpft_giou3d_yaw_f32
 *       pytorch3d.ops.box3d_overlap as used by src/dprt/utils/iou.py:121-210.
 */
#ifndef DPFT_HIP_H
#define DPFT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPFT_OK 0
#define DPFT_ERR_ARG (-1)
#define DPFT_ERR_LAUNCH (-2)
#define DPFT_ERR_UNSUPPORTED (-3)

typedef void* dpft_stream_t; /* hipStream_t */

int dpft_version(void);
const char* dpft_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family: NHWC activations, weights physically [Cout][kh][kw][Cin]
 * (= torch OIHW tensor in channels_last memory format), fp32, MFMA implicit GEMM.
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_conv_desc {
    int32_t B, H, W, C;      /* input  (B,H,W,C)                      */
    int32_t K;               /* output channels                       */
    int32_t kh, kw, stride, pad;
    int32_t OH, OW;          /* output spatial size                   */
    int32_t act16;           /* 0: activations (x, y, dy, dx, residual sources) are fp32 tensors;
                              * 1: they are bf16 tensors ("bf16 mixed precision" storage of BASELINE.json configs[4]:
                              *    4 channels = 8 bytes per loader lane, BatchNorm statistics still from the fp32
                              *    accumulators); weights, weight gradients, BN blocks and statistics stay fp32.
                              *    Needs C % 64 == 0 and K % 64 == 0; the pointers are still declared const float*.  */
    /* Round 5: the GEMM operands as THREE bf16 PLANES (dpft_split_planes_f32: an fp32 value is exactly the sum of three bf16
     * terms; plane p of element e at base + 2 (p E + e) bytes, E = the tensor's element count).  When BOTH are given and the
     * problem takes the C % 64 == 0 path without an operand prologue, the forward / data-gradient GEMM reads its operands
     * from them and multiplies on the bf16 matrix cores (six term products, fp32 accumulation: fp32-grade results at a
     * multiple of the fp32 MFMA rate, dpft_amd/csrc/conv_x3.hip).  a_planes = planes of the A operand (x of the forward,
     * dy of the data gradient), w_planes = planes of the weight tensor passed as w / w_t.  The fp32 tensors are still
     * required (fallback paths, epilogues).  NULL = fp32 operands (zero-initialise the descriptor). */
    const void* a_planes;
    const void* w_planes;
} dpft_conv_desc;

/* bytes of workspace dpft_conv2d_* may need for this problem: a ticket header + split-K partial slabs.
 * Contract (round 4): the first dpft_conv2d_workspace_header_bytes() bytes of a workspace buffer must be ZERO when the
 * buffer is first handed to the library (dpft_conv2d_workspace_init zero-fills them on `stream`); every call leaves
 * them zero again, so one initialisation per buffer is enough.  The header holds one ticket per output tile of a
 * split-K launch: the workgroup that draws a tile's last ticket sums the partial tiles in split order and runs the
 * whole epilogue -- there is no separate reduction launch.  Calls that share a workspace must be stream-ordered. */
int64_t dpft_conv2d_workspace_bytes(const dpft_conv_desc* d);
int64_t dpft_conv2d_workspace_header_bytes(void);
int dpft_conv2d_workspace_init(void* workspace, dpft_stream_t stream);
/* fp32 tensor of n elements (n % 4 == 0) -> three bf16 planes [3][n]: plane 0 = RNE bf16 of the value, plane 1 = RNE bf16 of
 * what plane 0 left, plane 2 = the (exact) rest.  src = plane0 + plane1 + plane2 exactly. */
int dpft_split_planes_f32(const float* src, void* planes, int64_t n, dpft_stream_t stream);
/* number of M-tiles the forward kernel uses for its fused BN-statistics epilogue
 * (rows of the `stats` buffer: stats is [mtiles][2][K] floats = per-tile mean and M2);
 * *tile_rows receives the tile height so the caller can recover per-tile counts. */
int32_t dpft_conv2d_stats_tiles(const dpft_conv_desc* d, int32_t* tile_rows);
/* the same for a launch that carries the BatchNorm + ReLU operand prologue (pro_bn != NULL in dpft_conv2d_nhwc_fwd_f32): with bf16
 * operands (act16 = 2) such a launch takes other tiles than the prologue-free one */
int32_t dpft_conv2d_stats_tiles_pro(const dpft_conv_desc* d, int32_t pro, int32_t* tile_rows);

/* Arithmetic of the forward / data-gradient GEMMs of every conv with C % 64 == 0 and of the 128 x 128-tiled weight
 * gradients (process-wide; call between launches):
 *   0  fp32 operands, fp32 accumulation -- the reference's arithmetic (config/kradar.json "dtype": "float32"), default;
 *   1  operands rounded to bf16 (RNE) on their way into LDS, fp32 accumulation (v_mfma_f32_32x32x16_bf16); tensors in
 *      memory, BatchNorm, the small weight gradients, optimizer stay fp32 ("bf16 mixed precision", BASELINE.json configs[4]).
 *   2  (experimental, forward / data gradient only) every operand value split into three bf16 terms on its way into
 *      LDS and the six term products of weight >= 2^-16 accumulated in fp32 on the bf16 matrix cores: fp32-grade
 *      results (measured 2-3e-7 relative L2 error vs fp64, the fp32 MFMA path 6-8e-7), 10-15 % faster than mode 0.
 * Returns DPFT_ERR_ARG for any other value. */
int dpft_conv_set_compute(int32_t mode);
int32_t dpft_conv_get_compute(void);
/* fp32 compute mode only: run the big multi-tap GEMMs (3x3 / 7x7 filters, C % 64 == 0, >= 2 GFLOP) as 3 x bf16 split products
 * on the bf16 matrix cores -- fp32 tensors in, fp32 results out, six exact term products per fp32 product accumulated in
 * fp32 (dpft_amd/csrc/conv_x3.hip; error vs fp64 below the fp32 MFMA path's).  Default on (DPFT_CONV_SPLIT=0 presets off);
 * off = v_mfma_f32_32x32x2_f32 everywhere.  Set before problems are sized: dpft_conv2d_stats_tiles depends on it. */
int dpft_conv_set_split(int32_t on);
int32_t dpft_conv_get_split(void);

/* BN parameter block convention used across the library: bnp[4][K] floats =
 *   row 0 mean, row 1 scale = gamma*invstd, row 2 beta, row 3 invstd;   bn(v) = (v - mean)*scale + beta
 * (mean is subtracted first: no beta - mean*scale cancellation when |mean| >> std). */

/* y[B,OH,OW,K] = conv(act(x), w) (+bias).  Optional fused prologue on the input operand:
 * act(x) = [max(., 0)] bn(x) with the PRODUCER's BN block pro_bn[4][C] (NULL = identity; pro_relu
 * selects the max).  Padding stays exactly zero.  Optional fused epilogue: per-M-tile per-channel
 * (mean, M2) of y into `stats` for train-mode BatchNorm (NULL to skip). */
int dpft_conv2d_nhwc_fwd_f32(const dpft_conv_desc* d, const float* x, const float* w,
                             const float* bias, const float* pro_bn, int32_t pro_relu, float* y,
                             float* stats, void* workspace, dpft_stream_t stream);
/* Inference form of conv + BatchNorm (+ residual add) (+ ReLU) -- what torchvision's Bottleneck.forward computes in eval
 * mode per conv (resnet.py Bottleneck: conv -> bn -> relu, and conv3 -> bn3 -> += identity -> relu; reached through
 * src/dprt/models/backbones/resnet.py:80-107):  y = [relu](bn(conv(x, w)) [+ residual]),  out_bn = BN block [4][K] of the
 * output channels (dpft_bn_eval_params_f32).  One launch where the problem takes the C % 64 == 0 path without split-K
 * (applied in the GEMM epilogue), otherwise the convolution followed by dpft_bn_act_f32 in place: same arithmetic. */
int dpft_conv2d_nhwc_fwd_bnact_f32(const dpft_conv_desc* desc, const float* x, const float* w, const float* out_bn,
                                   int32_t relu, const float* residual, float* y, void* workspace, dpft_stream_t stream);

/* dx[B,H,W,C] (+)= conv_transpose(dy, w).  w_t is the transposed weight [C][kh][kw][K]
 * (see dpft_weight_transpose_f32); accumulate != 0 adds into dx instead of overwriting it. */
int dpft_conv2d_nhwc_dgrad_f32(const dpft_conv_desc* d, const float* dy, const float* w_t,
                               float* dx, int32_t accumulate, void* workspace,
                               dpft_stream_t stream);

/* Data gradient with the FIRST PASS of a BatchNorm backward in its epilogue -- the form the ResNet launch plan
 * (dpft_resnet_backward_stage) runs: the tensor this call writes, dx (B,H,W,C), is the `dout` of a BatchNorm layer
 * whose input bn_y has the same shape and whose BN block (mean, gamma*invstd, beta, invstd) is bn_block [4][C];
 * with d = dx under that layer's ReLU mask (bn_mask8: one byte per 4 channels, bit e = element e passed; or
 * bn_self_mask: mask = bn(bn_y) > 0) the launch adds  sums[0][c] += sum_pixels d,  sums[1][c] += sum_pixels d * xhat,
 * xhat = (bn_y - mean) * invstd  -- what dpft_bn_bwd_reduce_f32 computes in a pass of its own.  sums [2][C] must be
 * zero (or hold a running total).  *applied = 1 when the launch carried the reduction; 0 when the problem takes a path
 * that cannot (split-K, thin channels, a strided 1x1 whose parity classes leave pixels untouched): sums is then
 * untouched and the caller runs the separate pass.
 * res_src != NULL: the bottleneck's identity branch folded in as well, dx = dgrad(dy) + (mask ? res_src : 0) with the
 * mask from res_mask8 (bytes) or res_out > 0 (the block output; always required for the split-K form); stride 1 only. */
int dpft_conv2d_nhwc_dgrad_bn_reduce_f32(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx,
                                         int32_t accumulate, const float* res_src, const float* res_out,
                                         const uint8_t* res_mask8, const float* bn_y, const float* bn_block,
                                         const uint8_t* bn_mask8, int32_t bn_self_mask, float* sums,
                                         int32_t* applied, void* workspace, dpft_stream_t stream);
/* dw[K][kh][kw][C] = sum_pixels dy (x) act(x); same optional prologue on x as forward.
 * dw is overwritten. */
int dpft_conv2d_nhwc_wgrad_f32(const dpft_conv_desc* d, const float* x, const float* dy,
                               const float* pro_bn, int32_t pro_relu, float* dw, void* workspace,
                               dpft_stream_t stream);
/* [K][taps][C] -> [C][taps][K] */
int dpft_weight_transpose_f32(const float* w, float* w_t, int32_t K, int32_t taps, int32_t C,
                              dpft_stream_t stream);
/* The same for n (<= 80) weight tensors in ONE launch: w[i] is [K[i]][taps[i]][C[i]], w_t[i] receives [C[i]][taps[i]][K[i]]
 * (an FPN's ten convs per backward, `necks/fpn.py:39-43`; pointer / size arrays live on the host). */
int dpft_weight_transpose_batch_f32(int32_t n, const float* const* w, float* const* w_t, const int32_t* K,
                                    const int32_t* taps, const int32_t* C, dpft_stream_t stream);
/* dw as dpft_conv2d_nhwc_wgrad_f32 (no prologue) AND db = the bias gradient of a conv with bias (`necks/fpn.py:39-43`:
 * torchvision's FPN convs), in one pass over dy where the weight-gradient kernel has dy at hand. */
int dpft_conv2d_nhwc_wgrad_bias_f32(const dpft_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                                    void* workspace, dpft_stream_t stream);
/* db[K] = sum over rows of dy[M][K] */
int dpft_bias_grad_f32(const float* dy, float* db, int64_t M, int32_t K, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (train + eval), ReLU, residual, max-pool on NHWC fp32
 * ---------------------------------------------------------------------------------------- */
/* standalone per-tile statistics of y[M][K] in the same [tiles][2][K] format (tile_rows rows each) */
int dpft_bn_stats_f32(const float* y, float* stats, int64_t M, int32_t K, int32_t tile_rows,
                      dpft_stream_t stream);
/* combine tiles (Chan's parallel variance), write the BN block bnp[4][K], update the running
 * statistics (momentum; unbiased variance) when running_mean != NULL. */
int dpft_bn_finalize_f32(const float* stats, int32_t tiles, int32_t tile_rows, int64_t M, int32_t K,
                         const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, float* bnp, dpft_stream_t stream);
/* eval mode: BN block from the running statistics */
int dpft_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, int32_t K, float* bnp,
                            dpft_stream_t stream);
/* out = [relu]( bn(y)  [+ (bn_res(res) | res)] ) elementwise over [M][K] */
int dpft_bn_act_f32(const float* y, const float* bnp, const float* res, const float* res_bnp,
                    int32_t relu, float* out, int64_t M, int32_t K, dpft_stream_t stream);
/* stem: out[B,PH,PW,K] = maxpool3x3s2p1( relu(bn(y)) ), y is [B,H,W,K] */
int dpft_bn_relu_maxpool_f32(const float* y, const float* bnp, float* out, int32_t B, int32_t H,
                             int32_t W, int32_t K, int32_t PH, int32_t PW, dpft_stream_t stream);
/* backward of the stem pool + ReLU: dz[B,H,W,K] = grad wrt bn(y) = (relu(bn(y))>0) * sum of dout over
 * the pool windows whose first arg-max (scan order, strict >) is this pixel */
int dpft_bn_relu_maxpool_bwd_f32(const float* y, const float* bnp, const float* dout, float* dz,
                                 int32_t B, int32_t H, int32_t W, int32_t K, int32_t PH, int32_t PW,
                                 dpft_stream_t stream);
/* BatchNorm backward, two passes.  dz = dout * mask with
 *   mask = (out > 0)              if out != NULL       (out = post-activation tensor)
 *   mask = (bn_mask(y) > 0)       elif mask_bnp != NULL (recompute the fused ReLU from its BN block)
 *   mask = 1                      otherwise.
 * pass 1: sums[0][k] = sum dz, sums[1][k] = sum dz * xhat   (xhat = (y-mean)*invstd from bnp) */
int dpft_bn_bwd_reduce_f32(const float* y, const float* dout, const float* out,
                           const float* mask_bnp, const float* bnp, float* sums, int64_t M,
                           int32_t K, dpft_stream_t stream);
/* pass 2: dy = gamma*invstd*(dz - sums0/M - xhat*sums1/M); dgamma = sums1, dbeta = sums0 */
int dpft_bn_bwd_apply_f32(const float* y, const float* dout, const float* out,
                          const float* mask_bnp, const float* bnp, const float* gamma,
                          const float* sums, float* dy, float* dgamma, float* dbeta, int64_t M,
                          int32_t K, dpft_stream_t stream);
/* dz = dout * (out > 0)  (residual branch gradient) */
int dpft_relu_bwd_f32(const float* dout, const float* out, float* dz, int64_t n,
                      dpft_stream_t stream);
/* a += b */
int dpft_add_inplace_f32(float* a, const float* b, int64_t n, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Native launch plan of the ResNet body (conv1/bn1/relu/maxpool/layer1..n of torchvision's
 * ResNet-50/101/152 behind IntermediateLayerGetter, src/dprt/models/backbones/resnet.py:54-55,
 * incl. the optional 1x1 adjustment conv :47-52).  One call enqueues a whole forward, or the
 * backward of one stage; everything lives in a caller-owned arena; nothing is allocated or
 * synchronised inside (hipGraph-capturable).
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_resnet_desc {
    int32_t B, H, W, in_channels;   /* NHWC input (B,H,W,in_channels); != 3 adds the adjustment conv */
    int32_t depths[4];              /* bottlenecks per stage, e.g. {3,4,23,3}                       */
    int32_t n_layers;               /* stages to run (multi_scale), 1..4                            */
    float eps, momentum;            /* BatchNorm2d hyper-parameters (1e-5, 0.1)                     */
    int32_t act16;                  /* 1: bf16 activation / gradient storage inside the body (dpft_conv_desc.act16;
                                     * with dpft_conv_set_compute(1) = the mixed precision of BASELINE.json configs[4]);
                                     * input image, stem conv output, stage outputs handed to the caller, all
                                     * parameters, BatchNorm statistics and parameter gradients stay fp32          */
} dpft_resnet_desc;

/* Pointer tables in plan order.  conv: [adjustment], stem conv1, then per bottleneck conv1, conv2,
 * conv3, [downsample.0].  bn: stem bn1, then per bottleneck bn1, bn2, bn3, [downsample.1].
 * Weights are physical [K][kh][kw][C]; *_dw / bn_d* are written (overwritten) by the backward. */
typedef struct dpft_resnet_tables {
    void* const* conv_w;
    void* const* conv_dw;
    void* const* bn_gamma;
    void* const* bn_beta;
    void* const* bn_rm;
    void* const* bn_rv;
    void* const* bn_dgamma;
    void* const* bn_dbeta;
} dpft_resnet_tables;

int64_t dpft_resnet_plan_create(const dpft_resnet_desc* desc);   /* 0 on error */
void dpft_resnet_plan_destroy(int64_t plan);
/* what: 0 arena bytes, 1 #convs, 2 #bns, 3 float offset of stage output idx in the arena,
 *       4 shape entry (idx = stage*4 + dim of (B,H,W,C)), 5 floats kept for backward */
int64_t dpft_resnet_plan_query(int64_t plan, int32_t what, int32_t idx);
/* x (B,H,W,in_channels).  train = 1: batch statistics + running-stat update, activations kept for the backward stages;
 * train = 2: frozen BatchNorm -- activations kept as well, but every BatchNorm normalises with its running statistics
 * (nothing is updated) and the backward stages drop the batch-statistics terms (dy = gamma invstd d): the gradient through
 * an eval-mode body / torchvision's FrozenBatchNorm2d; train = 0: inference (BatchNorm folded into the conv epilogues). */
int dpft_resnet_forward(int64_t plan, const float* x, const dpft_resnet_tables* tables, void* arena,
                        int32_t train, dpft_stream_t stream);
/* Backward of stage `stage` (call n_layers-1 ... 0 after a train forward; stage 0 also runs the
 * stem).  dout = external gradient of that stage's output (NULL = none).  frozen = the BatchNorm mode of the forward this
 * backward belongs to (1: it ran with train = 2, running statistics; 0: train = 1) -- the caller says it, the plan keeps
 * no per-forward state. */
int dpft_resnet_backward_stage(int64_t plan, int32_t stage, const float* x,
                               const dpft_resnet_tables* tables, void* arena, const float* dout,
                               int32_t frozen, dpft_stream_t stream);
/* The stream the plan's weight-gradient GEMMs run on while the data-gradient chain continues on the caller's stream
 * (default: a stream the plan creates on first use).  Passing the caller's own stream keeps everything in order on it. */
int dpft_resnet_plan_set_side_stream(int64_t plan, dpft_stream_t side);
/* on != 0: a train-mode forward and each backward stage are captured into hipGraphs (after two eager calls per argument
 * set) and replayed -- one host-side launch instead of several hundred.  Only calls that are a single-stream sequence are
 * captured (the forward always; a backward stage when the weight-gradient stream IS the launch stream); the caller must
 * keep x / arena / dout / the tables' pointers the same from call to call (a new set is captured anew). */
int dpft_resnet_plan_set_graph(int64_t plan, int32_t on);

/* ------------------------------------------------------------------------------------------
 * Streams on distinct hardware queues (MI355X-side addition; the reference is single-stream).
 * The runtime multiplexes HIP streams onto 4 hardware queues and streams sharing a queue run in order; the step's
 * concurrency (camera chain | its weight gradients | radar encoders) needs one queue each.  out[0..n-1] receive streams
 * (library-owned, never destroyed) that share a queue neither with main_stream nor with each other -- found by probing
 * with a spin kernel; when the device has fewer queues the entries repeat.  Returns the number of distinct queues found
 * or a negative error code.  Synchronises the device: set-up time only.
 * ---------------------------------------------------------------------------------------- */
int32_t dpft_stream_set(dpft_stream_t main_stream, int32_t n, dpft_stream_t* out);

/* ------------------------------------------------------------------------------------------
 * FPN glue + positional embedding
 * ---------------------------------------------------------------------------------------- */
/* lat[B,H,W,K] += nearest_upsample(top[B,TH,TW,K]) with src = min(floor(dst*in/out), in-1) */
int dpft_fpn_topdown_add_f32(float* lat, const float* top, int32_t B, int32_t H, int32_t W,
                             int32_t TH, int32_t TW, int32_t K, dpft_stream_t stream);
/* dtop[B,TH,TW,K] += sum of dlat over the pixels that map to each source pixel */
int dpft_fpn_topdown_add_bwd_f32(const float* dlat, float* dtop, int32_t B, int32_t H, int32_t W,
                                 int32_t TH, int32_t TW, int32_t K, dpft_stream_t stream);
/* x[B,H,W,K] += pos_x[W][K]; x += pos_y[H][K]  (two fp32 adds in the reference's order) */
int dpft_add_pos_f32(float* x, const float* pos_x, const float* pos_y, int32_t B, int32_t H,
                     int32_t W, int32_t K, dpft_stream_t stream);
/* One level of the neck's forward in two launches (src/dprt/models/necks/fpn.py:70-83 -> torchvision FeaturePyramidNetwork.forward;
 * src/dprt/models/embeddings/sinusoidal.py:107-108):
 *   lat = conv1x1(x, w, bias) [+ nearest_upsample(top[B,TH,TW,K])]          `d` = the 1x1 lateral conv's descriptor
 *   out = conv3x3(lat, w, bias) [+ pos_x[W][K]; + pos_y[H][K]]              `d` = the 3x3 output conv's descriptor
 * The adds ride in the convs' epilogues where a thin-channel kernel takes the shape (raw-input laterals C = 3 / 6 -> 16; the
 * 16 -> 16 3x3 convs): one pass over the level instead of two.  Everywhere else the function runs the conv followed by
 * dpft_fpn_topdown_add_f32 / dpft_add_pos_f32 -- the same arithmetic in the same order.  top / pos_x + pos_y may be NULL. */
int dpft_fpn_lateral_f32(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, const float* top,
                         int32_t TH, int32_t TW, float* lat, void* workspace, dpft_stream_t stream);
int dpft_fpn_output_f32(const dpft_conv_desc* d, const float* lat, const float* w, const float* bias, const float* pos_x,
                        const float* pos_y, float* out, void* workspace, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-scale deformable attention
 * ---------------------------------------------------------------------------------------- */
/* Operator-level drop-in for the MSDA extension (same tensor semantics):
 * value (N,S,M,D) f32; shapes (L,2) int64 [H,W]; lsi (L) int64; loc (N,Lq,M,L,P,2) f32 (x,y in
 * [0,1]); attn (N,Lq,M,L,P) f32; out (N,Lq,M*D). */
int dpft_msda_fwd_f32(const float* value, const int64_t* shapes, const int64_t* lsi,
                      const float* loc, const float* attn, float* out, int32_t N, int32_t S,
                      int32_t M, int32_t D, int32_t Lq, int32_t L, int32_t P, dpft_stream_t stream);
/* grad_value must be zero-filled by the caller (atomics accumulate into it). */
int dpft_msda_bwd_f32(const float* value, const int64_t* shapes, const int64_t* lsi,
                      const float* loc, const float* attn, const float* grad_out,
                      float* grad_value, float* grad_loc, float* grad_attn, int32_t N, int32_t S,
                      int32_t M, int32_t D, int32_t Lq, int32_t L, int32_t P, dpft_stream_t stream);

#define DPFT_MAX_LEVELS 8
typedef struct dpft_pyramid {
    const float* level[DPFT_MAX_LEVELS]; /* (B,H_l,W_l,C) NHWC, C = d_model */
    float* grad[DPFT_MAX_LEVELS];        /* same shapes, accumulated into (backward only) */
    int32_t H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int32_t L;
    /* grad_replicas[l] = R > 1: grad[l] is (R,B,H_l,W_l,C) and every query row adds into replica (row % R); the
     * caller sums the replicas.  Spreads the fp32 atomics of tiny levels (a 2x4 radar level would otherwise take
     * ~6000 serialized adds per address).  0 or 1 = plain buffer.  Honoured by dpft_xattn_ffn_train_bwd_f32. */
    int32_t grad_replicas[DPFT_MAX_LEVELS];
} dpft_pyramid;

/* Fused sample-then-project cross attention (C = M*D <= 64, L*P <= 32):
 *   samp[b,q,m,:C]  = sum_{l,p} attn * bilinear(level_l, ref + off/(W_l,H_l))     (raw features)
 *   mass[b,q,m]     = sum_{l,p} attn * (in-bounds bilinear weight mass)
 *   out[b,q,m*D+d]  = Wv[m*D+d,:] . samp[b,q,m,:] + bv[m*D+d] * mass[b,q,m]
 * which equals value_proj -> MSDA core of the reference (SURVEY App. C fusion identity).
 * ref (B,Q,2) (u,v); off (B,Q,M,L,P,2) raw sampling_offsets output; attn (B,Q,M,L,P) softmaxed.
 * samp (B,Q,M,C) and mass (B,Q,M) are saved for backward. */
int dpft_xattn_fwd_f32(const dpft_pyramid* pyr, const float* ref, const float* off,
                       const float* attn, const float* Wv, const float* bv, float* out,
                       float* samp, float* mass, int32_t B, int32_t Q, int32_t M, int32_t D,
                       int32_t P, dpft_stream_t stream);
/* grad_off/grad_attn/grad_ref are overwritten (grad_ref may be NULL); pyr->grad[] are accumulated
 * into with fp32 atomics (caller zero-fills).  Gradients of Wv/bv are tiny reductions of
 * grad_out (x) samp / mass and are left to the host framework. */
int dpft_xattn_bwd_f32(const dpft_pyramid* pyr, const float* ref, const float* off,
                       const float* attn, const float* Wv, const float* bv, const float* grad_out,
                       float* grad_off, float* grad_attn, float* grad_ref, int32_t B, int32_t Q,
                       int32_t M, int32_t D, int32_t P, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused inference decoder (eval mode, no autograd): d_model 16, 8 heads x head_dim 2, L*P <= 20, Mish
 * FFN (d_ffn 32), LayerNorm, 'linear' view reduction, 3-layer bias-free linear heads -- IMPFusion /
 * MPFusion / MLFusion / LinearDetectionHead at config/kradar*.json (src/dprt/models/fusers/mpfusion.py:122-229,
 * :416-514,:617-745, src/dprt/models/layers/ms_deform_attn.py:138-217, src/dprt/models/heads/detection.py:252-275).
 * Parameters are first PACKED (once per weight update) into lane-friendly transposed blobs; the forward is
 * then one call = 2 launches per iteration (self attention of all views | cross attention + FFN of all views +
 * view reduction + heads + next reference points).
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_decoder_view {   /* parameters of one MLFusion, torch layouts (out_features, in_features) */
    const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *norm1_w, *norm1_b;       /* self attention */
    const float *off_w, *off_b, *att_w, *att_b, *val_w, *val_b, *outp_w, *outp_b, *norm2_w, *norm2_b; /* MSDeformAttn */
    const float *ffn1_w, *ffn1_b, *ffn2_w, *ffn2_b, *norm3_w, *norm3_b;                       /* FFN */
} dpft_decoder_view;

int64_t dpft_decoder_packed_view_floats(void);   /* floats per packed MLFusion blob (training cross-attention kernels) */
int64_t dpft_decoder_packed_infer_floats(int32_t Q);  /* floats per packed MLFusion blob of the inference decoder */
int64_t dpft_decoder_packed_head_floats(void);   /* floats per packed reduction+head blob */
/* packed <- one MLFusion's parameters (L levels, P points of its MSDeformAttn) */
int dpft_decoder_pack_view_f32(const dpft_decoder_view* view, int32_t L, int32_t P, float* packed,
                               dpft_stream_t stream);
/* the same for all V views of a layer in one launch; packed = V consecutive blobs */
int dpft_decoder_pack_views_f32(const dpft_decoder_view* views, int32_t V, const int32_t* L, const int32_t* P, float* packed,
                                dpft_stream_t stream);
/* inference decoder's blob of one MLFusion = view `view_index` of one MPFusion layer: per-head in_proj rows with
 * 1/sqrt(d)*log2(e) folded into q, the offsets / logits matrix in the lanes' sample-slot order, the LDS image of the
 * cross-attention kernel, the input-independent part of the self-attention's q / k / v rows over pos (Q,16) =
 * query_embedding.weight (W pos_k + b; with query0 (Q,16) = the learned query table, non-NULL for the FIRST layer only,
 * the whole rows W (query0_k + pos_k) + b: that layer's input is a parameter),
 * and the hand-over to the NEXT layer's self-attention: next_in_proj_w[V] = in_proj_weight (48,16) of the next layer's
 * views (NULL for the last layer), composed with this layer's red_w = reduction_layer.weight (16, 16*V). */
int dpft_decoder_pack_infer_f32(const dpft_decoder_view* view, int32_t L, int32_t P, const float* red_w,
                                const float* const* next_in_proj_w, int32_t view_index, int32_t V,
                                const float* pos, const float* query0, int32_t Q, float* packed, dpft_stream_t stream);
/* packed <- reduction_layer.weight (16, 16*V) and head_w[4][3] = center/size/angle/class x (layers .0,.3,.6)
 * weights, passed as a flat array of 12 pointers */
int dpft_decoder_pack_head_f32(const float* red_w, const float* const* head_w, int32_t V, int32_t num_classes,
                               float* packed, dpft_stream_t stream);

/* work: caller-owned scratch of dpft_decoder_work_floats(B,Q,V) floats.  Final outputs go to
 * center/size/angle/cls: (B,Q,3) (B,Q,3) (B,Q,2) (B,Q,num_classes). */
typedef struct dpft_decoder_fwd {
    int32_t B, Q, V, iters, num_classes;
    int32_t n_points[4];
    const float* packed_views;           /* [iters][V] blobs of dpft_decoder_packed_infer_floats(Q) */
    const float* packed_heads;           /* [iters] blobs of dpft_decoder_packed_head_floats()     */
    const dpft_pyramid* pyr;             /* [V]                                          */
    const float* query0;                 /* fuser.query (Q,16)                           */
    const float* pos;                    /* fuser.query_embedding.weight (Q,16)          */
    const float* center0;                /* querent output (B,Q,3)                       */
    const float* T[4];                   /* label_to_X_t (B,4,4)                         */
    const float* P[4];                   /* label_to_X_p (B,p_rows,4)                    */
    const int64_t* shape[4];             /* X_shape[:, :2] contiguous (B,2) = (H, W)     */
    int32_t p_rows[4], has_t[4];         /* rows of P (3 or 4); transformation.any(): 1 / 0, or -1 = evaluated on the
                                            device (no host read-back of the matrices) */
    float* work;
    float *center, *size, *angle, *cls;
    const float* attn0;                  /* (V,Q,16) from dpft_decoder_attn0_f32, or NULL: the attention output of iteration 0
                                            (a constant of the weights) -- with it a forward is 2 * iters launches, not + 1 */
    int32_t shape_stride;                /* int64 elements between the rows of shape[v]: 0 = 2 (contiguous (B,2)); 3 reads the
                                            first two columns of the dataset's (B,3) X_shape rows in place */
} dpft_decoder_fwd;
/* iteration 0's self-attention output for one batch element -- its input is the learned query table (mpfusion.py:700-703),
 * so it depends on the weights only; call again after every weight change, like the pack functions */
int dpft_decoder_attn0_f32(const float* packed_views, const float* pos, int32_t Q, int32_t V, float* attn0,
                           dpft_stream_t stream);
int64_t dpft_decoder_work_floats(int32_t B, int32_t Q, int32_t V);
/* measurement aid: with DPFT_DEC_DBG=1024 in the environment the decoder kernels stamp a 100 MHz clock at their phase
 * boundaries; copies 2 kernels x 2048 blocks x 8 slots of uint64 (last launches) to dst (tools/decoder_stamps.py) */
int dpft_debug_decoder_stamps(uint64_t* dst);
int dpft_decoder_forward_f32(const dpft_decoder_fwd* d, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training path of the decoder's self-attention block, all V views of one MPFusion layer per call:
 *   y1[v] = LayerNorm1(x + dropout1(out_proj(MHA(x + pos, x + pos, x))))   with attention-probability dropout
 * = MLFusion.forward_self_attn, src/dprt/models/fusers/mpfusion.py:122-148 (nn.MultiheadAttention, d_model 16,
 * 8 heads).  Dropout masks are regenerated in the backward from (*seed, salt); p_drop = 0 disables them.
 * Saved for backward (caller-owned): lse (V,B,Q,8), attn (V,B,Q,16), zhat (V,B,Q,16), rstd (V,B,Q).
 * Backward: parameter gradients are ACCUMULATED into grads[v] (caller zero-fills); dx (V,B,Q,16) is the
 * gradient w.r.t. x through view v, dxp (V,B,Q,16) the part that also flows to pos; scratch holds
 * dpft_selfattn_train_scratch_floats(B,Q,V) floats.  x is (B,Q,16) with batch stride x_bstride (0 = broadcast).
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_sa_params {   /* torch layouts: in_proj (48,16), out_proj (16,16) */
    const float *in_w, *in_b, *out_w, *out_b, *n1_w, *n1_b;
} dpft_sa_params;
typedef struct dpft_sa_grads {
    float *in_w, *in_b, *out_w, *out_b, *n1_w, *n1_b;
} dpft_sa_grads;
int dpft_selfattn_train_fwd_f32(const dpft_sa_params* params, int32_t V, const float* x, int64_t x_bstride,
                                const float* pos, float p_drop, const int64_t* seed, int32_t salt,
                                float* y1, float* lse, float* attn, float* zhat, float* rstd,
                                int32_t B, int32_t Q, dpft_stream_t stream);
int dpft_selfattn_train_bwd_f32(const dpft_sa_params* params, int32_t V, const float* x, int64_t x_bstride,
                                const float* pos, float p_drop, const int64_t* seed, int32_t salt,
                                const float* dy1, const float* lse, const float* attn, const float* zhat,
                                const float* rstd, const dpft_sa_grads* grads, float* dx, float* dxp,
                                float* scratch, int32_t B, int32_t Q, dpft_stream_t stream);
int64_t dpft_selfattn_train_scratch_floats(int32_t B, int32_t Q, int32_t V);

/* ------------------------------------------------------------------------------------------
 * Training path of the decoder's deformable cross-attention + FFN block, all V views of one MPFusion layer:
 *   y2 = LayerNorm2(y1 + dropout2(output_proj(MSDeformAttn(y1 + pos, ref, pyramid))))
 *   y3 = LayerNorm3(y2 + dropout4(ffn2(dropout3(Mish(ffn1(y2))))))
 * = MLFusion.forward_cross_attn + forward_ffn, src/dprt/models/fusers/mpfusion.py:150-229 and
 * src/dprt/models/layers/ms_deform_attn.py:138-217.  `views` are the torch-layout parameters, `packed` the V
 * blobs made from them by dpft_decoder_pack_view_f32 (the caller re-packs after every weight update).
 * y1, y3, dy3, dy1, dqp are (V,B,Q,16); ref/dref (V,B,Q,2); pos (Q,16).
 * Forward: `saved` (V,B*Q,dpft_xattn_ffn_train_saved_floats()) or NULL receives what the backward would otherwise have to
 * gather again (the attention-weighted sampled features and in-bounds masses of every head).
 * Backward: recomputes the ALU-only part of the row's forward (takes the gathered part from `saved`; NULL = gathers
 * again), adds the pyramid gradients into pyr[v].grad[] (caller zero-fills) and writes the per-row factors of the
 * parameter gradients into rows (V,B*Q,dpft_xattn_ffn_train_row_floats()); the column layout is documented at the top
 * of dpft_amd/csrc/decoder_train_x.hip (XR_*) and consumed by dpft_amd/models/fusers/train_fused.py.
 * Pyramid gradients: `scratch` (V,B*Q,dpft_xattn_ffn_train_scratch_floats()) or NULL.  With it, maps of at most 2048
 * pixels whose gradient buffer is not replicated (grad_replicas <= 1) are scattered through an LDS image of the map by a
 * second launch (one fp32 atomic per touched element and query chunk instead of one per sample corner); larger maps, and
 * every map when scratch is NULL, take one fp32 atomic per sample corner and channel.  Both forms add the same terms;
 * only the order of the fp32 additions differs.
 * ---------------------------------------------------------------------------------------- */
int64_t dpft_xattn_ffn_train_row_floats(void);
int64_t dpft_xattn_ffn_train_saved_floats(void);
int64_t dpft_xattn_ffn_train_scratch_floats(void);
int dpft_xattn_ffn_train_fwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                 int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                 const float* ref, float p_drop, const int64_t* seed, int32_t salt, float* y3,
                                 float* saved, int32_t B, int32_t Q, dpft_stream_t stream);
int dpft_xattn_ffn_train_bwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                 int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                 const float* ref, float p_drop, const int64_t* seed, int32_t salt,
                                 const float* saved, const float* dy3, float* dy1, float* dqp, float* dref,
                                 float* rows, float* scratch, int32_t B, int32_t Q, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training path of the decoder's view reduction + detection head + next reference points (one layer):
 *   x = reduction_layer(cat_v y3[v]);  out_g = act_g(MLP_g(x));  center = out_center + prev_center;
 *   refs[v] = reference point of `center` in view v (input of the NEXT layer's cross attention)
 * = MPFusion 'linear' reduction (src/dprt/models/fusers/mpfusion.py:416-514), LinearDetectionHead
 * (src/dprt/models/heads/detection.py:252-275) and IMPFusion.get_reference_points (mpfusion.py:617-696).
 * `packed` comes from dpft_decoder_pack_head_f32.  y3 == NULL: only refs of prev_center are computed.
 * Backward: NULL gradient inputs mean "no gradient"; rows (B*Q, dpft_head_train_row_floats()) receives the
 * per-row factors of the weight gradients (columns HR_* in dpft_amd/csrc/decoder_train_h.hip).
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_head_train {
    const float* y3;               /* (V,B,Q,16) */
    const float* packed;           /* dpft_decoder_packed_head_floats() floats */
    const float* red_w;            /* reduction_layer.weight (16, 16*V) */
    const float* head_w[4][3];     /* center/size/angle/class x layers .0,.3,.6 (torch layouts) */
    const float* prev_center;      /* (B,Q,3) */
    const float* T[4];             /* (B,4,4) per view (may be NULL when has_t == 0) */
    const float* P[4];             /* (B,p_rows,4) */
    const int64_t* shape[4];       /* (B,2) = (H, W) */
    int32_t p_rows[4], has_t[4];
    int32_t num_classes;
    float *x, *center, *size, *angle, *cls;      /* forward outputs (B,Q,16) (B,Q,3) (B,Q,3) (B,Q,2) (B,Q,ncls) */
    float* refs;                                   /* (V,B,Q,2) or NULL */
    const float *dx, *dcenter, *dsize, *dangle, *dcls, *drefs;   /* backward inputs, any may be NULL */
    float *dy3, *dcenter_prev, *rows;              /* backward outputs */
    int32_t shape_stride;                          /* int64 elements between the rows of shape[v]; 0 = 2 */
} dpft_head_train;
int64_t dpft_head_train_row_floats(void);
int dpft_head_train_fwd_f32(const dpft_head_train* h, int32_t B, int32_t Q, int32_t V, dpft_stream_t stream);
int dpft_head_train_bwd_f32(const dpft_head_train* h, int32_t B, int32_t Q, int32_t V, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradients of the fused training decoder from the per-row factor matrices its backward kernels write:
 *   out[g][out_off + a * n_b + b] = sum_r rows[g][r][col_a + a] * rows[g][r][col_b + b]     (a < n_a, b < n_b)
 * col_b < 0 (with n_b == 1): column sums of the a-columns.  rows (G,R,W); specs is a HOST array of <= 40 entries;
 * one launch for all specs and groups, fixed summation order.  Replaces the bmm / einsum / sum launches of
 * dpft_amd/models/fusers/train_fused.py (torch -> Tensile kernels) on the training step.
 * ---------------------------------------------------------------------------------------- */
typedef struct dpft_outer_spec {
    int32_t col_a, n_a, col_b, n_b;
    int64_t out_off;
} dpft_outer_spec;
int dpft_rows_outer_f32(const float* rows, int32_t G, int32_t R, int32_t W, const dpft_outer_spec* specs,
                        int32_t n_specs, float* out, int64_t out_gstride, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Up to DPFT_MEMOPS_MAX device-to-device copies / zero fills (src == NULL) in ONE launch: the host glue's small per-step
 * tensor copies and clears (static inputs of a replayed decoder graph, static-address plan inputs, reducer clears) -- the
 * reference's counterparts are Tensor.copy_ / zero_ calls of its training loop (src/dprt/training/trainer.py:107-150).
 * Pointers 4-byte aligned, sizes multiples of 4 bytes (16-byte accesses when both pointers allow); ops is a HOST array.
 * ---------------------------------------------------------------------------------------- */
#define DPFT_MEMOPS_MAX 16
typedef struct dpft_memop {
    void* dst;
    const void* src;      /* NULL = zero fill */
    uint64_t bytes;
} dpft_memop;
int dpft_memops(int32_t n, const dpft_memop* ops, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * dst[i] (+)= sum_s sum_{l < n_lead_s} src_s[l * inner + i], i < inner: leading-axis sums of the training decoder's backward in
 * one launch, fixed summation order (table order, leading index ascending).  The reference's counterparts are the implicit
 * sums of autograd (expand / repeat backward, gradient accumulation of a parameter with several uses:
 * src/dprt/models/fusers/mpfusion.py:601-668 uses query_embedding.weight in every layer).  srcs is a HOST array.
 * ---------------------------------------------------------------------------------------- */
#define DPFT_SUM_SRCS_MAX 16
typedef struct dpft_sum_src {
    const float* src;
    int32_t n_lead;
} dpft_sum_src;
int dpft_sum_leading_f32(int32_t n_src, const dpft_sum_src* srcs, int64_t inner, float* dst, int32_t accumulate,
                         dpft_stream_t stream);

/* dst_i[k] += src_i[k] (fp32) for the n entries of a DEVICE-resident dpft_memop table (bytes = 4 * elements), one launch:
 * gradient accumulation of many small parameters (autograd's AccumulateGrad of the reference's training loop,
 * src/dprt/training/trainer.py:134-141).  The table may be written after the launch has been CAPTURED in a hipGraph. */
int dpft_add_many_f32(int32_t n, const dpft_memop* table_device, dpft_stream_t stream);

/* *ptrs[i] += increment for n <= DPFT_I64_PTRS_MAX int64 device scalars in one launch (ptrs: HOST array): the
 * num_batches_tracked counters a train-mode BatchNorm forward bumps (torch.nn.BatchNorm2d; the reference's backbones are
 * torchvision ResNets, src/dprt/models/backbones/resnet.py:120-176). */
#define DPFT_I64_PTRS_MAX 256
int dpft_i64_add_many(int32_t n, int64_t* const* ptrs, int64_t increment, dpft_stream_t stream);

/* Dropout seed of the fused training decoder: *snap = *state; *state += increment (one launch, capturable).  The reference
 * draws its dropout masks from torch's generator (nn.Dropout in src/dprt/models/fusers/mpfusion.py:95-119). */
int dpft_seed_advance(int64_t* state, int64_t* snap, int64_t increment, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Matcher cost helper: GIoU3D of yaw-only boxes, (B,N) predictions x (B,Mg) targets.
 * boxes are (x,y,z,l,w,h,yaw) rows of 7 floats; out (B,N,Mg).
 * ---------------------------------------------------------------------------------------- */
int dpft_giou3d_yaw_f32(const float* pred, const float* gt, float* out, int32_t B, int32_t N,
                        int32_t Mg, dpft_stream_t stream);

/* Per-sample label tensors (host arrays of B device pointers: gt_center (m,3), gt_size (m,3), gt_angle (m,2), gt_class (m,C)
 * one-hot, fp32 contiguous; counts_host[b] = m) -> the padded batch tensors the matcher / loss / metric kernels read:
 * gt_box (B,Mmax,8) = center | size | angle, gt_onehot (B,Mmax,C), gt_id (B,Mmax) = argmax(gt_class), counts (B).
 * Replaces the per-sample decollate of src/dprt/training/assigner.py:92-111 / loss.py:524-541 (B <= 32 per call). */
int dpft_pack_targets_f32(const float* const* center, const float* const* size, const float* const* angle,
                          const float* const* cls, const int32_t* counts_host, int32_t B, int32_t Mmax, int32_t C,
                          float* gt_box, float* gt_onehot, int32_t* gt_id, int32_t* counts, dpft_stream_t stream);

/* Whole matcher cost (src/dprt/training/assigner.py:113-132) in one launch:
 * cost[b,n,j] = w0*(-cls[b,n,gt_id[b,j]]) + w1*L1(center) + w2*L1(size) + w3*L1(angle) - w4*GIoU3D, 0 for j >= counts[b].
 * gt_box (B,Mmax,8) = center | size | angle(2); weights5 is a HOST array. */
int dpft_match_cost_f32(const float* cls, const float* center, const float* size, const float* angle,
                        const float* gt_box, const int32_t* gt_id, const int32_t* counts,
                        const float* weights5, float* cost, int32_t B, int32_t N, int32_t Mmax, int32_t C,
                        dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * HOST function (no launch): the assignments of a batch, cost (B,N,Mmax) fp32 in host memory as dpft_match_cost_f32 wrote it,
 * sample b restricted to its first counts[b] targets.  match (B,Mmax,2) int32 = (query, target) pairs by ascending query
 * index, -1 padded; n_matched[b] = min(N, counts[b]).  = scipy.optimize.linear_sum_assignment per sample
 * (src/dprt/training/loss.py:305 through the assigner; scipy is third-party: its published algorithm -- Crouse 2016, shortest
 * augmenting paths, double precision -- restated in dpft_amd/csrc/cabi.cpp).  Non-finite or infeasible costs are an error.
 * ---------------------------------------------------------------------------------------- */
int dpft_lsap_batch_f32(const float* cost, int32_t B, int32_t N, int32_t Mmax, const int32_t* counts, int32_t* match,
                        int32_t* n_matched);
/* The host window of a training step in one call (round 5): dpft_lsap_batch_f32 on the read-back cost matrices -> upload of
 * assignments | matched counts from the caller's page-locked `packed_host` into `packed_dev` (both (B * Mmax * 2 + B) int32) ->
 * dpft_set_loss_fwd_total_f32 -> and, when dcls is not NULL, dpft_set_loss_bwd_f32 with d total / d term = sel straight into
 * (dcls, dcenter, dsize, dangle).  Replaces training/loss.py:296-373 + the `loss.backward()` hop into the head outputs between the
 * matcher's sync and the decoder's backward. */
int dpft_assign_loss_f32(const float* cost_host, const int32_t* counts_host, int32_t* packed_host, int32_t* packed_dev,
                         const float* cls, const float* center, const float* size, const float* angle, const float* gt_box,
                         const float* gt_onehot, const float* weights5, float alpha, const float* sel, float* scratch,
                         float* losses5, float* total, float* dcls, float* dcenter, float* dsize, float* dangle, int32_t B,
                         int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream);

/* The same assignments ON THE DEVICE (round 5; dpft_amd/csrc/lsap.hip): one wavefront per sample runs the step sequence of
 * dpft_lsap_batch_f32 (double precision duals, scipy's tie rule as an associative reduction over the 64 lanes) on the cost
 * matrices dpft_match_cost_f32 left in HBM: cost (B,N,Mmax), counts (B), match (B,Mmax,2), n_matched (B) all DEVICE memory; the
 * same pairs in the same order as the host function.  A kernel cannot return an error: status (one int32 in device or
 * page-locked host memory, zero before the first call, may be NULL) receives 1 + b for a non-finite entry / 0x10000 + b for an
 * infeasible problem / 0x20000 + b for counts[b] > Mmax of sample b, and that sample gets no pairs.  With it the training step has no host round trip between the
 * matcher and the criterion (src/dprt/training/loss.py:296-373: scipy on a .cpu() copy per sample). */
int dpft_lsap_batch_dev_f32(const float* cost, const int32_t* counts, int32_t* match, int32_t* n_matched, int32_t* status,
                            int32_t B, int32_t N, int32_t Mmax, dpft_stream_t stream);
/* dpft_assign_loss_f32 without the host: dpft_lsap_batch_dev_f32 -> dpft_set_loss_fwd_total_f32 -> (dcls != NULL)
 * dpft_set_loss_bwd_f32, three launches from one call.  packed_dev (B * Mmax * 2 + B) int32 = assignments | matched counts. */
int dpft_assign_loss_dev_f32(const float* cost, const int32_t* counts, int32_t* packed_dev, int32_t* status, const float* cls,
                             const float* center, const float* size, const float* angle, const float* gt_box,
                             const float* gt_onehot, const float* weights5, float alpha, const float* sel, float* scratch,
                             float* losses5, float* total, float* dcls, float* dcenter, float* dsize, float* dangle, int32_t B,
                             int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SetCriterion + batch reduction 'mean' (src/dprt/training/loss.py:17-60 focal loss with the raw-logit p_t,
 * :176-373 criterion, :486-564 weighting / reduction) for given assignments.
 * losses5 = batch-reduced, weighted (total_class, object_class, center, size, angle); match (B,Mmax,2) int32 =
 * (query, target) pairs in assignment order; weights5 (HOST array) in the same term order; gout5 (DEVICE) = upstream
 * gradient per term.  The backward writes d(sum_k gout5[k]*losses5[k]) / d{cls,center,size,angle}.
 * ---------------------------------------------------------------------------------------- */
int dpft_set_loss_fwd_f32(const float* cls, const float* center, const float* size, const float* angle,
                          const float* gt_box, const float* gt_onehot, const int32_t* match,
                          const int32_t* counts, const float* weights5, float alpha, float* losses5,
                          int32_t B, int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream);
/* The same forward in ONE launch without a cleared output, plus total = sum_k sel[k] * losses5[k] (sel: DEVICE, 5 floats -- the
 * configured terms; the reference sums them in src/dprt/training/loss.py:556-562): per-block partial sums in `scratch`
 * (dpft_set_loss_scratch_floats(B, N) floats, ZERO when first handed over, left zero), added in block order by the block that
 * draws the last ticket. */
int64_t dpft_set_loss_scratch_floats(int32_t B, int32_t N);
int dpft_set_loss_fwd_total_f32(const float* cls, const float* center, const float* size, const float* angle,
                                const float* gt_box, const float* gt_onehot, const int32_t* match,
                                const int32_t* counts, const float* weights5, float alpha, const float* sel,
                                float* scratch, float* losses5, float* total, int32_t B, int32_t N, int32_t Mmax,
                                int32_t C, dpft_stream_t stream);
int dpft_set_loss_bwd_f32(const float* cls, const float* center, const float* size, const float* angle,
                          const float* gt_box, const float* gt_onehot, const int32_t* match,
                          const int32_t* counts, const float* weights5, float alpha, const float* gout5,
                          float* dcls, float* dcenter, float* dsize, float* dangle, int32_t B, int32_t N,
                          int32_t Mmax, int32_t C, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input preprocessing on the device (the reference does it per sample in DataLoader workers):
 * camera frame resize = torchvision.transforms.functional.resize(bilinear, no antialias) on the HWC frame
 * (src/dprt/datasets/kradar/dataset.py:319-341), from fp32 or straight from the decoded u8 bytes; radar map scaling
 * (v - in_lo) / (in_hi - in_lo) * (out_hi - out_lo) + out_lo clipped to [out_lo, out_hi] (dataset.py:295-317 with
 * in = (min_power, max_power) = (100, 200), out = (0, 255)).  NHWC in and out.
 * ---------------------------------------------------------------------------------------- */
int dpft_resize_bilinear_nhwc_f32(const float* src, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd,
                                  int32_t Wd, int32_t C, dpft_stream_t stream);
int dpft_resize_bilinear_nhwc_u8(const uint8_t* src, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd,
                                 int32_t Wd, int32_t C, dpft_stream_t stream);
int dpft_scale_clip_f32(const float* x, float* y, int64_t n, float in_lo, float in_hi, float out_lo,
                        float out_hi, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-step detection metrics of the reference's train / validation loops (src/dprt/training/trainer.py:134,
 * src/dprt/evaluation/metric.py): mAP3D (IoU threshold, nelem-point "interpolated" curve, :16-151) and mGIoU3D
 * (:154-253) of every sample, on padded targets.  out (B,2) = (mAP, mGIoU) per sample (the caller applies the batch
 * reduction); scratch holds 2*B*N*Mmax floats (IoU, GIoU of every pair).  N <= 1024, Mmax <= 256, C <= 16.
 * ---------------------------------------------------------------------------------------- */
int dpft_detection_metrics_f32(const float* cls, const float* center, const float* size, const float* angle,
                               const float* gt_box, const float* gt_onehot, const int32_t* counts,
                               float threshold, int32_t nelem, float* scratch, float* out, int32_t B,
                               int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * K-Radar export selection: KRadarExporter._construct_objects
 * (src/dprt/evaluation/exporters/kradar.py:231-294) for all B samples and T <= 8 confidence thresholds at once.
 * cls (B,N,C) raw scores, center (B,N,3), size (B,N,3), angle (B,N,2)=[sin,cos]; conf_thrs is a HOST array of T
 * floats.  An object survives threshold t iff  argmax(cls)-1 >= 0  &  max(cls) >= thr_t  &  0<x<72, -6.4<y<6.4,
 * -2<z<6, -50<yaw<50 (:268-277).  rows (B,T,N,8) receives the survivors in candidate order as
 * [category, h, w, l, y, z, x, theta] (the non-constant columns of :283-292), counts (B,T) their number;
 * mask (B,N) (optional, may be NULL) bit t = survived threshold t.  One block per sample.
 * ---------------------------------------------------------------------------------------- */
int dpft_export_select_f32(const float* cls, const float* center, const float* size, const float* angle,
                           const float* conf_thrs, int32_t T, float* rows, int32_t* counts, uint8_t* mask,
                           int32_t B, int32_t N, int32_t C, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Radar tesseract -> RA / EA feature maps: KRadarProcessor.get_radar_data,
 * src/dprt/datasets/kradar/processor.py:588-633.  tesseract (D,R,E,A) linear power; doppler_raster (D);
 * ra (R,A,6), ea (E,A,6) = (rcs max, rcs median, rcs var, doppler peak, doppler centre, doppler var); the EA map
 * folds the range bins [r_lo, r_hi) (reference: 4, 252).  scratch: dpft_radar_projection_scratch_floats(D,R,E,A);
 * D <= 64, folded axes <= 256 elements.
 * ---------------------------------------------------------------------------------------- */
int64_t dpft_radar_projection_scratch_floats(int32_t D, int32_t R, int32_t E, int32_t A);
int dpft_radar_projection_f32(const float* tesseract, const float* doppler_raster, float* ra, float* ea,
                              float* scratch, int32_t D, int32_t R, int32_t E, int32_t A, int32_t r_lo,
                              int32_t r_hi, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW (torch.optim.AdamW semantics: decoupled decay, bias correction, no amsgrad),
 * the optimizer the reference builds at src/dprt/training/trainer.py:233 / optimizer.py:6-7.
 * chunks: device array of {float* p; const float* g; float* m; float* v; int32 n; int32 tensor}
 * (4 pointers + 2 int32 = 40 bytes each); active: device int32 per tensor (0 = skip: gradient is None) or NULL.
 * skipped (or NULL): device int32 per tensor, the number of steps the tensor sat out; its bias corrections use
 * step - skipped[t] (the per-parameter state["step"] of torch.optim.AdamW).  A row with m == NULL is the tensor's
 * "marker row": it carries no elements and advances skipped[t] when the tensor is inactive (one such row per tensor).
 * gate (round 5, or NULL): device float, the step's loss -- the reference steps only `if loss > 0` (training/trainer.py:131);
 * with a gate that is not positive every tensor sits the launch out as if it had no gradient, so the host need not read the
 * loss back before it launches the backward and the optimizer.
 * ---------------------------------------------------------------------------------------- */
int dpft_adamw_f32(const void* chunks, int32_t n_chunks, const int32_t* active, int32_t* skipped, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, const float* gate, dpft_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py `roofline`): while started, every dpft_conv2d_nhwc_* call is bracketed
 * by HIP events on its launch stream.  Not thread-safe; do not use inside a graph capture.
 * ---------------------------------------------------------------------------------------- */
int dpft_profile_start(void);
/* on != 0: the ResNet plans keep weight gradients / weight transposes on the main stream (no side stream) WITHOUT event
 * brackets, so that an external tracer (rocprofv3 --kernel-trace) sees every conv kernel running alone: the same
 * serialized step bench.py brackets, reproducible from a profile.  Off by default. */
int dpft_profile_serialize(int32_t on);
int32_t dpft_profile_stop(void);                 /* -> number of recorded launches */
float dpft_profile_overhead_ms(void);            /* elapsed time of an empty event bracket, calibrated by
                                                    dpft_profile_start and already subtracted by dpft_profile_get */
/* kind 0 fwd / 1 dgrad / 2 wgrad; flops = algorithmic 2*M*K*kh*kw*C; ms = event-timed duration;
 * shape7 = B,H,W,C,K,k,stride.  The stream must have been synchronised. */
int dpft_profile_get(int32_t i, int32_t* kind, double* flops, float* ms, int32_t* shape7);
/* the pipe record i was launched on, written by the dispatch code (not derived from the shape): 0 fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), 1 three-term bf16 split (six v_mfma_f32_32x32x16_bf16 per fp32 product, conv_x3.hip),
 * 2 bf16 operands on the bf16 MFMA (mixed precision), 3 no matrix core (thin-channel / 16-channel / generic kernels).
 * bench.py prices every conv against ITS pipe's peak (roofline.frac_blended). */
int dpft_profile_get_family(int32_t i, int32_t* family);

#ifdef __cplusplus
}
#endif
#endif /* DPFT_HIP_H */
