// Shared helpers for libdpft_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dpft_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace dpft {

void set_error(const char* fmt, ...);
// First pass of a BatchNorm backward (sums[0][k] += sum d, sums[1][k] += sum d * xhat, d = dout under the layer's ReLU mask)
// folded into the epilogue of the data-gradient GEMM that PRODUCES dout: `y` = the layer's input, `bnp` its BN block,
// mask from `mask8` (byte mask of the block output) or recomputed as bn(y) > 0 (`self_mask`); `sums` pre-zeroed [2][K].
// `applied` (out): the launch carried the reduction; otherwise the caller runs bn_bwd_reduce_prezeroed as before.
struct BnReduceFuse {
    const float* y;
    const float* bnp;
    const unsigned char* mask8;
    int self_mask;
    float* sums;
    bool applied;
};
int conv_dgrad_residual(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx, const float* res_src,
                        const float* res_mask, void* workspace, dpft_stream_t stream, BnReduceFuse* fuse = nullptr,
                        const unsigned char* res_mask8 = nullptr);      // conv.hip
// Train-mode BatchNorm statistics as COLUMN SUMS ("sums" form, round 6): the producing conv adds n * mean and M2 + n * mean^2 of
// every tile to per-channel accumulators, and every forward CONSUMER -- the next conv's operand prologue, the block-closing
// elementwise pass -- derives (mean, scale, beta) for its channels itself (bn_from_sums below, the same arithmetic everywhere):
// no finalize launch between a conv and its consumer.  The BN blocks the backward reads and the running statistics come from
// ONE batched launch at the end of the forward (bn_finalize_sums_batch, bn.hip).
// The accumulators are 96-bit FIXED-POINT numbers in two 64-bit words (units of 2^-46: a high word counting fours, a low word of
// 48 bits per addend), added to with INTEGER atomics: integer addition is associative, so the totals -- and with them every BN block
// -- are the same bits whatever order the tiles arrive in (a floating-point atomic would move them from run to run), and exact to
// 2^-46 absolute per addend (the addends come from fp32 tile statistics).  Layout [4][K] words: sum 1 high, low, sum 2 high, low;
// room for 65 536 tiles per launch and |sum| < 2^65.
struct BnSumsRef {
    const unsigned long long* sums;      // [4][K]; null = off
    const float* gamma;
    const float* beta;
    double invn;             // 1 / (B * OH * OW)
    float eps;
    int K;
};
__device__ __forceinline__ void bn_sums_add(unsigned long long* acc, int K, int c, double s1, double s2) {
    const double h1 = floor(s1 * 0.25), h2 = floor(s2 * 0.25);
    const unsigned long long l1 = (unsigned long long)((s1 - h1 * 4.0) * 0x1p46), l2 = (unsigned long long)((s2 - h2 * 4.0) * 0x1p46);
    atomicAdd(acc + c, (unsigned long long)(long long)h1);      // (two's complement: the wrap-around add is the signed add)
    atomicAdd(acc + K + c, l1);
    atomicAdd(acc + 2 * K + c, (unsigned long long)(long long)h2);
    atomicAdd(acc + 3 * K + c, l2);
}
__device__ __forceinline__ double bn_sums_get(const unsigned long long* acc, int K, int c, int which) {
    return (double)(long long)acc[(2 * which) * K + c] * 4.0 + (double)acc[(2 * which + 1) * K + c] * 0x1p-46;
}
__device__ __forceinline__ void bn_from_sums(const BnSumsRef& r, int c, float& mean, float& scale, float& beta, float& invstd) {
    const double m = bn_sums_get(r.sums, r.K, c, 0) * r.invn;
    const double var = fmax(bn_sums_get(r.sums, r.K, c, 1) * r.invn - m * m, 0.0);
    mean = (float)m;
    invstd = 1.0f / sqrtf((float)var + r.eps);
    scale = r.gamma[c] * invstd;
    beta = r.beta[c];
}
// rows mean | scale | beta of a prologue table [3][K] in LDS, from a BN block or from the sums (all threads of the workgroup; the
// caller synchronises)
__device__ __forceinline__ void fill_pro_table(float* tab, const float* bnp, const BnSumsRef& r, int K, int tid, int nt) {
    if (r.sums) {
        for (int c = tid; c < K; c += nt) {
            float mean, scale, beta, invstd;
            bn_from_sums(r, c, mean, scale, beta, invstd);
            tab[c] = mean; tab[K + c] = scale; tab[2 * K + c] = beta;
        }
    } else {
        for (int i = tid; i < 3 * K; i += nt) tab[i] = bnp[i];
    }
}
// Train-mode BatchNorm finalize folded into the producing forward conv: tiles add pivoted sums to `acc` [2][K] (zero
// before the launch, like `ticket`), the workgroup with the last ticket writes the BN block `bnp` [4][K] and updates the
// running statistics.  `applied` (out) as in BnReduceFuse.
struct BnFinalFuse {
    float* acc;
    int* ticket;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* bnp;
    float eps, momentum;
    bool applied;
    int slab = 0;      // > 0: deterministic form for launches of at most `slab` row tiles (statistics slab + per-column-tile tickets)
    unsigned long long* sums = nullptr;      // the "sums" form (BnSumsRef): [4][K] words, zero before the launch; no ticket, nothing finalized here
};
// `pro_sums` (may be null): the prologue's BatchNorm comes as column sums; *pro_sums_used tells whether the kernel taken derives
// its table from them (otherwise NOTHING was launched: the caller finalizes the layer into `pro_bn` and calls again without)
int conv_fwd_bnfinal(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, const float* pro_bn,
                     int32_t pro_relu, float* y, float* stats, void* workspace, dpft_stream_t stream, BnFinalFuse* fuse,
                     const BnSumsRef* pro_sums = nullptr, bool* pro_sums_used = nullptr);
int conv_dgrad_fused(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx, int32_t accumulate,
                     void* workspace, dpft_stream_t stream, BnReduceFuse* fuse);
// bn.hip -- `act16`: the activation / gradient tensors (y, dout, out, dy, res) are bf16 in memory (the pointers keep
// their float* type); BN blocks, sums and parameter gradients are fp32
// `mask8` (one byte per 4 channels: bit e = element e of the group passed the ReLU): written by bn_act_any next to the
// activation, read by the backward passes INSTEAD of the block output `out` -- 1 byte where the mask test read 16
int bn_bwd_reduce_prezeroed(const float* y, const float* dout, const float* out, const float* mask_bnp, const float* bnp,
                            float* sums, int64_t M, int32_t K, bool act16, dpft_stream_t stream,
                            const unsigned char* mask8 = nullptr);
int bn_bwd_apply_zeroing(const float* y, const float* dout, const float* out, const float* mask_bnp, const float* bnp,
                         const float* gamma, const float* sums, float* dy, float* dgamma, float* dbeta, int64_t M,
                         int32_t K, float* zero_buf, int32_t zero_n, bool act16, dpft_stream_t stream,
                         const unsigned char* mask8 = nullptr, bool frozen = false);      // frozen: running-statistics BN (no mean terms)
int bn_act_any(const float* y, const float* bnp, const float* res, const float* res_bnp, int32_t relu, float* out,
               float* out32, int64_t M, int32_t K, bool act16, dpft_stream_t stream, unsigned char* mask8 = nullptr);
// the same pass with one or both BatchNorms given as column sums (ys / rs: .sums null = take the BN block); *used = false (and
// nothing launched) where only the generic kernel fits the shape
int bn_act_sums(const float* y, const float* bnp, const BnSumsRef& ys, const float* res, const float* res_bnp, const BnSumsRef& rs,
                int32_t relu, float* out, int64_t M, int32_t K, dpft_stream_t stream, unsigned char* mask8, bool* used,
                bool act16 = false, float* out32 = nullptr);
struct BnSumsBatch {      // sums -> BN block [4][K] + running statistics of up to MAX layers in one launch (bn.hip)
    static constexpr int MAX = 48;      // (kernel arguments: 68 bytes per layer)
    const unsigned long long* sums[MAX];
    const float *gamma[MAX], *beta[MAX];
    float *rm[MAX], *rv[MAX], *out[MAX];
    double invn[MAX];
    long long M[MAX];
    int K[MAX];
    int n;
    float eps, momentum;
};
int bn_finalize_sums_batch(const BnSumsBatch& batch, dpft_stream_t stream);
int bn_relu_maxpool_any(const float* y, const float* bnp, float* out, int32_t B, int32_t H, int32_t W, int32_t K,
                        int32_t PH, int32_t PW, bool out16, dpft_stream_t stream);
int bn_relu_maxpool_bwd_any(const float* y, const float* bnp, const float* dout, float* dact, int32_t B, int32_t H,
                            int32_t W, int32_t K, int32_t PH, int32_t PW, bool dout16, dpft_stream_t stream);
int add_inplace_any(float* a, const float* b, int64_t n, bool a16, dpft_stream_t stream);
int cvt_f32_to_bf16(const float* src, float* dst_bf16, int64_t n, dpft_stream_t stream);
struct BnEvalBatch {      // eval-mode BN blocks of up to 16 layers (bn.hip)
    const float *gamma[16], *beta[16], *rm[16], *rv[16];
    float* out[16];
    int K[16];
    int n;
    float eps;
};
int bn_eval_params_batch(const BnEvalBatch& batch, dpft_stream_t stream);
struct TransposeBatch {      // [K][taps][C] -> [C][taps][K] of up to MAX weight tensors in one launch (conv.hip)
    static constexpr int MAX = 80;
    const float* w[MAX];
    float* wt[MAX];
    int K[MAX], taps[MAX], C[MAX];
    int blk_start[MAX + 1];
    int n;
    int mode = 0;      // 0: transposed fp32 copy; 1: transposed, rounded to bf16; 2: rounded to bf16 in place order (no transpose)
    // tile edge of the launch: 64 (16-byte accesses on both sides, 256-byte row segments) for the tensors of a ResNet body
    // (K, C multiples of 64); the caller flushes the batch when a tensor that needs the 32-wide form arrives (fits())
    int tile = 0;
    static int tile_of(int k, int c) { return (k % 64 == 0 && c % 64 == 0) ? 64 : 32; }
    bool fits(int k, int c) const { return n == 0 || tile == tile_of(k, c); }
    void add(const float* src, float* dst, int k, int t, int c) {
        if (n == 0) tile = tile_of(k, c);
        w[n] = src; wt[n] = dst; K[n] = k; taps[n] = t; C[n] = c;
        if (n == 0) blk_start[0] = 0;
        blk_start[n + 1] = blk_start[n] + ((c + tile - 1) / tile) * ((k + tile - 1) / tile) * t;
        ++n;
    }
};
int weight_transpose_batch(const TransposeBatch& tb, dpft_stream_t stream);
int conv_mode_key();          // conv.hip: compute mode + split switch (what tile selection depends on)
bool profiling_active();      // conv.hip: true between dpft_profile_start / dpft_profile_stop
// zero `bytes` (multiple of 4) of device memory with a KERNEL (conv.hip).  hipMemsetAsync becomes a memset node when the
// call is captured into a hipGraph, and replayed backward stages with memset nodes intermittently produced garbage
// gradients (round 3, tools/plan_graph_check.py: never with a device sync around the graph launch, never in the
// memset-free forward graphs); kernel nodes only is also what the decoder graphs consist of.
int zero_fill(void* ptr, size_t bytes, dpft_stream_t stream);
int bias_grad_ws(const float* dy, float* db, int64_t M, int32_t K, void* workspace, dpft_stream_t stream);

// Dynamic-LDS cap of a kernel above 64 KiB (hipFuncAttributeMaxDynamicSharedMemorySize): the attribute is PER DEVICE, so the
// "largest size granted so far" cache is keyed by the current device -- one process driving several GPUs must not skip the
// call on the second one.  One LdsGrant per kernel instantiation (a function-local static at the launch site).
struct LdsGrant {
    size_t granted[16] = {};
};
inline bool lds_grant(LdsGrant& g, const void* fn, size_t lds) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& have = g.granted[dev & 15];
    if (lds <= have) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    have = lds;
    return true;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return DPFT_ERR_LAUNCH;
    }
    return DPFT_OK;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr int kNumCU = 256;   // MI355X
constexpr int kNumXCD = 8;

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids land on the same
// XCD (hardware places block b on XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / kNumXCD, r = nwg % kNumXCD;
    const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace dpft

// Wave priorities (s_setprio: the CU arbitrates between resident waves by priority, then age).  The BatchNorm passes and
// the forward / data-gradient GEMMs sit on the step's critical chain; the weight-gradient GEMMs that run beside them on
// their own stream do not, and their fp32 MFMAs occupy the vector ALUs every other kernel on the CU needs.
// Measured (tools/ab_libs.sh, same box, alternating builds): BN 3 + GEMM 1 = 29.3-29.5 ms per step, none = 30.1 ms.
#ifndef DPFT_PRIO_BN
#define DPFT_PRIO_BN 3
#endif
#ifndef DPFT_PRIO_IGEMM
#define DPFT_PRIO_IGEMM 1
#endif
#define DPFT_SETPRIO_BN() do { if (DPFT_PRIO_BN) __builtin_amdgcn_s_setprio(DPFT_PRIO_BN); } while (0)
#define DPFT_SETPRIO_IGEMM() do { if (DPFT_PRIO_IGEMM) __builtin_amdgcn_s_setprio(DPFT_PRIO_IGEMM); } while (0)

#define DPFT_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            dpft::set_error(__VA_ARGS__);       \
            return DPFT_ERR_ARG;                \
        }                                       \
    } while (0)
