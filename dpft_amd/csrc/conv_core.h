// Shared pieces of the convolution family (conv.hip, conv_x3.hip): launch arguments, the epilogue every implicit-GEMM
// main loop ends with (split-K fix-up, BatchNorm tile statistics, fused BatchNorm-backward reduction, staged stores),
// the tile decoder.  Moved out of conv.hip in round 5 so that the 3 x bf16 split kernels build as their own translation unit.
#pragma once
#include "common.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

namespace dpft {

constexpr int BK = 32;        // generic path K-step
constexpr int BKV = 64;       // vector path K-step: one barrier pair per 2048+ MFMA cycles, and the register
                              // prefetch of the next step has that long to land (HBM/L2 latency under load)
constexpr int LDK = BKV + 4;  // vector path: padded LDS row (272 B = 17 x 16 B: odd slot stride, conflict-free b128)
constexpr int LDKH = BKV + 8; // bf16 tiles (mixed-precision mode): 144-byte rows
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDG = BK + 1;   // generic path: odd pad, ds_read_b32 fragments
constexpr int BKP = 64;       // wgrad vector path: pixels per step

// Activation tensors may live in HBM as bf16 (dpft_conv_desc.act16: mixed-precision storage of BASELINE.json configs[4]).
// A loader lane still owns 4 consecutive channels: 8 bytes instead of 16.  Offsets keep their fp32 (x 4 bytes) form and
// are halved at the load (the out-of-range marker 0x80000000 >> 1 still lies beyond any tensor), the 4 values are
// widened to fp32 registers -- everything downstream (BN + ReLU prologue, LDS formats, MFMA) is unchanged.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 v) {
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                 __uint_as_float(v[1] & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, bool half16) {
    if (half16)
        return widen_bf16x4(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff >> 1), soff >> 1, 0)));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, 0));
}
// the same 4 channels as their raw bf16 bits: written into lanes 0,1 of an existing register quad, lanes 2,3 are left
// alone (building a fresh {lo, hi, 0, 0} quad is a USE of the load and makes the compiler wait for it on the spot).
// Operands that need no arithmetic on the way into a bf16 LDS tile (data gradients, un-normalised inputs) are copied
// as they are, the others are widened when their register set is consumed.
__device__ __forceinline__ void ld4_raw16(f32x4& dst, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
    const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(voff >> 1), soff >> 1, 0));
    dst[0] = __uint_as_float(v[0]);
    dst[1] = __uint_as_float(v[1]);
}
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load4_act(const float* base, size_t off, bool half16) {      // off in elements
    if (half16) return __builtin_convertvector(*reinterpret_cast<const bf16x4s*>(reinterpret_cast<const __bf16*>(base) + off), f32x4);
    return *reinterpret_cast<const f32x4*>(base + off);
}
__device__ __forceinline__ void store4_act(float* base, size_t off, f32x4 v, bool half16) {
    if (half16) *reinterpret_cast<bf16x4s*>(reinterpret_cast<__bf16*>(base) + off) = __builtin_convertvector(v, bf16x4s);
    else *reinterpret_cast<f32x4*>(base + off) = v;
}

struct IgemmArgs {
    const float* x;
    const float* w;
    float* y;
    const float* bias;
    const float* pro;      // producer BN block [4][C] = mean, scale, beta, invstd (or null)
    float* stats;     // [mtiles][2][N] or null
    unsigned long long* bns;      // BatchNorm statistics as column sums [4][N] (common.h: BnSumsRef), added to by every tile; or null
    BnSumsRef pro_s;  // the prologue's BatchNorm given as column sums (.sums null: `pro` holds the BN block)
    float* partial;   // split-K: [splits][M][N] or null
    int* sk_ticket;   // split-K fix-up (round 4): one zeroed ticket per output tile; the workgroup that draws the last one
                      // sums the partial tiles in split order and runs the normal epilogue -- no reduction launch.  null = the
                      // caller reduces `partial` with a kernel of its own
    int B, H, W, C;   // A-source tensor
    int OH, OW, N;    // output tensor
    int kh, kw, stride, pad;
    int M, Ktot;
    int mtiles, ntiles, splits, ksteps, ksteps_per_split;
    int pro_relu;
    int accumulate;   // y += result (dgrad into an existing gradient)
    // residual variant of `accumulate`: y = result + (res_mask > 0 ? res_src : 0) -- the identity branch of a
    // bottleneck (dz = dout * (out > 0), src/.../resnet block) folded into the conv1 data gradient
    const float* res_src;
    const float* res_mask;
    const unsigned char* res_mask8;   // the same mask as one byte per 4 channels (bit e = element e passed the ReLU): 1/16 of the bytes
    // dgrad of a strided conv as one launch per output-pixel parity class (sub_step = stride > 1): rows enumerate the
    // sub_oh x sub_ow output pixels (oh, ow) = (sub_step*i + sub_ph, sub_step*j + sub_pw) and only the taps
    // r = sub_r0 + sub_step*k, s = sub_s0 + sub_step*l can hit them (all others fall between the dy samples).
    // A plain launch over all pixels and taps would spend stride^2 = 4x the MFMA work on structural zeros.
    int sub_step, sub_ph, sub_pw, sub_oh, sub_ow, sub_r0, sub_s0, sub_nr, sub_ns;
    int x16, y16;     // the A-source tensor / the output tensor (and res_src, res_mask, accumulate source) are bf16
    int w16;          // the weight operand is bf16 too (dpft_conv_desc.act16 = 2): LDS-DMA bf16 kernel
    // operands as three bf16 planes (conv_x3.hip, igemm_x3p_kernel): plane p of element e at base + 2 (p E + e) bytes; both or none
    const void* x3;
    const void* w3;
    // inference epilogue (dpft_conv2d_nhwc_fwd_bnact_f32): y = [relu](bn(result) [+ oadd]) with a BN block [4][N] of the OUTPUT
    // channels -- BatchNorm + ReLU + residual add without a separate elementwise pass and without an operand prologue
    // in the consumer (which applies them once per tap and column tile instead of once per element)
    const float* obn;
    const float* oadd;
    int orelu;
    int ablate;       // tuning aid (DPFT_ABLATE): 1 no global loads, 2 no LDS stores, 4 no epilogue, 8 no MFMAs
    int epf;          // host: the launch qualifies for the epilogue-operand prefetch (EpiPrefetch below)
    // Fused first pass of a BatchNorm backward (dpft::BnReduceFuse): the tensor this launch writes IS the `dout` of a
    // BatchNorm layer whose input y has the same shape; the epilogue adds sum(d) and sum(d * xhat) of its tile to
    // bnr_sums[2][N] (d = stored value under that layer's ReLU mask) -- the separate reduction pass over (y, dout), its
    // launch and its second read of dout disappear.
    const float* bnr_y;
    const float* bnr_bnp;              // BN block [4][N] of that layer (mean, scale, beta, invstd)
    const unsigned char* bnr_mask8;    // ReLU byte mask of the layer's OUTPUT side (1 byte per 4 channels), or null
    int bnr_self_mask;                 // mask = bn(y) > 0 (the ReLU sits directly behind this BatchNorm)
    float* bnr_sums;                   // null = no fused reduction
    // Fused BatchNorm FINALIZE of a train-mode forward conv (dpft::BnFinalFuse): instead of writing its (mean, M2) pair to
    // the per-tile statistics table, a tile adds n (mean - p) and M2 + n (mean - p)^2 to two accumulators per channel
    // (p = running mean: the pivot of bn_finalize_kernel's merge, same algebra); the workgroup that draws the last
    // ticket turns them into the BN block and updates the running statistics -- no bn_finalize launch between the conv
    // and its consumer.
    float* bnf_acc;                    // [2][N], zero before the launch; null = off
    int* bnf_ticket;                   // zero before the launch
    int bnf_slab;                      // deterministic form: per-tile statistics slab + one ticket per column tile (below)
    const float* bnf_gamma;
    const float* bnf_beta;
    float* bnf_rm;                     // running mean / var (may be null)
    float* bnf_rv;
    float* bnf_bnp;                    // out: BN block [4][N]
    float bnf_eps, bnf_mom;
};

// output row (GEMM row m) -> pixel index of the output tensor
__device__ __forceinline__ size_t out_pixel(const IgemmArgs& a, int m) {
    if (a.sub_step <= 1) return (size_t)m;
    const int per = a.sub_oh * a.sub_ow;
    const int b = m / per;
    const int rem = m - b * per;
    const int i = rem / a.sub_ow, j = rem - i * a.sub_ow;
    return ((size_t)b * a.OH + (size_t)i * a.sub_step + a.sub_ph) * a.OW + (size_t)j * a.sub_step + a.sub_pw;
}

// ---------------------------------------------------------------------------------------------
// Epilogue operands fetched DURING the main loop (igemm_pipe_kernel<..., EPF = true>, round 4).  The residual data gradient
// of a bottleneck's conv1 (dx = conv_dgrad + masked identity gradient, with the next BatchNorm's backward reduction in its
// epilogue) reads two more full-size tensors than it writes; all workgroups of a launch reach that epilogue together, so
// its loads found an idle matrix pipe and its main loop an idle memory system (layer 3, fp32: ~40 us of MFMAs, then ~15 us
// of streaming).  Here a thread's epilogue operands -- per 4-channel quad: the BatchNorm input `bnr_y`, the identity
// gradient `res_src` and their two ReLU mask bytes -- are requested together with the first K-step's tile and simply are in
// registers when the epilogue starts: one memory round trip instead of two per workgroup, and the loads no longer queue up
// behind 900 workgroups that finish their MFMAs at the same moment.  Host side: only for the form with both byte masks, unsplit.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// MODE 1: the residual form above (bnr_y, res_src, both mask bytes).  MODE 2: a plain data gradient that carries a BatchNorm-
// backward reduction (conv2 / conv3 of a bottleneck): bnr_y, and the mask byte where that layer's ReLU mask is a byte mask.
// MODE 3 (round 6, FORWARD): the inference epilogue y = relu(bn(conv) + residual) of a bottleneck's last conv
// (dpft_conv2d_nhwc_fwd_bnact_f32): the residual (IgemmArgs::oadd) is the prefetched operand, in g[].
template <int ITER, bool H16, int MODE_>
struct EpiPrefetch {
    static constexpr int MODE = MODE_;
    using Q = typename std::conditional<H16, u32x2_t, f32x4>::type;      // 4 channels as stored (bf16 / fp32), not widened:
    Q y[MODE_ == 3 ? 1 : ITER], g[(MODE_ == 1 || MODE_ == 3) ? ITER : 1];  // a conversion would be a use of the load
    unsigned mk[MODE_ == 3 ? 1 : ITER], rm[MODE_ == 1 ? ITER : 1];
};
template <bool H16> __device__ __forceinline__ f32x4 pf_widen(f32x4 v) { return v; }
template <bool H16> __device__ __forceinline__ f32x4 pf_widen(u32x2_t v) {
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                 __uint_as_float(v[1] & 0xffff0000u)};
}
template <typename PF> struct pf_mode { static constexpr int value = PF::MODE; };
template <> struct pf_mode<std::nullptr_t> { static constexpr int value = 0; };

// ---------------------------------------------------------------------------------------------
// shared epilogue: store accumulators (+bias), optional split-K partial, optional BN tile stats
// ---------------------------------------------------------------------------------------------
// NT = threads of the workgroup: 256, or 512 for the K-split form of the vector kernel whose waves 4..7 have already
// handed their accumulators to waves 0..3 -- they own no results here (`own`) but take part in the barriers and in the
// LDS -> global store loops.
template <int BM, int BN, int WGM, int WGN, int RB, int CB, int NT = 256, typename PF = std::nullptr_t>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x16 (&acc)[RB][CB], int m0,
                                               int n0, int mt, int split, float* smem, PF* pf = nullptr) {
    constexpr bool HAS_PF = !std::is_same<PF, std::nullptr_t>::value;      // operands already requested (EpiPrefetch)
    constexpr int PFM = pf_mode<PF>::value;                                // 0 | 1 residual form | 2 reduction only
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const bool own = NT == 256 || tid < 256;
    const int wm = wave / WGN, wn = wave % WGN;
    const int rbase = m0 + wm * RB * 32 + 4 * (lane >> 5);
    __syncthreads();  // LDS operand tiles are dead now
    // ---- split-K fix-up (IgemmArgs::sk_ticket) -------------------------------------------------------------------------
    // Every split workgroup writes its partial tile to the slab with agent-scope (write-through, sc1) stores, waits for
    // their acknowledgement and takes a ticket of its output tile; the last one reads all partial tiles back with
    // agent-scope loads, sums them in split order (the order of splitk_reduce_kernel: the result does not depend on which
    // workgroup ends up last), puts the sums back into its accumulator registers and falls through to the epilogue of an
    // unsplit launch -- bias, residual, accumulate, BatchNorm tile statistics, the fused BatchNorm-backward reduction.
    // No fences (a release fence writes the whole L2 of the XCD back), no spinning (nobody waits for anybody).
    // (HAS_PF: the host launches that kernel only unsplit, without statistics / bias / accumulation / inference epilogue (mode 1:
    // with both byte masks) -- the other forms are compiled out of it: half the registers and scalar state of the generic epilogue)
    const bool fix = !HAS_PF && a.partial != nullptr && a.sk_ticket != nullptr;
    if (fix) {
        constexpr int FLDC = BN + 4, FC4 = BN / 4, FITER = BM * FC4 / NT;
        float* Fs = smem;
        if (own)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * RB * 32 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    Fs[row * FLDC + wn * CB * 32 + cb * 32 + (lane & 31)] = acc[rb][cb][r];
                }
        __syncthreads();
        const unsigned slab_bytes = (unsigned)a.M * (unsigned)a.N * 4u;
        __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.partial, 0, (int)(slab_bytes * (unsigned)a.splits), 0x00020000);
#pragma unroll
        for (int it = 0; it < FITER; ++it) {
            const int idx = tid + it * NT;
            const int row = idx / FC4, c4 = idx - row * FC4;
            if (m0 + row < a.M && n0 + c4 * 4 < a.N) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&Fs[row * FLDC + c4 * 4]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs,
                                                       (int)(((unsigned)(m0 + row) * (unsigned)a.N + n0 + c4 * 4) * 4u),
                                                       (int)(slab_bytes * (unsigned)split), 16);      // aux 16 = sc1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        int* ticket = a.sk_ticket + mt * a.ntiles + n0 / BN;
        if (tid == 0) *flag = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const bool is_last = *flag == a.splits - 1;
        __syncthreads();      // everybody has read the flag before the sums overwrite it
        if (!is_last) return;
        // (formal release / acquire fences around the ticket -- buffer_wbl2 / buffer_inv -- were tried when a model-level test
        // moved: results identical to the last digit, i.e. the sc1 stores / loads already give the ordering)
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left clean for the next launch
        constexpr int FCH = FITER < 8 ? FITER : 8;      // tile chunks in flight per thread
        static_assert(FITER % FCH == 0, "fix-up chunking");
#pragma unroll 1
        for (int it0 = 0; it0 < FITER; it0 += FCH) {
            f32x4 sum[FCH];
            int voff[FCH];
#pragma unroll
            for (int u = 0; u < FCH; ++u) {
                const int idx = tid + (it0 + u) * NT;
                const int row = idx / FC4, c4 = idx - row * FC4;
                const bool ok = m0 + row < a.M && n0 + c4 * 4 < a.N;
                voff[u] = ok ? (int)(((unsigned)(m0 + row) * (unsigned)a.N + n0 + c4 * 4) * 4u) : -1;      // -1: out of range -> 0
                sum[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, voff[u], 0, 16));
            }
            for (int k = 1; k < a.splits; ++k) {
                f32x4 t[FCH];
#pragma unroll
                for (int u = 0; u < FCH; ++u)
                    t[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, voff[u], (int)(slab_bytes * (unsigned)k), 16));
#pragma unroll
                for (int u = 0; u < FCH; ++u) sum[u] += t[u];
            }
#pragma unroll
            for (int u = 0; u < FCH; ++u) {
                const int idx = tid + (it0 + u) * NT;
                const int row = idx / FC4, c4 = idx - row * FC4;
                *reinterpret_cast<f32x4*>(&Fs[row * FLDC + c4 * 4]) = sum[u];
            }
        }
        __syncthreads();
        if (own)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * RB * 32 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    acc[rb][cb][r] = Fs[row * FLDC + wn * CB * 32 + cb * 32 + (lane & 31)];
                }
        __syncthreads();      // the statistics / staging below reuse the same LDS
    }
    const bool part = !HAS_PF && a.partial != nullptr && !fix;      // this launch leaves partial tiles for a reduction kernel
    if (!HAS_PF && (a.stats != nullptr || a.bnf_acc != nullptr || a.bns != nullptr)) {
        // ---- per-tile column statistics of the raw conv output (bias-free by construction) ----
        float* red = smem;               // [WGM][BN]
        float* smean = smem + WGM * BN;  // [BN]
        const int cnt = min(BM, a.M - m0);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float s = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + rb * 32 + (r & 3) + 8 * (r >> 2);
                    s += (row < a.M) ? acc[rb][cb][r] : 0.f;
                }
            s += __shfl_xor(s, 32);
            if (own && lane < 32) red[wm * BN + wn * CB * 32 + cb * 32 + lane] = s;
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < WGM; ++i) s += red[i * BN + tid];
            const float mean = s / (float)cnt;
            smean[tid] = mean;
            if (a.stats && n0 + tid < a.N) {
                float* dst = a.stats + ((size_t)mt * 2 + 0) * a.N + n0 + tid;
                if (a.bnf_slab) __hip_atomic_store(dst, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // read by ANOTHER workgroup of this launch
                else *dst = mean;
            }
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float mean = smean[wn * CB * 32 + cb * 32 + (lane & 31)];
            float s = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + rb * 32 + (r & 3) + 8 * (r >> 2);
                    const float d = acc[rb][cb][r] - mean;
                    s += (row < a.M) ? d * d : 0.f;
                }
            s += __shfl_xor(s, 32);
            if (own && lane < 32) red[wm * BN + wn * CB * 32 + cb * 32 + lane] = s;  // red is free: barrier above
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < WGM; ++i) s += red[i * BN + tid];
            if (a.stats) {
                float* dst = a.stats + ((size_t)mt * 2 + 1) * a.N + n0 + tid;
                if (a.bnf_slab) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *dst = s;
            }
            if (a.bns) {      // "sums" form: n * mean and M2 + n * mean^2 (fp64: exact products) into the fixed-point accumulators
                const double m = (double)smean[tid], fc = (double)cnt;
                bn_sums_add(a.bns, a.N, n0 + tid, fc * m, (double)s + fc * m * m);
            }
            if (a.bnf_acc) {      // fused finalize: this tile's share of the pivoted sums (device-scope atomics)
                const int n = n0 + tid;
                const float dlt = smean[tid] - (a.bnf_rm ? a.bnf_rm[n] : 0.f);
                const float fc = (float)cnt;
                atomicAdd(a.bnf_acc + n, fc * dlt);
                atomicAdd(a.bnf_acc + a.N + n, fmaf(fc * dlt, dlt, s));
            }
        }
        __syncthreads();
    }
    // ---- stage the tile through LDS so that global stores are full 16-byte lanes along rows --------
    // (per-register scalar stores are store-issue bound and, with vmcnt counting stores, serialise)
    constexpr int LDC = BN + 4;
    float* Cs = smem;  // [BM][LDC]
    if (own)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * RB * 32 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                Cs[row * LDC + wn * CB * 32 + cb * 32 + (lane & 31)] = acc[rb][cb][r];
            }
    __syncthreads();
    float* __restrict__ out = part ? a.partial + (size_t)split * a.M * a.N : a.y;
    const bool add_bias = !HAS_PF && (a.bias != nullptr) && !part;
    const bool accum = !HAS_PF && a.accumulate && !part;
    const bool y16 = a.y16 && !part;      // split-K partials stay fp32
    if (HAS_PF || (a.N & 3) == 0) {
        constexpr int C4 = BN / 4;
        constexpr int ITER = BM * C4 / NT;
        static_assert(BM * C4 % NT == 0, "tile / thread count");
        f32x4 old[ITER];
        // operands of the fused BatchNorm-backward reduction: requested here, consumed in the store loop below
        const bool bnr_pre = (HAS_PF && PFM != 3) || (!HAS_PF && (a.bnr_sums != nullptr) && !part);
        f32x4 bnr_yv[ITER];
        unsigned bnr_mk[ITER];
        if (bnr_pre && HAS_PF) {
            if constexpr (HAS_PF && PFM != 3) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    bnr_yv[it] = pf_widen<true>(pf->y[it]);
                    bnr_mk[it] = (PFM == 1 || a.bnr_mask8) ? pf->mk[it] : 0u;
                }
            }
        } else if (bnr_pre) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                bnr_yv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                bnr_mk[it] = 0u;
                if (m0 + row < a.M && n0 + c4 * 4 < a.N) {
                    const size_t off = out_pixel(a, m0 + row) * a.N + n0 + c4 * 4;
                    bnr_yv[it] = load4_act(a.bnr_y, off, y16);      // that layer's conv output: bf16 where this one's is
                    if (a.bnr_mask8) bnr_mk[it] = a.bnr_mask8[off >> 2];
                }
            }
        }
        const bool resid = PFM == 1 || (PFM == 0 && (a.res_src != nullptr) && !part);
        const bool obn = PFM == 3 || (!HAS_PF && (a.obn != nullptr) && !part);
        const bool oadd = PFM == 3 || (obn && a.oadd != nullptr);
        if (PFM == 3) {
            if constexpr (PFM == 3) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) old[it] = pf_widen<true>(pf->g[it]);
            }
        } else if (PFM == 1) {
            if constexpr (PFM == 1) {
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const f32x4 g = pf_widen<true>(pf->g[it]);
                    const unsigned mk = pf->rm[it];
#pragma unroll
                    for (int e = 0; e < 4; ++e) old[it][e] = ((mk >> e) & 1u) ? g[e] : 0.f;
                }
            }
        } else if (accum || resid || oadd) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                old[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m0 + row < a.M && n0 + c4 * 4 < a.N) {
                    const size_t off = out_pixel(a, m0 + row) * a.N + n0 + c4 * 4;
                    if (oadd) {
                        old[it] = load4_act(a.oadd, off, y16);
                    } else if (resid) {
                        const f32x4 g = load4_act(a.res_src, off, y16);
                        if (a.res_mask8) {
                            const unsigned mk = a.res_mask8[off >> 2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) old[it][e] = ((mk >> e) & 1u) ? g[e] : 0.f;
                        } else {
                            const f32x4 o = load4_act(a.res_mask, off, y16);
#pragma unroll
                            for (int e = 0; e < 4; ++e) old[it][e] = o[e] > 0.f ? g[e] : 0.f;
                        }
                    } else {
                        old[it] = load4_act(out, off, y16);
                    }
                }
            }
        }
        // fused BatchNorm-backward reduction (see IgemmArgs::bnr_*): a thread owns ONE 4-channel chunk (NT % C4 == 0)
        static_assert(NT % C4 == 0, "a thread's channel chunk must not depend on the pass");
        const bool bnr = (HAS_PF && PFM != 3) || (!HAS_PF && (a.bnr_sums != nullptr) && !part);
        f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = {0.f, 0.f, 0.f, 0.f}, bmu = bs0, bis = bs0, bsc = bs0, bbe = bs0;
        const int bc = n0 + (tid % C4) * 4;
        if (bnr && bc < a.N) {
            bmu = *reinterpret_cast<const f32x4*>(a.bnr_bnp + bc);
            bis = *reinterpret_cast<const f32x4*>(a.bnr_bnp + 3 * a.N + bc);
            if (PFM != 1 && a.bnr_self_mask) {
                bsc = *reinterpret_cast<const f32x4*>(a.bnr_bnp + a.N + bc);
                bbe = *reinterpret_cast<const f32x4*>(a.bnr_bnp + 2 * a.N + bc);
            }
        }
        auto store_all = [&](auto H16) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                if (m0 + row < a.M && n0 + c4 * 4 < a.N) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
                    if (add_bias) v += *reinterpret_cast<const f32x4*>(a.bias + n0 + c4 * 4);
                    if (obn) {      // the expression of bn_apply4 (bn.hip): (v - mean) * scale + beta
                        const int n = n0 + c4 * 4;
                        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.obn + n);
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.obn + a.N + n);
                        const f32x4 be = *reinterpret_cast<const f32x4*>(a.obn + 2 * a.N + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e] - mu[e], sc[e], be[e]);
                    }
                    if (accum || resid || oadd) v += old[it];
                    if (obn && a.orelu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    const size_t ooff = out_pixel(a, m0 + row) * a.N + n0 + c4 * 4;
                    store4_act(out, ooff, v, decltype(H16)::value);
                    if (bnr) {
                        const f32x4 yv = bnr_yv[it];
                        f32x4 d = v;
                        // bf16 storage: reduce what the stand-alone pass would read back, i.e. the rounded value
                        if (decltype(H16)::value) d = __builtin_convertvector(__builtin_convertvector(v, bf16x4s), f32x4);
                        if (PFM == 1 || a.bnr_mask8) {
                            const unsigned mk = bnr_mk[it];
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = ((mk >> e) & 1u) ? d[e] : 0.f;
                        } else if (a.bnr_self_mask) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = fmaf(yv[e] - bmu[e], bsc[e], bbe[e]) > 0.f ? d[e] : 0.f;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bs0[e] += d[e];
                            bs1[e] += d[e] * ((yv[e] - bmu[e]) * bis[e]);
                        }
                    }
                }
            }
        };
        if (y16) store_all(std::true_type{});
        else store_all(std::false_type{});
        if (bnr) {      // column sums over the tile's rows (NT / C4 threads per chunk), one atomic pair per column
            __syncthreads();      // the staged tile has been read
            float* red = smem;      // [NT][8]: s0[4], s1[4] of each thread
            *reinterpret_cast<f32x4*>(red + tid * 8) = bs0;
            *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bs1;
            __syncthreads();
            if (tid < BN && n0 + tid < a.N) {
                const int ch = tid >> 2, e = tid & 3;
                float t0 = 0.f, t1 = 0.f;
                for (int g = 0; g < NT / C4; ++g) {
                    t0 += red[(g * C4 + ch) * 8 + e];
                    t1 += red[(g * C4 + ch) * 8 + 4 + e];
                }
                atomicAdd(a.bnr_sums + n0 + tid, t0);
                atomicAdd(a.bnr_sums + a.N + n0 + tid, t1);
            }
        }
    } else {
        for (int idx = tid; idx < BM * BN; idx += NT) {
            const int row = idx / BN, c = idx - row * BN;
            if (m0 + row < a.M && n0 + c < a.N) {
                float v = Cs[row * LDC + c];
                if (add_bias) v += a.bias[n0 + c];
                const size_t off = out_pixel(a, m0 + row) * a.N + n0 + c;
                float* o = out + off;
                if (a.res_src != nullptr && !part) v += a.res_mask[off] > 0.f ? a.res_src[off] : 0.f;
                else if (accum) v += *o;
                *o = v;
            }
        }
    }
    if (!HAS_PF && a.bnf_slab) {
        // Deterministic BatchNorm finalize inside the forward conv: the workgroups of one COLUMN tile take tickets; whoever
        // draws the last one merges that tile's columns of the statistics slab with the arithmetic of bn_finalize_kernel
        // (bn.hip: 32 tile groups per channel, pivot = tile 0, groups summed in order) -- bit-identical to the separate
        // launch, whichever workgroup ends up doing it.  No fences: the slab entries were written with agent-scope stores
        // (performed at the coherence point once vmcnt acknowledges them), the tickets are agent-scope RMWs, the merger reads
        // the slab with agent-scope loads; its plain stores of the BN block are consumed by LATER kernels.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.bnf_ticket + n0 / BN, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag == a.mtiles - 1) {
            constexpr int R = NT / BN;      // threads per channel
            float* s1s = smem + 64;         // [32][BN]
            float* s2s = s1s + 32 * BN;
            const int cl = tid % BN, r = tid / BN, c = n0 + cl;
            const bool ok = c < a.N;
            auto ld = [&](size_t i) { return __hip_atomic_load(a.stats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
            const float pivot = ok ? ld(c) : 0.f;
            const float full = (float)BM;
            const float last = (float)((int64_t)a.M - (int64_t)(a.mtiles - 1) * BM);
            // all of this thread's slab entries are requested before the first is used (atomic loads are not hoisted or
            // pipelined by the compiler: one at a time they cost 57 memory round trips, ~14 us at the end of the conv)
            constexpr int GP = 32 / R, TPG = 2;      // groups per thread, tiles per group (launches of <= 64 row tiles)
            float mv[GP][TPG], qv[GP][TPG];
#pragma unroll
            for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                for (int ti = 0; ti < TPG; ++ti) {
                    const int t = r + gi * R + ti * 32;
                    const bool on = ok && t < a.mtiles;
                    mv[gi][ti] = on ? ld(((size_t)t * 2 + 0) * a.N + c) : 0.f;
                    qv[gi][ti] = on ? ld(((size_t)t * 2 + 1) * a.N + c) : 0.f;
                }
#pragma unroll
            for (int gi = 0; gi < GP; ++gi) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int ti = 0; ti < TPG; ++ti) {
                    const int t = r + gi * R + ti * 32;
                    if (ok && t < a.mtiles) {
                        const float cnt = t == a.mtiles - 1 ? last : full;
                        const float d = mv[gi][ti] - pivot;
                        s1 = fmaf(cnt, d, s1);
                        s2 += fmaf(cnt * d, d, qv[gi][ti]);
                    }
                }
                s1s[(r + gi * R) * BN + cl] = s1;
                s2s[(r + gi * R) * BN + cl] = s2;
            }
            __syncthreads();
            if (tid < BN && ok) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) { s1 += s1s[i * BN + cl]; s2 += s2s[i * BN + cl]; }
                const float invN = 1.0f / (float)a.M;
                const float mean = fmaf(s1, invN, pivot);
                const float m2 = fmaxf(s2 - s1 * s1 * invN, 0.f);
                const float var = m2 * invN;
                const float invstd = 1.0f / sqrtf(var + a.bnf_eps);
                a.bnf_bnp[c] = mean;
                a.bnf_bnp[a.N + c] = a.bnf_gamma[c] * invstd;
                a.bnf_bnp[2 * a.N + c] = a.bnf_beta[c];
                a.bnf_bnp[3 * a.N + c] = invstd;
                if (a.bnf_rm) {
                    const float unbiased = a.M > 1 ? m2 / (float)(a.M - 1) : var;
                    a.bnf_rm[c] = (1.f - a.bnf_mom) * a.bnf_rm[c] + a.bnf_mom * mean;
                    a.bnf_rv[c] = (1.f - a.bnf_mom) * a.bnf_rv[c] + a.bnf_mom * unbiased;
                }
            }
        }
    }
    if (!HAS_PF && a.bnf_acc != nullptr) {
        // Every tile has added its sums with device-scope atomics; an atomic is acknowledged (vmcnt) once it has been
        // performed at the coherence point, so "wait for mine, then take a ticket" orders them before the last ticket --
        // no release fence (nothing here publishes plain stores), the last workgroup reads the totals with device-scope
        // atomic loads.  Its plain stores of the BN block are consumed by LATER kernels.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) *flag = __hip_atomic_fetch_add(a.bnf_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag == a.mtiles * a.ntiles - 1) {
            const float invN = 1.0f / (float)a.M;
            for (int c = tid; c < a.N; c += NT) {
                const float s1 = __hip_atomic_load(a.bnf_acc + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float s2 = __hip_atomic_load(a.bnf_acc + a.N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float pivot = a.bnf_rm ? a.bnf_rm[c] : 0.f;
                const float mean = fmaf(s1, invN, pivot);
                const float m2 = fmaxf(s2 - s1 * s1 * invN, 0.f);
                const float var = m2 * invN;
                const float invstd = 1.0f / sqrtf(var + a.bnf_eps);
                a.bnf_bnp[c] = mean;
                a.bnf_bnp[a.N + c] = a.bnf_gamma[c] * invstd;
                a.bnf_bnp[2 * a.N + c] = a.bnf_beta[c];
                a.bnf_bnp[3 * a.N + c] = invstd;
                if (a.bnf_rm) {
                    const float unbiased = a.M > 1 ? m2 / (float)(a.M - 1) : var;
                    a.bnf_rm[c] = (1.f - a.bnf_mom) * pivot + a.bnf_mom * mean;
                    a.bnf_rv[c] = (1.f - a.bnf_mom) * a.bnf_rv[c] + a.bnf_mom * unbiased;
                }
            }
        }
    }
}

__device__ __forceinline__ void decode_tile(const IgemmArgs& a, int& mt, int& nt, int& split) {
    const int nwg = a.mtiles * a.ntiles * a.splits;
    int bid = xcd_remap(blockIdx.x, nwg);
    const int per = a.mtiles * a.ntiles;
    split = bid / per;
    bid -= split * per;
    mt = bid / a.ntiles;
    nt = bid - mt * a.ntiles;
}

template <typename Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct WgradArgs {
    const float* x;
    const float* dy;
    float* dw;        // [K][taps][C]
    float* partial;   // [splits][K*taps*C] or null
    const float* pro; // BN block [4][C] of the x operand or null
    int B, H, W, C;   // input
    int OH, OW, K;    // dy
    int kh, kw, stride, pad;
    int M;            // B*OH*OW pixels (reduction)
    int ktiles, ctiles, taps, splits, psteps, psteps_per_split;
    int pro_relu;
    int J;            // generic path: taps*C
    int x16, dy16;    // x / dy stored as bf16 (vector path only)
};

// conv_x3.hip: the 3 x bf16 split main loop (fp32 results from the bf16 matrix cores); `a` prepared as for igemm_pipe_kernel
int launch_igemm_x3(IgemmArgs& a, int bm, int bn, bool dgrad, bool pro, hipStream_t st);
// conv_x3w.hip: the same arithmetic on eight waves per 128 x 128 tile (`a` as launch_igemm_x3 prepared it)
int launch_igemm_x3w(IgemmArgs& a, bool dgrad, bool pro, hipStream_t st);
int split_planes(const float* src, void* dst, int64_t n, hipStream_t st);
// conv_stream.hip: 1x1 convs with 64 / 128 input channels and K % 256 == 0 on large maps (weights in LDS, rows private to a wave)
bool stream1x1_match(const dpft_conv_desc* d, int* tile_rows);
int launch_stream1x1(const dpft_conv_desc* d, const float* x, const float* w, const float* pro_bn, float* y, float* stats,
                     const float* out_bn, const float* residual, int relu, hipStream_t st, unsigned long long* bns = nullptr,
                     const BnSumsRef* pro_sums = nullptr);
// conv_b16w.hip: bf16 operands on 256-row tiles, eight waves (act16 = 2); `a` prepared as for igemm_pipe_kernel<bm, bn, .., B16>
int launch_igemm_b16w(IgemmArgs& a, int bm, int bn, bool dgrad, hipStream_t st);      // fp32 -> three bf16 planes
// weight gradient on the split kernels: 128 x 128 tiles, 32 pixels per step; `a` as for wgrad_pipe_kernel<128, 128, 2, 2, 32, *>
int launch_wgrad_x3(const WgradArgs& a, bool pro, dim3 grid, hipStream_t st);

}  // namespace dpft
