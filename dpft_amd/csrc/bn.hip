// BatchNorm / ReLU / residual / max-pool / FPN glue kernels on NHWC fp32 (HBM-bound, float4 lanes).
// Replaces the ATen/cuDNN elementwise + reduction kernels behind torchvision's ResNet body and FPN
// (reference call sites: src/dprt/models/backbones/resnet.py:54-55, src/dprt/models/necks/fpn.py:39-43)
// and the in-place sinusoidal embedding add (src/dprt/models/embeddings/sinusoidal.py:107-108).
#include "common.h"
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace dpft {

// ---------------------------------------------------------------------------------------------
// per-tile (mean, M2) of y[M][K]; tile = tile_rows consecutive rows. block = one tile x 256-col slab
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ stats,
                                                        int64_t M, int K, int tile_rows) {
    __shared__ float red[256];
    __shared__ float smean[256];
    const int tile = blockIdx.x;
    const int kc = min(K, 256);
    const int groups = 256 / kc;
    const int c_local = threadIdx.x % kc, g = threadIdx.x / kc;
    const int c = blockIdx.y * 256 + c_local;
    const int64_t r0 = (int64_t)tile * tile_rows;
    const int cnt = (int)min((int64_t)tile_rows, M - r0);
    const bool act = g < groups && c < K;
    float s = 0.f;
    if (act)
        for (int r = g; r < cnt; r += groups) s += y[(r0 + r) * K + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < kc) {
        float t = 0.f;
        for (int i = 0; i < groups; ++i) t += red[threadIdx.x + i * kc];
        smean[threadIdx.x] = t / (float)cnt;
    }
    __syncthreads();
    const float mean = smean[c_local];
    s = 0.f;
    if (act)
        for (int r = g; r < cnt; r += groups) {
            const float d = y[(r0 + r) * K + c] - mean;
            s += d * d;
        }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < kc && c < K) {
        float t = 0.f;
        for (int i = 0; i < groups; ++i) t += red[threadIdx.x + i * kc];
        stats[((size_t)tile * 2 + 0) * K + c] = mean;
        stats[((size_t)tile * 2 + 1) * K + c] = t;
    }
}

// One block = 32 channels x 32 tile groups (1024 threads): this kernel sits between every conv and its consumer on the
// forward's critical path, so what matters is its latency chain, not its throughput.  The merge of the per-tile
// (count, mean, M2) triples is written as three plain sums around a pivot p = mean of tile 0,
//   S1 = sum_t n_t (mean_t - p),  S2 = sum_t M2_t + n_t (mean_t - p)^2  ->  mean = p + S1 / N,  M2 = S2 - S1^2 / N,
// (exact algebra of Chan's merge; the pivot keeps S1^2 / N ~ sigma^2 / tile_rows of S2, so nothing cancels even for the
// raw 0..255 inputs of the stem) -- independent loads, no division chain: 8.6 -> ~4 us per BatchNorm layer.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stats, int tiles, int tile_rows, int64_t M,
                                   int K, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   float* bnp) {
    DPFT_SETPRIO_BN();
    __shared__ float s1s[32][33], s2s[32][33];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const bool ok = c < K;
    const float pivot = ok ? stats[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        const float full = (float)tile_rows;
        const float last = (float)(M - (int64_t)(tiles - 1) * tile_rows);
#pragma unroll 4
        for (int t = g; t < tiles; t += 32) {
            const float cnt = t == tiles - 1 ? last : full;
            const float d = stats[((size_t)t * 2 + 0) * K + c] - pivot;
            const float m2t = stats[((size_t)t * 2 + 1) * K + c];
            s1 = fmaf(cnt, d, s1);
            s2 += fmaf(cnt * d, d, m2t);
        }
    }
    s1s[g][cl] = s1; s2s[g][cl] = s2;
    __syncthreads();
    if (g != 0 || !ok) return;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { s1 += s1s[i][cl]; s2 += s2s[i][cl]; }
    const float invN = 1.0f / (float)M;
    const float mean = fmaf(s1, invN, pivot);
    const float m2 = fmaxf(s2 - s1 * s1 * invN, 0.f);
    const float var = m2 * invN;
    const float invstd = 1.0f / sqrtf(var + eps);
    bnp[c] = mean;
    bnp[K + c] = gamma[c] * invstd;
    bnp[2 * K + c] = beta[c];
    bnp[3 * K + c] = invstd;
    if (running_mean) {
        const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, int K,
                               float* bnp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    bnp[c] = rm[c];
    bnp[K + c] = gamma[c] * invstd;
    bnp[2 * K + c] = beta[c];
    bnp[3 * K + c] = invstd;
}

// up to 16 eval-mode BN blocks per launch (the launch plan computes all of a network's blocks up front: they do not
// depend on activations, and 100+ single-layer launches cost ~0.5 ms of a 13 ms inference forward)
__global__ void bn_eval_multi_kernel(BnEvalBatch a) {
    const int i = blockIdx.y;
    const int K = a.K[i];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    const float invstd = 1.0f / sqrtf(a.rv[i][c] + a.eps);
    float* bnp = a.out[i];
    bnp[c] = a.rm[i][c];
    bnp[K + c] = a.gamma[i][c] * invstd;
    bnp[2 * K + c] = a.beta[i][c];
    bnp[3 * K + c] = invstd;
}

// Activation storage: float (default) or __bf16 (mixed-precision storage, dpft_conv_desc.act16): a lane handles 4
// consecutive channels = one 16-byte or one 8-byte access; all arithmetic is fp32 either way.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ldv4(const void* base, int64_t i4);
template <> __device__ __forceinline__ f32x4 ldv4<float>(const void* base, int64_t i4) { return reinterpret_cast<const f32x4*>(base)[i4]; }
template <> __device__ __forceinline__ f32x4 ldv4<__bf16>(const void* base, int64_t i4) {
    return __builtin_convertvector(reinterpret_cast<const bf16x4_t*>(base)[i4], f32x4);
}
template <typename T> __device__ __forceinline__ void stv4(void* base, int64_t i4, f32x4 v);
template <> __device__ __forceinline__ void stv4<float>(void* base, int64_t i4, f32x4 v) { reinterpret_cast<f32x4*>(base)[i4] = v; }
template <> __device__ __forceinline__ void stv4<__bf16>(void* base, int64_t i4, f32x4 v) {
    reinterpret_cast<bf16x4_t*>(base)[i4] = __builtin_convertvector(v, bf16x4_t);
}

// BN block convention: bnp[4][K] = (mean, scale = gamma*invstd, beta, invstd); bn(y) = (y-mean)*scale+beta
__device__ __forceinline__ f32x4 bn_apply4(f32x4 v, const float* __restrict__ bnp, int K, int c) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnp + K + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(bnp + 2 * K + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e] - mu[e], sc[e], be[e]);
    return v;
}

// out = [relu](bn(y) [+ bn_r(res) | + res]); K % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                      const float* __restrict__ res, const float* __restrict__ rbnp,
                                                      int relu, float* __restrict__ out, float* __restrict__ out32,
                                                      int64_t n4, int K4, unsigned char* __restrict__ mask8) {
    DPFT_SETPRIO_BN();
    const int K = K4 * 4;
    // bf16 storage moves 8 bytes per lane and access: two independent groups per trip keep as many bytes in flight as
    // the fp32 form (these passes are pure HBM streaming)
    constexpr int U = std::is_same<T, float>::value ? 1 : 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
        f32x4 yv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                yv[u] = ldv4<T>(y, i);
                if (res) rv[u] = ldv4<T>(res, i);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n4) break;
            const int c = (int)(i % K4) * 4;
            f32x4 v = bn_apply4(yv[u], bnp, K, c);
            if (res) {
                f32x4 r = rv[u];
                if (rbnp) r = bn_apply4(r, rbnp, K, c);
                v += r;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            stv4<T>(out, i, v);
            if (out32) reinterpret_cast<f32x4*>(out32)[i] = v;      // fp32 copy of a stage output (consumers outside the plan)
            if (mask8)                                               // which elements passed the ReLU: all the backward needs of `out`
                mask8[i] = (unsigned char)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
        }
    }
}

// bn_act_kernel<float> for launches whose grid stride is a multiple of K / 4 (round 6, as bn_bwd_apply_fixc_kernel): the
// thread's parameter quads are loaded once, two element quads per tensor in flight.  Same expression per element.
// SUMS: either BatchNorm may come as fp64 column sums (common.h: BnSumsRef; .sums null = its BN block) -- the thread derives its
// four channels' parameters itself, once.
template <int U, bool SUMS = false>
__global__ __launch_bounds__(256) void bn_act_fixc_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                           const float* __restrict__ res, const float* __restrict__ rbnp_,
                                                           int relu, float* __restrict__ out, int64_t n4, int K4,
                                                           unsigned char* __restrict__ mask8, BnSumsRef ys, BnSumsRef rs) {
    DPFT_SETPRIO_BN();
    const int K = K4 * 4;
    const unsigned stride = gridDim.x * blockDim.x;      // % K4 == 0, n4 < 2^30 (host)
    const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(first % (unsigned)K4) * 4;
    // K4 < 256: the block's 256 threads meet only K4 distinct channel quads -- thread j < K4 derives quad (first_of_block + j) % K4
    // into LDS, everybody reads slot tid % K4 (the fp64 derivation is ~40 instructions per channel; a small-K pass over a large
    // map would otherwise repeat it 256 / K4 times per block)
    extern __shared__ float sums_tab[];      // [2][3][K] when shared (host: 24 K bytes), else nothing
    const bool share = SUMS && K4 < 256;
    const int slot = share ? ((int)threadIdx.x % K4) * 4 : 0;
    auto derive = [&](const BnSumsRef& r, int which, f32x4& o_mu, f32x4& o_sc, f32x4& o_be) {
        if (share) {
            if ((int)threadIdx.x < K4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float m_, s_, b_, is_;
                    bn_from_sums(r, c + e, m_, s_, b_, is_);
                    sums_tab[(which * 3 + 0) * K + slot + e] = m_; sums_tab[(which * 3 + 1) * K + slot + e] = s_; sums_tab[(which * 3 + 2) * K + slot + e] = b_;
                }
            }
            __syncthreads();
            o_mu = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 0) * K + slot);
            o_sc = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 1) * K + slot);
            o_be = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 2) * K + slot);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m_, s_, b_, is_;
                bn_from_sums(r, c + e, m_, s_, b_, is_);
                o_mu[e] = m_; o_sc[e] = s_; o_be[e] = b_;
            }
        }
    };
    f32x4 mu, sc, be;
    if (SUMS && ys.sums) {
        derive(ys, 0, mu, sc, be);
    } else {
        mu = *reinterpret_cast<const f32x4*>(bnp + c);
        sc = *reinterpret_cast<const f32x4*>(bnp + K + c);
        be = *reinterpret_cast<const f32x4*>(bnp + 2 * K + c);
    }
    f32x4 rmu = mu, rsc = sc, rbe = be;
    const bool rbnp = rbnp_ != nullptr || (SUMS && rs.sums != nullptr);      // the residual has a BatchNorm of its own (downsample branch)
    if (SUMS && rs.sums) {
        derive(rs, 1, rmu, rsc, rbe);
    } else if (rbnp_) {
        rmu = *reinterpret_cast<const f32x4*>(rbnp_ + c);
        rsc = *reinterpret_cast<const f32x4*>(rbnp_ + K + c);
        rbe = *reinterpret_cast<const f32x4*>(rbnp_ + 2 * K + c);
    }
    const unsigned n = (unsigned)n4;
    for (unsigned i0 = first; i0 < n; i0 += U * stride) {
        f32x4 yv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            if (i < n) {
                yv[u] = reinterpret_cast<const f32x4*>(y)[i];
                if (res) rv[u] = reinterpret_cast<const f32x4*>(res)[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            if (i >= n) break;
            f32x4 v = yv[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e] - mu[e], sc[e], be[e]);
            if (res) {
                f32x4 r = rv[u];
                if (rbnp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = fmaf(r[e] - rmu[e], rsc[e], rbe[e]);
                }
                v += r;
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            reinterpret_cast<f32x4*>(out)[i] = v;
            if (mask8) mask8[i] = (unsigned char)((v[0] > 0.f ? 1 : 0) | (v[1] > 0.f ? 2 : 0) | (v[2] > 0.f ? 4 : 0) | (v[3] > 0.f ? 8 : 0));
        }
    }
}

// bf16 storage with 16-byte accesses (round 4): a lane handles 8 consecutive channels per access (the 4-channel form above
// moves 8 bytes per lane and access; these passes are pure HBM streaming and ran at 2.8 TB/s), two accesses per tensor in
// flight.  K % 8 == 0; same arithmetic per element as bn_act_kernel<__bf16>.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void widen8(u32x4_t v, f32x4& lo, f32x4& hi) {
    lo = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
    hi = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xffff0000u), __uint_as_float(v[3] << 16), __uint_as_float(v[3] & 0xffff0000u)};
}
__device__ __forceinline__ u32x4_t narrow8(f32x4 lo, f32x4 hi) {
    const bf16x4_t a = __builtin_convertvector(lo, bf16x4_t), b = __builtin_convertvector(hi, bf16x4_t);      // RNE, as stv4<__bf16>
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const u32x2v ua = __builtin_bit_cast(u32x2v, a), ub = __builtin_bit_cast(u32x2v, b);
    return u32x4_t{ua[0], ua[1], ub[0], ub[1]};
}
__device__ __forceinline__ unsigned relu_bits(f32x4 v) {
    return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}
// FIXC (round 6): the grid stride is a multiple of K / 8 -- a thread meets the same 8 channels in every trip and loads their
// parameters once (as bn_act_fixc_kernel)
// SUMS (FIXC only): either BatchNorm may come as column sums (common.h: BnSumsRef), as in bn_act_fixc_kernel
template <bool FIXC, bool SUMS = false>
__global__ __launch_bounds__(256) void bn_act16_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                        const float* __restrict__ res, const float* __restrict__ rbnp_,
                                                        int relu, float* __restrict__ out, float* __restrict__ out32,
                                                        int64_t n8, int K8, unsigned char* __restrict__ mask8, BnSumsRef ys, BnSumsRef rs) {
    static_assert(FIXC || !SUMS, "column sums: fixed-channel form only");
    const bool rbnp_on = rbnp_ != nullptr || (SUMS && rs.sums != nullptr);
    const float* const rbnp = rbnp_on ? (rbnp_ ? rbnp_ : bnp) : nullptr;      // (non-null = the residual has a BatchNorm; FIXC reads the registers)
    DPFT_SETPRIO_BN();
    const int K = K8 * 8;
    constexpr int U = 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const u32x4_t* __restrict__ y8 = reinterpret_cast<const u32x4_t*>(y);
    const u32x4_t* __restrict__ r8 = reinterpret_cast<const u32x4_t*>(res);
    const int cfix = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % K8) * 8;
    f32x4 pmu[2], psc[2], pbe[2], qmu[2], qsc[2], qbe[2];
    if constexpr (FIXC) {
        // (K8 < 256: the block's distinct channel octets are derived once, into LDS -- see bn_act_fixc_kernel)
        extern __shared__ float sums_tab[];      // [2][3][K] when shared (host: 24 K bytes)
        const bool share = SUMS && K8 < 256;
        const int slot = share ? ((int)threadIdx.x % K8) * 8 : 0;
        auto derive = [&](const BnSumsRef& r, int which, f32x4* o_mu, f32x4* o_sc, f32x4* o_be) {
            if (share) {
                if ((int)threadIdx.x < K8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float m_, s_, b_, is_;
                        bn_from_sums(r, cfix + e, m_, s_, b_, is_);
                        sums_tab[(which * 3 + 0) * K + slot + e] = m_; sums_tab[(which * 3 + 1) * K + slot + e] = s_; sums_tab[(which * 3 + 2) * K + slot + e] = b_;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    o_mu[hh] = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 0) * K + slot + 4 * hh);
                    o_sc[hh] = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 1) * K + slot + 4 * hh);
                    o_be[hh] = *reinterpret_cast<const f32x4*>(sums_tab + (which * 3 + 2) * K + slot + 4 * hh);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float m_, s_, b_, is_;
                    bn_from_sums(r, cfix + e, m_, s_, b_, is_);
                    o_mu[e >> 2][e & 3] = m_; o_sc[e >> 2][e & 3] = s_; o_be[e >> 2][e & 3] = b_;
                }
            }
        };
        if (SUMS && ys.sums) derive(ys, 0, pmu, psc, pbe);
        if (SUMS && rs.sums) derive(rs, 1, qmu, qsc, qbe);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (SUMS && ys.sums) {
            } else {
                pmu[hh] = *reinterpret_cast<const f32x4*>(bnp + cfix + 4 * hh);
                psc[hh] = *reinterpret_cast<const f32x4*>(bnp + K + cfix + 4 * hh);
                pbe[hh] = *reinterpret_cast<const f32x4*>(bnp + 2 * K + cfix + 4 * hh);
            }
            if (!(SUMS && rs.sums)) { qmu[hh] = pmu[hh]; qsc[hh] = psc[hh]; qbe[hh] = pbe[hh]; }
            if (SUMS && rs.sums) {
            } else if (rbnp_) {
                qmu[hh] = *reinterpret_cast<const f32x4*>(rbnp_ + cfix + 4 * hh);
                qsc[hh] = *reinterpret_cast<const f32x4*>(rbnp_ + K + cfix + 4 * hh);
                qbe[hh] = *reinterpret_cast<const f32x4*>(rbnp_ + 2 * K + cfix + 4 * hh);
            }
        }
    }
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n8; i0 += U * stride) {
        u32x4_t yv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n8) {
                yv[u] = y8[i];
                if (res) rv[u] = r8[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n8) break;
            const int c = FIXC ? cfix : (int)(i % K8) * 8;
            f32x4 v[2], r[2];
            widen8(yv[u], v[0], v[1]);
            if (res) widen8(rv[u], r[0], r[1]);
            unsigned bits = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr (FIXC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[hh][e] = fmaf(v[hh][e] - pmu[hh][e], psc[hh][e], pbe[hh][e]);
                } else {
                    v[hh] = bn_apply4(v[hh], bnp, K, c + 4 * hh);
                }
                if (res) {
                    if (rbnp) {
                        if constexpr (FIXC) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) r[hh][e] = fmaf(r[hh][e] - qmu[hh][e], qsc[hh][e], qbe[hh][e]);
                        } else {
                            r[hh] = bn_apply4(r[hh], rbnp, K, c + 4 * hh);
                        }
                    }
                    v[hh] += r[hh];
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[hh][e] = fmaxf(v[hh][e], 0.f);
                }
                bits |= relu_bits(v[hh]) << (8 * hh);
            }
            reinterpret_cast<u32x4_t*>(out)[i] = narrow8(v[0], v[1]);
            if (out32) {
                reinterpret_cast<f32x4*>(out32)[2 * i] = v[0];
                reinterpret_cast<f32x4*>(out32)[2 * i + 1] = v[1];
            }
            if (mask8) reinterpret_cast<unsigned short*>(mask8)[i] = (unsigned short)bits;      // one byte per 4 channels
        }
    }
}

// stem: maxpool3x3/s2/p1 of relu(bn(y)); one thread per (b,ph,pw,4 channels)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                               float* __restrict__ out,
                                                               int B, int H, int W, int K4, int PH, int PW) {
    DPFT_SETPRIO_BN();
    const int K = K4 * 4;
    const int64_t total = (int64_t)B * PH * PW * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        int64_t p = i / K4;
        const int pw = (int)(p % PW); p /= PW;
        const int ph = (int)(p % PH);
        const int b = (int)(p / PH);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};  // relu >= 0 and every window holds a valid pixel
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int h = ph * 2 - 1 + di;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int w = pw * 2 - 1 + dj;
                if ((unsigned)w >= (unsigned)W) continue;
                const f32x4 v = bn_apply4(reinterpret_cast<const f32x4*>(y)[(((int64_t)b * H + h) * W + w) * K4 + c4], bnp, K, c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        stv4<T>(out, i, m);
    }
}

// backward of the above, gather form: dz[b,h,w,c] = (a>0) * sum_{windows containing (h,w) whose first
// arg-max is (h,w)} dout.  a = relu(bn(y)) recomputed on the fly.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                                   const float* __restrict__ dout,
                                                                   float* __restrict__ dz, int B, int H, int W, int K4, int PH, int PW) {
    DPFT_SETPRIO_BN();
    const int K = K4 * 4;
    const int64_t total = (int64_t)B * H * W * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        int64_t p = i / K4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int b = (int)(p / H);
        f32x4 a0 = bn_apply4(reinterpret_cast<const f32x4*>(y)[i], bnp, K, c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) a0[e] = fmaxf(a0[e], 0.f);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // candidate windows: ph in {(h+1)/2 - (0|1)} with 2*ph-1 <= h <= 2*ph+1
        const int ph_hi = (h + 1) >> 1, pw_hi = (w + 1) >> 1;
        for (int ph = ph_hi - 1; ph <= ph_hi; ++ph) {
            if (ph < 0 || ph >= PH || h < 2 * ph - 1 || h > 2 * ph + 1) continue;
            for (int pw = pw_hi - 1; pw <= pw_hi; ++pw) {
                if (pw < 0 || pw >= PW || w < 2 * pw - 1 || w > 2 * pw + 1) continue;
                // first arg-max of the window in scan order, per channel
                f32x4 best = {-1.f, -1.f, -1.f, -1.f};
                int bi[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const int hh = ph * 2 - 1 + di;
                    if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) {
                        const int ww = pw * 2 - 1 + dj;
                        if ((unsigned)ww >= (unsigned)W) continue;
                        const f32x4 v = bn_apply4(reinterpret_cast<const f32x4*>(y)[(((int64_t)b * H + hh) * W + ww) * K4 + c4], bnp, K, c4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float av = fmaxf(v[e], 0.f);
                            if (av > best[e]) { best[e] = av; bi[e] = di * 3 + dj; }
                        }
                    }
                }
                const int me = (h - (ph * 2 - 1)) * 3 + (w - (pw * 2 - 1));
                const f32x4 d = ldv4<T>(dout, (((int64_t)b * PH + ph) * PW + pw) * K4 + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += (bi[e] == me) ? d[e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = a0[e] > 0.f ? g[e] : 0.f;
        reinterpret_cast<f32x4*>(dz)[i] = g;
    }
}

// The same backward, tiled (round 4).  The gather form above recomputes relu(bn(y)) of a 3 x 3 window for each of the <= 4
// windows an element belongs to -- 37 loads and ~330 vector instructions per 4 channels (139 us on the camera's 4 x 256 x
// 455 x 64 stem output, 800 us at batch 8 in the mixed-precision step).  Here a workgroup owns 8 x 16 elements x 32
// channels: (1) relu(bn(y)) of the 11 x 19 pixels its 5 x 9 windows cover goes to LDS once (one load per element + halo),
// the windows' dout with it; (2) one thread per (window, 4 channels) finds the first arg-max in scan order; (3) one thread
// per owned element sums the dout of the windows that chose it, in the order of the gather form -- bit-identical results.
constexpr int PB_TH = 4, PB_TW = 8, PB_CQ = 8;                              // windows owned per tile (rows, cols), channel quads
constexpr int PB_RH = 2 * PB_TH + 3, PB_RW = 2 * PB_TW + 3;                 // pixels staged
constexpr int PB_WH = PB_TH + 1, PB_WW = PB_TW + 1;                         // windows evaluated
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_tiled_kernel(const float* __restrict__ y, const float* __restrict__ bnp,
                                                                         const float* __restrict__ dout, float* __restrict__ dz,
                                                                         int B, int H, int W, int K4, int PH, int PW, int tiles_w,
                                                                         int tiles_h) {
    DPFT_SETPRIO_BN();
    __shared__ f32x4 act[PB_RH * PB_RW][PB_CQ];          // relu(bn(y)), -1 outside the image
    __shared__ f32x4 dwin[PB_WH * PB_WW][PB_CQ];         // dout of the windows (0 outside the pooled map)
    __shared__ unsigned arg[PB_WH * PB_WW][PB_CQ];       // first arg-max (0..8) of each window, one byte per channel
    const int K = K4 * 4, tid = threadIdx.x;
    int t = blockIdx.x;
    const int tw = t % tiles_w; t /= tiles_w;
    const int th = t % tiles_h; t /= tiles_h;
    const int cq0 = (t % (K4 / PB_CQ)) * PB_CQ;
    const int b = t / (K4 / PB_CQ);
    const int ph0 = th * PB_TH, pw0 = tw * PB_TW;
    const int h0 = 2 * ph0 - 1, w0 = 2 * pw0 - 1;        // first staged pixel
    for (int i = tid; i < PB_RH * PB_RW * PB_CQ; i += 256) {
        const int q = i % PB_CQ, px = i / PB_CQ;
        const int r = px / PB_RW, c = px - r * PB_RW;
        const int h = h0 + r, w = w0 + c;
        f32x4 v = {-1.f, -1.f, -1.f, -1.f};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            v = bn_apply4(reinterpret_cast<const f32x4*>(y)[(((int64_t)b * H + h) * W + w) * K4 + cq0 + q], bnp, K, (cq0 + q) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        act[px][q] = v;
    }
    for (int i = tid; i < PB_WH * PB_WW * PB_CQ; i += 256) {
        const int q = i % PB_CQ, wi = i / PB_CQ;
        const int ph = ph0 + wi / PB_WW, pw = pw0 + wi % PB_WW;
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        if (ph < PH && pw < PW) d = ldv4<T>(dout, (((int64_t)b * PH + ph) * PW + pw) * K4 + cq0 + q);
        dwin[wi][q] = d;
    }
    __syncthreads();
    for (int i = tid; i < PB_WH * PB_WW * PB_CQ; i += 256) {
        const int q = i % PB_CQ, wi = i / PB_CQ;
        const int wr = wi / PB_WW, wc = wi - wr * PB_WW;
        f32x4 best = {-1.f, -1.f, -1.f, -1.f};
        unsigned bi = 0xffffffffu;                      // 0xff per channel: no valid pixel (cannot happen for a real window)
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const f32x4 v = act[(2 * wr + di) * PB_RW + 2 * wc + dj][q];      // window (ph, pw) starts at pixel (2 ph - 1, 2 pw - 1)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e]) {                // outside pixels hold -1: never chosen, as the gather form skips them
                        best[e] = v[e];
                        bi = (bi & ~(0xffu << (8 * e))) | ((unsigned)(di * 3 + dj) << (8 * e));
                    }
            }
        arg[wi][q] = bi;
    }
    __syncthreads();
    for (int i = tid; i < 2 * PB_TH * 2 * PB_TW * PB_CQ; i += 256) {
        const int q = i % PB_CQ, px = i / PB_CQ;
        const int r = px / (2 * PB_TW), c = px - r * (2 * PB_TW);
        const int h = 2 * ph0 + r, w = 2 * pw0 + c;
        if (h >= H || w >= W) continue;
        const f32x4 a0 = act[(r + 1) * PB_RW + c + 1][q];
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        const int ph_hi = (h + 1) >> 1, pw_hi = (w + 1) >> 1;
        for (int ph = ph_hi - 1; ph <= ph_hi; ++ph) {
            if (ph < 0 || ph >= PH || h < 2 * ph - 1 || h > 2 * ph + 1) continue;
            for (int pw = pw_hi - 1; pw <= pw_hi; ++pw) {
                if (pw < 0 || pw >= PW || w < 2 * pw - 1 || w > 2 * pw + 1) continue;
                const int wi = (ph - ph0) * PB_WW + (pw - pw0);
                const unsigned me = (unsigned)((h - (ph * 2 - 1)) * 3 + (w - (pw * 2 - 1)));
                const unsigned bi = arg[wi][q];
                const f32x4 d = dwin[wi][q];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] += (((bi >> (8 * e)) & 0xffu) == me) ? d[e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = a0[e] > 0.f ? g[e] : 0.f;
        reinterpret_cast<f32x4*>(dz)[(((int64_t)b * H + h) * W + w) * K4 + cq0 + q] = g;
    }
}

// BN backward pass 1: sums[0][k] += sum dz, sums[1][k] += sum dz*xhat
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                             const float* __restrict__ outp, const float* __restrict__ mbnp,
                                                             const float* __restrict__ bnp, float* __restrict__ sums,
                                                             int64_t M, int K, int rows_per_block, int slab,
                                                             const unsigned char* __restrict__ mask8) {
    DPFT_SETPRIO_BN();
    __shared__ float red0[256 * 4];
    __shared__ float red1[256 * 4];
    const int K4 = K / 4;
    // slab of up to `slab` float4-chunks of channels per blockIdx.y (narrow slabs = many row groups per block =
    // few atomics per block and low contention per accumulator)
    const int kc = min(K4 - (int)blockIdx.y * slab, slab);
    const int groups = 256 / kc;
    const int cl = threadIdx.x % kc, g = threadIdx.x / kc;
    const int c4 = blockIdx.y * slab + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (g < groups) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c4 * 4);
        const f32x4 is = *reinterpret_cast<const f32x4*>(bnp + 3 * K + c4 * 4);
        // RT rows per trip: 8-12 independent 16-byte (bf16 storage: 16-24 8-byte) loads in flight per lane (this kernel is
        // pure HBM streaming)
        constexpr int RT = std::is_same<T, float>::value ? 4 : 8;
        for (int64_t rb = r0 + g; rb < r1; rb += RT * groups) {
            f32x4 dv[RT], yv4[RT], ov[RT];
            unsigned mk[RT];
            bool ok[RT];
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                const int64_t r = rb + (int64_t)u * groups;
                ok[u] = r < r1;
                const int64_t idx = (ok[u] ? r : r0) * K4 + c4;
                dv[u] = ldv4<T>(dout, idx);
                yv4[u] = ldv4<T>(y, idx);
                if (mask8) mk[u] = mask8[idx];
                else if (outp) ov[u] = ldv4<T>(outp, idx);
            }
#pragma unroll
            for (int u = 0; u < RT; ++u) {
                if (!ok[u]) continue;
                f32x4 d = dv[u];
                const f32x4 yv = yv4[u];
                if (mask8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = ((mk[u] >> e) & 1u) ? d[e] : 0.f;
                } else if (outp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = ov[u][e] > 0.f ? d[e] : 0.f;
                } else if (mbnp) {
                    const f32x4 a = bn_apply4(yv, mbnp, K, c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s0[e] += d[e];
                    s1[e] += d[e] * ((yv[e] - mu[e]) * is[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red0[threadIdx.x * 4 + e] = s0[e];
        red1[threadIdx.x * 4 + e] = s1[e];
    }
    __syncthreads();
    if (threadIdx.x < kc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t0 = 0.f, t1 = 0.f;
            for (int i = 0; i < groups; ++i) {
                t0 += red0[(threadIdx.x + i * kc) * 4 + e];
                t1 += red1[(threadIdx.x + i * kc) * 4 + e];
            }
            atomicAdd(&sums[c4 * 4 + e], t0);
            atomicAdd(&sums[K + c4 * 4 + e], t1);
        }
    }
}

// BN backward pass 2
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                            const float* __restrict__ outp, const float* __restrict__ mbnp,
                                                            const float* __restrict__ bnp, const float* __restrict__ gamma,
                                                            const float* __restrict__ sums, float* __restrict__ dy,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int64_t n4, int K, float invM, float* __restrict__ zero_buf,
                                                            int zero_n, const unsigned char* __restrict__ mask8) {
    DPFT_SETPRIO_BN();
    const int K4 = K / 4;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < K; c += blockDim.x) {
            if (dbeta) dbeta[c] = sums[c];
            if (dgamma) dgamma[c] = sums[K + c];
        }
        // ping-pong accumulators of the launch plan: clear the buffer the NEXT reduction will add into
        for (int c = threadIdx.x; c < zero_n; c += blockDim.x) zero_buf[c] = 0.f;
    }
    constexpr int U = std::is_same<T, float>::value ? 1 : 2;      // see bn_act_kernel
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += U * stride) {
      f32x4 dq[U], yq[U], oq[U];
      unsigned mq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
          const int64_t i = i0 + u * stride;
          if (i < n4) {
              dq[u] = ldv4<T>(dout, i);
              yq[u] = ldv4<T>(y, i);
              if (mask8) mq[u] = mask8[i];
              else if (outp) oq[u] = ldv4<T>(outp, i);
          }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u * stride;
        if (i >= n4) break;
        const int c = (int)(i % K4) * 4;
        f32x4 d = dq[u];
        const f32x4 yv = yq[u];
        if (mask8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = ((mq[u] >> e) & 1u) ? d[e] : 0.f;
        } else if (outp) {
            const f32x4 o = oq[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        } else if (mbnp) {
            const f32x4 a = bn_apply4(yv, mbnp, K, c);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : 0.f;
        }
        const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(bnp + 3 * K + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + K + c);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mu[e]) * is[e];
            r[e] = ga[e] * is[e] * (d[e] - s0[e] * invM - xh * s1[e] * invM);
        }
        stv4<T>(dy, i, r);
      }
    }
}

// BN backward pass 2, fp32, for launches whose grid stride is a multiple of K / 4 (round 6): a thread then meets the SAME four
// channels in every trip -- its five parameter quads are loaded once -- and keeps U element quads of both tensors in flight.
// Why: inside the training step this pass shares its CUs with the weight-gradient stream's workgroups (the split kernels hold
// 456 of a SIMD's 512 registers: ONE wave of this kernel fits beside them), and with one trip's two loads per wave in flight
// it ran 2.3x slower there than alone (3.5 ms of the main queue, profiles/r06_bn_in_step.txt).  Stays within 56 registers so
// that it still fits beside such a wave.  Same expression per element as bn_bwd_apply_kernel.
template <int U>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(56)))
void bn_bwd_apply_fixc_kernel(const float* __restrict__ y, const float* __restrict__ dout, const float* __restrict__ outp,
                              const float* __restrict__ mbnp, const float* __restrict__ bnp, const float* __restrict__ gamma,
                              const float* __restrict__ sums, float* __restrict__ dy, float* __restrict__ dgamma,
                              float* __restrict__ dbeta, int64_t n4, int K, float invM, float* __restrict__ zero_buf, int zero_n,
                              const unsigned char* __restrict__ mask8) {
    DPFT_SETPRIO_BN();
    const int K4 = K / 4;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < K; c += blockDim.x) {
            if (dbeta) dbeta[c] = sums[c];
            if (dgamma) dgamma[c] = sums[K + c];
        }
        for (int c = threadIdx.x; c < zero_n; c += blockDim.x) zero_buf[c] = 0.f;
    }
    const unsigned stride = gridDim.x * blockDim.x;      // % K4 == 0 (host); n4 < 2^31 (host)
    const unsigned first = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(first % (unsigned)K4) * 4;
    // r = ga is (d - s0 / M - xhat s1 / M),  xhat = (y - mu) is   ==   A d + P + Q y  with three constants per channel
    // (12 registers instead of 20 for the five parameter quads: the kernel has to stay within 56)
    f32x4 A, P, Q;
    {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(bnp + 3 * K + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + K + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            A[e] = ga[e] * is[e];
            Q[e] = -A[e] * is[e] * (s1[e] * invM);
            P[e] = -A[e] * (s0[e] * invM) - Q[e] * mu[e];
        }
    }
    const unsigned n = (unsigned)n4;
    for (unsigned i0 = first; i0 < n; i0 += U * stride) {
        f32x4 dq[U], yq[U];
        unsigned mq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            if (i < n) {
                dq[u] = reinterpret_cast<const f32x4*>(dout)[i];
                yq[u] = reinterpret_cast<const f32x4*>(y)[i];
                if (mask8) mq[u] = mask8[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned i = i0 + u * stride;
            if (i >= n) break;
            f32x4 d = dq[u];
            const f32x4 yv = yq[u];
            if (mask8) {
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = ((mq[u] >> e) & 1u) ? d[e] : 0.f;
            } else if (outp) {
                const f32x4 o = reinterpret_cast<const f32x4*>(outp)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
            } else if (mbnp) {
                const f32x4 a = bn_apply4(yv, mbnp, K, c);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : 0.f;
            }
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaf(A[e], d[e], fmaf(Q[e], yv[e], P[e]));
            reinterpret_cast<f32x4*>(dy)[i] = r;
        }
    }
}

// BN backward pass 2, bf16 storage with 16-byte accesses (see bn_act16_kernel); K % 8 == 0
template <bool FIXC>      // FIXC: as bn_act16_kernel (three precombined constants per channel, see bn_bwd_apply_fixc_kernel)
__global__ __launch_bounds__(256) void bn_bwd_apply16_kernel(const float* __restrict__ y, const float* __restrict__ dout,
                                                              const float* __restrict__ outp, const float* __restrict__ mbnp,
                                                              const float* __restrict__ bnp, const float* __restrict__ gamma,
                                                              const float* __restrict__ sums, float* __restrict__ dy,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int64_t n8, int K, float invM, float* __restrict__ zero_buf,
                                                              int zero_n, const unsigned char* __restrict__ mask8) {
    DPFT_SETPRIO_BN();
    const int K8 = K / 8;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < K; c += blockDim.x) {
            if (dbeta) dbeta[c] = sums[c];
            if (dgamma) dgamma[c] = sums[K + c];
        }
        for (int c = threadIdx.x; c < zero_n; c += blockDim.x) zero_buf[c] = 0.f;
    }
    constexpr int U = 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const u32x4_t* __restrict__ d8 = reinterpret_cast<const u32x4_t*>(dout);
    const u32x4_t* __restrict__ y8 = reinterpret_cast<const u32x4_t*>(y);
    const u32x4_t* __restrict__ o8 = reinterpret_cast<const u32x4_t*>(outp);
    const int cfix = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % K8) * 8;
    f32x4 cA[2], cP[2], cQ[2];
    if constexpr (FIXC) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int c = cfix + 4 * hh;
            const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c);
            const f32x4 is = *reinterpret_cast<const f32x4*>(bnp + 3 * K + c);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + K + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cA[hh][e] = ga[e] * is[e];
                cQ[hh][e] = -cA[hh][e] * is[e] * (s1[e] * invM);
                cP[hh][e] = -cA[hh][e] * (s0[e] * invM) - cQ[hh][e] * mu[e];
            }
        }
    }
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n8; i0 += U * stride) {
        u32x4_t dq[U], yq[U], oq[U];
        unsigned mq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n8) {
                dq[u] = d8[i];
                yq[u] = y8[i];
                if (mask8) mq[u] = reinterpret_cast<const unsigned short*>(mask8)[i];
                else if (outp) oq[u] = o8[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n8) break;
            const int c0 = FIXC ? cfix : (int)(i % K8) * 8;
            f32x4 d[2], yv[2], o[2], r[2];
            widen8(dq[u], d[0], d[1]);
            widen8(yq[u], yv[0], yv[1]);
            if (!mask8 && outp) widen8(oq[u], o[0], o[1]);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int c = c0 + 4 * hh;
                if (mask8) {
                    const unsigned mk = mq[u] >> (8 * hh);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[hh][e] = ((mk >> e) & 1u) ? d[hh][e] : 0.f;
                } else if (outp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[hh][e] = o[hh][e] > 0.f ? d[hh][e] : 0.f;
                } else if (mbnp) {
                    const f32x4 a = bn_apply4(yv[hh], mbnp, K, c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[hh][e] = a[e] > 0.f ? d[hh][e] : 0.f;
                }
                if constexpr (FIXC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[hh][e] = fmaf(cA[hh][e], d[hh][e], fmaf(cQ[hh][e], yv[hh][e], cP[hh][e]));
                } else {
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(bnp + c);
                    const f32x4 is = *reinterpret_cast<const f32x4*>(bnp + 3 * K + c);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c);
                    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + K + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (yv[hh][e] - mu[e]) * is[e];
                        r[hh][e] = ga[e] * is[e] * (d[hh][e] - s0[e] * invM - xh * s1[e] * invM);
                    }
                }
            }
            reinterpret_cast<u32x4_t*>(dy)[i] = narrow8(r[0], r[1]);
        }
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ outp,
                                                        float* __restrict__ dz, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 d = reinterpret_cast<const f32x4*>(dout)[i];
        const f32x4 o = reinterpret_cast<const f32x4*>(outp)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        reinterpret_cast<f32x4*>(dz)[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        dz[i] = outp[i] > 0.f ? dout[i] : 0.f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = ldv4<T>(a, i);
        v += reinterpret_cast<const f32x4*>(b)[i];
        stv4<T>(a, i, v);
    }
    if (!std::is_same<T, float>::value) return;      // bf16 storage: n % 4 == 0 (host side)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        a[i] += b[i];
    }
}

__device__ __forceinline__ int nearest_src(int dst, float scale, int n_in) {
    return min((int)floorf((float)dst * scale), n_in - 1);
}

// dst (bf16) = src (fp32): the external gradient of a stage output entering a bf16-storage plan
__global__ __launch_bounds__(256) void cvt_f32_bf16_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        stv4<__bf16>(dst, i, reinterpret_cast<const f32x4*>(src)[i]);
}

// lat += nearest_upsample(top)
__global__ __launch_bounds__(256) void fpn_topdown_add_kernel(float* __restrict__ lat, const float* __restrict__ top,
                                                               int B, int H, int W, int TH, int TW, int K4,
                                                               float sh, float sw) {
    const int64_t total = (int64_t)B * H * W * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        int64_t p = i / K4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int b = (int)(p / H);
        const int th = (TH == H) ? h : nearest_src(h, sh, TH);
        const int tw = (TW == W) ? w : nearest_src(w, sw, TW);
        f32x4 v = reinterpret_cast<f32x4*>(lat)[i];
        v += reinterpret_cast<const f32x4*>(top)[(((int64_t)b * TH + th) * TW + tw) * K4 + c4];
        reinterpret_cast<f32x4*>(lat)[i] = v;
    }
}

// dtop += sum over destination pixels mapping to each source pixel (gather, deterministic)
__global__ __launch_bounds__(256) void fpn_topdown_add_bwd_kernel(const float* __restrict__ dlat, float* __restrict__ dtop,
                                                                   int B, int H, int W, int TH, int TW, int K4,
                                                                   float sh, float sw) {
    const int64_t total = (int64_t)B * TH * TW * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        int64_t p = i / K4;
        const int tw = (int)(p % TW); p /= TW;
        const int th = (int)(p % TH);
        const int b = (int)(p / TH);
        // candidate destination range (conservative), filtered by the exact forward map
        const int h_lo = max(0, (int)floorf((float)th / sh) - 2), h_hi = min(H - 1, (int)ceilf((float)(th + 1) / sh) + 2);
        const int w_lo = max(0, (int)floorf((float)tw / sw) - 2), w_hi = min(W - 1, (int)ceilf((float)(tw + 1) / sw) + 2);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int h = h_lo; h <= h_hi; ++h) {
            const int s_h = (TH == H) ? h : nearest_src(h, sh, TH);
            if (s_h != th) continue;
            for (int w = w_lo; w <= w_hi; ++w) {
                const int s_w = (TW == W) ? w : nearest_src(w, sw, TW);
                if (s_w != tw) continue;
                acc += reinterpret_cast<const f32x4*>(dlat)[(((int64_t)b * H + h) * W + w) * K4 + c4];
            }
        }
        f32x4 v = reinterpret_cast<f32x4*>(dtop)[i];
        v += acc;
        reinterpret_cast<f32x4*>(dtop)[i] = v;
    }
}

__global__ __launch_bounds__(256) void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos_x,
                                                       const float* __restrict__ pos_y, int B, int H, int W, int K4) {
    const int64_t total = (int64_t)B * H * W * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % K4);
        int64_t p = i / K4;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v += reinterpret_cast<const f32x4*>(pos_x)[(int64_t)w * K4 + c4];
        v += reinterpret_cast<const f32x4*>(pos_y)[(int64_t)h * K4 + c4];
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// fixed-channel forms (round 6): can the launch's grid stride (blocks x 256) be a multiple of `kq` channel groups?  May lower
// `blocks` to make it so.  DPFT_BN_FIXC=0: off (A/B switch).
static inline bool fixc_grid(int kq, int64_t items, int& blocks) {
    static const int fixc = getenv("DPFT_BN_FIXC") ? atoi(getenv("DPFT_BN_FIXC")) : 2;
    if (fixc <= 0 || items < 4096 || items >= (1ll << 30) || kq <= 0) return false;
    if ((256 % kq) == 0) return true;
    if ((kq % 256) != 0) return false;
    const int f = kq / 256;
    if (blocks < f) return false;
    blocks -= blocks % f;
    return true;
}

static inline int ew_blocks(int64_t work_items) {
    static const int per_cu = getenv("DPFT_EW_BLOCKS_PER_CU") ? atoi(getenv("DPFT_EW_BLOCKS_PER_CU")) : 8;      // tuning aid
    return (int)std::max<int64_t>(1, std::min<int64_t>((work_items + 255) / 256, (int64_t)kNumCU * per_cu));
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_bn_stats_f32(const float* y, float* stats, int64_t M, int32_t K, int32_t tile_rows,
                                 dpft_stream_t stream) {
    DPFT_REQUIRE(y && stats && M > 0 && K > 0 && tile_rows > 0, "bn_stats: bad arguments");
    dim3 grid(cdiv(M, tile_rows), cdiv(K, 256));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, stats, M, K, tile_rows);
    return check_launch("bn_stats");
}

extern "C" int dpft_bn_finalize_f32(const float* stats, int32_t tiles, int32_t tile_rows, int64_t M, int32_t K,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* bnp,
                                    dpft_stream_t stream) {
    {
        static const char* skip = getenv("DPFT_SKIP");      // timing experiment (wrong results): the family's cost on the critical path
        if (skip && (!strcmp(skip, "bnfinalize") || (strstr(skip, "bnfinK1024") && K >= 1024))) return DPFT_OK;
    }
    DPFT_REQUIRE(stats && gamma && beta && bnp, "bn_finalize: null tensor");
    DPFT_REQUIRE(tiles == cdiv(M, tile_rows), "bn_finalize: tiles (%d) != ceil(M/tile_rows)", tiles);
    DPFT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(K, 32)), dim3(1024), 0, (hipStream_t)stream, stats, tiles,
                       tile_rows, M, K, gamma, beta, eps, momentum, running_mean, running_var, bnp);
    return check_launch("bn_finalize");
}

extern "C" int dpft_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, int32_t K, float* bnp,
                                       dpft_stream_t stream) {
    DPFT_REQUIRE(gamma && beta && running_mean && running_var && bnp && K > 0, "bn_eval: bad arguments");
    hipLaunchKernelGGL(bn_eval_kernel, dim3(cdiv(K, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, K, bnp);
    return check_launch("bn_eval_params");
}

int dpft::bn_eval_params_batch(const BnEvalBatch& batch, dpft_stream_t stream) {
    DPFT_REQUIRE(batch.n >= 1 && batch.n <= 16, "bn_eval_params_batch: 1..16 layers per launch");
    int kmax = 0;
    for (int i = 0; i < batch.n; ++i) {
        DPFT_REQUIRE(batch.gamma[i] && batch.beta[i] && batch.rm[i] && batch.rv[i] && batch.out[i] && batch.K[i] > 0,
                     "bn_eval_params_batch: bad entry %d", i);
        kmax = std::max(kmax, batch.K[i]);
    }
    hipLaunchKernelGGL(bn_eval_multi_kernel, dim3(cdiv(kmax, 256), batch.n), dim3(256), 0, (hipStream_t)stream, batch);
    return check_launch("bn_eval_params_batch");
}

int dpft::bn_act_any(const float* y, const float* bnp, const float* res, const float* res_bnp, int32_t relu, float* out,
                     float* out32, int64_t M, int32_t K, bool act16, dpft_stream_t stream, unsigned char* mask8) {
    {
        static const char* skip = getenv("DPFT_SKIP");
        if (skip && strstr(skip, "bnact")) return DPFT_OK;
    }
    DPFT_REQUIRE(y && bnp && out && M > 0 && K > 0 && K % 4 == 0, "bn_act: bad arguments (K=%d)", K);
    DPFT_REQUIRE(res || !res_bnp, "bn_act: res_bnp without res");
    const int64_t n4 = M * K / 4;
    static const bool wide16 = getenv("DPFT_BN_WIDE16") == nullptr || atoi(getenv("DPFT_BN_WIDE16")) != 0;      // A/B switch
    if (act16 && wide16 && K % 8 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)res & 15) == 0 &&
        ((uintptr_t)mask8 & 1) == 0)
    {
        int blocks = ew_blocks(n4 / 2);
        if (fixc_grid(K / 8, n4 / 2, blocks))
            hipLaunchKernelGGL((bn_act16_kernel<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bnp, res, res_bnp, relu, out,
                               out32, n4 / 2, K / 8, mask8, BnSumsRef{}, BnSumsRef{});
        else
            hipLaunchKernelGGL((bn_act16_kernel<false, false>), dim3(ew_blocks(n4 / 2)), dim3(256), 0, (hipStream_t)stream, y, bnp, res, res_bnp, relu, out,
                               out32, n4 / 2, K / 8, mask8, BnSumsRef{}, BnSumsRef{});
    }
    else if (act16)
        hipLaunchKernelGGL(bn_act_kernel<__bf16>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, y, bnp, res,
                           res_bnp, relu, out, out32, n4, K / 4, mask8);
    else {
        static const int fixc = getenv("DPFT_BN_FIXC") ? atoi(getenv("DPFT_BN_FIXC")) : 2;      // see bn_bwd_apply_zeroing
        const int K4 = K / 4;
        static const int fat = getenv("DPFT_BN_FAT") ? atoi(getenv("DPFT_BN_FAT")) : 1;      // two quads per thread and trip (as bn_bwd_apply)
        int blocks = ew_blocks(fat && fixc >= 2 ? (n4 + 1) / 2 : n4);
        bool ok = fixc > 0 && !out32 && n4 >= 4096 && n4 < (1ll << 30);
        if (ok && (256 % K4) != 0) {
            const int f = K4 / 256;
            ok = (K4 % 256) == 0 && blocks >= f;
            if (ok) blocks -= blocks % f;
        }
        if (ok && fixc >= 2)
            hipLaunchKernelGGL((bn_act_fixc_kernel<2, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bnp, res, res_bnp, relu, out, n4,
                               K4, mask8, BnSumsRef{}, BnSumsRef{});
        else if (ok)
            hipLaunchKernelGGL((bn_act_fixc_kernel<1, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, bnp, res, res_bnp, relu, out, n4,
                               K4, mask8, BnSumsRef{}, BnSumsRef{});
        else
            hipLaunchKernelGGL(bn_act_kernel<float>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, y, bnp, res,
                               res_bnp, relu, out, out32, n4, K / 4, mask8);
    }
    return check_launch("bn_act");
}

int dpft::bn_act_sums(const float* y, const float* bnp, const BnSumsRef& ys, const float* res, const float* res_bnp, const BnSumsRef& rs,
                      int32_t relu, float* out, int64_t M, int32_t K, dpft_stream_t stream, unsigned char* mask8, bool* used,
                      bool act16, float* out32) {
    DPFT_REQUIRE(y && out && used && M > 0 && K > 0 && K % 4 == 0 && (bnp || ys.sums), "bn_act (column sums): bad arguments (K=%d)", K);
    DPFT_REQUIRE(res || !(res_bnp || rs.sums), "bn_act (column sums): a residual BatchNorm without a residual");
    *used = false;
    {
        static const char* skip = getenv("DPFT_SKIP");
        if (skip && strstr(skip, "bnact")) { *used = true; return DPFT_OK; }
    }
    static const int fixc = getenv("DPFT_BN_FIXC") ? atoi(getenv("DPFT_BN_FIXC")) : 2;
    static const int fat = getenv("DPFT_BN_FAT") ? atoi(getenv("DPFT_BN_FAT")) : 1;
    const int64_t n4 = M * K / 4;
    const int K4 = K / 4;
    if (act16) {      // bf16 storage: the 16-byte fixed-channel form (bn_act_any's conditions)
        static const bool wide16 = getenv("DPFT_BN_WIDE16") == nullptr || atoi(getenv("DPFT_BN_WIDE16")) != 0;
        if (!(wide16 && K % 8 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)res & 15) == 0 && ((uintptr_t)mask8 & 1) == 0))
            return DPFT_OK;
        int blocks = ew_blocks(n4 / 2);
        if (!fixc_grid(K / 8, n4 / 2, blocks)) return DPFT_OK;
        hipLaunchKernelGGL((bn_act16_kernel<true, true>), dim3(blocks), dim3(256), K / 8 < 256 ? (size_t)24 * K : 0, (hipStream_t)stream, y, bnp,
                           res, res_bnp, relu, out, out32, n4 / 2, K / 8, mask8, ys, rs);
        *used = true;
        return check_launch("bn_act (bf16 storage, column sums)");
    }
    if (out32) return DPFT_OK;
    int blocks = ew_blocks(fat && fixc >= 2 ? (n4 + 1) / 2 : n4);
    bool ok = fixc > 0 && n4 >= 4096 && n4 < (1ll << 30);
    if (ok && (256 % K4) != 0) {
        const int f = K4 / 256;
        ok = (K4 % 256) == 0 && blocks >= f;
        if (ok) blocks -= blocks % f;
    }
    if (!ok) return DPFT_OK;      // (the generic kernel reads BN blocks only: the caller finalizes first)
    if (fixc >= 2)
        hipLaunchKernelGGL((bn_act_fixc_kernel<2, true>), dim3(blocks), dim3(256), K4 < 256 ? (size_t)24 * K : 0, (hipStream_t)stream, y, bnp, res,
                           res_bnp, relu, out, n4, K4, mask8, ys, rs);
    else
        hipLaunchKernelGGL((bn_act_fixc_kernel<1, true>), dim3(blocks), dim3(256), K4 < 256 ? (size_t)24 * K : 0, (hipStream_t)stream, y, bnp, res,
                           res_bnp, relu, out, n4, K4, mask8, ys, rs);
    *used = true;
    return check_launch("bn_act (column sums)");
}

__global__ __launch_bounds__(256) void bn_finalize_sums_multi_kernel(BnSumsBatch a) {
    const int i = blockIdx.y;
    const int K = a.K[i];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    const BnSumsRef r{a.sums[i], a.gamma[i], a.beta[i], a.invn[i], a.eps, K};
    float mean, scale, beta, invstd;
    bn_from_sums(r, c, mean, scale, beta, invstd);      // what every forward consumer of the layer computed
    float* bnp = a.out[i];
    bnp[c] = mean;
    bnp[K + c] = scale;
    bnp[2 * K + c] = beta;
    bnp[3 * K + c] = invstd;
    if (a.rm[i]) {
        const double m = bn_sums_get(r.sums, K, c, 0) * r.invn;
        const double var = fmax(bn_sums_get(r.sums, K, c, 1) * r.invn - m * m, 0.0);
        const long long M = a.M[i];
        const float unbiased = (float)(M > 1 ? var * ((double)M / (double)(M - 1)) : var);
        a.rm[i][c] = (1.f - a.momentum) * a.rm[i][c] + a.momentum * mean;
        a.rv[i][c] = (1.f - a.momentum) * a.rv[i][c] + a.momentum * unbiased;
    }
}

int dpft::bn_finalize_sums_batch(const BnSumsBatch& batch, dpft_stream_t stream) {
    DPFT_REQUIRE(batch.n >= 1 && batch.n <= BnSumsBatch::MAX, "bn_finalize_sums_batch: 1..%d layers per launch (n=%d)", BnSumsBatch::MAX, batch.n);
    int kmax = 0;
    for (int i = 0; i < batch.n; ++i) kmax = std::max(kmax, batch.K[i]);
    hipLaunchKernelGGL(bn_finalize_sums_multi_kernel, dim3(cdiv(kmax, 256), batch.n), dim3(256), 0, (hipStream_t)stream, batch);
    return check_launch("bn_finalize (column sums, batched)");
}

extern "C" int dpft_bn_act_f32(const float* y, const float* bnp, const float* res, const float* res_bnp,
                               int32_t relu, float* out, int64_t M, int32_t K, dpft_stream_t stream) {
    return dpft::bn_act_any(y, bnp, res, res_bnp, relu, out, nullptr, M, K, false, stream);
}

int dpft::bn_relu_maxpool_any(const float* y, const float* bnp, float* out, int32_t B, int32_t H, int32_t W, int32_t K,
                              int32_t PH, int32_t PW, bool out16, dpft_stream_t stream) {
    DPFT_REQUIRE(y && bnp && out && K % 4 == 0, "bn_relu_maxpool: bad arguments");
    DPFT_REQUIRE(PH == (H + 2 - 3) / 2 + 1 && PW == (W + 2 - 3) / 2 + 1, "bn_relu_maxpool: PH/PW inconsistent");
    const int64_t total = (int64_t)B * PH * PW * (K / 4);
    if (out16)
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, bnp,
                           out, B, H, W, K / 4, PH, PW);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, bnp,
                           out, B, H, W, K / 4, PH, PW);
    return check_launch("bn_relu_maxpool");
}

extern "C" int dpft_bn_relu_maxpool_f32(const float* y, const float* bnp, float* out,
                                        int32_t B, int32_t H, int32_t W, int32_t K, int32_t PH, int32_t PW,
                                        dpft_stream_t stream) {
    return dpft::bn_relu_maxpool_any(y, bnp, out, B, H, W, K, PH, PW, false, stream);
}

int dpft::bn_relu_maxpool_bwd_any(const float* y, const float* bnp, const float* dout, float* dact, int32_t B, int32_t H,
                                  int32_t W, int32_t K, int32_t PH, int32_t PW, bool dout16, dpft_stream_t stream) {
    DPFT_REQUIRE(y && bnp && dout && dact && K % 4 == 0, "bn_relu_maxpool_bwd: bad arguments");
    const int64_t total = (int64_t)B * H * W * (K / 4);
    static const bool tiled = getenv("DPFT_POOL_BWD_TILED") == nullptr || atoi(getenv("DPFT_POOL_BWD_TILED")) != 0;      // A/B switch
    const int K4 = K / 4;
    if (tiled && K4 % PB_CQ == 0 && PH == (H - 1) / 2 + 1 && PW == (W - 1) / 2 + 1) {      // 3 x 3 / stride 2 / pad 1 geometry
        const int tiles_h = cdiv(H, 2 * PB_TH), tiles_w = cdiv(W, 2 * PB_TW);
        const int64_t nb = (int64_t)B * (K4 / PB_CQ) * tiles_h * tiles_w;
        DPFT_REQUIRE(nb < (1ll << 31), "bn_relu_maxpool_bwd: too many tiles");
        if (dout16)
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_tiled_kernel<__bf16>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, y, bnp,
                               dout, dact, B, H, W, K4, PH, PW, tiles_w, tiles_h);
        else
            hipLaunchKernelGGL(bn_relu_maxpool_bwd_tiled_kernel<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, y, bnp,
                               dout, dact, B, H, W, K4, PH, PW, tiles_w, tiles_h);
        return check_launch("bn_relu_maxpool_bwd (tiled)");
    }
    if (dout16)
        hipLaunchKernelGGL(bn_relu_maxpool_bwd_kernel<__bf16>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y,
                           bnp, dout, dact, B, H, W, K / 4, PH, PW);
    else
        hipLaunchKernelGGL(bn_relu_maxpool_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y,
                           bnp, dout, dact, B, H, W, K / 4, PH, PW);
    return check_launch("bn_relu_maxpool_bwd");
}

extern "C" int dpft_bn_relu_maxpool_bwd_f32(const float* y, const float* bnp,
                                            const float* dout, float* dact, int32_t B, int32_t H, int32_t W,
                                            int32_t K, int32_t PH, int32_t PW, dpft_stream_t stream) {
    return dpft::bn_relu_maxpool_bwd_any(y, bnp, dout, dact, B, H, W, K, PH, PW, false, stream);
}

extern "C" int dpft_bn_bwd_reduce_f32(const float* y, const float* dout, const float* out,
                                      const float* mask_bnp, const float* bnp, float* sums, int64_t M,
                                      int32_t K, dpft_stream_t stream) {
    DPFT_REQUIRE(sums && K > 0, "bn_bwd_reduce: bad arguments");
    DPFT_REQUIRE(hipMemsetAsync(sums, 0, sizeof(float) * 2 * K, (hipStream_t)stream) == hipSuccess, "bn_bwd_reduce: memset failed");
    return dpft::bn_bwd_reduce_prezeroed(y, dout, out, mask_bnp, bnp, sums, M, K, false, stream);
}

// `sums` (2K floats) must already be zero (the launch plan keeps two buffers and lets each apply pass clear the other)
int dpft::bn_bwd_reduce_prezeroed(const float* y, const float* dout, const float* out, const float* mask_bnp,
                                  const float* bnp, float* sums, int64_t M, int32_t K, bool act16, dpft_stream_t stream,
                                  const unsigned char* mask8) {
    DPFT_REQUIRE(y && dout && bnp && sums && M > 0 && K > 0 && K % 4 == 0, "bn_bwd_reduce: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int K4 = K / 4;
    // Tuned on MI355X (tools/bn_bench.py): 128 B wide channel slabs (8 float4) => 32 row groups per block and only
    // 64 accumulators per block; at most 64 blocks per slab keeps the fp32 atomics on one accumulator <= 64-deep
    // (the first version -- every block covering all K channels, 512 blocks -- spent most of its time in atomics).
    const int slab = std::min(K4, 8);
    const int slabs = cdiv(K4, slab);
    const int groups = 256 / slab;
    const int want_blocks = std::min(64, std::max(1, (kNumCU * 4) / slabs));
    int64_t rows_per_block = std::max<int64_t>((int64_t)groups * 4, (M + want_blocks - 1) / want_blocks);
    dim3 grid(cdiv(M, rows_per_block), slabs);
    if (act16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<__bf16>, grid, dim3(256), 0, st, y, dout, out, mask_bnp, bnp, sums, M, K,
                           (int)rows_per_block, slab, mask8);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, y, dout, out, mask_bnp, bnp, sums, M, K,
                           (int)rows_per_block, slab, mask8);
    return check_launch("bn_bwd_reduce");
}

extern "C" int dpft_bn_bwd_apply_f32(const float* y, const float* dout, const float* out,
                                     const float* mask_bnp, const float* bnp, const float* gamma,
                                     const float* sums, float* dy, float* dgamma, float* dbeta, int64_t M,
                                     int32_t K, dpft_stream_t stream) {
    return dpft::bn_bwd_apply_zeroing(y, dout, out, mask_bnp, bnp, gamma, sums, dy, dgamma, dbeta, M, K, nullptr, 0, false, stream);
}

int dpft::bn_bwd_apply_zeroing(const float* y, const float* dout, const float* out, const float* mask_bnp,
                               const float* bnp, const float* gamma, const float* sums, float* dy, float* dgamma,
                               float* dbeta, int64_t M, int32_t K, float* zero_buf, int32_t zero_n, bool act16,
                               dpft_stream_t stream, const unsigned char* mask8, bool frozen) {
    DPFT_REQUIRE(y && dout && bnp && gamma && sums && dy && M > 0 && K % 4 == 0, "bn_bwd_apply: bad arguments");
    const int64_t n4 = M * K / 4;
    {      // timing experiment (wrong gradients): what the pass costs on the step's critical path = the step time without it
        static const bool skip = getenv("DPFT_BN_SKIP_APPLY") != nullptr && atoi(getenv("DPFT_BN_SKIP_APPLY")) != 0;
        if (skip) return DPFT_OK;
    }
    // frozen (running-statistics) BatchNorm: mean and variance do not depend on the batch, so dy = gamma invstd d -- the
    // batch form with its two mean terms weighted by 1/M = 0; dgamma = sum d xhat and dbeta = sum d are the same sums
    const float invM = frozen ? 0.f : 1.0f / (float)M;
    static const bool wide16 = getenv("DPFT_BN_WIDE16") == nullptr || atoi(getenv("DPFT_BN_WIDE16")) != 0;      // A/B switch
    if (act16 && wide16 && K % 8 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)dy & 15) == 0 &&
        ((uintptr_t)out & 15) == 0 && ((uintptr_t)mask8 & 1) == 0)
    {
        int blocks = ew_blocks(n4 / 2);
        if (fixc_grid(K / 8, n4 / 2, blocks))
            hipLaunchKernelGGL(bn_bwd_apply16_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dout, out, mask_bnp, bnp,
                               gamma, sums, dy, dgamma, dbeta, n4 / 2, K, invM, zero_buf, (int)zero_n, mask8);
        else
            hipLaunchKernelGGL(bn_bwd_apply16_kernel<false>, dim3(ew_blocks(n4 / 2)), dim3(256), 0, (hipStream_t)stream, y, dout, out, mask_bnp, bnp,
                               gamma, sums, dy, dgamma, dbeta, n4 / 2, K, invM, zero_buf, (int)zero_n, mask8);
    }
    else if (act16)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, y, dout, out,
                           mask_bnp, bnp, gamma, sums, dy, dgamma, dbeta, n4, K, invM, zero_buf, (int)zero_n, mask8);
    else {
        // fixed-channel form (see bn_bwd_apply_fixc_kernel): the grid stride must be a multiple of K / 4.  DPFT_BN_FIXC=0: off; =U: quads in flight
        static const int fixc = getenv("DPFT_BN_FIXC") ? atoi(getenv("DPFT_BN_FIXC")) : 2;
        const int K4 = K / 4;
        // U quads per thread and trip: a grid of n4 / (256 U) workgroups (inside the step a CU that also holds a split weight-gradient
        // workgroup has room for ONE wave of this kernel per SIMD: the pass runs as rounds of 256 workgroups, so fewer, fatter ones)
        static const int fat = getenv("DPFT_BN_FAT") ? atoi(getenv("DPFT_BN_FAT")) : 1;      // A/B switch
        int blocks = ew_blocks(fat && fixc >= 2 ? (n4 + 1) / 2 : n4);
        bool ok = fixc > 0 && n4 >= 4096 && n4 < (1ll << 30);
        if (ok && (256 % K4) != 0) {      // K4 = 512 ...: the block count itself must carry the remaining factor
            const int f = K4 / 256;
            ok = (K4 % 256) == 0 && blocks >= f;
            if (ok) blocks -= blocks % f;
        }
        if (ok && fixc >= 3)
            hipLaunchKernelGGL(bn_bwd_apply_fixc_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dout, out, mask_bnp, bnp,
                               gamma, sums, dy, dgamma, dbeta, n4, K, invM, zero_buf, (int)zero_n, mask8);
        else if (ok && fixc == 2)
            hipLaunchKernelGGL(bn_bwd_apply_fixc_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dout, out, mask_bnp, bnp,
                               gamma, sums, dy, dgamma, dbeta, n4, K, invM, zero_buf, (int)zero_n, mask8);
        else if (ok)
            hipLaunchKernelGGL(bn_bwd_apply_fixc_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, dout, out, mask_bnp, bnp,
                               gamma, sums, dy, dgamma, dbeta, n4, K, invM, zero_buf, (int)zero_n, mask8);
        else
            hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, y, dout, out,
                               mask_bnp, bnp, gamma, sums, dy, dgamma, dbeta, n4, K, invM, zero_buf, (int)zero_n, mask8);
    }
    return check_launch("bn_bwd_apply");
}

extern "C" int dpft_relu_bwd_f32(const float* dout, const float* out, float* dz, int64_t n, dpft_stream_t stream) {
    DPFT_REQUIRE(dout && out && dz && n > 0, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dout, out, dz, n);
    return check_launch("relu_bwd");
}

int dpft::add_inplace_any(float* a, const float* b, int64_t n, bool a16, dpft_stream_t stream) {
    DPFT_REQUIRE(a && b && n > 0 && (!a16 || n % 4 == 0), "add_inplace: bad arguments");
    if (a16) hipLaunchKernelGGL(add_inplace_kernel<__bf16>, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, n);
    else hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, n);
    return check_launch("add_inplace");
}

extern "C" int dpft_add_inplace_f32(float* a, const float* b, int64_t n, dpft_stream_t stream) {
    return dpft::add_inplace_any(a, b, n, false, stream);
}

int dpft::cvt_f32_to_bf16(const float* src, float* dst_bf16, int64_t n, dpft_stream_t stream) {
    DPFT_REQUIRE(src && dst_bf16 && n > 0 && n % 4 == 0, "cvt_f32_to_bf16: bad arguments");
    hipLaunchKernelGGL(cvt_f32_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, src, dst_bf16, n / 4);
    return check_launch("cvt_f32_to_bf16");
}

extern "C" int dpft_fpn_topdown_add_f32(float* lat, const float* top, int32_t B, int32_t H, int32_t W,
                                        int32_t TH, int32_t TW, int32_t K, dpft_stream_t stream) {
    DPFT_REQUIRE(lat && top && K % 4 == 0 && TH <= H && TW <= W, "fpn_topdown_add: bad arguments");
    const int64_t total = (int64_t)B * H * W * (K / 4);
    hipLaunchKernelGGL(fpn_topdown_add_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, lat, top, B,
                       H, W, TH, TW, K / 4, (float)TH / (float)H, (float)TW / (float)W);
    return check_launch("fpn_topdown_add");
}

extern "C" int dpft_fpn_topdown_add_bwd_f32(const float* dlat, float* dtop, int32_t B, int32_t H, int32_t W,
                                            int32_t TH, int32_t TW, int32_t K, dpft_stream_t stream) {
    DPFT_REQUIRE(dlat && dtop && K % 4 == 0 && TH <= H && TW <= W, "fpn_topdown_add_bwd: bad arguments");
    const int64_t total = (int64_t)B * TH * TW * (K / 4);
    hipLaunchKernelGGL(fpn_topdown_add_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dlat,
                       dtop, B, H, W, TH, TW, K / 4, (float)TH / (float)H, (float)TW / (float)W);
    return check_launch("fpn_topdown_add_bwd");
}

extern "C" int dpft_add_pos_f32(float* x, const float* pos_x, const float* pos_y, int32_t B, int32_t H,
                                int32_t W, int32_t K, dpft_stream_t stream) {
    DPFT_REQUIRE(x && pos_x && pos_y && K % 4 == 0, "add_pos: bad arguments");
    const int64_t total = (int64_t)B * H * W * (K / 4);
    hipLaunchKernelGGL(add_pos_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, pos_x, pos_y, B,
                       H, W, K / 4);
    return check_launch("add_pos");
}
