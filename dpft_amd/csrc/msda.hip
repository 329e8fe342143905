// Multi-scale deformable attention for gfx950.
//
// (1) dpft_msda_{fwd,bwd}_f32: operator-level drop-in for the Deformable-DETR CUDA extension the
//     reference binds at src/dprt/models/layers/ms_deform_attn.py:24,32-39,58-66 (same tensors,
//     same maths: SURVEY.md App. C).  Compatibility path.
//     Attribution: the extension is the multi-scale deformable attention operator of Deformable DETR (X. Zhu et al., ICLR
//     2021; https://github.com/fundamentalvision/Deformable-DETR, models/ops/src/cuda/ms_deform_im2col_cuda.cuh, Copyright
//     (c) 2020 SenseTime, Apache License 2.0), absent from the reference checkout.  The bilinear-sampling variable names of
//     the compatibility kernels below (h_im / w_im, lh / lw / hh / hw, v1..v4 and their gradient counterparts) follow that
//     operator's published kernel body so that the two can be compared term by term; the code is written for gfx950 here,
//     none of it is copied.
// (2) dpft_xattn_{fwd,bwd}_f32: the MI355X hot path.  "Sample-then-project": the bilinear gather
//     runs directly on the NHWC FPN levels (no flatten/cat, no dense value_proj, no `value`
//     tensor: src/dprt/models/fusers/mpfusion.py:179, ms_deform_attn.py:172), and the per-head
//     2x16 slice of value_proj is applied to the 16-channel sample afterwards, with the bias
//     weighted by the in-bounds bilinear mass (zero padding contributes 0, not the bias).
//     One 64-lane wave per (batch, query): lane = head*8 + channel-pair, so the 8 lanes of a head
//     read one 64-byte NHWC pixel per corner (fully used 64-B segments), 80 independent 8-byte
//     gathers in flight per lane, cross-lane reductions by DPP/shuffles inside the 8-lane group.
#include "common.h"

namespace dpft {

// ---------------------------------------------------------------------------------------------
// (1) generic operator
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void msda_fwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                        const float* __restrict__ attn, float* __restrict__ out,
                                                        int N, int S, int M, int D, int Lq, int L, int P) {
    const int64_t total = (int64_t)N * Lq * M * D;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % D);
    int64_t t = idx / D;
    const int m = (int)(t % M); t /= M;
    const int q = (int)(t % Lq);
    const int b = (int)(t / Lq);
    const int64_t lp = (((int64_t)b * Lq + q) * M + m) * L * P;
    float col = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float* base = value + (((int64_t)b * S + lsi[l]) * M + m) * D + c;
        const int64_t rs = (int64_t)M * D;  // stride between consecutive pixels
        for (int p = 0; p < P; ++p) {
            const float lw = loc[(lp + l * P + p) * 2 + 0], lh = loc[(lp + l * P + p) * 2 + 1];
            const float a = attn[lp + l * P + p];
            const float h_im = lh * H - 0.5f, w_im = lw * W - 0.5f;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh_ = h_im - h_lo, lw_ = w_im - w_lo, hh = 1 - lh_, hw = 1 - lw_;
                float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (h_lo >= 0 && w_lo >= 0) v1 = base[((int64_t)h_lo * W + w_lo) * rs];
                if (h_lo >= 0 && w_hi <= W - 1) v2 = base[((int64_t)h_lo * W + w_hi) * rs];
                if (h_hi <= H - 1 && w_lo >= 0) v3 = base[((int64_t)h_hi * W + w_lo) * rs];
                if (h_hi <= H - 1 && w_hi <= W - 1) v4 = base[((int64_t)h_hi * W + w_hi) * rs];
                col += a * (hh * hw * v1 + hh * lw_ * v2 + lh_ * hw * v3 + lh_ * lw_ * v4);
            }
        }
    }
    out[idx] = col;
}

// one thread per (b,q,m): loops channels so that grad_loc / grad_attn need no cross-thread reduce
__global__ __launch_bounds__(256) void msda_bwd_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                                                        const float* __restrict__ attn, const float* __restrict__ gout,
                                                        float* __restrict__ gvalue, float* __restrict__ gloc,
                                                        float* __restrict__ gattn, int N, int S, int M, int D, int Lq,
                                                        int L, int P) {
    const int64_t total = (int64_t)N * Lq * M;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t t = idx;
    const int m = (int)(t % M); t /= M;
    const int b = (int)(t / Lq);
    const int64_t lp = idx * L * P;
    const float* go = gout + idx * D;
    const int64_t rs = (int64_t)M * D;
    for (int l = 0; l < L; ++l) {
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int64_t boff = (((int64_t)b * S + lsi[l]) * M + m) * D;
        for (int p = 0; p < P; ++p) {
            const float lw = loc[(lp + l * P + p) * 2 + 0], lh = loc[(lp + l * P + p) * 2 + 1];
            const float a = attn[lp + l * P + p];
            const float h_im = lh * H - 0.5f, w_im = lw * W - 0.5f;
            float ga = 0.f, gw = 0.f, gh = 0.f;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh_ = h_im - h_lo, lw_ = w_im - w_lo, hh = 1 - lh_, hw = 1 - lw_;
                const bool k1 = h_lo >= 0 && w_lo >= 0, k2 = h_lo >= 0 && w_hi <= W - 1;
                const bool k3 = h_hi <= H - 1 && w_lo >= 0, k4 = h_hi <= H - 1 && w_hi <= W - 1;
                const int64_t o1 = boff + ((int64_t)h_lo * W + w_lo) * rs, o2 = boff + ((int64_t)h_lo * W + w_hi) * rs;
                const int64_t o3 = boff + ((int64_t)h_hi * W + w_lo) * rs, o4 = boff + ((int64_t)h_hi * W + w_hi) * rs;
                for (int c = 0; c < D; ++c) {
                    const float g = go[c], tg = g * a;
                    const float v1 = k1 ? value[o1 + c] : 0.f, v2 = k2 ? value[o2 + c] : 0.f;
                    const float v3 = k3 ? value[o3 + c] : 0.f, v4 = k4 ? value[o4 + c] : 0.f;
                    if (k1) atomicAdd(gvalue + o1 + c, hh * hw * tg);
                    if (k2) atomicAdd(gvalue + o2 + c, hh * lw_ * tg);
                    if (k3) atomicAdd(gvalue + o3 + c, lh_ * hw * tg);
                    if (k4) atomicAdd(gvalue + o4 + c, lh_ * lw_ * tg);
                    ga += g * (hh * hw * v1 + hh * lw_ * v2 + lh_ * hw * v3 + lh_ * lw_ * v4);
                    gh += tg * (-hw * v1 - lw_ * v2 + hw * v3 + lw_ * v4);
                    gw += tg * (-hh * v1 + hh * v2 - lh_ * v3 + lh_ * v4);
                }
            }
            gattn[lp + l * P + p] = ga;
            gloc[(lp + l * P + p) * 2 + 0] = W * gw;
            gloc[(lp + l * P + p) * 2 + 1] = H * gh;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// (2) fused sample-then-project cross attention, M = 8 heads x D = 2 (C = 16)
// ---------------------------------------------------------------------------------------------
struct Pyr {
    const float* level[DPFT_MAX_LEVELS];
    float* grad[DPFT_MAX_LEVELS];
    int H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int L;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// sum over the 8 lanes of a head group (lane bits 0..2)
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group8_sum(v);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

constexpr int XM = 8, XD = 2, XC = 16;

__global__ __launch_bounds__(256) void xattn_fwd_kernel(Pyr pyr, const float* __restrict__ ref, const float* __restrict__ off,
                                                         const float* __restrict__ attn, const float* __restrict__ Wv,
                                                         const float* __restrict__ bv, float* __restrict__ out,
                                                         float* __restrict__ samp, float* __restrict__ mass, int B, int Q,
                                                         int P) {
    const int lane = threadIdx.x & 63;
    const int bq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bq >= B * Q) return;
    const int b = bq / Q;
    const int m = lane >> 3, j = lane & 7;
    const int L = pyr.L;
    const float rx = ref[bq * 2 + 0], ry = ref[bq * 2 + 1];
    const float* offp = off + ((int64_t)bq * XM + m) * L * P * 2;
    const float* attp = attn + ((int64_t)bq * XM + m) * L * P;
    f32x2 acc = {0.f, 0.f};
    float ms = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const float* base = pyr.level[l] + (int64_t)b * H * W * XC + j * 2;
        for (int p = 0; p < P; ++p) {
            const float ox = offp[(l * P + p) * 2 + 0], oy = offp[(l * P + p) * 2 + 1];
            const float a = attp[l * P + p];
            const float lx = rx + ox / (float)W, ly = ry + oy / (float)H;   // ms_deform_attn.py:186-191
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1 - lh, hw = 1 - lw;
                const bool k1 = h_lo >= 0 && w_lo >= 0, k2 = h_lo >= 0 && w_hi <= W - 1;
                const bool k3 = h_hi <= H - 1 && w_lo >= 0, k4 = h_hi <= H - 1 && w_hi <= W - 1;
                f32x2 v1 = {0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (k1) v1 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_lo * W + w_lo) * XC);
                if (k2) v2 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_lo * W + w_hi) * XC);
                if (k3) v3 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_hi * W + w_lo) * XC);
                if (k4) v4 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_hi * W + w_hi) * XC);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                acc += a * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                ms += a * ((k1 ? w1 : 0.f) + (k2 ? w2 : 0.f) + (k3 ? w3 : 0.f) + (k4 ? w4 : 0.f));
            }
        }
    }
    // save raw samples for backward: samp[bq][m][c], mass[bq][m]
    *reinterpret_cast<f32x2*>(samp + ((int64_t)bq * XM + m) * XC + j * 2) = acc;
    if (j == 0) mass[bq * XM + m] = ms;
    // out[m*D+d] = Wv[m*D+d,:] . samp[m,:] + bv[m*D+d]*mass[m]
    const f32x2 w0 = *reinterpret_cast<const f32x2*>(Wv + (m * XD + 0) * XC + j * 2);
    const f32x2 w1v = *reinterpret_cast<const f32x2*>(Wv + (m * XD + 1) * XC + j * 2);
    float o0 = group8_sum(w0[0] * acc[0] + w0[1] * acc[1]);
    float o1 = group8_sum(w1v[0] * acc[0] + w1v[1] * acc[1]);
    if (j == 0) {
        f32x2 o = {o0 + bv[m * XD + 0] * ms, o1 + bv[m * XD + 1] * ms};
        *reinterpret_cast<f32x2*>(out + (int64_t)bq * XC + m * XD) = o;
    }
}

__global__ __launch_bounds__(256) void xattn_bwd_kernel(Pyr pyr, const float* __restrict__ ref, const float* __restrict__ off,
                                                         const float* __restrict__ attn, const float* __restrict__ Wv,
                                                         const float* __restrict__ bv, const float* __restrict__ gout,
                                                         float* __restrict__ goff, float* __restrict__ gattn,
                                                         float* __restrict__ gref, int B, int Q, int P) {
    const int lane = threadIdx.x & 63;
    const int bq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bq >= B * Q) return;
    const int b = bq / Q;
    const int m = lane >> 3, j = lane & 7;
    const int L = pyr.L;
    const float rx = ref[bq * 2 + 0], ry = ref[bq * 2 + 1];
    const float* offp = off + ((int64_t)bq * XM + m) * L * P * 2;
    const float* attp = attn + ((int64_t)bq * XM + m) * L * P;
    float* goffp = goff + ((int64_t)bq * XM + m) * L * P * 2;
    float* gattp = gattn + ((int64_t)bq * XM + m) * L * P;
    // d samp[m][2j..2j+1] and d mass[m]
    const float g0 = gout[(int64_t)bq * XC + m * XD + 0], g1 = gout[(int64_t)bq * XC + m * XD + 1];
    const f32x2 w0 = *reinterpret_cast<const f32x2*>(Wv + (m * XD + 0) * XC + j * 2);
    const f32x2 w1v = *reinterpret_cast<const f32x2*>(Wv + (m * XD + 1) * XC + j * 2);
    const f32x2 dS = {w0[0] * g0 + w1v[0] * g1, w0[1] * g0 + w1v[1] * g1};
    const float dM = bv[m * XD + 0] * g0 + bv[m * XD + 1] * g1;
    float grx = 0.f, gry = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const int64_t lb = (int64_t)b * H * W * XC + j * 2;
        const float* base = pyr.level[l] + lb;
        float* gbase = pyr.grad[l] + lb;
        for (int p = 0; p < P; ++p) {
            const float ox = offp[(l * P + p) * 2 + 0], oy = offp[(l * P + p) * 2 + 1];
            const float a = attp[l * P + p];
            const float lx = rx + ox / (float)W, ly = ry + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            float ga = 0.f, gw = 0.f, gh = 0.f;   // partial over this lane's 2 channels (+ the mass term on j==0)
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1 - lh, hw = 1 - lw;
                const bool k1 = h_lo >= 0 && w_lo >= 0, k2 = h_lo >= 0 && w_hi <= W - 1;
                const bool k3 = h_hi <= H - 1 && w_lo >= 0, k4 = h_hi <= H - 1 && w_hi <= W - 1;
                const int64_t o1 = ((int64_t)h_lo * W + w_lo) * XC, o2 = ((int64_t)h_lo * W + w_hi) * XC;
                const int64_t o3 = ((int64_t)h_hi * W + w_lo) * XC, o4 = ((int64_t)h_hi * W + w_hi) * XC;
                f32x2 v1 = {0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (k1) v1 = *reinterpret_cast<const f32x2*>(base + o1);
                if (k2) v2 = *reinterpret_cast<const f32x2*>(base + o2);
                if (k3) v3 = *reinterpret_cast<const f32x2*>(base + o3);
                if (k4) v4 = *reinterpret_cast<const f32x2*>(base + o4);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                // feature gradients: a * w_k * dS
                if (k1) { atomicAdd(gbase + o1, a * w1 * dS[0]); atomicAdd(gbase + o1 + 1, a * w1 * dS[1]); }
                if (k2) { atomicAdd(gbase + o2, a * w2 * dS[0]); atomicAdd(gbase + o2 + 1, a * w2 * dS[1]); }
                if (k3) { atomicAdd(gbase + o3, a * w3 * dS[0]); atomicAdd(gbase + o3 + 1, a * w3 * dS[1]); }
                if (k4) { atomicAdd(gbase + o4, a * w4 * dS[0]); atomicAdd(gbase + o4 + 1, a * w4 * dS[1]); }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ga += dS[e] * (w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e]);
                    gh += dS[e] * a * (-hw * v1[e] - lw * v2[e] + hw * v3[e] + lw * v4[e]);
                    gw += dS[e] * a * (-hh * v1[e] + hh * v2[e] - lh * v3[e] + lh * v4[e]);
                }
                if (j == 0) {  // the "17th channel": 1 at in-bounds pixels, carries the value_proj bias
                    const float i1 = k1 ? 1.f : 0.f, i2 = k2 ? 1.f : 0.f, i3 = k3 ? 1.f : 0.f, i4 = k4 ? 1.f : 0.f;
                    ga += dM * (w1 * i1 + w2 * i2 + w3 * i3 + w4 * i4);
                    gh += dM * a * (-hw * i1 - lw * i2 + hw * i3 + lw * i4);
                    gw += dM * a * (-hh * i1 + hh * i2 - lh * i3 + lh * i4);
                }
            }
            ga = group8_sum(ga);
            gw = group8_sum(gw);
            gh = group8_sum(gh);
            // loc = ref + off / (W,H);  w_im = loc_x*W - 0.5  =>  d/d off_x = gw * W / W, d/d ref_x = gw * W
            const float glx = (float)W * gw, gly = (float)H * gh;
            if (j == 0) {
                gattp[l * P + p] = ga;
                goffp[(l * P + p) * 2 + 0] = glx / (float)W;
                goffp[(l * P + p) * 2 + 1] = gly / (float)H;
                grx += glx;
                gry += gly;
            }
        }
    }
    if (gref) {
        grx = wave_sum(grx);   // non-leader lanes hold 0
        gry = wave_sum(gry);
        if (lane == 0) {
            gref[bq * 2 + 0] = grx;
            gref[bq * 2 + 1] = gry;
        }
    }
}

static int copy_pyr(Pyr& k, const dpft_pyramid* p, bool need_grad) {
    DPFT_REQUIRE(p && p->L > 0 && p->L <= DPFT_MAX_LEVELS, "xattn: bad pyramid");
    k.L = p->L;
    for (int l = 0; l < p->L; ++l) {
        DPFT_REQUIRE(p->level[l] && p->H[l] > 0 && p->W[l] > 0, "xattn: bad level %d", l);
        DPFT_REQUIRE(!need_grad || p->grad[l], "xattn bwd: level %d has no grad buffer", l);
        k.level[l] = p->level[l];
        k.grad[l] = p->grad[l];
        k.H[l] = p->H[l];
        k.W[l] = p->W[l];
    }
    return DPFT_OK;
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_msda_fwd_f32(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                                 const float* attn, float* out, int32_t N, int32_t S, int32_t M, int32_t D,
                                 int32_t Lq, int32_t L, int32_t P, dpft_stream_t stream) {
    DPFT_REQUIRE(value && shapes && lsi && loc && attn && out, "msda_fwd: null tensor");
    DPFT_REQUIRE(N > 0 && S > 0 && M > 0 && D > 0 && Lq > 0 && L > 0 && P > 0, "msda_fwd: non-positive dims");
    const int64_t total = (int64_t)N * Lq * M * D;
    hipLaunchKernelGGL(msda_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, value, shapes, lsi,
                       loc, attn, out, N, S, M, D, Lq, L, P);
    return check_launch("msda_fwd");
}

extern "C" int dpft_msda_bwd_f32(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc,
                                 const float* attn, const float* grad_out, float* grad_value, float* grad_loc,
                                 float* grad_attn, int32_t N, int32_t S, int32_t M, int32_t D, int32_t Lq, int32_t L,
                                 int32_t P, dpft_stream_t stream) {
    DPFT_REQUIRE(value && shapes && lsi && loc && attn && grad_out && grad_value && grad_loc && grad_attn,
                 "msda_bwd: null tensor");
    const int64_t total = (int64_t)N * Lq * M;
    hipLaunchKernelGGL(msda_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, value, shapes, lsi,
                       loc, attn, grad_out, grad_value, grad_loc, grad_attn, N, S, M, D, Lq, L, P);
    return check_launch("msda_bwd");
}

extern "C" int dpft_xattn_fwd_f32(const dpft_pyramid* pyr, const float* ref, const float* off, const float* attn,
                                  const float* Wv, const float* bv, float* out, float* samp, float* mass, int32_t B,
                                  int32_t Q, int32_t M, int32_t D, int32_t P, dpft_stream_t stream) {
    DPFT_REQUIRE(M == XM && D == XD, "xattn: fused kernel supports M=8 heads x D=2 only (got %d x %d)", M, D);
    DPFT_REQUIRE(ref && off && attn && Wv && bv && out && samp && mass && B > 0 && Q > 0 && P > 0, "xattn_fwd: bad arguments");
    Pyr k;
    int rc = copy_pyr(k, pyr, false);
    if (rc) return rc;
    hipLaunchKernelGGL(xattn_fwd_kernel, dim3(cdiv((int64_t)B * Q, 4)), dim3(256), 0, (hipStream_t)stream, k, ref, off,
                       attn, Wv, bv, out, samp, mass, B, Q, P);
    return check_launch("xattn_fwd");
}

extern "C" int dpft_xattn_bwd_f32(const dpft_pyramid* pyr, const float* ref, const float* off, const float* attn,
                                  const float* Wv, const float* bv, const float* grad_out, float* grad_off,
                                  float* grad_attn, float* grad_ref, int32_t B, int32_t Q, int32_t M, int32_t D,
                                  int32_t P, dpft_stream_t stream) {
    DPFT_REQUIRE(M == XM && D == XD, "xattn: fused kernel supports M=8 heads x D=2 only (got %d x %d)", M, D);
    DPFT_REQUIRE(ref && off && attn && Wv && bv && grad_out && grad_off && grad_attn, "xattn_bwd: bad arguments");
    Pyr k;
    int rc = copy_pyr(k, pyr, true);
    if (rc) return rc;
    hipLaunchKernelGGL(xattn_bwd_kernel, dim3(cdiv((int64_t)B * Q, 4)), dim3(256), 0, (hipStream_t)stream, k, ref, off,
                       attn, Wv, bv, grad_out, grad_off, grad_attn, grad_ref, B, Q, P);
    return check_launch("xattn_bwd");
}
