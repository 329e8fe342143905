// Training path of the fusion decoder's self-attention block (all views of one MPFusion layer in one launch):
//   y1[v] = LayerNorm1(x + dropout1(out_proj(MHA(x + pos, x + pos, x))))       (attention-probability dropout inside)
// = MLFusion.forward_self_attn of the reference, src/dprt/models/fusers/mpfusion.py:122-148 (nn.MultiheadAttention
// with dropout, batch_first; d_model 16, 8 heads x head_dim 2).  Forward + hand-written backward (2 kernels):
// eager autograd runs ~45 forward and ~90 backward launches of (B*400 x 16)-sized ops per layer for the same math.
//
// Dropout masks are not stored: both passes regenerate them from a counter-based hash of
// (seed, view, batch, head, query, key) -- 16 random bits per decision, keep <=> bits >= p * 65536.
//
// Kernel shapes (Q = 400 queries, K/V of all keys live in LDS, scores never touch HBM):
//   sa_train_fwd   : block = QT = 4*QW queries of one (b, view); lane = (head, key slice); online softmax
//   sa_train_bwd_q : same tiling; LayerNorm/out_proj backward, dQ (loop over keys), parameter-gradient partials
//   sa_train_bwd_kv: block = KT = 4*KW keys of one (b, view); Q', dO, lse, delta of ALL queries in LDS; dK, dV
//                    (loop over queries), in_proj backward of the key/value rows
// QW/KW are picked on the host so that each grid is ONE round of <= 256 blocks (every block recomputes the
// projections of all 400 rows, so a second partial round would double the kernel time).
#include "common.h"

namespace dpft {

constexpr int TC = 16, TH = 8;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SaArgs {
    dpft_sa_params p[4];
    dpft_sa_grads g[4];
    const float* x;          // (B,Q,16) with batch stride xbs (0 = broadcast)
    const float* pos;        // (Q,16)
    const int64_t* seed;
    float* y1;               // (V,B,Q,16)
    float* lse;              // (V,B,Q,8)   log-sum-exp of the scaled scores
    float* attn;             // (V,B,Q,16)  attention output (after dropout), input of out_proj
    float* zhat;             // (V,B,Q,16)  normalised pre-affine LayerNorm1 input
    float* rstd;             // (V,B,Q)
    const float* dy1;        // (V,B,Q,16)
    float* dx;               // (V,B,Q,16)  d/dx of view v (residual + value path + q/k path)
    float* dxp;              // (V,B,Q,16)  d/d(x+pos) of view v (q/k path) -> summed into pos.grad by the caller
    float* dA;               // (V,B,Q,16)  scratch: gradient of the attention output
    float* delta;            // (V,B,Q,8)   scratch: dO . O per head
    long xbs;
    int B, Q, V, salt;
    float p_drop;
};

__device__ __forceinline__ uint32_t drop_hash(uint32_t idx, uint32_t s0, uint32_t s1) {
    uint32_t x = idx ^ s0;
    x *= 0xcc9e2d51u; x = (x << 15) | (x >> 17); x *= 0x1b873593u;
    x ^= s1;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
struct DropCtx {
    uint32_t s0, s1, thr;
    float inv;
};
__device__ __forceinline__ DropCtx drop_ctx(const int64_t* seed, int salt, int stream_id, float p) {
    const uint64_t s = (uint64_t)(*seed);
    DropCtx d;
    d.s0 = (uint32_t)s ^ ((uint32_t)salt * 0x9E3779B9u);
    d.s1 = (uint32_t)(s >> 32) + (uint32_t)stream_id * 0x7F4A7C15u;
    d.thr = (uint32_t)(p * 65536.f + 0.5f);
    d.inv = 1.f / (1.f - p);
    return d;
}
__device__ __forceinline__ bool drop_keep(uint32_t hash, int which, uint32_t thr) {
    return ((hash >> (16 * which)) & 0xFFFFu) >= thr;
}
// attention-probability mask: one hash per key pair (k, k+8) inside a 16-key group
__device__ __forceinline__ uint32_t attn_pair_index(int vb, int h, int q, int Q, int KP, int k) {
    return (uint32_t)(((vb * TH + h) * Q + q) * KP + (k >> 4) * 8 + (k & 7));
}

__device__ __forceinline__ float g16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float g8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// rows [row0, row0+16) of in_proj applied to x[16] -> dst[16] (times scale); weights staged in LDS
__device__ __forceinline__ void in_proj_rows(const float* Ws, int row0, const float* x, float scale, float* dst) {
#pragma unroll 4
    for (int o = 0; o < TC; ++o) {
        const float* wr = Ws + (row0 + o) * 16;
        float s = Ws[48 * 16 + row0 + o];
#pragma unroll
        for (int c = 0; c < TC; c += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + c);
            s = fmaf(w4[0], x[c], s); s = fmaf(w4[1], x[c + 1], s);
            s = fmaf(w4[2], x[c + 2], s); s = fmaf(w4[3], x[c + 3], s);
        }
        dst[o] = s * scale;
    }
}
__device__ __forceinline__ void load_row(const float* xrow, const float* prow, float* x) {
#pragma unroll
    for (int c = 0; c < TC; c += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xrow + c);
        if (prow) v += *reinterpret_cast<const f32x4*>(prow + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[c + e] = v[e];
    }
}

constexpr float kQScale = 0.70710678118654752f;   // 1/sqrt(head_dim = 2), folded into Q'

template <int QW>
__global__ __launch_bounds__(256) void sa_train_fwd_kernel(SaArgs a) {
    constexpr int QT = 4 * QW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q;
    float* Ks = sm;                   // [Q][16]
    float* Vs = Ks + Q * TC;          // [Q][16]
    float* Ws = Vs + Q * TC;          // in_proj rows [48][16] + bias [48]
    float* Qs = Ws + 48 * 16 + 48;    // [QT][16] scaled Q'; reused as the attention output tile
    float* Pt = Qs + QT * TC;         // [QT][8 heads][8 slices][4] partial (max, den, o0, o1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * QT;
    const int vb = view * a.B + b;
    const dpft_sa_params& p = a.p[view];
    const float* xb = a.x + (size_t)b * a.xbs;
    if (tid < 768 / 4) *reinterpret_cast<f32x4*>(Ws + tid * 4) = *reinterpret_cast<const f32x4*>(p.in_w + tid * 4);
    if (tid < 48) Ws[768 + tid] = p.in_b[tid];
    __syncthreads();
    for (int i = tid; i < 2 * Q + QT; i += 256) {
        const int isq = i >= 2 * Q;
        const int k = isq ? min(q0 + i - 2 * Q, Q - 1) : (i >> 1);
        const int isv = isq ? 0 : (i & 1);
        float x[TC];
        load_row(xb + (size_t)k * TC, isv ? nullptr : a.pos + (size_t)k * TC, x);
        float* dst = isq ? Qs + (i - 2 * Q) * TC : (isv ? Vs : Ks) + k * TC;
        in_proj_rows(Ws, isq ? 0 : 16 + 16 * isv, x, isq ? kQScale : 1.f, dst);
    }
    __syncthreads();
    const DropCtx dc = drop_ctx(a.seed, a.salt, 0, a.p_drop);
    const int KP = ((Q + 15) >> 4) * 8;
    {
        const int slice = lane & 7, h = lane >> 3;
        f32x2 qh[QW];
        float mx[QW], den[QW], o0[QW], o1[QW];
        uint32_t hbase[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            qh[i] = *reinterpret_cast<const f32x2*>(Qs + (wv * QW + i) * TC + h * 2);
            mx[i] = -INFINITY; den[i] = 0.f; o0[i] = 0.f; o1[i] = 0.f;
            hbase[i] = attn_pair_index(vb, h, min(q0 + wv * QW + i, Q - 1), Q, KP, 0);
        }
        const float* kp = Ks + h * 2;
        const float* vp = Vs + h * 2;
#pragma unroll 1
        for (int k = slice; k < Q; k += 16) {      // keys k and k+8: one mask hash per pair
            const int k2 = k + 8;
            const bool has2 = k2 < Q;
            const f32x2 ka = *reinterpret_cast<const f32x2*>(kp + k * TC), va = *reinterpret_cast<const f32x2*>(vp + k * TC);
            const f32x2 kb = *reinterpret_cast<const f32x2*>(kp + (has2 ? k2 : k) * TC);
            const f32x2 vb2 = *reinterpret_cast<const f32x2*>(vp + (has2 ? k2 : k) * TC);
            const uint32_t pidx = (uint32_t)((k >> 4) * 8 + slice);
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const float sa = qh[i][0] * ka[0] + qh[i][1] * ka[1];
                const float sb = has2 ? qh[i][0] * kb[0] + qh[i][1] * kb[1] : -INFINITY;
                const float m_new = fmaxf(fmaxf(sa, sb), mx[i]);
                const float corr = __expf(mx[i] - m_new);
                const float pa = __expf(sa - m_new), pb = __expf(sb - m_new);
                den[i] = den[i] * corr + pa + pb;
                const uint32_t hs = drop_hash(hbase[i] + pidx, dc.s0, dc.s1);
                const float ma = drop_keep(hs, 0, dc.thr) ? pa : 0.f, mb = drop_keep(hs, 1, dc.thr) ? pb : 0.f;
                o0[i] = fmaf(mb, vb2[0], fmaf(ma, va[0], o0[i] * corr));
                o1[i] = fmaf(mb, vb2[1], fmaf(ma, va[1], o1[i] * corr));
                mx[i] = m_new;
            }
        }
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const f32x4 part = {mx[i], den[i], o0[i], o1[i]};
            *reinterpret_cast<f32x4*>(Pt + (((wv * QW + i) * TH + h) * 8 + slice) * 4) = part;
        }
    }
    __syncthreads();
    if (tid < QT * TH) {      // merge the 8 key slices of (query, head); a slice may be empty (max = -inf)
        const float* pp = Pt + tid * 32;
        float mm = -INFINITY;
#pragma unroll 1
        for (int s2 = 0; s2 < 8; ++s2) mm = fmaxf(mm, pp[s2 * 4]);
        float dsum = 0.f, a0 = 0.f, a1 = 0.f;
#pragma unroll 1
        for (int s2 = 0; s2 < 8; ++s2) {
            const f32x4 part = *reinterpret_cast<const f32x4*>(pp + s2 * 4);
            const float cf = part[0] == -INFINITY ? 0.f : __expf(part[0] - mm);
            dsum = fmaf(part[1], cf, dsum);
            a0 = fmaf(part[2], cf, a0);
            a1 = fmaf(part[3], cf, a1);
        }
        const float sc = dc.inv / dsum;
        Qs[tid * 2 + 0] = a0 * sc;      // tid = query * 8 + head  ->  channel head*2 + d
        Qs[tid * 2 + 1] = a1 * sc;
        const int q = q0 + (tid >> 3);
        if (q < Q) a.lse[((size_t)vb * Q + q) * TH + (tid & 7)] = mm + __logf(dsum);
    }
    __syncthreads();
    for (int t = tid; t < QT * TC; t += 256) {      // out_proj + dropout1 + residual + LayerNorm1: 16 lanes per query
        const int ql = t >> 4, c = t & 15;      // (QT * 16 is a multiple of 64: whole waves stay active)
        const int q = q0 + ql;
        const bool ok = q < Q;
        const size_t row = (size_t)vb * Q + (ok ? q : Q - 1);
        const float at = Qs[ql * TC + c];
        float v = p.out_b[c];
#pragma unroll 4
        for (int j = 0; j < TC; ++j) v = fmaf(p.out_w[c * TC + j], Qs[ql * TC + j], v);
        const DropCtx d1 = drop_ctx(a.seed, a.salt, 1, a.p_drop);
        const uint32_t hs = drop_hash((uint32_t)(row * 8 + (c >> 1)), d1.s0, d1.s1);
        v = drop_keep(hs, c & 1, d1.thr) ? v * d1.inv : 0.f;
        v += xb[(size_t)(ok ? q : Q - 1) * TC + c];
        const float mean = g16_sum(v) * (1.f / 16.f);
        const float dlt = v - mean;
        const float var = g16_sum(dlt * dlt) * (1.f / 16.f);
        const float rs = 1.0f / sqrtf(var + 1e-5f);
        const float zh = dlt * rs;
        if (ok) {
            a.attn[row * TC + c] = at;
            a.zhat[row * TC + c] = zh;
            if (c == 0) a.rstd[row] = rs;
            a.y1[row * TC + c] = zh * p.n1_w[c] + p.n1_b[c];
        }
    }
}

// LayerNorm1 / dropout1 / out_proj backward, dQ' (loop over keys), in_proj backward of the query rows
template <int QW>
__global__ __launch_bounds__(256) void sa_train_bwd_q_kernel(SaArgs a) {
    constexpr int QT = 4 * QW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q;
    float* Ks = sm;                   // [Q][16]
    float* Vs = Ks + Q * TC;          // [Q][16]
    float* Ws = Vs + Q * TC;          // in_proj [48][16] + bias [48]
    float* Qs = Ws + 48 * 16 + 48;    // [QT][16] scaled Q'
    float* Xp = Qs + QT * TC;         // [QT][16] x + pos of the tile
    float* dYs = Xp + QT * TC;        // [QT][16] grad of out_proj output (after dropout1 backward)
    float* As = dYs + QT * TC;        // [QT][16] attention output
    float* dAs = As + QT * TC;        // [QT][16]
    float* dQs = dAs + QT * TC;       // [QT][16] grad of the unscaled q projection
    float* dl = dQs + QT * TC;        // [QT][8]
    float* R = dl + QT * TH;          // [2][QT][16] LayerNorm gradient partials; later [QT][8][8][2] dq partials
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * QT;
    const int vb = view * a.B + b;
    const dpft_sa_params& p = a.p[view];
    const dpft_sa_grads& g = a.g[view];
    const float* xb = a.x + (size_t)b * a.xbs;
    if (tid < 768 / 4) *reinterpret_cast<f32x4*>(Ws + tid * 4) = *reinterpret_cast<const f32x4*>(p.in_w + tid * 4);
    if (tid < 48) Ws[768 + tid] = p.in_b[tid];
    __syncthreads();
    for (int i = tid; i < 2 * Q + QT; i += 256) {
        const int isq = i >= 2 * Q;
        const int k = isq ? min(q0 + i - 2 * Q, Q - 1) : (i >> 1);
        const int isv = isq ? 0 : (i & 1);
        float x[TC];
        load_row(xb + (size_t)k * TC, isv ? nullptr : a.pos + (size_t)k * TC, x);
        if (isq) {
#pragma unroll
            for (int c = 0; c < TC; ++c) Xp[(i - 2 * Q) * TC + c] = x[c];
        }
        float* dst = isq ? Qs + (i - 2 * Q) * TC : (isv ? Vs : Ks) + k * TC;
        in_proj_rows(Ws, isq ? 0 : 16 + 16 * isv, x, isq ? kQScale : 1.f, dst);
    }
    // LayerNorm1 + dropout1 backward: thread = (query, channel)
    constexpr int NIT = (QT * TC + 255) / 256;      // (query, channel) items per thread
    float dZr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) dZr[it] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int t = tid + it * 256;
        if (t >= QT * TC) break;
        const int ql = t >> 4, c = t & 15;
        const int q = q0 + ql;
        const bool ok = q < Q;
        const size_t row = (size_t)vb * Q + (ok ? q : Q - 1);
        float dZ;
        const float dy = ok ? a.dy1[row * TC + c] : 0.f;
        const float zh = a.zhat[row * TC + c];
        const float dzh = dy * p.n1_w[c];
        const float m1 = g16_sum(dzh) * (1.f / 16.f), m2 = g16_sum(dzh * zh) * (1.f / 16.f);
        dZ = ok ? a.rstd[row] * (dzh - m1 - zh * m2) : 0.f;
        R[ql * TC + c] = dy * zh;
        R[QT * TC + ql * TC + c] = dy;
        const DropCtx d1 = drop_ctx(a.seed, a.salt, 1, a.p_drop);
        const uint32_t hs = drop_hash((uint32_t)(row * 8 + (c >> 1)), d1.s0, d1.s1);
        dYs[ql * TC + c] = drop_keep(hs, c & 1, d1.thr) ? dZ * d1.inv : 0.f;
        As[ql * TC + c] = ok ? a.attn[row * TC + c] : 0.f;
        dZr[it] = dZ;
    }
    __syncthreads();
    for (int t = tid; t < QT * TC; t += 256) {      // dA = dY Wo ; delta = dA . A per head
        const int ql = t >> 4, j = t & 15;
        const int q = q0 + ql;
        float s = 0.f;
#pragma unroll 4
        for (int c = 0; c < TC; ++c) s = fmaf(dYs[ql * TC + c], p.out_w[c * TC + j], s);
        dAs[ql * TC + j] = s;
        float pr = s * As[ql * TC + j];
        pr += __shfl_xor(pr, 1);
        if ((j & 1) == 0) dl[ql * TH + (j >> 1)] = pr;
        if (q < Q) {
            a.dA[((size_t)vb * Q + q) * TC + j] = s;
            if ((j & 1) == 0) a.delta[((size_t)vb * Q + q) * TH + (j >> 1)] = pr;
        }
    }
    {   // parameter gradients of out_proj (thread = (c, j)) and LayerNorm1 (threads 0..31)
        const int c = tid >> 4, j = tid & 15;
        float s = 0.f, sb = 0.f;
#pragma unroll 1
        for (int ql = 0; ql < QT; ++ql) {
            s = fmaf(dYs[ql * TC + c], As[ql * TC + j], s);
            sb += dYs[ql * TC + c];
        }
        atomicAdd(g.out_w + c * TC + j, s);
        if (j == 0) atomicAdd(g.out_b + c, sb);
        if (tid < 32) {
            const int which = tid >> 4, cc = tid & 15;
            float r = 0.f;
#pragma unroll 1
            for (int ql = 0; ql < QT; ++ql) r += R[which * QT * TC + ql * TC + cc];
            atomicAdd((which ? g.n1_b : g.n1_w) + cc, r);
        }
    }
    __syncthreads();
    const DropCtx dc = drop_ctx(a.seed, a.salt, 0, a.p_drop);
    const int KP = ((Q + 15) >> 4) * 8;
    {   // dQ'[q][h] = sum_k dS K_k, dS = P (dP - delta); lane = (head, key slice), QW queries in registers
        const int slice = lane & 7, h = lane >> 3;
        f32x2 qh[QW], dO[QW];
        float ls[QW], de[QW], dq0[QW], dq1[QW];
        uint32_t hbase[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const int ql = wv * QW + i, q = min(q0 + ql, Q - 1);
            qh[i] = *reinterpret_cast<const f32x2*>(Qs + ql * TC + h * 2);
            dO[i] = *reinterpret_cast<const f32x2*>(dAs + ql * TC + h * 2);
            ls[i] = a.lse[((size_t)vb * Q + q) * TH + h];
            de[i] = dl[ql * TH + h];
            dq0[i] = 0.f; dq1[i] = 0.f;
            hbase[i] = attn_pair_index(vb, h, q, Q, KP, 0);
        }
        const float* kp = Ks + h * 2;
        const float* vp = Vs + h * 2;
#pragma unroll 1
        for (int k = slice; k < Q; k += 16) {
            const int k2 = k + 8;
            const bool has2 = k2 < Q;
            const f32x2 ka = *reinterpret_cast<const f32x2*>(kp + k * TC), va = *reinterpret_cast<const f32x2*>(vp + k * TC);
            const f32x2 kb = *reinterpret_cast<const f32x2*>(kp + (has2 ? k2 : k) * TC);
            const f32x2 vb2 = *reinterpret_cast<const f32x2*>(vp + (has2 ? k2 : k) * TC);
            const uint32_t pidx = (uint32_t)((k >> 4) * 8 + slice);
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const float pa = __expf(qh[i][0] * ka[0] + qh[i][1] * ka[1] - ls[i]);
                const float pb = has2 ? __expf(qh[i][0] * kb[0] + qh[i][1] * kb[1] - ls[i]) : 0.f;
                const uint32_t hs = drop_hash(hbase[i] + pidx, dc.s0, dc.s1);
                const float dpa = drop_keep(hs, 0, dc.thr) ? (dO[i][0] * va[0] + dO[i][1] * va[1]) * dc.inv : 0.f;
                const float dpb = drop_keep(hs, 1, dc.thr) ? (dO[i][0] * vb2[0] + dO[i][1] * vb2[1]) * dc.inv : 0.f;
                const float dsa = pa * (dpa - de[i]), dsb = pb * (dpb - de[i]);
                dq0[i] = fmaf(dsb, kb[0], fmaf(dsa, ka[0], dq0[i]));
                dq1[i] = fmaf(dsb, kb[1], fmaf(dsa, ka[1], dq1[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const float s0 = g8_sum(dq0[i]), s1 = g8_sum(dq1[i]);
            if (slice == 0) {
                const int ql = wv * QW + i;
                const bool ok = q0 + ql < Q;
                dQs[ql * TC + h * 2 + 0] = ok ? s0 * kQScale : 0.f;
                dQs[ql * TC + h * 2 + 1] = ok ? s1 * kQScale : 0.f;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {      // in_proj backward of the query rows -> d(x+pos); dx = residual + q path
        const int t = tid + it * 256;       // (the k / v paths are added by bwd_kv)
        if (t >= QT * TC) break;
        const int ql = t >> 4, c = t & 15;
        const int q = q0 + ql;
        float s = 0.f;
#pragma unroll 4
        for (int o = 0; o < TC; ++o) s = fmaf(dQs[ql * TC + o], Ws[o * 16 + c], s);
        if (q < Q) {
            const size_t row = (size_t)vb * Q + q;
            a.dxp[row * TC + c] = s;
            a.dx[row * TC + c] = dZr[it] + s;
        }
    }
    {
        const int o = tid >> 4, c = tid & 15;
        float s = 0.f, sb = 0.f;
#pragma unroll 1
        for (int ql = 0; ql < QT; ++ql) {
            s = fmaf(dQs[ql * TC + o], Xp[ql * TC + c], s);
            sb += dQs[ql * TC + o];
        }
        atomicAdd(g.in_w + o * TC + c, s);
        if (c == 0) atomicAdd(g.in_b + o, sb);
    }
}

// dK, dV (loop over all queries) and the in_proj backward of the key / value rows
template <int KW>
__global__ __launch_bounds__(256) void sa_train_bwd_kv_kernel(SaArgs a) {
    constexpr int KT = 4 * KW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q;
    float* Qs = sm;                    // [Q][16] scaled Q' of every query
    float* dAs = Qs + Q * TC;          // [Q][16]
    float* Ls = dAs + Q * TC;          // [Q][8] lse
    float* Ds = Ls + Q * TH;           // [Q][8] delta
    float* Ws = Ds + Q * TH;           // in_proj
    float* Kt = Ws + 48 * 16 + 48;     // [KT][16]
    float* Vt = Kt + KT * TC;          // [KT][16]
    float* Xk = Vt + KT * TC;          // [KT][16] x + pos
    float* Xv = Xk + KT * TC;          // [KT][16] x
    float* dKs = Xv + KT * TC;         // [KT][16]
    float* dVs = dKs + KT * TC;        // [KT][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y, b = blockIdx.z, k0 = blockIdx.x * KT;
    const int vb = view * a.B + b;
    const dpft_sa_params& p = a.p[view];
    const dpft_sa_grads& g = a.g[view];
    const float* xb = a.x + (size_t)b * a.xbs;
    if (tid < 768 / 4) *reinterpret_cast<f32x4*>(Ws + tid * 4) = *reinterpret_cast<const f32x4*>(p.in_w + tid * 4);
    if (tid < 48) Ws[768 + tid] = p.in_b[tid];
    {   // dO, lse, delta of every query of this (view, b)
        const f32x4* s1 = reinterpret_cast<const f32x4*>(a.dA + (size_t)vb * Q * TC);
        for (int i = tid; i < Q * TC / 4; i += 256) reinterpret_cast<f32x4*>(dAs)[i] = s1[i];
        const f32x4* s2 = reinterpret_cast<const f32x4*>(a.lse + (size_t)vb * Q * TH);
        const f32x4* s3 = reinterpret_cast<const f32x4*>(a.delta + (size_t)vb * Q * TH);
        for (int i = tid; i < Q * TH / 4; i += 256) {
            reinterpret_cast<f32x4*>(Ls)[i] = s2[i];
            reinterpret_cast<f32x4*>(Ds)[i] = s3[i];
        }
    }
    __syncthreads();
    for (int i = tid; i < Q + 2 * KT; i += 256) {
        const int isq = i < Q;
        const int j = i - Q;
        const int k = isq ? i : min(k0 + (j >> 1), Q - 1);
        const int isv = isq ? 0 : (j & 1);
        float x[TC];
        load_row(xb + (size_t)k * TC, isv ? nullptr : a.pos + (size_t)k * TC, x);
        if (!isq) {
            float* xd = (isv ? Xv : Xk) + (j >> 1) * TC;
#pragma unroll
            for (int c = 0; c < TC; ++c) xd[c] = x[c];
        }
        float* dst = isq ? Qs + i * TC : (isv ? Vt : Kt) + (j >> 1) * TC;
        in_proj_rows(Ws, isq ? 0 : 16 + 16 * isv, x, isq ? kQScale : 1.f, dst);
    }
    __syncthreads();
    const DropCtx dc = drop_ctx(a.seed, a.salt, 0, a.p_drop);
    const int KP = ((Q + 15) >> 4) * 8;
    {   // lane = (head, query slice); KW keys of this wave in registers
        const int slice = lane & 7, h = lane >> 3;
        f32x2 kk[KW], vv[KW];
        float dk0[KW], dk1[KW], dv0[KW], dv1[KW];
        uint32_t koff[KW];
        int kwhich[KW];
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const int kl = wv * KW + i, k = min(k0 + kl, Q - 1);
            kk[i] = *reinterpret_cast<const f32x2*>(Kt + kl * TC + h * 2);
            vv[i] = *reinterpret_cast<const f32x2*>(Vt + kl * TC + h * 2);
            dk0[i] = dk1[i] = dv0[i] = dv1[i] = 0.f;
            koff[i] = (uint32_t)((k >> 4) * 8 + (k & 7));
            kwhich[i] = (k >> 3) & 1;
        }
#pragma unroll 1
        for (int q = slice; q < Q; q += 8) {
            const f32x2 qh = *reinterpret_cast<const f32x2*>(Qs + q * TC + h * 2);
            const f32x2 dO = *reinterpret_cast<const f32x2*>(dAs + q * TC + h * 2);
            const float ls = Ls[q * TH + h], de = Ds[q * TH + h];
            const uint32_t hb = attn_pair_index(vb, h, q, Q, KP, 0);
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const float pr = __expf(qh[0] * kk[i][0] + qh[1] * kk[i][1] - ls);
                const uint32_t hs = drop_hash(hb + koff[i], dc.s0, dc.s1);
                const float keep = drop_keep(hs, kwhich[i], dc.thr) ? dc.inv : 0.f;
                const float dp = (dO[0] * vv[i][0] + dO[1] * vv[i][1]) * keep;
                const float ds = pr * (dp - de);
                const float pd = pr * keep;
                dk0[i] = fmaf(ds, qh[0], dk0[i]);
                dk1[i] = fmaf(ds, qh[1], dk1[i]);
                dv0[i] = fmaf(pd, dO[0], dv0[i]);
                dv1[i] = fmaf(pd, dO[1], dv1[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < KW; ++i) {
            const float s0 = g8_sum(dk0[i]), s1 = g8_sum(dk1[i]), t0 = g8_sum(dv0[i]), t1 = g8_sum(dv1[i]);
            if (slice == 0) {
                const int kl = wv * KW + i;
                const bool ok = k0 + kl < Q;
                dKs[kl * TC + h * 2 + 0] = ok ? s0 : 0.f;
                dKs[kl * TC + h * 2 + 1] = ok ? s1 : 0.f;
                dVs[kl * TC + h * 2 + 0] = ok ? t0 : 0.f;
                dVs[kl * TC + h * 2 + 1] = ok ? t1 : 0.f;
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < KT * TC; t += 256) {      // in_proj backward of the key / value rows, added to bwd_q's result
        const int kl = t >> 4, c = t & 15;
        const int k = k0 + kl;
        float sk = 0.f, sv = 0.f;
#pragma unroll 4
        for (int o = 0; o < TC; ++o) {
            sk = fmaf(dKs[kl * TC + o], Ws[(16 + o) * 16 + c], sk);
            sv = fmaf(dVs[kl * TC + o], Ws[(32 + o) * 16 + c], sv);
        }
        if (k < Q) {
            const size_t row = (size_t)vb * Q + k;
            a.dxp[row * TC + c] += sk;
            a.dx[row * TC + c] += sk + sv;
        }
    }
    {
        const int o = tid >> 4, c = tid & 15;
        float s1 = 0.f, s2 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll 1
        for (int kl = 0; kl < KT; ++kl) {
            s1 = fmaf(dKs[kl * TC + o], Xk[kl * TC + c], s1);
            s2 = fmaf(dVs[kl * TC + o], Xv[kl * TC + c], s2);
            b1 += dKs[kl * TC + o];
            b2 += dVs[kl * TC + o];
        }
        atomicAdd(g.in_w + (16 + o) * TC + c, s1);
        atomicAdd(g.in_w + (32 + o) * TC + c, s2);
        if (c == 0) {
            atomicAdd(g.in_b + 16 + o, b1);
            atomicAdd(g.in_b + 32 + o, b2);
        }
    }
}

static int pick_qw(int B, int Q, int V, int which = 0) {      // which: 0 forward, 1 backward (queries), 2 backward (keys / values)
    static const int env[3] = {getenv("DPFT_SA_QW_FWD") ? atoi(getenv("DPFT_SA_QW_FWD")) : 0,
                               getenv("DPFT_SA_QW_BWD") ? atoi(getenv("DPFT_SA_QW_BWD")) : 0,
                               getenv("DPFT_SA_KW") ? atoi(getenv("DPFT_SA_KW")) : 0};      // tuning aid
    if (env[which] >= 1 && env[which] <= 8 && env[which] != 7) return env[which];
    const int tiles = std::max(1, kNumCU / (V * B));
    int qw = std::min(8, std::max(1, cdiv(cdiv(Q, tiles), 4)));
    if (qw == 7) qw = 8;
    return qw;
}

#define SA_DISPATCH(KERNEL, qw, grid, lds, stream, args)                                                    \
    do {                                                                                                    \
        void (*kfn)(SaArgs) = nullptr;                                                                      \
        switch (qw) {                                                                                       \
            case 1: kfn = KERNEL<1>; break;                                                                 \
            case 2: kfn = KERNEL<2>; break;                                                                 \
            case 3: kfn = KERNEL<3>; break;                                                                 \
            case 4: kfn = KERNEL<4>; break;                                                                 \
            case 5: kfn = KERNEL<5>; break;                                                                 \
            case 6: kfn = KERNEL<6>; break;                                                                 \
            default: kfn = KERNEL<8>; break;                                                                \
        }                                                                                                   \
        if ((lds) > 64 * 1024)                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                          \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, (hipStream_t)stream, args);                           \
    } while (0)

static int fill_common(SaArgs& a, const dpft_sa_params* params, int V, const float* x, int64_t xbs, const float* pos,
                       float p_drop, const int64_t* seed, int salt, int B, int Q) {
    DPFT_REQUIRE(params && x && pos && seed, "selfattn_train: null argument");
    DPFT_REQUIRE(V >= 1 && V <= 4 && B > 0 && Q > 0, "selfattn_train: bad sizes (V=%d, B=%d, Q=%d)", V, B, Q);
    DPFT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "selfattn_train: dropout probability must be in [0,1)");
    DPFT_REQUIRE((int64_t)V * B * TH * Q * (((Q + 15) >> 4) * 8) < (1ll << 32), "selfattn_train: problem too large for the mask index");
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < V; ++v) {
        a.p[v] = params[v];
        const float* const* f = reinterpret_cast<const float* const*>(params + v);
        for (size_t i = 0; i < sizeof(dpft_sa_params) / sizeof(float*); ++i)
            DPFT_REQUIRE(f[i], "selfattn_train: parameter %d of view %d is null", (int)i, v);
    }
    a.x = x; a.xbs = (long)xbs; a.pos = pos; a.seed = seed; a.salt = salt; a.p_drop = p_drop;
    a.B = B; a.Q = Q; a.V = V;
    return DPFT_OK;
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_selfattn_train_fwd_f32(const dpft_sa_params* params, int32_t V, const float* x, int64_t x_bstride,
                                           const float* pos, float p_drop, const int64_t* seed, int32_t salt,
                                           float* y1, float* lse, float* attn, float* zhat, float* rstd, int32_t B,
                                           int32_t Q, dpft_stream_t stream) {
    SaArgs a;
    int rc = fill_common(a, params, V, x, x_bstride, pos, p_drop, seed, salt, B, Q);
    if (rc) return rc;
    DPFT_REQUIRE(y1 && lse && attn && zhat && rstd, "selfattn_train_fwd: null output");
    a.y1 = y1; a.lse = lse; a.attn = attn; a.zhat = zhat; a.rstd = rstd;
    const int qw = pick_qw(B, Q, V), qt = 4 * qw;
    const size_t lds = ((size_t)Q * 32 + 48 * 16 + 48 + qt * TC + qt * TH * 8 * 4) * sizeof(float);
    DPFT_REQUIRE(lds <= 160 * 1024, "selfattn_train_fwd: %d queries do not fit the LDS", Q);
    SA_DISPATCH(sa_train_fwd_kernel, qw, dim3(cdiv(Q, qt), V, B), lds, stream, a);
    return check_launch("selfattn_train_fwd");
}

extern "C" int dpft_selfattn_train_bwd_f32(const dpft_sa_params* params, int32_t V, const float* x, int64_t x_bstride,
                                           const float* pos, float p_drop, const int64_t* seed, int32_t salt,
                                           const float* dy1, const float* lse, const float* attn, const float* zhat,
                                           const float* rstd, const dpft_sa_grads* grads, float* dx, float* dxp,
                                           float* scratch, int32_t B, int32_t Q, dpft_stream_t stream) {
    SaArgs a;
    int rc = fill_common(a, params, V, x, x_bstride, pos, p_drop, seed, salt, B, Q);
    if (rc) return rc;
    DPFT_REQUIRE(dy1 && lse && attn && zhat && rstd && grads && dx && dxp && scratch, "selfattn_train_bwd: null argument");
    for (int v = 0; v < V; ++v) {
        a.g[v] = grads[v];
        float* const* f = reinterpret_cast<float* const*>(grads + v);
        for (size_t i = 0; i < sizeof(dpft_sa_grads) / sizeof(float*); ++i)
            DPFT_REQUIRE(f[i], "selfattn_train_bwd: gradient buffer %d of view %d is null", (int)i, v);
    }
    a.dy1 = dy1; a.lse = const_cast<float*>(lse); a.attn = const_cast<float*>(attn); a.zhat = const_cast<float*>(zhat);
    a.rstd = const_cast<float*>(rstd); a.dx = dx; a.dxp = dxp;
    a.dA = scratch; a.delta = scratch + (size_t)V * B * Q * TC;
    int qw = pick_qw(B, Q, V, 1), qt = 4 * qw;
    const size_t lds_q = ((size_t)Q * 32 + 48 * 16 + 48 + 6 * qt * TC + qt * TH + std::max(2 * qt * TC, 0)) * sizeof(float);
    DPFT_REQUIRE(lds_q <= 160 * 1024, "selfattn_train_bwd: %d queries do not fit the LDS", Q);
    SA_DISPATCH(sa_train_bwd_q_kernel, qw, dim3(cdiv(Q, qt), V, B), lds_q, stream, a);
    rc = check_launch("selfattn_train_bwd_q");
    if (rc) return rc;
    qw = pick_qw(B, Q, V, 2); qt = 4 * qw;
    const size_t lds_kv = ((size_t)Q * 48 + 48 * 16 + 48 + 6 * qt * TC) * sizeof(float);
    DPFT_REQUIRE(lds_kv <= 160 * 1024, "selfattn_train_bwd: %d queries do not fit the LDS", Q);
    SA_DISPATCH(sa_train_bwd_kv_kernel, qw, dim3(cdiv(Q, qt), V, B), lds_kv, stream, a);
    return check_launch("selfattn_train_bwd_kv");
}

extern "C" int64_t dpft_selfattn_train_scratch_floats(int32_t B, int32_t Q, int32_t V) {
    return (int64_t)V * B * Q * (TC + TH);
}
