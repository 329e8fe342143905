// Fused inference path of the iterative fusion decoder (eval mode, no autograd) for d_model = 16, 8 heads
// (head_dim 2), 5 levels x 4 points, Mish FFN, LayerNorm, 'linear' view reduction, 3-layer linear heads --
// i.e. IMPFusion / MPFusion / MLFusion / LinearDetectionHead of the reference at config/kradar*.json:
//   src/dprt/models/fusers/mpfusion.py:122-148 (self attention), :150-208 + layers/ms_deform_attn.py:138-217
//   (deformable cross attention), :210-229 (FFN), :416-470,:472-514 (view reduction), :617-696 (reference
//   points), :698-745 (iteration), src/dprt/models/heads/detection.py:252-275 (head).
// The eager decoder is ~700 launches of B*400x16-sized ops per forward (launch-bound: ~5 ms even when
// replayed from a hipGraph); here one iteration is 2 kernels:
//   K1 decoder_selfattn_kernel   : all views; K/V of the 400 keys in LDS, lane = (head, key slice), 4 queries per
//                                  wave in registers, online softmax; out_proj + residual + LayerNorm1 epilogue
//   K2 decoder_xattn_head_kernel : block = the V waves of one (b, query).  Per wave: reference point from the
//                                  previous center, offsets/logits GEMV + softmax, sample-then-project gather on
//                                  the NHWC pyramid, output_proj + LN2, FFN (Mish) + LN3.  Then wave 0: 48->16
//                                  view reduction and the 4 head MLPs (16 lanes per branch), center += previous.
// All small matrices are read from PACKED blobs (dpft_decoder_pack_*): transposed so that the 64 lanes of a
// wave read consecutive floats (a torch (out,in) row per lane is a 64-cache-line gather per instruction and
// made the first version of K2 texture-addresser bound).
#include "common.h"
#include "decoder_pack.h"

#define RC(call)              \
    do {                      \
        int rc_ = (call);     \
        if (rc_) return rc_;  \
    } while (0)

namespace dpft {

__global__ void pack_view_kernel(dpft_decoder_view s, int n_off, int n_att, float* __restrict__ d) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PV_FLOATS; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        int r;
        if (i < PV_IN_B) v = s.in_proj_w[i];
        else if (i < PV_OUT_WT) v = s.in_proj_b[i - PV_IN_B];
        else if (i < PV_OUT_B) { r = i - PV_OUT_WT; v = s.out_proj_w[(r & 15) * 16 + (r >> 4)]; }
        else if (i < PV_N1_W) v = s.out_proj_b[i - PV_OUT_B];
        else if (i < PV_N1_B) v = s.norm1_w[i - PV_N1_W];
        else if (i < PV_OA_WT) v = s.norm1_b[i - PV_N1_B];
        else if (i < PV_OA_B) {
            r = i - PV_OA_WT;
            const int c = r / NOA, o = r - c * NOA;
            v = o < n_off ? s.off_w[o * 16 + c] : (o < n_off + n_att ? s.att_w[(o - n_off) * 16 + c] : 0.f);
        } else if (i < PV_VAL_W) {
            const int o = i - PV_OA_B;
            v = o < n_off ? s.off_b[o] : (o < n_off + n_att ? s.att_b[o - n_off] : 0.f);
        } else if (i < PV_VAL_B) v = s.val_w[i - PV_VAL_W];
        else if (i < PV_OUTP_WT) v = s.val_b[i - PV_VAL_B];
        else if (i < PV_OUTP_B) { r = i - PV_OUTP_WT; v = s.outp_w[(r & 15) * 16 + (r >> 4)]; }
        else if (i < PV_N2_W) v = s.outp_b[i - PV_OUTP_B];
        else if (i < PV_N2_B) v = s.norm2_w[i - PV_N2_W];
        else if (i < PV_F1_WT) v = s.norm2_b[i - PV_N2_B];
        else if (i < PV_F1_B) { r = i - PV_F1_WT; v = s.ffn1_w[(r & 31) * 16 + (r >> 5)]; }
        else if (i < PV_F2_WT) v = s.ffn1_b[i - PV_F1_B];
        else if (i < PV_F2_B) { r = i - PV_F2_WT; v = s.ffn2_w[(r & 15) * 32 + (r >> 4)]; }
        else if (i < PV_N3_W) v = s.ffn2_b[i - PV_F2_B];
        else if (i < PV_N3_B) v = s.norm3_w[i - PV_N3_W];
        else v = s.norm3_b[i - PV_N3_B];
        d[i] = v;
    }
}

struct HeadSrc {
    const float* red_w;
    const float* hw[4][3];
    int nout[4];
    int V;
};
__global__ void pack_head_kernel(HeadSrc s, float* __restrict__ d) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PH_FLOATS; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < PH_W) {
            const int vw = i >> 8, k = (i >> 4) & 15, o = i & 15;
            if (vw < s.V) v = s.red_w[o * DC * s.V + k * s.V + vw];
        } else {
            const int r = i - PH_W;
            const int layer = r >> 10, k = (r >> 6) & 15, g = (r >> 4) & 3, o = r & 15;
            const int rows = layer == 2 ? s.nout[g] : DC;
            if (o < rows) v = s.hw[g][layer][o * DC + k];
        }
        d[i] = v;
    }
}

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// LayerNorm over 16 channels held by 16 consecutive lanes (torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ float layernorm16(float v, float g, float b) {
    const float mean = group16_sum(v) * (1.f / 16.f);
    const float d = v - mean;
    const float var = group16_sum(d * d) * (1.f / 16.f);
    return d * (1.0f / sqrtf(var + 1e-5f)) * g + b;
}

struct SelfAttnArgs {
    const float* pv[4];   // packed view blobs
    const float* query;   // (B,Q,16), or (Q,16) broadcast over the batch when qstride == 0
    const float* pos;     // (Q,16)
    float* y1;            // (V,B,Q,16)
    int B, Q, V;
    long qstride;
};

// block = QT = 4*QW queries of one (b, view): 4 waves x QW queries (register-blocked: every K/V read from LDS is
// used for QW queries), lane = (head, key slice).  Every block recomputes K/V of all keys (~1/3 of its work), so
// the host picks QW such that the grid is ONE round of <= 256 blocks (one per CU): with 300 blocks of 16 queries
// 44 CUs ran two blocks back to back and the kernel took twice as long.
template <int QW>
__global__ __launch_bounds__(256) void decoder_selfattn_kernel(SelfAttnArgs a) {
    constexpr int QT = 4 * QW;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q;
    float* Ks = sm;                   // [Q][16]
    float* Vs = sm + Q * DC;          // [Q][16]
    float* Ws = Vs + Q * DC;          // in_proj rows [48][16] + bias [48]
    float* Qs = Ws + 48 * 16 + 48;    // [QT][16] projected, scaled queries; reused as the attention output tile
    float* Pt = Qs + QT * DC;         // [QT][8 heads][8 slices][4] partial (max, den, o0, o1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * QT;
    const float* pv = a.pv[view];
    const float* xb = a.query + (size_t)b * a.qstride;
    if (tid < (768 + 48) / 4) *reinterpret_cast<f32x4*>(Ws + tid * 4) = *reinterpret_cast<const f32x4*>(pv + PV_IN_W + tid * 4);
    __syncthreads();
    // rows of [Q | K | V] = in_proj(x + pos | x + pos | x): item = (key, K|V) plus (query of the tile, Q)
    for (int i = tid; i < 2 * Q + QT; i += 256) {
        const int isq = i >= 2 * Q;
        const int k = isq ? min(q0 + i - 2 * Q, Q - 1) : (i >> 1);
        const int isv = isq ? 0 : (i & 1);
        const int row0 = isq ? 0 : 16 + 16 * isv;
        float x[DC];
#pragma unroll
        for (int c = 0; c < DC; c += 4) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (size_t)k * DC + c);
            if (!isv) xv += *reinterpret_cast<const f32x4*>(a.pos + (size_t)k * DC + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[c + e] = xv[e];
        }
        float* dst = isq ? Qs + (i - 2 * Q) * DC : (isv ? Vs : Ks) + k * DC;
        const float scale = isq ? 0.70710678118654752f : 1.f;      // 1/sqrt(head_dim) folded into Q
#pragma unroll 4
        for (int o = 0; o < DC; ++o) {
            const float* wr = Ws + (row0 + o) * 16;
            float sacc = Ws[48 * 16 + row0 + o];
#pragma unroll
            for (int c = 0; c < DC; c += 4) {
                const f32x4 wv4 = *reinterpret_cast<const f32x4*>(wr + c);
                sacc = fmaf(wv4[0], x[c], sacc); sacc = fmaf(wv4[1], x[c + 1], sacc);
                sacc = fmaf(wv4[2], x[c + 2], sacc); sacc = fmaf(wv4[3], x[c + 3], sacc);
            }
            dst[o] = sacc * scale;
        }
    }
    __syncthreads();
    {
        const int slice = lane & 7, h = lane >> 3;
        f32x2 qh[QW];
        float mx[QW], den[QW], o0[QW], o1[QW];
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            qh[i] = *reinterpret_cast<const f32x2*>(Qs + (wv * QW + i) * DC + h * 2);
            mx[i] = -INFINITY; den[i] = 0.f; o0[i] = 0.f; o1[i] = 0.f;
        }
        const float* kp = Ks + h * 2;
        const float* vp = Vs + h * 2;
#pragma unroll 1
        for (int k = slice; k < Q; k += 16) {      // 2 keys of this slice per step, each used for QW queries
            const int k2 = k + 8;
            const bool has2 = k2 < Q;
            const f32x2 ka = *reinterpret_cast<const f32x2*>(kp + k * DC), va = *reinterpret_cast<const f32x2*>(vp + k * DC);
            const f32x2 kb = *reinterpret_cast<const f32x2*>(kp + (has2 ? k2 : k) * DC);
            const f32x2 vb = *reinterpret_cast<const f32x2*>(vp + (has2 ? k2 : k) * DC);
#pragma unroll
            for (int i = 0; i < QW; ++i) {
                const float sa = qh[i][0] * ka[0] + qh[i][1] * ka[1];
                const float sb = has2 ? qh[i][0] * kb[0] + qh[i][1] * kb[1] : -INFINITY;
                const float m_new = fmaxf(fmaxf(sa, sb), mx[i]);
                const float corr = __expf(mx[i] - m_new);
                const float pa = __expf(sa - m_new), pb = __expf(sb - m_new);
                den[i] = den[i] * corr + pa + pb;
                o0[i] = fmaf(pb, vb[0], fmaf(pa, va[0], o0[i] * corr));
                o1[i] = fmaf(pb, vb[1], fmaf(pa, va[1], o1[i] * corr));
                mx[i] = m_new;
            }
        }
#pragma unroll
        for (int i = 0; i < QW; ++i) {
            const f32x4 part = {mx[i], den[i], o0[i], o1[i]};
            *reinterpret_cast<f32x4*>(Pt + (((wv * QW + i) * DM + h) * 8 + slice) * 4) = part;
        }
    }
    __syncthreads();
    // merge the 8 key slices of each (query, head); a slice may be empty (max = -inf, den = 0)
    if (tid < QT * DM) {
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
        Qs[tid * 2 + 0] = a0 / dsum;      // tid = query * 8 + head  ->  channel head*2 + d
        Qs[tid * 2 + 1] = a1 / dsum;
    }
    __syncthreads();
    // out_proj + residual + LayerNorm1: 16 lanes per query
    for (int t = tid; t < QT * DC; t += 256) {      // QT * 16 is a multiple of 64: whole waves stay active
        const int ql2 = t >> 4, c = t & 15;
        const int q2 = q0 + ql2;
        float v = 0.f;
        if (q2 < Q) {
            v = pv[PV_OUT_B + c];
#pragma unroll 4
            for (int j = 0; j < DC; ++j) v = fmaf(pv[PV_OUT_WT + j * DC + c], Qs[ql2 * DC + j], v);
            v += xb[(size_t)q2 * DC + c];
        }
        v = layernorm16(v, pv[PV_N1_W + c], pv[PV_N1_B + c]);
        if (q2 < Q) a.y1[(((size_t)view * a.B + b) * Q + q2) * DC + c] = v;
    }
}

struct Pyr5 {
    const float* level[DPFT_MAX_LEVELS];
    int H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int L;
};
struct XattnHeadArgs {
    Pyr5 pyr[4];
    const float* pv[4];         // packed view blobs
    const float* ph;            // packed head blob
    const float* y1;            // (V,B,Q,16)
    const float* pos;           // (Q,16)
    const float* prev_center;   // (B,Q,3): reference points are projected from it; center = head + prev_center
    const float* T[4];          // (B,4,4)
    const float* Pm[4];         // (B,prow,4)
    const int64_t* shape[4];    // (B,2) = H, W
    int prow[4], flag[4], P[4];
    float* query_out;           // (B,Q,16)
    float *center, *size, *angle, *cls;
    int B, Q, V, ncls;
};

__device__ __forceinline__ float group8_sum_d(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}


// reference point of one view: cartesian center -> (optional T + spherical) -> projection P -> normalised, clamped
// (mpfusion.py:617-696)
__device__ __forceinline__ void reference_point(float cx, float cy, float cz, int flag, const float* T, const float* Pm,
                                                float Hs, float Ws, float& u, float& vv) {
    const float RAD2DEG = 57.29577951308232f;
    float p0 = cx, p1 = cy, p2 = cz;
    if (flag) {
        const float tx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];
        const float ty = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
        const float tz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
        const float r = sqrtf(tx * tx + ty * ty + tz * tz);
        p0 = r;
        p1 = atan2f(ty, tx) * RAD2DEG;
        p2 = asinf(r != 0.f ? tz / r : 0.f) * RAD2DEG;
    }
    u = Pm[0] * p0 + Pm[1] * p1 + Pm[2] * p2 + Pm[3];
    vv = Pm[4] * p0 + Pm[5] * p1 + Pm[6] * p2 + Pm[7];
    const float wq = Pm[8] * p0 + Pm[9] * p1 + Pm[10] * p2 + Pm[11];
    if (wq != 0.f) { u /= wq; vv /= wq; }
    u = fminf(fmaxf(u / Ws, 0.f), 1.f);
    vv = fminf(fmaxf(vv / Hs, 0.f), 1.f);
}

// block = the V waves of one (b, q); per-wave LDS scratch: qp[16] | lin[480] | vec[32]
__global__ __launch_bounds__(256) void decoder_xattn_head_kernel(XattnHeadArgs a) {
    __shared__ float sm[4][16 + NOA + 32];
    __shared__ float y3s[4][DC];
    __shared__ float hx[DC], hh1[4][DC], hh2[4][DC];
    const int lane = threadIdx.x & 63;
    const int view = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x;
    float* qp = sm[view];
    float* lin = qp + 16;     // [0,n_off): offsets (m,l,p,xy) ; [n_off, n_off+n_att): attention logits (m, l*P+p)
    float* vec = lin + NOA;
    const int b = bq / a.Q, q = bq - b * a.Q;
    const float* pv = a.pv[view];
    const Pyr5& pyr = a.pyr[view];
    const int L = pyr.L, P = a.P[view], LP = L * P;
    const size_t vbq = (size_t)view * a.B * a.Q + bq;
    const float y1c = lane < 16 ? a.y1[vbq * DC + lane] : 0.f;
    if (lane < 16) qp[lane] = y1c + a.pos[(size_t)q * DC + lane];
    __builtin_amdgcn_wave_barrier();
    // GEMV: n_off offsets + n_att logits from the 16-vector qp; lanes read consecutive columns of W^T
    const int n_off = DM * LP * 2, n_all = DM * LP * 3;
    {
        float x[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) x[c] = qp[c];
        for (int o = lane; o < n_all; o += 64) {
            float s = pv[PV_OA_B + o];
#pragma unroll
            for (int c = 0; c < DC; ++c) s = fmaf(pv[PV_OA_WT + c * NOA + o], x[c], s);
            lin[o] = s;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int m = lane >> 3, j = lane & 7;
    // softmax over the L*P logits of head m (each of the 8 lanes of the head computes it redundantly)
    const float* lg = lin + n_off + m * LP;
    float mx = -INFINITY;
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
    for (int i = 0; i < LP; ++i) den += __expf(lg[i] - mx);
    const float inv_den = 1.f / den;
    float rx, ry;
    {
        const float* pc = a.prev_center + (size_t)bq * 3;
        reference_point(pc[0], pc[1], pc[2], a.flag[view], a.T[view] ? a.T[view] + (size_t)b * 16 : nullptr,
                        a.Pm[view] + (size_t)b * a.prow[view] * 4, (float)a.shape[view][b * 2 + 0],
                        (float)a.shape[view][b * 2 + 1], rx, ry);
    }
    const float* offp = lin + m * LP * 2;
    f32x2 acc = {0.f, 0.f};
    float ms = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const float* base = pyr.level[l] + (int64_t)b * H * W * DC + j * 2;
        // all (<= 4) points of the level: addresses are clamped and the 16 gathers are issued back to back
        // (independent loads in flight); invalid corners / out-of-range samples get weight 0
        f32x2 v[4][4];
        float wgt[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool act = p < P;
            const int lp = act ? l * P + p : l * P;
            const float ox = offp[lp * 2 + 0], oy = offp[lp * 2 + 1];
            const float aw = act ? __expf(lg[lp] - mx) * inv_den : 0.f;
            const float lx = rx + ox / (float)W, ly = ry + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = act && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_lo = (int)hf, w_lo = (int)wf, h_hi = h_lo + 1, w_hi = w_lo + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
            const bool k1 = in && h_lo >= 0 && w_lo >= 0, k2 = in && h_lo >= 0 && w_hi <= W - 1;
            const bool k3 = in && h_hi <= H - 1 && w_lo >= 0, k4 = in && h_hi <= H - 1 && w_hi <= W - 1;
            const int hl = min(max(h_lo, 0), H - 1), hh_ = min(max(h_hi, 0), H - 1);
            const int wl = min(max(w_lo, 0), W - 1), wh_ = min(max(w_hi, 0), W - 1);
            v[p][0] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hl * W + wl) * DC);
            v[p][1] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hl * W + wh_) * DC);
            v[p][2] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hh_ * W + wl) * DC);
            v[p][3] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hh_ * W + wh_) * DC);
            wgt[p][0] = k1 ? aw * hh * hw : 0.f;
            wgt[p][1] = k2 ? aw * hh * lw : 0.f;
            wgt[p][2] = k3 ? aw * lh * hw : 0.f;
            wgt[p][3] = k4 ? aw * lh * lw : 0.f;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc += wgt[p][k] * v[p][k];
                ms += wgt[p][k];
            }
    }
    // value_proj on the sampled features (+ bias * in-bounds mass), head m -> channels 2m, 2m+1
    const f32x2 wv0 = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 0) * DC + j * 2);
    const f32x2 wv1 = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 1) * DC + j * 2);
    const float o0 = group8_sum_d(wv0[0] * acc[0] + wv0[1] * acc[1]);
    const float o1 = group8_sum_d(wv1[0] * acc[0] + wv1[1] * acc[1]);
    if (j == 0) {
        vec[m * 2 + 0] = o0 + pv[PV_VAL_B + m * 2 + 0] * ms;
        vec[m * 2 + 1] = o1 + pv[PV_VAL_B + m * 2 + 1] * ms;
    }
    __builtin_amdgcn_wave_barrier();
    // output_proj + residual + LayerNorm2 (lanes 0..15 = channels; other lanes mirror them)
    const int c = lane & 15;
    float vo = pv[PV_OUTP_B + c];
#pragma unroll
    for (int k = 0; k < DC; ++k) vo = fmaf(pv[PV_OUTP_WT + k * DC + c], vec[k], vo);
    vo += __shfl(y1c, c);
    const float y2 = layernorm16(vo, pv[PV_N2_W + c], pv[PV_N2_B + c]);
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) qp[lane] = y2;            // reuse qp for y2
    __builtin_amdgcn_wave_barrier();
    // FFN: 16 -> 32 (Mish) -> 16, residual, LayerNorm3
    {
        const int jf = lane & 31;
        float hsum = pv[PV_F1_B + jf];
#pragma unroll
        for (int k = 0; k < DC; ++k) hsum = fmaf(pv[PV_F1_WT + k * DFF + jf], qp[k], hsum);
        if (lane < DFF) vec[lane] = mishf(hsum);
    }
    __builtin_amdgcn_wave_barrier();
    float f = pv[PV_F2_B + c];
#pragma unroll
    for (int k = 0; k < DFF; ++k) f = fmaf(pv[PV_F2_WT + k * DC + c], vec[k], f);
    f += y2;
    const float y3 = layernorm16(f, pv[PV_N3_W + c], pv[PV_N3_B + c]);
    if (lane < 16) y3s[view][lane] = y3;
    __syncthreads();
    if (view != 0) return;
    // view reduction: queries.view(B,N,C*V) is channel-major / view-minor (mpfusion.py:436-438)
    const float* ph = a.ph;
    {
        float x = 0.f;
        for (int v2 = 0; v2 < a.V; ++v2)
#pragma unroll
            for (int k = 0; k < DC; ++k) x = fmaf(ph[PH_RED_WT + (v2 * DC + k) * DC + c], y3s[v2][k], x);
        if (lane < 16) {
            hx[lane] = x;
            a.query_out[(size_t)bq * DC + lane] = x;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // heads (heads/detection.py:252-275): branch g = lane / 16 (center, size, angle, class), row o = lane % 16
    const int g = lane >> 4, o = lane & 15;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (0 * DC + k) * 64 + lane], hx[k], t);
    hh1[g][o] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (1 * DC + k) * 64 + lane], hh1[g][k], t);
    hh2[g][o] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (2 * DC + k) * 64 + lane], hh2[g][k], t);
    const int nout = g == 0 ? 3 : (g == 1 ? 3 : (g == 2 ? 2 : a.ncls));
    if (o < nout) {
        if (g == 0) a.center[bq * 3 + o] = t + a.prev_center[bq * 3 + o];
        else if (g == 1) a.size[bq * 3 + o] = fmaxf(t, 0.f);
        else if (g == 2) a.angle[bq * 2 + o] = tanhf(t);
        else a.cls[bq * a.ncls + o] = t;
    }
}

}  // namespace dpft

using namespace dpft;

extern "C" int64_t dpft_decoder_packed_view_floats(void) { return PV_FLOATS; }
extern "C" int64_t dpft_decoder_packed_head_floats(void) { return PH_FLOATS; }

extern "C" int dpft_decoder_pack_view_f32(const dpft_decoder_view* view, int32_t L, int32_t P, float* packed,
                                          dpft_stream_t stream) {
    DPFT_REQUIRE(view && packed, "decoder_pack_view: null argument");
    DPFT_REQUIRE(L >= 1 && L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && L * P * DM * 3 <= NOA,
                 "decoder_pack_view: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", L, P);
    const float* const* f = reinterpret_cast<const float* const*>(view);
    for (size_t i = 0; i < sizeof(dpft_decoder_view) / sizeof(float*); ++i)
        DPFT_REQUIRE(f[i], "decoder_pack_view: parameter pointer %d is null", (int)i);
    hipLaunchKernelGGL(pack_view_kernel, dim3(cdiv(PV_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, *view,
                       DM * L * P * 2, DM * L * P, packed);
    return check_launch("decoder_pack_view");
}

extern "C" int dpft_decoder_pack_head_f32(const float* red_w, const float* const* head_w, int32_t V, int32_t num_classes,
                                          float* packed, dpft_stream_t stream) {
    DPFT_REQUIRE(red_w && head_w && packed && V >= 1 && V <= 4, "decoder_pack_head: bad arguments");
    DPFT_REQUIRE(num_classes >= 1 && num_classes <= 16, "decoder_pack_head: num_classes must be in [1,16]");
    HeadSrc s;
    s.red_w = red_w; s.V = V;
    for (int g = 0; g < 4; ++g)
        for (int k = 0; k < 3; ++k) {
            DPFT_REQUIRE(head_w[g * 3 + k], "decoder_pack_head: head weight %d.%d is null", g, k);
            s.hw[g][k] = head_w[g * 3 + k];
        }
    s.nout[0] = 3; s.nout[1] = 3; s.nout[2] = 2; s.nout[3] = num_classes;
    hipLaunchKernelGGL(pack_head_kernel, dim3(cdiv(PH_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, s, packed);
    return check_launch("decoder_pack_head");
}

// Whole IMPFusion forward from ONE call: 2 launches per iteration, nothing else on the host
extern "C" int dpft_decoder_forward_f32(const dpft_decoder_fwd* d, dpft_stream_t stream) {
    DPFT_REQUIRE(d && d->packed_views && d->packed_heads && d->pyr && d->query0 && d->pos && d->center0 && d->work,
                 "decoder_forward: null argument");
    DPFT_REQUIRE(d->center && d->size && d->angle && d->cls, "decoder_forward: null output");
    const int B = d->B, Q = d->Q, V = d->V;
    DPFT_REQUIRE(B > 0 && Q > 0 && V >= 1 && V <= 4 && d->iters >= 1 && d->iters <= 8, "decoder_forward: bad sizes");
    DPFT_REQUIRE(d->num_classes >= 1 && d->num_classes <= 16, "decoder_forward: num_classes must be in [1,16]");
    const size_t nq = (size_t)B * Q;
    float* w = d->work;
    float* qbuf[2] = {w, w + nq * DC};
    float* y1 = w + 2 * nq * DC;
    float* cbuf[2] = {y1 + (size_t)V * nq * DC, y1 + (size_t)V * nq * DC + nq * 3};
    XattnHeadArgs xa;
    memset(&xa, 0, sizeof(xa));
    for (int v = 0; v < V; ++v) {
        const dpft_pyramid* pyr = d->pyr + v;
        const int P = d->n_points[v];
        DPFT_REQUIRE(pyr->L >= 1 && pyr->L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && pyr->L * P * DM * 3 <= NOA,
                     "decoder_forward: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", pyr->L, P);
        xa.pyr[v].L = pyr->L;
        for (int l = 0; l < pyr->L; ++l) {
            DPFT_REQUIRE(pyr->level[l], "decoder_forward: view %d level %d is null", v, l);
            xa.pyr[v].level[l] = pyr->level[l]; xa.pyr[v].H[l] = pyr->H[l]; xa.pyr[v].W[l] = pyr->W[l];
        }
        xa.P[v] = P;
        xa.T[v] = d->T[v]; xa.Pm[v] = d->P[v]; xa.shape[v] = d->shape[v]; xa.prow[v] = d->p_rows[v]; xa.flag[v] = d->has_t[v];
        DPFT_REQUIRE(xa.Pm[v] && xa.shape[v] && (xa.T[v] || !xa.flag[v]) && xa.prow[v] >= 3,
                     "decoder_forward: projection inputs of view %d missing", v);
    }
    xa.y1 = y1; xa.pos = d->pos; xa.B = B; xa.Q = Q; xa.V = V; xa.ncls = d->num_classes;
    xa.size = d->size; xa.angle = d->angle; xa.cls = d->cls;
    const float* query = d->query0;
    const float* center = d->center0;
    // queries per wave: one round of blocks over the 256 CUs
    const int tiles = std::max(1, kNumCU / (V * B));
    int qw = std::min(8, std::max(1, cdiv(cdiv(Q, tiles), 4)));
    if (qw == 7) qw = 8;
    const int qt = 4 * qw;
    const size_t lds = ((size_t)Q * 32 + 48 * 16 + 48 + qt * DC + qt * DM * 8 * 4) * sizeof(float);
    DPFT_REQUIRE(lds <= 160 * 1024, "decoder_forward: %d queries do not fit the LDS", Q);
    void (*sa_kernel)(SelfAttnArgs) = nullptr;
    switch (qw) {
        case 1: sa_kernel = decoder_selfattn_kernel<1>; break;
        case 2: sa_kernel = decoder_selfattn_kernel<2>; break;
        case 3: sa_kernel = decoder_selfattn_kernel<3>; break;
        case 4: sa_kernel = decoder_selfattn_kernel<4>; break;
        case 5: sa_kernel = decoder_selfattn_kernel<5>; break;
        case 6: sa_kernel = decoder_selfattn_kernel<6>; break;
        default: sa_kernel = decoder_selfattn_kernel<8>; break;
    }
    static bool configured[9] = {false};
    if (!configured[qw] && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        configured[qw] = true;
    }
    for (int it = 0; it < d->iters; ++it) {
        SelfAttnArgs sa;
        for (int v = 0; v < 4; ++v) sa.pv[v] = xa.pv[v] = v < V ? d->packed_views + (size_t)(it * V + v) * PV_FLOATS : nullptr;
        sa.query = query; sa.pos = d->pos; sa.y1 = y1; sa.B = B; sa.Q = Q; sa.V = V;
        sa.qstride = (it == 0) ? 0 : (long)Q * DC;
        hipLaunchKernelGGL(sa_kernel, dim3(cdiv(Q, qt), V, B), dim3(256), lds, (hipStream_t)stream, sa);
        RC(check_launch("decoder_selfattn"));
        const bool last = it == d->iters - 1;
        xa.ph = d->packed_heads + (size_t)it * PH_FLOATS;
        xa.prev_center = center;
        xa.query_out = qbuf[it & 1];
        xa.center = last ? d->center : cbuf[it & 1];
        hipLaunchKernelGGL(decoder_xattn_head_kernel, dim3((unsigned)nq), dim3(64 * V), 0, (hipStream_t)stream, xa);
        RC(check_launch("decoder_xattn_head"));
        query = qbuf[it & 1];
        center = xa.center;
    }
    return DPFT_OK;
}

extern "C" int64_t dpft_decoder_work_floats(int32_t B, int32_t Q, int32_t V) {
    const int64_t nq = (int64_t)B * Q;
    return 2 * nq * DC + (int64_t)V * nq * DC + 2 * nq * 3 + 64;
}
