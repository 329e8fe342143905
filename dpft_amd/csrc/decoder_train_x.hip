// Training path of the decoder's deformable cross-attention + FFN block, all views of one MPFusion layer per launch:
//   y2 = LayerNorm2(y1 + dropout2(output_proj(MSDeformAttn(y1 + pos, ref, pyramid))))
//   y3 = LayerNorm3(y2 + dropout4(ffn2(dropout3(Mish(ffn1(y2))))))
// = MLFusion.forward_cross_attn + forward_ffn, src/dprt/models/fusers/mpfusion.py:150-229 with
// src/dprt/models/layers/ms_deform_attn.py:138-217 (8 heads x head_dim 2, L*P <= 20 samples per head).
// One wave per (view, b, query), lane = (head m, channel pair j) for the gather and lane % 16 = channel elsewhere.
//
// The backward RECOMPUTES the forward of its row (offsets/logits GEMV, gather, projections: a few us of ALU) instead
// of saving ~1 KB of intermediates per row, then walks back through it.  Parameter gradients are outer-product sums
// over the B*Q rows: the kernel writes each row's factors into `rows` (V,B*Q,XR_FLOATS) and the host framework turns
// them into the weight gradients with 5 small batched GEMMs + 1 column sum (see train_fused.py) -- no atomics on
// parameters.  Feature-pyramid gradients are scattered with fp32 atomics like dpft_xattn_bwd_f32.
#include "common.h"
#include "decoder_pack.h"

namespace dpft {

// ---- row buffer columns (floats) ----
constexpr int XR_DLIN = 0;                    // [480] d offsets | d logits        -> d sampling_offsets/attention_weights
constexpr int XR_DF = XR_DLIN + NOA;          // [16]  d ffn2 output               -> ffn2.bias ; x hd -> ffn2.weight
constexpr int XR_DPRE = XR_DF + 16;           // [32]  d ffn1 pre-activation       -> ffn1.bias ; x y2 -> ffn1.weight
constexpr int XR_DOUT = XR_DPRE + 32;         // [16]  d output_proj output        -> output_proj.bias ; x vec -> .weight
constexpr int XR_G3 = XR_DOUT + 16;           // [16]  dy3 * zhat3                 -> norm3.weight
constexpr int XR_B3 = XR_G3 + 16;             // [16]  dy3                         -> norm3.bias
constexpr int XR_G2 = XR_B3 + 16;             // [16]  dy2 * zhat2                 -> norm2.weight
constexpr int XR_B2 = XR_G2 + 16;             // [16]  dy2                         -> norm2.bias
constexpr int XR_DBV = XR_B2 + 16;            // [16]  dvec * mass                 -> value_proj.bias
constexpr int XR_DVEC = XR_DBV + 16;          // [16]  d value_proj output         ; x samp -> value_proj.weight
constexpr int XR_SUMMED = XR_DVEC + 16;       // columns [0, XR_SUMMED) are meaningful as column sums
constexpr int XR_QP = XR_SUMMED;              // [16]  y1 + pos
constexpr int XR_HD = XR_QP + 16;             // [32]  dropout3(Mish(ffn1(y2)))
constexpr int XR_Y2 = XR_HD + 32;             // [16]
constexpr int XR_VEC = XR_Y2 + 16;            // [16]  value-projected sampled features
constexpr int XR_SAMP = XR_VEC + 16;          // [8][16] attention-weighted sampled raw features per head
constexpr int XR_FLOATS = XR_SAMP + 128;

struct Pyr5g {
    const float* level[DPFT_MAX_LEVELS];
    float* grad[DPFT_MAX_LEVELS];
    int H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int rep[DPFT_MAX_LEVELS];      // gradient replicas (>= 1)
    int L;
};
struct XfArgs {
    Pyr5g pyr[4];
    const float* pv[4];         // packed view blobs (decoder_pack.h)
    dpft_decoder_view raw[4];   // torch layouts (backward: transposed products)
    const float* y1;            // (V,B,Q,16)
    const float* pos;           // (Q,16)
    const float* ref;           // (V,B,Q,2)
    const int64_t* seed;
    float* y3;                  // (V,B,Q,16)
    const float* dy3;           // (V,B,Q,16)
    float* dy1;                 // (V,B,Q,16)
    float* dqp;                 // (V,B,Q,16) gradient w.r.t. (y1 + pos) through the offsets/logits GEMV
    float* dref;                // (V,B,Q,2)
    float* rows;                // (V,B*Q,XR_FLOATS)
    int B, Q, salt;
    int P[4];
    float p_drop;
};

__device__ __forceinline__ uint32_t xdrop_hash(uint32_t idx, uint32_t s0, uint32_t s1) {
    uint32_t x = idx ^ s0;
    x *= 0xcc9e2d51u; x = (x << 15) | (x >> 17); x *= 0x1b873593u;
    x ^= s1;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// keep-scale (1/(1-p) or 0) of element `e` (two decisions per hash) of dropout stream `stream_id`
__device__ __forceinline__ float xdrop_scale(const int64_t* seed, int salt, int stream_id, float p, uint32_t e) {
    const uint64_t s = (uint64_t)(*seed);
    const uint32_t s0 = (uint32_t)s ^ ((uint32_t)salt * 0x9E3779B9u);
    const uint32_t s1 = (uint32_t)(s >> 32) + (uint32_t)stream_id * 0x7F4A7C15u;
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    const uint32_t h = xdrop_hash(e >> 1, s0, s1);
    return ((h >> (16 * (e & 1))) & 0xFFFFu) >= thr ? 1.f / (1.f - p) : 0.f;
}

__device__ __forceinline__ float xg16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float xg8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// Everything the backward needs from the forward of one row, per lane
struct XfFwd {
    float y1c;             // y1[c], c = lane & 15
    float mx, inv_den;     // softmax statistics of head m = lane >> 3
    float rx, ry;
    f32x2 acc;             // attention-weighted sampled features, channels 2j, 2j+1 of head m (before value_proj)
    float ms;              // in-bounds attention mass of head m
    float vecc;            // value-projected features, channel c
    float k2, y2, zhat2, rstd2;
    float pre, k3, hd;     // FFN hidden unit lane & 31
    float k4, zhat3, rstd3, y3;
};

// forward of one row; leaves qp[16] | lin[n_all] | vec[32] in the wave's LDS scratch (vec = hd at the end)
__device__ __forceinline__ void xf_forward_row(const XfArgs& a, int view, int bq, int b, int q, int lane, float* qp,
                                               float* lin, float* vec, XfFwd& f) {
    const float* pv = a.pv[view];
    const Pyr5g& pyr = a.pyr[view];
    const int L = pyr.L, P = a.P[view], LP = L * P;
    const size_t vbq = (size_t)view * a.B * a.Q + bq;
    const int c = lane & 15;
    f.y1c = a.y1[vbq * DC + c];
    if (lane < 16) qp[lane] = f.y1c + a.pos[(size_t)q * DC + lane];
    __builtin_amdgcn_wave_barrier();
    const int n_off = DM * LP * 2, n_all = DM * LP * 3;
    {
        float x[DC];
#pragma unroll
        for (int k = 0; k < DC; ++k) x[k] = qp[k];
        for (int o = lane; o < n_all; o += 64) {
            float s = pv[PV_OA_B + o];
#pragma unroll
            for (int k = 0; k < DC; ++k) s = fmaf(pv[PV_OA_WT + k * NOA + o], x[k], s);
            lin[o] = s;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int m = lane >> 3, j = lane & 7;
    const float* lg = lin + n_off + m * LP;
    float mx = -INFINITY;
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
    float den = 0.f;
    for (int i = 0; i < LP; ++i) den += __expf(lg[i] - mx);
    f.mx = mx;
    f.inv_den = 1.f / den;
    f.rx = a.ref[vbq * 2 + 0];
    f.ry = a.ref[vbq * 2 + 1];
    const float* offp = lin + m * LP * 2;
    f32x2 acc = {0.f, 0.f};
    float ms = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const float* base = pyr.level[l] + (int64_t)b * H * W * DC + j * 2;
        f32x2 v[4][4];
        float wgt[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool act = p < P;
            const int lp = act ? l * P + p : l * P;
            const float ox = offp[lp * 2 + 0], oy = offp[lp * 2 + 1];
            const float aw = act ? __expf(lg[lp] - mx) * f.inv_den : 0.f;
            const float lx = f.rx + ox / (float)W, ly = f.ry + oy / (float)H;
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
    f.acc = acc;
    f.ms = ms;
    const f32x2 wv0 = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 0) * DC + j * 2);
    const f32x2 wv1 = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 1) * DC + j * 2);
    const float o0 = xg8_sum(wv0[0] * acc[0] + wv0[1] * acc[1]);
    const float o1 = xg8_sum(wv1[0] * acc[0] + wv1[1] * acc[1]);
    if (j == 0) {
        vec[m * 2 + 0] = o0 + pv[PV_VAL_B + m * 2 + 0] * ms;
        vec[m * 2 + 1] = o1 + pv[PV_VAL_B + m * 2 + 1] * ms;
    }
    __builtin_amdgcn_wave_barrier();
    f.vecc = vec[c];
    float vo = pv[PV_OUTP_B + c];
#pragma unroll
    for (int k = 0; k < DC; ++k) vo = fmaf(pv[PV_OUTP_WT + k * DC + c], vec[k], vo);
    f.k2 = xdrop_scale(a.seed, a.salt, 2, a.p_drop, (uint32_t)(vbq * DC + c));
    vo = f.y1c + vo * f.k2;
    {
        const float mean = xg16_sum(vo) * (1.f / 16.f);
        const float d = vo - mean;
        const float var = xg16_sum(d * d) * (1.f / 16.f);
        f.rstd2 = 1.0f / sqrtf(var + 1e-5f);
        f.zhat2 = d * f.rstd2;
        f.y2 = f.zhat2 * pv[PV_N2_W + c] + pv[PV_N2_B + c];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) qp[lane] = f.y2;            // qp now holds y2 (y1 + pos is re-derived where needed)
    __builtin_amdgcn_wave_barrier();
    {
        const int jf = lane & 31;
        float hsum = pv[PV_F1_B + jf];
#pragma unroll
        for (int k = 0; k < DC; ++k) hsum = fmaf(pv[PV_F1_WT + k * DFF + jf], qp[k], hsum);
        f.pre = hsum;
        f.k3 = xdrop_scale(a.seed, a.salt, 3, a.p_drop, (uint32_t)(vbq * DFF + jf));
        f.hd = mishf(hsum) * f.k3;
        if (lane < DFF) vec[lane] = f.hd;
    }
    __builtin_amdgcn_wave_barrier();
    float ff = pv[PV_F2_B + c];
#pragma unroll
    for (int k = 0; k < DFF; ++k) ff = fmaf(pv[PV_F2_WT + k * DC + c], vec[k], ff);
    f.k4 = xdrop_scale(a.seed, a.salt, 4, a.p_drop, (uint32_t)(vbq * DC + c));
    ff = f.y2 + ff * f.k4;
    {
        const float mean = xg16_sum(ff) * (1.f / 16.f);
        const float d = ff - mean;
        const float var = xg16_sum(d * d) * (1.f / 16.f);
        f.rstd3 = 1.0f / sqrtf(var + 1e-5f);
        f.zhat3 = d * f.rstd3;
        f.y3 = f.zhat3 * pv[PV_N3_W + c] + pv[PV_N3_B + c];
    }
}

// grid (ceil(B*Q / 4), V), 4 waves (= 4 rows of one view) per block
__global__ __launch_bounds__(256) void xf_train_fwd_kernel(XfArgs a) {
    __shared__ float sm[4][16 + NOA + 32];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wv, view = blockIdx.y;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q, q = bq - b * a.Q;
    float* qp = sm[wv];
    XfFwd f;
    xf_forward_row(a, view, bq, b, q, lane, qp, qp + 16, qp + 16 + NOA, f);
    if (lane < 16) a.y3[((size_t)view * a.B * a.Q + bq) * DC + lane] = f.y3;
}

__global__ __launch_bounds__(256) void xf_train_bwd_kernel(XfArgs a) {
    __shared__ float sm[4][16 + NOA + 32 + NOA + 32 + 32 + 128 + 32 + 32];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wv, view = blockIdx.y;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q, q = bq - b * a.Q;
    float* qp = sm[wv];
    float* lin = qp + 16;
    float* vec = lin + NOA;
    float* dlin = vec + 32;      // [NOA]
    float* va = dlin + NOA;      // [32] scratch vector A
    float* vb2 = va + 32;        // [32] scratch vector B
    float* ds16 = vb2 + 32;      // [8 heads][16 channels] d(sampled features): the scatter's payload
    float* sc_w = ds16 + 128;    // [8 heads][4 corners] attention x bilinear weight (0 = corner not written)
    int* sc_o = reinterpret_cast<int*>(sc_w + 32);      // [8 heads][4 corners] pixel offset (floats) inside the level
    XfFwd f;
    xf_forward_row(a, view, bq, b, q, lane, qp, lin, vec, f);      // qp = y2, vec = hd, lin = offsets | logits
    const float* pv = a.pv[view];
    const dpft_decoder_view& rw = a.raw[view];
    const Pyr5g& pyr = a.pyr[view];
    const int L = pyr.L, P = a.P[view], LP = L * P;
    const int n_off = DM * LP * 2, n_all = DM * LP * 3;
    const size_t vbq = (size_t)view * a.B * a.Q + bq;
    float* row = a.rows + vbq * XR_FLOATS;
    const int c = lane & 15, m = lane >> 3, j = lane & 7, jf = lane & 31;
    const bool w16 = lane < 16, w32 = lane < 32;
    // ---- LayerNorm3 ----
    const float dy3 = a.dy3[vbq * DC + c];
    float dzh = dy3 * pv[PV_N3_W + c];
    float m1 = xg16_sum(dzh) * (1.f / 16.f), m2 = xg16_sum(dzh * f.zhat3) * (1.f / 16.f);
    const float dz3 = f.rstd3 * (dzh - m1 - f.zhat3 * m2);
    const float df = dz3 * f.k4;
    if (w16) {
        row[XR_G3 + c] = dy3 * f.zhat3;
        row[XR_B3 + c] = dy3;
        row[XR_DF + c] = df;
        row[XR_Y2 + c] = f.y2;
        va[c] = df;
    }
    if (w32) row[XR_HD + jf] = f.hd;
    __builtin_amdgcn_wave_barrier();
    // ---- ffn2 / dropout3 / Mish / ffn1 ----
    float dhd = 0.f;
#pragma unroll 4
    for (int k = 0; k < DC; ++k) dhd = fmaf(va[k], rw.ffn2_w[k * DFF + jf], dhd);
    float dpre;
    {
        const float x = f.pre;
        const float sp = x > 20.f ? x : log1pf(expf(x));
        const float t = tanhf(sp);
        const float sg = 1.f / (1.f + expf(-x));
        dpre = dhd * f.k3 * (t + x * (1.f - t * t) * sg);
    }
    if (w32) {
        row[XR_DPRE + jf] = dpre;
        vb2[jf] = dpre;
    }
    __builtin_amdgcn_wave_barrier();
    float dy2 = dz3;
#pragma unroll 4
    for (int k = 0; k < DFF; ++k) dy2 = fmaf(vb2[k], rw.ffn1_w[k * DC + c], dy2);
    // ---- LayerNorm2 ----
    dzh = dy2 * pv[PV_N2_W + c];
    m1 = xg16_sum(dzh) * (1.f / 16.f);
    m2 = xg16_sum(dzh * f.zhat2) * (1.f / 16.f);
    const float dz2 = f.rstd2 * (dzh - m1 - f.zhat2 * m2);
    const float dout = dz2 * f.k2;
    __builtin_amdgcn_wave_barrier();
    if (w16) {
        row[XR_G2 + c] = dy2 * f.zhat2;
        row[XR_B2 + c] = dy2;
        row[XR_DOUT + c] = dout;
        row[XR_VEC + c] = f.vecc;
        va[c] = dout;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- output_proj ----
    float dvec = 0.f;
#pragma unroll 4
    for (int k = 0; k < DC; ++k) dvec = fmaf(va[k], rw.outp_w[k * DC + c], dvec);
    {
        const float msk = __shfl(f.ms, (c >> 1) * 8);      // mass of head c / 2
        if (w16) {
            row[XR_DVEC + c] = dvec;
            row[XR_DBV + c] = dvec * msk;
            vb2[c] = dvec;
        }
    }
    *reinterpret_cast<f32x2*>(row + XR_SAMP + m * DC + j * 2) = f.acc;
    __builtin_amdgcn_wave_barrier();
    // ---- value_proj: d sampled features of head m (channels 2j, 2j+1) and d mass ----
    const float g0 = vb2[m * DD + 0], g1 = vb2[m * DD + 1];
    const f32x2 w0 = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 0) * DC + j * 2);
    const f32x2 w1v = *reinterpret_cast<const f32x2*>(pv + PV_VAL_W + (m * DD + 1) * DC + j * 2);
    const f32x2 dS = {w0[0] * g0 + w1v[0] * g1, w0[1] * g0 + w1v[1] * g1};
    const float dM = pv[PV_VAL_B + m * DD + 0] * g0 + pv[PV_VAL_B + m * DD + 1] * g1;
    *reinterpret_cast<f32x2*>(ds16 + m * DC + j * 2) = dS;
    // ---- bilinear gather backward (as dpft_xattn_bwd_f32): pyramid gradients, d attention prob, d offsets, d ref ----
    const float* lg = lin + n_off + m * LP;
    const float* offp = lin + m * LP * 2;
    float* dlg = dlin + n_off + m * LP;      // first d(prob), then d(logit)
    float* doff = dlin + m * LP * 2;
    float grx = 0.f, gry = 0.f, sdot = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const int64_t lb = (int64_t)b * H * W * DC + j * 2;
        const float* base = pyr.level[l] + lb;
        // tiny levels: replica (row % R) of the gradient buffer, so that the fp32 atomics of 1600 rows x 8 heads do
        // not serialise on a few hundred addresses
        // The scatter is issued by a DIFFERENT lane layout than the gather: its cost is the number of distinct 64-byte lines an
        // atomic instruction touches (measured: 8x fewer active lanes or a narrower scope change nothing, half the lines
        // halve it).  With lane = (head, channel pair) an instruction covers 8 pixels -- one per head -- and a pixel needs two
        // instructions (channels 2j, 2j+1); with lane = (head % 4, channel) it covers 4 pixels and a pixel needs ONE:
        // half the line requests.  Heads hand their corner weights / offsets over through LDS (sc_w, sc_o).
        float* gl = pyr.grad[l] + (int64_t)(bq % pyr.rep[l]) * a.B * H * W * DC + (int64_t)b * H * W * DC;
        for (int p = 0; p < P; ++p) {
            const int lp = l * P + p;
            const float ox = offp[lp * 2 + 0], oy = offp[lp * 2 + 1];
            const float aw = __expf(lg[lp] - f.mx) * f.inv_den;
            const float lx = f.rx + ox / (float)W, ly = f.ry + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            float ga = 0.f, gw = 0.f, gh = 0.f;
            const bool inb = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            if (!inb && j == 0) *reinterpret_cast<f32x4*>(sc_w + m * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (inb) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1 - lh, hw = 1 - lw;
                const bool k1 = h_lo >= 0 && w_lo >= 0, k2 = h_lo >= 0 && w_hi <= W - 1;
                const bool k3 = h_hi <= H - 1 && w_lo >= 0, k4 = h_hi <= H - 1 && w_hi <= W - 1;
                const int64_t o1 = ((int64_t)h_lo * W + w_lo) * DC, o2 = ((int64_t)h_lo * W + w_hi) * DC;
                const int64_t o3 = ((int64_t)h_hi * W + w_lo) * DC, o4 = ((int64_t)h_hi * W + w_hi) * DC;
                f32x2 v1 = {0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (k1) v1 = *reinterpret_cast<const f32x2*>(base + o1);
                if (k2) v2 = *reinterpret_cast<const f32x2*>(base + o2);
                if (k3) v3 = *reinterpret_cast<const f32x2*>(base + o3);
                if (k4) v4 = *reinterpret_cast<const f32x2*>(base + o4);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                if (j == 0) {
                    *reinterpret_cast<f32x4*>(sc_w + m * 4) = f32x4{k1 ? aw * w1 : 0.f, k2 ? aw * w2 : 0.f, k3 ? aw * w3 : 0.f, k4 ? aw * w4 : 0.f};
                    sc_o[m * 4 + 0] = (int)o1; sc_o[m * 4 + 1] = (int)o2; sc_o[m * 4 + 2] = (int)o3; sc_o[m * 4 + 3] = (int)o4;
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    ga += dS[e] * (w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e]);
                    gh += dS[e] * aw * (-hw * v1[e] - lw * v2[e] + hw * v3[e] + lw * v4[e]);
                    gw += dS[e] * aw * (-hh * v1[e] + hh * v2[e] - lh * v3[e] + lh * v4[e]);
                }
                if (j == 0) {      // the "17th channel": 1 at in-bounds pixels, carries the value_proj bias
                    const float i1 = k1 ? 1.f : 0.f, i2 = k2 ? 1.f : 0.f, i3 = k3 ? 1.f : 0.f, i4 = k4 ? 1.f : 0.f;
                    ga += dM * (w1 * i1 + w2 * i2 + w3 * i3 + w4 * i4);
                    gh += dM * aw * (-hw * i1 - lw * i2 + hw * i3 + lw * i4);
                    gw += dM * aw * (-hh * i1 + hh * i2 - lh * i3 + lh * i4);
                }
            }
            __builtin_amdgcn_wave_barrier();
            {      // scatter: lane = (head % 4, channel); two head halves x four corners, one 64-byte line per 16 lanes
                const int h4 = lane >> 4, ch = lane & 15;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int m2 = h4 + 4 * half;
                    const f32x4 cw = *reinterpret_cast<const f32x4*>(sc_w + m2 * 4);
                    const float val = ds16[m2 * DC + ch];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (cw[k] != 0.f) atomicAdd(gl + sc_o[m2 * 4 + k] + ch, cw[k] * val);
                }
            }
            __builtin_amdgcn_wave_barrier();
            ga = xg8_sum(ga);
            gw = xg8_sum(gw);
            gh = xg8_sum(gh);
            // loc = ref + off / (W,H); w_im = loc_x * W - 0.5  =>  d/d off_x = gw, d/d ref_x = gw * W
            sdot = fmaf(aw, ga, sdot);
            if (j == 0) {
                dlg[lp] = ga;
                doff[lp * 2 + 0] = gw;
                doff[lp * 2 + 1] = gh;
                grx = fmaf((float)W, gw, grx);
                gry = fmaf((float)H, gh, gry);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // softmax backward: d logit = prob * (d prob - sum prob * d prob); lanes j of head m take samples lp = j, j+8, ..
    for (int lp = j; lp < LP; lp += 8) {
        const float aw = __expf(lg[lp] - f.mx) * f.inv_den;
        dlg[lp] = aw * (dlg[lp] - sdot);
    }
    {
        float sx = grx, sy = gry;      // non-leader lanes hold 0
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            sx += __shfl_xor(sx, o);
            sy += __shfl_xor(sy, o);
        }
        if (lane == 0) {
            a.dref[vbq * 2 + 0] = sx;
            a.dref[vbq * 2 + 1] = sy;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- offsets / logits GEMV backward ----
    for (int o = lane; o < NOA; o += 64) row[XR_DLIN + o] = o < n_all ? dlin[o] : 0.f;
    const float qpc = f.y1c + a.pos[(size_t)q * DC + c];
    if (w16) row[XR_QP + c] = qpc;
    float dq = 0.f;
    {
        const int part = lane >> 4;
        for (int o = part; o < n_all; o += 4) {
            const float wgt = o < n_off ? rw.off_w[o * DC + c] : rw.att_w[(o - n_off) * DC + c];
            dq = fmaf(dlin[o], wgt, dq);
        }
        dq += __shfl_xor(dq, 16);
        dq += __shfl_xor(dq, 32);
    }
    if (w16) {
        a.dqp[vbq * DC + c] = dq;
        a.dy1[vbq * DC + c] = dz2 + dq;
    }
}

}  // namespace dpft

using namespace dpft;

static int xf_fill(XfArgs& a, const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed, int V,
                   const int32_t* n_points, const float* y1, const float* pos, const float* ref, float p_drop,
                   const int64_t* seed, int salt, int B, int Q, bool need_grad) {
    DPFT_REQUIRE(pyr && views && packed && n_points && y1 && pos && ref && seed, "xattn_ffn_train: null argument");
    DPFT_REQUIRE(V >= 1 && V <= 4 && B > 0 && Q > 0, "xattn_ffn_train: bad sizes (V=%d, B=%d, Q=%d)", V, B, Q);
    DPFT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "xattn_ffn_train: dropout probability must be in [0,1)");
    DPFT_REQUIRE((int64_t)V * B * Q * DFF < (1ll << 32), "xattn_ffn_train: problem too large for the mask index");
    memset(&a, 0, sizeof(a));
    for (int v = 0; v < V; ++v) {
        const dpft_pyramid* p = pyr + v;
        const int P = n_points[v];
        DPFT_REQUIRE(p->L >= 1 && p->L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && p->L * P * DM * 3 <= NOA,
                     "xattn_ffn_train: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", p->L, P);
        a.pyr[v].L = p->L;
        for (int l = 0; l < p->L; ++l) {
            DPFT_REQUIRE(p->level[l] && p->H[l] > 0 && p->W[l] > 0, "xattn_ffn_train: view %d level %d is invalid", v, l);
            DPFT_REQUIRE(!need_grad || p->grad[l], "xattn_ffn_train: view %d level %d has no gradient buffer", v, l);
            a.pyr[v].level[l] = p->level[l]; a.pyr[v].grad[l] = p->grad[l];
            a.pyr[v].rep[l] = p->grad_replicas[l] > 1 ? p->grad_replicas[l] : 1;
            a.pyr[v].H[l] = p->H[l]; a.pyr[v].W[l] = p->W[l];
        }
        a.P[v] = P;
        a.pv[v] = packed + (size_t)v * PV_FLOATS;
        a.raw[v] = views[v];
    }
    a.y1 = y1; a.pos = pos; a.ref = ref; a.seed = seed; a.salt = salt; a.p_drop = p_drop; a.B = B; a.Q = Q;
    return DPFT_OK;
}

extern "C" int64_t dpft_xattn_ffn_train_row_floats(void) { return XR_FLOATS; }

extern "C" int dpft_xattn_ffn_train_fwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                            int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                            const float* ref, float p_drop, const int64_t* seed, int32_t salt, float* y3,
                                            int32_t B, int32_t Q, dpft_stream_t stream) {
    XfArgs a;
    int rc = xf_fill(a, pyr, views, packed, V, n_points, y1, pos, ref, p_drop, seed, salt, B, Q, false);
    if (rc) return rc;
    DPFT_REQUIRE(y3, "xattn_ffn_train_fwd: null output");
    a.y3 = y3;
    hipLaunchKernelGGL(xf_train_fwd_kernel, dim3(cdiv((int64_t)B * Q, 4), V), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("xattn_ffn_train_fwd");
}

extern "C" int dpft_xattn_ffn_train_bwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                            int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                            const float* ref, float p_drop, const int64_t* seed, int32_t salt,
                                            const float* dy3, float* dy1, float* dqp, float* dref, float* rows,
                                            int32_t B, int32_t Q, dpft_stream_t stream) {
    XfArgs a;
    int rc = xf_fill(a, pyr, views, packed, V, n_points, y1, pos, ref, p_drop, seed, salt, B, Q, true);
    if (rc) return rc;
    DPFT_REQUIRE(dy3 && dy1 && dqp && dref && rows, "xattn_ffn_train_bwd: null argument");
    a.dy3 = dy3; a.dy1 = dy1; a.dqp = dqp; a.dref = dref; a.rows = rows;
    hipLaunchKernelGGL(xf_train_bwd_kernel, dim3(cdiv((int64_t)B * Q, 4), V), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("xattn_ffn_train_bwd");
}
