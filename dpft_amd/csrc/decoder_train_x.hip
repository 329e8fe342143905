// Training path of the decoder's deformable cross-attention + FFN block, all views of one MPFusion layer per launch:
//   y2 = LayerNorm2(y1 + dropout2(output_proj(MSDeformAttn(y1 + pos, ref, pyramid))))
//   y3 = LayerNorm3(y2 + dropout4(ffn2(dropout3(Mish(ffn1(y2))))))
// = MLFusion.forward_cross_attn + forward_ffn, src/dprt/models/fusers/mpfusion.py:150-229 with
// src/dprt/models/layers/ms_deform_attn.py:138-217 (8 heads x head_dim 2, L*P <= 20 samples per head).
// One wave per (view, b, query), lane = (head m, channel pair j) for the gather and lane % 16 = channel elsewhere.
//
// The backward RECOMPUTES the cheap part of its row's forward (offsets/logits GEMV, projections, LayerNorms, FFN: ALU
// only) and takes the one expensive intermediate -- the attention-weighted gathered features and in-bounds masses, 136
// floats per row -- from what the forward saved (`saved`; NULL = gather again), then walks back through it.  Parameter
// gradients are outer-product sums over the B*Q rows: the kernel writes each row's factors into `rows`
// (V,B*Q,XR_FLOATS) and the host framework turns them into the weight gradients (dpft_rows_outer_f32) -- no atomics on
// parameters.
// Feature-pyramid gradients (round 4: segmented scatter).  Every sample adds (attention x bilinear weight) x d(sampled
// features) to four 64-byte pixels; as fp32 atomics that is 3.07 M line requests per call at B = 4 and was 45 % of the
// kernel (the cost follows the number of atomic lines at the memory side, whatever the scope).  Ten of the fifteen
// (view, level) maps are SMALL (<= 2048 pixels: camera 32x57, 16x29; every radar level but the two input-sized ones)
// and take two thirds of those requests with 7-800x reuse per pixel.  For them the backward kernel only RECORDS the
// scatter (4 weights + pixel offset per sample, d(sampled features) once per row: `scratch`), and
// xf_scatter_small_kernel -- one workgroup per (map, batch element, query chunk) -- accumulates the records into an LDS
// image of the map (ds_add_f32) and flushes each touched pixel once: 2.05 M -> 0.16 M atomic lines, and no gradient
// replicas for the tiny maps.  The large maps keep per-sample atomics, now issued per level (4 points at a time).
#include "common.h"
#include "decoder_pack.h"
#include <stdlib.h>

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
    float* fsave;               // (V,B*Q,XS_FLOATS): forward = written (or NULL), backward = read (NULL: gather again)
    float* scratch;             // (V,B*Q,XC_FLOATS) scatter records of the small maps (backward; NULL = atomics everywhere)
    signed char slot[4][DPFT_MAX_LEVELS];      // backward: record slot of (view, level), -1 = large map (atomics)
    int B, Q, salt;
    int P[4];
    float p_drop;
    int exp;                    // timing experiments only (DPFT_XF_EXP; wrong gradients): 1 = no large-map atomics
};
// what the forward saves per row for the backward
constexpr int XS_ACC = 0;                     // [8 heads][16] attention-weighted sampled raw features (= XR_SAMP)
constexpr int XS_MS = 128;                    // [8] in-bounds attention mass per head
constexpr int XS_FLOATS = 136;
// scatter records of one row (small maps)
constexpr int XC_DS = 0;                      // [8 heads][16] d(sampled features)
constexpr int XC_SLOT = 128;                  // per slot: [4 points][8 heads] float4 weights, then [4][8] int pixel offsets
constexpr int XC_SLOT_FLOATS = 160;
constexpr int XC_MAX_SLOTS = 5;               // small maps per view (the smallest first when a view has more)
constexpr int XC_FLOATS = XC_SLOT + XC_MAX_SLOTS * XC_SLOT_FLOATS;      // 928
constexpr int XC_MAX_PIXELS = 2048;           // a small map's LDS image: H*W*64 B <= 128 KB

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

// sums over 16 / 8 consecutive lanes, result in every lane, as DPP operations (quad butterflies, then the mirrored half /
// row): a __shfl_xor is a ds_bpermute, an LDS-pipe round trip, and a row chains dozens of them
template <int CTRL>
__device__ __forceinline__ float xdpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float xg8_sum(float v) {
    v += xdpp<0xB1>(v);       // quad_perm [1,0,3,2]
    v += xdpp<0x4E>(v);       // quad_perm [2,3,0,1]
    v += xdpp<0x141>(v);      // row_half_mirror
    return v;
}
__device__ __forceinline__ float xg16_sum(float v) {
    v = xg8_sum(v);
    v += xdpp<0x140>(v);      // row_mirror
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

// forward of one row; leaves qp[16] | lin[n_all] | vec[32] in the wave's LDS scratch (vec = hd at the end).
// SAVED: the gathered features / masses come from a.fsave (written by the forward kernel) instead of the pyramid.
template <bool SAVED>
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
    if (SAVED) {
        const float* sv = a.fsave + vbq * XS_FLOATS;
        acc = *reinterpret_cast<const f32x2*>(sv + XS_ACC + m * DC + j * 2);
        ms = sv[XS_MS + m];
    }
    for (int l = 0; !SAVED && l < L; ++l) {
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

// Backward: 5 waves per SIMD (96 VGPRs + 26 spilled): the 4 800 rows of a call at B = 4 (1 200 blocks) are then resident
// TOGETHER on 256 CUs x 20 waves; at 4 per SIMD (123 VGPRs) 176 blocks ran as a second round: 226 -> 215 us.  The forward
// loses with the same budget (35.4 -> 37.2 us) and keeps 4.
constexpr int XF_WAVES = 5;
// grid (ceil(B*Q / 4), V), 4 waves (= 4 rows of one view) per block
__global__ __launch_bounds__(256) void xf_train_fwd_kernel(XfArgs a) {
    __shared__ float sm[4][16 + NOA + 32];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wv, view = blockIdx.y;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q, q = bq - b * a.Q;
    float* qp = sm[wv];
    XfFwd f;
    xf_forward_row<false>(a, view, bq, b, q, lane, qp, qp + 16, qp + 16 + NOA, f);
    const size_t vbq = (size_t)view * a.B * a.Q + bq;
    if (lane < 16) a.y3[vbq * DC + lane] = f.y3;
    if (a.fsave) {
        float* sv = a.fsave + vbq * XS_FLOATS;
        *reinterpret_cast<f32x2*>(sv + XS_ACC + (lane >> 3) * DC + (lane & 7) * 2) = f.acc;
        if ((lane & 7) == 0) sv[XS_MS + (lane >> 3)] = f.ms;
    }
}

template <bool SAVED>
__global__ __launch_bounds__(256, XF_WAVES) void xf_train_bwd_kernel(XfArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[4][16 + NOA + 32 + NOA + 32 + 32 + 128 + 128 + 32];
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
    float* sc_w = ds16 + 128;    // [4 points][8 heads][4 corners] attention x bilinear weight (0 = corner not written)
    int* sc_o = reinterpret_cast<int*>(sc_w + 128);      // [4 points][8 heads] offset (floats) of corner (h_lo, w_lo) inside the image
    XfFwd f;
    xf_forward_row<SAVED>(a, view, bq, b, q, lane, qp, lin, vec, f);      // qp = y2, vec = hd, lin = offsets | logits
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
    float* rec = a.scratch ? a.scratch + vbq * XC_FLOATS : nullptr;
    if (rec) *reinterpret_cast<f32x2*>(rec + XC_DS + m * DC + j * 2) = dS;
    // ---- bilinear gather backward: d attention prob, d offsets, d ref; pyramid gradients scattered or recorded ----
    // One level (its 4 points) at a time: all 16 corner loads of the lane are in flight together, the corner weights /
    // pixel offsets of the 32 (point, head) samples go through LDS ONCE, then either
    //   large map: the scatter, issued by a DIFFERENT lane layout than the gather -- its cost is the number of distinct
    //     64-byte lines an atomic instruction touches; with lane = (head % 4, channel) an instruction covers 4 pixels and a
    //     pixel needs ONE instruction (lane = (head, channel pair): two) -- or
    //   small map: the 640-byte record block of the level is copied to the row's scratch (xf_scatter_small_kernel).
    const float* lg = lin + n_off + m * LP;
    const float* offp = lin + m * LP * 2;
    float* dlg = dlin + n_off + m * LP;      // first d(prob), then d(logit)
    float* doff = dlin + m * LP * 2;
    float grx = 0.f, gry = 0.f, sdot = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = pyr.H[l], W = pyr.W[l];
        const float* base = pyr.level[l] + (int64_t)b * H * W * DC + j * 2;
        f32x2 v[4][4];
        float aw[4], hh[4], hw[4], lh[4], lw[4];
        bool kk[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool act = p < P;
            const int lp = act ? l * P + p : l * P;
            const float ox = offp[lp * 2 + 0], oy = offp[lp * 2 + 1];
            aw[p] = act ? __expf(lg[lp] - f.mx) * f.inv_den : 0.f;
            const float lx = f.rx + ox / (float)W, ly = f.ry + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = act && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_lo = (int)hf, w_lo = (int)wf, h_hi = h_lo + 1, w_hi = w_lo + 1;
            lh[p] = h_im - hf; lw[p] = w_im - wf; hh[p] = 1 - lh[p]; hw[p] = 1 - lw[p];
            kk[p][0] = in && h_lo >= 0 && w_lo >= 0;     kk[p][1] = in && h_lo >= 0 && w_hi <= W - 1;
            kk[p][2] = in && h_hi <= H - 1 && w_lo >= 0; kk[p][3] = in && h_hi <= H - 1 && w_hi <= W - 1;
            const int hl = min(max(h_lo, 0), H - 1), hh_ = min(max(h_hi, 0), H - 1);
            const int wl = min(max(w_lo, 0), W - 1), wh_ = min(max(w_hi, 0), W - 1);
            v[p][0] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hl * W + wl) * DC);
            v[p][1] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hl * W + wh_) * DC);
            v[p][2] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hh_ * W + wl) * DC);
            v[p][3] = *reinterpret_cast<const f32x2*>(base + ((int64_t)hh_ * W + wh_) * DC);
            if (j == 0) {
                const float w1 = hh[p] * hw[p], w2 = hh[p] * lw[p], w3 = lh[p] * hw[p], w4 = lh[p] * lw[p];
                *reinterpret_cast<f32x4*>(sc_w + (p * 8 + m) * 4) =
                    f32x4{kk[p][0] ? aw[p] * w1 : 0.f, kk[p][1] ? aw[p] * w2 : 0.f, kk[p][2] ? aw[p] * w3 : 0.f,
                          kk[p][3] ? aw[p] * w4 : 0.f};
                // (h_lo, w_lo) may lie one pixel outside the map: its offset is only ever used with a non-zero weight
                sc_o[p * 8 + m] = in ? (h_lo * W + w_lo) * DC : 0;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p >= P) break;
            const int lp = l * P + p;
            const float w1 = hh[p] * hw[p], w2 = hh[p] * lw[p], w3 = lh[p] * hw[p], w4 = lh[p] * lw[p];
            float ga = 0.f, gw = 0.f, gh = 0.f;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v1 = kk[p][0] ? v[p][0][e] : 0.f, v2 = kk[p][1] ? v[p][1][e] : 0.f;
                const float v3 = kk[p][2] ? v[p][2][e] : 0.f, v4 = kk[p][3] ? v[p][3][e] : 0.f;
                ga += dS[e] * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                gh += dS[e] * aw[p] * (-hw[p] * v1 - lw[p] * v2 + hw[p] * v3 + lw[p] * v4);
                gw += dS[e] * aw[p] * (-hh[p] * v1 + hh[p] * v2 - lh[p] * v3 + lh[p] * v4);
            }
            if (j == 0) {      // the "17th channel": 1 at in-bounds pixels, carries the value_proj bias
                const float i1 = kk[p][0] ? 1.f : 0.f, i2 = kk[p][1] ? 1.f : 0.f, i3 = kk[p][2] ? 1.f : 0.f, i4 = kk[p][3] ? 1.f : 0.f;
                ga += dM * (w1 * i1 + w2 * i2 + w3 * i3 + w4 * i4);
                gh += dM * aw[p] * (-hw[p] * i1 - lw[p] * i2 + hw[p] * i3 + lw[p] * i4);
                gw += dM * aw[p] * (-hh[p] * i1 + hh[p] * i2 - lh[p] * i3 + lh[p] * i4);
            }
            ga = xg8_sum(ga);
            gw = xg8_sum(gw);
            gh = xg8_sum(gh);
            // loc = ref + off / (W,H); w_im = loc_x * W - 0.5  =>  d/d off_x = gw, d/d ref_x = gw * W
            sdot = fmaf(aw[p], ga, sdot);
            if (j == 0) {
                dlg[lp] = ga;
                doff[lp * 2 + 0] = gw;
                doff[lp * 2 + 1] = gh;
                grx = fmaf((float)W, gw, grx);
                gry = fmaf((float)H, gh, gry);
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int slot = rec ? (int)a.slot[view][l] : -1;      // wave-uniform
        if (slot >= 0) {
            // 32 float4 weights + 32 int offsets, in the LDS order: lanes 0..31 copy one float4, lanes 32..63 one int
            float* dst = rec + XC_SLOT + slot * XC_SLOT_FLOATS;
            if (lane < 32) reinterpret_cast<f32x4*>(dst)[lane] = reinterpret_cast<const f32x4*>(sc_w)[lane];
            else reinterpret_cast<int*>(dst + 128)[lane - 32] = sc_o[lane - 32];
        } else if (!(a.exp & 1)) {
            // tiny maps without a record buffer: replica (row % R) of the gradient buffer, so that the fp32 atomics of
            // 1600 rows x 8 heads do not serialise on a few hundred addresses
            float* gl = pyr.grad[l] + (int64_t)(bq % pyr.rep[l]) * a.B * H * W * DC + (int64_t)b * H * W * DC;
            const int h4 = lane >> 4, ch = lane & 15, W16 = W * DC;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m2 = h4 + 4 * half;
                const float val = ds16[m2 * DC + ch];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 cw = *reinterpret_cast<const f32x4*>(sc_w + (p * 8 + m2) * 4);
                    float* g00 = gl + sc_o[p * 8 + m2] + ch;
                    if (cw[0] != 0.f) atomicAdd(g00, cw[0] * val);
                    if (cw[1] != 0.f) atomicAdd(g00 + DC, cw[1] * val);
                    if (cw[2] != 0.f) atomicAdd(g00 + W16, cw[2] * val);
                    if (cw[3] != 0.f) atomicAdd(g00 + W16 + DC, cw[3] * val);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    // softmax backward: d logit = prob * (d prob - sum prob * d prob); lanes j of head m take samples lp = j, j+8, ..
    for (int lp = j; lp < LP; lp += 8) {
        const float awp = __expf(lg[lp] - f.mx) * f.inv_den;
        dlg[lp] = awp * (dlg[lp] - sdot);
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

// Small-map scatter: workgroup = (query chunk, small map = (view, slot), batch element).  The map's gradient image lives in
// LDS (H*W*16 floats, zeroed); every wave walks rows of the chunk: d(sampled features) of the row (lane = (head % 4, channel),
// two head halves) x the 32 (point, head) records of the map -> ds_add_f32 into the image; then every touched element is
// added to the gradient buffer ONCE (fp32 atomic: other chunks, other decoder layers and -- for shared pyramids -- other
// calls accumulate into the same buffer).
struct XsArgs {
    const float* scratch;          // (V,B*Q,XC_FLOATS)
    float* grad[4 * XC_MAX_SLOTS];               // per small map
    int H[4 * XC_MAX_SLOTS], W[4 * XC_MAX_SLOTS], view[4 * XC_MAX_SLOTS], slot[4 * XC_MAX_SLOTS];
    int B, Q, qchunk;
    int exp;      // timing experiments (DPFT_XF_EXP): 2 = no LDS atomics, 4 = no flush, 8 = no row loop
};
__global__ __launch_bounds__(512) void xf_scatter_small_kernel(XsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = blockIdx.y, b = blockIdx.z;
    const int H = a.H[mi], W = a.W[mi], view = a.view[mi], slot = a.slot[mi];
    const int n = H * W * DC, W16 = W * DC;
    for (int i = tid * 4; i < n; i += 512 * 4) *reinterpret_cast<f32x4*>(img + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int q0 = blockIdx.x * a.qchunk, q1 = min(a.Q, q0 + a.qchunk);
    const int h4 = lane >> 4, ch = lane & 15;
    for (int q = q0 + wave; q < q1 && !(a.exp & 8); q += 8) {
        const float* rec = a.scratch + (((size_t)view * a.B + b) * a.Q + q) * XC_FLOATS;
        const float* wrec = rec + XC_SLOT + slot * XC_SLOT_FLOATS;
        const int* orec = reinterpret_cast<const int*>(wrec + 128);
        float val[2];
        f32x4 cw[2][4];
        int off[2][4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m2 = h4 + 4 * half;
            val[half] = rec[XC_DS + m2 * DC + ch];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                cw[half][p] = *reinterpret_cast<const f32x4*>(wrec + (p * 8 + m2) * 4);
                off[half][p] = orec[p * 8 + m2];
            }
        }
        if (a.exp & 2) {
            float acc = val[0] + val[1];
            for (int half = 0; half < 2; ++half)
                for (int p = 0; p < 4; ++p) acc += cw[half][p][0] + cw[half][p][3] + (float)off[half][p];
            if (acc == 12345.678f) img[0] = acc;
            continue;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float* g00 = img + off[half][p] + ch;
                const f32x4 w4 = cw[half][p];
                if (w4[0] != 0.f) atomicAdd(g00, w4[0] * val[half]);
                if (w4[1] != 0.f) atomicAdd(g00 + DC, w4[1] * val[half]);
                if (w4[2] != 0.f) atomicAdd(g00 + W16, w4[2] * val[half]);
                if (w4[3] != 0.f) atomicAdd(g00 + W16 + DC, w4[3] * val[half]);
            }
    }
    __syncthreads();
    float* g = a.grad[mi] + (size_t)b * n;
    for (int i = tid; i < n && !(a.exp & 4); i += 512) {
        const float v = img[i];
        if (v != 0.f) atomicAdd(g + i, v);
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
    memset(a.slot, -1, sizeof(a.slot));
    for (int v = 0; v < V; ++v) {
        const dpft_pyramid* p = pyr + v;
        const int P = n_points[v];
        DPFT_REQUIRE(p->L >= 1 && p->L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && p->L * P * DM * 3 <= NOA,
                     "xattn_ffn_train: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", p->L, P);
        a.pyr[v].L = p->L;
        for (int l = 0; l < p->L; ++l) {
            DPFT_REQUIRE(p->level[l] && p->H[l] > 0 && p->W[l] > 0, "xattn_ffn_train: view %d level %d is invalid", v, l);
            DPFT_REQUIRE(!need_grad || p->grad[l], "xattn_ffn_train: view %d level %d has no gradient buffer", v, l);
            DPFT_REQUIRE((int64_t)p->H[l] * p->W[l] * DC < (1ll << 31), "xattn_ffn_train: view %d level %d is too large", v, l);
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
extern "C" int64_t dpft_xattn_ffn_train_saved_floats(void) { return XS_FLOATS; }
extern "C" int64_t dpft_xattn_ffn_train_scratch_floats(void) { return XC_FLOATS; }

extern "C" int dpft_xattn_ffn_train_fwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                            int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                            const float* ref, float p_drop, const int64_t* seed, int32_t salt, float* y3,
                                            float* saved, int32_t B, int32_t Q, dpft_stream_t stream) {
    XfArgs a;
    int rc = xf_fill(a, pyr, views, packed, V, n_points, y1, pos, ref, p_drop, seed, salt, B, Q, false);
    if (rc) return rc;
    DPFT_REQUIRE(y3, "xattn_ffn_train_fwd: null output");
    a.y3 = y3;
    a.fsave = saved;
    hipLaunchKernelGGL(xf_train_fwd_kernel, dim3(cdiv((int64_t)B * Q, 4), V), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("xattn_ffn_train_fwd");
}

extern "C" int dpft_xattn_ffn_train_bwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* views, const float* packed,
                                            int32_t V, const int32_t* n_points, const float* y1, const float* pos,
                                            const float* ref, float p_drop, const int64_t* seed, int32_t salt,
                                            const float* saved, const float* dy3, float* dy1, float* dqp, float* dref,
                                            float* rows, float* scratch, int32_t B, int32_t Q, dpft_stream_t stream) {
    XfArgs a;
    int rc = xf_fill(a, pyr, views, packed, V, n_points, y1, pos, ref, p_drop, seed, salt, B, Q, true);
    if (rc) return rc;
    DPFT_REQUIRE(dy3 && dy1 && dqp && dref && rows, "xattn_ffn_train_bwd: null argument");
    a.dy3 = dy3; a.dy1 = dy1; a.dqp = dqp; a.dref = dref; a.rows = rows;
    a.fsave = const_cast<float*>(saved);
    static const int exp = getenv("DPFT_XF_EXP") ? atoi(getenv("DPFT_XF_EXP")) : 0;
    a.exp = exp;
    // small maps (<= XC_MAX_PIXELS pixels, unreplicated gradient buffer) are recorded and scattered through LDS; a view with
    // more than XC_MAX_SLOTS of them keeps atomics for its largest ones
    XsArgs xs;
    memset(&xs, 0, sizeof(xs));
    int n_maps = 0, max_px = 0;
    static const bool allow = !(getenv("DPFT_XF_SCATTER") && atoi(getenv("DPFT_XF_SCATTER")) == 0);      // A/B switch
    if (scratch && allow) {
        a.scratch = scratch;
        for (int v = 0; v < V; ++v) {
            int order[DPFT_MAX_LEVELS], n = 0;
            for (int l = 0; l < a.pyr[v].L; ++l)
                if (a.pyr[v].H[l] * a.pyr[v].W[l] <= XC_MAX_PIXELS && a.pyr[v].rep[l] == 1) order[n++] = l;
            for (int i = 1; i < n; ++i)                      // ascending pixel count
                for (int k = i; k > 0 && a.pyr[v].H[order[k]] * a.pyr[v].W[order[k]] < a.pyr[v].H[order[k - 1]] * a.pyr[v].W[order[k - 1]]; --k) {
                    const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t;
                }
            for (int i = 0; i < n && i < XC_MAX_SLOTS; ++i) {
                const int l = order[i];
                a.slot[v][l] = (signed char)i;
                xs.grad[n_maps] = a.pyr[v].grad[l];
                xs.H[n_maps] = a.pyr[v].H[l]; xs.W[n_maps] = a.pyr[v].W[l];
                xs.view[n_maps] = v; xs.slot[n_maps] = i;
                max_px = max_px > xs.H[n_maps] * xs.W[n_maps] ? max_px : xs.H[n_maps] * xs.W[n_maps];
                ++n_maps;
            }
        }
        if (n_maps == 0) a.scratch = nullptr;
    }
    if (saved)
        hipLaunchKernelGGL(xf_train_bwd_kernel<true>, dim3(cdiv((int64_t)B * Q, 4), V), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(xf_train_bwd_kernel<false>, dim3(cdiv((int64_t)B * Q, 4), V), dim3(256), 0, (hipStream_t)stream, a);
    rc = check_launch("xattn_ffn_train_bwd");
    if (rc || n_maps == 0) return rc;
    // about one workgroup per CU: fewer query chunks = fewer flushes of a map (one atomic line per touched pixel and chunk)
    int nchunk = kNumCU / (n_maps * B);
    nchunk = nchunk < 1 ? 1 : (nchunk > 8 ? 8 : nchunk);
    xs.scratch = scratch; xs.B = B; xs.Q = Q; xs.qchunk = cdiv(Q, nchunk); xs.exp = exp;
    nchunk = cdiv(Q, xs.qchunk);
    const size_t lds = (size_t)max_px * DC * sizeof(float);
    static LdsGrant grant;      // (per device)
    (void)lds_grant(grant, reinterpret_cast<const void*>(xf_scatter_small_kernel), (size_t)XC_MAX_PIXELS * DC * sizeof(float));
    hipLaunchKernelGGL(xf_scatter_small_kernel, dim3(nchunk, n_maps, B), dim3(512), lds, (hipStream_t)stream, xs);
    return check_launch("xattn_ffn_train_bwd (small-map scatter)");
}
