// Fused inference path of the iterative fusion decoder (eval mode, no autograd) for d_model = 16, 8 heads
// (head_dim 2), 5 levels x 4 points, Mish FFN, LayerNorm, 'linear' view reduction, 3-layer linear heads --
// i.e. IMPFusion / MPFusion / MLFusion / LinearDetectionHead of the reference at config/kradar*.json:
//   src/dprt/models/fusers/mpfusion.py:122-148 (self attention), :150-208 + layers/ms_deform_attn.py:138-217
//   (deformable cross attention), :210-229 (FFN), :416-470,:472-514 (view reduction), :617-696 (reference
//   points), :698-745 (iteration), src/dprt/models/heads/detection.py:252-275 (head).
// The eager decoder is ~700 launches of B*400x16-sized ops per forward (launch-bound: ~5 ms even when
// replayed from a hipGraph); here one iteration is 3 kernels per view-set:
//   K1 decoder_selfattn_kernel  : all views; K/V of the 400 keys in LDS, one thread per (query, head),
//                                 out_proj + residual + LayerNorm1 in the epilogue
//   K2 decoder_xattn_ffn_kernel : per view; one wave per (b, query): offsets/logits GEMV + softmax,
//                                 sample-then-project gather on the NHWC pyramid, output_proj + LN2,
//                                 FFN (Mish) + LN3
//   K3 decoder_head_kernel      : 48->16 view reduction, 4 head MLPs, center += previous, reference points of
//                                 every view for the next iteration
#include "common.h"

namespace dpft {

constexpr int DC = 16, DM = 8, DD = 2, DFF = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct ViewW {   // device pointers, one MLFusion
    const float *in_w, *in_b, *out_w, *out_b, *n1_w, *n1_b;
    const float *off_w, *off_b, *att_w, *att_b, *val_w, *val_b, *outp_w, *outp_b, *n2_w, *n2_b;
    const float *f1_w, *f1_b, *f2_w, *f2_b, *n3_w, *n3_b;
};
struct SelfAttnArgs {
    ViewW w[4];
    const float* query;   // (B,Q,16), or (Q,16) broadcast over the batch when qstride == 0
    const float* pos;     // (Q,16)
    float* y1;            // (V,B,Q,16)
    int B, Q, V;
    long qstride;
};

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

constexpr int QT = 32;   // queries per block

__global__ __launch_bounds__(256) void decoder_selfattn_kernel(SelfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q;
    float* Ks = sm;                 // [Q][16]
    float* Vs = sm + Q * DC;        // [Q][16]
    float* Ws = Vs + Q * DC;        // in_proj 48x16 + bias 48
    float* At = Ws + 48 * 16 + 48;  // [QT][16] attention output tile
    const int tid = threadIdx.x;
    const int view = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * QT;
    const ViewW& w = a.w[view];
    for (int i = tid; i < 48 * 16; i += 256) Ws[i] = w.in_w[i];
    if (tid < 48) Ws[48 * 16 + tid] = w.in_b[tid];
    __syncthreads();
    const float* xb = a.query + (size_t)b * a.qstride;
    // K = (x+pos) Wk^T + bk ; V = x Wv^T + bv for every key of this sample
    for (int k = tid; k < Q; k += 256) {
        float x[DC], xp[DC];
#pragma unroll
        for (int c = 0; c < DC; c += 4) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + (size_t)k * DC + c);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(a.pos + (size_t)k * DC + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[c + e] = xv[e]; xp[c + e] = xv[e] + pv[e]; }
        }
#pragma unroll
        for (int o = 0; o < DC; ++o) {
            float sk = Ws[48 * 16 + 16 + o], sv = Ws[48 * 16 + 32 + o];
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                sk = fmaf(Ws[(16 + o) * 16 + c], xp[c], sk);
                sv = fmaf(Ws[(32 + o) * 16 + c], x[c], sv);
            }
            Ks[k * DC + o] = sk;
            Vs[k * DC + o] = sv;
        }
    }
    __syncthreads();
    // one thread per (query, head): scaled dot-product attention over all keys
    const int ql = tid >> 3, h = tid & 7;
    const int q = q0 + ql;
    if (q < Q) {
        float qh[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float s = Ws[48 * 16 + h * 2 + d];
#pragma unroll
            for (int c = 0; c < DC; ++c)
                s = fmaf(Ws[(h * 2 + d) * 16 + c], xb[(size_t)q * DC + c] + a.pos[(size_t)q * DC + c], s);
            qh[d] = s * 0.70710678118654752f;      // 1/sqrt(head_dim)
        }
        float mx = -INFINITY;
        for (int k = 0; k < Q; ++k) {
            const f32x2 kv = *reinterpret_cast<const f32x2*>(&Ks[k * DC + h * 2]);
            mx = fmaxf(mx, qh[0] * kv[0] + qh[1] * kv[1]);
        }
        float den = 0.f, o0 = 0.f, o1 = 0.f;
        for (int k = 0; k < Q; ++k) {
            const f32x2 kv = *reinterpret_cast<const f32x2*>(&Ks[k * DC + h * 2]);
            const f32x2 vv = *reinterpret_cast<const f32x2*>(&Vs[k * DC + h * 2]);
            const float p = __expf(qh[0] * kv[0] + qh[1] * kv[1] - mx);
            den += p;
            o0 = fmaf(p, vv[0], o0);
            o1 = fmaf(p, vv[1], o1);
        }
        At[ql * DC + h * 2 + 0] = o0 / den;
        At[ql * DC + h * 2 + 1] = o1 / den;
    }
    __syncthreads();
    // out_proj + residual + LayerNorm1: 16 lanes per query (2 passes of 16 queries)
    for (int pass = 0; pass < 2; ++pass) {
        const int ql2 = pass * 16 + (tid >> 4), c = tid & 15;
        const int q2 = q0 + ql2;
        float v = 0.f;
        if (q2 < Q) {
            v = w.out_b[c];
#pragma unroll
            for (int j = 0; j < DC; ++j) v = fmaf(w.out_w[c * DC + j], At[ql2 * DC + j], v);
            v += xb[(size_t)q2 * DC + c];
        }
        v = layernorm16(v, w.n1_w[c], w.n1_b[c]);
        if (q2 < Q) a.y1[(((size_t)view * a.B + b) * Q + q2) * DC + c] = v;
    }
}

struct Pyr5 {
    const float* level[DPFT_MAX_LEVELS];
    int H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int L;
};
struct XattnFfnArgs {
    Pyr5 pyr;
    ViewW w;
    const float* y1;    // (B,Q,16)
    const float* pos;   // (Q,16)
    const float* ref;   // (B,Q,2)
    float* y3;          // (B,Q,16)
    int B, Q, P;
};

__device__ __forceinline__ float group8_sum_d(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

__device__ __forceinline__ float mishf(float x) {
    // torch: x * tanh(softplus(x)), softplus threshold 20
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

// one wave per (b, q); 4 waves per block.  Per-wave LDS scratch: qp[16] | lin[480] | vec[32]
__global__ __launch_bounds__(256) void decoder_xattn_ffn_kernel(XattnFfnArgs a) {
    __shared__ float sm[4][16 + 480 + 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bq = blockIdx.x * 4 + wv;
    if (bq >= a.B * a.Q) return;
    float* qp = sm[wv];
    float* lin = qp + 16;     // [0,320): offsets (m,l,p,xy) ; [320,480): attention logits (m, l*P+p)
    float* vec = lin + 480;
    const int b = bq / a.Q, q = bq - b * a.Q;
    const ViewW& w = a.w;
    const int L = a.pyr.L, P = a.P, LP = L * P;
    const float y1c = lane < 16 ? a.y1[(size_t)bq * DC + lane] : 0.f;
    if (lane < 16) qp[lane] = y1c + a.pos[(size_t)q * DC + lane];
    __builtin_amdgcn_wave_barrier();
    // GEMV: 320 offsets + 160 logits from the 16-vector qp
    const int n_off = DM * LP * 2, n_att = DM * LP;
    for (int o = lane; o < n_off + n_att; o += 64) {
        const float* wr = o < n_off ? w.off_w + (size_t)o * DC : w.att_w + (size_t)(o - n_off) * DC;
        float s = o < n_off ? w.off_b[o] : w.att_b[o - n_off];
#pragma unroll
        for (int c = 0; c < DC; c += 4) {
            const f32x4 wv4 = *reinterpret_cast<const f32x4*>(wr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(wv4[e], qp[c + e], s);
        }
        lin[o] = s;
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
    const float rx = a.ref[bq * 2 + 0], ry = a.ref[bq * 2 + 1];
    const float* offp = lin + m * LP * 2;
    f32x2 acc = {0.f, 0.f};
    float ms = 0.f;
    for (int l = 0; l < L; ++l) {
        const int H = a.pyr.H[l], W = a.pyr.W[l];
        const float* base = a.pyr.level[l] + (int64_t)b * H * W * DC + j * 2;
        for (int p = 0; p < P; ++p) {
            const float ox = offp[(l * P + p) * 2 + 0], oy = offp[(l * P + p) * 2 + 1];
            const float aw = __expf(lg[l * P + p] - mx) * inv_den;
            const float lx = rx + ox / (float)W, ly = ry + oy / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int h_lo = (int)floorf(h_im), w_lo = (int)floorf(w_im);
                const int h_hi = h_lo + 1, w_hi = w_lo + 1;
                const float lh = h_im - h_lo, lw = w_im - w_lo, hh = 1 - lh, hw = 1 - lw;
                const bool k1 = h_lo >= 0 && w_lo >= 0, k2 = h_lo >= 0 && w_hi <= W - 1;
                const bool k3 = h_hi <= H - 1 && w_lo >= 0, k4 = h_hi <= H - 1 && w_hi <= W - 1;
                f32x2 v1 = {0.f, 0.f}, v2 = v1, v3 = v1, v4 = v1;
                if (k1) v1 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_lo * W + w_lo) * DC);
                if (k2) v2 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_lo * W + w_hi) * DC);
                if (k3) v3 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_hi * W + w_lo) * DC);
                if (k4) v4 = *reinterpret_cast<const f32x2*>(base + ((int64_t)h_hi * W + w_hi) * DC);
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                acc += aw * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                ms += aw * ((k1 ? w1 : 0.f) + (k2 ? w2 : 0.f) + (k3 ? w3 : 0.f) + (k4 ? w4 : 0.f));
            }
        }
    }
    // value_proj on the sampled features (+ bias * in-bounds mass), head m -> channels 2m, 2m+1
    const f32x2 wv0 = *reinterpret_cast<const f32x2*>(w.val_w + (m * DD + 0) * DC + j * 2);
    const f32x2 wv1 = *reinterpret_cast<const f32x2*>(w.val_w + (m * DD + 1) * DC + j * 2);
    const float o0 = group8_sum_d(wv0[0] * acc[0] + wv0[1] * acc[1]);
    const float o1 = group8_sum_d(wv1[0] * acc[0] + wv1[1] * acc[1]);
    if (j == 0) {
        vec[m * 2 + 0] = o0 + w.val_b[m * 2 + 0] * ms;
        vec[m * 2 + 1] = o1 + w.val_b[m * 2 + 1] * ms;
    }
    __builtin_amdgcn_wave_barrier();
    // output_proj + residual + LayerNorm2 (lanes 0..15 = channels; other lanes mirror them)
    const int c = lane & 15;
    float v = w.outp_b[c];
#pragma unroll
    for (int k = 0; k < DC; ++k) v = fmaf(w.outp_w[c * DC + k], vec[k], v);
    v += __shfl(y1c, c);
    const float y2 = layernorm16(v, w.n2_w[c], w.n2_b[c]);
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) qp[lane] = y2;            // reuse qp for y2
    __builtin_amdgcn_wave_barrier();
    // FFN: 16 -> 32 (Mish) -> 16, residual, LayerNorm3
    if (lane < DFF) {
        float hsum = w.f1_b[lane];
#pragma unroll
        for (int k = 0; k < DC; ++k) hsum = fmaf(w.f1_w[lane * DC + k], qp[k], hsum);
        vec[lane] = mishf(hsum);
    }
    __builtin_amdgcn_wave_barrier();
    float f = w.f2_b[c];
#pragma unroll
    for (int k = 0; k < DFF; ++k) f = fmaf(w.f2_w[c * DFF + k], vec[k], f);
    f += y2;
    const float y3 = layernorm16(f, w.n3_w[c], w.n3_b[c]);
    if (lane < 16) a.y3[(size_t)bq * DC + lane] = y3;
}

struct HeadArgs {
    const float* y3;            // (V,B,Q,16), may be null (reference points only)
    const float* red_w;         // (16, 16*V)
    const float* hw[4][3];      // center/size/angle/class x (Linear0, Linear3, Linear6)
    const float* prev_center;   // (B,Q,3)
    const float* T[4];          // (B,4,4)
    const float* Pm[4];         // (B,prow,4)
    const int64_t* shape[4];    // (B,2) = H, W
    int prow[4], flag[4];
    float* query_out;           // (B,Q,16)
    float *center, *size, *angle, *cls;
    float* refs;                // (V,B,Q,2)
    int B, Q, V, ncls;
};

__device__ __forceinline__ void mlp3(const float* x, const float* w0, const float* w3, const float* w6, int nout,
                                     float* out) {
    float h1[DC], h2[DC];
#pragma unroll
    for (int o = 0; o < DC; ++o) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < DC; ++k) s = fmaf(w0[o * DC + k], x[k], s);
        h1[o] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int o = 0; o < DC; ++o) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < DC; ++k) s = fmaf(w3[o * DC + k], h1[k], s);
        h2[o] = fmaxf(s, 0.f);
    }
    for (int o = 0; o < nout; ++o) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < DC; ++k) s = fmaf(w6[o * DC + k], h2[k], s);
        out[o] = s;
    }
}

__global__ __launch_bounds__(64) void decoder_head_kernel(HeadArgs a) {
    const int bq = blockIdx.x * 64 + threadIdx.x;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q;
    float cx, cy, cz;
    if (a.y3 != nullptr) {
        // view reduction: queries.view(B,N,C*V) is channel-major / view-minor (mpfusion.py:436-438)
        float x[DC];
#pragma unroll
        for (int o = 0; o < DC; ++o) x[o] = 0.f;
        for (int v = 0; v < a.V; ++v) {
            float yv[DC];
#pragma unroll
            for (int c = 0; c < DC; c += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(a.y3 + (((size_t)v * a.B * a.Q) + bq) * DC + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[c + e] = t[e];
            }
#pragma unroll
            for (int o = 0; o < DC; ++o)
#pragma unroll
                for (int c = 0; c < DC; ++c) x[o] = fmaf(a.red_w[o * DC * a.V + c * a.V + v], yv[c], x[o]);
        }
#pragma unroll
        for (int o = 0; o < DC; ++o) a.query_out[(size_t)bq * DC + o] = x[o];
        float r[3];
        mlp3(x, a.hw[0][0], a.hw[0][1], a.hw[0][2], 3, r);
        cx = r[0] + a.prev_center[bq * 3 + 0];
        cy = r[1] + a.prev_center[bq * 3 + 1];
        cz = r[2] + a.prev_center[bq * 3 + 2];
        a.center[bq * 3 + 0] = cx; a.center[bq * 3 + 1] = cy; a.center[bq * 3 + 2] = cz;
        mlp3(x, a.hw[1][0], a.hw[1][1], a.hw[1][2], 3, r);
        for (int i = 0; i < 3; ++i) a.size[bq * 3 + i] = fmaxf(r[i], 0.f);
        mlp3(x, a.hw[2][0], a.hw[2][1], a.hw[2][2], 2, r);
        for (int i = 0; i < 2; ++i) a.angle[bq * 2 + i] = tanhf(r[i]);
        float rc[8];
        mlp3(x, a.hw[3][0], a.hw[3][1], a.hw[3][2], a.ncls, rc);
        for (int i = 0; i < a.ncls; ++i) a.cls[bq * a.ncls + i] = rc[i];
    } else {
        cx = a.prev_center[bq * 3 + 0]; cy = a.prev_center[bq * 3 + 1]; cz = a.prev_center[bq * 3 + 2];
    }
    if (a.refs == nullptr) return;
    // reference points of every view for the next iteration (mpfusion.py:617-696)
    const float RAD2DEG = 57.29577951308232f;
    for (int v = 0; v < a.V; ++v) {
        float p0 = cx, p1 = cy, p2 = cz;
        if (a.flag[v]) {
            const float* T = a.T[v] + (size_t)b * 16;
            const float tx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];
            const float ty = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
            const float tz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
            const float r = sqrtf(tx * tx + ty * ty + tz * tz);
            p0 = r;
            p1 = atan2f(ty, tx) * RAD2DEG;
            p2 = asinf(r != 0.f ? tz / r : 0.f) * RAD2DEG;
        }
        const float* Pm = a.Pm[v] + (size_t)b * a.prow[v] * 4;
        float u = Pm[0] * p0 + Pm[1] * p1 + Pm[2] * p2 + Pm[3];
        float vv = Pm[4] * p0 + Pm[5] * p1 + Pm[6] * p2 + Pm[7];
        const float wq = Pm[8] * p0 + Pm[9] * p1 + Pm[10] * p2 + Pm[11];
        if (wq != 0.f) { u /= wq; vv /= wq; }
        const float Hs = (float)a.shape[v][b * 2 + 0], Ws = (float)a.shape[v][b * 2 + 1];
        u = fminf(fmaxf(u / Ws, 0.f), 1.f);
        vv = fminf(fmaxf(vv / Hs, 0.f), 1.f);
        float* rp = a.refs + (((size_t)v * a.B * a.Q) + bq) * 2;
        rp[0] = u;
        rp[1] = vv;
    }
}

static void fill_view(ViewW& d, const dpft_decoder_view* s) {
    d.in_w = s->in_proj_w; d.in_b = s->in_proj_b; d.out_w = s->out_proj_w; d.out_b = s->out_proj_b;
    d.n1_w = s->norm1_w; d.n1_b = s->norm1_b;
    d.off_w = s->off_w; d.off_b = s->off_b; d.att_w = s->att_w; d.att_b = s->att_b;
    d.val_w = s->val_w; d.val_b = s->val_b; d.outp_w = s->outp_w; d.outp_b = s->outp_b;
    d.n2_w = s->norm2_w; d.n2_b = s->norm2_b;
    d.f1_w = s->ffn1_w; d.f1_b = s->ffn1_b; d.f2_w = s->ffn2_w; d.f2_b = s->ffn2_b;
    d.n3_w = s->norm3_w; d.n3_b = s->norm3_b;
}

}  // namespace dpft

using namespace dpft;

#define RC(call)              \
    do {                      \
        int rc_ = (call);     \
        if (rc_) return rc_;  \
    } while (0)

extern "C" int dpft_decoder_selfattn_fwd_f32(const float* query, const float* pos, const dpft_decoder_view* views,
                                             int32_t V, float* y1, int32_t B, int32_t Q, dpft_stream_t stream) {
    DPFT_REQUIRE(query && pos && views && y1 && V >= 1 && V <= 4 && B > 0 && Q > 0, "decoder_selfattn: bad arguments");
    SelfAttnArgs a;
    for (int v = 0; v < V; ++v) fill_view(a.w[v], views + v);
    a.query = query; a.pos = pos; a.y1 = y1; a.B = B; a.Q = Q; a.V = V; a.qstride = (long)Q * DC;
    const size_t lds = ((size_t)Q * 32 + 48 * 16 + 48 + QT * DC) * sizeof(float);
    DPFT_REQUIRE(lds <= 160 * 1024, "decoder_selfattn: %d queries do not fit the LDS", Q);
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_selfattn_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    hipLaunchKernelGGL(decoder_selfattn_kernel, dim3(cdiv(Q, QT), V, B), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("decoder_selfattn");
}

extern "C" int dpft_decoder_xattn_ffn_fwd_f32(const dpft_pyramid* pyr, const dpft_decoder_view* view, const float* y1,
                                              const float* pos, const float* ref, float* y3, int32_t B, int32_t Q,
                                              int32_t P, dpft_stream_t stream) {
    DPFT_REQUIRE(pyr && view && y1 && pos && ref && y3 && B > 0 && Q > 0, "decoder_xattn_ffn: bad arguments");
    DPFT_REQUIRE(pyr->L >= 1 && pyr->L <= DPFT_MAX_LEVELS && pyr->L * P * DM * 3 <= 480,
                 "decoder_xattn_ffn: L*P = %d*%d exceeds the fused kernel's budget (L*P <= 20)", pyr->L, P);
    XattnFfnArgs a;
    a.pyr.L = pyr->L;
    for (int l = 0; l < pyr->L; ++l) {
        DPFT_REQUIRE(pyr->level[l], "decoder_xattn_ffn: level %d is null", l);
        a.pyr.level[l] = pyr->level[l]; a.pyr.H[l] = pyr->H[l]; a.pyr.W[l] = pyr->W[l];
    }
    fill_view(a.w, view);
    a.y1 = y1; a.pos = pos; a.ref = ref; a.y3 = y3; a.B = B; a.Q = Q; a.P = P;
    hipLaunchKernelGGL(decoder_xattn_ffn_kernel, dim3(cdiv((int64_t)B * Q, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("decoder_xattn_ffn");
}

extern "C" int dpft_decoder_head_fwd_f32(const dpft_decoder_head* h, int32_t B, int32_t Q, int32_t V, dpft_stream_t stream) {
    DPFT_REQUIRE(h && B > 0 && Q > 0 && V >= 1 && V <= 4, "decoder_head: bad arguments");
    DPFT_REQUIRE(h->prev_center, "decoder_head: prev_center is null");
    DPFT_REQUIRE(h->y3 == nullptr || (h->red_w && h->query_out && h->center && h->size && h->angle && h->cls),
                 "decoder_head: missing output / weight pointers");
    DPFT_REQUIRE(h->num_classes >= 1 && h->num_classes <= 8, "decoder_head: num_classes must be in [1,8]");
    HeadArgs a;
    a.y3 = h->y3; a.red_w = h->red_w; a.prev_center = h->prev_center;
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) a.hw[i][k] = h->head_w[i][k];
    for (int v = 0; v < V; ++v) {
        a.T[v] = h->T[v]; a.Pm[v] = h->P[v]; a.shape[v] = h->shape[v]; a.prow[v] = h->p_rows[v]; a.flag[v] = h->has_t[v];
        DPFT_REQUIRE(h->refs == nullptr || (a.Pm[v] && a.shape[v] && (a.T[v] || !a.flag[v]) && a.prow[v] >= 3),
                     "decoder_head: projection inputs of view %d missing", v);
    }
    a.query_out = h->query_out; a.center = h->center; a.size = h->size; a.angle = h->angle; a.cls = h->cls;
    a.refs = h->refs; a.B = B; a.Q = Q; a.V = V; a.ncls = h->num_classes;
    hipLaunchKernelGGL(decoder_head_kernel, dim3(cdiv((int64_t)B * Q, 64)), dim3(64), 0, (hipStream_t)stream, a);
    return check_launch("decoder_head");
}

// Whole IMPFusion forward from ONE call: 1 + iters * (2 + V) launches, nothing else on the host
extern "C" int dpft_decoder_forward_f32(const dpft_decoder_fwd* d, dpft_stream_t stream) {
    DPFT_REQUIRE(d && d->views && d->pyr && d->query0 && d->pos && d->center0 && d->work, "decoder_forward: null argument");
    const int B = d->B, Q = d->Q, V = d->V;
    DPFT_REQUIRE(B > 0 && Q > 0 && V >= 1 && V <= 4 && d->iters >= 1 && d->iters <= 8, "decoder_forward: bad sizes");
    const size_t nq = (size_t)B * Q;
    float* w = d->work;
    float* qbuf[2] = {w, w + nq * DC};
    float* y1 = w + 2 * nq * DC;
    float* y3 = y1 + (size_t)V * nq * DC;
    float* refs[2] = {y3 + (size_t)V * nq * DC, y3 + (size_t)V * nq * DC + (size_t)V * nq * 2};
    float* cbuf[2] = {refs[1] + (size_t)V * nq * 2, refs[1] + (size_t)V * nq * 2 + nq * 3};
    dpft_decoder_head h;
    memset(&h, 0, sizeof(h));
    for (int v = 0; v < V; ++v) {
        h.T[v] = d->T[v]; h.P[v] = d->P[v]; h.shape[v] = d->shape[v]; h.p_rows[v] = d->p_rows[v]; h.has_t[v] = d->has_t[v];
    }
    h.num_classes = d->num_classes;
    // reference points of the initial (querent) centers
    h.y3 = nullptr; h.prev_center = d->center0; h.refs = refs[0];
    RC(dpft_decoder_head_fwd_f32(&h, B, Q, V, stream));
    const float* query = d->query0;
    const float* center = d->center0;
    int cur = 0;
    for (int it = 0; it < d->iters; ++it) {
        SelfAttnArgs sa;
        for (int v = 0; v < V; ++v) fill_view(sa.w[v], d->views + it * V + v);
        sa.query = query; sa.pos = d->pos; sa.y1 = y1; sa.B = B; sa.Q = Q; sa.V = V;
        sa.qstride = (it == 0) ? 0 : (long)Q * DC;
        const size_t lds = ((size_t)Q * 32 + 48 * 16 + 48 + QT * DC) * sizeof(float);
        DPFT_REQUIRE(lds <= 160 * 1024, "decoder_forward: %d queries do not fit the LDS", Q);
        static bool configured = false;
        if (!configured) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_selfattn_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            configured = true;
        }
        hipLaunchKernelGGL(decoder_selfattn_kernel, dim3(cdiv(Q, QT), V, B), dim3(256), lds, (hipStream_t)stream, sa);
        RC(check_launch("decoder_selfattn"));
        for (int v = 0; v < V; ++v)
            RC(dpft_decoder_xattn_ffn_fwd_f32(d->pyr + v, d->views + it * V + v, y1 + (size_t)v * nq * DC, d->pos,
                                              refs[cur] + (size_t)v * nq * 2, y3 + (size_t)v * nq * DC, B, Q,
                                              d->n_points[v], stream));
        const bool last = it == d->iters - 1;
        h.y3 = y3; h.red_w = d->red_w[it];
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 3; ++k) h.head_w[i][k] = d->head_w[it][i][k];
        h.prev_center = center;
        h.query_out = qbuf[it & 1];
        h.center = last ? d->center : cbuf[it & 1];
        h.size = d->size; h.angle = d->angle; h.cls = d->cls;
        h.refs = last ? nullptr : refs[cur ^ 1];
        RC(dpft_decoder_head_fwd_f32(&h, B, Q, V, stream));
        query = qbuf[it & 1];
        center = h.center;
        cur ^= 1;
    }
    return DPFT_OK;
}

extern "C" int64_t dpft_decoder_work_floats(int32_t B, int32_t Q, int32_t V) {
    const int64_t nq = (int64_t)B * Q;
    return 2 * nq * DC + 2 * (int64_t)V * nq * DC + 2 * (int64_t)V * nq * 2 + 2 * nq * 3 + 64;
}
