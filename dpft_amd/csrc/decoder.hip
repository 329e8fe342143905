// Fused inference path of the iterative fusion decoder (eval mode, no autograd) for d_model = 16, 8 heads
// (head_dim 2), 5 levels x 4 points, Mish FFN, LayerNorm, 'linear' view reduction, 3-layer linear heads --
// i.e. IMPFusion / MPFusion / MLFusion / LinearDetectionHead of the reference at config/kradar*.json:
//   src/dprt/models/fusers/mpfusion.py:122-148 (self attention), :150-208 + layers/ms_deform_attn.py:138-217
//   (deformable cross attention), :210-229 (FFN), :416-470,:472-514 (view reduction), :617-696 (reference
//   points), :698-745 (iteration), src/dprt/models/heads/detection.py:252-275 (head).
// The eager decoder is ~700 launches of B*400x16-sized ops per forward (launch-bound: ~5 ms even when
// replayed from a hipGraph); here one iteration is 3 kernels (round 2; the round-1 pair of kernels took 22 + 41 us
// per iteration, bound by redundant per-lane work and by re-reading 30 KB of weights per query row):
//   K1 decoder_scores_kernel  : block = (50-query chunk, head, view x batch).  A head's K/V are 2+2 of the 48 in_proj
//                               rows, so every block projects only ITS head (no redundancy across the 8 heads);
//                               thread = (query pair, key slice), exact two-pass softmax in the exp2 domain
//                               (1/sqrt(d) * log2(e) folded into the q rows), slice partials merged through LDS.
//                               Writes the pre-out_proj attention output (V,B,Q,16).  Iteration 0 runs it for ONE
//                               batch element: its input (learned query + embedding) does not depend on b.
//   K2 decoder_xattn_kernel   : block = R query rows of ONE view, one wave per row, the view's packed weights
//                               (40 KB) staged once per block in LDS.  Per row: out_proj + LN1 (epilogue of K1),
//                               offsets/logits GEMV whose packed column order delivers every lane exactly the
//                               three (head, level, point) samples it will process, softmax across lanes, bilinear
//                               weights / corner addresses computed ONCE per sample and handed to the 4-lane pixel
//                               groups through a 1.75 KB per-wave LDS scratch, dwordx4 gathers (4 lanes = one 64-B
//                               NHWC pixel), sample-then-project, output_proj + LN2, Mish FFN + LN3.
//   K3 decoder_reduce_head_kernel : wave = (b, q): 48->16 view reduction, the 4 head MLPs, center += previous,
//                               and the NEXT iteration's reference points of all views.
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

// ---------------------------------------------------------------------------------------------------------
// packed INFERENCE blob of one MLFusion (per iteration, view), made by pack_infer_kernel
// ---------------------------------------------------------------------------------------------------------
constexpr int NSLOT = 160;                          // 8 heads x 20 (level, point) slots, in the lanes' slot order
constexpr int PI_SA_IN = 0;                         // [8 heads][6 rows: q0 q1 k0 k1 v0 v1][16]; q rows pre-scaled
constexpr int PI_SA_INB = PI_SA_IN + 768;           // [8][6]
constexpr int PI_K2 = PI_SA_INB + 48;               // LDS image of decoder_xattn_kernel starts here (16-B aligned)
constexpr int K2_OFF = 0;                           // [17][NSLOT][2] sampling_offsets^T (row 16 = bias), slot order
constexpr int K2_LOG = K2_OFF + 17 * NSLOT * 2;     // [17][NSLOT]    attention_weights^T (row 16 = bias)
constexpr int K2_VALW = K2_LOG + 17 * NSLOT;        // value_proj.weight (16,16) as is
constexpr int K2_VALB = K2_VALW + 256;
constexpr int K2_OUTPT = K2_VALB + 16;              // output_proj.weight^T [k][c]
constexpr int K2_OUTPB = K2_OUTPT + 256;
constexpr int K2_N2W = K2_OUTPB + 16;
constexpr int K2_N2B = K2_N2W + 16;
constexpr int K2_F1T = K2_N2B + 16;                 // ffn1.weight^T [k 16][j 32]
constexpr int K2_F1B = K2_F1T + 512;
constexpr int K2_F2T = K2_F1B + 32;                 // ffn2.weight^T [k 32][c 16]
constexpr int K2_F2B = K2_F2T + 512;
constexpr int K2_N3W = K2_F2B + 16;
constexpr int K2_N3B = K2_N3W + 16;
constexpr int K2_SAOT = K2_N3B + 16;                // self_attn.out_proj.weight^T [k][c]
constexpr int K2_SAOB = K2_SAOT + 256;
constexpr int K2_N1W = K2_SAOB + 16;
constexpr int K2_N1B = K2_N1W + 16;
constexpr int K2_FLOATS = K2_N1B + 16;              // 10144 floats = 40 576 B
constexpr int PI_FLOATS = PI_K2 + K2_FLOATS;
static_assert(K2_FLOATS % 4 == 0 && PI_K2 % 4 == 0 && PI_FLOATS % 4 == 0, "float4 staging");

// slot (s, lane) -> head m, sample n of the head.  In gather round t = 4 s + (lane >> 4) the 4-lane pixel group
// g = lane >> 2 ... of the CONSUMER reads the sample the PRODUCER lane (t & 3) * 16 + g computed, so that group g always
// serves head g & 7 (its accumulator never changes head) and the two groups g, g + 8 split a head's 20 samples.
__device__ __forceinline__ void slot_decode(int s, int lane, int& m, int& n) {
    m = lane & 7;
    n = 2 * (4 * s + (lane >> 4)) + ((lane >> 3) & 1);
}

__global__ void pack_infer_kernel(dpft_decoder_view s, int L, int P, float* __restrict__ d) {
    const int LP = L * P, n_off = DM * LP * 2;
    const float qscale = 0.70710678118654752f * 1.4426950408889634f;      // 1/sqrt(head_dim) * log2(e)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PI_FLOATS; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < PI_SA_INB) {
            const int h = i / 96, r = (i / 16) % 6, c = i & 15;
            const int row = (r >> 1) * 16 + 2 * h + (r & 1);
            v = s.in_proj_w[row * 16 + c] * (r < 2 ? qscale : 1.f);
        } else if (i < PI_K2) {
            const int j = i - PI_SA_INB, h = j / 6, r = j % 6;
            v = s.in_proj_b[(r >> 1) * 16 + 2 * h + (r & 1)] * (r < 2 ? qscale : 1.f);
        } else {
            const int k = i - PI_K2;
            int r;
            if (k < K2_LOG) {
                const int c = k / (NSLOT * 2), slot = (k / 2) % NSLOT, xy = k & 1;
                int m, n;
                slot_decode(slot >> 6, slot & 63, m, n);
                if (n < LP) { const int o = (m * LP + n) * 2 + xy; v = c < 16 ? s.off_w[o * 16 + c] : s.off_b[o]; }
            } else if (k < K2_VALW) {
                r = k - K2_LOG;
                const int c = r / NSLOT, slot = r % NSLOT;
                int m, n;
                slot_decode(slot >> 6, slot & 63, m, n);
                if (n < LP) { const int o = m * LP + n; v = c < 16 ? s.att_w[o * 16 + c] : s.att_b[o]; }
            } else if (k < K2_VALB) v = s.val_w[k - K2_VALW];
            else if (k < K2_OUTPT) v = s.val_b[k - K2_VALB];
            else if (k < K2_OUTPB) { r = k - K2_OUTPT; v = s.outp_w[(r & 15) * 16 + (r >> 4)]; }
            else if (k < K2_N2W) v = s.outp_b[k - K2_OUTPB];
            else if (k < K2_N2B) v = s.norm2_w[k - K2_N2W];
            else if (k < K2_F1T) v = s.norm2_b[k - K2_N2B];
            else if (k < K2_F1B) { r = k - K2_F1T; v = s.ffn1_w[(r & 31) * 16 + (r >> 5)]; }
            else if (k < K2_F2T) v = s.ffn1_b[k - K2_F1B];
            else if (k < K2_F2B) { r = k - K2_F2T; v = s.ffn2_w[(r & 15) * 32 + (r >> 4)]; }
            else if (k < K2_N3W) v = s.ffn2_b[k - K2_F2B];
            else if (k < K2_N3B) v = s.norm3_w[k - K2_N3W];
            else if (k < K2_SAOT) v = s.norm3_b[k - K2_N3B];
            else if (k < K2_SAOB) { r = k - K2_SAOT; v = s.out_proj_w[(r & 15) * 16 + (r >> 4)]; }
            else if (k < K2_N1W) v = s.out_proj_b[k - K2_SAOB];
            else if (k < K2_N1B) v = s.norm1_w[k - K2_N1W];
            else v = s.norm1_b[k - K2_N1B];
        }
        d[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1: attention scores of one head
// ---------------------------------------------------------------------------------------------------------
constexpr int QC = 50;      // queries per block (25 pairs); 2 * QC must be a multiple of 4 (LDS alignment of Pt)
constexpr int NS = 10;      // key slices; 25 pairs x 10 slices = 250 of the 256 threads

struct ScoreArgs {
    const float* pi[4];   // packed inference blobs of this iteration
    const float* query;   // (B,Q,16), or (Q,16) when qstride == 0
    const float* pos;     // (Q,16)
    float* attn;          // (V,Bsa,Q,16) attention output before out_proj
    int Bsa, Q, V;
    long qstride;
};

__global__ __launch_bounds__(256) void decoder_scores_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Q = a.Q, SL = (Q + NS - 1) / NS;
    f32x4* KV = reinterpret_cast<f32x4*>(sm);                     // [Q + NS] (k0,k1,v0,v1); one pad entry per slice
    float* Ws = sm + 4 * (Q + NS);                                 // 96 weights + 6 biases (+2 pad)
    f32x2* Qs = reinterpret_cast<f32x2*>(Ws + 104);                // [QC]
    f32x4* Pt = reinterpret_cast<f32x4*>(Ws + 104 + 2 * QC);       // [QC][NS] (max, den, o0, o1); 16-byte aligned
    const int tid = threadIdx.x;
    const int h = blockIdx.y, view = blockIdx.z / a.Bsa, b = blockIdx.z - view * a.Bsa, q0 = blockIdx.x * QC;
    const float* pi = a.pi[view];
    const float* xb = a.query + (size_t)b * a.qstride;
    if (tid < 96) Ws[tid] = pi[PI_SA_IN + h * 96 + tid];
    else if (tid < 102) Ws[tid] = pi[PI_SA_INB + h * 6 + tid - 96];
    __syncthreads();
    for (int i = tid; i < Q + QC; i += 256) {
        const bool isq = i >= Q;
        const int k = isq ? min(q0 + i - Q, Q - 1) : i;
        f32x4 x[4], xp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x[c] = *reinterpret_cast<const f32x4*>(xb + (size_t)k * DC + 4 * c);
            xp[c] = x[c] + *reinterpret_cast<const f32x4*>(a.pos + (size_t)k * DC + 4 * c);
        }
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // query item: rows 0,1 of (x + pos); key item: rows 2,3 of (x + pos) and rows 4,5 of x
            const int row = isq ? (j & 1) : 2 + j;
            float acc = Ws[96 + row];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(Ws + row * 16 + 4 * c);
                const f32x4 v = (isq || j < 2) ? xp[c] : x[c];
                acc = fmaf(w[0], v[0], acc); acc = fmaf(w[1], v[1], acc);
                acc = fmaf(w[2], v[2], acc); acc = fmaf(w[3], v[3], acc);
            }
            r[j] = acc;
        }
        if (isq) Qs[i - Q] = f32x2{r[0], r[1]};
        else KV[k + k / SL] = f32x4{r[0], r[1], r[2], r[3]};
    }
    __syncthreads();
    if (tid < (QC / 2) * NS) {
        const int slice = tid / (QC / 2), pair = tid - slice * (QC / 2);
        const f32x2 qa = Qs[2 * pair], qb = Qs[2 * pair + 1];
        const int k0 = slice * SL, k1 = min(Q, k0 + SL);
        const f32x4* kv = KV + k0 + slice;
        float ma = -INFINITY, mb = -INFINITY;
        for (int k = 0; k < k1 - k0; ++k) {
            const f32x2 kk = *reinterpret_cast<const f32x2*>(kv + k);
            ma = fmaxf(ma, fmaf(qa[1], kk[1], qa[0] * kk[0]));
            mb = fmaxf(mb, fmaf(qb[1], kk[1], qb[0] * kk[0]));
        }
        float da = 0.f, db = 0.f, a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
        for (int k = 0; k < k1 - k0; ++k) {
            const f32x4 e = kv[k];
            const float pa = __builtin_amdgcn_exp2f(fmaf(qa[1], e[1], qa[0] * e[0]) - ma);
            const float pb = __builtin_amdgcn_exp2f(fmaf(qb[1], e[1], qb[0] * e[0]) - mb);
            da += pa; db += pb;
            a0 = fmaf(pa, e[2], a0); a1 = fmaf(pa, e[3], a1);
            b0 = fmaf(pb, e[2], b0); b1 = fmaf(pb, e[3], b1);
        }
        Pt[(2 * pair) * NS + slice] = f32x4{ma, da, a0, a1};
        Pt[(2 * pair + 1) * NS + slice] = f32x4{mb, db, b0, b1};
    }
    __syncthreads();
    if (tid < QC && q0 + tid < Q) {
        float mm = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) mm = fmaxf(mm, Pt[tid * NS + s2][0]);
        float den = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const f32x4 part = Pt[tid * NS + s2];
            const float cf = part[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(part[0] - mm);     // empty slice
            den = fmaf(part[1], cf, den);
            o0 = fmaf(part[2], cf, o0);
            o1 = fmaf(part[3], cf, o1);
        }
        float* dst = a.attn + (((size_t)view * a.Bsa + b) * Q + q0 + tid) * DC + 2 * h;
        *reinterpret_cast<f32x2*>(dst) = f32x2{o0 / den, o1 / den};
    }
}

// ---------------------------------------------------------------------------------------------------------
// K2: deformable cross attention + FFN of R query rows of one view
// ---------------------------------------------------------------------------------------------------------
struct Pyr5 {
    const float* level[DPFT_MAX_LEVELS];
    int H[DPFT_MAX_LEVELS], W[DPFT_MAX_LEVELS];
    int L;
};
struct XattnArgs {
    Pyr5 pyr[4];
    const float* pi[4];         // packed inference blobs
    const float* attn;          // (V,Bsa,Q,16)
    const float* query;         // (B,Q,16) or (Q,16): residual of the self-attention block
    const float* pos;           // (Q,16)
    const float* refs;          // (V,B,Q,2) or NULL: project prev_center here (first iteration)
    const float* prev_center;   // (B,Q,3)
    const float* T[4];          // (B,4,4)
    const float* Pm[4];         // (B,prow,4)
    const int64_t* shape[4];    // (B,2) = H, W
    int prow[4], flag[4], P[4];
    float* y3;                  // (V,B,Q,16)
    int B, Q, V, Bsa;
    long qstride;
};

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

__device__ __forceinline__ float rdlane(float v, int k) {     // k wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

typedef __attribute__((address_space(1))) char gbytes;          // explicit global address space: addresses rebuilt from
typedef __attribute__((address_space(1))) f32x4 gf32x4;        // integers would otherwise become flat loads
constexpr int XW_FLOATS = 480;      // per-wave scratch: wq float4[64] | addr uint2[64] | pitch int[64] | vec float[32]

template <int R>
__global__ __launch_bounds__(R * 64, 6) void decoder_xattn_kernel(XattnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ __attribute__((aligned(16))) int lvl_tab[DPFT_MAX_LEVELS][4];      // ptr lo, ptr hi, H, W
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y;
    const float* __restrict__ img = a.pi[view] + PI_K2;
    // stage the view's weights (the row's own loads are issued first so that their latency hides behind this)
    const int bq = blockIdx.x * R + wave;
    const bool live = bq < a.B * a.Q;
    const int bqc = live ? bq : a.B * a.Q - 1;
    const int b = bqc / a.Q, q = bqc - b * a.Q;
    const int c = lane & 15;
    const Pyr5& pyr = a.pyr[view];
    const int L = pyr.L, P = a.P[view], LP = L * P;
    const float av = a.attn[(((size_t)view * a.Bsa + (a.Bsa == 1 ? 0 : b)) * a.Q + q) * DC + c];
    const float xres = a.query[(size_t)b * a.qstride + (size_t)q * DC + c];
    const float posc = a.pos[(size_t)q * DC + c];
    {
        constexpr int NV = K2_FLOATS / 4, PER = (NV + R * 64 - 1) / (R * 64);
        f32x4 stage[PER];                      // all loads in flight before the first LDS store
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * R * 64 < NV) stage[u] = reinterpret_cast<const f32x4*>(img)[tid + u * R * 64];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * R * 64 < NV) reinterpret_cast<f32x4*>(sm)[tid + u * R * 64] = stage[u];
    }
    if (tid < L) {
        const uint64_t p = reinterpret_cast<uint64_t>(pyr.level[tid]);
        lvl_tab[tid][0] = (int)(uint32_t)p; lvl_tab[tid][1] = (int)(uint32_t)(p >> 32);
        lvl_tab[tid][2] = pyr.H[tid]; lvl_tab[tid][3] = pyr.W[tid];
    }
    float rx, ry;
    if (a.refs) {
        const f32x2 r2 = *reinterpret_cast<const f32x2*>(a.refs + ((size_t)view * a.B * a.Q + bqc) * 2);
        rx = r2[0]; ry = r2[1];
    } else {
        const float* pc = a.prev_center + (size_t)bqc * 3;
        reference_point(pc[0], pc[1], pc[2], a.flag[view], a.T[view] ? a.T[view] + (size_t)b * 16 : nullptr,
                        a.Pm[view] + (size_t)b * a.prow[view] * 4, (float)a.shape[view][b * 2 + 0],
                        (float)a.shape[view][b * 2 + 1], rx, ry);
    }
    __syncthreads();
    if (!live) return;
    float* ws = sm + K2_FLOATS + wave * XW_FLOATS;
    f32x4* wq = reinterpret_cast<f32x4*>(ws);
    uint2* adr = reinterpret_cast<uint2*>(ws + 256);
    int* pit = reinterpret_cast<int*>(ws + 384);
    float* vec = ws + 448;
    // ---- self-attention epilogue: out_proj + residual + LayerNorm1 (mpfusion.py:142-148) ----
    float y1c = sm[K2_SAOB + c];
#pragma unroll
    for (int k = 0; k < DC; ++k) y1c = fmaf(sm[K2_SAOT + k * DC + c], rdlane(av, k), y1c);
    y1c = layernorm16(y1c + xres, sm[K2_N1W + c], sm[K2_N1B + c]);
    const float qpc = y1c + posc;
    // ---- sampling offsets + attention logits of this lane's 3 slots (ms_deform_attn.py:177-182) ----
    int idx[3];
    f32x2 off[3];
    float lg[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        idx[s] = min(s * 64 + lane, NSLOT - 1);
        off[s] = *reinterpret_cast<const f32x2*>(sm + K2_OFF + (16 * NSLOT + idx[s]) * 2);
        lg[s] = sm[K2_LOG + 16 * NSLOT + idx[s]];
    }
#pragma unroll 2
    for (int k = 0; k < DC; ++k) {
        const float xk = rdlane(qpc, k);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const f32x2 w = *reinterpret_cast<const f32x2*>(sm + K2_OFF + (k * NSLOT + idx[s]) * 2);
            off[s][0] = fmaf(w[0], xk, off[s][0]);
            off[s][1] = fmaf(w[1], xk, off[s][1]);
            lg[s] = fmaf(sm[K2_LOG + k * NSLOT + idx[s]], xk, lg[s]);
        }
    }
    // ---- softmax over the L*P slots of the head (lanes with equal lane & 7; bits 3..5 + the 3 slots) ----
    int ns[3];
    bool ok[3];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        int m_;
        slot_decode(s, lane, m_, ns[s]);
        ok[s] = ns[s] < LP && (s < 2 || lane < 32);
        if (ok[s]) mx = fmaxf(mx, lg[s]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8)); mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    float aw[3], den = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) { aw[s] = ok[s] ? __expf(lg[s] - mx) : 0.f; den += aw[s]; }
    den += __shfl_xor(den, 8); den += __shfl_xor(den, 16); den += __shfl_xor(den, 32);
    const float inv_den = 1.f / den;
    // ---- sample-then-project: producer lanes write (weights, corner address), 4-lane pixel groups gather ----
    const int g = lane >> 2, j = lane & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ms = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        {
            const int l = ok[s] ? ns[s] / P : 0;
            const int4 lt = *reinterpret_cast<const int4*>(lvl_tab[l]);
            const int H = lt.z, W = lt.w;
            const float a_w = aw[s] * inv_den;
            const float lx = rx + off[s][0] / (float)W, ly = ry + off[s][1] / (float)H;
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = ok[s] && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_lo = (int)hf, w_lo = (int)wf, h_hi = h_lo + 1, w_hi = w_lo + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
            const bool k1 = in && h_lo >= 0 && w_lo >= 0, k2 = in && h_lo >= 0 && w_hi <= W - 1;
            const bool k3 = in && h_hi <= H - 1 && w_lo >= 0, k4 = in && h_hi <= H - 1 && w_hi <= W - 1;
            const int hl = min(max(h_lo, 0), H - 1), hh_ = min(max(h_hi, 0), H - 1);
            const int wl = min(max(w_lo, 0), W - 1), wh_ = min(max(w_hi, 0), W - 1);
            wq[lane] = f32x4{k1 ? a_w * hh * hw : 0.f, k2 ? a_w * hh * lw : 0.f, k3 ? a_w * lh * hw : 0.f,
                             k4 ? a_w * lh * lw : 0.f};
            const uint64_t base = ((uint64_t)(uint32_t)lt.y << 32 | (uint32_t)lt.x)
                                  + ((uint64_t)((int64_t)b * H + hl) * W + wl) * (DC * 4) + (wh_ - wl);   // bit 0: column step
            adr[lane] = uint2{(uint32_t)base, (uint32_t)(base >> 32)};
            pit[lane] = (hh_ - hl) * W * (DC * 4);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int k = 0; k < (s < 2 ? 4 : 2); k += 2) {      // two rounds (8 x 16-byte gathers per lane) in flight
            f32x4 w4[2], v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int src = (k + u) * 16 + g;
                w4[u] = wq[src];
                const uint2 ad = adr[src];
                const int pitch = pit[src];
                const int cstep = (ad.x & 1) ? DC * 4 : 0;
                const gbytes* A = reinterpret_cast<const gbytes*>(((uint64_t)ad.y << 32) | (ad.x & ~1u)) + j * 16;
                v[u][0] = *reinterpret_cast<const gf32x4*>(A);          // global_load_dwordx4 (not flat)
                v[u][1] = *reinterpret_cast<const gf32x4*>(A + cstep);
                v[u][2] = *reinterpret_cast<const gf32x4*>(A + pitch);
                v[u][3] = *reinterpret_cast<const gf32x4*>(A + pitch + cstep);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc += w4[u][0] * v[u][0]; acc += w4[u][1] * v[u][1]; acc += w4[u][2] * v[u][2]; acc += w4[u][3] * v[u][3];
                ms += (w4[u][0] + w4[u][1]) + (w4[u][2] + w4[u][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // value_proj on the sampled features (+ bias * in-bounds mass), head m = g & 7 -> channels 2m, 2m+1
    {
        const int m = g & 7;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + K2_VALW + (2 * m) * DC + 4 * j);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + K2_VALW + (2 * m + 1) * DC + 4 * j);
        float o0 = w0[0] * acc[0] + w0[1] * acc[1] + w0[2] * acc[2] + w0[3] * acc[3];
        float o1 = w1[0] * acc[0] + w1[1] * acc[1] + w1[2] * acc[2] + w1[3] * acc[3];
        o0 += __shfl_xor(o0, 1); o0 += __shfl_xor(o0, 2); o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 1); o1 += __shfl_xor(o1, 2); o1 += __shfl_xor(o1, 32);
        ms += __shfl_xor(ms, 32);
        if (lane < 32 && j == 0) {
            vec[2 * m] = o0 + sm[K2_VALB + 2 * m] * ms;
            vec[2 * m + 1] = o1 + sm[K2_VALB + 2 * m + 1] * ms;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- output_proj + residual + LayerNorm2 (lanes 0..15 = channels; other lanes mirror them) ----
    float vo = sm[K2_OUTPB + c];
#pragma unroll
    for (int k = 0; k < DC; ++k) vo = fmaf(sm[K2_OUTPT + k * DC + c], vec[k], vo);
    const float y2 = layernorm16(vo + y1c, sm[K2_N2W + c], sm[K2_N2B + c]);
    // ---- FFN: 16 -> 32 (Mish) -> 16, residual, LayerNorm3 ----
    const int jf = lane & 31;
    float hsum = sm[K2_F1B + jf];
#pragma unroll
    for (int k = 0; k < DC; ++k) hsum = fmaf(sm[K2_F1T + k * DFF + jf], rdlane(y2, k), hsum);
    const float hval = mishf(hsum);
    float f = sm[K2_F2B + c];
#pragma unroll
    for (int k = 0; k < DFF; ++k) f = fmaf(sm[K2_F2T + k * DC + c], rdlane(hval, k), f);
    const float y3 = layernorm16(f + y2, sm[K2_N3W + c], sm[K2_N3B + c]);
    if (lane < 16) a.y3[((size_t)view * a.B * a.Q + bq) * DC + lane] = y3;
}

// ---------------------------------------------------------------------------------------------------------
// K3: view reduction + heads + next reference points, one wave per (b, q)
// ---------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* y3;            // (V,B,Q,16)
    const float* ph;            // packed head blob
    const float* prev_center;   // (B,Q,3)
    const float* T[4];
    const float* Pm[4];
    const int64_t* shape[4];
    int prow[4], flag[4];
    float* query_out;           // (B,Q,16)
    float *center, *size, *angle, *cls;
    float* refs_out;            // (V,B,Q,2) reference points of the NEW center, or NULL (last iteration)
    int B, Q, V, ncls;
};

__global__ __launch_bounds__(256) void decoder_reduce_head_kernel(HeadArgs a) {
    __shared__ float hs[4][2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wave;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q;
    const int c = lane & 15, v2 = lane >> 4;
    const float* ph = a.ph;
    const float yv = v2 < a.V ? a.y3[((size_t)v2 * a.B * a.Q + bq) * DC + c] : 0.f;
    // view reduction: queries.view(B,N,C*V) is channel-major / view-minor (mpfusion.py:436-438)
    float x = 0.f;
    for (int v = 0; v < a.V; ++v)
#pragma unroll
        for (int k = 0; k < DC; ++k) x = fmaf(ph[PH_RED_WT + (v * DC + k) * DC + c], rdlane(yv, v * 16 + k), x);
    if (lane < 16) a.query_out[(size_t)bq * DC + lane] = x;
    // heads (heads/detection.py:252-275): branch g = lane / 16 (center, size, angle, class), row o = lane % 16
    const int g = lane >> 4, o = lane & 15;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (0 * DC + k) * 64 + lane], rdlane(x, k), t);
    hs[wave][0][lane] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (1 * DC + k) * 64 + lane], hs[wave][0][g * 16 + k], t);
    hs[wave][1][lane] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (2 * DC + k) * 64 + lane], hs[wave][1][g * 16 + k], t);
    const int nout = g == 0 ? 3 : (g == 1 ? 3 : (g == 2 ? 2 : a.ncls));
    float cen = 0.f;
    if (o < nout) {
        if (g == 0) { cen = t + a.prev_center[bq * 3 + o]; a.center[bq * 3 + o] = cen; }
        else if (g == 1) a.size[bq * 3 + o] = fmaxf(t, 0.f);
        else if (g == 2) a.angle[bq * 2 + o] = tanhf(t);
        else a.cls[bq * a.ncls + o] = t;
    }
    if (a.refs_out) {       // reference points of the new center for the next iteration: lane = view
        const float cx = rdlane(cen, 0), cy = rdlane(cen, 1), cz = rdlane(cen, 2);
        if (lane < a.V) {
            float u, vv;
            reference_point(cx, cy, cz, a.flag[lane], a.T[lane] ? a.T[lane] + (size_t)b * 16 : nullptr,
                            a.Pm[lane] + (size_t)b * a.prow[lane] * 4, (float)a.shape[lane][b * 2 + 0],
                            (float)a.shape[lane][b * 2 + 1], u, vv);
            *reinterpret_cast<f32x2*>(a.refs_out + ((size_t)lane * a.B * a.Q + bq) * 2) = f32x2{u, vv};
        }
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

extern "C" int64_t dpft_decoder_packed_infer_floats(void) { return PI_FLOATS; }

extern "C" int dpft_decoder_pack_infer_f32(const dpft_decoder_view* view, int32_t L, int32_t P, float* packed,
                                           dpft_stream_t stream) {
    DPFT_REQUIRE(view && packed, "decoder_pack_infer: null argument");
    DPFT_REQUIRE(L >= 1 && L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && L * P <= 20,
                 "decoder_pack_infer: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", L, P);
    const float* const* f = reinterpret_cast<const float* const*>(view);
    for (size_t i = 0; i < sizeof(dpft_decoder_view) / sizeof(float*); ++i)
        DPFT_REQUIRE(f[i], "decoder_pack_infer: parameter pointer %d is null", (int)i);
    hipLaunchKernelGGL(pack_infer_kernel, dim3(cdiv(PI_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, *view, L, P, packed);
    return check_launch("decoder_pack_infer");
}

constexpr int XR = 7;      // query rows (waves) per decoder_xattn_kernel block: 3 blocks of 54 KB LDS per CU

// Whole IMPFusion forward from ONE call: 3 launches per iteration, nothing else on the host
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
    float* attn = w + 2 * nq * DC;
    float* y3 = attn + (size_t)V * nq * DC;
    float* cbuf[2] = {y3 + (size_t)V * nq * DC, y3 + (size_t)V * nq * DC + nq * 3};
    float* refs = cbuf[1] + nq * 3;
    XattnArgs xa;
    HeadArgs ha;
    memset(&xa, 0, sizeof(xa));
    memset(&ha, 0, sizeof(ha));
    for (int v = 0; v < V; ++v) {
        const dpft_pyramid* pyr = d->pyr + v;
        const int P = d->n_points[v];
        DPFT_REQUIRE(pyr->L >= 1 && pyr->L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && pyr->L * P <= 20,
                     "decoder_forward: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", pyr->L, P);
        xa.pyr[v].L = pyr->L;
        for (int l = 0; l < pyr->L; ++l) {
            DPFT_REQUIRE(pyr->level[l], "decoder_forward: view %d level %d is null", v, l);
            DPFT_REQUIRE((reinterpret_cast<uintptr_t>(pyr->level[l]) & 63) == 0,
                         "decoder_forward: view %d level %d is not 64-byte aligned", v, l);
            xa.pyr[v].level[l] = pyr->level[l]; xa.pyr[v].H[l] = pyr->H[l]; xa.pyr[v].W[l] = pyr->W[l];
        }
        xa.P[v] = P;
        xa.T[v] = ha.T[v] = d->T[v]; xa.Pm[v] = ha.Pm[v] = d->P[v]; xa.shape[v] = ha.shape[v] = d->shape[v];
        xa.prow[v] = ha.prow[v] = d->p_rows[v]; xa.flag[v] = ha.flag[v] = d->has_t[v];
        DPFT_REQUIRE(xa.Pm[v] && xa.shape[v] && (xa.T[v] || !xa.flag[v]) && xa.prow[v] >= 3,
                     "decoder_forward: projection inputs of view %d missing", v);
    }
    xa.attn = attn; xa.pos = d->pos; xa.y3 = y3; xa.B = B; xa.Q = Q; xa.V = V;
    ha.y3 = y3; ha.B = B; ha.Q = Q; ha.V = V; ha.ncls = d->num_classes;
    ha.size = d->size; ha.angle = d->angle; ha.cls = d->cls;
    const float* query = d->query0;
    const float* center = d->center0;
    const size_t lds1 = (4 * (size_t)(Q + NS) + 104 + 2 * QC + 4 * QC * NS) * sizeof(float);
    const size_t lds2 = ((size_t)K2_FLOATS + XR * XW_FLOATS) * sizeof(float);
    DPFT_REQUIRE(lds1 <= 64 * 1024, "decoder_forward: %d queries do not fit the LDS of the score kernel", Q);
    for (int it = 0; it < d->iters; ++it) {
        const bool first = it == 0, last = it == d->iters - 1;
        ScoreArgs sa;
        for (int v = 0; v < 4; ++v)
            sa.pi[v] = xa.pi[v] = v < V ? d->packed_views + (size_t)(it * V + v) * PI_FLOATS : nullptr;
        // iteration 0: query = the learned (Q,16) table for every batch element -> one batch element of scores
        sa.query = query; sa.pos = d->pos; sa.attn = attn; sa.Q = Q; sa.V = V;
        sa.Bsa = first ? 1 : B;
        sa.qstride = first ? 0 : (long)Q * DC;
        hipLaunchKernelGGL(decoder_scores_kernel, dim3(cdiv(Q, QC), DM, V * sa.Bsa), dim3(256), lds1, (hipStream_t)stream, sa);
        RC(check_launch("decoder_scores"));
        xa.query = query; xa.qstride = sa.qstride; xa.Bsa = sa.Bsa;
        xa.refs = first ? nullptr : refs;
        xa.prev_center = center;
        hipLaunchKernelGGL(decoder_xattn_kernel<XR>, dim3(cdiv((int64_t)nq, XR), V), dim3(XR * 64), lds2, (hipStream_t)stream, xa);
        RC(check_launch("decoder_xattn"));
        ha.ph = d->packed_heads + (size_t)it * PH_FLOATS;
        ha.prev_center = center;
        ha.query_out = qbuf[it & 1];
        ha.center = last ? d->center : cbuf[it & 1];
        ha.refs_out = last ? nullptr : refs;
        hipLaunchKernelGGL(decoder_reduce_head_kernel, dim3(cdiv((int64_t)nq, 4)), dim3(256), 0, (hipStream_t)stream, ha);
        RC(check_launch("decoder_reduce_head"));
        query = qbuf[it & 1];
        center = ha.center;
    }
    return DPFT_OK;
}

extern "C" int64_t dpft_decoder_work_floats(int32_t B, int32_t Q, int32_t V) {
    const int64_t nq = (int64_t)B * Q;
    return 2 * nq * DC + 2 * (int64_t)V * nq * DC + 2 * nq * 3 + (int64_t)V * nq * 2 + 64;
}
