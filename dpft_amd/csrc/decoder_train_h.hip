// Training path of the decoder's view reduction + detection head + reference points, one wave per (b, query):
//   x      = reduction_layer(cat_v y3[v])                 (48 -> 16, no bias)   mpfusion.py:416-470, :472-514
//   out_g  = act_g(W6_g relu(W3_g relu(W0_g x)))          g = center, size, angle, class;  heads/detection.py:252-275
//   center = out_center + previous center
//   ref[v] = clip(project(P_v, spherical(T_v center))) / (W, H)   for the NEXT layer     mpfusion.py:617-696
// Forward + hand-written backward; the backward recomputes the row's forward and writes the factors of the weight
// gradients into `rows` (B*Q, HR_FLOATS) for the host framework's batched GEMMs (see train_fused.py), like
// decoder_train_x.hip.  The eager version of this block is ~120 forward + ~240 backward launches per layer.
#include "common.h"
#include "decoder_pack.h"

namespace dpft {

constexpr int HR_DX = 0;        // [16] d x                       x y3cat -> reduction_layer.weight
constexpr int HR_Y3C = 16;      // [64] y3 in reduction order (k * V + v), zero padded
constexpr int HR_D1 = 80;       // [4][16] d pre-activation of layer .0     x x  -> W0
constexpr int HR_D2 = 144;      // [4][16] d pre-activation of layer .3     x h1 -> W3
constexpr int HR_DO = 208;      // [4][16] d output of layer .6 (rows >= out_features are 0)  x h2 -> W6
constexpr int HR_H1 = 272;      // [4][16]
constexpr int HR_H2 = 336;      // [4][16]
constexpr int HR_X = 400;       // [16]
constexpr int HR_FLOATS = 416;

struct HdArgs {
    const float* y3;            // (V,B,Q,16); null = reference points of prev_center only
    const float* ph;            // packed head blob (decoder_pack.h)
    const float* red_w;         // raw reduction_layer.weight (16, 16*V)
    const float* hw[4][3];      // raw head weights
    const float* prev_center;   // (B,Q,3)
    const float* T[4];
    const float* Pm[4];
    const int64_t* shape[4];
    int sstride;                 // int64 elements between shape rows
    int prow[4], flag[4];
    float *x, *center, *size, *angle, *cls, *refs;                   // forward outputs (refs may be null)
    const float *dx, *dcenter, *dsize, *dangle, *dcls, *drefs;      // backward inputs (any may be null)
    float *dy3, *dcenter_prev, *rows;                                // backward outputs
    int B, Q, V, ncls;
};

constexpr float kRad2Deg = 57.29577951308232f;

struct RefPoint {
    float u, v;
};
// forward of one view's reference point; when GRAD, also the gradient of (u, v) w.r.t. the center
template <bool GRAD>
__device__ __forceinline__ RefPoint ref_point(float cx, float cy, float cz, int flag, const float* T, const float* Pm,
                                              float Hs, float Ws, float du, float dv, float* dc) {
    float s0 = cx, s1 = cy, s2 = cz, tx = 0.f, ty = 0.f, tz = 0.f, r = 0.f;
    if (flag) {
        tx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];
        ty = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
        tz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
        r = sqrtf(tx * tx + ty * ty + tz * tz);
        s0 = r;
        s1 = atan2f(ty, tx) * kRad2Deg;
        s2 = asinf(r != 0.f ? tz / r : 0.f) * kRad2Deg;
    }
    const float u_ = Pm[0] * s0 + Pm[1] * s1 + Pm[2] * s2 + Pm[3];
    const float v_ = Pm[4] * s0 + Pm[5] * s1 + Pm[6] * s2 + Pm[7];
    const float w = Pm[8] * s0 + Pm[9] * s1 + Pm[10] * s2 + Pm[11];
    const float u1 = w != 0.f ? u_ / w : u_, v1 = w != 0.f ? v_ / w : v_;
    const float un = u1 / Ws, vn = v1 / Hs;
    RefPoint out = {fminf(fmaxf(un, 0.f), 1.f), fminf(fmaxf(vn, 0.f), 1.f)};
    if (GRAD) {
        const float du1 = (un >= 0.f && un <= 1.f) ? du / Ws : 0.f;
        const float dv1 = (vn >= 0.f && vn <= 1.f) ? dv / Hs : 0.f;
        float du_ = du1, dv_ = dv1, dw = 0.f;
        if (w != 0.f) {
            du_ = du1 / w;
            dv_ = dv1 / w;
            dw = -(du1 * u_ + dv1 * v_) / (w * w);
        }
        const float ds0 = Pm[0] * du_ + Pm[4] * dv_ + Pm[8] * dw;
        const float ds1 = Pm[1] * du_ + Pm[5] * dv_ + Pm[9] * dw;
        const float ds2 = Pm[2] * du_ + Pm[6] * dv_ + Pm[10] * dw;
        if (flag) {
            float dtx = 0.f, dty = 0.f, dtz = 0.f;
            if (r > 0.f) {
                const float ir = 1.f / r;
                dtx = ds0 * tx * ir; dty = ds0 * ty * ir; dtz = ds0 * tz * ir;
                const float c = tz * ir;
                const float ga = ds2 * kRad2Deg / sqrtf(fmaxf(1.f - c * c, 1e-30f));
                const float ir3 = ir * ir * ir;
                dtx += ga * (-tz * tx * ir3);
                dty += ga * (-tz * ty * ir3);
                dtz += ga * (ir - tz * tz * ir3);
            }
            const float rho2 = tx * tx + ty * ty;
            if (rho2 > 0.f) {
                const float gp = ds1 * kRad2Deg / rho2;
                dtx += gp * (-ty);
                dty += gp * tx;
            }
            dc[0] = T[0] * dtx + T[4] * dty + T[8] * dtz;
            dc[1] = T[1] * dtx + T[5] * dty + T[9] * dtz;
            dc[2] = T[2] * dtx + T[6] * dty + T[10] * dtz;
        } else {
            dc[0] = ds0; dc[1] = ds1; dc[2] = ds2;
        }
    }
    return out;
}

struct HdFwd {
    float x;          // x[c], c = lane & 15
    float h1, h2, t;  // branch g = lane >> 4, row o = lane & 15: hidden activations and the last layer's output
};

// LDS scratch of one wave: y3s[4][16] | hx[16] | hh1[4][16] | hh2[4][16]
__device__ __forceinline__ void hd_forward_row(const HdArgs& a, int bq, int lane, float* y3s, float* hx, float* hh1,
                                               float* hh2, HdFwd& f) {
    const int V = a.V;
    const int c = lane & 15;
    const size_t nq = (size_t)a.B * a.Q;
    y3s[lane] = lane < V * DC ? a.y3[((size_t)(lane >> 4) * nq + bq) * DC + c] : 0.f;
    __builtin_amdgcn_wave_barrier();
    const float* ph = a.ph;
    float x = 0.f;
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int k = 0; k < DC; ++k) x = fmaf(ph[PH_RED_WT + (v * DC + k) * DC + c], y3s[v * DC + k], x);
    f.x = x;
    if (lane < 16) hx[lane] = x;
    __builtin_amdgcn_wave_barrier();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (0 * DC + k) * 64 + lane], hx[k], t);
    f.h1 = fmaxf(t, 0.f);
    hh1[lane] = f.h1;
    __builtin_amdgcn_wave_barrier();
    const int g = lane >> 4;
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (1 * DC + k) * 64 + lane], hh1[g * DC + k], t);
    f.h2 = fmaxf(t, 0.f);
    hh2[lane] = f.h2;
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(ph[PH_W + (2 * DC + k) * 64 + lane], hh2[g * DC + k], t);
    f.t = t;
}

__global__ __launch_bounds__(256) void hd_train_fwd_kernel(HdArgs a) {
    __shared__ float sm[4][64 + 16 + 64 + 64 + 4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wv;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q;
    float* y3s = sm[wv];
    float* hx = y3s + 64;
    float* hh1 = hx + 16;
    float* hh2 = hh1 + 64;
    float* cs = hh2 + 64;      // [3] new center
    if (a.y3 != nullptr) {
        HdFwd f;
        hd_forward_row(a, bq, lane, y3s, hx, hh1, hh2, f);
        const int g = lane >> 4, o = lane & 15;
        if (lane < 16) a.x[(size_t)bq * DC + lane] = f.x;
        const int nout = g == 0 ? 3 : (g == 1 ? 3 : (g == 2 ? 2 : a.ncls));
        if (o < nout) {
            if (g == 0) {
                const float cen = f.t + a.prev_center[bq * 3 + o];
                a.center[bq * 3 + o] = cen;
                cs[o] = cen;
            } else if (g == 1) a.size[bq * 3 + o] = fmaxf(f.t, 0.f);
            else if (g == 2) a.angle[bq * 2 + o] = tanhf(f.t);
            else a.cls[bq * a.ncls + o] = f.t;
        }
    } else if (lane < 3) {
        cs[lane] = a.prev_center[bq * 3 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    if (a.refs != nullptr && lane < a.V) {
        const int v = lane;
        const RefPoint rp = ref_point<false>(cs[0], cs[1], cs[2], a.flag[v], a.T[v] ? a.T[v] + (size_t)b * 16 : nullptr,
                                             a.Pm[v] + (size_t)b * a.prow[v] * 4, (float)a.shape[v][b * a.sstride + 0],
                                             (float)a.shape[v][b * a.sstride + 1], 0.f, 0.f, nullptr);
        float* r = a.refs + (((size_t)v * a.B * a.Q) + bq) * 2;
        r[0] = rp.u;
        r[1] = rp.v;
    }
}

__global__ __launch_bounds__(256) void hd_train_bwd_kernel(HdArgs a) {
    __shared__ float sm[4][64 + 16 + 64 + 64 + 4 + 12 + 64 + 64 + 64 + 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bq = blockIdx.x * 4 + wv;
    if (bq >= a.B * a.Q) return;
    const int b = bq / a.Q;
    float* y3s = sm[wv];
    float* hx = y3s + 64;
    float* hh1 = hx + 16;
    float* hh2 = hh1 + 64;
    float* cs = hh2 + 64;       // [3] new center (+1 pad)
    float* dcv = cs + 4;        // [4][3] d center through each view's reference point
    float* dO = dcv + 12;       // [4][16]
    float* d2 = dO + 64;        // [4][16]
    float* d1 = d2 + 64;        // [4][16]
    float* dxs = d1 + 64;       // [16]
    HdFwd f;
    hd_forward_row(a, bq, lane, y3s, hx, hh1, hh2, f);
    const int g = lane >> 4, o = lane & 15, V = a.V;
    const size_t nq = (size_t)a.B * a.Q;
    if (lane < 3) cs[lane] = f.t + a.prev_center[bq * 3 + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < 12) dcv[lane] = 0.f;
    __builtin_amdgcn_wave_barrier();
    if (a.drefs != nullptr && lane < V) {
        const int v = lane;
        const float* dr = a.drefs + ((size_t)v * nq + bq) * 2;
        float dc[3];
        ref_point<true>(cs[0], cs[1], cs[2], a.flag[v], a.T[v] ? a.T[v] + (size_t)b * 16 : nullptr,
                        a.Pm[v] + (size_t)b * a.prow[v] * 4, (float)a.shape[v][b * a.sstride + 0], (float)a.shape[v][b * a.sstride + 1],
                        dr[0], dr[1], dc);
        dcv[v * 3 + 0] = dc[0]; dcv[v * 3 + 1] = dc[1]; dcv[v * 3 + 2] = dc[2];
    }
    __builtin_amdgcn_wave_barrier();
    // gradient of the last layers' outputs
    float dout = 0.f;
    if (g == 0 && o < 3) {
        dout = (a.dcenter ? a.dcenter[bq * 3 + o] : 0.f) + dcv[o] + dcv[3 + o] + dcv[6 + o] + dcv[9 + o];
        a.dcenter_prev[bq * 3 + o] = dout;      // center = head output + previous center
    } else if (g == 1 && o < 3) {
        dout = a.dsize && f.t > 0.f ? a.dsize[bq * 3 + o] : 0.f;
    } else if (g == 2 && o < 2) {
        const float th = tanhf(f.t);
        dout = a.dangle ? a.dangle[bq * 2 + o] * (1.f - th * th) : 0.f;
    } else if (g == 3 && o < a.ncls) {
        dout = a.dcls ? a.dcls[bq * a.ncls + o] : 0.f;
    }
    dO[lane] = dout;
    __builtin_amdgcn_wave_barrier();
    const int nout = g == 0 ? 3 : (g == 1 ? 3 : (g == 2 ? 2 : a.ncls));
    // layer .6 backward: d h2[g][k] = sum_o dout[g][o] W6_g[o][k]
    float t = 0.f;
    for (int oo = 0; oo < nout; ++oo) t = fmaf(dO[g * DC + oo], a.hw[g][2][oo * DC + o], t);
    const float d2v = f.h2 > 0.f ? t : 0.f;
    d2[lane] = d2v;
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll 4
    for (int oo = 0; oo < DC; ++oo) t = fmaf(d2[g * DC + oo], a.hw[g][1][oo * DC + o], t);
    const float d1v = f.h1 > 0.f ? t : 0.f;
    d1[lane] = d1v;
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
#pragma unroll 4
    for (int oo = 0; oo < DC; ++oo) t = fmaf(d1[g * DC + oo], a.hw[g][0][oo * DC + o], t);
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    const float dxv = t + (a.dx ? a.dx[(size_t)bq * DC + o] : 0.f);      // every lane: d x[lane & 15]
    if (lane < 16) dxs[lane] = dxv;
    __builtin_amdgcn_wave_barrier();
    // reduction backward: d y3[v][k] = sum_o dx[o] W[o][k * V + v]
    if (lane < V * DC) {
        const int v = lane >> 4, k = lane & 15;
        float s = 0.f;
#pragma unroll 4
        for (int oo = 0; oo < DC; ++oo) s = fmaf(dxs[oo], a.red_w[oo * DC * V + k * V + v], s);
        a.dy3[((size_t)v * nq + bq) * DC + k] = s;
    }
    // per-row factors of the parameter gradients
    float* row = a.rows + (size_t)bq * HR_FLOATS;
    if (lane < 16) {
        row[HR_DX + lane] = dxv;
        row[HR_X + lane] = f.x;
    }
    {
        // y3 in reduction order: column k * V + v  <-  y3s[v][k]
        const int k = lane / V, v = lane - k * V;
        row[HR_Y3C + lane] = lane < V * DC ? y3s[v * DC + k] : 0.f;
    }
    row[HR_D1 + lane] = d1v;
    row[HR_D2 + lane] = d2v;
    row[HR_DO + lane] = dout;
    row[HR_H1 + lane] = f.h1;
    row[HR_H2 + lane] = f.h2;
}

}  // namespace dpft

using namespace dpft;

static int hd_fill(HdArgs& a, const dpft_head_train* h, int B, int Q, int V, bool need_y3) {
    DPFT_REQUIRE(h && B > 0 && Q > 0 && V >= 1 && V <= 4, "head_train: bad arguments");
    DPFT_REQUIRE(h->prev_center, "head_train: prev_center is null");
    DPFT_REQUIRE(h->num_classes >= 1 && h->num_classes <= 16, "head_train: num_classes must be in [1,16]");
    memset(&a, 0, sizeof(a));
    a.y3 = h->y3; a.ph = h->packed; a.red_w = h->red_w; a.prev_center = h->prev_center;
    if (need_y3 || h->y3) {
        DPFT_REQUIRE(h->y3 && h->packed && h->red_w, "head_train: y3 / packed weights / reduction weight missing");
        for (int i = 0; i < 4; ++i)
            for (int k = 0; k < 3; ++k) {
                DPFT_REQUIRE(h->head_w[i][k], "head_train: head weight %d.%d is null", i, k);
                a.hw[i][k] = h->head_w[i][k];
            }
    }
    for (int v = 0; v < V; ++v) {
        a.T[v] = h->T[v]; a.Pm[v] = h->P[v]; a.shape[v] = h->shape[v]; a.prow[v] = h->p_rows[v]; a.flag[v] = h->has_t[v];
    }
    a.B = B; a.Q = Q; a.V = V; a.ncls = h->num_classes;
    a.sstride = h->shape_stride > 0 ? h->shape_stride : 2;
    return DPFT_OK;
}

static int hd_check_proj(const HdArgs& a, int V) {
    for (int v = 0; v < V; ++v)
        DPFT_REQUIRE(a.Pm[v] && a.shape[v] && (a.T[v] || !a.flag[v]) && a.prow[v] >= 3,
                     "head_train: projection inputs of view %d missing", v);
    return DPFT_OK;
}

extern "C" int64_t dpft_head_train_row_floats(void) { return HR_FLOATS; }

extern "C" int dpft_head_train_fwd_f32(const dpft_head_train* h, int32_t B, int32_t Q, int32_t V, dpft_stream_t stream) {
    HdArgs a;
    int rc = hd_fill(a, h, B, Q, V, false);
    if (rc) return rc;
    if (h->y3) DPFT_REQUIRE(h->x && h->center && h->size && h->angle && h->cls, "head_train_fwd: null output");
    a.x = h->x; a.center = h->center; a.size = h->size; a.angle = h->angle; a.cls = h->cls; a.refs = h->refs;
    if (a.refs) {
        rc = hd_check_proj(a, V);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(hd_train_fwd_kernel, dim3(cdiv((int64_t)B * Q, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("head_train_fwd");
}

extern "C" int dpft_head_train_bwd_f32(const dpft_head_train* h, int32_t B, int32_t Q, int32_t V, dpft_stream_t stream) {
    HdArgs a;
    int rc = hd_fill(a, h, B, Q, V, true);
    if (rc) return rc;
    DPFT_REQUIRE(h->dy3 && h->dcenter_prev && h->rows, "head_train_bwd: null output");
    a.dx = h->dx; a.dcenter = h->dcenter; a.dsize = h->dsize; a.dangle = h->dangle; a.dcls = h->dcls; a.drefs = h->drefs;
    a.dy3 = h->dy3; a.dcenter_prev = h->dcenter_prev; a.rows = h->rows;
    if (a.drefs) {
        rc = hd_check_proj(a, V);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(hd_train_bwd_kernel, dim3(cdiv((int64_t)B * Q, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("head_train_bwd");
}
