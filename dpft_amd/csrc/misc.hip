// GIoU3D of yaw-only boxes for the Hungarian matcher cost (replaces pytorch3d.ops.box3d_overlap as
// used by src/dprt/utils/iou.py:121-210 + src/dprt/utils/bbox.py:77-163).  One thread per
// (sample, prediction, target) pair; BEV rectangle-rectangle clip (Sutherland-Hodgman) x z-overlap,
// evaluated in fp64 (400 x M pairs per sample: cost is irrelevant, robustness is not).
#include "common.h"

namespace dpft {

struct P2 { double x, y; };

__device__ void rect_corners(const float* b, P2* c, double& zlo, double& zhi) {
    const double cx = b[0], cy = b[1], cz = b[2], l = b[3], w = b[4], h = b[5], yaw = b[6];
    const double co = cos(yaw), si = sin(yaw);
    const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
    for (int i = 0; i < 4; ++i) {
        const double x = sx[i] * l / 2, y = sy[i] * w / 2;
        c[i].x = co * x - si * y + cx;
        c[i].y = si * x + co * y + cy;
    }
    zlo = cz - h / 2;
    zhi = cz + h / 2;
}

__device__ double poly_area(const P2* p, int n) {
    double a = 0;
    for (int i = 0; i < n; ++i) {
        const P2 u = p[i], v = p[(i + 1) % n];
        a += u.x * v.y - v.x * u.y;
    }
    return a / 2;
}

// GIoU3D (return value) and IoU3D (*iou_out, may be null) of two (x,y,z,l,w,h,yaw) boxes
__device__ float giou3d_yaw_pair(const float* pa, const float* pb, float* iou_out = nullptr) {
    const double eps = 1e-4;
    if (iou_out) *iou_out = 0.f;
    auto valid = [&](const float* s) {
        const double l = s[3], w = s[4], h = s[5];
        return fmin(fmin(l * w, l * h), w * h) / 2 > eps;
    };
    if (!(valid(pa) && valid(pb))) return -1.f;      // iou.py:159,185-208: evol keeps -1 => giou = -1
    P2 A[4], Bq[4];
    double azl, azh, bzl, bzh;
    rect_corners(pa, A, azl, azh);
    rect_corners(pb, Bq, bzl, bzh);
    // enclosing axis-aligned box over all 16 corners
    double xmin = A[0].x, xmax = A[0].x, ymin = A[0].y, ymax = A[0].y;
    for (int k = 0; k < 4; ++k) {
        xmin = fmin(xmin, fmin(A[k].x, Bq[k].x)); xmax = fmax(xmax, fmax(A[k].x, Bq[k].x));
        ymin = fmin(ymin, fmin(A[k].y, Bq[k].y)); ymax = fmax(ymax, fmax(A[k].y, Bq[k].y));
    }
    const double evol = (xmax - xmin) * (ymax - ymin) * (fmax(azh, bzh) - fmin(azl, bzl));
    // clip A by the half-planes of B (both CCW for positive sizes)
    P2 poly[10], tmp[10];
    int n = 4;
    for (int k = 0; k < 4; ++k) poly[k] = A[k];
    if (poly_area(poly, 4) < 0) { P2 t = poly[1]; poly[1] = poly[3]; poly[3] = t; }
    P2 Q[4];
    for (int k = 0; k < 4; ++k) Q[k] = Bq[k];
    if (poly_area(Q, 4) < 0) { P2 t = Q[1]; Q[1] = Q[3]; Q[3] = t; }
    for (int e = 0; e < 4 && n > 0; ++e) {
        const P2 a = Q[e], bb = Q[(e + 1) & 3];
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const P2 c = poly[k], d = poly[(k + 1) % n];
            const double sc = (bb.x - a.x) * (c.y - a.y) - (bb.y - a.y) * (c.x - a.x);
            const double sd = (bb.x - a.x) * (d.y - a.y) - (bb.y - a.y) * (d.x - a.x);
            if (sc >= 0) tmp[m++] = c;
            if ((sc >= 0) != (sd >= 0)) {
                const double t = sc / (sc - sd);
                tmp[m].x = c.x + t * (d.x - c.x);
                tmp[m].y = c.y + t * (d.y - c.y);
                ++m;
            }
        }
        n = m;
        for (int k = 0; k < n; ++k) poly[k] = tmp[k];
    }
    const double inter_a = n >= 3 ? fabs(poly_area(poly, n)) : 0.0;
    const double vol = inter_a * fmax(0.0, fmin(azh, bzh) - fmax(azl, bzl));
    const double v1 = (double)pa[3] * pa[4] * pa[5], v2 = (double)pb[3] * pb[4] * pb[5];
    const double iou = vol > 0 ? vol / (v1 + v2 - vol) : 0.0;
    const double uni = iou != 0 ? vol / iou : 0.0;
    if (iou_out) *iou_out = (float)iou;
    return (float)(evol != 0 ? iou - (evol - uni) / evol : 0.0);
}


__global__ void giou3d_yaw_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ out,
                                  int B, int N, int Mg) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * N * Mg) return;
    const int j = (int)(idx % Mg);
    const int i = (int)((idx / Mg) % N);
    const int b = (int)(idx / ((int64_t)Mg * N));
    out[idx] = giou3d_yaw_pair(pred + ((int64_t)b * N + i) * 7, gt + ((int64_t)b * Mg + j) * 7);
}

// ---- Hungarian matcher cost (src/dprt/training/assigner.py:113-132), one thread per (sample, query, target) ----
struct CostArgs {
    const float *cls, *center, *size, *angle;      // (B,N,C) (B,N,3) (B,N,3) (B,N,2)
    const float* gt_box;                           // (B,Mmax,8) center | size | angle
    const int32_t* gt_id;                          // (B,Mmax) class index
    const int32_t* counts;                         // (B)
    float* cost;                                   // (B,N,Mmax), 0 beyond counts[b]
    float w_class, w_center, w_size, w_angle, w_giou;
    int B, N, Mmax, C;
};
__global__ void match_cost_kernel(CostArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.B * a.N * a.Mmax) return;
    const int j = (int)(idx % a.Mmax);
    const int64_t bn = idx / a.Mmax;
    const int b = (int)(bn / a.N);
    if (j >= a.counts[b]) {
        a.cost[idx] = 0.f;
        return;
    }
    const float* g = a.gt_box + ((int64_t)b * a.Mmax + j) * 8;
    const float* ce = a.center + bn * 3;
    const float* sz = a.size + bn * 3;
    const float* an = a.angle + bn * 2;
    const float pa[7] = {ce[0], ce[1], ce[2], sz[0], sz[1], sz[2], atan2f(an[0], an[1])};
    const float pb[7] = {g[0], g[1], g[2], g[3], g[4], g[5], atan2f(g[6], g[7])};
    const float giou = giou3d_yaw_pair(pa, pb);
    const float l1c = fabsf(ce[0] - g[0]) + fabsf(ce[1] - g[1]) + fabsf(ce[2] - g[2]);
    const float l1s = fabsf(sz[0] - g[3]) + fabsf(sz[1] - g[4]) + fabsf(sz[2] - g[5]);
    const float l1a = fabsf(an[0] - g[6]) + fabsf(an[1] - g[7]);
    float c = a.w_class * (-a.cls[bn * a.C + a.gt_id[b * a.Mmax + j]]);
    c += a.w_center * l1c;
    c += a.w_size * l1s;
    c += a.w_angle * l1a;
    c += a.w_giou * (-giou);
    a.cost[idx] = c;
}

// ---- per-step detection metrics (src/dprt/evaluation/metric.py: mAP3D :16-151, mGIoU3D :154-253) --------------------
// Pair pass: IoU3D / GIoU3D of every (prediction, target) pair of a sample.  Metric pass: one block per sample.
// Both metrics reduce to a handful of per-class counts (see the comments in detection_metric_kernel): the reference's
// precision/recall "interpolation" is a straight line through the FIRST and LAST point of the curve
// (src/dprt/utils/misc.py:43-83), so no sort / scan of the 400 predictions is needed -- only the most confident row, the
// set of matched rows and three counters per class.
struct MetricArgs {
    const float *cls, *center, *size, *angle;      // (B,N,C) (B,N,3) (B,N,3) (B,N,2)
    const float* gt_box;                           // (B,Mmax,8)
    const float* gt_onehot;                        // (B,Mmax,C)
    const int32_t* counts;                         // (B)
    float* pair;                                   // (B,N,Mmax,2): iou, giou
    float* out;                                    // (B,2): mAP, mGIoU of each sample
    float thr;
    int nelem;
    int B, N, Mmax, C;
};
__global__ void metric_pair_kernel(MetricArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)a.B * a.N * a.Mmax) return;
    const int j = (int)(idx % a.Mmax);
    const int64_t bn = idx / a.Mmax;
    const int b = (int)(bn / a.N);
    float iou = 0.f, giou = -1.f;
    if (j < a.counts[b]) {
        const float* g = a.gt_box + ((int64_t)b * a.Mmax + j) * 8;
        const float* ce = a.center + bn * 3;
        const float* sz = a.size + bn * 3;
        const float* an = a.angle + bn * 2;
        const float pa[7] = {ce[0], ce[1], ce[2], sz[0], sz[1], sz[2], atan2f(an[0], an[1])};
        const float pb[7] = {g[0], g[1], g[2], g[3], g[4], g[5], atan2f(g[6], g[7])};
        giou = giou3d_yaw_pair(pa, pb, &iou);
    }
    a.pair[idx * 2 + 0] = iou;
    a.pair[idx * 2 + 1] = giou;
}

constexpr int kMetricMaxN = 1024, kMetricMaxM = 256, kMetricMaxC = 16;

__device__ __forceinline__ int argmax_row(const float* x, int n, int stride) {
    int best = 0;
    for (int i = 1; i < n; ++i)
        if (x[(size_t)i * stride] > x[(size_t)best * stride]) best = i;
    return best;
}

__global__ __launch_bounds__(256) void detection_metric_kernel(MetricArgs a) {
    __shared__ int label[kMetricMaxN];          // argmax class of every prediction
    __shared__ int gt_label[kMetricMaxM];
    __shared__ int best[kMetricMaxM];           // per target column: most confident matching prediction (or -1)
    __shared__ float colmax[kMetricMaxM];       // per target column: max GIoU over the predictions of the class
    __shared__ int present[kMetricMaxC];
    __shared__ float ap[kMetricMaxC], gi[kMetricMaxC];
    __shared__ int s_row0, s_nmask, s_nR, s_row0_in_R;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = a.N, C = a.C, M = a.counts[b];
    const float* cls = a.cls + (size_t)b * N * C;
    const float* pair = a.pair + (size_t)b * N * a.Mmax * 2;
    for (int c = tid; c < C; c += 256) present[c] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int l = argmax_row(cls + (size_t)i * C, C, 1);
        label[i] = l;
        atomicOr(&present[l], 1);
    }
    for (int j = tid; j < M; j += 256) {
        const int l = argmax_row(a.gt_onehot + ((size_t)b * a.Mmax + j) * C, C, 1);
        gt_label[j] = l;
        atomicOr(&present[l], 1);
    }
    __syncthreads();
    for (int l = 0; l < C; ++l) {
        if (tid == 0) { s_nmask = 0; s_nR = 0; s_row0_in_R = 0; }
        __syncthreads();
        // most confident prediction of the whole sample for this class score (first row of the sorted order)
        if (tid == 0) s_row0 = argmax_row(cls + l, N, C);
        int cnt = 0;
        for (int i = tid; i < N; i += 256) cnt += label[i] == l;
        if (cnt) atomicAdd(&s_nmask, cnt);
        // per target: the most confident prediction of class l with IoU > thr (the row torch.max picks in the sorted
        // candidate mask, metric.py:103-113) and the best GIoU over the predictions of class l (:229-233)
        for (int j = tid; j < M; j += 256) {
            int bi = -1;
            float bs = 0.f, gm = -1.f;
            if (gt_label[j] == l) {
                for (int i = 0; i < N; ++i) {
                    if (label[i] != l) continue;
                    const float iou = pair[((size_t)i * a.Mmax + j) * 2 + 0], gg = pair[((size_t)i * a.Mmax + j) * 2 + 1];
                    gm = fmaxf(gm, gg);
                    const float sc = cls[(size_t)i * C + l];
                    if (iou > a.thr && (bi < 0 || sc > bs)) { bi = i; bs = sc; }
                }
            }
            best[j] = bi;
            colmax[j] = gm;
        }
        __syncthreads();
        if (tid == 0) {
            int npos = 0, nR = 0, row0_in_R = 0;
            float gsum = 0.f;
            for (int j = 0; j < M; ++j) {
                npos += gt_label[j] == l;
                gsum += colmax[j];
                if (best[j] < 0) continue;
                bool dup = false;
                for (int k = 0; k < j; ++k) dup = dup || best[k] == best[j];
                nR += dup ? 0 : 1;
                row0_in_R |= best[j] == s_row0;
            }
            const int row0 = s_row0;
            // ---- average precision: straight line through the first and last precision/recall point ----
            const float tp0 = row0_in_R ? 1.f : 0.f;
            const float fp0 = (label[row0] == l && !row0_in_R) ? 1.f : 0.f;
            const float p0 = (tp0 + fp0 != 0.f) ? tp0 / (fp0 + tp0) : 0.f;
            const float TP = (float)nR, FP = (float)(s_nmask - nR);
            const float p1 = (TP + FP != 0.f) ? TP / (FP + TP) : 0.f;
            const float r0 = npos == 0 ? 1.f : tp0 / (float)npos, r1 = npos == 0 ? 1.f : TP / (float)npos;
            const bool flat = fabsf(r1 - r0) <= 1e-8f;
            const float step = 1.f / (float)(a.nelem - 1);
            float sum = 0.f;
            for (int k = 0; k < a.nelem; ++k) {
                const float x = k < a.nelem / 2 ? step * (float)k : 1.f - step * (float)(a.nelem - k - 1);   // torch.linspace
                float y = flat ? 0.f : p0 + (x - r0) * (p1 - p0) / (r1 - r0);
                if (x < r0) y = p0;
                if (x > r1) y = 0.f;
                sum += y * 1.f / (float)(a.nelem - 1);
            }
            ap[l] = sum;
            // ---- class GIoU: mean over ALL target columns of the best GIoU (-1 for other-class columns) ----
            float gv = -1.f;
            if (npos == 0) gv = 1.f;
            if (M > 0 && N > 0 && s_nmask > 0 && npos > 0) gv = gsum / (float)M;
            gi[l] = gv;
        }
        __syncthreads();
    }
    if (tid == 0) {
        // contributing classes: the present labels without the SMALLEST present one (metric.py:141,243)
        int first = -1, nsel = 0, anynz = 0;
        float s_ap = 0.f, s_gi = 0.f;
        for (int c = 0; c < C; ++c) {
            if (!present[c]) continue;
            if (first < 0) { first = c; continue; }
            ++nsel; anynz |= c != 0;
            s_ap += ap[c]; s_gi += gi[c];
        }
        const bool none = nsel == 0 || !anynz;
        a.out[b * 2 + 0] = none ? 1.f : s_ap / (float)nsel;
        a.out[b * 2 + 1] = none ? 1.f : s_gi / (float)nsel;
    }
}

// ---- SetCriterion + Loss (src/dprt/training/loss.py:17-60 focal, :176-373 criterion, :486-564 reduction) ----
// One thread per (sample, query).  Terms: 0 total_class, 1 object_class, 2 center, 3 size, 4 angle; every sample's
// term is weighted and divided by B (reduction 'mean'); samples without targets contribute 0.
struct LossArgs {
    const float *cls, *center, *size, *angle;
    const float* gt_box;        // (B,Mmax,8)
    const float* gt_onehot;     // (B,Mmax,C)
    const int32_t* match;       // (B,Mmax,2) (query i, target j) in assignment order
    const int32_t* counts;
    const float* gout;          // (5) upstream gradient of the batch-reduced terms (backward)
    float* losses;              // (5) accumulated (caller zero-fills)            (forward)
    // forward, one launch without a cleared output (dpft_set_loss_fwd_total_f32): per-block partial sums + a ticket; the last
    // block adds them in block order, writes losses[5] and total = sum_k sel[k] * losses[k]
    float* slab;                // [ticket (8 floats)][blocks][8] or null
    const float* sel;
    float* total;
    float *dcls, *dcenter, *dsize, *dangle;                                     // (backward)
    float w[5];
    float alpha;
    int B, N, Mmax, C;
};
__device__ __forceinline__ void focal_term(float x, float t, float alpha, float& loss, float& dldx) {
    const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float pt = x * t + (1.f - x) * (1.f - t);       // raw logits, as in the reference (loss.py:44)
    const float om = 1.f - pt;
    const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
    loss = at * ce * om * om;
    const float sg = 1.f / (1.f + expf(-x));
    dldx = at * ((sg - t) * om * om - ce * 2.f * om * (2.f * t - 1.f));
}
template <bool BWD>
__global__ __launch_bounds__(256) void set_loss_kernel(LossArgs a) {
    __shared__ float red[5][4];
    const int64_t bn = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float part[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (bn < (int64_t)a.B * a.N) {
        const int b = (int)(bn / a.N), n = (int)(bn - (int64_t)b * a.N);
        const int M = a.counts[b];
        const int C = a.C;
        float g[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) g[k] = BWD ? a.gout[k] * a.w[k] / (float)a.B : a.w[k] / (float)a.B;
        int kk = -1;
        for (int k = 0; k < M; ++k)
            if (a.match[((int64_t)b * a.Mmax + k) * 2] == n) kk = k;
        const int jj = kk >= 0 ? a.match[((int64_t)b * a.Mmax + kk) * 2 + 1] : -1;
        const float* x = a.cls + bn * C;
        const float invM = M > 0 ? 1.f / (float)M : 0.f;
        // total_class over every query (one-hot row: background unless matched; matched rows take the kk-th target
        // row -- assignment order, loss.py:305-306), object_class over the matched ones
        for (int c = 0; c < C; ++c) {
            float dl = 0.f;
            if (M > 0) {
                const float t = kk >= 0 ? a.gt_onehot[((int64_t)b * a.Mmax + kk) * C + c] : (c == 0 ? 1.f : 0.f);
                float l, d;
                focal_term(x[c], t, a.alpha, l, d);
                part[0] += l * invM * g[0];
                dl += d * invM * g[0];
                if (kk >= 0) {
                    const float t2 = a.gt_onehot[((int64_t)b * a.Mmax + jj) * C + c];
                    focal_term(x[c], t2, a.alpha, l, d);
                    const float sc = (float)a.N * invM * invM * g[1];
                    part[1] += l * sc;
                    dl += d * sc;
                }
            }
            if (BWD) a.dcls[bn * C + c] = dl;
        }
        const float* gb = kk >= 0 ? a.gt_box + ((int64_t)b * a.Mmax + jj) * 8 : nullptr;
        for (int d = 0; d < 3; ++d) {
            float dc = 0.f, ds = 0.f;
            if (gb) {
                const float e1 = a.center[bn * 3 + d] - gb[d], e2 = a.size[bn * 3 + d] - gb[3 + d];
                const float s1 = invM / 3.f * g[2], s2 = invM / 3.f * g[3];
                part[2] += fabsf(e1) * s1;
                part[3] += fabsf(e2) * s2;
                dc = (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) * s1;
                ds = (e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f)) * s2;
            }
            if (BWD) {
                a.dcenter[bn * 3 + d] = dc;
                a.dsize[bn * 3 + d] = ds;
            }
        }
        for (int d = 0; d < 2; ++d) {
            float da = 0.f;
            if (gb) {
                const float e = a.angle[bn * 2 + d] - gb[6 + d];
                const float s = invM / 2.f * g[4];
                part[4] += fabsf(e) * s;
                da = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * s;
            }
            if (BWD) a.dangle[bn * 2 + d] = da;
        }
    }
    if (!BWD) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float v = part[k];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[k][wv] = v;
        }
        __syncthreads();
        if (a.slab == nullptr) {
            if (threadIdx.x < 5) atomicAdd(a.losses + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
            return;
        }
        __shared__ int last;
        __shared__ float tot[5];
        if (threadIdx.x < 5)
            __hip_atomic_store(a.slab + 8 + (size_t)blockIdx.x * 8 + threadIdx.x,
                               red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* ticket = reinterpret_cast<int*>(a.slab);
        if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        __syncthreads();
        if (!last) return;
        if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left clean
        if (threadIdx.x < 5) {
            float t = 0.f;
            for (int b = 0; b < (int)gridDim.x; ++b)
                t += __hip_atomic_load(a.slab + 8 + (size_t)b * 8 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.losses[threadIdx.x] = t;
            tot[threadIdx.x] = a.sel ? t * a.sel[threadIdx.x] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0 && a.total) *a.total = tot[0] + tot[1] + tot[2] + tot[3] + tot[4];
    }
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_giou3d_yaw_f32(const float* pred, const float* gt, float* out, int32_t B, int32_t N, int32_t Mg,
                                   dpft_stream_t stream) {
    DPFT_REQUIRE(pred && gt && out && B > 0 && N > 0 && Mg > 0, "giou3d_yaw: bad arguments");
    const int64_t total = (int64_t)B * N * Mg;
    hipLaunchKernelGGL(giou3d_yaw_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, pred, gt, out, B,
                       N, Mg);
    return check_launch("giou3d_yaw");
}

// Per-sample label tensors -> the padded batch tensors of the loss / matcher / metric kernels in ONE launch (the torch
// form -- cat x B, pad_sequence x 2, zeros, argmax, casts -- is ~17 launches that sit between the decoder and the matcher):
// gt_box (B,Mmax,8) = center | size | angle, gt_onehot (B,Mmax,C), gt_id (B,Mmax) = argmax of the class row (first
// maximum, like torch.argmax), counts (B).  Rows >= count are zero.
namespace dpft {
constexpr int PACK_MAX_B = 32;
struct PackTargetsArgs {
    const float* center[PACK_MAX_B];
    const float* size[PACK_MAX_B];
    const float* angle[PACK_MAX_B];
    const float* cls[PACK_MAX_B];
    int count[PACK_MAX_B];
    float* gt_box;
    float* gt_onehot;
    int* gt_id;
    int* counts;
    int B, Mmax, C;
};
__global__ __launch_bounds__(64) void pack_targets_kernel(PackTargetsArgs a) {
    const int b = blockIdx.x, n = a.count[b];
    if (threadIdx.x == 0) a.counts[b] = n;
    for (int m = threadIdx.x; m < a.Mmax; m += 64) {
        float* box = a.gt_box + ((size_t)b * a.Mmax + m) * 8;
        float* oh = a.gt_onehot + ((size_t)b * a.Mmax + m) * a.C;
        int id = 0;
        if (m < n) {
            for (int k = 0; k < 3; ++k) box[k] = a.center[b][m * 3 + k];
            for (int k = 0; k < 3; ++k) box[3 + k] = a.size[b][m * 3 + k];
            for (int k = 0; k < 2; ++k) box[6 + k] = a.angle[b][m * 2 + k];
            float best = -INFINITY;
            for (int c = 0; c < a.C; ++c) {
                const float v = a.cls[b][m * a.C + c];
                oh[c] = v;
                if (v > best) { best = v; id = c; }
            }
        } else {
            for (int k = 0; k < 8; ++k) box[k] = 0.f;
            for (int c = 0; c < a.C; ++c) oh[c] = 0.f;
        }
        a.gt_id[(size_t)b * a.Mmax + m] = id;
    }
}
}  // namespace dpft

extern "C" int dpft_pack_targets_f32(const float* const* center, const float* const* size, const float* const* angle,
                                     const float* const* cls, const int32_t* counts_host, int32_t B, int32_t Mmax, int32_t C,
                                     float* gt_box, float* gt_onehot, int32_t* gt_id, int32_t* counts, dpft_stream_t stream) {
    DPFT_REQUIRE(center && size && angle && cls && counts_host && gt_box && gt_onehot && gt_id && counts, "pack_targets: null argument");
    DPFT_REQUIRE(B >= 1 && B <= dpft::PACK_MAX_B && Mmax >= 1 && C >= 1, "pack_targets: 1..%d samples per call", dpft::PACK_MAX_B);
    dpft::PackTargetsArgs a;
    memset(&a, 0, sizeof(a));
    for (int b = 0; b < B; ++b) {
        DPFT_REQUIRE(counts_host[b] >= 0 && counts_host[b] <= Mmax, "pack_targets: sample %d has %d targets (Mmax %d)", b, counts_host[b], Mmax);
        DPFT_REQUIRE(counts_host[b] == 0 || (center[b] && size[b] && angle[b] && cls[b]), "pack_targets: sample %d has null labels", b);
        a.center[b] = center[b]; a.size[b] = size[b]; a.angle[b] = angle[b]; a.cls[b] = cls[b]; a.count[b] = counts_host[b];
    }
    a.gt_box = gt_box; a.gt_onehot = gt_onehot; a.gt_id = gt_id; a.counts = counts; a.B = B; a.Mmax = Mmax; a.C = C;
    hipLaunchKernelGGL(dpft::pack_targets_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    return dpft::check_launch("pack_targets");
}

extern "C" int dpft_match_cost_f32(const float* cls, const float* center, const float* size, const float* angle,
                                   const float* gt_box, const int32_t* gt_id, const int32_t* counts, const float* weights5,
                                   float* cost, int32_t B, int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream) {
    DPFT_REQUIRE(cls && center && size && angle && gt_box && gt_id && counts && weights5 && cost, "match_cost: null argument");
    DPFT_REQUIRE(B > 0 && N > 0 && Mmax > 0 && C > 0, "match_cost: non-positive sizes");
    CostArgs a;
    a.cls = cls; a.center = center; a.size = size; a.angle = angle; a.gt_box = gt_box; a.gt_id = gt_id; a.counts = counts;
    a.cost = cost; a.w_class = weights5[0]; a.w_center = weights5[1]; a.w_size = weights5[2]; a.w_angle = weights5[3];
    a.w_giou = weights5[4]; a.B = B; a.N = N; a.Mmax = Mmax; a.C = C;
    const int64_t total = (int64_t)B * N * Mmax;
    hipLaunchKernelGGL(match_cost_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, a);
    return check_launch("match_cost");
}

static int loss_fill(LossArgs& a, const float* cls, const float* center, const float* size, const float* angle,
                     const float* gt_box, const float* gt_onehot, const int32_t* match, const int32_t* counts,
                     const float* weights5, float alpha, int B, int N, int Mmax, int C) {
    DPFT_REQUIRE(cls && center && size && angle && gt_box && gt_onehot && match && counts && weights5, "set_loss: null argument");
    DPFT_REQUIRE(B > 0 && N > 0 && Mmax > 0 && C > 0, "set_loss: non-positive sizes");
    memset(&a, 0, sizeof(a));
    a.cls = cls; a.center = center; a.size = size; a.angle = angle; a.gt_box = gt_box; a.gt_onehot = gt_onehot;
    a.match = match; a.counts = counts; a.alpha = alpha; a.B = B; a.N = N; a.Mmax = Mmax; a.C = C;
    for (int k = 0; k < 5; ++k) a.w[k] = weights5[k];
    return DPFT_OK;
}

extern "C" int dpft_set_loss_fwd_f32(const float* cls, const float* center, const float* size, const float* angle,
                                     const float* gt_box, const float* gt_onehot, const int32_t* match,
                                     const int32_t* counts, const float* weights5, float alpha, float* losses5,
                                     int32_t B, int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream) {
    LossArgs a;
    int rc = loss_fill(a, cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, B, N, Mmax, C);
    if (rc) return rc;
    DPFT_REQUIRE(losses5, "set_loss_fwd: null output");
    a.losses = losses5;
    DPFT_REQUIRE(hipMemsetAsync(losses5, 0, 5 * sizeof(float), (hipStream_t)stream) == hipSuccess, "set_loss_fwd: memset failed");
    hipLaunchKernelGGL(set_loss_kernel<false>, dim3(cdiv((int64_t)B * N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("set_loss_fwd");
}

extern "C" int64_t dpft_set_loss_scratch_floats(int32_t B, int32_t N) { return 8 + 8 * (int64_t)cdiv((int64_t)B * N, 256); }

extern "C" int dpft_set_loss_fwd_total_f32(const float* cls, const float* center, const float* size, const float* angle,
                                           const float* gt_box, const float* gt_onehot, const int32_t* match,
                                           const int32_t* counts, const float* weights5, float alpha, const float* sel,
                                           float* scratch, float* losses5, float* total, int32_t B, int32_t N, int32_t Mmax,
                                           int32_t C, dpft_stream_t stream) {
    LossArgs a;
    int rc = loss_fill(a, cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, B, N, Mmax, C);
    if (rc) return rc;
    DPFT_REQUIRE(losses5 && scratch && sel && total, "set_loss_fwd_total: null tensor");
    a.losses = losses5; a.slab = scratch; a.sel = sel; a.total = total;
    hipLaunchKernelGGL(set_loss_kernel<false>, dim3(cdiv((int64_t)B * N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("set_loss_fwd_total");
}

extern "C" int dpft_set_loss_bwd_f32(const float* cls, const float* center, const float* size, const float* angle,
                                     const float* gt_box, const float* gt_onehot, const int32_t* match,
                                     const int32_t* counts, const float* weights5, float alpha, const float* gout5,
                                     float* dcls, float* dcenter, float* dsize, float* dangle, int32_t B, int32_t N,
                                     int32_t Mmax, int32_t C, dpft_stream_t stream) {
    LossArgs a;
    int rc = loss_fill(a, cls, center, size, angle, gt_box, gt_onehot, match, counts, weights5, alpha, B, N, Mmax, C);
    if (rc) return rc;
    DPFT_REQUIRE(gout5 && dcls && dcenter && dsize && dangle, "set_loss_bwd: null argument");
    a.gout = gout5; a.dcls = dcls; a.dcenter = dcenter; a.dsize = dsize; a.dangle = dangle;
    hipLaunchKernelGGL(set_loss_kernel<true>, dim3(cdiv((int64_t)B * N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("set_loss_bwd");
}

extern "C" int dpft_detection_metrics_f32(const float* cls, const float* center, const float* size, const float* angle,
                                          const float* gt_box, const float* gt_onehot, const int32_t* counts,
                                          float threshold, int32_t nelem, float* scratch, float* out, int32_t B,
                                          int32_t N, int32_t Mmax, int32_t C, dpft_stream_t stream) {
    DPFT_REQUIRE(cls && center && size && angle && gt_box && gt_onehot && counts && scratch && out, "detection_metrics: null argument");
    DPFT_REQUIRE(B > 0 && N > 0 && N <= kMetricMaxN && Mmax > 0 && Mmax <= kMetricMaxM && C > 0 && C <= kMetricMaxC && nelem >= 2,
                 "detection_metrics: sizes out of range (N <= %d, Mmax <= %d, C <= %d)", kMetricMaxN, kMetricMaxM, kMetricMaxC);
    MetricArgs a;
    a.cls = cls; a.center = center; a.size = size; a.angle = angle; a.gt_box = gt_box; a.gt_onehot = gt_onehot;
    a.counts = counts; a.pair = scratch; a.out = out; a.thr = threshold; a.nelem = nelem; a.B = B; a.N = N; a.Mmax = Mmax; a.C = C;
    const int64_t total = (int64_t)B * N * Mmax;
    hipLaunchKernelGGL(metric_pair_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, a);
    int rc = check_launch("detection_metrics pairs");
    if (rc) return rc;
    hipLaunchKernelGGL(detection_metric_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("detection_metrics");
}

// ---------------------------------------------------------------------------------------------------------
// Weight gradients of the fused training decoder from the per-row factor matrices its backward kernels write
// (decoder_train_x.hip XR_*, decoder_train_h.hip HR_*):  out[a][b] = sum_r rows[r][col_a + a] * rows[r][col_b + b].
// Replaces ~10 torch bmm / einsum / sum launches (Tensile kernels) per decoder layer by one launch; fixed summation
// order (deterministic).  Block = one 16 x 16 output tile: thread = (row group of 16, column b), 16 accumulators (all a),
// the 16 row groups are merged through LDS.
// ---------------------------------------------------------------------------------------------------------
namespace dpft {

constexpr int OUTER_MAX_SPECS = 40;
constexpr int OUTER_MAX_TILES = 192;      // 16 x 16 output tiles of one launch (the fused cross-attention block has 83)
struct OuterSpecs {
    dpft_outer_spec s[OUTER_MAX_SPECS];
    int n;
    // round 4: the grid holds exactly the tiles that exist -- (spec, tile) of block x.  The first form launched
    // max_tiles x n_specs blocks of 1 024 threads and let 84 % of them return at once: 21 k empty wave launches per call.
    unsigned char tile_spec[OUTER_MAX_TILES];
    unsigned char tile_idx[OUTER_MAX_TILES];
    int n_tiles;      // 0: the dense (max_tiles, n_specs) grid
};

__global__ __launch_bounds__(1024) void rows_outer_kernel(const float* __restrict__ rows, int R, int W, int64_t gstride_rows,
                                                          OuterSpecs sp, float* __restrict__ out, int64_t gstride_out) {
    const int si = sp.n_tiles ? (int)sp.tile_spec[blockIdx.x] : (int)blockIdx.y;
    const int ti = sp.n_tiles ? (int)sp.tile_idx[blockIdx.x] : (int)blockIdx.x;
    const dpft_outer_spec s = sp.s[si];
    const int tb = (s.n_b + 15) / 16, ta = (s.n_a + 15) / 16;
    if (ti >= ta * tb) return;
    const int a0 = (ti / tb) * 16, b0 = (ti % tb) * 16;
    const int b = threadIdx.x & 15, rg = threadIdx.x >> 4;      // 64 row groups: the loop is a latency chain over R / 64 rows
    const float* base = rows + (int64_t)blockIdx.z * gstride_rows;
    const bool ones = s.col_b < 0;                       // column sums: the b operand is 1
    const bool bok = ones ? b == 0 : (b0 + b < s.n_b);
    const int na = min(16, s.n_a - a0);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // the 16 a-values of a row are one 64-byte segment (column offsets and tiles are multiples of 16 floats when the row
    // pitch is a multiple of 4): four 16-byte loads instead of sixteen scalar ones, four rows in flight per thread --
    // the loop is a latency chain over R / 16 rows, not a bandwidth problem
    const bool vec = ((s.col_a & 3) == 0) && ((W & 3) == 0) && na == 16 && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
    if (vec) {
#pragma unroll 4
        for (int r = rg; r < R; r += 64) {
            const float* row = base + (int64_t)r * W;
            const float bv = bok ? (ones ? 1.f : row[s.col_b + b0 + b]) : 0.f;
            const f32x4* ap = reinterpret_cast<const f32x4*>(row + s.col_a + a0);
            const f32x4 a0v = ap[0], a1v = ap[1], a2v = ap[2], a3v = ap[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = fmaf(a0v[i], bv, acc[i]);
                acc[4 + i] = fmaf(a1v[i], bv, acc[4 + i]);
                acc[8 + i] = fmaf(a2v[i], bv, acc[8 + i]);
                acc[12 + i] = fmaf(a3v[i], bv, acc[12 + i]);
            }
        }
    } else {
#pragma unroll 2
        for (int r = rg; r < R; r += 64) {
            const float* row = base + (int64_t)r * W;
            const float bv = bok ? (ones ? 1.f : row[s.col_b + b0 + b]) : 0.f;
            float av[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) av[i] = i < na ? row[s.col_a + a0 + i] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fmaf(av[i], bv, acc[i]);
        }
    }
    // the 4 row groups of a wave first (lanes b, b + 16, b + 32, b + 48), then the 16 waves through LDS, in a fixed order
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    __shared__ float red[16][16][17];                   // [wave][a][b]
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i][b] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x >= 256) return;
    const int a = threadIdx.x >> 4;                      // thread -> output (a, b): sum the 16 waves in order
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][a][b];
    const int nb_out = ones ? 1 : s.n_b;
    if (a < na && (ones ? b == 0 : b0 + b < s.n_b))
        out[(int64_t)blockIdx.z * gstride_out + s.out_off + (int64_t)(a0 + a) * nb_out + (ones ? 0 : b0 + b)] = t;
}

}  // namespace dpft

extern "C" int dpft_rows_outer_f32(const float* rows, int32_t G, int32_t R, int32_t W, const dpft_outer_spec* specs,
                                   int32_t n_specs, float* out, int64_t out_gstride, dpft_stream_t stream) {
    DPFT_REQUIRE(rows && specs && out && G > 0 && R > 0 && W > 0, "rows_outer: bad arguments");
    DPFT_REQUIRE(n_specs >= 1 && n_specs <= dpft::OUTER_MAX_SPECS, "rows_outer: 1..%d specs per call", dpft::OUTER_MAX_SPECS);
    dpft::OuterSpecs sp;
    sp.n = n_specs;
    int max_tiles = 1;
    for (int i = 0; i < n_specs; ++i) {
        const dpft_outer_spec& s = specs[i];
        DPFT_REQUIRE(s.n_a >= 1 && s.col_a >= 0 && s.col_a + s.n_a <= W && s.out_off >= 0 &&
                     (s.col_b < 0 ? s.n_b == 1 : (s.n_b >= 1 && s.col_b + s.n_b <= W)), "rows_outer: spec %d out of range", i);
        sp.s[i] = s;
        max_tiles = std::max(max_tiles, ((s.n_a + 15) / 16) * ((s.n_b + 15) / 16));
    }
    // compact grid: one block per existing tile (tiles per spec <= 255, all tiles <= OUTER_MAX_TILES), else the dense grid
    int nt = 0;
    bool compact = true;
    for (int i = 0; i < n_specs && compact; ++i) {
        const int tiles = ((specs[i].n_a + 15) / 16) * ((specs[i].n_b + 15) / 16);
        if (tiles > 255 || nt + tiles > dpft::OUTER_MAX_TILES) { compact = false; break; }
        for (int t = 0; t < tiles; ++t) { sp.tile_spec[nt] = (unsigned char)i; sp.tile_idx[nt] = (unsigned char)t; ++nt; }
    }
    sp.n_tiles = compact ? nt : 0;
    const dim3 grid = compact ? dim3(nt, 1, G) : dim3(max_tiles, n_specs, G);
    hipLaunchKernelGGL(dpft::rows_outer_kernel, grid, dim3(1024), 0, (hipStream_t)stream, rows, R, W,
                       (int64_t)R * W, sp, out, out_gstride);
    return dpft::check_launch("rows_outer");
}


// ---------------------------------------------------------------------------------------------------------------------
// dpft_memops: up to DPFT_MEMOPS_MAX device-to-device copies / zero fills in ONE launch (the small per-step input copies
// of a replayed decoder graph, the clears of a gradient reducer, the static-address input copies of the launch plans).
// The host glue used the runtime's blit kernels for these (one launch per tensor).
// ---------------------------------------------------------------------------------------------------------------------
namespace dpft {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct MemOps {
    int n;
    int first_block[DPFT_MEMOPS_MAX + 1];      // blocks [first_block[i], first_block[i + 1]) work on op i
    dpft_memop op[DPFT_MEMOPS_MAX];
};
constexpr int MEMOP_BLOCK_BYTES = 256 * 16 * 8;      // 8 sixteen-byte accesses per thread

__global__ __launch_bounds__(256) void memops_kernel(MemOps m) {
    int i = 0;
    while (i + 1 < m.n && (int)blockIdx.x >= m.first_block[i + 1]) ++i;
    const dpft_memop o = m.op[i];
    const int nblk = m.first_block[i + 1] - m.first_block[i];
    const int blk = blockIdx.x - m.first_block[i];
    const bool vec = (((uintptr_t)o.dst | (uintptr_t)o.src) & 15) == 0;
    if (vec) {
        const uint64_t n16 = o.bytes >> 4;
        u32x4* d = reinterpret_cast<u32x4*>(o.dst);
        const u32x4* s = reinterpret_cast<const u32x4*>(o.src);
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (uint64_t k = (uint64_t)blk * 256 + threadIdx.x; k < n16; k += (uint64_t)nblk * 256) d[k] = s ? s[k] : z;
        const uint64_t done = n16 << 4;                      // tail: 4-byte words
        unsigned* dt = reinterpret_cast<unsigned*>((char*)o.dst + done);
        const unsigned* st_ = s ? reinterpret_cast<const unsigned*>((const char*)o.src + done) : nullptr;
        if (blk == 0 && threadIdx.x < ((o.bytes - done) >> 2)) dt[threadIdx.x] = st_ ? st_[threadIdx.x] : 0u;
    } else {
        const uint64_t n4 = o.bytes >> 2;
        unsigned* d = reinterpret_cast<unsigned*>(o.dst);
        const unsigned* s = reinterpret_cast<const unsigned*>(o.src);
        for (uint64_t k = (uint64_t)blk * 256 + threadIdx.x; k < n4; k += (uint64_t)nblk * 256) d[k] = s ? s[k] : 0u;
    }
}
}  // namespace dpft

extern "C" int dpft_memops(int32_t n, const dpft_memop* ops, dpft_stream_t stream) {
    DPFT_REQUIRE(ops && n >= 1 && n <= DPFT_MEMOPS_MAX, "memops: 1..%d operations per call", DPFT_MEMOPS_MAX);
    dpft::MemOps m;
    m.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        DPFT_REQUIRE(ops[i].dst && (ops[i].bytes & 3) == 0 && (((uintptr_t)ops[i].dst | (uintptr_t)ops[i].src) & 3) == 0,
                     "memops: operation %d needs 4-byte aligned pointers and a size that is a multiple of 4", i);
        m.op[i] = ops[i];
        m.first_block[i] = blocks;
        const uint64_t want = (ops[i].bytes + dpft::MEMOP_BLOCK_BYTES - 1) / dpft::MEMOP_BLOCK_BYTES;
        blocks += (int)std::max<uint64_t>(1, std::min<uint64_t>(want, 2048));
    }
    m.first_block[n] = blocks;
    hipLaunchKernelGGL(dpft::memops_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, m);
    return dpft::check_launch("memops");
}


// ---------------------------------------------------------------------------------------------------------------------
// dpft_sum_leading_f32: dst[i] (+)= sum_s sum_l src_s[l * inner + i], i < inner -- the leading-axis sums of the training
// decoder's backward (gradients of a tensor broadcast over views / batch elements / replicas, and of a parameter used by
// several blocks) in ONE launch with a fixed summation order (sources in table order, leading index ascending).
// Replaces Tensor.sum(...) + the autograd engine's add chains.
// ---------------------------------------------------------------------------------------------------------------------
namespace dpft {
struct SumSrcs {
    int n;
    dpft_sum_src s[DPFT_SUM_SRCS_MAX];
};
__global__ __launch_bounds__(256) void sum_leading_kernel(SumSrcs t, int64_t inner4, float* __restrict__ dst, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= inner4) return;
    f32x4 acc = accumulate ? reinterpret_cast<const f32x4*>(dst)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < t.n; ++k) {
        const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(t.s[k].src) + i;
        const int nl = t.s[k].n_lead;
        int l = 0;
        for (; l + 4 <= nl; l += 4) {      // four loads in flight, added in order
            const f32x4 a = p[(int64_t)l * inner4], b = p[(int64_t)(l + 1) * inner4], c = p[(int64_t)(l + 2) * inner4],
                        d = p[(int64_t)(l + 3) * inner4];
            acc += a; acc += b; acc += c; acc += d;
        }
        for (; l < nl; ++l) acc += p[(int64_t)l * inner4];
    }
    reinterpret_cast<f32x4*>(dst)[i] = acc;
}
}  // namespace dpft

extern "C" int dpft_sum_leading_f32(int32_t n_src, const dpft_sum_src* srcs, int64_t inner, float* dst, int32_t accumulate,
                                    dpft_stream_t stream) {
    DPFT_REQUIRE(srcs && dst && n_src >= 1 && n_src <= DPFT_SUM_SRCS_MAX, "sum_leading: 1..%d sources per call", DPFT_SUM_SRCS_MAX);
    DPFT_REQUIRE(inner > 0 && (inner & 3) == 0 && ((uintptr_t)dst & 15) == 0, "sum_leading: inner %% 4 == 0, 16-byte aligned tensors");
    dpft::SumSrcs t;
    t.n = n_src;
    for (int i = 0; i < n_src; ++i) {
        DPFT_REQUIRE(srcs[i].src && srcs[i].n_lead >= 1 && ((uintptr_t)srcs[i].src & 15) == 0, "sum_leading: bad source %d", i);
        t.s[i] = srcs[i];
    }
    const int64_t inner4 = inner / 4;
    hipLaunchKernelGGL(dpft::sum_leading_kernel, dim3((unsigned)((inner4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, inner4,
                       dst, accumulate);
    return dpft::check_launch("sum_leading");
}


// dropout seed of the training decoder (train_fused.py advance_seed): snap = state; state += increment -- one launch
// (captured in the forward graph) instead of a clone and an add
namespace dpft {
__global__ void seed_advance_kernel(long long* __restrict__ state, long long* __restrict__ snap, long long inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long v = *state;
        *snap = v;
        *state = v + inc;
    }
}
}  // namespace dpft

extern "C" int dpft_seed_advance(int64_t* state, int64_t* snap, int64_t increment, dpft_stream_t stream) {
    DPFT_REQUIRE(state && snap, "seed_advance: null tensor");
    hipLaunchKernelGGL(dpft::seed_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)state, (long long*)snap,
                       (long long)increment);
    return dpft::check_launch("seed_advance");
}


// *ptrs[i] += increment for up to DPFT_I64_PTRS_MAX int64 scalars in one launch (BatchNorm's num_batches_tracked counters of
// a backbone: torch._foreach_add_ took a vendor multi-tensor kernel per step and encoder)
namespace dpft {
struct I64Ptrs {
    long long* p[DPFT_I64_PTRS_MAX];
};
__global__ void i64_add_many_kernel(I64Ptrs t, int n, long long inc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *t.p[i] += inc;
}
}  // namespace dpft

extern "C" int dpft_i64_add_many(int32_t n, int64_t* const* ptrs, int64_t increment, dpft_stream_t stream) {
    DPFT_REQUIRE(ptrs && n >= 1 && n <= DPFT_I64_PTRS_MAX, "i64_add_many: 1..%d pointers per call", DPFT_I64_PTRS_MAX);
    dpft::I64Ptrs t;
    for (int i = 0; i < n; ++i) {
        DPFT_REQUIRE(ptrs[i] && ((uintptr_t)ptrs[i] & 7) == 0, "i64_add_many: pointer %d null or not 8-byte aligned", i);
        t.p[i] = (long long*)ptrs[i];
    }
    hipLaunchKernelGGL(dpft::i64_add_many_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, t, n, (long long)increment);
    return dpft::check_launch("i64_add_many");
}


// dst_i[k] += src_i[k] for n (dst, src, bytes) entries of a DEVICE-resident table, one block per entry: the captured
// decoder backward adds its parameter gradients into the data-parallel buckets (static addresses: the table is written
// once after the capture) -- was torch._foreach_add_ (five vendor multi-tensor launches per replay)
namespace dpft {
__global__ __launch_bounds__(256) void add_many_kernel(const dpft_memop* __restrict__ table) {
    const dpft_memop o = table[blockIdx.x];
    float* __restrict__ d = reinterpret_cast<float*>(o.dst);
    const float* __restrict__ s = reinterpret_cast<const float*>(o.src);
    const uint64_t n = o.bytes >> 2;
    if (((((uintptr_t)d) | ((uintptr_t)s)) & 15) == 0) {
        const uint64_t n4 = n >> 2;
        for (uint64_t k = threadIdx.x; k < n4; k += 256) {
            f32x4 v = reinterpret_cast<f32x4*>(d)[k];
            v += reinterpret_cast<const f32x4*>(s)[k];
            reinterpret_cast<f32x4*>(d)[k] = v;
        }
        for (uint64_t k = (n4 << 2) + threadIdx.x; k < n; k += 256) d[k] += s[k];
    } else {
        for (uint64_t k = threadIdx.x; k < n; k += 256) d[k] += s[k];
    }
}
}  // namespace dpft

extern "C" int dpft_add_many_f32(int32_t n, const dpft_memop* table_device, dpft_stream_t stream) {
    DPFT_REQUIRE(table_device && n >= 1, "add_many: empty table");
    hipLaunchKernelGGL(dpft::add_many_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, table_device);
    return dpft::check_launch("add_many");
}
