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

__global__ void giou3d_yaw_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ out,
                                  int B, int N, int Mg) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * N * Mg) return;
    const int j = (int)(idx % Mg);
    const int i = (int)((idx / Mg) % N);
    const int b = (int)(idx / ((int64_t)Mg * N));
    const float* pa = pred + ((int64_t)b * N + i) * 7;
    const float* pb = gt + ((int64_t)b * Mg + j) * 7;
    const double eps = 1e-4;
    auto valid = [&](const float* s) {
        const double l = s[3], w = s[4], h = s[5];
        return fmin(fmin(l * w, l * h), w * h) / 2 > eps;
    };
    if (!(valid(pa) && valid(pb))) {  // iou.py:159,185-208: evol keeps -1 => giou = -1
        out[idx] = -1.f;
        return;
    }
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
    out[idx] = (float)(evol != 0 ? iou - (evol - uni) / evol : 0.0);
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
