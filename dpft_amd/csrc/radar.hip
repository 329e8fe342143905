// Radar tesseract -> range-azimuth (RA) and elevation-azimuth (EA) feature maps, the offline preprocessing step of
// the reference (KRadarProcessor.get_radar_data, src/dprt/datasets/kradar/processor.py:588-633) that turns each
// 64 x 256 x 37 x 107 fp32 power cube (259 MB) into the two 6-channel radar inputs of the model.
// Per output cell: dB = 10 log10(power);   rcs (max, median, variance)   = (max_d max_s, median_d median_s, var_d var_s)
//                                          doppler (peak, centre, spread) = (raster[argmax_d max_s], median_d / mean_d of
//                                          max_s, var_d max_s)            s = folded spatial dimension (elevation | range)
// with the reference's quirks: EA folds the cropped range bins [4, 252) and its doppler centre is a MEAN.
// Pass 1 (twice: fold elevation for RA, fold the cropped range for EA): every (doppler, row, azimuth) column is held
// in REGISTERS by 1-4 cooperating lanes, (max, median, variance) by compare-count bisection (exact elements, so medians
// are bit-exact given the dB values; variances in fp64).  Pass 2 folds the 64 dopplers of each output cell the same
// way.  The cube is read exactly twice: HBM-bound ideal ~2 x 259 MB / 8 TB/s = 65 us.
#include "common.h"

namespace dpft {

struct RadarArgs {
    const float* t;          // (D, R, E, A) linear power
    const float* raster;     // (D) doppler raster [m/s]
    float* ra;               // (R, A, 6)
    float* ea;               // (E, A, 6)
    float* sra;              // (D, R, A, 3): per-doppler (max, median, var) over elevation
    float* sea;              // (D, E, A, 3): per-doppler (max, median, var) over the cropped range
    int D, R, E, A;
    int r_lo, r_hi;          // EA range crop [r_lo, r_hi)
};

__device__ __forceinline__ uint32_t f2key(float v) {      // order-preserving float -> uint
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ float to_db(float p) { return 10.f * log10f(p); }

template <int SEG>
__device__ __forceinline__ float seg_sum(float v) {      // lanes of a column: lane ^ (64 / SEG), lane ^ (128 / SEG)
    if (SEG == 2) v += __shfl_xor(v, 32);
    if (SEG == 4) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); }
    return v;
}
template <int SEG>
__device__ __forceinline__ double seg_sum(double v) {      // lanes of a column: lane ^ (64 / SEG), lane ^ (128 / SEG)
    if (SEG == 2) v += __shfl_xor(v, 32);
    if (SEG == 4) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); }
    return v;
}
template <int SEG>
__device__ __forceinline__ int seg_sum(int v) {      // lanes of a column: lane ^ (64 / SEG), lane ^ (128 / SEG)
    if (SEG == 2) v += __shfl_xor(v, 32);
    if (SEG == 4) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); }
    return v;
}
template <int SEG>
__device__ __forceinline__ float seg_max(float v) {
    if (SEG == 2) v = fmaxf(v, __shfl_xor(v, 32));
    if (SEG == 4) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); }
    return v;
}
template <int SEG>
__device__ __forceinline__ float seg_min(float v) {
    if (SEG == 2) v = fminf(v, __shfl_xor(v, 32));
    if (SEG == 4) { v = fminf(v, __shfl_xor(v, 16)); v = fminf(v, __shfl_xor(v, 32)); }
    return v;
}

// (max, median, variance) of a column of n values held in REGISTERS: each of the SEG lanes that share the column
// (lane, lane ^ 16, lane ^ 32, ...) owns NREG values (v[i] valid for i < cnt).  The median is found by bisection on the
// order-preserving key: ~24 passes of NREG compare+add, counts summed over the SEG lanes -- no LDS, no data-dependent
// addressing, no divergence; exact element values (numpy.median semantics: mean of the two middle elements if n is even).
template <int NREG, int SEG>
__device__ __forceinline__ void column_stats(const float (&v)[NREG], int cnt, int n, float& mx, float& med, float& var) {
    float lmx = -INFINITY, lmn = INFINITY;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i)
        if (i < cnt) { lmx = fmaxf(lmx, v[i]); lmn = fminf(lmn, v[i]); s += v[i]; }
    mx = seg_max<SEG>(lmx);
    const float mn = seg_min<SEG>(lmn);
    const double mean = seg_sum<SEG>(s) / n;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i)
        if (i < cnt) { const double d = (double)v[i] - mean; q += d * d; }
    var = (float)(seg_sum<SEG>(q) / n);
    uint32_t key[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) key[i] = i < cnt ? f2key(v[i]) : 0xFFFFFFFFu;      // padding never counts
    const int k = n >> 1;
    uint32_t lo = f2key(mn), hi = f2key(mx);
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NREG; ++i) c += key[i] <= mid;
        c = seg_sum<SEG>(c);
        if (c >= k + 1) hi = mid;
        else lo = mid + 1;
    }
    const float hiv = key2f(lo);
    if (n & 1) { med = hiv; return; }
    int less = 0;
    float below = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i)
        if (i < cnt && v[i] < hiv) { ++less; below = fmaxf(below, v[i]); }
    less = seg_sum<SEG>(less);
    below = seg_max<SEG>(below);
    med = ((less >= k ? below : hiv) + hiv) * 0.5f;
}

// Fold one spatial axis (n values at element stride `fstride`) for every (doppler d, row, azimuth): a wave handles
// 64 / SEG azimuth columns; lane = seg * (64 / SEG) + column, element i of a segment = seg + SEG * i.
// grid = (ceil(A / cols), rows, D);  out (D, rows, A, 3)
template <int NREG, int SEG>
__global__ __launch_bounds__(64) void radar_fold_kernel(const float* __restrict__ t, float* __restrict__ out, int rows,
                                                        int A, int n, size_t fstride, size_t d_stride, size_t row_stride,
                                                        size_t base) {
    constexpr int COLS = 64 / SEG;
    const int lane = threadIdx.x, seg = lane / COLS, c = lane % COLS;
    const int az = blockIdx.x * COLS + c, row = blockIdx.y, d = blockIdx.z;
    const bool ok = az < A;
    const float* src = t + base + (size_t)d * d_stride + (size_t)row * row_stride + (ok ? az : 0);
    const int cnt = ok ? (n - seg + SEG - 1) / SEG : 0;          // elements seg, seg + SEG, ... < n
    float v[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = i < cnt ? src[(size_t)(seg + SEG * i) * fstride] : 1.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = to_db(v[i]);
    float mx, med, var;
    column_stats<NREG, SEG>(v, cnt, n, mx, med, var);
    if (ok && seg == 0) {
        float* o = out + (((size_t)d * rows + row) * A + az) * 3;
        o[0] = mx; o[1] = med; o[2] = var;
    }
}

// Fold the doppler axis of a (D, cells, 3) scratch: one lane per output cell, D <= 64 values in registers.
// feats (cells, 6) = (rcs max, median of medians, var of vars, raster[argmax peak], median | mean of peaks, var of peaks)
template <int NREG>
__global__ __launch_bounds__(64) void radar_finish_kernel(const float* __restrict__ scr, const float* __restrict__ raster,
                                                          float* __restrict__ feats, int64_t cells, int D, int mean_centre) {
    const int64_t cell = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const bool ok = cell < cells;
    const int cnt = ok ? D : 0;
    float v[NREG];
    float mx, med, var;
    // peaks: max / first argmax / median (or mean) / variance
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = i < cnt ? scr[((size_t)i * cells + cell) * 3 + 0] : 0.f;
    column_stats<NREG, 1>(v, cnt, D, mx, med, var);
    int arg = 0;
    double s = 0.0;
#pragma unroll
    for (int i = NREG - 1; i >= 0; --i)
        if (i < cnt) { if (v[i] == mx) arg = i; s += v[i]; }
    float o0 = mx, o3 = ok ? raster[arg] : 0.f, o4 = mean_centre ? (float)(s / D) : med, o5 = var;
    if (mean_centre) {      // numpy.mean sums in order; keep the forward order for the last bits
        s = 0.0;
#pragma unroll
        for (int i = 0; i < NREG; ++i)
            if (i < cnt) s += v[i];
        o4 = (float)(s / D);
    }
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = i < cnt ? scr[((size_t)i * cells + cell) * 3 + 1] : 0.f;
    float m2, d2, v2;
    column_stats<NREG, 1>(v, cnt, D, m2, d2, v2);
    const float o1 = d2;
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = i < cnt ? scr[((size_t)i * cells + cell) * 3 + 2] : 0.f;
    column_stats<NREG, 1>(v, cnt, D, m2, d2, v2);
    const float o2 = v2;
    if (ok) {
        float* o = feats + cell * 6;
        o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3; o[4] = o4; o[5] = o5;
    }
}

template <int SEG>
static int launch_fold(const float* t, float* out, int rows, int A, int n, size_t fstride, size_t d_stride, size_t row_stride,
                       size_t base, int D, hipStream_t st) {
    constexpr int COLS = 64 / SEG;
    const int per = cdiv(n, SEG);
    dim3 grid(cdiv(A, COLS), rows, D);
#define FOLD(NR)                                                                                                     \
    hipLaunchKernelGGL((radar_fold_kernel<NR, SEG>), grid, dim3(64), 0, st, t, out, rows, A, n, fstride, d_stride, \
                       row_stride, base)
    if (per <= 16) FOLD(16);
    else if (per <= 40) FOLD(40);
    else FOLD(64);
#undef FOLD
    return check_launch("radar_fold");
}

static int fold_axis(const float* t, float* out, int rows, int A, int n, size_t fstride, size_t d_stride, size_t row_stride,
                     size_t base, int D, hipStream_t st) {
    DPFT_REQUIRE(n >= 1 && n <= 256, "radar_projection: a folded axis of %d elements is outside [1, 256]", n);
    if (n <= 64) return launch_fold<1>(t, out, rows, A, n, fstride, d_stride, row_stride, base, D, st);
    if (n <= 128) return launch_fold<2>(t, out, rows, A, n, fstride, d_stride, row_stride, base, D, st);
    return launch_fold<4>(t, out, rows, A, n, fstride, d_stride, row_stride, base, D, st);
}

}  // namespace dpft

using namespace dpft;

extern "C" int64_t dpft_radar_projection_scratch_floats(int32_t D, int32_t R, int32_t E, int32_t A) {
    return (int64_t)D * ((int64_t)R + E) * A * 3;
}

extern "C" int dpft_radar_projection_f32(const float* tesseract, const float* doppler_raster, float* ra, float* ea,
                                         float* scratch, int32_t D, int32_t R, int32_t E, int32_t A, int32_t r_lo,
                                         int32_t r_hi, dpft_stream_t stream) {
    DPFT_REQUIRE(tesseract && doppler_raster && ra && ea && scratch, "radar_projection: null argument");
    DPFT_REQUIRE(D > 0 && D <= 64 && R > 0 && E > 0 && A > 0 && 0 <= r_lo && r_lo < r_hi && r_hi <= R,
                 "radar_projection: bad sizes (D <= 64)");
    hipStream_t st = (hipStream_t)stream;
    float* sra = scratch;
    float* sea = scratch + (size_t)D * R * A * 3;
    const size_t EA = (size_t)E * A, REA = (size_t)R * EA;
    // RA: rows = range bins, folded axis = elevation (stride A)
    int rc = fold_axis(tesseract, sra, R, A, E, (size_t)A, REA, EA, 0, D, st);
    if (rc) return rc;
    // EA: rows = elevations, folded axis = cropped range (stride E*A)
    rc = fold_axis(tesseract, sea, E, A, r_hi - r_lo, EA, REA, (size_t)A, (size_t)r_lo * EA, D, st);
    if (rc) return rc;
    hipLaunchKernelGGL(radar_finish_kernel<64>, dim3(cdiv((int64_t)R * A, 64)), dim3(64), 0, st, sra, doppler_raster, ra,
                       (int64_t)R * A, D, 0);
    rc = check_launch("radar_finish ra");
    if (rc) return rc;
    hipLaunchKernelGGL(radar_finish_kernel<64>, dim3(cdiv((int64_t)E * A, 64)), dim3(64), 0, st, sea, doppler_raster, ea,
                       (int64_t)E * A, D, 1);
    return check_launch("radar_finish ea");
}
