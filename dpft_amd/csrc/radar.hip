// Radar tesseract -> range-azimuth (RA) and elevation-azimuth (EA) feature maps, the offline preprocessing step of
// the reference (KRadarProcessor.get_radar_data, src/dprt/datasets/kradar/processor.py:588-633) that turns each
// 64 x 256 x 37 x 107 fp32 power cube (259 MB) into the two 6-channel radar inputs of the model.
// Per output cell: dB = 10 log10(power);   rcs (max, median, variance)   = (max_d max_s, median_d median_s, var_d var_s)
//                                          doppler (peak, centre, spread) = (raster[argmax_d max_s], median_d / mean_d of
//                                          max_s, var_d max_s)            s = folded spatial dimension (elevation | range)
// with the reference's quirks: EA folds the cropped range bins [4, 252) and its doppler centre is a MEAN.
// Pass 1 (twice: fold elevation for RA, fold the cropped range for EA): one LANE per (doppler, row, azimuth) column, the
// whole column (37 | 248 values) in that lane's REGISTERS -- no LDS, no cross-lane traffic, and the lanes of a wave walk
// consecutive cells of the flattened (row, azimuth) plane, so every load instruction covers 256 contiguous bytes.
// (max, median, variance) per column: the median by compare-count bisection on an order-preserving key with an exact-rank
// early exit (exact element values => bit-exact medians given the dB values).  Pass 2 folds the 64 dopplers of each
// output cell the same way.  The cube is read exactly twice: HBM-bound ideal ~2 x 259 MB / 8 TB/s = 65 us.
#include "common.h"

namespace dpft {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t f2key(float v) {      // order-preserving float -> uint
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// 10 log10(p) = log2(p) * 10 log10(2): v_log_f32 (1 ulp) and the constant split in two terms -- within ~2 ulp of the
// reference's numpy log10 for normal p (the cube holds powers >= 1; denormal powers lose the v_log_f32 guarantee)
__device__ __forceinline__ float to_db(float p) {
    const float l = __log2f(p);
    constexpr float c_hi = 3.0102999566f, c_lo = (float)(3.010299956639812 - (double)c_hi);
    return fmaf(l, c_hi, l * c_lo);
}

// all-reduce over the SEG (1 | 2) lanes that share a column: lane and lane ^ 32 (v_permlane32_swap, no LDS)
template <int SEG> __device__ __forceinline__ float seg_max(float v) {
    if (SEG == 1) return v;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return fmaxf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
template <int SEG> __device__ __forceinline__ float seg_min(float v) {
    if (SEG == 1) return v;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return fminf(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
template <int SEG> __device__ __forceinline__ float seg_sum(float v) {
    if (SEG == 1) return v;
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
template <int SEG> __device__ __forceinline__ int seg_sum(int v) {
    if (SEG == 1) return v;
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (int)r[0] + (int)r[1];
}

// Register columns carry NO per-element validity: a lane's NREG registers hold its elements followed by `npad` surplus
// copies of its last one (what a clamped load index delivers for free; v[NREG - 1] is always such a copy or the last
// element itself).  max / min / "largest <= x" ignore duplicates; sums and rank counts subtract npad * f(v[NREG - 1]).

// k-th smallest (k = n / 2) of a column of n values; numpy.median semantics: the mean of the two middle elements when n
// is even.  Bisection on the key space between the column's min and max; a probe whose rank count is exactly k + 1 ends
// the search (the answer is then the largest value <= probe), so a column of n spread values needs ~log2(n) + 2 probes
// instead of the ~24 a full key bisection takes; ties fall through to the full search.
template <int NREG, int SEG>
__device__ __forceinline__ float median_regs(const float (&v)[NREG], int npad, int n, float mn, float mx) {
    const int k = n >> 1;
    const float vl = v[NREG - 1];
    uint32_t lo = f2key(mn), hi = f2key(mx);
    // rank count of a probe, two elements per instruction: [v <= probe] = [v < next(probe)] = clamp01((next - v) * 2^100)
    // -- one v_pk_fma_f32 with the clamp modifier (exact: the product terms are scaled by a power of two and fused) and
    // one v_pk_add_f32 per PAIR, instead of compare + carry-add per element.  Holds for |v| < 2^27 and value gaps
    // >= 2^-100 (dB values and their variances are far inside).
    const f32x2 nh = {-0x1p100f, -0x1p100f};
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const float nxt = key2f(mid + 1);            // mid < hi <= key(max): finite
        const float mh = nxt * 0x1p100f;
        const f32x2 mh2 = {mh, mh};
        f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < NREG / 2; ++j) {
            const f32x2 pr = {v[2 * j], v[2 * j + 1]};
            f32x2 r;
            asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(pr), "v"(nh), "v"(mh2));
            acc[j & 1] += r;
        }
        const f32x2 a2 = acc[0] + acc[1];
        int c = (int)(a2[0] + a2[1]);
        c -= vl < nxt ? npad : 0;
        c = seg_sum<SEG>(c);
        if (c >= k + 1) hi = mid;
        else lo = mid + 1;
        if (c == k + 1) lo = hi;
    }
    const float thr = key2f(hi);
    float hiv = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) hiv = v[i] <= thr ? fmaxf(hiv, v[i]) : hiv;
    hiv = seg_max<SEG>(hiv);
    if (n & 1) return hiv;
    int less = 0;
    float below = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const bool lt = v[i] < hiv;
        less += lt;
        below = lt ? fmaxf(below, v[i]) : below;
    }
    less -= vl < hiv ? npad : 0;
    less = seg_sum<SEG>(less);
    below = seg_max<SEG>(below);
    return ((less >= k ? below : hiv) + hiv) * 0.5f;
}

template <int NREG, int SEG>
__device__ __forceinline__ void minmax_regs(const float (&v)[NREG], float& mn, float& mx) {
    mn = v[0]; mx = v[0];
#pragma unroll
    for (int i = 1; i < NREG; ++i) {
        mn = fminf(mn, v[i]);
        mx = fmaxf(mx, v[i]);
    }
    mn = seg_min<SEG>(mn);
    mx = seg_max<SEG>(mx);
}

// population variance, two passes, four interleaved fp32 accumulators (the reference: numpy float32 var)
template <int NREG, int SEG>
__device__ __forceinline__ float var_regs(const float (&v)[NREG], int npad, int n) {
    const float vl = v[NREG - 1], fp = (float)npad;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NREG; ++i) s[i & 3] += v[i];
    const float mean = seg_sum<SEG>(((s[0] + s[1]) + (s[2] + s[3])) - fp * vl) / (float)n;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
        const float d = v[i] - mean;
        q[i & 3] = fmaf(d, d, q[i & 3]);
    }
    const float dl = vl - mean;
    return seg_sum<SEG>(((q[0] + q[1]) + (q[2] + q[3])) - fp * dl * dl) / (float)n;
}

// Fold one spatial axis (n values at element stride `fstride`): cell = row * inner + a reads
// t[base + row * row_stride + a + e * fstride], e < n.   out (cells, 3) = (max, median, variance) in dB
//   RA: cell = (d, r, a), inner = A,     row_stride = E*A,   fstride = A   (fold elevation)
//   EA: cell = (d, e*A + a), inner = E*A, row_stride = R*E*A, fstride = E*A (fold the cropped range)
// SEG = 1: one lane per column, 64 columns per wave.  SEG = 2 (columns of more than 128 values: two lanes' registers):
// lanes l and l + 32 share a column, 32 columns per wave; c0 = ceil(n / 2): lane l holds elements [0, c0), lane l + 32
// the LAST c0 elements [n - c0, n) -- for odd n the middle element is held twice and counts as one more surplus copy
// of lane l's last element.
template <int NREG, int SEG>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NREG > 64 ? 2 : 3))) void radar_fold_kernel(const float* __restrict__ t, float* __restrict__ out, unsigned cells,
                                                        unsigned inner, size_t row_stride, size_t fstride, size_t base, int n,
                                                        unsigned total_bytes) {
    constexpr int COLS = 64 / SEG;
    const int half = SEG == 2 ? (int)threadIdx.x >> 5 : 0;
    const unsigned cell = blockIdx.x * COLS + (threadIdx.x & (COLS - 1));
    const bool ok = cell < cells;
    const unsigned cc = ok ? cell : cells - 1;
    const unsigned row = cc / inner;
    // buffer loads: ONE vector register of addressing per lane (its cell's byte offset), the element stride rides in the
    // scalar offset -- 64-bit global addresses would cost two registers per load in flight, i.e. more than the data
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(t), 0, total_bytes, 0x00020000);
    const unsigned sstep = (unsigned)(fstride * 4);
    const int c0 = (n + SEG - 1) / SEG;
    const unsigned voff = (unsigned)((base + (size_t)row * row_stride + (size_t)(cc - row * inner)) * 4) + half * (n - c0) * sstep;
    const int npad = NREG - c0 + ((SEG == 2 && half == 0) ? (n & 1) : 0);
    float v[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i)
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)min(i, c0 - 1) * sstep, 0));
#pragma unroll
    for (int i = 0; i < NREG; ++i) v[i] = to_db(v[i]);
    float mn, mx;
    minmax_regs<NREG, SEG>(v, mn, mx);
    const float var = var_regs<NREG, SEG>(v, npad, n);
    const float med = median_regs<NREG, SEG>(v, npad, n, mn, mx);
    if (ok && half == 0) {
        float* o = out + (size_t)cell * 3;
        o[0] = mx; o[1] = med; o[2] = var;
    }
}

// Fold the doppler axis of the two (D, cells, 3) scratches in one launch: blockIdx.x < blocks_ra -> RA cells, else EA cells;
// blockIdx.y = which statistic of the per-doppler triples is folded:
//   0 peaks   -> feats[0] = max, feats[3] = raster[first argmax], feats[4] = median (RA) | mean (EA), feats[5] = variance
//   1 medians -> feats[1] = median          2 variances -> feats[2] = variance
struct FinishArgs {
    const float* scr[2];
    float* feats[2];
    int64_t cells[2];
    int blocks_ra;
    const float* raster;
    int D;
};

__global__ __launch_bounds__(64) void radar_finish_kernel(FinishArgs a) {
    constexpr int NREG = 64;
    const int w = (int)blockIdx.x >= a.blocks_ra;
    const int64_t cells = a.cells[w];
    const int64_t cell = (int64_t)(blockIdx.x - (w ? a.blocks_ra : 0)) * 64 + threadIdx.x;
    const bool ok = cell < cells;
    const int64_t cc = ok ? cell : cells - 1;
    const int ch = blockIdx.y, D = a.D;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.scr[w]), 0, (unsigned)(cells * 3 * 4 * D), 0x00020000);
    const unsigned voff = (unsigned)((cc * 3 + ch) * 4), sstep = (unsigned)(cells * 3 * 4);
    float v[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i)
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)min(i, D - 1) * sstep, 0));
    const int npad = NREG - D;                     // v[D ..] = copies of v[D - 1] (clamped load index)
    float* o = a.feats[w] + cell * 6;
    if (ch == 2) {
        const float var = var_regs<NREG, 1>(v, npad, D);
        if (ok) o[2] = var;
        return;
    }
    float mn, mx;
    minmax_regs<NREG, 1>(v, mn, mx);
    if (ch == 1) {
        const float med = median_regs<NREG, 1>(v, npad, D, mn, mx);
        if (ok) o[1] = med;
        return;
    }
    int arg = 0;
#pragma unroll
    for (int i = NREG - 1; i >= 0; --i) arg = v[i] == mx ? i : arg;      // first argmax; the copies sit behind the elements
    float centre;
    if (w) {                      // EA quirk: the doppler centre is a mean (numpy sums in order; fp64 keeps the last bits)
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NREG; ++i) s += i < D ? (double)v[i] : 0.0;      // D is uniform: scalar selects
        centre = (float)(s / D);
    } else {
        centre = median_regs<NREG, 1>(v, npad, D, mn, mx);
    }
    const float var = var_regs<NREG, 1>(v, npad, D);
    if (ok) { o[0] = mx; o[3] = a.raster[arg]; o[4] = centre; o[5] = var; }
}

static int fold_axis(const float* t, float* out, int64_t cells, int64_t inner, size_t row_stride, size_t fstride, size_t base,
                     int n, unsigned total_bytes, hipStream_t st) {
    DPFT_REQUIRE(n >= 1 && n <= 256, "radar_projection: a folded axis of %d elements is outside [1, 256]", n);
    DPFT_REQUIRE(cells < ((int64_t)1 << 31), "radar_projection: too many cells");
#define FOLD(NR, SEG)                                                                                                     \
    hipLaunchKernelGGL((radar_fold_kernel<NR, SEG>), dim3((unsigned)cdiv(cells, (int64_t)(64 / SEG))), dim3(64), 0, st, t, out, \
                       (unsigned)cells, (unsigned)inner, row_stride, fstride, base, n, total_bytes)
    if (n <= 16) FOLD(16, 1);
    else if (n <= 40) FOLD(40, 1);
    else if (n <= 64) FOLD(64, 1);
    else if (n <= 128) FOLD(128, 1);
    else FOLD(128, 2);
#undef FOLD
    return check_launch("radar_fold");
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
    DPFT_REQUIRE((int64_t)D * R * E * A * 4 < ((int64_t)1 << 32), "radar_projection: cubes of 4 GB and more are not supported");
    const unsigned total_bytes = (unsigned)((int64_t)D * R * E * A * 4);
    hipStream_t st = (hipStream_t)stream;
    float* sra = scratch;
    float* sea = scratch + (size_t)D * R * A * 3;
    const size_t EA = (size_t)E * A, REA = (size_t)R * EA;
    // RA: cells (d, r, a), folded axis = elevation (stride A)
    int rc = fold_axis(tesseract, sra, (int64_t)D * R * A, A, EA, (size_t)A, 0, E, total_bytes, st);
    if (rc) return rc;
    // EA: cells (d, e, a), folded axis = cropped range (stride E*A)
    rc = fold_axis(tesseract, sea, (int64_t)D * E * A, (int64_t)EA, REA, EA, (size_t)r_lo * EA, r_hi - r_lo, total_bytes, st);
    if (rc) return rc;
    FinishArgs f;
    f.scr[0] = sra; f.scr[1] = sea; f.feats[0] = ra; f.feats[1] = ea;
    f.cells[0] = (int64_t)R * A; f.cells[1] = (int64_t)E * A;
    f.blocks_ra = (int)cdiv(f.cells[0], (int64_t)64);
    f.raster = doppler_raster; f.D = D;
    hipLaunchKernelGGL(radar_finish_kernel, dim3(f.blocks_ra + (int)cdiv(f.cells[1], (int64_t)64), 3), dim3(64), 0, st, f);
    return check_launch("radar_finish");
}
