// K-Radar export selection: KRadarExporter._construct_objects
// (src/dprt/evaluation/exporters/kradar.py:231-294) for every sample of a batch and every confidence threshold in ONE
// launch.  The reference runs ~30 tiny tensor ops and a host sync per (sample, threshold); here a block per sample
// walks its N candidates once, evaluates  cls_mask & conf_mask & fov_mask  and writes the surviving objects, in
// candidate order (what boolean-mask indexing gives), as compact 8-float rows per threshold.
#include "common.h"

namespace dpft {

constexpr int kExportMaxThr = 8;

struct ExportArgs {
    const float *cls, *center, *size, *angle;
    float thr[kExportMaxThr];
    float* rows;          // (B, T, N, 8): category, h, w, l, y, z, x, theta  (kradar.py:283-292 column order)
    int32_t* counts;      // (B, T)
    uint8_t* mask;        // (B, N) bit t = selected at threshold t (may be null)
    int B, N, C, T;
};

__global__ __launch_bounds__(256) void export_select_kernel(ExportArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wave_cnt[kExportMaxThr][4];
    __shared__ int base[kExportMaxThr];
    if (tid < kExportMaxThr) base[tid] = 0;
    __syncthreads();
    for (int n0 = 0; n0 < a.N; n0 += 256) {
        const int n = n0 + tid;
        const bool in = n < a.N;
        float conf = 0.f, yaw = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
        int cat = -1;
        bool fov = false;
        if (in) {
            const float* c = a.cls + ((int64_t)b * a.N + n) * a.C;
            conf = c[0];
            int arg = 0;
            for (int k = 1; k < a.C; ++k) {            // torch.max(dim=-1): first maximal entry, NaN wins (:259)
                const float v = c[k];
                if (v > conf || (v != v && conf == conf)) { conf = v; arg = k; }
            }
            cat = arg - 1;                             // background (index 0) -> -1 (:265)
            const float* an = a.angle + ((int64_t)b * a.N + n) * 2;
            yaw = atan2f(an[0], an[1]);                // (:262)
            const float* ce = a.center + ((int64_t)b * a.N + n) * 3;
            cx = ce[0]; cy = ce[1]; cz = ce[2];
            fov = (0.f < cx) && (cx < 72.f) && (-6.4f < cy) && (cy < 6.4f) && (-2.0f < cz) && (cz < 6.0f) &&
                  (-50.0f < yaw) && (yaw < 50.0f);     // (:268-272); the yaw bound is in radians there too
        }
        uint32_t bits = 0;
        for (int t = 0; t < a.T; ++t) {
            const bool sel = in && fov && (cat >= 0) && (conf >= a.thr[t]);      // (:275-277)
            const uint64_t bal = __ballot(sel);
            if (sel) bits |= 1u << t;
            if (lane == 0) wave_cnt[t][wave] = __popcll(bal);
            __syncthreads();
            int off = base[t];
            for (int w = 0; w < wave; ++w) off += wave_cnt[t][w];
            off += __popcll(bal & ((1ull << lane) - 1ull));
            if (sel) {
                float* r = a.rows + (((int64_t)b * a.T + t) * a.N + off) * 8;
                const float* sz = a.size + ((int64_t)b * a.N + n) * 3;
                r[0] = (float)cat; r[1] = sz[2]; r[2] = sz[1]; r[3] = sz[0];
                r[4] = cy; r[5] = cz; r[6] = cx; r[7] = yaw;
            }
            __syncthreads();
            if (tid == 0) base[t] += wave_cnt[t][0] + wave_cnt[t][1] + wave_cnt[t][2] + wave_cnt[t][3];
            __syncthreads();
        }
        if (in && a.mask) a.mask[(int64_t)b * a.N + n] = (uint8_t)bits;
    }
    if (tid < a.T) a.counts[b * a.T + tid] = base[tid];
}

}  // namespace dpft

extern "C" int dpft_export_select_f32(const float* cls, const float* center, const float* size, const float* angle,
                                      const float* conf_thrs, int32_t T, float* rows, int32_t* counts, uint8_t* mask,
                                      int32_t B, int32_t N, int32_t C, dpft_stream_t stream) {
    DPFT_REQUIRE(cls && center && size && angle && conf_thrs && rows && counts, "export_select: null argument");
    DPFT_REQUIRE(B >= 0 && N >= 0 && C >= 1, "export_select: bad shape B=%d N=%d C=%d", B, N, C);
    DPFT_REQUIRE(T >= 1 && T <= dpft::kExportMaxThr, "export_select: 1..%d thresholds, got %d", dpft::kExportMaxThr, T);
    if (B == 0) return DPFT_OK;
    dpft::ExportArgs a;
    a.cls = cls; a.center = center; a.size = size; a.angle = angle;
    for (int t = 0; t < dpft::kExportMaxThr; ++t) a.thr[t] = t < T ? conf_thrs[t] : 0.f;      // host array
    a.rows = rows; a.counts = counts; a.mask = mask;
    a.B = B; a.N = N; a.C = C; a.T = T;
    hipLaunchKernelGGL(dpft::export_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    return dpft::check_launch("export_select");
}
