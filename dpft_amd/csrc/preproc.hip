// On-device input preprocessing of the K-Radar training pipeline, done per sample on 16 CPU worker
// processes in the reference:
//   * camera: torchvision.transforms.functional.resize on the float HWC frame, bilinear, no antialias
//     (src/dprt/datasets/kradar/dataset.py:319-341) == F.interpolate(mode="bilinear", align_corners=False);
//     the u8 variant takes the decoded JPEG bytes directly (4x fewer bytes over PCIe) and yields the same floats.
//   * radar: (v - min_power) / (max_power - min_power) * 255 clipped to [0, 255] (dataset.py:295-317).
// Pure streaming kernels: one thread per output element, NHWC in and out (the layout the backbones consume).
#include "common.h"

namespace dpft {

template <typename T>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const T* __restrict__ src, float* __restrict__ dst, int B,
                                                               int Hs, int Ws, int Hd, int Wd, int C, float sh,
                                                               float sw) {
    const int64_t total = (int64_t)B * Hd * Wd * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int x = (int)(t % Wd); t /= Wd;
        const int y = (int)(t % Hd);
        const int b = (int)(t / Hd);
        // source index of the area-pixel model (align_corners = False), clamped at 0 like ATen's upsample kernels
        float fy = sh * ((float)y + 0.5f) - 0.5f, fx = sw * ((float)x + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < Hs - 1 ? 1 : 0), x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const T* p = src + (int64_t)b * Hs * Ws * C + c;
        const float v00 = (float)p[((int64_t)y0 * Ws + x0) * C], v01 = (float)p[((int64_t)y0 * Ws + x1) * C];
        const float v10 = (float)p[((int64_t)y1 * Ws + x0) * C], v11 = (float)p[((int64_t)y1 * Ws + x1) * C];
        dst[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

__global__ __launch_bounds__(256) void scale_clip_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                          float in_lo, float in_hi, float out_lo, float out_hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = (x[i] - in_lo) / (in_hi - in_lo) * (out_hi - out_lo) + out_lo;
        y[i] = fminf(fmaxf(v, out_lo), out_hi);
    }
}

static int resize_check(const void* src, const float* dst, int B, int Hs, int Ws, int Hd, int Wd, int C) {
    DPFT_REQUIRE(src && dst, "resize_bilinear: null tensor");
    DPFT_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, "resize_bilinear: non-positive sizes");
    return DPFT_OK;
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_resize_bilinear_nhwc_f32(const float* src, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd,
                                             int32_t Wd, int32_t C, dpft_stream_t stream) {
    int rc = resize_check(src, dst, B, Hs, Ws, Hd, Wd, C);
    if (rc) return rc;
    const int64_t total = (int64_t)B * Hd * Wd * C;
    hipLaunchKernelGGL(resize_bilinear_kernel<float>, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), kNumCU * 16)),
                       dim3(256), 0, (hipStream_t)stream, src, dst, B, Hs, Ws, Hd, Wd, C, (float)Hs / (float)Hd,
                       (float)Ws / (float)Wd);
    return check_launch("resize_bilinear_f32");
}

extern "C" int dpft_resize_bilinear_nhwc_u8(const uint8_t* src, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t Hd,
                                            int32_t Wd, int32_t C, dpft_stream_t stream) {
    int rc = resize_check(src, dst, B, Hs, Ws, Hd, Wd, C);
    if (rc) return rc;
    const int64_t total = (int64_t)B * Hd * Wd * C;
    hipLaunchKernelGGL(resize_bilinear_kernel<uint8_t>, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), kNumCU * 16)),
                       dim3(256), 0, (hipStream_t)stream, src, dst, B, Hs, Ws, Hd, Wd, C, (float)Hs / (float)Hd,
                       (float)Ws / (float)Wd);
    return check_launch("resize_bilinear_u8");
}

extern "C" int dpft_scale_clip_f32(const float* x, float* y, int64_t n, float in_lo, float in_hi, float out_lo,
                                   float out_hi, dpft_stream_t stream) {
    DPFT_REQUIRE(x && y && n > 0, "scale_clip: bad arguments");
    DPFT_REQUIRE(in_hi != in_lo, "scale_clip: empty input range");
    hipLaunchKernelGGL(scale_clip_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), kNumCU * 16)), dim3(256), 0,
                       (hipStream_t)stream, x, y, n, in_lo, in_hi, out_lo, out_hi);
    return check_launch("scale_clip");
}
