// Fused multi-tensor AdamW (torch.optim.AdamW semantics, decoupled weight decay, no amsgrad) -- one launch
// for all ~1650 parameter tensors instead of the ~10 ms of foreach kernels the reference's
// `getattr(torch.optim, name)` optimizer (src/dprt/training/optimizer.py:6-7) costs per step at 90 M parameters.
#include "common.h"

namespace dpft {

struct AdamChunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    int32_t n;        // elements in this chunk
    int32_t tensor;   // index into `active`
};

// `skipped[t]` = number of optimizer steps tensor t sat out (no gradient): its own step count is `step - skipped[t]`,
// like the per-parameter `state["step"]` of torch.optim.AdamW.  Only inactive blocks write it (the block of a tensor's
// first chunk), only active blocks read it, so there is no race inside a launch.
// `gate` (round 5, may be null): the step's loss on the device.  The reference steps only `if loss > 0` (training/trainer.py:131);
// the trainer launches backward and optimizer without reading the loss back, and a loss that is not positive closes the gate
// here: every tensor sits the step out exactly as if it had no gradient.
__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ chunks, const int32_t* __restrict__ active,
                                                     int32_t* __restrict__ skipped, int32_t step,
                                                     float lr, float beta1, float beta2, float eps, float decay,
                                                     float step_size, float inv_sqrt_bc2, const float* __restrict__ gate) {
    const AdamChunk c = chunks[blockIdx.x];
    const bool closed = gate != nullptr && !(gate[0] > 0.f);
    if (closed || (active && !active[c.tensor])) {             // parameters without a gradient are skipped (grad is None)
        if (skipped && c.m == nullptr && threadIdx.x == 0) skipped[c.tensor] += 1;   // marker row: one per tensor
        return;
    }
    if (c.m == nullptr) return;                    // marker row of an active tensor
    if (skipped) {
        const int32_t own = step - skipped[c.tensor];
        if (own != step) {                         // bias corrections of this tensor's own step count
            const double bc1 = 1.0 - pow((double)beta1, (double)own), bc2 = 1.0 - pow((double)beta2, (double)own);
            step_size = (float)((double)lr / bc1);
            inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        }
    }
    for (int i = threadIdx.x * 4; i < c.n; i += 256 * 4) {
        if (i + 3 < c.n && ((((uintptr_t)(c.p + i)) | ((uintptr_t)(c.g + i))) & 15) == 0) {
            f32x4 p = *reinterpret_cast<f32x4*>(c.p + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(c.g + i);
            f32x4 m = *reinterpret_cast<f32x4*>(c.m + i);
            f32x4 v = *reinterpret_cast<f32x4*>(c.v + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p[e] *= decay;
                m[e] = m[e] + (g[e] - m[e]) * (1.f - beta1);           // exp_avg.lerp_(grad, 1 - beta1)
                v[e] = v[e] * beta2 + (1.f - beta2) * g[e] * g[e];
                const float denom = sqrtf(v[e]) * inv_sqrt_bc2 + eps;
                p[e] -= step_size * (m[e] / denom);
            }
            *reinterpret_cast<f32x4*>(c.p + i) = p;
            *reinterpret_cast<f32x4*>(c.m + i) = m;
            *reinterpret_cast<f32x4*>(c.v + i) = v;
        } else {
            for (int e = i; e < min(i + 4, c.n); ++e) {
                float p = c.p[e] * decay;
                const float g = c.g[e];
                const float m = c.m[e] + (g - c.m[e]) * (1.f - beta1);
                const float v = c.v[e] * beta2 + (1.f - beta2) * g * g;
                p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
                c.p[e] = p; c.m[e] = m; c.v[e] = v;
            }
        }
    }
}

}  // namespace dpft

using namespace dpft;

extern "C" int dpft_adamw_f32(const void* chunks, int32_t n_chunks, const int32_t* active, int32_t* skipped, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int32_t step, const float* gate,
                              dpft_stream_t stream) {
    DPFT_REQUIRE(chunks && n_chunks > 0 && step >= 1, "adamw: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)chunks, active, skipped, step, lr,
                       beta1, beta2, eps, (float)(1.0 - (double)lr * weight_decay), (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), gate);
    return check_launch("adamw");
}
