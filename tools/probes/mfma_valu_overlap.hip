// Probe: can a wave's own VALU instructions run in the shadow of its MFMAs?  One wave per SIMD (256 threads per CU),
// fp32 32x32x2 MFMAs (64 cycles each) with NV independent-ish integer VALU ops either placed after every MFMA
// ("fine"), after every 4th MFMA in a block of 4*NV ("coarse"), or absent.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV, bool BF = false>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f;
    bf16x8 ab; for (int e = 0; e < 8; ++e) ab[e] = (__bf16)(a + e);
    unsigned v[8];
    for (int q = 0; q < 8; ++q) v[q] = seed + threadIdx.x * (q + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE != 2) { if (BF) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, acc[q], 0, 0, 0); else acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0); }
                if (MODE == 1 || MODE == 2) {
#pragma unroll
                    for (int n = 0; n < NV; ++n) v[n & 7] = v[n & 7] * 1664525u + v[(n + 1) & 7];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 3) {
#pragma unroll
                for (int n = 0; n < 4 * NV; ++n) v[n & 7] = v[n & 7] * 1664525u + v[(n + 1) & 7];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
    unsigned u = 0;
    for (int q = 0; q < 8; ++q) u ^= v[q];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)u;
}

static int g_blocks = 256;
template <int MODE, int NV, bool BF = false>
static float run(float* out, const char* what) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NV, BF><<<g_blocks, 256>>>(out, 10, 1u);
    hipEventRecord(e0);
    probe<MODE, NV, BF><<<g_blocks, 256>>>(out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)iters * 32 * (g_blocks / 256);
    printf("%-34s NV=%2d  %8.3f ms  %7.1f cycles per MFMA slot (at 2.4 GHz)\n", what, NV, ms, ms * 1e-3 * 2.4e9 / mfma);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    for (int wpc = 1; wpc <= 4; wpc *= 2) {
    g_blocks = 256 * wpc;
    printf("---- %d workgroup(s) of 4 waves per CU: cycles per MFMA slot are per SIMD (all its waves together)\n", wpc);
    run<0, 0>(out, "MFMA only");
    run<2, 4>(out, "VALU only");
    run<1, 4>(out, "MFMA + VALU after each MFMA");
    run<3, 4>(out, "MFMA x4 then VALU x4*NV");
    run<2, 8>(out, "VALU only");
    run<1, 8>(out, "MFMA + VALU after each MFMA");
    run<3, 8>(out, "MFMA x4 then VALU x4*NV");
    run<2, 12>(out, "VALU only");
    run<1, 12>(out, "MFMA + VALU after each MFMA");
    run<3, 12>(out, "MFMA x4 then VALU x4*NV");
    run<0, 0, true>(out, "bf16 32x32x16 MFMA only");
    run<1, 2, true>(out, "bf16 MFMA + VALU after each MFMA");
    run<1, 4, true>(out, "bf16 MFMA + VALU after each MFMA");
    run<1, 8, true>(out, "bf16 MFMA + VALU after each MFMA");
    }
    return 0;
}
