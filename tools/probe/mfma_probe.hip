// Ablation probe for the conv K-loop structure (tuning aid; not part of the product library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 36;

// VAR 0: pure MFMA. 1: + LDS fragment reads. 2: + two barriers + LDS stores per K-step. 3: + global loads.
template <int VAR, int RB, int CB>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int ksteps, int stride) {
    constexpr int BM = RB * 64, BN = CB * 64;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[RB][CB];
    for (int i = 0; i < RB; ++i) for (int j = 0; j < CB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int i = tid; i < (BM + BN) * LDK; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const float* a_frag = smem + (wm * RB * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* b_frag = smem + BM * LDK + (wn * CB * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const int chunk = tid & 7, rowl = tid >> 3;
    const float* gp = g + ((size_t)blockIdx.x * 64 + rowl) * stride + chunk * 4;
    f32x4 ra[BM / 32], rb[BN / 32];
    for (int i = 0; i < BM / 32; ++i) ra[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < BN / 32; ++i) rb[i] = f32x4{0, 0, 0, 0};
    f32x4 af[RB], bf[CB];
    for (int i = 0; i < RB; ++i) af[i] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    for (int j = 0; j < CB; ++j) bf[j] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    if constexpr (VAR >= 4) {
        // deeper register prefetch: DEPTH register sets in flight, consumed round-robin
        constexpr int DEPTH = (VAR >= 4) ? VAR - 2 : 1;   // 4 -> 2 sets, 5 -> 3 sets
        f32x4 pa[DEPTH][BM / 32], pb[DEPTH][BN / 32];
        auto issue = [&](int slot, int kt) {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) pa[slot][i] = *reinterpret_cast<const f32x4*>(gp + (size_t)(i * 32) * stride + kt * 32);
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) pb[slot][i] = *reinterpret_cast<const f32x4*>(gp + (size_t)(i * 32 + 7) * stride + kt * 32);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) issue(d, d);
        for (int kt0 = 0; kt0 < ksteps; kt0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int kt = kt0 + d;
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
                    for (int i = 0; i < RB; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDK + kg * 8);
#pragma unroll
                    for (int j = 0; j < CB; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDK + kg * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < RB; ++i)
#pragma unroll
                            for (int j = 0; j < CB; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < BM / 32; ++i) *reinterpret_cast<f32x4*>(&smem[(rowl + 32 * i) * LDK + chunk * 4]) = pa[d][i];
#pragma unroll
                for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<f32x4*>(&smem[BM * LDK + (rowl + 32 * i) * LDK + chunk * 4]) = pb[d][i];
                if (kt + DEPTH < ksteps) issue(d, kt + DEPTH);
                __syncthreads();
            }
        }
    } else
    for (int kt = 0; kt < ksteps; ++kt) {
        if (VAR >= 3) {
            for (int i = 0; i < BM / 32; ++i) ra[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)(i * 32) * stride + kt * 32);
            for (int i = 0; i < BN / 32; ++i) rb[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)(i * 32 + 7) * stride + kt * 32);
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if (VAR >= 1) {
#pragma unroll
                for (int i = 0; i < RB; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDK + kg * 8);
#pragma unroll
                for (int j = 0; j < CB; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDK + kg * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (VAR >= 2) {
            __syncthreads();
            for (int i = 0; i < BM / 32; ++i) *reinterpret_cast<f32x4*>(&smem[(rowl + 32 * i) * LDK + chunk * 4]) = ra[i];
            for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<f32x4*>(&smem[BM * LDK + (rowl + 32 * i) * LDK + chunk * 4]) = rb[i];
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < RB; ++i) for (int j = 0; j < CB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int VAR, int RB, int CB>
void run(const char* name, int nwg, int ksteps, const float* g, float* out, int stride) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<VAR, RB, CB>), dim3(nwg), dim3(256), 0, 0, g, out, ksteps, stride);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<VAR, RB, CB>), dim3(nwg), dim3(256), 0, 0, g, out, ksteps, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double flop = (double)nwg * 4 * ksteps * 16.0 * RB * CB * (2.0 * 32 * 32 * 2);
    printf("%-28s nwg=%4d ksteps=%3d  %8.1f us  %6.1f TF\n", name, nwg, ksteps, us, flop / us / 1e6);
}

int main() {
    float *g, *out;
    const int stride = 4096;
    const size_t gbytes = (size_t)(1024 * 64 + 512) * stride * sizeof(float);
    hipMalloc(&g, gbytes);
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipMemset(g, 0, gbytes);
    for (int nwg : {256, 512, 1024}) {
        run<0, 2, 2>("128x128 pure mfma", nwg, 72, g, out, stride);
        run<1, 2, 2>("128x128 +lds reads", nwg, 72, g, out, stride);
        run<2, 2, 2>("128x128 +barriers+stores", nwg, 72, g, out, stride);
        run<3, 2, 2>("128x128 +global loads", nwg, 72, g, out, stride);
        run<4, 2, 2>("128x128 prefetch x2", nwg, 72, g, out, stride);
        run<5, 2, 2>("128x128 prefetch x3", nwg, 72, g, out, stride);
        run<0, 1, 1>("64x64 pure mfma", nwg, 72, g, out, stride);
        run<1, 1, 1>("64x64 +lds reads", nwg, 72, g, out, stride);
        run<2, 1, 1>("64x64 +barriers+stores", nwg, 72, g, out, stride);
        run<3, 1, 1>("64x64 +global loads", nwg, 72, g, out, stride);
        run<4, 1, 1>("64x64 prefetch x2", nwg, 72, g, out, stride);
        run<5, 1, 1>("64x64 prefetch x3", nwg, 72, g, out, stride);
    }
    return 0;
}
