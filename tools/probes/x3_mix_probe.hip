// Probe for the 3 x bf16 split GEMM loop (conv_x3.h): what one wave per SIMD can hide behind v_mfma_f32_32x32x16_bf16.
//   A: dependent-accumulator chains -- NACC independent accumulators, MFMAs round-robin over them
//   B: 4 accumulators + the fp32 -> three bf16 planes split of Q register quads per 24 MFMAs (+ 3 ds_write_b64 per quad)
//   C: B + 12 ds_read_b128 fragment reads per 24 MFMAs
// Prints cycles per MFMA at a nominal 2.4 GHz (one 4-wave workgroup per CU, 256 CUs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NACC, int Q, int NRD>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[49152];
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    bf16x8 fr[6];
    for (int t = 0; t < 6; ++t) for (int e = 0; e < 8; ++e) fr[t][e] = (__bf16)(threadIdx.x * 0.001f + e + t);
    f32x4 x[Q > 0 ? Q : 1];
    for (int q = 0; q < (Q > 0 ? Q : 1); ++q) x[q] = f32x4{threadIdx.x * 1.1f, q * 0.3f, 1.7f, -2.1f};
    char* wp = lds + threadIdx.x * 8;
    const char* rp = lds + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        if constexpr (NRD > 0) {
#pragma unroll
            for (int r = 0; r < NRD && r < 6; ++r) fr[r] = *reinterpret_cast<const bf16x8*>(rp + r * 1024 + (it & 1) * 8192);
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % 6], fr[(m + 1) % 6], acc[m % NACC], 0, 0, 0);
            if constexpr (NRD > 6) {
                if (m >= 12 && m < 12 + NRD - 6) fr[(m - 12) % 6] = *reinterpret_cast<const bf16x8*>(rp + (m - 6) * 1024 + (it & 1) * 8192);
            }
            if constexpr (Q > 0) {
                constexpr int SP = 24 / Q;
                if (m % SP == SP - 1) {
                    const int q = m / SP;
                    f32x4 v = x[q];
                    asm volatile("" : "+v"(v));
                    const bf16x4 t1 = __builtin_convertvector(v, bf16x4);
                    const f32x4 r1 = v - __builtin_convertvector(t1, f32x4);
                    const bf16x4 t2 = __builtin_convertvector(r1, bf16x4);
                    const f32x4 r2 = r1 - __builtin_convertvector(t2, f32x4);
                    const bf16x4 t3 = __builtin_convertvector(r2, bf16x4);
                    *reinterpret_cast<bf16x4*>(wp + 16384 + q * 2048) = t1;
                    *reinterpret_cast<bf16x4*>(wp + 16384 + 16384 + q * 2048) = t2;
                    *reinterpret_cast<bf16x4*>(wp + 16384 + 32768 - 4096 + q * 2048) = t3;
                }
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) for (int e = 0; e < 16; ++e) s += acc[q][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int Q, int NRD>
static void run(float* out, const char* what) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC, Q, NRD><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    probe<NACC, Q, NRD><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s NACC=%d quads/24=%2d reads/24=%2d  %8.3f ms  %6.1f cycles per MFMA (2.4 GHz)  %6.1f TF-eq (6 products)\n", what, NACC, Q, NRD, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 24),
           256.0 * 4 * iters * 24 * 32768.0 / 6.0 / (ms * 1e-3) * 1e-12);
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    run<1, 0, 0>(out, "chain, one accumulator");
    run<2, 0, 0>(out, "two accumulators");
    run<3, 0, 0>(out, "three accumulators");
    run<4, 0, 0>(out, "four accumulators");
    run<8, 0, 0>(out, "eight accumulators");
    run<4, 2, 0>(out, "4 acc + split");
    run<4, 4, 0>(out, "4 acc + split");
    run<4, 6, 0>(out, "4 acc + split");
    run<4, 8, 0>(out, "4 acc + split");
    run<4, 12, 0>(out, "4 acc + split");
    run<8, 4, 12>(out, "8 acc + split + fragment reads");
    run<8, 8, 12>(out, "8 acc + split + fragment reads");
    run<8, 4, 6>(out, "8 acc + split + fragment reads");
    run<3, 2, 9>(out, "3 acc + split + fragment reads");
    return 0;
}
