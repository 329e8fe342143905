// Probe: issue rate of the vector instructions a 3 x bf16 operand split can be built from (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8]; unsigned u[8];
    for (int q = 0; q < 8; ++q) { a[q] = threadIdx.x * 0.37f + q; u[q] = threadIdx.x * 2654435761u + q; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                if (OP == 0) { bf16x2 p = __builtin_convertvector(f32x2{a[q], a[q + 1]}, bf16x2); u[q] ^= __builtin_bit_cast(unsigned, p); a[q] += 1.0f; }   // cvt_pk + xor + add
                if (OP == 1) { u[q] = (u[q] & 0xffff0000u) ^ u[q + 1]; a[q] += 1.0f; }                                                                      // and + xor + add
                if (OP == 2) { u[q] = __builtin_amdgcn_perm(u[q], u[q + 1], 0x07060302u); a[q] += 1.0f; }                                                     // perm + add
                if (OP == 3) { a[q] = a[q] - a[q + 1]; a[q + 1] += 1.0f; }                                                                                    // sub + add
            }
    }
    float s = 0; for (int q = 0; q < 8; ++q) s += a[q] + (float)u[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> static void run(float* out, const char* what, int nops) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<256, 256>>>(out, 10);
    (void)hipEventRecord(e0); k<OP><<<256, 256>>>(out, 4000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.3f ms  %5.2f cycles per loop body of %d instructions (2.4 GHz)\n", what, ms, ms * 1e-3 * 2.4e9 / (4000.0 * 32), nops);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<0>(out, "cvt_pk_bf16_f32 + xor + add", 3);
    run<1>(out, "and + xor + add", 3);
    run<2>(out, "perm + add", 2);
    run<3>(out, "sub + add", 2);
    return 0;
}
