// Probe: is the scalar soffset of a raw buffer load part of the hardware range check on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* out, int n, int voff, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    if (threadIdx.x == 0) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3]; }
}
int main() {
    const int n = 1024;
    float *x, *out, h[2 * n], o[4];
    for (int i = 0; i < 2 * n; ++i) h[i] = 1.0f + i;
    (void)hipMalloc(&x, 2 * n * 4); (void)hipMalloc(&out, 16);
    (void)hipMemcpy(x, h, 2 * n * 4, hipMemcpyHostToDevice);
    struct { int v, s; const char* what; } cases[] = {
        {16, 0, "in range (voffset)"}, {0, 16, "in range (soffset)"}, {n * 4, 0, "voffset past num_records"},
        {0, n * 4, "soffset past num_records"}, {n * 4 - 32, 64, "voffset in range, voffset+soffset past"},
        {(int)0x80000000u, 0, "voffset 0x80000000"}, {(int)0x80000000u, 4096, "voffset 0x80000000 + soffset"},
        {n * 4 - 8, 0, "straddles the end (last 8 bytes in range)"}};
    for (auto& c : cases) {
        k<<<1, 64>>>(x, out, n, c.v, c.s);
        (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
        printf("%-45s -> %g %g %g %g\n", c.what, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
