// ds_read_b64_tr_b16 semantics probe: every lane supplies its own 8-byte LDS address; which lane's data ends up where?
// hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
typedef __attribute__((address_space(3))) bf16x4v lds_bf16x4;
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = 0xffff;
    __syncthreads();
    // lane l owns the four shorts at a scattered address: value = lane * 4 + sub
    const int lane = threadIdx.x;
    const int slot = (lane * 37) % 64;                     // scattered 8-byte slots, 64-byte stride
    for (int s = 0; s < 4; ++s) lds[slot * 32 + s] = (unsigned short)(lane * 4 + s);
    __syncthreads();
    bf16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&lds[slot * 32]));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int s = 0; s < 4; ++s) out[lane * 4 + s] = r[s];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int src_lane = h[l * 4 + j] / 4, sub = h[l * 4 + j] % 4;
            printf("  (lane %2d, sub %d)", src_lane, sub);
            const int g = l & ~15, i = l & 15;
            if (src_lane != g + 4 * j + (i >> 2) || sub != (i & 3)) ++bad;
        }
        printf("\n");
    }
    printf("hypothesis result[lane i][j] = sub (i & 3) of lane (4 j + i / 4) of the 16-lane group: %s (%d mismatches)\n", bad ? "WRONG" : "holds", bad);
    return 0;
}
