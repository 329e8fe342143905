// Stand-in for the local footprint of an N = 8 ring all-reduce kernel (tools/exp_switches.py --standin-collective): RCCL is not
// available to more than one rank per GPU here, so the cost model of DESIGN section 6 ("the all-reduce kernels are work: CUs and
// HBM bandwidth next to the backward") is tested with a kernel of the same shape -- `blocks` workgroups (RCCL: 16-32 per ring
// channel set) that read and write the bucket `passes` times (a ring moves each byte 2 (N - 1) / N ~ 1.75 times through the
// rank's memory in each direction), in place and value-preserving (x = x * 1).
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(256) void standin_kernel(float4* buf, int64_t n4, int passes) {
    for (int p = 0; p < passes; ++p)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            float4 v = buf[i];
            v.x *= 1.0f; v.y *= 1.0f; v.z *= 1.0f; v.w *= 1.0f;
            buf[i] = v;
        }
}
extern "C" int standin_launch(void* buf, int64_t bytes, int passes, int blocks, void* stream) {
    hipLaunchKernelGGL(standin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)buf, bytes / 16, passes);
    return (int)hipGetLastError();
}
