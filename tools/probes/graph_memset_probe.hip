// Probe: are hipMemsetAsync nodes of a stream-captured hipGraph ordered against the kernels of the PREVIOUS launch of the same
// graph on the same stream?  (Round 3: launch plans replayed with memset nodes gave intermittent wrong gradients from about the
// 7th replay on; a device sync around the launch hid it; with zero_fill_kernel nodes the problem never showed.)
// Graph:  [memset buf <- 0]  ->  slow kernel: buf[i] += 1 after a spin  ->  kernel: out[rep][i] = buf[i]
// Variants: memset as the ROOT node / behind a dummy kernel; one stream / a forked side stream joined before the read.
// Expectation: every out[rep][i] == 1.  A memset of launch n+1 that overtakes launch n's kernels shows up as 0 (or 2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_add(float* buf, int n, long long spin) {
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] += 1.f;
}
__global__ void copy_out(const float* buf, float* out, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = buf[i];
}
__global__ void dummy(float* p) { if (threadIdx.x == 1000) p[0] = 0.f; }
__global__ void zero_k(float* buf, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = 0.f;
}

static int run(const char* name, bool use_memset, bool root, bool fork, int n, int reps, long long spin) {
    float *buf, *out, *scratch;
    CK(hipMalloc(&buf, n * sizeof(float)));
    CK(hipMalloc(&out, (size_t)n * reps * sizeof(float)));
    CK(hipMalloc(&scratch, 64));
    CK(hipMemset(buf, 0, n * sizeof(float)));
    hipStream_t st, side;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    std::vector<hipGraphExec_t> execs(reps);
    for (int r = 0; r < reps; ++r) {      // one graph per repetition slot (the output pointer differs); same buf
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        if (!root) dummy<<<1, 64, 0, st>>>(scratch);
        if (use_memset) CK(hipMemsetAsync(buf, 0, n * sizeof(float), st));
        else zero_k<<<64, 256, 0, st>>>(buf, n);
        if (fork) {
            CK(hipEventRecord(e0, st));
            CK(hipStreamWaitEvent(side, e0, 0));
            spin_add<<<64, 256, 0, side>>>(buf, n, spin);
            CK(hipEventRecord(e1, side));
            CK(hipStreamWaitEvent(st, e1, 0));
        } else {
            spin_add<<<64, 256, 0, st>>>(buf, n, spin);
        }
        copy_out<<<64, 256, 0, st>>>(buf, out + (size_t)r * n, n);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&execs[r], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    int bad_total = 0;
    for (int round = 0; round < 20; ++round) {
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(execs[r], st));      // back to back, no sync in between
        CK(hipStreamSynchronize(st));
        std::vector<float> h((size_t)n * reps);
        CK(hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        int bad = 0;
        for (size_t i = 0; i < h.size(); ++i) bad += h[i] != 1.f;
        bad_total += bad;
    }
    printf("%-52s n=%7d reps=%3d: %d wrong values in 20 rounds\n", name, n, reps, bad_total);
    for (auto e : execs) (void)hipGraphExecDestroy(e);
    (void)hipFree(buf); (void)hipFree(out); (void)hipFree(scratch);
    return 0;
}

__global__ void check_count(const float* buf, unsigned* wrong, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (buf[i] != 1.f) atomicAdd(wrong, 1u);
}

// the plans' situation: the SAME executable graph relaunched back to back, several such graphs on their own streams at once
// (three view encoders + the decoder), small fills (the BatchNorm sums: 16 KiB) between atomically-adding kernels
static int run_concurrent(const char* name, bool use_memset, int n, int streams, int launches, long long spin) {
    std::vector<hipStream_t> st(streams);
    std::vector<hipGraphExec_t> ex(streams);
    std::vector<float*> buf(streams);
    unsigned* wrong;
    CK(hipMalloc(&wrong, sizeof(unsigned)));
    CK(hipMemset(wrong, 0, sizeof(unsigned)));
    for (int s = 0; s < streams; ++s) {
        CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        CK(hipMalloc(&buf[s], n * sizeof(float)));
        hipGraph_t g;
        CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
        for (int layer = 0; layer < 6; ++layer) {      // fill -> add -> check, six times per graph
            if (use_memset) CK(hipMemsetAsync(buf[s], 0, n * sizeof(float), st[s]));
            else zero_k<<<8, 256, 0, st[s]>>>(buf[s], n);
            spin_add<<<8 + 8 * s, 256, 0, st[s]>>>(buf[s], n, spin * (1 + layer % 3));
            check_count<<<8, 256, 0, st[s]>>>(buf[s], wrong, n);
        }
        CK(hipStreamEndCapture(st[s], &g));
        CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    for (int l = 0; l < launches; ++l)
        for (int s = 0; s < streams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
    CK(hipDeviceSynchronize());
    unsigned h = 0;
    CK(hipMemcpy(&h, wrong, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-52s n=%7d streams=%d launches=%4d: %u wrong values\n", name, n, streams, launches, h);
    return 0;
}

int main() {
    const long long spin = 20000;      // ~10 us
    for (int n : {4096, 1 << 20}) {
        run("memset node, root, one stream", true, true, false, n, 32, spin);
        run("memset node, behind a kernel, one stream", true, false, false, n, 32, spin);
        run("memset node, root, forked side stream", true, true, true, n, 32, spin);
        run("memset node, behind a kernel, forked side stream", true, false, true, n, 32, spin);
        run("zero kernel node, root, one stream", false, true, false, n, 32, spin);
        run("zero kernel node, root, forked side stream", false, true, true, n, 32, spin);
    }
    run_concurrent("same exec relaunched, memset nodes, 4 streams", true, 4096, 4, 400, 4000);
    run_concurrent("same exec relaunched, memset nodes, 4 streams", true, 1 << 18, 4, 400, 4000);
    run_concurrent("same exec relaunched, kernel fills, 4 streams", false, 4096, 4, 400, 4000);
    run_concurrent("same exec relaunched, memset nodes, 8 streams", true, 4096, 8, 400, 2000);
    return 0;
}
