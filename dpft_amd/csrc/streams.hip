// Streams on DISTINCT hardware queues.
//
// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams that land on
// the same queue execute strictly in order.  The training step relies on four things running side by side -- the camera
// encoder's chain (main stream), its weight-gradient GEMMs, and the two radar encoders -- so WHICH queue a stream got
// decides whether the overlap exists (tools/probes/stream_queues.py: streams created back to back do collide; the
// step time moved by 2-4 ms with the creation order of unrelated streams).  dpft_stream_set() therefore creates
// candidate streams and keeps one per hardware queue other than the caller's, found by observation: a spin kernel on
// stream A, an event on stream B -- the event completing while A still spins means different queues.
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace dpft {

// spins until the host raises *flag (pinned host memory) -- no assumption about clock rates or API latencies
__global__ void spin_kernel(volatile int* flag) {
    const long long t0 = wall_clock64();      // bounded: ~100 MHz counter, gives up after ~0.2 s whatever happens to the flag
    while (__hip_atomic_load(const_cast<int*>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 &&
           wall_clock64() - t0 < 20000000LL)
        __builtin_amdgcn_s_sleep(32);
}

// true: an event recorded on b completes while a kernel on a is still running (different hardware queues)
static bool runs_beside(hipStream_t a, hipStream_t b, hipEvent_t eb, volatile int* flag) {
    (void)hipDeviceSynchronize();
    *flag = 0;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, flag);
    (void)hipEventRecord(eb, b);
    bool beside = false;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(3)) {
        if (hipEventQuery(eb) == hipSuccess) {
            beside = true;
            break;
        }
    }
    *flag = 1;      // release the spin
    (void)hipDeviceSynchronize();
    return beside;
}

}  // namespace dpft

using namespace dpft;

// out[0..n-1]: streams (hipStreamNonBlocking, owned by the library, never destroyed) that share a hardware queue neither
// with `main_stream` nor with each other, as far as the device has queues; the remaining entries repeat the found ones
// round-robin.  Returns the number of DISTINCT queues found (<= n) or a negative error code.  Synchronises the device
// (call it at set-up time, never during a graph capture).
extern "C" int32_t dpft_stream_set(dpft_stream_t main_stream, int32_t n, dpft_stream_t* out) {
    if (!out || n < 1 || n > 8) {
        set_error("stream_set: n must be 1..8");
        return DPFT_ERR_ARG;
    }
    hipEvent_t eb;
    int* flag = nullptr;
    if (hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&flag, sizeof(int), hipHostMallocDefault) != hipSuccess) {
        set_error("stream_set: event / pinned flag creation failed");
        return DPFT_ERR_LAUNCH;
    }
    // rejected candidates stay alive until the search ends: the runtime hands a new stream the least-used hardware queue,
    // i.e. the one a stream destroyed a moment ago has just left
    std::vector<hipStream_t> chosen, rejected;
    const hipStream_t mainq = (hipStream_t)main_stream;
    // Priority of the side streams (DPFT_STREAM_PRIORITY=low|normal|high, default normal): everything on them -- the radar
    // encoders, the camera's weight gradients -- is off the step's critical chain (the camera encoder on the caller's stream),
    // so "low" lets the dispatcher hand freed CUs to the chain's kernels first.
    int prio = 0;
    bool with_prio = false;
    if (const char* e = getenv("DPFT_STREAM_PRIORITY")) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
            if (!strcmp(e, "low")) { prio = least; with_prio = true; }
            else if (!strcmp(e, "high")) { prio = greatest; with_prio = true; }
        }
    }
    for (int attempt = 0; attempt < 24 && (int)chosen.size() < n; ++attempt) {
        hipStream_t s;
        if ((with_prio ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio)
                       : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) break;
        (void)hipEventRecord(eb, s);      // first use of a stream creates its hardware queue (milliseconds): not inside a probe
        (void)hipStreamSynchronize(s);
        bool fresh = runs_beside(mainq, s, eb, flag);
        for (size_t i = 0; fresh && i < chosen.size(); ++i) fresh = runs_beside(chosen[i], s, eb, flag);
        if (fresh) chosen.push_back(s);
        else rejected.push_back(s);
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    (void)hipEventDestroy(eb);
    (void)hipHostFree(flag);
    if (chosen.empty()) {      // a single hardware queue: any stream will do
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            set_error("stream_set: stream creation failed");
            return DPFT_ERR_LAUNCH;
        }
        chosen.push_back(s);
        for (int i = 0; i < n; ++i) out[i] = (dpft_stream_t)s;
        return 0;
    }
    for (int i = 0; i < n; ++i) out[i] = (dpft_stream_t)chosen[i % chosen.size()];
    return (int32_t)chosen.size();
}
