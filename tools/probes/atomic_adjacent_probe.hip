// Probe for the training decoder's pyramid-gradient scatter (xf_train_bwd): does it pay to lay the gradient maps
// out head-major ([head][pixel][2 channels]) so that the two horizontally adjacent bilinear corners of a sample are 16 contiguous
// bytes -- one atomic request instead of two?
//   A: [pixel][16 channels] (today): per (sample, head) 4 corner updates of 8 bytes, 64 bytes apart / a row apart
//   B: [head][pixel][2]: per (sample, head) 2 updates of 16 bytes (x0, x0 + 1), a row apart
// P = 768 k (sample, head) pairs per launch (one xf_train_bwd call at batch 4), uniform random positions; big map (4 x 128 x 228) and
// small map (4 x 16 x 29).  Prints us per launch and G requests / s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// lane l of a wave: pair = l / 2 within the instruction (32 pairs), channel = l % 2; 4 instructions (corners) per 32 pairs
__global__ __launch_bounds__(256) void scatter_pixel_major(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 1, c = gid & 1;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int head = pair & 7, b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + (k >> 1), x = x0 + (k & 1);
        atomicAdd(g + (((size_t)b * H + y) * W + x) * 16 + head * 2 + c, v);
    }
}

// lane l: pair = l / 4 (16 pairs per instruction), dx = (l / 2) % 2, channel = l % 2; 2 instructions (rows) per 16 pairs
__global__ __launch_bounds__(256) void scatter_head_major(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 2, dx = (gid >> 1) & 1, c = gid & 1;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int head = pair & 7, b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int y = y0 + k, x = x0 + dx;
        atomicAdd(g + ((((size_t)head * B + b) * H + y) * W + x) * 2 + c, v);
    }
}

// C: four replicas of the map, each tiled in 2 x 2-pixel blocks with the tiling shifted by (x0 % 2, y0 % 2): the corner quad of ANY
// sample is exactly one block of one replica = 32 contiguous bytes per (sample, head).  lane l: pair = l / 8, corner = (l / 2) % 4
__global__ __launch_bounds__(256) void scatter_quad_replica(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 3, k = (gid >> 1) & 3, c = gid & 1;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int head = pair & 7, b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
    const int TH = H / 2 + 1, TW = W / 2 + 1;
    const int rep = (y0 & 1) * 2 + (x0 & 1), ty = (y0 + 1) >> 1, tx = (x0 + 1) >> 1;      // block of the shifted tiling holding (x0, y0) as its first pixel
    atomicAdd(g + ((((((size_t)rep * 8 + head) * B + b) * TH + ty) * TW + tx) * 4 + k) * 2 + c, v);
}

// The decoder's real shape: a (sample, head) pair updates all 16 channels of each corner pixel (the value projection is applied
// after sampling).  D: [pixel][16], lane = (pair % 4, channel): 4 instructions of 4 x 64-byte rows per 4 pairs (today's form).
__global__ __launch_bounds__(256) void scatter16_pixel_major(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 4, c = gid & 15;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = y0 + (k >> 1), x = x0 + (k & 1);
        atomicAdd(g + (((size_t)b * H + y) * W + x) * 16 + c, v);
    }
}
// E: four shifted 2 x 2-tiled replicas of [pixel][16]: the corner quad = 256 contiguous bytes, lane = (corner, channel): one
// instruction per pair
__global__ __launch_bounds__(256) void scatter16_quad_replica(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 6, k = (gid >> 4) & 3, c = gid & 15;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
    const int TH = H / 2 + 1, TW = W / 2 + 1;
    const int rep = (y0 & 1) * 2 + (x0 & 1), ty = (y0 + 1) >> 1, tx = (x0 + 1) >> 1;
    atomicAdd(g + (((((size_t)rep * B + b) * TH + ty) * TW + tx) * 4 + k) * 16 + c, v);
}
// F: two replicas shifted in x only, tiled in 2 x 1 pixel pairs: 128 contiguous bytes per row of the quad: 2 instructions of 2 pairs
__global__ __launch_bounds__(256) void scatter16_pair_replica(float* g, int B, int H, int W, int pairs) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pair = gid >> 5, dx = (gid >> 4) & 1, c = gid & 15;
    if (pair >= pairs) return;
    const unsigned r = hash(pair * 2654435761u + 17);
    const int b = (r >> 3) % B, y0 = (r >> 8) % (H - 1), x0 = (r >> 20) % (W - 1);
    const float v = 1e-3f * (float)(r & 7);
    const int TW = W / 2 + 1;
    const int rep = x0 & 1, tx = (x0 + 1) >> 1;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        atomicAdd(g + (((((size_t)rep * B + b) * H + y0 + k) * TW + tx) * 2 + dx) * 16 + c, v);
}

template <typename K>
static void run(const char* what, K kernel, int threads_per_pair, float* g, size_t bytes, int B, int H, int W, int pairs, double reqs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (int)(((size_t)pairs * threads_per_pair + 255) / 256);
    hipMemset(g, 0, bytes);
    kernel<<<blocks, 256>>>(g, B, H, W, pairs);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) kernel<<<blocks, 256>>>(g, B, H, W, pairs);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s map %d x %3d x %3d  %8.1f us per launch  %6.1f G requests/s\n", what, B, H, W, ms * 1e3 / 20, reqs / (ms * 1e-3 / 20) * 1e-9);
}

int main() {
    const int pairs = 768 * 1024;
    const int maps[3][3] = {{4, 128, 228}, {4, 32, 57}, {4, 16, 29}};
    for (auto& m : maps) {
        const size_t bytes = (size_t)m[0] * m[1] * m[2] * 16 * sizeof(float);
        float* g;
        hipMalloc(&g, bytes);
        run("A [pixel][16]: 4 x 8-byte updates per pair", scatter_pixel_major, 2, g, bytes, m[0], m[1], m[2], pairs, 4.0 * pairs);
        run("B [head][pixel][2]: 2 x 16-byte updates per pair", scatter_head_major, 4, g, bytes, m[0], m[1], m[2], pairs, 2.0 * pairs);
        hipFree(g);
        {
            const size_t be = (size_t)4 * m[0] * (m[1] / 2 + 1) * (m[2] / 2 + 1) * 64 * sizeof(float);
            hipMalloc(&g, be);
            run("D 16 ch, [pixel][16]: 4 x 64-byte updates per pair", scatter16_pixel_major, 16, g, bytes, m[0], m[1], m[2], pairs, 4.0 * pairs);
            run("E 16 ch, 4 shifted 2x2-tiled replicas: 1 x 256 bytes", scatter16_quad_replica, 64, g, be, m[0], m[1], m[2], pairs, 1.0 * pairs);
            run("F 16 ch, 2 x-shifted 2x1-tiled replicas: 2 x 128 bytes", scatter16_pair_replica, 32, g, be, m[0], m[1], m[2], pairs, 2.0 * pairs);
            hipFree(g);
        }
        const size_t bytes_c = (size_t)4 * 8 * m[0] * (m[1] / 2 + 1) * (m[2] / 2 + 1) * 8 * sizeof(float);
        hipMalloc(&g, bytes_c);
        run("C 4 shifted 2x2-tiled replicas: 1 x 32-byte update per pair", scatter_quad_replica, 8, g, bytes_c, m[0], m[1], m[2], pairs, 1.0 * pairs);
        hipFree(g);
    }
    return 0;
}
