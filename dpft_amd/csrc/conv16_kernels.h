// Thin-channel convolutions of the FPN necks (src/dprt/models/necks/fpn.py:39-43 -> torchvision
// FeaturePyramidNetwork: 3x3 16->16 `layer_blocks`, 1x1 C->16 `inner_blocks`; C = 3 / 6 on the raw-input level).
// At the camera's level 0 they run over 4 x 512 x 910 pixels: the generic implicit-GEMM path pads N = 16 to a
// 32-wide tile and gathers scalars (fwd 530 us, dgrad 640 us, wgrad 1200 us per call); these kernels map the 16
// channels exactly onto v_mfma_f32_16x16x4_f32:
//   conv16_3x3_kernel<FLIP> : fwd (FLIP = false) / dgrad (FLIP = true, transposed weights, mirrored taps).
//                             Block = 8 x 32 output pixels, (8+2) x (32+2) x 16 halo tile in LDS; a wave owns four
//                             groups of 16 pixels; per tap one ds_read_b128 per lane feeds four MFMAs (the K order
//                             inside a tap is permuted so that lane k-quad q holds channels 4q..4q+3).
//   wgrad16_3x3_kernel      : dW[k][tap][c] = sum_pixels dy[p][k] x[p + tap][c]; A = dy^T, B = x, 4 pixels per MFMA,
//                             nine 16x16 accumulators (one per tap) per wave; per-block partial slabs + the
//                             deterministic split-K reduction (no atomics).
//   wgrad1x1_small_kernel   : K <= 16, C <= 8 (raw-input laterals, the 6->3 radar adjustment): pure streaming.
#pragma once
#include "common.h"

namespace dpft {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Conv16Args {
    const float* x;      // (B,H,W,16) source (activations, or dy for the data gradient)
    const float* w;      // [16][9][16]: fwd [k][tap][c]; dgrad [c][tap][k] (dpft_weight_transpose_f32)
    const float* bias;   // fwd only, may be null
    float* y;            // (B,H,W,16)
    int B, H, W;
    int accumulate;
    const float* pos_x;  // fwd only, may be null: y += pos_x[w][16]; y += pos_y[h][16] behind the bias (the neck's sinusoidal
    const float* pos_y;  // embedding, embeddings/sinusoidal.py: the reference's two fp32 adds in their order)
};

constexpr int T16H = 8, T16W = 32;

template <bool FLIP>
__global__ __launch_bounds__(256) void conv16_3x3_kernel(Conv16Args a) {
    __shared__ __attribute__((aligned(16))) float tile[(T16H + 2) * (T16W + 2) * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h0 = blockIdx.y * T16H, w0 = blockIdx.x * T16W;
    const float* xb = a.x + (size_t)b * a.H * a.W * 16;
    // halo tile (zero outside the image)
    for (int i = tid; i < (T16H + 2) * (T16W + 2) * 4; i += 256) {
        const int px = i >> 2, q = i & 3;
        const int r = px / (T16W + 2), c = px - r * (T16W + 2);
        const int h = h0 + r - 1, w = w0 + c - 1;
        f32x4v v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W)
            v = *reinterpret_cast<const f32x4v*>(xb + ((size_t)h * a.W + w) * 16 + q * 4);
        *reinterpret_cast<f32x4v*>(tile + px * 16 + q * 4) = v;
    }
    // weights of this lane: output channel n = lane % 16, input channels 4q..4q+3 (q = lane / 16), all 9 taps
    const int n = lane & 15, q = lane >> 4;
    f32x4v wreg[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wreg[t] = *reinterpret_cast<const f32x4v*>(a.w + ((size_t)n * 9 + t) * 16 + q * 4);
    __syncthreads();
    f32x4v acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = wv * 2 + (g >> 1), cb = (g & 1) * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dr = FLIP ? 2 - t / 3 : t / 3, dc = FLIP ? 2 - t % 3 : t % 3;
            const f32x4v av = *reinterpret_cast<const f32x4v*>(tile + ((r + dr) * (T16W + 2) + cb + n + dc) * 16 + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wreg[t][j], acc[g], 0, 0, 0);
        }
    }
    // D: column (output channel) = lane % 16, rows (pixels of the group) = 4 * (lane / 16) + i
    const float bv = (!FLIP && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int h = h0 + wv * 2 + (g >> 1);
        if (h >= a.H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int w = w0 + (g & 1) * 16 + q * 4 + i;
            if (w >= a.W) continue;
            float* o = a.y + (((size_t)b * a.H + h) * a.W + w) * 16 + n;
            float v = acc[g][i] + bv;
            if (a.accumulate) v += *o;
            if (!FLIP && a.pos_x) {
                v += a.pos_x[w * 16 + n];
                v += a.pos_y[h * 16 + n];
            }
            *o = v;
        }
    }
}

struct Wgrad16Args {
    const float* x;      // (B,H,W,16)
    const float* dy;     // (B,H,W,16)
    float* partial;      // [gridDim.x][16][9][16]
    int B, H, W;
    int groups_per_row;  // ceil(W / 4)
    long total_groups;   // B * H * groups_per_row
};

__global__ __launch_bounds__(256) void wgrad16_3x3_kernel(Wgrad16Args a) {
    __shared__ float red[3][9 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kk = lane >> 4;      // A: row k = col, pixel kk;  B: column c = col, pixel kk
    f32x4v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // contiguous chunk of pixel groups per wave (neighbouring rows stay in the same wave's cache footprint)
    const long nw = (long)gridDim.x * 4;
    const long per = (a.total_groups + nw - 1) / nw;
    const long g0 = ((long)blockIdx.x * 4 + wv) * per, g1 = min(a.total_groups, g0 + per);
    for (long g = g0; g < g1; g += 2) {      // two groups of 4 pixels per trip: 20 independent loads in flight
        float av[2], bv[2][9];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long gg = g + u;
            const bool live = gg < g1;
            const long gc = live ? gg : g;
            const int gw = (int)(gc % a.groups_per_row);
            const long bh = gc / a.groups_per_row;
            const int h = (int)(bh % a.H);
            const int b = (int)(bh / a.H);
            const int w = gw * 4 + kk;
            const bool ok = live && w < a.W;
            const size_t img = (size_t)b * a.H * a.W;
            const float a0 = a.dy[(img + (size_t)h * a.W + (ok ? w : 0)) * 16 + col];
            av[u] = ok ? a0 : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
                const bool v = ok && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
                const size_t idx = v ? (img + (size_t)hh * a.W + ww) * 16 + col : 0;
                const float t0 = a.x[idx];
                bv[u][t] = v ? t0 : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][t], acc[t], 0, 0, 0);
    }
    // D[t]: column c = lane % 16, rows k = 4 * (lane / 16) + i.  Reduce the 4 waves, then store the block's slab.
    if (wv > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wv - 1][(t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    if (wv == 0) {
        float* out = a.partial + (size_t)blockIdx.x * 2304;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = (t * 4 + i) * 64 + lane;
                const float s = acc[t][i] + red[0][e] + red[1][e] + red[2][e];
                out[((kk * 4 + i) * 9 + t) * 16 + col] = s;      // [k][tap][c]
            }
    }
}

// LDS-tiled form of the 3x3 16 -> 16 weight gradient (round 3).  The streaming form above loads every operand element as a
// scalar, once per TAP, with a division chain per pixel group: ~100 vector instructions per 9 MFMAs (240 us = 36 TF on the
// camera's 512 x 910 level).  Here a block walks 8 x 32-pixel tiles like conv16_3x3_kernel: dy tile and x halo tile are
// brought into LDS with 16-byte loads (each element read once from memory, 1.14x with the halo), a wave owns two rows of
// the tile = 16 groups of 4 pixels, and a group costs 10 conflict-free ds_read_b32 (A = dy^T once, B = x at the nine taps)
// for its nine v_mfma_f32_16x16x4_f32.  Accumulators stay in registers over all tiles of the block; same per-block partial
// slabs and deterministic reduction as before.
struct Wgrad16TArgs {
    const float* x;
    const float* dy;
    float* partial;      // [gridDim.x][16][9][16] (+ [16] column sums of dy behind each slab when `bias` is set: slabs of 2320)
    int B, H, W;
    int tiles_w, tiles_h;      // ceil(W / 32), ceil(H / 8)
    int total_tiles;           // B * tiles_h * tiles_w
    int bias;                  // also produce sum_p dy[p][k] (the conv's bias gradient: dy is in LDS anyway)
};

__global__ __launch_bounds__(256) void wgrad16_3x3_tiled_kernel(Wgrad16TArgs a) {
    constexpr int XW = T16W + 2, XH = T16H + 2;
    __shared__ __attribute__((aligned(16))) float tiles[XH * XW * 16 + T16H * T16W * 16];
    float* const xt = tiles;                      // halo tile of x, zero outside the image
    float* const yt = tiles + XH * XW * 16;       // dy tile, zero outside the image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kk = lane >> 4;      // A: row k = col, pixel kk;  B: column c = col, pixel kk
    f32x4v acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;      // sum of dy[.][col] over this lane's pixels
    // The next tile travels global -> registers WHILE this one multiplies (10 independent 16-byte loads per thread in
    // flight behind ~2 us of MFMAs); loaded one loop trip at a time the tile cost ~10 us of exposed latency.
    constexpr int XQ = XH * XW * 4, YQ = T16H * T16W * 4;      // 16-byte quads per tile
    constexpr int XS = (XQ + 255) / 256, YS = YQ / 256;
    f32x4v rx[XS], ry[YS];
    auto load_tile = [&](int tile) {
        const int tw = tile % a.tiles_w;
        const int rest = tile / a.tiles_w;
        const int th = rest % a.tiles_h, b = rest / a.tiles_h;
        const int h0 = th * T16H, w0 = tw * T16W;
        const float* xb = a.x + (size_t)b * a.H * a.W * 16;
        const float* yb = a.dy + (size_t)b * a.H * a.W * 16;
#pragma unroll
        for (int s_ = 0; s_ < XS; ++s_) {
            const int i = tid + s_ * 256;
            const int px = i >> 2, q = i & 3;
            const int r = px / XW, c = px - r * XW;
            const int h = h0 + r - 1, w = w0 + c - 1;
            const bool ok = i < XQ && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(xb + (ok ? ((size_t)h * a.W + w) * 16 + q * 4 : 0));
            rx[s_] = ok ? v : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s_ = 0; s_ < YS; ++s_) {
            const int i = tid + s_ * 256;
            const int px = i >> 2, q = i & 3;
            const int r = px / T16W, c = px - r * T16W;
            const int h = h0 + r, w = w0 + c;
            const bool ok = h < a.H && w < a.W;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(yb + (ok ? ((size_t)h * a.W + w) * 16 + q * 4 : 0));
            ry[s_] = ok ? v : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    int tile = blockIdx.x;
    if (tile < a.total_tiles) load_tile(tile);
    while (tile < a.total_tiles) {
        __syncthreads();      // the previous tile's fragments have been read
#pragma unroll
        for (int s_ = 0; s_ < XS; ++s_) {
            const int i = tid + s_ * 256;
            if (i < XQ) *reinterpret_cast<f32x4v*>(xt + i * 4) = rx[s_];
        }
#pragma unroll
        for (int s_ = 0; s_ < YS; ++s_) *reinterpret_cast<f32x4v*>(yt + (tid + s_ * 256) * 4) = ry[s_];
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < a.total_tiles) load_tile(next);
        // wave wv: tile rows 2 wv, 2 wv + 1; group g: row 2 wv + g / 8, columns 4 (g % 8) .. + 3; this lane: pixel kk of it
#pragma unroll 4
        for (int g = 0; g < 16; ++g) {
            const int r = wv * 2 + (g >> 3), c = (g & 7) * 4 + kk;
            const float av = yt[(r * T16W + c) * 16 + col];
            bsum += av;
            float bv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) bv[t] = xt[((r + t / 3) * XW + c + t % 3) * 16 + col];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
        }
        tile = next;
    }
    // D[t]: column c = lane % 16, rows k = 4 * (lane / 16) + i.  Reduce the 4 waves, then store the block's slab.
    __syncthreads();
    float* red = tiles;      // [3][9 * 256] over both tiles
    static_assert(3 * 9 * 256 <= XH * XW * 16 + T16H * T16W * 16, "wave reduction does not fit the tiles");
    if (wv > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wv - 1) * 2304 + (t * 4 + i) * 64 + lane] = acc[t][i];
    }
    __syncthreads();
    const int slab = a.bias ? 2320 : 2304;
    if (wv == 0) {
        float* out = a.partial + (size_t)blockIdx.x * slab;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = (t * 4 + i) * 64 + lane;
                const float s = acc[t][i] + red[e] + red[2304 + e] + red[4608 + e];
                out[((kk * 4 + i) * 9 + t) * 16 + col] = s;      // [k][tap][c]
            }
    }
    if (a.bias) {      // column sums of dy: the four pixel lanes of a column, then the four waves (fixed order)
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        __syncthreads();      // the wave reduction above has been read
        if (lane < 16) red[wv * 16 + lane] = bsum;
        __syncthreads();
        if (tid < 16) a.partial[(size_t)blockIdx.x * slab + 2304 + tid] = red[tid] + red[16 + tid] + red[32 + tid] + red[48 + tid];
    }
}

// Weight gradient of the ResNet stem (7x7, stride 2, pad 3, 3 -> 64): dW[k][n] = sum_pixels dy[p][k] * patch(p)[n], n = (r, s, c)
// flattened = 147 columns.  The generic gather kernel runs it at 38 TF (227 us on the camera's 4 x 512 x 910 input).  LDS-tiled
// like wgrad16_3x3_tiled_kernel: a block walks 4 x 16-pixel output tiles; the dy tile (64 pixels x 64 channels) and the
// 13 x 37-pixel input tile go to LDS (the next tile is prefetched into registers while this one multiplies).  For a fixed
// filter row r the 21 values (s, c) an output pixel needs are CONTIGUOUS in the input row (element 6 ow + 3 s + c), so the
// B operand of v_mfma_f32_16x16x4_f32 is one ds_read_b32 at  pixel base + column offset  with a per-lane constant offset
// (147 columns padded to 160: 10 column blocks x 4 channel blocks = 40 accumulators per wave, 92 % useful MFMAs).  A wave
// owns one output row of the tile (4 groups of 4 pixels) and the full 64 x 160 result; waves are summed through LDS at the
// end; per-block slabs + the deterministic reduction as everywhere.
struct WgradStemArgs {
    const float* x;      // (B,H,W,3)
    const float* dy;     // (B,OH,OW,64)
    float* partial;      // [gridDim.x][64][147]
    int B, H, W, OH, OW;
    int tiles_w, tiles_h, total_tiles;
};

constexpr int STEM_OH = 4, STEM_OW = 16, STEM_IH = 2 * STEM_OH + 5, STEM_IW = 2 * STEM_OW + 5, STEM_ROW = 112;

__global__ __launch_bounds__(256) void wgrad_stem7_kernel(WgradStemArgs a) {
    constexpr int XN = STEM_IH * STEM_ROW;             // x tile floats (rows padded 111 -> 112)
    constexpr int YN = STEM_OH * STEM_OW * 64;         // dy tile floats
    constexpr int RED = 40 * 256;                      // one wave's accumulators
    __shared__ __attribute__((aligned(16))) float sm[RED > XN + YN ? RED : XN + YN];
    float* const xt = sm;
    float* const yt = sm + XN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kk = lane >> 4;
    // column n = nb * 16 + col of the flattened (r, s, c) axis -> offset of its first element relative to the pixel base
    int coloff[10];
    bool colok[10];
#pragma unroll
    for (int nb = 0; nb < 10; ++nb) {
        const int n = nb * 16 + col;
        colok[nb] = n < 147;
        const int r = n / 21, rem = n - r * 21;
        coloff[nb] = colok[nb] ? r * STEM_ROW + rem : 0;
    }
    f32x4v acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    constexpr int XS = (STEM_IH * 111 + 255) / 256, YS = YN / 4 / 256;
    float rx[XS];
    f32x4v ry[YS];
    auto load_tile = [&](int tile) {
        const int tw = tile % a.tiles_w;
        const int rest = tile / a.tiles_w;
        const int th = rest % a.tiles_h, b = rest / a.tiles_h;
        const int oh0 = th * STEM_OH, ow0 = tw * STEM_OW;
        const float* xb = a.x + (size_t)b * a.H * a.W * 3;
        const float* yb = a.dy + (size_t)b * a.OH * a.OW * 64;
#pragma unroll
        for (int s_ = 0; s_ < XS; ++s_) {
            const int i = tid + s_ * 256;
            const int r = i / 111, e = i - r * 111;      // input row of the tile, element (w, c) of the row
            const int h = 2 * oh0 - 3 + r, w = 2 * ow0 - 3 + e / 3;
            const bool ok = i < STEM_IH * 111 && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            const float v = xb[ok ? ((size_t)h * a.W + w) * 3 + e % 3 : 0];
            rx[s_] = ok ? v : 0.f;
        }
#pragma unroll
        for (int s_ = 0; s_ < YS; ++s_) {
            const int i = tid + s_ * 256;
            const int px = i >> 4, q = i & 15;
            const int oh = oh0 + px / STEM_OW, ow = ow0 + px % STEM_OW;
            const bool ok = oh < a.OH && ow < a.OW;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(yb + (ok ? ((size_t)oh * a.OW + ow) * 64 + q * 4 : 0));
            ry[s_] = ok ? v : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    int tile = blockIdx.x;
    if (tile < a.total_tiles) load_tile(tile);
    while (tile < a.total_tiles) {
        __syncthreads();      // the previous tile's fragments have been read
#pragma unroll
        for (int s_ = 0; s_ < XS; ++s_) {
            const int i = tid + s_ * 256;
            if (i < STEM_IH * 111) {
                const int r = i / 111, e = i - r * 111;
                xt[r * STEM_ROW + e] = rx[s_];
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < YS; ++s_) *reinterpret_cast<f32x4v*>(yt + (tid + s_ * 256) * 4) = ry[s_];
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < a.total_tiles) load_tile(next);
        // wave wv: output row wv of the tile; group g: columns 4 g .. 4 g + 3; this lane: pixel kk of the group
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            const int px = wv * STEM_OW + g * 4 + kk;
            const int base = 2 * wv * STEM_ROW + 6 * (g * 4 + kk);
            float av[4], bv[10];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = yt[px * 64 + i * 16 + col];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float t = xt[base + coloff[j]];
                bv[j] = colok[j] ? t : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        tile = next;
    }
    // acc[i][j][e]: channel k = 16 i + 4 (lane / 16) + e, column n = 16 j + lane % 16.  Sum the waves (one at a time
    // through LDS: 40 KB), then wave 0 stores the block's slab [64][147].
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 10; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sm[((i * 10 + j) * 4 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 10; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] += sm[((i * 10 + j) * 4 + e) * 64 + lane];
        }
    }
    if (wv == 0) {
        float* out = a.partial + (size_t)blockIdx.x * (64 * 147);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int n = j * 16 + col;
                if (n < 147) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[(i * 16 + kk * 4 + e) * 147 + n] = acc[i][j][e];
                }
            }
    }
}

static bool wgrad_stem7_matches(const dpft_conv_desc* d) {
    // (each block ends with a 3-round wave reduction and a 37 KB slab: worth it from ~8 tiles per block on -- the radar stems,
    // 27 k output pixels, stay with the generic kernel: 26.6 vs 37.0 us)
    return d->kh == 7 && d->kw == 7 && d->stride == 2 && d->pad == 3 && d->C == 3 && d->K == 64 && !d->act16 &&
           (int64_t)d->B * d->OH * d->OW >= 131072;
}

// 1x1 weight gradient with very few channels: dW[k][c] = sum_p dy[p][k] * x[p][c]; one thread walks a pixel stripe
template <int K, int C>
__global__ __launch_bounds__(256) void wgrad1x1_small_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, long M) {
    __shared__ float red[4][K * C];
    float acc[K][C];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[k][c] = 0.f;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < M; p += (long)gridDim.x * 256) {
        float xv[C], dv[K];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = x[p * C + c];
#pragma unroll
        for (int k = 0; k < K; ++k) dv[k] = dy[p * K + k];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[k][c] = fmaf(dv[k], xv[c], acc[k][c]);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = acc[k][c];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
            if (lane == 0) red[wv][k * C + c] = v;
        }
    __syncthreads();
    if (threadIdx.x < K * C)
        partial[(size_t)blockIdx.x * K * C + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// K = 16 variant with coalesced operand reads: 4 lanes per pixel, each owns 4 consecutive output channels (one 16-byte
// load of dy) and all C input channels; 64 pixels per block iteration.  (One thread per pixel made every dy load
// instruction touch 64 cache lines: 290 us for the camera's raw-input lateral instead of ~40.)
// BIAS: a virtual input channel of ones -- partial slabs are [16][C + 1], column C = sum_p dy[p][k] (the bias gradient)
template <int C, bool BIAS = false>
__global__ __launch_bounds__(256) void wgrad1x1_k16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partial, long M) {
    constexpr int CC = C + (BIAS ? 1 : 0);
    __shared__ float red[4][16 * CC];
    const int tid = threadIdx.x, kq = tid & 3, pl = tid >> 2;
    float acc[4][CC];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[e][c] = 0.f;
    for (long p = (long)blockIdx.x * 64 + pl; p < M; p += (long)gridDim.x * 64) {
        const f32x4v d4 = *reinterpret_cast<const f32x4v*>(dy + p * 16 + kq * 4);
        float xv[CC];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = x[p * C + c];
        if (BIAS) xv[CC - 1] = 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CC; ++c) acc[e][c] = c < C ? fmaf(d4[e], xv[c], acc[e][c]) : acc[e][c] + d4[e];
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            float v = acc[e][c];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o);      // lanes with the same kq
            if (lane < 4) red[wv][(lane * 4 + e) * CC + c] = v;
        }
    __syncthreads();
    if (tid < 16 * CC)
        partial[(size_t)blockIdx.x * 16 * CC + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// slab reduction of a weight gradient whose slabs carry the bias gradient as well: rows of `cols` floats, the first
// `cols - 1` (or, tail form, the first `split`) belong to dw, the rest to db
__global__ __launch_bounds__(256) void slab_reduce_bias_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                               float* __restrict__ db, int MN, int slabs, int cols, int split) {
    const int j = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    float s = 0.f;
    if (j < MN)
        for (int k = part; k < slabs; k += 16) s += partial[(size_t)k * MN + j];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    __shared__ float red[4][16];
    if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && j < MN) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (cols > 0) {      // interleaved: row k = j / cols, column c = j % cols; column cols - 1 is the bias
            const int k = j / cols, c = j - k * cols;
            if (c == cols - 1) db[k] = t;
            else dw[k * (cols - 1) + c] = t;
        } else {             // tail: dw first, db behind it
            if (j < split) dw[j] = t;
            else db[j - split] = t;
        }
    }
}

// out[j] = sum_s partial[s][j] for MANY slabs of a SMALL result (hundreds of workgroup partials of a 48..2304-element
// weight gradient): 16 outputs x 16 slab-lanes per block, then a shuffle tree.  (The generic split-K reduction walks the
// slabs serially per thread: 500 dependent loads = 200 us.)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int MN,
                                                          int slabs) {
    const int j = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
    float s = 0.f;
    if (j < MN)
        for (int k = part; k < slabs; k += 16) s += partial[(size_t)k * MN + j];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    __shared__ float red[4][16];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 16) red[wv][lane] = s;
    __syncthreads();
    if (threadIdx.x < 16 && j < MN) out[j] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- host side: called by the generic conv entry points (conv.hip) when the shape matches --------------------

static bool conv16_matches(const dpft_conv_desc* d) {
    return d->C == 16 && d->K == 16 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1;
}

// 1x1 forward with very few input channels and 16 outputs (the FPN lateral on the raw-input level: 3 -> 16 over
// 4 x 512 x 910 pixels, 6 -> 16 on the radar maps): pure streaming, 4 C bytes in and 64 bytes out per pixel.  The generic
// implicit-GEMM path pads to a 32-wide tile and gathers scalars (92 us = 1.5 TB/s on the camera level); here 4 lanes share
// a pixel, each computes 4 outputs from the pixel's C inputs and stores one float4 -- a wave writes 1 KiB contiguous.
// TOP: + nearest_upsample(top[B][TH][TW][16]) (the neck's top-down path, src = min(floor(dst * in / out), in - 1) as
// fpn_topdown_add_kernel) behind the bias -- the lateral and the add were two passes over the level's 64 bytes per pixel.
template <int C, bool TOP = false>
__global__ __launch_bounds__(256) void conv1x1_to16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, long M,
                                                           const float* __restrict__ top = nullptr, int H = 0, int W = 0, int TH = 0,
                                                           int TW = 0, float sh = 0.f, float sw = 0.f) {
    const int q = threadIdx.x & 3;      // outputs 4 q .. 4 q + 3
    float wr[4][C], br[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        br[k] = bias ? bias[q * 4 + k] : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) wr[k][c] = w[(q * 4 + k) * C + c];
    }
    const long stride = (long)gridDim.x * 64;
    for (long p = (long)blockIdx.x * 64 + (threadIdx.x >> 2); p < M; p += stride) {
        float xv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = x[p * C + c];
        f32x4v o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.f;      // the reduction order of the generic path: c ascending, bias last
#pragma unroll
            for (int c = 0; c < C; ++c) s = fmaf(xv[c], wr[k][c], s);
            o[k] = s + br[k];
        }
        if constexpr (TOP) {
            const int wq = (int)(p % W);
            const long bh = p / W;
            const int h = (int)(bh % H);
            const long b = bh / H;
            const int th = (TH == H) ? h : min((int)floorf((float)h * sh), TH - 1);
            const int tw = (TW == W) ? wq : min((int)floorf((float)wq * sw), TW - 1);
            o += *reinterpret_cast<const f32x4v*>(top + ((b * TH + th) * TW + tw) * 16 + q * 4);
        }
        *reinterpret_cast<f32x4v*>(y + p * 16 + q * 4) = o;
    }
}

static bool conv1x1_to16_matches(const dpft_conv_desc* d) {
    // (below ~256 k pixels the launch is latency-sized and the generic path is as fast: radar laterals 13.1 vs 14.6 us)
    return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->K == 16 && (d->C == 3 || d->C == 6) && !d->act16 &&
           (int64_t)d->B * d->H * d->W >= 262144;
}

static int conv1x1_to16_forward(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const long M = (long)d->B * d->H * d->W;
    const int blocks = (int)std::max<long>(1, std::min<long>(kNumCU * 16, (M + 63) / 64));
    if (d->C == 3) hipLaunchKernelGGL((conv1x1_to16_kernel<3, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, y, M, (const float*)nullptr, 0, 0, 0, 0, 0.f, 0.f);
    else hipLaunchKernelGGL((conv1x1_to16_kernel<6, false>), dim3(blocks), dim3(256), 0, st, x, w, bias, y, M, (const float*)nullptr, 0, 0, 0, 0, 0.f, 0.f);
    return check_launch("conv 1x1 -> 16 fwd");
}

// the lateral of a raw-input level with the top-down add in its epilogue (any map size: it replaces two launches)
static bool conv1x1_to16_top_matches(const dpft_conv_desc* d) {
    return d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->K == 16 && (d->C == 3 || d->C == 6) && !d->act16;
}
static int conv1x1_to16_top_forward(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, const float* top, int TH,
                                    int TW, float* y, hipStream_t st) {
    const long M = (long)d->B * d->H * d->W;
    const int blocks = (int)std::max<long>(1, std::min<long>(kNumCU * 16, (M + 63) / 64));
    const float sh = (float)TH / (float)d->H, sw = (float)TW / (float)d->W;
    if (d->C == 3) hipLaunchKernelGGL((conv1x1_to16_kernel<3, true>), dim3(blocks), dim3(256), 0, st, x, w, bias, y, M, top, d->H, d->W, TH, TW, sh, sw);
    else hipLaunchKernelGGL((conv1x1_to16_kernel<6, true>), dim3(blocks), dim3(256), 0, st, x, w, bias, y, M, top, d->H, d->W, TH, TW, sh, sw);
    return check_launch("conv 1x1 -> 16 fwd + top-down add");
}

static int conv16_forward(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st,
                          const float* pos_x = nullptr, const float* pos_y = nullptr) {
    Conv16Args a{x, w, bias, y, d->B, d->H, d->W, 0, pos_x, pos_y};
    dim3 grid(cdiv(d->W, T16W), cdiv(d->H, T16H), d->B);
    hipLaunchKernelGGL(conv16_3x3_kernel<false>, grid, dim3(256), 0, st, a);
    return check_launch("conv16 fwd");
}

static int conv16_dgrad(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx, int accumulate, hipStream_t st) {
    Conv16Args a{dy, w_t, nullptr, dx, d->B, d->H, d->W, accumulate, nullptr, nullptr};
    dim3 grid(cdiv(d->W, T16W), cdiv(d->H, T16H), d->B);
    hipLaunchKernelGGL(conv16_3x3_kernel<true>, grid, dim3(256), 0, st, a);
    return check_launch("conv16 dgrad");
}

static int conv16_wgrad_blocks(const dpft_conv_desc* d) {      // streaming form; also the workspace bound of both forms
    const long groups = (long)d->B * d->H * cdiv(d->W, 4);
    return (int)std::max<long>(1, std::min<long>(kNumCU * 4, groups / 64));
}
static int conv16_wgrad_tiled_blocks(const dpft_conv_desc* d) {
    const long tiles = (long)d->B * cdiv(d->H, T16H) * cdiv(d->W, T16W);
    return (int)std::max<long>(1, std::min<long>(std::min<long>(kNumCU * 4, conv16_wgrad_blocks(d)), tiles));
}

}  // namespace dpft
