// Streaming 1x1 convolutions with a SHORT reduction (C = 64 / 128 input channels) and wide outputs (K % 256 == 0) on
// large maps -- conv3 / downsample of the camera's layers 1-2 (4 x 128 x 228 x 64 -> 256, 4 x 64 x 114 x 128 -> 512), round 6.
//
// Why.  These GEMMs are one or two K-steps deep: 3.8 GFLOP against 150-270 MB of HBM traffic, i.e. bound by memory and by the
// fp32 MFMA about equally (24 us each at the peaks), but the tiled kernels (conv_pipe.h) spend a workgroup's life in latencies
// that nothing overlaps -- first tile's round trip, barrier, 1-2 K-steps, barrier, staging the tile through LDS, barrier,
// stores -- and measure 70-104 us (2.5 TB/s).  Here nothing is staged and nothing is shared but the weights:
//   * a workgroup (4 waves) keeps its 256-column slice of the weights in LDS ([256][C] fp32, 16-byte chunks XOR-swizzled by the
//     row: conflict-free ds_read_b128 fragments), loaded once;
//   * every WAVE owns 32 whole rows: its A fragments come straight from global memory into registers (lane = (row, K half),
//     16-byte chunks -- the producer's BatchNorm + ReLU is applied there when the conv has that prologue), it multiplies them
//     against the 8 column blocks one after the other (16 accumulator registers), and each 32 x 32 block leaves the registers
//     directly: lanes 0-31 of an accumulator register are 128 contiguous bytes of one output row.  No barrier after the
//     weights have landed; waves of different workgroups (2 per CU) cover each other's memory round trips;
//   * epilogues: raw store + BatchNorm tile statistics (training; tile = the workgroup's rows, Chan-merged per column in
//     registers: block (mean, M2) -> wave -> workgroup), or the inference form relu(bn(conv) + residual) with the residual
//     read in the same lane layout (128-byte row segments) ahead of the block's MFMAs.
// Same arithmetic as the tiled kernels (v_mfma_f32_32x32x2_f32, fp32 accumulation); the k order inside a dot product differs.
#include "conv_core.h"

namespace dpft {

struct StreamArgs {
    const float* x;      // [M][C]
    const float* w;      // [N][C]
    float* y;            // [M][N]
    const float* pro;    // BN block [4][C] of the input (BatchNorm + ReLU prologue) or null
    float* stats;        // [tiles][2][N] or null (tile = the rows of one workgroup)
    unsigned long long* bns;   // the statistics as column sums instead (common.h: BnSumsRef), or null
    BnSumsRef pro_s;     // the prologue's BatchNorm as column sums (.sums null: `pro`)
    const float* obn;    // inference: BN block [4][N] of the output channels, or null
    const float* oadd;   // inference: residual [M][N] or null
    int orelu;
    int M, N;
    int rbw;             // 32-row blocks per wave
    int nslices;         // N / (columns per workgroup)
};

// KC input channels; NB output columns per workgroup (its weight slice [NB][KC] lives in LDS); NW waves per workgroup
template <int KC, int NB, int NW, bool PRO, bool EVAL>
__global__ __launch_bounds__(NW * 64) void conv1x1_stream_kernel(StreamArgs a) {
    constexpr int NT = NW * 64, NCB = NB / 32, CHK = KC / 4, NJ = KC / 8;      // chunks (16 B) per row; chunk pairs (one per lane half)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // weights [256][KC] | prologue table [3][KC]; at the end: statistics merge
    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x % a.nslices, wg = blockIdx.x / a.nslices;
    const int n0 = slice * NB;
    // ---- weights of the slice -> LDS (chunk c of row r at position c ^ (r & 15)) ----
#pragma unroll 4
    for (int i = 0; i < NB * CHK / NT; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / CHK, c = idx - row * CHK;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.w + (size_t)(n0 + row) * KC + c * 4);
        *reinterpret_cast<f32x4*>(smem + row * KC + ((c ^ (row & 15)) << 2)) = v;
    }
    float* ptab = smem + NB * KC;
    if constexpr (PRO) {
        fill_pro_table(ptab, a.pro, a.pro_s, KC, tid, NT);      // rows mean, scale, beta (from the BN block or the layer's column sums)
    }
    __syncthreads();
    // (restrict: without it every block's residual loads wait for the previous block's stores -- s_waitcnt vmcnt(0) -- because
    // y and the residual might alias)
    float* __restrict__ const yp = a.y;
    const float* __restrict__ const addp = a.oadd;
    const float* __restrict__ const xp = a.x;
    const float* __restrict__ const obnp = a.obn;
    const int r32 = lane & 31, h = lane >> 5;
    const int nrb = (a.M + 31) / 32;
    const int rb0 = (wg * NW + wave) * a.rbw, rb1 = min(nrb, rb0 + a.rbw);
    // running per-column statistics of this wave (column = n0 + cb * 32 + r32; both lane halves hold the same values)
    float s_n = 0.f, s_mean[NCB], s_m2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { s_mean[cb] = 0.f; s_m2[cb] = 0.f; }
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // fragment address of (column block 0, chunk pair j): row r32, chunk (2 j + h) ^ (r32 & 15)
    unsigned b_ad[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b_ad[j] = lds_base + (unsigned)(r32 * KC * 4 + (((2 * j + h) ^ (r32 & 15)) << 4));
    for (int rb = rb0; rb < rb1; ++rb) {
        const int m0 = rb * 32;
        const int row = min(m0 + r32, a.M - 1);
        const int valid = min(32, a.M - m0);
        f32x4 af[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) af[j] = *reinterpret_cast<const f32x4*>(xp + (size_t)row * KC + (2 * j + h) * 4);
        if constexpr (PRO) {      // (four chunk pairs at a time: all NJ parameter triples in flight at once spilled at C = 256)
            static_for<NJ / 4>([&](auto G) {
                constexpr int g = decltype(G)::value;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = g * 4 + jj;
                    const f32x4 mu = *reinterpret_cast<const f32x4*>(ptab + (2 * j + h) * 4);
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(ptab + KC + (2 * j + h) * 4);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(ptab + 2 * KC + (2 * j + h) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[j][e] = fmaxf(fmaf(af[j][e] - mu[e], sc[e], sh[e]), 0.f);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // Output / residual addressing: one per-lane base (row m0 + 4 h, this lane's column) per block; the 16 rows of an
        // accumulator are wave-uniform multiples of N away from it -- scalar offsets, no vector address arithmetic beside the
        // fp32 MFMAs (they share the vector ALUs).  Row checks only in the map's last, partly filled row block.
        const bool full = valid == 32;
        const size_t lane_off = (size_t)(m0 + 4 * h) * a.N + n0 + r32;
        // The 8 column blocks, unrolled: the residual of block cb + 1 is requested BEFORE block cb's MFMAs and stores, so the
        // wait in front of an epilogue never includes the previous block's stores (vmcnt counts loads and stores in order).
        float res[2][16];
        auto load_res = [&](auto CB_) {
            constexpr int cb = decltype(CB_)::value;
            if constexpr (EVAL) {
                if (addp) {
                    const float* __restrict__ const rb_ = addp + lane_off + cb * 32;
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) res[cb & 1][r] = rb_[(8 * (r >> 2) + (r & 3)) * a.N];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            res[cb & 1][r] = (8 * (r >> 2) + 4 * h + (r & 3) < valid) ? rb_[(8 * (r >> 2) + (r & 3)) * a.N] : 0.f;
                    }
                }
            }
        };
        load_res(std::integral_constant<int, 0>{});
        static_for<NCB>([&](auto CB_) {
            constexpr int cb = decltype(CB_)::value;
            const int col = n0 + cb * 32 + r32;
            float* __restrict__ const yb = yp + lane_off + cb * 32;
            float omu = 0.f, osc = 1.f, obe = 0.f;
            if constexpr (EVAL) { omu = obnp[col]; osc = obnp[a.N + col]; obe = obnp[2 * a.N + col]; }
            if constexpr (cb + 1 < NCB) load_res(std::integral_constant<int, (cb + 1 < NCB ? cb + 1 : 0)>{});
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // weight fragments two chunk pairs ahead of their MFMAs (an LDS round trip is ~100+ cycles, four MFMAs are 256)
            f32x4 bf[3];
            constexpr unsigned cbo = (unsigned)(cb * 32 * KC * 4);
            bf[0] = *(lds_f32x4*)(size_t)(b_ad[0] + cbo);
            bf[1] = *(lds_f32x4*)(size_t)(b_ad[1] + cbo);
            static_for<NJ>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if constexpr (j + 2 < NJ) bf[(j + 2) % 3] = *(lds_f32x4*)(size_t)(b_ad[j + 2] + cbo);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], bf[j % 3][e], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (EVAL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = fmaf(acc[r] - omu, osc, obe);
                    if (addp) v += res[cb & 1][r];
                    if (a.orelu) v = fmaxf(v, 0.f);
                    acc[r] = v;
                }
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yb[(8 * (r >> 2) + (r & 3)) * a.N] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (8 * (r >> 2) + 4 * h + (r & 3) < valid) yb[(8 * (r >> 2) + (r & 3)) * a.N] = acc[r];
            }
            if constexpr (!EVAL) {
                if (a.stats) {      // (mean, M2) of the block's valid rows, merged into the wave's running pair (Chan)
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += (full || 8 * (r >> 2) + 4 * h + (r & 3) < valid) ? acc[r] : 0.f;
                    s += __shfl_xor(s, 32);
                    const float nb = (float)valid, mb = s / nb;
                    float q = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = acc[r] - mb;
                        q += (full || 8 * (r >> 2) + 4 * h + (r & 3) < valid) ? d * d : 0.f;
                    }
                    q += __shfl_xor(q, 32);
                    const float nn = s_n + nb, dl = mb - s_mean[cb];
                    s_mean[cb] += dl * (nb / nn);
                    s_m2[cb] += q + dl * dl * (s_n * nb / nn);
                }
            }
        });
        s_n += (float)valid;
    }
    if constexpr (!EVAL) {
        if (a.stats) {      // the four waves' pairs -> one (mean, M2) per column of the workgroup's rows
            __syncthreads();      // every wave is done with the weights
            float* sm = smem;     // [NW][3][NB]
            if (h == 0) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    sm[(wave * 3 + 0) * NB + cb * 32 + r32] = s_n;
                    sm[(wave * 3 + 1) * NB + cb * 32 + r32] = s_mean[cb];
                    sm[(wave * 3 + 2) * NB + cb * 32 + r32] = s_m2[cb];
                }
            }
            __syncthreads();
            if (tid < NB) {
                float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
                for (int wv = 0; wv < NW; ++wv) {
                    const float nb = sm[(wv * 3 + 0) * NB + tid], mb = sm[(wv * 3 + 1) * NB + tid], qb = sm[(wv * 3 + 2) * NB + tid];
                    if (nb > 0.f) {
                        const float nn = n + nb, dl = mb - mean;
                        mean += dl * (nb / nn);
                        m2 += qb + dl * dl * (n * nb / nn);
                        n = nn;
                    }
                }
                if (n > 0.f && a.bns) {
                    const double m = (double)mean, fc = (double)n;
                    bn_sums_add(a.bns, a.N, n0 + tid, fc * m, (double)m2 + fc * m * m);
                } else if (n > 0.f) {
                    a.stats[((size_t)wg * 2 + 0) * a.N + n0 + tid] = mean;
                    a.stats[((size_t)wg * 2 + 1) * a.N + n0 + tid] = m2;
                }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Which problems take the kernel, and its row tiling (the statistics tile = the rows of one workgroup = 128 * rbw): decided
// from the descriptor alone so that dpft_conv2d_stats_tiles and the launch agree.  DPFT_STREAM1X1=0: off (A/B switch).
// geometry per input-channel count: columns per workgroup (its weight slice: NB x C x 4 bytes of LDS) and waves per workgroup
static bool stream_geom(int C, int& nb, int& nw) {
    // (C = 256 with 128-column slices on eight waves -- a 128 KB slice, 128 fragment registers per lane -- measured 54 us against
    // the tiled kernel's 48.9 on the layer-3 conv3 shape: one workgroup per CU, and every wave starts behind a 128 KB weight load)
    // tuning aids: C = 128 with 128-column slices (64 KB: two workgroups per CU) and at least DPFT_STREAM_RBW_MIN row blocks per wave --
    // alone the 128 -> 512 training form (prologue + statistics) goes 62.8 -> 53.5 us with (1, 2), 64 -> 256 60.1 -> 57.1; in the step and
    // in the inference forward both within the run-to-run spread (23.91 / 23.91 vs 24.00 / 23.92 ms; 1.78 vs 1.77 ms per frame): off
    static const int nb128 = getenv("DPFT_STREAM_NB128") ? atoi(getenv("DPFT_STREAM_NB128")) : 0;
    if (C == 128 && nb128) { nb = 128; nw = 4; return true; }
    if (C == 64 || C == 128) { nb = 256; nw = 4; return true; }      // 64 / 128 KB: two / one workgroup(s) per CU
    return false;
}

bool stream1x1_match(const dpft_conv_desc* d, int* tile_rows) {
    static const int on = getenv("DPFT_STREAM1X1") ? atoi(getenv("DPFT_STREAM1X1")) : 1;
    static const int min_rows = getenv("DPFT_STREAM1X1_MINROWS") ? atoi(getenv("DPFT_STREAM1X1_MINROWS")) : 4096;
    if (!on || d->act16 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0) return false;
    int nb = 0, nw = 0;
    if (!stream_geom(d->C, nb, nw) || (d->K % 256) != 0 || d->a_planes || d->w_planes) return false;
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    // enough row blocks to give every wave of a chip-filling grid one: the latency-sized problems keep the tiled kernels
    if (M < min_rows || (M / 32) * (d->K / nb) < 2 * kNumCU || M * std::max(d->C, d->K) >= (1ll << 31)) return false;
    if (tile_rows) {
        const int64_t nrb = (M + 31) / 32;
        int rbw = (int)((nrb + nw * 2048 - 1) / (nw * 2048));      // at most ~2048 workgroups per column slice
        static const int rbw_min = getenv("DPFT_STREAM_RBW_MIN") ? atoi(getenv("DPFT_STREAM_RBW_MIN")) : 1;      // (see stream_geom)
        rbw = std::max(rbw, rbw_min);
        *tile_rows = 32 * nw * (rbw < 1 ? 1 : rbw);
    }
    return true;
}

int launch_stream1x1(const dpft_conv_desc* d, const float* x, const float* w, const float* pro_bn, float* y, float* stats,
                     const float* out_bn, const float* residual, int relu, hipStream_t st, unsigned long long* bns,
                     const BnSumsRef* pro_sums) {
    int tile_rows = 0, nb = 0, nw = 0;
    DPFT_REQUIRE(stream1x1_match(d, &tile_rows) && stream_geom(d->C, nb, nw), "conv 1x1 (streaming): problem not supported");
    DPFT_REQUIRE(!(out_bn && (stats || pro_bn)), "conv 1x1 (streaming): inference epilogue takes no prologue / statistics");
    StreamArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.y = y; a.pro = pro_bn; a.stats = stats; a.bns = bns;      /* (bns: `stats` only says that statistics are wanted) */ a.obn = out_bn; a.oadd = residual; a.orelu = relu;
    if (pro_sums) a.pro_s = *pro_sums;
    a.M = d->B * d->OH * d->OW; a.N = d->K;
    a.rbw = tile_rows / (32 * nw);
    a.nslices = d->K / nb;
    const int wgs = cdiv(a.M, tile_rows);
    const dim3 grid(wgs * a.nslices), block(nw * 64);
    const size_t lds = (size_t)nb * d->C * 4 + (size_t)3 * d->C * 4;
    auto go = [&](auto kernel) {
        static LdsGrant grant;
        (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
        hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
    };
#define STREAM_FORMS(KC_, NB_, NW_)                                                          \
    do {                                                                                     \
        if (out_bn) go(conv1x1_stream_kernel<KC_, NB_, NW_, false, true>);                   \
        else if (pro_bn) go(conv1x1_stream_kernel<KC_, NB_, NW_, true, false>);              \
        else go(conv1x1_stream_kernel<KC_, NB_, NW_, false, false>);                         \
    } while (0)
    if (d->C == 64) STREAM_FORMS(64, 256, 4);
    else if (nb == 128) STREAM_FORMS(128, 128, 4);
    else STREAM_FORMS(128, 256, 4);
#undef STREAM_FORMS
    return check_launch("conv 1x1 (streaming, short reduction)");
}

}  // namespace dpft
