// bf16-operand implicit-GEMM convolutions on LARGE tiles (round 6): forward and data gradient of the mixed-precision
// plan (dpft_conv_desc.act16 = 2: bf16 activations, bf16 shadow weights, fp32 accumulation; BASELINE.json configs[4]).
//
// Why.  igemm_pipe_kernel<.., B16> (conv_pipe.h) runs 128 x 64 / 128 x 128 tiles on four waves: a wave owns 64 x 32 or
// 64 x 64 outputs, i.e. one ds_read_b128 fragment per MFMA (or 1.5), and a K-step of 64 is 8-16 MFMAs per wave (256-512
// cycles) that must cover 6-8 LDS-DMA pieces of ~60-100 issue cycles each (MI355X_MICROARCH.md price list) and a memory
// round trip: the loop is bound by the operand path (LDS-DMA issue, LDS reads, latency), the layer-3 problems at batch 8
// run at 200-460 TF of 2 500.  Here:
//   * 256 x 256 (or 256 x 128) outputs per workgroup on EIGHT waves, two per SIMD: a wave owns 64 x 128 (64 x 64) outputs --
//     6 fragments per 8 MFMAs (0.75 per MFMA; 1.0 on the narrow tile) -- and a K-step of 64 is 32 (16) MFMAs per wave,
//     64 (32) per SIMD = 2 048 (1 024) matrix cycles for 8 (6) LDS-DMA pieces per wave;
//   * the two waves of a SIMD cover each other: while one issues its loads or waits for fragments the other's MFMAs keep the
//     SIMD's matrix pipe fed (same rendezvous per K-step: one s_barrier);
//   * everything else as igemm_pipe_kernel: two LDS stages, operands by LDS-DMA (buffer_load ... lds; the descriptor's range
//     check supplies padding and row / channel tails), 128-byte rows with the 16-byte chunks XOR-swizzled by (row / 2) % 8,
//     precomputed fragment addresses + immediates, pinned instruction order.
// The epilogue walks the tile in four 64-row chunks through one 64 x BN staging buffer (a full 256 x 256 fp32 tile does not
// fit the LDS) and carries the forms the mixed-precision plan uses: BatchNorm tile statistics, bf16 / fp32 stores, residual
// + ReLU byte mask, accumulate, the fused BatchNorm-backward reduction, the inference BN / add / ReLU form, bias, and the
// in-launch split-K fix-up (ticket per output tile, as conv_core.h) -- which is what lets N = 256 problems (57 row tiles at
// batch 8) fill the chip with 256-row tiles.
#include "conv_core.h"

namespace dpft {

template <int BM, int BN, int WGM, int WGN, int RB, int CB>
__device__ __forceinline__ void epilogue_w8(const IgemmArgs& a, f32x16 (&acc)[RB][CB], int m0, int n0, int mt, int split,
                                            float* smem) {
    constexpr int NT = 512;
    constexpr int CR = BM / WGM;              // rows per chunk = rows of one wave row
    constexpr int LDC = BN + 4, C4 = BN / 4;
    constexpr int ITER = CR * C4 / NT;
    static_assert(CR == RB * 32 && CR * C4 % NT == 0 && NT % C4 == 0, "chunk geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    float* Cs = smem;      // [CR][LDC]
    __syncthreads();       // the operand stages are dead
    auto stage_chunk = [&](int c) {      // accumulators of wave row c -> Cs
        if (wm == c) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        Cs[row * LDC + wn * CB * 32 + cb * 32 + (lane & 31)] = acc[rb][cb][r];
                    }
        }
    };
    auto unstage_chunk = [&](int c) {
        if (wm == c) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        acc[rb][cb][r] = Cs[row * LDC + wn * CB * 32 + cb * 32 + (lane & 31)];
                    }
        }
    };
    // ---- split-K fix-up (protocol of conv_core.h: sc1 stores of the partial tile, ticket, the last workgroup sums in split order) ----
    const bool fix = a.partial != nullptr && a.sk_ticket != nullptr;
    if (fix) {
        const unsigned slab_bytes = (unsigned)a.M * (unsigned)a.N * 4u;
        __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.partial, 0, (int)(slab_bytes * (unsigned)a.splits), 0x00020000);
        for (int c = 0; c < WGM; ++c) {
            stage_chunk(c);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                const int m = m0 + c * CR + row;
                if (m < a.M && n0 + c4 * 4 < a.N) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs,
                                                           (int)(((unsigned)m * (unsigned)a.N + n0 + c4 * 4) * 4u),
                                                           (int)(slab_bytes * (unsigned)split), 16);      // aux 16 = sc1
                }
            }
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        int* ticket = a.sk_ticket + mt * a.ntiles + n0 / BN;
        if (tid == 0) *flag = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const bool is_last = *flag == a.splits - 1;
        __syncthreads();
        if (!is_last) return;
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        constexpr int FH = ITER < 4 ? ITER : 4;
        for (int c = 0; c < WGM; ++c) {
#pragma unroll 1
          for (int it0 = 0; it0 < ITER; it0 += FH) {
            f32x4 sum[FH];
            int voff[FH];
#pragma unroll
            for (int u = 0; u < FH; ++u) {
                const int idx = tid + (it0 + u) * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                const int m = m0 + c * CR + row;
                const bool ok = m < a.M && n0 + c4 * 4 < a.N;
                voff[u] = ok ? (int)(((unsigned)m * (unsigned)a.N + n0 + c4 * 4) * 4u) : -1;
                sum[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, voff[u], 0, 16));
            }
            for (int k = 1; k < a.splits; ++k) {
                f32x4 t[FH];
#pragma unroll
                for (int u = 0; u < FH; ++u)
                    t[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, voff[u], (int)(slab_bytes * (unsigned)k), 16));
#pragma unroll
                for (int u = 0; u < FH; ++u) sum[u] += t[u];
            }
#pragma unroll
            for (int u = 0; u < FH; ++u) {
                const int idx = tid + (it0 + u) * NT;
                const int row = idx / C4, c4 = idx - row * C4;
                *reinterpret_cast<f32x4*>(&Cs[row * LDC + c4 * 4]) = sum[u];
            }
          }
            __syncthreads();
            unstage_chunk(c);
            __syncthreads();
        }
    }
    const bool part = a.partial != nullptr && !fix;      // partial tiles for a reduction kernel of the caller
    // ---- per-tile column statistics of the raw conv output: (mean, M2) of the tile's rows ----
    if (a.stats != nullptr && !(a.ablate & 32)) {
        float* red = smem;               // [WGM][BN]
        float* smean = smem + WGM * BN;  // [BN]
        const int cnt = min(BM, a.M - m0);
        const int rbase = m0 + wm * CR + 4 * (lane >> 5);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float s = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + rb * 32 + (r & 3) + 8 * (r >> 2);
                    s += (row < a.M) ? acc[rb][cb][r] : 0.f;
                }
            s += __shfl_xor(s, 32);
            if (lane < 32) red[wm * BN + wn * CB * 32 + cb * 32 + lane] = s;
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < WGM; ++i) s += red[i * BN + tid];
            const float mean = s / (float)cnt;
            smean[tid] = mean;
            if (n0 + tid < a.N) a.stats[((size_t)mt * 2 + 0) * a.N + n0 + tid] = mean;
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const float mean = smean[wn * CB * 32 + cb * 32 + (lane & 31)];
            float s = 0.f;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + rb * 32 + (r & 3) + 8 * (r >> 2);
                    const float d = acc[rb][cb][r] - mean;
                    s += (row < a.M) ? d * d : 0.f;
                }
            s += __shfl_xor(s, 32);
            if (lane < 32) red[wm * BN + wn * CB * 32 + cb * 32 + lane] = s;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < WGM; ++i) s += red[i * BN + tid];
            a.stats[((size_t)mt * 2 + 1) * a.N + n0 + tid] = s;
        }
        __syncthreads();
    }
    // ---- bf16 output without an additive operand (forward convs; data gradients, also with the fused BatchNorm-backward
    // reduction): the WHOLE tile is rounded in registers and staged as bf16 by all eight waves at once, then stored with
    // 16-byte lanes (8 channels per thread).  The MFMA layout gives a lane one column and 16 rows; lanes 2i / 2i+1 swap one
    // value per row pair (DPP quad_perm) so that each holds (column 2i, column 2i+1) of alternate rows: one ds_write_b32 per
    // two outputs.  Staged rows are BN / 2 + 16 dwords: rows r and r + 1 (the even / odd lanes of one write) land in
    // different bank halves, and the 16-byte reads of the store loop are conflict-free.  (The forms that ADD to the result --
    // residual, accumulate, inference BN -- keep the fp32 staging below: they round once, after the addition.)
    if (a.y16 && !part && a.res_src == nullptr && !a.accumulate && a.obn == nullptr && a.bias == nullptr && !(a.ablate & 256)) {
        constexpr int LD2 = BN / 2 + 16, C8 = BN / 8, RSTEP = NT / C8, PIT = BM / RSTEP;
        static_assert(NT % C8 == 0 && BM % RSTEP == 0, "packed store geometry");
        unsigned* Cp = reinterpret_cast<unsigned*>(smem);      // [BM][LD2] pairs of bf16
        {
            const bool odd = (lane & 1) != 0;
            const int rl = wm * CR + 4 * (lane >> 5) + (odd ? 1 : 0);
            const int dc = (wn * CB * 32 + (lane & 31)) >> 1;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr) {      // row pair (r, r + 1), r = 2 pr
                        const float e = acc[rb][cb][2 * pr], o = acc[rb][cb][2 * pr + 1];
                        const float send = odd ? e : o;
                        const float got = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
                        const float lo = odd ? got : e, hi = odd ? o : got;
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
                        const int r = 2 * pr;
                        const int row = rl + rb * 32 + (r & 3) + 8 * (r >> 2);
                        Cp[row * LD2 + dc + cb * 16] = pk;
                    }
        }
        __syncthreads();
        const int c8 = tid % C8, row0 = tid / C8;
        const int bc8 = n0 + c8 * 8;
        const bool bnr8 = a.bnr_sums != nullptr;
        float s0[8], s1[8], mu8[8], is8[8], sc8[8], be8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu8[e] = 0.f; is8[e] = 0.f; sc8[e] = 0.f; be8[e] = 0.f; }
        if (bnr8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mu8[e] = a.bnr_bnp[bc8 + e];
                is8[e] = a.bnr_bnp[3 * a.N + bc8 + e];
                if (a.bnr_self_mask) { sc8[e] = a.bnr_bnp[a.N + bc8 + e]; be8[e] = a.bnr_bnp[2 * a.N + bc8 + e]; }
            }
        }
        __bf16* const out16 = reinterpret_cast<__bf16*>(a.y);
        const __bf16* const y16p = reinterpret_cast<const __bf16*>(a.bnr_y);
        constexpr int PH = PIT < 4 ? PIT : 4;
#pragma unroll 1
        for (int it0 = 0; it0 < PIT && !(a.ablate & 64); it0 += PH) {
            u32x4 v[PH], yv[PH];
            unsigned mk[PH];
            size_t off[PH];
            bool ok[PH];
#pragma unroll
            for (int u = 0; u < PH; ++u) {
                const int row = row0 + (it0 + u) * RSTEP;
                const int m = m0 + row;
                ok[u] = m < a.M;
                off[u] = ok[u] ? out_pixel(a, m) * a.N + bc8 : 0;
                yv[u] = u32x4{0u, 0u, 0u, 0u};
                mk[u] = 0u;
                if (bnr8 && ok[u]) {
                    yv[u] = *reinterpret_cast<const u32x4*>(y16p + off[u]);
                    if (a.bnr_mask8) mk[u] = *reinterpret_cast<const unsigned short*>(a.bnr_mask8 + (off[u] >> 2));
                }
                v[u] = *reinterpret_cast<const u32x4*>(&Cp[row * LD2 + c8 * 4]);
            }
#pragma unroll
            for (int u = 0; u < PH; ++u) {
                if (!ok[u]) continue;
                *reinterpret_cast<u32x4*>(out16 + off[u]) = v[u];
                if (bnr8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned dv = v[u][e >> 1], yw = yv[u][e >> 1];
                        float d = __uint_as_float((e & 1) ? (dv & 0xffff0000u) : (dv << 16));      // the ROUNDED value: what a separate pass reads back
                        const float yy = __uint_as_float((e & 1) ? (yw & 0xffff0000u) : (yw << 16));
                        if (a.bnr_mask8) d = ((mk[u] >> ((e & 3) + 8 * (e >> 2))) & 1u) ? d : 0.f;      // one mask byte per 4 channels
                        else if (a.bnr_self_mask) d = fmaf(yy - mu8[e], sc8[e], be8[e]) > 0.f ? d : 0.f;
                        s0[e] += d;
                        s1[e] += d * ((yy - mu8[e]) * is8[e]);
                    }
                }
            }
        }
        if (bnr8) {
            __syncthreads();      // the staged tile has been read
            float* red = smem;    // [NT][16]
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s0[e]; red[tid * 16 + 8 + e] = s1[e]; }
            __syncthreads();
            if (tid < BN && n0 + tid < a.N) {
                const int ch = tid >> 3, e = tid & 7;
                float t0 = 0.f, t1 = 0.f;
                for (int g = 0; g < RSTEP; ++g) {
                    t0 += red[(g * C8 + ch) * 16 + e];
                    t1 += red[(g * C8 + ch) * 16 + 8 + e];
                }
                atomicAdd(a.bnr_sums + n0 + tid, t0);
                atomicAdd(a.bnr_sums + a.N + n0 + tid, t1);
            }
        }
        return;
    }
    // ---- the tile, 64 rows at a time: staged through LDS -> full 16-byte (8-byte in bf16) lanes along rows ----
    float* __restrict__ out = part ? a.partial + (size_t)split * a.M * a.N : a.y;
    const bool add_bias = (a.bias != nullptr) && !part;
    const bool accum = a.accumulate && !part;
    const bool y16 = a.y16 && !part;
    const bool bnr = (a.bnr_sums != nullptr) && !part;
    const bool resid = (a.res_src != nullptr) && !part;
    const bool obn = (a.obn != nullptr) && !part;
    const bool oadd = obn && a.oadd != nullptr;
    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0, bmu = bs0, bis = bs0, bsc = bs0, bbe = bs0;
    const int c4t = tid % C4;                 // this thread's channel quad (the same in every pass: NT % C4 == 0)
    const int bc = n0 + c4t * 4;
    const bool col_ok = bc < a.N;
    if (bnr && col_ok) {
        bmu = *reinterpret_cast<const f32x4*>(a.bnr_bnp + bc);
        bis = *reinterpret_cast<const f32x4*>(a.bnr_bnp + 3 * a.N + bc);
        if (a.bnr_self_mask) {
            bsc = *reinterpret_cast<const f32x4*>(a.bnr_bnp + a.N + bc);
            bbe = *reinterpret_cast<const f32x4*>(a.bnr_bnp + 2 * a.N + bc);
        }
    }
    f32x4 omu = bs0, osc = bs0, obe = bs0, bias4 = bs0;
    if (obn && col_ok) {
        omu = *reinterpret_cast<const f32x4*>(a.obn + bc);
        osc = *reinterpret_cast<const f32x4*>(a.obn + a.N + bc);
        obe = *reinterpret_cast<const f32x4*>(a.obn + 2 * a.N + bc);
    }
    if (add_bias && col_ok) bias4 = *reinterpret_cast<const f32x4*>(a.bias + bc);
    // (a pass of IH quads per thread at a time: the whole chunk's operands in registers next to 128 accumulators spilled)
    constexpr int IH = ITER < 4 ? ITER : 4;
    auto chunk_pass = [&](auto H16) {
        constexpr bool h16 = decltype(H16)::value;
        for (int c = 0; c < WGM; ++c) {
            if (!(a.ablate & 128)) stage_chunk(c);
            __syncthreads();
#pragma unroll 1
            for (int it0 = 0; it0 < ITER && !(a.ablate & 64); it0 += IH) {
                f32x4 old[IH], yv[IH];
                unsigned mk[IH];
#pragma unroll
                for (int u = 0; u < IH; ++u) {
                    const int row = (tid + (it0 + u) * NT) / C4;
                    const int m = m0 + c * CR + row;
                    old[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    yv[u] = old[u];
                    mk[u] = 0u;
                    if (m < a.M && col_ok) {
                        const size_t off = out_pixel(a, m) * a.N + bc;
                        if (bnr) {
                            yv[u] = load4_act(a.bnr_y, off, h16);
                            if (a.bnr_mask8) mk[u] = a.bnr_mask8[off >> 2];
                        }
                        if (oadd) {
                            old[u] = load4_act(a.oadd, off, h16);
                        } else if (resid) {
                            const f32x4 g = load4_act(a.res_src, off, h16);
                            if (a.res_mask8) {
                                const unsigned rm = a.res_mask8[off >> 2];
#pragma unroll
                                for (int e = 0; e < 4; ++e) old[u][e] = ((rm >> e) & 1u) ? g[e] : 0.f;
                            } else {
                                const f32x4 o = load4_act(a.res_mask, off, h16);
#pragma unroll
                                for (int e = 0; e < 4; ++e) old[u][e] = o[e] > 0.f ? g[e] : 0.f;
                            }
                        } else if (accum) {
                            old[u] = load4_act(out, off, h16);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < IH; ++u) {
                    const int row = (tid + (it0 + u) * NT) / C4;
                    const int m = m0 + c * CR + row;
                    if (m < a.M && col_ok) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4t * 4]);
                        if (add_bias) v += bias4;
                        if (obn) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e] - omu[e], osc[e], obe[e]);
                        }
                        if (accum || resid || oadd) v += old[u];
                        if (obn && a.orelu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        const size_t ooff = out_pixel(a, m) * a.N + bc;
                        store4_act(out, ooff, v, h16);
                        if (bnr) {
                            f32x4 d = v;
                            if (h16) d = __builtin_convertvector(__builtin_convertvector(v, bf16x4s), f32x4);      // what a separate pass would read back
                            if (a.bnr_mask8) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) d[e] = ((mk[u] >> e) & 1u) ? d[e] : 0.f;
                            } else if (a.bnr_self_mask) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) d[e] = fmaf(yv[u][e] - bmu[e], bsc[e], bbe[e]) > 0.f ? d[e] : 0.f;
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                bs0[e] += d[e];
                                bs1[e] += d[e] * ((yv[u][e] - bmu[e]) * bis[e]);
                            }
                        }
                    }
                }
            }
            __syncthreads();      // Cs is rewritten by the next chunk
        }
    };
    if (y16) chunk_pass(std::true_type{});
    else chunk_pass(std::false_type{});
    if (bnr) {      // column sums over the tile's rows (NT / C4 threads per channel quad), one atomic pair per column
        float* red = smem;      // [NT][8]
        *reinterpret_cast<f32x4*>(red + tid * 8) = bs0;
        *reinterpret_cast<f32x4*>(red + tid * 8 + 4) = bs1;
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            const int ch = tid >> 2, e = tid & 3;
            float t0 = 0.f, t1 = 0.f;
            for (int g = 0; g < NT / C4; ++g) {
                t0 += red[(g * C4 + ch) * 8 + e];
                t1 += red[(g * C4 + ch) * 8 + 4 + e];
            }
            atomicAdd(a.bnr_sums + n0 + tid, t0);
            atomicAdd(a.bnr_sums + a.N + n0 + tid, t1);
        }
    }
}

// ABL (tuning aid, DPFT_ABLATE; wrong results): 1 no global loads, 4 no epilogue, 8 no MFMAs, 16 no fragment reads
template <int BM, int BN, bool DGRAD, int ABL = 0>
__global__ __launch_bounds__(512) void igemm_b16w_kernel(IgemmArgs a) {
    constexpr int NW = 8, WGM = 4, WGN = 2, PBK = 64, EB = 2, EPC = 8;
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;
    constexpr int CH = PBK / EPC;           // 8 chunks of 16 bytes per LDS row
    constexpr int RW = 64 / CH;             // 8 rows per wave instruction
    constexpr int RPP = NW * RW;            // 64 rows per pass of the 8 waves
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr int ROWB = PBK * EB;          // 128 bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NG = PBK / 16;            // K-groups (one v_mfma_f32_32x32x16_bf16 deep) per step
    static_assert(RB >= 1 && CB >= 1 && AP >= 1 && BP >= 1 && STAGE <= 65536, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // max(2 * STAGE, epilogue staging)
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem;

    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader geometry (as igemm_pipe_kernel, 64 rows per pass) ----
    const int rp = RW * wave + lane / CH;
    const int pos = lane % CH;
    const int lkey = (rp >> 1) & 7;
    const int chunk = pos ^ lkey;
    const bool sub = DGRAD && a.sub_step > 1;
    const int roww = sub ? a.sub_ow : a.OW;
    const int ohw = sub ? a.sub_oh * a.sub_ow : a.OH * a.OW;
    const int ntap_s = sub ? a.sub_ns : a.kw;
    const int ntap_r = sub ? a.sub_nr : a.kh;
    constexpr unsigned OOB = 0x80000000u;
    int a_row[AP];
    unsigned a_mask[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + rp + RPP * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw;
        const int rem = mm - b * ohw;
        const int oh = rem / roww, ow = rem - oh * roww;
        int h0, w0;
        if (!DGRAD) {
            h0 = oh * a.stride - a.pad;
            w0 = ow * a.stride - a.pad;
        } else if (sub) {
            h0 = oh + (a.sub_ph + a.pad - a.sub_r0) / a.sub_step;
            w0 = ow + (a.sub_pw + a.pad - a.sub_s0) / a.sub_step;
        } else {
            h0 = oh + a.pad;
            w0 = ow + a.pad;
        }
        a_row[i] = ((b * a.H + h0) * a.W + w0) * a.C;
        unsigned mask = 0;
        for (int ri = 0; ri < ntap_r; ++ri) {
            const int hi = DGRAD ? h0 - ri : h0 + ri;
            mask |= (ok && (unsigned)hi < (unsigned)a.H) ? (1u << ri) : 0u;
        }
        for (int si = 0; si < ntap_s; ++si) {
            const int wi = DGRAD ? w0 - si : w0 + si;
            mask |= (ok && (unsigned)wi < (unsigned)a.W) ? (256u << si) : 0u;
        }
        a_mask[i] = mask;
    }
    unsigned b_off[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + rp + RPP * i;
        b_off[i] = n < a.N ? (unsigned)(n * a.Ktot + chunk * EPC) * (unsigned)EB : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * EB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.N * a.Ktot * EB, 0x00020000);
    unsigned a_off[AP];
    auto set_tap = [&](int tap) {
        const int ri = tap / ntap_s, si = tap - ri * ntap_s;
        const int tapoff = (DGRAD ? -1 : 1) * (ri * a.W + si) * a.C;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool v = ((a_mask[i] >> ri) & (a_mask[i] >> (8 + si)) & 1u) != 0;
            a_off[i] = v ? (unsigned)(a_row[i] + tapoff + chunk * EPC) * (unsigned)EB : OOB;
        }
    };

    const int kt_begin = split * a.ksteps_per_split;
    const int kt_end = min(a.ksteps, kt_begin + a.ksteps_per_split);
    const int nsteps = max(kt_end - kt_begin, 0);
    const int cpt = a.C / PBK;      // K-steps per filter tap
    int run_tap = kt_begin / cpt, run_c0 = (kt_begin - run_tap * cpt) * PBK, run_koff = 0;
    bool tap_dirty = true;
    int so_a = 0, so_b = 0;
    auto prep = [&]() {
        if (tap_dirty) {
            set_tap(run_tap);
            const int ri = run_tap / ntap_s, si = run_tap - ri * ntap_s;
            run_koff = (sub ? (a.sub_r0 + a.sub_step * ri) * a.kw + a.sub_s0 + a.sub_step * si : run_tap) * a.C;
            tap_dirty = false;
        }
        so_a = __builtin_amdgcn_readfirstlane(run_c0 * EB);
        so_b = __builtin_amdgcn_readfirstlane((run_koff + run_c0) * EB);
    };
    auto advance = [&]() {
        run_c0 += PBK;
        if (run_c0 == a.C) {
            run_c0 = 0;
            ++run_tap;
            tap_dirty = true;
        }
    };
    // The loop below runs the K-steps in PAIRS (one code path: with separate tails for an odd / even remainder the register
    // allocator spilled the accumulators there); a pair whose second tile does not exist loads zeros for it (every offset out
    // of range -> the descriptor's range check writes zeros into the stage) and multiplies them.
    constexpr int NOPS = AP + BP;
    unsigned tile_oob = 0;      // 0, or the out-of-range marker for every lane: the tile being loaded lies beyond the reduction
    auto vmem_op = [&](auto STG, auto K) {
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value;
        if constexpr (ABL & 1) return;
        if constexpr (k < AP) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, lds0 + stg * STAGE + (RPP * k + RW * wave) * ROWB, 16,
                                                     (int)(a_off[k] | tile_oob), so_a, 0, 0);
        } else {
            constexpr int i = k - AP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, lds0 + stg * STAGE + A_BYTES + (RPP * i + RW * wave) * ROWB, 16,
                                                     (int)(b_off[i] | tile_oob), so_b, 0, 0);
        }
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read addresses: lane l reads row l % 32 of a 32-row block, K-group kg, half h = l / 32 -> chunk 2 kg + h
    const int fkey = (lane >> 1) & 7;
    const int h = lane >> 5;
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const unsigned lds_base = (unsigned)(size_t)lds0;
    unsigned a_ad[2][NG], b_ad[2][NG];      // per stage: a stage is up to 64 KB, the ds_read immediate 16 bits
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kg = 0; kg < NG; ++kg) {
            const int sw = ((2 * kg + h) ^ fkey) * 16;
            a_ad[s][kg] = lds_base + s * STAGE + (wm * RB * 32 + (lane & 31)) * ROWB + sw;
            b_ad[s][kg] = lds_base + s * STAGE + A_BYTES + (wn * CB * 32 + (lane & 31)) * ROWB + sw;
            asm volatile("" : "+v"(a_ad[s][kg]), "+v"(b_ad[s][kg]));
        }

    constexpr int MPG = RB * CB, SLOTS = NG * MPG;
    static_assert(NOPS <= SLOTS, "more loads than MFMA slots");
    constexpr int OPSP = SLOTS / NOPS >= 2 ? 2 : 1;      // loads behind every OPSP-th MFMA of the step's first part
    int issued = 0;      // tiles whose loads have been issued
    auto step = [&](auto STG, auto MORE) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool more = decltype(MORE)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (more) {
            tile_oob = issued < nsteps ? 0u : OOB;
            if (issued < nsteps) prep();
            ++issued;
        }
        f32x4 af[2][RB] = {}, bf[2][CB] = {};
        auto frags = [&](auto SET, auto KG) {
            constexpr int set = decltype(SET)::value, kg = decltype(KG)::value;
            if constexpr (ABL & 16) return;
#pragma unroll
            for (int i = 0; i < RB; ++i) af[set][i] = *(lds_f32x4*)(size_t)(a_ad[stg][kg] + (unsigned)(i * 32 * ROWB));
#pragma unroll
            for (int j = 0; j < CB; ++j) bf[set][j] = *(lds_f32x4*)(size_t)(b_ad[stg][kg] + (unsigned)(j * 32 * ROWB));
        };
        frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, i = w / CB, j = w % CB;
            if constexpr (w == 0 && g + 1 < NG)
                frags(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{});
            if constexpr (!(ABL & 8))
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[g & 1][i]),
                                                                __builtin_bit_cast(bf16x8, bf[g & 1][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more && (sl % OPSP) == OPSP - 1 && sl / OPSP < NOPS) {
                vmem_op(OTHER{}, std::integral_constant<int, (sl / OPSP < NOPS ? sl / OPSP : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if constexpr (more) advance();
    };
    auto fence = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    if (nsteps > 0) {
        prep();
        static_for<NOPS>([&](auto K) { vmem_op(S0{}, K); });
        advance();
        issued = 1;
    }
    fence();
    for (int s_ = 0; s_ < nsteps; s_ += 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, T{});
        fence();
    }
    if constexpr (ABL & 4) {
        if (acc[0][0][0] == 12345.678f) a.y[0] = 1.f;
        return;
    }
    epilogue_w8<BM, BN, WGM, WGN, RB, CB>(a, acc, m0, n0, mt, split, smem);
}

template <int BM, int BN>
static size_t b16w_lds() {
    const size_t stages = (size_t)2 * (BM + BN) * 64 * 2;
    const size_t staging = std::max((size_t)(BM / 4) * (BN + 4) * 4, (size_t)BM * (BN / 2 + 16) * 4);      // fp32 chunk | packed bf16 tile
    const size_t stats = (size_t)(4 + 1) * BN * 4, red = (size_t)512 * 8 * 4;
    return std::max(std::max(stages, staging), std::max(stats, red));
}

// `a` prepared as for igemm_pipe_kernel<BM, BN, ..> (mtiles, ntiles, splits, ksteps in units of 64, ksteps_per_split)
int launch_igemm_b16w(IgemmArgs& a, int bm, int bn, bool dgrad, hipStream_t st) {
    DPFT_REQUIRE(a.x16 && a.w16 && a.pro == nullptr, "conv (bf16, large tiles): bf16 operands without prologue only");
    DPFT_REQUIRE(a.bnf_acc == nullptr && a.bnf_slab == 0, "conv (bf16, large tiles): no fused BatchNorm finalize");
    DPFT_REQUIRE((a.N & 3) == 0 && a.C % 64 == 0, "conv (bf16, large tiles): N %% 4 == 0 and C %% 64 == 0");
    DPFT_REQUIRE(bm == 256 && (bn == 256 || bn == 128), "conv (bf16, large tiles): tile %d x %d", bm, bn);
    const dim3 grid(a.mtiles * a.ntiles * a.splits), block(512);
    auto go = [&](auto kernel, size_t lds) {
        static LdsGrant grant;
        (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
        hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
    };
    if (a.ablate && !dgrad) {      // tuning aid
#define ABL_CASE(V) case V: if (bn == 256) go(igemm_b16w_kernel<256, 256, false, V>, b16w_lds<256, 256>()); else go(igemm_b16w_kernel<256, 128, false, V>, b16w_lds<256, 128>()); break;
        switch (a.ablate & 31) { case 0: if (bn == 256) go(igemm_b16w_kernel<256, 256, false>, b16w_lds<256, 256>()); else go(igemm_b16w_kernel<256, 128, false>, b16w_lds<256, 128>()); break; ABL_CASE(1) ABL_CASE(4) ABL_CASE(5) ABL_CASE(8) ABL_CASE(12) ABL_CASE(13) ABL_CASE(28) ABL_CASE(29) default: break; }
#undef ABL_CASE
        return check_launch("conv igemm (bf16, large tiles, ablation)");
    }
    if (bn == 256) {
        if (dgrad) go(igemm_b16w_kernel<256, 256, true>, b16w_lds<256, 256>());
        else go(igemm_b16w_kernel<256, 256, false>, b16w_lds<256, 256>());
    } else {
        if (dgrad) go(igemm_b16w_kernel<256, 128, true>, b16w_lds<256, 128>());
        else go(igemm_b16w_kernel<256, 128, false>, b16w_lds<256, 128>());
    }
    return check_launch("conv igemm (bf16 operands, 256-row tiles, 8 waves)");
}

}  // namespace dpft
