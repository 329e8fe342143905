// Epilogue of the eight-wave implicit-GEMM kernels (conv_b16w.hip, conv_x3w.hip): the tile leaves the accumulators in 64-row
// (BM / 4) chunks through one staging buffer -- or, for bf16 outputs without an additive operand, packed as bf16 by all waves
// at once -- and carries every form of conv_core.h's igemm_epilogue: split-K fix-up (ticket per output tile), BatchNorm tile
// statistics, bias, residual + ReLU byte mask, accumulate, the fused BatchNorm-backward reduction, the inference BN / add / ReLU form.
#pragma once
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
    if ((a.stats != nullptr || a.bns != nullptr) && !(a.ablate & 32)) {
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
            if (a.stats && n0 + tid < a.N) a.stats[((size_t)mt * 2 + 0) * a.N + n0 + tid] = mean;
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
            if (a.stats) a.stats[((size_t)mt * 2 + 1) * a.N + n0 + tid] = s;
            if (a.bns) {      // the statistics as column sums (common.h: BnSumsRef)
                const double m = (double)smean[tid], fc = (double)cnt;
                bn_sums_add(a.bns, a.N, n0 + tid, fc * m, (double)s + fc * m * m);
            }
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

}  // namespace dpft
