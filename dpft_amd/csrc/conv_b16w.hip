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
#include "conv_epi8.h"

namespace dpft {

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
