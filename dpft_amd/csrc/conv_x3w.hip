// fp32 convolutions on the bf16 matrix cores (3 x bf16 split, six term products: conv_x3.hip) on EIGHT waves per 128 x 128
// tile, two per SIMD (round 6).
//
// Why.  igemm_x3_kernel runs four waves with 64 x 64 outputs each at 456 registers: ONE wave per SIMD, so every LDS or L2 wait
// of that wave lands in its SIMD's MFMA stream, and the loop had to be software-pipelined by hand (pieces pinned behind the
// MFMAs) to reach 125-150 TF of the pipe's ~330.  Here a wave owns 32 x 64 outputs (64 accumulator registers for the two
// accumulator sets, ~200 in all): two waves share a SIMD and cover each other -- while one waits for fragments or splits the
// next tile's operands, the other's MFMAs run -- and the code is plain: per 32-deep K-step a thread loads 4 quads (two K-steps
// ahead, two register sets), applies the producer's BatchNorm + ReLU where the conv has that prologue, splits them into three
// bf16 planes in LDS (the layout of conv_x3.hip: 64-byte plane rows, 16-byte chunks XOR-swizzled by (row / 4) % 4), one barrier
// per step.  Same arithmetic as igemm_x3_kernel: the leading term product in its own accumulator, the five small ones in a
// second, identical k order inside a K-group.  Epilogue: conv_epi8.h (all forms incl. the split-K fix-up).
#include "conv_core.h"
#include "conv_epi8.h"

namespace dpft {

typedef __bf16 x3w_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3w_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3w_cvt_pk(float x, float y) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x3w_f32x2{x, y}, x3w_bf16x2));
}

// PRO_: 0 no prologue, 1 BatchNorm + ReLU of the A operand, 2 the same on a conv with padding (taps that miss the image stay 0)
template <bool DGRAD, int PRO_>
__global__ __launch_bounds__(512) void igemm_x3w_kernel(IgemmArgs a) {
    constexpr bool PRO = PRO_ != 0, MASK = PRO_ == 2;
    constexpr int BM = 128, BN = 128, WGM = 4, WGN = 2, PBK = 32;
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;      // 1 x 2 blocks of 32 x 32 per wave
    constexpr int RPP = 64;                        // rows per loader pass: 8 lanes x 16 bytes per fp32 row segment, 8 rows per wave, 8 waves
    constexpr int AP = BM / RPP, BP = BN / RPP, NQ = AP + BP;
    constexpr int ROWB = PBK * 2;                  // bytes per plane row
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE, STAGE = A_BYTES + B_BYTES;
    constexpr int NG = PBK / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages | BatchNorm block [3][C] (PRO) ; epilogue staging
    char* const lds = reinterpret_cast<char*>(smem);

    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader geometry (as igemm_x3_kernel, 64 rows per pass) ----
    const int rp = tid >> 3;                       // row within a pass
    const int ch = tid & 7;                        // 16-byte fp32 chunk of the row segment: reduction indices 4 ch .. 4 ch + 3
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
        b_off[i] = n < a.N ? (unsigned)(n * a.Ktot + ch * 4) * 4u : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.N * a.Ktot * 4, 0x00020000);
    unsigned a_off[AP];
    unsigned a_valid_tap = 0;
    auto set_tap = [&](int tap) {
        const int ri = tap / ntap_s, si = tap - ri * ntap_s;
        const int tapoff = (DGRAD ? -1 : 1) * (ri * a.W + si) * a.C;
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool v = ((a_mask[i] >> ri) & (a_mask[i] >> (8 + si)) & 1u) != 0;
            a_off[i] = v ? (unsigned)(a_row[i] + tapoff + ch * 4) * 4u : OOB;
            valid |= v ? (1u << i) : 0u;
        }
        a_valid_tap = valid;
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
        so_a = __builtin_amdgcn_readfirstlane(run_c0 * 4);
        so_b = __builtin_amdgcn_readfirstlane((run_koff + run_c0) * 4);
    };
    auto advance = [&]() {
        run_c0 += PBK;
        if (run_c0 == a.C) {
            run_c0 = 0;
            ++run_tap;
            tap_dirty = true;
        }
    };
    int cons_c0 = (kt_begin - (kt_begin / cpt) * cpt) * PBK;      // channel offset of the tile being consumed (prologue parameters)

    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)smem;
    if constexpr (PRO) {
        float* tab = reinterpret_cast<float*>(lds + 2 * STAGE);
        fill_pro_table(tab, a.pro, a.pro_s, a.C, tid, 512);      // rows mean, scale, beta (from the BN block or the layer's column sums)
    }
    const float* const ptab = reinterpret_cast<const float*>(lds + 2 * STAGE) + ch * 4;

    // ---- register route: set s holds the tiles of parity s (two K-steps of flight time, as igemm_x3_kernel) ----
    f32x4 rq[2][NQ];
    unsigned rvalid[2] = {0u, 0u};
    auto load_tile = [&](auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int k = 0; k < AP; ++k) rq[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_off[k], so_a, 0));
#pragma unroll
        for (int k = 0; k < BP; ++k) rq[set][AP + k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off[k], so_b, 0));
        rvalid[set] = a_valid_tap;
    };
    // LDS destination of this thread's piece of a quad: row rp (+ 64 per pass), 8 bytes at half (ch & 1) of chunk (ch / 2) ^ key
    const int w_off = rp * ROWB + ((((ch >> 1) ^ ((rp >> 2) & 3)) << 4) | ((ch & 1) << 3));
    auto consume_quad = [&](auto SET, auto STG, auto K) {      // quad K of register set SET -> three plane pieces in stage STG
        constexpr int set = decltype(SET)::value, stg = decltype(STG)::value, k = decltype(K)::value;
        f32x4 v = rq[set][k];
        if constexpr (PRO && k < AP) {
            const f32x4 mu = *reinterpret_cast<const f32x4*>(ptab + cons_c0);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ptab + a.C + cons_c0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(ptab + 2 * a.C + cons_c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e] - mu[e], sc[e], sh[e]), 0.f);
            if constexpr (MASK) {      // padding stays exactly zero (BN(0) != 0)
                const unsigned keep = 0u - ((rvalid[set] >> k) & 1u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(__float_as_uint(v[e]) & keep);
            }
        }
        u32x2 p1, p2, p3;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const float x = v[2 * hh], y = v[2 * hh + 1];
            const unsigned q1 = x3w_cvt_pk(x, y);
            const float rx = x - __uint_as_float(q1 << 16), ry = y - __uint_as_float(q1 & 0xffff0000u);
            const unsigned q2 = x3w_cvt_pk(rx, ry);
            const float sx = rx - __uint_as_float(q2 << 16), sy = ry - __uint_as_float(q2 & 0xffff0000u);
            p1[hh] = q1;
            p2[hh] = q2;
            p3[hh] = x3w_cvt_pk(sx, sy);
        }
        constexpr int PL = k < AP ? A_PLANE : B_PLANE;
        char* dst = lds + stg * STAGE + (k < AP ? k * RPP * ROWB : A_BYTES + (k - AP) * RPP * ROWB) + w_off;
        *reinterpret_cast<u32x2*>(dst) = p1;
        *reinterpret_cast<u32x2*>(dst + PL) = p2;
        *reinterpret_cast<u32x2*>(dst + 2 * PL) = p3;
    };
    auto cons_advance = [&]() { cons_c0 = cons_c0 + PBK == a.C ? 0 : cons_c0 + PBK; };

    f32x16 acc[RB][CB], acl[RB][CB];      // leading term | the five small terms
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }

    // fragment addresses: lane l reads row l % 32 of a 32-row block, chunk 2 g + l / 32 of K-group g
    const int fkey = ((lane & 31) >> 2) & 3, hh_ = lane >> 5;
    unsigned a_ad[2][NG], b_ad[2][NG];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sw = ((2 * g + hh_) ^ fkey) << 4;
            a_ad[s][g] = lds_base + s * STAGE + (wm * RB * 32 + (lane & 31)) * ROWB + sw;
            b_ad[s][g] = lds_base + s * STAGE + A_BYTES + (wn * CB * 32 + (lane & 31)) * ROWB + sw;
        }
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    auto mfma_group = [&](auto STG, auto G) {
        constexpr int stg = decltype(STG)::value, g = decltype(G)::value;
        bf16x8_t af[3][RB], bf[3][CB];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                af[p][i] = __builtin_bit_cast(bf16x8_t, *(lds_f32x4*)(size_t)(a_ad[stg][g] + (unsigned)(p * A_PLANE + i * 32 * ROWB)));
#pragma unroll
            for (int j = 0; j < CB; ++j)
                bf[p][j] = __builtin_bit_cast(bf16x8_t, *(lds_f32x4*)(size_t)(b_ad[stg][g] + (unsigned)(p * B_PLANE + j * 32 * ROWB)));
        }
        // terms in the order  a3 b1, a2 b1, a1 b1, a2 b2, a1 b2, a1 b3  (plane indices 0 = leading), as igemm_x3_kernel
        constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    if (t == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                    else acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[t]][i], bf[TB[t]][j], acl[i][j], 0, 0, 0);
                }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // One K-step on stage STG (tile t).  Tile t + 1 (register set STG ^ 1) is split into the other stage between the two
    // K-groups' MFMAs; the load of tile t + 3 goes into the registers it frees.
    const int abl = a.ablate;      // tuning aid (DPFT_ABLATE, wrong results): 1 no global loads, 2 no split / LDS stores, 8 no MFMAs + fragment reads
    auto step = [&](auto STG, bool have1, bool have3) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        using OTH = std::integral_constant<int, (stg ^ 1)>;
        // The two waves of a SIMD (waves w and w + 4) run the step's halves in OPPOSITE order: one multiplies while the other
        // splits the next tile (vector ALU + LDS stores) -- in the same order both would split at the same time and leave the
        // matrix pipe idle meanwhile.  (The split writes the stage that is not being read: any order inside a step is legal.)
        if (wave < 4) {
            if (!(abl & 8)) { mfma_group(STG, S0{}); mfma_group(STG, S1{}); }
            if (have1 && !(abl & 2)) {
                static_for<NQ>([&](auto K) { consume_quad(OTH{}, OTH{}, K); });
                cons_advance();
            }
        } else {
            if (have1 && !(abl & 2)) {
                static_for<NQ>([&](auto K) { consume_quad(OTH{}, OTH{}, K); });
                cons_advance();
            }
            if (!(abl & 8)) { mfma_group(STG, S0{}); mfma_group(STG, S1{}); }
        }
        if (have3 && !(abl & 1)) {
            prep();
            load_tile(OTH{});
            advance();
        }
    };
    auto fence = [&]() {      // this wave's LDS writes have landed, then the workgroup meets; global loads stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    if constexpr (PRO) __syncthreads();      // the BatchNorm table is in LDS
    if (nsteps > 0) {      // tile 0 -> stage 0 (through set 0); tile 1 -> set 1, tile 2 -> set 0
        prep();
        load_tile(S0{});
        advance();
        static_for<NQ>([&](auto K) { consume_quad(S0{}, S0{}, K); });
        cons_advance();
        if (nsteps > 1) { prep(); load_tile(S1{}); advance(); }
        if (nsteps > 2) { prep(); load_tile(S0{}); advance(); }
    }
    fence();
    for (int t = 0; t < nsteps; t += 2) {
        step(S0{}, t + 1 < nsteps, t + 3 < nsteps);
        fence();
        if (t + 1 < nsteps) {
            step(S1{}, t + 2 < nsteps, t + 4 < nsteps);
            fence();
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] += acl[i][j];
    epilogue_w8<BM, BN, WGM, WGN, RB, CB>(a, acc, m0, n0, mt, split, smem);
}

// `a` as launch_igemm_x3 has prepared it (32-deep K-steps, tile counts for 128 x 128, splits)
int launch_igemm_x3w(IgemmArgs& a, bool dgrad, bool pro, hipStream_t st) {
    const dim3 grid(a.mtiles * a.ntiles * a.splits);
    const bool padded = a.kh * a.kw > 1 || a.pad > 0;
    const size_t stages = (size_t)2 * 3 * (128 + 128) * 64 + (pro ? (size_t)12 * a.C : 0);
    const size_t epi = std::max((size_t)(128 / 4) * (128 + 4) * 4, (size_t)512 * 8 * 4);
    const size_t lds = std::max(stages, epi);
    auto go = [&](auto kernel) {
        static LdsGrant grant;
        (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
        hipLaunchKernelGGL(kernel, grid, dim3(512), lds, st, a);
    };
    if (dgrad) go(igemm_x3w_kernel<true, 0>);
    else if (pro && padded) go(igemm_x3w_kernel<false, 2>);
    else if (pro) go(igemm_x3w_kernel<false, 1>);
    else go(igemm_x3w_kernel<false, 0>);
    return check_launch("conv igemm (3 x bf16 split, eight waves)");
}

}  // namespace dpft
