// Software-pipelined fp32 implicit-GEMM kernel (included by conv.hip after IgemmArgs / igemm_epilogue).
//
// Why a second main loop.  Ablation of igemm_vec_kernel on the layer-3 problems (tools/conv_ablate.sh, round 3): the MFMA
// loop alone runs at ~88 % of what the grid's quantisation allows, but the step is  compute -> barrier -> [wait for the
// register-staged tile, BatchNorm+ReLU prologue, 8 x ds_write_b128, issue the next loads] -> barrier  and in that second
// phase no wave of the workgroup has an MFMA in the pipe: +25 % on the 3x3 forward (21 us of 112), +10 % on the data
// gradients, and the loads' issue time on top.  Here a K-step has ONE barrier and nothing between the MFMAs but what has
// to be there:
//   * two LDS stages; the tile of step t+1 is brought in WHILE step t multiplies;
//   * operands that need no arithmetic (weights always; the A operand of data gradients and of prologue-free forward
//     convs) never touch a VGPR: `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per wave instruction) writes them straight
//     into the stage.  The hardware range check of the buffer descriptor still supplies the zeros of the padding, of rows
//     >= M and of weight rows >= N;
//   * LDS-DMA writes lane-linearly (wave base + 16 * lane), so rows cannot be padded against bank conflicts.  Instead the
//     16-byte chunks of a row are XOR-swizzled: lane (row r, position p) FETCHES chunk p ^ key(r) from global memory (a
//     permutation inside one 128 / 256-byte row segment: coalescing is unchanged) and a fragment read of chunk c of row r
//     goes to position c ^ key(r).  key(r) = r % 16 for 256-byte rows, (r / 2) % 8 for 128-byte rows: the 16 lanes that a
//     ds_read_b128 services together hit 16 different bank groups;
//   * an A operand with the fused BatchNorm+ReLU prologue takes the register route (it needs the vector ALU), but its
//     loads are issued at the START of the step and consumed behind the last MFMA groups of the same step -- the waits
//     land a K-step after the issue;
//   * per-lane LDS read addresses are precomputed (one VGPR per K-group and operand) and everything else of an address
//     is an instruction immediate: the steady state has no address arithmetic at all.
// The fp32 MFMA shares the vector ALUs on gfx950, so what remains per step is MFMA time + the
// prologue's arithmetic; loads, LDS-DMA, ds_reads and the barrier ride in the MFMAs' shadow.
//
// Same numerics as igemm_vec_kernel: identical k order inside a K-group, identical fragment permutation, fp32 MFMA.
#pragma once

namespace dpft {

// (static_for: conv_core.h)

// B16: BOTH operands are bf16 in memory (activations in bf16 storage, dpft_conv_desc.act16 = 2: the caller also passes
// bf16 weights) -- the same kernel with 2-byte elements: a 16-byte chunk is 8 reduction indices, a K-group is 16 of them
// and one v_mfma_f32_32x32x16_bf16 (fp32 accumulation) per 32x32 block and group.  No arithmetic touches an operand on
// its way to the matrix cores, so nothing but LDS-DMA feeds the stages -- unless the conv carries the producer's BatchNorm + ReLU
// (PRO, round 6): then the A operand takes the register route as in fp32 (8 bf16 per 16-byte quad: widened, normalised, rounded
// back to bf16 -- the value a materialising pass would have stored), its parameters from a [3][C] table in LDS behind the
// stages; the bf16 MFMA does not share the vector ALUs, so the prologue's arithmetic rides beside it.
// EPF: the epilogue's operands (residual data gradient with the fused BatchNorm-backward reduction, both byte masks) are
// requested behind the MFMAs of selected K-steps (EpiPrefetch, conv.hip) instead of after the last one.
template <int BM, int BN, int WGM, int WGN, int PBK, bool DGRAD, bool PRO, bool B16 = false, int EPF = 0>
__global__ __launch_bounds__(256) void igemm_pipe_kernel(IgemmArgs a) {
    static_assert(EPF == 0 || (EPF == 3 ? (!DGRAD && !PRO) : (DGRAD && !PRO)), "epilogue prefetch: data gradients (1, 2) / inference forward (3)");      // EPF = EpiPrefetch::MODE
    constexpr int EB = B16 ? 2 : 4;         // bytes per element
    constexpr int EPC = 16 / EB;            // elements per 16-byte chunk
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;
    constexpr int CH = PBK / EPC;           // 16-byte chunks per LDS row
    constexpr int RW = 64 / CH;             // rows one wave instruction covers
    constexpr int RPP = 4 * RW;             // rows per pass of the 4 waves
    constexpr int AP = BM / RPP, BP = BN / RPP;
    constexpr int ROWB = PBK * EB;          // bytes per LDS row
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NG = PBK / (2 * EPC);     // K-groups (two chunks: one per lane half) per step
    static_assert(CH == 8 || CH == 16, "rows of 128 or 256 bytes");
    static_assert(WGM * WGN == 4 && RB >= 1 && CB >= 1 && AP >= 1 && BP >= 1, "bad tile");
    static_assert(2 * STAGE <= 65536, "LDS offsets must fit the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // max(2 * STAGE, epilogue staging)
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem;

    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader geometry: lane -> (row of the pass, position in the row); the chunk it fetches is swizzled ----------
    const int rp = RW * wave + lane / CH;                 // row within a pass, 0 .. RPP-1
    const int pos = lane % CH;                            // LDS position (lane-linear destination)
    const int lkey = (CH == 16) ? (rp & 15) : ((rp >> 1) & 7);
    const int chunk = pos ^ lkey;                         // global 16-byte chunk of the row segment
    // ---- operand addressing (see igemm_vec_kernel: per-row tap masks, offsets rebuilt only when the tap changes) ---
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
    unsigned a_valid_tap = 0;
    unsigned long long a_inv_tap[PRO ? AP : 1] = {};      // per loader quad: lanes of this wave whose tap misses the image
    auto set_tap = [&](int tap) {
        const int ri = tap / ntap_s, si = tap - ri * ntap_s;
        const int tapoff = (DGRAD ? -1 : 1) * (ri * a.W + si) * a.C;
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool v = ((a_mask[i] >> ri) & (a_mask[i] >> (8 + si)) & 1u) != 0;
            a_off[i] = v ? (unsigned)(a_row[i] + tapoff + chunk * EPC) * (unsigned)EB : OOB;
            valid |= v ? (1u << i) : 0u;
            if constexpr (PRO) a_inv_tap[i] = __builtin_amdgcn_ballot_w64(!v);
        }
        a_valid_tap = valid;
    };

    const int kt_begin = split * a.ksteps_per_split;
    const int kt_end = min(a.ksteps, kt_begin + a.ksteps_per_split);
    const int nsteps = max(kt_end - kt_begin, 0);
    const int cpt = a.C / PBK;      // K-steps per filter tap
    // position of the NEXT tile to be issued: (tap, channel offset); prep() turns it into the loads' scalar operands
    int run_tap = kt_begin / cpt, run_c0 = (kt_begin - run_tap * cpt) * PBK, run_koff = 0;
    bool tap_dirty = true;
    int so_a = 0, so_b = 0;          // byte soffsets of the A / weight loads of the next tile
    auto prep = [&]() {
        if (tap_dirty) {
            set_tap(run_tap);
            const int ri = run_tap / ntap_s, si = run_tap - ri * ntap_s;
            run_koff = (sub ? (a.sub_r0 + a.sub_step * ri) * a.kw + a.sub_s0 + a.sub_step * si : run_tap) * a.C;
            tap_dirty = false;
        }
        // (scalar operands of the loads: tell the compiler -- a VGPR here costs a waterfall loop per load)
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

    // register route of the A operand (PRO only)
    f32x4 ra[PRO ? AP : 1], p_mu, p_sc, p_sh;
    f32x4 q_mu = {0.f, 0.f, 0.f, 0.f}, q_sc = q_mu, q_sh = q_mu;      // B16: channels 4..7 of the lane's 8-channel quad
    // PRO: rows mean, scale, beta of the producer's BatchNorm as a table [3][C] in LDS behind the stages -- copied from the BN block
    // or derived from the layer's column sums (common.h: fill_pro_table)
    const float* const ptab16 = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + 2 * STAGE);
    if constexpr (PRO) {
        fill_pro_table(reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + 2 * STAGE), a.pro, a.pro_s, a.C, tid, 256);
        __syncthreads();
    }
    unsigned ra_valid = 0;
    unsigned long long ra_inv[PRO ? AP : 1] = {};      // wave-uniform copies of a_inv_tap for the tile in the registers

    // The loads of a tile as NOPS separately placeable operations (the main loop puts one behind each of the first MFMAs
    // of a step: a vector-memory instruction takes tens of cycles to issue, which an MFMA in the pipe hides and an idle
    // pipe does not).  Order: A operand, weights.
    constexpr int NOPS = AP + BP;      // (the prologue parameters come from the LDS table, read with the tile's first load)
    auto vmem_op = [&](auto STG, auto K) {
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value;
        if constexpr (k < AP) {
            if constexpr (PRO) {
                ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_off[k], so_a, 0));
                if constexpr (k == 0) ra_valid = a_valid_tap;
                ra_inv[k] = a_inv_tap[k];
                if constexpr (B16 && k == 0) {      // this tile's parameters (channels so_a / 2 + 8 chunk .. + 7): consumed at the end of the step
                    const float* t = ptab16 + (so_a >> 1) + chunk * 8;
                    p_mu = *reinterpret_cast<const f32x4*>(t);          q_mu = *reinterpret_cast<const f32x4*>(t + 4);
                    p_sc = *reinterpret_cast<const f32x4*>(t + a.C);    q_sc = *reinterpret_cast<const f32x4*>(t + a.C + 4);
                    p_sh = *reinterpret_cast<const f32x4*>(t + 2 * a.C); q_sh = *reinterpret_cast<const f32x4*>(t + 2 * a.C + 4);
                }
                if constexpr (!B16 && k == 0) {      // fp32: channels so_a / 4 + 4 chunk .. + 3
                    const float* t = ptab16 + (so_a >> 2) + chunk * 4;
                    p_mu = *reinterpret_cast<const f32x4*>(t);
                    p_sc = *reinterpret_cast<const f32x4*>(t + a.C);
                    p_sh = *reinterpret_cast<const f32x4*>(t + 2 * a.C);
                }
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, lds0 + stg * STAGE + (RPP * k + RW * wave) * ROWB, 16,
                                                         (int)a_off[k], so_a, 0, 0);
            }
        } else {
            constexpr int i = k - AP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, lds0 + stg * STAGE + A_BYTES + (RPP * i + RW * wave) * ROWB, 16,
                                                     (int)b_off[i], so_b, 0, 0);
        }
    };
    // BatchNorm + ReLU of register quad i, written to its (lane-linear) place in stage `stg`.  The launcher takes this
    // kernel only for pro_relu != 0 (every prologue of the ResNet plan has the ReLU).  Padding must stay exactly zero
    // (BN(0) != 0): the select runs only in waves that have such a lane for this quad and tap (a scalar test of a ballot
    // taken when the tap changed) -- interior pixels, i.e. nearly all of them, pay no vector instruction for it.
    char* const w_ptr = reinterpret_cast<char*>(smem) + rp * ROWB + pos * 16;
    auto consume = [&](auto STG, auto I) {
        constexpr int stg = decltype(STG)::value, i = decltype(I)::value;
        f32x4 val = ra[i];
        if constexpr (B16) {      // 8 bf16: widen, BatchNorm + ReLU in fp32, round back (RNE) -- what the materialising pass stores
            const u32x4 raw = __builtin_bit_cast(u32x4, val);
            f32x4 lo = {__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
            f32x4 hi = {__uint_as_float(raw[2] << 16), __uint_as_float(raw[2] & 0xffff0000u), __uint_as_float(raw[3] << 16), __uint_as_float(raw[3] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = fmaxf(fmaf(lo[e] - p_mu[e], p_sc[e], p_sh[e]), 0.f);
                hi[e] = fmaxf(fmaf(hi[e] - q_mu[e], q_sc[e], q_sh[e]), 0.f);
            }
            const u32x2 pl = __builtin_bit_cast(u32x2, __builtin_convertvector(lo, bf16x4)), ph = __builtin_bit_cast(u32x2, __builtin_convertvector(hi, bf16x4));
            val = __builtin_bit_cast(f32x4, u32x4{pl[0], pl[1], ph[0], ph[1]});
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = fmaxf(fmaf(val[e] - p_mu[e], p_sc[e], p_sh[e]), 0.f);
        }
        if (ra_inv[i] != 0ull) {
            asm volatile("" ::: "memory");      // keeps this a branch (the compiler would turn it back into selects)
            if (!((ra_valid >> i) & 1u)) val = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(w_ptr + stg * STAGE + RPP * i * ROWB) = val;
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue-operand prefetch (EPF) ------------------------------------------------------------------------------
    // Quads outside the tensor read the tile's first element instead (never used): no branch around a load.  The element
    // offset is rebuilt at every use from an opaque copy of the thread index.
    constexpr int EP_ITER = BM * (BN / 4) / 256;
    using PFT = EpiPrefetch<EP_ITER, B16, EPF>;
    PFT pf;
    auto pf_op = [&](auto IT_, auto KIND_) __attribute__((always_inline)) {
        if constexpr (EPF != 0) {
            constexpr int it = decltype(IT_)::value, kind = decltype(KIND_)::value;
            constexpr int C4 = BN / 4;
            int t = tid;
            asm volatile("" : "+v"(t));
            const int idx = t + it * 256;
            const int row = idx / C4, c4 = idx - row * C4;
            const bool ok = m0 + row < a.M && n0 + c4 * 4 < a.N;
            const unsigned el = ok ? (unsigned)(m0 + row) * (unsigned)a.N + (unsigned)(n0 + c4 * 4)      // (stride-1 data gradient: output pixel = row)
                                   : (unsigned)m0 * (unsigned)a.N + (unsigned)n0;
            using Q = typename PFT::Q;
            if constexpr (EPF == 3) {      // forward: the residual of the inference epilogue
                if constexpr (kind == 1) pf.g[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.oadd) + (size_t)el * EB);
            } else if constexpr (kind == 0) pf.y[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.bnr_y) + (size_t)el * EB);
            else if constexpr (kind == 1) { if constexpr (EPF == 1) pf.g[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.res_src) + (size_t)el * EB); }
            else if constexpr (kind == 2) { if (EPF == 1 || a.bnr_mask8) pf.mk[it] = a.bnr_mask8[el >> 2]; }
            else { if constexpr (EPF == 1) pf.rm[it] = a.res_mask8[el >> 2]; }
        }
    };

    // fragment read addresses: lane l reads row l % 32 of a 32-row block, K-group kg, half h = l / 32 -> chunk 2 * kg + h
    const int fkey = (CH == 16) ? (lane & 15) : ((lane >> 1) & 7);
    const int h = lane >> 5;
    // 32-bit LDS addresses with the array's base folded in: a fragment read is one VGPR + an instruction immediate
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    const unsigned lds_base = (unsigned)(size_t)lds0;
    unsigned a_ad[NG], b_ad[NG];
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
        const int sw = ((2 * kg + h) ^ fkey) * 16;
        a_ad[kg] = lds_base + (wm * RB * 32 + (lane & 31)) * ROWB + sw;
        b_ad[kg] = lds_base + A_BYTES + (wn * CB * 32 + (lane & 31)) * ROWB + sw;
        // opaque: otherwise the compiler keeps (row base, swizzle) apart and re-adds them in front of every read -- 2 * NG
        // vector adds per K-step on the ALUs the fp32 MFMA needs
        asm volatile("" : "+v"(a_ad[kg]), "+v"(b_ad[kg]));
    }

    // One K-step on stage STG; MORE: the next tile is brought into the other stage meanwhile.  The MFMAs of the step are
    // SLOTS = NG * 4 * RB * CB; the instruction order is pinned (sched_barrier after every slot): fragment reads of the
    // next K-group in front of a group's first MFMA, one load behind each of the first NOPS MFMAs, the prologue's quads
    // behind MFMAs of the last groups.
    constexpr int EPG = B16 ? 1 : 4;              // MFMAs per 32x32 block and K-group (fp32: one per k pair)
    constexpr int MPG = EPG * RB * CB, SLOTS = NG * MPG;
    constexpr int LASTG = (NG >= 8) ? 4 : 2;      // K-groups (the last ones of a step) that carry the prologue
    constexpr int QPG = PRO ? AP / LASTG : 1;     // register quads per such group
    static_assert(NOPS <= SLOTS, "more loads than MFMA slots");
    static_assert(!PRO || (AP % LASTG == 0 && QPG >= 1 && QPG <= MPG), "prologue schedule");
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    auto step = [&](auto STG, auto MORE) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool more = decltype(MORE)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (more) prep();
        f32x4 af[2][RB], bf[2][CB];
        auto frags = [&](auto SET, auto KG) {
            constexpr int set = decltype(SET)::value, kg = decltype(KG)::value;
#pragma unroll
            for (int i = 0; i < RB; ++i) af[set][i] = *(lds_f32x4*)(size_t)(a_ad[kg] + (unsigned)(stg * STAGE + i * 32 * ROWB));
#pragma unroll
            for (int j = 0; j < CB; ++j) bf[set][j] = *(lds_f32x4*)(size_t)(b_ad[kg] + (unsigned)(stg * STAGE + j * 32 * ROWB));
        };
        frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, e = w / (RB * CB), ij = w % (RB * CB), i = ij / CB, j = ij % CB;
            if constexpr (w == 0 && g + 1 < NG)
                frags(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{});
            if constexpr (B16)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[g & 1][i]),
                                                                    __builtin_bit_cast(bf16x8, bf[g & 1][j]), acc[i][j], 0, 0, 0);
            else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][e], bf[g & 1][j][e], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more && sl < NOPS) {
                vmem_op(OTHER{}, std::integral_constant<int, (sl < NOPS ? sl : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (more && PRO) {
                constexpr int gq = g - (NG - LASTG);               // which of the last groups (negative: not one)
                constexpr int sp = MPG / QPG;                      // quads of a group are spaced over its MFMAs
                if constexpr (gq >= 0 && (w % sp) == sp - 1 && (w / sp) < QPG) {
                    consume(OTHER{}, std::integral_constant<int, (gq >= 0 ? gq * QPG + w / sp : 0)>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        });
        if constexpr (more) advance();
    };
    auto fence = [&]() {      // everything this wave issued has landed (LDS-DMA counts on vmcnt), then the workgroup meets
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    if (nsteps > 0) {      // first tile: nothing to hide the loads behind
        prep();
        static_for<NOPS>([&](auto K) { vmem_op(S0{}, K); });
        if constexpr (PRO) static_for<AP>([&](auto I) { consume(S0{}, I); });
        advance();
    }
    if constexpr (EPF != 0) {
        // All of the thread's epilogue operands are requested here, with the first tile: one memory round trip (the first
        // fence's) serves both, and the epilogue starts with its operands in registers.  (Tried: groups of 16 requests behind
        // the MFMAs of later K-steps with a fence that lets the newest 16 fly -- the register allocator gave a second request
        // site of a group its own destination registers and copied, 240 VGPRs instead of 200, and the isolated kernel time
        // did not move; with everything up front it went from 61 to 54 us on the layer-3 fp32 problem, 69 -> 40 us in bf16.)
        static_for<EP_ITER * 4>([&](auto O) {
            constexpr int o = decltype(O)::value;
            pf_op(std::integral_constant<int, o / 4>{}, std::integral_constant<int, o % 4>{});
        });
    }
    fence();
    int s_ = 0;
    for (; s_ + 2 < nsteps; s_ += 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, T{});
        fence();
    }
    if (nsteps - s_ == 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, F{});
    } else if (nsteps - s_ == 1) {
        step(S0{}, F{});
    }
    if constexpr (EPF != 0) igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256, PFT>(a, acc, m0, n0, mt, split, smem, &pf);
    else igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256>(a, acc, m0, n0, mt, split, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient, same pipeline.  dW[n][tap][c] = sum over pixels p of dY[p][n] * act(x)[p @ tap][c]: both operands are
// pixel-major in memory AND in LDS ([pixel][channels]: a lane's MFMA operand is one channel of one pixel, consecutive
// lanes read consecutive words -- no swizzle needed), the reduction runs over pixels in steps of PK.  dY always arrives by
// LDS-DMA; x too unless it carries the producer's BatchNorm+ReLU (register route, consumed behind the last MFMAs of the
// step).  Fragment reads are one base VGPR per operand + immediates.
// ---------------------------------------------------------------------------------------------------------------------
template <int BMn, int BNc, int WGM, int WGN, int PK, bool PRO>
__global__ __launch_bounds__(256) void wgrad_pipe_kernel(WgradArgs a) {
    constexpr int RB = BMn / WGM / 32, CB = BNc / WGN / 32;
    constexpr int YL = BMn / 4, XL = BNc / 4;              // lanes (16-byte chunks) per pixel row
    constexpr int YRW = 64 / YL, XRW = 64 / XL;            // pixel rows per wave instruction
    constexpr int YRPP = 4 * YRW, XRPP = 4 * XRW;          // pixel rows per pass of the 4 waves
    constexpr int YP = PK / YRPP, XP = PK / XRPP;
    constexpr int Y_BYTES = PK * BMn * 4, X_BYTES = PK * BNc * 4, STAGE = Y_BYTES + X_BYTES;
    static_assert(WGM * WGN == 4 && RB >= 1 && CB >= 1 && YP >= 1 && XP >= 1 && YL <= 64 && XL <= 64, "bad wgrad tile");
    static_assert(2 * STAGE <= 65536, "LDS offsets must fit the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int nwg = a.ktiles * a.ctiles * a.taps * a.splits;
    int bid = xcd_remap(blockIdx.x, nwg);
    const int split = bid / (a.ktiles * a.ctiles * a.taps);
    bid -= split * (a.ktiles * a.ctiles * a.taps);
    const int tap = bid / (a.ktiles * a.ctiles);
    bid -= tap * (a.ktiles * a.ctiles);
    const int kt_ = bid / a.ctiles, ct_ = bid - kt_ * a.ctiles;
    const int n0 = kt_ * BMn, c0 = ct_ * BNc;
    const int r = tap / a.kw, s = tap - r * a.kw;

    const int yr = YRW * wave + lane / YL, ych = lane % YL;      // this lane's pixel row within a pass / channel chunk
    const int xr = XRW * wave + lane / XL, xch = lane % XL;
    const bool y_ok = (n0 + ych * 4) < a.K;
    const bool x_ok = (c0 + xch * 4) < a.C;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (PRO && x_ok) {
        mu = *reinterpret_cast<const f32x4*>(a.pro + c0 + xch * 4);
        sc = *reinterpret_cast<const f32x4*>(a.pro + a.C + c0 + xch * 4);
        sh = *reinterpret_cast<const f32x4*>(a.pro + 2 * a.C + c0 + xch * 4);
    }
    const int ohw = a.OH * a.OW;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.M * a.K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * 4, 0x00020000);
    unsigned y_voff[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) y_voff[i] = y_ok ? (unsigned)((YRPP * i + yr) * a.K + n0 + ych * 4) * 4u : OOB;
    const int ps_begin = split * a.psteps_per_split;
    const int ps_end = min(a.psteps, ps_begin + a.psteps_per_split);
    const int nsteps = max(ps_end - ps_begin, 0);
    // x offsets.  The pixel -> (image, row, column) -> input pixel / validity arithmetic is ~25 vector instructions per
    // loader row, and every one of the XL lanes of a row would repeat it every step, on the ALUs the fp32 MFMA needs.  It
    // is done ONCE per workgroup instead: a table in LDS holds, for every pixel of this workgroup's pixel range, the byte
    // offset of the input pixel its tap reads (or the out-of-range marker: padding, pixel tail).  A step then costs one
    // ds_read_b32 + one add per loader row.
    unsigned* const tbl = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + 2 * STAGE);
    {
        const int npix = nsteps * PK;
        for (int idx = tid; idx < npix; idx += 256) {
            const int p = ps_begin * PK + idx;
            const int b = p / ohw;
            const int rem = p - b * ohw;
            const int oh = rem / a.OW, ow = rem - oh * a.OW;
            const int hi = oh * a.stride - a.pad + r, wi = ow * a.stride - a.pad + s;
            const bool v = p < a.M && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            tbl[idx] = v ? ((((unsigned)b * a.H + hi) * a.W + wi) * a.C) * 4u : OOB;
        }
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) const unsigned lds_u32;
    unsigned tbl_ad = (unsigned)(size_t)lds0 + 2 * STAGE + xr * 4;      // LDS address of this lane's row in the next tile
    const unsigned x_lane = x_ok ? (unsigned)(c0 + xch * 4) * 4u : OOB;  // channel chunks beyond C: out of range as well
    int next_ps = ps_begin;
    int so_y = 0;
    unsigned xoffs[XP];
    unsigned long long x_inv[PRO ? XP : 1] = {};
    auto prep = [&]() {
        so_y = __builtin_amdgcn_readfirstlane(next_ps * PK * a.K * 4);
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const unsigned t = *(lds_u32*)(size_t)(tbl_ad + (unsigned)(XRPP * i * 4));
            xoffs[i] = t + x_lane;      // marker + anything small stays beyond every tensor (< 2 GiB)
            if constexpr (PRO) x_inv[i] = __builtin_amdgcn_ballot_w64((int)xoffs[i] < 0);
        }
        tbl_ad += PK * 4;
        ++next_ps;
    };
    f32x4 rx[PRO ? XP : 1];
    unsigned rxo[PRO ? XP : 1] = {};      // the offsets the quads in rx were loaded with (validity, read in the rare branch)
    unsigned long long rx_inv[PRO ? XP : 1] = {};
    constexpr int NOPS = YP + XP;
    auto vmem_op = [&](auto STG, auto K) {      // dY first: the x offsets come out of LDS a few MFMAs after prep()
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value;
        if constexpr (k < YP) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, lds0 + stg * STAGE + (YRPP * k + YRW * wave) * BMn * 4, 16,
                                                     (int)y_voff[k], so_y, 0, 0);
        } else {
            constexpr int i = k - YP;
            if constexpr (PRO) {
                rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)xoffs[i], 0, 0));
                rxo[i] = xoffs[i];
                rx_inv[i] = x_inv[i];
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, lds0 + stg * STAGE + Y_BYTES + (XRPP * i + XRW * wave) * BNc * 4, 16,
                                                         (int)xoffs[i], 0, 0, 0);
            }
        }
    };
    char* const xw_ptr = reinterpret_cast<char*>(smem) + Y_BYTES + xr * BNc * 4 + xch * 16;
    auto consume = [&](auto STG, auto I) {
        constexpr int stg = decltype(STG)::value, i = decltype(I)::value;
        f32x4 val = rx[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = fmaxf(fmaf(val[e] - mu[e], sc[e], sh[e]), 0.f);
        if (rx_inv[i] != 0ull) {      // padding / pixel tail: BN(0) != 0 (rare: a scalar test skips the selects)
            asm volatile("" ::: "memory");
            if ((int)rxo[i] < 0) val = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(xw_ptr + stg * STAGE + XRPP * i * BNc * 4) = val;
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // volatile: keeps each fragment read a ds_read_b32 with a 16-bit immediate offset -- merged into ds_read2_b32 (8-bit
    // offsets) every pair would need a vector add for its base
    typedef __attribute__((address_space(3))) const volatile float lds_f32;
    const unsigned lds_base = (unsigned)(size_t)lds0;
    unsigned y_ad = lds_base + ((lane >> 5) * BMn + wm * RB * 32 + (lane & 31)) * 4;
    unsigned x_ad = lds_base + Y_BYTES + ((lane >> 5) * BNc + wn * CB * 32 + (lane & 31)) * 4;
    asm volatile("" : "+v"(y_ad), "+v"(x_ad));

    constexpr int NPAIR = PK / 2, MPP = RB * CB, SLOTS = NPAIR * MPP, PF = 2, NS = PF + 1;
    static_assert(NOPS <= SLOTS, "more loads than MFMA slots");
    // the prologue's XP quads: behind MFMAs of the last quarter of the step, evenly spaced
    constexpr int CSTART = SLOTS - SLOTS / 4, CSTEP = (SLOTS / 4) / (PRO ? XP : 1);
    static_assert(!PRO || CSTEP >= 1, "prologue schedule");
    auto step = [&](auto STG, auto MORE) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool more = decltype(MORE)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (more) prep();
        float av[NS][RB], bv[NS][CB];
        auto frags = [&](auto SET, auto KK) {
            constexpr int set = decltype(SET)::value, kk = decltype(KK)::value;
#pragma unroll
            for (int i = 0; i < RB; ++i) av[set][i] = *(lds_f32*)(size_t)(y_ad + (unsigned)(stg * STAGE + (kk * 2 * BMn + i * 32) * 4));
#pragma unroll
            for (int j = 0; j < CB; ++j) bv[set][j] = *(lds_f32*)(size_t)(x_ad + (unsigned)(stg * STAGE + (kk * 2 * BNc + j * 32) * 4));
        };
        static_for<PF>([&](auto Q) { frags(Q, Q); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int kk = sl / MPP, w = sl % MPP, i = w / CB, j = w % CB;
            if constexpr (w == 0 && kk + PF < NPAIR)
                frags(std::integral_constant<int, ((kk + PF) % NS)>{}, std::integral_constant<int, (kk + PF < NPAIR ? kk + PF : 0)>{});
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % NS][i], bv[kk % NS][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more && sl < NOPS) {
                vmem_op(OTHER{}, std::integral_constant<int, (sl < NOPS ? sl : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (more && PRO) {
                if constexpr (sl >= CSTART && (sl - CSTART) % CSTEP == 0 && (sl - CSTART) / CSTEP < XP) {
                    consume(OTHER{}, std::integral_constant<int, (sl >= CSTART ? ((sl - CSTART) / CSTEP) % XP : 0)>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        });
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
        if constexpr (PRO) static_for<XP>([&](auto I) { consume(S0{}, I); });
    }
    fence();
    int s_ = 0;
    for (; s_ + 2 < nsteps; s_ += 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, T{});
        fence();
    }
    if (nsteps - s_ == 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, F{});
    } else if (nsteps - s_ == 1) {
        step(S0{}, F{});
    }
    // epilogue: stage one wave-row of the tile at a time through LDS -> full 16-byte stores along c (as wgrad_vec_kernel)
    float* __restrict__ out = a.partial ? a.partial + (size_t)split * a.K * a.taps * a.C : a.dw;
    constexpr int RP = RB * 32;
    constexpr int LDC = BNc + 4;
    static_assert(RP * LDC * 4 <= 2 * STAGE, "wgrad epilogue staging does not fit the operand LDS");
    float* Cs = smem;
    for (int hh = 0; hh < WGM; ++hh) {
        __syncthreads();
        if (wm == hh) {
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                        Cs[row * LDC + wn * CB * 32 + j * 32 + (lane & 31)] = acc[i][j][q];
                    }
        }
        __syncthreads();
        constexpr int C4 = BNc / 4;
        for (int idx = tid; idx < RP * C4; idx += 256) {
            const int row = idx / C4, c4 = idx - row * C4;
            const int n = n0 + hh * RP + row, c = c0 + c4 * 4;
            if (n < a.K && c < a.C)
                *reinterpret_cast<f32x4*>(out + ((size_t)n * a.taps + tap) * a.C + c) =
                    *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient on bf16 tensors (dpft_conv_desc.act16 = 2: dY and the materialised activation are bf16 in memory, no
// prologue).  Same pipeline: two LDS stages filled by LDS-DMA, a per-workgroup pixel -> input-offset table, one barrier
// per step.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive PIXELS of one channel per lane, the tensors are pixel-major
// ([pixel][channel]) and LDS-DMA cannot transpose -- ds_read_b64_tr_b16 does: within a 16-lane group lane s supplies the
// 8-byte address of (pixel s / 4, channels 4 (s % 4) .. + 3) and lane i receives channel i of the four pixels
// (tools/probes/tr16_probe.hip).  Two such reads (pixels +0..3, +4..7) make one operand of the MFMA.
// Bank conflicts: a group reads 4 pixel rows x 32 bytes and rows are 256 (128) bytes apart, so the 16-byte chunks of a row
// are XOR-swizzled by the pixel: key = (pixel % 4) * 4 for 16-chunk rows, ((pixel / 2) % 2) * 4 for 8-chunk rows -- the
// loader lane at position p of a row FETCHES chunk p ^ key (coalescing unchanged), the reader looks for chunk c at c ^ key;
// the 32 lanes of a half-wave then cover all 64 banks once.  The key does not depend on the K-group / half / stage, so a
// lane keeps ONE address per 32-channel block and everything else is an instruction immediate.
// ---------------------------------------------------------------------------------------------------------------------
template <int BMn, int BNc, int WGM, int WGN, int PK>
__global__ __launch_bounds__(256) void wgrad_pipe16_kernel(WgradArgs a) {
    constexpr int RB = BMn / WGM / 32, CB = BNc / WGN / 32;
    constexpr int YL = BMn / 8, XL = BNc / 8;              // lanes (16-byte chunks of 8 channels) per pixel row
    constexpr int YRW = 64 / YL, XRW = 64 / XL;            // pixel rows per wave instruction
    constexpr int YRPP = 4 * YRW, XRPP = 4 * XRW;          // pixel rows per pass of the 4 waves
    constexpr int YP = PK / YRPP, XP = PK / XRPP;
    constexpr int Y_ROWB = BMn * 2, X_ROWB = BNc * 2;
    constexpr int Y_BYTES = PK * Y_ROWB, X_BYTES = PK * X_ROWB, STAGE = Y_BYTES + X_BYTES;
    static_assert(WGM * WGN == 4 && RB >= 1 && CB >= 1 && YP >= 1 && XP >= 1, "bad wgrad tile");
    static_assert((YL == 16 || YL == 8) && (XL == 16 || XL == 8), "rows of 256 or 128 bytes");
    static_assert(2 * STAGE <= 65536 && PK % 16 == 0, "LDS offsets must fit the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const lds0 = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int nwg = a.ktiles * a.ctiles * a.taps * a.splits;
    int bid = xcd_remap(blockIdx.x, nwg);
    const int split = bid / (a.ktiles * a.ctiles * a.taps);
    bid -= split * (a.ktiles * a.ctiles * a.taps);
    const int tap = bid / (a.ktiles * a.ctiles);
    bid -= tap * (a.ktiles * a.ctiles);
    const int kt_ = bid / a.ctiles, ct_ = bid - kt_ * a.ctiles;
    const int n0 = kt_ * BMn, c0 = ct_ * BNc;
    const int r = tap / a.kw, s = tap - r * a.kw;

    // loaders: pixel row within a pass / position in the row; the chunk fetched is the position XOR the row's key
    const int yr = YRW * wave + lane / YL, xr = XRW * wave + lane / XL;
    const int ykey = YL == 16 ? (yr & 3) << 2 : ((yr >> 1) & 1) << 2;
    const int xkey = XL == 16 ? (xr & 3) << 2 : ((xr >> 1) & 1) << 2;
    const int ych = (lane % YL) ^ ykey, xch = (lane % XL) ^ xkey;
    const bool y_ok = (n0 + ych * 8) < a.K;
    const bool x_ok = (c0 + xch * 8) < a.C;
    const int ohw = a.OH * a.OW;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.M * a.K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * 2, 0x00020000);
    unsigned y_voff[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) y_voff[i] = y_ok ? (unsigned)((YRPP * i + yr) * a.K + n0 + ych * 8) * 2u : OOB;
    const int ps_begin = split * a.psteps_per_split;
    const int ps_end = min(a.psteps, ps_begin + a.psteps_per_split);
    const int nsteps = max(ps_end - ps_begin, 0);
    unsigned* const tbl = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(smem) + 2 * STAGE);
    {      // input-pixel byte offsets of this workgroup's pixel range (see wgrad_pipe_kernel)
        const int npix = nsteps * PK;
        for (int idx = tid; idx < npix; idx += 256) {
            const int p = ps_begin * PK + idx;
            const int b = p / ohw;
            const int rem = p - b * ohw;
            const int oh = rem / a.OW, ow = rem - oh * a.OW;
            const int hi = oh * a.stride - a.pad + r, wi = ow * a.stride - a.pad + s;
            const bool v = p < a.M && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            tbl[idx] = v ? ((((unsigned)b * a.H + hi) * a.W + wi) * a.C) * 2u : OOB;
        }
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) const unsigned lds_u32;
    unsigned tbl_ad = (unsigned)(size_t)lds0 + 2 * STAGE + xr * 4;
    const unsigned x_lane = x_ok ? (unsigned)(c0 + xch * 8) * 2u : OOB;
    int next_ps = ps_begin;
    int so_y = 0;
    unsigned xoffs[XP], traw[XP];
    // Every LDS read of the main loop is inline assembly with hand-placed s_waitcnt lgkmcnt: a compiler-visible LDS read
    // behind an LDS-DMA gets an s_waitcnt vmcnt(0) in front of it (the DMA's LDS store "may alias"), which would expose the
    // HBM latency of the tile in flight in the middle of every step.
    auto tbl_read = [&](auto I) {
        constexpr int i = decltype(I)::value;
        unsigned& dst = traw[i];      // (named outside the asm: operands alone do not capture in a generic lambda)
        const unsigned ad = tbl_ad;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(XRPP * i * 4));
    };
    auto tbl_use = [&](auto I) {      // behind a wait that covers the reads of tbl_read
        constexpr int i = decltype(I)::value;
        unsigned& t = traw[i];
        asm volatile("" : "+v"(t));
        xoffs[i] = t + x_lane;      // marker + anything small stays beyond every tensor (< 2 GiB)
    };
    auto prep = [&]() {      // issue: the table entries of the next tile's loader rows
        so_y = __builtin_amdgcn_readfirstlane(next_ps * PK * a.K * 2);
        static_for<XP>(tbl_read);
        tbl_ad += PK * 4;
        ++next_ps;
    };
    constexpr int NOPS = YP + XP;
    auto vmem_op = [&](auto STG, auto K) {
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value;
        if constexpr (k < YP) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, lds0 + stg * STAGE + (YRPP * k + YRW * wave) * Y_ROWB, 16,
                                                     (int)y_voff[k], so_y, 0, 0);
        } else {
            constexpr int i = k - YP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, lds0 + stg * STAGE + Y_BYTES + (XRPP * i + XRW * wave) * X_ROWB, 16,
                                                     (int)xoffs[i], 0, 0, 0);
        }
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // transpose-read addresses: supplier lane (group G = lane / 16, s = lane % 16) -> pixel (G / 2) * 8 + s / 4 of the K-group,
    // channel quad s % 4 of the 16 channels (G % 2) of a 32-channel block
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const unsigned lds_base = (unsigned)(size_t)lds0;
    unsigned y_ad[RB], x_ad[CB];
    {
        const int G = lane >> 4, sl = lane & 15, r2 = sl >> 2, q = sl & 3;
        const int pix = (G >> 1) * 8 + r2;
        const int yk = YL == 16 ? r2 << 2 : (r2 >> 1) << 2;
        const int xk = XL == 16 ? r2 << 2 : (r2 >> 1) << 2;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int c = (wm * RB + i) * 4 + 2 * (G & 1) + (q >> 1);
            y_ad[i] = lds_base + pix * Y_ROWB + ((c ^ yk) << 4) + (q & 1) * 8;
        }
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int c = (wn * CB + j) * 4 + 2 * (G & 1) + (q >> 1);
            x_ad[j] = lds_base + Y_BYTES + pix * X_ROWB + ((c ^ xk) << 4) + (q & 1) * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) asm volatile("" : "+v"(y_ad[i]));
#pragma unroll
    for (int j = 0; j < CB; ++j) asm volatile("" : "+v"(x_ad[j]));

    constexpr int NG = PK / 16, MPG = RB * CB, SLOTS = NG * MPG, RPG = 2 * (RB + CB);      // RPG: reads per K-group
    static_assert(NOPS <= SLOTS, "more loads than MFMA slots");
    static_assert(XRPP * (XP - 1) * 4 < 65536, "table offsets");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // LDS returns data in order: "at most n reads outstanding" = everything older than the last n has landed.  The
    // registers are operands of the wait so that nothing that uses them can be placed in front of it.
    auto wait_set = [&](auto N, u32x2 (&ya)[RB][2], u32x2 (&xa)[CB][2]) {
        constexpr int n = decltype(N)::value;
        static_assert(RB <= 2 && CB <= 2, "operand lists below");
        if constexpr (RB == 2 && CB == 2)
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(ya[0][0]), "+v"(ya[0][1]), "+v"(ya[1][0]), "+v"(ya[1][1]), "+v"(xa[0][0]),
                         "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]) : "n"(n));
        else
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ya[0][0]), "+v"(ya[0][1]), "+v"(xa[0][0]), "+v"(xa[0][1]) : "n"(n));
    };
    auto step = [&](auto STG, auto MORE) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool more = decltype(MORE)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (more) prep();
        u32x2 av[2][RB][2], bv[2][CB][2];
        auto frags = [&](auto SET, auto KK) {
            constexpr int set = decltype(SET)::value, kk = decltype(KK)::value;
            static_for<RB * 2>([&](auto Q) {
                constexpr int i = decltype(Q)::value / 2, h = decltype(Q)::value % 2;
                u32x2& dst = av[set][i][h];
                const unsigned ad = y_ad[i];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(stg * STAGE + (kk * 16 + h * 4) * Y_ROWB));
            });
            static_for<CB * 2>([&](auto Q) {
                constexpr int j = decltype(Q)::value / 2, h = decltype(Q)::value % 2;
                u32x2& dst = bv[set][j][h];
                const unsigned ad = x_ad[j];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(stg * STAGE + (kk * 16 + h * 4) * X_ROWB));
            });
        };
        frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, i = w / CB, j = w % CB;
            if constexpr (w == 0) {
                if constexpr (g + 1 < NG)
                    frags(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{});
                wait_set(std::integral_constant<int, (g + 1 < NG ? RPG : 0)>{}, av[g & 1], bv[g & 1]);
                if constexpr (more && g == 0) static_for<XP>(tbl_use);      // older than this K-group's fragments: landed too
                __builtin_amdgcn_sched_barrier(0);
            }
            const u32x4 fa = __builtin_shufflevector(av[g & 1][i][0], av[g & 1][i][1], 0, 1, 2, 3);
            const u32x4 fb = __builtin_shufflevector(bv[g & 1][j][0], bv[g & 1][j][1], 0, 1, 2, 3);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                                acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more && sl < NOPS) {
                vmem_op(OTHER{}, std::integral_constant<int, (sl < NOPS ? sl : 0)>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<XP>(tbl_use);
        static_for<NOPS>([&](auto K) { vmem_op(S0{}, K); });
    }
    fence();
    int s_ = 0;
    for (; s_ + 2 < nsteps; s_ += 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, T{});
        fence();
    }
    if (nsteps - s_ == 2) {
        step(S0{}, T{});
        fence();
        step(S1{}, F{});
    } else if (nsteps - s_ == 1) {
        step(S0{}, F{});
    }
    // epilogue: as wgrad_pipe_kernel (fp32 weight gradients / split-K partials)
    float* __restrict__ out = a.partial ? a.partial + (size_t)split * a.K * a.taps * a.C : a.dw;
    constexpr int RP = RB * 32;
    constexpr int LDC = BNc + 4;
    static_assert(RP * LDC * 4 <= 2 * STAGE, "wgrad epilogue staging does not fit the operand LDS");
    float* Cs = smem;
    for (int hh = 0; hh < WGM; ++hh) {
        __syncthreads();
        if (wm == hh) {
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                        Cs[row * LDC + wn * CB * 32 + j * 32 + (lane & 31)] = acc[i][j][q];
                    }
        }
        __syncthreads();
        constexpr int C4 = BNc / 4;
        for (int idx = tid; idx < RP * C4; idx += 256) {
            const int row = idx / C4, c4 = idx - row * C4;
            const int n = n0 + hh * RP + row, c = c0 + c4 * 4;
            if (n < a.K && c < a.C)
                *reinterpret_cast<f32x4*>(out + ((size_t)n * a.taps + tap) * a.C + c) =
                    *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
        }
    }
}

}  // namespace dpft
