// NHWC fp32 convolution family for gfx950 as MFMA implicit GEMM (v_mfma_f32_32x32x2_f32).
//
//   forward :  Y[M = B*OH*OW][N = K]  = A[M][kh*kw*C] * W^T        A gathered from x on the fly
//   dgrad   :  dX[M = B*H*W][N = C]   = A'[M][kh*kw*K] * Wt^T      A' gathered from dy (stride via
//                                                                   divisibility test), Wt = [C][taps][K]
//   wgrad   :  dW[K][tap][C]          = sum_pixels dY[p][k] * A[p][tap][c]
//
// Replaces the cuDNN kernels behind torchvision's ResNet/FPN in the reference
// (src/dprt/models/backbones/resnet.py:47-55,80-107; src/dprt/models/necks/fpn.py:39-43).
//
// Tiling (64-lane waves, 4 waves / workgroup): workgroup tile BM x BN, wave tile (RB*32) x (CB*32)
// of 32x32 MFMA blocks, BK = 32 per K-step staged through LDS with a register prefetch of the next
// step.  The A/B fragments are read from LDS with ds_read_b128: lanes 0-31 take k0..k0+3 and lanes
// 32-63 take k0+4..k0+7 of their row, and MFMA j consumes register j of both operands, i.e. a
// fixed permutation of the K order that both operands share (fp32 sums are order-independent up to
// rounding).  Rows are padded to 36 floats so the 16-lane groups of ds_read_b128 hit 16 distinct
// 16-byte slots.  Workgroup ids are remapped so that consecutive tiles (same A rows) share an XCD L2.
//
// Fused prologue (PRO): the producer's BatchNorm-apply + ReLU is applied to the A operand while it
// is staged (padding stays exactly zero).  Fused epilogue: per-tile per-channel (mean, M2) of the
// raw conv output for train-mode BatchNorm (combined later with Chan's formula: deterministic, no
// atomics, no E[x^2]-E[x]^2 cancellation).
#include "conv_core.h"

namespace dpft {


// ---------------------------------------------------------------------------------------------
// vector path: C % 32 == 0 (every bottleneck conv); float4 global loads, b128 LDS fragments
// ---------------------------------------------------------------------------------------------
// LIN = false: data gradient of a strided conv over ALL taps in one launch (small maps, see dpft_conv2d_nhwc_dgrad_f32):
// the source pixel is (oh + pad - r) / stride where that divides, not linear in the tap -- only that instantiation
// carries the divisions.
// BF16 = true (mixed-precision mode, dpft_conv_set_compute): the operands are rounded to bf16 (RNE) when a tile is written
// to LDS and multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulation); tensors in memory stay fp32.
// (The three-term split of fp32 operands that this kernel carried as an experiment through rounds 1-4 is conv_x3.hip now.)
// KS = 2: 512 threads per tile -- waves 4..7 mirror waves 0..3 on the ODD K-groups of every step and hand their
// accumulators over through LDS at the end.  Same tile, same loads, same MFMA count, twice the waves per SIMD: the mid / late
// layers launch < 2 workgroups per CU (456 tiles at layer 3) and run latency-bound -- two of these kernels side by side
// finish in 1.55-1.68x the time of one (tools/occupancy_probe.py).
template <int BM, int BN, int WGM, int WGN, bool DGRAD, bool PRO, bool LIN = true, bool BF16 = false, int KS = 1>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void igemm_vec_kernel(IgemmArgs a) {
    static_assert(KS == 1 || KS == 2, "K-split form: two wave sets");
    constexpr int NT = 64 * WGM * WGN * KS, ROWS = NT / 16;      // loader: 16 lanes x 16 bytes per row, ROWS rows per pass
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;
    constexpr int AP = BM / ROWS, BP = BN / ROWS;      // ROWS rows x 16 chunks (of 16 B) per loader pass
    static_assert(WGM * WGN == 4 && RB >= 1 && CB >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // (BM + BN) * LDK floats
    float* As = smem;
    float* Bs = smem + BM * LDK;
    // bf16 tiles: rows of 64 + 8 halfs (144 B: 16 consecutive rows start in 16 different 16-byte bank groups)
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);
    __bf16* Bh = Ah + BM * LDKH;

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const int kh = KS == 1 ? 0 : tid >> 8;            // which K-groups of a step this wave multiplies (4-wave tiles)
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    const int chunk = tid & 15, rowl = tid >> 4;
    // ---- operand addressing --------------------------------------------------------------------------------------
    // fp32 "MFMA" runs on the vector ALUs of gfx950 (its peak IS the vector fp32 peak; tools/probes/mfma_valu_overlap:
    // MFMA + VALU time add up, at any occupancy), so every address / mask / clamp instruction of the loader is paid in
    // matrix throughput.  The loader therefore costs no vector instruction per K-step:
    //  * raw buffer loads: the hardware range check returns 0 for an offset beyond the tensor (voffset + soffset are
    //    both checked on gfx950, dword by dword: tools/probes/buffer_range.hip), which serves the zero padding, rows
    //    >= M and weight rows >= N without clamps, selects or a zeroing pass;
    //  * one 32-bit byte offset per loader row, recomputed only when the filter tap changes (every C/64 steps), from
    //    per-row filter-row / filter-column validity bits built once; the channel offset of the step travels in the
    //    scalar soffset.
    const bool sub = DGRAD && a.sub_step > 1;
    constexpr bool lin = LIN;      // source pixel is linear in the tap: forward, stride-1 dgrad, parity-class dgrad
    const int roww = sub ? a.sub_ow : a.OW;
    const int ohw = sub ? a.sub_oh * a.sub_ow : a.OH * a.OW;
    const int ntap_s = sub ? a.sub_ns : a.kw;
    const int ntap_r = sub ? a.sub_nr : a.kh;      // kh, kw <= 8 (host side)
    constexpr unsigned OOB = 0x80000000u;
    int a_h0[AP], a_w0[AP], a_row[AP];      // tap-origin pixel (may lie outside the image) and its element offset
    unsigned a_mask[AP];                    // which filter rows / columns read a real pixel (see below)
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + rowl + ROWS * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / ohw;
        const int rem = mm - b * ohw;
        int oh = rem / roww, ow = rem - oh * roww;
        int h0, w0;
        if (!DGRAD) {
            h0 = oh * a.stride - a.pad;
            w0 = ow * a.stride - a.pad;
        } else if (sub) {      // (oh,ow) = sub-grid index; source row = index + q0 - tap index (see IgemmArgs::sub_*)
            h0 = oh + (a.sub_ph + a.pad - a.sub_r0) / a.sub_step;
            w0 = ow + (a.sub_pw + a.pad - a.sub_s0) / a.sub_step;
        } else {
            h0 = oh + a.pad;
            w0 = ow + a.pad;
        }
        a_h0[i] = h0; a_w0[i] = w0;
        a_row[i] = lin ? ((b * a.H + h0) * a.W + w0) * a.C : b * a.H * a.W * a.C;
        // tap validity is separable: bits 0..7 = filter rows that land inside the image, bits 8..15 = filter columns
        unsigned mask = 0;
        for (int ri = 0; ri < ntap_r; ++ri) {
            int hi;
            bool v = ok;
            if (!DGRAD) hi = h0 + ri;
            else if (lin) hi = h0 - ri;
            else {
                const int th = h0 - ri;
                hi = th / a.stride;
                v = v && th >= 0 && hi * a.stride == th;
            }
            mask |= (v && (unsigned)hi < (unsigned)a.H) ? (1u << ri) : 0u;
        }
        for (int si = 0; si < ntap_s; ++si) {
            int wi;
            bool v = ok;
            if (!DGRAD) wi = w0 + si;
            else if (lin) wi = w0 - si;
            else {
                const int tw = w0 - si;
                wi = tw / a.stride;
                v = v && tw >= 0 && wi * a.stride == tw;
            }
            mask |= (v && (unsigned)wi < (unsigned)a.W) ? (256u << si) : 0u;
        }
        a_mask[i] = mask;
    }
    unsigned b_off[BP];      // byte offset of weight row n (+ this lane's 16-byte chunk); OOB for rows >= N
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + rowl + ROWS * i;
        b_off[i] = n < a.N ? (unsigned)(n * a.Ktot + chunk * 4) * 4u : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * (a.x16 ? 2 : 4), 0x00020000);
    const bool x16 = a.x16 != 0;
    const bool raw_a = BF16 && !PRO && x16;      // bf16 tensor -> bf16 LDS tile without arithmetic: copy the bits
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.N * a.Ktot * 4, 0x00020000);
    const int cpt = a.C / BKV;  // K-steps per filter tap
    unsigned a_off[AP];         // byte offsets of the current tap (OOB where the tap misses the image)
    unsigned a_valid_tap = 0;   // bit i: row i of the current tap is a real pixel
    int cur_tap = -1;
    auto set_tap = [&](int tap) {
        const int ri = tap / ntap_s, si = tap - ri * ntap_s;
        const int tapoff = (DGRAD ? -1 : 1) * (ri * a.W + si) * a.C;      // linear case
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool v = ((a_mask[i] >> ri) & (a_mask[i] >> (8 + si)) & 1u) != 0;
            int e;
            if (lin) {
                e = a_row[i] + tapoff;
            } else {
                const int hi = (a_h0[i] - ri) / a.stride, wi = (a_w0[i] - si) / a.stride;
                e = a_row[i] + (hi * a.W + wi) * a.C;
            }
            a_off[i] = v ? (unsigned)(e + chunk * 4) * 4u : OOB;
            valid |= v ? (1u << i) : 0u;
        }
        a_valid_tap = valid;
        cur_tap = tap;
    };

    // Raw operand registers of the next DEPTH K-steps (register ring).  Loads are issued two steps ahead:
    // under load HBM/L2 latency exceeds one step of MFMA work, and a wait right behind the MFMAs stalls every
    // wave of the workgroup at once (probe: +30 % at 64x64).  The producer's BN+ReLU (PRO) is applied when a
    // register set is written to LDS, never earlier: consuming a load puts its latency in front of the compute.
    constexpr int DEPTH = (BM * BN >= 128 * 128) ? 1 : 2;       // 128x128 would exceed the VGPR budget at depth 2
    f32x4 ra[DEPTH][AP], rbv[DEPTH][BP], p_mu[DEPTH], p_sc[DEPTH], p_sh[DEPTH];
    unsigned a_valid[DEPTH];
    // data-gradient kernels: load_tile is called once per K-step in step order, so (tap, channel offset, weight offset)
    // are running scalars instead of kt / cpt and two more divisions per step (-8 % on the 3x3 256-channel dgrad; the
    // forward kernels measured 2-7 % SLOWER with the same change and keep the plain form)
    int run_tap = -1, run_c0 = 0, run_koff = 0;
    auto load_tile = [&](auto S, int kt) {
        constexpr int sidx = decltype(S)::value;
        int c0, koff;
        if constexpr (DGRAD) {
            if (run_tap < 0) {      // first call of this workgroup
                run_tap = kt / cpt;
                run_c0 = (kt - run_tap * cpt) * BKV;
            }
            if (run_tap != cur_tap) {
                set_tap(run_tap);
                const int ri = run_tap / ntap_s, si = run_tap - ri * ntap_s;
                run_koff = (sub ? (a.sub_r0 + a.sub_step * ri) * a.kw + a.sub_s0 + a.sub_step * si : run_tap) * a.C;
            }
            c0 = run_c0;
            koff = run_koff + c0;
            run_c0 += BKV;
            if (run_c0 == a.C) { run_c0 = 0; ++run_tap; }
        } else {
            const int tap = kt / cpt;
            c0 = (kt - tap * cpt) * BKV;
            if (tap != cur_tap) set_tap(tap);      // uniform
            const int r = tap / a.kw, s_ = tap - r * a.kw;
            koff = (r * a.kw + s_) * a.C + c0;      // == kt * BKV
        }
        if (PRO) {
            p_mu[sidx] = *reinterpret_cast<const f32x4*>(a.pro + c0 + chunk * 4);
            p_sc[sidx] = *reinterpret_cast<const f32x4*>(a.pro + a.C + c0 + chunk * 4);
            p_sh[sidx] = *reinterpret_cast<const f32x4*>(a.pro + 2 * a.C + c0 + chunk * 4);
            // bf16 operands: one fma per element, bn(v) = v * scale + (beta - mean * scale).  (The fp32 path subtracts the
            // mean first: that form keeps digits when |mean| >> sigma; with an 8-bit mantissa downstream it buys nothing.)
            if (BF16) p_sh[sidx] -= p_mu[sidx] * p_sc[sidx];
        }
        // one uniform branch around the whole batch of loads (a select per load would put every load in its own basic
        // block and serialise them).  bf16 tensors arrive as raw bits (lanes 0,1 of the quad) and are widened only when
        // the register set is consumed in store_tile: touching a load here would put its latency in front of the MFMAs.
        if (x16) {
#pragma unroll
            for (int i = 0; i < AP; ++i) ld4_raw16(ra[sidx][i], rsrc_a, a_off[i], c0 * 4);
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) ra[sidx][i] = ld4(rsrc_a, a_off[i], c0 * 4, false);
        }
        a_valid[sidx] = a_valid_tap;
#pragma unroll
        for (int i = 0; i < BP; ++i)
            rbv[sidx][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off[i], koff * 4, 0));
    };
    auto store_bf16 = [&](__bf16* dst, f32x4 v) {      // rounded to bf16 (RNE)
        *reinterpret_cast<bf16x4*>(dst) = __builtin_convertvector(v, bf16x4);
    };
    const bool pro_mask = a.kh * a.kw > 1 || a.pad > 0;      // padding exists: BN(0) != 0 must be forced back to 0
    auto store_tile = [&](auto S) {
        constexpr int sidx = decltype(S)::value;
        // (storage-format branches stay OUTSIDE the unrolled loops: a branch per register quad puts every store in its
        // own basic block with its own wait)
        if (BF16 && raw_a) {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                *reinterpret_cast<u32x2*>(&Ah[(rowl + ROWS * i) * LDKH + chunk * 4]) =
                    u32x2{__float_as_uint(ra[sidx][i][0]), __float_as_uint(ra[sidx][i][1])};
        } else {
            if (x16) {
#pragma unroll
                for (int i = 0; i < AP; ++i)
                    ra[sidx][i] = widen_bf16x4(u32x2{__float_as_uint(ra[sidx][i][0]), __float_as_uint(ra[sidx][i][1])});
            }
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                f32x4 val = ra[sidx][i];
                if (PRO) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = BF16 ? fmaf(val[e], p_sc[sidx][e], p_sh[sidx][e])
                                       : fmaf(val[e] - p_mu[sidx][e], p_sc[sidx][e], p_sh[sidx][e]);
                        val[e] = a.pro_relu ? fmaxf(t, 0.f) : t;
                    }
                    if (pro_mask && !((a_valid[sidx] >> i) & 1u)) val = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (BF16) store_bf16(&Ah[(rowl + ROWS * i) * LDKH + chunk * 4], val);
                else *reinterpret_cast<f32x4*>(&As[(rowl + ROWS * i) * LDK + chunk * 4]) = val;
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            if (BF16) store_bf16(&Bh[(rowl + ROWS * i) * LDKH + chunk * 4], rbv[sidx][i]);
            else *reinterpret_cast<f32x4*>(&Bs[(rowl + ROWS * i) * LDK + chunk * 4]) = rbv[sidx][i];
        }
    };

    // NACC > 1 = K-interleaved partial accumulators per 32x32 block (summed in the epilogue) to break the
    // single-accumulator MFMA chain of the small wave tiles (PMC at 64x64: MFMA busy 44 %, SQ_WAIT_INST_ANY
    // 62 %).  Measured: no gain at 64x64 (70 vs 69 TF) and -15 % at 128x64 (occupancy), so it stays off.
    constexpr int NACC = 1;
    f32x16 accp[NACC][RB][CB];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[q][i][j][r] = 0.f;

    const int kt_begin = split * a.ksteps_per_split;
    const int kt_end = min(a.ksteps, kt_begin + a.ksteps_per_split);
    const float* a_frag = As + (wm * RB * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    const float* b_frag = Bs + (wn * CB * 32 + (lane & 31)) * LDK + (lane >> 5) * 4;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, DEPTH - 1>;      // == S0 when DEPTH == 1

    // The fragments of K-group kg+1 are read from LDS BEFORE the MFMAs of group kg are issued (two register sets;
    // sched_barrier keeps the order -- left alone the scheduler reads a group right before its use, and the wave then
    // waits out the LDS latency with nothing in the pipe).
    const __bf16* a_fragh = Ah + (wm * RB * 32 + (lane & 31)) * LDKH + (lane >> 5) * 8;
    const __bf16* b_fragh = Bh + (wn * CB * 32 + (lane & 31)) * LDKH + (lane >> 5) * 8;
    auto compute_bf16 = [&]() {      // 4 K-groups of 16: lane l holds k = 8 * (l / 32) .. + 7 of row / column l % 32
        bf16x8 af[2][RB], bf[2][CB];
        auto frags = [&](int set, int kg) {
#pragma unroll
            for (int i = 0; i < RB; ++i) af[set][i] = *reinterpret_cast<const bf16x8*>(a_fragh + i * 32 * LDKH + kg * 16);
#pragma unroll
            for (int j = 0; j < CB; ++j) bf[set][j] = *reinterpret_cast<const bf16x8*>(b_fragh + j * 32 * LDKH + kg * 16);
        };
        constexpr int NG = BKV / 16 / KS;      // K-groups of this wave: kg = KS * g + kh
        frags(0, kh);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) frags((g + 1) & 1, KS * (g + 1) + kh);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    accp[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][i], bf[g & 1][j], accp[0][i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto compute_f32 = [&]() {
        f32x4 af[2][RB], bf[2][CB];
        auto frags = [&](int set, int kg) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                af[set][i] = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDK + kg * 8);
#pragma unroll
            for (int j = 0; j < CB; ++j)
                bf[set][j] = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDK + kg * 8);
        };
        constexpr int NG = BKV / 8 / KS;      // K-groups of this wave: kg = KS * g + kh
        frags(0, kh);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) frags((g + 1) & 1, KS * (g + 1) + kh);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j)
                        accp[e % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][e], bf[g & 1][j][e],
                                                                                    accp[e % NACC][i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto compute = [&]() {
        if constexpr (BF16) compute_bf16();
        else compute_f32();
    };

    // tile t (relative to kt_begin) lives in register set t % DEPTH
    const bool ab_ld = !(a.ablate & 1), ab_st = !(a.ablate & 2), ab_mm = !(a.ablate & 8);
    if (kt_begin < kt_end) load_tile(S0{}, kt_begin);
    if (DEPTH == 2 && kt_begin + 1 < kt_end) load_tile(S1{}, kt_begin + 1);
    if (kt_begin < kt_end) {
        store_tile(S0{});
        if (kt_begin + DEPTH < kt_end) load_tile(S0{}, kt_begin + DEPTH);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; kt += DEPTH) {
        // ---- step kt: the next tile (kt+1) is in set 1 % DEPTH ----
        if (ab_mm) compute();
        __syncthreads();
        if (kt + 1 < kt_end) {
            if (ab_st) store_tile(S1{});
            if (kt + 1 + DEPTH < kt_end && ab_ld) load_tile(S1{}, kt + 1 + DEPTH);
        }
        __syncthreads();
        if (DEPTH == 2) {
            if (kt + 1 >= kt_end) break;
            // ---- step kt+1: the next tile (kt+2) is in set 0 ----
            if (ab_mm) compute();
            __syncthreads();
            if (kt + 2 < kt_end) {
                if (ab_st) store_tile(S0{});
                if (kt + 2 + DEPTH < kt_end && ab_ld) load_tile(S0{}, kt + 2 + DEPTH);
            }
            __syncthreads();
        }
    }
    f32x16 (&acc)[RB][CB] = accp[0];
    if (NACC == 2) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[i][j] += accp[NACC - 1][i][j];
    }
    if constexpr (KS == 2) {      // waves 4..7 hand their partial sums to waves 0..3: [wave][block][register][lane] in LDS
        static_assert((size_t)4 * RB * CB * 16 * 64 <= (size_t)(BM + BN) * LDK, "hand-over buffer exceeds the operand tiles");
        __syncthreads();          // the operand tiles are dead
        float* hand = smem + ((size_t)wave * RB * CB * 16) * 64 + lane;
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hand[((i * CB + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += hand[((i * CB + j) * 16 + r) * 64];
        }
    }
    if (a.ablate & 4) {      // keep the accumulators live without the epilogue
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 1.2345e-30f) a.y[0] = t;
        return;
    }
    igemm_epilogue<BM, BN, WGM, WGN, RB, CB, NT>(a, acc, m0, n0, mt, split, smem);
}

// ---------------------------------------------------------------------------------------------
// generic path: any C (stem C=3, FPN C=3/6/16); K flattened as (tap, c); scalar gathers
// workgroup tile 128 x BN (BN = 32 or 64), waves 4 x 1
// ---------------------------------------------------------------------------------------------
template <int BN, bool DGRAD>
__global__ __launch_bounds__(256) void igemm_gen_kernel(IgemmArgs a) {
    constexpr int BM = 128, WGM = 4, WGN = 1, RB = 1, CB = BN / 32;
    constexpr int KPB = BN / 8;  // k elements per thread for the B tile
    constexpr int SMEM = ((BM + BN) * LDG > BM * (BN + 4)) ? (BM + BN) * LDG : BM * (BN + 4);
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* As = smem;
    float* Bs = smem + BM * LDG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    // A: one row per thread, 16 consecutive k
    const int arow = tid & 127, akh = tid >> 7;
    int bh, bw;
    const float* abase;
    {
        const int m = m0 + arow;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int ohw = a.OH * a.OW;
        const int b = mm / ohw;
        const int rem = mm - b * ohw;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        if (!DGRAD) {
            bh = ok ? oh * a.stride - a.pad : -(1 << 28);
            bw = ow * a.stride - a.pad;
        } else {
            bh = ok ? oh + a.pad : -(1 << 28);
            bw = ow + a.pad;
        }
        abase = a.x + (size_t)b * a.H * a.W * a.C;
    }
    const int brow = tid % BN, bkq = tid / BN;
    const bool bok = (n0 + brow) < a.N;
    const float* bbase = a.w + (size_t)(bok ? n0 + brow : 0) * a.Ktot;

    float ra[16], rbv[KPB];
    auto load_tile = [&](int kt) {
        int k = kt * BK + akh * 16;
        int tap = k / a.C;
        int c = k - tap * a.C;
        int r = tap / a.kw, s = tap - r * a.kw;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int hi, wi;
            bool v = (k + e) < a.Ktot;
            if (!DGRAD) {
                hi = bh + r;
                wi = bw + s;
                v = v && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            } else {
                const int th = bh - r, tw = bw - s;
                v = v && th >= 0 && tw >= 0;
                hi = th / a.stride;
                wi = tw / a.stride;
                v = v && (hi * a.stride == th) && (wi * a.stride == tw) && hi < a.H && wi < a.W;
            }
            ra[e] = v ? abase[((size_t)hi * a.W + wi) * a.C + c] : 0.f;
            if (++c == a.C) {
                c = 0;
                if (++s == a.kw) {
                    s = 0;
                    ++r;
                }
            }
        }
        const int kb = kt * BK + bkq * KPB;
#pragma unroll
        for (int e = 0; e < KPB; ++e) rbv[e] = (bok && (kb + e) < a.Ktot) ? bbase[kb + e] : 0.f;
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 16; ++e) As[arow * LDG + akh * 16 + e] = ra[e];
#pragma unroll
        for (int e = 0; e < KPB; ++e) Bs[brow * LDG + bkq * KPB + e] = rbv[e];
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    const int kt_begin = split * a.ksteps_per_split;
    const int kt_end = min(a.ksteps, kt_begin + a.ksteps_per_split);
    const float* a_frag = As + (wm * 32 + (lane & 31)) * LDG + (lane >> 5);
    const float* b_frag = Bs + (lane & 31) * LDG + (lane >> 5);
    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile();
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float av = a_frag[kk * 2];
#pragma unroll
            for (int j = 0; j < CB; ++j)
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_frag[j * 32 * LDG + kk * 2],
                                                                 acc[0][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
    igemm_epilogue<BM, BN, WGM, WGN, RB, CB>(a, acc, m0, n0, mt, split, smem);
}

// split-K reduce: y = sum_s partial[s] (+bias)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                     float* __restrict__ y, int64_t MN, int N, int splits, int accumulate) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= MN) return;
    if (i + 3 < MN && (N % 4) == 0) {
        f32x4 s = *reinterpret_cast<const f32x4*>(partial + i);
        for (int k = 1; k < splits; ++k) {
            f32x4 t = *reinterpret_cast<const f32x4*>(partial + (size_t)k * MN + i);
            s += t;
        }
        if (bias) {
            const int c = (int)(i % N);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += bias[c + e];
        }
        if (accumulate) s += *reinterpret_cast<const f32x4*>(y + i);
        *reinterpret_cast<f32x4*>(y + i) = s;
    } else {
        for (int e = 0; e < 4 && i + e < MN; ++e) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += partial[(size_t)k * MN + i + e];
            if (bias) s += bias[(i + e) % N];
            if (accumulate) s += y[i + e];
            y[i + e] = s;
        }
    }
}

// split-K reduce of a forward conv WITH the BatchNorm tile statistics of the reduced output (what the non-split kernel's
// epilogue produces): one launch instead of reduce + bn_stats on the radar encoders' small maps, where every dependent
// launch costs more than the work in it.  block = one statistics tile (tile_rows <= 128 rows) x 64 columns;
// thread = one float4 column chunk x every 16th row.  N % 64 == 0.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ partial, float* __restrict__ y,
                                                                  float* __restrict__ stats, int64_t M, int N, int splits,
                                                                  int tile_rows) {
    __shared__ f32x4 red[16][16];
    __shared__ f32x4 smean[16];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int tile = blockIdx.x, c = blockIdx.y * 64 + cl * 4;
    const int64_t r0 = (int64_t)tile * tile_rows, MN = M * N;
    const int cnt = (int)min((int64_t)tile_rows, M - r0);
    f32x4 v[8];
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = rg + i * 16;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < cnt) {
            const int64_t o = (r0 + r) * N + c;
            f32x4 a = *reinterpret_cast<const f32x4*>(partial + o);
            for (int k = 1; k < splits; ++k) a += *reinterpret_cast<const f32x4*>(partial + (size_t)k * MN + o);
            *reinterpret_cast<f32x4*>(y + o) = a;
            v[i] = a;
            sum += a;
        }
    }
    red[rg][cl] = sum;
    __syncthreads();
    if (rg == 0) {
        f32x4 t = red[0][cl];
        for (int i = 1; i < 16; ++i) t += red[i][cl];
        smean[cl] = t * (1.0f / (float)cnt);
    }
    __syncthreads();
    const f32x4 mean = smean[cl];
    f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (rg + i * 16 < cnt) {
            const f32x4 dlt = v[i] - mean;
            m2 += dlt * dlt;
        }
    __syncthreads();
    red[rg][cl] = m2;
    __syncthreads();
    if (rg == 0) {
        f32x4 t = red[0][cl];
        for (int i = 1; i < 16; ++i) t += red[i][cl];
        *reinterpret_cast<f32x4*>(stats + ((size_t)tile * 2 + 0) * N + c) = mean;
        *reinterpret_cast<f32x4*>(stats + ((size_t)tile * 2 + 1) * N + c) = t;
    }
}

// weight-gradient slabs: many splits (up to 512) of a result that can be as small as 64 x 64 -- the slab index is
// spread over PARTS lanes per output chunk (the serial walk above needs `splits` dependent trips per thread)
template <int PARTS>
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const float* __restrict__ partial, float* __restrict__ y,
                                                                 int64_t MN, int splits) {
    constexpr int CH = 256 / PARTS;      // float4 chunks per block
    __shared__ f32x4 red[256];
    const int ch = threadIdx.x % CH, part = threadIdx.x / CH;
    const int64_t i = ((int64_t)blockIdx.x * CH + ch) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < MN)
        for (int k = part; k < splits; k += PARTS) s += *reinterpret_cast<const f32x4*>(partial + (size_t)k * MN + i);
    red[threadIdx.x] = s;
    __syncthreads();
    if (part == 0 && i < MN) {
#pragma unroll
        for (int q = 1; q < PARTS; ++q) s += red[q * CH + ch];
        *reinterpret_cast<f32x4*>(y + i) = s;
    }
}

// split-K reduce with the masked residual: y = sum_s partial[s] + (mask > 0 ? src : 0)
__global__ void splitk_reduce_residual_kernel(const float* __restrict__ partial, const float* __restrict__ src,
                                              const float* __restrict__ mask, float* __restrict__ y, int64_t MN,
                                              int splits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += partial[(size_t)k * MN + i];
    y[i] = s + (mask[i] > 0.f ? src[i] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[n][tap][c] = sum_p dY[p][n] * act(x)[p@tap][c]
// LDS holds [32 pixels][BMn] of dY and [32 pixels][BNc] of x; both fragments are ds_read_b32
// (consecutive lanes -> consecutive channels, conflict free).
// ---------------------------------------------------------------------------------------------

// BF16 = true (mixed-precision mode, square tiles): both operands are rounded to bf16 and TRANSPOSED on their
// way into LDS ([channel][pixel], so that a lane finds the 8 consecutive reduction indices v_mfma_f32_32x32x16_bf16
// wants); see store_tile.
template <int BMn, int BNc, int WGM, int WGN, bool PRO, bool BF16 = false>
__global__ __launch_bounds__(256) void wgrad_vec_kernel(WgradArgs a) {
    static_assert(!BF16 || (BMn == BNc && (BMn == 128 || BMn == 64)), "bf16 weight gradient: 128 x 128 and 64 x 64 tiles");
    constexpr int RB = BMn / WGM / 32, CB = BNc / WGN / 32;
    constexpr int YCH = BMn / 4, XCH = BNc / 4;        // float4 chunks per pixel row
    constexpr int YRP = 256 / YCH, XRP = 256 / XCH;    // pixel rows per pass
    constexpr int YP = BKP / YRP, XP = BKP / XRP;
    static_assert(WGM * WGN == 4 && YP >= 1 && XP >= 1, "bad wgrad tile");
    constexpr int SMEM_FLOATS = BKP * (BMn + BNc);
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* Ys = smem;
    float* Xs = smem + BKP * BMn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    const int ych = tid % YCH, yrow = tid / YCH;
    const int xch = tid % XCH, xrow = tid / XCH;
    const bool y_ok = (n0 + ych * 4) < a.K;
    const bool x_ok = (c0 + xch * 4) < a.C;
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (PRO && x_ok) {
        mu = *reinterpret_cast<const f32x4*>(a.pro + c0 + xch * 4);
        sc = *reinterpret_cast<const f32x4*>(a.pro + a.C + c0 + xch * 4);
        sh = *reinterpret_cast<const f32x4*>(a.pro + 2 * a.C + c0 + xch * 4);
        if (BF16) sh -= mu * sc;      // one fma per element (see igemm_vec_kernel)
    }
    const int ohw = a.OH * a.OW;

    f32x4 ry[YP], rx[XP];
    unsigned x_valid = 0;
    // Loader without divisions (see igemm_vec_kernel: every vector instruction here is paid in matrix throughput): raw
    // buffer loads, dy rows at fixed per-lane offsets + a scalar pixel offset, and for x a (b, oh, ow) triple per loader
    // row that is decomposed once and then ADVANCED by the 64 pixels of a step with two carries.
    constexpr unsigned OOB = 0x80000000u;
    const bool x16 = a.x16 != 0, dy16 = a.dy16 != 0;
    const __amdgpu_buffer_rsrc_t rsrc_y =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.M * a.K * (dy16 ? 2 : 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * (x16 ? 2 : 4), 0x00020000);
    // pixel (within the step) of loader pass i: fp32 -- row + i * rows-per-pass; bf16 -- passes 2k, 2k+1 are ADJACENT pixels
    // (they share one 32-bit LDS word of the transposed tile)
    auto ypix = [&](int i) { return BF16 ? 2 * YRP * (i >> 1) + 2 * yrow + (i & 1) : yrow + i * YRP; };
    auto xpix = [&](int i) { return BF16 ? 2 * XRP * (i >> 1) + 2 * xrow + (i & 1) : xrow + i * XRP; };
    unsigned y_voff[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) y_voff[i] = y_ok ? (unsigned)(ypix(i) * a.K + n0 + ych * 4) * 4u : OOB;
    int xb[XP], xoh[XP], xow[XP];      // pixel of loader row i in the NEXT step to be loaded
    {
        const int p_first = (split * a.psteps_per_split) * BKP;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int p = p_first + xpix(i);
            xb[i] = p / ohw;
            const int rem = p - xb[i] * ohw;
            xoh[i] = rem / a.OW;
            xow[i] = rem - xoh[i] * a.OW;
        }
    }
    const int adv_b = BKP / ohw, adv_rem = BKP - adv_b * ohw;      // one step = adv_b images + adv_oh rows + adv_ow pixels
    const int adv_oh = adv_rem / a.OW, adv_ow = adv_rem - adv_oh * a.OW;
    const int x_lane = (c0 + xch * 4);
    auto load_tile = [&](int ps) {      // called once per step, in step order
        const int soff_y = ps * BKP * a.K * 4;
        if (dy16) {      // bf16 tensors: raw bits now, widened when the registers are consumed (see igemm_vec_kernel)
#pragma unroll
            for (int i = 0; i < YP; ++i) ld4_raw16(ry[i], rsrc_y, y_voff[i], soff_y);
        } else {
#pragma unroll
            for (int i = 0; i < YP; ++i) ry[i] = ld4(rsrc_y, y_voff[i], soff_y, false);
        }
        unsigned valid = 0;
        unsigned xoffs[XP];
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int hi = xoh[i] * a.stride - a.pad + r, wi = xow[i] * a.stride - a.pad + s;
            const bool v = x_ok && xb[i] < a.B && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const unsigned e = ((((unsigned)xb[i] * a.H + hi) * a.W + wi) * a.C + x_lane) * 4u;      // garbage where !v
            xoffs[i] = v ? e : OOB;
            valid |= v ? (1u << i) : 0u;
            // advance to the next step's pixel
            int ow = xow[i] + adv_ow, oh = xoh[i] + adv_oh, b = xb[i] + adv_b;
            if (ow >= a.OW) { ow -= a.OW; ++oh; }
            if (oh >= a.OH) { oh -= a.OH; ++b; }
            xow[i] = ow; xoh[i] = oh; xb[i] = b;
        }
        if (x16) {
#pragma unroll
            for (int i = 0; i < XP; ++i) ld4_raw16(rx[i], rsrc_x, xoffs[i], 0);
        } else {
#pragma unroll
            for (int i = 0; i < XP; ++i) rx[i] = ld4(rsrc_x, xoffs[i], 0, false);
        }
        x_valid = valid;
    };
    // bf16 tiles: [channel][pixel], rows of WLD = 76 halfs (152 B = 38 words).  A lane owns 4 channels x 2 adjacent pixels
    // per pass pair = four 32-bit words in four rows; in the e-th of its four writes it takes channel (e + ych / 8) % 4, and
    // with the 38-word pitch the 64 lanes of a wave then hit 64 different banks (brute-forced; any plain order is 8-way
    // conflicted because the lanes of a wave differ in the CHANNEL chunk, i.e. by whole rows).
    constexpr int WLD = BMn == 128 ? 76 : 72;      // 64 x 64 tile (16 channel chunks per pixel row): 36 words, order (e + ych / 4) % 4
    constexpr int RSH = BMn == 128 ? 3 : 2;
    __bf16* Yh = reinterpret_cast<__bf16*>(smem);
    __bf16* Xh = Yh + BMn * WLD;
    auto rot4 = [](f32x4 v, int r) {      // v[(e + r) & 3] at position e, r per lane
        f32x4 t = (r & 1) ? f32x4{v[1], v[2], v[3], v[0]} : v;
        return (r & 2) ? f32x4{t[2], t[3], t[0], t[1]} : t;
    };
    auto store_pair = [&](__bf16* T, int ch, int rowp, f32x4 v0, f32x4 v1) {      // pixels 2 * rowp', 2 * rowp' + 1
        const int r = (ch >> RSH) & 3;
        const f32x4 a0 = rot4(v0, r), a1 = rot4(v1, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            f32x2_ v = {a0[e], a1[e]};
            __bf16* dst = &T[(ch * 4 + ((e + r) & 3)) * WLD + rowp];
            *reinterpret_cast<bf16x2*>(dst) = __builtin_convertvector(v, bf16x2);
        }
    };
    auto pro4 = [&](f32x4 v, int i) {
        if (PRO) {      // producer BN+ReLU applied after the MFMAs of the current step (see igemm_vec_kernel)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = BF16 ? fmaf(v[e], sc[e], sh[e]) : fmaf(v[e] - mu[e], sc[e], sh[e]);
                v[e] = a.pro_relu ? fmaxf(t, 0.f) : t;
            }
            if (!((x_valid >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};      // padding / pixel tail: BN(0) != 0
        }
        return v;
    };
    auto unraw = [](f32x4 v) { return widen_bf16x4(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}); };
    auto store_tile = [&]() {
        if (dy16) {
#pragma unroll
            for (int i = 0; i < YP; ++i) ry[i] = unraw(ry[i]);
        }
        if (x16) {
#pragma unroll
            for (int i = 0; i < XP; ++i) rx[i] = unraw(rx[i]);
        }
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < YP; i += 2) store_pair(Yh, ych, ypix(i), ry[i], ry[i + 1]);
#pragma unroll
            for (int i = 0; i < XP; i += 2) store_pair(Xh, xch, xpix(i), pro4(rx[i], i), pro4(rx[i + 1], i + 1));
        } else {
#pragma unroll
            for (int i = 0; i < YP; ++i)
                *reinterpret_cast<f32x4*>(&Ys[(yrow + i * YRP) * BMn + ych * 4]) = ry[i];
#pragma unroll
            for (int i = 0; i < XP; ++i)
                *reinterpret_cast<f32x4*>(&Xs[(xrow + i * XRP) * BNc + xch * 4]) = pro4(rx[i], i);
        }
    };

    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int ps_begin = split * a.psteps_per_split;
    const int ps_end = min(a.psteps, ps_begin + a.psteps_per_split);
    const float* y_frag = Ys + (lane >> 5) * BMn + wm * RB * 32 + (lane & 31);
    const float* x_frag = Xs + (lane >> 5) * BNc + wn * CB * 32 + (lane & 31);
    if (ps_begin < ps_end) {
        load_tile(ps_begin);
        store_tile();
    }
    __syncthreads();
    for (int ps = ps_begin; ps < ps_end; ++ps) {
        const bool more = ps + 1 < ps_end;
        if (more) load_tile(ps + 1);
        if constexpr (BF16) {      // 4 groups of 16 pixels; lane l: channel l % 32, pixels 8 * (l / 32) .. + 7 of the group
            const __bf16* yf = Yh + (wm * RB * 32 + (lane & 31)) * WLD + (lane >> 5) * 8;
            const __bf16* xf = Xh + (wn * CB * 32 + (lane & 31)) * WLD + (lane >> 5) * 8;
            auto frag8 = [](const __bf16* q) {      // rows are 8-byte, not 16-byte aligned (152-byte pitch): two b64 reads
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(q), hi = *reinterpret_cast<const bf16x4*>(q + 4);
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            bf16x8 ay[2][RB], bx[2][CB];
            auto frags = [&](int set, int kg) {
#pragma unroll
                for (int i = 0; i < RB; ++i) ay[set][i] = frag8(yf + i * 32 * WLD + kg * 16);
#pragma unroll
                for (int j = 0; j < CB; ++j) bx[set][j] = frag8(xf + j * 32 * WLD + kg * 16);
            };
            frags(0, 0);
#pragma unroll
            for (int kg = 0; kg < BKP / 16; ++kg) {
                if (kg + 1 < BKP / 16) frags((kg + 1) & 1, kg + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[kg & 1][i], bx[kg & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {   // fragments of pixel pair kk+PF are read before the MFMAs of pair kk are issued (see igemm_vec_kernel)
            constexpr int PF = 2, NS = PF + 1;
            float av[NS][RB], bv[NS][CB];
            auto frags = [&](int set, int kk) {
#pragma unroll
                for (int i = 0; i < RB; ++i) av[set][i] = y_frag[kk * 2 * BMn + i * 32];
#pragma unroll
                for (int j = 0; j < CB; ++j) bv[set][j] = x_frag[kk * 2 * BNc + j * 32];
            };
#pragma unroll
            for (int q = 0; q < PF; ++q) frags(q, q);
#pragma unroll
            for (int kk = 0; kk < BKP / 2; ++kk) {
                if (kk + PF < BKP / 2) frags((kk + PF) % NS, kk + PF);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk % NS][i], bv[kk % NS][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
    // epilogue: stage one wave-row of the tile at a time through LDS -> full 16-byte stores along c
    float* __restrict__ out = a.partial ? a.partial + (size_t)split * a.K * a.taps * a.C : a.dw;
    constexpr int RP = RB * 32;          // rows (n) per pass
    constexpr int LDC = BNc + 4;
    static_assert(RP * LDC <= BKP * (BMn + BNc), "wgrad epilogue staging does not fit the operand LDS");
    float* Cs = smem;
    for (int h = 0; h < WGM; ++h) {
        __syncthreads();
        if (wm == h) {
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
            const int n = n0 + h * RP + row, c = c0 + c4 * 4;
            if (n < a.K && c < a.C)
                *reinterpret_cast<f32x4*>(out + ((size_t)n * a.taps + tap) * a.C + c) =
                    *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
        }
    }
}

// generic wgrad: flattened j = tap*C + c (any C); tile 64 (n) x 64 (j); waves 2 x 2
__global__ __launch_bounds__(256) void wgrad_gen_kernel(WgradArgs a) {
    constexpr int BMn = 64, BNj = 64;
    __shared__ __attribute__((aligned(16))) float smem[BK * (BMn + BNj)];
    float* Ys = smem;
    float* Xs = smem + BK * BMn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int jtiles = a.ctiles;
    const int nwg = a.ktiles * jtiles * a.splits;
    int bid = xcd_remap(blockIdx.x, nwg);
    const int split = bid / (a.ktiles * jtiles);
    bid -= split * (a.ktiles * jtiles);
    const int kt_ = bid / jtiles, jt_ = bid - kt_ * jtiles;
    const int n0 = kt_ * BMn, j0 = jt_ * BNj;

    // thread -> column (n or j) and a group of 8 consecutive pixels
    const int col = tid & 63, pg = tid >> 6;
    const bool y_ok = (n0 + col) < a.K;
    const int j = j0 + col;
    const bool x_ok = j < a.J;
    const int tap = x_ok ? j / a.C : 0;
    const int c = x_ok ? j - tap * a.C : 0;
    const int r = tap / a.kw, s = tap - r * a.kw;
    const int ohw = a.OH * a.OW;

    float ry[8], rx[8];
    auto load_tile = [&](int ps) {
        const int p0 = ps * BK + pg * 8;
        int b = p0 / ohw;
        int rem = p0 - b * ohw;
        int oh = rem / a.OW, ow = rem - oh * a.OW;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int p = p0 + e;
            const bool pv = p < a.M;
            ry[e] = (y_ok && pv) ? a.dy[(size_t)p * a.K + n0 + col] : 0.f;
            const int hi = oh * a.stride - a.pad + r, wi = ow * a.stride - a.pad + s;
            const bool v = x_ok && pv && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            rx[e] = v ? a.x[(((size_t)b * a.H + hi) * a.W + wi) * a.C + c] : 0.f;
            if (++ow == a.OW) {
                ow = 0;
                if (++oh == a.OH) {
                    oh = 0;
                    ++b;
                }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Ys[(pg * 8 + e) * BMn + col] = ry[e];
            Xs[(pg * 8 + e) * BNj + col] = rx[e];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int ps_begin = split * a.psteps_per_split;
    const int ps_end = min(a.psteps, ps_begin + a.psteps_per_split);
    const float* y_frag = Ys + (lane >> 5) * BMn + wm * 32 + (lane & 31);
    const float* x_frag = Xs + (lane >> 5) * BNj + wn * 32 + (lane & 31);
    if (ps_begin < ps_end) {
        load_tile(ps_begin);
        store_tile();
    }
    __syncthreads();
    for (int ps = ps_begin; ps < ps_end; ++ps) {
        const bool more = ps + 1 < ps_end;
        if (more) load_tile(ps + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y_frag[kk * 2 * BMn], x_frag[kk * 2 * BNj], acc, 0, 0, 0);
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
    float* __restrict__ out = a.partial ? a.partial + (size_t)split * a.K * a.J : a.dw;
    const int jj = j0 + wn * 32 + (lane & 31);
    if (jj < a.J) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n0 + wm * 32 + 4 * (lane >> 5) + (q & 3) + 8 * (q >> 2);
            if (n < a.K) out[(size_t)n * a.J + jj] = acc[q];
        }
    }
}

// Data gradient of a conv whose INPUT has <= 4 channels (the 7x7/2 stems behind the radar encoders' 6->3 adjust conv):
// as an implicit GEMM this has N = 3 (padded to a 32-wide tile) and, at stride 2, three of four taps invalid per pixel
// -- 2.5 TF.  Here one thread owns one input pixel, walks only the taps that reach it and dots dy's K channels
// (float4 loads) with the transposed weights held in LDS.  w_t is [C][taps][K]; K % 4 == 0; taps * C * K floats <= 40 KB.
template <int CIN>
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w_t,
                                                         float* __restrict__ dx, int B, int H, int W, int OH, int OW,
                                                         int K, int kh, int kw, int stride, int pad, int accumulate) {
    extern __shared__ float wl[];      // [CIN][taps][K]
    const int taps = kh * kw, nw = CIN * taps * K;
    for (int i = threadIdx.x * 4; i < nw; i += 256 * 4) *reinterpret_cast<f32x4*>(wl + i) = *reinterpret_cast<const f32x4*>(w_t + i);
    __syncthreads();
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (int64_t)B * H * W) return;
    const int iw = (int)(pix % W), ih = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
    for (int r = (ih + pad) % stride; r < kh; r += stride) {
        const int oh = (ih + pad - r) / stride;
        if (ih + pad - r < 0 || oh >= OH) continue;
        for (int q = (iw + pad) % stride; q < kw; q += stride) {
            const int ow = (iw + pad - q) / stride;
            if (iw + pad - q < 0 || ow >= OW) continue;
            const float* g = dy + (((int64_t)b * OH + oh) * OW + ow) * K;
            const float* wt = wl + (r * kw + q) * K;
            for (int k = 0; k < K; k += 4) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + k);
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + c * taps * K + k);
                    acc[c] += gv[0] * wv[0] + gv[1] * wv[1] + gv[2] * wv[2] + gv[3] * wv[3];
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        float* o = dx + pix * CIN + c;
        *o = accumulate ? *o + acc[c] : acc[c];
    }
}

// [K][taps][C] -> [C][taps][K]
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K,
                                        int taps, int C) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, c = c0 + tx;
        tile[i][tx] = (k < K && c < C) ? w[((size_t)k * taps + tap) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, k = k0 + tx;
        if (c < C && k < K) wt[((size_t)c * taps + tap) * K + k] = tile[tx][i];
    }
}

// 64 x 64 tiles (all K, C of the batch multiples of 64): float4 reads along c, float4 writes along k (bf16: 8-byte writes),
// the transposition through a 65-float-pitch LDS tile -- 256-byte row segments on both sides instead of 128-byte ones and a
// quarter of the blocks (0.45 -> 0.25 ms per step for the 44.5 M camera weights)
__global__ __launch_bounds__(256) void weight_transpose_batch64_kernel(TransposeBatch tb) {
    __shared__ float tile[64 * 65];
    int i = 0;
    while (i + 1 < tb.n && (int)blockIdx.x >= tb.blk_start[i + 1]) ++i;      // wave-uniform scan (n <= 80)
    const int K = tb.K[i], taps = tb.taps[i], C = tb.C[i];
    const float* __restrict__ w = tb.w[i];
    float* __restrict__ wt = tb.wt[i];
    int rel = blockIdx.x - tb.blk_start[i];
    const int cb = C / 64, kb = K / 64;
    const int tap = rel / (cb * kb);
    rel -= tap * cb * kb;
    const int k0 = (rel / cb) * 64, c0 = (rel % cb) * 64;
    const int q = threadIdx.x & 15, r = threadIdx.x >> 4;      // 16 quads per row x 16 rows per pass
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int k = r + p * 16;
        const size_t idx = ((size_t)(k0 + k) * taps + tap) * C + c0 + q * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + idx);
        if (tb.mode == 2) {
            *reinterpret_cast<bf16x4s*>(reinterpret_cast<__bf16*>(wt) + idx) = __builtin_convertvector(v, bf16x4s);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[k * 65 + q * 4 + e] = v[e];
        }
    }
    if (tb.mode == 2) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = r + p * 16;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[(q * 4 + e) * 65 + c];
        const size_t o = ((size_t)(c0 + c) * taps + tap) * K + k0 + q * 4;
        if (tb.mode == 1) *reinterpret_cast<bf16x4s*>(reinterpret_cast<__bf16*>(wt) + o) = __builtin_convertvector(v, bf16x4s);
        else *reinterpret_cast<f32x4*>(wt + o) = v;
    }
}

// the same for up to 80 weight tensors in ONE launch (a ResNet stage's convs: 236 launches of ~5 us per step otherwise)
__global__ void weight_transpose_batch_kernel(TransposeBatch tb) {
    __shared__ float tile[32][33];
    int i = 0;
    while (i + 1 < tb.n && (int)blockIdx.x >= tb.blk_start[i + 1]) ++i;      // wave-uniform scan (n <= 80)
    const int K = tb.K[i], taps = tb.taps[i], C = tb.C[i];
    const float* __restrict__ w = tb.w[i];
    float* __restrict__ wt = tb.wt[i];
    int rel = blockIdx.x - tb.blk_start[i];
    const int cb = (C + 31) / 32, kb = (K + 31) / 32;
    const int tap = rel / (cb * kb);
    rel -= tap * cb * kb;
    const int k0 = (rel / cb) * 32, c0 = (rel % cb) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + tx;
        const size_t idx = ((size_t)k * taps + tap) * C + c;
        const float v = (k < K && c < C) ? w[idx] : 0.f;
        tile[j][tx] = v;
        if (tb.mode == 2 && k < K && c < C) reinterpret_cast<__bf16*>(wt)[idx] = (__bf16)v;      // bf16 shadow, same order
    }
    if (tb.mode == 2) return;
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, k = k0 + tx;
        if (c < C && k < K) {
            const size_t o = ((size_t)c * taps + tap) * K + k;
            if (tb.mode == 1) reinterpret_cast<__bf16*>(wt)[o] = (__bf16)tile[tx][j];
            else wt[o] = tile[tx][j];
        }
    }
}

// db[k] += sum_m dy[m][k]; db zeroed by the launcher; K <= 256
__global__ void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int64_t M, int K) {
    __shared__ float red[256];
    const int rpi = 256 / K;            // rows per iteration
    const int c = threadIdx.x % K, r0 = threadIdx.x / K;
    float s = 0.f;
    if (r0 < rpi)
        for (int64_t m = (int64_t)blockIdx.x * rpi + r0; m < M; m += (int64_t)gridDim.x * rpi)
            s += dy[m * K + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int i = 0; i < rpi; ++i) t += red[threadIdx.x + i * K];
        atomicAdd(&db[threadIdx.x], t);
    }
}

// The same column sums WITHOUT a cleared output and without atomics (round 4): every block leaves its K partial sums in a
// slab of the conv workspace (agent-scope stores), takes a ticket, and the block that draws the last one adds the partials
// in block order and writes db -- one launch, a fixed summation order.  kBiasBlocks blocks at most.
constexpr int kBiasBlocks = 256;
__global__ __launch_bounds__(256) void bias_grad_slab_kernel(const float* __restrict__ dy, float* __restrict__ db, int64_t M, int K,
                                                              float* __restrict__ slab, int* __restrict__ ticket) {
    __shared__ float red[256];
    __shared__ int last;
    const int rpi = 256 / K;            // rows per iteration
    const int c = threadIdx.x % K, r0 = threadIdx.x / K;
    float s = 0.f;
    if (r0 < rpi)
        for (int64_t m = (int64_t)blockIdx.x * rpi + r0; m < M; m += (int64_t)gridDim.x * rpi)
            s += dy[m * K + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int i = 0; i < rpi; ++i) t += red[threadIdx.x + i * K];
        __hip_atomic_store(slab + (size_t)blockIdx.x * K + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left clean
    float t = 0.f;
    if (r0 < rpi)
        for (int b = r0; b < (int)gridDim.x; b += rpi)
            t += __hip_atomic_load(slab + (size_t)b * K + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    red[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < K) {
        float u = 0.f;
        for (int i = 0; i < rpi; ++i) u += red[threadIdx.x + i * K];
        db[threadIdx.x] = u;
    }
}

// ---------------------------------------------------------------------------------------------
// optional per-launch instrumentation (bench.py roofline): HIP events on the launch stream
// ---------------------------------------------------------------------------------------------
struct ProfRec {
    int kind;            // 0 fwd, 1 dgrad, 2 wgrad
    double flops;
    hipEvent_t e0, e1;
    int shape[7];        // B,H,W,C,K,k,stride
    int family;          // which pipe the launch ran on (kFam*), written by the dispatch code, not derived from the shape
};
// kernel family of the conv launch being dispatched (bench.py prices each against ITS pipe's peak):
//   0 fp32 MFMA (v_mfma_f32_32x32x2_f32: igemm_pipe / igemm_vec / wgrad_pipe / wgrad_vec), 1 three-term bf16 split, six
//   v_mfma_f32_32x32x16_bf16 per fp32 product (conv_x3.hip), 2 bf16 operands on the bf16 MFMA (mixed precision),
//   3 no matrix core: thin-channel / 16-channel / generic vector-ALU kernels
enum { kFamF32 = 0, kFamX3 = 1, kFamBf16 = 2, kFamVector = 3 };
static int g_prof_family = kFamVector;
static bool g_prof_on = false;
static bool g_serialize = false;                   // dpft_profile_serialize: one stream, no event brackets
bool profiling_active() { return g_prof_on || g_serialize; }   // resnet_plan: keep everything on one stream while timing
static std::vector<ProfRec> g_prof;

struct ProfScope {
    bool on;
    hipStream_t st;
    ProfRec r;
    ProfScope(int kind, const dpft_conv_desc* d, hipStream_t s) : on(g_prof_on), st(s) {
        if (!on) return;
        r.kind = kind;
        r.flops = 2.0 * d->B * d->OH * d->OW * (double)d->K * d->kh * d->kw * d->C;
        const int sh[7] = {d->B, d->H, d->W, d->C, d->K, d->kh, d->stride};
        memcpy(r.shape, sh, sizeof(sh));
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, st);
        g_prof_family = kFamVector;      // the early-return paths (conv16, thin-channel) launch no MFMA kernel
    }
    ~ProfScope() {
        if (!on) return;
        r.family = g_prof_family;
        (void)hipEventRecord(r.e1, st);
        g_prof.push_back(r);
    }
};

}  // namespace dpft
#include "conv_pipe.h"
namespace dpft {

// ---------------------------------------------------------------------------------------------
// host-side tile selection
// ---------------------------------------------------------------------------------------------
struct TileChoice {
    int bm, bn, splits;
    bool vec;
    bool x3 = false;      // chosen for the 3 x bf16 split kernels (compute mode 2, multi-tap filters)
};

// Workspace layout (round 4): [ticket header: kWsHeader bytes][split-K partial slabs].  The header holds one int per output
// tile of a split-K launch (IgemmArgs::sk_ticket); it must be ZERO when a buffer is first handed to the library
// (dpft_conv2d_workspace_init) and every launch leaves it zero again.
constexpr size_t kWsHeader = 16384;
constexpr int kWsTickets = (int)(kWsHeader / sizeof(int));
static inline float* ws_slabs(void* workspace) { return workspace ? reinterpret_cast<float*>((char*)workspace + kWsHeader) : nullptr; }
// in-kernel split-K fix-up instead of a reduction launch (DPFT_SK_FIXUP=0: the separate reduction kernels, A/B switch)
static bool sk_fixup_ok(const IgemmArgs& a, int splits, int kind) {      // kind: 1 forward, 2 data gradient, 4 residual data gradient
    static const int on = getenv("DPFT_SK_FIXUP") == nullptr ? 7 : atoi(getenv("DPFT_SK_FIXUP"));
    return (on & kind) && splits > 1 && (a.N & 3) == 0 && a.sub_step <= 1 && (int64_t)splits * a.M * a.N * 4 < (1ll << 31);
}

// 0 = fp32 MFMA (the reference's arithmetic), 1 = bf16 operands / fp32 accumulation in the forward and data-gradient
// GEMMs of the C % 64 == 0 convs (BASELINE.json configs[4], "bf16 mixed precision"); process-wide, set between launches
static int initial_compute_mode() {      // DPFT_CONV_COMPUTE=fp32|bf16|bf16x3 presets the mode (tests, sweeps)
    const char* e = getenv("DPFT_CONV_COMPUTE");
    if (!e) return 0;
    return !strcmp(e, "bf16") ? 1 : (!strcmp(e, "bf16x3") ? 2 : 0);
}
static int g_conv_bf16 = initial_compute_mode();
// fp32 mode: the big multi-tap GEMMs (3x3 / 7x7 filters over C % 64 == 0 channels, >= 2 GFLOP) run as 3 x bf16 split products on
// the bf16 matrix cores (conv_x3.hip) -- fp32 operands, fp32 results, error against fp64 BELOW the fp32 MFMA's on every
// shape of the step (tests/test_gpu_conv_table.py).  dpft_conv_set_split(0) / DPFT_CONV_SPLIT=0: fp32 MFMA everywhere.
static int g_conv_split = getenv("DPFT_CONV_SPLIT") ? atoi(getenv("DPFT_CONV_SPLIT")) != 0 : 1;
static int g_split_override = -1;      // workspace sizing: both answers
int conv_mode_key();
static bool split_on() {
    if (g_split_override >= 0) return g_split_override != 0;
    return g_conv_bf16 == 2 || (g_conv_bf16 == 0 && g_conv_split);
}

// `taps` = kh * kw of the problem (1 = unknown / not asked): in compute mode 2 the filters with more than one tap -- the
// deep reductions -- take the 3 x bf16 split kernels (conv_x3.hip), whose best tiles differ: their operand stream
// (6 bytes per element through L2 -> LDS) is what bounds them (tools/x3_abl.sh), so 128 x 128 tiles with a K split
// that brings the grid to about one workgroup per CU; the 1x1 convs stay on the fp32 MFMA (measured at par there).
static inline int taps_of(const dpft_conv_desc* d) { return (d->kh * d->kw == 1 && d->stride == 1) ? -1 : d->kh * d->kw; }
static TileChoice choose_tile(int M, int N, int C, int ksteps, int taps = 1) {
    TileChoice t;
    t.vec = (C % BKV) == 0;
    t.x3 = false;
    // (only where the GEMM is big enough to be bound by the matrix pipe: the latency-sized problems of the radar encoders
    // measured 2x SLOWER on the 128-row tiles, profiles/r05_x3_table.txt)
    // taps = -1: a 1x1 stride-1 conv on fp32 operands; DPFT_X3_1X1_MINK (tuning aid, 0 = off) sends those with a reduction of at
    // least that many channels to the split kernels too.  Measured in the step (same box, two rounds): off 24.18 / 24.23 ms,
    // >= 1024 channels 24.66 / 24.71, >= 512 24.68 / 24.90, >= 256 27.5 / 25.8 -- the forward loses as well (1.78 -> 1.81 ms per frame)
    static const int x1k = getenv("DPFT_X3_1X1_MINK") ? atoi(getenv("DPFT_X3_1X1_MINK")) : 0;
    const bool deep1x1 = taps == -1 && x1k > 0 && (int64_t)ksteps * BKV >= x1k;
    if (split_on() && (taps > 1 || deep1x1) && t.vec && getenv("DPFT_FORCE_TILE") == nullptr &&
        2.0 * M * N * (double)ksteps * BKV >= 2e9) {
        t.x3 = true;
        if (N >= 128) {
            t.bm = 128; t.bn = 128;
            const int64_t nwg = (int64_t)cdiv(M, 128) * cdiv(N, 128);
            int sp = (int)std::max<int64_t>(1, std::min<int64_t>(8, kNumCU / nwg));      // at most one workgroup per CU: one round
            while (sp > 1 && ksteps / sp < 4) --sp;
            // Few row tiles (batch-1 inference: 15 at layer 3; layer 4 in training): a deep K split of 128 x 128 tiles ends in a
            // fix-up over 4-8 slabs of 64 KB read by ONE workgroup per tile -- 128 x 64 tiles reach the same workgroup count with
            // half the split and quarter-size fix-ups: batch-1 forward 3.50 -> 3.10-3.24 ms, training step unchanged
            // (tools/infer_ab.sh; DPFT_X3_NARROW=0 switches it off, =4 moves the threshold)
            static const int narrow = getenv("DPFT_X3_NARROW") ? atoi(getenv("DPFT_X3_NARROW")) : 2;
            if (narrow && sp > narrow) {
                t.bn = 64;
                const int64_t nwg64 = (int64_t)cdiv(M, 128) * cdiv(N, 64);
                sp = (int)std::max<int64_t>(1, std::min<int64_t>(8, kNumCU / nwg64));
                while (sp > 1 && ksteps / sp < 4) --sp;
            }
            t.splits = sp;
            // (measured, round-5 A/B, docs/history: 128 x 64 tiles without the K split -- which would keep the data gradients' epilogue
            // operand prefetch -- lose: layer-3 forward 82 vs 70 us per call, data gradient 70 vs 68.5)
        } else {
            t.bm = 64; t.bn = 64; t.splits = 1;
        }
        return t;
    }
    if (const char* f = getenv("DPFT_FORCE_TILE")) {      // tuning aid: "bm,bn,splits"
        int bm, bn, sp;
        if (t.vec && sscanf(f, "%d,%d,%d", &bm, &bn, &sp) == 3) {
            t.bm = bm; t.bn = bn; t.splits = sp < 1 ? 1 : sp;
            while (t.splits > 1 && ksteps / t.splits < 2) --t.splits;      // every split keeps at least one pipelined K-step
            return t;
        }
    }
    if (!t.vec) {
        t.bm = 128;
        t.bn = (N <= 32) ? 32 : 64;
        t.splits = 1;
        return t;
    }
    // candidate tiles ordered by per-flop efficiency; pick the one that fills the chip best
    const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    double eff[3] = {1.0, 0.92, 0.80};
    // Short reductions over many row tiles (the 1x1 convs of camera layers 1-2, K <= 256): the kernel is its epilogue
    // (statistics / residual / BatchNorm-reduction operands, staged stores), and with 128 x 128 tiles all workgroups of a
    // wave reach it together; 128 x 64 measured 10-25 % faster there (round-3 tile A/B, docs/history: 176 -> 129 us on the
    // 128x228 256<-64 data gradient), not on the 57-row-tile grids of layer 3.
    static const bool shortk_rule = getenv("DPFT_SHORTK_TILE") == nullptr || atoi(getenv("DPFT_SHORTK_TILE")) != 0;      // A/B switch
    if (shortk_rule && (int64_t)ksteps * BKV <= 256 && (int64_t)cdiv(M, 128) * cdiv(N, 128) >= 768) {
        eff[0] = 0.88; eff[1] = 1.0; eff[2] = 0.85;
    }
    if (const char* e = getenv("DPFT_TILE_EFF")) {      // tuning aid: "e128x128,e128x64,e64x64[,max row tiles it applies to]"
        double a0, a1, a2; int rows = 1 << 30;
        if (sscanf(e, "%lf,%lf,%lf,%d", &a0, &a1, &a2, &rows) >= 3 && cdiv(M, 128) <= rows) { eff[0] = a0; eff[1] = a1; eff[2] = a2; }
    }
    double best = -1;
    t.bm = 64; t.bn = 64;
    for (int i = 0; i < 3; ++i) {
        const int bm = cand[i][0], bn = cand[i][1];
        if (bn > 64 && N <= 64) continue;
        const int64_t nwg = (int64_t)cdiv(M, bm) * cdiv(N, bn);
        const int slots = kNumCU * 2;  // two resident workgroups per CU
        const double waves = (double)nwg / slots;
        const double fill = waves / ceil(waves);
        // padding waste in N and M
        const double pad = ((double)M * N) / ((double)cdiv(M, bm) * bm * (double)cdiv(N, bn) * bn);
        const double score = eff[i] * fill * pad;
        if (score > best) {
            best = score;
            t.bm = bm;
            t.bn = bn;
        }
    }
    // split-K when even the smallest tile leaves most CUs idle (radar branches, layer4)
    const int64_t nwg = (int64_t)cdiv(M, t.bm) * cdiv(N, t.bn);
    t.splits = 1;
    // DPFT_SPLIT_BELOW (tuning aid): the isolated 232-tile layer-4 1x1 forward runs 52.9 us split in three vs 35.9 us
    // unsplit (round-3 tile A/B, docs/history), but over the whole step 128 / 192 / 256 are within run-to-run noise (conv time 28.2 /
    // 27.9 / 28.0 ms): the threshold stays at one workgroup per CU.
    static const int split_below = getenv("DPFT_SPLIT_BELOW") ? atoi(getenv("DPFT_SPLIT_BELOW")) : kNumCU;
    // (round 4) 200 ... 255 tiles with a SHALLOW reduction (<= 32 K-steps: the 1x1 convs of camera layer 4, 232 tiles)
    // stay unsplit: the split's slab round trip + reduction launch cost more than the idle tenth of the chip
    // (round-3 tile A/B, docs/history: 16x29 2048->512 forward 53.7 -> 36.9 us, 512->2048 data gradient 55.1 -> 40.1 us)
    const bool shallow_nearly_full = nwg >= 200 && ksteps <= 32;
    if (nwg < split_below && ksteps >= 8 && !shallow_nearly_full) {
        int s = (int)((kNumCU * 2 + nwg - 1) / nwg);
        s = s > 16 ? 16 : s;
        while (s > 1 && ksteps / s < 4) --s;
        t.splits = s;
    }
    return t;
}

// act16 = 2 (bf16 activations and bf16 weights): the 256-row tiles of conv_b16w.hip (eight waves, 64 x 128 / 64 x 64 outputs
// per wave); the tile may carry a K split (in-launch fix-up).  The tile height must not depend on the workspace:
// dpft_conv2d_stats_tiles answers without one.  Alone, the 256 x 256 tiles win on the wide short-K 1x1 convs (batch 8,
// 256 -> 1024 forward + statistics 30.9 -> 22.7 us, its data gradient 28.0 -> 19.3, 128 -> 512 forward 46.1 -> 35.8) and lose on
// N = 256 problems (57 row tiles: a K split's slab round trip costs more than the idle CUs).  INSIDE the training step the data
// gradients lose -- a 512-thread workgroup that owns a CU's whole LDS cannot share the CU with the weight-gradient stream's
// workgroups the way three 48 KB workgroups do -- but the FORWARD has no weight-gradient stream beside it: bf16 batch 8, same box,
// two rounds: off 22.91 / 22.78 ms, forward only 22.60 / 22.60, forward + data gradients 24.14 / 24.21 (profiles/r06_b16w_*.txt).
// Default: forward only (DPFT_B16W=2); =1 both, =0 off.
static bool big16_tile(const dpft_conv_desc* d, const IgemmArgs& a, bool dgrad, bool has_ws, TileChoice& t) {
    static const int on = getenv("DPFT_B16W") ? atoi(getenv("DPFT_B16W")) : 2;      // 0 off | 1 forward + data gradients | 2 forward only (default)
    if (!on || d->act16 != 2 || (a.C % BKV) != 0 || (a.N % 128) != 0 || getenv("DPFT_FORCE_TILE") != nullptr) return false;
    if (on == 2 && dgrad) return false;      // 2: forward convs only (no weight-gradient stream beside them)
    if (dgrad && d->stride > 1) return false;      // (parity classes / the all-tap form keep their kernels)
    if ((int64_t)a.B * a.H * a.W * a.C >= (1ll << 29) || (int64_t)a.N * a.Ktot >= (1ll << 29)) return false;
    const int64_t mt = cdiv(a.M, 256);
    static const int min256 = getenv("DPFT_B16W_MIN256") ? atoi(getenv("DPFT_B16W_MIN256")) : 100;
    static const int min128 = getenv("DPFT_B16W_MIN128") ? atoi(getenv("DPFT_B16W_MIN128")) : (1 << 30);      // 256 x 128 tiles: measured at par or behind the four-wave kernels (profiles/r06_b16w_*.txt): off
    const int ksteps = a.Ktot / BKV;
    if ((a.N % 256) == 0 && mt * (a.N / 256) >= min256) {
        t.bm = 256; t.bn = 256; t.splits = 1; t.vec = true; t.x3 = false;
        return true;
    }
    const int64_t n128 = mt * (a.N / 128);
    if (n128 >= min128) {
        t.bm = 256; t.bn = 128; t.vec = true; t.x3 = false;
        int sp = 1;
        static const int max_split = getenv("DPFT_B16W_SPLIT") ? atoi(getenv("DPFT_B16W_SPLIT")) : 4;      // tuning aid
        if (has_ws && n128 < 200 && max_split > 1) {
            sp = (int)std::min<int64_t>(max_split, (kNumCU + n128 / 2) / n128);
            while (sp > 1 && ksteps / sp < 8) --sp;
        }
        t.splits = sp < 1 ? 1 : sp;
        return true;
    }
    return false;
}

static void fill_igemm(IgemmArgs& a, const dpft_conv_desc* d, bool dgrad) {
    a.obn = nullptr; a.oadd = nullptr; a.orelu = 0;
    memset(&a, 0, sizeof(a));
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    if (!dgrad) {
        a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C;
        a.OH = d->OH; a.OW = d->OW; a.N = d->K;
    } else {
        a.B = d->B; a.H = d->OH; a.W = d->OW; a.C = d->K;
        a.OH = d->H; a.OW = d->W; a.N = d->C;
    }
    a.M = a.B * a.OH * a.OW;
    a.Ktot = a.kh * a.kw * a.C;
    a.ksteps = (a.C % BKV) == 0 ? a.Ktot / BKV : cdiv(a.Ktot, BK);
}

// launch with dynamic LDS; raises the per-kernel dynamic-LDS cap once (tiles above 64 KiB)
template <typename K, typename A>
static void launch_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, const A& args) {
    static LdsGrant grant;      // one per kernel instantiation: the largest size allowed so far, per device
    (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args);
}


template <bool DGRAD>
static int launch_igemm(IgemmArgs& a, const TileChoice& t, bool pro, hipStream_t st) {
    if (t.vec) {      // the vector loader keeps a 32-bit tap mask per row and 32-bit buffer offsets
        DPFT_REQUIRE(a.kh <= 8 && a.kw <= 8, "conv: the C %% 64 == 0 path supports filters up to 8x8 (%dx%d)", a.kh, a.kw);
        DPFT_REQUIRE((int64_t)a.B * a.H * a.W * a.C < (1ll << 29) && (int64_t)a.N * a.Ktot < (1ll << 29),
                     "conv: operand larger than 2 GiB");
    }
    static const int ablate_env = getenv("DPFT_ABLATE") ? atoi(getenv("DPFT_ABLATE")) : 0;
    a.ablate = ablate_env;
    const bool nonlin = DGRAD && t.vec && a.stride > 1 && a.sub_step <= 1;
    const int bm = nonlin ? 64 : t.bm, bn = nonlin ? 64 : t.bn;
    a.mtiles = cdiv(a.M, bm);
    a.ntiles = cdiv(a.N, bn);
    a.splits = t.splits;
    a.ksteps_per_split = cdiv(a.ksteps, a.splits);
    const int nwg = a.mtiles * a.ntiles * a.splits;
    DPFT_REQUIRE(!a.sk_ticket || a.mtiles * a.ntiles <= kWsTickets, "conv: %d output tiles exceed the split-K ticket header", a.mtiles * a.ntiles);
    dim3 grid(nwg), block(256);
    g_prof_family = !t.vec ? kFamVector : (g_conv_bf16 == 1 ? kFamBf16 : kFamF32);      // (the x3 branches below override)
    if (nonlin) {
        if constexpr (DGRAD) {
            constexpr size_t lds = (size_t)(64 + 64) * LDK * sizeof(float);
            launch_lds(igemm_vec_kernel<64, 64, 2, 2, true, false, false>, grid, block, lds, st, a);
        }
        return check_launch("conv igemm (strided dgrad, all taps)");
    }
    // bf16 activations AND bf16 weights (act16 = 2): both operands go to the matrix cores untouched -- the pipelined kernel
    // with 2-byte elements, everything by LDS-DMA
    if (t.vec && a.w16 && a.x16 && pro && !DGRAD && a.pro_relu && !nonlin) {
        // bf16 operands WITH the producer's BatchNorm + ReLU (round 6): the A operand on the register route, its parameters from an
        // LDS table; weights by LDS-DMA.  Tiles as the prologue-free form (a 256-row choice falls back to 128 x 128).
        g_prof_family = kFamBf16;
        if constexpr (!DGRAD) {
            const int pbm = t.bm == 64 ? 64 : 128, pbn = (t.bm == 64 || t.bn == 64) ? 64 : 128;
            a.mtiles = cdiv(a.M, pbm); a.ntiles = cdiv(a.N, pbn);
            const dim3 pgrid(a.mtiles * a.ntiles * a.splits);
            auto gop = [&](auto kernel, int pbk, size_t lds) {
                a.ksteps = a.ksteps * BKV / pbk;
                a.ksteps_per_split = cdiv(a.ksteps, a.splits);
                launch_lds(kernel, pgrid, block, lds + (size_t)12 * a.C, st, a);
            };
#define PIPE16P_LDS(BM_, BN_, PBK_) std::max((size_t)2 * (BM_ + BN_) * PBK_ * 2, (size_t)BM_ * (BN_ + 4) * 4 + (size_t)3 * BN_ * 4)
            if (pbm == 128 && pbn == 128) gop(igemm_pipe_kernel<128, 128, 2, 2, 64, false, true, true>, 64, PIPE16P_LDS(128, 128, 64));
            else if (pbm == 128) gop(igemm_pipe_kernel<128, 64, 2, 2, 64, false, true, true>, 64, PIPE16P_LDS(128, 64, 64));
            else if (a.C % 128 == 0) gop(igemm_pipe_kernel<64, 64, 2, 2, 128, false, true, true>, 128, PIPE16P_LDS(64, 64, 128));
            else gop(igemm_pipe_kernel<64, 64, 2, 2, 64, false, true, true>, 64, PIPE16P_LDS(64, 64, 64));
#undef PIPE16P_LDS
        }
        return check_launch("conv igemm (pipelined, bf16 operands, BatchNorm + ReLU prologue)");
    }
    if (t.vec && a.w16 && a.x16 && !pro && !nonlin) {
        g_prof_family = kFamBf16;
        if (t.bm == 256) return launch_igemm_b16w(a, t.bm, t.bn, DGRAD, st);      // conv_b16w.hip: 256-row tiles, eight waves
        auto go16 = [&](auto kernel, int pbk, size_t lds) {
            a.ksteps = a.ksteps * BKV / pbk;      // (a parity class of a strided data gradient covers a subset of the taps)
            a.ksteps_per_split = cdiv(a.ksteps, a.splits);
            launch_lds(kernel, grid, block, lds, st, a);
        };
#define PIPE16_LDS(BM_, BN_, PBK_) std::max((size_t)2 * (BM_ + BN_) * PBK_ * 2, (size_t)BM_ * (BN_ + 4) * 4 + (size_t)3 * BN_ * 4)
        if constexpr (DGRAD) {      // epilogue operands requested with the first tile (EpiPrefetch; a.epf = its MODE)
            bool took = true;
            if (a.epf == 1 && t.bm == 128 && t.bn == 64) go16(igemm_pipe_kernel<128, 64, 2, 2, 64, true, false, true, 1>, 64, PIPE16_LDS(128, 64, 64));
            else if (a.epf == 2 && t.bm == 128 && t.bn == 64) go16(igemm_pipe_kernel<128, 64, 2, 2, 64, true, false, true, 2>, 64, PIPE16_LDS(128, 64, 64));
            else if (a.epf == 2 && t.bm == 64 && t.bn == 64 && a.C % 128 == 0) go16(igemm_pipe_kernel<64, 64, 2, 2, 128, true, false, true, 2>, 128, PIPE16_LDS(64, 64, 128));
            else if (a.epf == 2 && t.bm == 64 && t.bn == 64) go16(igemm_pipe_kernel<64, 64, 2, 2, 64, true, false, true, 2>, 64, PIPE16_LDS(64, 64, 64));
            else took = false;
            if (took) return check_launch("conv igemm (pipelined, bf16 operands, epilogue prefetch)");
        }
        if (t.bm == 128 && t.bn == 128) go16(igemm_pipe_kernel<128, 128, 2, 2, 64, DGRAD, false, true>, 64, PIPE16_LDS(128, 128, 64));
        else if (t.bm == 128 && t.bn == 64) go16(igemm_pipe_kernel<128, 64, 2, 2, 64, DGRAD, false, true>, 64, PIPE16_LDS(128, 64, 64));
        else if (a.C % 128 == 0) go16(igemm_pipe_kernel<64, 64, 2, 2, 128, DGRAD, false, true>, 128, PIPE16_LDS(64, 64, 128));
        else go16(igemm_pipe_kernel<64, 64, 2, 2, 64, DGRAD, false, true>, 64, PIPE16_LDS(64, 64, 64));
#undef PIPE16_LDS
        return check_launch("conv igemm (pipelined, bf16 operands)");
    }
    // operands given as three bf16 planes (dpft_conv_desc::a_planes / w_planes): the copy-only split kernel of conv_x3.hip
    if (t.vec && a.x3 && a.w3 && !pro && !a.x16 && !a.y16 && !a.w16 &&
        ((t.bm == 128 && (t.bn == 128 || t.bn == 64)) || (t.bm == 64 && t.bn == 64))) {
        g_prof_family = kFamX3;
        return launch_igemm_x3(a, t.bm, t.bn, DGRAD, false, st);
    }
    a.x3 = a.w3 = nullptr;
    // fp32 results from the bf16 matrix cores (compute mode 2, conv_x3.hip): three-term split of both operands, six MFMAs
    static const bool x3_all = getenv("DPFT_X3_ALL") != nullptr && atoi(getenv("DPFT_X3_ALL")) != 0;      // tuning aid: 1x1 convs too
    if (t.vec && (t.x3 || (g_conv_bf16 == 2 && (x3_all || getenv("DPFT_FORCE_TILE")))) && g_conv_bf16 != 1 && !a.x16 && !a.y16 && !a.w16 && (!pro || a.pro_relu) &&
        ((t.bm == 128 && (t.bn == 128 || t.bn == 64)) || (t.bm == 64 && t.bn == 64))) {
        g_prof_family = kFamX3;
        return launch_igemm_x3(a, t.bm, t.bn, DGRAD, pro, st);
    }
    // fp32, linear taps: the software-pipelined kernel (conv_pipe.h) -- LDS-DMA operands, one barrier per K-step.
    // DPFT_PIPE=0 keeps igemm_vec_kernel (A/B measurements).
    static const int pipe_env = getenv("DPFT_PIPE") ? atoi(getenv("DPFT_PIPE")) : 3;      // bit 0: igemm, bit 1: wgrad
    if (t.vec && g_conv_bf16 != 1 && (pipe_env & 1) && !a.x16 && !a.y16 && (!pro || a.pro_relu)) {
        g_prof_family = kFamF32;
        static const int shortk_env = getenv("DPFT_SHORTK") ? atoi(getenv("DPFT_SHORTK")) : 0;      // tuning aid
        const bool short_k = shortk_env > 0 && a.Ktot <= shortk_env;
        auto go = [&](auto kernel, int pbk, size_t lds) {
            a.ksteps = a.ksteps * BKV / pbk;
            a.ksteps_per_split = cdiv(a.ksteps, a.splits);
            launch_lds(kernel, grid, block, lds, st, a);
        };
#define PIPE_LDS(BM_, BN_, PBK_) std::max((size_t)2 * (BM_ + BN_) * PBK_ * 4, (size_t)BM_ * (BN_ + 4) * 4 + (size_t)3 * BN_ * 4)
#define LAUNCH_PIPE(BM_, BN_, PBK_)                                                                       \
    do {                                                                                                  \
        if (pro) go(igemm_pipe_kernel<BM_, BN_, 2, 2, PBK_, DGRAD, !DGRAD>, PBK_, PIPE_LDS(BM_, BN_, PBK_) + (size_t)12 * a.C); \
        else go(igemm_pipe_kernel<BM_, BN_, 2, 2, PBK_, DGRAD, false>, PBK_, PIPE_LDS(BM_, BN_, PBK_));      \
    } while (0)
        if constexpr (DGRAD) {      // epilogue operands requested with the first tile (EpiPrefetch; a.epf = its MODE)
            bool took = !pro;
            if (pro) {}
            else if (a.epf == 1 && t.bm == 128 && t.bn == 64) go(igemm_pipe_kernel<128, 64, 2, 2, 32, true, false, false, 1>, 32, PIPE_LDS(128, 64, 32));
            else if (a.epf == 2 && t.bm == 128 && t.bn == 64) go(igemm_pipe_kernel<128, 64, 2, 2, 32, true, false, false, 2>, 32, PIPE_LDS(128, 64, 32));
            else if (a.epf == 2 && t.bm == 64 && t.bn == 64 && !short_k) go(igemm_pipe_kernel<64, 64, 2, 2, 64, true, false, false, 2>, 64, PIPE_LDS(64, 64, 64));
            else took = false;
            if (took) return check_launch("conv igemm (pipelined, epilogue prefetch)");
        }
        if constexpr (!DGRAD) {      // inference epilogue with a residual: the residual is requested with the first tile (EpiPrefetch mode 3)
            bool took = !pro && a.epf == 3;
            if (!took) {}
            else if (t.bm == 128 && t.bn == 128) go(igemm_pipe_kernel<128, 128, 2, 2, 32, false, false, false, 3>, 32, PIPE_LDS(128, 128, 32));
            else if (t.bm == 128 && t.bn == 64) go(igemm_pipe_kernel<128, 64, 2, 2, 32, false, false, false, 3>, 32, PIPE_LDS(128, 64, 32));
            else if (t.bm == 64 && t.bn == 64 && !short_k) go(igemm_pipe_kernel<64, 64, 2, 2, 64, false, false, false, 3>, 64, PIPE_LDS(64, 64, 64));
            else took = false;
            if (took) return check_launch("conv igemm (pipelined, inference epilogue, residual prefetch)");
        }
        if (t.bm == 128 && t.bn == 128) LAUNCH_PIPE(128, 128, 32);
        else if (t.bm == 128 && t.bn == 64) LAUNCH_PIPE(128, 64, 32);
        else if (t.bm == 64 && t.bn == 128) LAUNCH_PIPE(64, 128, 32);
        else if (short_k) LAUNCH_PIPE(64, 64, 32);      // 32 KB of LDS: more resident workgroups to overlap prologues / epilogues
        else LAUNCH_PIPE(64, 64, 64);
#undef LAUNCH_PIPE
#undef PIPE_LDS
        return check_launch("conv igemm (pipelined)");
    }
    // K-split form (512 threads per tile) where the grid leaves the chip latency-bound: fewer than 3 workgroups per CU and
    // a reduction deep enough to amortise the hand-over.  Measured (tools/ksplit_ab.sh): -8...-10 % on the data gradients
    // of the 64 x 64-tile problems (layer-3 3x3: 102 -> 93.5 us), nothing or a loss on the forward kernels (prologue,
    // statistics epilogue) and on the 128-row tiles (one workgroup per CU either way) -- waves of ONE workgroup march in
    // step between barriers, so they do not fill each other's stalls the way a second resident workgroup does.  (A two-wave
    // 64 x 32 tile -- twice the workgroups -- was 15-90 % slower on the same problems: tools/tile6432.sh, removed.)
    static const int ks_env = getenv("DPFT_KSPLIT") ? atoi(getenv("DPFT_KSPLIT")) : -1;      // tuning aid: 0 / 1 force
    const bool ks2 = DGRAD && !pro && bm == 64 && bn == 64 &&
                     (ks_env >= 0 ? ks_env != 0 : (nwg < kNumCU * 3 && a.ksteps_per_split >= 8));
    const dim3 block2(512);
#define LAUNCH_VEC(BM_, BN_, WGM_, WGN_)                                                      \
    do {                                                                                      \
        constexpr size_t lds = (size_t)(BM_ + BN_) * LDK * sizeof(float);                     \
        if (g_conv_bf16 == 1 && ks2) {                                                        \
            if constexpr (DGRAD && BM_ == 64 && BN_ == 64)                                    \
                launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, true, false, true, true, 2>, grid, block2, lds, st, a); \
        } else if (g_conv_bf16 == 1) {                                                        \
            if (pro) launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, DGRAD, !DGRAD, true, true>, grid, block, lds, st, a); \
            else launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, DGRAD, false, true, true>, grid, block, lds, st, a);      \
        } else if (ks2) {                                                                     \
            if constexpr (DGRAD && BM_ == 64 && BN_ == 64)                                    \
                launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, true, false, true, false, 2>, grid, block2, lds, st, a); \
        } else if (pro) launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, DGRAD, !DGRAD>, grid, block, lds, st, a); \
        else launch_lds(igemm_vec_kernel<BM_, BN_, WGM_, WGN_, DGRAD, false>, grid, block, lds, st, a);      \
    } while (0)
    if (t.vec) {
        if (t.bm == 128 && t.bn == 128) LAUNCH_VEC(128, 128, 2, 2);
        else if (t.bm == 128 && t.bn == 64) LAUNCH_VEC(128, 64, 2, 2);
        else LAUNCH_VEC(64, 64, 2, 2);
    } else {
        if (t.bn == 32) hipLaunchKernelGGL((igemm_gen_kernel<32, DGRAD>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((igemm_gen_kernel<64, DGRAD>), grid, block, 0, st, a);
    }
#undef LAUNCH_VEC
    return check_launch("conv igemm");
}

}  // namespace dpft
#include "conv16_kernels.h"
namespace dpft {

// thin-channel fast paths (conv16_kernels.h); `handled` tells the caller whether the launch was taken
// `db` (may be null): also wanted -- the bias gradient sum_p dy[p][k]; *bias_done tells whether the launch carried it
static int thin_wgrad(const dpft_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                      hipStream_t st, bool& handled, float* db = nullptr, bool* bias_done = nullptr) {
    handled = false;
    if (bias_done) *bias_done = false;
    if (!workspace) return DPFT_OK;
    workspace = ws_slabs(workspace);      // the slabs start behind the ticket header
    if (conv16_matches(d)) {
        handled = true;
        static const bool tiled = getenv("DPFT_WGRAD16_TILED") == nullptr || atoi(getenv("DPFT_WGRAD16_TILED")) != 0;      // A/B switch
        int nb;
        if (tiled) {
            nb = conv16_wgrad_tiled_blocks(d);
            Wgrad16TArgs a{x, dy, (float*)workspace, d->B, d->H, d->W, cdiv(d->W, T16W), cdiv(d->H, T16H),
                           d->B * cdiv(d->H, T16H) * cdiv(d->W, T16W), db ? 1 : 0};
            hipLaunchKernelGGL(wgrad16_3x3_tiled_kernel, dim3(nb), dim3(256), 0, st, a);
            if (db) {
                int rc = check_launch("conv16 wgrad");
                if (rc) return rc;
                hipLaunchKernelGGL(slab_reduce_bias_kernel, dim3(cdiv(2320, 16)), dim3(256), 0, st, (const float*)workspace, dw, db,
                                   2320, nb, 0, 2304);
                *bias_done = true;
                return check_launch("conv16 wgrad + bias reduce");
            }
        } else {
            nb = conv16_wgrad_blocks(d);
            Wgrad16Args a{x, dy, (float*)workspace, d->B, d->H, d->W, cdiv(d->W, 4), (long)d->B * d->H * cdiv(d->W, 4)};
            hipLaunchKernelGGL(wgrad16_3x3_kernel, dim3(nb), dim3(256), 0, st, a);
        }
        int rc = check_launch("conv16 wgrad");
        if (rc) return rc;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(2304, 16)), dim3(256), 0, st, (const float*)workspace, dw, 2304, nb);
        return check_launch("conv16 wgrad reduce");
    }
    static const bool stem_tiled = getenv("DPFT_WGRAD_STEM") == nullptr || atoi(getenv("DPFT_WGRAD_STEM")) != 0;      // A/B switch
    if (stem_tiled && wgrad_stem7_matches(d)) {
        handled = true;
        const int tw = cdiv(d->OW, STEM_OW), th = cdiv(d->OH, STEM_OH);
        const int tiles = d->B * th * tw;
        const int nb = std::max(1, std::min(kNumCU * 2, tiles));      // (<= 512 slabs: the workspace bound of the pixel-split path)
        WgradStemArgs a{x, dy, (float*)workspace, d->B, d->H, d->W, d->OH, d->OW, tw, th, tiles};
        hipLaunchKernelGGL(wgrad_stem7_kernel, dim3(nb), dim3(256), 0, st, a);
        int rc = check_launch("stem wgrad");
        if (rc) return rc;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(64 * 147, 16)), dim3(256), 0, st, (const float*)workspace, dw, 64 * 147, nb);
        return check_launch("stem wgrad reduce");
    }
    if (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0) {
        const long M = (long)d->B * d->H * d->W;
        const int nb = (int)std::max<long>(1, std::min<long>(kNumCU * 2, M / 1024));
        const int kc = d->K * d->C;
#define THIN_1X1(K_, C_)                                                                                     \
    if (d->K == K_ && d->C == C_) {                                                                          \
        handled = true;                                                                                      \
        hipLaunchKernelGGL((wgrad1x1_small_kernel<K_, C_>), dim3(nb), dim3(256), 0, st, x, dy, (float*)workspace, M); \
    }
        if (d->K == 16 && (d->C == 3 || d->C == 6)) {
            handled = true;
            if (db) {      // bias gradient as a virtual input channel of ones
                if (d->C == 3) hipLaunchKernelGGL((wgrad1x1_k16_kernel<3, true>), dim3(nb), dim3(256), 0, st, x, dy, (float*)workspace, M);
                else hipLaunchKernelGGL((wgrad1x1_k16_kernel<6, true>), dim3(nb), dim3(256), 0, st, x, dy, (float*)workspace, M);
                int rc = check_launch("thin 1x1 wgrad");
                if (rc) return rc;
                const int cols = d->C + 1;
                hipLaunchKernelGGL(slab_reduce_bias_kernel, dim3(cdiv(16 * cols, 16)), dim3(256), 0, st, (const float*)workspace, dw, db,
                                   16 * cols, nb, cols, 0);
                *bias_done = true;
                return check_launch("thin 1x1 wgrad + bias reduce");
            }
            if (d->C == 3) hipLaunchKernelGGL((wgrad1x1_k16_kernel<3>), dim3(nb), dim3(256), 0, st, x, dy, (float*)workspace, M);
            else hipLaunchKernelGGL((wgrad1x1_k16_kernel<6>), dim3(nb), dim3(256), 0, st, x, dy, (float*)workspace, M);
        } else THIN_1X1(3, 6)
#undef THIN_1X1
        if (handled) {
            int rc = check_launch("thin 1x1 wgrad");
            if (rc) return rc;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(kc, 16)), dim3(256), 0, st, (const float*)workspace, dw, kc, nb);
            return check_launch("thin 1x1 wgrad reduce");
        }
    }
    return DPFT_OK;
}

static int check_desc(const dpft_conv_desc* d) {
    DPFT_REQUIRE(d != nullptr, "conv: null descriptor");
    DPFT_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0, "conv: non-positive dims");
    DPFT_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0, "conv: bad filter geometry");
    const int oh = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    const int ow = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DPFT_REQUIRE(oh == d->OH && ow == d->OW, "conv: OH/OW (%d,%d) inconsistent with geometry (%d,%d)",
                 d->OH, d->OW, oh, ow);
    DPFT_REQUIRE((int64_t)d->B * d->H * d->W * d->C < (1ll << 31) &&
                 (int64_t)d->B * d->OH * d->OW * d->K < (1ll << 31), "conv: tensor too large for 32-bit row indices");
    DPFT_REQUIRE(d->act16 >= 0 && d->act16 <= 2, "conv: act16 must be 0, 1 or 2 (is the descriptor zero-initialised?)");
    // bf16 activation storage rides on the vector loaders / the staged epilogue only
    DPFT_REQUIRE(!d->act16 || (d->C % BKV == 0 && d->K % BKV == 0 && d->kh <= 8 && d->kw <= 8),
                 "conv: act16 (bf16 activation storage) needs C %% 64 == 0 and K %% 64 == 0 (C=%d, K=%d)", d->C, d->K);
    return DPFT_OK;
}

}  // namespace dpft

using namespace dpft;

extern "C" int64_t dpft_conv2d_workspace_bytes(const dpft_conv_desc* d) {
    if (check_desc(d) != DPFT_OK) return -1;
    // worst case over fwd / dgrad / wgrad split-K partials
    int64_t best = 0;
    if (conv16_matches(d)) best = std::max<int64_t>(best, (int64_t)conv16_wgrad_blocks(d) * 2320 * 4);      // (+ bias column sums)
    if (d->kh == 1 && d->kw == 1 && d->K <= 16 && d->C <= 8) best = std::max<int64_t>(best, (int64_t)kNumCU * 2 * d->K * (d->C + 1) * 4);
    for (int mode : {0, 1}) {      // with or without the split kernels, whichever the launches will run with (they pick their own splits)
        g_split_override = mode;
        for (int dg = 0; dg < 2; ++dg) {
            IgemmArgs a; fill_igemm(a, d, dg != 0);
            TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, d->act16 ? 1 : taps_of(d));
            if (d->act16) t.splits = 1;
            (void)big16_tile(d, a, dg != 0, true, t);
            if (t.splits > 1) best = std::max<int64_t>(best, (int64_t)t.splits * a.M * a.N * 4);
        }
    }
    g_split_override = -1;
    // wgrad: pixel-split partial slabs, at most min(512 splits, 64 MiB)
    {
        const int64_t wbytes = (int64_t)d->K * d->kh * d->kw * d->C * 4;
        const int64_t ms = std::max<int64_t>(1, std::min<int64_t>(512, (64ll << 20) / wbytes));
        best = std::max<int64_t>(best, ms * wbytes);
    }
    best = std::max<int64_t>(best, (int64_t)kBiasBlocks * std::min(d->K, 256) * 4);      // bias_grad_slab_kernel's partial sums
    return best + (int64_t)kWsHeader;
}

extern "C" int64_t dpft_conv2d_workspace_header_bytes(void) { return (int64_t)kWsHeader; }

extern "C" int dpft_conv2d_workspace_init(void* workspace, dpft_stream_t stream) {
    DPFT_REQUIRE(workspace, "conv2d_workspace_init: null workspace");
    return dpft::zero_fill(workspace, kWsHeader, stream);
}

extern "C" int dpft_split_planes_f32(const float* src, void* planes, int64_t n, dpft_stream_t stream) {
    return dpft::split_planes(src, planes, n, (hipStream_t)stream);
}

extern "C" int dpft_conv_set_compute(int32_t mode) {
    DPFT_REQUIRE(mode >= 0 && mode <= 2, "conv_set_compute: mode 0 (fp32), 1 (bf16 operands) or 2 (3 x bf16 split), got %d", mode);
    dpft::g_conv_bf16 = mode;
    return DPFT_OK;
}

extern "C" int32_t dpft_conv_get_compute() { return dpft::g_conv_bf16; }

extern "C" int dpft_conv_set_split(int32_t on) {
    dpft::g_conv_split = on != 0;
    return DPFT_OK;
}

extern "C" int32_t dpft_conv_get_split() { return dpft::g_conv_split; }

int dpft::conv_mode_key() { return dpft::g_conv_bf16 * 2 + (dpft::g_conv_split ? 1 : 0); }

extern "C" int32_t dpft_conv2d_stats_tiles(const dpft_conv_desc* d, int32_t* tile_rows) {
    return dpft_conv2d_stats_tiles_pro(d, 0, tile_rows);
}

// `pro`: the launch will carry a BatchNorm + ReLU operand prologue (bf16 operands: the 256-row tiles have none)
extern "C" int32_t dpft_conv2d_stats_tiles_pro(const dpft_conv_desc* d, int32_t pro, int32_t* tile_rows) {
    if (check_desc(d) != DPFT_OK) return -1;
    IgemmArgs a; fill_igemm(a, d, false);
    {
        int tr = 0;
        if (g_conv_bf16 != 1 && stream1x1_match(d, &tr)) {      // conv_stream.hip: tile = the rows of one workgroup
            if (tile_rows) *tile_rows = tr;
            return cdiv(a.M, tr);
        }
    }
    TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, d->act16 ? 1 : taps_of(d));      // (as conv_fwd_bnfinal asks)
    if (!pro) (void)big16_tile(d, a, false, false, t);
    if (pro && d->act16 == 2 && t.bm != 64) t.bm = 128;      // (launch_igemm: the bf16 prologue kernels' tiles)
    if (tile_rows) *tile_rows = t.bm;
    return cdiv(a.M, t.bm);
}

extern "C" int dpft_conv2d_nhwc_fwd_f32(const dpft_conv_desc* d, const float* x, const float* w,
                                        const float* bias, const float* pro_bn, int32_t pro_relu,
                                        float* y, float* stats, void* workspace, dpft_stream_t stream) {
    return dpft::conv_fwd_bnfinal(d, x, w, bias, pro_bn, pro_relu, y, stats, workspace, stream, nullptr);
}

// Forward conv with the train-mode BatchNorm finalize folded into its epilogue (BnFinalFuse, common.h).  `fuse->applied`
// tells whether the launch carried it (not on split-K / thin-channel / bf16-storage paths: the caller then runs
// dpft_bn_finalize_f32 on `stats` as before).
int dpft::conv_fwd_bnfinal(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, const float* pro_bn,
                           int32_t pro_relu, float* y, float* stats, void* workspace, dpft_stream_t stream,
                           BnFinalFuse* fuse, const BnSumsRef* pro_sums, bool* pro_sums_used) {
    if (fuse) fuse->applied = false;
    if (pro_sums_used) *pro_sums_used = false;
    if (pro_sums && !pro_sums->sums) pro_sums = nullptr;
    DPFT_REQUIRE(!pro_sums || (pro_bn && pro_sums_used), "conv fwd: a prologue from column sums needs the BN block's address and the answer slot");
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(x && w && y, "conv fwd: null tensor");
    {
        static const char* skip = getenv("DPFT_SKIP");      // timing experiments (see dpft_conv2d_nhwc_wgrad_f32)
        if (skip && ((strstr(skip, "fwd3x3") && d->kh == 3) || (strstr(skip, "fwd1x1") && d->kh == 1)) && (int64_t)d->B * d->OH * d->OW >= 4096) return DPFT_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(0, d, st);
    if (conv16_matches(d) && !pro_bn && !stats) return conv16_forward(d, x, w, bias, y, st);
    static const bool thin_fwd = getenv("DPFT_THIN_FWD") == nullptr || atoi(getenv("DPFT_THIN_FWD")) != 0;      // A/B switch
    if (thin_fwd && conv1x1_to16_matches(d) && !pro_bn && !stats) return conv1x1_to16_forward(d, x, w, bias, y, st);
    if (g_conv_bf16 != 1 && !bias && (!pro_bn || pro_relu) && !(fuse && fuse->acc) && stream1x1_match(d, nullptr)) {
        g_prof_family = kFamF32;      // short reduction, wide output, large map: the streaming kernel (conv_stream.hip)
        unsigned long long* bns = fuse && fuse->sums && stats && (int64_t)d->B * d->OH * d->OW < (1 << 22) ? fuse->sums : nullptr;
        if (bns) fuse->applied = true;
        if (pro_sums) *pro_sums_used = true;
        return launch_stream1x1(d, x, w, pro_bn, y, stats, nullptr, nullptr, 0, st, bns, pro_sums);
    }
    IgemmArgs a; fill_igemm(a, d, false);
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.stats = stats;
    a.pro = pro_bn; a.pro_relu = pro_relu;
    a.x16 = a.y16 = d->act16 != 0;
    a.w16 = d->act16 == 2;
    if (d->a_planes && d->w_planes && !pro_bn && !d->act16) { a.x3 = d->a_planes; a.w3 = d->w_planes; }
    DPFT_REQUIRE(!(a.w16 && (bias || (pro_bn && !pro_relu))), "conv fwd: act16 = 2 (bf16 weights) takes no bias and only the BatchNorm + ReLU prologue");
    const bool pro = pro_bn != nullptr;
    TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, (d->act16 || (pro && !pro_relu)) ? 1 : taps_of(d));
    const bool big16 = !pro && !bias && big16_tile(d, a, false, workspace != nullptr, t);
    if (d->act16 && !big16) t.splits = 1;      // the split-K reduction kernels write fp32 tensors (big16: in-launch fix-up only)
    if (big16 && t.splits > 1 && !sk_fixup_ok(a, t.splits, 1)) t.splits = 1;
    DPFT_REQUIRE(!(pro && !t.vec), "conv fwd: fused prologue needs C %% 64 == 0 (C=%d)", d->C);
    if (pro_sums) {
        // the kernels that build their prologue table from the sums: the split kernels and the pipelined fp32 kernel (launch_igemm)
        static const int pipe_env = getenv("DPFT_PIPE") ? atoi(getenv("DPFT_PIPE")) : 3;
        if (!(t.vec && g_conv_bf16 != 1 && !d->act16 && pro_relu && (t.x3 || (pipe_env & 1)) && (int64_t)12 * d->C <= 16384)) return DPFT_OK;
        a.pro_s = *pro_sums;
        *pro_sums_used = true;
    }
    bool fixup = false;
    if (t.splits > 1) {
        DPFT_REQUIRE(workspace, "conv fwd: split-K selected but no workspace given");
        a.partial = ws_slabs(workspace);
        fixup = sk_fixup_ok(a, t.splits, 1);
        if (fixup) a.sk_ticket = reinterpret_cast<int*>(workspace);      // the last split workgroup of a tile runs the whole epilogue
        else a.stats = nullptr;
    } else if (fuse && fuse->acc && fuse->slab && stats && !bias && !d->act16 && t.vec && (a.N & 3) == 0 &&
               cdiv(a.M, t.bm) <= fuse->slab) {
        // deterministic form: the slab stays, one ticket per column tile (the zeroed accumulator region holds them: 2 K >= tiles)
        a.bnf_slab = 1; a.bnf_ticket = reinterpret_cast<int*>(fuse->acc); a.bnf_gamma = fuse->gamma; a.bnf_beta = fuse->beta;
        a.bnf_rm = fuse->running_mean; a.bnf_rv = fuse->running_var; a.bnf_bnp = fuse->bnp;
        a.bnf_eps = fuse->eps; a.bnf_mom = fuse->momentum;
        fuse->applied = true;
    } else if (fuse && fuse->acc && !fuse->slab && !bias && !d->act16 && t.vec && (a.N & 3) == 0) {
        a.bnf_acc = fuse->acc; a.bnf_ticket = fuse->ticket; a.bnf_gamma = fuse->gamma; a.bnf_beta = fuse->beta;
        a.bnf_rm = fuse->running_mean; a.bnf_rv = fuse->running_var; a.bnf_bnp = fuse->bnp;
        a.bnf_eps = fuse->eps; a.bnf_mom = fuse->momentum;
        a.stats = nullptr;
        fuse->applied = true;
    }
    // (the accumulators' low words hold 65 536 addends: tiles of >= 64 rows)
    if (fuse && fuse->sums && !fuse->applied && stats && !bias && t.vec && (a.N & 3) == 0 && (t.splits == 1 || fixup) &&
        a.M < (1 << 22)) {
        a.bns = fuse->sums;      // column sums instead of the per-tile table; finalized by nobody here
        a.stats = nullptr;
        fuse->applied = true;
    }
    rc = launch_igemm<false>(a, t, pro, st);
    if (rc) return rc;
    if (t.splits > 1 && !fixup) {
        const int64_t MN = (int64_t)a.M * a.N;
        if (stats && !bias && (a.N % 64) == 0 && t.bm <= 128) {      // reduce + per-tile (mean, M2) in one launch
            hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3(cdiv(a.M, t.bm), a.N / 64), dim3(256), 0, st, a.partial, y,
                               stats, (int64_t)a.M, a.N, t.splits, t.bm);
            return check_launch("conv fwd split-K reduce + stats");
        }
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 1024)), dim3(256), 0, st, a.partial, bias, y, MN, a.N, t.splits, 0);
        rc = check_launch("conv fwd split-K reduce");
        if (rc) return rc;
        if (stats) {  // stats from the reduced output (tile_rows = t.bm rows per tile)
            rc = dpft_bn_stats_f32(y, stats, a.M, a.N, t.bm, stream);
        }
    }
    return rc;
}

// Inference form of conv + BatchNorm (+ residual) (+ ReLU): y = [relu](bn(conv(x, w)) [+ residual]) with the BN block of the
// OUTPUT channels.  One launch when the problem takes the vector path without split-K (the epilogue does it); otherwise
// the plain convolution followed by the elementwise pass -- same arithmetic, same result.
extern "C" int dpft_conv2d_nhwc_fwd_bnact_f32(const dpft_conv_desc* d, const float* x, const float* w, const float* out_bn,
                                              int32_t relu, const float* residual, float* y, void* workspace,
                                              dpft_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(x && w && y && out_bn, "conv fwd_bnact: null tensor");
    hipStream_t st = (hipStream_t)stream;
    if (g_conv_bf16 != 1 && stream1x1_match(d, nullptr)) {      // conv_stream.hip
        ProfScope prof(0, d, st);
        g_prof_family = kFamF32;
        return launch_stream1x1(d, x, w, nullptr, y, nullptr, out_bn, residual, relu, st);
    }
    IgemmArgs a; fill_igemm(a, d, false);
    TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, d->act16 ? 1 : taps_of(d));
    if (d->act16) t.splits = 1;
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    static const bool fix_all = getenv("DPFT_BNACT_FIXUP") == nullptr || atoi(getenv("DPFT_BNACT_FIXUP")) != 0;      // A/B switch
    if ((t.x3 || (fix_all && t.vec && !conv16_matches(d))) && t.splits > 1 && (a.N & 3) == 0 && workspace && sk_fixup_ok(a, t.splits, 1)) {
        // a K split (the latency-sized problems: batch-1 inference, the radar encoders' maps; the split kernels): the last
        // workgroup of a tile runs the whole inference epilogue (in-launch fix-up) -- no reduction launch, no elementwise pass
        ProfScope prof(0, d, st);
        a.x = x; a.w = w; a.y = y;
        a.partial = ws_slabs(workspace);
        a.sk_ticket = reinterpret_cast<int*>(workspace);
        a.obn = out_bn; a.oadd = residual; a.orelu = relu;
        return launch_igemm<false>(a, t, false, st);
    }
    if (!t.vec || t.splits > 1 || (a.N & 3) != 0 || conv16_matches(d)) {
        rc = dpft_conv2d_nhwc_fwd_f32(d, x, w, nullptr, nullptr, 0, y, nullptr, workspace, stream);
        if (rc) return rc;
        return bn_act_any(y, out_bn, residual, nullptr, relu, y, nullptr, M, d->K, d->act16 != 0, stream);
    }
    ProfScope prof(0, d, st);
    a.x = x; a.w = w; a.y = y; a.bias = nullptr; a.stats = nullptr; a.pro = nullptr; a.pro_relu = 0;
    a.x16 = a.y16 = d->act16 != 0;
    if (d->a_planes && d->w_planes && !d->act16) { a.x3 = d->a_planes; a.w3 = d->w_planes; }
    a.obn = out_bn; a.oadd = residual; a.orelu = relu;
    // unsplit fp32 launch with a residual: the residual is read during the main loop instead of after it (DPFT_EPF bit 2)
    static const int epf_on = getenv("DPFT_EPF") == nullptr ? 7 : atoi(getenv("DPFT_EPF"));
    if ((epf_on & 4) && residual && !d->act16 && !a.x3 && !t.x3 && (int64_t)a.M * a.N < (1ll << 29) && d->stride == 1 && g_conv_bf16 != 1) a.epf = 3;
    return launch_igemm<false>(a, t, false, st);
}

// One level of the FPN neck's forward as two launches (include/dpft_hip.h): the top-down add rides in the lateral's epilogue on
// the raw-input levels, the positional embedding in the 3x3 conv's; elsewhere conv + the elementwise kernel, same arithmetic.
extern "C" int dpft_fpn_lateral_f32(const dpft_conv_desc* d, const float* x, const float* w, const float* bias, const float* top,
                                    int32_t TH, int32_t TW, float* lat, void* workspace, dpft_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(x && w && lat && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0, "fpn_lateral: a 1x1 conv (stride 1) with its tensors");
    DPFT_REQUIRE(!top || (TH >= 1 && TW >= 1 && TH <= d->H && TW <= d->W && d->K % 4 == 0), "fpn_lateral: bad top level (%d x %d)", TH, TW);
    static const bool fuse = getenv("DPFT_FPN_FUSE") == nullptr || atoi(getenv("DPFT_FPN_FUSE")) != 0;      // A/B switch
    if (top && fuse && conv1x1_to16_top_matches(d)) {
        ProfScope prof(0, d, (hipStream_t)stream);
        g_prof_family = kFamVector;
        return conv1x1_to16_top_forward(d, x, w, bias, top, TH, TW, lat, (hipStream_t)stream);
    }
    rc = dpft_conv2d_nhwc_fwd_f32(d, x, w, bias, nullptr, 0, lat, nullptr, workspace, stream);
    if (rc || !top) return rc;
    return dpft_fpn_topdown_add_f32(lat, top, d->B, d->H, d->W, TH, TW, d->K, stream);
}

extern "C" int dpft_fpn_output_f32(const dpft_conv_desc* d, const float* lat, const float* w, const float* bias, const float* pos_x,
                                   const float* pos_y, float* out, void* workspace, dpft_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(lat && w && out && (pos_x == nullptr) == (pos_y == nullptr), "fpn_output: null tensor / one embedding table without the other");
    DPFT_REQUIRE(d->H == d->OH && d->W == d->OW, "fpn_output: a same-size conv");
    static const bool fuse = getenv("DPFT_FPN_FUSE") == nullptr || atoi(getenv("DPFT_FPN_FUSE")) != 0;
    if (pos_x && fuse && conv16_matches(d) && !d->act16) {
        ProfScope prof(0, d, (hipStream_t)stream);
        g_prof_family = kFamVector;
        return conv16_forward(d, lat, w, bias, out, (hipStream_t)stream, pos_x, pos_y);
    }
    rc = dpft_conv2d_nhwc_fwd_f32(d, lat, w, bias, nullptr, 0, out, nullptr, workspace, stream);
    if (rc || !pos_x) return rc;
    return dpft_add_pos_f32(out, pos_x, pos_y, d->B, d->H, d->W, d->K, stream);
}

extern "C" int dpft_conv2d_nhwc_dgrad_f32(const dpft_conv_desc* d, const float* dy, const float* w_t,
                                          float* dx, int32_t accumulate, void* workspace,
                                          dpft_stream_t stream) {
    return dpft::conv_dgrad_fused(d, dy, w_t, dx, accumulate, workspace, stream, nullptr);
}

static void set_bnr(IgemmArgs& a, const dpft::BnReduceFuse* f) {
    a.bnr_y = f->y; a.bnr_bnp = f->bnp; a.bnr_mask8 = f->mask8; a.bnr_self_mask = f->self_mask; a.bnr_sums = f->sums;
}

// Data gradient with an optional fused BatchNorm-backward reduction (BnReduceFuse, common.h): `fuse->applied` tells the
// caller whether the launch(es) carried it -- split-K and thin-channel paths do not.
int dpft::conv_dgrad_fused(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx, int32_t accumulate,
                           void* workspace, dpft_stream_t stream, BnReduceFuse* fuse) {
    if (fuse) fuse->applied = false;
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(dy && w_t && dx, "conv dgrad: null tensor");
    {
        static const char* skip = getenv("DPFT_SKIP");      // timing experiments (see dpft_conv2d_nhwc_wgrad_f32)
        if (skip && ((strstr(skip, "dgrad3x3") && d->kh == 3) || (strstr(skip, "dgrad1x1") && d->kh == 1)) && (int64_t)d->B * d->H * d->W >= 4096) {
            if (fuse && fuse->sums) fuse->applied = true;
            return DPFT_OK;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(1, d, st);
    if (conv16_matches(d)) return conv16_dgrad(d, dy, w_t, dx, accumulate, st);
    if (d->C <= 4 && d->K % 4 == 0 && (size_t)d->C * d->kh * d->kw * d->K * 4 <= 40960 && ((d->C * d->kh * d->kw * d->K) % 4) == 0) {
        const size_t lds = (size_t)d->C * d->kh * d->kw * d->K * 4;
        const dim3 grid(cdiv((int64_t)d->B * d->H * d->W, 256));
#define THIN_DGRAD(CI)                                                                                                  \
    hipLaunchKernelGGL(thin_dgrad_kernel<CI>, grid, dim3(256), lds, st, dy, w_t, dx, d->B, d->H, d->W, d->OH, d->OW, \
                       d->K, d->kh, d->kw, d->stride, d->pad, accumulate)
        switch (d->C) {
            case 1: THIN_DGRAD(1); break;
            case 2: THIN_DGRAD(2); break;
            case 3: THIN_DGRAD(3); break;
            default: THIN_DGRAD(4); break;
        }
#undef THIN_DGRAD
        return check_launch("conv dgrad (thin input)");
    }
    IgemmArgs a; fill_igemm(a, d, true);
    a.x = dy; a.w = w_t; a.y = dx; a.accumulate = accumulate;
    a.x16 = a.y16 = d->act16 != 0;
    a.w16 = d->act16 == 2;
    if (d->a_planes && d->w_planes && !d->act16) { a.x3 = d->a_planes; a.w3 = d->w_planes; }
    const bool fuse_ok = fuse && fuse->sums && (a.N & 3) == 0;
    // Parity classes pay when each class fills the chip on its own (4x fewer MFMAs); on small maps (radar encoders) the
    // stride^2 classes are stride^2 dependent launches of a few workgroups with the whole tap x channel loop inside
    // (170 us for a 4x16x7 map) -- there one split-K launch over all taps is ~6x faster despite the wasted taps.
    const int64_t class_wgs = (int64_t)cdiv((int64_t)a.B * cdiv(a.OH, d->stride) * cdiv(a.OW, d->stride), 64) * cdiv(a.N, 64);
    if (d->stride > 1 && (a.C % BKV) == 0 && (class_wgs >= kNumCU / 2 || !workspace || d->act16)) {
        // one launch per output-pixel parity class, each over the taps that can reach it (see IgemmArgs::sub_*)
        const int sp = d->stride, cpt = a.C / BKV;
        bool empty_class = false;
        for (int ph = 0; ph < sp; ++ph)
            for (int pw = 0; pw < sp; ++pw) {
                const int r0 = (ph + d->pad) % sp, s0 = (pw + d->pad) % sp;
                if (r0 >= d->kh || s0 >= d->kw) empty_class = true;
            }
        if (empty_class && !accumulate)      // pixels no tap reaches (1x1 stride-2: three of four) are plain zeros
            DPFT_REQUIRE(zero_fill(dx, (d->act16 ? 2 : sizeof(float)) * (size_t)a.B * a.OH * a.OW * a.N, st) == DPFT_OK,
                         "conv dgrad: zero fill failed");
        for (int ph = 0; ph < sp; ++ph)
            for (int pw = 0; pw < sp; ++pw) {
                const int r0 = (ph + d->pad) % sp, s0 = (pw + d->pad) % sp;
                if (r0 >= d->kh || s0 >= d->kw || ph >= a.OH || pw >= a.OW) continue;
                IgemmArgs q = a;
                q.sub_step = sp; q.sub_ph = ph; q.sub_pw = pw;
                q.sub_oh = (a.OH - ph + sp - 1) / sp; q.sub_ow = (a.OW - pw + sp - 1) / sp;
                q.sub_r0 = r0; q.sub_s0 = s0;
                q.sub_nr = (d->kh - r0 + sp - 1) / sp; q.sub_ns = (d->kw - s0 + sp - 1) / sp;
                q.M = a.B * q.sub_oh * q.sub_ow;
                q.ksteps = q.sub_nr * q.sub_ns * cpt;
                TileChoice tq = choose_tile(q.M, q.N, q.C, q.ksteps);
                tq.splits = 1;
                if (fuse_ok && !empty_class) set_bnr(q, fuse);      // classes partition the pixels: every pixel is added once
                rc = launch_igemm<true>(q, tq, false, st);
                if (rc) return rc;
            }
        if (fuse_ok && !empty_class) fuse->applied = true;
        return DPFT_OK;
    }
    TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, (d->act16 || d->stride > 1) ? 1 : taps_of(d));
    const bool big16 = big16_tile(d, a, true, workspace != nullptr, t);
    if (d->act16 && !big16) t.splits = 1;
    if (big16 && t.splits > 1 && !sk_fixup_ok(a, t.splits, 2)) t.splits = 1;
    bool fixup = false;
    if (t.splits > 1) {
        DPFT_REQUIRE(workspace, "conv dgrad: split-K selected but no workspace given");
        a.partial = ws_slabs(workspace);
        fixup = sk_fixup_ok(a, t.splits, 2);
        if (fixup) a.sk_ticket = reinterpret_cast<int*>(workspace);
    }
    if ((t.splits == 1 || fixup) && fuse_ok) {      // (with the fix-up the last split workgroup of a tile runs the unsplit epilogue)
        set_bnr(a, fuse);
        fuse->applied = true;
    }
    // unsplit stride-1 launch that carries the reduction and nothing else in its epilogue: the reduction's operand (and mask
    // byte) is requested with the first tile (EpiPrefetch mode 2; DPFT_EPF bit 1)
    static const int epf_on = getenv("DPFT_EPF") == nullptr ? 3 : atoi(getenv("DPFT_EPF"));
    if ((epf_on & 2) && t.splits == 1 && a.bnr_sums && !accumulate && d->stride == 1 && t.vec && (int64_t)a.M * a.N < (1ll << 29))
        a.epf = 2;
    rc = launch_igemm<true>(a, t, false, st);
    if (rc) return rc;
    if (t.splits > 1 && !fixup) {
        const int64_t MN = (int64_t)a.M * a.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(MN, 1024)), dim3(256), 0, st, a.partial, (const float*)nullptr, dx, MN, a.N, t.splits, accumulate);
        rc = check_launch("conv dgrad split-K reduce");
    }
    return rc;
}

// dx = dgrad(dy) + (res_mask > 0 ? res_src : 0): stride-1 data gradient with the identity-branch ReLU backward folded
// into the epilogue (internal: used by the ResNet launch plan)
int dpft::conv_dgrad_residual(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx, const float* res_src,
                              const float* res_mask, void* workspace, dpft_stream_t stream, BnReduceFuse* fuse,
                              const unsigned char* res_mask8) {
    if (fuse) fuse->applied = false;
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(dy && w_t && dx && res_src && res_mask, "conv dgrad residual: null tensor");
    {
        static const char* skip = getenv("DPFT_SKIP");      // timing experiments (see dpft_conv2d_nhwc_wgrad_f32)
        if (skip && ((strstr(skip, "dgrad3x3") && d->kh == 3) || (strstr(skip, "dgrad1x1") && d->kh == 1)) && (int64_t)d->B * d->H * d->W >= 4096) {
            if (fuse && fuse->sums) fuse->applied = true;
            return DPFT_OK;
        }
    }

    DPFT_REQUIRE(d->stride == 1 && !conv16_matches(d), "conv dgrad residual: stride-1 bottleneck convs only");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(1, d, st);
    IgemmArgs a; fill_igemm(a, d, true);
    a.x = dy; a.w = w_t; a.y = dx; a.res_src = res_src; a.res_mask = res_mask;
    a.x16 = a.y16 = d->act16 != 0;
    a.w16 = d->act16 == 2;
    if (d->a_planes && d->w_planes && !d->act16) { a.x3 = d->a_planes; a.w3 = d->w_planes; }
    if (res_mask8 && (a.N & 3) == 0) a.res_mask8 = res_mask8;      // (the split-K reduction and the scalar tail read res_mask)
    TileChoice t = choose_tile(a.M, a.N, a.C, a.ksteps, d->act16 ? 1 : taps_of(d));
    const bool big16 = big16_tile(d, a, true, workspace != nullptr, t);
    if (d->act16 && !big16) t.splits = 1;
    if (big16 && t.splits > 1 && !sk_fixup_ok(a, t.splits, 4)) t.splits = 1;
    bool fixup = false;
    if (t.splits > 1) {
        DPFT_REQUIRE(workspace, "conv dgrad: split-K selected but no workspace given");
        a.partial = ws_slabs(workspace);
        fixup = sk_fixup_ok(a, t.splits, 4);
        if (fixup) a.sk_ticket = reinterpret_cast<int*>(workspace);
    }
    if ((t.splits == 1 || fixup) && fuse && fuse->sums && (a.N & 3) == 0) {
        set_bnr(a, fuse);
        fuse->applied = true;
    }
    // the full form (identity gradient + next BatchNorm's reduction, both ReLU masks as bytes), unsplit, with enough rows to
    // fill the chip with 128 x 64 tiles: its epilogue operands are fetched during the main loop (EpiPrefetch; the launcher
    // takes the kernel when the operand types allow the pipelined path).  DPFT_EPF=0: the plain epilogue, A/B switch.
    static const int epf_on = getenv("DPFT_EPF") == nullptr ? 3 : atoi(getenv("DPFT_EPF"));      // bit 0: this form, bit 1: mode 2
    if ((epf_on & 1) && t.splits == 1 && a.bnr_sums && a.bnr_mask8 && a.res_mask8 && t.vec && (int64_t)a.M * a.N < (1ll << 29) &&
        (int64_t)cdiv(a.M, 128) * cdiv(a.N, 64) >= kNumCU && getenv("DPFT_FORCE_TILE") == nullptr && !big16) {
        a.epf = 1;
        t.bm = 128; t.bn = 64;
    }
    rc = launch_igemm<true>(a, t, false, st);
    if (rc) return rc;
    if (t.splits > 1 && !fixup) {
        const int64_t MN = (int64_t)a.M * a.N;
        hipLaunchKernelGGL(splitk_reduce_residual_kernel, dim3(cdiv(MN, 256)), dim3(256), 0, st, a.partial, res_src,
                           res_mask, dx, MN, t.splits);
        rc = check_launch("conv dgrad split-K residual reduce");
    }
    return rc;
}

// C-ABI face of the launch plan's fused data gradients (kernel-level parity: tests/test_gpu_conv_table.py)
extern "C" int dpft_conv2d_nhwc_dgrad_bn_reduce_f32(const dpft_conv_desc* d, const float* dy, const float* w_t, float* dx,
                                                    int32_t accumulate, const float* res_src, const float* res_out,
                                                    const uint8_t* res_mask8, const float* bn_y, const float* bn_block,
                                                    const uint8_t* bn_mask8, int32_t bn_self_mask, float* sums,
                                                    int32_t* applied, void* workspace, dpft_stream_t stream) {
    DPFT_REQUIRE(bn_y && bn_block && sums && applied, "conv dgrad_bn_reduce: null BatchNorm argument");
    DPFT_REQUIRE((bn_mask8 != nullptr) != (bn_self_mask != 0), "conv dgrad_bn_reduce: exactly one of bn_mask8 / bn_self_mask");
    dpft::BnReduceFuse f{bn_y, bn_block, bn_mask8, bn_self_mask, sums, false};
    int rc;
    if (res_src) {
        DPFT_REQUIRE(!accumulate && res_out, "conv dgrad_bn_reduce: the residual form needs res_out and does not accumulate");
        rc = dpft::conv_dgrad_residual(d, dy, w_t, dx, res_src, res_out, workspace, stream, &f, res_mask8);
    } else {
        rc = dpft::conv_dgrad_fused(d, dy, w_t, dx, accumulate, workspace, stream, &f);
    }
    *applied = f.applied ? 1 : 0;
    return rc;
}

extern "C" int dpft_conv2d_nhwc_wgrad_f32(const dpft_conv_desc* d, const float* x, const float* dy,
                                          const float* pro_bn, int32_t pro_relu, float* dw,
                                          void* workspace, dpft_stream_t stream) {
    {      // timing experiments (wrong gradients, DPFT_SKIP=...): the step time without a family = its cost on the critical path
        static const char* skip = getenv("DPFT_SKIP");
        if (skip && strstr(skip, "wgrad")) return DPFT_OK;
    }
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(x && dy && dw, "conv wgrad: null tensor");
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof(2, d, st);
    if (!pro_bn) {
        bool handled = false;
        rc = thin_wgrad(d, x, dy, dw, workspace, st, handled);
        if (rc || handled) return rc;
    }
    WgradArgs a; memset(&a, 0, sizeof(a));
    a.x = x; a.dy = dy; a.dw = dw; a.pro = pro_bn; a.pro_relu = pro_relu;
    a.x16 = a.dy16 = d->act16 != 0;
    a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->C; a.OH = d->OH; a.OW = d->OW; a.K = d->K;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->OH * d->OW;
    a.taps = d->kh * d->kw;
    a.J = a.taps * a.C;
    const bool pro = pro_bn != nullptr;
    const bool vec = (d->C % 32 == 0) && (d->K % 4 == 0);
    a.psteps = cdiv(a.M, vec ? BKP : BK);
    DPFT_REQUIRE(!(pro && !vec), "conv wgrad: fused prologue needs C %% 32 == 0");
    DPFT_REQUIRE(!vec || ((int64_t)a.M * a.K < (1ll << 29) && (int64_t)a.B * a.H * a.W * a.C < (1ll << 29)),
                 "conv wgrad: operand larger than 2 GiB");
    int bmn, bnc;
    int64_t tiles;
    if (vec) {
        if (d->K <= 32) { bmn = 32; bnc = 128; }
        else if (d->K >= 128 && d->C >= 128 && (int64_t)cdiv(d->K, 128) * cdiv(d->C, 128) * a.taps >= 8) { bmn = 128; bnc = 128; }
        else { bmn = 64; bnc = 64; }      // also when 128x128 would leave < 8 output tiles to split over
        a.ktiles = cdiv(d->K, bmn);
        a.ctiles = cdiv(d->C, bnc);
        tiles = (int64_t)a.ktiles * a.ctiles * a.taps;
    } else {
        bmn = 64; bnc = 64;
        a.ktiles = cdiv(d->K, 64);
        a.ctiles = cdiv(a.J, 64);
        tiles = (int64_t)a.ktiles * a.ctiles;
    }
    // Pixel-splits: the grid should be ONE full round of workgroups -- 256 for the 128x128 tile, 512 (two per CU) for
    // the smaller ones -- and never spill a few workgroups into another round (tools/wgrad_sweep.sh: 36 tiles x 15
    // splits = 540 workgroups ran 25 % slower than 36 x 14 = 504 or 36 x 7 = 252).  Partial slabs are bounded by
    // 64 MiB (and by the workspace contract).
    const int round_wgs = (vec && bmn == 128) ? kNumCU : kNumCU * 2;
    int splits = (int)std::max<int64_t>(1, round_wgs / tiles);
    {   // when one round would stay badly filled (e.g. 144 tiles), fill two rounds instead
        const int s2 = (int)std::max<int64_t>(1, 2 * round_wgs / tiles);
        const double f1 = (double)tiles * splits / round_wgs, f2 = (double)tiles * s2 / (2.0 * round_wgs);
        if (f1 < 0.8 && f2 > f1 + 0.1) splits = s2;
    }
    const int64_t wbytes = (int64_t)d->K * a.J * 4;
    const int max_splits = (int)std::max<int64_t>(1, std::min<int64_t>(512, (64ll << 20) / wbytes));
    if (splits > max_splits) splits = max_splits;
    if (splits > 1 && a.psteps / splits < 4) {
        // Few pixels (radar encoders: 84 ... 1800): a workgroup would get under four pixel steps, so its fixed costs and
        // the partial slabs decide.  tools/wgrad_small_sweep.sh: the 64x64 tile with about one workgroup per CU wins
        // by 2-4x over fewer, fatter workgroups (448 pixels, 256->1024: 11.9 us vs 48.3 us unsplit 128x128 tiles).
        if (vec && bmn == 128) {
            bmn = 64; bnc = 64;
            a.ktiles = cdiv(d->K, bmn); a.ctiles = cdiv(d->C, bnc);
            tiles = (int64_t)a.ktiles * a.ctiles * a.taps;
        }
        splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(kNumCU / tiles, a.psteps), max_splits));
    }
    if (splits < 1) splits = 1;
    if (const char* f = getenv("DPFT_FORCE_WGRAD")) {      // tuning aid: "tile,splits" (tile 128 | 64 for the vec kernel)
        int tb, sp;
        if (vec && sscanf(f, "%d,%d", &tb, &sp) == 2 && (tb == 128 || tb == 64)) {
            bmn = tb; bnc = tb;
            a.ktiles = cdiv(d->K, bmn); a.ctiles = cdiv(d->C, bnc);
            tiles = (int64_t)a.ktiles * a.ctiles * a.taps;
            splits = std::max(1, std::min(sp, max_splits));
        }
    }
    if (splits > 1 && !workspace) splits = 1;
    a.splits = splits;
    a.psteps_per_split = cdiv(a.psteps, splits);
    a.partial = splits > 1 ? ws_slabs(workspace) : nullptr;
    const int nwg = (int)(tiles * splits);
    dim3 grid(nwg), block(256);
    static const int wpipe_env = getenv("DPFT_PIPE") ? atoi(getenv("DPFT_PIPE")) : 3;
    static const int w16_env = getenv("DPFT_WGRAD16_PIPE") ? atoi(getenv("DPFT_WGRAD16_PIPE")) : 1;      // A/B switch
    if (vec && d->act16 && !pro && w16_env && (bmn == 128 || bmn == 64) && (d->C % 8) == 0 && (d->K % 8) == 0) {
        // both operands bf16 in memory, no prologue: LDS-DMA + transpose reads + bf16 MFMA (wgrad_pipe16_kernel)
        const int pk = bmn == 128 ? 64 : 128;
        a.psteps = cdiv(a.M, pk);
        a.psteps_per_split = cdiv(a.psteps, splits);
        const size_t lds = (size_t)2 * pk * (bmn + bnc) * 2 + (size_t)a.psteps_per_split * pk * 4;
        g_prof_family = kFamBf16;
        if (bmn == 128) launch_lds(wgrad_pipe16_kernel<128, 128, 2, 2, 64>, grid, block, lds, st, a);
        else launch_lds(wgrad_pipe16_kernel<64, 64, 2, 2, 128>, grid, block, lds, st, a);
    } else if (vec && (wpipe_env & 2) && g_conv_bf16 != 1 && !d->act16 && (!pro || pro_relu) && (bmn == 128 || bmn == 64)) {
        // software-pipelined form (conv_pipe.h): psteps in units of its PK pixels
        const int pk = bmn == 128 ? 32 : 64;
        a.psteps = cdiv(a.M, pk);
        a.psteps_per_split = cdiv(a.psteps, splits);
        const size_t tbl = (size_t)a.psteps_per_split * pk * 4;      // per-workgroup input-pixel offset table
        g_prof_family = kFamF32;
        static const bool wx3 = getenv("DPFT_WGRAD_X3") == nullptr || atoi(getenv("DPFT_WGRAD_X3")) != 0;      // A/B switch
        static const bool wx3_1x1 = getenv("DPFT_WGRAD_X3_1X1") == nullptr || atoi(getenv("DPFT_WGRAD_X3_1X1")) != 0;      // A/B switch
        if (bmn == 128 && wx3 && split_on() && (a.taps > 1 || wx3_1x1) && 2.0 * a.M * (double)d->K * a.J >= 2e9) {
            // the big multi-tap weight gradients on the split kernels (conv_x3.hip), like their forward / data gradient
            g_prof_family = kFamX3;
            rc = launch_wgrad_x3(a, pro, grid, st);
            if (rc) return rc;
        } else if (bmn == 128) {
            const size_t lds = (size_t)2 * 32 * (128 + 128) * 4 + tbl;
            if (pro) launch_lds(wgrad_pipe_kernel<128, 128, 2, 2, 32, true>, grid, block, lds, st, a);
            else launch_lds(wgrad_pipe_kernel<128, 128, 2, 2, 32, false>, grid, block, lds, st, a);
        } else {
            const size_t lds = (size_t)2 * 64 * (64 + 64) * 4 + tbl;
            if (pro) launch_lds(wgrad_pipe_kernel<64, 64, 2, 2, 64, true>, grid, block, lds, st, a);
            else launch_lds(wgrad_pipe_kernel<64, 64, 2, 2, 64, false>, grid, block, lds, st, a);
        }
    } else if (vec) {
        g_prof_family = g_conv_bf16 == 1 && (bmn == 128 || bmn == 64) ? kFamBf16 : kFamF32;
#define LAUNCH_WG(BM_, BN_, WGM_, WGN_)                                                           \
    do {                                                                                          \
        if (pro) hipLaunchKernelGGL((wgrad_vec_kernel<BM_, BN_, WGM_, WGN_, true>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((wgrad_vec_kernel<BM_, BN_, WGM_, WGN_, false>), grid, block, 0, st, a);     \
    } while (0)
        if (bmn == 128 && g_conv_bf16 == 1) {      // mixed-precision mode (dpft_conv_set_compute): bf16 operands
            if (pro) hipLaunchKernelGGL((wgrad_vec_kernel<128, 128, 2, 2, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((wgrad_vec_kernel<128, 128, 2, 2, false, true>), grid, block, 0, st, a);
        } else if (bmn == 64 && g_conv_bf16 == 1) {
            if (pro) hipLaunchKernelGGL((wgrad_vec_kernel<64, 64, 2, 2, true, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((wgrad_vec_kernel<64, 64, 2, 2, false, true>), grid, block, 0, st, a);
        } else if (bmn == 128) LAUNCH_WG(128, 128, 2, 2);
        else if (bmn == 64) LAUNCH_WG(64, 64, 2, 2);
        else LAUNCH_WG(32, 128, 1, 4);
#undef LAUNCH_WG
    } else {
        hipLaunchKernelGGL(wgrad_gen_kernel, grid, block, 0, st, a);
    }
    rc = check_launch("conv wgrad");
    if (rc) return rc;
    {
        static const char* skip = getenv("DPFT_SKIP");      // timing experiment: the reduction launches' cost on the critical path
        if (skip && strstr(skip, "wreduce")) return rc;
    }
    if (splits > 1) {
        const int64_t n = (int64_t)d->K * a.J;
        if ((n & 3) == 0 && splits >= 16) {
            if (n / 4 <= (int64_t)kNumCU * 64) hipLaunchKernelGGL(splitk_reduce_wide_kernel<16>, dim3(cdiv(n / 4, 16)), dim3(256), 0, st, a.partial, dw, n, splits);
            else hipLaunchKernelGGL(splitk_reduce_wide_kernel<4>, dim3(cdiv(n / 4, 64)), dim3(256), 0, st, a.partial, dw, n, splits);
        } else {
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n, 1024)), dim3(256), 0, st, a.partial, (const float*)nullptr, dw, n, 4, splits, 0);
        }
        rc = check_launch("conv wgrad split-K reduce");
    }
    return rc;
}

extern "C" int dpft_weight_transpose_f32(const float* w, float* w_t, int32_t K, int32_t taps,
                                         int32_t C, dpft_stream_t stream) {
    DPFT_REQUIRE(w && w_t && K > 0 && taps > 0 && C > 0, "weight_transpose: bad arguments");
    dim3 grid(cdiv(C, 32), cdiv(K, 32), taps);
    hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, w_t, K, taps, C);
    return check_launch("weight_transpose");
}

extern "C" int dpft_weight_transpose_batch_f32(int32_t n, const float* const* w, float* const* w_t, const int32_t* K,
                                               const int32_t* taps, const int32_t* C, dpft_stream_t stream) {
    DPFT_REQUIRE(n >= 1 && n <= dpft::TransposeBatch::MAX && w && w_t && K && taps && C, "weight_transpose_batch: 1..80 tensors");
    dpft::TransposeBatch tb;
    tb.n = 0;
    tb.mode = 0;
    for (int i = 0; i < n; ++i) {
        DPFT_REQUIRE(w[i] && w_t[i] && K[i] > 0 && taps[i] > 0 && C[i] > 0, "weight_transpose_batch: bad entry %d", i);
        if (!tb.fits(K[i], C[i])) {      // tile edge of a launch (64 / 32): a second launch for the other kind
            int rc = dpft::weight_transpose_batch(tb, stream);
            if (rc) return rc;
            tb.n = 0;
        }
        tb.add(w[i], w_t[i], K[i], taps[i], C[i]);
    }
    return dpft::weight_transpose_batch(tb, stream);
}

int dpft::weight_transpose_batch(const TransposeBatch& tb, dpft_stream_t stream) {
    DPFT_REQUIRE(tb.n >= 1 && tb.n <= TransposeBatch::MAX, "weight_transpose_batch: 1..%d tensors", TransposeBatch::MAX);
    if (tb.tile == 64) hipLaunchKernelGGL(weight_transpose_batch64_kernel, dim3(tb.blk_start[tb.n]), dim3(256), 0, (hipStream_t)stream, tb);
    else hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3(tb.blk_start[tb.n]), dim3(256), 0, (hipStream_t)stream, tb);
    return check_launch("weight_transpose_batch");
}

namespace dpft {
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned* __restrict__ p, size_t n4, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) reinterpret_cast<uint4*>(p)[i] = uint4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) p[i] = 0u;
}
}  // namespace dpft

int dpft::zero_fill(void* ptr, size_t bytes, dpft_stream_t stream) {
    DPFT_REQUIRE(ptr && (bytes & 3) == 0 && ((uintptr_t)ptr & 15) == 0, "zero_fill: 16-byte aligned pointer, size a multiple of 4");
    const size_t n = bytes / 4, n4 = n / 4;
    const int blocks = (int)std::max<size_t>(1, std::min<size_t>(kNumCU * 8, (n4 + 255) / 256));
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)ptr, n4, n);
    return check_launch("zero_fill");
}

extern "C" int dpft_bias_grad_f32(const float* dy, float* db, int64_t M, int32_t K,
                                  dpft_stream_t stream) {
    DPFT_REQUIRE(dy && db && M > 0 && K > 0 && K <= 256, "bias_grad: bad arguments (K=%d)", K);
    (void)hipMemsetAsync(db, 0, sizeof(float) * K, (hipStream_t)stream);
    const int rpi = 256 / K;
    int blocks = (int)std::min<int64_t>(1024, (M + rpi - 1) / rpi);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, db, M, K);
    return check_launch("bias_grad");
}

// bias gradient behind a weight gradient on the same stream: the conv workspace is free again (ticket header zero, slabs
// consumed) -- the slab form above; without a workspace the cleared-output + atomics form
int dpft::bias_grad_ws(const float* dy, float* db, int64_t M, int32_t K, void* workspace, dpft_stream_t stream) {
    static const bool slab_on = getenv("DPFT_BIAS_SLAB") == nullptr || atoi(getenv("DPFT_BIAS_SLAB")) != 0;      // A/B switch
    if (!workspace || !slab_on || K > 256 || K <= 0) return dpft_bias_grad_f32(dy, db, M, K, stream);
    const int rpi = 256 / K;
    const int blocks = (int)std::min<int64_t>(kBiasBlocks, (M + rpi - 1) / rpi);
    hipLaunchKernelGGL(bias_grad_slab_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, db, M, (int)K, ws_slabs(workspace),
                       reinterpret_cast<int*>(workspace));
    return check_launch("bias_grad (slab)");
}

// Weight gradient AND bias gradient of a conv with bias (the FPN's convs): one pass over dy where the weight-gradient kernel
// has dy at hand (3x3 16 -> 16, thin 1x1 -> 16), the two separate launches otherwise.  Same results either way.
extern "C" int dpft_conv2d_nhwc_wgrad_bias_f32(const dpft_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                                               void* workspace, dpft_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    DPFT_REQUIRE(x && dy && dw && db, "conv wgrad_bias: null tensor");
    static const bool fuse = getenv("DPFT_BIAS_FUSE") == nullptr || atoi(getenv("DPFT_BIAS_FUSE")) != 0;      // A/B switch
    const bool fusable = dpft::conv16_matches(d) ||
                         (d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->K == 16 && (d->C == 3 || d->C == 6));
    if (fuse && fusable && workspace && !d->act16) {
        dpft::ProfScope prof(2, d, (hipStream_t)stream);
        bool handled = false, bias_done = false;
        rc = dpft::thin_wgrad(d, x, dy, dw, workspace, (hipStream_t)stream, handled, db, &bias_done);
        if (rc) return rc;
        if (handled && bias_done) return DPFT_OK;
        if (handled) return dpft::bias_grad_ws(dy, db, (int64_t)d->B * d->OH * d->OW, d->K, workspace, stream);
    }
    rc = dpft_conv2d_nhwc_wgrad_f32(d, x, dy, nullptr, 0, dw, workspace, stream);
    if (rc) return rc;
    return dpft::bias_grad_ws(dy, db, (int64_t)d->B * d->OH * d->OW, d->K, workspace, stream);
}

static float g_prof_overhead_ms = 0.f;      // elapsed time of an EMPTY event bracket (subtracted from every record)

extern "C" int dpft_profile_start(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    // calibrate the bracket itself: two back-to-back event records cost a few us of queue time that is not the
    // kernel's (rocprofv3's per-kernel durations do not contain it)
    {
        hipStream_t st;
        if (hipStreamCreate(&st) == hipSuccess) {
            float samples[15];
            int n = 0;
            for (int i = 0; i < 15; ++i) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) break;
                (void)hipEventRecord(a, st);
                (void)hipEventRecord(b, st);
                float ms = 0.f;
                if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess) samples[n++] = ms;
                (void)hipEventDestroy(a);
                (void)hipEventDestroy(b);
            }
            (void)hipStreamDestroy(st);
            if (n > 0) {
                std::sort(samples, samples + n);
                g_prof_overhead_ms = samples[n / 2];
            }
        }
    }
    g_prof_on = true;
    return DPFT_OK;
}

extern "C" float dpft_profile_overhead_ms(void) { return g_prof_overhead_ms; }

extern "C" int dpft_profile_serialize(int32_t on) {
    g_serialize = on != 0;
    return DPFT_OK;
}

extern "C" int32_t dpft_profile_stop(void) {
    g_prof_on = false;
    return (int32_t)g_prof.size();
}

extern "C" int dpft_profile_get_family(int32_t i, int32_t* family) {
    DPFT_REQUIRE(i >= 0 && i < (int32_t)g_prof.size() && family, "profile_get_family: bad index");
    *family = g_prof[i].family;
    return DPFT_OK;
}

extern "C" int dpft_profile_get(int32_t i, int32_t* kind, double* flops, float* ms, int32_t* shape7) {
    DPFT_REQUIRE(i >= 0 && i < (int32_t)g_prof.size() && kind && flops && ms && shape7, "profile_get: bad index");
    const ProfRec& r = g_prof[i];
    *kind = r.kind;
    *flops = r.flops;
    if (hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess) {
        set_error("profile_get: events not complete (synchronise the stream first)");
        return DPFT_ERR_LAUNCH;
    }
    *ms = fmaxf(*ms - g_prof_overhead_ms, 0.f);
    memcpy(shape7, r.shape, sizeof(r.shape));
    return DPFT_OK;
}
