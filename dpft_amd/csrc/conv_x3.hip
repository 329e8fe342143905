// fp32 implicit-GEMM convolutions on the bf16 matrix cores: the 3 x bf16 split main loop (round 5).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the vector fp32 rate (157 TF nominal, ~142 TF at the clock the chip holds under
// that load) and the forward / data-gradient GEMMs of the step sit at 100-110 TF on their best shapes: the fp32 matrix
// path has no headroom left.  v_mfma_f32_32x32x16_bf16 is 16x faster per reduction index.  An fp32 value is EXACTLY the
// sum of three bf16 terms a = a1 + a2 + a3 (each the round-to-nearest-even bf16 of what the previous ones left: 3 x 8 =
// 24 significand bits), every bf16 x bf16 product is exact in the MFMA's fp32 datapath, and of the nine term products
// only the six of weight >= 2^-16 matter at fp32 accuracy:
//     a b  ~=  a1 b1  +  (a1 b2 + a2 b1)  +  (a1 b3 + a2 b2 + a3 b1)          (dropped: <= 3 * 2^-24 |a b|)
// Six bf16 MFMAs (6 x 32 cycles) replace eight fp32 MFMAs (8 x 64 cycles) per 32x32 block and 16 reduction indices, and
// -- unlike the fp32 MFMA -- they leave the vector ALUs to the wave.  The leading term accumulates in its own register
// set (one rounding per 16 indices instead of one per 2), the five small terms in a second one: measured error against
// fp64 is BELOW the fp32-MFMA path's on every shape of the step (tests/test_gpu_conv_table.py).
//
// What the loop looks like (tools/probes/x3_mix_probe.hip priced the pieces: a bare MFMA stream runs at ~40 nominal
// cycles per MFMA, every vector instruction beside it costs ~2 more):
//   * tensors stay fp32 in HBM; both operands take the register route: a lane loads 4 consecutive reduction indices of a
//     row (16 bytes, raw buffer load, hardware range check = padding / row tails), applies the producer's BatchNorm +
//     ReLU where the conv has that prologue, splits (22 vector instructions per quad) and writes three 8-byte pieces,
//     one per bf16 plane, into the LDS stage of the NEXT K-step;
//   * K-step = 32 reduction indices, two LDS stages, ONE barrier per step; the quads of a tile are consumed one at a
//     time behind every few MFMAs of the current step, and the load of the same quad of the tile AFTER the next is
//     issued right behind its consumption -- every load has a whole K-step to land, no second register set;
//   * LDS: plane rows of 64 bytes (32 bf16), the four 16-byte chunks of a row XOR-swizzled by (row / 4) % 4 so that the
//     16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS table) see 64 distinct banks; fragment addresses are one
//     VGPR per (stage, K-group, operand) + immediates;
//   * the BatchNorm block of the prologue sits in LDS behind the stages (3 x C floats), read once per step.
// Epilogue, split-K fix-up, statistics, fused BatchNorm-backward reduction and the epilogue-operand prefetch are the shared
// ones of conv_core.h: a launch is interchangeable with igemm_pipe_kernel's.
#include "conv_core.h"

namespace dpft {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// (x, y) -> packed bf16 pair (RNE), and the two values it represents
__device__ __forceinline__ unsigned cvt_pk_bf16(float x, float y) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x, y}, bf16x2_t));
}
// 4 fp32 -> three planes of 4 bf16 (8 bytes each): 22 vector instructions
__device__ __forceinline__ void split3(const f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float x = v[2 * h], y = v[2 * h + 1];
        const unsigned q1 = cvt_pk_bf16(x, y);
        const float rx = x - __uint_as_float(q1 << 16), ry = y - __uint_as_float(q1 & 0xffff0000u);
        const unsigned q2 = cvt_pk_bf16(rx, ry);
        const float sx = rx - __uint_as_float(q2 << 16), sy = ry - __uint_as_float(q2 & 0xffff0000u);
        p1[h] = q1;
        p2[h] = q2;
        p3[h] = cvt_pk_bf16(sx, sy);
    }
}

// PRO_: 0 no prologue, 1 BatchNorm + ReLU of the A operand, 2 the same on a conv with padding (BN(0) != 0: elements of taps
// that miss the image are forced back to zero -- five vector instructions per quad, so only where padding exists)
template <int BM, int BN, bool DGRAD, int PRO_, int EPF = 0>
__global__ __launch_bounds__(256) void igemm_x3_kernel(IgemmArgs a) {
    constexpr bool PRO = PRO_ != 0, MASK = PRO_ == 2;
    static_assert(EPF == 0 || (DGRAD && !PRO), "epilogue prefetch: data gradients only");
    constexpr int WGM = 2, WGN = 2, PBK = 32;
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;
    constexpr int RPP = 32;                       // rows per loader pass: 8 lanes x 16 bytes per fp32 row segment, 8 rows per wave
    constexpr int AP = BM / RPP, BP = BN / RPP, NQ = AP + BP;
    constexpr int ROWB = PBK * 2;                 // bytes per plane row
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE, STAGE = A_BYTES + B_BYTES;
    constexpr int NG = PBK / 16;                  // K-groups (one v_mfma_f32_32x32x16_bf16 deep) per step
    static_assert(RB >= 1 && CB >= 1 && AP >= 1 && BP >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages | BatchNorm block [3][C] (PRO) ; epilogue staging
    char* const lds = reinterpret_cast<char*>(smem);

    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader geometry ------------------------------------------------------------------------------------------
    const int rp = 8 * wave + (lane >> 3);        // row within a pass
    const int ch = lane & 7;                      // 16-byte fp32 chunk of the row segment: reduction indices 4 ch .. 4 ch + 3
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
    // position of the next tile to be LOADED (tap, channel offset) and its scalar load operands
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
    // channel offset of the tile being CONSUMED (prologue parameters)
    int cons_c0 = (kt_begin - (kt_begin / cpt) * cpt) * PBK;

    // ---- BatchNorm block of the prologue -> LDS [3][C] behind the stages ---------------------------------------------
    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)smem;
    if constexpr (PRO) {
        float* tab = reinterpret_cast<float*>(lds + 2 * STAGE);
        fill_pro_table(tab, a.pro, a.pro_s, a.C, tid, 256);      // rows mean, scale, beta (from the BN block or the layer's column sums)
    }
    const unsigned ptab_ad = lds_base + 2 * STAGE + ch * 16;

    // ---- register route --------------------------------------------------------------------------------------------
    // Two register sets: set s holds the tiles of parity s -- tile t + 1 is consumed during step t, and the load of tile t + 3
    // goes into the registers it frees: TWO K-steps of flight time (tools/x3_abl.sh: with one step the loop waited for its
    // loads -- an L2 round trip under this load is ~1 us, a K-step of MFMAs ~0.9 us)
    f32x4 rq[2][NQ];                               // quads 0 .. AP-1: A rows, AP .. NQ-1: weight rows
    unsigned ra_valid[2] = {0, 0};                 // bit k: quad k of the set is a real pixel (MASK)
    f32x4 p_mu, p_sc, p_sh;
    auto load_quad = [&](auto SET, auto K) {
        constexpr int set = decltype(SET)::value, k = decltype(K)::value;
        if constexpr (k < AP) {
            rq[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_off[k], so_a, 0));
            if constexpr (MASK) ra_valid[set] = (ra_valid[set] & ~(1u << k)) | (a_valid_tap & (1u << k));      // v_bfi_b32
        } else {
            rq[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off[k - AP], so_b, 0));
        }
    };
    // LDS destination of this lane's piece of a quad: row rp (+ 32 per pass), 8 bytes at half (ch & 1) of chunk (ch / 2) ^ key
    const int w_off = rp * ROWB + ((((ch >> 1) ^ ((rp >> 2) & 3)) << 4) | ((ch & 1) << 3));
    auto read_pro = [&]() {      // the consumed tile's BatchNorm parameters for this lane's 4 channels
        if constexpr (PRO) {
            const unsigned ad = ptab_ad + (unsigned)cons_c0 * 4u;
            p_mu = *(lds_f32x4*)(size_t)(ad);
            p_sc = *(lds_f32x4*)(size_t)(ad + (unsigned)a.C * 4u);
            p_sh = *(lds_f32x4*)(size_t)(ad + (unsigned)a.C * 8u);
        }
    };
    // A quad's way into LDS as separately PLACEABLE pieces (the step pins one or two of them behind each MFMA: left to the
    // compiler, a quad's ~40 instructions come as one clump between two MFMA runs and the matrix pipe idles meanwhile;
    // sched_group_barrier pipelines of this size defeat the solver).  One quad is in flight at a time (cv, c1..c3).
    //   A quad with prologue: 0 BN+ReLU e0,e1 | 1 BN+ReLU e2,e3 | 2 padding mask | 3..6 split | 7 LDS writes
    //   other quads: pieces 3..7 only
    f32x4 cv;
    u32x2 c1, c2, c3;
    float rr0 = 0.f, rr1 = 0.f;
    auto piece = [&](auto STG, auto K, auto ID) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value, id = decltype(ID)::value;
        constexpr bool bn = PRO && k < AP;
        if constexpr (id == 0 || id == 1) {
            static_assert(bn, "BatchNorm pieces: A quads of a PRO kernel");
            if constexpr (id == 0) cv = rq[stg][k];
#pragma unroll
            for (int e = 2 * id; e < 2 * id + 2; ++e) cv[e] = fmaxf(fmaf(cv[e] - p_mu[e], p_sc[e], p_sh[e]), 0.f);
        } else if constexpr (id == 2) {      // padding stays exactly zero (BN(0) != 0): AND with 0 / ~0 from the quad's validity bit
            const unsigned keep = (unsigned)__builtin_amdgcn_sbfe(ra_valid[stg], k, 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) cv[e] = __uint_as_float(__float_as_uint(cv[e]) & keep);
        } else if constexpr (id == 3 || id == 5) {      // leading term of a half, residuals
            constexpr int h = (id - 3) / 2;
            if constexpr (id == 3 && !bn) cv = rq[stg][k];
            const float x = cv[2 * h], y = cv[2 * h + 1];
            const unsigned q = cvt_pk_bf16(x, y);
            c1[h] = q;
            rr0 = x - __uint_as_float(q << 16);
            rr1 = y - __uint_as_float(q & 0xffff0000u);
        } else if constexpr (id == 4 || id == 6) {      // second and third term
            constexpr int h = (id - 4) / 2;
            const unsigned q = cvt_pk_bf16(rr0, rr1);
            c2[h] = q;
            c3[h] = cvt_pk_bf16(rr0 - __uint_as_float(q << 16), rr1 - __uint_as_float(q & 0xffff0000u));
        } else {
            constexpr int PL = k < AP ? A_PLANE : B_PLANE;
            char* dst = lds + stg * STAGE + (k < AP ? k * RPP * ROWB : A_BYTES + (k - AP) * RPP * ROWB) + w_off;
            *reinterpret_cast<u32x2*>(dst) = c1;
            *reinterpret_cast<u32x2*>(dst + PL) = c2;
            *reinterpret_cast<u32x2*>(dst + 2 * PL) = c3;
        }
    };
    constexpr int NPA = PRO ? (MASK ? 8 : 7) : 5, NPB = 5, NPIECE = AP * NPA + BP * NPB;
    // piece number P of a tile (quads in order) -> (quad, id)
    auto piece_at = [&](auto STG, auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        if constexpr (P < AP * NPA) {
            constexpr int k = P / NPA, o = P % NPA;
            constexpr int id = !PRO ? o + 3 : (MASK ? o : (o < 2 ? o : o + 1));
            piece(STG, std::integral_constant<int, k>{}, std::integral_constant<int, id>{});
        } else {
            constexpr int k = AP + (P - AP * NPA) / NPB, id = 3 + (P - AP * NPA) % NPB;
            piece(STG, std::integral_constant<int, k>{}, std::integral_constant<int, id>{});
        }
    };
    // quad whose LAST piece is piece P (its registers are free for the next load), or -1
    auto quad_done_at = [](int P) constexpr {
        if (P < AP * NPA) return P % NPA == NPA - 1 ? P / NPA : -1;
        return (P - AP * NPA) % NPB == NPB - 1 ? AP + (P - AP * NPA) / NPB : -1;
    };

    f32x16 acc[RB][CB], acl[RB][CB];      // leading term | the five small terms
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }

    // ---- epilogue-operand prefetch (as igemm_pipe_kernel) -------------------------------------------------------------
    constexpr int EP_ITER = BM * (BN / 4) / 256;
    using PFT = EpiPrefetch<EP_ITER, false, EPF>;
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
            const unsigned el = ok ? (unsigned)(m0 + row) * (unsigned)a.N + (unsigned)(n0 + c4 * 4)
                                   : (unsigned)m0 * (unsigned)a.N + (unsigned)n0;
            using Q = typename PFT::Q;
            if constexpr (kind == 0) pf.y[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.bnr_y) + (size_t)el * 4);
            else if constexpr (kind == 1) { if constexpr (EPF == 1) pf.g[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.res_src) + (size_t)el * 4); }
            else if constexpr (kind == 2) { if (EPF == 1 || a.bnr_mask8) pf.mk[it] = a.bnr_mask8[el >> 2]; }
            else { if constexpr (EPF == 1) pf.rm[it] = a.res_mask8[el >> 2]; }
        }
    };

    // ---- fragment addresses: lane l reads row l % 32 of a 32-row block, chunk 2 g + l / 32 of K-group g --------------
    const int fkey = ((lane & 31) >> 2) & 3, hh = lane >> 5;
    unsigned a_ad[2][NG], b_ad[2][NG];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sw = ((2 * g + hh) ^ fkey) << 4;
            a_ad[s][g] = lds_base + s * STAGE + (wm * RB * 32 + (lane & 31)) * ROWB + sw;
            b_ad[s][g] = lds_base + s * STAGE + A_BYTES + (wn * CB * 32 + (lane & 31)) * ROWB + sw;
            asm volatile("" : "+v"(a_ad[s][g]), "+v"(b_ad[s][g]));
        }

    // One K-step on stage STG.  M1: tile t + 1 exists (its quads are in register set STG ^ 1 and are consumed into the other
    // stage); M2: tile t + 3 exists (its quads are loaded into that set behind the consumption).  MFMA order inside a K-group: term-major
    // over the wave's blocks; quads are spread evenly over the step's MFMAs.
    constexpr int MPG = 6 * RB * CB, SLOTS = NG * MPG, QSP = SLOTS / NQ;
    static_assert(SLOTS % NQ == 0 && QSP >= 1, "quads per MFMA slot");
    auto step = [&](auto STG, auto M1_, auto M2_) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool m1 = decltype(M1_)::value, m2 = decltype(M2_)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (m2) prep();
        if constexpr (m1) read_pro();
        bf16x8 af[2][3][RB], bf[2][3][CB];
        auto frags = [&](auto SET, auto G) {
            constexpr int set = decltype(SET)::value, g = decltype(G)::value;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    af[set][p][i] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(a_ad[stg][g] + (unsigned)(p * A_PLANE + i * 32 * ROWB)));
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    bf[set][p][j] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(b_ad[stg][g] + (unsigned)(p * B_PLANE + j * 32 * ROWB)));
            }
        };
        frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        constexpr int NRD = 3 * (RB + CB);      // fragment reads per K-group: spread over the first MFMAs of the group before
        auto frag_one = [&](auto SET, auto G, auto R) {
            constexpr int set = decltype(SET)::value, g = decltype(G)::value, r = decltype(R)::value;
            constexpr int p = r / (RB + CB), o = r % (RB + CB);
            if constexpr (o < RB)
                af[set][p][o] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(a_ad[stg][g] + (unsigned)(p * A_PLANE + o * 32 * ROWB)));
            else
                bf[set][p][o - RB] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(b_ad[stg][g] + (unsigned)(p * B_PLANE + (o - RB) * 32 * ROWB)));
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, t = w / (RB * CB), ij = w % (RB * CB), i = ij / CB, j = ij % CB;
            // terms in the order  a3 b1, a2 b1, a1 b1, a2 b2, a1 b2, a1 b3  (plane indices 0 = leading)
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 0, 0, 1, 1, 2};
            if constexpr (t == 2)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][0][i], bf[g & 1][0][j], acc[i][j], 0, 0, 0);
            else
                acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][TA[t]][i], bf[g & 1][TB[t]][j], acl[i][j], 0, 0, 0);
            if constexpr (g + 1 < NG) {      // the next K-group's fragments, NRD reads over MPG slots
                constexpr int r0 = w * NRD / MPG, r1 = (w + 1) * NRD / MPG;
                static_for<r1 - r0>([&](auto R) {
                    frag_one(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{},
                             std::integral_constant<int, r0 + decltype(R)::value>{});
                });
            }
            if constexpr (m1) {              // this slot's pieces of the next tile; a finished quad's registers take the load after next
                constexpr int p0 = sl * NPIECE / SLOTS, p1 = (sl + 1) * NPIECE / SLOTS;
                static_for<p1 - p0>([&](auto Q) {
                    constexpr int P = p0 + decltype(Q)::value;
                    piece_at(OTHER{}, std::integral_constant<int, P>{});
                    constexpr int done = quad_done_at(P);
                    if constexpr (m2 && done >= 0) load_quad(OTHER{}, std::integral_constant<int, (done >= 0 ? done : 0)>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (m2) advance();
        if constexpr (m1) cons_c0 = cons_c0 + PBK == a.C ? 0 : cons_c0 + PBK;
    };
    auto fence = [&]() {      // this wave's LDS writes have landed, then the workgroup meets; global loads stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    if constexpr (PRO) __syncthreads();      // the BatchNorm table is in LDS
    if (nsteps > 0) {      // tile 0 -> stage 0 (through set 0); tile 1 -> set 1, tile 2 -> set 0
        prep();
        static_for<NQ>([&](auto K) { load_quad(S0{}, K); });
        advance();
        read_pro();
        static_for<NPIECE>([&](auto P) { piece_at(S0{}, P); });
        cons_c0 = cons_c0 + PBK == a.C ? 0 : cons_c0 + PBK;
        if (nsteps > 1) {
            prep();
            static_for<NQ>([&](auto K) { load_quad(S1{}, K); });
            advance();
        }
        if (nsteps > 2) {
            prep();
            static_for<NQ>([&](auto K) { load_quad(S0{}, K); });
            advance();
        }
    }
    if constexpr (EPF != 0) {
        static_for<EP_ITER * 4>([&](auto O) {
            constexpr int o = decltype(O)::value;
            pf_op(std::integral_constant<int, o / 4>{}, std::integral_constant<int, o % 4>{});
        });
    }
    fence();
    int rem = nsteps;
    for (; rem >= 5; rem -= 2) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, T{});
        fence();
    }
    if (rem == 4) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 3) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, F{}, F{});
    } else if (rem == 2) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 1) {
        step(S0{}, F{}, F{});
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] += acl[i][j];
    if constexpr (EPF != 0) igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256, PFT>(a, acc, m0, n0, mt, split, smem, &pf);
    else igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256>(a, acc, m0, n0, mt, split, smem);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM on operands that ARRIVE split: both tensors exist in HBM as three bf16 planes (IgemmArgs::x3 / w3: plane p
// of element e at  base + 2 (p E + e),  E = the tensor's element count) -- written by the kernels that produce them
// (BatchNorm passes, the stem pool, the per-stage weight pass: all bandwidth-bound, the split rides in their idle vector
// ALUs).  Nothing but copies is left on the way to the matrix cores: a lane moves 16 bytes (8 reduction indices of a row
// of ONE plane) global -> register -> LDS, three planes x (BM + BN) / 64 such units per K-step and thread, each unit's
// store placed behind an MFMA and its next load right behind the store (a K-step of flight time, no second register
// set).  ~1.2 non-MFMA instructions per MFMA instead of ~6.7 with the in-kernel split.
// (Register route instead of LDS-DMA: a 1 KiB LDS-DMA piece costs ~60 issue cycles beside MFMAs, a b128 load + ds_write_b128
// ~20 -- MI355X_MICROARCH.md price list -- and 48 registers are there to take: one workgroup per CU at 128 x 128.)
// ---------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, bool DGRAD, int EPF = 0, int ABL = 0>      // ABL (tuning aid): 1 no global loads, 2 no LDS stores, 4 no MFMAs, 8 no fragment reads
__global__ __launch_bounds__(256) void igemm_x3p_kernel(IgemmArgs a) {
    constexpr int WGM = 2, WGN = 2, PBK = 32;
    constexpr int RB = BM / WGM / 32, CB = BN / WGN / 32;
    constexpr int RPP = 64;                       // rows per loader pass: 4 lanes x 16 bytes per plane row, 16 rows per wave
    constexpr int AP = BM / RPP, BP = BN / RPP, NU = 3 * (AP + BP);      // units (16-byte pieces per thread and step)
    constexpr int ROWB = PBK * 2;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE, STAGE = A_BYTES + B_BYTES;
    constexpr int NG = PBK / 16;
    static_assert(RB >= 1 && CB >= 1 && AP >= 1 && BP >= 1, "bad tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    DPFT_SETPRIO_IGEMM();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int mt, nt, split;
    decode_tile(a, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;

    const int rp = 16 * wave + (lane >> 2);       // row within a pass
    const int cq = lane & 3;                      // 16-byte chunk of the plane row: reduction indices 8 cq .. 8 cq + 7
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
        b_off[i] = n < a.N ? (unsigned)(n * a.Ktot + cq * 8) * 2u : OOB;
    }
    const int pa_bytes = a.B * a.H * a.W * a.C * 2, pb_bytes = a.N * a.Ktot * 2;      // plane strides
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x3), 0, 3 * pa_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w3), 0, 3 * pb_bytes, 0x00020000);
    unsigned a_off[AP];
    auto set_tap = [&](int tap) {
        const int ri = tap / ntap_s, si = tap - ri * ntap_s;
        const int tapoff = (DGRAD ? -1 : 1) * (ri * a.W + si) * a.C;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const bool v = ((a_mask[i] >> ri) & (a_mask[i] >> (8 + si)) & 1u) != 0;
            a_off[i] = v ? (unsigned)(a_row[i] + tapoff + cq * 8) * 2u : OOB;
        }
    };
    const int kt_begin = split * a.ksteps_per_split;
    const int kt_end = min(a.ksteps, kt_begin + a.ksteps_per_split);
    const int nsteps = max(kt_end - kt_begin, 0);
    const int cpt = a.C / PBK;
    int run_tap = kt_begin / cpt, run_c0 = (kt_begin - run_tap * cpt) * PBK, run_koff = 0;
    bool tap_dirty = true;
    int so_a[3], so_b[3];
    auto prep = [&]() {
        if (tap_dirty) {
            set_tap(run_tap);
            const int ri = run_tap / ntap_s, si = run_tap - ri * ntap_s;
            run_koff = (sub ? (a.sub_r0 + a.sub_step * ri) * a.kw + a.sub_s0 + a.sub_step * si : run_tap) * a.C;
            tap_dirty = false;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            so_a[p] = __builtin_amdgcn_readfirstlane(run_c0 * 2 + p * pa_bytes);
            so_b[p] = __builtin_amdgcn_readfirstlane((run_koff + run_c0) * 2 + p * pb_bytes);
        }
    };
    auto advance = [&]() {
        run_c0 += PBK;
        if (run_c0 == a.C) {
            run_c0 = 0;
            ++run_tap;
            tap_dirty = true;
        }
    };

    typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)smem;
    // unit u: u < 3 AP -> A rows of pass u / 3, plane u % 3; else weight rows of pass (u - 3 AP) / 3, plane (u - 3 AP) % 3
    u32x4 ru[2][NU];      // two sets: loads two K-steps ahead (see igemm_x3_kernel)
    auto load_unit = [&](auto SET, auto U) {
        constexpr int set = decltype(SET)::value, u = decltype(U)::value;
        if constexpr (ABL & 1) return;
        if constexpr (u < 3 * AP) ru[set][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)a_off[u / 3], so_a[u % 3], 0));
        else ru[set][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)b_off[(u - 3 * AP) / 3], so_b[(u - 3 * AP) % 3], 0));
    };
    const int w_off = rp * ROWB + ((cq ^ ((rp >> 2) & 3)) << 4);
    auto store_unit = [&](auto STG, auto U) {
        constexpr int stg = decltype(STG)::value, u = decltype(U)::value;
        constexpr int off = u < 3 * AP ? (u % 3) * A_PLANE + (u / 3) * RPP * ROWB
                                       : A_BYTES + ((u - 3 * AP) % 3) * B_PLANE + ((u - 3 * AP) / 3) * RPP * ROWB;
        if constexpr (ABL & 2) return;
        *reinterpret_cast<u32x4*>(lds + stg * STAGE + off + w_off) = ru[stg][u];
    };

    f32x16 acc[RB][CB], acl[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acl[i][j][r] = 0.f; }

    constexpr int EP_ITER = BM * (BN / 4) / 256;
    using PFT = EpiPrefetch<EP_ITER, false, EPF>;
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
            const unsigned el = ok ? (unsigned)(m0 + row) * (unsigned)a.N + (unsigned)(n0 + c4 * 4)
                                   : (unsigned)m0 * (unsigned)a.N + (unsigned)n0;
            using Q = typename PFT::Q;
            if constexpr (kind == 0) pf.y[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.bnr_y) + (size_t)el * 4);
            else if constexpr (kind == 1) { if constexpr (EPF == 1) pf.g[it] = *reinterpret_cast<const Q*>(reinterpret_cast<const char*>(a.res_src) + (size_t)el * 4); }
            else if constexpr (kind == 2) { if (EPF == 1 || a.bnr_mask8) pf.mk[it] = a.bnr_mask8[el >> 2]; }
            else { if constexpr (EPF == 1) pf.rm[it] = a.res_mask8[el >> 2]; }
        }
    };

    const int fkey = ((lane & 31) >> 2) & 3, hh = lane >> 5;
    unsigned a_ad[2][NG], b_ad[2][NG];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int sw = ((2 * g + hh) ^ fkey) << 4;
            a_ad[s][g] = lds_base + s * STAGE + (wm * RB * 32 + (lane & 31)) * ROWB + sw;
            b_ad[s][g] = lds_base + s * STAGE + A_BYTES + (wn * CB * 32 + (lane & 31)) * ROWB + sw;
            asm volatile("" : "+v"(a_ad[s][g]), "+v"(b_ad[s][g]));
        }

    constexpr int MPG = 6 * RB * CB, SLOTS = NG * MPG;
    static_assert(NU <= SLOTS, "more units than MFMA slots");
    auto step = [&](auto STG, auto M1_, auto M2_) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool m1 = decltype(M1_)::value, m2 = decltype(M2_)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (m2) prep();
        bf16x8 af[2][3][RB] = {}, bf[2][3][CB] = {};
        constexpr int NRD = 3 * (RB + CB);
        auto frag_one = [&](auto SET, auto G, auto R) {
            constexpr int set = decltype(SET)::value, g = decltype(G)::value, r = decltype(R)::value;
            constexpr int p = r / (RB + CB), o = r % (RB + CB);
            if constexpr (ABL & 8) return;
            if constexpr (o < RB)
                af[set][p][o] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(a_ad[stg][g] + (unsigned)(p * A_PLANE + o * 32 * ROWB)));
            else
                bf[set][p][o - RB] = __builtin_bit_cast(bf16x8, *(lds_f32x4*)(size_t)(b_ad[stg][g] + (unsigned)(p * B_PLANE + (o - RB) * 32 * ROWB)));
        };
        static_for<NRD>([&](auto R) { frag_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, R); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, t = w / (RB * CB), ij = w % (RB * CB), i = ij / CB, j = ij % CB;
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 0, 0, 1, 1, 2};
            if constexpr (ABL & 4) {}
            else if constexpr (t == 2)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][0][i], bf[g & 1][0][j], acc[i][j], 0, 0, 0);
            else
                acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[g & 1][TA[t]][i], bf[g & 1][TB[t]][j], acl[i][j], 0, 0, 0);
            if constexpr (g + 1 < NG) {
                constexpr int r0 = w * NRD / MPG, r1 = (w + 1) * NRD / MPG;
                static_for<r1 - r0>([&](auto R) {
                    frag_one(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{},
                             std::integral_constant<int, r0 + decltype(R)::value>{});
                });
            }
            if constexpr (m1) {
                constexpr int u0 = sl * NU / SLOTS, u1 = (sl + 1) * NU / SLOTS;
                static_for<u1 - u0>([&](auto Q) {
                    constexpr int u = u0 + decltype(Q)::value;
                    store_unit(OTHER{}, std::integral_constant<int, u>{});
                    if constexpr (m2) load_unit(OTHER{}, std::integral_constant<int, u>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (m2) advance();
    };
    auto fence = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    if (nsteps > 0) {      // tile 0 -> stage 0 (through set 0); tile 1 -> set 1, tile 2 -> set 0
        prep();
        static_for<NU>([&](auto U) { load_unit(S0{}, U); });
        advance();
        static_for<NU>([&](auto U) { store_unit(S0{}, U); });
        if (nsteps > 1) {
            prep();
            static_for<NU>([&](auto U) { load_unit(S1{}, U); });
            advance();
        }
        if (nsteps > 2) {
            prep();
            static_for<NU>([&](auto U) { load_unit(S0{}, U); });
            advance();
        }
    }
    if constexpr (EPF != 0) {
        static_for<EP_ITER * 4>([&](auto O) {
            constexpr int o = decltype(O)::value;
            pf_op(std::integral_constant<int, o / 4>{}, std::integral_constant<int, o % 4>{});
        });
    }
    fence();
    int rem = nsteps;
    for (; rem >= 5; rem -= 2) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, T{});
        fence();
    }
    if (rem == 4) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 3) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, F{}, F{});
    } else if (rem == 2) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 1) {
        step(S0{}, F{}, F{});
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] += acl[i][j];
    if constexpr (EPF != 0) igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256, PFT>(a, acc, m0, n0, mt, split, smem, &pf);
    else igemm_epilogue<BM, BN, WGM, WGN, RB, CB, 256>(a, acc, m0, n0, mt, split, smem);
}

// fp32 -> three bf16 planes (dst: plane p of element e at dst + p n + e), n % 4 == 0
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
        u32x2 p1, p2, p3;
        split3(v, p1, p2, p3);
        *reinterpret_cast<u32x2*>(dst + i) = p1;
        *reinterpret_cast<u32x2*>(dst + n + i) = p2;
        *reinterpret_cast<u32x2*>(dst + 2 * n + i) = p3;
    }
}
int split_planes(const float* src, void* dst, int64_t n, hipStream_t st) {
    DPFT_REQUIRE(src && dst && n > 0 && n % 4 == 0, "split_planes: bad arguments (n %% 4 == 0)");
    const int nb = (int)std::min<int64_t>((n / 4 + 255) / 256, kNumCU * 16);
    hipLaunchKernelGGL(split_planes_kernel, dim3(nb), dim3(256), 0, st, src, (__bf16*)dst, n);
    return check_launch("split_planes");
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient on the split kernels:  dW[n][tap][c] = sum over pixels p of dY[p][n] * act(x)[p @ tap][c].  Both operands are
// pixel-major in memory, the reduction runs over pixels, and v_mfma_f32_32x32x16_bf16 wants 8 consecutive PIXELS of one channel
// per lane: the planes are written pixel-major into LDS ([pixel][channel], a lane's quad = 4 channels of one pixel = 8 bytes per
// plane, 16-byte chunks XOR-swizzled by the pixel) and read with ds_read_b64_tr_b16, which transposes on the way out -- the
// layout and read addressing of wgrad_pipe16_kernel (conv_pipe.h), three planes deep.  Loader side as igemm_x3_kernel: register
// route, split in placeable pieces behind the MFMAs, two register sets (loads two steps ahead); the input-pixel offsets of a
// tap come from a per-workgroup LDS table (wgrad_pipe_kernel).  128 x 128 tile, 32 pixels per step.
// PRO_: 0 none | 1 BatchNorm + ReLU of x | 2 the same with padding (taps that miss the image must stay zero).
// ---------------------------------------------------------------------------------------------------------------------
template <int PRO_>
__global__ __launch_bounds__(256) void wgrad_x3_kernel(WgradArgs a) {
    constexpr bool PRO = PRO_ != 0, MASK = PRO_ == 2;
    constexpr int BMn = 128, BNc = 128, WGM = 2, WGN = 2, PK = 32;
    constexpr int RB = 2, CB = 2;
    constexpr int YL = BMn / 4, XL = BNc / 4;              // lanes (4-channel quads) per pixel row
    constexpr int YRW = 64 / YL, XRW = 64 / XL;            // pixel rows per wave instruction
    constexpr int YRPP = 4 * YRW, XRPP = 4 * XRW;          // pixel rows per pass of the 4 waves
    constexpr int YP = PK / YRPP, XP = PK / XRPP, NQ = YP + XP;
    constexpr int Y_ROWB = BMn * 2, X_ROWB = BNc * 2;      // bytes per plane row (one pixel)
    constexpr int Y_PLANE = PK * Y_ROWB, X_PLANE = PK * X_ROWB;
    constexpr int Y_BYTES = 3 * Y_PLANE, X_BYTES = 3 * X_PLANE, STAGE = Y_BYTES + X_BYTES;
    constexpr int NG = PK / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 stages | pixel-offset table ; epilogue staging
    char* const lds = reinterpret_cast<char*>(smem);
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)smem;
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

    const int yr = YRW * wave + lane / YL, ych = lane % YL;      // this lane's pixel row within a pass / channel quad
    const int xr = XRW * wave + lane / XL, xch = lane % XL;
    const bool y_ok = (n0 + ych * 4) < a.K;
    const bool x_ok = (c0 + xch * 4) < a.C;
    f32x4 p_mu = {0.f, 0.f, 0.f, 0.f}, p_sc = {1.f, 1.f, 1.f, 1.f}, p_sh = {0.f, 0.f, 0.f, 0.f};
    if (PRO && x_ok) {      // the reduction runs over pixels: a lane's four channels never change
        p_mu = *reinterpret_cast<const f32x4*>(a.pro + c0 + xch * 4);
        p_sc = *reinterpret_cast<const f32x4*>(a.pro + a.C + c0 + xch * 4);
        p_sh = *reinterpret_cast<const f32x4*>(a.pro + 2 * a.C + c0 + xch * 4);
    }
    const int ohw = a.OH * a.OW;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.M * a.K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.B * a.H * a.W * a.C * 4, 0x00020000);
    unsigned y_voff[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) y_voff[i] = y_ok ? (unsigned)((YRPP * i + yr) * a.K + n0 + ych * 4) * 4u : OOB;
    const int ps_begin = split * a.psteps_per_split;
    const int ps_end = min(a.psteps, ps_begin + a.psteps_per_split);
    const int nsteps = max(ps_end - ps_begin, 0);
    unsigned* const tbl = reinterpret_cast<unsigned*>(lds + 2 * STAGE);
    {      // input-pixel byte offsets of this workgroup's pixel range for its tap (or the out-of-range marker)
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
    unsigned tbl_ad = lds_base + 2 * STAGE + xr * 4;      // LDS address of this lane's first row of the next tile to load
    const unsigned x_lane = x_ok ? (unsigned)(c0 + xch * 4) * 4u : OOB;
    int next_ps = ps_begin;
    int so_y = 0;
    unsigned xoffs[XP];
    auto prep = [&]() {
        so_y = __builtin_amdgcn_readfirstlane(next_ps * PK * a.K * 4);
#pragma unroll
        for (int i = 0; i < XP; ++i) xoffs[i] = *(lds_u32*)(size_t)(tbl_ad + (unsigned)(XRPP * i * 4)) + x_lane;
        tbl_ad += PK * 4;
        ++next_ps;
    };
    f32x4 rq[2][NQ];                    // quads 0 .. YP-1: dY rows, YP .. NQ-1: x rows
    unsigned rx_bad[2] = {0, 0};        // bit i: x quad i of the set was loaded from outside the image (MASK)
    auto load_quad = [&](auto SET, auto K) {
        constexpr int set = decltype(SET)::value, k = decltype(K)::value;
        if constexpr (k < YP) {
            rq[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, (int)y_voff[k], so_y, 0));
        } else {
            constexpr int i = k - YP;
            rq[set][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)xoffs[i], 0, 0));
            if constexpr (MASK) rx_bad[set] = (rx_bad[set] & ~(1u << i)) | ((xoffs[i] >> 31) << i);
        }
    };
    // LDS destination of a quad: pixel row (yr / xr + rows per pass * pass), 8 bytes at half (quad & 1) of chunk (quad / 2) ^ key,
    // key = (pixel % 4) * 4 (wgrad_pipe16_kernel's swizzle; rows per pass are multiples of 4, so the key is the lane's)
    const int yw_off = yr * Y_ROWB + ((((ych >> 1) ^ ((yr & 3) << 2)) << 4) | ((ych & 1) << 3));
    const int xw_off = xr * X_ROWB + ((((xch >> 1) ^ ((xr & 3) << 2)) << 4) | ((xch & 1) << 3));
    f32x4 cv;
    u32x2 c1, c2, c3;
    float rr0 = 0.f, rr1 = 0.f;
    auto piece = [&](auto STG, auto K, auto ID) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value, k = decltype(K)::value, id = decltype(ID)::value;
        constexpr bool bn = PRO && k >= YP;
        if constexpr (id == 0 || id == 1) {
            if constexpr (id == 0) cv = rq[stg][k];
#pragma unroll
            for (int e = 2 * id; e < 2 * id + 2; ++e) cv[e] = fmaxf(fmaf(cv[e] - p_mu[e], p_sc[e], p_sh[e]), 0.f);
        } else if constexpr (id == 2) {
            const unsigned keep = ~(unsigned)__builtin_amdgcn_sbfe(rx_bad[stg], k - YP, 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) cv[e] = __uint_as_float(__float_as_uint(cv[e]) & keep);
        } else if constexpr (id == 3 || id == 5) {
            constexpr int h = (id - 3) / 2;
            if constexpr (id == 3 && !bn) cv = rq[stg][k];
            const float x = cv[2 * h], y = cv[2 * h + 1];
            const unsigned q = cvt_pk_bf16(x, y);
            c1[h] = q;
            rr0 = x - __uint_as_float(q << 16);
            rr1 = y - __uint_as_float(q & 0xffff0000u);
        } else if constexpr (id == 4 || id == 6) {
            constexpr int h = (id - 4) / 2;
            const unsigned q = cvt_pk_bf16(rr0, rr1);
            c2[h] = q;
            c3[h] = cvt_pk_bf16(rr0 - __uint_as_float(q << 16), rr1 - __uint_as_float(q & 0xffff0000u));
        } else {
            constexpr int PL = k < YP ? Y_PLANE : X_PLANE;
            char* dst = lds + stg * STAGE + (k < YP ? k * YRPP * Y_ROWB + 0 : Y_BYTES + (k - YP) * XRPP * X_ROWB) + (k < YP ? yw_off : xw_off);
            *reinterpret_cast<u32x2*>(dst) = c1;
            *reinterpret_cast<u32x2*>(dst + PL) = c2;
            *reinterpret_cast<u32x2*>(dst + 2 * PL) = c3;
        }
    };
    constexpr int NPY = 5, NPX = PRO ? (MASK ? 8 : 7) : 5, NPIECE = YP * NPY + XP * NPX;
    auto piece_at = [&](auto STG, auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        if constexpr (P < YP * NPY) {
            piece(STG, std::integral_constant<int, P / NPY>{}, std::integral_constant<int, 3 + P % NPY>{});
        } else {
            constexpr int k = YP + (P - YP * NPY) / NPX, o = (P - YP * NPY) % NPX;
            constexpr int id = !PRO ? o + 3 : (MASK ? o : (o < 2 ? o : o + 1));
            piece(STG, std::integral_constant<int, k>{}, std::integral_constant<int, id>{});
        }
    };
    auto quad_done_at = [](int P) constexpr {
        if (P < YP * NPY) return P % NPY == NPY - 1 ? P / NPY : -1;
        return (P - YP * NPY) % NPX == NPX - 1 ? YP + (P - YP * NPY) / NPX : -1;
    };

    f32x16 acc[RB][CB], acl[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[i][j][q] = 0.f; acl[i][j][q] = 0.f; }

    // transpose-read addresses (wgrad_pipe16_kernel): supplier lane (group G = lane / 16, s = lane % 16) -> pixel (G / 2) * 8 + s / 4
    // of the K-group, channel quad s % 4 of the 16 channels (G % 2) of a 32-channel block
    unsigned y_ad[2][RB], x_ad[2][CB];
    {
        const int G = lane >> 4, sl = lane & 15, r2 = sl >> 2, q = sl & 3;
        const int pix = (G >> 1) * 8 + r2;
        const int key = r2 << 2;
#pragma unroll
        for (int st_ = 0; st_ < 2; ++st_) {
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int c = (wm * RB + i) * 4 + 2 * (G & 1) + (q >> 1);
                y_ad[st_][i] = lds_base + st_ * STAGE + pix * Y_ROWB + ((c ^ key) << 4) + (q & 1) * 8;
            }
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int c = (wn * CB + j) * 4 + 2 * (G & 1) + (q >> 1);
                x_ad[st_][j] = lds_base + st_ * STAGE + Y_BYTES + pix * X_ROWB + ((c ^ key) << 4) + (q & 1) * 8;
            }
        }
    }
#pragma unroll
    for (int st_ = 0; st_ < 2; ++st_) {
#pragma unroll
        for (int i = 0; i < RB; ++i) asm volatile("" : "+v"(y_ad[st_][i]));
#pragma unroll
        for (int j = 0; j < CB; ++j) asm volatile("" : "+v"(x_ad[st_][j]));
    }

    constexpr int MPG = 6 * RB * CB, SLOTS = NG * MPG;
    constexpr int NRD = 3 * 2 * (RB + CB);      // transpose reads per K-group: 3 planes x 2 halves x (RB + CB)
    typedef __attribute__((address_space(3))) const u32x2 lds_u32x2;
    auto step = [&](auto STG, auto M1_, auto M2_) __attribute__((always_inline)) {
        constexpr int stg = decltype(STG)::value;
        constexpr bool m1 = decltype(M1_)::value, m2 = decltype(M2_)::value;
        using OTHER = std::integral_constant<int, (stg ^ 1)>;
        if constexpr (m2) prep();
        u32x2 av[2][3][RB][2], bv[2][3][CB][2];
        auto frag_one = [&](auto SET, auto G, auto R) {
            constexpr int set = decltype(SET)::value, g = decltype(G)::value, rr = decltype(R)::value;
            constexpr int p = rr / (2 * (RB + CB)), o = (rr % (2 * (RB + CB))) / 2, h = rr % 2;
            if constexpr (o < RB) {
                u32x2& dst = av[set][p][o][h];
                const unsigned ad = y_ad[stg][o];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(p * Y_PLANE + (g * 16 + h * 4) * Y_ROWB));
            } else {
                u32x2& dst = bv[set][p][o - RB][h];
                const unsigned ad = x_ad[stg][o - RB];
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(ad), "n"(p * X_PLANE + (g * 16 + h * 4) * X_ROWB));
            }
        };
        static_for<NRD>([&](auto R) { frag_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, R); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto S) {
            constexpr int sl = decltype(S)::value;
            constexpr int g = sl / MPG, w = sl % MPG, t = w / (RB * CB), ij = w % (RB * CB), i = ij / CB, j = ij % CB;
            constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 0, 0, 1, 1, 2};
            if constexpr (w == 0) {
                // the K-group's fragments (issued a group earlier) have landed.  Inline-asm reads are invisible to the compiler's
                // counters; the fragment registers are operands of the wait so that nothing that uses them can move in front of it
                auto& A_ = av[g & 1];
                auto& B_ = bv[g & 1];
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(A_[0][0][0]), "+v"(A_[0][0][1]), "+v"(A_[0][1][0]), "+v"(A_[0][1][1]), "+v"(A_[1][0][0]), "+v"(A_[1][0][1]),
                               "+v"(A_[1][1][0]), "+v"(A_[1][1][1]), "+v"(A_[2][0][0]), "+v"(A_[2][0][1]), "+v"(A_[2][1][0]), "+v"(A_[2][1][1]),
                               "+v"(B_[0][0][0]), "+v"(B_[0][0][1]), "+v"(B_[0][1][0]), "+v"(B_[0][1][1]), "+v"(B_[1][0][0]), "+v"(B_[1][0][1]),
                               "+v"(B_[1][1][0]), "+v"(B_[1][1][1]), "+v"(B_[2][0][0]), "+v"(B_[2][0][1]), "+v"(B_[2][1][0]), "+v"(B_[2][1][1])
                             :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int pa = t == 2 ? 0 : TA[t], pb = t == 2 ? 0 : TB[t];
            const u32x4 fa = __builtin_shufflevector(av[g & 1][pa][i][0], av[g & 1][pa][i][1], 0, 1, 2, 3);
            const u32x4 fb = __builtin_shufflevector(bv[g & 1][pb][j][0], bv[g & 1][pb][j][1], 0, 1, 2, 3);
            if constexpr (t == 2)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i][j], 0, 0, 0);
            else
                acl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acl[i][j], 0, 0, 0);
            if constexpr (g + 1 < NG) {
                constexpr int r0 = w * NRD / MPG, r1 = (w + 1) * NRD / MPG;
                static_for<r1 - r0>([&](auto R) {
                    frag_one(std::integral_constant<int, ((g + 1) & 1)>{}, std::integral_constant<int, (g + 1 < NG ? g + 1 : 0)>{},
                             std::integral_constant<int, r0 + decltype(R)::value>{});
                });
            }
            if constexpr (m1) {
                constexpr int p0 = sl * NPIECE / SLOTS, p1 = (sl + 1) * NPIECE / SLOTS;
                static_for<p1 - p0>([&](auto Q) {
                    constexpr int P = p0 + decltype(Q)::value;
                    piece_at(OTHER{}, std::integral_constant<int, P>{});
                    constexpr int done = quad_done_at(P);
                    if constexpr (m2 && done >= 0) load_quad(OTHER{}, std::integral_constant<int, (done >= 0 ? done : 0)>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto fence = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    if (nsteps > 0) {      // tile 0 -> stage 0 (through set 0); tile 1 -> set 1, tile 2 -> set 0
        prep();
        static_for<NQ>([&](auto K) { load_quad(S0{}, K); });
        static_for<NPIECE>([&](auto P) { piece_at(S0{}, P); });
        if (nsteps > 1) {
            prep();
            static_for<NQ>([&](auto K) { load_quad(S1{}, K); });
        }
        if (nsteps > 2) {
            prep();
            static_for<NQ>([&](auto K) { load_quad(S0{}, K); });
        }
    }
    fence();
    int rem = nsteps;
    for (; rem >= 5; rem -= 2) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, T{});
        fence();
    }
    if (rem == 4) {
        step(S0{}, T{}, T{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 3) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, T{}, F{});
        fence();
        step(S0{}, F{}, F{});
    } else if (rem == 2) {
        step(S0{}, T{}, F{});
        fence();
        step(S1{}, F{}, F{});
    } else if (rem == 1) {
        step(S0{}, F{}, F{});
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] += acl[i][j];
    // epilogue: as wgrad_pipe_kernel (fp32 weight gradients / pixel-split partials)
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
                *reinterpret_cast<f32x4*>(out + ((size_t)n * a.taps + tap) * a.C + c) = *reinterpret_cast<const f32x4*>(&Cs[row * LDC + c4 * 4]);
        }
    }
}

int launch_wgrad_x3(const WgradArgs& a, bool pro, dim3 grid, hipStream_t st) {
    const size_t lds = (size_t)2 * 3 * 32 * (128 + 128) * 2 + (size_t)a.psteps_per_split * 32 * 4;
    const bool padded = a.kh * a.kw > 1 || a.pad > 0;
    auto go = [&](auto kernel) {
        static LdsGrant grant;
        (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
    };
    if (pro && padded) go(wgrad_x3_kernel<2>);
    else if (pro) go(wgrad_x3_kernel<1>);
    else go(wgrad_x3_kernel<0>);
    return check_launch("conv wgrad (3 x bf16 split)");
}

template <typename K>
static void launch_x3(K kernel, dim3 grid, size_t lds, hipStream_t st, const IgemmArgs& args) {
    static LdsGrant grant;      // one per kernel instantiation
    (void)lds_grant(grant, reinterpret_cast<const void*>(kernel), lds);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, args);
}

// `a` as launch_igemm (conv.hip) has prepared it (tile counts, splits, ksteps in 64-deep units); bm x bn in {128x128, 128x64, 64x64}
int launch_igemm_x3(IgemmArgs& a, int bm, int bn, bool dgrad, bool pro, hipStream_t st) {
    a.ksteps = a.ksteps * 2;      // 32-deep K-steps
    a.ksteps_per_split = cdiv(a.ksteps, a.splits);
    const dim3 grid(a.mtiles * a.ntiles * a.splits);
    const bool padded = a.kh * a.kw > 1 || a.pad > 0;
    if (a.x3 && a.w3) {      // both operands arrive as planes
        DPFT_REQUIRE(!pro, "conv x3 planes: no operand prologue (the planes hold the activation)");
        DPFT_REQUIRE((int64_t)a.B * a.H * a.W * a.C * 6 < (1ll << 31) && (int64_t)a.N * a.Ktot * 6 < (1ll << 31), "conv x3 planes: operand larger than 2 GiB");
        auto ldsp = [&](int BM_, int BN_) { return std::max((size_t)2 * 3 * (BM_ + BN_) * 64, (size_t)BM_ * (BN_ + 4) * 4 + (size_t)3 * BN_ * 4); };
#define X3P_GO(BM_, BN_, DG_, EPF_) launch_x3(igemm_x3p_kernel<BM_, BN_, DG_, EPF_>, grid, ldsp(BM_, BN_), st, a)
#define X3P_TILE(BM_, BN_)                                                   \
    do {                                                                     \
        if (!dgrad) X3P_GO(BM_, BN_, false, 0);                              \
        else if (a.epf == 1 && BM_ * BN_ < 128 * 128) X3P_GO(BM_, BN_, true, 1); \
        else if (a.epf == 2 && BM_ * BN_ < 128 * 128) X3P_GO(BM_, BN_, true, 2); \
        else X3P_GO(BM_, BN_, true, 0);                                      \
    } while (0)
        static const int abl = getenv("DPFT_X3_ABL") ? atoi(getenv("DPFT_X3_ABL")) : 0;
        if (abl && !dgrad && bm == 128 && bn == 128) {
            switch (abl) {
                case 1: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 1>, grid, ldsp(128, 128), st, a); break;
                case 2: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 2>, grid, ldsp(128, 128), st, a); break;
                case 3: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 3>, grid, ldsp(128, 128), st, a); break;
                case 4: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 4>, grid, ldsp(128, 128), st, a); break;
                case 8: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 8>, grid, ldsp(128, 128), st, a); break;
                case 11: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 11>, grid, ldsp(128, 128), st, a); break;
                default: launch_x3(igemm_x3p_kernel<128, 128, false, 0, 12>, grid, ldsp(128, 128), st, a); break;
            }
            return check_launch("conv igemm (3 x bf16 planes, ablation)");
        }
        if (bm == 128 && bn == 128) X3P_TILE(128, 128);
        else if (bm == 128 && bn == 64) X3P_TILE(128, 64);
        else X3P_TILE(64, 64);
#undef X3P_TILE
#undef X3P_GO
        return check_launch("conv igemm (3 x bf16 planes)");
    }
    // 128 x 128 tiles on eight waves, two per SIMD (conv_x3w.hip, DPFT_X3W=1): the plain form measures within 3 % of this file's
    // hand-pipelined four-wave kernel on every 128 x 128 problem of the step (profiles/r06_x3w.txt) -- the plateau is not latency.  Off.
    static const int x3w = getenv("DPFT_X3W") ? atoi(getenv("DPFT_X3W")) : 0;
    if (x3w && bm == 128 && bn == 128 && !a.bnf_acc && !a.bnf_slab && (a.N & 3) == 0) return launch_igemm_x3w(a, dgrad, pro, st);
    auto lds_of = [&](int BM_, int BN_) {
        return std::max((size_t)2 * 3 * (BM_ + BN_) * 64 + (pro ? (size_t)12 * a.C : 0), (size_t)BM_ * (BN_ + 4) * 4 + (size_t)3 * BN_ * 4);
    };
#define X3_GO(BM_, BN_, DG_, PRO_, EPF_) launch_x3(igemm_x3_kernel<BM_, BN_, DG_, PRO_, EPF_>, grid, lds_of(BM_, BN_), st, a)
#define X3_TILE(BM_, BN_)                                                    \
    do {                                                                     \
        if (!dgrad) { if (pro && padded) X3_GO(BM_, BN_, false, 2, 0); else if (pro) X3_GO(BM_, BN_, false, 1, 0); else X3_GO(BM_, BN_, false, 0, 0); } \
        else if (a.epf == 1 && BM_ * BN_ < 128 * 128) X3_GO(BM_, BN_, true, 0, 1);                \
        else if (a.epf == 2 && BM_ * BN_ < 128 * 128) X3_GO(BM_, BN_, true, 0, 2);                \
        else X3_GO(BM_, BN_, true, 0, 0);                                \
    } while (0)
    if (bm == 128 && bn == 128) X3_TILE(128, 128);
    else if (bm == 128 && bn == 64) X3_TILE(128, 64);
    else X3_TILE(64, 64);
#undef X3_TILE
#undef X3_GO
    return check_launch("conv igemm (3 x bf16 split)");
}

}  // namespace dpft
