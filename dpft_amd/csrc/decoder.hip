// Fused inference path of the iterative fusion decoder (eval mode, no autograd) for d_model = 16, 8 heads
// (head_dim 2), 5 levels x 4 points, Mish FFN, LayerNorm, 'linear' view reduction, 3-layer linear heads --
// i.e. IMPFusion / MPFusion / MLFusion / LinearDetectionHead of the reference at config/kradar*.json:
//   src/dprt/models/fusers/mpfusion.py:122-148 (self attention), :150-208 + layers/ms_deform_attn.py:138-217
//   (deformable cross attention), :210-229 (FFN), :416-470,:472-514 (view reduction), :617-696 (reference
//   points), :698-745 (iteration), src/dprt/models/heads/detection.py:252-275 (head).
// The eager decoder is ~700 launches of B*400x16-sized ops per forward (launch-bound: ~5 ms even when
// replayed from a hipGraph).  Round 1 ran 2 kernels per iteration (22 + 41 us: redundant per-lane work, 30 KB of
// weights re-read per query row); round 2 runs 2 launches per iteration + 1 (9 for 4 iterations, ~130 us at B = 4):
//   A decoder_scores_head_kernel, two kinds of blocks in ONE launch:
//       score blocks  (50-query chunk, head, view x batch): a head's q/k/v are 2+2+2 of the 48 in_proj rows, so every
//           block builds only ITS head's K/V (no redundancy across heads).  From iteration 1 on these rows are not
//           projected at all: the previous cross-attention kernel already wrote each view's share of them (the next
//           layer's in_proj composed with the 48->16 view reduction, "hand-over"), the block sums V 16-byte partials per
//           key + a packed position table.  Thread = (query pair, key slice); one pass against the Cauchy-Schwarz
//           bound |q| max|k| in the exp2 domain (1/sqrt(d) log2(e) folded into q; exact two-pass redo should a
//           query underflow), two queries per packed-fp32 operation, slice partials merged through LDS.  Writes the
//           pre-out_proj attention output.  Iteration 0 runs for ONE batch element (its input does not depend on b).
//       head blocks   (wave = (b, q)) of the PREVIOUS iteration: 48->16 view reduction, the 4 head MLPs,
//           center += previous, and the next reference points of all views.  Both kinds depend only on the previous
//           cross-attention kernel, so the head MLPs' latency chain hides behind the scores.
//   B decoder_xattn_kernel: block = 7 query rows of ONE view, one wave per row, the view's packed weights (40 KB)
//       staged once per block in LDS (3 blocks per CU).  Per row: out_proj + LN1, offsets/logits GEMV whose packed
//       column order delivers every lane exactly the three (head, level, point) samples it will produce, softmax
//       across lanes (DPP / permlane swaps, no LDS round trips), bilinear weights + corner address computed ONCE per
//       sample and handed to the 4-lane pixel groups through a 1.5 KB per-wave LDS scratch, dwordx4 gathers (4 lanes
//       = one 64-byte NHWC pixel, 8 in flight per lane), sample-then-project, output_proj + LN2, Mish FFN + LN3,
//       and the hand-over rows for the next layer's self-attention.
// Measured phase structure (tools/decoder_stamps.py): every kernel pays ~2 us of launch + ~2-3 us until its first
// loads return (data written by the previous kernel from other XCDs); inside B the gather phase runs at the L2's
// line rate (197 MB of 64-byte line requests per launch) and, with all waves in lock step, does not overlap the GEMV.
#include "common.h"
#include "decoder_pack.h"
#include <stdlib.h>

#define RC(call)              \
    do {                      \
        int rc_ = (call);     \
        if (rc_) return rc_;  \
    } while (0)

namespace dpft {

__device__ __forceinline__ void pack_view_body(const dpft_decoder_view& s, int n_off, int n_att, float* __restrict__ d) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PV_FLOATS; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        int r;
        if (i < PV_IN_B) v = s.in_proj_w[i];
        else if (i < PV_OUT_WT) v = s.in_proj_b[i - PV_IN_B];
        else if (i < PV_OUT_B) { r = i - PV_OUT_WT; v = s.out_proj_w[(r & 15) * 16 + (r >> 4)]; }
        else if (i < PV_N1_W) v = s.out_proj_b[i - PV_OUT_B];
        else if (i < PV_N1_B) v = s.norm1_w[i - PV_N1_W];
        else if (i < PV_OA_WT) v = s.norm1_b[i - PV_N1_B];
        else if (i < PV_OA_B) {
            r = i - PV_OA_WT;
            const int c = r / NOA, o = r - c * NOA;
            v = o < n_off ? s.off_w[o * 16 + c] : (o < n_off + n_att ? s.att_w[(o - n_off) * 16 + c] : 0.f);
        } else if (i < PV_VAL_W) {
            const int o = i - PV_OA_B;
            v = o < n_off ? s.off_b[o] : (o < n_off + n_att ? s.att_b[o - n_off] : 0.f);
        } else if (i < PV_VAL_B) v = s.val_w[i - PV_VAL_W];
        else if (i < PV_OUTP_WT) v = s.val_b[i - PV_VAL_B];
        else if (i < PV_OUTP_B) { r = i - PV_OUTP_WT; v = s.outp_w[(r & 15) * 16 + (r >> 4)]; }
        else if (i < PV_N2_W) v = s.outp_b[i - PV_OUTP_B];
        else if (i < PV_N2_B) v = s.norm2_w[i - PV_N2_W];
        else if (i < PV_F1_WT) v = s.norm2_b[i - PV_N2_B];
        else if (i < PV_F1_B) { r = i - PV_F1_WT; v = s.ffn1_w[(r & 31) * 16 + (r >> 5)]; }
        else if (i < PV_F2_WT) v = s.ffn1_b[i - PV_F1_B];
        else if (i < PV_F2_B) { r = i - PV_F2_WT; v = s.ffn2_w[(r & 15) * 32 + (r >> 4)]; }
        else if (i < PV_N3_W) v = s.ffn2_b[i - PV_F2_B];
        else if (i < PV_N3_B) v = s.norm3_w[i - PV_N3_W];
        else v = s.norm3_b[i - PV_N3_B];
        d[i] = v;
    }
}

__global__ void pack_view_kernel(dpft_decoder_view s, int n_off, int n_att, float* __restrict__ d) { pack_view_body(s, n_off, n_att, d); }

// all views of a layer in one launch (blockIdx.y = view): the training decoder packed its three views with three launches
// per layer and forward (round 4)
struct PackViews {
    dpft_decoder_view v[4];
    int n_off[4], n_att[4];
    float* d[4];
};
__global__ void pack_views_kernel(PackViews p) {
    switch (blockIdx.y) {      // constant indices: the argument struct stays in the kernarg segment
        case 0: pack_view_body(p.v[0], p.n_off[0], p.n_att[0], p.d[0]); break;
        case 1: pack_view_body(p.v[1], p.n_off[1], p.n_att[1], p.d[1]); break;
        case 2: pack_view_body(p.v[2], p.n_off[2], p.n_att[2], p.d[2]); break;
        default: pack_view_body(p.v[3], p.n_off[3], p.n_att[3], p.d[3]); break;
    }
}

struct HeadSrc {
    const float* red_w;
    const float* hw[4][3];
    int nout[4];
    int V;
};
__global__ void pack_head_kernel(HeadSrc s, float* __restrict__ d) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < PH_FLOATS; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < PH_W) {
            const int vw = i >> 8, k = (i >> 4) & 15, o = i & 15;
            if (vw < s.V) v = s.red_w[o * DC * s.V + k * s.V + vw];
        } else {
            const int r = i - PH_W;
            const int layer = r >> 10, k = (r >> 6) & 15, g = (r >> 4) & 3, o = r & 15;
            const int rows = layer == 2 ? s.nout[g] : DC;
            if (o < rows) v = s.hw[g][layer][o * DC + k];
        }
        d[i] = v;
    }
}

// Cross-lane exchanges as DPP / permlane-swap VALU operations (a __shfl_xor is a ds_bpermute: an LDS-pipe round trip of
// ~100 cycles, and the LayerNorms / softmax reductions chain 4-8 of them per row).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
// value of lane ^ 16 / lane ^ 32 combined with the own value by `op` (all-reduce step)
template <class Op>
__device__ __forceinline__ float xor16_combine(float v, Op op) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return op(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
template <class Op>
__device__ __forceinline__ float xor32_combine(float v, Op op) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), false, false);
    return op(__builtin_bit_cast(float, (int)r[0]), __builtin_bit_cast(float, (int)r[1]));
}
struct OpAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };

// sum over the 16 lanes of a row, result in every lane (quad butterflies, then the mirrored half / row)
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_mov<DPP_XOR1>(v);
    v += dpp_mov<DPP_XOR2>(v);
    v += dpp_mov<DPP_HALF_MIRROR>(v);
    v += dpp_mov<DPP_MIRROR>(v);
    return v;
}

// sum over the 4 lanes c, c + 16, c + 32, c + 48 (k-slices of one output), result in all of them
__device__ __forceinline__ float slices4_sum(float v) {
    v = xor16_combine(v, OpAdd());
    return xor32_combine(v, OpAdd());
}

// LayerNorm over 16 channels held by 16 consecutive lanes (torch: biased variance, eps inside the sqrt)
__device__ __forceinline__ float layernorm16(float v, float g, float b) {
    const float mean = group16_sum(v) * (1.f / 16.f);
    const float d = v - mean;
    const float var = group16_sum(d * d) * (1.f / 16.f);
    return d * __builtin_amdgcn_rsqf(var + 1e-5f) * g + b;      // v_rsq_f32 (1 ulp) instead of an IEEE sqrt + division
}

// ---------------------------------------------------------------------------------------------------------
// packed INFERENCE blob of one MLFusion (per iteration, view), made by pack_infer_kernel
// ---------------------------------------------------------------------------------------------------------
constexpr int NSLOT = 160;                          // 8 heads x 20 (level, point) slots, in the lanes' slot order
constexpr int PI_K2 = 0;                            // LDS image of decoder_xattn_kernel (16-B aligned)
constexpr int K2_OFF = 0;                           // [17][NSLOT][2] sampling_offsets^T (row 16 = bias), slot order
constexpr int K2_LOG = K2_OFF + 17 * NSLOT * 2;     // [17][NSLOT]    attention_weights^T (row 16 = bias)
constexpr int K2_VALW = K2_LOG + 17 * NSLOT;        // value_proj.weight (16,16) as is
constexpr int K2_VALB = K2_VALW + 256;
// small matrices in torch's own (out, in) layout, rows padded (+4) so that 16 lanes reading 16-byte pieces of 16
// different rows hit 64 different LDS banks: every lane computes a k-SLICE of its output (the 64 lanes hold each of
// the 16 / 32 outputs 4 / 2 times) and the slices are merged with permlane swaps -- 8 + 2 instead of 32 x 4 instructions
constexpr int LD16 = 20, LD32 = 36;
constexpr int K2_OUTP = K2_VALB + 16;               // output_proj.weight [c 16][k 16 (+4)]
constexpr int K2_OUTPB = K2_OUTP + 16 * LD16;
constexpr int K2_N2W = K2_OUTPB + 16;
constexpr int K2_N2B = K2_N2W + 16;
constexpr int K2_F1 = K2_N2B + 16;                  // ffn1.weight [j 32][k 16 (+4)]
constexpr int K2_F1B = K2_F1 + 32 * LD16;
constexpr int K2_F2 = K2_F1B + 32;                  // ffn2.weight [c 16][k 32 (+4)]
constexpr int K2_F2B = K2_F2 + 16 * LD32;
constexpr int K2_N3W = K2_F2B + 16;
constexpr int K2_N3B = K2_N3W + 16;
constexpr int K2_SAO = K2_N3B + 16;                 // self_attn.out_proj.weight [c 16][k 16 (+4)]
constexpr int K2_SAOB = K2_SAO + 16 * LD16;
constexpr int K2_N1W = K2_SAOB + 16;
constexpr int K2_N1B = K2_N1W + 16;
constexpr int K2_SLOT = K2_N1B + 16;                // [NSLOT] int8: pyramid level of the slot, -1 = unused slot (n >= L*P)
constexpr int K2_FLOATS = K2_SLOT + NSLOT / 4;      // 10504 floats = 42 016 B
constexpr int PI_FLOATS = PI_K2 + K2_FLOATS;
static_assert(K2_FLOATS % 4 == 0 && PI_K2 % 4 == 0 && PI_FLOATS % 4 == 0, "float4 staging");
// Hand-over to the NEXT layer's self-attention (appended to the blob).  The next layer's query is R y3cat (R = this
// layer's 48->16 view reduction), and its q/k/v rows are linear in it, so every view's cross-attention wave can emit
// its own share of them, NX[v'] y3_v with NX[v'] = W_in(next layer, target view v') R[:, view block v], right where
// y3 is produced; the score kernel of the next iteration then sums V 16-byte partials per key instead of projecting
// 192 bytes of y3, and no longer depends on the reduction / head kernel (they share a launch).
//   PC_NX : [4 targets][16 k][64 outputs], output j = head * 8 + {0,1: q (pre-scaled) | 4,5: k | 6,7: v}, 2,3 unused
//   PC_T  : input-independent part of THIS layer's q / k / v rows
constexpr int PC_NX = PI_FLOATS;
constexpr int PC_T = PC_NX + 4 * 16 * 64;                      // [8 heads][Q][8] = q0 q1 - - k0 k1 v0 v1: the part of the rows that does
                                                    // not depend on the input: W (pos_k [+ query0_k in the first layer]) + b
__host__ __device__ constexpr int64_t pi_floats(int Q) { return PC_T + (int64_t)Q * 64; }

// slot (s, lane) -> head m, sample n of the head.  In gather round t = 4 s + (lane >> 4) the 4-lane pixel group
// g = lane >> 2 ... of the CONSUMER reads the sample the PRODUCER lane (t & 3) * 16 + g computed, so that group g always
// serves head g & 7 (its accumulator never changes head) and the two groups g, g + 8 split a head's 20 samples.
__device__ __forceinline__ void slot_decode(int s, int lane, int& m, int& n) {
    m = lane & 7;
    n = 2 * (4 * s + (lane >> 4)) + ((lane >> 3) & 1);
}

struct NextInProj {
    const float* w[4];      // in_proj_weight (48,16) of the NEXT layer's views, all NULL for the last layer
};
__global__ void pack_infer_kernel(dpft_decoder_view s, int L, int P, const float* __restrict__ red_w, NextInProj nx,
                                  int view, int V, const float* __restrict__ pos, const float* __restrict__ query0, int Q,
                                  float* __restrict__ d) {
    const int LP = L * P;
    const float qscale = 0.70710678118654752f * 1.4426950408889634f;      // 1/sqrt(head_dim) * log2(e)
    const int total = (int)pi_floats(Q);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i >= PC_T) {                          // input-independent part of the q / k / v rows of (head, key)
            const int j = i - PC_T, h = j / (Q * 8), k = (j >> 3) % Q, e = j & 7;
            if (e < 2 || e >= 4) {
                const int row = (e < 2 ? 0 : (e < 6 ? 16 : 32)) + 2 * h + (e & 1);
                v = s.in_proj_b[row];
                for (int c = 0; c < 16; ++c) {
                    const float x = (e < 6 ? pos[k * 16 + c] : 0.f) + (query0 ? query0[k * 16 + c] : 0.f);   // v rows see no pos
                    v = fmaf(s.in_proj_w[row * 16 + c], x, v);
                }
                if (e < 2) v *= qscale;
            }
        } else if (i >= PC_NX) {
            const int j = i - PC_NX, tv = j >> 10, k = (j >> 6) & 15, o = j & 63, h = o >> 3, e = o & 7;
            if (tv < V && nx.w[tv] && (e < 2 || e >= 4)) {
                const int row = (e < 2 ? 0 : (e < 6 ? 16 : 32)) + 2 * h + (e & 1);
                for (int c = 0; c < 16; ++c) v = fmaf(nx.w[tv][row * 16 + c], red_w[c * 16 * V + k * V + view], v);
                if (e < 2) v *= qscale;
            }
        } else {
            const int k = i - PI_K2;
            int r;
            if (k < K2_LOG) {
                const int c = k / (NSLOT * 2), slot = (k / 2) % NSLOT, xy = k & 1;
                int m, n;
                slot_decode(slot >> 6, slot & 63, m, n);
                if (n < LP) { const int o = (m * LP + n) * 2 + xy; v = c < 16 ? s.off_w[o * 16 + c] : s.off_b[o]; }
            } else if (k < K2_VALW) {
                r = k - K2_LOG;
                const int c = r / NSLOT, slot = r % NSLOT;
                int m, n;
                slot_decode(slot >> 6, slot & 63, m, n);
                if (n < LP) { const int o = m * LP + n; v = c < 16 ? s.att_w[o * 16 + c] : s.att_b[o]; }
            } else if (k < K2_VALB) v = s.val_w[k - K2_VALW];
            else if (k < K2_OUTP) v = s.val_b[k - K2_VALB];
            else if (k < K2_OUTPB) { r = k - K2_OUTP; if (r % LD16 < 16) v = s.outp_w[(r / LD16) * 16 + r % LD16]; }
            else if (k < K2_N2W) v = s.outp_b[k - K2_OUTPB];
            else if (k < K2_N2B) v = s.norm2_w[k - K2_N2W];
            else if (k < K2_F1) v = s.norm2_b[k - K2_N2B];
            else if (k < K2_F1B) { r = k - K2_F1; if (r % LD16 < 16) v = s.ffn1_w[(r / LD16) * 16 + r % LD16]; }
            else if (k < K2_F2) v = s.ffn1_b[k - K2_F1B];
            else if (k < K2_F2B) { r = k - K2_F2; if (r % LD32 < 32) v = s.ffn2_w[(r / LD32) * 32 + r % LD32]; }
            else if (k < K2_N3W) v = s.ffn2_b[k - K2_F2B];
            else if (k < K2_N3B) v = s.norm3_w[k - K2_N3W];
            else if (k < K2_SAO) v = s.norm3_b[k - K2_N3B];
            else if (k < K2_SAOB) { r = k - K2_SAO; if (r % LD16 < 16) v = s.out_proj_w[(r / LD16) * 16 + r % LD16]; }
            else if (k < K2_N1W) v = s.out_proj_b[k - K2_SAOB];
            else if (k < K2_N1B) v = s.norm1_w[k - K2_N1W];
            else if (k < K2_SLOT) v = s.norm1_b[k - K2_N1B];
            else {      // four int8 slot levels per word
                unsigned word = 0;
                for (int e = 0; e < 4; ++e) {
                    const int slot = (k - K2_SLOT) * 4 + e;
                    int m, n;
                    slot_decode(slot >> 6, slot & 63, m, n);
                    word |= (unsigned)((n < LP ? n / P : -1) & 0xff) << (8 * e);
                }
                v = __uint_as_float(word);
            }
        }
        d[i] = v;
    }
}

// optional in-kernel phase stamps (tools/decoder_stamps.py; DPFT_DEC_DBG & 1024): 100 MHz wall clock per (block, slot)
constexpr int STAMP_BLOCKS = 2048, STAMP_SLOTS = 8;
__device__ unsigned long long g_stamps[2][STAMP_BLOCKS * STAMP_SLOTS];
__device__ __forceinline__ void stamp(int on, int kernel, int block, int slot) {
    if (on && block < STAMP_BLOCKS && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == (slot == 0 ? 0 : (blockDim.x >> 6) - 1))
        g_stamps[kernel][block * STAMP_SLOTS + slot] = __builtin_amdgcn_s_memrealtime();
}

// ---------------------------------------------------------------------------------------------------------
// K1: attention scores of one head
// ---------------------------------------------------------------------------------------------------------
constexpr int QC = 50;      // queries per block (25 pairs); 2 * QC must be a multiple of 4 (LDS alignment of Pt)
constexpr int NS = 10;      // key slices (<= SC_W); 25 pairs x 10 slices = 250 of the 256 threads
constexpr int SC_W = 16;    // LDS floats: slice maxima

struct ScoreArgs {
    const float* pi[4];   // packed inference blobs of this iteration
    const float* part;    // later iterations: partial q/k/v rows written by the previous xattn (layout: see scores_block)
    const float* pos;     // (Q,16)
    float* attn;          // (V,Bsa,Q,16) attention output before out_proj
    int Bsa, B, Q, V, nchunk, stamps;
};

// COMPOSED = false: first layer -- the rows are constants of the weights (learned query table + embedding), read as packed
// COMPOSED = true : rows = packed position part + sum over the source views of the partials the previous xattn wrote
template <bool COMPOSED>
__device__ __forceinline__ void scores_block(const ScoreArgs& a, float* sm, int sid) {
    const int Q = a.Q, SL = (Q + NS - 1) / NS;
    f32x4* KV = reinterpret_cast<f32x4*>(sm);                     // [Q + NS] (k0,k1,v0,v1); one pad entry per slice
    float* Ws = sm + 4 * (Q + NS);                                 // SC_W floats
    f32x2* Qs = reinterpret_cast<f32x2*>(Ws + SC_W);               // [QC]
    f32x4* Pt = reinterpret_cast<f32x4*>(Ws + SC_W + 2 * QC);      // [QC][NS] (ref, den, o0, o1); 16-byte aligned
    const int tid = threadIdx.x;
    const int chunk = sid % a.nchunk, h = (sid / a.nchunk) & 7, vb = sid / (a.nchunk * 8);
    const int view = vb / a.Bsa, b = vb - view * a.Bsa, q0 = chunk * QC;
    const float* pi = a.pi[view];
    stamp(a.stamps, 0, sid, 0);
    unsigned* kmax = reinterpret_cast<unsigned*>(Ws);                // [NS] max |k|^2 of the slice's keys (float bits)
    if (tid < NS) kmax[tid] = 0u;
    __syncthreads();
    stamp(a.stamps, 0, sid, 1);
    {
        // q / k / v rows of the head = packed input-independent part (position embedding, biases; in the first layer the
        // whole row: its input is the learned query table) [+ COMPOSED: the V partial projections the previous
        // cross-attention kernel wrote].  Those lines were written by other XCDs a moment ago (first touch = a fabric
        // round trip of ~2 us): all loads of the thread's items are issued before the first one is consumed.
        constexpr int NIT = 2;
        static_assert(QC <= 112, "two items per thread cover Q + QC <= 512");
        f32x4 t4[NIT], p4[NIT][4];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * 256;
            const bool isq = i >= Q;
            const int k = isq ? min(q0 + i - Q, Q - 1) : i;
            if (i < Q + QC) t4[u] = *reinterpret_cast<const f32x4*>(pi + PC_T + ((size_t)h * Q + k) * 8 + (isq ? 0 : 4));
            if (COMPOSED && i < Q + QC) {
                // partials: kv (V targets,B,8 heads,V sources,Q,4) then q (same shape, q0 q1 - -): a block's reads
                // (one target, batch element and head; lanes = consecutive keys) are contiguous; one load type for both
                // item kinds (a select between a dwordx2 and a dwordx4 load serialises them)
                const size_t grp = (((size_t)view * a.B + b) * 8 + h) * a.V;
                const float* src = a.part + (isq ? (size_t)a.V * a.B * 8 * a.V * Q * 4 : 0) + (grp * Q + k) * 4;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (v < a.V) p4[u][v] = *reinterpret_cast<const f32x4*>(src + (size_t)v * Q * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * 256;
            if (i >= Q + QC) break;
            const bool isq = i >= Q;
            f32x4 r = t4[u];                                       // q items: lanes 2,3 unused
            if (COMPOSED) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (v < a.V) r += p4[u][v];
            }
            if (isq) Qs[i - Q] = f32x2{r[0], r[1]};
            else {
                KV[i + i / SL] = r;
                atomicMax(kmax + i / SL, __float_as_uint(fmaf(r[0], r[0], r[1] * r[1])));      // >= 0: uint order = float order
            }
        }
    }
    __syncthreads();
    stamp(a.stamps, 0, sid, 2);
    if (tid < (QC / 2) * NS) {
        const int slice = tid / (QC / 2), pair = tid - slice * (QC / 2);
        const f32x2 qa = Qs[2 * pair], qb = Qs[2 * pair + 1];
        const int k0 = slice * SL, k1 = min(Q, k0 + SL);
        const f32x4* kv = KV + k0 + slice;
        // One pass against an upper bound of the slice's scores, ref = |q| max|k| >= q.k (Cauchy-Schwarz; scores live in
        // the exp2 domain): exp2(s - ref) <= 1 cannot overflow, and the partials (ref, den, o) merge exactly like
        // (max, den, o) would.  Should every term of a query underflow (den below 2^-100: the bound is far above the
        // true maximum) the slice is redone with the exact two-pass form.
        const float kn = sqrtf(__uint_as_float(kmax[slice]));
        float ma = sqrtf(fmaf(qa[0], qa[0], qa[1] * qa[1])) * kn, mb = sqrtf(fmaf(qb[0], qb[0], qb[1] * qb[1])) * kn;
        // the two queries of the thread ride in the halves of packed fp32 operations (v_pk_mul / v_pk_fma / v_pk_add)
        const f32x2 q0 = {qa[0], qb[0]}, q1 = {qa[1], qb[1]}, ref = {ma, mb};
        f32x2 dd = {0.f, 0.f}, oo0 = {0.f, 0.f}, oo1 = {0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < k1 - k0; ++k) {
            const f32x4 e = kv[k];
            const f32x2 sc = q0 * e[0] + q1 * e[1] - ref;
            const f32x2 pp = {__builtin_amdgcn_exp2f(sc[0]), __builtin_amdgcn_exp2f(sc[1])};
            dd += pp;
            oo0 += pp * e[2];
            oo1 += pp * e[3];
        }
        float da = dd[0], db = dd[1], a0 = oo0[0], a1 = oo1[0], b0 = oo0[1], b1 = oo1[1];
        if ((da < 0x1p-100f || db < 0x1p-100f) && k1 > k0) {
            ma = mb = -INFINITY;
            for (int k = 0; k < k1 - k0; ++k) {
                const f32x2 kk = *reinterpret_cast<const f32x2*>(kv + k);
                ma = fmaxf(ma, fmaf(qa[1], kk[1], qa[0] * kk[0]));
                mb = fmaxf(mb, fmaf(qb[1], kk[1], qb[0] * kk[0]));
            }
            da = db = a0 = a1 = b0 = b1 = 0.f;
            for (int k = 0; k < k1 - k0; ++k) {
                const f32x4 e = kv[k];
                const float pa = __builtin_amdgcn_exp2f(fmaf(qa[1], e[1], qa[0] * e[0]) - ma);
                const float pb = __builtin_amdgcn_exp2f(fmaf(qb[1], e[1], qb[0] * e[0]) - mb);
                da += pa; db += pb;
                a0 = fmaf(pa, e[2], a0); a1 = fmaf(pa, e[3], a1);
                b0 = fmaf(pb, e[2], b0); b1 = fmaf(pb, e[3], b1);
            }
        }
        if (k1 <= k0) ma = mb = -INFINITY;                 // empty slice (fewer keys than slices)
        Pt[(2 * pair) * NS + slice] = f32x4{ma, da, a0, a1};
        Pt[(2 * pair + 1) * NS + slice] = f32x4{mb, db, b0, b1};
    }
    __syncthreads();
    stamp(a.stamps, 0, sid, 3);
    if (tid < QC && q0 + tid < Q) {
        float mm = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) mm = fmaxf(mm, Pt[tid * NS + s2][0]);
        float den = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const f32x4 part = Pt[tid * NS + s2];
            const float cf = part[0] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(part[0] - mm);     // empty slice
            den = fmaf(part[1], cf, den);
            o0 = fmaf(part[2], cf, o0);
            o1 = fmaf(part[3], cf, o1);
        }
        float* dst = a.attn + (((size_t)view * a.Bsa + b) * Q + q0 + tid) * DC + 2 * h;
        *reinterpret_cast<f32x2*>(dst) = f32x2{o0 / den, o1 / den};
    }
    stamp(a.stamps, 0, sid, 4);
}

// ---------------------------------------------------------------------------------------------------------
// K2: deformable cross attention + FFN of R query rows of one view
// ---------------------------------------------------------------------------------------------------------
constexpr int PYR_L = 5;      // levels the fused inference kernels carry in their arguments (every config: 5)
struct Pyr5 {
    const float* level[PYR_L];
    int H[PYR_L], W[PYR_L];
    int L;
};
// `transformation.any()` per view (mpfusion.py:647) left to the device (dpft_decoder_fwd.has_t < 0 -> flag[v] < 0): an
// extra block of the first launch evaluates it and leaves the result in the 16 ints in FRONT of y3 (a gap of the work
// buffer), where the kernels that need it find it through a pointer they already carry -- the argument structs do not
// grow (three more pointers cost every kernel of the forward 0.3-0.5 us).
__device__ __forceinline__ int* dev_flags(const float* y3) { return reinterpret_cast<int*>(const_cast<float*>(y3)) - 16; }

struct XattnArgs {
    Pyr5 pyr[4];
    const float* pi[4];         // packed inference blobs
    const float* attn;          // (V,Bsa,Q,16)
    const float* query;         // (B,Q,16) or (Q,16): residual of the self-attention block
    const float* pos;           // (Q,16)
    const float* refs;          // (V,B,Q,2) or NULL: project prev_center here (first iteration)
    const float* prev_center;   // (B,Q,3)
    const float* T[4];          // (B,4,4)
    const float* Pm[4];         // (B,prow,4)
    const int64_t* shape[4];    // (B,2) = H, W
    int sstride;                // int64 elements between shape rows
    int prow[4], flag[4], P[4];
    float* y3;                  // (V,B,Q,16)
    float* part;                // (V targets,B,Q,V sources,64) next layer's partial q/k/v rows, or NULL (last layer)
    int B, Q, V, Bsa;
    long qstride;
    int dbg;
    int* zero_tickets;          // first launch of a forward with a fused last launch: block (x, 0) clears ticket x
};

// reference point of one view: cartesian center -> (optional T + spherical) -> projection P -> normalised, clamped
// (mpfusion.py:617-696)
__device__ __forceinline__ void reference_point(float cx, float cy, float cz, int flag, const float* T, const float* Pm,
                                                float Hs, float Ws, float& u, float& vv) {
    const float RAD2DEG = 57.29577951308232f;
    float p0 = cx, p1 = cy, p2 = cz;
    if (flag) {
        const float tx = T[0] * cx + T[1] * cy + T[2] * cz + T[3];
        const float ty = T[4] * cx + T[5] * cy + T[6] * cz + T[7];
        const float tz = T[8] * cx + T[9] * cy + T[10] * cz + T[11];
        const float r = sqrtf(tx * tx + ty * ty + tz * tz);
        p0 = r;
        p1 = atan2f(ty, tx) * RAD2DEG;
        p2 = asinf(r != 0.f ? tz / r : 0.f) * RAD2DEG;
    }
    u = Pm[0] * p0 + Pm[1] * p1 + Pm[2] * p2 + Pm[3];
    vv = Pm[4] * p0 + Pm[5] * p1 + Pm[6] * p2 + Pm[7];
    const float wq = Pm[8] * p0 + Pm[9] * p1 + Pm[10] * p2 + Pm[11];
    if (wq != 0.f) { u /= wq; vv /= wq; }
    u = fminf(fmaxf(u / Ws, 0.f), 1.f);
    vv = fminf(fmaxf(vv / Hs, 0.f), 1.f);
}

__device__ __forceinline__ float rdlane(float v, int k) {     // k wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}

typedef __attribute__((address_space(1))) char gbytes;          // explicit global address space: addresses rebuilt from
typedef __attribute__((address_space(1))) f32x4 gf32x4;        // integers would otherwise become flat loads
// mish(x) = x tanh(softplus(x)) with tanh(log(1 + e^x)) = t / (t + 2), t = e^x (e^x + 2): one exp and one division
// instead of log1p(exp) + tanh; for x > 20 the ratio is 1 in fp32 (torch switches softplus to x there as well).
__device__ __forceinline__ float mish_fast(float x) {
    const float n = __expf(fminf(x, 20.f));
    const float t = n * (n + 2.f);
    return x * (t * __builtin_amdgcn_rcpf(t + 2.f));
}

// ---------------------------------------------------------------------------------------------------------
// K3: view reduction + heads + next reference points, one wave per (b, q)
// ---------------------------------------------------------------------------------------------------------
struct HeadArgs {
    const float* y3;            // (V,B,Q,16)
    const float* ph;            // packed head blob
    const float* prev_center;   // (B,Q,3)
    const float* T[4];
    const float* Pm[4];
    const int64_t* shape[4];
    int sstride;
    int prow[4], flag[4];
    float* query_out;           // (B,Q,16)
    float *center, *size, *angle, *cls;
    float* refs_out;            // (V,B,Q,2) reference points of the NEW center, or NULL (last iteration)
    int B, Q, V, ncls, stamps;
};

// `hsw`: 128 floats of LDS scratch of this wave.  SC1: y3 was written by OTHER workgroups of the SAME launch (the fused last
// launch, below) -- read with agent-scope loads.
constexpr int DEC_SH_BLOCKS = 5;      // resident 256-thread blocks per CU decoder_scores_head_kernel's register budget is sized for
// STAGED: one 16-float weight set in registers at a time (~20 registers instead of 112; one more L2 round trip per set)
template <bool SC1, bool STAGED = SC1>
__device__ __forceinline__ void reduce_head_rows(const HeadArgs& a, float* hsw, int bq, int hid) {
    const int lane = threadIdx.x & 63;
    if (bq >= a.B * a.Q) return;
    stamp(a.stamps, 0, 1024 + hid, 0);
    const int b = bq / a.Q;
    const int c = lane & 15, v2 = lane >> 4;
    // transformation.any() of view `lane` (device-side form): requested now, used at the very end of the block
    int t_flag = 0;
    if (a.refs_out && lane < a.V) {
        const int hf = lane == 0 ? a.flag[0] : lane == 1 ? a.flag[1] : lane == 2 ? a.flag[2] : a.flag[3];
        t_flag = hf < 0 ? dev_flags(a.y3)[lane] : hf;
    }
    const float* ph = a.ph;
    float yv = 0.f;
    if (v2 < a.V) {
        const float* src = a.y3 + ((size_t)v2 * a.B * a.Q + bq) * DC + c;
        yv = SC1 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
    }
    // !SC1: all weights of the wave (input-independent, L2) are requested BEFORE the first use of y3, which was written by
    // other XCDs a moment ago: one memory round trip for the whole block instead of one per MLP layer (112 registers).
    // SC1 (tail of the cross-attention kernel, 80-register budget): one 16-float weight set at a time.
    float wr[STAGED ? 1 : 4][DC], wh[STAGED ? 1 : 3][DC];
    if constexpr (!STAGED) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int k = 0; k < DC; ++k) wr[v][k] = v < a.V ? ph[PH_RED_WT + (v * DC + k) * DC + c] : 0.f;
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int k = 0; k < DC; ++k) wh[l][k] = ph[PH_W + (l * DC + k) * 64 + lane];
    }
    auto load_wh = [&](int l) {
        if constexpr (STAGED) {
#pragma unroll
            for (int k = 0; k < DC; ++k) wh[0][k] = ph[PH_W + (l * DC + k) * 64 + lane];
        }
    };
    // view reduction: queries.view(B,N,C*V) is channel-major / view-minor (mpfusion.py:436-438)
    float x = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if constexpr (STAGED) {
            if (v >= a.V) break;
#pragma unroll
            for (int k = 0; k < DC; ++k) wr[0][k] = ph[PH_RED_WT + (v * DC + k) * DC + c];
        }
#pragma unroll
        for (int k = 0; k < DC; ++k) x = fmaf(wr[STAGED ? 0 : v][k], rdlane(yv, v * 16 + k), x);
        if constexpr (STAGED) asm volatile("" ::: "memory");
    }
    if (lane < 16) a.query_out[(size_t)bq * DC + lane] = x;
    stamp(a.stamps, 0, 1024 + hid, 1);
    // heads (heads/detection.py:252-275): branch g = lane / 16 (center, size, angle, class), row o = lane % 16
    const int g = lane >> 4, o = lane & 15;
    float t = 0.f;
    load_wh(0);
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(wh[0][k], rdlane(x, k), t);
    hsw[lane] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
    load_wh(1);
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(wh[STAGED ? 0 : 1][k], hsw[g * 16 + k], t);
    hsw[64 + lane] = fmaxf(t, 0.f);
    __builtin_amdgcn_wave_barrier();
    t = 0.f;
    load_wh(2);
#pragma unroll
    for (int k = 0; k < DC; ++k) t = fmaf(wh[STAGED ? 0 : 2][k], hsw[64 + g * 16 + k], t);
    const int nout = g == 0 ? 3 : (g == 1 ? 3 : (g == 2 ? 2 : a.ncls));
    float cen = 0.f;
    if (o < nout) {
        if (g == 0) { cen = t + a.prev_center[bq * 3 + o]; a.center[bq * 3 + o] = cen; }
        else if (g == 1) a.size[bq * 3 + o] = fmaxf(t, 0.f);
        else if (g == 2) a.angle[bq * 2 + o] = tanhf(t);
        else a.cls[bq * a.ncls + o] = t;
    }
    stamp(a.stamps, 0, 1024 + hid, 2);
    if (a.refs_out) {       // reference points of the new center for the next iteration: lane = view
        const float cx = rdlane(cen, 0), cy = rdlane(cen, 1), cz = rdlane(cen, 2);
        if (lane < a.V) {
            float u, vv;
            reference_point(cx, cy, cz, t_flag, a.T[lane] ? a.T[lane] + (size_t)b * 16 : nullptr,
                            a.Pm[lane] + (size_t)b * a.prow[lane] * 4, (float)a.shape[lane][b * a.sstride + 0],
                            (float)a.shape[lane][b * a.sstride + 1], u, vv);
            *reinterpret_cast<f32x2*>(a.refs_out + ((size_t)lane * a.B * a.Q + bq) * 2) = f32x2{u, vv};
        }
    }
    stamp(a.stamps, 0, 1024 + hid, 3);
}


__device__ __forceinline__ void reduce_head_block(const HeadArgs& a, float (*hs)[2][64], int hid) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    reduce_head_rows<false, DEC_SH_BLOCKS >= 5>(a, &hs[wave][0][0], hid * 4 + wave, hid);
}

constexpr int XW_FLOATS = 384;      // per-wave scratch: wq float4[64] | addr uint2[64] (low bits: dw, dh, level); small vectors reuse wq

// LAST (round 4): the final iteration's launch also runs the view reduction + detection heads of its rows -- the three
// view blocks of a query group take a ticket when their y3 rows are out (agent-scope stores), whoever draws the last one
// runs reduce_head_rows for the group's R queries.  The forward's trailing head launch disappears.  `tickets`: one int per
// blockIdx.x, zeroed by the FIRST cross-attention launch of the forward (a.zero_tickets).
template <int R, bool LAST>
__device__ __forceinline__ void xattn_body(const XattnArgs& a, const HeadArgs* hap, int* tickets) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ __attribute__((aligned(16))) int lvl_tab[DPFT_MAX_LEVELS][4];      // ptr lo, ptr hi, H, W
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int view = blockIdx.y;
    // transformation.any() of this view (first iteration only: later ones read the reference points the heads wrote).
    // Requested first -- its round trip hides behind the staging; selects instead of a runtime index into the kernel
    // arguments (which would move them to scratch).
    int t_flag = 0;
    if (!a.refs) {
        const int hf = view == 0 ? a.flag[0] : view == 1 ? a.flag[1] : view == 2 ? a.flag[2] : a.flag[3];
        if (hf < 0) {
            // left to the device: every wave evaluates it for itself (B * 16 floats, one load + a ballot -- round 4: no
            // launch in front of the first cross-attention is left to do it); block 0 of the view leaves it where the
            // head blocks of the later launches look for it
            const float* Tv = view == 0 ? a.T[0] : view == 1 ? a.T[1] : view == 2 ? a.T[2] : a.T[3];
            bool nz = false;
            if (Tv)
                for (int i = lane; i < a.B * 16; i += 64) nz |= Tv[i] != 0.f;
            t_flag = __ballot(nz) != 0ull ? 1 : 0;
            if (blockIdx.x == 0 && tid == 0) dev_flags(a.y3)[view] = t_flag;
        } else {
            t_flag = hf;
        }
    }
    const float* __restrict__ img = a.pi[view] + PI_K2;
    // stage the view's weights (the row's own loads are issued first so that their latency hides behind this)
    const int sblk = blockIdx.y * gridDim.x + blockIdx.x, son = (a.dbg & 1024) && a.part;
    stamp(son, 1, sblk, 0);
    const int bq = blockIdx.x * R + wave;
    const bool live = bq < a.B * a.Q;
    const int bqc = live ? bq : a.B * a.Q - 1;
    const int b = bqc / a.Q, q = bqc - b * a.Q;
    const int c = lane & 15;
    const Pyr5& pyr = a.pyr[view];
    const int L = pyr.L, P = a.P[view], LP = L * P;
    const float av = a.attn[(((size_t)view * a.Bsa + (a.Bsa == 1 ? 0 : b)) * a.Q + q) * DC + c];
    const float xres = a.query[(size_t)b * a.qstride + (size_t)q * DC + c];
    const float posc = a.pos[(size_t)q * DC + c];
    {
        constexpr int NV = K2_FLOATS / 4, PER = (NV + R * 64 - 1) / (R * 64);
        f32x4 stage[PER];                      // all loads in flight before the first LDS store
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * R * 64 < NV) stage[u] = reinterpret_cast<const f32x4*>(img)[tid + u * R * 64];
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tid + u * R * 64 < NV) reinterpret_cast<f32x4*>(sm)[tid + u * R * 64] = stage[u];
    }
    if (tid < L) {
        const uint64_t p = reinterpret_cast<uint64_t>(pyr.level[tid]);
        lvl_tab[tid][0] = (int)(uint32_t)p; lvl_tab[tid][1] = (int)(uint32_t)(p >> 32);
        lvl_tab[tid][2] = pyr.H[tid]; lvl_tab[tid][3] = pyr.W[tid];
    }
    float rx, ry;
    if (a.refs) {
        const f32x2 r2 = *reinterpret_cast<const f32x2*>(a.refs + ((size_t)view * a.B * a.Q + bqc) * 2);
        rx = r2[0]; ry = r2[1];
    } else {
        const float* pc = a.prev_center + (size_t)bqc * 3;
        reference_point(pc[0], pc[1], pc[2], t_flag, a.T[view] ? a.T[view] + (size_t)b * 16 : nullptr,
                        a.Pm[view] + (size_t)b * a.prow[view] * 4, (float)a.shape[view][b * a.sstride + 0],
                        (float)a.shape[view][b * a.sstride + 1], rx, ry);
    }
    if (a.zero_tickets && blockIdx.y == 0 && tid == 0) a.zero_tickets[blockIdx.x] = 0;
    __syncthreads();
    if (!live) {
        if constexpr (LAST) {      // (waves without a row still take part in the ticket hand-over of their block)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __syncthreads();
        }
        return;
    }
    stamp(son, 1, sblk, 1);
    float* ws = sm + K2_FLOATS + wave * XW_FLOATS;
    f32x4* wq = reinterpret_cast<f32x4*>(ws);
    uint2* adr = reinterpret_cast<uint2*>(ws + 256);
    float* vec = ws + 64;       // [16] head outputs (written after the last gather round: the gather scratch is dead)
    // ---- self-attention epilogue: out_proj + residual + LayerNorm1 (mpfusion.py:142-148) ----
    // lane = (output c, k-slice kg): 4 of the 16 products per lane, slices merged with two permlane swaps
    const int kg = lane >> 4;
    float* xs = ws;                                   // 64 floats of wave scratch (the gather scratch is not live yet)
    if (lane < 16) xs[lane] = av;
    __builtin_amdgcn_wave_barrier();
    float y1c;
    {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sm + K2_SAO + c * LD16 + 4 * kg);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + 4 * kg);
        y1c = slices4_sum(w[0] * x4[0] + w[1] * x4[1] + w[2] * x4[2] + w[3] * x4[3]) + sm[K2_SAOB + c];
    }
    y1c = layernorm16(y1c + xres, sm[K2_N1W + c], sm[K2_N1B + c]);
    __builtin_amdgcn_wave_barrier();
    if (lane < 16) xs[lane] = y1c + posc;             // qp = y1 + pos: the GEMV's input, read back as LDS broadcasts
    __builtin_amdgcn_wave_barrier();
    // ---- sampling offsets + attention logits of this lane's 3 slots (ms_deform_attn.py:177-182) ----
    int idx[3];
    f32x2 off[3];
    float lg[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        idx[s] = min(s * 64 + lane, NSLOT - 1);
        off[s] = *reinterpret_cast<const f32x2*>(sm + K2_OFF + (16 * NSLOT + idx[s]) * 2);
        lg[s] = sm[K2_LOG + 16 * NSLOT + idx[s]];
    }
#pragma unroll 1
    for (int k4 = 0; k4 < DC; k4 += 4) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xs + k4);      // same address in every lane: broadcast
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k4 + e;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const f32x2 w = *reinterpret_cast<const f32x2*>(sm + K2_OFF + (k * NSLOT + idx[s]) * 2);
                off[s] += w * x4[e];
                lg[s] = fmaf(sm[K2_LOG + k * NSLOT + idx[s]], x4[e], lg[s]);
            }
        }
    }
    // ---- softmax over the L*P slots of the head (lanes with equal lane & 7; bits 3..5 + the 3 slots) ----
    int lvl[3];                                        // pyramid level of the slot, -1 = unused
    bool ok[3];
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        lvl[s] = reinterpret_cast<const signed char*>(sm + K2_SLOT)[idx[s]];
        ok[s] = lvl[s] >= 0 && (s < 2 || lane < 32);
        if (ok[s]) mx = fmaxf(mx, lg[s]);
    }
    mx = fmaxf(mx, dpp_mov<DPP_ROR8>(mx)); mx = xor16_combine(mx, OpMax()); mx = xor32_combine(mx, OpMax());
    float aw[3], den = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) { aw[s] = ok[s] ? __expf(lg[s] - mx) : 0.f; den += aw[s]; }
    den += dpp_mov<DPP_ROR8>(den); den = xor16_combine(den, OpAdd()); den = xor32_combine(den, OpAdd());
    const float inv_den = __builtin_amdgcn_rcpf(den);
    stamp(son, 1, sblk, 2);
    // ---- sample-then-project: producer lanes write (weights, corner address), 4-lane pixel groups gather ----
    const int g = lane >> 2, j = lane & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ms = 0.f;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        {
            const int l = ok[s] ? lvl[s] : 0;
            const int4 lt = *reinterpret_cast<const int4*>(lvl_tab[l]);
            const int H = lt.z, W = lt.w;
            const float a_w = aw[s] * inv_den;
            const float lx = rx + off[s][0] * __builtin_amdgcn_rcpf((float)W), ly = ry + off[s][1] * __builtin_amdgcn_rcpf((float)H);
            const float h_im = ly * H - 0.5f, w_im = lx * W - 0.5f;
            const bool in = ok[s] && h_im > -1 && w_im > -1 && h_im < H && w_im < W;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h_lo = (int)hf, w_lo = (int)wf, h_hi = h_lo + 1, w_hi = w_lo + 1;
            const float lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
            const bool k1 = in && h_lo >= 0 && w_lo >= 0, k2 = in && h_lo >= 0 && w_hi <= W - 1;
            const bool k3 = in && h_hi <= H - 1 && w_lo >= 0, k4 = in && h_hi <= H - 1 && w_hi <= W - 1;
            const int hl = min(max(h_lo, 0), H - 1), hh_ = min(max(h_hi, 0), H - 1);
            const int wl = min(max(w_lo, 0), W - 1), wh_ = min(max(w_hi, 0), W - 1);
            wq[lane] = f32x4{k1 ? a_w * hh * hw : 0.f, k2 ? a_w * hh * lw : 0.f, k3 ? a_w * lh * hw : 0.f,
                             k4 ? a_w * lh * lw : 0.f};
            const uint64_t base = ((uint64_t)(uint32_t)lt.y << 32 | (uint32_t)lt.x)
                                  + ((uint64_t)((int64_t)b * H + hl) * W + wl) * (DC * 4)
                                  + ((wh_ - wl) | (hh_ - hl) << 1 | l << 2);      // 64-byte aligned: low bits carry dw, dh, level
            adr[lane] = uint2{(uint32_t)base, (uint32_t)(base >> 32)};
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int k = 0; k < (s < 2 ? 4 : 2); k += 2) {      // two rounds (8 x 16-byte gathers per lane) in flight
            f32x4 w4[2], v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int src = (k + u) * 16 + g;
                w4[u] = wq[src];
                const uint2 ad = adr[src];
                const int pitch = (ad.x & 2) ? lvl_tab[(ad.x >> 2) & 7][3] * (DC * 4) : 0;
                const int cstep = (ad.x & 1) ? DC * 4 : 0;
                const gbytes* A = reinterpret_cast<const gbytes*>(((uint64_t)ad.y << 32) | (ad.x & ~63u)) + j * 16;
                v[u][0] = *reinterpret_cast<const gf32x4*>(A);          // global_load_dwordx4 (not flat)
                v[u][1] = *reinterpret_cast<const gf32x4*>(A + cstep);
                v[u][2] = *reinterpret_cast<const gf32x4*>(A + pitch);
                v[u][3] = *reinterpret_cast<const gf32x4*>(A + pitch + cstep);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc += w4[u][0] * v[u][0]; acc += w4[u][1] * v[u][1]; acc += w4[u][2] * v[u][2]; acc += w4[u][3] * v[u][3];
                ms += (w4[u][0] + w4[u][1]) + (w4[u][2] + w4[u][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    stamp(son, 1, sblk, 3);
    // value_proj on the sampled features (+ bias * in-bounds mass), head m = g & 7 -> channels 2m, 2m+1
    {
        const int m = g & 7;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sm + K2_VALW + (2 * m) * DC + 4 * j);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sm + K2_VALW + (2 * m + 1) * DC + 4 * j);
        float o0 = w0[0] * acc[0] + w0[1] * acc[1] + w0[2] * acc[2] + w0[3] * acc[3];
        float o1 = w1[0] * acc[0] + w1[1] * acc[1] + w1[2] * acc[2] + w1[3] * acc[3];
        o0 += dpp_mov<DPP_XOR1>(o0); o0 += dpp_mov<DPP_XOR2>(o0); o0 = xor32_combine(o0, OpAdd());
        o1 += dpp_mov<DPP_XOR1>(o1); o1 += dpp_mov<DPP_XOR2>(o1); o1 = xor32_combine(o1, OpAdd());
        ms = xor32_combine(ms, OpAdd());
        if (lane < 32 && j == 0) {
            vec[2 * m] = o0 + sm[K2_VALB + 2 * m] * ms;
            vec[2 * m + 1] = o1 + sm[K2_VALB + 2 * m + 1] * ms;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- output_proj + residual + LayerNorm2: lane = (output c, k-slice kg) ----
    float y2;
    {
        const f32x4 w = *reinterpret_cast<const f32x4*>(sm + K2_OUTP + c * LD16 + 4 * kg);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(vec + 4 * kg);
        const float vo = slices4_sum(w[0] * x4[0] + w[1] * x4[1] + w[2] * x4[2] + w[3] * x4[3]) + sm[K2_OUTPB + c];
        y2 = layernorm16(vo + y1c, sm[K2_N2W + c], sm[K2_N2B + c]);
    }
    // ---- FFN: 16 -> 32 (Mish) -> 16, residual, LayerNorm3 ----
    float* t16 = ws;            // [16] y2, later y3   (the gather scratch is dead now)
    float* t32 = ws + 16;       // [32] hidden activations
    if (lane < 16) t16[lane] = y2;
    __builtin_amdgcn_wave_barrier();
    {
        const int jf = lane & 31, k2 = lane >> 5;       // hidden unit, k-slice of 8
        const float* wr = sm + K2_F1 + jf * LD16 + 8 * k2;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(t16 + 8 * k2), x1 = *reinterpret_cast<const f32x4*>(t16 + 8 * k2 + 4);
        float hsum = (w0[0] * x0[0] + w0[1] * x0[1] + w0[2] * x0[2] + w0[3] * x0[3])
                   + (w1[0] * x1[0] + w1[1] * x1[1] + w1[2] * x1[2] + w1[3] * x1[3]);
        hsum = xor32_combine(hsum, OpAdd()) + sm[K2_F1B + jf];
        if (lane < 32) t32[lane] = mish_fast(hsum);
    }
    __builtin_amdgcn_wave_barrier();
    float y3;
    {
        const float* wr = sm + K2_F2 + c * LD32 + 8 * kg;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(t32 + 8 * kg), x1 = *reinterpret_cast<const f32x4*>(t32 + 8 * kg + 4);
        float f = (w0[0] * x0[0] + w0[1] * x0[1] + w0[2] * x0[2] + w0[3] * x0[3])
                + (w1[0] * x1[0] + w1[1] * x1[1] + w1[2] * x1[2] + w1[3] * x1[3]);
        f = slices4_sum(f) + sm[K2_F2B + c];
        y3 = layernorm16(f + y2, sm[K2_N3W + c], sm[K2_N3B + c]);
    }
    if constexpr (LAST) {
        if (lane < 16) __hip_atomic_store(a.y3 + ((size_t)view * a.B * a.Q + bq) * DC + lane, y3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(sm);      // the staged weights are dead: every wave of the block is past its FFN
        if (tid == 0) *flag = __hip_atomic_fetch_add(tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag == a.V - 1) reduce_head_rows<true>(*hap, ws, bq, blockIdx.x);
        return;
    }
    if (lane < 16) a.y3[((size_t)view * a.B * a.Q + bq) * DC + lane] = y3;
    stamp(son, 1, sblk, 4);
    if (a.part) {       // this view's share of the next layer's q/k/v rows of every target view (lane = output)
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) t16[lane] = y3;
        __builtin_amdgcn_wave_barrier();
        f32x4 y4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y4[i] = *reinterpret_cast<const f32x4*>(t16 + 4 * i);      // broadcast reads
        const float* nx = a.pi[view] + PC_NX + lane;
        for (int tv = 0; tv < a.V; ++tv) {
            float o = 0.f;
#pragma unroll
            for (int k = 0; k < DC; ++k) o = fmaf(nx[(tv * 16 + k) * 64], y4[k >> 2][k & 3], o);
            const int hh = lane >> 3, e = lane & 7;
            const size_t grp = ((((size_t)tv * a.B + b) * 8 + hh) * a.V + view) * a.Q + q;
            if (e >= 4) a.part[grp * 4 + (e - 4)] = o;
            else if (e < 2) a.part[(size_t)a.V * a.B * 8 * a.V * a.Q * 4 + grp * 4 + e] = o;
        }
    }
    stamp(son, 1, sblk, 5);
}

template <int R>
__global__ __launch_bounds__(R * 64, 6) void decoder_xattn_kernel(XattnArgs a) {
    xattn_body<R, false>(a, nullptr, nullptr);
}
template <int R>
__global__ __launch_bounds__(R * 64, 6) void decoder_xattn_last_kernel(XattnArgs a, HeadArgs ha, int* tickets) {
    xattn_body<R, true>(a, &ha, tickets);
}


// One launch = the score blocks of iteration it (blocks [0, n_score)) + the reduction / head blocks of iteration it-1
// (blocks [n_score, ...)): both only depend on the previous cross-attention kernel, so the head MLPs' latency chain
// hides behind the scores instead of sitting between two kernel boundaries.
__global__ __launch_bounds__(256, DEC_SH_BLOCKS) void decoder_scores_head_kernel(ScoreArgs sa, HeadArgs ha, int n_score, int composed) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float hs[4][2][64];
    // head blocks FIRST (round 4): their latency chain (first touch of y3, three dependent MLP layers) starts with the launch
    // and hides behind the score blocks; dispatched last they were the kernel's tail (13.3 us for 8.5 us of score blocks)
    const int n_head_blocks = (int)gridDim.x - n_score;
    const int bid = (int)blockIdx.x < n_head_blocks ? n_score + (int)blockIdx.x : (int)blockIdx.x - n_head_blocks;
    if (bid < n_score) {
        if (composed == 1) scores_block<true>(sa, sm, bid);
        else scores_block<false>(sa, sm, bid);
    } else {
        reduce_head_block(ha, hs, bid - n_score);
    }
}

}  // namespace dpft

using namespace dpft;

extern "C" int64_t dpft_decoder_packed_view_floats(void) { return PV_FLOATS; }
extern "C" int64_t dpft_decoder_packed_head_floats(void) { return PH_FLOATS; }

extern "C" int dpft_decoder_pack_view_f32(const dpft_decoder_view* view, int32_t L, int32_t P, float* packed,
                                          dpft_stream_t stream) {
    DPFT_REQUIRE(view && packed, "decoder_pack_view: null argument");
    DPFT_REQUIRE(L >= 1 && L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && L * P * DM * 3 <= NOA,
                 "decoder_pack_view: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", L, P);
    const float* const* f = reinterpret_cast<const float* const*>(view);
    for (size_t i = 0; i < sizeof(dpft_decoder_view) / sizeof(float*); ++i)
        DPFT_REQUIRE(f[i], "decoder_pack_view: parameter pointer %d is null", (int)i);
    hipLaunchKernelGGL(pack_view_kernel, dim3(cdiv(PV_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, *view,
                       DM * L * P * 2, DM * L * P, packed);
    return check_launch("decoder_pack_view");
}

extern "C" int dpft_decoder_pack_views_f32(const dpft_decoder_view* views, int32_t V, const int32_t* L, const int32_t* P,
                                           float* packed, dpft_stream_t stream) {
    DPFT_REQUIRE(views && L && P && packed && V >= 1 && V <= 4, "decoder_pack_views: bad arguments");
    PackViews pv;
    memset(&pv, 0, sizeof(pv));
    for (int v = 0; v < V; ++v) {
        DPFT_REQUIRE(L[v] >= 1 && L[v] <= DPFT_MAX_LEVELS && P[v] >= 1 && P[v] <= 4 && L[v] * P[v] * DM * 3 <= NOA,
                     "decoder_pack_views: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", L[v], P[v]);
        const float* const* f = reinterpret_cast<const float* const*>(views + v);
        for (size_t i = 0; i < sizeof(dpft_decoder_view) / sizeof(float*); ++i)
            DPFT_REQUIRE(f[i], "decoder_pack_views: view %d parameter pointer %d is null", v, (int)i);
        pv.v[v] = views[v];
        pv.n_off[v] = DM * L[v] * P[v] * 2;
        pv.n_att[v] = DM * L[v] * P[v];
        pv.d[v] = packed + (size_t)v * PV_FLOATS;
    }
    hipLaunchKernelGGL(pack_views_kernel, dim3(cdiv(PV_FLOATS, 256), V), dim3(256), 0, (hipStream_t)stream, pv);
    return check_launch("decoder_pack_views");
}

extern "C" int dpft_decoder_pack_head_f32(const float* red_w, const float* const* head_w, int32_t V, int32_t num_classes,
                                          float* packed, dpft_stream_t stream) {
    DPFT_REQUIRE(red_w && head_w && packed && V >= 1 && V <= 4, "decoder_pack_head: bad arguments");
    DPFT_REQUIRE(num_classes >= 1 && num_classes <= 16, "decoder_pack_head: num_classes must be in [1,16]");
    HeadSrc s;
    s.red_w = red_w; s.V = V;
    for (int g = 0; g < 4; ++g)
        for (int k = 0; k < 3; ++k) {
            DPFT_REQUIRE(head_w[g * 3 + k], "decoder_pack_head: head weight %d.%d is null", g, k);
            s.hw[g][k] = head_w[g * 3 + k];
        }
    s.nout[0] = 3; s.nout[1] = 3; s.nout[2] = 2; s.nout[3] = num_classes;
    hipLaunchKernelGGL(pack_head_kernel, dim3(cdiv(PH_FLOATS, 256)), dim3(256), 0, (hipStream_t)stream, s, packed);
    return check_launch("decoder_pack_head");
}

extern "C" int64_t dpft_decoder_packed_infer_floats(int32_t Q) { return pi_floats(Q); }

extern "C" int dpft_decoder_pack_infer_f32(const dpft_decoder_view* view, int32_t L, int32_t P, const float* red_w,
                                           const float* const* next_in_proj_w, int32_t view_index, int32_t V,
                                           const float* pos, const float* query0, int32_t Q, float* packed,
                                           dpft_stream_t stream) {
    DPFT_REQUIRE(view && packed && pos && red_w, "decoder_pack_infer: null argument");
    DPFT_REQUIRE(view_index >= 0 && view_index < V, "decoder_pack_infer: view index out of range");
    DPFT_REQUIRE(L >= 1 && L <= DPFT_MAX_LEVELS && P >= 1 && P <= 4 && L * P <= 20,
                 "decoder_pack_infer: L=%d, P=%d exceed the fused kernel's budget (P <= 4, L*P <= 20)", L, P);
    DPFT_REQUIRE(V >= 1 && V <= 4 && Q >= 1, "decoder_pack_infer: bad V / Q");
    const float* const* f = reinterpret_cast<const float* const*>(view);
    for (size_t i = 0; i < sizeof(dpft_decoder_view) / sizeof(float*); ++i)
        DPFT_REQUIRE(f[i], "decoder_pack_infer: parameter pointer %d is null", (int)i);
    NextInProj nx;
    for (int v = 0; v < 4; ++v) nx.w[v] = (next_in_proj_w && v < V) ? next_in_proj_w[v] : nullptr;
    hipLaunchKernelGGL(pack_infer_kernel, dim3(cdiv(pi_floats(Q), 256)), dim3(256), 0, (hipStream_t)stream, *view, L, P,
                       red_w, nx, view_index, V, pos, query0, Q, packed);
    return check_launch("decoder_pack_infer");
}

constexpr int XR = 7;      // query rows (waves) per decoder_xattn_kernel block: 3 blocks of 51 KB LDS per CU

// Whole IMPFusion forward from ONE call: 2 launches per iteration + 1, nothing else on the host:
//   [scores(0)] [xattn(0)] [scores(1) | heads(0)] [xattn(1)] ... [scores(I-1) | heads(I-2)] [xattn(I-1)] [heads(I-1)]
extern "C" int dpft_decoder_forward_f32(const dpft_decoder_fwd* d, dpft_stream_t stream) {
    DPFT_REQUIRE(d && d->packed_views && d->packed_heads && d->pyr && d->query0 && d->pos && d->center0 && d->work,
                 "decoder_forward: null argument");
    DPFT_REQUIRE(d->center && d->size && d->angle && d->cls, "decoder_forward: null output");
    const int B = d->B, Q = d->Q, V = d->V;
    DPFT_REQUIRE(B > 0 && Q > 0 && V >= 1 && V <= 4 && d->iters >= 1 && d->iters <= 8, "decoder_forward: bad sizes");
    DPFT_REQUIRE(d->num_classes >= 1 && d->num_classes <= 16, "decoder_forward: num_classes must be in [1,16]");
    const size_t nq = (size_t)B * Q;
    float* w = d->work;
    float* qbuf[2] = {w, w + nq * DC};
    float* attn = w + 2 * nq * DC;
    float* y3 = attn + (size_t)V * nq * DC + 64;      // 64-float gap: the device-side transformation flags sit in front of y3
    float* cbuf[2] = {y3 + (size_t)V * nq * DC, y3 + (size_t)V * nq * DC + nq * 3};
    float* refs = cbuf[1] + nq * 3;
    float* part = refs + (size_t)V * nq * 2;
    int* tickets = reinterpret_cast<int*>(part + (size_t)2 * V * V * nq * 32);      // behind `part` (dpft_decoder_work_floats)
    static const bool fuse_last = getenv("DPFT_DEC_FUSE_LAST") == nullptr || atoi(getenv("DPFT_DEC_FUSE_LAST")) != 0;      // A/B switch
    const int nxb = cdiv((int64_t)nq, XR);
    XattnArgs xa;
    HeadArgs ha;
    ScoreArgs sa;
    memset(&xa, 0, sizeof(xa));
    memset(&ha, 0, sizeof(ha));
    memset(&sa, 0, sizeof(sa));
    { const char* e = getenv("DPFT_DEC_DBG"); xa.dbg = e ? atoi(e) : 0; sa.stamps = ha.stamps = xa.dbg & 1024; }
    for (int v = 0; v < V; ++v) {
        const dpft_pyramid* pyr = d->pyr + v;
        const int P = d->n_points[v];
        DPFT_REQUIRE(pyr->L >= 1 && pyr->L <= PYR_L && P >= 1 && P <= 4 && pyr->L * P <= 20,
                     "decoder_forward: L=%d, P=%d exceed the fused kernel's budget (L <= 5, P <= 4, L*P <= 20)", pyr->L, P);
        xa.pyr[v].L = pyr->L;
        for (int l = 0; l < pyr->L; ++l) {
            DPFT_REQUIRE(pyr->level[l], "decoder_forward: view %d level %d is null", v, l);
            DPFT_REQUIRE((reinterpret_cast<uintptr_t>(pyr->level[l]) & 63) == 0,
                         "decoder_forward: view %d level %d is not 64-byte aligned", v, l);
            xa.pyr[v].level[l] = pyr->level[l]; xa.pyr[v].H[l] = pyr->H[l]; xa.pyr[v].W[l] = pyr->W[l];
        }
        xa.P[v] = P;
        xa.T[v] = ha.T[v] = d->T[v]; xa.Pm[v] = ha.Pm[v] = d->P[v]; xa.shape[v] = ha.shape[v] = d->shape[v];
        xa.prow[v] = ha.prow[v] = d->p_rows[v]; xa.flag[v] = ha.flag[v] = d->has_t[v];
        DPFT_REQUIRE(xa.Pm[v] && xa.shape[v] && (xa.T[v] || d->has_t[v] <= 0) && xa.prow[v] >= 3,
                     "decoder_forward: projection inputs of view %d missing", v);
    }
    // has_t < 0: `transformation.any()` is evaluated on the device by the first cross-attention kernel (dev_flags)
    xa.sstride = ha.sstride = d->shape_stride > 0 ? d->shape_stride : 2;
    xa.attn = attn; xa.pos = d->pos; xa.y3 = y3; xa.B = B; xa.Q = Q; xa.V = V;
    ha.y3 = y3; ha.B = B; ha.Q = Q; ha.V = V; ha.ncls = d->num_classes;
    ha.size = d->size; ha.angle = d->angle; ha.cls = d->cls;
    sa.part = part; sa.pos = d->pos; sa.attn = attn; sa.B = B; sa.Q = Q; sa.V = V;
    sa.nchunk = cdiv(Q, QC);
    const size_t lds1 = (4 * (size_t)(Q + NS) + SC_W + 2 * QC + 4 * QC * NS) * sizeof(float);
    const size_t lds2 = ((size_t)K2_FLOATS + XR * XW_FLOATS) * sizeof(float);
    DPFT_REQUIRE(lds1 <= 62 * 1024, "decoder_forward: %d queries do not fit the LDS of the score kernel", Q);
    const int n_head = cdiv((int64_t)nq, 4);
    const float* query = d->query0;
    const float* center = d->center0;
    for (int it = 0; it <= d->iters; ++it) {
        const bool first = it == 0, after_last = it == d->iters;
        // ---- scores of iteration `it` + reduction / heads of iteration it-1 ----
        int n_score = 0;
        if (!after_last) {
            for (int v = 0; v < 4; ++v)
                sa.pi[v] = xa.pi[v] = v < V ? d->packed_views + (size_t)(it * V + v) * pi_floats(Q) : nullptr;
            // iteration 0: query = the learned (Q,16) table for every batch element -> one batch element of scores
            sa.Bsa = first ? 1 : B;
            n_score = sa.nchunk * DM * V * sa.Bsa;
        }
        if (!first) {
            const bool last = it == d->iters;
            ha.ph = d->packed_heads + (size_t)(it - 1) * PH_FLOATS;
            ha.prev_center = center;
            ha.query_out = qbuf[(it - 1) & 1];
            ha.center = last ? d->center : cbuf[(it - 1) & 1];
            ha.refs_out = last ? nullptr : refs;
        }
        // iteration 0: the self-attention input is the learned query table -- its attention output is a constant of the
        // weights; with d->attn0 (dpft_decoder_attn0_f32, made when the weights are packed) the launch disappears
        const bool skip = first && d->attn0 != nullptr;
        if (!skip) {
            hipLaunchKernelGGL(decoder_scores_head_kernel, dim3(n_score + (first ? 0 : n_head)), dim3(256),
                               after_last ? 0 : lds1, (hipStream_t)stream, sa, ha, n_score, first ? 0 : 1);
            RC(check_launch("decoder_scores_head"));
        }
        if (!first) {
            query = ha.query_out;
            center = ha.center;
        }
        if (after_last) break;
        const bool last_it = it + 1 == d->iters;
        // ---- cross attention + FFN of iteration `it` ----
        xa.query = query; xa.qstride = first ? 0 : (long)Q * DC; xa.Bsa = sa.Bsa;
        xa.attn = skip ? d->attn0 : attn;
        xa.refs = first ? nullptr : refs;
        xa.prev_center = center;
        xa.part = it + 1 < d->iters ? part : nullptr;
        xa.zero_tickets = (fuse_last && first && d->iters > 1) ? tickets : nullptr;
        if (last_it && fuse_last && d->iters > 1) {
            // the heads of the last iteration ride in this launch (decoder_xattn_last_kernel): no trailing launch
            ha.ph = d->packed_heads + (size_t)it * PH_FLOATS;
            ha.prev_center = center;
            ha.query_out = qbuf[it & 1];
            ha.center = d->center;
            ha.refs_out = nullptr;
            hipLaunchKernelGGL(decoder_xattn_last_kernel<XR>, dim3(nxb, V), dim3(XR * 64), lds2, (hipStream_t)stream, xa, ha, tickets);
            RC(check_launch("decoder_xattn (+ heads)"));
            break;
        }
        hipLaunchKernelGGL(decoder_xattn_kernel<XR>, dim3(nxb, V), dim3(XR * 64), lds2, (hipStream_t)stream, xa);
        RC(check_launch("decoder_xattn"));
    }
    return DPFT_OK;
}

// Attention output of iteration 0 for ONE batch element, (V,Q,16): input = the learned query table + the query embedding,
// i.e. a function of the weights alone (mpfusion.py:700-703: `query = self.query` for every sample).  Made once per weight
// version next to the packed blobs; dpft_decoder_fwd.attn0 then replaces the first launch of every forward.
extern "C" int dpft_decoder_attn0_f32(const float* packed_views, const float* pos, int32_t Q, int32_t V, float* attn0,
                                      dpft_stream_t stream) {
    DPFT_REQUIRE(packed_views && pos && attn0 && Q >= 1 && V >= 1 && V <= 4, "decoder_attn0: bad arguments");
    ScoreArgs sa;
    HeadArgs ha;
    memset(&sa, 0, sizeof(sa));
    memset(&ha, 0, sizeof(ha));
    for (int v = 0; v < V; ++v) sa.pi[v] = packed_views + (size_t)v * pi_floats(Q);      // iteration 0's blobs
    sa.pos = pos; sa.attn = attn0; sa.B = 1; sa.Bsa = 1; sa.Q = Q; sa.V = V;
    sa.nchunk = cdiv(Q, QC);
    const size_t lds1 = (4 * (size_t)(Q + NS) + SC_W + 2 * QC + 4 * QC * NS) * sizeof(float);
    DPFT_REQUIRE(lds1 <= 62 * 1024, "decoder_attn0: %d queries do not fit the LDS of the score kernel", Q);
    const int n_score = sa.nchunk * DM * V;
    hipLaunchKernelGGL(decoder_scores_head_kernel, dim3(n_score), dim3(256), lds1, (hipStream_t)stream, sa, ha, n_score, 0);
    return check_launch("decoder_attn0");
}

extern "C" int64_t dpft_decoder_work_floats(int32_t B, int32_t Q, int32_t V) {
    const int64_t nq = (int64_t)B * Q;
    return 2 * nq * DC + 2 * (int64_t)V * nq * DC + 2 * nq * 3 + (int64_t)V * nq * 2 + (int64_t)V * V * nq * 64 + 64      // part: 2 x V*V*nq*8*4
           + cdiv(nq, (int64_t)XR) + 16;                                                                                    // tickets of the fused last launch
}

// debug: copy the phase stamps of the last launches (2 kernels x 2048 blocks x 8 slots of uint64) to host memory
extern "C" int dpft_debug_decoder_stamps(uint64_t* dst) {
    DPFT_REQUIRE(dst, "debug_decoder_stamps: null");
    DPFT_REQUIRE(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 2 * STAMP_BLOCKS * STAMP_SLOTS) == hipSuccess,
                 "debug_decoder_stamps: copy failed");
    return DPFT_OK;
}
