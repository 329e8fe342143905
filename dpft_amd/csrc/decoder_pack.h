// Packed (lane-friendly, transposed) parameter blobs of the fused decoder kernels -- layout shared by
// decoder.hip (inference) and decoder_train_x.hip (training cross-attention + FFN block).
#pragma once
#include "common.h"

namespace dpft {

constexpr int DC = 16, DM = 8, DD = 2, DFF = 32;
constexpr int NOA = 480;   // max offsets (8 heads * L*P * 2) + logits (8 * L*P), L*P <= 20
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- packed view blob (floats) ----
constexpr int PV_IN_W = 0;                       // in_proj_weight (48,16) as is (staged through LDS)
constexpr int PV_IN_B = PV_IN_W + 768;           // 48
constexpr int PV_OUT_WT = PV_IN_B + 48;          // out_proj.weight^T [k][c]
constexpr int PV_OUT_B = PV_OUT_WT + 256;
constexpr int PV_N1_W = PV_OUT_B + 16;
constexpr int PV_N1_B = PV_N1_W + 16;
constexpr int PV_OA_WT = PV_N1_B + 16;           // [16][NOA]: sampling_offsets rows then attention_weights rows, transposed
constexpr int PV_OA_B = PV_OA_WT + 16 * NOA;     // [NOA]
constexpr int PV_VAL_W = PV_OA_B + NOA;          // value_proj.weight (16,16) as is
constexpr int PV_VAL_B = PV_VAL_W + 256;
constexpr int PV_OUTP_WT = PV_VAL_B + 16;        // output_proj.weight^T [k][c]
constexpr int PV_OUTP_B = PV_OUTP_WT + 256;
constexpr int PV_N2_W = PV_OUTP_B + 16;
constexpr int PV_N2_B = PV_N2_W + 16;
constexpr int PV_F1_WT = PV_N2_B + 16;           // ffn1.weight^T [k 16][j 32]
constexpr int PV_F1_B = PV_F1_WT + 512;
constexpr int PV_F2_WT = PV_F1_B + 32;           // ffn2.weight^T [k 32][c 16]
constexpr int PV_F2_B = PV_F2_WT + 512;
constexpr int PV_N3_W = PV_F2_B + 16;
constexpr int PV_N3_B = PV_N3_W + 16;
constexpr int PV_FLOATS = PV_N3_B + 16;
// ---- packed head blob ----
constexpr int PH_RED_WT = 0;                     // [v 4][k 16][o 16]  = reduction_layer.weight[o][k*V + v]
constexpr int PH_W = PH_RED_WT + 4 * 256;        // [layer 3][k 16][branch 4][o 16] (rows >= out_features are 0)
constexpr int PH_FLOATS = PH_W + 3 * 1024;


// mish(x) = x * tanh(softplus(x)), softplus threshold 20 (torch)
__device__ __forceinline__ float mishf(float x) {
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

}  // namespace dpft
