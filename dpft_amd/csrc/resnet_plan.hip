// Native launch plan for the ResNet-50/101/152 body (forward + hand-scheduled backward).
//
// The Python mirror of src/dprt/models/backbones/resnet.py used to issue ~50 (fwd) + ~100 (bwd) C-ABI
// calls per bottleneck from the interpreter; at ~25 us of host time per call the 3 backbones made the
// training step host-bound.  A plan is built once per (architecture, input shape): it fixes every conv
// descriptor, every activation's offset in a caller-owned arena and the order of kernel launches, so a
// whole forward or a whole stage of the backward is ONE call that only enqueues work on the stream
// (no allocation, no synchronisation: hipGraph-capturable).  Parameters and gradient buffers are passed
// as pointer tables in registration order, so gradients land directly in the data-parallel buckets.
#include <algorithm>
#include <vector>

#include "common.h"

namespace dpft {

// bf16 plan (act16 = 2), round 6 experiment (DPFT_PRO16=1): conv2 / conv3 read the RAW bf16 outputs y1 / y2 through the BatchNorm + ReLU
// operand prologue of the bf16 pipelined kernel (conv_pipe.h) instead of materialised a1 / a2 -- two elementwise launches per bottleneck
// leave the forward's chain; the weight gradients, which still want the materialised operand, get it on the side stream right before
// they run.  Measured at bf16 batch 8 (same box, two rounds): 22.75 / 22.51 ms materialising vs 22.81 / 22.70 with the prologues -- the
// register-route GEMMs (and conv3 losing its 256-row tiles) cost what the 66 launches cost.  Off.
static bool pro16_on() {
    static const bool on = getenv("DPFT_PRO16") != nullptr && atoi(getenv("DPFT_PRO16")) != 0;
    return on;
}

struct ConvRef {
    dpft_conv_desc d{};
    int w;  // index into the conv table
    size_t wt = 0;  // float offset (inside the wt region) of this conv's transposed weights during its stage's backward
    size_t w16 = 0; // act16 = 2: float offset (in the arena) of this conv's bf16 shadow weights [K][taps][C]
};

struct BlockPlan {
    ConvRef c1, c2, c3, cd;
    bool has_ds;
    int bn1, bn2, bn3, bnd;
    int layer;
    // arena offsets (floats)
    size_t x, y1, y2, y3, yd, out, mask;   // mask: ReLU byte mask of `out` (1 byte per 4 channels)
    size_t a1 = 0, a2 = 0;      // act16 = 2: materialised relu(bn1(y1)), relu(bn2(y2)) in bf16 (no operand prologues there)
    size_t p1, p2, p3, pd;      // BN blocks [4][K]
    size_t s1, s2, s3, sd;      // stats [tiles][2][K]
    int t1, t2, t3, td, r1, r2, r3, rd;   // stats tiles / tile rows
};

struct ResnetPlan {
    dpft_resnet_desc desc;
    std::vector<BlockPlan> blocks;
    // stem
    ConvRef adj, c0;
    int bn0;
    size_t xa, y0, p0, s0, pool;
    int t0, r0;
    int PH, PW;
    int n_conv, n_bn;
    size_t out_off[4];
    size_t out32_off[4];      // act16: fp32 copies of the stage outputs (what the neck reads)
    int out_shape[4][4];
    size_t fwd_floats;      // activations kept for backward
    size_t bwd_floats;      // backward temporaries (placed after fwd_floats)
    size_t ws_bytes;        // split-K workspace (placed last)
    size_t arena_bytes;
    // backward state
    size_t g_off[2];
    size_t gmax;            // floats of the largest activation gradient
    size_t o_dy, o_da, o_dd, o_wt, o_sums;   // backward scratch offsets (floats)
    std::vector<size_t> bnacc;      // per BatchNorm: float offset of its fused-finalize accumulators [2][K] + ticket (16 floats)
    size_t o_bnacc = 0, bnacc_floats = 0;
    // "sums" form of the train-mode statistics (common.h: BnSumsRef): layers whose BN block has not been written yet in THIS forward
    // (their consumers derive the parameters from the column sums; the batched finalize at the end of the forward writes the blocks)
    struct PendingBn { int bn, K; long long M; size_t bnp; };
    std::vector<char> bn_pending;
    std::vector<PendingBn> pend_list;
    size_t o_dy2;           // second dy buffer: weight gradients run on a side stream while the data path moves on
    int g_cur;
    bool g_valid;
    // frozen BatchNorm (train = 2: a gradient is wanted through an eval-mode body, resnet.py:169-176 accepts
    // FrozenBatchNorm2d): the forward keeps the train path's tensors but normalises with the running statistics, the
    // backward's apply pass drops the batch-statistics terms (dy = gamma invstd d); set by the forward, read by the stages
    bool frozen = false;
    int mode_key = 0;      // conv_mode_key() at build time: tile shapes (statistics tiles, workspace) depend on the compute mode
    // side stream of the weight-gradient GEMMs (they are off the critical path of the backward)
    hipStream_t side = nullptr;
    bool side_owned = true, side_set = false;      // side_set: side is valid (it may be the null stream)
    bool ev_valid[3] = {false, false, false};      // ev_done[slot] has been recorded in THIS stage call
    // hipGraph replay of the launch sequences (dpft_resnet_plan_set_graph): one executable graph per (call kind, pointer
    // arguments); used when the whole call is a single-stream sequence (weight-gradient stream == launch stream)
    bool use_graph = false;
    struct GraphEntry {
        int kind;                    // -1 forward (train), 0..3 backward stage
        const void *x, *arena, *dout, *st;
        uint64_t tables_hash;
        int warm;                    // eager calls so far
        hipGraphExec_t exec;         // null until captured
        bool failed;
    };
    std::vector<GraphEntry> graphs;
    hipEvent_t ev_ready = nullptr, ev_done[3] = {nullptr, nullptr, nullptr}, ev_join = nullptr, ev_wt = nullptr;
    int dyi = 0;
    ~ResnetPlan() {
        if (side_set && side_owned) (void)hipStreamDestroy(side);
        for (auto& g : graphs)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_wt) (void)hipEventDestroy(ev_wt);
        for (auto& e : ev_done)
            if (e) (void)hipEventDestroy(e);
    }
};

static size_t align64(size_t f) { return (f + 63) & ~(size_t)63; }

static dpft_conv_desc mk(int B, int H, int W, int C, int K, int k, int s, int p) {
    dpft_conv_desc d{};      // (zero: fp32 storage, no operand planes)
    d.B = B; d.H = H; d.W = W; d.C = C; d.K = K; d.kh = k; d.kw = k; d.stride = s; d.pad = p;
    d.OH = (H + 2 * p - k) / s + 1;
    d.OW = (W + 2 * p - k) / s + 1;
    return d;
}

static size_t nelem_out(const dpft_conv_desc& d) { return (size_t)d.B * d.OH * d.OW * d.K; }
static size_t nelem_in(const dpft_conv_desc& d) { return (size_t)d.B * d.H * d.W * d.C; }
static size_t nelem_w(const dpft_conv_desc& d) { return (size_t)d.K * d.kh * d.kw * d.C; }

}  // namespace dpft

using namespace dpft;

extern "C" int64_t dpft_resnet_plan_create(const dpft_resnet_desc* desc) {
    if (!desc || desc->B <= 0 || desc->H <= 0 || desc->W <= 0 || desc->n_layers < 1 || desc->n_layers > 4 ||
        desc->act16 < 0 || desc->act16 > 2) {
        set_error("resnet_plan_create: bad descriptor");
        return 0;
    }
    ResnetPlan* p = new ResnetPlan();
    p->desc = *desc;
    p->mode_key = conv_mode_key();
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += align64(floats); return o; };
    int nconv = 0, nbn = 0;
    const int B = desc->B;
    size_t wsb = 0;
    auto track_ws = [&](const dpft_conv_desc& d) {
        int64_t w = dpft_conv2d_workspace_bytes(&d);
        if (w > 0 && (size_t)w > wsb) wsb = (size_t)w;
    };
    size_t gmax = 0;
    auto track_g = [&](size_t n) { if (n > gmax) gmax = n; };
    auto stats_of = [&](const dpft_conv_desc& d, int& tiles, int& rows, bool pro = false) {
        int32_t r = 0;
        tiles = dpft_conv2d_stats_tiles_pro(&d, pro ? 1 : 0, &r);
        rows = r;
        return (size_t)tiles * 2 * d.K;
    };
    // ---- stem ------------------------------------------------------------------------------------
    int H = desc->H, W = desc->W;
    if (desc->in_channels != 3) {
        p->adj.d = mk(B, H, W, desc->in_channels, 3, 1, 1, 0);
        p->adj.w = nconv++;
        p->xa = take(nelem_out(p->adj.d));
        track_ws(p->adj.d);
    } else {
        p->adj.w = -1;
        p->xa = (size_t)-1;
    }
    p->c0.d = mk(B, H, W, 3, 64, 7, 2, 3);
    p->c0.w = nconv++;
    p->bn0 = nbn++;
    track_ws(p->c0.d);
    p->y0 = take(nelem_out(p->c0.d));
    p->p0 = take(4 * 64);
    p->s0 = take(stats_of(p->c0.d, p->t0, p->r0));
    track_g(nelem_out(p->c0.d));
    p->PH = (p->c0.d.OH + 2 - 3) / 2 + 1;
    p->PW = (p->c0.d.OW + 2 - 3) / 2 + 1;
    p->pool = take((size_t)B * p->PH * p->PW * 64);
    track_g((size_t)B * p->PH * p->PW * 64);
    // ---- stages ----------------------------------------------------------------------------------
    int curH = p->PH, curW = p->PW, inC = 64;
    size_t cur = p->pool;
    const int planes_of[4] = {64, 128, 256, 512};
    for (int li = 0; li < desc->n_layers; ++li) {
        const int planes = planes_of[li];
        for (int b = 0; b < desc->depths[li]; ++b) {
            BlockPlan bp;
            const int stride = (li > 0 && b == 0) ? 2 : 1;
            bp.layer = li;
            bp.has_ds = (b == 0);
            bp.x = cur;
            bp.c1.d = mk(B, curH, curW, inC, planes, 1, 1, 0);   bp.c1.w = nconv++;  bp.bn1 = nbn++;
            bp.c2.d = mk(B, curH, curW, planes, planes, 3, stride, 1);  bp.c2.w = nconv++;  bp.bn2 = nbn++;
            const int oh = bp.c2.d.OH, ow = bp.c2.d.OW;
            bp.c3.d = mk(B, oh, ow, planes, planes * 4, 1, 1, 0);  bp.c3.w = nconv++;  bp.bn3 = nbn++;
            if (bp.has_ds) {
                bp.cd.d = mk(B, curH, curW, inC, planes * 4, 1, stride, 0);  bp.cd.w = nconv++;  bp.bnd = nbn++;
            } else {
                bp.cd.w = -1; bp.bnd = -1;
            }
            // bf16 activation storage inside the body; 2: bf16 weights too (LDS-DMA bf16 kernels, no operand prologues)
            bp.c1.d.act16 = bp.c2.d.act16 = bp.c3.d.act16 = desc->act16;
            if (bp.has_ds) bp.cd.d.act16 = desc->act16;
            if (desc->act16 == 2) {
                bp.a1 = take(nelem_out(bp.c1.d) / 2);
                bp.a2 = take(nelem_out(bp.c2.d) / 2);
                bp.c1.w16 = take((nelem_w(bp.c1.d) + 1) / 2);
                bp.c2.w16 = take((nelem_w(bp.c2.d) + 1) / 2);
                bp.c3.w16 = take((nelem_w(bp.c3.d) + 1) / 2);
                if (bp.has_ds) bp.cd.w16 = take((nelem_w(bp.cd.d) + 1) / 2);
            }
            bp.y1 = take(nelem_out(bp.c1.d));  bp.p1 = take(4 * planes);  bp.s1 = take(stats_of(bp.c1.d, bp.t1, bp.r1));
            bp.y2 = take(nelem_out(bp.c2.d));  bp.p2 = take(4 * planes);  bp.s2 = take(stats_of(bp.c2.d, bp.t2, bp.r2, desc->act16 == 2 && pro16_on()));
            bp.y3 = take(nelem_out(bp.c3.d));  bp.p3 = take(4 * planes * 4);  bp.s3 = take(stats_of(bp.c3.d, bp.t3, bp.r3, desc->act16 == 2 && pro16_on()));
            if (bp.has_ds) {
                bp.yd = take(nelem_out(bp.cd.d));  bp.pd = take(4 * planes * 4);  bp.sd = take(stats_of(bp.cd.d, bp.td, bp.rd));
                track_ws(bp.cd.d);
            } else {
                bp.yd = bp.pd = bp.sd = (size_t)-1; bp.td = bp.rd = 0;
            }
            bp.out = take(nelem_out(bp.c3.d));
            bp.mask = take((nelem_out(bp.c3.d) / 4 + 3) / 4);      // ReLU byte mask of the block output: 1 byte per 4 channels
            track_ws(bp.c1.d); track_ws(bp.c2.d); track_ws(bp.c3.d);
            track_g(nelem_in(bp.c1.d)); track_g(nelem_out(bp.c1.d)); track_g(nelem_out(bp.c2.d)); track_g(nelem_out(bp.c3.d));
            p->blocks.push_back(bp);
            cur = bp.out;
            curH = oh; curW = ow; inC = planes * 4;
        }
        p->out_off[li] = cur;
        p->out32_off[li] = desc->act16 ? take(nelem_out(p->blocks.back().c3.d)) : cur;
        p->out_shape[li][0] = B; p->out_shape[li][1] = curH; p->out_shape[li][2] = curW; p->out_shape[li][3] = inC;
    }
    p->n_conv = nconv;
    p->n_bn = nbn;
    {   // accumulators of the fused BatchNorm finalize (conv epilogue): one zero-fill per forward covers them all
        p->bnacc.assign(nbn, 0);
        size_t acc = 0;
        auto add = [&](int bn, int K) { p->bnacc[bn] = acc; acc += align64((size_t)8 * K + 16); };      // (the sums form: [4][K] 64-bit words)
        add(p->bn0, 64);
        for (const BlockPlan& b : p->blocks) {
            add(b.bn1, b.c1.d.K); add(b.bn2, b.c2.d.K); add(b.bn3, b.c3.d.K);
            if (b.has_ds) add(b.bnd, b.cd.d.K);
        }
        p->bnacc_floats = acc;
        p->bn_pending.assign(nbn, 0);
        p->o_bnacc = take(acc);
    }
    p->fwd_floats = off;
    // ---- backward temporaries: two running-gradient buffers + per-block scratch -----------------------
    p->gmax = gmax;
    p->g_off[0] = take(gmax);
    p->g_off[1] = take(gmax);
    // scratch: dy (2 x gmax) + da (gmax) + dyd (gmax) + transposed weights of one whole stage + bn sums.  The
    // transposes of a stage are queued on the side stream at the start of that stage's backward (they only depend on
    // the weights), so they are off the data-gradient chain.
    size_t wmax = nelem_w(p->c0.d);
    for (int li = 0; li < p->desc.n_layers; ++li) {
        size_t acc = 0;
        for (auto& b : p->blocks) {
            if (b.layer != li) continue;
            b.c1.wt = acc; acc += align64(nelem_w(b.c1.d));
            b.c2.wt = acc; acc += align64(nelem_w(b.c2.d));
            b.c3.wt = acc; acc += align64(nelem_w(b.c3.d));
            if (b.has_ds) { b.cd.wt = acc; acc += align64(nelem_w(b.cd.d)); }
        }
        wmax = std::max(wmax, acc);
    }
    p->bwd_floats = off;   // start of scratch
    p->o_dy = take(gmax);
    p->o_dy2 = take(gmax);
    p->o_da = take(gmax);
    p->o_dd = take(gmax);
    p->o_wt = take(wmax);
    p->o_sums = take(2 * 2 * 2048);      // two ping-pong BN-backward accumulators
    p->ws_bytes = wsb;
    p->arena_bytes = off * sizeof(float) + 2 * wsb + 384;      // two split-K workspaces (main / side stream)
    p->g_valid = false;
    p->g_cur = 0;
    return (int64_t)(intptr_t)p;
}

extern "C" void dpft_resnet_plan_destroy(int64_t h) { delete (ResnetPlan*)(intptr_t)h; }

extern "C" int64_t dpft_resnet_plan_query(int64_t h, int32_t what, int32_t idx) {
    ResnetPlan* p = (ResnetPlan*)(intptr_t)h;
    if (!p) return -1;
    switch (what) {
        case 0: return (int64_t)p->arena_bytes;
        case 1: return p->n_conv;
        case 2: return p->n_bn;
        case 3: return (int64_t)p->out32_off[idx];        // float offset of (the fp32 form of) stage output idx
        case 4: return p->out_shape[idx / 4][idx % 4];    // shape entries
        case 5: return (int64_t)p->fwd_floats;
        default: return -1;
    }
}

extern "C" int dpft_resnet_plan_set_side_stream(int64_t h, dpft_stream_t side) {
    ResnetPlan* p = (ResnetPlan*)(intptr_t)h;
    DPFT_REQUIRE(p, "resnet_plan_set_side_stream: null plan");
    if (p->side_set && p->side_owned && p->side != (hipStream_t)side) (void)hipStreamDestroy(p->side);
    p->side = (hipStream_t)side;      // may be 0: the null stream
    p->side_owned = false;
    p->side_set = true;
    return DPFT_OK;
}

namespace dpft {

struct Tables {
    const dpft_resnet_tables* t;
    const float* w(int i) const { return (const float*)t->conv_w[i]; }
    float* dw(int i) const { return (float*)t->conv_dw[i]; }
    const float* gamma(int i) const { return (const float*)t->bn_gamma[i]; }
    const float* beta(int i) const { return (const float*)t->bn_beta[i]; }
    float* rm(int i) const { return (float*)t->bn_rm[i]; }
    float* rv(int i) const { return (float*)t->bn_rv[i]; }
    float* dgamma(int i) const { return (float*)t->bn_dgamma[i]; }
    float* dbeta(int i) const { return (float*)t->bn_dbeta[i]; }
};

#define RC(call)              \
    do {                      \
        int rc_ = (call);     \
        if (rc_) return rc_;  \
    } while (0)


// act16: the last block of a stage also writes the fp32 copy of its output that the neck reads
static float* stage_out32(const ResnetPlan* p, const BlockPlan& b, float* A) {
    if (!p->desc.act16 || b.out != p->out_off[b.layer]) return nullptr;
    return A + p->out32_off[b.layer];
}

static int bn_params(const ResnetPlan* p, const Tables& T, int bn, const float* stats, int tiles, int rows, int64_t M,
                     int K, float* bnp, bool train, dpft_stream_t st) {
    if (train)
        return dpft_bn_finalize_f32(stats, tiles, rows, M, K, T.gamma(bn), T.beta(bn), p->desc.eps, p->desc.momentum,
                                    T.rm(bn), T.rv(bn), bnp, st);
    return DPFT_OK;      // eval: every BN block was produced up front by eval_bn_blocks()
}

// train-mode forward conv + its BatchNorm block.  Default: per-tile statistics in the conv epilogue + bn_finalize.  The
// finalize can ride in the conv's epilogue instead (BnFinalFuse, DPFT_BN_FINAL_FUSE):
//   1  atomic accumulation + last-ticket workgroup: -0.4 ... -1.1 ms per step, but the forward's BatchNorm blocks then differ
//      in the last bits from run to run, which the encoders' backward amplifies (the full-size repeatability property no
//      longer holds to 2e-4);
//   2  deterministic: the statistics slab stays, the workgroups of a column tile take tickets and the last one merges that
//      tile's columns with bn_finalize_kernel's own arithmetic (bit-identical results, all parity tests green), for launches
//      of <= 64 row tiles (camera layers 3-4: 80 of its 104 BatchNorms).  Measured: 80 launches fewer, step time unchanged
//      (28.95 vs 29.0 ms) -- the merger's ~3 us at the end of each conv (one ticket round trip + 64 slab loads in flight; a
//      first form with the loads one at a time cost 14 us and LOST 0.5 ms) are what the removed kernel boundary cost.
// Both stay off: neither beats the separate 5 us kernel by enough to carry the extra memory-ordering argument.
// The "sums" form (DPFT_BN_SUMS, fp32 tensors): no finalize launch between a conv and its consumer at all -- see common.h: BnSumsRef.
static bool bn_sums_on(const ResnetPlan* p) {
    static const int on = getenv("DPFT_BN_SUMS") ? atoi(getenv("DPFT_BN_SUMS")) : 1;      // A/B switch: 0 = per-tile tables + one bn_finalize launch per layer
    static const int fuse_mode = getenv("DPFT_BN_FINAL_FUSE") ? atoi(getenv("DPFT_BN_FINAL_FUSE")) : 0;
    // (bf16 operands, act16 = 2: built and measured -- DPFT_BN_SUMS=2 -- and off: there the consumers are the materialising
    // elementwise passes, and the step does not gain: batch 8, same box, 22.57 / 22.57 ms with the tables, 22.83 / 22.71 with the sums)
    return on != 0 && fuse_mode == 0 && (p->desc.act16 == 0 || (on == 2 && p->desc.act16 == 2)) && !p->frozen;
}
static BnSumsRef bn_sums_ref(const ResnetPlan* p, const Tables& T, float* A, int bn, int K, int64_t M) {
    return BnSumsRef{(const unsigned long long*)(A + p->o_bnacc + p->bnacc[bn]), T.gamma(bn), T.beta(bn), 1.0 / (double)M, p->desc.eps, K};
}
// the BN blocks + running statistics of the pending layers [from, end of the list) -- BnSumsBatch::MAX per launch
static int bn_finalize_pending(ResnetPlan* p, const Tables& T, float* A, size_t from, dpft_stream_t st) {
    BnSumsBatch batch;
    memset(&batch, 0, sizeof(batch));
    batch.eps = p->desc.eps; batch.momentum = p->desc.momentum;
    for (size_t i = from; i < p->pend_list.size(); ++i) {
        const ResnetPlan::PendingBn& e = p->pend_list[i];
        if (!p->bn_pending[e.bn]) continue;      // (finalized on its own already: a consumer without the sums form)
        const int j = batch.n++;
        batch.sums[j] = (const unsigned long long*)(A + p->o_bnacc + p->bnacc[e.bn]);
        batch.gamma[j] = T.gamma(e.bn); batch.beta[j] = T.beta(e.bn); batch.rm[j] = T.rm(e.bn); batch.rv[j] = T.rv(e.bn);
        batch.out[j] = A + e.bnp; batch.invn[j] = 1.0 / (double)e.M; batch.M[j] = e.M; batch.K[j] = e.K;
        p->bn_pending[e.bn] = 0;
        if (batch.n == BnSumsBatch::MAX) {
            RC(bn_finalize_sums_batch(batch, st));
            batch.n = 0;
        }
    }
    if (batch.n > 0) RC(bn_finalize_sums_batch(batch, st));
    return DPFT_OK;
}
static int bn_finalize_one(ResnetPlan* p, const Tables& T, float* A, int bn, dpft_stream_t st) {
    for (size_t i = 0; i < p->pend_list.size(); ++i)
        if (p->pend_list[i].bn == bn) {
            BnSumsBatch batch;
            memset(&batch, 0, sizeof(batch));
            const ResnetPlan::PendingBn& e = p->pend_list[i];
            batch.eps = p->desc.eps; batch.momentum = p->desc.momentum; batch.n = 1;
            batch.sums[0] = (const unsigned long long*)(A + p->o_bnacc + p->bnacc[bn]);
            batch.gamma[0] = T.gamma(bn); batch.beta[0] = T.beta(bn); batch.rm[0] = T.rm(bn); batch.rv[0] = T.rv(bn);
            batch.out[0] = A + e.bnp; batch.invn[0] = 1.0 / (double)e.M; batch.M[0] = e.M; batch.K[0] = e.K;
            p->bn_pending[bn] = 0;
            return bn_finalize_sums_batch(batch, st);
        }
    set_error("resnet plan: BatchNorm %d is not pending", bn);
    return DPFT_ERR_ARG;
}

// `pro_bn` / `pro_K` / `pro_M`: the BatchNorm layer behind the prologue block `pro` (its index, channels, rows), for the sums form
static int conv_bn_train(ResnetPlan* p, const Tables& T, const ConvRef& c, int bn, const float* x, const float* pro,
                         float* A, float* y, float* stats, int tiles, int rows, int64_t M, float* bnp, void* ws,
                         dpft_stream_t st, const float* w = nullptr, int pro_bn = -1, int pro_K = 0, int64_t pro_M = 0) {
    if (!w) w = T.w(c.w);
    if (p->frozen)      // the BN block comes from the running statistics (eval_bn_blocks): no statistics, no finalize
        return conv_fwd_bnfinal(&c.d, x, w, nullptr, pro, pro ? 1 : 0, y, nullptr, ws, st, nullptr);
    static const int fuse_mode = getenv("DPFT_BN_FINAL_FUSE") ? atoi(getenv("DPFT_BN_FINAL_FUSE")) : 0;      // 0 off | 1 atomics | 2 slab
    static const int slab_tiles = getenv("DPFT_BN_FINAL_TILES") ? std::min(64, atoi(getenv("DPFT_BN_FINAL_TILES"))) : 64;      // the merger holds two tiles per group in registers
    float* acc = A + p->o_bnacc + p->bnacc[bn];
    BnFinalFuse f{acc, (int*)(acc + 2 * c.d.K), T.gamma(bn), T.beta(bn), T.rm(bn), T.rv(bn), bnp, p->desc.eps, p->desc.momentum, false,
                  fuse_mode == 2 ? slab_tiles : 0};
    const bool sums = bn_sums_on(p);
    if (sums) { f.acc = nullptr; f.ticket = nullptr; f.sums = (unsigned long long*)acc; }
    bool launched = false;
    if (sums && pro && pro_bn >= 0 && p->bn_pending[pro_bn]) {
        const BnSumsRef ps = bn_sums_ref(p, T, A, pro_bn, pro_K, pro_M);
        bool used = false;
        RC(conv_fwd_bnfinal(&c.d, x, w, nullptr, pro, 1, y, stats, ws, st, &f, &ps, &used));
        launched = used;
        if (!used) RC(bn_finalize_one(p, T, A, pro_bn, st));      // this conv's kernel reads BN blocks only
    }
    if (!launched) RC(conv_fwd_bnfinal(&c.d, x, w, nullptr, pro, pro ? 1 : 0, y, stats, ws, st, (fuse_mode || sums) ? &f : nullptr));
    if (f.applied && f.sums) {
        p->bn_pending[bn] = 1;
        p->pend_list.push_back(ResnetPlan::PendingBn{bn, c.d.K, (long long)M, (size_t)(bnp - A)});
        return DPFT_OK;
    }
    if (f.applied) return DPFT_OK;
    return dpft_bn_finalize_f32(stats, tiles, rows, M, c.d.K, T.gamma(bn), T.beta(bn), p->desc.eps, p->desc.momentum, T.rm(bn),
                                T.rv(bn), bnp, st);
}

// eval mode: the BN blocks only depend on parameters/buffers -- all of them in ceil(n_bn / 16) launches
static int eval_bn_blocks(const ResnetPlan* p, const Tables& T, float* A, dpft_stream_t st) {
    BnEvalBatch batch;
    memset(&batch, 0, sizeof(batch));
    batch.eps = p->desc.eps;
    auto push = [&](int bn, int K, size_t off) -> int {
        const int i = batch.n++;
        batch.gamma[i] = T.gamma(bn); batch.beta[i] = T.beta(bn); batch.rm[i] = T.rm(bn); batch.rv[i] = T.rv(bn);
        batch.out[i] = A + off; batch.K[i] = K;
        if (batch.n == 16) {
            RC(bn_eval_params_batch(batch, st));
            batch.n = 0;
        }
        return DPFT_OK;
    };
    RC(push(p->bn0, 64, p->p0));
    for (const BlockPlan& b : p->blocks) {
        RC(push(b.bn1, b.c1.d.K, b.p1));
        RC(push(b.bn2, b.c2.d.K, b.p2));
        RC(push(b.bn3, b.c3.d.K, b.p3));
        if (b.has_ds) RC(push(b.bnd, b.cd.d.K, b.pd));
    }
    if (batch.n > 0) RC(bn_eval_params_batch(batch, st));
    return DPFT_OK;
}

}  // namespace dpft

// elementwise pass out = relu(bn(y) [+ bn_r(res)]) of the train-mode forward: a BatchNorm whose block is still pending (the sums
// form) is handed over as its column sums; where only the generic kernel fits, the layer is finalized first
namespace dpft {
static int train_act_pass(ResnetPlan* p, const Tables& T, float* A, const float* y, int bn, int K, int64_t M, const float* bnp,
                          const float* res, int res_bn, const float* res_bnp, float* out, float* out32, bool a16, dpft_stream_t st,
                          unsigned char* mask) {
    const bool py = p->bn_pending[bn] != 0, pr = res_bn >= 0 && p->bn_pending[res_bn] != 0;
    if (py || pr) {
        const BnSumsRef ys = py ? bn_sums_ref(p, T, A, bn, K, M) : BnSumsRef{};
        const BnSumsRef rs = pr ? bn_sums_ref(p, T, A, res_bn, K, M) : BnSumsRef{};
        bool done = false;
        RC(bn_act_sums(y, bnp, ys, res, res_bnp, rs, 1, out, M, K, st, mask, &done, a16, out32));
        if (done) return DPFT_OK;
        if (py) RC(bn_finalize_one(p, T, A, bn, st));
        if (pr) RC(bn_finalize_one(p, T, A, res_bn, st));
    }
    return bn_act_any(y, bnp, res, res_bnp, 1, out, out32, M, K, a16, st, mask);
}
}  // namespace dpft

static int forward_impl(ResnetPlan* p, const float* x, const dpft_resnet_tables* tables, void* arena, int32_t train,
                        dpft_stream_t st) {
    Tables T{tables};
    float* A = (float*)arena;
    void* ws = (char*)arena + (p->arena_bytes - 2 * p->ws_bytes - 256);
    const bool tr = train != 0;
    p->frozen = train == 2;      // (what THIS forward's kernels use; the backward is told by its caller, see dpft_resnet_backward_stage)
    // split-K ticket headers of both workspaces (conv.hip: kWsHeader): zero once per arena use, every conv leaves them zero
    RC(dpft_conv2d_workspace_init(ws, st));
    RC(dpft_conv2d_workspace_init((char*)arena + (p->arena_bytes - p->ws_bytes - 128), st));
    if (!tr || p->frozen) RC(eval_bn_blocks(p, T, A, st));
    const float* xa = x;
    if (p->adj.w >= 0) {
        RC(dpft_conv2d_nhwc_fwd_f32(&p->adj.d, x, T.w(p->adj.w), nullptr, nullptr, 0, A + p->xa, nullptr, ws, st));
        xa = A + p->xa;
    }
    static const bool final_fuse = getenv("DPFT_BN_FINAL_FUSE") != nullptr && atoi(getenv("DPFT_BN_FINAL_FUSE")) != 0;
    if (tr && (final_fuse || bn_sums_on(p))) RC(zero_fill(A + p->o_bnacc, p->bnacc_floats * sizeof(float), st));
    p->pend_list.clear();
    std::fill(p->bn_pending.begin(), p->bn_pending.end(), 0);
    RC(dpft_conv2d_nhwc_fwd_f32(&p->c0.d, xa, T.w(p->c0.w), nullptr, nullptr, 0, A + p->y0, tr && !p->frozen ? A + p->s0 : nullptr, ws, st));
    RC(bn_params(p, T, p->bn0, A + p->s0, p->t0, p->r0, (int64_t)p->c0.d.B * p->c0.d.OH * p->c0.d.OW, 64, A + p->p0, tr && !p->frozen, st));
    const bool a16 = p->desc.act16 != 0;
    RC(bn_relu_maxpool_any(A + p->y0, A + p->p0, A + p->pool, p->c0.d.B, p->c0.d.OH, p->c0.d.OW, 64, p->PH, p->PW, a16, st));
    if (!tr) {
        // Inference: BatchNorm, ReLU and the residual add ride in the epilogue of the conv that produces the tensor
        // (dpft_conv2d_nhwc_fwd_bnact_f32), so no conv carries an operand prologue (applied once per tap and column tile)
        // and no block needs the elementwise pass: 3 launches per bottleneck instead of 4.  y1 / y2 / yd hold ACTIVATED
        // tensors in this mode.
        for (const BlockPlan& b : p->blocks) {
            const int64_t M2 = (int64_t)b.c2.d.B * b.c2.d.OH * b.c2.d.OW;
            // (inference keeps the fp32 weights: descriptors with act16 <= 1)
            dpft_conv_desc d1 = b.c1.d, d2 = b.c2.d, d3 = b.c3.d, dd = b.cd.d;
            d1.act16 = d2.act16 = d3.act16 = dd.act16 = a16 ? 1 : 0;
            RC(dpft_conv2d_nhwc_fwd_bnact_f32(&d1, A + b.x, T.w(b.c1.w), A + b.p1, 1, nullptr, A + b.y1, ws, st));
            RC(dpft_conv2d_nhwc_fwd_bnact_f32(&d2, A + b.y1, T.w(b.c2.w), A + b.p2, 1, nullptr, A + b.y2, ws, st));
            const float* identity = A + b.x;
            if (b.has_ds) {
                RC(dpft_conv2d_nhwc_fwd_bnact_f32(&dd, A + b.x, T.w(b.cd.w), A + b.pd, 0, nullptr, A + b.yd, ws, st));
                identity = A + b.yd;
            }
            float* o32 = stage_out32(p, b, A);
            if (o32) {      // bf16 storage: the stage output also needs its fp32 copy -- the elementwise pass writes both
                RC(dpft_conv2d_nhwc_fwd_f32(&d3, A + b.y2, T.w(b.c3.w), nullptr, nullptr, 0, A + b.y3, nullptr, ws, st));
                RC(bn_act_any(A + b.y3, A + b.p3, identity, nullptr, 1, A + b.out, o32, M2, b.c3.d.K, a16, st));
            } else {
                RC(dpft_conv2d_nhwc_fwd_bnact_f32(&d3, A + b.y2, T.w(b.c3.w), A + b.p3, 1, identity, A + b.out, ws, st));
            }
        }
        p->g_valid = false;
        return DPFT_OK;
    }
    const bool w16 = p->desc.act16 == 2;
    if (w16) {      // bf16 shadow copies of the body's weights (they change with every optimizer step): ceil(n / 80) launches
        TransposeBatch tb;
        tb.n = 0;
        tb.mode = 2;
        for (const BlockPlan& b : p->blocks) {
            const ConvRef* cs[4] = {&b.c1, &b.c2, &b.c3, b.has_ds ? &b.cd : nullptr};
            for (const ConvRef* c : cs) {
                if (!c) continue;
                if (!tb.fits(c->d.K, c->d.C)) {
                    RC(weight_transpose_batch(tb, st));
                    tb.n = 0;
                }
                tb.add(T.w(c->w), A + c->w16, c->d.K, c->d.kh * c->d.kw, c->d.C);
                if (tb.n == TransposeBatch::MAX) {
                    RC(weight_transpose_batch(tb, st));
                    tb.n = 0;
                }
            }
        }
        if (tb.n > 0) RC(weight_transpose_batch(tb, st));
    }
    for (const BlockPlan& b : p->blocks) {
        const int64_t M1 = (int64_t)b.c1.d.B * b.c1.d.OH * b.c1.d.OW, M2 = (int64_t)b.c2.d.B * b.c2.d.OH * b.c2.d.OW;
        if (w16) {
            // bf16 weights: no operand prologues -- relu(bn(y)) is materialised (bf16) by one elementwise pass per layer and
            // every GEMM is the LDS-DMA kernel
            RC(conv_bn_train(p, T, b.c1, b.bn1, A + b.x, nullptr, A, A + b.y1, A + b.s1, b.t1, b.r1, M1, A + b.p1, ws, st, A + b.c1.w16));
            if (pro16_on()) {
                RC(conv_bn_train(p, T, b.c2, b.bn2, A + b.y1, A + b.p1, A, A + b.y2, A + b.s2, b.t2, b.r2, M2, A + b.p2, ws, st, A + b.c2.w16));
                RC(conv_bn_train(p, T, b.c3, b.bn3, A + b.y2, A + b.p2, A, A + b.y3, A + b.s3, b.t3, b.r3, M2, A + b.p3, ws, st, A + b.c3.w16));
            } else {
            RC(train_act_pass(p, T, A, A + b.y1, b.bn1, b.c1.d.K, M1, A + b.p1, nullptr, -1, nullptr, A + b.a1, nullptr, true, st, nullptr));
            RC(conv_bn_train(p, T, b.c2, b.bn2, A + b.a1, nullptr, A, A + b.y2, A + b.s2, b.t2, b.r2, M2, A + b.p2, ws, st, A + b.c2.w16));
            RC(train_act_pass(p, T, A, A + b.y2, b.bn2, b.c2.d.K, M2, A + b.p2, nullptr, -1, nullptr, A + b.a2, nullptr, true, st, nullptr));
            RC(conv_bn_train(p, T, b.c3, b.bn3, A + b.a2, nullptr, A, A + b.y3, A + b.s3, b.t3, b.r3, M2, A + b.p3, ws, st, A + b.c3.w16));
            }
        } else {
        RC(conv_bn_train(p, T, b.c1, b.bn1, A + b.x, nullptr, A, A + b.y1, A + b.s1, b.t1, b.r1, M1, A + b.p1, ws, st));
        RC(conv_bn_train(p, T, b.c2, b.bn2, A + b.y1, A + b.p1, A, A + b.y2, A + b.s2, b.t2, b.r2, M2, A + b.p2, ws, st, nullptr, b.bn1, b.c1.d.K, M1));
        RC(conv_bn_train(p, T, b.c3, b.bn3, A + b.y2, A + b.p2, A, A + b.y3, A + b.s3, b.t3, b.r3, M2, A + b.p3, ws, st, nullptr, b.bn2, b.c2.d.K, M2));
        }
        if (b.has_ds)
            RC(conv_bn_train(p, T, b.cd, b.bnd, A + b.x, nullptr, A, A + b.yd, A + b.sd, b.td, b.rd, M2, A + b.pd, ws, st,
                             w16 ? A + b.cd.w16 : nullptr));
        RC(train_act_pass(p, T, A, A + b.y3, b.bn3, b.c3.d.K, M2, A + b.p3, b.has_ds ? A + b.yd : A + b.x, b.has_ds ? b.bnd : -1,
                          b.has_ds ? A + b.pd : nullptr, A + b.out, stage_out32(p, b, A), a16, st, tr ? (unsigned char*)(A + b.mask) : nullptr));
    }
    RC(bn_finalize_pending(p, T, A, 0, st));      // the sums form: every BN block the backward reads + the running statistics, 48 layers per launch
    p->g_valid = false;
    return DPFT_OK;
}

namespace dpft {

// BN backward = reduce (atomics into `sums`) + apply.  The plan keeps TWO 2*2048-float accumulators: the reduction of
// call k adds into buffer k % 2 (all zero: cleared by the apply pass of call k - 1, or by the stage-start memset) and the
// apply pass of call k clears buffer (k + 1) % 2 -- no memset launch per BatchNorm layer.
struct BnSums {
    float* buf[2];
    int cur;
    bool frozen;      // running-statistics BatchNorm: the apply pass keeps dgamma / dbeta and drops the mean terms
};
// `reduced`: the reduction into bs.buf[bs.cur] has already been done by the kernel that produced `dout` (BnReduceFuse).
static int bn_backward(const float* y, const float* dout, const float* out, const float* mask_bnp, const float* bnp,
                       const float* gamma, BnSums& bs, float* dy, float* dgamma, float* dbeta, int64_t M, int K,
                       dpft_stream_t st, bool act16 = false, const unsigned char* mask8 = nullptr, bool reduced = false) {
    float* sums = bs.buf[bs.cur];
    float* other = bs.buf[bs.cur ^ 1];
    bs.cur ^= 1;
    if (!reduced) RC(bn_bwd_reduce_prezeroed(y, dout, out, mask_bnp, bnp, sums, M, K, act16, st, mask8));
    return bn_bwd_apply_zeroing(y, dout, out, mask_bnp, bnp, gamma, sums, dy, dgamma, dbeta, M, K, other, 2 * 2048, act16, st,
                                mask8, bs.frozen);
}

// Weight gradients do not feed the rest of the backward, so they run on the plan's side stream while the main
// stream continues with the data-gradient chain (mid/late layers launch < 2 workgroups per CU: two kernels in
// flight share the chip).  dy buffers alternate (dy / dy2 / dyd) and each has an event "last wgrad that read it
// is done" which the main stream waits on before overwriting the buffer.
// DPFT_SHARED_WGRAD_STREAM=1: the plans of a process share ONE weight-gradient side stream instead of owning one each
// (the runtime multiplexes HIP streams onto 4 hardware queues; with RCCL's stream next to main + view streams + one side
// stream per plan, streams start to share queues and serialise -- DESIGN.md section 6).  Ordering is by events, so which
// stream carries the work does not change results.
static hipStream_t shared_side_stream() {
    static hipStream_t s = nullptr;
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("DPFT_SHARED_WGRAD_STREAM");
        mode = (e && e[0] == '1') ? 1 : 0;
    }
    if (mode == 1 && !s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    return mode == 1 ? s : nullptr;
}

struct SideCtx {
    ResnetPlan* p;
    hipStream_t main;
    void* ws2;
    int init() {
        if (!p->side_set) {
            if (hipStream_t sh = shared_side_stream()) {
                p->side = sh;
                p->side_owned = false;
            } else {
                DPFT_REQUIRE(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess, "resnet_backward: side stream");
            }
            p->side_set = true;
        }
        if (!p->ev_ready) {
            DPFT_REQUIRE(hipEventCreateWithFlags(&p->ev_ready, hipEventDisableTiming) == hipSuccess, "resnet_backward: event");
            DPFT_REQUIRE(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) == hipSuccess, "resnet_backward: event");
            DPFT_REQUIRE(hipEventCreateWithFlags(&p->ev_wt, hipEventDisableTiming) == hipSuccess, "resnet_backward: event");
            for (auto& e : p->ev_done)
                DPFT_REQUIRE(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "resnet_backward: event");
        }
        return DPFT_OK;
    }
    // main stream is about to overwrite dy buffer `slot`
    int acquire(int slot) {
        // (a slot's previous reader of an EARLIER stage call is already ordered before this one: every stage call ends with
        // join(); waiting only on events of this call also keeps a captured call free of outside dependencies)
        if (!p->ev_valid[slot]) return DPFT_OK;
        DPFT_REQUIRE(hipStreamWaitEvent(main, p->ev_done[slot], 0) == hipSuccess, "resnet_backward: wait event");
        return DPFT_OK;
    }
    // dy buffer `slot` is complete on the main stream: launch the weight gradient that reads it on the side stream
    // `mat_y` (bf16 plan with operand prologues in the forward): x is relu(bn(mat_y)) and has not been materialised yet -- the
    // elementwise pass runs here, on the stream of the weight gradient that wants it
    int wgrad(int slot, const dpft_conv_desc* d, const float* x, const float* dy, const float* pro, int relu, float* dw,
              const float* mat_y = nullptr, const float* mat_bnp = nullptr, int64_t mat_M = 0, int mat_K = 0) {
        if (profiling_active()) {      // per-launch event timing wants each kernel alone on the device
            if (mat_y) RC(bn_act_any(mat_y, mat_bnp, nullptr, nullptr, 1, const_cast<float*>(x), nullptr, mat_M, mat_K, true, (dpft_stream_t)main));
            return dpft_conv2d_nhwc_wgrad_f32(d, x, dy, pro, relu, dw, ws2, (dpft_stream_t)main);
        }
        DPFT_REQUIRE(hipEventRecord(p->ev_ready, main) == hipSuccess, "resnet_backward: record event");
        DPFT_REQUIRE(hipStreamWaitEvent(p->side, p->ev_ready, 0) == hipSuccess, "resnet_backward: wait event");
        if (mat_y) RC(bn_act_any(mat_y, mat_bnp, nullptr, nullptr, 1, const_cast<float*>(x), nullptr, mat_M, mat_K, true, (dpft_stream_t)p->side));
        RC(dpft_conv2d_nhwc_wgrad_f32(d, x, dy, pro, relu, dw, ws2, (dpft_stream_t)p->side));
        DPFT_REQUIRE(hipEventRecord(p->ev_done[slot], p->side) == hipSuccess, "resnet_backward: record event");
        p->ev_valid[slot] = true;
        return DPFT_OK;
    }
    // transposed copies [I][kh][kw][O] of every conv weight of `stage` (the dgrad operand) on the side stream; the
    // main stream waits for them once.  The previous stage's dgrads (main stream) must be done with the region first.
    int transposes(const Tables& T, float* wt_base, int stage) {
        hipStream_t ts = profiling_active() ? main : p->side;
        if (ts != main) {
            DPFT_REQUIRE(hipEventRecord(p->ev_ready, main) == hipSuccess, "resnet_backward: record event");
            DPFT_REQUIRE(hipStreamWaitEvent(ts, p->ev_ready, 0) == hipSuccess, "resnet_backward: wait event");
        }
        TransposeBatch tb;
        tb.n = 0;
        tb.mode = p->desc.act16 == 2 ? 1 : 0;      // bf16 weights: the data-gradient operand is bf16 as well
        for (auto& b : p->blocks) {
            if (b.layer != stage) continue;
            const ConvRef* cs[4] = {&b.c3, &b.c2, &b.c1, b.has_ds ? &b.cd : nullptr};
            for (const ConvRef* c : cs) {
                if (!c) continue;
                if (!tb.fits(c->d.K, c->d.C)) {      // (tile edge of a launch: 64 or 32, common.h)
                    RC(weight_transpose_batch(tb, (dpft_stream_t)ts));
                    tb.n = 0;
                }
                tb.add(T.w(c->w), wt_base + c->wt, c->d.K, c->d.kh * c->d.kw, c->d.C);
                if (tb.n == TransposeBatch::MAX) {
                    RC(weight_transpose_batch(tb, (dpft_stream_t)ts));
                    tb.n = 0;
                }
            }
        }
        if (tb.n > 0) RC(weight_transpose_batch(tb, (dpft_stream_t)ts));
        if (ts != main) {
            DPFT_REQUIRE(hipEventRecord(p->ev_wt, ts) == hipSuccess, "resnet_backward: record event");
            DPFT_REQUIRE(hipStreamWaitEvent(main, p->ev_wt, 0) == hipSuccess, "resnet_backward: wait event");
        }
        return DPFT_OK;
    }
    // everything queued on the side stream is ordered before what follows on the main stream
    int join() {
        DPFT_REQUIRE(hipEventRecord(p->ev_join, p->side) == hipSuccess, "resnet_backward: record event");
        DPFT_REQUIRE(hipStreamWaitEvent(main, p->ev_join, 0) == hipSuccess, "resnet_backward: wait event");
        return DPFT_OK;
    }
};

// `bn3_reduced`: the kernel that produced `gp` already reduced this block's bn3 (see below).  `nextb`: the block whose
// backward follows on the same running gradient (the previous block of the stage, or null) -- the reduction of ITS bn3 is
// folded into the kernel that writes dx here; *next_reduced tells whether that happened.
static int block_backward(ResnetPlan* p, const BlockPlan& b, const Tables& T, float* A, void* ws, SideCtx& sc,
                          BnSums& sums, const float* gp, float* dx, dpft_stream_t st, bool bn3_reduced,
                          const BlockPlan* nextb, bool* next_reduced) {
    float* dyv[2] = {A + p->o_dy, A + p->o_dy2};
    float* dab = A + p->o_da;
    float* dyd = A + p->o_dd;
    float* wt = A + p->o_wt;
    const int planes = b.c1.d.K, K3 = b.c3.d.K;
    const bool a16 = p->desc.act16 != 0;
    static const bool fuse_on = getenv("DPFT_BN_FUSE") == nullptr || atoi(getenv("DPFT_BN_FUSE")) != 0;      // A/B switch
    const int64_t M1 = (int64_t)b.c1.d.B * b.c1.d.OH * b.c1.d.OW, M2 = (int64_t)b.c2.d.B * b.c2.d.OH * b.c2.d.OW;
    int& cur = p->dyi;
    // bn3 (+ the residual ReLU mask taken from the block output)
    RC(sc.acquire(cur));
    const unsigned char* m8 = (const unsigned char*)(A + b.mask);      // ReLU mask of the block output (written by the forward)
    RC(bn_backward(A + b.y3, gp, nullptr, nullptr, A + b.p3, T.gamma(b.bn3), sums, dyv[cur], T.dgamma(b.bn3), T.dbeta(b.bn3), M2, K3, st, a16, m8, bn3_reduced));
    const bool w16 = p->desc.act16 == 2;      // materialised activations a1 / a2: no prologue in the weight gradients either
    if (w16) RC(sc.wgrad(cur, &b.c3.d, A + b.a2, dyv[cur], nullptr, 0, T.dw(b.c3.w), pro16_on() ? A + b.y2 : nullptr, A + b.p2, M2, planes));
    else RC(sc.wgrad(cur, &b.c3.d, A + b.y2, dyv[cur], A + b.p2, 1, T.dw(b.c3.w)));
    // the data gradient of conv3 produces bn2's dout: bn2's reduction rides in its epilogue (mask = bn2(y2) > 0)
    BnReduceFuse f2{A + b.y2, A + b.p2, nullptr, 1, fuse_on ? sums.buf[sums.cur] : nullptr, false};
    RC(conv_dgrad_fused(&b.c3.d, dyv[cur], wt + b.c3.wt, dab, 0, ws, st, &f2));
    cur ^= 1;
    // bn2 (fused-ReLU mask recomputed from its BN block)
    RC(sc.acquire(cur));
    RC(bn_backward(A + b.y2, dab, nullptr, A + b.p2, A + b.p2, T.gamma(b.bn2), sums, dyv[cur], T.dgamma(b.bn2), T.dbeta(b.bn2), M2, planes, st, a16, nullptr, f2.applied));
    if (w16) RC(sc.wgrad(cur, &b.c2.d, A + b.a1, dyv[cur], nullptr, 0, T.dw(b.c2.w), pro16_on() ? A + b.y1 : nullptr, A + b.p1, M1, planes));
    else RC(sc.wgrad(cur, &b.c2.d, A + b.y1, dyv[cur], A + b.p1, 1, T.dw(b.c2.w)));
    BnReduceFuse f1{A + b.y1, A + b.p1, nullptr, 1, fuse_on ? sums.buf[sums.cur] : nullptr, false};
    RC(conv_dgrad_fused(&b.c2.d, dyv[cur], wt + b.c2.wt, dab, 0, ws, st, &f1));
    cur ^= 1;
    // bn1
    RC(sc.acquire(cur));
    RC(bn_backward(A + b.y1, dab, nullptr, A + b.p1, A + b.p1, T.gamma(b.bn1), sums, dyv[cur], T.dgamma(b.bn1), T.dbeta(b.bn1), M1, planes, st, a16, nullptr, f1.applied));
    RC(sc.wgrad(cur, &b.c1.d, A + b.x, dyv[cur], nullptr, 0, T.dw(b.c1.w)));
    // dx is the next block's gp: the reduction of that block's bn3 (input y3, ReLU byte mask of its output) rides in the
    // epilogue of the LAST kernel that writes dx
    BnReduceFuse fn{nullptr, nullptr, nullptr, 0, nullptr, false};
    if (b.has_ds) {
        RC(sc.acquire(2));
        RC(bn_backward(A + b.yd, gp, nullptr, nullptr, A + b.pd, T.gamma(b.bnd), sums, dyd, T.dgamma(b.bnd), T.dbeta(b.bnd), M2, K3, st, a16, m8));
        RC(sc.wgrad(2, &b.cd.d, A + b.x, dyd, nullptr, 0, T.dw(b.cd.w)));
        RC(dpft_conv2d_nhwc_dgrad_f32(&b.cd.d, dyd, wt + b.cd.wt, dx, 0, ws, st));
        if (nextb && fuse_on)
            fn = BnReduceFuse{A + nextb->y3, A + nextb->p3, (const unsigned char*)(A + nextb->mask), 0, sums.buf[sums.cur], false};
        RC(conv_dgrad_fused(&b.c1.d, dyv[cur], wt + b.c1.wt, dx, 1, ws, st, nextb ? &fn : nullptr));
    } else {
        if (nextb && fuse_on)
            fn = BnReduceFuse{A + nextb->y3, A + nextb->p3, (const unsigned char*)(A + nextb->mask), 0, sums.buf[sums.cur], false};
        // identity branch dz = dout * (out > 0) folded into the epilogue of the conv1 data gradient
        // (the byte mask also in bf16 storage: reading the block output back for `out > 0` was a third full-size operand of
        // this epilogue; DPFT_RES_MASK8_A16=0 = that form, A/B switch)
        static const bool m8_a16 = getenv("DPFT_RES_MASK8_A16") == nullptr || atoi(getenv("DPFT_RES_MASK8_A16")) != 0;
        RC(conv_dgrad_residual(&b.c1.d, dyv[cur], wt + b.c1.wt, dx, gp, A + b.out, ws, st, nextb ? &fn : nullptr, (a16 && !m8_a16) ? nullptr : m8));
    }
    if (next_reduced) *next_reduced = fn.applied;
    cur ^= 1;
    return DPFT_OK;
}

}  // namespace dpft

// Backward of one stage (stage = n_layers-1 ... 0; stage 0 also runs the stem).  `dout` is the external
// gradient of that stage's output (may be NULL).  Parameter gradients are written to tables->conv_dw /
// bn_dgamma / bn_dbeta (overwritten).  Stages must be called in descending order after a train forward.
static int backward_stage_impl(ResnetPlan* p, int32_t stage, const float* x, const dpft_resnet_tables* tables,
                               void* arena, const float* dout, dpft_stream_t st) {
    Tables T{tables};
    float* A = (float*)arena;
    void* ws = (char*)arena + (p->arena_bytes - 2 * p->ws_bytes - 256);
    SideCtx sc{p, (hipStream_t)st, (char*)arena + (p->arena_bytes - p->ws_bytes - 128)};
    RC(sc.init());
    {   // The call's buffer choices are a function of (plan, stage) alone -- not of what earlier calls left behind -- so that
        // an executed call and a replayed graph of it are interchangeable: running-gradient buffer and dy slot alternate
        // once per block, counted from the last stage.
        int done = 0;
        for (const BlockPlan& b : p->blocks) done += b.layer > stage ? 1 : 0;
        p->g_cur = done & 1;
        p->dyi = done & 1;
        p->g_valid = stage != p->desc.n_layers - 1;
        p->ev_valid[0] = p->ev_valid[1] = p->ev_valid[2] = false;
    }
    const size_t out_n = (size_t)p->out_shape[stage][0] * p->out_shape[stage][1] * p->out_shape[stage][2] * p->out_shape[stage][3];
    const float* gp;
    if (!p->g_valid) {
        DPFT_REQUIRE(stage == p->desc.n_layers - 1, "resnet_backward: stages must start at the last one");
        if (dout && p->desc.act16) {      // external gradient arrives in fp32: the body's running gradient is bf16
            RC(cvt_f32_to_bf16(dout, A + p->g_off[p->g_cur], (int64_t)out_n, st));
            gp = A + p->g_off[p->g_cur];
        } else if (dout) {
            gp = dout;
        } else {
            RC(zero_fill(A + p->g_off[p->g_cur], out_n * sizeof(float), st));
            gp = A + p->g_off[p->g_cur];
        }
    } else {
        if (dout) RC(add_inplace_any(A + p->g_off[p->g_cur], dout, (int64_t)out_n, p->desc.act16 != 0, st));
        gp = A + p->g_off[p->g_cur];
    }
    BnSums sums{{A + p->o_sums, A + p->o_sums + 2 * 2048}, 0, p->frozen};
    RC(zero_fill(sums.buf[0], 2 * 2 * 2048 * sizeof(float), st));
    RC(sc.transposes(T, A + p->o_wt, stage));
    bool reduced = false;      // the stage's first gradient may still get an external term added: its bn3 reduces on its own
    for (int i = (int)p->blocks.size() - 1; i >= 0; --i) {
        const BlockPlan& b = p->blocks[i];
        if (b.layer != stage) continue;
        float* dx = A + p->g_off[p->g_cur ^ 1];
        const BlockPlan* nextb = (i > 0 && p->blocks[i - 1].layer == stage) ? &p->blocks[i - 1] : nullptr;
        bool next_reduced = false;
        RC(block_backward(p, b, T, A, ws, sc, sums, gp, dx, st, reduced, nextb, &next_reduced));
        reduced = next_reduced;
        p->g_cur ^= 1;
        gp = dx;
    }
    p->g_valid = true;
    if (stage == 0) {
        // stem: maxpool + ReLU + bn1 + conv1 (+ the 1x1 adjustment conv of the radar views)
        float* dab = A + p->o_da;
        float* wt = A + p->o_wt;
        const dpft_conv_desc& d0 = p->c0.d;
        const int64_t M0 = (int64_t)d0.B * d0.OH * d0.OW;
        const int slot = p->dyi;
        float* dyb = A + (slot ? p->o_dy2 : p->o_dy);
        RC(bn_relu_maxpool_bwd_any(A + p->y0, A + p->p0, gp, dab, d0.B, d0.OH, d0.OW, 64, p->PH, p->PW, p->desc.act16 != 0, st));
        RC(sc.acquire(slot));
        RC(bn_backward(A + p->y0, dab, nullptr, nullptr, A + p->p0, T.gamma(p->bn0), sums, dyb, T.dgamma(p->bn0), T.dbeta(p->bn0), M0, 64, st));
        const float* xa = p->adj.w >= 0 ? A + p->xa : x;
        RC(sc.wgrad(slot, &d0, xa, dyb, nullptr, 0, T.dw(p->c0.w)));
        p->dyi ^= 1;
        if (p->adj.w >= 0) {
            RC(dpft_weight_transpose_f32(T.w(p->c0.w), wt, 64, 49, 3, st));
            RC(dpft_conv2d_nhwc_dgrad_f32(&d0, dyb, wt, dab, 0, ws, st));
            RC(dpft_conv2d_nhwc_wgrad_f32(&p->adj.d, x, dab, nullptr, 0, T.dw(p->adj.w), ws, st));
        }
        p->g_valid = false;
    }
    return sc.join();      // parameter gradients of this stage are complete for whatever follows on `st`
}

namespace dpft {

static uint64_t tables_hash(const ResnetPlan* p, const dpft_resnet_tables* t) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* q) { h = (h ^ (uint64_t)(uintptr_t)q) * 1099511628211ull; };
    for (int i = 0; i < p->n_conv; ++i) { mix(t->conv_w[i]); mix(t->conv_dw ? t->conv_dw[i] : nullptr); }
    for (int i = 0; i < p->n_bn; ++i) {
        mix(t->bn_gamma[i]); mix(t->bn_beta[i]); mix(t->bn_rm[i]); mix(t->bn_rv[i]);
        mix(t->bn_dgamma ? t->bn_dgamma[i] : nullptr); mix(t->bn_dbeta ? t->bn_dbeta[i] : nullptr);
    }
    return h;
}

// Run `fn(st)` eagerly, or -- for a plan with graphs enabled whose call is a single-stream sequence -- capture it once per
// argument set (after two eager warm-up calls: lazy stream / event / attribute set-up must not happen inside a capture)
// and replay the executable graph afterwards: one launch from the host instead of several hundred.
template <typename Fn>
static int run_graphed(ResnetPlan* p, int kind, const void* x, const void* arena, const void* dout,
                       const dpft_resnet_tables* tables, dpft_stream_t st, Fn&& fn) {
    const bool single_stream = kind < 0 || (p->side_set && p->side == (hipStream_t)st);
    static const char* kinds = getenv("DPFT_PLAN_GRAPH_KINDS");      // debugging aid: "f" / "b" restrict the captured calls
    if (kinds && ((kind < 0 && !strchr(kinds, 'f')) || (kind >= 0 && !strchr(kinds, 'b')))) return fn();
    if (!p->use_graph || profiling_active() || !single_stream) return fn();
    const uint64_t th = tables_hash(p, tables);
    ResnetPlan::GraphEntry* e = nullptr;
    for (auto& g : p->graphs)
        if (g.kind == kind && g.x == x && g.arena == arena && g.dout == dout && g.st == st && g.tables_hash == th) e = &g;
    if (!e) {
        if (p->graphs.size() > 64) return fn();      // pointers keep changing: the caller is not holding its buffers still
        p->graphs.push_back(ResnetPlan::GraphEntry{kind, x, arena, dout, st, th, 0, nullptr, false});
        e = &p->graphs.back();
    }
    if (e->exec) {
        static const int dbg_sync = getenv("DPFT_PLAN_GRAPH_SYNC") ? atoi(getenv("DPFT_PLAN_GRAPH_SYNC")) : 0;      // debugging aid
        if (dbg_sync & 1) (void)hipDeviceSynchronize();
        DPFT_REQUIRE(hipGraphLaunch(e->exec, (hipStream_t)st) == hipSuccess, "resnet plan: graph launch failed");
        if (dbg_sync & 2) (void)hipDeviceSynchronize();
        return DPFT_OK;
    }
    if (e->failed || e->warm < 2) {
        ++e->warm;
        return fn();
    }
    hipStream_t hs = (hipStream_t)st;
    if (hipStreamBeginCapture(hs, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        e->failed = true;
        return fn();
    }
    const int rc = fn();
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(hs, &graph);
    if (rc != DPFT_OK || ec != hipSuccess || !graph) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        e->failed = true;
        return rc != DPFT_OK ? rc : fn();      // nothing was executed during the capture
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) {
        (void)hipGetLastError();
        e->failed = true;
        return fn();
    }
    e->exec = exec;
    DPFT_REQUIRE(hipGraphLaunch(exec, hs) == hipSuccess, "resnet plan: graph launch failed");
    return DPFT_OK;
}

}  // namespace dpft

extern "C" int dpft_resnet_plan_set_graph(int64_t h, int32_t on) {
    ResnetPlan* p = (ResnetPlan*)(intptr_t)h;
    DPFT_REQUIRE(p, "resnet_plan_set_graph: null plan");
    p->use_graph = on != 0;
    return DPFT_OK;
}

extern "C" int dpft_resnet_forward(int64_t h, const float* x, const dpft_resnet_tables* tables, void* arena,
                                   int32_t train, dpft_stream_t st) {
    ResnetPlan* p = (ResnetPlan*)(intptr_t)h;
    DPFT_REQUIRE(p && x && tables && arena, "resnet_forward: null argument");
    DPFT_REQUIRE(conv_mode_key() == p->mode_key, "resnet_forward: the conv compute mode changed since the plan was built (rebuild the plan)");
    if (train != 1) return forward_impl(p, x, tables, arena, train, st);      // eval, frozen-BN: eager launches
    return run_graphed(p, -1, x, arena, nullptr, tables, st, [&]() { return forward_impl(p, x, tables, arena, train, st); });
}

extern "C" int dpft_resnet_backward_stage(int64_t h, int32_t stage, const float* x, const dpft_resnet_tables* tables,
                                          void* arena, const float* dout, int32_t frozen, dpft_stream_t st) {
    ResnetPlan* p = (ResnetPlan*)(intptr_t)h;
    DPFT_REQUIRE(p && x && tables && arena, "resnet_backward: null argument");
    DPFT_REQUIRE(stage >= 0 && stage < p->desc.n_layers, "resnet_backward: bad stage %d", stage);
    DPFT_REQUIRE(conv_mode_key() == p->mode_key, "resnet_backward: the conv compute mode changed since the plan was built (rebuild the plan)");
    // The BatchNorm mode of the forward this backward belongs to comes from the CALLER (the autograd node remembers it): plan
    // state written by forward_impl would be stale after a replayed (graphed) train forward, which never runs forward_impl, and
    // would flip under a pending backward when another forward of a different mode runs in between.
    p->frozen = frozen != 0;
    if (p->frozen) return backward_stage_impl(p, stage, x, tables, arena, dout, st);
    return run_graphed(p, stage, x, arena, dout, tables, st,
                       [&]() { return backward_stage_impl(p, stage, x, tables, arena, dout, st); });
}
