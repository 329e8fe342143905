// C-ABI plumbing: version + thread-local error string.
#include <stdarg.h>

#include "common.h"

namespace dpft {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dpft

extern "C" int dpft_version(void) { return 100; }
extern "C" const char* dpft_last_error(void) { return dpft::g_err; }

// ---------------------------------------------------------------------------------------------------------------------
// Host side of the matcher: rectangular linear sum assignment for a batch of cost matrices (dpft_lsap_batch_f32).
// The reference calls scipy.optimize.linear_sum_assignment per sample (src/dprt/training/assigner.py via
// training/loss.py:305); scipy is a third-party dependency, its algorithm is the shortest augmenting path method of
// D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- restated here step for step
// (dual variables u, v; Dijkstra-like search over the not yet scanned columns, unassigned columns preferred on ties; the wider
// side as columns), in double precision like scipy, so that the pairs and their ORDER (ascending row index) are the same.
// Attribution: the step structure and the working-array names (u, v, shortestPathCosts -> spc, path, row4col, col4row, SR, SC,
// remaining, minVal, sink) follow SciPy's implementation of that paper, scipy/optimize/rectangular_lsap/rectangular_lsap.cpp
// (Copyright (c) 2019, PM Larsen and the SciPy developers; BSD 3-Clause License: redistribution and use in source and binary
// forms, with or without modification, are permitted provided that the copyright notice, the list of conditions and the
// disclaimer of the license are retained -- https://github.com/scipy/scipy/blob/main/LICENSE.txt).  No SciPy source is
// included; the algorithm is restated, and tests/test_host.py holds it to scipy's output on 1 200 random problems.
// 4 x (400 x <= 20) problems take a few microseconds here against ~25 us of call overhead each through scipy, inside the
// one window of the training step in which the GPU waits for the host.
// ---------------------------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

namespace dpft {
// cost: nr x nc row-major (nr <= nc).  col4row[i] = column assigned to row i.  false = infeasible
static bool lsap_core(int nr, int nc, const std::vector<double>& cost, std::vector<int>& col4row) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    col4row.assign(nr, -1);
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;      // (scipy fills it in reverse order: ties resolve alike)
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INF) return false;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    return true;
}
}  // namespace dpft

extern "C" int dpft_lsap_batch_f32(const float* cost, int32_t B, int32_t N, int32_t Mmax, const int32_t* counts, int32_t* match,
                                   int32_t* n_matched) {
    DPFT_REQUIRE(cost && counts && match && n_matched && B > 0 && N > 0 && Mmax > 0, "lsap_batch: bad arguments");
    std::vector<double> c;
    std::vector<int> col4row, order;
    for (int b = 0; b < B; ++b) {
        int32_t* mb = match + (size_t)b * Mmax * 2;
        for (int k = 0; k < 2 * Mmax; ++k) mb[k] = -1;
        const int m = counts[b];
        n_matched[b] = 0;
        if (m <= 0) continue;
        DPFT_REQUIRE(m <= Mmax, "lsap_batch: counts[%d] = %d exceeds Mmax = %d", b, m, Mmax);
        const float* cb = cost + (size_t)b * N * Mmax;
        for (int n = 0; n < N; ++n)
            for (int j = 0; j < m; ++j)
                DPFT_REQUIRE(std::isfinite(cb[(size_t)n * Mmax + j]), "lsap_batch: cost matrix %d contains a non-finite entry", b);
        if (N <= m) {      // rows = queries
            c.resize((size_t)N * m);
            for (int n = 0; n < N; ++n)
                for (int j = 0; j < m; ++j) c[(size_t)n * m + j] = cb[(size_t)n * Mmax + j];
            DPFT_REQUIRE(dpft::lsap_core(N, m, c, col4row), "lsap_batch: cost matrix %d is infeasible", b);
            for (int n = 0; n < N; ++n) { mb[2 * n] = n; mb[2 * n + 1] = col4row[n]; }
            n_matched[b] = N;
        } else {           // more queries than targets: the transposed problem, pairs reported by ascending query index
            c.resize((size_t)m * N);
            for (int j = 0; j < m; ++j)
                for (int n = 0; n < N; ++n) c[(size_t)j * N + n] = cb[(size_t)n * Mmax + j];
            DPFT_REQUIRE(dpft::lsap_core(m, N, c, col4row), "lsap_batch: cost matrix %d is infeasible", b);
            order.resize(m);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return col4row[a] < col4row[b2]; });
            for (int k = 0; k < m; ++k) { mb[2 * k] = col4row[order[k]]; mb[2 * k + 1] = order[k]; }
            n_matched[b] = m;
        }
    }
    return DPFT_OK;
}

// The step's host window in ONE call (round 5): assignments of the batch (dpft_lsap_batch_f32) -> their upload from the caller's
// page-locked buffer -> the criterion launch (dpft_set_loss_fwd_total_f32) -> optionally its gradient launch
// (dpft_set_loss_bwd_f32 with d total / d term = sel) straight into the buffers the decoder's backward graph reads.  Between
// the matcher's read-back and the backward graph the GPU has nothing to run: every Python statement there is step time
// (tools/loss_window.py: 270 us of host work before, mostly tensor bookkeeping around four small calls).
// cost: HOST (B, N, Mmax); counts_host (B,): targets per sample; packed_host: page-locked (B * Mmax * 2 + B,) int32 that
// receives assignments | matched counts; packed_dev: its device twin (same layout).
extern "C" int dpft_assign_loss_f32(const float* cost, const int32_t* counts_host, int32_t* packed_host, int32_t* packed_dev,
                                    const float* cls, const float* center, const float* size, const float* angle,
                                    const float* gt_box, const float* gt_onehot, const float* weights5, float alpha,
                                    const float* sel, float* scratch, float* losses5, float* total, float* dcls, float* dcenter,
                                    float* dsize, float* dangle, int32_t B, int32_t N, int32_t Mmax, int32_t C,
                                    dpft_stream_t stream) {
    DPFT_REQUIRE(cost && counts_host && packed_host && packed_dev, "assign_loss: null argument");
    int rc = dpft_lsap_batch_f32(cost, B, N, Mmax, counts_host, packed_host, packed_host + (size_t)B * Mmax * 2);
    if (rc) return rc;
    const size_t bytes = ((size_t)B * Mmax * 2 + B) * sizeof(int32_t);
    if (hipMemcpyAsync(packed_dev, packed_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        dpft::set_error("assign_loss: upload failed");
        return DPFT_ERR_LAUNCH;
    }
    const int32_t* match = packed_dev;
    const int32_t* matched = packed_dev + (size_t)B * Mmax * 2;
    rc = dpft_set_loss_fwd_total_f32(cls, center, size, angle, gt_box, gt_onehot, match, matched, weights5, alpha, sel, scratch,
                                     losses5, total, B, N, Mmax, C, stream);
    if (rc || !dcls) return rc;
    return dpft_set_loss_bwd_f32(cls, center, size, angle, gt_box, gt_onehot, match, matched, weights5, alpha, sel, dcls, dcenter,
                                 dsize, dangle, B, N, Mmax, C, stream);
}
