// The matcher's assignment step on the device (round 5): rectangular linear sum assignment of a batch of cost matrices, one
// wavefront per sample, so that the training step has NO host round trip between the matcher's cost kernel and the criterion
// (before: cost -> D2H + stream sync -> host assignment -> H2D -> criterion: the one window of the step in which the GPU waits for
// the host).  The reference calls scipy.optimize.linear_sum_assignment per sample (src/dprt/training/assigner.py:134-150 via
// training/loss.py:305); scipy is third-party, its algorithm is the shortest augmenting path method of D. F. Crouse, "On implementing
// 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016.  This kernel runs THE SAME step sequence as the host restatement in
// cabi.cpp (lsap_core; see the attribution there: step structure and working-array names after
// scipy/optimize/rectangular_lsap/rectangular_lsap.cpp, BSD 3-Clause) -- dual variables u, v in double precision, the wider side as
// columns, the scan over the not yet visited columns with scipy's tie rule -- with the scan spread over the 64 lanes:
//   sequential rule:  index <- it  when  spc < lowest  or  (spc == lowest and the column is unassigned)
//   = over the set E of positions holding the minimum: the LAST unassigned position of E if there is one, else the FIRST of E,
// which is an associative reduction of (lowest, first position, last unassigned position).  Only additions and subtractions of
// doubles in the same order as the host code (no products: nothing to contract), so pairs AND their order are identical
// (tests/test_gpu_lsap.py: the host function on the same matrices, ties included).
// Errors (a non-finite cost, an infeasible problem) cannot be returned by a kernel: status[0] receives 1 + b (non-finite) or
// 0x10000 + b (infeasible) / 0x20000 + b (count above Mmax) of an offending sample, the sample gets no pairs, and the host raises when it next looks
// (Loss.check_assignment_status: the trainer's logging / epoch sync points).
#include "common.h"

namespace dpft {

struct LsapArgs {
    const float* cost;        // (B, N, Mmax)
    const int32_t* counts;    // (B,) targets per sample
    int32_t* match;           // (B, Mmax, 2)
    int32_t* n_matched;       // (B,)
    int32_t* status;          // (1,) or null
    int B, N, Mmax;
    int cost_in_lds;          // the sample's nr x nc matrix fits behind the working arrays
};

struct Pick {
    double low;
    int first, last_un;
};

__device__ __forceinline__ Pick pick_merge(const Pick& a, const Pick& b) {
    if (a.low < b.low) return a;
    if (b.low < a.low) return b;
    Pick r;
    r.low = a.low;
    r.first = a.first < b.first ? a.first : b.first;      // (positions are >= 0; "none" is INT_MAX / -1)
    r.last_un = a.last_un > b.last_un ? a.last_un : b.last_un;
    return r;
}

// status may be page-locked HOST memory (the host polls it without a sync): a plain store, no device-scope atomic; which of two
// offending samples is reported is not defined
__device__ __forceinline__ void report(int32_t* status, int32_t code) {
    volatile int32_t* s = status;
    if (*s == 0) *s = code;
    __threadfence_system();
}

__global__ __launch_bounds__(64) void lsap_batch_kernel(LsapArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int m = a.counts[b], N = a.N, Mmax = a.Mmax;
    int32_t* mb = a.match + (size_t)b * Mmax * 2;
    for (int k = lane; k < 2 * Mmax; k += 64) mb[k] = -1;
    if (lane == 0) a.n_matched[b] = 0;
    if (m <= 0) return;
    if (m > Mmax) {      // (the host function refuses it as an argument error; here: no pairs + the status word)
        if (lane == 0 && a.status) report(a.status, 0x20000 + b);
        return;
    }
    const float* cb = a.cost + (size_t)b * N * Mmax;
    // rows = the narrower side (scipy transposes a tall problem)
    const bool tall = N > m;
    const int nr = tall ? m : N, nc = tall ? N : m;
    const int rs = tall ? 1 : Mmax, cs = tall ? Mmax : 1;      // element (i, j) of the nr x nc problem = cb[i * rs + j * cs]
    const int NRmax = N < Mmax ? N : Mmax, NCmax = N > Mmax ? N : Mmax;
    double* u = reinterpret_cast<double*>(smem);
    double* v = u + NRmax;
    double* spc = v + NCmax;
    int* path = reinterpret_cast<int*>(spc + NCmax);
    int* row4col = path + NCmax;
    int* remaining = row4col + NCmax;
    int* SC = remaining + NCmax;
    int* col4row = SC + NCmax;
    int* SR = col4row + NRmax;
    float* cl = reinterpret_cast<float*>(SR + NRmax);

    // finite check (scipy raises on NaN / inf entries) + optional staging of the matrix, row-major nr x nc
    bool bad = false;
    for (int e = lane; e < nr * nc; e += 64) {
        const int i = e / nc, j = e - i * nc;
        const float c = cb[(size_t)i * rs + (size_t)j * cs];
        bad |= !(fabsf(c) <= 3.402823466e38f);
        if (a.cost_in_lds) cl[e] = c;
    }
    if (__any(bad)) {
        if (lane == 0 && a.status) report(a.status, 1 + b);
        return;
    }
    const float* cm = a.cost_in_lds ? cl : cb;
    const int crs = a.cost_in_lds ? nc : rs, ccs = a.cost_in_lds ? 1 : cs;
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    __syncthreads();
    const double INF = __longlong_as_double(0x7ff0000000000000LL);

    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; spc[it] = INF; }
        for (int r = lane; r < nr; r += 64) SR[r] = 0;
        __syncthreads();
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            Pick p{INF, 0x7fffffff, -1};
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = minVal + (double)cm[(size_t)i * crs + (size_t)j * ccs] - ui - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const bool un = row4col[j] == -1;
                if (s < p.low) { p.low = s; p.first = it; p.last_un = un ? it : -1; }
                else if (s == p.low) {
                    if (p.first == 0x7fffffff) p.first = it;       // (s == INF on a lane that has seen nothing yet)
                    if (un) p.last_un = it;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                Pick q;
                q.low = __shfl_xor(p.low, off);
                q.first = __shfl_xor(p.first, off);
                q.last_un = __shfl_xor(p.last_un, off);
                p = pick_merge(p, q);
            }
            minVal = p.low;
            if (minVal == INF) {
                if (lane == 0 && a.status) report(a.status, 0x10000 + b);
                return;
            }
            const int index = p.last_un >= 0 ? p.last_un : p.first;
            const int j = remaining[index];
            const int r4 = row4col[j];
            if (r4 == -1) sink = j; else i = r4;
            --num_remaining;
            __syncthreads();                                    // every lane has read remaining[index] / remaining[num_remaining]
            if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining]; }
            __syncthreads();
        }
        // dual update
        for (int r = lane; r < nr; r += 64) {
            if (r == cur) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= minVal - spc[j];
        __syncthreads();
        // augment along the path back to `cur`
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r];
                col4row[r] = j;
                j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    if (!tall) {      // rows are the queries: pairs (n, col4row[n])
        for (int n = lane; n < nr; n += 64) { mb[2 * n] = n; mb[2 * n + 1] = col4row[n]; }
        if (lane == 0) a.n_matched[b] = nr;
    } else {          // rows are the targets: pairs by ascending query index (the assigned queries are distinct)
        for (int t = lane; t < nr; t += 64) {
            const int q = col4row[t];
            int rank = 0;
            for (int o = 0; o < nr; ++o) rank += col4row[o] < q;
            mb[2 * rank] = q;
            mb[2 * rank + 1] = t;
        }
        if (lane == 0) a.n_matched[b] = nr;
    }
}

static size_t lsap_work_bytes(int N, int Mmax) {
    const size_t nrm = N < Mmax ? N : Mmax, ncm = N > Mmax ? N : Mmax;
    return nrm * 8 + ncm * 16 + ncm * 16 + nrm * 8;      // u | v, spc | path, row4col, remaining, SC | col4row, SR
}

}  // namespace dpft

extern "C" int dpft_lsap_batch_dev_f32(const float* cost, const int32_t* counts, int32_t* match, int32_t* n_matched,
                                       int32_t* status, int32_t B, int32_t N, int32_t Mmax, dpft_stream_t stream) {
    DPFT_REQUIRE(cost && counts && match && n_matched && B > 0 && N > 0 && Mmax > 0, "lsap_batch_dev: bad arguments");
    const size_t work = dpft::lsap_work_bytes(N, Mmax);
    const size_t mat = (size_t)N * Mmax * sizeof(float);
    DPFT_REQUIRE(work <= 150 * 1024, "lsap_batch_dev: %d x %d does not fit the LDS working set (use dpft_lsap_batch_f32)", N, Mmax);
    dpft::LsapArgs a;
    a.cost = cost; a.counts = counts; a.match = match; a.n_matched = n_matched; a.status = status;
    a.B = B; a.N = N; a.Mmax = Mmax;
    a.cost_in_lds = work + mat <= 150 * 1024;
    const size_t lds = work + (a.cost_in_lds ? mat : 0);
    if (lds > 64 * 1024) {
        static dpft::LdsGrant grant;
        if (!dpft::lds_grant(grant, reinterpret_cast<const void*>(dpft::lsap_batch_kernel), lds)) {
            dpft::set_error("lsap_batch_dev: cannot reserve %zu bytes of LDS", lds);
            return DPFT_ERR_LAUNCH;
        }
    }
    hipLaunchKernelGGL(dpft::lsap_batch_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, a);
    return dpft::check_launch("lsap_batch_dev");
}

// The step's loss section without a host round trip: assignments on the device -> criterion -> (dcls != NULL) its gradient.
// packed_dev: (B * Mmax * 2 + B) int32 = assignments | matched counts (the layout dpft_assign_loss_f32 uploads).
extern "C" int dpft_assign_loss_dev_f32(const float* cost, const int32_t* counts, int32_t* packed_dev, int32_t* status,
                                        const float* cls, const float* center, const float* size, const float* angle,
                                        const float* gt_box, const float* gt_onehot, const float* weights5, float alpha,
                                        const float* sel, float* scratch, float* losses5, float* total, float* dcls,
                                        float* dcenter, float* dsize, float* dangle, int32_t B, int32_t N, int32_t Mmax, int32_t C,
                                        dpft_stream_t stream) {
    DPFT_REQUIRE(packed_dev, "assign_loss_dev: null argument");
    int32_t* match = packed_dev;
    int32_t* matched = packed_dev + (size_t)B * Mmax * 2;
    int rc = dpft_lsap_batch_dev_f32(cost, counts, match, matched, status, B, N, Mmax, stream);
    if (rc) return rc;
    rc = dpft_set_loss_fwd_total_f32(cls, center, size, angle, gt_box, gt_onehot, match, matched, weights5, alpha, sel, scratch,
                                     losses5, total, B, N, Mmax, C, stream);
    if (rc || !dcls) return rc;
    return dpft_set_loss_bwd_f32(cls, center, size, angle, gt_box, gt_onehot, match, matched, weights5, alpha, sel, dcls, dcenter,
                                 dsize, dangle, B, N, Mmax, C, stream);
}
