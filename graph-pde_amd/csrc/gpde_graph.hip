// Radius-graph construction on the GPU.
// Replaces `SquareMeshGenerator.ball_connectivity` / `RandomMeshGenerator.ball_connectivity`
// (/root/reference/graph-neural-operator/utilities.py:250-255, 362-368) and the inner / inter-level graphs of
// `RandomMultiMeshGenerator.ball_connectivity` (/root/reference/multipole-graph-neural-operator/utilities.py:602-640):
// dense float64 `sklearn.metrics.pairwise_distances(X[, Y])` (n x n on the CPU: 27 GB at n = 241^2) followed by
// `np.vstack(np.where(pwd <= r))`.  Same output contract: int64 [2, E], edge (j in X -> i in Y) iff |x_j - y_i| <= r,
// sorted by source j then target i; one point set (Y is X): self-loops included.
//
// Two arithmetics for the test `|x_j - y_i| <= r`:
//   * exact (default): sum_k (dx_k)^2 <= r^2 in float64 - symmetric graphs;
//   * GPDE_RADIUS_REFERENCE_TIES: scikit-learn's dot-product expansion, operation by operation -
//         d2 = ((-2 * <x, y>) + |x|^2) + |y|^2,  <x, y> an FMA chain over k from 0 (BLAS dgemm), |.|^2 rounded products
//         summed (row_norms / einsum),  d2 = max(d2, 0),  d2 = 0 on the diagonal when Y is X,  sqrt(d2) <= r
//     - so that pairs at EXACTLY distance r fall on the same side as in the reference: its own default graph (s = 61,
//     r = 0.10, UAI1_full_resolution.py:39-46) has 376,471 edges under this rounding, not the 383,293 of the exact
//     float64 test.  Pinned by tests/golden/mesh_ties.npz (the reference's generator) through oracle/radius_oracle.c.
//
// One wave per source node, lanes stride over the targets (coalesced position loads); two passes:
// count (out-degree per source) and fill (ballot + prefix popcount keeps the targets in order).
// O(n_src * n_dst) pair tests = 3.4e9 at n = 58,081: a few milliseconds, no n x n matrix.
#include "gpde_common.h"
#include <math.h>

// every float64 operation below is meant exactly as written (the reference-ties arithmetic is defined by WHERE it
// rounds): no contraction of a * b + c into an fma unless the source says fma
#pragma clang fp contract(off)

namespace {

template <bool FILL, bool TIES>
__global__ __launch_bounds__(256) void radius_graph_kernel(const double* __restrict__ ps, int ns,
                                                          const double* __restrict__ pd, int nd, int dim, double r,
                                                          double d2_max, int same_set, int32_t* __restrict__ deg,
                                                          const int64_t* __restrict__ offs,
                                                          int64_t* __restrict__ ei, int64_t n_edges) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= ns) return;
    double pj[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < dim; ++k) pj[k] = ps[(size_t)j * dim + k];
    double xx = 0.0;
    if (TIES)
        for (int k = 0; k < dim; ++k) xx = xx + pj[k] * pj[k];
    const double r2 = r * r;
    int64_t base = FILL ? offs[j] : 0;
    int count = 0;
    for (int i0 = 0; i0 < nd; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        if (i < nd) {
            if (TIES) {
                double yy = 0.0, dot = 0.0;
                for (int k = 0; k < dim; ++k) {
                    const double y = pd[(size_t)i * dim + k];
                    yy = yy + y * y;
                    dot = fma(pj[k], y, dot);
                }
                double d2 = -2.0 * dot;
                d2 = d2 + xx;
                d2 = d2 + yy;
                if (d2 < 0.0) d2 = 0.0;
                if (same_set && i == j) d2 = 0.0;
                hit = d2 <= d2_max;          // <=> sqrt(d2) <= r with a correctly rounded sqrt (d2_max from the host)
            } else {
                double d2 = 0.0;
                for (int k = 0; k < dim; ++k) {
                    const double d = pd[(size_t)i * dim + k] - pj[k];
                    d2 += d * d;
                }
                hit = d2 <= r2;
            }
        }
        const unsigned long long m = __ballot(hit);
        if (FILL) {
            if (hit) {
                const int64_t slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (slot < n_edges) {
                    ei[slot] = j;
                    ei[n_edges + slot] = i;
                }
            }
            base += __popcll(m);
        } else {
            count += __popcll(m);
        }
    }
    if (!FILL && lane == 0) deg[j] = count;
}

// largest double t with sqrt(t) <= r under IEEE round-to-nearest: "sqrt(d2) <= r" <=> "d2 <= t" (sqrt is monotone),
// so the device never needs a correctly rounded f64 square root
double sqrt_threshold(double r) {
    double t = r * r;
    while (t > 0.0 && sqrt(t) > r) t = nextafter(t, 0.0);
    while (sqrt(nextafter(t, INFINITY)) <= r) t = nextafter(t, INFINITY);
    return t;
}

int check_args(const char* what, const double* ps, int64_t ns, const double* pd, int64_t nd, int dim, double r, uint32_t flags) {
    if (!ps || !pd || ns < 0 || nd < 0 || ns > 0x7fffffff || nd > 0x7fffffff || dim < 1 || dim > 3 || !(r >= 0.0) ||
        (flags & ~(uint32_t)GPDE_RADIUS_REFERENCE_TIES)) {
        gpde_set_error("%s: bad argument (dim must be 1..3, flags 0 | GPDE_RADIUS_REFERENCE_TIES)", what);
        return GPDE_EINVAL;
    }
    return GPDE_OK;
}

}  // namespace

extern "C" int gpde_radius_graph2_count(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                                        double r, uint32_t flags, int32_t* deg, void* stream_) {
    if (int rc = check_args("gpde_radius_graph2_count", pos_src, n_src, pos_dst, n_dst, dim, r, flags)) return rc;
    if (!deg) { gpde_set_error("gpde_radius_graph2_count: deg is null"); return GPDE_EINVAL; }
    if (n_src == 0) return GPDE_OK;
    const int same = pos_src == pos_dst && n_src == n_dst;
    const dim3 grid((unsigned)((n_src + 3) / 4)), block(256);
    if (flags & GPDE_RADIUS_REFERENCE_TIES)
        hipLaunchKernelGGL((radius_graph_kernel<false, true>), grid, block, 0, (hipStream_t)stream_, pos_src, (int)n_src, pos_dst,
                           (int)n_dst, dim, r, sqrt_threshold(r), same, deg, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0);
    else
        hipLaunchKernelGGL((radius_graph_kernel<false, false>), grid, block, 0, (hipStream_t)stream_, pos_src, (int)n_src, pos_dst,
                           (int)n_dst, dim, r, sqrt_threshold(r), same, deg, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0);
    GP_LAUNCH_CHECK("radius_graph_kernel<count>");
    return GPDE_OK;
}

extern "C" int gpde_radius_graph2_fill(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                                       double r, uint32_t flags, const int64_t* offsets, int64_t* edge_index,
                                       int64_t n_edges, void* stream_) {
    if (int rc = check_args("gpde_radius_graph2_fill", pos_src, n_src, pos_dst, n_dst, dim, r, flags)) return rc;
    if (!offsets || (n_edges > 0 && !edge_index)) { gpde_set_error("gpde_radius_graph2_fill: null offsets / edge_index"); return GPDE_EINVAL; }
    if (n_src == 0 || n_edges == 0) return GPDE_OK;
    const int same = pos_src == pos_dst && n_src == n_dst;
    const dim3 grid((unsigned)((n_src + 3) / 4)), block(256);
    if (flags & GPDE_RADIUS_REFERENCE_TIES)
        hipLaunchKernelGGL((radius_graph_kernel<true, true>), grid, block, 0, (hipStream_t)stream_, pos_src, (int)n_src, pos_dst,
                           (int)n_dst, dim, r, sqrt_threshold(r), same, (int32_t*)nullptr, offsets, edge_index, n_edges);
    else
        hipLaunchKernelGGL((radius_graph_kernel<true, false>), grid, block, 0, (hipStream_t)stream_, pos_src, (int)n_src, pos_dst,
                           (int)n_dst, dim, r, sqrt_threshold(r), same, (int32_t*)nullptr, offsets, edge_index, n_edges);
    GP_LAUNCH_CHECK("radius_graph_kernel<fill>");
    return GPDE_OK;
}
