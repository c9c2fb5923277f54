// Radius graph by CELL LIST, emitted directly as the destination-sorted CSR the operator consumes (SURVEY.md §8 row f2
// as worded there: "cell-list / lattice-stencil; emits reference edge order or CSR directly").
//
// Replaces, together with gpde_csr_from_coo, the chain  ball_connectivity (dense float64 pairwise_distances + np.where,
// /root/reference/graph-neural-operator/utilities.py:250-255; multipole-graph-neural-operator/utilities.py:602-640)
// -> int64 COO [2, E] -> sort by destination.  gpde_radius_graph2_* (gpde_graph.hip) tests all n_src x n_dst pairs and
// writes the COO list, which gpde_csr_from_coo then radix-sorts: O(N^2) tests and 16 + 16 bytes per edge of traffic
// that the operator never needs.  Here:
//   1. the source points are binned into cubic cells of edge >= r (stable sort by cell id: ascending point id inside a cell);
//   2. one wave per DESTINATION point walks the 3^dim neighbouring cells, lanes stride over the members: pass 1 counts
//      (in-degree -> the caller's exclusive scan is `rowptr`), pass 2 writes src / dst slots straight into the CSR;
//   3. the wave sorts its row by source id in LDS (bitonic, rows up to 4096 edges), so the row is in the order a stable
//      sort by destination gives the reference's source-major edge list: rowptr / src / dst are IDENTICAL to
//      gpde_radius_graph2_* + gpde_csr_from_coo, and so are the operator's results (same summation order).
// Same two arithmetics as gpde_graph.hip (exact float64 sum of squares; GPDE_RADIUS_REFERENCE_TIES = scikit-learn's
// dot-product expansion operation by operation).  Edge attributes are addressed by CSR slot (`perm` = identity):
// generated in CSR order (synth / NodeAttr), or read from node data inside the kernel (row f3).
#include "gpde_common.h"
#include <math.h>
#include <rocprim/device/device_radix_sort.hpp>

#pragma clang fp contract(off)      // the reference-ties arithmetic is defined by where it rounds

namespace {

constexpr int CG_SORT_MAX = 4096;    // rows up to this many edges are sorted by source id in LDS

struct CellGrid {
    double lo[3], inv[3];
    int nc[3];
    int dim;
};

__device__ __forceinline__ void cell_of(const CellGrid& g, const double* p, int (&c)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = 0;
        if (k < g.dim) {
            int v = (int)floor((p[k] - g.lo[k]) * g.inv[k]);
            c[k] = v < 0 ? 0 : (v >= g.nc[k] ? g.nc[k] - 1 : v);
        }
    }
}

__global__ void k_cell_ids(const double* __restrict__ pos, int n, CellGrid g, uint32_t* __restrict__ cell, uint32_t* __restrict__ id) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double p[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < g.dim; ++k) p[k] = pos[(size_t)j * g.dim + k];
    int c[3];
    cell_of(g, p, c);
    cell[j] = (uint32_t)((c[2] * g.nc[1] + c[1]) * g.nc[0] + c[0]);
    id[j] = (uint32_t)j;
}

// start[c] = first position in the sorted cell list whose cell id is >= c   (c = 0 .. ncells)
__global__ void k_cell_start(const uint32_t* __restrict__ sorted_cell, int n, int ncells, int32_t* __restrict__ start) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncells) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted_cell[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
    }
    start[c] = lo;
}

template <bool FILL, bool TIES>
__global__ __launch_bounds__(256) void k_cell_neighbors(const double* __restrict__ ps, const double* __restrict__ pd, int nd,
                                                        CellGrid g, double r2, double d2_max, int same_set,
                                                        const int32_t* __restrict__ cell_start, const uint32_t* __restrict__ order,
                                                        int32_t* __restrict__ deg, const int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    extern __shared__ uint32_t sbuf[];                  // FILL: [4 waves][CG_SORT_MAX]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= nd) return;
    double pi[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < g.dim; ++k) pi[k] = pd[(size_t)i * g.dim + k];
    double yy = 0.0;
    if (TIES)
        for (int k = 0; k < g.dim; ++k) yy = yy + pi[k] * pi[k];
    int ci[3];
    cell_of(g, pi, ci);
    const int r0 = FILL ? rowptr[i] : 0;
    const int n_row = FILL ? rowptr[i + 1] - r0 : 0;
    const bool in_lds = FILL && n_row <= CG_SORT_MAX;
    uint32_t* row = sbuf + wave * CG_SORT_MAX;
    int count = 0;
    for (int dz = (g.dim > 2 ? -1 : 0); dz <= (g.dim > 2 ? 1 : 0); ++dz)
        for (int dy = (g.dim > 1 ? -1 : 0); dy <= (g.dim > 1 ? 1 : 0); ++dy) {
            const int cz = ci[2] + dz, cy = ci[1] + dy;
            if (cz < 0 || cz >= g.nc[2] || cy < 0 || cy >= g.nc[1]) continue;
            // the x-neighbours of a cell row are contiguous in the sorted list: one range for dx = -1 .. 1
            const int cx0 = max(ci[0] - 1, 0), cx1 = min(ci[0] + 1, g.nc[0] - 1);
            const int base = (cz * g.nc[1] + cy) * g.nc[0];
            const int p0 = cell_start[base + cx0], p1 = cell_start[base + cx1 + 1];
            for (int q0 = p0; q0 < p1; q0 += 64) {
                const int q = q0 + lane;
                bool hit = false;
                int j = 0;
                if (q < p1) {
                    j = (int)order[q];
                    if (TIES) {
                        double xx = 0.0, dot = 0.0;
                        for (int k = 0; k < g.dim; ++k) {
                            const double x = ps[(size_t)j * g.dim + k];
                            xx = xx + x * x;
                            dot = fma(x, pi[k], dot);
                        }
                        double d2 = -2.0 * dot;
                        d2 = d2 + xx;
                        d2 = d2 + yy;
                        if (d2 < 0.0) d2 = 0.0;
                        if (same_set && i == j) d2 = 0.0;
                        hit = d2 <= d2_max;
                    } else {
                        double d2 = 0.0;
                        for (int k = 0; k < g.dim; ++k) {
                            const double d = pi[k] - ps[(size_t)j * g.dim + k];
                            d2 += d * d;
                        }
                        hit = d2 <= r2;
                    }
                }
                const unsigned long long m = __ballot(hit);
                if (FILL && hit) {
                    const int slot = count + __popcll(m & ((1ull << lane) - 1ull));
                    if (slot < n_row) {
                        if (in_lds) row[slot] = (uint32_t)j;
                        else src[r0 + slot] = j;              // very long rows: cell order (deterministic, not ascending)
                    }
                }
                count += __popcll(m);
            }
        }
    if (!FILL) {
        if (lane == 0) deg[i] = count;
        return;
    }
    if (in_lds) {
        // bitonic sort of the row by source id (distinct keys), padded with sentinels to a power of two
        int np2 = 64;
        while (np2 < n_row) np2 <<= 1;
        for (int t = n_row + lane; t < np2; t += 64) row[t] = 0xffffffffu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // LDS operations of one wave execute in order: a
        __builtin_amdgcn_wave_barrier();                             // compiler fence is all the exchange between lanes needs
        for (int k = 2; k <= np2; k <<= 1)
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
                for (int t = lane; t < np2; t += 64) {
                    const int p = t ^ jj;
                    if (p > t) {
                        const uint32_t a = row[t], b = row[p];
                        const bool up = (t & k) == 0;
                        if ((a > b) == up) { row[t] = b; row[p] = a; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        for (int t = lane; t < n_row; t += 64) src[r0 + t] = (int32_t)row[t];
    }
    for (int t = lane; t < n_row; t += 64) dst[r0 + t] = i;
}

double sqrt_threshold(double r) {      // largest double t with sqrt(t) <= r (see gpde_graph.hip)
    double t = r * r;
    while (t > 0.0 && sqrt(t) > r) t = nextafter(t, 0.0);
    while (sqrt(nextafter(t, INFINITY)) <= r) t = nextafter(t, INFINITY);
    return t;
}

size_t al256(size_t v) { return (v + 255) / 256 * 256; }

int make_grid(int dim, double r, const double* lo, const double* hi, CellGrid* g, int64_t* ncells) {
    double cs = r * 1.0001;                      // cell edge: a pair within r is never more than one cell apart
    if (!(cs > 0.0)) cs = 1.0;
    for (;;) {
        int64_t tot = 1;
        for (int k = 0; k < 3; ++k) {
            g->lo[k] = 0.0; g->inv[k] = 0.0; g->nc[k] = 1;
            if (k < dim) {
                const double ext = hi[k] - lo[k];
                if (!(ext >= 0.0) || !isfinite(ext)) { gpde_set_error("gpde_radius_csr: bad bounds in dimension %d", k); return GPDE_EINVAL; }
                int64_t nc = (int64_t)floor(ext / cs) + 1;
                if (nc < 1) nc = 1;
                if (nc > (1 << 20)) nc = (1 << 20) + 1;      // forces a coarser grid below
                g->lo[k] = lo[k]; g->inv[k] = 1.0 / cs; g->nc[k] = (int)nc;
                tot *= nc;
            }
        }
        if (tot <= ((int64_t)1 << 24)) { *ncells = tot; break; }
        cs *= 2.0;                               // coarser cells: still correct, more candidates per destination
    }
    g->dim = dim;
    return GPDE_OK;
}

int sort_bits_for(int64_t ncells) {
    int bits = 1;
    while (((int64_t)1 << bits) < ncells) ++bits;
    return bits;
}

struct CellWs { uint32_t *cell, *id, *cell_sorted, *order; int32_t* start; void* temp; size_t temp_bytes; size_t total; };

CellWs carve(void* ws, int64_t n_src, int64_t ncells) {
    CellWs w{};
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (size_t)n_src, 0, sort_bits_for(ncells), (hipStream_t)0);
    char* p = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    const size_t a = al256((size_t)(n_src > 0 ? n_src : 1) * 4);
    w.cell = (uint32_t*)p; p += a;
    w.id = (uint32_t*)p; p += a;
    w.cell_sorted = (uint32_t*)p; p += a;
    w.order = (uint32_t*)p; p += a;
    w.start = (int32_t*)p; p += al256((size_t)(ncells + 1) * 4);
    w.temp = p; w.temp_bytes = tb; p += al256(tb);
    w.total = (size_t)(p - (char*)ws) + 256;
    return w;
}

int check(const char* what, const double* ps, int64_t ns, const double* pd, int64_t nd, int dim, double r, uint32_t flags,
          const double* lo, const double* hi) {
    if (!ps || !pd || !lo || !hi || ns < 0 || nd < 0 || ns > 0x7fffffff || nd > 0x7fffffff || dim < 1 || dim > 3 || !(r >= 0.0) ||
        (flags & ~(uint32_t)GPDE_RADIUS_REFERENCE_TIES)) {
        gpde_set_error("%s: bad argument (dim must be 1..3, flags 0 | GPDE_RADIUS_REFERENCE_TIES, host bounds lo / hi required)", what);
        return GPDE_EINVAL;
    }
    return GPDE_OK;
}

}  // namespace

extern "C" size_t gpde_radius_csr_workspace_bytes(int64_t n_src, int dim, double r, const double* lo, const double* hi) {
    CellGrid g;
    int64_t ncells = 0;
    if (n_src < 0 || dim < 1 || dim > 3 || !lo || !hi || make_grid(dim, r, lo, hi, &g, &ncells) != GPDE_OK) return 0;
    return carve(nullptr, n_src, ncells).total + 256;
}

extern "C" int gpde_radius_csr_count(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                                     double r, uint32_t flags, const double* lo, const double* hi, int32_t* deg, void* ws,
                                     size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (int rc = check("gpde_radius_csr_count", pos_src, n_src, pos_dst, n_dst, dim, r, flags, lo, hi)) return rc;
    if (!deg || !ws) { gpde_set_error("gpde_radius_csr_count: deg / ws is null"); return GPDE_EINVAL; }
    CellGrid g;
    int64_t ncells = 0;
    if (int rc = make_grid(dim, r, lo, hi, &g, &ncells)) return rc;
    CellWs w = carve(ws, n_src, ncells);
    if (ws_bytes < w.total) { gpde_set_error("gpde_radius_csr_count: workspace %zu < %zu bytes", ws_bytes, w.total); return GPDE_EWORKSPACE; }
    if (n_dst == 0) return GPDE_OK;
    const int T = 256;
    if (n_src > 0) {
        hipLaunchKernelGGL(k_cell_ids, dim3((unsigned)((n_src + T - 1) / T)), dim3(T), 0, st, pos_src, (int)n_src, g, w.cell, w.id);
        GP_HIP_CHECK(rocprim::radix_sort_pairs(w.temp, w.temp_bytes, w.cell, w.cell_sorted, w.id, w.order, (size_t)n_src, 0,
                                               sort_bits_for(ncells), st));
    }
    hipLaunchKernelGGL(k_cell_start, dim3((unsigned)((ncells + 1 + T - 1) / T)), dim3(T), 0, st, w.cell_sorted, (int)n_src, (int)ncells, w.start);
    const int same = pos_src == pos_dst && n_src == n_dst;
    const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
    if (flags & GPDE_RADIUS_REFERENCE_TIES)
        hipLaunchKernelGGL((k_cell_neighbors<false, true>), grid, block, 0, st, pos_src, pos_dst, (int)n_dst, g, r * r, sqrt_threshold(r),
                           same, w.start, w.order, deg, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    else
        hipLaunchKernelGGL((k_cell_neighbors<false, false>), grid, block, 0, st, pos_src, pos_dst, (int)n_dst, g, r * r, sqrt_threshold(r),
                           same, w.start, w.order, deg, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    GP_LAUNCH_CHECK("gpde_radius_csr_count kernels");
    return GPDE_OK;
}

// `ws` must still hold what gpde_radius_csr_count left there (same arguments); rowptr = exclusive scan of its `deg`.
extern "C" int gpde_radius_csr_fill(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                                    double r, uint32_t flags, const double* lo, const double* hi, const int32_t* rowptr,
                                    int32_t* src, int32_t* dst, int64_t n_edges, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (int rc = check("gpde_radius_csr_fill", pos_src, n_src, pos_dst, n_dst, dim, r, flags, lo, hi)) return rc;
    if (!rowptr || !ws || (n_edges > 0 && (!src || !dst))) { gpde_set_error("gpde_radius_csr_fill: null rowptr / src / dst / ws"); return GPDE_EINVAL; }
    CellGrid g;
    int64_t ncells = 0;
    if (int rc = make_grid(dim, r, lo, hi, &g, &ncells)) return rc;
    CellWs w = carve(ws, n_src, ncells);
    if (ws_bytes < w.total) { gpde_set_error("gpde_radius_csr_fill: workspace %zu < %zu bytes", ws_bytes, w.total); return GPDE_EWORKSPACE; }
    if (n_dst == 0 || n_edges == 0) return GPDE_OK;
    const int same = pos_src == pos_dst && n_src == n_dst;
    const dim3 grid((unsigned)((n_dst + 3) / 4)), block(256);
    const size_t lds = (size_t)4 * CG_SORT_MAX * 4;
    if (flags & GPDE_RADIUS_REFERENCE_TIES)
        hipLaunchKernelGGL((k_cell_neighbors<true, true>), grid, block, lds, st, pos_src, pos_dst, (int)n_dst, g, r * r, sqrt_threshold(r),
                           same, w.start, w.order, (int32_t*)nullptr, rowptr, src, dst);
    else
        hipLaunchKernelGGL((k_cell_neighbors<true, false>), grid, block, lds, st, pos_src, pos_dst, (int)n_dst, g, r * r, sqrt_threshold(r),
                           same, w.start, w.order, (int32_t*)nullptr, rowptr, src, dst);
    GP_LAUNCH_CHECK("gpde_radius_csr_fill kernels");
    return GPDE_OK;
}
