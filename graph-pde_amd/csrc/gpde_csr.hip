// Destination-sorted CSR from a COO edge list (int64 [2,E], possibly a strided view).
// Replaces the per-call gather/scatter index handling of PyG's MessagePassing.propagate
// (call site /root/reference/graph-neural-operator/nn_conv.py:271).  The sort is a stable LSD
// radix sort (rocPRIM device primitive) on the destination id, so edges of one destination keep
// their input order and the per-destination summation order is deterministic.
#include "gpde_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

__global__ void csr_extract_kernel(const int64_t* __restrict__ ei, int64_t s_row, int64_t s_col,
                                   int n_edges, int n_nodes, uint32_t* __restrict__ keys,
                                   uint32_t* __restrict__ vals, int32_t* __restrict__ n_bad) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t j = ei[e * s_col];
    const int64_t i = ei[s_row + e * s_col];
    const bool bad = (j < 0) | (j >= n_nodes) | (i < 0) | (i >= n_nodes);
    if (bad) atomicAdd(n_bad, 1);
    keys[e] = bad ? (uint32_t)n_nodes : (uint32_t)i;   // bad edges sort behind every real row
    vals[e] = (uint32_t)e;
}

__global__ void csr_gather_src_kernel(const int64_t* __restrict__ ei, int64_t s_col, int n_edges,
                                      int n_nodes, const int32_t* __restrict__ perm,
                                      int32_t* __restrict__ src) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int64_t j = ei[(int64_t)perm[e] * s_col];
    src[e] = (j >= 0 && j < n_nodes) ? (int32_t)j : 0;
}

// rowptr[i] = first CSR slot whose destination is >= i   (i = 0..N)
__global__ void csr_rowptr_kernel(const int32_t* __restrict__ dst, int n_edges, int n_nodes,
                                  int32_t* __restrict__ rowptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_nodes) return;
    int lo = 0, hi = n_edges;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (dst[mid] < i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = lo;
}

int sort_bits(int64_t n_nodes) {
    int bits = 1;
    while (((int64_t)1 << bits) <= n_nodes) ++bits;   // keys go up to n_nodes inclusive
    return bits;
}

size_t sort_temp_bytes(int64_t n_edges, int64_t n_nodes) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n_edges, 0,
                                    sort_bits(n_nodes), (hipStream_t)0);
    return bytes;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

__global__ void csr_iota_kernel(uint32_t* __restrict__ v, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) v[e] = (uint32_t)e;
}

}  // namespace

extern "C" size_t gpde_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes) {
    if (n_edges < 0 || n_nodes < 0) return 0;
    return 2 * align256((size_t)n_edges * 4) + align256(sort_temp_bytes(n_edges, n_nodes)) + 256;
}

extern "C" int gpde_csr_from_coo(const int64_t* edge_index, int64_t stride_row, int64_t stride_col,
                                 int64_t n_edges, int64_t n_nodes, int32_t* rowptr, int32_t* src,
                                 int32_t* dst, int32_t* perm, int32_t* n_bad, void* ws,
                                 size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || n_nodes < 0 || !rowptr || !n_bad || (n_edges > 0 && (!edge_index || !src || !dst || !perm))) {
        gpde_set_error("gpde_csr_from_coo: null/negative argument");
        return GPDE_EINVAL;
    }
    if (n_edges >= ((int64_t)1 << 31) - 64 || n_nodes >= ((int64_t)1 << 31) - 64) {
        gpde_set_error("gpde_csr_from_coo: %lld edges / %lld nodes exceed the int32 CSR", (long long)n_edges,
                       (long long)n_nodes);
        return GPDE_EUNSUPPORTED;
    }
    if (ws_bytes < gpde_csr_workspace_bytes(n_edges, n_nodes) || (!ws && n_edges > 0)) {
        gpde_set_error("gpde_csr_from_coo: workspace %zu < %zu bytes", ws_bytes,
                       gpde_csr_workspace_bytes(n_edges, n_nodes));
        return GPDE_EWORKSPACE;
    }
    GP_HIP_CHECK(gpde_zero_async(n_bad, sizeof(int32_t), stream));
    const int T = 256;
    const int E = (int)n_edges, N = (int)n_nodes;
    if (E > 0) {
        char* w = (char*)ws;
        uint32_t* keys = (uint32_t*)w;  w += align256((size_t)E * 4);
        uint32_t* vals = (uint32_t*)w;  w += align256((size_t)E * 4);
        size_t temp_bytes = sort_temp_bytes(n_edges, n_nodes);
        hipLaunchKernelGGL(csr_extract_kernel, dim3((E + T - 1) / T), dim3(T), 0, stream, edge_index,
                           stride_row, stride_col, E, N, keys, vals, n_bad);
        GP_HIP_CHECK(rocprim::radix_sort_pairs((void*)w, temp_bytes, keys, (uint32_t*)dst, vals,
                                               (uint32_t*)perm, (size_t)E, 0, sort_bits(n_nodes),
                                               stream));
        hipLaunchKernelGGL(csr_gather_src_kernel, dim3((E + T - 1) / T), dim3(T), 0, stream,
                           edge_index, stride_col, E, N, perm, src);
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((N + 1 + T - 1) / T), dim3(T), 0, stream, dst, E, N,
                       rowptr);
    GP_LAUNCH_CHECK("gpde_csr kernels");
    return GPDE_OK;
}

// CSR slots regrouped by SOURCE node: src_slots = the slots 0..E-1 stably sorted by src[slot] (ascending slot inside a
// source), src_rowptr[j] = first position of source j.  The backward reduces dx_j = sum over the out-edges of j in
// THIS order instead of by atomics (what autograd's index_select backward does in the reference:
// /root/reference/graph-neural-operator/nn_conv.py:271 -> PyG propagate; result order there is unspecified).
extern "C" int gpde_csr_source_order(const int32_t* src, int64_t n_edges, int64_t n_nodes, int32_t* src_rowptr,
                                     int32_t* src_slots, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_edges < 0 || n_nodes < 0 || !src_rowptr || (n_edges > 0 && (!src || !src_slots))) {
        gpde_set_error("gpde_csr_source_order: null/negative argument");
        return GPDE_EINVAL;
    }
    if (n_edges >= ((int64_t)1 << 31) - 64 || n_nodes >= ((int64_t)1 << 31) - 64) {
        gpde_set_error("gpde_csr_source_order: sizes exceed the int32 CSR");
        return GPDE_EUNSUPPORTED;
    }
    if (ws_bytes < gpde_csr_workspace_bytes(n_edges, n_nodes) || (!ws && n_edges > 0)) {
        gpde_set_error("gpde_csr_source_order: workspace %zu < %zu bytes", ws_bytes, gpde_csr_workspace_bytes(n_edges, n_nodes));
        return GPDE_EWORKSPACE;
    }
    const int T = 256;
    const int E = (int)n_edges, N = (int)n_nodes;
    uint32_t* keys_out = (uint32_t*)ws;
    if (E > 0) {
        char* w = (char*)ws + align256((size_t)E * 4);
        uint32_t* vals = (uint32_t*)w;  w += align256((size_t)E * 4);
        size_t temp_bytes = sort_temp_bytes(n_edges, n_nodes);
        hipLaunchKernelGGL(csr_iota_kernel, dim3((E + T - 1) / T), dim3(T), 0, stream, vals, E);
        GP_HIP_CHECK(rocprim::radix_sort_pairs((void*)w, temp_bytes, (const uint32_t*)src, keys_out, vals,
                                               (uint32_t*)src_slots, (size_t)E, 0, sort_bits(n_nodes), stream));
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((N + 1 + T - 1) / T), dim3(T), 0, stream, (const int32_t*)keys_out, E, N,
                       src_rowptr);
    GP_LAUNCH_CHECK("gpde_csr_source_order kernels");
    return GPDE_OK;
}

// ---- per-edge rows in CSR slot order (ops.attr_in_slot_order) ------------------------------------------------------------
namespace {
// one thread per output element: a row is k0 <= 8 floats (24 bytes at k0 = 6), consecutive threads write consecutive
// floats (full lines out); the reads are the scattered rows - once per (graph, tensor)
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ rows, int k, const int32_t* __restrict__ perm,
                                                     size_t total, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t s = i / (unsigned)k;
    const int d = (int)(i - s * (unsigned)k);
    out[i] = rows[(size_t)perm[s] * k + d];
}
}  // namespace

extern "C" int gpde_gather_rows(const float* rows, int k, const int32_t* perm, int64_t n, float* out, void* stream_) {
    if (n < 0 || k < 1 || (n > 0 && (!rows || !perm || !out))) { gpde_set_error("gpde_gather_rows: null/negative argument"); return GPDE_EINVAL; }
    if (n == 0) return GPDE_OK;
    const size_t total = (size_t)n * k;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffull) { gpde_set_error("gpde_gather_rows: %lld rows x %d exceed one launch", (long long)n, k); return GPDE_EUNSUPPORTED; }
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, rows, k, perm, total, out);
    GP_LAUNCH_CHECK("k_gather_rows");
    return GPDE_OK;
}
