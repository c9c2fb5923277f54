// Pre-passes of the f16-split aggregation inside gpde_fused_f16v3_kernel<false, true>:
// the aggregation Z_i[c][k] = sum_e x_j[c] h_e[k] (NNConv_old.message + scatter,
// /root/reference/graph-neural-operator/nn_conv.py:273-275) contracts over EDGES, so its operands
// can only carry scales that are constant along the contraction: one global power of two for x
// (from max |x|) and one for h (from an a-priori bound that needs max_e B_e, B_e = the per-edge bound
// of the first hidden layer).  Three tiny kernels per forward call:
//   k_absmax_x      scal[0] = max |x|            (atomicMax on the bit pattern; scal zeroed by memset)
//   k_attr_bound    scal[1] = max_e sum_d max_k|W1b[k][d]| |attr_e[d]|   (bias slot d = k0 counts as 1)
//   k_split_x       xs[i][c] = (lo16 << 16) | hi16 of x[i][c] * 2^sx,  hi = rtz16, lo = rn16(rest)
#include "gpde_common.h"

namespace {

__global__ void k_absmax_x(const float* __restrict__ x, size_t n, unsigned* __restrict__ scal) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(scal, m);
}

__global__ void k_attr_bound(const float* __restrict__ attr, int64_t E, int k0, const float* __restrict__ wmax8,
                             unsigned* __restrict__ scal) {
    float w[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) w[d] = wmax8[(d & 1) * 4 + (d >> 1)];      // packed [2][4] order (gpde_pack.hip)
    float m = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        float bnd = w[k0 < 8 ? k0 : 7];                                     // bias slot
        for (int d = 0; d < k0 && d < 8; ++d) bnd = fmaf(w[d], fabsf(attr[(size_t)e * k0 + d]), bnd);
        m = fmaxf(m, bnd);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(scal + 1, __float_as_uint(m));
}

// the same bound with the attributes read from a node table (row f3), edges in CSR order
struct SelArr { int v[8]; };
__global__ void k_attr_bound_nodes(const float* __restrict__ table, int kt, SelArr sel, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, int64_t E, int k0,
                                   const float* __restrict__ wmax8, unsigned* __restrict__ scal) {
    float w[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) w[d] = wmax8[(d & 1) * 4 + (d >> 1)];
    float m = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int s_ = src[e], t_ = dst[e];
        float bnd = w[k0 < 8 ? k0 : 7];
        for (int d = 0; d < k0 && d < 8; ++d)
            bnd = fmaf(w[d], fabsf(table[(size_t)((sel.v[d] >> 8) ? t_ : s_) * kt + (sel.v[d] & 255)]), bnd);
        m = fmaxf(m, bnd);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(scal + 1, __float_as_uint(m));
}

__global__ void k_split_x(const float* __restrict__ x, size_t n, const unsigned* __restrict__ scal,
                          unsigned* __restrict__ xs) {
    const float sc = gpde_pow2_to_2p13(__uint_as_float(scal[0]));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float y = x[i] * sc;
        const _Float16 hi = (_Float16)__builtin_amdgcn_cvt_pkrtz(y, 0.f)[0];
        const _Float16 lo = (_Float16)(y - (float)hi);
        xs[i] = ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16) | __builtin_bit_cast(unsigned short, hi);
    }
}

// Work queue of gpde_fused_f16v6_kernel: the edges of destination nodes [nc0, nc1) cut into node-aligned blocks -
// GP_QBLOCK edges each over the first 7/8 of the edges, GP_QBLOCK / 8 over the rest (guided self-scheduling: big
// blocks keep the per-block partial tile rare, small blocks at the end keep the tail short).
// blk[b] = first CSR slot of the first node whose in-edges start at or after the b-th target offset.
__global__ void k_block_bounds(const int32_t* __restrict__ rowptr, int nc0, int nc1, int nblk_max, int32_t* __restrict__ blk,
                               int32_t* __restrict__ qn, unsigned* __restrict__ qctr, int n_slices) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int e_lo = rowptr[nc0], e_hi = rowptr[nc1];
    const long tot = (long)e_hi - e_lo;
    const int nbig = (int)(tot / 8 * 7 / GP_QBLOCK);
    const long rest = tot - (long)nbig * GP_QBLOCK;
    constexpr int SMALL = GP_QBLOCK / 8;
    const int nblk = min(nbig + (int)((rest + SMALL - 1) / SMALL), nblk_max);
    if (b == 0) *qn = nblk;
    if (b < n_slices) qctr[b] = 0u;
    if (b > nblk) return;
    if (b == nblk) { blk[b] = e_hi; return; }
    const long target = (long)e_lo + (b <= nbig ? (long)b * GP_QBLOCK : (long)nbig * GP_QBLOCK + (long)(b - nbig) * SMALL);
    int lo = nc0, hi = nc1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long)rowptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    blk[b] = rowptr[lo];
}

}  // namespace

int gpde_launch_block_bounds(const int32_t* rowptr, int nc0, int nc1, int nblk_max, int32_t* blk, int32_t* qn,
                             unsigned* qctr, int n_slices, hipStream_t stream) {
    const int threads = nblk_max + 1 > n_slices ? nblk_max + 1 : n_slices;
    hipLaunchKernelGGL(k_block_bounds, dim3((threads + 255) / 256), dim3(256), 0, stream, rowptr, nc0, nc1, nblk_max, blk, qn,
                       qctr, n_slices);
    GP_LAUNCH_CHECK("k_block_bounds");
    return GPDE_OK;
}

int gpde_launch_attr_bound(const float* attr, int64_t n_edges, int k0, const float* wmax8, unsigned* scal, hipStream_t stream,
                           int kt, const int* sel, const int32_t* src, const int32_t* dst) {
    GP_HIP_CHECK(gpde_zero_async(scal, 8, stream));
    if (n_edges <= 0) return GPDE_OK;
    const unsigned ge = (unsigned)((n_edges + 1023) / 1024 < 2048 ? (n_edges + 1023) / 1024 : 2048);
    if (kt) {
        SelArr sa;
        for (int d = 0; d < 8; ++d) sa.v[d] = sel[d];
        hipLaunchKernelGGL(k_attr_bound_nodes, dim3(ge ? ge : 1), dim3(256), 0, stream, attr, kt, sa, src, dst, n_edges, k0, wmax8, scal);
    } else {
        hipLaunchKernelGGL(k_attr_bound, dim3(ge ? ge : 1), dim3(256), 0, stream, attr, n_edges, k0, wmax8, scal);
    }
    GP_LAUNCH_CHECK("k_attr_bound");
    return GPDE_OK;
}

int gpde_launch_g2_prep(const float* x, int64_t n_nodes, const float* attr, int64_t n_edges, int k0,
                        const float* wmax8, unsigned* scal, unsigned* xs, hipStream_t stream, int kt,
                        const int* sel, const int32_t* src, const int32_t* dst) {
    GP_HIP_CHECK(gpde_zero_async(scal, 8, stream));
    const size_t n = (size_t)n_nodes * GP_W;
    const unsigned gx = (unsigned)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024);
    const unsigned ge = (unsigned)((n_edges + 1023) / 1024 < 2048 ? (n_edges + 1023) / 1024 : 2048);
    hipLaunchKernelGGL(k_absmax_x, dim3(gx ? gx : 1), dim3(256), 0, stream, x, n, scal);
    if (n_edges == 0 || !wmax8) {
        // no attributes to bound (aggregation from given hidden activations: scal[1] stays 0)
    } else if (kt) {
        SelArr sa;
        for (int d = 0; d < 8; ++d) sa.v[d] = sel[d];
        hipLaunchKernelGGL(k_attr_bound_nodes, dim3(ge ? ge : 1), dim3(256), 0, stream, attr, kt, sa, src, dst,
                           n_edges, k0, wmax8, scal);
    } else {
        hipLaunchKernelGGL(k_attr_bound, dim3(ge ? ge : 1), dim3(256), 0, stream, attr, n_edges, k0, wmax8, scal);
    }
    hipLaunchKernelGGL(k_split_x, dim3(gx ? gx : 1), dim3(256), 0, stream, x, n, scal, xs);
    GP_LAUNCH_CHECK("gpde_g2_prep kernels");
    return GPDE_OK;
}
