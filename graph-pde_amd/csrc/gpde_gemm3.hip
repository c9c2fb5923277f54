// Node-side kernels of the re-associated NNConv forward.
//
//  gemm3   : part[s][i][o] = sum_{c in split s} sum_k Z_i[c][k] * W3[c*64+o][k]
//            -- the last Linear of DenseNet (/root/reference/graph-neural-operator/utilities.py:223-227)
//            contracted with the message mat-vec of NNConv_old.message (nn_conv.py:273-275), applied
//            once per DESTINATION NODE instead of once per edge.
//  epilogue: out[i] = (sum_s part[s][i] + (sum_{e->i} x_j) . B3) / max(deg_i,1)   ('mean'; 'add' skips
//            the division) + x_i . root + bias   -- scatter-mean normalisation (PyG)
//            and NNConv_old.update (nn_conv.py:277-282).
//  dense   : Y = relu(X . W^T + b), the front layers of kernel MLPs deeper than 3 Linear layers
//            (UAI8_kernel.py:21) — plain LDS-tiled fp32 MFMA GEMM.
#include "gpde_common.h"

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int G3_TM = 64;     // nodes per workgroup
constexpr int G3_BK = 64;     // contraction elements per chunk
constexpr int G3_AS = 68;     // LDS row stride of the A tile (floats): 272-byte rows, conflict-free b128 reads

// grid = (node tiles, splits); block = 256 (4 waves: 2 node blocks x 2 out blocks of 32).
// The kernel streams Z once (256 KiB per node at k2 = 1024: HBM bound) against W3q from L2.  Chunks of 64
// contraction elements are double-buffered in LDS with ONE barrier per chunk; the next chunk's global loads are
// issued right behind the barrier, before the 32 MFMAs of the current one, so that a chunk's HBM latency sits
// under the previous chunk's arithmetic.  A 64-node tile without any in-edge (the destination sub-ranges of the
// MGKN inter-level graphs cover 0.5 - 35 % of the nodes, MGKN_general_darcy2d.py:76-90) returns at once: its Z
// rows were never written and the epilogue does not read its partial sums (deg = 0).
__global__ __launch_bounds__(256) void gpde_gemm3_kernel(GpdeGemm3Args a) {
    __shared__ __attribute__((aligned(16))) float As[2][G3_TM * G3_AS];
    __shared__ __attribute__((aligned(16))) float Bs[2][G3_BK / 4 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int mi = wave >> 1, ni = wave & 1;
    const int node0 = blockIdx.x * G3_TM;
    if (a.rowptr) {
        const int n1 = min(node0 + G3_TM, a.nn);
        if (a.rowptr[a.nc0 + n1] == a.rowptr[a.nc0 + node0]) return;
    }
    const int split = blockIdx.y;
    const size_t KK = (size_t)GP_W * a.K2P;
    const size_t klen = KK / a.splits;
    const size_t kk_lo = klen * split;
    const int nchunks = (int)(klen / G3_BK);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // this thread's 4 A pieces (row f >> 4, 16-byte unit f & 15) and 4 B pieces of a chunk
    f32x4 av[4], bv[4];
    auto gload = [&](size_t kk0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            const int row = f >> 4, kq = f & 15;
            av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (node0 + row < a.nn)
                av[i] = *(const f32x4*)&a.zbuf[(size_t)(node0 + row) * KK + kk0 + kq * 4];
            bv[i] = *(const f32x4*)&a.w3q[kk0 * 64 + (size_t)f * 4];
        }
    };
    gload(kk_lo);
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            *(f32x4*)&As[buf][(f >> 4) * G3_AS + (f & 15) * 4] = av[i];
            *(f32x4*)&Bs[buf][f * 4] = bv[i];
        }
        __syncthreads();          // chunk c visible; everybody is done with chunk c - 1 (the other buffer)
        if (c + 1 < nchunks) gload(kk_lo + (size_t)(c + 1) * G3_BK);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 af = *(const f32x4*)&As[buf][(mi * 32 + l31) * G3_AS + q * 8 + h * 4];
            const f32x4 bf = *(const f32x4*)&Bs[buf][((q * 2 + h) * 64 + ni * 32 + l31) * 4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = mfma32(af[t], bf[t], acc);
        }
    }
    float* p = a.part + ((size_t)split * a.nn) * GP_W + ni * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = node0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < a.nn) p[(size_t)row * GP_W] = acc[r];
    }
}

// one wave per destination node, lane = output channel
__global__ __launch_bounds__(256) void gpde_epilogue_kernel(GpdeEpilogueArgs a) {
    const int lane = threadIdx.x & 63;
    const int li = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (li >= a.nn) return;
    const int i = a.nc0 + li;
    const int r0 = a.rowptr[i], r1 = a.rowptr[i + 1];
    const int deg = r1 - r0;
    float t = 0.f;
    if (deg > 0) {
        // Loads in batches of eight, sums in the original order (same bits): one dependent load per step made this
        // kernel 24-40 us on the MGKN graphs (64 split partials; src -> x_j chains of the in-degree) and 5 ms per
        // step at in-degree 1645.
        {
            const float* pp = a.part + (size_t)li * GP_W + lane;
            const size_t ps = (size_t)a.nn * GP_W;
            int s = 0;
            for (; s + 8 <= a.splits; s += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = pp[(size_t)(s + q) * ps];
#pragma unroll
                for (int q = 0; q < 8; ++q) t += v[q];
            }
            for (; s < a.splits; ++s) t += pp[(size_t)s * ps];
        }
        if (a.b3) {
            float sx = 0.f;
            int e = r0;
            for (; e + 8 <= r1; e += 8) {
                int sj[8];
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) sj[q] = a.src[e + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = a.x[(size_t)sj[q] * GP_W + lane];
#pragma unroll
                for (int q = 0; q < 8; ++q) sx += v[q];
            }
            for (; e < r1; ++e) sx += a.x[(size_t)a.src[e] * GP_W + lane];
            float tb = 0.f;
#pragma unroll 8
            for (int c = 0; c < GP_W; ++c) tb = fmaf(__shfl(sx, c), a.b3[c * GP_W + lane], tb);
            t += tb;
        }
        if (a.aggr == GPDE_AGGR_MEAN) t = t / (float)deg;
    }
    if (a.root) {
        const float xi = a.x[(size_t)i * GP_W + lane];
        float tr = 0.f;
#pragma unroll 8
        for (int c = 0; c < GP_W; ++c) tr = fmaf(__shfl(xi, c), a.root[c * GP_W + lane], tr);
        t += tr;
    }
    if (a.bias) t += a.bias[lane];
    // opt-in caller glue: x' = relu(residual + conv(x)) of the MGKN V-cycles
    // (MGKN_general_darcy2d.py:79-80,89-90; MGKN_orthogonal_burgers1d.py:74-82) without two more elementwise launches
    if (a.residual) t += a.residual[(size_t)i * GP_W + lane];
    if (a.relu_out) t = fmaxf(t, 0.f);
    a.out[(size_t)i * GP_W + lane] = t;
}

// Y[rows][KoutP] = act(X[gather(row)][0:kin] . W[KoutP][ldw]^T + b); tile 128 rows x 64 cols,
// 4 waves (each 32 rows x 64 cols), K chunks of 32 through LDS.
constexpr int DN_TM = 128, DN_TN = 64, DN_S = 36;
__global__ __launch_bounds__(256) void gpde_dense_kernel(GpdeDenseArgs a) {
    __shared__ __attribute__((aligned(16))) float As[DN_TM * DN_S];
    __shared__ __attribute__((aligned(16))) float Bs[DN_TN * DN_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * DN_TM;
    const int col0 = blockIdx.y * DN_TN;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float bv = a.b ? a.b[col0 + nb * 32 + l31] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = bv;
    }
    for (int k0 = 0; k0 < a.ldw; k0 += 32) {
        __syncthreads();
        // A tile: 128 rows x 32 k, element-wise (kin may be tiny / unaligned)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = tid + 256 * i;           // 0..4095
            const int row = f >> 5, k = f & 31;
            float v = 0.f;
            const int rr = row0 + row;
            if (rr < a.rows && k0 + k < a.kin) {
                const size_t srow = a.gather ? (size_t)a.gather[a.row0 + rr] : (size_t)(a.row0 + rr);
                v = a.X[srow * a.ldx + k0 + k];
            }
            As[row * DN_S + k] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;           // 512 float4 = 64 rows x 8
            const int n = f >> 3, kq = f & 7;
            *(f32x4*)&Bs[n * DN_S + kq * 4] =
                *(const f32x4*)&a.W[(size_t)(col0 + n) * a.ldw + k0 + kq * 4];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 af = *(const f32x4*)&As[(wave * 32 + l31) * DN_S + q * 8 + h * 4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const f32x4 bf = *(const f32x4*)&Bs[(nb * 32 + l31) * DN_S + q * 8 + h * 4];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[nb] = mfma32(af[t], bf[t], acc[nb]);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.rows) {
                float v = acc[nb][r];
                if (a.relu) v = fmaxf(v, 0.f);
                a.Y[(size_t)row * a.KoutP + col0 + nb * 32 + l31] = v;
            }
        }
}

}  // namespace

int gpde_launch_gemm3(const GpdeGemm3Args& a, hipStream_t stream) {
    const dim3 grid((a.nn + G3_TM - 1) / G3_TM, a.splits), block(256);
    hipLaunchKernelGGL(gpde_gemm3_kernel, grid, block, 0, stream, a);
    GP_LAUNCH_CHECK("gpde_gemm3_kernel");
    return GPDE_OK;
}

int gpde_launch_epilogue(const GpdeEpilogueArgs& a, hipStream_t stream) {
    const dim3 grid((a.nn + 3) / 4), block(256);
    hipLaunchKernelGGL(gpde_epilogue_kernel, grid, block, 0, stream, a);
    GP_LAUNCH_CHECK("gpde_epilogue_kernel");
    return GPDE_OK;
}

int gpde_launch_dense(const GpdeDenseArgs& a, hipStream_t stream) {
    const dim3 grid((a.rows + DN_TM - 1) / DN_TM, a.KoutP / DN_TN), block(256);
    hipLaunchKernelGGL(gpde_dense_kernel, grid, block, 0, stream, a);
    GP_LAUNCH_CHECK("gpde_dense_kernel");
    return GPDE_OK;
}
