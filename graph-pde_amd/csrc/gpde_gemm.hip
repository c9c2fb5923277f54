// General fp32 MFMA GEMM used by the NNConv backward: the dense layers of the
// kernel MLP (`DenseNet`, /root/reference/graph-neural-operator/utilities.py:201-227) re-computed and
// differentiated over chunks of edges.  What autograd does through torch.nn.Linear / ReLU in the
// reference (`loss.backward()`, UAI1_full_resolution.py:266) maps to three operand layouts:
//
//   NT  C[M][N] = A[M][K] . B[N][K]^T (+bias) (relu)            forward  U = H . W^T + b
//   NN  C[M][N] = A[M][K] . B[K][N]   (* (mask > 0))            dH = dU . W,  masked by the ReLU
//   TN  C[M][N] = A[K][M]^T . B[K][N]   split over K            dW = dU^T . H  (K = edges)
//
// 128 x 128 tile per workgroup, 4 waves as 2 x 2 (each 64 x 64 = 2 x 2 v_mfma_f32_32x32x2_f32
// blocks), K chunks of 32 staged through LDS with register double-buffering.  An operand whose
// memory layout is K-contiguous is stored [row][36] and read as ds_read_b128 with the k
// permutation k = 8q + 4h + t (both operands use the same k for the same MFMA step); an operand
// with K as the slow index is stored [k][132] and read per k with lanes along the row index.
#include "gpde_common.h"
#include <cstdlib>
#include <cstdio>

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int GT = 128;      // tile edge
constexpr int GK = 32;       // K chunk
constexpr int SKC = 36;      // LDS row stride, K-contiguous operand  [128][36]
constexpr int SKS = 132;     // LDS row stride, K-strided operand     [32][132]

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gpde_gemm_kernel(GpdeGemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[2][A_KC ? GT * SKC : GK * SKS];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_KC ? GT * SKC : GK * SKS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT;
    const int batch = blockIdx.z / g.splits, split = blockIdx.z % g.splits;
    const float* A = g.A + (size_t)batch * g.strideA;
    const float* B = g.B + (size_t)batch * g.strideB;
    if (blockIdx.y & 1) gp_debug_skew(g.skew_us);
    // K range of this split, in chunks of 32
    const int nchunks = (g.K + GK - 1) / GK;
    const int cps = (nchunks + g.splits - 1) / g.splits;
    const int c_lo = split * cps, c_hi = min(nchunks, c_lo + cps);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    auto gload = [&](int chunk) {
        const int k0 = chunk * GK;
        if (A_KC) {          // A[m][k]: 128 rows x 8 float4
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, row = f >> 3, kq = f & 7;
                const int m = m0 + row, k = k0 + kq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < g.M && k < g.K) v = *(const f32x4*)&A[(size_t)m * g.lda + k];   // K % 4 == 0
                ra[i] = v;
            }
        } else {             // A[k][m]: 32 rows x 32 float4
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, kr = f >> 5, mq = f & 31;
                const int k = k0 + kr, m = m0 + mq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < g.K && m < g.M) v = *(const f32x4*)&A[(size_t)k * g.lda + m];   // M % 4 == 0
                ra[i] = v;
            }
        }
        if (B_KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, row = f >> 3, kq = f & 7;
                const int n = n0 + row, k = k0 + kq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < g.N && k < g.K) v = *(const f32x4*)&B[(size_t)n * g.ldb + k];
                rb[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = tid + 256 * i, kr = f >> 5, nq = f & 31;
                const int k = k0 + kr, n = n0 + nq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < g.K && n < g.N) v = *(const f32x4*)&B[(size_t)k * g.ldb + n];   // N % 4 == 0
                rb[i] = v;
            }
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            if (A_KC) *(f32x4*)&As[buf][(f >> 3) * SKC + (f & 7) * 4] = ra[i];
            else *(f32x4*)&As[buf][(f >> 5) * SKS + (f & 31) * 4] = ra[i];
            if (B_KC) *(f32x4*)&Bs[buf][(f >> 3) * SKC + (f & 7) * 4] = rb[i];
            else *(f32x4*)&Bs[buf][(f >> 5) * SKS + (f & 31) * 4] = rb[i];
        }
    };

    if (c_lo < c_hi) {
        gload(c_lo);
        swrite(0);
    }
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
        const int buf = (c - c_lo) & 1;
        if (c + 1 < c_hi) gload(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float av[2][4], bv[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (A_KC) {
                    const f32x4 v = *(const f32x4*)&as[(wm * 64 + i * 32 + l31) * SKC + q * 8 + h * 4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[i][t] = v[t];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[i][t] = as[(q * 8 + h * 4 + t) * SKS + wm * 64 + i * 32 + l31];
                }
                if (B_KC) {
                    const f32x4 v = *(const f32x4*)&bs[(wn * 64 + i * 32 + l31) * SKC + q * 8 + h * 4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[i][t] = v[t];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bv[i][t] = bs[(q * 8 + h * 4 + t) * SKS + wn * 64 + i * 32 + l31];
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(av[i][t], bv[j][t], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_hi) swrite(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    float* C = g.C + (size_t)batch * g.strideC + (size_t)split * g.strideSplit;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        if (n >= g.N) continue;
        const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // what the epilogue READS (the old C of an accumulating call, the ReLU mask) is loaded for all 16 rows before the first
            // store: C may alias neither (checked by the launcher), but the compiler cannot know - interleaved, every row paid one
            // full memory latency (64 dependent round trips: 14 us of a 17 us K = 64 call)
            float old[16], mk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                old[r] = (g.accumulate && m < g.M) ? C[(size_t)m * g.ldc + n] : 0.f;
                mk[r] = (g.mask && m < g.M) ? g.mask[(size_t)m * g.ldmask + n] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = (mk[r] > 0.f) ? v : 0.f;
                if (g.accumulate) v += old[r];
                C[(size_t)m * g.ldc + n] = v;
            }
        }
    }
}

// C[i] (+)= sum_s P[s][i]   (ordered: deterministic)
__global__ void gpde_reduce_splits_kernel(const float* __restrict__ P, size_t n, int splits,
                                          size_t stride, float* __restrict__ C, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // accumulate 1: C is the running sum the partials join one by one; 2: C += (sum of the partials) - what a separate
    // `C = C + new` after a writing call computes (the in-kernel sums over a module's applications: same bits as autograd's)
    float s = accumulate == 1 ? C[i] : 0.f;
    for (int k = 0; k < splits; ++k) s += P[(size_t)k * stride + i];
    C[i] = accumulate == 2 ? C[i] + s : s;
}

// first level for very many partials: P[g * G][i] = sum of the G partials of group g (in place: a thread reads its own element
// of every partial of the group, then overwrites the group's first)
__global__ void gpde_reduce_groups_kernel(float* __restrict__ P, size_t n, int splits, size_t stride, int G) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k0 = blockIdx.y * G, k1 = min(k0 + G, splits);
    float s = 0.f;
    for (int k = k0; k < k1; ++k) s += P[(size_t)k * stride + i];
    P[(size_t)k0 * stride + i] = s;
}

}  // namespace

int gpde_debug_skew_us() { return gpde_switches().debug_skew_us; }

int gpde_launch_gemm(const GpdeGemmArgs& g_in, hipStream_t stream) {
    GpdeGemmArgs g = g_in;
    g.skew_us = gpde_debug_skew_us();
    {   // C must not overlap an operand or the mask: several workgroups read the same operand rows at different times
        const size_t nb = (size_t)(g.batches > 0 ? g.batches - 1 : 0);
        const size_t ca = ((size_t)(g.splits > 1 ? g.splits - 1 : 0) * g.strideSplit + nb * g.strideC + (size_t)(g.M - 1) * g.ldc + g.N) * 4;
        const size_t aa = (nb * g.strideA + (g.a_kcontig ? (size_t)(g.M - 1) * g.lda + g.K : (size_t)(g.K - 1) * g.lda + g.M)) * 4;
        const size_t ba = (nb * g.strideB + (g.b_kcontig ? (size_t)(g.N - 1) * g.ldb + g.K : (size_t)(g.K - 1) * g.ldb + g.N)) * 4;
        const size_t ma = g.mask ? ((size_t)(g.M - 1) * g.ldmask + g.N) * 4 : 0;
        if (g.M > 0 && g.N > 0 && g.K > 0 &&
            (gp_overlap(g.C, ca, g.A, aa) || gp_overlap(g.C, ca, g.B, ba) || gp_overlap(g.C, ca, g.mask, ma))) {
            gpde_set_error("gpde_gemm: output overlaps an operand (internal buffer plan error)");
            return GPDE_EINVAL;
        }
    }
    const dim3 grid((g.M + GT - 1) / GT, (g.N + GT - 1) / GT, g.batches * g.splits), block(256);
    static const bool log_shapes = getenv("GPDE_DEBUG_GEMM_LOG") != nullptr;      // one stderr line per launch (pairs with a rocprofv3 kernel trace)
    if (log_shapes) fprintf(stderr, "[gpde_gemm] <%d,%d> M %d N %d K %d batches %d splits %d acc %d mask %d wgs %u\n", g.a_kcontig, g.b_kcontig, g.M, g.N, g.K,
                            g.batches, g.splits, g.accumulate, g.mask != nullptr, grid.x * grid.y * grid.z);
    if (g.a_kcontig && g.b_kcontig) hipLaunchKernelGGL((gpde_gemm_kernel<true, true>), grid, block, 0, stream, g);
    else if (g.a_kcontig && !g.b_kcontig) hipLaunchKernelGGL((gpde_gemm_kernel<true, false>), grid, block, 0, stream, g);
    else if (!g.a_kcontig && !g.b_kcontig) hipLaunchKernelGGL((gpde_gemm_kernel<false, false>), grid, block, 0, stream, g);
    else { gpde_set_error("gemm layout (A strided, B k-contiguous) not built"); return GPDE_EUNSUPPORTED; }
    GP_LAUNCH_CHECK("gpde_gemm_kernel");
    return GPDE_OK;
}

int gpde_launch_reduce_splits(const float* P, size_t n, int splits, size_t stride, float* C,
                              int accumulate, hipStream_t stream) {
    // one thread per output walks the partials in order.  Few outputs and thousands of partials (column sums per 1024-row
    // strip of a 3 M-row chunk: 1024 threads x 2900 serial loads = 2.1 ms) go through groups of 64 first - still a fixed order
    if (splits > 256 && n <= (1u << 20)) {
        const int G = 64, groups = (splits + G - 1) / G;
        hipLaunchKernelGGL(gpde_reduce_groups_kernel, dim3((unsigned)((n + 255) / 256), groups), dim3(256), 0, stream,
                           const_cast<float*>(P), n, splits, stride, G);
        splits = groups;
        stride *= G;
    }
    hipLaunchKernelGGL(gpde_reduce_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P,
                       n, splits, stride, C, accumulate);
    GP_LAUNCH_CHECK("gpde_reduce_splits_kernel");
    return GPDE_OK;
}
