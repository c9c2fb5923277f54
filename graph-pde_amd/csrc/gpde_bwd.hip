// Native backward of the fused NNConv.
//
// Replaces what autograd does in the reference on `loss.backward()`
// (/root/reference/graph-neural-operator/UAI1_full_resolution.py:266) through
// NNConv_old.update / message (nn_conv.py:273-282), the scatter-mean and gather of PyG's propagate
// and the Linear/ReLU chain of DenseNet (utilities.py:223-227) — without ever forming the
// [E,4096] per-edge weight tensor or its gradient.
//
// With  H_e = last hidden activation, Z_i = sum_{e->i} x_j (x) H_e, S_i = sum_{e->i} x_j,
//       T_i = Z_i : W3 + S_i . B3,  out_i = T_i / deg_i + x_i . root + bias            (forward)
// and g = dL/dout, gT_i = g_i / deg_i ('mean') or g_i ('add'):
//   dbias = sum_i g_i              droot = X^T g                  dx_i += g_i . root^T
//   dW3[c*64+o][k] = sum_i gT_i[o] Z_i[c][k]                      db3[c*64+o] = sum_i S_i[c] gT_i[o]
//   dZ_i[c][k] = sum_o W3[c*64+o][k] gT_i[o]                      dS_i[c] = sum_o b3[c*64+o] gT_i[o]
//   per edge e: j -> i:   dH_e = x_j . dZ_i ,   dx_j += dZ_i . H_e + dS_i
//   then the plain MLP backward over edges: dU_l = dH_l * (H_l > 0), dW_l = dU_l^T H_{l-1},
//   db_l = colsum dU_l, dH_{l-1} = dU_l W_l.
//
// Edges are processed in node-aligned chunks (host partition from rowptr_host) so that the hidden
// activations of a chunk ([edges][k], recomputed, never saved by the forward) fit the workspace;
// all contractions run on fp32 MFMA: gpde_gemm (dense layers, weight gradients), gpde_zagg_kernel
// (Z of the chunk from the recomputed or given H) and the two edge kernels below -- except the recompute
// of the last hidden layer of 3-Linear kernels, which reuses the forward's fused f16-split kernel with
// its store epilogue (gpde_fused_f16v3_kernel<true, ...>).
// Weight-gradient reductions use ordered split partials (deterministic).  dx_j: with the source-ordered slot list
// (src_rowptr / src_slots given) per-edge contributions are written out and summed per source node in slot order by
// k_dx_reduce (bit-reproducible); without it, fp32 atomics (as the reference's scatter backward does on a GPU).
#include "gpde_common.h"
#include <cstdlib>
#include <vector>

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- small layout kernels ------------------------------------------------------------------------
__global__ void k_pad_mat(const float* __restrict__ W, int rows, int cols, int ld, int rowsP, int colsP,
                          float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rowsP * colsP) return;
    const int c = i % colsP, r = i / colsP;
    out[i] = (r < rows && c < cols) ? W[(size_t)r * ld + c] : 0.f;
}
__global__ void k_transpose(const float* __restrict__ W, int rows, int cols, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[(size_t)c * rows + r] = W[i];
}

__global__ void k_unpad_mat(const float* __restrict__ P, int rows, int cols, int ldp, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const int c = i % cols, r = i / cols;
    out[i] = P[(size_t)r * ldp + c];
}
__global__ void k_gather_attr(const float* __restrict__ attr, const int32_t* __restrict__ perm, int e0,
                              int rows, int k0, int KP0, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * KP0) return;
    const int d = i % KP0, r = i / KP0;
    out[i] = (d < k0) ? attr[(size_t)perm[e0 + r] * k0 + d] : 0.f;
}
// grad_attr[perm[e0 + r]][d] = sum_k dU1[r][k] * W1p[k][d]  (d < k0): the gradient reaching the edge attributes through the
// first Linear of DenseNet (autograd gives `pseudo` a gradient when it requires one; no reference script asks for it).
// One wave per edge row (a 4 KiB read of dU_1), lanes split k, eight running dots, wave reduction; every edge written once.
__global__ __launch_bounds__(256) void k_grad_attr(const float* __restrict__ dU1, int K, const float* __restrict__ W1p, int ldw,
                                                   const int32_t* __restrict__ perm, int e0, int rows, int k0,
                                                   float* __restrict__ grad_attr) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
        const float u = dU1[(size_t)r * K + k];
        const f32x4 w0 = *(const f32x4*)(W1p + (size_t)k * ldw), w1 = *(const f32x4*)(W1p + (size_t)k * ldw + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) { acc[d] = fmaf(u, w0[d], acc[d]); acc[4 + d] = fmaf(u, w1[d], acc[4 + d]); }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[d] += __shfl_xor(acc[d], o);
    if (lane < k0) {
        float v = acc[0];
#pragma unroll
        for (int d = 1; d < 8; ++d) v = lane == d ? acc[d] : v;
        grad_attr[(size_t)perm[e0 + r] * k0 + lane] = v;
    }
}
// the same from a NODE table (SURVEY.md §8 row f3 in training): slot d of the edge in CSR slot s is
// table[(sel[d] >> 8 ? dst[s] : src[s]) * kt + (sel[d] & 255)] - no [E][k0] tensor, no slot-order copy
struct NodeAttrSel { int kt; int sel[8]; };
__global__ void k_gather_attr_nodes(const float* __restrict__ table, NodeAttrSel na, const int32_t* __restrict__ src,
                                    const int32_t* __restrict__ dst, int e0, int rows, int k0, int KP0, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * KP0) return;
    const int d = i % KP0, r = i / KP0;
    float v = 0.f;
    if (d < k0) {
        const int sd = na.sel[d];
        v = table[(size_t)((sd >> 8) ? dst[e0 + r] : src[e0 + r]) * na.kt + (sd & 255)];
    }
    out[i] = v;
}
// gT_i = g_i / max(deg,1) ('mean') or g_i ('add'); rows with no in-edge get 0 (they have no Z)
__global__ void k_scale_g(const float* __restrict__ g, const int32_t* __restrict__ rowptr, int aggr,
                          int n0, int nn, float* __restrict__ gT) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nn) return;
    const int deg = rowptr[n0 + i + 1] - rowptr[n0 + i];
    float v = g[(size_t)(n0 + i) * GP_W + lane];
    if (deg == 0) v = 0.f;
    else if (aggr == GPDE_AGGR_MEAN) v = v / (float)deg;
    gT[(size_t)i * GP_W + lane] = v;
}
// S_i = sum over the in-edges of node i of x_src(e)  (the (sum x_j) . B3 term's operand, and db3 = S (x) gT).  One WORKGROUP per
// node, its in-edges dealt round-robin to the four waves, eight gathered rows in flight per wave, partials combined in wave
// order (fixed summation order: bit-reproducible).  (Rounds 1-3: one wave per node walking its ~1,650 in-edges one dependent
// load at a time - 0.41 s of an 8 s training step on the 241^2 graph, round-4 profile.)
__global__ __launch_bounds__(256) void k_nbr_sum(const float* __restrict__ x, const int32_t* __restrict__ rowptr,
                                                 const int32_t* __restrict__ src, int n0, int nn, float* __restrict__ S) {
    __shared__ float part[4][GP_W];
    const int i = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= nn) return;
    const int r0 = rowptr[n0 + i], r1 = rowptr[n0 + i + 1];
    float s = 0.f;
    int e = r0 + wave;
    for (; e + 28 < r1; e += 32) {
        int j[8];
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) j[q] = src[e + 4 * q];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[(size_t)j[q] * GP_W + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; e < r1; e += 4) s += x[(size_t)src[e] * GP_W + lane];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) S[(size_t)i * GP_W + lane] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
// partial column sums: P[split][col] = sum over the split's rows of M[row][col].  A lane owns four consecutive
// columns (16-byte loads, a wave reads 1 KiB of a row), the four waves interleave rows, four rows in flight per wave:
// the bias gradients read every dU once (8 KiB per edge at two 1024-wide layers) and are pure HBM streaming.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ M, int rows, int cols, int ld, int splits,
                                                float* __restrict__ P, unsigned* __restrict__ absmax_bits) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4, split = blockIdx.y;
    const int rps = (rows + splits - 1) / splits;
    const int r_lo = split * rps, r_hi = min(rows, r_lo + rps);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    unsigned mx[4] = {0u, 0u, 0u, 0u};       // optional: column maxima of |M| (bit patterns) from the same pass
    auto upd = [&](const f32x4& v) {
#pragma unroll
        for (int j = 0; j < 4; ++j) mx[j] = max(mx[j], __float_as_uint(v[j]) & 0x7fffffffu);
    };
    if (col < cols) {
        const float* p = M + col;
        int r = r_lo + rg;
        for (; r + 12 < r_hi; r += 16) {
            const f32x4 a = *(const f32x4*)(p + (size_t)r * ld), b = *(const f32x4*)(p + (size_t)(r + 4) * ld);
            const f32x4 c = *(const f32x4*)(p + (size_t)(r + 8) * ld), d = *(const f32x4*)(p + (size_t)(r + 12) * ld);
            s0 += a; s1 += b; s2 += c; s3 += d;
            if (absmax_bits) { upd(a); upd(b); upd(c); upd(d); }
        }
        for (; r < r_hi; r += 4) {
            const f32x4 a = *(const f32x4*)(p + (size_t)r * ld);
            s0 += a;
            if (absmax_bits) upd(a);
        }
        if (absmax_bits)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (mx[j]) atomicMax(absmax_bits + col + j, mx[j]);
    }
    red[rg][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && col < cols) {
        const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        *(f32x4*)(P + (size_t)split * cols + col) = t;
    }
}

// amax[d] = max over rows of |T[row][col_d]| (fp32 bit patterns, zeroed by the caller), col_d = d for an attribute tensor [rows][k0],
// (a node table is read through the edge's end points): the maximum of every edge's slot d over the WHOLE call - the in-kernel first layer
// (gpde_gemm_f16s.hip, fl_mode) scales its operands from it, so that every edge chunk of the call - and every chunking - uses the
// same scales and forms the same H_1 bits (ReLU mask included) for an edge
// bits of the a-priori bound of every last-hidden activation: max|b2| + max_k ||W2_k||_1 . max_e B_e (fcol[8], fcol[9] of the pack
// image, scal[1] from k_attr_bound) - the word gpde_zagg_kernel<true> takes its power-of-two H scale from
__global__ void k_h_bound_word(const float* __restrict__ fcol, const unsigned* __restrict__ scal, unsigned* __restrict__ out) {
    out[0] = __float_as_uint(fcol[8] + fcol[9] * __uint_as_float(scal[1]));
}

__global__ __launch_bounds__(256) void k_attr_absmax_all(const float* __restrict__ T, int64_t rows, int ld, int k0, NodeAttrSel nas,
                                                         const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                         unsigned* __restrict__ amax) {
    // rows = edges.  Attribute tensor: T[edge][d]; node table (nas.kt > 0): slot d of edge e = T[(sel[d] >> 8 ? dst : src)[e]][sel[d] & 255]
    // - the maximum over the EDGES in both cases, i.e. the same number for a table and for the tensor materialised from it
    unsigned m[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
        const int64_t js = nas.kt ? src[r] : r, jd = nas.kt ? dst[r] : r;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (d < k0) m[d] = max(m[d], __float_as_uint(nas.kt ? T[((nas.sel[d] >> 8) ? jd : js) * ld + (nas.sel[d] & 255)] : T[r * ld + d]) & 0x7fffffffu);
    }
    // one atomic per workgroup and slot (round 6: per wave it was 98 k atomics on eight addresses at s=121 - 1.1 ms of a 0.05 ms read)
    __shared__ unsigned wm[4][8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[d] = max(m[d], (unsigned)__shfl_xor((int)m[d], o));
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6][d] = m[d];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const unsigned v = max(max(wm[0][threadIdx.x], wm[1][threadIdx.x]), max(wm[2][threadIdx.x], wm[3][threadIdx.x]));
        if (v) atomicMax(amax + threadIdx.x, v);
    }
}

// First-layer gradients in ONE pass over dU_1:  P[split][k][d] = sum_rows dU[row][k] * H0[row][d]  (d < 8: the gathered
// edge attributes, zero padded) and P[split][K*8 + k] = sum_rows dU[row][k]  (the bias gradient).  The product is
// 1024 x 8 wide and E deep: pure streaming of dU (4 KiB per edge); the fp32 GEMM read it at 1.5 TB/s with 8-column
// tiles and k_colsum read it a second time.  A lane owns four consecutive columns, the four waves interleave rows and
// are summed in wave order; splits are reduced in order by gpde_reduce_splits_kernel (deterministic).
__global__ __launch_bounds__(256) void k_dw_first(const float* __restrict__ dU, const float* __restrict__ H0, int ldh, int rows, int K,
                                                  int splits, float* __restrict__ P) {
    __shared__ float red[4][64][37];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4, split = blockIdx.y;
    const int rps = (rows + splits - 1) / splits;
    const int r_lo = split * rps, r_hi = min(rows, r_lo + rps);
    float acc[4][8], sb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sb[c] = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[c][d] = 0.f;
    }
    if (col < K) {
        const float* p = dU + col;
        auto step = [&](const f32x4& v, const f32x4& ha, const f32x4& hb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sb[c] += v[c];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    acc[c][d] = fmaf(v[c], ha[d], acc[c][d]);
                    acc[c][4 + d] = fmaf(v[c], hb[d], acc[c][4 + d]);
                }
            }
        };
        int r = r_lo + rg;
        for (; r + 12 < r_hi; r += 16) {
            f32x4 v[4], ha[4], hb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = *(const f32x4*)(p + (size_t)(r + 4 * q) * K);
                ha[q] = *(const f32x4*)(H0 + (size_t)(r + 4 * q) * ldh);
                hb[q] = *(const f32x4*)(H0 + (size_t)(r + 4 * q) * ldh + 4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) step(v[q], ha[q], hb[q]);
        }
        for (; r < r_hi; r += 4)
            step(*(const f32x4*)(p + (size_t)r * K), *(const f32x4*)(H0 + (size_t)r * ldh), *(const f32x4*)(H0 + (size_t)r * ldh + 4));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int d = 0; d < 8; ++d) red[rg][lane][c * 9 + d] = acc[c][d];
        red[rg][lane][c * 9 + 8] = sb[c];
    }
    __syncthreads();
    if (rg == 0 && col < K) {
        float* Pw = P + (size_t)split * K * 9;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < 9; ++d) {
                const float t = ((red[0][lane][c * 9 + d] + red[1][lane][c * 9 + d]) + red[2][lane][c * 9 + d]) + red[3][lane][c * 9 + d];
                if (d < 8) Pw[(size_t)(col + c) * 8 + d] = t;
                else Pw[(size_t)K * 8 + col + c] = t;
            }
        }
    }
}

// ordered reduction of k_dw_first's partials into the padded gradient buffers: dW[k][ldw] (first 8 columns), db[k]
__global__ void k_dw_first_reduce(const float* __restrict__ P, int splits, int K, int ldw, float* __restrict__ dW,
                                  float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * 9) return;
    float t = 0.f;
    for (int s = 0; s < splits; ++s) t += P[(size_t)s * K * 9 + i];
    if (i < K * 8) dW[(size_t)(i >> 3) * ldw + (i & 7)] += t;
    else db[i - K * 8] += t;
}

// First hidden layer as a streaming kernel: H1[row][k] = relu(b[k] + sum_{d < 8} W[k][d] * H0[row][d]) (DenseNet.forward's
// first Linear + ReLU, utilities.py:223-227, on the gathered attributes; slots >= k0 are zero columns of both operands).
// 8 FLOP per 4 bytes written: a GEMM shape only in name - the fp32 MFMA GEMM wrote it at 1.8 TB/s.  A lane owns four
// consecutive columns (weights in registers, 16-byte stores: a wave writes 1 KiB of a row), the waves interleave rows.
__global__ __launch_bounds__(256) void k_first_layer(const float* __restrict__ H0, int ld0, const float* __restrict__ Wp, int ldw,
                                                     const float* __restrict__ bp, int rows, int K, float* __restrict__ H1) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4;
    if (col >= K) return;
    float w[4][8], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        b[c] = bp[col + c];
        const f32x4 w0 = *(const f32x4*)(Wp + (size_t)(col + c) * ldw), w1 = *(const f32x4*)(Wp + (size_t)(col + c) * ldw + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) { w[c][d] = w0[d]; w[c][4 + d] = w1[d]; }
    }
    const int rps = (rows + gridDim.y - 1) / gridDim.y;
    const int r_lo = blockIdx.y * rps, r_hi = min(rows, r_lo + rps);
    for (int r = r_lo + rg; r < r_hi; r += 4) {
        const f32x4 a0 = *(const f32x4*)(H0 + (size_t)r * ld0), a1 = *(const f32x4*)(H0 + (size_t)r * ld0 + 4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t = b[c];
#pragma unroll
            for (int d = 0; d < 4; ++d) t = fmaf(w[c][d], a0[d], t);
#pragma unroll
            for (int d = 0; d < 4; ++d) t = fmaf(w[c][4 + d], a1[d], t);
            o[c] = fmaxf(t, 0.f);
        }
        *(f32x4*)(H1 + (size_t)r * K + col) = o;
    }
}

// ---- per-edge backward through the aggregation -----------------------------------------------------
//   dU[e][n]  = (sum_c x_j[c] dZ_i[c][n]) * (H[e][n] > 0)
//   dx[j][c] += sum_n dZ_i[c][n] H[e][n] + dS_i[c]                       (fp32 atomics)
// One wave per tile of 32 CSR slots; destination segments inside a tile are handled by masking,
// exactly like the forward aggregation.  dZ: [nodes of chunk][64][K2P].
constexpr int EB_XS = 65, EB_HS = 65, EB_NC = 64;   // 64-column chunks: 66 KiB of LDS per workgroup,
                                                       // two workgroups (two waves per SIMD) per CU
struct EdgeBwdArgs {
    const float* x; const int32_t* rowptr; const int32_t* src; const int32_t* dst;
    const float* dZ; const float* dS; const float* H; float* dU; float* dx;
    int e0, e1, n0, K2P;
    float* dxe;       // ordered mode: per-edge contributions [e1 - e0][64] instead of atomics on dx
};
__global__ __launch_bounds__(256, 2) void gpde_edge_bwd_kernel(EdgeBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Xs = smem + wave * (32 * EB_XS + 32 * EB_HS);
    float* Hs = Xs + 32 * EB_XS;
    const int t0 = a.e0 + (blockIdx.x * 4 + wave) * 32;
    if (t0 >= a.e1) return;
    const int t1 = min(t0 + 32, a.e1);

    // x_j rows of the tile -> LDS (zero rows past the end)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int er = (lane >> 4) + 4 * i, e = t0 + er;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e < t1) v = *(const f32x4*)&a.x[(size_t)a.src[e] * GP_W + (lane & 15) * 4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Xs[er * EB_XS + (lane & 15) * 4 + q] = v[q];
    }
    f32x16 dxa[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[cb][r] = 0.f;

    for (int nc = 0; nc < a.K2P; nc += EB_NC) {
        // H tile [32][64] -> LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = i * 64 + lane, er = f >> 4, q4 = f & 15, e = t0 + er;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < t1) v = *(const f32x4*)&a.H[(size_t)(e - a.e0) * a.K2P + nc + q4 * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) Hs[er * EB_HS + q4 * 4 + q] = v[q];
        }
        f32x16 dh[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dh[nb][r] = 0.f;

        int e_seg = t0;
        while (e_seg < t1) {
            const int node = a.dst[e_seg];
            const int seg_end = min(a.rowptr[node + 1], t1);
            const float* dZi = a.dZ + (size_t)(node - a.n0) * GP_W * a.K2P + nc;
            const bool mine = (t0 + l31 >= e_seg) && (t0 + l31 < seg_end);     // lane's edge in segment
            // dH[e][n] += sum_c x[e][c] dZ_i[c][n]      (A = x from LDS, B = dZ_i from L2); skipped without a dU output
            if (a.dU) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float av = mine ? Xs[l31 * EB_XS + c] : 0.f;
                    const float* zp = dZi + (size_t)c * a.K2P + l31;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) dh[nb] = mfma32(av, zp[nb * 32], dh[nb]);
                }
            }
            // dXg^T[c][e] += sum_n dZ_i[c][n] H[e][n]   (A = dZ_i rows c, B = H^T from LDS)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float bv = mine ? Hs[l31 * EB_HS + n] : 0.f;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        dxa[cb] = mfma32(dZi[(size_t)(cb * 32 + l31) * a.K2P + n], bv, dxa[cb]);
                }
            e_seg = seg_end;
        }
        // dU = dH * (H > 0)
        if (a.dU)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = (r & 3) + 8 * (r >> 2) + 4 * h, e = t0 + er;
                if (e < t1) {
                    const float v = (Hs[er * EB_HS + nb * 32 + l31] > 0.f) ? dh[nb][r] : 0.f;
                    a.dU[(size_t)(e - a.e0) * a.K2P + nc + nb * 32 + l31] = v;
                }
            }
    }
    // dx_j += dXg + dS_i
    const int e = t0 + l31;
    if (e < t1) {
        const int j = a.src[e], node = a.dst[e];
        const float* ds = a.dS + (size_t)(node - a.n0) * GP_W;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (a.dxe) a.dxe[(size_t)(e - a.e0) * GP_W + c] = dxa[cb][r] + ds[c];
                else atomicAdd(&a.dx[(size_t)j * GP_W + c], dxa[cb][r] + ds[c]);
            }
    }
}

// ---- per-edge backward, staged version (graphs with in-degree >= ~32) --------------------------------------
// Same two products as gpde_edge_bwd_kernel, but nothing is fetched per MFMA.  A workgroup owns 128
// consecutive CSR slots (4 tiles of 32, one per wave); in a radius graph these almost always belong
// to ONE destination node, so the dZ_i chunk is staged ONCE per workgroup and shared:
//   * hidden columns in chunks of 32; per chunk the dZ rows of up to two destination nodes
//     ([2][64 c][32 n], 16 KiB, by all four waves) and each wave's H tile ([32 e][32 n], 4 KiB) come
//     global -> LDS by DMA, double-buffered, one s_barrier per chunk (64 fp32 MFMAs = 4096 cycles);
//     16-byte units are XOR-swizzled by (row & 7) on the way in (a DMA lane may fetch any unit), so
//     ds_read_b128 along a row and ds_read_b32 down a column are both conflict-free without padding;
//   * x_j rows are the A operand of product (1) for every chunk: 32 registers per lane, loaded once;
//   * product (1)  dH[e][n] = sum_c x[e][c] dZ[c][n]   A = x (registers), B = dZs (ds_read_b32)
//     product (2)  dXg[e][c] = sum_n H[e][n] dZ[c][n]   A = Hs, B = dZs, both ds_read_b128 with the k
//     permutation n = 8q + 4h + t;  D rows = edges in both, so dU rows are written as 128-byte runs and
//     the dx atomics of one edge are 32 consecutive floats;
//   * a 128-slot group spanning more than two destinations repeats the chunk loop per node pair (rare
//     at in-degree >= 32; low-degree graphs use gpde_edge_bwd_kernel).
constexpr int EB2_NC = 32;                          // hidden columns per chunk
constexpr int EB2_DZ = 2 * 64 * EB2_NC;             // floats per dZ buffer (two nodes)
constexpr int EB2_H = 32 * EB2_NC;                  // floats per wave H buffer

__device__ __forceinline__ void eb2_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#ifdef GPDE_EB2_TIMING     // developer probe (scripts/eb2_timing.py): clock64 ticks summed over waves: group prologue, waiting at the top of a
                           // step (DMA + barrier), the step's products and stores, group epilogue; steps, groups
__device__ unsigned long long gpde_eb2_tm[8];
#define EB2_MARK(acc) do { const long long tm1_ = clock64(); acc += tm1_ - tm0_; tm0_ = tm1_; } while (0)
#else
#define EB2_MARK(acc) do { } while (0)
#endif

__global__ __launch_bounds__(256, 2) void gpde_edge_bwd2_kernel(EdgeBwdArgs a) {
#ifdef GPDE_EB2_TIMING
    long long tm_pro = 0, tm_wait = 0, tm_work = 0, tm_epi = 0, tm0_ = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dZs = smem;                                      // [2 buf][2 node][64][32]
    float* Hs_all = smem + 2 * EB2_DZ;                      // [4 waves][2 buf][32][32]
    int* idx_all = (int*)(Hs_all + 4 * 2 * EB2_H);          // [4 waves][src 32 | dst 32]
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Hs = Hs_all + wave * 2 * EB2_H;
    int* src_s = idx_all + wave * 64;
    int* dst_s = src_s + 32;

    // Workgroup b runs on XCD b % 8: XCD x takes the x-th CONTIGUOUS eighth of the groups, so that the groups of one
    // destination node (3.2 on average at s=121, each reading the node's 256 KiB of dZ) follow each other on one L2 instead
    // of fetching it into three (the grid is padded to a multiple of 8; FETCH_SIZE before: 41 GB per backward for 24 GB of H)
    const int grp = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int g0 = a.e0 + grp * 128;                        // first slot of the group
    if (g0 >= a.e1) return;                                 // padding workgroups (whole workgroup: before any barrier)
    const int g1 = min(g0 + 128, a.e1);
    const int t0 = g0 + wave * 32;                          // this wave's tile (may be empty)
    const int e_last = a.e1 - 1;
    const int nA0 = a.dst[g0], nB0 = a.dst[g1 - 1];         // destination range of the group

    // lane-as-edge data (A operands): destination and x_j row of edge t0 + l31
    const int eL = t0 + l31;
    const bool vL = eL < a.e1;
    const int nodeL = vL ? a.dst[eL] : -1;
    const int srcL = vL ? a.src[eL] : 0;
    if (h == 0) { src_s[l31] = srcL; dst_s[l31] = nodeL; }
    f32x4 xr[8];                                            // x[e = l31][c = 8q + 4h + t]
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vL) v = *(const f32x4*)&a.x[(size_t)srcL * GP_W + 8 * q + 4 * h];
        xr[q] = v;
    }
    f32x16 dxa[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[cb][r] = 0.f;

    const int NCH = a.K2P / EB2_NC;
    const int npass = (nB0 - nA0) / 2 + 1;
    const int nit = npass * NCH;
    // DMA lane roles: one instruction = 8 rows x 8 units of 16 B; LDS position (rr, p) holds unit p ^ rr
    const int rr = lane >> 3, uq = (lane & 7) ^ rr;
    auto issue = [&](int it) {
        const int pass = it / NCH, nc = (it - pass * NCH) * EB2_NC, buf = it & 1;
        const int nodeA = nA0 + 2 * pass;
#pragma unroll
        for (int nd = 0; nd < 2; ++nd) {
            if (nd == 1 && nodeA + 1 > nB0) break;          // no second node in this pass: its buffer is never read (anyB is false)
            const int node = nodeA + nd;
            const float* g = a.dZ + ((size_t)(node - a.n0) * GP_W) * a.K2P + nc + uq * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int blk = wave * 2 + i;               // rows 8 blk .. 8 blk + 7
                eb2_dma16(g + (size_t)(blk * 8 + rr) * a.K2P, dZs + buf * EB2_DZ + nd * 64 * EB2_NC + blk * 256);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = min(t0 + i * 8 + rr, e_last);
            eb2_dma16(a.H + (size_t)(e - a.e0) * a.K2P + nc + uq * 4, Hs + buf * EB2_H + i * 256);
        }
    };

    issue(0);
    EB2_MARK(tm_pro);
    for (int it = 0; it < nit; ++it) {
        const int pass = it / NCH, nc = (it - pass * NCH) * EB2_NC, buf = it & 1;
        const int nodeA = nA0 + 2 * pass, nodeB = nodeA + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        EB2_MARK(tm_wait);
        if (it + 1 < nit) issue(it + 1);
        const bool inA = nodeL == nodeA, inB = nodeL == nodeB;
        const bool anyA = __builtin_amdgcn_ballot_w64(inA) != 0, anyB = __builtin_amdgcn_ballot_w64(inB) != 0;
        if (anyA || anyB) {
        const float* hb = Hs + buf * EB2_H;
        // H of lane-as-edge (A operand of product 2): units 2q' + h of row l31
        f32x4 hv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) hv[q] = *(const f32x4*)&hb[l31 * EB2_NC + (((2 * q + h) ^ (l31 & 7)) << 2)];
        f32x16 dh;                       // (two accumulators for product (1) were tried in round 4: no change - the chain is not the limit)
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = 0.f;
#pragma unroll
        for (int nd = 0; nd < 2; ++nd) {
            if (nd == 0 ? !anyA : !anyB) continue;
            const bool in = nd == 0 ? inA : inB;
            const float* zb = dZs + buf * EB2_DZ + nd * 64 * EB2_NC;
            // product (1): k = c = 8q + 4h + t   (skipped without a dU output: the light pass of the depth-deferred backward)
            if (a.dU) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c = 8 * q + 4 * h + t;
                    const float bv = zb[c * EB2_NC + ((((l31 >> 2) ^ (c & 7)) << 2) | (l31 & 3))];
                    dh = mfma32(in ? xr[q][t] : 0.f, bv, dh);
                }
            }
            // product (2): k = n = 8q + 4h + t, B row c = l31 (+32)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int c = cb * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 zv = *(const f32x4*)&zb[c * EB2_NC + (((2 * q + h) ^ (c & 7)) << 2)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) dxa[cb] = mfma32(in ? hv[q][t] : 0.f, zv[t], dxa[cb]);
                }
            }
        }
        // dU rows of this pass's nodes: dH * (H > 0); lane = column nc + l31, rows (r, h)
        if (a.dU)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int er = (r & 3) + 8 * (r >> 2) + 4 * h, e = t0 + er;
            const int nd_e = dst_s[er];
            if (e < a.e1 && (nd_e == nodeA || nd_e == nodeB)) {
                const float hval = hb[er * EB2_NC + ((((l31 >> 2) ^ (er & 7)) << 2) | (l31 & 3))];
                a.dU[(size_t)(e - a.e0) * a.K2P + nc + l31] = hval > 0.f ? dh[r] : 0.f;
            }
        }
        }
        EB2_MARK(tm_work);
    }
    // dx_j += dXg + dS_i : lane = channel c = l31 (+32), rows = edges
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int er = (r & 3) + 8 * (r >> 2) + 4 * h, e = t0 + er;
        if (e < a.e1) {
            const int j = src_s[er], node = dst_s[er];
            const float* ds = a.dS + (size_t)(node - a.n0) * GP_W;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int c = cb * 32 + l31;
                if (a.dxe) a.dxe[(size_t)(e - a.e0) * GP_W + c] = dxa[cb][r] + ds[c];
                else atomicAdd(&a.dx[(size_t)j * GP_W + c], dxa[cb][r] + ds[c]);
            }
        }
    }
#ifdef GPDE_EB2_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EB2_MARK(tm_epi);
    if (lane == 0) {
        atomicAdd(&gpde_eb2_tm[0], (unsigned long long)tm_pro);
        atomicAdd(&gpde_eb2_tm[1], (unsigned long long)tm_wait);
        atomicAdd(&gpde_eb2_tm[2], (unsigned long long)tm_work);
        atomicAdd(&gpde_eb2_tm[3], (unsigned long long)tm_epi);
        atomicAdd(&gpde_eb2_tm[4], (unsigned long long)nit);
        atomicAdd(&gpde_eb2_tm[5], 1ull);
    }
#endif
}

#ifdef GPDE_EB2_TIMING
}  // namespace
extern "C" GPDE_API int gpde_debug_eb2_timing(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gpde_eb2_tm), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gpde_eb2_tm), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
namespace {
#endif

// Ordered mode: dx[j] += sum of the per-edge contributions of the out-edges of j that lie in this chunk's slot range
// [e0, e1), in ascending slot order (src_slots is ascending inside a source: the chunk's part is one sub-range).
// One wave per source node, lane = channel: a single owner per dx element, no atomics -> bit-reproducible.
// `nparts` > 1 (the one-pass kernel): dxe holds nparts partial rows per edge, part s at dxe + s * part_stride; an edge's
// contribution is their sum in ascending s (fixed order).
__global__ __launch_bounds__(256) void k_dx_reduce(const float* __restrict__ dxe, const int32_t* __restrict__ srp,
                                                   const int32_t* __restrict__ ssl, int n_nodes, int e0, int e1,
                                                   float* __restrict__ dx, int nparts, size_t part_stride, int init) {
    // `init`: dx is not initialised - the sum starts at 0 and a node without out-edges in the range gets 0 (the one-chunk callers:
    // saves the zero-fill launch; 0 + v is exact, the bits are those of the zero-filled form)
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_nodes) return;
    int lo = srp[j], hi = srp[j + 1];
    const int end = hi;
    while (lo < hi) {                                   // first position with slot >= e0
        const int mid = (lo + hi) >> 1;
        if (ssl[mid] < e0) lo = mid + 1; else hi = mid;
    }
    int p = lo;
    if (p >= end || ssl[p] >= e1) { if (init) dx[(size_t)j * GP_W + lane] = 0.f; return; }
    float acc = init ? 0.f : dx[(size_t)j * GP_W + lane];            // continue the running sum: the result does not depend on the chunking
    for (; p + 8 <= end; p += 8) {
        int sl[8];
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sl[q] = ssl[p + q];
        if (sl[7] >= e1) break;
        if (nparts == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dxe[(size_t)(sl[q] - e0) * GP_W + lane];
        } else {
            // an edge's value = its partial rows summed in part order (8 edges x nparts loads in flight)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dxe[(size_t)(sl[q] - e0) * GP_W + lane];
            for (int s_ = 1; s_ < nparts; ++s_) {
                float w[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) w[q] = dxe[(size_t)s_ * part_stride + (size_t)(sl[q] - e0) * GP_W + lane];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += w[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    for (; p < end; ++p) {
        const int sl = ssl[p];
        if (sl >= e1) break;
        float v = dxe[(size_t)(sl - e0) * GP_W + lane];
        for (int s_ = 1; s_ < nparts; ++s_) v += dxe[(size_t)s_ * part_stride + (size_t)(sl - e0) * GP_W + lane];
        acc += v;
    }
    dx[(size_t)j * GP_W + lane] = acc;
}

size_t al(size_t v) { return (v + 255) / 256 * 256; }
unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

// K splits of the dW_2 GEMM (contraction over the chunk's edges): 4 row quads x 8 splits x 8 slices = 256 workgroups up to
// 786 k edges, then 8 more per 786 k so that no fp32 accumulator runs over more than ~100 k edges: with the whole s=121
// graph as ONE chunk (5.9 M edges) 8 splits left 741 k-term chains and dW_2 3e-5 away from the 10-chunk result.
int tn_ksplits(int64_t rows, int n_out = 1024, int n_in = 1024) {
    int64_t g = (rows + 786431) / 786432;
    if (g < 1) g = 1;
    if (g > 16) g = 16;
    int ks = (int)(8 * g);
    // narrow kernel MLPs (the MGKN levels: a 256 x 256 gradient is four tiles): more splits until ~256 workgroups run - with 8
    // splits the 131 k edges of config 4's finest level were 16 k-row chains in 32 workgroups (588 us; 64 splits: round 6)
    const int tiles = ((n_out + 127) / 128) * ((n_in + 127) / 128);
    while (ks < 128 && tiles * ks < 256 && rows / (ks * 2) >= 512) ks *= 2;
    return ks;
}

struct BwdPlan {
    int n_layers, nh;                 // nh = hidden layers = n_layers - 1
    int KP[GPDE_MAX_LAYERS + 1];      // padded widths: KP[0] = pad32(k0), KP[l] = pad128(k_l), l < n_layers
    int K2P;
    int64_t Ec, Nc;                   // edges / nodes per chunk
    size_t off_wp[GPDE_MAX_LAYERS], off_bp[GPDE_MAX_LAYERS], off_dwp[GPDE_MAX_LAYERS], off_dbp[GPDE_MAX_LAYERS];
    size_t off_w3p, off_dw3p, off_b3, off_db3, off_part, part_floats, off_pack, pack_bytes;
    size_t off_H[GPDE_MAX_LAYERS + 1], off_dU[2], off_Z, off_dZ, off_gT, off_S, off_dS;
    size_t off_w2t, off_w2ts, off_ucol2, off_rowsc;   // dU_1 on split f16: W2^T fp32, its split tile image, 2^-t per k1, row scales
    bool f16s_du1;
    size_t off_dxe;                   // per-edge dx contributions of a chunk (ordered mode)
    bool f16s_dw2; size_t off_tnws;   // dW_2 on the split-f16 GEMM: transposed dU_2 + split image of H_1^T per edge chunk
    size_t off_dubits;                // column maxima of |dU_2| as bit patterns [kmax]
    size_t off_maskbits;              // ReLU mask of the first hidden layer as bits [Ec][KP1 / 32] (H_1 itself is not materialised)
    // depth-deferred form (gpde_nnconv_bwd_deferred): L = n_defer layers share one pass over the hidden layers
    int L, Lp;                        // Lp = K / 64 of the gather GEMM: L rounded up to an even count >= 4 (zero layers)
    size_t off_dzstack, off_dzimg, off_nbits, off_nscale, off_tiles, off_xsc;
    size_t off_dzun;                  // per chunk node un-scale of its split dZ image (gpde_edge_bwd3.hip)
    size_t off_tcs, off_tcm;          // per 32-slot tile column sums / max bits of dU_2 [Ec / 32 + 1][KP2] (gpde_edge_bwd3.hip -> dW_2 GEMM)
    size_t off_amax8;                 // [8] words: bits of max |attribute slot d| over ALL edges of the call (in-kernel first layer: one set of scales for every chunk)
    size_t off_xs;                    // [N][64] words: x as split-f16 pairs for the Z re-aggregation (gpde_zagg_kernel<true>); off_scal[0] = its scale, off_scal[2] = the H bound
    size_t off_scal;                  // [2] words: bits of max_e B_e for the one-pass kernel's global H scale (gpde_launch_attr_bound)
    size_t total;
    size_t one_chunk;                 // workspace bytes with which everything is one chunk
};

// h_given: the last hidden activations of EVERY edge come from the caller (kept by the forward): their per-chunk buffer
// (KP[n-1] floats per edge) is not part of the workspace - the same bytes hold more edges per chunk
int make_bwd_plan(int64_t N, int64_t E, int n_layers, const int32_t* dims, size_t ws_bytes, bool sizing,
                  BwdPlan* P, int n_defer = 0, bool h_given = false) {
    if (n_layers < 2 || n_layers > GPDE_MAX_LAYERS) { gpde_set_error("kernel MLP must have 2..%d Linear layers", GPDE_MAX_LAYERS); return GPDE_EUNSUPPORTED; }
    if (dims[n_layers] != GP_W * GP_W) { gpde_set_error("last layer must emit %d values", GP_W * GP_W); return GPDE_EUNSUPPORTED; }
    P->n_layers = n_layers; P->nh = n_layers - 1;
    P->KP[0] = gp_round_up(dims[0], 32);
    int kmax = 0;
    size_t hsum = P->KP[0];
    for (int l = 1; l < n_layers; ++l) { P->KP[l] = gp_round_up(dims[l], 128); kmax = kmax > P->KP[l] ? kmax : P->KP[l]; hsum += P->KP[l]; }
    P->K2P = P->KP[n_layers - 1];
    if (h_given) hsum -= P->KP[n_layers - 1];
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    size_t wmax = 0;
    for (int l = 1; l < n_layers; ++l) {
        const size_t w = (size_t)P->KP[l] * P->KP[l - 1];
        P->off_wp[l] = take(w); P->off_bp[l] = take(P->KP[l]);
        P->off_dwp[l] = take(w); P->off_dbp[l] = take(P->KP[l]);
        wmax = wmax > w ? wmax : w;
    }
    const size_t w3 = (size_t)GP_W * GP_W * P->K2P;
    P->off_w3p = take(w3); P->off_b3 = take(GP_W * GP_W);
    P->off_dw3p = take(w3); P->off_db3 = take(GP_W * GP_W);      // adjacent: zeroed by one launch
    const int ks_max = tn_ksplits(E, P->KP[n_layers - 1], n_layers >= 3 ? P->KP[n_layers - 2] : 1024);
    const int max_splits = ks_max > 16 ? ks_max : 16;
    P->part_floats = (size_t)max_splits * (wmax > 4096 ? wmax : 4096);
    if (P->part_floats < (size_t)64 * GP_W * GP_W) P->part_floats = (size_t)64 * GP_W * GP_W;      // 64 node-range splits of a 64 x 64 output (gemm_tn_acc)
    P->off_part = take(P->part_floats);
    // packed MLP image for the fused f16-split recompute of the last hidden layer (3-Linear kernels)
    P->pack_bytes = (n_layers == 3 && dims[0] + 1 <= 8) ? gpde_mlp_pack_bytes(n_layers, dims) : 0;
    P->off_pack = take(P->pack_bytes / 4);
    // dU_1 = (dU_2 . W_2) (.) [H_1 > 0] on split-f16 MFMA (gpde_gemm_f16s.hip): 3-Linear kernels, k2 padded >= 256
    P->f16s_du1 = n_layers == 3 && gpde_gemm_f16s_supported(1, P->KP[1], P->KP[2], P->KP[2]);
    P->off_w2t = P->off_w2ts = P->off_ucol2 = 0;
    if (P->f16s_du1) {
        P->off_w2t = take((size_t)P->KP[1] * P->KP[2]);
        P->off_w2ts = take((size_t)P->KP[1] * P->KP[2]);
        P->off_ucol2 = take(P->KP[1]);
    }
    P->f16s_dw2 = P->f16s_du1 && P->KP[2] % 64 == 0 && P->KP[1] % GP_TN == 0;
    P->L = n_defer; P->Lp = n_defer > 0 ? (n_defer + 1) / 2 * 2 : 0;
    if (n_defer > 0 && P->Lp < 4) P->Lp = 4;
    P->off_xsc = take(n_defer > 0 ? (size_t)2 * (N > 0 ? N : 1) : 1);      // per source node row scales of the layer-input stack
    P->off_scal = take(4);
    P->off_xs = take(P->pack_bytes ? (size_t)(N > 0 ? N : 1) * GP_W : 1);
    P->off_amax8 = take(16);
    const size_t fixed = off;
    // per-chunk buffers: per edge (hsum + 2*kmax) floats (+ KP1 + KP2 for the transposed operands of dW_2), per node
    // (2*64*K2P + 3*64) floats
    const size_t tn_edge = P->f16s_dw2 ? (size_t)P->KP[1] + P->KP[2] : 0;
    const size_t per_edge = (hsum + 2 * (size_t)kmax + (P->f16s_du1 ? 2 : 0) + tn_edge + GP_W) * 4 + (P->f16s_dw2 ? P->KP[1] / 8 + (P->KP[2] / 64 + 2) * 4 + 8 + P->KP[2] / 4 + 32 /* attribute operands of the in-kernel first layer */ : 0) + (n_defer > 0 ? 1 : 0),
                 per_node = ((size_t)2 * GP_W * P->K2P + 3 * GP_W + 1) * 4 +
                            (n_defer > 0 ? (size_t)(P->L + P->Lp) * GP_W * P->K2P * 4 + 64 : 0);   // dZ of every deferred layer (fp32) + the node's split image + tile records
    int64_t Ec, Nc;
    // alignment of the per-chunk buffers below + the K padding of the transposed operands
    const size_t slack = 64 * 256 + (P->f16s_dw2 ? gpde_gemm_f16s_tn_ws_floats(0, P->KP[2], P->KP[1], ks_max) * 4 : 0);
    P->one_chunk = fixed + (size_t)(E > 0 ? E : 1) * per_edge + (size_t)(N > 0 ? N : 1) * per_node + slack + (1 << 20);
    if (sizing) {
        Ec = (int64_t)(((size_t)(P->f16s_dw2 ? 18 : 12) << 30) / per_edge); Nc = (int64_t)(((size_t)8 << 30) / per_node);
    } else if (ws_bytes >= fixed + (size_t)(E > 0 ? E : 1) * per_edge + (size_t)(N > 0 ? N : 1) * per_node + slack) {
        Ec = E; Nc = N;                  // everything in one chunk
    } else {
        if (ws_bytes < fixed + (1 << 20)) { gpde_set_error("gpde_nnconv_bwd: workspace %zu bytes too small (%zu fixed)", ws_bytes, fixed); return GPDE_EWORKSPACE; }
        const size_t avail = ws_bytes - fixed - slack;
        // nodes first (at most 40 %); what they leave - all of it when every node fits - goes to the edges
        Nc = (int64_t)(avail * 4 / 10 / per_node);
        if (Nc > N) Nc = N;
        if (Nc < 1) Nc = 1;
        Ec = (int64_t)((avail - (size_t)Nc * per_node) / per_edge);
    }
    if (Ec > E) Ec = E;
    if (Nc > N) Nc = N;
    if (Ec < 1) Ec = 1;
    if (Nc < 1) Nc = 1;
    P->Ec = Ec; P->Nc = Nc;
    P->off_H[0] = take((size_t)Ec * P->KP[0]);
    for (int l = 1; l < n_layers; ++l) P->off_H[l] = take(h_given && l == n_layers - 1 ? 1 : (size_t)Ec * P->KP[l]);
    P->off_dU[0] = take((size_t)Ec * kmax); P->off_dU[1] = take((size_t)Ec * kmax);
    P->off_Z = take((size_t)Nc * GP_W * P->K2P); P->off_dZ = take((size_t)Nc * GP_W * P->K2P);
    P->off_gT = take((size_t)Nc * GP_W); P->off_S = take((size_t)Nc * GP_W); P->off_dS = take((size_t)Nc * GP_W);
    P->off_rowsc = take(P->f16s_du1 ? (size_t)2 * Ec : 1);
    P->off_dxe = take((size_t)Ec * GP_W);
    P->off_tnws = take(P->f16s_dw2 ? gpde_gemm_f16s_tn_ws_floats((int)Ec, P->KP[2], P->KP[1], tn_ksplits(Ec, P->KP[2], P->KP[1])) : 1);
    P->off_dubits = take((size_t)(kmax > 0 ? kmax : 1));
    P->off_maskbits = take(P->f16s_dw2 ? (size_t)Ec * (P->KP[1] / 32) : 1);
    P->off_dzstack = take(n_defer > 0 ? (size_t)P->L * Nc * GP_W * P->K2P : 1);
    P->off_dzimg = take(n_defer > 0 ? (size_t)P->Lp * Nc * GP_W * P->K2P : 1);
    P->off_nbits = take(n_defer > 0 ? (size_t)Nc : 1);
    P->off_nscale = take(n_defer > 0 ? (size_t)2 * Nc : 1);
    P->off_tiles = take(n_defer > 0 ? (size_t)4 * (Ec / 256 + Nc + 8) : 1);
    P->off_dzun = take((size_t)Nc);
    P->off_tcs = take(P->f16s_dw2 ? (size_t)(Ec / 32 + 1) * P->KP[2] : 1);
    P->off_tcm = take(P->f16s_dw2 ? (size_t)(Ec / 32 + 1) * P->KP[2] : 1);
    P->total = off + 256 + (sizing ? slack : 0);
    if (!sizing && P->total > ws_bytes) { gpde_set_error("gpde_nnconv_bwd: internal plan %zu > workspace %zu", P->total, ws_bytes); return GPDE_EWORKSPACE; }
    return GPDE_OK;
}

GpdeGemmArgs gemm0() {
    GpdeGemmArgs g{};
    g.a_kcontig = 1; g.b_kcontig = 1; g.batches = 1; g.splits = 1;
    return g;
}

// C (+)= A^T . B over `rows` rows, split-K with ordered partial reduction
int gemm_tn_acc(const float* A, int lda, int M, const float* B, int ldb, int Ncols, int rows, float* C,
                int ldc_dense, float* part, size_t part_floats, int accumulate, hipStream_t st) {
    const int tiles = ((M + 127) / 128) * ((Ncols + 127) / 128);
    int splits = 1;
    // skinny outputs (dW_1: 1024 x 8) are pure streaming of the tall operand: enough splits to put ~1024 workgroups
    // on it (16 splits = 128 workgroups read dU_1 at 0.8 TB/s)
    while (splits < 256 && tiles * splits < 1024 && rows / (splits * 2) >= (tiles == 1 ? 64 : 256)) splits *= 2;
    const size_t cn = (size_t)M * Ncols;
    if ((size_t)splits * cn > part_floats) splits = (int)(part_floats / cn) ? (int)(part_floats / cn) : 1;
    GpdeGemmArgs g = gemm0();
    g.A = A; g.lda = lda; g.a_kcontig = 0; g.B = B; g.ldb = ldb; g.b_kcontig = 0;
    g.M = M; g.N = Ncols; g.K = rows; g.ldc = Ncols;
    (void)ldc_dense;
    if (splits == 1) {            // one partial: straight into C (C + P is what either accumulate form computes)
        g.C = C; g.accumulate = accumulate ? 1 : 0;
        return gpde_launch_gemm(g, st);
    }
    g.C = part; g.splits = splits; g.strideSplit = cn;
    int rc = gpde_launch_gemm(g, st);
    if (rc != GPDE_OK) return rc;
    return gpde_launch_reduce_splits(part, cn, splits, cn, C, accumulate, st);
}


// ---- node-side terms of update() in ONE pass (round 6) ---------------------------------------------------------------------
// out_i += x_i . root + bias (nn_conv.py:277-282): dx_i += g_i . root^T, droot = X^T g, dbias = colsum g.  Rounds 2-5 ran three
// generic 128 x 128-tile GEMM / column-sum launches plus two partial reductions per call - 60 us of a 52-call MGKN training
// step's every call, for 64-wide operands with K = 64.  Here a workgroup walks a contiguous range of 64-node strips: x and g
// strips and root staged in LDS, each wave owns one 32 x 32 block of dx (32 v_mfma_f32_32x32x2_f32 per strip, added to dx in
// place) and one of droot (accumulated over the workgroup's strips in registers); partials [wg][64 x 64 + 64] summed in
// workgroup order by k_node_terms_reduce.  fp32 MFMA: the arithmetic class of the GEMMs it replaces.
constexpr int NT_LD = GP_W + 1;
struct NodeTermsArgs {
    const float* x; const float* g; const float* root; float* dx; float* part;
    int N, strips_per_wg, do_dx, do_root;
};
__global__ __launch_bounds__(256) void k_node_terms(NodeTermsArgs a) {
    __shared__ float gs[GP_W * NT_LD], xs[GP_W * NT_LD], rs[GP_W * NT_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, bm = wave >> 1, bn = wave & 1;
    if (a.do_dx)
        for (int f = tid; f < GP_W * GP_W; f += 256) rs[(f >> 6) * NT_LD + (f & 63)] = a.root[f];
    f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = 0.f;
    float bsum = 0.f;
    const int s_lo = blockIdx.x * a.strips_per_wg, s_hi = min(s_lo + a.strips_per_wg, (a.N + GP_W - 1) / GP_W);
    for (int sidx = s_lo; sidx < s_hi; ++sidx) {
        const int i0 = sidx * GP_W;
        __syncthreads();                      // the previous strip's LDS reads are done (and rs is written, first round)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f4 = tid + 256 * k, row = f4 >> 4, c4 = (f4 & 15) * 4;
            f32x4 gv = {0.f, 0.f, 0.f, 0.f}, xv = gv;
            if (i0 + row < a.N) {
                gv = *(const f32x4*)(a.g + (size_t)(i0 + row) * GP_W + c4);
                if (a.do_root) xv = *(const f32x4*)(a.x + (size_t)(i0 + row) * GP_W + c4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { gs[row * NT_LD + c4 + j] = gv[j]; xs[row * NT_LD + c4 + j] = xv[j]; }
        }
        __syncthreads();
        if (tid < GP_W) {
            float sm = 0.f;
#pragma unroll 8
            for (int i = 0; i < GP_W; ++i) sm += gs[i * NT_LD + tid];
            bsum += sm;
        }
        if (a.do_dx) {
            // dx[i][c] += sum_o g[i][o] root[c][o]: A[m = i][k = o], B[k = o][n = c]
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < GP_W / 2; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gs[(bm * 32 + l31) * NT_LD + 2 * t + h], rs[(bn * 32 + l31) * NT_LD + 2 * t + h], acc, 0, 0, 0);
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                old[r] = i < a.N ? a.dx[(size_t)i * GP_W + bn * 32 + l31] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < a.N) a.dx[(size_t)i * GP_W + bn * 32 + l31] = old[r] + acc[r];
            }
        }
        if (a.do_root) {
            // droot[c][o] += sum_i x[i][c] g[i][o]: A[m = c][k = i], B[k = i][n = o]
#pragma unroll
            for (int t = 0; t < GP_W / 2; ++t)
                racc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * t + h) * NT_LD + bm * 32 + l31], gs[(2 * t + h) * NT_LD + bn * 32 + l31], racc, 0, 0, 0);
        }
    }
    float* P = a.part + (size_t)blockIdx.x * (GP_W * GP_W + GP_W);
    if (a.do_root) {
#pragma unroll
        for (int r = 0; r < 16; ++r) P[(bm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * GP_W + bn * 32 + l31] = racc[r];
    }
    if (tid < GP_W) P[GP_W * GP_W + tid] = bsum;
}
// droot / dbias = (their old value, accumulate) + the workgroups' partials in order
__global__ void k_node_terms_reduce(const float* __restrict__ part, int nwg, float* __restrict__ droot, float* __restrict__ dbias,
                                    int acc_root, int acc_bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= GP_W * GP_W + GP_W) return;
    float* out = i < GP_W * GP_W ? (droot ? droot + i : nullptr) : (dbias ? dbias + (i - GP_W * GP_W) : nullptr);
    if (!out) return;
    float sm = 0.f;
    for (int k = 0; k < nwg; ++k) sm += part[(size_t)k * (GP_W * GP_W + GP_W) + i];
    const int acc = i < GP_W * GP_W ? acc_root : acc_bias;
    *out = acc ? *out + sm : sm;
}

// node-side terms of update() (nn_conv.py:277-282): dx += g root^T, droot = X^T g, dbias = colsum g
int bwd_node_terms(const float* x, int N, const float* root, const float* grad_out, float* dx, float* grad_root, float* grad_bias,
                   float* part, size_t part_floats, hipStream_t st, int acc_root = 0, int acc_bias = 0) {
    int rc;
    const size_t rec = (size_t)GP_W * GP_W + GP_W;
    if (!gpde_switches().bwd_node_terms_gemm && N > 0 && (grad_root || grad_bias || (root && dx)) && (!grad_root || x) && part_floats >= rec) {
        // one pass: workgroups of contiguous 64-node strips, as many as the partial buffer holds (<= 256)
        const int strips = (N + GP_W - 1) / GP_W;
        int nwg = strips < 256 ? strips : 256;
        if ((size_t)nwg * rec > part_floats) nwg = (int)(part_floats / rec);
        const int spw = (strips + nwg - 1) / nwg;
        nwg = (strips + spw - 1) / spw;
        NodeTermsArgs a{x, grad_out, root, dx, part, N, spw, (root && dx) ? 1 : 0, grad_root ? 1 : 0};
        hipLaunchKernelGGL(k_node_terms, dim3(nwg), dim3(256), 0, st, a);
        if (grad_root || grad_bias)
            hipLaunchKernelGGL(k_node_terms_reduce, dim3((unsigned)((rec + 255) / 256)), dim3(256), 0, st, part, nwg, grad_root, grad_bias, acc_root, acc_bias);
        GP_LAUNCH_CHECK("k_node_terms");
        return GPDE_OK;
    }
    if (root && dx) {
        GpdeGemmArgs g = gemm0();
        g.A = grad_out; g.lda = GP_W; g.B = root; g.ldb = GP_W; g.C = dx; g.ldc = GP_W;
        g.M = N; g.N = GP_W; g.K = GP_W; g.accumulate = 1;
        if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
    }
    if (grad_root)
        if ((rc = gemm_tn_acc(x, GP_W, GP_W, grad_out, GP_W, GP_W, N, grad_root, GP_W, part, part_floats, acc_root, st)) != GPDE_OK) return rc;
    if (grad_bias) {
        int splits = 1; while (splits < 64 && N / (splits * 2) >= 64) splits *= 2;
        hipLaunchKernelGGL(k_colsum, dim3(1, splits), dim3(256), 0, st, grad_out, N, GP_W, GP_W, splits, part, (unsigned*)nullptr);
        if ((rc = gpde_launch_reduce_splits(part, GP_W, splits, GP_W, grad_bias, acc_bias, st)) != GPDE_OK) return rc;
    }
    return GPDE_OK;
}

// ---- backward of the operator GIVEN the per-edge weights (gpde_weconv.hip's forward) --------------------------------------
// out_i = aggr_{e -> i} x_src(e) . W_e + x_i . root + bias (nn_conv.py:275-282).  With gT_i = g_i / deg_i ('mean') or g_i:
//   dW_e[c][o] = x_j[c] gT_i[o]          (16 KiB written per edge: what autograd forms for `weight` in nn_conv.py:274-275)
//   dx_j[c]   += sum_o W_e[c][o] gT_i[o]
// One workgroup per destination node, its in-edges dealt round-robin to the waves; lane = (q, o4) as in the forward (rows
// c = 4 cc + q, outputs o4 .. o4 + 3): 16 KiB read + 16 KiB written per edge, streaming.  dx: per-edge rows for the ordered
// reduction over the source's out-edges (k_dx_reduce), or atomics.
struct WeBwdArgs {
    const float* x; const float* we; const int32_t* rowptr; const int32_t* src; const float* g; int aggr;
    float* dwe; float* dxe; float* dx;
};
// ACC: dW_e += x_j (x) gT_i - the second .. last application of a module in one backward pass add to the tensor the first one
// wrote (the additions autograd would perform with one elementwise kernel per application, in the same order: the same bits)
template <bool ACC>
__global__ __launch_bounds__(256) void gpde_weconv_bwd_kernel(WeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x;
    const int q = lane >> 4, l15 = lane & 15, o4 = l15 * 4;
    const int r0 = a.rowptr[i], r1 = a.rowptr[i + 1];
    if (r0 == r1) return;
    f32x4 gt = *(const f32x4*)(a.g + (size_t)i * GP_W + o4);
    if (a.aggr == GPDE_AGGR_MEAN) {
        const float deg = (float)(r1 - r0);
#pragma unroll
        for (int j = 0; j < 4; ++j) gt[j] = gt[j] / deg;           // the same division k_scale_g performs
    }
    for (int e = r0 + wave; e < r1; e += 4) {
        const int j_ = a.src[e];
        const float xa = a.x[(size_t)j_ * GP_W + lane];
        const float* w = a.we + (size_t)e * (GP_W * GP_W);
        float* dw = a.dwe + (size_t)e * (GP_W * GP_W);
        f32x4 v[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) v[cc] = *(const f32x4*)(w + (size_t)(4 * cc + q) * GP_W + o4);
        [[maybe_unused]] f32x4 old[16];
        if constexpr (ACC) {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) old[cc] = *(const f32x4*)(dw + (size_t)(4 * cc + q) * GP_W + o4);
        }
        float mine = 0.f;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const float xv = __int_as_float(__builtin_amdgcn_ds_bpermute((4 * cc + q) * 4, __float_as_int(xa)));
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pr = xv * gt[j];
                if constexpr (ACC) {
                    asm volatile("" : "+v"(pr));      // the rounded product, THEN the sum - as the two kernels it replaces (hipcc contracts
                    pr = old[cc][j] + pr;             // a * b + c into one fma otherwise, and __fmul_rn is a plain `*` here)
                }
                o[j] = pr;
            }
            *(f32x4*)(dw + (size_t)(4 * cc + q) * GP_W + o4) = o;
            float p = fmaf(v[cc][3], gt[3], fmaf(v[cc][2], gt[2], fmaf(v[cc][1], gt[1], v[cc][0] * gt[0])));
            p += __shfl_xor(p, 1); p += __shfl_xor(p, 2); p += __shfl_xor(p, 4); p += __shfl_xor(p, 8);   // the quarter's 16 lanes: all 64 outputs
            if (l15 == cc) mine = p;                               // dx_e[c = 4 cc + q]
        }
        const int c = 4 * l15 + q;
        if (a.dxe) a.dxe[(size_t)e * GP_W + c] = mine;
        else atomicAdd(&a.dx[(size_t)j_ * GP_W + c], mine);
    }
}

}  // namespace

extern "C" size_t gpde_nnconv_bwd_edgeweights_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    if (n_nodes < 0 || n_edges < 0) return 0;
    return al((size_t)(n_edges > 0 ? n_edges : 1) * GP_W * 4) + al((size_t)64 * GP_W * GP_W * 4) + 1024;
}

extern "C" int gpde_nnconv_bwd_edgeweights(const float* x, int64_t n_nodes, const float* edge_weights, int64_t n_edges,
                                           const int32_t* rowptr, const int32_t* src, const int32_t* src_rowptr,
                                           const int32_t* src_slots, const float* root, int aggr, const float* grad_out,
                                           float* grad_x, float* grad_edge_weights, float* grad_root, float* grad_bias, void* ws,
                                           size_t ws_bytes, void* stream_) {
    return gpde_nnconv_bwd_edgeweights_acc(x, n_nodes, edge_weights, n_edges, rowptr, src, src_rowptr, src_slots, root, aggr, grad_out, grad_x,
                                           grad_edge_weights, grad_root, grad_bias, 0, ws, ws_bytes, stream_);
}

extern "C" int gpde_nnconv_bwd_edgeweights_acc(const float* x, int64_t n_nodes, const float* edge_weights, int64_t n_edges,
                                               const int32_t* rowptr, const int32_t* src, const int32_t* src_rowptr,
                                               const int32_t* src_slots, const float* root, int aggr, const float* grad_out,
                                               float* grad_x, float* grad_edge_weights, float* grad_root, float* grad_bias,
                                               int accumulate, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (accumulate & ~(GPDE_ACC_EDGE_WEIGHTS | GPDE_ACC_ROOT | GPDE_ACC_BIAS)) { gpde_set_error("gpde_nnconv_bwd_edgeweights_acc: unknown accumulate bits %d", accumulate); return GPDE_EINVAL; }
    if (n_nodes < 0 || n_edges < 0 || !rowptr || !grad_out || !ws || (n_nodes > 0 && (!x || !grad_x)) ||
        (n_edges > 0 && (!edge_weights || !src || !grad_edge_weights)) || n_edges >= ((int64_t)1 << 31) / GP_W) {
        gpde_set_error("gpde_nnconv_bwd_edgeweights: null/negative argument");
        return GPDE_EINVAL;
    }
    if (aggr != GPDE_AGGR_ADD && aggr != GPDE_AGGR_MEAN) { gpde_set_error("gpde_nnconv_bwd_edgeweights: aggr %d (the gradient of 'max' is composed by the caller)", aggr); return GPDE_EUNSUPPORTED; }
    if (ws_bytes < gpde_nnconv_bwd_edgeweights_workspace_bytes(n_nodes, n_edges)) { gpde_set_error("gpde_nnconv_bwd_edgeweights: workspace too small"); return GPDE_EWORKSPACE; }
    if (n_nodes == 0) return GPDE_OK;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    float* dxe = (float*)w;
    float* part = (float*)(w + al((size_t)(n_edges > 0 ? n_edges : 1) * GP_W * 4));
    const size_t part_floats = (size_t)64 * GP_W * GP_W;
    const bool ordered = src_rowptr && src_slots;
    if (!(ordered && n_edges > 0)) GP_HIP_CHECK(gpde_zero_async(grad_x, (size_t)n_nodes * GP_W * 4, st));      // (ordered: k_dx_reduce initialises it)
    if (n_edges > 0) {
        WeBwdArgs a{x, edge_weights, rowptr, src, grad_out, aggr, grad_edge_weights, ordered ? dxe : nullptr, grad_x};
        if (accumulate & GPDE_ACC_EDGE_WEIGHTS) hipLaunchKernelGGL(gpde_weconv_bwd_kernel<true>, dim3((unsigned)n_nodes), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gpde_weconv_bwd_kernel<false>, dim3((unsigned)n_nodes), dim3(256), 0, st, a);
        if (ordered) hipLaunchKernelGGL(k_dx_reduce, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, st, dxe, src_rowptr, src_slots, (int)n_nodes, 0, (int)n_edges, grad_x, 1, (size_t)0, 1);
        GP_LAUNCH_CHECK("gpde_weconv_bwd_kernel");
    }
    return bwd_node_terms(x, (int)n_nodes, root, grad_out, grad_x, grad_root, grad_bias, part, part_floats, st,
                          (accumulate & GPDE_ACC_ROOT) ? 2 : 0, (accumulate & GPDE_ACC_BIAS) ? 2 : 0);      // 2: += sum of the partials
}

// ---- backward of gpde_edge_weights_fwd: W_e = view(W3 . h_e + b3, 64, 64) -------------------------------------------------
//   dU[e][k]  = (sum_n dW_e[n] W3[n][k]) * (H[e][k] > 0)      the gradient reaching the last hidden layer's pre-activation
//   dW3[n][k] = sum_e dW_e[n] H[e][k]            db3[n] = sum_e dW_e[n]
// dW_e is the SUM over the applications of the module (autograd adds them): the two 4096 x k2 products per edge run once per
// step, on the split-f16 GEMMs where the shapes allow (K2P a multiple of 128), else on the fp32 MFMA GEMM.
namespace {
struct WeBwdPlan { size_t off_w3p, off_w3t, off_img, off_ucol, off_rsc, off_tn, off_part, off_dw3p, off_cbits, off_ntpart, total; int K2P; bool split; int ks, ks_nt; };
WeBwdPlan we_bwd_plan(int64_t E, int k2) {
    WeBwdPlan P{};
    P.K2P = gp_round_up(k2, 128);
    const size_t wn = (size_t)GP_W * GP_W * P.K2P;
    P.split = gpde_gemm_f16s_supported((int)(E > 0 ? E : 1), P.K2P, GP_W * GP_W, GP_W * GP_W);
    P.ks = 4;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    P.off_w3p = take(wn); P.off_w3t = take(wn); P.off_img = take(wn); P.off_ucol = take(P.K2P);
    P.off_rsc = take((size_t)2 * (E > 0 ? E : 1));
    P.off_tn = take(P.split ? gpde_gemm_f16s_tn_ws_floats((int)(E > 0 ? E : 1), GP_W * GP_W, P.K2P, P.ks) : 1);
    P.off_part = take((size_t)(P.split ? P.ks : 16) * wn);
    P.off_dw3p = take(wn);
    P.off_cbits = take((size_t)GP_W * GP_W);      // column maxima of dW_e (bit patterns) from the db3 pass, for the dW3 GEMM's scales
    // dU = dW_e . W3 with few rows and a narrow K2P (the coarse MGKN levels) is a handful of workgroups that each walk K = 4096
    // alone: 130 us whatever E is.  Up to 64 (row quad, column slice) workgroups: 8 K splits, partials summed in order
    P.ks_nt = (P.split && ((E + 255) / 256) * (int64_t)(P.K2P / GP_TN) <= 64) ? 8 : 1;
    P.off_ntpart = take(P.ks_nt > 1 ? (size_t)P.ks_nt * (E > 0 ? E : 1) * P.K2P : 1);
    P.total = off + 512;
    return P;
}
}  // namespace

extern "C" size_t gpde_edge_weights_bwd_workspace_bytes(int64_t n_edges, int n_layers, const int32_t* dims) {
    if (!dims || n_edges < 0 || n_layers < 2 || n_layers > GPDE_MAX_LAYERS) return 0;
    return we_bwd_plan(n_edges, dims[n_layers - 1]).total;
}

extern "C" int gpde_edge_weights_bwd(const float* grad_edge_weights, const float* hidden, int64_t n_edges, int n_layers,
                                     const int32_t* dims, const float* w_last, float* grad_hidden, float* grad_w_last,
                                     float* grad_b_last, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (n_edges < 0 || !dims || n_layers < 2 || n_layers > GPDE_MAX_LAYERS || !w_last || !ws ||
        (n_edges > 0 && (!grad_edge_weights || !hidden || !grad_hidden)) || n_edges >= ((int64_t)1 << 31) / GP_W) {
        gpde_set_error("gpde_edge_weights_bwd: null/negative argument");
        return GPDE_EINVAL;
    }
    const int k2 = dims[n_layers - 1], NW3 = GP_W * GP_W, T = 256;
    const WeBwdPlan P = we_bwd_plan(n_edges, k2);
    if (ws_bytes < P.total) { gpde_set_error("gpde_edge_weights_bwd: workspace %zu < %zu bytes", ws_bytes, P.total); return GPDE_EWORKSPACE; }
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    auto F = [&](size_t off) { return (float*)(w + off); };
    const int K2P = P.K2P, E = (int)n_edges;
    const size_t wn = (size_t)NW3 * K2P;
    int rc;
    if (E == 0) {
        if (grad_w_last) GP_HIP_CHECK(gpde_zero_async(grad_w_last, (size_t)NW3 * k2 * 4, st));
        if (grad_b_last) GP_HIP_CHECK(gpde_zero_async(grad_b_last, (size_t)NW3 * 4, st));
        return GPDE_OK;
    }
    hipLaunchKernelGGL(k_pad_mat, dim3(nblk(wn)), dim3(T), 0, st, w_last, NW3, k2, k2, NW3, K2P, F(P.off_w3p));
    // db3 = column sums of dW_e (ordered split partials) - FIRST: the same pass collects the column maxima the dW3 GEMM scales
    // its transposed operand with (round 6: that GEMM's own k_colabsmax pass read the 16 KiB per edge a second time)
    const unsigned* cbits = nullptr;
    if (grad_b_last) {
        const int cb = (NW3 + 255) / 256;
        int splits = 1; while (splits < 256 && cb * splits < 1024 && E / (splits * 2) >= 16) splits *= 2;
        const bool want_bits = P.split && grad_w_last;
        if (want_bits) GP_HIP_CHECK(gpde_zero_async(F(P.off_cbits), (size_t)NW3 * 4, st));
        hipLaunchKernelGGL(k_colsum, dim3(cb, splits), dim3(T), 0, st, grad_edge_weights, E, NW3, NW3, splits, F(P.off_part),
                           want_bits ? (unsigned*)F(P.off_cbits) : (unsigned*)nullptr);
        if ((rc = gpde_launch_reduce_splits(F(P.off_part), NW3, splits, NW3, grad_b_last, 0, st)) != GPDE_OK) return rc;
        if (want_bits) cbits = (const unsigned*)F(P.off_cbits);
    }
    if (P.split) {
        // dU = (dW_e . W3) (.) [H > 0]: A = dW_e rows [E][4096], B[n = k][K = (c, o)] = W3[(c, o)][k] = W3^T as a split image
        hipLaunchKernelGGL(k_transpose, dim3(nblk(wn)), dim3(T), 0, st, F(P.off_w3p), NW3, K2P, F(P.off_w3t));
        if ((rc = gpde_pack_split_nk(F(P.off_w3t), K2P, NW3, K2P, NW3, F(P.off_img), F(P.off_ucol), st)) != GPDE_OK) return rc;
        GpdeGemmF16sArgs g{};
        g.A = grad_edge_weights; g.lda = NW3; g.M = E; g.bsplit = F(P.off_img); g.ucol = F(P.off_ucol);
        g.mask = hidden; g.ldmask = K2P; g.C = grad_hidden; g.ldc = K2P; g.K = NW3; g.N = K2P; g.ksplits = 1;
        if (P.ks_nt > 1) {          // (the ReLU mask multiplies every partial: the sum of the masked partials is the masked sum)
            g.C = F(P.off_ntpart); g.ksplits = P.ks_nt; g.cstride = (size_t)E * K2P;
            if ((rc = gpde_launch_gemm_f16s_nt(g, F(P.off_rsc), st)) != GPDE_OK) return rc;
            if ((rc = gpde_launch_reduce_splits(F(P.off_ntpart), (size_t)E * K2P, P.ks_nt, (size_t)E * K2P, grad_hidden, 0, st)) != GPDE_OK) return rc;
        } else if ((rc = gpde_launch_gemm_f16s_nt(g, F(P.off_rsc), st)) != GPDE_OK) return rc;
        // dW3 = dW_e^T . H (contraction over the edges)
        if (grad_w_last) {
            if ((rc = gpde_launch_gemm_f16s_tn(grad_edge_weights, NW3, NW3, hidden, K2P, K2P, E, P.ks, F(P.off_tn), F(P.off_part), st, cbits)) != GPDE_OK) return rc;
            if ((rc = gpde_launch_reduce_splits(F(P.off_part), wn, P.ks, wn, F(P.off_dw3p), 0, st)) != GPDE_OK) return rc;
        }
    } else {
        GpdeGemmArgs g = gemm0();
        g.A = grad_edge_weights; g.lda = NW3; g.B = F(P.off_w3p); g.ldb = K2P; g.b_kcontig = 0;
        g.C = grad_hidden; g.ldc = K2P; g.M = E; g.N = K2P; g.K = NW3; g.mask = hidden; g.ldmask = K2P;
        if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
        if (grad_w_last)
            if ((rc = gemm_tn_acc(grad_edge_weights, NW3, NW3, hidden, K2P, K2P, E, F(P.off_dw3p), K2P, F(P.off_part), (size_t)16 * wn, 0, st)) != GPDE_OK) return rc;
    }
    if (grad_w_last)
        hipLaunchKernelGGL(k_unpad_mat, dim3(nblk((size_t)NW3 * k2)), dim3(T), 0, st, F(P.off_dw3p), NW3, k2, K2P, grad_w_last);
    GP_LAUNCH_CHECK("gpde_edge_weights_bwd kernels");
    return GPDE_OK;
}

extern "C" size_t gpde_nnconv_bwd_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers,
                                                  const int32_t* dims) {
    BwdPlan P;
    if (!dims || n_nodes < 0 || n_edges < 0) return 0;
    if (make_bwd_plan(n_nodes, n_edges, n_layers, dims, 0, true, &P) != GPDE_OK) return 0;
    return P.total;
}

namespace {

// One implementation, three entry points (the cross-depth reuse of SURVEY.md §8 row f4 splits the
// operator into hidden(edge_attr) -> H and conv(x, H)):
//   BWD_FULL  gpde_nnconv_bwd         hidden chain recomputed per chunk, everything differentiated
//   BWD_CONV  gpde_nnconv_bwd(hidden) H given ([CSR slot][K2P]); grads of x, W3, b3, root, bias and
//                                     dL/dU of the last hidden layer ([CSR slot][K2P]) written out
//   BWD_MLP   gpde_hidden_bwd         H and dL/dU given; grads of the hidden Linear layers
//   BWD_LIGHT gpde_nnconv_bwd_light     one application of a depth-shared module: everything BUT the hidden layers' gradients
//                                       (grad_x, last Linear, root, bias); the last hidden layer is recomputed, never differentiated
//   BWD_DEFER gpde_nnconv_bwd_deferred  the hidden layers' gradients of ALL those applications in one pass over the edges:
//                                       dU_2 = (sum_l x_j^(l) . dZ_i^(l)) (.) [H_2 > 0]  (a K = 64 L contraction), then ONE MLP backward
enum BwdPhase { BWD_FULL = 0, BWD_CONV = 1, BWD_MLP = 2, BWD_LIGHT = 3, BWD_DEFER = 4 };

int bwd_impl(BwdPhase phase, const float* x, int64_t n_nodes, const float* edge_attr, int64_t n_edges,
             const int32_t* rowptr, const int32_t* src, const int32_t* dst, const int32_t* perm,
             const int32_t* rowptr_host, int n_layers, const int32_t* dims, const float* const* W,
             const float* const* b, const float* root, int aggr, const float* grad_out, float* grad_x,
             float* const* grad_W, float* const* grad_b, float* grad_root, float* grad_bias,
             const float* hidden, float* grad_hidden_out, const float* grad_hidden_in, void* ws,
             size_t ws_bytes, hipStream_t st, const int32_t* src_rowptr = nullptr, const int32_t* src_slots = nullptr,
             const float* z_saved = nullptr, int n_defer = 0, const float* x_stack = nullptr, const float* g_stack = nullptr,
             const float* hpart = nullptr, int64_t hpart_nodes = 0, int kt = 0, const int32_t* sel = nullptr,
             float* grad_attr = nullptr, bool gh_accumulate = false) {
    // gh_accumulate (BWD_CONV): dL/dU of the last hidden layer is ADDED to grad_hidden_out (the applications of a module sharing its
    // hidden activations sum it there, in call order - the same additions autograd would make with six [E][K2P] tensors)
    // grad_attr (BWD_FULL, tensor attributes of <= 8 slots): [E][k0] in the CALLER's edge order, dL/d edge_attr
    // kt > 0: `edge_attr` is a NODE table [n_nodes][kt] and slot d of an edge's attribute is table[(sel[d] >> 8 ? dst : src)][sel[d] & 255]
    // (row f3: GpdeNodeAttr); perm is unused
    // hpart (BWD_LIGHT / BWD_DEFER): the last hidden activations of the in-edges of nodes [0, hpart_nodes) are GIVEN (a partial
    // H kept by the caller, CSR slots [0, rowptr[hpart_nodes])): node chunks below that bound read them instead of recomputing
    const bool do_conv = phase != BWD_MLP && phase != BWD_DEFER, do_mlp = phase != BWD_CONV && phase != BWD_LIGHT;
    const GpdeSwitches& SW = gpde_switches();       // developer / A-B switches, read once per process (gpde_common.h)
    BwdPlan P;
    const bool h_all = phase == BWD_FULL && hpart && hpart_nodes >= n_nodes;     // H of every edge kept by the forward
    int rc = make_bwd_plan(n_nodes, n_edges, n_layers, dims, ws_bytes, false, &P, phase == BWD_DEFER ? n_defer : 0, h_all);
    if (rc != GPDE_OK) return rc;
    const int n = n_layers, K2P = P.K2P;
    const int N = (int)n_nodes;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    auto F = [&](size_t off) { return (float*)(w + off); };
    const int T = 256;

    // ---- padded weights, zeroed gradient accumulators --------------------------------------------------
    for (int l = 1; l < n && do_mlp; ++l) {
        const size_t wn = (size_t)P.KP[l] * P.KP[l - 1];
        hipLaunchKernelGGL(k_pad_mat, dim3(nblk(wn)), dim3(T), 0, st, W[l - 1], dims[l], dims[l - 1], dims[l - 1],
                           P.KP[l], P.KP[l - 1], F(P.off_wp[l]));
        if (b[l - 1]) hipLaunchKernelGGL(k_pad_mat, dim3(nblk(P.KP[l])), dim3(T), 0, st, b[l - 1], 1, dims[l], dims[l], 1, P.KP[l], F(P.off_bp[l]));
        else GP_HIP_CHECK(gpde_zero_async(F(P.off_bp[l]), (size_t)P.KP[l] * 4, st));
        GP_HIP_CHECK(gpde_zero_async(F(P.off_dwp[l]), wn * 4, st));
        GP_HIP_CHECK(gpde_zero_async(F(P.off_dbp[l]), (size_t)P.KP[l] * 4, st));
    }
    const bool f16s_du1 = do_mlp && P.f16s_du1 && !SW.bwd_gemm_f32;
    const bool f16s_dw2 = f16s_du1 && P.f16s_dw2 && !SW.bwd_dw2_f32;
    if (f16s_du1) {
        // B operand of dU_1 = dU_2 . W_2: rows = k1 (output), contraction = k2  ->  W_2^T, split + swizzled like the forward's W2
        const size_t wn = (size_t)P.KP[2] * P.KP[1];
        hipLaunchKernelGGL(k_transpose, dim3(nblk(wn)), dim3(T), 0, st, F(P.off_wp[2]), P.KP[2], P.KP[1], F(P.off_w2t));
        if ((rc = gpde_pack_split_nk(F(P.off_w2t), P.KP[1], P.KP[2], P.KP[1], P.KP[2], F(P.off_w2ts), F(P.off_ucol2), st)) != GPDE_OK) return rc;
    }
    const size_t w3n = (size_t)GP_W * GP_W * K2P;
    if (phase == BWD_DEFER)
        hipLaunchKernelGGL(k_pad_mat, dim3(nblk(w3n)), dim3(T), 0, st, W[n - 1], GP_W * GP_W, dims[n - 1], dims[n - 1],
                           GP_W * GP_W, K2P, F(P.off_w3p));
    if (do_conv) {
        hipLaunchKernelGGL(k_pad_mat, dim3(nblk(w3n)), dim3(T), 0, st, W[n - 1], GP_W * GP_W, dims[n - 1], dims[n - 1],
                           GP_W * GP_W, K2P, F(P.off_w3p));
        if (b[n - 1]) GP_HIP_CHECK(gpde_copy_async(F(P.off_b3), b[n - 1], GP_W * GP_W * 4, st));
        else GP_HIP_CHECK(gpde_zero_async(F(P.off_b3), GP_W * GP_W * 4, st));
        GP_HIP_CHECK(gpde_zero_async(F(P.off_dw3p), (P.off_db3 - P.off_dw3p) + (size_t)GP_W * GP_W * 4, st));      // dw3p and db3
        if (grad_x) GP_HIP_CHECK(gpde_zero_async(grad_x, (size_t)N * GP_W * 4, st));
    }
    float* dx = grad_x;

    // FULL phase, 3-Linear kernels: the expensive last hidden layer of the recompute runs on the forward's
    // fused f16-split kernel with the store epilogue (3x the fp32 GEMM's rate); the first hidden layer is
    // still needed in memory (dW_2 = dU_2^T H_1) and comes from the GEMM below
    GpdePackLayout PL;
    bool fast_last = false;
    const bool recomputes = phase == BWD_FULL || phase == BWD_LIGHT || phase == BWD_DEFER;
    if (recomputes && P.pack_bytes && rowptr && (phase != BWD_FULL || !SW.bwd_recompute_f32) &&
        gpde_pack_layout(n, dims, &PL) == GPDE_OK && PL.mode == 1) {
        GpdeFusedArgs probe{};
        probe.k0 = PL.k0; probe.K1P = PL.K1P; probe.K2P = PL.K2P;
        if (gpde_fused_store_supported(probe)) {
            if ((rc = gpde_mlp_pack(n, dims, W, b, F(P.off_pack), P.pack_bytes, st)) != GPDE_OK) return rc;
            fast_last = true;
        }
    }
    // 3-Linear kernels on the split-f16 GEMMs: the first hidden layer H_1 is never written.  Its only consumers in the
    // backward are dW_2 = dU_2^T . H_1 (operand image generated straight from the 8 attribute slots, k_first_layer_pack)
    // and the ReLU mask of dU_1 (128 bytes of bits per edge instead of 4 KiB).  GPDE_BWD_H1_MATERIALIZE=1: the tensor (A/B).
    const bool h1_on_the_fly = n == 3 && f16s_du1 && f16s_dw2 && dims[0] <= 8 && P.KP[0] >= 8 && !SW.bwd_h1_materialize &&
                               !SW.bwd_h1_gemm;
    auto skip_h1 = [&](int rows) { return h1_on_the_fly && rows >= 8192; };
    const bool call_amax = do_mlp && h1_on_the_fly && n_edges > 0 && dims[0] <= 7 && (!kt || (src && dst));
    if (call_amax) {
        // one bound per attribute slot for the whole call (k_attr_absmax_all): 24 bytes per edge read once - 0.1 ms at s=121
        GP_HIP_CHECK(gpde_zero_async(F(P.off_amax8), 16 * 4, st));
        NodeAttrSel ns_{};
        ns_.kt = kt;
        for (int d_ = 0; d_ < 8; ++d_) ns_.sel[d_] = (kt && sel) ? sel[d_ < dims[0] ? d_ : dims[0] - 1] : 0;
        int nb_ = (int)((n_edges + 255) / 256); if (nb_ > 1024) nb_ = 1024;
        hipLaunchKernelGGL(k_attr_absmax_all, dim3(nb_), dim3(256), 0, st, edge_attr, n_edges, kt ? kt : dims[0], dims[0], ns_, src, dst,
                           (unsigned*)F(P.off_amax8));
    }
    if ((phase == BWD_LIGHT && !fast_last) || (phase == BWD_DEFER && !(fast_last && f16s_du1 && f16s_dw2 && n == 3))) {
        gpde_set_error("gpde_nnconv_bwd_%s: kernel MLP outside the depth-deferred form (3 Linear layers of widths that are multiples of 128, "
                       "k0 <= 7): use gpde_nnconv_bwd", phase == BWD_LIGHT ? "light" : "deferred");
        return GPDE_EUNSUPPORTED;
    }
    const bool light = phase == BWD_LIGHT;
    // recompute of the hidden chain for rows [e0, e0 + rows) = in-edges of nodes [na_, nb_): layers 1 .. last
    int rc_na = 0, rc_nb = 0;
    NodeAttrSel nas{};
    nas.kt = kt;
    for (int d_ = 0; d_ < 8; ++d_) nas.sel[d_] = (kt && sel) ? sel[d_ < dims[0] ? d_ : dims[0] - 1] : 0;
    if (kt && ((recomputes && !fast_last) || !src || !dst)) {
        gpde_set_error("gpde_nnconv_bwd: node-table attributes need the 3-Linear split-f16 form (and src / dst)");
        return GPDE_EUNSUPPORTED;
    }
    // ---- the one-pass kernel (round 5, gpde_fused_f16v6_kernel<2>): K loop of the recompute + both per-edge products, H_2 neither
    // written nor read.  Needs: the fused store kernel's shape, Z kept by the forward (dW_3 wants Z, and Z from H_2 is what this
    // path no longer has), the source-ordered slots (its dx comes out as per-slice partial rows for k_dx_reduce), and - in the
    // full backward - the split GEMMs that take the by-products.  OPT-IN (GPDE_BWD_ONE_PASS=1 or GPDE_EDGE_BWD=4): measured at
    // s=121 the kernel takes 52.6 ms where recompute-store + gpde_edge_bwd3_kernel take 32.1 + 17.5 - a one-wave-per-SIMD kernel
    // cannot hide the products' conversions, cross-lane column statistics and 12 KiB per edge of stores under anything
    // (ablations: profiles/r05_onepass_ablation.txt).
    bool onepass_ok = false;
    if ((phase == BWD_FULL || light) && fast_last && n == 3 && (SW.bwd_one_pass || SW.edge_bwd == 4) && (SW.edge_bwd == 0 || SW.edge_bwd == 4) && z_saved &&
        src_rowptr && src_slots && src && dst && K2P % GP_TN == 0 && K2P / GP_TN <= GP_W && n_edges > 0) {
        GpdeFusedArgs probe{};
        probe.k0 = PL.k0; probe.K1P = PL.K1P; probe.K2P = PL.K2P; probe.xs = (const unsigned*)x;
        onepass_ok = gpde_fused_f16v6_supported(probe);
        if (onepass_ok && (rc = gpde_launch_attr_bound(edge_attr, n_edges, PL.k0, F(P.off_pack) + PL.off_w1 + (size_t)PL.K1P * 8,
                                                       (unsigned*)F(P.off_scal), st, kt, kt ? nas.sel : nullptr, src, dst)) != GPDE_OK) return rc;
    }
    // Z re-aggregation (no Z kept by the forward: G241, the light passes) on the split-f16 aggregation kernel the forward's
    // given-H path uses (gpde_zagg_kernel<true>, ~3x the fp32-MFMA form's rate): x as split pairs once per call, and ONE bound for
    // every H value - the forward's a-priori bound max|b2| + max_k ||W2_k||_1 . max_e B_e, which holds for recomputed and for
    // given (partial-H) rows alike since both come from this kernel MLP.  GPDE_BWD_ZAGG_F32=1: the fp32 kernel (A/B).
    bool zagg16 = false;
    if ((phase == BWD_FULL || light) && fast_last && !z_saved && !SW.bwd_zagg_f32 && n_edges >= 32768 && N > 0 && x) {
        if ((rc = gpde_launch_g2_prep(x, N, edge_attr, n_edges, PL.k0, F(P.off_pack) + PL.off_w1 + (size_t)PL.K1P * 8, (unsigned*)F(P.off_scal),
                                      (unsigned*)F(P.off_xs), st, kt, kt ? nas.sel : nullptr, src, dst)) != GPDE_OK) return rc;
        hipLaunchKernelGGL(k_h_bound_word, dim3(1), dim3(1), 0, st, F(P.off_pack) + PL.off_fcol, (const unsigned*)F(P.off_scal), (unsigned*)F(P.off_scal) + 2);
        zagg16 = true;
    }
    const float* chunk_h = nullptr;          // the current chunk's last hidden activations when they are given (hpart)
    bool skip_store = false;                 // the chunk runs the one-pass kernel: the last hidden layer is not written
    auto recompute = [&](int e0, int rows, int last) -> int {
        if (!light) {    // (the light pass needs the last hidden layer only, which the fused kernel forms from the attributes itself)
            if (kt) hipLaunchKernelGGL(k_gather_attr_nodes, dim3(nblk((size_t)rows * P.KP[0])), dim3(T), 0, st, edge_attr, nas, src, dst, e0,
                                       rows, dims[0], P.KP[0], F(P.off_H[0]));
            else hipLaunchKernelGGL(k_gather_attr, dim3(nblk((size_t)rows * P.KP[0])), dim3(T), 0, st, edge_attr, perm, e0,
                                    rows, dims[0], P.KP[0], F(P.off_H[0]));
        }
        if (last == n - 1 && chunk_h) last = n - 2;          // given (partial H of the caller / H kept by the forward): read, not recomputed
        else if (fast_last && last == n - 1 && skip_store) last = n - 2;
        else if (fast_last && last == n - 1) {
            const float* pk = F(P.off_pack);
            GpdeFusedArgs f{};
            f.attr = edge_attr; f.rowptr = rowptr; f.perm = perm;
            f.src = src; f.dst = dst; f.kt = kt;
            for (int d_ = 0; d_ < 8; ++d_) f.sel[d_] = nas.sel[d_];
            f.w1 = pk + PL.off_w1; f.w2t = pk + PL.off_w2t; f.b2 = pk + PL.off_b2;
            f.w2h = pk + PL.off_w2h; f.ucol = pk + PL.off_ucol; f.w1h = pk + PL.off_w1h; f.fcol = pk + PL.off_fcol;
            f.hout = F(P.off_H[n - 1]); f.k0 = PL.k0; f.K1P = PL.K1P; f.K2P = PL.K2P;
            f.nc0 = rc_na; f.nc1 = rc_nb; f.e_chunk0 = e0;
            if (kt && !gpde_fused_f16v6_supported(f)) { gpde_set_error("node-table attributes: kernel MLP outside the one-wave-per-SIMD store kernel"); return GPDE_EUNSUPPORTED; }
            const int ns = PL.K2P / GP_TN;
            int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
            const int gcap = (rows / GP_TE + GP_WAVES) / GP_WAVES; if (groups > gcap) groups = gcap;
            f.n_groups = groups;
            int rc2 = gpde_launch_fused_store(f, st);
            if (rc2 != GPDE_OK) return rc2;
            last = n - 2;
        }
        if (light) return GPDE_OK;
        for (int l = 1; l <= last; ++l) {
            if (l == 1 && last == 1 && skip_h1(rows)) continue;
            if (l == 1 && dims[0] <= 8 && P.KP[0] >= 8 && P.KP[1] % 4 == 0 && rows >= 1024 && !SW.bwd_h1_gemm) {
                const int cb = (P.KP[1] + 255) / 256;
                int rb = rows / 64; if (rb > 4096 / cb) rb = 4096 / cb; if (rb < 1) rb = 1;
                hipLaunchKernelGGL(k_first_layer, dim3(cb, rb), dim3(T), 0, st, F(P.off_H[0]), P.KP[0], F(P.off_wp[1]), P.KP[0],
                                   F(P.off_bp[1]), rows, P.KP[1], F(P.off_H[1]));
                continue;
            }
            GpdeGemmArgs g = gemm0();
            g.A = F(P.off_H[l - 1]); g.lda = P.KP[l - 1]; g.B = F(P.off_wp[l]); g.ldb = P.KP[l - 1];
            g.C = F(P.off_H[l]); g.ldc = P.KP[l]; g.M = rows; g.N = P.KP[l]; g.K = P.KP[l - 1];
            g.bias = F(P.off_bp[l]); g.relu = 1;
            int rc2 = gpde_launch_gemm(g, st);
            if (rc2 != GPDE_OK) return rc2;
        }
        return GPDE_OK;
    };
    // MLP backward over rows [e0, e0 + rows): dU_last given (read only), activations H[0 .. n-2] in the
    // workspace, H[n-1] = Hlast
    if (grad_attr && (phase != BWD_FULL || kt || dims[0] > 8 || P.KP[0] < 8 || !perm)) {
        gpde_set_error("gpde_nnconv_bwd: the edge-attribute gradient is built for the full backward on an attribute tensor of <= 8 slots");
        return GPDE_EUNSUPPORTED;
    }
    int mlp_e0 = 0;                          // first CSR slot of the chunk mlp_backward is working on
    bool du_pre = false;                     // the per-edge kernel of this chunk wrote dU_2^T, its row scales and tile column partials
    auto mlp_backward = [&](const float* dUlast, int rows) -> int {
        const float* dUc = dUlast;
        float* bufs[2] = {F(P.off_dU[0]), F(P.off_dU[1])};
        // The layer GEMMs below read every row of dU_l from all their column-slice workgroups and write dU_{l-1}
        // rows of the same indices: the output must never be the buffer the input lives in.  In the FULL phase
        // dU_last IS bufs[0] (the per-edge kernel wrote it there), so the first output goes to bufs[1].  (Rounds 1-2
        // started at bufs[0]: a workgroup that finished its tile early overwrote rows a sibling slice was still
        // reading - the intermittent 1e-3 error in grad_W1)
        int nb_ = dUlast == bufs[0] ? 1 : 0;
        bool fl_in_kernel = false;           // the dW_2 GEMM of this chunk generated H_1 itself (round 6 plan)
        bool dw1_done = false;               // the dU_1 GEMM's epilogue formed dW_1 / db_1 (k_dw_first has nothing left to do)
        for (int l = n - 1; l >= 1; --l) {
            const int Kl = P.KP[l], Kin = P.KP[l - 1];
            int rc2;
            if (l == 1 && grad_attr)         // dU_1 is complete here: the gradient of the attributes through W_1
                hipLaunchKernelGGL(k_grad_attr, dim3((rows + 3) / 4), dim3(T), 0, st, dUc, Kl, F(P.off_wp[1]), Kin, perm, mlp_e0, rows,
                                   dims[0], grad_attr);
            if (l == 1 && dw1_done) continue;
            if (l == 1 && dims[0] <= 8 && Kin >= 8 && Kin % 4 == 0 && Kl % 4 == 0 && rows >= 1024 && !SW.bwd_dw1_gemm) {
                // dW_1 and db_1 from one pass over dU_1 (k_dw_first; attribute slots beyond k0 are zero columns of H_0)
                const int cb = (Kl + 255) / 256;
                int splits = 1; while (splits < 256 && cb * splits < 1024 && rows / (splits * 2) >= 64) splits *= 2;
                while (splits > 1 && (size_t)splits * Kl * 9 > P.part_floats) splits /= 2;
                if ((size_t)splits * Kl * 9 <= P.part_floats) {
                    hipLaunchKernelGGL(k_dw_first, dim3(cb, splits), dim3(T), 0, st, dUc, F(P.off_H[0]), Kin, rows, Kl, splits, F(P.off_part));
                    hipLaunchKernelGGL(k_dw_first_reduce, dim3(nblk((size_t)Kl * 9)), dim3(T), 0, st, F(P.off_part), splits, Kl, Kin,
                                       F(P.off_dwp[l]), F(P.off_dbp[l]));
                    continue;      // l == 1 is the last layer of the loop: nothing below it to back-propagate into
                }
            }
            const bool tn_split = l == 2 && f16s_dw2 && rows >= 8192;
            // one pass over dU_2 for its transposed copy, db_2, and the row scales of the dU_1 GEMM (GPDE_BWD_DU_PASSES=1:
            // the separate k_colsum / k_row_scale_kernel passes of round 2, A/B)
            const bool du_one_pass = tn_split && f16s_du1 && !SW.bwd_du_passes;
            unsigned* du_bits = tn_split ? (unsigned*)F(P.off_dubits) : nullptr;
            if (!du_one_pass) {   // db_l = column sums of dU_l; the same pass collects the column maxima the split dW_2 GEMM scales with
                const int cb = (Kl + 255) / 256;
                int splits = 1; while (splits < 512 && cb * splits < 2048 && rows / (splits * 2) >= 64) splits *= 2;
                while (splits > 1 && (size_t)splits * Kl > P.part_floats) splits /= 2;
                if (du_bits) GP_HIP_CHECK(gpde_zero_async(du_bits, (size_t)Kl * 4, st));
                hipLaunchKernelGGL(k_colsum, dim3(cb, splits), dim3(T), 0, st, dUc, rows, Kl, Kl, splits, F(P.off_part), du_bits);
                if ((rc2 = gpde_launch_reduce_splits(F(P.off_part), Kl, splits, Kl, F(P.off_dbp[l]), 1, st)) != GPDE_OK) return rc2;
            }
            if (tn_split) {
                // dW_2 += dU_2^T . H_1 on the split-f16 GEMM (contraction over the edges: both operands transposed)
                GpdeFirstLayerSpec fl{F(P.off_H[0]), P.KP[0], F(P.off_wp[1]), P.KP[0], F(P.off_bp[1]), (uint32_t*)F(P.off_maskbits), dims[0],
                                      call_amax ? (const unsigned*)F(P.off_amax8) : nullptr};
                fl_in_kernel = skip_h1(rows) && gpde_first_layer_in_kernel(fl, rows, tn_ksplits(rows, P.KP[2], P.KP[1]));
                GpdeDuStats dst_{F(P.off_dbp[l]), F(P.off_rowsc), F(P.off_rowsc) + rows,
                                 du_pre ? F(P.off_tcs) : nullptr, du_pre ? (const unsigned*)F(P.off_tcm) : nullptr};
                if ((rc2 = gpde_launch_gemm_f16s_tn(dUc, Kl, Kl, F(P.off_H[l - 1]), Kin, Kin, rows, tn_ksplits(rows, P.KP[2], P.KP[1]),
                                                    F(P.off_tnws), F(P.off_part), st, du_one_pass ? nullptr : du_bits,
                                                    skip_h1(rows) ? &fl : nullptr, du_one_pass ? &dst_ : nullptr)) != GPDE_OK) return rc2;
                if ((rc2 = gpde_launch_reduce_splits(F(P.off_part), (size_t)Kl * Kin, tn_ksplits(rows, P.KP[2], P.KP[1]), (size_t)Kl * Kin,
                                                     F(P.off_dwp[l]), 1, st)) != GPDE_OK) return rc2;
            } else if ((rc2 = gemm_tn_acc(dUc, Kl, Kl, F(P.off_H[l - 1]), Kin, Kin, rows, F(P.off_dwp[l]), Kin,
                                          F(P.off_part), P.part_floats, 1, st)) != GPDE_OK) return rc2;
            if (l > 1 && l == 2 && f16s_du1 && rows >= 64) {
                float* dUo = bufs[nb_]; nb_ ^= 1;
                GpdeGemmF16sArgs g{};
                g.A = dUc; g.lda = Kl; g.M = rows; g.bsplit = F(P.off_w2ts); g.ucol = F(P.off_ucol2);
                g.mask = F(P.off_H[l - 1]); g.ldmask = Kin; g.C = dUo; g.ldc = Kin; g.K = Kl; g.N = Kin;
                if (tn_split && skip_h1(rows)) { g.mask = nullptr; g.ldmask = 0; g.maskbits = (const uint32_t*)F(P.off_maskbits); g.ldmb = Kin / 32; }
                if (tn_split && fl_in_kernel && !SW.bwd_dw1_pass && dims[0] <= 7 && !grad_attr &&
                    gpde_gemm_f16s_dw_part_floats(rows, Kin) <= (size_t)rows * Kin) {
                    // round 6: this GEMM's epilogue forms dW_1 / db_1 from the tile in its registers: dU_1 (4 KiB per edge) is neither
                    // written nor read back by k_dw_first - unless the attribute gradient wants the tensor (GPDE_BWD_DW1_PASS=1: the
                    // separate pass, A/B).  Scratch of the tile partials: the dU_1 buffer itself (512 bytes per row of its 4 KiB)
                    g.fl_mode = 2; g.fl_attr = F(P.off_H[0]); g.fl_ld0 = P.KP[0]; g.fl_rows = rows;
                    g.fl_dw_part = dUo; g.fl_dw_out = F(P.off_dwp[1]); g.fl_dw_ld = P.KP[0]; g.fl_db_out = F(P.off_dbp[1]);
                    g.fl_skip_store = 1;
                    dw1_done = true;
                }
                // (row scales: a pass over dU_2, 3.9 ms at s=121.  Collecting the row maxima inside gpde_edge_bwd2_kernel was
                // tried in round 3: 16 more registers spill 15 VGPRs of a kernel that sits at its 256-register limit, +5 ms.)
                if (du_one_pass) {                    // row scales left in off_rowsc by the transposing pass above
                    g.sc = F(P.off_rowsc); g.isc = F(P.off_rowsc) + rows;
                    if ((rc2 = gpde_launch_gemm_f16s_nt(g, nullptr, st)) != GPDE_OK) return rc2;
                } else if ((rc2 = gpde_launch_gemm_f16s_nt(g, F(P.off_rowsc), st)) != GPDE_OK) return rc2;
                dUc = dUo;
            } else if (l > 1) {
                float* dUo = bufs[nb_]; nb_ ^= 1;
                GpdeGemmArgs g = gemm0();
                g.A = dUc; g.lda = Kl; g.B = F(P.off_wp[l]); g.ldb = Kin; g.b_kcontig = 0;
                g.C = dUo; g.ldc = Kin; g.M = rows; g.N = Kin; g.K = Kl;
                g.mask = F(P.off_H[l - 1]); g.ldmask = Kin;
                if ((rc2 = gpde_launch_gemm(g, st)) != GPDE_OK) return rc2;
                dUc = dUo;
            }
        }
        return GPDE_OK;
    };

    if (phase == BWD_MLP) {
        // plain edge chunks: nothing here depends on the destination structure
        for (int64_t e0 = 0; e0 < n_edges; e0 += P.Ec) {
            const int rows = (int)((n_edges - e0) < P.Ec ? (n_edges - e0) : P.Ec);
            if ((rc = recompute((int)e0, rows, n - 2)) != GPDE_OK) return rc;
            // activations below the last hidden layer are recomputed; the last one is needed only as
            // the ReLU mask, which the incoming dL/dU already carries
            if ((rc = mlp_backward(grad_hidden_in + (size_t)e0 * K2P, rows)) != GPDE_OK) return rc;
        }
        GP_LAUNCH_CHECK("gpde_hidden_bwd kernels");
    }

    // ---- node-aligned chunks ----------------------------------------------------------------------------------
    if (phase == BWD_DEFER && n_edges > 0)
        if ((rc = gpde_launch_xstack_scales(x_stack, (size_t)N * GP_W, P.L, N, F(P.off_xsc), F(P.off_xsc) + N, st)) != GPDE_OK) return rc;
    int na = 0;
    while ((do_conv || phase == BWD_DEFER) && na < N && n_edges > 0) {
        // largest nb with (nb - na) <= Nc and edges <= Ec (at least one node)
        int lo = na + 1, hi = (int)((int64_t)na + P.Nc < N ? na + P.Nc : N);
        const bool from_h = hpart && na < hpart_nodes;
        if (from_h && hi > hpart_nodes) hi = (int)hpart_nodes;        // a chunk never straddles the end of the given H
        const int64_t ebase = rowptr_host[na];
        if (phase == BWD_CONV) P.Ec = n_edges;       // H and dU live outside the workspace: no edge limit
        if (rowptr_host[lo] - ebase > P.Ec) {
            gpde_set_error("gpde_nnconv_bwd: node %d has in-degree %d > %lld edges per chunk; give more workspace",
                           na, (int)(rowptr_host[lo] - ebase), (long long)P.Ec);
            return GPDE_EWORKSPACE;
        }
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (rowptr_host[mid] - ebase <= P.Ec) lo = mid; else hi = mid - 1;
        }
        const int nb = lo, nn = nb - na;
        const int e0 = (int)ebase, e1 = rowptr_host[nb], rows = e1 - e0;
        float* gT = F(P.off_gT); float* S = F(P.off_S); float* dS = F(P.off_dS);
        float* Z = F(P.off_Z); float* dZ = F(P.off_dZ);
        chunk_h = from_h ? hpart + (size_t)e0 * K2P : nullptr;
        if (phase == BWD_DEFER) {
            if (rows > 0) {
                rc_na = na; rc_nb = nb;
                if ((rc = recompute(e0, rows, n - 1)) != GPDE_OK) return rc;
                const float* H2 = chunk_h ? chunk_h : F(P.off_H[n - 1]);
                // dZ^(l)[i][c][k] = sum_o gT^(l)[i][o] W3[c*64+o][k] of every deferred layer, then the nodes' split images
                const size_t dzl = (size_t)P.Nc * GP_W * K2P;
                for (int l = 0; l < P.L; ++l) {
                    hipLaunchKernelGGL(k_scale_g, dim3((nn + 3) / 4), dim3(T), 0, st, g_stack + (size_t)l * N * GP_W, rowptr, aggr, na, nn, gT);
                    GpdeGemmArgs g = gemm0();
                    g.A = gT; g.lda = GP_W; g.B = F(P.off_w3p); g.ldb = K2P; g.b_kcontig = 0;
                    g.C = F(P.off_dzstack) + (size_t)l * dzl; g.ldc = GP_W * K2P; g.M = nn; g.N = K2P; g.K = GP_W;
                    g.batches = GP_W; g.strideA = 0; g.strideB = (size_t)GP_W * K2P; g.strideC = K2P;
                    if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
                }
                if ((rc = gpde_launch_dz_image(F(P.off_dzstack), dzl, P.L, P.Lp, nn, K2P, (unsigned*)F(P.off_nbits), F(P.off_nscale),
                                               F(P.off_nscale) + P.Nc, F(P.off_dzimg), st)) != GPDE_OK) return rc;
                const int ntiles = gpde_tile_count(rowptr_host, na, nb);
                int32_t* tiles = (int32_t*)F(P.off_tiles);
                if ((rc = gpde_launch_tile_list(rowptr, na, nn, e0, tiles, st)) != GPDE_OK) return rc;
                // dU_2 of the chunk's edges: (sum over the layers of x_j . dZ_i) masked by the recomputed H_2 > 0
                float* dU2 = F(P.off_dU[0]);
                if ((rc = gpde_launch_gemm_f16s_gather(x_stack, (size_t)N * GP_W, P.Lp, F(P.off_xsc), F(P.off_xsc) + N, src + e0, rows, tiles, ntiles,
                                                       F(P.off_dzimg), F(P.off_nscale) + P.Nc, H2, K2P, dU2, K2P, K2P, st)) != GPDE_OK) return rc;
                if ((rc = mlp_backward(dU2, rows)) != GPDE_OK) return rc;
            }
            na = nb;
            continue;
        }
        hipLaunchKernelGGL(k_scale_g, dim3((nn + 3) / 4), dim3(T), 0, st, grad_out, rowptr, aggr, na, nn, gT);
        if (rows > 0) {
            // the chunk's per-edge part on the one-pass kernel: in-degree >= 32 (a 64-slot tile rarely spans more than two
            // destinations), H not given, and - full backward - enough rows for the split GEMMs that take its by-products
            const bool use1 = onepass_ok && !chunk_h && (int64_t)rows >= (int64_t)32 * nn;
            const bool use1_by = use1 && !light && f16s_dw2 && f16s_du1 && rows >= 8192 && !SW.bwd_du_passes && !SW.bwd_du_transpose_pass;
            // hidden activations of the chunk's edges: recomputed, or rows of the given cache
            skip_store = use1;
            if (phase == BWD_FULL || light) { rc_na = na; rc_nb = nb; if ((rc = recompute(e0, rows, n - 1)) != GPDE_OK) return rc; }
            skip_store = false;
            const float* Hlast = chunk_h ? chunk_h : (phase == BWD_FULL || light) ? F(P.off_H[n - 1]) : hidden + (size_t)e0 * K2P;
            // Z of the chunk's nodes: kept by the forward (gpde_nnconv_fwd_keepz), else re-aggregated from the recomputed /
            // given activations
            if (z_saved) Z = const_cast<float*>(z_saved) + (size_t)na * GP_W * K2P;      // read only below
            else GP_HIP_CHECK(gpde_zero_async(Z, (size_t)nn * GP_W * K2P * 4, st));
            if (!z_saved) {
                GpdeFusedArgs f{};
                f.x = x; f.attr = edge_attr; f.rowptr = rowptr; f.src = src; f.dst = dst; f.perm = perm;
                f.hbuf = Hlast; f.zbuf = Z; f.k0 = dims[0]; f.K1P = 32; f.K2P = K2P;
                f.nc0 = na; f.nc1 = nb; f.e_chunk0 = e0;
                if (zagg16) { f.xs = (const unsigned*)F(P.off_xs); f.scal = (const unsigned*)F(P.off_scal); f.hmax = (const unsigned*)F(P.off_scal) + 2; }
                const int ns = K2P / GP_TN;
                int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
                const int gcap = (rows / GP_TE + GP_WAVES) / GP_WAVES; if (groups > gcap) groups = gcap;
                f.n_groups = groups;
                if ((rc = gpde_launch_zagg(f, st)) != GPDE_OK) return rc;
            }
            hipLaunchKernelGGL(k_nbr_sum, dim3(nn), dim3(T), 0, st, x, rowptr, src, na, nn, S);
            // db3[c][o] += S^T gT ;  dW3[c][o][k] += gT^T Z[:, c, :]
            if ((rc = gemm_tn_acc(S, GP_W, GP_W, gT, GP_W, GP_W, nn, F(P.off_db3), GP_W, F(P.off_part), P.part_floats, 1, st)) != GPDE_OK) return rc;
            {
                GpdeGemmArgs g = gemm0();
                g.A = gT; g.lda = GP_W; g.a_kcontig = 0; g.B = Z; g.ldb = GP_W * K2P; g.b_kcontig = 0;
                g.C = F(P.off_dw3p); g.ldc = K2P; g.M = GP_W; g.N = K2P; g.K = nn; g.accumulate = 1;
                g.batches = GP_W; g.strideA = 0; g.strideB = K2P; g.strideC = (size_t)GP_W * K2P;
                // narrow kernels (the MGKN levels: K2P = 128 / 256) give 64 - 128 workgroups that each walk all nn nodes (365 us at
                // nn = 4525): split the node range (ordered partial reduction: deterministic) until ~512 workgroups are busy.  The
                // partials live in the chunk's dZ buffer - nn x 64 x K2P floats that are written only by the next GEMM below, and
                // sp <= nn / 128 of these 64 x 64 x K2P images always fit
                int sp = 1;
                const int wgs = GP_W * ((K2P + 127) / 128);
                while (sp < 16 && wgs * sp < 512 && nn / (sp * 2) >= 64) sp *= 2;
                if (sp > 1) {
                    g.C = dZ; g.accumulate = 0; g.splits = sp; g.strideSplit = w3n;
                    if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
                    if ((rc = gpde_launch_reduce_splits(dZ, w3n, sp, w3n, F(P.off_dw3p), 1, st)) != GPDE_OK) return rc;
                } else if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
            }
            // dZ[i][c][k] = sum_o gT[i][o] W3[c*64+o][k] ;  dS[i][c] = sum_o gT[i][o] b3[c*64+o]
            {
                GpdeGemmArgs g = gemm0();
                g.A = gT; g.lda = GP_W; g.B = F(P.off_w3p); g.ldb = K2P; g.b_kcontig = 0;
                g.C = dZ; g.ldc = GP_W * K2P; g.M = nn; g.N = K2P; g.K = GP_W;
                g.batches = GP_W; g.strideA = 0; g.strideB = (size_t)GP_W * K2P; g.strideC = K2P;
                if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
                GpdeGemmArgs s2 = gemm0();
                s2.A = gT; s2.lda = GP_W; s2.B = F(P.off_b3); s2.ldb = GP_W; s2.C = dS; s2.ldc = GP_W;
                s2.M = nn; s2.N = GP_W; s2.K = GP_W;
                if ((rc = gpde_launch_gemm(s2, st)) != GPDE_OK) return rc;
            }
            // per-edge backward through the aggregation -> dU_{n-1}, dx_j
            float* dUc = phase == BWD_FULL ? F(P.off_dU[0]) : light ? nullptr : grad_hidden_out + (size_t)e0 * K2P;   // light: dx only
            if (use1) {
                // dZ_i as the two split images (img2 in place, img1 into the Z buffer - Z itself is the forward's: z_saved), then
                // ONE kernel: K loop + dU_2 (+ transposed copy, row maxima, tile column statistics) + per-slice partial dx rows
                du_pre = false;
                if (!dx) { gpde_set_error("gpde_nnconv_bwd: grad_x must be provided"); return GPDE_EINVAL; }
                if ((rc = gpde_launch_dz_images(dZ, nn, K2P, F(P.off_Z), F(P.off_dzun), st)) != GPDE_OK) return rc;
                const float* pk = F(P.off_pack);
                GpdeFusedArgs f{};
                f.attr = edge_attr; f.rowptr = rowptr; f.perm = perm; f.src = src; f.dst = dst; f.kt = kt;
                for (int d_ = 0; d_ < 8; ++d_) f.sel[d_] = nas.sel[d_];
                f.w1 = pk + PL.off_w1; f.w2t = pk + PL.off_w2t; f.b2 = pk + PL.off_b2;
                f.w2h = pk + PL.off_w2h; f.ucol = pk + PL.off_ucol; f.w1h = pk + PL.off_w1h; f.fcol = pk + PL.off_fcol;
                f.k0 = PL.k0; f.K1P = PL.K1P; f.K2P = PL.K2P; f.nc0 = na; f.nc1 = nb; f.e_chunk0 = e0;
                f.xs = (const unsigned*)x; f.scal = (const unsigned*)F(P.off_scal);
                f.bw_img1 = F(P.off_Z); f.bw_img2 = dZ; f.bw_unscale = F(P.off_dzun); f.bw_dS = dS;
                f.bw_dxp = F(P.off_H[n - 1]); f.bw_rows = rows;
                const int ns = K2P / GP_TN;
                if (dUc) f.bw_dU = dUc;
                if (use1_by) {
                    f.bw_dUt = gpde_gemm_f16s_tn_at(F(P.off_tnws), rows, tn_ksplits(rows, P.KP[2], P.KP[1]), &f.bw_ldt);
                    f.bw_rowmax = F(P.off_dxe);            // [ns][rows] (the per-edge dx rows of the two-pass form are not used)
                    f.bw_csum = F(P.off_tcs); f.bw_cmax = (unsigned*)F(P.off_tcm);
                }
                int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
                const int gcap = (rows / 64 + GP_WAVES) / GP_WAVES; if (groups > gcap) groups = gcap;
                f.n_groups = groups;
                if ((rc = gpde_launch_fused_bwd(f, st)) != GPDE_OK) return rc;
                if (use1_by) {
                    if ((rc = gpde_launch_row_scales_from_slices(F(P.off_dxe), ns, rows, F(P.off_rowsc), F(P.off_rowsc) + rows, st)) != GPDE_OK) return rc;
                    du_pre = true;
                }
                hipLaunchKernelGGL(k_dx_reduce, dim3((N + 3) / 4), dim3(T), 0, st, F(P.off_H[n - 1]), src_rowptr, src_slots, N, e0, e1, dx, ns,
                                   (size_t)rows * GP_W, 0);
            } else {
                const bool ordered = src_rowptr && src_slots;
                EdgeBwdArgs ea{x, rowptr, src, dst, dZ, dS, Hlast, dUc, dx, e0, e1, na, K2P, ordered ? F(P.off_dxe) : nullptr};
                const size_t lds = (size_t)4 * (32 * EB_XS + 32 * EB_HS) * 4;
                const size_t lds2 = (size_t)(2 * EB2_DZ + 4 * 2 * EB2_H) * 4 + 4 * 64 * 4;
                static GpdeLdsOnce once;
                if (int rc_ = once.ensure(gpde_edge_bwd_kernel, gpde_edge_bwd2_kernel)) return rc_;
                if (!dx) { gpde_set_error("gpde_nnconv_bwd: grad_x must be provided"); return GPDE_EINVAL; }
                // staged kernel where a 128-slot group rarely spans more than two destinations
                const int force = SW.edge_bwd;                     // GPDE_EDGE_BWD = 1 / 2 / 3: force a variant (tests, A/B)
                // (accumulating dL/dU lives in the split-f16 kernel: it is correct for any in-degree, the host asks for it on dense graphs)
                const bool gh_acc = gh_accumulate && phase == BWD_CONV;
                // (round 6: from mean in-degree 4 - the split-f16 kernel is correct for any in-degree and measured faster on the MGKN
                // levels of in-degree 6 - 22 too: 1.4 ms of a 30 ms training step of config 4)
                const bool staged = force ? force >= 2 : (gh_acc || (int64_t)rows >= (int64_t)(K2P % 32 == 0 ? 4 : 32) * nn);
                du_pre = false;
                if (staged && force != 2 && K2P % 32 == 0) {      // split-f16 MFMA (default); GPDE_EDGE_BWD=2: the fp32-MFMA staged kernel
                    if ((rc = gpde_launch_dz_split(dZ, nn, K2P, F(P.off_dzun), st)) != GPDE_OK) return rc;
                    GpdeEdgeBwd3Args e3{x, src, dst, dZ, F(P.off_dzun), dS, Hlast, dUc, dx, ordered ? F(P.off_dxe) : nullptr, e0, e1, na, K2P};
                    e3.du_accumulate = gh_acc;
                    // full backward on the split GEMMs: the kernel also leaves what the dW_2 GEMM's pass over dU_2 would form
                    // (mlp_backward's tn_split && du_one_pass case; GPDE_BWD_DU_TRANSPOSE_PASS=1: that pass, A/B)
                    if (phase == BWD_FULL && n == 3 && f16s_dw2 && f16s_du1 && rows >= 8192 && !SW.bwd_du_passes &&
                        !SW.bwd_du_transpose_pass) {
                        e3.dUt = gpde_gemm_f16s_tn_at(F(P.off_tnws), rows, tn_ksplits(rows, P.KP[2], P.KP[1]), &e3.ldt);
                        e3.row_sc = F(P.off_rowsc); e3.row_isc = F(P.off_rowsc) + rows;
                        e3.csum_part = F(P.off_tcs); e3.cmax_part = (unsigned*)F(P.off_tcm);
                        du_pre = true;
                    }
                    if ((rc = gpde_launch_edge_bwd3(e3, st)) != GPDE_OK) return rc;
                } else if (gh_acc) {
                    gpde_set_error("gpde_nnconv_bwd: GPDE_BWD_ACCUMULATE_GRAD_HIDDEN is built into the split-f16 per-edge kernel only "
                                   "(GPDE_EDGE_BWD=2 forces the other one)");
                    return GPDE_EUNSUPPORTED;
                } else if (staged) hipLaunchKernelGGL(gpde_edge_bwd2_kernel, dim3(((rows + 127) / 128 + 7) / 8 * 8), dim3(T), lds2, st, ea);
                else hipLaunchKernelGGL(gpde_edge_bwd_kernel, dim3((rows + 127) / 128), dim3(T), lds, st, ea);
                if (ordered) hipLaunchKernelGGL(k_dx_reduce, dim3((N + 3) / 4), dim3(T), 0, st, F(P.off_dxe), src_rowptr, src_slots, N, e0, e1, dx, 1, (size_t)0, 0);
            }
            // MLP backward over the chunk's edges
            if (phase == BWD_FULL) { mlp_e0 = e0; if ((rc = mlp_backward(dUc, rows)) != GPDE_OK) return rc; }
        }
        na = nb;
    }
    GP_LAUNCH_CHECK("gpde_nnconv_bwd kernels");

    // ---- node-side terms of update(): dx += g root^T, droot = X^T g, dbias = colsum g ----------------------
    if (do_conv && (rc = bwd_node_terms(x, N, root, grad_out, dx, grad_root, grad_bias, F(P.off_part), P.part_floats, st)) != GPDE_OK) return rc;
    // ---- un-pad the weight gradients into torch layout -------------------------------------------------------
    for (int l = 1; l < n && do_mlp; ++l) {
        if (grad_W && grad_W[l - 1])
            hipLaunchKernelGGL(k_unpad_mat, dim3(nblk((size_t)dims[l] * dims[l - 1])), dim3(T), 0, st, F(P.off_dwp[l]),
                               dims[l], dims[l - 1], P.KP[l - 1], grad_W[l - 1]);
        if (grad_b && grad_b[l - 1])
            GP_HIP_CHECK(gpde_copy_async(grad_b[l - 1], F(P.off_dbp[l]), (size_t)dims[l] * 4, st));
    }
    if (do_conv && grad_W && grad_W[n - 1])
        hipLaunchKernelGGL(k_unpad_mat, dim3(nblk((size_t)GP_W * GP_W * dims[n - 1])), dim3(T), 0, st, F(P.off_dw3p),
                           GP_W * GP_W, dims[n - 1], K2P, grad_W[n - 1]);
    if (do_conv && grad_b && grad_b[n - 1])
        GP_HIP_CHECK(gpde_copy_async(grad_b[n - 1], F(P.off_db3), GP_W * GP_W * 4, st));
    GP_LAUNCH_CHECK("gpde_nnconv_bwd epilogue kernels");
    return GPDE_OK;
}

}  // namespace

extern "C" size_t gpde_nnconv_bwd_workspace_bytes_one_chunk(int64_t n_nodes, int64_t n_edges, int n_layers, const int32_t* dims) {
    BwdPlan P;
    if (!dims || n_nodes < 0 || n_edges < 0) return 0;
    if (make_bwd_plan(n_nodes, n_edges, n_layers, dims, 0, true, &P) != GPDE_OK) return 0;
    if (P.Ec >= n_edges && P.Nc >= n_nodes) return P.total;          // the default size already runs everything as one chunk
    return P.one_chunk > P.total ? P.one_chunk : P.total;
}

// attributes described by node data (include/gpde.h GpdeNodeAttr)
namespace {
bool na_ok(const GpdeNodeAttr* na, const int32_t* dims, const char* who) {
    if (!na || !na->table || na->stride < 1 || !dims || na->n_slots != dims[0] || dims[0] < 1 || dims[0] > 7) {
        gpde_set_error("%s: GpdeNodeAttr must describe dims[0] = 1..7 slots of a node table", who);
        return false;
    }
    return true;
}
}  // namespace

// The backward of the operator, one entry point for its forms (round 5: the plain / source-ordered / kept-Z / given-hidden /
// edge-attribute-gradient / node-table calls of rounds 1-4 folded into this one):
//   attributes  `edge_attr` + `perm` (a tensor in the caller's edge order) | `node_attr` (read from node data, row f3) |
//               `hidden` (the last hidden activations given, [CSR slot][K2P]: only the last Linear, root, bias, x are
//               differentiated and dL/dU of the last hidden layer is written to `grad_hidden`; W / b / grad_W / grad_b then
//               carry their LAST entries only)
//   src_rowptr / src_slots   (nullable) the CSR slots regrouped by source (gpde_csr_source_order): grad_x is summed in slot
//               order, bit-reproducible; NULL: fp32 atomics
//   z_saved     (nullable) Z of the keep-Z forward: dW_3 is taken from it instead of re-aggregating
//   grad_edge_attr (nullable; tensor attributes of <= 8 slots only) dL/d edge_attr [E][k0] in the caller's edge order
extern "C" int gpde_nnconv_bwd(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr,
                               const float* hidden, int64_t n_edges, const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                               const int32_t* perm, const int32_t* rowptr_host, const int32_t* src_rowptr, const int32_t* src_slots,
                               int n_layers, const int32_t* dims, const float* const* W, const float* const* b, const float* root,
                               int aggr, const float* grad_out, const float* z_saved, float* grad_x, float* grad_hidden,
                               float* grad_edge_attr, float* const* grad_W, float* const* grad_b, float* grad_root,
                               float* grad_bias, uint32_t flags, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !W || !b || !grad_out || !rowptr || !rowptr_host || !ws || !grad_W || !grad_b ||
        n_layers < 2 || n_layers > GPDE_MAX_LAYERS || (n_nodes > 0 && (!x || !grad_x)) || (n_edges > 0 && (!src || !dst))) {
        gpde_set_error("gpde_nnconv_bwd: null/negative argument");
        return GPDE_EINVAL;
    }
    if (aggr != GPDE_AGGR_ADD && aggr != GPDE_AGGR_MEAN) { gpde_set_error("gpde_nnconv_bwd: aggr %d", aggr); return GPDE_EUNSUPPORTED; }
    // `hidden` TOGETHER with an attribute source and no grad_hidden: the FULL backward with the last hidden activations kept by the
    // forward (gpde_hidden_fwd + gpde_nnconv_fwd_keepz(hidden)) - read where they would be recomputed (round 5)
    const bool h_kept = hidden && !grad_hidden && (node_attr || (edge_attr && perm));
    if ((hidden && !h_kept && node_attr) || (grad_edge_attr && (hidden || node_attr))) {
        gpde_set_error("gpde_nnconv_bwd: one attribute source (edge_attr + perm | node_attr | hidden); grad_edge_attr needs the tensor");
        return GPDE_EINVAL;
    }
    if ((flags & GPDE_BWD_ACCUMULATE_GRAD_HIDDEN) && !(hidden && !h_kept)) {
        gpde_set_error("gpde_nnconv_bwd: GPDE_BWD_ACCUMULATE_GRAD_HIDDEN belongs to the `hidden` form (grad_hidden given)");
        return GPDE_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream_;
    const float* hk = h_kept ? hidden : nullptr;
    const int64_t hk_nodes = h_kept ? n_nodes : 0;
    if (hidden && !h_kept) {
        if (n_edges > 0 && !grad_hidden) { gpde_set_error("gpde_nnconv_bwd: grad_hidden is null"); return GPDE_EINVAL; }
        return bwd_impl(BWD_CONV, x, n_nodes, nullptr, n_edges, rowptr, src, dst, nullptr, rowptr_host, n_layers, dims, W, b, root,
                        aggr, grad_out, grad_x, grad_W, grad_b, grad_root, grad_bias, hidden, grad_hidden, nullptr, ws, ws_bytes,
                        st, src_rowptr, src_slots, z_saved, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr,
                        (flags & GPDE_BWD_ACCUMULATE_GRAD_HIDDEN) != 0);
    }
    if (node_attr) {
        if (!na_ok(node_attr, dims, "gpde_nnconv_bwd")) return GPDE_EINVAL;
        return bwd_impl(BWD_FULL, x, n_nodes, node_attr->table, n_edges, rowptr, src, dst, nullptr, rowptr_host, n_layers, dims, W, b, root,
                        aggr, grad_out, grad_x, grad_W, grad_b, grad_root, grad_bias, nullptr, nullptr, nullptr, ws, ws_bytes, st,
                        src_rowptr, src_slots, z_saved, 0, nullptr, nullptr, hk, hk_nodes, node_attr->stride, node_attr->sel);
    }
    if (n_edges > 0 && (!edge_attr || !perm)) { gpde_set_error("gpde_nnconv_bwd: edge_attr / perm is null"); return GPDE_EINVAL; }
    return bwd_impl(BWD_FULL, x, n_nodes, edge_attr, n_edges, rowptr, src, dst, perm, rowptr_host, n_layers, dims, W, b, root, aggr,
                    grad_out, grad_x, grad_W, grad_b, grad_root, grad_bias, nullptr, nullptr, nullptr, ws, ws_bytes, st,
                    src_rowptr, src_slots, z_saved, 0, nullptr, nullptr, hk, hk_nodes, 0, nullptr, n_edges > 0 ? grad_edge_attr : nullptr);
}

// ---- depth-deferred backward: a module applied `depth` times with the same edge_attr and weights ----------------------
// (KernelNN.forward applies ONE conv1 `depth` times, /root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30, and
// loss.backward(), :266, sums the kernel MLP's gradients over those uses.)  When the hidden activations do not fit memory
// every application runs gpde_nnconv_bwd_light (grad_x and the node-side gradients only) and ONE gpde_nnconv_bwd_deferred
// pass forms the hidden layers' gradients of all of them.
extern "C" int gpde_nnconv_bwd_deferred_supported(int n_layers, const int32_t* dims) {
    BwdPlan P;
    GpdePackLayout PL;
    if (!dims || n_layers != 3 || make_bwd_plan(1, 1, n_layers, dims, 0, true, &P, 2) != GPDE_OK) return 0;
    if (!(P.f16s_du1 && P.f16s_dw2 && P.pack_bytes) || dims[0] > 7 || gpde_pack_layout(n_layers, dims, &PL) != GPDE_OK || PL.mode != 1) return 0;
    GpdeFusedArgs probe{};
    probe.k0 = PL.k0; probe.K1P = PL.K1P; probe.K2P = PL.K2P;
    return gpde_fused_store_supported(probe) ? 1 : 0;
}

extern "C" size_t gpde_nnconv_bwd_deferred_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers, const int32_t* dims,
                                                           int n_defer) {
    BwdPlan P;
    if (!dims || n_nodes < 0 || n_edges < 0 || n_defer < 1) return 0;
    if (make_bwd_plan(n_nodes, n_edges, n_layers, dims, 0, true, &P, n_defer) != GPDE_OK) return 0;
    return P.total;
}

extern "C" int gpde_nnconv_bwd_light(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* rowptr,
                                     const int32_t* src, const int32_t* dst, const int32_t* perm, const int32_t* rowptr_host,
                                     const int32_t* src_rowptr, const int32_t* src_slots, int n_layers, const int32_t* dims,
                                     const float* const* W, const float* const* b, const float* root, int aggr, const float* grad_out,
                                     const float* z_saved, const float* hidden_part, int64_t hidden_nodes, float* grad_x,
                                     float* grad_w_last, float* grad_b_last, float* grad_root,
                                     float* grad_bias, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !W || !b || !grad_out || !rowptr || !rowptr_host || !ws || n_layers < 2 ||
        n_layers > GPDE_MAX_LAYERS || (n_nodes > 0 && (!x || !grad_x)) || (n_edges > 0 && (!src || !dst || (!node_attr && (!edge_attr || !perm)))) ||
        hidden_nodes < 0 || hidden_nodes > n_nodes || (hidden_nodes > 0 && !hidden_part)) {
        gpde_set_error("gpde_nnconv_bwd_light: null/negative argument");
        return GPDE_EINVAL;
    }
    if (node_attr && !na_ok(node_attr, dims, "gpde_nnconv_bwd_light")) return GPDE_EINVAL;
    if (aggr != GPDE_AGGR_ADD && aggr != GPDE_AGGR_MEAN) { gpde_set_error("gpde_nnconv_bwd_light: aggr %d", aggr); return GPDE_EUNSUPPORTED; }
    float* gW[GPDE_MAX_LAYERS] = {};
    float* gb[GPDE_MAX_LAYERS] = {};
    gW[n_layers - 1] = grad_w_last; gb[n_layers - 1] = grad_b_last;
    return bwd_impl(BWD_LIGHT, x, n_nodes, node_attr ? node_attr->table : edge_attr, n_edges, rowptr, src, dst, node_attr ? nullptr : perm,
                    rowptr_host, n_layers, dims, W, b, root, aggr,
                    grad_out, grad_x, gW, gb, grad_root, grad_bias, nullptr, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream_,
                    src_rowptr, src_slots, z_saved, 0, nullptr, nullptr, hidden_nodes > 0 ? hidden_part : nullptr, hidden_nodes,
                    node_attr ? node_attr->stride : 0, node_attr ? node_attr->sel : nullptr);
}

extern "C" int gpde_nnconv_bwd_deferred(const float* x_stack, const float* grad_out_stack, int n_defer, int64_t n_nodes,
                                        const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* rowptr, const int32_t* src,
                                        const int32_t* dst, const int32_t* perm, const int32_t* rowptr_host, int n_layers,
                                        const int32_t* dims, const float* const* W, const float* const* b, int aggr,
                                        const float* hidden_part, int64_t hidden_nodes,
                                        float* const* grad_W, float* const* grad_b, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || n_defer < 1 || !dims || !W || !b || !grad_W || !grad_b || !rowptr || !rowptr_host || !ws ||
        n_layers < 2 || n_layers > GPDE_MAX_LAYERS || (n_nodes > 0 && (!x_stack || !grad_out_stack)) ||
        (n_edges > 0 && (!src || !dst || (!node_attr && (!edge_attr || !perm)))) || hidden_nodes < 0 || hidden_nodes > n_nodes ||
        (hidden_nodes > 0 && !hidden_part)) {
        gpde_set_error("gpde_nnconv_bwd_deferred: null/negative argument");
        return GPDE_EINVAL;
    }
    if (node_attr && !na_ok(node_attr, dims, "gpde_nnconv_bwd_deferred")) return GPDE_EINVAL;
    if (aggr != GPDE_AGGR_ADD && aggr != GPDE_AGGR_MEAN) { gpde_set_error("gpde_nnconv_bwd_deferred: aggr %d", aggr); return GPDE_EUNSUPPORTED; }
    return bwd_impl(BWD_DEFER, nullptr, n_nodes, node_attr ? node_attr->table : edge_attr, n_edges, rowptr, src, dst, node_attr ? nullptr : perm,
                    rowptr_host, n_layers, dims, W, b, nullptr,
                    aggr, nullptr, nullptr, grad_W, grad_b, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                    (hipStream_t)stream_, nullptr, nullptr, nullptr, n_defer, x_stack, grad_out_stack,
                    hidden_nodes > 0 ? hidden_part : nullptr, hidden_nodes, node_attr ? node_attr->stride : 0,
                    node_attr ? node_attr->sel : nullptr);
}

extern "C" int gpde_hidden_bwd(const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* perm,
                               const int32_t* src, const int32_t* dst, int n_layers,
                               const int32_t* dims, const float* const* W, const float* const* b,
                               const float* grad_hidden, float* const* grad_W, float* const* grad_b, void* ws,
                               size_t ws_bytes, void* stream_) {
    if (n_edges < 0 || !dims || !W || !b || !ws || !grad_W || !grad_b ||
        (n_edges > 0 && (!grad_hidden || (node_attr ? (!src || !dst) : (!edge_attr || !perm))))) {
        gpde_set_error("gpde_hidden_bwd: null/negative argument");
        return GPDE_EINVAL;
    }
    if (node_attr) {
        if (!na_ok(node_attr, dims, "gpde_hidden_bwd")) return GPDE_EINVAL;
        return bwd_impl(BWD_MLP, nullptr, 0, node_attr->table, n_edges, nullptr, src, dst, nullptr, nullptr, n_layers, dims, W, b, nullptr,
                        GPDE_AGGR_ADD, nullptr, nullptr, grad_W, grad_b, nullptr, nullptr, nullptr, nullptr, grad_hidden, ws, ws_bytes,
                        (hipStream_t)stream_, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, node_attr->stride, node_attr->sel);
    }
    return bwd_impl(BWD_MLP, nullptr, 0, edge_attr, n_edges, nullptr, nullptr, nullptr, perm, nullptr, n_layers, dims,
                    W, b, nullptr, GPDE_AGGR_ADD, nullptr, nullptr, grad_W, grad_b, nullptr, nullptr, nullptr, nullptr,
                    grad_hidden, ws, ws_bytes, (hipStream_t)stream_);
}


// ---- cross-depth reuse: the hidden activations as a tensor -------------------------
// `conv1` is applied depth x with the same edge_attr and weights
// (/root/reference/graph-neural-operator/UAI1_full_resolution.py:29-30; the MGKN V-cycle,
// multipole-graph-neural-operator/MGKN_general_darcy2d.py:76-90): H_e = relu(L_{n-1}(...relu(L_1(attr_e))))
// is identical in all of them.  gpde_hidden_fwd writes it once ([CSR slot][K2P] fp32), the *_hidden
// entry points consume it.
extern "C" size_t gpde_hidden_workspace_bytes(int64_t n_edges, int n_layers, const int32_t* dims) {
    BwdPlan P;
    if (!dims || n_edges < 0) return 0;
    if (make_bwd_plan(1, n_edges, n_layers, dims, 0, true, &P) != GPDE_OK) return 0;
    return P.total;
}

extern "C" int gpde_hidden_fwd(const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* rowptr, int64_t n_nodes,
                               const int32_t* perm, const int32_t* src, const int32_t* dst, int n_layers, const int32_t* dims, const void* packed,
                               const float* const* W, const float* const* b, uint32_t flags, float* hidden,
                               float* hidden_absmax, void* ws, size_t ws_bytes, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    // hidden_absmax (nullable, one float): max |H|, recorded by the fused path only (0 = not recorded);
    // it lets gpde_nnconv_fwd_hidden run the aggregation on split-f16 MFMA
    if (n_edges < 0 || n_nodes < 0 || !dims || (n_edges > 0 && (!hidden || !rowptr || (node_attr ? (!src || !dst || !packed) : (!edge_attr || !perm))))) {
        gpde_set_error("gpde_hidden_fwd: null/negative argument");
        return GPDE_EINVAL;
    }
    if (node_attr && !na_ok(node_attr, dims, "gpde_hidden_fwd")) return GPDE_EINVAL;
    if (hidden_absmax) GP_HIP_CHECK(gpde_zero_async(hidden_absmax, sizeof(float), st));
    if (node_attr) {          // attributes from node data: the one-wave-per-SIMD store kernel reads the table
        const GpdeNodeAttr* na = node_attr;
        if (n_edges == 0) return GPDE_OK;
        GpdePackLayout L;
        int rc = gpde_pack_layout(n_layers, dims, &L);
        if (rc != GPDE_OK) return rc;
        if (!(flags & GPDE_FWD_F16SPLIT) || L.mode != 1) { gpde_set_error("gpde_hidden_fwd (node_attr): 3-Linear kernel MLPs on the split-f16 kernel only"); return GPDE_EUNSUPPORTED; }
        const float* pk = (const float*)packed;
        GpdeFusedArgs f{};
        f.attr = na->table; f.rowptr = rowptr; f.src = src; f.dst = dst; f.kt = na->stride;
        for (int d = 0; d < 8; ++d) f.sel[d] = na->sel[d < dims[0] ? d : dims[0] - 1];
        f.w1 = pk + L.off_w1; f.w2t = pk + L.off_w2t; f.b2 = pk + L.off_b2;
        f.w2h = pk + L.off_w2h; f.ucol = pk + L.off_ucol; f.w1h = pk + L.off_w1h; f.fcol = pk + L.off_fcol;
        f.hout = hidden; f.hmax_out = (unsigned*)hidden_absmax; f.k0 = L.k0; f.K1P = L.K1P; f.K2P = L.K2P;
        f.nc0 = 0; f.nc1 = (int)n_nodes; f.e_chunk0 = 0;
        const int ns = L.K2P / GP_TN;
        int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
        const int64_t gcap = ((n_edges + GP_TE - 1) / GP_TE + GP_WAVES - 1) / GP_WAVES;
        if (groups > gcap) groups = (int)gcap;
        f.n_groups = groups;
        if (!gpde_fused_f16v6_supported(f)) { gpde_set_error("gpde_hidden_fwd (node_attr): kernel MLP outside the one-wave-per-SIMD store kernel (>= 8 k1 chunks)"); return GPDE_EUNSUPPORTED; }
        return gpde_launch_fused_f16v6(f, st);
    }
    if (n_edges == 0) return GPDE_OK;
    GpdePackLayout L;
    int rc = gpde_pack_layout(n_layers, dims, &L);
    if (rc != GPDE_OK) return rc;
    // fast path: the f16-split fused kernel with a store epilogue instead of the aggregation
    if (packed && (flags & GPDE_FWD_F16SPLIT) && L.mode == 1) {
        const float* pk = (const float*)packed;
        GpdeFusedArgs f{};
        f.attr = edge_attr; f.rowptr = rowptr; f.perm = perm;
        f.w1 = pk + L.off_w1; f.w2t = pk + L.off_w2t; f.b2 = pk + L.off_b2;
        f.w2h = pk + L.off_w2h; f.ucol = pk + L.off_ucol; f.w1h = pk + L.off_w1h; f.fcol = pk + L.off_fcol;
        f.hout = hidden; f.hmax_out = (unsigned*)hidden_absmax; f.k0 = L.k0; f.K1P = L.K1P; f.K2P = L.K2P;
        f.nc0 = 0; f.nc1 = (int)n_nodes; f.e_chunk0 = 0;
        const int ns = L.K2P / GP_TN;
        int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
        const int64_t gcap = ((n_edges + GP_TE - 1) / GP_TE + GP_WAVES - 1) / GP_WAVES;
        if (groups > gcap) groups = (int)gcap;
        f.n_groups = groups;
        if (gpde_fused_store_supported(f)) return gpde_launch_fused_store(f, st);
    }
    // general path (any layer count, exact fp32): chunks of edges through the dense GEMM
    if (!W || !b || !ws) { gpde_set_error("gpde_hidden_fwd: weights / workspace needed for the general path"); return GPDE_EINVAL; }
    BwdPlan P;
    rc = make_bwd_plan(1, n_edges, n_layers, dims, ws_bytes, false, &P);
    if (rc != GPDE_OK) return rc;
    const int n = n_layers, K2P = P.K2P, T = 256;
    char* w = (char*)(((uintptr_t)ws + 255) / 256 * 256);
    auto F = [&](size_t off) { return (float*)(w + off); };
    for (int l = 1; l < n; ++l) {
        const size_t wn = (size_t)P.KP[l] * P.KP[l - 1];
        hipLaunchKernelGGL(k_pad_mat, dim3(nblk(wn)), dim3(T), 0, st, W[l - 1], dims[l], dims[l - 1], dims[l - 1],
                           P.KP[l], P.KP[l - 1], F(P.off_wp[l]));
        if (b[l - 1]) hipLaunchKernelGGL(k_pad_mat, dim3(nblk(P.KP[l])), dim3(T), 0, st, b[l - 1], 1, dims[l], dims[l], 1, P.KP[l], F(P.off_bp[l]));
        else GP_HIP_CHECK(gpde_zero_async(F(P.off_bp[l]), (size_t)P.KP[l] * 4, st));
    }
    for (int64_t e0 = 0; e0 < n_edges; e0 += P.Ec) {
        const int rows = (int)((n_edges - e0) < P.Ec ? (n_edges - e0) : P.Ec);
        hipLaunchKernelGGL(k_gather_attr, dim3(nblk((size_t)rows * P.KP[0])), dim3(T), 0, st, edge_attr, perm, (int)e0,
                           rows, dims[0], P.KP[0], F(P.off_H[0]));
        for (int l = 1; l < n; ++l) {
            GpdeGemmArgs g = gemm0();
            g.A = F(P.off_H[l - 1]); g.lda = P.KP[l - 1]; g.B = F(P.off_wp[l]); g.ldb = P.KP[l - 1];
            g.C = (l == n - 1) ? hidden + (size_t)e0 * K2P : F(P.off_H[l]);
            g.ldc = P.KP[l]; g.M = rows; g.N = P.KP[l]; g.K = P.KP[l - 1];
            g.bias = F(P.off_bp[l]); g.relu = 1;
            if ((rc = gpde_launch_gemm(g, st)) != GPDE_OK) return rc;
        }
    }
    GP_LAUNCH_CHECK("gpde_hidden_fwd kernels");
    return GPDE_OK;
}
