// Kernel-MLP weight repacking into MFMA-tile order (layout only, no arithmetic).
// Source layout: torch `nn.Linear` weight[out][in], bias[out], as held by DenseNet
// (/root/reference/graph-neural-operator/utilities.py:201-227).
#include "gpde_common.h"
#include <string.h>

int gpde_pack_layout(int n_layers, const int32_t* dims, GpdePackLayout* L) {
    memset(L, 0, sizeof(*L));
    if (n_layers < 2 || n_layers > GPDE_MAX_LAYERS) {
        gpde_set_error("kernel MLP must have 2..%d Linear layers, got %d", GPDE_MAX_LAYERS, n_layers);
        return GPDE_EUNSUPPORTED;
    }
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1) { gpde_set_error("dims[%d] = %d", l, dims[l]); return GPDE_EINVAL; }
    if (dims[n_layers] != GP_W * GP_W) {
        gpde_set_error("last layer must emit %d = width^2 values (width %d), got %d", GP_W * GP_W,
                       GP_W, dims[n_layers]);
        return GPDE_EUNSUPPORTED;
    }
    L->n_layers = n_layers;
    L->k0 = dims[0];
    const bool attr_fits = dims[0] + 1 <= 8;      // attributes + bias slot fit one K=8 MFMA group
    if (n_layers == 2 && attr_fits) L->mode = 0;
    else if (n_layers == 3 && attr_fits) L->mode = 1;
    else L->mode = 2;
    L->k2 = dims[n_layers - 1];
    L->K2P = gp_round_up(L->k2, GP_TN);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 3) / 4 * 4; return o; };
    if (L->mode == 0) {
        L->off_w1 = take((size_t)L->K2P * 8);
    } else if (L->mode == 1) {
        L->k1 = dims[1];
        L->K1P = gp_round_up(L->k1, GP_BK);
        L->off_w1 = take((size_t)(L->K1P + 1) * 8);
        L->off_w2t = take((size_t)L->K2P * L->K1P);
        L->off_b2 = take((size_t)L->K2P);
        L->off_w2h = take((size_t)L->K2P * L->K1P);
        L->off_ucol = take((size_t)L->K2P);
        L->off_w1h = take((size_t)L->K1P * 8);
        L->off_fcol = take(16);      // [0..7] 2^-u_d ; [8] max_k |b2_k| ; [9] max_k sum_j |W2[k][j]|
    } else {
        // front layers 0 .. n_layers-2 as dense layers; widths padded to 128 (inputs of layer 0: 32)
        L->frontKP[0] = gp_round_up(dims[0], 32);
        for (int l = 1; l <= n_layers - 1; ++l) L->frontKP[l] = gp_round_up(dims[l], GP_TN);
        for (int l = 0; l < n_layers - 1; ++l) {
            L->off_front_w[l] = take((size_t)L->frontKP[l + 1] * L->frontKP[l]);
            L->off_front_b[l] = take((size_t)L->frontKP[l + 1]);
        }
    }
    L->off_w3q = take((size_t)GP_W * L->K2P * GP_W);
    L->off_b3 = take((size_t)GP_W * GP_W);
    L->has_w3s = L->K2P >= 256;
    if (L->has_w3s) {
        L->off_w3s = take((size_t)GP_W * GP_W * L->K2P);
        L->off_ucol3 = take((size_t)GP_W * GP_W);
    }
    L->total_floats = off;
    return GPDE_OK;
}

namespace {

// W1 with the bias folded in as input slot k0:  out[row][h][s] = W1b[row][2s+h]
__global__ void pack_w1_kernel(const float* __restrict__ W, const float* __restrict__ b, int k_out,
                               int k0, int rowsP, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rowsP * 8) return;
    const int row = i >> 3, hh = (i >> 2) & 1, s = i & 3;
    const int d = 2 * s + hh;
    float v = 0.f;
    if (row < k_out) {
        if (d < k0) v = W[(size_t)row * k0 + d];
        else if (d == k0 && b) v = b[row];
    }
    out[i] = v;
}

// W2 [k2][k1] -> tiles [K2P/128][K1P/32][128][32], zero padded
__global__ void pack_w2_kernel(const float* __restrict__ W, int k2, int k1, int K2P, int K1P,
                               float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)K2P * K1P) return;
    const int k = i & 31;
    const int n = (i >> 5) & 127;
    const size_t tile = i >> 12;
    const int NKC = K1P / 32;
    const int kc = tile % NKC, slice = tile / NKC;
    const int ng = slice * 128 + n, kg = kc * 32 + k;
    out[i] = (ng < k2 && kg < k1) ? W[(size_t)ng * k1 + kg] : 0.f;
}

// per-input-slot max_k |W1b[k][d]| appended to the packed W1 as row `rows` ([h][s] order): the
// fused kernel bounds max_k |H1[e][k]| <= sum_d wmax[d] |attr_e[d]| with it (f16-split scaling)
__global__ __launch_bounds__(256) void pack_w1max_kernel(const float* __restrict__ w1p, int rows, float* __restrict__ out,
                                                         float* __restrict__ zero2) {
    // thread t: slot position t & 7 of rows t >> 3, + 32, ...; then the 32 row groups fold through LDS
    __shared__ float sm[256];
    const int d8 = threadIdx.x & 7;
    float m = 0.f;
    for (int r = threadIdx.x >> 3; r < rows; r += 32) m = fmaxf(m, fabsf(w1p[(size_t)r * 8 + d8]));
    sm[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < 8) {
#pragma unroll 8
        for (int g = 1; g < 32; ++g) m = fmaxf(m, sm[g * 8 + threadIdx.x]);
        out[threadIdx.x] = m;
    }
    if (threadIdx.x < 2 && zero2) zero2[threadIdx.x] = 0.f;          // the two maxima pack_g2_consts_kernel forms with atomicMax
}

// (W1|b1) -> f16 two-term split image for the f16 H1 generation: column d (input slot) is scaled
// by 2^u_d so that its largest magnitude lies in [2^6, 2^7); fcol[d] = 2^-u_d (0 if the column is
// all zero) is applied to the attributes in the kernel.  w1p/wmax are the fp32 packed W1 ([h][s]
// order: slot d = 2s+h at position h*4+s) and its appended max row.
__global__ void pack_w1_f16split_kernel(const float* __restrict__ w1p, int rows, _Float16* __restrict__ out,
                                        float* __restrict__ fcol) {
    __shared__ float sc[8];
    if (threadIdx.x < 8) {
        const int d = threadIdx.x;
        const float m = w1p[(size_t)rows * 8 + (d & 1) * 4 + (d >> 1)];
        const int eb = (__float_as_int(m) >> 23) & 0xff;
        const bool ok = (m > 0.f) && eb >= 20 && eb <= 230;
        const int u = ok ? 6 - (eb - 127) : 0;
        sc[d] = ok ? __int_as_float((u + 127) << 23) : 0.f;
        if (blockIdx.x == 0) fcol[d] = ok ? __int_as_float((127 - u) << 23) : 0.f;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 8) return;
    const int row = i >> 3, d = i & 7;
    const float w = w1p[(size_t)row * 8 + (d & 1) * 4 + (d >> 1)] * sc[d];
    const _Float16 hi = (_Float16)w;
    out[(size_t)row * 16 + d] = hi;
    out[(size_t)row * 16 + 8 + d] = (_Float16)(w - (float)hi);
}

// W2 [k2][k1] -> f16 two-term split tiles. Row n is scaled by 2^t_n so that its largest
// magnitude lies in [2^13, 2^14); hi = rn16(w), lo = rn16(w - hi).  Within a 32-wide k chunk the
// halves are stored in MFMA operand order: position p = (m*2+h)*8 + j  <->
// k = 16m + 8(j>>2) + 4h + (j&3)   (matches the D-layout of the on-the-fly H1);
// rows are 128 B = 8 units of 16 B: units 0-3 = hi (m,h), 4-7 = lo (m,h), XOR-swizzled (below).
__global__ void pack_w2_f16split_kernel(const float* __restrict__ W, int k2, int k1, int K2P,
                                        int K1P, _Float16* __restrict__ out,
                                        float* __restrict__ ucol) {
    __shared__ float red[256];
    const int ng = blockIdx.x;           // hidden row (output unit of layer 2)
    float m = 0.f;
    if (ng < k2)
        for (int k = threadIdx.x; k < k1; k += blockDim.x) m = fmaxf(m, fabsf(W[(size_t)ng * k1 + k]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    m = red[0];
    int tn = 0;
    const int eb = (__float_as_int(m) >> 23) & 0xff;
    if (eb >= 20 && eb <= 230) tn = 13 - (eb - 127);
    const float scale = __int_as_float((tn + 127) << 23);
    if (threadIdx.x == 0) ucol[ng] = __int_as_float((127 - tn) << 23);
    const int NKC = K1P / 32;
    const int slice = ng / 128, nl = ng % 128;
    for (int k = threadIdx.x; k < K1P; k += blockDim.x) {
        const float w = (ng < k2 && k < k1) ? W[(size_t)ng * k1 + k] * scale : 0.f;
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        const int kc = k >> 5, kk = k & 31;
        const int mm = kk >> 4, k16 = kk & 15;
        const int jh = k16 >> 3, hh = (k16 >> 2) & 1, jl = k16 & 3;
        // 16-byte unit q = (part*4 + m*2 + h) is stored at unit q ^ ((n>>1)&7): a linear (DMA) copy
        // of the tile into LDS is then conflict-free for ds_read_b128 without row padding
        const int sw = (nl >> 1) & 7;
        const int uh = (mm * 2 + hh) ^ sw, ul = (4 + mm * 2 + hh) ^ sw;
        const int j = jh * 4 + jl;
        _Float16* row = out + ((size_t)(slice * NKC + kc) * 128 + nl) * 64;
        row[uh * 8 + j] = hi;
        row[ul * 8 + j] = lo;
    }
}

__global__ void pack_pad_vec_kernel(const float* __restrict__ v, int n, int nP, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nP) out[i] = (v && i < n) ? v[i] : 0.f;
}

// W [kout][kin] -> [KoutP][KinP] zero padded
__global__ void pack_pad_mat_kernel(const float* __restrict__ W, int kout, int kin, int KoutP,
                                    int KinP, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)KoutP * KinP) return;
    const int c = i % KinP;
    const int r = i / KinP;
    out[i] = (r < kout && c < kin) ? W[(size_t)r * kin + c] : 0.f;
}

// W3 [4096][k2] -> w3q[c][k/4][o][k%4]  (= W3[c*64+o][k]), zero padded in k
__global__ void pack_w3_kernel(const float* __restrict__ W3, int k2, int K2P, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)GP_W * K2P * GP_W) return;
    const int k4 = i & 3;
    const int o = (i >> 2) & 63;
    const size_t rest = i >> 8;              // c * (K2P/4) + kq
    const int kq = rest % (K2P / 4);
    const int c = rest / (K2P / 4);
    const int k = kq * 4 + k4;
    out[i] = (k < k2) ? W3[((size_t)c * GP_W + o) * k2 + k] : 0.f;
}

// a-priori bound of the last hidden layer for the f16-split aggregation (gpde_fused_f16v3.hip):
// h_e[k] <= |b2_k| + (sum_j |W2[k][j]|) * max_j H1_e[j];  out[0] = max_k |b2_k|, out[1] = max_k ||W2_k||_1
// One wave per row of W2 (coalesced), maxima of non-negative floats as bit patterns; out[0..1] zeroed by pack_w1max_kernel
// (same stream, earlier).  (Rounds 1-5: one workgroup, a thread per row striding k1 floats - 55 us at 1024 x 1024, once per
// module and optimisation step.)
__global__ __launch_bounds__(256) void pack_g2_consts_kernel(const float* __restrict__ W2, const float* __restrict__ b2, int k2,
                                                             int k1, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= k2) return;
    float l1 = 0.f;
    for (int j = lane; j < k1; j += 64) l1 += fabsf(W2[(size_t)k * k1 + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l1 += __shfl_xor(l1, o);
    if (lane == 0) {
        atomicMax((unsigned*)out + 1, __float_as_uint(l1));
        if (b2) atomicMax((unsigned*)out, __float_as_uint(fabsf(b2[k])));
    }
}

}  // namespace

int gpde_pack_split_nk(const float* Wnk, int n, int k, int NP, int KP, void* out, float* ucol, hipStream_t stream) {
    hipLaunchKernelGGL(pack_w2_f16split_kernel, dim3(NP), dim3(256), 0, stream, Wnk, n, k, NP, KP, (_Float16*)out, ucol);
    GP_LAUNCH_CHECK("pack_w2_f16split_kernel");
    return GPDE_OK;
}

extern "C" size_t gpde_mlp_pack_bytes(int n_layers, const int32_t* dims) {
    GpdePackLayout L;
    if (!dims || gpde_pack_layout(n_layers, dims, &L) != GPDE_OK) return 0;
    return L.total_floats * sizeof(float);
}

extern "C" int gpde_mlp_pack(int n_layers, const int32_t* dims, const float* const* W,
                             const float* const* b, void* packed, size_t packed_bytes,
                             void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dims || !W || !b || !packed) { gpde_set_error("gpde_mlp_pack: null argument"); return GPDE_EINVAL; }
    GpdePackLayout L;
    int rc = gpde_pack_layout(n_layers, dims, &L);
    if (rc != GPDE_OK) return rc;
    if (packed_bytes < L.total_floats * sizeof(float)) {
        gpde_set_error("gpde_mlp_pack: packed buffer %zu < %zu bytes", packed_bytes,
                       L.total_floats * sizeof(float));
        return GPDE_EWORKSPACE;
    }
    for (int l = 0; l < n_layers; ++l)
        if (!W[l]) { gpde_set_error("gpde_mlp_pack: W[%d] is null", l); return GPDE_EINVAL; }
    float* P = (float*)packed;
    const int T = 256;
    auto blocks = [&](size_t n) { return (unsigned)((n + T - 1) / T); };
    if (L.mode == 0) {
        hipLaunchKernelGGL(pack_w1_kernel, dim3(blocks((size_t)L.K2P * 8)), dim3(T), 0, stream, W[0],
                           b[0], dims[1], L.k0, L.K2P, P + L.off_w1);
    } else if (L.mode == 1) {
        hipLaunchKernelGGL(pack_w1_kernel, dim3(blocks((size_t)L.K1P * 8)), dim3(T), 0, stream, W[0],
                           b[0], dims[1], L.k0, L.K1P, P + L.off_w1);
        hipLaunchKernelGGL(pack_w2_kernel, dim3(blocks((size_t)L.K2P * L.K1P)), dim3(T), 0, stream,
                           W[1], dims[2], dims[1], L.K2P, L.K1P, P + L.off_w2t);
        hipLaunchKernelGGL(pack_pad_vec_kernel, dim3(blocks(L.K2P)), dim3(T), 0, stream, b[1],
                           dims[2], L.K2P, P + L.off_b2);
        hipLaunchKernelGGL(pack_w1max_kernel, dim3(1), dim3(256), 0, stream, P + L.off_w1, L.K1P,
                           P + L.off_w1 + (size_t)L.K1P * 8, P + L.off_fcol + 8);
        hipLaunchKernelGGL(pack_w1_f16split_kernel, dim3((L.K1P * 8 + 255) / 256), dim3(256), 0, stream,
                           P + L.off_w1, L.K1P, (_Float16*)(P + L.off_w1h), P + L.off_fcol);
        hipLaunchKernelGGL(pack_w2_f16split_kernel, dim3(L.K2P), dim3(256), 0, stream, W[1], dims[2],
                           dims[1], L.K2P, L.K1P, (_Float16*)(P + L.off_w2h), P + L.off_ucol);
        hipLaunchKernelGGL(pack_g2_consts_kernel, dim3((dims[2] + 3) / 4), dim3(256), 0, stream, W[1], b[1], dims[2], dims[1],
                           P + L.off_fcol + 8);
    } else {
        for (int l = 0; l < n_layers - 1; ++l) {
            hipLaunchKernelGGL(pack_pad_mat_kernel,
                               dim3(blocks((size_t)L.frontKP[l + 1] * L.frontKP[l])), dim3(T), 0,
                               stream, W[l], dims[l + 1], dims[l], L.frontKP[l + 1], L.frontKP[l],
                               P + L.off_front_w[l]);
            hipLaunchKernelGGL(pack_pad_vec_kernel, dim3(blocks(L.frontKP[l + 1])), dim3(T), 0,
                               stream, b[l], dims[l + 1], L.frontKP[l + 1], P + L.off_front_b[l]);
        }
    }
    hipLaunchKernelGGL(pack_w3_kernel, dim3(blocks((size_t)GP_W * L.K2P * GP_W)), dim3(T), 0, stream,
                       W[n_layers - 1], L.k2, L.K2P, P + L.off_w3q);
    hipLaunchKernelGGL(pack_pad_vec_kernel, dim3(blocks(GP_W * GP_W)), dim3(T), 0, stream,
                       b[n_layers - 1], GP_W * GP_W, GP_W * GP_W, P + L.off_b3);
    if (L.has_w3s) {
        rc = gpde_pack_split_nk(W[n_layers - 1], GP_W * GP_W, L.k2, GP_W * GP_W, L.K2P, P + L.off_w3s, P + L.off_ucol3, stream);
        if (rc != GPDE_OK) return rc;
    }
    GP_LAUNCH_CHECK("gpde_mlp_pack kernels");
    return GPDE_OK;
}
