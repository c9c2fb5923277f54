// Operand pre-passes of the backward's one-pass kernel (gpde_fused_f16v6_kernel<2>, gpde_fused_f16v6.hip; round 5).
// Backward of NNConv_old.message + the scatter (/root/reference/graph-neural-operator/nn_conv.py:273-275) reached from
// loss.backward() (UAI1_full_resolution.py:266): per edge j -> i
//   dU_2[e][n] = (sum_c x_j[c] dZ_i[c][n]) [H_2[e][n] > 0]         dx_e[c] = sum_n H_2[e][n] dZ_i[c][n] + dS_i[c]
// with dZ_i[c][n] = sum_o W3[c*64+o][n] gT_i[o] per destination node.  The one-pass kernel takes dZ_i as TWO split-f16 images
// (the same numbers, laid out as the A operands of its two products) so that every fragment is one 16-byte unit:
//   img1 [node][n][hi: 64 c halves | lo: 64 c halves]                      P1: lane = column n, K = channel c
//   img2 [node][c][group of 32 n][hi: 32 halves | lo: 32 halves]           P2: lane = channel c, K = column n, the halves of a group
//                                                                           in the MFMA accumulator's row order (gpde_dz_img2_col)
// img2 is written IN PLACE over the fp32 dZ (the same 128 bytes per (c, group)), img1 into a second buffer of the same size.
#include "gpde_common.h"
#include "gpde_split.h"

namespace {

// One workgroup per node: max |dZ_i| over its [64][K2P] block -> s_i = 2^(13 - E(max)); then, per block of 64 columns, the
// [64 c][64 n] tile goes through LDS as packed (hi | lo << 16) words and comes out as the two images.
__global__ __launch_bounds__(256) void k_dz_images(float* __restrict__ dZ, int K2P, unsigned* __restrict__ img1, float* __restrict__ unscale) {
    __shared__ unsigned red[4];
    __shared__ unsigned tile[64 * 65];                         // [c][n] packed halves, rows padded to 65 words (bank spread)
    const int row_floats = GP_W * K2P;
    float* p = dZ + (size_t)blockIdx.x * row_floats;
    unsigned* o1 = img1 + (size_t)blockIdx.x * row_floats;     // K2P rows of 64 words
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned m = 0;
    for (int i = threadIdx.x * 4; i < row_floats; i += 1024) {
        const f32x4 v = *(const f32x4*)(p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    float sc, isc;
    gp_pow2_scale(__uint_as_float(m), sc, isc);
    if (threadIdx.x == 0) unscale[blockIdx.x] = isc;
    for (int jb = 0; jb < K2P / 64; ++jb) {
        __syncthreads();                                       // (the previous block's tile has been read)
        // ---- load [64 c][64 n]: thread -> (c = idx / 16, four columns) ----
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = threadIdx.x + 256 * k, c = idx >> 4, n4 = (idx & 15) * 4;
            const f32x4 v = *(const f32x4*)(p + (size_t)c * K2P + jb * 64 + n4);
            unsigned h01, l01, h23, l23;
            gp_split2(v[0], v[1], sc, h01, l01);
            gp_split2(v[2], v[3], sc, h23, l23);
            unsigned* t = tile + c * 65 + n4;
            t[0] = (h01 & 0xffffu) | (l01 << 16);
            t[1] = (h01 >> 16) | (l01 & 0xffff0000u);
            t[2] = (h23 & 0xffffu) | (l23 << 16);
            t[3] = (h23 >> 16) | (l23 & 0xffff0000u);
        }
        __syncthreads();                                       // (all loads of the block precede its in-place stores)
        // ---- img2, in place: row c, group gq: words 0..15 = hi halves at positions 2w, 2w + 1, words 16..31 = lo ----
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = threadIdx.x + 256 * k, c = idx >> 6, rem = idx & 63, gq = rem >> 5, w = rem & 31;
            const int plane = w >> 4, wp = w & 15;
            const unsigned v0 = tile[c * 65 + 32 * gq + gpde_dz_img2_col(2 * wp)];
            const unsigned v1 = tile[c * 65 + 32 * gq + gpde_dz_img2_col(2 * wp + 1)];
            const unsigned word = plane == 0 ? ((v0 & 0xffffu) | (v1 << 16)) : ((v0 >> 16) | (v1 & 0xffff0000u));
            ((unsigned*)(p + (size_t)c * K2P + jb * 64 + 32 * gq))[w] = word;
        }
        // ---- img1: row n, words 0..31 = hi halves of channels 2w, 2w + 1, words 32..63 = lo ----
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = threadIdx.x + 256 * k, n = idx >> 6, w = idx & 63;
            const int plane = w >> 5, wp = w & 31;
            const unsigned v0 = tile[(2 * wp) * 65 + n], v1 = tile[(2 * wp + 1) * 65 + n];
            const unsigned word = plane == 0 ? ((v0 & 0xffffu) | (v1 << 16)) : ((v0 >> 16) | (v1 & 0xffff0000u));
            o1[(size_t)(jb * 64 + n) * 64 + w] = word;
        }
    }
}

__global__ void k_row_scales_from_slices(const float* __restrict__ rowmax, int nparts, int rows, float* __restrict__ sc,
                                         float* __restrict__ isc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows) return;
    float m = 0.f;
    for (int s = 0; s < nparts; ++s) m = fmaxf(m, rowmax[(size_t)s * rows + e]);
    float a, b;
    gp_pow2_scale(m, a, b);
    sc[e] = a;
    isc[e] = b;
}

}  // namespace

int gpde_launch_dz_images(float* dZ, int nn, int K2P, void* img1, float* unscale, hipStream_t stream) {
    if (nn <= 0) return GPDE_OK;
    if (K2P % 64 != 0 || !dZ || !img1 || !unscale) { gpde_set_error("gpde_launch_dz_images: K2P = %d", K2P); return GPDE_EINVAL; }
    hipLaunchKernelGGL(k_dz_images, dim3(nn), dim3(256), 0, stream, dZ, K2P, (unsigned*)img1, unscale);
    GP_LAUNCH_CHECK("k_dz_images");
    return GPDE_OK;
}

int gpde_launch_row_scales_from_slices(const float* rowmax, int nparts, int rows, float* row_sc, float* row_isc, hipStream_t stream) {
    if (rows <= 0) return GPDE_OK;
    hipLaunchKernelGGL(k_row_scales_from_slices, dim3((rows + 255) / 256), dim3(256), 0, stream, rowmax, nparts, rows, row_sc, row_isc);
    GP_LAUNCH_CHECK("k_row_scales_from_slices");
    return GPDE_OK;
}
