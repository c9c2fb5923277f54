// Aggregation kernel of the "hidden activations given" path (gpde_nnconv_fwd_hidden and the Z
// recompute of the backward):
//
//     Z_i[c][k] = sum_{e -> i} x_j(e)[c] * H_e[k]          H [CSR slot][K2P] fp32 streamed from HBM
//
// i.e. NNConv_old.message + PyG's gather / scatter (/root/reference/graph-neural-operator/
// nn_conv.py:271-275) for edges whose kernel-MLP hidden activations already exist.  Per edge the
// kernel moves 4*K2P bytes of H once and does 2*64*K2P FLOPs on v_mfma_f32_32x32x2_f32 (exact fp32):
// at k2 = 1024 that is 4 KiB and 131 kFLOP per edge, 32 FLOP/B -- the fp32 matrix pipe (157 TF/s ->
// 1.2 G-edges/s) binds slightly before HBM (8 TB/s -> 1.95 G-edges/s), so the job of the data path is
// simply never to leave the matrix pipe waiting:
//   * one workgroup = 4 waves (one per SIMD) on one 128-column slice; each wave owns a contiguous,
//     node-aligned edge range and walks it in tiles of 32 edges (as the fused kernels do);
//   * the H tile of the NEXT tile is fetched into registers while the current one is multiplied:
//     16 global_load_dwordx4 per lane (a lane owns 4 CONSECUTIVE hidden columns 4*l31 .. 4*l31+3 of the
//     slice, so one 16-byte load feeds the four column blocks and a Z row is flushed as float4);
//   * the x_j rows of the next tile go global -> LDS by DMA (double-buffered), their source indices
//     are read one tile earlier still; everything in flight is retired by one s_waitcnt at the top of
//     the next tile, a full tile (~3.5 us of MFMA) after issue;
//   * destination segments inside a tile by masking the A operand, one plain-store flush of the
//     64 x 128 accumulator per destination node: deterministic, no atomics.
//
// F16 variant (from 32768 edges on, when max |H| is known): the same products on
// v_mfma_f32_32x32x16_f16 with two-term split operands (3 MFMAs per 16 edges instead of 8 fp32 ones),
// global power-of-two scales from max |x| (x pre-split by gpde_prep.hip) and max |H| (recorded when H
// was built); H is converted tile by tile in registers.  The kernel is then bound by the H stream.
#include "gpde_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int lower_bound_node(const int32_t* __restrict__ rowptr, int lo, int hi,
                                                long target) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((long)rowptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int XS_TILE = GP_TE * GP_W;      // floats per x stage

template <bool F16>
__global__ __launch_bounds__(256, 1) void gpde_zagg_kernel(GpdeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [4 waves][2][32][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    float* Xs0 = smem + wave * 2 * XS_TILE;

    const int ns = a.K2P / GP_TN;
    const int slice = blockIdx.x % ns;
    const int group = blockIdx.x / ns;

    // ---- this wave's node-aligned edge range ----------------------------------------------------------
    const int e_lo = a.rowptr[a.nc0], e_hi = a.rowptr[a.nc1];
    const long tot = (long)e_hi - e_lo;
    const int nwaves = a.n_groups * GP_WAVES;
    const int wg = group * GP_WAVES + wave;
    const int na = lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * wg / nwaves);
    const int nb_ = (wg == nwaves - 1) ? a.nc1
                                       : lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * (wg + 1) / nwaves);
    const int ea = a.rowptr[na], eb = a.rowptr[nb_];
    const int ntiles = (eb - ea + GP_TE - 1) / GP_TE;
    if (ntiles == 0) return;
    const int e_clamp = max(e_hi - 1, 0);

    // H rows: lane (l31, h) reads hidden columns slice*128 + 4*l31 .. +3 of edge rows er + 4h
    const float* hbase = a.hbuf + (size_t)slice * GP_TN + 4 * l31;
    auto load_h = [&](int e0, f32x4 (&dst)[16]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = min(e0 + (r & 3) + 8 * (r >> 2) + 4 * h, e_clamp);
            dst[r] = *(const f32x4*)(hbase + (size_t)(e - a.e_chunk0) * a.K2P);
        }
    };
    int sidx[8];
    auto load_sidx = [&](int e0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sidx[i] = a.src[min(e0 + (lane >> 4) + 4 * i, e_clamp)];
    };
    auto issue_x = [&](float* Xs) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            dma16((F16 ? (const float*)a.xs : a.x) + (size_t)sidx[i] * GP_W + (lane & 15) * 4, Xs + i * 4 * GP_W);
    };

    f32x16 Z[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[cb][nb][r] = 0.f;
    int cur = -1;
    float sh = 1.f, z_unscale = 1.f;
    if constexpr (F16) {
        const float sx = gpde_pow2_to_2p13(__uint_as_float(a.scal[0]));
        sh = gpde_pow2_to_2p13(__uint_as_float(a.hmax[0]));
        z_unscale = (1.f / sx) * (1.f / sh);      // two exact reciprocals: sx * sh may exceed the float range
    }
    auto flush = [&](int node) {
        float* zrow = a.zbuf + ((size_t)(node - a.nc0) * GP_W) * a.K2P + slice * GP_TN + 4 * l31;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                f32x4 v = {Z[cb][0][r], Z[cb][1][r], Z[cb][2][r], Z[cb][3][r]};
                if constexpr (F16) v *= z_unscale;
                *(f32x4*)(zrow + (size_t)c * a.K2P) = v;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) Z[cb][nb][r] = 0.f;
            }
    };

    // ---- prologue: tile 0's H and x rows, tile 1's source indices ------------------------------------------
    f32x4 hn[16];
    load_sidx(ea);
    load_h(ea, hn);
    issue_x(Xs0);                 // waits for sidx (compiler), then 8 DMA
    load_sidx(ea + GP_TE);

    for (int t = 0; t < ntiles; ++t) {
        const int e0 = ea + t * GP_TE;
        const int e_end = min(e0 + GP_TE, eb);
        float* Xs = Xs0 + (t & 1) * XS_TILE;
        // everything issued a tile ago has had a whole tile of MFMA time to land
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 hc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) hc[r] = hn[r];
        const int n_first = a.dst[min(e0, e_clamp)];
        const int n_last = a.dst[min(max(e_end - 1, e0), e_clamp)];
        // prefetch: H and x rows of tile t + 1, source indices of tile t + 2
        load_h(e0 + GP_TE, hn);
        issue_x(Xs0 + ((t + 1) & 1) * XS_TILE);
        load_sidx(e0 + 2 * GP_TE);
        __builtin_amdgcn_sched_barrier(0);

        // F16: B operands of MFMA m, column q: edges er(8m + t) + 4h, t = 0..7 = registers hc[8m + t][q]
        h8 bhi[2][4], blo[2][4];
        if constexpr (F16) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u4 hv, lv;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        const float y0 = hc[8 * m + 2 * jp][q] * sh, y1 = hc[8 * m + 2 * jp + 1][q] * sh;
                        const unsigned ph = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(y0, y1));
                        unsigned pl;
                        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(pl) : "v"(y0), "v"(ph));
                        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(pl) : "v"(y1), "v"(ph));
                        hv[jp] = ph;
                        lv[jp] = pl;
                    }
                    bhi[m][q] = __builtin_bit_cast(h8, hv);
                    blo[m][q] = __builtin_bit_cast(h8, lv);
                }
        }
        int e_seg = e0;
        int node = n_first;
        while (e_seg < e_end) {
            const int seg_end = (node == n_last) ? e_end : min(a.rowptr[node + 1], e_end);
            if (node != cur) {
                if (cur >= 0) flush(cur);
                cur = node;
            }
            const int lo = e_seg - e0 - 4 * h, hi = seg_end - e0 - 4 * h;
            if constexpr (F16) {
                const unsigned* xu = (const unsigned*)Xs;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        unsigned w[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int er = ((8 * m + t) & 3) + 8 * ((8 * m + t) >> 2);
                            const unsigned v = xu[(er + 4 * h) * GP_W + cb * 32 + l31];
                            w[t] = (er >= lo && er < hi) ? v : 0u;
                        }
                        u4 ah, al;
#pragma unroll
                        for (int jp = 0; jp < 4; ++jp) {
                            ah[jp] = __builtin_amdgcn_perm(w[2 * jp + 1], w[2 * jp], 0x05040100u);
                            al[jp] = __builtin_amdgcn_perm(w[2 * jp + 1], w[2 * jp], 0x07060302u);
                        }
                        const h8 xhi = __builtin_bit_cast(h8, ah), xlo = __builtin_bit_cast(h8, al);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            Z[cb][q] = mfma16(xhi, bhi[m][q], Z[cb][q]);
                            Z[cb][q] = mfma16(xhi, blo[m][q], Z[cb][q]);
                            Z[cb][q] = mfma16(xlo, bhi[m][q], Z[cb][q]);
                        }
                    }
            } else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = (r & 3) + 8 * (r >> 2);
                const bool m = (er >= lo) && (er < hi);
                const float* xp = Xs + (er + 4 * h) * GP_W + l31;
                float av0 = xp[0], av1 = xp[32];
                av0 = m ? av0 : 0.f;
                av1 = m ? av1 : 0.f;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    Z[0][nb] = mfma32(av0, hc[r][nb], Z[0][nb]);
                    Z[1][nb] = mfma32(av1, hc[r][nb], Z[1][nb]);
                }
            }
            e_seg = seg_end;
            if (e_seg < e_end) node = a.dst[e_seg];
        }
    }
    if (cur >= 0) flush(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

int gpde_launch_zagg(const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = (size_t)GP_WAVES * 2 * XS_TILE * sizeof(float);
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_zagg_kernel<false>, gpde_zagg_kernel<true>)) return rc;
    if (a.xs && a.hmax) hipLaunchKernelGGL(gpde_zagg_kernel<true>, grid, block, lds, stream, a);
    else hipLaunchKernelGGL(gpde_zagg_kernel<false>, grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_zagg_kernel");
    return GPDE_OK;
}
