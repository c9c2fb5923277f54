// Fused edge kernel, f16-split variant with LDS-DMA streaming (GPDE_FWD_F16SPLIT, 3-Linear MLPs).
//
// Same contract and math as gpde_fused_kernel<1> (gpde_fused.hip; replaces DenseNet.forward hidden
// part, /root/reference/graph-neural-operator/utilities.py:223-227, NNConv_old.message,
// nn_conv.py:273-275, and PyG's gather/scatter), with two differences:
//
//  1. The k1 x k2 hidden GEMM runs on v_mfma_f32_32x32x16_f16 with two-term operand splitting:
//     every fp32 operand y (pre-scaled by a power of two so that its row maximum is < 2^14) is
//     written y = hi + lo + O(2^-21 y), hi = rtz16(y), lo = rn16(y - hi), and
//     a.b ~= hi_a.hi_b + hi_a.lo_b + lo_a.hi_b accumulated in fp32 (hi.hi is exact).  Power-of-two
//     scales (per edge for H1: a bound from sum_d max_k|W1b[k][d]| |attr_e[d]|; per row for W2: its
//     max magnitude) are exact and undone after the K loop.  DESIGN.md §3b has the error budget.
//  2. No instruction of the K loop waits on memory it just asked for: W2 tiles (pre-swizzled
//     16 KiB images) stream global -> LDS by global_load_lds DMA into a 3-deep ring, two chunks
//     ahead, retired by a counted s_waitcnt vmcnt before one s_barrier per chunk; (W1|b1) lives
//     in LDS for the whole kernel; the next tile's indices / attributes / x_j rows are fetched
//     during the current tile's K loop (x_j rows also by DMA into a double-buffered stage).
//     With one wave per SIMD there is no other wave to hide a stall behind, so every load is
//     issued >= 1 chunk (~1k cycles) before its first use; the first B fragments of the next chunk
//     are read before the barrier that ends the current one (4-deep ring).
#include "gpde_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// NOT inline asm: hipcc does not pad the MFMA-result -> reader hazard for an asm statement that
// reads a VGPR an MFMA has just written (seen as ~1e-5 errors when the accumulators live in VGPRs)
__device__ __forceinline__ float relu1(float v) { return fmaxf(v, 0.f); }
// 16-byte-per-lane DMA: LDS destination = wave-uniform base + lane*16, global source per lane
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

constexpr int RING = 4;                    // W2 chunk images in LDS; DMA runs 3 chunks ahead
constexpr int TILE_B = GP_TN * 128;        // 16 KiB per W2 chunk image: [128 rows][8 x 16 B units]
constexpr int XS_WAVE = GP_TE * GP_W;      // floats per wave x-stage buffer

__global__ __launch_bounds__(256, 1) void gpde_fused_f16_kernel(GpdeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                               // [4][16 KiB]
    float* w1s = (float*)(smem + RING * TILE_B);                     // [(K1P+1)][2][4]
    float* Xs_all = w1s + (size_t)(a.K1P + 1) * 8;                   // [4 waves][32][64]
    int* red = (int*)(Xs_all + GP_WAVES * XS_WAVE);                  // [4]
    float* Es_all = (float*)(red + 4);                               // [4][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Es = Es_all + wave * GP_TE;
    float* Xs = Xs_all + wave * XS_WAVE;

    const int ns = a.K2P / GP_TN;
    const int slice = blockIdx.x % ns;
    const int group = blockIdx.x / ns;
    const int NKC = a.K1P / GP_BK;

    // (W1|b1) and the appended max row -> LDS, once
    for (int i = tid; i < (a.K1P + 1) * 2; i += 256) ((f32x4*)w1s)[i] = ((const f32x4*)a.w1)[i];

    // ---- this wave's node-aligned edge range (same partition as gpde_fused_kernel) --------------
    const int e_lo = a.rowptr[a.nc0], e_hi = a.rowptr[a.nc1];
    const long tot = (long)e_hi - e_lo;
    const int nwaves = a.n_groups * GP_WAVES;
    const int wg = group * GP_WAVES + wave;
    const int na = lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * wg / nwaves);
    const int nb_ = (wg == nwaves - 1) ? a.nc1
                                       : lower_bound_node(a.rowptr, a.nc0, a.nc1,
                                                          e_lo + tot * (wg + 1) / nwaves);
    const int ea = a.rowptr[na], eb = a.rowptr[nb_];
    const int ntiles = (eb - ea + GP_TE - 1) / GP_TE;
    if (lane == 0) red[wave] = ntiles;
    __syncthreads();
    const int maxtiles = max(max(red[0], red[1]), max(red[2], red[3]));
    if (maxtiles == 0) return;

    // ---- W2 chunk DMA: 4 x 1 KiB per wave per chunk --------------------------------------------------
    const char* w2g = (const char*)a.w2h + (size_t)slice * NKC * TILE_B + wave * 1024 + lane * 16;
    auto issue_w2 = [&](int chunk, int slot) {
        const char* g = w2g + (size_t)chunk * TILE_B;
        char* l = ring + slot * TILE_B + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(g + i * 4096, l + i * 4096);
    };

    // per-lane constants
    float b2v[4], ucv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        b2v[nb] = a.b2[slice * GP_TN + nb * 32 + l31];
        ucv[nb] = a.ucol[slice * GP_TN + nb * 32 + l31];
    }
    const f32x4 wmx = *(const f32x4*)&w1s[((size_t)a.K1P * 2 + h) * 4];
    const int sw = (l31 >> 1) & 7;                               // XOR swizzle of 16-byte units
    // B-fragment byte offsets inside a chunk image (row = nb*32 + l31): hi unit (m*2+h)^sw, lo = ^4
    const int boff0 = l31 * 128 + (((0 + h) ^ sw) << 4);
    const int boff1 = l31 * 128 + (((2 + h) ^ sw) << 4);

    // ---- per-tile side loads (all unconditional with clamped indices: exact VMEM op counts) -------
    // stage A (K-loop iteration 0):  CSR slot -> edge id of the NEXT tile (1 plain load) and the
    //                                source nodes of THIS tile's edges (8 plain loads)
    // stage B (K-loop iteration K1): attributes of the NEXT tile (4 plain loads) and THIS tile's
    //                                x_j rows by DMA (8), needed only after the K loop (GEMM2)
    const int e_clamp = max(e_hi - 1, 0);
    int perm_n = 0, sidx[GP_TE / 4];
    float attr_n[4];
    auto load_perm = [&](int e0n) { perm_n = a.perm[min(e0n + l31, e_clamp)]; };
    auto load_sidx = [&](int e0c) {
#pragma unroll
        for (int i = 0; i < GP_TE / 4; ++i) sidx[i] = a.src[min(e0c + (lane >> 4) + 4 * i, e_clamp)];
    };
    auto load_attr = [&]() {                      // raw values; validity applied at the consumer
        const float* ap = a.attr + (size_t)perm_n * a.k0;
#pragma unroll
        for (int s = 0; s < 4; ++s) attr_n[s] = ap[min(2 * s + h, a.k0 - 1)];
    };
    auto issue_x = [&]() {                        // rows of edges past the range are masked by GEMM2
#pragma unroll
        for (int i = 0; i < GP_TE / 4; ++i)
            dma16(a.x + (size_t)sidx[i] * GP_W + (lane & 15) * 4, Xs + i * 4 * GP_W);
    };
#ifndef GPDE_K1
#define GPDE_K1 8
#endif
    const int K1 = (NKC >= GPDE_K1 + 2) ? GPDE_K1 : NKC - 2;     // iteration of stage B (1 <= K1 <= NKC-2)

    // ---- kernel prologue: fill the ring three chunks deep, fetch tile 0's attributes ---------------
    issue_w2(0, 0);
    issue_w2(1 % NKC, 1);
    issue_w2(2 % NKC, 2);
    load_perm(ea);
    load_attr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 Z[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[cb][nb][r] = 0.f;
    int cur = -1;

    auto flush = [&](int node) {
        float* zrow = a.zbuf + ((size_t)(node - a.nc0) * GP_W) * a.K2P + slice * GP_TN + l31;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    zrow[(size_t)c * a.K2P + nb * 32] = Z[cb][nb][r];
                    Z[cb][nb][r] = 0.f;
                }
    };

    auto load_w1 = [&](int chunk) {
        return *(const f32x4*)&w1s[((size_t)(chunk * GP_BK + l31) * 2 + h) * 4];
    };
    auto conv_to = [&](const f32x16& v, int p_, h8 (&hi)[2], h8 (&lo)[2]) {
#ifdef GPDE_ABL_NOCONV
        if (p_ >= 0) { asm volatile("" ::"v"(v)); return; }
#endif
        const int m = p_ >> 2, jp = p_ & 3;
        const float y0 = relu1(v[8 * m + 2 * jp]), y1 = relu1(v[8 * m + 2 * jp + 1]);
        const auto pk = __builtin_amdgcn_cvt_pkrtz(y0, y1);
        const _Float16 p0 = (_Float16)pk[0], p1 = (_Float16)pk[1];
        hi[m][2 * jp] = p0;
        hi[m][2 * jp + 1] = p1;
        lo[m][2 * jp] = (_Float16)(y0 - (float)p0);
        lo[m][2 * jp + 1] = (_Float16)(y1 - (float)p1);
    };

    // B fragments of the chunk about to be consumed, m = 0 half (prefetched across the barrier)
    h8 b0hi[4], b0lo[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        b0hi[nb] = *(const h8*)(ring + nb * 4096 + boff0);
        b0lo[nb] = *(const h8*)(ring + nb * 4096 + (boff0 ^ 64));
    }

    int g = 0;        // running chunk counter: chunk g lives in ring slot g % 4
    for (int t = 0; t < maxtiles; ++t) {
        const int e0 = ea + t * GP_TE;
        const int e_end = min(e0 + GP_TE, eb);

        // ---- per-edge power-of-two scale (bound on max_k H1[e][k]), applied to the attributes ----
        float attrv[4];
        {
            const bool valid = (e0 + l31) < eb;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int d = 2 * s + h;
                float v = (valid && d < a.k0) ? attr_n[s] : 0.f;
                if (valid && d == a.k0) v = 1.f;      // bias slot
                attr_n[s] = v;
            }
            float part = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) part = fmaf(wmx[s], fabsf(attr_n[s]), part);
            const float bnd = part + __shfl_xor(part, 32);
            const int ebits = (__float_as_int(bnd) >> 23) & 0xff;
            const bool okb = (ebits >= 20) && (ebits <= 230);
            const float sc = okb ? __int_as_float((267 - ebits) << 23) : 1.f;     // 2^(13 - E(B))
            const float isc = okb ? __int_as_float((ebits - 13) << 23) : 1.f;     // 2^(E(B) - 13)
#pragma unroll
            for (int s = 0; s < 4; ++s) attrv[s] = attr_n[s] * sc;
            if (h == 0) Es[l31] = isc;
        }

        f32x16 acc1[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nb][r] = 0.f;

        // operands of chunk 0 and the raw H1 of chunk 1 (exposed once per tile: 8 fp32 MFMAs)
        h8 ahi[2], alo[2], ahi_n[2], alo_n[2];
        f32x16 a_raw;
        {
            const f32x4 w0 = load_w1(0), w1c = load_w1(1 % NKC);
            f32x16 a0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a_raw[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0 = mfma32(w0[s], attrv[s], a0);
                a_raw = mfma32(w1c[s], attrv[s], a_raw);
            }
#pragma unroll
            for (int p_ = 0; p_ < 8; ++p_) conv_to(a0, p_, ahi, alo);
#ifdef GPDE_ABL_NOCONV
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 8; ++j) { ahi[m][j] = (_Float16)a0[j]; alo[m][j] = (_Float16)a0[8 + j]; ahi_n[m][j] = ahi[m][j]; alo_n[m][j] = alo[m][j]; }
#endif
        }
        f32x4 w1f = load_w1(2 % NKC);               // (W1|b1) rows of chunk 2, used in iteration 0
        const int e0n = e0 + GP_TE;

        for (int kc = 0; kc < NKC; ++kc, ++g) {
            const char* rb = ring + (g % RING) * TILE_B;             // chunk being consumed
            const char* rbn = ring + ((g + 1) % RING) * TILE_B;      // next chunk (already complete)
            int c2 = kc + 2, c3 = kc + 3;
            while (c2 >= NKC) c2 -= NKC;
            while (c3 >= NKC) c3 -= NKC;
            // ---- R0: memory issue; nothing here waits --------------------------------------------------
#ifndef GPDE_ABL_NOSTAGE
            issue_w2(c3, (g + 3) % RING);
#endif
            if (kc == 0) {
                load_perm(e0n);
                load_sidx(e0);
            } else if (kc == K1) {
                load_attr();
                issue_x();
            }
            h8 b1hi[4], b1lo[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                b1hi[nb] = *(const h8*)(rb + nb * 4096 + boff1);
                b1lo[nb] = *(const h8*)(rb + nb * 4096 + (boff1 ^ 64));
            }
            const f32x4 w1f_n = load_w1(c3);                         // rows for the next iteration's H1
            __builtin_amdgcn_sched_barrier(0);

            // ---- 24 f16 MFMAs (6 groups of 4) + the 4 fp32 MFMAs of H1 chunk kc+2 + the 8 conversion
            //      pieces of chunk kc+1 (~70 VALU).  Everything is independent of everything else in
            //      this iteration; the sched_group_barrier pipeline below tells the machine scheduler
            //      to lay it out as (1 MFMA, 3 VALU) slots so the VALU hides under the matrix pipe.
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#ifndef GPDE_ABL_NOH1
#pragma unroll
            for (int s = 0; s < 4; ++s) d = mfma32(w1f[s], attrv[s], d);
#endif
#pragma unroll
            for (int p_ = 0; p_ < 8; ++p_) conv_to(a_raw, p_, ahi_n, alo_n);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[0], b0hi[nb], acc1[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[0], b0lo[nb], acc1[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(alo[0], b0hi[nb], acc1[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[1], b1hi[nb], acc1[nb]);
            // b0 registers are free now: prefetch the next chunk's m = 0 half across the barrier
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                b0hi[nb] = *(const h8*)(rbn + nb * 4096 + boff0);
                b0lo[nb] = *(const h8*)(rbn + nb * 4096 + (boff0 ^ 64));
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[1], b1lo[nb], acc1[nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(alo[1], b1hi[nb], acc1[nb]);
            // keep the conversion results (and the H1 chain) in THIS basic block: without a use here
            // LLVM sinks them behind the wait/branch below, where nothing overlaps them
            asm volatile("" ::"v"(ahi_n[0]), "v"(alo_n[0]), "v"(ahi_n[1]), "v"(alo_n[1]));
            asm volatile("" ::"a"(d));
            // pipeline description (masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ahi[m] = ahi_n[m];
                alo[m] = alo_n[m];
            }
#ifndef GPDE_ABL_NOH1
            a_raw = d;
#endif
            w1f = w1f_n;
            // counted wait: everything older than this iteration's own VMEM ops is retired, i.e.
            // the W2 chunk issued one iteration ago (chunk g+2); this iteration's 4 DMA (+ 9 index
            // loads at kc == 0, + 4 attribute loads and 8 x DMA at kc == K1) stay in flight
#ifndef GPDE_ABL_NOSTAGE
            if (kc == 0) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            else if (kc == K1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
#ifndef GPDE_ABL_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- undo the row (edge) and column scales, bias, ReLU ---------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ie = Es[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc1[nb][r] = relu1(fmaf(acc1[nb][r], ie * ucv[nb], b2v[nb]));
        }

        // ---- GEMM2 with destination segments (fp32 MFMA) ----------------------------------------------
        int e_seg = e0;
#ifdef GPDE_ABL_NOGEMM2
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) asm volatile("" ::"v"(acc1[nb]));
        e_seg = e_end;
#endif
        while (e_seg < e_end) {
            const int node = a.dst[e_seg];
            const int seg_end = min(a.rowptr[node + 1], e_end);
            if (node != cur) {
                if (cur >= 0) flush(cur);
                cur = node;
            }
            const int lo = e_seg - e0 - 4 * h, hi = seg_end - e0 - 4 * h;   // per half-wave
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
                    Z[0][nb] = mfma32(av0, acc1[nb][r], Z[0][nb]);
                    Z[1][nb] = mfma32(av1, acc1[nb][r], Z[1][nb]);
                }
            }
            e_seg = seg_end;
        }
    }
    if (cur >= 0) flush(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the run-ahead DMA before exit
}

}  // namespace

size_t gpde_fused_f16_lds_bytes(int K1P) {
    return (size_t)RING * TILE_B + (size_t)(K1P + 1) * 32 + (size_t)GP_WAVES * XS_WAVE * 4 + 16 +
           GP_WAVES * GP_TE * 4 + 64;
}

bool gpde_fused_f16_supported(const GpdeFusedArgs& a) {
    return a.K1P / GP_BK >= 3 && gpde_fused_f16_lds_bytes(a.K1P) <= 160 * 1024;
}

int gpde_launch_fused_f16(const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = gpde_fused_f16_lds_bytes(a.K1P);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        GP_HIP_CHECK(hipFuncSetAttribute((const void*)gpde_fused_f16_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(gpde_fused_f16_kernel, grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_fused_f16_kernel");
    return GPDE_OK;
}
