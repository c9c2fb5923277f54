// Fused edge kernel, f16-split, v4: TWO INDEPENDENT 4-wave workgroups per CU (64-column slices).
//
// Same contract, math and arithmetic as gpde_fused_f16v3_kernel (gpde_fused_f16v3.hip; replaces
// DenseNet.forward hidden part, /root/reference/graph-neural-operator/utilities.py:223-227,
// NNConv_old.message, nn_conv.py:273-275, and PyG's gather/scatter).  v3 puts two waves on every
// SIMD but couples all 8 through one s_barrier, and the barrier turned out to be the most expensive
// item of the K loop (scripts/ubench/kloop_model_v3.hip).  Here the two waves of a SIMD belong to
// DIFFERENT workgroups: each workgroup is 4 waves (one per SIMD), wave tile 32 edges x 64 columns,
// slice width 64, with its own LDS ring (W2 half-tile 8 KiB + the next chunk's (W1|b1) rows 1 KiB
// per slot) and its own barrier, so a workgroup parked at its barrier leaves the matrix pipe to
// the other one.  70 KiB of LDS per workgroup -> two fit a CU.
#include "gpde_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// NOT inline asm: hipcc does not pad the MFMA-result -> reader hazard for an asm statement that
// reads a VGPR an MFMA has just written (seen as ~1e-5 errors when the accumulators live in VGPRs)
__device__ __forceinline__ float relu1(float v) { return fmaxf(v, 0.f); }
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

constexpr int RING = 4;                    // W2 chunk images in LDS: two pairs of chunks
constexpr int SLW = 64;                    // hidden columns per workgroup slice
constexpr int W2_B = SLW * 128;            // 8 KiB: W2 rows of the slice for one chunk
constexpr int TILE_B = W2_B + 1024;        // + the NEXT chunk's (W1|b1) rows [32][hi 16 B | lo 16 B]
constexpr int XS_TILE = GP_TE * GP_W;      // floats per wave x stage
constexpr int NW = 4;                      // waves per workgroup = edge tiles per workgroup
constexpr int NET = 4;

__global__ __launch_bounds__(256, 2) void gpde_fused_f16v4_kernel(GpdeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                               // [4][9 KiB]
    float* Xs_all = (float*)(smem + RING * TILE_B);                  // [4 waves][32][64]
    int* red = (int*)(Xs_all + NET * XS_TILE);                       // [4]
    float* Es_all = (float*)(red + 4);                               // [4][32]
    const char* w1c0 = (const char*)(Es_all + NW * GP_TE);           // (W1|b1) rows of chunk 0, resident

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int et = wave;                                             // one edge tile per wave
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Es = Es_all + wave * GP_TE;
    float* Xs = Xs_all + et * XS_TILE;

    const int ns = a.K2P / SLW;
    const int slice = blockIdx.x % ns;                               // 64-column slice
    const int group = blockIdx.x / ns;
    const int NKC = a.K1P / GP_BK;

    // ---- node-aligned edge range of this wave PAIR -------------------------------------------------
    const int e_lo = a.rowptr[a.nc0], e_hi = a.rowptr[a.nc1];
    const long tot = (long)e_hi - e_lo;
    const int nranges = a.n_groups * NET;
    const int wg = group * NET + et;
    const int na = lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * wg / nranges);
    const int nb_ = (wg == nranges - 1) ? a.nc1
                                        : lower_bound_node(a.rowptr, a.nc0, a.nc1,
                                                           e_lo + tot * (wg + 1) / nranges);
    const int ea = a.rowptr[na], eb = a.rowptr[nb_];
    const int ntiles = (eb - ea + GP_TE - 1) / GP_TE;
    if (lane == 0) red[et] = ntiles;
    __syncthreads();
    const int maxtiles = max(max(red[0], red[1]), max(red[2], red[3]));
    if (maxtiles == 0) return;

    // ---- chunk DMA into a ring slot: this slice's 64 W2 rows of `chunk` (8 x 1 KiB pieces, two per
    //      wave) and the (W1|b1) rows of chunk+1 (one 1 KiB piece, issued by wave chunk&3) --------------
    const char* w2g = (const char*)a.w2h + (size_t)(slice >> 1) * NKC * (GP_TN * 128) + (slice & 1) * W2_B +
                      wave * 1024 + lane * 16;
    const char* w1g = (const char*)a.w1h + lane * 16;
    auto issue_w2 = [&](int chunk, int slot) {
        const char* g = w2g + (size_t)chunk * (GP_TN * 128);
        char* l = ring + slot * TILE_B + wave * 1024;
        dma16(g, l);
        dma16(g + 4096, l + 4096);
        if (wave == (chunk & 3)) {
            const int cn = (chunk + 1 < NKC) ? chunk + 1 : 0;
            dma16(w1g + (size_t)cn * 1024, ring + slot * TILE_B + W2_B);
        }
    };

    float b2v[2], ucv[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        b2v[nb] = a.b2[slice * SLW + nb * 32 + l31];
        ucv[nb] = a.ucol[slice * SLW + nb * 32 + l31];
    }
    // per-input-slot constants: bound weights max_k|W1b[k][d]| and column un-scales 2^-u_d
    float wmx8[8], fcol8[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        wmx8[d] = a.w1[(size_t)a.K1P * 8 + (d & 1) * 4 + (d >> 1)];
        fcol8[d] = a.fcol[d];
    }
    const int sw = (l31 >> 1) & 7;
    const int rowb = l31 * 128;                                   // byte offset of this lane's W2 row
    const int boff0 = rowb + (((0 + h) ^ sw) << 4);
    const int boff1 = rowb + (((2 + h) ^ sw) << 4);

    // ---- per-tile side loads (unconditional, clamped: exact VMEM op counts) ----------------------------
    // stage A (pair 0):   edge id of the NEXT tile (1 load) + source nodes of THIS tile's rows (8)
    // stage B (pair KP1): attributes of the NEXT tile (8 loads) + this tile's x_j rows (8 DMA)
    const int e_clamp = max(e_hi - 1, 0);
    int perm_n = 0, sidx[8];
    float attr_n[8];
    auto load_perm = [&](int e0n) { perm_n = a.perm[min(e0n + l31, e_clamp)]; };
    auto load_sidx = [&](int e0c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sidx[i] = a.src[min(e0c + (lane >> 4) + 4 * i, e_clamp)];
    };
    auto load_attr = [&]() {
        const float* ap = a.attr + (size_t)perm_n * a.k0;
#pragma unroll
        for (int d = 0; d < 8; ++d) attr_n[d] = ap[min(d, a.k0 - 1)];
    };
    auto issue_x = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            dma16(a.x + (size_t)sidx[i] * GP_W + (lane & 15) * 4, Xs + i * 4 * GP_W);
    };
    const int NP = NKC / 2;                         // chunk pairs per tile (NKC is even)
    const int KP1 = NP >= 3 ? 1 : NP - 1;           // pair that issues stage B

    issue_w2(0, 0);
    issue_w2(1, 1);
    if (wave == 0) dma16(w1g, (void*)w1c0);
    load_perm(ea);
    load_attr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 Z[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[cb][nb][r] = 0.f;
    int cur = -1;

    auto flush = [&](int node) {
        float* zrow = a.zbuf + ((size_t)(node - a.nc0) * GP_W) * a.K2P + slice * SLW + l31;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    zrow[(size_t)c * a.K2P + nb * 32] = Z[cb][nb][r];
                    Z[cb][nb][r] = 0.f;
                }
    };

    // two-term split of a pair of non-negative values: hi = rtz16(y) (packed convert),
    // lo = rn16(y - hi) as ONE v_fma_mix per value (f32 y, f16 hi operand, f16 result): 5 VALU per
    // pair instead of 10.  The asm reads only VALU results (y from v_max), never an MFMA register.
    auto conv_to = [&](const f32x16& v, int p_, h8 (&hi)[2], h8 (&lo)[2]) {
        const int m = p_ >> 2, jp = p_ & 3;
        const float y0 = relu1(v[8 * m + 2 * jp]), y1 = relu1(v[8 * m + 2 * jp + 1]);
        const unsigned ph = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(y0, y1));
        unsigned pl;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(pl) : "v"(y0), "v"(ph));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(pl) : "v"(y1), "v"(ph));
        u4 hv = __builtin_bit_cast(u4, hi[m]), lv = __builtin_bit_cast(u4, lo[m]);
        hv[jp] = ph;
        lv[jp] = pl;
        hi[m] = __builtin_bit_cast(h8, hv);
        lo[m] = __builtin_bit_cast(h8, lv);
    };

    int g = 0;
    for (int t = 0; t < maxtiles; ++t) {
        const int e0 = ea + t * GP_TE;
        const int e_end = min(e0 + GP_TE, eb);

        // ---- attributes of this lane's edge: validity, bias slot, per-edge scale, f16 split --------
        h8 B1, B2;          // H1 MFMA operands: B1 = h ? attr_lo : attr_hi ; B2 = h ? 0 : attr_hi
        {
            const bool valid = (e0 + l31) < eb;
            float bnd = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float v = (valid && d < a.k0) ? attr_n[d] : 0.f;
                if (valid && d == a.k0) v = 1.f;
                attr_n[d] = v;
                bnd = fmaf(wmx8[d], fabsf(v), bnd);
            }
            const int ebits = (__float_as_int(bnd) >> 23) & 0xff;
            const bool okb = (ebits >= 20) && (ebits <= 230);
            const float sc = okb ? __int_as_float((267 - ebits) << 23) : 1.f;     // 2^(13 - E(B))
            const float isc = okb ? __int_as_float((ebits - 13) << 23) : 1.f;
            if (h == 0) Es[l31] = isc;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float s = attr_n[d] * fcol8[d] * sc;
                const _Float16 hi = (_Float16)s;
                const _Float16 lo = (_Float16)(s - (float)hi);
                B1[d] = h ? lo : hi;
                B2[d] = h ? (_Float16)0.f : hi;
            }
        }
        auto h1gen = [&](const char* w1rows) {
            const char* wp = w1rows + (size_t)l31 * 32;
            const h8 A1 = *(const h8*)wp, A2 = *(const h8*)(wp + 16);
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = mfma16(A1, B1, d);
            d = mfma16(A2, B2, d);
            return d;
        };

        f32x16 acc1[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nb][r] = 0.f;

        h8 ahi[2], alo[2];
        {
            const f32x16 a0 = h1gen(w1c0);
#pragma unroll
            for (int p_ = 0; p_ < 8; ++p_) conv_to(a0, p_, ahi, alo);
        }
        const int e0n = e0 + GP_TE;
        // destination of the tile's first / last edge (scalar loads issued now, used after the K
        // loop): a tile inside one destination node needs no further index loads
        const int n_first = a.dst[min(e0, e_clamp)];
        const int n_last = a.dst[min(max(e_end - 1, e0), e_clamp)];

        // K loop in PAIRS of chunks: one s_barrier per two chunks (the barrier is the most expensive
        // thing in the loop, scripts/ubench/kloop_model_v3.hip).  Pair G lives in ring slots
        // {2(G&1), 2(G&1)+1}; the next pair's 4 DMA are issued at the top of the iteration into the
        // other two slots (free since the previous barrier) and retired before the closing barrier.
        for (int kp = 0; kp < NP; ++kp, ++g) {
            const int sb = (g & 1) * 2;
            int cA = 2 * kp + 2, cB = 2 * kp + 3;
            if (cA >= NKC) cA -= NKC;
            if (cB >= NKC) cB -= NKC;
#ifndef GPDE_ABL_NOSTAGE
            issue_w2(cA, sb ^ 2);
            issue_w2(cB, (sb ^ 2) + 1);
#endif
            if (kp == 0) {
                load_perm(e0n);
                load_sidx(e0);
            }
            if (kp == KP1) {
                load_attr();
                issue_x();
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const char* rb = ring + (sb + cc) * TILE_B;
                // raw H1 of the NEXT chunk first (2 MFMAs): converted behind this chunk's MFMAs, in
                // place, as soon as the operand registers of each k-half are free.  The partner wave
                // on this SIMD runs its MFMAs under this wave's conversion VALU and vice versa.
                f32x16 d = h1gen(rb + W2_B);                 // this slot carries the next chunk's (W1|b1) rows
                h8 bhi[2], blo[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int bo = m ? boff1 : boff0;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        bhi[nb] = *(const h8*)(rb + nb * 4096 + bo);
                        blo[nb] = *(const h8*)(rb + nb * 4096 + (bo ^ 64));
                    }
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc1[nb] = mfma16(ahi[m], bhi[nb], acc1[nb]);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc1[nb] = mfma16(ahi[m], blo[nb], acc1[nb]);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc1[nb] = mfma16(alo[m], bhi[nb], acc1[nb]);
                    __builtin_amdgcn_sched_barrier(0);
#ifndef GPDE_ABL_NOCONV
#pragma unroll
                    for (int p_ = 0; p_ < 4; ++p_) conv_to(d, 4 * m + p_, ahi, alo);
#else
                    asm volatile("" ::"v"(d));
#endif
                    asm volatile("" ::"v"(ahi[m]), "v"(alo[m]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // counted wait: everything up to and including this iteration's 4 W2 DMA is retired; only
            // the side loads issued after them (9 at kp == 0, 16 at kp == KP1) may stay in flight
#ifndef GPDE_ABL_NOSTAGE
            if (kp == 0 && KP1 == 0) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
            else if (kp == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (kp == KP1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef GPDE_ABL_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NP < 3) {      // the x_j rows were issued in the last pair: land them before the aggregation
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }

        // ---- undo the row (edge) and column scales, bias, ReLU ---------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ie = Es[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                acc1[nb][r] = relu1(fmaf(acc1[nb][r], ie * ucv[nb], b2v[nb]));
        }

        // ---- GEMM2 with destination segments (fp32 MFMA) ----------------------------------------------
        int e_seg = e0;
#ifdef GPDE_ABL_NOGEMM2
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) asm volatile("" ::"v"(acc1[nb]));
        e_seg = e_end;
#endif
        int node = n_first;
        while (e_seg < e_end) {
            const int seg_end = (node == n_last) ? e_end : min(a.rowptr[node + 1], e_end);
            if (node != cur) {
                if (cur >= 0) flush(cur);
                cur = node;
            }
            const int lo = e_seg - e0 - 4 * h, hi = seg_end - e0 - 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int er = (r & 3) + 8 * (r >> 2);
                const bool m = (er >= lo) && (er < hi);
                const float* xp = Xs + (er + 4 * h) * GP_W + l31;
                float av0 = xp[0], av1 = xp[32];
                av0 = m ? av0 : 0.f;
                av1 = m ? av1 : 0.f;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    Z[0][nb] = mfma32(av0, acc1[nb][r], Z[0][nb]);
                    Z[1][nb] = mfma32(av1, acc1[nb][r], Z[1][nb]);
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

static size_t v4_lds_bytes(int K1P) {
    (void)K1P;
    return (size_t)RING * TILE_B + (size_t)NET * XS_TILE * 4 + 16 + NW * GP_TE * 4 + 1024 + 64;
}

bool gpde_fused_f16v4_supported(const GpdeFusedArgs& a) {
    return a.K1P / GP_BK >= 2 && (a.K1P / GP_BK) % 2 == 0 && a.k0 + 1 <= 8 && v4_lds_bytes(a.K1P) <= 80 * 1024;
}

int gpde_launch_fused_f16v4(const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / SLW;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = v4_lds_bytes(a.K1P);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        GP_HIP_CHECK(hipFuncSetAttribute((const void*)gpde_fused_f16v4_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(gpde_fused_f16v4_kernel, grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_fused_f16v4_kernel");
    return GPDE_OK;
}
