// Fused edge kernel, f16-split, TWO WAVES PER SIMD (v3): 8 waves per workgroup, wave tile = 32 edges x
// 64 hidden columns, H1 generated on f16 MFMA as well.  Since round 2 the big-graph default is
// gpde_fused_f16v6_kernel (gpde_fused_f16v6.hip); this kernel serves graphs below 32,768 edges (fp32
// aggregation, no pre-pass launches), node-table attributes (NODEATTR) and gpde_hidden_fwd (WRITE_H).
//
// Replaces the hidden part of DenseNet.forward (/root/reference/graph-neural-operator/utilities.py:223-227),
// NNConv_old.message (nn_conv.py:273-275) and PyG's gather / scatter.  Why this shape: a wave alone on its SIMD hides
// only ~4 non-MFMA instructions per 32-cycle MFMA (scripts/ubench/mfma_valu_overlap.hip), so in the
// 4-wave kernel MFMA time and issue time ADD (47 % matrix-pipe occupancy).  Halving the wave tile to
// 64 columns brings the accumulators (32 + 64 registers) under the 256-register budget of two waves
// per SIMD.  The two waves of a pair (same 32 edges, column halves 0/1) both need H1, so its
// generation must be cheap: (W1|b1) . attr is done as 2 f16 MFMAs (K = 16 holds [hi|hi] x [hi;lo]
// and [lo|lo] x [hi;0]) instead of 4 fp32 ones, with per-input-slot column scales 2^u_d folded
// into the attributes (pack_w1_f16split_kernel).
//
// What the K loop looks like, and why (the story with numbers is DESIGN.md §3b):
//   * chunks of 32 k in PAIRS, one s_barrier per pair; pair G lives in ring slots {2(G&1), 2(G&1)+1},
//     the next pair's four 1-KiB DMA per wave are issued at the top of the iteration and retired by a
//     counted s_waitcnt before the closing barrier;
//   * per chunk: [MFMA, one conversion pair of the NEXT chunk's H1 (5 VALU, into the other operand
//     buffer)] x 12 in pinned source order, then the 2 H1 MFMAs of chunk + 2.  A long VALU burst of one
//     wave starves its SIMD partner's MFMA issue (same issue port, age priority): cross-wave overlap
//     alone hides nothing;
//   * W2 fragments are read half a chunk ahead, into the registers the previous MFMAs just released;
//   * s_setprio alternates per chunk between the two waves of a SIMD (w and w + 4) so that neither
//     runs ahead and then idles at the barrier;
//   * side loads (next tile's edge ids / attributes, this tile's x_j rows by DMA) at fixed iterations,
//     unconditional and clamped, so the wait counts are exact;
//   * after the loop: un-scale + bias + ReLU, then the aggregation Z += x_j (x) h_e per destination
//     segment (fp32 MFMA, or split f16 with global scales: G2F16), one plain-store flush per node.
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
// Integer max on the bit pattern: negative floats are negative ints -> 0, everything else passes
// (fmaxf costs two VALU: LLVM canonicalises an MFMA result before v_max_f32).
__device__ __forceinline__ float relu1(float v) { return __int_as_float(max(__float_as_int(v), 0)); }
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
constexpr int TILE_B = GP_TN * 128;        // 16 KiB per W2 chunk image
constexpr int XS_TILE = GP_TE * GP_W;      // floats per edge-tile x stage (shared by a wave pair)
constexpr int NW = 8;                      // waves per workgroup
constexpr int NET = 4;                     // edge tiles per workgroup

// WRITE_H: store the hidden activations [CSR slot][K2P] (a.hout) instead of aggregating them
// (gpde_hidden_fwd: the cross-depth cache of SURVEY.md §8 row f4); no x_j staging, no Z.
// G2F16: the aggregation Z += x_j (x) h_e on f16 MFMA with the same two-term operand split as the
// hidden layer (3 x v_mfma_f32_32x32x16_f16 instead of 8 x v_mfma_f32_32x32x2_f32 per 16 edges).  The
// contraction runs over edges, so the operands carry GLOBAL power-of-two scales: x comes pre-split
// from gpde_prep.hip (a.xs, scaled by max |x|), h is scaled by an a-priori bound.
// NODEATTR (opt-in): no [E][k0] attribute tensor; slot d of an edge's attribute is
// read from a node table, a.attr[(sel_d >> 8 ? dst_e : src_e) * a.kt + (sel_d & 255)] -- the reference
// builds edge_attr exactly so, [pos_src, pos_dst, a_src, a_dst] (graph-neural-operator/utilities.py:274-277).
template <bool WRITE_H, bool G2F16, bool NODEATTR>
__global__ __launch_bounds__(512, 2) void gpde_fused_f16v3_kernel(GpdeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                               // [4][16 KiB]
    char* w1s = smem + RING * TILE_B;                                // [K1P][hi 16 B | lo 16 B]
    float* Xs_all = (float*)(w1s + (size_t)a.K1P * 32);              // [4 edge tiles][32][64]
    int* red = (int*)(Xs_all + NET * XS_TILE);                       // [4]
    float* Es_all = (float*)(red + 4);                               // [8][32]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int et = wave >> 1, ch = wave & 1;                         // edge tile, column half
    const bool roleB = wave >= 4;                                    // SIMD partner of wave - 4
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Es = Es_all + wave * GP_TE;
    float* Xs = Xs_all + et * XS_TILE;

    const int ns = a.K2P / GP_TN;
    const int slice = blockIdx.x % ns;
    const int group = blockIdx.x / ns;
    const int NKC = a.K1P / GP_BK;

    for (int i = tid; i < a.K1P * 2; i += 512) ((f32x4*)w1s)[i] = ((const f32x4*)a.w1h)[i];

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
    if (lane == 0 && ch == 0) red[et] = ntiles;
    __syncthreads();
    const int maxtiles = max(max(red[0], red[1]), max(red[2], red[3]));
    if (maxtiles == 0) return;

    // ---- W2 chunk DMA: 2 x 1 KiB per wave per chunk ---------------------------------------------------
    // The address is rebuilt from a SCALAR base at every issue (one 64-bit add per DMA).  Left to
    // itself the compiler keeps four running per-lane pointers in VGPR pairs, spills them, and every
    // DMA of the steady-state loop then waits for a scratch load (~130 cycles each, 4 per iteration).
    const unsigned long long w2base = (unsigned long long)a.w2h + (size_t)slice * NKC * TILE_B + wave * 1024;
    const unsigned lane16 = lane * 16;
    auto issue_w2 = [&](int chunk, int slot) {
        unsigned long long gb = w2base + (size_t)chunk * TILE_B;
        asm volatile("" : "+s"(gb));
        char* l = ring + slot * TILE_B + wave * 1024;
        dma16((const char*)(gb + lane16), l);
        unsigned long long gb2 = gb + 8192;
        asm volatile("" : "+s"(gb2));
        dma16((const char*)(gb2 + lane16), l + 8192);
    };

    float b2v[2], ucv[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        b2v[nb] = a.b2[slice * GP_TN + ch * 64 + nb * 32 + l31];
        ucv[nb] = a.ucol[slice * GP_TN + ch * 64 + nb * 32 + l31];
    }
    float z_unscale = 1.f;
    if constexpr (G2F16) {
        // h_e[k] <= max|b2| + max_k ||W2_k||_1 * max_e B_e  (pack-time constants in fcol[8..9])
        const float sx = gpde_pow2_to_2p13(__uint_as_float(a.scal[0]));
        const float hb = a.fcol[8] + a.fcol[9] * __uint_as_float(a.scal[1]);
        const float sh = gpde_pow2_to_2p13(hb);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) { b2v[nb] *= sh; ucv[nb] *= sh; }       // exact: sh is a power of two
        z_unscale = (1.f / sx) * (1.f / sh);      // two exact reciprocals: sx * sh may exceed the float range
    }
    // per-input-slot constants: bound weights max_k|W1b[k][d]| and column un-scales 2^-u_d
    float wmx8[8], fcol8[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        wmx8[d] = a.w1[(size_t)a.K1P * 8 + (d & 1) * 4 + (d >> 1)];
        fcol8[d] = a.fcol[d];
    }
    const int sw = (l31 >> 1) & 7;
    const int rowb = (ch * 64 + l31) * 128;                       // byte offset of this lane's W2 row
    const int boff0 = rowb + (((0 + h) ^ sw) << 4);
    const int boff1 = rowb + (((2 + h) ^ sw) << 4);

    // ---- per-tile side loads (unconditional, clamped: exact VMEM op counts) ----------------------------
    // stage A (iteration 0):  edge id of the NEXT tile (1 load) + source nodes of THIS tile's rows
    //                         this wave stages (4 loads)
    // stage B (iteration K1): attributes of the NEXT tile (8 loads) + this tile's x_j rows, the
    //                         wave's half (4 DMA)
    const int e_clamp = max(e_hi - 1, 0);
    int perm_n = 0, dstn_n = 0, sidx[4];
    float attr_n[8];
    auto load_perm = [&](int e0n) {
        const int e = min(e0n + l31, e_clamp);
        if constexpr (NODEATTR) { perm_n = a.src[e]; dstn_n = a.dst[e]; }
        else perm_n = a.perm[e];
    };
    constexpr int A_CNT = (WRITE_H ? 0 : 4) + 1 + (NODEATTR ? 1 : 0);     // stage A VMEM ops after the W2 DMA
    constexpr int B_CNT = (WRITE_H ? 0 : 4) + 8;                          // stage B
    auto load_sidx = [&](int e0c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sidx[i] = a.src[min(e0c + (lane >> 4) + 4 * (ch * 4 + i), e_clamp)];
    };
    auto load_attr = [&]() {
        if constexpr (NODEATTR) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int sel = a.sel[min(d, a.k0 - 1)];
                attr_n[d] = a.attr[(size_t)((sel >> 8) ? dstn_n : perm_n) * a.kt + (sel & 255)];
            }
            return;
        }
        const float* ap = a.attr + (size_t)perm_n * a.k0;
#pragma unroll
        for (int d = 0; d < 8; ++d) attr_n[d] = ap[min(d, a.k0 - 1)];
    };
    auto issue_x = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16((G2F16 ? (const float*)a.xs : a.x) + (size_t)sidx[i] * GP_W + (lane & 15) * 4,
                  Xs + (ch * 4 + i) * 4 * GP_W);
    };
    const int NP = NKC / 2;                         // chunk pairs per tile (NKC is even)
    const int KP1 = NP >= 3 ? 1 : NP - 1;           // pair that issues stage B

    issue_w2(0, 0);
    issue_w2(1, 1);
    load_perm(ea);
    load_attr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 Z[WRITE_H ? 1 : 2][WRITE_H ? 1 : 2];
    if constexpr (!WRITE_H) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) Z[cb][nb][r] = 0.f;
    }
    int cur = -1;

    auto flush = [&](int node) {
        if constexpr (WRITE_H) return;
        float* zrow = a.zbuf + ((size_t)(node - a.nc0) * GP_W) * a.K2P + slice * GP_TN + ch * 64 + l31;
#pragma unroll
        for (int cb = 0; cb < (WRITE_H ? 1 : 2); ++cb)
#pragma unroll
            for (int nb = 0; nb < (WRITE_H ? 1 : 2); ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    zrow[(size_t)c * a.K2P + nb * 32] = G2F16 ? Z[cb][nb][r] * z_unscale : Z[cb][nb][r];
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

    [[maybe_unused]] float hmax_run = 0.f;      // WRITE_H: running max of the written activations (>= 0)
    int g = 0;
    for (int t = 0; t < maxtiles; ++t) {
        const int e0 = ea + t * GP_TE;
        const int e_end = min(e0 + GP_TE, eb);
        // One chunk pair per tile (K1P = 64): the x_j rows of this tile are issued in pair 0, with no barrier since
        // the previous tile's aggregation - the partner wave of the pair (same edges, other column half) may still be
        // reading the shared x stage.  (With two or more pairs the rows are issued behind pair 0's closing barrier.)
        if (!WRITE_H && NP == 1 && t > 0) __builtin_amdgcn_s_barrier();

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
        auto h1gen = [&](int chunk) {
            const char* wp = w1s + (size_t)(chunk * GP_BK + l31) * 32;
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

        h8 ahi[2][2], alo[2][2];      // two operand buffers: chunk parity (NKC is even: a tile starts at 0)
        f32x16 d;                     // raw H1 of the chunk after the current one (loop carried)
        {
            const f32x16 a0 = h1gen(0);
            d = h1gen(1);
#pragma unroll
            for (int p_ = 0; p_ < 8; ++p_) conv_to(a0, p_, ahi[0], alo[0]);
        }
        const int e0n = e0 + GP_TE;
        // destination of the tile's first / last edge (scalar loads issued now, used after the K
        // loop): a tile inside one destination node needs no further index loads
        const int n_first = WRITE_H ? 0 : a.dst[min(e0, e_clamp)];
        const int n_last = WRITE_H ? 0 : a.dst[min(max(e_end - 1, e0), e_clamp)];

        // K loop in PAIRS of chunks: one s_barrier per two chunks (the barrier is the most expensive
        // thing in the loop, scripts/ubench/kloop_model_v3.hip).  Pair G lives in ring slots
        // {2(G&1), 2(G&1)+1}; the next pair's 4 DMA are issued at the top of the iteration into the
        // other two slots (free since the previous barrier) and retired before the closing barrier.
        for (int kp = 0; kp < NP; ++kp, ++g) {
            const int sb = (g & 1) * 2;
            int cA = 2 * kp + 2, cB = 2 * kp + 3;
            if (cA >= NKC) cA -= NKC;
            if (cB >= NKC) cB -= NKC;
            issue_w2(cA, sb ^ 2);
            issue_w2(cB, (sb ^ 2) + 1);
            // The counted waits below assume that the four DMA are OLDER than the side loads.  Nothing else orders a
            // plain global load against an LDS-DMA: hipcc hoisted the next tile's edge-id load above the fourth DMA,
            // vmcnt(1) then let that DMA (the second piece of chunk cB) fly past the barrier, and after a cold-cache
            // stall its lo units were read stale (1e-4 errors in one tile, one in ~10 runs under memory pressure).
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kp == 0) {
                load_perm(e0n);
                if constexpr (!WRITE_H) load_sidx(e0);
            }
            if (kp == KP1) {
                load_attr();
                if constexpr (!WRITE_H) issue_x();
            }
            // B fragments (W2 hi / lo units of this lane's two rows) are read ONE HALF-CHUNK AHEAD, in
            // place: the lo units are consumed by the first two MFMAs of a half and re-loaded right
            // after them, the hi units after the last four.  Only the first half of a pair (the DMA
            // landed at the previous barrier) has its reads exposed.
            h8 bhi[2], blo[2];
            auto ld_lo = [&](const char* rbx, int m) {
                const int bo = m ? boff1 : boff0;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) blo[nb] = *(const h8*)(rbx + nb * 4096 + (bo ^ 64));
            };
            auto ld_hi = [&](const char* rbx, int m) {
                const int bo = m ? boff1 : boff0;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) bhi[nb] = *(const h8*)(rbx + nb * 4096 + bo);
            };
            ld_lo(ring + sb * TILE_B, 0);
            ld_hi(ring + sb * TILE_B, 0);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const char* rb = ring + (sb + cc) * TILE_B;
                // The SIMD arbitrates its two waves by age: alternate the priority per chunk so that
                // neither wave of a pair runs ahead and then idles at the barrier.
                if ((cc == 0) != roleB) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
                int c2 = 2 * kp + cc + 2;
                if (c2 >= NKC) c2 -= NKC;
                // (W1|b1) rows of chunk c + 2, read now, used by the two H1 MFMAs at the end of the chunk
                const char* wp = w1s + (size_t)(c2 * GP_BK + l31) * 32;
                const h8 A1 = *(const h8*)wp, A2 = *(const h8*)(wp + 16);
                // MFMA, one conversion pair of the NEXT chunk's H1 (5 VALU, into the other operand
                // buffer), MFMA, ... in pinned source order: a long VALU burst of one wave starves its
                // SIMD partner's MFMA issue (same issue port), so cross-wave overlap alone hides
                // nothing (scripts/v3_timing.py ablations).
#pragma unroll
                for (int m = 0; m < 2; ++m) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int nb = j & 1, t = j >> 1;       // t: 0 = hi x lo, 1 = hi x hi, 2 = lo x hi
                        acc1[nb] = mfma16(t == 2 ? alo[cc][m] : ahi[cc][m], t == 0 ? blo[nb] : bhi[nb], acc1[nb]);
                        if (j == 0 || j == 1 || j == 3 || j == 4) {
                            conv_to(d, 4 * m + (j < 2 ? j : j - 1), ahi[cc ^ 1], alo[cc ^ 1]);
                            asm volatile("" ::"v"(ahi[cc ^ 1][m]), "v"(alo[cc ^ 1][m]));
                        }
                        if (j == 1) {
                            if (m == 0) ld_lo(rb, 1);
                            else if (cc == 0) ld_lo(rb + TILE_B, 0);
                        }
                        if (j == 5) {
                            if (m == 0) ld_hi(rb, 1);
                            else if (cc == 0) ld_hi(rb + TILE_B, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                d = mfma16(A1, B1, d);
                d = mfma16(A2, B2, d);
                asm volatile("" ::"v"(d));
                __builtin_amdgcn_sched_barrier(0);
            }
            // counted wait: everything up to and including this iteration's 4 W2 DMA is retired; only
            // the side loads issued after them (5 at kp == 0, 12 at kp == KP1) may stay in flight
            if (kp == 0 && KP1 == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_CNT + B_CNT) : "memory");
            else if (kp == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_CNT) : "memory");
            else if (kp == KP1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_CNT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
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
                acc1[nb][r] = G2F16 ? fmaf(acc1[nb][r], ie * ucv[nb], b2v[nb])      // ReLU in the split below
                                    : relu1(fmaf(acc1[nb][r], ie * ucv[nb], b2v[nb]));
        }

        if constexpr (WRITE_H) {
            // hidden activations of this tile: row = CSR slot, 128-byte runs along the columns
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = e0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (e < eb) {
                    float* hp = a.hout + (size_t)(e - a.e_chunk0) * a.K2P + slice * GP_TN + ch * 64 + l31;
                    hp[0] = acc1[0][r];
                    hp[32] = acc1[1][r];
                    hmax_run = fmaxf(hmax_run, fmaxf(acc1[0][r], acc1[1][r]));
                }
            }
            continue;
        }
        // ---- GEMM2 with destination segments (fp32 MFMA) ----------------------------------------------
        // f16-split aggregation: B operands = the tile's h values, k slot (h, t) of MFMA m <-> edge
        // er(8m + t) + 4h, i.e. registers 8m .. 8m+7 of the accumulator, as they are
        h8 g2hi[2][2], g2lo[2][2];
        if constexpr (G2F16) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int p_ = 0; p_ < 8; ++p_) conv_to(acc1[nb], p_, g2hi[nb], g2lo[nb]);
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
            if constexpr (G2F16) {
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
                        for (int nb = 0; nb < (WRITE_H ? 0 : 2); ++nb) {
                            Z[WRITE_H ? 0 : cb][nb] = mfma16(xhi, g2hi[nb][m], Z[WRITE_H ? 0 : cb][nb]);
                            Z[WRITE_H ? 0 : cb][nb] = mfma16(xhi, g2lo[nb][m], Z[WRITE_H ? 0 : cb][nb]);
                            Z[WRITE_H ? 0 : cb][nb] = mfma16(xlo, g2hi[nb][m], Z[WRITE_H ? 0 : cb][nb]);
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
                for (int nb = 0; nb < (WRITE_H ? 0 : 2); ++nb) {
                    Z[0][nb] = mfma32(av0, acc1[nb][r], Z[0][nb]);
                    Z[WRITE_H ? 0 : 1][nb] = mfma32(av1, acc1[nb][r], Z[WRITE_H ? 0 : 1][nb]);
                }
            }
            e_seg = seg_end;
            if (e_seg < e_end) node = a.dst[e_seg];
        }
    }
    if (cur >= 0) flush(cur);
    if constexpr (WRITE_H) {
        if (a.hmax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) hmax_run = fmaxf(hmax_run, __shfl_xor(hmax_run, o));
            if (lane == 0 && hmax_run > 0.f) atomicMax(a.hmax_out, __float_as_uint(hmax_run));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

static size_t v3_lds_bytes(int K1P) {
    return (size_t)RING * TILE_B + (size_t)K1P * 32 + (size_t)NET * XS_TILE * 4 + 16 + NW * GP_TE * 4 + 64 + 64;
}


bool gpde_fused_f16v3_supported(const GpdeFusedArgs& a) {
    return a.K1P / GP_BK >= 2 && (a.K1P / GP_BK) % 2 == 0 && a.k0 + 1 <= 8 && v3_lds_bytes(a.K1P) <= 80 * 1024 * 2;
}

int gpde_launch_fused_f16v3(const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(512);
    const size_t lds = v3_lds_bytes(a.K1P);
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_fused_f16v3_kernel<false, false, false>, gpde_fused_f16v3_kernel<false, true, false>,
                             gpde_fused_f16v3_kernel<true, false, false>, gpde_fused_f16v3_kernel<false, false, true>,
                             gpde_fused_f16v3_kernel<false, true, true>)) return rc;
    if (a.hout) {
        if (a.kt) { gpde_set_error("hidden-activation output from node-table attributes is not built"); return GPDE_EUNSUPPORTED; }
        hipLaunchKernelGGL((gpde_fused_f16v3_kernel<true, false, false>), grid, block, lds, stream, a);
    } else if (a.kt) {
        if (a.xs) hipLaunchKernelGGL((gpde_fused_f16v3_kernel<false, true, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((gpde_fused_f16v3_kernel<false, false, true>), grid, block, lds, stream, a);
    } else if (a.xs) hipLaunchKernelGGL((gpde_fused_f16v3_kernel<false, true, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gpde_fused_f16v3_kernel<false, false, false>), grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_fused_f16v3_kernel");
    return GPDE_OK;
}
