// Fused edge kernel, f16-split, ONE WAVE PER SIMD with a 512-register budget (v6, the default for
// graphs from 32,768 edges on): 4 waves per workgroup, wave tile = 64 edges x 128 hidden columns.
//
// Same contract and math as gpde_fused_f16v3_kernel<false, true, false> (gpde_fused_f16v3.hip; replaces
// the hidden part of DenseNet.forward, /root/reference/graph-neural-operator/utilities.py:223-227,
// NNConv_old.message, nn_conv.py:273-275, and PyG's gather / scatter): per edge
//   H1 = relu((W1|b1) . attr)  ->  H2 = relu(W2 . H1 + b2)  ->  Z_dst[c][k] += x_src[c] * H2[k]
// with the k1 x k2 layer and the aggregation on 2-term split f16 MFMA, fp32 accumulation.
//
// Why this shape (scripts/ubench/kloop_model_v6.hip, measured on MI355X: the K loop below runs at 85 %
// matrix-pipe occupancy = 1.73 PFLOP/s chip-wide, which is what a bare MFMA stream sustains under the
// chip's power limit; the 8-wave v3 kernel sits at 51 %):
//   * the v3 kernel's two waves per SIMD both convert the same H1 tile, share the SIMD's issue port and
//     lose ~40 % of the loop to conversions, barrier waits and staging.  One wave per SIMD with a wave
//     tile twice as tall halves conversions, W2 fragment reads, DMA pieces and barriers per MFMA
//     (2.9 non-MFMA instructions per MFMA, spread so that no gap holds more than 4);
//   * 512 registers: accumulators (128) and Z (128) live in AGPRs for the whole kernel, nothing spills
//     (v3: 256 registers, 50 spilled);
//   * MFMA order is pinned by an empty asm on the accumulator ("+a") after every MFMA plus
//     sched_barrier(0): hipcc otherwise sinks the MFMAs of a step below its conversions;
//   * H1 is produced by an asm MFMA with a VGPR destination (the builtin lands in AGPRs in a 512-register
//     kernel: one v_accvgpr_read per value) and converted by asm pairs (relu, rtz16 hi, rn16 lo) - hipcc
//     pads no hazards for asm operands, so the producer is placed >= 2 MFMAs before the first consumer;
//   * W2 chunk images (16 KiB, pre-swizzled) stream through a 3-slot LDS ring by LDS-DMA: chunk c + 2 is
//     issued during chunk c (4 pieces per wave, one address + immediate offsets), retired by the
//     s_waitcnt vmcnt(0) + s_barrier that ends chunk c; fragments are read half a chunk ahead, in place.
#include "gpde_common.h"
#include "gpde_split.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#define GPDE_GLDS(g, l, off)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),         \
                                     (__attribute__((address_space(3))) void*)(l), 16, off, 0)

__device__ __forceinline__ int lower_bound_node(const int32_t* __restrict__ rowptr, int lo, int hi, long target) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((long)rowptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#ifdef GPDE_V6_TIMING      // developer probe (scripts/v6_timing.py): cycles per phase summed over waves, wave-tiles
__device__ unsigned long long gpde_v6_tm[8];      // prologue, K loop, post, tiles, peeled chunks 0-5 (part of the K loop)
#define TM_MARK(acc) do { const long long tm1_ = clock64(); acc += tm1_ - tm0_; tm0_ = tm1_; } while (0)
#else
#define TM_MARK(acc) do { } while (0)
#endif

constexpr int NS = 3;                      // W2 ring slots
constexpr int TILE_B = GP_TN * 128;        // 16 KiB per W2 chunk image: [128 columns][hi 64 B | lo 64 B]
constexpr int TE = 64;                     // edges per wave tile (two MFMA row blocks)
constexpr int NW = 4;                      // waves per workgroup

// WRITE_H: the store variant (gpde_hidden_fwd, the backward's recompute of H_2, the per-edge path of gpde_api.hip): the
// hidden activations of the tile go to a.hout ([CSR slot][K2P] fp32) instead of into the aggregation; no x_j, no Z, static
// edge ranges (rows are independent).
// NODEATTR (gpde_nnconv_fwd_nodeattr): the edge attributes are not a tensor - slot d of edge
// (j -> i) is table[(sel[d] >> 8 ? i : j) * kt + (sel[d] & 255)], read from the node table (a.attr) with the tile's
// source / destination ids (the reference builds the [E, 6] tensor from node data, utilities.py:274-277).  No `perm`,
// no per-edge attribute traffic: the table (N x kt floats) lives in L2.
//
// MODE 2, the backward's ONE-PASS kernel (round 5; backward of NNConv_old.message + the scatter, nn_conv.py:273-275, reached from
// loss.backward(), UAI1_full_resolution.py:266).  The store variant wrote H_2 (4 KiB per edge) for gpde_edge_bwd3_kernel to read
// back; here the K loop runs with the MFMA operands SWAPPED (A = W2 fragment, B = H1 operand), so the accumulators hold H_2^T -
// lane = edge, registers = hidden columns, gpde_edge_bwd3_kernel's orientation - and the two per-edge products are formed from
// them in place:
//   P1  dU[e][n]  = (sum_c x_e[c] dZ_i[c][n]) [H_2[e][n] > 0]    A = dZ_i^T image (lane = n, K = c), B = x_e (lane = e): the result
//                                                               has the accumulators' layout, the ReLU mask is `y > 0` in place
//   P2  dx_e[c]   =  sum_n H_2[e][n] dZ_i[c][n]                  A = dZ_i image (lane = c, K = n in the accumulator's row order),
//                                                               B = the accumulator registers themselves (split in the lane)
// plus what gpde_edge_bwd3_kernel left for the next GEMMs (transposed dU, row maxima, per-tile column sums / maxima).  Every
// per-edge quantity (validity, destination node, scales, mask) is per lane; a tile spanning several destination nodes runs one
// pass per node with the other nodes' lanes masked.  The workgroup owns a 128-column slice, so dx_e comes out as K2P / 128
// partial rows (2 KiB per edge instead of the 8 KiB H_2 round trip), summed per source node by k_dx_reduce.  Wave tiles are
// 64-slot aligned and dealt round-robin over the waves of a slice (no node alignment needed: nothing is accumulated across
// edges), so that the waves of an XCD work on neighbouring tiles and a node's two 32 KiB image slices are fetched once.
// The dZ_i fragments are staged per 32-column block through the wave's 16 KiB x stage by LDS-DMA, after the x_j rows (DMA'd
// during the K loop as in the forward) have been converted into registers.
template <int MODE, bool NODEATTR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gpde_fused_f16v6_kernel(GpdeFusedArgs a) {
    constexpr bool WRITE_H = MODE == 1;       // store variant
    constexpr bool BWD = MODE == 2;           // backward one-pass
    extern __shared__ __attribute__((aligned(16))) char smem[];     // ONE LDS object (cdna guide, glds trap a)
    char* ring = smem;                                               // [3][16 KiB]
    char* w1s = smem + NS * TILE_B;                                  // [K1P][hi 16 B | lo 16 B]
    unsigned* Xs_all = (unsigned*)(w1s + (size_t)a.K1P * 32);        // [4 waves][64 rows][64 words]
    float* Es_all = (float*)(Xs_all + NW * TE * GP_W);               // [4][64]
    int* red = (int*)(Es_all + NW * TE);                             // [2][4]
    [[maybe_unused]] float* cst = (float*)(red + 16);                // BWD: [128] ucol * sh, [128] b2 * sh of this column slice

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Es = Es_all + wave * TE;
    unsigned* Xs = Xs_all + wave * TE * GP_W;

    const int ns = a.K2P / GP_TN;
    const int slice = blockIdx.x % ns;
    const int group = blockIdx.x / ns;
    const int NKC = a.K1P / GP_BK;

    for (int i = tid; i < a.K1P * 2; i += 256) ((f32x4*)w1s)[i] = ((const f32x4*)a.w1h)[i];
    [[maybe_unused]] float bw_ish = 1.f;        // BWD: 1 / (global power-of-two scale of the hidden activations)
    if constexpr (BWD) {
        // h <= max|b2| + max_k ||W2_k||_1 * max_e B_e (the forward's a-priori bound): ONE power of two for all H
        const float hb = a.fcol[8] + a.fcol[9] * __uint_as_float(a.scal[1]);
        const float sh = gpde_pow2_to_2p13(hb);
        bw_ish = 1.f / sh;
        if (tid < GP_TN) {
            cst[tid] = a.ucol[(blockIdx.x % (a.K2P / GP_TN)) * GP_TN + tid] * sh;             // exact: sh is a power of two
            cst[GP_TN + tid] = a.b2[(blockIdx.x % (a.K2P / GP_TN)) * GP_TN + tid] * sh;
        }
    }

    // ---- work assignment ---------------------------------------------------------------------------------
    // Queue mode (a.blk != nullptr, the default): the chunk's edges are cut into node-aligned BLOCKS of ~4096 edges
    // (a.blk[b] .. a.blk[b+1], gpde_prep.hip) and every wave draws its next block from ONE counter per column slice
    // (a.qctr[slice]; the workgroups of a slice share an XCD).  The 128 waves of a slice therefore work on
    // CONSECUTIVE blocks at any time - a window of a few hundred destination nodes whose sources (x_j rows) are the
    // same lattice band for all of them, so the gather hits in the XCD's L2 (4 MiB) instead of missing it on every
    // edge (measured: HBM-side fetch per launch 56 GB with contiguous per-wave ranges).  A node's in-edges lie in one
    // block and one wave walks them in CSR order: the sum order does not depend on which wave drew the block -
    // results stay bit-reproducible.  The queue also levels the load (the tail is one block, not one range).
    // Static mode (a.blk == nullptr): one contiguous node-aligned range per wave.
    const int e_lo = a.rowptr[a.nc0], e_hi = a.rowptr[a.nc1];
    const bool queue = a.blk != nullptr;
    int blk_a = 0, blk_b = 0;           // current block / range [blk_a, blk_b)
    const int nblk = queue ? *a.qn : 0;
    bool have;
    auto draw_block = [&](int& ba, int& bb) {        // wave-uniform; false when the queue is empty
        for (;;) {
            int b = 0;
            if (lane == 0) b = (int)atomicAdd(a.qctr + slice, 1u);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b >= nblk) return false;
            ba = a.blk[b];
            bb = a.blk[b + 1];
            if (bb > ba) return true;
        }
    };
    // BWD: 64-slot aligned tiles dealt round-robin over the waves of the slice (tile k of wave w = w + k * waves)
    [[maybe_unused]] const int bw_stride = TE * a.n_groups * NW;
    if (queue) {
        have = draw_block(blk_a, blk_b);
    } else if constexpr (BWD) {
        blk_a = e_lo + TE * (group * NW + wave);
        blk_b = min(blk_a + TE, e_hi);
        have = blk_a < e_hi;
    } else {
        const long tot = (long)e_hi - e_lo;
        const int nranges = a.n_groups * NW;
        const int wg = group * NW + wave;
        const int na = lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * wg / nranges);
        const int nb_ = (wg == nranges - 1) ? a.nc1
                                            : lower_bound_node(a.rowptr, a.nc0, a.nc1, e_lo + tot * (wg + 1) / nranges);
        blk_a = a.rowptr[na];
        blk_b = a.rowptr[nb_];
        have = blk_b > blk_a;
    }
    // red[parity][wave]: "this wave has no tile" flags of the current tile round (double-buffered by parity)
    if (lane == 0) { red[wave] = have ? 0 : 1; red[4 + wave] = 0; }
    __syncthreads();
    if (red[0] + red[1] + red[2] + red[3] == 4) return;

    // ---- W2 chunk DMA: 4 x 1 KiB per wave per chunk, one address, immediate offsets on both sides -----
    const unsigned long long w2base = (unsigned long long)a.w2h + (size_t)slice * NKC * TILE_B + wave * 4096;
    const unsigned lane16 = lane * 16;
    auto w2_src = [&](int chunk) {
        unsigned long long gb = w2base + (size_t)chunk * TILE_B;
        asm volatile("" : "+s"(gb));
        return (const char*)(gb + lane16);
    };

    // constants of the un-scaling: h <= max|b2| + max_k ||W2_k||_1 * max_e B_e (pack-time constants in fcol[8..9])
    float b2v[4] = {0.f, 0.f, 0.f, 0.f}, ucv[4] = {0.f, 0.f, 0.f, 0.f};
    float z_unscale = 1.f;
    if constexpr (BWD) {
        // (per-register constants: read from `cst` in the epilogue)
    } else if constexpr (WRITE_H) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            b2v[nb] = a.b2[slice * GP_TN + nb * 32 + l31];
            ucv[nb] = a.ucol[slice * GP_TN + nb * 32 + l31];
        }
    } else {
        const float sx = gpde_pow2_to_2p13(__uint_as_float(a.scal[0]));
        const float hb = a.fcol[8] + a.fcol[9] * __uint_as_float(a.scal[1]);
        const float sh = gpde_pow2_to_2p13(hb);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            b2v[nb] = a.b2[slice * GP_TN + nb * 32 + l31] * sh;          // exact: sh is a power of two
            ucv[nb] = a.ucol[slice * GP_TN + nb * 32 + l31] * sh;
        }
        z_unscale = (1.f / sx) * (1.f / sh);      // two exact reciprocals: sx * sh may exceed the float range
    }
    float wmx8[8], fcol8[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        wmx8[d] = a.w1[(size_t)a.K1P * 8 + (d & 1) * 4 + (d >> 1)];
        fcol8[d] = a.fcol[d];
    }
    const int sw = (l31 >> 1) & 7;
    const int boff[2] = {l31 * 128 + (((0 + h) ^ sw) << 4), l31 * 128 + (((2 + h) ^ sw) << 4)};

    // ---- per-tile side loads ----------------------------------------------------------------------------
    const int e_clamp = max(e_hi - 1, 0);
    // lane l owns edge (tile start + l): its edge id, source node and 8 attribute slots; the MFMA operand
    // layout needs edge 32 b + (l & 31) in BOTH lane halves - exchanged by v_permlane32_swap in the prologue
    int perm_n = 0;
    [[maybe_unused]] int dst_n = 0;     // NODEATTR: perm_n holds the SOURCE node of the next tile's edge, dst_n its destination
    int src_l = 0;                      // source node of edge (tile start + lane), for the x_j row DMA
    [[maybe_unused]] int dst_e = 0;     // BWD: destination node of edge (tile start + lane)
    float attr_n[8];
    auto load_perm = [&](int e0n) {
        if constexpr (NODEATTR) {
            perm_n = a.src[min(e0n + lane, e_clamp)];
            dst_n = a.dst[min(e0n + lane, e_clamp)];
        } else perm_n = a.perm[min(e0n + lane, e_clamp)];
    };
    auto load_attr_d = [&](int d) {             // one attribute slot (the K loop issues them one per MFMA gap)
        if constexpr (NODEATTR) {
            const int sd = a.sel[d];                                         // scalar (kernel argument)
            attr_n[d] = a.attr[(size_t)((sd >> 8) ? dst_n : perm_n) * a.kt + (sd & 255)];
        } else {
            attr_n[d] = a.attr[(size_t)perm_n * a.k0 + min(d, a.k0 - 1)];
        }
    };
    auto load_attr = [&]() {
#pragma unroll
        for (int d = 0; d < 8; ++d) load_attr_d(d);
    };
    // x_j rows of this tile: piece i = rows 4i .. 4i+3 (lane >> 4 picks the row, lane & 15 its 16-byte unit).
    // The source node comes from the lane that loaded that edge's src (ds_bpermute) at the START of a chunk,
    // the four DMA are issued at its END, behind the chunk's W2 pieces (see the K loop).
    int xsidx[4];
    auto x_addr1 = [&](int i0, int i) { xsidx[i] = __builtin_amdgcn_ds_bpermute((4 * (i0 + i) + (lane >> 4)) * 4, src_l); };
    auto x_issue1 = [&](int i0, int i) {
        int unit = lane & 15;
        // BWD: the rows are read back one row per lane (256 B apart): unit p of row r is stored at position p ^ (r & 15)
        if constexpr (BWD) unit ^= (4 * (i0 + i) + (lane >> 4)) & 15;
        GPDE_GLDS(a.xs + (size_t)xsidx[i] * GP_W + unit * 4, Xs + (i0 + i) * 4 * GP_W, 0);
    };

    // ---- conversions and H1 generation (asm: see the header) ----------------------------------------------
    auto conv_a = [&](float v0, float v1, unsigned& ph, unsigned& t0_, unsigned& t1_) {
        asm("v_max_i32 %1, 0, %3\n\t"
            "v_max_i32 %2, 0, %4\n\t"
            "v_cvt_pkrtz_f16_f32 %0, %1, %2"
            : "=&v"(ph), "=&v"(t0_), "=&v"(t1_) : "v"(v0), "v"(v1));
    };
    auto conv_b = [&](unsigned ph, unsigned t0_, unsigned t1_, unsigned& pl) {
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(pl) : "v"(t0_), "v"(t1_), "v"(ph));
    };
    auto conv_pair = [&](float v0, float v1, unsigned& ph, unsigned& pl) {      // both parts (outside the K loop)
        unsigned t0_, t1_;
        conv_a(v0, v1, ph, t0_, t1_);
        conv_b(ph, t0_, t1_, pl);
    };
    // the leading s_nop covers a VALU write of an operand right in front of the statement
    auto h1gen = [&](f32x16& dd, h8 a1, h8 a2, h8 b1, h8 b2) {
        asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(dd) : "v"(a1), "v"(b1));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(dd) : "v"(a2), "v"(b2));
    };

    // ---- prologue: chunks 0 and 1 of the W2 slice, the first tile's edges and attributes ---------------
    {
        const char* g0 = w2_src(0);
        char* l0 = ring + wave * 4096;
        GPDE_GLDS(g0, l0, 0); GPDE_GLDS(g0, l0, 1024); GPDE_GLDS(g0, l0, 2048); GPDE_GLDS(g0, l0, 3072);
        const char* g1 = w2_src(1);
        char* l1 = ring + TILE_B + wave * 4096;
        GPDE_GLDS(g1, l1, 0); GPDE_GLDS(g1, l1, 1024); GPDE_GLDS(g1, l1, 2048); GPDE_GLDS(g1, l1, 3072);
    }
    load_perm(have ? blk_a : e_hi);
    load_attr();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[2][4], Z[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[e][nb][r] = 0.f; if constexpr (!BWD) Z[e][nb][r] = 0.f; }
    [[maybe_unused]] f32x16 D2[2][2];       // BWD: dx accumulators [channel block][edge block] (Z is not used there)
    if constexpr (BWD) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) D2[cb][b][r] = 0.f;
    }
    int cur = -1;
    [[maybe_unused]] float hmax_run = 0.f;      // WRITE_H: running maximum of the stored activations (>= 0)

    // one plain-store flush per (node, slice): the row base is wave-uniform (scalar address arithmetic), the lane
    // part of the address is ONE tile-invariant offset - 32 per-row vector addresses would be hoisted out of the
    // tile loop and spilled
    const int z_loff = 4 * h * a.K2P + l31;
    auto flush = [&](int node) {
        float* zb = a.zbuf + ((size_t)(node - a.nc0) * GP_W) * a.K2P + slice * GP_TN;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+a"(Z[cb][nb]));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* rp = zb + (size_t)(cb * 32 + (r & 3) + 8 * (r >> 2)) * a.K2P;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    rp[z_loff + nb * 32] = Z[cb][nb][r] * z_unscale;
                    Z[cb][nb][r] = 0.f;
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // 16 values in flight, not 128
            }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+a"(Z[cb][nb]));
    };

    int slot = 0;           // ring slot of the current chunk

#ifdef GPDE_V6_TIMING
    long long tm_pro = 0, tm_loop = 0, tm_post = 0, tm_peel = 0, tm0_ = clock64();
#endif
    int e0 = have ? blk_a : e_hi;
    int n_rounds = 0;
    for (int t = 0;; ++t) {
        // ---- this tile, and where the NEXT one starts (its edge ids / attributes are prefetched during this one) ----
        const int eb = have ? blk_b : e0;                        // no tile: everything below is masked out
        const int e_end = min(e0 + TE, eb);
        const bool last_in_blk = have && (e0 + TE >= blk_b);
        int na_ = 0, nb_ = 0;
        bool have_n = have && !last_in_blk;
        if (last_in_blk && queue) have_n = draw_block(na_, nb_);      // drawn ONE tile ahead: no block is held in reserve
        if constexpr (BWD) {
            if (last_in_blk) { na_ = blk_a + bw_stride; nb_ = min(na_ + TE, e_hi); have_n = na_ < e_hi; }
        }
        const int e0n = !have ? e_hi : (last_in_blk ? (have_n ? na_ : e_hi) : e0 + TE);
        if (t > 0) {
            // all four waves out of tiles -> done.  Flags of this round were written before the barrier; the other
            // parity is rewritten two rounds later, behind at least one K loop of barriers
            if (lane == 0) red[(t & 1) * 4 + wave] = have ? 0 : 1;
            __syncthreads();
            const int* rf = red + (t & 1) * 4;
            if (rf[0] + rf[1] + rf[2] + rf[3] == 4) break;
        }
        ++n_rounds;

        // ---- attributes of this lane's two edges: validity, bias slot, per-edge scale, f16 split ------------
        h8 B1[2], B2[2];        // H1 MFMA operands: B1 = h ? attr_lo : attr_hi ; B2 = h ? 0 : attr_hi
        {
            float av[2][8];     // [edge block][slot]: attributes of edge 32 b + l31 in both lane halves
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(attr_n[d]), __float_as_uint(attr_n[d]), false, false);
                av[0][d] = __uint_as_float(r2[0]);      // [lower | lower]
                av[1][d] = __uint_as_float(r2[1]);      // [upper | upper]
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool valid = (e0 + 32 * b + l31) < eb;
                float v[8];
                float bnd = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float q = (valid && d < a.k0) ? av[b][d] : 0.f;
                    if (valid && d == a.k0) q = 1.f;
                    v[d] = q;
                    bnd = fmaf(wmx8[d], fabsf(q), bnd);
                }
                const int ebits = (__float_as_int(bnd) >> 23) & 0xff;
                const bool okb = (ebits >= 20) && (ebits <= 230);
                const float sc = okb ? __int_as_float((267 - ebits) << 23) : 1.f;     // 2^(13 - E(B))
                const float isc = okb ? __int_as_float((ebits - 13) << 23) : 1.f;
                if (h == 0) Es[32 * b + l31] = isc;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const float s_ = v[d] * fcol8[d] * sc;
                    const _Float16 hi = (_Float16)s_;
                    const _Float16 lo = (_Float16)(s_ - (float)hi);
                    B1[b][d] = h ? lo : hi;
                    B2[b][d] = h ? (_Float16)0.f : hi;
                }
            }
        }
        // W2 fragments of step (0, 0) (within the tile they are read half a chunk ahead, in place; not carried
        // across the aggregation phase: 32 registers)
        h8 bhi[4], blo[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            bhi[nb] = *(const h8*)(ring + slot * TILE_B + nb * 4096 + boff[0]);
            blo[nb] = *(const h8*)(ring + slot * TILE_B + nb * 4096 + (boff[0] ^ 64));
        }
        // ---- raw H1 of chunk 0 and the operands of step (0, 0) -------------------------------------------
        f32x16 d[2];
        u4 ahi[2][2], alo[2][2];          // [operand buffer = step parity][edge block]
        {
            const char* wp = w1s + (size_t)l31 * 32;
            const h8 A1 = *(const h8*)wp, A2 = *(const h8*)(wp + 16);
            h1gen(d[0], A1, A2, B1[0], B2[0]);
            h1gen(d[1], A1, A2, B1[1], B2[1]);
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");           // MFMA result -> VALU read (asm operands are not padded)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    unsigned ph, pl;
                    conv_pair(d[e][2 * jp], d[e][2 * jp + 1], ph, pl);
                    ahi[0][e][jp] = ph;
                    alo[0][e][jp] = pl;
                }
        }
        // destination of the first / last edge of the two 32-edge halves: loaded with the other side loads at the
        // end of chunk 0 (all lanes the same address), made scalar after the K loop
        int nf_v[2] = {0, 0}, nl_v[2] = {0, 0};
        unsigned cph = 0, ct0 = 0, ct1 = 0;
        h8 A1, A2;
        {
            const char* wp = w1s + (size_t)(GP_BK + l31) * 32;           // (W1|b1) rows of chunk 1
            A1 = *(const h8*)wp;
            A2 = *(const h8*)(wp + 16);
        }

        // One k1 chunk.  The first seven chunks of a tile are peeled (PH = compile-time phase) so that the side
        // loads are straight-line code: inside a runtime `if (c == ..)` hipcc copies the loaded registers into
        // their loop-carried homes and waits vmcnt(0) for them right behind the load (measured: one exposed
        // memory round trip per chunk, ~40 % of the tile).  Side loads are issued at the END of a chunk, BEHIND
        // its four W2 pieces, and the closing wait is vmcnt(<side loads of this chunk>): the W2 pieces are
        // retired, the side loads fly for a whole chunk and are retired by the next chunk's closing wait.
        //   PH 0: next tile's edge ids, this tile's source nodes, 4 segment-end nodes (6 loads)   -> vmcnt(6)
        //         (NODEATTR: next tile's source AND destination ids instead of the edge ids: 7 loads -> vmcnt(7))
        //   PH 1: nothing (the three loads land)                                                   -> vmcnt(0)
        //   PH 2: next tile's attributes (8 loads) + x_j rows 0..15 (4 DMA)                        -> vmcnt(12)
        //   PH 3/4/5: x_j rows 16..31 / 32..47 / 48..63 (4 DMA each)                               -> vmcnt(4)
        //   PH 6: steady state                                                                     -> vmcnt(0)
        TM_MARK(tm_pro);
        auto chunk = [&](auto ph_tag, int c) {
            constexpr int PH = decltype(ph_tag)::value;
            asm volatile("" : "+s"(c));         // opaque: keeps the peeled chunks' addresses from being hoisted out of
                                                // the tile loop (and spilled)
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            const int slot2 = slot1 + 1 == NS ? 0 : slot1 + 1;
            int c2 = c + 2;
            if (c2 >= NKC) c2 -= NKC;
            const char* gsrc = w2_src(c2);
            char* ldst = ring + slot2 * TILE_B + wave * 4096;
            const char* rb0 = ring + slot * TILE_B;
            const char* rb1 = ring + slot1 * TILE_B;
            int c1 = c + 2;                                               // (W1|b1) rows read in this chunk: chunk c + 2
            if (c1 >= NKC) c1 -= NKC;
            const char* w1n = w1s + (size_t)(c1 * GP_BK + l31) * 32;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int cbuf = m;                                       // operand buffer of this step
                const char* rn = (m == 0) ? rb0 : rb1;                    // where the NEXT step's fragments live
                const int bo = boff[m ^ 1];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int tt = j >> 1, e = j & 1;                 // tt: 0 = hi x lo, 1 = hi x hi, 2 = lo x hi
                        const int i = nb * 6 + j;
#ifdef GPDE_ABL_2MFMA      // ablation build ONLY (scripts/gpu/clock_evidence.sh; WRONG results): two of the three split products, to
                           // measure how time and clock answer to a third less matrix work per edge
                        if (tt != 2)
#endif
                        if constexpr (BWD)       // operands swapped: the accumulator holds the TRANSPOSED tile (rows = columns n, lane = edge)
                            acc[e][nb] = mfma16(tt == 0 ? blo[nb] : bhi[nb],
                                                __builtin_bit_cast(h8, tt == 2 ? alo[cbuf][e] : ahi[cbuf][e]), acc[e][nb]);
                        else
                        acc[e][nb] = mfma16(__builtin_bit_cast(h8, tt == 2 ? alo[cbuf][e] : ahi[cbuf][e]),
                                            tt == 0 ? blo[nb] : bhi[nb], acc[e][nb]);
                        asm volatile("" : "+a"(acc[e][nb]));
                        // conversion pair p of the NEXT step's operands: first part after MFMA 3p, second after
                        // 3p + 1; in step 1 (raw H1 of the next chunk issued at the end of step 0) one MFMA later
                        {
                            const int q = m == 0 ? i : i - 1;
                            if (q >= 0 && q % 3 == 0 && q / 3 < 8)
                                conv_a(d[(q / 3) >> 2][8 * (m ^ 1) + 2 * ((q / 3) & 3)],
                                       d[(q / 3) >> 2][8 * (m ^ 1) + 2 * ((q / 3) & 3) + 1], cph, ct0, ct1);
                            if (q >= 0 && q % 3 == 1 && q / 3 < 8) {
                                unsigned pl;
                                conv_b(cph, ct0, ct1, pl);
                                ahi[cbuf ^ 1][(q / 3) >> 2][(q / 3) & 3] = cph;
                                alo[cbuf ^ 1][(q / 3) >> 2][(q / 3) & 3] = pl;
                                asm volatile("" ::"v"(ahi[cbuf ^ 1][(q / 3) >> 2]), "v"(alo[cbuf ^ 1][(q / 3) >> 2]));
                            }
                        }
                        if (j == 1) blo[nb] = *(const h8*)(rn + nb * 4096 + (bo ^ 64));
                        if (j == 5) bhi[nb] = *(const h8*)(rn + nb * 4096 + bo);
                        if (m == 0) {
                            if (i == 2) GPDE_GLDS(gsrc, ldst, 0);
                            if (i == 8) GPDE_GLDS(gsrc, ldst, 1024);
                            if (i == 14) GPDE_GLDS(gsrc, ldst, 2048);
                            if (i == 20) GPDE_GLDS(gsrc, ldst, 3072);
                        }
                        // The tile's side loads, ONE PER MFMA GAP of step 1 - i.e. behind the chunk's four W2 pieces, so
                        // that the counted wait at the chunk end retires exactly those (round 3: issued as one block after
                        // the last MFMA they cost ~570 cycles in each of the six peeled chunks, the matrix pipe idling
                        // while ~60 address / load instructions went out).  The x_j row addresses (ds_bpermute of the
                        // source ids loaded two chunks earlier) go into the gaps of step 0.
                        if constexpr (BWD && PH == 0) {
                            if (m == 1) {
                                if (i == 2) load_perm(e0n);
                                if (i == 5) src_l = a.src[min(e0 + lane, e_clamp)];
                                if (i == 8) dst_e = a.dst[min(e0 + lane, e_clamp)];
                            }
                        } else if constexpr (MODE == 0 && PH == 0) {
                            if (m == 1) {
                                if (i == 2) load_perm(e0n);
                                if (i == 5) src_l = a.src[min(e0 + lane, e_clamp)];
                                if (i == 8) nf_v[0] = a.dst[min(e0, e_clamp)];
                                if (i == 11) nl_v[0] = a.dst[min(max(min(e0 + 32, eb) - 1, e0), e_clamp)];
                                if (i == 14) nf_v[1] = a.dst[min(e0 + 32, e_clamp)];
                                if (i == 17) nl_v[1] = a.dst[min(max(min(e0 + 64, eb) - 1, e0 + 32), e_clamp)];
                            }
                        } else if constexpr (WRITE_H && PH == 0) {
                            if (m == 1 && i == 2) load_perm(e0n);
                        }
                        if constexpr (PH == 2) {
                            if (m == 1 && (i & 1) == 1 && i < 16) load_attr_d(i >> 1);
                        }
                        if constexpr (!WRITE_H && PH >= 2 && PH <= 5) {
                            if (m == 0 && (i == 1 || i == 4 || i == 7 || i == 10)) x_addr1(4 * (PH - 2), (i - 1) / 3);
                            if constexpr (PH == 2) {
                                if (m == 1 && (i == 17 || i == 19 || i == 21 || i == 23)) x_issue1(0, (i - 17) / 2);
                            } else {
                                if (m == 1 && (i == 3 || i == 9 || i == 15 || i == 21)) x_issue1(4 * (PH - 2), (i - 3) / 6);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (m == 0) {
                    // raw H1 of chunk c + 1 (single-buffered: its last reader ran two MFMAs ago); at the last
                    // chunk this is chunk 0 with THIS tile's attributes - unused, the next tile starts afresh
                    h1gen(d[0], A1, A2, B1[0], B2[0]);
                    h1gen(d[1], A1, A2, B1[1], B2[1]);
                    A1 = *(const h8*)w1n;
                    A2 = *(const h8*)(w1n + 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // closing wait: the chunk's W2 pieces are retired, its side loads (issued behind them, above) stay in flight
            if constexpr (BWD && PH == 0 && NODEATTR) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // next tile's src + dst ids, this tile's
            else if constexpr (BWD && PH == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                // next tile's edge ids, this tile's src + dst
            else if constexpr (WRITE_H && PH == 0 && NODEATTR) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // source + destination ids
            else if constexpr (WRITE_H && PH == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if constexpr (WRITE_H && PH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (WRITE_H) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (PH == 0 && NODEATTR) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if constexpr (PH == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (PH == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (PH >= 3 && PH <= 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot1;
        };
        chunk(std::integral_constant<int, 0>{}, 0);
        chunk(std::integral_constant<int, 1>{}, 1);
        chunk(std::integral_constant<int, 2>{}, 2);
        chunk(std::integral_constant<int, 3>{}, 3);
        chunk(std::integral_constant<int, 4>{}, 4);
        chunk(std::integral_constant<int, 5>{}, 5);
#ifdef GPDE_V6_TIMING
        tm_peel += clock64() - tm0_;
#endif
        for (int c = 6; c < NKC; ++c) chunk(std::integral_constant<int, 6>{}, c);


        TM_MARK(tm_loop);
        if constexpr (BWD) {
          if (have) {       // (wave-uniform; a wave without a tile only keeps the workgroup's barriers company)
            char* S1 = (char*)Xs;                  // dZ_i^T fragments of one 32-column block: [32 n][hi 128 B | lo 128 B], units swizzled by row
            char* S2 = (char*)Xs + 8192;           // dZ_i fragments: [64 c][hi 64 B | lo 64 B], units swizzled by row
            const int erow0 = e0 - a.e_chunk0;     // chunk-local row of the tile's first slot (a multiple of 64)
            const bool with_du = a.bw_dU != nullptr;
            const bool with_by = a.bw_dUt != nullptr;          // the by-products for the split GEMMs (else: dU rows only)
            // ---- per-lane data of the two 32-edge blocks: lane (l31, h) <-> edge e0 + 32 b + l31 in both halves ----
            int nodeL[2];
            bool vL[2];
            float unL[2], isx[2], ie[2];
            u4 xhi[2][4], xlo[2][4];               // P1's B operand: step s <-> channels 16 s + 8 h + 0..7, scaled by the row's 2^t
            {
                const auto nd2 = __builtin_amdgcn_permlane32_swap((unsigned)dst_e, (unsigned)dst_e, false, false);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    vL[b] = (e0 + 32 * b + l31) < eb;
                    nodeL[b] = vL[b] ? (int)nd2[b] : -1;
                    unL[b] = vL[b] ? a.bw_unscale[nodeL[b] - a.nc0] : 0.f;
                    ie[b] = Es[32 * b + l31];
                    const int row = 32 * b + l31;
                    const char* xrow = (const char*)Xs + row * 256;
                    f32x4 xr[4][2];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int u = 0; u < 2; ++u) xr[s][u] = *(const f32x4*)(xrow + (((4 * s + 2 * h + u) ^ (row & 15)) << 4));
                    float m = 0.f;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int t2 = 0; t2 < 4; ++t2) m = fmaxf(m, fabsf(xr[s][u][t2]));
                    m = fmaxf(m, gp_other_half(m));
                    float sx;
                    gp_pow2_scale(m, sx, isx[b]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            unsigned p0, q0, p1, q1;
                            gp_split2(xr[s][u][0], xr[s][u][1], sx, p0, q0);
                            gp_split2(xr[s][u][2], xr[s][u][3], sx, p1, q1);
                            xhi[b][s][2 * u] = p0; xhi[b][s][2 * u + 1] = p1;
                            xlo[b][s][2 * u] = q0; xlo[b][s][2 * u + 1] = q1;
                        }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the x stage has been read: it becomes S1 | S2
            __builtin_amdgcn_sched_barrier(0);
            // destination range of the tile (CSR order: ascending with the lane)
            const int nFirst = __builtin_amdgcn_readfirstlane(nodeL[0]);
            const int lastIdx = __builtin_amdgcn_readfirstlane(eb - e0 - 1);
            const int nLast = lastIdx < 32 ? __builtin_amdgcn_readlane(nodeL[0], lastIdx) : __builtin_amdgcn_readlane(nodeL[1], lastIdx - 32);
            // Fragment DMA of one 32-column block: 8 pieces of 1 KiB into `buf` (8 KiB).  dZ^T rows (P1): row r = 4 i + (lane >> 4),
            // position p = lane & 15 holds unit p ^ (r & 15); dZ rows (P2): row c = 8 i + (lane >> 3), position p = lane & 7 holds
            // unit p ^ (c & 7).  Each product walks the node's four blocks with the two halves of the stage as a double buffer.
            // Addresses = a SCALAR base rebuilt at every issue (opaque to the optimiser: per-lane 64-bit addresses of all 64 pieces
            // would be hoisted out of the node loop and spilled) + one of a few 32-bit lane offsets.
            unsigned s1_off[4];        // piece i uses s1_off[i & 3]: (4 i + (lane >> 4)) & 15 = 4 (i & 3) + (lane >> 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) s1_off[i] = (unsigned)((lane >> 4) * 256 + (((lane & 15) ^ (4 * i + (lane >> 4))) << 4));
            const unsigned s2_off = (unsigned)((lane >> 3) * a.K2P * 4 + (((lane & 7) ^ (lane >> 3)) << 4));
            auto issue_S1 = [&](const char* g1, int nb, char* buf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned long long gb = (unsigned long long)g1 + (size_t)(nb * 32 + 4 * i) * 256;
                    asm volatile("" : "+s"(gb));
                    GPDE_GLDS((const char*)gb + s1_off[i & 3], buf + i * 1024, 0);
                }
            };
            auto issue_S2 = [&](const char* g2, int nb, char* buf) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned long long gb = (unsigned long long)g2 + ((size_t)(8 * i) * a.K2P + nb * 32) * 4;
                    asm volatile("" : "+s"(gb));
                    GPDE_GLDS((const char*)gb + s2_off, buf + i * 1024, 0);
                }
            };
            // y = sh * (pre-activation of the H_2^T block [32 n][64 e] of column block nb); y > 0 is the ReLU mask
            auto y_block = [&](auto nb_tag, float (&y)[2][16]) {
                constexpr int nb = decltype(nb_tag)::value;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 uc4 = *(const f32x4*)&cst[nb * 32 + 8 * g + 4 * h];
                    const f32x4 b24 = *(const f32x4*)&cst[GP_TN + nb * 32 + 8 * g + 4 * h];
#pragma unroll
                    for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                        for (int b = 0; b < 2; ++b) y[b][4 * g + t2] = fmaf(acc[b][nb][4 * g + t2], ie[b] * uc4[t2], b24[t2]);
                }
            };
            float rmax[2] = {0.f, 0.f};            // max |dU| of this lane's rows (its half of the slice's columns)
            bool seen[2] = {false, false};         // block b's tile partials have been written by an earlier node pass
            const int sw7 = l31 & 7, sw15 = l31 & 15;
            for (int node = nFirst; node <= nLast; ++node) {
                const bool inN[2] = {nodeL[0] == node, nodeL[1] == node};
                const bool anyb[2] = {__builtin_amdgcn_ballot_w64(inN[0]) != 0, __builtin_amdgcn_ballot_w64(inN[1]) != 0};
                if (!anyb[0] && !anyb[1]) continue;
                const char* g1 = (const char*)a.bw_img1 + ((size_t)(node - a.nc0) * a.K2P + slice * GP_TN) * 256;
                const char* g2 = (const char*)a.bw_img2 + ((size_t)(node - a.nc0) * GP_W * a.K2P + slice * GP_TN) * 4;
                // ================= P1 and the dU outputs =================================================================
#ifdef GPDE_BWABL
                if (with_du && !(GPDE_BWABL & 8)) {
#else
                if (with_du) {
#endif
                    issue_S1(g1, 0, S1);
                    auto p1_block = [&](auto nb_tag) {
                        constexpr int nb = decltype(nb_tag)::value;
                        char* buf = (nb & 1) ? S2 : S1;
                        float y[2][16];
                        y_block(nb_tag, y);
                        // block nb's fragments have landed: the first block waits here, the others were waited for between the
                        // previous block's output arithmetic and its stores (so that no wait covers a block's own stores)
                        if (nb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        if (nb + 1 < 4) issue_S1(g1, nb + 1, (nb & 1) ? S1 : S2);      // (the other half: read one block ago)
                        __builtin_amdgcn_sched_barrier(0);
                        // ---- D1[n][e] = sum_c dZ[c][n] x_e[c] ----
                        f32x16 d1[2];
                        const char* r1 = buf + l31 * 256;
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r) d1[b][r] = 0.f;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const h8 Ah = *(const h8*)(r1 + (((2 * s + h) ^ sw15) << 4));
                            const h8 Al = *(const h8*)(r1 + (((8 + 2 * s + h) ^ sw15) << 4));
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                if (!anyb[b]) continue;
                                const h8 Bh = __builtin_bit_cast(h8, xhi[b][s]), Bl = __builtin_bit_cast(h8, xlo[b][s]);
                                d1[b] = mfma16(Ah, Bh, d1[b]);
                                d1[b] = mfma16(Ah, Bl, d1[b]);
                                d1[b] = mfma16(Al, Bh, d1[b]);
                            }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the fragments have been read
                        __builtin_amdgcn_sched_barrier(0);
                        // ---- dU of the block: rows, transposed copy, row maxima, per-tile column sums / maxima ----
                        const int nc = slice * GP_TN + nb * 32;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            float o[16], cs[16], cm[16];
                            if (anyb[b]) {
                                const float un = isx[b] * unL[b];
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    o[r] = (inN[b] && y[b][r] > 0.f) ? d1[b][r] * un : 0.f;
                                    rmax[b] = fmaxf(rmax[b], fabsf(o[r]));
                                }
#ifdef GPDE_BWABL
                                if (!(GPDE_BWABL & 1))
#endif
                                if (with_by) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) gp_half_wave_sum_max(o[r], cs[r], cm[r]);
                                }
                            }
                            // the next block's fragments (issued before this block's MFMAs) have landed; the stores follow the wait
                            if (b == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                            if (!anyb[b]) continue;
                            const int er = erow0 + 32 * b + l31;
                            if (inN[b]) {
                                float* du = a.bw_dU + (size_t)er * a.K2P + nc + 4 * h;
#ifdef GPDE_BWABL
                                if (!(GPDE_BWABL & 4))
#endif
#pragma unroll
                                for (int g = 0; g < 4; ++g) *(f32x4*)(du + 8 * g) = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                                // transposed copy: row nc + 8 g + 4 h + t, column er: scalar row base + one 32-bit lane offset
                                const unsigned dt_off = (unsigned)(4 * h * a.bw_ldt + er) * 4u;
#ifdef GPDE_BWABL
                                if (!(GPDE_BWABL & 2))
#endif
                                if (with_by)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
#pragma unroll
                                    for (int t2 = 0; t2 < 4; ++t2) {
                                        unsigned long long tb = (unsigned long long)a.bw_dUt + (size_t)(nc + 8 * g + t2) * a.bw_ldt * 4;
                                        asm volatile("" : "+s"(tb));
                                        *(float*)((char*)tb + dt_off) = o[4 * g + t2];
                                    }
                            }
#ifdef GPDE_BWABL
                            if (!(GPDE_BWABL & 1))
#endif
                            if (with_by && l31 == 31) {
                                const size_t po = (size_t)((erow0 >> 5) + b) * a.K2P + nc + 4 * h;
                                float* ps = a.bw_csum + po;
                                unsigned* pm = a.bw_cmax + po;
                                const bool first = !seen[b];
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    f32x4 sv4 = {cs[4 * g], cs[4 * g + 1], cs[4 * g + 2], cs[4 * g + 3]};
                                    u4 mv4 = {__float_as_uint(cm[4 * g]), __float_as_uint(cm[4 * g + 1]), __float_as_uint(cm[4 * g + 2]), __float_as_uint(cm[4 * g + 3])};
                                    if (!first) {
                                        const f32x4 ps0 = *(const f32x4*)(ps + 8 * g);
                                        const u4 pm0 = *(const u4*)(pm + 8 * g);
#pragma unroll
                                        for (int t2 = 0; t2 < 4; ++t2) { sv4[t2] += ps0[t2]; mv4[t2] = max(mv4[t2], pm0[t2]); }
                                    }
                                    *(f32x4*)(ps + 8 * g) = sv4;
                                    *(u4*)(pm + 8 * g) = mv4;
                                }
                            }
                        }
                    };
                    p1_block(std::integral_constant<int, 0>{});
                    p1_block(std::integral_constant<int, 1>{});
                    p1_block(std::integral_constant<int, 2>{});
                    p1_block(std::integral_constant<int, 3>{});
                }
                // ================= P2: D2[c][e] += sum_n dZ[c][n] H[e][n] ====================================================
                // B = the lane's own accumulator rows (split in the lane, zero outside this node), A = the node's dZ rows
#ifdef GPDE_BWABL
                if (!(GPDE_BWABL & 16)) {
#endif
                issue_S2(g2, 0, S1);
                auto p2_block = [&](auto nb_tag) {
                    constexpr int nb = decltype(nb_tag)::value;
                    const char* buf = (nb & 1) ? S2 : S1;
                    float y[2][16];
                    y_block(nb_tag, y);
                    u4 yh[2][2], yl[2][2];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int jp = 0; jp < 8; ++jp) {
                            unsigned ph, pl;
                            conv_pair(y[b][2 * jp], y[b][2 * jp + 1], ph, pl);       // relu, rtz16 hi, rn16 lo
                            yh[b][jp >> 2][jp & 3] = inN[b] ? ph : 0u;
                            yl[b][jp >> 2][jp & 3] = inN[b] ? pl : 0u;
                        }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // block nb's fragments (issued a block ago) have landed
                    __builtin_amdgcn_sched_barrier(0);
                    if (nb + 1 < 4) issue_S2(g2, nb + 1, (nb & 1) ? S1 : S2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
                            const char* row = buf + (cb * 32 + l31) * 128;
                            const h8 Ah = *(const h8*)(row + (((2 * q + h) ^ sw7) << 4));
                            const h8 Al = *(const h8*)(row + (((4 + 2 * q + h) ^ sw7) << 4));
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                if (!anyb[b]) continue;
                                const h8 Bh = __builtin_bit_cast(h8, yh[b][q]), Bl = __builtin_bit_cast(h8, yl[b][q]);
                                D2[cb][b] = mfma16(Ah, Bh, D2[cb][b]);
                                D2[cb][b] = mfma16(Ah, Bl, D2[cb][b]);
                                D2[cb][b] = mfma16(Al, Bh, D2[cb][b]);
                                asm volatile("" : "+a"(D2[cb][b]));
                            }
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the fragments have been read
                    __builtin_amdgcn_sched_barrier(0);
                };
                p2_block(std::integral_constant<int, 0>{});
                p2_block(std::integral_constant<int, 1>{});
                p2_block(std::integral_constant<int, 2>{});
                p2_block(std::integral_constant<int, 3>{});
#ifdef GPDE_BWABL
                }
#endif
                seen[0] = seen[0] || anyb[0];
                seen[1] = seen[1] || anyb[1];
            }
            // ---- row maxima over the slice's columns; the edge's partial dx row (slice 0 adds dS_i) ---------------------------
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float rm = fmaxf(rmax[b], gp_other_half(rmax[b]));
                const int er = erow0 + 32 * b + l31;
                if (with_by && vL[b] && h == 0) a.bw_rowmax[(size_t)slice * a.bw_rows + er] = rm;
                if (vL[b]) {
                    const float un = bw_ish * unL[b];
                    float* dxr = a.bw_dxp + ((size_t)slice * a.bw_rows + er) * GP_W + 4 * h;
                    const float* ds = a.bw_dS + (size_t)(nodeL[b] - a.nc0) * GP_W + 4 * h;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c0 = 32 * cb + 8 * g;
                            f32x4 o4;
#pragma unroll
                            for (int t2 = 0; t2 < 4; ++t2) o4[t2] = D2[cb][b][4 * g + t2] * un;
                            if (slice == 0) {
                                const f32x4 dv = *(const f32x4*)(ds + c0);
#pragma unroll
                                for (int t2 = 0; t2 < 4; ++t2) o4[t2] += dv[t2];
                            }
                            *(f32x4*)(dxr + c0) = o4;
                        }
                }
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                  for (int r = 0; r < 16; ++r) { acc[b][nb][r] = 0.f; if (nb < 2) D2[b][nb][r] = 0.f; }
        } else if constexpr (WRITE_H) {
            // ---- un-scale + bias + ReLU, store the tile's hidden activations (128-byte runs along the columns) ----
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float ie[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 t4 = *(const f32x4*)&Es[32 * b + 8 * q4 + 4 * h];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ie[4 * q4 + j] = t4[j];
                }
                auto store_rows = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e = e0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float* hp = a.hout + (size_t)(e - a.e_chunk0) * a.K2P + slice * GP_TN + l31;
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            const float y = fmaxf(fmaf(acc[b][nb][r], ie[r] * ucv[nb], b2v[nb]), 0.f);
                            acc[b][nb][r] = 0.f;
                            if (FULL || e < eb) { hp[nb * 32] = y; hmax_run = fmaxf(hmax_run, y); }
                        }
                    }
                };
                if (e0 + TE <= eb) store_rows(std::true_type{});
                else store_rows(std::false_type{});
            }
        } else
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            u4 g2hi[4][2], g2lo[4][2];
            float ie[16];           // per-edge un-scale of this half's 16 accumulator rows: four 16-byte LDS reads
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 t4 = *(const f32x4*)&Es[32 * b + 8 * q4 + 4 * h];
#pragma unroll
                for (int j = 0; j < 4; ++j) ie[4 * q4 + j] = t4[j];
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float y[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    y[r] = fmaf(acc[b][nb][r], ie[r] * ucv[nb], b2v[nb]);
                    acc[b][nb][r] = 0.f;
                }
#pragma unroll
                for (int p_ = 0; p_ < 8; ++p_) {
                    unsigned ph, pl;
                    conv_pair(y[2 * p_], y[2 * p_ + 1], ph, pl);
                    g2hi[nb][p_ >> 2][p_ & 3] = ph;
                    g2lo[nb][p_ >> 2][p_ & 3] = pl;
                }
                __builtin_amdgcn_sched_barrier(0);      // one column block at a time: 16 live values, not 64
            }
            const int s0 = e0 + 32 * b;
            const int s_end = min(s0 + 32, e_end);
            int e_seg = s0;
            const int n_last_b = __builtin_amdgcn_readfirstlane(nl_v[b]);
            int node = __builtin_amdgcn_readfirstlane(nf_v[b]);
            while (e_seg < s_end) {
                const int seg_end = (node == n_last_b) ? s_end : min(a.rowptr[node + 1], s_end);
                if (node != cur) {
                    if (cur >= 0) flush(cur);
                    cur = node;
                }
                const int lo = e_seg - s0 - 4 * h, hi = seg_end - s0 - 4 * h;
                const bool full = (e_seg == s0) && (seg_end == s0 + 32);       // the whole half is one segment (scalar)
                const unsigned* xu = Xs + 32 * b * GP_W;
                // four operand groups g = (cb, m): 8 x_j words each (k slot tq <-> edge er(8m + tq) + 4h), read one
                // group ahead of the 12 MFMAs that use them.  The loads are unconditional and materialised by an asm
                // before the masks: hipcc otherwise sinks each load into its mask's branch and waits per word.
                unsigned wq[2][8];
                auto ldx = [&](int g, unsigned (&w)[8]) {
                    const int cb = g >> 1, m = g & 1;
#pragma unroll
                    for (int tq = 0; tq < 8; ++tq) {
                        const int er = ((8 * m + tq) & 3) + 8 * ((8 * m + tq) >> 2);
                        w[tq] = xu[(er + 4 * h) * GP_W + cb * 32 + l31];
                    }
                };
                ldx(0, wq[0]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cb = g >> 1, m = g & 1;
                    unsigned (&w)[8] = wq[g & 1];
                    if (g + 1 < 4) ldx(g + 1, wq[(g + 1) & 1]);
                    asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
                    if (!full) {
#pragma unroll
                        for (int tq = 0; tq < 8; ++tq) {
                            const int er = ((8 * m + tq) & 3) + 8 * ((8 * m + tq) >> 2);
                            const unsigned keep = (er >= lo && er < hi) ? 0xffffffffu : 0u;
                            w[tq] &= keep;
                        }
                    }
                    u4 ah, al;
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        ah[jp] = __builtin_amdgcn_perm(w[2 * jp + 1], w[2 * jp], 0x05040100u);
                        al[jp] = __builtin_amdgcn_perm(w[2 * jp + 1], w[2 * jp], 0x07060302u);
                    }
                    const h8 xhi = __builtin_bit_cast(h8, ah), xlo = __builtin_bit_cast(h8, al);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        const h8 gh = __builtin_bit_cast(h8, g2hi[nb][m]), gl = __builtin_bit_cast(h8, g2lo[nb][m]);
                        Z[cb][nb] = mfma16(xhi, gh, Z[cb][nb]);
                        Z[cb][nb] = mfma16(xhi, gl, Z[cb][nb]);
                        Z[cb][nb] = mfma16(xlo, gh, Z[cb][nb]);
                        asm volatile("" : "+a"(Z[cb][nb]));      // Z stays in AGPRs (else: 128 copies per segment)
                    }
                }
                e_seg = seg_end;
                if (e_seg < s_end) node = a.dst[e_seg];
            }
        }
        if constexpr (MODE == 0)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+a"(Z[cb][nb]));
        TM_MARK(tm_post);
        // ---- advance ---------------------------------------------------------------------------------------
        if (last_in_blk) {
            have = have_n;
            blk_a = na_;
            blk_b = nb_;
            e0 = have ? blk_a : e_hi;
        } else if (have) {
            e0 += TE;
        }
    }
    if constexpr (WRITE_H) {
        if (a.hmax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) hmax_run = fmaxf(hmax_run, __shfl_xor(hmax_run, o));
            if (lane == 0 && hmax_run > 0.f) atomicMax(a.hmax_out, __float_as_uint(hmax_run));
        }
    } else if constexpr (MODE == 0) {
        if (cur >= 0) flush(cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GPDE_V6_TIMING
    if (lane == 0) {
        atomicAdd(&gpde_v6_tm[0], (unsigned long long)tm_pro);
        atomicAdd(&gpde_v6_tm[1], (unsigned long long)tm_loop);
        atomicAdd(&gpde_v6_tm[2], (unsigned long long)tm_post);
        atomicAdd(&gpde_v6_tm[3], (unsigned long long)n_rounds);
        atomicAdd(&gpde_v6_tm[4], (unsigned long long)tm_peel);
    }
#endif
}

size_t v6_lds_bytes(int K1P) {
    return (size_t)NS * TILE_B + (size_t)K1P * 32 + (size_t)NW * TE * GP_W * 4 + (size_t)NW * TE * 4 + 64 +
           2 * GP_TN * 4;       // (+ the backward mode's per-slice constants)
}

}  // namespace

#ifdef GPDE_V6_TIMING
extern "C" GPDE_API int gpde_debug_v6_timing(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(gpde_v6_tm), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gpde_v6_tm), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// 3-Linear kernels with at least 8 k1 chunks (the side loads of a tile are spread over its first seven), attributes +
// bias slot within one K = 8 group, split-x input (a.xs) present
bool gpde_fused_f16v6_supported(const GpdeFusedArgs& a) {
    return a.K1P / GP_BK >= 8 && a.k0 + 1 <= 8 && v6_lds_bytes(a.K1P) <= 160 * 1024 &&
           (a.hout != nullptr || a.xs != nullptr);
}

int gpde_launch_fused_f16v6(const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = v6_lds_bytes(a.K1P);
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_fused_f16v6_kernel<0, false>, gpde_fused_f16v6_kernel<1, false>, gpde_fused_f16v6_kernel<0, true>,
                             gpde_fused_f16v6_kernel<1, true>)) return rc;
    if (a.hout) {
        GpdeFusedArgs b = a;
        b.blk = nullptr; b.qn = nullptr; b.qctr = nullptr;             // rows are independent: static ranges
        if (a.kt) hipLaunchKernelGGL((gpde_fused_f16v6_kernel<1, true>), grid, block, lds, stream, b);     // row f3 in training
        else hipLaunchKernelGGL((gpde_fused_f16v6_kernel<1, false>), grid, block, lds, stream, b);
    } else if (a.kt) hipLaunchKernelGGL((gpde_fused_f16v6_kernel<0, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gpde_fused_f16v6_kernel<0, false>), grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_fused_f16v6_kernel");
    return GPDE_OK;
}

// The backward's one-pass mode (MODE 2): grid = edge groups x column slices like the other modes; tiles are dealt round-robin
int gpde_launch_fused_bwd(const GpdeFusedArgs& a, hipStream_t stream) {
    if (!a.xs || !a.scal || !a.bw_img2 || !a.bw_unscale || !a.bw_dS || !a.bw_dxp || !a.src || !a.dst || a.hout || a.blk ||
        (a.bw_dU && !a.bw_img1) || (a.bw_dUt && (!a.bw_dU || !a.bw_rowmax || !a.bw_csum || !a.bw_cmax)) || a.bw_rows <= 0 || a.n_groups < 1) {
        gpde_set_error("gpde_launch_fused_bwd: incomplete arguments");
        return GPDE_EINVAL;
    }
    if (!gpde_fused_f16v6_supported(a) || a.K2P % GP_TN != 0) {
        gpde_set_error("gpde_launch_fused_bwd: kernel MLP outside the one-wave-per-SIMD kernel (K1P = %d, K2P = %d, k0 = %d)", a.K1P, a.K2P, a.k0);
        return GPDE_EUNSUPPORTED;
    }
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = v6_lds_bytes(a.K1P);
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_fused_f16v6_kernel<2, false>, gpde_fused_f16v6_kernel<2, true>)) return rc;
    if (a.kt) hipLaunchKernelGGL((gpde_fused_f16v6_kernel<2, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gpde_fused_f16v6_kernel<2, false>), grid, block, lds, stream, a);
    GP_LAUNCH_CHECK("gpde_fused_f16v6_kernel<2>");
    return GPDE_OK;
}

// The store variant of the fused kernel (H rows instead of the aggregation): the one-wave-per-SIMD kernel where its
// shape is covered (>= 8 k1 chunks), else the 8-wave kernel.  GPDE_STORE_V3=1 forces the latter (A/B).
bool gpde_fused_store_supported(GpdeFusedArgs probe) {
    if (!probe.hout) probe.hout = (float*)(uintptr_t)8;      // "will be present"
    return gpde_fused_f16v6_supported(probe) || gpde_fused_f16v3_supported(probe);
}

int gpde_launch_fused_store(const GpdeFusedArgs& f, hipStream_t stream) {
    if (!f.hout) { gpde_set_error("gpde_launch_fused_store: hout is null"); return GPDE_EINVAL; }
    const bool force_v3 = gpde_switches().store_v3;
    if (!force_v3 && gpde_fused_f16v6_supported(f)) return gpde_launch_fused_f16v6(f, stream);
    if (gpde_fused_f16v3_supported(f)) return gpde_launch_fused_f16v3(f, stream);
    gpde_set_error("fused store kernel: unsupported kernel MLP (K1P = %d, k0 = %d)", f.K1P, f.k0);
    return GPDE_EUNSUPPORTED;
}
