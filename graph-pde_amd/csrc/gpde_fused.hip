// Fused edge-MLP + outer-product aggregation kernel (the dominant kernel of the NNConv forward).
//
// Replaces, for one destination-node chunk and one 128-column slice of the last hidden layer:
//   DenseNet.forward   (/root/reference/graph-neural-operator/utilities.py:223-227)  hidden part
//   NNConv_old.message (/root/reference/graph-neural-operator/nn_conv.py:273-275)     re-associated
//   PyG gather x[edge_index[0]] + scatter-add over edge_index[1]  
//
// Math.  With h_e = last hidden activation of edge e (k2 values) and
// W_e[c][o] = sum_k W3[c*64+o][k] h_e[k] + b3[c*64+o], the reference computes
//   sum_{e -> i} x_j(e) . W_e   =   sum_{c,k} W3[c*64+o][k] * Z_i[c][k]  +  (sum_e x_j(e)) . B3
// where   Z_i[c][k] = sum_{e -> i} x_j(e)[c] * h_e[k]      <-- this kernel.
// The [E,4096] weight tensor is never formed: per edge the work is the k1 x k2 hidden layer plus
// a 64 x k2 outer-product accumulation instead of the 4096 x k2 last layer (64x fewer FLOPs there).
//
// Mapping to CDNA4 (one wave per SIMD, 4 waves per workgroup, v_mfma_f32_32x32x2_f32 only —
// exact fp32, bitwise an fmaf chain):
//   * edges are in destination-sorted CSR order; each WAVE owns a contiguous, node-aligned edge
//     range and walks it in tiles of 32 edges (one MFMA row block); all 4 waves of a workgroup work
//     on the same 128-wide column slice of the hidden layer and share its W2 tiles through LDS;
//   * GEMM1: H2[32 x 128] = relu(H1[32 x k1] . W2s^T + b2).  H1 is never stored: for each 32-wide
//     k1 chunk it is produced by 4 MFMAs per row block as (W1|b1)[32 x 8] . attr^T[8 x 32], whose
//     D-layout (col = edge = lane&31) is exactly the A-operand layout of the next MFMA, with the
//     k order permuted the same way on the W2 side (k = 8q + 4h + t: one ds_read_b128 per lane);
//   * GEMM2: Z[64 c x 128] += Xg^T[64 x 32 edges] . H2[32 edges x 128]: H2 stays in the GEMM1
//     accumulators (its D-layout is the B-operand layout, k = edge), x_j rows are staged in LDS;
//     segment boundaries (several destinations inside one tile) are handled by masking the
//     A operand by edge range and flushing the 64x128 accumulator once per destination node.
//     Each (node, slice) is written by exactly one wave with plain stores: deterministic, no atomics.
#include "gpde_common.h"

namespace {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// NOT inline asm: hipcc does not pad the MFMA-result -> reader hazard for an asm statement that
// reads a VGPR an MFMA has just written (seen as ~1e-5 errors when the accumulators live in VGPRs)
__device__ __forceinline__ float relu1(float v) { return fmaxf(v, 0.f); }

// two-term f16 split of 16 non-negative, pre-scaled H1 values (registers r = 0..15 of the H1
// MFMA result) into the two K=16 A operands of v_mfma_f32_32x32x16_f16: operand m, element j
// = register 8m + j.  hi = rtz16(y) (packed convert), lo = rn16(y - hi):  y = hi + lo + O(2^-21 y).
__device__ __forceinline__ void split_f16(const f32x16& v, h8 (&hi)[2], h8 (&lo)[2]) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
            const float y0 = v[8 * m + 2 * jp], y1 = v[8 * m + 2 * jp + 1];
            const auto p = __builtin_amdgcn_cvt_pkrtz(y0, y1);
            const _Float16 p0 = (_Float16)p[0], p1 = (_Float16)p[1];
            hi[m][2 * jp] = p0;
            hi[m][2 * jp + 1] = p1;
            lo[m][2 * jp] = (_Float16)(y0 - (float)p0);
            lo[m][2 * jp + 1] = (_Float16)(y1 - (float)p1);
        }
}

// first node n in [lo, hi] with rowptr[n] >= target
__device__ __forceinline__ int lower_bound_node(const int32_t* __restrict__ rowptr, int lo, int hi,
                                                long target) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((long)rowptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int BS_TILE = GP_TN * GP_BS_STRIDE;        // floats per LDS W2 buffer
constexpr int XS_WAVE = GP_TE * GP_W;                // floats per wave x-stage

template <int MODE, bool F16S>
__global__ __launch_bounds__(256, 1) void gpde_fused_kernel(GpdeFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                                 // [2][128][36]      (mode 1)
    float* Xs_all = smem + 2 * BS_TILE;               // [4][32][64]
    int* red = (int*)(Xs_all + GP_WAVES * XS_WAVE);   // [4]
    float* Es_all = (float*)(red + 4);                // [4][32] per-edge 2^-s_e (f16 split)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Xs = Xs_all + wave * XS_WAVE;
    float* Es = Es_all + wave * GP_TE;

    const int ns = a.K2P / GP_TN;
    const int slice = blockIdx.x % ns;
    const int group = blockIdx.x / ns;
    const int NKC = a.K1P / GP_BK;

    // ---- this wave's node-aligned edge range ---------------------------------------------------
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
    int maxtiles = ntiles;
    if (MODE == 1) {
        if (lane == 0) red[wave] = ntiles;
        __syncthreads();
        maxtiles = max(max(red[0], red[1]), max(red[2], red[3]));
    }

    // ---- W2 tile staging (mode 1) ---------------------------------------------------------------
    const float* w2s = (F16S ? (const float*)a.w2h : a.w2t) + (size_t)slice * NKC * (GP_TN * GP_BK);
    f32x4 stage[4];
    auto load_stage = [&](int kc) {
        const f32x4* p = (const f32x4*)(w2s + (size_t)kc * (GP_TN * GP_BK));
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[i] = p[tid + 256 * i];
    };
    auto write_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i;
            *(f32x4*)&Bs[buf * BS_TILE + (f >> 3) * GP_BS_STRIDE + (f & 7) * 4] = stage[i];
        }
    };

    auto load_w1 = [&](int kc) {
        return *(const f32x4*)&a.w1[((size_t)(kc * GP_BK + l31) * 2 + h) * 4];
    };
    float b2v[4], ucv[4];
    f32x4 w1f0 = {0.f, 0.f, 0.f, 0.f}, w1f1 = w1f0, w1f2 = w1f0, wmx = w1f0;
    if (MODE == 1) {
        w1f0 = load_w1(0);
        w1f1 = load_w1(1 % NKC);
        w1f2 = load_w1(2 % NKC);
        if (F16S) {
            wmx = *(const f32x4*)&a.w1[((size_t)a.K1P * 2 + h) * 4];     // appended max-|W1b| row
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ucv[nb] = a.ucol[slice * GP_TN + nb * 32 + l31];
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) b2v[nb] = a.b2[slice * GP_TN + nb * 32 + l31];
        if (maxtiles > 0) {
            load_stage(0);
            write_stage(0);
        }
        __syncthreads();
    }

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

    // The tile loop is branch-free on purpose: waves that have run out of edges keep executing
    // the same instruction stream on empty tiles (all loads predicated off, zero operands), which
    // costs nothing because the workgroup advances at the pace of its slowest wave anyway, and
    // it keeps the accumulators pinned in registers across the loop (no phi copies).
    int g = 0;  // running W2 chunk counter (buffer parity)
    for (int t = 0; t < maxtiles; ++t) {
        const int e0 = ea + t * GP_TE;
        const int e_end = min(e0 + GP_TE, eb);

        // ---- per-tile edge setup: attributes (MFMA operand), x_j rows -> LDS -------------------
        float attrv[4];
        if (MODE != 2) {
            const int e = e0 + l31;
            const bool valid = e < eb;
            const int p = valid ? a.perm[e] : 0;
            const float* ap = a.attr + (size_t)p * a.k0;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int d = 2 * s + h;
                float v = 0.f;
                if (valid && d < a.k0) v = ap[d];
                if (valid && d == a.k0) v = 1.f;   // bias slot
                attrv[s] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < GP_TE / 4; ++i) {
            const int er = (lane >> 4) + 4 * i;
            const int e = e0 + er;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < eb) v = *(const f32x4*)&a.x[(size_t)a.src[e] * GP_W + (lane & 15) * 4];
            *(f32x4*)&Xs[er * GP_W + (lane & 15) * 4] = v;
        }

        // ---- H2 pre-activation tile [32 edges x 128 cols] in acc1 ---------------------------------
        f32x16 acc1[4];
        if (MODE == 1) {
            if (F16S) {
                // Per-edge power-of-two scale for the f16 split: max_k H1[e][k] <= B_e =
                // sum_d max_k|W1b[k][d]| * |attr_e[d]|; 2^s_e puts B_e into [2^13, 2^14), well inside
                // the f16 range, and is applied to the attributes (exact), so H1 comes out of the
                // MFMA pre-scaled.  2^-s_e goes to LDS for the un-scaling after the K loop.
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) part = fmaf(wmx[s], fabsf(attrv[s]), part);
                const float bnd = part + __shfl_xor(part, 32);
                const int eb = (__float_as_int(bnd) >> 23) & 0xff;
                const bool okb = (eb >= 20) && (eb <= 230);
                const float sc = okb ? __int_as_float((267 - eb) << 23) : 1.f;      // 2^(13 - E(B))
                const float isc = okb ? __int_as_float((eb - 13) << 23) : 1.f;      // 2^(E(B) - 13)
#pragma unroll
                for (int s = 0; s < 4; ++s) attrv[s] *= sc;
                if (h == 0) Es[l31] = isc;
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[nb][r] = F16S ? 0.f : b2v[nb];

            auto h1gen = [&](const f32x4& w1f) {
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) d = mfma32(w1f[s], attrv[s], d);
                return d;
            };
            if constexpr (F16S) {
                // ---- f16-split hidden GEMM, software-pipelined two chunks deep ----------------------
                // iteration kc: (a) 4 fp32 MFMAs produce the raw H1 of chunk kc+2, (b) VALU turns
                // the raw H1 of chunk kc+1 (made in the previous iteration) into f16 hi/lo operands,
                // (c) 24 f16 MFMAs consume chunk kc's operands.  (a)-(c) are independent, so the
                // conversion hides under the MFMAs; sched_barrier(0) fences keep that interleave.
                h8 ahi[2], alo[2], ahi_n[2], alo_n[2];
                f32x16 a_raw;
                {
                    f32x16 a0 = h1gen(w1f0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) a0[r] = relu1(a0[r]);
                    split_f16(a0, ahi, alo);
                    a_raw = h1gen(w1f1);
                }
                f32x4 w1f_nxt = w1f2;                               // (W1|b1) rows of chunk 2
                auto conv_piece = [&](int p_) {                     // one register pair of a_raw
                    const int m = p_ >> 2, jp = p_ & 3;
                    const float y0 = relu1(a_raw[8 * m + 2 * jp]), y1 = relu1(a_raw[8 * m + 2 * jp + 1]);
                    const auto pk = __builtin_amdgcn_cvt_pkrtz(y0, y1);
                    const _Float16 p0 = (_Float16)pk[0], p1 = (_Float16)pk[1];
                    ahi_n[m][2 * jp] = p0;
                    ahi_n[m][2 * jp + 1] = p1;
                    alo_n[m][2 * jp] = (_Float16)(y0 - (float)p0);
                    alo_n[m][2 * jp + 1] = (_Float16)(y1 - (float)p1);
                };
                for (int kc = 0; kc < NKC; ++kc, ++g) {
                    const int buf = g & 1;
                    [[maybe_unused]] const int kn = (kc + 1 < NKC) ? kc + 1 : 0;   // next chunk
                    int kn3 = kc + 3;
                    while (kn3 >= NKC) kn3 -= NKC;
                    const f32x4 w1f_use = w1f_nxt;
                    // W2 tile rows: [hi: 4 x 8 halves | lo: 4 x 8 halves], 144-byte row stride
                    const char* bt = (const char*)(Bs + buf * BS_TILE) + l31 * (GP_BS_STRIDE * 4);
                    const int sw = (l31 >> 1) & 7;            // unit swizzle of the packed image
                    const int u0 = ((0 + h) ^ sw) << 4, u1 = ((2 + h) ^ sw) << 4;
                    h8 bhi[2][4], blo[2][4];
#ifndef GPDE_ABL_NOSTAGE
                    load_stage(kn);
#endif
                    w1f_nxt = load_w1(kn3);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        bhi[0][nb] = *(const h8*)(bt + nb * 32 * (GP_BS_STRIDE * 4) + u0);
                        blo[0][nb] = *(const h8*)(bt + nb * 32 * (GP_BS_STRIDE * 4) + (u0 ^ 64));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x16 a_raw_n = h1gen(w1f_use);                  // raw H1 of chunk kc+2
                    conv_piece(0);
                    conv_piece(1);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        bhi[1][nb] = *(const h8*)(bt + nb * 32 * (GP_BS_STRIDE * 4) + u1);
                        blo[1][nb] = *(const h8*)(bt + nb * 32 * (GP_BS_STRIDE * 4) + (u1 ^ 64));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[m], bhi[m][nb], acc1[nb]);
                        conv_piece(2 + 3 * m);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(ahi[m], blo[m][nb], acc1[nb]);
                        conv_piece(3 + 3 * m);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) acc1[nb] = mfma16(alo[m], bhi[m][nb], acc1[nb]);
                        conv_piece(4 + 3 * m);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ahi[m] = ahi_n[m];
                        alo[m] = alo_n[m];
                    }
                    a_raw = a_raw_n;
#ifndef GPDE_ABL_NOSTAGE
                    write_stage(buf ^ 1);
#endif
#ifndef GPDE_ABL_NOBARRIER
                    __syncthreads();
#endif
                }
            } else {
            f32x16 a_cur = h1gen(w1f0);
#pragma unroll
            for (int r = 0; r < 16; ++r) a_cur[r] = relu1(a_cur[r]);
            f32x4 w1f_nxt = w1f1;                               // (W1|b1) rows of chunk 1

            for (int kc = 0; kc < NKC; ++kc, ++g) {
                const int buf = g & 1;
                const int kn = (kc + 1 < NKC) ? kc + 1 : 0;    // next chunk (of this or the next tile)
                const int kn2 = (kn + 1 < NKC) ? kn + 1 : 0;
                const f32x4 w1f_use = w1f_nxt;
                // issue next chunk's W2 tile loads and the W1 rows of the chunk after it right
                // behind the barrier; they are consumed a full chunk of MFMAs later
#ifndef GPDE_ABL_NOSTAGE
                load_stage(kn);
#endif
                w1f_nxt = load_w1(kn2);
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 a_nxt = h1gen(w1f_use);            // wasted only on the tile's last chunk
                const float* bt = Bs + buf * BS_TILE + l31 * GP_BS_STRIDE + h * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 bf[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        bf[nb] = *(const f32x4*)&bt[nb * 32 * GP_BS_STRIDE + q * 8];
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            acc1[nb] = mfma32(a_cur[q * 4 + t4], bf[nb][t4], acc1[nb]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) a_cur[r] = relu1(a_nxt[r]);
                __builtin_amdgcn_sched_barrier(0);
#ifndef GPDE_ABL_NOSTAGE
                write_stage(buf ^ 1);
#endif
#ifndef GPDE_ABL_NOBARRIER
                __syncthreads();
#endif
            }
            }
            if (F16S) {
                // undo the row (edge) and column scales, add the bias
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ie = Es[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc1[nb][r] = fmaf(acc1[nb][r], ie * ucv[nb], b2v[nb]);
                }
            }
        } else if (MODE == 0) {
            // single hidden layer: H = relu((W1|b1) . attr); operands swapped so that D is [edge][col]
            f32x4 w1n[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                w1n[nb] = *(const f32x4*)&a.w1[((size_t)(slice * GP_TN + nb * 32 + l31) * 2 + h) * 4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) d = mfma32(attrv[s], w1n[nb][s], d);
                acc1[nb] = d;
            }
        } else {
            // hidden activations precomputed by the dense front layers: hbuf[slot][K2P]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = e0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float* hp = a.hbuf + (size_t)(e - a.e_chunk0) * a.K2P + slice * GP_TN + l31;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc1[nb][r] = (e < eb) ? hp[nb * 32] : 0.f;
            }
        }

        // relu (mode 2 already has it; harmless there)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nb][r] = relu1(acc1[nb][r]);

        // ---- GEMM2 with destination segments -------------------------------------------------------
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
                const int er = (r & 3) + 8 * (r >> 2);                      // + 4h folded in lo/hi
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
}

}  // namespace

int gpde_launch_fused(int mode, bool f16split, const GpdeFusedArgs& a, hipStream_t stream) {
    const int ns = a.K2P / GP_TN;
    const dim3 grid(a.n_groups * ns), block(256);
    const size_t lds = (size_t)(2 * BS_TILE + GP_WAVES * XS_WAVE + GP_WAVES * GP_TE) * sizeof(float) + 64;
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_fused_kernel<0, false>, gpde_fused_kernel<1, false>, gpde_fused_kernel<1, true>,
                             gpde_fused_kernel<2, false>)) return rc;
    switch (mode) {
        case 0: hipLaunchKernelGGL((gpde_fused_kernel<0, false>), grid, block, lds, stream, a); break;
        case 1:
            if (f16split) hipLaunchKernelGGL((gpde_fused_kernel<1, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((gpde_fused_kernel<1, false>), grid, block, lds, stream, a);
            break;
        case 2: hipLaunchKernelGGL((gpde_fused_kernel<2, false>), grid, block, lds, stream, a); break;
        default: gpde_set_error("bad fused mode %d", mode); return GPDE_EINVAL;
    }
    GP_LAUNCH_CHECK("gpde_fused_kernel");
    return GPDE_OK;
}
