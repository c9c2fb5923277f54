// Per-edge backward through the aggregation on split-f16 MFMA (round 4; replaces gpde_edge_bwd2_kernel's fp32 MFMAs for
// graphs with in-degree >= ~32).  Same two products (backward of NNConv_old.message + the scatter,
// /root/reference/graph-neural-operator/nn_conv.py:273-275, reached from loss.backward(), UAI1_full_resolution.py:266):
//   dU[e][n]  = (sum_c x_j[c] dZ_i[c][n]) * [H[e][n] > 0]
//   dx_e[c]   =  sum_n H[e][n] dZ_i[c][n] + dS_i[c]                  (per-edge rows, reduced per source by k_dx_reduce)
//
// Why: the fp32 kernel spends 80 % of a step inside its 64 v_mfma_f32_32x32x2_f32 (64 cycles each, one LDS word and one
// select per MFMA; scripts/eb2_timing.py: 14.5 k of 17.4 k cycles per step, two waves per SIMD) - not waiting for memory.
// On v_mfma_f32_32x32x16_f16 with 2-term split operands (hi x hi + hi x lo + lo x hi, error < 2^-21 per product) a step is
// 24 MFMAs of 32 cycles.
//
// Layout of a step (wave tile = 32 CSR slots x 32 hidden columns, as before), both products TRANSPOSED so that the lane is
// the edge in every result and every per-edge quantity (validity, node membership, scales, ReLU mask) is per lane:
//   P1  D1[n][e] = sum_c dZ[c][n] x_e[c]      A = dZ^T (lane n: 8 channels per k16 step, 16 ds_read_u16 from the node image),
//                                             B = x_e  (registers, converted once per tile);  4 steps x 3 MFMAs
//   P2  D2[c][e] = sum_n dZ[c][n] H[e][n]     A = dZ   (lane c: 8 columns = one 16-byte unit of hi + one of lo),
//                                             B = H_e  (this step's 16 values, converted with the row's own scale); 2 x 2 x 3
// Scales (powers of two, exact): x per edge row, H per edge row AND step (its un-scale is applied when the step's D2 is
// added to the running dx accumulators), dZ per destination node (pre-pass k_dz_split: the node's [64][K2P] block becomes,
// in place, per 32-column group 32 hi halves | 32 lo halves - the same 128 bytes, so the DMA geometry is the fp32 kernel's).
// A 128-slot group spanning two destination nodes runs both nodes' products on all lanes and selects per lane (P1) /
// zeroes the other node's lanes of B (P2); groups spanning more than two run the step loop once per node pair.
//
// By-products (full backward only, GpdeEdgeBwd3Args::dUt): with the dU tile in registers in the transposed orientation the kernel
// also writes dU^T (what the dW_2 = dU_2^T . H_1 GEMM contracts over), each row's power-of-two scale (for the dU_1 = dU_2 . W_2
// GEMM's A operand) and per-tile column sums / maxima (db_2, the dW_2 GEMM's row scales) - gpde_launch_gemm_f16s_tn then skips
// its own 48 GB pass over dU_2 (k_transpose_stats).  Measured at s=121: 25 ms of fp32-MFMA kernel + 12 ms of
// transposing pass -> 17 ms; the kernel is HBM-bound (reads H, writes dU twice: 12 KiB per edge at k2 = 1024, 4.3 TB/s).
#include "gpde_common.h"
#include <cstdlib>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int E3_NC = 32;                          // hidden columns per step
constexpr int E3_DZ = 2 * 64 * E3_NC;              // 4-byte units per dZ buffer (two nodes)
constexpr int E3_H = 32 * E3_NC;                   // floats per wave H buffer

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void e3_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// y = v * sc; hi = rtz16(y); lo = rn16(y - hi): two values per call, packed halves
__device__ __forceinline__ void split2(float v0, float v1, float sc, unsigned& ph, unsigned& pl) {
    unsigned t0, t1;
    asm("v_mul_f32 %1, %5, %3\n\t"
        "v_mul_f32 %2, %5, %4\n\t"
        "v_cvt_pkrtz_f16_f32 %0, %1, %2"
        : "=&v"(ph), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(pl) : "v"(t0), "v"(t1), "v"(ph));
}
// 2^(13 - floor(log2 m)) and its reciprocal for m in the normal range (else 1, 1): m -> [2^13, 2^14)
__device__ __forceinline__ void pow2_scale(float m, float& sc, float& isc) {
    const int eb = (__float_as_int(m) >> 23) & 0xff;
    const bool ok = eb >= 20 && eb <= 230;
    sc = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    isc = ok ? __int_as_float((eb - 13) << 23) : 1.f;
}
__device__ __forceinline__ float other_half(float v) {       // the value of lane (l ^ 32)
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r2[0] : r2[1]);
}

template <int CTRL> __device__ __forceinline__ float dpp_shr(float v) {      // the value CTRL lanes down the 16-lane row, 0 beyond it
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// ---- pre-pass: per-node scale + in-place split image of dZ ------------------------------------------------------------
// One workgroup per node: max |dZ_i| over its [64][K2P] block, s_i = 2^(13 - E(max)); then every 32-column group (128 bytes)
// is rewritten as hi[32] | lo[32] halves of dZ * s_i.  In place: a wave owns whole 1 KiB segments (8 groups), reads a segment
// with ONE load per lane and writes it back after - no other wave touches it.
__global__ __launch_bounds__(256) void k_dz_split(float* __restrict__ dZ, int row_floats, float* __restrict__ unscale) {
    __shared__ unsigned red[4];
    float* p = dZ + (size_t)blockIdx.x * row_floats;
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
    pow2_scale(__uint_as_float(m), sc, isc);
    if (threadIdx.x == 0) unscale[blockIdx.x] = isc;
    for (int seg = wave * 256; seg < row_floats; seg += 1024) {        // 256 floats = 8 groups of 32 columns per wave and turn
        const f32x4 v = *(const f32x4*)(p + seg + lane * 4);
        unsigned h01, l01, h23, l23;
        split2(v[0], v[1], sc, h01, l01);
        split2(v[2], v[3], sc, h23, l23);
        asm volatile("" ::: "memory");                                  // (the loads of a segment precede its stores)
        unsigned* g = (unsigned*)(p + seg + (lane >> 3) * 32);         // the lane's group; its piece t = lane & 7: columns 4t .. 4t+3
        const int t = lane & 7;
        g[2 * t] = h01; g[2 * t + 1] = h23;                              // hi halves: bytes [8t, 8t + 8)
        g[16 + 2 * t] = l01; g[16 + 2 * t + 1] = l23;                    // lo halves: bytes [64 + 8t, ..)
    }
}

// ---- the edge kernel ------------------------------------------------------------------------------------------------------
// ACC: dU is ACCUMULATED into (dU[e][n] += ...) instead of written - the `depth` applications of a module that shares its hidden
// activations sum their dL/dU there (autograd.NNConvHiddenFunction; otherwise autograd adds six [E][K2P] tensors out of the kernel)
template <bool ACC> __device__ __forceinline__ void edge_bwd3_body(const GpdeEdgeBwd3Args& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dZs = smem;                                      // [2 buf][2 node][64 c][128 B: hi 32 | lo 32]
    float* Hs_all = smem + 2 * E3_DZ;                       // [4 waves][2 buf][32 e][32 n] fp32
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Hs = Hs_all + wave * 2 * E3_H;

    // workgroup b runs on XCD b % 8: XCD x takes the x-th contiguous eighth of the groups (see gpde_edge_bwd2_kernel)
    const int grp = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int g0 = a.e0 + grp * 128;
    if (g0 >= a.e1) return;
    const int g1 = min(g0 + 128, a.e1);
    const int t0 = g0 + wave * 32;                          // this wave's tile (may be empty)
    const int e_last = a.e1 - 1;
    const int nA0 = a.dst[g0], nB0 = a.dst[g1 - 1];         // destination range of the group

    // lane = edge t0 + l31 (both halves): destination, source, x_j row as the B operand of P1
    const int eL = t0 + l31;
    const bool vL = eL < a.e1;
    const int nodeL = vL ? a.dst[eL] : -1;
    const int srcL = vL ? a.src[eL] : 0;
    u4 xhi[4], xlo[4];                                      // step s: channels 16 s + 8 h + 0..7
    float isx;
    {
        f32x4 xr[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (vL) v = *(const f32x4*)&a.x[(size_t)srcL * GP_W + 16 * s + 8 * h + 4 * u];
                xr[s][u] = v;
            }
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) m = fmaxf(m, fabsf(xr[s][u][t]));
        m = fmaxf(m, other_half(m));
        float sx;
        pow2_scale(m, sx, isx);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                unsigned p0, q0, p1, q1;
                split2(xr[s][u][0], xr[s][u][1], sx, p0, q0);
                split2(xr[s][u][2], xr[s][u][3], sx, p1, q1);
                xhi[s][2 * u] = p0; xhi[s][2 * u + 1] = p1;
                xlo[s][2 * u] = q0; xlo[s][2 * u + 1] = q1;
            }
    }
    f32x16 dxa[2];                                          // dx_e[c = 32 cb + (r & 3) + 8 (r >> 2) + 4 h], un-scaled by the H row scales
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[cb][r] = 0.f;

    const int NCH = a.K2P / E3_NC;
    const int npass = (nB0 - nA0) / 2 + 1;
    const int nit = npass * NCH;
    // DMA lane roles: one instruction = 8 rows x 8 units of 16 B; LDS position (rr, p) holds unit p ^ rr
    const int rr = lane >> 3, uq = (lane & 7) ^ rr;
    auto issue = [&](int it) {
        const int pass = it / NCH, nc = (it - pass * NCH) * E3_NC, buf = it & 1;
        const int nodeA = nA0 + 2 * pass;
#pragma unroll
        for (int nd = 0; nd < 2; ++nd) {
            if (nd == 1 && nodeA + 1 > nB0) break;          // no second node in this pass: its buffer is never read
            const float* g = a.dZ + ((size_t)(nodeA + nd - a.n0) * GP_W) * a.K2P + nc + uq * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int blk = wave * 2 + i;               // rows 8 blk .. 8 blk + 7
                e3_dma16(g + (size_t)(blk * 8 + rr) * a.K2P, dZs + buf * E3_DZ + nd * 64 * E3_NC + blk * 256);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = min(t0 + i * 8 + rr, e_last);
            e3_dma16(a.H + (size_t)(e - a.e0) * a.K2P + nc + uq * 4, Hs + buf * E3_H + i * 256);
        }
    };

    const int sw = l31 & 7;
    const float unL = vL ? a.dz_unscale[nodeL - a.n0] : 0.f;        // 1 / s_dZ of this lane's destination node
    float rmax = 0.f;                                                // max |dU| of this lane's row (its half of the columns)
    // first pass in which this wave's tile has edges (CSR order: the tile's first edge has its smallest destination)
    const int pass_first = t0 < a.e1 ? (__builtin_amdgcn_readfirstlane(nodeL) - nA0) / 2 : 0;
    issue(0);
    for (int it = 0; it < nit; ++it) {
        const int pass = it / NCH, nc = (it - pass * NCH) * E3_NC, buf = it & 1;
        const int nodeA = nA0 + 2 * pass, nodeB = nodeA + 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 1 < nit) issue(it + 1);
        const bool inA = nodeL == nodeA, inB = nodeL == nodeB;
        const bool anyA = __builtin_amdgcn_ballot_w64(inA) != 0, anyB = __builtin_amdgcn_ballot_w64(inB) != 0;
        if (!anyA && !anyB) continue;
        const float* hb = Hs + buf * E3_H;
        // ---- H_e of this step: columns 16 s2 + 8 h + 0..7 (units 4 s2 + 2 h, + 1 of row e), the row's scale, the split ----
        f32x4 hv[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int u = 0; u < 2; ++u) hv[s2][u] = *(const f32x4*)&hb[l31 * E3_NC + (((4 * s2 + 2 * h + u) ^ sw) << 2)];
        float m = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) m = fmaxf(m, fabsf(hv[s2][u][t]));
        m = fmaxf(m, other_half(m));
        float sh, ish;
        pow2_scale(m, sh, ish);
        u4 bh[2], bl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                unsigned p0, q0, p1, q1;
                split2(hv[s2][u][0], hv[s2][u][1], sh, p0, q0);
                split2(hv[s2][u][2], hv[s2][u][3], sh, p1, q1);
                bh[s2][2 * u] = p0; bh[s2][2 * u + 1] = p1;
                bl[s2][2 * u] = q0; bl[s2][2 * u + 1] = q1;
            }
        // ---- P2: D2[c][e] = sum_n dZ[c][n] H_e[n] over this step's 32 columns; lanes of the other node contribute zeros ----
        f32x16 d2[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[cb][r] = 0.f;
#pragma unroll
        for (int nd = 0; nd < 2; ++nd) {
            if (nd == 0 ? !anyA : !anyB) continue;
            const bool in = nd == 0 ? inA : inB;
            const char* zb = (const char*)(dZs + buf * E3_DZ + nd * 64 * E3_NC);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u4 oh, ol;
#pragma unroll
                for (int q = 0; q < 4; ++q) { oh[q] = in ? bh[s2][q] : 0u; ol[q] = in ? bl[s2][q] : 0u; }
                const h8 Bh = __builtin_bit_cast(h8, oh), Bl = __builtin_bit_cast(h8, ol);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const char* row = zb + (cb * 32 + l31) * 128;            // dZ row c = 32 cb + l31: (c & 7) == sw
                    const h8 Ah = *(const h8*)(row + (((2 * s2 + h) ^ sw) << 4));
                    const h8 Al = *(const h8*)(row + (((4 + 2 * s2 + h) ^ sw) << 4));
                    d2[cb] = mfma16(Ah, Bh, d2[cb]);
                    d2[cb] = mfma16(Ah, Bl, d2[cb]);
                    d2[cb] = mfma16(Al, Bh, d2[cb]);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dxa[cb][r] = fmaf(d2[cb][r], ish, dxa[cb][r]);
        if (!a.dU) continue;
        f32x4 du_old[4];
        if constexpr (ACC) {          // the running sum's piece of this lane: requested before P1's MFMAs, added after them
            const float* du = a.dU + (size_t)((vL ? eL : a.e0) - a.e0) * a.K2P + nc + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) du_old[g] = *(const f32x4*)(du + 8 * g);
        }
        // ---- P1: D1[n][e] = sum_c dZ[c][n] x_e[c]; both nodes' products on all lanes, selected per lane ----------------
        auto p1 = [&](int nd) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            const char* zb = (const char*)(dZs + buf * E3_DZ + nd * 64 * E3_NC);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // A: lane n = l31, slot j <-> channel c = 16 s + 8 h + j (c & 7 == j): one half of the hi plane, one of the lo plane
                h8 Ah, Al;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const char* row = zb + (16 * s + 8 * h + j) * 128 + (l31 & 7) * 2;
                    Ah[j] = *(const _Float16*)(row + ((((l31 >> 3)) ^ j) << 4));
                    Al[j] = *(const _Float16*)(row + ((((l31 >> 3) + 4) ^ j) << 4));
                }
                const h8 Bh = __builtin_bit_cast(h8, xhi[s]), Bl = __builtin_bit_cast(h8, xlo[s]);
                d = mfma16(Ah, Bh, d);
                d = mfma16(Ah, Bl, d);
                d = mfma16(Al, Bh, d);
            }
            return d;
        };
        f32x16 d1;
        if (anyA) d1 = p1(0);
        if (anyB) {
            const f32x16 d1b = p1(1);
            if (anyA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d1[r] = inB ? d1b[r] : d1[r];
            } else d1 = d1b;
        }
        // dU row of this lane's edge: columns nc + 8 g + 4 h + 0..3 per group g = r >> 2 (zeros for a lane outside this pass)
        const bool mine = vL && (inA || inB);
        float o[16];
        {
            const float un = isx * unL;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 hm = *(const f32x4*)&hb[l31 * E3_NC + (((2 * g + h) ^ sw) << 2)];
#pragma unroll
                for (int t = 0; t < 4; ++t) o[4 * g + t] = (mine && hm[t] > 0.f) ? d1[4 * g + t] * un : 0.f;
            }
        }
        if (mine) {
            float* du = a.dU + (size_t)(eL - a.e0) * a.K2P + nc + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
                if constexpr (ACC) v += du_old[g];
                *(f32x4*)(du + 8 * g) = v;
            }
        }
        if (a.dUt) {
            // ---- what the dW_2 GEMM and the dU_1 GEMM need from a pass over dU, formed here where the tile sits in registers
            // (gpde_launch_gemm_f16s_tn's k_transpose_stats otherwise: 24 GB read + 24 GB written per backward at s=121) ------
            // transposed copy: row nc + n, 32 consecutive edges per half wave = one 128-byte run
            if (mine) {
                float* dt = a.dUt + (size_t)(nc + 4 * h) * a.ldt + (eL - a.e0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dt[(size_t)(8 * g + t) * a.ldt] = o[4 * g + t];
            }
            // row maximum (this lane's 16 of the step's 32 columns; the halves are joined at the end)
#pragma unroll
            for (int r = 0; r < 16; ++r) rmax = fmaxf(rmax, fabsf(o[r]));
            // column sums and maxima over the tile's 32 edges: DPP tree inside each half wave (fixed order: reproducible),
            // totals in lanes 31 / 63 -> per-tile partials [tile][K2P], reduced in tile order by k_tile_col_reduce
            float cs[16], cm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sv = o[r], mv = fabsf(o[r]);
                sv += dpp_shr<0x111>(sv); mv = fmaxf(mv, dpp_shr<0x111>(mv));      // row_shr:1, 2, 4, 8 (zero fill at the row start)
                sv += dpp_shr<0x112>(sv); mv = fmaxf(mv, dpp_shr<0x112>(mv));
                sv += dpp_shr<0x114>(sv); mv = fmaxf(mv, dpp_shr<0x114>(mv));
                sv += dpp_shr<0x118>(sv); mv = fmaxf(mv, dpp_shr<0x118>(mv));
                // lane 15 of rows 0 / 2 into every lane of rows 1 / 3: lanes 31 and 63 hold their half wave's total
                sv += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv), 0x142, 0xa, 0xf, false));
                mv = fmaxf(mv, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mv), 0x142, 0xa, 0xf, false)));
                cs[r] = sv; cm[r] = mv;
            }
            if (l31 == 31 && t0 < a.e1) {
                const size_t po = (size_t)((t0 - a.e0) >> 5) * a.K2P + nc + 4 * h;
                float* ps = a.csum_part + po;
                unsigned* pm = a.cmax_part + po;
                const bool first = pass == pass_first;          // this wave's first pass with edges: plain stores, later passes add
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 sv4 = {cs[4 * g], cs[4 * g + 1], cs[4 * g + 2], cs[4 * g + 3]};
                    u4 mv4 = {__float_as_uint(cm[4 * g]), __float_as_uint(cm[4 * g + 1]), __float_as_uint(cm[4 * g + 2]), __float_as_uint(cm[4 * g + 3])};
                    if (!first) {
                        const f32x4 ps0 = *(const f32x4*)(ps + 8 * g);
                        const u4 pm0 = *(const u4*)(pm + 8 * g);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { sv4[t] += ps0[t]; mv4[t] = max(mv4[t], pm0[t]); }
                    }
                    *(f32x4*)(ps + 8 * g) = sv4;
                    *(u4*)(pm + 8 * g) = mv4;
                }
            }
        }
    }
    if (a.dUt && vL) {                   // row scales of dU for the dU . W^T GEMM (k_row_scales_from_parts' formula)
        rmax = fmaxf(rmax, other_half(rmax));
        float rs, irs;
        pow2_scale(rmax, rs, irs);
        if (h == 0) { a.row_sc[eL - a.e0] = rs; a.row_isc[eL - a.e0] = irs; }
    }
    // ---- dx_e[c] = D2 sum / s_dZ(node) + dS_i[c]: 16-byte pieces of the edge's row --------------------------------------
    if (vL) {
        const float un = unL;
        const float* ds = a.dS + (size_t)(nodeL - a.n0) * GP_W + 4 * h;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = 32 * cb + 8 * g;             // + 4 h + t
                const f32x4 dv = *(const f32x4*)(ds + c0);
                f32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = fmaf(dxa[cb][4 * g + t], un, dv[t]);
                if (a.dxe) *(f32x4*)(a.dxe + (size_t)(eL - a.e0) * GP_W + c0 + 4 * h) = o;
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) atomicAdd(&a.dx[(size_t)srcL * GP_W + c0 + 4 * h + t], o[t]);
                }
            }
    }
}
__global__ __launch_bounds__(256, 2) void gpde_edge_bwd3_kernel(GpdeEdgeBwd3Args a) { edge_bwd3_body<false>(a); }
__global__ __launch_bounds__(256, 2) void gpde_edge_bwd3_acc_kernel(GpdeEdgeBwd3Args a) { edge_bwd3_body<true>(a); }

}  // namespace

int gpde_launch_dz_split(float* dZ, int nn, int K2P, float* unscale, hipStream_t stream) {
    if (nn <= 0) return GPDE_OK;
    if (K2P % 32 != 0 || (GP_W * K2P) % 256 != 0) { gpde_set_error("gpde_launch_dz_split: K2P = %d", K2P); return GPDE_EINVAL; }
    hipLaunchKernelGGL(k_dz_split, dim3(nn), dim3(256), 0, stream, dZ, GP_W * K2P, unscale);
    GP_LAUNCH_CHECK("k_dz_split");
    return GPDE_OK;
}

int gpde_launch_edge_bwd3(const GpdeEdgeBwd3Args& a, hipStream_t stream) {
    const int rows = a.e1 - a.e0;
    if (rows <= 0) return GPDE_OK;
    if (a.K2P % E3_NC != 0) { gpde_set_error("gpde_launch_edge_bwd3: K2P = %d", a.K2P); return GPDE_EINVAL; }
    const size_t lds = (size_t)(2 * E3_DZ + 4 * 2 * E3_H) * 4;
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_edge_bwd3_kernel, gpde_edge_bwd3_acc_kernel)) return rc;
    if (a.du_accumulate) {
        if (!a.dU || a.dUt) { gpde_set_error("gpde_launch_edge_bwd3: accumulation needs dU and excludes the by-products"); return GPDE_EINVAL; }
        hipLaunchKernelGGL(gpde_edge_bwd3_acc_kernel, dim3(((rows + 127) / 128 + 7) / 8 * 8), dim3(256), lds, stream, a);
    } else hipLaunchKernelGGL(gpde_edge_bwd3_kernel, dim3(((rows + 127) / 128 + 7) / 8 * 8), dim3(256), lds, stream, a);
    GP_LAUNCH_CHECK("gpde_edge_bwd3_kernel");
    return GPDE_OK;
}
