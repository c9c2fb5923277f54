// C[M][N] = (A[M][K] . B[N][K]^T) (.) [mask > 0]   on 2-term split f16 MFMA, fp32 accumulate - the backward's
// dU_1 = (dU_2 . W_2) (.) [H_1 > 0]  (what autograd computes through the ReLU + Linear of DenseNet.forward,
// /root/reference/graph-neural-operator/utilities.py:223-227, on `loss.backward()`, UAI1_full_resolution.py:266).
//
// Same arithmetic and the same K loop as the forward kernel gpde_fused_f16v6_kernel (gpde_fused_f16v6.hip): one wave
// per SIMD, 512 registers, wave tile 64 rows x 128 columns, 3 MFMAs per product (hi.lo, hi.hi, lo.hi), B = the
// pre-split, pre-swizzled 16 KiB chunk images of gpde_mlp_pack's W2 layout streamed through a 3-slot LDS ring by
// LDS-DMA.  What differs is where the A operand comes from: fp32 rows in memory instead of the H1 MFMA.
//   * every wave stages the 32-k chunks of ITS 64 rows through a private 3-slot LDS ring by LDS-DMA in FULL 128-byte
//     lines (8 rows per 1 KiB piece, 8 pieces per chunk).  The first version loaded the MFMA fragments straight from
//     memory (32 bytes of 32 different lines per instruction): the texture-address path, not HBM, bound it at 12.6 ns
//     per row (cdna guide: "fragment-shaped loads", +18..45 %).  Units are XOR-swizzled through the source address so
//     that the fragment reads - lane (row l31, k-group h) wants k = 16 m + 8 half + 4 h + {0..3}, the order the packed
//     B image pairs with its operand slots (gpde_pack.hip) - are conflict-free ds_read_b128;
//   * the pieces of chunk c + 3 are issued during chunk c BEHIND its four B pieces and the chunk ends with
//     s_waitcnt vmcnt(8): B(c + 2) and A(c + 2) are retired, A(c + 3) stays in flight across the barrier; chunk c + 1
//     is read and converted (16 pairs) during chunk c into the other operand buffer (the loop is unrolled by two);
//   * rows carry per-row power-of-two scales 2^(13 - E(max_k |A[row][k]|)) from a pre-pass (k_row_scale_kernel: one
//     read of A), B rows their pack-time scales: both exact, undone once after the K loop.  Error per product
//     < 2^-20 as in the forward.
// Rows are independent: tiles of 64 rows are dealt round-robin to the waves, no node alignment, no aggregation.
#include "gpde_common.h"
#include <cstdlib>
#include <cstdio>
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

constexpr int NS = 3;                      // ring slots
constexpr int TILE_B = GP_TN * 128;        // 16 KiB per B chunk image
constexpr int TE = 64;                     // rows per wave tile
constexpr int NW = 4;

// per-row scale: sc[row] = 2^(13 - E(max_k |A[row][k]|)) (1 for an all-zero / non-finite-range row), isc = 1 / sc
__global__ void k_row_scale_kernel(const float* __restrict__ A, int M, int K, int lda, float* __restrict__ sc,
                                   float* __restrict__ isc) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    unsigned m = 0;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 v = *(const f32x4*)&A[(size_t)row * lda + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) {
        const int eb = (int)((m >> 23) & 0xff);
        const bool ok = eb >= 20 && eb <= 230;
        sc[row] = ok ? __int_as_float((267 - eb) << 23) : 1.f;
        isc[row] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
    }
}

#ifdef GPDE_NT_TIMING      // developer probe (scripts/nt_timing.py): clock64 ticks per phase summed over waves and tiles;
                           // [0..4] plain row tiles, [5..9] split-K (dW_2), [10..14] gather: prologue, K loop, drain, epilogue, tiles
__device__ unsigned long long gpde_nt_tm[16];
#define NT_MARK(acc) do { const long long tm1_ = clock64(); acc += tm1_ - tm0_; tm0_ = tm1_; } while (0)
#else
#define NT_MARK(acc) do { } while (0)
#endif

constexpr int A_SLOT = NW * TE * 128;      // 32 KiB: one 32-k chunk of the workgroup's 256 rows (fp32)

// GATHER (the depth-deferred backward's dU_2 = (sum_l x_j^(l) . dZ_i^(l)) (.) [H_2 > 0], gpde_launch_gemm_f16s_gather): row = CSR
// slot e (j -> i); A[e][(l, c)] = x^(l)[j][c] is gathered through g_src from the stack of node tables (chunk = one half row of
// one layer, a full 128-byte line); B depends on the destination node: a workgroup tile is 256 consecutive slots of ONE node
// (g_tile), its four waves share that node's image through the ring.  Row scales are per SOURCE node (sc / isc indexed by
// g_src), the image carries one scale per node (g_unscale).
// FL (round 6, GpdeGemmF16sArgs::fl_mode): the first hidden layer H_1 = relu(W_1 . attr + b_1) of the kernel MLP is produced INSIDE
// this kernel from the 8 attribute slots of an edge, by the forward kernel's split-f16 MFMA pair ([a_hi|a_hi] x [w_hi;w_lo] +
// [a_lo|a_lo] x [w_hi;0], gpde_fused_f16v6.hip), with A = attributes (rows = edges) and B = the column image: D[edge][n] has the
// accumulator layout of this kernel's tiles AND - eight consecutive registers of a lane - the B-operand layout of its K loop when the
// contraction runs over the edges.
//   FL == 1 (split-K form, dW_2 = dU_2^T . H_1): wave w generates column block w of chunk c + 2's B image during chunk c - 2 MFMAs,
//     8 conversion pairs, 4 ds_write_b128 into the ring slot the DMA used to fill.  The H_1^T split image (4 bytes per edge and
//     column: 25.8 GB written and read back per backward at s=121) and k_first_layer_pack (9 ms) are gone.  The chunk's attribute
//     OPERANDS (32 edges x [a_hi | a_lo] = 1 KiB, scaled and split once per edge by k_first_layer_aops: 32 bytes per edge instead of
//     28 VALU per chunk in each of the four waves of each of the 32 workgroups that share a K range) arrive by ONE LDS-DMA per wave
//     and chunk into a private 3-slot stage, issued five chunks ahead in front of the A pieces.
//   FL == 2 (plain row tiles, dU_1 = (dU_2 . W_2) (.) [H_1 > 0]): the epilogue also forms the FIRST layer's gradients from the tile in
//     its registers - dW_1[n][d] += dU_1[e][n] . attr[e][d], db_1[n] += dU_1[e][n], one [128][8] partial per 64-row tile summed in tile
//     order by the launcher - so dU_1 (4 KiB per edge) is neither written nor read back by k_dw_first.  The ReLU mask comes as bits from
//     k_first_layer_maskbits (the exact fp32 fmaf chain of the forward's first layer).
//     (Tried first, and measured: the mask from the same in-kernel MFMA product as FL == 1.  Its sign differs from the fp32 chain's for
//     pre-activations within 2^-18 of the column bound instead of 2^-24 - on zero-mean attributes that put dW_1 2.4e-4 from float64
//     where the fp32 mask sits at 7e-7 - so values inside the product's error bound of zero needed the exact chain in a rare branch;
//     MFMAs + wait states + that branch cost the dU_1 GEMM 10 % and cancelled the gain.  A separate 2 ms pass for the bits is cheaper.)
template <bool GATHER, int FL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gpde_gemm_f16s_nt_kernel(GpdeGemmF16sArgs a) {
    static_assert(!(GATHER && FL != 0), "the gather form takes its mask from memory");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                               // [3][16 KiB]  B chunk images
    char* aring = smem + NS * TILE_B;                                // [3][4 waves][64 rows][128 B]  A chunks (fp32)
    float* Es_all = (float*)(aring + NS * A_SLOT);                   // [4][64] per-row un-scales
    [[maybe_unused]] char* flst_all = (char*)(Es_all + NW * TE) + 64;   // FL == 1: [4 waves][3][1 KiB] attribute stage

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    float* Es = Es_all + wave * TE;
    char* awave = aring + wave * (TE * 128);

    const int ns = a.N / GP_TN;
    // Workgroup b runs on XCD b % 8.  The ns column slices of one row group read the SAME A rows (4 KiB per row at
    // K = 1024): they share an XCD, so that seven of eight reads of a row can hit that XCD's L2.
    int slice, group;
    if (a.n_groups % 8 == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = j % ns;
        group = xcd + 8 * (j / ns);
    } else {
        slice = blockIdx.x % ns;
        group = blockIdx.x / ns;
    }
    if (slice & 1) gp_debug_skew(a.skew_us);
    const int NKCT = a.K / GP_BK;                   // chunks of the whole K (the B image's row of tiles)
    const int ntile = GATHER ? a.g_ntiles : (a.M + TE - 1) / TE;      // GATHER: workgroup tiles (<= 256 slots of one node)
    // split-K (ksplits > 1: the weight-gradient use, few rows and a very long K): group = (K split, row quad), one
    // tile per wave, partial products of split s at C + s * cstride
    int NKC = NKCT, kc0 = 0, rounds, tgroup = group, tstride = a.n_groups;
    float* Cout = a.C;
    if (a.ksplits > 1) {
        const int nq = (ntile + NW - 1) / NW;
        int ks = group / nq;
        tgroup = group % nq;
        if (a.ksplits % 8 == 0 && a.n_groups == nq * a.ksplits && !a.no_ks_xcd) {
            // One K range per XCD (workgroup b runs on XCD b % 8): the nq row quads of a K split read the same B image
            // pieces and its ns column slices the same A rows.  With the row-group mapping above the quads of a split sat
            // on different XCDs and every XCD fetched the split's B image for itself (rocprofv3 FETCH_SIZE of the dW_2
            // GEMM at s=121: 13.2 GB per launch for 5.2 GB of operands, profiles/traffic_r04_bwd.json).  Which workgroup
            // computes a (split, quad, slice) tile does not enter its arithmetic: results are bit-identical.
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = nq * ns;
            ks = xcd + 8 * (j / per);
            slice = (j % per) % ns;
            tgroup = (j % per) / ns;
        }
        if (ks >= a.ksplits) return;
        NKC = NKCT / a.ksplits;
        kc0 = ks * NKC;
        tstride = nq;
        rounds = 1;
        Cout += (size_t)ks * a.cstride;
    } else {
        const int stride = a.n_groups * (GATHER ? 1 : NW);
        rounds = (ntile + stride - 1) / stride;
        if (rounds == 0) return;
    }
    auto kc = [&](int c) { return kc0 + c; };
    // GATHER: workgroup tile of round t.  The groups of one XCD (group & 7) take CONSECUTIVE tiles - the row tiles of one
    // destination node read the same image slice and neighbouring source rows: one L2 serves them
    const int gu = (GATHER && a.n_groups % 8 == 0) ? (group >> 3) + (group & 7) * (a.n_groups >> 3) : group;
    auto tile_rec = [&](int t, int& node, int& rs, int& re_load, int& re_store) {
        const int wt = t * a.n_groups + gu;
        const int wc = min(wt, ntile - 1);
        node = __builtin_amdgcn_readfirstlane(a.g_tile[4 * wc]);
        rs = __builtin_amdgcn_readfirstlane(a.g_tile[4 * wc + 1]);
        re_load = __builtin_amdgcn_readfirstlane(a.g_tile[4 * wc + 2]);
        re_store = wt < ntile ? re_load : rs;                     // a workgroup beyond the list computes and stores nothing
    };
    const unsigned long long bbase0 = (unsigned long long)a.bsplit + (size_t)slice * NKCT * TILE_B + wave * 4096;
    unsigned long long bbase = bbase0, nbbase = bbase0;             // GATHER: image of this / the next tile's node
    int g_node = 0, g_rs = 0, g_rel = 1, g_res = 0;
    if (GATHER) {
        tile_rec(0, g_node, g_rs, g_rel, g_res);
        bbase = bbase0 + (size_t)g_node * a.g_bnode_bytes;
    }
    const unsigned lane16 = lane * 16;
    auto b_src = [&](int chunk, unsigned long long base) {
        unsigned long long gb = base + (size_t)chunk * TILE_B;
        asm volatile("" : "+s"(gb));
        return (const char*)(gb + lane16);
    };
    auto aofs = [&](int chunk) -> size_t {
        return GATHER ? (size_t)(chunk >> 1) * a.g_layer_stride + (size_t)(chunk & 1) * GP_BK : (size_t)chunk * GP_BK;
    };
    float ucv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) ucv[nb] = GATHER ? 1.f : a.ucol[slice * GP_TN + nb * 32 + l31];
    const int sw = (l31 >> 1) & 7;
    const int boff[2] = {l31 * 128 + (((0 + h) ^ sw) << 4), l31 * 128 + (((2 + h) ^ sw) << 4)};

    // ---- FL: operands of the H_1 MFMA pair --------------------------------------------------------------------------
    [[maybe_unused]] h8 flB1 = {}, flB2 = {};          // FL == 1: this wave's column block (nb = wave) of the image: h ? w_lo : w_hi ; h ? 0 : w_hi
    [[maybe_unused]] char* flst = flst_all + wave * (3 * 1024);
    auto fl_wops = [&](int nb, h8& b1, h8& b2) {
        const char* wp = (const char*)a.fl_wimg + (size_t)(slice * GP_TN + nb * 32 + l31) * 32;
        const h8 whi = *(const h8*)wp, wlo = *(const h8*)(wp + 16);
        const h8 zero = {};
        b1 = h ? wlo : whi;
        b2 = h ? zero : whi;
    };
    if constexpr (FL == 1) fl_wops(wave, flB1, flB2);
    // D[edge][n] = [a_hi|a_hi] x [w_hi;w_lo] + [a_lo|a_lo] x [w_hi;0]  (asm: VGPR destination; the leading s_nop covers a VALU write
    // of an operand right in front - gpde_fused_f16v6.hip)
    // BOTH MFMAs in ONE asm statement: as two statements the compiler scheduled the VALU that forms b2 (h ? 0 : w_hi) between them,
    // right in front of the second MFMA - it does not know that asm is an MFMA, inserted no wait states, and the matrix pipe read b2
    // before the write had landed: the a_lo . w_hi term (2^-11 of the value) came out wrong now and then, signs of near-zero H_1 values
    // flipped from run to run (found in the ISA after two hours of looking elsewhere)
    auto fl_h1gen = [&](f32x16& dd, h8 a1, h8 a2, h8 b1, h8 b2) {
        asm volatile("s_nop 4\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_32x32x16_f16 %0, %3, %4, %0"
                     : "=&v"(dd) : "v"(a1), "v"(b1), "v"(a2), "v"(b2));
    };
    // relu + split of a pair of H_1 values (the forward's conv_a / conv_b: v_max_i32 is the ReLU on the bit pattern)
    auto fl_conv = [&](float v0, float v1, unsigned& ph, unsigned& pl) {
        unsigned t0_, t1_;
        asm("v_max_i32 %1, 0, %3\n\t"
            "v_max_i32 %2, 0, %4\n\t"
            "v_cvt_pkrtz_f16_f32 %0, %1, %2"
            : "=&v"(ph), "=&v"(t0_), "=&v"(t1_) : "v"(v0), "v"(v1));
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(pl) : "v"(t0_), "v"(t1_), "v"(ph));
    };
    // FL == 1: the attribute rows of chunk `ch` (32 edges x the first 32 bytes of each row) -> the 1 KiB stage slot ch % 3 by ONE LDS-DMA
    // (lane = 16 bytes; rows beyond fl_rows are clamped here and zeroed when read)
    auto fl_issue_attr = [&](int ch) {
        // (fl_attr is the operand IMAGE here: [K padded][2][8] f16 = a_hi | a_lo per edge, zero rows beyond the tensor -
        //  k_first_layer_aops; one wave-wide DMA = the chunk's 32 edges x 32 bytes, contiguous)
        const int e = kc(min(ch, NKC - 1)) * GP_BK + (lane >> 1);
        GPDE_GLDS(a.fl_attr + (size_t)e * 8 + 4 * (lane & 1), flst + (ch % 3) * 1024, 0);
    };
    // ... read back by lane (edge = l31): the A operands [a_hi | a_hi] and [a_lo | a_lo] as they lie
    auto fl_read_attr = [&](int ch, h8& a1, h8& a2) {
        const char* l = flst + (ch % 3) * 1024 + l31 * 32;
        a1 = *(const h8*)l;
        a2 = *(const h8*)(l + 16);
    };
    // ... and the generated column block of chunk `ch` written into ring slot `sl`: lane (n = l31 of block `wave`, half h) holds the
    // edges 16 m + {4h .. 4h+3, 8+4h .. 8+4h+3} of k16 step m in registers 8 m .. 8 m + 7 = unit 2 m + h of its row (hi), unit
    // 4 + 2 m + h (lo) - the layout of gpde_pack.hip's chunk images (k_pack_split_kn)
    auto fl_store = [&](const f32x16& dd, int sl) {
        char* row = ring + sl * TILE_B + wave * 4096 + l31 * 128;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            u4 ph, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned hh, ll;
                fl_conv(dd[8 * m + 2 * p], dd[8 * m + 2 * p + 1], hh, ll);
                ph[p] = hh; pl[p] = ll;
            }
            *(u4*)(row + (((2 * m + h) ^ sw) << 4)) = ph;
            *(u4*)(row + (((4 + 2 * m + h) ^ sw) << 4)) = pl;
        }
    };

    // conversion of a pair of raw values: y = v * row scale, hi = rtz16(y), lo = rn16(y - hi)
    auto conv_a = [&](float v0, float v1, float sc, unsigned& ph, unsigned& t0_, unsigned& t1_) {
        asm("v_mul_f32 %1, %5, %3\n\t"
            "v_mul_f32 %2, %5, %4\n\t"
            "v_cvt_pkrtz_f16_f32 %0, %1, %2"
            : "=&v"(ph), "=&v"(t0_), "=&v"(t1_) : "v"(v0), "v"(v1), "v"(sc));
    };
    auto conv_b = [&](unsigned ph, unsigned t0_, unsigned t1_, unsigned& pl) {
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(pl) : "v"(t0_), "v"(t1_), "v"(ph));
    };

    [[maybe_unused]] h8 fl_a1 = {}, fl_a2 = {};      // FL == 1: A operands of the chunk whose image is generated next (chunk c + 2 in chunk c)
    [[maybe_unused]] f32x16 fl_d;
    // ... and the pieces of one chunk's generation, spread over the MFMA gaps of the K loop (state between the gaps:)
    [[maybe_unused]] u4 fl_oh = {}, fl_ol = {};      // the four pairs of one 16-byte unit of the generated row
    auto fl_dpair = [&](int p) {
        unsigned hh, ll;
        fl_conv(fl_d[2 * p], fl_d[2 * p + 1], hh, ll);
        fl_oh[p & 3] = hh; fl_ol[p & 3] = ll;
    };
    auto fl_dwrite = [&](int m, int sl) {
        char* row = ring + sl * TILE_B + wave * 4096 + l31 * 128;
        *(u4*)(row + (((2 * m + h) ^ sw) << 4)) = fl_oh;
        *(u4*)(row + (((4 + 2 * m + h) ^ sw) << 4)) = fl_ol;
    };
    if constexpr (FL == 1) {
        // chunks 0 and 1 of this slice's B generated here, the A operands of chunk 2 prepared, the attributes of chunks 3 and 4 staged
        fl_issue_attr(0); fl_issue_attr(1); fl_issue_attr(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fl_read_attr(c, fl_a1, fl_a2);
            fl_h1gen(fl_d, fl_a1, fl_a2, flB1, flB2);
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(fl_d));            // MFMA result -> VALU read: not padded for an asm MFMA, and - tied to
                                                                           // the register - not reorderable behind the first reader
            fl_store(fl_d, c);
        }
        fl_read_attr(2, fl_a1, fl_a2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        fl_issue_attr(3); fl_issue_attr(4);
    } else {   // chunks 0 and 1 of this slice's B
        const char* g0 = b_src(kc(0), bbase);
        char* l0 = ring + wave * 4096;
        GPDE_GLDS(g0, l0, 0); GPDE_GLDS(g0, l0, 1024); GPDE_GLDS(g0, l0, 2048); GPDE_GLDS(g0, l0, 3072);
        const char* g1 = b_src(kc(1), bbase);
        char* l1 = ring + TILE_B + wave * 4096;
        GPDE_GLDS(g1, l1, 0); GPDE_GLDS(g1, l1, 1024); GPDE_GLDS(g1, l1, 2048); GPDE_GLDS(g1, l1, 3072);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][nb][r] = 0.f;
    int slot = 0;           // B ring slot of the current chunk
    int aslot = 0;          // A ring slot of the current chunk
    [[maybe_unused]] char* flas = flst_all + wave * 2048;       // FL == 2: the tile's 64 attribute rows (32 bytes each) of this wave

    // Plain (non-gather) row tiles, one after the other per wave: the A chunks 0..2, the row scales and (bit masks) the
    // mask words of the NEXT tile are requested before this tile's results are stored, so that a tile boundary costs one
    // exposed memory round trip instead of four (A chunk 0, the scales, two batches of mask words), none of them queued
    // behind the tile's 128 stores (vmcnt retires in order).  `pref`: this tile's A chunks 0..2 are already in the ring.
    bool pref = false, nxt_ok = false;
    [[maybe_unused]] float scn[2] = {1.f, 1.f}, iscn = 1.f;
    const int drows = tstride * NW * TE;                            // rows between a wave's consecutive tiles
#ifdef GPDE_NT_TIMING
    long long tm_pro = 0, tm_loop = 0, tm_drain = 0, tm_epi = 0, tm0_ = clock64();
#endif
    for (int t = 0; t < rounds; ++t) {
        int r0, rmax, rend;                                        // first row, last loadable row, end of the stored rows
        if (GATHER) {
            if (t > 0) { g_node = 0; tile_rec(t, g_node, g_rs, g_rel, g_res); bbase = nbbase; }
            int nn_, nrs_, nrl_, nre_;
            tile_rec(t + 1, nn_, nrs_, nrl_, nre_);
            nbbase = bbase0 + (size_t)nn_ * a.g_bnode_bytes;
            r0 = g_rs + wave * TE; rmax = g_rel - 1; rend = g_res;
            const float un = a.g_unscale[g_node];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ucv[nb] = un;
        } else {
            const int tile = (t * tstride + tgroup) * NW + wave;
            r0 = tile * TE; rmax = a.M - 1; rend = a.M;            // may lie beyond M: loads clamp, stores are masked
            pref = nxt_ok;
            nxt_ok = a.ksplits <= 1 && !a.no_tile_prefetch && t + 1 < rounds && (long)r0 + drows + TE <= (long)a.M;   // the next tile is a full one
        }
        // GATHER: the source nodes of the wave tile's 64 rows in ONE load (lane = row), handed round by ds_bpermute - ten
        // dependent index loads per tile otherwise
        int srcv = 0;
        if (GATHER) srcv = a.g_src[min(r0 + lane, rmax)];
        float sc[2];
        if (!GATHER && pref) {
            sc[0] = scn[0]; sc[1] = scn[1];
            Es[lane] = iscn;
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int row = min(r0 + 32 * e + l31, rmax);
                const int srow = GATHER ? __shfl(srcv, 32 * e + l31) : row;
                sc[e] = a.sc[srow];
                if (h == 0) Es[32 * e + l31] = a.isc[srow];
            }
        }
        // A chunk DMA: piece j = rows 8j .. 8j+7 of the wave's tile in FULL 128-byte lines (8 lanes per row), the
        // 16-byte units of a row XOR-swizzled through the SOURCE address (unit u of row r is stored at u ^ ((r>>1)&7):
        // conflict-free ds_read_b128 of the fragments below, as for the B image)
        const float* apiece[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * j + (lane >> 3);
            const int row = min(r0 + r, rmax);
            const size_t arow = GATHER ? (size_t)__shfl(srcv, r) * GP_W : (size_t)row * a.lda;
            apiece[j] = a.A + arow + 4 * ((lane & 7) ^ ((r >> 1) & 7));
        }
        auto issue_a = [&](int as, int chunk) {
            char* l = awave + as * A_SLOT;
            const size_t co = aofs(chunk);
#pragma unroll
            for (int j = 0; j < 8; ++j) GPDE_GLDS(apiece[j] + co, l + j * 1024, 0);
        };
        // fragment reads: piece q = 2 m + half of edge block e holds k = 16 m + 8 half + 4 h + {0..3} = unit 2 q + h
        auto read_a = [&](int as, f32x4 (&raw)[2][4]) {
            const char* l = awave + as * A_SLOT;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    raw[e][q] = *(const f32x4*)(l + (32 * e + l31) * 128 + (((2 * q + h) ^ sw) << 4));
        };
        const int as1 = aslot + 1 == NS ? 0 : aslot + 1, as2 = as1 + 1 == NS ? 0 : as1 + 1;
        if (GATHER || !pref) {
            issue_a(aslot, kc(0));
            issue_a(as1, kc(1));
            issue_a(as2, kc(2));
        }
        h8 bhi[4], blo[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            bhi[nb] = *(const h8*)(ring + slot * TILE_B + nb * 4096 + boff[0]);
            blo[nb] = *(const h8*)(ring + slot * TILE_B + nb * 4096 + (boff[0] ^ 64));
        }
        if (GATHER || !pref) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        // chunk 0 of A landed (1 and 2 may still fly)
        __builtin_amdgcn_sched_barrier(0);
        u4 ahi[2][2][2], alo[2][2][2];                            // [chunk parity][edge block][k16 step]
        f32x4 raw[2][4];
        read_a(aslot, raw);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int p = 0; p < 8; ++p) {                         // p = 4 m + jp
                unsigned ph, pl, t0_, t1_;
                conv_a(raw[e][p >> 1][2 * (p & 1)], raw[e][p >> 1][2 * (p & 1) + 1], sc[e], ph, t0_, t1_);
                conv_b(ph, t0_, t1_, pl);
                ahi[0][e][p >> 2][p & 3] = ph;
                alo[0][e][p >> 2][p & 3] = pl;
            }
        if (GATHER || !pref) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // chunk 1 of A landed (converted during chunk 0)
        __builtin_amdgcn_sched_barrier(0);
        unsigned cph = 0, ct0 = 0, ct1 = 0;
        const size_t nxt_a = (size_t)drows * a.lda;              // the same lanes' rows of the wave's next tile (floats)

        // one chunk; PAR = c & 1 selects the operand buffers (static: the loop is unrolled by two)
        auto chunk = [&](auto par_tag, int c) {
            constexpr int PAR = decltype(par_tag)::value;
            const int slot1 = slot + 1 == NS ? 0 : slot + 1;
            const int slot2 = slot1 + 1 == NS ? 0 : slot1 + 1;
            const int an1 = aslot + 1 == NS ? 0 : aslot + 1;      // A(c + 1): converted during this chunk
            int c2 = c + 2;
            const bool wrap = c2 >= NKC;                              // chunks 0 / 1 of the NEXT tile (GATHER: of its node's image)
            if (wrap) c2 -= NKC;
            const char* gsrc = b_src(kc(c2), wrap ? nbbase : bbase);
            char* ldst = ring + slot2 * TILE_B + wave * 4096;
            const char* rb0 = ring + slot * TILE_B;
            const char* rb1 = ring + slot1 * TILE_B;
            // A(c + 3) -> A(c)'s slot.  Beyond the tile's end: chunks 0..2 of the wave's NEXT tile when that is a full tile
            // of plain rows (nxt_ok), else a clamped re-read that nobody uses
            const size_t c3 = (!GATHER && nxt_ok && c + 3 >= NKC) ? aofs(kc(c + 3 - NKC)) + nxt_a : aofs(kc(min(c + 3, NKC - 1)));
            char* adst = awave + aslot * A_SLOT;
            read_a(an1, raw);      // (reading it one barrier earlier was tried in round 4: no gain, 71.0k vs 70.0k ticks per K loop)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const char* rn = (m == 0) ? rb0 : rb1;
                const int bo = boff[m ^ 1];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int tt = j >> 1, e = j & 1;
                        const int i = nb * 6 + j;
                        acc[e][nb] = mfma16(__builtin_bit_cast(h8, tt == 2 ? alo[PAR][e][m] : ahi[PAR][e][m]),
                                            tt == 0 ? blo[nb] : bhi[nb], acc[e][nb]);
                        asm volatile("" : "+a"(acc[e][nb]));
                        // the 16 conversion pairs of chunk c + 1, eight per step: pair p = 8 m + i / 3 <-> edge block
                        // p >> 3 ... laid out as (e, p8 = 4 m' + jp)
                        {
                            const int pp = 8 * m + i / 3;                 // 0..15
                            const int ce = pp >> 3, p8 = pp & 7;
                            if (i % 3 == 0 && i / 3 < 8)
                                conv_a(raw[ce][p8 >> 1][2 * (p8 & 1)], raw[ce][p8 >> 1][2 * (p8 & 1) + 1], sc[ce], cph, ct0, ct1);
                            if (i % 3 == 1 && i / 3 < 8) {
                                unsigned pl;
                                conv_b(cph, ct0, ct1, pl);
                                ahi[PAR ^ 1][ce][p8 >> 2][p8 & 3] = cph;
                                alo[PAR ^ 1][ce][p8 >> 2][p8 & 3] = pl;
                                asm volatile("" ::"v"(ahi[PAR ^ 1][ce][p8 >> 2]), "v"(alo[PAR ^ 1][ce][p8 >> 2]));
                            }
                        }
                        if (j == 1) blo[nb] = *(const h8*)(rn + nb * 4096 + (bo ^ 64));
                        if (j == 5) bhi[nb] = *(const h8*)(rn + nb * 4096 + bo);
                        if constexpr (FL == 1) {
                            // chunk c + 2's column block of the B image is GENERATED: H_1 MFMA pair early in the chunk, its
                            // conversion + the four ds_write_b128 well behind it, then the A operands of chunk c + 3 from the
                            // staged attributes; the attribute DMA of chunk c + 5 goes out in front of the A pieces (vmcnt order)
                            // (one small piece per MFMA gap: issued as blocks - 40 VALU + 4 LDS writes in one gap, 28 + 2 reads in
                            //  another - the K loop was 19 % slower than with the image DMA; the forward kernel's rule, <= ~5 per gap)
                            if (m == 0) {
                                if (i == 2) fl_h1gen(fl_d, fl_a1, fl_a2, flB1, flB2);
                                if (i == 4) fl_issue_attr(c + 5);
                                if (i == 8) fl_dpair(0);
                                if (i == 11) fl_dpair(1);
                                if (i == 14) fl_dpair(2);
                                if (i == 17) { fl_dpair(3); fl_dwrite(0, slot2); }
                                if (i == 20) fl_dpair(4);
                                if (i == 23) fl_dpair(5);
                            } else {
                                if (i == 5) fl_dpair(6);
                                if (i == 11) { fl_dpair(7); fl_dwrite(1, slot2); }
                                if (i == 17) fl_read_attr(c + 3, fl_a1, fl_a2);       // (ready-made operands: two ds_read_b128)
                            }
                        } else if (m == 0) {
                            if (i == 2) GPDE_GLDS(gsrc, ldst, 0);
                            if (i == 8) GPDE_GLDS(gsrc, ldst, 1024);
                            if (i == 14) GPDE_GLDS(gsrc, ldst, 2048);
                            if (i == 20) GPDE_GLDS(gsrc, ldst, 3072);
                        }
                        // the eight A pieces of chunk c + 3, behind the four B pieces, two per MFMA gap
                        if (m == 1 && (i == 2 || i == 8 || i == 14 || i == 20)) {
                            const int k = (i - 2) / 6;
                            GPDE_GLDS(apiece[2 * k] + c3, adst + (2 * k) * 1024, 0);
                            GPDE_GLDS(apiece[2 * k + 1] + c3, adst + (2 * k + 1) * 1024, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (FL == 1) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");   // A(c + 2), attr(c + 4) landed, the generated block is in the ring; A(c + 3) + attr(c + 5) fly
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // B(c + 2) and A(c + 2) landed; A(c + 3) flies
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot1;
            aslot = an1;
        };
        NT_MARK(tm_pro);
        for (int c = 0; c < NKC; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            chunk(std::integral_constant<int, 1>{}, c + 1);
        }
        NT_MARK(tm_loop);
        // requested before the drain below, used after it: the mask words of this tile's rows (lane = row) and the next
        // tile's row scales
        [[maybe_unused]] u4 mw = {0u, 0u, 0u, 0u};
        [[maybe_unused]] f32x4 fa[2][2];
        if constexpr (FL == 2) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float* ap = a.fl_attr + (size_t)min(r0 + 32 * e + l31, min(rmax, a.fl_rows - 1)) * a.fl_ld0;
                fa[e][0] = *(const f32x4*)ap;
                fa[e][1] = *(const f32x4*)(ap + 4);
            }
        }
        if constexpr (!GATHER) {
            if (a.maskbits && !a.xc_x) mw = *(const u4*)(a.maskbits + (size_t)min(r0 + lane, rmax) * a.ldmb + slice * (GP_TN / 32));
            if (nxt_ok) {
                scn[0] = a.sc[r0 + drows + l31];
                scn[1] = a.sc[r0 + drows + 32 + l31];
                iscn = a.isc[r0 + drows + lane];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail pieces: nothing may land in the ring later
        __builtin_amdgcn_sched_barrier(0);
        NT_MARK(tm_drain);

        // ---- un-scale, mask, store --------------------------------------------------------------------------
        // The 64 mask words of an edge block are loaded as ONE batch (rows clamped, no branches) before the first
        // store: a load -> wait -> select -> store chain per element cost 60 us of a 90 us tile.
        if (!GATHER && a.xc_x) {
            // Contract epilogue (per-edge last layer of the NNConv forward, gpde_api.hip): row = CSR slot e, column
            // n = c * 64 + o of W_e = W3 . h_e (nn_conv.py:273-274: `weight = self.nn(pseudo).view(-1, 64, 64)`); this
            // slice holds c = 2 * slice + {0, 1}.  m_e[o] += x_j[c] * W_e[c][o] (nn_conv.py:275) is formed here, so the
            // [E][4096] tensor is never written: P[slice][e][o] = partial message over the slice's two input channels.
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float xv0[16], xv1[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(r0 + 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h, rmax);
                    const float* xp = a.xc_x + (size_t)a.xc_src[row] * GP_W + 2 * slice;
                    xv0[r] = xp[0];
                    xv1[r] = xp[1];
                }
                auto store_rows = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int row = r0 + rr;
                        const float ie = Es[rr];
                        const float m0 = ie * (acc[e][0][r] * ucv[0] * xv0[r] + acc[e][2][r] * ucv[2] * xv1[r]);
                        const float m1 = ie * (acc[e][1][r] * ucv[1] * xv0[r] + acc[e][3][r] * ucv[3] * xv1[r]);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) acc[e][nb][r] = 0.f;
                        float* pp = Cout + ((size_t)slice * a.M + row) * GP_W + l31;
                        if (FULL || row < rend) { pp[0] = m0; pp[32] = m1; }
                    }
                };
                if (r0 + TE <= rend) store_rows(std::true_type{});
                else store_rows(std::false_type{});
            }
            NT_MARK(tm_epi);
            continue;
        }
        if constexpr (FL == 2) {
            const bool dw_on = a.fl_dw_part != nullptr;
            // this tile's partial first-layer gradients: column (nb, l31), slot d < 7 | 7 = bias, over the lane's 32 rows - as four
            // float pairs per column block (v_pk_fma_f32: two slots per instruction; slot 7's multiplier is 1.0, and fma(v, 1, s) is the
            // sum's own rounding)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 dws[4][4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int d = 0; d < 4; ++d) dws[nb][d] = f32x2{0.f, 0.f};
            {     // the tile's attribute rows where every lane can read the rows its accumulator registers belong to (rows beyond the
                  // end hold a clamped copy: their sums are masked by the row's validity, their stores skipped)
                *(f32x4*)(flas + lane * 32) = h ? fa[1][0] : fa[0][0];
                *(f32x4*)(flas + lane * 32 + 16) = h ? fa[1][1] : fa[0][1];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // row-major over the lane's 16 rows of this half, the four column blocks inside: the row's staged attributes are
                // read ONCE (round 6 first had the column block outside: 256 ds_read_b128 per lane and tile instead of 64, and the
                // epilogue took 14 k cycles of a tile's 84 k).  For a fixed (column, slot) the rows still arrive in ascending order:
                // the sums keep their bits.
                auto store_rows2 = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int row = r0 + rr;
                        const float es = Es[rr];
                        const bool valid = FULL || row < rend;
                        f32x2 t[4];
                        if (dw_on) {
                            const f32x4 t0 = *(const f32x4*)(flas + rr * 32), t1 = *(const f32x4*)(flas + rr * 32 + 16);
                            t[0] = f32x2{t0[0], t0[1]}; t[1] = f32x2{t0[2], t0[3]}; t[2] = f32x2{t1[0], t1[1]}; t[3] = f32x2{t1[2], 1.f};
                        }
                        // keep-mask of the 64 lanes for (row, column block): lanes 0..31 hold row rr0's columns, lanes 32..63 row
                        // rr0 + 4's - the two mask words as one 64-bit lane mask (as in the bit-mask epilogue below)
                        const int rr0 = 32 * e + (r & 3) + 8 * (r >> 2);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            float v = acc[e][nb][r] * (es * ucv[nb]);
                            acc[e][nb][r] = 0.f;
                            const unsigned w0 = __builtin_amdgcn_readlane(mw[nb], rr0), w1 = __builtin_amdgcn_readlane(mw[nb], rr0 + 4);
                            const bool pos = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)w1 << 32) | w0);
                            v = pos ? v : 0.f;
                            if (!a.fl_skip_store && valid) Cout[(size_t)row * a.ldc + slice * GP_TN + nb * 32 + l31] = v;
                            if (dw_on) {
                                // dW_1[col][d] += v * attr[row][d], db_1[col] += v  (rows beyond the end: v is whatever the clamped
                                // loads gave - the sums take the row's validity)
                                const float vb = valid ? v : 0.f;
                                const f32x2 vv = f32x2{vb, vb};
#pragma unroll
                                for (int d = 0; d < 4; ++d) dws[nb][d] = __builtin_elementwise_fma(vv, t[d], dws[nb][d]);
                            }
                        }
                    }
                };
                if (r0 + TE <= rend) store_rows2(std::true_type{});
                else store_rows2(std::false_type{});
            }
            if (dw_on && r0 < rend) {
                // the two lane halves hold different rows of the same column: fold them; the lower half writes the TILE's partial
                // [tile][col][8] (one writer per element; summed over the tiles in tile order by the launcher: deterministic and
                // independent of which wave ran the tile and of the launch geometry; running sums over a wave's tiles would be 32
                // more registers that must survive the 512-register K loop)
                float* pp = a.fl_dw_part + ((size_t)(r0 / TE) * a.N + slice * GP_TN) * 8;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float o[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) o[d] = dws[nb][d >> 1][d & 1] + __shfl_xor(dws[nb][d >> 1][d & 1], 32);
                    if (h == 0) {
                        f32x4* q = (f32x4*)(pp + (size_t)(nb * 32 + l31) * 8);
                        q[0] = f32x4{o[0], o[1], o[2], o[3]};
                        q[1] = f32x4{o[4], o[5], o[6], o[7]};
                    }
                }
            }
            NT_MARK(tm_epi);
            continue;
        }
        const bool has_mask = a.mask != nullptr;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if constexpr (!GATHER) if (a.maskbits) {
                // ReLU mask as bits (the first hidden layer is never materialised, GpdeFirstLayerSpec): row rr's four words sit
                // in lane rr of `mw` (one 16-byte load per lane, above) - no load in the epilogue
                auto store_rows_b = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int row = r0 + rr;
                        const float ie = Es[rr];
                        float* cp = Cout + (size_t)row * a.ldc + slice * GP_TN + l31;
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            // keep-mask of the 64 lanes for (row, column block): lanes 0..31 hold row rr0's columns, lanes
                            // 32..63 row rr0 + 4's - exactly the two words as one 64-bit lane mask (two v_readlane + v_cndmask)
                            const int rr0 = 32 * e + (r & 3) + 8 * (r >> 2);
                            const unsigned w0 = __builtin_amdgcn_readlane(mw[nb], rr0), w1 = __builtin_amdgcn_readlane(mw[nb], rr0 + 4);
                            const bool keep = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)w1 << 32) | w0);
                            float v = acc[e][nb][r] * (ie * ucv[nb]);
                            acc[e][nb][r] = 0.f;
                            v = keep ? v : 0.f;
                            if (FULL || row < rend) cp[nb * 32] = v;
                        }
                    }
                };
                if (r0 + TE <= rend) store_rows_b(std::true_type{});
                else store_rows_b(std::false_type{});
                continue;
            }
            float mk[16][4];
            if (has_mask) {                  // (bit masks: handled above; the gather form takes a float mask only)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(r0 + 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h, rmax);
                    const float* mp = a.mask + (size_t)row * a.ldmask + slice * GP_TN + l31;
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) mk[r][nb] = mp[nb * 32];
                }
            }
            // full tiles (all but the last) store without per-row branches: a branch per row makes hipcc wait for
            // the previous row's stores (vmcnt(0)) before the next
            auto store_rows = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = 32 * e + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int row = r0 + rr;
                    const float ie = Es[rr];
                    float* cp = Cout + (size_t)row * a.ldc + slice * GP_TN + l31;
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) {
                        float v = acc[e][nb][r] * (ie * ucv[nb]);
                        acc[e][nb][r] = 0.f;
                        if (has_mask) v = mk[r][nb] > 0.f ? v : 0.f;
                        if (FULL || row < rend) cp[nb * 32] = v;
                    }
                }
            };
            if (r0 + TE <= rend) store_rows(std::true_type{});
            else store_rows(std::false_type{});
        }
        NT_MARK(tm_epi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GPDE_NT_TIMING
    if (lane == 0) {
        const int tb = GATHER ? 10 : a.ksplits > 1 ? 5 : 0;
        atomicAdd(&gpde_nt_tm[tb + 0], (unsigned long long)tm_pro);
        atomicAdd(&gpde_nt_tm[tb + 1], (unsigned long long)tm_loop);
        atomicAdd(&gpde_nt_tm[tb + 2], (unsigned long long)tm_drain);
        atomicAdd(&gpde_nt_tm[tb + 3], (unsigned long long)tm_epi);
        atomicAdd(&gpde_nt_tm[tb + 4], (unsigned long long)rounds);
    }
#endif
}

}  // namespace

#ifdef GPDE_NT_TIMING
extern "C" GPDE_API int gpde_debug_nt_timing(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(gpde_nt_tm), 128) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gpde_nt_tm), z, 128) != hipSuccess) return -1;
    }
    return 0;
}
#endif

namespace {
// dW[col][d] += T[col][d] (d < 7), db[col] += T[col][7]: the summed tile partials of the dU_1 epilogue into the padded gradient buffers
__global__ void k_dw_part_scatter(const float* __restrict__ T, int N, int ldw, float* __restrict__ dW, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 8) return;
    if ((i & 7) < 7) dW[(size_t)(i >> 3) * ldw + (i & 7)] += T[i];
    else db[i >> 3] += T[i];
}
}  // namespace
// scratch of the dU_1 epilogue's first-layer gradients for M rows: one [N][8] block per 64-row tile + one for their sum
size_t gpde_gemm_f16s_dw_part_floats(int M, int N) { return ((size_t)(M + TE - 1) / TE + 1) * (size_t)N * 8; }

size_t gpde_gemm_f16s_workspace_floats(int M) { return (size_t)2 * (M > 0 ? M : 1); }

// K a multiple of 128 and >= 256 (the chunk loop is unrolled by four, three chunks of rows are in flight),
// N a multiple of 128, rows 16-byte aligned
bool gpde_gemm_f16s_supported(int M, int N, int K, int lda) {
    return M > 0 && N % GP_TN == 0 && K % 128 == 0 && K >= 256 && lda % 4 == 0;
}

int gpde_launch_gemm_f16s_nt(const GpdeGemmF16sArgs& a_in, float* row_scale_ws, hipStream_t stream) {
    GpdeGemmF16sArgs a = a_in;
    if (a.ksplits < 1) a.ksplits = 1;
    a.skew_us = gpde_debug_skew_us();
    a.no_tile_prefetch = gpde_switches().nt_no_prefetch; a.no_ks_xcd = gpde_switches().tn_no_ks_xcd;
    static const bool log_shapes = getenv("GPDE_DEBUG_GEMM_LOG") != nullptr;      // one stderr line per launch (pairs with a rocprofv3 kernel trace)
    if (log_shapes) fprintf(stderr, "[gpde_gemm_f16s] M %d N %d K %d ksplits %d fl %d gather %d mask %d\n", a.M, a.N, a.K, a.ksplits, a.fl_mode,
                            a.g_src != nullptr, a.mask != nullptr);
    if (!gpde_gemm_f16s_supported(a.M, a.N, a.K, a.lda) || a.K % (64 * a.ksplits) != 0 || a.K / a.ksplits < 256) {
        gpde_set_error("gpde_gemm_f16s_nt: unsupported shape M=%d N=%d K=%d ksplits=%d", a.M, a.N, a.K, a.ksplits);
        return GPDE_EUNSUPPORTED;
    }
    {   // C must not overlap A or the mask: the N / 128 column-slice workgroups of a row tile read the same A rows at
        // different times (rounds 1-2 ran dU_1 in place over dU_2: the intermittent wrong grad_W1)
        const size_t ca = a.xc_x ? (size_t)(a.N / GP_TN) * a.M * GP_W * 4
                                 : ((size_t)(a.ksplits - 1) * a.cstride + (size_t)(a.M - 1) * a.ldc + a.N) * 4;
        const size_t aa = ((size_t)(a.M - 1) * a.lda + a.K) * 4;
        const size_t ma = a.mask ? ((size_t)(a.M - 1) * a.ldmask + a.N) * 4 : 0;
        const size_t mba = a.maskbits ? ((size_t)(a.M - 1) * a.ldmb + a.N / 32) * 4 : 0;
        if (a.maskbits && (a.ldmb % 4 != 0 || a.mask)) { gpde_set_error("gpde_gemm_f16s_nt: maskbits rows must be 16-byte multiples and exclude `mask`"); return GPDE_EINVAL; }
        if (gp_overlap(a.C, ca, a.A, aa) || gp_overlap(a.C, ca, a.mask, ma) || gp_overlap(a.C, ca, a.maskbits, mba)) {
            gpde_set_error("gpde_gemm_f16s_nt: output overlaps an operand (internal buffer plan error)");
            return GPDE_EINVAL;
        }
    }
    if (row_scale_ws) {                      // per-row scales from one read of A; otherwise the caller filled sc / isc
        float* sc = row_scale_ws;
        float* isc = row_scale_ws + a.M;
        hipLaunchKernelGGL(k_row_scale_kernel, dim3((a.M + 3) / 4), dim3(256), 0, stream, a.A, a.M, a.K, a.lda, sc, isc);
        a.sc = sc;
        a.isc = isc;
    }
    const int ns = a.N / GP_TN;
    const int ntile = (a.M + TE - 1) / TE;
    int groups;
    if (a.ksplits > 1) {
        groups = (ntile + NW - 1) / NW * a.ksplits;
    } else {
        groups = gpde_num_cus() / ns;
        if (groups < 1) groups = 1;
        const int gcap = (ntile + NW - 1) / NW;
        if (groups > gcap) groups = gcap;
    }
    a.n_groups = groups;
    const size_t lds = (size_t)NS * TILE_B + (size_t)NS * A_SLOT + NW * TE * 4 + 64;
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_gemm_f16s_nt_kernel<false, 0>, gpde_gemm_f16s_nt_kernel<true, 0>, gpde_gemm_f16s_nt_kernel<false, 1>,
                             gpde_gemm_f16s_nt_kernel<false, 2>)) return rc;
    if (a.fl_mode) {
        // the first hidden layer generated in the kernel (see the kernel's header): split-K form -> B chunks, row tiles -> mask
        if ((a.fl_mode == 1) != (a.ksplits > 1) || a.fl_mode > 2 || !a.fl_attr || a.fl_ld0 < 8 || a.fl_ld0 % 4 != 0 || a.fl_rows < 1 || a.xc_x || a.mask ||
            (a.fl_mode == 1 && (!a.fl_wimg || !a.fl_alpha || a.maskbits || a.K / a.ksplits / GP_BK < 6)) ||
            (a.fl_mode == 2 && (!a.maskbits || !a.fl_dw_part))) {
            gpde_set_error("gpde_gemm_f16s_nt: inconsistent in-kernel first-layer arguments (mode %d, ksplits %d)", a.fl_mode, a.ksplits);
            return GPDE_EINVAL;
        }
        if (a.fl_mode == 1) hipLaunchKernelGGL((gpde_gemm_f16s_nt_kernel<false, 1>), dim3(groups * ns), dim3(256), lds + NW * 3 * 1024, stream, a);
        else {
            if (a.fl_dw_part && (!a.fl_dw_out || !a.fl_db_out || a.fl_dw_ld < 8)) { gpde_set_error("gpde_gemm_f16s_nt: first-layer gradient outputs missing"); return GPDE_EINVAL; }
            hipLaunchKernelGGL((gpde_gemm_f16s_nt_kernel<false, 2>), dim3(groups * ns), dim3(256), lds + NW * 2048, stream, a);
            if (a.fl_dw_part) {
                // tile partials [ntile][N][8] -> their ordered sum [N][8] (behind the partials) -> dW_1 / db_1
                float* tot = a.fl_dw_part + (size_t)ntile * a.N * 8;
                if (int rc = gpde_launch_reduce_splits(a.fl_dw_part, (size_t)a.N * 8, ntile, (size_t)a.N * 8, tot, 0, stream)) return rc;
                hipLaunchKernelGGL(k_dw_part_scatter, dim3((a.N * 8 + 255) / 256), dim3(256), 0, stream, tot, a.N, a.fl_dw_ld, a.fl_dw_out, a.fl_db_out);
            }
        }
    } else
        hipLaunchKernelGGL((gpde_gemm_f16s_nt_kernel<false, 0>), dim3(groups * ns), dim3(256), lds, stream, a);
    GP_LAUNCH_CHECK("gpde_gemm_f16s_nt_kernel");
    return GPDE_OK;
}

// ---- per-edge last layer of the forward: messages of the low in-degree graphs ------------------------------------
namespace {
// part[node][o] = sum over the in-edges e of the node (CSR order) of sum over the slices s (ascending) of P[s][e][o]:
// the scatter-add of PyG's propagate (call site nn_conv.py:271) over the partial messages of the contract epilogue.
// One wave per destination node, lane = output channel; loads in batches of eight, sums in order.
__global__ __launch_bounds__(256) void k_edge_slices_reduce(const float* __restrict__ P, int ns, int M,
                                                            const int32_t* __restrict__ rowptr, int n_nodes,
                                                            float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_nodes) return;
    const int r0 = rowptr[i], r1 = rowptr[i + 1];
    float t = 0.f;
    for (int e = r0; e < r1; ++e) {
        const float* pe = P + (size_t)e * GP_W + lane;
        int s = 0;
        for (; s + 8 <= ns; s += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = pe[(size_t)(s + q) * M * GP_W];
#pragma unroll
            for (int q = 0; q < 8; ++q) t += v[q];
        }
        for (; s < ns; ++s) t += pe[(size_t)s * M * GP_W];
    }
    part[(size_t)i * GP_W + lane] = t;
}
}  // namespace

size_t gpde_edge_messages_ws_floats(int64_t n_edges, int n_out) {
    return (size_t)(n_out / GP_TN) * (size_t)n_edges * GP_W + 2 * (size_t)n_edges + 64;
}

// part[node][64] = sum_{e -> node} x_src(e) . view(W3 . H[e], 64, 64)   (H [E][K2P] fp32 rows in CSR order, w3s / ucol3 =
// split image of W3 [4096][K2P] from gpde_mlp_pack); ws: gpde_edge_messages_ws_floats(E, 4096) floats
int gpde_launch_edge_messages(const float* H, int K2P, int64_t n_edges, const void* w3s, const float* ucol3,
                              const float* x, const int32_t* src, const int32_t* rowptr, int64_t n_nodes, float* ws,
                              float* part, hipStream_t stream) {
    const int n_out = GP_W * GP_W, ns = n_out / GP_TN;
    float* Pm = ws;
    float* rsc = ws + (size_t)ns * n_edges * GP_W;
    GpdeGemmF16sArgs g{};
    g.A = H; g.lda = K2P; g.M = (int)n_edges; g.bsplit = w3s; g.ucol = ucol3; g.mask = nullptr; g.ldmask = 0;
    g.C = Pm; g.ldc = GP_W; g.K = K2P; g.N = n_out; g.ksplits = 1; g.cstride = 0;
    g.xc_x = x; g.xc_src = src;
    if (int rc = gpde_launch_gemm_f16s_nt(g, rsc, stream)) return rc;
    hipLaunchKernelGGL(k_edge_slices_reduce, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, stream, Pm, ns, (int)n_edges,
                       rowptr, (int)n_nodes, part);
    GP_LAUNCH_CHECK("k_edge_slices_reduce");
    return GPDE_OK;
}

// ---- operands of the weight-gradient GEMM  dW[n_out][n_in] = sum_e dU[e][n_out] . H[e][n_in] -----------------------
// (the contraction runs over the edges: both operands are transposed into K-contiguous form, one pass each)
namespace {
// bits[col] = max over rows of |M[row][col]| as fp32 bit pattern (bits zeroed by the caller)
__global__ __launch_bounds__(256) void k_colabsmax(const float* __restrict__ M, int rows, int cols, int ld, int splits,
                                                   unsigned* __restrict__ bits) {
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 256 + lane * 4, split = blockIdx.y;
    if (col >= cols) return;
    const int rps = (rows + splits - 1) / splits;
    const int r_lo = split * rps, r_hi = min(rows, r_lo + rps);
    unsigned m[4] = {0u, 0u, 0u, 0u};
    const float* p = M + col;
    for (int r = r_lo + rg; r < r_hi; r += 4) {
        const f32x4 v = *(const f32x4*)(p + (size_t)r * ld);
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = max(m[j], __float_as_uint(v[j]) & 0x7fffffffu);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (m[j]) atomicMax(bits + col + j, m[j]);
}
// the scales of k_row_scale_kernel / pack_w2_f16split_kernel from the maxima
__global__ void k_scales_from_max(const unsigned* __restrict__ bits, int n, float* __restrict__ sc, float* __restrict__ isc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int eb = (int)((bits[i] >> 23) & 0xff);
    const bool ok = eb >= 20 && eb <= 230;
    sc[i] = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    isc[i] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
}
// dst[c][e] = src[e][c] for e < rows, 0 for rows <= e < ldd   (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_pad(const float* __restrict__ src, int rows, int ld,
                                                       float* __restrict__ dst, int ldd) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int e0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = e0 + ty + 4 * i;
        tile[ty + 4 * i][tx] = e < rows ? src[(size_t)e * ld + c0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        dst[(size_t)c * ldd + e0 + tx] = tile[tx][ty + 4 * i];
    }
}
// k_transpose_pad + everything else a single read of dU can give.  A workgroup walks a strip of 1024 rows x 64 columns in
// 16 tiles: transposed copy; column sums and maxima of the strip (registers of wave 0; one partial sum per strip, reduced
// in strip order by the caller: deterministic; maxima by atomicMax: order-free, 1024 x strips of them); row maxima of
// every tile into rowpart[column block][row] (no atomics: 16 partial maxima per row, folded by k_row_scales_from_parts).
// (First version, round 3: one tile per workgroup with atomics for all three - 9.5 M atomicMax on 1024 addresses per edge
// chunk made the backward 35 ms SLOWER than the three separate passes.)
constexpr int TS_STRIP = 1024;
__global__ __launch_bounds__(256) void k_transpose_stats(const float* __restrict__ src, int rows, int ld, float* __restrict__ dst,
                                                         int ldd, int n_out, float* __restrict__ csum_part,
                                                         unsigned* __restrict__ colbits, unsigned* __restrict__ rowpart) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64;
    float sacc = 0.f;
    unsigned cm = 0u;
    for (int e0 = blockIdx.x * TS_STRIP; e0 < min((int)(blockIdx.x + 1) * TS_STRIP, ldd); e0 += 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = e0 + ty + 4 * i;
            tile[ty + 4 * i][tx] = e < rows ? src[(size_t)e * ld + c0 + tx] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty + 4 * i;
            dst[(size_t)c * ldd + e0 + tx] = tile[tx][ty + 4 * i];
        }
        if (ty == 0) {                   // column tx: running sum (rows ascending) and maximum
#pragma unroll 16
            for (int r = 0; r < 64; ++r) {
                const float v = tile[r][tx];
                sacc += v;
                cm = max(cm, __float_as_uint(v) & 0x7fffffffu);
            }
        } else if (ty == 1) {            // row tx: maximum over the tile's 64 columns
            unsigned m = 0u;
#pragma unroll 16
            for (int c = 0; c < 64; ++c) m = max(m, __float_as_uint(tile[tx][c]) & 0x7fffffffu);
            rowpart[(size_t)blockIdx.y * ldd + e0 + tx] = m;
        }
        __syncthreads();
    }
    if (ty == 0) {
        csum_part[(size_t)blockIdx.x * n_out + c0 + tx] = sacc;
        if (cm) atomicMax(colbits + c0 + tx, cm);
    }
}
// Column statistics from per-tile partials (a producer that had the dU tiles in registers, gpde_edge_bwd3.hip): a thread walks
// one column over the tiles of a strip (at most 64 strips: the second level is one thread per column walking the strips) in
// tile order -> the strip's column sum (csum_part[strip][col], reduced in strip order by the caller) and the column maximum
// (one atomicMax per strip)
__global__ __launch_bounds__(256) void k_tile_col_reduce(const float* __restrict__ tcs, const unsigned* __restrict__ tcm, int ntile,
                                                         int tiles_per_strip, int n_out, float* __restrict__ csum_part,
                                                         unsigned* __restrict__ colbits) {
    const int col = blockIdx.y * 256 + threadIdx.x, strip = blockIdx.x;
    if (col >= n_out) return;
    const int t1 = min((strip + 1) * tiles_per_strip, ntile);
    float sacc = 0.f;
    unsigned cm = 0u;
    for (int t = strip * tiles_per_strip; t < t1; ++t) {
        sacc += tcs[(size_t)t * n_out + col];
        cm = max(cm, tcm[(size_t)t * n_out + col]);
    }
    csum_part[(size_t)strip * n_out + col] = sacc;
    if (cm) atomicMax(colbits + col, cm);
}
__global__ void k_row_scales_from_parts(const unsigned* __restrict__ rowpart, int nparts, int ldd, int rows, float* __restrict__ sc,
                                        float* __restrict__ isc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows) return;
    unsigned m = 0u;
    for (int p = 0; p < nparts; ++p) m = max(m, rowpart[(size_t)p * ldd + e]);
    const int eb = (int)((m >> 23) & 0xff);
    const bool ok = eb >= 20 && eb <= 230;
    sc[e] = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    isc[e] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
}
// The split tile image of gpde_pack.hip's pack_w2_f16split_kernel for B[n][k] = H[k][n] * sc[n] (H row-major [rows][ld],
// k = the edge): workgroup = one 16 KiB tile (slice of 128 n, chunk of 32 k), thread = (n, k16 step m).
__global__ __launch_bounds__(256) void k_pack_split_kn(const float* __restrict__ H, int rows, int ld,
                                                       const float* __restrict__ sc, int nkct,
                                                       _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 img[128 * 64];      // the 16 KiB tile, final layout
    const int n = threadIdx.x & 127, m = threadIdx.x >> 7;
    const int kcn = blockIdx.x, slice = blockIdx.y;
    const int e0 = kcn * 32 + 16 * m;
    const float s = sc[slice * 128 + n];
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = (e0 + k < rows) ? H[(size_t)(e0 + k) * ld + slice * 128 + n] * s : 0.f;
    const int sw = (n >> 1) & 7;
    _Float16* row = img + n * 64;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = w[8 * (j >> 2) + 4 * hh + (j & 3)];
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        *(h8*)(row + (((m * 2 + hh) ^ sw) << 3)) = hi;
        *(h8*)(row + (((4 + m * 2 + hh) ^ sw) << 3)) = lo;
    }
    __syncthreads();
    // the image leaves in full lines: a row's units written straight from the registers were 16-byte pieces 128 bytes
    // apart (2.5 TB/s for read + write); 256 threads x 4 x 16 bytes, consecutive lanes consecutive units
    h8* dst = (h8*)(out + ((size_t)slice * (size_t)nkct + kcn) * (128 * 64));
    const h8* srcl = (const h8*)img;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * 256 + threadIdx.x] = srcl[q * 256 + threadIdx.x];
}
// ---- the first hidden layer as an operand generator (GpdeFirstLayerSpec) ------------------------------------------
// amax[d] = max over rows of |H0[row][d]| (d < 8) as fp32 bit patterns (zeroed by the caller)
__global__ __launch_bounds__(256) void k_attr_absmax8(const float* __restrict__ H0, int rows, int ld0, unsigned* __restrict__ amax) {
    const int d = threadIdx.x & 7;
    unsigned m = 0;
    for (int r = blockIdx.x * 32 + (threadIdx.x >> 3); r < rows; r += gridDim.x * 32)
        m = max(m, __float_as_uint(H0[(size_t)r * ld0 + d]) & 0x7fffffffu);
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) < 8 && m) atomicMax(amax + d, m);
}
// column scales from the a-priori bound H[e][n] <= |bp[n]| + sum_d |Wp[n][d]| * amax[d]: sc = 2^(13 - E(bound)), usc = 1 / sc
__global__ void k_first_layer_scales(const float* __restrict__ Wp, int ldw, const float* __restrict__ bp,
                                     const unsigned* __restrict__ amax, int n, float* __restrict__ sc, float* __restrict__ usc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float b = fabsf(bp[i]);
#pragma unroll
    for (int d = 0; d < 8; ++d) b = fmaf(fabsf(Wp[(size_t)i * ldw + d]), __uint_as_float(amax[d]), b);
    const int eb = (int)((__float_as_uint(b) >> 23) & 0xff);
    const bool ok = eb >= 20 && eb <= 230;
    sc[i] = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    usc[i] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
}
// Operands of the IN-KERNEL first layer (GpdeGemmF16sArgs::fl_mode): per input slot d < k0 the attribute scale alpha[d] =
// 2^-E(amax[d]) (|attr'| < 2), beta = 1 in the bias slot k0 (alpha = 0 there and beyond); per column n the scale sc[n] of
// k_first_layer_scales and the split image of w'[n][d] = Wp[n][d] * sc[n] / alpha[d] (d < k0), w'[n][k0] = bp[n] * sc[n]:
// |w'[n][d] * attr'[d]| <= bound * sc < 2^14, |w'| < 2^15 - inside the f16 range; all scales are powers of two (exact).
__global__ void k_first_layer_wimg(const float* __restrict__ Wp, int ldw, const float* __restrict__ bp, const unsigned* __restrict__ amax,
                                   int k0, int n, float* __restrict__ sc, float* __restrict__ usc, _Float16* __restrict__ wimg,
                                   float* __restrict__ alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float al[8], ial[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int eb = (int)((amax[d] >> 23) & 0xff);
        const bool ok = d < k0 && eb >= 20 && eb <= 230;
        al[d] = d < k0 ? (ok ? __int_as_float((254 - eb) << 23) : 1.f) : 0.f;           // 2^-(eb - 127)
        ial[d] = d < k0 ? (ok ? __int_as_float(eb << 23) : 1.f) : 0.f;
    }
    if (i == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) { alpha[d] = al[d]; alpha[8 + d] = d == k0 ? 1.f : 0.f; }
    }
    if (i >= n) return;
    float b = fabsf(bp[i]);
#pragma unroll
    for (int d = 0; d < 8; ++d) b = fmaf(fabsf(Wp[(size_t)i * ldw + d]), __uint_as_float(amax[d]), b);
    const int eb = (int)((__float_as_uint(b) >> 23) & 0xff);
    const bool ok = eb >= 20 && eb <= 230;
    const float s = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    sc[i] = s;
    usc[i] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
    _Float16* o = wimg + (size_t)i * 16;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float w = d < k0 ? Wp[(size_t)i * ldw + d] * s * ial[d] : d == k0 ? bp[i] * s : 0.f;
        const _Float16 hi = (_Float16)w;             // (rn here, rtz in the kernels' operand conversions: any split hi + lo = w to 2^-22 serves)
        o[d] = hi;
        o[8 + d] = (_Float16)(w - (float)hi);
    }
}
// A operands of the in-kernel first layer, once per edge: out[e] = [a_hi(8) | a_lo(8)] f16 with a[d] = attr[e][d] * alpha[d] + beta[d]
// (beta = 1 in the bias slot), hi = rtz16, lo = rn16(a - hi); rows beyond `rows` (the K padding of the split GEMM) are zero
__global__ __launch_bounds__(256) void k_first_layer_aops(const float* __restrict__ H0, int ld0, int rows, int epad, const float* __restrict__ alpha,
                                                          _Float16* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= epad) return;
    h8 hi = {}, lo = {};
    if (e < rows) {
        const f32x4 v0 = *(const f32x4*)(H0 + (size_t)e * ld0), v1 = *(const f32x4*)(H0 + (size_t)e * ld0 + 4);
#pragma unroll
        for (int d = 0; d < 8; d += 2) {
            const float q0 = fmaf(d < 4 ? v0[d & 3] : v1[d & 3], alpha[d], alpha[8 + d]);
            const float q1 = fmaf(d + 1 < 4 ? v0[(d + 1) & 3] : v1[(d + 1) & 3], alpha[d + 1], alpha[8 + d + 1]);
            const auto pk = __builtin_amdgcn_cvt_pkrtz(q0, q1);
            hi[d] = pk[0]; hi[d + 1] = pk[1];
            lo[d] = (_Float16)(q0 - (float)pk[0]); lo[d + 1] = (_Float16)(q1 - (float)pk[1]);
        }
    }
    *(h8*)(out + (size_t)e * 16) = hi;
    *(h8*)(out + (size_t)e * 16 + 8) = lo;
}
// k_pack_split_kn with H computed on the fly: B[n][k = edge] = relu(bp[n] + sum_d Wp[n][d] H0[edge][d]) * sc[n], plus
// the ReLU mask bits [rows][n_in / 32].  Workgroup = one 16 KiB tile (slice of 128 n, chunk of 32 edges), thread =
// (n, k16 step m).  The fp32 fmaf chain is the one k_first_layer / the fp32 GEMM path evaluate (d ascending from the bias).
constexpr int FLP_TILES = 8;            // 32-edge tiles per workgroup: the column's weights / bias / scale are loaded once for all
__global__ __launch_bounds__(256) void k_first_layer_pack(GpdeFirstLayerSpec f, int rows, int n_in, const float* __restrict__ sc,
                                                          int nkct, _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 img[128 * 64];      // the 16 KiB tile, final layout
    __shared__ __attribute__((aligned(16))) float h0s[32][8];
    const int n = threadIdx.x & 127, m = threadIdx.x >> 7;
    const int slice = blockIdx.y;
    const int col = slice * 128 + n;
    const float s = sc[col], b = f.bp[col];
    float wd[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) wd[d] = f.Wp[(size_t)col * f.ldw + d];
    const int sw = (n >> 1) & 7;
    _Float16* row = img + n * 64;
    // the attributes of the NEXT tile are requested before this tile's arithmetic (a workgroup's tiles used to run load ->
    // barrier -> compute -> barrier -> store -> barrier back to back: 9.1 ms per backward at s=121 for 24 GB of image)
    auto load_h0 = [&](int kc) {
        const int e = kc * 32 + (threadIdx.x >> 3), d = threadIdx.x & 7;
        return e < rows ? f.H0[(size_t)e * f.ld0 + d] : 0.f;
    };
    const int kc_end = min((int)(blockIdx.x + 1) * FLP_TILES, nkct);
    float h0n = blockIdx.x * FLP_TILES < kc_end ? load_h0(blockIdx.x * FLP_TILES) : 0.f;
    for (int kcn = blockIdx.x * FLP_TILES; kcn < kc_end; ++kcn) {
        h0s[threadIdx.x >> 3][threadIdx.x & 7] = h0n;
        __syncthreads();
        if (kcn + 1 < kc_end) h0n = load_h0(kcn + 1);
        const int e0 = kcn * 32 + 16 * m;
        float w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float t = b;
#pragma unroll
            for (int d = 0; d < 8; ++d) t = fmaf(wd[d], h0s[16 * m + k][d], t);
            t = fmaxf(t, 0.f);
            // mask bits of edge e0 + k: one 64-bit ballot per wave = columns n & ~63 .. +63 -> two words
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(t > 0.f);
            if ((threadIdx.x & 63) == 0 && e0 + k < rows) {
                uint32_t* mp = f.maskbits + (size_t)(e0 + k) * (n_in / 32) + slice * 4 + ((n >> 6) << 1);
                mp[0] = (uint32_t)bal;
                mp[1] = (uint32_t)(bal >> 32);
            }
            w[k] = (e0 + k < rows) ? t * s : 0.f;
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            h8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = w[8 * (j >> 2) + 4 * hh + (j & 3)];
                hi[j] = (_Float16)v;
                lo[j] = (_Float16)(v - (float)hi[j]);
            }
            *(h8*)(row + (((m * 2 + hh) ^ sw) << 3)) = hi;
            *(h8*)(row + (((4 + m * 2 + hh) ^ sw) << 3)) = lo;
        }
        __syncthreads();
        h8* dst = (h8*)(out + ((size_t)slice * (size_t)nkct + kcn) * (128 * 64));
        const h8* srcl = (const h8*)img;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * 256 + threadIdx.x] = srcl[q * 256 + threadIdx.x];
        __syncthreads();                     // img / h0s are rewritten by the next tile
    }
}
// The ReLU mask of the first hidden layer as bits [rows][n_in / 32], from the exact fp32 chain (k_first_layer's: bias, then d ascending) -
// k_first_layer_pack without its image: what is left of that kernel when the dW_2 GEMM generates H_1 itself (fl_mode 1).
__global__ __launch_bounds__(256) void k_first_layer_maskbits(GpdeFirstLayerSpec f, int rows, int n_in) {
    // A wave owns 8 of the tile's 32 rows and all 128 columns of the slice: lane l computes columns l and l + 64 as one float pair
    // (v_pk_fma_f32: the same bias-then-d-ascending chain per column, two columns per instruction), two ballots give the row's four
    // mask words, lane 0 stores them as 16 bytes.  (Round 6 first had one column per thread: twice the LDS reads, FMA and store
    // instructions per row - 3.2 ms per s=121 backward where the FMA count allows 0.6.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float h0s[2][32][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int slice = blockIdx.y;
    const int c0 = slice * 128 + lane, c1 = c0 + 64;
    const f32x2 b = f32x2{f.bp[c0], f.bp[c1]};
    f32x2 wd[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) wd[d] = f32x2{f.Wp[(size_t)c0 * f.ldw + d], f.Wp[(size_t)c1 * f.ldw + d]};
    const int nkct = (rows + 31) / 32;
    const int kc_end = min((int)(blockIdx.x + 1) * FLP_TILES, nkct);
    auto load_h0 = [&](int kc) {
        const int e = kc * 32 + (threadIdx.x >> 3), d = threadIdx.x & 7;
        return e < rows ? f.H0[(size_t)e * f.ld0 + d] : 0.f;
    };
    int buf = 0;
    float h0n = blockIdx.x * FLP_TILES < kc_end ? load_h0(blockIdx.x * FLP_TILES) : 0.f;
    for (int kcn = blockIdx.x * FLP_TILES; kcn < kc_end; ++kcn, buf ^= 1) {
        h0s[buf][threadIdx.x >> 3][threadIdx.x & 7] = h0n;
        __syncthreads();                     // (double-buffered: the next tile's store cannot overtake this tile's readers)
        if (kcn + 1 < kc_end) h0n = load_h0(kcn + 1);
        const int e0 = kcn * 32 + 8 * w;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 ha = *(const f32x4*)&h0s[buf][8 * w + k][0], hb = *(const f32x4*)&h0s[buf][8 * w + k][4];
            f32x2 t = b;
#pragma unroll
            for (int d = 0; d < 4; ++d) t = __builtin_elementwise_fma(wd[d], f32x2{ha[d], ha[d]}, t);
#pragma unroll
            for (int d = 0; d < 4; ++d) t = __builtin_elementwise_fma(wd[4 + d], f32x2{hb[d], hb[d]}, t);
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(t[0] > 0.f), b1 = __builtin_amdgcn_ballot_w64(t[1] > 0.f);
            if (lane == 0 && e0 + k < rows) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                *(u32x4*)(f.maskbits + (size_t)(e0 + k) * (n_in / 32) + slice * 4) =
                    u32x4{(uint32_t)b0, (uint32_t)(b0 >> 32), (uint32_t)b1, (uint32_t)(b1 >> 32)};
            }
        }
    }
}
}  // namespace

size_t gpde_gemm_f16s_tn_ws_floats(int rows_max, int n_out, int n_in, int ksplits) {
    size_t epad = (size_t)rows_max + 64 * (size_t)ksplits;
    if (epad < (size_t)256 * ksplits) epad = (size_t)256 * ksplits;       // the launcher pads every K split to >= 256 rows
    return epad * n_out + epad * n_in + 3 * (size_t)(n_out + n_in) + 64 +
           (size_t)n_in * 8 + 64 + epad * 8 +                                             // in-kernel first layer: column image + slot scales, [epad][16] f16 attribute operands
           (epad / 1024 + 2) * (size_t)n_out + (size_t)(n_out / 64 + 1) * epad + 64;   // GpdeDuStats: column-sum partials per
                                                                                          // 1024-row strip, row maxima per column block
}

float* gpde_gemm_f16s_tn_at(float* ws, int rows, int ksplits, int* ld) {
    int ksp = ((rows + 64 * ksplits - 1) / (64 * ksplits)) * 64;
    if (ksp < 256) ksp = 256;
    *ld = ksp * ksplits;
    return ws;
}

// part[s][n_out][n_in] (s < ksplits, stride n_out * n_in) = partial sums of dU^T . H over the K splits
int gpde_launch_gemm_f16s_tn(const float* dU, int ldu, int n_out, const float* H, int ldh, int n_in, int rows,
                             int ksplits, float* ws, float* part, hipStream_t stream, const unsigned* du_absmax_bits,
                             const GpdeFirstLayerSpec* fl, const GpdeDuStats* st_) {
    if (fl && (!fl->H0 || !fl->Wp || !fl->bp || !fl->maskbits || fl->ld0 < 8 || fl->ldw < 8)) {
        gpde_set_error("gpde_gemm_f16s_tn: incomplete first-layer spec");
        return GPDE_EINVAL;
    }
    if (fl) ldh = 4;
    if (rows < 1 || n_out % 64 != 0 || n_in % GP_TN != 0 || ldu % 4 != 0 || ldh % 4 != 0 || ksplits < 1) {
        gpde_set_error("gpde_gemm_f16s_tn: unsupported shape rows=%d n_out=%d n_in=%d", rows, n_out, n_in);
        return GPDE_EUNSUPPORTED;
    }
    int ksp = ((rows + 64 * ksplits - 1) / (64 * ksplits)) * 64;       // K per split (a multiple of 64: even chunk count)
    if (ksp < 256) ksp = 256;
    const int epad = ksp * ksplits;
    float* At = ws;                                            // [n_out][epad]
    _Float16* Bimg = (_Float16*)(At + (size_t)n_out * epad);   // [n_in / 128][epad / 32][128][64]
    float* sca = (float*)(Bimg + (size_t)n_in * epad * 2);
    float* isca = sca + n_out;
    float* scb = isca + n_out;
    float* ucolb = scb + n_in;
    unsigned* bits = (unsigned*)(ucolb + n_in);                // [n_out + n_in]
    const int nstrip = (epad + TS_STRIP - 1) / TS_STRIP;
    float* csum_part = (float*)(bits + n_out + n_in) + 16;    // [nstrip][n_out]   (GpdeDuStats)
    unsigned* rowpart = (unsigned*)(csum_part + (size_t)nstrip * n_out);        // [n_out / 64][epad]
    float* flimg = (float*)(rowpart + (size_t)(n_out / 64 + 1) * epad) + 16;    // [n_in][16] f16 image, then 16 floats (in-kernel first layer)
    float* flaops = flimg + (size_t)n_in * 8 + 32;                              // [epad][16] f16 attribute operands
    GP_HIP_CHECK(gpde_zero_async(bits, (size_t)(n_out + n_in) * 4, stream));
    int splits = 1;
    while (splits < 256 && rows / (splits * 2) >= 64) splits *= 2;
    // column maxima of dU: given by the caller when another pass over dU has already collected them (k_colsum)
    if (st_ && st_->tile_csum) {
        // the producer of dU wrote the transposed copy and the row scales itself: zero the K padding, fold its tile partials
        if (epad > rows) GP_HIP_CHECK(gpde_zero2d_async(At + rows, (size_t)epad * 4, (size_t)(epad - rows) * 4, (size_t)n_out, stream));
        const int ntile = (rows + 31) / 32;
        const int ns2 = nstrip < 64 ? nstrip : 64, tps = (ntile + ns2 - 1) / ns2;
        hipLaunchKernelGGL(k_tile_col_reduce, dim3(ns2, (n_out + 255) / 256), dim3(256), 0, stream, st_->tile_csum, st_->tile_cmax, ntile,
                           tps, n_out, csum_part, bits);
        if (int rc = gpde_launch_reduce_splits(csum_part, (size_t)n_out, ns2, (size_t)n_out, st_->db_accumulate, 1, stream)) return rc;
        hipLaunchKernelGGL(k_scales_from_max, dim3((n_out + 255) / 256), dim3(256), 0, stream, bits, n_out, sca, isca);
    } else if (st_) {
        // ONE pass over dU: transposed copy, column sums (-> bias gradient), column maxima (-> scales of the A rows here),
        // row maxima (-> row scales of the dU . W^T GEMM that follows)
        hipLaunchKernelGGL(k_transpose_stats, dim3(nstrip, n_out / 64), dim3(256), 0, stream, dU, rows, ldu, At, epad, n_out,
                           csum_part, bits, rowpart);
        if (int rc = gpde_launch_reduce_splits(csum_part, (size_t)n_out, nstrip, (size_t)n_out, st_->db_accumulate, 1, stream)) return rc;
        hipLaunchKernelGGL(k_row_scales_from_parts, dim3((rows + 255) / 256), dim3(256), 0, stream, rowpart, n_out / 64, epad, rows,
                           st_->row_sc, st_->row_isc);
        hipLaunchKernelGGL(k_scales_from_max, dim3((n_out + 255) / 256), dim3(256), 0, stream, bits, n_out, sca, isca);
    } else {
        if (du_absmax_bits) GP_HIP_CHECK(gpde_copy_async(bits, du_absmax_bits, (size_t)n_out * 4, stream));
        else hipLaunchKernelGGL(k_colabsmax, dim3((n_out + 255) / 256, splits), dim3(256), 0, stream, dU, rows, n_out, ldu, splits, bits);
        hipLaunchKernelGGL(k_scales_from_max, dim3((n_out + 255) / 256), dim3(256), 0, stream, bits, n_out, sca, isca);
        hipLaunchKernelGGL(k_transpose_pad, dim3(epad / 64, n_out / 64), dim3(256), 0, stream, dU, rows, ldu, At, epad);
    }
    const bool fl_gen = fl && gpde_first_layer_in_kernel(*fl, rows, ksplits);
    if (fl) {
        int nb = (rows + 31) / 32; if (nb > 2048) nb = 2048;
        if (fl_gen && fl->amax_bits) GP_HIP_CHECK(gpde_copy_async(bits + n_out, fl->amax_bits, 8 * 4, stream));     // the call-wide bound (see GpdeFirstLayerSpec)
        else hipLaunchKernelGGL(k_attr_absmax8, dim3(nb), dim3(256), 0, stream, fl->H0, rows, fl->ld0, bits + n_out);
        if (fl_gen) {
            // round 6: no image - the GEMM below generates its B chunks from the attributes; the ReLU mask bits of the dU_1 GEMM come
            // from the exact fp32 chain in a pass of their own (k_first_layer_pack minus its 4 bytes per edge and column of image)
            hipLaunchKernelGGL(k_first_layer_wimg, dim3((n_in + 255) / 256), dim3(256), 0, stream, fl->Wp, fl->ldw, fl->bp, bits + n_out,
                               fl->k0, n_in, scb, ucolb, (_Float16*)flimg, flimg + (size_t)n_in * 8);
            hipLaunchKernelGGL(k_first_layer_aops, dim3((epad + 255) / 256), dim3(256), 0, stream, fl->H0, fl->ld0, rows, epad, flimg + (size_t)n_in * 8,
                               (_Float16*)flaops);
            hipLaunchKernelGGL(k_first_layer_maskbits, dim3(((rows + 31) / 32 + FLP_TILES - 1) / FLP_TILES, n_in / 128), dim3(256), 0, stream, *fl, rows, n_in);
        } else {
            hipLaunchKernelGGL(k_first_layer_scales, dim3((n_in + 255) / 256), dim3(256), 0, stream, fl->Wp, fl->ldw, fl->bp,
                               bits + n_out, n_in, scb, ucolb);
            hipLaunchKernelGGL(k_first_layer_pack, dim3((epad / 32 + FLP_TILES - 1) / FLP_TILES, n_in / 128), dim3(256), 0, stream, *fl, rows, n_in, scb, epad / 32, Bimg);
        }
    } else {
        hipLaunchKernelGGL(k_colabsmax, dim3((n_in + 255) / 256, splits), dim3(256), 0, stream, H, rows, n_in, ldh, splits, bits + n_out);
        hipLaunchKernelGGL(k_scales_from_max, dim3((n_in + 255) / 256), dim3(256), 0, stream, bits + n_out, n_in, scb, ucolb);
        hipLaunchKernelGGL(k_pack_split_kn, dim3(epad / 32, n_in / 128), dim3(256), 0, stream, H, rows, ldh, scb, epad / 32, Bimg);
    }
    GP_LAUNCH_CHECK("gpde_gemm_f16s_tn operand kernels");
    GpdeGemmF16sArgs g{};
    g.A = At; g.lda = epad; g.M = n_out; g.bsplit = Bimg; g.ucol = ucolb; g.mask = nullptr; g.ldmask = 0;
    g.C = part; g.ldc = n_in; g.K = epad; g.N = n_in; g.sc = sca; g.isc = isca;
    g.ksplits = ksplits; g.cstride = (size_t)n_out * n_in;
    if (fl_gen) {
        g.fl_mode = 1; g.fl_attr = flaops; g.fl_ld0 = 8; g.fl_rows = epad;          // (the operand image: 8 floats = 16 halves per edge, K padding included)
        g.fl_wimg = flimg; g.fl_alpha = flimg + (size_t)n_in * 8;
    }
    return gpde_launch_gemm_f16s_nt(g, nullptr, stream);
}

// Whether gpde_launch_gemm_f16s_tn generates the first hidden layer inside its GEMM for this spec (`maskbits` receives the mask words
// either way): a free slot for the bias, 16-byte aligned attribute rows,
// the split-K form (ksplits > 1), and not switched off (GPDE_BWD_H1_IMAGE=1: rounds 3-5's image + mask bits, A/B)
bool gpde_first_layer_in_kernel(const GpdeFirstLayerSpec& f, int rows, int ksplits) {
    return f.k0 >= 1 && f.k0 <= 7 && f.ld0 >= 8 && f.ld0 % 4 == 0 && rows >= 1 && ksplits > 1 && !gpde_switches().bwd_h1_image;     // (a K split is >= 256 edges = 8 chunks)
}

// ---- gather form: operands and launcher (depth-deferred backward, gpde_bwd.hip) ------------------------------------------
namespace {
constexpr int GT_ROWS = NW * TE;        // slots per workgroup tile

// bits[node] = max |dZ[node][..]| over one layer's [64][K2P] block as an fp32 bit pattern (bits zeroed by the caller; the
// maximum over layers accumulates): one workgroup per node
__global__ __launch_bounds__(256) void k_node_absmax(const float* __restrict__ dZ, int row_floats, unsigned* __restrict__ bits) {
    const float* p = dZ + (size_t)blockIdx.x * row_floats;
    unsigned m = 0;
    for (int i = threadIdx.x * 4; i < row_floats; i += 1024) {
        const f32x4 v = *(const f32x4*)(p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) m = max(m, __float_as_uint(v[j]) & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(bits + blockIdx.x, m);
}
// The split tile image (k_pack_split_kn's layout) of B_i[n][K = 64 l + c] = dZ^(l)[i][c][n] * sc[i] for one destination
// node i: workgroup = one 16 KiB tile (node, slice of 128 n, chunk of 32 K = half the channels of one layer), thread =
// (n, k16 step m).  Layers l >= L (K padding) are zero tiles.
__global__ __launch_bounds__(256) void k_pack_dz_image(const float* __restrict__ dZ, size_t dz_layer_stride, int L, int K2P,
                                                       const float* __restrict__ sc, int nkct, _Float16* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) _Float16 img[128 * 64];
    const int n = threadIdx.x & 127, m = threadIdx.x >> 7;
    const int kcn = blockIdx.x, slice = blockIdx.y, node = blockIdx.z;
    const int l = kcn >> 1, c0 = (kcn & 1) * 32 + 16 * m;
    const float s = sc[node];
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        w[k] = l < L ? dZ[(size_t)l * dz_layer_stride + ((size_t)node * GP_W + c0 + k) * K2P + slice * 128 + n] * s : 0.f;
    const int sw = (n >> 1) & 7;
    _Float16* row = img + n * 64;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        h8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = w[8 * (j >> 2) + 4 * hh + (j & 3)];
            hi[j] = (_Float16)v;
            lo[j] = (_Float16)(v - (float)hi[j]);
        }
        *(h8*)(row + (((m * 2 + hh) ^ sw) << 3)) = hi;
        *(h8*)(row + (((4 + m * 2 + hh) ^ sw) << 3)) = lo;
    }
    __syncthreads();
    const int ns = K2P / 128;
    h8* dst = (h8*)(out + (((size_t)node * ns + slice) * (size_t)nkct + kcn) * (128 * 64));
    const h8* srcl = (const h8*)img;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * 256 + threadIdx.x] = srcl[q * 256 + threadIdx.x];
}
// per source node j: sc = 2^(13 - E(max over the L layers and 64 channels of |xstack[l][j][c]|)), isc = 1 / sc
__global__ __launch_bounds__(256) void k_xstack_scales(const float* __restrict__ xs, size_t layer_stride, int L, int64_t n_nodes,
                                                       float* __restrict__ sc, float* __restrict__ isc) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n_nodes) return;
    unsigned m = 0;
    for (int l = 0; l < L; ++l) m = max(m, __float_as_uint(xs[(size_t)l * layer_stride + (size_t)j * GP_W + lane]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (lane == 0) {
        const int eb = (int)((m >> 23) & 0xff);
        const bool ok = eb >= 20 && eb <= 230;
        sc[j] = ok ? __int_as_float((267 - eb) << 23) : 1.f;
        isc[j] = ok ? __int_as_float((eb - 13) << 23) : 1.f;
    }
}
// tiles[t] = (chunk-local node, first row, end row, 0) for every run of <= 256 in-edges of the nodes na .. na + nn: one
// workgroup, nodes in batches of 1024 with an LDS scan (a chunk holds a few thousand nodes at most)
__global__ __launch_bounds__(1024) void k_tile_list(const int32_t* __restrict__ rowptr, int na, int nn, int e0, int32_t* __restrict__ tiles) {
    __shared__ int scan[1024];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nn; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        int r0 = 0, r1 = 0;
        if (i < nn) { r0 = rowptr[na + i] - e0; r1 = rowptr[na + i + 1] - e0; }
        const int cnt = (r1 - r0 + GT_ROWS - 1) / GT_ROWS;
        scan[threadIdx.x] = cnt;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = threadIdx.x >= o ? scan[threadIdx.x - o] : 0;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        const int first = base + scan[threadIdx.x] - cnt;
        for (int q = 0; q < cnt; ++q) {
            int32_t* t = tiles + 4 * (size_t)(first + q);
            t[0] = i; t[1] = r0 + q * GT_ROWS; t[2] = r1; t[3] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 1023) base += scan[1023];
        __syncthreads();
    }
}
}  // namespace

size_t gpde_dz_image_bytes_per_node(int Lp, int K2P) { return (size_t)(K2P / 128) * (2 * Lp) * TILE_B; }

int gpde_tile_count(const int32_t* rowptr_host, int na, int nb) {
    long long t = 0;
    for (int i = na; i < nb; ++i) t += (rowptr_host[i + 1] - rowptr_host[i] + GT_ROWS - 1) / GT_ROWS;
    return (int)t;
}

int gpde_launch_tile_list(const int32_t* rowptr, int na, int nn, int e0, int32_t* tiles, hipStream_t stream) {
    hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, stream, rowptr, na, nn, e0, tiles);
    GP_LAUNCH_CHECK("k_tile_list");
    return GPDE_OK;
}

int gpde_launch_xstack_scales(const float* xstack, size_t layer_stride, int L, int64_t n_nodes, float* sc, float* isc, hipStream_t stream) {
    if (n_nodes <= 0) return GPDE_OK;
    hipLaunchKernelGGL(k_xstack_scales, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, stream, xstack, layer_stride, L, n_nodes, sc, isc);
    GP_LAUNCH_CHECK("k_xstack_scales");
    return GPDE_OK;
}

int gpde_launch_dz_image(const float* dZ, size_t dz_layer_stride, int L, int Lp, int nn, int K2P, unsigned* bits, float* scale,
                         float* unscale, void* img, hipStream_t stream) {
    if (nn <= 0) return GPDE_OK;
    if (K2P % 128 != 0 || L < 1 || Lp < L || Lp % 2 != 0) { gpde_set_error("gpde_launch_dz_image: bad shape L=%d Lp=%d K2P=%d", L, Lp, K2P); return GPDE_EINVAL; }
    GP_HIP_CHECK(gpde_zero_async(bits, (size_t)nn * 4, stream));
    for (int l = 0; l < L; ++l)
        hipLaunchKernelGGL(k_node_absmax, dim3(nn), dim3(256), 0, stream, dZ + (size_t)l * dz_layer_stride, GP_W * K2P, bits);
    hipLaunchKernelGGL(k_scales_from_max, dim3((nn + 255) / 256), dim3(256), 0, stream, bits, nn, scale, unscale);
    hipLaunchKernelGGL(k_pack_dz_image, dim3(2 * Lp, K2P / 128, nn), dim3(256), 0, stream, dZ, dz_layer_stride, L, K2P, scale,
                       2 * Lp, (_Float16*)img);
    GP_LAUNCH_CHECK("gpde_launch_dz_image kernels");
    return GPDE_OK;
}

int gpde_launch_gemm_f16s_gather(const float* xstack, size_t layer_stride, int Lp, const float* sc_src, const float* isc_src,
                                 const int32_t* src_rows, int rows, const int32_t* tiles, int ntiles, const void* bimg,
                                 const float* unscale, const float* mask, int ldmask, float* C, int ldc, int N, hipStream_t stream) {
    if (rows <= 0 || ntiles <= 0) return GPDE_OK;
    const int K = GP_W * Lp;
    if (N % GP_TN != 0 || Lp < 4 || Lp % 2 != 0 || !mask) {
        gpde_set_error("gpde_gemm_f16s_gather: unsupported shape N=%d layers=%d", N, Lp);
        return GPDE_EUNSUPPORTED;
    }
    if (gp_overlap(C, ((size_t)(rows - 1) * ldc + N) * 4, mask, ((size_t)(rows - 1) * ldmask + N) * 4)) {
        gpde_set_error("gpde_gemm_f16s_gather: output overlaps the mask (internal buffer plan error)");
        return GPDE_EINVAL;
    }
    GpdeGemmF16sArgs a{};
    a.A = xstack; a.lda = GP_W; a.M = rows; a.bsplit = bimg; a.ucol = nullptr; a.mask = mask; a.ldmask = ldmask;
    a.C = C; a.ldc = ldc; a.K = K; a.N = N; a.sc = sc_src; a.isc = isc_src; a.ksplits = 1; a.cstride = 0;
    a.skew_us = gpde_debug_skew_us();
    a.g_src = src_rows; a.g_tile = tiles; a.g_ntiles = ntiles; a.g_layer_stride = layer_stride;
    a.g_bnode_bytes = gpde_dz_image_bytes_per_node(Lp, N); a.g_unscale = unscale;
    const int ns = N / GP_TN;
    int groups = gpde_num_cus() / ns;
    if (groups < 1) groups = 1;
    if (groups > ntiles) groups = ntiles;
    if (groups >= 8) groups = groups / 8 * 8;                    // whole XCD rounds: consecutive tiles share an L2
    a.n_groups = groups;
    const size_t lds = (size_t)NS * TILE_B + (size_t)NS * A_SLOT + NW * TE * 4 + 64;
    static GpdeLdsOnce once;
    if (int rc = once.ensure(gpde_gemm_f16s_nt_kernel<true>)) return rc;
    hipLaunchKernelGGL(gpde_gemm_f16s_nt_kernel<true>, dim3(groups * ns), dim3(256), lds, stream, a);
    GP_LAUNCH_CHECK("gpde_gemm_f16s_nt_kernel<gather>");
    return GPDE_OK;
}
