// Shared declarations for libgpde.so (gfx950 only; no CUDA path, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gpde.h"
#include <atomic>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- tile constants of the fused kernel ---------------------------------------
constexpr int GP_W = GPDE_WIDTH;   // node feature width (in = out = 64)
constexpr int GP_TE = 32;          // edges per wave tile (one MFMA row block)
constexpr int GP_TN = 128;         // hidden columns per workgroup slice (4 MFMA col blocks)
constexpr int GP_BK = 32;          // k1 chunk streamed through LDS per step
constexpr int GP_WAVES = 4;        // waves per workgroup, one per SIMD
constexpr int GP_BS_STRIDE = 36;   // LDS row stride (floats) of a [128][32] W2 tile: 144 B rows
                                   // -> conflict-free ds_read_b128 across a 16-lane group

static inline int gp_round_up(int v, int m) { return (v + m - 1) / m * m; }

// byte ranges [a, a + na) and [b, b + nb) intersect (the GEMM launchers refuse an output that overlaps an operand:
// their column-slice workgroups read the same operand rows at different times)
static inline bool gp_overlap(const void* a, size_t na, const void* b, size_t nb) {
    if (!a || !b || !na || !nb) return false;
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

// ---- packed MLP layout (all offsets in floats from the start of `packed`) --------------------
//  mode 0: n_layers == 2            H = relu(L1(attr))                      (K2 = k1)
//  mode 1: n_layers == 3            H = relu(L2(relu(L1(attr))))            (K1 = k1, K2 = k2)
//  mode 2: n_layers >= 4 or k0 > 7  front layers run as plain dense layers, H read from memory
struct GpdePackLayout {
    int mode;
    int n_layers;
    int k0;          // edge-attribute width
    int k1, K1P;     // first hidden width and its padding to GP_BK      (mode 1)
    int k2, K2P;     // last hidden width (input of the last Linear) and its padding to GP_TN
    size_t off_w1;   // [rows][2][4]  rows = K1P (mode 1) or K2P (mode 0): W1 with b1 folded in;
                     // mode 1 appends one row: max_k |W1b[k][d]| per input slot d (same [2][4] order)
    size_t off_w2t;  // [K2P/128][K1P/32][128][32]  W2 in LDS-tile order        (mode 1)
    size_t off_b2;   // [K2P]                                                     (mode 1)
    size_t off_w2h;  // f16-split W2 tiles [K2P/128][K1P/32][128][2 parts][32] halves, rows scaled
                     // by 2^t_n (mode 1); same byte size as off_w2t
    size_t off_ucol; // [K2P] 2^-t_n per hidden column                            (mode 1)
    size_t off_w1h;  // [K1P][hi 8 halves | lo 8 halves]: (W1|b1) columns scaled by 2^u_d, f16 split (mode 1)
    size_t off_fcol; // [8] 2^-u_d per input slot (0 for an all-zero column)      (mode 1)
    size_t off_w3q;  // [64 c][K2P/4][64 o][4]  last Linear, re-associated order
    size_t off_b3;   // [64 c][64 o]            last Linear bias as a 64x64 matrix (zeros if none)
    size_t off_front;// mode 2: per front layer l: W [KP(l+1)][KP(l)] zero padded, then b [KP(l+1)]
    int frontKP[GPDE_MAX_LAYERS + 1];
    size_t off_front_w[GPDE_MAX_LAYERS];
    size_t off_front_b[GPDE_MAX_LAYERS];
    size_t off_w3s;  // split tile image of W3 [4096][K2P] (rows n = c*64+o, gpde_pack.hip W2 layout) for the per-edge last
    size_t off_ucol3;// layer of low in-degree graphs (gpde_launch_edge_messages) + its [4096] row un-scales; K2P >= 256 only
    int has_w3s;
    size_t total_floats;
    int has_b3;
};

int gpde_pack_layout(int n_layers, const int32_t* dims, GpdePackLayout* L);

void gpde_set_error(const char* fmt, ...);

// Zero-fill / device-to-device copy as KERNELS of this library (gpde_api.hip), instead of hipMemsetAsync / hipMemcpyAsync:
// on ROCm 7.0 a hipMemsetAsync of >= 256 bytes recorded into a HIP graph replays correctly ONCE and writes garbage from the
// second replay on (scripts/dbg_memset_graph.py; found through gp.capture of a training step, round 5) - and nothing here
// should be the reason a caller cannot record the operator.  Byte counts must be multiples of 4 (every buffer here is).
hipError_t gpde_zero_async(void* p, size_t bytes, hipStream_t st);
hipError_t gpde_zero2d_async(void* p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t st);
hipError_t gpde_copy_async(void* dst, const void* src, size_t bytes, hipStream_t st);

// Developer / A-B switches (GPDE_* environment variables consumed by native code).  They are read ONCE - on the first native
// call of the process - and never inside a launch path; gpde_reload_switches() (include/gpde.h) re-reads them (the test
// suite's monkeypatch fixture calls it, tests/conftest.py).
struct GpdeSwitches {
    bool bwd_gemm_f32;           // GPDE_BWD_GEMM_F32: dU_1 / dW_2 on the fp32-MFMA GEMMs
    bool bwd_dw2_f32;            // GPDE_BWD_DW2_F32
    bool bwd_recompute_f32;      // GPDE_BWD_RECOMPUTE_F32: H recomputed by fp32 GEMMs in the full backward
    bool bwd_h1_materialize;     // GPDE_BWD_H1_MATERIALIZE: round-2 plan (H_1 written)
    bool bwd_h1_gemm;            // GPDE_BWD_H1_GEMM
    bool bwd_node_terms_gemm;    // GPDE_BWD_NODE_TERMS_GEMM: dx += g root^T, droot, dbias on the generic GEMM / column-sum launches (rounds 2-5) - A/B
    bool bwd_zagg_f32;           // GPDE_BWD_ZAGG_F32: the backward's Z re-aggregation on fp32 MFMA (gpde_zagg_kernel<false>, rounds 2-5) - A/B
    bool bwd_dw1_pass;           // GPDE_BWD_DW1_PASS: dW_1 / db_1 from k_dw_first's pass over a materialised dU_1 (rounds 2-5) - A/B
    bool bwd_h1_image;           // GPDE_BWD_H1_IMAGE: rounds 3-5 plan (H_1^T split image + mask bits written by k_first_layer_pack) - A/B
    bool bwd_dw1_gemm;           // GPDE_BWD_DW1_GEMM
    bool bwd_du_passes;          // GPDE_BWD_DU_PASSES: separate bias / maxima / transpose passes over dU_2
    bool bwd_du_transpose_pass;  // GPDE_BWD_DU_TRANSPOSE_PASS: k_transpose_stats instead of the per-edge kernel's by-products
    bool bwd_one_pass;           // GPDE_BWD_ONE_PASS: the one-pass kernel (gpde_fused_f16v6_kernel<2>, round 5; measured slower than
                                 // recompute-store + gpde_edge_bwd3: opt-in) where its conditions hold; also GPDE_EDGE_BWD=4
    bool store_v3;               // GPDE_STORE_V3: hidden-activation store on the 8-wave kernel
    bool nt_no_prefetch;         // GPDE_NT_NO_PREFETCH
    bool tn_no_ks_xcd;           // GPDE_TN_NO_KS_XCD
    int edge_bwd;                // GPDE_EDGE_BWD = 1 | 2 | 3: force a per-edge backward kernel (0: by in-degree)
    int debug_skew_us;           // GPDE_DEBUG_SKEW_US: odd column slices of the GEMMs start late (race tests)
};
const GpdeSwitches& gpde_switches();

#define GP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            gpde_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                           __FILE__, __LINE__);                                         \
            return GPDE_EHIP;                                                           \
        }                                                                               \
    } while (0)

#define GP_LAUNCH_CHECK(name)                                                           \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            gpde_set_error("launch of %s failed: %s", name, hipGetErrorString(_e));     \
            return GPDE_EHIP;                                                           \
        }                                                                               \
    } while (0)

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; a process-wide "done" flag
// breaks a second GPU in the same process).  Thread-safe: the worst case is two threads setting
// the same value.  The limit is set to the whole 160 KiB so that later, larger requests need no second call.
struct GpdeLdsOnce {
    std::atomic<unsigned> done{0};       // bit d: device d has the attribute
    template <typename... Fn>
    int ensure(Fn... fns) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
        if (done.load(std::memory_order_acquire) & (1u << dev)) return GPDE_OK;
        const void* fl[] = {(const void*)fns...};
        for (const void* f : fl) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) {
                gpde_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
                return GPDE_EHIP;
            }
        }
        done.fetch_or(1u << dev, std::memory_order_release);
        return GPDE_OK;
    }
};

// ---- kernel launchers (one translation unit each) ---------------------------------------------
struct GpdeFusedArgs {
    const float* x;        // [N][64]
    const float* attr;     // [E][k0], original edge order
    const int32_t* rowptr; // [N+1]
    const int32_t* src;    // [E] CSR order
    const int32_t* dst;    // [E] CSR order
    const int32_t* perm;   // [E] CSR slot -> original edge id
    const float* w1;       // packed, see GpdePackLayout
    const float* w2t;
    const float* b2;
    const void* w2h;       // f16-split W2 tiles (mode 1, GPDE_FWD_F16SPLIT)
    const float* ucol;     // [K2P] 2^-t_n
    const void* w1h;       // f16-split (W1|b1) image, column-scaled
    const float* fcol;     // [8] per-input-slot 2^-u_d
    const float* hbuf;     // mode 2: [edges of chunk][K2P] in CSR order, relu already applied
    float* zbuf;           // [nc1-nc0][64][K2P]
    float* hout;           // f16v3 only: when set, write the hidden activations [slot - e_chunk0][K2P]
                           // instead of aggregating (gpde_hidden_fwd)
    const unsigned* xs;    // f16v3, f16-split aggregation: x as (lo16 << 16 | hi16) words, globally scaled
    const unsigned* scal;  // [0] bits of max |x|, [1] bits of max_e B_e   (gpde_prep.hip)
    const unsigned* hmax;  // zagg f16: bits of max |H| (recorded by gpde_hidden_fwd)
    unsigned* hmax_out;    // f16v3 WRITE_H: atomicMax target for the bits of max |H| (nullable)
    const int32_t* blk;    // f16v6 work queue: node-aligned block bounds (edge offsets) [*qn + 1], or nullptr = static ranges
    const int32_t* qn;     //   number of blocks of this launch (device scalar, written by the block-bounds pre-pass)
    unsigned* qctr;        //   one draw counter per column slice, zeroed by the pre-pass
    int kt;                // f16v3, attributes from a node table (row f3): table row stride, 0 = edge_attr tensor
    int sel[8];            // slot d of an edge's attribute = attr[(sel[d] >> 8 ? dst : src) * kt + (sel[d] & 255)]
    int k0, K1P, K2P;
    int nc0, nc1;          // destination-node chunk
    int e_chunk0;          // rowptr[nc0] (mode 2: first row of hbuf)
    int n_groups;          // edge groups (workgroups per slice)
    // ---- the backward's one-pass mode (gpde_fused_f16v6_kernel<2>, gpde_launch_fused_bwd; round 5): the K loop of the store
    // variant with H_2^T left in the accumulators (lane = edge) and the two per-edge products of the backward through the
    // aggregation taken from there - H_2 is neither written nor read.  `xs` carries the fp32 x rows here ([N][64] floats),
    // `scal[1]` the bits of max_e B_e (gpde_launch_attr_bound).  Rows = CSR slots [e_chunk0, rowptr[nc1]) of the chunk.
    const void* bw_img1;   // [chunk node][K2P n][hi 64 c halves | lo 64 c halves]: dZ_i^T split image (gpde_launch_dz_images)
    const void* bw_img2;   // [chunk node][64 c][K2P/32 groups][hi 32 halves | lo 32 halves], in place over the fp32 dZ, halves
                           // ordered for the accumulator's row order (gpde_dz_img2_col)
    const float* bw_unscale;   // [chunk nodes] 2^-t of the node's images
    const float* bw_dS;        // [chunk nodes][64]
    float* bw_dU;          // [rows][K2P] dU_2 rows, or nullptr: dx only (the light pass of the depth-deferred backward)
    float* bw_dUt; int bw_ldt; // optional with bw_dU (all four by-products or none): transposed copy [K2P][ldt] (gpde_gemm_f16s_tn_at)
    float* bw_rowmax;      // with bw_dUt: [K2P/128][rows] max_n |dU[e][n]| over the slice's columns
    float* bw_csum; unsigned* bw_cmax;   // with bw_dUt: [ceil(rows/32)][K2P] per-32-slot tile column sums / column max bits
    float* bw_dxp;         // [K2P/128][rows][64] per-slice partial rows of dx_e (slice 0 carries dS_i); summed by k_dx_reduce
    int bw_rows;
};
int gpde_launch_fused(int mode, bool f16split, const GpdeFusedArgs& a, hipStream_t stream);
// pre-passes of the f16-split aggregation (gpde_prep.hip): scal[0..1], xs [n_nodes][64]
int gpde_launch_g2_prep(const float* x, int64_t n_nodes, const float* attr, int64_t n_edges, int k0,
                        const float* wmax8, unsigned* scal, unsigned* xs, hipStream_t stream,
                        int kt = 0, const int* sel = nullptr, const int32_t* src = nullptr,
                        const int32_t* dst = nullptr);
// block bounds + queue counters of the f16v6 work queue (gpde_prep.hip): blk [nblk_max + 1], qn [1], qctr [n_slices]
constexpr int GP_QBLOCK = 4096;    // edges per big block (node-aligned: a block holds whole destination nodes); the
                                   // last 1/8 of the edges is cut into blocks of GP_QBLOCK / 8
int gpde_launch_block_bounds(const int32_t* rowptr, int nc0, int nc1, int nblk_max, int32_t* blk, int32_t* qn,
                             unsigned* qctr, int n_slices, hipStream_t stream);
// 2^(13 - floor(log2 v)) for v in the normal range, else 1: puts a maximum v into [2^13, 2^14)
__host__ __device__ static inline float gpde_pow2_to_2p13(float v) {
    union { float f; unsigned u; } a; a.f = v;
    const int eb = (int)((a.u >> 23) & 0xff);
    if (eb < 20 || eb > 230) return 1.f;
    a.u = (unsigned)(267 - eb) << 23;
    return a.f;
}
// Z = sum x_j (x) H_e from given hidden activations a.hbuf (gpde_zagg.hip); uses x, rowptr, src, dst,
// hbuf, zbuf, K2P, nc0, nc1, e_chunk0, n_groups
int gpde_launch_zagg(const GpdeFusedArgs& a, hipStream_t stream);
// 8-wave (two per SIMD) variant with f16 H1 generation (gpde_fused_f16v3.hip)
bool gpde_fused_f16v3_supported(const GpdeFusedArgs& a);
int gpde_launch_fused_f16v3(const GpdeFusedArgs& a, hipStream_t stream);
// store variant (a.hout): gpde_fused_f16v6_kernel<true> where covered, else gpde_fused_f16v3_kernel<true>
bool gpde_fused_store_supported(GpdeFusedArgs probe);
int gpde_launch_fused_store(const GpdeFusedArgs& a, hipStream_t stream);
// one wave per SIMD, 64 x 128 wave tile, 512 registers (gpde_fused_f16v6.hip): the default from 32768 edges on
bool gpde_fused_f16v6_supported(const GpdeFusedArgs& a);
int gpde_launch_fused_f16v6(const GpdeFusedArgs& a, hipStream_t stream);
// the backward's one-pass mode of the same kernel (bw_* fields above) and its operand pre-passes
int gpde_launch_fused_bwd(const GpdeFusedArgs& a, hipStream_t stream);
// scal[1] = bits of max_e B_e (the per-edge first-layer bound of DESIGN.md §3c) alone; scal[0] is zeroed
int gpde_launch_attr_bound(const float* attr, int64_t n_edges, int k0, const float* wmax8, unsigned* scal, hipStream_t stream,
                           int kt = 0, const int* sel = nullptr, const int32_t* src = nullptr, const int32_t* dst = nullptr);
// per chunk node: scale from max |dZ_i|, img1 (out) and img2 (in place over dZ) of GpdeFusedArgs::bw_img1 / bw_img2
int gpde_launch_dz_images(float* dZ, int nn, int K2P, void* img1, float* unscale, hipStream_t stream);
// row_sc / row_isc [rows] = 2^(13 - E(m)), 2^(E(m) - 13) with m = max over the `nparts` slices of rowmax[s][row]
int gpde_launch_row_scales_from_slices(const float* rowmax, int nparts, int rows, float* row_sc, float* row_isc, hipStream_t stream);
// column of a 32-column group stored at half position p of bw_img2's planes: the accumulator row order of the 32x32x16 MFMA
__host__ __device__ static inline int gpde_dz_img2_col(int p) { return 16 * (p >> 4) + (p & 3) + 8 * ((p & 7) >> 2) + 4 * ((p >> 3) & 1); }

struct GpdeGemm3Args {
    const float* zbuf;     // [nn][64*K2P]
    const float* w3q;      // [64*K2P/4][64][4]
    float* part;           // [splits][nn][64]
    int nn, K2P, splits;
    const int32_t* rowptr; // optional: tiles of 64 nodes without in-edges are skipped (nodes nc0 .. nc0 + nn)
    int nc0;
};
int gpde_launch_gemm3(const GpdeGemm3Args& a, hipStream_t stream);

struct GpdeEpilogueArgs {
    const float* part;     // [splits][nn][64]
    const float* x;        // [N][64]
    const int32_t* rowptr;
    const int32_t* src;
    const float* b3;       // [64][64] or nullptr
    const float* root;     // [64][64] or nullptr
    const float* bias;     // [64] or nullptr
    float* out;            // [N][64]
    int nc0, nn, splits, aggr;
    const float* residual; // optional [N][64] added to the result (may alias x, never out)
    int relu_out;          // clamp the result at 0
};
int gpde_launch_epilogue(const GpdeEpilogueArgs& a, hipStream_t stream);

// generic dense layer  Y[rows][KoutP] = relu(X[row_or_perm][0:kin] . W^T + b)   (mode 2 front)
struct GpdeDenseArgs {
    const float* X; int ldx; int kin;   // X row stride (floats) and valid input width
    const int32_t* gather;              // optional row gather (perm), nullptr = identity
    int row0;                           // first row (CSR slot) handled; output row 0 = row0
    const float* W; int ldw;            // [KoutP][ldw] zero padded, ldw multiple of 32
    const float* b;                     // [KoutP]
    float* Y; int KoutP;                // [rows][KoutP]
    int rows; int relu;
};
int gpde_launch_dense(const GpdeDenseArgs& a, hipStream_t stream);

// general fp32 MFMA GEMM (gpde_gemm.hip): C = op(A).op(B) with optional bias / relu / relu-mask,
// batching (grid.z) and split-K partial outputs
struct GpdeGemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    int lda, ldb, ldc;
    int a_kcontig, b_kcontig;       // 1: operand stored [row][k] ; 0: stored [k][row]
    const float* bias;              // [N] or nullptr
    int relu;
    const float* mask; int ldmask;  // C = (mask > 0) ? C : 0
    int accumulate;                 // C += result (after bias/relu/mask)
    int batches; size_t strideA, strideB, strideC;
    int splits; size_t strideSplit; // split-K: partial s written at C + s*strideSplit
    int skew_us;                    // test hook (GPDE_DEBUG_SKEW_US): odd column tiles start this many us late
};
// Test hook shared by the GEMM kernels: with GPDE_DEBUG_SKEW_US=<n> in the environment, workgroups of odd column
// slices spin n microseconds before their first load.  Sibling slices read the same operand rows; a buffer plan that
// lets one slice's output land in rows another slice still has to read then fails deterministically instead of once
// in a dozen runs (tests/test_gpu_repeat.py).  0 / unset: one scalar compare per workgroup.
int gpde_debug_skew_us();
__device__ __forceinline__ void gp_debug_skew(int us) {
    if (us > 0) {
        const unsigned long long t0 = wall_clock64();                // 100 MHz
        while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(64);
    }
}
int gpde_launch_gemm(const GpdeGemmArgs& g, hipStream_t stream);
int gpde_launch_reduce_splits(const float* P, size_t n, int splits, size_t stride, float* C,
                              int accumulate, hipStream_t stream);
// C = (A . B^T) (.) [mask > 0] on split-f16 MFMA (gpde_gemm_f16s.hip): A [M][K] fp32 rows, B given as the split tile
// image of gpde_pack_split_nk ([N/128][K/32][128 rows][hi 64 B | lo 64 B], rows scaled by 2^t_n; ucol = 2^-t_n)
struct GpdeGemmF16sArgs {
    const float* A; int lda; int M;
    const void* bsplit; const float* ucol;
    const float* mask; int ldmask;      // optional [M][ldmask]
    const uint32_t* maskbits; int ldmb; // optional instead of `mask`: bit (n & 31) of maskbits[m * ldmb + n / 32] = "keep"
    float* C; int ldc;
    int K, N;                           // padded sizes: K % 128 == 0, N % 128 == 0
    const float* sc; const float* isc;  // per-row scales (filled by the launcher's pre-pass)
    int n_groups;
    int ksplits; size_t cstride;        // split-K (> 1): partial s at C + s * cstride; M <= 4 * 64 * n per launch group
    const float* xc_x; const int32_t* xc_src;   // contract epilogue (per-edge last layer): C = P[N/128][M][64], see the kernel
    int skew_us;                        // test hook (GPDE_DEBUG_SKEW_US): odd column slices start this many us late
    int no_tile_prefetch;               // A/B hook (GPDE_NT_NO_PREFETCH): per-tile loads as in round 3 (no cross-tile prefetch)
    int no_ks_xcd;                      // A/B hook (GPDE_TN_NO_KS_XCD): split-K workgroups mapped by row group, not one split per XCD
    // gather form (gpde_launch_gemm_f16s_gather; set by that launcher only): A = stack of node tables, B per destination node
    const int32_t* g_src;               // [M] source node of each row (CSR slot)
    const int32_t* g_tile;              // [g_ntiles][4]: destination node (chunk-local), first row, end row, 0
    int g_ntiles;
    size_t g_layer_stride;              // floats between the node tables of consecutive layers (K = 64 * layers)
    size_t g_bnode_bytes;               // bytes of one node's split tile image [N/128][K/32][16 KiB]
    const float* g_unscale;             // [nodes] 2^-t of the node's image
    // first hidden layer generated INSIDE the kernel (round 6; gpde_launch_gemm_f16s_tn / the dU_1 launch of gpde_bwd.hip):
    // H_1[e][n] * sc[n] = relu(sum_d attr'[e][d] * w'[n][d]) on the split-f16 MFMA pair of the forward kernel, attr'[d] = attr[d] *
    // alpha[d] (+ 1 in the bias slot), w' = the per-column image of k_first_layer_wimg.  fl_mode 1 (split-K form): the B chunk
    // images [128 n][32 edges] are generated from the attributes instead of read from memory (no H_1^T image, no k_first_layer_pack);
    // fl_mode 2 (plain row tiles): the epilogue forms dW_1 / db_1 from the tile and the attribute rows (below)
    int fl_mode;
    const float* fl_attr; int fl_ld0;   // fl_mode 2: gathered attributes [fl_rows][fl_ld0] (the first 8 floats of a row are used), slots >= k0 zero;
                                        // fl_mode 1: the split OPERAND image of k_first_layer_aops, [fl_rows = padded K][16] f16 (fl_ld0 = 8 floats)
    int fl_rows;                        // rows of fl_attr that exist (edges beyond it are zero columns of the K padding)
    const void* fl_wimg;                // fl_mode 1: [N][2][8] f16: w'_hi | w'_lo per column
    const float* fl_alpha;              // fl_mode 1: [16]: alpha[d] (0 in the bias slot and beyond), then beta[d] (1 in the bias slot, else 0)
    // fl_mode 2: the FIRST layer's gradients from the epilogue (mask: `maskbits`) - dW_1[n][d] += sum_rows C[row][n] * attr[row][d] (d < 7),
    // db_1[n] += sum_rows C[row][n] - per 64-row tile a partial [N][8], summed in tile order by the launcher (deterministic, independent of
    // the launch geometry).  fl_dw_part: gpde_gemm_f16s_dw_part_floats(M, N) floats of scratch; with fl_skip_store C is not written at
    // all (nothing else reads dU_1: 4 KiB per edge neither written nor read back by k_dw_first - its buffer can be that scratch)
    float* fl_dw_part; float* fl_dw_out; int fl_dw_ld; float* fl_db_out; int fl_skip_store;
};
size_t gpde_gemm_f16s_dw_part_floats(int M, int N);
bool gpde_gemm_f16s_supported(int M, int N, int K, int lda);

// Per-edge backward through the aggregation on split-f16 MFMA (gpde_edge_bwd3.hip): dU[e][n] = (x_j . dZ_i)[n] [H > 0],
// dx_e[c] = sum_n H[e][n] dZ_i[c][n] + dS_i[c] for the CSR slots [e0, e1) = in-edges of the chunk's nodes n0 ..
struct GpdeEdgeBwd3Args {
    const float* x; const int32_t* src; const int32_t* dst;
    const float* dZ;            // split image of the chunk nodes' dZ (gpde_launch_dz_split, in place over the fp32 tensor)
    const float* dz_unscale;    // [chunk nodes] 2^-t of each node's image
    const float* dS;            // [chunk nodes][64]
    const float* H;             // [e1 - e0][K2P] fp32, row 0 = slot e0
    float* dU;                  // [e1 - e0][K2P] or nullptr (dx only)
    float* dx;                  // [N][64] (atomics) when dxe is null
    float* dxe;                 // [e1 - e0][64] per-edge rows (ordered reduction by k_dx_reduce)
    int e0, e1, n0, K2P;
    // optional by-products of the dU tile (all or none; dU must be set): what gpde_launch_gemm_f16s_tn's pass over dU forms
    float* dUt; int ldt;        // transposed copy [K2P][ldt], column e - e0 (columns >= e1 - e0 are the caller's to zero)
    float* row_sc; float* row_isc;               // [e1 - e0] 2^(13 - E(max_n |dU[e][n]|)) and its reciprocal
    float* csum_part; unsigned* cmax_part;       // [(e1 - e0 + 31) / 32][K2P] per 32-slot tile: column sums, column max bits
    int du_accumulate;          // dU += instead of dU = (no by-products then): the applications of a module sharing H sum dL/dU in place
};
int gpde_launch_dz_split(float* dZ, int nn, int K2P, float* unscale, hipStream_t stream);
int gpde_launch_edge_bwd3(const GpdeEdgeBwd3Args& a, hipStream_t stream);
int gpde_launch_gemm_f16s_nt(const GpdeGemmF16sArgs& a, float* row_scale_ws /* 2 * M floats */, hipStream_t stream);
// per-edge last layer of the forward for low in-degree graphs (the reference's own association, never forming [E][4096])
size_t gpde_edge_messages_ws_floats(int64_t n_edges, int n_out);
int gpde_launch_edge_messages(const float* H, int K2P, int64_t n_edges, const void* w3s, const float* ucol3,
                              const float* x, const int32_t* src, const int32_t* rowptr, int64_t n_nodes, float* ws,
                              float* part, hipStream_t stream);
// weight-gradient form: part[s] = partial sums over K split s of dU^T . H (dU [rows][n_out], H [rows][n_in], fp32,
// contraction over the rows); ws = gpde_gemm_f16s_tn_ws_floats(...) floats, part = ksplits * n_out * n_in floats
size_t gpde_gemm_f16s_tn_ws_floats(int rows_max, int n_out, int n_in, int ksplits);
// H given NOT as a tensor but as the first hidden layer of the kernel MLP over gathered attributes:
// H[e][n] = relu(bp[n] + sum_{d < 8} Wp[n][d] * H0[e][d])  (DenseNet's first Linear + ReLU, utilities.py:223-227).  The
// launcher writes the split tile image of H^T straight from H0 (no [rows][n_in] tensor, no column-maximum pass: the
// scales come from the bound |bp[n]| + sum_d |Wp[n][d]| * amax[d]) and the ReLU mask as bits for the dU_1 epilogue.
struct GpdeFirstLayerSpec {
    const float* H0; int ld0;            // gathered attributes [rows][ld0], slots >= k0 zero (ld0 >= 8)
    const float* Wp; int ldw;            // padded first-layer weight [n_in][ldw] (ldw >= 8), bias bp [n_in]
    const float* bp;
    uint32_t* maskbits;                  // out: [rows][n_in / 32]
    int k0;                              // attribute slots in use (the bias takes slot k0); 0: unknown -> image path
    const unsigned* amax_bits;           // [8] optional: bits of a bound of |slot d| over ALL rows the caller will ever pass (every chunk of a
                                         // backward call): the in-kernel first layer then uses ONE set of scales per call - the same H_1 bits,
                                         // ReLU mask included, for an edge whatever the chunking.  NULL: the maxima of these rows
};
bool gpde_first_layer_in_kernel(const GpdeFirstLayerSpec& f, int rows, int ksplits);
// Optional by-products of the pass that transposes dU (it reads every element once): the bias gradient and the per-row
// scales the NEXT GEMM over dU (dU_1 = dU . W^T) needs - instead of two more passes over dU (k_colsum, k_row_scale_kernel).
struct GpdeDuStats {
    float* db_accumulate;                // [n_out] += column sums of dU (ordered: 64-row blocks ascending)
    float* row_sc; float* row_isc;       // [rows] 2^(13 - E(max_k |dU[row][k]|)) and its inverse
    // set by a producer that has ALREADY written the transposed copy (at gpde_gemm_f16s_tn_at: [n_out][ld], columns < rows)
    // and the row scales above (gpde_edge_bwd3.hip): its per-32-row-tile column sums / column max bits [ceil(rows / 32)][n_out]
    const float* tile_csum; const unsigned* tile_cmax;
};
// where gpde_launch_gemm_f16s_tn keeps the transposed copy of dU inside `ws`, and its row length (>= rows, zero padded)
float* gpde_gemm_f16s_tn_at(float* ws, int rows, int ksplits, int* ld);
int gpde_launch_gemm_f16s_tn(const float* dU, int ldu, int n_out, const float* H, int ldh, int n_in, int rows,
                             int ksplits, float* ws, float* part, hipStream_t stream,
                             const unsigned* du_absmax_bits = nullptr /* [n_out] column maxima of |dU| as bit patterns */,
                             const GpdeFirstLayerSpec* first_layer = nullptr /* non-null: H is this layer (H / ldh unused) */,
                             const GpdeDuStats* du_stats = nullptr);
// ---- depth-deferred backward (gpde_nnconv_bwd_deferred): dU_2 of the edges of a node chunk from the stacked layer inputs ----
// dU[e][n] = (sum_{l < Lp, c < 64} xstack[l][src[e]][c] * dZ^(l)[dst e][c][n]) (.) [mask[e][n] > 0]   for the chunk's rows.
//   tiles   [ntiles][4] from gpde_launch_tile_list (host count: gpde_tile_count)
//   bimg    per chunk node the split tile image of B_i[n][(l, c)] = dZ^(l)_i[c][n] (gpde_launch_dz_image), unscale [nodes]
//   sc/isc  [N] per source node (gpde_launch_xstack_scales)
int gpde_launch_gemm_f16s_gather(const float* xstack, size_t layer_stride, int Lp, const float* sc_src, const float* isc_src,
                                 const int32_t* src_rows, int rows, const int32_t* tiles, int ntiles, const void* bimg,
                                 const float* unscale, const float* mask, int ldmask, float* C, int ldc, int N, hipStream_t stream);
size_t gpde_dz_image_bytes_per_node(int Lp, int K2P);
// dZ [L][nn_alloc][64][K2P] fp32 (layer stride dz_layer_stride floats) -> per node image + scales; bits [nn] scratch
int gpde_launch_dz_image(const float* dZ, size_t dz_layer_stride, int L, int Lp, int nn, int K2P, unsigned* bits, float* scale,
                         float* unscale, void* img, hipStream_t stream);
int gpde_launch_xstack_scales(const float* xstack, size_t layer_stride, int L, int64_t n_nodes, float* sc, float* isc, hipStream_t stream);
int gpde_tile_count(const int32_t* rowptr_host, int na, int nb);          // sum over nodes of ceil(deg / 256)
int gpde_launch_tile_list(const int32_t* rowptr, int na, int nn, int e0, int32_t* tiles, hipStream_t stream);
// split tile image of a row-major [n][k] matrix (ld = k): the W2 layout of gpde_mlp_pack for any operand
int gpde_pack_split_nk(const float* Wnk, int n, int k, int NP, int KP, void* out, float* ucol, hipStream_t stream);
int gpde_num_cus();
