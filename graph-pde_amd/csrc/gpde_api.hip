// C-ABI entry points of libgpde.so: launch planning and orchestration of the NNConv forward.
// Boundary replaced: nn_conv.NNConv_old.forward -> propagate -> message -> aggregate -> update
// (/root/reference/graph-neural-operator/nn_conv.py:267-282) — see include/gpde.h.
#include "gpde_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static thread_local char g_err[512] = "";

void gpde_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
__global__ void k_zero_words(unsigned* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}
__global__ void k_zero_vec4(uint4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = uint4{0u, 0u, 0u, 0u};
}
__global__ void k_zero2d_words(unsigned* __restrict__ p, size_t pitch_words, size_t width_words, size_t rows) {
    const size_t n = width_words * rows, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[(i / width_words) * pitch_words + i % width_words] = 0u;
}
__global__ void k_copy_words(unsigned* __restrict__ d, const unsigned* __restrict__ s_, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s_[i];
}
unsigned fill_blocks(size_t items) {
    size_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
}  // namespace

hipError_t gpde_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (bytes % 4 != 0 || ((uintptr_t)p & 3)) return hipErrorInvalidValue;
    if (bytes % 16 == 0 && ((uintptr_t)p & 15) == 0) hipLaunchKernelGGL(k_zero_vec4, dim3(fill_blocks(bytes / 16)), dim3(256), 0, st, (uint4*)p, bytes / 16);
    else hipLaunchKernelGGL(k_zero_words, dim3(fill_blocks(bytes / 4)), dim3(256), 0, st, (unsigned*)p, bytes / 4);
    return hipGetLastError();
}
hipError_t gpde_zero2d_async(void* p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t st) {
    if (width_bytes == 0 || rows == 0) return hipSuccess;
    if (pitch_bytes % 4 != 0 || width_bytes % 4 != 0 || ((uintptr_t)p & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_zero2d_words, dim3(fill_blocks(width_bytes / 4 * rows)), dim3(256), 0, st, (unsigned*)p, pitch_bytes / 4, width_bytes / 4, rows);
    return hipGetLastError();
}
hipError_t gpde_copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (bytes % 4 != 0 || ((uintptr_t)dst & 3) || ((uintptr_t)src & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_copy_words, dim3(fill_blocks(bytes / 4)), dim3(256), 0, st, (unsigned*)dst, (const unsigned*)src, bytes / 4);
    return hipGetLastError();
}

namespace {
GpdeSwitches load_switches() {
    auto on = [](const char* n) { return getenv(n) != nullptr; };
    auto num = [](const char* n) { const char* e = getenv(n); return e ? atoi(e) : 0; };
    GpdeSwitches s{};
    s.bwd_gemm_f32 = on("GPDE_BWD_GEMM_F32");
    s.bwd_dw2_f32 = on("GPDE_BWD_DW2_F32");
    s.bwd_recompute_f32 = on("GPDE_BWD_RECOMPUTE_F32");
    s.bwd_h1_materialize = on("GPDE_BWD_H1_MATERIALIZE");
    s.bwd_h1_gemm = on("GPDE_BWD_H1_GEMM");
    s.bwd_h1_image = on("GPDE_BWD_H1_IMAGE");
    s.bwd_dw1_pass = on("GPDE_BWD_DW1_PASS");
    s.bwd_zagg_f32 = on("GPDE_BWD_ZAGG_F32");
    s.bwd_node_terms_gemm = on("GPDE_BWD_NODE_TERMS_GEMM");
    s.bwd_dw1_gemm = on("GPDE_BWD_DW1_GEMM");
    s.bwd_du_passes = on("GPDE_BWD_DU_PASSES");
    s.bwd_du_transpose_pass = on("GPDE_BWD_DU_TRANSPOSE_PASS");
    s.bwd_one_pass = on("GPDE_BWD_ONE_PASS");
    s.store_v3 = on("GPDE_STORE_V3");
    s.nt_no_prefetch = on("GPDE_NT_NO_PREFETCH");
    s.tn_no_ks_xcd = on("GPDE_TN_NO_KS_XCD");
    s.edge_bwd = num("GPDE_EDGE_BWD");
    const int v = num("GPDE_DEBUG_SKEW_US");
    s.debug_skew_us = v > 0 ? (v < 100000 ? v : 100000) : 0;
    return s;
}
GpdeSwitches& switches_storage() {
    static GpdeSwitches s = load_switches();        // first native call of the process (thread-safe static initialisation)
    return s;
}
}  // namespace

const GpdeSwitches& gpde_switches() { return switches_storage(); }

extern "C" int gpde_reload_switches(void) {
    switches_storage() = load_switches();
    return GPDE_OK;
}

// An ablation build (-DGPDE_ABL_*: part of the arithmetic removed, WRONG results, timing only) or an instrumented build
// (-DGPDE_*_TIMING: clock64 probes) says so in its version word; graph-pde_amd/_lib.py refuses the former unless asked.
#if defined(GPDE_ABL_2MFMA) || defined(GPDE_ABL_NOSTAGE) || defined(GPDE_ABL_NOBARRIER) || defined(GPDE_ABL_NOGEMM2) || defined(GPDE_ABL_NOWAIT) || defined(GPDE_BWABL)
#define GPDE_BUILD_FLAGS_ GPDE_VERSION_ABLATION
#elif defined(GPDE_V6_TIMING) || defined(GPDE_NT_TIMING) || defined(GPDE_EB2_TIMING) || defined(GPDE_V3_TIMING)
#define GPDE_BUILD_FLAGS_ GPDE_VERSION_INSTRUMENTED
#else
#define GPDE_BUILD_FLAGS_ 0
#endif
extern "C" int gpde_version(void) { return GPDE_VERSION | GPDE_BUILD_FLAGS_; }
extern "C" const char* gpde_last_error(void) { return g_err; }

namespace {

// ---- optional HIP-event timing of the kernels of gpde_nnconv_fwd (bench.py roofline leg) -------
struct EvPair { hipEvent_t a, b; int kind; };   // kind: GPDE_PROF_* (include/gpde.h)
thread_local bool g_prof_on = false;
thread_local std::vector<EvPair> g_prof;

struct ProfScope {
    bool on; EvPair p; hipStream_t s;
    ProfScope(int kind, hipStream_t stream) : on(g_prof_on), s(stream) {
        if (!on) return;
        p.kind = kind;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(p.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(p.b, s);
        g_prof.push_back(p);
    }
};

constexpr size_t kAlign = 256;
size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

int num_cus_impl() {
    // per device (a process may drive several GPUs), cached after the first query
    static std::atomic<int> cus[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;   // MI355X
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

struct Plan {
    GpdePackLayout L;
    int64_t nodes_per_chunk;
    int n_chunks;
    int n_groups;        // fused-kernel edge groups (workgroups per hidden slice)
    size_t off_part, off_z, off_ha, off_hb, off_xs, off_scal, off_blk;
    int nblk_max;        // f16v6 work queue: block-bound slots (0: no queue)
    size_t h_floats;     // mode 2: floats per ping-pong activation buffer
};

int pick_splits(int64_t nn) {
    // gemm3 streams Z from HBM with one 16 KiB chunk in flight per workgroup: ~1000 resident workgroups
    // (4 per CU) keep ~16 MB in flight, enough for the HBM latency-bandwidth product
    const int64_t tiles = (nn + 63) / 64;
    int s = 1;
    while (s < 64 && tiles * s < 1024) s *= 2;
    return s;
}

int make_plan(int64_t N, int64_t E, int n_layers, const int32_t* dims, size_t ws_bytes, bool sizing,
              Plan* P, size_t* needed, bool hidden_given = false) {
    int rc = gpde_pack_layout(n_layers, dims, &P->L);
    if (rc != GPDE_OK) return rc;
    const GpdePackLayout& L = P->L;
    const size_t zrow = (size_t)GP_W * L.K2P * sizeof(float);
    const size_t prow = (size_t)64 * GP_W * sizeof(float);   // worst case: 64 splits
    size_t fixed = 0;
    P->h_floats = 0;
    // f16-split aggregation (mode 1): x as split f16 words [N][64] + two scalars (gpde_prep.hip)
    // (sized for every kernel shape: gpde_nnconv_fwd_hidden uses it whatever the MLP looks like)
    const size_t xs_bytes = (L.mode == 1 || hidden_given || sizing) ? align_up((size_t)(N > 0 ? N : 1) * GP_W * 4) + kAlign : 0;
    if (L.mode == 2 && !hidden_given) {
        int kmax = 0;
        for (int l = 1; l <= n_layers - 1; ++l) kmax = kmax > L.frontKP[l] ? kmax : L.frontKP[l];
        P->h_floats = (size_t)(E > 0 ? E : 1) * kmax;
        fixed = 2 * align_up(P->h_floats * sizeof(float));
    }
    fixed += xs_bytes;
    // f16v6 work queue: block bounds [nblk_max + 1] + block count + one counter per column slice
    P->nblk_max = (L.mode == 1 && E < ((int64_t)1 << 31) - GP_QBLOCK) ? (int)(E / GP_QBLOCK + E / 8 / (GP_QBLOCK / 8)) + 10 : 0;
    const size_t blk_bytes = P->nblk_max ? align_up(((size_t)P->nblk_max + 2 + 64) * 4) : 0;
    fixed += blk_bytes;
    int64_t npc;
    if (sizing) {
        // recommended: all nodes if Z fits 4 GiB, else 4 GiB worth of nodes (at least one tile)
        int64_t cap = (int64_t)(((size_t)4 << 30) / zrow);
        if (cap < 64) cap = 64;
        npc = N < cap ? N : cap;
        if (npc < 1) npc = 1;
    } else {
        if (ws_bytes < fixed + 2 * kAlign) {
            gpde_set_error("gpde_nnconv_fwd: workspace %zu bytes cannot hold the %zu-byte hidden "
                           "activation buffers", ws_bytes, fixed);
            return GPDE_EWORKSPACE;
        }
        npc = (int64_t)((ws_bytes - fixed - 2 * kAlign) / (zrow + prow));
        // the whole graph in one chunk whenever it fits exactly (the estimate above gives up two
        // alignment units, which used to split a graph sized with the recommended workspace)
        if (N > 0 && align_up((size_t)N * prow) + align_up((size_t)N * zrow) + fixed <= ws_bytes) npc = N;
        if (npc > N) npc = N;
        const int64_t min_nodes = N < 64 ? N : 64;
        if (npc < min_nodes || (N > 0 && npc < 1)) {
            gpde_set_error("gpde_nnconv_fwd: workspace %zu bytes holds %lld destination nodes, need >= %lld "
                           "(%zu bytes per node)", ws_bytes, (long long)npc, (long long)min_nodes, zrow + prow);
            return GPDE_EWORKSPACE;
        }
        if (npc < N && npc > 64) npc = npc / 64 * 64;
    }
    if (npc < 1) npc = 1;
    P->nodes_per_chunk = npc;
    P->n_chunks = (int)((N + npc - 1) / npc);
    if (P->n_chunks < 1) P->n_chunks = 1;
    size_t off = 0;
    P->off_part = off; off += align_up((size_t)npc * prow);
    P->off_z = off;    off += align_up((size_t)npc * zrow);
    P->off_ha = off;   off += (L.mode == 2 && !hidden_given) ? align_up(P->h_floats * sizeof(float)) : 0;
    P->off_hb = off;   off += (L.mode == 2 && !hidden_given) ? align_up(P->h_floats * sizeof(float)) : 0;
    P->off_xs = off;   off += xs_bytes ? xs_bytes - kAlign : 0;
    P->off_scal = off; off += xs_bytes ? kAlign : 0;
    P->off_blk = off;  off += blk_bytes;
    if (needed) *needed = off;
    // fused-kernel grid: ~one workgroup per CU, never more edge groups than 4-wave tile sets
    const int ns = L.K2P / GP_TN;
    int groups = num_cus_impl() / ns;
    if (groups < 1) groups = 1;
    const int64_t tiles_chunk = ((E + GP_TE - 1) / GP_TE + P->n_chunks - 1) / P->n_chunks;
    int64_t gcap = (tiles_chunk + GP_WAVES - 1) / GP_WAVES;
    // small graphs (MGKN levels): let a wave own as little as ONE destination node.  Its in-edges are a partial
    // tile (the matrix pipe idles anyway), but the per-node flush of Z - 32 KiB per node and slice - is what such a
    // launch spends its time on, and it is spread over the chip instead of a handful of waves
    const int64_t gcap_nodes = (npc + GP_WAVES - 1) / GP_WAVES;
    if (gcap < gcap_nodes) gcap = gcap_nodes;
    if (groups > gcap) groups = (int)(gcap < 1 ? 1 : gcap);
    P->n_groups = groups;
    return GPDE_OK;
}

// Which fused edge kernel serves a forward call -- ONE place, used by fwd_impl and by the query
// gpde_nnconv_fwd_kernel (bench.py / tests key profiles and expectations on the kernel symbol).
enum FusedKind { FK_GENERIC = 0, FK_V3 = 1, FK_V6 = 2 };
struct FusedChoice { FusedKind kind; bool g2f16; };
FusedChoice choose_fused(const GpdePackLayout& L, int mode, uint32_t flags, int64_t n_edges, int kt) {
    FusedChoice c{FK_GENERIC, false};
    if (mode != 1 || !(flags & GPDE_FWD_F16SPLIT)) return c;
    GpdeFusedArgs probe{};
    probe.k0 = L.k0; probe.K1P = L.K1P; probe.K2P = L.K2P; probe.kt = kt;
    if (!gpde_fused_f16v3_supported(probe)) return c;
    c.kind = FK_V3;
    // f16-split aggregation inside the fused kernel: worth its three tiny pre-pass launches from a few ten
    // thousand edges on; GPDE_FWD_AGG_F16 / GPDE_FWD_AGG_F32 force it on / off
    c.g2f16 = !(flags & GPDE_FWD_AGG_F32) && n_edges > 0 && ((flags & GPDE_FWD_AGG_F16) || n_edges >= 32768);
    probe.xs = c.g2f16 ? (const unsigned*)(uintptr_t)8 : nullptr;      // "will be present"
    if (c.g2f16 && !(flags & GPDE_FWD_F16SPLIT_8WAVE) && gpde_fused_f16v6_supported(probe)) c.kind = FK_V6;
    return c;
}

}  // namespace

int gpde_num_cus() { return num_cus_impl(); }

extern "C" const char* gpde_nnconv_fwd_kernel(int64_t n_edges, int n_layers, const int32_t* dims, uint32_t flags) {
    GpdePackLayout L;
    if (!dims || gpde_pack_layout(n_layers, dims, &L) != GPDE_OK) return "";
    switch (choose_fused(L, L.mode, flags, n_edges, 0).kind) {
        case FK_V6: return "gpde_fused_f16v6_kernel";
        case FK_V3: return "gpde_fused_f16v3_kernel";
        default: return "gpde_fused_kernel";
    }
}

extern "C" size_t gpde_nnconv_fwd_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers,
                                                  const int32_t* dims) {
    Plan P;
    size_t need = 0;
    if (!dims || n_nodes < 0 || n_edges < 0) return 0;
    if (make_plan(n_nodes, n_edges, n_layers, dims, 0, true, &P, &need) != GPDE_OK) return 0;
    return need + 2 * kAlign;
}

extern "C" int gpde_nnconv_fwd_plan(int64_t n_nodes, int64_t n_edges, int n_layers,
                                    const int32_t* dims, size_t ws_bytes, int32_t* n_chunks,
                                    int64_t* nodes_per_chunk, int32_t* fused_workgroups,
                                    int32_t* mode) {
    Plan P;
    if (!dims) { gpde_set_error("gpde_nnconv_fwd_plan: dims is null"); return GPDE_EINVAL; }
    int rc = make_plan(n_nodes, n_edges, n_layers, dims, ws_bytes, false, &P, nullptr);
    if (rc != GPDE_OK) return rc;
    if (n_chunks) *n_chunks = P.n_chunks;
    if (nodes_per_chunk) *nodes_per_chunk = P.nodes_per_chunk;
    if (fused_workgroups) *fused_workgroups = P.n_groups * (P.L.K2P / GP_TN);
    if (mode) *mode = P.L.mode;
    return GPDE_OK;
}

namespace {

// `hidden` != nullptr: the last hidden activations are given ([CSR slot][K2P], gpde_hidden_fwd); only the
// aggregation, the last Linear and update() run (gpde_nnconv_fwd_hidden)
// `kt` != 0: edge_attr is a NODE table [n_nodes][kt] and slot d of an edge's attribute is
// table[(sel[d] >> 8 ? dst : src)][sel[d] & 255] (gpde_nnconv_fwd_nodeattr)
int fwd_impl(const float* x, int64_t n_nodes, const float* edge_attr, int64_t n_edges,
             const int32_t* rowptr, const int32_t* src, const int32_t* dst, const int32_t* perm,
             int n_layers, const int32_t* dims, const void* packed, const float* root,
             const float* bias, int aggr, uint32_t flags, const float* hidden, float* out, void* ws,
             size_t ws_bytes, hipStream_t stream, int kt = 0, const int* sel = nullptr,
             const float* hidden_absmax = nullptr, int64_t hidden_nodes = -1, const float* residual = nullptr,
             int relu_out = 0, float* z_keep = nullptr) {
    // hidden_nodes in [0, n_nodes): MIXED call (gpde_nnconv_fwd_mixed_keepz) -- `hidden` covers the in-edges of
    // nodes [0, hidden_nodes) only (a graph whose H does not fit memory, e.g. 391 GB at the 241^2 graph);
    // those nodes aggregate from it, the others run the fused kernel on edge_attr
    const bool mixed = hidden && hidden_nodes >= 0 && hidden_nodes < n_nodes;
    if (aggr != GPDE_AGGR_ADD && aggr != GPDE_AGGR_MEAN) {
        gpde_set_error("gpde_nnconv_fwd: aggr %d not implemented (add=0, mean=1)", aggr);
        return GPDE_EUNSUPPORTED;
    }
    if (n_nodes == 0) return GPDE_OK;
    if (!ws) { gpde_set_error("gpde_nnconv_fwd: workspace is null"); return GPDE_EWORKSPACE; }
    Plan P;
    int rc = make_plan(n_nodes, n_edges, n_layers, dims, ws_bytes, false, &P, nullptr, hidden != nullptr && !mixed);
    if (rc != GPDE_OK) return rc;
    const GpdePackLayout& L = P.L;
    const int mode = (hidden && !mixed) ? 2 : L.mode;
    if (mixed && L.mode != 1) { gpde_set_error("gpde_nnconv_fwd_mixed_keepz: built for 3-Linear kernel MLPs"); return GPDE_EUNSUPPORTED; }
    const float* pk = (const float*)packed;
    char* w = (char*)ws;
    w = (char*)(((uintptr_t)w + kAlign - 1) / kAlign * kAlign);
    float* part = (float*)(w + P.off_part);
    float* zbuf = (float*)(w + P.off_z);
    float* ha = (float*)(w + P.off_ha);
    float* hb = (float*)(w + P.off_hb);

    const float* hfinal = hidden;
    if (!hidden && L.mode == 2 && n_edges > 0) {   // (a mixed call has L.mode == 1)
        // front layers over all edges (CSR order), ping-pong between ha / hb
        const float* in = edge_attr;
        int ldx = L.k0, kin = L.k0;
        const int32_t* gather = perm;
        float* bufs[2] = {ha, hb};
        for (int l = 0; l < n_layers - 1; ++l) {
            GpdeDenseArgs d;
            d.X = in; d.ldx = ldx; d.kin = kin; d.gather = gather; d.row0 = 0;
            d.W = pk + L.off_front_w[l]; d.ldw = L.frontKP[l];
            d.b = pk + L.off_front_b[l];
            d.Y = bufs[l & 1]; d.KoutP = L.frontKP[l + 1];
            d.rows = (int)n_edges; d.relu = 1;
            rc = gpde_launch_dense(d, stream);
            if (rc != GPDE_OK) return rc;
            in = d.Y; ldx = kin = d.KoutP; gather = nullptr;
        }
        hfinal = in;
    }

    // ---- low in-degree graphs: the last Linear per EDGE (the reference's own association) --------------------------
    // The re-association of DESIGN.md §2 trades 8.4 MFLOP per edge for 8.4 MFLOP per NODE plus a 256 KiB Z round
    // trip per node - a loss when a node has two or three in-edges (MGKN-orthogonal Burgers: 8192 nodes, 16 k edges,
    // 2 x 2.1 GB of Z per call).  There W_e = W3 . h_e is formed tile by tile on the split-f16 GEMM and contracted
    // with x_j in its epilogue ([E][4096] never exists), messages are summed per destination in CSR order.
    bool store_ok = false;
    if (L.mode == 1 && L.k0 + 1 <= 8) {
        GpdeFusedArgs probe{};
        probe.k0 = L.k0; probe.K1P = L.K1P; probe.K2P = L.K2P;
        store_ok = gpde_fused_store_supported(probe);
    }
    // z_keep: the caller wants Z_i = sum_e x_j (x) H_e of EVERY node ([N][64][K2P], zero-initialised by the caller: nodes
    // without in-edges are not written) - the backward's dW_3 reads it instead of re-aggregating (gpde_nnconv_fwd_keepz)
    const bool edge_path = !z_keep && ((flags & GPDE_FWD_F16SPLIT) || (hidden && hidden_absmax)) && !mixed && !kt && L.has_w3s && P.n_chunks == 1 &&
                           n_edges >= 4096 && n_edges <= 4 * n_nodes && n_edges < ((int64_t)1 << 24) &&
                           !(flags & GPDE_FWD_NO_EDGE_PATH) &&
                           (hidden || (L.mode == 1 && store_ok)) &&   // H by a fused store kernel (its own support check: LDS limits, chunk parity)
                           (size_t)(n_edges) * L.K2P + gpde_edge_messages_ws_floats(n_edges, GP_W * GP_W) <=
                               (size_t)n_nodes * GP_W * L.K2P;
    if (edge_path) {
        float* Hbuf = zbuf;                                  // the Z region is free on this path
        float* ews = zbuf + (hidden ? 0 : (size_t)n_edges * L.K2P);
        const float* Hrows = hidden;
        if (!hidden) {
            ProfScope ps(GPDE_PROF_FUSED, stream);
            GpdeFusedArgs f{};
            f.attr = edge_attr; f.rowptr = rowptr; f.perm = perm;
            f.w1 = pk + L.off_w1; f.w2t = pk + L.off_w2t; f.b2 = pk + L.off_b2;
            f.w2h = pk + L.off_w2h; f.ucol = pk + L.off_ucol; f.w1h = pk + L.off_w1h; f.fcol = pk + L.off_fcol;
            f.hout = Hbuf; f.hmax_out = nullptr; f.k0 = L.k0; f.K1P = L.K1P; f.K2P = L.K2P;
            f.nc0 = 0; f.nc1 = (int)n_nodes; f.e_chunk0 = 0;
            const int ns = L.K2P / GP_TN;
            int groups = gpde_num_cus() / ns; if (groups < 1) groups = 1;
            const int64_t gcap = ((n_edges + GP_TE - 1) / GP_TE + GP_WAVES - 1) / GP_WAVES;
            if (groups > gcap) groups = (int)gcap;
            f.n_groups = groups;
            if ((rc = gpde_launch_fused_store(f, stream)) != GPDE_OK) return rc;
            Hrows = Hbuf;
        }
        {
            ProfScope ps(GPDE_PROF_GEMM3, stream);
            rc = gpde_launch_edge_messages(Hrows, L.K2P, n_edges, pk + L.off_w3s, pk + L.off_ucol3, x, src, rowptr, n_nodes,
                                           ews, part, stream);
            if (rc != GPDE_OK) return rc;
        }
        GpdeEpilogueArgs e;
        e.part = part; e.x = x; e.rowptr = rowptr; e.src = src;
        e.b3 = pk + L.off_b3; e.root = root; e.bias = bias; e.out = out;
        e.nc0 = 0; e.nn = (int)n_nodes; e.splits = 1; e.aggr = aggr;
        e.residual = residual; e.relu_out = relu_out;
        ProfScope ps2(GPDE_PROF_EPILOGUE, stream);
        return gpde_launch_epilogue(e, stream);
    }

    const unsigned* xs = nullptr;
    const unsigned* scal = nullptr;
    const FusedChoice fc = choose_fused(L, mode, flags, n_edges, kt);
    if (kt && fc.kind == FK_GENERIC) {
        gpde_set_error("gpde_nnconv_fwd_mixed_keepz (node_attr): built for 3-Linear kernel MLPs on the f16-split kernel only");
        return GPDE_EUNSUPPORTED;
    }
    if ((!hidden || mixed) && fc.g2f16) {
        ProfScope psp(GPDE_PROF_PREP, stream);
        rc = gpde_launch_g2_prep(x, n_nodes, edge_attr, n_edges, L.k0, pk + L.off_w1 + (size_t)L.K1P * 8,
                                 (unsigned*)(w + P.off_scal), (unsigned*)(w + P.off_xs), stream, kt, sel, src, dst);
        if (rc != GPDE_OK) return rc;
        xs = (const unsigned*)(w + P.off_xs);
        scal = (const unsigned*)(w + P.off_scal);
    }

    // hidden activations given together with their maximum: the streaming aggregation on split f16
    if (hidden && !mixed && hidden_absmax && n_edges >= 32768 && !(flags & GPDE_FWD_AGG_F32)) {
        rc = gpde_launch_g2_prep(x, n_nodes, nullptr, 0, 0, nullptr, (unsigned*)(w + P.off_scal),
                                 (unsigned*)(w + P.off_xs), stream);
        if (rc != GPDE_OK) return rc;
        xs = (const unsigned*)(w + P.off_xs);
        scal = (const unsigned*)(w + P.off_scal);
    }

    const int64_t hn = mixed ? hidden_nodes : (hidden ? n_nodes : 0);     // nodes served from `hidden`
    for (int64_t nc0 = 0, nc1 = 0; nc0 < n_nodes; nc0 = nc1) {
        const int64_t lim = nc0 < hn ? hn : n_nodes;                       // a chunk never straddles hn
        nc1 = (nc0 + P.nodes_per_chunk < lim) ? nc0 + P.nodes_per_chunk : lim;
        const bool from_h = nc0 < hn;
        const int nn = (int)(nc1 - nc0);
        const int splits = pick_splits(nn);
        if (n_edges > 0) {
            GpdeFusedArgs f{};
            f.x = x; f.attr = edge_attr; f.rowptr = rowptr; f.src = src; f.dst = dst; f.perm = perm;
            f.w1 = pk + L.off_w1; f.w2t = pk + L.off_w2t; f.b2 = pk + L.off_b2;
            f.w2h = pk + L.off_w2h; f.ucol = pk + L.off_ucol;
            f.w1h = pk + L.off_w1h; f.fcol = pk + L.off_fcol;
            float* zc = z_keep ? z_keep + (size_t)nc0 * GP_W * L.K2P : zbuf;
            f.hbuf = hfinal; f.zbuf = zc; f.xs = xs; f.scal = scal;
            f.kt = kt; f.hmax = (from_h && xs) ? (const unsigned*)hidden_absmax : nullptr;
            for (int d = 0; d < 8; ++d) f.sel[d] = (kt && sel) ? sel[d < L.k0 ? d : L.k0 - 1] : 0;
            f.k0 = L.k0; f.K1P = L.K1P; f.K2P = L.K2P;
            f.nc0 = (int)nc0; f.nc1 = (int)nc1; f.e_chunk0 = 0; f.n_groups = P.n_groups;
            {
                ProfScope ps(GPDE_PROF_FUSED, stream);
                if (!from_h && fc.kind == FK_V6 && P.nblk_max && L.K2P / GP_TN <= 64 && !(flags & GPDE_FWD_STATIC_RANGES)) {
                    int32_t* blk = (int32_t*)(w + P.off_blk);
                    f.blk = blk; f.qn = blk + P.nblk_max + 1; f.qctr = (unsigned*)(blk + P.nblk_max + 2);
                    rc = gpde_launch_block_bounds(rowptr, (int)nc0, (int)nc1, P.nblk_max, blk, blk + P.nblk_max + 1,
                                                  (unsigned*)(blk + P.nblk_max + 2), L.K2P / GP_TN, stream);
                    if (rc != GPDE_OK) return rc;
                }
                if (from_h) rc = gpde_launch_zagg(f, stream);
                else if (fc.kind == FK_V6) rc = gpde_launch_fused_f16v6(f, stream);
                else if (fc.kind == FK_V3) rc = gpde_launch_fused_f16v3(f, stream);
                else rc = gpde_launch_fused(mode, (flags & GPDE_FWD_F16SPLIT) != 0 && mode == 1, f, stream);
            }
            if (rc != GPDE_OK) return rc;
            ProfScope ps1(GPDE_PROF_GEMM3, stream);
            GpdeGemm3Args g;
            g.zbuf = zc; g.w3q = pk + L.off_w3q; g.part = part; g.nn = nn; g.K2P = L.K2P;
            g.splits = splits; g.rowptr = rowptr; g.nc0 = (int)nc0;
            rc = gpde_launch_gemm3(g, stream);
            if (rc != GPDE_OK) return rc;
        }
        GpdeEpilogueArgs e;
        e.part = part; e.x = x; e.rowptr = rowptr; e.src = src;
        e.b3 = pk + L.off_b3; e.root = root; e.bias = bias; e.out = out;
        e.nc0 = (int)nc0; e.nn = nn; e.splits = splits; e.aggr = aggr;
        e.residual = residual; e.relu_out = relu_out;
        ProfScope ps2(GPDE_PROF_EPILOGUE, stream);
        rc = gpde_launch_epilogue(e, stream);
        if (rc != GPDE_OK) return rc;
    }
    return GPDE_OK;
}

}  // namespace

extern "C" int gpde_nnconv_fwd(const float* x, int64_t n_nodes, const float* edge_attr,
                               int64_t n_edges, const int32_t* rowptr, const int32_t* src,
                               const int32_t* dst, const int32_t* perm, int n_layers,
                               const int32_t* dims, const void* packed, const float* root,
                               const float* bias, int aggr, uint32_t flags, float* out, void* ws,
                               size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out)) ||
        (n_edges > 0 && (!edge_attr || !src || !dst || !perm))) {
        gpde_set_error("gpde_nnconv_fwd: null/negative argument");
        return GPDE_EINVAL;
    }
    return fwd_impl(x, n_nodes, edge_attr, n_edges, rowptr, src, dst, perm, n_layers, dims, packed, root, bias,
                    aggr, flags, nullptr, out, ws, ws_bytes, (hipStream_t)stream_);
}

// The general forward (round 5: the mixed, node-table and node-table + partial-H forwards folded into one): attributes from `edge_attr` +
// `perm` or from node data (`node_attr`: row f3; then edge_attr / perm are unused and may be NULL), optionally the part of the
// hidden activations that fits memory (`hidden`: rows of the in-edges of nodes [0, hidden_nodes); 0 = none), optionally Z kept
// for the backward (`z_keep`; NULL = none).
extern "C" int gpde_nnconv_fwd_mixed_keepz(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr,
                                           const float* hidden, const float* hidden_absmax, int64_t hidden_nodes, int64_t n_edges,
                                           const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                                           const int32_t* perm, int n_layers, const int32_t* dims, const void* packed,
                                           const float* root, const float* bias, int aggr, uint32_t flags, float* z_keep,
                                           float* out, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out)) ||
        (n_edges > 0 && (!src || !dst || (!node_attr && (!edge_attr || !perm)))) || hidden_nodes < 0 || hidden_nodes > n_nodes ||
        (hidden_nodes > 0 && !hidden)) {
        gpde_set_error("gpde_nnconv_fwd_mixed_keepz: null/negative argument");
        return GPDE_EINVAL;
    }
    int kt = 0;
    int sel[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* attr = edge_attr;
    if (node_attr) {
        const GpdeNodeAttr* na = node_attr;
        if (!na->table || na->stride < 1 || na->n_slots != dims[0] || dims[0] < 1 || dims[0] > 7) {
            gpde_set_error("gpde_nnconv_fwd_mixed_keepz: GpdeNodeAttr must describe dims[0] = 1..7 slots of a node table");
            return GPDE_EINVAL;
        }
        for (int d = 0; d < 8; ++d) {
            sel[d] = na->sel[d < dims[0] ? d : dims[0] - 1];
            if ((sel[d] >> 8) < 0 || (sel[d] >> 8) > 1 || (sel[d] & 255) >= na->stride) {
                gpde_set_error("gpde_nnconv_fwd_mixed_keepz: GpdeNodeAttr.sel[%d] = 0x%x (endpoint << 8 | column, column < %d)", d, sel[d], na->stride);
                return GPDE_EINVAL;
            }
        }
        kt = na->stride;
        attr = na->table;
        perm = nullptr;
    }
    if (hidden_nodes == 0)
        return fwd_impl(x, n_nodes, attr, n_edges, rowptr, src, dst, perm, n_layers, dims, packed, root, bias,
                        aggr, flags, nullptr, out, ws, ws_bytes, (hipStream_t)stream_, kt, kt ? sel : nullptr, nullptr, -1, nullptr, 0, z_keep);
    return fwd_impl(x, n_nodes, attr, n_edges, rowptr, src, dst, perm, n_layers, dims, packed, root, bias,
                    aggr, flags, hidden, out, ws, ws_bytes, (hipStream_t)stream_, kt, kt ? sel : nullptr, hidden_absmax,
                    hidden_nodes, nullptr, 0, z_keep);
}

extern "C" int gpde_nnconv_fwd_hidden(const float* x, int64_t n_nodes, const float* hidden,
                                      const float* hidden_absmax, int64_t n_edges, const int32_t* rowptr, const int32_t* src,
                                      const int32_t* dst, int n_layers, const int32_t* dims,
                                      const void* packed, const float* root, const float* bias,
                                      int aggr, float* out, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out)) ||
        (n_edges > 0 && (!hidden || !src || !dst))) {
        gpde_set_error("gpde_nnconv_fwd_hidden: null/negative argument");
        return GPDE_EINVAL;
    }
    return fwd_impl(x, n_nodes, nullptr, n_edges, rowptr, src, dst, nullptr, n_layers, dims, packed, root, bias,
                    aggr, 0, hidden, out, ws, ws_bytes, (hipStream_t)stream_, 0, nullptr, hidden_absmax);
}

extern "C" int gpde_nnconv_fwd_act(const float* x, int64_t n_nodes, const float* edge_attr, int64_t n_edges,
                                   const int32_t* rowptr, const int32_t* src, const int32_t* dst, const int32_t* perm,
                                   int n_layers, const int32_t* dims, const void* packed, const float* root,
                                   const float* bias, int aggr, uint32_t flags, const float* residual, int relu_out,
                                   float* out, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out)) ||
        (n_edges > 0 && (!edge_attr || !src || !dst || !perm)) || (residual && residual == out)) {
        gpde_set_error("gpde_nnconv_fwd_act: null/negative argument (or residual aliases out)");
        return GPDE_EINVAL;
    }
    return fwd_impl(x, n_nodes, edge_attr, n_edges, rowptr, src, dst, perm, n_layers, dims, packed, root, bias,
                    aggr, flags, nullptr, out, ws, ws_bytes, (hipStream_t)stream_, 0, nullptr, nullptr, -1, residual,
                    relu_out);
}

extern "C" int gpde_nnconv_fwd_hidden_act(const float* x, int64_t n_nodes, const float* hidden, const float* hidden_absmax,
                                          int64_t n_edges, const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                                          int n_layers, const int32_t* dims, const void* packed, const float* root,
                                          const float* bias, int aggr, const float* residual, int relu_out, float* out,
                                          void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out)) ||
        (n_edges > 0 && (!hidden || !src || !dst)) || (residual && residual == out)) {
        gpde_set_error("gpde_nnconv_fwd_hidden_act: null/negative argument (or residual aliases out)");
        return GPDE_EINVAL;
    }
    return fwd_impl(x, n_nodes, nullptr, n_edges, rowptr, src, dst, nullptr, n_layers, dims, packed, root, bias,
                    aggr, 0, hidden, out, ws, ws_bytes, (hipStream_t)stream_, 0, nullptr, hidden_absmax, -1, residual,
                    relu_out);
}

extern "C" int gpde_nnconv_fwd_keepz(const float* x, int64_t n_nodes, const float* edge_attr, const float* hidden,
                                     const float* hidden_absmax, int64_t n_edges, const int32_t* rowptr, const int32_t* src,
                                     const int32_t* dst, const int32_t* perm, int n_layers, const int32_t* dims,
                                     const void* packed, const float* root, const float* bias, int aggr, uint32_t flags,
                                     float* z_keep, float* out, void* ws, size_t ws_bytes, void* stream_) {
    if (n_nodes < 0 || n_edges < 0 || !dims || !packed || !rowptr || (n_nodes > 0 && (!x || !out || !z_keep)) ||
        (n_edges > 0 && (!src || !dst || (!hidden && (!edge_attr || !perm))))) {
        gpde_set_error("gpde_nnconv_fwd_keepz: null/negative argument");
        return GPDE_EINVAL;
    }
    if (hidden)
        return fwd_impl(x, n_nodes, nullptr, n_edges, rowptr, src, dst, nullptr, n_layers, dims, packed, root, bias, aggr, 0,
                        hidden, out, ws, ws_bytes, (hipStream_t)stream_, 0, nullptr, hidden_absmax, -1, nullptr, 0, z_keep);
    return fwd_impl(x, n_nodes, edge_attr, n_edges, rowptr, src, dst, perm, n_layers, dims, packed, root, bias, aggr, flags,
                    nullptr, out, ws, ws_bytes, (hipStream_t)stream_, 0, nullptr, nullptr, -1, nullptr, 0, z_keep);
}

extern "C" int gpde_profile_begin(void) {
    for (auto& p : g_prof) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    g_prof.clear();
    g_prof_on = true;
    return GPDE_OK;
}

extern "C" int gpde_profile_end_kinds(double* ms_by_kind, int32_t* launches_by_kind) {
    g_prof_on = false;
    for (int k = 0; k < GPDE_PROF_KINDS; ++k) {
        if (ms_by_kind) ms_by_kind[k] = 0.0;
        if (launches_by_kind) launches_by_kind[k] = 0;
    }
    for (auto& p : g_prof) {
        float ms = 0.f;
        GP_HIP_CHECK(hipEventSynchronize(p.b));
        GP_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        const int k = (p.kind >= 0 && p.kind < GPDE_PROF_KINDS) ? p.kind : GPDE_PROF_KINDS - 1;
        if (ms_by_kind) ms_by_kind[k] += ms;
        if (launches_by_kind) launches_by_kind[k] += 1;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    g_prof.clear();
    return GPDE_OK;
}

