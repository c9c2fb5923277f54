/*
 * gpde.h — C ABI of libgpde.so: the MI355X (gfx950) edge-conditioned graph convolution
 * (NNConv) hot path of neuraloperator/graph-pde.
 *
 * The reference has no FFI: the path sits behind a Python `torch.nn.Module`
 * (`nn_conv.NNConv_old`, /root/reference/graph-neural-operator/nn_conv.py:197-286, and
 * `torch_geometric.nn.NNConv`) whose `forward(x, edge_index, edge_attr)` dispatches stock
 * PyTorch / torch_scatter kernels.  These entry points are what a binding for that path binds
 * (INTEGRATION.md shows the ctypes stub); each one names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless marked "host"; buffers are row-major, contiguous,
 *    float32 / int32 / int64 exactly as named; `edge_index` alone carries explicit strides;
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream),
 *    re-entrant, never synchronise, never allocate: the caller supplies the workspace
 *    (query with the *_workspace_bytes functions);
 *  - return value: GPDE_OK or a negative GPDE_E* code; gpde_last_error() (thread-local) gives
 *    the message.  Nothing throws across the ABI.
 */
#ifndef GPDE_H
#define GPDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPDE_VERSION 101 /* 0.1.1: + gpde_nnconv_bwd_edgeweights_acc */
/* gpde_version() of a developer build carries one of these on top of GPDE_VERSION: an ABLATION build has part of the arithmetic
 * compiled out (timing experiments; its results are WRONG and a binding must refuse it), an INSTRUMENTED build carries
 * clock probes (results correct). */
#define GPDE_VERSION_ABLATION 0x10000
#define GPDE_VERSION_INSTRUMENTED 0x20000

/* libgpde.so is built with -fvisibility=hidden: the entry points declared here are its ONLY dynamic symbols (internal
 * launchers and kernels stay out of the dynamic symbol table; tests/test_abi.py diffs `nm -D` against this header). */
#define GPDE_API __attribute__((visibility("default")))

enum {
    GPDE_OK = 0,
    GPDE_EINVAL = -1,       /* bad argument (null pointer, negative size, ...) */
    GPDE_EUNSUPPORTED = -2, /* shape outside what the kernels implement */
    GPDE_EWORKSPACE = -3,   /* workspace too small */
    GPDE_EHIP = -4          /* a HIP runtime call / kernel launch failed */
};

enum { GPDE_AGGR_ADD = 0, GPDE_AGGR_MEAN = 1,
       GPDE_AGGR_MAX = 2 /* gpde_nnconv_fwd_edgeweights_group only: 'max' cannot use the re-associated contraction */ };

/* gpde_nnconv_fwd flags */
enum {
    GPDE_FWD_DEFAULT = 0,  /* every contraction on v_mfma_f32_32x32x2_f32: exact fp32 (fmaf chains) */
    GPDE_FWD_F16SPLIT = 1, /* hidden k1 x k2 layer on f16 MFMA with two-term operand splitting
                              (x = hi + lo, 3 MFMAs, fp32 accumulate; per-product error < 2^-21,
                              DESIGN.md §3b); ignored for kernels without a hidden GEMM */
    GPDE_FWD_F16SPLIT_8WAVE = 2, /* with F16SPLIT: force the 8-wave kernel (gpde_fused_f16v3_kernel: the path of small
                                    graphs, node-table attributes and gpde_hidden_fwd) where the default would be the
                                    one-wave-per-SIMD kernel gpde_fused_f16v6_kernel (A/B, parity tests) */
    GPDE_FWD_STATIC_RANGES = 4,  /* gpde_fused_f16v6_kernel: one contiguous edge range per wave instead of the block work queue
                                    (A/B of the x_j gather locality, parity tests) */
    GPDE_FWD_AGG_F16 = 16,       /* with F16SPLIT: aggregation x_j (x) h_e on split-f16 MFMA too, also for small
                                    graphs (default: from 32768 edges on; three tiny pre-pass launches) */
    GPDE_FWD_AGG_F32 = 32,       /* with F16SPLIT: keep the aggregation on fp32 MFMA (A/B) */
    GPDE_FWD_NO_EDGE_PATH = 64   /* never take the per-edge last layer of low in-degree graphs (mean in-degree <= 4,
                                  * >= 4096 edges, k2 >= 256: W_e = W3 . h_e on the split-f16 GEMM, contracted with x_j
                                  * in its epilogue) - A/B against the re-associated path */
};

#define GPDE_MAX_LAYERS 8
#define GPDE_WIDTH 64 /* node-feature width (in_channels == out_channels) the kernels are built for */

/* Edge attributes described by NODE data instead of an [E][k0] tensor (details at gpde_nnconv_fwd_mixed_keepz):
 * slot d of edge (j -> i) is table[(sel[d] >> 8 ? i : j) * stride + (sel[d] & 255)].  A HOST struct; every entry point that takes
 * `edge_attr` + `perm` also takes a `const GpdeNodeAttr* node_attr` (NULL = the tensor) - round 5 folded the `_na` twins into it. */
typedef struct GpdeNodeAttr {
    const float* table;   /* device [n_nodes][stride] */
    int32_t stride;
    int32_t n_slots;      /* = dims[0], 1..7 */
    int32_t sel[8];       /* slot d: endpoint << 8 | column (endpoint 0 = source j, 1 = target i) */
} GpdeNodeAttr;

GPDE_API int gpde_version(void);
GPDE_API const char* gpde_last_error(void);
/* The library's developer / A-B switches (GPDE_BWD_*, GPDE_EDGE_BWD, GPDE_DEBUG_SKEW_US, ...: INTEGRATION.md) are environment
 * variables read ONCE, at the first native call of the process - no launch path calls getenv().  This re-reads them (test
 * suites and A/B scripts that flip a switch inside one process). */
GPDE_API int gpde_reload_switches(void);

/* ---------------------------------------------------------------------------------------------
 * Destination-sorted CSR of a COO edge list.
 * Replaces what `MessagePassing.propagate` does implicitly on every call: the gather by
 * `edge_index[0]` and the scatter by `edge_index[1]` (call site nn_conv.py:271; PyG semantics in
 * SURVEY.md Appendix B).  The sort is STABLE: within one destination the edges keep their input
 * order, so the per-destination summation order is fixed run to run.
 *
 *   edge_index : int64, element (r, e) at edge_index[r*stride_row + e*stride_col]
 *                (row 0 = source j, row 1 = target i; may be a strided view)
 *   rowptr[N+1]: in-edges of node i are CSR slots rowptr[i] .. rowptr[i+1]-1
 *   src[E], dst[E] : source / target node of each CSR slot;  perm[E] : its original edge id
 *   n_bad      : int32 device word, receives the number of edges with an endpoint outside [0,N)
 *                (those edges are dropped from the CSR; the reference would raise IndexError)
 */
GPDE_API size_t gpde_csr_workspace_bytes(int64_t n_edges, int64_t n_nodes);
GPDE_API int gpde_csr_from_coo(const int64_t* edge_index, int64_t stride_row, int64_t stride_col,
                      int64_t n_edges, int64_t n_nodes, int32_t* rowptr, int32_t* src,
                      int32_t* dst, int32_t* perm, int32_t* n_bad, void* ws, size_t ws_bytes,
                      void* stream);
/* out[s][0..k) = rows[perm[s]][0..k) for the n CSR slots: a per-edge tensor (edge_attr [E][k0], the `pseudo` of
 * nn_conv.py:271) laid out by CSR slot, once per (graph, tensor), so that the fused kernels stream it instead of chasing
 * perm (8 column slices x one cache line per edge otherwise).  Same values: results are bit-identical. */
GPDE_API int gpde_gather_rows(const float* rows, int k, const int32_t* perm, int64_t n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel-MLP weights, repacked once per parameter update into MFMA-tile order.
 * Replaces nothing arithmetic: it is the layout step for `DenseNet`'s `nn.Linear` weights
 * (/root/reference/graph-neural-operator/utilities.py:201-227; torch layout weight[out][in]).
 *
 *   n_layers      : number of Linear layers (>= 2); ReLU between them, none after the last
 *   dims[n_layers+1] (host): k0, k1, ..., 4096 (= GPDE_WIDTH^2)
 *   W[l], b[l] (host arrays of device pointers): weight [dims[l+1]][dims[l]], bias [dims[l+1]]
 *                (b[l] may be NULL = no bias)
 *   packed        : device buffer of gpde_mlp_pack_bytes() bytes
 */
GPDE_API size_t gpde_mlp_pack_bytes(int n_layers, const int32_t* dims);
GPDE_API int gpde_mlp_pack(int n_layers, const int32_t* dims, const float* const* W,
                  const float* const* b, void* packed, size_t packed_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused NNConv forward:  out[i] = aggr_{e: j->i} ( x[j] . reshape(mlp(edge_attr[e]), [64,64]) )
 *                                 + x[i] . root + bias
 * Replaces NNConv_old.forward / message / update (nn_conv.py:267-282), DenseNet.forward
 * (utilities.py:223-227) and PyG's gather + scatter-add/mean in one pass
 * that never materialises the [E,4096] per-edge weight tensor.
 *
 *   x [N][64], edge_attr [E][k0] in ORIGINAL edge order (gathered through perm),
 *   rowptr/src/dst/perm from gpde_csr_from_coo, packed from gpde_mlp_pack (same n_layers/dims),
 *   root [64][64] or NULL, bias [64] or NULL, aggr = GPDE_AGGR_ADD | GPDE_AGGR_MEAN
 *   (mean = sum / max(in_degree,1): zero in-degree rows receive only root/bias terms),
 *   out [N][64] (fully overwritten).
 *   ws: gpde_nnconv_fwd_workspace_bytes() is the recommended size; any size that holds one
 *   64-node tile works (more workspace = more destination nodes per launch), else GPDE_EWORKSPACE.
 */
GPDE_API size_t gpde_nnconv_fwd_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers,
                                       const int32_t* dims);
GPDE_API int gpde_nnconv_fwd(const float* x, int64_t n_nodes, const float* edge_attr, int64_t n_edges,
                    const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                    const int32_t* perm, int n_layers, const int32_t* dims, const void* packed,
                    const float* root, const float* bias, int aggr, uint32_t flags, float* out,
                    void* ws, size_t ws_bytes, void* stream);

/* gpde_nnconv_fwd / gpde_nnconv_fwd_hidden with the callers' elementwise glue fused into the last kernel
 * (opt-in, the module surface does not change):
 *     out = act(residual + NNConv(x))        residual [N][64] or NULL (may alias x, must not alias out),
 *                                            relu_out != 0: act = max(., 0), else identity
 * - `F.relu(x + conv(x, ...))` of the MGKN V-cycles (MGKN_general_darcy2d.py:79-80,89-90;
 * MGKN_orthogonal_burgers1d.py:74-82) and `F.relu(conv(...))` of KernelNN.forward (UAI1_full_resolution.py:30). */
GPDE_API int gpde_nnconv_fwd_act(const float* x, int64_t n_nodes, const float* edge_attr, int64_t n_edges,
                        const int32_t* rowptr, const int32_t* src, const int32_t* dst, const int32_t* perm,
                        int n_layers, const int32_t* dims, const void* packed, const float* root,
                        const float* bias, int aggr, uint32_t flags, const float* residual, int relu_out,
                        float* out, void* ws, size_t ws_bytes, void* stream);
GPDE_API int gpde_nnconv_fwd_hidden_act(const float* x, int64_t n_nodes, const float* hidden, const float* hidden_absmax,
                               int64_t n_edges, const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                               int n_layers, const int32_t* dims, const void* packed, const float* root,
                               const float* bias, int aggr, const float* residual, int relu_out, float* out,
                               void* ws, size_t ws_bytes, void* stream);

/* Launch plan gpde_nnconv_fwd will follow for these sizes and this workspace (host-side query, no
 * device work): number of destination-node chunks, nodes per chunk, workgroups of the fused
 * kernel per chunk, and which fused variant runs (0: one hidden layer, 1: two hidden layers with
 * the first generated on the fly, 2: hidden activations precomputed by dense front layers). */
GPDE_API int gpde_nnconv_fwd_plan(int64_t n_nodes, int64_t n_edges, int n_layers, const int32_t* dims,
                         size_t ws_bytes, int32_t* n_chunks, int64_t* nodes_per_chunk,
                         int32_t* fused_workgroups, int32_t* mode);

/* Symbol (without template arguments) of the fused edge kernel gpde_nnconv_fwd launches for a graph of
 * `n_edges` edges, this kernel MLP and these flags: "gpde_fused_f16v6_kernel" (one wave per SIMD, 64 x 128
 * wave tile; GPDE_FWD_F16SPLIT from 32768 edges on), "gpde_fused_f16v3_kernel" (8 waves; smaller graphs,
 * GPDE_FWD_F16SPLIT_8WAVE) or "gpde_fused_kernel" (fp32 MFMA; 2-Linear and >= 4-Linear kernels).  Host-side
 * query, no device work; static storage.  bench.py keys its rocprofv3 / PMC records on this name.
 * It names the kernel of the RE-ASSOCIATED path; a low in-degree graph (mean in-degree <= 4, >= 4096 edges, k2 >= 256,
 * one node chunk: DESIGN.md §3e) runs the store variant of the same kernel family plus gpde_gemm_f16s_nt_kernel instead -
 * whether it does depends on the node count and the workspace, which this query does not see. */
GPDE_API const char* gpde_nnconv_fwd_kernel(int64_t n_edges, int n_layers, const int32_t* dims, uint32_t flags);

/* ---------------------------------------------------------------------------------------------
 * Backward of the fused NNConv (what autograd computes through nn_conv.py:267-282,
 * utilities.py:223-227 and PyG's gather/scatter on `loss.backward()`,
 * UAI1_full_resolution.py:266).  Given grad_out = dL/d(out) [N][64] it writes (overwrites)
 *   grad_x [N][64], grad_W[l] [dims[l+1]][dims[l]], grad_b[l] [dims[l+1]] (torch layouts; entries
 *   or whole arrays may be NULL to skip), grad_root [64][64], grad_bias [64] (NULL to skip).
 * W, b, grad_W, grad_b are HOST arrays of device pointers; rowptr_host is a HOST copy of rowptr
 * (the edge chunks are planned on the host).  Hidden activations are recomputed per node-aligned
 * chunk of edges; nothing from the forward needs to be saved (z_saved is an option, below).  fp32-class
 * arithmetic throughout (fp32 MFMA; split-f16 MFMA at < 2^-21 per product where DESIGN.md §6b says so). */
GPDE_API size_t gpde_nnconv_bwd_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers,
                                       const int32_t* dims);
/* The workspace with which the backward runs ALL edges and nodes as ONE chunk (the call above caps its answer at ~26 GB and
 * the backward then walks node-aligned chunks of ~640 k edges at k = 1024).  Any size in between is accepted and gives
 * proportionally fewer chunks; fewer chunks are faster (s=121, 5.9 M edges: 152 ms with 10 chunks, 147 ms with one) - a
 * caller with memory to spare may pass up to this many bytes (ops.py: when it is below GPDE_BWD_WS_FRACTION of the free device
 * memory). */
GPDE_API size_t gpde_nnconv_bwd_workspace_bytes_one_chunk(int64_t n_nodes, int64_t n_edges, int n_layers,
                                                 const int32_t* dims);
enum { GPDE_BWD_ACCUMULATE_GRAD_HIDDEN = 1 /* gpde_nnconv_bwd `flags`, `hidden` form: grad_hidden += dL/dU instead of = (see below) */ };
GPDE_API int gpde_nnconv_bwd(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr,
                    const float* hidden, int64_t n_edges, const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                    const int32_t* perm, const int32_t* rowptr_host, const int32_t* src_rowptr, const int32_t* src_slots,
                    int n_layers, const int32_t* dims, const float* const* W, const float* const* b, const float* root,
                    int aggr, const float* grad_out, const float* z_saved, float* grad_x, float* grad_hidden,
                    float* grad_edge_attr, float* const* grad_W, float* const* grad_b, float* grad_root,
                    float* grad_bias, uint32_t flags, void* ws, size_t ws_bytes, void* stream);
/* ^ ONE backward entry point for every form of the operator (round 5 folded seven: the plain, source-ordered, kept-Z,
 *   given-hidden, edge-attribute-gradient and node-table calls of rounds 1-4 were this call with some arguments fixed):
 *   attribute source   edge_attr + perm (tensor, caller's edge order) | node_attr (node data) | hidden (the last hidden
 *                      activations given, [CSR slot][K2P], e.g. from gpde_hidden_fwd: only x, the LAST Linear (W / b / grad_W /
 *                      grad_b carry their last entries), root and bias are differentiated, and dL/dU of the last hidden layer
 *                      is written to grad_hidden [CSR slot][K2P] for gpde_hidden_bwd).  Exactly one of the three - except:
 *   hidden + an attribute source, grad_hidden == NULL   the FULL backward (every layer differentiated) with the last hidden
 *                      activations KEPT BY THE FORWARD (gpde_hidden_fwd, then gpde_nnconv_fwd_keepz on them): read where they would
 *                      be recomputed - the K loop of the hidden layer runs once per training step instead of twice, for 4 KiB
 *                      per edge of memory between forward and backward (the host mirror keeps them when they fit, GPDE_SAVE_H_GB).
 *   src_rowptr, src_slots  nullable: the CSR slots regrouped by source node (gpde_csr_source_order) - grad_x is then summed per
 *                      source in slot order, bit-reproducible and independent of the chunking; NULL: fp32 atomics.
 *   z_saved            nullable: the Z buffer of gpde_nnconv_fwd_keepz / _mixed_keepz - dW_3 is taken from it.
 *   flags              0, or GPDE_BWD_ACCUMULATE_GRAD_HIDDEN (the `hidden` form): dL/dU is ADDED to grad_hidden.  A module applied
 *                      `depth` times on shared hidden activations (UAI1_full_resolution.py:29-30) sums the dL/dU of its applications
 *                      before gpde_hidden_bwd: the first application's backward writes the tensor, the others add to it in the per-edge
 *                      kernel, in call order - the additions autograd would make with `depth` [E][K2P] tensors, without the tensors
 *                      (built into the split-f16 per-edge kernel; GPDE_EUNSUPPORTED elsewhere).
 *   grad_hidden        the `hidden` form only.   grad_edge_attr  nullable, tensor form with <= 8 slots only: dL/d edge_attr
 *                      [E][k0] in the caller's edge order (what autograd hands `pseudo`, nn_conv.py:273-275; no reference
 *                      script asks for it).
 * On grad_x: without the source order dx_j is accumulated over the out-edges of j by fp32 atomics (run-to-run differences at
 * the 1e-7 level, like the reference's index_select backward on a GPU); with it the per-edge contributions are written out
 * and summed per source in ascending slot order by one owner per element.  Weight gradients are ordered either way. */
/* src_slots = CSR slots 0..E-1 stably sorted by their source node, src_rowptr[j] = first position of source j;
 * `src` is the array gpde_csr_from_coo wrote; workspace: gpde_csr_workspace_bytes(n_edges, n_nodes). */
GPDE_API int gpde_csr_source_order(const int32_t* src, int64_t n_edges, int64_t n_nodes, int32_t* src_rowptr,
                          int32_t* src_slots, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cross-depth reuse of the kernel MLP.  The reference applies ONE NNConv
 * module `depth` times per forward with the same edge_attr and the same weights
 * (UAI1_full_resolution.py:29-30 `for k in range(self.depth): x = F.relu(self.conv1(x, ...))`;
 * MGKN_general_darcy2d.py:76-90, MGKN_orthogonal_burgers1d.py:65-82), so the hidden activation
 *   H_e = relu(L_{n-1}(... relu(L_1(edge_attr_e))))     (DenseNet.forward, utilities.py:223-227,
 *                                                         all but the last Linear)
 * is identical in every one of those calls.  The operator splits into
 *   gpde_hidden_fwd:         edge_attr -> H            [E][K2P] fp32, rows in CSR (destination-sorted)
 *                            order, K2P = dims[n_layers-1] rounded up to 128, padding columns zero
 *   gpde_nnconv_fwd_hidden:  (x, H) -> out             aggregation + last Linear + update()
 * and, for training, their backward halves
 *   gpde_nnconv_bwd(hidden): grad_out -> grad_x, grad of the last Linear / root / bias, and
 *                            grad_hidden = dL/dU of the last hidden layer (already multiplied by the
 *                            ReLU mask H > 0), [E][K2P], overwritten
 *   gpde_hidden_bwd:         grad_hidden (summed over the `depth` uses by the caller's autograd)
 *                            -> gradients of the hidden Linear layers 0 .. n_layers-2 (grad_W / grad_b
 *                            entries of the last layer are ignored)
 * so the k1 x k2 layer (94 % of the FLOPs) and its backward run once per step instead of `depth`
 * times.  The caller owns H (E * K2P * 4 bytes) and decides whether it fits.
 *
 * gpde_hidden_fwd: with flags & GPDE_FWD_F16SPLIT, `packed` (gpde_mlp_pack) and a 3-Linear MLP the
 * fused f16-split kernel computes H (same arithmetic as gpde_nnconv_fwd); otherwise the layers run
 * as fp32-MFMA GEMMs over chunks of edges and W, b (HOST arrays of device pointers, torch layouts)
 * and ws (gpde_hidden_workspace_bytes) are required.  Workspaces: gpde_nnconv_fwd_hidden ->
 * gpde_nnconv_fwd_workspace_bytes; gpde_nnconv_bwd -> gpde_nnconv_bwd_workspace_bytes; gpde_hidden_bwd:
 * gpde_nnconv_bwd_workspace_bytes(0, E, ...).
 * hidden_absmax (nullable, one device float): gpde_hidden_fwd records max |H| there when its fused
 * path computed H (the general path leaves 0: the value is then NOT a maximum and must not be
 * passed on).  Given back to gpde_nnconv_fwd_hidden it lets the aggregation run on split-f16 MFMA
 * from 32768 edges on; NULL: fp32 MFMA. */
GPDE_API size_t gpde_hidden_workspace_bytes(int64_t n_edges, int n_layers, const int32_t* dims);
GPDE_API int gpde_hidden_fwd(const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* rowptr,
                    int64_t n_nodes, const int32_t* perm, const int32_t* src, const int32_t* dst, int n_layers,
                    const int32_t* dims, const void* packed, const float* const* W, const float* const* b, uint32_t flags,
                    float* hidden, float* hidden_absmax, void* ws, size_t ws_bytes, void* stream);
/* ^ node_attr != NULL: attributes from node data (src / dst required, edge_attr / perm / W / b / ws unused; 3-Linear kernel MLPs of
 *   >= 8 k1 chunks on GPDE_FWD_F16SPLIT); else src / dst may be NULL. */
GPDE_API int gpde_nnconv_fwd_hidden(const float* x, int64_t n_nodes, const float* hidden,
                           const float* hidden_absmax, int64_t n_edges,
                           const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                           int n_layers, const int32_t* dims, const void* packed, const float* root,
                           const float* bias, int aggr, float* out, void* ws, size_t ws_bytes,
                           void* stream);
/* attributes: edge_attr + perm, or node_attr + src + dst (as gpde_hidden_fwd) */
GPDE_API int gpde_hidden_bwd(const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* perm,
                    const int32_t* src, const int32_t* dst, int n_layers,
                    const int32_t* dims, const float* const* W, const float* const* b,
                    const float* grad_hidden, float* const* grad_W, float* const* grad_b, void* ws,
                    size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Keep-Z pair (training): the forward already forms Z_i = sum_{e -> i} x_j (x) h_e for every node and the
 * backward's dW_3 = sum_i gT_i (x) Z_i needs exactly that; without it the backward re-aggregates it from
 * the recomputed (or given) hidden activations.  gpde_nnconv_fwd_keepz = gpde_nnconv_fwd (hidden == NULL) or
 * gpde_nnconv_fwd_hidden (hidden != NULL; edge_attr / perm unused) writing Z into z_keep [N][64][K2P] (K2P = last hidden
 * width padded to 128; ZERO-INITIALISED by the caller: nodes without in-edges are not written); the per-edge last layer of
 * low in-degree graphs is not taken.  The backward takes that buffer as `z_saved` (gpde_nnconv_bwd, gpde_nnconv_bwd_light). */
GPDE_API int gpde_nnconv_fwd_keepz(const float* x, int64_t n_nodes, const float* edge_attr, const float* hidden,
                          const float* hidden_absmax, int64_t n_edges, const int32_t* rowptr, const int32_t* src,
                          const int32_t* dst, const int32_t* perm, int n_layers, const int32_t* dims, const void* packed,
                          const float* root, const float* bias, int aggr, uint32_t flags, float* z_keep, float* out,
                          void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depth-deferred backward (SURVEY.md §8 rows f1 x f4 when the hidden activations do NOT fit memory: H of the 241^2 graph is
 * 391 GB).  `KernelNN.forward` applies the SAME conv1 `depth` times (UAI1_full_resolution.py:29-30) and `loss.backward()`
 * (:266) sums the kernel MLP's gradients over those applications.  With dU_2^(l)[e] = (x_j^(l) . dZ_i^(l)) (.) [H_2[e] > 0]
 * and H_2 the same in every application, the sum over l can be taken BEFORE the hidden layers' backward:
 *   dU_2[e][k] = (sum_l sum_c x_j^(l)[c] dZ_i^(l)[c][k]) (.) [H_2[e][k] > 0]       one K = 64 * depth contraction per edge
 * so the two 1024 x 1024 GEMMs per edge (dU_1, dW_2), the transposes and dW_1 / db_* run ONCE per step.
 *   gpde_nnconv_bwd_light     one application: grad_x, grad of the last Linear (grad_w_last [4096][k2], grad_b_last [4096]),
 *                             grad_root, grad_bias - everything gpde_nnconv_bwd writes except the hidden layers'
 *                             gradients.  z_saved: NULL or the keep-Z forward's buffer (gpde_nnconv_fwd_keepz).
 *                             Workspace: gpde_nnconv_bwd_workspace_bytes.
 *   gpde_nnconv_bwd_deferred  all applications: x_stack [Lp][N][64] = the inputs x^(l) of the n_defer applications, ZERO
 *                             layers appended up to Lp = max(4, n_defer rounded up to even); grad_out_stack [n_defer][N][64] =
 *                             their output gradients (any order, the same in both stacks).  Writes grad_W[l] / grad_b[l] of
 *                             the hidden layers l = 0 .. n_layers - 2 (the last entries are ignored).  Workspace:
 *                             gpde_nnconv_bwd_deferred_workspace_bytes.
 * hidden_part / hidden_nodes (both entry points; NULL / 0: none): the last hidden activations of the in-edges of nodes
 * [0, hidden_nodes) as gpde_hidden_fwd(n_nodes = hidden_nodes) wrote them (the partial H of gpde_nnconv_fwd_mixed_keepz) - node
 * chunks below that bound read them instead of recomputing the hidden chain (the 241^2 graph: 44 % of the edges fit 170 GB).
 * Built for the kernel MLPs gpde_nnconv_bwd_deferred_supported() accepts (3 Linear layers, hidden widths multiples of 128,
 * at most 7 attributes: the split-f16 path); others return GPDE_EUNSUPPORTED - use gpde_nnconv_bwd per application.
 * Results: grad_x etc. of the light pass are the bits of gpde_nnconv_bwd (source-ordered); the hidden layers' gradients equal the
 * sum of the per-application ones up to fp32 / split-f16 summation order (tests/test_gpu_deferred.py: <= 2e-5). */
GPDE_API int gpde_nnconv_bwd_deferred_supported(int n_layers, const int32_t* dims);
GPDE_API size_t gpde_nnconv_bwd_deferred_workspace_bytes(int64_t n_nodes, int64_t n_edges, int n_layers, const int32_t* dims,
                                                int n_defer);
GPDE_API int gpde_nnconv_bwd_light(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr,
                          int64_t n_edges, const int32_t* rowptr,
                          const int32_t* src, const int32_t* dst, const int32_t* perm, const int32_t* rowptr_host,
                          const int32_t* src_rowptr, const int32_t* src_slots, int n_layers, const int32_t* dims,
                          const float* const* W, const float* const* b, const float* root, int aggr, const float* grad_out,
                          const float* z_saved, const float* hidden_part, int64_t hidden_nodes, float* grad_x,
                          float* grad_w_last, float* grad_b_last, float* grad_root,
                          float* grad_bias, void* ws, size_t ws_bytes, void* stream);
GPDE_API int gpde_nnconv_bwd_deferred(const float* x_stack, const float* grad_out_stack, int n_defer, int64_t n_nodes,
                             const float* edge_attr, const GpdeNodeAttr* node_attr, int64_t n_edges, const int32_t* rowptr,
                             const int32_t* src, const int32_t* dst, const int32_t* perm, const int32_t* rowptr_host,
                             int n_layers, const int32_t* dims, const float* const* W, const float* const* b, int aggr,
                             const float* hidden_part, int64_t hidden_nodes,
                             float* const* grad_W, float* const* grad_b, void* ws, size_t ws_bytes, void* stream);
/* The GENERAL forward.  Mixed call for graphs whose H does not fit memory (E * K2P * 4 bytes: 391 GB at the 241^2 graph):
 * `hidden` holds the rows of the in-edges of nodes [0, hidden_nodes) only (CSR slots [0, rowptr[hidden_nodes]); build it with
 * gpde_hidden_fwd(..., n_edges = rowptr[hidden_nodes], n_nodes = hidden_nodes)); those nodes aggregate from it, all others run the
 * fused kernel on the attributes.  Same result as gpde_nnconv_fwd.  3-Linear kernel MLPs. */
GPDE_API int gpde_nnconv_fwd_mixed_keepz(const float* x, int64_t n_nodes, const float* edge_attr, const GpdeNodeAttr* node_attr,
                                const float* hidden, const float* hidden_absmax, int64_t hidden_nodes, int64_t n_edges,
                                const int32_t* rowptr, const int32_t* src, const int32_t* dst,
                                const int32_t* perm, int n_layers, const int32_t* dims, const void* packed,
                                const float* root, const float* bias, int aggr, uint32_t flags, float* z_keep, float* out,
                                void* ws, size_t ws_bytes, void* stream);
/* ^ attributes from edge_attr + perm or from node_attr (then edge_attr / perm may be NULL); hidden_nodes = 0 / hidden = NULL: no
 *   partial H; z_keep (as in gpde_nnconv_fwd_keepz) = NULL: Z not kept.  (Round 5 folded the mixed, node-table and node-table +
 *   partial-H forwards of round 4 into this call.) */

/* ---------------------------------------------------------------------------------------------
 * The operator given the PER-EDGE WEIGHTS (second half; row a6 'max').
 * `weight = self.nn(pseudo).view(-1, in, out)` (nn_conv.py:274) is what the reference forms on every call.  For the
 * MGKN V-cycles' low in-degree / small graphs (MGKN_orthogonal_burgers1d.py:73-82: 2-3 in-edges per node;
 * MGKN_general_darcy2d.py:76-90: coarse levels of a few thousand edges) the same module runs `depth` times per forward
 * with the same edge_attr and weights, so that tensor is the same in every call: gpde_edge_weights_fwd builds it once
 * from the hidden activations of gpde_hidden_fwd ([E][4096] fp32 in CSR slot order, the last Linear's bias folded in),
 * and gpde_nnconv_fwd_edgeweights_group then runs any number of INDEPENDENT calls in one launch each doing gather,
 * message (nn_conv.py:275), aggregation (add / mean / max) and update() (nn_conv.py:277-282; + opt-in residual and
 * ReLU, the callers' `relu(x + conv(x))` glue) in a single streaming kernel (forward; its backward pair follows below). */
GPDE_API size_t gpde_edge_weights_workspace_bytes(int64_t n_edges, int n_layers, const int32_t* dims);
GPDE_API int gpde_edge_weights_fwd(const float* hidden /* [E][K2P], gpde_hidden_fwd */, int64_t n_edges, int n_layers,
                          const int32_t* dims, const void* packed /* gpde_mlp_pack image (k2 padded >= 256), else unused */,
                          const float* w_last /* [4096][k2] */, const float* b_last /* [4096] or NULL */,
                          float* edge_weights /* [E][4096] */, void* ws, size_t ws_bytes, void* stream);
#define GPDE_WECONV_MAX_GROUP 16 /* descriptors per launch; longer lists take several launches */
typedef struct GpdeWeConvDesc {
    const float* x;            /* [n_nodes][64] */
    const float* edge_weights; /* [E][4096] from gpde_edge_weights_fwd, CSR slot order */
    const int32_t* rowptr;     /* [n_nodes + 1] destination CSR (gpde_csr_from_coo) */
    const int32_t* src;        /* [E] source node per CSR slot */
    const float* root;         /* [64][64] or NULL */
    const float* bias;         /* [64] or NULL */
    const float* residual;     /* [n_nodes][64] or NULL: out = act(residual + NNConv(x)) */
    float* out;                /* [n_nodes][64] */
    int32_t n_nodes;
    int32_t aggr;              /* GPDE_AGGR_ADD | GPDE_AGGR_MEAN | GPDE_AGGR_MAX */
    int32_t relu;              /* 1: ReLU on the result */
    int32_t reserved;
} GpdeWeConvDesc;
GPDE_API int gpde_nnconv_fwd_edgeweights_group(const GpdeWeConvDesc* descs /* HOST array */, int n_descs, void* stream);

/* Training on the per-edge weights (both MGKN scripts are training scripts: MGKN_general_darcy2d.py:260-282,
 * MGKN_orthogonal_burgers1d.py:226-242).  W_e is an autograd node shared by the `depth` applications of a module:
 *   gpde_nnconv_bwd_edgeweights  backward of the operator given W_e: grad_x [N][64] (ordered over each source's out-edges
 *                                when src_rowptr / src_slots are given, else fp32 atomics), grad_edge_weights [E][4096]
 *                                = x_j (x) gT_i (what autograd forms for `weight` in nn_conv.py:274-275), grad_root,
 *                                grad_bias (NULL to skip).  'add' / 'mean'.
 *   gpde_edge_weights_bwd        backward of gpde_edge_weights_fwd given the SUM of grad_edge_weights over the applications:
 *                                grad_hidden [E][K2P] = (grad_W_e . W3) (.) [hidden > 0] (the input of gpde_hidden_bwd),
 *                                grad_w_last [4096][k2], grad_b_last [4096] - the two 4096 x k2 products per edge once per
 *                                step instead of once per application, on the split-f16 GEMMs. */
GPDE_API size_t gpde_nnconv_bwd_edgeweights_workspace_bytes(int64_t n_nodes, int64_t n_edges);
GPDE_API int gpde_nnconv_bwd_edgeweights(const float* x, int64_t n_nodes, const float* edge_weights, int64_t n_edges,
                                const int32_t* rowptr, const int32_t* src, const int32_t* src_rowptr,
                                const int32_t* src_slots, const float* root, int aggr, const float* grad_out,
                                float* grad_x, float* grad_edge_weights, float* grad_root, float* grad_bias, void* ws,
                                size_t ws_bytes, void* stream);
/* ... the same call adding into gradients that already hold the sum of the module's earlier applications of this backward pass
 * (`accumulate`: GPDE_ACC_* bits; 0 = gpde_nnconv_bwd_edgeweights).  What autograd's `grad = grad + new` does for a tensor used
 * by several applications (the reference's MGKN loops apply each NNConv `depth` times, MGKN_general_darcy2d.py:76-90): one
 * elementwise kernel per application and gradient - 16 KiB per edge read twice and written once for W_e.  In-kernel: the same
 * additions in the same order (unfused multiply, then add: the same bits), the old value read once. */
enum {
    GPDE_ACC_EDGE_WEIGHTS = 1,   /* grad_edge_weights[e] += x_j (x) gT_i */
    GPDE_ACC_ROOT = 2,           /* grad_root += X^T g */
    GPDE_ACC_BIAS = 4            /* grad_bias += colsum g */
};
GPDE_API int gpde_nnconv_bwd_edgeweights_acc(const float* x, int64_t n_nodes, const float* edge_weights, int64_t n_edges,
                                    const int32_t* rowptr, const int32_t* src, const int32_t* src_rowptr,
                                    const int32_t* src_slots, const float* root, int aggr, const float* grad_out,
                                    float* grad_x, float* grad_edge_weights, float* grad_root, float* grad_bias,
                                    int accumulate, void* ws, size_t ws_bytes, void* stream);
GPDE_API size_t gpde_edge_weights_bwd_workspace_bytes(int64_t n_edges, int n_layers, const int32_t* dims);
GPDE_API int gpde_edge_weights_bwd(const float* grad_edge_weights, const float* hidden, int64_t n_edges, int n_layers,
                          const int32_t* dims, const float* w_last, float* grad_hidden, float* grad_w_last,
                          float* grad_b_last, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Edge attributes on the fly (opt-in): the `node_attr` argument (GpdeNodeAttr, declared at the top).
 * The reference materialises edge_attr[e] = [pos_src(2), pos_dst(2), a_src, a_dst] from node data
 * (SquareMeshGenerator.attributes, utilities.py:274-277; multi_pole_grid1d, multipole utilities.py:1771-1777): 24 of the 40
 * algorithmic bytes per edge and a 2.3 GB tensor on the 241^2 graph.  With node_attr the kernels read them from a node table
 * instead: slot d of edge (j -> i) is
 *     table[(sel[d] >> 8 ? i : j) * stride + (sel[d] & 255)]        (endpoint 0 = source j, 1 = target i)
 * Same result as the call on the materialised tensor (bitwise: the attribute values are the same floats;
 * tests/test_gpu_nodeattr.py, test_gpu_nodeattr_train.py).  Taken by the general forward, gpde_hidden_fwd, gpde_nnconv_bwd,
 * gpde_nnconv_bwd_light, gpde_nnconv_bwd_deferred and gpde_hidden_bwd - a training step on a graph built from positions
 * (gpde_radius_csr_*) needs neither the [E][k0] tensor nor its slot-order copy.  Built for 3-Linear kernel MLPs of >= 8 k1
 * chunks on the default GPDE_FWD_F16SPLIT kernels, 1..7 slots; anything else returns GPDE_EUNSUPPORTED / GPDE_EINVAL. */

/* ---------------------------------------------------------------------------------------------
 * Radius graph on the GPU.  Replaces SquareMeshGenerator / RandomMeshGenerator.ball_connectivity
 * (utilities.py:250-255, 362-368: dense float64 pairwise_distances + np.where).  pos_src / pos_dst [n][dim]
 * float64 (dim 1..3).  Pass 1 writes the out-degree of every source; the caller forms the
 * exclusive prefix sum `offsets` [n+1] (int64) and allocates edge_index int64 [2][E], E =
 * offsets[n]; pass 2 fills it: edges (j in pos_src -> i in pos_dst) within distance r, self-loops included, sorted by
 * source then target (row-major np.where) - the reference's order.  Two point sets, because that is what
 * the inner (pos_dst == pos_src) and inter-level graphs of RandomMultiMeshGenerator.ball_connectivity are
 * (multipole-graph-neural-operator/utilities.py:602-640: pairwise_distances(X, Y) <= r).  The choice of
 * arithmetic: flags = 0 tests sum_k (dx_k)^2 <= r^2 exactly in float64 (symmetric graphs);
 * GPDE_RADIUS_REFERENCE_TIES evaluates scikit-learn's dot-product expansion operation by operation, so that pairs at
 * exactly distance r are kept or dropped as in the reference (its default s = 61, r = 0.10 graph: 376,471 edges).
 * pos_dst == pos_src with n_dst == n_src means ONE point set: self-loops, diagonal forced to distance 0. */
enum { GPDE_RADIUS_REFERENCE_TIES = 1 };
GPDE_API int gpde_radius_graph2_count(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                             double r, uint32_t flags, int32_t* deg, void* stream);
GPDE_API int gpde_radius_graph2_fill(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim,
                            double r, uint32_t flags, const int64_t* offsets, int64_t* edge_index, int64_t n_edges,
                            void* stream);

/* The same graphs by CELL LIST, emitted directly as the destination-sorted CSR the operator consumes (no int64 COO list,
 * no sort by destination; O(N * neighbourhood) tests instead of O(N^2)): the source points are binned into cells of
 * edge >= r, one wave per destination point walks the 3^dim neighbouring cells.  Pass 1 (count) builds the cell
 * structure in `ws` and writes the in-degree of every destination; the caller forms rowptr = exclusive scan (int32
 * [n_dst + 1]) and allocates src / dst int32 [E]; pass 2 (fill, same arguments, same untouched `ws`) writes them, each
 * row in ascending source order - exactly what gpde_csr_from_coo makes of the reference's source-major edge list
 * (rows longer than 4096 edges keep cell order).  lo / hi: HOST arrays [dim], a bounding box of both point sets.
 * Edge attributes for such a graph are addressed by CSR slot (perm = identity). */
GPDE_API size_t gpde_radius_csr_workspace_bytes(int64_t n_src, int dim, double r, const double* lo, const double* hi);
GPDE_API int gpde_radius_csr_count(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim, double r,
                          uint32_t flags, const double* lo, const double* hi, int32_t* deg, void* ws, size_t ws_bytes,
                          void* stream);
GPDE_API int gpde_radius_csr_fill(const double* pos_src, int64_t n_src, const double* pos_dst, int64_t n_dst, int dim, double r,
                         uint32_t flags, const double* lo, const double* hi, const int32_t* rowptr, int32_t* src,
                         int32_t* dst, int64_t n_edges, void* ws, size_t ws_bytes, void* stream);

/* HIP-event timing of the kernels launched by gpde_nnconv_fwd on the calling thread (used by
 * bench.py for the roofline figure; events are recorded on the same stream as the kernels).
 * gpde_profile_begin() arms it; gpde_profile_end_kinds() disarms it, SYNCHRONISES on the recorded
 * events and returns the summed duration (ms) and launch count per kernel kind: ms_by_kind / launches_by_kind are arrays of
 * GPDE_PROF_KINDS entries.  Not for production calls. */
GPDE_API int gpde_profile_begin(void);
enum {
    GPDE_PROF_FUSED = 0,     /* fused edge kernel (gpde_nnconv_fwd_kernel names it) or gpde_zagg_kernel */
    GPDE_PROF_GEMM3 = 1,     /* per-node last Linear Z . W3 */
    GPDE_PROF_EPILOGUE = 2,  /* split-K sum, b3, mean, x . root + bias */
    GPDE_PROF_PREP = 3,      /* pre-passes of the split-f16 aggregation (3 tiny kernels + memset) */
    GPDE_PROF_OTHER = 4,
    GPDE_PROF_KINDS = 5
};
GPDE_API int gpde_profile_end_kinds(double* ms_by_kind, int32_t* launches_by_kind);

#ifdef __cplusplus
}
#endif
#endif /* GPDE_H */
