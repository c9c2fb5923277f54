"""C-ABI checks that need no GPU: the library loads, exports every symbol include/gpde.h
declares, and its host-side queries / argument validation behave."""
import ctypes
import os
import re

import pytest

import graph_pde_amd as gp
from graph_pde_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(REPO, "include", "gpde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpde_[a-z0-9_]+)\s*\(", src)))


def _prototypes():
    """{name: (return C type, [argument C types])} parsed from include/gpde.h - the binding is checked against THIS, not against
    a hand-kept list (the ABI grew faster than hand lists)."""
    src = open(os.path.join(REPO, "include", "gpde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"GPDE_API\s+([\w \*]+?)\s*\b(gpde_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                t = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a       # drop the parameter name
                types.append(t.replace(" *", "*"))
        out[name] = (ret.replace(" *", "*"), types)
    return out


_SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32,
            "size_t": ctypes.c_size_t, "double": ctypes.c_double}


def _is_pointer_ctype(t):
    return t in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(t, type) and issubclass(t, ctypes._Pointer)


def test_header_and_binding_agree():
    declared = _declared_functions()
    assert declared, "no functions parsed from include/gpde.h"
    assert sorted(_lib.SIGNATURES) == declared
    assert sorted(_prototypes()) == declared, "a declaration without GPDE_API (it would not be exported: -fvisibility=hidden)"


def test_every_binding_signature_is_the_header_prototype():
    """Argument COUNT, scalar argument types (int / int32_t / int64_t / uint32_t / size_t / double: exact ctypes type) and
    pointer-ness of every argument and of the return value, for every entry point, generated from the header."""
    protos = _prototypes()
    for name, (res, args) in _lib.SIGNATURES.items():
        ret, ctypes_ = protos[name]
        assert len(args) == len(ctypes_), f"{name}: binding has {len(args)} arguments, header {len(ctypes_)}"
        if ret.endswith("*"):
            assert _is_pointer_ctype(res), (name, ret, res)
        else:
            assert res is _SCALARS[ret], (name, ret, res)
        for i, (ct, bt) in enumerate(zip(ctypes_, args)):
            if ct.endswith("*"):
                assert _is_pointer_ctype(bt), f"{name} argument {i}: header {ct}, binding {bt}"
            else:
                base = ct.replace("const ", "")
                assert base in _SCALARS, f"{name} argument {i}: unparsed C type {ct!r}"
                assert bt is _SCALARS[base], f"{name} argument {i}: header {ct}, binding {bt}"


def test_dynamic_symbols_are_exactly_the_header(tmp_path):
    """libgpde.so is built with -fvisibility=hidden: `nm -D` shows the GPDE_API entry points and nothing else of ours - no mangled
    internal launcher (round 4 exported 36 of them), no kernel stub."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if l.strip()]
    ours = sorted(s for s in syms if not s.startswith("__hip_cuid_"))        # (hipcc's per-translation-unit id words)
    assert ours == _declared_functions(), sorted(set(ours) ^ set(_declared_functions()))


def test_ablation_builds_are_refused(monkeypatch):
    """A library whose gpde_version() carries GPDE_VERSION_ABLATION (arithmetic compiled out: wrong results) is not loaded unless
    the experiment says GPDE_ALLOW_ABLATION=1; the package directory holds the production library only."""
    src = open(os.path.join(REPO, "include", "gpde.h")).read()
    assert int(re.search(r"#define GPDE_VERSION_ABLATION (0x[0-9a-f]+)", src).group(1), 16) == _lib.GPDE_VERSION_ABLATION
    assert int(re.search(r"#define GPDE_VERSION_INSTRUMENTED (0x[0-9a-f]+)", src).group(1), 16) == _lib.GPDE_VERSION_INSTRUMENTED
    pkg = os.path.dirname(_lib.__file__)
    assert [f for f in os.listdir(pkg) if f.endswith(".so")] == ["libgpde.so"], "developer builds belong under scripts/ubench/lib/"

    class Fake:
        def __getattr__(self, name):
            f = lambda *a: _lib.GPDE_VERSION | _lib.GPDE_VERSION_ABLATION
            return f
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(ctypes, "CDLL", lambda path: Fake())
    monkeypatch.delenv("GPDE_ALLOW_ABLATION", raising=False)
    with pytest.raises(_lib.GpdeError, match="ABLATION"):
        _lib.lib()
    monkeypatch.setenv("GPDE_ALLOW_ABLATION", "1")
    assert _lib.lib() is not None
    monkeypatch.setattr(_lib, "_lib", None)


def test_a_library_of_another_header_generation_is_refused(monkeypatch):
    """The ctypes binding is generated from include/gpde.h and every pointer is a void* to it: a libgpde.so built from another
    generation of the header (GPDE_LIB pointing at another tree) would be called with the wrong argument lists without a sound.
    gpde_version() must equal the header's GPDE_VERSION."""
    class Fake:
        def __getattr__(self, name):
            return lambda *a: _lib.GPDE_VERSION + 1
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(ctypes, "CDLL", lambda path: Fake())
    with pytest.raises(_lib.GpdeError, match="do not belong together"):
        _lib.lib()
    monkeypatch.setattr(_lib, "_lib", None)


def test_library_exports_every_declared_symbol():
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(l, name), f"libgpde.so does not export {name}"


def test_version_and_pack_queries():
    l = _lib.lib()
    assert l.gpde_version() == _lib.GPDE_VERSION == 101
    d = _lib.dims_array([6, 1024, 1024, 4096])
    nbytes = l.gpde_mlp_pack_bytes(3, d)
    # W1|b1 [1024+1][8] + W2 tiles (fp32 and f16-split) 2*1024*1024 + b2, ucol 2*1024 + W3 + B3
    # + the split image of W3 [4096][1024] and its 4096 row un-scales (per-edge last layer, k2 >= 256)
    assert nbytes == 4 * (1025 * 8 + 2 * (1024 * 1024 + 1024) + 1024 * 8 + 16 + 64 * 1024 * 64 + 4096 + 4096 * 1024 + 4096)
    # widths that are not tile multiples are padded (1000 -> K1P 1024 / K2P 1024, 500 -> 512)
    d = _lib.dims_array([6, 500, 1000, 4096])
    assert l.gpde_mlp_pack_bytes(3, d) == 4 * (513 * 8 + 2 * (1024 * 512 + 1024) + 512 * 8 + 16 + 64 * 1024 * 64 + 4096
                                                  + 4096 * 1024 + 4096)
    # last layer must emit width^2 values
    d = _lib.dims_array([6, 32, 100])
    assert l.gpde_mlp_pack_bytes(2, d) == 0
    assert b"width" in l.gpde_last_error()


def test_plan_modes_and_chunking():
    l = _lib.lib()
    i32, i64 = ctypes.c_int32, ctypes.c_int64
    nch, npc, wgs, mode = i32(), i64(), i32(), i32()
    d = _lib.dims_array([6, 1024, 1024, 4096])
    n, e = 58081, 95539625
    ws = l.gpde_nnconv_fwd_workspace_bytes(n, e, 3, d)
    assert ws > 0
    rc = l.gpde_nnconv_fwd_plan(n, e, 3, d, ws, ctypes.byref(nch), ctypes.byref(npc),
                                ctypes.byref(wgs), ctypes.byref(mode))
    assert rc == 0 and mode.value == 1
    assert nch.value * npc.value >= n and nch.value >= 1
    assert wgs.value % 8 == 0                      # 8 hidden slices of 128 columns
    # a 1 GiB workspace still works, with more chunks
    nch2 = i32()
    rc = l.gpde_nnconv_fwd_plan(n, e, 3, d, 1 << 30, ctypes.byref(nch2), ctypes.byref(npc),
                                ctypes.byref(wgs), ctypes.byref(mode))
    assert rc == 0 and nch2.value > nch.value
    # too small a workspace is an error, not a crash
    rc = l.gpde_nnconv_fwd_plan(n, e, 3, d, 1 << 20, ctypes.byref(nch2), ctypes.byref(npc),
                                ctypes.byref(wgs), ctypes.byref(mode))
    assert rc == -3 and b"workspace" in l.gpde_last_error()
    for dims, want in (([6, 64, 4096], 0), ([4, 16, 16, 4096], 1), ([6, 8, 16, 24, 24, 4096], 2),
                       ([9, 16, 4096], 2)):
        dd = _lib.dims_array(dims)
        rc = l.gpde_nnconv_fwd_plan(100, 1000, len(dims) - 1, dd, 1 << 30, ctypes.byref(nch),
                                    ctypes.byref(npc), ctypes.byref(wgs), ctypes.byref(mode))
        assert rc == 0 and mode.value == want, (dims, mode.value)


def test_argument_validation_without_gpu():
    l = _lib.lib()
    d = _lib.dims_array([6, 16, 4096])
    # null pointers / bad aggr are rejected before any device work
    rc = l.gpde_nnconv_fwd(None, 4, None, 0, None, None, None, None, 2, d, None, None, None, 1,
                           0, None, None, 0, None)
    assert rc == -1
    rc = l.gpde_csr_from_coo(None, 0, 0, -1, 4, None, None, None, None, None, None, 0, None)
    assert rc == -1
    rc = l.gpde_mlp_pack(2, d, None, None, None, 0, None)
    assert rc == -1
    # the split operator, the mixed forward and the node-table forward validate before touching the device
    d3 = _lib.dims_array([6, 16, 32, 4096])
    # gpde_hidden_fwd(edge_attr, node_attr, n_edges, rowptr, n_nodes, perm, src, dst, n_layers, dims, packed, W, b, flags, ...)
    assert l.gpde_hidden_fwd(None, None, 8, None, 4, None, None, None, 3, d3, None, None, None, 1, None, None, None, 0, None) == -1
    assert l.gpde_hidden_fwd(None, None, 0, None, 4, None, None, None, 3, d3, None, None, None, 1, None, None, None, 0, None) == 0   # no edges
    assert l.gpde_nnconv_fwd_hidden(None, 4, None, None, 8, None, None, None, 3, d3, None, None, None, 1,
                                    None, None, 0, None) == -1
    # gpde_nnconv_fwd_mixed_keepz(x, n, edge_attr, node_attr, hidden, hidden_absmax, hidden_nodes, n_edges, rowptr, src, dst, perm, ...)
    assert l.gpde_nnconv_fwd_mixed_keepz(None, 4, None, None, None, None, 9, 8, None, None, None, None, 3, d3, None, None,
                                         None, 1, 1, None, None, None, 0, None) == -1
    # the backward in its `hidden` form: null arrays are refused before any device work
    assert l.gpde_nnconv_bwd(None, 4, None, None, None, 8, None, None, None, None, None, None, None, 3, d3, None, None, None, 1, None,
                             None, None, None, None, None, None, None, None, 0, None, 0, None) == -1
    assert l.gpde_hidden_bwd(None, None, 8, None, None, None, 3, d3, None, None, None, None, None, None, 0, None) == -1
    # ... and two attribute sources at once (hidden + node_attr) are an argument error, not a guess
    na = _lib.GpdeNodeAttr()
    one = (ctypes.c_void_p * 3)()
    buf = ctypes.create_string_buffer(64)
    assert l.gpde_nnconv_bwd(None, 0, None, ctypes.byref(na), buf, 0, buf, None, None, None, buf, None, None, 3, d3, one, one, None, 1, buf,
                             None, None, buf, None, one, one, None, None, 0, buf, 64, None) == -1
    assert b"one attribute source" in l.gpde_last_error()
    # the accumulate flag belongs to the `hidden` form
    assert l.gpde_nnconv_bwd(None, 0, buf, None, None, 0, buf, None, None, buf, buf, None, None, 3, d3, one, one, None, 1, buf,
                             None, None, None, None, one, one, None, None, _lib.GPDE_BWD_ACCUMULATE_GRAD_HIDDEN, buf, 64, None) == -1
    assert b"ACCUMULATE" in l.gpde_last_error()
    hdr = open(os.path.join(REPO, "include", "gpde.h")).read()
    assert int(re.search(r"GPDE_BWD_ACCUMULATE_GRAD_HIDDEN = (\d+)", hdr).group(1)) == _lib.GPDE_BWD_ACCUMULATE_GRAD_HIDDEN
    # (hidden + an attribute source WITHOUT grad_hidden is the full backward with the forward's kept activations - legal)
    assert l.gpde_hidden_workspace_bytes(1000, 3, d3) > 0 and l.gpde_hidden_workspace_bytes(-1, 3, d3) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgpde.so")
    with pytest.raises(_lib.GpdeError):
        _lib.lib()


def test_forward_flags_of_the_binding_match_the_header():
    """The A/B flags the host layer ORs into `flags` (ops._PRECISION) are the enum values of include/gpde.h."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gpde.h")).read()
    vals = {m.group(1): int(m.group(2)) for m in re.finditer(r"(GPDE_FWD_\w+)\s*=\s*(\d+)", hdr)}
    for name in ("GPDE_FWD_DEFAULT", "GPDE_FWD_F16SPLIT", "GPDE_FWD_F16SPLIT_8WAVE", "GPDE_FWD_STATIC_RANGES",
                 "GPDE_FWD_AGG_F16", "GPDE_FWD_AGG_F32", "GPDE_FWD_NO_EDGE_PATH"):
        assert vals[name] == getattr(_lib, name), name
    from graph_pde_amd import ops
    assert ops._PRECISION["f32"] == 0 and ops._PRECISION["f16split"] == vals["GPDE_FWD_F16SPLIT"]
    assert ops._PRECISION["f16split_noedge"] == vals["GPDE_FWD_F16SPLIT"] | vals["GPDE_FWD_NO_EDGE_PATH"]
    assert len(set(ops._PRECISION.values())) == len(ops._PRECISION)


def test_weconv_descriptor_layout_matches_the_header():
    """ctypes mirror of `GpdeWeConvDesc` (include/gpde.h): field order, 8 pointers + 4 int32 = 80 bytes; the group entry
    point validates its descriptors on the host (no GPU needed)."""
    src = open(os.path.join(REPO, "include", "gpde.h")).read()
    body = re.search(r"typedef struct GpdeWeConvDesc \{(.*?)\} GpdeWeConvDesc;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(\w+);", body)
    assert names == [f[0] for f in _lib.GpdeWeConvDesc._fields_]
    assert ctypes.sizeof(_lib.GpdeWeConvDesc) == 80
    assert int(re.search(r"#define GPDE_WECONV_MAX_GROUP (\d+)", src).group(1)) == _lib.GPDE_WECONV_MAX_GROUP
    assert re.search(r"GPDE_AGGR_MAX = (\d+)", src).group(1) == str(_lib.GPDE_AGGR_MAX)
    l = _lib.lib()
    assert l.gpde_nnconv_fwd_edgeweights_group(None, 0, None) == 0
    d = (_lib.GpdeWeConvDesc * 1)()
    d[0].n_nodes, d[0].aggr = 4, 7
    assert l.gpde_nnconv_fwd_edgeweights_group(d, 1, None) == -1 and b"descriptor 0" in l.gpde_last_error()
    assert l.gpde_edge_weights_fwd(None, -1, 3, _lib.dims_array([6, 8, 8, 4096]), None, None, None, None, None, 0, None) == -1
    assert l.gpde_edge_weights_workspace_bytes(1000, 3, _lib.dims_array([6, 256, 256, 4096])) >= 8000


def test_node_attr_descriptor_layout_matches_the_header():
    """ctypes mirror of `GpdeNodeAttr` (include/gpde.h): pointer + 2 int32 + 8 int32 = 48 bytes; ops.NodeAttr.c_struct fills it
    as the header says; the entry points taking `node_attr` validate it on the host."""
    import torch
    from graph_pde_amd import ops
    src = open(os.path.join(REPO, "include", "gpde.h")).read()
    body = re.search(r"typedef struct GpdeNodeAttr \{(.*?)\} GpdeNodeAttr;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"(\w+)(?:\[\d+\])?;", body)
    assert names == [f[0] for f in _lib.GpdeNodeAttr._fields_]
    assert ctypes.sizeof(_lib.GpdeNodeAttr) == 48
    na = ops.NodeAttr.darcy(torch.zeros(10, 2), torch.zeros(10))
    c = na.c_struct()
    assert (c.stride, c.n_slots) == (3, 6) and list(c.sel)[:6] == [0, 1, 256, 257, 2, 258] and c.table == na.table.data_ptr()
    l = _lib.lib()
    dims = _lib.dims_array([6, 256, 256, 4096])
    c.n_slots = 5                                                # disagrees with dims[0]: refused on the host by every entry point
    buf = ctypes.create_string_buffer(64)
    assert l.gpde_hidden_fwd(None, ctypes.byref(c), 5, buf, 5, None, buf, buf, 3, dims, buf, None, None, 1, buf, None, None, 0, None) == -1
    assert b"GpdeNodeAttr" in l.gpde_last_error()
    arr = (ctypes.c_void_p * 3)()
    assert l.gpde_nnconv_bwd(None, 0, None, ctypes.byref(c), None, 0, buf, None, None, None, buf, None, None, 3, dims, arr, arr, None, 1,
                             buf, None, None, None, None, arr, arr, None, None, 0, buf, 64, None) == -1
    assert b"GpdeNodeAttr" in l.gpde_last_error()
    assert l.gpde_nnconv_bwd_deferred_supported(3, dims) == 1 and l.gpde_nnconv_bwd_deferred_supported(3, _lib.dims_array([6, 64, 128, 4096])) == 0
    assert l.gpde_nnconv_bwd_deferred_workspace_bytes(100, 5000, 3, dims, 6) > l.gpde_nnconv_bwd_workspace_bytes(100, 5000, 3, dims)
    # the one-chunk size: never below the default, and for a graph far beyond the default's chunk far above it
    assert l.gpde_nnconv_bwd_workspace_bytes_one_chunk(100, 5000, 3, dims) >= l.gpde_nnconv_bwd_workspace_bytes(100, 5000, 3, dims)
    d1k = _lib.dims_array([6, 1024, 1024, 4096])
    assert l.gpde_nnconv_bwd_workspace_bytes_one_chunk(14641, 5931137, 3, d1k) > 4 * l.gpde_nnconv_bwd_workspace_bytes(14641, 5931137, 3, d1k)
    assert l.gpde_gather_rows(None, 6, None, -1, None, None) == -1 and l.gpde_gather_rows(None, 6, None, 0, None, None) == 0


def test_cell_list_queries_validate_on_the_host():
    l = _lib.lib()
    D = ctypes.c_double * 2
    lo, hi = D(0.0, 0.0), D(1.0, 1.0)
    assert l.gpde_radius_csr_workspace_bytes(58081, 2, 0.1, lo, hi) > 4 * 4 * 58081
    assert l.gpde_radius_csr_workspace_bytes(10, 4, 0.1, lo, hi) == 0                     # dim 1..3
    # an absurdly fine grid is coarsened instead of allocating 1e12 cells
    assert l.gpde_radius_csr_workspace_bytes(10, 2, 1e-9, lo, hi) < (1 << 28)
    assert l.gpde_radius_csr_count(None, 5, None, 5, 2, 0.1, 0, lo, hi, None, None, 0, None) == -1
    assert l.gpde_radius_csr_fill(None, 5, None, 5, 2, 0.1, 7, lo, hi, None, None, None, 0, None, 0, None) == -1
