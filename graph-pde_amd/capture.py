"""Record a model function's native calls once into a HIP graph and replay it (opt-in).

The MGKN V-cycles issue 52 / 65 NNConv calls per forward on graphs of a few hundred to a few ten-thousand edges
(/root/reference/multipole-graph-neural-operator/MGKN_orthogonal_burgers1d.py:65-82, MGKN_general_darcy2d.py:76-90): the
GPU work of such a forward is ~1-2 ms while the host needs longer than that to ISSUE the calls (Python dispatch, cache
look-ups, ctypes, kernel launches).  Shapes, graphs and pointers are the same for every call of one sample, so the whole
sequence can be recorded once (`hipStreamBeginCapture` through `torch.cuda.CUDAGraph`) and replayed with one launch:

    fwd = gp.capture(lambda x: model(x, edge_index, edge_attr), x_example)      # warm-up calls, then ONE recorded call
    y = fwd(x)                                                                   # copies x into the static input, replays

What is recorded is exactly what the uncaptured call launches - the same kernels of libgpde.so with the same arguments - so
the result is bit-identical to calling the function directly.  Everything the host decides (CSR look-ups, weight packing,
the cache policies of hidden_cache.py) is decided during the warm-up / recording call and frozen:
  * tensors passed as arguments are copied into static buffers at every call; everything else the function reads (weights,
    edge_index, edge_attr) is read from where it lay at recording time, and what the library DERIVED from them during the
    warm-up - packed weight images, hidden activations H, per-edge weights W_e, CSRs - is frozen with the recording: a replay
    does not re-pack or rebuild.  An in-place update of such a tensor outside the graph (`p.mul_()`, an `optimizer.step()`
    between replays) therefore makes the recording STALE.  The recording notes every tensor a cache key was built from with
    its version counter; a replay that finds one moved records again first (warm-up + one recorded call: `recordings` counts
    them), so the result is that of a direct call with the new values.  Writes that do not move the counter (`p.data.add_()`)
    are invisible to it, as they are to the caches themselves (ops.clear_caches); NEW tensors need a new capture;
  * the device buffers the recorded kernels read but the library's caches own (CSR arrays, slot-ordered attribute copies,
    packed weights, cached H / W_e) are pinned on the `Captured` object: cache eviction, `hidden_cache.clear()` (every later
    `gp.capture` calls it), `release_all` under memory pressure or `ops.clear_caches()` cannot free them under the graph;
  * the outputs live in the graph's memory pool and are overwritten by the next replay (`copy_outputs=True` returns clones);
  * a whole training step can be recorded too (`zero_grad(set_to_none=True)`, forward, `loss.backward()`, `optimizer.step()`
    inside `fn`): the optimizer must be capturable (`torch.optim.Adam(..., capturable=True)`: its step count lives on the
    device) and `capture(..., updates_parameters=True)` must be said, because the recorded step rewrites the weights behind
    Python's back (tests/test_gpu_capture.py).
    No autograd graph of an EARLIER direct step may still be alive when a training step is recorded (a `loss` tensor kept in a
    variable is enough): its AccumulateGrad nodes stay bound to the stream that step ran on, and the recorded gradient
    accumulation would fork onto that stream - `hipStreamEndCapture` aborts the process on such an unjoined fork.
    Known limit (round 6, not understood): a SECOND model training eagerly on the same device between a recorded training step and
    its replays changed what the next replay computed (tests/test_gpu_capture.py; a forward-only neighbour did not, and a recording
    replayed on its own follows the direct steps exactly).  Record and replay a training step without other training in between.
Not a tracing compiler: nothing is transformed, fused or re-ordered.
"""
from __future__ import annotations

from typing import Any, Callable

import torch


def _map_tensors(obj: Any, f: Callable[[torch.Tensor], Any]) -> Any:
    if isinstance(obj, torch.Tensor):
        return f(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(o, f) for o in obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, f) for k, v in obj.items()}
    return obj


class Captured:
    """A recorded call of `fn(*args)`; calling it replays the HIP graph (see the module docstring)."""

    def __init__(self, fn: Callable, args: tuple, warmup: int = 3, copy_outputs: bool = False, updates_parameters: bool = False):
        devs = {a.device for a in args if isinstance(a, torch.Tensor)}
        if any(d.type != "cuda" for d in devs):
            raise ValueError("capture() records a HIP graph: tensor arguments must live on the GPU")
        self._static_in = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
        self._copy_outputs = copy_outputs
        self._updates_parameters = updates_parameters
        self._fn, self._warmup = fn, warmup
        self._dev = next(iter(devs)) if devs else torch.device("cuda", torch.cuda.current_device())
        self.replays = 0
        self.recordings = 0
        self._retired = []           # earlier recordings (graph, outputs, pins): kept - a caller may still hold their outputs
        self._record()

    def _record(self):
        fn, warmup, dev = self._fn, self._warmup, self._dev
        # The caches keep tensors WITH their autograd history (a shared H node, W_e): as long as such a graph lives, the
        # parameters' AccumulateGrad nodes live - bound to the stream of the step that created them, usually the default stream of
        # earlier direct calls.  Recorded that way the gradient accumulation is an unjoined fork onto the legacy stream
        # (hipStreamEndCapture crashed on it).  Drop them: the warm-up below rebuilds everything on the recording stream.
        from . import hidden_cache, ops
        hidden_cache.clear()
        import gc
        gc.collect()
        # warm-up on a side stream (torch's capture protocol): builds the CSRs, packs the weights, lets the cache policies see
        # the module repeat - host-side state that the recording call must find settled
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn(*self._static_in)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # "relaxed": the host side of a native call may query free memory (workspace / cache budgets) - not a stream operation
        # (recorded on the warm-up's stream: autograd's AccumulateGrad nodes - created by the warm-up steps, one per parameter,
        # alive across iterations - run on the stream they were created on; on another stream the gradient accumulation of a
        # recorded training step is an unjoined fork of the capture)
        outer, ops._capture_watch = ops._capture_watch, []
        try:
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="relaxed"):
                self._static_out = fn(*self._static_in)
            watched = ops._capture_watch
        finally:
            ops._capture_watch = outer
        torch.cuda.synchronize(dev)
        # the tensors the recorded call's cache keys were built from, with the version each had (one entry per tensor: the
        # version it had when LAST keyed - a recorded training step moves its weights' counters itself)
        seen = {}
        for t, v in watched:
            seen[id(t)] = (t, v)
        self._watched = list(seen.values())
        # what the recorded kernels read from cache-owned memory: pinned for the life of this recording
        self._pins = (ops.cache_snapshot(), hidden_cache.snapshot())
        self.recordings += 1

    def stale(self) -> bool:
        """A tensor one of the recording's cache keys was built from has been modified in place since (version counter)."""
        if self._updates_parameters:
            return False            # the graph itself rewrites the weights and re-derives everything from them every replay
        return any(t._version != v for t, v in self._watched)

    def __call__(self, *args):
        if len(args) != len(self._static_in):
            raise TypeError(f"captured with {len(self._static_in)} arguments, called with {len(args)}")
        for s, a in zip(self._static_in, args):
            if isinstance(s, torch.Tensor):
                if not isinstance(a, torch.Tensor) or a.shape != s.shape or a.dtype != s.dtype:
                    raise ValueError("a captured call replays fixed shapes: argument of other shape / dtype - capture again")
                if a.data_ptr() != s.data_ptr():
                    s.copy_(a)
            elif a != s:
                raise ValueError("non-tensor arguments are frozen at capture time")
        if self.stale():
            # weights / attributes / indices changed in place outside the graph: the packed images, H, W_e the recording
            # reads belong to the old values.  Record again (the old graph and its pool stay alive: its outputs may be held)
            self._retired.append((self.graph, self._static_out, self._pins))
            self._record()
        self.graph.replay()
        self.replays += 1
        if self._updates_parameters:
            # the recorded optimizer step rewrote the weights without Python seeing it: the host-side caches key on tensor
            # version counters, which a replay does not move - drop them, or a later DIRECT call would reuse the packed image
            # / hidden activations of the weights as they were at recording time
            from . import hidden_cache, ops
            ops.clear_param_caches()
            hidden_cache.clear()
        return _map_tensors(self._static_out, lambda t: t.clone()) if self._copy_outputs else self._static_out


def capture(fn: Callable, *example_args, warmup: int = 3, copy_outputs: bool = False, updates_parameters: bool = False) -> Captured:
    """Record `fn(*example_args)` into a HIP graph after `warmup` ordinary calls; returns the replaying callable.
    `updates_parameters`: `fn` contains an optimizer step (the weights change inside the graph): every replay then drops the
    host-side weight / activation caches (see Captured.__call__)."""
    return Captured(fn, example_args, warmup=warmup, copy_outputs=copy_outputs, updates_parameters=updates_parameters)
