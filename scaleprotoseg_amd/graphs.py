"""HIP-graph capture of a training / inference step built from this package's operators.

The operators are capture-safe by construction: no host synchronisation, no host read-back, every kernel on torch's
current stream, scratch buffers from torch's caching allocator (which serves a capture from a private pool).  What is
NOT safe is a property of autograd + the HIP runtime, diagnosed in round 2 (``tools/probes/capture_repro.py``,
``profiles/r2_capture_repro.log``; it reproduces with stock torch ops alone):

    an autograd graph created EAGERLY ON THE LEGACY DEFAULT STREAM that is still alive when the capture runs keeps the
    leaves' ``AccumulateGrad`` nodes alive, and those nodes remember the default stream (torch/csrc/autograd/function.h,
    "Function Streams").  The captured backward then runs them on the default stream behind an event of the capturing
    stream; that pulls the legacy stream into the capture and ``hipStreamEndCapture`` of ROCm 7 dereferences a null
    stream (SIGSEGV inside libamdhip64.so under ``CUDAGraph::capture_end``; CUDA returns
    ``cudaErrorStreamCaptureImplicit`` for the same misuse).

``capture_step`` therefore warms the step up and captures it on ONE side stream, and refuses to start the capture while
such a stale graph is alive (torch reports the stream mismatch as a warning during the warm-up; it is turned into an
error here, before any capture has begun).
"""
from __future__ import annotations

import warnings
from typing import Any, Callable, Optional, Tuple

import torch

from ._lib import SpxError

_STALE = "AccumulateGrad node's stream does not match"


class StepGraph:
    """A captured step.  ``replay()`` re-runs it and then drops the package's pack cache: a captured step that contains the
    optimizer edits the parameters IN PLACE without touching their ``_version`` or storage - the cache's key - so a pack
    built by an eager forward before the replay (evaluation, the prototype push) would otherwise be served again after it.
    (The captured step itself never uses the cache: its pack kernels are part of the graph.)  ``graph`` is the underlying
    ``torch.cuda.CUDAGraph``; whoever replays THAT directly must call ``functional.invalidate_pack_cache()`` themselves."""

    def __init__(self, graph: "torch.cuda.CUDAGraph"):
        self.graph = graph

    def replay(self) -> None:
        from .functional import invalidate_pack_cache

        self.graph.replay()
        invalidate_pack_cache()

    def __getattr__(self, name):
        return getattr(self.graph, name)


def capture_step(step: Callable[[], Any], warmup: int = 3, stream: Optional[torch.cuda.Stream] = None,
                 ) -> Tuple[StepGraph, Any]:
    """Warm ``step`` up ``warmup`` times and capture one call of it into a HIP graph.

    ``step`` is a closure over static input tensors (refill them in place between replays) that runs forward and - if
    it trains - backward.  Returns ``(graph, outputs)``: ``graph.replay()`` re-runs the step, ``outputs`` (whatever
    ``step`` returned during the capture, plus any ``.grad`` it produced) are overwritten in place by every replay.
    Raises ``SpxError`` instead of capturing when an autograd graph from another stream is still alive (see the module
    docstring)."""
    if not torch.cuda.is_available():
        raise SpxError("capture_step needs an AMD GPU")
    s = stream or torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    # torch reports the stream mismatch through TORCH_WARN_ONCE: once per process unless "warn always" is on, so a second
    # refused capture (or any earlier eager side-stream step) would otherwise go unseen and the capture would crash.  Both
    # global switches are restored afterwards, and warnings that are not ours are re-emitted to the caller.
    prev_always = torch.is_warn_always_enabled()
    torch.set_warn_always(True)
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(s):
                for _ in range(max(1, int(warmup))):
                    step()
    finally:
        torch.set_warn_always(prev_always)
    torch.cuda.current_stream().wait_stream(s)
    stale = [w for w in caught if _STALE in str(w.message)]
    for w in caught:
        if w not in stale:
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
    if stale:
        raise SpxError(
            "capture_step: an autograd graph created on another stream (an eager step's outputs or loss) is still alive; "
            "its AccumulateGrad nodes would pull the legacy default stream into the capture and crash "
            "hipStreamEndCapture.  Delete those tensors (or run the eager steps under the same side stream) first."
        )
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        outputs = step()
    from .functional import invalidate_pack_cache

    invalidate_pack_cache()          # (the warm-up ran eagerly: whatever it cached predates the graph's own updates)
    return StepGraph(graph), outputs
