"""hipGraph replay for launch-bound shapes.

One layer-forward is 7 kernel launches (x depth for EGNN_Network), each issued from Python through ctypes plus a few
torch allocations: ~0.1 ms of host time per layer.  At the BASELINE batch sizes the GPU work hides that completely; for
small inputs (one protein, a few hundred nodes) it dominates.  `graphed(module, *example_inputs)` captures the launch
sequence once into a HIP graph (torch.cuda.CUDAGraph; the kernels run on the capture stream because every launch takes
`torch.cuda.current_stream()`), and returns a callable that copies new inputs into the captured buffers and replays.

Restrictions (same as any stream capture): fixed shapes / dtypes / which optional inputs are present; no host
synchronisation inside forward, i.e. not `only_sparse_neighbors=True` (its K is read back from the device,
egnn_pytorch.py:249).  The returned outputs are the graph's own buffers: they are overwritten by the next replay --
clone them if they have to survive it."""
from __future__ import annotations

import torch

from . import _ops


def graphed(module, *example_inputs, warmup: int = 2, range_check: str | None = None, **example_kwargs):
    """Capture `module(*example_inputs, **example_kwargs)` into a HIP graph; returns `run(*inputs, **kwargs)`.
    range_check: how the range status word is examined after a replay -- None = the process setting (EGNN_RANGE_CHECK),
    "deferred" = copied to pinned memory behind the replay and looked at by the next call (keeps the replay asynchronous),
    "sync" = one blocking 4-byte read per replay, "off"."""
    # training-mode dropout draws its mask seed on the host (egnn_pytorch_amd/_dropout.py) and the seed is a kernel ARGUMENT: a captured
    # graph would replay the same mask every step, silently (torch's own graph-safe RNG advances a device-side Philox offset instead)
    for m in module.modules():
        if callable(getattr(m, "dropout_active", None)) and m.dropout_active():
            raise NotImplementedError("graphed(): a layer has training-mode dropout active; its mask seed would be frozen into the graph "
                                      "(call .eval(), or run the module eagerly)")
    args = [a.clone() if torch.is_tensor(a) else a for a in example_inputs]
    kwargs = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_kwargs.items()}
    dev = next(a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a)).device
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):                       # warm-up: weight re-layout, allocator pools, lazy module loads
        for _ in range(max(1, warmup)):
            module(*args, **kwargs)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outputs = module(*args, **kwargs)

    def run(*inputs, **kw):
        if len(inputs) != len(args) or set(kw) != set(kwargs):
            raise ValueError("graphed(): call with the same arguments (positional / keyword) it was captured with")
        for dst, src in list(zip(args, inputs)) + [(kwargs[k], kw[k]) for k in kwargs]:
            if torch.is_tensor(dst):
                if not torch.is_tensor(src) or src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("graphed(): input shape / dtype differs from the captured one")
                dst.copy_(src)
            elif dst is not src and dst != src:
                raise ValueError("graphed(): non-tensor arguments are baked into the graph and cannot change")
        graph.replay()
        _ops.range_check_after_forward(dev, mode=range_check)   # the captured forward could not read the status word itself
        return outputs

    run.graph = graph
    return run
