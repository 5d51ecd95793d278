"""hipGraph replay of the static part of a training step -- EXPERIMENTAL, not on any default path.

Status (end of round 1): `bench.py --graph on` runs (same 19.8 ms/step as the eager step on a box
whose host keeps up; the point is the 20 x lower host cost on slower hosts), but the capture is
not trustworthy yet on this ROCm stack: `hipStreamEndCapture` crashed when an ordinary eager
autograd step had run in the process before the capture, and one parity run produced a wrong
loss on its third replay.  Until both are understood there is no test that vouches for it and
nothing uses it by default.

Behind the set-abstraction stage every tensor of the 3DETR step has a fixed shape: the
transformer encoder / decoder, the prediction heads and the criterion are ~900 launch-sized
kernels whose enqueueing costs the host more (17-25 ms, depending on the host CPU) than the
GPU needs to run them.  ``GraphedTail`` captures ``loss = fn(*inputs)`` together with its
backward pass once and replays both as ONE hipGraph launch (0.4 ms of host time per 1000
kernels measured with tools/graph_probe.py, against 9.4 ms for the same kernels enqueued one
by one).  The set-abstraction stage itself stays eager: its de-duplicated groups have a
data-dependent row count (pointnet2/fused_sa_mlp.py), and its sampling runs ahead on a side
stream (pointnet2_utils.SamplingPrefetcher).

No reference counterpart (the reference enqueues every op eagerly, engine.py:144-159); the
captured kernels are exactly the ones the eager path launches.

Dropout: the kernels fold a device-resident seed word into their counter hash
(attention_core.use_device_seed); the graph bumps that word itself, so every replay draws
fresh masks.
"""
import torch

from . import attention_core


class GraphedTail:
    """fn(*inputs) -> scalar loss, forward + backward captured as one hipGraph.

    inputs:  example tensors (shapes / dtypes / devices are frozen); the ones with
             ``requires_grad`` get their gradient returned by ``replay``.
    params:  the parameters ``fn`` uses.  Their gradients live in the graph's memory pool and are
             OVERWRITTEN (not accumulated) by every replay; ``replay`` attaches them as ``p.grad``
             (replacing whatever was there), so ``optimizer.zero_grad(set_to_none=True)`` between
             steps is harmless and gradient accumulation over several replays is not supported.
    warmup:  eager forward+backward runs before the capture (library workspaces, lazily built
             caches; batch-norm running statistics move ``warmup`` extra times).
    """

    def __init__(self, fn, inputs, params, warmup=3):
        if not all(t.is_cuda for t in inputs):
            raise ValueError("GraphedTail needs CUDA tensors")
        self._fn = fn
        self._params = [p for p in params if p.requires_grad]
        self._static = [t.detach().clone().requires_grad_(t.requires_grad) for t in inputs]
        dev = self._static[0].device
        self._seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self._prev_seed = attention_core._DEVICE_SEED
        attention_core.use_device_seed(self._seed)

        # Gradients are taken with torch.autograd.grad (as torch.cuda.make_graphed_callables does), not
        # .backward(): no AccumulateGrad nodes take part, so the capture does not depend on the stream
        # those nodes were created on by earlier eager steps.  Warm-up and capture share one side stream.
        wanted = [t for t in self._static if t.requires_grad] + self._params

        def run():
            loss = fn(*self._static)
            return loss, torch.autograd.grad(loss, wanted, allow_unused=True)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._seed += 1
                run()
        torch.cuda.current_stream(dev).wait_stream(side)

        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=side):
            self._seed += 1
            self._loss, grads = run()
        self._loss = self._loss.detach()
        grads = list(grads)
        self._input_grads = [grads.pop(0) if t.requires_grad else None for t in self._static]
        self._param_grads = [(p, g) for p, g in zip(self._params, grads) if g is not None]

    def replay(self, *inputs):
        """-> (loss, [gradient of each input or None]); both are static tensors that the next
        replay overwrites."""
        with torch.no_grad():
            for dst, src in zip(self._static, inputs):
                dst.copy_(src)
        self._graph.replay()
        for p, g in self._param_grads:
            p.grad = g
        return self._loss, self._input_grads

    def close(self):
        """Back to host-side dropout seeds (eager calls after this draw independent masks)."""
        attention_core.use_device_seed(self._prev_seed)
