"""Opt-in HIP-graph replay of a module call (config key `graph: True`).

At the deltaMG minibatch shape (100 basins x 16 members, 365 + 365 days) the ~20 launches of one step take the
GPU 0.39 ms and the host 0.47-0.62 ms to enqueue (profiles/r03_host_overhead.txt): the step is host-bound.  A
module built with `graph=True` captures, per input shape, the launch sequence of its forward
(hbv.py:303-361: warm-up pass, main pass, routing, BFI) and of its backward into two HIP graphs
(`torch.cuda.CUDAGraph`) and replays them; the host then pays two graph launches per step.

How it fits autograd: the captured forward ran on STATIC input copies (`x`, `parameters`) and left an ordinary
autograd graph from them to static outputs.  `_Replay` is the node the caller sees: its forward copies the
caller's inputs into the static buffers and replays the forward graph; its backward copies the incoming
gradients into static buffers and replays a backward graph that was captured -- on the first backward with that
pattern of present / absent gradients -- from `torch.autograd.grad` over the retained static autograd graph.
The kernels, their order and their arithmetic are exactly the eager path's: results are bit-identical
(tests/test_graphed.py).

Restrictions (each raises): dy_drop > 0 (the masks are drawn on the host per call), `muwts`, `cache_states`,
`check_finite`, `initialize`.  The CPU generator is still advanced per call as the eager path does (hbv.py:240), so
a script's random stream does not depend on the switch.  Outputs are views of static buffers: they are
overwritten by the module's next call with the same shape (the contract of torch.cuda.make_graphed_callables).
"""
from __future__ import annotations

from typing import Optional

import torch


class _Captured:
    """The graphs and static buffers of one (module settings, input shape, grad mode)."""

    def __init__(self, module, x: torch.Tensor, parameters: torch.Tensor, want_grad: bool):
        dev = x.device
        self.want_grad = want_grad
        self.x = x.detach().clone()
        self.p = parameters.detach().clone().requires_grad_(want_grad)
        self.pool = torch.cuda.graph_pool_handle()
        self.bwd = {}
        self.src = [None, None]              # weak references to the caller's last (x, parameters) and their versions
        rng = torch.get_rng_state()          # the warm-up and capture passes draw; the call itself draws once (below)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):        # warm-up off the capture: lazy initialisation (LDS attributes, plans)
            for _ in range(2):
                out = self._run(module)
                if want_grad:
                    torch.autograd.grad(out["streamflow"].sum(), self.p)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=self.pool):
            self.out = self._run(module)
        self.keys = list(self.out.keys())
        self.states = module._states_cache
        torch.set_rng_state(rng)
        module._advance_rng(x.shape[1])

    def load(self, x, parameters):
        """Caller's inputs -> static buffers.  A tensor that is the very object seen last time, with the same version
        counter, holds the bytes that are already there: no copy (a fixed forcing block; a leaf parameter tensor)."""
        import weakref
        for slot, (src, dst) in enumerate(((x, self.x), (parameters, self.p))):
            seen = self.src[slot]
            if seen is not None and seen[0]() is src and seen[1] == src._version:
                continue
            # through .data: the static tensors are saved in the captured autograd graph, whose version check must not
            # see the refresh (their CONTENT is what the replayed kernels read; the graph's nodes hold pointers)
            dst.data.copy_(src.detach())
            self.src[slot] = (weakref.ref(src), src._version)

    def _run(self, module):
        with torch.set_grad_enabled(self.want_grad):
            return module._forward_eager({"x_phy": self.x}, self.p)

    def backward_graph(self, pattern, grads):
        """(graph, static grad_outputs, static grad of parameters) for this pattern of present gradients."""
        hit = self.bwd.get(pattern)
        if hit is not None:
            return hit
        outs = [self.out[k] for k, has in zip(self.keys, pattern) if has]
        gos = [torch.zeros_like(g) for g in grads if g is not None]
        dev = self.p.device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            torch.autograd.grad(outs, [self.p], gos, retain_graph=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            (gp,) = torch.autograd.grad(outs, [self.p], gos, retain_graph=True)
        self.bwd[pattern] = (g, gos, gp)
        return self.bwd[pattern]


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap: _Captured, x, parameters):
        cap.load(x, parameters)
        cap.fwd.replay()
        ctx.cap = cap
        ctx.set_materialize_grads(False)
        outs = tuple(cap.out[k].detach() for k in cap.keys)
        ctx.mark_non_differentiable(*[o for o, k in zip(outs, cap.keys) if not cap.out[k].requires_grad])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        cap: _Captured = ctx.cap
        grads = tuple(g if cap.out[k].requires_grad else None for g, k in zip(grads, cap.keys))
        pattern = tuple(g is not None for g in grads)
        if not any(pattern):
            return None, None, torch.zeros_like(cap.p)
        graph, gos, gp = cap.backward_graph(pattern, grads)
        for s, g in zip(gos, (g for g in grads if g is not None)):
            s.data.copy_(g)
        graph.replay()
        return None, None, gp.detach()


def graphed_forward(module, x_dict: dict, parameters: torch.Tensor) -> dict:
    """`module.forward` through captured HIP graphs (see the module docstring)."""
    x = x_dict["x_phy"]
    if module.dy_drop > 0:
        raise ValueError("graph=True: dy_drop > 0 draws its masks on the host per call; not capturable")
    if x_dict.get("muwts", None) is not None or module.cache_states or module.check_finite or module.initialize:
        raise ValueError("graph=True does not support muwts, cache_states, check_finite or initialize")
    if x.requires_grad:
        raise ValueError("graph=True does not differentiate the forcings")
    if not parameters.is_contiguous():
        parameters = parameters.contiguous()
    want_grad = torch.is_grad_enabled() and parameters.requires_grad
    key = (tuple(x.shape), tuple(x.stride()), tuple(parameters.shape), want_grad, module._settings_key())
    caps = module.__dict__.setdefault("_graph_cache", {})
    cap: Optional[_Captured] = caps.get(key)
    if cap is None:
        if len(caps) > 4:
            caps.clear()
        cap = caps[key] = _Captured(module, x, parameters, want_grad)
    else:
        module._advance_rng(x.shape[1])      # the draw the eager path makes per call (hbv.py:240)
    if want_grad:
        outs = _Replay.apply(cap, x, parameters)
    else:
        cap.load(x, parameters)
        cap.fwd.replay()
        outs = tuple(cap.out[k] for k in cap.keys)
    module.__dict__["_states_cache"] = cap.states
    return dict(zip(cap.keys, outs))
