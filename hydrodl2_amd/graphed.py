"""Opt-in HIP-graph replay of a module call (config key `graph: True`).

At the deltaMG minibatch shape (100 basins x 16 members, 365 + 365 days) the ~20 launches of one step take the
GPU 0.39 ms and the host 0.47-0.62 ms to enqueue (profiles/r03_host_overhead.txt): the step is host-bound.  A
module built with `graph=True` captures, per input shape, the launch sequence of its forward
(hbv.py:303-361: warm-up pass, main pass, routing, BFI) and of its backward into two HIP graphs
(`torch.cuda.CUDAGraph`) and replays them; the host then pays two graph launches per step.

Every class of the family takes it: `Hbv` / `Hbv_1_1p` (one raw parameter tensor), `Hbv_2` (the parameter tuple,
`ac_all` / `elev_all`; hbv_2.py:324-390), `Hbv_2_hourly` (three parameter tensors, the gage topology) and `HbvAdj`.
What a class provides: `_forward_eager(x_dict, parameters)` (its ordinary forward), `_settings_key()`,
`_advance_rng(ngrid)` (the host draws one eager call makes), and optionally `_graph_state_attrs` (attributes the
call leaves on the module: the state caches) and `_graph_structural` (x_dict entries whose CONTENT shapes the launch
sequence -- the hourly model's `outlet_topo` / `areas`: a new object or version means a new capture).

How it fits autograd: the captured forward ran on STATIC copies of the inputs and left an ordinary autograd graph
from the static parameter tensors to static outputs.  `_Replay` is the node the caller sees: its forward copies the
caller's inputs into the static buffers and replays the forward graph; its backward copies the incoming gradients
into static buffers and replays a backward graph that was captured -- on the first backward with that pattern of
present / absent gradients -- from `torch.autograd.grad` over the retained static autograd graph.  The kernels, their
order and their arithmetic are exactly the eager path's: results are bit-identical (tests/test_graphed.py).

Restrictions (each raises): dy_drop > 0 (the masks are drawn on the host per call), `muwts`, `cache_states`,
`check_finite`, `initialize`, a forcing tensor that requires grad.  The CPU generator is still advanced per call as
the eager path does (hbv.py:240), so a script's random stream does not depend on the switch.  Outputs are views of
static buffers: they are overwritten by the module's next call with the same shape (the contract of
torch.cuda.make_graphed_callables).  The gradients handed to autograd are COPIES of the static gradient buffers: a
leaf's `.grad` may be accumulated into across steps.  That copy is why the switch is for SMALL steps only:
at the headline shape (a 3.8 GB gradient) the graphed step takes 4.05 ms against 2.45 ms eager (round 5 measurement).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch


def _as_list(parameters):
    return list(parameters) if isinstance(parameters, (tuple, list)) else [parameters]


class _Captured:
    """The graphs and static buffers of one (module settings, input shapes, grad pattern)."""

    def __init__(self, module, x_dict: dict, parameters, want: tuple):
        self.is_tuple = isinstance(parameters, (tuple, list))
        plist = _as_list(parameters)
        dev = x_dict["x_phy"].device
        self.want = want                          # per parameter tensor: does the caller differentiate it
        self.xkeys = [k for k, v in x_dict.items() if torch.is_tensor(v)]
        self.xs = {k: (x_dict[k].detach().clone() if k in self.xkeys else x_dict[k]) for k in x_dict}
        self.ps = [p.detach().clone().requires_grad_(w) for p, w in zip(plist, want)]
        self.pool = torch.cuda.graph_pool_handle()
        self.bwd = {}
        self.src = {}                            # slot -> (weak reference to the caller's tensor, its version)
        rng = torch.get_rng_state()              # the warm-up and capture passes draw; the call itself draws once (below)
        diff = [p for p, w in zip(self.ps, want) if w]
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):            # warm-up off the capture: lazy initialisation (LDS attributes, plans)
            for _ in range(2):
                out = self._run(module)
                if diff:
                    first = next(v for v in out.values() if v.requires_grad)
                    torch.autograd.grad(first.sum(), diff, allow_unused=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd, pool=self.pool):
            self.out = self._run(module)
        self.keys = list(self.out.keys())
        self.state_attrs = {a: getattr(module, a) for a in getattr(module, "_graph_state_attrs", ("_states_cache",))
                            if hasattr(module, a)}
        torch.set_rng_state(rng)
        module._advance_rng(x_dict["x_phy"].shape[1])

    def _packed(self):
        return tuple(self.ps) if self.is_tuple else self.ps[0]

    def load(self, x_dict: dict, plist: list):
        """Caller's inputs -> static buffers.  A tensor that is the very object seen last time, with the same version
        counter, holds the bytes that are already there: no copy (a fixed forcing block; a leaf parameter tensor)."""
        pairs = [(("x", k), x_dict[k], self.xs[k]) for k in self.xkeys]
        pairs += [(("p", i), p, s) for i, (p, s) in enumerate(zip(plist, self.ps))]
        for slot, src, dst in pairs:
            seen = self.src.get(slot)
            version = src._version if not src.is_inference() else None
            if seen is not None and seen[0]() is src and version is not None and seen[1] == version:
                continue
            # through .data: the static tensors are saved in the captured autograd graph, whose version check must not
            # see the refresh (their CONTENT is what the replayed kernels read; the graph's nodes hold pointers)
            dst.data.copy_(src.detach())
            self.src[slot] = (weakref.ref(src), version)

    def _run(self, module):
        with torch.set_grad_enabled(any(self.want)):
            return module._forward_eager(dict(self.xs), self._packed())

    def backward_graph(self, pattern, grads):
        """(graph, static grad_outputs, static gradients of the differentiated parameters) for this pattern of
        present gradients."""
        hit = self.bwd.get(pattern)
        if hit is not None:
            return hit
        outs = [self.out[k] for k, has in zip(self.keys, pattern) if has]
        gos = [torch.zeros_like(g) for g in grads if g is not None]
        diff = [p for p, w in zip(self.ps, self.want) if w]
        dev = diff[0].device
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            torch.autograd.grad(outs, diff, gos, retain_graph=True, allow_unused=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            gps = torch.autograd.grad(outs, diff, gos, retain_graph=True, allow_unused=True)
        self.bwd[pattern] = (g, gos, gps)
        return self.bwd[pattern]


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap: _Captured, x_dict: dict, *plist):
        cap.load(x_dict, list(plist))
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
            return (None, None) + tuple(torch.zeros_like(p) if w else None for p, w in zip(cap.ps, cap.want))
        graph, gos, gps = cap.backward_graph(pattern, grads)
        for s, g in zip(gos, (g for g in grads if g is not None)):
            s.data.copy_(g)
        graph.replay()
        # clones, not the static buffers: AccumulateGrad may adopt what it is handed as a leaf's .grad, and the next
        # replay would then overwrite an accumulated gradient in place
        it = iter(gps)
        res = []
        for p, w in zip(cap.ps, cap.want):
            if not w:
                res.append(None)
                continue
            gp = next(it)
            res.append(gp.detach().clone() if gp is not None else torch.zeros_like(p))
        return (None, None) + tuple(res)


def graphed_forward(module, x_dict: dict, parameters) -> dict:
    """`module.forward` through captured HIP graphs (see the module docstring)."""
    x = x_dict["x_phy"]
    if getattr(module, "dy_drop", 0.0) > 0:
        raise ValueError("graph=True: dy_drop > 0 draws its masks on the host per call; not capturable")
    if (x_dict.get("muwts", None) is not None or getattr(module, "cache_states", False)
            or getattr(module, "check_finite", False) or getattr(module, "initialize", False)):
        raise ValueError("graph=True does not support muwts, cache_states, check_finite or initialize")
    if x.requires_grad:
        raise ValueError("graph=True does not differentiate the forcings")
    is_tuple = isinstance(parameters, (tuple, list))
    plist = [p if p.is_contiguous() else p.contiguous() for p in _as_list(parameters)]
    grad_on = torch.is_grad_enabled()
    want = tuple(bool(grad_on and p.requires_grad) for p in plist)
    tens = [(k, v) for k, v in x_dict.items() if torch.is_tensor(v)]
    structural = tuple((k, id(x_dict[k]), x_dict[k]._version) for k in getattr(module, "_graph_structural", ())
                       if k in x_dict)
    key = (tuple((k, tuple(v.shape), tuple(v.stride()), v.dtype, str(v.device)) for k, v in tens),
           tuple((tuple(p.shape), p.dtype, str(p.device)) for p in plist), want, is_tuple, structural,
           module._settings_key())
    caps = module.__dict__.setdefault("_graph_cache", {})
    cap: Optional[_Captured] = caps.get(key)
    if cap is None:
        if len(caps) > 4:
            caps.clear()
        cap = caps[key] = _Captured(module, x_dict, tuple(plist) if is_tuple else plist[0], want)
    else:
        module._advance_rng(x.shape[1])      # the draw the eager path makes per call (hbv.py:240)
    if any(want):
        outs = _Replay.apply(cap, x_dict, *plist)
    else:
        cap.load(x_dict, plist)
        cap.fwd.replay()
        outs = tuple(cap.out[k] for k in cap.keys)
    for a, v in cap.state_attrs.items():
        module.__dict__[a] = v
    return dict(zip(cap.keys, outs))
