"""hipGraph capture of ONE iteration of a reference-style driver loop.

The reference's drivers run ``cost -> loss.backward() -> optimizer.step() -> optimizer.zero_grad()`` as eager PyTorch
(``odometery/two_frame_sfm.py:150-207``, ``odometery/odometery.py:375-407,756-915``).  Behind the fused cost kernel (29 us) that glue --
the autograd engine, ``torch.optim.Adam``'s Python, lietorch-style ``retr().matrix()`` and its backward, a dozen small pose-algebra
launches -- is 450-600 us of INTERPRETER time per iteration; the GPU work itself is ~25 small kernels.  A loop whose iterations all
issue the same launches on the same buffers can be recorded once and replayed: ``GraphedStep(fn, optimizers)`` warms ``fn`` up on a side
stream, captures one call of it into a ``torch.cuda.CUDAGraph`` (a hipGraph) and ``replay()`` re-issues it as ONE graph launch.

What makes an iteration capturable (the drop-in functions of this package satisfy it):
* no host synchronisation inside ``fn`` (``.item()``, ``bool(tensor)``, ``nonzero``, the reference's ``assert isfinite`` -- here opt-in);
* every launch on ``torch.cuda.current_stream()`` (the C ABI takes the stream; ``_lib.stream_ptr()`` hands over the capturing one);
* persistent state is updated IN PLACE (``T.copy_(T @ inv(Exp(d)))``, not ``T = T @ ...``: a rebound name points at capture-time memory);
* ``torch.optim.Adam(..., capturable=True)`` -- ``GraphedStep`` switches the given optimisers' parameter groups to capturable and moves an
  existing ``state['step']`` to the device (bit-identical arithmetic; only where the step counter lives changes).
"""
from __future__ import annotations

import torch


class GraphedStep:
    def __init__(self, fn, optimizers=(), warmup=2, device=None):
        """fn(): one whole iteration (forward, backward, optimiser step, zero_grad, in-place bookkeeping); returns a tensor or a
        tuple of tensors that stay valid across replays (e.g. the loss).  ``warmup`` REAL iterations run first (they count: the
        caller's loop should account for ``self.warmup_outputs``); then one more is captured WITHOUT being executed."""
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        for opt in optimizers:
            for grp in opt.param_groups:
                grp["capturable"] = True
                if grp.get("fused"):
                    grp["fused"] = False
            for st in opt.state.values():
                if "step" in st and torch.is_tensor(st["step"]) and st["step"].device != self.device:
                    st["step"] = st["step"].to(self.device)
        self.warmup_outputs = []
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(int(warmup)):
                self.warmup_outputs.append(self._detach(fn()))
        cur.wait_stream(side)
        torch.cuda.synchronize(self.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    @staticmethod
    def _detach(out):
        if torch.is_tensor(out):
            return out.detach().clone()
        return tuple(o.detach().clone() for o in out)

    def replay(self):
        """One iteration.  Returns the (static) outputs: clone what must survive the next replay."""
        self.graph.replay()
        return self.outputs
