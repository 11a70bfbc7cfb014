"""CPU: ParamArena bookkeeping that needs no kernel -- per-slot gradient liveness (overwrite vs accumulate decided at
backward time), fused-group mixed state, outermost-forward shadow refresh, optimizer-step tracking."""
import torch
import torch.nn as nn

from icka_amd import arena as AR
from icka_amd.arena import ArenaModule, ParamArena


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 8)
        self.b = nn.Linear(8, 8)


def _write(A, p, value):
    """what a backward kernel does: beta from the arena, then out = value + beta * out"""
    beta = A.grad_beta(p)
    g = A.g(p)
    g.mul_(beta).add_(value)
    return beta


def test_liveness_is_decided_at_backward_time():
    m = Tiny()
    A = ParamArena(m)
    w = m.a.weight
    assert _write(A, w, 1.0) == 0.0 and w.grad is not None and w.grad.data_ptr() == A.g(w).data_ptr()
    assert _write(A, w, 1.0) == 1.0 and float(w.grad[0, 0]) == 2.0          # accumulation (grad_accum micro-steps)
    # zero_grad BETWEEN a forward and its backward (loss = model(x); opt.zero_grad(); loss.backward()): the drop is seen
    w.grad = None
    assert _write(A, w, 5.0) == 0.0 and float(w.grad[0, 0]) == 5.0
    # zero_grad(set_to_none=False) zeroes the arena view in place: accumulate onto zeros
    w.grad.zero_()
    assert _write(A, w, 3.0) == 1.0 and float(w.grad[0, 0]) == 3.0
    # a foreign gradient tensor (user assigned p.grad) is not ours to accumulate into
    w.grad = torch.full_like(w, 7.0)
    assert _write(A, w, 1.0) == 0.0 and float(w.grad[0, 0]) == 1.0 and w.grad.data_ptr() == A.g(w).data_ptr()


def test_partially_dropped_group_and_parameters_outside_the_optimizer():
    m = Tiny()
    A = ParamArena(m)
    group = (m.a.weight, m.a.bias)       # adjacent slots written by ONE fused kernel
    A.grad_beta(group)
    A.g(m.a.weight).fill_(2.0)
    A.g(m.a.bias).fill_(2.0)
    _write(A, m.b.weight, 1.0)
    opt = torch.optim.SGD([m.a.weight], lr=0.1)          # bias / module b are NOT in the optimizer
    opt.zero_grad()
    assert m.a.weight.grad is None and m.a.bias.grad is not None
    beta = A.grad_beta(group)                            # mixed state: fresh slot zeroed, group accumulates
    assert beta == 1.0
    assert float(A.g(m.a.weight).abs().max()) == 0.0 and float(A.g(m.a.bias)[0]) == 2.0
    assert _write(A, m.b.weight, 1.0) == 1.0 and float(m.b.weight.grad[0, 0]) == 2.0   # torch semantics: keeps adding
    A.zero_grad()
    assert _write(A, m.b.weight, 1.0) == 0.0


def test_shadow_refresh_runs_in_the_outermost_forward_only(monkeypatch):
    calls = []
    monkeypatch.setattr(ParamArena, "sync", lambda self, force=False: calls.append(1) or setattr(self, "_synced", 0))

    class Inner(ArenaModule):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)

        def forward(self, x):
            self._arena_cpu()
            return x

    class Outer(ArenaModule):
        def __init__(self):
            super().__init__()
            self.layers = nn.ModuleList([Inner() for _ in range(3)])

        def forward(self, x):
            self._arena_cpu()
            for l in self.layers:
                x = l(x)
            return x

    def _arena_cpu(self):   # ArenaModule._arena minus the "must be a ROCm device" check
        A = AR.arena_of(self)
        if AR._FWD_DEPTH[0] <= 1 or A._synced is None:
            A.sync()
        return A

    monkeypatch.setattr(ArenaModule, "_arena_cpu", _arena_cpu, raising=False)
    m = Outer()
    m(torch.zeros(1))
    assert len(calls) == 1                 # one cast per outermost forward ("always" policy), not one per block
    m(torch.zeros(1))
    assert len(calls) == 2
    m.layers[0](torch.zeros(1))            # a block called on its own is its own outermost forward
    assert len(calls) == 3
    assert AR._FWD_DEPTH[0] == 0


def test_optimizer_steps_are_counted_for_the_tracked_policy():
    m = Tiny()
    before = AR._OPT_STEPS[0]
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert AR._OPT_STEPS[0] == before + 1
