"""CPU: the algebra and the rounding of the backward recurrence's reduce-scatter form (lstm.hip: lstm_bwd_rs_kernel) in torch.
Each block of 16 hidden units multiplies ITS 64 gate columns of dgates_t by W_hh and sends bf16 partial tiles to the owners of
the columns; the owners add the H/16 tiles to dy.  Checked against autograd through a float64 nn.LSTM-style cell: same
gradients up to the bf16 rounding of the operands (dgates, W_hh) and of the partial tiles."""
import torch


def test_reduce_scatter_backward_matches_autograd():
    torch.manual_seed(0)
    B, S, H, U = 4, 10, 64, 16
    nb = H // U
    f64 = torch.float64
    W = (torch.randn(4 * H, H, dtype=f64) * H ** -0.5)          # W_hh, gate-major rows (i, f, g, o)
    gx = torch.randn(B, S, 4 * H, dtype=f64) * 0.7              # input projection + biases
    dy = torch.randn(B, S, H, dtype=f64) * 0.3

    # reference: autograd through the float64 recurrence
    gxr = gx.clone().requires_grad_(True)
    h = torch.zeros(B, H, dtype=f64); c = torch.zeros(B, H, dtype=f64)
    hs, acts, cs = [], [], []
    for t in range(S):
        z = gxr[:, t] + h @ W.t()
        i, f, g, o = z[:, :H].sigmoid(), z[:, H:2 * H].sigmoid(), z[:, 2 * H:3 * H].tanh(), z[:, 3 * H:].sigmoid()
        c = f * c + i * g
        h = o * c.tanh()
        hs.append(h); acts.append((i.detach(), f.detach(), g.detach(), o.detach())); cs.append(c.detach())
    (torch.stack(hs, 1) * dy).sum().backward()
    ref = gxr.grad                                                # = dgates of every step

    # reduce-scatter form with the kernel's roundings
    bf = lambda x: x.to(torch.bfloat16).to(f64)
    Wb = bf(W)
    dg_all = torch.zeros(B, S, 4 * H, dtype=f64)
    carry = torch.zeros(B, H, dtype=f64)
    partial_sum = torch.zeros(B, H, dtype=f64)                    # sum of the tiles addressed to each unit (from step t+1)
    for t in range(S - 1, -1, -1):
        i, f, g, o = acts[t]
        cprev = cs[t - 1] if t > 0 else torch.zeros(B, H, dtype=f64)
        dh = dy[:, t] + partial_sum
        tc = cs[t].tanh()
        dc = dh * o * (1 - tc * tc) + carry
        carry = dc * f
        dgt = bf(torch.cat([dc * g * i * (1 - i), dc * cprev * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)], 1))
        dg_all[:, t] = dgt
        partial_sum = torch.zeros(B, H, dtype=f64)
        for blk in range(nb):                                     # producer blk: its 4 x 16 gate columns times W_hh rows
            cols = torch.cat([torch.arange(q * H + blk * U, q * H + (blk + 1) * U) for q in range(4)])
            P = dgt[:, cols] @ Wb[cols, :]                        # [B, H], K = 64
            partial_sum += bf(P)                                  # tiles travel as bf16 and are summed by their owners
    rel = ((dg_all - ref).norm() / ref.norm()).item()
    assert rel < 1e-2, rel
    # and without any rounding the two forms are the same recurrence
    bf = lambda x: x
    Wb = W
    carry = torch.zeros(B, H, dtype=f64); partial_sum = torch.zeros(B, H, dtype=f64)
    exact = torch.zeros_like(dg_all)
    for t in range(S - 1, -1, -1):
        i, f, g, o = acts[t]
        cprev = cs[t - 1] if t > 0 else torch.zeros(B, H, dtype=f64)
        dh = dy[:, t] + partial_sum
        tc = cs[t].tanh()
        dc = dh * o * (1 - tc * tc) + carry
        carry = dc * f
        dgt = torch.cat([dc * g * i * (1 - i), dc * cprev * f * (1 - f), dc * i * (1 - g * g), dh * tc * o * (1 - o)], 1)
        exact[:, t] = dgt
        partial_sum = sum(dgt[:, torch.cat([torch.arange(q * H + b * U, q * H + (b + 1) * U) for q in range(4)])] @
                          Wb[torch.cat([torch.arange(q * H + b * U, q * H + (b + 1) * U) for q in range(4)]), :] for b in range(nb))
    assert ((exact - ref).norm() / ref.norm()).item() < 1e-12
