"""Plain PyTorch-CPU port of the reference's train step.  TEST INFRASTRUCTURE ONLY.

Purpose: the *timed CPU baseline* of bench.py (`cpu_baseline.kind == "port"`) — the reference's
Python files cannot travel to the GPU box, so this file restates the same torch op sequence the
reference executes per batch (torch.cdist with its default compute mode, ~8 elementwise passes
for the basis, nn.Linear / nn.LayerNorm / nn.ReLU, nn.MSELoss, autograd backward,
clip_grad_norm_, AdamW, EMA) so that its wall time on the box's host cores is what the reference
would take there.  It is also a second, independent checker for the product path.

Parity status: PINNED against tests/golden/*.npz (generated from the real reference) in
tests/test_oracle_golden.py — fp32 outputs agree with the reference's own fp32 run to rounding
because the op sequence is the same.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Citations are relative to the reference root.
"""
import math

import torch
import torch.nn.functional as F

CAL = {"wendland": 1.000000, "gaussian": 0.223477, "triangular": 0.654714}


def grid_knots(k_list):
    """stnf/models/st_interp.py:152-185 — uniform multi-resolution knot table."""
    cs, bs = [], []
    for k in k_list:
        side = int(math.sqrt(k))
        assert side * side == k, f"n_centers must be perfect squares, got {k}"
        lin = torch.linspace(0, 1, side)
        gx, gy = torch.meshgrid(lin, lin, indexing="ij")
        cs.append(torch.stack([gx.reshape(-1), gy.reshape(-1)], -1))
        bs.append(torch.full((k,), 2.5 * (1.0 / (side - 1) if side > 1 else 1.0)))
    return torch.cat(cs), torch.cat(bs)


def time_knots(n_list):
    """stnf/models/st_interp.py:557-581."""
    cs = [torch.linspace(0.0, 1.0, n) for n in n_list]
    bs = [torch.full((n,), 2.5 * (1.0 / (n - 1) if n > 1 else 1.0)) for n in n_list]
    return torch.cat(cs), torch.cat(bs)


def phi_ref(coords, centers, bw, basis="wendland"):
    """stnf/models/st_interp.py:433-491, same op order (cdist default mode => matmul expansion)."""
    r = torch.cdist(coords.unsqueeze(0), centers.unsqueeze(0)).squeeze(0) / (bw * CAL[basis])
    if basis == "wendland":
        r = r.clamp(max=1.0)
        return torch.pow(1 - r, 6) * (35 * r ** 2 + 18 * r + 3) / 3
    if basis == "gaussian":
        return torch.exp(-0.5 * r ** 2)
    if basis == "triangular":
        return torch.clamp(1 - r, min=0.0)
    raise ValueError(f"Unknown basis function: {basis}")


def psi_ref(t, centers, bw):
    """stnf/models/st_interp.py:583-596."""
    return torch.exp(-0.5 * ((t - centers.view(1, -1)) / bw.view(1, -1)) ** 2)


class PortModel:
    """Functional model: parameters in a dict keyed like the reference's state_dict."""

    def __init__(self, cfg, state=None, dtype=torch.float32, seed=0):
        self.cfg = cfg
        self.dtype = dtype
        self.centers, self.bw = (a.to(dtype) for a in grid_knots(cfg["k_spatial_centers"]))
        self.tc, self.tb = (a.to(dtype) for a in time_knots(cfg["k_temporal_centers"]))
        self.p = cfg.get("p", 0)
        self.ln = cfg.get("layernorm", True)
        self.drop = cfg.get("dropout", 0.0)
        self.basis = cfg.get("basis", "wendland")
        D = self.p + self.centers.shape[0] + self.tc.shape[0]
        self.params = {}
        self.order = []          # [(kind, key_w, key_b)] in Sequential order
        g = torch.Generator().manual_seed(seed)
        idx, prev = 0, D
        for h in list(cfg["hidden_dims"]) + [None]:
            out = cfg.get("output_dim", 1) if h is None else h
            b = 1.0 / math.sqrt(prev)
            W = (torch.rand(out, prev, generator=g) * 2 - 1) * b
            bias = (torch.rand(out, generator=g) * 2 - 1) * b
            self._add("lin", idx, W, bias)
            idx += 1
            if h is None:
                break
            if self.ln:
                self._add("ln", idx, torch.ones(h), torch.zeros(h))
                idx += 1
            idx += 1                      # ReLU
            if self.drop > 0:
                idx += 1                  # Dropout
            prev = h
        if state is not None:
            for k, v in state.items():
                self.params[k] = torch.as_tensor(v).to(dtype).clone().requires_grad_(True)

    def _add(self, kind, idx, w, b):
        kw, kb = f"mlp.{idx}.weight", f"mlp.{idx}.bias"
        self.params[kw] = w.to(self.dtype).requires_grad_(True)
        self.params[kb] = b.to(self.dtype).requires_grad_(True)
        self.order.append((kind, kw, kb))

    def forward(self, X, coords, t, train=True):
        """stnf/models/st_interp.py:827-882 (standard head)."""
        phi = phi_ref(coords, self.centers, self.bw, self.basis)
        psi = psi_ref(t, self.tc, self.tb)
        a = torch.cat([X, phi, psi], -1) if (self.p > 0 and X is not None and X.numel() > 0) \
            else torch.cat([phi, psi], -1)
        n_lin = sum(1 for o in self.order if o[0] == "lin")
        seen = 0
        for kind, kw, kb in self.order:
            if kind == "lin":
                seen += 1
                a = F.linear(a, self.params[kw], self.params[kb])
                if seen < n_lin and not self.ln:
                    a = torch.relu(a)
                    if self.drop > 0:
                        a = F.dropout(a, self.drop, train)
            else:
                a = F.layer_norm(a, (a.shape[-1],), self.params[kw], self.params[kb], 1e-5)
                a = torch.relu(a)
                if self.drop > 0:
                    a = F.dropout(a, self.drop, train)
        return a


class PortTrainer:
    """The reference's batch body (scripts/train_st_interp.py:608-721) on a PortModel."""

    def __init__(self, model, lr=2e-2, weight_decay=5e-4, grad_clip=10.0, ema_decay=0.999,
                 betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.plist = list(model.params.values())
        self.opt = torch.optim.AdamW(self.plist, lr=lr, weight_decay=weight_decay, betas=betas,
                                     eps=eps)
        self.grad_clip = grad_clip
        self.ema_decay = ema_decay
        self.shadow = {k: v.detach().clone() for k, v in model.params.items()}

    def step(self, X, coords, t, y):
        self.opt.zero_grad()
        yp = self.model.forward(X, coords, t, train=True)
        loss = F.mse_loss(yp, y)
        loss.backward()
        if self.grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(self.plist, self.grad_clip)
        self.opt.step()
        with torch.no_grad():
            d = self.ema_decay
            for k, v in self.model.params.items():
                self.shadow[k] = d * self.shadow[k] + (1.0 - d) * v.detach()
        return loss.item()
