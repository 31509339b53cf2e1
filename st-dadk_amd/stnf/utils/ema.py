"""Exponential moving average of the trainable parameters (reference stnf/utils/ema.py:9-105).

Same public surface (decay, shadow/backup dicts keyed by parameter name, update / apply_shadow /
restore / state_dict / load_state_dict).  update() is one fused multi-tensor lerp on the device
instead of a Python loop of three kernels per parameter."""
import torch
import torch.nn as nn


class ModelEMA:
    def __init__(self, model: nn.Module, decay: float = 0.999):
        self.decay = decay
        self.model = model
        self.shadow = {n: p.data.clone() for n, p in model.named_parameters() if p.requires_grad}
        self.backup = {}

    def update(self, model: nn.Module):
        """shadow = decay*shadow + (1-decay)*param, after optimizer.step()."""
        with torch.no_grad():
            names, params = [], []
            for n, p in model.named_parameters():
                if p.requires_grad:
                    assert n in self.shadow, f"Parameter {n} not in shadow"
                    names.append(n)
                    params.append(p.data)
            shadows = [self.shadow[n] for n in names]
            # lerp(s, p, 1-d) = s + (1-d)(p-s) = d*s + (1-d)*p
            torch._foreach_lerp_(shadows, params, 1.0 - self.decay)

    def apply_shadow(self):
        """Swap the EMA weights in for validation (reference ema.py:68-77).  The values are copied INTO the
        parameters' existing storage: a parameter may be a view of an engine's flat buffer (stnf.engine), and
        rebinding `.data` would silently detach it from the buffer the fused kernels train."""
        with torch.no_grad():
            for n, p in self.model.named_parameters():
                if p.requires_grad:
                    self.backup[n] = p.data.clone()
                    p.copy_(self.shadow[n])          # on the parameter, not .data: the autograd version moves
        self._touched()

    def restore(self):
        """Put the training weights back (reference ema.py:79-89), again in place."""
        with torch.no_grad():
            for n, p in self.model.named_parameters():
                if p.requires_grad:
                    p.copy_(self.backup[n])
        self.backup = {}
        self._touched()

    def _touched(self):
        """Everything derived from the weights (a Predictor's cached output layer / transposed W0, an engine's bf16
        operand copies) keys on the parameters' versions and on this counter: both move with every swap."""
        self.model._engine_version = getattr(self.model, "_engine_version", 0) + 1

    def state_dict(self):
        return {'decay': self.decay, 'shadow': self.shadow}

    def load_state_dict(self, state_dict):
        self.decay = state_dict['decay']
        self.shadow = state_dict['shadow']
