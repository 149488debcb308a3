"""Closed-form deterministic parameter fill shared by the golden-vector generator and the tests.

Weights are not stored in the fixtures: both the reference (when the fixtures are made) and the
implementations under test load exactly this fill through load_state_dict, so a fixture only
holds inputs and expected outputs.  The fill is deliberately "unfriendly": BatchNorm scales of
both signs (the max-over-k of a stage with negative scale is attained at the minimum of the
pre-activation), non-trivial shifts and running statistics.
"""
import math

import torch


def _phase(name):
    return float(sum((i + 1) * ord(ch) for i, ch in enumerate(name)) % 1000) * 0.001 * 2.0 * math.pi


def fill_tensor(name, t):
    """Return the closed-form value of state_dict entry `name` with the shape/dtype of `t`."""
    n = t.numel()
    i = torch.arange(n, dtype=torch.float64)
    ph = _phase(name)
    if name.endswith("num_batches_tracked"):
        return torch.zeros_like(t)
    if name.endswith("running_mean"):
        v = 0.2 * torch.sin(0.731 * i + ph)
    elif name.endswith("running_var"):
        v = 0.6 + 0.5 * torch.sin(0.377 * i + ph) ** 2
    elif t.dim() == 1 and name.endswith("weight"):      # BatchNorm scale: both signs, away from 0
        mag = 0.7 + 0.5 * torch.sin(0.913 * i + ph) ** 2
        sign = torch.where((i.long() % 5) == 3, -1.0, 1.0).double()
        v = mag * sign
    elif t.dim() == 1 and name.endswith("bias"):        # BatchNorm shift
        v = 0.15 * torch.cos(0.577 * i + ph)
    else:                                                # convolution weight [out, in, 1(,1)]
        fan_in = t.shape[1]
        v = math.sqrt(3.0 / fan_in) * torch.sin(1.6180339887 * i + 0.0007 * i * i % 6.2831853 + ph)
    return v.reshape(t.shape).to(t.dtype)


def fill_module(module):
    """Overwrite every parameter and buffer of `module` with the closed-form fill (in place)."""
    sd = module.state_dict()
    new = {k: fill_tensor(k, v) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def cotangent(shape, tag="R"):
    """Fixed closed-form cotangent for gradient fixtures."""
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    return (torch.sin(0.37 * i + _phase(tag)) + 0.25 * torch.cos(0.011 * i)).reshape(shape).float()
