"""Weights: deterministic synthetic state dicts (no checkpoint exists offline: SURVEY.md §0.6) and safe loaders.

The generator is pure numpy with a per-tensor Philox stream keyed by (seed, crc32(name)), so the same state dict
is rebuilt bit-for-bit in this container (to produce the committed golden vectors) and on the GPU box (to run the
HIP path against them) regardless of creation order.  State dicts use the Hugging Face / Ultralytics parameter names so
the same loader packs real checkpoints (``load_state_dict_file``: safetensors or ``torch.load(weights_only=True)``).
"""
import zlib

import numpy as np


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def synth_tensor(seed, name, shape, kind):
    """kind: 'w' (fan-in scaled normal; fan_in = prod(shape[1:])), 'b' (small normal), 'g' (norm gain around 1),
    'ls' (LayerScale in [0.5, 1.5)), 'tok' (unit-ish normal: tokens, position tables), 'zero'."""
    r = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if kind.startswith("wc"):  # conv followed by SiLU: "wc@g" = normal * g / sqrt(fan_in); g = 1.68 / rms(input)
        fan_in = int(np.prod(shape[1:]))  # (1.68 = 1/rms(silu(N(0,1))) keeps activations O(1) through ~60 layers)
        gain = float(kind.split("@")[1]) if "@" in kind else 1.68
        return (gain * r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if kind == "bnb":  # BatchNorm beta > 0: SiLU mostly in its near-linear range, so that a random 60-layer stack
        return (1.0 + 0.3 * r.standard_normal(shape)).astype(np.float32)  # does not amplify rounding chaotically
    if kind == "var":  # BatchNorm running_var
        return (0.5 + r.random(shape)).astype(np.float32)
    if kind == "boxb":  # Detect box-branch bias (ultralytics bias_init sets 1.0)
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    if kind == "clsb":  # Detect class-branch bias: strongly negative, as in a trained detector (few positives)
        return (-5.0 + 0.5 * r.standard_normal(shape)).astype(np.float32)
    if kind == "b":
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if kind == "g":
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    if kind == "ls":
        return (0.5 + r.random(shape)).astype(np.float32)
    if kind == "tok":
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    if kind == "zero":
        return np.zeros(shape, np.float32)
    raise ValueError(kind)


def synth_state_dict(spec, seed):
    """spec: ordered {name: (shape, kind)} -> {name: np.float32 array}."""
    return {name: synth_tensor(seed, name, shape, kind) for name, (shape, kind) in spec.items()}


def load_state_dict_file(path):
    """Load a real checkpoint without executing anything from the file: .safetensors, or torch.load(weights_only=True)
    for .pt/.pth holding a plain tensor dict.  Returns {name: np.float32}."""
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        return {k: np.asarray(v, dtype=np.float32) for k, v in load_file(path).items()}
    import torch

    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v.detach().to(torch.float32).numpy() for k, v in sd.items() if hasattr(v, "detach")}
