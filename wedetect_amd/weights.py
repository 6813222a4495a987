"""Name-keyed deterministic synthetic weights with trained-like statistics.

No checkpoints exist offline, so every parity test, golden fixture and benchmark
uses weights regenerated from ``(seed, tensor name)`` — identical here, on the GPU
box and inside the golden-generation script that loads them into the reference's
own modules.  The statistics are chosen so that activations stay
O(1) through 36 ConvNeXt blocks and scores spread over a realistic range:
BN running_var in [0.5, 1.5], non-zero running_mean, layer-scale gamma O(0.1-0.5),
distinct per-level logit_scale / bias, unit-norm prompt rows.

Also implements the checkpoint key remap that the reference applies before
``load_state_dict`` into its pure-torch copy (generate_proposal.py:1236-1254):
mmdet ConvModule names -> nn.Sequential indices 0,1,3,4,6.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .arch import ArchSpec, all_params, get_arch


def _rng(seed: int, name: str) -> np.random.Generator:
    key = (np.uint64(seed) << np.uint64(32)) | np.uint64(zlib.crc32(name.encode()))
    return np.random.Generator(np.random.Philox(key=int(key)))


def _uniform(rng, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = rng.random(n, dtype=np.float64)
    return (lo + (hi - lo) * u).reshape(shape).astype(np.float32)


def _normal(rng, shape, std):
    # Box-Muller on the raw uniform stream: independent of numpy's ziggurat tables.
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u1 = rng.random(m, dtype=np.float64)
    u2 = rng.random(m, dtype=np.float64)
    r = np.sqrt(-2.0 * np.log1p(-u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (std * z).reshape(shape).astype(np.float32)


_NECK_GAIN = 1.2
_ALPHA = (0.4, 0.7)
_EMBED_GAIN = 60.0
_LEVEL_LOGIT_SCALE = (-0.35, -0.55, -0.20)   # exp() ~ 0.70 / 0.58 / 0.82
_LEVEL_BIAS = (-2.6, -2.2, -1.9)


def make_tensor(name: str, shape, role: str, seed: int) -> np.ndarray:
    rng = _rng(seed, name)
    if role == "conv":            # (O, I, kh, kw)
        fan_in = shape[1] * shape[2] * shape[3]
        # backbone patchify convs see zero-mean LN outputs; neck/head convs see
        # non-negative ReLU/SiLU outputs and sit in residual chains -> smaller gain
        gain = 1.6 if name.startswith("backbone.") else _NECK_GAIN
        if ".cls_preds." in name and name.endswith(".2.weight"):
            gain = _EMBED_GAIN      # final 256->768 projection: embeddings of O(1) per element
        return _normal(rng, shape, (gain / fan_in) ** 0.5)
    if role == "deconv":          # ConvTranspose2d (I, O, 2, 2): one tap per output pixel
        return _normal(rng, shape, (_NECK_GAIN / shape[0]) ** 0.5)
    if role == "dwconv":          # (C, 1, 7, 7)
        return _normal(rng, shape, (1.0 / 49.0) ** 0.5)
    if role == "linear":          # (out, in)
        return _normal(rng, shape, (1.2 / shape[1]) ** 0.5)
    if role == "bias":
        return _uniform(rng, shape, -0.1, 0.1)
    if role == "dfl_bias":        # spreads the 16-bin distributions a little
        return _uniform(rng, shape, -0.5, 0.5)
    if role == "ln_w":
        return _uniform(rng, shape, 0.8, 1.2)
    if role == "ln_b":
        return _uniform(rng, shape, -0.1, 0.1)
    if role == "layer_scale":
        return _uniform(rng, shape, 0.1, 0.5)
    if role == "bn_w":
        return _uniform(rng, shape, 0.5, 1.5)
    if role == "bn_b":
        return _uniform(rng, shape, -0.2, 0.2)
    if role == "bn_m":
        return _uniform(rng, shape, -0.3, 0.3)
    if role == "bn_v":
        return _uniform(rng, shape, 0.5, 1.5)
    if role == "alpha":
        return _uniform(rng, shape, _ALPHA[0], _ALPHA[1])
    if role.startswith("logit_scale"):
        return np.asarray(_LEVEL_LOGIT_SCALE[int(role[-1])], dtype=np.float32).reshape(shape)
    if role.startswith("contrast_bias"):
        return np.asarray(_LEVEL_BIAS[int(role[-1])], dtype=np.float32).reshape(shape)
    if role == "prompts":
        w = _normal(rng, shape, 1.0).astype(np.float64)
        w /= np.linalg.norm(w, axis=-1, keepdims=True)
        return w.astype(np.float32)
    raise ValueError(f"unknown role {role!r} for {name}")


def make_state_dict(arch, seed: int = 2026, num_prompts: int = 0) -> Dict[str, np.ndarray]:
    """Full image-side state dict in mmdet checkpoint naming (numpy fp32)."""
    a: ArchSpec = get_arch(arch) if isinstance(arch, str) else arch
    return {name: make_tensor(name, shape, role, seed)
            for name, shape, role in all_params(a, num_prompts)}


def make_text_bank(k: int, dim: int = 768, seed: int = 4321) -> np.ndarray:
    """Synthetic stand-in for the XLM-R class-embedding bank: N(0,1) rows, L2-normalised
    (the text tower's own last step, mm_backbone.py:388-389).  Tokenizer blobs are
    missing offline so real strings cannot be encoded."""
    rng = _rng(seed, f"text_bank[{k},{dim}]")
    w = _normal(rng, (k, dim), 1.0).astype(np.float64)
    w /= np.linalg.norm(w, axis=-1, keepdims=True)
    return w.astype(np.float32)


def make_regions(r: int, dim: int = 768, seed: int = 11, std: float = 1.4) -> np.ndarray:
    """Synthetic region embeddings [R, dim] (post-BN statistics of the head: ~N(0, 1.4^2))."""
    return _normal(_rng(seed, f"regions[{r},{dim}]"), (r, dim), std)


def make_images(b: int, h: int, w: int, seed: int = 1234) -> np.ndarray:
    """uint8 RGB images [B, H, W, 3], uniform — the synthetic input of bench and tests."""
    rng = _rng(seed, f"images[{b},{h},{w}]")
    return (rng.random((b, h, w, 3), dtype=np.float32) * 256.0).astype(np.uint8)


# --------------------------------------------------------------------------
# key remap: mmdet checkpoint names -> the reference's pure-torch module names
# --------------------------------------------------------------------------
def to_uni_keys(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Rename like generate_proposal.py:1236-1254 does before load_state_dict:
    ``backbone.image_model.model.X`` -> ``backbone.X``;
    ``bbox_head.head_module.{cls,reg}_preds.L.{0,1}.{conv,bn}`` -> Sequential slots
    0,1 / 3,4 and the final 1x1 conv ``.2`` -> ``.6``."""
    out = {}
    for k, v in sd.items():
        nk = k
        if nk.startswith("backbone.image_model.model."):
            nk = "backbone." + nk[len("backbone.image_model.model."):]
        elif nk.startswith("bbox_head.head_module."):
            nk = "bbox_head." + nk[len("bbox_head.head_module."):]
            parts = nk.split(".")
            # bbox_head . (cls_preds|reg_preds) . L . slot . ...
            if parts[1] in ("cls_preds", "reg_preds"):
                slot = parts[3]
                if slot == "2":
                    parts[3] = "6"
                    nk = ".".join(parts)
                else:
                    base = 0 if slot == "0" else 3
                    sub = parts[4]            # conv | bn
                    parts[3] = str(base + (0 if sub == "conv" else 1))
                    del parts[4]
                    nk = ".".join(parts)
        out[nk] = v
    return out
