"""Raw weight directory format of the reference: one little-endian fp32 file per
tensor under `<dir>/`, named `model-<tf variable with '/' -> '-'>` (download_weights.py:57-64),
`*/w` matrices already transposed to [out, in] (download_weights.py:60-61) — exactly what
load_linear / load_layer_norm / load_embedding of src/main.zig:210-269 read with hard-coded
`models/124M/raw/` paths.  Here the directory and the config are arguments.

  model-wte, model-wpe, model-ln_f-{g,b},
  model-h{i}-ln_1-{g,b}, model-h{i}-attn-c_attn-{w,b}, model-h{i}-attn-c_proj-{w,b},
  model-h{i}-ln_2-{g,b}, model-h{i}-mlp-c_fc-{w,b}, model-h{i}-mlp-c_proj-{w,b}
"""
import os

import numpy as np

from .synth import GPTConfig, tensor_specs

# our tensor name -> reference file stem
_BLOCK_FILES = {
    "ln_1_g": "ln_1-g", "ln_1_b": "ln_1-b", "c_attn_w": "attn-c_attn-w", "c_attn_b": "attn-c_attn-b",
    "c_proj_w": "attn-c_proj-w", "c_proj_b": "attn-c_proj-b", "ln_2_g": "ln_2-g", "ln_2_b": "ln_2-b",
    "c_fc_w": "mlp-c_fc-w", "c_fc_b": "mlp-c_fc-b", "mlp_proj_w": "mlp-c_proj-w", "mlp_proj_b": "mlp-c_proj-b",
}
_TOP_FILES = {"wte": "wte", "wpe": "wpe", "ln_f_g": "ln_f-g", "ln_f_b": "ln_f-b"}


def file_name(tensor_name):
    if "." in tensor_name:
        layer, slot = tensor_name.split(".")
        return f"model-{layer}-{_BLOCK_FILES[slot]}"
    return f"model-{_TOP_FILES[tensor_name]}"


def load_raw_dir(path, cfg: GPTConfig):
    """Read every tensor of `cfg` from a reference-format directory.  Unlike the reference's
    load_tensor (src/ops.zig:318, short reads silently ignored) a size mismatch is an error."""
    out = {}
    for name, shape, _, _ in tensor_specs(cfg):
        f = os.path.join(path, file_name(name))
        a = np.fromfile(f, dtype="<f4")
        want = int(np.prod(shape))
        if a.size != want:
            raise ValueError(f"{f}: {a.size} floats on disk, config needs {want} for {name}{tuple(shape)}")
        out[name] = np.ascontiguousarray(a.reshape(shape))
    return out


def save_raw_dir(path, cfg: GPTConfig, weights):
    os.makedirs(path, exist_ok=True)
    for name, shape, _, _ in tensor_specs(cfg):
        np.ascontiguousarray(weights[name], dtype="<f4").reshape(-1).tofile(os.path.join(path, file_name(name)))


def bf16_representable(a):
    """True when rounding `a` to bf16 changes nothing (its low 16 mantissa bits are zero)."""
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return not bool(np.any(b & np.uint32(0xFFFF)))


def flags_for_checkpoint(weights):
    """How a handle should store the matrices of THIS checkpoint: bf16 storage is lossless only for bf16-representable
    weights (the synthetic ones); on an ordinary fp32 checkpoint it moves the logits by about 1e-2 of their scale
    (tests/test_weight_storage_gpu.py), outside north_star's 1e-3 — such a checkpoint gets ZG_GPT_WEIGHTS_F32, the reference's
    own precision (src/main.zig:210-269 keeps fp32).  Returns keyword arguments for gpt.GPT."""
    mats = [v for k, v in weights.items() if np.ndim(v) == 2]
    return {"weights_f32": not all(bf16_representable(m) for m in mats)}
