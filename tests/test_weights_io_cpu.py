"""Raw weight directory reader (reference format: download_weights.py:57-64, src/main.zig:210-269)."""
import numpy as np
import pytest

from zig_gpt2_amd import synth, weights_io


def test_file_names_follow_the_reference_convention():
    assert weights_io.file_name("wte") == "model-wte"                      # main.zig:305 load_embedding("wte")
    assert weights_io.file_name("ln_f_g") == "model-ln_f-g"                # main.zig:311 + :240
    assert weights_io.file_name("h3.c_attn_w") == "model-h3-attn-c_attn-w"  # main.zig:276 + :216
    assert weights_io.file_name("h11.mlp_proj_b") == "model-h11-mlp-c_proj-b"
    assert weights_io.file_name("h0.ln_2_g") == "model-h0-ln_2-g"


def test_round_trip_and_size_check(tmp_path):
    cfg = synth.CONFIGS["tiny3"]
    w = synth.make_weights(cfg, seed=3, bf16=False)
    weights_io.save_raw_dir(tmp_path, cfg, w)
    r = weights_io.load_raw_dir(tmp_path, cfg)
    assert set(r) == set(w)
    for k in w:
        assert r[k].dtype == np.float32 and np.array_equal(r[k], w[k]), k
    with open(tmp_path / "model-h1-mlp-c_fc-b", "ab") as f:  # wrong size must not pass silently
        f.write(b"\0\0\0\0")
    with pytest.raises(ValueError):
        weights_io.load_raw_dir(tmp_path, cfg)


def test_checkpoint_storage_policy():
    """bf16 storage is lossless only for bf16-representable matrices; anything else keeps the reference's fp32
    (tests/test_weight_storage_gpu.py measures why: 6e-3 of the logit scale at 124M)."""
    cfg = synth.CONFIGS["tiny"]
    assert weights_io.flags_for_checkpoint(synth.make_weights(cfg, seed=3, bf16=True)) == {"weights_f32": False}
    w = synth.make_weights(cfg, seed=3, bf16=False)
    assert weights_io.flags_for_checkpoint(w) == {"weights_f32": True}
    # one unrounded matrix is enough; vectors (biases, LayerNorm) never decide — they stay fp32 in every handle
    w2 = synth.make_weights(cfg, seed=3, bf16=True)
    w2["h0.c_attn_b"] = w["h0.c_attn_b"]
    assert weights_io.flags_for_checkpoint(w2) == {"weights_f32": False}
    w2["h1.c_fc_w"] = w["h1.c_fc_w"]
    assert weights_io.flags_for_checkpoint(w2) == {"weights_f32": True}
