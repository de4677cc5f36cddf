"""Fixture loading + the tolerances used across the tests (test infrastructure)."""
import json
import os

import numpy as np

from zig_gpt2_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_ops():
    """name -> ndarray for every tensor of generate_test_data.py (inputs regenerated, outputs stored)."""
    with open(os.path.join(GOLDEN, "ops_manifest.json")) as f:
        man = json.load(f)["tensors"]
    stored = np.load(os.path.join(GOLDEN, "ops.npz"))
    out = {}
    for name, e in man.items():
        n = int(np.prod(e["shape"]))
        if e["gen"] == "stored":
            a = stored[name]
        elif e["gen"] == "normal":
            a = synth.fill_normal(e["seed"], n, e["mean"], e["std"])
        elif e["gen"] == "uniform":
            a = synth.fill_uniform(e["seed"], n, e["lo"], e["hi"])
        elif e["gen"] == "const":
            a = np.full(n, e["value"], np.float32)
        else:
            raise ValueError(e)
        out[name] = np.ascontiguousarray(a.reshape(e["shape"]))
    return out


def load_gpt(name):
    z = np.load(os.path.join(GOLDEN, f"gpt_{name}.npz"))
    cfg = synth.GPTConfig(*[int(v) for v in z["config"]])
    return cfg, {k: z[k] for k in z.files}


def assert_ref_close(expected, actual, what="", scale_floor=0.0):
    """expectTensorsApproxEqual of src/tests.zig:4-20: |e| < 1e-3 -> abs 5e-7, else rel 6e-4.

    scale_floor > 0 (seeded sweeps only, never the golden vectors) additionally accepts an absolute
    error of scale_floor * max|expected|: with long dot products and larger weights than the
    reference's fixtures, an output that happens to cancel to ~1e-3 of the tensor scale carries
    fp32 accumulation-order noise that no fp32 implementation (the oracle included) can avoid."""
    e = np.asarray(expected, np.float64).ravel()
    a = np.asarray(actual, np.float64).ravel()
    assert e.shape == a.shape, (what, e.shape, a.shape)
    small = np.abs(e) < 1e-3
    err = np.abs(e - a)
    # std.testing.expectApproxEqRel: |e - a| <= tol * max(|e|, |a|)
    ok = np.where(small, err <= 5e-7, err <= 6e-4 * np.maximum(np.abs(e), np.abs(a)))
    if scale_floor > 0 and e.size:
        ok |= err <= scale_floor * np.abs(e).max()
    if not ok.all():
        i = int(np.argmax(~ok))
        raise AssertionError(f"{what}: {int((~ok).sum())}/{e.size} outside reference tolerance; first at {i}: expected {e[i]!r} got {a[i]!r}")


def assert_model_close(expected, actual, what="", rtol=1e-3):
    """north_star tolerance for model-level fp32 outputs: relative 1e-3, with an absolute floor of
    rtol * 1e-2 * rms(expected) for elements that happen to lie near zero."""
    e = np.asarray(expected, np.float64).ravel()
    a = np.asarray(actual, np.float64).ravel()
    assert e.shape == a.shape, (what, e.shape, a.shape)
    floor = 1e-2 * float(np.sqrt(np.mean(e * e)))
    err = np.abs(e - a)
    ok = err <= rtol * np.maximum(np.abs(e), floor)
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    if not ok.all():
        i = int(np.argmax(err / np.maximum(np.abs(e), floor)))
        raise AssertionError(f"{what}: {int((~ok).sum())}/{e.size} outside rel {rtol}; worst at {i}: expected {e[i]!r} got {a[i]!r}")
    return float((err / np.maximum(np.abs(e), floor)).max())


def assert_greedy_ids_match(expected_ids, actual_ids, top1, top2, what="", gap_tol=1e-4):
    """Greedy ids must be identical; a differing id is tolerated only where the oracle's own top-2
    logit gap is below gap_tol (a numerical tie) — and is reported."""
    expected_ids = np.asarray(expected_ids).astype(np.int64)
    actual_ids = np.asarray(actual_ids).astype(np.int64)
    bad = np.nonzero(expected_ids != actual_ids)[0]
    for i in bad:
        gap = float(top1[i] - top2[i])
        assert gap < gap_tol, f"{what}: greedy id differs at step {i} (expected {expected_ids[i]} got {actual_ids[i]}), top-2 gap {gap:.3e}"
    return len(bad)
