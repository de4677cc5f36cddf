"""zig_gpt2_amd — MI355X-native GPT-2 forward hot path behind zig_gpt2's ops.zig interface.

The compute lives in csrc/ (hand-written HIP for gfx950) behind the C ABI of include/zgpt2.h;
this package is the Python-side mirror of the reference's ops/model interface used by the tests
and the benchmark.  Nothing here falls back to a CPU implementation.
"""
from .synth import CONFIGS, GPTConfig  # noqa: F401
