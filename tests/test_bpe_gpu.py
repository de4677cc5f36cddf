"""The tokenizer's C-ABI tests (zg_bpe_*, src/bpe.zig:59-118 restated as host code) once more under the `gpu` marker,
so that they also run on the driver's GPU box against the library that was built for it."""
import pytest

import test_bpe_cpu as cpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_c_abi_encoder_matches_restatement_on_the_gpu_box(seed):
    cpu.test_c_abi_encoder_matches_restatement(seed)


def test_reference_quirks_on_the_gpu_box():
    cpu.test_reference_quirks()
    cpu.test_greedy_prefix_drops_the_rest_of_an_unknown_word()
    cpu.test_errors()
