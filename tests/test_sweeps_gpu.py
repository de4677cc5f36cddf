"""The seeded random sweeps (tests/sweeps/*.py: each drives one public surface with random shapes / arguments against the
oracle or float64 numpy and exits non-zero when a case fails) as part of the suite: a seconds-sized slice of every sweep by
default, the long form under `-m fuzz` (minutes).  Each sweep is its own process — several of them end by checking that the
library still works after hundreds of refused calls.  `python tests/sweeps/<name>.py [first_seed] [count]` runs one by hand."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name: (count in the suite, count under -m fuzz)
SWEEPS = {
    "ops": (30, 600), "gemm": (12, 200), "attn": (12, 200), "prefill": (6, 150), "prefill_kv": (8, 150), "generate": (8, 200),
    "session": (5, 150), "sample": (10, 150), "errors": (150, 1500), "errors_gpt": (80, 800), "bpe": (100, 2000),
}


def run_sweep(name, first, count):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", f"{name}.py"), str(first), str(count)], cwd=ROOT,
                         capture_output=True, text=True, timeout=3000)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, f"sweep {name} from seed {first}, {count} cases:\n{tail}"
    return out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SWEEPS))
def test_sweep_slice(name):
    print(run_sweep(name, 500, SWEEPS[name][0]))


@pytest.mark.gpu
@pytest.mark.fuzz
@pytest.mark.parametrize("name", sorted(SWEEPS))
def test_sweep_long(name):
    print(run_sweep(name, 40000, SWEEPS[name][1]))
