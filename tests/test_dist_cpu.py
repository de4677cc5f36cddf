"""The C++ host's prompt partition (zgpt2_main --gpus N --plan: no GPU is touched) against zig_gpt2_amd.shard.shard_prompts,
the partition bench.py uses — one process per GPU, contiguous blocks whose sizes differ by at most one."""
import os
import subprocess

import pytest

from zig_gpt2_amd import shard

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zig_gpt2_amd", "bin", "zgpt2_main")


@pytest.mark.parametrize("n_prompts,world", [(1, 1), (5, 3), (8, 8), (64, 8), (7, 2), (3, 4)])
def test_cpp_host_partition_equals_shard_prompts(n_prompts, world):
    if not os.path.exists(BIN):
        pytest.skip("zgpt2_main not built (python -c 'import __graft_entry__ as g; g.build()')")
    arg = ";".join(str(i + 1) for i in range(n_prompts))
    out = subprocess.run([BIN, "tiny", "1", arg, "4", "--gpus", str(world), "--plan"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert len(lines) == world
    for r, line in enumerate(lines):
        head, _, rest = line.partition(":")
        assert head == f"rank {r}"
        assert [int(t) for t in rest.split()] == shard.shard_prompts(n_prompts, world, r)
