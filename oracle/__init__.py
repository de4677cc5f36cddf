"""CPU oracle (test infrastructure only) — see oracle/zgpt2_oracle.c header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .oracle import *  # noqa: F401,F403
