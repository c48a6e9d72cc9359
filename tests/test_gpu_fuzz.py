"""Differential fuzz across the engines (tools/fuzz_engines.py): random regular, quasi-cyclic, NR-shaped and
unstructured codes, batch sizes around the engine thresholds, all three check-node rules, early exit and
fixed iterations; every engine must return the bits of the row-parallel sweeps."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engines_agree_on_random_cases():
    env = {k: v for k, v in os.environ.items() if not k.startswith("LDPC_AMD_")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_engines.py"), "250", "7"], env=env,
                         capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout
    for kernel in ("bp_lds_kernel", "bp_lds_gen_kernel", "bp_nodes_kernel", "rows_cn_kernel", "bp_decode_kernel"):
        assert kernel in out.stdout, out.stdout[-500:]
