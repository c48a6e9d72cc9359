"""Differential fuzz across the engines (tools/fuzz_engines.py): random regular, quasi-cyclic, NR-shaped and
unstructured codes, batch sizes around the engine thresholds, all three check-node rules, LLR and hard-word
input, early exit and fixed iterations; every engine must return the bits of the row-parallel sweeps."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("LDPC_AMD_ENGINE", "LDPC_AMD_ROWS", "LDPC_AMD_NODES", "LDPC_AMD_GENERIC_LDS_KB")


def test_engines_agree_on_random_cases(pkg):
    spec = importlib.util.spec_from_file_location("fuzz_engines", os.path.join(ROOT, "tools", "fuzz_engines.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = {k: os.environ.pop(k, None) for k in KEYS}
    try:
        seen, bad = mod.fuzz(250, 7)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert bad == 0, seen
    for kernel in ("bp_lds_kernel", "bp_lds_gen_kernel", "bp_nodes_kernel", "rows_cn_kernel", "bp_decode_kernel"):
        assert seen.get(kernel, 0) > 0, seen
