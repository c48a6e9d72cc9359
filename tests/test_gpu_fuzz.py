"""Differential fuzz across the engines (tools/fuzz_engines.py): random regular, quasi-cyclic, NR-shaped and
unstructured codes, batch sizes around the engine thresholds, all three check-node rules, LLR and hard-word
input, early exit and fixed iterations; every engine must return the bits of the row-parallel sweeps."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_engines_agree_on_random_cases(pkg):
    spec = importlib.util.spec_from_file_location("fuzz_engines", os.path.join(ROOT, "tools", "fuzz_engines.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pkg.capi.set_default_options()
    seen, bad = mod.fuzz(250, 7)
    assert bad == 0, seen
    for kernel in ("bp_lds_kernel", "bp_lds_gen_kernel", "bp_nodes_kernel", "rows_cn_kernel", "bp_decode_kernel"):
        assert seen.get(kernel, 0) > 0, seen
