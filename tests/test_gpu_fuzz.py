"""Randomised shape / leading-dimension / mode sweep through the C ABI (tools/fuzz_parity.py): properties on the device in fp64."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_and_leading_dimensions(seed):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(seed=seed, count=120, verbose=True) == 0


def test_extreme_input_scales():
    """Entries from 1e-30 to 1e25: where fp32 products of the bf16-split Gram level would overflow or sit near the denormal range
    the engine must step to the fp64 level (acceptance test on the Gram diagonal) -- full accuracy everywhere, never NaN."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "extreme_scale.py")
    spec = importlib.util.spec_from_file_location("extreme_scale", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(m=1 << 15, n=64, verbose=True) == 0


def test_robustness_probes():
    """Square matrices, columns / rows scaled over 16 / 6 decades, exactly dependent constant columns (bounded, residual at rounding
    level), non-finite input (returns)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "robust_probe.py")
    spec = importlib.util.spec_from_file_location("robust_probe", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(verbose=True) == 0
