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
