"""Randomised parity: detection and matching through the C ABI against the oracle on random shapes, parameters, tie-heavy
and degenerate inputs (scripts/fuzz_detect.py, scripts/fuzz_match.py hold the generators), bit for bit."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_detection_fuzz_is_bit_exact():
    import fuzz_detect
    assert fuzz_detect.run(160, seed=11) == 0


def test_matcher_fuzz_is_bit_exact():
    import fuzz_match
    assert fuzz_match.run(80, seed=12) == 0
