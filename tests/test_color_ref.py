"""CPU: pins oracle/color_ref.py (the restatement of the reference's ColorJitter arithmetic) against fixtures
produced by the real reference (tests/golden/color_jitter.npz, oracle/make_golden.py)."""
import os
import sys

import pytest
import torch

from _util import golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import color_ref  # noqa: E402


def _t(d, k):
    return torch.from_numpy(d[k])


def test_single_stages_match_reference():
    d = golden("color_jitter")
    x = _t(d, "x")
    assert torch.equal(color_ref.adjust_brightness_accumulative(x, _t(d, "bf")), _t(d, "brightness"))
    assert torch.allclose(color_ref.adjust_contrast_with_mean_subtraction(x, _t(d, "cf")), _t(d, "contrast"), atol=1e-7, rtol=0)
    assert torch.equal(color_ref.adjust_saturation_with_gray_subtraction(x, _t(d, "sf")), _t(d, "saturation"))
    # the reference multiplies by its float32 pi tensor, the restatement by math.pi: one ulp of the shift
    assert torch.allclose(color_ref.adjust_hue(x, _t(d, "hf") * 2 * 3.141592653589793), _t(d, "hue"), atol=2e-6, rtol=0)


@pytest.mark.parametrize("order", ["0123", "3210", "2031", "1302"])
def test_sequences_match_reference(order):
    d = golden("color_jitter")
    out = color_ref.color_jitter(_t(d, "x"), _t(d, "bf"), _t(d, "cf"), _t(d, "sf"), _t(d, "hf"), [int(c) for c in order])
    assert torch.allclose(out, _t(d, "seq_" + order), atol=3e-6, rtol=0)


def test_module_replay_matches_reference():
    d = golden("color_jitter")
    out = color_ref.color_jitter(_t(d, "x"), _t(d, "bf"), _t(d, "cf"), _t(d, "sf"), _t(d, "hf"), [2, 0, 3, 1])
    assert torch.allclose(out, _t(d, "module_2031"), atol=3e-6, rtol=0)
