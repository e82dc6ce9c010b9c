"""The whole path at the configured size against the fp32 CPU chain, as a test: N synthetic
640 x 480 pairs through the GPU path (split-f16 MFMA ALIKE-t -> NMS / top-K -> sampling -> float64 match) and through oracle/
(torch-fp32 ALIKE-t restatement + C detection / sampling / match).  The stages are bit-exact on equal inputs; the net's score map
differs from the CPU's by a few 1e-6, which may permute rows of near-equal score, so keypoints and matches are compared as PIXEL
sets: identical keypoint pixel sets in every image, identical match pixel pairs in every pair, and the two differences bounded.
scripts/parity_sweep.py runs the same comparison over 48 pairs and writes profiles/r03_parity_sweep.json."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.timeout(900)
def test_eight_full_size_pairs_give_the_cpu_chains_keypoints_and_matches():
    import parity_sweep
    r = parity_sweep.sweep(8, first=200)
    assert r["max_abs_score_diff"] <= 1e-5, r               # tests/test_gpu_alike.py's bound, over 16 full-size images
    assert r["max_abs_descriptor_diff"] <= 1e-4, r          # north_star's bound, at the keypoints
    assert r["images_with_identical_keypoint_sets"] == r["images"] == 16, r
    assert r["pairs_with_identical_match_sets"] == r["pairs"] == 8, r
    assert r["keypoints"] == 16000 and r["matches"] > 4000, r
