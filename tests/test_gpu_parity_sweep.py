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


ATOL_SCORE = 7e-6       # tests/test_gpu_alike.py's bound (r05 measured 5.0e-6 on 256 pairs, 5.6e-6 on 1 024): a regression to r03's 8e-6 fails
ATOL_DESC = 1e-4        # north_star's bound, at the keypoints


@pytest.mark.timeout(900)
def test_eight_full_size_pairs_give_the_cpu_chains_keypoints_and_matches():
    import parity_sweep
    r = parity_sweep.sweep(8, first=200)
    assert r["max_abs_score_diff"] <= ATOL_SCORE, r
    assert r["max_abs_descriptor_diff"] <= ATOL_DESC, r
    assert r["images_with_identical_keypoint_sets"] == r["images"] == 16, r
    assert r["pairs_with_identical_match_sets"] == r["pairs"] == 8, r
    assert r["keypoints"] == 16000 and r["matches"] > 4000, r


@pytest.mark.timeout(900)
def test_sixty_four_pairs_bound_the_rate_of_differing_keypoints_and_matches():
    """at scale the path is NOT set-identical to the fp32 CPU chain -- a top-K cut between two scores closer than
    the arithmetic's error goes the other way (profiles/r05_parity_sweep_1024.json: 5 of 2 048 000 keypoints, 2 of 744 712 matches).
    Exact equality on a few pairs does not notice that rate growing; this leg does: 64 pairs = 128 000 keypoints, at most ONE per
    100 000 may differ (so at most 2 here; r05's rate predicts 0.3), likewise the matches, and every difference must be such a cut --
    the image keeps 1000 keypoints and loses / gains the same number."""
    import parity_sweep
    r = parity_sweep.sweep(64, first=1000)
    assert r["max_abs_score_diff"] <= ATOL_SCORE, r
    assert r["max_abs_descriptor_diff"] <= ATOL_DESC, r
    assert r["keypoints"] == 128000 and r["matches"] > 32000, r
    assert r["keypoints_differing"] * 100000 <= 2 * r["keypoints"], r         # <= 1 per 100 000, rounded up to whole keypoints (2 of 128 000)
    assert r["matches_differing"] * 100000 <= 4 * r["matches"], r             # a differing keypoint takes at most its own match with it, on either side
    assert r["images"] - r["images_with_identical_keypoint_sets"] <= r["keypoints_differing"], r
    assert r["pairs"] - r["pairs_with_identical_match_sets"] <= r["matches_differing"] + r["keypoints_differing"], r
