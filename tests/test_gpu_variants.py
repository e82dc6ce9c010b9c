"""Every kernel variant that ships is exercised.  The library reads its two remaining environment knobs
once per process, so each knob set runs the golden-vector tests in a fresh CHILD python (subprocess.run: a child, never an exec
of this GPU-initialised process):

  KPB_FP32_MATRIX=1      the strict-fp32 kernels everywhere -- alike_block1 / conv3x3_k on the fp32 vector ALUs, conv_mfma and
                         alike_head_hyb on v_mfma_f32_32x32x2_f32, lg_flash in fp32 -- instead of the split-f16 matrix forms.
                         The same goldens, the same tolerances: this is the companion figure's code path (bench.py variant_fp32).
  KPB_MATCH_PREFILTER=0  the exact float64 tile kernel on every pair (the path a pair with out-of-range or non-finite
                         descriptors takes under the default, and -- r04 -- every call with fewer than 8 pairs: for them the
                         prefilter's five dependent launches are latency, not throughput).
  KPB_MATCH_PREFILTER=2  the MFMA prefilter + exact refinement for ANY number of pairs: the single-pair match goldens and the fuzz
                         then run on the path the batched pipelines take from 8 pairs up.

Experiment knobs of r05 (KPB_PRESPLIT=0 / 1: SuperPoint's conv1a as its own kernel in front of conv1b -- superseded by conv1b generating
those channels while it stages, measured, removed in r06 with their kernels) and of r02 (KPB_HEAD_MAP / PIPE / WPS / PF, KPB_*_MT1, KPB_GEMM_*, KPB_BLOCK*_H16, KPB_CONV_H16) lost their
non-default branches: the measured choice is the code."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLDEN_TESTS = [
    "tests/test_gpu_alike.py",
    "tests/test_gpu_match.py",
    "tests/test_gpu_superpoint.py::test_superpoint_small_against_reference_golden",
    "tests/test_gpu_xfeat.py",
    "tests/test_gpu_disk.py",
    "tests/test_gpu_lightglue.py::test_lightglue_against_reference_golden",
    "tests/test_gpu_pipeline.py",
]


def _child(env_extra, tests):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + tests
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = "\n".join(p.stdout.splitlines()[-25:])
    assert p.returncode == 0, "child pytest with %s failed:\n%s" % (env_extra, tail)
    assert " passed" in tail and "skipped" not in tail.split("passed")[-1], tail
    return tail


@pytest.mark.timeout(1800)
def test_strict_fp32_kernels_pass_the_goldens():
    _child({"KPB_FP32_MATRIX": "1"}, GOLDEN_TESTS)


@pytest.mark.timeout(900)
def test_exact_match_kernel_without_the_prefilter_passes_the_goldens():
    _child({"KPB_MATCH_PREFILTER": "0"}, ["tests/test_gpu_match.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_pipeline.py"])


@pytest.mark.timeout(900)
def test_prefilter_forced_on_single_pairs_passes_the_goldens():
    _child({"KPB_MATCH_PREFILTER": "2"}, ["tests/test_gpu_match.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_range.py", "tests/test_gpu_pipeline.py"])


def test_no_other_environment_knob_selects_a_kernel():
    """`kpb_env_int` / getenv may appear for exactly the knobs above, the NMS schedule knobs tests/test_gpu_detect.py drives, and
    KPB_LOG_ALLOC (a diagnostic: prints where each workspace landed, scripts/head_modes.py; selects nothing)."""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(ROOT, "keypoint_bench_amd", "csrc", "*")):
        names |= set(re.findall(r'(?:env_int|getenv)\("(KPB_[A-Z0-9_]+)"', open(f).read()))
    allowed = {"KPB_FP32_MATRIX", "KPB_MATCH_PREFILTER", "KPB_NMS_TILED", "KPB_NMS_PRUNE", "KPB_NMS_TAIL_ROUNDS", "KPB_LOG_ALLOC"}
    assert names <= allowed, names - allowed
