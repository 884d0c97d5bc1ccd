"""The dispatch's A/B switches select fallback compositions of HIP kernels (plane GEMMs + attention core instead of the fused
deep-level launches, the 4-wave MLP, the LDS-staged patch kernel, ...).  Some of them are also the live path for shapes the fused
kernels do not cover, so they must stay correct: each set below runs the wide-block and small whole-model parity tests (fast tier,
against the oracle / the reference goldens) in a fresh child process with the switches forced.  The switches are read once per
process and only under SWF_DEBUG_SWITCHES=1 (csrc/swf_common.h: debug_env), hence one child per set — started with subprocess
before it touches the GPU, never an exec of this process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SETS = {
    "plane_mlp_and_proj_launch": {"SWF_NO_FUSED_MLP": "1", "SWF_NO_PROJFUSE": "1"},
    "no_qkvattn": {"SWF_NO_QKVATTN": "1"},
    "plane_gemms_and_attention_core": {"SWF_NO_QKVATTN": "1", "SWF_NO_ATTNPROJ": "1", "SWF_NO_DEEP_QKV": "1", "SWF_NO_DEEP_PROJ": "1"},
    "mlp4_and_window96_four_waves": {"SWF_MLP8": "0", "SWF_WIN96X8": "0"},
    "lds_staged_patch_kernels": {"SWF_NO_DEEP_PATCH": "1", "SWF_NO_PATCH_RR": "1"},
    "unfused_patch_and_generic_deep": {"SWF_NO_FUSED_PATCH": "1", "SWF_NO_DEEP": "1"},
}
_SELECT = ("(test_basic_block_wide_vs_oracle and fast) or (test_basic_block_single_stream_vs_oracle and fast) or "
           "(test_model_golden and fast and (win8_4stage_128 or win8_256_default or tiny))")


@pytest.mark.parametrize("name", sorted(_SETS))
def test_fallback_paths_stay_in_parity(name):
    env = dict(os.environ)
    env.update(_SETS[name])
    env["SWF_DEBUG_SWITCHES"] = "1"
    env["SWF_PARITY_LOG"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", _SELECT], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, f"switch set {name} {_SETS[name]}:\n{tail}"
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail
