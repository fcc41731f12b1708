"""Measurement tooling that must keep up with the kernels (CPU; no GPU, no oracle).

Round 4's last PMC session lost its summary because a kernel added that afternoon (`conv3x3_c4_kernel`) had no family in
`tools/pmc_summarize.py` and the dispatch <-> wrapper-call pairing stopped at its first launch."""
import importlib.util
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_summarize_knows_every_kernel_of_the_library():
    from aniportrait_amd import _lib
    lib = _lib.LIB_PATH
    if not os.path.exists(lib):
        pytest.skip("library not built")
    out = subprocess.run(["nm", "-C", lib], capture_output=True, text=True, check=True).stdout
    kernels = sorted(set(re.findall(r"__device_stub__(\w+)", out)))
    assert len(kernels) > 40, kernels
    summ = _load(os.path.join(ROOT, "tools", "pmc_summarize.py"), "pmc_summarize")
    unknown = [k for k in kernels if summ.family(k) is None]
    assert not unknown, f"tools/pmc_summarize.py: no kernel family for {unknown}"


def test_gpu_session_script_parses():
    for script in ("tools/gpu_round4.sh",):
        subprocess.run(["bash", "-n", os.path.join(ROOT, script)], check=True)
