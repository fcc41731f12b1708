"""The C-ABI shared library builds, loads and exports every symbol include/aniportrait_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "aniportrait_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(anip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    from aniportrait_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.anip_version() == _lib.ABI_VERSION
    assert lib.anip_profile_kernel_name(0).decode().startswith("gemm_kernel")


def test_gemm_params_struct_matches_header():
    """field order of the ctypes mirror == field order of struct anip_gemm_params"""
    from aniportrait_amd import _lib
    src = open(os.path.join(REPO, "include", "aniportrait_hip.h")).read()
    body = src[src.index("typedef struct anip_gemm_params {"):src.index("} anip_gemm_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(","):
            fields.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert fields == [f[0] for f in _lib.GemmParams._fields_]


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(REPO, "aniportrait_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
    for f in os.listdir(os.path.join(REPO, "src", "models")) + os.listdir(os.path.join(REPO, "src", "pipelines")):
        pass


def test_hot_path_fails_loudly_without_gpu():
    import pytest
    import torch
    from aniportrait_amd import configs as C
    from aniportrait_amd._lib import HipLibraryError
    from aniportrait_amd.unet import UNet3DConditionModel
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = UNet3DConditionModel(**C.unet3d_kwargs(True))
    with pytest.raises(HipLibraryError):
        m(torch.zeros(2, 4, 2, 8, 8), 10, torch.zeros(2, 1, 64))
    with pytest.raises(HipLibraryError):
        m.packed()
