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


def test_split_k_decision_is_host_logic_and_covers_the_small_m_problems():
    """anip_gemm_workspace_bytes is pure host logic (no launch): which problems the library splits over K.
    Round 4: the ReferenceNet's deep levels (M = 128 / 512 rows for the CFG pair of one reference frame, K = 11 520 ...
    23 040 in its 3x3 convolutions) are split; short-K and chip-filling problems are not; the workspace is
    [slices][M][N] fp32."""
    import ctypes as C
    from aniportrait_amd import _lib
    lib = _lib.load()

    def ws(M, N, K, conv=None, act=0, batch=1, trans=0):
        p = _lib.GemmParams()
        p.A, p.W, p.out = 256, 256, 256            # never dereferenced by the query; 16-B aligned
        p.M, p.N, p.K, p.lda, p.ldw, p.ldo = M, N, K, K, K, N
        p.alpha, p.act, p.batch, p.trans_out = 1.0, act, batch, trans
        if conv:                                    # (Nimg, H, Cin): 3x3, stride 1, pad 1, channel-block-major K order
            p.conv, p.Nimg, p.Hin, p.Win, p.Cin = 2, conv[0], conv[1], conv[1], conv[2]
            p.Hout = p.Wout = conv[1]
            p.stride, p.pad = 1, 1
        return lib.anip_gemm_workspace_bytes(C.byref(p))

    def slices(nbytes, M, N):
        assert nbytes % (M * N * 4) == 0
        return nbytes // (M * N * 4)

    # ReferenceNet 8x8 / 16x16 convolutions (two frames): one / four 128-row tiles x 8 column tiles of 160 -> many slices
    assert 16 <= slices(ws(128, 1280, 11520, conv=(2, 8, 1280)), 128, 1280) <= 32
    assert 4 <= slices(ws(512, 1280, 23040, conv=(2, 16, 2560)), 512, 1280) <= 32
    assert 2 <= slices(ws(512, 1280, 5120), 512, 1280) <= 32          # its ff-out Linear
    # short K, GEGLU, transposed or batched problems and tiny M are never split
    assert ws(128, 1280, 320) == 0 and ws(512, 1280, 640) == 0
    assert ws(128, 10240, 1280, act=1) == 0 and ws(128, 1280, 1280, trans=1) == 0 and ws(128, 1280, 5120, batch=2) == 0
    assert ws(32, 1280, 5120) == 0
    # the denoising UNet: 8x8 convolutions (M = 2048, K = 11 520) on up to 8 slices of the wide tiles; chip-filling shapes unsplit
    assert 2 <= slices(ws(2048, 1280, 11520, conv=(32, 8, 1280)), 2048, 1280) <= 8
    assert ws(8192, 1280, 1280) == 0 and ws(131072, 320, 320) == 0 and ws(32768, 640, 2560) == 0


def test_ref_attention_wrapper_validates_the_frame_modulus():
    """frame_mod travels in bits 16.. of a C int — the wrapper refuses values that would wrap, a frame count that is
    not a multiple of it, and operands that do not hold `frame_mod` frames (checked before any pointer reaches the library)"""
    import pytest
    import torch

    from aniportrait_amd import hipops
    T, heads, d = 4, 2, 8
    q = torch.zeros((2 * T, heads * d), dtype=torch.float16)
    k = torch.zeros_like(q)
    vt = torch.zeros((heads * d, 2 * T), dtype=torch.float16)
    for kw in (dict(n_frames=4, frame_mod=40000), dict(n_frames=5, frame_mod=2), dict(n_frames=6, frame_mod=3)):
        with pytest.raises(ValueError):
            hipops.ref_attention(q, heads * d, k, heads * d, vt, 2 * T, kw["n_frames"], T, heads, d, frame_mod=kw["frame_mod"])
