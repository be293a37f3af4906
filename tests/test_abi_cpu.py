"""CPU checks of the C ABI: the library loads, exports every symbol include/spx_hip.h declares, and the
host-side entry points (plan builder, size queries, argument validation) behave.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from scaleprotoseg_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        from scaleprotoseg_amd.build import build

        build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spx_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spx_[a-zA-Z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from scaleprotoseg_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in spx_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.spx_version() == _lib.ABI_VERSION


def test_plan_struct_matches_header(lib):
    from scaleprotoseg_amd import _lib

    text = open(os.path.join(ROOT, "include", "spx_hip.h")).read()
    assert int(re.search(r"#define SPX_MAX_PANELS (\d+)", text).group(1)) == _lib.SPX_MAX_PANELS
    assert C.sizeof(_lib.SpxPlan) == 4 * (8 + 3 * _lib.SPX_MAX_PANELS)


@pytest.mark.parametrize(
    "P,K,S,Cs,npanels,npb,ncb",
    [
        (228, 19, 4, 64, 4, 2, 1),     # scaleproto_cityscapes.gin
        (190, 19, 1, 256, 1, 6, 1),    # north-star bank
        (210, 21, 1, 64, 2, 4, 1),     # baseline_pascal.gin: 2 panels of 105
        (1800, 150, 4, 64, 12, 6, 5),  # scaleproto_ade.gin: 3 panels per scale
        (16, 3, 2, 16, 2, 2, 1),
    ],
)
def test_make_plan(lib, P, K, S, Cs, npanels, npb, ncb):
    from scaleprotoseg_amd import _lib

    per = P // S
    plan = _lib.make_plan(P, K, S, Cs, [s * per for s in range(S)], [(s + 1) * per for s in range(S)])
    assert (plan.npanels, plan.npb, plan.ncb, plan.kc) == (npanels, npb, ncb, 32)
    covered = []
    for q in range(plan.npanels):
        assert 0 < plan.panel_np[q] <= 32 * plan.npb
        assert plan.panel_ch0[q] % Cs == 0
        covered += list(range(plan.panel_p0[q], plan.panel_p0[q] + plan.panel_np[q]))
    assert covered == list(range(P))
    pp = C.byref(plan)
    assert lib.spx_packed_bank_bytes(pp) == npanels * npb * 32 * (-(-Cs // 32) * 32) * 2
    assert lib.spx_packed_p2_bytes(pp) == npanels * npb * 32 * 4
    assert lib.spx_packed_head_bytes(pp) == ncb * npanels * npb * 4096
    blobs = npanels * 2 * (-(-129 * 257 // 128)) * 4 * npb * 2 * 1024
    tiles = 2 * (-(-129 * 257 // 128))
    # + the per-(lane, block) exponents of the activation blob + the format word + the G blob's exponent per (panel, tile)
    assert lib.spx_bwd_scratch_bytes(pp, 2, 129 * 257) == blobs + blobs // 8 + 16 + 4 * npanels * tiles + 16
    assert lib.spx_bank_bwd_workspace_bytes(pp, 1, 65 * 65) > 0


def test_plan_errors(lib):
    from scaleprotoseg_amd import _lib

    with pytest.raises(_lib.SpxError, match="multiple of 16"):
        _lib.make_plan(20, 2, 1, 24, [0], [20])
    # P % S != 0: the reference's F.linear rejects the narrower distance map (model_multiscale.py:244)
    with pytest.raises(_lib.SpxError, match="covers 189"):
        _lib.make_plan(190, 19, 3, 64, [0, 63, 126], [63, 126, 189])
    with pytest.raises(_lib.SpxError, match="contiguous"):
        _lib.make_plan(20, 2, 2, 64, [0, 12], [10, 20])
    with pytest.raises(_lib.SpxError, match="num_classes"):
        _lib.make_plan(400, 200, 1, 64, [0], [400])


def test_null_and_range_validation(lib):
    from scaleprotoseg_amd import _lib

    plan = _lib.make_plan(190, 19, 1, 256, [0], [190])
    pp = C.byref(plan)
    assert lib.spx_dist_fwd(pp, None, 0, 1, 64, None, None, None, None, None, None, 1e-4, 0, None) != 0
    assert b"NULL" in lib.spx_last_error()
    assert lib.spx_dist_fwd(pp, 16, 7, 1, 64, 16, 16, None, None, None, None, 1e-4, 0, None) != 0
    assert b"x_dtype" in lib.spx_last_error()
    assert lib.spx_dist_fwd(pp, 16, 0, 0, 64, 16, 16, None, None, None, None, 1e-4, 0, None) != 0
    assert b"empty" in lib.spx_last_error()
    assert lib.spx_push_argmin(None, None, None, 1, 1, 1, 1, 0, 1e10, None, None, None, None) != 0
    assert lib.spx_argmin_images(None, 1, 1, None, None) != 0


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    import torch

    import scaleprotoseg_amd as spx

    lay = spx.BankLayout(10, 1, 1, 64, ((0, 10),))
    with pytest.raises(spx.SpxError, match="no CPU fallback"):
        spx.proto_head_forward(torch.zeros(1, 64, 4, 4), torch.zeros(10, 64, 1, 1), None, lay)
    with pytest.raises(spx.SpxError):
        spx.push_masked_argmin(torch.zeros(1, 10, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long), torch.zeros(10, 2))


def test_product_path_does_not_import_the_oracle():
    import subprocess
    import sys

    code = "import sys; import scaleprotoseg_amd; assert not any(m.startswith('oracle') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scaleprotoseg_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f
