"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, exports every symbol include/difacto_hip.h declares, and refuses to run
without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from difacto_amd import build, capi as m
    build.build_hip()
    return m


def test_exports_every_declared_symbol(capi):
    hdr = open(os.path.join(ROOT, "include", "difacto_hip.h")).read()
    declared = set(re.findall(r"\b(dfh_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_host_helpers_bit_exact(capi, oracle):
    # ReverseBytes / EncodeFeaGrpID need no GPU (include/difacto/base.h:39-63)
    for x in (0, 1, 0x0123456789ABCDEF, 2 ** 64 - 1, 2 ** 64 - 2, 47149):
        assert capi.reverse_bytes(x) == oracle.reverse_bytes(x)
    assert capi.lib().dfh_encode_fea_grp_id(0xABCDEF, 7, 12) == oracle.encode_fea_grp_id(0xABCDEF, 7, 12)
    assert capi.row_stride(0) == 4 and capi.row_stride(5) == 12 and capi.row_stride(64) == 68


def test_param_defaults_match_reference(capi):
    # src/sgd/sgd_param.h:95-105
    p = capi.make_param(V_dim=3)
    assert (p.l1, p.l2, p.V_dim, p.V_threshold, p.seed) == (1.0, 0.0, 3, 10, 0)
    for name in ("V_l2", "lr", "V_lr", "V_init_scale"):
        assert abs(getattr(p, name) - 0.01) < 1e-9
    assert p.lr_beta == 1.0 and p.V_lr_beta == 1.0


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.DfhError) as e:
        capi.Context(0)
    assert e.value.code == 2  # DFH_ERR_HIP


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under difacto_amd/ may reference it"""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "difacto_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|difacto_oracle\.h|libdifacto_ref", src, re.M):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
