"""The C-ABI library loads and exports every symbol include/smcnuts_hip.h
declares (no compute calls: runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "smcnuts_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef SMCN_LEGACY_ABI.*?#endif", "", txt, flags=re.S)     # not exported by the product build
    return sorted(set(re.findall(r"\b(smcn_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from smcnuts_amd import _capi, build
    build.build()
    lib = _capi.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    for n in _capi.SIGNATURES:
        assert n in names, f"{n} bound in _capi but not declared in include/smcnuts_hip.h"
    assert lib.smcn_version() >= 1


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle."""
    pkg = os.path.join(ROOT, "smcnuts_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle.blocked_cumsum", ""), f"{f} mentions the oracle"


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smcnuts_amd import ArmaModel, SMCSampler
    from smcnuts_amd._capi import SmcnError
    with pytest.raises(SmcnError):
        SMCSampler(K=1, N=64, target=ArmaModel(), step_size=0.01)
