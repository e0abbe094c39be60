"""The C-ABI library loads without a GPU and exports every symbol the headers declare; nothing here
launches compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ngsld_amd import capi

INCLUDE = os.path.join(capi.REPO_DIR, "include")


def _declared(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ngsld_[a-z_0-9]+)\s*\(", text)) - {"ngsld_sink_fn"})


def test_library_builds_and_loads():
    capi.build()
    assert os.path.exists(capi.LIB_PATH) and os.path.exists(capi.CLI_PATH)
    assert b"gfx950" in capi.lib().ngsld_version()


@pytest.mark.parametrize("header", ["ngsld.h", "ngsld_host.h"])
def test_every_declared_symbol_is_exported(header):
    names = _declared(header)
    assert len(names) >= 8
    L = capi.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in {header} but not exported"
        assert n in capi.SYMBOLS, f"{n} missing from the ctypes binding"


def test_binding_lists_nothing_extra():
    declared = set(_declared("ngsld.h")) | set(_declared("ngsld_host.h"))
    assert set(capi.SYMBOLS) == declared


def test_record_layouts_match_header():
    assert capi.REC_STD.itemsize == 32 and capi.REC_EXT.itemsize == 40      # sizes stated in ngsld.h
    assert capi.REC_EXT.fields["n_ind_data"][1] == 32 and capi.REC_EXT.fields["n_iter"][1] == 36
    assert C.sizeof(capi.Params) == 56 and capi.ITEM.itemsize == 32 and C.sizeof(capi.Batch) == 72


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful where no GPU is present")
def test_no_gpu_fails_loudly_no_fallback():
    with pytest.raises(capi.NgsldError) as e:
        capi.Engine(0)
    assert e.value.code == capi.ERR_DEVICE and "no CPU fallback" in e.value.msg


def test_product_never_touches_the_oracle():
    """The product tree must not reference oracle/ in any form (checker != product)."""
    pkg = os.path.join(capi.REPO_DIR, "ngsld_amd")
    for root, _, files in os.walk(pkg):
        if "build" in root.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.lower().replace("the cpu oracle", ""), f"{f} mentions the oracle"
    out = os.popen(f"ldd {capi.LIB_PATH}").read()
    assert "liborc" not in out and "ngsld_ref" not in out


def test_dispatch_table_is_the_committed_one():
    """Cohort size -> kernel family and shape (ngsld_describe_dispatch, no device needed) over 1..12,000, both settings of
    ignore_miss_data: tests/golden/dispatch_table.txt is that function's output -- a change of the dispatch is a change of this
    file, made on purpose (regenerate: python -c "from ngsld_amd import capi; open('tests/golden/dispatch_table.txt','w').write(capi.dispatch_table())")."""
    from ngsld_amd import capi
    want = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dispatch_table.txt")).read()
    assert capi.dispatch_table() == want
    # the BASELINE.json cohort sizes
    assert capi.describe_dispatch(100) == "group 1x7 lanes=16 np=112"
    assert capi.describe_dispatch(500) == "run 1x8 lanes=64 np=512"
    assert capi.describe_dispatch(1000) == "multi 2x8 lanes=64 np=1024"
    assert capi.describe_dispatch(2000) == "multi 4x8 lanes=64 np=2048"
    with pytest.raises(capi.NgsldError):
        capi.describe_dispatch(0)
