"""CPU: the C-ABI library loads and exports every symbol declared in include/cogview_hip.h, and the ctypes
signature table mirrors the header (no compute calls here -- there is no GPU on this box)."""
import os
import re

from cogview_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "cogview_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cogv_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_the_same_functions():
    hdr = _header_functions()
    assert hdr, "no functions parsed from the header"
    assert sorted(_lib.SIGNATURES) == hdr


def test_library_loads_and_exports_every_symbol():
    lib = _lib.lib()
    for name in _header_functions():
        assert hasattr(lib, name), name
    assert lib.cogv_version() == 1
    assert lib.cogv_arch() == b"gfx950"
    # pure host helpers are callable without a device
    assert lib.cogv_gemm_pick_splitk(32640, 3072, 1024) == 1
    assert lib.cogv_gemm_pick_splitk(1024, 1024, 17408) > 1
    assert lib.cogv_ln_bwd_workspace_bytes(1088, 1024) == lib.cogv_ln_bwd_num_blocks(1088) * 3 * 1024 * 4


def test_struct_layouts_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "cogview_hip.h")).read()
    for cname, cls in (("cogv_gemm_desc", _lib.GemmDesc), ("cogv_attn_desc", _lib.AttnDesc), ("cogv_adam_desc", _lib.AdamDesc), ("cogv_attn_decode_desc", _lib.AttnDecodeDesc), ("cogv_ln_prologue", _lib.LnPrologue),
                       ("cogv_conv_desc", _lib.ConvDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            for i, part in enumerate(parts):
                tok = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part.split("[")[0])
                names.append(tok[-1])
        assert names == [f[0] for f in cls._fields_], cname


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from cogview_amd import ops
    with pytest.raises(_lib.CogviewHipError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16))
    # nothing under cogview_amd/ may import the oracle
    for dp, _, fs in os.walk(os.path.join(ROOT, "cogview_amd")):
        for f in fs:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle/", ""), os.path.join(dp, f)
