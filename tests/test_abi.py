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


def test_build_flags_reads_of_in_flight_asm_loads():
    """cogview_amd/csrc/build.py::scan_asm_hazards on hand-written assembly: a copy of an asm load's destination in front
    of its s_waitcnt is reported, the same copy behind the wait is not, counted waits retire the oldest loads only, and
    LDS reads (lgkmcnt) are tracked separately from global loads (vmcnt)."""
    from cogview_amd.csrc.build import scan_asm_hazards

    def asm(*body):
        return ["_ZN4testE:"] + ["\t" + b for b in body] + [".Lfunc_end0:"]
    load = lambda dst, addr: [";;#ASMSTART", f"global_load_dwordx4 {dst}, {addr}, off", ";;#ASMEND"]
    wait = lambda n: [";;#ASMSTART", f"s_waitcnt vmcnt({n})", ";;#ASMEND"]
    bad = asm(*load("v[2:5]", "v[20:21]"), "v_mov_b64_e32 v[30:31], v[4:5]", *wait(0))
    assert [h[2] for h in scan_asm_hazards(bad)] == ["v_mov_b64_e32 v[30:31], v[4:5]"]
    good = asm(*load("v[2:5]", "v[20:21]"), *wait(0), "v_mov_b64_e32 v[30:31], v[4:5]")
    assert scan_asm_hazards(good) == []
    # two loads, vmcnt(1): the older one is retired, the younger one is still in flight
    two = asm(*load("v[2:5]", "v[20:21]"), *load("v[6:9]", "v[22:23]"), *wait(1), "v_add_f32_e32 v40, v2, v3", "v_add_f32_e32 v41, v6, v7")
    assert [h[2] for h in scan_asm_hazards(two)] == ["v_add_f32_e32 v41, v6, v7"]
    # an LDS read issued through asm is retired by lgkmcnt, not by vmcnt
    lds = asm(";;#ASMSTART", "ds_read_b64_tr_b16 v[10:11], v50", ";;#ASMEND", "s_waitcnt vmcnt(0)", "v_mov_b32_e32 v60, v10",
              "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v61, v11")
    assert [h[2] for h in scan_asm_hazards(lds)] == ["v_mov_b32_e32 v60, v10"]
    # using an in-flight destination as a STORE source is a read too; writing an unrelated register is fine
    st = asm(*load("v[2:5]", "v[20:21]"), "ds_write_b128 v70, v[2:5]", "v_mov_b32_e32 v80, v81", *wait(0))
    assert [h[2] for h in scan_asm_hazards(st)] == ["ds_write_b128 v70, v[2:5]"]
    # a RETURNING atomic issued through asm (the generation-4 GEMM's hand-issued work-queue grab, -DCOGV_W4_ASYNC_GRAB) is a
    # load of its destination register; without sc0 nothing comes back and nothing is tracked
    grab = [";;#ASMSTART", "global_atomic_add v135, v168, v135, s[68:69] sc0", ";;#ASMEND"]
    early = asm(*grab, "v_lshl_or_b32 v0, v135, 3, s81", *wait(0))
    assert [h[2] for h in scan_asm_hazards(early)] == ["v_lshl_or_b32 v0, v135, 3, s81"]
    assert scan_asm_hazards(asm(*grab, *wait(0), "v_lshl_or_b32 v0, v135, 3, s81")) == []
    noret = asm(";;#ASMSTART", "global_atomic_add v168, v135, s[68:69]", ";;#ASMEND", "v_mov_b32_e32 v1, v135", *wait(0))
    assert scan_asm_hazards(noret) == []
