"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/x265_amd.h declares; argument errors come back as status codes, never aborts."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="x265_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(x265amd_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for need in ("x265amd_init", "x265amd_sad_blocks", "x265amd_me_fullsearch", "x265amd_tq_forward",
                 "x265amd_tq_inverse", "x265amd_tq_roundtrip", "x265amd_quant_plane", "x265amd_dequant_plane",
                 "x265amd_strerror"):
        assert need in syms


def test_library_exports_every_declared_symbol():
    from x265_amd import api
    assert os.path.exists(api.LIB_PATH), "build first: make -C x265_amd/csrc (or __graft_entry__.build())"
    lib = C.CDLL(api.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/x265_amd.h but not exported"
    assert sorted(api.SIGNATURES) == declared_symbols(), "python binding table out of sync with the header"


def test_library_exports_every_dropin_symbol():
    """include/x265_amd_dropin.h: one entry per slot of xSadN[] / xDctN[] / xInvDctN[] and per free function of
    the reference's pixel layer (inc/x265.h:290-298)."""
    from x265_amd import api
    syms = declared_symbols("x265_amd_dropin.h")
    assert len(syms) == 3 + 5 + 10 + 4
    lib = C.CDLL(api.LIB_PATH)
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/x265_amd_dropin.h but not exported"
    assert sorted(api.DROPIN_SIGNATURES) == syms
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["x265_amd.h", "x265_amd_debug.h", "x265_amd_dropin.h"]


def test_every_exported_symbol_is_declared():
    """The converse: nothing named x265amd_* leaves the library without a declaration in one of the three
    headers -- the ABI (x265_amd.h), the per-block drop-ins (x265_amd_dropin.h) or the instrumentation hooks
    (x265_amd_debug.h)."""
    import subprocess
    from x265_amd import api
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("x265amd_")})
    declared = set(declared_symbols()) | set(declared_symbols("x265_amd_dropin.h")) | set(declared_symbols("x265_amd_debug.h"))
    assert [s_ for s_ in exported if s_ not in declared] == []
    assert sorted(api.DEBUG_SIGNATURES) == declared_symbols("x265_amd_debug.h")


def test_dropins_report_errors_without_gpu():
    """No device / not initialised: nothing is written, the status says why, nothing aborts."""
    import numpy as np
    from x265_amd import api
    lib = api.load_library()
    a = np.zeros((8, 8), np.uint8)
    r = lib.x265amd_xSad8xN(8, a.ctypes.data, 8, a.ctypes.data, 8)
    if lib.x265amd_dropin_status() != 0:       # on the GPU box after init this call simply works
        assert r == 0xFFFFFFFF
    r = lib.x265amd_xSad8xN(65, a.ctypes.data, 8, a.ctypes.data, 8)   # more rows than a CU has
    assert r == 0xFFFFFFFF and lib.x265amd_dropin_status() == -1


def test_host_side_helpers_without_gpu():
    from x265_amd import api
    lib = api.load_library()
    assert lib.x265amd_version() == 100
    # g_aucChromaScale, src/table.cpp:166-171
    want = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
    assert [api.chroma_qp(q) for q in range(52)] == want
    with pytest.raises(api.X265AmdError):
        api.chroma_qp(52)
    assert b"argument" in lib.x265amd_strerror(-1)
    assert b"gfx950" in lib.x265amd_strerror(-3)


def test_desc_struct_layout_matches_header():
    """sizeof / field order of the ctypes mirror of x265amd_tq_desc (natural C alignment)."""
    from x265_amd import api
    names = [f[0] for f in api.TqDesc._fields_]
    text = open(os.path.join(ROOT, "include", "x265_amd.h")).read()
    body = text[text.index("typedef struct x265amd_tq_desc {"):text.index("} x265amd_tq_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decl = re.findall(r"\b(?:const\s+)?[a-z0-9_]+(?:\s*\*)?\s*\**\s*([a-z_0-9]+(?:\s*,\s*[a-z_0-9]+)*)\s*;", body)
    flat = [n.strip() for d in decl for n in d.split(",")]
    assert flat == names
    assert C.sizeof(api.TqDesc) == 192   # sizeof(x265amd_tq_desc) from gcc
