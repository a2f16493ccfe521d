"""The drop-ins of include/x265_amd_dropin.h really have the reference's signatures: a translation unit that
includes the reference's own header assigns them into xSadN[] / xDctN[] / xInvDctN[] and passes them where
xSubDct / xIDctAdd / xQuant / xDeQuant are expected, and g++ accepts it with -Werror (compile only).
Needs /root/reference (its header); skipped on the GPU box, where the reference does not exist."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/inc"

TU = r"""
#include "x265.h"
#include "x265_amd_dropin.h"

// the reference's declarations, as function-pointer types
typedef void   (*SubDctFn)(Int16 *, UInt8 *, UInt8 *, UInt, Int16 *, Int16 *, Int, Int, UInt);
typedef void   (*IDctAddFn)(UInt8 *, Int16 *, UInt8 *, UInt, Int16 *, Int16 *, Int, Int, UInt);
typedef UInt32 (*QuantFn)(Int16 *, Int16 *, UInt, UInt, Int, Int, X265_SliceType);
typedef void   (*DeQuantFn)(Int16 *, Int16 *, UInt, UInt, Int, Int, X265_SliceType);

void bind_gpu_pixel_layer()
{
    xSadN[0] = x265amd_xSad4xN;  xSadN[1] = x265amd_xSad8xN;  xSadN[2] = x265amd_xSad16xN;
    xSadN[3] = x265amd_xSad32xN; xSadN[4] = x265amd_xSad64xN;
    xDctN[0] = x265amd_xDST4;  xDctN[1] = x265amd_xDCT4;  xDctN[2] = x265amd_xDCT8;
    xDctN[3] = x265amd_xDCT16; xDctN[4] = x265amd_xDCT32;
    xInvDctN[0] = x265amd_xInvDST4;  xInvDctN[1] = x265amd_xInvDCT4;  xInvDctN[2] = x265amd_xInvDCT8;
    xInvDctN[3] = x265amd_xInvDCT16; xInvDctN[4] = x265amd_xInvDCT32;
    SubDctFn  a = x265amd_xSubDct;   SubDctFn  a0 = xSubDct;   (void)a; (void)a0;
    IDctAddFn b = x265amd_xIDctAdd;  IDctAddFn b0 = xIDctAdd;  (void)b; (void)b0;
    // X265_SliceType is an int-sized enum; the C-ABI spells it int
    QuantFn   c0 = xQuant;   UInt32 (*c)(Int16 *, Int16 *, UInt, UInt, Int, Int, int) = x265amd_xQuant;   (void)c; (void)c0;
    DeQuantFn d0 = xDeQuant; void   (*d)(Int16 *, Int16 *, UInt, UInt, Int, Int, int) = x265amd_xDeQuant; (void)d; (void)d0;
    static_assert(sizeof(X265_SliceType) == sizeof(int), "enum passed as int");
}
"""


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="needs the reference's header (not on the GPU box)")
@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_dropins_bind_into_the_reference_tables(tmp_path):
    src = tmp_path / "bind.cpp"
    src.write_text(TU)
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-variable",
                        "-I", REF_INC, "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
