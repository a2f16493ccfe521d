"""The C++ host layer (host/x265amd_host.hpp + host/encode_p.cpp): a compiled host that sees only
include/x265_amd.h and libx265amd.so -- what the north_star calls "C++ host code calls hand-written HIP
kernels through a thin C-ABI layer".  CPU: it is built and starts without a GPU.  GPU: its closed-loop
P-picture output (vectors, levels, reconstruction MD5) equals the oracle's, picture by picture."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "x265_amd", "lib", "x265amd_encode_p")


def test_cpp_host_is_built_and_reports_errors():
    assert os.path.exists(EXE), "build first: make -C host (or __graft_entry__.build())"
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    # geometry the reference would exit() on (src/params.cpp:73-83) comes back as a message and status 1
    r = subprocess.run([EXE, "/dev/null", "100", "64", "2", "30", "16", "/tmp/x265amd_none"], capture_output=True, text=True)
    assert r.returncode == 1 and "multiples of the 64x64 CTU" in r.stderr


def test_cpp_host_header_needs_no_hip(tmp_path):
    """The host header compiles with plain g++ and only this repo's include/ on the path."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "x265amd_host.hpp"\nint main() { return sizeof(x265amd_host::GopEncoder) > 0 ? 0 : 1; }\n')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "host"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cpp_host_closed_loop_vs_oracle(gpu, oracle, tmp_path):
    from x265_amd import synth, MV_DTYPE
    W, H, R, QP, T = 256, 192, 24, 28, 4
    y, u, v = synth.make_sequence(W, H, T, search_range=R, seed=17, chroma=True)
    yuv = np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1)
    ysz, csz = W * H, (W // 2) * (H // 2)
    fsz = ysz + 2 * csz
    src = tmp_path / "in.yuv"
    yuv.tofile(src)
    prefix = str(tmp_path / "out")
    r = subprocess.run([EXE, str(src), str(W), str(H), str(T), str(QP), str(R), prefix], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("POC") == T - 1 and "every-candidate" not in r.stdout
    assert r.stdout.count("packed levels") == T - 1  # each picture's packed levels unpacked and compared with the dense ones
    nctu = (W // 64) * (H // 64)
    gmv = np.fromfile(prefix + ".mv", MV_DTYPE).reshape(T - 1, nctu)
    glvl = np.fromfile(prefix + ".lvl", np.int16).reshape(T - 1, fsz)
    grec = np.fromfile(prefix + ".rec", np.uint8).reshape(T - 1, fsz)
    qpc = oracle.chroma_qp(QP)
    ref = yuv[0].copy()
    for t in range(1, T):
        cur = yuv[t]
        cy, ry = cur[:ysz].reshape(H, W), ref[:ysz].reshape(H, W)
        omv = oracle.me_fullsearch(cy, ry, 64, R, nthreads=8)
        np.testing.assert_array_equal(gmv[t - 1], omv)
        ly, _ = oracle.tq_forward(cy, ry, 32, QP, 1, mv=omv, mv_ctu=64)
        rec = [oracle.tq_inverse(ly, ry, 32, QP, 1, mv=omv, mv_ctu=64).reshape(-1)]
        lev = [ly.reshape(-1)]
        cmv = omv.copy()
        cmv["mvx"] >>= 1
        cmv["mvy"] >>= 1
        for k in range(2):
            a, b = ysz + k * csz, ysz + (k + 1) * csz
            cc, rc = cur[a:b].reshape(H // 2, W // 2), ref[a:b].reshape(H // 2, W // 2)
            lc, _ = oracle.tq_forward(cc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32)
            rec.append(oracle.tq_inverse(lc, rc, 16, qpc, 1, mv=cmv, mv_ctu=32).reshape(-1))
            lev.append(lc.reshape(-1))
        ref = np.concatenate(rec)
        np.testing.assert_array_equal(glvl[t - 1], np.concatenate(lev))
        assert hashlib.md5(grec[t - 1].tobytes()).hexdigest() == hashlib.md5(ref.tobytes()).hexdigest(), t


@pytest.mark.gpu
def test_cpp_host_bench_mode_streams_and_pinned_memory(gpu, tmp_path):
    """--bench: pinned host buffers (x265amd_malloc_host), one and two streams (x265amd_stream_create) -- the
    PCIe-inclusive loop of DESIGN.md section 5 on a small picture; both variants run and report."""
    import json
    from x265_amd import synth
    W, H, T = 256, 192, 3
    y, u, v = synth.make_sequence(W, H, T, search_range=16, seed=2, chroma=True)
    src = tmp_path / "in.yuv"
    np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1).tofile(src)
    r = subprocess.run([EXE, "--bench", str(src), str(W), str(H), str(T), "30", "16", "2"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    want = [("dense", 1, "c++"), ("dense", 2, "c++")]
    for lv in ("packed", "packed_direct"):
        want += [(lv, 1, "c++"), (lv, 2, "c++"), (lv, 2, "c++, a thread per stream")]
    assert [(l["levels"], l["streams"], l["host"]) for l in lines] == want
    assert lines[2]["bytes_down_per_step"] < lines[0]["bytes_down_per_step"]
    for l in lines:
        assert l["pcie_inclusive"] and l["nseg"] == 2 and l["mpixels_per_s_mean"] > 0
        assert l["bytes_up_per_step"] == 2 * W * H * 3 // 2


@pytest.mark.gpu
def test_two_host_threads_share_one_device(gpu):
    """host/two_hosts.cpp: two threads, each with its own stream, encoder and geometry (8-wave and 16-wave
    shapes of the exact kernel, different LDS sizes and workspaces), run concurrently; every picture's vectors,
    levels and reconstruction equal the same encoder's run alone.  The per-device state of the library
    ("per-device state") is what makes that safe."""
    exe = os.path.join(ROOT, "x265_amd", "lib", "x265amd_two_hosts")
    assert os.path.exists(exe), "build first: make -C host"
    r = subprocess.run([exe, "0", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "two_hosts ok" in r.stdout, r.stderr + r.stdout


@pytest.mark.gpu
def test_cpp_host_intra_encoder_reconstruction_md5(gpu, oracle, tmp_path):
    """host/encode_i.cpp (x265amd_host::IntraEncoder over x265amd_intra_encode_frame + x265amd_intra_write_picture,
    plain g++, the C-ABI only) on BASELINE configs[0]: the reconstruction file it writes has, frame by frame, the
    MD5s the reference encoder printed and the bitstream file has the MD5 of the reference's stream
    (tests/golden/intra_cfg1.json); modes and levels equal the oracle's restatement of the loop."""
    import hashlib
    import json
    from x265_amd import synth
    exe = os.path.join(ROOT, "x265_amd", "lib", "x265amd_encode_i")
    assert os.path.exists(exe), "build it with make -C host (or __graft_entry__.build())"
    with open(os.path.join(ROOT, "tests", "golden", "intra_cfg1.json")) as f:
        doc = json.load(f)
    w, h, cu, qp, nf = doc["width"], doc["height"], doc["cu"], doc["qp"], doc["frames"]
    yuv = synth.make_intra_probe(w, h, nf)
    src = tmp_path / "in.yuv"
    yuv.tofile(src)
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(src), str(w), str(h), str(nf), str(cu), str(qp), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    fsz = w * h * 3 // 2
    rec = np.fromfile(prefix + ".rec", np.uint8).reshape(nf, fsz)
    lvl = np.fromfile(prefix + ".lvl", np.int16).reshape(nf, fsz)
    modes = np.fromfile(prefix + ".modes", np.uint8).reshape(nf, 2, -1)
    for fr in range(nf):
        assert hashlib.md5(rec[fr].tobytes()).hexdigest() == doc["recon_md5"][fr], fr
    # ... and the bitstream file is the reference's, byte for byte (IntraEncoder::write_picture)
    bits = np.fromfile(prefix + ".bit", np.uint8)
    assert len(bits) == sum(doc["bytes_per_frame"])
    assert hashlib.md5(bits.tobytes()).hexdigest() == doc["bitstream_md5"]
    orec, my, mc, olv = oracle.intra_encode_frame(yuv[2], w, h, cu, qp, want_levels=True)
    np.testing.assert_array_equal(modes[2, 0], my)
    np.testing.assert_array_equal(modes[2, 1], mc)
    np.testing.assert_array_equal(lvl[2], olv)
    # batches of three frames through the GPU, four writer threads: the same four files
    prefix2 = str(tmp_path / "out_b3")
    r = subprocess.run([exe, str(src), str(w), str(h), str(nf), str(cu), str(qp), prefix2, "3", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for ext in (".rec", ".lvl", ".modes", ".bit"):
        with open(prefix + ext, "rb") as fa, open(prefix2 + ext, "rb") as fb:
            assert fa.read() == fb.read(), ext
    # bitstream only (dumps = 0: the reconstruction stays on the GPU): the same stream
    prefix3 = str(tmp_path / "out_bits")
    r = subprocess.run([exe, str(src), str(w), str(h), str(nf), str(cu), str(qp), prefix3, "2", "2", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(prefix3 + ".rec")
    with open(prefix + ".bit", "rb") as fa, open(prefix3 + ".bit", "rb") as fb:
        assert fa.read() == fb.read()
    # geometry the reference rejects (src/params.cpp:73-77) comes back as an error, not a crash
    r = subprocess.run([exe, str(src), "100", "64", "1", "16", "30", prefix], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "IntraEncoder" in r.stderr


@pytest.mark.gpu
def test_cpp_host_gop_with_intra_first_picture(gpu, oracle, tmp_path):
    """x265amd_encode_p ... <intra_cu>: the segment's first picture goes through GopEncoder::encode_i (the library's
    intra path); its reconstruction, levels and modes equal the oracle's intra frame, and the first P picture is
    searched against and predicted from that reconstruction."""
    from x265_amd import synth, MV_DTYPE
    W, H, R, QP, T, CU = 256, 128, 16, 31, 2, 16
    y, u, v = synth.make_sequence(W, H, T, search_range=R, seed=23, chroma=True)
    yuv = np.concatenate([y.reshape(T, -1), u.reshape(T, -1), v.reshape(T, -1)], axis=1)
    ysz = W * H
    src = tmp_path / "in.yuv"
    yuv.tofile(src)
    prefix = str(tmp_path / "out")
    r = subprocess.run([EXE, str(src), str(W), str(H), str(T), str(QP), str(R), prefix, str(CU)], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert "intra, CU 16" in r.stdout
    orec, my, mc, lv = oracle.intra_encode_frame(yuv[0], W, H, CU, QP, want_levels=True)
    np.testing.assert_array_equal(np.fromfile(prefix + ".i.rec", np.uint8), orec)
    np.testing.assert_array_equal(np.fromfile(prefix + ".i.lvl", np.int16), lv)
    np.testing.assert_array_equal(np.fromfile(prefix + ".i.modes", np.uint8), np.concatenate([my, mc]))
    gmv = np.fromfile(prefix + ".mv", MV_DTYPE)
    omv = oracle.me_fullsearch(yuv[1][:ysz].reshape(H, W), orec[:ysz].reshape(H, W), 64, R, nthreads=8)
    np.testing.assert_array_equal(gmv, omv.ravel())
