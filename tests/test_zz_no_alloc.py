"""x265amd_me_fullsearch_sea_static: the elimination search on the library's static scratch array, for
callers that cannot allocate (HIP graph capture, tight serving loops).  It has no counterpart in the
reference; the check is against x265amd_me_fullsearch_sea (caller workspace) on the same inputs, bit for bit,
on two calls in a row of which the second is smaller -- the scratch keeps the first call's contents -- and
that the call neither allocates nor frees device memory.

Collected last on purpose (pytest -x at round end): a failure here hides no other test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _both(gpu, cur, ref, R, ctu=64):
    """(static-scratch result, workspace result, device bytes in use before / after the static call)"""
    f = cur.shape[0] if cur.dim() == 3 else 1
    nrefs = ref.shape[1] if ref.dim() == 4 else 1
    h, w = cur.shape[-2:]
    nctu = (h // ctu) * (w // ctu)
    want = gpu.me_fullsearch(cur, ref, ctu, R, method="sea").clone()
    out = torch.full((f, nrefs, nctu, 2), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    got = gpu.me_fullsearch(cur, ref, ctu, R, out=out, method="sea_static")
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    return got, want, free0, free1


def test_sea_static_matches_workspace_entry_point_twice(gpu):
    from x265_amd import synth, api
    assert api.me_static_workspace_bytes() == 40 << 20
    # call 1: a 1080p picture at +-64 (20.4 MB of scratch)
    big = synth.make_sequence(1920, 1088, 2, search_range=64, seed=12345)
    got, want, f0, f1 = _both(gpu, dev(big[1]), dev(big[0]), 64)
    assert torch.equal(got, want)
    assert f0 == f1, "the static entry point changed the amount of free device memory"
    # call 2, right after it and SMALLER: two frames x two references of 320x192 at +-33 -- other plane count,
    # other pitch of the bounds rows, other queue lengths: whatever call 1 left in the scratch must not matter
    seq = synth.make_sequence(320, 192, 4, search_range=33, seed=9)
    fr = dev(seq)
    cur = fr[2:]
    refs = torch.stack([fr[1:3], fr[0:2]], dim=1).contiguous()     # [2][2][H][W]
    got2, want2, f0, f1 = _both(gpu, cur, refs, 33)
    assert torch.equal(got2, want2)
    assert f0 == f1
    # a 32x32 block grid on the same scratch (other kernels' instantiations, four times the blocks)
    got4, want4, f0, f1 = _both(gpu, cur, refs, 33, ctu=32)
    assert torch.equal(got4, want4)
    assert f0 == f1
    # and the first job again: the small job's leftovers do not matter either
    got3, want3, _, _ = _both(gpu, dev(big[1]), dev(big[0]), 64)
    assert torch.equal(got3, want) and torch.equal(want3, want)


def test_sea_static_refuses_jobs_beyond_the_scratch(gpu):
    from x265_amd import api
    a = torch.zeros((3, 1088, 1920), dtype=torch.uint8, device="cuda")     # three 1080p planes: 53 MB needed
    assert api.me_workspace_bytes(1920, 1088, 64, 3, 1) > api.me_static_workspace_bytes()
    out = torch.zeros((3, 1, 510, 2), dtype=torch.int32, device="cuda")
    with pytest.raises(gpu.X265AmdError) as e:
        gpu.me_fullsearch(a, a, 64, 64, out=out, method="sea_static")
    assert e.value.status == -6 and "static scratch" in str(e.value)
    with pytest.raises(gpu.X265AmdError):
        gpu.me_fullsearch(a[:1, :100], a[:1, :100], 64, 16, out=out, method="sea_static")   # geometry errors as in _sea
