"""The hot path as the host drives it: for every P frame of a resident
sequence, exhaustive motion search against the previous source frame(s), then
residual -> transform -> quant -> dequant -> inverse -> reconstruction with the
integer-pel compensated prediction.  All frames of the shard go down in ONE
launch per kernel; every buffer is allocated once, here.

This is the loop src/encode.cpp:668-941 would run per CU for a P slice (which the
reference leaves as `assert(0)`, :581-584), turned inside out: frames x CTUs are
the grid instead of the loop.
"""
import torch

from . import api


class MeTqPipeline:
    def __init__(self, frames, ctu=64, search_range=64, nrefs=1, tu=32, qp=32, slice_type=api.SLICE_P,
                 method="sea"):
        """frames: uint8 tensor [F+nrefs][H][W] on the GPU (H, W multiples of ctu).  Frames
        nrefs..F+nrefs-1 are encoded; frame i uses frames i-1..i-nrefs as references."""
        if frames.dim() != 3 or frames.dtype != torch.uint8 or not frames.is_cuda:
            raise TypeError("frames: uint8 CUDA tensor [F][H][W]")
        self.frames = frames.contiguous()
        self.ctu, self.range, self.nrefs, self.tu, self.qp, self.slice_type = ctu, search_range, nrefs, tu, qp, slice_type
        n, h, w = self.frames.shape
        if n <= nrefs:
            raise ValueError("need at least nrefs + 1 frames")
        self.nframes = n - nrefs
        self.h, self.w = h, w
        self.nctu = (h // ctu) * (w // ctu)
        dev = self.frames.device
        self.cur = self.frames[nrefs:]
        # ref(f, r) = frame (nrefs + f) - 1 - r: a negative stride between references, which
        # torch cannot express as a view -- motion_search() hands the C-ABI the raw geometry.
        self.refs = self.frames[nrefs - 1:n - 1].unsqueeze(1) if nrefs == 1 else None
        self.mv = torch.empty((self.nframes, nrefs, self.nctu, 2), dtype=torch.int32, device=dev)
        self.level = torch.empty((self.nframes, h, w), dtype=torch.int16, device=dev)
        self.abs_sum = torch.empty((self.nframes, (h // tu) * (w // tu)), dtype=torch.int32, device=dev)
        self.rec = torch.empty((self.nframes, h, w), dtype=torch.uint8, device=dev)
        self.method = method
        self.workspace = None
        if method == "sea":
            self.workspace = torch.empty(max(api.me_workspace_bytes(w, h, search_range, self.nframes, nrefs), 16), dtype=torch.uint8,
                                         device=dev)
        self._mv0 = torch.empty((self.nframes, self.nctu, 2), dtype=torch.int32, device=dev) if nrefs > 1 else None

    def motion_search(self):
        fs = self.frames.stride(0)
        base = self.frames.data_ptr() + (self.nrefs - 1) * fs   # ref(f, r) = base + f*fs - r*fs
        args = (self.cur.data_ptr(), base, self.w, self.h, self.frames.stride(1), self.ctu, self.range, self.nframes,
                fs, fs, self.nrefs, -fs, self.mv.data_ptr())
        if self.method == "sea":
            st = api._lib().x265amd_me_fullsearch_sea(*args, self.workspace.data_ptr(), self.workspace.numel(),
                                                      api._stream())
            if st == -5:  # X265AMD_ERR_LDS: window + bounds do not fit (R = 128): exhaustive kernel
                self.method = "brute"
            else:
                api._check(st, "x265amd_me_fullsearch_sea")
                return self.mv
        api._check(api._lib().x265amd_me_fullsearch(*args, api._stream()), "x265amd_me_fullsearch")
        return self.mv

    def best_ref_mv(self):
        """Per CTU, the vector of reference 0 (single-reference prediction; multi-reference
        selection is the host's business and not part of the measured path)."""
        if self.nrefs > 1:
            self._mv0.copy_(self.mv[:, 0])
            return self._mv0
        return self.mv.view(self.nframes, self.nctu, 2)

    def transform_quant(self):
        mv = self.best_ref_mv()
        pred = self.frames[self.nrefs - 1:self.nrefs - 1 + self.nframes]
        return api.tq_roundtrip(self.cur, pred, self.tu, self.qp, self.slice_type, mv=mv, mv_ctu=self.ctu,
                                out=(self.level, self.abs_sum, self.rec))

    def step(self):
        self.motion_search()
        self.transform_quant()
