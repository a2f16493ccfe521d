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


class ClosedLoopGop:
    """Closed-loop P pictures: every picture is searched against, and predicted
    from, the RECONSTRUCTION of the picture before it, with the frame / reference bookkeeping of
    xEncEncode (src/encode.cpp:619-621, 965-970): a ring `refn[0..MAX_REF_NUM]`, the new reconstruction
    is written into refn[MAX_REF_NUM], then the ring rotates so that refn[0] is the newest.

    Time is serial inside a GOP segment, so the batch dimension is SEGMENTS: picture t of all `nseg`
    independent segments goes down in one launch per kernel (search, luma TQ, U TQ, V TQ).  Nothing is
    allocated and nothing synchronises after construction.

    Pictures are the reference's frame layout (src/encode.cpp:41-45): one W*H*3/2 block, Y then U then V,
    stride = width.  Luma: 64x64 CTU search, 32x32 TUs.  Chroma (4:2:0): the luma vector >> 1
    (integer-pel; the build's own definition -- the reference has no inter prediction), 32x32 blocks,
    16x16 TUs at g_aucChromaScale[qp] (src/encode.cpp:596).  The I picture that opens a segment goes through
    the intra path of the library (encode_i: x265amd_intra_encode_frames, the segments being its batch of
    independent frames); a caller that makes it elsewhere hands its reconstruction to set_reference().
    """
    MAX_REF_NUM = 1  # inc/config.h

    def __init__(self, width, height, nseg, ctu=64, search_range=64, qp=32, device="cuda", method="sea"):
        if width % 64 or height % 64 or ctu != 64:
            raise ValueError("width and height must be multiples of the 64x64 CTU")
        self.w, self.h, self.nseg, self.ctu, self.range, self.qp = width, height, nseg, ctu, search_range, qp
        self.qpc = api.chroma_qp(qp)
        self.ysz, self.csz = width * height, (width // 2) * (height // 2)
        self.fsz = self.ysz + 2 * self.csz
        self.nctu = (height // ctu) * (width // ctu)
        dev = torch.device(device)
        self.refn = [torch.zeros((nseg, self.fsz), dtype=torch.uint8, device=dev) for _ in range(self.MAX_REF_NUM + 1)]
        self.mv = torch.empty((nseg, 1, self.nctu, 2), dtype=torch.int32, device=dev)
        self.level = torch.empty((nseg, self.fsz), dtype=torch.int16, device=dev)   # same Y,U,V layout as the picture
        self.cbf_y = torch.empty((nseg, (height // 32) * (width // 32)), dtype=torch.int32, device=dev)
        self.cbf_c = [torch.empty((nseg, (height // 32) * (width // 32)), dtype=torch.int32, device=dev) for _ in range(2)]
        self.method = method
        self.workspace = torch.empty(max(api.me_workspace_bytes(width, height, search_range, nseg, 1), 16),
                                     dtype=torch.uint8, device=dev)
        self.poc = -1  # h->iPoc, src/encode.cpp:624
        self.intra = None  # encode_i's buffers, made on first use

    def encode_i(self, src_yuv, cu=32):
        """The I picture of every segment (the reference's own picture type: src/encode.cpp:668-941), CU = PU = TU =
        `cu`.  src_yuv: uint8 [S][W*H*3/2] on the GPU.  Leaves the levels in .level, the modes in .mode_y / .mode_c
        ([S][ncus]), the reconstruction in refn[0]."""
        if src_yuv.shape != (self.nseg, self.fsz) or src_yuv.dtype != torch.uint8 or not src_yuv.is_cuda:
            raise TypeError("src_yuv: uint8 CUDA tensor [nseg][W*H*3/2]")
        ncus = (self.h // cu) * (self.w // cu)
        if self.intra is None or self.intra["mode_y"].shape[1] != ncus:
            dev = src_yuv.device
            zi = lambda: torch.zeros((self.nseg, ncus), dtype=torch.int32, device=dev)
            self.intra = {"pred": torch.zeros((self.nseg, self.fsz), dtype=torch.uint8, device=dev),
                          "mode_y": torch.zeros((self.nseg, ncus), dtype=torch.uint8, device=dev),
                          "mode_c": torch.zeros((self.nseg, ncus), dtype=torch.uint8, device=dev),
                          "cost_y": zi(), "cost_c": zi(), "abs_sum_y": zi(), "abs_sum_u": zi(), "abs_sum_v": zi()}
        self.poc += 1
        out = dict(self.intra, rec=self.refn[self.MAX_REF_NUM], level=self.level)
        api.intra_encode_frames(src_yuv, self.w, self.h, cu, self.qp, out=out)
        self.mode_y, self.mode_c = self.intra["mode_y"], self.intra["mode_c"]
        self._rotate()
        return self.refn[0]

    def planes(self, buf):
        """[S][W*H*3/2] picture (or level) buffer -> (Y [S][H][W], U, V [S][H/2][W/2]) views."""
        s, w, h = buf.shape[0], self.w, self.h
        y = buf[:, :self.ysz].unflatten(1, (h, w))
        u = buf[:, self.ysz:self.ysz + self.csz].unflatten(1, (h // 2, w // 2))
        v = buf[:, self.ysz + self.csz:].unflatten(1, (h // 2, w // 2))
        return y, u, v

    def _rotate(self):  # src/encode.cpp:965-970
        tmp = self.refn[self.MAX_REF_NUM]
        for i in range(self.MAX_REF_NUM, 0, -1):
            self.refn[i] = self.refn[i - 1]
        self.refn[0] = tmp

    def set_reference(self, rec_yuv):
        """The reconstruction of the segment's I picture (uint8 [S][W*H*3/2] on the GPU), made by the host."""
        self.refn[self.MAX_REF_NUM].copy_(rec_yuv)
        self.poc += 1
        self._rotate()

    def encode_p(self, src_yuv):
        """One P picture per segment.  src_yuv: uint8 [S][W*H*3/2] on the GPU.  Leaves vectors in .mv,
        levels in .level, per-TU sum|level| in .cbf_y / .cbf_c, the reconstruction in refn[0]."""
        if src_yuv.shape != (self.nseg, self.fsz) or src_yuv.dtype != torch.uint8 or not src_yuv.is_cuda:
            raise TypeError("src_yuv: uint8 CUDA tensor [nseg][W*H*3/2]")
        self.poc += 1
        sy, su, sv = self.planes(src_yuv)
        py, pu, pv = self.planes(self.refn[0])
        ry, ru, rv = self.planes(self.refn[self.MAX_REF_NUM])
        ly, lu, lv = self.planes(self.level)
        lib = api._lib()
        args = (sy.data_ptr(), py.data_ptr(), self.w, self.h, self.w, self.ctu, self.range, self.nseg,
                self.fsz, self.fsz, 1, 0, self.mv.data_ptr())
        if self.method == "sea":
            st = lib.x265amd_me_fullsearch_sea(*args, self.workspace.data_ptr(), self.workspace.numel(), api._stream())
            if st == -5:  # X265AMD_ERR_LDS
                self.method = "brute"
            else:
                api._check(st, "x265amd_me_fullsearch_sea")
        if self.method != "sea":
            api._check(lib.x265amd_me_fullsearch(*args, api._stream()), "x265amd_me_fullsearch")
        mv = self.mv.view(self.nseg, self.nctu, 2)
        api.tq_roundtrip(sy, py, 32, self.qp, api.SLICE_P, mv=mv, mv_ctu=self.ctu, out=(ly, self.cbf_y, ry))
        for s_, p_, l_, r_, c_ in ((su, pu, lu, ru, self.cbf_c[0]), (sv, pv, lv, rv, self.cbf_c[1])):
            api.tq_roundtrip(s_, p_, 16, self.qpc, api.SLICE_P, mv=mv, mv_ctu=self.ctu // 2, mv_shift=1,
                             out=(l_, c_, r_))
        self._rotate()
        return self.refn[0]
