"""Frame / GOP-segment sharding across the GPUs of one node.

The path has no exchange step: with open-loop (source
frame) references every frame is independent, with closed-loop references
every GOP segment is.  So the data path needs NO collective: each rank owns a
contiguous run of frames (plus the reference frames that precede it), keeps
them in its own HBM, and returns motion vectors and levels to the host.
torch.distributed (RCCL on the GPU box, gloo in CPU tests) is used only for the
timing barrier and to gather small result digests.
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    first: int      # first frame this rank encodes
    count: int      # number of frames this rank encodes
    ref_first: int  # first frame it must hold (first - nrefs, clamped to 0)

    @property
    def frames_held(self):
        return self.first + self.count - self.ref_first


def shard_frames(nframes, world, rank, nrefs=1, segment=1):
    """Split frames 1..nframes-1 (frame 0 is only ever a reference) into `world`
    contiguous runs whose lengths are multiples of `segment` (GOP segment length)
    where possible.  Every frame is owned by exactly one rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    total = max(nframes - 1, 0)
    nseg = (total + segment - 1) // segment
    base, extra = divmod(nseg, world)
    seg_first = rank * base + min(rank, extra)
    seg_count = base + (1 if rank < extra else 0)
    first = 1 + seg_first * segment
    last = min(1 + (seg_first + seg_count) * segment, 1 + total)
    count = max(last - first, 0)
    if count == 0:
        first = min(first, nframes)
    return Shard(rank, world, first, count, max(first - nrefs, 0))
