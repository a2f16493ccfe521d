"""x265_amd -- MI355X (gfx950) native ME + transform/quant hot path of chenm001/x265.

The product is lib/libx265amd.so (hand-written HIP kernels behind the C-ABI of
include/x265_amd.h).  This package is its Python host side: ctypes bindings that
take torch CUDA tensors (torch = device memory + streams only), the synthetic
sequence generator, and the frame-sharding helper for multi-GPU runs.
"""
from .api import (LIB_PATH, MAX_SAD, MODE_INVALID, MV_DTYPE, SLICE_B, SLICE_I, SLICE_P, X265AmdError,  # noqa: F401
                  chroma_qp, dequant_plane, idct_add, init, intra_decide_luma, intra_encode_frame, intra_encode_frames, intra_write_picture, load_library, me_fullsearch, me_fullsearch_pu, me_fullsearch_pu_all, PU_SHAPES, me_static_workspace_bytes, me_workspace_bytes, mv_to_numpy, pack_levels, quant_plane,
                  sad_blocks, satd_blocks, tq_forward, tq_inverse, tq_roundtrip, unpack_levels)

from . import dropin  # noqa: F401,E402
from .pipeline import ClosedLoopGop, MeTqPipeline  # noqa: F401,E402
from .sharding import Shard, shard_frames  # noqa: F401,E402

__version__ = "0.1.0"
