// tq_common.h -- what the two implementations of the transform / quantisation path share
// (tq_kernels.hip: LDS-resident partial butterflies on the VALU; tq_mfma.hip: integer matrix products)
#pragma once
#include "common.h"

namespace x265amd {

struct TqParams {
    x265amd_tq_desc d;
    int tiles_x, tiles_y, ntiles;  // 32x32 pixel tiles
    // quantiser constants (src/pixel.cpp:595-602, 638-644), computed once on the host
    unsigned q_scale;
    int q_bits;
    unsigned q_rnd;
    int dq_scale, dq_shift, dq_rnd;
    int mv_ctu_log2, mv_ctus_x;    // with d.d_mv: log2 of d.mv_ctu, vectors per picture row
    unsigned long long *d_trace;   // x265amd_debug_tq_trace: four 64-bit words per wave of tq_mfma_kernel, or null
    int trace_waves;               // .. and how many waves it has room for
    // x265amd_intra_encode_frame: only the tile units that hold the CUs of one anti-diagonal (cux + 2 cuy == diag_d,
    // cuy = diag_lo + blockIdx.x, CUs of 1 << diag_cu_log2 pixels in THIS plane); diag_count = 0: the whole plane
    int diag_count, diag_d, diag_lo, diag_cu_log2;
    // .. for a batch of frames that hold `diag_fgroup` planes each (U and V of 4:2:0 frame buffers): blockIdx.y = g *
    // diag_fgroup + j is plane j of frame g and sits d.*_frame_stride * (g * diag_fmul + j) behind the first plane (the
    // frame strides are then the distance from one plane to the next inside a frame); its sums go to d_abs_sum +
    // g * d.abs_sum_frame_stride + j * diag_sum_pair.  diag_fgroup = 0: blockIdx.y is the frame.
    int diag_fgroup, diag_fmul;
    long long diag_sum_pair;
};

enum { TQ_FWD = 1, TQ_INV = 2, TQ_BOTH = 3 };

// tq_mfma.hip: all TU sizes, widths that are multiples of 16; the 4x4 DST (per-TU intra modes) stays with the butterfly kernel
int launch_tq_mfma(int mode, const TqParams &p, hipStream_t stream);

// what x265amd_intra_encode_frame tells the transform stage about the anti-diagonal it follows
struct TqDiag {
    int count, d, lo, cu_log2;
    int fgroup = 0, fmul = 0;   // see TqParams::diag_fgroup; the descriptor's nframes is then the number of frames g
    long long sum_pair = 0;
};
// whether launch_tq would take the matrix-core kernel for this descriptor (the only one that follows a diagonal)
bool tq_takes_mfma(const x265amd_tq_desc *desc);
// x265amd_tq_roundtrip restricted to the tile units of an anti-diagonal's CUs where the matrix-core kernel runs
// (the whole plane otherwise: the result is the same, see intra_kernels.hip)
int tq_roundtrip_diag(const x265amd_tq_desc *desc, const TqDiag &diag, hipStream_t stream);
// the luma and the chroma launch of one anti-diagonal step as ONE launch (both descriptors on the matrix-core path,
// chroma TU = luma TU / 2, no vectors; X265AMD_ERR_ARG otherwise: the caller then launches them one by one)
int tq_roundtrip_diag_yuv(const x265amd_tq_desc *luma, const TqDiag &dl, const x265amd_tq_desc *chroma, const TqDiag &dc,
                          hipStream_t stream);
int launch_tq_mfma_diag_yuv(const TqParams &py, const TqParams &pc, hipStream_t stream);
// .. and the same step's kernel parameters without the launch, checked (for a caller whose own kernel runs tq_mfma_body
// in diagonal mode: intra_wave_tq_kernel); ny_luma / ny_chroma = planes that py / pc address (the `by` range)
int tq_diag_yuv_params(const x265amd_tq_desc *luma, const TqDiag &dl, const x265amd_tq_desc *chroma, const TqDiag &dc, TqParams &py,
                       TqParams &pc, int &ny_luma, int &ny_chroma);
int tq_mfma_diag_yuv_check(TqParams &py, TqParams &pc, int &ny_luma, int &ny_chroma);

}  // namespace x265amd
