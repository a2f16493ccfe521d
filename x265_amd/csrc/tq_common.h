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
};

enum { TQ_FWD = 1, TQ_INV = 2, TQ_BOTH = 3 };

// tq_mfma.hip: all TU sizes, widths that are multiples of 16; the 4x4 DST (per-TU intra modes) stays with the butterfly kernel
int launch_tq_mfma(int mode, const TqParams &p, hipStream_t stream);

}  // namespace x265amd
