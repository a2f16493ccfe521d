// me_qsad_body.h -- the every-candidate search of one block, shared by me_fullsearch_qsad (me_kernels.hip: one
// workgroup per block) and me_list_qsad (me_sea.hip: the blocks the elimination search hands over).
#pragma once
#include "common.h"
#include "me_common.h"

namespace x265amd {

// ---------------------------------------------------------------------------
// Exhaustive search of one block (64x64 CTU; 32 / 16 for the PU sizes) by one 256-thread workgroup.
//
// v_qsad_pk_u16_u8 does the work: one instruction = the 4-pixel SADs of one
// current-block dword against FOUR consecutive byte offsets of an 8-byte
// reference span, accumulated into four u16 lanes -- 16 pixel-SADs per lane per
// issue, and no alignment shuffles at all.  Measured on gfx950 it issues at
// 16.5 cycles/wave (tools/ubench_valu), i.e. 1.03 cycles per pixel-SAD-lane,
// the best the integer VALU offers (v_sad_u8: 4.3 cycles for 4).
//
// Only the candidates that keep the block inside the picture are enumerated
// (border blocks do less work).  A lane owns a 4 (dx) x T (dy) tile of
// candidates.  For every reference row rho of its tile the lane reads the 16+1
// window dwords it needs from LDS (ds_read2_b32 pairs, exactly the 64-bit
// operands qsad wants) and applies them to the T current rows rho, rho-1, ..,
// rho-T+1.  The current rows are the same for every lane, so they live in
// SGPRs (s_load_dwordx16) and enter qsad as the scalar operand: zero VGPRs,
// zero LDS traffic for the current block.  The u16 partial sums are widened to
// u32 every 4 rows (64 * 1020 < 65536).  Winner = min over
// (sad << 32 | raster index) by wave shuffles -- the reference's "first strict
// minimum in raster order" (src/encode.cpp:762-765).
//
// gy0 .. gy1: the rows of candidate tiles (T dy each) this call takes -- the whole block's, or a slice when several
// workgroups share a block.  `win` = dynamic LDS of at least (B + T * (gy1 - gy0)) rows x pitch dwords.  Returns
// the winning key in thread 0 (~0 if the slice holds no legal candidate); ends behind a barrier that every thread
// has passed, but the caller must put another before `win` is written again.
// ---------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ unsigned long long me_qsad_block(uint32_t *__restrict__ win, const uint8_t *__restrict__ cur,
                                                            const uint8_t *__restrict__ ref, const MeGeom &g, int cx, int cy,
                                                            int slice, int nslices)
{
    constexpr int T = kMeT, DW = B / 4;
    const int tid = threadIdx.x;
    const int R = g.range;
    int dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows;
    me_qsad_layout(g.width, g.height, R, cx, cy, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows, B);
    const int gy0 = (int)((long long)ngy * slice / nslices), gy1 = (int)((long long)ngy * (slice + 1) / nslices);
    const int rows = B + T * (gy1 - gy0);   // window rows of this slice
    const int wy0 = cy + dylo + T * gy0;    // picture row of the first

    // ---- stage the reference window: coalesced dword loads of 8-bit luma ----
    for (int i = tid; i < rows * pitch; i += 256) {
        const int wy = i / pitch, wq = i - wy * pitch;
        const int gx = wx0 + 4 * wq, gy = wy0 + wy;
        uint32_t v = 0;  // outside the picture: only ever paired with candidates that are masked below
        if (gy < g.height && gx + 4 <= g.width) v = *reinterpret_cast<const uint32_t *>(ref + (int64_t)gy * g.stride + gx);
        win[i] = v;
    }
    __syncthreads();

    unsigned long long best = ~0ull;
    const int ntask = ngx * (gy1 - gy0);
    const int span = 2 * R + 1;
    const uint32_t *__restrict__ cbase = reinterpret_cast<const uint32_t *>(cur + (int64_t)cy * g.stride + cx);
    const int cstr = g.stride >> 2;
    // whole waves iterate together (the current rows are wave-uniform scalar operands)
    for (int task = tid; task < ((ntask + 63) & ~63); task += 256) {
        const int tk = min(task, max(ntask - 1, 0));
        const int gyl = tk / ngx, gx = tk - gyl * ngx;
        const uint32_t *wrow = win + (T * gyl) * pitch + gx;
        uint64_t acc16[T];
        uint32_t acc32[T][4];
#pragma unroll
        for (int t = 0; t < T; t++) {
            acc16[t] = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) acc32[t][j] = 0;
        }
        uint32_t S[T][DW];  // sliding window of T current rows, SGPRs
#pragma unroll 1  // one T-row step per trip: unrolled further the 32x32 instance spills
        for (int m = 0; m < (B + T - 1 + T - 1) / T; m++) {
#pragma unroll
            for (int j = 0; j < T; j++) {
                const int rho = T * m + j;  // reference row of the tile == newest current row
                if (rho < B) {
#pragma unroll
                    for (int k = 0; k < DW; k++) S[j][k] = cbase[rho * cstr + k];
                }
                if (rho < B + T - 1) {
                    const uint32_t *p = wrow + rho * pitch;
                    uint64_t pr[DW];
#pragma unroll
                    for (int k = 0; k < DW; k++) pr[k] = ((uint64_t)p[k + 1] << 32) | p[k];
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const int r = rho - t;  // current row paired with this reference row for dy offset t
                        if (r >= 0 && r < B) {
                            const int sj = (j - t + T) % T;
#pragma unroll
                            for (int k = 0; k < DW; k++)
                                acc16[t] = __builtin_amdgcn_qsad_pk_u16_u8(pr[k], S[sj][k], acc16[t]);
                            if ((r & 3) == 3) {  // 4 rows * B/4 qsads * 1020 < 2^16: widen before it can wrap
#pragma unroll
                                for (int q = 0; q < 4; q++) acc32[t][q] += (uint32_t)(acc16[t] >> (16 * q)) & 0xffffu;
                                acc16[t] = 0;
                            }
                        }
                    }
                }
            }
        }
        if (task < ntask) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                const int dy = dylo + T * (gy0 + gyl) + t;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int dx = wx0 - cx + 4 * gx + q;
                    if (dy <= dyhi && dx >= dxlo && dx <= dxhi) {
                        const unsigned long long key =
                            ((unsigned long long)acc32[t][q] << 32) | (unsigned)((dy + R) * span + dx + R);
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    __shared__ unsigned long long wbest[4];
    if ((tid & 63) == 0) wbest[tid >> 6] = best;
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w < 4; w++) best = wbest[w] < best ? wbest[w] : best;
    return best;
}

__device__ __forceinline__ x265amd_mv me_key_to_mv(unsigned long long best, int R)
{
    const int span = 2 * R + 1;
    x265amd_mv mv;
    const int c = (int)(best & 0xffffffffu);
    mv.mvx = (int16_t)(c % span - R);
    mv.mvy = (int16_t)(c / span - R);
    mv.sad = (uint32_t)(best >> 32);
    return mv;
}

}  // namespace x265amd
