// me_common.h -- geometry shared by the motion-search kernels
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace x265amd {

struct MeGeom {
    int width, height, stride, range, nrefs, ctus_x, ctus_y;
    int64_t cur_fs, ref_fs, ref_rs;
};

constexpr int kMeT = 3;  // dy rows per lane: 129 = 43 * 3 and 257 = 86 * 3 - 1 leave (almost) no waste

// blk = block size (64 for a CTU; 32 / 16 for the PU sizes)
__host__ __device__ inline void me_qsad_layout(int w, int h, int range, int cx, int cy, int &dylo, int &dyhi, int &dxlo,
                                               int &dxhi, int &wx0, int &ngx, int &ngy, int &pitch, int &win_rows,
                                               int blk = 64)
{
    dylo = -range > -cy ? -range : -cy;
    dyhi = range < h - blk - cy ? range : h - blk - cy;
    dxlo = -range > -cx ? -range : -cx;
    dxhi = range < w - blk - cx ? range : w - blk - cx;
    wx0 = (cx + dxlo) & ~3;                       // window origin on a dword boundary
    ngx = (cx + dxlo - wx0 + (dxhi - dxlo + 1) + 3) >> 2;  // groups of 4 consecutive dx
    ngy = (dyhi - dylo + 1 + kMeT - 1) / kMeT;
    pitch = (ngx + blk / 4 + 1) | 1;              // dwords; odd => rows spread over the banks
    win_rows = blk + kMeT * ngy;
}


}  // namespace x265amd
