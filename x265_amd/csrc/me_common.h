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

// Reductions over the 64 lanes of a wave without the LDS crossbar: four DPP steps inside the rows of 16 lanes (pairs,
// quads, half rows, rows), then the four row results through v_readlane -- every lane gets the answer.  ALL 64 lanes
// must be active at the call.  (A __shfl_xor ladder is six dependent ds_bpermute trips, twelve for a 64-bit key:
// about a thousand cycles on the critical path of every round of the exact kernel.)
template <int CTRL> __device__ __forceinline__ uint32_t me_dpp(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t me_rows4_min(uint32_t s)
{
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)s, 0), (uint32_t)__builtin_amdgcn_readlane((int)s, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)s, 32), (uint32_t)__builtin_amdgcn_readlane((int)s, 48)));
}
__device__ __forceinline__ uint32_t me_wave_min_u32(uint32_t s)
{
    s = min(s, me_dpp<0xb1>(s));   // quad_perm [1,0,3,2]
    s = min(s, me_dpp<0x4e>(s));   // quad_perm [2,3,0,1]
    s = min(s, me_dpp<0x141>(s));  // row_half_mirror
    s = min(s, me_dpp<0x140>(s));  // row_mirror
    return me_rows4_min(s);
}
__device__ __forceinline__ uint32_t me_wave_max_u32(uint32_t s)
{
    s = max(s, me_dpp<0xb1>(s));
    s = max(s, me_dpp<0x4e>(s));
    s = max(s, me_dpp<0x141>(s));
    s = max(s, me_dpp<0x140>(s));
    return max(max((uint32_t)__builtin_amdgcn_readlane((int)s, 0), (uint32_t)__builtin_amdgcn_readlane((int)s, 16)),
               max((uint32_t)__builtin_amdgcn_readlane((int)s, 32), (uint32_t)__builtin_amdgcn_readlane((int)s, 48)));
}
__device__ __forceinline__ uint32_t me_wave_sum_u32(uint32_t s)
{
    s += me_dpp<0xb1>(s);
    s += me_dpp<0x4e>(s);
    s += me_dpp<0x141>(s);
    s += me_dpp<0x140>(s);
    return (uint32_t)__builtin_amdgcn_readlane((int)s, 0) + (uint32_t)__builtin_amdgcn_readlane((int)s, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)s, 32) + (uint32_t)__builtin_amdgcn_readlane((int)s, 48);
}
// inclusive prefix sum over the lanes of a wave: shifts by 1, 2, 4, 8 inside the rows (a lane without a source gets 0),
// then the totals of the rows before the lane's own through v_readlane.  ALL 64 lanes active.
__device__ __forceinline__ uint32_t me_wave_scan_u32(uint32_t s, int lane)
{
    s += me_dpp<0x111>(s);  // row_shr:1
    s += me_dpp<0x112>(s);  // row_shr:2
    s += me_dpp<0x114>(s);  // row_shr:4
    s += me_dpp<0x118>(s);  // row_shr:8
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)s, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)s, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)s, 47);
    return s + (lane >= 16 ? t0 : 0u) + (lane >= 32 ? t1 : 0u) + (lane >= 48 ? t2 : 0u);
}
// minimum of 64-bit keys (sad << 32 | index): the smallest upper half first, then the smallest lower half among the
// lanes that hold it -- two 32-bit reductions, fewer registers and instructions than 64-bit compares and selects
__device__ __forceinline__ unsigned long long me_wave_min_u64(unsigned long long k)
{
    const uint32_t hi = (uint32_t)(k >> 32);
    const uint32_t m = me_wave_min_u32(hi);
    const uint32_t lo = me_wave_min_u32(hi == m ? (uint32_t)k : 0xffffffffu);
    return ((unsigned long long)m << 32) | lo;
}

// XCD-aware block order for a kernel launched with one workgroup per block on a (blocks_x, blocks_y, planes) grid: workgroups
// are dispatched round-robin over the 8 XCDs, each with an L2 of its own, and neighbouring blocks share most of their search
// windows -- so XCD k takes the k-th contiguous eighth of the blocks, in raster order (the every-candidate kernel at 1080p:
// 1.150 -> 1.059 ms per 4 frames).  A bijection of the grid: every workgroup gets a block.
__device__ __forceinline__ void me_xcd_block(int &bx, int &by, int &bz)
{
    const int nx = (int)gridDim.x, per_plane = nx * (int)gridDim.y, total = per_plane * (int)gridDim.z;
    const int lin = (int)blockIdx.x + nx * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    const int blk = xcd * base + (xcd < rem ? xcd : rem) + slot;  // slot < base + (xcd < rem)
    bz = blk / per_plane;
    const int cin = blk - bz * per_plane;
    by = cin / nx;
    bx = cin - by * nx;
}

}  // namespace x265amd
