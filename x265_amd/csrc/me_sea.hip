// me_sea.hip -- exhaustive integer-pel search with EXACT successive elimination.
//
// Same result as x265amd_me_fullsearch (identical vectors and costs, ties
// included), fewer SADs.  For a candidate c the sum over the sixteen 16x16
// sub-blocks b of |sum_b(cur) - sum_b(ref at c)| is a lower bound LB(c) <= SAD(c)
// (triangle inequality).  A candidate whose bound exceeds the SAD of ANY
// candidate already evaluated cannot be the strict minimum, and cannot tie it
// either, so dropping it changes nothing: if the true winner c* were dropped we
// would have LB(c*) > best >= SAD(c*) >= LB(c*).  Everything with LB <= best is
// evaluated exactly with v_qsad_pk_u16_u8, and the winner is the minimum of the
// 64-bit key (sad << 32 | raster index) over what was evaluated -- the
// reference's "first strict minimum" (src/encode.cpp:762-765).
//
// Three kernels:
//   box16_pairs       : P[y][x] = B16[y][x] | B16[y][x+16] << 16, B16 = sum of the 16x16 reference block at
//                       (x, y) (u16): horizontal sums, vertical running sums and pairing in one tiled pass;
//   sea_bounds_kernel : per (CTU, ref, frame) workgroup, small LDS -- bounds of every group of four
//                       candidates (v_sad_u16 on the pair plane against the CTU's own sub-block sums),
//                       histogram pick of a first batch, an exact-SAD probe for the work estimate;
//   sea_exact_kernel  : persistent workgroups, longest CTUs first: window to LDS, exact SADs of the first
//                       batch, then of everything whose bound is still <= best.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "me_common.h"
#include "me_qsad_body.h"

namespace x265amd {

// ------------------------------------------------------------------ box sums
// P[y][x] = B16[y][x] | B16[y][x + 16] << 16, where B16[y][x] = sum of the 16x16 reference block at (x, y)
// (u16: 16 * 16 * 255 < 2^16).  The two sub-blocks of one candidate that sit side by side share a dword,
// so that one v_sad_u16 adds both |differences| to the candidate's bound.
//
// One kernel, one pass over the picture: a workgroup makes a 112 x 64 tile of P.  (1) horizontal 16-sums
// H16 of the 128 columns x 79 rows the tile needs, 8 outputs per item from 6 dwords of pixels, into LDS;
// (2) vertical running sums down the columns -- a thread owns a column pair (packed u16) and a quarter of
// the rows -- into LDS; (3) pairs out, 8 bytes per lane.  HBM: 1.5 B/px read, 4 B/px written (two kernels
// with H16 in memory moved 9.5 B/px).  (3) writes 16 bytes per lane.
constexpr int kBoxTW = 112, kBoxTH = 64;   // tile of P
constexpr int kBoxPitch = 136;             // u16 per LDS row: 272 B, a multiple of 8
// S = side of the sub-blocks: 16 for the 64x64 search, 8 for the 32x32 search (a quarter of the block either way)
template <int S>
__global__ __launch_bounds__(256) void box16_pairs(const uint8_t *__restrict__ ref_base, uint32_t *__restrict__ pbuf,
                                                   int width, int height, int stride, int nrefs, int64_t ref_fs,
                                                   int64_t ref_rs, uint32_t *__restrict__ ctrl)
{
    constexpr int kBoxCols = kBoxTW + S;      // B / H columns per tile (S = 16: 64 pairs = one per thread and quarter)
    constexpr int kBoxRows = kBoxTH + S - 1;  // H rows per tile
    constexpr int ND = S / 4 + 1;             // dwords of pixels under four neighbouring S-sums
    static_assert(kBoxCols % 4 == 0 && kBoxCols <= kBoxPitch && kBoxCols <= 128, "tile geometry");
    __shared__ __attribute__((aligned(16))) uint16_t Ht[kBoxTH + 15][kBoxPitch];
    __shared__ __attribute__((aligned(16))) uint16_t Bt[kBoxTH][kBoxPitch];
    // the queue counters of the later kernels start from zero (stream order: this grid ends first)
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) ctrl[threadIdx.x] = 0;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kBoxTW, y0 = blockIdx.y * kBoxTH, pl = blockIdx.z;
    const int f = pl / nrefs, r_ = pl - f * nrefs;
    const uint8_t *ref = ref_base + f * ref_fs + r_ * ref_rs;

    // (1) H[r][c] = sum_{k<S} p[y0 + r][x0 + c + k]; pixels outside the picture count as 0 (sums that
    //     include them belong to sub-blocks no legal candidate uses)
    static_assert(kBoxCols % 8 == 0, "eight outputs per item");
    for (int i = tid; i < kBoxRows * (kBoxCols / 8); i += 256) {
        const int r = i / (kBoxCols / 8), go = i - r * (kBoxCols / 8);
        const int y = y0 + r, x = x0 + 8 * go;
        uint32_t d[ND + 1];  // the pixels under eight neighbouring S-sums
#pragma unroll
        for (int k = 0; k < ND + 1; k++)
            d[k] = (y < height && x + 4 * k + 4 <= width) ? *reinterpret_cast<const uint32_t *>(ref + (int64_t)y * stride + x + 4 * k) : 0u;
        uint32_t o[8];
        o[0] = 0;
#pragma unroll
        for (int k = 0; k < ND - 1; k++) o[0] = __builtin_amdgcn_sad_u8(d[k], 0u, o[0]);
#pragma unroll
        for (int j = 1; j < 8; j++) {  // slide by one pixel: byte j - 1 leaves, byte j - 1 + S enters
            const int out = j - 1, in = j - 1 + S;
            o[j] = o[j - 1] - ((d[out >> 2] >> (8 * (out & 3))) & 255u) + ((d[in >> 2] >> (8 * (in & 3))) & 255u);
        }
        *reinterpret_cast<uint4 *>(&Ht[r][8 * go]) = make_uint4(o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16));
    }
    __syncthreads();

    // (2) B[r][c] = sum_{j<S} H[r + j][c]: packed u16 pairs; S * S * 255 < 2^16, so the packed adds never
    //     carry across, and the old row is taken out before the new one goes in (no borrow either)
    {
        const int cp = tid & 63, q = tid >> 6;  // column pair, quarter of the tile's rows
        constexpr int pd = kBoxPitch / 2;       // LDS pitch in dwords
        if (2 * cp < kBoxCols) {
            const uint32_t *h = reinterpret_cast<const uint32_t *>(&Ht[0][0]) + cp;
            uint32_t *bt = reinterpret_cast<uint32_t *>(&Bt[0][0]) + cp;
            const int rb = 16 * q;
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < S; k++) acc += h[(rb + k) * pd];
            bt[rb * pd] = acc;
#pragma unroll
            for (int j = 1; j < 16; j++) {
                acc = (acc - h[(rb + j - 1) * pd]) + h[(rb + j + S - 1) * pd];
                bt[(rb + j) * pd] = acc;
            }
        }
    }
    __syncthreads();

    // (3) pairs out: P[y][x] = B[y][x] | B[y][x + S] << 16, four columns (16 bytes) per item
    static_assert(kBoxTW % 4 == 0, "four columns per item");
    for (int i = tid; i < kBoxTH * (kBoxTW / 4); i += 256) {
        const int r = i / (kBoxTW / 4), c = i - r * (kBoxTW / 4);
        const int y = y0 + r, x = x0 + 4 * c;
        if (y + S > height || x + 4 > width) continue;  // the width is a multiple of 4 (of the block side, in fact)
        const uint2 an = *reinterpret_cast<const uint2 *>(&Bt[r][4 * c]);
        const uint2 af = *reinterpret_cast<const uint2 *>(&Bt[r][4 * c + S]);
        *reinterpret_cast<uint4 *>(pbuf + ((int64_t)pl * height + y) * width + x) =
            make_uint4((an.x & 0xffffu) | (af.x << 16), (an.x >> 16) | (af.x & 0xffff0000u), (an.y & 0xffffu) | (af.y << 16),
                       (an.y >> 16) | (af.y & 0xffff0000u));
    }
}

// ------------------------------------------------------------------ search
struct SeaLayout {
    int dylo, dyhi, dxlo, dxhi, wx0, ngx, ndy, pitch, win_rows;
    int ngroups;
};

// B = side of the searched block (64 or 32)
__host__ __device__ inline SeaLayout sea_layout(int w, int h, int range, int cx, int cy, int B = 64)
{
    SeaLayout L;
    L.dylo = -range > -cy ? -range : -cy;
    L.dyhi = range < h - B - cy ? range : h - B - cy;
    L.dxlo = -range > -cx ? -range : -cx;
    L.dxhi = range < w - B - cx ? range : w - B - cx;
    L.wx0 = (cx + L.dxlo) & ~3;
    L.ngx = (cx + L.dxlo - L.wx0 + (L.dxhi - L.dxlo + 1) + 3) >> 2;
    L.ndy = L.dyhi - L.dylo + 1;
    L.pitch = (L.ngx + B / 4 + 1) | 1;
    L.win_rows = B + L.ndy;
    L.ngroups = L.ngx * L.ndy;
    return L;
}

constexpr int kBins = 96;         // quarter-octave bins of the bound: covers 24 octaves

// Monotone bin of a stored bound: exponent and top two mantissa bits of its float image (exact for
// lb < 2^24): v_cvt_f32_u32 + shift + subtract.  0 -> 0, 1 -> 1, 2 -> 5, 3 -> 7, 4 -> 9, ..., 65534 -> 64.
__device__ __forceinline__ int lb_bin(uint32_t lb)
{
    const int e = (int)(__float_as_uint((float)lb) >> 21) - 507;
    return e < 0 ? 0 : e;
}
// The smallest stored bound of bin `bin` or of a later one (bins that no integer reaches are empty): lb_bin(v) < bin  <=>
// v < lb_bin_first(bin).  0 for bin <= 0; kNone (0xffff, above every stored bound) from kBins on.
__device__ __forceinline__ uint32_t lb_bin_first(int bin)
{
    if (bin <= 0) return 0u;
    if (bin >= kBins) return 0xffffu;
    const float f = __uint_as_float((uint32_t)(bin + 507) << 21);  // the float whose bits open the bin: >= 1, below 2^24
    const uint32_t v = (uint32_t)f;
    return min(v + ((float)v < f ? 1u : 0u), 0xffffu);  // its ceiling
}

// chain steps of S rows that a lane may walk (a multiple of 5, the rotation period of its band registers):
// ndy <= 257 (R <= 128) needs 17 at S = 16 and 33 at S = 8
__host__ __device__ constexpr int sea_max_chain(int S) { return S == 16 ? 20 : 35; }
constexpr int kBoundsThreads = 512;  // 8 waves, 4 workgroups per CU; blocks of 32 with a first batch of 256: 4 waves, 8 workgroups
                                     // (a +-64 window has 8 x 33 = 264 chains there: half of 512 threads would walk none)
constexpr uint32_t kIllegal = 1u << 30;  // > any bound (16 * 65280)
// the search geometry of one launch: the kernels are instantiated for blocks of 64 (sub-blocks of 16) and of 32 (8)
constexpr uint32_t kNone = 0xffffu;  // stored bound: no legal member, or already scheduled

// Bounds are kept as u16 = floor(LB / 16) (LB <= 16 * 65280): still a lower bound, 15 units coarser.
__device__ __forceinline__ uint32_t lb_store(uint32_t lb) { return lb >> 4; }

struct SeaHdr {  // per (plane, CTU)
    uint32_t n0_rest;    // low half: groups in the first batch (<= kFirstMax); high half: smallest stored bound among the
                         // groups NOT in it (kNone: none left)
    uint32_t est;        // work estimate: groups with a bound <= the probed SAD (an upper bound of n0 + second pass)
    uint32_t probe_sad;  // the probe's best candidate: its exact SAD and raster index -- where kernel 2's search starts
    uint32_t probe_idx;  // (0xffffffff / 0xffffffff: no probe)
};
constexpr uint32_t kRouted = 0xffffffffu;  // SeaHdr::n0_rest of a block on the every-candidate list (n0 <= kFirstMax otherwise)
constexpr int kFirstMax = 512;  // first batch = ONE half round of kernel 2: 256 groups for its 8-wave shape, 512 for the 16-wave one
constexpr int kMaxWgLists = 512;  // persistent workgroups that may need a schedule slice in the workspace
constexpr int kClasses = 4;   // launch-order classes of the work estimate, heaviest first
// ctrl[] (8 counters, zeroed by box16_pairs): [0 .. kClasses) lengths of the class queues, [kClasses] next item of kernel 2,
// [kClasses + 1] length of the every-candidate list
static_assert(kClasses + 2 <= 8, "box16_pairs zeroes 8 counters");
__device__ __forceinline__ int sea_class(int est) { return est >= 2048 ? 0 : est >= 1024 ? 1 : est >= 850 ? 2 : 3; }

// ---------------------------------------------------------------------------------------------
// Kernel 1: bounds.  No window, 9 KB of LDS => high occupancy hides the box-sum loads.
//   out: gl[plane][ctu][max_groups] u16 bound per group (kNone where scheduled / illegal),
//        list0[plane][ctu][..] the first batch (the `nfirst` groups of the lowest histogram bins), hdr
// ---------------------------------------------------------------------------------------------
template <int B, int NT>
__global__ __launch_bounds__(NT, 8) void sea_bounds_kernel(const uint8_t *__restrict__ cur_base,
                                                                   const uint8_t *__restrict__ ref_base,
                                                                 const uint32_t *__restrict__ pbuf,
                                                                 uint16_t *__restrict__ gl, uint16_t *__restrict__ list0,
                                                                 SeaHdr *__restrict__ hdr, uint32_t *__restrict__ order,
                                                                 uint32_t *__restrict__ ctrl, const MeGeom g, int gl_pitch,
                                                                 int nplanes, int nfirst, uint32_t *__restrict__ stats)
{
    constexpr int S = B / 4;  // side of a sub-block
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint16_t *glb = reinterpret_cast<uint16_t *>(smem);  // [ndy][ngx]
    __shared__ uint32_t hist[kBins];
    __shared__ uint32_t csum[16];
    __shared__ int list_n, eq_n, pick_bin, below_n, est_n;
    __shared__ uint32_t min_key, max_st, probe_sad, probe_idx, pref[kBins];
    __shared__ uint32_t rest_lb;
    __shared__ uint32_t bank_cnt[32], rank_base[64], rank_mask[64];
    __shared__ uint16_t sel_ent[kFirstMax];

    const int tid = threadIdx.x, lane = tid & 63;
    // XCD-aware order: workgroup w runs on XCD w % 8 (round-robin dispatch), each XCD has its own L2,
    // and neighbouring CTUs share two thirds of their box-sum windows.  So XCD k takes the k-th
    // contiguous eighth of the CTUs, in raster order (PMC: 905 -> 166 MB fetched per 16 1080p frames,
    // 0.170 -> 0.143 ms).
    const int total = g.ctus_x * g.ctus_y * nplanes;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int base = total >> 3, rem = total & 7;
    if (slot >= base + (xcd < rem ? 1 : 0)) return;  // grid is rounded up to a multiple of 8 (block-uniform exit)
    const int ctu_lin = xcd * base + (xcd < rem ? xcd : rem) + slot;
    const int plane = ctu_lin / (g.ctus_x * g.ctus_y), cin = ctu_lin - plane * (g.ctus_x * g.ctus_y);
    const int cty = cin / g.ctus_x, ctx = cin - cty * g.ctus_x;
    const int f = plane / g.nrefs;
    const int cx = ctx * B, cy = cty * B;
    const SeaLayout L = sea_layout(g.width, g.height, g.range, cx, cy, B);
    const uint8_t *cur = cur_base + f * g.cur_fs;
    const uint32_t *P = pbuf + (int64_t)plane * g.height * g.width;
    const int64_t ctu_id = ctu_lin;

    const long long tb0 = stats ? clock64() : 0;
    if (tid < kBins) hist[tid] = 0;
    if (tid < 32) bank_cnt[tid] = 0;
    if (tid == 0) {
        list_n = 0;
        eq_n = 0;
        min_key = 0xffffffffu;
        max_st = 0;
        rest_lb = kNone;
    }
    // ---- the block's sub-block sums: S lanes per sub-block, one S-pixel row each ----
    if (tid < 16 * S) {
        const int b = tid / S, rrow = tid % S;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(cur + (int64_t)(cy + S * (b >> 2) + rrow) * g.stride + cx + S * (b & 3));
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < S / 4; k++) s = __builtin_amdgcn_sad_u8(p[k], 0u, s);
        // the S lanes of a sub-block lie inside one row of 16 lanes (whole waves are here: 16 * S is a multiple of 64)
        s += me_dpp<0xb1>(s);
        s += me_dpp<0x4e>(s);
        if constexpr (S >= 8) s += me_dpp<0x141>(s);
        if constexpr (S >= 16) s += me_dpp<0x140>(s);
        if (rrow == 0) csum[b] = s;
    }
    __syncthreads();
    const long long tb1 = stats ? clock64() : 0;
    uint32_t cs2[8];  // [4 rows of sub-blocks][2 pairs]: sums of sub-blocks (j, 2kp) | (j, 2kp+1) << 16, <= 65280 each
#pragma unroll
    for (int b = 0; b < 8; b++) cs2[b] = __builtin_amdgcn_readfirstlane(csum[2 * b] | (csum[2 * b + 1] << 16));

    // ---- bounds.  Group = 4 window-aligned dx at one dy.  The bound of the group at row dyi needs the
    //      box sums of rows dyi, dyi+16, dyi+32, dyi+48 -- and so do its neighbours 16 rows up and
    //      down.  A lane therefore walks a CHAIN dyi = c, c+16, c+32, .. at fixed gx and keeps the row
    //      bands in registers: 4 eight-byte loads per group instead of 16 (coalesced along gx). ----
    const int a = cx + L.dxlo - L.wx0;  // window column of dxlo
    const int ndx = L.dxhi - L.dxlo + 1;
    uint32_t my_min = 0xffffffffu;  // (stored bound << 16 | group) of the lane's best group
    uint32_t my_max = 0;            // largest stored bound of the lane's groups
    {
        const int nchain = S * L.ngx;  // (c, gx)
        // whole sweeps of chains; a short remainder (the 33rd column of groups of a +-64 window: 16 of
        // 528 chains) is done group by group below instead of costing a second walk
        int nwalk = nchain - nchain % NT;
        if (nchain - nwalk >= NT / 4) nwalk = nchain;
        for (int t = tid; t < nwalk; t += NT) {
            const int c = t / L.ngx, gx = t - c * L.ngx;
            if (c >= L.ndy) continue;
            const uint32_t *pp = P + (int64_t)(cy + L.dylo + c) * g.width + L.wx0 + 4 * gx;
            const int64_t band = (int64_t)S * g.width;
            uint32_t lb0[4];  // accumulator start: 0 for a legal member, 2^30 for one outside the picture
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int dxi = 4 * gx + q - a;
                lb0[q] = (dxi >= 0 && dxi < ndx) ? 0u : kIllegal;
            }
            // five band registers used round-robin: while step m is being summed, band m+4 is in flight
            uint4 bnd[5][2];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kp = 0; kp < 2; kp++) bnd[j][kp] = *reinterpret_cast<const uint4 *>(pp + j * band + 2 * S * kp);
            const uint32_t *ppre = pp + 3 * band;  // the band loaded last
            for (int m0 = 0; m0 < sea_max_chain(S); m0 += 5)  // unrolled by the rotation period: static register indices
#pragma unroll
                for (int mm = 0; mm < 5; mm++) {
                    const int m = m0 + mm;
                    const int dyi = c + S * m;
                    if (dyi >= L.ndy) break;
                    {   // always in flight behind the step's sums: a load under a condition makes the compiler wait for it at
                        // once (both paths must be safe).  The chain's last step reloads the band before.
                        ppre += dyi + S < L.ndy ? band : 0;
#pragma unroll
                        for (int kp = 0; kp < 2; kp++) bnd[(mm + 4) % 5][kp] = *reinterpret_cast<const uint4 *>(ppre + 2 * S * kp);
                    }
                    uint32_t lb[4] = {lb0[0], lb0[1], lb0[2], lb0[3]};
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int kp = 0; kp < 2; kp++) {
                            const uint4 v = bnd[(mm + j) % 5][kp];
                            const uint32_t sc = cs2[2 * j + kp];
                            lb[0] = __builtin_amdgcn_sad_u16(v.x, sc, lb[0]);  // both sub-blocks of the pair at once
                            lb[1] = __builtin_amdgcn_sad_u16(v.y, sc, lb[1]);
                            lb[2] = __builtin_amdgcn_sad_u16(v.z, sc, lb[2]);
                            lb[3] = __builtin_amdgcn_sad_u16(v.w, sc, lb[3]);
                        }
                    const uint32_t mn = min(min(lb[0], lb[1]), min(lb[2], lb[3]));
                    const uint32_t st = mn >= kIllegal ? kNone : lb_store(mn);
                    glb[dyi * L.ngx + gx] = (uint16_t)st;
                    if (st != kNone) {
                        atomicAdd(&hist[lb_bin(st)], 1u);
                        my_min = min(my_min, (st << 16) | (uint32_t)(dyi * L.ngx + gx));
                        my_max = max(my_max, st);
                    }
                }
        }
        // remainder chains, one group per lane: item i = (step m, chain r)
        const int nrem = nchain - nwalk;
        const int nsteps = (L.ndy + S - 1) / S;
        for (int i = tid; i < nrem * nsteps; i += NT) {
            const int m = i / nrem, t = nwalk + (i - m * nrem);
            const int c = t / L.ngx, gx = t - c * L.ngx;
            const int dyi = c + S * m;
            if (dyi >= L.ndy) continue;
            const uint32_t *pp = P + (int64_t)(cy + L.dylo + dyi) * g.width + L.wx0 + 4 * gx;
            const int64_t band = (int64_t)S * g.width;
            uint4 bnd[4][2];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kp = 0; kp < 2; kp++) bnd[j][kp] = *reinterpret_cast<const uint4 *>(pp + j * band + 2 * S * kp);
            uint32_t lb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kp = 0; kp < 2; kp++) {
                    const uint4 v = bnd[j][kp];
                    const uint32_t sc = cs2[2 * j + kp];
                    lb[0] = __builtin_amdgcn_sad_u16(v.x, sc, lb[0]);
                    lb[1] = __builtin_amdgcn_sad_u16(v.y, sc, lb[1]);
                    lb[2] = __builtin_amdgcn_sad_u16(v.z, sc, lb[2]);
                    lb[3] = __builtin_amdgcn_sad_u16(v.w, sc, lb[3]);
                }
            uint32_t mn = 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int dxi = 4 * gx + q - a;
                if (dxi >= 0 && dxi < ndx) mn = min(mn, lb[q]);
            }
            const uint32_t st = mn == 0xffffffffu ? kNone : lb_store(mn);
            glb[dyi * L.ngx + gx] = (uint16_t)st;
            if (st != kNone) {
                atomicAdd(&hist[lb_bin(st)], 1u);
                my_min = min(my_min, (st << 16) | (uint32_t)(dyi * L.ngx + gx));
                my_max = max(my_max, st);
            }
        }
    }
    my_min = me_wave_min_u32(my_min);
    my_max = me_wave_max_u32(my_max);
    if (lane == 0 && my_min != 0xffffffffu) {
        atomicMin(&min_key, my_min);
        atomicMax(&max_st, my_max);
    }
    __syncthreads();

    const long long tb2 = stats ? clock64() : 0;
    // ---- first batch: the nfirst groups of the lowest bins.  pick_bin = the bin in which the running
    //      count reaches nfirst; below_n = groups in the bins under it (all taken), the rest of the
    //      quota comes from pick_bin itself, whichever of its groups the scan reaches first ----
    if (tid < 64) {
        uint32_t c0 = hist[tid], c1 = (tid + 64 < kBins) ? hist[tid + 64] : 0u;
        const uint32_t s0 = me_wave_scan_u32(c0, lane);
        const uint32_t tot0 = (uint32_t)__builtin_amdgcn_readlane((int)s0, 63);
        const uint32_t s1 = me_wave_scan_u32(c1, lane) + tot0;
        const unsigned long long m0 = __ballot(s0 >= (uint32_t)nfirst), m1 = __ballot(s1 >= (uint32_t)nfirst && tid + 64 < kBins);
        const int pbv = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : kBins);
        const int pl = __builtin_amdgcn_readfirstlane(pbv & 63);  // (wave-uniform: it comes from ballots)
        const uint32_t incl = (uint32_t)__builtin_amdgcn_readlane((int)(pbv < 64 ? s0 : s1), pl);  // count through pick_bin
        const uint32_t own = (uint32_t)__builtin_amdgcn_readlane((int)(pbv < 64 ? c0 : c1), pl);
        pref[tid] = s0;
        if (tid + 64 < kBins) pref[tid + 64] = s1;
        if (tid == 0) {
            pick_bin = pbv;
            below_n = pbv < kBins ? (int)(incl - own) : 0;
        }
    }
    __syncthreads();
    const int pb = pick_bin;
    const int nbelow = below_n;            // list0[0 .. nbelow) = bins < pb
    const int quota = nfirst - nbelow;     // list0[nbelow .. nbelow + min(quota, |bin pb|)) = bin pb

    const long long tb3 = stats ? clock64() : 0;
    // ---- the first batch is picked out of the bounds in LDS -- a lane per group, consecutive groups to consecutive
    //      lanes; count, one scan and one atomic per counter and wave, then take -- and marked there; the
    //      smallest bound left over goes to the header.  Then the bounds go out as they stand, 16 bytes per lane. ----
    uint16_t *glo = gl + ctu_id * gl_pitch;
    uint16_t *lo = list0 + ctu_id * gl_pitch;
    uint32_t rest_min = kNone;
    int n0 = 0;
    const int nvec = (L.ngroups + 7) >> 3;
    const int wtid = tid < 64 ? tid : tid - 64;  // wave 1 sits the write-out out: it runs the probe meanwhile
    if (tid >= 64 && tid < 128) {
        // ---- probe (wave 1, while the other seven waves write out): the exact SADs of the group with
        //      the smallest bound.  Their minimum UB is an upper bound of the final best SAD, so the
        //      groups with a bound <= UB are an upper bound of everything kernel 2 will have to
        //      evaluate: its work estimate. ----
        const uint32_t key = min_key;
        unsigned long long ubk = ~0ull;  // exact SAD << 32 | raster index of the best member
        if (key != 0xffffffffu) {  // block-uniform
            const int gi = (int)(key & 0xffffu);
            const int dyi = gi / L.ngx, gx = gi - dyi * L.ngx;
            const int rr = plane - f * g.nrefs;
            const uint8_t *ref = ref_base + f * g.ref_fs + rr * g.ref_rs;
            const int x0 = L.wx0 + 4 * gx;
            constexpr int NW = B / 4;          // dwords of a block row
            const int prow = lane < B ? lane : 0;  // a lane per block row (B = 32: the upper half of the wave adds nothing)
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(ref + (int64_t)(cy + L.dylo + dyi + prow) * g.stride + x0);
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(cur + (int64_t)(cy + prow) * g.stride + cx);
            uint32_t d[NW + 1], cw[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                d[k] = rp[k];
                cw[k] = cp[k];
            }
            d[NW] = x0 + B + 4 <= g.width ? rp[NW] : 0u;  // past the row end only members outside the picture read it
            uint64_t acc16 = 0;
#pragma unroll
            for (int k = 0; k < NW; k++)
                acc16 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 1] << 32) | d[k], cw[k], acc16);  // <= 16 * 1020 per lane
            if (lane >= B) acc16 = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t sq = me_wave_sum_u32((uint32_t)(acc16 >> (16 * q)) & 0xffffu);
                const int dxi = 4 * gx + q - a;
                if (dxi >= 0 && dxi < ndx) {
                    const unsigned long long k = ((unsigned long long)sq << 32) |
                                                 (unsigned)((L.dylo + dyi + g.range) * (2 * g.range + 1) + L.dxlo + dxi + g.range);
                    ubk = k < ubk ? k : ubk;
                }
            }
        }
        if (tid == 64) {
            probe_sad = (uint32_t)(ubk >> 32);
            probe_idx = (uint32_t)ubk;
        }
    } else {
        // The other seven waves, a lane per group and sweep, in two passes over the bounds in LDS: first every lane COUNTS
        // its groups under the pick bin and in it, one wave scan and one atomic per counter and wave hand out the slots,
        // then the lanes go over their groups again and take them.  (A ballot, two atomics and their round trips per sweep
        // made this selection as expensive as the bounds themselves -- half of the kernel at +-128.)
        constexpr int kSweep = NT - 64;
        // the bin is monotone in the bound: "under the pick bin" and "in it" are two comparisons with the smallest bounds of
        // bins pb and pb + 1 (kNone = 0xffff is under neither; pb == kBins: every legal group is under it)
        const uint32_t t0 = lb_bin_first(pb), t1 = lb_bin_first(pb + 1);
        uint32_t cnt = 0;  // groups under the pick bin | groups in it << 16  (at most 74 sweeps: the sums fit 16 bits)
        for (int gi = wtid; gi < L.ngroups; gi += kSweep) {
            const uint32_t v = (uint32_t)glb[gi];
            cnt += (v < t0 ? 1u : 0u) + ((v >= t0 && v < t1) ? 0x10000u : 0u);
        }
        const uint32_t inc = me_wave_scan_u32(cnt, lane), tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        int wl = 0, we = 0;
        if (lane == 0) {
            if (tot & 0xffffu) wl = atomicAdd(&list_n, (int)(tot & 0xffffu));
            if (tot >> 16) we = atomicAdd(&eq_n, (int)(tot >> 16));
        }
        int sl = __builtin_amdgcn_readfirstlane(wl) + (int)((inc - cnt) & 0xffffu);  // the lane's next slot among the groups under pb
        int se = __builtin_amdgcn_readfirstlane(we) + (int)((inc - cnt) >> 16);      // its next rank within bin pb
        for (int gi = wtid; gi < L.ngroups; gi += kSweep) {
            const uint32_t v = (uint32_t)glb[gi];
            const bool below = v < t0, equal = v >= t0 && v < t1;  // (kNone: neither, and it leaves rest_min as it is)
            const bool take = below || (equal && se < quota);
            const int slot = equal ? nbelow + se : sl;
            sl += below ? 1 : 0;
            se += equal ? 1 : 0;
            if (take) {
                sel_ent[slot] = (uint16_t)gi;  // schedule entry = the group's index dyi * ngx + gx
                glb[gi] = (uint16_t)kNone;     // scheduled
            } else {
                rest_min = min(rest_min, v);
            }
        }
    }
    rest_min = me_wave_min_u32(rest_min);
    if (lane == 0 && rest_min != kNone) atomicMin(&rest_lb, rest_min);
    __syncthreads();
    // (what lies past a block's last group in its last 16 bytes is never looked at: kernel 2 checks the index)
    for (int vi = tid; vi < nvec; vi += NT) reinterpret_cast<uint4 *>(glo)[vi] = reinterpret_cast<const uint4 *>(glb)[vi];
    // ---- order the first batch for kernel 2's LDS reads: a lane there reads window dwords
    //      (dyi + row) * pitch + gx + k, so the 32 lanes of a half wave collide whenever their
    //      dyi * pitch + gx agree mod 32.  Deal the entries out bank by bank (rank within the bank first,
    //      then bank): any 32 consecutive slots then hold at most two entries of one bank. ----
    n0 = pb == kBins ? list_n : nbelow + min(quota, eq_n);
    int my_ent = 0, my_bank = 0, my_rank = 0;
    if (tid < n0) {
        my_ent = sel_ent[tid];
        const int dyi = my_ent / L.ngx, gx = my_ent - dyi * L.ngx;
        my_bank = (dyi * L.pitch + gx) & 31;
        my_rank = (int)atomicAdd(&bank_cnt[my_bank], 1u);
    }
    __syncthreads();
    // position of (rank r, bank k) = sum_b min(count_b, r) + #{b < k : count_b > r}: wave 0 tabulates, per rank, the
    // first term (a running sum of how many banks hold more than r entries) and the banks that do, as a bit mask
    if (tid < 64) {
        uint32_t cnt = 0, msk = 0;
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const uint32_t gt = bank_cnt[b] > (uint32_t)tid ? 1u : 0u;
            cnt += gt;
            msk |= gt << b;
        }
        const uint32_t inc = me_wave_scan_u32(cnt, lane);
        rank_base[tid] = inc - cnt;
        rank_mask[tid] = msk;
    }
    __syncthreads();
    if (tid < n0) {
        int pos = 0;
        if (my_rank < 64) {
            pos = (int)rank_base[my_rank] + __popc(rank_mask[my_rank] & ((1u << my_bank) - 1u));
        } else {  // more than 64 entries of one bank: the sum written out
            for (int b = 0; b < 32; b++) {
                const int cb = (int)bank_cnt[b];
                pos += min(cb, my_rank) + ((b < my_bank && cb > my_rank) ? 1 : 0);
            }
        }
        lo[pos] = (uint16_t)my_ent;
    }
    if (tid == 0) {
        const uint32_t ub = probe_sad;  // published by the barriers since
        // A block on which the bounds are useless by a wide margin -- even the largest is under a quarter of the probe's
        // SAD: noise, grain -- costs the elimination search more than every candidate's SAD costs the plain kernel: it
        // goes on the list of kernel 3.  Not when the probe found a perfect match (flat content: ties in raster order
        // end that search after the first batch).
        const bool route = ub != 0xffffffffu && ub > 0u && pref[kBins - 1] >= 64u && (max_st << 6) <= ub;
        const uint32_t ubs = ub == 0xffffffffu ? 0u : min(ub >> 4, 0xfffeu);  // same units as the stored bounds
        est_n = (int)pref[min(lb_bin(ubs), kBins - 1)];
        SeaHdr h;
        h.n0_rest = (uint32_t)n0 | (rest_lb << 16);  // kNone: nothing left over
        h.est = (uint32_t)est_n;
        h.probe_sad = ub;
        h.probe_idx = probe_idx;
        hdr[ctu_id] = h;
        // launch order of kernel 2: longest first, in kClasses classes of the estimate (a queue each)
        if (stats) {
            stats[16 * ctu_id + 10] = (uint32_t)(tb1 - tb0);        // cycles: sub-block sums of the CTU
            stats[16 * ctu_id + 11] = (uint32_t)(tb2 - tb1);        //         bounds
            stats[16 * ctu_id + 12] = (uint32_t)(tb3 - tb2);        //         histogram scan
            stats[16 * ctu_id + 13] = (uint32_t)(clock64() - tb3);  //         write-out
        }
        if (route) {
            hdr[ctu_id].n0_rest = kRouted;  // (kernel 2 never sees this block: it is in none of its queues)
            unsigned long long *bk = reinterpret_cast<unsigned long long *>(glo);  // the block's row of bounds is free: merge key + count
            bk[0] = ~0ull;
            reinterpret_cast<uint32_t *>(glo)[2] = 0u;
            order[(size_t)kClasses * (uint32_t)total + atomicAdd(&ctrl[kClasses + 1], 1u)] = (uint32_t)ctu_id;
        } else {
            const int cls = sea_class(est_n);
            order[(size_t)cls * (uint32_t)total + atomicAdd(&ctrl[cls], 1u)] = (uint32_t)ctu_id;
        }
    }
}

// One wave's part of a (half) round of kernel 2: the RQ block rows from `row0` on (wave-uniform) of the lanes' groups
// (`ent` = dyi * ngx + gx, valid where `have`): partial SADs of the four members to `dst` (the lane's 16 bytes of part[]).
// The current rows are wave-uniform: s_load into SGPRs, the scalar operand of v_qsad_pk_u16_u8.
template <int B, int RQ>
__device__ __forceinline__ void sea_half_partial(const uint32_t *__restrict__ win, int pitch, int ngx,
                                                 const uint32_t *__restrict__ cbase, int cstr, uint32_t *__restrict__ dst,
                                                 int ent, bool have, int row0)
{
    if (__ballot(have) == 0ull) return;  // wave-uniform
    ent = have ? ent : 0;
    const int dyi = ent / ngx, gx = ent - dyi * ngx;
    constexpr int NW = B / 4;  // dwords of a block row; RQ = rows taken (even)
    static_assert(RQ % 2 == 0, "two-row pipeline");
    const uint32_t *wrow = win + (dyi + row0) * pitch + gx;
    const uint32_t *crow = cbase + row0 * cstr;
    uint32_t acc32[4] = {0u, 0u, 0u, 0u};
    // two-row software pipeline: the loads of row r+1 (NW + 1 window dwords by ds_read2,
    // NW current dwords by s_load) are in flight while row r's NW qsads issue
    uint32_t dA[NW + 1], dB[NW + 1], SA[NW], SB[NW];
    uint64_t acc16 = 0;
#define SEA_LOAD(r, d, S)                                                   \
    {                                                                       \
        const uint32_t *p_ = wrow + (r) * pitch;                            \
        _Pragma("unroll") for (int k = 0; k < NW + 1; k++) d[k] = p_[k];    \
        _Pragma("unroll") for (int k = 0; k < NW; k++) S[k] = crow[(r) * cstr + k]; \
    }
#define SEA_ROW(d, S)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < NW; k++)                                                     \
        acc16 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 1] << 32) | d[k], S[k], acc16);
    SEA_LOAD(0, dA, SA)
#pragma unroll
    for (int r = 0; r < RQ; r += 2) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): row r has landed ...
        SEA_LOAD(r + 1, dB, SB)              // ... row r+1 goes out before row r's qsads
        __builtin_amdgcn_sched_barrier(0);
        SEA_ROW(dA, SA)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (r + 2 < RQ) SEA_LOAD(r + 2, dA, SA)
        __builtin_amdgcn_sched_barrier(0);
        SEA_ROW(dB, SB)
        __builtin_amdgcn_sched_barrier(0);
        if ((r & 3) == 2) {  // every 4 rows: at most 64 qsads * 1020 < 2^16
#pragma unroll
            for (int q = 0; q < 4; q++) acc32[q] += (uint32_t)(acc16 >> (16 * q)) & 0xffffu;
            acc16 = 0;
        }
    }
    if constexpr (RQ % 4 != 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc32[q] += (uint32_t)(acc16 >> (16 * q)) & 0xffffu;
    }
#undef SEA_LOAD
#undef SEA_ROW
    *reinterpret_cast<uint4 *>(dst) = make_uint4(acc32[0], acc32[1], acc32[2], acc32[3]);
}

// ---------------------------------------------------------------------------------------------
// Kernel 2: exact SADs.  Window in LDS.  Everything runs in HALF ROUNDS: a lane = one group (4 dx at one dy) x
// B / 4 rows, a pair of waves = 64 groups x the two row quarters of ONE half of the block, so the WAVES waves
// take 32 x WAVES groups' upper (or lower) halves at a time; the even wave of a pair adds the two quarters up
// (double-buffered through LDS: one barrier per half round).
//
// A partial SAD is a lower bound of the whole one.  A group whose legal members all exceed the best SAD found
// so far on the UPPER half of the block alone is dropped there -- exactly as a bound drops it: no winner and no
// tie is lost.  The others are kept (group + four partial SADs, 18 bytes, in LDS) and finished by lower-half
// rounds.
//
// Pass 0 = the first batch (list0; its entries are in registers when the window has landed).  The search starts
// from the exact SAD the bounds kernel's probe measured (a real candidate's key).  Upper halves of the whole
// batch; then the lower halves of tier 1 = the groups within an eighth of the smallest partial SAD (on typical
// content: the true match and little else), which gives a tight best SAD; then of tier 2 = whatever of the
// rest still has a partial SAD under it (typically nothing).  Pass 1, only if a group outside the batch has a
// bound <= the best SAD (hdr.rest_min tells without a scan): one sweep over the bounds builds the schedule, half
// rounds over it with the kept groups finished whenever half a round's worth has gathered.
//
// Ties: a group whose bound EQUALS the best SAD can at most tie it, and a tie is won by the lower raster index
// (the reference's strict `<` in raster order, src/encode.cpp:762-765): such a group is scheduled only if its
// first legal member comes before the current winner in raster order.
// ---------------------------------------------------------------------------------------------
template <bool kStats, int WAVES, int B>
__global__ __launch_bounds__(WAVES * 64, 4) void sea_exact_kernel(const uint8_t *__restrict__ cur_base,
                                                                   const uint8_t *__restrict__ ref_base,
                                                                   const uint16_t *__restrict__ gl,
                                                                   const uint16_t *__restrict__ list0,
                                                                   uint16_t *__restrict__ wg_list,
                                                                   const SeaHdr *__restrict__ hdr,
                                                                   const uint32_t *__restrict__ order,
                                                                   uint32_t *__restrict__ ctrl, int total,
                                                                   x265amd_mv *__restrict__ out, const MeGeom g,
                                                                   int gl_pitch, int win_dwords, int surv_off, int surv_cap,
                                                                   uint32_t *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *win = smem;                                                     // [win_rows][pitch] reference window
    constexpr int kThreads = WAVES * 64, kHalf = 32 * WAVES;  // groups of a half round: WAVES / 2 sets of 64
    static_assert(kFirstMax <= 32 * 16 && kBoundsThreads >= kFirstMax, "the first batch is one half round; kernel 1 deals it a thread per entry");
    // second-pass schedule: in LDS behind the window (8 waves), or in this workgroup's own slice of the
    // workspace when the LDS is needed for a +-128 window and 16 waves' partial sums (same-CU L1: coherent
    // across the barrier for the waves of one workgroup)
    uint16_t *list = WAVES == 8 ? reinterpret_cast<uint16_t *>(smem + win_dwords) : wg_list + (size_t)blockIdx.x * gl_pitch;
    uint32_t *surv_sum = smem + surv_off;                                          // [surv_cap][4] upper-half SADs of the kept groups
    uint16_t *surv_ent = reinterpret_cast<uint16_t *>(surv_sum + 4 * surv_cap);    // [surv_cap] the groups
    __shared__ __attribute__((aligned(16))) uint32_t part[2][WAVES * 64 * 4];  // partial SADs: [round & 1][wave][lane][4 dx]
    __shared__ unsigned long long best_key;
    __shared__ int list_n, next_ctu, surv_n;
    __shared__ uint32_t u_min;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: keeps the current rows in SGPRs
    const int hset = wave >> 1, hq = wave & 1;                  // 64 groups a set; the two quarters of a half
    const int R = g.range;
    const int span = 2 * R + 1;
    const int cstr = g.stride >> 2;
    const int per_plane = g.ctus_x * g.ctus_y;
    int round = 0;

    // Persistent workgroups: each takes the next CTU of the launch order (ctrl[kClasses] counts them out) until
    // none is left -- every wave reaches that exit.  The index of the NEXT CTU is fetched while this
    // one is being searched.
    uint32_t cls_end[kClasses];  // running ends of the class queues in hand-out order (final: kernel 1 is done)
    {
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < kClasses; c++) {
            acc += ctrl[c];
            cls_end[c] = acc;
        }
    }
    const int nitems = (int)cls_end[kClasses - 1];  // CTUs queued for this kernel (the others went to the every-candidate list)
    auto item_ctu = [&](int it) __attribute__((always_inline)) -> uint32_t {  // it < nitems
        int c = 0;
        uint32_t beg = 0;
#pragma unroll
        for (int k = 0; k < kClasses - 1; k++)
            if ((uint32_t)it >= cls_end[k]) {
                c = k + 1;
                beg = cls_end[k];
            }
        return order[(size_t)c * total + ((uint32_t)it - beg)];
    };
    int it_now = 0;  // (thread 0) position of the current CTU in the hand-out order
    if (tid == 0) {
        it_now = (int)atomicAdd(&ctrl[kClasses], 1u);
        next_ctu = it_now < nitems ? (int)item_ctu(it_now) : -1;
    }
    __syncthreads();
    int ctu_now = next_ctu;
  while (ctu_now >= 0) {
    // The next CTU's index is fetched while this one is searched -- but only when this one is light.  A
    // workgroup that reserves its next CTU while it starts on one of the two heaviest classes may reserve
    // another of them (the queues' heads go out within microseconds of the launch) and run the two
    // back to back.
    int it_next = 0;
    uint32_t ctu_next_v = 0;
    const bool reserve = tid == 0 && (uint32_t)it_now >= cls_end[1];
    if (reserve) {
        it_next = (int)atomicAdd(&ctrl[kClasses], 1u);
        ctu_next_v = item_ctu(it_next < nitems ? it_next : 0);
    }
    const int64_t ctu_id = ctu_now;
    const int plane = ctu_now / per_plane, cin = ctu_now - plane * per_plane;
    const int cty = cin / g.ctus_x, ctx = cin - cty * g.ctus_x;
    const int f = plane / g.nrefs, rr = plane - f * g.nrefs;
    const int cx = ctx * B, cy = cty * B;
    const SeaLayout L = sea_layout(g.width, g.height, R, cx, cy, B);
    const uint8_t *cur = cur_base + f * g.cur_fs;
    const uint8_t *ref = ref_base + f * g.ref_fs + rr * g.ref_rs;
    const uint16_t *glo = gl + ctu_id * gl_pitch;
    const uint16_t *l0 = list0 + ctu_id * gl_pitch;

    // the first batch's entry of this lane and the header: in flight with the window
    const long long t_begin = kStats ? clock64() : 0;
    const int slot0 = hset * 64 + lane;
    const int ent0 = (int)l0[min(slot0, gl_pitch - 1)];
    const SeaHdr h = hdr[ctu_id];

    // ---- stage the reference window: a wave per row, a lane per dword, 13 rows in flight ----
    constexpr int kWinInFlight = 13;
    if (WAVES == 16 && L.pitch > 64) {  // (the 8-wave shape never sees such a window: it would not fit twice into the LDS)
        // a row wider than a wave (+-128: 83 dwords): the window as ONE run of 16-byte chunks (21 a row) dealt over the
        // whole workgroup -- seven loads a lane, all in flight at once: one memory latency for the window.  This shape has
        // the compute unit to itself, so nothing else runs under its staging.  Rows start on dword boundaries only, which a
        // four-dword load takes; a chunk that reaches past the right edge of the picture goes dword by dword.
        struct __attribute__((packed, aligned(4))) Dw4 {
            uint32_t v[4];
        };
        constexpr int kChunks = 7;
        const int nch = (L.pitch + 3) >> 2, total_ch = L.win_rows * nch;
        const uint32_t recip = 0xffffffffu / (uint32_t)nch + 1u;  // e / nch = (e * recip) >> 32 for e < 2^16 x nch
        for (int e0 = tid; e0 < total_ch; e0 += kThreads * kChunks) {
            Dw4 v[kChunks];
#pragma unroll
            for (int j = 0; j < kChunks; j++) {
                const int e = e0 + kThreads * j;
                const int row = (int)__umulhi((uint32_t)e, recip), ch = e - row * nch;
                const int gx = L.wx0 + 16 * ch, gy = cy + L.dylo + row;
                v[j] = Dw4{{0u, 0u, 0u, 0u}};
                if (e < total_ch && gy < g.height) {
                    const uint8_t *p = ref + (int64_t)gy * g.stride + gx;
                    if (gx + 16 <= g.width) {
                        v[j] = *reinterpret_cast<const Dw4 *>(p);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (gx + 4 * k + 4 <= g.width) v[j].v[k] = *reinterpret_cast<const uint32_t *>(p + 4 * k);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kChunks; j++) {
                const int e = e0 + kThreads * j;
                const int row = (int)__umulhi((uint32_t)e, recip), ch = e - row * nch;
                if (e < total_ch) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (4 * ch + k < L.pitch) win[row * L.pitch + 4 * ch + k] = v[j].v[k];
                }
            }
        }
    } else if (WAVES == 8 && L.pitch <= 64) {
        // a row is at most 64 dwords: 16 bytes a lane, several rows per load instruction of a wave (13 chunks a row
        // at +-64: four rows), and all of a wave's instructions in flight at once -- the whole window for ONE memory
        // latency (a lane per dword took two batches of 13 rows).  Rows start on dword boundaries only, which a
        // four-dword load takes; a chunk that reaches past the right edge of the picture goes dword by dword.
        struct __attribute__((packed, aligned(4))) Dw4 {
            uint32_t v[4];
        };
        const int nch = (L.pitch + 3) >> 2, rpi = 64 / nch;  // chunks per row (<= 16), rows per instruction (>= 4)
        const int sub = lane / nch, ch = lane - sub * nch;
        const bool lane_on = sub < rpi;
        const int gx = L.wx0 + 16 * ch;
        constexpr int kChunksInFlight = 6;
        for (int r0 = wave * rpi; r0 < L.win_rows; r0 += WAVES * rpi * kChunksInFlight) {
            Dw4 v[kChunksInFlight];
#pragma unroll
            for (int j = 0; j < kChunksInFlight; j++) {
                const int row = r0 + WAVES * rpi * j + sub, gy = cy + L.dylo + row;
                v[j] = Dw4{{0u, 0u, 0u, 0u}};  // outside the picture: only ever paired with candidates that are masked below
                if (lane_on && row < L.win_rows && gy < g.height) {
                    const uint8_t *p = ref + (int64_t)gy * g.stride + gx;
                    if (gx + 16 <= g.width) {
                        v[j] = *reinterpret_cast<const Dw4 *>(p);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++)
                            if (gx + 4 * k + 4 <= g.width) v[j].v[k] = *reinterpret_cast<const uint32_t *>(p + 4 * k);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kChunksInFlight; j++) {
                const int row = r0 + WAVES * rpi * j + sub;
                if (lane_on && row < L.win_rows) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (4 * ch + k < L.pitch) win[row * L.pitch + 4 * ch + k] = v[j].v[k];
                }
            }
        }
    } else
    for (int qb = 0; qb < L.pitch; qb += 64) {
        const int wq = qb + lane;
        const int gx = L.wx0 + 4 * wq;
        const bool colok = wq < L.pitch;
        const bool inpic = colok && gx + 4 <= g.width;
        for (int row0 = wave; row0 < L.win_rows; row0 += WAVES * kWinInFlight) {
            uint32_t v[kWinInFlight];
#pragma unroll
            for (int j = 0; j < kWinInFlight; j++) {
                const int row = row0 + WAVES * j, gy = cy + L.dylo + row;
                v[j] = 0;  // outside the picture: only ever paired with candidates that are masked below
                if (inpic && row < L.win_rows && gy < g.height)
                    v[j] = *reinterpret_cast<const uint32_t *>(ref + (int64_t)gy * g.stride + gx);
            }
#pragma unroll
            for (int j = 0; j < kWinInFlight; j++) {
                const int row = row0 + WAVES * j;
                if (colok && row < L.win_rows) win[row * L.pitch + wq] = v[j];
            }
        }
    }
    if (tid == 0) {
        best_key = ((unsigned long long)h.probe_sad << 32) | h.probe_idx;  // the probe's candidate: evaluated exactly by kernel 1
        list_n = 0;
        surv_n = 0;
        u_min = 0xffffffffu;
    }
    const int n0 = __builtin_amdgcn_readfirstlane((int)(h.n0_rest & 0xffffu));
    const uint32_t rest_min = __builtin_amdgcn_readfirstlane(h.n0_rest >> 16);
    __syncthreads();

    const long long t_staged = kStats ? clock64() : 0;
    const uint32_t *__restrict__ cbase = reinterpret_cast<const uint32_t *>(cur + (int64_t)cy * g.stride + cx);

    // this wave's part of a half round (sea_half_partial): the B / 4 rows from `row0` on of the lanes' groups
    auto half_partial = [&](int ent, bool have, int row0) __attribute__((always_inline)) {
        sea_half_partial<B, B / 4>(win, L.pitch, L.ngx, cbase, cstr, part[round & 1] + (wave * 64 + lane) * 4, ent, have, row0);
    };
    // the two quarters of a half that the waves `wave` and `wave + 1` left in part[]
    auto half_sum = [&]() __attribute__((always_inline)) -> uint4 {
        const uint32_t *pbuf = part[round & 1];
        const uint4 a = *reinterpret_cast<const uint4 *>(&pbuf[(wave * 64 + lane) * 4]);
        const uint4 b = *reinterpret_cast<const uint4 *>(&pbuf[((wave + 1) * 64 + lane) * 4]);
        return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    };
    // smallest partial SAD among the legal members of a group
    auto legal_min = [&](int ent, const uint4 s) __attribute__((always_inline)) -> uint32_t {
        const int dyi = ent / L.ngx, gx = ent - dyi * L.ngx;
        const uint32_t sad[4] = {s.x, s.y, s.z, s.w};
        uint32_t mn = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int dx = L.wx0 - cx + 4 * gx + q;
            if (dx >= L.dxlo && dx <= L.dxhi) mn = min(mn, sad[q]);
        }
        return mn;
    };
    // even waves: append the lanes with `keep` (group + upper-half SADs) to the kept list
    auto keep_append = [&](bool keep, int ent, const uint4 sm) __attribute__((always_inline)) {
        const unsigned long long km = __ballot(keep);
        if (km != 0ull) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&surv_n, __popcll(km));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (keep) {
                const int slot = wbase + __popcll(km & ((1ull << lane) - 1ull));
                surv_ent[slot] = (uint16_t)ent;
                *reinterpret_cast<uint4 *>(&surv_sum[4 * slot]) = sm;
            }
        }
    };
    // lower halves of the kept groups [from, to): their SADs are complete, the winner so far goes to best_key.
    // Block-uniform arguments; the caller's next barrier publishes best_key.
    auto lower_rounds = [&](int from, int to) __attribute__((always_inline)) {
        for (int kb = from; kb < to; kb += kHalf) {
            const int li = kb + hset * 64 + lane;
            const bool have = li < to;
            const int ent = have ? (int)surv_ent[li] : 0;
            half_partial(ent, have, B / 2 + (B / 4) * hq);
            __syncthreads();
            if (hq == 0 && __ballot(have) != 0ull) {
                unsigned long long k = ~0ull;
                if (have) {
                    const int dyi = ent / L.ngx, gx = ent - dyi * L.ngx, dy = L.dylo + dyi;
                    uint4 sm = half_sum();
                    const uint4 up = *reinterpret_cast<const uint4 *>(&surv_sum[4 * li]);
                    sm = make_uint4(sm.x + up.x, sm.y + up.y, sm.z + up.z, sm.w + up.w);
                    const uint32_t sad[4] = {sm.x, sm.y, sm.z, sm.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int dx = L.wx0 - cx + 4 * gx + q;
                        if (dx >= L.dxlo && dx <= L.dxhi) {
                            const unsigned long long key = ((unsigned long long)sad[q] << 32) | (unsigned)((dy + R) * span + dx + R);
                            k = key < k ? key : k;
                        }
                    }
                }
                k = me_wave_min_u64(k);
                if (lane == 0 && k != ~0ull) atomicMin(&best_key, k);
            }
            round++;  // the next half round writes the other half of part[]: no second barrier
        }
    };

    // ---- at most 64 groups: one set of lanes, so the ROWS are split over eight waves instead -- a wave takes an
    //      eighth of them, wave 0 adds the eight partial sums up.  Same arithmetic, a quarter (lower halves) or half
    //      (whole blocks) of the latency of a half round in which six of eight waves would stand by.
    //      kLower: the lower halves of kept groups (surv_ent[from .. to), + their upper-half SADs);
    //      else whole blocks of schedule entries list[from .. to). ----
    auto split_round = [&](auto lower_c, int from, int to) __attribute__((always_inline)) {
        constexpr bool kLower = decltype(lower_c)::value;
        constexpr int NR = kLower ? B / 16 : B / 8;  // rows per wave
        const int li = from + lane;
        const bool have = li < to;
        const int ent = have ? (int)(kLower ? surv_ent[li] : list[li]) : 0;
        if (wave < 8)
            sea_half_partial<B, NR>(win, L.pitch, L.ngx, cbase, cstr, part[round & 1] + (wave * 64 + lane) * 4, ent, have,
                                    (kLower ? B / 2 : 0) + NR * wave);
        __syncthreads();
        if (wave == 0) {
            unsigned long long k = ~0ull;
            if (have) {
                const int dyi = ent / L.ngx, gx = ent - dyi * L.ngx, dy = L.dylo + dyi;
                uint4 sm = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (kLower) sm = *reinterpret_cast<const uint4 *>(&surv_sum[4 * li]);
                const uint32_t *pbuf = part[round & 1];
#pragma unroll
                for (int w8 = 0; w8 < 8; w8++) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(&pbuf[(w8 * 64 + lane) * 4]);
                    sm = make_uint4(sm.x + t.x, sm.y + t.y, sm.z + t.z, sm.w + t.w);
                }
                const uint32_t sad[4] = {sm.x, sm.y, sm.z, sm.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int dx = L.wx0 - cx + 4 * gx + q;
                    if (dx >= L.dxlo && dx <= L.dxhi) {
                        const unsigned long long key = ((unsigned long long)sad[q] << 32) | (unsigned)((dy + R) * span + dx + R);
                        k = key < k ? key : k;
                    }
                }
            }
            k = me_wave_min_u64(k);
            if (lane == 0 && k != ~0ull) atomicMin(&best_key, k);
        }
        round++;
    };
    using SeaLowerHalves = std::integral_constant<bool, true>;
    using SeaWholeBlocks = std::integral_constant<bool, false>;

    // ---- a handful of groups: a LANE PER ROW.  A wave takes one group (whole blocks of 64 rows) or 64 / rows of them, a
    //      lane one row of it: NW window dwords + 1 from LDS, the row of the current block from memory (16 bytes a
    //      load; the block is in the caches from the rounds before), NW qsads, and the rows are added up across the
    //      lanes by DPP.  The split round above costs eight (four) rows' worth of qsads at a quarter of the rate with one
    //      lane in 64 at work -- 11 000 (5 000) cycles for the one or two groups that tier 1 or a light second pass
    //      brings; this costs one row's worth.  Same arguments and barriers as split_round; part[] is not touched. ----
    auto rows_round = [&](auto lower_c, int from, int to) __attribute__((always_inline)) {
        constexpr bool kLower = decltype(lower_c)::value;
        constexpr int RW = kLower ? B / 2 : B;  // rows = lanes per group: 64, 32 or 16
        constexpr int GPW = 64 / RW;            // groups per wave
        constexpr int NW = B / 4;
        struct __attribute__((packed, aligned(4))) Dw4 {
            uint32_t v[4];
        };
        const int sub = lane / RW, rin = lane - sub * RW, row = (kLower ? B / 2 : 0) + rin;
        for (int g0 = from; g0 < to; g0 += WAVES * GPW) {  // block-uniform bounds
            const int li = g0 + wave * GPW + sub;
            const bool have = li < to;
            const int ent = have ? (int)(kLower ? surv_ent[li] : list[li]) : 0;
            const int dyi = ent / L.ngx, gx = ent - dyi * L.ngx;
            const uint32_t *wrow = win + (dyi + row) * L.pitch + gx;
            const Dw4 *crow = reinterpret_cast<const Dw4 *>(cbase + row * cstr);
            uint32_t d[NW + 1], c[NW];
#pragma unroll
            for (int k = 0; k < NW / 4; k++) {
                const Dw4 t = crow[k];
#pragma unroll
                for (int j = 0; j < 4; j++) c[4 * k + j] = t.v[j];
            }
#pragma unroll
            for (int k = 0; k < NW + 1; k++) d[k] = wrow[k];
            uint64_t acc = 0;  // a row's four SADs: at most 64 * 255 each
#pragma unroll
            for (int k = 0; k < NW; k++) acc = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 1] << 32) | d[k], c[k], acc);
            uint32_t sad[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t v = (uint32_t)(acc >> (16 * q)) & 0xffffu;
                v += me_dpp<0xb1>(v);
                v += me_dpp<0x4e>(v);
                v += me_dpp<0x141>(v);
                v += me_dpp<0x140>(v);  // the sum of the lane's row of 16 lanes, in each of them
                if constexpr (RW > 16) {
                    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
                    v = RW == 64 ? r0 + r1 + r2 + r3 : (lane < 32 ? r0 + r1 : r2 + r3);
                }
                sad[q] = v;
            }
            if (have && rin == 0) {  // the group's first lane: its winner so far
                if constexpr (kLower) {
                    const uint4 up = *reinterpret_cast<const uint4 *>(&surv_sum[4 * li]);
                    sad[0] += up.x, sad[1] += up.y, sad[2] += up.z, sad[3] += up.w;
                }
                const int dy = L.dylo + dyi;
                unsigned long long k = ~0ull;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int dx = L.wx0 - cx + 4 * gx + q;
                    if (dx >= L.dxlo && dx <= L.dxhi) {
                        const unsigned long long key = ((unsigned long long)sad[q] << 32) | (unsigned)((dy + R) * span + dx + R);
                        k = key < k ? key : k;
                    }
                }
                if (k != ~0ull) atomicMin(&best_key, k);
            }
        }
    };
    // up to this many groups go a lane per row (whole blocks / lower halves); above: split_round, above 64: half rounds
    // (two sweeps of the workgroup; four or eight measured the same on the headline and no better in the closed loop)
    constexpr int kRowsWhole = 2 * WAVES, kRowsLower = 2 * WAVES * (B >= 64 ? 2 : 4);

    // ---- pass 0: the first batch.  Upper halves of all of it ... ----
    const bool have0 = slot0 < n0;
    half_partial(ent0, have0, (B / 4) * hq);
    __syncthreads();
    uint4 p0 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t pmin0 = 0xffffffffu;
    if (hq == 0 && __ballot(have0) != 0ull) {
        if (have0) {
            p0 = half_sum();
            pmin0 = legal_min(ent0, p0);
        }
        const uint32_t wm = me_wave_min_u32(pmin0);
        if (lane == 0) atomicMin(&u_min, wm);
    }
    round++;
    __syncthreads();
    // ... then the lower halves, in two tiers.  Tier 1: within an eighth of the smallest partial SAD (and not beyond
    // the probe's SAD: such a group is out for good); tier 2: what else of the batch is still under the best SAD.
    const uint32_t um = __builtin_amdgcn_readfirstlane(u_min);
    const uint32_t t1 = um > 0xf0000000u ? um : um + (um >> 3) + 64u;
    bool taken = false;
    int nk2 = 0;
#pragma unroll 1
    for (int tier = 0; tier < 2; tier++) {
        const uint32_t bsf = __builtin_amdgcn_readfirstlane((uint32_t)(best_key >> 32));
        const uint32_t thr = tier == 0 ? min(t1, bsf) : bsf;
        const bool keep = hq == 0 && have0 && !taken && pmin0 <= thr;
        taken = taken || keep;
        if (hq == 0) keep_append(keep, ent0, p0);
        __syncthreads();
        const int nk = surv_n;
        if (nk - nk2 > 64)
            lower_rounds(nk2, nk);
        else if (nk - nk2 > kRowsLower)
            split_round(SeaLowerHalves{}, nk2, nk);
        else if (nk > nk2)
            rows_round(SeaLowerHalves{}, nk2, nk);
        nk2 = nk;
        __syncthreads();
    }

    const long long t_pass0 = kStats ? clock64() : 0;
    // ---- pass 1, only if some group left over has a bound <= the best SAD so far ----
    int n1 = 0;
    [[maybe_unused]] int n_kept = 0;  // (statistics) groups of the second pass that outlived their upper half
    const unsigned long long bk0 = best_key;
    const uint32_t best_sad = (uint32_t)(bk0 >> 32), best_idx = (uint32_t)bk0;
    if (rest_min != kNone && (rest_min << 4) <= best_sad) {
        // one sweep: a thread takes 2 x 8 consecutive bounds (two 16-byte loads, all of a CTU's loads in flight at
        // once); a wave without a taker is done; in the others a wave scan ranks the takers and ONE LDS atomic per
        // wave reserves their slots
        const int nvec = (L.ngroups + 7) >> 3;  // gl rows are 16-byte aligned and padded to a multiple of 8
        auto takers = [&](const uint4 wv, int vi) __attribute__((always_inline)) -> uint32_t {
            const uint32_t w[4] = {wv.x, wv.y, wv.z, wv.w};
            uint32_t tm = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t v = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                if (8 * vi + e < L.ngroups && v != kNone && (v << 4) <= best_sad) {
                    bool take = true;
                    if ((v << 4) == best_sad) {  // can only tie: matters if it comes first in raster order
                        const int gi = 8 * vi + e, dyi = gi / L.ngx, gx = gi - dyi * L.ngx;
                        const int dx = max(L.wx0 - cx + 4 * gx, L.dxlo);
                        take = (uint32_t)((L.dylo + dyi + R) * span + dx + R) < best_idx;
                    }
                    if (take) tm |= 1u << e;
                }
            }
            return tm;
        };
        for (int v0 = 0; v0 < nvec; v0 += 2 * kThreads) {
            const int vi0 = v0 + tid, vi1 = vi0 + kThreads;
            uint4 wv0 = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu), wv1 = wv0;
            if (vi0 < nvec) wv0 = reinterpret_cast<const uint4 *>(glo)[vi0];
            if (vi1 < nvec) wv1 = reinterpret_cast<const uint4 *>(glo)[vi1];
            const uint32_t tm0 = takers(wv0, vi0), tm1 = takers(wv1, vi1);
            if (__ballot((tm0 | tm1) != 0u) == 0ull) continue;  // wave-uniform
            const int cnt = __popc(tm0) + __popc(tm1);
            const int inc = (int)me_wave_scan_u32((uint32_t)cnt, lane);
            const int tot = __builtin_amdgcn_readlane(inc, 63);
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&list_n, tot);
            int slot = __builtin_amdgcn_readfirstlane(wbase) + inc - cnt;
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (tm0 & (1u << e)) list[slot++] = (uint16_t)(8 * vi0 + e);
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (tm1 & (1u << e)) list[slot++] = (uint16_t)(8 * vi1 + e);
        }
        if (tid == 0) surv_n = 0;
        __syncthreads();
        n1 = list_n;
        // upper halves over the schedule, the groups that stay under the best SAD are kept; whenever the kept list
        // could not take another half round's worth -- and at the end -- lower halves over the kept groups finish
        // them.  A schedule of one set of lanes goes through whole, its rows split over the waves.
        int base = 0;
        if (n1 <= 64) {
            if (n1 <= kRowsWhole)
                rows_round(SeaWholeBlocks{}, 0, n1);
            else
                split_round(SeaWholeBlocks{}, 0, n1);
            base = n1;
            if constexpr (kStats) n_kept += n1;
            __syncthreads();
        }
        while (base < n1) {
            // the threshold follows the search: every lower-half round may have brought a better SAD
            const uint32_t cur_best = __builtin_amdgcn_readfirstlane((uint32_t)(best_key >> 32));
            const int li = base + hset * 64 + lane;
            const bool have = li < n1;
            const int ent = have ? (int)list[li] : 0;
            half_partial(ent, have, (B / 4) * hq);
            __syncthreads();
            if (hq == 0 && __ballot(have) != 0ull) {
                uint4 sm = make_uint4(0u, 0u, 0u, 0u);
                bool keep = false;
                if (have) {
                    sm = half_sum();
                    keep = legal_min(ent, sm) <= cur_best;
                }
                keep_append(keep, ent, sm);
            }
            round++;
            base += kHalf;
            __syncthreads();  // surv_n and the kept entries are visible; part[] may be written again
            const int kept = surv_n;
            // finish what is kept as soon as it fills half a round (a better best SAD prunes what follows), when
            // the list could not take another half round's worth, and at the end
            if (base >= n1 || kept >= kHalf / 2 || kept + kHalf > surv_cap) {
                if constexpr (kStats) n_kept += kept;
                if (kept > 64)
                    lower_rounds(0, kept);
                else if (kept > kRowsLower)
                    split_round(SeaLowerHalves{}, 0, kept);
                else if (kept > 0)
                    rows_round(SeaLowerHalves{}, 0, kept);
                __syncthreads();  // every wave is done with the kept list; best_key is visible
                if (tid == 0) surv_n = 0;
                __syncthreads();
            }
        }
    }
    if (tid == 0) {
        const unsigned long long best = best_key;
        x265amd_mv mv;
        const int c = (int)(best & 0xffffffffu);
        mv.mvx = (int16_t)(c % span - R);
        mv.mvy = (int16_t)(c / span - R);
        mv.sad = (uint32_t)(best >> 32);
        out[ctu_id] = mv;
        if constexpr (kStats) {
            stats[16 * ctu_id + 0] = (uint32_t)n0;
            stats[16 * ctu_id + 1] = (uint32_t)n1;
            stats[16 * ctu_id + 2] = (uint32_t)(t_staged - t_begin);  // cycles: window staging
            stats[16 * ctu_id + 3] = (uint32_t)(t_pass0 - t_staged);  //         first batch
            const long long t_end = clock64();
            stats[16 * ctu_id + 4] = (uint32_t)(t_end - t_pass0);     //         second pass
            stats[16 * ctu_id + 5] = (uint32_t)t_begin;               // absolute, low 32 bits
            stats[16 * ctu_id + 6] = (uint32_t)t_end;
            stats[16 * ctu_id + 8] = h.est;
            stats[16 * ctu_id + 14] = (uint32_t)n_kept;               // lower halves of the second pass
            stats[16 * ctu_id + 15] = (uint32_t)nk2;                  // lower halves of the first batch (tier 1 + tier 2)
            stats[16 * ctu_id + 9] = (uint32_t)blockIdx.x;
            stats[16 * ctu_id + 7] = (__builtin_amdgcn_s_getreg(63492) & 0xffffu) | (__builtin_amdgcn_s_getreg(6164) << 16);  // HW_ID | XCC_ID
        }
        if (!reserve) {  // a heavy CTU: ask only now that it is done
            it_next = (int)atomicAdd(&ctrl[kClasses], 1u);
            ctu_next_v = item_ctu(it_next < nitems ? it_next : 0);
        }
        next_ctu = it_next < nitems ? (int)ctu_next_v : -1;
        it_now = it_next;
    }
    __syncthreads();  // the window and part[] are free again; next_ctu is visible
    ctu_now = next_ctu;
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel 3: every candidate's SAD (me_qsad_block, the search of x265amd_me_fullsearch) for the blocks kernel 1 put
// on its list.  One workgroup per (block, slice) item; the grid is sized for the case that EVERY block is listed, and
// a workgroup beyond the items exits at once (all of them, when the list is empty).  A short list is cut into slices
// of candidate rows, as many as the grid has room for (at most 8); slices of a block meet in a 64-bit atomic minimum
// on the key (in the block's own, otherwise unused, row of bounds) and the last one to arrive writes the vector out.
// ---------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(256, 2) void me_list_qsad(const uint8_t *__restrict__ cur_base, const uint8_t *__restrict__ ref_base,
                                                       x265amd_mv *__restrict__ out, const MeGeom g,
                                                       const uint32_t *__restrict__ blist, const uint32_t *__restrict__ ctrl,
                                                       uint16_t *__restrict__ gl, int gl_pitch, const SeaHdr *__restrict__ hdr)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int n = (int)ctrl[kClasses + 1];
    if (n == 0) return;  // grid-uniform
    const int nsl = min((int)gridDim.x / n, 8);  // >= 1: the grid has a workgroup per block
    int ctu, slice = 0;
    if (nsl == 1) {
        // More than half of the blocks are listed: a workgroup per BLOCK, in the XCD-aware order of the other kernels (XCD k
        // takes the k-th contiguous eighth of the blocks: neighbours share most of their windows, and the list's own order
        // -- the order in which kernel 1's workgroups finished -- scatters them over the L2s), the unlisted ones exit.
        const int total = (int)gridDim.x, lin = (int)blockIdx.x;
        const int xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
        ctu = xcd * base + min(xcd, rem) + slot;
        if (hdr[ctu].n0_rest != kRouted) return;  // block-uniform
    } else {
        const int item = blockIdx.x;
        if (item >= n * nsl) return;  // block-uniform
        ctu = (int)blist[item / nsl], slice = item % nsl;
    }
    const int per_plane = g.ctus_x * g.ctus_y;
    const int plane = ctu / per_plane, cin = ctu - plane * per_plane;
    const int cty = cin / g.ctus_x, ctx = cin - cty * g.ctus_x;
    const int f = plane / g.nrefs, rr = plane - f * g.nrefs;
    const unsigned long long best = me_qsad_block<B>(smem, cur_base + f * g.cur_fs, ref_base + f * g.ref_fs + rr * g.ref_rs, g,
                                                     ctx * B, cty * B, slice, nsl);
    if (threadIdx.x == 0) {
        if (nsl == 1) {
            out[ctu] = me_key_to_mv(best, g.range);
        } else {
            unsigned long long *bk = reinterpret_cast<unsigned long long *>(gl + (size_t)ctu * gl_pitch);
            atomicMin(bk, best);
            __threadfence();
            if (atomicAdd(reinterpret_cast<uint32_t *>(bk) + 2, 1u) == (uint32_t)(nsl - 1))  // the last slice of the block
                out[ctu] = me_key_to_mv(atomicMin(bk, ~0ull), g.range);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Static scratch for x265amd_me_fullsearch_sea_static: the workspace of the search (pair plane, bounds,
// first-batch lists, headers, queues, second-pass schedules) for callers that cannot allocate.  Every GPU
// the library runs on holds its own copy of this array, at its own address, until the process ends;
// host code never uses the array's name as an address -- only the pointer hipGetSymbolAddress returned
// for that device (DeviceState::sea_scratch).
// ---------------------------------------------------------------------------------------------
constexpr size_t kSeaStaticBytes = (size_t)X265AMD_ME_STATIC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char sea_scratch[kSeaStaticBytes];

int me_sea_device_setup(DeviceState &ds)
{
    const void *fns[] = {(const void *)sea_bounds_kernel<64, kBoundsThreads>, (const void *)sea_exact_kernel<false, 8, 64>,
                         (const void *)sea_exact_kernel<true, 8, 64>, (const void *)sea_exact_kernel<false, 16, 64>,
                         (const void *)sea_exact_kernel<true, 16, 64>,
                         (const void *)sea_bounds_kernel<32, kBoundsThreads>, (const void *)sea_bounds_kernel<32, 256>,
                         (const void *)sea_exact_kernel<false, 8, 32>,
                         (const void *)sea_exact_kernel<true, 8, 32>, (const void *)sea_exact_kernel<false, 16, 32>,
                         (const void *)sea_exact_kernel<true, 16, 32>, (const void *)me_list_qsad<64>, (const void *)me_list_qsad<32>};
    for (const void *fn : fns) {
        const int st = allow_full_lds(fn, ds);
        if (st != X265AMD_OK) return st;
    }
    void *p = nullptr;
    X265AMD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(sea_scratch)));  // this device's copy of the array
    if (!p || ((uintptr_t)p & 255)) return X265AMD_ERR_HIP - (int)hipErrorInvalidSymbol;
    ds.sea_scratch = p;
    ds.sea_scratch_bytes = kSeaStaticBytes;
    return X265AMD_OK;
}

}  // namespace x265amd

using namespace x265amd;

// ---- debugging aids (include/x265_amd_debug.h): state of the device current on the calling thread ----
// 16 u32 per CTU; [0] / [1] = groups evaluated in pass 0 / 1; switches both kernels to their stats variants
extern "C" void x265amd_debug_set_me_stats(uint32_t *d_stats)
{
    if (device_state()) debug_state().me_stats = d_stats;
}

// per-kernel HIP-event timing of the last _sea call on its own stream.  bench.py uses it so that the
// roofline line can name each kernel's duration without a profiler attached.
extern "C" int x265amd_debug_me_timing(int enable)
{
    if (!device_state()) return X265AMD_ERR_NOT_INIT;
    DebugState &dbg = debug_state();
    if (enable && !dbg.me_ev[0])
        for (auto &e : dbg.me_ev) X265AMD_HIP(hipEventCreate(&e));
    dbg.me_timing = enable != 0;
    return X265AMD_OK;
}
// ms of box16_pairs, (nothing), sea_bounds_kernel, sea_exact_kernel of the last timed call (waits for it)
extern "C" int x265amd_debug_me_last_ms(float *out4)
{
    if (!device_state()) return X265AMD_ERR_NOT_INIT;
    DebugState &dbg = debug_state();
    if (!dbg.me_ev[0] || !out4) return X265AMD_ERR_ARG;
    X265AMD_HIP(hipEventSynchronize(dbg.me_ev[4]));
    for (int i = 0; i < 4; i++) X265AMD_HIP(hipEventElapsedTime(&out4[i], dbg.me_ev[i], dbg.me_ev[i + 1]));
    return X265AMD_OK;
}
// shape experiments of the exact kernel (A/B runs of tools/): extra dynamic LDS per workgroup, forced wave count
// (0 = the library's own choice)
extern "C" void x265amd_debug_sea_shape(int lds_pad_bytes, int waves)
{
    if (!device_state()) return;
    debug_state().sea_lds_pad = lds_pad_bytes > 0 ? lds_pad_bytes : 0;
    debug_state().sea_waves = waves;
}

// largest per-CTU geometry of a picture: rows / pitch of the window, groups per row, rows of candidates
static void sea_extents(int width, int height, int range, int ctu, int &max_rows, int &max_pitch, int &max_ngx, int &max_ndy)
{
    max_rows = max_pitch = max_ngx = max_ndy = 0;
    for (int cy = 0; cy < height; cy += ctu) {
        const SeaLayout L = sea_layout(width, height, range, 0, cy, ctu);
        max_rows = L.win_rows > max_rows ? L.win_rows : max_rows;
        max_ndy = L.ndy > max_ndy ? L.ndy : max_ndy;
    }
    for (int cx = 0; cx < width; cx += ctu) {
        const SeaLayout L = sea_layout(width, height, range, cx, 0, ctu);
        max_pitch = L.pitch > max_pitch ? L.pitch : max_pitch;
        max_ngx = L.ngx > max_ngx ? L.ngx : max_ngx;
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t x265amd_me_workspace_bytes_ctu(int width, int height, int ctu, int range, int nframes, int nrefs)
{
    if ((ctu != 64 && ctu != 32) || width <= 0 || height <= 0 || range < 0 || nframes <= 0 || nrefs <= 0 || width % ctu || height % ctu)
        return 0;
    int max_rows, max_pitch, max_ngx, max_ndy;
    sea_extents(width, height, range, ctu, max_rows, max_pitch, max_ngx, max_ndy);
    const size_t planes = (size_t)nframes * nrefs, nctu = (size_t)(width / ctu) * (height / ctu);
    const size_t per_ctu = ((size_t)max_ngx * max_ndy + 7) & ~(size_t)7;  // 16-byte rows
    return align256(4 * (size_t)width * height * planes)                    // pair plane (u32) per (frame, ref) plane
           + 2 * align256(sizeof(uint16_t) * per_ctu * nctu * planes)      // bounds + first-batch list per CTU
           + align256(sizeof(SeaHdr) * nctu * planes)
           + align256(sizeof(uint32_t) * (kClasses + 1) * nctu * planes)   // launch-order queues of the exact kernel + the every-candidate list
           + 256                                                           // their counters
           + align256(sizeof(uint16_t) * per_ctu * kMaxWgLists);           // second-pass schedules of 16-wave workgroups
}

extern "C" size_t x265amd_me_workspace_bytes(int width, int height, int range, int nframes, int nrefs)
{
    return x265amd_me_workspace_bytes_ctu(width, height, 64, range, nframes, nrefs);
}

// workgroups of one exact-kernel variant that a CU holds at this dynamic-LDS size: asked once per device,
// variant and size (a host-side computation, but not a free one), then served from the device's cache
static int exact_per_cu(DeviceState &ds, const void *fn, int threads, size_t lds, int &per_cu)
{
    std::lock_guard<std::mutex> lk(state_mutex());
    for (int i = 0; i < ds.nocc; i++)
        if (ds.occ[i].fn == fn && ds.occ[i].lds == lds) {
            per_cu = ds.occ[i].per_cu;
            return X265AMD_OK;
        }
    int q = 0;
    X265AMD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, threads, lds));
    OccEntry &e = ds.occ[ds.nocc < kOccSlots ? ds.nocc++ : (kOccSlots - 1)];  // full: the last slot is recycled
    e.fn = fn;
    e.lds = lds;
    e.per_cu = q;
    per_cu = q;
    return X265AMD_OK;
}

// as many workgroups as the chip holds at once (LDS- and register-bound), each looping over CTUs
template <int WAVES, int B>
static int launch_exact(DeviceState &ds, size_t lds, hipStream_t st, const uint8_t *d_cur, const uint8_t *d_ref,
                        const uint16_t *gl, const uint16_t *list0, uint16_t *wg_list, const SeaHdr *hdr,
                        const uint32_t *order, uint32_t *ctrl, int total, x265amd_mv *d_out, const MeGeom &g, int gl_pitch,
                        int win_dwords, int surv_off, int surv_cap)
{
    uint32_t *const stats = ds.dbg.me_stats;  // read once: the variant and its argument must agree
    const void *fn = stats ? (const void *)sea_exact_kernel<true, WAVES, B> : (const void *)sea_exact_kernel<false, WAVES, B>;
    int per_cu = 0;
    const int ost = exact_per_cu(ds, fn, WAVES * 64, lds, per_cu);
    if (ost != X265AMD_OK) return ost;
    if (per_cu < 1) return X265AMD_ERR_LDS;
    int resident = per_cu * ds.cu_count;
    if (WAVES == 16 && resident > kMaxWgLists) resident = kMaxWgLists;  // schedule slices in the workspace
    const dim3 grid((unsigned)(total < resident ? total : resident));
    if (stats)
        hipLaunchKernelGGL((sea_exact_kernel<true, WAVES, B>), grid, dim3(WAVES * 64), lds, st, d_cur, d_ref, gl, list0, wg_list,
                           hdr, order, ctrl, total, d_out, g, gl_pitch, win_dwords, surv_off, surv_cap, stats);
    else
        hipLaunchKernelGGL((sea_exact_kernel<false, WAVES, B>), grid, dim3(WAVES * 64), lds, st, d_cur, d_ref, gl, list0, wg_list,
                           hdr, order, ctrl, total, d_out, g, gl_pitch, win_dwords, surv_off, surv_cap, stats);
    return X265AMD_OK;
}

// the search on a workspace that the caller brought (x265amd_me_fullsearch_sea) or that the library
// holds per device (x265amd_me_fullsearch_sea_static): three launches, nothing allocated
static int sea_search(DeviceState &ds, const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride, int ctu,
                      int range, int nframes, int64_t cur_frame_stride, int64_t ref_frame_stride, int nrefs,
                      int64_t ref_ref_stride, x265amd_mv *d_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!d_cur || !d_ref || !d_out || !d_workspace) return X265AMD_ERR_ARG;
    if (ctu != 64 && ctu != 32) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % ctu || height % ctu || stride < width) return X265AMD_ERR_ARG;
    if (range < 0 || range > 1024 || nframes < 0 || nrefs < 1) return X265AMD_ERR_ARG;
    if ((stride & 3) || ((uintptr_t)d_cur & 3) || ((uintptr_t)d_ref & 3) || (cur_frame_stride & 3) ||
        (ref_frame_stride & 3) || (ref_ref_stride & 3) || ((uintptr_t)d_workspace & 15))
        return X265AMD_ERR_ARG;
    if (nframes == 0) return X265AMD_OK;
    const int64_t nplanes = (int64_t)nframes * nrefs;
    if (nplanes > 65535) return X265AMD_ERR_ARG;
    if (workspace_bytes < x265amd_me_workspace_bytes_ctu(width, height, ctu, range, nframes, nrefs)) return X265AMD_ERR_ARG;

    int max_rows, max_pitch, max_ngx, max_ndy;
    sea_extents(width, height, range, ctu, max_rows, max_pitch, max_ngx, max_ndy);
    const int max_groups = max_ngx * max_ndy;
    const int gl_pitch = (max_groups + 7) & ~7;
    if (max_groups > 0xffff || max_ndy > (ctu / 4) * sea_max_chain(ctu / 4)) return X265AMD_ERR_LDS;  // 16-bit schedule entries
    const int win_dwords = max_rows * max_pitch;
    const size_t list_bytes = ((size_t)max_groups * 2 + 15) & ~(size_t)15;
    // 8-wave workgroups (window + schedule + kept list in LDS, 16 KB of partial sums) when two of them fit a CU;
    // otherwise ONE 16-wave workgroup per CU (window + kept list, 32 KB of partial sums, schedule in the workspace)
    constexpr int kSurv8 = 512;  // kept groups of an 8-wave workgroup: two half rounds' worth, 18 bytes each
    const size_t surv8_off = (((size_t)win_dwords * 4 + list_bytes + 15) & ~(size_t)15);
    const size_t lds8 = surv8_off + (size_t)kSurv8 * 18 + (size_t)ds.dbg.sea_lds_pad;
    size_t lds16 = (size_t)win_dwords * 4;
    const size_t lds_cap = (size_t)ds.lds_limit;
    bool waves16 = 2 * (lds8 + 17408) > lds_cap;
    if (ds.dbg.sea_waves) waves16 = ds.dbg.sea_waves == 16;
    if (waves16 && lds16 + 33792 > lds_cap) waves16 = false;
    // 16-wave shape: what is left of the LDS holds the groups that outlive the upper half of the block in the second
    // pass (18 bytes each; at least two half rounds' worth = 1024, else the second pass runs whole rounds)
    int surv_cap = 0;
    if (waves16) {
        const size_t base = (((size_t)win_dwords + 3) & ~(size_t)3) * 4;
        const size_t room = lds_cap > base + 33792 + 64 ? lds_cap - base - 33792 - 64 : 0;
        const size_t cap = (room / 18) & ~(size_t)511;
        if (cap >= 1024) {
            surv_cap = (int)(cap > 4096 ? 4096 : cap);
            lds16 = base + (size_t)surv_cap * 18;
        }
    }
    if (waves16 && surv_cap == 0) return X265AMD_ERR_LDS;  // the second pass of the 16-wave shape needs its kept list
    if (!waves16 && lds8 + 17408 > lds_cap) return X265AMD_ERR_LDS;  // + the kernel's static LDS
    const size_t lds_bounds = ((size_t)max_groups * 2 + 15) & ~(size_t)15;

    const size_t nctu = (size_t)(width / ctu) * (height / ctu);
    uint8_t *ws = (uint8_t *)d_workspace;
    uint32_t *pbuf = (uint32_t *)ws;  // pair plane (16-byte loads)
    ws += align256(4 * (size_t)width * height * (size_t)nplanes);
    uint16_t *gl = (uint16_t *)ws;
    ws += align256(sizeof(uint16_t) * (size_t)gl_pitch * nctu * (size_t)nplanes);
    uint16_t *list0 = (uint16_t *)ws;
    ws += align256(sizeof(uint16_t) * (size_t)gl_pitch * nctu * (size_t)nplanes);
    SeaHdr *hdr = (SeaHdr *)ws;
    ws += align256(sizeof(SeaHdr) * nctu * (size_t)nplanes);
    uint32_t *order = (uint32_t *)ws;
    ws += align256(sizeof(uint32_t) * (kClasses + 1) * nctu * (size_t)nplanes);
    uint32_t *ctrl = (uint32_t *)ws;  // queue lengths and hand-out counters (see kClasses)
    if ((int64_t)nctu * nplanes > 0x7fffffff) return X265AMD_ERR_ARG;
    const int total = (int)(nctu * (size_t)nplanes);

    hipStream_t st = (hipStream_t)stream;
    const bool timing = ds.dbg.me_timing;
    hipEvent_t *const g_me_ev = ds.dbg.me_ev;
    uint32_t *const bounds_stats = ds.dbg.me_stats;
    if (timing) (void)hipEventRecord(g_me_ev[0], st);
    const int S = ctu / 4;  // sub-block side: rows y of the pair plane exist for y + S <= height
    const dim3 box_grid((width + kBoxTW - 1) / kBoxTW, (height - (S - 1) + kBoxTH - 1) / kBoxTH, (unsigned)nplanes);
    if (ctu == 64)
        hipLaunchKernelGGL(box16_pairs<16>, box_grid, dim3(256), 0, st, d_ref, pbuf, width, height, stride, nrefs, ref_frame_stride,
                           ref_ref_stride, ctrl);
    else
        hipLaunchKernelGGL(box16_pairs<8>, box_grid, dim3(256), 0, st, d_ref, pbuf, width, height, stride, nrefs, ref_frame_stride,
                           ref_ref_stride, ctrl);
    if (timing) (void)hipEventRecord(g_me_ev[1], st);
    if (timing) (void)hipEventRecord(g_me_ev[2], st);
    MeGeom g;
    g.width = width;
    g.height = height;
    g.stride = stride;
    g.range = range;
    g.nrefs = nrefs;
    g.ctus_x = width / ctu;
    g.ctus_y = height / ctu;
    g.cur_fs = cur_frame_stride;
    g.ref_fs = ref_frame_stride;
    g.ref_rs = ref_ref_stride;
    const int nfirst = waves16 ? 512 : 256;  // one half round of kernel 2's shape; kernel 1 deals it a thread per entry
    const dim3 bgrid((unsigned)((total + 7) & ~7));
    if (ctu == 64)
        hipLaunchKernelGGL((sea_bounds_kernel<64, kBoundsThreads>), bgrid, dim3(kBoundsThreads), lds_bounds, st, d_cur, d_ref, pbuf, gl, list0,
                           hdr, order, ctrl, g, gl_pitch, (int)nplanes, nfirst, bounds_stats);
    else if (nfirst <= 256)
        hipLaunchKernelGGL((sea_bounds_kernel<32, 256>), bgrid, dim3(256), lds_bounds, st, d_cur, d_ref, pbuf, gl, list0, hdr, order, ctrl,
                           g, gl_pitch, (int)nplanes, nfirst, bounds_stats);
    else
        hipLaunchKernelGGL((sea_bounds_kernel<32, kBoundsThreads>), bgrid, dim3(kBoundsThreads), lds_bounds, st, d_cur, d_ref, pbuf, gl, list0,
                           hdr, order, ctrl, g, gl_pitch, (int)nplanes, nfirst, bounds_stats);
    if (timing) (void)hipEventRecord(g_me_ev[3], st);
    uint16_t *wg_list = (uint16_t *)(ws + 256);
    int status;
#define X265AMD_EXACT(WV, BB, LDS) \
    launch_exact<WV, BB>(ds, LDS, st, d_cur, d_ref, gl, list0, wg_list, hdr, order, ctrl, total, d_out, g, gl_pitch, win_dwords, \
                         waves16 ? (int)(((size_t)win_dwords + 3) & ~(size_t)3) : (int)(surv8_off / 4), waves16 ? surv_cap : kSurv8)
    if (ctu == 64)
        status = waves16 ? X265AMD_EXACT(16, 64, lds16) : X265AMD_EXACT(8, 64, lds8);
    else
        status = waves16 ? X265AMD_EXACT(16, 32, lds16) : X265AMD_EXACT(8, 32, lds8);
#undef X265AMD_EXACT
    if (status != X265AMD_OK) return status;
    {
        // kernel 3: a workgroup per block, of which those beyond the every-candidate list exit at once (usually all)
        int mr = 0, mp = 0, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows;
        for (int cy = 0; cy < height; cy += ctu) {
            me_qsad_layout(width, height, range, 0, cy, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows, ctu);
            mr = win_rows > mr ? win_rows : mr;
        }
        for (int cx = 0; cx < width; cx += ctu) {
            me_qsad_layout(width, height, range, cx, 0, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows, ctu);
            mp = pitch > mp ? pitch : mp;
        }
        const size_t lds3 = (size_t)mr * mp * 4;
        if (lds3 + 128 > lds_cap) return X265AMD_ERR_LDS;
        const dim3 grid3((unsigned)total);
        const uint32_t *blist = order + (size_t)kClasses * total;
        if (ctu == 64)
            hipLaunchKernelGGL(me_list_qsad<64>, grid3, dim3(256), lds3, st, d_cur, d_ref, d_out, g, blist, ctrl, gl, gl_pitch, hdr);
        else
            hipLaunchKernelGGL(me_list_qsad<32>, grid3, dim3(256), lds3, st, d_cur, d_ref, d_out, g, blist, ctrl, gl, gl_pitch, hdr);
    }
    if (timing) (void)hipEventRecord(g_me_ev[4], st);
    return hip_status(hipGetLastError());
}

extern "C" int x265amd_me_fullsearch_sea(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                                         int stride, int ctu, int range, int nframes, int64_t cur_frame_stride,
                                         int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride,
                                         x265amd_mv *d_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    DeviceState *ds = device_state();
    if (!ds) return X265AMD_ERR_NOT_INIT;
    return sea_search(*ds, d_cur, d_ref, width, height, stride, ctu, range, nframes, cur_frame_stride, ref_frame_stride,
                      nrefs, ref_ref_stride, d_out, d_workspace, workspace_bytes, stream);
}

extern "C" size_t x265amd_me_static_workspace_bytes(void) { return kSeaStaticBytes; }

// The same three kernels on the device's static scratch array: no hipMalloc / hipFree / allocating routine on
// this path (the LDS opt-ins and the array's address were taken in x265amd_init; the occupancy query is host
// arithmetic).  The kernels zero their own queue counters, and every other part of the scratch is written
// before it is read within a call, so nothing of the previous call's contents can leak into this one.
extern "C" int x265amd_me_fullsearch_sea_static(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                                                int stride, int ctu, int range, int nframes, int64_t cur_frame_stride,
                                                int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride,
                                                x265amd_mv *d_out, void *stream)
{
    DeviceState *ds = device_state();
    if (!ds) return X265AMD_ERR_NOT_INIT;
    if (!ds->sea_scratch) return X265AMD_ERR_NOT_INIT;
    if (nframes > 0 && nrefs > 0) {
        const size_t need = x265amd_me_workspace_bytes_ctu(width, height, ctu, range, nframes, nrefs);
        if (need == 0) return X265AMD_ERR_ARG;
        if (need > ds->sea_scratch_bytes) return X265AMD_ERR_WORKSPACE;  // beyond the static maximum: use _sea
    }
    return sea_search(*ds, d_cur, d_ref, width, height, stride, ctu, range, nframes, cur_frame_stride, ref_frame_stride,
                      nrefs, ref_ref_stride, d_out, ds->sea_scratch, ds->sea_scratch_bytes, stream);
}
