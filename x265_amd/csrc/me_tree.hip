// me_tree.hip -- every square PU size of a CTU in ONE exhaustive pass ("sub-CTU PU costs,
// SAD tree -> 16/32/64"; BASELINE.json north_star: "per-CTU/PU integer-pel motion search").
//
// For each 64x64 CTU the kernel walks every displacement (dx, dy) in [-R, R]^2 once, forms the sixteen
// 16x16 SADs of the CTU at that displacement (v_qsad_pk_u16_u8, current rows in SGPRs, reference window
// in LDS) and adds them up the tree: 16 PUs of 16x16, 4 of 32x32, the CTU itself.  Each PU keeps its own
// winner under its OWN legality (a PU may move where the whole CTU may not: the window is staged unclipped,
// zero outside the picture), with the reference's argmin idiom (first strict minimum in raster order,
// src/encode.cpp:725-727, 762-765).  Every output entry equals what x265amd_me_fullsearch returns for
// ctu = 64 / 32 / 16 -- that is the parity test -- at the cost of one pass instead of three.
#include "common.h"
#include "me_common.h"

namespace x265amd {

constexpr int kTreePUs = 21;  // 16 + 4 + 1

struct TreeOut {
    x265amd_mv *o64, *o32, *o16;
};

__global__ __launch_bounds__(256, 2) void me_tree_kernel(const uint8_t *__restrict__ cur_base, const uint8_t *__restrict__ ref_base,
                                                         const TreeOut out, const MeGeom g, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *win = smem;  // [2R + 64][pitch]
    __shared__ unsigned long long wbest[4][kTreePUs];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ctx, cty, bz;
    me_xcd_block(ctx, cty, bz);
    const int f = bz / g.nrefs, rr = bz - f * g.nrefs;
    const int cx = ctx * 64, cy = cty * 64;
    const int R = g.range, span = 2 * R + 1;
    const uint8_t *cur = cur_base + f * g.cur_fs;
    const uint8_t *ref = ref_base + f * g.ref_fs + rr * g.ref_rs;
    const int wx0 = (cx - R) & ~3;  // window origin on a dword boundary (may be negative)
    const int a = cx - R - wx0;     // window column of dx = -R
    const int ngx = (a + span + 3) >> 2;
    const int win_rows = 2 * R + 64;

    // ---- stage the window, unclipped: zero outside the picture ----
    for (int i = tid; i < win_rows * pitch; i += 256) {
        const int wy = i / pitch, wq = i - wy * pitch;
        const int gx = wx0 + 4 * wq, gy = cy - R + wy;
        uint32_t v = 0;
        if (gy >= 0 && gy < g.height && gx >= 0 && gx + 4 <= g.width)
            v = *reinterpret_cast<const uint32_t *>(ref + (int64_t)gy * g.stride + gx);
        win[i] = v;
    }
    __syncthreads();

    uint32_t bsad[kTreePUs], bidx[kTreePUs];  // a lane meets its candidates in raster order: strict < keeps the first
#pragma unroll
    for (int p = 0; p < kTreePUs; p++) {
        bsad[p] = 0xffffffffu;
        bidx[p] = 0xffffffffu;
    }
    const uint32_t *__restrict__ cbase = reinterpret_cast<const uint32_t *>(cur + (int64_t)cy * g.stride + cx);
    const int cstr = g.stride >> 2;
    const int ntask = ngx * span;
    for (int task = tid; task < ((ntask + 63) & ~63); task += 256) {  // whole waves: the current rows are scalar operands
        const int tk = min(task, ntask - 1);
        const int dyi = tk / ngx, gx = tk - dyi * ngx;
        const uint32_t *wrow = win + dyi * pitch + gx;
        uint64_t s16[4][4];  // [block row][block column]: four dx as packed u16 (a 16x16 SAD is at most 65280)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint64_t acc[4] = {0, 0, 0, 0};
#pragma unroll 2
            for (int r = 0; r < 16; r++) {
                const uint32_t *p = wrow + (16 * j + r) * pitch;
                const uint32_t *c = cbase + (16 * j + r) * cstr;
                uint32_t d[17];
#pragma unroll
                for (int k = 0; k < 17; k++) d[k] = p[k];
#pragma unroll
                for (int k = 0; k < 16; k++)
                    acc[k >> 2] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k + 1] << 32) | d[k], c[k], acc[k >> 2]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) s16[j][k] = acc[k];
        }
        if (task < ntask) {
            const int dy = dyi - R;
            // rows: a PU at row offset oy of size B is legal iff 0 <= cy + oy + dy and cy + oy + dy + B <= height
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int dx = wx0 - cx + 4 * gx + q;
                if (dx < -R || dx > R) continue;
                const uint32_t idx = (uint32_t)((dy + R) * span + dx + R);
                uint32_t v16[4][4];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v16[j][k] = (uint32_t)(s16[j][k] >> (16 * q)) & 0xffffu;
                auto legal = [&](int ox, int oy, int B) {
                    const int x = cx + ox + dx, y = cy + oy + dy;
                    return x >= 0 && x + B <= g.width && y >= 0 && y + B <= g.height;
                };
                auto take = [&](int p, uint32_t sad, bool ok) {
                    if (ok && sad < bsad[p]) {
                        bsad[p] = sad;
                        bidx[p] = idx;
                    }
                };
                uint32_t v32[2][2];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int k = 0; k < 4; k++) take(4 * j + k, v16[j][k], legal(16 * k, 16 * j, 16));
#pragma unroll
                for (int J = 0; J < 2; J++)
#pragma unroll
                    for (int K = 0; K < 2; K++) {
                        v32[J][K] = v16[2 * J][2 * K] + v16[2 * J][2 * K + 1] + v16[2 * J + 1][2 * K] + v16[2 * J + 1][2 * K + 1];
                        take(16 + 2 * J + K, v32[J][K], legal(32 * K, 32 * J, 32));
                    }
                take(20, v32[0][0] + v32[0][1] + v32[1][0] + v32[1][1], legal(0, 0, 64));
            }
        }
    }
    // ---- winners: 64-bit key (sad << 32 | raster index) across lanes, then across the four waves ----
#pragma unroll
    for (int p = 0; p < kTreePUs; p++) {
        unsigned long long k = ((unsigned long long)bsad[p] << 32) | bidx[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(k, o);
            k = other < k ? other : k;
        }
        if (lane == 0) wbest[wave][p] = k;
    }
    __syncthreads();
    if (tid < kTreePUs) {
        unsigned long long k = wbest[0][tid];
        for (int w = 1; w < 4; w++) k = wbest[w][tid] < k ? wbest[w][tid] : k;
        x265amd_mv mv;
        const int c = (int)(k & 0xffffffffu);
        mv.mvx = (int16_t)(c % span - R);
        mv.mvy = (int16_t)(c / span - R);
        mv.sad = (uint32_t)(k >> 32);
        const int64_t plane = (int64_t)f * g.nrefs + rr;
        const int64_t nctu = (int64_t)g.ctus_x * g.ctus_y;
        if (tid < 16) {
            if (out.o16) out.o16[plane * 16 * nctu + (int64_t)(4 * cty + (tid >> 2)) * (4 * g.ctus_x) + 4 * ctx + (tid & 3)] = mv;
        } else if (tid < 20) {
            const int t = tid - 16;
            if (out.o32) out.o32[plane * 4 * nctu + (int64_t)(2 * cty + (t >> 1)) * (2 * g.ctus_x) + 2 * ctx + (t & 1)] = mv;
        } else if (out.o64) {
            out.o64[plane * nctu + (int64_t)cty * g.ctus_x + ctx] = mv;
        }
    }
}

int me_tree_device_setup(DeviceState &ds) { return allow_full_lds((const void *)me_tree_kernel, ds); }

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_me_fullsearch_pu(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                                        int range, int nframes, int64_t cur_frame_stride, int64_t ref_frame_stride,
                                        int nrefs, int64_t ref_ref_stride, x265amd_mv *d_out64, x265amd_mv *d_out32,
                                        x265amd_mv *d_out16, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_cur || !d_ref || (!d_out64 && !d_out32 && !d_out16)) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % 64 || height % 64 || stride < width) return X265AMD_ERR_ARG;
    if (range < 0 || range > 1024 || nframes < 0 || nrefs < 1) return X265AMD_ERR_ARG;
    if ((stride & 3) || ((uintptr_t)d_cur & 3) || ((uintptr_t)d_ref & 3) || (cur_frame_stride & 3) ||
        (ref_frame_stride & 3) || (ref_ref_stride & 3))
        return X265AMD_ERR_ARG;
    if (nframes == 0) return X265AMD_OK;
    if ((int64_t)nframes * nrefs > 65535) return X265AMD_ERR_ARG;
    // the widest window: 3 leading bytes of alignment + (2R + 1) displacements + 63 more columns + 4 of the last qsad
    const int pitch = ((3 + 2 * range + 64 + 4 + 3) / 4 + 1) | 1;
    const size_t lds = (size_t)(2 * range + 64) * pitch * 4;
    if (lds + 1024 > (size_t)lds_limit_bytes()) return X265AMD_ERR_LDS;
    MeGeom g;
    g.width = width;
    g.height = height;
    g.stride = stride;
    g.range = range;
    g.nrefs = nrefs;
    g.ctus_x = width / 64;
    g.ctus_y = height / 64;
    g.cur_fs = cur_frame_stride;
    g.ref_fs = ref_frame_stride;
    g.ref_rs = ref_ref_stride;
    TreeOut out{d_out64, d_out32, d_out16};
    hipLaunchKernelGGL(me_tree_kernel, dim3(g.ctus_x, g.ctus_y, (unsigned)(nframes * nrefs)), dim3(256), lds, (hipStream_t)stream,
                       d_cur, d_ref, out, g, pitch);
    return hip_status(hipGetLastError());
}
