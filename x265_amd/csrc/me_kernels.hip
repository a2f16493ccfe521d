// me_kernels.hip -- 8-bit SAD and exhaustive integer-pel motion search for gfx950.
//
// Replaces xSadN[] (src/pixel.cpp:30-101) for batches of blocks and supplies the
// search loop the reference never had (src/encode.cpp:581-584 "No P-Slice now"),
// defined by SURVEY.md App. A.7 with the reference's argmin idiom
// (src/encode.cpp:725-727,762-765).
#include "common.h"
#include "me_common.h"
#include "me_qsad_body.h"

namespace x265amd {

struct __attribute__((packed, aligned(1))) U32 {
    uint32_t v;
};

__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc)
{
    return __builtin_amdgcn_sad_u8(a, b, acc);  // v_sad_u8: acc + sum_{4 bytes} |a - b|
}

// ---------------------------------------------------------------------------
// Batched SAD: one wave per block pair.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sad_blocks_kernel(int wq /* w/4 */, int n, const uint8_t *src, int ss,
                                                         const uint8_t *ref, int sr, const uint32_t *src_off,
                                                         const uint32_t *ref_off, int nblocks, uint32_t *out)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= nblocks) return;  // wave-uniform
    const uint8_t *s = src + src_off[b];
    const uint8_t *r = ref + ref_off[b];
    uint32_t acc = 0;
    for (int i = lane; i < wq * n; i += 64) {
        const int y = i / wq, x = (i - y * wq) * 4;
        acc = sad4(reinterpret_cast<const U32 *>(s + (int64_t)y * ss + x)->v,
                   reinterpret_cast<const U32 *>(r + (int64_t)y * sr + x)->v, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc;
}

// ---------------------------------------------------------------------------
// Exhaustive search, baseline kernel: one workgroup per (CTU, ref, frame).
// The (ctu + 2R)-wide reference window and the current block are staged in LDS
// with coalesced dword loads; each lane owns one candidate at a time, builds
// the unaligned reference dwords with v_alignbyte_b32 and accumulates with
// v_sad_u8.  Winner = min over the 64-bit key (sad << 32 | raster index),
// reduced with wave shuffles, which is exactly "first strict minimum in raster
// order".
// ---------------------------------------------------------------------------
struct MeParams {
    const uint8_t *cur, *ref;
    x265amd_mv *out;
    int width, height, stride, ctu, range;
    int nrefs, ctus_x, ctus_y;
    int64_t cur_fs, ref_fs, ref_rs;
    int win_dw;    // window row pitch in dwords
    int win_rows;  // ctu + 2R
};

__global__ __launch_bounds__(256) void me_fullsearch_basic(const MeParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *win = smem;                            // [win_rows][win_dw]
    uint32_t *curb = smem + p.win_rows * p.win_dw;   // [ctu][ctu/4]
    const int tid = threadIdx.x;
    const int ctx = blockIdx.x, cty = blockIdx.y;
    const int f = blockIdx.z / p.nrefs, r = blockIdx.z - f * p.nrefs;
    const int cx = ctx * p.ctu, cy = cty * p.ctu;
    const int cq = p.ctu >> 2;
    const uint8_t *cur = p.cur + f * p.cur_fs;
    const uint8_t *ref = p.ref + f * p.ref_fs + r * p.ref_rs;

    // window origin, rounded down to a dword boundary (floor also for negatives)
    const int wx0 = (cx - p.range) & ~3;
    const int wy0 = cy - p.range;
    for (int i = tid; i < p.win_rows * p.win_dw; i += 256) {
        const int wy = i / p.win_dw, wq = i - wy * p.win_dw;
        const int gx = wx0 + 4 * wq, gy = wy0 + wy;
        uint32_t v = 0;
        if (gy >= 0 && gy < p.height && gx >= 0 && gx + 4 <= p.width)
            v = *reinterpret_cast<const uint32_t *>(ref + (int64_t)gy * p.stride + gx);
        win[i] = v;
    }
    for (int i = tid; i < p.ctu * cq; i += 256) {
        const int y = i / cq, q = i - y * cq;
        curb[i] = *reinterpret_cast<const uint32_t *>(cur + (int64_t)(cy + y) * p.stride + cx + 4 * q);
    }
    __syncthreads();

    const int span = 2 * p.range + 1;
    const int ncand = span * span;
    unsigned long long best = ~0ull;
    for (int c = tid; c < ncand; c += 256) {
        const int iy = c / span, ix = c - iy * span;
        const int dy = iy - p.range, dx = ix - p.range;
        const int rx = cx + dx, ry = cy + dy;
        if (rx < 0 || ry < 0 || rx + p.ctu > p.width || ry + p.ctu > p.height) continue;
        const int bx = rx - wx0;         // byte offset inside the window row
        const int q0 = bx >> 2, sh = bx & 3;
        uint32_t acc = 0;
        for (int y = 0; y < p.ctu; y++) {
            const uint32_t *wr = win + (iy + y) * p.win_dw + q0;
            const uint32_t *cr = curb + y * cq;
            uint32_t lo = wr[0];
            for (int q = 0; q < cq; q++) {
                const uint32_t hi = wr[q + 1];
                acc = sad4(__builtin_amdgcn_alignbyte(hi, lo, sh), cr[q], acc);
                lo = hi;
            }
        }
        const unsigned long long key = ((unsigned long long)acc << 32) | (unsigned)c;
        best = key < best ? key : best;
    }
    // wave min, then across the four waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    __shared__ unsigned long long wbest[4];
    if ((tid & 63) == 0) wbest[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) best = wbest[w] < best ? wbest[w] : best;
        x265amd_mv mv;
        const int c = (int)(best & 0xffffffffu);
        mv.mvx = (int16_t)(c % span - p.range);
        mv.mvy = (int16_t)(c / span - p.range);
        mv.sad = (uint32_t)(best >> 32);
        p.out[((int64_t)f * p.nrefs + r) * p.ctus_x * p.ctus_y + cty * p.ctus_x + ctx] = mv;
    }
}


// ---------------------------------------------------------------------------
// Exhaustive search, production kernel: one workgroup per (block, ref, frame); the search itself is
// me_qsad_block (me_qsad_body.h).
// ---------------------------------------------------------------------------
template <int B>  // block size: 64 (CTU), 32 or 16 (PU sizes); B / 4 dwords per row
__global__ __launch_bounds__(256, 2) void me_fullsearch_qsad(const uint8_t *__restrict__ cur_base,
                                                             const uint8_t *__restrict__ ref_base,
                                                             x265amd_mv *__restrict__ out, const MeGeom g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    int ctx, cty, z;
    me_xcd_block(ctx, cty, z);
    const int f = z / g.nrefs, rr = z - f * g.nrefs;
    const unsigned long long best = me_qsad_block<B>(smem, cur_base + f * g.cur_fs, ref_base + f * g.ref_fs + rr * g.ref_rs, g,
                                                     ctx * B, cty * B, 0, 1);
    if (threadIdx.x == 0)
        out[((int64_t)f * g.nrefs + rr) * g.ctus_x * g.ctus_y + cty * g.ctus_x + ctx] = me_key_to_mv(best, g.range);
}

// x265amd_init, once per device: every search kernel may use the device's whole LDS from then on
int me_kernels_device_setup(DeviceState &ds)
{
    const void *fns[] = {(const void *)me_fullsearch_qsad<64>, (const void *)me_fullsearch_qsad<32>,
                         (const void *)me_fullsearch_qsad<16>, (const void *)me_fullsearch_basic};
    for (const void *fn : fns) {
        const int st = allow_full_lds(fn, ds);
        if (st != X265AMD_OK) return st;
    }
    return X265AMD_OK;
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_sad_blocks(int w, int n, const uint8_t *d_src, int stride_src, const uint8_t *d_ref,
                                  int stride_ref, const uint32_t *d_src_off, const uint32_t *d_ref_off, int nblocks,
                                  uint32_t *d_out, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (w != 4 && w != 8 && w != 16 && w != 32 && w != 64) return X265AMD_ERR_ARG;
    if (n < 1 || n > 64 || nblocks < 0 || !d_src || !d_ref || !d_src_off || !d_ref_off || !d_out)
        return X265AMD_ERR_ARG;
    if (nblocks == 0) return X265AMD_OK;
    hipLaunchKernelGGL(sad_blocks_kernel, dim3((nblocks + 3) / 4), dim3(256), 0, (hipStream_t)stream, w / 4, n,
                       d_src, stride_src, d_ref, stride_ref, d_src_off, d_ref_off, nblocks, d_out);
    return hip_status(hipGetLastError());
}

extern "C" int x265amd_me_fullsearch(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                                     int ctu, int range, int nframes, int64_t cur_frame_stride,
                                     int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride, x265amd_mv *d_out,
                                     void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_cur || !d_ref || !d_out) return X265AMD_ERR_ARG;
    if (ctu != 16 && ctu != 32 && ctu != 64) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % ctu || height % ctu || stride < width) return X265AMD_ERR_ARG;
    if (range < 0 || range > 1024 || nframes < 0 || nrefs < 1) return X265AMD_ERR_ARG;
    if ((stride & 3) || ((uintptr_t)d_cur & 3) || ((uintptr_t)d_ref & 3) || (cur_frame_stride & 3) ||
        (ref_frame_stride & 3) || (ref_ref_stride & 3))
        return X265AMD_ERR_ARG;
    if (nframes == 0) return X265AMD_OK;
    if ((int64_t)nframes * nrefs > 65535) return X265AMD_ERR_ARG;

    {
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
        // window rows depend on the block row only, the pitch on the block column only
        int max_rows = 0, max_pitch = 0;
        int dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows;
        for (int cy = 0; cy < height; cy += ctu) {
            me_qsad_layout(width, height, range, 0, cy, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows, ctu);
            max_rows = win_rows > max_rows ? win_rows : max_rows;
        }
        for (int cx = 0; cx < width; cx += ctu) {
            me_qsad_layout(width, height, range, cx, 0, dylo, dyhi, dxlo, dxhi, wx0, ngx, ngy, pitch, win_rows, ctu);
            max_pitch = pitch > max_pitch ? pitch : max_pitch;
        }
        const size_t lds = (size_t)max_rows * max_pitch * 4;
        if (lds + 64 > (size_t)lds_limit_bytes()) return X265AMD_ERR_LDS;
        const dim3 grid(g.ctus_x, g.ctus_y, nframes * nrefs);
        if (ctu == 64)
            hipLaunchKernelGGL(me_fullsearch_qsad<64>, grid, dim3(256), lds, (hipStream_t)stream, d_cur, d_ref, d_out, g);
        else if (ctu == 32)
            hipLaunchKernelGGL(me_fullsearch_qsad<32>, grid, dim3(256), lds, (hipStream_t)stream, d_cur, d_ref, d_out, g);
        else
            hipLaunchKernelGGL(me_fullsearch_qsad<16>, grid, dim3(256), lds, (hipStream_t)stream, d_cur, d_ref, d_out, g);
        return hip_status(hipGetLastError());
    }
}

// The one-candidate-per-lane formulation (v_alignbyte_b32 + v_sad_u8, current block in LDS): an independent
// cross-check of the production kernel, reachable through the debug entry point below only.
extern "C" int x265amd_debug_me_fullsearch_basic(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                                                 int stride, int ctu, int range, int nframes, int64_t cur_frame_stride,
                                                 int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride,
                                                 x265amd_mv *d_out, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_cur || !d_ref || !d_out) return X265AMD_ERR_ARG;
    if (ctu != 16 && ctu != 32 && ctu != 64) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % ctu || height % ctu || stride < width) return X265AMD_ERR_ARG;
    if (range < 0 || range > 1024 || nframes <= 0 || nrefs < 1) return X265AMD_ERR_ARG;
    if ((stride & 3) || ((uintptr_t)d_cur & 3) || ((uintptr_t)d_ref & 3) || (cur_frame_stride & 3) ||
        (ref_frame_stride & 3) || (ref_ref_stride & 3))
        return X265AMD_ERR_ARG;
    if ((int64_t)nframes * nrefs > 65535) return X265AMD_ERR_ARG;
    MeParams p;
    p.cur = d_cur;
    p.ref = d_ref;
    p.out = d_out;
    p.width = width;
    p.height = height;
    p.stride = stride;
    p.ctu = ctu;
    p.range = range;
    p.nrefs = nrefs;
    p.ctus_x = width / ctu;
    p.ctus_y = height / ctu;
    p.cur_fs = cur_frame_stride;
    p.ref_fs = ref_frame_stride;
    p.ref_rs = ref_ref_stride;
    p.win_rows = ctu + 2 * range;
    p.win_dw = (ctu + 2 * range + 3) / 4 + 2;  // + dword-alignment slack + the hi dword of the last alignbyte
    const size_t lds = ((size_t)p.win_rows * p.win_dw + (size_t)ctu * ctu / 4) * 4;
    if (lds + 64 > (size_t)lds_limit_bytes()) return X265AMD_ERR_LDS;
    hipLaunchKernelGGL(me_fullsearch_basic, dim3(p.ctus_x, p.ctus_y, nframes * nrefs), dim3(256), lds,
                       (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}
