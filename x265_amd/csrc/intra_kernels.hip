// intra_kernels.hip -- the intra luma mode decision of a batch of CUs (first slice):
// xEncIntraPredLuma for all 35 modes (src/encode.cpp:1617-1652 over xPredIntraPlanar :1353, xPredIntraDc :1418,
// xPredIntraAng :1446), the SAD of each prediction against the source CU and the cost / argmin of
// src/encode.cpp:725-765.  Input per CU = what xEncIntraLoadRef (src/encode.cpp:1163) leaves in the encoder's
// cache: the padded reference samples pucPixRef[0] and the three most probable modes; the [1 2 1] smoothing of the
// samples (:1260-1265) is done here.  The CUs of one launch are independent (in the encoder: the CUs of one
// anti-diagonal, whose neighbours are reconstructed).
//
// One workgroup per CU.  The samples and their smoothed copy sit in LDS; a prediction is never stored: every
// lane evaluates ITS pixels of a mode in closed form (planar and angular are written without the reference's
// running sums and without its transpose: pixel (r, c) of a horizontal mode is pixel (c, r) of the mirrored
// vertical construction), takes |source - prediction| and the wave adds up.  The 35 modes are dealt round the
// four waves; thread 0 does the argmin in mode order.  On request the winner's prediction is written out for the
// residual stage (src/encode.cpp:867).
#include "common.h"

namespace x265amd {

// src/table.cpp:32-39, 49-56, 59-66
__device__ const uint8_t kIntraFilter[4][35] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1}};
__device__ const int8_t kIntraAngle[35] = {0,   0,   32,  26,  21,  17,  13,  9,  5,  2,  0,  -2, -5, -9, -13, -17, -21, -26,
                                           -32, -26, -21, -17, -13, -9,  -5,  -2, 0,  2,  5,  9,  13, 17, 21,  26,  32};
__device__ const int16_t kIntraInvAngle[35] = {0,   0,   256, 315, 390, 482, 630, 910, 1638, 4096, 0,   4096, 1638, 910, 630, 482, 390, 315,
                                               256, 315, 390, 482, 630, 910, 1638, 4096, 0,  4096, 1638, 910, 630, 482, 390, 315, 256};

constexpr int kMaxRef = 4 * 32 + 1;

// pixel (r, c) of mode `mode`; ref = the sample array the mode reads (smoothed or not), dc = the DC value
template <int N> __device__ __forceinline__ int intra_pixel(const uint8_t *ref, int mode, int r, int c, int dc)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    const uint8_t *tl = ref + 2 * N;  // the corner; tl[1 + j] = above column j, tl[-1 - i] = left of row i
    if (mode == 0) {  // planar (src/encode.cpp:1353-1396)
        const int left = tl[-1 - r], top = tl[1 + c], bl = tl[-1 - N], tr = tl[1 + N];
        const int hp = (left << L2) + N + (c + 1) * (tr - left);
        const int tp = (top << L2) + (r + 1) * (bl - top);
        return ((hp + tp) >> (L2 + 1)) & 0xff;
    }
    if (mode == 1) {  // DC with its edge smoothing (src/encode.cpp:1418-1444)
        if (r == 0 && c == 0) return (tl[1] + tl[-1] + 2 * dc + 2) >> 2;
        if (r == 0) return (tl[1 + c] + 3 * dc + 2) >> 2;
        if (c == 0) return (tl[-1 - r] + 3 * dc + 2) >> 2;
        return dc;
    }
    // angular (src/encode.cpp:1446-1534): the main side is the left column for modes 2..17 (built as a row and
    // transposed at the end: swap the roles of r and c instead), the row above for 18..34
    const bool hor = mode < 18;
    const int ang = kIntraAngle[mode], inv = kIntraInvAngle[mode];
    const int k = hor ? c : r, x = hor ? r : c;
    const int sgn = hor ? -1 : 1;  // main side: tl[sgn * i]; the other side, projected onto it: tl[-sgn * offset]
    auto main_ref = [&](int i) -> int {
        if (i >= 0) return tl[sgn * i];
        const int off = (128 + (-i) * inv) >> 8;
        return tl[-sgn * off];
    };
    const int pos = (k + 1) * ang;
    const int idx = pos >> 5, fact = pos & 31;
    int p = main_ref(x + idx + 1);
    if (fact) p = ((32 - fact) * p + fact * main_ref(x + idx + 2) + 16) >> 5;
    if (ang == 0 && x == 0) {  // modes 10 and 26: the first column / row follows the other side's gradient
        p += (tl[-sgn * (k + 1)] - tl[0]) >> 1;
        p = min(max(p, 0), 255);
    }
    return p;
}

template <int N>
__global__ __launch_bounds__(256) void intra_decide_kernel(const uint8_t *__restrict__ src, int src_stride, int width, int height,
                                                           const int32_t *__restrict__ xy, const uint8_t *__restrict__ refs,
                                                           int ref_stride, const uint8_t *__restrict__ mpms, uint32_t lambda,
                                                           x265amd_intra_best *__restrict__ best, uint32_t *__restrict__ sads,
                                                           uint8_t *__restrict__ pred)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    constexpr int NPIX = N * N;
    constexpr int PER = (NPIX + 63) / 64;  // pixels of a mode per lane
    __shared__ uint8_t s_ref[2][kMaxRef + 3];
    __shared__ uint32_t s_sad[35];
    __shared__ int s_dc, s_best;
    const int cu = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = xy[2 * cu], y0 = xy[2 * cu + 1];
    if (x0 < 0 || y0 < 0 || x0 + N > width || y0 + N > height) {  // not a CU of this plane: say so, touch nothing
        if (tid == 0) best[cu] = x265amd_intra_best{0xffffffffu, 0xffffffffu};
        return;
    }
    const uint8_t *ref0 = refs + (int64_t)cu * ref_stride;
    for (int i = tid; i <= 4 * N; i += 256) {
        const int a = ref0[i];
        s_ref[0][i] = (uint8_t)a;
        // [1 2 1], the two ends kept (src/encode.cpp:1260-1265)
        s_ref[1][i] = (i == 0 || i == 4 * N) ? (uint8_t)a : (uint8_t)((ref0[i - 1] + 2 * a + ref0[i + 1] + 2) >> 2);
    }
    __syncthreads();
    if (wave == 0) {  // the DC value (src/encode.cpp:1398-1416): mode 1 reads the unsmoothed samples at every size
        int s = 0;
        for (int i = lane; i < 2 * N; i += 64) s += i < N ? s_ref[0][2 * N + 1 + i] : s_ref[0][2 * N - 1 - (i - N)];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) s_dc = (s + N) / (2 * N);
    }
    // the lane's pixels of the source CU: pixel i = lane + 64 t, raster order
    int sp[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) {
        const int i = lane + 64 * t;
        sp[t] = i < NPIX ? src[(int64_t)(y0 + (i >> L2)) * src_stride + x0 + (i & (N - 1))] : 0;
    }
    __syncthreads();
    const int dc = s_dc;
    for (int mode = wave; mode < 35; mode += 4) {
        const uint8_t *ref = s_ref[kIntraFilter[L2 - 2][mode]];
        uint32_t s = 0;
#pragma unroll
        for (int t = 0; t < PER; t++) {
            const int i = lane + 64 * t;
            if (i < NPIX) {
                const int p = intra_pixel<N>(ref, mode, i >> L2, i & (N - 1), dc);
                s += (uint32_t)abs(sp[t] - p);
            }
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) s_sad[mode] = s;
    }
    __syncthreads();
    if (tid == 0) {
        // cost and argmin exactly as src/encode.cpp:725-765: first strict minimum in mode order from MAX_SAD
        const uint8_t *mpm = mpms + 3 * (int64_t)cu;
        const int m0 = mpm[0], m1 = mpm[1], m2 = mpm[2];
        uint32_t bc = X265AMD_MAX_SAD, bm = 0;
        for (int m = 0; m < 35; m++) {
            const uint32_t cost = s_sad[m] + lambda * (m == m0 ? 1u : (m == m1 || m == m2) ? 2u : 3u);
            if (cost < bc) {
                bc = cost;
                bm = (uint32_t)m;
            }
        }
        best[cu] = x265amd_intra_best{bm, bc};
        s_best = (int)bm;
    }
    if (sads && tid < 35) sads[35 * (int64_t)cu + tid] = s_sad[tid];
    if (pred) {  // workgroup-uniform
        __syncthreads();
        const int bm = s_best;
        const uint8_t *ref = s_ref[kIntraFilter[L2 - 2][bm]];
        for (int i = tid; i < NPIX; i += 256)
            pred[(int64_t)cu * NPIX + i] = (uint8_t)intra_pixel<N>(ref, bm, i >> L2, i & (N - 1), dc);
    }
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_intra_decide_luma(const uint8_t *d_src, int src_stride, int width, int height, const int32_t *d_xy,
                                         int ncus, int size, const uint8_t *d_ref, int ref_stride, const uint8_t *d_mpm,
                                         uint32_t lambda, x265amd_intra_best *d_best, uint32_t *d_sads, uint8_t *d_pred,
                                         void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_src || !d_xy || !d_ref || !d_mpm || !d_best) return X265AMD_ERR_ARG;
    if (size != 4 && size != 8 && size != 16 && size != 32) return X265AMD_ERR_ARG;  // MAX_CU_SIZE, inc/config.h:38
    if (width < size || height < size || src_stride < width || ref_stride < 4 * size + 1 || ncus < 0) return X265AMD_ERR_ARG;
    if (lambda > (1u << 20)) return X265AMD_ERR_ARG;  // cost = SAD + 3 lambda stays far below MAX_SAD
    if (ncus == 0) return X265AMD_OK;
    const dim3 grid((unsigned)ncus), block(256);
    hipStream_t st = (hipStream_t)stream;
#define X265AMD_INTRA_LAUNCH(NN)                                                                                       \
    hipLaunchKernelGGL(intra_decide_kernel<NN>, grid, block, 0, st, d_src, src_stride, width, height, d_xy, d_ref,     \
                       ref_stride, d_mpm, lambda, d_best, d_sads, d_pred)
    switch (size) {
    case 4: X265AMD_INTRA_LAUNCH(4); break;
    case 8: X265AMD_INTRA_LAUNCH(8); break;
    case 16: X265AMD_INTRA_LAUNCH(16); break;
    default: X265AMD_INTRA_LAUNCH(32); break;
    }
#undef X265AMD_INTRA_LAUNCH
    return hip_status(hipGetLastError());
}
