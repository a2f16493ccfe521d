// dropin.hip -- per-block drop-ins with the reference's OWN signatures (inc/x265.h:290-298) over HOST
// pointers, so that a maintainer can write `xSadN[4] = x265amd_xSad64xN;` or swap `xSubDct` for
// `x265amd_xSubDct` and run the encoder's CHECK_TV blocks unchanged.  Each call stages its one block
// in a small device buffer, runs the SAME kernels the batched entry points run and copies the
// result back -- a launch and two copies per block, so this layer is for bring-up and stage-by-stage
// checking; the batched entry points of x265_amd.h are the product.  No CPU arithmetic in here.
#include <atomic>
#include <mutex>

#include "../../include/x265_amd_dropin.h"
#include "common.h"
#include "hevc_transform.h"

namespace x265amd {

// one 1-D pass of xDctN[] / xInvDctN[] on compact blocks: forward reads `lines` rows of N and writes
// the transposed N x lines block with a truncating store (src/pixel.cpp:126-129); inverse reads the
// columns of an N x lines block and writes `lines` rows of N, store-then-clip (src/pixel.cpp:334-342:
// the clip acts on the already truncated Int16, so it is a wrap).
template <int N, bool INV, bool DST>
__global__ void pass1d_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, int lines, int shift)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lines) return;
    int in[N], out[N];
#pragma unroll
    for (int k = 0; k < N; k++) in[k] = INV ? src[k * lines + i] : src[i * N + k];
    if constexpr (DST) {
        if constexpr (INV)
            inv_dst4(in, out);
        else
            fwd_dst4(in, out);
    } else {
        if constexpr (INV)
            inv_butterfly<N>(in, out);
        else
            fwd_butterfly<N>(in, out);
    }
    const int rnd = 1 << (shift - 1);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int16_t v = (int16_t)((out[k] + rnd) >> shift);
        if (INV)
            dst[i * N + k] = v;
        else
            dst[k * lines + i] = v;
    }
}

namespace {

constexpr int kMaxLines = 64;
struct Stage {  // device-side staging of one block
    uint8_t src[64 * 64];
    uint8_t ref[64 * 64];
    int16_t a[32 * kMaxLines];
    int16_t b[32 * kMaxLines];
    uint32_t zero[2];
    uint32_t out[2];
    uint8_t mode[4];
};

std::mutex g_mu;                   // one drop-in call at a time (they are synchronous, null-stream helpers)
thread_local Stage *g_stage = nullptr;  // the staging block of the device current on this thread (set by stage_ready)
thread_local int t_status = X265AMD_OK;
std::atomic<uint64_t> g_calls{0};  // calls that reached the GPU and came back ok

// one staging block per device, allocated at the first drop-in call on that device, freed by
// x265amd_dropin_teardown().  Callers hold g_mu.
int stage_ready()
{
    DeviceState *ds = device_state();
    if (!ds) return X265AMD_ERR_NOT_INIT;
    if (!ds->dropin_stage) {
        void *p = nullptr;
        X265AMD_HIP(hipMalloc(&p, sizeof(Stage)));
        const hipError_t e = hipMemset(p, 0, sizeof(Stage));
        if (e != hipSuccess) {
            (void)hipFree(p);
            return hip_status(e);
        }
        ds->dropin_stage = p;
    }
    g_stage = (Stage *)ds->dropin_stage;
    return X265AMD_OK;
}

#define DROPIN_TRY(expr)                \
    do {                                \
        const int s_ = (expr);          \
        if (s_ != X265AMD_OK) return s_; \
    } while (0)

// rows x cols elements of `esz` bytes between a strided host block and a compact device block
int h2d(void *d, const void *h, size_t hstride_elems, int cols, int rows, int esz)
{
    return hip_status(hipMemcpy2D(d, (size_t)cols * esz, h, hstride_elems * esz, (size_t)cols * esz, rows,
                                  hipMemcpyHostToDevice));
}
int d2h(void *h, size_t hstride_elems, const void *d, int cols, int rows, int esz)
{
    return hip_status(hipMemcpy2D(h, hstride_elems * esz, d, (size_t)cols * esz, (size_t)cols * esz, rows,
                                  hipMemcpyDeviceToHost));
}

bool tu_size(int w, int h) { return w == h && (w == 4 || w == 8 || w == 16 || w == 32); }

int sad(int w, unsigned n, const uint8_t *src, unsigned ss, const uint8_t *ref, unsigned sr, uint32_t *out)
{
    if (!src || !ref || n < 1 || n > 64 || ss < (unsigned)w || sr < (unsigned)w) return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    DROPIN_TRY(h2d(g_stage->src, src, ss, w, (int)n, 1));
    DROPIN_TRY(h2d(g_stage->ref, ref, sr, w, (int)n, 1));
    DROPIN_TRY(x265amd_sad_blocks(w, (int)n, g_stage->src, w, g_stage->ref, w, g_stage->zero, g_stage->zero + 1, 1,
                                  g_stage->out, nullptr));
    return hip_status(hipMemcpy(out, g_stage->out, 4, hipMemcpyDeviceToHost));
}

template <int N, bool INV, bool DST>
int pass1d(int16_t *dst, const int16_t *src, unsigned stride, int lines, int shift)
{
    if (DST) lines = 4;             // xDST4 / xInvDST4 loop over 4 lines whatever nLines says (src/pixel.cpp:138,351)
    if (INV && N == 32) lines = 32;  // and so does xInvDCT32 over 32 (src/pixel.cpp:513)
    if (!dst || !src || lines < 1 || lines > kMaxLines || shift < 1 || shift > 30 || stride < (unsigned)N ||
        stride < (unsigned)lines)
        return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    // forward: `lines` rows of N in, N rows of `lines` out; inverse: the other way round
    DROPIN_TRY(INV ? h2d(g_stage->a, src, stride, lines, N, 2) : h2d(g_stage->a, src, stride, N, lines, 2));
    hipLaunchKernelGGL((pass1d_kernel<N, INV, DST>), dim3(1), dim3(kMaxLines), 0, nullptr, g_stage->a, g_stage->b,
                       lines, shift);
    DROPIN_TRY(hip_status(hipGetLastError()));
    return INV ? d2h(dst, stride, g_stage->b, N, lines, 2) : d2h(dst, stride, g_stage->b, lines, N, 2);
}

x265amd_tq_desc block_desc(int n)
{
    x265amd_tq_desc t = {};
    t.width = t.height = t.tu = n;
    t.qp = 0;
    t.slice_type = X265AMD_SLICE_I;
    t.nframes = 1;
    t.d_src = g_stage->src;
    t.d_pred = g_stage->ref;
    t.src_stride = t.pred_stride = t.level_stride = t.rec_stride = n;
    t.d_modes = g_stage->mode;
    return t;
}

int sub_dct(int16_t *dst, const uint8_t *src, const uint8_t *ref, unsigned stride, int w, int h, unsigned mode)
{
    if (!dst || !src || !ref || !tu_size(w, h) || stride < (unsigned)w || mode > 255) return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    const uint8_t m = (uint8_t)mode;
    DROPIN_TRY(h2d(g_stage->src, src, stride, w, h, 1));
    DROPIN_TRY(h2d(g_stage->ref, ref, stride, w, h, 1));
    DROPIN_TRY(hip_status(hipMemcpy(g_stage->mode, &m, 1, hipMemcpyHostToDevice)));
    x265amd_tq_desc t = block_desc(w);
    t.d_level = g_stage->b;  // by-product of the fused kernel, not returned
    t.d_coef = g_stage->a;
    DROPIN_TRY(x265amd_tq_forward(&t, nullptr));
    return d2h(dst, stride, g_stage->a, w, h, 2);
}

int idct_add(uint8_t *dst, const int16_t *src, const uint8_t *ref, unsigned stride, int w, int h, unsigned mode)
{
    if (!dst || !src || !ref || !tu_size(w, h) || stride < (unsigned)w || mode > 255) return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    const uint8_t m = (uint8_t)mode;
    DROPIN_TRY(h2d(g_stage->a, src, stride, w, h, 2));
    DROPIN_TRY(h2d(g_stage->ref, ref, stride, w, h, 1));
    DROPIN_TRY(hip_status(hipMemcpy(g_stage->mode, &m, 1, hipMemcpyHostToDevice)));
    x265amd_tq_desc t = block_desc(w);
    t.d_level = g_stage->a;  // dequantised coefficients: x265amd_idct_add skips xDeQuant
    t.d_rec = g_stage->src;
    DROPIN_TRY(x265amd_idct_add(&t, nullptr));
    return d2h(dst, stride, g_stage->src, w, h, 1);
}

int quant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h, int st, uint32_t *sum)
{
    if (!dst || !src || !tu_size(w, h) || stride < (unsigned)w || qp > 51) return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    DROPIN_TRY(h2d(g_stage->a, src, stride, w, h, 2));
    DROPIN_TRY(x265amd_quant_plane(g_stage->a, g_stage->b, w, w, h, w, (int)qp, st, g_stage->out, nullptr));
    DROPIN_TRY(d2h(dst, stride, g_stage->b, w, h, 2));
    return hip_status(hipMemcpy(sum, g_stage->out, 4, hipMemcpyDeviceToHost));
}

int dequant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h)
{
    if (!dst || !src || !tu_size(w, h) || stride < (unsigned)w || qp > 51) return X265AMD_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    DROPIN_TRY(stage_ready());
    DROPIN_TRY(h2d(g_stage->a, src, stride, w, h, 2));
    DROPIN_TRY(x265amd_dequant_plane(g_stage->a, g_stage->b, w, w, h, w, (int)qp, nullptr));
    return d2h(dst, stride, g_stage->b, w, h, 2);
}

}  // namespace
}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_dropin_status(void) { return t_status; }
extern "C" int x265amd_dropin_teardown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceState *ds = device_state();
    if (!ds) return X265AMD_ERR_NOT_INIT;
    if (ds->dropin_stage) {
        X265AMD_HIP(hipDeviceSynchronize());
        X265AMD_HIP(hipFree(ds->dropin_stage));
        ds->dropin_stage = nullptr;
    }
    g_stage = nullptr;
    return X265AMD_OK;
}
extern "C" uint64_t x265amd_dropin_calls(void) { return g_calls.load(); }

static int served(int status)
{
    if (status == X265AMD_OK) g_calls++;
    return status;
}

#define DROPIN_SAD(W)                                                                                         \
    extern "C" uint32_t x265amd_xSad##W##xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc,  \
                                            const uint8_t *pRef, const unsigned nStrideRef)                   \
    {                                                                                                         \
        uint32_t v = 0xffffffffu;                                                                             \
        t_status = served(sad(W, N, pSrc, nStrideSrc, pRef, nStrideRef, &v));                                        \
        return t_status == X265AMD_OK ? v : 0xffffffffu;                                                      \
    }
DROPIN_SAD(4)
DROPIN_SAD(8)
DROPIN_SAD(16)
DROPIN_SAD(32)
DROPIN_SAD(64)

#define DROPIN_1D(NAME, N, INV, DST)                                                                      \
    extern "C" void x265amd_##NAME(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift) \
    {                                                                                                     \
        t_status = served(pass1d<N, INV, DST>(pDst, pSrc, nStride, nLines, nShift));                         \
    }
DROPIN_1D(xDST4, 4, false, true)
DROPIN_1D(xDCT4, 4, false, false)
DROPIN_1D(xDCT8, 8, false, false)
DROPIN_1D(xDCT16, 16, false, false)
DROPIN_1D(xDCT32, 32, false, false)
DROPIN_1D(xInvDST4, 4, true, true)
DROPIN_1D(xInvDCT4, 4, true, false)
DROPIN_1D(xInvDCT8, 8, true, false)
DROPIN_1D(xInvDCT16, 16, true, false)
DROPIN_1D(xInvDCT32, 32, true, false)

extern "C" void x265amd_xSubDct(int16_t *pDst, uint8_t *pSrc, uint8_t *pRef, unsigned nStride, int16_t *piTmp0,
                                int16_t *piTmp1, int iWidth, int iHeight, unsigned nMode)
{
    (void)piTmp0;  // the reference's scratch: the intermediate never leaves the chip here
    (void)piTmp1;
    t_status = served(sub_dct(pDst, pSrc, pRef, nStride, iWidth, iHeight, nMode));
}

extern "C" void x265amd_xIDctAdd(uint8_t *pDst, int16_t *pSrc, uint8_t *pRef, unsigned nStride, int16_t *piTmp0,
                                 int16_t *piTmp1, int iWidth, int iHeight, unsigned nMode)
{
    (void)piTmp0;
    (void)piTmp1;
    t_status = served(idct_add(pDst, pSrc, pRef, nStride, iWidth, iHeight, nMode));
}

extern "C" uint32_t x265amd_xQuant(int16_t *pDst, int16_t *pSrc, unsigned nStride, unsigned nQP, int iWidth,
                                   int iHeight, int eSType)
{
    uint32_t sum = 0;
    t_status = served(quant(pDst, pSrc, nStride, nQP, iWidth, iHeight, eSType, &sum));
    return t_status == X265AMD_OK ? sum : 0;
}

extern "C" void x265amd_xDeQuant(int16_t *pDst, int16_t *pSrc, unsigned nStride, unsigned nQP, int iWidth,
                                 int iHeight, int eSType)
{
    (void)eSType;  // unused by the reference too (src/pixel.cpp:635-651)
    t_status = served(dequant(pDst, pSrc, nStride, nQP, iWidth, iHeight));
}
