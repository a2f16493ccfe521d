// pack_levels.hip -- compaction of quantised-level planes for the trip to the host.
//
// What the host's entropy coder walks per transform unit is its coded-block flag and its NON-ZERO levels
// (xEncodeCoeffNxN, src/encode.cpp:270-461; bCbf from xQuant's sum, :888-890); after quantisation at working
// QPs most levels are zero, and the dense Int16 planes (3 bytes per pixel at 4:2:0) are what bounds a host loop
// over PCIe.  x265amd_pack_levels leaves, per TU, where its non-zero levels are and how many (an entry of the TU
// table), and the levels themselves as (position in the TU's raster << 16 | level) words, the TU's own in raster
// order and contiguous; runs of TUs take their room from one cursor in the order they finish (the table says where).
//
// HBM-bound in the worst case (every level read once: 2 B per sample); with the per-TU sums that the transform
// stage already wrote, a TU without a level is not even read.
#include "common.h"
#include "me_common.h"  // the DPP wave scan

namespace x265amd {

struct PackParams {
    const int16_t *level;
    const uint32_t *abs_sum;  // per TU, or null
    uint32_t *table;          // [nframes][ntu][2]: first word of the TU in `words`, number of words
    uint32_t *words;
    uint32_t *cursor;
    int width, height, stride, tu_log2, nframes, tus_x, ntu;
    int64_t frame_stride, abs_sum_frame_stride;
    uint32_t capacity;
};

// A wave takes kTusPerWave consecutive TUs, one after the other.  A lane reads chunks of CH consecutive levels of a
// row (16 bytes; 8 for the 4x4 TU): chunk c of the TU's raster is lane c % 64's in pass c / 64, so a wave's loads cover
// whole rows and the raster order of the levels is (pass, lane, place in the chunk).  The words go in order into the
// wave's stretch of LDS, TU after TU; when the next TU would not fit (and at the end) the wave asks the cursor for
// room ONCE for all it holds -- the cursor is one address, and an atomic per TU was what the kernel's time consisted
// of (11 ns each, one after the other) -- writes the table entries of those TUs and lets the words go a lane per
// word: the stores to HBM are contiguous whatever the TUs' patterns are.
constexpr int kTusPerWave = 16;
constexpr int kStageWords = 2048;  // per wave; >= 32 * 32, the largest TU

template <int N, bool ALIGNED>
__global__ __launch_bounds__(256) void pack_levels_kernel(const PackParams p)
{
    constexpr int CH = N >= 8 ? 8 : 4;
    constexpr int CPR = N / CH;       // chunks per row
    constexpr int NCH = N * N / CH;   // 128, 32, 8, 4
    constexpr int PASSES = NCH > 64 ? NCH / 64 : 1;
    static_assert(PASSES <= 2, "the scan carries one 16-bit count per pass");
    __shared__ uint32_t stage_all[4][kStageWords];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *stage = stage_all[wave];
    const int64_t ntotal = (int64_t)p.nframes * p.ntu;
    const int64_t k0 = ((int64_t)blockIdx.x * 4 + wave) * kTusPerWave;  // TUs in (frame, raster) order: wave-uniform
    uint32_t used = 0;                        // words in the stage
    int held = 0;                             // TUs in the stage: lane i keeps the i-th one's entry
    int64_t my_k = 0;
    uint32_t my_off = 0, my_cnt = 0;

    auto flush = [&]() __attribute__((always_inline)) {
        if (held == 0) return;  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(p.cursor, used);  // keeps counting past the capacity: the host sees what was needed
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < held) {
            p.table[2 * my_k] = base + my_off;
            p.table[2 * my_k + 1] = my_cnt;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < used; i += 64)
            if ((uint64_t)base + i < p.capacity) p.words[base + i] = stage[i];
        __builtin_amdgcn_wave_barrier();
        used = 0, held = 0;
    };

    // The TUs' sums, all at once (lane i: TU k0 + i): which of them are coded is known before the first level is read.
    uint32_t my_sum = 0;
    if (lane < kTusPerWave && k0 + lane < ntotal) {
        my_sum = 1u;
        if (p.abs_sum) {
            const int64_t kk = k0 + lane;
            const int f = (int)(kk / p.ntu), t = (int)(kk - (int64_t)f * p.ntu);
            my_sum = p.abs_sum[(int64_t)f * p.abs_sum_frame_stride + t];
        }
    }
    // the lane's chunks of TU k, as they lie in memory (pairs of levels per dword); zeros where the lane has no chunk
    auto load_tu = [&](int64_t k, uint32_t (&raw)[PASSES][CH / 2]) __attribute__((always_inline)) {
        const int f = (int)(k / p.ntu), t = (int)(k - (int64_t)f * p.ntu);
        const int ty = t / p.tus_x, tx = t - ty * p.tus_x;
        const int16_t *tu0 = p.level + f * p.frame_stride + (int64_t)(ty * N) * p.stride + tx * N;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int c = lane + 64 * ps;
#pragma unroll
            for (int j = 0; j < CH / 2; j++) raw[ps][j] = 0u;
            if (c < NCH) {
                const int16_t *src = tu0 + (int64_t)(c / CPR) * p.stride + (c % CPR) * CH;
                if constexpr (ALIGNED) {  // rows of the TU start on CH * 2-byte boundaries (the entry point checked)
                    if constexpr (CH == 8) {
                        const uint4 q = *reinterpret_cast<const uint4 *>(src);
                        raw[ps][0] = q.x, raw[ps][1] = q.y, raw[ps][2] = q.z, raw[ps][3] = q.w;
                    } else {
                        const uint2 q = *reinterpret_cast<const uint2 *>(src);
                        raw[ps][0] = q.x, raw[ps][1] = q.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CH / 2; j++) raw[ps][j] = (uint32_t)(uint16_t)src[2 * j] | ((uint32_t)(uint16_t)src[2 * j + 1] << 16);
                }
            }
        }
    };
    // One TU ahead: the loads of the next coded TU are in flight while this one is counted, scanned and staged (a wave
    // walks its TUs one after the other; without this each of them cost a memory latency of its own).
    uint32_t nxt[PASSES][CH / 2];
    const int ntus = (int)min((int64_t)kTusPerWave, ntotal - k0);
    const unsigned long long coded = __ballot(my_sum != 0u);  // bit i: TU k0 + i has levels (or is not known not to)
    int it = coded ? __ffsll((long long)coded) - 1 : ntus;
    if (lane < ntus && my_sum == 0u) p.table[2 * (k0 + lane)] = 0u, p.table[2 * (k0 + lane) + 1] = 0u;
    if (it < ntus) load_tu(k0 + it, nxt);
#pragma unroll 1
    while (it < ntus) {
        const int64_t k = k0 + it;
        uint32_t raw[PASSES][CH / 2];
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++)
#pragma unroll
            for (int j = 0; j < CH / 2; j++) raw[ps][j] = nxt[ps][j];
        const unsigned long long rest = it + 1 < 64 ? coded >> (it + 1) : 0ull;
        it = rest ? it + 1 + (__ffsll((long long)rest) - 1) : ntus;
        if (it < ntus) load_tu(k0 + it, nxt);
        int v[PASSES][CH];
        uint32_t packed = 0;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            uint32_t cnt = 0;
#pragma unroll
            for (int j = 0; j < CH / 2; j++) {
                v[ps][2 * j] = (int)(int16_t)(raw[ps][j] & 0xffffu);
                v[ps][2 * j + 1] = (int)(int16_t)(raw[ps][j] >> 16);
            }
#pragma unroll
            for (int j = 0; j < CH; j++) cnt += v[ps][j] != 0;
            packed |= cnt << (16 * ps);  // at most 8 * 64 = 512 a pass: the two counts scan as one word
        }
        const uint32_t inc = me_wave_scan_u32(packed, lane);  // (all 64 lanes are here: the branches above are wave-uniform)
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), tot0 = tot & 0xffffu, total = tot0 + (tot >> 16);
        if (total == 0) {  // wave-uniform
            if (lane == 0) p.table[2 * k] = 0u, p.table[2 * k + 1] = 0u;
            continue;
        }
        if (used + total > (uint32_t)kStageWords) flush();
        const uint32_t before = inc - packed;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            uint32_t slot = used + (ps ? tot0 + (before >> 16) : (before & 0xffffu));
            const int e0 = (lane + 64 * ps) * CH;
#pragma unroll
            for (int j = 0; j < CH; j++)
                if (v[ps][j] != 0) stage[slot++] = ((uint32_t)(e0 + j) << 16) | ((uint32_t)v[ps][j] & 0xffffu);
        }
        if (lane == held) my_k = k, my_off = used, my_cnt = total;
        held++;
        used += total;
    }
    flush();
}

}  // namespace x265amd

using namespace x265amd;

// The header invites page-locked host memory for the table and the words: make sure that what came is memory a kernel can
// write (device, or host memory the runtime has mapped) -- a pageable host pointer here would be a GPU fault, not an error code.
static bool kernel_can_write(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged;
}

extern "C" int x265amd_pack_levels_begin(uint32_t *d_cursor, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_cursor) return X265AMD_ERR_ARG;
    X265AMD_HIP(hipMemsetAsync(d_cursor, 0, sizeof(uint32_t), (hipStream_t)stream));
    return X265AMD_OK;
}

extern "C" int x265amd_pack_levels(const int16_t *d_level, int width, int height, int stride, int tu, int nframes,
                                   int64_t frame_stride, const uint32_t *d_abs_sum, int64_t abs_sum_frame_stride,
                                   uint32_t *d_tu_table, uint32_t *d_words, uint32_t capacity, uint32_t *d_cursor, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_level || !d_tu_table || !d_cursor || (!d_words && capacity)) return X265AMD_ERR_ARG;
    if (tu != 4 && tu != 8 && tu != 16 && tu != 32) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % tu || height % tu || stride < width || nframes < 0) return X265AMD_ERR_ARG;
    if ((uintptr_t)d_level & 1) return X265AMD_ERR_ARG;
    if (nframes == 0) return X265AMD_OK;
    if (!kernel_can_write(d_tu_table) || (capacity && !kernel_can_write(d_words))) return X265AMD_ERR_ARG;
    PackParams p;
    p.level = d_level;
    p.abs_sum = d_abs_sum;
    p.table = d_tu_table;
    p.words = d_words;
    p.cursor = d_cursor;
    p.width = width, p.height = height, p.stride = stride, p.tu_log2 = ilog2(tu), p.nframes = nframes;
    p.tus_x = width / tu;
    p.ntu = (width / tu) * (height / tu);
    p.frame_stride = frame_stride;
    p.abs_sum_frame_stride = abs_sum_frame_stride;
    p.capacity = capacity;
    const int64_t n = (int64_t)nframes * p.ntu;
    if (n > 0x7fffffff) return X265AMD_ERR_ARG;
    const dim3 grid((unsigned)((n + 4 * kTusPerWave - 1) / (4 * kTusPerWave))), block(256);
    hipStream_t st = (hipStream_t)stream;
    // a lane's loads are 16 bytes (8 for the 4x4 TU): aligned when the plane's rows are
    const int al = tu == 4 ? 8 : 16;
    const bool aligned = ((uintptr_t)d_level % al) == 0 && (stride * 2) % al == 0 && (frame_stride * 2) % al == 0;
#define X265AMD_PACK(N)                                                                               \
    if (aligned)                                                                                      \
        hipLaunchKernelGGL((pack_levels_kernel<N, true>), grid, block, 0, st, p);                     \
    else                                                                                              \
        hipLaunchKernelGGL((pack_levels_kernel<N, false>), grid, block, 0, st, p)
    switch (tu) {
    case 4: X265AMD_PACK(4); break;
    case 8: X265AMD_PACK(8); break;
    case 16: X265AMD_PACK(16); break;
    default: X265AMD_PACK(32); break;
    }
#undef X265AMD_PACK
    return hip_status(hipGetLastError());
}
