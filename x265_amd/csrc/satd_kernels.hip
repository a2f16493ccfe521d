// satd_kernels.hip -- SATD (sum of absolute Hadamard-transformed differences) of 8-bit blocks for gfx950.
//
// The reference has NO SATD (BASELINE.json's north_star names it as a primitive the
// search may be built on).  The definition used is the one the HEVC test model publishes
// (TComRdCost::xCalcHADs8x8 / xCalcHADs4x4): per 8x8 sub-block (sum |H8 D H8^T| + 2) >> 2, per 4x4
// sub-block (sum |H4 D H4^T| + 1) >> 1 when the block is only 4 wide or 4 high, summed over the block.
// PARITY UNPINNED: the checker is oracle/x265_oracle.c:orc_satd (a matrix restatement) plus closed-form cases.
//
// A lane takes one sub-block: the differences as packed Int16 pairs (|d| <= 255, five butterfly stages
// keep |x| <= 8160), the vertical stages and the two wide horizontal stages as v_pk_add/sub_i16 between
// registers, the last horizontal stage never formed: |a + b| + |a - b| = 2 max(|a|, |b|).
#include "common.h"

namespace x265amd {

typedef short v2s __attribute__((ext_vector_type(2)));

struct __attribute__((packed, aligned(1))) U32u {
    uint32_t v;
};

__device__ __forceinline__ v2s diff_pair(uint32_t a, uint32_t b, int hi)
{
    const int sh = 16 * hi;
    return v2s{(short)(int)((a >> sh) & 0xffu), (short)(int)((a >> (sh + 8)) & 0xffu)} -
           v2s{(short)(int)((b >> sh) & 0xffu), (short)(int)((b >> (sh + 8)) & 0xffu)};
}

// 2 * max(|lo|, |hi|) of a packed pair = |lo + hi| + |lo - hi|
__device__ __forceinline__ uint32_t last_stage(v2s x)
{
    const v2s a = __builtin_elementwise_max(x, v2s{0, 0} - x);
    const uint32_t u = __builtin_bit_cast(uint32_t, a);
    return 2u * max(u & 0xffffu, u >> 16);
}

template <int N> __device__ __forceinline__ void hadamard_regs(v2s *x, int stride)  // in place, N a power of two
{
#pragma unroll
    for (int h = N / 2; h >= 1; h >>= 1)
#pragma unroll
        for (int i = 0; i < N; i++)
            if (!(i & h)) {
                const v2s a = x[i * stride], b = x[(i + h) * stride];
                x[i * stride] = a + b;
                x[(i + h) * stride] = a - b;
            }
}

__device__ __forceinline__ uint32_t satd8x8(const uint8_t *s, int ss, const uint8_t *r, int sr)
{
    v2s p[8][4];  // row, column pair
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const uint32_t a0 = reinterpret_cast<const U32u *>(s + (int64_t)y * ss)->v, a1 = reinterpret_cast<const U32u *>(s + (int64_t)y * ss + 4)->v;
        const uint32_t b0 = reinterpret_cast<const U32u *>(r + (int64_t)y * sr)->v, b1 = reinterpret_cast<const U32u *>(r + (int64_t)y * sr + 4)->v;
        p[y][0] = diff_pair(a0, b0, 0);
        p[y][1] = diff_pair(a0, b0, 1);
        p[y][2] = diff_pair(a1, b1, 0);
        p[y][3] = diff_pair(a1, b1, 1);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) hadamard_regs<8>(&p[0][j], 4);  // vertical, both columns of the pair at once
    uint32_t sum = 0;
#pragma unroll
    for (int y = 0; y < 8; y++) {
        hadamard_regs<4>(&p[y][0], 1);  // horizontal distances 4 and 2 (in pairs: 2 and 1)
#pragma unroll
        for (int j = 0; j < 4; j++) sum += last_stage(p[y][j]);
    }
    return (sum + 2) >> 2;
}

__device__ __forceinline__ uint32_t satd4x4(const uint8_t *s, int ss, const uint8_t *r, int sr)
{
    v2s p[4][2];
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint32_t a = reinterpret_cast<const U32u *>(s + (int64_t)y * ss)->v, b = reinterpret_cast<const U32u *>(r + (int64_t)y * sr)->v;
        p[y][0] = diff_pair(a, b, 0);
        p[y][1] = diff_pair(a, b, 1);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) hadamard_regs<4>(&p[0][j], 2);
    uint32_t sum = 0;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        hadamard_regs<2>(&p[y][0], 1);
#pragma unroll
        for (int j = 0; j < 2; j++) sum += last_stage(p[y][j]);
    }
    return (sum + 1) >> 1;
}

// one wave per block pair, one lane per sub-block (a 64x64 block has 64 sub-blocks of 8x8)
__global__ __launch_bounds__(256) void satd_blocks_kernel(int w, int n, const uint8_t *src, int ss, const uint8_t *ref,
                                                          int sr, const uint32_t *src_off, const uint32_t *ref_off,
                                                          int nblocks, uint32_t *out)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= nblocks) return;  // wave-uniform
    const uint8_t *s = src + src_off[b];
    const uint8_t *r = ref + ref_off[b];
    const int sub = (w == 4 || n == 4) ? 4 : 8;
    const int nsx = w / sub, nsub = nsx * (n / sub);
    uint32_t acc = 0;
    for (int i = lane; i < nsub; i += 64) {
        const int y = (i / nsx) * sub, x = (i % nsx) * sub;
        const uint8_t *sp = s + (int64_t)y * ss + x, *rp = r + (int64_t)y * sr + x;
        acc += sub == 8 ? satd8x8(sp, ss, rp, sr) : satd4x4(sp, ss, rp, sr);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc;
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_satd_blocks(int w, int n, const uint8_t *d_src, int stride_src, const uint8_t *d_ref,
                                   int stride_ref, const uint32_t *d_src_off, const uint32_t *d_ref_off, int nblocks,
                                   uint32_t *d_out, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (w != 4 && w != 8 && w != 16 && w != 32 && w != 64) return X265AMD_ERR_ARG;
    if (n != 4 && n != 8 && n != 16 && n != 32 && n != 64) return X265AMD_ERR_ARG;
    if (nblocks < 0 || !d_src || !d_ref || !d_src_off || !d_ref_off || !d_out) return X265AMD_ERR_ARG;
    if (nblocks == 0) return X265AMD_OK;
    hipLaunchKernelGGL(satd_blocks_kernel, dim3((nblocks + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, n, d_src,
                       stride_src, d_ref, stride_ref, d_src_off, d_ref_off, nblocks, d_out);
    return hip_status(hipGetLastError());
}
