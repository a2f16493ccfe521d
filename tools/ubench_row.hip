// ubench_row.hip -- what the exact-SAD round of me_sea.hip costs per row on gfx950, piece by piece, with
// the kernel's own two-row software pipeline: 16 dependent v_qsad_pk_u16_u8 per row, the 17-dword window
// row from LDS (ds_read2_b32 + the v_pk_mov_b32 that form the odd dword pairs), the current row by
// s_load_dwordx16.  16 waves per CU like the kernel.  Not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_row.hip -o tools/ubench_row
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int BLOCKS = 256;  // 16-row blocks per wave
constexpr int PITCH = 65;

template <int LDS, int SLOAD, int WAVES, int PACKED = 0>
__global__ __launch_bounds__(WAVES * 64) void k(uint32_t *__restrict__ out, const uint32_t *__restrict__ cur, int cstr,
                                                uint32_t seed, int pitch)
{
    __shared__ uint32_t win[200 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 200 * PITCH; i += WAVES * 64) win[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc32[4] = {0u, 0u, 0u, 0u};
    uint32_t dA[17], dB[17], SA[16], SB[16];
#pragma unroll
    for (int k2 = 0; k2 < 17; k2++) dA[k2] = dB[k2] = tid * 77u + k2;
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) SA[k2] = SB[k2] = seed + k2;
    const uint64_t c0 = clock64();
    for (int blk = 0; blk < BLOCKS; blk++) {
        const uint32_t *wrow = win + (lane & 31) + ((lane >> 5) + (wave & 3) + (blk & 63)) * pitch;
        const uint32_t *crow = cur + (wave & 3) * 16 * cstr;
        uint64_t acc16 = 0;
#define SEA_LOAD(r, d, S)                                                                     \
    {                                                                                         \
        if (LDS) {                                                                            \
            const uint32_t *p_ = wrow + (r) * pitch;                                          \
            _Pragma("unroll") for (int k2 = 0; k2 < 17; k2++) d[k2] = p_[k2];                 \
        }                                                                                     \
        if (SLOAD) { _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) S[k2] = crow[(r) * (PACKED ? 16 : cstr) + k2]; } \
    }
#define SEA_ROW(d, S)                                                                                   \
    _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++)                                                  \
        acc16 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k2 + 1] << 32) | d[k2], S[k2], acc16);
        SEA_LOAD(0, dA, SA)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            SEA_LOAD(r + 1, dB, SB)
            __builtin_amdgcn_sched_barrier(0);
            SEA_ROW(dA, SA)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (r + 2 < 16) SEA_LOAD(r + 2, dA, SA)
            __builtin_amdgcn_sched_barrier(0);
            SEA_ROW(dB, SB)
            __builtin_amdgcn_sched_barrier(0);
            if ((r & 3) == 2) {
#pragma unroll
                for (int q = 0; q < 4; q++) acc32[q] += (uint32_t)(acc16 >> (16 * q)) & 0xffffu;
                acc16 = 0;
            }
        }
#undef SEA_LOAD
#undef SEA_ROW
    }
    const uint64_t c1 = clock64();
    const uint32_t a = acc32[0] ^ acc32[1] ^ acc32[2] ^ acc32[3];
    if (a == 0x12345678u) out[tid] = a;
    if (blockIdx.x == 7 && lane == 0) ((uint64_t *)out)[512 + wave] = c1 - c0;
}

template <int LDS, int SLOAD, int WAVES, int PACKED = 0> static void run(const char *name, uint32_t *d_out, uint32_t *d_cur, int pitch)
{
    uint64_t h[16];
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k<LDS, SLOAD, WAVES, PACKED>), dim3(256), dim3(WAVES * 64), 0, 0, d_out, d_cur, 480, 1u + rep, pitch);
    hipDeviceSynchronize();
    hipMemcpy(h, (uint64_t *)d_out + 512, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t mx = 0;
    for (int i = 0; i < WAVES; i++) mx = h[i] > mx ? h[i] : mx;
    const double per_row = (double)mx / (BLOCKS * 16) / (WAVES / 4);
    printf("%-44s %2d waves/CU: %6.1f SIMD cycles per wave-row (16 qsad = 258 at the issue limit: %4.1f %%)\n", name, WAVES,
           per_row, 100.0 * 258.0 / per_row);
}

// Two dy per lane: window row r pairs with current row r (candidate dy) and r-1 (candidate dy+1); the
// LDS row and the s_load are paid once per 32 qsads.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k2(uint32_t *__restrict__ out, const uint32_t *__restrict__ cur, int cstr,
                                                 uint32_t seed, int pitch)
{
    __shared__ uint32_t win[200 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 200 * PITCH; i += WAVES * 64) win[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc32[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t dA[17], dB[17], S0[16], S1[16], S2[16];
    const uint64_t c0 = clock64();
    for (int blk = 0; blk < BLOCKS; blk++) {
        const uint32_t *wrow = win + (lane & 31) + ((lane >> 5) + (wave & 3) + (blk & 63)) * pitch;
        const uint32_t *crow = cur + (wave & 3) * 16 * cstr;
        uint64_t a0 = 0, a1 = 0;
#define LD_W(r, d) { const uint32_t *p_ = wrow + (r) * pitch; _Pragma("unroll") for (int k2 = 0; k2 < 17; k2++) d[k2] = p_[k2]; }
#define LD_C(r, S) { _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) S[k2] = crow[(r) * cstr + k2]; }
#define ROW2(d, Sa, Sb)                                                                                      \
    _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) {                                                     \
        const uint64_t x_ = ((uint64_t)d[k2 + 1] << 32) | d[k2];                                             \
        a0 = __builtin_amdgcn_qsad_pk_u16_u8(x_, Sa[k2], a0);                                                \
        a1 = __builtin_amdgcn_qsad_pk_u16_u8(x_, Sb[k2], a1);                                                \
    }
#define FLUSH()                                                                                              \
    { _Pragma("unroll") for (int q = 0; q < 4; q++) { acc32[q] += (uint32_t)(a0 >> (16 * q)) & 0xffffu; acc32[4 + q] += (uint32_t)(a1 >> (16 * q)) & 0xffffu; } a0 = a1 = 0; }
        LD_W(0, dA) LD_C(0, S0) LD_C(1, S1)
        // rows in triples so that the three current-row buffers rotate with static names
#pragma unroll
        for (int r = 0; r < 15; r += 3) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            LD_W(r + 1, dB) LD_C(r + 2, S2)
            __builtin_amdgcn_sched_barrier(0);
            ROW2(dA, S1, S0)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            LD_W(r + 2, dA) LD_C(r + 3, S0)
            __builtin_amdgcn_sched_barrier(0);
            ROW2(dB, S2, S1)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            LD_W(r + 3, dB) LD_C(r + 4, S1)
            __builtin_amdgcn_sched_barrier(0);
            ROW2(dA, S0, S2)
            __builtin_amdgcn_sched_barrier(0);
            FLUSH()
#pragma unroll
            for (int k2 = 0; k2 < 17; k2++) dA[k2] = dB[k2];
        }
#undef LD_W
#undef LD_C
#undef ROW2
#undef FLUSH
    }
    const uint64_t c1 = clock64();
    uint32_t a = 0;
    for (int q = 0; q < 8; q++) a ^= acc32[q];
    if (a == 0x12345678u) out[tid] = a;
    if (blockIdx.x == 7 && lane == 0) ((uint64_t *)out)[512 + wave] = c1 - c0;
}

template <int WAVES> static void run2(const char *name, uint32_t *d_out, uint32_t *d_cur, int pitch)
{
    uint64_t h[16];
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k2<WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d_out, d_cur, 480, 1u + rep, pitch);
    hipDeviceSynchronize();
    hipMemcpy(h, (uint64_t *)d_out + 512, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t mx = 0;
    for (int i = 0; i < WAVES; i++) mx = h[i] > mx ? h[i] : mx;
    // 15 window rows x 32 qsads per block = 30 row-equivalents of 16 qsads
    const double per_row = (double)mx / (BLOCKS * 30) / (WAVES / 4);
    printf("%-44s %2d waves/CU: %6.1f SIMD cycles per 16 qsads (258 at the issue limit: %4.1f %%)\n", name, WAVES, per_row,
           100.0 * 258.0 / per_row);
}

// Two dy per lane, the structure a kernel would use: 17 window rows against 16 current rows, every step
// statically unrolled (window buffers alternate, the three current-row buffers rotate), no copies.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k2b(uint32_t *__restrict__ out, const uint32_t *__restrict__ cur, int cstr,
                                                  uint32_t seed, int pitch)
{
    __shared__ uint32_t win[200 * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 200 * PITCH; i += WAVES * 64) win[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc32[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const uint64_t c0 = clock64();
    for (int blk = 0; blk < BLOCKS; blk++) {
        const uint32_t *wrow = win + (lane & 31) + ((lane >> 5) + (wave & 3) + (blk & 63)) * pitch;
        const uint32_t *crow = cur + (wave & 3) * 16 * cstr;
        uint32_t d[2][17], S[3][16];
        uint64_t a0 = 0, a1 = 0;
#define LD_W(w, b) { const uint32_t *p_ = wrow + (w) * pitch; _Pragma("unroll") for (int k2 = 0; k2 < 17; k2++) d[b][k2] = p_[k2]; }
#define LD_C(r, b) { _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) S[b][k2] = crow[(r) * cstr + k2]; }
        LD_W(0, 0) LD_C(0, 0)
#pragma unroll
        for (int w = 0; w <= 16; w++) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (w < 16) LD_W(w + 1, (w + 1) & 1)
            if (w < 15) LD_C(w + 1, (w + 1) % 3)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) {
                const uint64_t x_ = ((uint64_t)d[w & 1][k2 + 1] << 32) | d[w & 1][k2];
                if (w < 16) a0 = __builtin_amdgcn_qsad_pk_u16_u8(x_, S[w % 3][k2], a0);
                if (w > 0) a1 = __builtin_amdgcn_qsad_pk_u16_u8(x_, S[(w + 2) % 3][k2], a1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((w & 3) == 3 || w == 16) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc32[q] += (uint32_t)(a0 >> (16 * q)) & 0xffffu;
                    acc32[4 + q] += (uint32_t)(a1 >> (16 * q)) & 0xffffu;
                }
                a0 = a1 = 0;
            }
        }
#undef LD_W
#undef LD_C
    }
    const uint64_t c1 = clock64();
    uint32_t a = 0;
    for (int q = 0; q < 8; q++) a ^= acc32[q];
    if (a == 0x12345678u) out[tid] = a;
    if (blockIdx.x == 7 && lane == 0) ((uint64_t *)out)[512 + wave] = c1 - c0;
}

template <int WAVES> static void run2b(const char *name, uint32_t *d_out, uint32_t *d_cur, int pitch)
{
    uint64_t h[16];
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k2b<WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d_out, d_cur, 480, 1u + rep, pitch);
    hipDeviceSynchronize();
    hipMemcpy(h, (uint64_t *)d_out + 512, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t mx = 0;
    for (int i = 0; i < WAVES; i++) mx = h[i] > mx ? h[i] : mx;
    // 512 qsads per block = 32 row-equivalents of 16 qsads
    const double per_row = (double)mx / (BLOCKS * 32) / (WAVES / 4);
    printf("%-44s %2d waves/CU: %6.1f SIMD cycles per 16 qsads (258 at the issue limit: %4.1f %%)\n", name, WAVES, per_row,
           100.0 * 258.0 / per_row);
}

// Eight dx per lane: an 18-dword window row (nine 8-byte-aligned ds_read_b64) pairs with ONE current row
// for two groups of four candidates; the odd dword pairs (v_pk_mov) are shared by both groups.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k8(uint32_t *__restrict__ out, const uint32_t *__restrict__ cur, int cstr,
                                                 uint32_t seed, int pitch)
{
    __shared__ __attribute__((aligned(16))) uint32_t win[200 * 66];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 200 * 66; i += WAVES * 64) win[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc32[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t dA[18], dB[18], SA[16], SB[16];
    const uint64_t c0 = clock64();
    for (int blk = 0; blk < BLOCKS; blk++) {
        const uint32_t *wrow = win + 2 * (lane & 15) + ((lane >> 4) + (wave & 3) + (blk & 63)) * pitch;
        const uint32_t *crow = cur + (wave & 3) * 16 * cstr;
        uint64_t a0 = 0, a1 = 0;
#define LD8(r, d, S)                                                                          \
    {                                                                                         \
        const uint64_t *p_ = reinterpret_cast<const uint64_t *>(wrow + (r) * pitch);          \
        _Pragma("unroll") for (int k2 = 0; k2 < 9; k2++) {                                    \
            const uint64_t v_ = p_[k2];                                                       \
            d[2 * k2] = (uint32_t)v_;                                                         \
            d[2 * k2 + 1] = (uint32_t)(v_ >> 32);                                             \
        }                                                                                     \
        _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) S[k2] = crow[(r) * cstr + k2];      \
    }
#define ROW8(d, S)                                                                                           \
    _Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) {                                                     \
        a0 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k2 + 1] << 32) | d[k2], S[k2], a0);                \
        a1 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d[k2 + 2] << 32) | d[k2 + 1], S[k2], a1);            \
    }
        LD8(0, dA, SA)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            LD8(r + 1, dB, SB)
            __builtin_amdgcn_sched_barrier(0);
            ROW8(dA, SA)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (r + 2 < 16) LD8(r + 2, dA, SA)
            __builtin_amdgcn_sched_barrier(0);
            ROW8(dB, SB)
            __builtin_amdgcn_sched_barrier(0);
            if ((r & 3) == 2) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc32[q] += (uint32_t)(a0 >> (16 * q)) & 0xffffu;
                    acc32[4 + q] += (uint32_t)(a1 >> (16 * q)) & 0xffffu;
                }
                a0 = a1 = 0;
            }
        }
#undef LD8
#undef ROW8
    }
    const uint64_t c1 = clock64();
    uint32_t a = 0;
    for (int q = 0; q < 8; q++) a ^= acc32[q];
    if (a == 0x12345678u) out[tid] = a;
    if (blockIdx.x == 7 && lane == 0) ((uint64_t *)out)[512 + wave] = c1 - c0;
}

template <int WAVES> static void run8(const char *name, uint32_t *d_out, uint32_t *d_cur, int pitch)
{
    uint64_t h[16];
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k8<WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d_out, d_cur, 480, 1u + rep, pitch);
    hipDeviceSynchronize();
    hipMemcpy(h, (uint64_t *)d_out + 512, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t mx = 0;
    for (int i = 0; i < WAVES; i++) mx = h[i] > mx ? h[i] : mx;
    // 16 window rows x 32 qsads per block = 32 row-equivalents of 16 qsads
    const double per_row = (double)mx / (BLOCKS * 32) / (WAVES / 4);
    printf("%-44s %2d waves/CU: %6.1f SIMD cycles per 16 qsads (258 at the issue limit: %4.1f %%), pitch %d\n", name, WAVES,
           per_row, 100.0 * 258.0 / per_row, pitch);
}

int main()
{
    uint32_t *d_out, *d_cur;
    hipMalloc(&d_out, 1 << 16);
    hipMalloc(&d_cur, 480 * 4 * 300);
    hipMemset(d_cur, 1, 480 * 4 * 300);
    run<0, 0, 16>("qsad only", d_out, d_cur, PITCH);
    run<1, 0, 16>("qsad + LDS row (+ v_pk_mov)", d_out, d_cur, PITCH);
    run<0, 1, 16>("qsad + s_load", d_out, d_cur, PITCH);
    run<1, 1, 16>("qsad + LDS row + s_load (the kernel)", d_out, d_cur, PITCH);
    run<1, 1, 8>("qsad + LDS row + s_load", d_out, d_cur, PITCH);
    run<1, 1, 8, 1>("same, current block packed (immediate offsets)", d_out, d_cur, PITCH);
    run<1, 1, 16, 1>("same, current block packed (immediate offsets)", d_out, d_cur, PITCH);
    run<1, 1, 4>("qsad + LDS row + s_load", d_out, d_cur, PITCH);
    run2<8>("two dy per lane (32 qsads per window row)", d_out, d_cur, PITCH);
    run2<16>("two dy per lane (32 qsads per window row)", d_out, d_cur, PITCH);
    run2b<8>("two dy per lane, 17 static steps", d_out, d_cur, PITCH);
    run2b<16>("two dy per lane, 17 static steps", d_out, d_cur, PITCH);
    run8<8>("eight dx per lane (32 qsads per window row)", d_out, d_cur, 66);
    run8<16>("eight dx per lane (32 qsads per window row)", d_out, d_cur, 66);
    run8<16>("eight dx per lane (32 qsads per window row)", d_out, d_cur, 50);
    return 0;
}
