// ubench_valu.hip -- issue-rate microbenchmark for the integer VALU instructions the
// ME and TQ kernels are built from (gfx950).  Not part of the product; its numbers are
// recorded in DESIGN.md.   Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

constexpr int ITERS = 32768;
constexpr int CHAINS = 8;

// each kernel: CHAINS independent dependency chains of one instruction, ITERS iterations
#define KERNEL(name, decl, body, fold)                                             \
    __global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed)      \
    {                                                                              \
        uint32_t t = threadIdx.x + blockIdx.x * 256u + seed;                       \
        decl;                                                                      \
        const uint64_t c0 = clock64(), w0 = wall_clock64();                        \
        for (int it = 0; it < ITERS; it++) {                                       \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { body; }           \
        }                                                                          \
        uint32_t r = 0;                                                            \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) { fold; }               \
        const uint64_t c1 = clock64(), w1 = wall_clock64();                        \
        if (r == 0x12345678u) out[t & 1023] = r;                                   \
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                 \
            ((uint64_t *)out)[600] = c1 - c0;                                      \
            ((uint64_t *)out)[601] = w1 - w0;                                      \
        }                                                                          \
    }

KERNEL(k_add_u32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c,
       asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(t)), r ^= a[c])
KERNEL(k_sad_u8, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = ~x,
       asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_sad_u8_sgpr, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = seed * 77u,
       asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "s"(y)), r ^= a[c])
KERNEL(k_msad_u8, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = ~x,
       asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_sad_u16, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = ~x,
       asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_qsad_pk, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint64_t x = t * 0x9E3779B97F4A7C15ull; uint32_t y = ~t,
       asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= (uint32_t)a[c] ^ (uint32_t)(a[c] >> 32))
KERNEL(k_qsad_pk_sgpr, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint64_t x = t * 0x9E3779B97F4A7C15ull; uint32_t y = seed * 77u,
       asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "s"(y)), r ^= (uint32_t)a[c] ^ (uint32_t)(a[c] >> 32))
KERNEL(k_mqsad_pk, uint64_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint64_t x = t * 0x9E3779B97F4A7C15ull; uint32_t y = ~t,
       asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= (uint32_t)a[c] ^ (uint32_t)(a[c] >> 32))
KERNEL(k_mqsad_u32, uint4_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = (uint4_t)(t + c); uint64_t x = t * 0x9E3779B97F4A7C15ull; uint32_t y = ~t,
       asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c].x ^ a[c].y ^ a[c].z ^ a[c].w)
KERNEL(k_alignbyte, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u,
       asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[c]) : "v"(x)), r ^= a[c])
KERNEL(k_perm, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t s = 0x06050403u,
       asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(x), "v"(s)), r ^= a[c])
KERNEL(k_mad_i24, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t & 0xffff; uint32_t y = 83,
       asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_mul_lo_u32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t | 1,
       asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(x)), r ^= a[c])
KERNEL(k_dot2_i16, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = 0x00530024u,
       asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_dot4_i8, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = 0x5a573219u,
       asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_pk_add_i16, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u,
       asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[c]) : "v"(x)), r ^= a[c])
KERNEL(k_pk_mad_i16, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = 0x00530024u,
       asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_mad_u32_u16, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u; uint32_t y = 1,
       asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y)), r ^= a[c])
KERNEL(k_add_sdwa, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u,
       asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[c]) : "v"(x)), r ^= a[c])
KERNEL(k_ashr, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c,
       asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[c])), r ^= a[c])
KERNEL(k_med3_i32, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t lo = 0xffff8000u; uint32_t hi = 0x7fffu,
       asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(lo), "v"(hi)), r ^= a[c])
KERNEL(k_min_u64ish, uint32_t a[CHAINS]; for (int c = 0; c < CHAINS; c++) a[c] = t + c; uint32_t x = t * 2654435761u,
       asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[c]) : "v"(x)), r ^= a[c])

struct Entry {
    const char *name;
    void (*fn)(uint32_t *, uint32_t);
    double work;  // "useful" units per lane-instruction (pixels for SAD, MACs for dot)
    const char *unit;
};

int main(int argc, char **argv)
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate, prop.sharedMemPerBlock);
    uint32_t *out;
    CHECK(hipMalloc(&out, 8192));
    Entry tab[] = {
        {"v_add_u32", k_add_u32, 1, "op"},          {"v_sad_u8", k_sad_u8, 4, "px"},
        {"v_sad_u8(sgpr)", k_sad_u8_sgpr, 4, "px"},  {"v_msad_u8", k_msad_u8, 4, "px"},
        {"v_sad_u16", k_sad_u16, 2, "px"},           {"v_qsad_pk_u16_u8", k_qsad_pk, 16, "px"},
        {"v_qsad_pk(sgpr)", k_qsad_pk_sgpr, 16, "px"}, {"v_mqsad_pk_u16_u8", k_mqsad_pk, 16, "px"},
        {"v_mqsad_u32_u8", k_mqsad_u32, 16, "px"},   {"v_alignbyte_b32", k_alignbyte, 1, "op"},
        {"v_perm_b32", k_perm, 1, "op"},             {"v_mad_i32_i24", k_mad_i24, 1, "mac"},
        {"v_mul_lo_u32", k_mul_lo_u32, 1, "mac"},    {"v_dot2_i32_i16", k_dot2_i16, 2, "mac"},
        {"v_dot4_i32_i8", k_dot4_i8, 4, "mac"},      {"v_pk_add_i16", k_pk_add_i16, 2, "op"},
        {"v_pk_mad_i16", k_pk_mad_i16, 2, "mac"},    {"v_mad_u32_u16", k_mad_u32_u16, 1, "mac"},
        {"v_add_u32_sdwa", k_add_sdwa, 1, "op"},     {"v_ashrrev_i32", k_ashr, 1, "op"},
        {"v_med3_i32", k_med3_i32, 1, "op"},         {"v_min_u32", k_min_u64ish, 1, "op"},
    };
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    const int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-20s %10s %14s %16s %12s %14s %10s\n", "instruction", "ms", "Tlane-op/s", "cyc@2.4GHz", "T<unit>/s", "shader-cyc/wi", "GHz");
    for (auto &e : tab) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);  // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u + r);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double lane_ops = (double)blocks * 256 * ITERS * CHAINS;
        const double rate = lane_ops / (ms * 1e-3);
        // cycles one SIMD spends per wave-instruction, assuming 2.4 GHz and all SIMDs busy
        const double wave_instr_per_simd = (double)waves_per_simd * ITERS * CHAINS;
        const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
        uint64_t tm[2];
        CHECK(hipMemcpy(tm, (uint64_t *)out + 600, 16, hipMemcpyDeviceToHost));
        // in-kernel: shader cycles one SIMD spends per wave-instruction, and the shader clock (s_memrealtime = 100 MHz)
        const double scyc = (double)tm[0] / wave_instr_per_simd;
        const double ghz = (double)tm[0] / ((double)tm[1] * 10.0);
        printf("%-20s %10.3f %14.3f %16.2f %9.2f %-3s %14.2f %10.3f\n", e.name, ms, rate * 1e-12, cyc, rate * e.work * 1e-12,
               e.unit, scyc, ghz);
    }
    CHECK(hipFree(out));
    return 0;
}
