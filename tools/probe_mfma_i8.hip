// probe_mfma_i8.hip -- which (row, k) does byte j of lane l hold in the A / B operands of
// v_mfma_i32_32x32x32_i8 on gfx950?  Exact integer data, asymmetric operands (the guide's rule).
//   hipcc --offload-arch=gfx950 -O2 -o probe_mfma_i8 probe_mfma_i8.hip && ./probe_mfma_i8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int8_t *A, const int8_t *B, int *C, int hyp)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v4i v; int8_t b[16]; } a, b;
    for (int j = 0; j < 16; j++) {
        const int k = hyp == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3);
        a.b[j] = A[r * 32 + k];   // A[row r][k]
        b.b[j] = B[k * 32 + r];   // B[k][col r]
    }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = c[i];
    }
}

int main()
{
    int8_t hA[1024], hB[1024];
    int want[1024], got[1024];
    srand(1);
    for (int i = 0; i < 1024; i++) {
        hA[i] = (int8_t)(rand() % 256 - 128);
        hB[i] = (int8_t)(rand() % 256 - 128);
    }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0;
            for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 32 + j];
            want[i * 32 + j] = s;
        }
    int8_t *dA, *dB;
    int *dC;
    hipMalloc(&dA, 1024);
    hipMalloc(&dB, 1024);
    hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; hyp++) {
        hipMemset(dC, 0, 4096);
        probe<<<1, 64>>>(dA, dB, dC, hyp);
        hipMemcpy(got, dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; i++) bad += got[i] != want[i];
        printf("hypothesis %d (%s): %d mismatches of 1024\n", hyp, hyp == 0 ? "k = 16h + j" : "k = 8h + (j&7) + 16(j>>3)", bad);
    }
    return 0;
}
