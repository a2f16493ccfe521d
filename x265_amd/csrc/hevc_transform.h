// hevc_transform.h -- compile-time HEVC core-transform ROM and exact integer
// partial butterflies for gfx950 device code.
//
// The ROM (reference: g_aiT4/8/16/32, src/table.cpp:72-145) is never stored: it
// is generated at compile time from the 32 distinct magnitudes of the 32-point
// matrix, and every coefficient ends up as an instruction literal.  The
// butterflies are the recursive even/odd factorisation the reference unrolls by
// hand (src/pixel.cpp:114-312 forward, 321-569 inverse); all sums are Int32, so
// they equal the reference's sums bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

namespace x265amd {

// |T32[a][0]| for a = 0..32 (cosine magnitudes, 64 * sqrt(2) * cos(a*pi/64) rounded by the standard)
template <int A> struct CosMag;
#define X265AMD_COS(a, v) \
    template <> struct CosMag<a> { static constexpr int value = v; };
X265AMD_COS(0, 64) X265AMD_COS(1, 90) X265AMD_COS(2, 90) X265AMD_COS(3, 90) X265AMD_COS(4, 89)
X265AMD_COS(5, 88) X265AMD_COS(6, 87) X265AMD_COS(7, 85) X265AMD_COS(8, 83) X265AMD_COS(9, 82)
X265AMD_COS(10, 80) X265AMD_COS(11, 78) X265AMD_COS(12, 75) X265AMD_COS(13, 73) X265AMD_COS(14, 70)
X265AMD_COS(15, 67) X265AMD_COS(16, 64) X265AMD_COS(17, 61) X265AMD_COS(18, 57) X265AMD_COS(19, 54)
X265AMD_COS(20, 50) X265AMD_COS(21, 46) X265AMD_COS(22, 43) X265AMD_COS(23, 38) X265AMD_COS(24, 36)
X265AMD_COS(25, 31) X265AMD_COS(26, 25) X265AMD_COS(27, 22) X265AMD_COS(28, 18) X265AMD_COS(29, 13)
X265AMD_COS(30, 9) X265AMD_COS(31, 4) X265AMD_COS(32, 0)
#undef X265AMD_COS

// T_N[K][J] = T32[K * 32 / N][J]; T32[k][j] = sgn * mag[fold(k * (2j + 1) mod 128)]
template <int N, int K, int J> struct Coef {
    static constexpr int a0 = ((K * (32 / N)) * (2 * J + 1)) & 127;
    static constexpr int a1 = a0 > 64 ? 128 - a0 : a0;
    static constexpr int value = a1 <= 32 ? CosMag<(a1 <= 32 ? a1 : 0)>::value
                                          : -CosMag<(a1 > 32 ? 64 - a1 : 0)>::value;
};

// 24-bit multiply is full rate on the VALU (v_mul_i32_i24 / v_mad_i32_i24); every
// operand here is at most 18 bits (Int16 data after <= 2 butterfly add levels,
// 8-bit coefficients), so it is exact.
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// sum_{j < M} T_N[R][j] * v[j]
template <int N, int R, int... J>
__device__ __forceinline__ int dot_row(const int *v, std::integer_sequence<int, J...>)
{
    return (mul24(Coef<N, R, J>::value, v[J]) + ...);
}
// sum_{k < M} T_N[2k+1][C] * v[2k+1]   (odd rows, column C)
template <int N, int C, int... K>
__device__ __forceinline__ int dot_col_odd(const int *v, std::integer_sequence<int, K...>)
{
    return (mul24(Coef<N, 2 * K + 1, C>::value, v[2 * K + 1]) + ...);
}

template <int N, int... K>
__device__ __forceinline__ void fwd_odd(const int *o, int *y, std::integer_sequence<int, K...>)
{
    ((y[2 * K + 1] = dot_row<N, 2 * K + 1>(o, std::make_integer_sequence<int, N / 2>{})), ...);
}
template <int N, int... C>
__device__ __forceinline__ void inv_odd(const int *y, int *o, std::integer_sequence<int, C...>)
{
    ((o[C] = dot_col_odd<N, C>(y, std::make_integer_sequence<int, N / 2>{})), ...);
}

// y[k] = sum_j T_N[k][j] x[j]   (xDCT4/8/16/32 without the rounding shift)
template <int N> __device__ __forceinline__ void fwd_butterfly(const int *x, int *y)
{
    if constexpr (N == 2) {
        y[0] = (x[0] + x[1]) * 64;
        y[1] = (x[0] - x[1]) * 64;
    } else {
        int e[N / 2], o[N / 2], ye[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            e[k] = x[k] + x[N - 1 - k];
            o[k] = x[k] - x[N - 1 - k];
        }
        fwd_butterfly<N / 2>(e, ye);
#pragma unroll
        for (int k = 0; k < N / 2; k++) y[2 * k] = ye[k];
        fwd_odd<N>(o, y, std::make_integer_sequence<int, N / 2>{});
    }
}

// x[j] = sum_k T_N[k][j] y[k]   (xInvDCT4/8/16/32 without the rounding shift)
template <int N> __device__ __forceinline__ void inv_butterfly(const int *y, int *x)
{
    if constexpr (N == 2) {
        x[0] = (y[0] + y[1]) * 64;
        x[1] = (y[0] - y[1]) * 64;
    } else {
        int ye[N / 2], e[N / 2], o[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; k++) ye[k] = y[2 * k];
        inv_butterfly<N / 2>(ye, e);
        inv_odd<N>(y, o, std::make_integer_sequence<int, N / 2>{});
#pragma unroll
        for (int j = 0; j < N / 2; j++) {
            x[j] = e[j] + o[j];
            x[N - 1 - j] = e[j] - o[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Packed variants: two Int16 operands per register, v_dot2c_i32_i16 (two multiply-adds per
// instruction, the coefficient pair as its 32-bit literal), v_pk_add/sub_i16 with op_sel doing the
// butterfly's mirror.  Exact as long as every dot2 operand fits Int16 -- stated at each use.
// ---------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s swap2(v2s a) { return v2s{a.y, a.x}; }
__device__ __forceinline__ int dot2(v2s a, v2s b, int c) { return __builtin_amdgcn_sdot2(a, b, c, false); }
__device__ __forceinline__ v2s pack2(int lo, int hi) { return v2s{(short)lo, (short)hi}; }  // truncating

// Every packed transform takes SC: all coefficients are multiplied by 2^SC, so the result is
// 2^SC * (rnd + sum) modulo 2^32.  The callers only ever keep 16 bits [sh+15 : sh] of the sum; with
// SC chosen so that sh + SC is 8 or 16 those bits sit on a byte boundary and one v_perm_b32 packs two
// results -- no shift instructions.  Wrapping modulo 2^32 cannot disturb bits below 32.

// sum_{m < M} dot2(v[m], (T_N[R][2m], T_N[R][2m+1]))
template <int N, int SC, int R, int... M>
__device__ __forceinline__ int dot_row_pk(const v2s *v, int acc, std::integer_sequence<int, M...>)
{
    ((acc = dot2(v[M], v2s{(short)(Coef<N, R, 2 * M>::value * (1 << SC)), (short)(Coef<N, R, 2 * M + 1>::value * (1 << SC))}, acc)), ...);
    return acc;
}
template <int N, int SC, int... K>
__device__ __forceinline__ void fwd_odd_pk(const v2s *o, int *y, int rnd, std::integer_sequence<int, K...>)
{
    ((y[2 * K + 1] = dot_row_pk<N, SC, 2 * K + 1>(o, rnd, std::make_integer_sequence<int, N / 4>{})), ...);
}
// y[k] = rnd + 2^SC sum_j T_N[k][j] x[j];  P[m] = (x[2m], x[2m+1]), m < N/2.  Operands grow one bit per
// level: |x| <= 2^(15 - log2 N) keeps every level inside Int16 (residuals: |x| <= 255, N <= 32: fine).
template <int N, int SC> __device__ __forceinline__ void fwd_butterfly_pk(const v2s *P, int *y, int rnd)
{
    static_assert((90 << SC) < 32768, "scaled coefficient must fit Int16");
    if constexpr (N == 2) {
        y[0] = dot2(P[0], v2s{(short)(64 << SC), (short)(64 << SC)}, rnd);
        y[1] = dot2(P[0], v2s{(short)(64 << SC), (short)(-(64 << SC))}, rnd);
    } else {
        v2s E[N / 4], O[N / 4];
        int ye[N / 2];
#pragma unroll
        for (int m = 0; m < N / 4; m++) {
            const v2s sw = swap2(P[N / 2 - 1 - m]);  // (x[N-1-2m], x[N-2-2m])
            E[m] = P[m] + sw;
            O[m] = P[m] - sw;
        }
        fwd_butterfly_pk<N / 2, SC>(E, ye, rnd);
#pragma unroll
        for (int k = 0; k < N / 2; k++) y[2 * k] = ye[k];
        fwd_odd_pk<N, SC>(O, y, rnd, std::make_integer_sequence<int, N / 2>{});
    }
}

// sum_{m < M} dot2((y[4m+1], y[4m+3]), (T_N[4m+1][C], T_N[4m+3][C]))   (odd rows, column C)
template <int N, int SC, int C, int... M>
__device__ __forceinline__ int dot_col_odd_pk(const v2s *yo, std::integer_sequence<int, M...>)
{
    int acc = 0;
    ((acc = dot2(yo[M], v2s{(short)(Coef<N, 4 * M + 1, C>::value * (1 << SC)), (short)(Coef<N, 4 * M + 3, C>::value * (1 << SC))}, acc)), ...);
    return acc;
}
template <int N, int SC, int... C>
__device__ __forceinline__ void inv_odd_pk(const v2s *yo, int *o, std::integer_sequence<int, C...>)
{
    ((o[C] = dot_col_odd_pk<N, SC, C>(yo, std::make_integer_sequence<int, N / 4>{})), ...);
}
// x[j] = rnd + 2^SC sum_k T_N[k][j] y[k]; y[] are Int16 values held in ints (every dot2 operand is a
// raw input, so any Int16 input is exact).  S = index stride of this level's inputs in y[].
template <int N, int SC, int S = 1> __device__ __forceinline__ void inv_butterfly_pk(const int *y, int *x, int rnd)
{
    static_assert((90 << SC) < 32768, "scaled coefficient must fit Int16");
    if constexpr (N == 2) {
        const v2s p = pack2(y[0], y[S]);
        x[0] = dot2(p, v2s{(short)(64 << SC), (short)(64 << SC)}, rnd);
        x[1] = dot2(p, v2s{(short)(64 << SC), (short)(-(64 << SC))}, rnd);
    } else {
        int e[N / 2], o[N / 2];
        v2s yo[N / 4];
#pragma unroll
        for (int m = 0; m < N / 4; m++) yo[m] = pack2(y[(4 * m + 1) * S], y[(4 * m + 3) * S]);
        inv_butterfly_pk<N / 2, SC, 2 * S>(y, e, rnd);
        inv_odd_pk<N, SC>(yo, o, std::make_integer_sequence<int, N / 2>{});
#pragma unroll
        for (int j = 0; j < N / 2; j++) {
            x[j] = e[j] + o[j];
            x[N - 1 - j] = e[j] - o[j];
        }
    }
}

// Forward transform of FULL-RANGE Int16 inputs (the column pass: x[k] - x[N-1-k] needs 17 bits).
// Q[k] = (x[k], x[N-1-k]): the odd rows take dot2(Q[k], (T, -T)) -- raw operands, exact -- and the even
// half goes on in 32 bits through the scalar butterfly (scaled afterwards: one v_lshl_add with rnd).
template <int N, int SC, int R, int... K>
__device__ __forceinline__ int dot_row_mirror(const v2s *Q, int acc, std::integer_sequence<int, K...>)
{
    ((acc = dot2(Q[K], v2s{(short)(Coef<N, R, K>::value * (1 << SC)), (short)(-(Coef<N, R, K>::value * (1 << SC)))}, acc)), ...);
    return acc;
}
template <int N, int SC, int... K>
__device__ __forceinline__ void fwd_odd_mirror(const v2s *Q, int *y, int rnd, std::integer_sequence<int, K...>)
{
    ((y[2 * K + 1] = dot_row_mirror<N, SC, 2 * K + 1>(Q, rnd, std::make_integer_sequence<int, N / 2>{})), ...);
}
template <int N, int SC> __device__ __forceinline__ void fwd_butterfly_wide(const v2s *Q, int *y, int rnd)
{
    static_assert((90 << SC) < 32768, "scaled coefficient must fit Int16");
    int e[N / 2], ye[N / 2];
#pragma unroll
    for (int k = 0; k < N / 2; k++) e[k] = (int)Q[k].x + (int)Q[k].y;
    fwd_butterfly<N / 2>(e, ye);
#pragma unroll
    for (int k = 0; k < N / 2; k++) y[2 * k] = (int)(((unsigned)ye[k] << SC) + (unsigned)rnd);
    fwd_odd_mirror<N, SC>(Q, y, rnd, std::make_integer_sequence<int, N / 2>{});
}

// 4-point DST-VII (xDST4 src/pixel.cpp:133-151; xInvDST4 :346-369), matrix
// {29,55,74,84; 74,74,0,-74; 84,-29,-74,55; 55,-84,74,-29}
__device__ __forceinline__ void fwd_dst4(const int *x, int *y)
{
    y[0] = mul24(29, x[0]) + mul24(55, x[1]) + mul24(74, x[2]) + mul24(84, x[3]);
    y[1] = mul24(74, x[0] + x[1] - x[3]);
    y[2] = mul24(84, x[0]) - mul24(29, x[1]) - mul24(74, x[2]) + mul24(55, x[3]);
    y[3] = mul24(55, x[0]) - mul24(84, x[1]) + mul24(74, x[2]) - mul24(29, x[3]);
}
__device__ __forceinline__ void inv_dst4(const int *y, int *x)
{
    x[0] = mul24(29, y[0]) + mul24(74, y[1]) + mul24(84, y[2]) + mul24(55, y[3]);
    x[1] = mul24(55, y[0]) + mul24(74, y[1]) - mul24(29, y[2]) - mul24(84, y[3]);
    x[2] = mul24(74, y[0] - y[2] + y[3]);
    x[3] = mul24(84, y[0]) - mul24(74, y[1]) + mul24(55, y[2]) - mul24(29, y[3]);
}

// DST selection by intra mode (src/pixel.cpp:678-680).  `hor` = the pass over x.
__device__ __forceinline__ bool use_dst(int n, unsigned mode, bool hor)
{
    if (n != 4 || mode == 255u) return false;
    if (mode == 0) return true;
    return hor ? (mode >= 2 && mode <= 25) : (mode >= 11 && mode <= 34);
}

}  // namespace x265amd
