// tq_mfma.hip -- the transform / quantisation path on the matrix cores, bit-exact.
//
// Replaces xSubDct / xQuant / xDeQuant / xIDctAdd (src/pixel.cpp:591-776) for whole picture planes, like
// tq_kernels.hip, with the four 1-D transform passes as INTEGER matrix products:
// v_mfma_i32_32x32x32_i8 multiplies signed bytes and accumulates in Int32, and integer addition is
// associative modulo 2^32, so a matrix product returns exactly the reference's Int32 sums
// (sum_j T[k][j] * x[j], src/pixel.cpp:114-312 / 321-569) whatever order the hardware adds in.  (The
// north_star's "MFMA is not used because the transform must be bit-exact" holds for the floating-point
// matrix instructions; the i8 form is exact.)  The HEVC matrices are Int8 by definition
// (g_aiT4/8/16/32, src/table.cpp:72-145); the data are not, so each pass splits them into byte planes:
//   forward pass 1 : residual = src - pred = (src - 128) + (127 - pred) + 1: two products with signed
//                    bytes, K = 2 x 32; the "+ 1" adds 64 N to the DC column before the shift, = +128 after
//                    it, which pass 2 turns into +128 on the TU's DC coefficient -- added there;
//   other passes   : Int16 data x = 256 H + (L' + 128), H = x >> 8 (signed byte), L' = (x & 255) - 128:
//                    sum = 256 (T.H) + T.L' + 128 (T.1).  T.1 is 64 N on DC rows and 0 elsewhere for the forward
//                    matrix; for the inverse it is the column sums of T: constants, folded into the accumulator
//                    start values.
// A 32x32 pixel tile per wave; TUs smaller than 32 use block-diagonal matrices (same instruction, zeros
// off the diagonal).  Each product's result is laid out by the hardware as "column on the lane, rows in
// the 16 accumulator registers"; the next pass sums over exactly those rows, so all four passes chain in
// registers: no LDS, no transposes (the fragment of the constant matrix is permuted instead).
//   src/pred rows (lane = row)  --pass 1-->  M1 (lane = kx, regs = y)  --pass 2-->  coefficients
//   (lane = x, regs = y: quantised, stored)  --dequant, inverse 1-->  (lane = y, regs = kx)  --inverse 2-->
//   residual (lane = y, regs = x)  + pred, clip  -->  16-byte row stores.
// The element-wise work between the products (round, shift, split into bytes, quantise, dequantise,
// clip) is what is left on the VALU: ~23 instructions per pixel-register instead of the butterflies' ~67.
#include "tq_mfma_body.h"

namespace x265amd {

template <int N, int MODE, bool MV>
__global__ __launch_bounds__(256, 2) void tq_mfma_kernel(const TqParams p)
{
    tq_mfma_body<N, MODE, MV>(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)(threadIdx.x >> 6));
}

// One anti-diagonal step of x265amd_intra_encode_frame(s) in ONE launch: blockIdx.z = 0 the luma plane (TU = N), 1 the
// chroma planes (TU = N / 2); the grid's y extent is the larger of the two plane counts.
template <int N>
__global__ __launch_bounds__(256, 2) void tq_mfma_diag_yuv_kernel(const TqParams py, const TqParams pc, const int ny_luma, const int ny_chroma)
{
    if (blockIdx.z == 0) {
        if ((int)blockIdx.y < ny_luma) tq_mfma_body<N, TQ_BOTH, false>(py, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)(threadIdx.x >> 6));
    } else {
        if ((int)blockIdx.y < ny_chroma) tq_mfma_body<N / 2, TQ_BOTH, false>(pc, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)(threadIdx.x >> 6));
    }
}

template <int MODE, bool MV> static void launch_n(int tu, dim3 grid, hipStream_t stream, const TqParams &p)
{
    switch (tu) {
    case 4: hipLaunchKernelGGL((tq_mfma_kernel<4, MODE, MV>), grid, dim3(256), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((tq_mfma_kernel<8, MODE, MV>), grid, dim3(256), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((tq_mfma_kernel<16, MODE, MV>), grid, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((tq_mfma_kernel<32, MODE, MV>), grid, dim3(256), 0, stream, p); break;
    }
}
template <int MODE> static void launch_mv(dim3 grid, hipStream_t stream, const TqParams &p)
{
    if (p.d.d_mv)
        launch_n<MODE, true>(p.d.tu, grid, stream, p);
    else
        launch_n<MODE, false>(p.d.tu, grid, stream, p);
}

// Units per workgroup.  0 = an even split over the workgroups the chip holds at once (four per CU: four waves per
// SIMD); n > 0 = n units per workgroup, the dispatcher handing out workgroups as others finish.
extern "C" void x265amd_debug_tq_units_per_wg(int n)
{
    if (device_state()) debug_state().tq_units_per_wg = n < 0 ? 0 : n;
}

int launch_tq_mfma(int mode, const TqParams &p_in, hipStream_t stream)
{
    constexpr int kWgPerCu = 4;
    TqParams p = p_in;
    const int64_t units = (int64_t)((p.tiles_x + 3) / 4) * p.tiles_y * p.d.nframes;
    if (units > 0x7fffffff) return X265AMD_ERR_ARG;
    const int64_t resident = (int64_t)cu_count() * kWgPerCu;
    int64_t nwg = units < resident ? units : resident;
    const int units_per_wg = debug_state().tq_units_per_wg;
    if (units_per_wg > 0) nwg = (units + units_per_wg - 1) / units_per_wg;
    dim3 grid((unsigned)nwg);
    if (p.diag_count > 0) {
        // every CU of the diagonal must lie inside the plane: the kernel takes its unit from the CU's position
        const int cu = 1 << p.diag_cu_log2;
        const int hi = p.diag_lo + p.diag_count - 1;
        if (p.diag_cu_log2 < 3 || p.diag_cu_log2 > 5 || p.diag_lo < 0 || (hi + 1) * cu > p.d.height || p.diag_d - 2 * hi < 0 ||
            (p.diag_d - 2 * p.diag_lo + 1) * cu > p.d.width || p.d.nframes > 65535)
            return X265AMD_ERR_ARG;
        if (p.diag_fgroup < 0 || (p.diag_fgroup > 0 && (p.diag_fmul < p.diag_fgroup || (int64_t)p.d.nframes * p.diag_fgroup > 65535)))
            return X265AMD_ERR_ARG;
        const int ny = p.diag_fgroup ? p.d.nframes * p.diag_fgroup : p.d.nframes;
        nwg = (int64_t)p.diag_count * ny;
        grid = dim3((unsigned)p.diag_count, (unsigned)ny);
    }
    if (p.trace_waves < nwg * 4) p.d_trace = nullptr;
    if (mode == TQ_FWD)
        launch_mv<TQ_FWD>(grid, stream, p);
    else if (mode == TQ_INV)
        launch_mv<TQ_INV>(grid, stream, p);
    else
        launch_mv<TQ_BOTH>(grid, stream, p);
    return hip_status(hipGetLastError());
}

// validation of a diagonal launch's parameters; ny = planes in the grid's y
static int diag_check(const TqParams &p, int &ny)
{
    const int cu = 1 << p.diag_cu_log2;
    const int hi = p.diag_lo + p.diag_count - 1;
    if (p.diag_count <= 0 || p.diag_cu_log2 < 3 || p.diag_cu_log2 > 5 || p.diag_lo < 0 || (hi + 1) * cu > p.d.height || p.diag_d - 2 * hi < 0 ||
        (p.diag_d - 2 * p.diag_lo + 1) * cu > p.d.width || p.d.nframes > 65535)
        return X265AMD_ERR_ARG;
    if (p.diag_fgroup < 0 || (p.diag_fgroup > 0 && (p.diag_fmul < p.diag_fgroup || (int64_t)p.d.nframes * p.diag_fgroup > 65535)))
        return X265AMD_ERR_ARG;
    ny = p.diag_fgroup ? p.d.nframes * p.diag_fgroup : p.d.nframes;
    return X265AMD_OK;
}

int tq_mfma_diag_yuv_check(TqParams &py, TqParams &pc, int &nyl, int &nyc)
{
    int st = diag_check(py, nyl);
    if (st == X265AMD_OK) st = diag_check(pc, nyc);
    if (st != X265AMD_OK) return st;
    // the same CUs in both: a chroma CU is half the luma CU
    if (py.diag_count != pc.diag_count || py.diag_d != pc.diag_d || py.diag_lo != pc.diag_lo || py.diag_cu_log2 != pc.diag_cu_log2 + 1 ||
        py.d.tu != 2 * pc.d.tu || (py.d.tu != 32 && py.d.tu != 16) || py.d.d_mv || pc.d.d_mv)
        return X265AMD_ERR_ARG;
    py.d_trace = pc.d_trace = nullptr;
    return X265AMD_OK;
}

int launch_tq_mfma_diag_yuv(const TqParams &py_in, const TqParams &pc_in, hipStream_t stream)
{
    TqParams py = py_in, pc = pc_in;
    int nyl = 0, nyc = 0;
    const int st = tq_mfma_diag_yuv_check(py, pc, nyl, nyc);
    if (st != X265AMD_OK) return st;
    const dim3 grid((unsigned)py.diag_count, (unsigned)(nyl > nyc ? nyl : nyc), 2);
    if (py.d.tu == 32)
        hipLaunchKernelGGL(tq_mfma_diag_yuv_kernel<32>, grid, dim3(256), 0, stream, py, pc, nyl, nyc);
    else
        hipLaunchKernelGGL(tq_mfma_diag_yuv_kernel<16>, grid, dim3(256), 0, stream, py, pc, nyl, nyc);
    return hip_status(hipGetLastError());
}

}  // namespace x265amd
