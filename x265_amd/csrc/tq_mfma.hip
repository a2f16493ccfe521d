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
#include "common.h"
#include "hevc_transform.h"
#include "tq_common.h"

namespace x265amd {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ---- compile-time ROM ------------------------------------------------------------------------------
constexpr int kCosMagTab[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
constexpr int tcoef(int n, int k, int j)  // T_n[k][j], as Coef<n, k, j>::value
{
    const int a0 = ((k * (32 / n)) * (2 * j + 1)) & 127;
    const int a1 = a0 > 64 ? 128 - a0 : a0;
    return a1 <= 32 ? kCosMagTab[a1] : -kCosMagTab[64 - a1];
}
static_assert(tcoef(32, 1, 0) == Coef<32, 1, 0>::value && tcoef(32, 31, 5) == Coef<32, 31, 5>::value &&
                  tcoef(16, 3, 7) == Coef<16, 3, 7>::value && tcoef(8, 5, 2) == Coef<8, 5, 2>::value &&
                  tcoef(4, 3, 1) == Coef<4, 3, 1>::value && tcoef(4, 2, 3) == Coef<4, 2, 3>::value,
              "ROM generators disagree");
// the 32x32 block-diagonal matrix of (32/n)^2 transforms of size n
constexpr int tblock(int n, int k, int j) { return k / n == j / n ? tcoef(n, k % n, j % n) : 0; }
// row of the product held by accumulator register r of a lane in half h (v_mfma_*_32x32: C/D layout)
__host__ __device__ constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct FragSet {             // per lane (n = lane & 31, h = lane >> 5): 16 bytes of the constant operand
    int8_t f1[64][16];       // T[n][16 h + j]      forward pass 1 (B operand; data rows are read 16 bytes per lane)
    int8_t f2[64][16];       // T[n][rho(j, h)]     forward pass 2 (A operand; data = previous accumulators)
    int8_t f3[64][16];       // T[rho(j, h)][n]     both inverse passes
    int32_t kappa[32];       // 128 * sum_k T[k][n]: what the "+ 128" of the low data bytes adds to inverse output n
};
constexpr FragSet make_frags(int n)
{
    FragSet f = {};
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) {
            const int c = l & 31, h = l >> 5;
            f.f1[l][j] = (int8_t)tblock(n, c, 16 * h + j);
            f.f2[l][j] = (int8_t)tblock(n, c, rho(j, h));
            f.f3[l][j] = (int8_t)tblock(n, rho(j, h), c);
        }
    for (int c = 0; c < 32; c++) {
        int s = 0;
        for (int k = 0; k < 32; k++) s += tblock(n, k, c);
        f.kappa[c] = 128 * s;
    }
    return f;
}
__device__ const FragSet kFrags[4] = {make_frags(4), make_frags(8), make_frags(16), make_frags(32)};

// ---- small helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ v4i as_v4i(const uint32_t (&w)[4]) { return v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}; }
__device__ __forceinline__ v16i splat16(int v) { return v16i{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v}; }
__device__ __forceinline__ v16i mfma(v4i a, v4i b, v16i c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

// four values whose low 16 bits are wanted -> (high bytes, low bytes - 128) as the two operand dwords
__device__ __forceinline__ void split4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &hi, uint32_t &lo)
{
    const uint32_t ab = perm(b, a, 0x05010400u), cd = perm(d, c, 0x05010400u);  // a0 b0 a1 b1
    lo = perm(cd, ab, 0x05040100u) ^ 0x80808080u;
    hi = perm(cd, ab, 0x07060302u);
}
// two Int16 pairs {b, a}, {d, c} -> the same
__device__ __forceinline__ void split2x2(uint32_t ba, uint32_t dc, uint32_t &hi, uint32_t &lo)
{
    lo = perm(dc, ba, 0x06040200u) ^ 0x80808080u;
    hi = perm(dc, ba, 0x07050301u);
}
__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t pair)  // {sat8(hi), sat8(lo)} in bits [15:0]
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(pair));
    return r;
}
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)  // |a - b| + acc: v_sad_u32
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
__device__ __forceinline__ void swap32(uint32_t &x, uint32_t &y)  // x's upper half-wave <-> y's lower half-wave
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// ---- the kernel ------------------------------------------------------------------------------------
// One wave = one 32x32 tile at a time.  Grid: x = groups of four tile columns (a workgroup's four waves take
// four tiles that are neighbours in x: 128 contiguous bytes per picture row), y = runs of kRowsPerWave tile
// rows (the constant operands are set up once per wave and serve the run), z = frame.  Every address is a
// wave-uniform base plus a 32-bit lane offset.
constexpr int kRowsPerWave = 4;

template <int N, int MODE>
__global__ __launch_bounds__(256, 2) void tq_mfma_kernel(const TqParams p)
{
    constexpr int L2 = (N == 4) ? 2 : (N == 8) ? 3 : (N == 16) ? 4 : 5;
    constexpr int FI = L2 - 2;
    constexpr int sh1 = L2 - 1, sh2 = L2 + 6;
    const x265amd_tq_desc &d = p.d;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int tx = blockIdx.x * 4 + wave;
    if (tx >= p.tiles_x) return;  // wave-uniform; no barriers in this kernel
    const int f = blockIdx.z;
    const int x0 = tx * 32;
    const int wvalid = min(32, d.width - x0);  // a multiple of N

    // constant operands of this lane
    const FragSet &F = kFrags[FI];
    const v4i f1 = *reinterpret_cast<const v4i *>(F.f1[lane]);
    const v4i f2 = *reinterpret_cast<const v4i *>(F.f2[lane]);
    const v4i f3 = *reinterpret_cast<const v4i *>(F.f3[lane]);
    v16i cin_i1, cin_i2;  // accumulator starts of the inverse passes: rounding + the low-byte bias
    if constexpr (MODE & TQ_INV) {
        const int k1 = 64 + F.kappa[c];  // SHIFT_INV_1ST = 7; output index = the lane
#pragma unroll
        for (int r = 0; r < 16; r++) {
            cin_i1[r] = k1;
            cin_i2[r] = 2048 + F.kappa[rho(r, h)];  // SHIFT_INV_2ND = 12; output index = the register's row
        }
    }
    // xQuant (src/pixel.cpp:595-616) in signed form: level = (c Q + rnd + (c < 0 ? 2^qbits - 1 - 2 rnd : 0)) >> qbits
    // equals sign(c) ((|c| Q + rnd) >> qbits); 2^30 on top keeps the sum positive, so the shift is logical and
    // level + 2^(30 - qbits) comes out: |level| is then one v_sad_u32 away.
    const int q_pos = (int)p.q_rnd + (1 << 30);
    const int q_neg = q_pos + (int)((1u << p.q_bits) - 1u - 2u * p.q_rnd);
    const uint32_t q_bias = 1u << (30 - p.q_bits);
    // the "+ 1" of pass 1 and the "+ 128" of pass 2's low bytes, both on DC rows only (header comment)
    const int dc_corr = (N == 4 || h == 0) ? ((1 + ((c & (N - 1)) == 0 ? 1 : 0)) << (13 + L2)) : 0;

    // ---- memory: straight-line, branch-free accesses ----
    // Every access is a buffer instruction.  Planes whose extent the descriptor gives (source, reconstruction,
    // levels, coefficients) are resources of exactly that extent: a lane that has nothing to load or store is
    // given an offset past the end, which the hardware answers with zeros / drops -- no divergent control flow
    // around the accesses, and no access of this kernel to those planes can leave them.  The prediction may be
    // a window into a larger plane reached through vectors (offsets below d_pred included), so its resource
    // starts 2^30 bytes BELOW the window and has no end; lanes with nothing to read get the window's first bytes.
    constexpr uint32_t kOob = 0x7ffffff0u;  // beyond any plane (launch_tq caps a plane at 2^30 bytes)
    constexpr uint32_t kPredBias = 1u << 30;
    const uint8_t *pred_pl = d.d_pred + (int64_t)f * d.pred_frame_stride;
    const __amdgpu_buffer_rsrc_t pred_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(pred_pl) - kPredBias), 0, 0xffffffffu, 0x00020000);
    int16_t *lvl_pl = d.d_level + (int64_t)f * d.level_frame_stride;
    const uint32_t lvl_bytes = (uint32_t)(((int64_t)(d.height - 1) * d.level_stride + d.width) * 2);
    const __amdgpu_buffer_rsrc_t lvl_rs = __builtin_amdgcn_make_buffer_rsrc(lvl_pl, 0, lvl_bytes, 0x00020000);
    const bool want_coef = (MODE & TQ_FWD) && d.d_coef;
    const __amdgpu_buffer_rsrc_t coef_rs = __builtin_amdgcn_make_buffer_rsrc(
        want_coef ? d.d_coef + (int64_t)f * d.level_frame_stride : lvl_pl, 0, want_coef ? lvl_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        (MODE & TQ_FWD) ? const_cast<uint8_t *>(d.d_src) + (int64_t)f * d.src_frame_stride : const_cast<uint8_t *>(pred_pl), 0,
        (MODE & TQ_FWD) ? (uint32_t)((int64_t)(d.height - 1) * d.src_stride + d.width) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rec_rs = __builtin_amdgcn_make_buffer_rsrc(
        (MODE & TQ_INV) ? d.d_rec + (int64_t)f * d.rec_frame_stride : const_cast<uint8_t *>(pred_pl), 0,
        (MODE & TQ_INV) ? (uint32_t)((int64_t)(d.height - 1) * d.rec_stride + d.width) : 0u, 0x00020000);
    const bool want_sum = (MODE & TQ_FWD) && d.d_abs_sum;
    const __amdgpu_buffer_rsrc_t sum_rs = __builtin_amdgcn_make_buffer_rsrc(
        want_sum ? d.d_abs_sum + (int64_t)f * d.abs_sum_frame_stride : reinterpret_cast<uint32_t *>(lvl_pl), 0,
        want_sum ? (uint32_t)((d.height >> L2) * (d.width >> L2)) * 4u : 0u, 0x00020000);
    const int row_bytes = d.level_stride * 2;
    const x265amd_mv *mv_pl = d.d_mv ? d.d_mv + (int64_t)f * d.mv_frame_stride : nullptr;
    const int tus_x = d.width >> L2;
    const int xh = x0 + 16 * h;                    // the lane's 16 bytes of a picture row start here
    const bool half_ok = 16 * h < wvalid;
    const bool col_ok = c < wvalid;                // lane = column c of the tile in the coefficient layout
    const bool wide = (wvalid & 15) == 0;          // wave-uniform: the row halves are whole 16-byte pieces

    const int ty_beg = blockIdx.y * kRowsPerWave;
    const int ty_end = min(ty_beg + kRowsPerWave, p.tiles_y);

    // the vectors of every tile of the run, fetched up front (only mvx | mvy: the first dword of x265amd_mv): the
    // prediction loads that depend on them can then be issued a whole tile ahead
    const __amdgpu_buffer_rsrc_t mv_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<x265amd_mv *>(mv_pl ? mv_pl : reinterpret_cast<const x265amd_mv *>(pred_pl)), 0,
        mv_pl ? (uint32_t)(p.mv_ctus_x * (d.height >> p.mv_ctu_log2)) * 8u : 0u, 0x00020000);
    uint32_t mvw[kRowsPerWave];
#pragma unroll
    for (int k = 0; k < kRowsPerWave; k++) mvw[k] = 0u;
    if (mv_pl) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < kRowsPerWave; k++) {
            const int yk = (ty_beg + k) * 32 + c;
            const bool ok = ty_beg + k < ty_end && yk < d.height && half_ok;
            const uint32_t idx = ok ? (uint32_t)((yk >> p.mv_ctu_log2) * p.mv_ctus_x + (xh >> p.mv_ctu_log2)) : 0u;
            mvw[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mv_rs, (int)(idx * 8u), 0, 0);  // entry 0 exists: the picture has a block
        }
    }

    // the lane's 16 bytes of its picture row (lane = row c of the tile, bytes 16 h .. 16 h + 15) of tile row `ty`
    auto load_rows = [&](int ty, bool active, uint32_t mv, uint32_t (&sw)[4], uint32_t (&pw)[4]) {
        const int y = ty * 32 + c;
        const bool row_ok = active && y < d.height && half_ok;
        // offset from the biased base, modulo 2^32: with vectors the prediction may lie above / left of d_pred, and
        // the wrap-around of unsigned arithmetic gives exactly the low 32 bits of the signed offset
        uint32_t poff = (uint32_t)y * (uint32_t)d.pred_stride + (uint32_t)xh;
        if (mv_pl) {  // wave-uniform; integer-pel motion compensation fused into the load (mv_ctu >= 16: one vector per half row)
            const int mvx = (int)(int16_t)(mv & 0xffffu), mvy = (int)mv >> 16;
            poff = (uint32_t)(y + (mvy >> d.mv_shift)) * (uint32_t)d.pred_stride + (uint32_t)(xh + (mvx >> d.mv_shift));
        }
        const uint32_t pofs = kPredBias + poff;
        const uint32_t soff = (uint32_t)y * (uint32_t)d.src_stride + (uint32_t)xh;
        if (wide) {  // wave-uniform
            const v4i a = __builtin_amdgcn_raw_buffer_load_b128(pred_rs, (int)(row_ok ? pofs : kPredBias), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) pw[i] = row_ok ? (uint32_t)a[i] : 0u;
            if constexpr (MODE & TQ_FWD) {
                const v4i b = __builtin_amdgcn_raw_buffer_load_b128(src_rs, (int)(row_ok ? soff : kOob), 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) sw[i] = (uint32_t)b[i];
            }
        } else {  // widths that are not a multiple of 16: dword by dword
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool ok = row_ok && xh + 4 * i < d.width;
                const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pred_rs, (int)(ok ? pofs + 4 * i : kPredBias), 0, 0);
                pw[i] = ok ? v : 0u;
                if constexpr (MODE & TQ_FWD)
                    sw[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(src_rs, (int)(ok ? soff + 4 * i : kOob), 0, 0);
                else
                    sw[i] = 0u;
            }
        }
        if constexpr (!(MODE & TQ_FWD)) {
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = 0u;
        }
    };

    uint32_t srcw[4], predw[4], src_nx[4], pred_nx[4];
    load_rows(ty_beg, true, mvw[0], srcw, predw);
#pragma unroll
    for (int k = 0; k < kRowsPerWave; k++) {
        const int ty = ty_beg + k;
        if (ty >= ty_end) break;  // wave-uniform
        // the next tile's rows are on their way while this one is transformed (past the run's end: all lanes idle)
        if (k + 1 < kRowsPerWave) load_rows(ty + 1, ty + 1 < ty_end, mvw[k + 1 < kRowsPerWave ? k + 1 : 0], src_nx, pred_nx);
        const int y0 = ty * 32;
        const int hvalid = min(32, d.height - y0);  // a multiple of N
        const int y = y0 + c;                       // lane = row c of the tile in the row layout
        const bool row_ok = c < hvalid && half_ok;

        int level[16];  // the lane's column of quantised levels: x = x0 + c, y = y0 + rho(r, h)
        // byte offset of the lane's (column c, row 4 h) in the level / coefficient planes; register r adds rho(r, 0)
        // rows through the scalar offset.  Lanes outside the plane's width are sent past the end; rows past its
        // height ARE past the end of the resource.
        const int lv_off = col_ok ? (int)(((uint32_t)(y0 + 4 * h) * (uint32_t)d.level_stride + (uint32_t)(x0 + c)) * 2u) : (int)kOob;
        if constexpr (MODE & TQ_FWD) {
            // ================= pass 1: residual x T^T (rows of the tile) =================
            uint32_t sa[4], pa[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                sa[i] = srcw[i] ^ 0x80808080u;   // src - 128
                pa[i] = predw[i] ^ 0x7f7f7f7fu;  // 127 - pred
            }
            v16i acc = mfma(as_v4i(sa), f1, splat16(1 << (sh1 - 1)));
            acc = mfma(as_v4i(pa), f1, acc);
            // (sum + rnd) >> sh1, truncating Int16 store (src/pixel.cpp:126-129) -- minus the 128 of the header
            // comment on DC columns; split into bytes for the next product
            uint32_t mh[4], ml[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                split4((uint32_t)acc[4 * g] >> sh1, (uint32_t)acc[4 * g + 1] >> sh1, (uint32_t)acc[4 * g + 2] >> sh1,
                       (uint32_t)acc[4 * g + 3] >> sh1, mh[g], ml[g]);
            // ================= pass 2: T x M1 (columns) =================
            constexpr int rnd2 = 1 << (sh2 - 1);
            v16i ah = mfma(f2, as_v4i(mh), splat16(N >= 8 ? rnd2 >> 8 : 0));  // 256 * (rnd2 >> 8) = rnd2 for N >= 8
            v16i al = mfma(f2, as_v4i(ml), splat16(0));
#pragma unroll
            for (int r = 0; r < 16; r++)
                if ((rho(r, 0) & (N - 1)) == 0) al[r] += dc_corr;  // rho(r, 1) = rho(r, 0) + 4: a DC row too iff N == 4
            uint32_t psum[4] = {0u, 0u, 0u, 0u};
            int cf[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int t = (int)(((uint32_t)ah[r] << 8) + (uint32_t)al[r]);
                if constexpr (N == 4) t += rnd2;
                cf[r] = __builtin_amdgcn_sbfe(t, sh2, 16);  // truncating Int16 store of the coefficient
                // xQuant
                const int q = mul24(cf[r], (int)p.q_scale) + (cf[r] < 0 ? q_neg : q_pos);
                const uint32_t lb = (uint32_t)q >> p.q_bits;   // level + q_bias
                psum[r >> 2] = sad_u32(lb, q_bias, psum[r >> 2]);
                level[r] = (int)(lb - q_bias);
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
                __builtin_amdgcn_raw_buffer_store_b16((short)level[r], lvl_rs, lv_off, rho(r, 0) * row_bytes, 0);
            if (want_coef) {  // wave-uniform; stage-by-stage checks only
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __builtin_amdgcn_raw_buffer_store_b16((short)cf[r], coef_rs, lv_off, rho(r, 0) * row_bytes, 0);
            }
            if (want_sum) {  // wave-uniform
                // xQuant's return value per TU: registers 4q .. 4q+3 are rows 8q + 4h .. + 3
                uint32_t s[4];
                constexpr int ns = N == 32 ? 1 : N == 16 ? 2 : 4;
                if constexpr (N == 32) {
                    s[0] = psum[0] + psum[1] + psum[2] + psum[3];
                } else if constexpr (N == 16) {
                    s[0] = psum[0] + psum[1];
                    s[1] = psum[2] + psum[3];
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) s[q4] = psum[q4];
                }
#pragma unroll
                for (int i = 0; i < ns; i++) {
#pragma unroll
                    for (int o = 1; o < N && o < 32; o <<= 1) s[i] += __shfl_xor(s[i], o);
                    if constexpr (N >= 8) s[i] += __shfl_xor(s[i], 32);
                }
                // one lane per TU column writes; TU rows below the picture are past the end of the resource
                const bool writer = (c & (N - 1)) == 0 && (N == 4 || h == 0) && col_ok;
                const uint32_t a_off = writer ? (uint32_t)((y0 >> L2) * tus_x + ((x0 + c) >> L2)) * 4u : kOob;
#pragma unroll
                for (int i = 0; i < ns; i++) {
                    const int tuy = N == 4 ? 2 * i + h : i;  // TU row inside the tile
                    __builtin_amdgcn_raw_buffer_store_b32((int)s[i], sum_rs, (int)(a_off + (uint32_t)(tuy * tus_x) * 4u), 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++)  // out of the plane: zeros
                level[r] = (int)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(lvl_rs, lv_off, rho(r, 0) * row_bytes, 0);
        }

        if constexpr (MODE & TQ_INV) {
            // ================= xDeQuant (src/pixel.cpp:649-650): clip16((level * scale + rnd) >> shift) =================
            uint32_t dh[4], dl[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = (mul24(level[4 * g + k], p.dq_scale) + p.dq_rnd) >> p.dq_shift;
                // v_cvt_pk_i16_i32 saturates: the clip and the pack in one
                const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[0], v[1]));
                const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[2], v[3]));
                split2x2(p0, p1, dh[g], dl[g]);
            }
            // ================= inverse pass 1 (columns): coefficients^T x T =================
            v16i ah = mfma(as_v4i(dh), f3, splat16(0));
            v16i al = mfma(as_v4i(dl), f3, cin_i1);
            // (sum + 64) >> 7, store-then-clip = wrap (src/pixel.cpp:334-342); lane = picture row now
            uint32_t uh[4], ul[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = (((uint32_t)ah[4 * g + k] << 8) + (uint32_t)al[4 * g + k]) >> 7;
                split4(t[0], t[1], t[2], t[3], uh[g], ul[g]);
            }
            // ================= inverse pass 2 (rows): T^T x intermediate =================
            ah = mfma(f3, as_v4i(uh), splat16(0));
            al = mfma(f3, as_v4i(ul), cin_i2);
            // the prediction in the result's layout: register group g = bytes x = 8 g + 4 h .. + 3 of the lane's row
            uint32_t pq[4] = {predw[0], predw[1], predw[2], predw[3]};  // dwords 4 h .. 4 h + 3 of the row
            swap32(pq[0], pq[1]);  // -> pq[0] = dword h, pq[1] = dword 4 + h
            swap32(pq[2], pq[3]);  // -> pq[2] = dword 2 + h, pq[3] = dword 6 + h
            const uint32_t pg[4] = {pq[0], pq[2], pq[1], pq[3]};
            uint32_t rq[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                // (sum + 2048) >> 12, wrap (truncating pack), + pred, Clip (src/pixel.cpp:757-761): the saturating
                // Int16 add cannot change a sum that is clipped to 0..255 anyway
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = (((uint32_t)ah[4 * g + k] << 8) + (uint32_t)al[4 * g + k]) << 4;
                const v2s r0 = __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, perm(t[1], t[0], 0x07060302u)),
                                                             __builtin_bit_cast(v2s, perm(0u, pg[g], 0x0c010c00u)));
                const v2s r1 = __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, perm(t[3], t[2], 0x07060302u)),
                                                             __builtin_bit_cast(v2s, perm(0u, pg[g], 0x0c030c02u)));
                rq[g] = perm(sat_pk_u8_i16(__builtin_bit_cast(uint32_t, r1)), sat_pk_u8_i16(__builtin_bit_cast(uint32_t, r0)),
                             0x05040100u);
            }
            // back to "16 consecutive bytes per lane": dwords 4 h .. 4 h + 3
            uint32_t ro[4] = {rq[0], rq[2], rq[1], rq[3]};  // dword h, 4 + h, 2 + h, 6 + h
            swap32(ro[0], ro[1]);  // -> dwords 4 h, 4 h + 1
            swap32(ro[2], ro[3]);  // -> dwords 4 h + 2, 4 h + 3
            const uint32_t off = (uint32_t)y * (uint32_t)d.rec_stride + (uint32_t)xh;
            if (wide) {  // wave-uniform
                __builtin_amdgcn_raw_buffer_store_b128(v4i{(int)ro[0], (int)ro[1], (int)ro[2], (int)ro[3]}, rec_rs,
                                                      (int)(row_ok ? off : kOob), 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    __builtin_amdgcn_raw_buffer_store_b32((int)ro[i], rec_rs, (int)(row_ok && xh + 4 * i < d.width ? off + 4 * i : kOob), 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            srcw[i] = src_nx[i];
            predw[i] = pred_nx[i];
        }
    }
}

template <int MODE> static void launch_n(int tu, dim3 grid, hipStream_t stream, const TqParams &p)
{
    switch (tu) {
    case 4: hipLaunchKernelGGL((tq_mfma_kernel<4, MODE>), grid, dim3(256), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((tq_mfma_kernel<8, MODE>), grid, dim3(256), 0, stream, p); break;
    case 16: hipLaunchKernelGGL((tq_mfma_kernel<16, MODE>), grid, dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((tq_mfma_kernel<32, MODE>), grid, dim3(256), 0, stream, p); break;
    }
}

int launch_tq_mfma(int mode, const TqParams &p, hipStream_t stream)
{
    if (p.d.nframes > 65535) return X265AMD_ERR_ARG;  // grid.z
    const dim3 grid((unsigned)((p.tiles_x + 3) / 4), (unsigned)((p.tiles_y + kRowsPerWave - 1) / kRowsPerWave), (unsigned)p.d.nframes);
    if (mode == TQ_FWD)
        launch_n<TQ_FWD>(p.d.tu, grid, stream, p);
    else if (mode == TQ_INV)
        launch_n<TQ_INV>(p.d.tu, grid, stream, p);
    else
        launch_n<TQ_BOTH>(p.d.tu, grid, stream, p);
    return hip_status(hipGetLastError());
}

}  // namespace x265amd
