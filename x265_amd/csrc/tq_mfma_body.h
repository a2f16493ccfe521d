// tq_mfma_body.h -- the device side of the matrix-core transform / quantisation path (see tq_mfma.hip for the
// method): constant operands, helpers and tq_mfma_body, the kernel's body as a device function.  Included by
// tq_mfma.hip (the plane kernels) and by intra_kernels.hip (the transform stage at the end of an anti-diagonal
// step's prediction kernel); every translation unit holds its own copy of the constant operands.
#pragma once
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
static __device__ const FragSet kFrags[4] = {make_frags(4), make_frags(8), make_frags(16), make_frags(32)};

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
// the same for the fields at bits [8, 24) of the four values (no shift needed to reach them)
__device__ __forceinline__ void split4_b8(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &hi, uint32_t &lo)
{
    const uint32_t ab = perm(b, a, 0x06020501u), cd = perm(d, c, 0x06020501u);  // a1 b1 a2 b2
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
__device__ __forceinline__ int bfi(int mask, int a, int b)  // mask ? a : b, bit by bit: v_bfi_b32
{
    int r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int mad24(int a, int b, int c)  // a * b + c on the low 24 bits of a and b: v_mad_i32_i24
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)  // |a - b| + acc: v_sad_u32
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
// row * stride for a byte offset: both below 2^23 in magnitude (launch_tq sends other planes to the butterfly
// kernel), so the full-rate 24-bit multiply is exact; the 32-bit one runs at a quarter of the rate
__device__ __forceinline__ int rowmul(int row, int stride) { return __mul24(row, stride); }
// bits [lo, lo + 16) of t, sign-extended: v_bfe_i32 (written out: left to itself the compiler shifts the word up
// front instead and pays an instruction more per coefficient)
template <int LO> __device__ __forceinline__ int field16(int t)
{
    int r;
    asm("v_bfe_i32 %0, %1, %2, 16" : "=v"(r) : "v"(t), "n"(LO));
    return r;
}
// x + (x of the lane that the DPP pattern CTRL names); rows outside ROW_MASK add nothing
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ uint32_t dpp_add(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ void swap32(uint32_t &x, uint32_t &y)  // x's upper half-wave <-> y's lower half-wave
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// ---- the kernel ------------------------------------------------------------------------------------
// One wave = one 32x32 tile at a time; a workgroup's four waves take four tiles that are neighbours in x (128
// contiguous bytes per picture row): a UNIT.  Units are numbered frame-major, then tile row, then along x; the
// grid is sized to what the chip holds at once and every workgroup walks a contiguous, equal share of the
// units (the work per tile is the same everywhere, so an even split is a balanced one and there is no partial
// last round of workgroups).  While a unit is transformed, the next unit's rows are in flight, and the vectors
// of the one after that.
struct TilePos {  // wave-uniform
    int f, ty, txg;
};

// the kernel's body; bx / by / gdx = the workgroup's place in the grid (blockIdx.x, blockIdx.y, gridDim.x of a plain launch);
// wave_in = which of the unit's four tiles this wave takes (threadIdx.x >> 6 in a 256-thread workgroup), or -1 in
// diagonal mode: the tile that holds the CU (a caller that brings ONE wave per plane)
template <int N, int MODE, bool MV>
__device__ __forceinline__ void tq_mfma_body(const TqParams &p, const int bx, const int by, const int gdx, const int wave_in)
{
    constexpr int L2 = (N == 4) ? 2 : (N == 8) ? 3 : (N == 16) ? 4 : 5;
    constexpr int FI = L2 - 2;
    constexpr int sh1 = L2 - 1, sh2 = L2 + 6;
    const x265amd_tq_desc &d = p.d;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    // diagonal mode (p.diag_count > 0): ONE unit per workgroup, the one that holds CU (diag_d - 2 cuy, cuy) of frame
    // by, cuy = diag_lo + bx (the launcher sizes the grid; units that two CUs share are done twice,
    // to the same values)
    const bool diag = p.diag_count > 0;
    // ... and of that unit only the tile that holds the CU: the other three waves of the workgroup load nothing and skip
    const int only_tx = diag ? __builtin_amdgcn_readfirstlane(((p.diag_d - 2 * (p.diag_lo + bx)) << p.diag_cu_log2) >> 5) : -1;
    const int wave = __builtin_amdgcn_readfirstlane(wave_in >= 0 ? wave_in : (only_tx & 3));

    // x265amd_debug_tq_trace: when and where this wave ran (wave-uniform branch, off in normal use)
    unsigned long long t_wall0 = 0, t_clk0 = 0;
    if (p.d_trace) {
        t_wall0 = wall_clock64();  // 100 MHz
        t_clk0 = clock64();        // shader clock
    }
    auto trace_end = [&]() {
        if (p.d_trace && lane == 0) {
            unsigned long long *rec = p.d_trace + ((size_t)bx * 4 + wave) * 4;
            const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
            const unsigned xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
            rec[0] = t_wall0;
            rec[1] = wall_clock64();
            rec[2] = clock64() - t_clk0;
            rec[3] = ((unsigned long long)xcc_id << 32) | hw_id;
        }
    };

    const int ntxg = (p.tiles_x + 3) >> 2;
    const int nunits = ntxg * p.tiles_y * d.nframes;
    const int per = (nunits + gdx - 1) / gdx;
    const int u_beg = diag ? 0 : bx * per, u_end = diag ? 1 : min(u_beg + per, nunits);
    if (u_beg >= u_end) return;  // block-uniform; no barriers in this kernel

    // constant operands of this lane
    const FragSet &F = kFrags[FI];
    const v4i f1 = *reinterpret_cast<const v4i *>(F.f1[lane]);
    const v4i f2 = *reinterpret_cast<const v4i *>(F.f2[lane]);
    const v4i f3 = *reinterpret_cast<const v4i *>(F.f3[lane]);
    v16i cin_i1, cin_i2;  // accumulator starts of the inverse passes: rounding + the low-byte bias
    if constexpr (MODE & TQ_INV) {
        const int k1 = 2 * (64 + F.kappa[c]);  // SHIFT_INV_1ST = 7; output index = the lane; twice: see inverse pass 1
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

    // ---- memory: every access is a buffer instruction ----
    // Planes whose extent the descriptor gives (source, reconstruction, levels, coefficients, sums, vectors) are
    // resources of exactly that extent: a lane that has nothing to load or store is given an offset past the
    // end, which the hardware answers with zeros / drops -- no divergent control flow around the accesses, and
    // no access of this kernel to those planes can leave them.  The prediction may be a window into a larger
    // plane reached through vectors (offsets below d_pred included), so its resource starts 2^30 bytes BELOW
    // the window and has no end; lanes with nothing to read get the window's first bytes.
    constexpr uint32_t kOob = 0x7ffffff0u;  // beyond any plane (launch_tq caps a plane at 2^30 bytes)
    constexpr uint32_t kPredBias = 1u << 30;
    constexpr int kRsrcFlags = 0x00020000;
    const uint32_t lvl_bytes = (uint32_t)(((int64_t)(d.height - 1) * d.level_stride + d.width) * 2);
    const bool want_coef = (MODE & TQ_FWD) && d.d_coef;
    const bool want_sum = (MODE & TQ_FWD) && d.d_abs_sum;
    constexpr bool have_mv = MV;  // the launcher picks the instantiation: no branch on it between a load and its use
    const int row_bytes = d.level_stride * 2;
    const int tus_x = d.width >> L2;

    auto pred_rsrc = [&](int f) {
        const uint8_t *pl = d.d_pred + (int64_t)f * d.pred_frame_stride;
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(pl) - kPredBias), 0, 0xffffffffu,
                                                 kRsrcFlags);
    };
    // per-lane geometry of the tile this wave takes in unit `pos`
    struct Geo {
        int x0, xh, wvalid;
        bool tile_ok, half_ok, col_ok;
    };
    auto geo = [&](const TilePos &pos) {
        Geo g;
        const int tx = pos.txg * 4 + wave;
        g.tile_ok = tx < p.tiles_x && (only_tx < 0 || tx == only_tx);  // wave-uniform
        g.x0 = tx * 32;
        g.wvalid = g.tile_ok ? min(32, d.width - g.x0) : 0;  // a multiple of N
        g.xh = g.x0 + 16 * h;                             // the lane's 16 bytes of a picture row start here
        g.half_ok = 16 * h < g.wvalid;
        g.col_ok = c < g.wvalid;                          // lane = column c of the tile in the coefficient layout
        return g;
    };
    auto advance = [&](TilePos &pos) {
        if (++pos.txg == ntxg) {
            pos.txg = 0;
            if (++pos.ty == p.tiles_y) {
                pos.ty = 0;
                pos.f++;
            }
        }
    };
    // mvx | mvy (the first dword of x265amd_mv) of the lane's half row; mv_ctu >= 16: one vector per half row
    auto load_mv = [&](const TilePos &pos, bool active) -> uint32_t {
        if constexpr (!have_mv) return 0u;
        const Geo g = geo(pos);
        const int f = active ? pos.f : 0;
        const __amdgpu_buffer_rsrc_t mv_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<x265amd_mv *>(d.d_mv) + (int64_t)f * d.mv_frame_stride, 0, (uint32_t)(p.mv_ctus_x * (d.height >> p.mv_ctu_log2)) * 8u,
            kRsrcFlags);
        const int y = pos.ty * 32 + c;
        const bool ok = active && y < d.height && g.half_ok;
        const uint32_t idx = ok ? (uint32_t)(rowmul(y >> p.mv_ctu_log2, p.mv_ctus_x) + (g.xh >> p.mv_ctu_log2)) : 0u;
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mv_rs, (int)(idx * 8u), 0, 0);  // entry 0 exists
    };
    // the lane's 16 bytes of its picture row (lane = row c of the tile, bytes 16 h .. 16 h + 15)
    // (with vectors: pw = the five aligned dwords the compensated row lies in, psh = its byte offset in the first;
    //  pred_align shifts the row out where it is USED -- here it would make the wave wait for the loads)
    auto load_rows = [&](const TilePos &pos, bool active, uint32_t mv, uint32_t (&sw)[4], uint32_t (&pw)[5], uint32_t &psh) {
        const Geo g = geo(pos);
        const int f = active ? pos.f : 0;
        const __amdgpu_buffer_rsrc_t pred_rs = pred_rsrc(f);
        const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
            (MODE & TQ_FWD) ? const_cast<uint8_t *>(d.d_src) + (int64_t)f * d.src_frame_stride : const_cast<uint8_t *>(d.d_pred), 0,
            (MODE & TQ_FWD) ? (uint32_t)((int64_t)(d.height - 1) * d.src_stride + d.width) : 0u, kRsrcFlags);
        const int y = pos.ty * 32 + c;
        const bool row_ok = active && y < d.height && g.half_ok;
        // offset from the biased base, modulo 2^32: with vectors the prediction may lie above / left of d_pred, and
        // the wrap-around of unsigned arithmetic gives exactly the low 32 bits of the signed offset
        uint32_t poff = (uint32_t)rowmul(y, d.pred_stride) + (uint32_t)g.xh;
        if constexpr (have_mv) {  // integer-pel motion compensation fused into the load
            const int mvx = (int)(int16_t)(mv & 0xffffu), mvy = (int)mv >> 16;
            poff = (uint32_t)rowmul(y + (mvy >> d.mv_shift), d.pred_stride) + (uint32_t)(g.xh + (mvx >> d.mv_shift));
        }
        const uint32_t pofs = kPredBias + poff;
        const uint32_t soff = (uint32_t)rowmul(y, d.src_stride) + (uint32_t)g.xh;
        // The plane's width is a multiple of 16 (the launcher sends other widths to the butterfly kernel), so a lane's
        // 16 bytes are inside the row or not at all.  Idle lanes read the window's first bytes / zeros: what they
        // hold never reaches a valid output (rows and columns past the picture belong to transform units that
        // lie wholly outside it, and the matrices are block-diagonal per unit).  No select on the loaded VALUES:
        // it would make the wave wait for them here, and these loads are a tile ahead of their use.
        psh = 0;
        pw[4] = 0;
        if constexpr (have_mv) {
            // a compensated row starts at any byte: the four aligned dwords it begins in and, unless it is aligned itself,
            // the one behind them (never a byte past the dword that holds the row's last sample).  An unaligned 16-byte
            // load costs the memory pipeline more than the two aligned ones.
            const uint32_t po = row_ok ? pofs : kPredBias;
            const uint32_t pa = po & ~3u;
            psh = po & 3u;
            const v4i a = __builtin_amdgcn_raw_buffer_load_b128(pred_rs, (int)pa, 0, 0);
            pw[4] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pred_rs, (int)(pa + (psh ? 16u : 12u)), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) pw[i] = (uint32_t)a[i];
        } else {
            const v4i a = __builtin_amdgcn_raw_buffer_load_b128(pred_rs, (int)(row_ok ? pofs : kPredBias), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) pw[i] = (uint32_t)a[i];
        }
        if constexpr (MODE & TQ_FWD) {
            const v4i b = __builtin_amdgcn_raw_buffer_load_b128(src_rs, (int)(row_ok ? soff : kOob), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = (uint32_t)b[i];
        }
        if constexpr (!(MODE & TQ_FWD)) {
#pragma unroll
            for (int i = 0; i < 4; i++) sw[i] = 0u;
        }
    };

    // ---- the walk over this workgroup's units ----
    TilePos pos0;
    {
        const int upf = ntxg * p.tiles_y;
        pos0.f = u_beg / upf;
        const int rem = u_beg - pos0.f * upf;
        pos0.ty = rem / ntxg;
        pos0.txg = rem - pos0.ty * ntxg;
        if (diag) {
            const int cuy = p.diag_lo + bx, cux = p.diag_d - 2 * cuy;
            pos0.f = p.diag_fgroup ? (by / p.diag_fgroup) * p.diag_fmul + by % p.diag_fgroup : by;
            pos0.ty = (cuy << p.diag_cu_log2) >> 5;
            pos0.txg = ((cux << p.diag_cu_log2) >> 5) >> 2;
        }
        pos0.f = __builtin_amdgcn_readfirstlane(pos0.f);
        pos0.ty = __builtin_amdgcn_readfirstlane(pos0.ty);
        pos0.txg = __builtin_amdgcn_readfirstlane(pos0.txg);
    }
    TilePos pos1 = pos0;
    advance(pos1);
    TilePos pos2 = pos1;
    advance(pos2);
    uint32_t srcw[4], predr[5], src_nx[4], pred_nx[5], psh = 0, psh_nx = 0;  // predr: the prediction row as loaded
    load_rows(pos0, true, load_mv(pos0, true), srcw, predr, psh);
    uint32_t mv1 = load_mv(pos1, u_beg + 1 < u_end);
    // everything loaded so far is waited for before the loop, so that inside it only its own prefetches are pending
    asm volatile("" ::"v"(srcw[0]), "v"(srcw[1]), "v"(srcw[2]), "v"(srcw[3]), "v"(predr[0]), "v"(predr[1]), "v"(predr[2]),
                 "v"(predr[3]), "v"(predr[4]), "v"(mv1), "v"(f1), "v"(f2), "v"(f3));

#pragma unroll 1
    for (int u = u_beg; u < u_end; u++) {
        // the next unit's rows and the vectors of the one after it are on their way while this one is transformed
        load_rows(pos1, u + 1 < u_end, mv1, src_nx, pred_nx, psh_nx);
        const uint32_t mv2 = load_mv(pos2, u + 2 < u_end);
        uint32_t predw[4];  // this tile's prediction row: arrived an iteration ago
#pragma unroll
        for (int i = 0; i < 4; i++) predw[i] = have_mv ? __builtin_amdgcn_alignbyte(predr[i + 1], predr[i], psh) : predr[i];

        const Geo g = geo(pos0);
        if (g.tile_ok) {  // wave-uniform (the last group of a tile row may hold fewer than four tiles)
        const int f = pos0.f;
        const int x0 = g.x0, xh = g.xh;
        const bool col_ok = g.col_ok;
        const int y0 = pos0.ty * 32;
        const int y = y0 + c;                       // lane = row c of the tile in the row layout
        const bool row_ok = y < d.height && g.half_ok;
        int16_t *lvl_pl = d.d_level + (int64_t)f * d.level_frame_stride;
        const __amdgpu_buffer_rsrc_t lvl_rs = __builtin_amdgcn_make_buffer_rsrc(lvl_pl, 0, lvl_bytes, kRsrcFlags);

        int level[16];  // the lane's column of quantised levels: x = x0 + c, y = y0 + rho(r, h)
        // byte offset of the lane's (column c, row 4 h) in the level / coefficient planes; register r adds rho(r, 0)
        // rows through the scalar offset.  Lanes outside the plane's width are sent past the end; rows past its
        // height ARE past the end of the resource.
        const int lv_off = col_ok ? (int)(((uint32_t)rowmul(y0 + 4 * h, d.level_stride) + (uint32_t)(x0 + c)) * 2u) : (int)kOob;
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
            for (int g4 = 0; g4 < 4; g4++)
                split4((uint32_t)acc[4 * g4] >> sh1, (uint32_t)acc[4 * g4 + 1] >> sh1, (uint32_t)acc[4 * g4 + 2] >> sh1,
                       (uint32_t)acc[4 * g4 + 3] >> sh1, mh[g4], ml[g4]);
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
                cf[r] = field16<sh2>(t);  // truncating Int16 store of the coefficient
                // xQuant
                const int sg = cf[r] >> 31;  // all ones for a negative coefficient
                const int q = mad24(cf[r], (int)p.q_scale, bfi(sg, q_neg, q_pos));
                const uint32_t lb = (uint32_t)q >> p.q_bits;   // level + q_bias
                psum[r >> 2] = sad_u32(lb, q_bias, psum[r >> 2]);
                level[r] = (int)(lb - q_bias);
            }
            // The prefetched rows (and vectors) are waited for HERE, before this tile's first store is issued: with
            // loads and stores in flight together the wait could only be "everything", i.e. a round trip to memory
            // per tile for the stores.  By now the loads have had the whole forward half of the tile to arrive.
            // (level[15] among the operands keeps the scheduler from moving the wait up ahead of the quantiser.)
            asm volatile("" ::"v"(src_nx[0]), "v"(src_nx[1]), "v"(src_nx[2]), "v"(src_nx[3]), "v"(pred_nx[0]), "v"(pred_nx[1]),
                         "v"(pred_nx[2]), "v"(pred_nx[3]), "v"(pred_nx[4]), "v"(mv2), "v"(level[15]), "v"(level[0]));
#pragma unroll
            for (int r = 0; r < 16; r++)
                __builtin_amdgcn_raw_buffer_store_b16((short)level[r], lvl_rs, lv_off, rho(r, 0) * row_bytes, 0);
            if (want_coef) {  // wave-uniform; stage-by-stage checks only
                const __amdgpu_buffer_rsrc_t coef_rs =
                    __builtin_amdgcn_make_buffer_rsrc(d.d_coef + (int64_t)f * d.level_frame_stride, 0, lvl_bytes, kRsrcFlags);
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
                // sum over the lanes of a TU column (c / N) -- and over both h unless N = 4, where h tells TU rows apart --
                // with DPP adds inside rows of 16 lanes (quads, halves, the row) and, past that, the two steps of
                // the classic wave reduction for N = 32 (the total arrives in the last row) or one exchange of the
                // wave's halves for N = 8, 16
#pragma unroll
                for (int i = 0; i < ns; i++) {
                    s[i] = dpp_add<0xb1>(s[i]);                          // quad_perm [1,0,3,2]
                    s[i] = dpp_add<0x4e>(s[i]);                          // quad_perm [2,3,0,1]
                    if constexpr (N >= 8) s[i] = dpp_add<0x141>(s[i]);   // row_half_mirror
                    if constexpr (N >= 16) s[i] = dpp_add<0x140>(s[i]);  // row_mirror
                    if constexpr (N == 32) {
                        s[i] = dpp_add<0x142, 0xa>(s[i]);  // row_bcast15: rows 1, 3 += row 0, 2
                        s[i] = dpp_add<0x143, 0xc>(s[i]);  // row_bcast31: rows 2, 3 += rows 0 + 1
                    } else if constexpr (N >= 8) {
                        uint32_t lo = s[i], hi = s[i];
                        swap32(lo, hi);  // lo = {lower half, lower half}, hi = {upper half, upper half}
                        s[i] = lo + hi;
                    }
                }
                // one lane per TU column writes; TU rows below the picture are past the end of the resource
                const __amdgpu_buffer_rsrc_t sum_rs = __builtin_amdgcn_make_buffer_rsrc(
                    d.d_abs_sum + (p.diag_fgroup ? (int64_t)(by / p.diag_fgroup) * d.abs_sum_frame_stride +
                                                       (int64_t)(by % p.diag_fgroup) * p.diag_sum_pair
                                                 : (int64_t)f * d.abs_sum_frame_stride),
                    0, (uint32_t)((d.height >> L2) * (d.width >> L2)) * 4u, kRsrcFlags);
                const bool writer = (N == 32 ? lane == 63 : (c & (N - 1)) == 0 && (N == 4 || h == 0)) && col_ok;
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
            // QP / 6 >= log2 N - 1: the scale has taken the shift in (quant_constants) and the stage is one multiply
            if (p.dq_shift == 0) {  // wave-uniform
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    int v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = mul24(level[4 * g4 + k], p.dq_scale);
                    // v_cvt_pk_i16_i32 saturates: the clip and the pack in one
                    const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[0], v[1]));
                    const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[2], v[3]));
                    split2x2(p0, p1, dh[g4], dl[g4]);
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    int v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = (mul24(level[4 * g4 + k], p.dq_scale) + p.dq_rnd) >> p.dq_shift;
                    const uint32_t p0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[0], v[1]));
                    const uint32_t p1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(v[2], v[3]));
                    split2x2(p0, p1, dh[g4], dl[g4]);
                }
            }
            // ================= inverse pass 1 (columns): coefficients^T x T =================
            // The low-byte product is taken TWICE (the matrix pipe has room, the vector ALU has none): twice the sum puts
            // the Int16 that (sum + 64) >> 7 leaves at bits [8, 24), on a byte boundary, and saves the shift per value.
            v16i ah = mfma(as_v4i(dh), f3, splat16(0));
            v16i al = mfma(as_v4i(dl), f3, mfma(as_v4i(dl), f3, cin_i1));
            // (sum + 64) >> 7, store-then-clip = wrap (src/pixel.cpp:334-342); lane = picture row now
            uint32_t uh[4], ul[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = ((uint32_t)ah[4 * g4 + k] << 9) + (uint32_t)al[4 * g4 + k];
                split4_b8(t[0], t[1], t[2], t[3], uh[g4], ul[g4]);
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
            for (int g4 = 0; g4 < 4; g4++) {
                // (sum + 2048) >> 12, wrap (truncating pack), + pred, Clip (src/pixel.cpp:757-761): the saturating
                // Int16 add cannot change a sum that is clipped to 0..255 anyway
                uint32_t t[4];
#pragma unroll
                for (int k = 0; k < 4; k++) t[k] = (((uint32_t)ah[4 * g4 + k] << 8) + (uint32_t)al[4 * g4 + k]) << 4;
                const v2s r0 = __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, perm(t[1], t[0], 0x07060302u)),
                                                             __builtin_bit_cast(v2s, perm(0u, pg[g4], 0x0c010c00u)));
                const v2s r1 = __builtin_elementwise_add_sat(__builtin_bit_cast(v2s, perm(t[3], t[2], 0x07060302u)),
                                                             __builtin_bit_cast(v2s, perm(0u, pg[g4], 0x0c030c02u)));
                rq[g4] = perm(sat_pk_u8_i16(__builtin_bit_cast(uint32_t, r1)), sat_pk_u8_i16(__builtin_bit_cast(uint32_t, r0)),
                              0x05040100u);
            }
            // back to "16 consecutive bytes per lane": dwords 4 h .. 4 h + 3
            uint32_t ro[4] = {rq[0], rq[2], rq[1], rq[3]};  // dword h, 4 + h, 2 + h, 6 + h
            swap32(ro[0], ro[1]);  // -> dwords 4 h, 4 h + 1
            swap32(ro[2], ro[3]);  // -> dwords 4 h + 2, 4 h + 3
            const __amdgpu_buffer_rsrc_t rec_rs = __builtin_amdgcn_make_buffer_rsrc(
                d.d_rec + (int64_t)f * d.rec_frame_stride, 0, (uint32_t)((int64_t)(d.height - 1) * d.rec_stride + d.width), kRsrcFlags);
            const uint32_t off = (uint32_t)rowmul(y, d.rec_stride) + (uint32_t)xh;
            __builtin_amdgcn_raw_buffer_store_b128(v4i{(int)ro[0], (int)ro[1], (int)ro[2], (int)ro[3]}, rec_rs, (int)(row_ok ? off : kOob), 0, 0);
        }
        } else {  // the same wait on the path that skips the tile
            asm volatile("" ::"v"(src_nx[0]), "v"(src_nx[1]), "v"(src_nx[2]), "v"(src_nx[3]), "v"(pred_nx[0]), "v"(pred_nx[1]),
                         "v"(pred_nx[2]), "v"(pred_nx[3]), "v"(pred_nx[4]), "v"(mv2));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) srcw[i] = src_nx[i];
#pragma unroll
        for (int i = 0; i < 5; i++) predr[i] = pred_nx[i];
        psh = psh_nx;
        mv1 = mv2;
        pos0 = pos1;
        pos1 = pos2;
        advance(pos2);
    }
    trace_end();
}

}  // namespace x265amd
