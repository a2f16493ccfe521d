// me_pu.hip -- EVERY inter PU shape of a CTU in one exhaustive pass (remainder: "8x8 leaves and
// the rectangular PUs 2NxN, Nx2N at 64 / 32 / 16"; BASELINE.json north_star: "per-CTU/PU integer-pel motion search").
//
// Ten shapes, 169 PUs per 64x64 CTU: 64x64, 64x32 (2), 32x64 (2), 32x32 (4), 32x16 (8), 16x32 (8), 16x16 (16),
// 16x8 (32), 8x16 (32), 8x8 (64).  All of them are sums of the CTU's sixty-four 8x8 SADs at one displacement
// (the reference's primitives take any number of rows: xSadN[log2(w) - 2](h, ..), src/pixel.cpp:30-101), so the
// kernel forms those once per displacement and adds them up the tree.  me_tree.hip's layout -- one lane = the whole
// CTU at four displacements, one running winner per PU in registers -- stops at 21 PUs; here the CTU is spread
// over a ROW OF 16 LANES instead (the unit DPP instructions work in):
//
//   lane r16 of a row  = one 16x16 quadrant of the CTU: 32x32 block Q = r16 >> 2 (raster 2 J + K), quadrant l = r16 & 3
//                        inside it (raster 2 j' + k'); its sixteen current rows stay in registers (64 dwords)
//   a row of 16 lanes  = one task: one dy and four consecutive dx (v_qsad_pk_u16_u8 against the LDS window)
//   4 rows x 4 waves   = 16 tasks in flight per step
//
// A lane owns the nine PUs inside its quadrant (four 8x8, two 16x8, two 8x16, the 16x16; sums on the packed u16
// quadruples, winners as single-dword keys sad << IDXB | raster index under v_min_u32; with 16 index bits the key is
// the SAD's own field of the packed sums over the index, one instruction).  Above that, DPP adds inside the quad give
// 32x16 / 16x32 / 32x32 and across quads 64x32 / 32x64 / 64x64; each lane tracks ONE 2:1 PU of its 32x32 block, the
// block, ONE 2:1 PU of the CTU and the CTU (which one: by its position), 13 PUs a lane -- the last three for one dx in
// four only (dx = l mod 4: these sums repeat over the lanes of a quad, and the winners meet at the end).  Every PU keeps its winner under its OWN legality (the window is staged unclipped, zeros outside the
// picture) with the reference's argmin idiom: first strict minimum in raster order (src/encode.cpp:725-727,
// 762-765).  Every output entry equals what the plain exhaustive search returns for blocks of that shape tiling
// the picture -- that is the parity test (oracle: orc_me_fullsearch_wh / the reference's xSadN through
// oracle/ref_shim.cpp).  CTUs at least `range` away from every border skip the legality arithmetic.
#include <type_traits>

#include "common.h"
#include "me_common.h"

namespace x265amd {

constexpr int kPuShapes = X265AMD_PU_SHAPES;
constexpr int kLanePUs = 13;

struct PuOut {
    x265amd_mv *o[kPuShapes];
};
__device__ const int kPuW[kPuShapes] = {64, 64, 32, 32, 32, 16, 16, 16, 8, 8};
__device__ const int kPuH[kPuShapes] = {64, 32, 64, 32, 16, 32, 16, 8, 16, 8};

template <int CTRL> __device__ __forceinline__ uint32_t dpp(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}

template <int IDXB>
__global__ __launch_bounds__(256, 2) void me_pu_kernel(const uint8_t *__restrict__ cur_base, const uint8_t *__restrict__ ref_base,
                                                       const PuOut out, const MeGeom g, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *win = smem;  // [2R + 64][pitch]
    __shared__ unsigned long long wbest[4][16][kLanePUs];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int drow = lane >> 4, r16 = lane & 15;
    const int Q = r16 >> 2, l = r16 & 3;
    const int J = Q >> 1, K = Q & 1;
    const int jq = 2 * J + (l >> 1), kq = 2 * K + (l & 1);  // the lane's quadrant: rows 16 jq .., columns 16 kq ..
    int ctx, cty, bz;
    me_xcd_block(ctx, cty, bz);
    const int f = bz / g.nrefs, rr = bz - f * g.nrefs;
    const int cx = ctx * 64, cy = cty * 64;
    const int R = g.range, span = 2 * R + 1;
    const int W = g.width, H = g.height;
    const uint8_t *cur = cur_base + f * g.cur_fs;
    const uint8_t *ref = ref_base + f * g.ref_fs + rr * g.ref_rs;
    const int wx0 = (cx - R) & ~3;  // window origin on a dword boundary (may be negative)
    const int a0 = cx - R - wx0;    // window column of dx = -R
    const int ngx = (a0 + span + 3) >> 2;
    const int win_rows = 2 * R + 64;

    // ---- stage the window, unclipped: zero outside the picture ----
    for (int i = tid; i < win_rows * pitch; i += 256) {
        const int wy = i / pitch, wq = i - wy * pitch;
        const int gx = wx0 + 4 * wq, gy = cy - R + wy;
        uint32_t v = 0;
        if (gy >= 0 && gy < H && gx >= 0 && gx + 4 <= W) v = *reinterpret_cast<const uint32_t *>(ref + (int64_t)gy * g.stride + gx);
        win[i] = v;
    }
    // ---- the lane's quadrant of the current CTU ----
    uint32_t cw[16][4];
    {
        const uint32_t *cb = reinterpret_cast<const uint32_t *>(cur + (int64_t)(cy + 16 * jq) * g.stride + cx + 16 * kq);
        const int cstr = g.stride >> 2;
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) cw[r][k] = cb[r * cstr + k];
    }
    __syncthreads();

    // ---- winners ----
    constexpr bool kKey16 = 16 + IDXB <= 32;  // does the 16x16 SAD (<= 65280) fit a single-dword key?
    constexpr bool kKeyPk = IDXB == 16;       // keys cut straight out of the packed sums
    uint32_t k8[4], k168[2], k816[2], k16 = 0xffffffffu;  // keys: sad << IDXB | raster index
    uint32_t bs[5], bi[5];  // (sad, index) pairs: 16x16 when it has no key, the 2:1 PU and the block at 32, the same at 64
#pragma unroll
    for (int i = 0; i < 4; i++) k8[i] = 0xffffffffu;
    k168[0] = k168[1] = k816[0] = k816[1] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 5; i++) bs[i] = bi[i] = 0xffffffffu;

    // which 2:1 PU the lane tracks.  After the add with the horizontal neighbour, quadrants (j', 0) and (j', 1) both hold
    // the 32x16 PU j' of their block; after the add with the vertical neighbour, (0, k') and (1, k') hold the 16x32 PU k'.
    // l = 0 takes the upper 32x16, l = 3 the lower, l = 2 the left 16x32, l = 1 the right; blocks Q likewise one level up.
    const bool hor32 = l == 0 || l == 3, hor64 = Q == 0 || Q == 3;
    // position and size of the lane's PUs above the quadrant, for the legality tests
    const int ox32r = 32 * K + ((l == 1) ? 16 : 0), oy32r = 32 * J + ((l == 3) ? 16 : 0);
    const int w32r = hor32 ? 32 : 16, h32r = hor32 ? 16 : 32;
    const int ox64r = (Q == 1) ? 32 : 0, oy64r = (Q == 3) ? 32 : 0;
    const int w64r = hor64 ? 64 : 32, h64r = hor64 ? 32 : 64;

    const int ntask = ngx * span;
    const int ts = wave * 4 + drow;  // the row's place among the 16 tasks of a step
    int dyi = ts / ngx, gx = ts - dyi * ngx;

    auto run = [&](auto border_tag) {
        constexpr bool BORDER = decltype(border_tag)::value;
        for (int base = 0; base < ntask; base += 16) {
            const bool valid = base + ts < ntask;     // uniform in the row of 16 lanes
            const int dyc = valid ? dyi : 0, gxc = valid ? gx : 0;
            const uint32_t *wrow = win + (dyc + 16 * jq) * pitch + gxc + 4 * kq;
            uint64_t acc[2][2] = {{0, 0}, {0, 0}};    // the quadrant's 8x8 SADs [row half][column half], four dx packed
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t *p = wrow + r * pitch;
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
                uint64_t &aL = acc[r >> 3][0], &aR = acc[r >> 3][1];
                aL = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d1 << 32) | d0, cw[r][0], aL);
                aL = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d2 << 32) | d1, cw[r][1], aL);
                aR = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d3 << 32) | d2, cw[r][2], aR);
                aR = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)d4 << 32) | d3, cw[r][3], aR);
            }
            // packed sums: every field stays below 65536 (a 16x16 SAD is at most 65280), so the fields do not mix
            const uint64_t h8[2] = {acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]};  // 16x8: upper, lower
            const uint64_t v8[2] = {acc[0][0] + acc[1][0], acc[0][1] + acc[1][1]};  // 8x16: left, right
            const uint64_t s16 = h8[0] + h8[1];

            const int dy = dyc - R;
            const int dxb = wx0 - cx + 4 * gxc;  // dx of the first of the four
            const uint32_t idxrow = (uint32_t)(dyc * span + R);
            // rows: a PU at row offset oy of height h is legal iff 0 <= cy + oy + dy and cy + oy + dy + h <= H
            bool oky0 = true, oky1 = true, oky01 = true, oky32r = true, oky32 = true, oky64r = true, oky64 = true;
            if constexpr (BORDER) {
                const int Y = cy + 16 * jq + dy;
                oky0 = (unsigned)Y <= (unsigned)(H - 8);
                oky1 = (unsigned)(Y + 8) <= (unsigned)(H - 8);
                oky01 = (unsigned)Y <= (unsigned)(H - 16);
                oky32r = (unsigned)(cy + oy32r + dy) <= (unsigned)(H - h32r);
                oky32 = (unsigned)(cy + 32 * J + dy) <= (unsigned)(H - 32);
                oky64r = (unsigned)(cy + oy64r + dy) <= (unsigned)(H - h64r);
                oky64 = (unsigned)(cy + dy) <= (unsigned)(H - 64);
            }
            // a lane meets its candidates in raster order: strict < keeps the first
            auto paired = [&](int i, uint32_t sad, uint32_t idx, bool ok) {
                if (ok && sad < bs[i]) {
                    bs[i] = sad;
                    bi[i] = idx;
                }
            };
            uint32_t s32[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int dx = dxb + q;
                const bool inr = valid && (unsigned)(dx + R) <= (unsigned)(2 * R);
                const uint32_t idx = idxrow + (uint32_t)dx;  // (dy + R) * span + dx + R
                // kKeyPk: the key is the SAD's own 16-bit field of the packed sums above the 16-bit index -- one instruction
                // (v_lshl_or_b32 / v_and_or_b32) -- and a displacement outside the range carries 0xffff there instead,
                // above every SAD (<= 0xff00), through the index it is or-ed with
                const uint32_t idxp = (kKeyPk && !inr) ? (idx | 0xffff0000u) : idx;
                auto fld = [&](uint64_t v) { return (uint32_t)(v >> (16 * q)) & 0xffffu; };
                bool okx0 = true, okx1 = true, okx01 = true, okx32r = true;
                if constexpr (BORDER) {
                    const int X = cx + 16 * kq + dx;
                    okx0 = (unsigned)X <= (unsigned)(W - 8);
                    okx1 = (unsigned)(X + 8) <= (unsigned)(W - 8);
                    okx01 = (unsigned)X <= (unsigned)(W - 16);
                    okx32r = (unsigned)(cx + ox32r + dx) <= (unsigned)(W - w32r);
                }
                auto keyed = [&](uint32_t &best, uint64_t v, bool ok) {
                    uint32_t key;
                    if constexpr (kKeyPk) {
                        const uint32_t dw = q < 2 ? (uint32_t)v : (uint32_t)(v >> 32);
                        key = (q & 1) ? ((dw & 0xffff0000u) | idxp) : ((dw << 16) | idxp);
                        if constexpr (BORDER) key = ok ? key : 0xffffffffu;
                    } else {
                        key = (fld(v) << IDXB) + idx;
                        key = (inr && ok) ? key : 0xffffffffu;
                    }
                    best = min(best, key);
                };
                keyed(k8[0], acc[0][0], okx0 && oky0);
                keyed(k8[1], acc[0][1], okx1 && oky0);
                keyed(k8[2], acc[1][0], okx0 && oky1);
                keyed(k8[3], acc[1][1], okx1 && oky1);
                keyed(k168[0], h8[0], okx01 && oky0);
                keyed(k168[1], h8[1], okx01 && oky1);
                keyed(k816[0], v8[0], okx0 && oky01);
                keyed(k816[1], v8[1], okx1 && oky01);
                const uint32_t a = fld(s16);
                if constexpr (kKey16)
                    keyed(k16, s16, okx01 && oky01);
                else
                    paired(0, a, idx, inr && okx01 && oky01);
                // up the tree: the quad = a 32x32 block.  One add brings every lane ITS 2:1 PU (l = 0: + the quadrant to the
                // right, 1: + the one below, 2: + the one above, 3: + the one to the left), a second the block
                const uint32_t r32 = a + dpp<0x8d>(a);  // quad_perm [1,3,0,2]
                paired(1, r32, idx, inr && okx32r && oky32r);
                s32[q] = r32 + dpp<0x1b>(r32);          // quad_perm [3,2,1,0]: the other half of the block
            }
            // From the block upwards the sums repeat over the lanes of a quad, so lane l follows them for dx = dxb + l only
            // (the winners meet across the quad at the end).  Same l = same dx in every quad of the row: the adds across
            // quads stay single values.
            {
                const uint32_t sl = l == 0 ? s32[0] : l == 1 ? s32[1] : l == 2 ? s32[2] : s32[3];
                const int dx = dxb + l;
                const bool inr = valid && (unsigned)(dx + R) <= (unsigned)(2 * R);
                const uint32_t idx = idxrow + (uint32_t)dx;
                bool okx32 = true, okx64r = true, okx64 = true;
                if constexpr (BORDER) {
                    okx32 = (unsigned)(cx + 32 * K + dx) <= (unsigned)(W - 32);
                    okx64r = (unsigned)(cx + ox64r + dx) <= (unsigned)(W - w64r);
                    okx64 = (unsigned)(cx + dx) <= (unsigned)(W - 64);
                }
                const uint32_t v64 = sl + dpp<0x128>(sl);   // row_ror:8: + the block below / above
                const uint32_t hp = sl + dpp<0x124>(sl);    // row_ror:4: + the block of the quad before (right for Q = 1, 3)
                const uint32_t hn = sl + dpp<0x12c>(sl);    // row_ror:12: + the block of the quad after (right for Q = 0, 2)
                const uint32_t s64 = v64 + dpp<0x124>(v64); // left column + right column, whichever the lane holds
                const uint32_t r64 = Q == 0 ? hn : Q == 3 ? hp : v64;  // upper 64x32, lower 64x32, 32x64 right (Q = 1) / left (Q = 2)
                paired(2, sl, idx, inr && okx32 && oky32);
                paired(3, r64, idx, inr && okx64r && oky64r);
                paired(4, s64, idx, inr && okx64 && oky64);
            }
            gx += 16;
            while (gx >= ngx) {
                gx -= ngx;
                dyi++;
            }
        }
    };
    // every PU legal at every displacement?  (workgroup-uniform)
    if (cx - R >= 0 && cx + 64 + R <= W && cy - R >= 0 && cy + 64 + R <= H)
        run(std::false_type{});
    else
        run(std::true_type{});

    // ---- winners: 64-bit key (sad << 32 | raster index) across the four rows of a wave, then across the four waves ----
    unsigned long long key[kLanePUs];
    auto unkey = [&](uint32_t k) -> unsigned long long {
        const bool none = kKeyPk ? (k >> 16) == 0xffffu : k == 0xffffffffu;
        return none ? ~0ull : ((unsigned long long)(k >> IDXB) << 32) | (k & ((1u << IDXB) - 1u));
    };
#pragma unroll
    for (int i = 0; i < 4; i++) key[i] = unkey(k8[i]);
    key[4] = unkey(k168[0]);
    key[5] = unkey(k168[1]);
    key[6] = unkey(k816[0]);
    key[7] = unkey(k816[1]);
    key[8] = kKey16 ? unkey(k16) : ((unsigned long long)bs[0] << 32) | bi[0];
#pragma unroll
    for (int i = 1; i < 5; i++) key[8 + i] = ((unsigned long long)bs[i] << 32) | bi[i];
#pragma unroll
    for (int p = 0; p < kLanePUs; p++) {
        unsigned long long k = key[p];
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const unsigned long long other = __shfl_xor(k, o);
            k = other < k ? other : k;
        }
        if (drow == 0) wbest[wave][r16][p] = k;
    }
    __syncthreads();
    if (tid < 16 * kLanePUs) {
        const int r = tid / kLanePUs, p = tid - r * kLanePUs;
        const int Qr = r >> 2, lr = r & 3, Jr = Qr >> 1, Kr = Qr & 1;
        unsigned long long k = ~0ull;
        const int lanes = p >= 10 ? 4 : 1;  // from the block upwards the four lanes of a quad each followed one dx in four
        for (int w = 0; w < 4; w++)
            for (int i = 0; i < lanes; i++) {
                const unsigned long long o = wbest[w][lanes == 4 ? 4 * Qr + i : r][p];
                k = o < k ? o : k;
            }
        const int jr = 2 * Jr + (lr >> 1), kr = 2 * Kr + (lr & 1);
        int s, X, Y;
        bool write = true;
        if (p < 4) {  // 8x8
            s = X265AMD_PU_8x8, X = 8 * ctx + 2 * kr + (p & 1), Y = 8 * cty + 2 * jr + (p >> 1);
        } else if (p < 6) {  // 16x8
            s = X265AMD_PU_16x8, X = 4 * ctx + kr, Y = 8 * cty + 2 * jr + (p - 4);
        } else if (p < 8) {  // 8x16
            s = X265AMD_PU_8x16, X = 8 * ctx + 2 * kr + (p - 6), Y = 4 * cty + jr;
        } else if (p == 8) {
            s = X265AMD_PU_16x16, X = 4 * ctx + kr, Y = 4 * cty + jr;
        } else if (p == 9) {
            if (lr == 0 || lr == 3)
                s = X265AMD_PU_32x16, X = 2 * ctx + Kr, Y = 4 * cty + 2 * Jr + (lr == 3 ? 1 : 0);
            else
                s = X265AMD_PU_16x32, X = 4 * ctx + 2 * Kr + (lr == 1 ? 1 : 0), Y = 2 * cty + Jr;
        } else if (p == 10) {
            write = lr == 0;
            s = X265AMD_PU_32x32, X = 2 * ctx + Kr, Y = 2 * cty + Jr;
        } else if (p == 11) {
            write = lr == 0;
            if (Qr == 0 || Qr == 3)
                s = X265AMD_PU_64x32, X = ctx, Y = 2 * cty + (Qr == 3 ? 1 : 0);
            else
                s = X265AMD_PU_32x64, X = 2 * ctx + (Qr == 1 ? 1 : 0), Y = cty;
        } else {
            write = r == 0;
            s = X265AMD_PU_64x64, X = ctx, Y = cty;
        }
        if (write && out.o[s]) {
            x265amd_mv mv;
            const int c = (int)(k & 0xffffffffu);
            mv.mvx = (int16_t)(c % span - R);
            mv.mvy = (int16_t)(c / span - R);
            mv.sad = (uint32_t)(k >> 32);
            const int nx = W / kPuW[s], ny = H / kPuH[s];
            const int64_t plane = (int64_t)f * g.nrefs + rr;
            out.o[s][plane * nx * ny + (int64_t)Y * nx + X] = mv;
        }
    }
}

int me_pu_device_setup(DeviceState &ds)
{
    int rc = allow_full_lds((const void *)me_pu_kernel<16>, ds);
    if (rc != X265AMD_OK) return rc;
    return allow_full_lds((const void *)me_pu_kernel<17>, ds);
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_me_fullsearch_pu_all(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                                            int range, int nframes, int64_t cur_frame_stride, int64_t ref_frame_stride,
                                            int nrefs, int64_t ref_ref_stride, x265amd_mv *const *d_out, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_cur || !d_ref || !d_out) return X265AMD_ERR_ARG;
    PuOut out;
    bool any = false;
    for (int s = 0; s < kPuShapes; s++) {
        out.o[s] = d_out[s];
        any = any || d_out[s];
    }
    if (!any) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % 64 || height % 64 || stride < width) return X265AMD_ERR_ARG;
    if (range < 0 || range > 180 || nframes < 0 || nrefs < 1) return X265AMD_ERR_ARG;  // (2 range + 1)^2 = 361^2 <= 2^17: the keys
    if ((stride & 3) || ((uintptr_t)d_cur & 3) || ((uintptr_t)d_ref & 3) || (cur_frame_stride & 3) ||
        (ref_frame_stride & 3) || (ref_ref_stride & 3))
        return X265AMD_ERR_ARG;
    if (nframes == 0) return X265AMD_OK;
    if ((int64_t)nframes * nrefs > 65535) return X265AMD_ERR_ARG;
    // the widest window: 3 leading bytes of alignment + (2R + 1) displacements + 63 more columns + 4 of the last qsad
    const int pitch = ((3 + 2 * range + 64 + 4 + 3) / 4 + 1) | 1;
    const size_t lds = (size_t)(2 * range + 64) * pitch * 4;
    if (lds + 8192 > (size_t)lds_limit_bytes()) return X265AMD_ERR_LDS;
    MeGeom g;
    g.width = width;
    g.height = height;
    g.stride = stride;
    g.range = range;
    g.nrefs = nrefs;
    g.ctus_x = width / 64;
    g.ctus_y = height / 64;
    g.cur_fs = cur_frame_stride;
    g.ref_fs = ref_frame_stride;
    g.ref_rs = ref_ref_stride;
    const dim3 grid(g.ctus_x, g.ctus_y, (unsigned)(nframes * nrefs));
    const int span = 2 * range + 1;
    if (span * span <= (1 << 16))
        hipLaunchKernelGGL(me_pu_kernel<16>, grid, dim3(256), lds, (hipStream_t)stream, d_cur, d_ref, out, g, pitch);
    else
        hipLaunchKernelGGL(me_pu_kernel<17>, grid, dim3(256), lds, (hipStream_t)stream, d_cur, d_ref, out, g, pitch);
    return hip_status(hipGetLastError());
}
