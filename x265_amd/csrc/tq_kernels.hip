// tq_kernels.hip -- residual + forward transform + quantisation, and
// dequantisation + inverse transform + reconstruction, for gfx950.
//
// Replaces xSubDct / xQuant / xDeQuant / xIDctAdd of the reference
// (src/pixel.cpp:591-776) for whole picture planes.
//
// Work decomposition: a 32x32 pixel TILE per 32 lanes (two tiles per wave64,
// eight per 256-thread workgroup).  A tile holds (32/N)^2 transform units.
//   pass 1 : lane r owns picture ROW r of the tile  (32 residuals in VGPRs),
//            runs the N-point butterfly on each of its 32/N segments;
//   LDS    : the Int16 intermediate (the reference's piTmp1; the truncating
//            store is semantic) is written row-wise and read column-wise --
//            17-dword row pitch makes both conflict-free;
//   pass 2 : lane c owns COLUMN c, butterflies over y, quantises in registers,
//            reduces sum|level| across the N lanes of each TU (xQuant's return);
//   inverse: the same lane already holds the column the inverse's first pass
//            needs, so dequant + vertical inverse run in registers, one more LDS
//            transpose, horizontal inverse, + pred, clip, 32-byte row store.
// MFMA is deliberately not used: every sum must equal the reference's Int32 sum.
#include "common.h"
#include "hevc_transform.h"
#include "tq_common.h"

namespace x265amd {

struct __attribute__((packed, aligned(1))) U128 {
    uint32_t v[4];
};
struct __attribute__((packed, aligned(1))) U32 {
    uint32_t v;
};

constexpr int kPitch = 34;  // Int16 elements per LDS row: 17 dwords (odd) => conflict-free both ways

__device__ __forceinline__ v2s as_v2s(uint32_t u) { return __builtin_bit_cast(v2s, u); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }
// bytes (2h, 2h+1) of a dword as two u16 lanes (v_perm_b32; selector 0x0c = constant zero)
__device__ __forceinline__ v2s byte_pair(uint32_t w, int h)
{
    return as_v2s(__builtin_amdgcn_perm(0u, w, h ? 0x0c030c02u : 0x0c010c00u));
}

// two results whose wanted 16 bits sit at [23:8] / at [31:16] -> one Int16 pair
__device__ __forceinline__ v2s pack_mid(int a, int b) { return as_v2s(__builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x06050201u)); }
__device__ __forceinline__ v2s pack_hi(int a, int b) { return as_v2s(__builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x07060302u)); }

__device__ __forceinline__ int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int wrap16(int v) { return (int)(short)v; }

// 32 bytes of one picture row -> 8 dwords.  `full` (tile-uniform): the whole 32-byte
// span is inside the plane; otherwise dwords at x >= wvalid read as zero.
__device__ __forceinline__ void load_row32(const uint8_t *p, bool full, int wvalid, uint32_t (&w)[8])
{
    if (full) {
        U128 a = *reinterpret_cast<const U128 *>(p);
        U128 b = *reinterpret_cast<const U128 *>(p + 16);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            w[i] = a.v[i];
            w[4 + i] = b.v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (4 * i < wvalid) ? reinterpret_cast<const U32 *>(p + 4 * i)->v : 0u;
    }
}

template <int N, int MODE>
__global__ __launch_bounds__(256, 4) void tq_kernel(const TqParams p)
{
    __shared__ int16_t lds[8][32][kPitch];
    constexpr int L2 = (N == 4) ? 2 : (N == 8) ? 3 : (N == 16) ? 4 : 5;
    constexpr int NSEG = 32 / N;
    const x265amd_tq_desc &d = p.d;

    const int lt = threadIdx.x >> 5;  // tile within the workgroup
    const int ln = threadIdx.x & 31;  // row (pass 1) / column (pass 2) within the tile
    const int tile = blockIdx.x * 8 + lt;
    const bool tile_ok = tile < p.ntiles;
    const int tpf = p.tiles_x * p.tiles_y;
    const int f = tile_ok ? tile / tpf : 0;
    const int trem = tile_ok ? tile - f * tpf : 0;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int x0 = tx * 32, y0 = ty * 32;
    const int wvalid = min(32, d.width - x0);   // multiples of N
    const int hvalid = min(32, d.height - y0);
    const bool full_w = wvalid == 32;
    const int tus_x = d.width / N;  // TUs per picture row

    int16_t(*T)[kPitch] = lds[lt];

    // ---- the lane's picture row: source and prediction (kept for the reconstruction) ----
    const bool row_ok = tile_ok && ln < hvalid;
    uint32_t predw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) predw[i] = 0;
    if (row_ok) {
        const int y = y0 + ln;
        const uint8_t *pp = d.d_pred + (int64_t)f * d.pred_frame_stride;
        if (d.d_mv) {
            // integer-pel motion compensation fused into the load: one vector per mv_ctu block.
            // mv_ctu >= 16, so a 16-pixel half row never straddles two vectors.
            const x265amd_mv *mvf = d.d_mv + (int64_t)f * d.mv_frame_stride;
            const int ctus_x = d.width / d.mv_ctu;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int xh = x0 + 16 * h;
                if (xh < d.width) {
                    const x265amd_mv m = mvf[(y / d.mv_ctu) * ctus_x + xh / d.mv_ctu];
                    const uint8_t *q = pp + (int64_t)(y + (m.mvy >> d.mv_shift)) * d.pred_stride + xh + (m.mvx >> d.mv_shift);
                    if (xh + 16 <= d.width) {
                        U128 a = *reinterpret_cast<const U128 *>(q);
#pragma unroll
                        for (int i = 0; i < 4; i++) predw[4 * h + i] = a.v[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (xh + 4 * i < d.width) predw[4 * h + i] = reinterpret_cast<const U32 *>(q + 4 * i)->v;
                    }
                }
            }
        } else {
            load_row32(pp + (int64_t)y * d.pred_stride + x0, full_w, wvalid, predw);
        }
    }

    int16_t lvl[32];  // the lane's column of quantised levels (pass 2 onwards)

    if constexpr (MODE & TQ_FWD) {
        // ================= pass 1: residual + horizontal transform =================
        uint32_t srcw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) srcw[i] = 0;
        if (row_ok)
            load_row32(d.d_src + (int64_t)f * d.src_frame_stride + (int64_t)(y0 + ln) * d.src_stride + x0, full_w,
                       wvalid, srcw);
        constexpr int sh1 = L2 - 1, rnd1 = 1 << (sh1 - 1);
        if constexpr (N >= 8) {
            // packed: residual pairs (|r| <= 255), butterflies in Int16 lanes, two MACs per v_dot2c
            v2s P[16];
#pragma unroll
            for (int m = 0; m < 16; m++) P[m] = byte_pair(srcw[m >> 1], m & 1) - byte_pair(predw[m >> 1], m & 1);
#pragma unroll
            for (int s = 0; s < NSEG; s++) {
                int yv[N];
                constexpr int SC = 8 - sh1;  // bits [sh1+15 : sh1] of the sum land on [23:8]
                fwd_butterfly_pk<N, SC>(&P[s * (N / 2)], yv, rnd1 << SC);
#pragma unroll
                for (int k = 0; k < N; k += 2)  // truncating Int16 stores
                    *reinterpret_cast<v2s *>(&T[ln][s * N + k]) = pack_mid(yv[k], yv[k + 1]);
            }
        } else {
            int res[32];
#pragma unroll
            for (int i = 0; i < 32; i++)
                res[i] = (int)((srcw[i >> 2] >> (8 * (i & 3))) & 255u) - (int)((predw[i >> 2] >> (8 * (i & 3))) & 255u);
#pragma unroll
            for (int s = 0; s < NSEG; s++) {
                int yv[N];
                bool dst = false;
                if (d.d_modes && row_ok && s * 4 < wvalid)
                    dst = use_dst(4, d.d_modes[(int64_t)f * d.modes_frame_stride + ((y0 + ln) >> 2) * tus_x + (x0 >> 2) + s], true);
                if (dst)
                    fwd_dst4(&res[s * N], yv);
                else
                    fwd_butterfly<N>(&res[s * N], yv);
#pragma unroll
                for (int k = 0; k < N; k += 2) {
                    const int a = (yv[k] + rnd1) >> sh1, b = (yv[k + 1] + rnd1) >> sh1;  // truncating Int16 stores
                    *reinterpret_cast<uint32_t *>(&T[ln][s * N + k]) = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
                }
            }
        }
        __syncthreads();

        // ================= pass 2: vertical transform + quantisation =================
        const bool col_ok = tile_ok && ln < wvalid;
        // the lane's column, as mirror pairs (x[k], x[N-1-k]) per segment
        v2s Q[16];
#pragma unroll
        for (int s = 0; s < NSEG; s++)
#pragma unroll
            for (int k = 0; k < N / 2; k++) Q[s * (N / 2) + k] = v2s{T[s * N + k][ln], T[s * N + N - 1 - k][ln]};
        __syncthreads();  // the inverse reuses the tile buffer

        constexpr int sh2 = L2 + 6, rnd2 = 1 << (sh2 - 1);
        int16_t *lvp = d.d_level + (int64_t)f * d.level_frame_stride + (int64_t)y0 * d.level_stride + x0 + ln;
        int16_t *cfp = d.d_coef ? d.d_coef + (int64_t)f * d.level_frame_stride + (int64_t)y0 * d.level_stride + x0 + ln : nullptr;
#pragma unroll
        for (int s = 0; s < NSEG; s++) {
            int yv[N];  // includes the rounding offset; scaled by 2^SC2 (wanted bits at [31:16]) when N >= 8
            constexpr int SC2 = N >= 8 ? 16 - sh2 : 0;
            if constexpr (N >= 8) {
                fwd_butterfly_wide<N, SC2>(&Q[s * (N / 2)], yv, (int)((unsigned)rnd2 << SC2));
            } else {
                int v[N];
#pragma unroll
                for (int k = 0; k < N / 2; k++) {
                    v[k] = Q[s * (N / 2) + k].x;
                    v[N - 1 - k] = Q[s * (N / 2) + k].y;
                }
                bool dst = false;
                if (d.d_modes && col_ok && s * 4 < hvalid)
                    dst = use_dst(4, d.d_modes[(int64_t)f * d.modes_frame_stride + ((y0 >> 2) + s) * tus_x + ((x0 + ln) >> 2)], false);
                if (dst)
                    fwd_dst4(v, yv);
                else
                    fwd_butterfly<N>(v, yv);
#pragma unroll
                for (int m = 0; m < N; m++) yv[m] += rnd2;
            }
            unsigned sum = 0;
            const bool seg_ok = col_ok && s * N < hvalid;
#pragma unroll
            for (int m = 0; m < N; m++) {
                const int c = N >= 8 ? yv[m] >> 16 : wrap16(yv[m] >> sh2);
                yv[m] = c;
                // xQuant, src/pixel.cpp:609-616: unsigned 32-bit multiply-add, logical shift
                const int sg = c >> 31;  // 0 / -1
                const unsigned a = (unsigned)((c ^ sg) - sg);
                const unsigned lv = (__umul24(a, p.q_scale) + p.q_rnd) >> p.q_bits;
                sum += lv;
                const int sl = ((int)lv ^ sg) - sg;
                lvl[s * N + m] = (int16_t)sl;
            }
            if (seg_ok) {  // one predicate around the column's stores
#pragma unroll
                for (int m = 0; m < N; m++) lvp[(int64_t)(s * N + m) * d.level_stride] = lvl[s * N + m];
                if (cfp) {
#pragma unroll
                    for (int m = 0; m < N; m++) cfp[(int64_t)(s * N + m) * d.level_stride] = (int16_t)yv[m];
                }
            }
            if (d.d_abs_sum) {
#pragma unroll
                for (int o = 1; o < N; o <<= 1) sum += __shfl_xor(sum, o);
                if (seg_ok && (ln & (N - 1)) == 0)
                    d.d_abs_sum[(int64_t)f * d.abs_sum_frame_stride + ((y0 / N) + s) * tus_x + (x0 + ln) / N] = sum;
            }
        }
    } else {
        // inverse only: the lane's column of levels comes from the plane
        const bool col_ok = tile_ok && ln < wvalid;
        const int16_t *lvp = d.d_level + (int64_t)f * d.level_frame_stride + (int64_t)y0 * d.level_stride + x0 + ln;
#pragma unroll
        for (int m = 0; m < 32; m++) lvl[m] = (col_ok && m < hvalid) ? lvp[(int64_t)m * d.level_stride] : (int16_t)0;
    }

    if constexpr (MODE & TQ_INV) {
        // ================= dequantisation + vertical inverse (registers) =================
        const bool col_ok = tile_ok && ln < wvalid;
#pragma unroll
        for (int s = 0; s < NSEG; s++) {
            int in[N], xv[N];
#pragma unroll
            for (int m = 0; m < N; m++)  // xDeQuant, src/pixel.cpp:649-650
                in[m] = clip16((mul24((int)lvl[s * N + m], p.dq_scale) + p.dq_rnd) >> p.dq_shift);
            bool dst = false;
            if constexpr (N == 4) {
                if (d.d_modes && col_ok && s * 4 < hvalid)
                    dst = use_dst(4, d.d_modes[(int64_t)f * d.modes_frame_stride + ((y0 >> 2) + s) * tus_x + ((x0 + ln) >> 2)], false);
            }
            if constexpr (N >= 8) {
                inv_butterfly_pk<N, 1>(in, xv, 64 << 1);  // every dot2 operand is a clipped Int16 coefficient
#pragma unroll
                for (int k = 0; k < N; k += 2)  // SHIFT_INV_1ST = 7 (+1: scaled); store-then-clip => wraps (src/pixel.cpp:334-342)
                    *reinterpret_cast<v2s *>(&T[ln][s * N + k]) = pack_mid(xv[k], xv[k + 1]);
            } else {
                if (dst)
                    inv_dst4(in, xv);
                else
                    inv_butterfly<N>(in, xv);
#pragma unroll
                for (int k = 0; k < N; k += 2)  // SHIFT_INV_1ST = 7; store-then-clip => wraps (src/pixel.cpp:334-342)
                    *reinterpret_cast<v2s *>(&T[ln][s * N + k]) = pack2((xv[k] + 64) >> 7, (xv[k + 1] + 64) >> 7);
            }
        }
        __syncthreads();

        // ================= horizontal inverse + prediction + clip =================
        int v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = T[i][ln];
        uint32_t recw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) recw[i] = 0;
#pragma unroll
        for (int s = 0; s < NSEG; s++) {
            int xv[N];
            bool dst = false;
            if constexpr (N == 4) {
                if (d.d_modes && row_ok && s * 4 < wvalid)
                    dst = use_dst(4, d.d_modes[(int64_t)f * d.modes_frame_stride + ((y0 + ln) >> 2) * tus_x + (x0 >> 2) + s], true);
            }
            if constexpr (N >= 8) {
                inv_butterfly_pk<N, 4>(&v[s * N], xv, 2048 << 4);  // operands: the Int16 intermediate, raw
                // SHIFT_INV_2ND = 12 (+4: scaled, bits [31:16]), wrap (the pack truncates), then Clip(res + pred) (src/pixel.cpp:759) on
                // Int16 pairs: the saturating add cannot change a sum that is clipped to 0..255 anyway
#pragma unroll
                for (int k = 0; k < N; k += 4) {
                    const int i = s * N + k;
                    v2s r0 = __builtin_elementwise_add_sat(pack_hi(xv[k], xv[k + 1]), byte_pair(predw[i >> 2], 0));
                    v2s r1 = __builtin_elementwise_add_sat(pack_hi(xv[k + 2], xv[k + 3]), byte_pair(predw[i >> 2], 1));
                    r0 = __builtin_elementwise_min(__builtin_elementwise_max(r0, v2s{0, 0}), v2s{255, 255});
                    r1 = __builtin_elementwise_min(__builtin_elementwise_max(r1, v2s{0, 0}), v2s{255, 255});
                    recw[i >> 2] = __builtin_amdgcn_perm(as_u32(r1), as_u32(r0), 0x06040200u);
                }
            } else {
                if (dst)
                    inv_dst4(&v[s * N], xv);
                else
                    inv_butterfly<N>(&v[s * N], xv);
#pragma unroll
                for (int k = 0; k < N; k++) {  // SHIFT_INV_2ND = 12, wrap, then Clip(res + pred) (src/pixel.cpp:759)
                    const int i = s * N + k;
                    const int r = wrap16((xv[k] + 2048) >> 12) + (int)((predw[i >> 2] >> (8 * (i & 3))) & 255u);
                    const unsigned c = (unsigned)(r < 0 ? 0 : (r > 255 ? 255 : r));
                    recw[i >> 2] |= c << (8 * (i & 3));
                }
            }
        }
        if (row_ok) {
            uint8_t *rp = d.d_rec + (int64_t)f * d.rec_frame_stride + (int64_t)(y0 + ln) * d.rec_stride + x0;
            if (full_w) {
                U128 a, b;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    a.v[i] = recw[i];
                    b.v[i] = recw[4 + i];
                }
                *reinterpret_cast<U128 *>(rp) = a;
                *reinterpret_cast<U128 *>(rp + 16) = b;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (4 * i < wvalid) reinterpret_cast<U32 *>(rp + 4 * i)->v = recw[i];
            }
        }
    }
}

// ---- stand-alone quantiser stages (stage-by-stage checking) ----
__global__ void quant_plane_kernel(const int16_t *coef, int16_t *level, int stride, int width, int height,
                                   unsigned q_scale, int q_bits, unsigned q_rnd)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int c = coef[(int64_t)y * stride + x];
    const unsigned a = (unsigned)(c < 0 ? -c : c);
    const unsigned lv = (a * q_scale + q_rnd) >> q_bits;
    level[(int64_t)y * stride + x] = (int16_t)(c < 0 ? -(int)lv : (int)lv);
}

// xQuant's return value per TU.  lv < 2^15 for every Int16 input at every QP, so the
// truncating store never changes it and sum(lv) == sum(|level|).
__global__ void abs_sum_kernel(const int16_t *level, int stride, int tus_x, int ntus, int tu, uint32_t *out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntus) return;
    const int16_t *p = level + (int64_t)(t / tus_x) * tu * stride + (t % tus_x) * tu;
    unsigned s = 0;
    for (int y = 0; y < tu; y++)
        for (int x = 0; x < tu; x++) {
            const int v = p[(int64_t)y * stride + x];
            s += (unsigned)(v < 0 ? -v : v);
        }
    out[t] = s;
}

__global__ void dequant_plane_kernel(const int16_t *level, int16_t *coef, int stride, int width, int height,
                                     int scale, int shift, int rnd)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width || y >= height) return;
    const int v = ((int)level[(int64_t)y * stride + x] * scale + rnd) >> shift;
    coef[(int64_t)y * stride + x] = (int16_t)clip16(v);
}

static const unsigned kQuantScales[6] = {26214, 23302, 20560, 18396, 16384, 14564};  // src/table.cpp:156-158
static const int kInvQuantScales[6] = {40, 45, 51, 57, 64, 72};                      // src/table.cpp:161-163

static void quant_constants(int tu, int qp, int slice_type, TqParams &p)
{
    const int l2 = ilog2(tu);
    const int tshift = 15 - 8 - l2;             // MAX_TR_DYNAMIC_RANGE - 8 - log2N
    p.q_scale = kQuantScales[qp % 6];
    p.q_bits = 14 + qp / 6 + tshift;            // QUANT_SHIFT + qp/6 + tshift
    p.q_rnd = (unsigned)((slice_type == X265AMD_SLICE_I ? 171 : 85) << (p.q_bits - 9));
    p.dq_shift = 20 - 14 - tshift;              // QUANT_IQUANT_SHIFT - QUANT_SHIFT - tshift
    p.dq_rnd = 1 << (p.dq_shift - 1);
    p.dq_scale = kInvQuantScales[qp % 6] << (qp / 6);
    if (qp / 6 >= p.dq_shift) {
        // level * (IQ << qp/6) is a multiple of 2^shift then: the rounding term falls out under the floor, and
        // (level * scale + rnd) >> shift = level * (IQ << (qp/6 - shift)) exactly, either sign
        p.dq_scale = kInvQuantScales[qp % 6] << (qp / 6 - p.dq_shift);
        p.dq_rnd = 0;
        p.dq_shift = 0;
    }
}

// X265AMD_TQ_CHECK_MV: does every vector keep its block inside the prediction plane?  One thread per vector.
__global__ __launch_bounds__(256) void mv_check_kernel(const x265amd_mv *__restrict__ mv, int64_t frame_stride, int per_frame,
                                                       int ctus_x, int mv_ctu, int shift, int width, int height, int nframes,
                                                       uint32_t *__restrict__ verdict)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)per_frame * nframes) return;
    const int f = (int)(i / per_frame), k = (int)(i - (int64_t)f * per_frame);
    const x265amd_mv v = mv[(int64_t)f * frame_stride + k];
    const int x = (k % ctus_x) * mv_ctu + (v.mvx >> shift), y = (k / ctus_x) * mv_ctu + (v.mvy >> shift);
    if (x < 0 || y < 0 || x + mv_ctu > width || y + mv_ctu > height) atomicOr(verdict, 1u);
}

// validation + the kernels' parameter block; `empty`: nothing to launch (nframes = 0)
template <int MODE> static int tq_build(const x265amd_tq_desc *desc, bool coef_in, const TqDiag *diag, TqParams &p, bool &empty)
{
    empty = false;
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!desc) return X265AMD_ERR_ARG;
    const x265amd_tq_desc &d = *desc;
    if (d.tu != 4 && d.tu != 8 && d.tu != 16 && d.tu != 32) return X265AMD_ERR_ARG;  // the reference asserts (pixel.cpp:684)
    if (d.width <= 0 || d.height <= 0 || d.width % d.tu || d.height % d.tu) return X265AMD_ERR_ARG;
    if (d.qp < 0 || d.qp > 51 || d.nframes < 0) return X265AMD_ERR_ARG;
    if (d.slice_type < X265AMD_SLICE_I || d.slice_type > X265AMD_SLICE_B) return X265AMD_ERR_ARG;
    if (!d.d_pred || !d.d_level) return X265AMD_ERR_ARG;
    if ((MODE & TQ_FWD) && !d.d_src) return X265AMD_ERR_ARG;
    if ((MODE & TQ_INV) && !d.d_rec) return X265AMD_ERR_ARG;
    if (d.d_mv && (d.mv_ctu < 16 || (d.mv_ctu & (d.mv_ctu - 1)) || d.width % d.mv_ctu || d.height % d.mv_ctu ||
                   d.mv_ctu % d.tu))
        return X265AMD_ERR_ARG;
    if (d.mv_shift < 0 || d.mv_shift > 3) return X265AMD_ERR_ARG;
    if (d.flags & ~X265AMD_TQ_CHECK_MV) return X265AMD_ERR_ARG;
    // strides: a row must hold the picture (a zero-initialised stride would make every row alias); outputs of a
    // batch need their own planes.  With d_mv the prediction plane is addressed relative to d_pred in both
    // directions (vectors), so only its sign is checked.
    if (d.level_stride < d.width || d.pred_stride < (d.d_mv ? 1 : d.width)) return X265AMD_ERR_ARG;
    if ((MODE & TQ_FWD) && d.src_stride < d.width) return X265AMD_ERR_ARG;
    if ((MODE & TQ_INV) && d.rec_stride < d.width) return X265AMD_ERR_ARG;
    if (d.nframes > 1 && (d.level_frame_stride == 0 || ((MODE & TQ_INV) && d.rec_frame_stride == 0) ||
                          (d.d_abs_sum && (MODE & TQ_FWD) && d.abs_sum_frame_stride == 0)))
        return X265AMD_ERR_ARG;
    if ((int64_t)d.height * d.level_stride > 0x1fffffff || (int64_t)d.height * d.src_stride > 0x3fffffff ||
        (int64_t)d.height * d.rec_stride > 0x3fffffff || (int64_t)d.height * d.pred_stride > 0x3fffffff)
        return X265AMD_ERR_ARG;  // 32-bit byte offsets inside a plane, all below the out-of-range sentinel of tq_mfma.hip
    if (d.nframes == 0) {
        empty = true;
        return X265AMD_OK;
    }

    p.d = d;
    p.tiles_x = (d.width + 31) / 32;
    p.tiles_y = (d.height + 31) / 32;
    const int64_t nt = (int64_t)p.tiles_x * p.tiles_y * d.nframes;
    if (nt > 0x7fffffff) return X265AMD_ERR_ARG;
    p.ntiles = (int)nt;
    quant_constants(d.tu, d.qp, d.slice_type, p);
    p.mv_ctu_log2 = d.d_mv ? ilog2(d.mv_ctu) : 0;
    p.mv_ctus_x = d.d_mv ? d.width / d.mv_ctu : 0;
    p.d_trace = debug_state().tq_trace;
    p.trace_waves = debug_state().tq_trace_waves;
    p.diag_count = 0;
    p.diag_d = p.diag_lo = p.diag_cu_log2 = 0;
    p.diag_fgroup = p.diag_fmul = 0;
    p.diag_sum_pair = 0;
    if (diag && diag->count > 0) {
        p.diag_count = diag->count;
        p.diag_d = diag->d;
        p.diag_lo = diag->lo;
        p.diag_cu_log2 = diag->cu_log2;
        p.diag_fgroup = diag->fgroup;
        p.diag_fmul = diag->fmul;
        p.diag_sum_pair = diag->sum_pair;
    }
    if (coef_in) {  // the plane already holds xDeQuant's output: (c * 1 + 0) >> 0, and the clip is a no-op on Int16
        p.dq_scale = 1;
        p.dq_rnd = 0;
        p.dq_shift = 0;
    }
    return X265AMD_OK;
}

template <int MODE> static int launch_tq(const x265amd_tq_desc *desc, hipStream_t stream, bool coef_in = false, const TqDiag *diag = nullptr)
{
    TqParams p;
    bool empty;
    const int st = tq_build<MODE>(desc, coef_in, diag, p, empty);
    if (st != X265AMD_OK || empty) return st;
    const x265amd_tq_desc &d = p.d;
    if (d.d_mv && (d.flags & X265AMD_TQ_CHECK_MV)) {
        DeviceState *ds = device_state();
        if (!ds || !ds->host_word) return X265AMD_ERR_NOT_INIT;
        std::lock_guard<std::mutex> lk(state_mutex());  // one verdict word per device
        *ds->host_word = 0u;
        const int per_frame = (d.width / d.mv_ctu) * (d.height / d.mv_ctu);
        const int64_t nvec = (int64_t)per_frame * d.nframes;
        hipLaunchKernelGGL(mv_check_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, stream, d.d_mv, d.mv_frame_stride,
                           per_frame, d.width / d.mv_ctu, d.mv_ctu, d.mv_shift, d.width, d.height, d.nframes, ds->host_word_dev);
        X265AMD_HIP(hipGetLastError());
        X265AMD_HIP(hipStreamSynchronize(stream));
        if (*(volatile uint32_t *)ds->host_word) return X265AMD_ERR_ARG;  // a vector leaves the plane: nothing was transformed
    }
    const int64_t nt = p.ntiles;
    // the matrix-core formulation (tq_mfma.hip) wherever the transform matrix is the same for every TU of the
    // plane and the rows split into 16-byte pieces; per-TU DST selection at 4x4 (d_modes), widths that are not a
    // multiple of 16 and the debug switch keep the butterfly kernel
    // .. and so do planes whose strides or height do not fit the 24-bit multiplies of its address arithmetic
    if (tq_takes_mfma(desc)) return launch_tq_mfma(MODE, p, stream);
    if (p.diag_fgroup) return X265AMD_ERR_ARG;  // plane groups exist in the matrix-core kernel only (callers ask tq_takes_mfma first)
    const dim3 grid((unsigned)((nt + 7) / 8)), block(256);
    switch (d.tu) {
    case 4: hipLaunchKernelGGL((tq_kernel<4, MODE>), grid, block, 0, stream, p); break;
    case 8: hipLaunchKernelGGL((tq_kernel<8, MODE>), grid, block, 0, stream, p); break;
    case 16: hipLaunchKernelGGL((tq_kernel<16, MODE>), grid, block, 0, stream, p); break;
    default: hipLaunchKernelGGL((tq_kernel<32, MODE>), grid, block, 0, stream, p); break;
    }
    return hip_status(hipGetLastError());
}

bool tq_takes_mfma(const x265amd_tq_desc *desc)
{
    const x265amd_tq_desc &d = *desc;
    const int kLim = 1 << 23;
    const bool small = d.height < kLim && d.level_stride < kLim && d.pred_stride < kLim && d.src_stride < kLim &&
                       d.rec_stride < kLim;
    return debug_state().tq_path != 1 && !(d.tu == 4 && d.d_modes) && d.width % 16 == 0 && small;
}

int tq_diag_yuv_params(const x265amd_tq_desc *luma, const TqDiag &dl, const x265amd_tq_desc *chroma, const TqDiag &dc, TqParams &py,
                       TqParams &pc, int &ny_luma, int &ny_chroma)
{
    bool e1, e2;
    int st = tq_build<TQ_BOTH>(luma, false, &dl, py, e1);
    if (st == X265AMD_OK) st = tq_build<TQ_BOTH>(chroma, false, &dc, pc, e2);
    if (st != X265AMD_OK) return st;
    if (e1 || e2 || !tq_takes_mfma(luma) || !tq_takes_mfma(chroma)) return X265AMD_ERR_ARG;
    return tq_mfma_diag_yuv_check(py, pc, ny_luma, ny_chroma);
}

int tq_roundtrip_diag_yuv(const x265amd_tq_desc *luma, const TqDiag &dl, const x265amd_tq_desc *chroma, const TqDiag &dc,
                          hipStream_t stream)
{
    TqParams py, pc;
    int nyl = 0, nyc = 0;
    const int st = tq_diag_yuv_params(luma, dl, chroma, dc, py, pc, nyl, nyc);
    if (st != X265AMD_OK) return st;
    return launch_tq_mfma_diag_yuv(py, pc, stream);
}

int tq_roundtrip_diag(const x265amd_tq_desc *desc, const TqDiag &diag, hipStream_t stream)
{
    return launch_tq<TQ_BOTH>(desc, stream, false, &diag);
}

}  // namespace x265amd

using namespace x265amd;

extern "C" void x265amd_debug_tq_path(int path)
{
    if (device_state()) debug_state().tq_path = path;
}
extern "C" void x265amd_debug_tq_trace(unsigned long long *d_records, int nwaves)
{
    if (!device_state()) return;
    debug_state().tq_trace = d_records;
    debug_state().tq_trace_waves = d_records ? nwaves : 0;
}

extern "C" int x265amd_tq_forward(const x265amd_tq_desc *desc, void *stream)
{
    return launch_tq<TQ_FWD>(desc, (hipStream_t)stream);
}
extern "C" int x265amd_tq_inverse(const x265amd_tq_desc *desc, void *stream)
{
    return launch_tq<TQ_INV>(desc, (hipStream_t)stream);
}
extern "C" int x265amd_idct_add(const x265amd_tq_desc *desc, void *stream)
{
    return launch_tq<TQ_INV>(desc, (hipStream_t)stream, true);
}
extern "C" int x265amd_tq_roundtrip(const x265amd_tq_desc *desc, void *stream)
{
    return launch_tq<TQ_BOTH>(desc, (hipStream_t)stream);
}

extern "C" int x265amd_quant_plane(const int16_t *d_coef, int16_t *d_level, int stride, int width, int height,
                                   int tu, int qp, int slice_type, uint32_t *d_abs_sum, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_coef || !d_level || (tu != 4 && tu != 8 && tu != 16 && tu != 32)) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % tu || height % tu || qp < 0 || qp > 51 || stride < width) return X265AMD_ERR_ARG;
    if (slice_type < X265AMD_SLICE_I || slice_type > X265AMD_SLICE_B) return X265AMD_ERR_ARG;
    TqParams p;
    quant_constants(tu, qp, slice_type, p);
    hipLaunchKernelGGL(quant_plane_kernel, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream,
                       d_coef, d_level, stride, width, height, p.q_scale, p.q_bits, p.q_rnd);
    if (d_abs_sum) {
        const int ntus = (width / tu) * (height / tu);
        hipLaunchKernelGGL(abs_sum_kernel, dim3((ntus + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_level,
                           stride, width / tu, ntus, tu, d_abs_sum);
    }
    return hip_status(hipGetLastError());
}

extern "C" int x265amd_dequant_plane(const int16_t *d_level, int16_t *d_coef, int stride, int width, int height,
                                     int tu, int qp, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_coef || !d_level || (tu != 4 && tu != 8 && tu != 16 && tu != 32)) return X265AMD_ERR_ARG;
    if (width <= 0 || height <= 0 || width % tu || height % tu || qp < 0 || qp > 51 || stride < width) return X265AMD_ERR_ARG;
    TqParams p;
    quant_constants(tu, qp, X265AMD_SLICE_I, p);
    hipLaunchKernelGGL(dequant_plane_kernel, dim3((width + 255) / 256, height), dim3(256), 0, (hipStream_t)stream,
                       d_level, d_coef, stride, width, height, p.dq_scale, p.dq_shift, p.dq_rnd);
    return hip_status(hipGetLastError());
}
