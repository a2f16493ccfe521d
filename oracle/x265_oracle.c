/*
 * x265_oracle.c -- CPU restatement of the chenm001/x265 ME+TQ hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see x265_oracle.h).  Written fresh from SURVEY.md
 * App. A: the transforms are stated in table-driven matrix form (the reference
 * uses partial butterflies, which are an exact integer factorisation of the
 * same sums), with the reference's Int32 accumulators, arithmetic shifts and
 * truncating Int16 stores between passes.
 */
#include "x265_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* ROM (src/table.cpp:72-171).  The HEVC core transform is generated from    */
/* its 32 distinct magnitudes (first column of the 32x32 matrix):            */
/* T32[k][j] = sgn * c[fold(k*(2j+1) mod 128)], T_N[k][j] = T32[k*32/N][j].   */
/* ------------------------------------------------------------------------ */
static const int8_t kCos[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
static const int8_t kDst4[16] = {29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29};
static const int16_t kQuantScale[6] = {26214, 23302, 20560, 18396, 16384, 14564}; /* table.cpp:156 */
static const uint8_t kInvQuantScale[6] = {40, 45, 51, 57, 64, 72};               /* table.cpp:161 */

static int8_t g_t[4][32 * 32]; /* [log2n-2][k*n+j] */
static int g_t_ready = 0;

static int t32(int k, int j)
{
    int a = (k * (2 * j + 1)) & 127;
    if (a > 64) a = 128 - a;
    return a <= 32 ? kCos[a] : -kCos[64 - a];
}

static void build_tables(void)
{
    for (int l = 0; l < 4; l++) {
        int n = 4 << l;
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++) g_t[l][k * n + j] = (int8_t)t32(k * (32 / n), j);
    }
    g_t_ready = 1;
}

static int log2n(int n) /* xLog2(n-1), inc/utils.h:28-36 */
{
    int l = 0;
    unsigned x = (unsigned)(n - 1);
    while (x) {
        x >>= 1;
        l++;
    }
    return l;
}

const int8_t *orc_dct_matrix(int n)
{
    if (!g_t_ready) build_tables();
    return g_t[log2n(n) - 2];
}

const int8_t *orc_dst4_matrix(void) { return kDst4; }

int orc_chroma_qp(int qp)
{
    /* g_aucChromaScale (src/table.cpp:166-171): identity to 29, then the HEVC
     * draft chroma mapping, saturating at 39. */
    static const uint8_t tail[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36,
                                     36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
    return qp < 30 ? qp : tail[qp - 30];
}

static int clip3(int lo, int hi, int a) { return a < lo ? lo : (a > hi ? hi : a); }

/* ------------------------------------------------------------------------ */
/* SAD: src/pixel.cpp:30-93                                                  */
/* ------------------------------------------------------------------------ */
uint32_t orc_sad(int w, int n, const uint8_t *src, int ss, const uint8_t *ref, int sr)
{
    uint32_t sad = 0;
    for (int y = 0; y < n; y++) {
        const uint8_t *s = src + (size_t)y * ss, *r = ref + (size_t)y * sr;
        for (int x = 0; x < w; x++) sad += (uint32_t)abs((int)s[x] - (int)r[x]);
    }
    return sad;
}

/* ------------------------------------------------------------------------ */
/* SATD.  NOT in the reference: PARITY UNPINNED.  The   */
/* definition is the HEVC test model's (TComRdCost::xCalcHADs8x8 / 4x4),      */
/* restated as the matrix product it is: H[i][k] = (-1)^popcount(i & k);      */
/* 8x8 sub-blocks (sum |H D H^T| + 2) >> 2; 4x4 sub-blocks, used when the     */
/* block is 4 wide or 4 high, (sum |H D H^T| + 1) >> 1.                       */
/* ------------------------------------------------------------------------ */
static int had_sign(int i, int k) { return __builtin_popcount((unsigned)(i & k)) & 1 ? -1 : 1; }

static uint32_t satd_sub(int m, const uint8_t *s, int ss, const uint8_t *r, int sr)
{
    int d[8][8], t[8][8];
    for (int y = 0; y < m; y++)
        for (int x = 0; x < m; x++) d[y][x] = (int)s[(size_t)y * ss + x] - (int)r[(size_t)y * sr + x];
    for (int i = 0; i < m; i++) /* t = H d */
        for (int x = 0; x < m; x++) {
            int a = 0;
            for (int k = 0; k < m; k++) a += had_sign(i, k) * d[k][x];
            t[i][x] = a;
        }
    uint32_t sum = 0;
    for (int i = 0; i < m; i++) /* |t H^T| */
        for (int j = 0; j < m; j++) {
            int a = 0;
            for (int k = 0; k < m; k++) a += t[i][k] * had_sign(j, k);
            sum += (uint32_t)abs(a);
        }
    return m == 8 ? (sum + 2) >> 2 : (sum + 1) >> 1;
}

uint32_t orc_satd(int w, int n, const uint8_t *src, int ss, const uint8_t *ref, int sr)
{
    const int m = (w == 4 || n == 4) ? 4 : 8;
    uint32_t satd = 0;
    for (int y = 0; y < n; y += m)
        for (int x = 0; x < w; x += m) satd += satd_sub(m, src + (size_t)y * ss + x, ss, ref + (size_t)y * sr + x, sr);
    return satd;
}

static uint32_t sad64x64(const uint8_t *src, int ss, const uint8_t *ref, int sr)
{
    uint32_t sad = 0;
    for (int y = 0; y < 64; y++) {
        const uint8_t *s = src + (size_t)y * ss, *r = ref + (size_t)y * sr;
        uint32_t row = 0;
        for (int x = 0; x < 64; x++) row += (uint32_t)abs((int)s[x] - (int)r[x]);
        sad += row;
    }
    return sad;
}

/* ------------------------------------------------------------------------ */
/* 1-D transforms: src/pixel.cpp:114-312 (forward), 321-569 (inverse)        */
/* ------------------------------------------------------------------------ */
static const int8_t *matrix_for(int idx, int *n)
{
    if (idx == 0) {
        *n = 4;
        return kDst4;
    }
    *n = 2 << idx; /* 1->4 2->8 3->16 4->32 */
    return orc_dct_matrix(*n);
}

void orc_fwd_1d(int idx, int16_t *dst, const int16_t *src, unsigned stride, int lines, int shift)
{
    int n;
    const int8_t *m = matrix_for(idx, &n);
    const int32_t rnd = 1 << (shift - 1);
    if (idx == 0) lines = 4; /* xDST4 ignores nLines: src/pixel.cpp:138 */
    for (int i = 0; i < lines; i++)
        for (int k = 0; k < n; k++) {
            int32_t acc = 0;
            for (int j = 0; j < n; j++) acc += (int32_t)m[k * n + j] * (int32_t)src[i * stride + j];
            dst[k * stride + i] = (int16_t)((acc + rnd) >> shift); /* truncating store */
        }
}

void orc_inv_1d(int idx, int16_t *dst, const int16_t *src, unsigned stride, int lines, int shift)
{
    int n;
    const int8_t *m = matrix_for(idx, &n);
    const int32_t rnd = 1 << (shift - 1);
    if (idx == 0) lines = 4;  /* xInvDST4 ignores nLines: src/pixel.cpp:351 */
    if (idx == 4) lines = 32; /* xInvDCT32 ignores nLines: src/pixel.cpp:513 */
    for (int i = 0; i < lines; i++)
        for (int j = 0; j < n; j++) {
            int32_t acc = 0;
            for (int k = 0; k < n; k++) acc += (int32_t)m[k * n + j] * (int32_t)src[k * stride + i];
            /* stored to Int16 first, then Clip3 on the stored value => wraps
             * (src/pixel.cpp:334-342 and SURVEY.md App. C) */
            int16_t v = (int16_t)((acc + rnd) >> shift);
            dst[i * stride + j] = (int16_t)clip3(-32768, 32767, v);
        }
}

/* ------------------------------------------------------------------------ */
/* xSubDct: src/pixel.cpp:668-722                                            */
/* ------------------------------------------------------------------------ */
void orc_sub_dct(int16_t *dst, const uint8_t *src, const uint8_t *ref, unsigned stride,
                 int16_t *tmp0, int16_t *tmp1, int w, int h, unsigned mode)
{
    const int lw = log2n(w), lh = log2n(h);
    const int use_dst = (w + h == 8) && (mode != ORC_MODE_INVALID);
    const int dst_hor = use_dst && (!mode || (mode >= 2 && mode <= 25));  /* :679 */
    const int dst_ver = use_dst && (!mode || (mode >= 11 && mode <= 34)); /* :680 */
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            tmp0[i * stride + j] = (int16_t)((int)src[i * stride + j] - (int)ref[i * stride + j]);
    orc_fwd_1d(lw - 1 - dst_hor, tmp1, tmp0, stride, h, lw - 1); /* :706 */
    orc_fwd_1d(lh - 1 - dst_ver, dst, tmp1, stride, w, lh + 6);  /* :707 */
}

/* ------------------------------------------------------------------------ */
/* xIDctAdd: src/pixel.cpp:724-776                                           */
/* ------------------------------------------------------------------------ */
void orc_idct_add(uint8_t *dst, const int16_t *src, const uint8_t *ref, unsigned stride,
                  int16_t *tmp0, int16_t *tmp1, int w, int h, unsigned mode)
{
    const int lw = log2n(w), lh = log2n(h);
    const int use_dst = (w + h == 8) && (mode != ORC_MODE_INVALID);
    const int dst_hor = use_dst && (!mode || (mode >= 11 && mode <= 34)); /* :735 (swapped) */
    const int dst_ver = use_dst && (!mode || (mode >= 2 && mode <= 25));  /* :736 */
    orc_inv_1d(lw - 1 - dst_hor, tmp0, src, stride, h, 7);   /* SHIFT_INV_1ST */
    orc_inv_1d(lh - 1 - dst_ver, tmp1, tmp0, stride, w, 12); /* SHIFT_INV_2ND */
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            dst[i * stride + j] =
                (uint8_t)clip3(0, 255, (int)tmp1[i * stride + j] + (int)ref[i * stride + j]);
}

/* ------------------------------------------------------------------------ */
/* xQuant: src/pixel.cpp:591-633                                             */
/* ------------------------------------------------------------------------ */
uint32_t orc_quant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h,
                   int slice_type)
{
    const unsigned q = (unsigned)kQuantScale[qp % 6];
    const int tshift = 15 - 8 - log2n(w);
    const int qbits = 14 + (int)(qp / 6) + tshift;
    const int32_t rnd = (slice_type == ORC_SLICE_I ? 171 : 85) << (qbits - 9);
    uint32_t sum = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int c = src[y * stride + x];
            int sign = c < 0 ? -1 : 1;
            /* unsigned 32-bit multiply-add, logical shift (uiQ is UInt) */
            int lvl = (int)(((unsigned)abs(c) * q + (unsigned)rnd) >> qbits);
            sum += (uint32_t)lvl;
            lvl *= sign;
            int16_t st = (int16_t)lvl; /* truncating store, then a no-op clip */
            dst[y * stride + x] = (int16_t)clip3(-32768, 32767, st);
        }
    return sum;
}

/* ------------------------------------------------------------------------ */
/* xDeQuant: src/pixel.cpp:635-666                                           */
/* ------------------------------------------------------------------------ */
void orc_dequant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h,
                 int slice_type)
{
    (void)slice_type;
    const int tshift = 15 - 8 - log2n(w);
    const int shift = 20 - 14 - tshift;
    const int32_t rnd = 1 << (shift - 1);
    const int scale = (int)kInvQuantScale[qp % 6] << (qp / 6);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int32_t v = ((int32_t)src[y * stride + x] * scale + rnd) >> shift;
            dst[y * stride + x] = (int16_t)clip3(-32768, 32767, v);
        }
}

/* ------------------------------------------------------------------------ */
/* Motion search: SURVEY.md App. A.7; argmin idiom src/encode.cpp:725-727,   */
/* 762-765; MAX_SAD inc/config.h:52.                                         */
/* ------------------------------------------------------------------------ */
static void me_one_ctu(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                       int ctu, int range, int cx, int cy, orc_mv *out)
{
    uint32_t best = ORC_MAX_SAD;
    int bx = 0, by = 0;
    const uint8_t *c = cur + (size_t)cy * stride + cx;
    for (int dy = -range; dy <= range; dy++) {
        int ry = cy + dy;
        if (ry < 0 || ry + ctu > height) continue;
        for (int dx = -range; dx <= range; dx++) {
            int rx = cx + dx;
            if (rx < 0 || rx + ctu > width) continue;
            const uint8_t *r = ref + (size_t)ry * stride + rx;
            uint32_t cost = (ctu == 64) ? sad64x64(c, stride, r, stride)
                                        : orc_sad(ctu, ctu, c, stride, r, stride);
            if (cost < best) {
                best = cost;
                bx = dx;
                by = dy;
            }
        }
    }
    out->mvx = (int16_t)bx;
    out->mvy = (int16_t)by;
    out->sad = best;
}

void orc_me_fullsearch(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                       int ctu, int range, orc_mv *out, int first_ctu, int step, int count,
                       int nthreads)
{
    const int ncx = width / ctu, ncy = height / ctu, nctu = ncx * ncy;
    if (step < 1) step = 1;
    if (count < 0) count = (nctu - first_ctu + step - 1) / step;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int i = 0; i < count; i++) {
        int idx = first_ctu + i * step;
        if (idx >= nctu) continue;
        me_one_ctu(cur, ref, width, height, stride, ctu, range, (idx % ncx) * ctu,
                   (idx / ncx) * ctu, &out[idx]);
    }
}

/* The same search for w x h blocks tiling the picture (the inter PU shapes: 2NxN, Nx2N and the squares down
 * to 8x8).  The reference's SAD primitives take the number of rows as an argument, xSadN[log2(w) - 2](h, ..)
 * (src/pixel.cpp:30-101), so a rectangular PU is the same call with h != w; argmin idiom as above. */
void orc_me_fullsearch_wh(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                          int bw, int bh, int range, orc_mv *out, int first, int step, int count,
                          int nthreads)
{
    const int nbx = width / bw, nby = height / bh, nb = nbx * nby;
    if (step < 1) step = 1;
    if (count < 0) count = (nb - first + step - 1) / step;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
    for (int i = 0; i < count; i++) {
        int idx = first + i * step;
        if (idx >= nb) continue;
        const int bx0 = (idx % nbx) * bw, by0 = (idx / nbx) * bh;
        const uint8_t *c = cur + (size_t)by0 * stride + bx0;
        uint32_t best = ORC_MAX_SAD;
        int bx = 0, by = 0;
        for (int dy = -range; dy <= range; dy++) {
            if (by0 + dy < 0 || by0 + dy + bh > height) continue;
            for (int dx = -range; dx <= range; dx++) {
                if (bx0 + dx < 0 || bx0 + dx + bw > width) continue;
                uint32_t cost = orc_sad(bw, bh, c, stride, ref + (size_t)(by0 + dy) * stride + bx0 + dx, stride);
                if (cost < best) {
                    best = cost;
                    bx = dx;
                    by = dy;
                }
            }
        }
        out[idx].mvx = (int16_t)bx;
        out[idx].mvy = (int16_t)by;
        out[idx].sad = best;
    }
}

/* ------------------------------------------------------------------------ */
/* Frame-level TQ.  Each TU is copied into 32-stride scratch exactly like    */
/* the reference's per-CU cache (inc/x265.h:105-131) and run through the     */
/* per-block restatements above.                                             */
/* ------------------------------------------------------------------------ */
static void fetch_pred(uint8_t *dst32, const uint8_t *pred, int pstride, const orc_mv *mv,
                       int mv_ctu, int ncx, int x0, int y0, int n)
{
    int ox = 0, oy = 0;
    if (mv) {
        const orc_mv *m = &mv[(y0 / mv_ctu) * ncx + (x0 / mv_ctu)];
        ox = m->mvx;
        oy = m->mvy;
    }
    for (int i = 0; i < n; i++)
        memcpy(dst32 + i * 32, pred + (size_t)(y0 + oy + i) * pstride + x0 + ox, (size_t)n);
}

void orc_tq_forward(const uint8_t *src, int sstride, const uint8_t *pred, int pstride,
                    const orc_mv *mv, int mv_ctu, int width, int height, int n, int qp,
                    int slice_type, const uint8_t *modes, int16_t *level, int cstride,
                    uint32_t *abs_sum, int nthreads)
{
    const int ntx = width / n, nty = height / n;
    const int ncx = mv ? width / mv_ctu : 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int ty = 0; ty < nty; ty++) {
        uint8_t s32[32 * 32], p32[32 * 32];
        int16_t t0[32 * 32], t1[32 * 32], q[32 * 32];
        for (int tx = 0; tx < ntx; tx++) {
            const int x0 = tx * n, y0 = ty * n;
            for (int i = 0; i < n; i++)
                memcpy(s32 + i * 32, src + (size_t)(y0 + i) * sstride + x0, (size_t)n);
            fetch_pred(p32, pred, pstride, mv, mv_ctu, ncx, x0, y0, n);
            unsigned mode = modes ? modes[ty * ntx + tx] : ORC_MODE_INVALID;
            orc_sub_dct(t0, s32, p32, 32, t0, t1, n, n, mode); /* in place, encode.cpp:868 */
            uint32_t s = orc_quant(q, t0, 32, (unsigned)qp, n, n, slice_type);
            if (abs_sum) abs_sum[ty * ntx + tx] = s;
            for (int i = 0; i < n; i++)
                memcpy(level + (size_t)(y0 + i) * cstride + x0, q + i * 32, (size_t)n * 2);
        }
    }
}

void orc_tq_inverse(const int16_t *level, int cstride, const uint8_t *pred, int pstride,
                    const orc_mv *mv, int mv_ctu, int width, int height, int n, int qp,
                    int slice_type, const uint8_t *modes, uint8_t *rec, int rstride, int nthreads)
{
    const int ntx = width / n, nty = height / n;
    const int ncx = mv ? width / mv_ctu : 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int ty = 0; ty < nty; ty++) {
        uint8_t p32[32 * 32], r32[32 * 32];
        int16_t t0[32 * 32], t1[32 * 32], q[32 * 32];
        for (int tx = 0; tx < ntx; tx++) {
            const int x0 = tx * n, y0 = ty * n;
            uint32_t sum = 0;
            for (int i = 0; i < n; i++) {
                memcpy(q + i * 32, level + (size_t)(y0 + i) * cstride + x0, (size_t)n * 2);
                for (int j = 0; j < n; j++) sum += (uint32_t)abs((int)q[i * 32 + j]);
            }
            fetch_pred(p32, pred, pstride, mv, mv_ctu, ncx, x0, y0, n);
            unsigned mode = modes ? modes[ty * ntx + tx] : ORC_MODE_INVALID;
            if (sum) { /* encode.cpp:893-900 */
                orc_dequant(t0, q, 32, (unsigned)qp, n, n, slice_type);
                orc_idct_add(r32, t0, p32, 32, t1, t0, n, n, mode);
            } else { /* encode.cpp:901-905 */
                for (int i = 0; i < n; i++) memcpy(r32 + i * 32, p32 + i * 32, (size_t)n);
            }
            for (int i = 0; i < n; i++)
                memcpy(rec + (size_t)(y0 + i) * rstride + x0, r32 + i * 32, (size_t)n);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* Intra luma prediction and the 35-mode decision.     */
/* ref[4n + 1] is the reference's sample array (src/encode.cpp:1193):        */
/* ref[0 .. 2n-1] the left column from the bottom up (ref[2n-1] beside row   */
/* 0), ref[2n] the corner, ref[2n+1 .. 4n] the row above, left to right.     */
/* ------------------------------------------------------------------------ */
/* src/table.cpp:32-39 (which modes read the smoothed samples), :49-56, :59-66 */
static const uint8_t kIntraFilter[5][35] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};
static const int8_t kIntraAngle[35] = {0,   0,   32,  26,  21,  17,  13,  9,  5,  2,  0,  -2, -5, -9, -13, -17, -21, -26,
                                       -32, -26, -21, -17, -13, -9,  -5,  -2, 0,  2,  5,  9,  13, 17, 21,  26,  32};
static const int16_t kIntraInvAngle[35] = {0,   0,   256, 315, 390, 482, 630, 910, 1638, 4096, 0,   4096, 1638, 910, 630, 482, 390, 315,
                                           256, 315, 390, 482, 630, 910, 1638, 4096, 0,  4096, 1638, 910, 630, 482, 390, 315, 256};

/* the [1 2 1] smoothing of the sample array, ends kept (src/encode.cpp:1260-1265) */
void orc_intra_filter(const uint8_t *ref0, uint8_t *ref1, int n)
{
    ref1[0] = ref0[0];
    ref1[4 * n] = ref0[4 * n];
    for (int i = 1; i < 4 * n; i++) ref1[i] = (uint8_t)((ref0[i - 1] + 2 * ref0[i] + ref0[i + 1] + 2) >> 2);
}

static void intra_planar(const uint8_t *ref, uint8_t *dst, int ds, int n) /* src/encode.cpp:1353-1396 */
{
    const int lg = log2n(n);
    const uint8_t *left = ref + 2 * n - 1, *top = ref + 2 * n + 1;
    int16_t lc[65], tr[65], br[65], rc[65];
    for (int i = 0; i <= n; i++) {
        tr[i] = top[i];
        lc[i] = left[-i];
    }
    const int bl = left[-n], trt = top[n];
    for (int i = 0; i < n; i++) {
        br[i] = (int16_t)(bl - tr[i]);
        rc[i] = (int16_t)(trt - lc[i]);
        tr[i] = (int16_t)(tr[i] << lg);
        lc[i] = (int16_t)(lc[i] << lg);
    }
    for (int i = 0; i < n; i++) {
        int16_t hp = (int16_t)(lc[i] + n);
        for (int j = 0; j < n; j++) {
            hp = (int16_t)(hp + rc[i]);
            tr[j] = (int16_t)(tr[j] + br[j]);
            dst[i * ds + j] = (uint8_t)((hp + tr[j]) >> (lg + 1));
        }
    }
}

static void intra_dc(const uint8_t *ref, uint8_t *dst, int ds, int n) /* src/encode.cpp:1398-1444, luma */
{
    const uint8_t *left = ref + 2 * n - 1, *top = ref + 2 * n + 1;
    uint32_t st = 0, sl = 0;
    for (int i = 0; i < n; i++) {
        st += top[i];
        sl += left[-i];
    }
    const uint8_t dc = (uint8_t)((st + sl + n) / (2 * n));
    for (int i = 0; i < n; i++) memset(dst + i * ds, dc, n);
    dst[0] = (uint8_t)((top[0] + left[0] + 2 * dst[0] + 2) >> 2);
    for (int i = 1; i < n; i++) {
        dst[i] = (uint8_t)((top[i] + 3 * dst[i] + 2) >> 2);
        dst[i * ds] = (uint8_t)((left[-i] + 3 * dst[i * ds] + 2) >> 2);
    }
}

static void intra_ang(const uint8_t *ref, uint8_t *dst, int ds, int n, int mode, int luma) /* src/encode.cpp:1446-1534 */
{
    const int hor = mode < 18, ang = kIntraAngle[mode], inv = kIntraInvAngle[mode];
    const uint8_t *tl = ref + 2 * n;
    uint8_t buf[2 * 64 + 1 + 64];
    uint8_t *rm = buf + (ang < 0 ? 64 : 0);
    for (int x = 0; x <= n; x++) rm[x] = hor ? tl[-x] : tl[x];
    if (ang < 0) {
        int sum = 128;
        for (int x = -1; x > ((n * ang) >> 5); x--) {
            sum += inv;
            rm[x] = tl[hor ? (sum >> 8) : -(sum >> 8)];
        }
    } else {
        for (int x = n + 1; x <= 2 * n; x++) rm[x] = hor ? tl[-x] : tl[x];
    }
    int pos = 0;
    for (int k = 0; k < n; k++) {
        pos += ang;
        const int idx = pos >> 5, fact = pos & 31;
        for (int x = 0; x < n; x++)
            dst[k * ds + x] = fact ? (uint8_t)(((32 - fact) * rm[x + idx + 1] + fact * rm[x + idx + 2] + 16) >> 5) : rm[x + idx + 1];
    }
    if (luma && ang == 0) {
        const int o = hor ? 1 : -1;
        for (int x = 0; x < n; x++) {
            int v = dst[x * ds] + ((tl[(x + 1) * o] - tl[0]) >> 1);
            dst[x * ds] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    if (hor)
        for (int k = 0; k < n - 1; k++)
            for (int x = k + 1; x < n; x++) {
                uint8_t t = dst[k * ds + x];
                dst[k * ds + x] = dst[x * ds + k];
                dst[x * ds + k] = t;
            }
}

/* xEncIntraPredLuma (src/encode.cpp:1617-1652): ref0 = the samples, ref1 = smoothed */
void orc_intra_pred_luma(const uint8_t *ref0, const uint8_t *ref1, int n, int mode, uint8_t *dst, int ds)
{
    const uint8_t *ref = kIntraFilter[log2n(n) - 2][mode] ? ref1 : ref0;
    if (mode == 0)
        intra_planar(ref, dst, ds, n);
    else if (mode == 1)
        intra_dc(ref, dst, ds, n);
    else
        intra_ang(ref, dst, ds, n, mode, 1);
}

/* The luma mode decision of one CU (src/encode.cpp:725-765): cost = SAD + lambda * (1 for the first most
 * probable mode, 2 for the other two, 3 otherwise), first strict minimum in mode order.  sads[35] receives
 * the plain SADs, best_pred (n*n, stride n; may be NULL) the winner's prediction. */
void orc_intra_decide_luma(const uint8_t *src, int ss, const uint8_t *ref0, const uint8_t *mpm, uint32_t lambda,
                           int n, uint32_t *sads, uint32_t *best_mode, uint32_t *best_cost, uint8_t *best_pred)
{
    uint8_t ref1[4 * 64 + 1], pred[64 * 64];
    orc_intra_filter(ref0, ref1, n);
    uint32_t best = ORC_MAX_SAD, bm = 0;
    for (int m = 0; m < 35; m++) {
        orc_intra_pred_luma(ref0, ref1, n, m, pred, n);
        const uint32_t sad = orc_sad(n, n, src, ss, pred, n);
        const uint32_t cost = sad + lambda * (m == mpm[0] ? 1u : (m == mpm[1] || m == mpm[2]) ? 2u : 3u);
        if (sads) sads[m] = sad;
        if (cost < best) {
            best = cost;
            bm = (uint32_t)m;
        }
    }
    *best_mode = bm;
    *best_cost = best;
    if (best_pred) orc_intra_pred_luma(ref0, ref1, n, (int)bm, best_pred, n);
}

/* ------------------------------------------------------------------------ */
/* One intra frame: the pixel path of the encode loop (src/encode.cpp:668-941 */
/* without the entropy coder), CU = PU = TU, 4:2:0.                           */
/* ------------------------------------------------------------------------ */
/* xEncIntraLoadRef + xPaddingRef (src/encode.cpp:1133-1266) read from the reconstructed planes instead of the
 * line buffers (same pixels: the buffers hold the row above and the column to the left of the CU). */
static void intra_load_ref(const uint8_t *rec, int stride, int x0, int y0, int n, int bL, int bT, int bTR, uint8_t *ref)
{
    const int valid[5] = {0, bL, bL && bT, bT, bTR};
    const int off[6] = {0, n, 2 * n, 2 * n + 1, 3 * n + 1, 4 * n + 1};
    if (!(bL || bT || bTR)) {
        memset(ref, 0x80, 4 * n + 1);
        return;
    }
    if (bL)
        for (int i = 0; i < n; i++) ref[n + i] = rec[(size_t)(y0 + n - 1 - i) * stride + x0 - 1];
    if (bL && bT) ref[2 * n] = rec[(size_t)(y0 - 1) * stride + x0 - 1];
    if (bT)
        for (int i = 0; i < n; i++) ref[2 * n + 1 + i] = rec[(size_t)(y0 - 1) * stride + x0 + i];
    if (bTR)
        for (int i = 0; i < n; i++) ref[3 * n + 1 + i] = rec[(size_t)(y0 - 1) * stride + x0 + n + i];
    int b = 0;
    while (!valid[b]) b++;
    for (int i = 0; i < off[b]; i++) ref[i] = ref[off[b]];
    for (; b < 5; b++)
        if (!valid[b])
            for (int i = off[b]; i < off[b + 1]; i++) ref[i] = ref[off[b] - 1];
}

static void intra_pred_chroma(const uint8_t *ref, int n, int mode, uint8_t *dst) /* src/encode.cpp:1654-1706 */
{
    if (mode == 0)
        intra_planar(ref, dst, n, n);
    else if (mode == 1) {
        const uint8_t *left = ref + 2 * n - 1, *top = ref + 2 * n + 1;
        uint32_t s = 0;
        for (int i = 0; i < n; i++) s += top[i] + left[-i];
        memset(dst, (int)((s + n) / (2 * n)), (size_t)n * n);
    } else {
        intra_ang(ref, dst, n, n, mode, 0); /* bLuma FALSE: no gradient filter on the zero-angle modes */
    }
}

/* yuv: one frame, planes Y (w x h), U, V (w/2 x h/2), strides = widths.  rec: the same layout.  mode_y, mode_c:
 * per CU in raster order (mode_c = the real chroma mode).  levels (may be NULL): int16, same layout as yuv. */
void orc_intra_encode_frame(const uint8_t *yuv, int w, int h, int cu, int qp, uint8_t *rec, uint8_t *mode_y,
                            uint8_t *mode_c, int16_t *levels)
{
    const int wc = w / 2, hc = h / 2, nc = cu / 2, ncx = w / cu, ncy = h / cu;
    const uint8_t *sy = yuv, *su = yuv + (size_t)w * h, *sv = su + (size_t)wc * hc;
    uint8_t *ry = rec, *ru = rec + (size_t)w * h, *rv = ru + (size_t)wc * hc;
    const int qpc = orc_chroma_qp(qp);
    int16_t t0[32 * 32], t1[32 * 32], q[32 * 32];
    uint8_t s32[32 * 32], p32[32 * 32], r32[32 * 32];
    for (int cy = 0; cy < ncy; cy++)
        for (int cx = 0; cx < ncx; cx++) {
            const int x0 = cx * cu, y0 = cy * cu, idx = cy * ncx + cx;
            const int bL = cx > 0, bT = cy > 0, bTR = bT && cx + 1 < ncx;
            uint8_t ref0[129], refu[65], refv[65], mpm[3], pred[32 * 32], predu[16 * 16], predv[16 * 16];
            intra_load_ref(ry, w, x0, y0, cu, bL, bT, bTR, ref0);
            intra_load_ref(ru, wc, x0 / 2, y0 / 2, nc, bL, bT, bTR, refu);
            intra_load_ref(rv, wc, x0 / 2, y0 / 2, nc, bL, bT, bTR, refv);
            /* most probable modes, src/encode.cpp:1268-1293 (the mode above is DC there: `bT && uiY`, uiY == 0) */
            const int left = bL ? mode_y[idx - 1] : 1, top = 1;
            if (left == top) {
                if (left > 1)
                    mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)(((left + 29) % 32) + 2), mpm[2] = (uint8_t)(((left - 1) % 32) + 2);
                else
                    mpm[0] = 0, mpm[1] = 1, mpm[2] = 26;
            } else {
                mpm[0] = (uint8_t)left, mpm[1] = (uint8_t)top;
                mpm[2] = (left != 0 && top != 0) ? 0 : (left + top) < 2 ? 26 : 1;
            }
            uint32_t bm, bc;
            orc_intra_decide_luma(sy + (size_t)y0 * w + x0, w, ref0, mpm, (uint32_t)qp, cu, NULL, &bm, &bc, pred);
            mode_y[idx] = (uint8_t)bm;
            /* chroma: src/encode.cpp:781-853 */
            int cand[6] = {0, 26, 10, 1, -1, (int)bm};
            for (int i = 0; i < 4; i++)
                if (cand[i] == (int)bm) {
                    cand[i] = 34;
                    break;
                }
            uint32_t bestc = ORC_MAX_SAD;
            int bk = 0;
            for (int k = 0; k < 6; k++) {
                if (k == 4) continue; /* LM_CHROMA is skipped */
                intra_pred_chroma(refu, nc, cand[k], predu);
                intra_pred_chroma(refv, nc, cand[k], predv);
                const uint32_t s = orc_sad(nc, nc, su + (size_t)(y0 / 2) * wc + x0 / 2, wc, predu, nc) +
                                   orc_sad(nc, nc, sv + (size_t)(y0 / 2) * wc + x0 / 2, wc, predv, nc);
                if (s < bestc) {
                    bestc = s;
                    bk = k;
                }
            }
            const int mc = cand[bk];
            mode_c[idx] = (uint8_t)mc;
            intra_pred_chroma(refu, nc, mc, predu);
            intra_pred_chroma(refv, nc, mc, predv);
            /* transform stage, src/encode.cpp:863-925, per plane */
            for (int pl = 0; pl < 3; pl++) {
                const int n = pl ? nc : cu, st = pl ? wc : w, px = pl ? x0 / 2 : x0, py = pl ? y0 / 2 : y0;
                const uint8_t *src = pl == 0 ? sy : pl == 1 ? su : sv;
                uint8_t *dst = pl == 0 ? ry : pl == 1 ? ru : rv;
                const uint8_t *pp = pl == 0 ? pred : pl == 1 ? predu : predv;
                for (int i = 0; i < n; i++) {
                    memcpy(s32 + i * 32, src + (size_t)(py + i) * st + px, n);
                    memcpy(p32 + i * 32, pp + i * n, n);
                }
                const unsigned mode = pl ? (unsigned)mc : bm;
                orc_sub_dct(t0, s32, p32, 32, t0, t1, n, n, mode);
                const uint32_t sum = orc_quant(q, t0, 32, (unsigned)(pl ? qpc : qp), n, n, ORC_SLICE_I);
                if (sum) {
                    orc_dequant(t0, q, 32, (unsigned)(pl ? qpc : qp), n, n, ORC_SLICE_I);
                    orc_idct_add(r32, t0, p32, 32, t1, t0, n, n, mode);
                } else {
                    for (int i = 0; i < n; i++) memcpy(r32 + i * 32, p32 + i * 32, n);
                }
                for (int i = 0; i < n; i++) memcpy(dst + (size_t)(py + i) * st + px, r32 + i * 32, n);
                if (levels) {
                    int16_t *lv = levels + (pl == 0 ? 0 : pl == 1 ? (size_t)w * h : (size_t)w * h + (size_t)wc * hc);
                    for (int i = 0; i < n; i++) memcpy(lv + (size_t)(py + i) * st + px, q + i * 32, (size_t)n * 2);
                }
            }
        }
}
