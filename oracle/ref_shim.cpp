/*
 * ref_shim.cpp -- extern "C" doorway into the REAL reference, compiled where it
 * lies under /root/reference (see oracle/Makefile target `ref`).  The reference
 * has C++ linkage and no main(); this file only forwards calls so that tests
 * and the golden-vector generator can reach it through ctypes.
 *
 * TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/ (git-ignored).  No
 * reference source is copied: this file #includes the reference headers from
 * /root/reference/inc at build time and holds loops of our own only where the
 * reference has none (motion search: SURVEY.md App. A.7 over the reference's
 * own xSadN[]; frame-level TQ: the per-CU call sequence of
 * src/encode.cpp:868-925 over a TU grid).
 */
#include "x265.h"
#include "md5.h"

#include <string.h>

void xWriteCU(X265_t *h, UInt nDepth, UInt bLastCU); /* src/encode.cpp:474: defined there, not declared in the header */

extern "C" {

struct ref_mv {
    int16_t mvx, mvy;
    uint32_t sad;
};

const int8_t *ref_dct_matrix(int n)
{
    switch (n) {
    case 4: return (const int8_t *)g_aiT4;
    case 8: return (const int8_t *)g_aiT8;
    case 16: return (const int8_t *)g_aiT16;
    default: return (const int8_t *)g_aiT32;
    }
}
int ref_chroma_qp(int qp) { return g_aucChromaScale[qp]; }
int ref_quant_scale(int i) { return g_quantScales[i]; }
int ref_inv_quant_scale(int i) { return g_invQuantScales[i]; }

uint32_t ref_sad(int w, int n, const uint8_t *src, int ss, const uint8_t *ref, int sr)
{
    return xSadN[xLog2(w - 1) - 2](n, src, ss, ref, sr);
}
void ref_fwd_1d(int idx, int16_t *dst, int16_t *src, unsigned stride, int lines, int shift)
{
    xDctN[idx](dst, src, stride, lines, shift);
}
void ref_inv_1d(int idx, int16_t *dst, int16_t *src, unsigned stride, int lines, int shift)
{
    xInvDctN[idx](dst, src, stride, lines, shift);
}
void ref_sub_dct(int16_t *dst, uint8_t *src, uint8_t *ref, unsigned stride, int16_t *t0,
                 int16_t *t1, int w, int h, unsigned mode)
{
    xSubDct(dst, src, ref, stride, t0, t1, w, h, mode);
}
void ref_idct_add(uint8_t *dst, int16_t *src, uint8_t *ref, unsigned stride, int16_t *t0,
                  int16_t *t1, int w, int h, unsigned mode)
{
    xIDctAdd(dst, src, ref, stride, t0, t1, w, h, mode);
}
uint32_t ref_quant(int16_t *dst, int16_t *src, unsigned stride, unsigned qp, int w, int h, int st)
{
    return xQuant(dst, src, stride, qp, w, h, (X265_SliceType)st);
}
void ref_dequant(int16_t *dst, int16_t *src, unsigned stride, unsigned qp, int w, int h, int st)
{
    xDeQuant(dst, src, stride, qp, w, h, (X265_SliceType)st);
}

/* The reference's plug-in points are three MUTABLE global tables (src/pixel.cpp:95,571,579).  Tests point
 * them at other implementations with the same signatures -- the GPU drop-ins of
 * include/x265_amd_dropin.h -- and then run the reference's own encoder over them.
 * get: copy the current 5+5+5 pointers out; set: NULL arrays / NULL entries keep the current ones. */
void ref_tables_get(void **sad5, void **dct5, void **inv5)
{
    for (int i = 0; i < 5; i++) {
        sad5[i] = (void *)xSadN[i];
        dct5[i] = (void *)xDctN[i];
        inv5[i] = (void *)xInvDctN[i];
    }
}
void ref_tables_set(void *const *sad5, void *const *dct5, void *const *inv5)
{
    for (int i = 0; i < 5; i++) {
        if (sad5 && sad5[i]) xSadN[i] = (xSad *)sad5[i];
        if (dct5 && dct5[i]) xDctN[i] = (xDCT *)dct5[i];
        if (inv5 && inv5[i]) xInvDctN[i] = (xDCT *)inv5[i];
    }
}

/* Exhaustive search over the reference's xSadN[] with the reference's argmin
 * idiom (src/encode.cpp:725-727,762-765). */
void ref_me_fullsearch(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                       int ctu, int range, ref_mv *out, int first_ctu, int step, int count,
                       int nthreads)
{
    const int ncx = width / ctu, ncy = height / ctu, nctu = ncx * ncy;
    xSad *sad = xSadN[xLog2(ctu - 1) - 2];
    if (step < 1) step = 1;
    if (count < 0) count = (nctu - first_ctu + step - 1) / step;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int i = 0; i < count; i++) {
        int idx = first_ctu + i * step;
        if (idx >= nctu) continue;
        const int cx = (idx % ncx) * ctu, cy = (idx / ncx) * ctu;
        UInt32 best = MAX_SAD;
        int bx = 0, by = 0;
        for (int dy = -range; dy <= range; dy++) {
            if (cy + dy < 0 || cy + dy + ctu > height) continue;
            for (int dx = -range; dx <= range; dx++) {
                if (cx + dx < 0 || cx + dx + ctu > width) continue;
                UInt32 c = sad(ctu, cur + (size_t)cy * stride + cx, stride,
                               ref + (size_t)(cy + dy) * stride + cx + dx, stride);
                if (c < best) {
                    best = c;
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

/* The same for bw x bh blocks tiling the picture: the reference's primitive for width bw called with bh rows. */
void ref_me_fullsearch_wh(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                          int bw, int bh, int range, ref_mv *out, int first, int step, int count,
                          int nthreads)
{
    const int nbx = width / bw, nby = height / bh, nb = nbx * nby;
    xSad *sad = xSadN[xLog2(bw - 1) - 2];
    if (step < 1) step = 1;
    if (count < 0) count = (nb - first + step - 1) / step;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
    for (int i = 0; i < count; i++) {
        int idx = first + i * step;
        if (idx >= nb) continue;
        const int bx0 = (idx % nbx) * bw, by0 = (idx / nbx) * bh;
        UInt32 best = MAX_SAD;
        int bx = 0, by = 0;
        for (int dy = -range; dy <= range; dy++) {
            if (by0 + dy < 0 || by0 + dy + bh > height) continue;
            for (int dx = -range; dx <= range; dx++) {
                if (bx0 + dx < 0 || bx0 + dx + bw > width) continue;
                UInt32 c = sad(bh, cur + (size_t)by0 * stride + bx0, stride,
                               ref + (size_t)(by0 + dy) * stride + bx0 + dx, stride);
                if (c < best) {
                    best = c;
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

static void fetch(uint8_t *d, const uint8_t *p, int ps, const ref_mv *mv, int mv_ctu, int ncx,
                  int x0, int y0, int n)
{
    int ox = 0, oy = 0;
    if (mv) {
        ox = mv[(y0 / mv_ctu) * ncx + x0 / mv_ctu].mvx;
        oy = mv[(y0 / mv_ctu) * ncx + x0 / mv_ctu].mvy;
    }
    for (int i = 0; i < n; i++) memcpy(d + i * MAX_CU_SIZE, p + (size_t)(y0 + oy + i) * ps + x0 + ox, n);
}

/* Per-TU xSubDct + xQuant, then (optionally) xDeQuant + xIDctAdd, with the
 * reference's own scratch layout (stride MAX_CU_SIZE, in-place tmp0) */
void ref_tq_frame(const uint8_t *src, int sstride, const uint8_t *pred, int pstride,
                  const ref_mv *mv, int mv_ctu, int width, int height, int n, int qp, int st,
                  const uint8_t *modes, int16_t *level, int cstride, uint32_t *abs_sum,
                  uint8_t *rec, int rstride)
{
    const int ntx = width / n, nty = height / n, ncx = mv ? width / mv_ctu : 0;
    UInt8 s[MAX_CU_SIZE * MAX_CU_SIZE], p[MAX_CU_SIZE * MAX_CU_SIZE], r[MAX_CU_SIZE * MAX_CU_SIZE];
    Int16 t0[MAX_CU_SIZE * MAX_CU_SIZE], t1[MAX_CU_SIZE * MAX_CU_SIZE], q[MAX_CU_SIZE * MAX_CU_SIZE];
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            const int x0 = tx * n, y0 = ty * n;
            for (int i = 0; i < n; i++) memcpy(s + i * MAX_CU_SIZE, src + (size_t)(y0 + i) * sstride + x0, n);
            fetch(p, pred, pstride, mv, mv_ctu, ncx, x0, y0, n);
            UInt mode = modes ? modes[ty * ntx + tx] : MODE_INVALID;
            xSubDct(t0, s, p, MAX_CU_SIZE, t0, t1, n, n, mode);
            UInt32 sum = xQuant(q, t0, MAX_CU_SIZE, qp, n, n, (X265_SliceType)st);
            if (abs_sum) abs_sum[ty * ntx + tx] = sum;
            if (level)
                for (int i = 0; i < n; i++)
                    memcpy(level + (size_t)(y0 + i) * cstride + x0, q + i * MAX_CU_SIZE, n * 2);
            if (rec) {
                if (sum) {
                    xDeQuant(t0, q, MAX_CU_SIZE, qp, n, n, (X265_SliceType)st);
                    xIDctAdd(r, t0, p, MAX_CU_SIZE, t1, t0, n, n, mode);
                } else {
                    for (int i = 0; i < n; i++) memcpy(r + i * MAX_CU_SIZE, p + i * MAX_CU_SIZE, n);
                }
                for (int i = 0; i < n; i++) memcpy(rec + (size_t)(y0 + i) * rstride + x0, r + i * MAX_CU_SIZE, n);
            }
        }
}

/* Whole intra encoder (BASELINE.json configs[0]): xDefaultParams -> xCheckParams ->
 * xEncInit -> xEncEncode per frame.  yuv: nframes * w*h*3/2 bytes (Y,U,V planes).
 * rec_out (same size) receives each frame's reconstruction, bytes_out the
 * bitstream size per frame, md5_out 16 bytes per frame from the reference's
 * own MD5 (src/md5.cpp). */
int ref_encode_intra(const uint8_t *yuv, int w, int h, int cu, int qp, int nframes,
                     uint8_t *rec_out, int32_t *bytes_out, uint8_t *md5_out, uint8_t *bs_out,
                     int bs_cap)
{
    static X265_t hnd;
    xDefaultParams(&hnd);
    hnd.usWidth = (UInt16)w;
    hnd.usHeight = (UInt16)h;
    hnd.ucMaxCUWidth = (UInt8)cu;
    hnd.iQP = qp;
    hnd.eSliceType = SLICE_I;
    if (w % cu || h % cu || cu < 16 || w > MAX_WIDTH) return -1; /* xCheckParams would exit() */
    xCheckParams(&hnd);
    xEncInit(&hnd);
    const size_t fsz = (size_t)w * h * 3 / 2;
    UInt8 *buf = (UInt8 *)malloc(fsz + 65536);
    int bs_pos = 0;
    for (int f = 0; f < nframes; f++) {
        X265_Frame fr;
        fr.pucY = (UInt8 *)yuv + f * fsz;
        fr.pucU = fr.pucY + (size_t)w * h;
        fr.pucV = fr.pucU + (size_t)w * h / 4;
        Int32 nb = xEncEncode(&hnd, &fr, buf, (UInt32)(fsz + 65536));
        bytes_out[f] = nb;
        if (bs_out && bs_pos + nb <= bs_cap) {
            memcpy(bs_out + bs_pos, buf, nb);
            bs_pos += nb;
        }
        /* after the rotation at src/encode.cpp:965-970 the recon is refn[0] */
        memcpy(rec_out + f * fsz, hnd.refn[0].pucY, fsz);
        MD5Context ctx;
        MD5Init(&ctx);
        MD5Update(&ctx, hnd.refn[0].pucY, (UInt32)fsz);
        MD5Final(&ctx, md5_out + 16 * f);
    }
    free(buf);
    xEncFree(&hnd);
    return bs_pos;
}

/* ---- intra luma: the reference's own prediction, and a recording of its own mode decisions ---- */

static X265_t g_ih; /* a handle whose cache the prediction entry points read and write */

/* xEncIntraPredLuma (src/encode.cpp:1617) on a caller's sample arrays: ref0 = pucPixRef[0], ref1 = pucPixRef[1]
 * (4 * size + 1 bytes each).  dst: size x size, stride size. */
void ref_intra_pred_luma(const uint8_t *ref0, const uint8_t *ref1, int size, int mode, uint8_t *dst)
{
    memcpy(g_ih.cache.pucPixRef[0], ref0, 4 * size + 1);
    memcpy(g_ih.cache.pucPixRef[1], ref1, 4 * size + 1);
    xEncIntraPredLuma(&g_ih, (UInt)mode, (UInt)size);
    for (int i = 0; i < size; i++) memcpy(dst + i * size, g_ih.cache.pucPredY + i * MAX_CU_SIZE, size);
}

/* One record per CU of the recorded frame, filled from INSIDE the reference's encode loop: the luma-size entry
 * of xSadN[] is pointed at a wrapper that calls the reference's function and, when the call is one of the 35 of
 * the luma decision (src/encode.cpp:756: source = the CU cache, candidate = pucPredY), notes what the encoder
 * had in its cache at that moment. */
struct ref_intra_rec {
    int32_t x, y;
    uint8_t ref0[4 * MAX_CU_SIZE + 1]; /* pucPixRef[0] after xEncIntraLoadRef */
    uint8_t ref1[4 * MAX_CU_SIZE + 1]; /* pucPixRef[1] */
    uint8_t mpm[3];                    /* ucMostModeY */
    uint8_t best_mode;                 /* nBestModeY as stored at src/encode.cpp:864 */
    uint32_t sad[35];                  /* what xSadN returned for each mode */
    uint32_t pred_fnv[35];             /* FNV-1a of each mode's size x size prediction */
};
static X265_t *g_rec_h;
static ref_intra_rec *g_rec;
static int g_rec_cap, g_rec_n, g_rec_calls, g_rec_on, g_rec_size;
static xSad *g_rec_orig;

static UInt32 rec_sad(const UInt N, const UInt8 *pSrc, const UInt ss, const UInt8 *pRef, const UInt sr)
{
    const UInt32 v = g_rec_orig(N, pSrc, ss, pRef, sr);
    X265_Cache *c = &g_rec_h->cache;
    if (g_rec_on && pSrc == c->pucPixY && pRef == c->pucPredY) {
        const int mode = g_rec_calls % 35, cu = g_rec_calls / 35;
        if (mode == 0) {
            if (cu > 0 && cu - 1 < g_rec_cap) g_rec[cu - 1].best_mode = c->nBestModeY; /* the CU before this one */
            if (cu < g_rec_cap) {
                ref_intra_rec *r = &g_rec[cu];
                r->x = (int32_t)g_rec_h->uiCUX;
                r->y = (int32_t)g_rec_h->uiCUY;
                memcpy(r->ref0, c->pucPixRef[0], sizeof r->ref0);
                memcpy(r->ref1, c->pucPixRef[1], sizeof r->ref1);
                memcpy(r->mpm, c->ucMostModeY, 3);
            }
            g_rec_n = cu + 1;
        }
        if (cu < g_rec_cap) {
            uint32_t hsh = 2166136261u;
            for (int i = 0; i < g_rec_size; i++)
                for (int j = 0; j < g_rec_size; j++) hsh = (hsh ^ pRef[i * sr + j]) * 16777619u;
            g_rec[cu].sad[mode] = v;
            g_rec[cu].pred_fnv[mode] = hsh;
        }
        g_rec_calls++;
    }
    return v;
}

/* Encode frames 0 .. frame of yuv like ref_encode_intra and record the luma decisions of frame `frame`.
 * Returns the number of CUs recorded (<= cap), -1 for a geometry the reference rejects. */
int ref_intra_record(const uint8_t *yuv, int w, int h, int cu, int qp, int frame, ref_intra_rec *out, int cap)
{
    static X265_t hnd;
    xDefaultParams(&hnd);
    hnd.usWidth = (UInt16)w;
    hnd.usHeight = (UInt16)h;
    hnd.ucMaxCUWidth = (UInt8)cu;
    hnd.iQP = qp;
    hnd.eSliceType = SLICE_I;
    if (w % cu || h % cu || cu < 16 || cu > MAX_CU_SIZE || w > MAX_WIDTH) return -1;
    xCheckParams(&hnd);
    xEncInit(&hnd);
    const size_t fsz = (size_t)w * h * 3 / 2;
    UInt8 *buf = (UInt8 *)malloc(fsz + 65536);
    const int slot = xLog2(cu - 1) - 2;
    g_rec_h = &hnd;
    g_rec = out;
    g_rec_cap = cap;
    g_rec_n = g_rec_calls = 0;
    g_rec_size = cu;
    g_rec_orig = xSadN[slot];
    xSadN[slot] = rec_sad;
    for (int f = 0; f <= frame; f++) {
        X265_Frame fr;
        fr.pucY = (UInt8 *)yuv + f * fsz;
        fr.pucU = fr.pucY + (size_t)w * h;
        fr.pucV = fr.pucU + (size_t)w * h / 4;
        g_rec_on = f == frame;
        xEncEncode(&hnd, &fr, buf, (UInt32)(fsz + 65536));
    }
    xSadN[slot] = g_rec_orig;
    g_rec_on = 0;
    if (g_rec_n > 0 && g_rec_n - 1 < cap) out[g_rec_n - 1].best_mode = hnd.cache.nBestModeY; /* the last CU */
    free(buf);
    xEncFree(&hnd);
    return g_rec_n < cap ? g_rec_n : cap;
}

/* The reference's own bitstream writer over a pixel path computed elsewhere (x265amd_intra_encode_frame on the GPU):
 * xEncEncode's frame sequence (src/encode.cpp:587-973) with Stages 1b .. 3b -- prediction, decisions, transform
 * stage -- replaced by the caller's per-frame results; headers, xEncIntraLoadRef (for the most probable modes the
 * mode syntax needs), xWriteCU, the cache / line-buffer updates and the slice end are the reference's functions,
 * called in the reference's order.  This is INTEGRATION.md's recipe for intra pictures, executed.
 *   rec, levels: [nframes][w*h*3/2] (Y,U,V, picture geometry); mode_y, mode_c: [nframes][ncus] (mode_c = the real
 *   chroma mode).  bs_out receives the stream, bytes_out its size per frame.  Returns the total size, -1 on a
 *   geometry the reference rejects, -2 if a chroma mode is not one of the CU's candidates. */
int ref_write_intra_frames(const uint8_t *yuv, int w, int hgt, int cu, int qp, int nframes, const uint8_t *rec,
                           const int16_t *levels, const uint8_t *mode_y, const uint8_t *mode_c, uint8_t *bs_out,
                           int bs_cap, int32_t *bytes_out)
{
    static X265_t hnd;
    X265_t *h = &hnd;
    xDefaultParams(h);
    h->usWidth = (UInt16)w;
    h->usHeight = (UInt16)hgt;
    h->ucMaxCUWidth = (UInt8)cu;
    h->iQP = qp;
    h->eSliceType = SLICE_I;
    if (w % cu || hgt % cu || cu < 16 || cu > MAX_CU_SIZE || w > MAX_WIDTH) return -1;
    xCheckParams(h);
    xEncInit(h);
    const size_t ysz = (size_t)w * hgt, csz = ysz / 4, fsz = ysz + 2 * csz;
    const int ncx = w / cu, ncy = hgt / cu, nc = cu / 2;
    UInt8 *buf = (UInt8 *)malloc(fsz + 65536);
    X265_BitStream *pBS = &h->bs;
    X265_Cache *c = &h->cache;
    int pos = 0, bad = 0;
    for (int f = 0; f < nframes && !bad; f++) {
        X265_Frame fr;
        fr.pucY = (UInt8 *)yuv + f * fsz;
        fr.pucU = fr.pucY + ysz;
        fr.pucV = fr.pucU + csz;
        h->pFrameCur = &fr;
        h->pFrameRec = &h->refn[MAX_REF_NUM];
        h->iPoc++;
        xBitStreamInit(pBS, buf, (Int)(fsz + 65536));
        xPutBits32(pBS, 0x01000000); /* SPS, src/encode.cpp:626-631 */
        xPutBits(pBS, 0x47, 8);
        xPutBits(pBS, 0x01, 8);
        xWriteSPS(h);
        xBitFlush(pBS);
        xPutBits32(pBS, 0x01000000); /* PPS, :633-638 */
        xPutBits(pBS, 0x48, 8);
        xPutBits(pBS, 0x01, 8);
        xWritePPS(h);
        xBitFlush(pBS);
        xPutBits32(pBS, (h->iPoc == 0 ? 0x45010000 : 0x41010000)); /* slice, :657-662 */
        xPutBits(pBS, 0x01, 8);
        xWriteSliceHeader(h);
        xEncCahceInit(h);
        xCabacInit(h);
        xCabacReset(&h->cabac);
        const uint8_t *ry = rec + f * fsz, *ru = ry + ysz, *rv = ru + csz;
        const int16_t *ly = levels + f * fsz, *lu = ly + ysz, *lv = lu + csz;
        for (int cy = 0; cy < ncy && !bad; cy++) {
            h->uiCUY = (UInt32)(cy * cu);
            xEncCahceInitLine(h, (UInt)(cy * cu));
            for (int cx = 0; cx < ncx; cx++) {
                const int x0 = cx * cu, y0 = cy * cu, idx = f * ncx * ncy + cy * ncx + cx;
                h->uiCUX = (UInt32)x0;
                xEncCacheLoadCU(h, (UInt)x0, (UInt)y0);
                xEncIntraLoadRef(h, 0, 0, (UInt)cu); /* ucMostModeY, bValid */
                const int my = mode_y[idx], mc = mode_c[idx];
                int cand[6] = {PLANAR_IDX, VER_IDX, HOR_IDX, DC_IDX, LM_CHROMA_IDX, my}; /* :781-793 */
                for (int i = 0; i < 4; i++)
                    if (cand[i] == my) {
                        cand[i] = 34;
                        break;
                    }
                int kc = -1;
                for (int k = 0; k < 6 && kc < 0; k++)
                    if (k != 4 && cand[k] == mc) kc = k;
                if (kc < 0) {
                    bad = 1;
                    break;
                }
                c->nBestModeY = (UInt8)my;
                c->nBestModeC = (UInt8)kc;
                UInt32 sum[3] = {0, 0, 0};
                for (int i = 0; i < cu; i++)
                    for (int j = 0; j < cu; j++) {
                        const Int16 v = ly[(size_t)(y0 + i) * w + x0 + j];
                        c->psCoefY[i * MAX_CU_SIZE + j] = v;
                        sum[0] += (UInt32)abs(v);
                        c->pucRecY[i * MAX_CU_SIZE + j] = ry[(size_t)(y0 + i) * w + x0 + j];
                    }
                for (int i = 0; i < nc; i++)
                    for (int j = 0; j < nc; j++) {
                        const size_t o = (size_t)(y0 / 2 + i) * (w / 2) + x0 / 2 + j;
                        c->psCoefU[i * MAX_CU_SIZE / 2 + j] = lu[o];
                        c->psCoefV[i * MAX_CU_SIZE / 2 + j] = lv[o];
                        sum[1] += (UInt32)abs(lu[o]);
                        sum[2] += (UInt32)abs(lv[o]);
                        c->pucRecU[i * MAX_CU_SIZE / 2 + j] = ru[o];
                        c->pucRecV[i * MAX_CU_SIZE / 2 + j] = rv[o];
                    }
                c->bCbf[0] = sum[0] != 0; /* :888-890 */
                c->bCbf[1] = sum[1] != 0;
                c->bCbf[2] = sum[2] != 0;
                xWriteCU(h, 0, (UInt)(cy == ncy - 1 && cx == ncx - 1)); /* Stage 4, :928 */
                xEncCacheStoreCU(h, (UInt)x0, (UInt)y0);               /* Stage 5a, :931 */
                xEncCacheUpdate(h, 0, 0, (UInt)cu, (UInt)cu);          /* Stage 5b, :934-935 */
                c->uiOffset += cu;
            }
        }
        if (bad) break;
        xCabacFlush(&h->cabac, pBS);
        xWriteSliceEnd(h);
        X265_Frame tmp = h->refn[MAX_REF_NUM]; /* :965-970 */
        for (int i = MAX_REF_NUM; i > 0; i--) h->refn[i] = h->refn[i - 1];
        h->refn[0] = tmp;
        const Int32 nb = xBitFlush(pBS);
        bytes_out[f] = nb;
        if (pos + nb <= bs_cap) memcpy(bs_out + pos, buf, nb);
        pos += nb;
    }
    free(buf);
    xEncFree(h);
    return bad ? -2 : pos;
}

} /* extern "C" */
