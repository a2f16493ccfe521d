/*
 * x265_oracle.h -- CPU restatement of the chenm001/x265 ME+TQ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (x265_amd/, the C-ABI
 * library) may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * Parity status: PINNED for SAD, forward/inverse transforms, xSubDct, xQuant,
 * xDeQuant and xIDctAdd -- checked against the reference itself compiled from
 * /root/reference (oracle/_ref, see oracle/Makefile) and against the committed
 * fixtures in tests/golden/ that were generated from that build.
 * Motion search has no reference implementation (src/encode.cpp:581-584); its
 * rule is SURVEY.md App. A.7 and it is pinned against a loop over the compiled
 * reference's own xSadN[4] (oracle/ref_shim.cpp).
 *
 * Every function cites the reference lines it restates (paths under
 * /root/reference).
 */
#ifndef X265_ORACLE_H
#define X265_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MODE_INVALID 255u /* inc/x265.h:70 */
#define ORC_SLICE_I 0         /* inc/x265.h:40 */
#define ORC_SLICE_P 1         /* inc/x265.h:41 */
#define ORC_MAX_SAD (1u << 30) /* inc/config.h:52 */

typedef struct orc_mv {
    int16_t mvx, mvy;
    uint32_t sad;
} orc_mv;

/* transform ROM, row-major [k][j] (src/table.cpp:72-145); n in {4,8,16,32} */
const int8_t *orc_dct_matrix(int n);
/* 4x4 DST matrix (src/pixel.cpp:133-151, src/table.cpp:148-153) */
const int8_t *orc_dst4_matrix(void);
int orc_chroma_qp(int qp); /* g_aucChromaScale, src/table.cpp:166-171 */

/* xSad{4,8,16,32,64}xN: src/pixel.cpp:30-93 */
uint32_t orc_sad(int w, int n, const uint8_t *src, int ss, const uint8_t *ref, int sr);
/* SATD, HM definition; the reference has none -- PARITY UNPINNED (see x265_oracle.c) */
uint32_t orc_satd(int w, int n, const uint8_t *src, int ss, const uint8_t *ref, int sr);

/* xDctN[idx] / xInvDctN[idx]: src/pixel.cpp:114-312,321-569,571-585.
 * idx: 0=DST4 1=DCT4 2=DCT8 3=DCT16 4=DCT32 (same table order). */
void orc_fwd_1d(int idx, int16_t *dst, const int16_t *src, unsigned stride, int lines, int shift);
void orc_inv_1d(int idx, int16_t *dst, const int16_t *src, unsigned stride, int lines, int shift);

/* xSubDct: src/pixel.cpp:668-722 */
void orc_sub_dct(int16_t *dst, const uint8_t *src, const uint8_t *ref, unsigned stride,
                 int16_t *tmp0, int16_t *tmp1, int w, int h, unsigned mode);
/* xIDctAdd: src/pixel.cpp:724-776 */
void orc_idct_add(uint8_t *dst, const int16_t *src, const uint8_t *ref, unsigned stride,
                  int16_t *tmp0, int16_t *tmp1, int w, int h, unsigned mode);
/* xQuant: src/pixel.cpp:591-633 */
uint32_t orc_quant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h,
                   int slice_type);
/* xDeQuant: src/pixel.cpp:635-666 */
void orc_dequant(int16_t *dst, const int16_t *src, unsigned stride, unsigned qp, int w, int h,
                 int slice_type);

/* ---- frame-level restatements (the shapes the C-ABI entry points take) ---- */

/* Exhaustive integer-pel search, SURVEY.md App. A.7 with the argmin idiom of
 * src/encode.cpp:725-727,762-765 and the SAD of src/pixel.cpp:30-41.
 * out[cty * (width/ctu) + ctx].  first_ctu/step/count select a deterministic
 * subset of CTUs (count<0: all) so the bench can time a bounded sample. */
void orc_me_fullsearch(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                       int ctu, int range, orc_mv *out, int first_ctu, int step, int count,
                       int nthreads);

/* Forward TQ of one plane cut into n x n TUs (encode.cpp:868-873 sequence per TU):
 * residual = src - pred ; xSubDct ; xQuant.  Picture-geometry outputs: level[y][x]
 * with stride `cstride` (int16 elements), abs_sum[tuY * (width/n) + tuX].
 * pred is either a plane with stride `pstride`, or, if mv != NULL, the
 * integer-pel motion-compensated block of `pred` addressed per mv_ctu x mv_ctu CTU.
 * modes: per-TU intra mode (NULL = MODE_INVALID everywhere). */
void orc_tq_forward(const uint8_t *src, int sstride, const uint8_t *pred, int pstride,
                    const orc_mv *mv, int mv_ctu, int width, int height, int n, int qp,
                    int slice_type, const uint8_t *modes, int16_t *level, int cstride,
                    uint32_t *abs_sum, int nthreads);

/* Reconstruction (encode.cpp:893-905 per TU): xDeQuant ; xIDctAdd ; or copy pred if abs sum 0 */
void orc_tq_inverse(const int16_t *level, int cstride, const uint8_t *pred, int pstride,
                    const orc_mv *mv, int mv_ctu, int width, int height, int n, int qp,
                    int slice_type, const uint8_t *modes, uint8_t *rec, int rstride, int nthreads);

/* The same search for bw x bh blocks tiling the picture (the inter PU shapes): xSadN[log2(bw) - 2](bh, ..),
 * src/pixel.cpp:30-101.  out[(by0 / bh) * (width / bw) + bx0 / bw]. */
void orc_me_fullsearch_wh(const uint8_t *cur, const uint8_t *ref, int width, int height, int stride,
                          int bw, int bh, int range, orc_mv *out, int first, int step, int count,
                          int nthreads);

/* ---- intra luma prediction and mode decision ----
 * ref0[4n + 1]: the CU's reference samples as xEncIntraLoadRef leaves them in pucPixRef[0] after padding
 * (src/encode.cpp:1193-1256): left column bottom-up, corner, row above. */
void orc_intra_filter(const uint8_t *ref0, uint8_t *ref1, int n);                     /* src/encode.cpp:1260-1265 */
void orc_intra_pred_luma(const uint8_t *ref0, const uint8_t *ref1, int n, int mode,  /* src/encode.cpp:1617-1652 */
                         uint8_t *dst, int ds);
void orc_intra_decide_luma(const uint8_t *src, int ss, const uint8_t *ref0, const uint8_t *mpm, uint32_t lambda,
                           int n, uint32_t *sads, uint32_t *best_mode, uint32_t *best_cost,
                           uint8_t *best_pred);                                       /* src/encode.cpp:725-765 */

/* One intra frame, the pixel path of the encode loop (src/encode.cpp:668-941 without the entropy coder):
 * yuv / rec: planes Y, U, V back to back (strides = widths); mode_y / mode_c per CU; levels may be NULL. */
void orc_intra_encode_frame(const uint8_t *yuv, int w, int h, int cu, int qp, uint8_t *rec, uint8_t *mode_y,
                            uint8_t *mode_c, int16_t *levels);

#ifdef __cplusplus
}
#endif
#endif
