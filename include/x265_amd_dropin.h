/*
 * x265_amd_dropin.h -- per-block drop-ins with the reference's OWN signatures.
 *
 * The reference's pixel layer is three mutable function-pointer tables and four
 * free functions (inc/x265.h:290-298).  Every function below has the parameter
 * list of the one it replaces, takes HOST pointers like it, and computes on the
 * GPU with the kernels of the batched API (x265_amd.h), so a maintainer can write
 *
 *     xSadN[4]    = x265amd_xSad64xN;          // src/pixel.cpp:95-101
 *     xDctN[4]    = x265amd_xDCT32;            // src/pixel.cpp:571-577
 *     xInvDctN[0] = x265amd_xInvDST4;          // src/pixel.cpp:579-585
 *
 * or call x265amd_xSubDct where src/encode.cpp:868 calls xSubDct, and run the
 * encoder (and its CHECK_TV comparisons) unchanged.  One block per call means a
 * launch and two PCIe copies per call: this layer is for bring-up and
 * stage-by-stage checking.  The batched entry points are the fast path.
 *
 * Differences from the reference, all forced by the signatures:
 *  - the functions return what the reference's return; the status of the last
 *    call on the calling thread is x265amd_dropin_status() (0 = ok).  On error
 *    nothing is written, xSad returns 0xFFFFFFFF and xQuant returns 0.
 *  - x265amd_init() must have been called for the current device; the first
 *    drop-in call on a device allocates that device's 16 KB staging block, which
 *    lives until x265amd_dropin_teardown() (or the process) ends it.
 *  - the drop-ins are SYNCHRONOUS helpers: they run on the null stream with
 *    blocking copies, one call at a time per process (a mutex serialises them),
 *    and so serialise against the caller's other default-stream work.
 *  - piTmp0 / piTmp1 (the reference's scratch) are accepted and left untouched;
 *    pDst may alias either, as at src/encode.cpp:868-872.
 *  - blocks are square, 4..32 (the only shapes the reference ever passes:
 *    src/encode.cpp:868-917); SAD heights 1..64; 1-D passes up to 64 lines.
 */
#ifndef X265_AMD_DROPIN_H
#define X265_AMD_DROPIN_H

#include "x265_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

int x265amd_dropin_status(void);
/* number of drop-in calls served by the GPU (status ok) since the library was loaded */
uint64_t x265amd_dropin_calls(void);
/* frees the current device's staging block (the next drop-in call allocates it again) */
int x265amd_dropin_teardown(void);

/* xSadN[0..4] = xSad4xN .. xSad64xN (typedef xSad inc/x265.h:290; src/pixel.cpp:30-101) */
uint32_t x265amd_xSad4xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc, const uint8_t *pRef, const unsigned nStrideRef);
uint32_t x265amd_xSad8xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc, const uint8_t *pRef, const unsigned nStrideRef);
uint32_t x265amd_xSad16xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc, const uint8_t *pRef, const unsigned nStrideRef);
uint32_t x265amd_xSad32xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc, const uint8_t *pRef, const unsigned nStrideRef);
uint32_t x265amd_xSad64xN(const unsigned N, const uint8_t *pSrc, const unsigned nStrideSrc, const uint8_t *pRef, const unsigned nStrideRef);

/* xDctN[0..4] = {xDST4, xDCT4, xDCT8, xDCT16, xDCT32} (typedef xDCT inc/x265.h:292; src/pixel.cpp:114-312,
 * 571-577): one 1-D pass over nLines rows, output transposed, (sum + 2^(nShift-1)) >> nShift, truncating store.
 * xDST4 always processes 4 lines (src/pixel.cpp:138). */
void x265amd_xDST4(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xDCT4(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xDCT8(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xDCT16(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xDCT32(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);

/* xInvDctN[0..4] (src/pixel.cpp:321-569, 579-585): reads columns, writes rows, store-then-clip.
 * xInvDST4 always processes 4 lines and xInvDCT32 always 32 (src/pixel.cpp:351, 513). */
void x265amd_xInvDST4(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xInvDCT4(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xInvDCT8(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xInvDCT16(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);
void x265amd_xInvDCT32(int16_t *pDst, int16_t *pSrc, unsigned nStride, int nLines, int nShift);

/* xSubDct, inc/x265.h:295, src/pixel.cpp:668-722 */
void x265amd_xSubDct(int16_t *pDst, uint8_t *pSrc, uint8_t *pRef, unsigned nStride, int16_t *piTmp0, int16_t *piTmp1, int iWidth, int iHeight, unsigned nMode);
/* xIDctAdd, inc/x265.h:296, src/pixel.cpp:724-776 */
void x265amd_xIDctAdd(uint8_t *pDst, int16_t *pSrc, uint8_t *pRef, unsigned nStride, int16_t *piTmp0, int16_t *piTmp1, int iWidth, int iHeight, unsigned nMode);
/* xQuant, inc/x265.h:297, src/pixel.cpp:591-633 (eSType: X265_SliceType, an int-sized enum) */
uint32_t x265amd_xQuant(int16_t *pDst, int16_t *pSrc, unsigned nStride, unsigned nQP, int iWidth, int iHeight, int eSType);
/* xDeQuant, inc/x265.h:298, src/pixel.cpp:635-666 */
void x265amd_xDeQuant(int16_t *pDst, int16_t *pSrc, unsigned nStride, unsigned nQP, int iWidth, int iHeight, int eSType);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_DROPIN_H */
