/*
 * x265_amd.h -- C-ABI of the MI355X (gfx950) ME + transform/quant hot path.
 *
 * This is the drop-in boundary for the data-parallel path of chenm001/x265:
 * the three plug-in tables and four free functions of the reference's pixel
 * layer (inc/x265.h:290-298), re-cut from "one block per call" into "one
 * picture plane (or a batch of planes) per call", because a GPU launch per
 * 32x32 block would be all overhead.  Every entry point below names the
 * reference interface it replaces (paths are under /root/reference).
 *
 * Conventions
 *  - plain C types only; every pointer marked `d_` is DEVICE memory (HBM).
 *  - every function returns int: 0 = ok, negative = error (x265amd_strerror()).
 *    Nothing aborts or exits (the reference asserts / exit()s instead:
 *    src/params.cpp:80-83, src/pixel.cpp:684).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    Calls are asynchronous on that stream; nothing here allocates or
 *    synchronises except x265amd_malloc, _free, _memcpy_h2d, _memcpy_d2h and _stream_sync.
 *  - picture planes are 8-bit, row-major, stride in bytes (the reference keeps
 *    stride == width: src/encode.cpp:41-45; here any stride works).  Widths and
 *    heights must be multiples of the block size in use, exactly like
 *    src/params.cpp:73-74 demands of the CU size.
 *  - coefficient planes are Int16 in PICTURE GEOMETRY: level[y][x] belongs to
 *    the pixel position (x, y); the TU that owns it is (x / tu, y / tu).
 *  - all library state is kept PER DEVICE (the role of the reference's handle X265_t,
 *    inc/x265.h:186-212): x265amd_init(device) sets it up once per device and makes that device
 *    current on the calling thread; every other call works on the thread's current HIP device
 *    and returns X265AMD_ERR_NOT_INIT if x265amd_init has not run for it.  One host thread or
 *    process per GPU, or several threads on one GPU, may call concurrently.
 */
#ifndef X265_AMD_H
#define X265_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265AMD_VERSION 100

/* eIntraMode MODE_INVALID, inc/x265.h:70: "no intra mode" => always DCT */
#define X265AMD_MODE_INVALID 255

/* X265_SliceType, inc/x265.h:39-43 (selects the quantiser rounding offset) */
#define X265AMD_SLICE_I 0
#define X265AMD_SLICE_P 1
#define X265AMD_SLICE_B 2

/* MAX_SAD, inc/config.h:52 */
#define X265AMD_MAX_SAD (1u << 30)

enum {
    X265AMD_OK = 0,
    X265AMD_ERR_ARG = -1,       /* bad geometry / null pointer / unsupported size */
    X265AMD_ERR_NO_DEVICE = -2, /* no HIP device, or device index out of range */
    X265AMD_ERR_ARCH = -3,      /* device is not gfx950 */
    X265AMD_ERR_NOT_INIT = -4,  /* x265amd_init() not called */
    X265AMD_ERR_LDS = -5,       /* search window does not fit the 160 KiB LDS */
    X265AMD_ERR_WORKSPACE = -6, /* x265amd_me_fullsearch_sea_static: the job needs more than the static scratch */
    X265AMD_ERR_BUFFER = -7,    /* x265amd_intra_write_picture: the output buffer is too small */
    X265AMD_ERR_HIP = -1000     /* -1000 - hipError_t */
};

/* One motion vector + its cost, 8 bytes.  (The reference has no MV type:
 * P slices are `assert(0)` at src/encode.cpp:581-584.) */
typedef struct x265amd_mv {
    int16_t mvx, mvy; /* integer-pel displacement of the best candidate */
    uint32_t sad;     /* its SAD (what xSadN[] would return) */
} x265amd_mv;

/* ---------------------------------------------------------------- runtime */

/* Select `device` (made current on the calling thread), check it is gfx950, opt every
 * kernel in to the device's full LDS.  Replaces nothing in the reference (it has no device);
 * closest relative is xEncInit, src/encode.cpp:33-52.  Idempotent, thread-safe; call it once
 * per device, from each thread that will use that device (later calls only select it). */
int x265amd_init(int device);
int x265amd_version(void);
const char *x265amd_strerror(int status);

/* Device memory + copies so that a C/C++ host needs no HIP headers
 * (xEncInit's MALLOC of the frame buffers, src/encode.cpp:40-46; xEncFree :54-62). */
int x265amd_malloc(void **d_ptr, size_t bytes);
int x265amd_free(void *d_ptr);
/* page-locked host memory for the buffers that cross PCIe every picture (source frames up; vectors, levels
 * down): copies from / to it run at link speed and truly asynchronously */
int x265amd_malloc_host(void **h_ptr, size_t bytes);
int x265amd_free_host(void *h_ptr);
int x265amd_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int x265amd_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
/* Device memory into page-locked host memory (x265amd_malloc_host's; anything else is X265AMD_ERR_ARG) by a kernel
 * of `stream` that stores across the link, instead of by the copy engine.  For what comes down after the kernels
 * of a picture when several streams are at work: a copy-engine download queued behind its stream's kernels keeps
 * the uploads of the OTHER streams from starting until that stream has drained (measured: host/encode_p.cpp
 * --bench, DESIGN.md section 5 -- 14 Gpixels/s with such downloads, 23 with kernels writing the results);
 * a kernel holds up nothing.  bytes and both pointers multiples of 4.  The reference has no device boundary and
 * nothing like it. */
int x265amd_copy_to_host(void *h_dst, const void *d_src, size_t bytes, void *stream);
int x265amd_stream_sync(void *stream);
/* a hipStream_t for hosts without HIP headers: two batches of segments on two streams overlap one batch's
 * PCIe copies with the other's kernels (host/encode_p.cpp --bench) */
int x265amd_stream_create(void **stream);
int x265amd_stream_destroy(void *stream);

/* g_aucChromaScale[qp], src/table.cpp:166-171 (QP used for the chroma planes,
 * src/encode.cpp:596).  Host-side helper, no device work.  qp 0..51. */
int x265amd_chroma_qp(int qp);

/* -------------------------------------------------------------------- SAD */

/* Batched xSadN[log2(w)-2](n, src, stride_src, ref, stride_ref)
 * (typedef xSad + table: inc/x265.h:290-291, src/pixel.cpp:30-101).
 * Block b compares the w x n block at d_src + src_off[b] with the one at
 * d_ref + ref_off[b]; out[b] receives the UInt32 the reference returns.
 * w in {4,8,16,32,64}, 1 <= n <= 64. */
int x265amd_sad_blocks(int w, int n, const uint8_t *d_src, int stride_src, const uint8_t *d_ref,
                       int stride_ref, const uint32_t *d_src_off, const uint32_t *d_ref_off,
                       int nblocks, uint32_t *d_out, void *stream);

/* Batched SATD with the calling convention of x265amd_sad_blocks.  The reference has NO SATD
 * (SURVEY.md 8a row a14); BASELINE.json's north_star names it as a primitive of the search.  Definition:
 * the HEVC test model's (TComRdCost::xCalcHADs8x8 / xCalcHADs4x4) -- per 8x8 sub-block
 * (sum |H8 D H8^T| + 2) >> 2; blocks 4 wide or 4 high use 4x4 sub-blocks, (sum |H4 D H4^T| + 1) >> 1.
 * PARITY UNPINNED: checked against this repo's own CPU restatement and closed-form cases only.
 * w, n in {4, 8, 16, 32, 64}. */
int x265amd_satd_blocks(int w, int n, const uint8_t *d_src, int stride_src, const uint8_t *d_ref,
                        int stride_ref, const uint32_t *d_src_off, const uint32_t *d_ref_off,
                        int nblocks, uint32_t *d_out, void *stream);

/* ---------------------------------------------------------- motion search */

/* Exhaustive integer-pel search, one vector per ctu x ctu block (SURVEY.md
 * App. A.7).  The reference has no search of its own; this is the loop its
 * encode.cpp would run over xSadN[log2(ctu)-2] (src/pixel.cpp:30-41) with its
 * argmin idiom (`best = MAX_SAD; if (cost < best)`, src/encode.cpp:725-727,
 * 762-765): candidates (dx, dy) in [-range, range]^2 in raster order, dy
 * outer; candidates whose block leaves the picture are skipped; the first
 * strict minimum wins.
 *
 * Batched over frames and references:
 *   cur(f)    = d_cur + f * cur_frame_stride
 *   ref(f, r) = d_ref + f * ref_frame_stride + r * ref_ref_stride   (strides in bytes, may be negative)
 *   d_out[(f * nrefs + r) * nctu + cty * (width / ctu) + ctx]
 * ctu in {16, 32, 64} -- the CTU or, for partition searches, the 32x32 / 16x16
 * PU size: the same kernel serves all three; width, height multiples of ctu;
 * stride a multiple of 4 and planes 4-byte aligned; the (ctu + 2*range + 8)^2-byte
 * window must fit in LDS (range <= 128 at ctu 64), else X265AMD_ERR_LDS. */
int x265amd_me_fullsearch(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                          int stride, int ctu, int range, int nframes, int64_t cur_frame_stride,
                          int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride,
                          x265amd_mv *d_out, void *stream);

/* The same search with EXACT successive elimination: identical vectors and
 * costs (ties included), far fewer SADs.  A candidate is dropped only when a
 * lower bound of its SAD (sum over the sixteen 16x16 sub-blocks of
 * |sum(cur) - sum(ref)|) exceeds the SAD of a candidate already evaluated, so
 * the first strict minimum of src/encode.cpp:762-765 is untouched.  Needs a
 * caller-provided device workspace of x265amd_me_workspace_bytes() bytes
 * (16-byte aligned) for the reference box sums, the per-group bounds and the
 * first-batch schedule.  ctu: 64, or 32 (a 32x32 PU grid; its bound is the sum over the sixteen 8x8 sub-blocks:
 * the same kernels instantiated at half the size) -- then the workspace is x265amd_me_workspace_bytes_ctu(). */
size_t x265amd_me_workspace_bytes(int width, int height, int range, int nframes, int nrefs); /* ctu = 64 */
size_t x265amd_me_workspace_bytes_ctu(int width, int height, int ctu, int range, int nframes, int nrefs);
int x265amd_me_fullsearch_sea(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                              int stride, int ctu, int range, int nframes,
                              int64_t cur_frame_stride, int64_t ref_frame_stride, int nrefs,
                              int64_t ref_ref_stride, x265amd_mv *d_out, void *d_workspace,
                              size_t workspace_bytes, void *stream);

/* x265amd_me_fullsearch_sea for callers that cannot allocate or bring a workspace (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.
 * Same kernels, same arguments minus the workspace: the scratch lives in a static device array of
 * X265AMD_ME_STATIC_WORKSPACE_BYTES (40 MiB per GPU, held from library load to process end) whose address is
 * taken once per device in x265amd_init.  The call allocates nothing.  Jobs whose
 * x265amd_me_workspace_bytes() exceeds the array (e.g. more than two 1080p planes at range 64) are refused with
 * X265AMD_ERR_WORKSPACE -- use x265amd_me_fullsearch_sea for those.  Every call on a GPU shares that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race; serialise them. */
#define X265AMD_ME_STATIC_WORKSPACE_BYTES (40u << 20)
size_t x265amd_me_static_workspace_bytes(void);
int x265amd_me_fullsearch_sea_static(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height,
                                     int stride, int ctu, int range, int nframes,
                                     int64_t cur_frame_stride, int64_t ref_frame_stride, int nrefs,
                                     int64_t ref_ref_stride, x265amd_mv *d_out, void *stream);

/* Every square PU size of every CTU in ONE exhaustive pass ("per-CTU/PU" search; SURVEY.md 8f rank 4: SAD
 * tree 16 -> 32 -> 64): each displacement's sixteen 16x16 SADs are added up the tree and each PU keeps its own
 * winner under its own legality.  d_out64 / d_out32 / d_out16 receive exactly what x265amd_me_fullsearch
 * writes for ctu = 64 / 32 / 16 (same layout: [(f * nrefs + r)][block raster]); any of them may be NULL.
 * width, height multiples of 64; the unclipped (64 + 2*range)^2 window must fit in LDS, else X265AMD_ERR_LDS. */
int x265amd_me_fullsearch_pu(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                             int range, int nframes, int64_t cur_frame_stride, int64_t ref_frame_stride,
                             int nrefs, int64_t ref_ref_stride, x265amd_mv *d_out64, x265amd_mv *d_out32,
                             x265amd_mv *d_out16, void *stream);

/* EVERY inter PU shape of every CTU in one exhaustive pass (remainder: 8x8 leaves and the
 * rectangular PUs 2NxN / Nx2N at 64, 32 and 16): 169 PUs per 64x64 CTU, all sums of its sixty-four 8x8 SADs at a
 * displacement -- the reference's primitives take any number of rows, xSadN[log2(w) - 2](h, ..)
 * (src/pixel.cpp:30-101).  d_out: HOST array of X265AMD_PU_SHAPES device pointers, indexed by x265amd_pu_shape;
 * entry s receives, for plane (f, r) at offset (f * nrefs + r) * (width / w_s) * (height / h_s), the raster of
 * the w_s x h_s blocks tiling the picture: for each the first strict SAD minimum in raster order over the
 * displacements that keep THAT block inside the picture (src/encode.cpp:725-727, 762-765), i.e. what
 * x265amd_me_fullsearch returns for the square shapes.  Entries may be NULL (not all).  width, height multiples
 * of 64; range <= 180; the unclipped (64 + 2*range)^2 window must fit in LDS, else X265AMD_ERR_LDS. */
enum x265amd_pu_shape {
    X265AMD_PU_64x64 = 0,
    X265AMD_PU_64x32,
    X265AMD_PU_32x64,
    X265AMD_PU_32x32,
    X265AMD_PU_32x16,
    X265AMD_PU_16x32,
    X265AMD_PU_16x16,
    X265AMD_PU_16x8,
    X265AMD_PU_8x16,
    X265AMD_PU_8x8,
    X265AMD_PU_SHAPES
};
int x265amd_me_fullsearch_pu_all(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                                 int range, int nframes, int64_t cur_frame_stride, int64_t ref_frame_stride,
                                 int nrefs, int64_t ref_ref_stride, x265amd_mv *const *d_out, void *stream);

/* ------------------------------------------------------- intra decision */

/* The luma intra mode decision of `ncus` independent CUs (first slice): for each CU the 35
 * predictions of xEncIntraPredLuma (src/encode.cpp:1617-1652 over xPredIntraPlanar :1353, xPredIntraDc :1418,
 * xPredIntraAng :1446), their SADs against the source block (xSadN, src/pixel.cpp:30-101) and the cost / argmin of
 * src/encode.cpp:725-765: cost = SAD + lambda * (1 for the first most probable mode, 2 for the other two, 3
 * otherwise; the reference sets lambda = QP, :597), first strict minimum in mode order from MAX_SAD.
 *   d_src, src_stride, width, height: the luma plane; d_xy[2 * i], d_xy[2 * i + 1]: top-left corner of CU i (a CU that
 *     does not lie inside the plane gets {0xffffffff, 0xffffffff} and nothing else is written for it)
 *   size: 4, 8, 16 or 32 (MAX_CU_SIZE, inc/config.h:38)
 *   d_ref + i * ref_stride: the 4 * size + 1 reference samples of CU i as xEncIntraLoadRef (src/encode.cpp:1163-1256)
 *     leaves them in pucPixRef[0] after padding -- left column from the bottom up, the corner, the row above; the
 *     [1 2 1] smoothing (:1260-1265) is applied here
 *   d_mpm + 3 * i: ucMostModeY of CU i (src/encode.cpp:1268-1293)
 *   d_best[i]: winner and its cost; d_sads (may be NULL): [ncus][35] plain SADs; d_pred (may be NULL):
 *     [ncus][size * size] the winner's prediction (what src/encode.cpp:867 regenerates for the residual).
 * In the encoder the CUs of one launch are those of one anti-diagonal: their neighbours are reconstructed. */
typedef struct x265amd_intra_best {
    uint32_t mode;
    uint32_t cost;
} x265amd_intra_best;
int x265amd_intra_decide_luma(const uint8_t *d_src, int src_stride, int width, int height, const int32_t *d_xy,
                              int ncus, int size, const uint8_t *d_ref, int ref_stride, const uint8_t *d_mpm,
                              uint32_t lambda, x265amd_intra_best *d_best, uint32_t *d_sads, uint8_t *d_pred,
                              void *stream);

/* One intra frame on the GPU: the pixel path of the reference's encode loop (src/encode.cpp:668-941 minus the
 * entropy coder) -- reference samples from the reconstruction (xEncIntraLoadRef :1163), luma and chroma mode
 * decisions (:725-853), prediction, xSubDct + xQuant, xDeQuant + xIDctAdd (:863-925), reconstruction written back
 * as the next CUs' neighbours (:930-934).  CU = PU = TU as there; 4:2:0; cu = 16 or 32.  The CUs with the same
 * cux + 2 * cuy do not depend on one another: a frame is x265amd_intra_waves() steps, each one prediction launch
 * over that anti-diagonal's CUs plus the transform stage.  Outputs: the reconstruction (bit-exact with the
 * reference's, i.e. its MD5 SEI), per-CU modes, and the quantised levels the entropy coder needs.
 * All planes are device memory; strides in elements; the level planes have the strides of their pixel planes. */
typedef struct x265amd_intra_frame {
    int32_t width, height; /* luma size; multiples of cu */
    int32_t cu;            /* 16 or 32 (src/params.cpp:76, MAX_CU_SIZE inc/config.h:38) */
    int32_t qp;            /* also the lambda of the luma decision (src/encode.cpp:597); chroma uses g_aucChromaScale[qp] */
    int32_t stride_y, stride_c;
    const uint8_t *d_src_y, *d_src_u, *d_src_v;
    uint8_t *d_rec_y, *d_rec_u, *d_rec_v;       /* OUT, and IN of later CUs */
    uint8_t *d_pred_y, *d_pred_u, *d_pred_v;    /* OUT of the prediction stage, IN of the transform stage */
    int16_t *d_level_y, *d_level_u, *d_level_v; /* OUT: quantised levels, picture geometry */
    uint8_t *d_mode_y, *d_mode_c;               /* OUT [ncus]: luma mode, chroma mode (the real mode, not the index) */
    uint32_t *d_cost_y, *d_cost_c;              /* optional OUT [ncus]: the winners' costs */
    uint32_t *d_abs_sum_y, *d_abs_sum_u, *d_abs_sum_v; /* optional OUT [ncus]: xQuant's return values (the cbf) */
} x265amd_intra_frame;
/* number of anti-diagonals of the frame: width / cu + 2 * (height / cu - 1); -1 for a bad descriptor */
int x265amd_intra_waves(const x265amd_intra_frame *frame);
/* the prediction stage of one anti-diagonal (Stage 1b .. 2 and the prediction calls of Stage 3a): reads d_rec_*
 * of earlier waves and d_mode_y of the left neighbours, writes d_pred_*, d_mode_*, d_cost_* of this wave's CUs */
int x265amd_intra_predict_wave(const x265amd_intra_frame *frame, int wave, void *stream);
/* the whole frame: for every wave, x265amd_intra_predict_wave, then x265amd_tq_roundtrip of the three planes */
int x265amd_intra_encode_frame(const x265amd_intra_frame *frame, void *stream);
/* A batch of independent intra frames (every frame of the reference is one: SLICE_I only, src/encode.cpp:491) in
 * the same x265amd_intra_waves() steps: step d works on anti-diagonal d of every frame at once, which is how a
 * picture whose anti-diagonals hold only a few dozen CUs fills 256 compute units.  `frame` describes frame 0; frame k
 * uses every luma plane pointer + k * luma_frame_stride, every chroma plane pointer + k * chroma_frame_stride and
 * every per-CU array + k * cu_frame_stride (strides in elements of the respective array; for 4:2:0 frame buffers
 * stored one after the other both plane strides are width * height * 3 / 2).  Frames must not overlap
 * (X265AMD_ERR_ARG); nframes <= 65535.  Same results as nframes calls of x265amd_intra_encode_frame. */
int x265amd_intra_encode_frames(const x265amd_intra_frame *frame, int nframes, int64_t luma_frame_stride,
                                int64_t chroma_frame_stride, int64_t cu_frame_stride, void *stream);

/* The syntax of one intra picture from what x265amd_intra_encode_frame left: HOST code, no GPU call, HOST pointers.
 * Replaces, for that picture, everything xEncEncode writes (src/encode.cpp:626-662 start codes, SPS, PPS, slice
 * header -- src/bitstream.cpp:112-295 --, xWriteCU per CU :474-585,928 with the residual coding of :270-461 and
 * the CABAC engine of src/bitstream.cpp:526-828, xCabacFlush + xWriteSliceEnd :942-943), with xDefaultParams'
 * settings (src/params.cpp:27-56), SLICE_I, no SEI.  The most probable modes come from mode_y as
 * xEncIntraLoadRef derives them (:1268-1293), the coded-block flags from the levels.  Byte for byte the
 * reference's stream when levels and modes are the reference's (tests: BASELINE configs[0]'s bitstream MD5).
 *   poc: 0 for the first picture of the stream (an IDR slice), then 1, 2, ... (src/encode.cpp:624,660)
 *   mode_c: the real chroma mode, as x265amd_intra_frame.d_mode_c holds it; X265AMD_ERR_ARG if it is not one of
 *     the reference's candidates for the CU's luma mode (src/encode.cpp:781-793)
 *   out / cap: receives the bytes; X265AMD_ERR_BUFFER if cap is too small (*bytes then holds the size needed);
 *     x265amd_intra_stream_bound(width, height) (7 bytes per sample) is enough for any levels the library's own
 *     quantiser produces and for planes of full-range Int16 levels at random (tests/test_intra_stream.py pins
 *     both); a plane built to be hostile that overruns it is refused with X265AMD_ERR_BUFFER, never written past. */
typedef struct x265amd_intra_syntax {
    int32_t width, height, cu, qp, poc;
    int32_t stride_y, stride_c;                    /* of the level planes, in elements */
    const int16_t *level_y, *level_u, *level_v;    /* picture geometry, as d_level_* */
    const uint8_t *mode_y, *mode_c;                /* [ncus], raster order */
} x265amd_intra_syntax;
size_t x265amd_intra_stream_bound(int width, int height);
int x265amd_intra_write_picture(const x265amd_intra_syntax *syntax, uint8_t *out, size_t cap, size_t *bytes);

/* -------------------------------------------------------- transform/quant */

/* One descriptor for the three TQ entry points.  Planes are batched over
 * `nframes`; plane(f) = base + f * <name>_frame_stride (in ELEMENTS of that
 * plane's type).  Unused pointers may be NULL. */
typedef struct x265amd_tq_desc {
    int32_t width, height; /* plane size in pixels; multiples of `tu` */
    int32_t tu;            /* transform size: 4, 8, 16 or 32 (CU = PU = TU like src/encode.cpp:486-489) */
    int32_t qp;            /* 0..51; pass x265amd_chroma_qp(qp) for chroma planes */
    int32_t slice_type;    /* X265AMD_SLICE_*: rounding 171 (I) or 85 (P/B), src/pixel.cpp:602 */
    int32_t nframes;

    const uint8_t *d_src; /* original picture (forward, roundtrip) */
    int32_t src_stride;
    int64_t src_frame_stride;

    /* prediction.  With d_mv == NULL: a plane in picture geometry (intra
     * prediction output / co-located block).  With d_mv != NULL: the REFERENCE
     * picture; the prediction of pixel (x, y) is ref[y + mvy][x + mvx] with the
     * vector of the mv_ctu x mv_ctu block containing (x, y) -- integer-pel motion
     * compensation fused into the load (see mv_shift below).  The vectors must keep every compensated
     * block inside the prediction plane, as the library's own searches guarantee ("legal" candidates,
     * SURVEY App. A.7): the kernels do not clamp, a vector from elsewhere that points outside reads outside --
     * unless `flags` asks for the check (X265AMD_TQ_CHECK_MV below). */
    const uint8_t *d_pred;
    int32_t pred_stride;
    int64_t pred_frame_stride;
    const x265amd_mv *d_mv;
    int32_t mv_ctu;
    int64_t mv_frame_stride;

    /* per-TU intra mode (nMode of xSubDct/xIDctAdd; selects DST for 4x4:
     * src/pixel.cpp:678-680,734-736); NULL = X265AMD_MODE_INVALID everywhere */
    const uint8_t *d_modes;
    int64_t modes_frame_stride;

    int16_t *d_level; /* quantised levels: OUT of forward/roundtrip, IN of inverse */
    int32_t level_stride;
    int64_t level_frame_stride;

    int16_t *d_coef; /* optional OUT of forward: unquantised xSubDct output, same layout as level */

    uint32_t *d_abs_sum; /* optional OUT: xQuant's return value per TU, [ (height/tu) * (width/tu) ] */
    int64_t abs_sum_frame_stride;

    uint8_t *d_rec; /* reconstruction: OUT of inverse/roundtrip */
    int32_t rec_stride;
    int64_t rec_frame_stride;

    /* with d_mv: the vectors are applied as (mvx >> mv_shift, mvy >> mv_shift), arithmetic shift.
     * 0 for the plane the search ran on; 1 for the 4:2:0 chroma planes of the same picture (luma
     * vectors, mv_ctu = half the luma CTU).  0..3. */
    int32_t mv_shift;
    /* X265AMD_TQ_* bits, 0 by default.  X265AMD_TQ_CHECK_MV (with d_mv): before anything else a kernel checks that
     * every vector keeps its mv_ctu x mv_ctu block inside the width x height plane at d_pred; the call waits for it
     * (one stream synchronisation, calls with the flag are serialised per process) and returns X265AMD_ERR_ARG,
     * with nothing transformed, if one does not.  For vectors that did not come from this library's searches.
     * Other bits: X265AMD_ERR_ARG. */
    int32_t flags;
} x265amd_tq_desc;
#define X265AMD_TQ_CHECK_MV 1

/* xSubDct + xQuant for every TU of the plane(s)
 * (inc/x265.h:295,297; src/pixel.cpp:668-722,591-633; call site src/encode.cpp:868-873). */
int x265amd_tq_forward(const x265amd_tq_desc *desc, void *stream);

/* xDeQuant + xIDctAdd for every TU (inc/x265.h:296,298; src/pixel.cpp:635-666,
 * 724-776; call site src/encode.cpp:893-925; a TU whose levels are all zero
 * reconstructs to the prediction exactly as the memcpy branch there does). */
int x265amd_tq_inverse(const x265amd_tq_desc *desc, void *stream);

/* xIDctAdd alone (inc/x265.h:296, src/pixel.cpp:724-776): as x265amd_tq_inverse, but d_level holds
 * DEQUANTISED coefficients (xDeQuant's output) and qp is ignored -- the reconstruction stage on its
 * own, for stage-by-stage checks like the reference's CHECK_TV blocks. */
int x265amd_idct_add(const x265amd_tq_desc *desc, void *stream);

/* forward + inverse in one pass over the picture (the whole "Stage 3a/3b" of
 * src/encode.cpp:863-925): writes level, abs_sum and rec without the
 * coefficient round trip through HBM. */
int x265amd_tq_roundtrip(const x265amd_tq_desc *desc, void *stream);

/* Stand-alone quantiser stages on a coefficient plane, for stage-by-stage
 * checking in the manner of the reference's CHECK_TV blocks
 * (xQuant src/pixel.cpp:591-633, xDeQuant :635-666).  In and out may alias. */
int x265amd_quant_plane(const int16_t *d_coef, int16_t *d_level, int stride, int width, int height,
                        int tu, int qp, int slice_type, uint32_t *d_abs_sum, void *stream);
int x265amd_dequant_plane(const int16_t *d_level, int16_t *d_coef, int stride, int width,
                          int height, int tu, int qp, void *stream);

/* ------------------------------------------------ levels for the trip to the host */

/* Compaction of a plane (or a batch of planes) of quantised levels: what the host's entropy coder walks per transform
 * unit is its coded-block flag and its NON-ZERO levels (xEncodeCoeffNxN, src/encode.cpp:270-461; bCbf from xQuant's
 * sum, :888-890), and the dense Int16 planes are what bounds a host loop over PCIe.  The reference has no such
 * interface (no device boundary: it reads psCoef in place, inc/x265.h:125-131).
 *
 *   d_tu_table [nframes][ntu][2]  per TU (raster order of the plane): index of its first word in d_words, number of words
 *   d_words                       word = (position in the TU's raster, y * tu + x) << 16 | (level & 0xffff); a TU's words
 *                                 are contiguous and in raster order; TUs take their room from *d_cursor in the order they
 *                                 finish (the table says where), so several calls -- Y, U, V -- may share cursor and words
 *   d_cursor                      words asked for so far; x265amd_pack_levels_begin zeroes it.  It keeps counting past
 *                                 `capacity`: words at or beyond it are not written (a TU whose entry ends beyond it is
 *                                 incomplete), and cursor > capacity tells the host to take the dense plane instead
 *   d_abs_sum                     optional: xQuant's per-TU sums as the transform entry points leave them; a TU whose sum
 *                                 is 0 is entered as empty without being read
 * d_tu_table and d_words may be page-locked host memory (x265amd_malloc_host): the words then cross the link as
 * they are written and there is nothing to copy afterwards (d_cursor stays in device memory: it is an atomic);
 * ordinary (pageable) host memory is refused with X265AMD_ERR_ARG.
 * Any stride works; planes whose rows start on 16-byte boundaries (d_level, 2 * stride, 2 * frame_stride multiples
 * of 16) take the fast path with one 16-byte load per lane. */
int x265amd_pack_levels_begin(uint32_t *d_cursor, void *stream);
int x265amd_pack_levels(const int16_t *d_level, int width, int height, int stride, int tu, int nframes,
                        int64_t frame_stride, const uint32_t *d_abs_sum, int64_t abs_sum_frame_stride,
                        uint32_t *d_tu_table, uint32_t *d_words, uint32_t capacity, uint32_t *d_cursor, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_H */
