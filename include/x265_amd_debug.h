/*
 * x265_amd_debug.h -- instrumentation hooks of libx265amd.so.  NOT part of the drop-in ABI
 * (include/x265_amd.h): tools/, bench.py and the tests use them to time kernels, count the work of the
 * elimination search, try kernel shapes and reach the cross-check formulation of the search.
 *
 * The switches are PER DEVICE, like all of the library's state: a setter changes the device that is current on the
 * calling thread (after x265amd_init for it; before that it does nothing), and only calls made on that device see
 * the change.  They are not synchronised with launches in flight on other host threads of the same device: set them
 * while that device is quiet.  Nothing in include/x265_amd.h depends on them.  The reference has no counterpart
 * (its only instrumentation is the CHECK_TV harness, src/testvec.cpp).
 */
#ifndef X265_AMD_DEBUG_H
#define X265_AMD_DEBUG_H

#include "x265_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* d_stats != NULL: the successive-elimination kernels run their statistics variants and write 16 u32 per
 * (plane, CTU): [0] / [1] groups of four candidates evaluated in the first batch / the second pass, [2..6]
 * cycle stamps of the exact kernel, [8] the work estimate, [10..13] cycle stamps of the bounds kernel.
 * [14] / [15] lower halves finished of the second pass / the first batch.  NULL switches back.  Applies to every
 * later x265amd_me_fullsearch_sea[_static] call on this device. */
void x265amd_debug_set_me_stats(uint32_t *d_stats);

/* HIP events around the three kernels of x265amd_me_fullsearch_sea on the call's own stream; _last_ms
 * waits for the last timed call and returns {box16_pairs, 0, sea_bounds_kernel, sea_exact_kernel + me_list_qsad} in ms. */
int x265amd_debug_me_timing(int enable);
int x265amd_debug_me_last_ms(float *out4);

/* shape experiments of sea_exact_kernel: extra dynamic LDS per workgroup (lowers occupancy) and a forced
 * wave count (8 or 16; 0 = the library's choice) */
void x265amd_debug_sea_shape(int lds_pad_bytes, int waves);

/* transform / quantisation path: 0 = the library's choice (integer matrix products, tq_mfma.hip, wherever one
 * matrix serves the whole plane), 1 = always the LDS-resident partial-butterfly kernel (tq_kernels.hip) -- the
 * two are independent formulations of the same arithmetic and must agree bit for bit */
void x265amd_debug_tq_path(int path);

/* x265amd_intra_encode_frame(s): 1 (default) = an anti-diagonal step is ONE launch, the transform stage of a CU runs at
 * the end of its prediction workgroup (intra_wave_tq_kernel); 0 = prediction kernel and transform stage as two
 * launches.  Same output either way (A/B timing, parity tests). */
void x265amd_debug_intra_fused(int on);
/* .. threads per workgroup of that kernel: 1024 (16 waves, one workgroup to a compute unit: the shape for a step with
 * few CUs), 512 (8 waves, two to a compute unit) or 256 (4 waves, four to a compute unit): a batch's steps; 0 (default)
 * = by the number of CUs in the step */
void x265amd_debug_intra_threads(int threads);
/* d_stats != NULL: the fused intra kernel (CU 32) leaves 16 u32 per (frame of the batch, CU): shader-clock stamps [0] start,
 * [1] source and reference samples in LDS, [2] smoothing / DC / most probable modes, [3] the 35 luma and 5 chroma SADs, [4] luma
 * argmin, [5] luma prediction stored (and the chroma SAD of the luma mode), [6] chroma argmin, [7] chroma predictions stored,
 * [8] the barrier before the transform stage, [9] its luma wave done.  NULL switches back.  tools/intra_stats.py reads them. */
void x265amd_debug_set_intra_stats(uint32_t *d_stats);

/* Where and when the waves of the matrix-core TQ kernel ran: with a non-null device buffer of nwaves x 4
 * 64-bit words (nwaves >= 4 x the workgroups of the launch, else nothing is recorded; zeroed by the caller) every wave of the following
 * launches of that kernel leaves {start, end (100 MHz wall clock), shader clocks spent,
 * XCC_ID << 32 | HW_ID} at record 4 * workgroup + wave.  Null switches it off.  Per device like the hooks
 * above; tools/tq_trace.py prints the launch ramp, the spread of the durations and the clock from it. */
void x265amd_debug_tq_trace(unsigned long long *d_records, int nwaves);

/* Tile units (4 neighbouring 32x32 tiles) per workgroup of the matrix-core TQ kernel: 0 (default) = an even
 * split over the workgroups the chip holds at once; n > 0 = n units per workgroup.  For tuning runs. */
void x265amd_debug_tq_units_per_wg(int n);

/* x265amd_me_fullsearch in its independent one-candidate-per-lane formulation (v_alignbyte_b32 + v_sad_u8,
 * current block in LDS): the cross-check of the production kernels, 1.0-1.6 Gpix/s */
int x265amd_debug_me_fullsearch_basic(const uint8_t *d_cur, const uint8_t *d_ref, int width, int height, int stride,
                                      int ctu, int range, int nframes, int64_t cur_frame_stride,
                                      int64_t ref_frame_stride, int nrefs, int64_t ref_ref_stride, x265amd_mv *d_out,
                                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_DEBUG_H */
