// intra_kernels.hip -- the intra luma mode decision of a batch of CUs (first slice):
// xEncIntraPredLuma for all 35 modes (src/encode.cpp:1617-1652 over xPredIntraPlanar :1353, xPredIntraDc :1418,
// xPredIntraAng :1446), the SAD of each prediction against the source CU and the cost / argmin of
// src/encode.cpp:725-765.  Input per CU = what xEncIntraLoadRef (src/encode.cpp:1163) leaves in the encoder's
// cache: the padded reference samples pucPixRef[0] and the three most probable modes; the [1 2 1] smoothing of the
// samples (:1260-1265) is done here.  The CUs of one launch are independent (in the encoder: the CUs of one
// anti-diagonal, whose neighbours are reconstructed).
//
// One workgroup per CU.  The samples and their smoothed copy sit in LDS; a prediction is never stored: every
// lane evaluates ITS pixels of a mode in closed form (planar and angular are written without the reference's
// running sums and without its transpose: pixel (r, c) of a horizontal mode is pixel (c, r) of the mirrored
// vertical construction), takes |source - prediction| and the wave adds up.  The 35 modes are dealt round the
// four waves; thread 0 does the argmin in mode order.  On request the winner's prediction is written out for the
// residual stage (src/encode.cpp:867).
#include "common.h"
#include "tq_common.h"
#include "tq_mfma_body.h"

namespace x265amd {

// src/table.cpp:32-39, 49-56, 59-66.  The tables as they stand there, and what the kernels use instead: the same values
// out of packed constants by shifts (a mode is wave-uniform nearly everywhere, so this is scalar arithmetic; indexing
// a table in memory cost a vector load per task of the SAD phase, waited for before anything else could start).
constexpr uint8_t kIntraFilterTab[4][35] = {
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1},
    {1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1}};
constexpr int8_t kIntraAngleTab[35] = {0,   0,   32,  26,  21,  17,  13,  9,  5,  2,  0,  -2, -5, -9, -13, -17, -21, -26,
                                       -32, -26, -21, -17, -13, -9,  -5,  -2, 0,  2,  5,  9,  13, 17, 21,  26,  32};
constexpr int16_t kIntraInvAngleTab[35] = {0,   0,   256, 315, 390, 482, 630, 910, 1638, 4096, 0,   4096, 1638, 910, 630, 482, 390, 315,
                                           256, 315, 390, 482, 630, 910, 1638, 4096, 0,  4096, 1638, 910, 630, 482, 390, 315, 256};

// |angle| by distance from the pure horizontal / vertical mode, 6 bits each; the inverse angles of distances 1..4 and 5..8,
// 13 bits each; the filter flags of a block size as one bit per mode
constexpr uint64_t kAbsAngles = 0ull | (2ull << 6) | (5ull << 12) | (9ull << 18) | (13ull << 24) | (17ull << 30) | (21ull << 36) | (26ull << 42) | (32ull << 48);
constexpr uint64_t kInvLo = 4096ull | (1638ull << 13) | (910ull << 26) | (630ull << 39);
constexpr uint64_t kInvHi = 482ull | (390ull << 13) | (315ull << 26) | (256ull << 39);
constexpr uint64_t intra_filter_mask(int size_idx)
{
    uint64_t m = 0;
    for (int i = 0; i < 35; i++) m |= (uint64_t)kIntraFilterTab[size_idx][i] << i;
    return m;
}
constexpr uint64_t kFilterMask[4] = {intra_filter_mask(0), intra_filter_mask(1), intra_filter_mask(2), intra_filter_mask(3)};

__host__ __device__ constexpr int intra_angle(int mode)
{
    if (mode < 2) return 0;
    const int d = mode < 18 ? mode - 10 : mode - 26;
    const int ad = d < 0 ? -d : d;
    const int a = (int)((kAbsAngles >> (6 * ad)) & 63u);
    return ((d < 0) == (mode < 18)) ? a : -a;
}
__host__ __device__ constexpr int intra_inv_angle(int mode)
{
    if (mode < 2) return 0;
    const int d = mode < 18 ? mode - 10 : mode - 26;
    const int ad = d < 0 ? -d : d;
    if (ad == 0) return 0;
    return (int)(((ad <= 4 ? kInvLo >> (13 * (ad - 1)) : kInvHi >> (13 * (ad - 5)))) & 8191u);
}
// size_idx = log2(N) - 2
template <int SIZE_IDX> __host__ __device__ constexpr int intra_filter(int mode)
{
    constexpr uint64_t m = kFilterMask[SIZE_IDX];
    return (int)((m >> mode) & 1u);
}
constexpr bool intra_tables_agree()
{
    for (int m = 0; m < 35; m++) {
        if (intra_angle(m) != kIntraAngleTab[m] || intra_inv_angle(m) != kIntraInvAngleTab[m]) return false;
        if (intra_filter<0>(m) != kIntraFilterTab[0][m] || intra_filter<1>(m) != kIntraFilterTab[1][m] ||
            intra_filter<2>(m) != kIntraFilterTab[2][m] || intra_filter<3>(m) != kIntraFilterTab[3][m])
            return false;
    }
    return true;
}
static_assert(intra_tables_agree(), "packed intra tables differ from src/table.cpp");

// |a - b| + acc in one instruction (v_sad_u32)
__device__ __forceinline__ uint32_t sad_acc(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}

// Sum / minimum over the 64 lanes of a wave, the same value in every lane's scalar copy: four DPP steps inside the rows of
// 16 lanes (pairs, quads, half rows, rows), then the four row results through v_readlane.  All 64 lanes must be active.
// (__shfl_xor goes through the LDS crossbar: six dependent trips of it per reduction were a fifth of the SAD phase.)
template <int CTRL> __device__ __forceinline__ uint32_t intra_dpp(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t s)
{
    s += intra_dpp<0xb1>(s);   // quad_perm [1,0,3,2]
    s += intra_dpp<0x4e>(s);   // quad_perm [2,3,0,1]
    s += intra_dpp<0x141>(s);  // row_half_mirror
    s += intra_dpp<0x140>(s);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)s, 0) + (uint32_t)__builtin_amdgcn_readlane((int)s, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)s, 32) + (uint32_t)__builtin_amdgcn_readlane((int)s, 48);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t s)
{
    s = min(s, intra_dpp<0xb1>(s));
    s = min(s, intra_dpp<0x4e>(s));
    s = min(s, intra_dpp<0x141>(s));
    s = min(s, intra_dpp<0x140>(s));
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)s, 0), (uint32_t)__builtin_amdgcn_readlane((int)s, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)s, 32), (uint32_t)__builtin_amdgcn_readlane((int)s, 48)));
}

constexpr int kMaxRef = 4 * 32 + 1;
constexpr int kMainRow = 3 * 32 + 8;
constexpr int kRefRow = kMaxRef + 7;  // a sample array in the frame kernels' LDS: 4-byte multiples, room for the last quad's window  // refMain of a mode: samples -N .. 2N, and one past the end that a zero weight may touch

// pixel (r, c) of mode `mode`; ref = the sample array the mode reads (smoothed or not), dc = the DC value.
// LUMA: the edge smoothing of DC and of the pure horizontal / vertical modes (bLuma of the reference's functions)
template <int N, bool LUMA = true> __device__ __forceinline__ int intra_pixel(const uint8_t *ref, int mode, int r, int c, int dc)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    const uint8_t *tl = ref + 2 * N;  // the corner; tl[1 + j] = above column j, tl[-1 - i] = left of row i
    if (mode == 0) {  // planar (src/encode.cpp:1353-1396)
        const int left = tl[-1 - r], top = tl[1 + c], bl = tl[-1 - N], tr = tl[1 + N];
        const int hp = (left << L2) + N + (c + 1) * (tr - left);
        const int tp = (top << L2) + (r + 1) * (bl - top);
        return ((hp + tp) >> (L2 + 1)) & 0xff;
    }
    if (mode == 1) {  // DC with its edge smoothing (src/encode.cpp:1418-1444)
        if (LUMA && r == 0 && c == 0) return (tl[1] + tl[-1] + 2 * dc + 2) >> 2;
        if (LUMA && r == 0) return (tl[1 + c] + 3 * dc + 2) >> 2;
        if (LUMA && c == 0) return (tl[-1 - r] + 3 * dc + 2) >> 2;
        return dc;
    }
    // angular (src/encode.cpp:1446-1534): the main side is the left column for modes 2..17 (built as a row and
    // transposed at the end: swap the roles of r and c instead), the row above for 18..34
    const bool hor = mode < 18;
    const int ang = intra_angle(mode), inv = intra_inv_angle(mode);
    const int k = hor ? c : r, x = hor ? r : c;
    const int sgn = hor ? -1 : 1;  // main side: tl[sgn * i]; the other side, projected onto it: tl[-sgn * offset]
    auto main_ref = [&](int i) -> int {
        if (i >= 0) return tl[sgn * i];
        const int off = (128 + (-i) * inv) >> 8;
        return tl[-sgn * off];
    };
    const int pos = (k + 1) * ang;
    const int idx = pos >> 5, fact = pos & 31;
    int p = main_ref(x + idx + 1);
    if (fact) p = ((32 - fact) * p + fact * main_ref(x + idx + 2) + 16) >> 5;
    if (LUMA && ang == 0 && x == 0) {  // modes 10 and 26: the first column / row follows the other side's gradient
        p += (tl[-sgn * (k + 1)] - tl[0]) >> 1;
        p = min(max(p, 0), 255);
    }
    return p;
}

// The main reference row of an angular mode, the reference's refMain (src/encode.cpp:1461-1485), in LDS: entry
// i + N holds sample i, i in ((N * angle) >> 5, 2 N]: the main side, and for negative i the other side projected
// onto it.  One wave fills its own row (two or three entries a lane) and reads it back: same-wave LDS accesses
// complete in order, the fence keeps the compiler from moving the reads up.
template <int N> __device__ __forceinline__ void intra_fill_main(const uint8_t *ref, int mode, int lane, uint8_t *row)
{
    const uint8_t *tl = ref + 2 * N;
    const bool hor = mode < 18;
    const int ang = intra_angle(mode), inv = intra_inv_angle(mode);
    const int sgn = hor ? -1 : 1;
    const int lo = ang < 0 ? ((N * ang) >> 5) + 1 : 0;
    for (int i = lane - N; i <= 2 * N; i += 64) {
        int v = 0;
        if (i >= 0)
            v = tl[sgn * i];
        else if (i >= lo)
            v = tl[-sgn * ((128 + (-i) * inv) >> 8)];
        row[i + N] = (uint8_t)v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The same row for the frame kernels, which keep every sample array a second time in reverse (rev[j] = ref[4 N - j]):
// R = the array in the order the mode walks it -- ref for the modes from 18 on, rev below 18 -- so that, with T = R + 2 N,
// sample i of the main side is T[i] and the other side projected onto it is T[-offset], whatever the direction.  A mode
// with a positive angle reads no projected sample: its row IS the array (returned as it lies, nothing is copied).  A
// negative angle reads entries lo .. N only, at most 2 N of them: one store per lane, no loop.  Returns the row
// (row[N + i] = sample i), 4-byte aligned like R and `row`.
template <int N> __device__ __forceinline__ const uint8_t *intra_main_row(const uint8_t *R, int mode, int lane, uint8_t *row)
{
    const int ang = intra_angle(mode);
    if (ang >= 0) return R + N;  // wave-uniform
    const int inv = intra_inv_angle(mode);
    const int lo = ((N * ang) >> 5) + 1;
    const int i = max(lane - (N - 1), lo);  // lanes below lo write entry lo once more (the same value)
    if (lane < 2 * N) {
        const int idx = i >= 0 ? i : -(int)((128u + __umul24((unsigned)(-i), (unsigned)inv)) >> 8);
        row[N + i] = R[2 * N + idx];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return row;
}
// pixel (r, c) of angular mode `mode` from its main row (see intra_pixel for the construction)
template <int N, bool LUMA> __device__ __forceinline__ int intra_ang_pixel(const uint8_t *ref, const uint8_t *row, int mode, int r, int c)
{
    const bool hor = mode < 18;
    const int ang = intra_angle(mode);
    const int k = hor ? c : r, x = hor ? r : c;
    const int pos = (k + 1) * ang;
    const int idx = pos >> 5, fact = pos & 31;
    const uint8_t *m = row + N + x + idx + 1;
    int p = ((32 - fact) * m[0] + fact * m[1] + 16) >> 5;  // fact == 0: (32 m[0] + 16) >> 5 = m[0]; m[1] is a valid entry (kMainRow)
    if (LUMA && ang == 0 && x == 0) {
        const uint8_t *tl = ref + 2 * N;
        p += (tl[(hor ? 1 : -1) * (k + 1)] - tl[0]) >> 1;
        p = min(max(p, 0), 255);
    }
    return p;
}

// Four neighbouring pixels of an angular mode at once.  Along the main index x (the column for the modes from 18 on, the
// row below 18) the pixels x0 .. x0 + 3 of line k all have the same position idx / weight fact, so they are
//   ((32 - fact) * m[a .. a + 3] + fact * m[a + 1 .. a + 4] + 16) >> 5,   a = N + x0 + idx + 1 in the mode's main row:
// two aligned dwords of the row, two 64-bit shifts for the byte windows, even and odd bytes as packed u16 pairs (the
// sums stay below 2^14), the four results back in one dword for v_sad_u8 against four source pixels.  fact = 0 gives
// m[a ..] itself, as the reference's unweighted branch does.  (Modes 10 and 26: the caller smooths the first pixel.)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t intra_ang_quad(const uint8_t *row, int a, int fact)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(row + (a & ~3));  // row is 4-byte aligned
    const uint64_t win = ((uint64_t)w[1] << 32) | w[0];
    const int sh = 8 * (a & 3);
    const uint32_t A = (uint32_t)(win >> sh), Bq = (uint32_t)(win >> (sh + 8));
    const uint32_t fp = (uint32_t)fact * 0x00010001u, gp = 0x00200020u - fp;
    auto blend = [&](uint32_t x, uint32_t y) -> uint32_t {  // two multiply-adds: y * f + 16, then x * (32 - f) + that
        const u16x2 t = __builtin_bit_cast(u16x2, y) * __builtin_bit_cast(u16x2, fp) + __builtin_bit_cast(u16x2, 0x00100010u);
        const u16x2 r = (__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, gp) + t) >> (unsigned short)5;
        return __builtin_bit_cast(uint32_t, r);
    };
    // bytes 0, 2 and bytes 1, 3 of a dword as zero-extended u16 pairs: one v_perm_b32 each (selector 0x0c = a zero byte)
    const uint32_t even = blend(__builtin_amdgcn_perm(0u, A, 0x0c020c00u), __builtin_amdgcn_perm(0u, Bq, 0x0c020c00u));
    const uint32_t odd = blend(__builtin_amdgcn_perm(0u, A, 0x0c030c01u), __builtin_amdgcn_perm(0u, Bq, 0x0c030c01u));
    return even | (odd << 8);
}

// Four neighbouring pixels of the planar mode (src/encode.cpp:1353-1396): row k, columns x0 .. x0 + 3; top[j] = the sample
// above column x0 + j.  ((left << L2) + N + (c + 1)(tr - left) + (top << L2) + (r + 1)(bl - top)) >> (L2 + 1)
//   = (base + j dh + top_j (N - 1 - r)) >> (L2 + 1),  base = (left << L2) + N + (r + 1) bl + (x0 + 1) dh,  dh = tr - left.
template <int N> __device__ __forceinline__ uint32_t intra_planar_quad(const uint8_t *tl, int k, int x0, const int (&top)[4], int tr, int bl)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    const int left = tl[-1 - k], dh = tr - left, wn = N - 1 - k;
    const int base = (left << L2) + N + __mul24(k + 1, bl) + __mul24(x0 + 1, dh);
    uint32_t pq = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) pq |= (uint32_t)(((__mul24(top[j], wn) + base + j * dh) >> (L2 + 1)) & 0xff) << (8 * j);
    return pq;
}

// ---- pieces shared by the two kernels (one workgroup per CU) ----

// the DC value of a sample array (src/encode.cpp:1398-1416), by wave 0; the caller puts a barrier after it
template <int N> __device__ __forceinline__ void intra_dc_value(const uint8_t *ref, int lane, int *out)
{
    int s = 0;
    for (int i = lane; i < 2 * N; i += 64) s += i < N ? ref[2 * N + 1 + i] : ref[2 * N - 1 - (i - N)];
    s = (int)wave_sum_u32((uint32_t)s);
    if (lane == 0) *out = (s + N) / (2 * N);
}

// SADs of the 35 luma modes (dealt round the workgroup's waves) into s_sad[]; s_ref = samples and smoothed samples, sp[] =
// the lane's source pixels i = lane + 64 t.  The caller puts a barrier after it.
template <int N, int PER, int NWAVES = 4>
__device__ __forceinline__ void intra_luma_sads(const uint8_t (*s_ref)[kMaxRef + 3], const int (&sp)[PER], int dc, int wave, int lane,
                                                uint32_t *s_sad, uint8_t *main_row)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    for (int mode = wave; mode < 35; mode += NWAVES) {
        const uint8_t *ref = s_ref[intra_filter<L2 - 2>(mode)];
        uint32_t s = 0;
        if (mode >= 2) {  // wave-uniform
            intra_fill_main<N>(ref, mode, lane, main_row);
#pragma unroll
            for (int t = 0; t < PER; t++) {
                const int i = lane + 64 * t;
                if (i < N * N) s = sad_acc((uint32_t)sp[t], (uint32_t)intra_ang_pixel<N, true>(ref, main_row, mode, i >> L2, i & (N - 1)), s);
            }
        } else {
#pragma unroll
            for (int t = 0; t < PER; t++) {
                const int i = lane + 64 * t;
                if (i < N * N) s = sad_acc((uint32_t)sp[t], (uint32_t)intra_pixel<N>(ref, mode, i >> L2, i & (N - 1), dc), s);
            }
        }
        s = wave_sum_u32(s);
        if (lane == 0) s_sad[mode] = s;
    }
}

// The same for the frame kernel's 16 / 32 wide CUs, four pixels a step (intra_ang_quad): srcv[t] = the source pixels of
// quad q = lane + 64 t read along rows (line k = q / (N / 4), x0 = 4 (q % (N / 4)): row k, columns x0 ..), srch[t] = the
// same quad of the TRANSPOSED source (column k, rows x0 ..) -- a mode below 18 is the mirrored construction, so it is
// compared against the transposed block.  Planar and DC are put together a quad at a time too.
template <int N, int PER>
__device__ __forceinline__ uint32_t intra_luma_sad_quads(const uint8_t (*s_ref)[kRefRow], const uint8_t (*s_rev)[kRefRow],
                                                         const uint32_t (&srcv)[PER / 4], const uint32_t (&srch)[PER / 4], int dc, int mode,
                                                         int lane, uint8_t *row_buf)
{
    constexpr int L2 = N == 16 ? 4 : 5;
    constexpr int QL = N / 4;  // quads per line
    const int filt = intra_filter<L2 - 2>(mode);
    const uint8_t *ref = s_ref[filt];
    uint32_t s = 0;
    if (mode >= 2) {  // wave-uniform
        const int ang = intra_angle(mode);
        const bool hor = mode < 18;
        const uint8_t *main_row = intra_main_row<N>(hor ? s_rev[filt] : ref, mode, lane, row_buf);
        if (ang != 0) {  // wave-uniform
#pragma unroll
            for (int t = 0; t < PER / 4; t++) {
                const int q = lane + 64 * t, k = q / QL, x0 = 4 * (q % QL);
                const int pos = __mul24(k + 1, ang);
                s = __builtin_amdgcn_sad_u8(intra_ang_quad(main_row, N + x0 + (pos >> 5) + 1, pos & 31), hor ? srch[t] : srcv[t], s);
            }
        } else {  // modes 10 and 26: every line is the main side itself; its first pixel follows the other side's gradient
            const uint8_t *tl = ref + 2 * N;
#pragma unroll
            for (int t = 0; t < PER / 4; t++) {
                const int q = lane + 64 * t, k = q / QL, x0 = 4 * (q % QL);
                uint32_t pq = intra_ang_quad(main_row, N + x0 + 1, 0);
                if (x0 == 0) {
                    const int p = (int)(pq & 0xffu) + ((tl[(hor ? 1 : -1) * (k + 1)] - tl[0]) >> 1);
                    pq = (pq & 0xffffff00u) | (uint32_t)min(max(p, 0), 255);
                }
                s = __builtin_amdgcn_sad_u8(pq, hor ? srch[t] : srcv[t], s);
            }
        }
    } else if (mode == 1) {
        // DC: the value everywhere but on the first row and column, which lean on the samples beside them
        // (src/encode.cpp:1418-1444) -- a quad is four copies of dc, or, in row 0 / for the first pixel of a row, smoothed
        const uint8_t *tl = ref + 2 * N;
        const uint32_t dc4 = (uint32_t)dc * 0x01010101u;
#pragma unroll
        for (int t = 0; t < PER / 4; t++) {
            const int q = lane + 64 * t, k = q / QL, x0 = 4 * (q % QL);
            uint32_t pq = dc4;
            if (k == 0) {
                pq = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) pq |= (uint32_t)((tl[1 + x0 + j] + 3 * dc + 2) >> 2) << (8 * j);
                if (x0 == 0) pq = (pq & 0xffffff00u) | (uint32_t)((tl[1] + tl[-1] + 2 * dc + 2) >> 2);
            } else if (x0 == 0) {
                pq = (pq & 0xffffff00u) | (uint32_t)((tl[-1 - k] + 3 * dc + 2) >> 2);
            }
            s = __builtin_amdgcn_sad_u8(pq, srcv[t], s);
        }
    } else {
        // planar: a lane's quads all lie in the same four columns (64 is a multiple of the quads per line)
        const uint8_t *tl = ref + 2 * N;
        const int tr = tl[1 + N], bl = tl[-1 - N];
        const int x0 = 4 * (lane % QL);
        int top[4];
#pragma unroll
        for (int j = 0; j < 4; j++) top[j] = tl[1 + x0 + j];
#pragma unroll
        for (int t = 0; t < PER / 4; t++)
            s = __builtin_amdgcn_sad_u8(intra_planar_quad<N>(tl, lane / QL + (64 / QL) * t, x0, top, tr, bl), srcv[t], s);
    }
    return wave_sum_u32(s);
}

// cost and argmin exactly as src/encode.cpp:725-765: first strict minimum in mode order from MAX_SAD (one thread)
__device__ __forceinline__ x265amd_intra_best intra_luma_argmin(const uint32_t *s_sad, int m0, int m1, int m2, uint32_t lambda)
{
    uint32_t bc = X265AMD_MAX_SAD, bm = 0;
    for (int m = 0; m < 35; m++) {
        const uint32_t cost = s_sad[m] + lambda * (m == m0 ? 1u : (m == m1 || m == m2) ? 2u : 3u);
        if (cost < bc) {
            bc = cost;
            bm = (uint32_t)m;
        }
    }
    return x265amd_intra_best{bm, bc};
}

// The same by a whole wave (all 64 lanes active): lane m < 35 holds mode m's cost, the minimum of (cost << 6 | mode) is
// the first strict minimum in mode order; every lane gets the answer.  cost < 2^26 (SAD <= 2^18, lambda <= 2^20).
__device__ __forceinline__ x265amd_intra_best intra_luma_argmin_wave(const uint32_t *s_sad, int m0, int m1, int m2, uint32_t lambda, int lane)
{
    uint32_t key = 0xffffffffu;
    if (lane < 35) {
        const uint32_t cost = s_sad[lane] + lambda * (lane == m0 ? 1u : (lane == m1 || lane == m2) ? 2u : 3u);
        if (cost < X265AMD_MAX_SAD) key = (cost << 6) | (uint32_t)lane;
    }
    key = wave_min_u32(key);
    if (key == 0xffffffffu) return x265amd_intra_best{0u, X265AMD_MAX_SAD};
    return x265amd_intra_best{key & 63u, key >> 6};
}

template <int N>
__global__ __launch_bounds__(256) void intra_decide_kernel(const uint8_t *__restrict__ src, int src_stride, int width, int height,
                                                           const int32_t *__restrict__ xy, const uint8_t *__restrict__ refs,
                                                           int ref_stride, const uint8_t *__restrict__ mpms, uint32_t lambda,
                                                           x265amd_intra_best *__restrict__ best, uint32_t *__restrict__ sads,
                                                           uint8_t *__restrict__ pred)
{
    constexpr int L2 = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    constexpr int NPIX = N * N;
    constexpr int PER = (NPIX + 63) / 64;  // pixels of a mode per lane
    __shared__ uint8_t s_ref[2][kMaxRef + 3];
    __shared__ uint8_t s_main[4][kMainRow];
    __shared__ uint32_t s_sad[35];
    __shared__ int s_dc, s_best;
    const int cu = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = xy[2 * cu], y0 = xy[2 * cu + 1];
    if (x0 < 0 || y0 < 0 || x0 + N > width || y0 + N > height) {  // not a CU of this plane: say so, touch nothing
        if (tid == 0) best[cu] = x265amd_intra_best{0xffffffffu, 0xffffffffu};
        return;
    }
    const uint8_t *ref0 = refs + (int64_t)cu * ref_stride;
    for (int i = tid; i <= 4 * N; i += 256) {
        const int a = ref0[i];
        s_ref[0][i] = (uint8_t)a;
        // [1 2 1], the two ends kept (src/encode.cpp:1260-1265)
        s_ref[1][i] = (i == 0 || i == 4 * N) ? (uint8_t)a : (uint8_t)((ref0[i - 1] + 2 * a + ref0[i + 1] + 2) >> 2);
    }
    __syncthreads();
    if (wave == 0) intra_dc_value<N>(s_ref[0], lane, &s_dc);  // mode 1 reads the unsmoothed samples at every size
    // the lane's pixels of the source CU: pixel i = lane + 64 t, raster order
    int sp[PER];
#pragma unroll
    for (int t = 0; t < PER; t++) {
        const int i = lane + 64 * t;
        sp[t] = i < NPIX ? src[(int64_t)(y0 + (i >> L2)) * src_stride + x0 + (i & (N - 1))] : 0;
    }
    __syncthreads();
    const int dc = s_dc;
    intra_luma_sads<N, PER>(s_ref, sp, dc, wave, lane, s_sad, s_main[wave]);
    __syncthreads();
    if (tid == 0) {
        const uint8_t *mpm = mpms + 3 * (int64_t)cu;
        const x265amd_intra_best b = intra_luma_argmin(s_sad, mpm[0], mpm[1], mpm[2], lambda);
        best[cu] = b;
        s_best = (int)b.mode;
    }
    if (sads && tid < 35) sads[35 * (int64_t)cu + tid] = s_sad[tid];
    if (pred) {  // workgroup-uniform
        __syncthreads();
        const int bm = s_best;
        const uint8_t *ref = s_ref[intra_filter<L2 - 2>(bm)];
        for (int i = tid; i < NPIX; i += 256)
            pred[(int64_t)cu * NPIX + i] = (uint8_t)intra_pixel<N>(ref, bm, i >> L2, i & (N - 1), dc);
    }
}

// ---------------------------------------------------------------------------------------------
// The prediction stage of the intra encoder for the CUs of ONE anti-diagonal of a frame (Stage 1b .. 2 of
// src/encode.cpp:719-861 and the two prediction calls of Stage 3a, :867 and :876): reference samples from the
// reconstruction written so far (xEncIntraLoadRef, :1163-1293, with xPaddingRef :1133), most probable modes, the
// luma decision, the chroma decision over {planar, vertical, horizontal, DC, the luma mode} (:781-853; LM is
// skipped there), and the winners' predictions into the prediction planes for the transform stage.
//
// Which neighbours exist follows from the position alone in this encoder (CU = PU = TU, raster order): left iff
// x > 0, above iff y > 0, above-right iff y > 0 and the CU is not the last of its row, below-left never (the
// line buffer of modes is reset per row and written for one CU height, :989-994, :1107-1111).  CU (x, y) needs
// (x - 1, y), (x, y - 1) and (x + 1, y - 1): all CUs with the same x + 2 y are independent.
// ---------------------------------------------------------------------------------------------
struct IntraWave {
    x265amd_intra_frame f;
    int ncx, ncy;
    int d;      // the anti-diagonal: CUs with cux + 2 cuy == d
    int cy_lo;  // smallest cuy on it
    uint32_t lambda;
    int64_t fs_y, fs_c, fs_cu;  // a batch of independent frames (blockIdx.y): frame to frame, in elements, of the luma
                                // planes, the chroma planes and the per-CU arrays
    uint32_t *stats;            // x265amd_debug_set_intra_stats: 16 cycle stamps per (frame, CU), or null
};
// stamp i of this workgroup's CU (thread 0; block-uniform test)
#define X265AMD_INTRA_STAMP(i) \
    if (w.stats && threadIdx.x == 0) w.stats[((int64_t)blockIdx.y * w.ncx * w.ncy + cu) * 16 + (i)] = (uint32_t)clock64()


// sample i of the reference array of an n x n block at (x0, y0) of plane `rec`: xEncIntraLoadRef + xPaddingRef
__device__ __forceinline__ int intra_ref_sample(const uint8_t *rec, int stride, int x0, int y0, int n, int i, bool bL, bool bT, bool bTR)
{
    // blocks of the array: 0 below-left [0, n), 1 left [n, 2n), 2 corner {2n}, 3 above [2n+1, 3n+1), 4 above-right [3n+1, 4n+1)
    const bool valid[5] = {false, bL, bL && bT, bT, bTR};
    const int off[6] = {0, n, 2 * n, 2 * n + 1, 3 * n + 1, 4 * n + 1};
    if (!(bL || bT || bTR)) return 0x80;  // :1203-1209
    auto raw = [&](int j) -> int {  // sample j of a block that exists
        if (j < 2 * n) return rec[(int64_t)(y0 + 2 * n - 1 - j) * stride + x0 - 1];  // the left column, bottom-up
        if (j == 2 * n) return rec[(int64_t)(y0 - 1) * stride + x0 - 1];
        return rec[(int64_t)(y0 - 1) * stride + x0 + (j - 2 * n - 1)];
    };
    const int b = i < n ? 0 : i < 2 * n ? 1 : i == 2 * n ? 2 : i < 3 * n + 1 ? 3 : 4;
    if (valid[b]) return raw(i);
    int first = 0;
    while (!valid[first]) first++;
    if (b < first) return raw(off[first]);  // everything before the first block that exists: its first sample
    int m = b - 1;
    while (!valid[m]) m--;                  // a gap after it: the last sample before the gap
    return raw(off[m + 1] - 1);
}

// 16 waves per CU here: an anti-diagonal has few CUs (34 at 1080p with CU 32), so the launch is as long as one
// CU takes, and the 35 + 5 candidate modes of a CU are what there is to spread.  A batch of frames brings more CUs per
// step than the chip has compute units: then 8 or 4 waves per CU, two or four workgroups to a compute unit (the
// registers allow 16 waves there either way), so that one workgroup's serial stretches -- sample loads, the argmin
// threads, the transform stage's three waves -- run beside the others' SADs.
// the body of the prediction kernel of a step; false: this workgroup has no CU (block-uniform)
template <int N, int NT>
__device__ __forceinline__ bool intra_wave_body(const IntraWave &w)
{
    constexpr int NWAVES = NT / 64;
    constexpr int L2 = N == 16 ? 4 : 5;
    constexpr int NC = N / 2, L2C = L2 - 1;
    constexpr int NPIX = N * N, PER = NPIX / 64;
    constexpr int NPIXC = NC * NC;
    x265amd_intra_frame f = w.f;
    {   // this workgroup's frame of the batch (block-uniform: scalar address arithmetic)
        const int64_t oy = (int64_t)blockIdx.y * w.fs_y, oc = (int64_t)blockIdx.y * w.fs_c, ou = (int64_t)blockIdx.y * w.fs_cu;
        f.d_src_y += oy, f.d_rec_y += oy, f.d_pred_y += oy;
        f.d_src_u += oc, f.d_src_v += oc, f.d_rec_u += oc, f.d_rec_v += oc, f.d_pred_u += oc, f.d_pred_v += oc;
        f.d_mode_y += ou, f.d_mode_c += ou;
        if (f.d_cost_y) f.d_cost_y += ou;
        if (f.d_cost_c) f.d_cost_c += ou;
    }
    __shared__ __attribute__((aligned(16))) uint8_t s_ref[2][kRefRow], s_rev[2][kRefRow];  // the samples, smoothed; both in reverse
    __shared__ __attribute__((aligned(16))) uint8_t s_refc[2][4 * 16 + 8], s_revc[2][4 * 16 + 8];  // U, V: chroma reads the samples as they are
    __shared__ __attribute__((aligned(16))) uint8_t s_main[NWAVES][2][kMainRow];
    __shared__ __attribute__((aligned(16))) uint8_t s_srcq[2][N][N + 4];  // the source CU, [row][column] and transposed
    __shared__ __attribute__((aligned(16))) uint8_t s_srcc[2][2][NC][NC + 4];  // its chroma: [U, V][as it is, transposed]
    __shared__ uint32_t s_sad[35], s_sadc[6];
    __shared__ int s_dc, s_dcc[2], s_mpm[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cuy = w.cy_lo + blockIdx.x, cux = w.d - 2 * cuy;
    if (cux < 0 || cux >= w.ncx || cuy >= w.ncy) return false;  // block-uniform (the host sizes the grid to fit)
    const int x0 = cux * N, y0 = cuy * N, cu = cuy * w.ncx + cux;
    const bool bL = cux > 0, bT = cuy > 0, bTR = bT && cux + 1 < w.ncx;
    X265AMD_INTRA_STAMP(0);

    // The source CU, a pixel per thread, and the left neighbour's mode: asked for first, so that they travel together
    // with the reference samples (one memory latency instead of two).  The source goes to LDS by rows and by columns, luma
    // and chroma: the quads of four pixels that the SADs compare against are aligned dwords there whatever the plane's
    // stride and address.
    const int left_mode = (tid == NT - 1 && bL) ? f.d_mode_y[cu - 1] : 1;
    constexpr int TPC = (2 * NPIXC + NT - 1) / NT;
    int cpix[TPC];
#pragma unroll
    for (int t = 0; t < TPC; t++) {
        const int i = tid + NT * t, pl = i >= NPIXC, j = i - pl * NPIXC;
        cpix[t] = i < 2 * NPIXC ? (pl ? f.d_src_v : f.d_src_u)[(int64_t)(y0 / 2 + (j >> L2C)) * f.stride_c + x0 / 2 + (j & (NC - 1))] : 0;
    }
    constexpr int TP = (NPIX + NT - 1) / NT;
    int tpix[TP];
#pragma unroll
    for (int t = 0; t < TP; t++) {
        const int i = tid + NT * t;
        tpix[t] = i < NPIX ? f.d_src_y[(int64_t)(y0 + (i >> L2)) * f.stride_y + x0 + (i & (N - 1))] : 0;
    }

    // ---- reference samples: luma, and its smoothed copy; U and V ----
#pragma unroll
    for (int t = 0; t < TP; t++) {
        const int i = tid + NT * t;
        if (i < NPIX) {
            s_srcq[0][i >> L2][i & (N - 1)] = (uint8_t)tpix[t];
            s_srcq[1][i & (N - 1)][i >> L2] = (uint8_t)tpix[t];
        }
    }
#pragma unroll
    for (int t = 0; t < TPC; t++) {
        const int i = tid + NT * t, pl = i >= NPIXC, j = i - pl * NPIXC;
        if (i < 2 * NPIXC) {
            s_srcc[pl][0][j >> L2C][j & (NC - 1)] = (uint8_t)cpix[t];
            s_srcc[pl][1][j & (NC - 1)][j >> L2C] = (uint8_t)cpix[t];
        }
    }
    if (tid <= 4 * N) s_ref[0][tid] = (uint8_t)intra_ref_sample(f.d_rec_y, f.stride_y, x0, y0, N, tid, bL, bT, bTR);
    for (int i = tid; i < 2 * (4 * NC + 1); i += NT) {
        const int pl = i >= 4 * NC + 1, j = i - pl * (4 * NC + 1);
        s_refc[pl][j] = (uint8_t)intra_ref_sample(pl ? f.d_rec_v : f.d_rec_u, f.stride_c, x0 / 2, y0 / 2, NC, j, bL, bT, bTR);
    }
    __syncthreads();
    X265AMD_INTRA_STAMP(1);
    if (tid <= 4 * N) {
        const int a = s_ref[0][tid];
        const uint8_t sm = (tid == 0 || tid == 4 * N) ? (uint8_t)a : (uint8_t)((s_ref[0][tid - 1] + 2 * a + s_ref[0][tid + 1] + 2) >> 2);
        s_ref[1][tid] = sm;
        s_rev[0][4 * N - tid] = (uint8_t)a;
        s_rev[1][4 * N - tid] = sm;
    }
    for (int i = tid; i < 2 * (4 * NC + 1); i += NT) {
        const int pl = i >= 4 * NC + 1, j = i - pl * (4 * NC + 1);
        s_revc[pl][4 * NC - j] = s_refc[pl][j];
    }
    if (wave == 0) intra_dc_value<N>(s_ref[0], lane, &s_dc);
    if (wave == 1) intra_dc_value<NC>(s_refc[0], lane, &s_dcc[0]);
    if (wave == 2) intra_dc_value<NC>(s_refc[1], lane, &s_dcc[1]);
    if (tid == NT - 1) {
        // most probable modes (:1268-1293).  The mode above counts only inside the same CU there (`bT && uiY`
        // with uiY the offset inside the CU, always 0): it is DC whatever the CU above chose.
        const int left = left_mode, top = 1;
        int m0, m1, m2;
        if (left == top) {
            if (left > 1)
                m0 = left, m1 = ((left + 29) % 32) + 2, m2 = ((left - 1) % 32) + 2;
            else
                m0 = 0, m1 = 1, m2 = 26;
        } else {
            m0 = left, m1 = top;
            m2 = (left != 0 && top != 0) ? 0 : (left + top) < 2 ? 26 : 1;
        }
        s_mpm[0] = m0, s_mpm[1] = m1, s_mpm[2] = m2;
    }
    __syncthreads();
    X265AMD_INTRA_STAMP(2);

    // ---- the SADs: 35 luma modes and, with them, the five chroma candidates that do not depend on the luma decision
    //      (planar, vertical, horizontal, DC, and 34, which stands in for whichever of the four the luma mode equals:
    //      src/encode.cpp:781-853), one list of 40 tasks dealt round the waves ----
    const int dc = s_dc;
    uint32_t srcv[PER / 4], srch[PER / 4];
#pragma unroll
    for (int t = 0; t < PER / 4; t++) {
        const int q = lane + 64 * t;
        srcv[t] = *reinterpret_cast<const uint32_t *>(&s_srcq[0][q / (N / 4)][4 * (q % (N / 4))]);
        srch[t] = *reinterpret_cast<const uint32_t *>(&s_srcq[1][q / (N / 4)][4 * (q % (N / 4))]);
    }
    // SAD(U) + SAD(V) of chroma mode `mode` (wave-uniform), by one wave, a quad of four pixels a lane and step like luma
    // (no edge smoothing for chroma: bLuma = false in the reference's calls)
    auto chroma_sad = [&](int mode) __attribute__((always_inline)) -> uint32_t {
        constexpr int QLC = NC / 4, NQC = NPIXC / 4;  // quads per line, per plane (64 at CU 32, 16 at CU 16)
        const bool hor = mode >= 2 && mode < 18;
        const int ang = intra_angle(mode);
        const uint8_t *rows[2] = {nullptr, nullptr};
        if (mode >= 2) {
            rows[0] = intra_main_row<NC>(hor ? s_revc[0] : s_refc[0], mode, lane, s_main[wave][0]);
            rows[1] = intra_main_row<NC>(hor ? s_revc[1] : s_refc[1], mode, lane, s_main[wave][1]);
        }
        uint32_t s = 0;
#pragma unroll
        for (int t = 0; t < (NQC + 63) / 64; t++) {
            const int q = lane + 64 * t;
            const bool on = q < NQC;
            const int qc = on ? q : 0, k = qc / QLC, x0 = 4 * (qc % QLC);
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                uint32_t pq;
                if (mode >= 2) {
                    const int pos = __mul24(k + 1, ang);
                    pq = intra_ang_quad(rows[pl], NC + x0 + (pos >> 5) + 1, pos & 31);
                } else if (mode == 1) {
                    pq = (uint32_t)s_dcc[pl] * 0x01010101u;
                } else {
                    const uint8_t *tl = s_refc[pl] + 2 * NC;
                    int top[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) top[j] = tl[1 + x0 + j];
                    pq = intra_planar_quad<NC>(tl, k, x0, top, tl[1 + NC], tl[-1 - NC]);
                }
                const uint32_t sq = *reinterpret_cast<const uint32_t *>(&s_srcc[pl][hor ? 1 : 0][k][x0]);
                if (on) s = __builtin_amdgcn_sad_u8(pq, sq, s);
            }
        }
        return wave_sum_u32(s);
    };
    for (int task = wave; task < 40; task += NWAVES) {  // wave-uniform
        if (task < 35) {
            const uint32_t sd = intra_luma_sad_quads<N, PER>(s_ref, s_rev, srcv, srch, dc, task, lane, s_main[wave][0]);
            if (lane == 0) s_sad[task] = sd;
        } else {
            const int k = task - 35;  // s_sadc[0..3]: planar, vertical, horizontal, DC; [4]: mode 34
            const uint32_t sd = chroma_sad(k == 0 ? 0 : k == 1 ? 26 : k == 2 ? 10 : k == 3 ? 1 : 34);
            if (lane == 0) s_sadc[k] = sd;
        }
    }
    __syncthreads();
    X265AMD_INTRA_STAMP(3);
    // ---- luma decision: every wave works it out for itself (no barrier to hand it round) ----
    const x265amd_intra_best lb = intra_luma_argmin_wave(s_sad, s_mpm[0], s_mpm[1], s_mpm[2], w.lambda, lane);
    const int bm = (int)lb.mode;
    if (tid == 0) {
        f.d_mode_y[cu] = (uint8_t)bm;
        if (f.d_cost_y) f.d_cost_y[cu] = lb.cost;
    }
    X265AMD_INTRA_STAMP(4);
    // the one chroma candidate that does depend on it: the luma mode itself, unless it is one of the five above
    const bool known = bm == 0 || bm == 26 || bm == 10 || bm == 1 || bm == 34;
    if (wave == NWAVES - 1 && !known) {
        const uint32_t sd = chroma_sad(bm);
        if (lane == 0) s_sadc[5] = sd;
    }
    {
        const uint8_t *ref = s_ref[intra_filter<L2 - 2>(bm)];
        for (int i = tid; i < NPIX; i += NT)
            f.d_pred_y[(int64_t)(y0 + (i >> L2)) * f.stride_y + x0 + (i & (N - 1))] = (uint8_t)intra_pixel<N>(ref, bm, i >> L2, i & (N - 1), dc);
    }
    __syncthreads();
    X265AMD_INTRA_STAMP(5);

    // ---- chroma decision (:781-853): candidates in the reference's order -- planar, vertical, horizontal, DC with the one
    //      equal to the luma mode replaced by 34, LM (index 4) skipped, then the luma mode; cost = SAD(U) + SAD(V); first
    //      strict minimum.  Six numbers: every wave again for itself ----
    int bmc;
    {
        const int base[4] = {0, 26, 10, 1};
        uint32_t bc = X265AMD_MAX_SAD;
        bmc = 0;  // candidate 0's mode, should nothing be under MAX_SAD
        bool dup_seen = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool dup = base[k] == bm && !dup_seen;
            dup_seen = dup_seen || dup;
            const int mode = dup ? 34 : base[k];
            const uint32_t c = dup ? s_sadc[4] : s_sadc[k];
            if (k == 0) bmc = mode;
            if (c < bc) bc = c, bmc = mode;
        }
        {   // candidate 5: the luma mode
            const uint32_t c = bm == 0 ? s_sadc[0] : bm == 26 ? s_sadc[1] : bm == 10 ? s_sadc[2] : bm == 1 ? s_sadc[3] : bm == 34 ? s_sadc[4] : s_sadc[5];
            if (c < bc) bc = c, bmc = bm;
        }
        if (tid == 0) {
            f.d_mode_c[cu] = (uint8_t)bmc;
            if (f.d_cost_c) f.d_cost_c[cu] = bc;
        }
    }
    X265AMD_INTRA_STAMP(6);
    for (int i = tid; i < 2 * NPIXC; i += NT) {
        const int pl = i >= NPIXC, j = i - pl * NPIXC;
        const int64_t o = (int64_t)(y0 / 2 + (j >> L2C)) * f.stride_c + x0 / 2 + (j & (NC - 1));
        (pl ? f.d_pred_v : f.d_pred_u)[o] = (uint8_t)intra_pixel<NC, false>(s_refc[pl], bmc, j >> L2C, j & (NC - 1), s_dcc[pl]);
    }
    X265AMD_INTRA_STAMP(7);
    return true;
}

template <int N, int NT>
__global__ __launch_bounds__(NT, 1024 / NT) void intra_wave_kernel(const IntraWave w)
{
    intra_wave_body<N, NT>(w);
}

// A whole anti-diagonal step in one launch: the prediction of the workgroup's CU as above, then its transform stage
// (forward + inverse of the 32x32 tiles that hold the CU's luma and chroma blocks: tq_mfma_body in diagonal mode,
// exactly what tq_mfma_diag_yuv_kernel runs as a launch of its own) on three of the workgroup's waves -- luma, U, V.
// The predictions they read were stored by this workgroup: visible after the barrier (one compute unit, one L1).
// py / pc: the transform stage's parameters for the luma plane and for the chroma planes; the workgroup's frame g
// takes chroma planes by = g * cby_mul and g * cby_mul + cby_add of pc (see intra_encode for the three layouts).
template <int N, int NT>
__global__ __launch_bounds__(NT, 1024 / NT) void intra_wave_tq_kernel(const IntraWave w, const TqParams py, const TqParams pc, const int cby_mul,
                                                                      const int cby_add)
{
    if (!intra_wave_body<N, NT>(w)) return;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cu = (w.cy_lo + (int)blockIdx.x) * w.ncx + w.d - 2 * (w.cy_lo + (int)blockIdx.x);
    X265AMD_INTRA_STAMP(8);
    if (wave == 0) {
        tq_mfma_body<N, TQ_BOTH, false>(py, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, -1);
        X265AMD_INTRA_STAMP(9);
    } else if (wave <= 2)
        tq_mfma_body<N / 2, TQ_BOTH, false>(pc, (int)blockIdx.x, (int)blockIdx.y * cby_mul + (wave == 2 ? cby_add : 0), (int)gridDim.x, -1);
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_intra_decide_luma(const uint8_t *d_src, int src_stride, int width, int height, const int32_t *d_xy,
                                         int ncus, int size, const uint8_t *d_ref, int ref_stride, const uint8_t *d_mpm,
                                         uint32_t lambda, x265amd_intra_best *d_best, uint32_t *d_sads, uint8_t *d_pred,
                                         void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!d_src || !d_xy || !d_ref || !d_mpm || !d_best) return X265AMD_ERR_ARG;
    if (size != 4 && size != 8 && size != 16 && size != 32) return X265AMD_ERR_ARG;  // MAX_CU_SIZE, inc/config.h:38
    if (width < size || height < size || src_stride < width || ref_stride < 4 * size + 1 || ncus < 0) return X265AMD_ERR_ARG;
    if (lambda > (1u << 20)) return X265AMD_ERR_ARG;  // cost = SAD + 3 lambda stays far below MAX_SAD
    if (ncus == 0) return X265AMD_OK;
    const dim3 grid((unsigned)ncus), block(256);
    hipStream_t st = (hipStream_t)stream;
#define X265AMD_INTRA_LAUNCH(NN)                                                                                       \
    hipLaunchKernelGGL(intra_decide_kernel<NN>, grid, block, 0, st, d_src, src_stride, width, height, d_xy, d_ref,     \
                       ref_stride, d_mpm, lambda, d_best, d_sads, d_pred)
    switch (size) {
    case 4: X265AMD_INTRA_LAUNCH(4); break;
    case 8: X265AMD_INTRA_LAUNCH(8); break;
    case 16: X265AMD_INTRA_LAUNCH(16); break;
    default: X265AMD_INTRA_LAUNCH(32); break;
    }
#undef X265AMD_INTRA_LAUNCH
    return hip_status(hipGetLastError());
}

static int intra_frame_check(const x265amd_intra_frame *f)
{
    if (!f) return X265AMD_ERR_ARG;
    if (!f->d_src_y || !f->d_src_u || !f->d_src_v || !f->d_rec_y || !f->d_rec_u || !f->d_rec_v || !f->d_pred_y ||
        !f->d_pred_u || !f->d_pred_v || !f->d_mode_y || !f->d_mode_c || !f->d_level_y || !f->d_level_u || !f->d_level_v)
        return X265AMD_ERR_ARG;
    if (f->cu != 16 && f->cu != 32) return X265AMD_ERR_ARG;  // src/params.cpp:76, inc/config.h:38
    if (f->width <= 0 || f->height <= 0 || f->width % f->cu || f->height % f->cu) return X265AMD_ERR_ARG;
    if (f->stride_y < f->width || f->stride_c < f->width / 2 || f->qp < 0 || f->qp > 51) return X265AMD_ERR_ARG;
    return X265AMD_OK;
}

// x265amd_debug_intra_fused (include/x265_amd_debug.h): 0 keeps the transform stage a launch of its own (A/B runs)
// x265amd_debug_intra_threads: threads per workgroup of the prediction kernel, 256, 512 or 1024 (0 = by the step's size)

struct IntraBatch {
    int nframes;
    int64_t fs_y, fs_c, fs_cu;
};

// the transform stage that rides at the end of the prediction kernel (intra_wave_tq_kernel), or none
struct IntraTq {
    const TqParams *py = nullptr, *pc = nullptr;
    int cby_mul = 0, cby_add = 0;
};

static int intra_wave_launch(const x265amd_intra_frame *f, int d, hipStream_t st, const IntraBatch &b = IntraBatch{1, 0, 0, 0},
                             const IntraTq &tq = IntraTq{})
{
    IntraWave w;
    w.f = *f;
    w.fs_y = b.fs_y, w.fs_c = b.fs_c, w.fs_cu = b.fs_cu;
    w.ncx = f->width / f->cu;
    w.ncy = f->height / f->cu;
    w.d = d;
    w.lambda = (uint32_t)f->qp;  // src/encode.cpp:597
    w.stats = debug_state().intra_stats;
    // cux = d - 2 cuy in [0, ncx): cuy from ceil((d - ncx + 1) / 2) to d / 2
    const int lo = d - (w.ncx - 1) > 0 ? (d - (w.ncx - 1) + 1) / 2 : 0;
    const int hi = d / 2 < w.ncy - 1 ? d / 2 : w.ncy - 1;
    if (hi < lo) return X265AMD_OK;
    w.cy_lo = lo;
    // more CUs in the step than the chip has compute units: the 8-wave shape, two workgroups to a compute unit; more
    // than twice as many: 4 waves, four to a compute unit
    const int64_t nwg = (int64_t)(hi - lo + 1) * b.nframes;
    const int forced_threads = debug_state().intra_threads;
    const int threads = forced_threads ? forced_threads : nwg > 2 * cu_count() ? 256 : nwg > cu_count() ? 512 : 1024;
    const dim3 grid((unsigned)(hi - lo + 1), (unsigned)b.nframes), block((unsigned)threads);
#define X265AMD_INTRA_SHAPES(KERNEL, NN, ...)                                                              \
    switch (threads) {                                                                                     \
    case 256: hipLaunchKernelGGL((KERNEL<NN, 256>), grid, block, 0, st, __VA_ARGS__); break;             \
    case 512: hipLaunchKernelGGL((KERNEL<NN, 512>), grid, block, 0, st, __VA_ARGS__); break;             \
    default: hipLaunchKernelGGL((KERNEL<NN, 1024>), grid, block, 0, st, __VA_ARGS__); break;             \
    }
    if (tq.py) {
        // what the kernel assumes of the stage's parameters: this diagonal, these CUs, a grid that matches
        if (tq.py->diag_d != d || tq.py->diag_lo != lo || tq.py->diag_count != hi - lo + 1 || (1 << tq.py->diag_cu_log2) != f->cu ||
            tq.py->d.tu != f->cu || f->cu != 32)
            return X265AMD_ERR_ARG;
        X265AMD_INTRA_SHAPES(intra_wave_tq_kernel, 32, w, *tq.py, *tq.pc, tq.cby_mul, tq.cby_add)
    } else if (f->cu == 16) {
        X265AMD_INTRA_SHAPES(intra_wave_kernel, 16, w)
    } else {
        X265AMD_INTRA_SHAPES(intra_wave_kernel, 32, w)
    }
#undef X265AMD_INTRA_SHAPES
    return hip_status(hipGetLastError());
}

extern "C" void x265amd_debug_set_intra_stats(uint32_t *d_stats)
{
    if (device_state()) debug_state().intra_stats = d_stats;
}
extern "C" void x265amd_debug_intra_fused(int on)
{
    if (device_state()) debug_state().intra_fused = on;
}
extern "C" void x265amd_debug_intra_threads(int threads)
{
    if (device_state()) debug_state().intra_threads = threads == 256 || threads == 512 || threads == 1024 ? threads : 0;
}

extern "C" int x265amd_intra_waves(const x265amd_intra_frame *f)
{
    if (intra_frame_check(f) != X265AMD_OK) return -1;
    return f->width / f->cu + 2 * (f->height / f->cu - 1);
}

extern "C" int x265amd_intra_predict_wave(const x265amd_intra_frame *f, int wave, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    const int rc = intra_frame_check(f);
    if (rc != X265AMD_OK) return rc;
    if (wave < 0 || wave >= x265amd_intra_waves(f)) return X265AMD_ERR_ARG;
    return intra_wave_launch(f, wave, (hipStream_t)stream);
}

// one frame, or a batch of independent frames whose planes lie at fixed distances from the first frame's
static int intra_encode(const x265amd_intra_frame *f, const IntraBatch &b, hipStream_t stream)
{
    // the transform stage of a wave: forward + inverse of the three planes (Stage 3a / 3b, src/encode.cpp:863-925)
    // over the 128 x 32 pixel tile units that hold the wave's CUs (the whole plane where the matrix-core kernel does
    // not apply) -- TUs of earlier waves that share a unit come out as they already are (same source, same
    // prediction), TUs of later waves are overwritten when their wave comes
    x265amd_tq_desc ty = {}, tu = {}, tv = {};
    ty.width = f->width, ty.height = f->height, ty.tu = f->cu, ty.qp = f->qp, ty.slice_type = X265AMD_SLICE_I, ty.nframes = b.nframes;
    ty.d_src = f->d_src_y, ty.src_stride = f->stride_y;
    ty.d_pred = f->d_pred_y, ty.pred_stride = f->stride_y;
    ty.d_level = f->d_level_y, ty.level_stride = f->stride_y;
    ty.d_rec = f->d_rec_y, ty.rec_stride = f->stride_y;
    ty.d_abs_sum = f->d_abs_sum_y;
    ty.src_frame_stride = ty.pred_frame_stride = ty.rec_frame_stride = ty.level_frame_stride = b.fs_y;
    ty.abs_sum_frame_stride = b.fs_cu;
    tu.width = f->width / 2, tu.height = f->height / 2, tu.tu = f->cu / 2, tu.qp = x265amd_chroma_qp(f->qp);
    tu.slice_type = X265AMD_SLICE_I, tu.nframes = b.nframes;
    tu.src_stride = tu.pred_stride = tu.level_stride = tu.rec_stride = f->stride_c;
    tu.src_frame_stride = tu.pred_frame_stride = tu.rec_frame_stride = tu.level_frame_stride = b.fs_c;
    tu.abs_sum_frame_stride = b.fs_cu;
    tv = tu;
    tu.d_src = f->d_src_u, tu.d_pred = f->d_pred_u, tu.d_level = f->d_level_u, tu.d_rec = f->d_rec_u, tu.d_abs_sum = f->d_abs_sum_u;
    tv.d_src = f->d_src_v, tv.d_pred = f->d_pred_v, tv.d_level = f->d_level_v, tv.d_rec = f->d_rec_v, tv.d_abs_sum = f->d_abs_sum_v;
    // U and V in one launch (one batch of planes) when all V planes continue the U planes' progression: a single
    // frame whose four V planes sit at one common distance behind their U planes (a 4:2:0 frame buffer), or a batch
    // stored plane-major (all U planes, then all V planes, one stride)
    TqDiag pair{};  // plane grouping of the chroma launch (fgroup = 0: none)
    const int64_t du = f->d_src_v - f->d_src_u;
    const int64_t nu = (int64_t)(tu.height / tu.tu) * (tu.width / tu.tu);
    bool both = du > 0 && f->d_pred_v - f->d_pred_u == du && f->d_rec_v - f->d_rec_u == du &&
                f->d_level_v - f->d_level_u == du && du >= (int64_t)tu.height * f->stride_c &&
                (!f->d_abs_sum_u == !f->d_abs_sum_v);
    if (both && b.nframes == 1) {
        both = !f->d_abs_sum_u || f->d_abs_sum_v - f->d_abs_sum_u >= nu;
        if (both) {
            tu.nframes = 2;
            tu.src_frame_stride = tu.pred_frame_stride = tu.rec_frame_stride = tu.level_frame_stride = du;
            tu.abs_sum_frame_stride = f->d_abs_sum_u ? f->d_abs_sum_v - f->d_abs_sum_u : 0;
        }
    } else if (both) {
        const bool plane_major = du == (int64_t)b.nframes * b.fs_c &&
                                 (!f->d_abs_sum_u || f->d_abs_sum_v - f->d_abs_sum_u == (int64_t)b.nframes * b.fs_cu);
        if (plane_major) {
            tu.nframes = 2 * b.nframes;
        } else if (b.fs_c % du == 0 && b.fs_c / du >= 2 && b.fs_c / du < 65536 && (int64_t)b.nframes * 2 <= 65535 &&
                   tq_takes_mfma(&tu)) {  // (plane groups are the grid's y: above 32767 frames U and V go separately)
            // frame buffers one after the other (4:2:0: a frame is six chroma-plane sizes long): plane j of frame g is
            // du * (g * fmul + j) behind the first U plane
            pair.fgroup = 2;
            pair.fmul = (int)(b.fs_c / du);
            pair.sum_pair = f->d_abs_sum_u ? (long long)(f->d_abs_sum_v - f->d_abs_sum_u) : 0;
            tu.src_frame_stride = tu.pred_frame_stride = tu.rec_frame_stride = tu.level_frame_stride = du;
        } else {
            both = false;
        }
    }
    // the whole transform stage of a step as one launch where U and V already share theirs and every plane takes the
    // matrix-core kernel (luma width a multiple of 32: the chroma width must be one of 16)
    const bool one_launch = both && tq_takes_mfma(&ty) && tq_takes_mfma(&tu);
    const int nw = x265amd_intra_waves(f);
    const int ncx = f->width / f->cu, ncy = f->height / f->cu;
    const int l2 = f->cu == 16 ? 4 : 5;
    // chroma planes of frame g as tq_mfma_body counts them (`by`): U and V of the one frame; planes 2 g and 2 g + 1 of
    // the pairs; or g among the U planes and nframes + g among the V planes
    // Only where no two CUs of one anti-diagonal share a 32x32 tile of any plane: the transform stage writes whole
    // tiles (blocks of later steps come out as garbage until their own step), which is harmless between launches but
    // not inside one.  The CUs of a tile are on diagonals d0 + i + 2 l, i, l < 32 / block: distinct for blocks of 32
    // and of 16 (CU 32: luma 32, chroma 16), not for the 8x8 chroma blocks of CU 16 (i = 2, l = 0 meets i = 0, l = 1).
    const bool fused_ok = one_launch && debug_state().intra_fused && f->cu == 32;
    IntraTq fused;
    fused.cby_mul = b.nframes == 1 ? 0 : pair.fgroup ? 2 : 1;
    fused.cby_add = b.nframes == 1 || pair.fgroup ? 1 : b.nframes;
    for (int d = 0; d < nw; d++) {
        // the transform stage follows the diagonal: only the tile units that hold its CUs
        const int lo = d - (ncx - 1) > 0 ? (d - (ncx - 1) + 1) / 2 : 0;
        const int hi = d / 2 < ncy - 1 ? d / 2 : ncy - 1;
        if (hi < lo) continue;  // a picture one CU wide has no CU on its odd diagonals (a count of 0 would mean "the whole plane")
        const TqDiag dy{hi - lo + 1, d, lo, l2}, dc{hi - lo + 1, d, lo, l2 - 1, pair.fgroup, pair.fmul, pair.sum_pair};
        if (fused_ok) {
            // prediction and transform stage of the step as one launch
            TqParams py, pc;
            int nyl = 0, nyc = 0;
            int rc = tq_diag_yuv_params(&ty, dy, &tu, dc, py, pc, nyl, nyc);
            if (rc == X265AMD_OK && (nyl != b.nframes || nyc != 2 * b.nframes)) rc = X265AMD_ERR_ARG;  // the `by` the kernel forms
            fused.py = &py, fused.pc = &pc;
            if (rc == X265AMD_OK) rc = intra_wave_launch(f, d, stream, b, fused);
            if (rc != X265AMD_OK) return rc;
            continue;
        }
        int rc = intra_wave_launch(f, d, stream, b);
        if (rc == X265AMD_OK && one_launch) {
            rc = tq_roundtrip_diag_yuv(&ty, dy, &tu, dc, stream);  // luma and both chroma planes of every frame at once
        } else {
            if (rc == X265AMD_OK) rc = tq_roundtrip_diag(&ty, dy, stream);
            if (rc == X265AMD_OK) rc = tq_roundtrip_diag(&tu, dc, stream);
            if (rc == X265AMD_OK && !both) rc = tq_roundtrip_diag(&tv, dc, stream);
        }
        if (rc != X265AMD_OK) return rc;
    }
    return X265AMD_OK;
}

extern "C" int x265amd_intra_encode_frame(const x265amd_intra_frame *f, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    const int rc = intra_frame_check(f);
    if (rc != X265AMD_OK) return rc;
    return intra_encode(f, IntraBatch{1, 0, 0, 0}, (hipStream_t)stream);
}

extern "C" int x265amd_intra_encode_frames(const x265amd_intra_frame *f, int nframes, int64_t luma_frame_stride,
                                           int64_t chroma_frame_stride, int64_t cu_frame_stride, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    const int rc = intra_frame_check(f);
    if (rc != X265AMD_OK) return rc;
    if (nframes < 0 || nframes > 65535) return X265AMD_ERR_ARG;  // frames are the grid's y dimension
    if (nframes == 0) return X265AMD_OK;
    if (nframes > 1) {
        // a frame's planes must end before the next frame's begin (outputs of different frames never alias)
        const int64_t need_y = (int64_t)(f->height - 1) * f->stride_y + f->width;
        const int64_t need_c = (int64_t)(f->height / 2 - 1) * f->stride_c + f->width / 2;
        const int64_t ncus = (int64_t)(f->width / f->cu) * (f->height / f->cu);
        if (luma_frame_stride < need_y || chroma_frame_stride < need_c || cu_frame_stride < ncus) return X265AMD_ERR_ARG;
    }
    return intra_encode(f, IntraBatch{nframes, luma_frame_stride, chroma_frame_stride, cu_frame_stride}, (hipStream_t)stream);
}
