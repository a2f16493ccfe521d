// intra_stream.hip -- HOST code only (no kernel in this file): the syntax writer of one intra picture.
//
// x265amd_intra_encode_frame leaves, per picture, the quantised levels and the per-CU modes; what the reference
// does with them next is Stage 4 of its encode loop (xWriteCU, src/encode.cpp:474-585,928) between the parameter
// sets / slice header it wrote before the loop (:626-662) and the CABAC flush + slice end after it (:942-943).
// That part is bit-serial and stays on the CPU; this file is it, restated for exactly what
// the reference can emit -- SLICE_I, one CU depth, CU = PU = TU of 16 or 32 (src/params.cpp:72-77), 4:2:0, the
// HM-6.1 syntax of src/bitstream.cpp -- so that x265amd_intra_write_picture's bytes are the reference's bytes.
// At these sizes every coefficient scan is the up-right diagonal one (getCoefScanIdx, src/encode.cpp:81-109: the
// mode-dependent scans apply to 4x4 / 8x8 luma and 4x4 chroma only), which is the only scan built here.
#include <stdint.h>
#include <string.h>

#include "../../include/x265_amd.h"

namespace {

// ------------------------------------------------------------------------------------------------------------
// Bytes out.  MSB-first bit packing; a 0x03 goes in front of any byte <= 3 that follows two zero bytes
// (inc/bitstream.h:48-59: the reference checks the two bytes before the write position, whatever wrote them).
// ------------------------------------------------------------------------------------------------------------
class ByteSink {
public:
    ByteSink(uint8_t *buf, size_t cap) : base_(buf), cap_(cap) {}
    // start code + NAL header as the reference stores them: a raw 32-bit little-endian word, never escaped
    // (xPutBits32, inc/bitstream.h:72-77)
    void raw_word(uint32_t le)
    {
        for (int i = 0; i < 4; i++) push((uint8_t)(le >> (8 * i)));
    }
    void bits(uint32_t value, int n)  // n <= 32, value < 2^n
    {
        for (int taken = 0; taken < n;) {
            const int room = 8 - fill_, take = n - taken < room ? n - taken : room;
            const uint32_t chunk = (value >> (n - taken - take)) & ((1u << take) - 1u);
            cur_ = (uint8_t)(cur_ | (chunk << (room - take)));
            fill_ += take;
            taken += take;
            if (fill_ == 8) {
                escaped(cur_);
                cur_ = 0;
                fill_ = 0;
            }
        }
    }
    void byte(uint32_t b)  // eight bits at once; the arithmetic coder's output is always byte-aligned
    {
        if (fill_ == 0)
            escaped((uint8_t)b);
        else
            bits(b & 0xffu, 8);
    }
    void flag(uint32_t b) { bits(b & 1u, 1); }
    void ue(uint32_t v)  // Exp-Golomb, xWriteUvlc (inc/bitstream.h:134-140)
    {
        const uint32_t code = v + 1;
        int len = 0;
        while ((code >> (len + 1)) != 0) len++;
        if (len) bits(0, len);
        bits(code, len + 1);
    }
    void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }  // xWriteSvlc (:142-153)
    void align(uint32_t fill_bit)  // xWriteAlignZero / xWriteAlignOne (:101-111)
    {
        const int n = (8 - fill_) & 7;
        if (n) bits(fill_bit ? (1u << n) - 1u : 0u, n);
    }
    void trailing()  // xWriteRBSPTrailingBits (:113-117)
    {
        flag(1);
        align(0);
    }
    size_t size() const { return n_; }
    bool overflow() const { return overflow_; }

private:
    void push(uint8_t b)
    {
        if (n_ < cap_)
            base_[n_] = b;
        else
            overflow_ = true;
        n_++;  // keeps counting, so that the emulation checks below stay in step; nothing is stored past cap_
        prev2_ = prev1_;
        prev1_ = b;
    }
    void escaped(uint8_t b)
    {
        if (prev1_ == 0 && prev2_ == 0 && b <= 3) push(3);
        push(b);
    }
    uint8_t *base_;
    size_t cap_, n_ = 0;
    uint8_t cur_ = 0, prev1_ = 0xff, prev2_ = 0xff;
    int fill_ = 0;
    bool overflow_ = false;
};

// ------------------------------------------------------------------------------------------------------------
// Context models.  The reference keeps one flat array addressed through OFF_*_CTX (inc/x265.h:144-167); only the
// ones an intra CU touches exist here, each set under its own name.  Initial values: the SLICE_I rows of
// src/bitstream.cpp:343-520 (HM-6.1's tables); CNU entries of sets that are only partly used are kept so that the
// indices stay those of the reference.
// ------------------------------------------------------------------------------------------------------------
enum Ctx : int {
    kPartSize = 0,                 // INIT_PART_SIZE[0][0]
    kPrevIntraLuma = 1,            // INIT_INTRA_PRED_MODE[0][0]
    kChromaPred = 2,               // INIT_CHROMA_PRED_MODE[0][0]
    kCbfLuma = 3,                  // INIT_QT_CBF[0][1]      (getCtxQtCbf(luma) = 1, src/encode.cpp:463-472)
    kCbfChroma = 4,                // INIT_QT_CBF[0][5 + 0]
    kSigGroup = 5,                 // 2 luma + 2 chroma
    kSig = kSigGroup + 4,          // 27 luma + 21 chroma
    kLastX = kSig + 48,            // 15 luma + 15 chroma
    kLastY = kLastX + 30,          // the same initial values as kLastX (src/bitstream.cpp:593-596)
    kGreater1 = kLastY + 30,       // 16 luma + 8 chroma
    kGreater2 = kGreater1 + 24,    // 4 luma + 2 chroma
    kNumCtx = kGreater2 + 6
};
constexpr uint8_t kCnu = 154;
const uint8_t kInitSigGroup[4] = {91, 171, 134, 141};
const uint8_t kInitSig[48] = {141, 111, 125, 110, 110, 94,  124, 108, 124, 125, 139, 124, 63,  139, 168, 138,
                              107, 123, 92,  111, 141, 107, 125, 141, 179, 153, 125, 140, 139, 182, 123, 47,
                              153, 182, 137, 149, 192, 152, 224, 136, 31,  136, 74,  140, 141, 136, 139, 111};
const uint8_t kInitLast[30] = {110, 110, 124, 110, 140, 111, 125, 111, 127, 111, 111, 156, 127, 127, 111,
                               108, 123, 63,  kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu, kCnu};
const uint8_t kInitGreater1[24] = {140, 92,  137, 138, 140, 152, 138, 139, 153, 74,  149, 92,
                                   139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197};
const uint8_t kInitGreater2[6] = {138, 153, 136, 167, 152, 152};

// state = (pStateIdx << 1) | valMPS from an initial value and the slice QP (xCabacInitEntry, src/bitstream.cpp:526-542)
uint8_t ctx_state(uint8_t init, int qp)
{
    const int slope = (init >> 4) * 5 - 45, offset = ((init & 15) << 3) - 16;
    int s = ((slope * qp) >> 4) + offset;
    s = s < 1 ? 1 : s > 126 ? 126 : s;
    const int mps = s >= 64;
    return (uint8_t)(((mps ? s - 64 : 63 - s) << 1) + mps);
}

// rangeTabLPS (the standard's table; src/table.cpp:196-261)
const uint8_t kRangeLps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195},
    {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},
    {85, 104, 123, 142},  {81, 99, 117, 135},   {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},
    {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},     {56, 69, 81, 94},     {53, 65, 77, 89},
    {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},     {41, 50, 59, 69},
    {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},
    {23, 28, 33, 39},     {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},
    {18, 22, 26, 30},     {17, 21, 25, 28},     {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},
    {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},     {12, 14, 17, 20},     {11, 14, 16, 19},
    {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},      {9, 11, 12, 14},
    {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
// transIdxLps of the standard; the reference's 128-entry tables (src/table.cpp:174-193) are this and "state + 1,
// saturating at 62" applied to (pStateIdx << 1 | valMPS)
const uint8_t kNextLps[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
                              18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
                              31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

// State transitions on (pStateIdx << 1 | valMPS) as the reference tabulates them (src/table.cpp:174-193), built from
// transIdxLps: an LPS at state 0 flips the MPS, an MPS moves one state up to 62.  renorm[lps >> 3] = g_aucRenormTable
// (src/table.cpp:263-272): the shift that brings the LPS range to >= 256, except that everything under 8 takes 6.
struct Transitions {
    uint8_t lps[128], mps[128], renorm[32];
    Transitions()
    {
        for (int st = 0; st < 128; st++) {
            const int s = st >> 1, m = st & 1;
            lps[st] = s == 0 ? (uint8_t)(st ^ 1) : (uint8_t)((kNextLps[s] << 1) | m);
            mps[st] = s < 62 ? (uint8_t)(st + 2) : (uint8_t)st;
        }
        for (int i = 0; i < 32; i++) {
            int shift = 1;
            if (i == 0)
                shift = 6;
            else
                while (((8 * i) << shift) < 256) shift++;
            renorm[i] = (uint8_t)shift;
        }
    }
};
const Transitions kTrans;

// The arithmetic coder (src/bitstream.cpp:611-775: HM's TEncBinCABAC with its 32-bit low register, 23 spare bits
// and the deferred carry byte).
class Cabac {
public:
    Cabac(ByteSink &out, int qp) : out_(out)
    {
        auto fill = [&](int first, const uint8_t *init, int n) {
            for (int i = 0; i < n; i++) state_[first + i] = ctx_state(init[i], qp);
        };
        const uint8_t single[5] = {184, 184, 63, 141, 94};
        fill(kPartSize, single, 5);
        fill(kSigGroup, kInitSigGroup, 4);
        fill(kSig, kInitSig, 48);
        fill(kLastX, kInitLast, 30);
        fill(kLastY, kInitLast, 30);
        fill(kGreater1, kInitGreater1, 24);
        fill(kGreater2, kInitGreater2, 6);
    }
    void bin(uint32_t b, int ctx)  // xCabacEncodeBin
    {
        const uint32_t st = state_[ctx];
        const uint32_t lps = kRangeLps[st >> 1][(range_ >> 6) & 3];
        range_ -= lps;
        if (b != (st & 1u)) {
            const int shift = kTrans.renorm[lps >> 3];
            low_ = (low_ + range_) << shift;
            range_ = lps << shift;
            left_ -= shift;
            state_[ctx] = kTrans.lps[st];
        } else {
            state_[ctx] = kTrans.mps[st];
            if (range_ >= 256) return;  // nothing left `low`: no byte can be due
            low_ <<= 1;
            range_ <<= 1;
            left_--;
        }
        if (left_ < 12) drain();
    }
    void bypass(uint32_t b)  // xCabacEncodeBinEP
    {
        low_ <<= 1;
        if (b) low_ += range_;
        left_--;
        if (left_ < 12) drain();
    }
    void bypass_bits(uint32_t v, int n)  // xCabacEncodeBinsEP: n bins at once, most significant first
    {
        while (n > 8) {
            n -= 8;
            const uint32_t top = v >> n;
            low_ = (low_ << 8) + range_ * top;
            v -= top << n;
            left_ -= 8;
            if (left_ < 12) drain();
        }
        low_ = (low_ << n) + range_ * v;
        left_ -= n;
        if (left_ < 12) drain();
    }
    void terminate(uint32_t b)  // xCabacEncodeTerminatingBit
    {
        range_ -= 2;
        if (b) {
            low_ = (low_ + range_) << 7;
            range_ = 256;
            left_ -= 7;
        } else if (range_ < 256) {
            low_ <<= 1;
            range_ <<= 1;
            left_--;
        }
        drain();
    }
    void finish()  // xCabacFlush
    {
        if (low_ >> (32 - left_)) {
            out_.byte((uint32_t)held_ + 1u);
            for (; run_ > 1; run_--) out_.byte(0x00);
            low_ -= 1u << (32 - left_);
        } else {
            if (run_ > 0) out_.byte(held_);
            for (; run_ > 1; run_--) out_.byte(0xff);
        }
        out_.bits(low_ >> 8, 24 - left_);
    }

private:
    // testAndWriteOut (src/bitstream.cpp:657-688): bytes leave `low` as soon as fewer than 12 spare bits remain; a
    // byte is held back (with the run of 0xff bytes behind it) until a carry into it can no longer happen
    void drain()
    {
        while (left_ < 12) {
            const uint32_t lead = low_ >> (24 - left_);
            left_ += 8;
            low_ &= 0xffffffffu >> left_;
            if (lead == 0xff) {
                run_++;
            } else if (run_ > 0) {
                const uint32_t carry = lead >> 8;
                out_.byte((held_ + carry) & 0xffu);
                held_ = (uint8_t)lead;
                for (; run_ > 1; run_--) out_.byte((0xffu + carry) & 0xffu);
            } else {
                run_ = 1;
                held_ = (uint8_t)lead;
            }
        }
    }
    ByteSink &out_;
    uint8_t state_[kNumCtx];
    uint32_t low_ = 0, range_ = 510, run_ = 0;
    int left_ = 23;
    uint8_t held_ = 0xff;
};

// ------------------------------------------------------------------------------------------------------------
// Residual coding of one N x N block (N = 8, 16, 32), xEncodeCoeffNxN (src/encode.cpp:270-461).
// ------------------------------------------------------------------------------------------------------------
// up-right diagonal order of a K x K grid (K <= 8): every anti-diagonal from its bottom-left end to its top-right
// end (ausScanIdx4[SCAN_DIAG], g_sigLastScan8x8[SCAN_DIAG], g_sigLastScanCG32x32: src/table.cpp:300-305,700-716)
struct Diag {
    uint8_t x[64], y[64];
    explicit Diag(int k)
    {
        int n = 0;
        for (int d = 0; d < 2 * k - 1; d++)
            for (int yy = d < k ? d : k - 1; yy >= 0 && d - yy < k; yy--) {
                x[n] = (uint8_t)(d - yy);
                y[n] = (uint8_t)yy;
                n++;
            }
    }
};
const Diag kDiag2(2), kDiag4(4), kDiag8(8);

// prefix bin contexts of the last position, luma, by log2(N) - 2 (g_uiLastCtx, src/table.cpp:721-726)
const uint8_t kLastCtxLuma[4][10] = {{0, 1, 2, 2}, {3, 4, 5, 5, 2, 2}, {6, 7, 8, 8, 9, 9, 2, 2}, {10, 11, 12, 12, 13, 13, 14, 14, 2, 2}};
// Golomb-Rice parameter tables (src/table.cpp:728-744)
const uint8_t kRiceMax[5] = {7, 14, 26, 46, 78}, kRicePrefix[5] = {8, 7, 6, 5, 4};
const uint8_t kRiceNext[5][24] = {{0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4},
                                  {1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4},
                                  {2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4},
                                  {3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 4},
                                  {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4}};

// group of a last-position coordinate and the first coordinate of a group (g_uiGroupIdx / g_uiMinInGroup,
// src/table.cpp:719-720): 0 1 2 3 | 4-5 | 6-7 | 8-11 | 12-15 | 16-23 | 24-31
int last_group(int pos)
{
    if (pos < 4) return pos;
    int lg = 2;
    while ((pos >> (lg + 1)) != 0) lg++;
    return 2 * lg + ((pos >> (lg - 1)) & 1);
}
int group_first(int g) { return g < 4 ? g : (1 << (g >> 1)) + (g & 1) * (1 << ((g >> 1) - 1)); }

// Where a block's levels are: one pass over its rows, four levels (one 64-bit load) at a time.  row[y] has bit x set
// where level (x, y) is not zero -- two empty rows follow the block and no bit is set at or beyond x = N, so the
// neighbourhood tests of the significance contexts need no edge cases; has[] flags the 4x4 groups that hold a level.
struct BlockMap {
    uint32_t row[34];
    uint8_t has[64];
    bool any;
    void scan(const int16_t *blk, int stride, int n)
    {
        const int side = n >> 2;
        any = false;
        row[n] = row[n + 1] = 0;
        for (int gy = 0; gy < side; gy++) {
            uint32_t r[4] = {0, 0, 0, 0};
            for (int gx = 0; gx < side; gx++) {
                uint32_t in_group = 0;
                for (int k = 0; k < 4; k++) {
                    uint64_t q;
                    memcpy(&q, blk + (4 * gy + k) * stride + 4 * gx, 8);
                    if (q) {
                        const uint32_t m = (uint32_t)((q & 0xffffull) != 0) | (uint32_t)((q & 0xffff0000ull) != 0) << 1 |
                                           (uint32_t)((q & 0xffff00000000ull) != 0) << 2 | (uint32_t)((q >> 48) != 0) << 3;
                        r[k] |= m << (4 * gx);
                        in_group |= m;
                    }
                }
                has[gy * side + gx] = in_group != 0;
                any |= in_group != 0;
            }
            for (int k = 0; k < 4; k++) row[4 * gy + k] = r[k];
        }
    }
};

class ResidualCoder {
public:
    explicit ResidualCoder(Cabac &c) : c_(c) {}

    // blk: top-left of the block in its plane, `stride` elements per row; m: its map (m.any is true)
    void code(const int16_t *blk, int stride, int n, bool luma, BlockMap &m)
    {
        const int lg = n == 8 ? 3 : n == 16 ? 4 : 5, side = n >> 2;  // side: coefficient groups per row
        const Diag &groups = side == 2 ? kDiag2 : side == 4 ? kDiag4 : kDiag8;
        uint8_t *has = m.has;

        // the last level in coding order (group by group along the diagonal order of the groups, diagonal order
        // inside each: ausScanIdx8/16/32[SCAN_DIAG])
        int last_set = side * side - 1;
        while (!has[groups.y[last_set] * side + groups.x[last_set]]) last_set--;
        int last_in = 15;
        {
            const int gx = groups.x[last_set], gy = groups.y[last_set];
            while (!((m.row[4 * gy + kDiag4.y[last_in]] >> (4 * gx + kDiag4.x[last_in])) & 1u)) last_in--;
        }
        const int lx = 4 * groups.x[last_set] + kDiag4.x[last_in], ly = 4 * groups.y[last_set] + kDiag4.y[last_in];
        last_position(lx, ly, n, lg, luma);

        const int sig_group0 = kSigGroup + (luma ? 0 : 2), sig0 = kSig + (luma ? 0 : 27);
        uint32_t ones_before = 0;  // uiNumOne: levels > 1 seen in the groups coded so far (halved per group)
        for (int s = last_set; s >= 0; s--) {
            const int gx = groups.x[s], gy = groups.y[s];
            uint32_t rice = 0, signs = 0;
            int mag[16], cnt = 0;
            int i = 15;  // next position of this group, in reverse coding order
            const int16_t *g0 = blk + 4 * gy * stride + 4 * gx;
            if (s == last_set) {
                const int v = g0[kDiag4.y[last_in] * stride + kDiag4.x[last_in]];
                mag[cnt++] = v < 0 ? -v : v;
                signs = v < 0;
                i = last_in - 1;
            }
            // coded_sub_block_flag: implied for the group of the last level and for the DC group
            if (s == last_set || s == 0) {
                has[gy * side + gx] = 1;
            } else {
                const uint32_t right = gx < side - 1 && has[gy * side + gx + 1];
                const uint32_t below = gy < side - 1 && has[(gy + 1) * side + gx];
                c_.bin(has[gy * side + gx], sig_group0 + (int)(right | below));
            }
            if (has[gy * side + gx]) {
                // the group's neighbourhood once: bits x .. x + 5 of its four rows and of the two below
                uint32_t win[6];
                for (int k = 0; k < 6; k++) win[k] = (m.row[4 * gy + k] >> (4 * gx)) & 0x3fu;
                const SigGroup sg(gx, gy, n, lg, luma);
                for (; i >= 0; i--) {
                    const int dx = kDiag4.x[i], dy = kDiag4.y[i];
                    const uint32_t set = (win[dy] >> dx) & 1u;
                    // the first position of a group that was flagged and has shown no level yet is known to hold one
                    if (i != 0 || s == 0 || cnt) c_.bin(set, sig0 + sg.ctx(win, dx, dy));
                    if (set) {
                        const int v = g0[dy * stride + dx];
                        mag[cnt++] = v < 0 ? -v : v;
                        signs = (signs << 1) + (v < 0);
                    }
                }
            }
            if (!cnt) {
                ones_before >>= 1;
                continue;
            }
            // greater-than-1 flags of the first eight levels, one greater-than-2 flag, the signs, the remainders
            int set = (s > 0 && luma) ? 2 : 0;
            if (ones_before > 0) set++;
            ones_before >>= 1;
            const int g1 = kGreater1 + 4 * set + (luma ? 0 : 16);
            uint32_t c1 = 1;
            int first_big = -1;
            const int nflags = cnt < 8 ? cnt : 8;  // C1FLAG_NUMBER
            for (int k = 0; k < nflags; k++) {
                const uint32_t big = mag[k] > 1;
                c_.bin(big, g1 + (int)c1);
                if (big) {
                    c1 = 0;
                    if (first_big < 0) first_big = k;
                } else if (c1 != 0 && c1 < 3) {
                    c1++;
                }
            }
            if (c1 == 0 && first_big >= 0) c_.bin(mag[first_big] > 2, kGreater2 + set + (luma ? 0 : 4));
            c_.bypass_bits(signs, cnt);
            if (c1 == 0 || cnt > 8) {
                bool no_two_yet = true;
                for (int k = 0; k < cnt; k++) {
                    const int base = k < 8 ? (no_two_yet ? 3 : 2) : 1;
                    if (mag[k] >= base) remainder((uint32_t)(mag[k] - base), rice);
                    if (mag[k] >= 2) {
                        no_two_yet = false;
                        ones_before++;
                    }
                }
            }
        }
    }

private:
    // codeLastSignificantXY (src/encode.cpp:221-268), diagonal scan (no swap of the coordinates)
    void last_position(int x, int y, int n, int lg, bool luma)
    {
        const int gmax = last_group(n - 1);
        const int coord[2] = {x, y}, first[2] = {kLastX + (luma ? 0 : 15), kLastY + (luma ? 0 : 15)};
        for (int a = 0; a < 2; a++) {
            const int g = last_group(coord[a]);
            auto ctx = [&](int k) { return first[a] + (luma ? kLastCtxLuma[lg - 2][k] : (k >> (lg - 2))); };
            for (int k = 0; k < g; k++) c_.bin(1, ctx(k));
            if (g < gmax) c_.bin(0, ctx(g));
        }
        for (int a = 0; a < 2; a++) {
            const int g = last_group(coord[a]);
            if (g > 3) c_.bypass_bits((uint32_t)(coord[a] - group_first(g)), (g - 2) >> 1);
        }
    }

    // getSigCtxInc (src/encode.cpp:111-190) for N = 8 (blockType 3) and N = 16, 32, on the group's window of the
    // block's bitmap (win[k] bit j = position (4 gx + j, 4 gy + k)): the reference's `x < width - 1`,
    // `y < height - 1` ... guards are the map's empty margins.  What depends on the group alone is worked out once.
    struct SigGroup {
        int lg, off, dc_ctx, far_ctx, near_base, gx2, gy2;
        bool far, first;
        SigGroup(int gx, int gy, int n, int lg_, bool luma) : lg(lg_), gx2(2 * gx), gy2(2 * gy)
        {
            first = gx + gy == 0;
            if (lg == 3) {
                off = luma ? 9 : 6;
                dc_ctx = off + 10;
                far = false;
                far_ctx = near_base = 0;
            } else {
                off = luma ? 20 : 17;
                dc_ctx = off;
                far = gx + gy >= 3 * (n >> 4);
                far_ctx = luma ? 24 : 18;
                near_base = ((luma && !first) ? 4 : 1) + off;
            }
        }
        int ctx(const uint32_t *win, int dx, int dy) const
        {
            if (first && dx + dy == 0) return dc_ctx;
            if (lg == 3) {
                static const uint8_t map[16] = {0, 1, 2, 3, 4, 5, 6, 3, 8, 6, 6, 7, 9, 9, 7, 7};
                return off + map[4 * (gy2 + (dy >> 1)) + gx2 + (dx >> 1)];
            }
            if (far) return far_ctx;
            const uint32_t r0 = win[dy] >> dx, r1 = win[dy + 1] >> dx, r2 = win[dy + 2] >> dx;
            int cnt = (int)((r0 >> 1) & 1u) + (int)((r1 >> 1) & 1u) + (int)((r0 >> 2) & 1u);
            if ((dx || dy) && (((dx + 1) & 3) || ((dy + 2) & 3))) cnt += (int)(r1 & 1u);
            if (cnt < 4) cnt += (int)(r2 & 1u);
            return near_base + ((cnt + 1) >> 1);
        }
    };

    // xWriteGoRiceExGolomb + xWriteEpExGolomb (src/bitstream.cpp:777-828)
    void remainder(uint32_t v, uint32_t &rice)
    {
        const uint32_t vmax = kRiceMax[rice], word = v < vmax + 1 ? v : vmax + 1, q = word >> rice;
        const uint32_t plen = kRicePrefix[rice];
        const uint32_t nbins = q >= plen ? plen : q + 1, prefix = q >= plen ? (1u << nbins) - 1u : (1u << nbins) - 2u;
        c_.bypass_bits((prefix << rice) + word - (q << rice), (int)(nbins + rice));
        const uint32_t r = rice;
        rice = kRiceNext[r][v < 23 ? v : 23];
        if (v > vmax) {  // escape: 0-th order Exp-Golomb of the rest
            uint32_t rest = v - vmax - 1, pre = 0;
            int k = 0, nb = 0;
            while (rest >= (1u << k)) {
                pre = 2 * pre + 1;
                nb++;
                rest -= 1u << k;
                k++;
            }
            pre = 2 * pre;
            nb++;
            c_.bypass_bits((pre << k) | rest, nb + k);
        }
    }
    Cabac &c_;
};

int ilog2(uint32_t v)  // xLog2 (inc/utils.h:28-36): the bit length
{
    int n = 0;
    for (; v; v >>= 1) n++;
    return n;
}

// xWriteSPS (src/bitstream.cpp:112-188) with xDefaultParams' values (src/params.cpp:27-56)
void write_sps(ByteSink &o, int width, int height, int cu)
{
    const int depth = 1, tu_lo = 2, tu_hi = 5;  // ucMaxCUDepth, ucQuadtreeTULog2MinSize / MaxSize
    o.bits(0, 8);   // profile_idc
    o.bits(0, 8);   // reserved_zero_8bits
    o.bits(0, 8);   // level_idc
    o.ue(0);        // seq_parameter_set_id
    o.ue(1);        // chroma_format_idc: 4:2:0
    o.bits(0, 3);   // max_temporal_layers_minus1
    o.ue((uint32_t)width);
    o.ue((uint32_t)height);
    o.flag(0);      // pic_cropping_flag
    o.ue(0);        // bit_depth_luma_minus8
    o.ue(0);        // bit_depth_chroma_minus8
    o.flag(0);      // pcm_enabled_flag
    o.flag(0);      // qpprime_y_zero_transquant_bypass_flag (LOSSLESS_CODING, inc/config.h:107)
    o.ue(8 - 4);    // log2_max_pic_order_cnt_lsb_minus4 (ucBitsForPOC = 8)
    o.ue(1);        // max_dec_pic_buffering
    o.ue(0);        // num_reorder_pics
    o.ue(0);        // max_latency_increase
    const int log2_min_cu = ilog2((uint32_t)(cu >> (depth - 1))) - 1;
    o.flag(1);      // restricted_ref_pic_lists_flag
    o.flag(0);      // lists_modification_present_flag
    o.ue((uint32_t)(log2_min_cu - 3));
    o.ue(depth - 1);           // log2_diff_max_min_coding_block_size
    o.ue(tu_lo - 2);           // log2_min_transform_block_size_minus2
    o.ue(tu_hi - tu_lo);       // log2_diff_max_min_transform_block_size
    if (log2_min_cu == 3) o.flag(1);  // DisableInter4x4
    o.ue(0);        // max_transform_hierarchy_depth_inter (depth 1, minus 1)
    o.ue(0);        // max_transform_hierarchy_depth_intra
    o.flag(0);      // scaling_list_enabled_flag
    o.flag(0);      // chroma_pred_from_luma_enabled_flag
    o.flag(0);      // deblocking_filter_in_aps_enabled_flag
    o.flag(1);      // seq_loop_filter_across_slices_enabled_flag
    o.flag(1);      // asymmetric_motion_partitions_enabled_flag
    o.flag(1);      // non_square_quadtree_enabled_flag
    o.flag(0);      // sample_adaptive_offset_enabled_flag
    o.flag(0);      // adaptive_loop_filter_enabled_flag
    o.flag(0);      // temporal_id_nesting_flag
    o.ue(1);        // num_short_term_ref_pic_sets, then the set itself (xWriteShortTermRefPicSet, :101-110)
    o.flag(0);      //   inter_ref_pic_set_prediction_flag
    o.ue(1);        //   num_negative_pics
    o.ue(0);        //   num_positive_pics
    o.ue(0);        //   delta_poc_s0_minus1
    o.flag(1);      //   used_by_curr_pic_s0_flag
    o.flag(0);      // long_term_ref_pics_present_flag
    // one AMVPMode flag per depth, plus the levels between the smallest CU's quarter and ... the reference compares
    // a size with a log2 here (:171-177, its own CHECK_ME) and the count is reproduced as written
    const int split = cu >> depth;
    int namvp = depth;
    if (split > tu_lo) namvp += ilog2((uint32_t)(split - 1)) - ilog2((uint32_t)tu_lo);
    for (int i = 0; i < namvp; i++) o.flag(1);
    o.bits(0, 2);   // tiles_or_entropy_coding_sync_idc
    o.flag(0);      // sps_extension_flag
    o.trailing();
}

// xWritePPS (src/bitstream.cpp:190-229)
void write_pps(ByteSink &o)
{
    o.ue(0);        // pic_parameter_set_id
    o.ue(0);        // seq_parameter_set_id
    o.flag(0);      // sign_data_hiding_flag (bSignHideFlag = FALSE)
    o.flag(1);      // cabac_init_present_flag
    o.bits(0, 3);   // num_ref_idx_l0_default_active_minus1 (ucMaxNumRefFrames = 1)
    o.bits(0, 3);   // num_ref_idx_l1_default_active_minus1
    o.se(0);        // pic_init_qp_minus26
    o.flag(0);      // constrained_intra_pred_flag
    o.flag(1);      // enable_temporal_mvp_flag (bEnableTMVPFlag = TRUE)
    o.bits(0, 2);   // slice_granularity
    o.ue(0);        // max_cu_qp_delta_depth
    o.se(0);        // cb_qp_offset
    o.se(0);        // cr_qp_offset
    o.flag(0);      // weighted_pred_flag
    o.bits(0, 2);   // weighted_bipred_idc
    o.flag(0);      // output_flag_present_flag
    o.flag(1);      // deblocking_filter_control_present_flag
    o.ue(0);        // log2_parallel_merge_level_minus2
    o.flag(0);      // pps_extension_flag
    o.trailing();
}

// xWriteSliceHeader (src/bitstream.cpp:231-295), SLICE_I
void write_slice_header(ByteSink &o, int poc, int qp)
{
    o.flag(1);      // first_slice_in_pic_flag
    o.ue(0);        // slice_type: SLICE_I = 0 (inc/x265.h:40)
    o.flag(0);      // lightweight_slice_flag
    o.ue(0);        // pic_parameter_set_id
    if (poc == 0) {
        o.ue(0);    // idr_pic_id
        o.flag(0);  // no_output_of_prior_pics_flag
    } else {
        o.bits((uint32_t)poc & 255u, 8);  // pic_order_cnt_lsb
        o.flag(1);  // short_term_ref_pic_set_pps_flag
        o.ue(0);    // short_term_ref_pic_set_idx
    }
    o.se(qp - 26);  // slice_qp_delta
    o.flag(1);      // loop_filter_disable (bLoopFilterDisable = TRUE)
    o.ue(0);        // maxNumMergeCand: MRG_MAX_NUM_CANDS - ucMaxNumMergeCand = 5 - 5
    o.flag(0);      // encodeTileMarkerFlag
    o.align(1);
}

}  // namespace

extern "C" size_t x265amd_intra_stream_bound(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    // headers + at most ~5 bytes per level of full Int16 range (24 prefix / escape bins + sign + context bins),
    // times 4/3 for emulation prevention
    return 256 + (size_t)width * height * 3 / 2 * 7;
}

extern "C" int x265amd_intra_write_picture(const x265amd_intra_syntax *s, uint8_t *out, size_t cap, size_t *bytes)
{
    if (!s || !out || !bytes) return X265AMD_ERR_ARG;
    const int w = s->width, h = s->height, cu = s->cu;
    if ((cu != 16 && cu != 32) || w <= 0 || h <= 0 || w % cu || h % cu || w > 65535 || h > 65535) return X265AMD_ERR_ARG;
    if (s->qp < 0 || s->qp > 51 || s->poc < 0) return X265AMD_ERR_ARG;
    if (!s->level_y || !s->level_u || !s->level_v || !s->mode_y || !s->mode_c) return X265AMD_ERR_ARG;
    if (s->stride_y < w || s->stride_c < w / 2) return X265AMD_ERR_ARG;
    const int ncx = w / cu, ncy = h / cu, half = cu / 2;
    for (int i = 0; i < ncx * ncy; i++)
        if (s->mode_y[i] > 34) return X265AMD_ERR_ARG;

    ByteSink o(out, cap);
    // parameter sets in front of every picture, as xEncEncode writes them (src/encode.cpp:626-639); no SEI
    // (CHECK_SEI is not defined in the reference's build)
    o.raw_word(0x01000000u);
    o.bits(0x47, 8);
    o.bits(0x01, 8);
    write_sps(o, w, h, cu);
    o.raw_word(0x01000000u);
    o.bits(0x48, 8);
    o.bits(0x01, 8);
    write_pps(o);
    o.raw_word(s->poc == 0 ? 0x45010000u : 0x41010000u);  // :660
    o.bits(0x01, 8);
    write_slice_header(o, s->poc, s->qp);

    Cabac cabac(o, s->qp);
    ResidualCoder res(cabac);
    BlockMap map_y, map_u, map_v;
    for (int cy = 0; cy < ncy; cy++)
        for (int cx = 0; cx < ncx; cx++) {
            const int idx = cy * ncx + cx, my = s->mode_y[idx], mc = s->mode_c[idx];
            // most probable modes (xEncIntraLoadRef, src/encode.cpp:1268-1293).  The mode above never takes part:
            // `bT && uiY` is evaluated with uiY = 0 there, so the top candidate is always DC.
            const int left = cx > 0 ? s->mode_y[idx - 1] : 1, top = 1;
            int mpm[3];
            if (left == top) {
                mpm[0] = 0, mpm[1] = 1, mpm[2] = 26;  // (left > 1 cannot happen with top = DC)
            } else {
                mpm[0] = left, mpm[1] = top;
                mpm[2] = (left != 0 && top != 0) ? 0 : (left + top < 2 ? 26 : 1);
            }
            // the chroma candidates of src/encode.cpp:781-793: planar, vertical, horizontal, DC (the one that equals
            // the luma mode replaced by 34), LM (never chosen: bUseLMChroma = FALSE), the luma mode itself
            int cand[6] = {0, 26, 10, 1, -1, my};
            for (int i = 0; i < 4; i++)
                if (cand[i] == my) {
                    cand[i] = 34;
                    break;
                }
            int mci = -1;
            for (int k = 0; k < 6 && mci < 0; k++)
                if (cand[k] == mc) mci = k;
            if (mci < 0) return X265AMD_ERR_ARG;

            const int16_t *by = s->level_y + (size_t)(cy * cu) * s->stride_y + cx * cu;
            const int16_t *bc[2] = {s->level_u + (size_t)(cy * half) * s->stride_c + cx * half,
                                    s->level_v + (size_t)(cy * half) * s->stride_c + cx * half};
            map_y.scan(by, s->stride_y, cu);
            map_u.scan(bc[0], s->stride_c, half);
            map_v.scan(bc[1], s->stride_c, half);
            const uint32_t cbf_y = map_y.any, cbf_u = map_u.any, cbf_v = map_v.any;

            // xWriteCU (src/encode.cpp:474-580)
            cabac.bin(1, kPartSize);  // 2Nx2N
            int which = -1;
            for (int k = 2; k >= 0; k--)
                if (my == mpm[k]) which = k;
            cabac.bin(which >= 0, kPrevIntraLuma);
            if (which == 0) {
                cabac.bypass(0);
            } else if (which > 0) {
                cabac.bypass_bits((uint32_t)which + 1u, 2);
            } else {
                int a = mpm[0], b = mpm[1], c = mpm[2], t;
                if (a > b) t = a, a = b, b = t;
                if (a > c) t = a, a = c, c = t;
                if (b > c) t = b, b = c, c = t;
                int rem = my;
                if (rem > c) rem--;
                if (rem > b) rem--;
                if (rem > a) rem--;
                cabac.bypass_bits((uint32_t)rem, 5);
            }
            cabac.bin(mci != 5, kChromaPred);
            if (mci < 4) cabac.bypass_bits((uint32_t)mci, 2);
            cabac.bin(cbf_u, kCbfChroma);
            cabac.bin(cbf_v, kCbfChroma);
            cabac.bin(cbf_y, kCbfLuma);
            if (cbf_y) res.code(by, s->stride_y, cu, true, map_y);
            if (cbf_u) res.code(bc[0], s->stride_c, half, false, map_u);
            if (cbf_v) res.code(bc[1], s->stride_c, half, false, map_v);
            cabac.terminate(cy == ncy - 1 && cx == ncx - 1);
        }
    cabac.finish();
    // xWriteSliceEnd (src/bitstream.cpp:297-304)
    o.flag(1);
    o.align(0);
    o.trailing();
    *bytes = o.size();
    return o.overflow() ? X265AMD_ERR_BUFFER : X265AMD_OK;
}
