// x265amd_host.hpp -- C++ host side of the MI355X ME + transform/quant path, above the C-ABI only.
//
// What the reference's encode.cpp would hold for P pictures if it had them (src/encode.cpp:581-584 is
// `assert(0)`): the frame / reference bookkeeping of xEncEncode (refn[] ring and its rotation,
// src/encode.cpp:619-621, 965-970; frame layout Y,U,V in one W*H*3/2 block, stride = width,
// src/encode.cpp:41-45), one search launch and three transform/quant launches per picture.  Needs nothing
// but include/x265_amd.h and libx265amd.so: no HIP headers, no torch.  (x265_amd/pipeline.py is the same
// loop in Python for the tests and the bench.)
#pragma once
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <memory>
#include <utility>
#include <vector>

#include "x265_amd.h"

namespace x265amd_host {

struct Error : std::runtime_error {
    int status;
    Error(int st, const char *where)
        : std::runtime_error(std::string(where) + ": " + x265amd_strerror(st) + " (status " + std::to_string(st) + ")"),
          status(st)
    {
    }
};
inline void check(int st, const char *where)
{
    if (st != X265AMD_OK) throw Error(st, where);
}

class DeviceBuffer {  // xEncInit's MALLOC / xEncFree's FREE (src/encode.cpp:40-46, 54-62), on the device
  public:
    explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(x265amd_malloc(&p_, bytes ? bytes : 16), "x265amd_malloc"); }
    ~DeviceBuffer() { x265amd_free(p_); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    template <class T = uint8_t> T *as(size_t byte_off = 0) const { return reinterpret_cast<T *>(static_cast<uint8_t *>(p_) + byte_off); }
    size_t bytes() const { return bytes_; }

  private:
    void *p_ = nullptr;
    size_t bytes_;
};

class PinnedBuffer {  // page-locked host memory for what crosses PCIe every picture
  public:
    explicit PinnedBuffer(size_t bytes) : bytes_(bytes) { check(x265amd_malloc_host(&p_, bytes ? bytes : 16), "x265amd_malloc_host"); }
    ~PinnedBuffer() { x265amd_free_host(p_); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    template <class T = uint8_t> T *as() const { return static_cast<T *>(p_); }
    size_t bytes() const { return bytes_; }

  private:
    void *p_ = nullptr;
    size_t bytes_;
};

// Closed-loop P pictures for `nseg` independent GOP segments (the unit that shards across GPUs): picture t
// of every segment goes down in one launch per kernel.  Luma: 64x64 CTU search on the previous
// reconstruction, 32x32 TUs.  Chroma 4:2:0: the luma vector >> 1, 16x16 TUs at g_aucChromaScale[qp].
class GopEncoder {
  public:
    static constexpr int kMaxRefNum = 1;  // MAX_REF_NUM, inc/config.h

    // `stream`: a stream from x265amd_stream_create (nullptr = the default stream).  Encoders on different
    // streams overlap: one's PCIe copies run under the other's kernels.
    GopEncoder(int width, int height, int nseg, int range, int qp, int device = 0, void *stream = nullptr)
        : w_(width), h_(height), nseg_(nseg), range_(range), qp_(qp), stream_(stream)
    {
        if (width <= 0 || height <= 0 || width % 64 || height % 64 || nseg < 1)  // src/params.cpp:73-83 exit()s
            throw Error(X265AMD_ERR_ARG, "GopEncoder: width and height must be multiples of the 64x64 CTU");
        check(x265amd_init(device), "x265amd_init");
        qpc_ = x265amd_chroma_qp(qp);
        if (qpc_ < 0) throw Error(qpc_, "x265amd_chroma_qp");
        ysz_ = (size_t)width * height;
        csz_ = ysz_ / 4;
        fsz_ = ysz_ + 2 * csz_;
        nctu_ = (size_t)(width / 64) * (height / 64);
        for (int i = 0; i <= kMaxRefNum; i++) refn_.emplace_back(new DeviceBuffer(fsz_ * nseg));
        cur_.reset(new DeviceBuffer(fsz_ * nseg));
        level_.reset(new DeviceBuffer(2 * fsz_ * nseg));
        mv_.reset(new DeviceBuffer(sizeof(x265amd_mv) * nctu_ * nseg));
        ntu_ = (size_t)(width / 32) * (height / 32);  // luma 32x32 TUs; the chroma planes have as many 16x16 TUs
        cbf_.reset(new DeviceBuffer(4 * ntu_ * nseg * 3));
        ws_bytes_ = x265amd_me_workspace_bytes(width, height, range, nseg, 1);
        ws_.reset(new DeviceBuffer(ws_bytes_));
    }

    size_t frame_bytes() const { return fsz_; }
    size_t ctus() const { return nctu_; }
    size_t tus() const { return ntu_; }
    int poc() const { return poc_; }
    bool exhaustive_fallback() const { return brute_; }

    // The reconstruction of each segment's I picture when it is made elsewhere (encode_i below makes it here).
    void set_reference(const uint8_t *yuv)
    {
        check(x265amd_memcpy_h2d(refn_[kMaxRefNum]->as(), yuv, fsz_ * nseg_, stream_), "x265amd_memcpy_h2d");
        poc_++;
        rotate();
    }

    // The I picture of every segment through the library's intra path (x265amd_intra_encode_frames: the segments
    // are its batch of independent frames), CU = PU = TU = cu (16 or 32).  Levels land where encode_p leaves them;
    // the reconstruction becomes refn[0].  Asynchronous; fetch() waits.  fetch_modes() brings the modes back.
    void encode_i(const uint8_t *yuv, int cu = 32)
    {
        if ((cu != 16 && cu != 32)) throw Error(X265AMD_ERR_ARG, "GopEncoder::encode_i: CU 16 or 32");
        const size_t ncu = (size_t)(w_ / cu) * (h_ / cu);
        if (!pred_) pred_.reset(new DeviceBuffer(fsz_ * nseg_));
        if (!mode_ || mode_->bytes() < 2 * ncu * nseg_) mode_.reset(new DeviceBuffer(2 * ncu * nseg_));
        icu_ = cu;
        last_intra_ = true;  // its levels come without per-TU sums at encode_p's TU sizes
        check(x265amd_memcpy_h2d(cur_->as(), yuv, fsz_ * nseg_, stream_), "x265amd_memcpy_h2d");
        poc_++;
        x265amd_intra_frame f{};
        f.width = w_, f.height = h_, f.cu = cu, f.qp = qp_, f.stride_y = w_, f.stride_c = w_ / 2;
        uint8_t *rec = refn_[kMaxRefNum]->as();
        f.d_src_y = cur_->as(), f.d_src_u = cur_->as(ysz_), f.d_src_v = cur_->as(ysz_ + csz_);
        f.d_rec_y = rec, f.d_rec_u = rec + ysz_, f.d_rec_v = rec + ysz_ + csz_;
        f.d_pred_y = pred_->as(), f.d_pred_u = pred_->as(ysz_), f.d_pred_v = pred_->as(ysz_ + csz_);
        f.d_level_y = level_->as<int16_t>(), f.d_level_u = level_->as<int16_t>(2 * ysz_), f.d_level_v = level_->as<int16_t>(2 * (ysz_ + csz_));
        f.d_mode_y = mode_->as(), f.d_mode_c = mode_->as(ncu * nseg_);
        check(x265amd_intra_encode_frames(&f, nseg_, (int64_t)fsz_, (int64_t)fsz_, (int64_t)ncu, stream_), "x265amd_intra_encode_frames");
        rotate();
    }
    // luma / chroma mode per CU of the last encode_i, [nseg][ncu] each (after sync())
    void fetch_modes(uint8_t *mode_y, uint8_t *mode_c)
    {
        if (!mode_ || !icu_) throw Error(X265AMD_ERR_ARG, "GopEncoder::fetch_modes: no encode_i yet");
        const size_t ncu = (size_t)(w_ / icu_) * (h_ / icu_);
        check(x265amd_memcpy_d2h(mode_y, mode_->as(), ncu * nseg_, stream_), "x265amd_memcpy_d2h");
        check(x265amd_memcpy_d2h(mode_c, mode_->as(ncu * nseg_), ncu * nseg_, stream_), "x265amd_memcpy_d2h");
        sync();
    }

    // One P picture per segment: yuv = nseg frames, Y,U,V each.  Asynchronous; fetch() waits.
    void encode_p(const uint8_t *yuv)
    {
        check(x265amd_memcpy_h2d(cur_->as(), yuv, fsz_ * nseg_, stream_), "x265amd_memcpy_h2d");
        poc_++;
        last_intra_ = false;
        const uint8_t *src = cur_->as(), *ref = refn_[0]->as();
        uint8_t *rec = refn_[kMaxRefNum]->as();
        x265amd_mv *mv = mv_->as<x265amd_mv>();
        if (!brute_) {
            const int st = x265amd_me_fullsearch_sea(src, ref, w_, h_, w_, 64, range_, nseg_, (int64_t)fsz_, (int64_t)fsz_, 1, 0,
                                                     mv, ws_->as(), ws_bytes_, stream_);
            if (st == X265AMD_ERR_LDS)
                brute_ = true;  // window + schedule beyond the LDS: every-candidate kernel, same output
            else
                check(st, "x265amd_me_fullsearch_sea");
        }
        if (brute_)
            check(x265amd_me_fullsearch(src, ref, w_, h_, w_, 64, range_, nseg_, (int64_t)fsz_, (int64_t)fsz_, 1, 0, mv, stream_),
                  "x265amd_me_fullsearch");
        // Stage 3a/3b of src/encode.cpp:863-925 for the three planes
        tq_plane(0, 0, w_, h_, 32, qp_, 64, 0);
        tq_plane(ysz_, 1, w_ / 2, h_ / 2, 16, qpc_, 32, 1);
        tq_plane(ysz_ + csz_, 2, w_ / 2, h_ / 2, 16, qpc_, 32, 1);
        (void)rec;
        rotate();  // src/encode.cpp:965-970: the new reconstruction becomes refn[0]
    }

    // Vectors [nseg][nctu], levels [nseg][W*H*3/2] (Y,U,V like the picture), reconstruction [nseg][W*H*3/2],
    // per-TU sum|level| [3][nseg][ntu] (what sets bCbf, src/encode.cpp:888).  Null pointers are skipped.
    void fetch(x265amd_mv *mv, int16_t *level, uint8_t *rec, uint32_t *abs_sum = nullptr)
    {
        fetch_async(mv, level, rec, abs_sum);
        sync();
    }
    void sync() { check(x265amd_stream_sync(stream_), "x265amd_stream_sync"); }
    // the copies only: valid after sync(); the host buffers should be pinned (PinnedBuffer) to be asynchronous
    void fetch_async(x265amd_mv *mv, int16_t *level, uint8_t *rec, uint32_t *abs_sum = nullptr)
    {
        if (mv) check(x265amd_memcpy_d2h(mv, mv_->as(), sizeof(x265amd_mv) * nctu_ * nseg_, stream_), "x265amd_memcpy_d2h");
        if (level) check(x265amd_memcpy_d2h(level, level_->as(), 2 * fsz_ * nseg_, stream_), "x265amd_memcpy_d2h");
        if (rec) check(x265amd_memcpy_d2h(rec, refn_[0]->as(), fsz_ * nseg_, stream_), "x265amd_memcpy_d2h");
        if (abs_sum) check(x265amd_memcpy_d2h(abs_sum, cbf_->as(), 4 * ntu_ * nseg_ * 3, stream_), "x265amd_memcpy_d2h");
    }

    // ---- the levels in packed form (x265amd_pack_levels): what crosses PCIe is the non-zero levels only ----
    // table: [Y, U, V][nseg][ntu] entries of {first word, number of words}, then ONE more word: how many words the
    // picture needed in all.  words: (position in the TU's raster << 16) | (level & 0xffff), a TU's own in raster
    // order (luma TUs 32x32, chroma 16x16, as encode_p codes them).
    size_t packed_table_words() const { return 2 * 3 * ntu_ * nseg_ + 1; }
    // Room for every level of a picture (the worst case); a host that knows its QP can give far less: a picture
    // that needs more than `capacity` comes back with the count it needed and the TUs beyond left out (the
    // dense fetch() is still there for it).
    size_t packed_words_worst() const { return fsz_ * nseg_; }

    // Launches the compaction of the last encode_p's (or encode_i's) levels and the copies of the table, of the
    // vectors and of the first `expect` words (0 = what the last picture needed, plus a quarter).  After sync(),
    // fetch_packed_finish() returns the number of words and brings the rest if the picture needed more than
    // `expect`.  Host buffers pinned, as for fetch_async.
    //
    // `direct`: mv, table and words are page-locked (PinnedBuffer) and the kernels write them across the link
    // themselves -- no copy after the kernels and no guess at the number of words (1080p x 16 segments, one
    // stream: 2.23 -> 2.12 ms per picture), and, with several encoders on streams and host threads of their own,
    // no copy-engine download in the way of the others' uploads (1.44 ms: x265amd_copy_to_host in x265_amd.h);
    // all words are there after sync().
    void fetch_packed_async(x265amd_mv *mv, uint32_t *table, uint32_t *words, size_t capacity, size_t expect = 0, bool direct = false)
    {
        if (!table || (!words && capacity)) throw Error(X265AMD_ERR_ARG, "GopEncoder::fetch_packed_async");
        if (capacity > 0xffffffffu) capacity = 0xffffffffu;
        if (!ptable_) ptable_.reset(new DeviceBuffer(4 * packed_table_words()));
        uint32_t *tb = ptable_->as<uint32_t>(), *cursor = tb + packed_table_words() - 1;
        if (direct) {
            check(x265amd_pack_levels_begin(cursor, stream_), "x265amd_pack_levels_begin");
            pack_planes(table, words, capacity, cursor);
            if (mv) check(x265amd_copy_to_host(mv, mv_->as(), sizeof(x265amd_mv) * nctu_ * nseg_, stream_), "x265amd_copy_to_host");
            check(x265amd_copy_to_host(table + packed_table_words() - 1, cursor, 4, stream_), "x265amd_copy_to_host");
            ptab_host_ = table, pwords_host_ = words, pcap_ = capacity, sent_ = capacity;
            return;
        }
        if (!pwords_ || pwords_->bytes() < 4 * capacity) pwords_.reset(new DeviceBuffer(4 * capacity));
        check(x265amd_pack_levels_begin(cursor, stream_), "x265amd_pack_levels_begin");
        pack_planes(tb, pwords_->as<uint32_t>(), capacity, cursor);
        if (mv) check(x265amd_memcpy_d2h(mv, mv_->as(), sizeof(x265amd_mv) * nctu_ * nseg_, stream_), "x265amd_memcpy_d2h");
        check(x265amd_memcpy_d2h(table, tb, 4 * packed_table_words(), stream_), "x265amd_memcpy_d2h");
        if (!expect) expect = last_words_ ? last_words_ + last_words_ / 4 + 1024 : capacity / 8;
        sent_ = expect < capacity ? expect : capacity;
        if (sent_) check(x265amd_memcpy_d2h(words, pwords_->as(), 4 * sent_, stream_), "x265amd_memcpy_d2h");
        ptab_host_ = table, pwords_host_ = words, pcap_ = capacity;
    }
    // after sync(): the number of words the picture needed (> capacity: incomplete, see above)
    size_t fetch_packed_finish()
    {
        if (!ptab_host_) throw Error(X265AMD_ERR_ARG, "GopEncoder::fetch_packed_finish: no fetch_packed_async");
        const size_t need = ptab_host_[packed_table_words() - 1], have = need < pcap_ ? need : pcap_;
        if (have > sent_) {
            check(x265amd_memcpy_d2h(pwords_host_ + sent_, pwords_->as(4 * sent_), 4 * (have - sent_), stream_), "x265amd_memcpy_d2h");
            sync();
        }
        last_words_ = need;
        ptab_host_ = nullptr;
        return need;
    }
    size_t fetch_packed(x265amd_mv *mv, uint32_t *table, uint32_t *words, size_t capacity)
    {
        fetch_packed_async(mv, table, words, capacity);
        sync();
        return fetch_packed_finish();
    }
    // The dense levels of plane `pl` (0 Y, 1 U, 2 V) of segment `seg` from the packed form: out = [h][w] of that plane.
    void unpack_plane(const uint32_t *table, const uint32_t *words, int pl, int seg, int16_t *out) const
    {
        const int sh = pl ? 1 : 0, pw = w_ >> sh, ph = h_ >> sh, n = 32 >> sh, tus_x = pw / n;
        std::fill(out, out + (size_t)pw * ph, (int16_t)0);
        const uint32_t *tb = table + 2 * ((size_t)pl * nseg_ + seg) * ntu_;
        for (size_t t = 0; t < ntu_; t++) {
            const uint32_t *wd = words + tb[2 * t];
            int16_t *blk = out + (t / tus_x) * n * (size_t)pw + (t % tus_x) * n;
            for (uint32_t i = 0; i < tb[2 * t + 1]; i++) {
                const uint32_t pos = wd[i] >> 16;
                blk[(pos / n) * (size_t)pw + pos % n] = (int16_t)(wd[i] & 0xffffu);
            }
        }
    }

  private:
    void pack_planes(uint32_t *table, uint32_t *words, size_t capacity, uint32_t *cursor)
    {
        const size_t offs[3] = {0, ysz_, ysz_ + csz_};
        for (int pl = 0; pl < 3; pl++) {
            const int sh = pl ? 1 : 0;
            check(x265amd_pack_levels(level_->as<int16_t>() + offs[pl], w_ >> sh, h_ >> sh, w_ >> sh, 32 >> sh, nseg_, (int64_t)fsz_,
                                      last_intra_ ? nullptr : cbf_->as<uint32_t>() + (size_t)pl * ntu_ * nseg_, (int64_t)ntu_,
                                      table + 2 * (size_t)pl * ntu_ * nseg_, words, (uint32_t)capacity, cursor, stream_),
                  "x265amd_pack_levels");
        }
    }
    void rotate()
    {
        std::unique_ptr<DeviceBuffer> tmp = std::move(refn_[kMaxRefNum]);
        for (int i = kMaxRefNum; i > 0; i--) refn_[i] = std::move(refn_[i - 1]);
        refn_[0] = std::move(tmp);
    }

    void tq_plane(size_t off, int plane, int pw, int ph, int tu, int qp, int mv_ctu, int mv_shift)
    {
        x265amd_tq_desc t = {};
        t.width = pw;
        t.height = ph;
        t.tu = tu;
        t.qp = qp;
        t.slice_type = X265AMD_SLICE_P;
        t.nframes = nseg_;
        t.d_src = cur_->as() + off;
        t.src_stride = pw;
        t.src_frame_stride = (int64_t)fsz_;
        t.d_pred = refn_[0]->as() + off;  // the reference picture: motion compensation is fused into the load
        t.pred_stride = pw;
        t.pred_frame_stride = (int64_t)fsz_;
        t.d_mv = mv_->as<x265amd_mv>();
        t.mv_ctu = mv_ctu;
        t.mv_frame_stride = (int64_t)nctu_;
        t.mv_shift = mv_shift;
        t.d_level = level_->as<int16_t>() + off;
        t.level_stride = pw;
        t.level_frame_stride = (int64_t)fsz_;
        t.d_abs_sum = cbf_->as<uint32_t>() + (size_t)plane * ntu_ * nseg_;
        t.abs_sum_frame_stride = (int64_t)ntu_;
        t.d_rec = refn_[kMaxRefNum]->as() + off;
        t.rec_stride = pw;
        t.rec_frame_stride = (int64_t)fsz_;
        check(x265amd_tq_roundtrip(&t, stream_), "x265amd_tq_roundtrip");
    }

    int w_, h_, nseg_, range_, qp_, qpc_ = 0, poc_ = -1;
    void *stream_ = nullptr;
    bool brute_ = false;
    size_t ysz_ = 0, csz_ = 0, fsz_ = 0, nctu_ = 0, ntu_ = 0, ws_bytes_ = 0;
    std::vector<std::unique_ptr<DeviceBuffer>> refn_;
    std::unique_ptr<DeviceBuffer> cur_, level_, mv_, cbf_, ws_;
    std::unique_ptr<DeviceBuffer> pred_, mode_;  // encode_i's prediction planes and per-CU modes, made on first use
    std::unique_ptr<DeviceBuffer> ptable_, pwords_;  // fetch_packed's table (+ count) and words, made on first use
    uint32_t *ptab_host_ = nullptr, *pwords_host_ = nullptr;
    size_t pcap_ = 0, sent_ = 0, last_words_ = 0;
    int icu_ = 0;
    bool last_intra_ = false;
};

// Intra pictures: the pixel path of xEncEncode's CU loop (src/encode.cpp:668-941) on the GPU, frame by frame --
// neighbour samples, luma and chroma mode decisions, prediction, transform stage, reconstruction.  What comes back
// is what Stage 4 (xWriteCU, the entropy coder: host code, write_picture() below) consumes: per CU the luma mode,
// the chroma mode, the coded-block sums, and the quantised levels; plus the reconstruction, over which the
// reference takes its MD5 SEI (src/encode.cpp:945-950).
class IntraEncoder {
  public:
    // `batch`: how many independent frames one encode() may take (they share every anti-diagonal step on the GPU:
    // x265amd_intra_encode_frames); the device planes are allocated for that many.
    IntraEncoder(int width, int height, int cu, int qp, int device = 0, void *stream = nullptr, int batch = 1)
        : w_(width), h_(height), cu_(cu), qp_(qp), batch_(batch), stream_(stream)
    {
        if (width <= 0 || height <= 0 || (cu != 16 && cu != 32) || width % cu || height % cu || batch < 1)  // src/params.cpp:73-77
            throw Error(X265AMD_ERR_ARG, "IntraEncoder: CU 16 or 32, width and height multiples of it");
        check(x265amd_init(device), "x265amd_init");
        ysz_ = (size_t)width * height;
        csz_ = ysz_ / 4;
        fsz_ = ysz_ + 2 * csz_;
        ncu_ = (size_t)(width / cu) * (height / cu);
        const size_t nb = (size_t)batch;
        src_.reset(new DeviceBuffer(fsz_ * nb));
        rec_.reset(new DeviceBuffer(fsz_ * nb));
        pred_.reset(new DeviceBuffer(fsz_ * nb));
        level_.reset(new DeviceBuffer(2 * fsz_ * nb));
        mode_.reset(new DeviceBuffer(2 * ncu_ * nb));      // [luma, chroma][batch][ncu]
        sum_.reset(new DeviceBuffer(4 * 3 * ncu_ * nb));   // [Y, U, V][batch][ncu]
        f_ = x265amd_intra_frame{};
        f_.width = width, f_.height = height, f_.cu = cu, f_.qp = qp, f_.stride_y = width, f_.stride_c = width / 2;
        f_.d_src_y = src_->as(), f_.d_src_u = src_->as(ysz_), f_.d_src_v = src_->as(ysz_ + csz_);
        f_.d_rec_y = rec_->as(), f_.d_rec_u = rec_->as(ysz_), f_.d_rec_v = rec_->as(ysz_ + csz_);
        f_.d_pred_y = pred_->as(), f_.d_pred_u = pred_->as(ysz_), f_.d_pred_v = pred_->as(ysz_ + csz_);
        f_.d_level_y = level_->as<int16_t>(), f_.d_level_u = level_->as<int16_t>(2 * ysz_), f_.d_level_v = level_->as<int16_t>(2 * (ysz_ + csz_));
        f_.d_mode_y = mode_->as(), f_.d_mode_c = mode_->as(ncu_ * nb);
        f_.d_abs_sum_y = sum_->as<uint32_t>(), f_.d_abs_sum_u = sum_->as<uint32_t>(4 * ncu_ * nb), f_.d_abs_sum_v = sum_->as<uint32_t>(8 * ncu_ * nb);
    }

    size_t frame_bytes() const { return fsz_; }
    size_t cus() const { return ncu_; }
    int batch() const { return batch_; }
    int waves() const { return x265amd_intra_waves(&f_); }

    // n <= batch frames, Y,U,V each, one after the other (the reference's input format).  Asynchronous; fetch() waits.
    void encode(const uint8_t *yuv, int n = 1)
    {
        if (n < 1 || n > batch_) throw Error(X265AMD_ERR_ARG, "IntraEncoder::encode: more frames than the batch holds");
        check(x265amd_memcpy_h2d(src_->as(), yuv, fsz_ * n, stream_), "x265amd_memcpy_h2d");
        check(x265amd_intra_encode_frames(&f_, n, (int64_t)fsz_, (int64_t)fsz_, (int64_t)ncu_, stream_), "x265amd_intra_encode_frames");
    }

    // of the n frames last encoded: reconstruction [n][W*H*3/2], levels [n][W*H*3/2] (Y,U,V like the picture), luma /
    // chroma mode per CU [n][ncu] each, xQuant's sums [3][n][ncu] (what sets bCbf, src/encode.cpp:888-890).  Null
    // pointers are skipped.
    void fetch(uint8_t *rec, int16_t *level, uint8_t *mode_y, uint8_t *mode_c, uint32_t *abs_sum = nullptr, int n = 1)
    {
        fetch_async(rec, level, mode_y, mode_c, abs_sum, n);
        wait();
    }

    // The same copies queued behind the encode on the encoder's stream, without waiting: with page-locked destinations
    // (PinnedBuffer) the call returns at once and the host is free until wait() -- e.g. to write the pictures of the
    // batch before on its threads while this one is on the GPU (host/encode_i.cpp).
    void fetch_async(uint8_t *rec, int16_t *level, uint8_t *mode_y, uint8_t *mode_c, uint32_t *abs_sum = nullptr, int n = 1)
    {
        if (n < 1 || n > batch_) throw Error(X265AMD_ERR_ARG, "IntraEncoder::fetch: more frames than the batch holds");
        const size_t nb = (size_t)batch_;
        if (rec) check(x265amd_memcpy_d2h(rec, rec_->as(), fsz_ * n, stream_), "x265amd_memcpy_d2h");
        if (level) check(x265amd_memcpy_d2h(level, level_->as(), 2 * fsz_ * n, stream_), "x265amd_memcpy_d2h");
        if (mode_y) check(x265amd_memcpy_d2h(mode_y, mode_->as(), ncu_ * n, stream_), "x265amd_memcpy_d2h");
        if (mode_c) check(x265amd_memcpy_d2h(mode_c, mode_->as(ncu_ * nb), ncu_ * n, stream_), "x265amd_memcpy_d2h");
        if (abs_sum)
            for (int pl = 0; pl < 3; pl++)
                check(x265amd_memcpy_d2h(abs_sum + (size_t)pl * n * ncu_, sum_->as(4 * ncu_ * nb * pl), 4 * ncu_ * n, stream_), "x265amd_memcpy_d2h");
    }
    void wait() { check(x265amd_stream_sync(stream_), "x265amd_stream_sync"); }

    // the number the next picture written will carry (pictures of several encoders that take turns on one sequence:
    // the caller keeps the count)
    int poc() const { return poc_; }
    void set_poc(int poc) { poc_ = poc; }

    // The picture's bitstream from what fetch() brought back (Stage 4 of the reference's loop plus its headers and
    // slice end: x265amd_intra_write_picture, host code).  Pictures are numbered in the order they are written:
    // the first one is the IDR picture (src/encode.cpp:624,660).  Appends to `out`, returns the bytes added.
    size_t write_picture(const int16_t *level, const uint8_t *mode_y, const uint8_t *mode_c, std::vector<uint8_t> &out)
    {
        return write_at(poc_++, level, mode_y, mode_c, out);
    }

    // n pictures (fetch()'s layout) on up to `threads` host threads -- every picture is a slice of its own with a fresh
    // arithmetic-coder state, so they are written independently; out[k] receives picture k.
    void write_pictures(const int16_t *level, const uint8_t *mode_y, const uint8_t *mode_c, int n, int threads,
                        std::vector<std::vector<uint8_t>> &out)
    {
        out.assign((size_t)n, {});
        const int first = poc_;
        poc_ += n;
        const int nt = threads < 1 ? 1 : threads > n ? n : threads;
        std::vector<std::thread> pool;
        std::vector<std::exception_ptr> failed((size_t)nt);
        for (int t = 0; t < nt; t++)
            pool.emplace_back([&, t] {
                try {
                    for (int k = t; k < n; k += nt)
                        write_at(first + k, level + (size_t)k * fsz_, mode_y + (size_t)k * ncu_, mode_c + (size_t)k * ncu_, out[(size_t)k]);
                } catch (...) {  // a library status, or the host running out of memory: nothing may leave a thread
                    failed[(size_t)t] = std::current_exception();
                }
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < nt; t++)
            if (failed[(size_t)t]) std::rethrow_exception(failed[(size_t)t]);  // the first worker's own exception, message and all
    }

  private:
    size_t write_at(int poc, const int16_t *level, const uint8_t *mode_y, const uint8_t *mode_c, std::vector<uint8_t> &out) const
    {
        x265amd_intra_syntax s{};
        s.width = w_, s.height = h_, s.cu = cu_, s.qp = qp_, s.poc = poc;
        s.stride_y = w_, s.stride_c = w_ / 2;
        s.level_y = level, s.level_u = level + ysz_, s.level_v = level + ysz_ + csz_;
        s.mode_y = mode_y, s.mode_c = mode_c;
        // The bound is the worst case (7 bytes per sample: 22 MB at 1080p) and a picture is a few hundred KB: the
        // writer gets an UNINITIALISED block of that size (only the pages it reaches are ever touched; a vector
        // resized to the bound would zero all of it for every picture) and the bytes it made are appended.
        const size_t cap = x265amd_intra_stream_bound(w_, h_);
        std::unique_ptr<uint8_t[]> tmp(new uint8_t[cap]);
        size_t n = 0;
        check(x265amd_intra_write_picture(&s, tmp.get(), cap, &n), "x265amd_intra_write_picture");
        out.insert(out.end(), tmp.get(), tmp.get() + n);
        return n;
    }

    int w_, h_, cu_, qp_, batch_, poc_ = 0;
    void *stream_ = nullptr;
    size_t ysz_ = 0, csz_ = 0, fsz_ = 0, ncu_ = 0;
    std::unique_ptr<DeviceBuffer> src_, rec_, pred_, level_, mode_, sum_;
    x265amd_intra_frame f_;
};

}  // namespace x265amd_host
