// encode_p -- closed-loop P-picture path of one GOP segment from a C++ host, through the C-ABI only.
//
//   x265amd_encode_p <in.yuv> <width> <height> <nframes> <qp> <range> <out_prefix> [intra_cu]
//
// in.yuv: 8-bit 4:2:0 frames, Y,U,V each (the reference's input format, src/encode.cpp:1002-1003); width and
// height multiples of 64.  Frame 0 opens the segment: with intra_cu (16 or 32) it is coded as the reference codes
// every picture -- GopEncoder::encode_i, the library's intra path -- and written to <out_prefix>.i.rec / .i.lvl /
// .i.modes; without it its reconstruction is taken to be the source.  Every following frame is searched against and predicted from the
// reconstruction before it.  Writes <out_prefix>.mv (x265amd_mv per CTU), .lvl (Int16 levels, Y,U,V),
// .rec (reconstruction) -- frame after frame -- and prints one line per picture.
//
//   x265amd_encode_p --bench <in.yuv> <width> <height> <nframes> <qp> <range> <nseg>
//
// times the same loop for `nseg` segments (each fed the file's frames) with everything a real host pays:
// source frames up over PCIe from pinned memory, search + transform/quant, vectors + levels + coded flags
// back down, one stream sync per picture.  Prints Mpixels/s (luma pixels, like bench.py).  Nothing is written.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "x265amd_host.hpp"

using namespace x265amd_host;

static uint64_t fnv1a(const void *p, size_t n)
{
    const uint8_t *b = static_cast<const uint8_t *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

struct File {
    FILE *f;
    File(const std::string &path, const char *mode) : f(fopen(path.c_str(), mode))
    {
        if (!f) throw std::runtime_error("cannot open " + path);
    }
    ~File() { fclose(f); }
};

struct Batch {  // one batch of segments: its encoder, its stream, its pinned buffers
    void *stream = nullptr;
    std::unique_ptr<GopEncoder> enc;
    std::unique_ptr<PinnedBuffer> up, lvl, mv, cbf, table, words;
    bool packed, direct = false;
    size_t words_down = 0;
    size_t step_bytes = 0;
    Batch(int w, int h, int nseg, int range, int qp, bool own_stream, bool packed_, int nframes) : packed(packed_)
    {
        check(x265amd_init(0), "x265amd_init");
        if (own_stream) check(x265amd_stream_create(&stream), "x265amd_stream_create");
        enc.reset(new GopEncoder(w, h, nseg, range, qp, 0, stream));
        step_bytes = enc->frame_bytes() * nseg;
        up.reset(new PinnedBuffer(step_bytes * nframes));  // every picture of the run waits here
        if (packed) {
            table.reset(new PinnedBuffer(4 * enc->packed_table_words()));
            words.reset(new PinnedBuffer(4 * enc->packed_words_worst()));
        } else {
            lvl.reset(new PinnedBuffer(2 * enc->frame_bytes() * nseg));
        }
        mv.reset(new PinnedBuffer(sizeof(x265amd_mv) * enc->ctus() * nseg));
        cbf.reset(new PinnedBuffer(12 * enc->tus() * nseg));
    }
    ~Batch()
    {
        enc.reset();
        if (stream) x265amd_stream_destroy(stream);
    }
    void go(int t)
    {
        enc->encode_p(up->as() + (size_t)t * step_bytes);
        if (packed)  // the table says which TUs are coded: the sums stay on the device
            enc->fetch_packed_async(mv->as<x265amd_mv>(), table->as<uint32_t>(), words->as<uint32_t>(), enc->packed_words_worst(), 0, direct);
        else
            enc->fetch_async(mv->as<x265amd_mv>(), lvl->as<int16_t>(), nullptr, cbf->as<uint32_t>());
    }
    void finish()
    {
        enc->sync();
        if (packed) words_down = enc->fetch_packed_finish();
    }
};

// nbatch = 1: upload, kernels, download strictly one after the other.  nbatch = 2, 4: the segments split into
// batches on streams of their own, so that one batch's PCIe copies run under the others' kernels.
// threads: every batch on a host thread of its own, so that no batch's next picture waits for another batch's sync --
// with the kernels writing the results into host memory themselves (mode 2) this is what lets uploads run under
// kernels picture after picture.
// X265AMD_BENCH_ONLY=mode,streams [X265AMD_BENCH_THREADS=1]: one configuration only (tools/host_prof.sh's traces).
static void bench_one(const std::vector<uint8_t> &file, int w, int h, int nframes, int qp, int range, int nseg, int nbatch, int mode, bool threads = false)
{
    const bool packed = mode >= 1;  // mode 2: the kernels write the packed levels into the host's memory themselves

    std::vector<std::unique_ptr<Batch>> bs;
    const int nsegb = nseg / nbatch;
    for (int b = 0; b < nbatch; b++) {
        bs.emplace_back(new Batch(w, h, nsegb, range, qp, nbatch > 1, packed, nframes));
        bs.back()->direct = mode == 2;
    }
    const size_t fsz = bs[0]->enc->frame_bytes();
    for (auto &b : bs)  // a real host would read / receive the frames: they wait in pinned memory, not timed
        for (int t = 0; t < nframes; t++)
            for (int s = 0; s < nsegb; s++) memcpy(b->up->as() + ((size_t)t * nsegb + s) * fsz, file.data() + (size_t)t * fsz, fsz);
    double best = 1e30, total = 0, down_total = 0;
    int pics = 0;
    bool counted = false;
    auto finish = [&](Batch &b) {
        b.finish();
        if (counted)
            down_total += packed ? 4.0 * b.words_down + b.table->bytes() + b.mv->bytes() : (double)(b.lvl->bytes() + b.mv->bytes() + b.cbf->bytes());
    };
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    for (int rep = 0; rep < 3; rep++) {  // first repetition = warm-up
        counted = rep > 0;
        for (auto &b : bs) b->enc->set_reference(b->up->as());
        for (auto &b : bs) b->enc->sync();
        if (threads) {  // every batch on a host thread of its own: picture after picture, nobody waits for anybody
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            std::vector<std::exception_ptr> err(bs.size());
            for (size_t i = 0; i < bs.size(); i++)
                th.emplace_back([&, i] {
                    try {
                        check(x265amd_init(0), "x265amd_init");  // the device is per thread
                        for (int t = 1; t < nframes; t++) {
                            bs[i]->go(t);
                            bs[i]->finish();
                        }
                    } catch (...) {
                        err[i] = std::current_exception();
                    }
                });
            for (auto &t : th) t.join();
            for (auto &e : err)
                if (e) std::rethrow_exception(e);
            const double ms = since(t0) / (nframes - 1);
            if (counted)
                for (auto &b : bs)
                    down_total += (nframes - 1) * (packed ? 4.0 * b->words_down + b->table->bytes() + b->mv->bytes()
                                                          : (double)(b->lvl->bytes() + b->mv->bytes() + b->cbf->bytes()));
            if (rep) {
                best = ms < best ? ms : best;
                total += ms * (nframes - 1);
                pics += nframes - 1;
            }
            continue;
        }
        for (int t = 1; t < nframes; t++) {
            const auto t0 = std::chrono::steady_clock::now();
            for (auto &b : bs) b->go(t);
            for (auto &b : bs) finish(*b);
            const double ms = since(t0);
            if (rep) {
                best = ms < best ? ms : best;
                total += ms;
                pics++;
            }
        }
    }
    const double mpix = (double)w * h * nseg / 1e6;
    printf("{\"host\": \"%s\", \"pcie_inclusive\": true, \"levels\": \"%s\", \"width\": %d, \"height\": %d, \"nseg\": %d, "
           "\"streams\": %d, \"pictures\": %d, \"ms_per_step_mean\": %.4f, \"ms_per_step_best\": %.4f, \"mpixels_per_s_mean\": %.1f, "
           "\"bytes_up_per_step\": %zu, \"bytes_down_per_step\": %.0f}\n",
           threads ? "c++, a thread per stream" : "c++", mode == 2 ? "packed_direct" : packed ? "packed" : "dense", w, h, nseg, nbatch, pics, total / pics, best,
           mpix / (total / pics) * 1e3, fsz * nseg, down_total / pics);
}

static int bench(char **a)
{
    const int w = atoi(a[1]), h = atoi(a[2]), nframes = atoi(a[3]), qp = atoi(a[4]), range = atoi(a[5]), nseg = atoi(a[6]);
    std::vector<uint8_t> file;
    {
        GopEncoder probe(w, h, 1, range, qp);  // geometry check + frame size
        file.resize(probe.frame_bytes() * nframes);
        File in(a[0], "rb");
        if (fread(file.data(), 1, file.size(), in.f) != file.size()) throw std::runtime_error("short read");
    }
    if (const char *only = getenv("X265AMD_BENCH_ONLY")) {  // "mode,streams": one configuration (for traces)
        bench_one(file, w, h, nframes, qp, range, nseg, atoi(strchr(only, ',') ? strchr(only, ',') + 1 : "1"), atoi(only),
                  getenv("X265AMD_BENCH_THREADS") != nullptr);
        return 0;
    }
    for (int mode = 0; mode < 3; mode++) {  // the levels as dense planes, then as x265amd_pack_levels leaves them (copied / written direct)
        bench_one(file, w, h, nframes, qp, range, nseg, 1, mode);
        for (int nb = 2; nb <= 8; nb *= 2)
            if (nseg % nb == 0 && nseg >= 2 * nb - 2 && (nb < 8 || mode)) {
                bench_one(file, w, h, nframes, qp, range, nseg, nb, mode);
                if (mode) bench_one(file, w, h, nframes, qp, range, nseg, nb, mode, true);
            }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 9 && !strcmp(argv[1], "--bench")) {
        try {
            return bench(argv + 2);
        } catch (const std::exception &e) {
            fprintf(stderr, "x265amd_encode_p: %s\n", e.what());
            return 1;
        }
    }
    if (argc != 8 && argc != 9) {
        fprintf(stderr, "usage: %s <in.yuv> <width> <height> <nframes> <qp> <range> <out_prefix> [intra_cu]\n", argv[0]);
        return 2;
    }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]), nframes = atoi(argv[4]), qp = atoi(argv[5]), range = atoi(argv[6]);
        const std::string prefix = argv[7];
        const int icu = argc == 9 ? atoi(argv[8]) : 0;  // 16 / 32: the first picture goes through the intra path
        GopEncoder enc(w, h, 1, range, qp);
        const size_t fsz = enc.frame_bytes();
        std::vector<uint8_t> frame(fsz), rec(fsz);
        std::vector<int16_t> level(fsz);
        std::vector<x265amd_mv> mv(enc.ctus());
        std::vector<uint32_t> cbf(3 * enc.tus()), ptable(enc.packed_table_words()), pwords(enc.packed_words_worst());
        std::vector<int16_t> plane((size_t)w * h);
        size_t packed_words = 0;
        PinnedBuffer dtable(4 * ptable.size()), dwords(4 * pwords.size()), dmv(sizeof(x265amd_mv) * mv.size());
        File in(argv[1], "rb"), fmv(prefix + ".mv", "wb"), flvl(prefix + ".lvl", "wb"), frec(prefix + ".rec", "wb");
        for (int t = 0; t < nframes; t++) {
            if (fread(frame.data(), 1, fsz, in.f) != fsz) throw std::runtime_error("short read at frame " + std::to_string(t));
            if (t == 0 && icu) {  // the I picture: <prefix>.i.rec / .i.lvl / .i.modes (luma modes, then chroma modes)
                enc.encode_i(frame.data(), icu);
                enc.fetch(nullptr, level.data(), rec.data());
                std::vector<uint8_t> modes(2 * (size_t)(w / icu) * (h / icu));
                enc.fetch_modes(modes.data(), modes.data() + modes.size() / 2);
                File irec(prefix + ".i.rec", "wb"), ilvl(prefix + ".i.lvl", "wb"), imod(prefix + ".i.modes", "wb");
                fwrite(rec.data(), 1, rec.size(), irec.f);
                fwrite(level.data(), 2, level.size(), ilvl.f);
                fwrite(modes.data(), 1, modes.size(), imod.f);
                printf("POC %d: intra, CU %d, recon fnv1a %016llx\n", enc.poc(), icu, (unsigned long long)fnv1a(rec.data(), fsz));
                continue;
            }
            if (t == 0) {
                enc.set_reference(frame.data());
                continue;
            }
            enc.encode_p(frame.data());
            enc.fetch(mv.data(), level.data(), rec.data(), cbf.data());
            {  // the packed form of the same levels must say the same thing
                const size_t words = enc.fetch_packed(nullptr, ptable.data(), pwords.data(), pwords.size());
                if (words > pwords.size()) throw std::runtime_error("packed levels: more words than levels");
                const size_t ysz = (size_t)w * h, offs[3] = {0, ysz, ysz + ysz / 4};
                for (int pl = 0; pl < 3; pl++) {
                    enc.unpack_plane(ptable.data(), pwords.data(), pl, 0, plane.data());
                    if (memcmp(plane.data(), level.data() + offs[pl], 2 * (pl ? ysz / 4 : ysz)))
                        throw std::runtime_error("packed levels differ from the dense plane " + std::to_string(pl));
                }
                packed_words = words;
                // and so must what the kernels write into page-locked memory themselves
                enc.fetch_packed_async(dmv.as<x265amd_mv>(), dtable.as<uint32_t>(), dwords.as<uint32_t>(), pwords.size(), 0, true);
                enc.sync();
                if (enc.fetch_packed_finish() != words) throw std::runtime_error("packed levels, direct: another number of words");
                for (int pl = 0; pl < 3; pl++) {
                    enc.unpack_plane(dtable.as<uint32_t>(), dwords.as<uint32_t>(), pl, 0, plane.data());
                    if (memcmp(plane.data(), level.data() + offs[pl], 2 * (pl ? ysz / 4 : ysz)))
                        throw std::runtime_error("packed levels, direct, differ from the dense plane " + std::to_string(pl));
                }
                if (memcmp(dmv.as(), mv.data(), sizeof(x265amd_mv) * mv.size())) throw std::runtime_error("vectors, direct, differ");
            }
            fwrite(mv.data(), sizeof(x265amd_mv), mv.size(), fmv.f);
            fwrite(level.data(), 2, level.size(), flvl.f);
            fwrite(rec.data(), 1, rec.size(), frec.f);
            uint64_t sad = 0, coded = 0;
            for (const x265amd_mv &m : mv) sad += m.sad;
            for (uint32_t c : cbf) coded += c != 0;
            printf("POC %d: %zu CTUs, sum of best SADs %llu, coded TUs %llu of %zu, packed levels %zu words, recon fnv1a %016llx%s\n", enc.poc(), mv.size(),
                   (unsigned long long)sad, (unsigned long long)coded, cbf.size(), packed_words, (unsigned long long)fnv1a(rec.data(), fsz),
                   enc.exhaustive_fallback() ? " (every-candidate search)" : "");
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "x265amd_encode_p: %s\n", e.what());
        return 1;
    }
    return 0;
}
