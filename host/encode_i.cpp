// encode_i -- the reference's intra encoder from a C++ host, through the C-ABI only: pixel path on the GPU
// (x265amd_intra_encode_frames), syntax on host threads (x265amd_intra_write_picture).
//
//   x265amd_encode_i <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix> [batch [threads [dumps]]]
//
// in.yuv: 8-bit 4:2:0 frames, Y,U,V each (src/encode.cpp:1002-1003); width and height multiples of the CU (16 or
// 32).  Every frame is an intra picture on its own, as in the reference (SLICE_I only), so `batch` of them (default
// 1) go through the GPU together and their bitstreams are written on `threads` host threads (default 1).  Writes
// <out_prefix>.rec (reconstruction, frame after frame: the bytes the reference's MD5 SEI covers), .lvl (Int16
// levels, Y,U,V), .modes (per frame: luma mode per CU, then chroma mode per CU), .bit (the bitstream: what the
// reference's xEncEncode returns for each frame, one after the other) and prints one line per picture; `dumps` = 0
// keeps only .bit (and leaves the reconstruction on the GPU).
//
// Three slots go round, each with its own stream, device planes and page-locked host buffers: the main thread reads
// a batch and queues its GPU work (upload, pixel path, download -- nothing waits), the writer thread takes the
// batches in order and writes their pictures on the host threads.  Per frame the encoder costs max(reading, GPU
// side, syntax), not their sum.
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

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

struct Stream {
    void *s = nullptr;
    Stream() { check(x265amd_stream_create(&s), "x265amd_stream_create"); }
    ~Stream() { x265amd_stream_destroy(s); }
    Stream(const Stream &) = delete;
    Stream &operator=(const Stream &) = delete;
};

struct Slot {
    Stream stream;
    IntraEncoder enc;
    PinnedBuffer yuv, rec, lvl, mode_y, mode_c, sums;
    int first = 0, n = 0;  // the frames in flight
    Slot(int w, int h, int cu, int qp, int batch)
        : enc(w, h, cu, qp, 0, stream.s, batch), yuv(enc.frame_bytes() * batch), rec(enc.frame_bytes() * batch),
          lvl(2 * enc.frame_bytes() * batch), mode_y(enc.cus() * batch), mode_c(enc.cus() * batch), sums(4 * 3 * enc.cus() * batch)
    {
    }
};

static double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv)
{
    if (argc < 8 || argc > 11) {
        fprintf(stderr, "usage: %s <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix> [batch [threads [dumps]]]\n", argv[0]);
        return 2;
    }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), cu = atoi(argv[5]), qp = atoi(argv[6]);
        const std::string prefix = argv[7];
        const int batch = argc > 8 ? atoi(argv[8]) : 1, threads = argc > 9 ? atoi(argv[9]) : 1;
        const bool dumps = argc > 10 ? atoi(argv[10]) != 0 : true;
        if (batch < 1 || batch > 4096 || threads < 1 || threads > 256) throw Error(X265AMD_ERR_ARG, "batch / threads");
        check(x265amd_init(0), "x265amd_init");
        constexpr int kSlots = 3;
        std::unique_ptr<Slot> slots[kSlots];
        for (auto &s : slots) s.reset(new Slot(w, h, cu, qp, batch));
        const size_t fsz = slots[0]->enc.frame_bytes(), ncu = slots[0]->enc.cus();
        std::vector<std::vector<uint8_t>> bits;
        File in(argv[1], "rb"), fbit(prefix + ".bit", "wb");
        std::unique_ptr<File> frec, flvl, fmod;
        if (dumps) {
            frec.reset(new File(prefix + ".rec", "wb"));
            flvl.reset(new File(prefix + ".lvl", "wb"));
            fmod.reset(new File(prefix + ".modes", "wb"));
        }
        double read_ms = 0, wait_ms = 0, syntax_ms = 0, out_ms = 0;
        int poc = 0;

        // the batch a slot holds: wait for its copies, write its pictures, hand everything out (the writer thread)
        auto finish = [&](Slot &s) {
            auto c = std::chrono::steady_clock::now();
            s.enc.wait();
            wait_ms += ms_since(c);
            c = std::chrono::steady_clock::now();
            s.enc.set_poc(poc);  // pictures are numbered across the slots, in input order
            s.enc.write_pictures(s.lvl.as<int16_t>(), s.mode_y.as(), s.mode_c.as(), s.n, threads, bits);
            poc += s.n;
            syntax_ms += ms_since(c);
            c = std::chrono::steady_clock::now();
            const uint32_t *sums = s.sums.as<uint32_t>();
            for (int k = 0; k < s.n; k++) {
                size_t coded = 0;
                for (int pl = 0; pl < 3; pl++)
                    for (size_t i = 0; i < ncu; i++) coded += sums[((size_t)pl * s.n + k) * ncu + i] != 0;
                if (dumps) {
                    printf("frame %d: %d anti-diagonals, %zu CUs, %zu of %zu blocks coded, %zu bytes, rec fnv %016llx\n", s.first + k,
                           s.enc.waves(), ncu, coded, 3 * ncu, bits[k].size(), (unsigned long long)fnv1a(s.rec.as() + k * fsz, fsz));
                    fwrite(s.rec.as() + k * fsz, 1, fsz, frec->f);
                    fwrite(s.lvl.as<int16_t>() + k * fsz, 2, fsz, flvl->f);
                    fwrite(s.mode_y.as() + k * ncu, 1, ncu, fmod->f);
                    fwrite(s.mode_c.as() + k * ncu, 1, ncu, fmod->f);
                } else {
                    printf("frame %d: %d anti-diagonals, %zu CUs, %zu of %zu blocks coded, %zu bytes\n", s.first + k, s.enc.waves(), ncu,
                           coded, 3 * ncu, bits[k].size());
                }
                fwrite(bits[k].data(), 1, bits[k].size(), fbit.f);
            }
            out_ms += ms_since(c);
        };

        // Slots go round: the main thread fills the next free one (read, queue the GPU work) and hands it to the writer
        // thread, which takes them in order.  `filled` / `done` count batches; slot = batch % kSlots.
        std::mutex mu;
        std::condition_variable cv;
        int filled = 0, done = 0;
        bool stop = false;
        std::exception_ptr failure;
        std::thread writer([&] {
            try {
                for (;;) {
                    int k;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return done < filled || stop; });
                        if (done == filled) return;
                        k = done;
                    }
                    finish(*slots[k % kSlots]);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        done = k + 1;
                    }
                    cv.notify_all();
                }
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                failure = std::current_exception();
                done = 1 << 30;  // nobody waits for a slot any more
                cv.notify_all();
            }
        });
        struct Joiner {  // the writer is joined on every way out of this block
            std::thread &t;
            std::mutex &mu;
            std::condition_variable &cv;
            bool &stop;
            ~Joiner()
            {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    stop = true;
                }
                cv.notify_all();
                if (t.joinable()) t.join();
            }
        } joiner{writer, mu, cv, stop};

        const auto wall0 = std::chrono::steady_clock::now();
        int nbatch = 0;
        for (int t0 = 0; t0 < nf; t0 += batch, nbatch++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done > nbatch - kSlots; });  // the batch that last used this slot has been written
                if (failure) break;
            }
            Slot &s = *slots[nbatch % kSlots];
            s.first = t0, s.n = nf - t0 < batch ? nf - t0 : batch;
            auto c = std::chrono::steady_clock::now();
            if (fread(s.yuv.as(), 1, fsz * s.n, in.f) != fsz * s.n) throw std::runtime_error("short read at frame " + std::to_string(t0));
            read_ms += ms_since(c);
            s.enc.encode(s.yuv.as(), s.n);
            s.enc.fetch_async(dumps ? s.rec.as() : nullptr, s.lvl.as<int16_t>(), s.mode_y.as(), s.mode_c.as(), s.sums.as<uint32_t>(), s.n);
            {
                std::lock_guard<std::mutex> lk(mu);
                filled = nbatch + 1;
            }
            cv.notify_all();
        }
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done >= filled; });
        }
        if (failure) std::rethrow_exception(failure);
        const double wall = ms_since(wall0);
        printf("%d frames, batch %d, %d writer thread(s): %.2f ms/frame wall (main thread: reading the input %.2f; writer thread: waiting "
               "for the GPU %.2f, syntax %.2f, writing the output%s %.2f)\n",
               nf, batch, threads, wall / nf, read_ms / nf, wait_ms / nf, syntax_ms / nf, dumps ? " and the dumps" : "", out_ms / nf);
    } catch (const Error &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return e.status == X265AMD_ERR_ARG ? 3 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return 1;
    }
    return 0;
}
