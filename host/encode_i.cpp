// encode_i -- the reference's intra encoder from a C++ host, through the C-ABI only: pixel path on the GPU
// (x265amd_intra_encode_frames), syntax on host threads (x265amd_intra_write_picture).
//
//   x265amd_encode_i <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix> [batch [threads]]
//
// in.yuv: 8-bit 4:2:0 frames, Y,U,V each (src/encode.cpp:1002-1003); width and height multiples of the CU (16 or
// 32).  Every frame is an intra picture on its own, as in the reference (SLICE_I only), so `batch` of them (default
// 1) go through the GPU together and their bitstreams are written on `threads` host threads (default 1).  Writes
// <out_prefix>.rec (reconstruction, frame after frame: the bytes the reference's MD5 SEI covers), .lvl (Int16
// levels, Y,U,V), .modes (per frame: luma mode per CU, then chroma mode per CU), .bit (the bitstream: what the
// reference's xEncEncode returns for each frame, one after the other) and prints one line per picture.
#include <chrono>
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

int main(int argc, char **argv)
{
    if (argc < 8 || argc > 10) {
        fprintf(stderr, "usage: %s <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix> [batch [threads]]\n", argv[0]);
        return 2;
    }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), cu = atoi(argv[5]), qp = atoi(argv[6]);
        const std::string prefix = argv[7];
        const int batch = argc > 8 ? atoi(argv[8]) : 1, threads = argc > 9 ? atoi(argv[9]) : 1;
        if (batch < 1 || batch > 4096 || threads < 1 || threads > 256) throw Error(X265AMD_ERR_ARG, "batch / threads");
        IntraEncoder enc(w, h, cu, qp, 0, nullptr, batch);
        const size_t fsz = enc.frame_bytes(), ncu = enc.cus(), nb = (size_t)batch;
        std::vector<uint8_t> yuv(fsz * nb), rec(fsz * nb), mode_y(ncu * nb), mode_c(ncu * nb);
        std::vector<int16_t> lvl(fsz * nb);
        std::vector<uint32_t> sums(3 * ncu * nb);
        std::vector<std::vector<uint8_t>> bits;
        File in(argv[1], "rb"), frec(prefix + ".rec", "wb"), flvl(prefix + ".lvl", "wb"), fmod(prefix + ".modes", "wb"),
            fbit(prefix + ".bit", "wb");
        double gpu_ms = 0, host_ms = 0;
        for (int t0 = 0; t0 < nf; t0 += batch) {
            const int n = nf - t0 < batch ? nf - t0 : batch;
            if (fread(yuv.data(), 1, fsz * n, in.f) != fsz * n) throw std::runtime_error("short read at frame " + std::to_string(t0));
            const auto c0 = std::chrono::steady_clock::now();
            enc.encode(yuv.data(), n);
            enc.fetch(rec.data(), lvl.data(), mode_y.data(), mode_c.data(), sums.data(), n);
            const auto c1 = std::chrono::steady_clock::now();
            enc.write_pictures(lvl.data(), mode_y.data(), mode_c.data(), n, threads, bits);
            const auto c2 = std::chrono::steady_clock::now();
            gpu_ms += std::chrono::duration<double, std::milli>(c1 - c0).count();
            host_ms += std::chrono::duration<double, std::milli>(c2 - c1).count();
            for (int k = 0; k < n; k++) {
                size_t coded = 0;
                for (int pl = 0; pl < 3; pl++)
                    for (size_t i = 0; i < ncu; i++) coded += sums[((size_t)pl * n + k) * ncu + i] != 0;
                printf("frame %d: %d anti-diagonals, %zu CUs, %zu of %zu blocks coded, %zu bytes, rec fnv %016llx\n", t0 + k, enc.waves(),
                       ncu, coded, 3 * ncu, bits[k].size(), (unsigned long long)fnv1a(rec.data() + k * fsz, fsz));
                fwrite(rec.data() + k * fsz, 1, fsz, frec.f);
                fwrite(lvl.data() + k * fsz, 2, fsz, flvl.f);
                fwrite(mode_y.data() + k * ncu, 1, ncu, fmod.f);
                fwrite(mode_c.data() + k * ncu, 1, ncu, fmod.f);
                fwrite(bits[k].data(), 1, bits[k].size(), fbit.f);
            }
        }
        printf("%d frames, batch %d, %d writer thread(s): upload + pixel path + download %.2f ms/frame, syntax %.2f ms/frame\n", nf, batch,
               threads, gpu_ms / nf, host_ms / nf);
    } catch (const Error &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return e.status == X265AMD_ERR_ARG ? 3 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return 1;
    }
    return 0;
}
