// encode_i -- the pixel path of the reference's intra encoder from a C++ host, through the C-ABI only.
//
//   x265amd_encode_i <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix>
//
// in.yuv: 8-bit 4:2:0 frames, Y,U,V each (src/encode.cpp:1002-1003); width and height multiples of the CU (16 or
// 32).  Every frame is an intra picture on its own, as in the reference (SLICE_I only).  Writes <out_prefix>.rec
// (reconstruction, frame after frame: the bytes the reference's MD5 SEI covers), .lvl (Int16 levels, Y,U,V),
// .modes (per frame: luma mode per CU, then chroma mode per CU), .bit (the bitstream: what the reference's
// xEncEncode returns for each frame, one after the other) and prints one line per picture.
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
    if (argc != 8) {
        fprintf(stderr, "usage: %s <in.yuv> <width> <height> <nframes> <cu> <qp> <out_prefix>\n", argv[0]);
        return 2;
    }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), cu = atoi(argv[5]), qp = atoi(argv[6]);
        const std::string prefix = argv[7];
        IntraEncoder enc(w, h, cu, qp);
        const size_t fsz = enc.frame_bytes(), ncu = enc.cus();
        std::vector<uint8_t> yuv(fsz), rec(fsz), modes(2 * ncu);
        std::vector<int16_t> lvl(fsz);
        std::vector<uint32_t> sums(3 * ncu);
        std::vector<uint8_t> bits;
        File in(argv[1], "rb"), frec(prefix + ".rec", "wb"), flvl(prefix + ".lvl", "wb"), fmod(prefix + ".modes", "wb"),
            fbit(prefix + ".bit", "wb");
        for (int t = 0; t < nf; t++) {
            if (fread(yuv.data(), 1, fsz, in.f) != fsz) throw std::runtime_error("short read on frame " + std::to_string(t));
            const auto t0 = std::chrono::steady_clock::now();
            enc.encode(yuv.data());
            enc.fetch(rec.data(), lvl.data(), modes.data(), modes.data() + ncu, sums.data());
            bits.clear();
            const size_t nbits = enc.write_picture(lvl.data(), modes.data(), modes.data() + ncu, bits);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            size_t coded = 0;
            for (size_t i = 0; i < 3 * ncu; i++) coded += sums[i] != 0;
            printf("frame %d: %d anti-diagonals, %zu CUs, %zu of %zu blocks coded, %zu bytes, rec fnv %016llx, %.2f ms\n", t,
                   enc.waves(), ncu, coded, 3 * ncu, nbits, (unsigned long long)fnv1a(rec.data(), fsz), ms);
            fwrite(bits.data(), 1, bits.size(), fbit.f);
            fwrite(rec.data(), 1, fsz, frec.f);
            fwrite(lvl.data(), 2, fsz, flvl.f);
            fwrite(modes.data(), 1, 2 * ncu, fmod.f);
        }
    } catch (const Error &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return e.status == X265AMD_ERR_ARG ? 3 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "x265amd_encode_i: %s\n", e.what());
        return 1;
    }
    return 0;
}
