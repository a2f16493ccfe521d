// encode_p -- closed-loop P-picture path of one GOP segment from a C++ host, through the C-ABI only.
//
//   x265amd_encode_p <in.yuv> <width> <height> <nframes> <qp> <range> <out_prefix>
//
// in.yuv: 8-bit 4:2:0 frames, Y,U,V each (the reference's input format, src/encode.cpp:1002-1003); width and
// height multiples of 64.  Frame 0 opens the segment; its reconstruction is taken to be the source (intra
// coding is outside the path).  Every following frame is searched against and predicted from the
// reconstruction before it.  Writes <out_prefix>.mv (x265amd_mv per CTU), .lvl (Int16 levels, Y,U,V),
// .rec (reconstruction) -- frame after frame -- and prints one line per picture.
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

int main(int argc, char **argv)
{
    if (argc != 8) {
        fprintf(stderr, "usage: %s <in.yuv> <width> <height> <nframes> <qp> <range> <out_prefix>\n", argv[0]);
        return 2;
    }
    try {
        const int w = atoi(argv[2]), h = atoi(argv[3]), nframes = atoi(argv[4]), qp = atoi(argv[5]), range = atoi(argv[6]);
        const std::string prefix = argv[7];
        GopEncoder enc(w, h, 1, range, qp);
        const size_t fsz = enc.frame_bytes();
        std::vector<uint8_t> frame(fsz), rec(fsz);
        std::vector<int16_t> level(fsz);
        std::vector<x265amd_mv> mv(enc.ctus());
        std::vector<uint32_t> cbf(3 * enc.tus());
        File in(argv[1], "rb"), fmv(prefix + ".mv", "wb"), flvl(prefix + ".lvl", "wb"), frec(prefix + ".rec", "wb");
        for (int t = 0; t < nframes; t++) {
            if (fread(frame.data(), 1, fsz, in.f) != fsz) throw std::runtime_error("short read at frame " + std::to_string(t));
            if (t == 0) {
                enc.set_reference(frame.data());
                continue;
            }
            enc.encode_p(frame.data());
            enc.fetch(mv.data(), level.data(), rec.data(), cbf.data());
            fwrite(mv.data(), sizeof(x265amd_mv), mv.size(), fmv.f);
            fwrite(level.data(), 2, level.size(), flvl.f);
            fwrite(rec.data(), 1, rec.size(), frec.f);
            uint64_t sad = 0, coded = 0;
            for (const x265amd_mv &m : mv) sad += m.sad;
            for (uint32_t c : cbf) coded += c != 0;
            printf("POC %d: %zu CTUs, sum of best SADs %llu, coded TUs %llu of %zu, recon fnv1a %016llx%s\n", enc.poc(), mv.size(),
                   (unsigned long long)sad, (unsigned long long)coded, cbf.size(), (unsigned long long)fnv1a(rec.data(), fsz),
                   enc.exhaustive_fallback() ? " (every-candidate search)" : "");
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "x265amd_encode_p: %s\n", e.what());
        return 1;
    }
    return 0;
}
