// two_hosts.cpp -- two host threads drive the library at once, each with its own stream, its own encoder
// and a different geometry (so that the two ask for different LDS sizes, occupancies and workspaces).
// What the reference keeps in its handle X265_t (inc/x265.h:186-212) the library keeps per device; this
// program checks that nothing of it is shared between callers in a way that changes results: every
// picture's vectors, levels and reconstruction from the concurrent run must equal the same encoder's run
// alone.  Exit status 0 and a line "two_hosts ok" on success.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

#include "x265amd_host.hpp"

using namespace x265amd_host;

struct Job {
    int w, h, range, qp, pics;
    uint32_t seed;
};

static void make_yuv(std::vector<uint8_t> &yuv, const Job &j, int t)
{
    const size_t ysz = (size_t)j.w * j.h, fsz = ysz * 3 / 2;
    yuv.resize(fsz);
    uint32_t s = j.seed + 977u * (uint32_t)t;
    const int sx = 3 * t, sy = 2 * t;  // global motion inside every range used here
    for (int y = 0; y < j.h; y++)
        for (int x = 0; x < j.w; x++) {
            s = s * 1664525u + 1013904223u;
            yuv[(size_t)y * j.w + x] = (uint8_t)((((x + sx) * 5 + (y + sy) * 3) >> 2) + ((s >> 24) & 7));
        }
    for (size_t i = ysz; i < fsz; i++) {
        s = s * 1664525u + 1013904223u;
        yuv[i] = (uint8_t)(128 + ((s >> 24) & 7));
    }
}

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const uint8_t *b = static_cast<const uint8_t *>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// digest over every picture's vectors, levels and reconstruction
static uint64_t run(const Job &j, int device, bool own_stream)
{
    void *stream = nullptr;
    check(x265amd_init(device), "x265amd_init");  // per thread: makes the device current here
    if (own_stream) check(x265amd_stream_create(&stream), "x265amd_stream_create");
    uint64_t h = 1469598103934665603ull;
    {
        GopEncoder enc(j.w, j.h, 1, j.range, j.qp, device, stream);
        std::vector<uint8_t> yuv, rec(enc.frame_bytes());
        std::vector<int16_t> lvl(enc.frame_bytes());
        std::vector<x265amd_mv> mv(enc.ctus());
        make_yuv(yuv, j, 0);
        enc.set_reference(yuv.data());
        for (int t = 1; t <= j.pics; t++) {
            make_yuv(yuv, j, t);
            enc.encode_p(yuv.data());
            enc.fetch(mv.data(), lvl.data(), rec.data());
            h = fnv(h, mv.data(), mv.size() * sizeof(x265amd_mv));
            h = fnv(h, lvl.data(), lvl.size() * 2);
            h = fnv(h, rec.data(), rec.size());
        }
    }
    if (stream) check(x265amd_stream_destroy(stream), "x265amd_stream_destroy");
    return h;
}

int main(int argc, char **argv)
{
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    try {
        const Job a{256, 192, 24, 28, 4, 11u};    // small window: two 8-wave workgroups per CU
        const Job b{384, 256, 100, 35, 3, 23u};   // +-100 window: the 16-wave shape, > 64 KB of LDS
        const uint64_t want_a = run(a, device, false), want_b = run(b, device, false);
        std::atomic<int> bad{0};
        for (int r = 0; r < rounds; r++) {
            uint64_t got_a = 0, got_b = 0;
            std::thread ta([&] {
                try {
                    got_a = run(a, device, true);
                } catch (const std::exception &e) {
                    fprintf(stderr, "thread a: %s\n", e.what());
                    bad++;
                }
            });
            std::thread tb([&] {
                try {
                    got_b = run(b, device, true);
                } catch (const std::exception &e) {
                    fprintf(stderr, "thread b: %s\n", e.what());
                    bad++;
                }
            });
            ta.join();
            tb.join();
            if (got_a != want_a || got_b != want_b) {
                fprintf(stderr, "round %d: digests differ from the single-threaded run (a %d, b %d)\n", r, got_a == want_a,
                        got_b == want_b);
                bad++;
            }
        }
        // a thread that never called x265amd_init must be told so, not served with another thread's state --
        // unless the device it lands on (HIP's default: device 0) is one that has been initialised
        if (bad) return 1;
        printf("two_hosts ok: %d concurrent rounds, digests %016llx %016llx\n", rounds, (unsigned long long)want_a,
               (unsigned long long)want_b);
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "two_hosts: %s\n", e.what());
        return 1;
    }
}
