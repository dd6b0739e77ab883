// quick mutation fuzzer for the probes (ASan/UBSan build)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "rcgpu.h"
static std::vector<uint8_t> slurp(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != size_t(n)) v.clear(); fclose(f); return v; }
static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return uint32_t(rs >> 16); }
int main(int argc, char** argv)
{
    long iters = atol(argv[1]);
    size_t nprobe = 0, ok = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> seed = slurp(argv[a]);
        if (seed.empty()) { fprintf(stderr, "cannot read %s\n", argv[a]); return 2; }
        const size_t head = std::min<size_t>(seed.size(), 4096);
        for (long it = 0; it < iters; it++) {
            // exact-size heap copy so that ASan sees any read past the end
            size_t n = seed.size();
            const uint32_t mode = rnd() % 8;
            if (mode == 0) n = rnd() % (head + 1);                    // truncate inside the header
            else if (mode == 1) n = seed.size() - (rnd() % std::min<size_t>(64, seed.size() + 1));       // cut the tail
            uint8_t* buf = static_cast<uint8_t*>(malloc(n ? n : 1));
            memcpy(buf, seed.data(), n);
            const int flips = 1 + rnd() % 6;
            for (int k = 0; k < flips && n; k++) {
                const size_t at = rnd() % std::min(n, head);
                switch (rnd() % 4) { case 0: buf[at] ^= 1u << (rnd() % 8); break; case 1: buf[at] = uint8_t(rnd()); break; case 2: buf[at] = 0xFF; break; default: buf[at] = 0; }
            }
            if (mode == 2 && n >= 8) { const uint32_t v = rnd() % 3 ? 0xFFFFFFF0u + rnd() % 16 : rnd(); memcpy(buf + (rnd() % std::min(n - 4, head)), &v, 4); }   // huge 32-bit fields
            rcgpu_image_info ii; rcgpu_audio_info ai;
            ok += rcgpu_dpx_probe(buf, n, &ii) == 0; ok += rcgpu_tiff_probe(buf, n, &ii) == 0; ok += rcgpu_exr_probe(buf, n, &ii) == 0; ok += rcgpu_wav_probe(buf, n, &ai) == 0;
            nprobe += 4;
            free(buf);
        }
    }
    printf("fuzz: %zu probes, %zu accepted\n", nprobe, ok);
    return 0;
}
