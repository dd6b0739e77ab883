#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rcgpu.h"
static std::vector<uint8_t> slurp(const char* p) { std::vector<uint8_t> v; FILE* f = fopen(p, "rb"); if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != size_t(n)) v.clear(); fclose(f); return v; }
static uint64_t rs = 0x1234567887654321ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return uint32_t(rs >> 16); }
int main(int argc, char** argv)
{
    long iters = atol(argv[1]); size_t ok = 0, n_all = 0;
    for (int a = 2; a < argc; a++) {
        std::vector<uint8_t> seed = slurp(argv[a]);
        for (long it = 0; it < iters; it++) {
            size_t n = seed.size();
            if (rnd() % 4 == 0) n = rnd() % (seed.size() + 1);
            uint8_t* buf = static_cast<uint8_t*>(malloc(n ? n : 1));
            memcpy(buf, seed.data(), n);
            const int flips = rnd() % 5;
            for (int k = 0; k < flips && n; k++) { const size_t at = rnd() % n; buf[at] = (rnd() & 1) ? uint8_t(rnd()) : uint8_t(buf[at] ^ (1u << (rnd() % 8))); }
            if (n > 4 && (rnd() & 1)) { const uint32_t c = rcgpu_crc32_ffv1(buf, n - 4); buf[n - 4] = uint8_t(c >> 24); buf[n - 3] = uint8_t(c >> 16); buf[n - 2] = uint8_t(c >> 8); buf[n - 1] = uint8_t(c); }
            for (uint32_t pf = 0; pf < RCGPU_PIX_COUNT; pf += 1 + rnd() % 5) {
                rcgpu_ffv1_config cfg; memset(&cfg, 0, sizeof cfg); cfg.width = 64; cfg.height = 48; cfg.pixfmt = pf; cfg.line_bytes = 64 * 8;
                ok += rcgpu_ffv1_config_from_record(buf, n, &cfg) == 0; n_all++;
                // ... and the reader of the first slice header, with the same bytes as a "packet" and with a mutated packet behind a good record
                ok += rcgpu_ffv1_config_from_stream(buf, n, buf, n, &cfg) == 0; n_all++;
                ok += rcgpu_ffv1_config_from_stream(seed.data(), seed.size(), buf, n, &cfg) == 0; n_all++;
            }
            // ... and the reader of everything parameters::Parse accepts (rcgpu_ffv1_stream_parse): the bytes as a record with themselves as the first
            // packet, as a record-less (version 0 / 1) packet, and as a mutated packet behind the good record
            for (int mode = 0; mode < 3; mode++) {
                rcgpu_ffv1_stream* st = nullptr;
                const int r = mode == 0 ? rcgpu_ffv1_stream_parse(buf, n, buf, n, &st) : mode == 1 ? rcgpu_ffv1_stream_parse(nullptr, 0, buf, n, &st)
                                                                                                     : rcgpu_ffv1_stream_parse(seed.data(), seed.size(), buf, n, &st);
                n_all++;
                if (r == 0) {
                    ok++;
                    rcgpu_ffv1_stream_info info;
                    if (rcgpu_ffv1_stream_get_info(st, &info) != 0 || info.quant_table_set_count < 1 || info.quant_table_set_count > 8 || info.version > 3) abort();
                    for (uint32_t g = 0; g < info.quant_table_set_index_count && g < 3; g++) if (info.quant_table_set_index[g] >= info.quant_table_set_count) abort();
                    rcgpu_ffv1_stream_free(st);
                } else if (st) abort();
            }
            free(buf);
        }
    }
    printf("fuzz_rec: %zu calls, %zu accepted\n", n_all, ok);
}
