// rc_common.h -- internal helpers shared by the host-side translation units of librcgpu.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rcgpu.h"

namespace rc {

// Thread-local error text behind rcgpu_last_error().  Returns `code` so callers can `return fail(...)`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

inline uint16_t rd16(const uint8_t* p, bool be) { return be ? uint16_t((p[0] << 8) | p[1]) : uint16_t((p[1] << 8) | p[0]); }
inline uint32_t rd32(const uint8_t* p, bool be)
{
    return be ? (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]
              : (uint32_t(p[3]) << 24) | (uint32_t(p[2]) << 16) | (uint32_t(p[1]) << 8) | p[0];
}
inline uint64_t rd64le(const uint8_t* p) { return uint64_t(rd32(p, false)) | (uint64_t(rd32(p + 4, false)) << 32); }

// Pixel-format facts (one row per RCGPU_PIX_*), used by the probes, the encoder and the kernels' host side.
struct pix_desc {
    uint8_t bits;        // bits_per_raw_sample
    uint8_t planes;      // 1 (Y), 3 (RGB) or 4 (RGBA)
    uint8_t bytes_pp;    // bytes per pixel in the payload
    bool    big_endian;
    bool    gb_swap;     // FFV1 codes 9..15-bit RGB without alpha with G and B exchanged (Lib/Transform/Transform.cpp:104,126,338,363)
};
const pix_desc& pix(uint32_t pixfmt);

}  // namespace rc
