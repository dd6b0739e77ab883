// rc_common.h -- internal helpers shared by the host-side translation units of librcgpu.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rcgpu.h"

// Measuring switches -- kernels skipped, mappings forced, copy-stream counts -- exist in the TIMING build only (`make timing`:
// librcgpu_timing.so, -DRCGPU_TIMING_BUILD; tools/sweep_*.sh and tools/ab.sh load it through RCGPU_LIB).  The shipped library reads none of
// them: the only inputs that change its bytes are the fields of its configuration structs, as the reference's only ones are its
// command line's (CLI/Global.cpp:938-989).
#ifdef RCGPU_TIMING_BUILD
#include <cstdlib>
#define TIMING_ENV(name) getenv(name)
#else
#define TIMING_ENV(name) static_cast<const char*>(nullptr)
#endif

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
    uint8_t bytes_pp;    // bytes per pixel in the payload; 0 when fields straddle bytes (see fields)
    bool    big_endian;
    bool    gb_swap;     // FFV1 codes 9..15-bit RGB without alpha with G and B exchanged (Lib/Transform/Transform.cpp:104,126,338,363)
    uint8_t fields;      // layout of the bit-packed DPX flavors, a stream of 32-bit words per line:
                         //   kFieldsBytes   whole bytes per pixel
                         //   kFieldsPacked  12-bit fields filling each big-endian word from the LSB up (Transform.cpp:214-322, 521-550, 905-990)
                         //   kFieldsTop     three 10-bit fields per word at <<22, <<12, <<2 (RGBA 10-bit FilledA, Transform.cpp:445-518)
                         //   kFieldsLow     three 10-bit fields per word at <<fill, <<10+fill, <<20+fill (Y 10-bit FilledA/B, Transform.cpp:781-796)
    uint8_t fill;        // kFieldsLow: 2 = FilledA, 0 = FilledB
    uint8_t px_per_block;// pixels per indivisible block when slices may NOT cut a block (DPX.cpp:184-207 without BlockSpan), else 1
};
enum { kFieldsBytes = 0, kFieldsPacked = 1, kFieldsTop = 2, kFieldsLow = 3, kFieldsExr = 4 };   // kFieldsExr: [y][size][B..][G..][R..] per line, u16 LE
const pix_desc& pix(uint32_t pixfmt);
// bytes between payload lines: DPX pads every line to 32 bit (RawFrame.cpp:109, DPX.cpp:463-483), TIFF does not
uint32_t payload_line_bytes(uint32_t pixfmt, uint32_t width, bool dpx_padding);
// bytes of one payload; RCGPU_FLAG_ALTERN payloads are one word stream without line padding (DPX.cpp:465-469)
uint64_t payload_bytes(uint32_t pixfmt, uint32_t width, uint32_t height, uint32_t line_bytes, uint32_t flags);

}  // namespace rc
