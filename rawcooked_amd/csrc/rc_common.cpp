// rc_common.cpp -- error plumbing, version string, pixel-format table.
#include "rc_common.h"

namespace rc {

static thread_local char g_err[1024] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
void clear_error() { g_err[0] = 0; }

static const pix_desc k_pix[RCGPU_PIX_COUNT] = {
    /* RGB8             */ {  8, 3, 3, false, false, kFieldsBytes,  0, 1 },
    /* RGB10_FILLEDA_BE */ { 10, 3, 4, true,  true,  kFieldsBytes,  0, 1 },
    /* RGB10_FILLEDA_LE */ { 10, 3, 4, false, true,  kFieldsBytes,  0, 1 },
    /* RGB12_FILLEDA_BE */ { 12, 3, 6, true,  true,  kFieldsBytes,  0, 1 },
    /* RGB12_FILLEDA_LE */ { 12, 3, 6, false, true,  kFieldsBytes,  0, 1 },
    /* RGB16_BE         */ { 16, 3, 6, true,  false, kFieldsBytes,  0, 1 },
    /* RGB16_LE         */ { 16, 3, 6, false, false, kFieldsBytes,  0, 1 },
    /* RGBA8            */ {  8, 4, 4, false, false, kFieldsBytes,  0, 1 },
    /* RGBA16_BE        */ { 16, 4, 8, true,  false, kFieldsBytes,  0, 1 },
    /* RGBA16_LE        */ { 16, 4, 8, false, false, kFieldsBytes,  0, 1 },
    /* Y8               */ {  8, 1, 1, false, false, kFieldsBytes,  0, 1 },
    /* Y16_BE           */ { 16, 1, 2, true,  false, kFieldsBytes,  0, 1 },
    /* Y16_LE           */ { 16, 1, 2, false, false, kFieldsBytes,  0, 1 },
    /* RGB12_PACKED_BE  */ { 12, 3, 0, true,  true,  kFieldsPacked, 0, 1 },
    /* RGBA10_FILLEDA_BE*/ { 10, 4, 0, true,  false, kFieldsTop,    0, 3 },
    /* RGBA10_FILLEDA_LE*/ { 10, 4, 0, false, false, kFieldsTop,    0, 3 },
    /* RGBA12_PACKED_BE */ { 12, 4, 0, true,  false, kFieldsPacked, 0, 2 },
    /* RGBA12_FILLEDA_BE*/ { 12, 4, 8, true,  false, kFieldsBytes,  0, 1 },
    /* RGBA12_FILLEDA_LE*/ { 12, 4, 8, false, false, kFieldsBytes,  0, 1 },
    /* Y10_FILLEDA_BE   */ { 10, 1, 0, true,  false, kFieldsLow,    2, 1 },
    /* Y10_FILLEDB_BE   */ { 10, 1, 0, true,  false, kFieldsLow,    0, 1 },
    /* Y12_PACKED_BE    */ { 12, 1, 0, true,  false, kFieldsPacked, 0, 1 },
    /* EXR_RGB16        */ { 16, 3, 0, false, false, kFieldsExr,    0, 1 },
};
const pix_desc& pix(uint32_t pixfmt) { return k_pix[pixfmt < RCGPU_PIX_COUNT ? pixfmt : 0]; }

// 0: the line does not fit 32 bits (a width only a hostile header names; every caller refuses it)
uint32_t payload_line_bytes(uint32_t pixfmt, uint32_t width, bool dpx_padding)
{
    const pix_desc& d = pix(pixfmt);
    const uint64_t nfields = uint64_t(width) * d.planes;
    uint64_t n;
    if (d.fields == kFieldsExr) n = 8 + 6 * uint64_t(width);                          // EXR.cpp:601-606
    else if (d.fields == kFieldsPacked) n = (nfields * 12 + 31) / 32 * 4;
    else if (d.fields != kFieldsBytes) n = (nfields + 2) / 3 * 4;
    else { n = uint64_t(width) * d.bytes_pp; if (dpx_padding) n = (n + 3) / 4 * 4; }
    return n > 0xFFFFFFFFull ? 0u : uint32_t(n);
}
uint64_t payload_bytes(uint32_t pixfmt, uint32_t width, uint32_t height, uint32_t line_bytes, uint32_t flags)
{
    (void)pixfmt;
    if (flags & RCGPU_FLAG_ALTERN) return (uint64_t(width) * height + 2) / 3 * 4;
    return uint64_t(line_bytes) * height;
}

}  // namespace rc

extern "C" const char* rcgpu_last_error(void) { return rc::g_err; }
extern "C" const char* rcgpu_version(void)
{
    return "rcgpu version 0.1 -- MI355X-native FFV1/FLAC/Matroska encode path for RAWcooked (gfx950)";
}
