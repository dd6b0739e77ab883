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
    /* RGB8            */ {  8, 3, 3, false, false },
    /* RGB10_FILLEDA_BE*/ { 10, 3, 4, true,  true  },
    /* RGB10_FILLEDA_LE*/ { 10, 3, 4, false, true  },
    /* RGB12_FILLEDA_BE*/ { 12, 3, 6, true,  true  },
    /* RGB12_FILLEDA_LE*/ { 12, 3, 6, false, true  },
    /* RGB16_BE        */ { 16, 3, 6, true,  false },
    /* RGB16_LE        */ { 16, 3, 6, false, false },
    /* RGBA8           */ {  8, 4, 4, false, false },
    /* RGBA16_BE       */ { 16, 4, 8, true,  false },
    /* RGBA16_LE       */ { 16, 4, 8, false, false },
    /* Y8              */ {  8, 1, 1, false, false },
    /* Y16_BE          */ { 16, 1, 2, true,  false },
    /* Y16_LE          */ { 16, 1, 2, false, false },
};
const pix_desc& pix(uint32_t pixfmt) { return k_pix[pixfmt < RCGPU_PIX_COUNT ? pixfmt : 0]; }

}  // namespace rc

extern "C" const char* rcgpu_last_error(void) { return rc::g_err; }
extern "C" const char* rcgpu_version(void)
{
    return "rcgpu version 0.1 -- MI355X-native FFV1/FLAC/Matroska encode path for RAWcooked (gfx950)";
}
