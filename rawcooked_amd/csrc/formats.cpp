// formats.cpp -- header probes for the uncompressed inputs of the encode path.
//
// The reference's parsers (Lib/Uncompressed/{DPX,TIFF,WAV}) also write the reversibility data; that half
// stays in RAWcooked (north_star: "reversibility-data attachment unchanged").  The encoder side only needs
// what FFmpeg's demuxers extracted for it: where the payload starts, its geometry and layout.  Each probe
// re-derives the reference's flavor / slice decisions so that a stream described only by a file template
// yields the same encoder configuration `rawcooked -d` prints.
#include "rc_common.h"
#include <algorithm>

using namespace rc;

// slice_x rule shared by dpx::ParseBuffer (DPX.cpp:428-441) and tiff::ParseBuffer (TIFF.cpp:657-669)
static uint32_t reference_slice_x(uint32_t width, uint32_t height, uint32_t bitdepth)
{
    uint32_t s = 4;
    if (width >= 1440) s <<= 1;
    if (width >= 2880) s <<= 1;
    if (bitdepth > 10) s = s * 3 / 2;
    s = std::min(s, width / 2);
    s = std::min(s, height / 2);
    return s ? s : 1;
}

static void flavor_string(char out[64], const char* container, uint32_t pixfmt, const char* dpx_packing)
{
    // "<container>/Raw/<RGB|RGBA|Y>/<bits>bit/U/<BE|LE>[/<Packing>]": Raw_Flavor_String (Lib/Common/Common.cpp:123-139) and, for
    // DPX bit depths that are not whole bytes, the packing name appended (DPX.cpp:762-778).  8-bit rows of both tables are "LE".
    const pix_desc& d = pix(pixfmt);
    const char* cs = d.planes == 1 ? "Y" : d.planes == 4 ? "RGBA" : "RGB";
    snprintf(out, 64, "%s/Raw/%s/%ubit/U/%s%s%s", container, cs, d.bits, d.bits > 8 && d.big_endian ? "BE" : "LE",
             dpx_packing && d.bits % 8 ? "/" : "", dpx_packing && d.bits % 8 ? dpx_packing : "");
}

extern "C" int rcgpu_dpx_probe(const uint8_t* f, size_t size, rcgpu_image_info* out)
{
    clear_error();
    if (!f || !out) return fail(1, "dpx: null argument");
    if (size < 1664) return fail(2, "dpx: file too small for a header");
    memset(out, 0, sizeof *out);
    bool be;
    const uint32_t magic = rd32(f, true);
    if (magic == 0x53445058) be = true;          // "SDPX"   DPX.cpp:296-299
    else if (magic == 0x58504453) be = false;    // "XPDS"   DPX.cpp:292-295
    else return fail(3, "dpx: bad magic number");
    const uint32_t offset_to_image = rd32(f + 4, be);
    const uint32_t version = rd32(f + 8, true);
    if (version != 0 && version != 0x56312E30 && version != 0x56322E30 && version != 0x76312E30 && version != 0x76322E30)
        return fail(4, "dpx: unsupported version number");                                   // DPX.cpp:306-318
    uint32_t industry = rd32(f + 28, be);
    if (industry == 0xFFFFFFFFu) industry = 0;
    const uint32_t encryption = rd32(f + 660, be);
    if (encryption != 0xFFFFFFFFu && encryption != 0) return fail(5, "dpx: encrypted content");
    const uint16_t orientation = rd16(f + 768, be);
    if (rd16(f + 770, be) != 1) return fail(6, "dpx: number of image elements is not 1");
    const uint32_t width = rd32(f + 772, be), height = rd32(f + 776, be);
    if (rd32(f + 780, be) != 0) return fail(7, "dpx: signed data");
    const uint8_t descriptor = f[800], bitdepth = f[803];
    // (the reference keeps the packing field in an 8-bit enum, `Info.Packing = (packing)Get_X2()`, DPX.cpp:134,348: only its low byte counts)
    const uint16_t packing = rd16(f + 804, be) & 0xFF, encoding = rd16(f + 806, be);
    if (encoding) return fail(8, "dpx: RLE encoding");
    uint32_t offset_to_data = rd32(f + 808, be);
    if (offset_to_data) {
        if (offset_to_data < 1664 || offset_to_data > size) return fail(9, "dpx: bad offset to data");
        if (offset_to_data != offset_to_image) return fail(10, "dpx: offset to image data differs from element offset");
    } else
        offset_to_data = offset_to_image;                                                     // DPX.cpp:352-361
    if (rd32(f + 812, be) != 0) return fail(11, "dpx: end-of-line padding");
    // orientation: only 2 (bottom to top) means anything to the reference, and only for the flavors it flags VFlip (DPX.cpp:411-412,495); any
    // other value is a field of the header like the others -- the picture is coded as its lines lie in the file
    if (!width || !height) return fail(13, "dpx: empty image");

    // flavor table, DPX.cpp:184-231 (Tested + Also rows reachable with the layouts this encoder implements)
    int pf = -1;
    const bool filledA = packing == 1, filledB = packing == 2, packed = packing == 0;
    switch (descriptor) {
    case 50:   // RGB
        if (bitdepth == 8 && (packed || filledA)) pf = RCGPU_PIX_RGB8;
        else if (bitdepth == 10 && filledA) pf = be ? RCGPU_PIX_RGB10_FILLEDA_BE : RCGPU_PIX_RGB10_FILLEDA_LE;
        else if (bitdepth == 12 && filledA) pf = be ? RCGPU_PIX_RGB12_FILLEDA_BE : RCGPU_PIX_RGB12_FILLEDA_LE;
        else if (bitdepth == 12 && packed && be) pf = RCGPU_PIX_RGB12_PACKED_BE;
        else if (bitdepth == 16 && (packed || filledA)) pf = be ? RCGPU_PIX_RGB16_BE : RCGPU_PIX_RGB16_LE;
        break;
    case 51:   // RGBA
        if (bitdepth == 8 && (packed || filledA)) pf = RCGPU_PIX_RGBA8;
        else if (bitdepth == 10 && filledA) pf = be ? RCGPU_PIX_RGBA10_FILLEDA_BE : RCGPU_PIX_RGBA10_FILLEDA_LE;
        else if (bitdepth == 12 && filledA) pf = be ? RCGPU_PIX_RGBA12_FILLEDA_BE : RCGPU_PIX_RGBA12_FILLEDA_LE;
        else if (bitdepth == 12 && packed && be) pf = RCGPU_PIX_RGBA12_PACKED_BE;
        else if (bitdepth == 16 && (packed || filledA)) pf = be ? RCGPU_PIX_RGBA16_BE : RCGPU_PIX_RGBA16_LE;
        break;
    case 6:    // Y
        if (bitdepth == 8 && (packed || filledA)) pf = RCGPU_PIX_Y8;
        else if (bitdepth == 10 && be && (filledA || filledB)) pf = filledA ? RCGPU_PIX_Y10_FILLEDA_BE : RCGPU_PIX_Y10_FILLEDB_BE;
        else if (bitdepth == 12 && packed && be) pf = RCGPU_PIX_Y12_PACKED_BE;
        else if (bitdepth == 16 && (packed || filledA || packing == 3)) pf = be ? RCGPU_PIX_Y16_BE : RCGPU_PIX_Y16_LE;
        break;
    }
    if (pf < 0)
        return fail(14, "dpx: flavor (descriptor %u, %u bit, packing %u, %s) is not supported",
                    descriptor, bitdepth, packing, be ? "BE" : "LE");
    const pix_desc& d = pix(uint32_t(pf));
    // only the 12-bit Packed flavors may be stored bottom-up, only Y 10-bit may run words across lines (DPX.cpp:189,202-204,409-412)
    const bool vflip_ok = pf == RCGPU_PIX_RGB12_PACKED_BE || pf == RCGPU_PIX_Y12_PACKED_BE;
    if (orientation == 2 && !vflip_ok) return fail(12, "dpx: orientation 2 is not supported for this flavor");
    const bool altern = bitdepth == 10 && descriptor != 50 && size >= 1563 &&
                        (!memcmp(f + 160, "Lasergraphics Inc.", 18) || !memcmp(f + 160, "DIAMANT-Film", 12) || !memcmp(f + 1556, "Scanity", 7));   // DPX.cpp:363-368
    if (altern && d.fields != kFieldsLow) return fail(12, "dpx: words running across lines are not supported for this flavor");
    out->flags = (orientation == 2 ? RCGPU_FLAG_VFLIP : 0) | (altern ? RCGPU_FLAG_ALTERN : 0);
    out->width = width; out->height = height; out->pixfmt = uint32_t(pf); out->bits_per_sample = d.bits;
    out->line_bytes = altern ? 0 : payload_line_bytes(uint32_t(pf), width, true);
    if (!altern && !out->line_bytes) return fail(13, "dpx: a line of %u pixels does not fit 32 bits", width);
    out->data_offset = offset_to_data;
    out->data_size = payload_bytes(uint32_t(pf), width, height, out->line_bytes, out->flags);
    if (out->data_offset + out->data_size > size) return fail(15, "dpx: truncated image data");
    out->slices = rcgpu_reference_slices(width, height, bitdepth, d.px_per_block);
    if (!out->slices) return fail(16, "dpx: no slice layout keeps %u-pixel blocks whole at width %u", d.px_per_block, width);   // DPX.cpp:443-456
    if (industry && size >= 1944) {     // DPX.cpp:370-387
        auto f32 = [&](size_t o) { uint32_t u = rd32(f + o, be); float v; memcpy(&v, &u, 4); return (u == 0xFFFFFFFFu || v != v) ? 0.0 : double(v); };
        const double film = f32(1724), tv = f32(1940);
        out->framerate = film ? film : tv;
    }
    flavor_string(out->flavor, "DPX", out->pixfmt, packed ? "Packed" : filledA ? "FilledA" : filledB ? "FilledB" : nullptr);
    return 0;
}

namespace {
struct tiff_reader {
    const uint8_t* f; size_t size; bool be; bool bad = false;
    uint16_t u16(size_t o) { if (o + 2 > size) { bad = true; return 0; } return rd16(f + o, be); }
    uint32_t u32(size_t o) { if (o + 4 > size) { bad = true; return 0; } return rd32(f + o, be); }
    // all values of one IFD entry (TIFF.cpp:283-378)
    std::vector<uint32_t> values(size_t entry)
    {
        std::vector<uint32_t> v;
        const uint16_t type = u16(entry + 2);
        const uint32_t count = u32(entry + 4);
        static const uint8_t tsz[6] = { 0, 1, 1, 2, 4, 8 };
        if (!type || type > 5 || !count || count > (1u << 24)) { bad = true; return v; }
        const size_t esz = tsz[type];
        size_t o = entry + 8;
        if (esz * count > 4) o = u32(entry + 8);
        for (uint32_t i = 0; i < count && !bad; i++, o += esz)
            v.push_back(esz == 1 ? (o < size ? f[o] : (bad = true, 0)) : esz == 2 ? u16(o) : u32(o));
        return v;
    }
};
}

extern "C" int rcgpu_tiff_probe(const uint8_t* f, size_t size, rcgpu_image_info* out)
{
    clear_error();
    if (!f || !out) return fail(1, "tiff: null argument");
    if (size < 8) return fail(2, "tiff: file too small");
    memset(out, 0, sizeof *out);
    tiff_reader r{ f, size, false };
    const uint32_t magic = rd32(f, true);
    if (magic == 0x49492A00) r.be = false;       // "II*\0"  TIFF.cpp:391-394
    else if (magic == 0x4D4D002A) r.be = true;   // "MM\0*"  TIFF.cpp:395-398
    else return fail(3, "tiff: bad magic number");
    const uint32_t ifd = r.u32(4);
    if (ifd > size || ifd + 2 > size) return fail(4, "tiff: bad first IFD offset");
    const uint32_t n = r.u16(ifd);
    if (size_t(ifd) + 2 + 12 * size_t(n) + 4 > size) return fail(5, "tiff: bad directory count");
    uint32_t width = 0, height = 0, bps = 0, photometric = ~0u, compression = 0, spp = 1, planar = 1, fill = 1,
             orientation = 1, sample_format = 1, extra = 0;
    bool has_w = false, has_h = false, has_bps = false, has_comp = false, has_extra = false;
    std::vector<uint32_t> offs, counts;
    for (uint32_t i = 0; i < n; i++) {
        const size_t e = size_t(ifd) + 2 + 12 * size_t(i);
        const uint16_t tag = r.u16(e);
        auto one = [&]() { auto v = r.values(e); return v.empty() ? 0u : v[0]; };
        switch (tag) {
        case 256: width = one(); has_w = true; break;
        case 257: height = one(); has_h = true; break;
        case 258: { auto v = r.values(e); if (!v.empty()) { bps = v[0]; has_bps = true; for (auto x : v) if (x != bps) return fail(6, "tiff: BitsPerSample values differ"); } break; }
        case 259: compression = one(); has_comp = true; break;
        case 262: photometric = one(); break;
        case 266: fill = one(); break;
        case 273: offs = r.values(e); break;
        case 274: orientation = one(); break;
        case 277: spp = one(); break;
        case 279: counts = r.values(e); break;
        case 284: planar = one(); break;
        case 338: extra = one(); has_extra = true; break;
        case 339: sample_format = one(); break;
        case 254: if (one()) return fail(7, "tiff: NewSubfileType"); break;
        case 255: return fail(7, "tiff: SubfileType");
        case 317: case 320: case 322: case 323: case 324: case 325:   // Predictor, ColorMap, tiles (TIFF.cpp:527-551)
            return fail(8, "tiff: unsupported IFD tag %u", tag);
        default: break;
        }
    }
    if (r.bad) return fail(9, "tiff: directory points outside the file");
    // checks of TIFF.cpp:556-590
    if (!has_w || !has_h || !has_bps || !has_comp || photometric == ~0u) return fail(10, "tiff: missing mandatory tag");
    if (compression != 1) return fail(11, "tiff: compressed content");
    if (fill != 1 || orientation != 1 || planar != 1 || sample_format != 1) return fail(12, "tiff: unsupported FillOrder/Orientation/PlanarConfiguration/SampleFormat");
    if (offs.empty() || offs.size() != counts.size()) return fail(13, "tiff: bad StripOffsets/StripByteCounts");
    int pf = -1;
    if (photometric == 2 && spp == 3 && !has_extra) {
        if (bps == 8) pf = RCGPU_PIX_RGB8;
        else if (bps == 16) pf = r.be ? RCGPU_PIX_RGB16_BE : RCGPU_PIX_RGB16_LE;
    } else if (photometric == 2 && spp == 4 && has_extra && extra == 2) {
        if (bps == 8) pf = RCGPU_PIX_RGBA8;
        else if (bps == 16 && !r.be) pf = RCGPU_PIX_RGBA16_LE;         // TIFF.cpp:161-162: no RGBA16 BE row
    } else if (photometric == 1 && spp == 1 && !has_extra) {
        if (bps == 8) pf = RCGPU_PIX_Y8;
        else if (bps == 16) pf = r.be ? RCGPU_PIX_Y16_BE : RCGPU_PIX_Y16_LE;
    }
    if (pf < 0) return fail(14, "tiff: flavor (photometric %u, %u x %u bit, %s) is not supported", photometric, spp, bps, r.be ? "BE" : "LE");
    // strips must be contiguous and exactly cover the image (TIFF.cpp:638-645, 675-678)
    uint64_t last = uint64_t(offs[0]) + counts[0];
    for (size_t i = 1; i < offs.size(); i++) {
        if (last != offs[i]) return fail(15, "tiff: strips are not contiguous");
        last += counts[i];
    }
    const pix_desc& d = pix(uint32_t(pf));
    out->width = width; out->height = height; out->pixfmt = uint32_t(pf); out->bits_per_sample = d.bits;
    out->line_bytes = payload_line_bytes(uint32_t(pf), width, false);
    if (!out->line_bytes) return fail(13, "tiff: a line of %u pixels does not fit 32 bits", width);
    out->data_offset = offs[0];
    out->data_size = uint64_t(out->line_bytes) * height;
    if (out->data_offset + out->data_size != last) return fail(16, "tiff: strip sizes do not match the image size");
    if (last > size) return fail(17, "tiff: truncated image data");
    const uint32_t sx = reference_slice_x(width, height, bps);
    out->slices = sx * sx;
    flavor_string(out->flavor, "TIFF", out->pixfmt, nullptr);
    return 0;
}

extern "C" int rcgpu_exr_probe(const uint8_t* f, size_t size, rcgpu_image_info* out)
{
    clear_error();
    if (!f || !out) return fail(1, "exr: null argument");
    memset(out, 0, sizeof *out);
    if (size < 8 || rd32(f, true) != 0x762F3101) return fail(3, "exr: bad magic number");                 // EXR.cpp:210-217
    const uint32_t version = rd32(f + 4, true);
    if ((version >> 24) != 2) return fail(4, "exr: unsupported version number");                          // EXR.cpp:220-227
    if (version & 0xFFFFFF) return fail(4, "exr: version flags (tiles / long names / deep / multi-part) are not supported");
    size_t o = 8;
    uint32_t width = 0, height = 0, dw = 0, dh = 0;
    bool have_display = false, rgb_half = false, have_channels = false;
    double fps = 0;
    auto cstr = [&](size_t at, size_t& len) -> const char* {      // NUL-terminated name of at most 31 characters
        len = 0;
        while (at + len < size && f[at + len] && len <= 31) len++;
        if (at + len >= size || len > 31) return nullptr;
        return reinterpret_cast<const char*>(f + at);
    };
    // names the reference knows and skips (EXR.cpp:310-520); anything else makes it refuse the file
    static const char* const kSkipped[] = { "acesImageContainerFlag", "adoptedNeutral", "capDate", "chromaticities", "comments", "expTime", "focalLength", "focus",
        "imageCounter", "isoSpeed", "lensMake", "lensSerialNumber", "originalImageFlag", "owner", "pixelAspectRatio", "reelName", "recorderFirmwareVersion",
        "recorderMake", "recorderModel", "storageMediaSerialNumber", "timeCode", "timecodeRate" };
    static const char* const kSkippedPrefix[] = { "arri.", "camera", "com.arri.", "interim." };
    for (;;) {
        size_t nlen, tlen;
        const char* name = cstr(o, nlen);
        if (!name) return fail(5, "exr: bad attribute name");
        if (!nlen) { o++; break; }                                 // end of header
        const char* type = cstr(o + nlen + 1, tlen);
        if (!type || o + nlen + 1 + tlen + 1 + 4 > size) return fail(5, "exr: bad attribute type");
        o += nlen + 1 + tlen + 1;
        const uint32_t asz = rd32(f + o, false); o += 4;
        if (asz > size - o) return fail(5, "exr: attribute larger than the file");
        const uint8_t* v = f + o;
        const std::string n(name), t(type);
        if (n == "channels" && t == "chlist") {                    // EXR.cpp:331-398
            size_t q = 0; uint32_t code = 0, count = 0; bool same_half = true;
            while (q + 1 < asz) {
                size_t cl = 0; while (q + cl < asz && v[q + cl]) cl++;
                if (q + cl + 17 > asz) return fail(6, "exr: bad channel list");
                if (count > 3 || cl != 1) code = 0xFFFFFFFFu; else code |= uint32_t(v[q]) << (8 * count);
                q += cl + 1;
                if (rd32(v + q, false) != 1) same_half = false;                                              // pixel type HALF
                if (rd32(v + q + 4, false) != 0 || rd32(v + q + 8, false) != 1 || rd32(v + q + 12, false) != 1)
                    return fail(6, "exr: channel list features (pLinear / sampling) are not supported");
                q += 16; count++;
            }
            if (q + 1 != asz || v[q]) return fail(6, "exr: bad channel list");
            rgb_half = code == 0x00524742 && same_half;           // "B","G","R"
            have_channels = true;
        } else if (n == "compression" && t == "compression") { if (asz != 1 || v[0]) return fail(7, "exr: compressed files are not supported"); }
        else if (n == "dataWindow" && t == "box2i") {
            if (asz != 16 || rd32(v, false) || rd32(v + 4, false)) return fail(8, "exr: dataWindow does not start at the origin");
            width = rd32(v + 8, false); height = rd32(v + 12, false);
            if (!width || !height) return fail(8, "exr: dataWindow");                                       // EXR.cpp:421-423 (sic: 1-pixel-wide pictures are refused)
        } else if (n == "displayWindow" && t == "box2i") {
            if (asz != 16 || rd32(v, false) || rd32(v + 4, false)) return fail(8, "exr: displayWindow does not start at the origin");
            dw = rd32(v + 8, false); dh = rd32(v + 12, false); have_display = true;
        } else if ((n == "framesPerSecond" || n == "captureRate") && t == "rational") {
            if (asz != 8) return fail(9, "exr: %s", n.c_str());
            const uint32_t num = rd32(v, false), den = rd32(v + 4, false);
            if (n == "framesPerSecond") { if (!num || !den) return fail(9, "exr: framesPerSecond"); fps = double(num) / den; }
            else if (num && den && fps == 0) fps = double(num) / den;
        } else if (n == "imageRotation" && t == "float") { if (asz != 4 || rd32(v, false)) return fail(9, "exr: imageRotation"); }
        else if (n == "lineOrder" && t == "lineOrder") { if (asz != 1 || v[0]) return fail(9, "exr: only increasing-Y line order is supported"); }
        else if (n == "screenWindowCenter" && t == "v2f") { if (asz != 8 || rd32(v, false) || rd32(v + 4, false)) return fail(9, "exr: screenWindowCenter"); }
        else if (n == "screenWindowWidth" && t == "float") { if (asz != 4 || rd32(v, false) != 0x3F800000u) return fail(9, "exr: screenWindowWidth"); }
        else {
            bool known = false;
            for (const char* k : kSkipped) known |= n == k;
            for (const char* k : kSkippedPrefix) known |= n.compare(0, strlen(k), k) == 0;
            if (!known) return fail(10, "exr: header field %s is not supported", n.c_str());               // EXR.cpp:521-525,547-548
        }
        o += asz;
    }
    if (have_display && (width != dw || height != dh)) return fail(8, "exr: displayWindow differs from dataWindow");
    if (!have_channels || !rgb_half) return fail(14, "exr: flavor (only B,G,R channels of type HALF) is not supported");
    width++; height++;
    out->width = width; out->height = height; out->pixfmt = RCGPU_PIX_EXR_RGB16; out->bits_per_sample = 16;
    out->line_bytes = payload_line_bytes(RCGPU_PIX_EXR_RGB16, width, false);
    if (!out->line_bytes) return fail(13, "exr: a line of %u pixels does not fit 32 bits", width);
    out->data_offset = o + 8ull * height;                                                                   // line offset table, EXR.cpp:597-598
    out->data_size = uint64_t(out->line_bytes) * height;
    if (out->data_offset + out->data_size > size) return fail(15, "exr: truncated image data");
    uint32_t sx = 4;                                                                                       // EXR.cpp:579-590: the 16-bit rule always
    if (width >= 1440) sx <<= 1;
    if (width >= 2880) sx <<= 1;
    sx = sx * 3 / 2;
    sx = std::min(sx, width / 2); sx = std::min(sx, height / 2);
    if (!sx) sx = 1;
    out->slices = sx * sx;
    out->framerate = fps;
    snprintf(out->flavor, sizeof out->flavor, "EXR/Raw/RGB/16bit/F/BE");                                   // EXR_Flavor_String, EXR.cpp:660-666
    return 0;
}

extern "C" int rcgpu_wav_probe(const uint8_t* f, size_t size, rcgpu_audio_info* out)
{
    clear_error();
    if (!f || !out) return fail(1, "wav: null argument");
    if (size < 12) return fail(2, "wav: file too small");
    memset(out, 0, sizeof *out);
    const uint32_t first = rd32(f, true);
    const bool rf64 = first == 0x52463634;                                   // WAV.cpp:281-286
    if ((first != 0x52494646 && !rf64) || rd32(f + 8, true) != 0x57415645) return fail(3, "wav: not a RIFF/RF64 WAVE file");
    uint64_t pos = 12, ds64_data = 0;
    bool have_fmt = false;
    uint16_t tag = 0;
    while (pos + 8 <= size) {
        const uint32_t name = rd32(f + pos, true);
        uint64_t csize = rd32(f + pos + 4, false);
        pos += 8;
        if (name == 0x64733634) {                 // "ds64"  WAV.cpp:438-459
            if (pos + 28 > size) return fail(4, "wav: truncated ds64 chunk");
            ds64_data = rd64le(f + pos + 8);
            if (rd32(f + pos + 24, false)) return fail(5, "wav: ds64 table is not supported");
        } else if (name == 0x666D7420) {          // "fmt "  WAV.cpp:461-542
            if (csize < 16 || pos + csize > size) return fail(6, "wav: bad fmt chunk");
            tag = rd16(f + pos, false);
            out->channels = rd16(f + pos + 2, false);
            out->sample_rate = rd32(f + pos + 4, false);
            const uint32_t avg = rd32(f + pos + 8, false);
            out->block_align = rd16(f + pos + 12, false);
            out->bits_per_sample = rd16(f + pos + 14, false);
            // validated here, before any arithmetic uses them: an all-zero fmt chunk is coherent with itself (0 == 0) and would
            // otherwise reach the `% block_align` of the data chunk (the reference rejects such files, WAV.cpp:125-221)
            if (out->channels < 1 || out->channels > 8) return fail(13, "wav: %u channels not supported", out->channels);
            if (out->bits_per_sample != 8 && out->bits_per_sample != 16 && out->bits_per_sample != 24 && out->bits_per_sample != 32)
                return fail(14, "wav: %u-bit PCM is not supported", out->bits_per_sample);
            if (!out->block_align || !out->sample_rate) return fail(7, "wav: BlockAlign or SamplesPerSec is zero");
            // (both sides in 32 bits, as the reference has them, WAV.cpp:476: a field whose top bits wrap away is coherent to it)
            if (uint32_t(avg * 8u) != uint32_t(uint32_t(out->channels) * out->bits_per_sample * out->sample_rate)) return fail(7, "wav: incoherent AvgBytesPerSec");
            if (out->block_align * 8u != out->channels * out->bits_per_sample) return fail(7, "wav: incoherent BlockAlign");
            if (tag == 0xFFFE) {
                if (csize != 40 || rd16(f + pos + 16, false) != 22) return fail(8, "wav: bad WAVE_FORMAT_EXTENSIBLE chunk");
                if (rd16(f + pos + 18, false) != out->bits_per_sample) return fail(8, "wav: ValidBitsPerSample differs");
                tag = uint16_t(rd32(f + pos + 24, false));
            }
            if (tag != 1 && tag != 3) return fail(9, "wav: format tag %u is neither integer PCM nor IEEE float", tag);
            have_fmt = true;
        } else if (name == 0x64617461) {          // "data"  WAV.cpp:390-436
            if (!have_fmt) return fail(10, "wav: data chunk before fmt chunk");
            if (rf64 && (csize == 0xFFFFFFFFu || csize == ds64_data)) csize = ds64_data;
            if (csize > size - pos) return fail(11, "wav: truncated data chunk");
            if (csize % out->block_align) return fail(12, "wav: data size is not a multiple of the block size");
            out->data_offset = pos; out->data_size = csize;
            // flavor table WAV.cpp:125-221: channels {1,2,4,6,8}... kept permissive up to 8 channels (checked with the fmt chunk).
            // WAV_Tested, WAV.cpp:125-203: 8/16/24/32-bit integer, 32-bit float; more than 24 bits (and float) cannot be FLAC and
            // travel as PCM (`-c:a copy`, CLI/Main.cpp:300-317) -- the job decides, the probe only describes
            if (tag == 3 && out->bits_per_sample != 32) return fail(14, "wav: %u-bit float is not supported", out->bits_per_sample);
            out->format_tag = tag;
            snprintf(out->flavor, sizeof out->flavor, "WAV/PCM/%ukHz/%ubit/%uch/%s/LE", out->sample_rate / 1000,
                     out->bits_per_sample, out->channels, tag == 3 ? "F" : out->bits_per_sample == 8 ? "U" : "S");
            return 0;
        }
        pos += csize;
        if ((pos & 1) && pos < size && !f[pos]) pos++;     // padding byte, WAV.cpp:360-365
    }
    return fail(15, "wav: no data chunk");
}

// The reference's flavor string -> pixel layout: "DPX/Raw/RGB/10bit/U/BE/FilledA", "TIFF/Raw/RGBA/16bit/U/LE", "EXR/Raw/RGB/16bit/F/BE"
// (DPX_Flavor_String DPX.cpp:762-778, TIFF_Flavor_String TIFF.cpp:744-750, EXR.cpp:660-666).  What a decoder-side binding has at hand:
// the flavor restored from the reversibility data (raw_frame::Flavor / Flavor_Private).
extern "C" int rcgpu_pixfmt_from_flavor(const char* flavor, uint32_t* pixfmt)
{
    clear_error();
    if (!flavor || !pixfmt) return fail(1, "flavor: null argument");
    if (!strcmp(flavor, "EXR/Raw/RGB/16bit/F/BE")) { *pixfmt = RCGPU_PIX_EXR_RGB16; return 0; }
    for (uint32_t pf = 0; pf < RCGPU_PIX_COUNT; pf++) {
        if (pf == RCGPU_PIX_EXR_RGB16) continue;
        const pix_desc& d = pix(pf);
        const char* packing = d.fields == kFieldsPacked ? "Packed" : (d.fields == kFieldsLow && d.fill == 0) ? "FilledB" : "FilledA";
        char t[64];
        flavor_string(t, "DPX", pf, packing);
        if (!strcmp(t, flavor)) { *pixfmt = pf; return 0; }
        if (d.bits % 8 == 0) { flavor_string(t, "TIFF", pf, nullptr); if (!strcmp(t, flavor)) { *pixfmt = pf; return 0; } }
    }
    return fail(14, "flavor %s is not supported", flavor);
}

extern "C" uint32_t rcgpu_reference_slices(uint32_t width, uint32_t height, uint32_t bitdepth, uint32_t pixels_per_block)
{
    uint32_t sx = reference_slice_x(width, height, bitdepth);
    if (pixels_per_block > 1)                       // DPX.cpp:443-456: no slice may start inside a block
        for (; sx; sx--) if (width % (sx * pixels_per_block) == 0) break;
    return sx * sx;
}

extern "C" int rcgpu_slices_to_grid(uint32_t n, uint32_t* num_h, uint32_t* num_v)
{
    // FFmpeg's ffv1 encoder maps -slices N to the first (v in [2,31], h in [v,2v-1]) with h*v == N; the set of
    // N this accepts is exactly valid_slices in Project/GNU/CLI/test/slices.sh:12.  N == 1 is the 1x1 case the
    // reference allows with -level 0/1 only (CLI/Global.cpp:976-985).
    if (!num_h || !num_v) return 1;
    if (n == 1) { *num_h = *num_v = 1; return 0; }
    for (uint32_t v = 2; v < 32; v++)
        for (uint32_t h = v; h < 2 * v; h++)
            if (h * v == n) { *num_h = h; *num_v = v; return 0; }
    return fail(1, "slices: %u cannot be laid out as h x v with v <= h < 2v", n);
}
