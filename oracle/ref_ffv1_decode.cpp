// ref_ffv1_decode.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REAL reference's FFV1 decoder: compiled against the headers under
// /root/reference and linked with the reference's own objects (oracle/Makefile.ref, target `ffv1_decode`).  It hands a track's CodecPrivate
// and its frames to ffv1_frame::OutOfBand / ::Process (FFV1_Frame.cpp:105-228) the way track_info does (Track.cpp:199-230: the flavor the
// reversibility data names, the picture's size), without a slice pool, and writes what the decoder leaves in the frame's plane -- the bytes
// frame_writer would put behind the file's header.  No Matroska file is needed: tests decode thousands of random streams this way
// (tests/test_oracle.py, tests/test_gpu_check.py) where tests/golden/make_golden.py wraps thirteen of them into files for `rawcooked --check`.
// Nothing here restates the reference: every sample is decoded and placed by its code; this file only feeds it and copies the result out.
//
//   ref_ffv1_decode <cases.bin> <out.bin>
//   cases: { u32 flavor_len, flavor string ("DPX/Raw/RGB/16bit/U/BE", "TIFF/...", "EXR/..."), u32 flags (1 = VFlip, 2 = Altern), u32 width, u32 height,
//            u32 record_size, record, u32 n_frames, { u32 size, bytes } x n_frames } repeated (little endian)
//   out:   per frame { u32 verdict (0 = decoded without complaint), u32 size, the plane's bytes }; stdout: one line per case
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/resource.h>
#include "Lib/CoDec/FFV1/FFV1_Frame.h"
#include "Lib/Utils/RawFrame/RawFrame.h"
#include "Lib/Uncompressed/DPX/DPX.h"
#include "Lib/Uncompressed/TIFF/TIFF.h"
#include "Lib/Uncompressed/EXR/EXR.h"

static bool rd32(FILE* f, uint32_t& v) { return fread(&v, 4, 1, f) == 1; }
static bool rdbytes(FILE* f, std::vector<uint8_t>& b, uint32_t n) { b.assign(size_t(n) + 16, 0); return !n || fread(b.data(), 1, n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_ffv1_decode <cases.bin> <out.bin>\n"); return 2; }
    struct rlimit lim = { size_t(6) << 30, size_t(6) << 30 };
    setrlimit(RLIMIT_AS, &lim);
    FILE* f = fopen(argv[1], "rb"); FILE* o = fopen(argv[2], "wb");
    if (!f || !o) { fprintf(stderr, "cannot open the files\n"); return 2; }
    for (uint32_t c = 0;; c++) {
        uint32_t flen, flags, width, height, rsize, nframes;
        std::vector<uint8_t> fl, rec, pkt;
        if (!rd32(f, flen) || !rdbytes(f, fl, flen) || !rd32(f, flags) || !rd32(f, width) || !rd32(f, height) || !rd32(f, rsize) || !rdbytes(f, rec, rsize) || !rd32(f, nframes)) break;
        const std::string flavor(reinterpret_cast<const char*>(fl.data()), flen);
        raw_frame RawFrame;
        bool known = false;
        // which of the reference's flavors carries this name (its own tables, its own strings)
        for (size_t i = 0; i < dpx::flavor_Max && !known; i++) if (DPX_Flavor_String(uint8_t(i)) == flavor) { RawFrame.Flavor = raw_frame::flavor::DPX; RawFrame.Flavor_Private = i; known = true; }
        for (size_t i = 0; i < tiff::flavor_Max && !known; i++) if (TIFF_Flavor_String(uint8_t(i)) == flavor) { RawFrame.Flavor = raw_frame::flavor::TIFF; RawFrame.Flavor_Private = i; known = true; }
        for (size_t i = 0; i < exr::flavor_Max && !known; i++) if (EXR_Flavor_String(uint8_t(i)) == flavor) { RawFrame.Flavor = raw_frame::flavor::EXR; RawFrame.Flavor_Private = i; known = true; }
        if (known && RawFrame.Flavor == raw_frame::flavor::DPX) {
            if (flags & 1) RawFrame.Flavor_Private |= uint64_t(1) << int(dpx::feature::VFlip);
            if (flags & 2) RawFrame.Flavor_Private |= uint64_t(1) << int(dpx::feature::Altern);
        }
        ffv1_frame Frame(nullptr);                               // no slice pool: the slices are decoded one after the other (FFV1_Frame.cpp:216-220)
        Frame.RawFrame = &RawFrame;
        Frame.SetWidth(width); Frame.SetHeight(height);
        if (known && rsize) Frame.OutOfBand(rec.data(), rsize);
        const char* record_error = Frame.ErrorMessage();
        uint32_t bad = 0;
        for (uint32_t k = 0; k < nframes; k++) {
            uint32_t size;
            if (!rd32(f, size) || !rdbytes(f, pkt, size)) { fprintf(stderr, "case %u: cases file cut short\n", c); return 2; }
            uint32_t verdict = 1, n = 0;
            const uint8_t* data = nullptr;
            if (known && !record_error) {
                const bool failed = Frame.Process(pkt.data(), size);
                verdict = failed || Frame.ErrorMessage() ? 1 : 0;
                if (!RawFrame.Planes().empty() && RawFrame.Plane(0)) { data = RawFrame.Plane(0)->Buffer().Data(); n = uint32_t(RawFrame.Plane(0)->Buffer().Size()); }
            }
            bad += verdict;
            fwrite(&verdict, 4, 1, o); fwrite(&n, 4, 1, o);
            if (n) fwrite(data, 1, n, o);
        }
        printf("case %u %s %ux%u frames %u complaints %u%s%s\n", c, known ? flavor.c_str() : "(flavor unknown to the reference)", width, height, nframes, bad,
               Frame.ErrorMessage() ? " " : "", Frame.ErrorMessage() ? Frame.ErrorMessage() : "");
    }
    fclose(f); fclose(o);
    return 0;
}
