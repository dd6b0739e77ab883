// ref_probe.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REAL reference's file parsers: compiled against the headers under
// /root/reference and linked with the reference's own objects (oracle/Makefile.ref, target `probe`).  Every case -- the bytes of a file --
// is shown to wav, dpx, tiff and exr in the order CLI/Main.cpp:328-390 tries them, with the actions of an encoding run, and the line says
// what the analysis would go on with: which parser recognised the file, whether it supports it, the flavor string it prints with --info,
// and slice_x * slice_y -- the number it passes to FFmpeg as -slices (Main.cpp:350-386, Output.cpp:299-304).  rcgpu's own probes
// (rcgpu_dpx_probe, rcgpu_tiff_probe, rcgpu_exr_probe, rcgpu_wav_probe: what the shim analyses its inputs with) are held to these lines
// on mutated headers (tests/golden/make_probe_golden.py, tests/test_host.py).  Nothing here restates the reference.
//
//   ref_probe <scratch file for the reversibility data> <cases.bin>      cases: { u32 size, bytes } repeated (little endian)
//   one line per case:  <wav|dpx|tiff|exr|none> supported=<0|1> flavor=<string or -> slices=<n>
#include <bitset>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <new>
#include <string>
#include <vector>
#include <sys/resource.h>
#include "Lib/Uncompressed/DPX/DPX.h"
#include "Lib/Uncompressed/TIFF/TIFF.h"
#include "Lib/Uncompressed/EXR/EXR.h"
#include "Lib/Uncompressed/WAV/WAV.h"
#include "Lib/Compressed/RAWcooked/RAWcooked.h"

static size_t slices_of(wav&) { return 0; }
template <class parser> static size_t slices_of(parser& P) { return P.slice_x * P.slice_y; }      // (not part of the common base)

template <class parser> static bool show(const char* name, const std::vector<uint8_t>& data, const char* scratch)
{
    errors Errors;
    rawcooked RAWcooked;
    user_mode Mode = AlwaysYes;
    RAWcooked.Mode = &Mode;
    RAWcooked.Errors = &Errors;
    RAWcooked.FileName = scratch;
    RAWcooked.OutputFileName = "probe.bin";
    parser P(&Errors);
    P.Actions.set(Action_Encode);
    P.RAWcooked = &RAWcooked;
    P.FileName = &RAWcooked.OutputFileName;
    // (a header may name any size; under the address-space limit set in main an allocation of that size is a bad_alloc, not this machine's memory)
    try { P.Parse(buffer_view(data.data(), data.size())); } catch (const std::bad_alloc&) { printf("%s supported=0 flavor=- slices=0 bad_alloc\n", name); RAWcooked.Close(); RAWcooked.Delete(); return true; }
    const bool detected = P.IsDetected();
    if (detected) {
        const bool supported = P.IsSupported();
        printf("%s supported=%d flavor=%s slices=%zu\n", name, int(supported), supported ? P.Flavor_String().c_str() : "-", supported ? slices_of(P) : size_t(0));
    }
    RAWcooked.Close();
    RAWcooked.Delete();
    return detected;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_probe <scratch reversibility file> <cases.bin>\n"); return 2; }
    struct rlimit lim = { size_t(4) << 30, size_t(4) << 30 };
    setrlimit(RLIMIT_AS, &lim);
    FILE* f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    for (;;) {
        uint32_t n;
        if (fread(&n, 4, 1, f) != 1) break;
        std::vector<uint8_t> data(n);
        if (n && fread(data.data(), 1, n, f) != n) break;
        if (show<wav>("wav", data, argv[1])) continue;
        if (show<dpx>("dpx", data, argv[1])) continue;
        if (show<tiff>("tiff", data, argv[1])) continue;
        if (show<exr>("exr", data, argv[1])) continue;
        printf("none supported=0 flavor=- slices=0\n");
    }
    fclose(f);
    return 0;
}
