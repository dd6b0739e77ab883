// ref_padding_probe.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REAL reference parser: it is compiled against the headers
// under /root/reference and linked with the reference's own objects (oracle/Makefile.ref, target `padding_probe`), runs
// dpx::ParseBuffer (Lib/Uncompressed/DPX/DPX.cpp:250-634) with --check-padding on every file named on the command line, and prints
// what the parser keeps to itself: whether it stored an `In` block and its In_FirstNonZero (DPX.cpp:501-608).  tests/golden/
// make_padding_golden.py turns that output into the vectors that pin oracle/dpx_oracle.c and rcgpu_dpx_padding_scan_device.
#include <bitset>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#define private public        // the two members of interest are private to class dpx
#define protected public
#include "Lib/Uncompressed/DPX/DPX.h"
#include "Lib/Compressed/RAWcooked/RAWcooked.h"
#undef private
#undef protected

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_padding_probe <scratch reversibility file> <file.dpx>...\n"); return 2; }
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (data.empty()) { printf("%s unreadable\n", argv[a]); continue; }
        errors Errors;
        rawcooked RAWcooked;
        user_mode Mode = AlwaysYes;
        RAWcooked.Mode = &Mode;
        RAWcooked.Errors = &Errors;
        RAWcooked.FileName = argv[1];
        RAWcooked.OutputFileName = "probe.dpx";
        dpx DPX(&Errors);
        DPX.Actions.set(Action_Encode);
        DPX.Actions.set(Action_CheckPadding);
        DPX.RAWcooked = &RAWcooked;
        DPX.FileName = &RAWcooked.OutputFileName;
        DPX.Parse(buffer_view(data.data(), data.size()));
        const bool supported = DPX.IsSupported();
        printf("%s supported=%d flavor=%s in_size=%zu first_nonzero=%lld\n", argv[a], int(supported), supported ? DPX.Flavor_String().c_str() : "-",
               DPX.In.Size(), DPX.In.Size() ? (long long)DPX.In_FirstNonZero : -1LL);
        RAWcooked.Close();
        RAWcooked.Delete();
    }
    return 0;
}
