// ref_flac_decode.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REAL reference's FLAC path: the wrapper RAWcooked makes for an A_FLAC
// track (CreateWrapper(format::FLAC), Lib/CoDec/Wrapper.cpp:131-384: libFLAC's stream decoder fed block by block, its output turned into
// WAV-style bytes), compiled against the headers under /root/reference and linked with the reference's own objects, libFLAC included
// (oracle/Makefile.ref, target `flac_decode`).  It is given a track's CodecPrivate and its blocks the way track_info gives them
// (Track.cpp:199-230: SetConfig from the WAV flavor, OutOfBand, Process per block) and writes the bytes every decoded block hands to the
// frame writer.  Nothing here restates the reference or libFLAC.
//
//   ref_flac_decode <cases.bin> <out.bin>
//   cases: { u32 bits (of the WAV: 8 = unsigned, else signed little endian), u32 codec_private_size, bytes, u32 n_blocks, { u32 size, bytes } x n } repeated
//   out:   per case { u32 complaints, u32 size, the PCM bytes of all blocks }
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "Lib/CoDec/Wrapper.h"
#include "Lib/Utils/RawFrame/RawFrame.h"

struct collector : public raw_frame_process
{
    std::vector<uint8_t> pcm;
private:
    void FrameCall(raw_frame* RawFrame) override { pcm.insert(pcm.end(), RawFrame->Buffer().Data(), RawFrame->Buffer().Data() + RawFrame->Buffer().Size()); }
};

static bool rd32(FILE* f, uint32_t& v) { return fread(&v, 4, 1, f) == 1; }
static bool rdbytes(FILE* f, std::vector<uint8_t>& b, uint32_t n) { b.assign(size_t(n) + 16, 0); return !n || fread(b.data(), 1, n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_flac_decode <cases.bin> <out.bin>\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); FILE* o = fopen(argv[2], "wb");
    if (!f || !o) { fprintf(stderr, "cannot open the files\n"); return 2; }
    for (uint32_t c = 0;; c++) {
        uint32_t bits, cps, n;
        std::vector<uint8_t> cp, blk;
        if (!rd32(f, bits) || !rd32(f, cps) || !rdbytes(f, cp, cps) || !rd32(f, n)) break;
        raw_frame RawFrame;
        collector Out;
        RawFrame.FrameProcess = &Out;
        audio_wrapper* W = static_cast<audio_wrapper*>(CreateWrapper(format::FLAC, nullptr));
        W->RawFrame = &RawFrame;
        W->SetConfig(uint8_t(bits), bits <= 8 ? sign::U : sign::S, endianness::LE);
        uint32_t bad = W->OutOfBand(cp.data(), cps) ? 1 : 0;
        for (uint32_t k = 0; k < n; k++) {
            uint32_t size;
            if (!rd32(f, size) || !rdbytes(f, blk, size)) { fprintf(stderr, "case %u: cases file cut short\n", c); return 2; }
            bad += W->Process(blk.data(), size) ? 1 : 0;
        }
        delete W;
        const uint32_t size = uint32_t(Out.pcm.size());
        fwrite(&bad, 4, 1, o); fwrite(&size, 4, 1, o);
        if (size) fwrite(Out.pcm.data(), 1, size, o);
        printf("case %u: %u blocks, %u bytes of PCM, %u complaints\n", c, n, size, bad);
    }
    fclose(f); fclose(o);
    return 0;
}
