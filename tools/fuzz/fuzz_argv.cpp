// Mutation fuzzing of the job front end on the CPU (ASan + UBSan): rcgpu_main_ffmpeg_argv with mutated argv vectors and mutated ffconcat
// lists, up to -- not including -- device creation.  The grammar is what output::FFmpeg_Command emits (CLI/Output.cpp:81-332: per-stream
// input options, `-f image2 -c:v dpx -start_number N -i <template>` or `-f concat -safe 0 -i <list>` with `file '...'` / `duration` lines
// (:138-251), -map, the sorted output options, -attach / -metadata:s:K pairs, `-f matroska <out>`, `-f framemd5 <file>`); every run adds
// `-rcgpu_plan_only 1`, with which rcgpu_encode analyses the inputs (sequences enumerated, files probed, grids and rates settled), prints
// the plan and stops.  Everything the front end links to on the device side is stubbed below: there is no device (rcgpu_device_count 0) and nothing else
// may be reached.
//   fuzz_argv <iterations> <work dir with a.dpx b.dpx a.tif a.exr a.wav seq/f_%06d.dpx>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include "rcgpu.h"
#include "pipeline.h"
#include "ffv1_internal.h"

// ---- device-side symbols job.cpp refers to: reaching one of them in plan-only mode is a bug of the front end
[[noreturn]] static void reached(const char* what) { dprintf(2, "fuzz_argv: %s reached without a device\n", what); abort(); }
namespace rc {
struct pipeline::impl {};
pipeline::pipeline() {}                          // (a job that codes nothing on the device -- PCM copied as it is -- constructs one and never prepares it)
pipeline::~pipeline() {}
int pipeline::prepare(const std::vector<pipe_video>&, const pipe_options&) { reached("pipeline::prepare"); }
int pipeline::run(const std::vector<pipe_frame>&, const pipe_io&, pipe_stats*) { reached("pipeline::run"); }
rcgpu_ffv1* pipeline::encoder(uint32_t) const { reached("pipeline::encoder"); }
uint32_t pipeline::batch_frames(uint32_t) const { reached("pipeline::batch_frames"); }
std::vector<uint8_t> ffv1_config_record_for(const rcgpu_ffv1_config&) { reached("ffv1_config_record_for"); }
size_t ffv1_max_packet_bytes_for(const rcgpu_ffv1_config&) { reached("ffv1_max_packet_bytes_for"); }
}
extern "C" {
int rcgpu_device_count(void) { return 0; }      // a command line whose mutation lost `-rcgpu_plan_only 1` (taken as a path, say) is a job on a box without a device: refused
size_t rcgpu_ffv1_config_record(const rcgpu_ffv1*, uint8_t*, size_t) { reached("rcgpu_ffv1_config_record"); }
int rcgpu_ffv1_framemd5_last(rcgpu_ffv1*, uint32_t, uint8_t*, uint64_t*) { reached("rcgpu_ffv1_framemd5_last"); }
size_t rcgpu_ffv1_max_packet_bytes(const rcgpu_ffv1*) { reached("rcgpu_ffv1_max_packet_bytes"); }
size_t rcgpu_flac_codec_private(const rcgpu_flac*, uint8_t*, size_t) { reached("rcgpu_flac_codec_private"); }
int rcgpu_flac_create(const rcgpu_flac_config*, rcgpu_flac**) { reached("rcgpu_flac_create"); }
void rcgpu_flac_destroy(rcgpu_flac*) { reached("rcgpu_flac_destroy"); }
int rcgpu_flac_encode_host(rcgpu_flac*, const uint8_t*, uint64_t, uint8_t*, size_t, uint32_t*, uint32_t, uint32_t*) { reached("rcgpu_flac_encode_host"); }
}

static uint64_t rs = 0x243F6A8885A308D3ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return uint32_t(rs >> 16); }
static std::string junk(size_t n) { std::string s; for (size_t i = 0; i < n; i++) s += char(1 + rnd() % 255); return s; }

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const long iters = atol(argv[1]);
    const std::string W = argv[2];
    if (chdir(W.c_str()) != 0) return 2;
    // the command lines the reference prints for: a DPX sequence, a TIFF, an EXR with its extra input option, DPX + WAV + sidecar + framemd5, a gapped sequence
    const std::vector<std::vector<std::string>> seeds = {
        { "ffmpeg", "-xerror", "-framerate", "24.000000", "-r", "24.000000", "-f", "image2", "-c:v", "dpx", "-start_number", "000000", "-i", "seq/f_%06d.dpx", "-c:a", "flac", "-c:v", "ffv1",
          "-coder", "1", "-context", "1", "-f", "matroska", "-g", "1", "-level", "3", "-slicecrc", "1", "-slices", "4", "-y", "-attach", "rev", "-metadata:s:1", "mimetype=application/octet-stream",
          "-metadata:s:1", "filename=RAWcooked reversibility data", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-xerror", "-framerate", "25", "-r", "25", "-f", "image2", "-c:v", "tiff", "-i", "a.tif", "-c:v", "ffv1", "-coder", "2", "-context", "0", "-g", "1", "-level", "3", "-slices", "4", "-n", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-xerror", "-consider_float16_as_uint16", "1", "-framerate", "24", "-r", "24", "-f", "image2", "-c:v", "exr", "-i", "a.exr", "-c:v", "ffv1", "-g", "1", "-slices", "4", "-metadata:s:v",
          "WARNING=x", "-y", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-xerror", "-framerate", "30000/1001", "-r", "30000/1001", "-f", "image2", "-c:v", "dpx", "-start_number", "0", "-i", "seq/f_%06d.dpx", "-i", "a.wav", "-map", "0", "-map", "1", "-c:a", "flac",
          "-c:v", "ffv1", "-g", "1", "-level", "1", "-slices", "1", "-y", "-attach", "b.dpx", "-metadata:s:2", "mimetype=application/octet-stream", "-metadata:s:2", "filename=side/car.bin", "-attach", "rev",
          "-metadata:s:3", "mimetype=application/octet-stream", "-metadata:s:3", "filename=RAWcooked reversibility data", "-f", "matroska", "out.mkv", "-an", "-f", "framemd5", "out.framemd5" },
        { "ffmpeg", "-xerror", "-framerate", "24", "-r", "24", "-f", "concat", "-safe", "0", "-c:v", "dpx", "-i", "list.txt", "-c:v", "ffv1", "-g", "1", "-slices", "4", "-vf", "vflip", "-y", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-i", "a.wav", "-c:a", "copy", "-y", "-f", "matroska", "out.mkv" },
        // rawcooked -c:v ffv1_vulkan:1 (CLI/Global.cpp:367-378) and test/vulkan.sh's hand-made form
        { "ffmpeg", "-xerror", "-framerate", "24.000000", "-r", "24.000000", "-f", "image2", "-c:v", "dpx", "-start_number", "000000", "-i", "seq/f_%06d.dpx", "-c:a", "flac", "-c:v", "ffv1_vulkan",
          "-f", "matroska", "-init_hw_device", "vulkan=vk:1", "-slices", "4", "-vf", "hwupload", "-y", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-xerror", "-init_hw_device", "vulkan=vk:0,debug=0", "-hwaccel", "vulkan", "-hwaccel_output_format", "vulkan", "-framerate", "24", "-f", "image2", "-c:v", "dpx", "-start_number", "0",
          "-i", "seq/f_%06d.dpx", "-vf", "hwupload", "-c:v", "ffv1_vulkan", "-coder", "1", "-g", "1", "-level", "3", "-slices", "4", "-rcgpu_own_slice_buffers", "1", "-y", "-f", "matroska", "out.mkv" },
        { "ffmpeg", "-version" },
    };
    static const char* const words[] = { "-i", "-f", "concat", "image2", "matroska", "framemd5", "-attach", "-metadata:s:1", "-metadata:s:v", "filename=RAWcooked reversibility data", "filename=", "-map", "-an",
                                         "-y", "-n", "-slices", "-coder", "-context", "-level", "-g", "-slicecrc", "-vf", "vflip", "-c:v", "-c:a", "ffv1", "flac", "copy", "dpx", "tiff", "exr", "-start_number",
                                         "-framerate", "-r", "-safe", "-rcgpu_context_model", "compact", "-rcgpu_plan_only", "0", "1", "2", "3", "576", "4294967295", "-1", "99999999999999999999", "0/0", "1/0",
                                         "24000/1001", "1e309", "nan", "", "%", "seq/f_%06d.dpx", "seq/f_%09d.dpx", "seq/f_%d.dpx", "seq/f_%s.dpx", "seq/f_%06d%06d.dpx", "a.dpx", "b.dpx", "a.tif", "a.exr", "a.wav",
                                         "list.txt", "missing.dpx", ".", "/", "/dev/null", "seq", "rev", "-xerror", "-version",
                                         "-init_hw_device", "vulkan", "vulkan=vk", "vulkan=vk:1", "vulkan=vk:99999999999", "vulkan=:", "vulkan:2,x=y", "vulkanx", "cuda=cu:0", "ffv1_vulkan", "hwupload",
                                         "vflip,hwupload", "hwupload,,vflip", ",", "-hwaccel", "-rcgpu_own_slice_buffers" };
    const size_t nwords = sizeof words / sizeof words[0];
    size_t accepted = 0, refused = 0;
    FILE* devnull = fopen("/dev/null", "w");
    for (long it = 0; it < iters; it++) {
        std::vector<std::string> a = seeds[rnd() % seeds.size()];
        // ---- the ffconcat list of this round (Output.cpp:236-246 writes `file '<path>'` + `duration <seconds>` per frame; at 25 fps bare paths)
        {
            FILE* f = fopen("list.txt", "wb");
            const uint32_t kind = it % 5000 == 17 ? 0 : 1 + rnd() % 11;
            const uint32_t n = kind == 0 ? 200000 + rnd() % 800000 : rnd() % 40;        // every 5000th list has 10^5..10^6 entries
            if (rnd() % 4) fputs("ffconcat version 1.0\n", f);
            for (uint32_t i = 0; i < n; i++) {
                const char* nl = (rnd() % 9 == 0) ? "\r\n" : "\n";
                const uint32_t how = rnd() % 16;
                const int k = kind == 0 ? int(i) : int(rnd() % 6);
                if (how == 0) fprintf(f, "file 'seq/f_%06d.dpx%s", k, nl);                       // no closing quote
                else if (how == 1) fprintf(f, "file seq/f_%06d.dpx%s", k, nl);                    // no quotes
                else if (how == 2) fprintf(f, "file ''%s", nl);
                else if (how == 3) fprintf(f, "duration %s%s", junk(rnd() % 12).c_str(), nl);
                else if (how == 4) fprintf(f, "%s%s", junk(rnd() % 200).c_str(), nl);
                else if (how == 5) fprintf(f, "file '%s'%s", std::string(rnd() % 5000, 'x').c_str(), nl);
                else if (how == 6) fprintf(f, "seq/f_%06d.dpx%s", k, nl);                         // the 25 fps form: bare paths
                else if (how == 7) fprintf(f, "file 'missing_%u.dpx'%sduration 0.041667%s", rnd(), nl, nl);
                else if (how == 8) fprintf(f, "file 'a.tif'%s", nl);                              // another format in the middle of a DPX list
                else fprintf(f, "file 'seq/f_%06d.dpx'%sduration 0.0416%02u%s", k, nl, rnd() % 100, nl);
            }
            if (rnd() % 5 == 0) fputs("file 'seq/f_000000.dpx'", f);                             // no newline at the end
            fclose(f);
        }
        // ---- mutate the argv
        const int muts = rnd() % 5;
        for (int m = 0; m < muts && !a.empty(); m++) {
            const size_t at = 1 + rnd() % a.size();
            switch (rnd() % 7) {
            case 0: a.insert(a.begin() + std::min(at, a.size()), words[rnd() % nwords]); break;
            case 1: if (at < a.size()) a.erase(a.begin() + at); break;
            case 2: if (at < a.size()) a[at] = words[rnd() % nwords]; break;
            case 3: if (at < a.size()) a[at] = junk(rnd() % 64); break;
            case 4: if (at < a.size() && !a[at].empty()) a[at][rnd() % a[at].size()] = char(rnd()); break;
            case 5: if (at + 1 < a.size()) std::swap(a[at], a[at + 1]); break;
            default: a.resize(1 + rnd() % a.size()); break;                                        // cut the command line short
            }
        }
        if (rnd() % 50) { a.push_back("-rcgpu_plan_only"); a.push_back("1"); }
        else { a.insert(a.begin() + 1, "1"); a.insert(a.begin() + 1, "-rcgpu_plan_only"); }       // (in front: an input option by position -- Output.cpp:111-131)
        bool plan = false;                                                                         // the mutation may have removed or re-valued it: then only a refusal is acceptable
        for (size_t i = 1; i + 1 < a.size(); i++) plan |= a[i] == "-rcgpu_plan_only" && a[i + 1] == "1";
        if (!plan) { a.push_back("-rcgpu_plan_only"); a.push_back("1"); }
        std::vector<const char*> av; for (auto& s : a) av.push_back(s.c_str());
        // exact-size heap copies so that ASan sees reads past an argument's end
        std::vector<char*> heap; for (const char* s : av) { char* h = static_cast<char*>(malloc(strlen(s) + 1)); strcpy(h, s); heap.push_back(h); }
        FILE* so = stdout; FILE* se = stderr; stdout = devnull; stderr = devnull;
        const int r = rcgpu_main_ffmpeg_argv(int(heap.size()), heap.data());
        stdout = so; stderr = se;
        (r == 0 ? accepted : refused)++;
        for (char* h : heap) free(h);
        unlink("out.mkv"); unlink("out.framemd5");
    }
    printf("fuzz_argv: %ld command lines (%ld lists of 10^5..10^6 entries among them), %zu planned, %zu refused\n", iters, (iters + 4982) / 5000, accepted, refused);
    return 0;
}
