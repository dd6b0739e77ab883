// job.cpp -- the drop-in boundary: one encode job = what RAWcooked hands to `ffmpeg` today.
//
// rcgpu_encode() replaces `system(Command)` at CLI/Output.cpp:356; rcgpu_main_ffmpeg_argv() accepts the argv
// grammar that output::FFmpeg_Command assembles (CLI/Output.cpp:81-332), so an UNMODIFIED rawcooked can run this
// encoder through `--bin-name` (CLI/Global.cpp:543-550).  Files are read on the host, frames are sharded
// round-robin over the selected devices in batches, packets come back over each device's own PCIe link and one
// muxer writes them in frame order.  No collective is involved: every frame is a key frame (-g 1).
#include "rc_common.h"
#include "pipeline.h"
#include "ffv1_internal.h"
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <fcntl.h>
#include <map>
#include <memory>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

using namespace rc;

namespace {

struct mapped_file {
    const uint8_t* data = nullptr; size_t size = 0; int fd = -1;
    mapped_file() = default;
    mapped_file(const mapped_file&) = delete;
    ~mapped_file() { if (data) munmap(const_cast<uint8_t*>(data), size); if (fd >= 0) ::close(fd); }
    bool open(const std::string& path, bool keep_fd = false)
    {
        int f = ::open(path.c_str(), O_RDONLY);
        if (f < 0) return false;
        struct stat st;
        if (fstat(f, &st) != 0 || st.st_size <= 0) { ::close(f); return false; }
        void* p = mmap(nullptr, size_t(st.st_size), PROT_READ, MAP_PRIVATE, f, 0);   // the reference's default reader, Lib/Utils/FileIO/FileIO.cpp:274
        if (keep_fd) fd = f; else ::close(f);
        if (p == MAP_FAILED) return false;
        data = static_cast<const uint8_t*>(p); size = size_t(st.st_size);
        return true;
    }
    // the bulk of a file into a (pinned) buffer: read() copies straight out of the page cache, without a fault per page
    bool read_at(uint64_t off, uint8_t* dst, size_t n) const
    {
        if (fd < 0) { if (off + n > size) return false; memcpy(dst, data + off, n); return true; }
        while (n) {
            const ssize_t r = ::pread(fd, dst, n, off_t(off));
            if (r < 0) { if (errno == EINTR) continue; return false; }
            if (r == 0) return false;
            dst += r; off += uint64_t(r); n -= size_t(r);
        }
        return true;
    }
};

bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

std::string lower_ext(const std::string& p)
{
    const size_t d = p.rfind('.');
    std::string e = d == std::string::npos ? "" : p.substr(d + 1);
    for (char& c : e) c = char(tolower(c));
    return e;
}

// "%06d"-style template expansion (the only conversion input::DetectSequence emits, CLI/Input.cpp:305-306)
bool expand_template(const std::string& tpl, unsigned long long n, std::string& out)
{
    const size_t pc = tpl.find('%');
    if (pc == std::string::npos) return false;
    size_t i = pc + 1; int width = 0; bool zero = false;
    if (i < tpl.size() && tpl[i] == '0') { zero = true; i++; }
    while (i < tpl.size() && isdigit(uint8_t(tpl[i]))) { width = width * 10 + (tpl[i++] - '0'); if (width > 20) return false; }     // (a frame number has 20 digits at most; `num` below holds them)
    if (i >= tpl.size() || tpl[i] != 'd') return false;
    char num[32];
    snprintf(num, sizeof num, zero ? "%0*llu" : "%*llu", width, n);
    out = tpl.substr(0, pc) + num + tpl.substr(i + 1);
    return true;
}

struct rational { uint32_t num = 24, den = 1; };
rational parse_framerate(const char* s)
{
    rational r;
    if (!s || !*s) return r;
    char* end = nullptr;
    const double a = strtod(s, &end);
    if (end && *end == '/') {
        const double b = strtod(end + 1, nullptr);
        if (a > 0 && b > 0) { r.num = uint32_t(std::llround(a)); r.den = uint32_t(std::llround(b)); }
        return r;
    }
    if (!(a > 0)) return r;
    // decimal: NTSC-style rates are n*1000/1001
    const double n1001 = a * 1001.0 / 1000.0;
    if (std::fabs(a - std::round(a)) < 1e-6) { r.num = uint32_t(std::llround(a)); r.den = 1; }
    else if (std::fabs(n1001 - std::round(n1001)) < 2e-3) { r.num = uint32_t(std::llround(n1001)) * 1000; r.den = 1001; }
    else { r.num = uint32_t(std::llround(a * 1000.0)); r.den = 1000; }
    return r;
}

struct options {
    std::map<std::string, std::string> kv;
    const char* get(const char* k) const { auto i = kv.find(k); return i == kv.end() ? nullptr : i->second.c_str(); }
    long num(const char* k, long def) const { const char* v = get(k); return v && *v ? strtol(v, nullptr, 10) : def; }
    bool has(const char* k) const { return kv.count(k) != 0; }
};

struct video_plan {
    std::vector<std::string> files;
    rcgpu_image_info info{};
    bool tiff = false;
    rational fps;
    uint32_t num_h = 1, num_v = 1;
    bool vflip = false;
    int track = 0;
};
struct audio_plan {
    std::string file;
    rcgpu_audio_info info{};
    std::vector<uint8_t> frames;              // concatenated FLAC frames
    std::vector<uint32_t> frame_sizes;
    uint32_t block_size = 0;                  // samples per frame / per PCM block
    std::unique_ptr<mapped_file> pcm;         // -c:a copy: the WAV itself; blocks are cut out of its data chunk
    const uint8_t* bytes() const { return pcm ? pcm->data + info.data_offset : frames.data(); }
    std::vector<uint8_t> codec_private;
    int track = 0;
};

int probe_image(const std::string& path, bool& tiff, rcgpu_image_info& info)
{
    mapped_file f;
    if (!f.open(path)) return fail(30, "cannot open %s: %s", path.c_str(), strerror(errno));
    const std::string ext = lower_ext(path);
    const bool looks_tiff = f.size >= 4 && ((f.data[0] == 'I' && f.data[1] == 'I') || (f.data[0] == 'M' && f.data[1] == 'M'));
    tiff = looks_tiff || ext == "tif" || ext == "tiff";
    if (f.size >= 4 && f.data[0] == 0x76 && f.data[1] == 0x2F && f.data[2] == 0x31 && f.data[3] == 0x01) { tiff = false; return rcgpu_exr_probe(f.data, f.size, &info); }
    return tiff ? rcgpu_tiff_probe(f.data, f.size, &info) : rcgpu_dpx_probe(f.data, f.size, &info);
}

}  // namespace

extern "C" int rcgpu_encode(const rcgpu_job* job)
{
    clear_error();
    auto bail = [](int code) { fprintf(stderr, "Error: %s\n", rcgpu_last_error()); return code ? code : 1; };
    // RCGPU_TRACE=1: wall-clock of the job's phases on stderr (prefix "rcgpu trace:", never "Error:")
    const bool trace = getenv("RCGPU_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (trace) fprintf(stderr, "rcgpu trace: %8.3f s  %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), what);
    };
    if (!job || !job->streams || !job->n_streams || !job->output_path) return bail(fail(1, "job: missing streams or output path"));
    options opt;
    for (size_t i = 0; i + 1 < job->n_options; i += 2)
        if (job->options[i]) opt.kv[job->options[i]] = job->options[i + 1] ? job->options[i + 1] : "";
    // What this encoder implements of the option surface (defaults: CLI/Global.cpp:938-989)
    // `rawcooked -c:v ffv1_vulkan[:N]` is the reference's one GPU-selection surface: it puts `-init_hw_device "vulkan=vk:N" -vf hwupload -c:v
    // ffv1_vulkan` on the command line (CLI/Global.cpp:367-378; test/vulkan.sh:48-56 adds `,debug=0`).  Here that reads "ffv1 on device N":
    // the codec is this encoder either way, `hwupload` is what the pipeline does anyway, and N picks the HIP device -- unless the caller chose
    // devices itself (rcgpu_job::device_count, the shim's RCGPU_DEVICES), which stands above the command line.
    const char* cv = opt.get("c:v");
    if (cv && strcmp(cv, "ffv1") != 0 && strcmp(cv, "ffv1_vulkan") != 0) return bail(fail(2, "video codec %s is not supported by rcgpu (ffv1, or ffv1_vulkan = ffv1 on the device -init_hw_device names)", cv));
    int hw_device = -1;
    if (const char* hw = opt.get("init_hw_device")) {
        // FFmpeg's syntax: type[=name][:device[,key=value...]]; the reference only ever writes vulkan=vk:N
        const char* q = hw;
        if (strncmp(q, "vulkan", 6) != 0 || (q[6] && q[6] != '=' && q[6] != ':' && q[6] != ',')) return bail(fail(2, "-init_hw_device %s is not supported by rcgpu (vulkan[=name][:N]: HIP device N)", hw));
        q += 6;
        if (*q == '=') { q++; while (*q && *q != ':' && *q != ',') q++; }
        hw_device = 0;
        if (*q == ':') {
            char* end = nullptr;
            const long n = strtol(q + 1, &end, 10);
            if (end == q + 1 || n < 0 || n > 1023 || (*end && *end != ',')) return bail(fail(2, "-init_hw_device %s: the device is a number (vulkan=vk:N)", hw));
            hw_device = int(n);
        }
    } else if (cv && !strcmp(cv, "ffv1_vulkan"))
        hw_device = 0;                                                  // ffmpeg would pick the first Vulkan device
    // -c:a copy: what the reference asks for above 24 bits (CLI/Main.cpp:300-317) and what test/pcm.sh asks for by hand
    bool audio_copy = false;
    if (const char* ca = opt.get("c:a")) {
        audio_copy = !strcmp(ca, "copy");
        if (!audio_copy && strcmp(ca, "flac") != 0) return bail(fail(2, "audio codec %s is not supported by rcgpu (flac or copy)", ca));
    }
    const long coder = opt.num("coder", 1);
    if (coder != 1 && coder != 2) return bail(fail(2, "-coder %ld is not supported by rcgpu (1: range coder, 2: range coder with a transmitted state table)", coder));
    const long level = opt.num("level", 3);
    if (level != 1 && level != 3) return bail(fail(2, "-level %ld is not supported by rcgpu (3, or 1 with -slices 1)", level));
    if (opt.num("g", 1) != 1) return bail(fail(2, "-g %ld is not supported by rcgpu (intra only)", opt.num("g", 1)));
    // -f framemd5 (Output.cpp:312-332): a second output with FFmpeg's default stream choice -- one video stream and one audio stream,
    // the latter unless `-an` stands in front of it (--framemd5-an)
    const bool want_framemd5 = job->framemd5_path && *job->framemd5_path;
    // the only filter the reference ever asks for is `-vf vflip`, for DPX stored bottom-up (CLI/Main.cpp:207-211)
    // ... and `-vf hwupload` beside ffv1_vulkan (CLI/Global.cpp:374), which means nothing here: every frame is uploaded
    bool vflip_all = false;
    if (const char* vf = opt.get("vf")) {
        std::string rest = vf;
        while (!rest.empty()) {
            const size_t c = rest.find(',');
            const std::string one = rest.substr(0, c);
            rest = c == std::string::npos ? std::string() : rest.substr(c + 1);
            if (one == "vflip") vflip_all = true;
            else if (one != "hwupload") return bail(fail(2, "-vf %s is not supported by rcgpu (vflip, hwupload)", vf));
        }
    }
    uint32_t context = uint32_t(opt.num("context", 0));
    const uint32_t slicecrc = uint32_t(opt.num("slicecrc", 1));
    // -context 1 uses FFmpeg's level maps unless the compact 5-input model is asked for (the option rcgpu_context_model=compact; on the
    // shim's command line `-rcgpu_context_model compact`): same bitstream syntax, tables in the configuration record, states in LDS.
    // An option like every other that changes bytes -- never the environment.
    const char* model = opt.get("rcgpu_context_model");
    if (context == 1 && model && !strcmp(model, "compact")) context = 2;
    const bool overwrite = opt.has("y") && !opt.has("n");
    if (!overwrite && file_exists(job->output_path)) return bail(fail(3, "output file %s already exists (use -y)", job->output_path));

    // Every coded byte comes from the device; the only job that needs none is one that codes nothing (PCM tracks copied as they are)
    bool codes_something = !audio_copy;
    for (size_t si = 0; si < job->n_streams && !codes_something; si++) codes_something = job->streams[si].slices != 0;
    // `-rcgpu_plan_only 1` (an option like every other; the shim's command line or rcgpu_job::options): analyse the inputs -- sequences enumerated, files
    // probed, slice grids and frame rates settled, the flavor checks made -- print the plan and stop.  Needs no device and writes nothing: what a caller
    // runs to see what a job would do, and what tools/fuzz/fuzz_argv.cpp drives with hostile command lines and file lists.
    const bool plan_only = opt.num("rcgpu_plan_only", 0) != 0;
    int ndev_visible = plan_only ? 0 : rcgpu_device_count();
    if (ndev_visible <= 0 && codes_something && !plan_only) return bail(fail(4, "no HIP device available -- rcgpu has no CPU encode path"));
    const bool hw_chooses = hw_device >= 0 && job->device_count <= 0 && job->device_first <= 0;       // the caller named no devices: the command line does
    const int dev0 = hw_chooses ? hw_device : std::max(0, job->device_first);
    int ndev = hw_chooses ? 1 : job->device_count > 0 ? job->device_count : ndev_visible - dev0;
    if (codes_something && !plan_only) {
        if (dev0 >= ndev_visible || ndev <= 0) return bail(fail(4, "device selection %d+%d is outside the %d visible devices", dev0, job->device_count, ndev_visible));
        ndev = std::min(ndev, ndev_visible - dev0);
    } else
        ndev = 1;

    mark("options parsed, devices counted");
    // ---- analyse the streams
    std::vector<video_plan> videos; std::vector<audio_plan> audios;
    std::vector<std::pair<bool, size_t>> order;        // stream order -> (is_video, index)
    for (size_t si = 0; si < job->n_streams; si++) {
        const rcgpu_stream& s = job->streams[si];
        if (!s.path_or_template && !s.filelist) return bail(fail(5, "stream %zu has no input", si));
        std::vector<std::string> files;
        if (s.filelist && *s.filelist) {
            const char* p = s.filelist;
            while (*p) { const char* e = strchr(p, '\n'); std::string one = e ? std::string(p, e) : std::string(p); if (!one.empty()) files.push_back(one); if (!e) break; p = e + 1; }
        } else if (s.start_number && strchr(s.path_or_template, '%')) {
            unsigned long long n = strtoull(s.start_number, nullptr, 10);
            for (;; n++) { std::string f; if (!expand_template(s.path_or_template, n, f) || !file_exists(f)) break; files.push_back(f); }   // stop at the first gap, like image2
        } else
            files.push_back(s.path_or_template);
        if (files.empty()) return bail(fail(5, "stream %zu: no input file matches %s", si, s.path_or_template ? s.path_or_template : "(list)"));
        const std::string ext = lower_ext(files[0]);
        const bool is_audio = s.slices == 0 && (ext == "wav" || (s.flavor && !strncmp(s.flavor, "WAV/", 4)));
        if (is_audio) {
            audio_plan a; a.file = files[0];
            mapped_file f;
            if (!f.open(a.file)) return bail(fail(30, "cannot open %s: %s", a.file.c_str(), strerror(errno)));
            if (int r = rcgpu_wav_probe(f.data, f.size, &a.info)) return bail(r);
            audios.push_back(std::move(a)); order.push_back({ false, audios.size() - 1 });
        } else {
            video_plan v; v.files = std::move(files);
            if (int r = probe_image(v.files[0], v.tiff, v.info)) return bail(r);
            v.vflip = s.vflip || vflip_all;
            if (s.flavor && *s.flavor && strcmp(s.flavor, v.info.flavor) != 0)
                return bail(fail(6, "stream %zu: caller says flavor %s, file is %s", si, s.flavor, v.info.flavor));
            uint32_t slices = uint32_t(opt.num("slices", 0));
            if (!slices && s.slices && s.slices != 0xFFFFFFFFu) slices = s.slices;    // 0xFFFFFFFF: video, count left to the probe
            if (!slices) slices = v.info.slices;
            if (int r = rcgpu_slices_to_grid(slices, &v.num_h, &v.num_v)) return bail(r);
            v.fps = parse_framerate(s.framerate && *s.framerate ? s.framerate : nullptr);
            if ((!s.framerate || !*s.framerate) && v.info.framerate > 0) { char t[32]; snprintf(t, sizeof t, "%.6f", v.info.framerate); v.fps = parse_framerate(t); }
            videos.push_back(std::move(v)); order.push_back({ true, videos.size() - 1 });
        }
    }

    // FFmpeg's default stream choice for an output without -map: the video stream with the most pixels, the audio stream with the most
    // channels (first one on a tie) [ffmpeg-knowledge]; `-an` in front of it (--framemd5-an) drops the audio
    size_t md5_video = 0, md5_audio = size_t(-1);
    if (want_framemd5) {
        if (videos.empty()) return bail(fail(2, "-f framemd5 needs a video stream"));
        for (size_t vi = 1; vi < videos.size(); vi++)
            if (uint64_t(videos[vi].info.width) * videos[vi].info.height > uint64_t(videos[md5_video].info.width) * videos[md5_video].info.height) md5_video = vi;
        if (!opt.has("an"))
            for (size_t ai = 0; ai < audios.size(); ai++)
                if (md5_audio == size_t(-1) || audios[ai].info.channels > audios[md5_audio].info.channels) md5_audio = ai;
        if (md5_audio != size_t(-1) && (audios[md5_audio].info.format_tag != 1 || audios[md5_audio].info.bits_per_sample > 32))
            return bail(fail(2, "-f framemd5 of float audio is not supported by rcgpu, use --framemd5-an"));
        if (videos[md5_video].info.pixfmt == RCGPU_PIX_EXR_RGB16) return bail(fail(2, "-f framemd5 of EXR input is not supported by rcgpu"));
    }
    std::vector<uint8_t> framemd5_sums(want_framemd5 ? videos[md5_video].files.size() * 16 : 0);
    std::atomic<uint64_t> framemd5_frame_bytes{ 0 };
    mark("streams analysed");
    if (plan_only) {
        for (const auto& o : order) {
            if (o.first) {
                const video_plan& v = videos[o.second];
                printf("rcgpu plan: video %zu frames %ux%u %s slices %ux%u fps %u/%u%s first %s\n", v.files.size(), v.info.width, v.info.height, v.info.flavor, v.num_h, v.num_v,
                       v.fps.num, v.fps.den, v.vflip ? " vflip" : "", v.files[0].c_str());
            } else {
                const audio_plan& a = audios[o.second];
                printf("rcgpu plan: audio %s %u ch %u Hz %u bit%s\n", a.info.flavor, a.info.channels, a.info.sample_rate, a.info.bits_per_sample, audio_copy ? " (copied)" : " -> FLAC");
            }
        }
        if (hw_device >= 0) printf("rcgpu plan: device %d%s\n", dev0, hw_chooses ? " (-init_hw_device)" : " (the caller's choice stands above -init_hw_device)");
        printf("rcgpu plan: output %s%s, %zu attachment(s)%s\n", job->output_path, want_framemd5 ? " + framemd5" : "", job->n_attachments, job->reversibility_path ? " + reversibility data" : "");
        return 0;
    }
    // ---- audio first: A_FLAC CodecPrivate (STREAMINFO) must be final before the header is written
    for (audio_plan& a : audios) {
        if (audio_copy) {
            // blocks as FFmpeg's wav demuxer cuts them: at most 4096 bytes, whole sample frames (the reader concatenates them)
            a.pcm.reset(new mapped_file);
            if (!a.pcm->open(a.file)) return bail(fail(30, "cannot open %s", a.file.c_str()));
            a.block_size = std::max(1u, 4096u / a.info.block_align);
            const uint64_t nsamples = a.info.data_size / a.info.block_align;
            a.frame_sizes.assign(size_t(nsamples / a.block_size), a.block_size * a.info.block_align);
            if (nsamples % a.block_size) a.frame_sizes.push_back(uint32_t(nsamples % a.block_size) * a.info.block_align);
            continue;
        }
        if (a.info.bits_per_sample > 24 || a.info.format_tag != 1)
            return bail(fail(14, "FLAC encoding is not supported with %u-bit%s audio input, use -c:a copy", a.info.bits_per_sample, a.info.format_tag == 3 ? " float" : ""));   // wording: CLI/Main.cpp:311-313
        mapped_file f;
        if (!f.open(a.file)) return bail(fail(30, "cannot open %s", a.file.c_str()));
        rcgpu_flac_config fc{}; fc.channels = a.info.channels; fc.sample_rate = a.info.sample_rate; fc.bits_per_sample = a.info.bits_per_sample;
        fc.block_size = 0; fc.max_lpc_order = 8; fc.device = dev0;
        rcgpu_flac* fe = nullptr;
        if (int r = rcgpu_flac_create(&fc, &fe)) return bail(r);
        const uint64_t nsamples = a.info.data_size / a.info.block_align;
        a.frames.resize(size_t(a.info.data_size + a.info.data_size / 4 + (1 << 16)));
        a.frame_sizes.resize(size_t(nsamples / 192 + 16));
        uint32_t nframes = 0;
        int r = rcgpu_flac_encode_host(fe, f.data + a.info.data_offset, a.info.data_size, a.frames.data(), a.frames.size(), a.frame_sizes.data(),
                                       uint32_t(a.frame_sizes.size()), &nframes);
        if (!r) {
            a.frame_sizes.resize(nframes);
            a.codec_private.resize(64);
            a.codec_private.resize(rcgpu_flac_codec_private(fe, a.codec_private.data(), a.codec_private.size()));
            a.block_size = a.codec_private.size() >= 12 ? (uint32_t(a.codec_private[10]) << 8 | a.codec_private[11]) : 4608;   // STREAMINFO max blocksize
        }
        rcgpu_flac_destroy(fe);
        if (r) return bail(r);
    }

    mark("audio encoded");
    // ---- container header
    rcgpu_mkv* mux = nullptr;
    if (int r = rcgpu_mkv_open(job->output_path, 1, &mux)) return bail(r);
    struct mux_guard { rcgpu_mkv*& m; const char* path; bool ok = false; ~mux_guard() { if (m) { rcgpu_mkv_close(m); if (!ok) unlink(path); } } } guard{ mux, job->output_path };

    // ---- the device side: one pipeline for all picture sequences of the job (pipeline.h): encoders per device, batches sized from the
    // device's free memory, pinned upload slots and download rings
    std::unique_ptr<rc::pipeline> plp(new rc::pipeline);
    rc::pipeline& pl = *plp;
    std::vector<rc::pipe_video> pvideos;
    for (video_plan& v : videos) {
        rc::pipe_video pv; pv.frames = v.files.size();
        rcgpu_ffv1_config& c = pv.cfg;
        c.width = v.info.width; c.height = v.info.height; c.pixfmt = v.info.pixfmt; c.line_bytes = v.info.line_bytes;
        c.flags = (v.info.flags & RCGPU_FLAG_ALTERN) | (v.vflip ? RCGPU_FLAG_VFLIP : 0);
        // `-rcgpu_own_slice_buffers 1`: for content that the overflow report names (a slice that codes to more than 4 bytes per sample early on)
        if (opt.num("rcgpu_own_slice_buffers", 0)) c.flags |= RCGPU_FLAG_OWN_SLICE_BUFFERS;
        c.num_h_slices = v.num_h; c.num_v_slices = v.num_v; c.slicecrc = slicecrc; c.context = context; c.coder = uint32_t(coder); c.level = uint32_t(level);
        if (level == 1) { if (v.num_h * v.num_v != 1) return bail(fail(2, "-level 1 (FFV1 version 1) has no slices: use -slices 1")); c.slicecrc = 0; }
        pvideos.push_back(pv);
    }
    for (auto& o : order) {
        if (o.first) {
            video_plan& v = videos[o.second];
            // the record depends on the configuration only: the header is written, and the file laid out, before the encoders exist
            const std::vector<uint8_t> rec = rc::ffv1_config_record_for(pvideos[o.second].cfg);
            v.track = rcgpu_mkv_add_video(mux, rec.data(), rec.size(), v.info.width, v.info.height, v.fps.num, v.fps.den);
            if (v.track < 0) return bail(8);
            if (const char* md = opt.get("metadata:s:v")) {
                const std::string kv = md; const size_t eq = kv.find('=');
                if (eq != std::string::npos && eq > 0)
                    if (int r = rcgpu_mkv_add_tag(mux, v.track, kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str())) return bail(r);
            }
        } else {
            audio_plan& a = audios[o.second];
            a.track = audio_copy ? rcgpu_mkv_add_audio_pcm(mux, a.info.format_tag == 3, a.info.channels, a.info.sample_rate, a.info.bits_per_sample)
                                 : rcgpu_mkv_add_audio(mux, a.codec_private.data(), a.codec_private.size(), a.info.channels, a.info.sample_rate, a.info.bits_per_sample);
            if (a.track < 0) return bail(8);
        }
    }
    auto attach = [&](const char* path, const char* name) -> int {
        mapped_file f;
        if (!f.open(path)) return fail(30, "cannot open attachment %s: %s", path, strerror(errno));
        return rcgpu_mkv_add_attachment(mux, name, "application/octet-stream", f.data, f.size);
    };
    for (size_t i = 0; i < job->n_attachments; i++)
        if (int r = attach(job->attachments[i].path_in, job->attachments[i].name_out ? job->attachments[i].name_out : job->attachments[i].path_in)) return bail(r);
    if (job->reversibility_path)
        if (int r = attach(job->reversibility_path, "RAWcooked reversibility data")) return bail(r);
    if (int r = rcgpu_mkv_begin(mux)) return bail(r);
    mark("header written");
    uint64_t max_bytes = 0, n_blocks = 0;
    for (size_t vi = 0; vi < videos.size(); vi++) {
        max_bytes += uint64_t(videos[vi].files.size()) * rc::ffv1_max_packet_bytes_for(pvideos[vi].cfg);
        n_blocks += videos[vi].files.size();
    }
    for (const audio_plan& a : audios) for (uint32_t fs : a.frame_sizes) max_bytes += fs + 32;
    // on tmpfs the muxer now starts allocating the file's pages -- while the encoders (170 GB of device buffers: seconds) come up
    if (!videos.empty()) if (int r = rcgpu_mkv_expect(mux, max_bytes, n_blocks)) return bail(r);
    if (!videos.empty()) {
        rc::pipe_options po;
        po.device_first = dev0; po.device_count = ndev; po.trace = trace;
        po.batch = uint32_t(std::max(0L, opt.num("rcgpu_batch", 0)));      // 0: sized from the device's free memory and the sequence
        if (const char* e = getenv("RCGPU_BATCH")) if (!po.batch) po.batch = uint32_t(std::max(0, atoi(e)));
        if (const char* e = getenv("RCGPU_LANES")) po.lanes_per_device = uint32_t(std::max(1, atoi(e)));
        // files come through pread() out of the page cache or tmpfs at ~2 GB/s per thread (a memcpy between user buffers does 14): sixteen
        // readers per device instead of the pipeline's eight (measured on 1000 4K frames: all files read after 2.6 s instead of 3.2-4.2 s)
        po.readers = std::max(2u, std::min(std::max(2u, std::thread::hardware_concurrency()) / 2, 16u * unsigned(std::max(1, ndev))));
        if (const char* e = getenv("RCGPU_READERS")) po.readers = uint32_t(std::max(1, atoi(e)));
        if (const char* e = getenv("RCGPU_WRITERS")) po.writers = uint32_t(std::max(1, atoi(e)));
        if (int r = pl.prepare(pvideos, po)) return bail(r);
        for (size_t vi = 0; vi < videos.size(); vi++) {          // what was promised to the container is what the encoders do
            uint8_t rec[4096];
            const size_t n = rcgpu_ffv1_config_record(pl.encoder(uint32_t(vi)), rec, sizeof rec);
            const std::vector<uint8_t> want = rc::ffv1_config_record_for(pvideos[vi].cfg);
            if (n != want.size() || memcmp(rec, want.data(), n) != 0 || rcgpu_ffv1_max_packet_bytes(pl.encoder(uint32_t(vi))) > rc::ffv1_max_packet_bytes_for(pvideos[vi].cfg))
                return bail(fail(100, "internal: the encoder's configuration record differs from the one in the container header"));
        }
    }
    mark("encoders created");
    // ---- blocks, in timestamp order: audio frames are interleaved in front of the video frame they precede
    std::vector<size_t> audio_pos(audios.size(), 0), audio_off(audios.size(), 0);
    auto write_audio_until = [&](uint64_t pts_ns_limit) -> int {
        for (size_t ai = 0; ai < audios.size(); ai++) {
            audio_plan& a = audios[ai];
            while (audio_pos[ai] < a.frame_sizes.size()) {
                const uint64_t pts = uint64_t(audio_pos[ai]) * a.block_size * 1000000000ull / a.info.sample_rate;
                if (pts > pts_ns_limit) break;
                if (int r = rcgpu_mkv_write_block(mux, a.track, pts, a.bytes() + audio_off[ai], a.frame_sizes[audio_pos[ai]], 1)) return r;
                audio_off[ai] += a.frame_sizes[audio_pos[ai]++];
            }
        }
        return 0;
    };

    if (!videos.empty()) {
        // frames of all picture sequences in timestamp order, so that tracks interleave in the file; the pipeline cuts them into
        // batches, shards the batches over the devices and returns the packets in this order
        std::vector<rc::pipe_frame> frames;
        for (size_t vi = 0; vi < videos.size(); vi++)
            for (size_t i = 0; i < videos[vi].files.size(); i++) frames.push_back({ uint32_t(vi), i });
        auto pts_of = [&](const rc::pipe_frame& f) { const video_plan& v = videos[f.video]; return uint64_t(f.index) * v.fps.den * 1000000000ull / v.fps.num; };
        if (videos.size() > 1) std::stable_sort(frames.begin(), frames.end(), [&](const rc::pipe_frame& a, const rc::pipe_frame& b) { return pts_of(a) < pts_of(b); });
        std::vector<std::vector<uint64_t>> block_off(videos.size());
        std::vector<std::vector<uint8_t*>> block_dst(videos.size());
        for (size_t vi = 0; vi < videos.size(); vi++) { block_off[vi].assign(videos[vi].files.size(), 0); block_dst[vi].assign(videos[vi].files.size(), nullptr); }
        rc::pipe_io io;
        io.read = [&](const rc::pipe_frame& f, uint8_t* dst) -> int {
            const video_plan& v = videos[f.video];
            const std::string& path = v.files[size_t(f.index)];
            mapped_file m;
            if (!m.open(path, true)) return fail(30, "cannot open %s: %s", path.c_str(), strerror(errno));
            rcgpu_image_info fi{};
            const int r = v.info.pixfmt == RCGPU_PIX_EXR_RGB16 ? rcgpu_exr_probe(m.data, m.size, &fi)
                        : v.tiff ? rcgpu_tiff_probe(m.data, m.size, &fi) : rcgpu_dpx_probe(m.data, m.size, &fi);
            if (r) return r;
            if (fi.width != v.info.width || fi.height != v.info.height || fi.pixfmt != v.info.pixfmt || fi.line_bytes != v.info.line_bytes || fi.flags != v.info.flags)
                return fail(31, "%s differs in geometry/flavor from the first frame of the sequence", path.c_str());
            if (!m.read_at(fi.data_offset, dst, size_t(fi.data_size))) return fail(30, "cannot read %s: %s", path.c_str(), strerror(errno));
            return 0;
        };
        io.place = [&](const rc::pipe_frame& f, size_t size) -> uint8_t* {
            clear_error();
            const video_plan& v = videos[f.video];
            const uint64_t pts = pts_of(f);
            if (write_audio_until(pts)) return nullptr;
            uint8_t* dst = nullptr; uint64_t off = 0;
            if (rcgpu_mkv_reserve_block(mux, v.track, pts, size, 1, &dst, &off)) return nullptr;
            block_off[f.video][size_t(f.index)] = off; block_dst[f.video][size_t(f.index)] = dst;
            return dst;
        };
        io.copy = [&](uint8_t* dst, const uint8_t* src, size_t size) -> int { return rcgpu_mkv_copy_in(mux, dst, src, size); };
        io.done = [&](const rc::pipe_frame& f, const uint8_t* data, size_t size) -> int {
            if (block_dst[f.video][size_t(f.index)]) return 0;                   // a writer thread copied it into the mapped file
            return rcgpu_mkv_fill(mux, block_off[f.video][size_t(f.index)], data, size);
        };
        if (want_framemd5)
            io.after_batch = [&](uint32_t video, rcgpu_ffv1* enc, uint64_t first, uint32_t n) -> int {
                if (video != md5_video) return 0;
                uint64_t fb = 0;
                if (int r = rcgpu_ffv1_framemd5_last(enc, n, framemd5_sums.data() + size_t(first) * 16, &fb)) return r;
                framemd5_frame_bytes = fb;
                return 0;
            };
        rc::pipe_stats ps;
        if (int r = pl.run(frames, io, &ps)) return bail(r);
        if (trace) {
            char b[512];
            snprintf(b, sizeof b, "pipeline: %llu frames in %.3f s (%.1f frames/s; %.1f between the first and the last batch), prepare %.3f s, all files read after %.3f s, "
                     "last batch coded after %.3f s, first packet after %.3f s, batches of %u on %u lane(s), %u readers, %u writers",
                     (unsigned long long)ps.frames, ps.seconds, double(ps.frames) / std::max(ps.seconds, 1e-9), ps.steady_frames_per_second, ps.prepare_seconds, ps.reads_done_seconds,
                     ps.last_batch_seconds, ps.first_packet_seconds, ps.batch_frames, ps.lanes, ps.readers, ps.writers);
            mark(b);
        }
    }
    mark("all batches encoded and written");
    if (int r = write_audio_until(~0ull)) return bail(r);
    if (want_framemd5) {
        // libavformat's framehash layout, version 2 [ffmpeg-knowledge]: header, then one line per packet, streams interleaved by time.
        // Audio: the framemd5 muxer's default audio codec is pcm_s16le, its packets are the WAV demuxer's (at most 4096 bytes of whole
        // sample frames), samples converted the way libswresample does without dither (24/32 bit: arithmetic shift, 8 bit: offset binary
        // to signed, << 8).
        const video_plan& v = videos[md5_video];
        FILE* fh = fopen(job->framemd5_path, "w");
        if (!fh) return bail(fail(30, "cannot create %s: %s", job->framemd5_path, strerror(errno)));
        fprintf(fh, "#format: frame checksums\n#version: 2\n#hash: MD5\n#software: %s\n", rcgpu_version());
        fprintf(fh, "#tb 0: %u/%u\n#media_type 0: video\n#codec_id 0: rawvideo\n#dimensions 0: %ux%u\n#sar 0: 0/1\n", v.fps.den, v.fps.num, v.info.width, v.info.height);
        mapped_file wav;
        const audio_plan* a = md5_audio == size_t(-1) ? nullptr : &audios[md5_audio];
        uint64_t a_samples = 0, a_pos = 0; uint32_t a_per_packet = 0;
        if (a) {
            if (!wav.open(a->file)) { fclose(fh); return bail(fail(30, "cannot open %s", a->file.c_str())); }
            const uint32_t ch = a->info.channels;
            const char* layout = ch == 1 ? "mono" : ch == 2 ? "stereo" : ch == 4 ? "quad" : ch == 6 ? "5.1" : ch == 8 ? "7.1" : nullptr;
            fprintf(fh, "#tb 1: 1/%u\n#media_type 1: audio\n#codec_id 1: pcm_s16le\n#sample_rate 1: %u\n", a->info.sample_rate, a->info.sample_rate);
            if (layout) fprintf(fh, "#channel_layout_name 1: %s\n", layout); else fprintf(fh, "#channel_layout_name 1: %u channels\n", ch);
            a_samples = a->info.data_size / a->info.block_align;
            a_per_packet = std::max(1u, 4096u / a->info.block_align);
        }
        fprintf(fh, "#stream#, dts,        pts, duration,     size, hash\n");
        auto audio_row = [&]() {
            const uint32_t n = uint32_t(std::min<uint64_t>(a_per_packet, a_samples - a_pos)), ch = a->info.channels, bps = a->info.bits_per_sample / 8;
            std::vector<uint8_t> s16(size_t(n) * ch * 2);
            const uint8_t* src = wav.data + a->info.data_offset + a_pos * a->info.block_align;
            for (size_t k = 0; k < size_t(n) * ch; k++) {
                int32_t x;
                if (bps == 1) x = (int32_t(src[k]) - 128) << 8;
                else { uint32_t u = 0; for (uint32_t b = 0; b < bps; b++) u |= uint32_t(src[k * bps + b]) << (8 * (b + 4 - bps)); x = int32_t(u) >> 16; }      // left-justified in 32 bits, then >> 16
                s16[2 * k] = uint8_t(x); s16[2 * k + 1] = uint8_t(x >> 8);
            }
            uint8_t md[16]; rcgpu_md5(s16.data(), s16.size(), md);
            fprintf(fh, "1, %10llu, %10llu, %8u, %8zu, ", (unsigned long long)a_pos, (unsigned long long)a_pos, n, s16.size());
            for (int k = 0; k < 16; k++) fprintf(fh, "%02x", md[k]);
            fputc('\n', fh);
            a_pos += n;
        };
        for (size_t i = 0; i < v.files.size(); i++) {
            // audio packets that start before this frame (a tie goes to the video frame)
            while (a && a_pos < a_samples && a_pos * uint64_t(v.fps.num) < uint64_t(i) * v.fps.den * a->info.sample_rate) audio_row();
            fprintf(fh, "0, %10llu, %10llu, %8d, %8llu, ", (unsigned long long)i, (unsigned long long)i, 1, (unsigned long long)framemd5_frame_bytes.load());
            for (int k = 0; k < 16; k++) fprintf(fh, "%02x", framemd5_sums[i * 16 + size_t(k)]);
            fputc('\n', fh);
        }
        while (a && a_pos < a_samples) audio_row();
        if (fclose(fh)) return bail(fail(30, "cannot write %s", job->framemd5_path));
    }
    // closing the file (unmapping ~50 GB of it: seconds) and giving back the device and pinned memory (seconds, too) side by side
    // (the shim's process ends here: it leaves both to the kernel, shim_main.cpp)
    const char* at_exit = getenv("RCGPU_RELEASE_AT_EXIT");
    const bool leave = at_exit && *at_exit == '1';
    std::thread release([&] { if (leave) (void)plp.release(); else plp.reset(); });
    rcgpu_mkv* m = mux; mux = nullptr;
    const int rc_close = rcgpu_mkv_close(m);
    mark("file closed");
    release.join();
    mark(leave ? "device and pinned memory left to process exit" : "device and pinned memory released");
    if (rc_close) { unlink(job->output_path); return bail(rc_close); }
    guard.ok = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// argv front end: the grammar of output::FFmpeg_Command (CLI/Output.cpp:81-332)
// ---------------------------------------------------------------------------------------------------------
extern "C" int rcgpu_main_ffmpeg_argv(int argc, const char* const* argv)
{
    clear_error();
    struct in_stream { std::string path, start, framerate, filelist, fmt; bool is_video_hint = false; };
    std::vector<in_stream> ins; in_stream cur;
    std::vector<std::pair<std::string, std::string>> attach_files;   // path, display name
    std::map<std::string, std::string> out_opts;
    std::string output, rev_path, framemd5;
    std::string pending_f;
    auto need = [&](int i) { return i + 1 < argc; };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-version" || a == "--version") { printf("%s\n", rcgpu_version()); return 0; }
        if (a == "-xerror" || a == "-nostdin" || a == "-hide_banner") continue;
        if (a == "-y" || a == "-n") { out_opts[a.substr(1)] = ""; continue; }
        if (a == "-i" && need(i)) {
            cur.path = argv[++i]; cur.fmt = pending_f; pending_f.clear();
            if (cur.fmt == "concat") {     // ffconcat list written at Output.cpp:236-246: file '<path>' / duration x
                mapped_file f;
                if (!f.open(cur.path)) { fprintf(stderr, "Error: cannot open file list %s\n", cur.path.c_str()); return 1; }
                std::string text(reinterpret_cast<const char*>(f.data), f.size), list;
                size_t p = 0;
                while (p < text.size()) {
                    size_t e = text.find('\n', p); if (e == std::string::npos) e = text.size();
                    std::string line = text.substr(p, e - p); p = e + 1;
                    while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();     // a list that went through a CR LF system
                    if (line.compare(0, 6, "file '") == 0 && line.size() > 7 && line.back() == '\'') { if (!list.empty()) list += '\n'; list += line.substr(6, line.size() - 7); }
                    // at exactly 25 frames per second the reference skips its own rewriting and the list holds bare paths (Output.cpp:162-163)
                    else if (!line.empty() && line.compare(0, 9, "duration ") != 0 && line.compare(0, 5, "file ") != 0 && line[0] != '#' && line.compare(0, 8, "ffconcat") != 0)
                        { if (!list.empty()) list += '\n'; list += line; }
                }
                cur.filelist = list; cur.path.clear();
            }
            ins.push_back(cur); cur = in_stream(); continue;
        }
        if (a == "-f" && need(i)) {
            const std::string v = argv[++i];
            if (v == "matroska") { if (need(i) && argv[i + 1][0] != '-') output = argv[++i]; continue; }
            if (v == "framemd5") { if (need(i)) framemd5 = argv[++i]; continue; }
            pending_f = v; continue;
        }
        if (a == "-attach" && need(i)) { attach_files.push_back({ argv[++i], "" }); continue; }
        if (a.compare(0, 12, "-metadata:s:") == 0 && need(i)) {
            const std::string kv = argv[++i];
            if (kv.compare(0, 9, "filename=") == 0 && !attach_files.empty()) attach_files.back().second = kv.substr(9);
            else if (a == "-metadata:s:v") out_opts["metadata:s:v"] = kv;          // e.g. the reference's WARNING=... on EXR packages
            continue;
        }
        if (a == "-map" && need(i)) { i++; continue; }
        if (a == "-an") { out_opts["an"] = ""; continue; }          // only ever in front of `-f framemd5` (Output.cpp:326-329)
        if (a[0] == '-' && need(i)) {
            const std::string k = a.substr(1), v = argv[++i];
            // options in front of an -i belong to that input (Output.cpp:111-131); the rest are output options
            if (k == "framerate") { cur.framerate = v; cur.is_video_hint = true; continue; }
            if (k == "r") { if (cur.framerate.empty()) cur.framerate = v; continue; }
            if (k == "start_number") { cur.start = v; continue; }
            if (k == "safe" || k == "consider_float16_as_uint16") continue;
            if (k == "c:v" && (v == "dpx" || v == "tiff" || v == "exr")) { cur.is_video_hint = true; continue; }
            out_opts[k] = v;
            continue;
        }
        if (a[0] != '-') { output = a; continue; }
        fprintf(stderr, "Error: rcgpu-ffmpeg does not understand argument %s\n", a.c_str());
        return 1;
    }
    if (ins.empty() || output.empty()) { fprintf(stderr, "Error: rcgpu-ffmpeg needs at least one -i and an output (-f matroska <file>)\n"); return 1; }
    // the reversibility file is the attachment displayed as "RAWcooked reversibility data" (Output.cpp:289-291)
    std::vector<rcgpu_attachment> atts;
    for (auto& af : attach_files) {
        if (af.second == "RAWcooked reversibility data") rev_path = af.first;
        else atts.push_back({ af.first.c_str(), af.second.empty() ? af.first.c_str() : af.second.c_str() });
    }
    std::vector<rcgpu_stream> streams;
    for (in_stream& s : ins) {
        rcgpu_stream r{};
        r.path_or_template = s.path.empty() ? nullptr : s.path.c_str();
        r.start_number = s.start.empty() ? nullptr : s.start.c_str();
        r.filelist = s.filelist.empty() ? nullptr : s.filelist.c_str();
        r.framerate = s.framerate.empty() ? nullptr : s.framerate.c_str();
        const std::string ext = lower_ext(s.path);
        r.slices = (ext == "wav") ? 0 : 0xFFFFFFFFu;      // 0 = audio; 0xFFFFFFFF = video, count from -slices or the probe
        streams.push_back(r);
    }
    std::vector<const char*> kv;
    for (auto& o : out_opts) { kv.push_back(o.first.c_str()); kv.push_back(o.second.c_str()); }
    rcgpu_job job{};
    job.streams = streams.data(); job.n_streams = streams.size();
    job.attachments = atts.data(); job.n_attachments = atts.size();
    job.reversibility_path = rev_path.empty() ? nullptr : rev_path.c_str();
    job.output_path = output.c_str();
    job.framemd5_path = framemd5.empty() ? nullptr : framemd5.c_str();
    job.options = kv.data(); job.n_options = kv.size();
    const char* dv = getenv("RCGPU_DEVICES");     // "first,count"; default: all visible
    if (dv) { int a = 0, b = 0; if (sscanf(dv, "%d,%d", &a, &b) >= 1) { job.device_first = a; job.device_count = b; } }
    return rcgpu_encode(&job);
}
