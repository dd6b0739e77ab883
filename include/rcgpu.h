/* rcgpu.h -- C ABI of the MI355X-native FFV1 + FLAC encode path for RAWcooked.
 *
 * This library replaces the one hot path of MediaArea/RAWcooked that the reference delegates to an
 * external `ffmpeg` process: `int Value = system(Command.c_str());` at Source/CLI/Output.cpp:356
 * (command assembled at Output.cpp:81-310).  Everything here is extern "C", plain pointers and sizes.
 * See INTEGRATION.md for the patch a RAWcooked maintainer would apply, and for the argv-compatible
 * shim (`rcgpu-ffmpeg`) that works with an unmodified rawcooked through `--bin-name`.
 *
 * All `file:line` citations are relative to /root/reference/Source.
 *
 * Error convention (all functions returning int): 0 = ok, >0 = error; rcgpu_last_error() returns a
 * thread-local, NUL-terminated description.  Nothing throws across this boundary.
 */
#ifndef RCGPU_H
#define RCGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ===========================================================================================
 * 0. Library
 * ======================================================================================== */
const char* rcgpu_version(void);       /* "rcgpu <x.y> ..."; the shim prints it for `-version` (CLI/Main.cpp:751-774) */
const char* rcgpu_last_error(void);    /* thread-local static storage */
int         rcgpu_device_count(void);  /* number of visible HIP devices (0 when there is none) */

/* ===========================================================================================
 * 1. Job level -- replaces output::FFmpeg_Command's system() call (CLI/Output.cpp:36-378)
 * ======================================================================================== */
typedef struct {
    const char* path_or_template;   /* stream::FileName_Template or FileName (CLI/Output.h:23-24), printf-style %0Nd */
    const char* start_number;       /* stream::FileName_StartNumber (Output.h:25); NULL for a single file */
    const char* filelist;           /* stream::FileList, '\n' separated paths (Output.h:27); NULL if none */
    const char* flavor;             /* stream::Flavor, e.g. "DPX/Raw/RGB/16bit/U/BE" (Lib/Common/Common.cpp:123-157); may be NULL: probed */
    const char* framerate;          /* decimal or "n/d" (Output.cpp:58-79,149-160) */
    uint32_t    slices;             /* stream::Slices (slice_x*slice_y) or user -slices; 0 = audio */
    int         vflip;              /* `-vf vflip` (CLI/Main.cpp:207-211) */
} rcgpu_stream;

typedef struct {
    const char* path_in;            /* attachment::FileName_In  (Output.h:36) */
    const char* name_out;           /* attachment::FileName_Out (Output.h:37) */
} rcgpu_attachment;

typedef struct {
    const rcgpu_stream*     streams;      size_t n_streams;
    const rcgpu_attachment* attachments;  size_t n_attachments;
    const char* reversibility_path;       /* Global.rawcooked_reversibility_FileName; NULL when IgnoreReversibilityFile (Output.cpp:291) */
    const char* output_path;              /* Global.OutputFileName (Output.cpp:292-305) */
    const char* framemd5_path;            /* optional (Output.cpp:312-332); NULL = none */
    const char* const* options;           /* key,value,key,value... = Global.OutputOptions (Output.cpp:273-278):           */
    size_t      n_options;                /*   coder, context, g, level, slicecrc, slices, threads, c:a, c:v, y, n, loglevel, */
                                          /*   and the reference's GPU selection (CLI/Global.cpp:367-378): c:v = ffv1_vulkan,  */
                                          /*   init_hw_device = vulkan=vk:N, vf = hwupload -> ffv1 on HIP device N             */
    int device_first;                     /* first HIP device to use */
    int device_count;                     /* number of devices (frames shard i mod device_count); 0 = all visible -- or, with  */
                                          /*   device_first 0 and an init_hw_device option, the one device that option names   */
} rcgpu_job;

/* 0 ok; >0 error (becomes the process exit code the reference propagates, Output.cpp:356-374).
 * Human-readable text goes to stderr prefixed "Error: " (Project/GNU/CLI/test/helpers.sh:81). */
int rcgpu_encode(const rcgpu_job* job);

/* The reference's argv grammar (Output.cpp:81-332) -> job -> rcgpu_encode.  This is what the shim's main() calls. */
int rcgpu_main_ffmpeg_argv(int argc, const char* const* argv);

/* ===========================================================================================
 * 2. Uncompressed-format probes -- the subset of dpx::ParseBuffer (Lib/Uncompressed/DPX/DPX.cpp:250-634),
 *    tiff::ParseBuffer (TIFF/TIFF.cpp:380-717) and wav::ParseBuffer (WAV/WAV.cpp:271-542) the encoder
 *    side needs: payload offset, geometry, flavor, slice count.
 * ======================================================================================== */
enum {   /* pixel layouts; names follow the reference flavors (DPX.cpp:184-231, TIFF.cpp:157-173) */
    RCGPU_PIX_RGB8 = 0,
    RCGPU_PIX_RGB10_FILLEDA_BE = 1, RCGPU_PIX_RGB10_FILLEDA_LE = 2,
    RCGPU_PIX_RGB12_FILLEDA_BE = 3, RCGPU_PIX_RGB12_FILLEDA_LE = 4,
    RCGPU_PIX_RGB16_BE = 5, RCGPU_PIX_RGB16_LE = 6,
    RCGPU_PIX_RGBA8 = 7, RCGPU_PIX_RGBA16_BE = 8, RCGPU_PIX_RGBA16_LE = 9,
    RCGPU_PIX_Y8 = 10, RCGPU_PIX_Y16_BE = 11, RCGPU_PIX_Y16_LE = 12,
    /* bit-packed DPX flavors (DPX.cpp:189,194-198,202-204; packers Transform.cpp:161-322,445-600,709-990) */
    RCGPU_PIX_RGB12_PACKED_BE = 13,
    RCGPU_PIX_RGBA10_FILLEDA_BE = 14, RCGPU_PIX_RGBA10_FILLEDA_LE = 15,
    RCGPU_PIX_RGBA12_PACKED_BE = 16, RCGPU_PIX_RGBA12_FILLEDA_BE = 17, RCGPU_PIX_RGBA12_FILLEDA_LE = 18,
    RCGPU_PIX_Y10_FILLEDA_BE = 19, RCGPU_PIX_Y10_FILLEDB_BE = 20, RCGPU_PIX_Y12_PACKED_BE = 21,
    /* OpenEXR scan lines, three HALF channels taken as uint16 (the reference runs FFmpeg with -consider_float16_as_uint16 1,
     * Output.cpp:120-122): every line is [y:u32][bytes:u32][B x width][G x width][R x width], little endian
     * (Transform.cpp:1062-1127, EXR.cpp:601-606) */
    RCGPU_PIX_EXR_RGB16 = 22,
    RCGPU_PIX_COUNT
};
/* payload layout variants the DPX header announces; the FFV1 bitstream does not know about them */
#define RCGPU_FLAG_VFLIP  1u   /* orientation 2: lines stored bottom to top; the reference then adds "-vf vflip" (Main.cpp:207-211),
                                  picture line y = file line height-1-y (Transform.cpp:181-185).  12-bit Packed flavors only (DPX.cpp:189,204) */
#define RCGPU_FLAG_ALTERN 2u   /* Y 10-bit from some scanners: words are filled across line ends, no line padding (DPX.cpp:363-368,465-469) */
/* encoder only (rcgpu_ffv1_config::flags), nothing a header announces: */
#define RCGPU_FLAG_OWN_SLICE_BUFFERS 0x100u   /* Where every slice holds >= 64 K samples the encoder keeps a slice's coded bytes in the slice's own area of the
                                  symbol buffer -- the coder writes at most 3.4 bytes where a 4-byte symbol lay that k_resolve has read already -- and
                                  allocates no slice byte buffers (96 MB per 4K frame in flight, twice that in run-on mode).  A slice whose FIRST
                                  segments code to more than 4 bytes per sample (16-bit noise with untrained states: 3.4) is then reported as
                                  overflowing although it may fit 1.5 x raw in the end.  This flag gives the slices buffers of their own again. */

typedef struct {
    uint32_t width, height;
    uint32_t pixfmt;            /* RCGPU_PIX_* */
    uint32_t bits_per_sample;
    uint64_t data_offset;       /* OffsetToImageData / StripOffsets[0] */
    uint64_t data_size;         /* line_bytes * height */
    uint32_t line_bytes;        /* DPX: padded to 32 bit (Utils/RawFrame/RawFrame.cpp:109); TIFF: unpadded */
    uint32_t slices;            /* slice_x*slice_y the reference would pass as -slices (DPX.cpp:428-458, TIFF.cpp:657-672) */
    double   framerate;         /* DPX only (DPX.cpp:370-387); 0 when absent */
    char     flavor[64];        /* "DPX/Raw/RGB/16bit/U/BE", "DPX/Raw/RGB/10bit/U/BE/FilledA" ... (DPX.cpp:762-778, Common.cpp:123-139) */
    uint32_t flags;             /* RCGPU_FLAG_* found in the header */
} rcgpu_image_info;

typedef struct {
    uint32_t channels, sample_rate, bits_per_sample;
    uint32_t block_align;
    uint64_t data_offset, data_size;
    char     flavor[64];        /* "WAV/PCM/48kHz/24bit/6ch/S/LE" */
    uint32_t format_tag;        /* 1 integer PCM, 3 IEEE float (WAV.cpp:205-221); bits > 24 or float: only `-c:a copy` (CLI/Main.cpp:300-317) */
} rcgpu_audio_info;

int rcgpu_dpx_probe (const uint8_t* file, size_t size, rcgpu_image_info* out);
int rcgpu_tiff_probe(const uint8_t* file, size_t size, rcgpu_image_info* out);
/* exr::ParseBuffer (Lib/Uncompressed/EXR/EXR.cpp:199-633): version 2 single-part scan-line files, channels B,G,R of type HALF,
 * no compression, increasing-Y line order, data window == display window at the origin. */
int rcgpu_exr_probe (const uint8_t* file, size_t size, rcgpu_image_info* out);
int rcgpu_wav_probe (const uint8_t* file, size_t size, rcgpu_audio_info* out);

/* The reference's flavor string (stream::Flavor, CLI/Output.h:28; DPX_Flavor_String DPX.cpp:762-778, TIFF.cpp:744-750, EXR.cpp:660-666)
 * -> RCGPU_PIX_*: what a decode-side binding has at hand (raw_frame::Flavor / Flavor_Private restored from the reversibility data). */
int rcgpu_pixfmt_from_flavor(const char* flavor, uint32_t* pixfmt);
/* slice_x*slice_y the reference computes for a DPX/TIFF picture (DPX.cpp:428-458, TIFF.cpp:657-672): pixels_per_block > 1
 * models the flavors whose slices must start on a block boundary (e.g. 3 for DPX RGBA 10-bit FilledA); 0 = unsupported. */
uint32_t rcgpu_reference_slices(uint32_t width, uint32_t height, uint32_t bitdepth, uint32_t pixels_per_block);
/* -slices N -> h x v the way FFmpeg's ffv1 encoder factorises it (accepted set == test/slices.sh:12). 0 ok. */
int rcgpu_slices_to_grid(uint32_t n, uint32_t* num_h, uint32_t* num_v);

/* ===========================================================================================
 * 3. FFV1 encoder (device) -- replaces FFmpeg's ffv1enc; inverse of the in-tree decoder
 *    ffv1_frame::{OutOfBand,Process} (Lib/CoDec/FFV1/FFV1_Frame.cpp:105-228),
 *    slice::{Parse,SliceHeader,Line} (FFV1_Slice.cpp:113-472), rangecoder (FFV1_RangeCoder.cpp:71-305),
 *    and of the packers in Lib/Transform/Transform.cpp.
 * ======================================================================================== */
typedef struct {
    uint32_t width, height;
    uint32_t pixfmt;          /* RCGPU_PIX_* */
    uint32_t line_bytes;      /* bytes between payload lines */
    uint32_t num_h_slices, num_v_slices;   /* num_h >= num_v (FFV1_Slice.cpp:127) */
    uint32_t slicecrc;        /* -slicecrc (ec) */
    uint32_t context;         /* -context 0|1 with FFmpeg's level maps (365 / 5063 contexts for > 8 bit); 2 = 5-input model with
                                 compact level maps (338 contexts): adaptive states stay in LDS */
    uint32_t max_batch;       /* frames encoded per call (frames in flight on the device) */
    int      device;          /* HIP device ordinal */
    uint32_t segments;        /* hand-over granularity between state resolution and range coding: each slice's decision
                                 stream is produced/consumed in this many windows (0 = automatic, 1 = whole slice) */
    uint32_t flags;           /* RCGPU_FLAG_VFLIP | RCGPU_FLAG_ALTERN: how the payload is laid out (line_bytes is ignored for ALTERN); RCGPU_FLAG_OWN_SLICE_BUFFERS */
    uint32_t coder;           /* -coder: 0 or 1 = range coder with the default state transitions; 2 = range coder whose transition
                                 table travels in the configuration record (FFV1_Parameters.cpp:41-55) */
    uint32_t level;           /* -level: 0 or 3 = FFV1 version 3; 1 = version 1, what the reference asks for with -slices 1
                                 (Global.cpp:961-968): num_h = num_v = 1, slicecrc 0, no configuration record (the header travels inside
                                 every frame, FFV1_Slice.cpp:224-268), no slice footer.  One chain per frame; the decoder insists on exactly
                                 the header this configuration produces. */
    uint32_t rc_span;         /* how the range coder is mapped: 0 = automatic; RCGPU_RC_WHOLE = one lane codes a whole slice (few, long
                                 chains: right when there are thousands of slices in flight); N >= 8 = split coder -- one lane per slice
                                 runs only the serial `range` recurrence and leaves a checkpoint every N 56-decision pieces, one lane
                                 per (slice, span of N pieces) then codes the span's bytes from its checkpoint, and the spans' residual
                                 `low` values are added into the bytes that follow them.  Same bytes either way. */
    uint32_t slice_buffer_div; /* 0 or 1 = slice byte buffers of the normal size; k > 1 = a k-th of it (tests of the overflow report) */
} rcgpu_ffv1_config;
#define RCGPU_RC_WHOLE 1u

typedef struct rcgpu_ffv1 rcgpu_ffv1;

int    rcgpu_ffv1_create(const rcgpu_ffv1_config* cfg, rcgpu_ffv1** enc);
/* Device memory ONE frame in flight costs an encoder of this configuration, the caller's payload and packet buffers included (symbols, context
 * states, decision windows, checkpoints, slice byte buffers where they are not overlaid; run_on != 0: with the second bank of run-on mode).
 * 4096x2160 RGB16, 64 slices: 379 MB, 506 MB in run-on mode -- with decision windows for the worst case of 35 decisions per sample; film content
 * measures ~330 / ~460.  Needs no device: what a caller sizes max_batch with (the job level does). */
uint64_t rcgpu_ffv1_device_bytes_per_frame(const rcgpu_ffv1_config* cfg, int run_on);
void   rcgpu_ffv1_destroy(rcgpu_ffv1* enc);
/* FFV1 configuration record incl. CRC = Matroska CodecPrivate (parsed at FFV1_Parameters.cpp:23-183). */
size_t rcgpu_ffv1_config_record(const rcgpu_ffv1* enc, uint8_t* out, size_t cap);
/* Worst-case packet bytes per frame (size d_packets as n * this). */
size_t rcgpu_ffv1_max_packet_bytes(const rcgpu_ffv1* enc);

/* Encode n (<= max_batch) frames whose payloads (data_size bytes each, first byte = first pixel) are
 * already resident in device memory.  All work is ordered on `hip_stream` (a hipStream_t, NULL = the default
 * stream); the call itself blocks once (exact decision counts come back to size the stream windows) and returns with the
 * remaining kernels enqueued.  On completion packet i occupies
 * d_packets[i*packet_stride .. + d_packet_sizes[i]) and is a complete FFV1 frame (all slices, footers, CRCs).
 *   d_frames       host array of n device pointers
 *   d_packets      device buffer, n * packet_stride bytes, packet_stride >= rcgpu_ffv1_max_packet_bytes()
 *   d_packet_sizes device array of n uint64
 */
int rcgpu_ffv1_encode_device(rcgpu_ffv1* enc, const void* const* d_frames, uint32_t n,
                             void* d_packets, size_t packet_stride, uint64_t* d_packet_sizes, void* hip_stream);

/* RUN-ON mode for a caller that encodes batch after batch with the frames already on the device.  A batch is k_model, then 32
 * segments of k_resolve with the range coder one segment behind, then footer / scan / gather: with one batch at a time the coder's
 * chain -- the critical path -- stands still for a ninth of a step.  With run-on mode on, rcgpu_ffv1_encode_device(k+1) models batch k+1
 * while batch k is in flight (the call still blocks until ITS k_model has run), lets its segments follow batch k's without a gap, and
 * makes `hip_stream` wait for batch k only: packets and sizes of the batch passed to a call are complete, in stream order, after the
 * NEXT call or after rcgpu_ffv1_join().  A batch's frames may be reused as soon as its call has returned, as before (k_model has
 * read them); consecutive batches get different d_packets / d_packet_sizes if the caller reads them in between.  Costs a second set
 * of per-batch buffers (symbols, states, coder output: rcgpu_ffv1_set_run_on fails if they do not fit); same packets, byte for byte.
 * Works with either mapping of the range coder (rc_span): the split coder's checkpoints are kept per bank as well. */
int rcgpu_ffv1_set_run_on(rcgpu_ffv1* enc, int on);
int rcgpu_ffv1_run_on(const rcgpu_ffv1* enc);                 /* 1 when the mode is on */
/* Makes `hip_stream` wait for every batch issued so far (the last one, in run-on mode). */
int rcgpu_ffv1_join(rcgpu_ffv1* enc, void* hip_stream);

/* Error word of the last batch (the device-pointer call above enqueues work and cannot report it): bit 0 a slice outgrew its byte
 * buffer, bit 1 a slice does not fit its footer or the 24-bit slice size field -- its packet is incomplete --, bit 2 more than 4096
 * late carries.  Synchronises the encoder's streams; returns an error (and the text) when *flags != 0. */
int rcgpu_ffv1_last_error_flags(rcgpu_ffv1* enc, uint32_t* flags);

/* Host-buffer convenience for one batch: H2D, encode, D2H in series, synchronous (sequences: rcgpu_ffv1_encode_sequence below).
 * out_packets[i] must hold rcgpu_ffv1_max_packet_bytes(). */
int rcgpu_ffv1_encode_host(rcgpu_ffv1* enc, const uint8_t* const* frames, uint32_t n,
                           uint8_t* const* out_packets, size_t* out_sizes);

/* A whole picture sequence with upload, encoding and download overlapped: what the ffmpeg process RAWcooked starts at
 * CLI/Output.cpp:356 does with the files of one `-i` (FileIO.cpp:274 maps them for the analysis before).  Reader threads call
 * read_frame to fill PINNED upload slots, each device uploads batch k+1 and downloads batch k-1 while batch k is coded, and the packets
 * come back in frame order: place_packet (one thread, frame order; may be NULL) says where packet `frame` of `size` bytes belongs --
 * e.g. inside a mapped output file, see rcgpu_mkv_reserve_block -- and writer threads copy it there and call packet_done; when
 * place_packet is NULL or returns NULL, packet_done receives the pinned buffer itself (valid during the call).  Frames shard over the
 * selected devices by batch, no collective.  cfg->device is ignored (options.device_first / device_count); the batch is options.batch,
 * else cfg->max_batch when that is above 1, else sized from the device's free memory and the sequence length.  All callbacks return 0
 * for success; a failure ends the job with that code. */
typedef struct {
    uint32_t struct_size;               /* sizeof(rcgpu_sequence_io) as the caller was compiled: a caller built against another rcgpu.h is refused
                                           instead of having a slot it never set called */
    int      (*read_frame)(void* user, uint64_t frame, uint8_t* dst, size_t payload_bytes);     /* reader threads, concurrent */
    uint8_t* (*place_packet)(void* user, uint64_t frame, size_t size);                         /* one thread, frame order; optional */
    int      (*packet_done)(void* user, uint64_t frame, const uint8_t* data, size_t size);     /* writer threads, concurrent */
    void* user;
    /* optional, INSTEAD of read_frame (exactly one of the two is set; both is refused): the frame's payload lies in PINNED host memory already
     * (hipHostMalloc / hipHostRegister) and stays there until its batch has been modelled; it is uploaded from where it is -- no upload slots,
     * no reader threads, no copy.  SURVEY.md 8d's "inputs resident in pinned host memory, H2D included".  Called from the LANE threads (one per
     * device and lane), concurrently, in no order across lanes, and with run-on encoders while the lane's previous batch is still in flight. */
    const uint8_t* (*locate_frame)(void* user, uint64_t frame, size_t payload_bytes);
} rcgpu_sequence_io;
typedef struct {
    uint32_t struct_size;               /* sizeof(rcgpu_sequence_options) as the caller was compiled (see rcgpu_sequence_io) */
    int device_first, device_count;     /* 0 devices = all visible */
    uint32_t batch;                     /* frames per batch and device; 0 = automatic */
    uint32_t readers, writers;          /* host threads; 0 = automatic */
    uint32_t in_ring_frames;            /* pinned upload slots; 0 = automatic */
    uint64_t out_ring_bytes;            /* pinned download ring per device; 0 = automatic */
    uint32_t lanes_per_device;          /* encoder instances per device, their batches staggered (shorter head and tail of a job); 0 = 1 */
    uint32_t copy_streams;              /* copy streams per direction and lane; 0 = automatic (two: one stream's copies run strictly one after the other) */
    uint32_t device_aliases;            /* test hook, 0 or 1 = off: k > 1 presents every physical device k times to device_first / device_count, so
                                           that the lane-per-device path (one encoder, ring and set of copy streams per device, one placer across
                                           them) runs on a box with a single GPU; the lanes then share its memory -- pass `batch` */
    uint32_t frames_pinned;             /* rcgpu_ffv1_encode_sequence_memory only: 1 = frames[] point into pinned host memory: they are uploaded from
                                           there (locate_frame instead of read_frame) */
    uint32_t run_on;                    /* 1 = the encoders run on from batch to batch (rcgpu_ffv1_set_run_on) where the device has room for their second
                                           bank; 0 = one batch at a time (the pipeline is paced by its transfers: measured +0.5 % for 75 GB) */
    uint32_t numa;                      /* 0 = automatic: lanes are grouped by the NUMA node their device hangs on (hipDeviceGetPCIBusId ->
                                           /sys/bus/pci/devices/<id>/numa_node); each group has its own pinned upload slots, reader and writer threads,
                                           all bound to the node's CPUs, and every lane's download ring is allocated there -- a lane moves ~118 GB/s
                                           through host memory, which eight lanes cannot take across a socket link.  1 = off (one group, nothing
                                           bound).  2 = test hook: lane i is treated as attached to node i mod (nodes of the host) */
} rcgpu_sequence_options;
typedef struct {
    double   seconds;                   /* first read_frame .. last packet_done */
    double   first_packet_seconds, prepare_seconds;
    double   device_busy_seconds;       /* device 0: first encode call .. completion of its last batch */
    uint64_t frames, payload_bytes, packet_bytes, batches;
    uint32_t batch_frames, devices, readers, writers;
    double   steady_frames_per_second;  /* all batches but the first / time from the first batch's completion to the last's */
    double   reads_done_seconds, last_batch_seconds;
    double   upload_wait_seconds;       /* device 0: time its host thread waited for the readers */
    double   h2d_span_seconds;          /* device 0: sum over batches of first upload start .. last upload end (device clock, HIP events) */
    double   read_call_seconds, write_call_seconds;   /* average duration of one read_frame call / one packet copy + packet_done call */
    uint32_t host_groups;               /* NUMA groups the lanes fell into */
    int32_t  lane_device[16], lane_numa_node[16], lane_pinned_node[16];   /* per lane (first 16): its device, the node that device hangs on, and the node
                                           the kernel reports for the lane's pinned download ring (move_pages); -1 = unknown / no such lane */
} rcgpu_sequence_stats;
/* record/record_size: optional, the FFV1 configuration record (Matroska CodecPrivate), *record_size = capacity in, size out. */
int rcgpu_ffv1_encode_sequence(const rcgpu_ffv1_config* cfg, uint64_t n_frames, const rcgpu_sequence_io* io,
                               const rcgpu_sequence_options* options, rcgpu_sequence_stats* stats, uint8_t* record, size_t* record_size);

/* The sharding rcgpu_ffv1_encode_sequence follows, without a device: frames are coded in batches of `batch` consecutive frames, batch b on
 * lane b mod L (frames shard with no exchange because every frame is a key frame and every slice resets its contexts,
 * CLI/Global.cpp:959-960).  `lanes` = the devices of the selection (x lanes_per_device); L = the lanes the job really gets -- a short job
 * uses no more devices than it has batches of 8 frames for, exactly as the pipeline decides it: rcgpu_sequence_plan_lanes(n_frames, devices,
 * lanes_per_device) says how many (20 frames on 8 devices: 3).  It matches the pipeline when `batch` is the batch the pipeline uses
 * (options.batch passed explicitly; an automatic batch comes from the device's free memory).  lane_of_frame / batch_of_frame: n_frames
 * entries each, either may be NULL.  For callers that want their input staged next to the device that will read it. */
int rcgpu_sequence_plan(uint64_t n_frames, uint32_t batch, uint32_t lanes, uint32_t* lane_of_frame, uint32_t* batch_of_frame);
uint32_t rcgpu_sequence_plan_lanes(uint64_t n_frames, uint32_t devices, uint32_t lanes_per_device);

/* The same with host memory on both ends: frame i = frames[i % n_in], packet i -> out[i % n_out] (out_cap bytes each, may be NULL to
 * drop the bytes) and sizes[i] (n_frames entries, may be NULL).  n_in == n_out == n_frames: rcgpu_ffv1_encode_host for a whole
 * sequence, pipelined. */
int rcgpu_ffv1_encode_sequence_memory(const rcgpu_ffv1_config* cfg, const uint8_t* const* frames, uint64_t n_in, uint64_t n_frames,
                                      uint8_t* const* out, uint64_t n_out, size_t out_cap, uint64_t* sizes,
                                      const rcgpu_sequence_options* options, rcgpu_sequence_stats* stats, uint8_t* record, size_t* record_size);

/* `-f framemd5` (CLI/Output.cpp:312-332): MD5 of the first n frames of the LAST batch as the bytes FFmpeg's rawvideo encoder would hash
 * (rgb24/rgba/gray, rgb48/rgba64/gray16 in the file's endianness, gbrp/gbrap/gray 10/12 little-endian planar) [ffmpeg-knowledge];
 * *frame_bytes = size of one such frame.  Call between two batches, once the batch's stream is synchronised (rcgpu_ffv1_encode_host
 * returns that way) and while the payload buffers it read are still alive; the symbol buffer serves as scratch.  Synchronous. */
int  rcgpu_ffv1_framemd5_last(rcgpu_ffv1* enc, uint32_t n, uint8_t* out_md5, uint64_t* frame_bytes);
/* Per-kernel device time (summed over its launches) of the last encode call on this encoder, measured with HIP events
 * on the stream the kernels were launched on.  names[i] is a static string. Returns the number of entries written. */
int rcgpu_ffv1_last_kernel_times(const rcgpu_ffv1* enc, const char** names, float* ms, int cap);
int rcgpu_ffv1_last_kernel_launches(const rcgpu_ffv1* enc, int index);   /* launches of kernel `index` in the last call */
/* Device time from the end of one batch to the end of the next for the last n <= min(cap, 63) pairs of batches issued through
 * rcgpu_ffv1_encode_device (HIP events behind each batch's last kernel, on the stream it ran on; the call waits for the newest): ms[0] is the
 * oldest interval.  In run-on mode these are the step times of a caller that issues batch after batch -- a stream of the caller's own that waited
 * for the batches to time them would share a hardware queue with the encoder's and hold the next batch back.  Returns n. */
int rcgpu_ffv1_batch_intervals(const rcgpu_ffv1* enc, float* ms, int cap);
/* Totals of the last batch (valid after the stream is synchronised): binary range-coder decisions and packet bytes. */
int rcgpu_ffv1_last_stats(const rcgpu_ffv1* enc, uint64_t* decisions, uint64_t* packet_bytes);

/* ---- FFV1 decoder + verification (device): the `--check` half (BASELINE config 5).  Restates ffv1_frame::Process
 * (FFV1_Frame.cpp:134-228), slice::Parse/Line (FFV1_Slice.cpp:210-472), rangecoder (FFV1_RangeCoder.cpp:71-305) and
 * Transform::From (Lib/Transform/Transform.cpp) on the GPU; the comparison semantics are frame_writer's
 * (Lib/Utils/FileIO/FileWriter.cpp:448-463 byte compare, :596-727 MD5).  The configuration is the encoder's
 * (same struct; `segments` ignored): it must describe the stream being decoded. */
typedef struct rcgpu_ffv1_decoder rcgpu_ffv1_decoder;
int  rcgpu_ffv1_decoder_create(const rcgpu_ffv1_config* cfg, rcgpu_ffv1_decoder** dec);
void rcgpu_ffv1_decoder_destroy(rcgpu_ffv1_decoder* dec);
/* Decode n (<= max_batch) packets resident in device memory into n payload buffers in device memory (data_size bytes
 * each, padding bits zero -- the caller XORs the reversibility `InData`, RawFrame.cpp:184-206).  When h_err_flags is
 * not NULL the call synchronises and returns an error if any slice failed (bad split 1|2, CRC 4, header 32, underrun 64,
 * junk 128, error_status 256); with NULL it is asynchronous on `hip_stream`.
 * On an MI355X the slices are decoded on 248 of the 256 CUs and rcgpu_md5_device / verify_kept hash on the other eight (CU-masked
 * streams inside the library, joined to `hip_stream` by events: the order the caller sees is unchanged).  Such streams synchronise
 * with the legacy default stream: to hash batch k-1 WHILE batch k is decoded, give both calls streams of their own, not NULL. */
int  rcgpu_ffv1_decoder_decode_device(rcgpu_ffv1_decoder* dec, const void* const* d_packets, const uint64_t* packet_sizes, uint32_t n,
                                      void* const* d_payloads, uint32_t* h_err_flags, void* hip_stream);
/* Host-buffer convenience for a caller that holds the Matroska blocks in memory (what ffv1_wrapper::Process receives one at a
 * time, Lib/CoDec/Wrapper.cpp:115-121): packets go up, payloads (payload_bytes each, see rcgpu_image_info.data_size) come back.
 * Synchronous; returns an error if any slice of the batch is undecodable. */
int  rcgpu_ffv1_decoder_decode_host(rcgpu_ffv1_decoder* dec, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n,
                                    uint8_t* const* payloads);
/* `--check` with nothing but verdicts coming back.  frame_writer::FrameCall (FileWriter.cpp:73-300) receives every rebuilt file on the
 * host, hashes it (CheckMD5, :596-727, one core) and compares it with the file on disk (CheckFile / CheckFile_Compare, :448-463,
 * :464-594).  These three calls keep the decoded payloads on the device and do both there, for a whole batch of files at once:
 *   decode_keep   decodes n (<= max_batch) packets held by the host (the Matroska blocks); payload i stays on the device as slot i
 *                 until the next decode_keep / decode_host of this decoder;
 *   kept_to_host  one kept payload after all (a caller that has `InData` to merge, RawFrame.cpp:184-206, or a file to write);
 *   verify_kept   for n files, each = `before` + the payload of `slot` + `after` (raw_frame's Pre / plane / Post): its MD5 when
 *                 RCGPU_KEPT_MD5 is set, and, when `on_disk` is not NULL, the offset of the first byte at which it differs from the
 *                 on_disk_size bytes at on_disk (e.g. the mapped source file): UINT64_MAX when they are the same, the shorter length when
 *                 one is the beginning of the other.  `before` and `after` are at most RCGPU_KEPT_ROOM bytes each. */
#define RCGPU_KEPT_MD5  1u
#define RCGPU_KEPT_ROOM 65536u
typedef struct rcgpu_kept_file {
    uint32_t slot, flags;
    const uint8_t* before;  uint64_t before_size;
    const uint8_t* after;   uint64_t after_size;
    const uint8_t* on_disk; uint64_t on_disk_size;
    const char* on_disk_path;   /* when on_disk is NULL: the file to compare with, by name -- read with pread() straight into the staging buffers
                                   (256 4K files: 0.5-0.8 s; mapped by the caller and passed as on_disk: 0.25 s plus the unmapping) */
} rcgpu_kept_file;
typedef struct rcgpu_kept_verdict {
    uint8_t  md5[16];
    uint64_t first_diff;
} rcgpu_kept_verdict;
int  rcgpu_ffv1_decoder_decode_keep(rcgpu_ffv1_decoder* dec, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n);
/* The same with the packets named by their place in an open file -- the Matroska file the blocks lie in -- instead of by address, for a caller
 * that has not mapped it: they are read with pread() straight into the staging buffers.  (From a mapping the bytes go up faster -- 6 GB in 0.3 s
 * against 0.35-0.7 s -- at the price of ~0.4 s when the mapping with all its pages touched is torn down, as matroska::ParseBuffer does after
 * every MiB, Matroska.cpp:394-419: about even.) */
/* One batch AHEAD: the batch the caller will pass to its next decode_keep, started now on a thread of the library's own (packets up,
 * decode, into slots that are neither the current ones nor under verification) while the caller hands out the current batch's frames
 * and has them verified.  The packets must stay where they are until that decode_keep (a mapped file; not a buffer that is refilled).
 * The next decode_keep with the same pointers and sizes adopts the result; with any other batch, or after an error, the batch is
 * decoded as if nothing had been hinted.  rcgpu_ffv1_decoder_decode_host drops a hinted batch; decode_device must not be called
 * while one is in flight. */
int  rcgpu_ffv1_decoder_decode_keep_hint(rcgpu_ffv1_decoder* dec, const uint8_t* const* packets, const uint64_t* packet_sizes, uint32_t n);
/* The same for a caller whose pointers do not last (the reference maps its Matroska file anew every megabyte, Matroska.cpp:394-408): the
 * packets by their place in the file, which the library maps for itself; and the adoption as a call of its own -- the caller knows
 * whether the batch it is about to ask for is the one it hinted.  An error from _adopt (nothing hinted, or the hinted batch failed)
 * leaves the caller to decode the batch with decode_keep. */
int  rcgpu_ffv1_decoder_decode_keep_hint_file(rcgpu_ffv1_decoder* dec, const char* path, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n);
int  rcgpu_ffv1_decoder_decode_keep_adopt(rcgpu_ffv1_decoder* dec);
int  rcgpu_ffv1_decoder_decode_keep_fd(rcgpu_ffv1_decoder* dec, int fd, const uint64_t* offsets, const uint64_t* packet_sizes, uint32_t n);
int  rcgpu_ffv1_decoder_kept_to_host(rcgpu_ffv1_decoder* dec, uint32_t slot, uint8_t* payload);
int  rcgpu_ffv1_decoder_verify_kept(rcgpu_ffv1_decoder* dec, const rcgpu_kept_file* files, uint32_t n, rcgpu_kept_verdict* verdicts);
/* The same in two calls, for a caller with more than one batch: _begin takes what it needs of the caller's memory (it may be released when
 * the call returns), compares, and starts the hashes -- MD5 is serial per file, 0.8 s for a 53 MB file however many there are; the next
 * rcgpu_ffv1_decoder_decode_keep then fills a second set of slots while they run, and _end (before the decode_keep after that) hands out the n
 * verdicts.  One verification at a time. */
int  rcgpu_ffv1_decoder_verify_kept_begin(rcgpu_ffv1_decoder* dec, const rcgpu_kept_file* files, uint32_t n);
int  rcgpu_ffv1_decoder_verify_kept_end(rcgpu_ffv1_decoder* dec, rcgpu_kept_verdict* verdicts);
/* Fill the stream-dependent fields of `cfg` (num_h_slices, num_v_slices, slicecrc, context, coder) from a Matroska CodecPrivate =
 * FFV1 configuration record, the way parameters::Parse reads it (FFV1_Parameters.cpp:23-183); width, height, pixfmt, line_bytes
 * and flags describe the files and are the caller's (they come from the reversibility data / the probes).  Fails when the record
 * does not describe `pixfmt` (colorspace, bit depth, alpha) or uses a feature this decoder lacks. */
int  rcgpu_ffv1_config_from_record(const uint8_t* record, size_t size, rcgpu_ffv1_config* cfg);
/* The same plus the one stream fact the record does not hold: which of its table sets the planes use (quant_table_set_index in every
 * slice header, FFV1_Slice.cpp:159-168), read from the first slice header of `packet` (the first frame of the track). */
int  rcgpu_ffv1_config_from_stream(const uint8_t* record, size_t size, const uint8_t* packet, size_t packet_size, rcgpu_ffv1_config* cfg);
/* ---- Any stream the reference's decoder takes.  The two calls above describe a stream through rcgpu_ffv1_config, which can only say
 * what this library's ENCODER writes (FFmpeg's default table sets or the compact model, every plane on the same set, one of two known
 * transition tables).  `--check` also meets files other encoders wrote.  rcgpu_ffv1_stream_parse reads a stream the way
 * parameters::Parse does (Lib/CoDec/FFV1/FFV1_Parameters.cpp:23-183, tables :206-253): versions 0, 1 (no CodecPrivate: record = NULL,
 * record_size = 0 and the header is read out of `packet`, slice::Parse FFV1_Slice.cpp:224-245) and 3, coder_type 1 and 2 with ANY
 * transmitted transition table (:41-55), 1 to 8 quantisation table sets of arbitrary tables up to 32 768 contexts, coded initial states
 * (states_coded, read as the reference reads them, :103-107), and from the first slice header of `packet` -- the first frame of the
 * track -- the quant_table_set_index of every plane group (FFV1_Slice.cpp:158-168).
 *   return 0                       parsed; *stream must be freed with rcgpu_ffv1_stream_free
 *   return RCGPU_FFV1_UNSUPPORTED  a valid stream the device decoder does not take (Golomb-Rice; more slice rows than columns -- which the
 *                                  reference itself misreads, FFV1_Slice.cpp:125, so that its verdict must stay its own; and from _create_for_stream: YUV planes,
 *                                  gray with an alpha plane, inter frames, a state 0 within reach of the initial states): the caller decodes it with its own decoder --
 *                                  ffv1_frame::Process stays on its slice pool (oracle/route_c_ffv1_frame_cpp.patch)
 *   any other value                what the reference refuses as well (its error in rcgpu_last_error()), a record that ends before its fields do,
 *                                  or a first frame that is no key frame / whose first slice header the reference flags.  The reader is held to
 *                                  the reference's own (tests/golden/parse_cases.*): what the reference refuses it refuses, what both read they read alike.
 *   from _create_for_stream also 5 a valid stream that describes OTHER files than `files` (RGB against gray, another bit depth, alpha against none;
 *                                  with colorspace_type 1 chroma_planes is ignored, as the reference ignores it, FFV1_Parameters.cpp:174)
 * rcgpu_ffv1_decoder_create_for_stream takes width, height, pixfmt, line_bytes, flags, max_batch and device from `files` (the stream fields
 * of the struct are ignored) and everything else from `stream`; it fails when the stream does not describe `pixfmt` (colorspace, bit
 * depth, alpha).  The decoder it makes is used like any other; a slice whose header names other table sets than the first slice did
 * (they are per-slice fields) is reported as undecodable (flag 32) and left to the caller's decoder. */
#define RCGPU_FFV1_UNSUPPORTED 20
typedef struct rcgpu_ffv1_stream rcgpu_ffv1_stream;
typedef struct {
    uint32_t version, micro_version, coder_type /* 1, or 2 = transmitted transitions */, colorspace_type, bits_per_raw_sample;
    uint32_t chroma_planes, alpha_plane, num_h_slices, num_v_slices, quant_table_set_count, ec, intra;
    uint32_t quant_table_set_index_count, quant_table_set_index[3];   /* plane group 0 (Y), 1 (Cb, Cr), 2 (alpha) */
    uint32_t context_count[8], states_coded[8];
} rcgpu_ffv1_stream_info;
int  rcgpu_ffv1_stream_parse(const uint8_t* record, size_t record_size, const uint8_t* packet, size_t packet_size, rcgpu_ffv1_stream** stream);
void rcgpu_ffv1_stream_free(rcgpu_ffv1_stream* stream);
int  rcgpu_ffv1_stream_get_info(const rcgpu_ffv1_stream* stream, rcgpu_ffv1_stream_info* info);
/* What the description holds beyond the numbers, for a caller that wants to look (tests/test_host.py compares it with what the reference's
 * own parameters::Parse made of the same bytes): the state transition table in force (256 bytes, one_state; may be NULL), and of table set
 * `set` the five quantisation tables (5 x 256 values as parameters::QuantizationTable leaves them, FFV1_Parameters.cpp:222-253; may be NULL)
 * and the coded initial states (context_count x 32 bytes into initial_states when it is not NULL and the capacity suffices;
 * *initial_size = their size, 0 when the set's states start at 128). */
int  rcgpu_ffv1_stream_get_tables(const rcgpu_ffv1_stream* stream, uint32_t set, uint8_t* state_transitions, int16_t* quant_tables,
                                  uint8_t* initial_states, uint64_t initial_capacity, uint64_t* initial_size);
int  rcgpu_ffv1_decoder_create_for_stream(const rcgpu_ffv1_config* files, const rcgpu_ffv1_stream* stream, rcgpu_ffv1_decoder** dec);
int  rcgpu_ffv1_decoder_last_kernel_times(const rcgpu_ffv1_decoder* dec, float ms[3]);   /* split+crc, slices, pack */
/* First differing byte of two device buffers; *first_diff = UINT64_MAX when they are equal (FileWriter.cpp:448-463). */
int  rcgpu_compare_device(const void* d_a, const void* d_b, uint64_t n, uint64_t* first_diff, void* hip_stream);
/* The same for n pairs of device buffers in one launch: first_diff[i] (host) for pair i.  Ordered on hip_stream alone, like rcgpu_md5_device:
 * a --check binding compares batch k-1 on a side stream while batch k is decoded. */
int  rcgpu_compare_device_batch(const void* const* d_a, const void* const* d_b, const uint64_t* sizes, uint32_t n, uint64_t* first_diff, void* hip_stream);
/* The library's CU-masked streams -- one hash stream per device for rcgpu_md5_device / rcgpu_analysis_host_batch, and the (decode, hash) pairs its
 * decoders borrow from a pool -- live as long as the library (they are pooled, not destroyed with a decoder: the HIP runtime of ROCm 7.0 cannot
 * survive an out-of-memory hipMalloc once such a stream has been destroyed).  This call is for the END OF A PROCESS only (a profiler's
 * finalisation after this library's streams; not needed for correctness): it destroys the ones not in use and latches -- from then on the library
 * makes and hands out plain streams only (same results; the hash shares CUs with the decoder), a decoder still alive gives its pair back to
 * be destroyed, and an rcgpu_md5_device that is in flight on another thread must have returned before the call.  On ROCm 7.0 an out-of-memory
 * hipMalloc AFTER this call may crash the process (the runtime's defect above): do not rely on allocation errors behind it. */
void rcgpu_release_device_streams(void);
/* MD5 of n device buffers, one lane per buffer; out_md5 = n x 16 bytes on the host (FileWriter.cpp:596-727). */
int  rcgpu_md5_device(const void* const* d_bufs, const uint64_t* sizes, uint32_t n, uint8_t* out_md5, void* hip_stream);
/* The same for n host buffers, e.g. memory-mapped source files during analysis (input_base::Hash, Lib/Uncompressed/../Input_Base.cpp:54-81
 * hashes them one at a time on one core): uploaded once, hashed side by side. */
int  rcgpu_md5_host_batch(const uint8_t* const* bufs, const uint64_t* sizes, uint32_t n, uint8_t* out_md5, int device);
/* Both passes the analysis makes over every byte of a source file, for n mapped files that go up once: the whole-file MD5 (out_md5, n x 16;
 * may be NULL) and, for the files rcgpu_dpx_probe recognises, the padding-bit test of dpx::ParseBuffer (DPX.cpp:501-608): scanned[i] = 1
 * and first_nonzero[i] = the reference's In_FirstNonZero relative to the payload, or UINT64_MAX when every padding bit is zero -- then the
 * reference's loop has nothing to find (first_nonzero and scanned may both be NULL). */
int  rcgpu_analysis_host_batch(const uint8_t* const* files, const uint64_t* sizes, uint32_t n, uint8_t* out_md5, uint64_t* first_nonzero, uint8_t* scanned, int device);
/* The padding-bit test of the DPX analysis (dpx::ParseBuffer, Lib/Uncompressed/DPX/DPX.cpp:501-608) for n device payloads of one
 * layout -- e.g. the frames just uploaded for encoding: first_nonzero[i] (host) = the reference's In_FirstNonZero relative to the
 * payload, or UINT64_MAX when every padding bit is zero (then no "In" block is needed in the reversibility data). */
int  rcgpu_dpx_padding_scan_device(const void* const* d_payloads, uint32_t n, uint32_t pixfmt, uint32_t width, uint32_t height, uint32_t flags,
                                   uint64_t* first_nonzero, void* hip_stream);

/* ===========================================================================================
 * 4. FLAC encoder (device) -- replaces FFmpeg's flacenc; inverse of flac_wrapper (Lib/CoDec/Wrapper.cpp:131-373)
 *    + libFLAC stream_decoder.c:2012-2788.
 * ======================================================================================== */
typedef struct {
    uint32_t channels, sample_rate, bits_per_sample;   /* 1..8, 1..655350, 8/16/24 */
    uint32_t block_size;                               /* samples per channel per FLAC frame; 0 = 4608 @48k like FFmpeg */
    uint32_t max_lpc_order;                            /* 0 = fixed predictors only; else 1..32 */
    int      device;
} rcgpu_flac_config;

typedef struct rcgpu_flac rcgpu_flac;

int    rcgpu_flac_create(const rcgpu_flac_config* cfg, rcgpu_flac** enc);
void   rcgpu_flac_destroy(rcgpu_flac* enc);
/* Encode interleaved little-endian PCM (WAV data chunk bytes).  frames_out receives the concatenated FLAC
 * frames; frame_sizes[i] the size of frame i.  Returns number of FLAC frames via *n_frames. */
int    rcgpu_flac_encode_host(rcgpu_flac* enc, const uint8_t* pcm, uint64_t pcm_bytes,
                              uint8_t* frames_out, size_t cap, uint32_t* frame_sizes, uint32_t frame_cap, uint32_t* n_frames);
/* "fLaC" + STREAMINFO = Matroska CodecPrivate for A_FLAC (Wrapper.cpp:138; stream_decoder.c:1565-1634).
 * Valid after the whole stream was encoded (total samples, min/max frame size, MD5). */
size_t rcgpu_flac_codec_private(const rcgpu_flac* enc, uint8_t* out, size_t cap);

/* ===========================================================================================
 * 5. Matroska muxer -- replaces FFmpeg's matroskaenc for this path; emits what the reference's reader
 *    requires (Lib/Compressed/Matroska/Matroska.cpp:128-217, :863-873, :934-953, :1007-1030, :1259-1277).
 * ======================================================================================== */
typedef struct rcgpu_mkv rcgpu_mkv;

int  rcgpu_mkv_open(const char* path, int overwrite, rcgpu_mkv** mux);
/* returns 1-based track number, <0 on error */
int  rcgpu_mkv_add_video(rcgpu_mkv* mux, const uint8_t* codec_private, size_t cp_size, uint32_t width, uint32_t height,
                         uint32_t fps_num, uint32_t fps_den);
int  rcgpu_mkv_add_audio(rcgpu_mkv* mux, const uint8_t* codec_private, size_t cp_size, uint32_t channels,
                         uint32_t sample_rate, uint32_t bits_per_sample);
/* A PCM track for `-c:a copy` (CLI/Main.cpp:300-317, test/pcm.sh): CodecID A_PCM/INT/LIT or A_PCM/FLOAT/IEEE, no CodecPrivate;
   the blocks carry whole sample frames of the WAV data chunk as they are (reader: pcm_wrapper, Lib/CoDec/Wrapper.cpp:376-388). */
int  rcgpu_mkv_add_audio_pcm(rcgpu_mkv* mux, int is_float, uint32_t channels, uint32_t sample_rate, uint32_t bits_per_sample);
int  rcgpu_mkv_add_attachment(rcgpu_mkv* mux, const char* name, const char* mime, const uint8_t* data, size_t size);
/* A track-level SimpleTag (what FFmpeg writes for `-metadata:s:N name=value`; the reference adds one to EXR packages). */
int  rcgpu_mkv_add_tag(rcgpu_mkv* mux, int track, const char* name, const char* value);
/* Writes EBML header, Segment, SeekHead, Info, Tracks, Attachments.  Call once after all add_* calls. */
int  rcgpu_mkv_begin(rcgpu_mkv* mux);
/* One SimpleBlock (one FFV1 frame or >= 1 whole FLAC frames); pts in nanoseconds, non-decreasing per track. */
int  rcgpu_mkv_write_block(rcgpu_mkv* mux, int track, uint64_t pts_ns, const uint8_t* data, size_t size, int keyframe);
/* Parallel writers: after begin(), announce an upper bound of the block payload still to come.  On tmpfs the muxer sizes the file for
 * it, maps that part and allocates its pages ahead of the writers (one output file is one inode: through write() it takes 6 GB/s
 * there, through pre-allocated mapped pages 50+); on other file systems this is a no-op and the payloads go through pwrite().  reserve_block() then lays out one
 * Cluster + SimpleBlock in stream order and returns where its `size` payload bytes belong: *dst inside the mapping (copy there from
 * any thread), or *dst == NULL and *file_offset for rcgpu_mkv_fill() (pwrite; thread-safe).  close() cuts the file to its real size. */
int  rcgpu_mkv_expect(rcgpu_mkv* mux, uint64_t max_block_bytes, uint64_t max_blocks);
int  rcgpu_mkv_reserve_block(rcgpu_mkv* mux, int track, uint64_t pts_ns, size_t size, int keyframe, uint8_t** dst, uint64_t* file_offset);
int  rcgpu_mkv_fill(rcgpu_mkv* mux, uint64_t file_offset, const uint8_t* data, size_t size);
int  rcgpu_mkv_copy_in(rcgpu_mkv* mux, uint8_t* dst, const uint8_t* src, size_t size);   /* copies a payload to its *dst (any thread): waits for the
                                     pages to exist, maps the range in one call, copies.  When the pages cannot be had (the file system is
                                     full) the payload goes through pwrite() instead and its error comes back: never a SIGBUS */
/* Patches A_FLAC CodecPrivate written by begin() (same size) once STREAMINFO is final. */
int  rcgpu_mkv_update_codec_private(rcgpu_mkv* mux, int track, const uint8_t* codec_private, size_t cp_size);
/* Writes Cues, patches Segment size / SeekHead / Duration; closes the file. */
int  rcgpu_mkv_close(rcgpu_mkv* mux);

/* ===========================================================================================
 * 6. Hashes -- Lib/ThirdParty/md5/md5.c as used by Utils/FileIO/Input_Base.cpp:54-81, FileWriter.cpp:596-727
 * ======================================================================================== */
void     rcgpu_md5(const uint8_t* data, size_t size, uint8_t out[16]);
uint32_t rcgpu_crc32_ffv1(const uint8_t* data, size_t size);   /* Utils/CRC32/ZenCRC32.cpp:1097-1135 */

#ifdef __cplusplus
}
#endif
#endif
