/* ffv1_oracle.h -- CPU oracle for the FFV1 v3 intra path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the algorithm on RAWcooked's DPX/TIFF -> FFV1 hot path
 * (SURVEY.md section 8 rows a6, a7, a9).  The decode half follows the reference's in-tree
 * decoder line by line; the encode half is its exact inverse (the reference delegates encoding
 * to an external FFmpeg process, Source/CLI/Output.cpp:356, so the inverse is pinned by
 * round-tripping through the real reference binary oracle/_ref/rawcooked, see
 * tests/test_oracle_vs_reference.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (rawcooked_amd/csrc) never includes or links anything from oracle/.
 */
#ifndef FFV1_ORACLE_H
#define FFV1_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Pixel layouts of the uncompressed payload, named after the reference's flavors
 * (Source/Lib/Uncompressed/DPX/DPX.cpp:184-231, TIFF/TIFF.cpp:157-173). */
enum {
    FFV1O_RGB8 = 0,            /* DPX/TIFF Raw/RGB/8bit            3 B/px  R,G,B                         */
    FFV1O_RGB10_FILLEDA_BE = 1,/* DPX Raw/RGB/10bit/FilledA BE     4 B/px  R<<22|G<<12|B<<2             */
    FFV1O_RGB10_FILLEDA_LE = 2,
    FFV1O_RGB12_FILLEDA_BE = 3,/* DPX Raw/RGB/12bit/FilledA        6 B/px  3 x (v<<4) u16               */
    FFV1O_RGB12_FILLEDA_LE = 4,
    FFV1O_RGB16_BE = 5,        /* DPX/TIFF Raw/RGB/16bit           6 B/px  R,G,B u16                     */
    FFV1O_RGB16_LE = 6,
    FFV1O_RGBA8 = 7,           /* 4 B/px R,G,B,A */
    FFV1O_RGBA16_BE = 8,       /* 8 B/px */
    FFV1O_RGBA16_LE = 9,
    FFV1O_Y8 = 10,             /* 1 B/px */
    FFV1O_Y16_BE = 11,         /* 2 B/px */
    FFV1O_Y16_LE = 12,
    /* bit-packed DPX flavors (DPX.cpp:184-207); "packed" = 12-bit fields filling big-endian 32-bit words from the LSB up,
     * a line padded to a whole word (Transform.cpp:161-322, 521-550, 851-990) */
    FFV1O_RGB12_PACKED_BE = 13,
    FFV1O_RGBA10_FILLEDA_BE = 14, /* 3 fields per 32-bit word at <<22,<<12,<<2 running R,G,B,A,R,... (Transform.cpp:445-518) */
    FFV1O_RGBA10_FILLEDA_LE = 15,
    FFV1O_RGBA12_PACKED_BE = 16,
    FFV1O_RGBA12_FILLEDA_BE = 17, /* 8 B/px 4 x (v<<4) u16 (Transform.cpp:553-600) */
    FFV1O_RGBA12_FILLEDA_LE = 18,
    FFV1O_Y10_FILLEDA_BE = 19,    /* 3 samples per big-endian word at <<2,<<12,<<22 (Transform.cpp:709-822, Offset 2) */
    FFV1O_Y10_FILLEDB_BE = 20,    /* ... at <<0,<<10,<<20 (Offset 0) */
    FFV1O_Y12_PACKED_BE = 21,
    FFV1O_EXR_RGB16 = 22,         /* OpenEXR scan line: [y:u32][bytes:u32][B x w][G x w][R x w] u16 LE (Transform.cpp:1062-1127) */
    FFV1O_PIXFMT_COUNT
};
#define FFV1O_FLAG_VFLIP  1u   /* picture line y is file line height-1-y (DPX orientation 2 + "-vf vflip", Main.cpp:207-211) */
#define FFV1O_FLAG_ALTERN 2u   /* Y 10-bit only: words are filled across line ends, no line padding (DPX.cpp:363-368) */

/* A stream laid out as freely as parameters::Parse reads one (FFV1_Parameters.cpp:23-183, :206-253): what FFmpeg's defaults never
 * produce but the reference's decoder accepts -- up to 8 quantisation table sets of arbitrary level maps, a set index per plane
 * group, any transmitted state-transition table, coded initial states, the version 0 / 1 header inside every frame.  Attached to
 * ffv1o_params.ext; NULL there = the two FFmpeg-default sets of build_quant_sets_c. */
typedef struct ffv1o_stream_ext {
    uint32_t version;              /* 0 or 1: the header travels in every frame, one slice, no footer; 3: configuration record */
    uint32_t micro_version;        /* version 3 only; the reference refuses < 4 (:36-37) */
    uint32_t custom_transitions;   /* 1: coder_type 2, one_state[1..255] travels as deltas to the default table (:41-55) */
    uint8_t  one_state[256];
    uint32_t set_count;            /* quant_table_set_count, 1..8 (version 0 / 1: 1) */
    uint8_t  levels[8][5][128];    /* level of |difference| k: 0 at k = 0, non-decreasing, steps of 1 (the record carries the run lengths, :222-253) */
    uint32_t states_coded[8];      /* version 3: the set's initial states travel in the record */
    const uint8_t* initial_states[8]; /* context_count x 32 states, stored the way the REFERENCE reads them: every value as it is
                                         (`States[k] = E.s(States)`, :103-107) -- not RFC 9043's delta to the context before */
    uint32_t set_index[3];         /* quant_table_set_index of plane group 0 (Y), 1 (Cb, Cr), 2 (alpha) in every slice header (FFV1_Slice.cpp:159-168) */
    uint32_t intra;                /* version 3 record field; 1 = every frame is a key frame (-g 1).  (0 with key frames only is a valid stream too) */
    uint32_t alt_slices;           /* 1: slices with odd sx + sy write set_index_alt instead -- the index is a per-SLICE field */
    uint32_t set_index_alt[3];
} ffv1o_stream_ext;
/* contexts of set i of such a stream: (product over the five tables of (2 * levels - 1) + 1) / 2, FFV1_Parameters.cpp:206-219 */
uint32_t ffv1o_ext_context_count(const ffv1o_stream_ext* e, uint32_t set);

typedef struct {
    uint32_t width, height;
    uint32_t pixfmt;        /* FFV1O_* */
    uint32_t num_h_slices;  /* >= num_v_slices (reference decoder limit, FFV1_Slice.cpp:127) */
    uint32_t num_v_slices;
    uint32_t ec;            /* slicecrc: 0/1 */
    uint32_t context_model; /* -context: 0 (3 inputs) / 1 (5 inputs, FFmpeg's level maps) / 2 (5 inputs, compact 5,5,3,3,3 maps) */
    uint32_t flags;         /* FFV1O_FLAG_* (payload layout only; the bitstream does not know about them) */
    uint32_t coder;         /* -coder: 0/1 range coder with the default state transitions; 2 range coder with the alternate table, which
                               then travels in the configuration record as 255 deltas (FFV1_Parameters.cpp:41-55) */
    uint32_t level;         /* -level: 0/3 = FFV1 version 3 (configuration record, slices, footers); 1 = version 1: one slice = the frame,
                               the header travels inside every frame after the keyframe bit, no footer (FFV1_Slice.cpp:224-268) */
    const ffv1o_stream_ext* ext; /* NULL, or the stream's free-form description: then context_model, coder and level are ignored */
} ffv1o_params;

/* geometry helpers */
uint32_t ffv1o_bits_per_raw_sample(uint32_t pixfmt);
uint32_t ffv1o_plane_count(uint32_t pixfmt);        /* 1, 3 or 4 */
uint32_t ffv1o_bytes_per_pixel(uint32_t pixfmt);
/* bytes per payload line: DPX pads each line to 32 bit (RawFrame.cpp:109); TIFF does not. */
size_t   ffv1o_line_bytes(uint32_t pixfmt, uint32_t width, int dpx_line_padding);
/* bytes of a whole payload: height * line_bytes, or the continuous word stream of the "altern" layout */
size_t   ffv1o_payload_bytes(const ffv1o_params* p, size_t line_bytes);

/* FFV1 configuration record incl. CRC (inverse of FFV1_Parameters.cpp:23-183).  Returns size. */
size_t ffv1o_config_record(const ffv1o_params* p, uint8_t* out, size_t cap);

/* payload -> FFV1 planes (forward RCT + unpack; inverse of Transform.cpp). planes[i] has width*height int32. */
void ffv1o_unpack(const ffv1o_params* p, const uint8_t* payload, size_t line_bytes, int32_t* const planes[4]);
/* planes -> payload (restates Transform.cpp From()). Padding bits are written as zero. */
void ffv1o_pack(const ffv1o_params* p, int32_t* const planes[4], uint8_t* payload, size_t line_bytes);

/* One intra frame: planes -> packet.  Returns packet size, 0 on overflow.
 * slice_sizes (optional, num_h*num_v entries) receives the byte size of each slice incl. footer. */
size_t ffv1o_encode_frame(const ffv1o_params* p, int32_t* const planes[4], uint8_t* out, size_t cap,
                          uint32_t* slice_sizes);
/* One slice with its intermediates, for stage-by-stage parity tests of the device pipeline: sym_out receives one
 * word per sample in coding order (set << 30 | |ctx| << 17 | residual & 0x1FFFF), dec_out every binary decision
 * (state | bit << 8, incl. keyframe bit, slice header and the state-129 end bit), raw_out the slice bytes incl. footer.
 * Returns the slice size; *ndec the number of decisions. */
size_t ffv1o_trace_slice(const ffv1o_params* p, int32_t* const planes[4], uint32_t sx, uint32_t sy,
                         uint32_t* sym_out, uint16_t* dec_out, size_t dec_cap, size_t* ndec, uint8_t* raw_out, size_t raw_cap);
/* Convenience: unpack + encode. */
size_t ffv1o_encode_payload(const ffv1o_params* p, const uint8_t* payload, size_t line_bytes,
                            uint8_t* out, size_t cap, uint32_t* slice_sizes);
/* Packet -> planes (restates FFV1_Frame.cpp:134-228 + FFV1_Slice.cpp). 0 ok, else error code. */
int ffv1o_decode_frame(const ffv1o_params* p, const uint8_t* pkt, size_t size, int32_t* const planes[4]);
/* Convenience: decode + pack.  0 ok. */
int ffv1o_decode_payload(const ffv1o_params* p, const uint8_t* pkt, size_t size, uint8_t* payload, size_t line_bytes);

/* The decoder as the reference runs it (ffv1_frame::OutOfBand + Process): everything about the stream is read from the stream -- the
 * configuration record, or with rec_size == 0 the header inside the packet (version 0 / 1) -- up to 8 arbitrary table sets, per-slice
 * set indices, any transition table, coded initial states; p gives width, height, pixfmt, flags only.  0 ok, 20 / 21 = a valid stream
 * outside this oracle (Golomb-Rice, YUV). */
int ffv1o_decode_stream(const ffv1o_params* p, const uint8_t* rec, size_t rec_size, const uint8_t* pkt, size_t size, uint8_t* payload, size_t line_bytes);

/* Parse a configuration record back into params (width/height/pixfmt are not in the record: the
 * caller supplies them; checks that the record agrees with them).  0 ok. */
int ffv1o_parse_config_record(const uint8_t* rec, size_t size, ffv1o_params* p);

/* CRC-32, poly 0x04C11DB7, MSB first, init 0, no final xor (ZenCRC32.cpp:1097-1135). */
uint32_t ffv1o_crc32(const uint8_t* d, size_t n);

/* -slices N -> (num_h, num_v): first v in [2,31], h in [v, 2v-1] with h*v==N (FFmpeg's rule; the set
 * of N it accepts equals Project/GNU/CLI/test/slices.sh:12 valid_slices).  N==1 -> 1x1.  0 ok, -1 invalid. */
int ffv1o_slices_to_grid(uint32_t n, uint32_t* num_h, uint32_t* num_v);

/* Number of binary range-coder decisions the encoder performed for the last ffv1o_encode_frame call
 * on this thread (instrumentation for DESIGN.md's per-unit figures). */
uint64_t ffv1o_last_decisions(void);

#ifdef __cplusplus
}
#endif
#endif
