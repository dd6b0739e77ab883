// ffv1_host.h -- host-side FFV1 pieces of the encoder: bitstream constants, the configuration record,
// and the per-slice header decisions that are prepended to each slice's decision stream on the device.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rc { namespace ffv1 {

constexpr int kContextSize = 32;      // states per context (Lib/CoDec/FFV1/FFV1_RangeCoder.h:23)

// FFV1 default state transition table (bitstream constant; the decoder's copy is FFV1_Frame.cpp:35-55) and
// its mirror zero_state[i] = 256 - one_state[256 - i] (FFV1_RangeCoder.cpp:35-41).
extern const uint8_t kOneState[256];
// The table of `-coder 2` (coder_type 2: the transitions travel in the configuration record as deltas to the default table,
// FFV1_Parameters.cpp:41-55).  FFmpeg's choice ("ver2_state"), restated from memory: any table is conformant.
extern const uint8_t kOneStateAlt[256];
inline const uint8_t* one_state_table(uint32_t coder) { return coder == 2 ? kOneStateAlt : kOneState; }
void make_zero_state(uint8_t zero[256], const uint8_t* one = kOneState);

struct quant_model {
    int16_t  q[5][256];           // value = level * scale, negative half mirrored (FFV1_Parameters.cpp:243-245)
    uint32_t context_count;       // (scale_final + 1) / 2
};
// The two table sets carried in the configuration record: [0] 3-input model, [1] 5-input model
// (level maps as FFmpeg's ffv1enc chooses them for <= 8 bit and > 8 bit material).
// compact = true replaces [1] by a 5-input model with 5,5,3,3,3 levels (338 contexts): small enough for a slice's adaptive
// states to live in LDS, and -- on the 10^5-sample planes of a 64..576-slice 4K frame -- a better fit than 5063 contexts.
void build_quant_models(uint32_t bits_per_raw_sample, quant_model out[2], bool compact = false);

struct stream_params {
    uint32_t bits_per_raw_sample;
    bool     rgb;                 // colorspace_type 1 (JPEG 2000 RCT) vs 0 (gray)
    bool     alpha;
    uint32_t num_h_slices, num_v_slices;
    uint32_t ec;                  // slicecrc
    uint32_t context_model;       // quant table set index used by every plane (-context)
    bool     compact = false;     // table set 1 is the compact 5-input model
    uint32_t coder = 1;           // 1: default state transitions; 2: kOneStateAlt, carried in the record
    uint32_t version = 3;         // 3 (-level 3) or 1 (-level 1: one slice, header inside every frame, no record, no footer)
};

// Configuration record incl. CRC (what parameters::Parse reads, FFV1_Parameters.cpp:23-183).
std::vector<uint8_t> config_record(const stream_params& p);

// (state | bit << 8) decisions of [keyframe bit ||] slice header (FFV1_Frame.cpp:148-156, FFV1_Slice.cpp:113-177)
// for slice (sx, sy); the adaptive states involved are private to the header, so the host can resolve them.
std::vector<uint16_t> slice_header_decisions(const stream_params& p, uint32_t sx, uint32_t sy, bool first_slice);
// version 1: (state | bit << 8) decisions of keyframe bit || stream header (parameters::Parse(E, false), FFV1_Parameters.cpp:23-104,
// read with the default transitions, FFV1_Slice.cpp:214) -- there is no slice header.
std::vector<uint16_t> v1_frame_header_decisions(const stream_params& p);

// ---- the decode side: what a stream says about itself.  parameters (FFV1_Parameters.h; parameters::Parse, FFV1_Parameters.cpp:23-183)
// plus the one fact that stands in the slice headers, the quant_table_set_index tuple (FFV1_Slice.cpp:158-168) -- read from the first
// slice of the first packet; the device insists on it in every slice.
struct stream_desc {
    uint32_t version = 3, micro_version = 4;        // 0 / 1: the header travels inside every frame, one slice, no footer; 3: configuration record
    bool     custom_transitions = false;            // coder_type 2
    uint8_t  one_state[256];                        // the transitions the slices are coded with
    uint32_t colorspace_type = 0, bits_per_raw_sample = 8;
    bool     chroma_planes = false, alpha_plane = false;
    uint32_t log2_h_chroma_subsample = 0, log2_v_chroma_subsample = 0;
    uint32_t num_h_slices = 1, num_v_slices = 1, ec = 0, intra = 0;
    uint32_t set_count = 0;                         // quant_table_set_count, 1..8
    quant_model sets[8];
    std::vector<uint8_t> initial[8];                // states_coded: context_count x 32 states as the reference reads them (:103-107); empty = all 128
    uint32_t index_count = 0, set_index[3] = { 0, 0, 0 };   // plane group 0 (Y), 1 (Cb, Cr), 2 (alpha)
    std::vector<uint16_t> inband;                   // version 0 / 1: keyframe bit + header of a frame as (state | bit << 8) decisions
};
// rec_size == 0: no configuration record, the packet carries the header (version 0 / 1).  0 ok; else rc::fail()'s code with its text set:
// kUnsupported = a valid stream outside the device decoder (the caller's own decoder takes it), anything else = what the reference refuses too.
constexpr int kUnsupported = 20;
int parse_stream(const uint8_t* rec, size_t rec_size, const uint8_t* packet, size_t packet_size, stream_desc& out);
// the stream this library's encoder writes for a configuration (two table sets, all planes on one of them)
void stream_of_encoder(const stream_params& p, stream_desc& out);
// can state 0 be reached from the initial states of the sets in use?  (one_state[0] of a transmitted table is never set in the reference,
// FFV1_Parameters.cpp:43-53: what such a stream decodes to is not defined there)
bool reaches_state_zero(const stream_desc& s);

}}  // namespace rc::ffv1

struct rcgpu_ffv1_stream { rc::ffv1::stream_desc d; };
