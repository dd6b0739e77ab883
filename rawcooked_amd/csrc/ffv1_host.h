// ffv1_host.h -- host-side FFV1 pieces of the encoder: bitstream constants, the configuration record,
// and the per-slice header decisions that are prepended to each slice's decision stream on the device.
#pragma once
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

}}  // namespace rc::ffv1
