// ffv1_host.cpp -- host-side FFV1: constants, quantisation models, configuration record, slice-header decisions.
//
// The configuration record and the slice headers are a few dozen range-coded symbols per stream / slice; they are
// produced here on the host.  Everything per-sample runs on the device (ffv1_gpu.hip).
#include "ffv1_host.h"
#include "rc_common.h"

namespace rc { namespace ffv1 {

// RFC 9043 default state transition table ("default_state_transition"); index = current state, value = next
// state after coding a 1.  States 0 and 249..255 are unreachable.
const uint8_t kOneState[256] = {
      0,   0,   0,   0,   0,   0,   0,   0,  20,  21,  22,  23,  24,  25,  26,  27,  28,  29,  30,  31,  32,  33,  34,  35,  36,  37,  37,  38,  39,  40,  41,  42,
     43,  44,  45,  46,  47,  48,  49,  50,  51,  52,  53,  54,  55,  56,  56,  57,  58,  59,  60,  61,  62,  63,  64,  65,  66,  67,  68,  69,  70,  71,  72,  73,
     74,  75,  75,  76,  77,  78,  79,  80,  81,  82,  83,  84,  85,  86,  87,  88,  89,  90,  91,  92,  93,  94,  94,  95,  96,  97,  98,  99, 100, 101, 102, 103,
    104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 126, 127, 128, 129, 130, 131, 132, 133, 133,
    134, 135, 136, 137, 138, 139, 140, 141, 142, 143, 144, 145, 146, 147, 148, 149, 150, 151, 152, 152, 153, 154, 155, 156, 157, 158, 159, 160, 161, 162, 163, 164,
    165, 166, 167, 168, 169, 170, 171, 171, 172, 173, 174, 175, 176, 177, 178, 179, 180, 181, 182, 183, 184, 185, 186, 187, 188, 189, 190, 190, 191, 192, 194, 194,
    195, 196, 197, 198, 199, 200, 201, 202, 202, 204, 205, 206, 207, 208, 209, 209, 210, 211, 212, 213, 215, 215, 216, 217, 218, 219, 220, 220, 222, 223, 224, 225,
    226, 227, 227, 229, 229, 230, 231, 232, 234, 234, 235, 236, 237, 238, 239, 240, 241, 242, 243, 244, 245, 246, 247, 248, 248,   0,   0,   0,   0,   0,   0,   0,
};

const uint8_t kOneStateAlt[256] = {
      0,  10,  10,  10,  10,  16,  16,  16,  28,  16,  16,  29,  42,  49,  20,  49,  59,  25,  26,  26,  27,  31,  33,  33,  33,  34,  34,  37,  67,  38,  39,  39,
     40,  40,  41,  79,  43,  44,  45,  45,  48,  48,  64,  50,  51,  52,  88,  52,  53,  74,  55,  57,  58,  58,  74,  60, 101,  61,  62,  84,  66,  66,  68,  69,
     87,  82,  71,  97,  73,  73,  82,  75, 111,  77,  94,  78,  87,  81,  83,  97,  85,  83,  94,  86,  99,  89,  90,  99, 111,  92,  93, 134,  95,  98, 105,  98,
    105, 110, 102, 108, 102, 118, 103, 106, 106, 113, 109, 112, 114, 112, 116, 125, 115, 116, 117, 117, 126, 119, 125, 121, 121, 123, 145, 124, 126, 131, 127, 129,
    165, 130, 132, 138, 133, 135, 145, 136, 137, 139, 146, 141, 143, 142, 144, 148, 147, 155, 151, 149, 151, 150, 152, 157, 153, 154, 156, 168, 158, 162, 161, 160,
    172, 163, 169, 164, 166, 184, 167, 170, 177, 174, 171, 173, 182, 176, 180, 178, 175, 189, 179, 181, 186, 183, 192, 185, 200, 187, 191, 188, 190, 197, 193, 196,
    197, 194, 195, 196, 198, 202, 199, 201, 210, 203, 207, 204, 205, 206, 208, 214, 209, 211, 221, 212, 213, 215, 224, 216, 217, 218, 219, 220, 222, 228, 223, 225,
    226, 224, 227, 229, 240, 230, 231, 232, 233, 234, 235, 236, 238, 239, 237, 242, 241, 243, 242, 244, 245, 246, 247, 248, 249, 250, 251, 252, 252, 253, 254, 255,
};

void make_zero_state(uint8_t zero[256], const uint8_t* one)
{
    zero[0] = 0;
    for (int i = 1; i < 256; i++) zero[i] = uint8_t(256 - one[256 - i]);
}

// Level maps over |difference| 0..127 as run lengths (the record carries exactly these runs, FFV1_Parameters.cpp:222-253).
static void fill(int16_t* q, std::initializer_list<int> runs, int scale)
{
    int k = 0, level = 0;
    for (int r : runs) { for (int a = 0; a < r; a++) q[k++] = int16_t(level * scale); level++; }
    for (int i = 1; i < 128; i++) q[256 - i] = int16_t(-q[i]);
    q[128] = int16_t(-q[127]);
}

void build_quant_models(uint32_t bps, quant_model m[2], bool compact)
{
    memset(m, 0, 2 * sizeof(quant_model));
    if (bps <= 8) {            // 11-level / 5-level maps for 8-bit material
        const std::initializer_list<int> q11 = { 1, 1, 3, 7, 20, 96 }, q5 = { 1, 3, 124 };
        fill(m[0].q[0], q11, 1); fill(m[0].q[1], q11, 11); fill(m[0].q[2], q11, 121);
        m[0].context_count = (11 * 11 * 11 + 1) / 2;
        fill(m[1].q[0], q11, 1); fill(m[1].q[1], q11, 11); fill(m[1].q[2], q5, 121); fill(m[1].q[3], q5, 605); fill(m[1].q[4], q5, 3025);
        m[1].context_count = (11 * 11 * 5 * 5 * 5 + 1) / 2;
    } else {                   // 9-level / 5-level maps for high bit depth
        const std::initializer_list<int> q9 = { 5, 8, 14, 29, 72 }, q5 = { 11, 53, 64 };
        fill(m[0].q[0], q9, 1); fill(m[0].q[1], q9, 9); fill(m[0].q[2], q9, 81);
        m[0].context_count = (9 * 9 * 9 + 1) / 2;
        fill(m[1].q[0], q9, 1); fill(m[1].q[1], q9, 9); fill(m[1].q[2], q5, 81); fill(m[1].q[3], q5, 405); fill(m[1].q[4], q5, 2025);
        m[1].context_count = (9 * 9 * 5 * 5 * 5 + 1) / 2;
    }
    if (compact) {             // 5,5,3,3,3 levels: thresholds of the 5-level map above, and one threshold for the 3-level inputs
        memset(&m[1], 0, sizeof(quant_model));
        if (bps <= 8) {
            const std::initializer_list<int> q5 = { 1, 3, 124 }, q3 = { 4, 124 };
            fill(m[1].q[0], q5, 1); fill(m[1].q[1], q5, 5); fill(m[1].q[2], q3, 25); fill(m[1].q[3], q3, 75); fill(m[1].q[4], q3, 225);
        } else {
            const std::initializer_list<int> q5 = { 11, 53, 64 }, q3 = { 24, 104 };
            fill(m[1].q[0], q5, 1); fill(m[1].q[1], q5, 5); fill(m[1].q[2], q3, 25); fill(m[1].q[3], q3, 75); fill(m[1].q[4], q3, 225);
        }
        m[1].context_count = (5 * 5 * 3 * 3 * 3 + 1) / 2;
    }
}

namespace {
// Scalar range encoder (RFC 9043 3.8.1; inverse of rangecoder::b, FFV1_RangeCoder.cpp:71-102).  With `trace` set it
// records (state | bit << 8) instead of producing bytes -- the device consumes that form.
struct host_rc {
    uint32_t low = 0, range = 0xFF00;
    int outstanding = -1, run = 0;
    std::vector<uint8_t> out;
    std::vector<uint16_t>* trace = nullptr;
    uint8_t zero[256];
    const uint8_t* one = kOneState;
    explicit host_rc(uint32_t coder = 1) : one(one_state_table(coder)) { make_zero_state(zero, one); }
    void renorm()
    {
        while (range < 0x100) {
            if (outstanding < 0) outstanding = int(low >> 8);
            else if (low <= 0xFF00) { out.push_back(uint8_t(outstanding)); for (; run; run--) out.push_back(0xFF); outstanding = int(low >> 8); }
            else if (low >= 0x10000) { out.push_back(uint8_t(outstanding + 1)); for (; run; run--) out.push_back(0x00); outstanding = int((low >> 8) & 0xFF); }
            else run++;
            low = (low & 0xFF) << 8; range <<= 8;
        }
    }
    void put(uint8_t& st, int bit)
    {
        if (trace) trace->push_back(uint16_t(st | (bit << 8)));
        else {
            const uint32_t r1 = (range * st) >> 8;
            if (bit) { low += range - r1; range = r1; } else range -= r1;
            renorm();
        }
        st = bit ? one[st] : zero[st];
    }
    void symbol(uint8_t* st, int32_t v, bool is_signed)   // inverse of rangecoder::u / ::s, FFV1_RangeCoder.cpp:105-305
    {
        if (!v) { put(st[0], 1); return; }
        const uint32_t a = uint32_t(v < 0 ? -v : v);
        int e = 31 - __builtin_clz(a);
        put(st[0], 0);
        for (int i = 0; i < e; i++) put(st[1 + (i < 9 ? i : 9)], 1);
        put(st[1 + (e < 9 ? e : 9)], 0);
        for (int i = e - 1; i >= 0; i--) put(st[22 + (i < 9 ? i : 9)], int((a >> i) & 1));
        if (is_signed) put(st[11 + (e < 10 ? e : 10)], v < 0);
    }
    void finish() { range = 0xFF; low += 0xFF; renorm(); range = 0xFF; renorm(); }   // no end bit: record only
};

void write_quant_table(host_rc& c, const int16_t* q)
{
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    int last = 0, i;
    for (i = 1; i < 128; i++)
        if (q[i] != q[i - 1]) { c.symbol(st, i - last - 1, false); last = i; }
    c.symbol(st, i - last - 1, false);
}
}  // namespace

std::vector<uint8_t> config_record(const stream_params& p)
{
    if (p.version == 1) return {};                  // version 1: no out-of-band record
    quant_model m[2];
    build_quant_models(p.bits_per_raw_sample, m, p.compact);
    host_rc c;
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    c.symbol(st, 3, false);                         // version
    c.symbol(st, 4, false);                         // micro_version
    c.symbol(st, p.coder == 2 ? 2 : 1, false);      // coder_type: range coder with the default (-coder 1) or a transmitted (-coder 2) table
    if (p.coder == 2)                               // state_transition_delta[1..255]; the record itself is coded with the default table
        for (int i = 1; i < 256; i++) c.symbol(st, int32_t(kOneStateAlt[i]) - int32_t(kOneState[i]), true);
    c.symbol(st, p.rgb ? 1 : 0, false);             // colorspace_type
    c.symbol(st, int32_t(p.bits_per_raw_sample), false);
    c.put(st[0], p.rgb ? 1 : 0);                    // chroma_planes
    c.symbol(st, 0, false); c.symbol(st, 0, false); // log2 chroma subsampling
    c.put(st[0], p.alpha ? 1 : 0);                  // alpha_plane
    c.symbol(st, int32_t(p.num_h_slices) - 1, false);
    c.symbol(st, int32_t(p.num_v_slices) - 1, false);
    c.symbol(st, 2, false);                         // quant_table_set_count
    for (int i = 0; i < 2; i++) for (int j = 0; j < 5; j++) write_quant_table(c, m[i].q[j]);
    for (int i = 0; i < 2; i++) c.put(st[0], 0);    // states_coded
    c.symbol(st, int32_t(p.ec), false);
    c.symbol(st, 1, false);                         // intra (-g 1)
    c.finish();
    const uint32_t crc = rcgpu_crc32_ffv1(c.out.data(), c.out.size());
    for (int s = 24; s >= 0; s -= 8) c.out.push_back(uint8_t(crc >> s));   // parity: CRC(record || crc) == 0 (FFV1_Frame.cpp:116)
    return c.out;
}

std::vector<uint16_t> v1_frame_header_decisions(const stream_params& p)
{
    quant_model m[2];
    build_quant_models(p.bits_per_raw_sample, m, p.compact);
    std::vector<uint16_t> d;
    host_rc c(1); c.trace = &d;                     // the header is read with the default transitions whatever coder_type says
    { uint8_t ks = 128; c.put(ks, 1); }             // keyframe
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    c.symbol(st, 1, false);                         // version
    c.symbol(st, p.coder == 2 ? 2 : 1, false);      // coder_type
    if (p.coder == 2)
        for (int i = 1; i < 256; i++) c.symbol(st, int32_t(kOneStateAlt[i]) - int32_t(kOneState[i]), true);
    c.symbol(st, p.rgb ? 1 : 0, false);             // colorspace_type
    c.symbol(st, int32_t(p.bits_per_raw_sample), false);
    c.put(st[0], p.rgb ? 1 : 0);                    // chroma_planes
    c.symbol(st, 0, false); c.symbol(st, 0, false); // chroma subsampling
    c.put(st[0], p.alpha ? 1 : 0);                  // alpha_plane
    for (int j = 0; j < 5; j++) write_quant_table(c, m[p.context_model].q[j]);     // the one table set
    return d;
}

std::vector<uint16_t> slice_header_decisions(const stream_params& p, uint32_t sx, uint32_t sy, bool first_slice)
{
    std::vector<uint16_t> d;
    host_rc c(p.coder); c.trace = &d;
    if (first_slice) { uint8_t ks = 128; c.put(ks, 1); }           // keyframe
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    c.symbol(st, int32_t(sx), false);
    c.symbol(st, int32_t(sy), false);
    c.symbol(st, 0, false); c.symbol(st, 0, false);                 // slice_width-1, slice_height-1 in slice units
    const uint32_t index_count = p.rgb ? (p.alpha ? 3u : 2u) : 2u;  // quant_table_set_index_count, FFV1_Parameters.cpp:164-178
    for (uint32_t i = 0; i < index_count; i++) c.symbol(st, int32_t(p.context_model), false);
    c.symbol(st, 3, false);                                         // picture_structure: progressive
    c.symbol(st, 0, false); c.symbol(st, 0, false);                 // sar_num / sar_den unknown
    return d;
}

}}  // namespace rc::ffv1

// ---- the decode side's reader (parameters::Parse, FFV1_Parameters.cpp:23-183; slice::SliceHeader, FFV1_Slice.cpp:113-177): a scalar range
// decoder over a few hundred bytes -- or, with coded initial states, up to a million symbols once per stream
namespace {
struct host_rd {
    const uint8_t* buf; size_t pos, size; uint32_t current, mask; uint8_t zero[256]; const uint8_t* one = rc::ffv1::kOneState;
    std::vector<uint16_t>* trace = nullptr;          // every decision as state | bit << 8
    host_rd(const uint8_t* p, size_t n) : buf(p), pos(1), size(n) { current = n ? p[0] : 0; mask = 0xFF; rc::ffv1::make_zero_state(zero); }
    void transitions(const uint8_t* t) { one = t; rc::ffv1::make_zero_state(zero, t); }          // AssignStateTransitions, FFV1_RangeCoder.cpp:35-41
    bool underrun() const { return pos - (mask < 0x100 ? 0 : 1) > size; }                          // IsUnderrun, :59-62
    void force_underrun() { mask = 0; pos = size + 1; }                                            // ForceUnderrun, :308-312: an exponent beyond 31 condemns the stream
    bool bit(uint8_t& st)
    {
        if (mask < 0x100) {                                                                        // :74-88: past the end every decision is 0 and nothing moves
            current <<= 8;
            if (pos > size) return false;
            if (pos < size) current |= buf[pos];
            mask <<= 8; pos++;
        }
        const uint32_t m2 = (mask * st) >> 8;
        mask -= m2;
        const bool b = current >= mask;
        if (trace) trace->push_back(uint16_t(st | (b ? 0x100 : 0)));
        if (!b) { st = zero[st]; return false; }
        current -= mask; mask = m2; st = one[st];
        return true;
    }
    uint32_t u(uint8_t* st)
    {
        if (bit(st[0])) return 0;
        int e = 0;
        while (bit(st[1 + (e < 9 ? e : 9)])) if (++e > 31) { force_underrun(); return 0; }
        uint32_t a = 1;
        for (int i = e - 1; i >= 0; i--) a = (a << 1) | uint32_t(bit(st[22 + (i < 9 ? i : 9)]));
        return a;
    }
    int32_t s(uint8_t* st)
    {
        if (bit(st[0])) return 0;
        int e = 0;
        while (bit(st[1 + (e < 9 ? e : 9)])) if (++e > 31) { force_underrun(); return 0; }
        int32_t a = 1;
        for (int i = e - 1; i >= 0; i--) a = (a << 1) | int32_t(bit(st[22 + (i < 9 ? i : 9)]));
        return bit(st[11 + (e < 10 ? e : 10)]) ? -a : a;
    }
};

// parameters::Parse(E, ConfigurationRecord_IsPresent) with QuantizationTableSet / QuantizationTable (:206-253), field for field.  The
// errors are the reference's (its message in the text); rc::ffv1::kUnsupported marks what the reference decodes and the device does not.
int parse_parameters(host_rd& r, rc::ffv1::stream_desc& s, bool record)
{
    using namespace rc; using namespace rc::ffv1;
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    s.version = r.u(st);
    if (record ? s.version <= 1 : s.version > 1) return fail(4, "ffv1 stream: version %u %s a configuration record (FFV1-HEADER-version-OUTOFBAND)", s.version, record ? "with" : "without");
    if (s.version == 2 || s.version > 3) return fail(4, "ffv1 stream: version %u (FFV1_Parameters.cpp:33-34)", s.version);
    s.micro_version = s.version >= 3 ? r.u(st) : 0;
    if (s.version == 3 && s.micro_version < 4) return fail(4, "ffv1 stream: micro_version %u < 4 (FFV1_Parameters.cpp:36-37)", s.micro_version);
    uint32_t coder_type = r.u(st);
    if (coder_type > 2) return fail(4, "ffv1 stream: coder_type %u (FFV1_Parameters.cpp:39-40)", coder_type);
    s.custom_transitions = false;
    memcpy(s.one_state, kOneState, 256);
    if (coder_type == 2) {                                   // state_transition_delta[1..255], :41-55
        for (int i = 1; i < 256; i++) {
            const int32_t v = int32_t(kOneState[i]) + r.s(st);
            if (v < 0 || v > 0xFF) return fail(4, "ffv1 stream: state_transition_delta out of range (FFV1_Parameters.cpp:49-50)");
            s.one_state[i] = uint8_t(v);
        }
        s.custom_transitions = true;
        coder_type = 1;
    }
    if (coder_type != 1) return fail(kUnsupported, "ffv1 stream: Golomb-Rice coded (coder_type 0): not decoded on the device");
    s.colorspace_type = r.u(st);
    if (s.colorspace_type > 1) return fail(4, "ffv1 stream: colorspace_type %u (FFV1_Parameters.cpp:59-60)", s.colorspace_type);
    if (s.version) {
        s.bits_per_raw_sample = r.u(st);
        if (s.bits_per_raw_sample > 64) return fail(4, "ffv1 stream: bits_per_raw_sample %u (FFV1_Parameters.cpp:64-65)", s.bits_per_raw_sample);
        if (!s.bits_per_raw_sample) s.bits_per_raw_sample = 8;
    } else s.bits_per_raw_sample = 8;
    s.chroma_planes = r.bit(st[0]);
    s.log2_h_chroma_subsample = r.u(st); s.log2_v_chroma_subsample = r.u(st);
    s.alpha_plane = r.bit(st[0]);
    if (s.version > 1) {
        s.num_h_slices = r.u(st) + 1; s.num_v_slices = r.u(st) + 1;
        s.set_count = r.u(st);
        if (s.set_count > 8) return fail(6, "ffv1 stream: %u quantisation table sets (FFV1_Parameters.cpp:84-85)", s.set_count);
    } else { s.num_h_slices = s.num_v_slices = 1; s.set_count = 1; }
    for (uint32_t i = 0; i < s.set_count; i++) {
        int64_t scale = 1;                                   // contexts so far; the reference stops at 32768
        memset(&s.sets[i], 0, sizeof s.sets[i]);
        for (int j = 0; j < 5; j++) {
            uint8_t qst[kContextSize]; memset(qst, 128, sizeof qst);
            int16_t* q = s.sets[i].q[j];
            int32_t v = 0;
            for (uint64_t k = 0; k < 128;) {                 // 64 bits as in the reference (size_t): len_minus1 reaches 2^32 - 1
                const uint32_t len1 = r.u(qst);
                if (k + len1 >= 128) return fail(6, "ffv1 stream: bad quantisation table (FFV1_Parameters.cpp:231-232)");
                for (uint32_t a = 0; a <= len1; a++, k++) q[k] = int16_t(scale * v);
                v++;
            }
            for (int k = 1; k < 128; k++) q[256 - k] = int16_t(-q[k]);
            q[128] = int16_t(-q[127]);
            scale *= 2 * v - 1;
            if (scale > 32768) return fail(6, "ffv1 stream: more than 32768 contexts (FFV1_Parameters.cpp:247-248)");
        }
        s.sets[i].context_count = uint32_t((scale + 1) >> 1);
    }
    for (uint32_t i = 0; i < s.set_count; i++) {
        s.initial[i].clear();
        if (s.version >= 3 && r.bit(st[0])) {                // states_coded: `States[k] = E.s(States)` (:103-107), every value as it stands
            s.initial[i].resize(size_t(s.sets[i].context_count) * kContextSize);
            size_t n = 0;
            for (uint8_t& v : s.initial[i]) {
                v = uint8_t(r.s(st));
                if (!(++n & 1023) && r.underrun()) return fail(3, "ffv1 stream: the parameters end inside the coded initial states");
            }
        }
    }
    if (s.version >= 3) {
        s.ec = r.u(st);
        if (s.ec > 1) return fail(7, "ffv1 stream: ec %u (FFV1_Parameters.cpp:137)", s.ec);
        s.intra = 0;
        if (s.micro_version) { s.intra = r.u(st); if (s.intra > 1) return fail(7, "ffv1 stream: intra %u (FFV1_Parameters.cpp:142-143)", s.intra); }
    } else { s.ec = 0; s.intra = 0; }
    if (r.underrun()) return fail(3, "ffv1 stream: the parameters end before they are complete");
    // quant_table_set_index_count, :164-178
    s.index_count = s.colorspace_type == 1 ? (s.alpha_plane ? 3u : 2u) : 2u + (s.alpha_plane ? 1u : 0u);
    return 0;
}
}  // namespace

namespace rc { namespace ffv1 {

int parse_stream(const uint8_t* rec, size_t rec_size, const uint8_t* packet, size_t packet_size, stream_desc& s)
{
    if (rec_size) {                                          // ffv1_frame::OutOfBand, FFV1_Frame.cpp:105-131
        if (!rec) return fail(1, "ffv1 stream: null argument");
        if (rec_size < 5 || rcgpu_crc32_ffv1(rec, rec_size)) return fail(3, "ffv1 record: CRC mismatch (FFV1_Frame.cpp:116)");
        host_rd r(rec, rec_size - 4);
        if (int e = parse_parameters(r, s, true)) return e;
    }
    if (!packet || packet_size < (rec_size ? 8u : 2u)) return fail(8, "ffv1 stream: the first packet of the track is needed (the %s)", rec_size ? "quantisation table set of every plane stands in its slice headers" : "stream's header travels inside it");
    host_rd r(packet, packet_size);
    s.inband.clear();
    if (!rec_size) r.trace = &s.inband;
    { uint8_t ks = 128; if (!r.bit(ks)) return fail(8, "ffv1 stream: the first frame is not a key frame (FFV1-FRAME-key_frame-NOINFIRSTFRAME)"); }
    if (!rec_size) {                                         // slice::Parse without a record, FFV1_Slice.cpp:224-245: the parameters, then the samples
        if (int e = parse_parameters(r, s, false)) return e;
        r.trace = nullptr;
        s.set_index[0] = s.set_index[1] = s.set_index[2] = 0;
        return 0;
    }
    // The reference compares slice_y with num_h_slices (FFV1_Slice.cpp:125): in a stream with more slice rows than columns it reports
    // FFV1-SLICE-slice_xywh for every slice below row num_h_slices and decodes it where the slice before it lay.  No encoder RAWcooked drives
    // writes such a grid (FFmpeg picks columns >= rows); one that has it keeps the reference's decoder and the reference's verdict.
    if (s.num_v_slices > s.num_h_slices) return fail(kUnsupported, "ffv1 stream: %u x %u slices: more rows than columns, which the reference itself misreads (FFV1_Slice.cpp:125): left to it", s.num_h_slices, s.num_v_slices);
    if (s.custom_transitions) r.transitions(s.one_state);    // FFV1_Slice.cpp:254-255
    uint8_t st[kContextSize]; memset(st, 128, sizeof st);
    // slice::SliceHeader's tests, in its order and its arithmetic (FFV1_Slice.cpp:117-147): 32-bit sums that may wrap, and slice_y against
    // num_H_slices.  What gets through here and is still nonsense meets the device's own comparison of every slice header with its place.
    const uint32_t sx = r.u(st);
    bool bad = sx >= s.num_h_slices;
    const uint32_t sy = bad ? 0 : r.u(st);
    bad = bad || sy >= s.num_h_slices;
    const uint32_t sw1 = bad ? 0 : r.u(st);
    bad = bad || uint32_t(sx + sw1 + 1) > s.num_h_slices;
    const uint32_t sh1 = bad ? 0 : r.u(st);
    bad = bad || uint32_t(sy + sh1 + 1) > s.num_v_slices;
    if (bad) return fail(8, "ffv1 stream: slice geometry of the first slice header (FFV1-SLICE-slice_xywh)");
    for (uint32_t i = 0; i < s.index_count; i++) {
        s.set_index[i] = r.u(st);
        if (s.set_index[i] >= s.set_count) return fail(8, "ffv1 stream: quant_table_set_index %u of %u sets (FFV1_Slice.cpp:162-167)", s.set_index[i], s.set_count);
    }
    if (r.underrun()) return fail(8, "ffv1 stream: the first packet ends inside its first slice header");
    return 0;
}

void stream_of_encoder(const stream_params& p, stream_desc& s)
{
    quant_model m[2];
    build_quant_models(p.bits_per_raw_sample, m, p.compact);
    s = stream_desc();
    s.version = p.version == 1 ? 1 : 3; s.micro_version = s.version == 3 ? 4 : 0;
    s.custom_transitions = p.coder == 2;
    memcpy(s.one_state, one_state_table(p.coder), 256);
    s.colorspace_type = p.rgb ? 1 : 0; s.bits_per_raw_sample = p.bits_per_raw_sample; s.chroma_planes = p.rgb; s.alpha_plane = p.alpha;
    s.num_h_slices = p.num_h_slices; s.num_v_slices = p.num_v_slices; s.ec = p.ec; s.intra = s.version == 3 ? 1 : 0;
    s.index_count = p.rgb ? (p.alpha ? 3u : 2u) : 2u;
    if (s.version == 1) { s.set_count = 1; s.sets[0] = m[p.context_model]; s.inband = v1_frame_header_decisions(p); }
    else { s.set_count = 2; s.sets[0] = m[0]; s.sets[1] = m[1]; s.set_index[0] = s.set_index[1] = s.set_index[2] = p.context_model; }
}

bool reaches_state_zero(const stream_desc& s)
{
    uint8_t zero[256];
    make_zero_state(zero, s.one_state);
    bool seen[256] = {};
    std::vector<uint8_t> todo;
    auto visit = [&](uint8_t v) { if (!seen[v]) { seen[v] = true; todo.push_back(v); } };
    visit(128);
    const uint32_t groups = s.colorspace_type == 1 ? s.index_count : 1u;           // gray: only the luma group has samples
    for (uint32_t g = 0; g < groups; g++) for (uint8_t v : s.initial[s.set_index[g]]) visit(v);
    while (!todo.empty()) { const uint8_t v = todo.back(); todo.pop_back(); if (!v) return true; visit(s.one_state[v]); visit(zero[v]); }
    return false;
}

}}  // namespace rc::ffv1

extern "C" int rcgpu_ffv1_stream_parse(const uint8_t* record, size_t record_size, const uint8_t* packet, size_t packet_size, rcgpu_ffv1_stream** out)
{
    using namespace rc;
    clear_error();
    if (!out) return fail(1, "ffv1 stream: null argument");
    *out = nullptr;
    rcgpu_ffv1_stream* s = new rcgpu_ffv1_stream;
    if (const int e = ffv1::parse_stream(record, record_size, packet, packet_size, s->d)) { delete s; return e; }
    *out = s;
    return 0;
}

extern "C" void rcgpu_ffv1_stream_free(rcgpu_ffv1_stream* s) { delete s; }

extern "C" int rcgpu_ffv1_stream_get_info(const rcgpu_ffv1_stream* s, rcgpu_ffv1_stream_info* info)
{
    using namespace rc;
    clear_error();
    if (!s || !info) return fail(1, "ffv1 stream: null argument");
    const ffv1::stream_desc& d = s->d;
    memset(info, 0, sizeof *info);
    info->version = d.version; info->micro_version = d.micro_version; info->coder_type = d.custom_transitions ? 2 : 1;
    info->colorspace_type = d.colorspace_type; info->bits_per_raw_sample = d.bits_per_raw_sample;
    info->chroma_planes = d.chroma_planes; info->alpha_plane = d.alpha_plane;
    info->num_h_slices = d.num_h_slices; info->num_v_slices = d.num_v_slices; info->quant_table_set_count = d.set_count;
    info->ec = d.ec; info->intra = d.intra; info->quant_table_set_index_count = d.index_count;
    for (int g = 0; g < 3; g++) info->quant_table_set_index[g] = d.set_index[g];
    for (uint32_t i = 0; i < d.set_count; i++) { info->context_count[i] = d.sets[i].context_count; info->states_coded[i] = !d.initial[i].empty(); }
    return 0;
}

extern "C" int rcgpu_ffv1_stream_get_tables(const rcgpu_ffv1_stream* s, uint32_t set, uint8_t* state_transitions, int16_t* quant_tables,
                                            uint8_t* initial_states, uint64_t initial_capacity, uint64_t* initial_size)
{
    using namespace rc;
    clear_error();
    if (!s) return fail(1, "ffv1 stream: null argument");
    const ffv1::stream_desc& d = s->d;
    if (set >= d.set_count) return fail(2, "ffv1 stream: table set %u of %u", set, d.set_count);
    if (state_transitions) memcpy(state_transitions, d.one_state, 256);
    if (quant_tables) memcpy(quant_tables, d.sets[set].q, sizeof d.sets[set].q);
    if (initial_size) *initial_size = d.initial[set].size();
    if (initial_states && !d.initial[set].empty()) {
        if (initial_capacity < d.initial[set].size()) return fail(2, "ffv1 stream: %zu bytes of initial states, room for %llu", d.initial[set].size(), (unsigned long long)initial_capacity);
        memcpy(initial_states, d.initial[set].data(), d.initial[set].size());
    }
    return 0;
}

// The older, narrower form: the stream must be one this library's encoder writes (FFmpeg's two default table sets or the compact
// model, all planes on one of them, one of the two known transition tables), and `cfg` then describes it to rcgpu_ffv1_decoder_create.
// Anything else parameters::Parse accepts goes through rcgpu_ffv1_stream_parse + rcgpu_ffv1_decoder_create_for_stream.
static int config_from_desc(const rc::ffv1::stream_desc& s, rcgpu_ffv1_config* cfg)
{
    using namespace rc; using namespace rc::ffv1;
    const pix_desc& d = pix(cfg->pixfmt);
    const bool rgb = d.planes != 1;
    if (s.version != 3) return fail(4, "ffv1 record: only version 3 is described by a configuration (rcgpu_ffv1_stream_parse takes the others)");
    if (s.colorspace_type != (rgb ? 1u : 0u) || s.bits_per_raw_sample != d.bits || s.chroma_planes != rgb || s.log2_h_chroma_subsample || s.log2_v_chroma_subsample || s.alpha_plane != (d.planes == 4))
        return fail(5, "ffv1 record: stream (colorspace %u, %u bit%s) does not match the pixel format of the files", s.colorspace_type, s.bits_per_raw_sample, s.alpha_plane ? ", alpha" : "");
    if (s.num_h_slices > cfg->width || s.num_v_slices > cfg->height || s.num_h_slices > 0x10000 || s.num_v_slices > 0x10000)
        return fail(5, "ffv1 record: %u x %u slices do not fit the picture", s.num_h_slices, s.num_v_slices);
    if (s.custom_transitions && memcmp(s.one_state + 1, kOneStateAlt + 1, 255)) return fail(kUnsupported, "ffv1 record: a transition table of the stream's own (rcgpu_ffv1_stream_parse takes it)");
    if (s.set_count != 2) return fail(kUnsupported, "ffv1 record: %u quantisation table sets (rcgpu_ffv1_stream_parse takes them)", s.set_count);
    quant_model ref[2], compact[2];
    build_quant_models(d.bits, ref, false); build_quant_models(d.bits, compact, true);
    const bool is_ref = !memcmp(s.sets[0].q, ref[0].q, sizeof ref[0].q) && !memcmp(s.sets[1].q, ref[1].q, sizeof ref[1].q);
    const bool is_compact = !memcmp(s.sets[0].q, compact[0].q, sizeof ref[0].q) && !memcmp(s.sets[1].q, compact[1].q, sizeof ref[1].q);
    if (!is_ref && !is_compact) return fail(kUnsupported, "ffv1 record: quantisation tables other than this encoder's two models (rcgpu_ffv1_stream_parse takes them)");
    if (!s.initial[0].empty() || !s.initial[1].empty()) return fail(kUnsupported, "ffv1 record: coded initial states (rcgpu_ffv1_stream_parse takes them)");
    if (s.intra != 1) return fail(kUnsupported, "ffv1 record: inter frames (intra = 0) are not decoded on the device");
    cfg->num_h_slices = s.num_h_slices; cfg->num_v_slices = s.num_v_slices; cfg->slicecrc = s.ec; cfg->coder = s.custom_transitions ? 2 : 1;
    // which of the two table sets the planes use is a per-slice field; this encoder always picks set 1 for -context 1 / compact,
    // set 0 for -context 0 -- the caller keeps what it asked for unless the tables say "compact"
    if (is_compact && !is_ref) cfg->context = 2;
    else if (cfg->context == 2) cfg->context = 1;
    return 0;
}

extern "C" int rcgpu_ffv1_config_from_record(const uint8_t* rec, size_t size, rcgpu_ffv1_config* cfg)
{
    using namespace rc; using namespace rc::ffv1;
    clear_error();
    if (!rec || !cfg) return fail(1, "ffv1 record: null argument");
    if (cfg->pixfmt >= RCGPU_PIX_COUNT) return fail(2, "ffv1 record: unknown pixel format %u", cfg->pixfmt);
    if (size < 5 || rcgpu_crc32_ffv1(rec, size)) return fail(3, "ffv1 record: CRC mismatch (FFV1_Frame.cpp:116)");
    stream_desc s;
    host_rd r(rec, size - 4);
    if (const int e = parse_parameters(r, s, true)) return e;
    return config_from_desc(s, cfg);
}

// The record says which table sets exist; which one the planes USE stands in every slice header (quant_table_set_index,
// FFV1_Slice.cpp:159-168).  A decoder-side caller that holds the first packet of the stream settles it here.
extern "C" int rcgpu_ffv1_config_from_stream(const uint8_t* rec, size_t size, const uint8_t* packet, size_t packet_size, rcgpu_ffv1_config* cfg)
{
    using namespace rc; using namespace rc::ffv1;
    clear_error();
    if (!rec || !cfg) return fail(1, "ffv1 record: null argument");
    if (cfg->pixfmt >= RCGPU_PIX_COUNT) return fail(2, "ffv1 record: unknown pixel format %u", cfg->pixfmt);
    stream_desc s;
    if (const int e = parse_stream(rec, size, packet, packet_size, s)) return e;
    if (const int e = config_from_desc(s, cfg)) return e;
    const uint32_t groups = s.colorspace_type == 1 ? s.index_count : 2u;
    for (uint32_t i = 1; i < groups; i++) if (s.set_index[i] != s.set_index[0]) return fail(kUnsupported, "ffv1 stream: planes with different quantisation table sets (rcgpu_ffv1_stream_parse takes them)");
    if (s.set_index[0] == 0) cfg->context = 0;
    else if (cfg->context != 2) cfg->context = 1;
    return 0;
}
