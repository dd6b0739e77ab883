// ref_ffv1_parse.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REAL reference's FFV1 header readers: it is compiled against the
// headers under /root/reference and linked with the reference's own objects (oracle/Makefile.ref, target `ffv1_parse`), and prints what
// ffv1_frame::OutOfBand (FFV1_Frame.cpp:105-131), parameters::Parse (FFV1_Parameters.cpp:23-183) and slice::SliceHeader
// (FFV1_Slice.cpp:113-177) make of the bytes it is given -- the fields rcgpu_ffv1_stream_parse must read the same way.  Nothing here
// restates the reference: every decision is taken by its code; this file only feeds it and prints.
//
//   ref_ffv1_parse <cases.bin>      cases: { u32 record_size, u32 packet_size, record bytes, packet bytes } repeated (little endian)
//   one line per case:
//     ERR <where> <the reference's message>
//     OK version micro coder colorspace bits chroma log2h log2v alpha num_h num_v sets ec intra index_count custom underrun
//        T<fnv of the transition table> then per set: Q<fnv of the 5 x 256 table values> C<contexts> S<fnv of the initial states>
//        then, with a record: K<key frame bit> H<SliceHeader's verdict> I<index,index,...> E<message or ->
// tests/golden/make_parse_golden.py turns that output into the vectors of tests/golden/parse_cases.*
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <sys/resource.h>
#define private public        // ffv1_frame::E / ::P and slice::E / ::SliceHeader are not public
#define protected public
#include "Lib/CoDec/FFV1/FFV1_Frame.h"
#include "Lib/CoDec/FFV1/FFV1_Slice.h"
#undef private
#undef protected

extern const state_transitions_struct default_state_transitions;

static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

static void print_parameters(parameters& P, rangecoder& E)
{
    printf("OK %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u", P.version, P.version >= 3 ? P.micro_version : 0u, P.coder_type, P.colorspace_type, P.bits_per_raw_sample,
           unsigned(P.chroma_planes), P.log2_h_chroma_subsample, P.log2_v_chroma_subsample, unsigned(P.alpha_plane), P.num_h_slices, P.num_v_slices, P.quant_table_set_count,
           P.ec, P.intra, unsigned(P.quant_table_set_index_count), unsigned(P.RC_state_transitions_custom != nullptr), unsigned(E.IsUnderrun()));
    const state_transitions_struct& T = P.RC_state_transitions_custom ? *P.RC_state_transitions_custom : default_state_transitions;
    printf(" T%016llx", (unsigned long long)fnv(0xcbf29ce484222325ull, T.States + 1, 255));
    for (uint32_t i = 0; i < P.quant_table_set_count; i++) {
        uint64_t q = 0xcbf29ce484222325ull;
        for (int j = 0; j < MAX_CONTEXT_INPUTS; j++)
            for (int k = 0; k < MAX_QUANT_TABLE_SIZE; k++) { const int32_t v = P.QuantTableSets[i].QuantTables[j][k]; q = fnv(q, &v, 4); }
        uint64_t s = 0xcbf29ce484222325ull;
        if (P.coder_type == 1)
            for (size_t c = 0; c < P.QuantTableSets[i].Contexts_Count; c++) s = fnv(s, P.RC_ContextSets[i].RC_Contexts[c].States, states_size);
        printf(" Q%016llx C%zu S%016llx", (unsigned long long)q, P.QuantTableSets[i].Contexts_Count, (unsigned long long)s);
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_ffv1_parse <cases.bin>\n"); return 2; }
    struct rlimit lim = { size_t(3) << 30, size_t(3) << 30 };
    setrlimit(RLIMIT_AS, &lim);
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    for (;;) {
        uint32_t n[2];
        if (fread(n, 4, 2, f) != 2) break;
        std::vector<uint8_t> rec(n[0] + 8), pkt(n[1] + 8);                       // (slack: the reader looks at Buffer[0] of an empty buffer's neighbour at most)
        if ((n[0] && fread(rec.data(), 1, n[0], f) != n[0]) || (n[1] && fread(pkt.data(), 1, n[1], f) != n[1])) break;
        ffv1_frame F(nullptr);                                                    // (no slice pool: nothing here decodes samples)
        if (n[0]) {
            // the track's CodecPrivate: exactly what track_info::OutOfBand hands over
            // (OutOfBand ends by making num_h_slices x num_v_slices slice slots, whatever the record says they are: under the address-space limit set
            // in main that is a bad_alloc instead of this machine's memory)
            try { F.OutOfBand(rec.data(), n[0]); } catch (const std::bad_alloc&) { printf("ERR record bad_alloc (%u x %u slices)\n", F.P.num_h_slices, F.P.num_v_slices); continue; }
            if (F.P.Error()) { printf("ERR record %s\n", F.P.Error()); continue; }
            print_parameters(F.P, F.E);
            // the first slice of the first frame as slice::Parse opens it (FFV1_Slice.cpp:210-257): key frame bit, the stream's own transitions, SliceHeader
            F.P.width = 4096; F.P.height = 4096;
            slice S(&F.P);
            S.E.AssignBuffer(pkt.data(), n[1]);
            S.E.AssignStateTransitions(default_state_transitions);
            uint8_t State = states_default;
            const bool key = S.E.b(State);
            if (F.P.RC_state_transitions_custom) S.E.AssignStateTransitions(*F.P.RC_state_transitions_custom);
            const bool ok = S.SliceHeader();
            printf(" K%u H%u I", unsigned(key), unsigned(ok));
            for (size_t i = 0; ok && i < F.P.quant_table_set_index_count; i++) printf("%s%u", i ? "," : "", S.quant_table_set_indexes[i].Index);
            printf(" U%u E%s\n", unsigned(S.E.IsUnderrun()), F.P.Error() ? F.P.Error() : "-");
        } else {
            // versions 0 and 1: the parameters stand in the key frame itself (slice::Parse without a record, FFV1_Slice.cpp:210-231)
            rangecoder E;
            E.AssignBuffer(pkt.data(), n[1]);
            E.AssignStateTransitions(default_state_transitions);
            uint8_t State = states_default;
            const bool key = E.b(State);
            if (!key) { printf("ERR frame not a key frame\n"); continue; }
            if (F.P.Parse(E, false)) { printf("ERR frame %s\n", F.P.Error() ? F.P.Error() : "?"); continue; }
            print_parameters(F.P, E);
            printf("\n");
        }
    }
    fclose(f);
    return 0;
}
