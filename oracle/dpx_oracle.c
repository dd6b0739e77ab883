/* dpx_oracle.c -- TEST INFRASTRUCTURE ONLY (imported by tests/; never linked into the product).
 *
 * Scalar restatement of the padding-bit test at the end of dpx::ParseBuffer, /root/reference/Source/Lib/Uncompressed/DPX/DPX.cpp:
 * 501-608: the walk over the payload, the per-unit byte mask of the FilledA / FilledB layouts (:523-533), the end-of-line word of
 * the packed layouts (:507-521) and of Y 10-bit (:535-566), and In_FirstNonZero = min(i, EOL_i) (:582-591).  It is written from the
 * header facts the reference uses (bit depth, component count, endianness, packing, Altern), not from this repository's layout table,
 * so that the two can disagree.  Parity PINNED: tests/golden/padding_vectors.json holds In_FirstNonZero as the real parser computed
 * it for 684 files (every layout with filler bits or line padding, clean and with bits poked into samples, filler bits, line padding
 * and the last word) -- oracle/ref_padding_probe.cpp drives the reference's own dpx::ParseBuffer and prints the private member;
 * tests/test_oracle.py::test_padding_oracle_matches_the_reference_parser compares this function with it, and
 * tests/test_gpu_check.py compares rcgpu_dpx_padding_scan_device with both.
 */
#include <stddef.h>
#include <stdint.h>

/* packing: 0 = Packed, 1 = FilledA, 2 = FilledB (of the canonical flavor, DPX.cpp:184-207).  Returns the offset inside the payload,
 * or UINT64_MAX when all padding bits are zero. */
uint64_t dpxo_padding_first_nonzero(const uint8_t* payload, uint64_t size, uint32_t width, uint32_t height, uint32_t bit_depth, uint32_t components,
                                    int big_endian, int packing, int altern)
{
    const uint64_t OffsetAfterData = size;
    uint64_t i, EOL_i, Step, EOL_Step = 0;
    uint32_t Mask = 0, EOL_Mask = 0;
    if (packing == 0) {
        const uint64_t UsedBits = (uint64_t)width * bit_depth * components;
        const uint64_t RemainingPaddingBits = UsedBits % 32;
        if (RemainingPaddingBits) {
            const uint64_t BytesPerLineMinus4 = (UsedBits / 32) * 4;
            i = EOL_i = BytesPerLineMinus4;
            Step = EOL_Step = BytesPerLineMinus4 + 4;
            EOL_Mask = ((uint32_t)-1) << RemainingPaddingBits;
        } else {
            return UINT64_MAX;                    /* i = OffsetAfterData: no padding */
        }
    } else {
        const int IsFilledB = packing == 2;
        i = 0;
        Step = bit_depth == 10 ? 4 : 2;
        if ((big_endian != 0) ^ IsFilledB) i += Step - 1;
        Mask = bit_depth == 10 ? 0x3 : 0xF;
        if (IsFilledB) Mask <<= bit_depth == 10 ? 6 : 4;
        if (components == 1 && bit_depth == 10) {             /* Raw_Y_10_FilledA_BE / Raw_Y_10_FilledB_BE */
            uint64_t EOL_RemainingPaddingBits = altern ? ((uint64_t)width * height) % 3 : width % 3;
            if (EOL_RemainingPaddingBits) {
                if (altern) { EOL_i = OffsetAfterData - 4; EOL_Step = 4; }
                else { const uint64_t BytesPerLineMinus4 = (width / 3) * 4; EOL_i = BytesPerLineMinus4; EOL_Step = BytesPerLineMinus4 + 4; }
                EOL_RemainingPaddingBits *= 10;
                if (!IsFilledB) EOL_RemainingPaddingBits += 2;
                EOL_Mask = ((uint32_t)-1) << EOL_RemainingPaddingBits;
                if (!IsFilledB) EOL_Mask |= 0x3;
            } else
                EOL_i = OffsetAfterData;
        } else
            EOL_i = OffsetAfterData;
    }
    for (; i < OffsetAfterData; i += Step) {
        if (i >= EOL_i) {
            const uint32_t w = ((uint32_t)payload[EOL_i] << 24) | ((uint32_t)payload[EOL_i + 1] << 16) | ((uint32_t)payload[EOL_i + 2] << 8) | payload[EOL_i + 3];   /* ntoh */
            if (w & EOL_Mask) break;
            EOL_i += EOL_Step;
        } else if (payload[i] & Mask)
            break;
    }
    if (i < OffsetAfterData) return i < EOL_i ? i : EOL_i;
    return UINT64_MAX;
}
