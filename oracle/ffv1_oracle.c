/* ffv1_oracle.c -- CPU oracle for the FFV1 intra path (version 3 as FFmpeg writes it for RAWcooked; versions 0 / 1 and everything else
 * parameters::Parse accepts through ffv1o_stream_ext).  TEST INFRASTRUCTURE ONLY (see ffv1_oracle.h).
 *
 * Every function cites the reference lines (under /root/reference/Source/Lib) it restates or inverts.
 * The encoder half has no in-tree counterpart (the reference runs FFmpeg, CLI/Output.cpp:356); it is
 * written as the exact inverse of the in-tree decoder following RFC 9043, and is pinned by feeding its
 * output to the real reference binary (oracle/_ref/rawcooked --check), see tests/.
 *
 * Scalar, single threaded, no dependencies beyond libc.
 */
#include "ffv1_oracle.h"
#include <stdlib.h>
#include <string.h>

#define CONTEXT_SIZE 32          /* states_size, CoDec/FFV1/FFV1_RangeCoder.h:23 */
#define MAX_QUANT 256            /* MAX_QUANT_TABLE_SIZE, CoDec/FFV1/Coder/FFV1_Coder.h:21 */

/* ------------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------------- */
/* default_state_transitions, CoDec/FFV1/FFV1_Frame.cpp:35-55 (a bitstream constant of FFV1, RFC 9043
 * section 4.2 "state_transition_delta" default) -- generated here as data, values identical. */
static const uint8_t one_state_default[256] = {
      0,  0,  0,  0,  0,  0,  0,  0, 20, 21, 22, 23, 24, 25, 26, 27,
     28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42,
     43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 56, 57,
     58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73,
     74, 75, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86, 87, 88,
     89, 90, 91, 92, 93, 94, 94, 95, 96, 97, 98, 99,100,101,102,103,
    104,105,106,107,108,109,110,111,112,113,114,114,115,116,117,118,
    119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,133,
    134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,
    150,151,152,152,153,154,155,156,157,158,159,160,161,162,163,164,
    165,166,167,168,169,170,171,171,172,173,174,175,176,177,178,179,
    180,181,182,183,184,185,186,187,188,189,190,190,191,192,194,194,
    195,196,197,198,199,200,201,202,202,204,205,206,207,208,209,209,
    210,211,212,213,215,215,216,217,218,219,220,220,222,223,224,225,
    226,227,227,229,229,230,231,232,234,234,235,236,237,238,239,240,
    241,242,243,244,245,246,247,248,248,  0,  0,  0,  0,  0,  0,  0,
};
/* The table FFmpeg's encoder switches to with -coder 2 (its "ver2_state"), restated from memory [ffmpeg-knowledge]: any table is
 * conformant, because with coder_type 2 it is part of the stream -- the decoder adds the transmitted deltas to its default table. */
static const uint8_t one_state_alt[256] = {
      0,  10,  10,  10,  10,  16,  16,  16,  28,  16,  16,  29,  42,  49,  20,  49,  59,  25,  26,  26,  27,  31,  33,  33,  33,  34,  34,  37,  67,  38,  39,  39,
     40,  40,  41,  79,  43,  44,  45,  45,  48,  48,  64,  50,  51,  52,  88,  52,  53,  74,  55,  57,  58,  58,  74,  60, 101,  61,  62,  84,  66,  66,  68,  69,
     87,  82,  71,  97,  73,  73,  82,  75, 111,  77,  94,  78,  87,  81,  83,  97,  85,  83,  94,  86,  99,  89,  90,  99, 111,  92,  93, 134,  95,  98, 105,  98,
    105, 110, 102, 108, 102, 118, 103, 106, 106, 113, 109, 112, 114, 112, 116, 125, 115, 116, 117, 117, 126, 119, 125, 121, 121, 123, 145, 124, 126, 131, 127, 129,
    165, 130, 132, 138, 133, 135, 145, 136, 137, 139, 146, 141, 143, 142, 144, 148, 147, 155, 151, 149, 151, 150, 152, 157, 153, 154, 156, 168, 158, 162, 161, 160,
    172, 163, 169, 164, 166, 184, 167, 170, 177, 174, 171, 173, 182, 176, 180, 178, 175, 189, 179, 181, 186, 183, 192, 185, 200, 187, 191, 188, 190, 197, 193, 196,
    197, 194, 195, 196, 198, 202, 199, 201, 210, 203, 207, 204, 205, 206, 208, 214, 209, 211, 221, 212, 213, 215, 224, 216, 217, 218, 219, 220, 222, 228, 223, 225,
    226, 224, 227, 229, 240, 230, 231, 232, 233, 234, 235, 236, 238, 239, 237, 242, 241, 243, 242, 244, 245, 246, 247, 248, 249, 250, 251, 252, 252, 253, 254, 255,
};

/* Quantisation levels for indices 0..127 given as run lengths of equal levels (that is how the record
 * carries them, FFV1_Parameters.cpp:222-253).  These are the level maps FFmpeg's encoder uses
 * ([ffmpeg-knowledge], SURVEY.md 8c: not pinned by any in-tree vector; any monotone map is conformant
 * because the tables travel in the configuration record). */
static const uint8_t runs_q11[]      = { 1, 1, 3, 7, 20, 96 };          /* levels 0..5, <=8 bit */
static const uint8_t runs_q5[]       = { 1, 3, 124 };                   /* levels 0..2, <=8 bit */
static const uint8_t runs_q9_10bit[] = { 5, 8, 14, 29, 72 };            /* levels 0..4, >8 bit  */
static const uint8_t runs_q5_10bit[] = { 11, 53, 64 };                  /* levels 0..2, >8 bit  */

typedef struct {
    int16_t  q[5][MAX_QUANT];   /* value = level * scale, negative half mirrored */
    uint32_t context_count;
} quant_set;

static void fill_quant(int16_t* q, const uint8_t* runs, int nruns, int scale)
{
    int k = 0;
    for (int v = 0; v < nruns; v++)
        for (int a = 0; a < runs[v]; a++)
            q[k++] = (int16_t)(scale * v);
    /* mirror, FFV1_Parameters.cpp:243-245 */
    for (int i = 1; i < 128; i++)
        q[256 - i] = (int16_t)-q[i];
    q[128] = (int16_t)-q[127];
}

/* Two table sets as FFmpeg transmits them: set 0 = 3-input model, set 1 = 5-input model. */
static const uint8_t runs_q3[]       = { 4, 124 };                      /* compact model, <=8 bit */
static const uint8_t runs_q3_10bit[] = { 24, 104 };                     /* compact model, >8 bit  */

static void build_quant_sets_c(uint32_t bps, quant_set qs[2], int compact)
{
    memset(qs, 0, 2 * sizeof(quant_set));
    if (bps <= 8) {
        fill_quant(qs[0].q[0], runs_q11, 6, 1);
        fill_quant(qs[0].q[1], runs_q11, 6, 11);
        fill_quant(qs[0].q[2], runs_q11, 6, 11 * 11);
        qs[0].context_count = (11 * 11 * 11 + 1) / 2;
        fill_quant(qs[1].q[0], runs_q11, 6, 1);
        fill_quant(qs[1].q[1], runs_q11, 6, 11);
        fill_quant(qs[1].q[2], runs_q5, 3, 11 * 11);
        fill_quant(qs[1].q[3], runs_q5, 3, 5 * 11 * 11);
        fill_quant(qs[1].q[4], runs_q5, 3, 5 * 5 * 11 * 11);
        qs[1].context_count = (11 * 11 * 5 * 5 * 5 + 1) / 2;
    } else {
        fill_quant(qs[0].q[0], runs_q9_10bit, 5, 1);
        fill_quant(qs[0].q[1], runs_q9_10bit, 5, 9);
        fill_quant(qs[0].q[2], runs_q9_10bit, 5, 9 * 9);
        qs[0].context_count = (9 * 9 * 9 + 1) / 2;
        fill_quant(qs[1].q[0], runs_q9_10bit, 5, 1);
        fill_quant(qs[1].q[1], runs_q9_10bit, 5, 9);
        fill_quant(qs[1].q[2], runs_q5_10bit, 3, 9 * 9);
        fill_quant(qs[1].q[3], runs_q5_10bit, 3, 5 * 9 * 9);
        fill_quant(qs[1].q[4], runs_q5_10bit, 3, 5 * 5 * 9 * 9);
        qs[1].context_count = (9 * 9 * 5 * 5 * 5 + 1) / 2;
    }
    if (compact) {   /* context_model 2: the device's LDS-sized 5-input model, 5,5,3,3,3 levels = 338 contexts (same rule as ffv1_host.cpp) */
        const uint8_t* q5 = bps <= 8 ? runs_q5 : runs_q5_10bit;
        const uint8_t* q3 = bps <= 8 ? runs_q3 : runs_q3_10bit;
        memset(&qs[1], 0, sizeof(quant_set));
        fill_quant(qs[1].q[0], q5, 3, 1); fill_quant(qs[1].q[1], q5, 3, 5); fill_quant(qs[1].q[2], q3, 2, 25);
        fill_quant(qs[1].q[3], q3, 2, 75); fill_quant(qs[1].q[4], q3, 2, 225);
        qs[1].context_count = (5 * 5 * 3 * 3 * 3 + 1) / 2;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Pixel formats
 * ---------------------------------------------------------------------------------------------- */
uint32_t ffv1o_bits_per_raw_sample(uint32_t f)
{
    switch (f) {
    case FFV1O_RGB8: case FFV1O_RGBA8: case FFV1O_Y8: return 8;
    case FFV1O_RGB10_FILLEDA_BE: case FFV1O_RGB10_FILLEDA_LE: return 10;
    case FFV1O_RGBA10_FILLEDA_BE: case FFV1O_RGBA10_FILLEDA_LE: case FFV1O_Y10_FILLEDA_BE: case FFV1O_Y10_FILLEDB_BE: return 10;
    case FFV1O_RGB12_FILLEDA_BE: case FFV1O_RGB12_FILLEDA_LE: return 12;
    case FFV1O_RGB12_PACKED_BE: case FFV1O_RGBA12_PACKED_BE: case FFV1O_RGBA12_FILLEDA_BE: case FFV1O_RGBA12_FILLEDA_LE:
    case FFV1O_Y12_PACKED_BE: return 12;
    default: return 16;
    }
}
uint32_t ffv1o_plane_count(uint32_t f)
{
    switch (f) {
    case FFV1O_Y8: case FFV1O_Y16_BE: case FFV1O_Y16_LE: case FFV1O_Y10_FILLEDA_BE: case FFV1O_Y10_FILLEDB_BE: case FFV1O_Y12_PACKED_BE: return 1;
    case FFV1O_RGBA8: case FFV1O_RGBA16_BE: case FFV1O_RGBA16_LE: case FFV1O_RGBA10_FILLEDA_BE: case FFV1O_RGBA10_FILLEDA_LE:
    case FFV1O_RGBA12_PACKED_BE: case FFV1O_RGBA12_FILLEDA_BE: case FFV1O_RGBA12_FILLEDA_LE: return 4;
    default: return 3;
    }
}
uint32_t ffv1o_bytes_per_pixel(uint32_t f)
{
    switch (f) {
    case FFV1O_RGB8: return 3;
    case FFV1O_RGB10_FILLEDA_BE: case FFV1O_RGB10_FILLEDA_LE: return 4;
    case FFV1O_RGB12_FILLEDA_BE: case FFV1O_RGB12_FILLEDA_LE: return 6;
    case FFV1O_RGB16_BE: case FFV1O_RGB16_LE: return 6;
    case FFV1O_RGBA8: return 4;
    case FFV1O_RGBA16_BE: case FFV1O_RGBA16_LE: case FFV1O_RGBA12_FILLEDA_BE: case FFV1O_RGBA12_FILLEDA_LE: return 8;
    case FFV1O_Y8: return 1;
    case FFV1O_RGB12_PACKED_BE: case FFV1O_RGBA12_PACKED_BE: case FFV1O_Y12_PACKED_BE:
    case FFV1O_RGBA10_FILLEDA_BE: case FFV1O_RGBA10_FILLEDA_LE: case FFV1O_Y10_FILLEDA_BE: case FFV1O_Y10_FILLEDB_BE: return 0;   /* fields straddle bytes */
    case FFV1O_EXR_RGB16: return 0;                                                                                                   /* planar inside the line */
    default: return 2;
    }
}
/* 0 = whole bytes per pixel; else fields per 32-bit word stream: 12 = packed 12-bit, 10 = three 10-bit fields per word */
static int word_layout(uint32_t f)
{
    switch (f) {
    case FFV1O_RGB12_PACKED_BE: case FFV1O_RGBA12_PACKED_BE: case FFV1O_Y12_PACKED_BE: return 12;
    case FFV1O_RGBA10_FILLEDA_BE: case FFV1O_RGBA10_FILLEDA_LE: case FFV1O_Y10_FILLEDA_BE: case FFV1O_Y10_FILLEDB_BE: return 10;
    default: return 0;
    }
}
size_t ffv1o_line_bytes(uint32_t f, uint32_t width, int dpx_line_padding)
{
    const size_t fields = (size_t)width * ffv1o_plane_count(f);
    if (f == FFV1O_EXR_RGB16) return 8 + 6 * (size_t)width;         /* EXR.cpp:601-606 */
    if (word_layout(f) == 12) return (fields * 12 + 31) / 32 * 4;    /* Transform.cpp:191-194, 881-884 */
    if (word_layout(f) == 10) return (fields + 2) / 3 * 4;           /* Transform.cpp:751-753 */
    size_t n = (size_t)width * ffv1o_bytes_per_pixel(f);
    if (dpx_line_padding)      /* Line_Alignment = 32 bit, Utils/RawFrame/RawFrame.cpp:109 */
        n = (n + 3) & ~(size_t)3;
    return n;
}
size_t ffv1o_payload_bytes(const ffv1o_params* p, size_t line_bytes)
{
    if (p->flags & FFV1O_FLAG_ALTERN) return ((size_t)p->width * p->height + 2) / 3 * 4;   /* DPX.cpp:465-469 */
    return line_bytes * p->height;
}
static int is_rgb(uint32_t f) { return ffv1o_plane_count(f) != 1; }
static int has_alpha(uint32_t f) { return ffv1o_plane_count(f) == 4; }
static int is_be(uint32_t f)
{
    return f == FFV1O_RGB10_FILLEDA_BE || f == FFV1O_RGB12_FILLEDA_BE || f == FFV1O_RGB16_BE ||
           f == FFV1O_RGBA16_BE || f == FFV1O_Y16_BE || f == FFV1O_RGB12_PACKED_BE || f == FFV1O_RGBA10_FILLEDA_BE ||
           f == FFV1O_RGBA12_PACKED_BE || f == FFV1O_RGBA12_FILLEDA_BE || f == FFV1O_Y10_FILLEDA_BE || f == FFV1O_Y10_FILLEDB_BE ||
           f == FFV1O_Y12_PACKED_BE;
}
static inline uint32_t rd16(const uint8_t* p, int be) { return be ? ((uint32_t)p[0] << 8) | p[1] : ((uint32_t)p[1] << 8) | p[0]; }
static inline void wr16(uint8_t* p, uint32_t v, int be) { if (be) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; } else { p[1] = (uint8_t)(v >> 8); p[0] = (uint8_t)v; } }
static inline uint32_t rd32(const uint8_t* p, int be)
{
    return be ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
              : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
}
static inline void wr32(uint8_t* p, uint32_t v, int be)
{
    if (be) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
    else    { p[3] = (uint8_t)(v >> 24); p[2] = (uint8_t)(v >> 16); p[1] = (uint8_t)(v >> 8); p[0] = (uint8_t)v; }
}

/* Forward of Transform.cpp: file components (R,G,B[,A]) of one pixel. */
static inline void load_px(uint32_t f, const uint8_t* p, int be, uint32_t c[4])
{
    switch (f) {
    case FFV1O_RGB8:  c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; break;                       /* Transform.cpp:70-88 */
    case FFV1O_RGBA8: c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; c[3] = p[3]; break;           /* :423-442 */
    case FFV1O_RGB10_FILLEDA_BE: case FFV1O_RGB10_FILLEDA_LE: {                             /* :91-132 */
        uint32_t w = rd32(p, be);
        c[0] = (w >> 22) & 0x3FF; c[1] = (w >> 12) & 0x3FF; c[2] = (w >> 2) & 0x3FF; break; }
    case FFV1O_RGB12_FILLEDA_BE: case FFV1O_RGB12_FILLEDA_LE:                               /* :325-372 */
        c[0] = rd16(p, be) >> 4; c[1] = rd16(p + 2, be) >> 4; c[2] = rd16(p + 4, be) >> 4; break;
    case FFV1O_RGB16_BE: case FFV1O_RGB16_LE:                                               /* :375-420 */
        c[0] = rd16(p, be); c[1] = rd16(p + 2, be); c[2] = rd16(p + 4, be); break;
    case FFV1O_RGBA16_BE: case FFV1O_RGBA16_LE:                                             /* :603-650 */
        c[0] = rd16(p, be); c[1] = rd16(p + 2, be); c[2] = rd16(p + 4, be); c[3] = rd16(p + 6, be); break;
    case FFV1O_RGBA12_FILLEDA_BE: case FFV1O_RGBA12_FILLEDA_LE:                             /* :553-600 */
        c[0] = rd16(p, be) >> 4; c[1] = rd16(p + 2, be) >> 4; c[2] = rd16(p + 4, be) >> 4; c[3] = rd16(p + 6, be) >> 4; break;
    case FFV1O_Y8: c[0] = p[0]; break;                                                      /* :653-668 */
    default: c[0] = rd16(p, be); break;                                                     /* :993-1034 */
    }
}
static inline void store_px(uint32_t f, uint8_t* p, int be, const uint32_t c[4])
{
    switch (f) {
    case FFV1O_RGB8:  p[0] = (uint8_t)c[0]; p[1] = (uint8_t)c[1]; p[2] = (uint8_t)c[2]; break;
    case FFV1O_RGBA8: p[0] = (uint8_t)c[0]; p[1] = (uint8_t)c[1]; p[2] = (uint8_t)c[2]; p[3] = (uint8_t)c[3]; break;
    case FFV1O_RGB10_FILLEDA_BE: case FFV1O_RGB10_FILLEDA_LE:
        wr32(p, ((c[0] & 0x3FF) << 22) | ((c[1] & 0x3FF) << 12) | ((c[2] & 0x3FF) << 2), be); break;
    case FFV1O_RGB12_FILLEDA_BE: case FFV1O_RGB12_FILLEDA_LE:
        wr16(p, (c[0] << 4) & 0xFFFF, be); wr16(p + 2, (c[1] << 4) & 0xFFFF, be); wr16(p + 4, (c[2] << 4) & 0xFFFF, be); break;
    case FFV1O_RGB16_BE: case FFV1O_RGB16_LE:
        wr16(p, c[0] & 0xFFFF, be); wr16(p + 2, c[1] & 0xFFFF, be); wr16(p + 4, c[2] & 0xFFFF, be); break;
    case FFV1O_RGBA16_BE: case FFV1O_RGBA16_LE:
        wr16(p, c[0] & 0xFFFF, be); wr16(p + 2, c[1] & 0xFFFF, be); wr16(p + 4, c[2] & 0xFFFF, be); wr16(p + 6, c[3] & 0xFFFF, be); break;
    case FFV1O_RGBA12_FILLEDA_BE: case FFV1O_RGBA12_FILLEDA_LE:
        wr16(p, (c[0] << 4) & 0xFFFF, be); wr16(p + 2, (c[1] << 4) & 0xFFFF, be); wr16(p + 4, (c[2] << 4) & 0xFFFF, be); wr16(p + 6, (c[3] << 4) & 0xFFFF, be); break;
    case FFV1O_Y8: p[0] = (uint8_t)c[0]; break;
    default: wr16(p, c[0] & 0xFFFF, be); break;
    }
}
/* FFV1 codes 9..15-bit RGB without alpha with G and B exchanged (RFC 9043 3.7.2.1; Transform.cpp:104,126,
 * 338,363 "g and b are inverted"); 8 and 16 bit and every RGBA flavor are not exchanged (:78-83,388-390,460). */
static int gb_swapped(uint32_t f)
{
    uint32_t bps = ffv1o_bits_per_raw_sample(f);
    return is_rgb(f) && !has_alpha(f) && bps >= 9 && bps <= 15;
}

/* Sequential reader / writer of the word-stream layouts, shaped like the reference's From() loops: fields are
 * taken from / put into 32-bit words one after the other (Transform.cpp:214-322 for 12-bit packed, :451-480 for
 * RGBA 10-bit, :781-796 for Y 10-bit). */
typedef struct { const uint8_t* p; uint8_t* q; int be, kind, off; uint64_t acc; int nbits; int slot; } field_io;
static void fio_start(field_io* s, uint32_t f, const uint8_t* rd, uint8_t* wr)
{
    s->p = rd; s->q = wr; s->be = is_be(f); s->kind = word_layout(f); s->acc = 0; s->nbits = 0; s->slot = 0;
    s->off = f == FFV1O_Y10_FILLEDB_BE ? 0 : 2;
    if (s->kind == 10 && ffv1o_plane_count(f) == 4) s->kind = 11;   /* RGBA 10-bit: first field in the top bits */
}
static uint32_t fio_get(field_io* s)
{
    if (s->kind == 12) {                                   /* LSB-first bit stream over big-endian words */
        if (s->nbits < 12) { s->acc |= (uint64_t)rd32(s->p, 1) << s->nbits; s->p += 4; s->nbits += 32; }
        uint32_t v = (uint32_t)(s->acc & 0xFFF); s->acc >>= 12; s->nbits -= 12; return v;
    }
    if (s->slot == 0) s->acc = rd32(s->p, s->be), s->p += 4;
    uint32_t v = s->kind == 11 ? (uint32_t)(s->acc >> (22 - 10 * s->slot)) & 0x3FF : (uint32_t)(s->acc >> (10 * s->slot + s->off)) & 0x3FF;
    s->slot = (s->slot + 1) % 3;
    return v;
}
static void fio_put(field_io* s, uint32_t v)
{
    if (s->kind == 12) {
        s->acc |= (uint64_t)(v & 0xFFF) << s->nbits; s->nbits += 12;
        if (s->nbits >= 32) { wr32(s->q, (uint32_t)s->acc, 1); s->q += 4; s->acc >>= 32; s->nbits -= 32; }
        return;
    }
    v &= 0x3FF;
    s->acc |= s->kind == 11 ? (uint64_t)v << (22 - 10 * s->slot) : (uint64_t)v << (10 * s->slot + s->off);
    if (++s->slot == 3) { wr32(s->q, (uint32_t)s->acc, s->be); s->q += 4; s->acc = 0; s->slot = 0; }
}
static void fio_flush(field_io* s)                         /* the last, partly filled word of a line; padding bits are zero */
{
    if (s->kind == 12) { if (s->nbits) { wr32(s->q, (uint32_t)s->acc, 1); s->q += 4; } s->acc = 0; s->nbits = 0; return; }
    if (s->slot) { wr32(s->q, (uint32_t)s->acc, s->be); s->q += 4; s->acc = 0; s->slot = 0; }
}

void ffv1o_unpack(const ffv1o_params* p, const uint8_t* payload, size_t line_bytes, int32_t* const planes[4])
{
    const uint32_t f = p->pixfmt, bpp = ffv1o_bytes_per_pixel(f);
    const int be = is_be(f), swap = gb_swapped(f), rgb = is_rgb(f), alpha = has_alpha(f), words = word_layout(f) != 0;
    const int vflip = (p->flags & FFV1O_FLAG_VFLIP) != 0, altern = (p->flags & FFV1O_FLAG_ALTERN) != 0;
    const int32_t off = (int32_t)1 << ffv1o_bits_per_raw_sample(f);
    field_io io;
    if (altern) fio_start(&io, f, payload, NULL);
    for (uint32_t fy = 0; fy < p->height; fy++) {          /* fy: line in the file */
        const uint32_t y = vflip ? p->height - 1 - fy : fy;
        const uint8_t* s = payload + (size_t)fy * line_bytes;
        if (words && !altern) fio_start(&io, f, s, NULL);
        size_t o = (size_t)y * p->width;
        for (uint32_t x = 0; x < p->width; x++, s += bpp) {
            uint32_t c[4] = { 0, 0, 0, 0 };
            if (f == FFV1O_EXR_RGB16) {                     /* inverse of transform_jpeg2000rct_exr_Raw_RGB_16::From, Transform.cpp:1105-1127 */
                const uint8_t* l = payload + (size_t)fy * line_bytes + 8;
                c[2] = rd16(l + 2 * (size_t)x, 0); c[1] = rd16(l + 2 * ((size_t)p->width + x), 0); c[0] = rd16(l + 2 * (2 * (size_t)p->width + x), 0);
            }
            else if (words) for (uint32_t i = 0; i < ffv1o_plane_count(f); i++) c[i] = fio_get(&io);
            else load_px(f, s, be, c);
            if (!rgb) { planes[0][o + x] = (int32_t)c[0]; continue; }
            /* inverse of JPEG2000RCT, Transform.cpp:29-37 */
            int32_t r = (int32_t)c[0], g = (int32_t)c[1], b = (int32_t)c[2];
            if (swap) { int32_t t = g; g = b; b = t; }
            b -= g; r -= g;
            g += (b + r) >> 2;
            planes[0][o + x] = g;
            planes[1][o + x] = b + off;
            planes[2][o + x] = r + off;
            if (alpha) planes[3][o + x] = (int32_t)c[3];
        }
    }
}

void ffv1o_pack(const ffv1o_params* p, int32_t* const planes[4], uint8_t* payload, size_t line_bytes)
{
    const uint32_t f = p->pixfmt, bpp = ffv1o_bytes_per_pixel(f);
    const int be = is_be(f), swap = gb_swapped(f), rgb = is_rgb(f), alpha = has_alpha(f), words = word_layout(f) != 0;
    const int vflip = (p->flags & FFV1O_FLAG_VFLIP) != 0, altern = (p->flags & FFV1O_FLAG_ALTERN) != 0;
    const int32_t off = (int32_t)1 << ffv1o_bits_per_raw_sample(f);
    field_io io;
    if (altern) fio_start(&io, f, NULL, payload);
    for (uint32_t fy = 0; fy < p->height; fy++) {
        const uint32_t y = vflip ? p->height - 1 - fy : fy;
        uint8_t* d = payload + (size_t)fy * line_bytes;
        if (!altern) memset(d, 0, line_bytes);
        if (words && !altern) fio_start(&io, f, NULL, d);
        size_t o = (size_t)y * p->width;
        for (uint32_t x = 0; x < p->width; x++, d += bpp) {
            uint32_t c[4] = { 0, 0, 0, 0 };
            if (!rgb) c[0] = (uint32_t)planes[0][o + x];
            else {
                /* JPEG2000RCT, Transform.cpp:29-37 */
                int32_t g = planes[0][o + x], b = planes[1][o + x], r = planes[2][o + x];
                b -= off; r -= off;
                g -= (b + r) >> 2;
                b += g; r += g;
                if (swap) { int32_t t = g; g = b; b = t; }
                c[0] = (uint32_t)r; c[1] = (uint32_t)g; c[2] = (uint32_t)b;
                if (alpha) c[3] = (uint32_t)planes[3][o + x];
            }
            if (f == FFV1O_EXR_RGB16) {
                uint8_t* l = payload + (size_t)fy * line_bytes + 8;
                wr16(l + 2 * (size_t)x, c[2] & 0xFFFF, 0); wr16(l + 2 * ((size_t)p->width + x), c[1] & 0xFFFF, 0); wr16(l + 2 * (2 * (size_t)p->width + x), c[0] & 0xFFFF, 0);
            }
            else if (words) for (uint32_t i = 0; i < ffv1o_plane_count(f); i++) fio_put(&io, c[i]);
            else store_px(f, d, be, c);
        }
        if (f == FFV1O_EXR_RGB16) {                         /* line header: y coordinate and byte count, Transform.cpp:1078-1085 */
            uint8_t* l = payload + (size_t)fy * line_bytes;
            wr32(l, fy, 0); wr32(l + 4, 6 * p->width, 0);
        }
        if (words && !altern) fio_flush(&io);
    }
    if (altern) fio_flush(&io);
}

/* ------------------------------------------------------------------------------------------------
 * CRC-32 (ZenCRC32.cpp:1062-1135: poly 0x04C11DB7, MSB first, init 0, no xor-out) -- bitwise table.
 * ---------------------------------------------------------------------------------------------- */
static uint32_t crc_table[256];
static int crc_table_ready;
static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i << 24;
        for (int k = 0; k < 8; k++)
            c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
        crc_table[i] = c;
    }
    crc_table_ready = 1;
}
uint32_t ffv1o_crc32(const uint8_t* d, size_t n)
{
    if (!crc_table_ready) crc_init();
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++)
        c = (c << 8) ^ crc_table[(c >> 24) ^ d[i]];
    return c;
}

/* ------------------------------------------------------------------------------------------------
 * Range coder, encoder side: inverse of rangecoder::b, FFV1_RangeCoder.cpp:71-102 (RFC 9043 3.8.1).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t low, range;
    int      outstanding_byte, outstanding_count;
    uint8_t* p; uint8_t* end; int overflow;
    uint8_t  one_state[256], zero_state[256];
    uint64_t decisions;
} rc_enc;

/* Optional per-thread trace of one slice (ffv1o_trace_slice): every decision as state | bit << 8, and every
 * sample as set << 30 | |ctx| << 17 | (residual & 0x1FFFF) -- the intermediates of the device pipeline. */
static __thread uint16_t* t_dec; static __thread size_t t_dec_cap, t_dec_n;
static __thread uint32_t* t_sym; static __thread size_t t_sym_n;

static void rc_tables_from(uint8_t* one, uint8_t* zero, const uint8_t* one_src)
{
    /* AssignStateTransitions, FFV1_RangeCoder.cpp:35-41 */
    memcpy(one, one_src, 256);
    zero[0] = 0;
    for (int i = 1; i < 256; i++)
        zero[i] = (uint8_t)(256 - one[256 - i]);
}
static void rce_init(rc_enc* c, uint8_t* buf, size_t cap)       /* with the default table: the record, and every slice up to its header */
{
    c->low = 0; c->range = 0xFF00; c->outstanding_byte = -1; c->outstanding_count = 0;
    c->p = buf; c->end = buf + cap; c->overflow = 0; c->decisions = 0;
    rc_tables_from(c->one_state, c->zero_state, one_state_default);
}
static inline void rce_out(rc_enc* c, int v) { if (c->p < c->end) *c->p++ = (uint8_t)v; else c->overflow = 1; }
static void rce_renorm(rc_enc* c)
{
    while (c->range < 0x100) {
        if (c->outstanding_byte < 0) {
            c->outstanding_byte = (int)(c->low >> 8);
        } else if (c->low <= 0xFF00) {
            rce_out(c, c->outstanding_byte);
            for (; c->outstanding_count; c->outstanding_count--) rce_out(c, 0xFF);
            c->outstanding_byte = (int)(c->low >> 8);
        } else if (c->low >= 0x10000) {
            rce_out(c, c->outstanding_byte + 1);
            for (; c->outstanding_count; c->outstanding_count--) rce_out(c, 0x00);
            c->outstanding_byte = (int)((c->low >> 8) & 0xFF);
        } else {
            c->outstanding_count++;
        }
        c->low = (c->low & 0xFF) << 8;
        c->range <<= 8;
    }
}
static inline void rce_put(rc_enc* c, uint8_t* state, int bit)
{
    uint32_t r1 = (c->range * *state) >> 8;      /* Mask2, FFV1_RangeCoder.cpp:90 */
    if (t_dec) { if (t_dec_n < t_dec_cap) t_dec[t_dec_n] = (uint16_t)(*state | (bit ? 0x100 : 0)); t_dec_n++; }
    if (!bit) { c->range -= r1; *state = c->zero_state[*state]; }
    else { c->low += c->range - r1; c->range = r1; *state = c->one_state[*state]; }
    c->decisions++;
    rce_renorm(c);
}
/* Ends the coder; with_end_bit = slices (decoder consumes state 129 at FFV1_Slice.cpp:336-340). */
static size_t rce_terminate(rc_enc* c, uint8_t* start, int with_end_bit)
{
    if (with_end_bit) { uint8_t s = 129; rce_put(c, &s, 0); }
    c->range = 0xFF; c->low += 0xFF; rce_renorm(c);
    c->range = 0xFF; rce_renorm(c);
    return (size_t)(c->p - start);
}
static inline int ilog2(uint32_t a) { int e = 0; while (a >>= 1) e++; return e; }
/* inverse of rangecoder::u / ::s, FFV1_RangeCoder.cpp:105-305 */
static void rce_symbol(rc_enc* c, uint8_t* st, int32_t v, int is_signed)
{
    if (!v) { rce_put(c, st + 0, 1); return; }
    uint32_t a = (uint32_t)(v < 0 ? -v : v);
    int e = ilog2(a);
    rce_put(c, st + 0, 0);
    for (int i = 0; i < e; i++) rce_put(c, st + 1 + (i < 9 ? i : 9), 1);
    rce_put(c, st + 1 + (e < 9 ? e : 9), 0);
    for (int i = e - 1; i >= 0; i--) rce_put(c, st + 22 + (i < 9 ? i : 9), (a >> i) & 1);
    if (is_signed) rce_put(c, st + 11 + (e < 10 ? e : 10), v < 0);
}

/* ------------------------------------------------------------------------------------------------
 * The stream's parameters (parameters, FFV1_Parameters.h / FFV1_Parameters.cpp:23-183): what the configuration record of a
 * version 3 stream or the header inside every version 0 / 1 frame says.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t bps, bits, planes, set_index_count;
    int rgb, alpha, overflow16;
    uint32_t version, micro, intra, ec;     /* version 0, 1 or 3 */
    uint32_t num_h, num_v;
    int custom;                             /* coder_type 2: one_state travels in the stream */
    uint8_t one_state[256];                 /* the transitions the slices are coded with */
    uint32_t nsets;
    uint32_t idx[3];                        /* quant_table_set_index this ENCODER writes for plane group 0 (Y), 1 (Cb, Cr), 2 (alpha) */
    uint32_t idx_alt[3]; int alt;           /* ... and in the slices with odd sx + sy when alt is set */
    quant_set qs[8];
    const uint8_t* init[8];                 /* coded initial states of set i (context_count x 32) or NULL = all 128 */
    uint8_t* init_own[8];                   /* the same when a parser allocated them */
} codec_ctx;

static void ctx_geometry(codec_ctx* k, uint32_t pixfmt)
{
    k->bps = ffv1o_bits_per_raw_sample(pixfmt);
    k->rgb = is_rgb(pixfmt);
    k->alpha = has_alpha(pixfmt);
    k->planes = ffv1o_plane_count(pixfmt);
    /* FFV1_Parameters.cpp:160-181 */
    k->overflow16 = (!k->rgb && k->bps == 16);
    k->bits = k->rgb ? k->bps + 1 : (k->bps <= 8 ? 8 : k->bps);
    k->set_index_count = k->rgb ? k->planes - 1 : 2;       /* version < 4: 1 + 1 (+ alpha) */
}
static void ctx_free(codec_ctx* k)
{
    for (int i = 0; i < 8; i++) free(k->init_own[i]);
    free(k);
}
/* level map (one level per |difference| 0..127) -> one table of a set; *scale is the context count so far (QuantizationTable, :222-253) */
static void quant_from_levels(int16_t* q, const uint8_t* levels, int32_t* scale)
{
    for (int i = 0; i < 128; i++) q[i] = (int16_t)(*scale * levels[i]);
    for (int i = 1; i < 128; i++) q[256 - i] = (int16_t)-q[i];
    q[128] = (int16_t)-q[127];
    *scale *= 2 * (levels[127] + 1) - 1;
}
uint32_t ffv1o_ext_context_count(const ffv1o_stream_ext* e, uint32_t set)
{
    int32_t scale = 1;
    for (int j = 0; j < 5; j++) scale *= 2 * (e->levels[set][j][127] + 1) - 1;
    return (uint32_t)((scale + 1) >> 1);
}
static codec_ctx* ctx_from_params(const ffv1o_params* p)
{
    codec_ctx* k = calloc(1, sizeof *k);
    ctx_geometry(k, p->pixfmt);
    k->num_h = p->num_h_slices; k->num_v = p->num_v_slices; k->ec = p->ec;
    const ffv1o_stream_ext* e = p->ext;
    if (!e) {
        /* what FFmpeg's encoder sends: two table sets (version 3), the planes all on the set -context picks; version 1 carries that one set */
        quant_set d[2];
        build_quant_sets_c(k->bps, d, p->context_model == 2);
        const uint32_t qidx = p->context_model ? 1 : 0;
        k->version = p->level == 1 ? 1 : 3; k->micro = 4; k->intra = 1;
        k->custom = p->coder == 2;
        memcpy(k->one_state, k->custom ? one_state_alt : one_state_default, 256);
        if (k->version == 1) { k->nsets = 1; k->qs[0] = d[qidx]; }
        else { k->nsets = 2; k->qs[0] = d[0]; k->qs[1] = d[1]; k->idx[0] = k->idx[1] = k->idx[2] = qidx; }
        return k;
    }
    k->version = e->version; k->micro = e->micro_version; k->intra = e->intra;
    k->custom = e->custom_transitions != 0;
    memcpy(k->one_state, k->custom ? e->one_state : one_state_default, 256);
    k->nsets = k->version <= 1 ? 1 : e->set_count;
    for (uint32_t i = 0; i < k->nsets; i++) {
        int32_t scale = 1;
        for (int j = 0; j < 5; j++) quant_from_levels(k->qs[i].q[j], e->levels[i][j], &scale);
        k->qs[i].context_count = (uint32_t)((scale + 1) >> 1);
        if (k->version == 3 && e->states_coded[i]) k->init[i] = e->initial_states[i];
    }
    for (int g = 0; g < 3; g++) k->idx[g] = k->version <= 1 ? 0 : e->set_index[g];
    k->alt = k->version == 3 && e->alt_slices;
    for (int g = 0; g < 3; g++) k->idx_alt[g] = e->set_index_alt[g];
    if (k->version <= 1) { k->num_h = k->num_v = 1; k->ec = 0; }
    return k;
}

/* ------------------------------------------------------------------------------------------------
 * Writing the parameters (inverse of parameters::Parse, FFV1_Parameters.cpp:23-183, :206-253)
 * ---------------------------------------------------------------------------------------------- */
static void write_quant_table(rc_enc* c, const int16_t* q)
{
    uint8_t st[CONTEXT_SIZE]; memset(st, 128, sizeof st);     /* fresh states per table, :224 */
    int last = 0, i;
    for (i = 1; i < 128; i++)
        if (q[i] != q[i - 1]) { rce_symbol(c, st, i - last - 1, 0); last = i; }
    rce_symbol(c, st, i - last - 1, 0);
}
/* the fields in the order Parse reads them; `record` = out of band (version 3), else the header inside a version 0 / 1 frame */
static void write_parameters(rc_enc* c, uint8_t* st, const codec_ctx* k, int record)
{
    rce_symbol(c, st, (int32_t)k->version, 0);
    if (k->version >= 3) rce_symbol(c, st, (int32_t)k->micro, 0);      /* >= 4 required, :36-37 */
    rce_symbol(c, st, k->custom ? 2 : 1, 0);                           /* coder_type: range coder, default or transmitted table */
    if (k->custom)                                                     /* state_transition_delta[1..255], :41-55 */
        for (int i = 1; i < 256; i++) rce_symbol(c, st, (int32_t)k->one_state[i] - (int32_t)one_state_default[i], 1);
    rce_symbol(c, st, k->rgb ? 1 : 0, 0);                              /* colorspace_type */
    if (k->version >= 1) rce_symbol(c, st, (int32_t)k->bps, 0);        /* bits_per_raw_sample; version 0 has none: 8 (:62-73) */
    rce_put(c, st, k->rgb ? 1 : 0);                                    /* chroma_planes */
    rce_symbol(c, st, 0, 0);                                           /* log2_h_chroma_subsample */
    rce_symbol(c, st, 0, 0);                                           /* log2_v_chroma_subsample */
    rce_put(c, st, k->alpha);                                          /* alpha_plane */
    if (record) {
        rce_symbol(c, st, (int32_t)k->num_h - 1, 0);
        rce_symbol(c, st, (int32_t)k->num_v - 1, 0);
        rce_symbol(c, st, (int32_t)k->nsets, 0);                       /* quant_table_set_count */
    }
    for (uint32_t i = 0; i < k->nsets; i++)
        for (int j = 0; j < 5; j++)
            write_quant_table(c, k->qs[i].q[j]);
    if (!record) return;
    for (uint32_t i = 0; i < k->nsets; i++) {
        rce_put(c, st, k->init[i] != NULL);                            /* states_coded */
        if (k->init[i])                                                /* as the reference reads them: `States[k] = E.s(States)`, :103-107 */
            for (size_t j = 0; j < (size_t)k->qs[i].context_count * CONTEXT_SIZE; j++) rce_symbol(c, st, (int32_t)k->init[i][j], 1);
    }
    rce_symbol(c, st, (int32_t)k->ec, 0);                              /* ec */
    rce_symbol(c, st, (int32_t)k->intra, 0);                           /* intra (micro_version != 0, :139-144) */
}
size_t ffv1o_config_record(const ffv1o_params* p, uint8_t* out, size_t cap)
{
    codec_ctx* k = ctx_from_params(p);
    if (k->version <= 1) { ctx_free(k); return 0; }  /* version 0 / 1 have no out-of-band record */
    rc_enc c; rce_init(&c, out, cap);
    uint8_t st[CONTEXT_SIZE]; memset(st, 128, sizeof st);
    write_parameters(&c, st, k, 1);
    ctx_free(k);
    size_t n = rce_terminate(&c, out, 0);
    if (c.overflow || n + 4 > cap) return 0;
    uint32_t crc = ffv1o_crc32(out, n);              /* parity: CRC over record||crc == 0, FFV1_Frame.cpp:116 */
    out[n] = (uint8_t)(crc >> 24); out[n + 1] = (uint8_t)(crc >> 16); out[n + 2] = (uint8_t)(crc >> 8); out[n + 3] = (uint8_t)crc;
    return n + 4;
}

/* ------------------------------------------------------------------------------------------------
 * Slice geometry, predictor, contexts (FFV1_Slice.cpp:21-93, :153-156)
 * ---------------------------------------------------------------------------------------------- */
static inline int32_t median3(int32_t a, int32_t b, int32_t c)
{
    /* get_median_number, FFV1_Slice.cpp:21-49 */
    if (a > b) { if (b > c) return b; if (c > a) return a; return c; }
    if (c > b) return b;
    if (c > a) return c;
    return a;
}
static inline int32_t sign_extend(int32_t v, int bits)
{
    uint32_t m = (uint32_t)1 << (bits - 1);
    uint32_t u = (uint32_t)v & ((m << 1) - 1);
    return (int32_t)((u ^ m) - m);
}

/* context states of one slice: an array per plane group, copied from the set's initial states (coder_rangecoder::GOP_Init,
 * Coder/FFV1_Coder_RangeCoder.cpp:34-57) */
static void slice_states(const codec_ctx* k, const uint32_t idx[3], uint8_t (*states[3])[CONTEXT_SIZE])
{
    for (uint32_t g = 0; g < k->set_index_count; g++) {
        const size_t n = (size_t)k->qs[idx[g]].context_count * CONTEXT_SIZE;
        states[g] = malloc(n);
        if (k->init[idx[g]]) memcpy(states[g], k->init[idx[g]], n); else memset(states[g], 128, n);
    }
}

/* One line of one plane.  cur/prev point at x=0 of buffers with 2 guard samples on the left and 1 on the
 * right (SamplesBuffer layout, FFV1_Slice.cpp:406-425).  On entry cur[] holds the line two above (TT). */
static void encode_line(rc_enc* c, const codec_ctx* k, const quant_set* qs, uint8_t (*states)[CONTEXT_SIZE], uint32_t w,
                        int32_t* cur, int32_t* prev, const int32_t* src, uint32_t group)
{
    const int is5 = qs->q[3][127] != 0;                       /* FFV1_Slice.cpp:453 */
    const int32_t mask = (int32_t)(((uint32_t)1 << k->bits) - 1);
    for (uint32_t x = 0; x < w; x++) {
        int32_t* s1 = cur + x; int32_t* s0 = prev + x;
        const int32_t LT = s0[-1], T = s0[0], RT = s0[1], L = s1[-1];
        int32_t ctx = qs->q[0][(L - LT) & 0xFF] + qs->q[1][(LT - T) & 0xFF] + qs->q[2][(T - RT) & 0xFF];
        if (is5) ctx += qs->q[3][(s1[-2] - L) & 0xFF] + qs->q[4][(s1[0] - T) & 0xFF];
        int32_t pred;
        if (k->overflow16) pred = median3((int16_t)L, (int16_t)L + (int16_t)T - (int16_t)LT, (int16_t)T);   /* :52-57 */
        else pred = median3(L, L + T - LT, T);
        int32_t v = src[x] & mask;
        int32_t d = v - pred;
        if (ctx < 0) { ctx = -ctx; d = -d; }
        d = sign_extend(d, (int)k->bits);
        if (t_sym) t_sym[t_sym_n++] = (group << 30) | ((uint32_t)ctx << 17) | ((uint32_t)d & 0x1FFFFu);
        rce_symbol(c, states[ctx], d, 1);
        *s1 = v;
    }
}

static void slice_rect(const ffv1o_params* p, uint32_t nh, uint32_t nv, uint32_t sx, uint32_t sy, uint32_t* x, uint32_t* y, uint32_t* w, uint32_t* h)
{
    /* FFV1_Slice.cpp:153-156 */
    *x = sx * p->width / nh;
    *y = sy * p->height / nv;
    *w = (sx + 1) * p->width / nh - *x;
    *h = (sy + 1) * p->height / nv - *y;
}

static __thread uint64_t g_last_decisions;
uint64_t ffv1o_last_decisions(void) { return g_last_decisions; }

static size_t encode_slice(const ffv1o_params* p, const codec_ctx* k, int32_t* const planes[4], uint32_t sx, uint32_t sy,
                           int first, uint8_t* out, size_t cap)
{
    uint32_t x0, y0, w, h;
    slice_rect(p, k->num_h, k->num_v, sx, sy, &x0, &y0, &w, &h);
    const int inband = k->version <= 1;
    const uint32_t* idx = k->alt && ((sx + sy) & 1) ? k->idx_alt : k->idx;
    rc_enc c; rce_init(&c, out, cap);                         /* default transitions (FFV1_Slice.cpp:214) ... */
    if (!inband) rc_tables_from(c.one_state, c.zero_state, k->one_state);   /* ... the stream's own from the slice header on (:254-255) */
    if (first) { uint8_t ks = 128; rce_put(&c, &ks, 1); }     /* keyframe, FFV1_Frame.cpp:148-156 */
    uint8_t hs[CONTEXT_SIZE]; memset(hs, 128, sizeof hs);
    if (inband) {
        /* version 0 / 1: the stream header inside the frame, parameters::Parse(E, false) (FFV1_Parameters.cpp:23-104), read with the
         * default transitions (FFV1_Slice.cpp:214, 254-255) */
        write_parameters(&c, hs, k, 0);
        rc_tables_from(c.one_state, c.zero_state, k->one_state);
    } else {
    /* slice header, FFV1_Slice.cpp:113-177 */
    rce_symbol(&c, hs, (int32_t)sx, 0);
    rce_symbol(&c, hs, (int32_t)sy, 0);
    rce_symbol(&c, hs, 0, 0);                                 /* slice_width - 1 (slice units) */
    rce_symbol(&c, hs, 0, 0);
    for (uint32_t i = 0; i < k->set_index_count; i++) rce_symbol(&c, hs, (int32_t)idx[i], 0);
    rce_symbol(&c, hs, 3, 0);                                 /* picture_structure: progressive */
    rce_symbol(&c, hs, 0, 0); rce_symbol(&c, hs, 0, 0);       /* sar 0/0 = unknown */
    }

    uint8_t (*states[3])[CONTEXT_SIZE] = { 0, 0, 0 };
    slice_states(k, idx, states);
    int32_t* buf = calloc((size_t)2 * k->planes * (w + 3), sizeof(int32_t));
    int32_t* sample[4][2];
    for (uint32_t pl = 0; pl < k->planes; pl++) {
        sample[pl][0] = buf + 2 * pl * (w + 3) + 2;
        sample[pl][1] = sample[pl][0] + w + 3;
    }
    if (k->rgb) {
        /* SliceContent_LineThenPlane, FFV1_Slice.cpp:406-444 */
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t pl = 0; pl < k->planes; pl++) {
                int32_t* t = sample[pl][0]; sample[pl][0] = sample[pl][1]; sample[pl][1] = t;
                sample[pl][1][-1] = sample[pl][0][0];
                sample[pl][0][w] = sample[pl][0][w - 1];
                const uint32_t g = (pl + 1) >> 1;
                encode_line(&c, k, &k->qs[idx[g]], states[g], w, sample[pl][1], sample[pl][0],
                            planes[pl] + (size_t)(y0 + y) * p->width + x0, g);
            }
    } else {
        /* SliceContent_PlaneThenLine, FFV1_Slice.cpp:346-403 (luma only: chroma_planes = 0) */
        for (uint32_t y = 0; y < h; y++) {
            int32_t* t = sample[0][0]; sample[0][0] = sample[0][1]; sample[0][1] = t;
            sample[0][1][-1] = sample[0][0][0];
            sample[0][0][w] = sample[0][0][w - 1];
            encode_line(&c, k, &k->qs[idx[0]], states[0], w, sample[0][1], sample[0][0], planes[0] + (size_t)(y0 + y) * p->width + x0, 0);
        }
    }
    free(buf);
    for (uint32_t i = 0; i < k->set_index_count; i++) free(states[i]);

    size_t n = rce_terminate(&c, out, 1);
    g_last_decisions += c.decisions;
    if (inband) return c.overflow ? 0 : n;                    /* version 0 / 1: the frame is the coder's bytes, nothing after them */
    /* footer, FFV1_Frame.cpp:177-196, FFV1_Slice.cpp:301-314 */
    if (c.overflow || n + 8 > cap || n > 0xFFFFFF) return 0;
    out[n] = (uint8_t)(n >> 16); out[n + 1] = (uint8_t)(n >> 8); out[n + 2] = (uint8_t)n; n += 3;
    if (k->ec) {
        out[n++] = 0;                                         /* error_status */
        uint32_t crc = ffv1o_crc32(out, n);
        out[n] = (uint8_t)(crc >> 24); out[n + 1] = (uint8_t)(crc >> 16); out[n + 2] = (uint8_t)(crc >> 8); out[n + 3] = (uint8_t)crc;
        n += 4;
    }
    return n;
}

size_t ffv1o_encode_frame(const ffv1o_params* p, int32_t* const planes[4], uint8_t* out, size_t cap, uint32_t* slice_sizes)
{
    codec_ctx* k = ctx_from_params(p);
    size_t pos = 0;
    g_last_decisions = 0;
    for (uint32_t sy = 0; sy < k->num_v; sy++)
        for (uint32_t sx = 0; sx < k->num_h; sx++) {
            size_t n = encode_slice(p, k, planes, sx, sy, pos == 0, out + pos, cap - pos);
            if (!n) { ctx_free(k); return 0; }
            if (slice_sizes) slice_sizes[sy * k->num_h + sx] = (uint32_t)n;
            pos += n;
        }
    ctx_free(k);
    return pos;
}

size_t ffv1o_trace_slice(const ffv1o_params* p, int32_t* const planes[4], uint32_t sx, uint32_t sy,
                         uint32_t* sym_out, uint16_t* dec_out, size_t dec_cap, size_t* ndec, uint8_t* raw_out, size_t raw_cap)
{
    codec_ctx* k = ctx_from_params(p);
    t_sym = sym_out; t_sym_n = 0; t_dec = dec_out; t_dec_cap = dec_cap; t_dec_n = 0;
    size_t n = encode_slice(p, k, planes, sx, sy, sx == 0 && sy == 0, raw_out, raw_cap);
    if (ndec) *ndec = t_dec_n;
    t_sym = NULL; t_dec = NULL;
    ctx_free(k);
    return n;
}

size_t ffv1o_encode_payload(const ffv1o_params* p, const uint8_t* payload, size_t line_bytes, uint8_t* out, size_t cap,
                            uint32_t* slice_sizes)
{
    size_t n = (size_t)p->width * p->height;
    int32_t* planes[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < ffv1o_plane_count(p->pixfmt); i++) planes[i] = malloc(n * sizeof(int32_t));
    ffv1o_unpack(p, payload, line_bytes, planes);
    size_t r = ffv1o_encode_frame(p, planes, out, cap, slice_sizes);
    for (int i = 0; i < 4; i++) free(planes[i]);
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * Decoder: restates rangecoder (FFV1_RangeCoder.cpp:21-132), parameters::Parse (FFV1_Parameters.cpp:23-183),
 * slice::Parse (FFV1_Slice.cpp:210-318), ffv1_frame::OutOfBand / Process (FFV1_Frame.cpp:105-228)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t current, mask;
    const uint8_t *beg, *cur, *end;
    uint8_t one_state[256], zero_state[256];
} rc_dec;
static void rcd_init(rc_dec* c, const uint8_t* buf, size_t n)
{
    c->beg = buf; c->cur = buf; c->end = buf + n;
    c->current = n ? *c->cur : 0; c->mask = 0xFF; c->cur++;       /* AssignBuffer, :22-33 */
    rc_tables_from(c->one_state, c->zero_state, one_state_default);
}
static int rcd_b(rc_dec* c, uint8_t* state)
{
    if (c->mask < 0x100) {
        c->current <<= 8;
        if (c->cur > c->end) return 0;
        if (c->cur < c->end) c->current |= *c->cur;
        c->mask <<= 8;
        c->cur++;
    }
    uint32_t m2 = (c->mask * *state) >> 8;
    c->mask -= m2;
    if (c->current < c->mask) { *state = c->zero_state[*state]; return 0; }
    c->current -= c->mask; c->mask = m2; *state = c->one_state[*state];
    return 1;
}
static size_t rcd_bytes_used(const rc_dec* c)
{
    if (c->cur > c->end) return (size_t)(c->end - c->beg);
    return (size_t)(c->cur - c->beg) - (c->mask < 0x100 ? 0 : 1);
}
static uint32_t rcd_u(rc_dec* c, uint8_t* st)
{
    if (rcd_b(c, st)) return 0;
    int e = 0;
    while (rcd_b(c, st + 1 + (e < 9 ? e : 9))) { if (++e > 31) { c->mask = 0; c->cur = c->end + 1; return 0; } }    /* ForceUnderrun, :308-312 */
    uint32_t a = 1;
    for (int i = e - 1; i >= 0; i--) a = (a << 1) | (uint32_t)rcd_b(c, st + 22 + (i < 9 ? i : 9));
    return a;
}
static int32_t rcd_s(rc_dec* c, uint8_t* st)
{
    if (rcd_b(c, st)) return 0;
    int e = 0;
    while (rcd_b(c, st + 1 + (e < 9 ? e : 9))) { if (++e > 31) { c->mask = 0; c->cur = c->end + 1; return 0; } }    /* ForceUnderrun, :308-312 */
    uint32_t a = 1;                     /* (unsigned: a corrupted stream reaches e = 31, and a signed shift there is undefined in C; the value is the reference's) */
    for (int i = e - 1; i >= 0; i--) a = (a << 1) | (uint32_t)rcd_b(c, st + 22 + (i < 9 ? i : 9));
    return rcd_b(c, st + 11 + (e < 10 ? e : 10)) ? (int32_t)(0u - a) : (int32_t)a;
}

/* parameters::Parse(E, ConfigurationRecord_IsPresent), FFV1_Parameters.cpp:23-183 with QuantizationTableSet / QuantizationTable
 * (:206-253), line by line.  Fills what the stream says into k (geometry derived from it, not from a pixel format); 0 ok, else the
 * step that refused. */
static int parse_parameters(rc_dec* c, codec_ctx* k, int record)
{
    uint8_t st[CONTEXT_SIZE]; memset(st, 128, sizeof st);
    k->version = rcd_u(c, st);
    if (record && k->version <= 1) return 3;
    if (!record && k->version > 1) return 3;
    if (k->version == 2 || k->version > 3) return 3;
    k->micro = 0;
    if (k->version >= 3) k->micro = rcd_u(c, st);
    if (k->version == 3 && k->micro < 4) return 4;
    uint32_t coder_type = rcd_u(c, st);
    if (coder_type > 2) return 5;
    k->custom = 0;
    memcpy(k->one_state, one_state_default, 256);
    if (coder_type == 2) {
        for (int i = 1; i < 256; i++) {
            const int32_t v = (int32_t)one_state_default[i] + rcd_s(c, st);
            if (v < 0 || v > 0xFF) return 5;
            k->one_state[i] = (uint8_t)v;
        }
        k->custom = 1;
        coder_type = 1;
    }
    if (coder_type != 1) return 20;                    /* Golomb-Rice: a valid stream, outside this oracle */
    const uint32_t colorspace = rcd_u(c, st);
    if (colorspace > 1) return 6;
    if (k->version) {
        k->bps = rcd_u(c, st);
        if (k->bps > 64) return 6;
        if (!k->bps) k->bps = 8;
    } else k->bps = 8;
    const int chroma = rcd_b(c, st);
    const uint32_t hsub = rcd_u(c, st), vsub = rcd_u(c, st);
    k->alpha = rcd_b(c, st);
    if (k->version > 1) {
        k->num_h = rcd_u(c, st) + 1;
        k->num_v = rcd_u(c, st) + 1;
        k->nsets = rcd_u(c, st);
        if (k->nsets > 8) return 7;
    } else { k->num_h = k->num_v = 1; k->nsets = 1; }
    for (uint32_t i = 0; i < k->nsets; i++) {
        int32_t scale = 1;
        for (int j = 0; j < 5; j++) {
            uint8_t qst[CONTEXT_SIZE]; memset(qst, 128, sizeof qst);
            int32_t v = 0;
            int16_t* q = k->qs[i].q[j];
            for (uint64_t kk = 0; kk < 128;) {      /* size_t in the reference: len_minus1 reaches 2^32 - 1 and must not wrap */
                const uint32_t len1 = rcd_u(c, qst);
                if (kk + len1 >= 128) return 9;
                for (uint32_t a = 0; a <= len1; a++, kk++) q[kk] = (int16_t)(scale * v);
                v++;
            }
            for (int a = 1; a < 128; a++) q[256 - a] = (int16_t)-q[a];
            q[128] = (int16_t)-q[127];
            scale *= 2 * v - 1;
            if (scale > 32768) return 10;
        }
        k->qs[i].context_count = (uint32_t)((scale + 1) >> 1);
    }
    for (uint32_t i = 0; i < k->nsets; i++) {
        int coded = 0;
        if (k->version >= 3) coded = rcd_b(c, st);
        if (coded) {
            const size_t n = (size_t)k->qs[i].context_count * CONTEXT_SIZE;
            k->init_own[i] = malloc(n);
            for (size_t j = 0; j < n; j++) k->init_own[i][j] = (uint8_t)rcd_s(c, st);
            k->init[i] = k->init_own[i];
        }
    }
    if (k->version >= 3) {
        k->ec = rcd_u(c, st);
        if (k->ec > 1) return 13;
        k->intra = 0;
        if (k->micro) { k->intra = rcd_u(c, st); if (k->intra > 1) return 14; }
    } else { k->ec = 0; k->intra = 0; }
    k->rgb = colorspace == 1;
    if (hsub || vsub || chroma != k->rgb) return 21;   /* subsampled / planar YUV: valid streams, outside this oracle */
    k->planes = k->rgb ? (k->alpha ? 4 : 3) : (k->alpha ? 2 : 1);
    k->overflow16 = (!k->rgb && k->bps == 16);
    k->bits = k->rgb ? k->bps + 1 : (k->bps <= 8 ? 8 : k->bps);
    k->set_index_count = k->rgb ? k->planes - 1 : 2 + (k->alpha ? 1 : 0);
    return 0;
}
/* does the stream describe pictures of this pixel format? */
static int ctx_matches(const codec_ctx* k, uint32_t pixfmt)
{
    return k->rgb == is_rgb(pixfmt) && k->alpha == has_alpha(pixfmt) && k->bps == ffv1o_bits_per_raw_sample(pixfmt) && k->planes == ffv1o_plane_count(pixfmt);
}

/* samples of one slice: SliceContent (FFV1_Slice.cpp:321-444) + Line (:447-472) with the end bit and the underrun / junk tests
 * (:286-299, :334-345); idx = the slice's own quant_table_set_index values */
static int decode_content(const ffv1o_params* p, const codec_ctx* k, rc_dec* c, const uint32_t idx[3], uint32_t x0, uint32_t y0, uint32_t w, uint32_t h,
                          size_t coded_size, int32_t* const planes[4])
{
    const int32_t mask = (int32_t)(((uint32_t)1 << k->bits) - 1);
    uint8_t (*states[3])[CONTEXT_SIZE] = { 0, 0, 0 };
    slice_states(k, idx, states);
    int32_t* sb = calloc((size_t)2 * k->planes * (w + 3), sizeof(int32_t));
    int32_t* sample[4][2];
    for (uint32_t pl = 0; pl < k->planes; pl++) { sample[pl][0] = sb + 2 * pl * (w + 3) + 2; sample[pl][1] = sample[pl][0] + w + 3; }
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t pl = 0; pl < k->planes; pl++) {
            int32_t* t = sample[pl][0]; sample[pl][0] = sample[pl][1]; sample[pl][1] = t;
            sample[pl][1][-1] = sample[pl][0][0];
            sample[pl][0][w] = sample[pl][0][w - 1];
            const uint32_t g = k->rgb ? (pl + 1) >> 1 : 0;
            uint8_t (*st)[CONTEXT_SIZE] = states[g];
            const quant_set* qs = &k->qs[idx[g]];
            const int is5 = qs->q[3][127] != 0;
            int32_t* dst = planes[pl] + (size_t)(y0 + y) * p->width + x0;
            /* slice::Line, FFV1_Slice.cpp:447-472 */
            for (uint32_t x = 0; x < w; x++) {
                int32_t* s1 = sample[pl][1] + x; int32_t* s0 = sample[pl][0] + x;
                const int32_t LT = s0[-1], T = s0[0], RT = s0[1], L = s1[-1];
                int32_t ctx = qs->q[0][(L - LT) & 0xFF] + qs->q[1][(LT - T) & 0xFF] + qs->q[2][(T - RT) & 0xFF];
                if (is5) ctx += qs->q[3][(s1[-2] - L) & 0xFF] + qs->q[4][(s1[0] - T) & 0xFF];
                int32_t v;
                if (k->overflow16) v = median3((int16_t)L, (int16_t)L + (int16_t)T - (int16_t)LT, (int16_t)T);
                else v = median3(L, L + T - LT, T);
                if (ctx >= 0) v += rcd_s(c, st[ctx]); else v -= rcd_s(c, st[-ctx]);
                *s1 = v & mask;
                dst[x] = *s1;
            }
        }
    free(sb);
    for (uint32_t i = 0; i < k->set_index_count; i++) free(states[i]);
    { uint8_t es = 129; rcd_b(c, &es); }                        /* FFV1_Slice.cpp:336-340 */
    if (c->cur - (c->mask < 0x100 ? 0 : 1) > c->end) return 14; /* IsUnderrun */
    if (rcd_bytes_used(c) < coded_size) return 15;              /* FFV1-SLICE-JUNK, :297-299 */
    return 0;
}

static int decode_slice(const ffv1o_params* p, const codec_ctx* k, const uint8_t* buf, size_t size, int first, int32_t* const planes[4])
{
    const size_t tail = k->ec ? 8 : 3;
    if (size < tail) return 10;
    if (k->ec && ffv1o_crc32(buf, size)) return 11;             /* FFV1_Slice.cpp:247-249 */
    rc_dec c; rcd_init(&c, buf, size - tail);
    if (first) { uint8_t ks = 128; rcd_b(&c, &ks); }
    rc_tables_from(c.one_state, c.zero_state, k->one_state);    /* :254-255 */
    uint8_t hs[CONTEXT_SIZE]; memset(hs, 128, sizeof hs);
    uint32_t sx = rcd_u(&c, hs), sy = rcd_u(&c, hs);
    uint32_t sw1 = rcd_u(&c, hs), sh1 = rcd_u(&c, hs);
    if (sx >= k->num_h || sy >= k->num_v || sw1 || sh1) return 12;
    uint32_t idx[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < k->set_index_count; i++) { idx[i] = rcd_u(&c, hs); if (idx[i] >= k->nsets) return 13; }   /* :159-168 */
    (void)rcd_u(&c, hs); (void)rcd_u(&c, hs); (void)rcd_u(&c, hs);
    uint32_t x0, y0, w, h;
    slice_rect(p, k->num_h, k->num_v, sx, sy, &x0, &y0, &w, &h);
    int r = decode_content(p, k, &c, idx, x0, y0, w, h, size - tail, planes);
    if (r) return r;
    if (k->ec && buf[size - 5]) return 16;                       /* error_status */
    return 0;
}

/* ffv1_frame::Process (FFV1_Frame.cpp:134-228) for the stream k describes */
static int decode_frame_ctx(const ffv1o_params* p, codec_ctx* k, const uint8_t* pkt, size_t size, int32_t* const planes[4])
{
    { rc_dec c; rcd_init(&c, pkt, size); uint8_t ks = 128; if (!rcd_b(&c, &ks)) return 1; }     /* keyframe bit, :148-156 (every frame is one: -g 1) */
    if (k->version <= 1) {
        /* no configuration record: one slice = the packet, its parameters in front of the samples (slice::Parse, FFV1_Slice.cpp:224-268) */
        rc_dec c; rcd_init(&c, pkt, size);
        { uint8_t ks = 128; rcd_b(&c, &ks); }
        codec_ctx* s = calloc(1, sizeof *s);
        int r = parse_parameters(&c, s, 0);
        if (!r && !ctx_matches(s, p->pixfmt)) r = 8;
        if (!r) {
            rc_tables_from(c.one_state, c.zero_state, s->one_state);
            const uint32_t idx[3] = { 0, 0, 0 };
            r = decode_content(p, s, &c, idx, 0, 0, p->width, p->height, size, planes);
        }
        ctx_free(s);
        return r;
    }
    const size_t tail = k->ec ? 8 : 3;
    /* split from the tail, FFV1_Frame.cpp:177-198 */
    size_t pos = size; uint32_t count = 0; int err = 0;
    while (pos && !err) {
        if (pos < tail) { err = 2; break; }
        size_t s = (((size_t)pkt[pos - tail]) << 16) | (((size_t)pkt[pos - tail + 1]) << 8) | pkt[pos - tail + 2];
        s += tail;
        if (s > pos) { err = 3; break; }
        pos -= s;
        err = decode_slice(p, k, pkt + pos, s, pos == 0, planes);
        count++;
    }
    if (!err && count != k->num_h * k->num_v) err = 4;
    return err;
}

int ffv1o_decode_frame(const ffv1o_params* p, const uint8_t* pkt, size_t size, int32_t* const planes[4])
{
    codec_ctx* k = ctx_from_params(p);
    int err = decode_frame_ctx(p, k, pkt, size, planes);
    ctx_free(k);
    return err;
}

int ffv1o_decode_payload(const ffv1o_params* p, const uint8_t* pkt, size_t size, uint8_t* payload, size_t line_bytes)
{
    size_t n = (size_t)p->width * p->height;
    int32_t* planes[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < ffv1o_plane_count(p->pixfmt); i++) planes[i] = calloc(n, sizeof(int32_t));
    int r = ffv1o_decode_frame(p, pkt, size, planes);
    if (!r) ffv1o_pack(p, planes, payload, line_bytes);
    for (int i = 0; i < 4; i++) free(planes[i]);
    return r;
}

/* ffv1_frame::OutOfBand (FFV1_Frame.cpp:105-131): CRC, then parameters::Parse */
static int parse_record(const uint8_t* rec, size_t size, codec_ctx* k)
{
    if (size < 5) return 1;
    if (ffv1o_crc32(rec, size)) return 2;                          /* FFV1_Frame.cpp:116 */
    rc_dec c; rcd_init(&c, rec, size - 4);
    return parse_parameters(&c, k, 1);
}

/* The decoder as the reference runs it: everything about the stream comes from the stream -- the configuration record (rec_size > 0) or
 * the header inside the packet (rec_size == 0) -- and only width, height, pixfmt and flags from p.  0 ok. */
int ffv1o_decode_stream(const ffv1o_params* p, const uint8_t* rec, size_t rec_size, const uint8_t* pkt, size_t size, uint8_t* payload, size_t line_bytes)
{
    codec_ctx* k = calloc(1, sizeof *k);
    int r = 0;
    if (rec_size) { r = parse_record(rec, rec_size, k); if (!r && !ctx_matches(k, p->pixfmt)) r = 8; }
    else k->version = 1;                                         /* (whatever the header in the packet then says: 0 or 1) */
    size_t n = (size_t)p->width * p->height;
    int32_t* planes[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < ffv1o_plane_count(p->pixfmt); i++) planes[i] = calloc(n, sizeof(int32_t));
    if (!r) r = decode_frame_ctx(p, k, pkt, size, planes);
    if (!r) ffv1o_pack(p, planes, payload, line_bytes);
    for (int i = 0; i < 4; i++) free(planes[i]);
    ctx_free(k);
    return r;
}

/* The record against what this oracle's encoder would have written for p (and num_h, num_v, ec out of the record into p).  0 = same. */
int ffv1o_parse_config_record(const uint8_t* rec, size_t size, ffv1o_params* p)
{
    codec_ctx* s = calloc(1, sizeof *s);
    int r = parse_record(rec, size, s);
    if (!r) {
        p->num_h_slices = s->num_h; p->num_v_slices = s->num_v; p->ec = s->ec;
        codec_ctx* k = ctx_from_params(p);
        if (!ctx_matches(s, p->pixfmt)) r = 8;
        else if (s->version != k->version || s->custom != k->custom || memcmp(s->one_state + 1, k->one_state + 1, 255)) r = 5;
        else if (s->nsets != k->nsets) r = 7;
        else if (s->intra != k->intra) r = 14;
        for (uint32_t i = 0; !r && i < s->nsets; i++) {
            if (memcmp(s->qs[i].q, k->qs[i].q, sizeof s->qs[i].q)) r = 10;
            else if (s->qs[i].context_count != k->qs[i].context_count) r = 11;
            else if ((s->init[i] != NULL) != (k->init[i] != NULL)) r = 12;
            else if (s->init[i] && memcmp(s->init[i], k->init[i], (size_t)s->qs[i].context_count * CONTEXT_SIZE)) r = 12;
        }
        ctx_free(k);
    }
    ctx_free(s);
    return r;
}

int ffv1o_slices_to_grid(uint32_t n, uint32_t* num_h, uint32_t* num_v)
{
    if (n == 1) { *num_h = 1; *num_v = 1; return 0; }
    for (uint32_t v = 2; v < 32; v++)
        for (uint32_t h = v; h < 2 * v; h++)
            if (h * v == n) { *num_h = h; *num_v = v; return 0; }
    return -1;
}
