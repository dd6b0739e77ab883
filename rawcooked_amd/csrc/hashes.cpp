// hashes.cpp -- MD5 (RFC 1321) and the FFV1 flavour of CRC-32, host side.
//
// MD5: what the reference computes per source file (Lib/Utils/FileIO/Input_Base.cpp:54-81 via
// Lib/ThirdParty/md5/md5.c) and what FLAC's STREAMINFO carries.  CRC-32: Lib/Utils/CRC32/ZenCRC32.cpp:1097-1135
// (poly 0x04C11DB7, MSB first, init 0, no final xor) -- used for the configuration record; slice CRCs are
// computed on the device.
#include "rc_common.h"

namespace {
struct md5_ctx { uint32_t a, b, c, d; uint64_t len; uint8_t buf[64]; size_t fill; };

inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

void md5_block(md5_ctx& c, const uint8_t* p)
{
    static const uint32_t K[64] = {
        0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
        0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
        0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
        0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391 };
    static const uint8_t S[64] = { 7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22, 5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,
                                   4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23, 6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21 };
    uint32_t m[16];
    for (int i = 0; i < 16; i++) m[i] = rc::rd32(p + 4 * i, false);
    uint32_t a = c.a, b = c.b, cc = c.c, d = c.d;
    for (int i = 0; i < 64; i++) {
        uint32_t f; int g;
        if (i < 16) { f = (b & cc) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & cc); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ cc ^ d; g = (3 * i + 5) & 15; }
        else { f = cc ^ (b | ~d); g = (7 * i) & 15; }
        const uint32_t t = d; d = cc; cc = b;
        b = b + rol(a + f + K[i] + m[g], S[i]);
        a = t;
    }
    c.a += a; c.b += b; c.c += cc; c.d += d;
}
}  // namespace

extern "C" void rcgpu_md5(const uint8_t* data, size_t size, uint8_t out[16])
{
    md5_ctx c{ 0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476, uint64_t(size) * 8, {0}, 0 };
    size_t i = 0;
    for (; i + 64 <= size; i += 64) md5_block(c, data + i);
    uint8_t tail[128] = { 0 };
    const size_t rem = size - i;
    if (rem) memcpy(tail, data + i, rem);
    tail[rem] = 0x80;
    const size_t tl = rem < 56 ? 64 : 128;
    for (int k = 0; k < 8; k++) tail[tl - 8 + k] = uint8_t(c.len >> (8 * k));
    md5_block(c, tail);
    if (tl == 128) md5_block(c, tail + 64);
    const uint32_t w[4] = { c.a, c.b, c.c, c.d };
    for (int k = 0; k < 16; k++) out[k] = uint8_t(w[k / 4] >> (8 * (k % 4)));
}

namespace {
struct crc_table {
    uint32_t t[256];
    constexpr crc_table() : t{}
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i << 24;
            for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
            t[i] = c;
        }
    }
};
constexpr crc_table kCrc{};      // built at compile time: concurrent callers (one worker per device) share a constant
}

extern "C" uint32_t rcgpu_crc32_ffv1(const uint8_t* d, size_t n)
{
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++) c = (c << 8) ^ kCrc.t[(c >> 24) ^ d[i]];
    return c;
}
