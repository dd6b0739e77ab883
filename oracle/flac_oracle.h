/* flac_oracle.h -- CPU oracle for the WAV -> FLAC path.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference has no FLAC encoder (it runs FFmpeg's, Source/CLI/Output.cpp:356) but vendors libFLAC's
 * DECODER (Source/Lib/ThirdParty/flac/src/libFLAC/stream_decoder.c) behind flac_wrapper
 * (Source/Lib/CoDec/Wrapper.cpp:131-373).  This oracle therefore holds
 *   - a decoder that restates the bitstream exactly as stream_decoder.c parses it (:2012-2788), and
 *   - an encoder that states, in scalar C, the deterministic encoding rule the device kernels implement
 *     (integer-exact windowed autocorrelation, Levinson-Durbin in IEEE double with a fixed operation order,
 *     exhaustive order search by integer cost, partitioned Rice), so that device bytes == oracle bytes.
 * FLAC leaves predictor choice to the encoder; what is pinned is losslessness (decode == PCM) and that the
 * reference binary accepts the stream (tests/test_reference_check.py).
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t channels, sample_rate, bits_per_sample;   /* 8 / 16 / 24 */
    uint32_t block_size;                               /* 0: largest standard size <= rate * 105 ms (4608 @ 48 kHz) */
    uint32_t max_lpc_order;                            /* 0..32; 0 = fixed predictors only */
} flaco_params;

uint32_t flaco_default_block_size(uint32_t sample_rate);

/* Encode interleaved little-endian PCM exactly as found in a WAV data chunk (8-bit = offset binary).
 * out receives the concatenated frames, frame_sizes the size of each; returns the number of frames, or -1. */
long flaco_encode(const flaco_params* p, const uint8_t* pcm, uint64_t pcm_bytes, uint8_t* out, size_t cap,
                  uint32_t* frame_sizes, size_t frame_cap);
/* "fLaC" + STREAMINFO block (Matroska CodecPrivate) for a stream encoded with flaco_encode. Returns 42. */
size_t flaco_codec_private(const flaco_params* p, uint64_t total_samples, uint32_t min_frame, uint32_t max_frame,
                           const uint8_t md5[16], uint8_t* out);
/* Decode concatenated frames back to WAV-layout PCM bytes.  Returns bytes written, or -(error code). */
long long flaco_decode(const flaco_params* p, const uint8_t* frames, size_t size, uint8_t* pcm, size_t cap);

uint8_t  flaco_crc8(const uint8_t* d, size_t n);    /* poly 0x07, init 0  (libFLAC crc.c:366) */
uint16_t flaco_crc16(const uint8_t* d, size_t n);   /* poly 0x8005, init 0 (crc.c:376) */
void     flaco_md5(const uint8_t* d, size_t n, uint8_t out[16]);

#ifdef __cplusplus
}
#endif
#endif
