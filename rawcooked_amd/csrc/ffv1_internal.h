// ffv1_internal.h -- what the upload/download pipeline (pipeline.hip) needs from the encoder beyond the C ABI: its staging areas on
// the device and one ordering hook.  Not part of include/rcgpu.h.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include "rcgpu.h"

namespace rc {

struct enc_staging {
    uint8_t* d_in;            // max_batch payload slots, in_stride apart: read by k_model only, free again once k_model has run
    size_t   in_stride, payload_bytes;
    uint8_t* d_packets;       // max_batch packets, packet_stride apart: written by k_gather at the very end of a batch
    size_t   packet_stride;
    uint64_t* d_psizes;       // max_batch packet sizes
    uint32_t* d_err;          // [0] error flags, [1] carry events of the batch
    void*    compute_stream;  // the encoder's own stream (hipStream_t)
    uint32_t max_batch;
    int      device;
};
// Allocates the staging areas on first use.
int  ffv1_staging(rcgpu_ffv1* e, enc_staging* out);
// The next rcgpu_ffv1_encode_device call makes k_gather wait for this event (a hipEvent_t already recorded): the download of the
// previous batch's packets out of d_packets.  nullptr = no wait.
void ffv1_set_gather_wait(rcgpu_ffv1* e, void* hip_event);
// Deferred gather: with `on`, rcgpu_ffv1_encode_device stops after k_scan (slice sizes and packet layout are known, the slices still sit
// in their own buffers) and ffv1_gather() compacts them into the packets later, on the same stream.  The pipeline uses the gap to issue
// the previous batch's downloads AFTER the next batch has been started: the device never waits for the host between two batches.
void ffv1_set_defer_gather(rcgpu_ffv1* e, bool on);
int  ffv1_gather(rcgpu_ffv1* e, void* d_packets, size_t packet_stride, void* hip_stream);
// Run-on mode (rcgpu_ffv1_set_run_on): the next rcgpu_ffv1_encode_device models its batch as soon as this event (a hipEvent_t already
// recorded: the frames' uploads) has happened, not behind everything the caller's stream carries -- the previous batch's gather is there.
void ffv1_set_input_event(rcgpu_ffv1* e, void* hip_event);
// The configuration record and the worst-case packet size of an encoder of this configuration, without a device (the container's header
// can be written, and its file laid out, while the encoders are still being created).
std::vector<uint8_t> ffv1_config_record_for(const rcgpu_ffv1_config& cfg);
size_t ffv1_max_packet_bytes_for(const rcgpu_ffv1_config& cfg);
// true: an encoder of this configuration keeps its slice byte buffers inside the symbol buffer (no memory of their own)
bool ffv1_overlays_slice_buffers(const rcgpu_ffv1_config& cfg);
// Per-kernel device time of the call before the last one (see rcgpu_ffv1_last_kernel_times).
int  ffv1_prev_kernel_times(const rcgpu_ffv1* e, const char** names, float* ms, int cap);
int  ffv1_prev_timeline(const rcgpu_ffv1* e, float* t6);      // see ffv1_gpu.hip; for RCGPU_TRACE
// A small copy done by a kernel on `hip_stream` (hipStream_t) instead of a copy engine: device <-> pinned host memory, 8-byte granules.
// Control data must not queue behind the payload copies that keep the engines busy.  Returns a hipError_t as int.
int  copy_by_kernel_on(void* dst, const void* src, size_t bytes, void* hip_stream);
// Text for the device error word (0 = none).
const char* ffv1_error_flags_text(uint32_t flags);

}  // namespace rc
