// pipeline.h -- the upload / encode / download pipeline behind rcgpu_encode and rcgpu_ffv1_encode_sequence.
//
// What it stands for in the reference: the ffmpeg subprocess started at CLI/Output.cpp:356 reads the files RAWcooked analysed
// (Lib/Utils/FileIO/FileIO.cpp:274 maps them), codes them and writes the Matroska file, all inside one process.  Here the three
// parts run side by side per device ("lane"):
//
//   reader threads ---> pinned upload slots ---H2D (copy stream)---> d_in ---k_model ... k_gather---> d_packets
//                                                                                                        |
//   writer threads <--- pinned download ring <---D2H (second copy stream)--------------------------------+
//
// d_in is read by k_model only (the first ~5 % of a batch), so the next batch is uploaded while the current one is resolved and
// range-coded; d_packets is written by k_gather only (the last ~2 %), so the previous batch is downloaded meanwhile and k_gather
// waits for that download's event.  One set of device buffers, three batches in flight.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>
#include "rcgpu.h"

namespace rc {

struct pipe_video {                 // one picture sequence = one FFV1 track
    rcgpu_ffv1_config cfg{};        // device, max_batch are the pipeline's to fill
    uint64_t frames = 0;            // how many frames of it the job holds (bounds the batch)
};
struct pipe_frame { uint32_t video; uint64_t index; };     // a frame of `video`; the job lists them in OUTPUT order

struct pipe_io {
    // reader threads, concurrently, any order: put the payload of `f` (payload_bytes of its video) at dst (pinned).  0 = ok.
    std::function<int(const pipe_frame& f, uint8_t* dst)> read;
    // optional, instead of read: the payload of `f` lies in PINNED host memory already (hipHostMalloc / hipHostRegister) and stays there
    // until its batch has been modelled -- it is uploaded from where it is: no upload slots, no reader threads, no copy.  nullptr = error.
    std::function<const uint8_t*(const pipe_frame& f)> locate;
    // one thread, output order: where do the `size` bytes of this packet go?  nullptr = "hand them to done()"
    std::function<uint8_t*(const pipe_frame& f, size_t size)> place;
    // writer threads, concurrently: the packet is at `data` (== the place, or a pinned buffer valid during the call).  0 = ok.
    std::function<int(const pipe_frame& f, const uint8_t* data, size_t size)> done;
    // optional, writer threads: how a packet gets to its place (default: memcpy); a mapped file wants its range pre-faulted first.  0 = ok.
    std::function<int(uint8_t* dst, const uint8_t* src, size_t size)> copy;
    // optional, lane thread, after a batch has run and while its payloads are still on the device (e.g. frame checksums).
    // Setting it serialises upload and encoding of consecutive batches.
    std::function<int(uint32_t video, rcgpu_ffv1* enc, uint64_t first_index, uint32_t n)> after_batch;
};

struct pipe_options {
    int device_first = 0, device_count = 0;     // 0 = all visible
    uint32_t lanes_per_device = 0;               // encoder instances per device whose batches run staggered; 0 = 1
    uint32_t batch = 0;                          // frames per batch; 0 = from free device memory and the sequence length
    uint32_t readers = 0, writers = 0;           // host threads; 0 = automatic
    uint32_t in_slots = 0;                       // pinned upload slots; 0 = automatic
    uint64_t out_ring_bytes = 0;                 // pinned download ring per lane; 0 = automatic
    uint32_t device_aliases = 0;                 // test hook: every physical device presented this many times (0, 1 = as they are)
    uint32_t copy_streams = 0;                   // copy streams per direction and lane; 0 = automatic (two)
    uint32_t numa = 0;                           // 0 = lanes grouped by their device's NUMA node, host threads and pinned memory bound to it; 1 = off; 2 = test hook
    uint32_t run_on = 0;                         // 1 = the encoders run on from batch to batch where device memory allows (rcgpu_ffv1_set_run_on); 0 = one batch at a time
    bool trace = false;
};

struct pipe_stats {
    double seconds = 0, first_packet_seconds = 0, prepare_seconds = 0;
    uint64_t frames = 0, payload_bytes = 0, packet_bytes = 0, batches = 0;
    uint32_t batch_frames = 0, lanes = 0, readers = 0, writers = 0;
    double device_busy_seconds = 0;             // lane 0: first encode call .. completion of its last batch (its device has a batch in flight throughout)
    double steady_frames_per_second = 0;        // frames of all batches but the first / time from the first batch's completion to the last's
    double reads_done_seconds = 0, last_batch_seconds = 0;
    double upload_wait_seconds = 0;             // lane 0: time its thread waited for the readers to fill a slot
    double read_call_seconds = 0, write_call_seconds = 0;   // average duration of one read callback / one packet copy + done callback
    double h2d_span_seconds = 0;                // lane 0: sum over batches of first upload start .. last upload end (device clock)
    uint32_t groups = 0;                        // host-side groups (NUMA nodes the lanes' devices hang on)
    int32_t lane_device[16], lane_node[16], lane_pinned_node[16];   // per lane: device, its NUMA node, the node its pinned ring was found on (-1 unknown)
    pipe_stats() { for (int i = 0; i < 16; i++) lane_device[i] = lane_node[i] = lane_pinned_node[i] = -1; }
};

class pipeline {
public:
    pipeline();
    ~pipeline();
    // Creates the encoders (one per lane and video) and sizes the batches.  Errors through rc::fail.
    int prepare(const std::vector<pipe_video>& videos, const pipe_options& opt);
    rcgpu_ffv1* encoder(uint32_t video) const;           // lane 0's: configuration record, max packet size
    uint32_t batch_frames(uint32_t video) const;
    int run(const std::vector<pipe_frame>& frames, const pipe_io& io, pipe_stats* stats);
    struct impl;
private:
    std::unique_ptr<impl> p;
};

// Device memory one frame in flight needs inside an encoder of this configuration (symbols, states, stream windows, slice and
// packet buffers, payload), without creating one.
uint64_t ffv1_device_bytes_per_frame(const rcgpu_ffv1_config& cfg, bool run_on = false);   // run_on: with the encoder's second bank

}  // namespace rc
