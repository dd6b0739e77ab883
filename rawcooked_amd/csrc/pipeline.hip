// pipeline.hip -- host side only (HIP runtime calls, no kernels): see pipeline.h for the picture.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include "pipeline.h"
#include "ffv1_internal.h"
#include "ffv1_host.h"
#include "rc_common.h"

namespace rc {

namespace {

using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

constexpr size_t kUploadGroup = 8, kDownloadGroup = 16;       // copies per completion event
constexpr uint32_t kCopyStreams = 2;                          // copy streams per direction and lane

struct batch_t { uint32_t video; size_t first, n; int lane; };       // frames[first .. first+n) of the output order

// ---- NUMA.  A lane moves ~30 GB/s each way through pinned host memory, and every byte crosses that memory about four times (reader copy
// in, DMA out, DMA in, writer copy out): ~118 GB/s of DRAM traffic per GPU.  On a two-socket host with eight GPUs
// that only works when a lane's pinned slots, its download ring and the threads that fill and drain them live on the socket its GPU
// hangs on; across the inter-socket link eight lanes would meet its limit long before PCIe's.  So lanes are grouped by the NUMA node of
// their device (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node), every group has its own slot pool, readers and writers,
// bound to the node's CPUs, and pinned memory is allocated by a thread that runs there (first touch) with the lane's device current
// (the runtime then prefers the device's node as well).
struct node_cpus { cpu_set_t set; int count = 0; };

int device_numa_node(int device)
{
    char id[64] = {};
    if (hipDeviceGetPCIBusId(id, int(sizeof id), device) != hipSuccess) return -1;
    for (char* c = id; *c; c++) if (*c >= 'A' && *c <= 'F') *c = char(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", id);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

int host_node_count()
{
    int n = 0;
    for (;; n++) { char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", n); if (access(path, R_OK) != 0) break; }
    return n;
}

// the CPUs of a node that this process may run on (its affinity mask at the time of the call: cgroups and taskset are respected)
node_cpus cpus_of_node(int node, const cpu_set_t& allowed)
{
    node_cpus out; CPU_ZERO(&out.set);
    char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char text[4096] = {};
    if (!fgets(text, sizeof text, f)) text[0] = 0;
    fclose(f);
    for (char* p = text; *p;) {
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (CPU_ISSET(int(c), &allowed)) { CPU_SET(int(c), &out.set); out.count++; }
        p = *e == ',' ? e + 1 : e;
        if (*p == '\n') break;
    }
    return out;
}

void bind_this_thread(const node_cpus& c) { if (c.count > 0) (void)sched_setaffinity(0, sizeof c.set, &c.set); }

// the node a page of this process lies on, as the kernel reports it (move_pages with no target nodes only asks); -1 = unknown
int node_of_page(const void* p)
{
    void* page = reinterpret_cast<void*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(4095));
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) != 0) return -1;
    return status;
}

struct out_entry {            // one packet on its way down
    size_t frame = 0; uint8_t* src = nullptr; size_t size = 0; int chunk = -1; hipEvent_t ev = nullptr; int device = 0, lane = 0;
    int* ev_users = nullptr;      // the event is shared by a group of packets: the last one done with it returns it
    uint8_t* dst = nullptr;
};

struct lane_t {
    int id = 0, device = 0;
    int node = -1, group = 0;                     // NUMA node of the device (-1 unknown), index of the host-side group the lane belongs to
    int pinned_node = -1;                         // where the kernel says the lane's first ring chunk lies
    std::vector<rcgpu_ffv1*> enc;                 // per video
    std::vector<hipStream_t> cin, cout;          // copy streams per direction: groups of copies are dealt to them in turn
    std::vector<hipEvent_t> join_ev;             // one per extra stream: joins it to stream 0 of its direction at the end of a batch
    size_t up_turn = 0, dn_turn = 0;
    std::vector<hipEvent_t> dl_done;              // per video: the download of the last batch out of that encoder's d_packets
    std::vector<bool> dl_valid;
    hipEvent_t ev_up = nullptr, ev_up0 = nullptr, ev_done[2] = { nullptr, nullptr };     // ev_up0 / ev_up time the uploads of a batch
    double upload_wait = 0, h2d_span = 0, copy_calls = 0, dl_calls = 0, dl_wait = 0;
    bool up_span_valid = false;                   // ev_up0 / ev_up hold a recorded pair
    // download ring: pinned chunks, a chunk is re-entered when nothing in it is outstanding
    std::vector<uint8_t*> chunks; std::vector<int> outstanding; size_t chunk_bytes = 0, max_chunks = 0; int cur = -1; size_t cur_off = 0;
    std::deque<out_entry> outq;                   // issued downloads in frame order, consumed by the placer
    std::vector<hipEvent_t> free_events;
    struct pending_up { std::vector<uint8_t*> slots; hipEvent_t ev; };     // a group of copies and the event recorded behind the last of them
    std::deque<pending_up> pending;                // uploads in flight, in issue order: their slots return to the pool once the copy is done
    uint64_t* h_sizes = nullptr; uint32_t* h_err = nullptr; size_t h_sizes_stride = 0;     // pinned, two batches' worth
};

}  // namespace

struct pipeline::impl {
    std::vector<pipe_video> videos;
    pipe_options opt;
    std::vector<lane_t> lanes;
    std::vector<uint32_t> F;                      // batch frames per video
    std::vector<size_t> payload;                  // payload bytes per video
    std::vector<size_t> max_packet;
    size_t slot_bytes = 0;
    double prepare_seconds = 0;

    // ---- run state (one big lock: events here are per frame, a few thousand per second)
    std::mutex m;
    // one condition per kind of waiter (a single one woke ~20 threads for every frame event)
    std::condition_variable cv_slots /* readers: a free upload slot */, cv_ready /* lanes: a frame was read */, cv_ring /* lanes: ring space, chunks */,
                            cv_out /* placer: a download was issued */, cv_jobs /* writers: a packet was placed */;
    void wake_all() { cv_slots.notify_all(); cv_ready.notify_all(); cv_ring.notify_all(); cv_out.notify_all(); cv_jobs.notify_all(); }
    int error = 0; std::string error_msg;
    // host-side groups: the lanes of one NUMA node share a pool of pinned upload slots, reader threads and writer threads, all of them on
    // that node.  One group when the host has one node (or the nodes are unknown).
    struct group_t {
        int node = -1; node_cpus cpus; int first_lane = 0; uint32_t lanes = 0;
        std::vector<uint8_t*> all_slots, free_slots; size_t slots_wanted = 0;
        std::vector<size_t> frames; size_t next_read = 0;      // the output frames its lanes code, in order; the next one a reader takes
        std::deque<out_entry> jobs;                            // placed packets of its lanes waiting for a writer
        uint32_t readers = 0, writers = 0; double reads_done = 0;
    };
    std::vector<group_t> groups;
    std::vector<uint8_t*> ready;                  // per output frame: the filled slot (nullptr until read)
    bool placer_finished = false;
    bool alloc_done = false;

    void set_error(int code, const char* msg)
    {
        std::lock_guard<std::mutex> l(m);
        if (!error) { error = code ? code : 1; error_msg = msg ? msg : ""; }
        wake_all();
    }
    bool failed() { std::lock_guard<std::mutex> l(m); return error != 0; }
    ~impl();
};

pipeline::impl::~impl()
{
    // Device buffers (140 GB: ~2 s) first, then the pinned ones (download ring + upload slots, ~7 GB: 0.5 s).  Side by side they
    // take longer than one after the other (measured: 4.0 against 3.3 s): both unmap address space of this process, as does the muxer
    // closing its mapped file meanwhile.
    const bool tr = getenv("RCGPU_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    for (lane_t& L : lanes) {
        (void)hipSetDevice(L.device);
        for (rcgpu_ffv1* e : L.enc) if (e) rcgpu_ffv1_destroy(e);
        if (tr) fprintf(stderr, "rcgpu trace: release: encoders of lane %d after %.3f s\n", L.id, since());
        for (hipEvent_t e : L.free_events) (void)hipEventDestroy(e);
        for (hipEvent_t e : L.dl_done) if (e) (void)hipEventDestroy(e);
        if (L.ev_up) (void)hipEventDestroy(L.ev_up);
        if (L.ev_up0) (void)hipEventDestroy(L.ev_up0);
        for (hipEvent_t e : L.ev_done) if (e) (void)hipEventDestroy(e);
        for (hipStream_t c : L.cin) if (c) (void)hipStreamDestroy(c);
        for (hipStream_t c : L.cout) if (c) (void)hipStreamDestroy(c);
        for (hipEvent_t e : L.join_ev) if (e) (void)hipEventDestroy(e);
        if (L.h_sizes) (void)hipHostFree(L.h_sizes);
        if (L.h_err) (void)hipHostFree(L.h_err);
    }
    for (lane_t& L : lanes) for (uint8_t* c : L.chunks) if (c) (void)hipHostFree(c);
    for (group_t& g : groups) for (uint8_t* s : g.all_slots) (void)hipHostFree(s);
    if (tr) fprintf(stderr, "rcgpu trace: release: pinned buffers after %.3f s\n", since());
}

pipeline::pipeline() : p(new impl) {}
pipeline::~pipeline() {}

rcgpu_ffv1* pipeline::encoder(uint32_t video) const { return p->lanes.empty() || video >= p->lanes[0].enc.size() ? nullptr : p->lanes[0].enc[video]; }
uint32_t pipeline::batch_frames(uint32_t video) const { return video < p->F.size() ? p->F[video] : 0; }

// devices and lanes a job of `longest` frames gets out of `devices` devices: a device per 8 frames at most, lanes_per_device (<= 4) lanes on each
static int effective_lanes(uint64_t longest, uint32_t devices, uint32_t lanes_per_device, int* devices_used)
{
    const uint64_t min_batch = 8, chunks = (longest + min_batch - 1) / min_batch;
    const int ndev_used = int(std::max<uint64_t>(1, std::min<uint64_t>(devices, chunks)));
    const int per_dev = int(std::max(1u, std::min(4u, lanes_per_device)));
    if (devices_used) *devices_used = ndev_used;
    return int(std::max<uint64_t>(ndev_used, std::min<uint64_t>(uint64_t(ndev_used) * per_dev, chunks)));
}

uint64_t ffv1_device_bytes_per_frame(const rcgpu_ffv1_config& c, bool run_on)
{
    if (c.pixfmt >= RCGPU_PIX_COUNT) return 0;
    const pix_desc& d = pix(c.pixfmt);
    const uint64_t px = uint64_t(c.width) * c.height, samples = px * d.planes;
    const uint64_t S = uint64_t(std::max(1u, c.num_h_slices)) * std::max(1u, c.num_v_slices);
    const uint64_t raw = payload_bytes(c.pixfmt, c.width, c.height, c.line_bytes, c.flags);
    // contexts: FFmpeg's maps give 5063 (5 inputs, > 8 bit) / 6561 (8 bit) / 365.. (3 inputs); the compact model 338
    const uint64_t nctx = c.context == 2 ? 338 : c.context == 1 ? (d.bits > 8 ? 5063 : 6561) : (d.bits > 8 ? 365 : 666);
    const uint64_t nsets = d.planes == 1 ? 1 : d.planes == 4 ? 3 : 2;
    const uint64_t states = nctx * nsets * 32 <= (48u << 10) ? 0 : S * nctx * nsets * 32;
    const uint64_t nseg = c.segments ? c.segments : std::max<uint64_t>(1, std::min<uint64_t>(32, samples / S / 1024));
    const uint64_t windows = samples * 35 * 8 / 7 * (nseg > 1 ? 3 : 1) / nseg;              // worst case: 35 decisions per sample, 64 bytes per 56 of them; three windows (split coder)
    const uint64_t ckpt = samples * 35 / 56 / 8 + S * nseg * 8;                              // split coder: 8 bytes per span of >= 8 pieces and slice
    const uint64_t cbuf = raw * 3 / 2 + S * ((256u << 10) + 4096 + 32);
    const uint64_t own_cbuf = ffv1_overlays_slice_buffers(c) ? 0 : cbuf;                    // large slices: the byte buffers lie inside the symbol buffer
    // run-on mode: the encoder's second bank (rcgpu_ffv1_set_run_on) -- symbols, context states, slice byte buffers once more
    const uint64_t second_bank = run_on ? samples * 4 + states + own_cbuf + S * 64 : 0;
    return samples * 4 + states + windows + ckpt + own_cbuf + cbuf /* packets */ + raw + (1u << 20) + second_bank;
}

}  // namespace rc
extern "C" uint64_t rcgpu_ffv1_device_bytes_per_frame(const rcgpu_ffv1_config* cfg, int run_on) { return cfg ? rc::ffv1_device_bytes_per_frame(*cfg, run_on != 0) : 0; }
namespace rc {

int pipeline::prepare(const std::vector<pipe_video>& videos, const pipe_options& opt)
{
    const auto t0 = clk::now();
    impl& s = *p;
    s.videos = videos; s.opt = opt;
    if (videos.empty()) return fail(1, "pipeline: no video");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(4, "no HIP device available -- rcgpu has no CPU encode path");
    // Test hook (rcgpu_sequence_options::device_aliases; tests/test_gpu_pipeline.py, bench.py --gpus N on a one-GPU box): every physical
    // device is presented k times, so that the lane-per-device path -- a lane with its own encoder, copy streams, events and ring per entry
    // of device_first/device_count, one placer across them -- runs on a box with a single GPU.  The lanes then share that GPU's memory:
    // the caller passes its own batch size.
    const int ndev_phys = ndev;
    if (opt.device_aliases > 1) ndev *= int(std::min(8u, opt.device_aliases));
    const int dev0 = std::max(0, opt.device_first);
    int cnt = opt.device_count > 0 ? opt.device_count : ndev - dev0;
    if (dev0 >= ndev || cnt <= 0) return fail(4, "device selection %d+%d is outside the %d visible devices", dev0, opt.device_count, ndev);
    cnt = std::min(cnt, ndev - dev0);
    uint64_t longest = 0;
    for (const pipe_video& v : videos) longest = std::max(longest, v.frames);
    // no more lanes than there is work for: a short job on an 8-GPU node uses the devices it can fill (effective_lanes: also behind rcgpu_sequence_plan)
    int ndev_used = 1;
    cnt = effective_lanes(longest, uint32_t(cnt), opt.lanes_per_device, &ndev_used);
    s.lanes.resize(size_t(cnt));
    s.F.assign(videos.size(), 1); s.payload.assign(videos.size(), 0); s.max_packet.assign(videos.size(), 0);
    for (int li = 0; li < cnt; li++) {
        lane_t& L = s.lanes[size_t(li)];
        L.id = li; L.device = (dev0 + li % ndev_used) % ndev_phys;
        if (hipSetDevice(L.device) != hipSuccess) return fail(4, "pipeline: cannot select device %d", L.device);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return fail(100, "pipeline: hipMemGetInfo failed");
        L.enc.assign(videos.size(), nullptr);
        for (size_t vi = 0; vi < videos.size(); vi++) {
            rcgpu_ffv1_config c = videos[vi].cfg;
            c.device = L.device;
            if (li == 0) {
                // frames in flight: what the device holds (85 % of its free memory, shared by the job's tracks), at most 336 frames of 64 slices
                // = 21504 slice chains -- where k_resolve's time, which grows with the batch, meets the serial range-coder chain of a slice,
                // which does not --, and evened out over the batches of the sequence.  The bound is in CHAINS: the 576
                // slices RAWcooked asks for at 4K (DPX.cpp:428-458) fill the device with 40 frames (571 frames/s; 546 with 256, whose 71 GB of
                // context states alone take 4.3 s to allocate).
                const uint64_t per = std::max<uint64_t>(1, ffv1_device_bytes_per_frame(c, opt.run_on == 1));
                const uint64_t share = videos.size() * uint64_t((cnt + ndev_used - 1) / ndev_used);      // encoders that will live on this device
                const uint64_t slices = std::max<uint64_t>(1, uint64_t(c.num_h_slices) * c.num_v_slices);
                const uint64_t by_chains = std::min<uint64_t>(336, std::max<uint64_t>(8, (336 * 64 + slices - 1) / slices));
                uint64_t f = opt.batch ? opt.batch : std::min<uint64_t>(std::max<uint64_t>(1, by_chains / uint64_t((cnt + ndev_used - 1) / ndev_used)), uint64_t(double(free_b) * 0.85 / double(share)) / per);
                f = std::max<uint64_t>(1, f);
                const uint64_t n = std::max<uint64_t>(1, videos[vi].frames);
                const uint64_t per_lane = (n + uint64_t(cnt) - 1) / uint64_t(cnt);
                if (!opt.batch) { const uint64_t nb = (per_lane + f - 1) / f; f = (per_lane + nb - 1) / nb; }
                s.F[vi] = uint32_t(std::min<uint64_t>(f, n));
            }
            c.max_batch = s.F[vi];
            rcgpu_ffv1* e = nullptr;
            if (int r = rcgpu_ffv1_create(&c, &e)) return r;
            L.enc[vi] = e;
            enc_staging sg;
            if (int r = ffv1_staging(e, &sg)) return r;
            // on request: batch k+1 is started while batch k is coded, where the device has room for the encoder's second bank (if not: one at a
            // time).  Not the default: between its first and last batch the pipeline gains 0.5 % (672.7 against 669 frames/s) for 75 GB
            if (opt.run_on == 1 && rcgpu_ffv1_set_run_on(e, 1) != 0) clear_error();
            s.payload[vi] = sg.payload_bytes; s.max_packet[vi] = sg.packet_stride;
        }
        // The two copy streams get priorities of their own.  Events and stream waits are barrier packets in a stream's HARDWARE queue, the
        // runtime has four of those per priority level, and streams of one level share them: with both copy streams at the default level
        // the uploads' barriers stood in one queue behind the downloads' -- measured: not one upload of batch k+1 completed before the last
        // download of batch k-1 had (0.33 s per batch).  A level per copy stream keeps their barriers apart from each other and from the
        // encoder's two compute streams.
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        // Two streams per direction: the copies of ONE stream run strictly one after the other, and beside the kernels that chain got 26-36
        // GB/s per direction while the engines had room for twice as much (a pump of copies beside the pipeline moved 44 GB/s more, and
        // the pipeline itself got faster).  Measured, frames/s between the first and the last batch: 529 with one stream per direction,
        // 643 with two (= the device-resident rate), 649 with three, 460 with four.
        uint32_t ncopy = cnt > ndev_used ? 1 : kCopyStreams;         // several lanes on a device: their streams add up, and four per direction were worse
        if (opt.copy_streams) ncopy = std::min(8u, opt.copy_streams);
        L.cin.assign(ncopy, nullptr); L.cout.assign(ncopy, nullptr); L.join_ev.assign(2 * ncopy, nullptr);
        for (uint32_t k = 0; k < ncopy; k++)
            if (hipStreamCreateWithPriority(&L.cin[k], hipStreamNonBlocking, prio_greatest) != hipSuccess ||
                hipStreamCreateWithPriority(&L.cout[k], hipStreamNonBlocking, prio_least) != hipSuccess ||
                hipEventCreateWithFlags(&L.join_ev[2 * k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&L.join_ev[2 * k + 1], hipEventDisableTiming) != hipSuccess)
                return fail(100, "pipeline: cannot create copy streams");
        L.dl_done.assign(videos.size(), nullptr); L.dl_valid.assign(videos.size(), false);
        for (auto& e : L.dl_done) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(100, "pipeline: cannot create events");
        if (hipEventCreate(&L.ev_up) != hipSuccess || hipEventCreate(&L.ev_up0) != hipSuccess || hipEventCreateWithFlags(&L.ev_done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.ev_done[1], hipEventDisableTiming) != hipSuccess) return fail(100, "pipeline: cannot create events");
        uint32_t maxF = 1;
        for (uint32_t f : s.F) maxF = std::max(maxF, f);
        L.h_sizes_stride = maxF;
        if (hipHostMalloc(reinterpret_cast<void**>(&L.h_sizes), size_t(maxF) * 8 * 2, hipHostMallocPortable) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&L.h_err), 32, hipHostMallocPortable) != hipSuccess) return fail(100, "pipeline: cannot allocate pinned memory");
    }
    for (size_t vi = 0; vi < videos.size(); vi++) s.slot_bytes = std::max(s.slot_bytes, (s.payload[vi] + 4095) & ~size_t(4095));
    // ---- host-side groups by NUMA node (see the top of this file).  numa: 0 = by the devices' nodes, 1 = one group, nothing bound,
    // 2 = test hook: lane i is treated as attached to node i mod (nodes of the host), so that the grouped paths run on a one-GPU box.
    cpu_set_t allowed; CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &allowed);
    const int host_nodes = host_node_count();
    s.groups.clear();
    for (lane_t& L : s.lanes) {
        L.node = opt.numa == 1 || host_nodes < 1 ? -1 : opt.numa == 2 ? L.id % host_nodes : device_numa_node(L.device);
        if (L.node >= host_nodes) L.node = -1;
        int g = -1;
        for (size_t k = 0; k < s.groups.size(); k++) if (s.groups[k].node == L.node) g = int(k);
        if (g < 0) {
            impl::group_t G; G.node = L.node; G.first_lane = L.id;
            if (L.node >= 0) G.cpus = cpus_of_node(L.node, allowed);
            s.groups.push_back(G);
            g = int(s.groups.size()) - 1;
        }
        L.group = g; s.groups[size_t(g)].lanes++;
    }
    s.prepare_seconds = since(t0);
    return 0;
}

// The sharding of a job: runs of consecutive frames of one video (at most F[video] of them) are the batches, dealt to the lanes -- the
// devices -- in turn.  Frames are independent (every frame a key frame, every slice resets its contexts: CLI/Global.cpp:959-960,
// FFV1_Slice.cpp:180-197,274-275), so nothing else is shared: no collective, one placer restores the order.  Also behind
// rcgpu_sequence_plan, which is how the CPU tests see the plan the devices follow.
static void plan_batches(const std::vector<uint32_t>& video_of, const std::vector<uint32_t>& F, const std::vector<uint64_t>& video_frames, int nl, bool ramp,
                         std::vector<batch_t>& batches, std::vector<uint32_t>& batch_of)
{
    const size_t N = video_of.size();
    batches.clear(); batch_of.assign(N, 0);
    std::vector<uint32_t> made(F.size(), 0);
    for (size_t i = 0; i < N;) {
        const uint32_t v = video_of[i];
        size_t cap = F[v];
        if (ramp && video_frames[v] > 2 * uint64_t(F[v]) && F[v] >= 40) {
            const uint32_t k = made[v] / uint32_t(nl);                       // every lane gets a short first and second batch
            if (k == 0) cap = F[v] / 5; else if (k == 1) cap = F[v] / 2;
        }
        made[v]++;
        size_t n = 1;
        while (i + n < N && n < cap && video_of[i + n] == v) n++;
        for (size_t k = 0; k < n; k++) batch_of[i + k] = uint32_t(batches.size());
        batches.push_back({ v, i, n, int(batches.size() % size_t(nl)) });
        i += n;
    }
}

int pipeline::run(const std::vector<pipe_frame>& frames, const pipe_io& io, pipe_stats* stats)
{
    impl& s = *p;
    const auto t0 = clk::now();
    const size_t N = frames.size();
    if (stats) { *stats = pipe_stats(); stats->prepare_seconds = s.prepare_seconds; }
    if (!N) return 0;
    if (s.lanes.empty()) return fail(1, "pipeline: run before prepare");
    for (const pipe_frame& f : frames) if (f.video >= s.videos.size()) return fail(1, "pipeline: frame of an unknown video");
    const int nl = int(s.lanes.size());

    // ---- batches: runs of consecutive frames of one video, dealt round-robin to the lanes
    // RCGPU_RAMP=1 (timing build) makes the first two batches of a long sequence short (a fifth and a half).  Measured in round 3 on 1000 4K files:
    // the first packet leaves after 0.8 s instead of 1.3 s, but five batches instead of three have five starts and drains -- the job
    // takes 8.1 s instead of 7.1 s, host to host 550 instead of 573 frames/s.  For callers that want the first packet early, not the last.
    std::vector<batch_t> batches; std::vector<uint32_t> batch_of(N);
    static const bool ramp = TIMING_ENV("RCGPU_RAMP") != nullptr;
    {
        std::vector<uint32_t> video_of(N);
        for (size_t i = 0; i < N; i++) video_of[i] = frames[i].video;
        std::vector<uint64_t> vframes(s.videos.size());
        for (size_t vi = 0; vi < s.videos.size(); vi++) vframes[vi] = s.videos[vi].frames;
        plan_batches(video_of, s.F, vframes, nl, ramp, batches, batch_of);
    }
    uint32_t maxF = 1; size_t max_pkt = 0, max_payload = 0;
    for (size_t vi = 0; vi < s.videos.size(); vi++) { maxF = std::max(maxF, s.F[vi]); max_pkt = std::max(max_pkt, s.max_packet[vi]); max_payload = std::max(max_payload, s.payload[vi]); }

    unsigned hw = std::thread::hardware_concurrency(); if (!hw) hw = 8;
    const uint32_t readers = s.opt.readers ? s.opt.readers : uint32_t(std::max(2u, std::min(hw / 2, 8u * unsigned(nl))));
    const uint32_t writers = s.opt.writers ? s.opt.writers : uint32_t(std::max(2u, std::min(hw / 2, 8u * unsigned(nl))));
    size_t slots_total = s.opt.in_slots ? s.opt.in_slots : std::min<size_t>(N, std::max<size_t>(2 * readers, std::min<size_t>(64 * size_t(nl), (size_t(8) << 30) / std::max<size_t>(1, s.slot_bytes))));
    slots_total = std::max<size_t>(slots_total, std::min<size_t>(N, 2));
    // every group gets its lanes' share of the threads and of the slots (at least one reader, one writer and two slots), and the list of
    // the frames its lanes code: its readers fill its slots with the lowest of THOSE frames not yet read
    for (impl::group_t& G : s.groups) {
        G.readers = std::max<uint32_t>(1, readers * G.lanes / uint32_t(nl));
        G.writers = std::max<uint32_t>(1, writers * G.lanes / uint32_t(nl));
        G.slots_wanted = std::max<size_t>(std::min<size_t>(N, 2), slots_total * G.lanes / size_t(nl));
        if (io.locate) { G.readers = 0; G.slots_wanted = 0; }          // the payloads are uploaded from where they lie
        G.frames.clear(); G.next_read = 0; G.jobs.clear(); G.reads_done = 0;
    }
    for (const batch_t& b : batches) { impl::group_t& G = s.groups[size_t(s.lanes[size_t(b.lane)].group)]; for (size_t k = 0; k < b.n; k++) G.frames.push_back(b.first + k); }
    for (lane_t& L : s.lanes) {
        // ring: 1..4 GB in chunks that hold at least two worst-case packets.  It only has to cover the writers' reaction time: a ring that
        // held a whole batch of packets (17.8 GB at 4K) gave the same rates (host to host 482 frames/s with 2, 4, 8 or 17.8 GB) and cost
        // 1.4 s of page-locking at the start of a job and as much again when it ended (1000 4K files: 7.4 s with 17.8 GB, 6.5 s with 4)
        uint64_t total_need = 0; for (const batch_t& b : batches) if (b.lane == L.id) total_need += uint64_t(b.n) * s.max_packet[b.video];
        // ... and a small job pins what it needs, not 512 MB per lane: chunks of half the lane's packets, at least two worst-case packets
        // (chunks of an earlier run() on this pipeline are kept, and their size with them)
        if (L.chunks.empty()) L.chunk_bytes = std::max<size_t>(2 * max_pkt, size_t(std::min<uint64_t>(uint64_t(256) << 20, ((total_need / 2 + 4095) & ~uint64_t(4095)))));
        uint64_t want = s.opt.out_ring_bytes ? s.opt.out_ring_bytes : std::min<uint64_t>(uint64_t(4) << 30, std::max<uint64_t>(uint64_t(1) << 30, uint64_t(maxF) * max_payload));
        if (const char* x = TIMING_ENV("RCGPU_OUT_RING_MB")) if (!s.opt.out_ring_bytes && atoll(x) > 0) want = uint64_t(atoll(x)) << 20;        // for sizing experiments
        want = std::min<uint64_t>(want, std::max<uint64_t>(total_need, L.chunk_bytes));
        L.max_chunks = total_need ? std::max<size_t>(2, size_t((want + L.chunk_bytes - 1) / L.chunk_bytes)) : 0;      // a lane without batches pins nothing
        L.cur = -1; L.cur_off = 0; L.outq.clear(); std::fill(L.outstanding.begin(), L.outstanding.end(), 0); L.up_span_valid = false; L.upload_wait = 0; L.h2d_span = 0; L.copy_calls = L.dl_calls = L.dl_wait = 0;
        std::fill(L.dl_valid.begin(), L.dl_valid.end(), false);
    }
    s.ready.assign(N, nullptr); for (lane_t& L : s.lanes) L.pending.clear(); s.placer_finished = false; s.alloc_done = false;
    s.error = 0; s.error_msg.clear();

    std::vector<double> batch_done(batches.size(), 0.0);
    double reads_done = 0;
    std::atomic<uint64_t> packet_bytes{ 0 };
    std::atomic<double> read_busy{ 0.0 }, write_busy{ 0.0 };
    std::atomic<bool> first_seen{ false }; double first_packet_seconds = 0;
    double busy0 = 0;
    const bool trace = s.opt.trace;
    auto mark = [&](const char* what, long a = -1) {
        if (trace) fprintf(stderr, "rcgpu trace: %8.3f s  pipeline: %s%s%s\n", since(t0), what, a >= 0 ? " " : "", a >= 0 ? std::to_string(a).c_str() : "");
    };

    // ---- pinned memory is page-locked at ~13 GB/s: allocated in the background, first users served first
    std::thread allocator([&] {
        // every buffer is allocated from the node it will be used on: by this thread while it runs on that node's CPUs (first touch), with a
        // device of that node current (the runtime picks the pool next to the current device)
        for (impl::group_t& G : s.groups) { std::lock_guard<std::mutex> l(s.m); G.free_slots = G.all_slots; }
        bool more = true;
        while (more && !s.failed()) {
            more = false;
            for (impl::group_t& G : s.groups) {
                if (G.all_slots.size() >= G.slots_wanted) continue;
                bind_this_thread(G.cpus);
                (void)hipSetDevice(s.lanes[size_t(G.first_lane)].device);
                uint8_t* ptr = nullptr;
                if (hipHostMalloc(reinterpret_cast<void**>(&ptr), s.slot_bytes, hipHostMallocPortable) != hipSuccess) {
                    if (G.all_slots.size() < 2) { s.set_error(100, "pipeline: cannot allocate pinned upload slots"); break; }
                    std::lock_guard<std::mutex> l(s.m);
                    G.slots_wanted = G.all_slots.size();
                } else {
                    std::lock_guard<std::mutex> l(s.m);
                    G.all_slots.push_back(ptr); G.free_slots.push_back(ptr);
                    s.cv_slots.notify_all();
                }
                more = true;
            }
            for (lane_t& L : s.lanes) {
                bool need; bool slots_short;
                { std::lock_guard<std::mutex> l(s.m); need = L.chunks.size() < L.max_chunks; const impl::group_t& G = s.groups[size_t(L.group)]; slots_short = G.all_slots.size() < G.slots_wanted; }
                if (!need) continue;
                // the first chunks of every lane come before the bulk of the upload slots
                if (L.chunks.size() >= 2 && slots_short) continue;
                uint8_t* ptr = nullptr;
                bind_this_thread(s.groups[size_t(L.group)].cpus);
                (void)hipSetDevice(L.device);
                if (hipHostMalloc(reinterpret_cast<void**>(&ptr), L.chunk_bytes, hipHostMallocPortable) != hipSuccess) {
                    std::lock_guard<std::mutex> l(s.m);
                    if (L.chunks.size() < 2) { s.error = 100; s.error_msg = "pipeline: cannot allocate the pinned download ring"; }
                    L.max_chunks = L.chunks.size();
                    s.wake_all();
                } else {
                    std::lock_guard<std::mutex> l(s.m);
                    if (L.chunks.empty()) L.pinned_node = node_of_page(ptr);
                    L.chunks.push_back(ptr); L.outstanding.push_back(0);
                    s.cv_ring.notify_all();
                }
                more = true;
            }
        }
        { std::lock_guard<std::mutex> l(s.m); s.alloc_done = true; s.wake_all(); }
    });

    // uploads whose copy has completed give their slot back; called with the lock held.  Copies of one lane complete in order, so only
    // the oldest pending one of each lane is asked about, and not more often than every 100 us whoever asks: the HIP runtime's locks
    // are shared with the lane thread that is launching kernels.
    clk::time_point last_reap = clk::now();
    auto reap = [&]() {
        const auto now = clk::now();
        if (std::chrono::duration<double>(now - last_reap).count() < 100e-6) return;
        last_reap = now;
        for (lane_t& L : s.lanes) {
            // the groups of one stream complete in order: of every stream's groups only the oldest is asked about
            size_t asked = 0;
            for (auto it = L.pending.begin(); it != L.pending.end() && asked < L.cin.size();) {
                if (hipEventQuery(it->ev) != hipSuccess) { ++it; ++asked; continue; }
                for (uint8_t* sl : it->slots) s.groups[size_t(L.group)].free_slots.push_back(sl);
                L.free_events.push_back(it->ev);
                it = L.pending.erase(it);
            }
        }
    };

    // ---- readers
    auto reader = [&](impl::group_t& G) {
        bind_this_thread(G.cpus);
        (void)hipSetDevice(s.lanes[size_t(G.first_lane)].device);
        for (;;) {
            // slot first, frame second, under one lock: the filled slots of a group then always hold the LOWEST of its frames not yet
            // uploaded, which are the ones its lanes wait for (a reader that took its frame number first could be overtaken for the last free
            // slot by readers of later frames, and the pool would fill up with frames nobody can use yet)
            uint8_t* slot = nullptr; size_t i = 0;
            {
                std::unique_lock<std::mutex> l(s.m);
                for (;;) {
                    if (s.error) return;
                    if (G.next_read >= G.frames.size()) { if (!G.reads_done) G.reads_done = since(t0); return; }
                    reap();
                    if (!G.free_slots.empty()) { slot = G.free_slots.back(); G.free_slots.pop_back(); i = G.frames[G.next_read++]; break; }
                    s.cv_slots.wait_for(l, std::chrono::microseconds(200));
                }
            }
            const auto tr = clk::now();
            if (int r = io.read(frames[i], slot)) { s.set_error(r, rcgpu_last_error()); return; }
            { const double d = since(tr); double cur = read_busy.load(); while (!read_busy.compare_exchange_weak(cur, cur + d)) {} }
            { std::lock_guard<std::mutex> l(s.m); s.ready[i] = slot; }
            s.cv_ready.notify_all();
        }
    };

    // ---- lanes
    auto lane_main = [&](lane_t& L) {
        bind_this_thread(s.groups[size_t(L.group)].cpus);
        if (hipSetDevice(L.device) != hipSuccess) { s.set_error(100, "pipeline: hipSetDevice failed"); return; }
        std::vector<size_t> mine;
        for (size_t b = 0; b < batches.size(); b++) if (batches[b].lane == L.id) mine.push_back(b);
        auto hip_ok = [&](hipError_t e, const char* what) { if (e == hipSuccess) return true; char t[256]; snprintf(t, sizeof t, "pipeline: %s: %s", what, hipGetErrorString(e)); s.set_error(100, t); return false; };
        auto get_event = [&]() -> hipEvent_t {
            { std::lock_guard<std::mutex> l(s.m); if (!L.free_events.empty()) { hipEvent_t e = L.free_events.back(); L.free_events.pop_back(); return e; } }
            hipEvent_t e = nullptr;
            // blocking sync: a writer waiting for a download sleeps in the kernel instead of spinning on the runtime
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return nullptr;
            return e;
        };
        // ---- transfers.  Uploads of one batch and downloads of another are issued by ONE loop, a copy of each per turn: a whole batch of
        // downloads queued at once kept every upload issued after it waiting until the last download had finished (measured: 0.33 s
        // per batch in which the device-to-host direction ran at 48 GB/s and the host-to-device direction stood still), although the two
        // directions overlap perfectly when their copies are issued side by side (tools/probe_dma.py).
        std::vector<uint8_t*> ugroup;
        // A group must never hold so much of the pool (or of the ring) that what it waits for cannot happen before its event is recorded:
        // at most a quarter of this lane's share, from the sizes as they are NOW (the allocator shrinks slots_wanted and max_chunks when
        // pinned memory runs short; both are read under the lock), and a group that is still open is closed before its lane blocks.
        auto flush_ugroup = [&]() -> bool {
            if (ugroup.empty()) return true;
            hipStream_t cs = L.cin[L.up_turn % L.cin.size()];
            hipEvent_t ev = get_event();
            if (!ev) { s.set_error(100, "pipeline: cannot create an event"); return false; }
            if (!hip_ok(hipEventRecord(ev, cs), "hipEventRecord")) { std::lock_guard<std::mutex> l(s.m); L.free_events.push_back(ev); return false; }
            { std::lock_guard<std::mutex> l(s.m); L.pending.push_back({ ugroup, ev }); }
            ugroup.clear(); L.up_turn++;
            return true;
        };
        auto upload_one = [&](const batch_t& B, const enc_staging& sg, size_t k) -> bool {
            if (io.locate) {
                // from the caller's pinned memory, groups of copies dealt to the copy streams in turn; the batch's last copy is followed by
                // ev_up (transfers below), nothing is to be given back
                const uint8_t* src = io.locate(frames[B.first + k]);
                if (!src) { s.set_error(21, *rcgpu_last_error() ? rcgpu_last_error() : "pipeline: a frame could not be located"); return false; }
                const auto tc = clk::now();
                hipStream_t cs = L.cin[(k / kUploadGroup) % L.cin.size()];
                if (!hip_ok(hipMemcpyAsync(sg.d_in + k * sg.in_stride, src, sg.payload_bytes, hipMemcpyHostToDevice, cs), "upload")) return false;
                L.copy_calls += since(tc);
                return true;
            }
            uint8_t* slot = nullptr;
            size_t ugroup_max = 1;
            {
                const auto tw = clk::now();
                std::unique_lock<std::mutex> l(s.m);
                if (!s.error && !s.ready[B.first + k] && !ugroup.empty()) {      // about to wait for a reader: what is open becomes reclaimable first
                    l.unlock();
                    if (!flush_ugroup()) return false;
                    l.lock();
                }
                s.cv_ready.wait(l, [&] { return s.error || s.ready[B.first + k]; });
                if (s.error) return false;
                slot = s.ready[B.first + k];
                { const impl::group_t& G = s.groups[size_t(L.group)]; ugroup_max = std::max<size_t>(1, std::min<size_t>(kUploadGroup, G.slots_wanted / (4 * size_t(std::max(1u, G.lanes))))); }
                L.upload_wait += since(tw);
            }
            const auto tc = clk::now();
            hipStream_t cs = L.cin[L.up_turn % L.cin.size()];          // a group stays on one stream: its event covers all of it
            if (!hip_ok(hipMemcpyAsync(sg.d_in + k * sg.in_stride, slot, sg.payload_bytes, hipMemcpyHostToDevice, cs), "upload")) return false;
            ugroup.push_back(slot);                          // one completion event per group of copies: an event is a barrier packet
            if ((ugroup.size() >= ugroup_max || k + 1 == B.n) && !flush_ugroup()) return false;
            L.copy_calls += since(tc);
            return true;
        };
        std::vector<out_entry> dgroup;
        size_t dgroup_max = 1;
        auto flush_dgroup = [&]() -> bool {                 // the group becomes visible to the placer once its event is recorded
            if (dgroup.empty()) return true;
            hipStream_t cs = L.cout[L.dn_turn % L.cout.size()];
            hipEvent_t ev = get_event();
            if (!ev) { s.set_error(100, "pipeline: cannot create an event"); return false; }
            if (!hip_ok(hipEventRecord(ev, cs), "hipEventRecord")) { std::lock_guard<std::mutex> l(s.m); L.free_events.push_back(ev); return false; }
            L.dn_turn++;
            int* users = new int(int(dgroup.size()));
            { std::lock_guard<std::mutex> l(s.m); for (out_entry& g : dgroup) { g.ev = ev; g.ev_users = users; L.outq.push_back(g); } }
            dgroup.clear();
            s.cv_out.notify_one();
            return true;
        };
        auto ring_alloc = [&](size_t size, int& chunk) -> uint8_t* {
            const size_t need = (size + 4095) & ~size_t(4095);
            std::unique_lock<std::mutex> l(s.m);
            dgroup_max = std::max<size_t>(1, std::min<size_t>(kDownloadGroup, L.max_chunks * (L.chunk_bytes / std::max<size_t>(1, (max_pkt + 4095) & ~size_t(4095))) / 4));
            for (;;) {
                if (s.error) return nullptr;
                if (L.cur >= 0) {
                    if (L.outstanding[size_t(L.cur)] == 0) L.cur_off = 0;          // everything in it was consumed: start over
                    if (L.cur_off + need <= L.chunk_bytes) break;
                }
                const int nc = int(L.chunks.size());
                int pick = -1;
                for (int k = 1; k <= nc && pick < 0; k++) { const int c = (L.cur + k) % nc; if (c != L.cur && L.outstanding[size_t(c)] == 0) pick = c; }
                if (pick >= 0) { L.cur = pick; L.cur_off = 0; continue; }
                if (!dgroup.empty()) {                      // the writers cannot see an open group, and this lane is about to wait for them
                    l.unlock();
                    if (!flush_dgroup()) return nullptr;
                    l.lock();
                    continue;
                }
                s.cv_ring.wait_for(l, std::chrono::milliseconds(1));
            }
            chunk = L.cur;
            uint8_t* ptr = L.chunks[size_t(L.cur)] + L.cur_off;
            L.cur_off += need; L.outstanding[size_t(L.cur)]++;
            return ptr;
        };
        // One batch = upload | k_model ... k_scan | k_gather | download.  The loop keeps the DEVICE busy: batch k+1 is started (its call
        // returns once k_model(k+1) has run, i.e. after everything of batch k) before the host turns to batch k's sizes and downloads.
        // (the packet sizes of consecutive batches go to the two halves of d_psizes: in run-on mode batch k+1 is issued before batch k's sizes are read)
        auto start_batch = [&](const batch_t& B, int par) -> bool {        // uploads of B are on their way
            rcgpu_ffv1* enc = L.enc[B.video];
            enc_staging sg; if (ffv1_staging(enc, &sg)) { s.set_error(100, rcgpu_last_error()); return false; }
            hipStream_t st = static_cast<hipStream_t>(sg.compute_stream);
            if (!hip_ok(hipStreamWaitEvent(st, L.ev_up, 0), "hipStreamWaitEvent")) return false;
            ffv1_set_input_event(enc, L.ev_up);                 // (run-on encoders model the batch behind its uploads, not behind the stream's gather)
            ffv1_set_defer_gather(enc, true);
            std::vector<const void*> ptrs(B.n);
            for (size_t i = 0; i < B.n; i++) ptrs[i] = sg.d_in + i * sg.in_stride;
            if (int r = rcgpu_ffv1_encode_device(enc, ptrs.data(), uint32_t(B.n), sg.d_packets, sg.packet_stride, sg.d_psizes + size_t(par) * sg.max_batch, st)) { s.set_error(r, rcgpu_last_error()); return false; }
            return true;
        };
        auto finish_batch = [&](const batch_t& B, int par) -> bool {   // k_gather behind the previous download, then sizes and flags to the host
            rcgpu_ffv1* enc = L.enc[B.video];
            enc_staging sg; if (ffv1_staging(enc, &sg)) { s.set_error(100, rcgpu_last_error()); return false; }
            hipStream_t st = static_cast<hipStream_t>(sg.compute_stream);
            ffv1_set_gather_wait(enc, L.dl_valid[B.video] ? L.dl_done[B.video] : nullptr);
            if (int r = ffv1_gather(enc, sg.d_packets, sg.packet_stride, st)) { s.set_error(r, rcgpu_last_error()); return false; }
            // by a kernel, not by a copy engine: these few bytes must not wait behind the packets of the previous batch (ffv1_internal.h)
            return hip_ok(hipError_t(copy_by_kernel_on(L.h_sizes + size_t(par) * L.h_sizes_stride, sg.d_psizes + size_t(par) * sg.max_batch, 8 * B.n, st)), "sizes") &&
                   hip_ok(hipError_t(copy_by_kernel_on(L.h_err + 4 * par, sg.d_err, 16, st)), "flags") &&
                   hip_ok(hipEventRecord(L.ev_done[par], st), "hipEventRecord");
        };
        auto download_one = [&](const batch_t& B, const enc_staging& sg, int par, size_t i) -> bool {
            const uint64_t* sizes = L.h_sizes + size_t(par) * L.h_sizes_stride;
            out_entry o; o.frame = B.first + i; o.size = size_t(sizes[i]); o.device = L.device; o.lane = L.id;
            if (!o.size || o.size > sg.packet_stride) { s.set_error(102, "ffv1: the device returned an impossible packet size"); return false; }
            const auto tw = clk::now();
            o.src = ring_alloc(o.size, o.chunk);
            if (!o.src) return false;
            const auto tc = clk::now();
            L.dl_wait += std::chrono::duration<double>(tc - tw).count();
            // whole 4 KB pages: a copy of an odd number of bytes does not take the copy engines' fast path (the slot in the ring and the
            // packet's stride on the device both have the room)
            const size_t whole = std::min<size_t>((o.size + 4095) & ~size_t(4095), sg.packet_stride);
            hipStream_t cs = L.cout[L.dn_turn % L.cout.size()];
            if (!hip_ok(hipMemcpyAsync(o.src, sg.d_packets + i * sg.packet_stride, whole, hipMemcpyDeviceToHost, cs), "download")) return false;
            L.dl_calls += since(tc);
            dgroup.push_back(o);
            if ((dgroup.size() >= dgroup_max || i + 1 == B.n) && !flush_dgroup()) return false;
            return true;
        };
        // everything issued so far on the streams of one direction is behind what stream 0 of it does next
        auto join = [&](std::vector<hipStream_t>& cs, int dir) -> bool {
            for (size_t k = 1; k < cs.size(); k++)
                if (!hip_ok(hipEventRecord(L.join_ev[2 * k + size_t(dir)], cs[k]), "hipEventRecord") ||
                    !hip_ok(hipStreamWaitEvent(cs[0], L.join_ev[2 * k + size_t(dir)], 0), "hipStreamWaitEvent")) return false;
            return true;
        };
        // downloads of batch D (complete on the device, sizes and flags on the host) and uploads of batch U, either may be absent
        auto transfers = [&](const batch_t* D, int par, const batch_t* U) -> bool {
            enc_staging sd{}, su{};
            if (D) {
                if (ffv1_staging(L.enc[D->video], &sd)) { s.set_error(100, rcgpu_last_error()); return false; }
                const uint32_t* err = L.h_err + 4 * par;
                if (err[0]) { char t[256]; snprintf(t, sizeof t, "ffv1: %s (flags %u)", ffv1_error_flags_text(err[0]), err[0]); s.set_error(102, t); return false; }
                for (hipStream_t c : L.cout) if (!hip_ok(hipStreamWaitEvent(c, L.ev_done[par], 0), "hipStreamWaitEvent")) return false;
            }
            if (U && ffv1_staging(L.enc[U->video], &su)) { s.set_error(100, rcgpu_last_error()); return false; }
            if (U) {
                // device clock around this batch's uploads; the pair recorded for the previous batch has completed (its k_model has run)
                float ms = 0;
                if (L.up_span_valid && hipEventElapsedTime(&ms, L.ev_up0, L.ev_up) == hipSuccess) L.h2d_span += double(ms) * 1e-3;
                if (!hip_ok(hipEventRecord(L.ev_up0, L.cin[0]), "hipEventRecord")) return false;
                L.up_span_valid = true;
            }
            const auto tu = clk::now();
            const size_t nd = D ? D->n : 0, nu = U ? U->n : 0;
            for (size_t i = 0; i < std::max(nd, nu); i++) {
                if (i < nd && !download_one(*D, sd, par, i)) return false;
                if (i < nu && !upload_one(*U, su, i)) return false;
            }
            if (D) {
                if (!join(L.cout, 0) || !hip_ok(hipEventRecord(L.dl_done[D->video], L.cout[0]), "hipEventRecord")) return false;
                L.dl_valid[D->video] = true;
            }
            if (U && (!join(L.cin, 1) || !hip_ok(hipEventRecord(L.ev_up, L.cin[0]), "hipEventRecord"))) return false;
            if (trace && L.id == 0) { char b[160]; snprintf(b, sizeof b, "transfers issued in %.3f s: %zu downloads, %zu uploads", since(tu), nd, nu); mark(b); }
            return true;
        };
        if (mine.empty()) return;
        const bool serial = bool(io.after_batch);
        if (!transfers(nullptr, 0, &batches[mine[0]])) return;
        const double first_call = since(t0);
        if (!start_batch(batches[mine[0]], 0)) return;
        if (!serial && mine.size() > 1 && !transfers(nullptr, 0, &batches[mine[1]])) return;      // d_in is free: k_model(0) has run
        for (size_t k = 0; k < mine.size(); k++) {
            const batch_t& B = batches[mine[k]];
            const int par = int(k & 1);
            const bool more = k + 1 < mine.size();
            // here: batch k is modelled and running, the uploads of batch k+1 are issued (not in serial mode)
            if (serial) {      // the hook wants the payloads on the device: nothing may overwrite them before it has run
                if (!finish_batch(B, par) || !hip_ok(hipEventSynchronize(L.ev_done[par]), "batch")) return;
                if (int r = io.after_batch(B.video, L.enc[B.video], frames[B.first].index, uint32_t(B.n))) { s.set_error(r, rcgpu_last_error()); return; }
                if (more && !transfers(nullptr, 0, &batches[mine[k + 1]])) return;
            } else if (!finish_batch(B, par)) return;
            // returns after k_model(k+1): which follows batch k on the stream -- or, with run-on encoders (rcgpu_ffv1_set_run_on), runs beside it
            if (more && !start_batch(batches[mine[k + 1]], par ^ 1)) return;
            if (!hip_ok(hipEventSynchronize(L.ev_done[par]), "batch")) return;
            batch_done[mine[k]] = since(t0);
            if (L.id == 0) busy0 = batch_done[mine[k]] - first_call;       // the device of lane 0 has had a batch in flight since its first encode call
            if (trace && L.id == 0) {      // this batch's kernel times (HIP events): the next batch is already running on the other event set
                mark("batch complete:", long(mine[k]));
                const char* names[12]; float ms[12];
                const int nk = more ? ffv1_prev_kernel_times(L.enc[B.video], names, ms, 12) : rcgpu_ffv1_last_kernel_times(L.enc[B.video], names, ms, 12);
                std::string t = "kernel ms of this batch:";
                for (int i = 0; i < nk; i++) if (ms[i] > 0) { char b[64]; snprintf(b, sizeof b, " %s %.1f", names[i], ms[i]); t += b; }
                mark(t.c_str());
                if (more) { float tl[6]; if (ffv1_prev_timeline(L.enc[B.video], tl)) { char b2[256]; snprintf(b2, sizeof b2, "device timeline from k_model: first k_resolve +%.1f, last k_resolve ends +%.1f (gaps between them %.1f), last k_rangecode ends +%.1f, gather ends +%.1f, next k_model starts +%.1f ms", tl[0], tl[1], tl[2], tl[3], tl[4], tl[5]); mark(b2); } }
                char b[200]; snprintf(b, sizeof b, "lane 0 so far: waiting for readers %.3f s, inside upload calls %.3f s, waiting for ring space %.3f s, inside download calls %.3f s",
                                      L.upload_wait, L.copy_calls, L.dl_wait, L.dl_calls); mark(b);
            }
            // downloads of this batch side by side with the uploads of the batch after the next (d_in is free: k_model(k+1) has run)
            if (!transfers(&B, par, !serial && k + 2 < mine.size() ? &batches[mine[k + 2]] : nullptr)) return;
        }
    };

    // ---- placer: output order
    auto placer = [&] {
        for (size_t i = 0; i < N; i++) {
            lane_t& L = s.lanes[size_t(batches[batch_of[i]].lane)];
            out_entry o;
            {
                std::unique_lock<std::mutex> l(s.m);
                s.cv_out.wait(l, [&] { return s.error || (!L.outq.empty() && L.outq.front().frame == i); });
                if (s.error) break;
                o = L.outq.front(); L.outq.pop_front();
            }
            o.dst = io.place ? io.place(frames[i], o.size) : nullptr;
            if (io.place && !o.dst && *rcgpu_last_error()) { s.set_error(20, rcgpu_last_error()); break; }
            { std::lock_guard<std::mutex> l(s.m); s.groups[size_t(L.group)].jobs.push_back(o); }
            s.cv_jobs.notify_all();            // the writers of every group sleep on this condition: the group's own must hear it
        }
        { std::lock_guard<std::mutex> l(s.m); s.placer_finished = true; }
        s.cv_jobs.notify_all();
    };

    // ---- writers
    auto writer = [&](impl::group_t& G) {
        bind_this_thread(G.cpus);
        int cur_dev = -1;
        for (;;) {
            out_entry o;
            {
                std::unique_lock<std::mutex> l(s.m);
                s.cv_jobs.wait(l, [&] { return s.error || !G.jobs.empty() || s.placer_finished; });
                if (G.jobs.empty()) return;             // error or finished
                o = G.jobs.front(); G.jobs.pop_front();
            }
            if (cur_dev != o.device) { (void)hipSetDevice(o.device); cur_dev = o.device; }
            int r = 0;
            if (hipEventSynchronize(o.ev) != hipSuccess) r = fail(100, "pipeline: a download failed");
            if (!r) {
                if (!first_seen.exchange(true)) first_packet_seconds = since(t0);
                const auto tw = clk::now();
                if (o.dst) { if (io.copy) r = io.copy(o.dst, o.src, o.size); else memcpy(o.dst, o.src, o.size); }
                if (!r && io.done) r = io.done(frames[o.frame], o.dst ? o.dst : o.src, o.size);
                { const double d = since(tw); double cur = write_busy.load(); while (!write_busy.compare_exchange_weak(cur, cur + d)) {} }
                packet_bytes += o.size;
            }
            {
                std::lock_guard<std::mutex> l(s.m);
                lane_t& L = s.lanes[size_t(o.lane)];
                L.outstanding[size_t(o.chunk)]--;
                if (--*o.ev_users == 0) { L.free_events.push_back(o.ev); delete o.ev_users; }
                if (r && !s.error) { s.error = r; s.error_msg = rcgpu_last_error(); }
            }
            if (r) s.wake_all(); else s.cv_ring.notify_all();
        }
    };

    std::vector<std::thread> threads;
    for (impl::group_t& G : s.groups) for (uint32_t i = 0; i < G.readers; i++) threads.emplace_back(reader, std::ref(G));
    for (lane_t& L : s.lanes) threads.emplace_back(lane_main, std::ref(L));
    std::thread placer_thread(placer);
    std::vector<std::thread> wthreads;
    for (impl::group_t& G : s.groups) for (uint32_t i = 0; i < G.writers; i++) wthreads.emplace_back(writer, std::ref(G));
    for (auto& t : threads) t.join();
    // a lane that stopped early (error) leaves the placer waiting: the error flag wakes it
    placer_thread.join();
    for (auto& t : wthreads) t.join();
    allocator.join();
    for (lane_t& L : s.lanes) {
        (void)hipSetDevice(L.device); (void)hipDeviceSynchronize();
        float ms = 0;
        if (L.up_span_valid && hipEventElapsedTime(&ms, L.ev_up0, L.ev_up) == hipSuccess) L.h2d_span += double(ms) * 1e-3;      // the last batch's uploads
        L.up_span_valid = false;
    }
    {   // slots still listed as pending are free now; after an error, so is whatever the placer and the writers left behind: their
        // events go back, and the ring is empty again for the next run() on this pipeline
        std::lock_guard<std::mutex> l(s.m);
        auto drop = [&](std::deque<out_entry>& q) {
            for (out_entry& o : q) if (o.ev_users && --*o.ev_users == 0) { s.lanes[size_t(o.lane)].free_events.push_back(o.ev); delete o.ev_users; }
            q.clear();
        };
        for (lane_t& L : s.lanes) {
            for (auto& u : L.pending) L.free_events.push_back(u.ev);
            L.pending.clear();
            drop(L.outq);
        }
        for (impl::group_t& G : s.groups) { drop(G.jobs); G.free_slots.clear(); }
        for (lane_t& L : s.lanes) { std::fill(L.outstanding.begin(), L.outstanding.end(), 0); L.cur = -1; L.cur_off = 0; }
    }
    if (stats) {
        stats->seconds = since(t0); stats->first_packet_seconds = first_packet_seconds; stats->frames = N;
        for (const pipe_frame& f : frames) stats->payload_bytes += s.payload[f.video];
        stats->packet_bytes = packet_bytes; stats->batches = batches.size(); stats->batch_frames = maxF; stats->lanes = uint32_t(nl);
        stats->readers = 0; stats->writers = 0; stats->device_busy_seconds = busy0;
        for (const impl::group_t& G : s.groups) { stats->readers += G.readers; stats->writers += G.writers; reads_done = std::max(reads_done, G.reads_done); }
        stats->groups = uint32_t(s.groups.size());
        for (size_t li = 0; li < s.lanes.size() && li < 16; li++) { stats->lane_device[li] = s.lanes[li].device; stats->lane_node[li] = s.lanes[li].node; stats->lane_pinned_node[li] = s.lanes[li].pinned_node; }
        stats->read_call_seconds = read_busy.load() / double(N); stats->write_call_seconds = write_busy.load() / double(N);
        stats->reads_done_seconds = reads_done; stats->upload_wait_seconds = s.lanes[0].upload_wait; stats->h2d_span_seconds = s.lanes[0].h2d_span;
        if (!batches.empty()) {
            double first = 1e300, last = 0; size_t first_b = 0;
            for (size_t b = 0; b < batches.size(); b++) { if (batch_done[b] < first) { first = batch_done[b]; first_b = b; } last = std::max(last, batch_done[b]); }
            stats->last_batch_seconds = last;
            if (batches.size() > 1 && last > first) stats->steady_frames_per_second = double(N - batches[first_b].n) / (last - first);
        }
    }
    if (s.error) return fail(s.error, "%s", s.error_msg.c_str());
    return 0;
}

}  // namespace rc

// ---------------------------------------------------------------------------------------------------------
// C ABI: one picture sequence through the pipeline, with the caller's callbacks on both ends
// ---------------------------------------------------------------------------------------------------------
extern "C" int rcgpu_ffv1_encode_sequence(const rcgpu_ffv1_config* cfg, uint64_t n_frames, const rcgpu_sequence_io* io,
                                          const rcgpu_sequence_options* opt, rcgpu_sequence_stats* stats, uint8_t* record, size_t* record_size)
{
    using namespace rc;
    clear_error();
    if (!cfg || !io) return fail(1, "sequence: null argument");
    if (io->struct_size != sizeof(rcgpu_sequence_io) || (opt && opt->struct_size != sizeof(rcgpu_sequence_options)))
        return fail(1, "sequence: struct_size %u / %u, this library's structs have %zu / %zu bytes -- the caller was built against another rcgpu.h",
                    io->struct_size, opt ? opt->struct_size : 0u, sizeof(rcgpu_sequence_io), sizeof(rcgpu_sequence_options));
    if ((!io->read_frame && !io->locate_frame) || !io->packet_done) return fail(1, "sequence: null argument");
    if (io->read_frame && io->locate_frame) return fail(1, "sequence: read_frame and locate_frame are alternatives -- set one");
    pipe_video v; v.cfg = *cfg; v.frames = n_frames;
    pipe_options po;
    if (opt) { po.device_first = opt->device_first; po.device_count = opt->device_count; po.readers = opt->readers; po.writers = opt->writers;
               po.in_slots = opt->in_ring_frames; po.out_ring_bytes = opt->out_ring_bytes; po.batch = opt->batch; po.lanes_per_device = opt->lanes_per_device;
               po.device_aliases = opt->device_aliases; po.copy_streams = opt->copy_streams; po.numa = opt->numa; po.run_on = opt->run_on; }
    if (const char* e = getenv("RCGPU_LANES")) if (!po.lanes_per_device) po.lanes_per_device = uint32_t(std::max(1, atoi(e)));
    if (!po.batch) po.batch = cfg->max_batch > 1 ? cfg->max_batch : 0;
    po.trace = getenv("RCGPU_TRACE") != nullptr;
    pipeline pl;
    if (int r = pl.prepare({ v }, po)) return r;
    if (record_size) {
        const size_t cap = *record_size;
        *record_size = rcgpu_ffv1_config_record(pl.encoder(0), record, record ? cap : 0);
    }
    std::vector<pipe_frame> frames(n_frames);
    for (uint64_t i = 0; i < n_frames; i++) frames[i] = { 0, i };
    const size_t payload = size_t(payload_bytes(cfg->pixfmt, cfg->width, cfg->height, cfg->line_bytes, cfg->flags));
    pipe_io pio;
    if (io->read_frame) pio.read = [&](const pipe_frame& f, uint8_t* dst) { return io->read_frame(io->user, f.index, dst, payload); };
    if (io->locate_frame) pio.locate = [&](const pipe_frame& f) { return io->locate_frame(io->user, f.index, payload); };
    if (io->place_packet) pio.place = [&](const pipe_frame& f, size_t size) { return io->place_packet(io->user, f.index, size); };
    pio.done = [&](const pipe_frame& f, const uint8_t* data, size_t size) { return io->packet_done(io->user, f.index, data, size); };
    pipe_stats ps;
    const int r = pl.run(frames, pio, &ps);
    if (stats) {
        stats->seconds = ps.seconds; stats->first_packet_seconds = ps.first_packet_seconds; stats->prepare_seconds = ps.prepare_seconds;
        stats->frames = ps.frames; stats->payload_bytes = ps.payload_bytes; stats->packet_bytes = ps.packet_bytes; stats->batches = ps.batches;
        stats->batch_frames = ps.batch_frames; stats->devices = ps.lanes; stats->readers = ps.readers; stats->writers = ps.writers;
        stats->device_busy_seconds = ps.device_busy_seconds; stats->steady_frames_per_second = ps.steady_frames_per_second;
        stats->reads_done_seconds = ps.reads_done_seconds; stats->last_batch_seconds = ps.last_batch_seconds;
        stats->upload_wait_seconds = ps.upload_wait_seconds; stats->h2d_span_seconds = ps.h2d_span_seconds;
        stats->read_call_seconds = ps.read_call_seconds; stats->write_call_seconds = ps.write_call_seconds;
        stats->host_groups = ps.groups;
        for (int i = 0; i < 16; i++) { stats->lane_device[i] = ps.lane_device[i]; stats->lane_numa_node[i] = ps.lane_node[i]; stats->lane_pinned_node[i] = ps.lane_pinned_node[i]; }
    }
    return r;
}

// Which lane -- which device of device_first .. device_first + device_count - 1 -- codes which frame of a sequence of n_frames coded in batches
// of `batch` frames: lane_of_frame and batch_of_frame receive n_frames entries each (either may be NULL).  No device needed.
extern "C" int rcgpu_sequence_plan(uint64_t n_frames, uint32_t batch, uint32_t lanes, uint32_t* lane_of_frame, uint32_t* batch_of_frame)
{
    rc::clear_error();
    if (!batch || !lanes) return rc::fail(1, "sequence plan: batch and lanes must be at least 1");
    std::vector<uint32_t> video_of(size_t(n_frames), 0), batch_of;
    std::vector<rc::batch_t> batches;
    rc::plan_batches(video_of, { batch }, { n_frames }, rc::effective_lanes(n_frames, lanes, 1, nullptr), false, batches, batch_of);
    for (uint64_t i = 0; i < n_frames; i++) {
        if (lane_of_frame) lane_of_frame[i] = uint32_t(batches[batch_of[size_t(i)]].lane);
        if (batch_of_frame) batch_of_frame[i] = batch_of[size_t(i)];
    }
    return 0;
}

extern "C" uint32_t rcgpu_sequence_plan_lanes(uint64_t n_frames, uint32_t devices, uint32_t lanes_per_device)
{
    return devices ? uint32_t(rc::effective_lanes(n_frames, devices, lanes_per_device, nullptr)) : 0u;
}

// Host memory in, host memory out: frame i is frames[i % n_in], packet i lands in out[i % n_out] (out_cap bytes each) and sizes[i].
// With n_in == n_out == n_frames this is rcgpu_ffv1_encode_host for a whole sequence, pipelined; smaller rings re-use buffers (the
// bench's synthetic sequence; a caller that consumes packets as they arrive would rather pass its own callbacks).
namespace {
struct memory_io { const uint8_t* const* frames; uint64_t n_in; uint8_t* const* out; uint64_t n_out; size_t out_cap; uint64_t* sizes; };
int memory_read(void* user, uint64_t frame, uint8_t* dst, size_t bytes)
{
    const memory_io* m = static_cast<const memory_io*>(user);
    memcpy(dst, m->frames[frame % m->n_in], bytes);
    return 0;
}
const uint8_t* memory_locate(void* user, uint64_t frame, size_t)
{
    const memory_io* m = static_cast<const memory_io*>(user);
    return m->frames[frame % m->n_in];
}
int memory_done(void* user, uint64_t frame, const uint8_t* data, size_t size)
{
    const memory_io* m = static_cast<const memory_io*>(user);
    if (m->sizes) m->sizes[frame] = size;
    if (!m->out || !m->n_out) return 0;
    if (size > m->out_cap) return rc::fail(90, "sequence: packet %llu of %zu bytes does not fit the caller's %zu-byte buffers", (unsigned long long)frame, size, m->out_cap);
    memcpy(m->out[frame % m->n_out], data, size);
    return 0;
}
}  // namespace

extern "C" int rcgpu_ffv1_encode_sequence_memory(const rcgpu_ffv1_config* cfg, const uint8_t* const* frames, uint64_t n_in, uint64_t n_frames,
                                                 uint8_t* const* out, uint64_t n_out, size_t out_cap, uint64_t* sizes,
                                                 const rcgpu_sequence_options* opt, rcgpu_sequence_stats* stats, uint8_t* record, size_t* record_size)
{
    rc::clear_error();
    if (!cfg || !frames || !n_in) return rc::fail(1, "sequence: null argument");
    memory_io m{ frames, n_in, out, n_out, out_cap, sizes };
    if (opt && opt->struct_size != sizeof(rcgpu_sequence_options)) return rc::fail(1, "sequence: options.struct_size %u, this library's struct has %zu bytes", opt->struct_size, sizeof(rcgpu_sequence_options));
    rcgpu_sequence_io io{ uint32_t(sizeof(rcgpu_sequence_io)), memory_read, nullptr, memory_done, &m, nullptr };
    if (opt && opt->frames_pinned) { io.read_frame = nullptr; io.locate_frame = memory_locate; }
    return rcgpu_ffv1_encode_sequence(cfg, n_frames, &io, opt, stats, record, record_size);
}
