// mkv_mux.cpp -- minimal Matroska writer for the FFV1 (+FLAC) + attachments files RAWcooked produces.
//
// Replaces FFmpeg's matroska muxer on the encode path (the `-f matroska "<out>"` tail of the command assembled at
// CLI/Output.cpp:303-305).  Layout rules come from what the reference's own reader accepts
// (Lib/Compressed/Matroska/Matroska.cpp): one Segment with a KNOWN size (:1259-1277), Tracks in stream order with
// CodecID/CodecPrivate/PixelWidth/PixelHeight as 1-2 byte uints (:992-1030), Attachments BEFORE the first Cluster
// (:863-873), SimpleBlocks only with single-byte track numbers (:934-953).  SeekHead, Info, Cues and the
// per-track UIDs/durations are there for stock players (they are skipped by the reference, :128-217).
#include "rc_common.h"
#include <cerrno>
#include <cmath>
#include <fcntl.h>
#include <atomic>
#include <chrono>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <deque>
#include <mutex>
#include <thread>
#include <unistd.h>

using namespace rc;

namespace {

struct ebuf {                       // growable byte buffer with EBML primitives
    std::vector<uint8_t> b;
    void raw(const void* p, size_t n) { const uint8_t* s = static_cast<const uint8_t*>(p); b.insert(b.end(), s, s + n); }
    void id(uint32_t v) { for (int s = 24; s >= 0; s -= 8) if (v >> s) { for (; s >= 0; s -= 8) b.push_back(uint8_t(v >> s)); return; } }
    void size(uint64_t v)           // shortest EBML size
    {
        int n = 1;
        while (n < 8 && v >= (uint64_t(1) << (7 * n)) - 1) n++;
        size_n(v, n);
    }
    void size_n(uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) b.push_back(uint8_t((v >> (8 * i)) & 0xFF) | (i == n - 1 ? uint8_t(1 << (8 - n)) : 0)); }
    void uint(uint32_t i, uint64_t v) { id(i); int n = 1; while (n < 8 && (v >> (8 * n))) n++; size(uint64_t(n)); for (int k = n - 1; k >= 0; k--) b.push_back(uint8_t(v >> (8 * k))); }
    void uint_n(uint32_t i, uint64_t v, int n) { id(i); size(uint64_t(n)); for (int k = n - 1; k >= 0; k--) b.push_back(uint8_t(v >> (8 * k))); }
    void f64(uint32_t i, double v) { id(i); size(8); uint64_t u; memcpy(&u, &v, 8); for (int k = 7; k >= 0; k--) b.push_back(uint8_t(u >> (8 * k))); }
    void str(uint32_t i, const std::string& s) { id(i); size(s.size()); raw(s.data(), s.size()); }
    void bin(uint32_t i, const void* p, size_t n) { id(i); size(n); raw(p, n); }
    void master(uint32_t i, const ebuf& c) { id(i); size(c.b.size()); raw(c.b.data(), c.b.size()); }
    void voidel(size_t total)       // EBML Void occupying exactly `total` bytes (total >= 2)
    {
        b.push_back(0xEC);
        if (total < 10) { size_n(total - 2, 1); b.insert(b.end(), total - 2, 0); }
        else { size_n(total - 9, 8); b.insert(b.end(), total - 9, 0); }
    }
};

struct track {
    bool video;
    std::vector<uint8_t> codec_private;
    uint32_t width = 0, height = 0, fps_num = 0, fps_den = 0;
    uint32_t channels = 0, sample_rate = 0, bits = 0;
    const char* codec_id = nullptr; // audio only; nullptr = A_FLAC
    uint64_t cp_file_pos = 0;      // absolute file offset of the CodecPrivate payload
    uint64_t uid = 0;              // TrackUID, referenced by Tags
    uint64_t last_pts_ms = 0, end_ms = 0;
};
struct attach { std::string name, mime; std::vector<uint8_t> data; };
struct tag { int track; std::string name, value; };
struct cue { uint64_t time_ms; int track; uint64_t cluster_pos; };

}  // namespace

struct rcgpu_mkv {
    int fd = -1;
    std::string path;
    std::vector<track> tracks;
    std::vector<attach> attachments;
    std::vector<tag> tags;
    std::vector<cue> cues;
    uint64_t pos = 0;                 // bytes written so far
    uint64_t segment_data = 0;        // file offset of the first byte after the Segment size field
    uint64_t seekhead_pos = 0, info_pos = 0, tracks_pos = 0, tags_pos = 0, attachments_pos = 0, duration_pos = 0;
    bool begun = false;
    // open cluster
    ebuf cluster; uint64_t cluster_ts = 0; bool cluster_open = false; uint64_t cluster_file_pos = 0;
    uint64_t uid_seed = 0x9E3779B97F4A7C15ull;
    // parallel writers (rcgpu_mkv_expect / reserve_block / fill): the part of the file that will hold the blocks, mapped shared
    uint8_t* map = nullptr; uint64_t map_base = 0, map_len = 0;
    // tmpfs: pages are allocated ahead of the writers by one thread (fallocate: ~17 GB/s under the inode lock, no copy), so that the
    // writers' page faults only map pages that exist (see rcgpu_mkv_expect)
    std::thread prealloc; std::atomic<bool> prealloc_stop{ false }, prealloc_alive{ false }, allocating{ false };
    std::atomic<uint64_t> reserved_to{ 0 }, prealloc_to{ 0 }; std::atomic<int> active_copies{ 0 }; uint64_t prealloc_first = 0;
    // fallocate() said no (ENOSPC, or a file system without it): nothing behind prealloc_to is written through the mapping from then on --
    // a fault on a page tmpfs cannot back is a SIGBUS, a pwrite() there is an error code
    std::atomic<bool> alloc_failed{ false };

    uint64_t next_uid() { uid_seed ^= uid_seed << 13; uid_seed ^= uid_seed >> 7; uid_seed ^= uid_seed << 17; return uid_seed | 1; }
    int put(const void* p, size_t n)
    {
        const uint8_t* s = static_cast<const uint8_t*>(p);
        // inside the mapped part of the file the bytes go through the mapping: a pwrite() there would queue behind the thread that
        // allocates pages ahead (fallocate holds the file's lock for a whole chunk) -- 7 ms per block head, measured.  But only where the
        // pages EXIST (pos + n <= prealloc_to): a head written beyond them faults a tmpfs page in, and if the file system fills at that
        // moment the fault is a SIGBUS, where the pwrite() below returns an error.  Heads are a few bytes; the allocation runs 2 GB ahead.
        if (map && pos >= map_base && pos + n <= map_base + map_len && pos + n <= prealloc_to.load()) { memcpy(map + (pos - map_base), p, n); pos += n; return 0; }
        while (n) {
            ssize_t w = ::pwrite(fd, s, n, off_t(pos));     // positional: other threads fill reserved blocks through the same descriptor
            if (w < 0) { if (errno == EINTR) continue; return fail(20, "mkv: write to %s failed: %s", path.c_str(), strerror(errno)); }
            s += w; n -= size_t(w); pos += uint64_t(w);
        }
        return 0;
    }
    int put_at(uint64_t off, const void* p, size_t n)
    {
        if (::pwrite(fd, p, n, off_t(off)) != ssize_t(n)) return fail(21, "mkv: patching %s failed: %s", path.c_str(), strerror(errno));
        return 0;
    }
    int flush_cluster()
    {
        if (!cluster_open) return 0;
        ebuf head; head.id(0x1F43B675); head.size(cluster.b.size());
        cluster_open = false;
        if (int r = put(head.b.data(), head.b.size())) return r;
        return put(cluster.b.data(), cluster.b.size());
    }
};

static const uint64_t kSeekHeadReserve = 192;     // room for 6 Seek entries (Info, Tracks, Tags, Attachments, Cues + one spare), padded with Void

extern "C" int rcgpu_mkv_open(const char* path, int overwrite, rcgpu_mkv** out)
{
    clear_error();
    if (!path || !out) return fail(1, "mkv: null argument");
    int flags = O_RDWR | O_CREAT | (overwrite ? O_TRUNC : O_EXCL);       // O_RDWR: a shared mapping for writing needs read access
    int fd = ::open(path, flags, 0644);
    if (fd < 0) return fail(2, "mkv: cannot create %s: %s", path, strerror(errno));
    rcgpu_mkv* m = new rcgpu_mkv;
    m->fd = fd; m->path = path;
    for (const char* c = path; *c; c++) m->uid_seed = m->uid_seed * 1099511628211ull + uint8_t(*c);
    *out = m;
    return 0;
}

extern "C" int rcgpu_mkv_add_video(rcgpu_mkv* m, const uint8_t* cp, size_t cp_size, uint32_t w, uint32_t h, uint32_t fps_num, uint32_t fps_den)
{
    if (!m || m->begun) { fail(1, "mkv: add_video after begin"); return -1; }
    if (w > 0xFFFF || h > 0xFFFF) { fail(1, "mkv: picture size above 65535 is not readable by the reference (Matroska.cpp:1007-1030)"); return -1; }
    if (m->tracks.size() >= 126) { fail(1, "mkv: too many tracks"); return -1; }
    track t; t.video = true; t.codec_private.assign(cp, cp + cp_size); t.width = w; t.height = h;
    t.fps_num = fps_num ? fps_num : 24; t.fps_den = fps_den ? fps_den : 1;
    m->tracks.push_back(t);
    return int(m->tracks.size());
}

extern "C" int rcgpu_mkv_add_audio(rcgpu_mkv* m, const uint8_t* cp, size_t cp_size, uint32_t ch, uint32_t rate, uint32_t bits)
{
    if (!m || m->begun) { fail(1, "mkv: add_audio after begin"); return -1; }
    if (m->tracks.size() >= 126) { fail(1, "mkv: too many tracks"); return -1; }
    track t; t.video = false; if (cp && cp_size) t.codec_private.assign(cp, cp + cp_size); t.channels = ch; t.sample_rate = rate; t.bits = bits;
    m->tracks.push_back(t);
    return int(m->tracks.size());
}

extern "C" int rcgpu_mkv_add_audio_pcm(rcgpu_mkv* m, int is_float, uint32_t ch, uint32_t rate, uint32_t bits)
{
    // what FFmpeg's muxer writes for pcm_u8 / pcm_s16le / pcm_s24le / pcm_s32le and pcm_f32le; the reader maps all of them to its
    // pass-through wrapper (Lib/CoDec/Wrapper.cpp:41-49,376-388) and takes depth, sign and endianness from the reversibility data
    const int trk = rcgpu_mkv_add_audio(m, nullptr, 0, ch, rate, bits);
    if (trk > 0) m->tracks.back().codec_id = is_float ? "A_PCM/FLOAT/IEEE" : "A_PCM/INT/LIT";
    return trk;
}

extern "C" int rcgpu_mkv_add_attachment(rcgpu_mkv* m, const char* name, const char* mime, const uint8_t* data, size_t size)
{
    if (!m || m->begun) return fail(1, "mkv: add_attachment after begin");
    attach a; a.name = name ? name : ""; a.mime = mime ? mime : "application/octet-stream"; a.data.assign(data, data + size);
    m->attachments.push_back(std::move(a));
    return 0;
}

extern "C" int rcgpu_mkv_add_tag(rcgpu_mkv* m, int trk, const char* name, const char* value)
{
    if (!m || m->begun || !name || !value) return fail(1, "mkv: add_tag after begin or null argument");
    if (trk < 1 || size_t(trk) > m->tracks.size()) return fail(1, "mkv: bad track number %d", trk);
    m->tags.push_back({ trk, name, value });
    return 0;
}

extern "C" int rcgpu_mkv_begin(rcgpu_mkv* m)
{
    clear_error();
    if (!m || m->begun) return fail(1, "mkv: begin called twice");
    if (m->tracks.empty()) return fail(1, "mkv: no track");
    m->begun = true;
    ebuf f;
    {   // EBML header
        ebuf h;
        h.uint(0x4286, 1); h.uint(0x42F7, 1); h.uint(0x42F2, 4); h.uint(0x42F3, 8);
        h.str(0x4282, "matroska"); h.uint(0x4287, 4); h.uint(0x4285, 2);
        f.master(0x1A45DFA3, h);
    }
    f.id(0x18538067); f.size_n(0, 8);             // Segment, size patched by close()
    m->segment_data = f.b.size();
    m->seekhead_pos = f.b.size();
    f.voidel(kSeekHeadReserve);                   // SeekHead goes here at close()
    {   // Info
        m->info_pos = f.b.size();
        ebuf i;
        i.uint(0x2AD7B1, 1000000);                // TimestampScale: 1 ms
        i.str(0x4D80, "rcgpu"); i.str(0x5741, rcgpu_version());
        { uint8_t uid[16]; for (int k = 0; k < 16; k += 8) { uint64_t u = m->next_uid(); memcpy(uid + k, &u, 8); } i.bin(0x73A4, uid, 16); }
        const size_t dur_rel = i.b.size();
        i.f64(0x4489, 0.0);                       // Duration, patched by close()
        ebuf tmp; tmp.id(0x1549A966); tmp.size(i.b.size());
        m->duration_pos = f.b.size() + tmp.b.size() + dur_rel + 3;   // id(2) + size(1) -> payload
        f.master(0x1549A966, i);
    }
    {   // Tracks
        m->tracks_pos = f.b.size();
        ebuf ts;
        for (size_t k = 0; k < m->tracks.size(); k++) {
            track& t = m->tracks[k];
            ebuf e;
            t.uid = m->next_uid() & 0x7FFFFFFFFFFFFFFFull;
            e.uint(0xD7, k + 1); e.uint(0x73C5, t.uid);
            e.uint(0x83, t.video ? 1 : 2); e.uint(0x9C, 0);
            e.str(0x22B59C, "und");
            e.str(0x86, t.video ? "V_FFV1" : t.codec_id ? t.codec_id : "A_FLAC");
            const size_t cp_rel_before = e.b.size();
            if (!t.codec_private.empty()) e.bin(0x63A2, t.codec_private.data(), t.codec_private.size());      // FFV1 version 1 has none
            const size_t cp_payload_rel = e.b.size() - t.codec_private.size();
            (void)cp_rel_before;
            if (t.video) {
                e.uint(0x23E383, uint64_t(std::llround(1e9 * double(t.fps_den) / double(t.fps_num))));   // DefaultDuration
                ebuf v; v.uint(0x9A, 2); v.uint(0xB0, t.width); v.uint(0xBA, t.height); v.uint(0x54B0, t.width); v.uint(0x54BA, t.height);
                e.master(0xE0, v);
            } else {
                ebuf a; a.f64(0xB5, double(t.sample_rate)); a.uint(0x9F, t.channels); a.uint(0x6264, t.bits);
                e.master(0xE1, a);
            }
            ebuf tmp; tmp.id(0xAE); tmp.size(e.b.size());
            // remember where this entry's CodecPrivate payload lands (relative to the Tracks payload for now)
            t.cp_file_pos = ts.b.size() + tmp.b.size() + cp_payload_rel;
            ts.master(0xAE, e);
        }
        ebuf tmp; tmp.id(0x1654AE6B); tmp.size(ts.b.size());
        for (track& t : m->tracks) t.cp_file_pos += f.b.size() + tmp.b.size();
        f.master(0x1654AE6B, ts);
    }
    if (!m->tags.empty()) {   // Tags: what FFmpeg writes for -metadata:s:N key=value (e.g. the reference's EXR warning, Main.cpp / Output.cpp:273-278)
        m->tags_pos = f.b.size();
        ebuf all;
        for (const tag& g : m->tags) {
            ebuf targets; targets.uint(0x68CA, 30); targets.uint(0x63C5, m->tracks[size_t(g.track) - 1].uid);      // TargetTypeValue 30 = track
            ebuf simple; simple.str(0x45A3, g.name); simple.str(0x447A, "und"); simple.str(0x4487, g.value);
            ebuf one; one.master(0x63C0, targets); one.master(0x67C8, simple);
            all.master(0x7373, one);
        }
        f.master(0x1254C367, all);
    }
    if (int r = m->put(f.b.data(), f.b.size())) return r;
    if (!m->attachments.empty()) {   // Attachments (before any Cluster, Matroska.cpp:863-873); streamed to keep big sidecars out of one buffer
        m->attachments_pos = m->pos;
        std::vector<ebuf> heads(m->attachments.size());
        uint64_t total = 0;
        for (size_t k = 0; k < m->attachments.size(); k++) {
            attach& a = m->attachments[k];
            ebuf inner;
            inner.str(0x466E, a.name); inner.str(0x4660, a.mime); inner.uint(0x46AE, m->next_uid() & 0x7FFFFFFFFFFFFFFFull);
            inner.id(0x465C); inner.size(a.data.size());
            ebuf& h = heads[k];
            h.id(0x61A7); h.size(inner.b.size() + a.data.size()); h.raw(inner.b.data(), inner.b.size());
            total += h.b.size() + a.data.size();
        }
        ebuf top; top.id(0x1941A469); top.size(total);
        if (int r = m->put(top.b.data(), top.b.size())) return r;
        for (size_t k = 0; k < m->attachments.size(); k++) {
            if (int r = m->put(heads[k].b.data(), heads[k].b.size())) return r;
            if (int r = m->put(m->attachments[k].data.data(), m->attachments[k].data.size())) return r;
            std::vector<uint8_t>().swap(m->attachments[k].data);
        }
    }
    return 0;
}

extern "C" int rcgpu_mkv_write_block(rcgpu_mkv* m, int trk, uint64_t pts_ns, const uint8_t* data, size_t size, int keyframe)
{
    if (!m || !m->begun) return fail(1, "mkv: write_block before begin");
    if (trk < 1 || size_t(trk) > m->tracks.size()) return fail(1, "mkv: bad track number %d", trk);
    track& t = m->tracks[size_t(trk) - 1];
    const uint64_t ms = (pts_ns + 500000) / 1000000;
    // FFmpeg-like clustering: a new Cluster on every video keyframe once the open one holds >= 5 MB or spans
    // >= 5 s, and always before the 16-bit relative timestamp would overflow.
    const bool big = m->cluster.b.size() + size > (5u << 20);
    const bool far = m->cluster_open && (ms < m->cluster_ts || ms - m->cluster_ts > 5000);
    if (m->cluster_open && (far || (big && (t.video || m->tracks.size() == 1)) || ms - m->cluster_ts > 32000))
        if (int r = m->flush_cluster()) return r;
    if (!m->cluster_open) {
        m->cluster.b.clear();
        m->cluster.uint(0xE7, ms);
        m->cluster_ts = ms; m->cluster_open = true; m->cluster_file_pos = m->pos;
    }
    if (t.video && keyframe)
        m->cues.push_back({ ms, trk, m->cluster_file_pos - m->segment_data });
    ebuf& c = m->cluster;
    c.id(0xA3); c.size(size + 4);
    c.b.push_back(uint8_t(0x80 | trk));
    const int16_t rel = int16_t(ms - m->cluster_ts);
    c.b.push_back(uint8_t(uint16_t(rel) >> 8)); c.b.push_back(uint8_t(rel));
    c.b.push_back(keyframe ? 0x80 : 0x00);
    if (size > (1u << 20)) {      // large frame: write the cluster directly, avoiding a second copy of 50 MB packets
        ebuf head; head.id(0x1F43B675); head.size(c.b.size() + size);
        m->cluster_open = false;
        if (int r = m->put(head.b.data(), head.b.size())) return r;
        if (int r = m->put(c.b.data(), c.b.size())) return r;
        if (int r = m->put(data, size)) return r;
    } else
        c.raw(data, size);
    t.last_pts_ms = ms;
    uint64_t end = ms;
    if (t.video) end = ms + uint64_t(std::llround(1000.0 * t.fps_den / t.fps_num));
    if (end > t.end_ms) t.end_ms = end;
    return 0;
}

// ---- parallel writers.  A job that moves ~30 GB/s of packets cannot push them through one write() loop (a tmpfs or page-cache write
// is a single-threaded copy under the inode lock, ~5 GB/s).  The muxer therefore only lays the file out -- Cluster and SimpleBlock
// heads in stream order, payload bytes reserved -- and hands back where each payload goes; any number of threads then copy payloads
// into a shared mapping of the file (page faults on distinct pages do not serialise) or, without a mapping, pwrite() them.
extern "C" int rcgpu_mkv_expect(rcgpu_mkv* m, uint64_t max_block_bytes, uint64_t max_blocks)
{
    clear_error();
    if (!m || !m->begun) return fail(1, "mkv: expect before begin");
    if (m->map) return 0;
    if (const char* e = getenv("RCGPU_MKV_NO_MMAP")) if (*e && *e != '0') return 0;
    // Measured on the GPU box (tools/probe_fs.py, one 8 GiB file): a page-cache file takes ~12 GB/s through pwrite() from one or
    // many threads, and only 2-4 GB/s through a shared mapping (every fault allocates a page under the file's locks); tmpfs takes
    // 6 GB/s through pwrite, 3-4 GB/s through a mapping -- but 57 GB/s through a mapping once fallocate() has allocated the pages
    // (17 GB/s, one thread) and the writers pre-fault their range with MADV_POPULATE_WRITE.  So: map on tmpfs only, and allocate ahead.
    struct statfs sf;
    const bool force = getenv("RCGPU_MKV_MMAP") != nullptr;
    if (!force && (fstatfs(m->fd, &sf) != 0 || uint32_t(sf.f_type) != 0x01021994u)) return 0;          // TMPFS_MAGIC
    const uint64_t page = 4096;
    const uint64_t base = m->pos & ~(page - 1);
    const uint64_t len = (m->pos - base) + max_block_bytes + max_blocks * 64 + (1u << 20);
    if (ftruncate(m->fd, off_t(base + len)) != 0) return 0;                 // no sparse files here: stay with pwrite
    void* p = mmap(nullptr, size_t(len), PROT_READ | PROT_WRITE, MAP_SHARED, m->fd, off_t(base));
    if (p == MAP_FAILED) { if (ftruncate(m->fd, off_t(m->pos)) != 0) {} return 0; }
    m->map = static_cast<uint8_t*>(p); m->map_base = base; m->map_len = len;
    m->reserved_to = m->pos; m->prealloc_to = base; m->prealloc_alive = true;
    // the first stride starts at once, while the encoders are still being set up: packets are about half of their worst case, and a
    // small job must not hold more memory than it writes
    m->prealloc_first = base + std::min<uint64_t>(max_block_bytes / 2, uint64_t(8) << 30);

    m->prealloc = std::thread([m] {
        // fallocate() and the writers' page faults slow each other down when they overlap (a fault that meets an allocation in progress
        // takes the file's spin lock; measured: 6 GB/s together against 17 GB/s + 57 GB/s apart), so they take turns: allocation
        // happens in strides during which no copy is running.  The first stride starts at once -- while the encoders are still being
        // set up -- and covers what the job will probably write.
        const uint64_t chunk = uint64_t(64) << 20, ahead = uint64_t(2) << 30, stride = uint64_t(4) << 30;
        uint64_t done = m->map_base;
        while (!m->prealloc_stop.load()) {
            const uint64_t end = m->map_base + m->map_len;
            const uint64_t want = std::min<uint64_t>(end, std::max<uint64_t>(m->prealloc_first, m->reserved_to.load() + ahead));
            if (done >= want) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
            m->allocating = true;
            while (m->active_copies.load() > 0 && !m->prealloc_stop.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
            const uint64_t upto = std::min<uint64_t>(want, done + stride);
            bool ok = true;
            while (done < upto && !m->prealloc_stop.load()) {
                const uint64_t n = std::min<uint64_t>(chunk, upto - done);
                if (fallocate(m->fd, 0, off_t(done), off_t(n)) != 0) { ok = false; break; }   // not supported, or no room
                done += n;
                m->prealloc_to = done;
            }
            if (!ok) m->alloc_failed = true;         // before the writers are let go: they take the pwrite() path behind prealloc_to
            m->allocating = false;
            if (!ok) break;
        }
        m->allocating = false;
        m->prealloc_alive = false;
    });
    return 0;
}

// Writer threads: copy a payload to where reserve_block() put it.  Waits until the allocating thread has passed the range (a fault on a
// page that does not exist yet allocates it the slow way), stays out of its strides, maps the range in one call instead of one fault
// per page, copies.
extern "C" int rcgpu_mkv_copy_in(rcgpu_mkv* m, uint8_t* dst, const uint8_t* src, size_t size)
{
    if (!m || !m->map || !dst || dst < m->map || dst + size > m->map + m->map_len) { if (dst && src) memcpy(dst, src, size); return 0; }
    const uint64_t begin_off = m->map_base + uint64_t(dst - m->map), end_off = begin_off + size;
    for (;;) {
        while (m->prealloc_alive.load() && (m->prealloc_to.load() < end_off || m->allocating.load())) std::this_thread::sleep_for(std::chrono::microseconds(50));
        m->active_copies++;
        if (!m->prealloc_alive.load() || !m->allocating.load()) break;
        m->active_copies--;                                  // a stride began in between: step back
    }
    if (m->alloc_failed.load() && end_off > m->prealloc_to.load()) {
        // the pages were never allocated: through the descriptor, where "no space left" is an error code and not a SIGBUS
        m->active_copies--;
        return rcgpu_mkv_fill(m, begin_off, src, size);
    }
    const uintptr_t a = reinterpret_cast<uintptr_t>(dst) & ~uintptr_t(4095), b = (reinterpret_cast<uintptr_t>(dst) + size + 4095) & ~uintptr_t(4095);
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
    // maps the range in one call instead of one fault per page.  EINVAL: a kernel without it, the copy faults page by page.  Anything
    // else (ENOMEM, EFAULT: the pages cannot be had) would be a SIGBUS inside the memcpy: pwrite() instead, and report what it says.
    if (madvise(reinterpret_cast<void*>(a), size_t(b - a), MADV_POPULATE_WRITE) != 0 && errno != EINVAL) {
        m->active_copies--;
        return rcgpu_mkv_fill(m, begin_off, src, size);
    }
    memcpy(dst, src, size);
    m->active_copies--;
    return 0;
}

extern "C" int rcgpu_mkv_reserve_block(rcgpu_mkv* m, int trk, uint64_t pts_ns, size_t size, int keyframe, uint8_t** dst, uint64_t* file_offset)
{
    if (!m || !m->begun) return fail(1, "mkv: reserve_block before begin");
    if (trk < 1 || size_t(trk) > m->tracks.size()) return fail(1, "mkv: bad track number %d", trk);
    if (!dst || !file_offset) return fail(1, "mkv: null argument");
    track& t = m->tracks[size_t(trk) - 1];
    const uint64_t ms = (pts_ns + 500000) / 1000000;
    if (int r = m->flush_cluster()) return r;            // small blocks (audio) buffered so far come first
    // one Cluster per reserved block: Timestamp + SimpleBlock head, payload reserved behind them
    ebuf c; c.uint(0xE7, ms);
    c.id(0xA3); c.size(size + 4);
    c.b.push_back(uint8_t(0x80 | trk)); c.b.push_back(0); c.b.push_back(0); c.b.push_back(keyframe ? 0x80 : 0x00);
    ebuf head; head.id(0x1F43B675); head.size(c.b.size() + size);
    if (t.video && keyframe) m->cues.push_back({ ms, trk, m->pos - m->segment_data });
    if (int r = m->put(head.b.data(), head.b.size())) return r;
    if (int r = m->put(c.b.data(), c.b.size())) return r;
    *file_offset = m->pos;
    *dst = (m->map && m->pos >= m->map_base && m->pos + size <= m->map_base + m->map_len
            && !(m->alloc_failed.load() && m->pos + size > m->prealloc_to.load())) ? m->map + (m->pos - m->map_base) : nullptr;
    m->pos += size;
    m->reserved_to = m->pos;
    t.last_pts_ms = ms;
    uint64_t end = ms;
    if (t.video) end = ms + uint64_t(std::llround(1000.0 * t.fps_den / t.fps_num));
    if (end > t.end_ms) t.end_ms = end;
    return 0;
}

extern "C" int rcgpu_mkv_fill(rcgpu_mkv* m, uint64_t file_offset, const uint8_t* data, size_t size)
{
    if (!m || m->fd < 0) return fail(1, "mkv: fill on a closed file");
    while (size) {
        const ssize_t w = ::pwrite(m->fd, data, size, off_t(file_offset));
        if (w < 0) { if (errno == EINTR) continue; return fail(20, "mkv: write to %s failed: %s", m->path.c_str(), strerror(errno)); }
        data += w; size -= size_t(w); file_offset += uint64_t(w);
    }
    return 0;
}

extern "C" int rcgpu_mkv_update_codec_private(rcgpu_mkv* m, int trk, const uint8_t* cp, size_t cp_size)
{
    if (!m || !m->begun) return fail(1, "mkv: update_codec_private before begin");
    if (trk < 1 || size_t(trk) > m->tracks.size()) return fail(1, "mkv: bad track number %d", trk);
    track& t = m->tracks[size_t(trk) - 1];
    if (cp_size != t.codec_private.size()) return fail(1, "mkv: CodecPrivate size changed");
    t.codec_private.assign(cp, cp + cp_size);
    return m->put_at(t.cp_file_pos, cp, cp_size);
}

extern "C" int rcgpu_mkv_close(rcgpu_mkv* m)
{
    if (!m) return 0;
    int r = 0;
    if (m->prealloc.joinable()) { m->prealloc_stop = true; m->prealloc.join(); }
    if (m->map) {      // blocks were laid out inside a mapping of a generously sized file: cut it back to what was used
        // (12 million page-table entries for a 50 GB file: 1.8 s in this one thread.  Dropping them per packet with MADV_DONTNEED cost the
        // pipeline 1.9 s in TLB shoot-downs, dropping them here from eight threads took 4 s: measured, not kept.  The job overlaps this
        // call with giving back its device and pinned memory instead.)
        // Unmapping finished ranges during the allocation strides instead was measured in round 3: close() 1.8 -> 0.3 s, the pipeline
        // 4.3 -> 6.0 s -- the strides wait for the unmapping, the writers for the strides.
        munmap(m->map, size_t(m->map_len)); m->map = nullptr;
        if (ftruncate(m->fd, off_t(m->pos)) != 0) r = fail(22, "mkv: cannot size %s: %s", m->path.c_str(), strerror(errno));
    }
    if (m->begun && !r) {
        r = m->flush_cluster();
        uint64_t cues_pos = 0;
        if (!r && !m->cues.empty()) {
            cues_pos = m->pos;
            ebuf cs;
            for (const cue& c : m->cues) {
                ebuf tp; tp.uint(0xF7, uint64_t(c.track)); tp.uint(0xF1, c.cluster_pos);
                ebuf cp; cp.uint(0xB3, c.time_ms); cp.master(0xB7, tp);
                cs.master(0xBB, cp);
            }
            ebuf top; top.master(0x1C53BB6B, cs);
            r = m->put(top.b.data(), top.b.size());
        }
        if (!r) {   // Segment size
            ebuf s; s.size_n(m->pos - m->segment_data, 8);
            r = m->put_at(m->segment_data - 8, s.b.data(), 8);
        }
        if (!r) {   // SeekHead in the reserved Void
            ebuf sh;
            auto seek = [&](uint32_t id, uint64_t at) {
                ebuf e; ebuf idb; idb.id(id); e.bin(0x53AB, idb.b.data(), idb.b.size()); e.uint_n(0x53AC, at - m->segment_data, 8); sh.master(0x4DBB, e);
            };
            seek(0x1549A966, m->info_pos); seek(0x1654AE6B, m->tracks_pos);
            if (m->tags_pos) seek(0x1254C367, m->tags_pos);
            if (m->attachments_pos) seek(0x1941A469, m->attachments_pos);
            if (cues_pos) seek(0x1C53BB6B, cues_pos);
            ebuf top; top.master(0x114D9B74, sh);
            if (top.b.size() + 2 <= kSeekHeadReserve) {
                top.voidel(kSeekHeadReserve - top.b.size());
                r = m->put_at(m->seekhead_pos, top.b.data(), top.b.size());
            }
        }
        if (!r) {   // Duration (ms)
            uint64_t end = 0;
            for (const track& t : m->tracks) end = std::max(end, t.end_ms);
            double d = double(end); uint64_t u; memcpy(&u, &d, 8);
            uint8_t be[8]; for (int k = 0; k < 8; k++) be[k] = uint8_t(u >> (8 * (7 - k)));
            r = m->put_at(m->duration_pos, be, 8);
        }
    }
    if (::close(m->fd) != 0 && !r) r = fail(22, "mkv: closing %s failed: %s", m->path.c_str(), strerror(errno));
    delete m;
    return r;
}
