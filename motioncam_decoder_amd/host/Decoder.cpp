// Decoder.cpp -- container side of motioncam::Decoder (host only) and the hand-off of
// compressed frames to the MI355X decode path.  Written from the container layout
// (Container.hpp); behaviour and error texts follow the reference's public contract
// (lib/Decoder.cpp:97-319) so callers cannot tell the difference, the mechanics do not:
// positional reads, one index built up front, batched GPU decode.
#include <motioncam/Decoder.hpp>
#include <motioncam/RawData.hpp>
#include <motioncam/mcraw_container.h>

#include "mcraw_hip.h"
#include "WorkerPool.hpp"

#include <algorithm>
#if defined(__linux__)
#include <sys/mman.h>
#endif
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <cstring>
#include <unistd.h>

namespace motioncam {

namespace {

using namespace container;

constexpr int kTypeLegacy = MCRAW_TYPE_LEGACY; // frame JSON "compressionType" (lib/Decoder.cpp:20-21)
constexpr int kTypeBlock = MCRAW_TYPE_BLOCK;

// Positional reader over the container file: no shared file position, so frame
// payloads can be fetched in any order (and, later, from several threads).
class FileReader {
public:
    explicit FileReader(FILE *f) : mFile(f), mFd(f ? fileno(f) : -1) {}
    ~FileReader()
    {
        if (mFile)
            std::fclose(mFile);
    }
    bool ok() const { return mFd >= 0; }

    // Reads exactly `size` bytes at `offset` or throws.
    void readAt(int64_t offset, void *dst, size_t size) const
    {
        if (!tryReadAt(offset, dst, size))
            throw IOException("Failed to read data");
    }
    bool tryReadAt(int64_t offset, void *dst, size_t size) const
    {
        if (offset < 0)
            return false;
        uint8_t *p = static_cast<uint8_t *>(dst);
        while (size > 0) {
            ssize_t n = ::pread(mFd, p, size, static_cast<off_t>(offset));
            if (n < 0 && errno == EINTR)
                continue;
            if (n <= 0)
                return false;
            p += n;
            offset += n;
            size -= static_cast<size_t>(n);
        }
        return true;
    }
    int64_t size() const
    {
        off_t end = ::lseek(mFd, 0, SEEK_END);
        return static_cast<int64_t>(end);
    }

private:
    FILE *mFile;
    int mFd;
};

template <typename T> T readPod(const FileReader &r, int64_t offset)
{
    T v{};
    r.readAt(offset, &v, sizeof(T));
    return v;
}

nlohmann::json readJson(const FileReader &r, int64_t offset, uint32_t size)
{
    if (offset < 0 || offset + static_cast<int64_t>(size) > r.size()) // before allocating `size` bytes
        throw IOException("Invalid metadata");
    std::string text(size, '\0');
    if (size)
        r.readAt(offset, &text[0], size);
    return nlohmann::json::parse(text);
}

// Where the pieces of one frame live in the file.
struct FrameSpan {
    int64_t payload = 0;
    uint32_t payloadSize = 0;
    int64_t json = 0;
    uint32_t jsonSize = 0;
};

bool loadAudioChunkAt(const FileReader &r, const Locator &o, AudioChunk &out)
{
    Chunk item{};
    if (!r.tryReadAt(o.position, &item, sizeof(item)))
        return false;
    if (item.kind != Kind::PCM)
        throw IOException("Invalid audio data");
    if (o.position < 0 || o.position + static_cast<int64_t>(sizeof(Chunk)) + static_cast<int64_t>(item.bytes) > r.size())
        throw IOException("Invalid audio data"); // before sizing a vector by a corrupt length
    std::vector<int16_t> samples((static_cast<size_t>(item.bytes) + 1) / 2);
    r.readAt(o.position + static_cast<int64_t>(sizeof(Chunk)), samples.data(), item.bytes);
    // newer files follow the samples with their capture time; older ones do not
    Timestamp ts = -1;
    const int64_t next = o.position + static_cast<int64_t>(sizeof(Chunk)) + item.bytes;
    Chunk meta{};
    r.readAt(next, &meta, sizeof(meta));
    if (meta.kind == Kind::PCM_TIME)
        ts = readPod<PcmTime>(r, next + static_cast<int64_t>(sizeof(Chunk))).timeNs;
    out = std::make_pair(ts, std::move(samples));
    return true;
}

} // namespace

struct Decoder::Impl {
    explicit Impl(FILE *f) : reader(f) {}

    FileReader reader;
    nlohmann::json metadata;
    std::vector<Timestamp> frames;               // ascending
    std::map<Timestamp, int64_t> frameOffsets;   // timestamp -> offset of its BUFFER item
    std::vector<Locator> audioOffsets;

    struct Loader : AudioChunkLoader {
        Loader(const FileReader &r, const std::vector<Locator> &o) : reader(r), offsets(o) {}
        bool next(AudioChunk &output) override
        {
            if (index >= offsets.size() || !loadAudioChunkAt(reader, offsets[index], output))
                return false;
            ++index;
            return true;
        }
        const FileReader &reader;
        const std::vector<Locator> &offsets;
        size_t index = 0;
    };
    std::unique_ptr<Loader> loader;

    // Pinned staging reused across loadFrames() calls.  Every slot has one slice per pool member (GPU),
    // allocated by that member's own thread: on the NUMA node of its GPU.
    static constexpr int kOutSlots = 3; // chunk queued, chunk finishing on the GPU, chunk being copied out
    static constexpr int kInSlots = 3;  // chunk being read, chunk queued, chunk finishing on the GPU
    struct Slice {
        uint8_t *p = nullptr;
        size_t cap = 0;
    };
    std::vector<Slice> pinIn[kInSlots], pinOut[kOutSlots]; // [slot][member]
    // Per-frame callers (the reference's own loop, example.cpp:182-188 over lib/Decoder.cpp:184-235, asks for one frame after the
    // other in index order) are served ahead of their calls, two frames deep: while frame i is copied out to the caller, frame
    // i + 1 -- its payload read during the call before -- is being decoded (a ticket of the pool: pinned in -> GPU -> pinned out),
    // and the payload of frame i + 2 is on its way from the file into the other pinned input buffer.  A call that asks for
    // anything else, or with other output options, or whose frame failed ahead of time, takes the ordinary path and reports
    // what that path reports.
    struct Ahead {
        Slice in[2], out[2];
        struct Read { // a payload on its way into in[buf] (or there already: `done` not valid any more)
            bool armed = false;
            Timestamp ts = 0;
            int64_t payload = 0;
            uint32_t size = 0;
            int buf = 0;
            std::future<bool> done;
        } rd;
        struct Decode { // a frame on the GPU: in[inbuf] -> out[outbuf]
            bool armed = false;
            Timestamp ts = 0;
            int64_t payload = 0;
            int inbuf = 0, outbuf = 0;
            int bits = 16;
            bool black = false;
            size_t outBytes = 0;
            mcraw_frame f{}; // (the ticket's descriptor: lives as long as the ticket)
            mcraw_pool_ticket *ticket = nullptr;
        } dc;
        // The last single-frame call's timestamp: a frame is read ahead only behind a call that walks the index -- the first frame
        // of the file, or the successor of the frame the call before asked for.  A caller that jumps about then never has a
        // payload read that it does not want (9 MB per call at UHD), nor waits at its next call for that read to end.
        bool walked = false;
        Timestamp last = 0;
    } ahead;
    // Both over: was the read good, did the frame decode?  (Whatever the answers, nothing is under way afterwards.)
    void settleAhead(bool &rdOk, bool &dcOk)
    {
        bool ok = true;
        if (ahead.rd.done.valid())
            ok = ahead.rd.done.get();
        rdOk = ahead.rd.armed && ok;
        ahead.rd.armed = rdOk;
        dcOk = false;
        if (ahead.dc.ticket) {
            size_t w = 0;
            int32_t st = 0;
            const int rc = mcraw_pool_ticket_wait(ahead.dc.ticket, &w, &st);
            ahead.dc.ticket = nullptr;
            dcOk = ahead.dc.armed && rc == 0 && st == 0 && w != 0;
        }
        ahead.dc.armed = dcOk;
    }
    std::unique_ptr<detail::WorkerPool> workers; // copy-out of a lone frame, sliced over the threads; made on first use
    std::unique_ptr<detail::WorkerPool> readers; // a frame's payload read ahead, in slices; made on first use
    mcraw_pool *pool = nullptr; // the GPUs this decoder shards its batches over (frame i of a batch -> member i mod G)
    std::vector<int> devices;   // empty: MCRAW_DEVICES / MCRAW_DEVICE / the current device

    void releaseGpu()
    {
        bool rdOk, dcOk; // (a read into a pinned buffer, a frame on the GPU may still be under way)
        settleAhead(rdOk, dcOk);
        ahead.rd.armed = ahead.dc.armed = false;
        for (Slice *set : {ahead.in, ahead.out})
            for (int i = 0; i < 2; i++) {
                mcraw_host_free(set[i].p);
                set[i] = Slice{};
            }
        for (auto &slot : pinIn)
            for (Slice &s : slot)
                mcraw_host_free(s.p);
        for (auto &slot : pinOut)
            for (Slice &s : slot)
                mcraw_host_free(s.p);
        for (auto &slot : pinIn)
            slot.clear();
        for (auto &slot : pinOut)
            slot.clear();
        if (pool)
            mcraw_pool_destroy(pool);
        pool = nullptr;
    }
    ~Impl() { releaseGpu(); }

    void open();
    FrameSpan locate(Timestamp ts) const;
};

void Decoder::Impl::open()
{
    if (!reader.ok())
        throw IOException("Invalid file");

    const FileMagic header = readPod<FileMagic>(reader, 0);
    if (header.version != kVersion)
        throw IOException("Invalid container version");
    if (std::memcmp(header.magic, kMagic, sizeof(kMagic)) != 0)
        throw IOException("Invalid header id");

    // camera metadata follows the header
    const Chunk camera = readPod<Chunk>(reader, sizeof(FileMagic));
    if (camera.kind != Kind::JSON)
        throw IOException("Invalid camera metadata");
    metadata = readJson(reader, sizeof(FileMagic) + sizeof(Chunk), camera.bytes);

    // the frame index hangs off the last 24 bytes of the file
    const int64_t fileSize = reader.size();
    const int64_t tail = fileSize - static_cast<int64_t>(sizeof(Chunk) + sizeof(FrameTable));
    Chunk indexItem{};
    if (tail < 0 || !reader.tryReadAt(tail, &indexItem, sizeof(indexItem)))
        throw IOException("Failed to get end chunk");
    if (indexItem.kind != Kind::FRAME_TABLE)
        throw IOException("Invalid file");
    const FrameTable index = readPod<FrameTable>(reader, tail + static_cast<int64_t>(sizeof(Chunk)));
    if (static_cast<uint32_t>(index.magic) != kFrameTableMagic)
        throw IOException("Corrupted file");
    if (index.rows < 0)
        throw IOException("Invalid index");
    // a corrupted count must not turn into a multi-gigabyte allocation: the rows have to lie inside the file
    if (index.rowsPosition < 0 || index.rowsPosition > fileSize ||
        static_cast<uint64_t>(index.rows) > static_cast<uint64_t>(fileSize - index.rowsPosition) / sizeof(Locator))
        throw IOException("Invalid index");

    std::vector<Locator> offsets(static_cast<size_t>(index.rows));
    if (!offsets.empty())
        reader.readAt(index.rowsPosition, offsets.data(), offsets.size() * sizeof(Locator));
    std::stable_sort(offsets.begin(), offsets.end(),
                     [](const Locator &a, const Locator &b) { return a.time < b.time; });
    for (const Locator &o : offsets) {
        frames.push_back(o.time);
        frameOffsets.insert({o.time, o.position});
    }

    // The audio index, when present, is found by hopping over the items that follow the
    // last (by timestamp) frame.
    if (!offsets.empty()) {
        int64_t pos = offsets.back().position;
        for (;;) {
            Chunk item{};
            if (!reader.tryReadAt(pos, &item, sizeof(item)))
                break;
            pos += static_cast<int64_t>(sizeof(Chunk));
            if (item.kind == Kind::FRAME || item.kind == Kind::JSON || item.kind == Kind::PCM ||
                item.kind == Kind::PCM_TIME) {
                pos += item.bytes;
            } else if (item.kind == Kind::PCM_TABLE) {
                const PcmTable ai = readPod<PcmTable>(reader, pos);
                pos += static_cast<int64_t>(sizeof(PcmTable));
                if (ai.rows < 0 || pos > fileSize ||
                    static_cast<uint64_t>(ai.rows) > static_cast<uint64_t>(fileSize - pos) / sizeof(Locator))
                    break; // not an audio table this file can hold
                audioOffsets.resize(static_cast<size_t>(ai.rows));
                if (!audioOffsets.empty())
                    reader.readAt(pos, audioOffsets.data(), audioOffsets.size() * sizeof(Locator));
                pos += static_cast<int64_t>(audioOffsets.size() * sizeof(Locator));
            } else {
                break;
            }
        }
    }
    loader.reset(new Loader(reader, audioOffsets));
}

FrameSpan Decoder::Impl::locate(Timestamp ts) const
{
    auto it = frameOffsets.find(ts);
    if (it == frameOffsets.end())
        throw IOException("Frame not found (timestamp: " + std::to_string(ts) + ")");
    FrameSpan s;
    Chunk buffer{};
    if (!reader.tryReadAt(it->second, &buffer, sizeof(buffer)))
        throw IOException("Invalid offset");
    if (buffer.kind != Kind::FRAME)
        throw IOException("Invalid buffer type");
    s.payload = it->second + static_cast<int64_t>(sizeof(Chunk));
    s.payloadSize = buffer.bytes;
    const int64_t metaPos = s.payload + buffer.bytes;
    const Chunk meta = readPod<Chunk>(reader, metaPos);
    if (meta.kind != Kind::JSON)
        throw IOException("Invalid metadata");
    s.json = metaPos + static_cast<int64_t>(sizeof(Chunk));
    s.jsonSize = meta.bytes;
    if (s.json + static_cast<int64_t>(s.jsonSize) > reader.size()) // (the payload is inside the file: its JSON item was read behind it)
        throw IOException("Invalid metadata");
    return s;
}

Decoder::Decoder(FILE *file) : mImpl(new Impl(file))
{
    if (!file)
        throw IOException("Invalid file");
    mImpl->open();
}

Decoder::Decoder(const std::string &path) : mImpl(new Impl(std::fopen(path.c_str(), "rb")))
{
    if (!mImpl->reader.ok())
        throw IOException("Failed to open " + path);
    mImpl->open();
}

Decoder::~Decoder() = default;

const std::vector<Timestamp> &Decoder::getFrames() const { return mImpl->frames; }

const nlohmann::json &Decoder::getContainerMetadata() const { return mImpl->metadata; }

int Decoder::audioSampleRateHz() const { return mImpl->metadata.at("extraData").at("audioSampleRate"); } // throws when absent

int Decoder::numAudioChannels() const { return mImpl->metadata.at("extraData").at("audioChannels"); }

void Decoder::loadAudio(std::vector<AudioChunk> &outAudioChunks)
{
    for (const Locator &o : mImpl->audioOffsets) {
        AudioChunk chunk;
        if (loadAudioChunkAt(mImpl->reader, o, chunk))
            outAudioChunks.emplace_back(std::move(chunk));
    }
}

AudioChunkLoader &Decoder::loadAudio() const { return *mImpl->loader; }

void Decoder::loadFramePayload(const Timestamp timestamp, std::vector<uint8_t> &outPayload, nlohmann::json &outMetadata)
{
    const FrameSpan span = mImpl->locate(timestamp);
    if (span.payload + static_cast<int64_t>(span.payloadSize) > mImpl->reader.size())
        throw IOException("Invalid offset");
    outPayload.resize(span.payloadSize);
    if (span.payloadSize)
        mImpl->reader.readAt(span.payload, outPayload.data(), span.payloadSize);
    outMetadata = readJson(mImpl->reader, span.json, span.jsonSize);
}

void Decoder::loadFrame(const Timestamp timestamp, std::vector<uint8_t> &outData, nlohmann::json &outMetadata)
{
    // one-frame batch through the pinned staging of loadFrames (same checks, same error texts)
    std::vector<std::vector<uint8_t>> data(1);
    std::vector<nlohmann::json> meta(1);
    data[0].swap(outData); // keep the caller's capacity across calls
    try {
        loadFrames(std::vector<Timestamp>{timestamp}, data, meta);
    } catch (...) {
        outData.swap(data[0]);
        throw;
    }
    outData.swap(data[0]);
    outMetadata = std::move(meta[0]);
}

// Run fn(i) for i in [0, n) on up to `threads` host threads (file reads and copies out of the
// pinned staging are memory-bound and scale with a few threads).
template <typename F> void parallelFor(size_t n, unsigned threads, F fn)
{
    if (n == 0)
        return;
    threads = static_cast<unsigned>(std::min<size_t>(std::max(1u, threads), n));
    std::vector<std::future<void>> jobs;
    std::atomic<size_t> next{0};
    for (unsigned t = 0; t < threads; t++)
        jobs.emplace_back(std::async(std::launch::async, [&]() {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1))
                fn(i);
        }));
    for (auto &j : jobs)
        j.get(); // rethrows the first exception of a worker
}

// Batched load: the frames are cut into chunks that fit two bounded pinned staging slots, and
// three stages overlap -- chunk c+1 is read from the file (positional reads, several threads) while
// chunk c is decoded on the GPU (one mcraw_decode_batch: H2D / kernels / D2H pipelined on the
// context's streams) and chunk c-1 is copied out of pinned memory into the caller's vectors.
void Decoder::loadFrames(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> &outData,
                         std::vector<nlohmann::json> &outMetadata)
{
    loadFrames(timestamps, outData, outMetadata, FrameOutput());
}

void Decoder::loadFrames(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> &outData,
                         std::vector<nlohmann::json> &outMetadata, const FrameOutput &output)
{
    loadFramesImpl(timestamps, &outData, nullptr, outMetadata, output);
}

void Decoder::loadFramesInto(const std::vector<Timestamp> &timestamps, const std::vector<uint8_t *> &outBuffers,
                             std::vector<nlohmann::json> &outMetadata, const FrameOutput &output)
{
    if (outBuffers.size() != timestamps.size())
        throw IOException("loadFramesInto: one buffer per timestamp");
    for (uint8_t *p : outBuffers)
        if (!p || reinterpret_cast<uintptr_t>(p) % 2 != 0)
            throw IOException("loadFramesInto: buffers must be non-null and 2-byte aligned");
    loadFramesImpl(timestamps, nullptr, &outBuffers, outMetadata, output);
}

int Decoder::bitsForWhiteLevel(double whiteLevel)
{
    for (int b : {10, 12, 14})
        if (whiteLevel <= static_cast<double>((1 << b) - 1))
            return b;
    return 16;
}

void Decoder::useDevices(const std::vector<int> &devices)
{
    mImpl->releaseGpu(); // staging lives on the old members' NUMA nodes
    mImpl->devices = devices;
}

int Decoder::deviceCount()
{
    Impl &I = *mImpl;
    if (!I.pool && mcraw_pool_create(I.devices.empty() ? nullptr : I.devices.data(), static_cast<int>(I.devices.size()), &I.pool) != 0)
        throw IOException(std::string("No GPU to decode on (") + mcraw_pool_last_error() + ")");
    return mcraw_pool_size(I.pool);
}

void Decoder::loadFrameMetadata(const Timestamp timestamp, nlohmann::json &outMetadata)
{
    Impl &I = *mImpl;
    const FrameSpan span = I.locate(timestamp); // throws "Frame not found" like loadFrame
    outMetadata = readJson(I.reader, span.json, span.jsonSize);
}

size_t Decoder::frameBytes(int width, int height, const FrameOutput &output)
{
    if (width <= 0 || height <= 0)
        return 0;
    const size_t rowBytes = (static_cast<size_t>(width) * static_cast<size_t>(output.bitsPerSample) + 7) / 8;
    return rowBytes * static_cast<size_t>(height);
}

// outData (vectors, filled through pinned output slots and copy-out threads) or outBuffers (the
// caller's memory, written by the GPU pipeline directly).
void Decoder::loadFramesImpl(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> *outDataPtr,
                             const std::vector<uint8_t *> *outBuffers, std::vector<nlohmann::json> &outMetadata,
                             const FrameOutput &output)
{
    Impl &I = *mImpl;
    std::vector<std::vector<uint8_t>> noVectors;
    std::vector<std::vector<uint8_t>> &outData = outDataPtr ? *outDataPtr : noVectors;
    const bool direct = outBuffers != nullptr;
    const auto tEnter = std::chrono::steady_clock::now();
    if (output.bitsPerSample != 16 && output.bitsPerSample != 14 && output.bitsPerSample != 12 && output.bitsPerSample != 10)
        throw IOException("Unsupported bitsPerSample (16, 14, 12 or 10)");
    mcraw_post post{};
    if (output.subtractBlackLevel) {
        const nlohmann::json &cm = getContainerMetadata();
        const auto bl = cm.find("blackLevel"); // (no json::contains in the nlohmann the reference vendors)
        if (!cm.is_object() || bl == cm.end() || !bl->is_array() || bl->size() != 4)
            throw IOException("Container metadata has no blackLevel[4]");
        post.flags |= MCRAW_POST_BLACK;
        for (int i = 0; i < 4; i++) {
            const double v = (*bl)[i].get<double>();
            post.black[i] = static_cast<uint16_t>(std::min(65535.0, std::max(0.0, v + 0.5)));
        }
    }
    if (output.bitsPerSample == 12)
        post.flags |= MCRAW_POST_PACK12;
    else if (output.bitsPerSample == 10)
        post.flags |= MCRAW_POST_PACK10;
    else if (output.bitsPerSample == 14)
        post.flags |= MCRAW_POST_PACK14;
    const size_t n = timestamps.size();
    if (!direct)
        outData.resize(n);
    outMetadata.assign(n, nlohmann::json());
    if (n == 0)
        return;
    // locate every frame and parse its JSON (container errors come first, like in the reference)
    std::vector<FrameSpan> spans(n);
    std::vector<mcraw_frame> frames(n);
    std::vector<size_t> outBytes(n); // bytes handed back per frame
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    for (size_t i = 0; i < n; i++) {
        spans[i] = I.locate(timestamps[i]);
        outMetadata[i] = readJson(I.reader, spans[i].json, spans[i].jsonSize);
        const int width = outMetadata[i]["width"];
        const int height = outMetadata[i]["height"];
        const int type = outMetadata[i]["compressionType"];
        if (type != kTypeBlock && type != kTypeLegacy)
            throw IOException("Invalid compression type");
        if (width <= 0 || height <= 0)
            throw IOException("Failed to uncompress frame");
        // before any staging is sized by them: dimensions the payload cannot possibly hold (a corrupt JSON).
        // A type-7 frame carries two 2-byte side-stream records per 64 blocks of 64 samples, a legacy
        // frame one 2-byte record per 16 samples at the very least.
        const uint64_t px = static_cast<uint64_t>(width) * static_cast<uint64_t>(height);
        const uint64_t least = type == kTypeBlock ? px / 1024 : px / 8;
        if (px >= (1ull << 31) || spans[i].payloadSize < least)
            throw IOException(type == kTypeBlock ? "Failed to uncompress frame" : "Failed to uncompress legacy frame");
        mcraw_frame &f = frames[i];
        f.in = nullptr;
        f.out = nullptr;
        f.len = spans[i].payloadSize;
        f.width = width;
        f.height = height;
        f.type = type;
        f.reserved = 0;
        const size_t rowBytes = (static_cast<size_t>(width) * static_cast<size_t>(output.bitsPerSample) + 7) / 8;
        outBytes[i] = rowBytes * static_cast<size_t>(height);
        f.out_capacity = (outBytes[i] + 1) / 2; // counted in uint16 units
    }

    // no GPU: decoding fails, there is no CPU codec behind this class
    if (!I.pool && mcraw_pool_create(I.devices.empty() ? nullptr : I.devices.data(), static_cast<int>(I.devices.size()), &I.pool) != 0)
        throw IOException(std::string(frames[0].type == kTypeBlock ? "Failed to uncompress frame"
                                                                   : "Failed to uncompress legacy frame") +
                          " (" + mcraw_pool_last_error() + ")");
    const size_t G = static_cast<size_t>(mcraw_pool_size(I.pool)); // frame k of a chunk is decoded by member k mod G
    auto grow = [&](Impl::Slice &sl, size_t member, size_t want) {
        if (want <= sl.cap)
            return;
        mcraw_host_free(sl.p);
        sl.p = static_cast<uint8_t *>(mcraw_pool_host_alloc(I.pool, static_cast<int>(member), want));
        sl.cap = sl.p ? want : 0;
        if (!sl.p)
            throw IOException("Failed to allocate pinned staging");
    };
    const unsigned hostThreads = std::max(2u, std::min(8u, std::thread::hardware_concurrency() / 2));
    const unsigned readThreads = std::max(2u, std::min(16u, std::thread::hardware_concurrency() / 2));
    // A fresh vector of a frame's size is fresh memory: its first touch faults in 4 000 pages of 4 KiB per UHD frame, which is what
    // the copy-out spends its time on.  Where the kernel hands out transparent huge pages on request (THP mode "madvise"), asking
    // for them before the first touch makes that eight faults of 2 MiB.  (Linux only; a no-op elsewhere and for vectors that own
    // enough memory already.)
    auto reserveHuge = [](std::vector<uint8_t> &v, size_t bytes) {
        if (v.capacity() >= bytes)
            return;
        std::vector<uint8_t>().swap(v);
        v.reserve(bytes);
#if defined(__linux__) && defined(MADV_HUGEPAGE)
        const uintptr_t a = (reinterpret_cast<uintptr_t>(v.data()) + 4095u) & ~static_cast<uintptr_t>(4095u);
        const uintptr_t e = (reinterpret_cast<uintptr_t>(v.data()) + bytes) & ~static_cast<uintptr_t>(4095u);
        if (e > a && bytes >= (4u << 20))
            (void)madvise(reinterpret_cast<void *>(a), e - a, MADV_HUGEPAGE);
#endif
    };
    auto copyFrame = [&](std::vector<uint8_t> &dst, const uint8_t *src, size_t bytes) { // one frame, sliced over the threads that stay
        reserveHuge(dst, bytes);
        if (dst.size() != bytes)
            dst.resize(bytes);
        const size_t slice = (bytes + hostThreads - 1) / hostThreads;
        if (!I.workers)
            I.workers.reset(new detail::WorkerPool(hostThreads - 1));
        I.workers->run(hostThreads, [&](size_t t) {
            const size_t lo = std::min(bytes, t * slice), hi = std::min(bytes, lo + slice);
            std::memcpy(dst.data() + lo, src + lo, hi - lo);
        });
    };
    static const bool trace = std::getenv("MCRAW_TRACE") != nullptr;
    static const bool trace2 = trace && std::atoi(std::getenv("MCRAW_TRACE")) >= 2; // a line per chunk
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };

    // ---- one frame asked for: was it decoded, or at least read, ahead of this call?  (Impl::Ahead)
    Impl::Ahead &A = I.ahead;
    bool rdOk = false, dcOk = false;
    const auto tSettle0 = now();
    I.settleAhead(rdOk, dcOk); // (whatever this call is: nothing is under way behind this line)
    const auto tSettle1 = now();
    const bool single = n == 1 && !direct;
    // the payload of the frame behind `after` in the index, on its way into in[buf]
    auto startRead = [&](Timestamp after, int buf) {
        A.rd.armed = false;
        try {
            const auto it = I.frameOffsets.upper_bound(after);
            if (it == I.frameOffsets.end())
                return;
            const FrameSpan nx = I.locate(it->first);
            if (!nx.payloadSize || nx.payloadSize > (1u << 30))
                return;
            grow(A.in[buf], 0, up(nx.payloadSize));
            A.rd.ts = it->first;
            A.rd.payload = nx.payload;
            A.rd.size = nx.payloadSize;
            A.rd.buf = buf;
            uint8_t *dst = A.in[buf].p;
            const FileReader *rd = &I.reader;
            if (!I.readers)
                I.readers.reset(new detail::WorkerPool(3));
            detail::WorkerPool *pool = I.readers.get();
            // (one thread moves a UHD frame's 9 MB out of the page cache in 0.9 ms, which would be the loop's period: four slices)
            A.rd.done = std::async(std::launch::async, [rd, nx, dst, pool]() {
                std::atomic<bool> good{true};
                const size_t slice = ((static_cast<size_t>(nx.payloadSize) + 3) / 4 + 4095) & ~static_cast<size_t>(4095);
                pool->run(4, [&](size_t t) {
                    const size_t lo = std::min<size_t>(nx.payloadSize, t * slice), hi = std::min<size_t>(nx.payloadSize, lo + slice);
                    try {
                        if (hi > lo)
                            rd->readAt(nx.payload + static_cast<int64_t>(lo), dst + lo, hi - lo);
                    } catch (...) {
                        good = false;
                    }
                });
                return good.load();
            });
            A.rd.armed = true;
        } catch (...) { // (a frame that cannot be located is the next call's to report)
            A.rd.armed = false;
        }
    };
    // the frame whose payload has been read (A.rd, settled and good) onto the GPU, with this call's output options; false: not
    // queued (its own call will say why)
    auto submitAhead = [&](int outbuf) {
        A.dc.armed = false;
        try {
            const FrameSpan sp = I.locate(A.rd.ts);
            if (sp.payload != A.rd.payload || sp.payloadSize != A.rd.size)
                return false;
            nlohmann::json m = readJson(I.reader, sp.json, sp.jsonSize);
            const int width = m["width"], height = m["height"], type = m["compressionType"];
            if ((type != kTypeBlock && type != kTypeLegacy) || width <= 0 || height <= 0)
                return false;
            const uint64_t px = static_cast<uint64_t>(width) * static_cast<uint64_t>(height);
            if (px >= (1ull << 31) || sp.payloadSize < (type == kTypeBlock ? px / 1024 : px / 8))
                return false;
            const size_t bytes = frameBytes(width, height, output);
            grow(A.out[outbuf], 0, up(bytes + 2));
            mcraw_frame &f = A.dc.f;
            f = mcraw_frame{};
            f.in = A.in[A.rd.buf].p;
            f.out = reinterpret_cast<uint16_t *>(A.out[outbuf].p);
            f.len = sp.payloadSize;
            f.width = width;
            f.height = height;
            f.type = type;
            f.out_capacity = (bytes + 1) / 2;
            int rc = post.flags ? mcraw_pool_set_post(I.pool, &post) : 0;
            if (rc == 0)
                rc = mcraw_pool_decode_batch_async(I.pool, &f, 1, &A.dc.ticket);
            if (post.flags)
                (void)mcraw_pool_set_post(I.pool, nullptr);
            if (rc != 0) {
                A.dc.ticket = nullptr;
                return false;
            }
            A.dc.ts = A.rd.ts;
            A.dc.payload = A.rd.payload;
            A.dc.inbuf = A.rd.buf;
            A.dc.outbuf = outbuf;
            A.dc.bits = output.bitsPerSample;
            A.dc.black = output.subtractBlackLevel;
            A.dc.outBytes = bytes;
            A.dc.armed = true;
            A.rd.armed = false; // (its buffer is the decode's now)
            return true;
        } catch (...) {
            return false;
        }
    };
    bool hitRd = false;
    if (!single) {
        A.rd.armed = A.dc.armed = false;
    } else {
        const mcraw_frame &f0 = frames[0];
        const bool hitDc = dcOk && A.dc.ts == timestamps[0] && A.dc.payload == spans[0].payload && A.dc.f.len == f0.len &&
                           A.dc.f.width == f0.width && A.dc.f.height == f0.height && A.dc.f.type == f0.type &&
                           A.dc.bits == output.bitsPerSample && A.dc.black == output.subtractBlackLevel && A.dc.outBytes == outBytes[0];
        A.dc.armed = false;
        if (hitDc) {
            const int ob = A.dc.outbuf, freeIn = A.dc.inbuf;
            // the frame behind this one: read already?  Onto the GPU now, and the one behind THAT on its way from the file --
            // both while this frame is copied out and the caller works on it
            const auto nxt = I.frameOffsets.upper_bound(timestamps[0]);
            const bool nextRead = rdOk && nxt != I.frameOffsets.end() && A.rd.ts == nxt->first;
            if (nextRead) {
                const Timestamp t = A.rd.ts;
                if (submitAhead(ob ^ 1))
                    startRead(t, freeIn);
                // (not queued: its payload stays where it is, the next call decodes it from there)
            } else {
                startRead(timestamps[0], freeIn);
            }
            const auto t1 = now();
            copyFrame(outData[0], A.out[ob].p, outBytes[0]);
            if (trace)
                std::fprintf(stderr, "[mcraw] loadFrames n=1 decoded ahead: pipeline %.2f ms: wait-read 0.00, gpu batch %.2f, wait-copy 0.00, tail copy %.2f (index + JSON %.2f, queue the next %.2f)\n",
                             ms(tEnter, now()), ms(tSettle0, tSettle1), ms(t1, now()), ms(tEnter, tSettle0), ms(tSettle1, t1));
            return;
        }
        hitRd = rdOk && A.rd.ts == timestamps[0] && A.rd.payload == spans[0].payload && A.rd.size == f0.len;
    }

    // chunks: as many frames as fit the staging budget of one slot (at least one frame)
    // (MCRAW_SLOT_MB: 240 UHD frames run through the pipeline in 140 / 117 / 120 / 108 ms with slots of 96 / 128 / 192 / 384 MB -- three
    // for the inputs, three for the outputs -- and the pinned staging of the first call costs 0.2 ms per MB.)
    static const size_t kSlotBudget = []() {
        const char *e = std::getenv("MCRAW_SLOT_MB");
        return (e && std::atoi(e) > 0 ? static_cast<size_t>(std::atoi(e)) : size_t(128)) << 20;
    }();
    struct Chunk {
        size_t first, count, inBytes, outBytes;
    };
    std::vector<Chunk> chunks;
    for (size_t i = 0; i < n;) {
        Chunk c{i, 0, 0, 0};
        while (i < n) {
            const size_t ib = up(frames[i].len), ob = up(frames[i].out_capacity * 2);
            if (c.count > 0 && c.inBytes + c.outBytes + ib + ob > kSlotBudget)
                break;
            c.inBytes += ib;
            c.outBytes += ob;
            c.count++;
            i++;
        }
        chunks.push_back(c);
    }
    // staging bytes a chunk needs from every member's slice
    std::vector<size_t> maxIn(G, 0), maxOut(G, 0);
    for (const Chunk &c : chunks) {
        std::vector<size_t> in(G, 0), out(G, 0);
        for (size_t k = 0; k < c.count; k++) {
            in[k % G] += up(frames[c.first + k].len);
            out[k % G] += up(frames[c.first + k].out_capacity * 2);
        }
        for (size_t m = 0; m < G; m++) {
            maxIn[m] = std::max(maxIn[m], in[m]);
            maxOut[m] = std::max(maxOut[m], out[m]);
        }
    }
    // the GPU stage runs on tickets (below: chunk ci is queued behind chunk ci - 1 before that one is waited for, so the PCIe
    // lanes never drain between chunks): three input slots, and three output slots where the frames go through pinned staging
    const int nslots = static_cast<int>(std::min<size_t>(Impl::kInSlots, chunks.size()));
    const int noutslots = static_cast<int>(std::min<size_t>(Impl::kOutSlots, chunks.size()));
    for (int sl = 0; sl < nslots; sl++) {
        I.pinIn[sl].resize(G);
        for (size_t m = 0; m < G; m++)
            grow(I.pinIn[sl][m], m, maxIn[m]);
    }
    for (int sl = 0; sl < noutslots && !direct; sl++) {
        I.pinOut[sl].resize(G);
        for (size_t m = 0; m < G; m++)
            grow(I.pinOut[sl][m], m, maxOut[m]);
    }
    auto readChunk = [&](size_t ci) { // file -> pinned input slot
        const Chunk &c = chunks[ci];
        std::vector<Impl::Slice> &slot = I.pinIn[ci % nslots];
        std::vector<size_t> off(c.count), fill(G, 0);
        for (size_t k = 0; k < c.count; k++) { // frame k lands in the slice of the GPU that will decode it
            off[k] = fill[k % G];
            fill[k % G] += up(frames[c.first + k].len);
        }
        // A thread moves 2 GB/s out of the page cache into pinned memory, a chunk of seven UHD frames (65 MB) on seven threads
        // takes 5 ms -- twice what the GPU needs for it.  Frames are read in pieces of 2 MiB on twice the threads.
        constexpr size_t PIECE = 2u << 20;
        std::vector<std::pair<size_t, size_t>> tasks; // (frame of the chunk, offset)
        for (size_t k = 0; k < c.count; k++) {
            frames[c.first + k].in = slot[k % G].p + off[k];
            for (size_t o = 0; o < frames[c.first + k].len; o += PIECE)
                tasks.emplace_back(k, o);
        }
        parallelFor(tasks.size(), readThreads, [&](size_t i) {
            const size_t k = tasks[i].first, o = tasks[i].second;
            const mcraw_frame &f = frames[c.first + k];
            I.reader.readAt(spans[c.first + k].payload + static_cast<int64_t>(o), slot[k % G].p + off[k] + o, std::min<size_t>(PIECE, f.len - o));
        });
    };
    auto copyOut = [&](size_t ci) { // pinned output slot -> the caller's vectors
        const Chunk &c = chunks[ci];
        if (c.count >= 3) { // one frame per task
            parallelFor(c.count, hostThreads, [&](size_t k) {
                const mcraw_frame &f = frames[c.first + k];
                const uint8_t *src = reinterpret_cast<const uint8_t *>(f.out);
                reserveHuge(outData[c.first + k], outBytes[c.first + k]);
                outData[c.first + k].assign(src, src + outBytes[c.first + k]);
            });
            return;
        }
        for (size_t k = 0; k < c.count; k++) // few frames: slice every frame over the threads
            copyFrame(outData[c.first + k], reinterpret_cast<const uint8_t *>(frames[c.first + k].out), outBytes[c.first + k]);
    };

    double tWaitRead = 0, tDecode = 0, tWaitCopy = 0;
    const auto tStart = now();
    std::vector<size_t> written(n);
    std::vector<int32_t> status(n);
    if (hitRd) { // read ahead: decoded from where it is, while the frame behind it is read into the other buffer
        const int b = A.rd.buf;
        frames[0].in = A.in[b].p;
        startRead(timestamps[0], b ^ 1);
    } else if (single) {
        const auto it = I.frameOffsets.find(timestamps[0]);
        const bool first = it != I.frameOffsets.end() && it == I.frameOffsets.begin();
        const bool next = A.walked && it != I.frameOffsets.end() && it != I.frameOffsets.begin() && std::prev(it)->first == A.last;
        if (first || next)
            startRead(timestamps[0], 0);
    }
    if (single) {
        A.walked = true;
        A.last = timestamps[0];
    } else if (!direct) {
        A.walked = false; // (a batch in between: the walk, if it is one, starts again)
    }
    std::future<void> reading = hitRd ? std::async(std::launch::deferred, []() {}) : std::async(std::launch::async, readChunk, size_t(0));
    std::vector<std::future<void>> copying(noutslots); // copy-out of the chunk that last used each output slot
    auto checkChunk = [&](size_t ci) {
        const Chunk &c = chunks[ci];
        for (size_t k = c.first; k < c.first + c.count; k++)
            if (status[k] != 0 || written[k] == 0)
                throw IOException(frames[k].type == kTypeBlock ? "Failed to uncompress frame"
                                                               : "Failed to uncompress legacy frame");
    };
    struct Queued { // the chunk whose ticket is still open
        mcraw_pool_ticket *ticket = nullptr;
        size_t ci = 0;
        ~Queued()
        {
            if (ticket) // an exception is on its way out: the batch still writes into the caller's buffers
                (void)mcraw_pool_ticket_wait(ticket, nullptr, nullptr);
        }
    } queued;
    // Wait for a queued chunk.  The ticket is taken out of `queued` first, so that whatever is put there next is
    // already owned (and waited for by ~Queued) if this chunk turns out to have failed.
    auto finishTicket = [&](mcraw_pool_ticket *t, size_t ci) {
        if (!t)
            return;
        const Chunk &c = chunks[ci];
        if (mcraw_pool_ticket_wait(t, written.data() + c.first, status.data() + c.first) != 0)
            throw IOException(std::string("GPU decode failed: ") + mcraw_pool_last_error());
        checkChunk(ci);
    };
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        const Chunk &c = chunks[ci];
        auto t0 = now();
        reading.get();
        tWaitRead += ms(t0, now());
        if (ci + 1 < chunks.size())
            reading = std::async(std::launch::async, readChunk, ci + 1);
        t0 = now();
        if (copying[ci % noutslots].valid())
            copying[ci % noutslots].get(); // chunk ci - noutslots used this output slot
        tWaitCopy += ms(t0, now());
        std::vector<size_t> ofill(G, 0);
        for (size_t k = 0; k < c.count; k++) {
            uint8_t *dst = direct ? (*outBuffers)[c.first + k] : I.pinOut[ci % noutslots][k % G].p + ofill[k % G];
            frames[c.first + k].out = reinterpret_cast<uint16_t *>(dst);
            ofill[k % G] += up(frames[c.first + k].out_capacity * 2);
        }
        t0 = now();
        // the post stage is a property of this call, not of the contexts: set for the batch, cleared behind it
        int rc = post.flags ? mcraw_pool_set_post(I.pool, &post) : 0;
        mcraw_pool_ticket *ticket = nullptr;
        if (rc == 0) // queued BEHIND chunk ci-1, which is waited for below: the PCIe lanes never drain
            rc = mcraw_pool_decode_batch_async(I.pool, frames.data() + c.first, static_cast<int>(c.count), &ticket);
        if (post.flags)
            (void)mcraw_pool_set_post(I.pool, nullptr);
        const auto tSub = now();
        if (rc != 0)
            throw IOException(std::string("GPU decode failed: ") + mcraw_pool_last_error());
        {
            mcraw_pool_ticket *prev = queued.ticket;
            const size_t prevCi = queued.ci;
            queued.ticket = ticket; // owned from here on, whatever chunk ci-1 turns out to be
            queued.ci = ci;
            finishTicket(prev, prevCi);
            if (prev && !direct)
                copying[prevCi % noutslots] = std::async(std::launch::async, copyOut, prevCi);
        }
        tDecode += ms(t0, now());
        if (trace2)
            std::fprintf(stderr, "[mcraw2] chunk %zu (%zu frames) at %.2f ms: batch call %.2f, behind it %.2f\n", ci, c.count, ms(tStart, t0), ms(t0, tSub), ms(tSub, now()));
    }
    {
        const auto t0 = now();
        mcraw_pool_ticket *lastTicket = queued.ticket;
        queued.ticket = nullptr;
        finishTicket(lastTicket, queued.ci);
        tDecode += ms(t0, now());
    }
    const auto t1 = now();
    if (!direct)
        copyOut(queued.ci); // (the last chunk: on this thread)
    for (size_t k = 0; k < copying.size(); k++) { // oldest first
        std::future<void> &f = copying[(chunks.size() + k) % copying.size()];
        if (f.valid())
            f.get();
    }
    if (single && hitRd && A.rd.armed && A.rd.done.valid() && A.rd.done.wait_for(std::chrono::seconds(0)) == std::future_status::ready) {
        // the caller walks the index (this frame had been read ahead) and the next frame's payload is here already: onto the GPU
        // behind this call (the caller works on this frame meanwhile), and the payload of the one behind it on its way into the
        // buffer this call's frame came from.  (A caller that jumps about never gets here: its next call would have to wait
        // for a frame it does not want.)
        if (A.rd.done.get()) {
            const Timestamp t = A.rd.ts;
            const int b = A.rd.buf;
            if (submitAhead(0))
                startRead(t, b ^ 1);
        } else {
            A.rd.armed = false;
        }
    }
    if (trace)
        std::fprintf(stderr, "[mcraw] loadFrames n=%zu chunks=%zu setup %.2f ms (index, JSON, context, pinned staging), pipeline %.2f ms: wait-read %.2f, gpu batch %.2f, wait-copy %.2f, tail copy %.2f\n",
                     n, chunks.size(), ms(tEnter, tStart), ms(tStart, now()), tWaitRead, tDecode, tWaitCopy, ms(t1, now()));
}

} // namespace motioncam
