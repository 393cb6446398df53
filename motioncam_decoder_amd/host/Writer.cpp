// Writer.cpp -- the .mcraw container writer (host only; see Writer.hpp).  Records are the packed little-endian PODs of
// mcraw_container.h; every chunk is {kind, bytes} followed by its bytes.
#include <motioncam/Decoder.hpp> // IOException
#include <motioncam/Writer.hpp>
#include <motioncam/mcraw_container.h>

#include <algorithm>
#include <cstring>
#include <limits>

namespace motioncam {

using namespace container;

Writer::Writer(const std::string &path, const nlohmann::json &cameraMetadata, const Options &options)
    : mFile(std::fopen(path.c_str(), "wb")), mPos(0), mOptions(options), mFinished(false)
{
    if (!mFile)
        throw IOException("Failed to open " + path);
    try {
        FileMagic magic{};
        std::memcpy(magic.magic, kMagic, sizeof(kMagic));
        magic.version = kVersion;
        put(&magic, sizeof(magic));
        const std::string text = cameraMetadata.dump();
        putChunk(static_cast<uint32_t>(Kind::JSON), static_cast<uint32_t>(text.size()));
        put(text.data(), text.size());
    } catch (...) { // (no destructor runs for an object whose constructor throws)
        std::fclose(mFile);
        mFile = nullptr;
        throw;
    }
}

Writer::~Writer()
{
    try {
        finish();
    } catch (...) { // (a destructor has nobody to tell)
    }
    if (mFile)
        std::fclose(mFile);
}

void Writer::put(const void *data, size_t size)
{
    if (size && std::fwrite(data, 1, size, mFile) != size) {
        mFinished = true; // a file with a hole gets no index: it must not read as a valid container
        throw IOException("Failed to write data");
    }
    mPos += static_cast<int64_t>(size);
}

void Writer::putChunk(uint32_t kind, uint32_t size)
{
    Chunk c{};
    c.kind = static_cast<Kind>(kind);
    c.bytes = size;
    put(&c, sizeof(c));
}

void Writer::addFrame(int64_t timestamp, const uint8_t *payload, size_t size, const nlohmann::json &frameMetadata)
{
    if (mFinished)
        throw IOException("Writer is finished");
    if (size > std::numeric_limits<uint32_t>::max())
        throw IOException("Frame too large");
    const int64_t at = mPos;
    putChunk(static_cast<uint32_t>(Kind::FRAME), static_cast<uint32_t>(size));
    put(payload, size);
    const std::string text = frameMetadata.dump();
    putChunk(static_cast<uint32_t>(Kind::JSON), static_cast<uint32_t>(text.size()));
    put(text.data(), text.size());
    mFrames.push_back({at, timestamp}); // (the index only lists frames that are in the file whole)
}

void Writer::putAudio(int64_t timestampNs, const int16_t *samples, size_t count)
{
    mAudio.push_back({mPos, timestampNs});
    putChunk(static_cast<uint32_t>(Kind::PCM), static_cast<uint32_t>(count * sizeof(int16_t)));
    put(samples, count * sizeof(int16_t));
    if (timestampNs >= 0) { // newer files follow the samples with their capture time (lib/Decoder.cpp:60-72)
        PcmTime t{};
        t.timeNs = timestampNs;
        putChunk(static_cast<uint32_t>(Kind::PCM_TIME), sizeof(t));
        put(&t, sizeof(t));
    }
}

void Writer::addAudio(int64_t timestampNs, const int16_t *samples, size_t count)
{
    if (mFinished)
        throw IOException("Writer is finished");
    if (count * sizeof(int16_t) > std::numeric_limits<uint32_t>::max())
        throw IOException("Audio chunk too large");
    if (mOptions.audioBehindFrames)
        mPendingAudio.push_back({timestampNs, std::vector<int16_t>(samples, samples + count)});
    else
        putAudio(timestampNs, samples, count);
}

void Writer::finish()
{
    if (mFinished || !mFile)
        return;
    mFinished = true;
    for (const Pending &p : mPendingAudio)
        putAudio(p.time, p.samples.data(), p.samples.size());
    mPendingAudio.clear();
    if (mOptions.audioIndex && !mAudio.empty()) {
        PcmTable table{};
        table.rows = static_cast<int64_t>(mAudio.size());
        table.firstTimeMs = 0;
        putChunk(static_cast<uint32_t>(Kind::PCM_TABLE), static_cast<uint32_t>(sizeof(table) + mAudio.size() * sizeof(Locator)));
        put(&table, sizeof(table));
        for (const Row &r : mAudio) {
            Locator l{};
            l.position = r.position;
            l.time = r.time;
            put(&l, sizeof(l));
        }
    }
    std::vector<Row> rows = mFrames;
    if (!mOptions.indexInArrivalOrder)
        std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.time < b.time; });
    putChunk(static_cast<uint32_t>(Kind::FRAME_TABLE_ROWS), static_cast<uint32_t>(rows.size() * sizeof(Locator)));
    const int64_t rowsPosition = mPos;
    for (const Row &r : rows) {
        Locator l{};
        l.position = r.position;
        l.time = r.time;
        put(&l, sizeof(l));
    }
    if (rows.size() > static_cast<size_t>(std::numeric_limits<int32_t>::max()))
        throw IOException("Too many frames");
    FrameTable table{};
    table.magic = static_cast<int32_t>(kFrameTableMagic);
    table.rows = static_cast<int32_t>(rows.size());
    table.rowsPosition = rowsPosition;
    putChunk(static_cast<uint32_t>(Kind::FRAME_TABLE), sizeof(table));
    put(&table, sizeof(table));
    FILE *f = mFile;
    mFile = nullptr;
    if (std::fclose(f) != 0) // (what the C library still held goes to the file here: a full disk shows now)
        throw IOException("Failed to write data");
}

} // namespace motioncam
