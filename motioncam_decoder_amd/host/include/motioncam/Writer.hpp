// Writer.hpp -- motioncam::Writer: writes the .mcraw container that motioncam::Decoder (this build's and the
// reference's, lib/Decoder.cpp:97-319) reads.  The reference has no writer (its files come from the recording app;
// lib/include/motioncam/Container.hpp:22-72 only describes the records); this one is the build's own, the counterpart
// of the reader in Decoder.cpp: it appends frames as they are handed over and leaves the two indices behind them.
//
//   Writer w("out.mcraw", cameraJson);
//   w.addFrame(timestampNs, payload, size, frameJson);     // compressed frame buffer + its metadata, any order
//   w.addAudio(timestampNs, samples, count);               // interleaved int16 PCM; timestamp < 0: chunk without one
//   w.finish();                                            // audio index, frame index, trailer
//
// Layout written: header, camera JSON, {BUFFER, METADATA}*, {AUDIO_DATA [AUDIO_DATA_METADATA]}*,
// [AUDIO_INDEX], BUFFER_INDEX_DATA, BUFFER_INDEX.  The options reproduce habits of real recorders that a reader has
// to cope with (tests/test_host_writer.py runs the reference against every one of them).
#ifndef MCRAW_HOST_WRITER_HPP
#define MCRAW_HOST_WRITER_HPP

#include <nlohmann/json.hpp>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace motioncam {

class Writer {
public:
    struct Options {
        // audio chunks are written when finish() is called, behind the frames (true: the order the reader's walk from
        // the last frame expects, lib/Decoder.cpp:281-315) or in the order they were added between the frames (false)
        bool audioBehindFrames;
        // write the audio index at all (a file without one has no audio for either reader)
        bool audioIndex;
        // the frame index lists its rows in the order the frames were added (true) or sorted by timestamp (false);
        // readers sort by timestamp (lib/Decoder.cpp:266-279)
        bool indexInArrivalOrder;
        Options() : audioBehindFrames(true), audioIndex(true), indexInArrivalOrder(true) {}
    };

    Writer(const std::string &path, const nlohmann::json &cameraMetadata, const Options &options = Options());
    ~Writer(); // finishes the file if finish() was not called

    Writer(const Writer &) = delete;
    Writer &operator=(const Writer &) = delete;

    void addFrame(int64_t timestamp, const uint8_t *payload, size_t size, const nlohmann::json &frameMetadata);
    void addAudio(int64_t timestampNs, const int16_t *samples, size_t count);
    void finish();

    size_t frameCount() const { return mFrames.size(); }

private:
    struct Row {
        int64_t position, time;
    };
    struct Pending {
        int64_t time;
        std::vector<int16_t> samples;
    };
    void put(const void *data, size_t size);
    void putChunk(uint32_t kind, uint32_t size);
    void putAudio(int64_t timestampNs, const int16_t *samples, size_t count);

    FILE *mFile;
    int64_t mPos;
    Options mOptions;
    std::vector<Row> mFrames, mAudio;
    std::vector<Pending> mPendingAudio;
    bool mFinished;
};

} // namespace motioncam

#endif
