// Decoder.hpp -- motioncam::Decoder with the public interface of the reference
// (lib/include/motioncam/Decoder.hpp:28-74), so that example.cpp and other callers build
// unchanged, implemented from scratch on top of the MI355X decode path:
//
//   * the container is indexed once at construction and read with positional reads (pread),
//     not with the reference's seek+read on a shared FILE position;
//   * loadFrame() decodes on the GPU through mcraw_decode7 / mcraw_decode6;
//   * loadFrames() (an addition) reads a whole set of frames into pinned memory and decodes
//     them as ONE batch (mcraw_decode_batch, host-memory mode: H2D copies of a sub-batch overlap
//     the decode of the previous one) -- the replacement for the per-frame loop of
//     example.cpp:187-195.
#ifndef MCRAW_HOST_DECODER_HPP
#define MCRAW_HOST_DECODER_HPP

#include <motioncam/Container.hpp>
#include <nlohmann/json.hpp>

#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace motioncam {

typedef int64_t Timestamp;
typedef std::pair<Timestamp, std::vector<int16_t>> AudioChunk;

class MotionCamException : public std::runtime_error {
public:
    MotionCamException(const std::string &error) : runtime_error(error) {}
};

class IOException : public MotionCamException {
public:
    IOException(const std::string &error) : MotionCamException(error) {}
};

class AudioChunkLoader {
public:
    virtual bool next(AudioChunk &output) = 0;
    virtual ~AudioChunkLoader() = default;
};

class Decoder {
public:
    Decoder(const std::string &path);
    Decoder(FILE *file); // takes ownership, like the reference (closed by the destructor)
    ~Decoder();

    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;

    // Container (camera) metadata.
    const nlohmann::json &getContainerMetadata() const;

    // Timestamps of all frames, ascending.
    const std::vector<Timestamp> &getFrames() const;

    // Decode one frame: outData becomes width*height uint16 LE (row-major Bayer mosaic).
    void loadFrame(const Timestamp timestamp, std::vector<uint8_t> &outData, nlohmann::json &outMetadata);

    // Decode many frames as one GPU batch (addition to the reference API).
    void loadFrames(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> &outData,
                    std::vector<nlohmann::json> &outMetadata);

    // What loadFrames hands back per frame.  The defaults are loadFrame()'s output (uint16 LE mosaic,
    // 2 * width * height bytes).  The other forms are produced by a stage fused into the GPU decode
    // (mcraw_ctx_set_post) for the consumer example.cpp:55-139 stands for, a DNG writer:
    //   subtractBlackLevel  sample = max(sample - blackLevel[(row & 1) * 2 + (col & 1)], 0), levels from the
    //                       container metadata ("blackLevel", example.cpp:65) -- write BlackLevel 0 then;
    //   bitsPerSample = 10 / 12 / 14   rows as strips of that many bits per sample, ceil(width * bits / 8) bytes
    //                       each, MSB-first (SetBitsPerSample, example.cpp:116-117): a quarter (12) to three
    //                       eighths (10) less to copy and store.  bitsForWhiteLevel() picks the narrowest form
    //                       that holds the container's "whiteLevel" (example.cpp:66, :91).
    struct FrameOutput {
        bool subtractBlackLevel;
        int bitsPerSample; // 16, 14, 12 or 10
        FrameOutput() : subtractBlackLevel(false), bitsPerSample(16) {}
    };
    static int bitsForWhiteLevel(double whiteLevel);
    void loadFrames(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> &outData,
                    std::vector<nlohmann::json> &outMetadata, const FrameOutput &output);

    // The GPUs loadFrames / loadFramesInto shard their batches over: frame i of a batch is decoded by device
    // devices[i mod size], each device fed by its own host thread (bound to the GPU's NUMA node) from pinned
    // staging on that node; the result does not depend on the set.  Default (never called, or an empty list):
    // the environment -- MCRAW_DEVICES ("all" or "0,1,5"), else MCRAW_DEVICE, else the current HIP device.
    // This is the many-GPU form of the loop the reference runs frame by frame (example.cpp:187-195).
    void useDevices(const std::vector<int> &devices);
    int deviceCount(); // members of the pool in use (creates it)

    // JSON of one frame without decoding it (width, height, compressionType ...: what sizes a buffer).
    void loadFrameMetadata(const Timestamp timestamp, nlohmann::json &outMetadata);

    // The compressed frame buffer as it sits in the file (the BUFFER item, lib/Decoder.cpp:193-206), and its JSON:
    // what a remux / trim tool hands to motioncam::Writer (Writer.hpp).  Host only, no GPU involved.
    void loadFramePayload(const Timestamp timestamp, std::vector<uint8_t> &outPayload, nlohmann::json &outMetadata);

    // The same batch into memory the caller owns: frame i is written to outBuffers[i], which must hold
    // frameBytes(width, height, output) bytes.  The GPU pipeline downloads straight into these buffers (no
    // staging copy, no std::vector to fault in), so pinned memory is the fast choice: mcraw_host_alloc() /
    // hipHostMalloc -- pageable memory works, through the HIP runtime's own staging.
    void loadFramesInto(const std::vector<Timestamp> &timestamps, const std::vector<uint8_t *> &outBuffers,
                        std::vector<nlohmann::json> &outMetadata, const FrameOutput &output = FrameOutput());
    static size_t frameBytes(int width, int height, const FrameOutput &output = FrameOutput());

    int audioSampleRateHz() const;
    int numAudioChannels() const;

    // All audio chunks at once / one at a time.
    void loadAudio(std::vector<AudioChunk> &outAudioChunks);
    AudioChunkLoader &loadAudio() const;

private:
    void loadFramesImpl(const std::vector<Timestamp> &timestamps, std::vector<std::vector<uint8_t>> *outData,
                        const std::vector<uint8_t *> *outBuffers, std::vector<nlohmann::json> &outMetadata,
                        const FrameOutput &output);
    struct Impl;
    std::unique_ptr<Impl> mImpl;
};

} // namespace motioncam

#endif
