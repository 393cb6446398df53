// mcraw_container.h -- the .mcraw container as this build reads it (host side only).
//
// Own description of the on-disk layout (what the bytes are); all
// records are packed little-endian PODs.
//
//   file := FileMagic, Chunk{JSON} camera-json,
//           { Chunk{FRAME} frame-bytes, Chunk{JSON} frame-json }*,
//           { Chunk{PCM} int16 samples [, Chunk{PCM_TIME} PcmTime] }*,
//           [ Chunk{PCM_TABLE} PcmTable Locator[n] ],
//           Chunk{FRAME_TABLE_ROWS} Locator[n],
//           Chunk{FRAME_TABLE} FrameTable                       <- the last 24 bytes of the file
#ifndef MCRAW_CONTAINER_H
#define MCRAW_CONTAINER_H

#include <cstdint>

namespace motioncam {
namespace container {

constexpr uint32_t kFrameTableMagic = 0x8A905612u;
constexpr uint8_t kVersion = 3;
constexpr char kMagic[7] = {'M', 'O', 'T', 'I', 'O', 'N', ' '};

// What a chunk holds (the u32 in front of every chunk).  Every value has two spellings: this build's, and
// the one code written against the reference header uses (lib/include/motioncam/Container.hpp:38-46).
enum class Kind : uint32_t {
    FRAME_TABLE = 0,      // FrameTable, at the very end
    FRAME_TABLE_ROWS = 1, // Locator[] of the frames
    FRAME = 2,            // one compressed frame buffer
    JSON = 3,             // camera metadata (once) or per-frame metadata (after each FRAME)
    PCM_TABLE = 4,        // PcmTable + Locator[] of the audio chunks
    PCM = 5,              // interleaved int16 samples
    PCM_TIME = 6,         // capture time of the PCM chunk before it
    BUFFER_INDEX = FRAME_TABLE,
    BUFFER_INDEX_DATA = FRAME_TABLE_ROWS,
    BUFFER = FRAME,
    METADATA = JSON,
    AUDIO_INDEX = PCM_TABLE,
    AUDIO_DATA = PCM,
    AUDIO_DATA_METADATA = PCM_TIME
};

// The records below carry every field under both names as well (anonymous unions of one type: the two
// names are the same bytes), so that sources written against either header compile against these.
struct FileMagic {
    union {
        uint8_t magic[7];
        uint8_t ident[7];
    };
    uint8_t version;
};

struct Chunk {
    union {
        Kind kind;
        Kind type;
    };
    union {
        uint32_t bytes;
        uint32_t size;
    };
};

struct Locator {
    union {
        int64_t position; // file offset of the chunk header
        int64_t offset;
    };
    union {
        int64_t time; // timestamp (ns) the entry is addressed by
        int64_t timestamp;
    };
};

struct FrameTable {
    union {
        int32_t magic;
        int32_t magicNumber;
    };
    union {
        int32_t rows;
        int32_t numOffsets;
    };
    union {
        int64_t rowsPosition; // file offset of Locator[0]
        int64_t indexDataOffset;
    };
};

struct PcmTable {
    union {
        int64_t rows;
        int64_t numOffsets;
    };
    union {
        int64_t firstTimeMs;
        int64_t startTimestampMs;
    };
};

struct PcmTime {
    union {
        int64_t timeNs;
        int64_t timestampNs;
    };
};

static_assert(sizeof(FileMagic) == 8 && sizeof(Chunk) == 8 && sizeof(Locator) == 16 && sizeof(FrameTable) == 16 &&
                  sizeof(PcmTable) == 16 && sizeof(PcmTime) == 8,
              "container records are packed");

} // namespace container
} // namespace motioncam

#endif
