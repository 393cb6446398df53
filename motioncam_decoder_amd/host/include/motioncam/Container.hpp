// Container.hpp -- on-disk records of the .mcraw container (host side only).
//
// Own restatement of the file format read by the reference
// (lib/include/motioncam/Container.hpp:22-72, lib/Decoder.cpp:116-151,237-315); the
// names match the reference so code written against it keeps compiling.  All records
// are packed native-endian (little-endian) PODs.
//
//   file := Header, Item{METADATA} camera-json,
//           { Item{BUFFER} frame-bytes, Item{METADATA} frame-json }*,
//           { Item{AUDIO_DATA} pcm16 [, Item{AUDIO_DATA_METADATA} AudioMetadata] }*,
//           [ Item{AUDIO_INDEX} AudioIndex BufferOffset[n] ],
//           Item{BUFFER_INDEX_DATA} BufferOffset[n],
//           Item{BUFFER_INDEX} BufferIndex                     <- last 24 bytes
#ifndef MCRAW_HOST_CONTAINER_HPP
#define MCRAW_HOST_CONTAINER_HPP

#include <cstdint>

namespace motioncam {

const uint32_t INDEX_MAGIC_NUMBER = 0x8A905612;
const uint8_t CONTAINER_VERSION = 3;
const uint8_t CONTAINER_ID[7] = {'M', 'O', 'T', 'I', 'O', 'N', ' '};

struct Header {
    uint8_t ident[7];
    uint8_t version;
};

enum VideoType { VIDEO, TIMELAPSE };

enum class Type : uint32_t {
    BUFFER_INDEX = 0,
    BUFFER_INDEX_DATA = 1,
    BUFFER = 2,
    METADATA = 3,
    AUDIO_INDEX = 4,
    AUDIO_DATA = 5,
    AUDIO_DATA_METADATA = 6
};

struct Item {
    Type type;
    uint32_t size;
};

struct BufferOffset {
    int64_t offset;
    int64_t timestamp;
};

struct BufferIndex {
    int32_t magicNumber;
    int32_t numOffsets;
    int64_t indexDataOffset;
};

struct AudioIndex {
    int64_t numOffsets;
    int64_t startTimestampMs;
};

struct AudioMetadata {
    int64_t timestampNs;
};

static_assert(sizeof(Header) == 8 && sizeof(Item) == 8 && sizeof(BufferOffset) == 16 && sizeof(BufferIndex) == 16 &&
                  sizeof(AudioIndex) == 16 && sizeof(AudioMetadata) == 8,
              "container records are packed");

} // namespace motioncam

#endif
