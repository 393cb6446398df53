// Container.hpp -- source compatibility with the reference header of the same name
// (lib/include/motioncam/Container.hpp:22-72): its type names, member names, enumerators and constants,
// all mapped onto this build's own description of the container (mcraw_container.h), whose records carry
// every field and enumerator under both spellings.  tests/cpp/container_compat.cpp is a translation unit
// written purely in the reference's vocabulary; it is compiled against this header by the CPU tests.
#ifndef MCRAW_HOST_CONTAINER_COMPAT_HPP
#define MCRAW_HOST_CONTAINER_COMPAT_HPP

#include <motioncam/mcraw_container.h>

namespace motioncam {

using Header = container::FileMagic;        // ident[7], version
using Item = container::Chunk;              // type, size
using Type = container::Kind;               // BUFFER_INDEX .. AUDIO_DATA_METADATA
using BufferOffset = container::Locator;    // offset, timestamp
using BufferIndex = container::FrameTable;  // magicNumber, numOffsets, indexDataOffset
using AudioIndex = container::PcmTable;     // numOffsets, startTimestampMs
using AudioMetadata = container::PcmTime;   // timestampNs

constexpr uint32_t INDEX_MAGIC_NUMBER = container::kFrameTableMagic;
constexpr uint8_t CONTAINER_VERSION = container::kVersion;
constexpr uint8_t CONTAINER_ID[7] = {'M', 'O', 'T', 'I', 'O', 'N', ' '};

enum VideoType { VIDEO, TIMELAPSE }; // how a clip was recorded (camera metadata; not used by the decoder)

} // namespace motioncam

#endif
