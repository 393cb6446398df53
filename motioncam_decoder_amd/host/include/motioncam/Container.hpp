// Container.hpp -- compatibility names.  Code written against the reference
// (lib/include/motioncam/Container.hpp) may mention its container records; they are
// aliases of this build's own definitions in mcraw_container.h.
#ifndef MCRAW_HOST_CONTAINER_COMPAT_HPP
#define MCRAW_HOST_CONTAINER_COMPAT_HPP

#include <motioncam/mcraw_container.h>

namespace motioncam {

using Header = container::FileMagic;
using Item = container::Chunk;
using Type = container::Kind;
using BufferOffset = container::Locator;
using BufferIndex = container::FrameTable;
using AudioIndex = container::PcmTable;
using AudioMetadata = container::PcmTime;

constexpr uint32_t INDEX_MAGIC_NUMBER = container::kFrameTableMagic;
constexpr uint8_t CONTAINER_VERSION = container::kVersion;

} // namespace motioncam

#endif
