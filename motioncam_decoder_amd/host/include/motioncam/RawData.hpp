// RawData.hpp -- the codec seam of the reference (lib/include/motioncam/RawData.hpp:25-37),
// kept source-compatible.  Both functions forward to the MI355X decode path through the C ABI
// (include/mcraw_hip.h: mcraw_decode7 / mcraw_decode6); there is no CPU implementation.
#ifndef MCRAW_HOST_RAWDATA_HPP
#define MCRAW_HOST_RAWDATA_HPP

#include <cstddef>
#include <cstdint>

namespace motioncam {
namespace raw {

// Current encoding (compressionType 7).  Returns the number of uint16 written
// (width * encodedHeight for well-formed input, like lib/RawData.cpp:611), 0 on failure.
// Unlike the reference, never writes past width * height elements.
size_t Decode(uint16_t *output, const int width, const int height, const uint8_t *input, const size_t len);

// Legacy encoding (compressionType 6).  Returns width * height, 0 on failure.
size_t DecodeLegacy(uint16_t *output, const int width, const int height, const uint8_t *input, const size_t len);

} // namespace raw
} // namespace motioncam

#endif
