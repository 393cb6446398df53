// RawData.cpp -- motioncam::raw::Decode / DecodeLegacy over the C ABI of the HIP library.
#include <motioncam/RawData.hpp>

#include "mcraw_hip.h"

namespace motioncam {
namespace raw {

size_t Decode(uint16_t *output, const int width, const int height, const uint8_t *input, const size_t len)
{
    return mcraw_decode7(output, width, height, input, len);
}

size_t DecodeLegacy(uint16_t *output, const int width, const int height, const uint8_t *input, const size_t len)
{
    return mcraw_decode6(output, width, height, input, len);
}

} // namespace raw
} // namespace motioncam
