// Forwarding header: the public motioncam::Decoder API carries nlohmann::json in its
// signatures (reference lib/include/motioncam/Decoder.hpp:55,61).  nlohmann/json is a
// third-party MIT library that this repository does not vendor: use the copy the build
// environment provides (a later <nlohmann/json.hpp> on the include path, or the
// single-header copy shipped with the image's conda).
#pragma once
#if defined(__has_include_next)
#  if __has_include_next(<nlohmann/json.hpp>)
#    include_next <nlohmann/json.hpp>
#    define MCRAW_HAVE_NLOHMANN 1
#  endif
#endif
#ifndef MCRAW_HAVE_NLOHMANN
#  if __has_include("/opt/conda/include/json.hpp")
#    include "/opt/conda/include/json.hpp"
#  else
#    error "nlohmann/json.hpp not found: add its directory to the include path"
#  endif
#endif
