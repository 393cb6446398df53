// Forwarding header: the public motioncam::Decoder API carries nlohmann::json in its signatures (reference
// lib/include/motioncam/Decoder.hpp:55,61).  nlohmann/json is a third-party MIT library that this repository does not carry.
// Looked for, in this order:
//   1. -DMCRAW_NLOHMANN_JSON_HPP='"/path/to/json.hpp"'                          (a packager's choice)
//   2. _vendored/json.hpp next to this file: what `tools/fetch_nlohmann_json.sh` puts there (release v3.11.3, the version the
//      reference ships in thirdparty/; git-ignored)
//   3. a later <nlohmann/json.hpp> on the include path                         (distribution package `nlohmann-json3-dev`,
//      or the reference's own thirdparty/ directory when the facade is built into the reference's tree: INTEGRATION.md §3)
//   4. the single-header copy this image's conda provides (3.1.1: enough for the facade, which uses parse / dump / operator[] /
//      get<> / contains-by-find only).
// A translation unit must see ONE version of the library (its namespace is version-tagged from 3.11 on): a program that
// includes the reference's copy first -- the reference's example.cpp does -- gets that one, through its include guard.
#pragma once
// (a helper macro first: `defined(X) && X(...)` in one #if is ill-formed where the preprocessor does not know X -- the identifier
// becomes 0 and `0(<...>)` does not parse, whatever defined() said)
#if defined(__has_include_next)
#  if __has_include_next(<nlohmann/json.hpp>)
#    define MCRAW_JSON_NEXT_ON_PATH 1
#  endif
#endif
#if defined(MCRAW_NLOHMANN_JSON_HPP)
#  include MCRAW_NLOHMANN_JSON_HPP
#elif __has_include("_vendored/json.hpp")
#  include "_vendored/json.hpp"
#elif defined(MCRAW_JSON_NEXT_ON_PATH)
#  include_next <nlohmann/json.hpp>
#elif __has_include("/usr/include/nlohmann/json.hpp")
#  include "/usr/include/nlohmann/json.hpp"
#elif __has_include("/opt/conda/include/nlohmann/json.hpp")
#  include "/opt/conda/include/nlohmann/json.hpp"
#elif __has_include("/opt/conda/include/json.hpp")
#  include "/opt/conda/include/json.hpp"
#else
#  error "nlohmann/json.hpp (3.1 or later; the reference ships 3.11.3) not found: install nlohmann-json3-dev, run tools/fetch_nlohmann_json.sh, or pass -DMCRAW_NLOHMANN_JSON_HPP='\"/path/json.hpp\"'"
#endif
#if !defined(NLOHMANN_JSON_VERSION_MAJOR) || NLOHMANN_JSON_VERSION_MAJOR < 3 || (NLOHMANN_JSON_VERSION_MAJOR == 3 && NLOHMANN_JSON_VERSION_MINOR < 1)
#  error "nlohmann/json older than 3.1: the facade needs json::parse(string), find(), get<T>() and dump()"
#endif
