#!/bin/bash
# Puts the single-header release of nlohmann/json that the reference ships (v3.11.3, MIT) where the facade's forwarding header
# looks first.  Needs network access: not for the build containers of this repository, which use the image's own copy.
set -eu
VER=${1:-v3.11.3}
DST="$(cd "$(dirname "$0")/.." && pwd)/motioncam_decoder_amd/host/thirdparty/nlohmann/_vendored"
mkdir -p "$DST"
curl -fL --retry 3 -o "$DST/json.hpp.part" "https://github.com/nlohmann/json/releases/download/$VER/json.hpp"
grep -q "NLOHMANN_JSON_VERSION_MAJOR" "$DST/json.hpp.part" || { echo "not a nlohmann/json header" >&2; exit 1; }
mv "$DST/json.hpp.part" "$DST/json.hpp"
grep -m3 "define NLOHMANN_JSON_VERSION_" "$DST/json.hpp"
sha256sum "$DST/json.hpp"
