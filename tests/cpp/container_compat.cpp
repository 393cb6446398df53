// A translation unit in the vocabulary of the reference's Container.hpp (type, member, enumerator and
// constant names of lib/include/motioncam/Container.hpp:22-72), compiled against THIS repository's header
// of the same name: walks the items of a .mcraw file the way a reader written for the reference would.
//   container_compat <file.mcraw>   -> prints "frames N audio M first_ts T"
#include <motioncam/Container.hpp>

#include <cstdio>
#include <cstring>
#include <vector>

using namespace motioncam;

int main(int argc, char **argv)
{
    static_assert(sizeof(Header) == 8 && sizeof(Item) == 8 && sizeof(BufferOffset) == 16 && sizeof(BufferIndex) == 16 &&
                      sizeof(AudioIndex) == 16 && sizeof(AudioMetadata) == 8,
                  "record sizes of the container");
    static_assert(static_cast<uint32_t>(Type::BUFFER_INDEX) == 0 && static_cast<uint32_t>(Type::BUFFER_INDEX_DATA) == 1 &&
                      static_cast<uint32_t>(Type::BUFFER) == 2 && static_cast<uint32_t>(Type::METADATA) == 3 &&
                      static_cast<uint32_t>(Type::AUDIO_INDEX) == 4 && static_cast<uint32_t>(Type::AUDIO_DATA) == 5 &&
                      static_cast<uint32_t>(Type::AUDIO_DATA_METADATA) == 6,
                  "item types");
    VideoType vt = VIDEO;
    if (vt == TIMELAPSE || argc < 2)
        return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f)
        return 3;
    Header header;
    if (std::fread(&header, sizeof(Header), 1, f) != 1 || header.version != CONTAINER_VERSION ||
        std::memcmp(header.ident, CONTAINER_ID, sizeof(CONTAINER_ID)) != 0)
        return 4;
    BufferIndex index;
    std::fseek(f, -static_cast<long>(sizeof(BufferIndex)), SEEK_END);
    if (std::fread(&index, sizeof(BufferIndex), 1, f) != 1 || static_cast<uint32_t>(index.magicNumber) != INDEX_MAGIC_NUMBER)
        return 5;
    std::vector<BufferOffset> offsets(static_cast<size_t>(index.numOffsets));
    std::fseek(f, static_cast<long>(index.indexDataOffset), SEEK_SET);
    if (std::fread(offsets.data(), sizeof(BufferOffset), offsets.size(), f) != offsets.size())
        return 6;
    long frames = 0, audio = 0;
    int64_t first = -1;
    for (const BufferOffset &o : offsets) {
        Item item;
        std::fseek(f, static_cast<long>(o.offset), SEEK_SET);
        if (std::fread(&item, sizeof(Item), 1, f) != 1 || item.type != Type::BUFFER || item.size == 0)
            return 7;
        frames++;
        if (first < 0 || o.timestamp < first)
            first = o.timestamp;
    }
    // items behind the last frame: audio data, its timestamps, the audio index
    std::fseek(f, static_cast<long>(sizeof(Header)), SEEK_SET);
    for (;;) {
        Item item;
        if (std::fread(&item, sizeof(Item), 1, f) != 1)
            break;
        if (item.type == Type::AUDIO_DATA)
            audio++;
        if (item.type == Type::AUDIO_DATA_METADATA) {
            AudioMetadata m;
            if (std::fread(&m, sizeof(m), 1, f) != 1 || m.timestampNs < -1)
                return 8;
            continue;
        }
        if (item.type == Type::AUDIO_INDEX) {
            AudioIndex ai;
            if (std::fread(&ai, sizeof(ai), 1, f) != 1 || ai.numOffsets != audio || ai.startTimestampMs < -1)
                return 9;
            std::fseek(f, static_cast<long>(item.size - sizeof(ai)), SEEK_CUR);
            continue;
        }
        if (item.type == Type::BUFFER_INDEX)
            break;
        std::fseek(f, static_cast<long>(item.size), SEEK_CUR);
    }
    std::fclose(f);
    std::printf("frames %ld audio %ld first_ts %lld\n", frames, audio, static_cast<long long>(first));
    return 0;
}
