// frame_order <file.mcraw> <i|b|oN> ... -- loadFrame() calls in the order the arguments give (each a position in the sorted
// frame list), every one into the SAME vector as the reference's example does; "b" runs a loadFrames() batch of every frame
// in between.  Prints "<position> <bytes> <crc32>" per call: whatever the order, a call's bytes must be its frame's
// (the facade reads the index-next frame ahead of the call that may never come).
#include <motioncam/Decoder.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint32_t crc32(const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    if (!table[1])
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++)
                c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++)
        c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

int main(int argc, char **argv)
{
    if (argc < 3)
        return 2;
    try {
        motioncam::Decoder d(argv[1]);
        const std::vector<motioncam::Timestamp> frames = d.getFrames();
        std::vector<uint8_t> data;
        nlohmann::json meta;
        motioncam::Decoder::FrameOutput output;
        for (int a = 2; a < argc; a++) {
            if (!std::strcmp(argv[a], "b")) {
                std::vector<std::vector<uint8_t>> all;
                std::vector<nlohmann::json> metas;
                d.loadFrames(frames, all, metas);
                for (size_t i = 0; i < all.size(); i++)
                    std::printf("b%zu %zu %08x\n", i, all[i].size(), crc32(all[i].data(), all[i].size()));
                continue;
            }
            if (argv[a][0] == 'o') { // "o12": the calls behind it ask for 12-bit strip rows (a one-frame loadFrames); "o16": plain again
                output.bitsPerSample = std::atoi(argv[a] + 1);
                continue;
            }
            const size_t i = static_cast<size_t>(std::atol(argv[a]));
            if (i >= frames.size())
                return 2;
            try {
                if (output.bitsPerSample == 16) {
                    d.loadFrame(frames[i], data, meta);
                } else {
                    std::vector<std::vector<uint8_t>> one;
                    std::vector<nlohmann::json> metas;
                    d.loadFrames({frames[i]}, one, metas, output);
                    data = one[0];
                    meta = metas[0];
                }
            } catch (const motioncam::IOException &e) { // (a frame that does not decode: said, and the walk goes on)
                std::printf("%zu failed: %s\n", i, e.what());
                continue;
            }
            const int w = meta["width"], h = meta["height"];
            std::printf("%zu %zu %08x %dx%d\n", i, data.size(), crc32(data.data(), data.size()), w, h);
        }
    } catch (const motioncam::MotionCamException &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
