// facade_probe -- exercises the host side of motioncam::Decoder without decoding a frame on
// the GPU (container parsing, audio, error texts).  Prints one line per observation.
#include <motioncam/Decoder.hpp>

#include <iostream>

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    try {
        motioncam::Decoder d(argv[1]);
        const auto &frames = d.getFrames();
        std::cout << "frames";
        for (auto t : frames)
            std::cout << " " << t;
        std::cout << "\n";
        std::cout << "camera " << d.getContainerMetadata().at("sensorArrangment").get<std::string>() << " "
                  << d.audioSampleRateHz() << " " << d.numAudioChannels() << "\n";
        std::vector<motioncam::AudioChunk> chunks;
        d.loadAudio(chunks);
        for (auto &c : chunks) {
            long sum = 0;
            for (auto v : c.second)
                sum += v;
            std::cout << "audio " << c.first << " " << c.second.size() << " " << sum << "\n";
        }
        motioncam::AudioChunk one;
        int n = 0;
        auto &loader = d.loadAudio();
        while (loader.next(one))
            n++;
        std::cout << "loader " << n << "\n";
        std::vector<uint8_t> data;
        nlohmann::json meta;
        try {
            d.loadFrame(123456789, data, meta);
        } catch (const motioncam::IOException &e) {
            std::cout << "missing: " << e.what() << "\n";
        }
        if (argc > 2 && !frames.empty()) { // decode attempt (fails cleanly where there is no GPU)
            try {
                d.loadFrame(frames[0], data, meta);
                std::cout << "decoded " << data.size() << " " << meta["width"] << "x" << meta["height"] << "\n";
            } catch (const motioncam::IOException &e) {
                std::cout << "decode: " << e.what() << "\n";
            }
        }
    } catch (const motioncam::MotionCamException &e) {
        std::cout << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
