// mcraw_export -- decode a .mcraw file on the GPU and dump what it holds.
//
//   mcraw_export <file.mcraw> [-n frames] [-o outdir] [--single [--reuse]] [--no-write] [--black] [--bits 10|12|14|auto] [--pinned]
//   mcraw_export <file.mcraw> --remux <out.mcraw> [-n frames] [--sorted-index] [--audio-inline] [--no-audio-index]
//
// --remux copies the first N frames (compressed as they are), their metadata and the audio into a new container
// written by motioncam::Writer (host only, no GPU): a trim / repair tool, and the round trip of the build's own writer.
//
// Writes outdir/frame_%06d.u16 (width*height uint16 LE, row-major Bayer mosaic) for the
// first N frames (by timestamp) and outdir/audio.s16 (interleaved PCM), and prints one line
// per frame with its geometry and a CRC-32 of the pixels.  Frames are decoded as one GPU
// batch (Decoder::loadFrames); --single uses the per-frame loadFrame() path instead.
// --single --reuse is the reference example's loop as it stands (one `data` vector for every frame, example.cpp:182-188):
// a timing mode, prints the rate and the last frame's checksum only.
// --black subtracts the container's black levels, --bits N writes frame_%06d.pN (N-bit strip rows; auto: the
// narrowest form that holds the container's whiteLevel) instead: both are done by the stage fused into the GPU
// decode (Decoder::FrameOutput).
// --pinned decodes into pinned buffers this tool allocates (Decoder::loadFramesInto: no copy-out stage).
#include <motioncam/Decoder.hpp>
#include <motioncam/Writer.hpp>

#include "mcraw_hip.h" // mcraw_host_alloc / mcraw_host_free for --pinned

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

static uint32_t crc32(const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++)
                c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++)
        c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static bool writeFile(const std::string &path, const void *data, size_t size)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f)
        return false;
    const bool ok = size == 0 || std::fwrite(data, 1, size, f) == size;
    std::fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "Usage: mcraw_export <input file> [-n frames] [-o outdir] [--single]" << std::endl;
        return 2;
    }
    std::string input = argv[1], outdir = ".", remux;
    motioncam::Writer::Options wopt;
    long limit = -1;
    bool single = false, nowrite = false, pinned = false, reuse = false;
    motioncam::Decoder::FrameOutput output;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "-n") && i + 1 < argc)
            limit = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "-o") && i + 1 < argc)
            outdir = argv[++i];
        else if (!std::strcmp(argv[i], "--remux") && i + 1 < argc)
            remux = argv[++i];
        else if (!std::strcmp(argv[i], "--sorted-index"))
            wopt.indexInArrivalOrder = false;
        else if (!std::strcmp(argv[i], "--audio-inline"))
            wopt.audioBehindFrames = false;
        else if (!std::strcmp(argv[i], "--no-audio-index"))
            wopt.audioIndex = false;
        else if (!std::strcmp(argv[i], "--single"))
            single = true;
        else if (!std::strcmp(argv[i], "--reuse"))
            reuse = true;
        else if (!std::strcmp(argv[i], "--no-write"))
            nowrite = true; // decode and checksum only (timing runs)
        else if (!std::strcmp(argv[i], "--pinned"))
            pinned = true;
        else if (!std::strcmp(argv[i], "--black"))
            output.subtractBlackLevel = true;
        else if (!std::strcmp(argv[i], "--bits") && i + 1 < argc)
            output.bitsPerSample = !std::strcmp(argv[i + 1], "auto") ? -1 : std::atoi(argv[i + 1]), i++;
    }
    try {
        motioncam::Decoder decoder(input);
        if (output.bitsPerSample < 0) { // --bits auto: the narrowest strip form that holds the container's white level
            const nlohmann::json &cm = decoder.getContainerMetadata();
            const auto wl = cm.find("whiteLevel");
            output.bitsPerSample = (cm.is_object() && wl != cm.end() && wl->is_number())
                                       ? motioncam::Decoder::bitsForWhiteLevel(wl->get<double>()) : 16;
            std::cout << "bits per sample: " << output.bitsPerSample << std::endl;
        }
        std::vector<motioncam::Timestamp> frames = decoder.getFrames();
        std::cout << "Found " << frames.size() << " frames" << std::endl;
        if (limit >= 0 && static_cast<size_t>(limit) < frames.size())
            frames.resize(static_cast<size_t>(limit));

        if (!remux.empty()) {
            std::vector<motioncam::AudioChunk> chunks;
            decoder.loadAudio(chunks);
            motioncam::Writer writer(remux, decoder.getContainerMetadata(), wopt);
            std::vector<uint8_t> payload;
            nlohmann::json m;
            size_t bytes = 0, a = 0;
            for (size_t i = 0; i < frames.size(); i++) {
                decoder.loadFramePayload(frames[i], payload, m);
                writer.addFrame(frames[i], payload.data(), payload.size(), m);
                bytes += payload.size();
                // (--audio-inline: the chunks go between the frames, a few behind each)
                for (; !wopt.audioBehindFrames && a < chunks.size() && a * frames.size() < (i + 1) * chunks.size(); a++)
                    writer.addAudio(chunks[a].first, chunks[a].second.data(), chunks[a].second.size());
            }
            for (; a < chunks.size(); a++)
                writer.addAudio(chunks[a].first, chunks[a].second.data(), chunks[a].second.size());
            writer.finish();
            std::cout << "remuxed " << frames.size() << " frames (" << bytes << " payload bytes), " << chunks.size()
                      << " audio chunks -> " << remux << std::endl;
            return 0;
        }

        std::vector<motioncam::AudioChunk> audio;
        decoder.loadAudio(audio);
        std::vector<int16_t> pcm;
        for (const auto &chunk : audio)
            pcm.insert(pcm.end(), chunk.second.begin(), chunk.second.end());
        if (!audio.empty()) {
            writeFile(outdir + "/audio.s16", pcm.data(), pcm.size() * sizeof(int16_t));
            std::cout << "audio: " << audio.size() << " chunks, " << pcm.size() << " samples, "
                      << decoder.audioSampleRateHz() << " Hz x " << decoder.numAudioChannels() << std::endl;
        }

        std::vector<std::vector<uint8_t>> data;
        std::vector<nlohmann::json> meta;
        const auto t0 = std::chrono::steady_clock::now();
        if (single && reuse) {
            std::vector<uint8_t> one;
            nlohmann::json m;
            for (size_t i = 0; i < frames.size(); i++)
                decoder.loadFrame(frames[i], one, m);
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << "decoded " << frames.size() << " frames in " << secs << " s (" << (secs > 0 ? frames.size() / secs : 0.0)
                      << " frames/s, one vector for every frame)" << std::endl;
            std::printf("last frame crc32 %08x\n", crc32(one.data(), one.size()));
            return 0;
        } else if (single) {
            data.resize(frames.size());
            meta.resize(frames.size());
            for (size_t i = 0; i < frames.size(); i++)
                decoder.loadFrame(frames[i], data[i], meta[i]);
        } else if (pinned) {
            // geometry of every frame first (metadata only), then one pinned buffer per frame
            std::vector<uint8_t *> bufs(frames.size(), nullptr);
            std::vector<size_t> sizes(frames.size(), 0);
            for (size_t i = 0; i < frames.size(); i++) {
                nlohmann::json m;
                decoder.loadFrameMetadata(frames[i], m);
                sizes[i] = motioncam::Decoder::frameBytes(m["width"], m["height"], output);
                bufs[i] = static_cast<uint8_t *>(mcraw_host_alloc(sizes[i]));
                if (!bufs[i])
                    throw motioncam::IOException("Failed to allocate pinned memory");
            }
            const auto t1 = std::chrono::steady_clock::now();
            decoder.loadFramesInto(frames, bufs, meta, output);
            const auto t2 = std::chrono::steady_clock::now();
            std::cout << "pinned: allocation " << std::chrono::duration<double>(t1 - t0).count() << " s, loadFramesInto "
                      << std::chrono::duration<double>(t2 - t1).count() << " s ("
                      << frames.size() / std::chrono::duration<double>(t2 - t1).count() << " frames/s)" << std::endl;
            data.resize(frames.size());
            for (size_t i = 0; i < frames.size(); i++) { // (outside the decode: only for the checksums / files below)
                data[i].assign(bufs[i], bufs[i] + sizes[i]);
                mcraw_host_free(bufs[i]);
            }
        } else {
            decoder.loadFrames(frames, data, meta, output);
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "decoded " << frames.size() << " frames in " << secs << " s (" << (secs > 0 ? frames.size() / secs : 0.0)
                  << " frames/s, file read + GPU decode + copy out)" << std::endl;
        for (size_t i = 0; i < frames.size(); i++) {
            char name[64];
            if (output.bitsPerSample == 16)
                std::snprintf(name, sizeof(name), "/frame_%06zu.u16", i);
            else
                std::snprintf(name, sizeof(name), "/frame_%06zu.p%d", i, output.bitsPerSample);
            if (!nowrite && !writeFile(outdir + name, data[i].data(), data[i].size()))
                throw motioncam::IOException("Failed to write " + outdir + name);
            const int w = meta[i]["width"], h = meta[i]["height"], t = meta[i]["compressionType"];
            std::printf("frame %zu ts %lld %dx%d type %d crc32 %08x\n", i, static_cast<long long>(frames[i]), w, h, t,
                        crc32(data[i].data(), data[i].size()));
        }
    } catch (const motioncam::MotionCamException &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
