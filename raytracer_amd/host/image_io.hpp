// image_io.hpp — writes the frame render() produced (BGR8, row-major, ref src/raytracer.cpp:434,446-448) to a file,
// in place of the reference's cv::imwrite(outFilename, img) (ref src/raytracer.cpp:720). No dependencies:
//   *.ppm  binary PPM (P6, RGB)
//   *.png  PNG, 8-bit RGB, zlib "stored" (uncompressed) deflate blocks: valid for every decoder, ~3 bytes per pixel
// Anything else throws (the reference leaves the choice to OpenCV's extension table).
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_image {

inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

inline void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

inline void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    be32(out, uint32_t(data.size()));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(out.data() + start, out.size() - start));
}

inline std::vector<uint8_t> encode_png(const uint8_t* bgr, uint32_t W, uint32_t H) {
    std::vector<uint8_t> raw;                       // filter byte 0 + RGB per row
    raw.reserve(size_t(H) * (size_t(W) * 3 + 1));
    for (uint32_t y = 0; y < H; ++y) {
        raw.push_back(0);
        const uint8_t* row = bgr + size_t(y) * W * 3;
        for (uint32_t x = 0; x < W; ++x) { raw.push_back(row[3 * x + 2]); raw.push_back(row[3 * x + 1]); raw.push_back(row[3 * x]); }
    }
    std::vector<uint8_t> z;                         // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;                          // Adler-32
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8)); z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    } while (pos < raw.size());
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, W); be32(ihdr, H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return out;
}

inline bool ends_with(const std::string& s, const char* suffix) {
    const std::string t(suffix);
    return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0;
}

}  // namespace rt_image

// Writes the W x H BGR8 frame to `filename` (.ppm or .png).
inline void write_image(const std::string& filename, const unsigned char* bgr, unsigned int W, unsigned int H) {
    if (!bgr || W == 0 || H == 0) throw std::runtime_error("write_image: empty image");
    std::vector<uint8_t> bytes;
    if (rt_image::ends_with(filename, ".png")) {
        bytes = rt_image::encode_png(bgr, W, H);
    } else if (rt_image::ends_with(filename, ".ppm")) {
        const std::string head = "P6\n" + std::to_string(W) + " " + std::to_string(H) + "\n255\n";
        bytes.assign(head.begin(), head.end());
        for (size_t i = 0; i < size_t(W) * H; ++i) { bytes.push_back(bgr[3 * i + 2]); bytes.push_back(bgr[3 * i + 1]); bytes.push_back(bgr[3 * i]); }
    } else {
        throw std::runtime_error("write_image: unsupported extension of " + filename + " (use .png or .ppm)");
    }
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) throw std::runtime_error("write_image: cannot open " + filename);
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    std::fclose(f);
    if (n != bytes.size()) throw std::runtime_error("write_image: short write to " + filename);
}
