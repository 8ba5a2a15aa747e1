// png_writer.cpp -- raytracer::write_png: 8-bit RGB PNG (what the reference gets from the vendored
// stb_image_write, cpu_launcher.cpp:719).  Filter type 0 on every scanline, one zlib stream, one IDAT.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../../include/raytracer.hpp"

namespace raytracer {
namespace {
void put32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
bool chunk(FILE *f, const char type[4], const unsigned char *data, size_t n) {
    std::vector<unsigned char> head;
    put32(head, (uint32_t)n);
    head.insert(head.end(), type, type + 4);
    uLong crc = crc32(0L, head.data() + 4, 4);
    if (n) crc = crc32(crc, data, (uInt)n);
    std::vector<unsigned char> tail;
    put32(tail, (uint32_t)crc);
    return std::fwrite(head.data(), 1, 8, f) == 8 && (n == 0 || std::fwrite(data, 1, n, f) == n) && std::fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

bool write_png(const char *path, int W, int H, const unsigned char *rgb) {
    if (W <= 0 || H <= 0 || !rgb) return false;
    std::vector<unsigned char> raw((size_t)H * ((size_t)W * 3 + 1));
    for (int y = 0; y < H; ++y) {
        unsigned char *row = raw.data() + (size_t)y * ((size_t)W * 3 + 1);
        row[0] = 0;
        std::memcpy(row + 1, rgb + (size_t)y * W * 3, (size_t)W * 3);
    }
    // run-length strategy: a rendered image is noise on gradients, on which the match search of the default strategy finds next to nothing and takes 4x the time
    // (512x512: 25 ms at level 6 against 7 ms, same file size; the encoder was 10 % of `rt_launcher 8 3`: tools/launcher_timing.py)
    z_stream zs{};
    if (deflateInit2(&zs, 1, Z_DEFLATED, 15, 8, Z_RLE) != Z_OK) return false;
    uLongf zn = deflateBound(&zs, (uLong)raw.size());
    std::vector<unsigned char> z(zn);
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = z.data(); zs.avail_out = (uInt)zn;
    const int zr = deflate(&zs, Z_FINISH);
    zn = zs.total_out;
    deflateEnd(&zs);
    if (zr != Z_STREAM_END) return false;
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)W); put32(ihdr, (uint32_t)H);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && chunk(f, "IHDR", ihdr.data(), ihdr.size()) &&
              chunk(f, "IDAT", z.data(), zn) && chunk(f, "IEND", nullptr, 0);
    return std::fclose(f) == 0 && ok;
}
}  // namespace raytracer
