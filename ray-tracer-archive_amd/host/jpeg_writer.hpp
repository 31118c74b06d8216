// jpeg_writer.hpp — baseline JPEG (JFIF, 8-bit, YCbCr 4:4:4, standard Huffman tables of ITU-T T.81 Annex K) for the host side of the boundary.
//
// The reference stores its frame as JPEG at quality 100 through the third-party `image` crate (main.rs:721, 791-796:
// `output_image.write_to(&mut output_file, ImageOutputFormat::Jpeg(quality))`). That crate is not part of the reference's sources and this
// image has no libjpeg, so the encoder is written out here from the standard: forward DCT in f64, the Annex K quantisation tables scaled
// by the IJG quality rule (quality 100 = all ones), no chroma subsampling, one scan. It is host code, not part of the hot path.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace rtjpeg {

namespace detail {
static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
static const uint8_t kQLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                   18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kQChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
// Annex K.3 Huffman tables: BITS (codes per length 1..16) and HUFFVAL
static const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
    0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39,
    0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8,
    0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
    0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
    0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff { uint16_t code[256]; uint8_t len[256]; };
inline Huff make_huff(const uint8_t* bits, const uint8_t* vals) {       // Annex C: canonical codes from BITS / HUFFVAL
    Huff h{}; uint16_t code = 0; int k = 0;
    for (int l = 1; l <= 16; ++l) { for (int i = 0; i < bits[l - 1]; ++i) { h.code[vals[k]] = code++; h.len[vals[k]] = (uint8_t)l; ++k; } code <<= 1; }
    return h;
}
struct BitWriter {
    std::vector<uint8_t>& out; uint32_t acc = 0; int n = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t bits, int len) {
        acc = (acc << len) | (bits & ((1u << len) - 1u)); n += len;
        while (n >= 8) { const uint8_t b = (uint8_t)(acc >> (n - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); n -= 8; }
    }
    void flush() { if (n > 0) put(0x7Fu, 8 - n); }                          // pad with ones
};
inline int bit_size(int v) { v = v < 0 ? -v : v; int s = 0; while (v) { ++s; v >>= 1; } return s; }
inline void fdct8x8(const double in[64], double out[64]) {                 // separable, straight from the definition (host code: clarity over speed)
    static double c[8][8]; static bool init = false;
    if (!init) { for (int u = 0; u < 8; ++u) for (int x = 0; x < 8; ++x) c[u][x] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0); init = true; }
    double tmp[64];
    for (int y = 0; y < 8; ++y) for (int u = 0; u < 8; ++u) { double s = 0; for (int x = 0; x < 8; ++x) s += c[u][x] * in[y * 8 + x]; tmp[y * 8 + u] = s; }
    for (int u = 0; u < 8; ++u) for (int v = 0; v < 8; ++v) { double s = 0; for (int y = 0; y < 8; ++y) s += c[v][y] * tmp[y * 8 + u]; out[v * 8 + u] = s; }
}
}  // namespace detail

// RGB8 (row 0 = top) -> baseline JPEG bytes. quality 1..100 (the reference passes 100, main.rs:721).
inline std::vector<uint8_t> encode(const uint8_t* rgb, uint32_t w, uint32_t h, int quality) {
    using namespace detail;
    quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    uint8_t q[2][64];
    for (int t = 0; t < 2; ++t) for (int i = 0; i < 64; ++i) { int v = ((t ? kQChroma[i] : kQLuma[i]) * scale + 50) / 100; q[t][i] = (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v)); }
    std::vector<uint8_t> o;
    auto be16 = [&](uint32_t v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); };
    auto marker = [&](uint8_t m) { o.push_back(0xFF); o.push_back(m); };
    marker(0xD8);                                                             // SOI
    marker(0xE0); be16(16); for (char ch : {'J', 'F', 'I', 'F', '\0'}) o.push_back((uint8_t)ch); o.push_back(1); o.push_back(1); o.push_back(0); be16(1); be16(1); o.push_back(0); o.push_back(0);
    for (int t = 0; t < 2; ++t) { marker(0xDB); be16(67); o.push_back((uint8_t)t); for (int i = 0; i < 64; ++i) o.push_back(q[t][kZigzag[i]]); }   // DQT, zigzag order
    marker(0xC0); be16(17); o.push_back(8); be16(h); be16(w); o.push_back(3);   // SOF0: 3 components, 1x1 sampling each
    for (int cidx = 0; cidx < 3; ++cidx) { o.push_back((uint8_t)(cidx + 1)); o.push_back(0x11); o.push_back(cidx ? 1 : 0); }
    auto dht = [&](uint8_t id, const uint8_t* bits, const uint8_t* vals, int n) { marker(0xC4); be16(19 + n); o.push_back(id); for (int i = 0; i < 16; ++i) o.push_back(bits[i]); for (int i = 0; i < n; ++i) o.push_back(vals[i]); };
    dht(0x00, kDcLumaBits, kDcVals, 12); dht(0x10, kAcLumaBits, kAcLumaVals, 162); dht(0x01, kDcChromaBits, kDcVals, 12); dht(0x11, kAcChromaBits, kAcChromaVals, 162);
    marker(0xDA); be16(12); o.push_back(3); o.push_back(1); o.push_back(0x00); o.push_back(2); o.push_back(0x11); o.push_back(3); o.push_back(0x11); o.push_back(0); o.push_back(63); o.push_back(0);
    const Huff hdc[2] = {make_huff(kDcLumaBits, kDcVals), make_huff(kDcChromaBits, kDcVals)}, hac[2] = {make_huff(kAcLumaBits, kAcLumaVals), make_huff(kAcChromaBits, kAcChromaVals)};
    BitWriter bw(o);
    int pred[3] = {0, 0, 0};
    for (uint32_t by = 0; by < h; by += 8)
        for (uint32_t bx = 0; bx < w; bx += 8) {
            double blk[3][64];
            for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
                const uint32_t px = bx + x < w ? bx + x : w - 1, py = by + y < h ? by + y : h - 1;    // edge blocks repeat the last row / column
                const uint8_t* p = rgb + ((size_t)py * w + px) * 3;
                const double R = p[0], G = p[1], B = p[2];
                blk[0][y * 8 + x] = 0.299 * R + 0.587 * G + 0.114 * B - 128.0;
                blk[1][y * 8 + x] = -0.168735892 * R - 0.331264108 * G + 0.5 * B;
                blk[2][y * 8 + x] = 0.5 * R - 0.418687589 * G - 0.081312411 * B;
            }
            for (int cidx = 0; cidx < 3; ++cidx) {
                const int t = cidx ? 1 : 0;
                double f[64]; fdct8x8(blk[cidx], f);
                int zz[64];
                for (int i = 0; i < 64; ++i) zz[i] = (int)std::lround(f[kZigzag[i]] / q[t][kZigzag[i]]);
                const int diff = zz[0] - pred[cidx]; pred[cidx] = zz[0];
                int s = bit_size(diff);
                bw.put(hdc[t].code[s], hdc[t].len[s]);
                if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
                int run = 0;
                for (int i = 1; i < 64; ++i) {
                    if (zz[i] == 0) { ++run; continue; }
                    while (run > 15) { bw.put(hac[t].code[0xF0], hac[t].len[0xF0]); run -= 16; }
                    s = bit_size(zz[i]);
                    const int sym = (run << 4) | s;
                    bw.put(hac[t].code[sym], hac[t].len[sym]);
                    bw.put((uint32_t)(zz[i] < 0 ? zz[i] - 1 : zz[i]), s);
                    run = 0;
                }
                if (run) bw.put(hac[t].code[0x00], hac[t].len[0x00]);         // EOB
            }
        }
    bw.flush();
    marker(0xD9);                                                             // EOI
    return o;
}

}  // namespace rtjpeg
