// image_io.hpp — the image side of the PBRT-v4 front end (host/pbrt_loader.cpp): what the reference does between a texture's
// "filename" and the MIP pyramid its lookups read. Restates (paths relative to /root/reference/src):
//   image.rs:1140-1311      Image::read / read_png: PNG only; 8-bit -> U256 with the texture's colour encoding, 16-bit -> Half after
//                           to_float_linear; Gray(+alpha) -> "Y", RGB(A) -> "R","G","B"(,"A"); RGB images carry the sRGB colour space
//   color.rs:420-724        ColorEncoding: linear / sRGB (256-entry decode table, rational-polynomial encode) / gamma
//   image.rs:134-180        remap_pixel_coords (repeat / clamp / black / octahedral sphere)
//   image.rs:699-802        Image::generate_pyramid: levels stored in the ORIGINAL pixel format (an 8-bit image's levels are re-encoded
//                           to 8 bits), box-filtered in f32; non-power-of-two images first go through float_resize_up
//   image.rs:1007-1138      float_resize_up + resample_weights (8x8 tiles, separable 4-tap filter — with the reference's weights as they
//                           are computed there: all four taps share one position, so after normalisation each is ~1/4)
//   mipmap.rs:42-99         MIPMap::create_from_file: RGBA with an all-ones alpha keeps RGB only
// The PNG container (signature, chunks, zlib / DEFLATE, scanline filters) is the published format (RFC 2083 / 1950 / 1951); the
// reference reads it with the `png` crate (un-vendored), configured for no transformations, so the decoded samples are the file's.
// Third-party arithmetic without vendored source: `half` 2.2.1 (f32 <-> f16: IEEE round-to-nearest-even, restated), `fast_polynomial`
// 0.1.0 (`poly` inside the sRGB curves: Estrin's scheme with FMAs as in shm/texture.h poly7_estrin — parity unpinned), f32::powf /
// f32::sin of the platform (gamma encodings, the resampling window — parity unpinned).
// Host-side only: texels reach the device as the f32 values Image::get_channel returns for each stored level.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "scene_assembly.hpp"

namespace pbrt {

// ---- DEFLATE (RFC 1951) inside a zlib stream (RFC 1950) -----------------------------------------------------------------------------
class Inflater {
public:
    Inflater(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    std::vector<uint8_t> run(size_t size_hint) {
        if (n_ < 6) fail("PNG: truncated zlib stream");
        const unsigned cmf = p_[0], flg = p_[1];
        if ((cmf & 15u) != 8u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) fail("PNG: not a zlib/deflate stream");
        pos_ = 2;
        // size_hint comes from an unchecked IHDR: it bounds the output (a PNG's inflated size is exactly (stride + 1) * height) but is
        // not trusted as an allocation size — DEFLATE expands at most 1032 : 1, so a few hundred bytes cannot reserve gigabytes
        limit_ = size_hint;
        out_.reserve(std::min(size_hint, n_ > ((size_t)-1) / 1032 ? size_hint : n_ * 1032));
        bool last = false;
        while (!last) {
            last = bits(1) != 0;
            const unsigned type = bits(2);
            if (type == 0) stored();
            else if (type == 1) { fixed_tables(); codes(); }
            else if (type == 2) { dynamic_tables(); codes(); }
            else fail("PNG: invalid deflate block type");
        }
        // Adler-32 of the output follows, big-endian, at the next byte boundary
        nbits_ = 0; acc_ = 0;
        if (pos_ + 4 > n_) fail("PNG: truncated zlib stream (checksum)");
        const uint32_t want = ((uint32_t)p_[pos_] << 24) | ((uint32_t)p_[pos_ + 1] << 16) | ((uint32_t)p_[pos_ + 2] << 8) | p_[pos_ + 3];
        uint32_t a = 1, b = 0;
        for (size_t i = 0; i < out_.size();) {
            const size_t end = std::min(out_.size(), i + 5552);
            for (; i < end; ++i) { a += out_[i]; b += a; }
            a %= 65521u; b %= 65521u;
        }
        if (((b << 16) | a) != want) fail("PNG: zlib checksum mismatch");
        return std::move(out_);
    }

private:
    struct Huff { uint16_t count[16]; uint16_t symbol[320]; };
    const uint8_t* p_;
    size_t n_, pos_ = 0;
    uint32_t acc_ = 0;
    int nbits_ = 0;
    std::vector<uint8_t> out_;
    size_t limit_ = (size_t)-1;  // the raw size the header implies: more output than that is a malformed file
    Huff lit_, dist_;

    unsigned bits(int need) {
        while (nbits_ < need) {
            if (pos_ >= n_) fail("PNG: truncated deflate data");
            acc_ |= (uint32_t)p_[pos_++] << nbits_;
            nbits_ += 8;
        }
        const unsigned v = acc_ & ((1u << need) - 1u);
        acc_ >>= need;
        nbits_ -= need;
        return v;
    }
    static void build(Huff& h, const uint8_t* len, int n) {
        memset(h.count, 0, sizeof(h.count));
        for (int i = 0; i < n; ++i) h.count[len[i]]++;
        h.count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) { left <<= 1; left -= h.count[l]; if (left < 0) fail("PNG: over-subscribed Huffman code"); }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h.count[l];
        for (int i = 0; i < n; ++i) if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(const Huff& h) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)bits(1);
            const int count = h.count[l];
            if (code - count < first) return h.symbol[index + (code - first)];
            index += count;
            first += count;
            first <<= 1;
            code <<= 1;
        }
        fail("PNG: invalid Huffman code");
    }
    void stored() {
        nbits_ = 0; acc_ = 0;
        if (pos_ + 4 > n_) fail("PNG: truncated stored block");
        const unsigned len = p_[pos_] | (p_[pos_ + 1] << 8), nlen = p_[pos_ + 2] | (p_[pos_ + 3] << 8);
        pos_ += 4;
        if ((len ^ 0xffffu) != nlen) fail("PNG: stored block length mismatch");
        if (pos_ + len > n_) fail("PNG: truncated stored block");
        if (out_.size() + len > limit_) fail("PNG: image data longer than the header's dimensions imply");
        out_.insert(out_.end(), p_ + pos_, p_ + pos_ + len);
        pos_ += len;
    }
    void fixed_tables() {
        uint8_t len[320];
        int i = 0;
        for (; i < 144; ++i) len[i] = 8;
        for (; i < 256; ++i) len[i] = 9;
        for (; i < 280; ++i) len[i] = 7;
        for (; i < 288; ++i) len[i] = 8;
        build(lit_, len, 288);
        for (i = 0; i < 30; ++i) len[i] = 5;
        build(dist_, len, 30);
    }
    void dynamic_tables() {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (nlen > 286 || ndist > 30) fail("PNG: bad deflate code counts");
        uint8_t len[320];
        memset(len, 0, sizeof(len));
        for (int i = 0; i < ncode; ++i) len[order[i]] = (uint8_t)bits(3);
        Huff cl;
        build(cl, len, 19);
        int idx = 0;
        uint8_t lens[320];
        while (idx < nlen + ndist) {
            const int sym = decode(cl);
            if (sym < 16) lens[idx++] = (uint8_t)sym;
            else {
                uint8_t prev = 0;
                int rep;
                if (sym == 16) { if (idx == 0) fail("PNG: repeat without a previous length"); prev = lens[idx - 1]; rep = 3 + (int)bits(2); }
                else if (sym == 17) rep = 3 + (int)bits(3);
                else rep = 11 + (int)bits(7);
                if (idx + rep > nlen + ndist) fail("PNG: too many code lengths");
                while (rep--) lens[idx++] = prev;
            }
        }
        if (lens[256] == 0) fail("PNG: deflate block without an end code");
        build(lit_, lens, nlen);
        build(dist_, lens + nlen, ndist);
    }
    void codes() {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (;;) {
            int sym = decode(lit_);
            if (sym < 256) { if (out_.size() >= limit_) fail("PNG: image data longer than the header's dimensions imply"); out_.push_back((uint8_t)sym); }
            else if (sym == 256) return;
            else {
                sym -= 257;
                if (sym >= 29) fail("PNG: invalid length code");
                const size_t len = lbase[sym] + bits(lext[sym]);
                const int ds = decode(dist_);
                if (ds >= 30) fail("PNG: invalid distance code");
                const size_t dist = dbase[ds] + bits(dext[ds]);
                if (dist > out_.size()) fail("PNG: distance beyond the start of the output");
                if (out_.size() + len > limit_) fail("PNG: image data longer than the header's dimensions imply");
                size_t from = out_.size() - dist;
                for (size_t k = 0; k < len; ++k) out_.push_back(out_[from + k]);
            }
        }
    }
};

// ---- PNG (the samples as the file holds them: rows top-down, 16-bit samples big-endian) -----------------------------------------------
struct PngImage {
    int width = 0, height = 0, channels = 0, depth = 0;
    int color_type = 0;  // 0 gray, 2 rgb, 4 gray + alpha, 6 rgb + alpha
    std::vector<uint8_t> data;
};
inline uint32_t png_crc(const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
inline PngImage png_decode(const std::vector<uint8_t>& file, const std::string& name) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) fail(name + ": not a PNG file");
    PngImage img;
    std::vector<uint8_t> idat;
    bool have_ihdr = false, done = false;
    int interlace = 0;
    size_t pos = 8;
    auto be32 = [&](size_t at) { return ((uint32_t)file[at] << 24) | ((uint32_t)file[at + 1] << 16) | ((uint32_t)file[at + 2] << 8) | file[at + 3]; };
    while (!done) {
        if (pos + 12 > file.size()) fail(name + ": truncated PNG file");
        const uint32_t len = be32(pos);
        if ((uint64_t)pos + 12ull + len > file.size()) fail(name + ": truncated PNG chunk");
        const uint8_t* type = &file[pos + 4];
        const uint8_t* body = &file[pos + 8];
        if (png_crc(type, 4 + (size_t)len) != be32(pos + 8 + len)) fail(name + ": PNG chunk checksum mismatch");
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) fail(name + ": bad IHDR");
            img.width = (int)be32(pos + 8);
            img.height = (int)be32(pos + 12);
            img.depth = body[8];
            img.color_type = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0) fail(name + ": unknown PNG compression / filter method");
            have_ihdr = true;
        } else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) done = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || img.width <= 0 || img.height <= 0) fail(name + ": PNG without a valid IHDR");
    if (img.color_type == 3) fail(name + ": Indexed PNGs are not supported!");  // image.rs:1296
    if (img.color_type != 0 && img.color_type != 2 && img.color_type != 4 && img.color_type != 6) fail(name + ": unknown PNG colour type");
    if (img.depth != 8 && img.depth != 16) fail(name + ": Unsupported bit depth");  // image.rs:1200, 1292
    if (interlace != 0) fail(name + ": interlaced PNGs are not supported", SHM_ERR_UNSUPPORTED);
    img.channels = img.color_type == 0 ? 1 : (img.color_type == 2 ? 3 : (img.color_type == 4 ? 2 : 4));
    const size_t bpp = (size_t)img.channels * (size_t)(img.depth / 8), stride = bpp * (size_t)img.width;
    if ((uint64_t)stride * (uint64_t)img.height > (1ull << 32)) fail(name + ": PNG too large");
    std::vector<uint8_t> raw = Inflater(idat.data(), idat.size()).run((stride + 1) * (size_t)img.height);
    if (raw.size() != (stride + 1) * (size_t)img.height) fail(name + ": PNG image data has the wrong size");
    img.data.resize(stride * (size_t)img.height);
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < img.height; ++y) {
        const uint8_t* in = &raw[(stride + 1) * (size_t)y];
        const int filter = in[0];
        ++in;
        uint8_t* cur = &img.data[stride * (size_t)y];
        const uint8_t* up = y ? cur - stride : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: fail(name + ": unknown PNG scanline filter");
            }
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    return img;
}
inline std::vector<uint8_t> read_file_bytes(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) fail("unable to read " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

// ---- f16 (crate half 2.2.1: IEEE 754 binary16, round to nearest even) ---------------------------------------------------------------
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xffu;
    uint32_t man = x & 0x7fffffu;
    if (exp == 0xffu) return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u));
    const int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;
        uint32_t half = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;  // may carry into the exponent, up to infinity: as IEEE
    return (uint16_t)(sign | half);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// ---- ColorEncoding (color.rs:420-724) ------------------------------------------------------------------------------------------------
// fast_polynomial::poly over n coefficients, lowest order first (Estrin with FMAs; unpinned, see the header)
inline float poly5(float x, const float c[5]) { const float x2 = x * x, x4 = x2 * x2; return fmaf(c[4], x4, fmaf(fmaf(c[3], x, c[2]), x2, fmaf(c[1], x, c[0]))); }
inline float poly6(float x, const float c[6]) { const float x2 = x * x, x4 = x2 * x2; return fmaf(fmaf(c[5], x, c[4]), x4, fmaf(fmaf(c[3], x, c[2]), x2, fmaf(c[1], x, c[0]))); }
inline float linear_to_srgb(float value) {  // color.rs:653-682
    if (value <= 0.0031308f) return 12.92f * value;
    const float s = value < 0.0f ? 0.0f : sqrtf(value);
    static const float p[6] = {-0.0016829072605308378f, 0.03453868659826638f, 0.7642611304733891f, 2.0041169284241644f, 0.7551545191665577f, -0.016202083165206348f};
    static const float q[6] = {4.178892964897981e-7f, -0.00004375359692957097f, 0.03467195408529984f, 0.6085338522168684f, 1.8970238036421054f, 1.0f};
    return poly6(s, p) / poly6(s, q) * value;
}
inline float srgb_to_linear(float value) {  // color.rs:684-711
    if (value <= 0.04045f) return value * (1.0f / 12.92f);
    static const float p[5] = {-0.0163933279112946f, -0.7386328024653209f, -11.199318357635072f, -47.46726633009393f, -36.04572663838034f};
    static const float q[5] = {-0.004261480793199332f, -19.140923959601675f, -59.096406619244426f, -18.225745396846637f, 1.0f};
    return poly5(value, p) / poly5(value, q) * value;
}
struct ColorEncoding {
    enum Kind { LINEAR, SRGB, GAMMA } kind = SRGB;
    float gamma = 1.0f;
    std::vector<float> apply_lut, inverse_lut;
    static ColorEncoding get(const std::string& name) {  // ColorEncoding::get (color.rs:487-525)
        ColorEncoding e;
        if (name == "linear") e.kind = LINEAR;
        else if (name == "srgb" || name == "sRGB") e.kind = SRGB;
        else {
            char word[16] = {0};
            float g = 0.0f;
            char tail = 0;
            if (sscanf(name.c_str(), "%15s %f %c", word, &g, &tail) != 2 || strcmp(word, "gamma") != 0) fail("Expected gamma <value> for color encoding.");
            if (g == 0.0f) fail("Gamma value cannot be 0.0");
            e.kind = GAMMA;
            e.gamma = g;
            e.apply_lut.resize(256);
            for (int i = 0; i < 256; ++i) e.apply_lut[i] = powf((float)i / 255.0f, g);
            e.inverse_lut.resize(1024);
            for (int i = 0; i < 1024; ++i) { const float v = 255.0f * powf((float)i / 1023.0f, 1.0f / g) + 0.5f; e.inverse_lut[i] = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v); }
        }
        return e;
    }
    float to_linear(uint8_t v) const {
        if (kind == LINEAR) return (float)v / 255.0f;
        if (kind == SRGB) { static const std::vector<float> lut = PBRT_TABLE(SRGB_TO_LINEAR_LUT); return lut[v]; }
        return apply_lut[v];
    }
    static uint8_t sat_u8(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v); }  // Rust `as u8`: saturating, NaN -> 0
    uint8_t from_linear(float v) const {
        if (kind == LINEAR) { const float t = v * 255.0f + 0.5f; return sat_u8(t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t)); }
        if (kind == SRGB) {  // linear_to_srgb_8(value, 0.0), color.rs:713-721
            if (v <= 0.0f) return 0;
            if (v >= 1.0f) return 255;
            const float t = roundf(255.0f * linear_to_srgb(v) + 0.0f);
            return sat_u8(t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t));
        }
        float t = v * 1023.0f;
        t = t < 0.0f ? 0.0f : (t > 1023.0f ? 1023.0f : t);
        return sat_u8(inverse_lut[!(t > 0.0f) ? 0 : (size_t)t]);
    }
    float to_float_linear(float v) const { return kind == LINEAR ? v : (kind == SRGB ? srgb_to_linear(v) : powf(v, gamma)); }
};

// ---- Image (the subset the front end needs) ---------------------------------------------------------------------------------------------
enum WrapMode { WRAP_BLACK = SHM_WRAP_BLACK, WRAP_CLAMP = SHM_WRAP_CLAMP, WRAP_REPEAT = SHM_WRAP_REPEAT, WRAP_OCTAHEDRAL = SHM_WRAP_OCTAHEDRAL_SPHERE };
inline bool remap_pixel_coords(int p[2], const int res[2], WrapMode wrap) {  // image.rs:134-180
    if (wrap == WRAP_OCTAHEDRAL) {
        if (p[0] < 0) { p[0] = -p[0]; p[1] = res[1] - 1 - p[1]; }
        else if (p[0] >= res[0]) { p[0] = 2 * res[0] - 1 - p[0]; p[1] = res[1] - 1 - p[1]; }
        if (p[1] < 0) { p[0] = res[0] - 1 - p[0]; p[1] = -p[1]; }
        else if (p[1] >= res[1]) { p[0] = res[0] - 1 - p[0]; p[1] = 2 * res[1] - 1 - p[1]; }
        if (res[0] == 1) p[0] = 0;
        if (res[1] == 1) p[1] = 0;
        return true;
    }
    for (int c = 0; c < 2; ++c) {
        if (p[c] >= 0 && p[c] < res[c]) continue;
        if (wrap == WRAP_BLACK) return false;
        if (wrap == WRAP_CLAMP) p[c] = p[c] < 0 ? 0 : res[c] - 1;
        else { int r = p[c] - (p[c] / res[c]) * res[c]; p[c] = r < 0 ? r + res[c] : r; }  // math.rs:439-452 modulo
    }
    return true;
}
struct HostImage {
    enum Format { U256, HALF, FLOAT } format = FLOAT;
    int res[2] = {0, 0};
    int nc = 0;
    ColorEncoding encoding;   // U256 only
    bool srgb_color_space = false;  // ImageMetadata::color_space (RGB PNGs: sRGB)
    std::vector<uint8_t> p8;
    std::vector<uint16_t> p16;
    std::vector<float> p32;
    static HostImage make(Format f, int w, int h, int nc, const ColorEncoding& enc) {
        HostImage im;
        im.format = f; im.res[0] = w; im.res[1] = h; im.nc = nc; im.encoding = enc;
        const size_t n = (size_t)w * (size_t)h * (size_t)nc;
        if (f == U256) im.p8.assign(n, 0); else if (f == HALF) im.p16.assign(n, 0); else im.p32.assign(n, 0.0f);
        return im;
    }
    size_t offset(int x, int y) const { return (size_t)nc * ((size_t)y * (size_t)res[0] + (size_t)x); }
    float get(int x, int y, int c) const {  // Image::get_channel (image.rs:445-450)
        const size_t o = offset(x, y) + (size_t)c;
        return format == U256 ? encoding.to_linear(p8[o]) : (format == HALF ? f16_to_f32(p16[o]) : p32[o]);
    }
    void set(int x, int y, int c, float v) {  // Image::set_channel (image.rs:648-661)
        const size_t o = offset(x, y) + (size_t)c;
        if (v != v) v = 0.0f;
        if (format == U256) p8[o] = encoding.from_linear(v); else if (format == HALF) p16[o] = f32_to_f16(v); else p32[o] = v;
    }
    HostImage select_channels(const std::vector<int>& ch) const {  // image.rs:677-697
        HostImage out = make(format, res[0], res[1], (int)ch.size(), encoding);
        out.srgb_color_space = srgb_color_space;
        // through get_channel / set_channel, as the reference (a decode + encode round trip for 8-bit images)
        for (int y = 0; y < res[1]; ++y) for (int x = 0; x < res[0]; ++x) for (size_t k = 0; k < ch.size(); ++k) out.set(x, y, (int)k, get(x, y, ch[k]));
        return out;
    }
    std::vector<float> to_floats() const {
        std::vector<float> v((size_t)res[0] * (size_t)res[1] * (size_t)nc);
        size_t k = 0;
        for (int y = 0; y < res[1]; ++y) for (int x = 0; x < res[0]; ++x) for (int c = 0; c < nc; ++c) v[k++] = get(x, y, c);
        return v;
    }
};

// Image::read_png (image.rs:1151-1311)
inline HostImage image_from_png(const PngImage& png, const ColorEncoding& encoding, const std::string& name) {
    const bool gray = png.color_type == 0 || png.color_type == 4;
    const int w = png.width, h = png.height;
    if (png.depth == 8) {
        if (gray) {  // "Y"; the alpha of GrayscaleAlpha is stripped (image.rs:1169-1176)
            HostImage im = HostImage::make(HostImage::U256, w, h, 1, encoding);
            for (size_t i = 0; i < im.p8.size(); ++i) im.p8[i] = png.data[i * (size_t)png.channels];
            return im;
        }
        HostImage im = HostImage::make(HostImage::U256, w, h, png.channels, encoding);
        im.p8 = png.data;
        im.srgb_color_space = true;
        return im;
    }
    if (gray) {
        // image.rs:1184-1200: the two bytes of each 16-bit sample are reinterpreted as a LITTLE-endian f16 (the file holds big-endian
        // integers) before to_float_linear — kept as the reference has it. With an alpha channel the reference strips every second
        // BYTE, which leaves too few bytes for the loop that follows and panics.
        if (png.color_type == 4) fail(name + ": 16-bit grey + alpha PNGs make the reference panic (image.rs:1169-1200)", SHM_ERR_UNSUPPORTED);
        HostImage im = HostImage::make(HostImage::HALF, w, h, 1, encoding);
        for (size_t i = 0; i < im.p16.size(); ++i) {
            const uint16_t bits16 = (uint16_t)(png.data[2 * i] | (png.data[2 * i + 1] << 8));
            const float v = encoding.to_float_linear(f16_to_f32(bits16));
            im.p16[i] = f32_to_f16(v != v ? 0.0f : v);  // set_channel stores NaN as 0
        }
        return im;
    }
    HostImage im = HostImage::make(HostImage::HALF, w, h, png.channels, encoding);  // image.rs:1232-1290
    for (size_t i = 0; i < im.p16.size(); ++i) {
        const float v = (float)(((int)png.data[2 * i] << 8) + (int)png.data[2 * i + 1]) / 65535.0f;
        im.p16[i] = f32_to_f16(encoding.to_float_linear(v));
    }
    im.srgb_color_space = true;
    return im;
}
inline HostImage image_read(const std::string& path, const ColorEncoding& encoding) {  // Image::read (image.rs:1140-1149)
    const size_t dot = path.find_last_of('.');
    if (dot == std::string::npos || path.substr(dot + 1) != "png") fail("Unsupported file extension for " + path, SHM_ERR_UNSUPPORTED);
    return image_from_png(png_decode(read_file_bytes(path), path), encoding, path);
}

// MIPMap::create_from_file's channel selection (mipmap.rs:50-88)
inline HostImage mipmap_select_channels(const HostImage& image) {
    if (image.nc == 1) return image;
    if (image.nc == 4) {
        bool all_one = true;
        for (int y = 0; y < image.res[1] && all_one; ++y) for (int x = 0; x < image.res[0]; ++x) if (image.get(x, y, 3) != 1.0f) { all_one = false; break; }
        return all_one ? image.select_channels({0, 1, 2}) : image;
    }
    return image;  // "R", "G", "B"
}

// windowed_sinc (math.rs:404-434); sin_over_x(x) = 1 when 1 - x*x == 1, else sin(x) / x
inline float windowed_sinc(float x, float radius, float tau) {
    if (fabsf(x) > radius) return 0.0f;
    auto sinc = [](float v) { const float t = 3.14159265358979323846f * v; return (1.0f - t * t == 1.0f) ? 1.0f : sinf(t) / t; };
    return sinc(x) * sinc(x / tau);
}
struct ResampleWeight { int first_pixel = 0; float weight[4] = {0, 0, 0, 0}; };
inline std::vector<ResampleWeight> resample_weights(size_t old_res, size_t new_res) {  // image.rs:1113-1138
    std::vector<ResampleWeight> wt(new_res);
    const float filter_radius = 2.0f, tau = 2.0f;
    for (size_t i = 0; i < new_res; ++i) {
        const float center = ((float)i + 0.5f) * (float)old_res / (float)new_res;
        wt[i].first_pixel = std::max((int)floorf(center - filter_radius + 0.5f), 0);
        for (int j = 0; j < 4; ++j) {
            const float pos = (float)wt[i].first_pixel + 0.5f;  // (the same position for the four taps: as the reference computes it)
            wt[i].weight[j] = windowed_sinc(pos - center, filter_radius, tau);
        }
        const float inv = 1.0f / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; ++j) wt[i].weight[j] *= inv;
    }
    return wt;
}
// Image::float_resize_up (image.rs:1007-1111)
inline HostImage float_resize_up(const HostImage& src, int new_w, int new_h, WrapMode wrap) {
    if (!(new_w > src.res[0]) || !(new_h > src.res[1]))
        fail("image of " + std::to_string(src.res[0]) + " x " + std::to_string(src.res[1]) + ": float_resize_up needs BOTH sides to grow to the next power of two (image.rs:1009-1010 asserts it)");
    HostImage out = HostImage::make(HostImage::FLOAT, new_w, new_h, src.nc, ColorEncoding());
    out.srgb_color_space = src.srgb_color_space;
    const std::vector<ResampleWeight> xw = resample_weights((size_t)src.res[0], (size_t)new_w), yw = resample_weights((size_t)src.res[1], (size_t)new_h);
    const int nc = src.nc;
    for (int ty = 0; ty < new_h; ty += 8) for (int tx = 0; tx < new_w; tx += 8) {  // Tile::tile(bounds, 8, 8)
        const int tx1 = std::min(tx + 8, new_w), ty1 = std::min(ty + 8, new_h);
        const int in_min[2] = {xw[tx].first_pixel, yw[ty].first_pixel}, in_max[2] = {xw[tx1 - 1].first_pixel + 4, yw[ty1 - 1].first_pixel + 4};
        const int nx_in = in_max[0] - in_min[0], ny_in = in_max[1] - in_min[1], nx_out = tx1 - tx, ny_out = ty1 - ty;
        std::vector<float> in_buf((size_t)nx_in * (size_t)ny_in * (size_t)nc);
        size_t k = 0;
        for (int y = in_min[1]; y < in_max[1]; ++y) for (int x = in_min[0]; x < in_max[0]; ++x) {  // copy_rect_out (image.rs:918-978)
            int p[2] = {x, y};
            if (!remap_pixel_coords(p, src.res, wrap)) fail("resizing an image with wrap mode \"black\" panics in the reference (image.rs:835)");
            for (int c = 0; c < nc; ++c) in_buf[k++] = src.get(p[0], p[1], c);
        }
        std::vector<float> x_buf((size_t)ny_in * (size_t)nx_out * (size_t)nc);
        k = 0;
        for (int y = in_min[1]; y < in_max[1]; ++y) for (int x = tx; x < tx1; ++x) {
            const ResampleWeight& w = xw[x];
            size_t o = (size_t)nc * (size_t)((w.first_pixel - in_min[0]) + (y - in_min[1]) * nx_in);
            for (int c = 0; c < nc; ++c, ++o) x_buf[k++] = w.weight[0] * in_buf[o] + w.weight[1] * in_buf[o + nc] + w.weight[2] * in_buf[o + 2 * nc] + w.weight[3] * in_buf[o + 3 * nc];
        }
        for (int x = 0; x < nx_out; ++x) for (int y = 0; y < ny_out; ++y) {
            const ResampleWeight& w = yw[y + ty];
            size_t o = (size_t)nc * (size_t)(x + nx_out * (w.first_pixel - in_min[1]));
            const size_t step = (size_t)nc * (size_t)nx_out;
            for (int c = 0; c < nc; ++c, ++o)
                out.p32[out.offset(tx + x, ty + y) + (size_t)c] =
                    std::max(0.0f, w.weight[0] * x_buf[o] + w.weight[1] * x_buf[o + step] + w.weight[2] * x_buf[o + 2 * step] + w.weight[3] * x_buf[o + 3 * step]);
        }
    }
    return out;
}
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
// Image::generate_pyramid (image.rs:699-802): finest level first, the last one 1x1; every level in the source image's format
inline std::vector<HostImage> generate_pyramid(const HostImage& source, WrapMode wrap) {
    HostImage image;
    if (!is_pow2(source.res[0]) || !is_pow2(source.res[1])) image = float_resize_up(source, next_pow2(source.res[0]), next_pow2(source.res[1]), wrap);
    else if (source.format != HostImage::FLOAT) {
        image = HostImage::make(HostImage::FLOAT, source.res[0], source.res[1], source.nc, source.encoding);
        image.p32 = source.to_floats();
    } else image = source;
    const int nc = image.nc;
    const int n_levels = 1 + (int)log2f((float)std::max(image.res[0], image.res[1]));
    std::vector<HostImage> pyramid;
    auto store = [&](const HostImage& level) {  // copy_rect_in into a level of the original format (image.rs:848-916)
        HostImage st = HostImage::make(source.format, level.res[0], level.res[1], nc, source.encoding);
        st.srgb_color_space = source.srgb_color_space;
        for (size_t i = 0; i < level.p32.size(); ++i) {
            if (source.format == HostImage::U256) st.p8[i] = source.encoding.from_linear(level.p32[i]);
            else if (source.format == HostImage::HALF) st.p16[i] = f32_to_f16(level.p32[i]);
            else st.p32[i] = level.p32[i];
        }
        pyramid.push_back(std::move(st));
    };
    for (int i = 0; i < n_levels - 1; ++i) {
        store(image);
        const int nw = std::max(1, (image.res[0] + 1) / 2), nh = std::max(1, (image.res[1] + 1) / 2);
        HostImage next = HostImage::make(HostImage::FLOAT, nw, nh, nc, source.encoding);
        size_t d1 = (size_t)nc, d2 = (size_t)nc * (size_t)image.res[0], d3 = (size_t)nc * ((size_t)image.res[0] + 1);
        if (image.res[0] == 1) { d1 = 0; d3 -= (size_t)nc; }
        if (image.res[1] == 1) { d2 = 0; d3 -= (size_t)nc * (size_t)image.res[0]; }
        for (int y = 0; y < nh; ++y) {
            size_t s = image.offset(0, 2 * y), o = next.offset(0, y);
            for (int x = 0; x < nw; ++x) {
                for (int c = 0; c < nc; ++c, ++s, ++o) next.p32[o] = 0.25f * (image.p32[s] + image.p32[s + d1] + image.p32[s + d2] + image.p32[s + d3]);
                s += (size_t)nc;
            }
        }
        image = std::move(next);
    }
    if (image.res[0] != 1 || image.res[1] != 1) fail("generate_pyramid: the coarsest level is not 1 x 1");
    store(image);
    return pyramid;
}

}  // namespace pbrt
