// Own OBJ / MTL / XML / texture reader reproducing the subset the reference accepts (src/model.cpp:44-281):
//   OBJ  : mtllib, v, vn, vt, usemtl, f a/b/c a/b/c a/b/c (first three corners only, all indices mandatory, 1-based)
//   MTL  : newmtl, Kd, Ks, Tr, Ns, Ni, map_Kd; a line containing '#' anywhere is skipped (model.cpp:174)
//   XML  : <camera width height fovy><eye/><lookat/><up/></camera>, <light mtlname radiance="r,g,b"/>
// Face corners follow Wavefront semantics v/vt/vn; `reference_index_order` reproduces the reference's reading of the
// second index as the NORMAL and the third as the TEXCOORD (SURVEY A-14) for files that were authored against it.
#include "Model.h"
#include "Jpeg.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <charconv>
#include <sstream>
#include <system_error>
#include <thread>
#include <zlib.h>

namespace {
// Text after the first n characters; empty when the line is shorter (a bare "vn" must not throw).
std::string rest(const std::string& line, size_t n) { return line.size() > n ? line.substr(n) : std::string(); }
std::string dir_of(const std::string& p) { size_t k = p.find_last_of("/\\"); return k == std::string::npos ? std::string(".") : p.substr(0, k); }
// Whole file as bytes; empty for anything that is not a readable regular file (reading a directory through a
// streambuf iterator throws in libstdc++, e.g. "map_Kd" with no name resolves to the scene directory).
template <class Bytes> Bytes read_file(const std::string& filename) {
    Bytes out;
    std::error_code ec;
    if (!std::filesystem::is_regular_file(filename, ec) || ec) return out;
    std::ifstream in(filename, std::ios::binary);
    if (!in) return out;
    try { out.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>()); } catch (const std::ios_base::failure&) { out.clear(); }
    return out;
}
bool starts(const std::string& s, const char* t) { return s.compare(0, std::strlen(t), t) == 0; }

// ---- PNG via zlib: every colour type (grey, RGB, palette, grey+alpha, RGBA), bit depths 1-16, plain or Adam7-interlaced; returns RGB bytes
//      (16-bit samples keep their high byte, alpha is dropped -- what stb_image's 8-bit RGB request does)
uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
bool load_png(const std::vector<unsigned char>& f, int& w, int& h, std::vector<unsigned char>& rgb) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (f.size() < 33 || std::memcmp(f.data(), sig, 8) != 0) return false;
    size_t pos = 8; int depth = 0, ctype = 0, interlace = 0; std::vector<unsigned char> idat, plte;
    while (pos + 12 <= f.size()) {
        uint32_t len = be32(&f[pos]); const char* type = (const char*)&f[pos + 4];
        if (pos + 12 + size_t(len) > f.size()) return false;
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = int(be32(&f[pos + 8])); h = int(be32(&f[pos + 12])); depth = f[pos + 16]; ctype = f[pos + 17]; interlace = f[pos + 20]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(f.begin() + pos + 8, f.begin() + pos + 8 + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), f.begin() + pos + 8, f.begin() + pos + 8 + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + size_t(len);
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (ch == 0 || !depth_ok || interlace > 1 || w <= 0 || h <= 0 || size_t(w) * size_t(h) > (size_t(1) << 28) || (ctype == 3 && plte.size() < 3)) return false;
    const size_t bpp = std::max<size_t>(1, size_t(ch) * depth / 8);            // filter distance in bytes
    // One pass = one filtered sub-image.  A non-interlaced file has a single pass (every pixel); an Adam7 file (PNG spec section 8.2) has
    // seven, pass p holding the pixels (x0 + i dx, y0 + j dy) -- each with its own scanlines, filter bytes and row padding.
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const int n_pass = interlace ? 7 : 1;
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; k++) {
        const Pass& ps = interlace ? adam7[k] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw > 0 && ph > 0) raw_size += (((size_t(pw) * ch * depth + 7) / 8) + 1) * size_t(ph);
    }
    std::vector<unsigned char> raw(raw_size);
    uLongf out = uLongf(raw.size());
    if (uncompress(raw.data(), &out, idat.data(), uLong(idat.size())) != Z_OK || out != raw.size()) return false;
    rgb.resize(size_t(w) * h * 3);
    const int maxv = (1 << (depth < 8 ? depth : 8)) - 1;
    size_t rp = 0;
    std::vector<unsigned char> img;
    for (int k = 0; k < n_pass; k++) {
        const Pass& ps = interlace ? adam7[k] : whole;
        const int pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = (size_t(pw) * ch * depth + 7) / 8;
        img.assign(stride * size_t(ph), 0);
        for (int y = 0; y < ph; y++) {
            const unsigned char ft = raw[rp]; const unsigned char* s = &raw[rp + 1]; rp += stride + 1;
            unsigned char* d = &img[size_t(y) * stride]; const unsigned char* up = y ? &img[size_t(y - 1) * stride] : nullptr;
            for (size_t i = 0; i < stride; i++) {
                int a = i >= bpp ? d[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, v = s[i];
                switch (ft) {
                    case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break;
                    case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: break;
                }
                d[i] = (unsigned char)v;
            }
        }
        for (int y = 0; y < ph; y++) {
            const unsigned char* row = &img[size_t(y) * stride];
            for (int x = 0; x < pw; x++) {
                unsigned char* o = &rgb[3 * (size_t(ps.y0 + y * ps.dy) * w + size_t(ps.x0 + x * ps.dx))];
                auto sample = [&](int kk) -> int {                                     // kk-th channel of pixel x, 8 significant bits
                    if (depth == 16) return row[2 * (size_t(x) * ch + kk)];
                    if (depth == 8) return row[size_t(x) * ch + kk];
                    const size_t bit = size_t(x) * depth; return (row[bit >> 3] >> (8 - depth - int(bit & 7))) & maxv;
                };
                if (ctype == 3) { const size_t i = size_t(sample(0)); const unsigned char* p = 3 * i + 2 < plte.size() ? &plte[3 * i] : &plte[0]; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
                else if (ch <= 2) { int g = sample(0); if (depth < 8) g = g * 255 / maxv; o[0] = o[1] = o[2] = (unsigned char)g; }
                else { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); }
            }
        }
    }
    return true;
}
// ---- BMP (BITMAPINFOHEADER and later): 8-bit palettised (uncompressed or RLE8), 24- and 32-bit true colour; bottom-up or top-down rows.
//      Every header field is checked against the file size BEFORE it is used as an offset or a count (a crafted biClrUsed < 0 or a huge
//      header size used to put the palette outside the buffer: ASan repro in tools/fuzz_loader.py's header mutations).
bool load_bmp(const std::vector<unsigned char>& f, int& w, int& h, std::vector<unsigned char>& rgb) {
    auto le16 = [&](size_t p) { return int(f[p]) | (int(f[p + 1]) << 8); };
    auto le32 = [&](size_t p) { return int32_t(uint32_t(f[p]) | (uint32_t(f[p + 1]) << 8) | (uint32_t(f[p + 2]) << 16) | (uint32_t(f[p + 3]) << 24)); };
    if (f.size() < 54 || f[0] != 'B' || f[1] != 'M') return false;
    const size_t off = size_t(uint32_t(le32(10))); const int hdr = le32(14);
    if (hdr < 40 || size_t(hdr) > f.size() - 14 || off > f.size()) return false;
    w = le32(18); int hh = le32(22); const int planes = le16(26), bpp = le16(28), comp = le32(30);
    const bool top_down = hh < 0; h = top_down ? -hh : hh;
    const bool rle8 = comp == 1 && bpp == 8;
    if (w <= 0 || h <= 0 || planes != 1 || (comp != 0 && !(comp == 3 && bpp == 32) && !rle8) || (bpp != 8 && bpp != 24 && bpp != 32) || size_t(w) * size_t(h) > (size_t(1) << 28)) return false;
    if (rle8 && top_down) return false;                                  // (the format forbids it)
    const size_t stride = ((size_t(w) * bpp + 31) / 32) * 4;
    if (!rle8 && stride * size_t(h) > f.size() - off) return false;
    const size_t pal = 14 + size_t(hdr); int ncol = le32(46); if (ncol == 0) ncol = 256;
    if (bpp == 8 && (ncol < 0 || ncol > 256 || size_t(ncol) > (f.size() - pal) / 4)) return false;
    std::vector<unsigned char> idx;                                       // RLE8: the decoded index plane, bottom-up like the file
    if (rle8) {
        idx.assign(size_t(w) * h, 0);
        size_t pos = off; int x = 0, y = 0;
        for (bool end = false; !end;) {
            if (pos + 2 > f.size()) return false;
            const int n = f[pos], v = f[pos + 1]; pos += 2;
            if (n > 0) { for (int i = 0; i < n; i++) { if (x < w && y < h) idx[size_t(y) * w + x] = (unsigned char)v; x++; } }
            else if (v == 0) { x = 0; y++; }                               // end of line
            else if (v == 1) end = true;                                   // end of bitmap
            else if (v == 2) { if (pos + 2 > f.size()) return false; x += f[pos]; y += f[pos + 1]; pos += 2; }   // delta
            else {                                                         // absolute run of v indices, padded to 16 bits
                if (pos + size_t(v) + (v & 1) > f.size()) return false;
                for (int i = 0; i < v; i++) { if (x < w && y < h) idx[size_t(y) * w + x] = f[pos + i]; x++; }
                pos += size_t(v) + (v & 1);
            }
            if (y >= h && !end) { if (pos + 2 <= f.size() && f[pos] == 0 && f[pos + 1] == 1) end = true; else if (y > h) return false; }
        }
    }
    rgb.resize(size_t(w) * h * 3);
    for (int y = 0; y < h; y++) {
        const size_t src_row = size_t(top_down ? y : h - 1 - y);
        const unsigned char* row = rle8 ? &idx[src_row * w] : &f[off + stride * src_row];
        for (int x = 0; x < w; x++) {
            unsigned char* o = &rgb[3 * (size_t(y) * w + x)];
            if (bpp == 8) { const int i = row[x] < ncol ? row[x] : 0; const unsigned char* p = &f[pal + 4 * size_t(i)]; o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
            else { const unsigned char* p = row + size_t(x) * (bpp / 8); o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
        }
    }
    return true;
}
// ---- TGA: true colour (types 2 / 10; 24 or 32 bits), grey (3 / 11; 8 bits) and colour-mapped (1 / 9; 8-bit indices into a 24- or 32-bit
//      map), raw or run-length coded, either orientation.  No signature exists for this format: tried last, only for a plausible header,
//      and nothing is allocated before the pixel count has been checked against the 2^28 cap and against what the file can hold.
bool load_tga(const std::vector<unsigned char>& f, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (f.size() < 18) return false;
    const int idlen = f[0], cmap = f[1], type = f[2], bpp = f[16], desc = f[17];
    const int cm_first = int(f[3]) | (int(f[4]) << 8), cm_len = int(f[5]) | (int(f[6]) << 8), cm_bits = f[7];
    w = int(f[12]) | (int(f[13]) << 8); h = int(f[14]) | (int(f[15]) << 8);
    const bool grey = type == 3 || type == 11, mapped = type == 1 || type == 9, rle = type == 9 || type == 10 || type == 11;
    if (!(type == 1 || type == 2 || type == 3 || type == 9 || type == 10 || type == 11) || w <= 0 || h <= 0 || (desc & 0xC0)) return false;
    if (mapped ? (cmap != 1 || bpp != 8 || (cm_bits != 24 && cm_bits != 32) || cm_len == 0) : (cmap != 0 || (grey ? bpp != 8 : (bpp != 24 && bpp != 32)))) return false;
    const size_t px = size_t(bpp / 8), n = size_t(w) * h;
    if (n > (size_t(1) << 28)) return false;
    const size_t cm_px = size_t(cm_bits / 8), cm_at = 18 + size_t(idlen), cm_bytes = mapped ? size_t(cm_len) * cm_px : 0;
    size_t pos = cm_at + cm_bytes, o = 0;
    if (pos > f.size()) return false;
    if (rle ? n * px > 128 * px * (f.size() - pos) : n * px > f.size() - pos) return false;   // more pixels than the rest of the file can code
    std::vector<unsigned char> raw(n * px);
    if (!rle) std::memcpy(raw.data(), &f[pos], raw.size());
    else while (o < raw.size()) {
        if (pos >= f.size()) return false;
        const int c = f[pos++]; const size_t cnt = size_t(c & 127) + 1;
        if (o + cnt * px > raw.size()) return false;
        if (c & 128) { if (pos + px > f.size()) return false; for (size_t i = 0; i < cnt; i++) { std::memcpy(&raw[o], &f[pos], px); o += px; } pos += px; }
        else { if (pos + cnt * px > f.size()) return false; std::memcpy(&raw[o], &f[pos], cnt * px); o += cnt * px; pos += cnt * px; }
    }
    rgb.resize(n * 3);
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const unsigned char* p = &raw[(size_t(top_down ? y : h - 1 - y) * w + size_t(right_left ? w - 1 - x : x)) * px];
            unsigned char* q = &rgb[3 * (size_t(y) * w + x)];
            if (mapped) { const int i = int(p[0]) - cm_first; const unsigned char* m = &f[cm_at + cm_px * size_t(i >= 0 && i < cm_len ? i : 0)]; q[0] = m[2]; q[1] = m[1]; q[2] = m[0]; }
            else if (grey) q[0] = q[1] = q[2] = p[0];
            else { q[0] = p[2]; q[1] = p[1]; q[2] = p[0]; }
        }
    return true;
}
bool load_ppm(const std::vector<unsigned char>& f, int& w, int& h, std::vector<unsigned char>& rgb) {
    if (f.size() < 11 || f[0] != 'P' || f[1] != '6') return false;
    size_t pos = 2; int vals[3], n = 0;
    while (n < 3 && pos < f.size()) {
        while (pos < f.size() && std::isspace(f[pos])) pos++;
        if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
        int v = 0; bool any = false;
        while (pos < f.size() && std::isdigit(f[pos])) { v = v * 10 + (f[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[n++] = v;
    }
    pos++;   // single whitespace after maxval
    w = vals[0]; h = vals[1];
    if (vals[2] != 255 || pos + size_t(w) * h * 3 > f.size()) return false;
    rgb.assign(f.begin() + pos, f.begin() + pos + size_t(w) * h * 3);
    return true;
}
// ---- Radiance RGBE (.hdr): the one input stbi_loadf returns LINEAR (model.cpp:11; every LDR format goes through (c/255)^2.2).  Header
//      "#?RADIANCE" or "#?RGBE", a FORMAT=32-bit_rle_rgbe line, a blank line, "-Y h +X w"; scanlines either flat RGBE quadruples or --
//      for 8 <= w < 32768 -- the "new" run-length form (2, 2, w >> 8, w & 255, then four component planes of runs / dumps).  A texel is
//      mantissa * 2^(e - 136), or 0 where e == 0.  Like stb, a file whose first scanline does not start with the RLE marker is read as flat
//      quadruples from that point on.
bool load_hdr(const std::vector<unsigned char>& f, int& w, int& h, std::vector<float>& rgbf) {
    size_t pos = 0;
    auto line = [&](std::string& out) { out.clear(); while (pos < f.size() && f[pos] != '\n') { if (out.size() < 1023) out.push_back(char(f[pos])); pos++; } if (pos < f.size()) pos++; return true; };
    std::string ln;
    line(ln);
    if (ln != "#?RADIANCE" && ln != "#?RGBE") return false;
    bool fmt = false;
    for (;;) { if (pos >= f.size()) return false; line(ln); if (ln.empty()) break; if (ln == "FORMAT=32-bit_rle_rgbe") fmt = true; }
    if (!fmt) return false;
    line(ln);
    if (ln.compare(0, 3, "-Y ") != 0) return false;
    char* e = nullptr; const long hh = std::strtol(ln.c_str() + 3, &e, 10);
    while (*e == ' ') e++;
    if (std::strncmp(e, "+X ", 3) != 0) return false;
    const long ww = std::strtol(e + 3, nullptr, 10);
    if (ww <= 0 || hh <= 0 || ww > (1 << 24) || hh > (1 << 24) || size_t(ww) * size_t(hh) > (size_t(1) << 28)) return false;
    w = int(ww); h = int(hh);
    if (size_t(w) * h > (f.size() - pos) * 128) return false;            // more texels than the rest of the file can code
    rgbf.assign(size_t(w) * h * 3, 0.0f);
    auto put = [&](size_t texel, const unsigned char* q) {
        if (q[3] == 0) return;
        const float s = std::ldexp(1.0f, int(q[3]) - 136);
        rgbf[3 * texel] = q[0] * s; rgbf[3 * texel + 1] = q[1] * s; rgbf[3 * texel + 2] = q[2] * s;
    };
    auto flat_from = [&](size_t texel) {                                  // the rest of the image as plain quadruples
        for (; texel < size_t(w) * h; texel++) { if (pos + 4 > f.size()) return false; put(texel, &f[pos]); pos += 4; }
        return true;
    };
    if (w < 8 || w >= 32768) return flat_from(0);
    std::vector<unsigned char> scan(size_t(w) * 4);
    for (int y = 0; y < h; y++) {
        if (pos + 4 > f.size()) return false;
        if (f[pos] != 2 || f[pos + 1] != 2 || (f[pos + 2] & 0x80)) {
            // not run-length coded: stb takes these four bytes as texel 0 and reads everything after them as flat data (whatever the row)
            std::fill(rgbf.begin(), rgbf.end(), 0.0f);
            put(0, &f[pos]); pos += 4;
            return flat_from(1);
        }
        if (((int(f[pos + 2]) << 8) | f[pos + 3]) != w) return false;
        pos += 4;
        for (int k = 0; k < 4; k++)
            for (int x = 0; x < w;) {
                if (pos >= f.size()) return false;
                int cnt = f[pos++];
                if (cnt > 128) { cnt -= 128; if (cnt > w - x || pos >= f.size()) return false; const unsigned char v = f[pos++]; for (int z = 0; z < cnt; z++) scan[size_t(x++) * 4 + k] = v; }
                else { if (cnt == 0 || cnt > w - x || pos + size_t(cnt) > f.size()) return false; for (int z = 0; z < cnt; z++) scan[size_t(x++) * 4 + k] = f[pos++]; }
            }
        for (int x = 0; x < w; x++) put(size_t(y) * w + x, &scan[size_t(x) * 4]);
    }
    return true;
}
}  // namespace

bool load_image_rgb8(const std::string& filename, int& w, int& h, std::vector<unsigned char>& rgb) {
    const std::vector<unsigned char> bytes = read_file<std::vector<unsigned char>>(filename);
    return !bytes.empty() && (load_png(bytes, w, h, rgb) || load_jpeg(bytes, w, h, rgb) || load_ppm(bytes, w, h, rgb) || load_bmp(bytes, w, h, rgb) || load_tga(bytes, w, h, rgb));
}
bool load_image_hdr(const std::string& filename, int& w, int& h, std::vector<float>& rgbf) {
    const std::vector<unsigned char> bytes = read_file<std::vector<unsigned char>>(filename);
    return !bytes.empty() && load_hdr(bytes, w, h, rgbf);
}

Texture::Texture(const std::string& filename) {
    std::vector<unsigned char> rgb; int w = 0, h = 0;
    {   // Radiance .hdr: stbi_loadf hands its floats over as they are -- linear, no gamma (stb_image.h: stbi__loadf_main -> stbi__hdr_load)
        std::vector<float> lin;
        if (load_image_hdr(filename, w, h, lin)) {
            image_w = w; image_h = h; image_color.resize(size_t(w) * h);
            for (size_t i = 0; i < image_color.size(); i++) { image_color[i].x = lin[3 * i]; image_color[i].y = lin[3 * i + 1]; image_color[i].z = lin[3 * i + 2]; }
            return;
        }
    }
    if (!load_image_rgb8(filename, w, h, rgb)) {
        std::cerr << "Error: cannot decode texture (PNG, JPEG, binary PPM, BMP, TGA or Radiance HDR expected): " << filename << std::endl;
        ok = false; image_color.push_back(Color3f{0.5f, 0.5f, 0.5f}); return;
    }
    image_w = w; image_h = h;
    image_color.resize(size_t(w) * h);
    for (size_t i = 0; i < image_color.size(); i++) {   // stbi_loadf's LDR->float: pow(c/255, 2.2) (stb_image.h:1553,1849)
        image_color[i].x = std::pow(rgb[3 * i] / 255.0f, 2.2f);
        image_color[i].y = std::pow(rgb[3 * i + 1] / 255.0f, 2.2f);
        image_color[i].z = std::pow(rgb[3 * i + 2] / 255.0f, 2.2f);
    }
}
Texture::Texture(Color3f c) { image_color.push_back(c); }
Color3f Texture::get_color(const dvec2& uv) const {
    if (image_color.size() == 1) return image_color[0];
    auto clamp01 = [](float d) -> double { if (d > 0.999f) return 0.999; if (d < 0.0f) return 0.0; return d; };
    double u = clamp01(float(uv.x - std::floor(uv.x))), v = clamp01(float(uv.y - std::floor(uv.y)));
    return image_color.at(size_t(int(v * image_h)) * image_w + int(u * image_w));
}

// ---- OBJ text.  One line = one record, classified exactly like the reference's chain of `starts_with` tests (model.cpp:62-155).  The
// records of a multi-million-triangle file are parsed by all host cores: the text is cut into chunks at line ends, every chunk is
// parsed on its own (numbers through std::from_chars when a token is plain decimal -- [-]digits[.digits][e[+-]digits] -- and through
// the same istringstream extraction as before for anything else, so odd tokens keep their stream semantics: "+1", "1.5abc", "nan",
// overflow ...), and the chunks are stitched together in file order.  `mtllib` and `usemtl` lines are order-dependent (a usemtl
// resolves against the file's materials, model.cpp:131-136): chunks only record them, the stitching replays them in order.
namespace {
// regex_search(line, "<keyword>\\s+(\\S+)") of model.cpp:67 / :134 without <regex>: the leftmost place where the keyword is followed by at least
// one white-space character and then a non-empty run of non-white-space characters; returns that run, or false.  ("usemtl" alone, or
// "usemtlwood", matches nothing and leaves the current material as it is; "u usemtl wood" does switch.)
bool keyword_arg(const std::string& line, const char* kw, std::string& arg) {
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
    const size_t kn = std::strlen(kw);
    for (size_t at = line.find(kw); at != std::string::npos; at = line.find(kw, at + 1)) {
        size_t q = at + kn;
        if (q >= line.size() || !is_space(line[q])) continue;
        while (q < line.size() && is_space(line[q])) q++;
        size_t e = q;
        while (e < line.size() && !is_space(line[e])) e++;
        if (e > q) { arg = line.substr(q, e - q); return true; }
    }
    return false;
}
struct ObjEvent { bool is_mtllib; std::string text; };          // the whole line (it starts with 'm' or with 'u')
struct ObjChunk {
    std::vector<dvec3> vertex, normal;
    std::vector<dvec2> texture;
    std::vector<imat3x4> face;
    std::vector<int> face_event;                                 // per face: index into `events` of the usemtl in force, -1 = inherited
    std::vector<ObjEvent> events;
    int last_usemtl = -1;
};
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; }
// plain decimal token at p (after optional blanks), fully consumed and followed by a blank or the end of the line
inline bool fast_double(const char*& p, const char* e, double& out) {
    while (p < e && is_space(*p)) p++;
    const char* t = p;
    while (t < e && !is_space(*t)) t++;
    if (t == p || *p == '+') return false;
    for (const char* c = p; c < t; c++) if (!((*c >= '0' && *c <= '9') || *c == '.' || *c == '-' || *c == '+' || *c == 'e' || *c == 'E')) return false;
    const auto r = std::from_chars(p, t, out);
    if (r.ec != std::errc() || r.ptr != t) return false;
    p = t;
    return true;
}
inline bool fast_int(const char*& p, const char* e, int& out) {
    if (p < e && *p == '+') return false;
    const auto r = std::from_chars(p, e, out);
    if (r.ec != std::errc() || r.ptr == p) return false;
    p = r.ptr;
    return true;
}
void slow_vec(const std::string& line, size_t skip, int n, double* out) {       // the stream extraction the loader always used
    std::istringstream ss(rest(line, skip));
    for (int i = 0; i < n; i++) ss >> out[i];
}
bool slow_face(const std::string& line, bool reference_index_order, imat3x4& f) {
    std::istringstream ss(rest(line, 2)); bool good = true;
    for (int i = 0; i < 3 && good; i++) {
        int a = 0, b = 0, c = 0; char s1 = 0, s2 = 0;
        ss >> a >> s1 >> b >> s2 >> c;
        good = bool(ss) && s1 == '/' && s2 == '/';
        f[i][0] = a - 1;
        if (reference_index_order) { f[i][1] = b - 1; f[i][2] = c - 1; }   // reference: second = normal, third = texcoord
        else { f[i][1] = c - 1; f[i][2] = b - 1; }                          // Wavefront: v / vt / vn
        f[i][3] = 0;
    }
    return good;
}
void parse_obj_chunk(const char* b, const char* e, bool reference_index_order, ObjChunk& out) {
    static const bool slow_only = std::getenv("MCPT_LOADER_SLOW") != nullptr;   // developer knob: every record through the stream extraction (tests compare)
    const char* p = b;
    while (p < e) {
        const char* le = static_cast<const char*>(std::memchr(p, '\n', size_t(e - p)));
        const char* next = le ? le + 1 : e;
        if (!le) le = e;
        if (le > p && le[-1] == '\r') le--;
        const size_t n = size_t(le - p);
        auto begins = [&](const char* t, size_t tn) { return n >= tn && std::memcmp(p, t, tn) == 0; };
        if (n && p[0] == 'm') out.events.push_back({true, std::string(p, le)});          // model.cpp:64: any line that starts with 'm'; the name is searched for later
        else if (begins("v ", 2)) {
            dvec3 v; const char* q = p + 2;
            if (slow_only || !(fast_double(q, le, v.x) && fast_double(q, le, v.y) && fast_double(q, le, v.z))) { v = dvec3(); slow_vec(std::string(p, le), 2, 3, &v.x); }
            out.vertex.push_back(v);
        } else if (begins("vn", 2)) {
            dvec3 v; const char* q = p + std::min<size_t>(3, n);
            if (slow_only || !(fast_double(q, le, v.x) && fast_double(q, le, v.y) && fast_double(q, le, v.z))) { v = dvec3(); slow_vec(std::string(p, le), 3, 3, &v.x); }
            out.normal.push_back(v);
        } else if (begins("vt", 2)) {
            dvec2 v; const char* q = p + std::min<size_t>(3, n);
            if (slow_only || !(fast_double(q, le, v.x) && fast_double(q, le, v.y))) { v = dvec2(); slow_vec(std::string(p, le), 3, 2, &v.x); }
            out.texture.push_back(v);
        } else if (n && p[0] == 'u') { out.last_usemtl = int(out.events.size()); out.events.push_back({false, std::string(p, le)}); }   // model.cpp:88: any line that starts with 'u'
        else if (begins("f ", 2)) {
            imat3x4 f; const char* q = p + 2; bool fast = !slow_only;
            for (int i = 0; i < 3 && fast; i++) {
                while (q < le && is_space(*q)) q++;
                int a = 0, b2 = 0, c = 0;
                fast = fast_int(q, le, a) && q < le && *q == '/' && fast_int(++q, le, b2) && q < le && *q == '/' && fast_int(++q, le, c) && (q == le || is_space(*q));
                f[i][0] = a - 1;
                if (reference_index_order) { f[i][1] = b2 - 1; f[i][2] = c - 1; } else { f[i][1] = c - 1; f[i][2] = b2 - 1; }
                f[i][3] = 0;
            }
            if (!fast) fast = slow_face(std::string(p, le), reference_index_order, f);
            if (fast) { out.face.push_back(f); out.face_event.push_back(out.last_usemtl); }
        }
        p = next;
    }
}
}  // namespace

Model::Model(const std::string& filename, bool reference_index_order) {
    std::cout << "[Model] " << filename << std::endl;
    std::error_code ec;
    if (!std::filesystem::is_regular_file(filename, ec) || ec) { std::cerr << "Error: Cannot open OBJ file: " << filename << std::endl; return; }
    const std::string text = read_file<std::string>(filename);
    const std::string parent = dir_of(filename);
    // ---- cut at line ends, parse the chunks side by side
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
    const size_t n_chunks = std::max<size_t>(1, std::min<size_t>(std::min(hw, 32u), text.size() / (size_t(1) << 20)));
    std::vector<size_t> cut(n_chunks + 1, text.size());
    cut[0] = 0;
    for (size_t k = 1; k < n_chunks; k++) {
        size_t at = std::max(cut[k - 1], text.size() * k / n_chunks);
        const size_t nl = text.find('\n', at);
        cut[k] = nl == std::string::npos ? text.size() : nl + 1;
    }
    std::vector<ObjChunk> chunks(n_chunks);
    auto work = [&](size_t k) { parse_obj_chunk(text.data() + cut[k], text.data() + cut[k + 1], reference_index_order, chunks[k]); };
    {
        std::vector<std::thread> pool;
        std::vector<char> started(n_chunks, 0);
        for (size_t k = 1; k < n_chunks; k++) {
            try { pool.emplace_back(work, k); started[k] = 1; } catch (const std::system_error&) {}    // no thread to be had: parsed below, in line
        }
        work(0);
        for (size_t k = 1; k < n_chunks; k++) if (!started[k]) work(k);
        for (auto& t : pool) t.join();
    }
    // ---- stitch in file order; replay mtllib / usemtl
    size_t nv = 0, nn = 0, nt = 0, nf = 0;
    for (const ObjChunk& c : chunks) { nv += c.vertex.size(); nn += c.normal.size(); nt += c.texture.size(); nf += c.face.size(); }
    vertex.reserve(nv); normal.reserve(nn); texture.reserve(nt); face.reserve(nf);
    // the reference reads every line first and resolves `usemtl` afterwards (model.cpp:62-92 then :125-136): a usemtl names a material
    // of ANY mtllib line of the file, also a later one; an unknown name is material 0 (`material_map[name]` default-inserts 0)
    for (const ObjChunk& c : chunks)
        for (const ObjEvent& ev : c.events) {
            std::string name;
            if (!ev.is_mtllib || !keyword_arg(ev.text, "mtllib", name)) continue;
            if (name.size() > 3) {
                std::string xml = name; xml.replace(xml.size() - 3, 3, "xml");
                loadCameraFromXML(parent + "/" + xml);              // camera first: radiance is attached while materials load (model.cpp:71-72)
                load_material(parent + "/" + name);
            }
        }
    int cur_mtl = 0;
    for (ObjChunk& c : chunks) {
        std::vector<int> resolved(c.events.size(), 0);
        const int carried = cur_mtl;
        for (size_t i = 0; i < c.events.size(); i++) {
            if (c.events[i].is_mtllib) continue;
            std::string name;
            if (keyword_arg(c.events[i].text, "usemtl", name)) {            // no match: the material in force stays (model.cpp:134)
                auto it = material_map.find(name);
                cur_mtl = it == material_map.end() ? 0 : it->second;
            }
            resolved[i] = cur_mtl;
        }
        vertex.insert(vertex.end(), c.vertex.begin(), c.vertex.end());
        normal.insert(normal.end(), c.normal.begin(), c.normal.end());
        texture.insert(texture.end(), c.texture.begin(), c.texture.end());
        for (size_t j = 0; j < c.face.size(); j++) {
            imat3x4 f = c.face[j];
            const int m = c.face_event[j] < 0 ? carried : resolved[size_t(c.face_event[j])];
            f[0][3] = f[1][3] = f[2][3] = m;
            face.push_back(f);
        }
        c = ObjChunk();                                             // free as we go
    }
    ok = !face.empty() && !materials.empty() && camerainfo.width > 0 && camerainfo.height > 0;
}

void Model::load_material(const std::string& filename) {
    std::ifstream fs(filename);
    if (!fs.is_open()) { std::cerr << "Error: Cannot open mtl file: " << filename << std::endl; return; }
    const std::string parent = dir_of(filename);
    std::string line;
    while (std::getline(fs, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.find('#') != std::string::npos) continue;
        std::istringstream ss(line); std::string key; ss >> key;
        if (key.empty()) continue;
        if (key == "newmtl") {
            std::string name; ss >> name;
            material_map[name] = int(materials.size());
            materials.push_back(Material());
            auto it = camerainfo.lightinfo.find(name);
            if (it != camerainfo.lightinfo.end()) materials.back().radiance = it->second;
            continue;
        }
        if (materials.empty()) continue;
        Material& m = materials.back();
        if (key == "Kd") { float r = 0, g = 0, b = 0; ss >> r >> g >> b; m.Map_Kd = std::make_shared<Texture>(Color3f{r, g, b}); }
        else if (key == "Ks") ss >> m.Ks.x >> m.Ks.y >> m.Ks.z;
        else if (key == "Tr") ss >> m.Tr.x >> m.Tr.y >> m.Tr.z;
        else if (key == "Ns") ss >> m.Ns;
        else if (key == "Ni") ss >> m.Ni;
        else if (key == "map_Kd") { std::string name; ss >> name; m.Map_Kd = std::make_shared<Texture>(parent + "/" + name); }
    }
    for (Material& m : materials)                      // the reference dereferences a null Map_Kd when Kd is missing (model.h:38)
        if (!m.Map_Kd) m.Map_Kd = std::make_shared<Texture>(Color3f{0.f, 0.f, 0.f});
}

namespace {
bool xml_attr(const std::string& tag, const char* name, std::string& out) {
    const std::string key = std::string(name) + "=\"";
    size_t p = 0;
    while ((p = tag.find(key, p)) != std::string::npos) {
        if (p == 0 || std::isspace((unsigned char)tag[p - 1])) { size_t e = tag.find('"', p + key.size()); if (e == std::string::npos) return false; out = tag.substr(p + key.size(), e - p - key.size()); return true; }
        p += key.size();
    }
    return false;
}
double xml_num(const std::string& tag, const char* name) { std::string s; return xml_attr(tag, name, s) ? std::atof(s.c_str()) : 0.0; }
}  // namespace

void Model::loadCameraFromXML(const std::string& filename) {
    const std::string text = read_file<std::string>(filename);
    if (text.empty()) { std::cerr << "Error: Failed to load XML file: " << filename << std::endl; return; }
    size_t p = 0; bool have_camera = false;
    while ((p = text.find('<', p)) != std::string::npos) {
        size_t e = text.find('>', p); if (e == std::string::npos) break;
        std::string tag = text.substr(p + 1, e - p - 1); p = e + 1;
        auto vec = [&](dvec3& v) { v.x = xml_num(tag, "x"); v.y = xml_num(tag, "y"); v.z = xml_num(tag, "z"); };
        if (starts(tag, "camera")) { camerainfo.width = int(xml_num(tag, "width")); camerainfo.height = int(xml_num(tag, "height")); camerainfo.fovy = xml_num(tag, "fovy"); have_camera = true; }
        else if (starts(tag, "eye")) vec(camerainfo.eye);
        else if (starts(tag, "lookat")) vec(camerainfo.lookat);
        else if (starts(tag, "up")) vec(camerainfo.up);
        else if (starts(tag, "light")) {
            std::string name, rad;
            if (xml_attr(tag, "mtlname", name) && xml_attr(tag, "radiance", rad)) {
                size_t c1 = rad.find(','), c2 = rad.find(',', c1 + 1);
                if (c1 == std::string::npos || c2 == std::string::npos) { std::cerr << "Error: Invalid radiance format in <light> node." << std::endl; continue; }
                dvec3 r; r.x = std::atof(rad.substr(0, c1).c_str()); r.y = std::atof(rad.substr(c1 + 1, c2 - c1 - 1).c_str()); r.z = std::atof(rad.substr(c2 + 1).c_str());
                camerainfo.lightinfo[name] = r;
            }
        }
    }
    if (!have_camera) std::cerr << "Error: No <camera> node found in XML file." << std::endl;
}
