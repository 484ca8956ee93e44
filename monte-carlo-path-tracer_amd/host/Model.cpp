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
#include <sstream>
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

// ---- PNG via zlib: every colour type (grey, RGB, palette, grey+alpha, RGBA), bit depths 1-16, non-interlaced; returns RGB bytes
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
    if (ch == 0 || !depth_ok || interlace != 0 || w <= 0 || h <= 0 || size_t(w) * size_t(h) > (size_t(1) << 28) || (ctype == 3 && plte.size() < 3)) return false;
    const size_t bpp = std::max<size_t>(1, size_t(ch) * depth / 8);            // filter distance in bytes
    const size_t stride = (size_t(w) * ch * depth + 7) / 8;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf out = uLongf(raw.size());
    if (uncompress(raw.data(), &out, idat.data(), uLong(idat.size())) != Z_OK || out != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (int y = 0; y < h; y++) {
        const unsigned char ft = raw[y * (stride + 1)]; const unsigned char* s = &raw[y * (stride + 1) + 1];
        unsigned char* d = &img[y * stride]; const unsigned char* up = y ? &img[(y - 1) * stride] : nullptr;
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
    rgb.resize(size_t(w) * h * 3);
    const int maxv = (1 << (depth < 8 ? depth : 8)) - 1;
    for (int y = 0; y < h; y++) {
        const unsigned char* row = &img[size_t(y) * stride];
        for (int x = 0; x < w; x++) {
            unsigned char* o = &rgb[3 * (size_t(y) * w + x)];
            auto sample = [&](int k) -> int {                                      // k-th channel of pixel x, 8 significant bits
                if (depth == 16) return row[2 * (size_t(x) * ch + k)];
                if (depth == 8) return row[size_t(x) * ch + k];
                const size_t bit = size_t(x) * depth; return (row[bit >> 3] >> (8 - depth - int(bit & 7))) & maxv;
            };
            if (ctype == 3) { const size_t i = size_t(sample(0)); const unsigned char* p = 3 * i + 2 < plte.size() ? &plte[3 * i] : &plte[0]; o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
            else if (ch <= 2) { int g = sample(0); if (depth < 8) g = g * 255 / maxv; o[0] = o[1] = o[2] = (unsigned char)g; }
            else { o[0] = (unsigned char)sample(0); o[1] = (unsigned char)sample(1); o[2] = (unsigned char)sample(2); }
        }
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
}  // namespace

bool load_image_rgb8(const std::string& filename, int& w, int& h, std::vector<unsigned char>& rgb) {
    const std::vector<unsigned char> bytes = read_file<std::vector<unsigned char>>(filename);
    return !bytes.empty() && (load_png(bytes, w, h, rgb) || load_jpeg(bytes, w, h, rgb) || load_ppm(bytes, w, h, rgb));
}

Texture::Texture(const std::string& filename) {
    std::vector<unsigned char> rgb; int w = 0, h = 0;
    if (!load_image_rgb8(filename, w, h, rgb)) {
        std::cerr << "Error: cannot decode texture (non-interlaced PNG, baseline JPEG or binary PPM expected): " << filename << std::endl;
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

Model::Model(const std::string& filename, bool reference_index_order) {
    std::cout << "[Model] " << filename << std::endl;
    std::ifstream file(filename);
    if (!file.is_open()) { std::cerr << "Error: Cannot open OBJ file: " << filename << std::endl; return; }
    const std::string parent = dir_of(filename);
    std::string line; int cur_mtl = 0;
    while (std::getline(file, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (starts(line, "mtllib")) {
            std::istringstream ss(rest(line, 6)); std::string name; ss >> name;
            if (name.size() > 3) {
                std::string xml = name; xml.replace(xml.size() - 3, 3, "xml");
                loadCameraFromXML(parent + "/" + xml);          // camera first: radiance is attached while materials load (model.cpp:71-72)
                load_material(parent + "/" + name);
            }
        } else if (starts(line, "v ")) { dvec3 v; std::istringstream ss(rest(line, 2)); ss >> v.x >> v.y >> v.z; vertex.push_back(v); }
        else if (starts(line, "vn")) { dvec3 n; std::istringstream ss(rest(line, 3)); ss >> n.x >> n.y >> n.z; normal.push_back(n); }
        else if (starts(line, "vt")) { dvec2 t; std::istringstream ss(rest(line, 3)); ss >> t.x >> t.y; texture.push_back(t); }
        else if (starts(line, "usemtl")) { std::istringstream ss(rest(line, 6)); std::string name; ss >> name; auto it = material_map.find(name); cur_mtl = it == material_map.end() ? 0 : it->second; }
        else if (starts(line, "f ")) {
            std::istringstream ss(rest(line, 2)); imat3x4 f; bool good = true;
            for (int i = 0; i < 3 && good; i++) {
                int a = 0, b = 0, c = 0; char s1 = 0, s2 = 0;
                ss >> a >> s1 >> b >> s2 >> c;
                good = bool(ss) && s1 == '/' && s2 == '/';
                f[i][0] = a - 1;
                if (reference_index_order) { f[i][1] = b - 1; f[i][2] = c - 1; }   // reference: second = normal, third = texcoord
                else { f[i][1] = c - 1; f[i][2] = b - 1; }                          // Wavefront: v / vt / vn
                f[i][3] = cur_mtl;
            }
            if (good) face.push_back(f);
        }
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
