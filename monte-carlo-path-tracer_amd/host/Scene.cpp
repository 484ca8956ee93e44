#include "Scene.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <zlib.h>

Scene::Scene(int width, int heigh) : w(width), h(heigh) {
    m_Pixels = std::unique_ptr<Pixels[]>(new Pixels[size_t(w) * h]);
    m_ColorsUchar = std::unique_ptr<std::vector<Color3b>>(new std::vector<Color3b>(size_t(w) * h));
}
void Scene::set_Pixel(const Point2i& location, Color3f& color) {
    const int idx = location.y * w + location.x;
    if (color.x != color.x) color.x = 0.0f;
    if (color.y != color.y) color.y = 0.0f;
    if (color.z != color.z) color.z = 0.0f;
    m_Pixels[idx].color.x += color.x; m_Pixels[idx].color.y += color.y; m_Pixels[idx].color.z += color.z;
    m_Pixels[idx].spp += 1.0f;
    m_host_samples = true;
}
void Scene::add_film(const float* f) {
    m_host_samples = true;
    for (size_t i = 0; i < size_t(w) * h; i++) {
        m_Pixels[i].color.x += f[4 * i]; m_Pixels[i].color.y += f[4 * i + 1]; m_Pixels[i].color.z += f[4 * i + 2]; m_Pixels[i].spp += f[4 * i + 3];
    }
}
Scene::~Scene() { if (m_source) m_source->scene_gone(*this); }
void Scene::attach(FilmSource* source) {
    if (m_source == source) return;
    sync();
    if (m_source) m_source->displaced(*this);         // (two Renders sharing one Scene: the first must not call detach() on a Scene that died meanwhile)
    m_source = source;
}
void Scene::detach(FilmSource* source) { if (m_source == source) m_source = nullptr; }
void Scene::sync() { if (m_source) m_source->flush_into(*this); }
const Color3b* Scene::getPixelsColor() {
    // The reference's loop calls this after EVERY render(scene) (main.cpp:26-33).  While the whole film is on the device -- nothing was ever
    // written or folded into m_Pixels -- the device tonemaps it where it lies (mcpt_tonemap_map: one small kernel + a 3-byte-per-pixel copy
    // into pinned memory) instead of 16 B per pixel coming back, a host add and a host pow per channel.  Same arithmetic, <= 1 LSB
    // (tests: test_tonemap_matches_reference_film, test_facade_getPixelsColor_runs_on_the_device).
    if (m_source && !m_host_samples) { if (const Color3b* px = m_source->tonemapped(*this)) return px; }
    sync();
    for (size_t i = 0; i < size_t(w) * h; i++) {
        const float c[3] = {m_Pixels[i].color.x / m_Pixels[i].spp, m_Pixels[i].color.y / m_Pixels[i].spp, m_Pixels[i].color.z / m_Pixels[i].spp};
        uint8_t o[3];
        for (int k = 0; k < 3; k++) { float m = std::min(std::max(c[k], 0.f), 1.f); o[k] = uint8_t(std::pow(m, 0.5f) * 255.99f); }
        (*m_ColorsUchar)[i] = Color3b{o[0], o[1], o[2]};
    }
    return m_ColorsUchar->data();
}
void Scene::save_image(int frame, std::string filename) {
    const std::string file = filename + std::to_string(frame) + ".png";
    std::vector<uint8_t> flipped(size_t(w) * h * 3);
    const Color3b* px = getPixelsColor();
    for (int y = 0; y < h; y++) std::memcpy(&flipped[size_t(h - 1 - y) * w * 3], &px[size_t(y) * w], size_t(w) * 3);
    if (write_png_rgb8(file, w, h, flipped.data())) std::cout << "Image saved successfully: " << file << std::endl;
    else std::cerr << "Failed to save image: " << file << std::endl;
}

bool write_png_rgb8(const std::string& path, int w, int h, const uint8_t* rgb) {
    std::vector<uint8_t> raw((size_t(w) * 3 + 1) * h);
    for (int y = 0; y < h; y++) { raw[size_t(y) * (w * 3 + 1)] = 0; std::memcpy(&raw[size_t(y) * (w * 3 + 1) + 1], rgb + size_t(y) * w * 3, size_t(w) * 3); }
    uLongf clen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), uLong(raw.size()), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8] = {uint8_t(len >> 24), uint8_t(len >> 16), uint8_t(len >> 8), uint8_t(len), uint8_t(type[0]), uint8_t(type[1]), uint8_t(type[2]), uint8_t(type[3])};
        std::fwrite(hdr, 1, 8, f); if (len) std::fwrite(data, 1, len, f);
        uLong crc = crc32(0L, hdr + 4, 4); if (len) crc = crc32(crc, data, len);
        uint8_t c[4] = {uint8_t(crc >> 24), uint8_t(crc >> 16), uint8_t(crc >> 8), uint8_t(crc)}; std::fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {uint8_t(w >> 24), uint8_t(w >> 16), uint8_t(w >> 8), uint8_t(w), uint8_t(h >> 24), uint8_t(h >> 16), uint8_t(h >> 8), uint8_t(h), 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), uint32_t(clen)); chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}
