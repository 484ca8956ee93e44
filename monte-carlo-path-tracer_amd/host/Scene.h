// Film with the reference's surface (src/Scene.h:7-27): sum + count per pixel, NaN scrub, tonemap, PNG.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "Model.h"

struct Point2i { int x, y; };
struct Color3b { uint8_t x, y, z; };
struct Pixels { Color3f color; float spp = 0.f; };    // Scene.h:7-12 -- also the layout mcpt_read_accum fills (16 B / pixel)

class Scene {
public:
    Scene(int width, int heigh);
    void set_Pixel(const Point2i& location, Color3f& color);          // Scene.cpp:12-21
    const Color3b* getPixelsColor();                                   // Scene.cpp:23-33
    void save_image(int frame, std::string filename);                 // Scene.cpp:35-53 (writes ./<filename><frame>.png)
    // extensions used by Render: bulk accumulate of a device film (same {sum rgb, count} records)
    void add_film(const float* rgba_sum_count);
    Pixels* pixels() { return m_Pixels.get(); }
    int width() const { return w; }
    int height() const { return h; }
private:
    int w, h;
    std::unique_ptr<Pixels[]> m_Pixels;
    std::unique_ptr<std::vector<Color3b>> m_ColorsUchar;
};
bool write_png_rgb8(const std::string& path, int w, int h, const uint8_t* rgb);
