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

// Whoever holds samples of a Scene's film somewhere else (Render: in HBM) until the Scene is looked at.
class FilmSource {
public:
    virtual ~FilmSource() {}
    virtual void flush_into(class Scene& scene) = 0;     // add the held samples to `scene` (Scene::add_film) and forget them
    virtual void scene_gone(class Scene& scene) = 0;     // `scene` is being destroyed: drop what was held for it
    virtual void displaced(class Scene& scene) = 0;      // another source took `scene` over (this one was flushed first): forget the scene --
                                                         // it may be destroyed without this source ever hearing of it again
    // Scene::getPixelsColor of the held samples ALONE, without moving them (the Scene calls this only while its own host part is empty):
    // a pointer to width * height tonemapped pixels that stays valid until the next call, or nullptr if the source cannot do that
    virtual const struct Color3b* tonemapped(class Scene& scene) { (void)scene; return nullptr; }
};

class Scene {
public:
    Scene(int width, int heigh);
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    void set_Pixel(const Point2i& location, Color3f& color);          // Scene.cpp:12-21
    const Color3b* getPixelsColor();                                   // Scene.cpp:23-33
    void save_image(int frame, std::string filename);                 // Scene.cpp:35-53 (writes ./<filename><frame>.png)
    // extensions used by Render: bulk accumulate of a device film (same {sum rgb, count} records)
    void add_film(const float* rgba_sum_count);
    // The film is the sum of what is in m_Pixels and what the attached source still holds on the device: the reference's loop calls
    // render(scene) once per sample (main.cpp:28-30), and reading 16 B per pixel back after every call would cost more than the call.
    // Every reader below folds the device part in first.
    void attach(FilmSource* source);                                  // at most one source at a time: attaching another flushes the first
    void detach(FilmSource* source);                                  // (no flush: the source is going away and has flushed itself)
    FilmSource* source() const { return m_source; }
    void sync();                                                      // fold the device part in now
    Pixels* pixels() { sync(); return m_Pixels.get(); }
    int width() const { return w; }
    int height() const { return h; }
private:
    int w, h;
    std::unique_ptr<Pixels[]> m_Pixels;
    std::unique_ptr<std::vector<Color3b>> m_ColorsUchar;
    FilmSource* m_source = nullptr;
    bool m_host_samples = false;                                       // m_Pixels holds something (set_Pixel / add_film): the film is host part + device part
};
bool write_png_rgb8(const std::string& path, int w, int h, const uint8_t* rgb);
