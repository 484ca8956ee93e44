// Host-side scene model with the reference's public surface (src/model.h:21-67): Texture, Material, CameraInfo, Model.
// Written from scratch: no glm / pugixml / stb / regex.  `Model(filename)` reads the same three files the reference does
// (X.obj + the .mtl named by its mtllib line + the .xml with the same stem).
#pragma once
#include <array>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct dvec3 { double x = 0, y = 0, z = 0; };
struct dvec2 { double x = 0, y = 0; };
struct Color3f { float x = 0, y = 0, z = 0; };
typedef std::array<std::array<int, 4>, 3> imat3x4;   // per corner {v_idx, vn_idx, vt_idx, material_idx} (model.h:57)

// 8-bit RGB pixels of a PNG, JPEG, binary PPM, BMP or TGA file, row 0 = top of the image
bool load_image_rgb8(const std::string& filename, int& w, int& h, std::vector<unsigned char>& rgb);
// linear float RGB texels of a Radiance RGBE (.hdr) file, row 0 = top of the image
bool load_image_hdr(const std::string& filename, int& w, int& h, std::vector<float>& rgbf);

class Texture {                                       // model.h:21-30
public:
    explicit Texture(const std::string& filename);   // LDR formats: texels -> (c/255)^2.2 like stbi_loadf; Radiance .hdr: linear, as stbi_loadf returns it
    explicit Texture(Color3f c);                      // constant Kd
    std::vector<Color3f> image_color;
    int image_w = 1, image_h = 1;
    Color3f get_color(const dvec2& uv) const;         // model.cpp:30-41 (host copy; the device has its own)
    bool ok = true;
};

struct Material {                                     // model.h:32-40
    dvec3 Ks, Tr;
    double Ns = 1, Ni = 1;
    std::shared_ptr<Texture> Map_Kd;
    dvec3 radiance;
};

class CameraInfo {                                    // model.h:42-49
public:
    dvec3 eye, lookat, up;
    double fovy = 0;
    int height = 0, width = 0;
    std::map<std::string, dvec3> lightinfo;
};

class Model {                                         // model.h:51-67
public:
    std::vector<dvec3> vertex, normal;
    std::vector<dvec2> texture;
    std::vector<imat3x4> face;
    std::vector<Material> materials;
    CameraInfo camerainfo;
    explicit Model(const std::string& filename, bool reference_index_order = false);
    bool ok = false;                                  // the reference only prints to cerr on failure; this also records it
private:
    std::map<std::string, int> material_map;
    void load_material(const std::string& filename);
    void loadCameraFromXML(const std::string& filename);
};
