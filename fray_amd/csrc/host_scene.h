// Host-side scene layer: .fray parser, OBJ/BMP/EXR loaders, KD-tree builder, flattening into
// frayhip_scene_desc.  Stands behind Scene::parseScene + Scene::beginRender of the reference
// (src/scene.cpp:751-767).  Pure C++, no GPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "frayhip.h"

namespace frayhost {

struct MeshData {
    std::vector<double> vertices, normals, uvs;       // xyz triples
    std::vector<frayhip_triangle> triangles;
    std::vector<frayhip_kdnode> kdnodes;
    std::vector<int32_t> trirefs;
    double bbox_min[3], bbox_max[3];
    bool faceted = false, backfaceCulling = true, useKD = true;
    int maxDepth = 0, depthSum = 0;
};

struct Image {                 // Bitmap, src/bitmap.h:31-35
    int w = -1, h = -1;
    std::vector<float> rgb;    // w*h*3
    bool ok() const { return !rgb.empty(); }
};

bool load_bmp(const char* path, Image& img, std::string& err);   // bitmap.cpp:117-195
bool load_exr(const char* path, Image& img, std::string& err);   // bitmap.cpp:238-264 (own PIZ decoder)
bool load_obj(const char* path, MeshData& mesh);                 // mesh.cpp:203-258
void build_kd(MeshData& mesh);                                   // mesh.cpp:67-94,320-355

// Owns every array the flattened description points into.
struct HostScene {
    frayhip_scene_desc desc{};
    std::vector<frayhip_node> nodes;
    std::vector<frayhip_geom_ref> geoms;
    std::vector<frayhip_plane> planes;
    std::vector<frayhip_sphere> spheres;
    std::vector<frayhip_cube> cubes;
    std::vector<frayhip_csg> csgs;
    std::vector<frayhip_mesh> meshes;
    std::vector<MeshData> meshData;
    std::vector<frayhip_shader> shaders;
    std::vector<frayhip_layer> layers;
    std::vector<frayhip_texture> textures;
    std::vector<frayhip_light> lights;
    std::vector<float> texels;
    std::vector<std::string> warnings;
    void finalize();           // fills desc pointers/counts from the vectors
};

// Returns nullptr and sets err on failure.
HostScene* parse_scene_file(const char* path, std::string& err);

}  // namespace frayhost
