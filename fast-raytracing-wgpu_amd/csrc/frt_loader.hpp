// frt_loader.hpp — model import for the scene builder: the reference's glTF path (src/scene/loader.rs:9-181 load_gltf, which
// rests on the `gltf 1.4.1` and `image 0.25.9` crates) restated over a small JSON reader, a GLB / buffer / accessor reader, a PNG
// decoder on zlib's inflate and image-0.25's Lanczos3 `resize_exact`; plus a Wavefront OBJ subset (an extension: the reference
// has no OBJ path) so that the usual Bunny / Sponza distributions load. Host only; nothing here touches the GPU.
#pragma once
#include "frt_scene.hpp"
#include <string>
#include <vector>

namespace frt {

struct LoadedModel {                              // the 4-tuple load_gltf returns, loader.rs:12
    std::vector<Geometry> geometries;             // one per mesh primitive, in document order
    std::vector<frt_material> materials;          // texture slots hold IMAGE indices until add_gltf_materials remaps them
    std::vector<std::vector<uint8_t>> images;     // RGBA8, TEXTURE_WIDTH x TEXTURE_HEIGHT (1024 x 1024) each
    std::vector<uint32_t> material_indices;       // per geometry: index into `materials` (0 when the primitive names none)
    std::vector<std::string> warnings;            // what the reference prints (unsupported image format, …)
};

bool load_gltf(const std::string& path, LoadedModel& out, std::string& err);   // .gltf (+ .bin / data: URIs) and .glb
bool load_obj(const std::string& path, LoadedModel& out, std::string& err);    // v / vt / vn / f, fan-triangulated
bool load_model(const std::string& path, LoadedModel& out, std::string& err);  // by extension

// image 0.25.9 imageops::resize(.., FilterType::Lanczos3) for RGBA8 (src W x H -> dst 1024 x 1024 when used by the loader)
void resize_lanczos3_rgba8(const uint8_t* src, uint32_t sw, uint32_t sh, uint8_t* dst, uint32_t dw, uint32_t dh);
// PNG (8-bit RGB / RGBA / palette, non-interlaced; what gltf's importer hands the reference as R8G8B8 / R8G8B8A8). channels: 3 or 4.
bool decode_png(const uint8_t* data, size_t size, std::vector<uint8_t>& pixels, uint32_t& w, uint32_t& h, uint32_t& channels, std::string& why);

// Huffman JPEG (baseline and progressive, 8-bit, 3-component) -> RGB8
bool decode_jpeg(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h, std::string& why);

// builder.rs:191-314
std::vector<uint32_t> add_gltf_materials(SceneBuilder& b, const LoadedModel& m);
std::vector<uint32_t> add_gltf_meshes(SceneBuilder& b, const LoadedModel& m);
void add_gltf_instances(SceneBuilder& b, const std::vector<uint32_t>& mesh_ids, const std::vector<uint32_t>& mat_ids,
                        const std::vector<uint32_t>& material_indices, const Mat4& transform);

namespace scenes {
// scenes.rs:246-322. Returns false (and the loader's message) when the file cannot be loaded; the builder is then left empty
// (the reference logs the error and builds an empty scene; this builder has no empty scenes, so callers report the failure).
bool create_gltf_scene(SceneBuilder& b, const std::string& path, const Mat4& model_transform, const Mat4& light_transform, std::string& err);
}

} // namespace frt

struct frt_model { frt::LoadedModel m; };
