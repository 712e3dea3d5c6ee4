// Blender-ASCII scene reader: the reference's `Scene::parseSceneFile` grammar (Code/utilities/scene.cpp:124-546,
// SURVEY Appendix B) producing the POD RtSceneDesc of the C-ABI.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "host_config.h"
#include "host_image.h"

namespace rt {

// Owns every array an RtSceneDesc points at.
struct SceneFile {
    RtSceneDesc desc{};
    std::vector<RtLight> lights;
    std::vector<RtShape> shapes;
    std::vector<RtMaterial> materials;
    std::vector<RtTexture> textures;
    std::vector<std::shared_ptr<Image8>> texture_images;
    std::vector<std::string> texture_names;
    ImageF env;
    std::vector<std::string> warnings;
    void refresh_desc();  // re-point desc at the vectors after they changed
};

// Compose T*Rz*Ry*Rx*S and its inverse (scene.cpp:277-287) into shape.transform / shape.inv_transform.
void shape_set_trs(RtShape& shape, const double translation[3], const double rotation[3], const double scale[3]);

// Throws std::runtime_error on the conditions the reference throws on (unreadable file, vector with fewer than
// three numbers, no camera).  Missing textures / HDR files only add a warning, as in the reference.
std::unique_ptr<SceneFile> parse_scene_text(const std::string& text, const std::string& texture_root,
                                            const ConfigData& cfg, double shutter_time);
std::unique_ptr<SceneFile> load_scene_file(const std::string& path, const std::string& texture_root,
                                           const ConfigData& cfg, double shutter_time);

}  // namespace rt
