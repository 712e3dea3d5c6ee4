#include "host_scene.h"

#include <array>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "host_math.h"

namespace rt {

void SceneFile::refresh_desc() {
    textures.resize(texture_images.size());
    for (size_t i = 0; i < texture_images.size(); ++i) {
        textures[i].width = texture_images[i]->width;
        textures[i].height = texture_images[i]->height;
        textures[i].rgb = texture_images[i]->rgb.data();
    }
    desc.n_lights = static_cast<int32_t>(lights.size());
    desc.n_shapes = static_cast<int32_t>(shapes.size());
    desc.n_materials = static_cast<int32_t>(materials.size());
    desc.n_textures = static_cast<int32_t>(textures.size());
    desc.lights = lights.data();
    desc.shapes = shapes.data();
    desc.materials = materials.data();
    desc.textures = textures.data();
    if (desc.env.present) {
        desc.env.width = env.width;
        desc.env.height = env.height;
        desc.env.rgb = env.rgb.empty() ? nullptr : env.rgb.data();
    }
}

void shape_set_trs(RtShape& shape, const double t[3], const double r[3], const double s[3]) {
    M4 m = mul(mul(mul(mul(translation(V3(t)), rotation_z(r[2])), rotation_y(r[1])), rotation_x(r[0])), scaling(V3(s)));
    M4 inv = inverse(m);
    to_row_major(m, shape.transform);
    to_row_major(inv, shape.inv_transform);
}

namespace {

// material.h:51 defaults
RtMaterial default_material() {
    RtMaterial m{};
    for (int i = 0; i < 3; ++i) {
        m.ambient[i] = 0.1;
        m.diffuse[i] = 0.7;
        m.specular[i] = 1.0;
    }
    m.shininess = 32.0;
    m.reflectivity = 0.0;
    m.transparency = 0.0;
    m.refractive_index = 1.0;
    m.texture = -1;
    m.bump_map = -1;
    m.is_metal = 0;  // type defaults to "glass" (material.h:47)
    return m;
}

// scene.cpp:24-28
void read_vec(std::istringstream& ss, double v[3]) {
    if (!(ss >> v[0] >> v[1] >> v[2])) throw std::runtime_error("Error reading vector components.");
}

struct PendingShape {
    double translation[3] = {0, 0, 0};
    double rotation[3] = {0, 0, 0};
    double scale[3] = {1, 1, 1};
    std::vector<std::array<double, 3>> corners;
    RtMaterial mat = default_material();
    std::string texture_file, bump_file, type = "glass";
    double velocity[3] = {0, 0, 0};
    void reset_transform() {
        for (int i = 0; i < 3; ++i) {
            translation[i] = 0;
            rotation[i] = 0;
            scale[i] = 1;
        }
    }
    void reset_material() {
        mat = default_material();
        texture_file.clear();
        bump_file.clear();
        type = "glass";
        for (double& v : velocity) v = 0;
    }
};

enum class Block { None, Camera, Light, Sphere, ComplexSphere, Cube, ComplexCube, Plane, ComplexPlane };

struct Parser {
    SceneFile& out;
    const std::string& texture_root;
    const ConfigData& cfg;
    std::map<std::string, int> texture_cache;

    std::string resolve(const std::string& name) const {
        if (texture_root.empty()) return name;
        return texture_root + "/" + name;
    }

    // scene.cpp:269-275: textures are (re)loaded at END_<SHAPE>; a failed load leaves the slot empty.
    int texture_index(const std::string& name) {
        if (name.empty()) return -1;
        auto it = texture_cache.find(name);
        if (it != texture_cache.end()) return it->second;
        auto img = std::make_shared<Image8>();
        std::string why;
        int idx = -1;
        if (load_texture(resolve(name), *img, &why)) {
            idx = static_cast<int>(out.texture_images.size());
            out.texture_images.push_back(img);
            out.texture_names.push_back(name);
        } else {
            out.warnings.push_back("Error loading texture: " + name + ": " + why);
        }
        texture_cache[name] = idx;
        return idx;
    }

    // Material keys shared by every shape block (scene.cpp:438-450 and its five copies).
    static bool material_key(const std::string& key, std::istringstream& ss, PendingShape& p) {
        if (key == "ambient") read_vec(ss, p.mat.ambient);
        else if (key == "diffuse") read_vec(ss, p.mat.diffuse);
        else if (key == "specular") read_vec(ss, p.mat.specular);
        else if (key == "shininess") ss >> p.mat.shininess;
        else if (key == "reflectivity") ss >> p.mat.reflectivity;
        else if (key == "transparency") ss >> p.mat.transparency;
        else if (key == "refractive_index") ss >> p.mat.refractive_index;
        else if (key == "texture_file") ss >> p.texture_file;
        else if (key == "bump_map_file") ss >> p.bump_file;
        else if (key == "velocity") read_vec(ss, p.velocity);
        else if (key == "material") ss >> p.type;
        else return false;
        return true;
    }

    int finish_material(PendingShape& p) {
        p.mat.texture = texture_index(p.texture_file);
        p.mat.bump_map = texture_index(p.bump_file);
        p.mat.is_metal = (p.type == "metal") ? 1 : 0;
        out.materials.push_back(p.mat);
        return static_cast<int>(out.materials.size()) - 1;
    }

    void emit_transformed(PendingShape& p, int type) {
        RtShape s{};
        s.type = type;
        s.material = finish_material(p);
        shape_set_trs(s, p.translation, p.rotation, p.scale);
        std::memcpy(s.velocity, p.velocity, sizeof(s.velocity));
        // complex_sphere.cpp:17 / complex_cube.cpp:21 / complex_plane.cpp:20: per-class default when the key is absent
        double def = (type == RT_SHAPE_COMPLEX_SPHERE) ? 0.15 : 0.2;
        s.max_displacement = (type >= RT_SHAPE_COMPLEX_SPHERE) ? cfg.get_double("advanced.displacement_strength", def) : 0.0;
        out.shapes.push_back(s);
    }

    void emit_plane(PendingShape& p) {
        int mat = finish_material(p);  // textures load before the corner-count check (scene.cpp:373-384)
        if (p.corners.size() != 4) {
            out.materials.pop_back();
            out.warnings.push_back("Warning: Plane block ended with " + std::to_string(p.corners.size()) + " corners, expected 4.");
            return;
        }
        RtShape s{};
        s.type = RT_SHAPE_PLANE;
        s.material = mat;
        M4 id;
        to_row_major(id, s.transform);
        to_row_major(id, s.inv_transform);
        for (int c = 0; c < 4; ++c)
            for (int k = 0; k < 3; ++k) s.corners[c * 3 + k] = p.corners[c][k];
        std::memcpy(s.velocity, p.velocity, sizeof(s.velocity));
        out.shapes.push_back(s);
    }
};

}  // namespace

std::unique_ptr<SceneFile> parse_scene_text(const std::string& text, const std::string& texture_root,
                                            const ConfigData& cfg, double shutter_time) {
    auto scene = std::make_unique<SceneFile>();
    Parser P{*scene, texture_root, cfg, {}};
    scene->desc.shutter_time = shutter_time;

    // scene.cpp:134-138 camera staging values (persist across CAMERA blocks; the last END_CAMERA wins)
    RtCamera cam{};
    cam.f_stop = 99999.0;
    cam.focal_distance = 10.0;
    bool have_camera = false;

    RtLight light{};
    PendingShape shp;
    Block block = Block::None;

    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok.empty() || tok[0] == '#') continue;

        if (tok == "HDR_BACKGROUND") {  // scene.cpp:166-176, recognised anywhere in the file
            std::string name;
            ss >> name;
            if (!name.empty()) {
                scene->desc.env.present = 1;
                std::string why;
                if (!read_pfm(P.resolve(name), scene->env, &why)) scene->warnings.push_back("Failed to load HDR image: " + name + ": " + why);
            }
            continue;
        }

        // ---- block openers (scene.cpp:178-246) ----
        if (tok == "CAMERA") { block = Block::Camera; continue; }
        if (tok == "POINT_LIGHT") {
            block = Block::Light;
            light = RtLight{};
            for (double& v : light.intensity) v = 1.0;
            continue;
        }
        if (tok == "SPHERE" || tok == "COMPLEX_SPHERE" || tok == "CUBE" || tok == "COMPLEX_CUBE" || tok == "COMPLEX_PLANE") {
            block = tok == "SPHERE" ? Block::Sphere : tok == "COMPLEX_SPHERE" ? Block::ComplexSphere : tok == "CUBE" ? Block::Cube
                  : tok == "COMPLEX_CUBE" ? Block::ComplexCube : Block::ComplexPlane;
            shp.reset_transform();
            shp.reset_material();
            continue;
        }
        if (tok == "PLANE") {
            block = Block::Plane;
            shp.corners.clear();
            shp.reset_material();
            continue;
        }

        // ---- block closers (scene.cpp:250-410): act on the staged values whatever block is open ----
        if (tok == "END_CAMERA") { have_camera = true; scene->desc.camera = cam; block = Block::None; continue; }
        if (tok == "END_POINT_LIGHT") { scene->lights.push_back(light); block = Block::None; continue; }
        if (tok == "END_SPHERE") { P.emit_transformed(shp, RT_SHAPE_SPHERE); block = Block::None; continue; }
        if (tok == "END_COMPLEX_SPHERE") { P.emit_transformed(shp, RT_SHAPE_COMPLEX_SPHERE); block = Block::None; continue; }
        if (tok == "END_CUBE") { P.emit_transformed(shp, RT_SHAPE_CUBE); block = Block::None; continue; }
        if (tok == "END_COMPLEX_CUBE") { P.emit_transformed(shp, RT_SHAPE_COMPLEX_CUBE); block = Block::None; continue; }
        if (tok == "END_PLANE") { P.emit_plane(shp); block = Block::None; continue; }
        if (tok == "END_COMPLEX_PLANE") { P.emit_transformed(shp, RT_SHAPE_COMPLEX_PLANE); block = Block::None; continue; }

        // ---- block bodies (scene.cpp:415-545); unknown keys are ignored ----
        switch (block) {
            case Block::Camera:
                if (tok == "location") read_vec(ss, cam.location);
                else if (tok == "gaze_direction") read_vec(ss, cam.gaze_direction);
                else if (tok == "up_vector") read_vec(ss, cam.up_vector);
                else if (tok == "focal_length") ss >> cam.focal_length;
                else if (tok == "sensor_size") ss >> cam.sensor_width >> cam.sensor_height;
                else if (tok == "resolution") ss >> cam.resolution_x >> cam.resolution_y;
                else if (tok == "f_stop") ss >> cam.f_stop;
                else if (tok == "focal_distance") ss >> cam.focal_distance;
                break;
            case Block::Light:
                if (tok == "location") read_vec(ss, light.position);
                else if (tok == "intensity") read_vec(ss, light.intensity);
                else if (tok == "radius") ss >> light.radius;
                break;
            case Block::Plane:
                if (tok == "corner") {
                    std::array<double, 3> c{};
                    read_vec(ss, c.data());
                    shp.corners.push_back(c);
                } else {
                    Parser::material_key(tok, ss, shp);
                }
                break;
            case Block::Sphere:
            case Block::ComplexSphere:
            case Block::Cube:
            case Block::ComplexCube:
            case Block::ComplexPlane:
                if (tok == "translation") read_vec(ss, shp.translation);
                else if (tok == "rotation_euler_radians") read_vec(ss, shp.rotation);
                else if (tok == "scale") read_vec(ss, shp.scale);
                else Parser::material_key(tok, ss, shp);
                break;
            case Block::None:
                break;
        }
    }
    if (!have_camera) throw std::runtime_error("Scene file error: No camera data found.");  // scene.cpp:98-100
    scene->refresh_desc();
    return scene;
}

std::unique_ptr<SceneFile> load_scene_file(const std::string& path, const std::string& texture_root,
                                           const ConfigData& cfg, double shutter_time) {
    std::ifstream f(path);
    if (!f.is_open()) throw std::runtime_error("Failed to open scene file: " + path);  // scene.cpp:126-128
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_scene_text(ss.str(), texture_root, cfg, shutter_time);
}

}  // namespace rt
