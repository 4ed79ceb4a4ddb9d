// Scene description files: the "self-defined material description file" the reference's roadmap
// lists as future work (README.md:62), extended to the whole scene so that nothing but the
// built-in Cornell box needs a recompile (the reference: "Change the definition here to change
// resolution", gpu_path_tracer/main.cpp:30; scene hard-coded in common/Scene.cpp:25-73).
//
// Plain text, one statement per line, '#' starts a comment, paths relative to the file:
//
//   name       <words...>                      scene name (the output file is out/gpu/<name>.ppm)
//   resolution <width> <height>
//   camera     <x> <y> <z> [fov <degrees>]
//   material   <id> lambert [albedo r g b] [emission r g b]
//   material   <id> metal   [albedo r g b] [metallic m] [roughness r] [emission r g b]
//   material   <id> glass   [roughness r] [ior n] [albedo r g b]
//   mesh       <file.obj> <material id> [translate x y z] [rotate x y z] [scale x y z | scale s]
//   heightfield <n> <material id>              the built-in n x n displaced height field
//   sphere     <x> <y> <z> <radius> <material id>
//
// Unset material parameters keep the defaults of the reference's Material (common/Material.hpp:137-142).
// Meshes are added in file order (= Scene::Add order = primitive id order).
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "Scene.hpp"

namespace {

[[noreturn]] void bad(const std::string& path, int line, const std::string& what) {
    throw std::runtime_error("scene file '" + path + "' line " + std::to_string(line) + ": " + what);
}

bool read_floats(std::istringstream& in, float* out, int n) {
    for (int i = 0; i < n; ++i)
        if (!(in >> out[i])) return false;
    return true;
}

}  // namespace

Scene Scene::LoadSceneFile(const std::string& path, int maxDepth) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("scene file '" + path + "' cannot be opened");
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    Scene scene(512, 512);
    scene.name = "custom";
    scene.maxDepth = maxDepth;
    std::map<std::string, Material> materials;
    std::string text;
    int lineno = 0, meshes = 0;
    while (std::getline(file, text)) {
        ++lineno;
        const size_t hash = text.find('#');
        if (hash != std::string::npos) text.erase(hash);
        std::istringstream in(text);
        std::string key;
        if (!(in >> key)) continue;
        if (key == "name") {
            std::string rest;
            std::getline(in, rest);
            const size_t b = rest.find_first_not_of(" \t"), e = rest.find_last_not_of(" \t\r");
            if (b == std::string::npos) bad(path, lineno, "name needs a value");
            scene.name = rest.substr(b, e - b + 1);
        } else if (key == "resolution") {
            if (!(in >> scene.width >> scene.height) || scene.width <= 0 || scene.height <= 0) bad(path, lineno, "resolution <width> <height>");
        } else if (key == "camera") {
            float p[3];
            if (!read_floats(in, p, 3)) bad(path, lineno, "camera <x> <y> <z> [fov <degrees>]");
            scene.camPos = Vec3(p[0], p[1], p[2]);
            std::string opt;
            while (in >> opt) {
                if (opt != "fov" || !(in >> scene.fov)) bad(path, lineno, "camera <x> <y> <z> [fov <degrees>]");
            }
        } else if (key == "material") {
            std::string id, kind;
            if (!(in >> id >> kind)) bad(path, lineno, "material <id> lambert|metal|glass ...");
            Material m;
            if (kind == "lambert") m.type = Material::Type::Lambertian;
            else if (kind == "metal") m.type = Material::Type::MetallicWorkflow;
            else if (kind == "glass") m.type = Material::Type::Glass;
            else bad(path, lineno, "material kind '" + kind + "' is none of lambert, metal, glass");
            std::string opt;
            while (in >> opt) {
                float v[3];
                if (opt == "albedo" && read_floats(in, v, 3)) m._albedo = Vec3(v[0], v[1], v[2]);
                else if (opt == "emission" && read_floats(in, v, 3)) m._emission = Vec3(v[0], v[1], v[2]);
                else if (opt == "metallic" && read_floats(in, v, 1)) m._metallic = v[0];
                else if (opt == "roughness" && read_floats(in, v, 1)) m._roughness = v[0];
                else if (opt == "ior" && read_floats(in, v, 1)) m._ior = v[0];
                else bad(path, lineno, "bad material parameter '" + opt + "'");
            }
            if (m.type == Material::Type::Glass && !(m._ior > 0.f)) bad(path, lineno, "a glass material needs ior > 0");
            materials[id] = m;
        } else if (key == "mesh") {
            std::string obj, id;
            if (!(in >> obj >> id)) bad(path, lineno, "mesh <file.obj> <material id> [translate|rotate|scale ...]");
            if (!materials.count(id)) bad(path, lineno, "material '" + id + "' has not been defined");
            Transform t;
            std::string opt;
            while (in >> opt) {
                float v[3];
                if (opt == "translate" && read_floats(in, v, 3)) t.translation = Vec3(v[0], v[1], v[2]);
                else if (opt == "rotate" && read_floats(in, v, 3)) t.rotation = Vec3(v[0], v[1], v[2]);
                else if (opt == "scale" && read_floats(in, v, 1)) {
                    float w[2];
                    std::istringstream::pos_type at = in.tellg();
                    if (read_floats(in, w, 2)) t.scale = Vec3(v[0], w[0], w[1]);
                    else {
                        in.clear();
                        in.seekg(at);
                        t.scale = Vec3(v[0]);
                    }
                } else bad(path, lineno, "bad mesh parameter '" + opt + "'");
            }
            const std::string full = !obj.empty() && obj[0] == '/' ? obj : dir + obj;
            scene.Add(new MeshTriangle(full, materials[id], t));
            ++meshes;
        } else if (key == "sphere") {
            float v[4];
            std::string id;
            if (!read_floats(in, v, 4) || !(in >> id)) bad(path, lineno, "sphere <x> <y> <z> <radius> <material id>");
            if (!materials.count(id)) bad(path, lineno, "material '" + id + "' has not been defined");
            scene.Add(new Sphere(Vec3(v[0], v[1], v[2]), v[3], materials[id]));
            ++meshes;
        } else if (key == "heightfield") {
            int n = 0;
            std::string id;
            if (!(in >> n >> id)) bad(path, lineno, "heightfield <n> <material id>");
            if (!materials.count(id)) bad(path, lineno, "material '" + id + "' has not been defined");
            scene.Add(new MeshTriangle(HeightfieldVertices(n), materials[id]));
            ++meshes;
        } else {
            bad(path, lineno, "unknown statement '" + key + "'");
        }
    }
    if (meshes == 0) throw std::runtime_error("scene file '" + path + "' defines no mesh");
    scene.buildBVH();
    return scene;
}
