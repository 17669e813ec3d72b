// .crtscene (JSON) and .obj readers.  The .crtscene grammar is the one R/CRTSceneParser.cpp accepts
// (top-level keys settings / camera / objects / lights / materials / textures, parse order of :414-419);
// where the reference reads an absent key through MemberEnd() or leaves a field uninitialised
// (:87,:123-130,:196,:347-391 -- SURVEY.md section 5 "Parser hazards") this reader uses a default instead.
#include "json_min.h"
#include "scene.h"

#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace crt {
namespace {

using json::Value;

Vector vectorAt(const Value& arr, size_t start)
{
    if (!arr.isArray() || start + 3 > arr.size())
        throw std::runtime_error("crtscene: expected an array with 3 numbers at offset " + std::to_string(start));
    return Vector(static_cast<float>(arr.numberAt(start)), static_cast<float>(arr.numberAt(start + 1)),
                  static_cast<float>(arr.numberAt(start + 2)));
}

Matrix matrixFrom(const Value& arr)
{
    if (!arr.isArray() || arr.size() != 9) throw std::runtime_error("crtscene: camera matrix needs 9 numbers");
    float m[9];
    for (int i = 0; i < 9; i++) m[i] = static_cast<float>(arr.numberAt(i));
    return Matrix(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]);
}

float numberOr(const Value* v, float fallback) { return (v && v->isNumber()) ? static_cast<float>(v->num) : fallback; }

void readSettings(const Value& doc, Settings& st)
{
    const Value* s = doc.find("settings");
    if (!s || !s->isObject()) return;
    if (const Value* bg = s->find("background_color")) st.backgroundColor = vectorAt(*bg, 0);
    if (const Value* img = s->find("image_settings")) {
        if (const Value* w = img->find("width")) st.imageWidth = static_cast<int>(numberOr(w, 0.f));
        if (const Value* h = img->find("height")) st.imageHeight = static_cast<int>(numberOr(h, 0.f));
    }
}

void readCamera(const Value& doc, Camera& cam)
{
    const Value* c = doc.find("camera");
    if (!c || !c->isObject()) return;
    if (const Value* m = c->find("matrix")) cam.setRotationMatrix(matrixFrom(*m));
    if (const Value* p = c->find("position")) cam.setPosition(vectorAt(*p, 0));
}

void readMesh(const Value& o, Mesh& mesh)
{
    const Value* uvs = o.find("uvs"); // optional (absent in the shipped scene)
    if (uvs && uvs->isArray())
        for (size_t i = 0; i + 2 < uvs->size(); i += 3) mesh.addUV(vectorAt(*uvs, i));

    const Value* verts = o.find("vertices");
    const Value* tris = o.find("triangles");
    const size_t nv = (verts && verts->isArray()) ? verts->size() / 3 : 0;
    const size_t ni = (tris && tris->isArray()) ? tris->size() : 0;
    mesh.reserve(nv, ni);
    for (size_t i = 0; i < nv; i++) mesh.addVertex(vectorAt(*verts, 3 * i));
    for (size_t i = 0; i < ni; i++) {
        const double d = tris->numberAt(i);
        if (d < 0 || d >= static_cast<double>(nv)) throw std::runtime_error("crtscene: triangle index out of range");
        mesh.addIndex(static_cast<int>(d));
    }
    if (ni % 3 != 0) throw std::runtime_error("crtscene: 'triangles' length is not a multiple of 3");
    mesh.setMaterialIndex(static_cast<int>(numberOr(o.find("material_index"), 0.f)));
    mesh.calculateVertexNormals(); // R/CRTSceneParser.cpp:131
}

MaterialType materialTypeFrom(const std::string& s) // R/CRTSceneParser.cpp:323-343: anything else is refractive
{
    if (s == "diffuse") return MaterialType::DIFFUSE;
    if (s == "reflective") return MaterialType::REFLECTIVE;
    if (s == "constant") return MaterialType::CONSTANT;
    return MaterialType::REFRACTIVE;
}

void readMaterial(const Value& m, Material& mat)
{
    const Value* type = m.find("type");
    if (type && type->isString()) mat.setType(materialTypeFrom(type->str));
    if (mat.getType() == MaterialType::REFRACTIVE) {
        mat.setIor(numberOr(m.find("ior"), 1.f));
        mat.setAlbedo(Vector(1.f, 1.f, 1.f));
    } else if (const Value* alb = m.find("albedo")) {
        if (alb->isArray()) mat.setAlbedo(vectorAt(*alb, 0));
        else if (alb->isString()) mat.setTextureName(alb->str);
    }
    const Value* smooth = m.find("smooth_shading");
    mat.setSmoothShading(smooth && smooth->isBool() && smooth->b);
}

void readTexture(const Value& t, TextureDesc& d) // R/CRTSceneParser.cpp:210-306, kept as data
{
    if (const Value* n = t.find("name"); n && n->isString()) d.name = n->str;
    std::string type;
    if (const Value* ty = t.find("type"); ty && ty->isString()) type = ty->str;
    if (type == "albedo") {
        d.type = type;
        if (const Value* a = t.find("albedo")) d.colorA = vectorAt(*a, 0);
    } else if (type == "edges") {
        d.type = type;
        if (const Value* a = t.find("edge_color")) d.colorA = vectorAt(*a, 0);
        if (const Value* b = t.find("inner_color")) d.colorB = vectorAt(*b, 0);
        d.scalar = numberOr(t.find("edge_width"), 0.f);
    } else if (type == "checker") {
        d.type = type;
        if (const Value* a = t.find("color_A")) d.colorA = vectorAt(*a, 0);
        if (const Value* b = t.find("color_B")) d.colorB = vectorAt(*b, 0);
        d.scalar = numberOr(t.find("square_size"), 0.f);
    } else {
        d.type = "bitmap";
        if (const Value* f = t.find("file_path"); f && f->isString()) d.filePath = f->str;
    }
}

std::string slurp(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open scene file '" + path + "'");
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

bool endsWith(const std::string& s, const char* suffix)
{
    const size_t n = std::char_traits<char>::length(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) {
        char a = s[s.size() - n + i], b = suffix[i];
        if (a >= 'A' && a <= 'Z') a = char(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

} // namespace

void SceneParser::parseCrtscene(const std::string& text, Scene& scene)
{
    const Value doc = json::parse(text);
    if (!doc.isObject()) throw std::runtime_error("crtscene: top level must be an object");
    readSettings(doc, scene.settings);
    readCamera(doc, scene.camera);
    if (const Value* objs = doc.find("objects"); objs && objs->kind == Value::Array)
        for (const Value& o : objs->arr) {
            scene.geometryObjects.emplace_back();
            readMesh(o, scene.geometryObjects.back());
        }
    if (const Value* ls = doc.find("lights"); ls && ls->kind == Value::Array)
        for (const Value& l : ls->arr) {
            Vector pos;
            if (const Value* p = l.find("position")) pos = vectorAt(*p, 0);
            scene.lights.emplace_back(pos, numberOr(l.find("intensity"), 0.f));
        }
    if (const Value* ms = doc.find("materials"); ms && ms->kind == Value::Array)
        for (const Value& m : ms->arr) {
            Material mat;
            readMaterial(m, mat);
            scene.materials.push_back(mat);
        }
    if (const Value* ts = doc.find("textures"); ts && ts->kind == Value::Array)
        for (const Value& t : ts->arr) {
            TextureDesc d;
            readTexture(t, d);
            scene.textures.push_back(d);
        }
}

// Wavefront .obj (extension; BASELINE.json's north_star speaks of ".obj scenes", the reference reads none):
// v / f records, faces fan-triangulated, negative (relative) indices, v/vt/vn forms; 'o' and 'g' start a new
// mesh that shares nothing with the previous one (vertices are re-indexed per mesh). One default diffuse
// material, identity camera at the origin.
void SceneParser::parseObj(const std::string& text, Scene& scene)
{
    std::vector<Vector> positions;
    struct Group { std::vector<int> idx; };
    std::vector<Group> groups(1);
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            float x = 0, y = 0, z = 0;
            ls >> x >> y >> z;
            positions.emplace_back(x, y, z);
        } else if (tag == "o" || tag == "g") {
            if (!groups.back().idx.empty()) groups.emplace_back();
        } else if (tag == "f") {
            std::vector<int> face;
            std::string tok;
            while (ls >> tok) {
                const int raw = std::atoi(tok.c_str()); // leading integer of v, v/vt, v//vn, v/vt/vn
                if (raw == 0) throw std::runtime_error("obj: bad face index '" + tok + "'");
                const int idx = raw > 0 ? raw - 1 : static_cast<int>(positions.size()) + raw;
                if (idx < 0 || idx >= static_cast<int>(positions.size())) throw std::runtime_error("obj: face index out of range");
                face.push_back(idx);
            }
            for (size_t k = 1; k + 1 < face.size(); k++) {
                groups.back().idx.push_back(face[0]);
                groups.back().idx.push_back(face[k]);
                groups.back().idx.push_back(face[k + 1]);
            }
        }
    }
    for (const Group& g : groups) {
        if (g.idx.empty()) continue;
        Mesh& mesh = scene.addObject();
        std::vector<int> remap(positions.size(), -1);
        int next = 0;
        for (int gi : g.idx) {
            if (remap[gi] < 0) {
                remap[gi] = next++;
                mesh.addVertex(positions[gi]);
            }
            mesh.addIndex(remap[gi]);
        }
        mesh.setMaterialIndex(0);
        mesh.calculateVertexNormals();
    }
    Material m;
    m.setType(MaterialType::DIFFUSE);
    m.setAlbedo(Vector(0.8f, 0.8f, 0.8f));
    scene.addMaterial(m);
}

void SceneParser::parseScene(const std::string& sceneFileName, Scene& scene)
{
    const std::string text = slurp(sceneFileName);
    if (endsWith(sceneFileName, ".obj")) parseObj(text, scene);
    else parseCrtscene(text, scene);
}

} // namespace crt
