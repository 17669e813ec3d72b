// .crtscene (JSON) and .obj readers.  The .crtscene grammar is the one R/CRTSceneParser.cpp accepts
// (top-level keys settings / camera / objects / lights / materials / textures, parse order of :414-419);
// where the reference reads an absent key through MemberEnd() or leaves a field uninitialised
// (:87,:123-130,:196,:347-391 -- SURVEY.md section 5 "Parser hazards") this reader uses a default instead.
#include "mem_util.h"
#include "json_min.h"
#include "scene.h"

#include <cstdio>
#include <cstring>
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

// ---- binary scene cache (.crtbin) ------------------------------------------------------------------------------
namespace {
constexpr uint32_t kBinMagic = 0x42545243u; // "CRTB"
constexpr uint32_t kBinVersion = 1;

struct Writer {
    std::ofstream out;
    explicit Writer(const std::string& path) : out(path, std::ios::binary) { if (!out) throw std::runtime_error("cannot open '" + path + "' for writing"); }
    template <class T> void pod(const T& v) { out.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
    void bytes(const void* p, size_t n) { out.write(static_cast<const char*>(p), static_cast<std::streamsize>(n)); }
    void str(const std::string& s) { pod(static_cast<uint32_t>(s.size())); bytes(s.data(), s.size()); }
    void vec(const Vector& v) { bytes(v.data(), 12); }
};

struct Reader {
    const char* p;
    const char* end;
    template <class T> T pod()
    {
        if (static_cast<size_t>(end - p) < sizeof(T)) throw std::runtime_error("crtbin: truncated file");
        T v;
        crt::copyBytes(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    void bytes(void* dst, size_t n)
    {
        if (static_cast<size_t>(end - p) < n) throw std::runtime_error("crtbin: truncated file");
        crt::copyBytes(dst, p, n);
        p += n;
    }
    std::string str()
    {
        const uint32_t n = pod<uint32_t>();
        if (static_cast<size_t>(end - p) < n) throw std::runtime_error("crtbin: truncated file");
        std::string s(p, n);
        p += n;
        return s;
    }
    Vector vec() { float f[3]; bytes(f, 12); return Vector(f[0], f[1], f[2]); }
};
} // namespace

void SceneParser::saveBinary(const std::string& fileName, const Scene& scene)
{
    Writer w(fileName);
    w.pod(kBinMagic);
    w.pod(kBinVersion);
    w.vec(scene.settings.backgroundColor);
    w.pod(static_cast<int32_t>(scene.settings.imageWidth));
    w.pod(static_cast<int32_t>(scene.settings.imageHeight));
    w.vec(scene.camera.getPosition());
    w.bytes(scene.camera.getRotationMatrix().data(), 36);
    w.pod(static_cast<uint32_t>(scene.lights.size()));
    for (const Light& l : scene.lights) { w.vec(l.getPosition()); w.pod(l.getIntensity()); }
    w.pod(static_cast<uint32_t>(scene.materials.size()));
    for (const Material& m : scene.materials) {
        w.pod(static_cast<uint32_t>(m.getType()));
        w.vec(m.getAlbedo());
        w.pod(static_cast<uint32_t>(m.isSmoothShading() ? 1 : 0));
        w.pod(m.getIor());
        w.str(m.getTextureName());
    }
    w.pod(static_cast<uint32_t>(scene.textures.size()));
    for (const TextureDesc& t : scene.textures) {
        w.str(t.name); w.str(t.type); w.vec(t.colorA); w.vec(t.colorB); w.pod(t.scalar); w.str(t.filePath);
    }
    w.pod(static_cast<uint32_t>(scene.geometryObjects.size()));
    for (const Mesh& m : scene.geometryObjects) {
        w.pod(static_cast<int32_t>(m.getMaterialIndex()));
        w.pod(static_cast<uint64_t>(m.getVertices().size()));
        w.pod(static_cast<uint64_t>(m.getIndices().size()));
        w.pod(static_cast<uint64_t>(m.getVertexNormals().size()));
        w.pod(static_cast<uint64_t>(m.getUV().size()));
        w.bytes(m.getVertices().data(), 12 * m.getVertices().size());
        w.bytes(m.getIndices().data(), 4 * m.getIndices().size());
        w.bytes(m.getVertexNormals().data(), 12 * m.getVertexNormals().size());
        w.bytes(m.getUV().data(), 12 * m.getUV().size());
    }
    if (!w.out) throw std::runtime_error("write error on '" + fileName + "'");
}

void SceneParser::parseBinary(const std::string& bytes, Scene& scene)
{
    Reader r{ bytes.data(), bytes.data() + bytes.size() };
    if (r.pod<uint32_t>() != kBinMagic) throw std::runtime_error("crtbin: bad magic");
    if (r.pod<uint32_t>() != kBinVersion) throw std::runtime_error("crtbin: unsupported version");
    scene.settings.backgroundColor = r.vec();
    scene.settings.imageWidth = r.pod<int32_t>();
    scene.settings.imageHeight = r.pod<int32_t>();
    scene.camera.setPosition(r.vec());
    float m[9];
    r.bytes(m, 36);
    scene.camera.setRotationMatrix(Matrix(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]));
    for (uint32_t i = 0, n = r.pod<uint32_t>(); i < n; i++) {
        const Vector pos = r.vec();
        scene.lights.emplace_back(pos, r.pod<float>());
    }
    for (uint32_t i = 0, n = r.pod<uint32_t>(); i < n; i++) {
        Material mat;
        const uint32_t type = r.pod<uint32_t>();
        if (type > 4) throw std::runtime_error("crtbin: bad material type");
        mat.setType(static_cast<MaterialType>(type));
        mat.setAlbedo(r.vec());
        mat.setSmoothShading(r.pod<uint32_t>() != 0);
        mat.setIor(r.pod<float>());
        mat.setTextureName(r.str());
        scene.materials.push_back(mat);
    }
    for (uint32_t i = 0, n = r.pod<uint32_t>(); i < n; i++) {
        TextureDesc t;
        t.name = r.str(); t.type = r.str(); t.colorA = r.vec(); t.colorB = r.vec(); t.scalar = r.pod<float>(); t.filePath = r.str();
        scene.textures.push_back(t);
    }
    for (uint32_t i = 0, n = r.pod<uint32_t>(); i < n; i++) {
        Mesh& mesh = scene.addObject();
        mesh.setMaterialIndex(r.pod<int32_t>());
        const uint64_t nv = r.pod<uint64_t>(), ni = r.pod<uint64_t>(), nn = r.pod<uint64_t>(), nu = r.pod<uint64_t>();
        const uint64_t remaining = static_cast<uint64_t>(r.end - r.p);
        if (nv > remaining / 12 || ni > remaining / 4 || nn > remaining / 12 || nu > remaining / 12 || ni % 3 != 0 || (nn != 0 && nn != nv))
            throw std::runtime_error("crtbin: inconsistent mesh header");
        std::vector<Vector> v(nv), nrm(nn), uv(nu);
        std::vector<int> idx(ni);
        r.bytes(v.data(), 12 * nv);
        r.bytes(idx.data(), 4 * ni);
        r.bytes(nrm.data(), 12 * nn);
        r.bytes(uv.data(), 12 * nu);
        for (int k : idx)
            if (k < 0 || static_cast<uint64_t>(k) >= nv) throw std::runtime_error("crtbin: triangle index out of range");
        mesh.assign(std::move(v), std::move(idx), std::move(nrm), std::move(uv));
    }
    if (r.p != r.end) throw std::runtime_error("crtbin: trailing bytes");
}

void SceneParser::parseScene(const std::string& sceneFileName, Scene& scene)
{
    const std::string text = slurp(sceneFileName);
    if (endsWith(sceneFileName, ".obj")) parseObj(text, scene);
    else if (endsWith(sceneFileName, ".crtbin")) parseBinary(text, scene);
    else parseCrtscene(text, scene);
    // bitmap textures are decoded at load time, like CRTTextureBitmap's constructor does (R/CRTTextureBitmap.cpp:6-10)
    const size_t slash = sceneFileName.find_last_of("/\\");
    const std::string dir = slash == std::string::npos ? std::string() : sceneFileName.substr(0, slash);
    for (TextureDesc& t : scene.textures)
        if (t.typeCode() == 3u && t.pixels.empty()) t.loadBitmap(dir);
}

} // namespace crt
