// Host scene layer of the MI355X renderer: the API surface the reference's renderer and app are written
// against (R/ = /root/reference/DirectX-RayTracer/DirectX-RayTracer/):
//   crt::Vector   <-> CRTVector   R/CRTVector.h:4-33      crt::Matrix   <-> CRTMatrix  R/CRTMatrix.h:4-24
//   crt::Triangle <-> CRTTriangle R/CRTTriangle.h:4-27    crt::Mesh     <-> CRTMesh    R/CRTMesh.h:6-31
//   crt::Camera   <-> CRTCamera   R/CRTCamera.h:5-32      crt::Light    <-> CRTLight   R/CRTLight.h:4-16
//   crt::Material <-> CRTMaterial R/CRTMaterial.h:4-36    crt::Scene    <-> CRTScene   R/CRTScene.h:9-42
// Same member names and observable behaviour (pinned by tests/golden/dragon_scene_layer.json), written from
// scratch; crt_compat.h maps the CRT* names onto these classes for code that wants the old spelling.
#pragma once

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace crt {

class Vector {
public:
    Vector() : x(0.f), y(0.f), z(0.f) {}
    Vector(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}

    float length() const;
    void normalise(); // no zero guard, like the reference (R/CRTVector.cpp:18-25)

    float getX() const { return x; }
    float getY() const { return y; }
    float getZ() const { return z; }
    float getByIndex(int index) const { return index == 0 ? x : (index == 1 ? y : z); }
    const float* data() const { return &x; }

    friend Vector operator+(const Vector& a, const Vector& b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z); }
    friend Vector operator-(const Vector& a, const Vector& b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z); }
    friend Vector operator*(const Vector& v, float s) { return Vector(v.x * s, v.y * s, v.z * s); }
    friend Vector operator*(float s, const Vector& v) { return Vector(v.x * s, v.y * s, v.z * s); }
    friend Vector cross(const Vector& a, const Vector& b);
    friend float dot(const Vector& a, const Vector& b);
    friend bool operator==(const Vector& a, const Vector& b); // |delta| < 1e-6 per component

    void print(std::ostream& os) const;

private:
    float x, y, z;
};
static_assert(sizeof(Vector) == 12, "vertex buffers are uploaded as raw Vector arrays (R/DXRTRenderer.cpp:391-392)");

class Matrix {
public:
    Matrix(); // identity
    Matrix(float c00, float c01, float c02, float c10, float c11, float c12, float c20, float c21, float c22);

    friend Matrix operator*(const Matrix& a, const Matrix& b);
    friend Vector operator*(const Vector& v, const Matrix& m); // ROW vector times matrix (R/CRTMatrix.cpp:26-38)

    float get(int row, int col) const { return m[row][col]; }
    const float* data() const { return &m[0][0]; }
    void print() const;

private:
    float m[3][3];
};
static_assert(sizeof(Matrix) == 36, "3x3 row-major");

class Triangle {
public:
    static constexpr int vertsInTriangle = 3;
    Triangle() = default;
    Triangle(const Vector& v0, const Vector& v1, const Vector& v2);
    const Vector& getNormal() const { return normal; }
    const Vector& getVertex(int index) const { return verts[index]; }
    friend bool operator==(const Triangle& a, const Triangle& b);

private:
    Vector verts[vertsInTriangle];
    Vector normal; // normalize(cross(v1 - v0, v2 - v0))
};
static_assert(sizeof(Triangle) == 48, "3 vertices + normal");

class Mesh {
public:
    void addVertex(const Vector& v) { vertices.push_back(v); }
    void addIndex(int index) { indices.push_back(index); }
    void setMaterialIndex(int index) { materialIndex = index; }
    void addUV(const Vector& uv) { uvData.push_back(uv); }
    void setUVs(std::vector<Vector>&& uvs) { uvData = std::move(uvs); }
    void reserve(size_t n_vertices, size_t n_indices);
    // bulk setters used by the binary cache reader (the JSON path goes through addVertex/addIndex like the reference)
    void assign(std::vector<Vector>&& v, std::vector<int>&& idx, std::vector<Vector>&& normals, std::vector<Vector>&& uvs)
    {
        vertices = std::move(v); indices = std::move(idx); vertexNormals = std::move(normals); uvData = std::move(uvs);
    }

    const std::vector<Vector>& getVertices() const { return vertices; }
    const std::vector<int>& getIndices() const { return indices; }
    const std::vector<Vector>& getVertexNormals() const { return vertexNormals; }
    const std::vector<Vector>& getUV() const { return uvData; }
    int getMaterialIndex() const { return materialIndex; }

    void calculateVertexNormals();
    void print() const;

private:
    std::vector<Vector> vertices;
    std::vector<int> indices;
    std::vector<Vector> vertexNormals;
    std::vector<Vector> uvData;
    int materialIndex = 0; // the reference leaves it uninitialised when the key is absent
};

class Camera {
public:
    void pan(float degrees);
    void tilt(float degrees);
    void roll(float degrees);
    void zoom(float amount);
    void rotate(float deltaYawDeg, float deltaPitchDeg);
    void moveForward(float distance);
    void moveRight(float distance);
    void panAroundTarget(float degrees, const Vector& target);

    const Vector& getPosition() const { return position; }
    const Matrix& getRotationMatrix() const { return rotationMatrix; }
    void setRotationMatrix(const Matrix& m) { rotationMatrix = m; }
    void setPosition(const Vector& p) { position = p; }

private:
    Matrix rotationMatrix;
    Vector position;
    float yaw = 0.f; // radians, about world Y
    float pitch = 0.f;
};

class Light {
public:
    Light(const Vector& position_, float intensity_) : position(position_), intensity(intensity_) {}
    const Vector& getPosition() const { return position; }
    float getIntensity() const { return intensity; }

private:
    Vector position;
    float intensity;
};

enum class MaterialType { INVALID, DIFFUSE, REFLECTIVE, REFRACTIVE, CONSTANT };

class Material {
public:
    MaterialType getType() const { return type; }
    const Vector& getAlbedo() const { return albedo; }
    bool isSmoothShading() const { return smoothShading; }
    float getIor() const { return ior; }
    bool isTexture() const { return !textureName.empty(); }
    const std::string& getTextureName() const { return textureName; }

    void setTextureName(const std::string& n) { textureName = n; }
    void setType(MaterialType t) { type = t; }
    void setAlbedo(const Vector& a) { albedo = a; }
    void setSmoothShading(bool s) { smoothShading = s; }
    void setIor(float i) { ior = i; }

private:
    MaterialType type = MaterialType::INVALID;
    Vector albedo;
    std::string textureName;
    bool smoothShading = false;
    float ior = 1.f;
};

// Texture descriptions are parsed and kept as data; the reference's renderer never samples them
// ("CRTTexture hierarchy": out of scope for the hot path, next-row f3).
struct TextureDesc {
    std::string name;
    std::string type; // albedo | edges | checker | bitmap
    Vector colorA, colorB;
    float scalar = 0.f; // edge_width / square_size
    std::string filePath;
    // bitmap texels (rows top to bottom, `channels` bytes per texel), filled by loadBitmap()
    std::vector<unsigned char> pixels;
    int width = 0, height = 0, channels = 0;
    int fileChannels = 0; // what the file held (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA); `channels` is what `pixels` holds (>= 3)

    // CRTTexture::getColor (R/CRTTextureAlbedo.cpp, R/CRTTextureEdges.cpp:9-15, R/CRTTextureChecker.cpp:9-20,
    // R/CRTTextureBitmap.cpp:12-36); pinned by tests/golden/texture_known_answers.json
    Vector getColor(float u = 0.f, float v = 0.f) const;
    // PNG, BMP, TGA, binary PPM / PGM (image_decode.h); the reference decodes through stb_image, which is third party.
    // Relative paths are tried as given and next to `sceneDir`. Throws std::runtime_error.
    void loadBitmap(const std::string& sceneDir);
    uint32_t typeCode() const; // 0 albedo, 1 edges, 2 checker, 3 bitmap (crt_texture.type)
};

struct Settings {
    Vector backgroundColor;
    int imageWidth = 0;
    int imageHeight = 0;
};

class Scene {
public:
    Scene() = default;
    explicit Scene(const std::string& sceneFileName); // throws std::runtime_error (the reference asserts)

    void parseSceneFile(const std::string& sceneFileName);
    const Settings& getSettings() const { return settings; }
    const Camera& getCamera() const { return camera; }
    Camera& getCamera() { return camera; }
    const std::vector<Mesh>& getObjects() const { return geometryObjects; }
    const std::vector<Light>& getLights() const { return lights; }
    const std::vector<Material>& getMaterials() const { return materials; }
    const std::vector<TextureDesc>& getTextures() const { return textures; }
    const TextureDesc* getTextureByName(const std::string& name) const;

    // programmatic construction (synthetic scenes)
    Mesh& addObject() { geometryObjects.emplace_back(); return geometryObjects.back(); }
    void addLight(const Light& l) { lights.push_back(l); }
    void addMaterial(const Material& m) { materials.push_back(m); }
    void addTexture(const TextureDesc& t) { textures.push_back(t); }
    std::vector<Material>& materialsRef() { return materials; }
    std::vector<Mesh>& objectsRef() { return geometryObjects; }
    int textureIndexByName(const std::string& name) const; // -1 when absent
    Settings& settingsRef() { return settings; }

private:
    friend class SceneParser;
    std::vector<Mesh> geometryObjects;
    Camera camera;
    Settings settings;
    std::vector<Light> lights;
    std::vector<Material> materials;
    std::vector<TextureDesc> textures;
};

class SceneParser {
public:
    // .crtscene JSON (R/CRTSceneParser.cpp:407-427), Wavefront .obj or the binary cache .crtbin (extensions);
    // throws std::runtime_error
    static void parseScene(const std::string& sceneFileName, Scene& scene);
    // binary scene cache: everything parseCrtscene produces, including the vertex normals,
    // as raw little-endian arrays -- a 5M-triangle scene is ~0.5 GB of JSON text but 150 MB here and loads at memcpy speed
    static void saveBinary(const std::string& fileName, const Scene& scene);
    static void parseBinary(const std::string& bytes, Scene& scene);
    static void parseCrtscene(const std::string& text, Scene& scene);
    static void parseObj(const std::string& text, Scene& scene);
};

} // namespace crt
