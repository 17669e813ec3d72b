// Host scene layer: math, mesh, camera.  Behaviour follows the reference classes cited in scene.h and is
// pinned by tests/golden/dragon_scene_layer.json (values produced by the reference's own sources).
#include "scene.h"
#include "image_decode.h"

#include <iterator>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace crt {

// ---------------------------------------------------------------------------------------------- Vector
float Vector::length() const { return sqrtf((x * x) + (y * y) + (z * z)); }

void Vector::normalise()
{
    const float len = length();
    x /= len;
    y /= len;
    z /= len;
}

Vector cross(const Vector& a, const Vector& b)
{
    return Vector(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

float dot(const Vector& a, const Vector& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

bool operator==(const Vector& a, const Vector& b)
{
    const float eps = 1e-6f;
    return std::fabs(a.x - b.x) < eps && std::fabs(a.y - b.y) < eps && std::fabs(a.z - b.z) < eps;
}

void Vector::print(std::ostream& os) const { os << "( " << x << ", " << y << ", " << z << " )" << std::endl; }

// ---------------------------------------------------------------------------------------------- Matrix
Matrix::Matrix() : Matrix(1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f) {}

Matrix::Matrix(float c00, float c01, float c02, float c10, float c11, float c12, float c20, float c21, float c22)
{
    const float v[9] = { c00, c01, c02, c10, c11, c12, c20, c21, c22 };
    for (int i = 0; i < 9; i++) m[i / 3][i % 3] = v[i];
}

Matrix operator*(const Matrix& a, const Matrix& b)
{
    Matrix r;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            float acc = 0.0f; // accumulate from zero in p order, like the reference's triple loop
            for (int p = 0; p < 3; p++) acc += a.m[i][p] * b.m[p][j];
            r.m[i][j] = acc;
        }
    }
    return r;
}

Vector operator*(const Vector& v, const Matrix& m)
{
    float out[3];
    for (int i = 0; i < 3; i++) {
        float acc = 0.f;
        for (int j = 0; j < 3; j++) acc += v.getByIndex(j) * m.m[j][i];
        out[i] = acc;
    }
    return Vector(out[0], out[1], out[2]);
}

void Matrix::print() const
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) std::cout << m[i][j] << ' ';
        std::cout << std::endl;
    }
}

// -------------------------------------------------------------------------------------------- Triangle
Triangle::Triangle(const Vector& v0, const Vector& v1, const Vector& v2)
{
    verts[0] = v0;
    verts[1] = v1;
    verts[2] = v2;
    normal = cross(v1 - v0, v2 - v0);
    normal.normalise();
}

bool operator==(const Triangle& a, const Triangle& b)
{
    return a.verts[0] == b.verts[0] && a.verts[1] == b.verts[1] && a.verts[2] == b.verts[2];
}

// ------------------------------------------------------------------------------------------------ Mesh
void Mesh::reserve(size_t n_vertices, size_t n_indices)
{
    vertices.reserve(n_vertices);
    indices.reserve(n_indices);
}

// Unweighted sum of unit face normals per vertex, in triangle order, then normalised (R/CRTMesh.cpp:66-94).
// A vertex no triangle references ends as 0/0 = NaN, as in the reference.
void Mesh::calculateVertexNormals()
{
    vertexNormals.assign(vertices.size(), Vector(0.f, 0.f, 0.f));
    const size_t n = indices.size();
    for (size_t i = 0; i + 2 < n; i += 3) {
        const int a = indices[i], b = indices[i + 1], c = indices[i + 2];
        const Vector fn = Triangle(vertices[a], vertices[b], vertices[c]).getNormal();
        vertexNormals[a] = vertexNormals[a] + fn;
        vertexNormals[b] = vertexNormals[b] + fn;
        vertexNormals[c] = vertexNormals[c] + fn;
    }
    for (Vector& nrm : vertexNormals) nrm.normalise();
}

void Mesh::print() const
{
    for (const Vector& v : vertices) v.print(std::cout);
    for (size_t i = 0; i < indices.size(); i++) {
        if (i % 3 == 0) std::cout << std::endl;
        std::cout << indices[i] << ' ';
    }
}

// ---------------------------------------------------------------------------------------------- Camera
namespace {
// degrees -> radians the way the reference's pan/tilt/roll do it: double product, rounded to float once
inline float toRadians(float degrees) { return static_cast<float>(degrees * (3.14159265358979323846 / 180.f)); }

inline Matrix aboutY(float r) { return Matrix(cosf(r), 0.f, -sinf(r), 0.f, 1.f, 0.f, sinf(r), 0.f, cosf(r)); }
inline Matrix aboutX(float r) { return Matrix(1.f, 0.f, 0.f, 0.f, cosf(r), -sinf(r), 0.f, sinf(r), cosf(r)); }
inline Matrix aboutZ(float r) { return Matrix(cosf(r), -sinf(r), 0.f, sinf(r), cosf(r), 0.f, 0.f, 0.f, 1.f); }

inline Vector column(const Matrix& m, int c) { return Vector(m.get(0, c), m.get(1, c), m.get(2, c)); }
} // namespace

void Camera::pan(float degrees) { rotationMatrix = rotationMatrix * aboutY(toRadians(degrees)); }   // R/CRTCamera.cpp:9-19
void Camera::tilt(float degrees) { rotationMatrix = rotationMatrix * aboutX(toRadians(degrees)); }  // :21-31
void Camera::roll(float degrees) { rotationMatrix = rotationMatrix * aboutZ(toRadians(degrees)); }  // :33-43

// zoom and moveForward both translate along column 2, moveRight along column 0 (R/CRTCamera.cpp:45-55,89-111)
void Camera::zoom(float amount) { position = position + column(rotationMatrix, 2) * amount; }
void Camera::moveForward(float distance) { position = position + column(rotationMatrix, 2) * distance; }
void Camera::moveRight(float distance) { position = position + column(rotationMatrix, 0) * distance; }

// FPS-style look: accumulate yaw/pitch, clamp pitch to +-89 degrees, rebuild the basis with
// columns = right, up, forward (R/CRTCamera.cpp:57-87). The trigonometry runs in double there
// (::cos/::sin on a float argument), so it does here.
void Camera::rotate(float deltaYawDeg, float deltaPitchDeg)
{
    const float deg2rad = 3.14159265359f / 180.0f;
    yaw += deltaYawDeg * deg2rad;
    pitch += deltaPitchDeg * deg2rad;
    const float limit = 89.f * deg2rad;
    pitch = std::clamp(pitch, -limit, limit);

    const double cp = std::cos(static_cast<double>(pitch)), sp = std::sin(static_cast<double>(pitch));
    const double cy = std::cos(static_cast<double>(yaw)), sy = std::sin(static_cast<double>(yaw));
    Vector forward(static_cast<float>(cp * sy), static_cast<float>(sp), static_cast<float>(cp * cy));
    forward.normalise();
    Vector right = cross(Vector(0.f, 1.f, 0.f), forward);
    right.normalise();
    const Vector up = cross(forward, right);
    rotationMatrix = Matrix(right.getX(), up.getX(), forward.getX(),
                            right.getY(), up.getY(), forward.getY(),
                            right.getZ(), up.getZ(), forward.getZ());
}

void Camera::panAroundTarget(float degrees, const Vector& target) // R/CRTCamera.cpp:113-130
{
    const Matrix ry = aboutY(toRadians(degrees));
    position = target + (position - target) * ry;
    rotationMatrix = rotationMatrix * ry;
}

// -------------------------------------------------------------------------------------------- textures
uint32_t TextureDesc::typeCode() const
{
    if (type == "edges") return 1u;
    if (type == "checker") return 2u;
    if (type == "bitmap") return 3u;
    return 0u;
}

Vector TextureDesc::getColor(float u, float v) const
{
    switch (typeCode()) {
    case 1u:
        return (u < scalar || v < scalar || (1 - u - v) < scalar) ? colorA : colorB;
    case 2u: {
        const int n = static_cast<int>(1.f / scalar);
        const int cu = static_cast<int>(std::floor(u * n)), cv = static_cast<int>(std::floor(v * n));
        return ((cu + cv) % 2 == 0) ? colorA : colorB;
    }
    case 3u: {
        if (pixels.empty() || channels < 3) return Vector(0.f, 0.f, 0.f);
        u = std::fmin(std::fmax(u, 0.0f), 1.0f);
        v = std::fmin(std::fmax(v, 0.0f), 1.0f);
        const int row = static_cast<int>((1.0f - v) * (height - 1));
        const int col = static_cast<int>(u * (width - 1));
        const size_t at = (static_cast<size_t>(row) * width + col) * channels;
        return Vector(pixels[at] / 255.0f, pixels[at + 1] / 255.0f, pixels[at + 2] / 255.0f);
    }
    default:
        return colorA;
    }
}

void TextureDesc::loadBitmap(const std::string& sceneDir)
{
    // the path as the scene gives it (the reference hands it to stbi_load unchanged, R/CRTSceneParser.cpp:294-302), then relative to
    // the scene file; a scene written on the reference's platform may spell it with backslashes: tried with '/' for '\\' last
    std::ifstream in(filePath, std::ios::binary);
    if (!in && !sceneDir.empty()) in.open(sceneDir + "/" + filePath, std::ios::binary);
    if (!in && filePath.find('\\') != std::string::npos) {
        std::string portable = filePath;
        for (char& ch : portable)
            if (ch == '\\') ch = '/';
        in.open(portable, std::ios::binary);
        if (!in && !sceneDir.empty()) in.open(sceneDir + "/" + portable, std::ios::binary);
    }
    if (!in) throw std::runtime_error("cannot open texture file '" + filePath + "'");
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    DecodedImage img = decodeImage(file, filePath); // every format the reference's loader reads: image_decode.cpp, jpeg_decode.cpp
    width = img.width;
    height = img.height;
    fileChannels = img.channels;
    if (img.channels >= 3) {
        channels = img.channels;
        pixels = std::move(img.pixels);
        return;
    }
    // Grey (1 channel) and grey + alpha (2): R/CRTTextureBitmap.cpp:24-33 reads buffer[i], buffer[i + 1] and, only when channels > 2,
    // buffer[i + 2] at i = texel * channels -- so its green is the NEXT byte of the file's layout (the next texel's grey, or this
    // texel's alpha) and its blue is 0.  Kept as that: the texels are stored as the RGB triple the reference would return (the
    // read past the last texel, undefined there, gives 0 here), so host, oracle and kernels sample one and the same image.
    const size_t n = static_cast<size_t>(width) * height;
    channels = 3;
    pixels.assign(n * 3, 0);
    for (size_t i = 0; i < n; i++) {
        const size_t at = i * static_cast<size_t>(img.channels);
        pixels[3 * i] = img.pixels[at];
        pixels[3 * i + 1] = at + 1 < img.pixels.size() ? img.pixels[at + 1] : 0;
    }
}

// ----------------------------------------------------------------------------------------------- Scene
int Scene::textureIndexByName(const std::string& name) const
{
    for (size_t i = 0; i < textures.size(); i++)
        if (textures[i].name == name) return static_cast<int>(i);
    return -1;
}

Scene::Scene(const std::string& sceneFileName) { parseSceneFile(sceneFileName); }

void Scene::parseSceneFile(const std::string& sceneFileName) { SceneParser::parseScene(sceneFileName, *this); }

const TextureDesc* Scene::getTextureByName(const std::string& name) const
{
    for (const TextureDesc& t : textures)
        if (t.name == name) return &t;
    return nullptr;
}

} // namespace crt
