#include "renderer.h"

#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <thread>

namespace crt {

Renderer::Renderer() = default;

Renderer::~Renderer()
{
    if (ctx) crt_destroy(ctx);
}

void Renderer::check(int rc, const char* what) const
{
    if (rc != CRT_OK) throw std::runtime_error(std::string(what) + ": " + crt_last_error(ctx));
}

void Renderer::prepareForRendering(const std::string& sceneFile, int deviceId)
{
    prepareForRendering(std::make_unique<Scene>(sceneFile), deviceId);
}

void Renderer::prepareForRendering(std::unique_ptr<Scene> ownedScene, int deviceId)
{
    scene = std::move(ownedScene);
    if (!ctx) {
        const int rc = crt_create(&ctx, deviceId);
        if (rc != CRT_OK) throw std::runtime_error(std::string("crt_create: ") + crt_last_error(nullptr));
    }
    uploadScene();
    prepareForRayTracing();
}

void Renderer::prepareForRayTracing() {}

void Renderer::uploadScene()
{
    std::vector<crt_mesh_view> meshes;
    for (const Mesh& m : scene->getObjects()) {
        crt_mesh_view v{};
        v.xyz = m.getVertices().empty() ? nullptr : m.getVertices().data()->data();
        v.idx = reinterpret_cast<const uint32_t*>(m.getIndices().data());
        v.normals = (m.getVertexNormals().size() == m.getVertices().size() && !m.getVertices().empty())
                        ? m.getVertexNormals().data()->data() : nullptr;
        v.uvs = (m.getUV().size() == m.getVertices().size() && !m.getUV().empty()) ? m.getUV().data()->data() : nullptr;
        v.n_vertices = static_cast<uint32_t>(m.getVertices().size());
        v.n_triangles = static_cast<uint32_t>(m.getIndices().size() / 3);
        v.material_index = m.getMaterialIndex();
        meshes.push_back(v);
    }
    std::vector<crt_light> lights;
    for (const Light& l : scene->getLights())
        lights.push_back(crt_light{ { l.getPosition().getX(), l.getPosition().getY(), l.getPosition().getZ() }, l.getIntensity() });
    std::vector<crt_material> mats;
    for (const Material& m : scene->getMaterials())
        mats.push_back(crt_material{ { m.getAlbedo().getX(), m.getAlbedo().getY(), m.getAlbedo().getZ() },
                                     static_cast<uint32_t>(m.getType()), m.isSmoothShading() ? 1u : 0u, m.getIor(),
                                     m.isTexture() ? scene->textureIndexByName(m.getTextureName()) : -1 });
    check(crt_upload_scene(ctx, meshes.data(), static_cast<uint32_t>(meshes.size()), lights.data(),
                           static_cast<uint32_t>(lights.size()), mats.data(), static_cast<uint32_t>(mats.size())),
          "crt_upload_scene");
    std::vector<crt_texture> tex;
    for (const TextureDesc& t : scene->getTextures()) {
        crt_texture x{};
        x.type = t.typeCode();
        x.color_a[0] = t.colorA.getX(); x.color_a[1] = t.colorA.getY(); x.color_a[2] = t.colorA.getZ();
        x.color_b[0] = t.colorB.getX(); x.color_b[1] = t.colorB.getY(); x.color_b[2] = t.colorB.getZ();
        x.scalar = t.scalar;
        x.pixels = t.pixels.empty() ? nullptr : t.pixels.data();
        x.width = static_cast<uint32_t>(t.width);
        x.height = static_cast<uint32_t>(t.height);
        x.channels = static_cast<uint32_t>(t.channels);
        tex.push_back(x);
    }
    check(crt_set_textures(ctx, tex.data(), static_cast<uint32_t>(tex.size())), "crt_set_textures");
}

void Renderer::render() { renderFrame(); }

void Renderer::renderFrame()
{
    if (!ctx || !scene) throw std::runtime_error("renderFrame before prepareForRendering");
    if (isChangedShadingMode) { // frameBegin: updateDebugCB only when dirty (R/DXRTRenderer.cpp:457-463)
        check(crt_set_shading_mode(ctx, currentShadingMode), "crt_set_shading_mode");
        isChangedShadingMode = false;
    }
    // updateCameraCB every frame (R/DXRTRenderer.cpp:464)
    check(crt_set_camera(ctx, scene->getCamera().getPosition().data(), scene->getCamera().getRotationMatrix().data()), "crt_set_camera");
    frame.resize(static_cast<size_t>(width) * height * 4);
    if (nRanks) check(crt_render_frame_distributed(ctx, width, height, nullptr, frame.data(), &stats), "crt_render_frame_distributed");
    else check(crt_render_frame(ctx, width, height, frame.data(), nullptr, nullptr, nullptr, nullptr, &stats), "crt_render_frame");
}

void Renderer::setOption(const char* name, int value)
{
    if (!ctx) throw std::runtime_error("setOption before prepareForRendering");
    check(crt_set_option(ctx, name, value), name);
}

// The id file holds {nonce, communicator id}.  The nonce names the launch (crt_render --ranks draws a fresh one per run): a file
// left behind by an earlier run, or by another launch that was given the same path, carries a different nonce and is waited out
// like a file that is not there yet instead of being taken for this run's id (ranks joining a dead communicator block for good).
void Renderer::joinRanks(uint32_t rankIn, uint32_t nRanksIn, const std::string& idFile, unsigned long long nonce)
{
    if (!ctx) throw std::runtime_error("joinRanks before prepareForRendering");
    unsigned char id[CRT_COMM_ID_BYTES];
    if (rankIn == 0) {
        if (crt_comm_unique_id(id) != CRT_OK) throw std::runtime_error(std::string("crt_comm_unique_id: ") + crt_last_error(nullptr));
        const std::string tmp = idFile + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        const bool ok = f && std::fwrite(&nonce, 1, sizeof(nonce), f) == sizeof(nonce) && std::fwrite(id, 1, sizeof(id), f) == sizeof(id);
        if (f) std::fclose(f);
        if (!ok) throw std::runtime_error("cannot write '" + tmp + "'");
        if (std::rename(tmp.c_str(), idFile.c_str()) != 0) throw std::runtime_error("cannot publish '" + idFile + "'");
    } else {
        bool have = false;
        for (int tries = 0; tries < 6000 && !have; tries++) { // up to 60 s for rank 0 to come up
            FILE* f = std::fopen(idFile.c_str(), "rb");
            if (f) {
                unsigned long long seen = 0;
                have = std::fread(&seen, 1, sizeof(seen), f) == sizeof(seen) && seen == nonce && std::fread(id, 1, sizeof(id), f) == sizeof(id);
                std::fclose(f);
            }
            if (!have) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!have) throw std::runtime_error("rank " + std::to_string(rankIn) + ": no communicator id of this launch in '" + idFile + "'");
    }
    check(crt_comm_init(ctx, rankIn, nRanksIn, id), "crt_comm_init");
    rank = rankIn;
    nRanks = nRanksIn;
}

void Renderer::stopRendering()
{
    if (ctx) check(crt_synchronize(ctx), "crt_synchronize");
}

void Renderer::changeShadingMode(uint32_t value)
{
    currentShadingMode = value;
    isChangedShadingMode = true;
}

Scene& Renderer::getScene()
{
    if (!scene) throw std::runtime_error("getScene before prepareForRendering");
    return *scene;
}

void Renderer::setFrameSize(uint32_t w, uint32_t h)
{
    width = w;
    height = h;
}

void Renderer::setCounting(bool on)
{
    if (ctx) check(crt_set_counting(ctx, on ? 1 : 0), "crt_set_counting");
}

void Renderer::writePPM(const std::string& path) const
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row(static_cast<size_t>(width) * 3);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* src = frame.data() + static_cast<size_t>(y) * width * 4;
        for (uint32_t x = 0; x < width; x++) {
            row[3 * x] = src[4 * x];
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + 2];
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
}

} // namespace crt
