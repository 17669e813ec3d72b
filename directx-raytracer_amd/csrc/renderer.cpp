#include "renderer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
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

void Renderer::joinRanksThroughHostMemory(uint32_t rankIn, uint32_t nRanksIn, unsigned long long nonce)
{
    if (!ctx) throw std::runtime_error("joinRanksThroughHostMemory before prepareForRendering");
    char name[64];
    std::snprintf(name, sizeof(name), "/crt_render_%016llx", nonce); // the launch's nonce names the shared-memory object
    check(crt_comm_init_host(ctx, rankIn, nRanksIn, name), "crt_comm_init_host");
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

// PNG, 8-bit RGB, scanlines unfiltered, the zlib stream made of stored blocks: every PNG reader accepts it and the writer needs
// no compressor (an 8 MB frame becomes an 8 MB file; the viewer's frames are for looking at and for regression checks).
void Renderer::writePNG(const std::string& path) const
{
    static uint32_t crcTable[256];
    static bool haveTable = false;
    if (!haveTable) {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crcTable[n] = c;
        }
        haveTable = true;
    }
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = uint8_t(v >> 24); p[1] = uint8_t(v >> 16); p[2] = uint8_t(v >> 8); p[3] = uint8_t(v); };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& body) {
        uint8_t head[8];
        be32(head, static_cast<uint32_t>(body.size()));
        std::memcpy(head + 4, type, 4);
        uint32_t c = 0xFFFFFFFFu;
        for (int i = 4; i < 8; i++) c = crcTable[(c ^ head[i]) & 255u] ^ (c >> 8);
        for (uint8_t b : body) c = crcTable[(c ^ b) & 255u] ^ (c >> 8);
        uint8_t tail[4];
        be32(tail, c ^ 0xFFFFFFFFu);
        std::fwrite(head, 1, 8, f);
        if (!body.empty()) std::fwrite(body.data(), 1, body.size(), f);
        std::fwrite(tail, 1, 4, f);
    };
    static const uint8_t magic[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    std::fwrite(magic, 1, 8, f);
    std::vector<uint8_t> ihdr(13);
    be32(ihdr.data(), width);
    be32(ihdr.data() + 4, height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0; // 8 bits, RGB, deflate, filter method 0, not interlaced
    chunk("IHDR", ihdr);
    // raw image data: a filter byte (0) and 3 bytes per pixel for every scanline
    std::vector<uint8_t> raw;
    raw.reserve(static_cast<size_t>(height) * (1 + 3 * static_cast<size_t>(width)));
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);
        const uint8_t* src = frame.data() + static_cast<size_t>(y) * width * 4;
        for (uint32_t x = 0; x < width; x++) raw.insert(raw.end(), src + 4 * x, src + 4 * x + 3);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0; // Adler-32 of the raw data
    for (size_t at = 0; at < raw.size() || at == 0; at += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + n >= raw.size() ? 1 : 0); // last block?
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8)); z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + static_cast<std::ptrdiff_t>(at), raw.begin() + static_cast<std::ptrdiff_t>(at + n));
        for (size_t i = at; i < at + n; i++) {
            a = (a + raw[i]) % 65521u;
            b = (b + a) % 65521u;
        }
        if (raw.empty()) break;
    }
    uint8_t adler[4];
    be32(adler, (b << 16) | a);
    z.insert(z.end(), adler, adler + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::fclose(f);
}

} // namespace crt
