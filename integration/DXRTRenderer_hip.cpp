// INTEGRATION.md variant B, compiled: DXRTRenderer's members bound 1:1 to the C ABI of include/crt_hip.h, over the
// reference's own CRT* classes (compiled in place from the reference tree; only their public API is called).
// Replaces R/DXRTRenderer.cpp; each member cites the lines it stands in for.
#include "DXRTRenderer.h"

#include "../include/crt_hip.h"

#include <stdexcept>

static_assert(sizeof(CRTVector) == 12, "createVertexBuffers memcpy's CRTVector as 3 floats (R/DXRTRenderer.cpp:391-392)");

DXRTRenderer::DXRTRenderer() = default;

DXRTRenderer::~DXRTRenderer()
{
    if (ctx) crt_destroy(ctx);
}

void DXRTRenderer::prepareForRendering(void*) // R/DXRTRenderer.cpp:44-62
{
    scene = std::make_unique<CRTScene>(sceneFile); // createScene, R/DXRTRenderer.cpp:243-246
    if (crt_create(&ctx, 0) != CRT_OK) throw std::runtime_error(crt_last_error(nullptr));
    std::vector<crt_mesh_view> meshes;
    for (const CRTMesh& m : scene->getObjects()) { // the loop of createVertexBuffers / createIndexBuffers, :381-392 / :304-315
        crt_mesh_view v{};
        v.xyz = reinterpret_cast<const float*>(m.getVertices().data());
        v.idx = reinterpret_cast<const uint32_t*>(m.getIndices().data()); // vector<int> reinterpreted like :314-315
        v.normals = m.getVertexNormals().size() == m.getVertices().size() ? reinterpret_cast<const float*>(m.getVertexNormals().data()) : nullptr;
        v.uvs = (!m.getUV().empty() && m.getUV().size() == m.getVertices().size()) ? reinterpret_cast<const float*>(m.getUV().data()) : nullptr;
        v.n_vertices = static_cast<uint32_t>(m.getVertices().size());
        v.n_triangles = static_cast<uint32_t>(m.getIndices().size() / 3);
        v.material_index = m.getMaterialIndex();
        meshes.push_back(v);
    }
    std::vector<crt_light> lights; // parsed but never read by the reference's renderer; mode 100 uses them
    for (const CRTLight& l : scene->getLights())
        lights.push_back(crt_light{ { l.getPosition().getX(), l.getPosition().getY(), l.getPosition().getZ() }, l.getIntensity() });
    std::vector<crt_material> mats;
    for (const CRTMaterial& m : scene->getMaterials())
        mats.push_back(crt_material{ { m.getAlbedo().getX(), m.getAlbedo().getY(), m.getAlbedo().getZ() }, static_cast<uint32_t>(m.getType()),
                                     m.isSmoothShading() ? 1u : 0u, m.getIor(), -1 }); // textures: see INTEGRATION.md (private fields)
    // replaces createAccelerationStructures, R/DXRTRenderer.cpp:548-806
    if (crt_upload_scene(ctx, meshes.data(), static_cast<uint32_t>(meshes.size()), lights.data(), static_cast<uint32_t>(lights.size()),
                         mats.data(), static_cast<uint32_t>(mats.size())) != CRT_OK)
        throw std::runtime_error(crt_last_error(ctx));
}

void DXRTRenderer::prepareForRayTracing() {} // root signature / PSO / SBT (R/DXRTRenderer.cpp:64-70): nothing to do

void DXRTRenderer::render() { renderFrame(); } // R/DXRTRenderer.cpp:35-42

void DXRTRenderer::renderFrame() // R/DXRTRenderer.cpp:1370-1408
{
    const CRTCamera& cam = scene->getCamera();
    const CRTMatrix& r = cam.getRotationMatrix();
    const float pos[3] = { cam.getPosition().getX(), cam.getPosition().getY(), cam.getPosition().getZ() };
    float rot[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) rot[3 * i + j] = r.get(i, j); // updateCameraCB, R/DXRTRenderer.cpp:259-264
    crt_set_camera(ctx, pos, rot);
    if (isChangedShadingMode) { // updateDebugCB only when dirty, R/DXRTRenderer.cpp:457-463
        crt_set_shading_mode(ctx, currentShadingMode);
        isChangedShadingMode = false;
    }
    frame.resize(static_cast<size_t>(width) * height * 4);
    // DispatchRays + fence wait, R/DXRTRenderer.cpp:1405, 521-527
    if (crt_render_frame(ctx, width, height, frame.data(), nullptr, nullptr, nullptr, nullptr, nullptr) != CRT_OK)
        throw std::runtime_error(crt_last_error(ctx));
}

void DXRTRenderer::changeShadingMode(uint32_t value) // R/DXRTRenderer.cpp:1359-1363
{
    currentShadingMode = value;
    isChangedShadingMode = true;
}

void DXRTRenderer::stopRendering() // the reference's body is empty (R/DXRTRenderer.cpp:1354-1357); here the stream is drained
{
    if (ctx) crt_synchronize(ctx);
}

CRTScene& DXRTRenderer::getScene() { return *scene; } // R/DXRTRenderer.cpp:1365-1368
