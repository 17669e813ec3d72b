// INTEGRATION.md variant B: the replacement for the reference's DXRTRenderer.h -- same class name, same public section
// (R/DXRTRenderer.h:74-94), the D3D12 / DXR private section (R/DXRTRenderer.h:95-250) shrunk to a context handle.
// It keeps the reference's OWN scene layer: CRTScene.h below is the reference's header, compiled in place; nothing of it
// is copied here.  Builds only where the reference tree is mounted (oracle/Makefile, target _ref/ref_shim_render).
#pragma once

#include "CRTScene.h" // the reference's (-I R/)

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

struct crt_ctx;

class DXRTRenderer {
public:
    DXRTRenderer();
    ~DXRTRenderer();
    void render();
    void renderFrame();
    // the reference takes the window handle of its Qt viewport (R/DXRTRenderer.h:84); headless: any value, unused
    void prepareForRendering(void* hwnd);
    void prepareForRayTracing();
    void stopRendering();
    void changeShadingMode(uint32_t value);
    CRTScene& getScene();

    // what the swap chain's Present showed (R/DXRTRenderer.cpp:505,523): the frame, RGBA8 row-major, 1920x1080 like the
    // reference's literals (R/DXRTRenderer.cpp:1348-1349, hlsl:24-25) unless setFrameSize is called first
    void setFrameSize(uint32_t w, uint32_t h) { width = w; height = h; }
    const std::vector<uint8_t>& getFrame() const { return frame; }
    // the reference hard-codes "Scenes/Dragon.crtscene" (R/DXRTRenderer.cpp:245); settable here because the test box holds
    // the scene elsewhere
    void setSceneFile(const std::string& path) { sceneFile = path; }

private:
    crt_ctx* ctx = nullptr;
    std::unique_ptr<CRTScene> scene; // R/DXRTRenderer.h:242
    std::string sceneFile = "Scenes/Dragon.crtscene";
    std::vector<uint8_t> frame;
    uint32_t width = 1920, height = 1080;
    uint32_t currentShadingMode = 0; // R/DXRTRenderer.h:246
    bool isChangedShadingMode = true; // R/DXRTRenderer.h:247
};
