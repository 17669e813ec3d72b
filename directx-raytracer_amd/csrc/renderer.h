// crt::Renderer -- C++ host class with the public interface of DXRTRenderer (R/DXRTRenderer.h:74-94):
//   prepareForRendering / prepareForRayTracing / render / renderFrame / stopRendering / changeShadingMode / getScene
// so that the reference's app loop (R/DXRTApp.cpp:29-120) can drive the MI355X renderer unchanged.  It owns the
// scene like the reference (std::unique_ptr<CRTScene>, R/DXRTRenderer.h:242) and talks to the GPU only through
// the C ABI of include/crt_hip.h.  The window handle of the reference is replaced by a scene path + device id;
// the swap chain by a host frame buffer (R8G8B8A8) that can be written as PPM.
#pragma once

#include "../../include/crt_hip.h"
#include "scene.h"

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace crt {

class Renderer {
public:
    Renderer();
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    // R/DXRTRenderer.cpp:44-62: device + scene + geometry upload + acceleration structure.
    // sceneFile defaults to the path the reference hard-codes (R/DXRTRenderer.cpp:245). Throws std::runtime_error.
    void prepareForRendering(const std::string& sceneFile = "Scenes/Dragon.crtscene", int deviceId = 0);
    void prepareForRendering(std::unique_ptr<Scene> ownedScene, int deviceId = 0);
    void prepareForRayTracing(); // root signature / PSO / SBT have no HIP counterpart: kept as a no-op
    void render();               // R/DXRTRenderer.cpp:35-42: one frame
    void renderFrame();          // R/DXRTRenderer.cpp:1370-1408: camera CB update + dispatch + sync
    void stopRendering();        // drains the stream (the reference's body is empty, R/DXRTRenderer.cpp:1354-1357)
    void changeShadingMode(uint32_t value);
    Scene& getScene();

    // headless replacements for the swap chain
    void setFrameSize(uint32_t width, uint32_t height); // reference: fixed 1920x1080 (R/DXRTRenderer.cpp:1348-1349)
    uint32_t getFrameWidth() const { return width; }
    uint32_t getFrameHeight() const { return height; }
    const std::vector<uint8_t>& getFrame() const { return frame; } // RGBA8, row-major, top-left origin
    void writePPM(const std::string& path) const;
    void writePNG(const std::string& path) const; // 8-bit RGB, stored (uncompressed) deflate blocks
    const crt_frame_stats& getLastFrameStats() const { return stats; }
    void setCounting(bool on);
    void setOption(const char* name, int value); // crt_set_option: "spp", "max_bounces", "seed", "phong_ks", "phong_exponent", ...

    // N GPUs, one process each (no reference counterpart): join the RCCL communicator of an N-rank run.  Rank 0 creates the
    // 128-byte id and publishes it as `idFile` (written under a temporary name, then renamed); the other ranks wait for the
    // file.  Afterwards renderFrame() renders this rank's tiles, gathers and de-interleaves: every rank holds the frame.
    void joinRanks(uint32_t rank, uint32_t nRanks, const std::string& idFile, unsigned long long nonce = 0);
    // the same with shared host memory as the transport (crt_comm_init_host): ranks that share one GPU -- a rehearsal, not a measurement
    void joinRanksThroughHostMemory(uint32_t rank, uint32_t nRanks, unsigned long long nonce);
    uint32_t getRank() const { return rank; }
    uint32_t getRankCount() const { return nRanks; }

private:
    crt_ctx* ctx = nullptr;
    std::unique_ptr<Scene> scene;
    uint32_t width = 1920, height = 1080;
    uint32_t currentShadingMode = 0; // R/DXRTRenderer.h:246
    bool isChangedShadingMode = true;
    std::vector<uint8_t> frame;
    crt_frame_stats stats{};
    uint32_t rank = 0, nRanks = 0; // nRanks = 0: single-GPU path (crt_render_frame)
    void uploadScene();
    void check(int rc, const char* what) const;
};

} // namespace crt
