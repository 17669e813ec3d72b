// crt_render -- headless driver: the MI355X replacement for DXRTApp's idle-tick loop (R/DXRTApp.cpp:92-120)
// with the Qt window, input widgets and swap chain removed.  Loads a .crtscene / .obj, plays a scripted camera
// path through the same Camera calls the reference's input handlers make (rotate / zoom / moveForward /
// moveRight; R/DXRTApp.cpp:36-47,92-107), renders N frames, prints ms/frame and Mray/s (the reference shows an
// FPS label, R/DXRTApp.cpp:82-90) and optionally writes PPM images.
#include "renderer.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

static void usage()
{
    std::fprintf(stderr,
                 "usage: crt_render <scene.crtscene|scene.obj> [--mode M] [--size WxH] [--frames N] [--device D]\n"
                 "                  [--orbit DEG_PER_FRAME] [--forward UNITS_PER_FRAME] [--out prefix] [--count]\n");
}

int main(int argc, char** argv)
{
    if (argc < 2) { usage(); return 2; }
    std::string scenePath = argv[1], out;
    uint32_t mode = 0, w = 1920, h = 1080;
    int frames = 1, device = 0;
    float orbit = 0.f, forward = 0.f;
    bool count = false;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (a == "--mode") mode = static_cast<uint32_t>(std::atoi(next("--mode")));
        else if (a == "--size") { if (std::sscanf(next("--size"), "%ux%u", &w, &h) != 2) { usage(); return 2; } }
        else if (a == "--frames") frames = std::atoi(next("--frames"));
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--orbit") orbit = static_cast<float>(std::atof(next("--orbit")));
        else if (a == "--forward") forward = static_cast<float>(std::atof(next("--forward")));
        else if (a == "--out") out = next("--out");
        else if (a == "--count") count = true;
        else { usage(); return 2; }
    }
    try {
        crt::Renderer renderer;
        renderer.prepareForRendering(scenePath, device);
        renderer.setFrameSize(w, h);
        renderer.changeShadingMode(mode);
        renderer.setCounting(count);
        double sumMs = 0.0;
        for (int f = 0; f < frames; f++) {
            if (f > 0) { // scripted input, same calls as the reference's handlers
                if (orbit != 0.f) renderer.getScene().getCamera().rotate(orbit, 0.f);
                if (forward != 0.f) renderer.getScene().getCamera().moveForward(-forward);
            }
            renderer.renderFrame();
            const crt_frame_stats& st = renderer.getLastFrameStats();
            sumMs += st.kernel_ms;
            const double rays = static_cast<double>(st.rays_primary + st.rays_shadow);
            std::printf("frame %d: kernel %.3f ms, call %.3f ms, %.1f Mray/s", f, st.kernel_ms, st.total_ms, rays / st.kernel_ms * 1e-3);
            if (count) std::printf(", nodes %llu, tris %llu, shadow rays %llu", (unsigned long long)st.nodes_visited,
                                   (unsigned long long)st.tris_tested, (unsigned long long)st.rays_shadow);
            std::printf("\n");
            if (!out.empty()) renderer.writePPM(out + "_" + std::to_string(f) + ".ppm");
        }
        std::printf("average kernel %.3f ms/frame over %d frames (%ux%u, mode %u)\n", sumMs / frames, frames, w, h, mode);
        renderer.stopRendering();
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "crt_render: %s\n", ex.what());
        return 1;
    }
    return 0;
}
