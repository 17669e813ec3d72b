// What DXRTApp does with its DXRTRenderer member (R/DXRTApp.cpp:29-120), without Qt: init, then per frame the camera calls
// of the input handlers on the REFERENCE's CRTCamera, renderFrame, and the frame written out as PPM.  Prints the camera each
// frame was rendered with, so that a checker can render the same view elsewhere.
//   ref_shim_render <scene.crtscene> <out_prefix> [WxH]
#include "DXRTRenderer.h"

#include <cstdio>
#include <exception>
#include <string>

static void writePPM(const std::string& path, const std::vector<uint8_t>& rgba, uint32_t w, uint32_t h)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (size_t i = 0; i < static_cast<size_t>(w) * h; i++) std::fwrite(&rgba[4 * i], 1, 3, f);
    std::fclose(f);
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: ref_shim_render <scene.crtscene> <out_prefix> [WxH]\n"); return 2; }
    uint32_t w = 1920, h = 1080;
    if (argc > 3 && std::sscanf(argv[3], "%ux%u", &w, &h) != 2) return 2;
    try {
        DXRTRenderer renderer;
        renderer.setSceneFile(argv[1]);
        renderer.setFrameSize(w, h);
        renderer.prepareForRendering(nullptr); // DXRTApp::init, R/DXRTApp.cpp:12
        renderer.prepareForRayTracing();
        const uint32_t modes[3] = { 0, 3, 100 };
        for (int f = 0; f < 3; f++) {
            CRTCamera& cam = renderer.getScene().getCamera();
            if (f == 1) { cam.rotate(20.0f, -5.0f); cam.moveForward(-2.0f); } // rotateCamera / W key, R/DXRTApp.cpp:36-41, 96-97
            if (f == 2) { cam.moveRight(1.5f); cam.zoom(0.25f); }             // D key / wheel, R/DXRTApp.cpp:105-106, 43-46
            renderer.changeShadingMode(modes[f]);                             // combo box, R/DXRTMainWindow.cpp:114-121
            renderer.renderFrame();
            const CRTMatrix& r = cam.getRotationMatrix();
            std::printf("frame %d mode %u camera %.9g %.9g %.9g", f, modes[f], cam.getPosition().getX(), cam.getPosition().getY(), cam.getPosition().getZ());
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) std::printf(" %.9g", r.get(i, j));
            std::printf("\n");
            writePPM(std::string(argv[2]) + "_" + std::to_string(f) + ".ppm", renderer.getFrame(), w, h);
        }
        renderer.stopRendering();
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "ref_shim_render: %s\n", ex.what());
        return 1;
    }
    return 0;
}
