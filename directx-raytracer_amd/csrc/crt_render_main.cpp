// crt_render -- headless driver: the MI355X replacement for DXRTApp's idle-tick loop (R/DXRTApp.cpp:92-120)
// with the Qt window, input widgets and swap chain removed.  Loads a .crtscene / .obj / .crtbin, plays a scripted
// camera path through the same Camera calls the reference's input handlers make -- W/S -> moveForward, A/D ->
// moveRight (R/DXRTApp.cpp:91-107), mouse -> rotate (R/DXRTViewportWidget.cpp:50-72), wheel -> zoom (:74-78) -- and
// the shading-mode switch of its combo box (R/DXRTMainWindow.cpp:114-121), renders N frames, prints ms/frame and
// Mray/s (the reference shows an FPS label, R/DXRTApp.cpp:82-90) and optionally writes PPM or PNG images.
//
// --ranks N: one process per GPU, launched from here, no Python: the parent (which never touches the GPU) starts N
// copies of itself with --rank r; rank r renders on device r, the frame is tile-partitioned and assembled with one
// RCCL all-gather per frame (crt_render_frame_distributed); rank 0 prints and writes the images.
#include "renderer.h"

#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

static void usage()
{
    std::fprintf(stderr,
                 "usage: crt_render <scene.crtscene|scene.obj|scene.crtbin> [--mode M] [--size WxH] [--frames N] [--device D]\n"
                 "   per-frame camera script (applied before every frame but the first):\n"
                 "       [--orbit DEG] [--pitch DEG] [--forward UNITS] [--right UNITS] [--zoom AMOUNT]\n"
                 "       [--path FILE]   one line per frame, commands separated by ';':\n"
                 "                       rotate YAW PITCH | forward D | right D | zoom A | pan DEG | tilt DEG | roll DEG | mode M\n"
                 "   [--mode-at FRAME:MODE]...  switch the shading mode from that frame on\n"
                 "   [--spp N] [--bounces N] [--seed N]   mode 200 (path tracing)\n"
                 "   [--phong KS_PERMILLE:EXPONENT]       mode 100 specular term\n"
                 "   [--out prefix] [--png] [--count]      frames as prefix_N.ppm, or prefix_N.png with --png\n"
                 "   [--ranks N [--device-base D] [--id-file PATH]]   N processes / GPUs, RCCL gather per frame\n"
                 "   [--host-exchange [--same-device]]   with --ranks: tiles through shared host memory instead of RCCL; --same-device puts every\n"
                 "                                       rank on --device (a rehearsal of the multi-rank path on one GPU)\n");
}

namespace {
struct Args {
    std::string scene, out, pathFile, idFile;
    uint32_t mode = 0, w = 1920, h = 1080;
    int frames = 1, device = 0, deviceBase = 0, ranks = 0, rank = -1;
    int spp = -1, bounces = -1, seed = -1, phongKs = -1, phongExp = -1;
    float orbit = 0.f, pitch = 0.f, forward = 0.f, right = 0.f, zoom = 0.f;
    bool count = false, png = false, hostExchange = false, sameDevice = false;
    unsigned long long nonce = 0; // names the launch in the id file (set by the --ranks parent)
    std::map<int, uint32_t> modeAt;
};

// one line of a --path file applied to the camera / renderer: the calls the reference's handlers make
void applyScriptLine(const std::string& line, crt::Renderer& renderer)
{
    std::stringstream cmds(line);
    std::string cmd;
    while (std::getline(cmds, cmd, ';')) {
        std::stringstream ss(cmd);
        std::string op;
        if (!(ss >> op) || op[0] == '#') continue;
        crt::Camera& cam = renderer.getScene().getCamera();
        float a = 0.f, b = 0.f;
        if (op == "rotate" && (ss >> a >> b)) cam.rotate(a, b);
        else if (op == "forward" && (ss >> a)) cam.moveForward(-a); // W: moveForward(-speed * dt), R/DXRTApp.cpp:96-97
        else if (op == "right" && (ss >> a)) cam.moveRight(a);      // D: moveRight(+speed * dt), R/DXRTApp.cpp:105-106
        else if (op == "zoom" && (ss >> a)) cam.zoom(a);
        else if (op == "pan" && (ss >> a)) cam.pan(a);
        else if (op == "tilt" && (ss >> a)) cam.tilt(a);
        else if (op == "roll" && (ss >> a)) cam.roll(a);
        else if (op == "mode" && (ss >> a)) renderer.changeShadingMode(static_cast<uint32_t>(a));
        else throw std::runtime_error("camera path: cannot parse '" + cmd + "'");
    }
}

int runRank(const Args& a)
{
    crt::Renderer renderer;
    const int device = (a.ranks > 0 && !a.sameDevice) ? a.deviceBase + a.rank : a.device;
    renderer.prepareForRendering(a.scene, device);
    renderer.setFrameSize(a.w, a.h);
    renderer.changeShadingMode(a.mode);
    renderer.setCounting(a.count);
    if (a.spp >= 0) renderer.setOption("spp", a.spp);
    if (a.bounces >= 0) renderer.setOption("max_bounces", a.bounces);
    if (a.seed >= 0) renderer.setOption("seed", a.seed);
    if (a.phongKs >= 0) renderer.setOption("phong_ks", a.phongKs);
    if (a.phongExp >= 0) renderer.setOption("phong_exponent", a.phongExp);
    if (a.ranks > 0 && a.hostExchange) renderer.joinRanksThroughHostMemory(static_cast<uint32_t>(a.rank), static_cast<uint32_t>(a.ranks), a.nonce);
    else if (a.ranks > 0) renderer.joinRanks(static_cast<uint32_t>(a.rank), static_cast<uint32_t>(a.ranks), a.idFile, a.nonce);
    const bool talk = a.ranks <= 0 || a.rank == 0;
    std::vector<std::string> script;
    if (!a.pathFile.empty()) {
        std::ifstream f(a.pathFile);
        if (!f) throw std::runtime_error("cannot open camera path '" + a.pathFile + "'");
        for (std::string line; std::getline(f, line);) script.push_back(line);
    }
    double sumMs = 0.0;
    uint32_t mode = a.mode;
    for (int f = 0; f < a.frames; f++) {
        if (f > 0) { // scripted input, same calls as the reference's handlers
            crt::Camera& cam = renderer.getScene().getCamera();
            if (a.orbit != 0.f || a.pitch != 0.f) cam.rotate(a.orbit, a.pitch);
            if (a.forward != 0.f) cam.moveForward(-a.forward);
            if (a.right != 0.f) cam.moveRight(a.right);
            if (a.zoom != 0.f) cam.zoom(a.zoom);
        }
        if (static_cast<size_t>(f) < script.size()) applyScriptLine(script[static_cast<size_t>(f)], renderer);
        const auto sw = a.modeAt.find(f);
        if (sw != a.modeAt.end()) {
            mode = sw->second;
            renderer.changeShadingMode(mode);
        }
        renderer.renderFrame();
        const crt_frame_stats& st = renderer.getLastFrameStats();
        sumMs += st.kernel_ms;
        if (!talk) continue;
        const double rays = static_cast<double>(st.rays_primary + st.rays_shadow);
        std::printf("frame %d: kernel %.3f ms, call %.3f ms, %.1f Mray/s", f, st.kernel_ms, st.total_ms, rays / st.kernel_ms * 1e-3);
        if (a.ranks > 0) std::printf(" (rank 0's tile share of %d ranks)", a.ranks);
        if (a.count) std::printf(", nodes %llu, tris %llu, shadow rays %llu", (unsigned long long)st.nodes_visited,
                                 (unsigned long long)st.tris_tested, (unsigned long long)st.rays_shadow);
        std::printf("\n");
        if (!a.out.empty()) {
            if (a.png) renderer.writePNG(a.out + "_" + std::to_string(f) + ".png");
            else renderer.writePPM(a.out + "_" + std::to_string(f) + ".ppm");
        }
    }
    if (talk) std::printf("average kernel %.3f ms/frame over %d frames (%ux%u, mode %u)\n", sumMs / a.frames, a.frames, a.w, a.h, a.mode);
    renderer.stopRendering();
    return 0;
}

// parent of an N-rank run: starts the rank processes BEFORE anything here touches the GPU and waits for them.  The first rank
// that fails (non-zero exit or a signal: bad device index, scene that does not load, out of memory) ends the run: its peers
// would otherwise sit in ncclCommInitRank / ncclAllGather for good, so they are sent SIGTERM, then SIGKILL, and the parent
// reports which rank went first.  (Ending the children is all that happens: nothing that has touched a GPU is re-executed.)
int launchRanks(const Args& a, int argc, char** argv)
{
    std::string idFile = a.idFile;
    if (idFile.empty()) idFile = "/tmp/crt_render_comm_" + std::to_string(static_cast<long>(getpid())) + ".id";
    std::remove(idFile.c_str()); // whatever an earlier run left under this name ...
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    // ... and a per-launch nonce in the file, for ranks that find a file another launch writes under the same name
    const unsigned long long nonce = (static_cast<unsigned long long>(ts.tv_sec) << 30) ^ static_cast<unsigned long long>(ts.tv_nsec) ^
                                     (static_cast<unsigned long long>(getpid()) << 44) ^ 1ull;
    std::vector<pid_t> kids;
    for (int r = 0; r < a.ranks; r++) {
        const pid_t pid = fork();
        if (pid < 0) {
            std::perror("fork");
            for (pid_t k : kids) kill(k, SIGKILL);
            for (pid_t k : kids) waitpid(k, nullptr, 0);
            return 1;
        }
        if (pid == 0) {
            std::vector<std::string> args(argv, argv + argc);
            args.push_back("--rank");
            args.push_back(std::to_string(r));
            args.push_back("--id-file");
            args.push_back(idFile);
            args.push_back("--nonce");
            args.push_back(std::to_string(nonce));
            std::vector<char*> cargs;
            for (std::string& s : args) cargs.push_back(s.data());
            cargs.push_back(nullptr);
            execv("/proc/self/exe", cargs.data());
            std::perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    size_t left = kids.size();
    while (left > 0) {
        int status = 0;
        const pid_t pid = waitpid(-1, &status, 0);
        if (pid < 0) { rc = 1; break; }
        size_t who = 0;
        while (who < kids.size() && kids[who] != pid) who++;
        if (who == kids.size()) continue; // not one of ours
        kids[who] = -1;
        left--;
        const bool good = WIFEXITED(status) && WEXITSTATUS(status) == 0;
        if (good || rc != 0) continue;
        rc = 1;
        if (WIFSIGNALED(status)) std::fprintf(stderr, "crt_render: rank %zu ended by signal %d; stopping the other ranks\n", who, WTERMSIG(status));
        else std::fprintf(stderr, "crt_render: rank %zu exited with code %d; stopping the other ranks\n", who, WIFEXITED(status) ? WEXITSTATUS(status) : -1);
        for (pid_t k : kids)
            if (k > 0) kill(k, SIGTERM);
        for (int waited = 0; waited < 200 && left > 0; waited++) { // up to 2 s to leave on their own
            for (size_t i = 0; i < kids.size(); i++) {
                if (kids[i] > 0 && waitpid(kids[i], nullptr, WNOHANG) == kids[i]) {
                    kids[i] = -1;
                    left--;
                }
            }
            if (left > 0) usleep(10000);
        }
        for (pid_t k : kids)
            if (k > 0) kill(k, SIGKILL);
    }
    std::remove(idFile.c_str());
    if (a.hostExchange) { // normally gone already (rank 0 removes it); a rank that was stopped may have left it behind
        char name[64];
        std::snprintf(name, sizeof(name), "/crt_render_%016llx", nonce);
        shm_unlink(name);
    }
    return rc;
}
} // namespace

int main(int argc, char** argv)
{
    if (argc < 2) { usage(); return 2; }
    Args a;
    a.scene = argv[1];
    for (int i = 2; i < argc; i++) {
        const std::string s = argv[i];
        auto next = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (s == "--mode") a.mode = static_cast<uint32_t>(std::atoi(next("--mode")));
        else if (s == "--size") { if (std::sscanf(next("--size"), "%ux%u", &a.w, &a.h) != 2) { usage(); return 2; } }
        else if (s == "--frames") a.frames = std::atoi(next("--frames"));
        else if (s == "--device") a.device = std::atoi(next("--device"));
        else if (s == "--orbit") a.orbit = static_cast<float>(std::atof(next("--orbit")));
        else if (s == "--pitch") a.pitch = static_cast<float>(std::atof(next("--pitch")));
        else if (s == "--forward") a.forward = static_cast<float>(std::atof(next("--forward")));
        else if (s == "--right") a.right = static_cast<float>(std::atof(next("--right")));
        else if (s == "--zoom") a.zoom = static_cast<float>(std::atof(next("--zoom")));
        else if (s == "--path") a.pathFile = next("--path");
        else if (s == "--mode-at") {
            int f = 0;
            unsigned m = 0;
            if (std::sscanf(next("--mode-at"), "%d:%u", &f, &m) != 2 || f < 0) { usage(); return 2; }
            a.modeAt[f] = m;
        }
        else if (s == "--spp") a.spp = std::atoi(next("--spp"));
        else if (s == "--bounces") a.bounces = std::atoi(next("--bounces"));
        else if (s == "--seed") a.seed = std::atoi(next("--seed"));
        else if (s == "--phong") { if (std::sscanf(next("--phong"), "%d:%d", &a.phongKs, &a.phongExp) != 2) { usage(); return 2; } }
        else if (s == "--out") a.out = next("--out");
        else if (s == "--count") a.count = true;
        else if (s == "--png") a.png = true;
        else if (s == "--host-exchange") a.hostExchange = true;
        else if (s == "--same-device") a.sameDevice = true;
        else if (s == "--ranks") a.ranks = std::atoi(next("--ranks"));
        else if (s == "--rank") a.rank = std::atoi(next("--rank"));
        else if (s == "--device-base") a.deviceBase = std::atoi(next("--device-base"));
        else if (s == "--id-file") a.idFile = next("--id-file");
        else if (s == "--nonce") a.nonce = std::strtoull(next("--nonce"), nullptr, 10);
        else { usage(); return 2; }
    }
    if (a.frames < 1 || a.ranks < 0 || a.ranks > 64) { usage(); return 2; }
    try {
        if (a.ranks > 0 && a.rank < 0) return launchRanks(a, argc, argv);
        if (a.ranks > 0 && (a.rank >= a.ranks || a.idFile.empty())) { usage(); return 2; }
        return runRank(a);
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "crt_render: %s\n", ex.what());
        return 1;
    }
}
