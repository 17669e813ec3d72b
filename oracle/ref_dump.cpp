// TEST INFRASTRUCTURE (oracle side) -- not product code.
//
// Driver that links against the *reference's own* platform-neutral scene layer
// (CRTVector/CRTMatrix/CRTTriangle/CRTMesh/CRTCamera/CRTLight/CRTMaterial/CRTScene/
// CRTSceneParser + the rapidjson the reference vendors), compiled in place from
// /root/reference by oracle/Makefile into oracle/_ref/ref_dump (git-ignored).
// It prints known answers of that layer as JSON; oracle/make_golden.py stores them
// under tests/golden/ so the build's own scene layer can be pinned against the
// reference on machines where /root/reference does not exist (the GPU box).
//
// Nothing of the reference is copied: this file only *calls* its public API
// (R/CRTScene.h:21-36, R/CRTCamera.h:8-24, R/CRTMesh.h:10-23, R/CRTMatrix.h:12-18).
#include "CRTScene.h"
#include "CRTTriangle.h"
#include "CRTTextureAlbedo.h"
#include "CRTTextureBitmap.h"
#include "CRTTextureChecker.h"
#include "CRTTextureEdges.h"
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

static void putv(FILE* f, const CRTVector& v)
{
    fprintf(f, "[%.9g,%.9g,%.9g]", v.getX(), v.getY(), v.getZ());
}

static void putm(FILE* f, const CRTMatrix& m)
{
    fprintf(f, "[");
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            fprintf(f, "%s%.9g", (i + j) ? "," : "", m.get(i, j));
    fprintf(f, "]");
}

static void putcam(FILE* f, const char* name, const CRTCamera& c, bool last = false)
{
    fprintf(f, "  {\"op\":\"%s\",\"position\":", name);
    putv(f, c.getPosition());
    fprintf(f, ",\"matrix\":");
    putm(f, c.getRotationMatrix());
    fprintf(f, "}%s\n", last ? "" : ",");
}

int main(int argc, char** argv)
{
    if (argc >= 4 && std::string(argv[1]) == "--bitmap") {
        // known answers of the reference's CRTTextureBitmap (R/CRTTextureBitmap.cpp: stbi_load + getColor) for one image file:
        // argv[2] = the file, argv[3] = output json: [[u, v, r, g, b], ...] on a 21 x 21 grid of (u, v)
        FILE* f = fopen(argv[3], "w");
        if (!f) return 3;
        CRTTextureBitmap bmp(argv[2], "b");
        fprintf(f, "[");
        bool first = true;
        for (int iu = 0; iu <= 20; iu++)
            for (int iv = 0; iv <= 20; iv++) {
                const float u = iu * 0.05f + (iu % 3) * 0.003f, v = iv * 0.05f + (iv % 4) * 0.002f;
                const CRTVector c = bmp.getColor(u, v);
                fprintf(f, "%s[%.9g,%.9g,%.9g,%.9g,%.9g]", first ? "" : ",", u, v, c.getX(), c.getY(), c.getZ());
                first = false;
            }
        fprintf(f, "]\n");
        fclose(f);
        return 0;
    }
    if (argc >= 4 && std::string(argv[1]) == "--textures") {
        // known answers of the reference's texture classes (R/CRTTexture*.cpp getColor): argv[2] = a bitmap the vendored
        // stb_image can read, argv[3] = output json
        FILE* f = fopen(argv[3], "w");
        if (!f) return 3;
        CRTTextureAlbedo alb(CRTVector(0.25f, 0.5f, 0.75f), "a");
        CRTTextureEdges edg(CRTVector(1.f, 0.f, 0.f), CRTVector(0.f, 0.f, 1.f), 0.1f, "e");
        CRTTextureChecker chk(CRTVector(0.9f, 0.8f, 0.7f), CRTVector(0.1f, 0.2f, 0.3f), 0.125f, "c");
        CRTTextureBitmap bmp(argv[2], "b");
        const CRTTexture* tex[4] = { &alb, &edg, &chk, &bmp };
        const char* names[4] = { "albedo", "edges", "checker", "bitmap" };
        fprintf(f, "{\n \"params\":{\"albedo\":[0.25,0.5,0.75],\"edges\":{\"edge_color\":[1,0,0],\"inner_color\":[0,0,1],\"edge_width\":0.1},"
                   "\"checker\":{\"color_A\":[0.9,0.8,0.7],\"color_B\":[0.1,0.2,0.3],\"square_size\":0.125}},\n");
        for (int t = 0; t < 4; t++) {
            fprintf(f, " \"%s\":[", names[t]);
            bool first = true;
            for (int iu = 0; iu <= 20; iu++)
                for (int iv = 0; iv <= 20; iv++) {
                    const float u = iu * 0.05f + (iu % 3) * 0.003f, v = iv * 0.05f + (iv % 4) * 0.002f;
                    if (t == 1 && u + v > 1.0f) continue; // edges: barycentric domain
                    const CRTVector c = tex[t]->getColor(u, v);
                    fprintf(f, "%s[%.9g,%.9g,%.9g,%.9g,%.9g]", first ? "" : ",", u, v, c.getX(), c.getY(), c.getZ());
                    first = false;
                }
            fprintf(f, "]%s\n", t < 3 ? "," : "");
        }
        fprintf(f, "}\n");
        fclose(f);
        return 0;
    }
    if (argc < 3) { fprintf(stderr, "usage: ref_dump <scene.crtscene> <out.json>\n"); return 2; }
    // the reference parser chats on std::cout; keep it out of our way
    std::ostringstream sink;
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    CRTScene scene(argv[1]);
    std::cout.rdbuf(old);

    FILE* f = fopen(argv[2], "w");
    if (!f) return 3;
    fprintf(f, "{\n");
    fprintf(f, " \"sizeof\":{\"CRTVector\":%zu,\"CRTMatrix\":%zu,\"CRTTriangle\":%zu},\n",
            sizeof(CRTVector), sizeof(CRTMatrix), sizeof(CRTTriangle));
    const CRTSettings& st = scene.getSettings();
    fprintf(f, " \"settings\":{\"width\":%d,\"height\":%d,\"background_color\":", st.imageWidth, st.imageHeight);
    putv(f, st.backgroundColor);
    fprintf(f, "},\n \"camera\":{\"position\":");
    putv(f, scene.getCamera().getPosition());
    fprintf(f, ",\"matrix\":");
    putm(f, scene.getCamera().getRotationMatrix());
    fprintf(f, "},\n \"lights\":[");
    for (size_t i = 0; i < scene.getLights().size(); i++) {
        const CRTLight& l = scene.getLights()[i];
        fprintf(f, "%s{\"intensity\":%.9g,\"position\":", i ? "," : "", l.getIntensity());
        putv(f, l.getPosition());
        fprintf(f, "}");
    }
    fprintf(f, "],\n \"materials\":[");
    for (size_t i = 0; i < scene.getMaterials().size(); i++) {
        const CRTMaterial& m = scene.getMaterials()[i];
        fprintf(f, "%s{\"type\":%d,\"smooth_shading\":%d,\"is_texture\":%d,\"albedo\":", i ? "," : "",
                (int)m.getType(), (int)m.isSmoothShading(), (int)m.isTexture());
        putv(f, m.getAlbedo());
        fprintf(f, "}");
    }
    fprintf(f, "],\n \"n_textures\":%zu,\n \"meshes\":[\n", scene.getTextures().size());
    for (size_t mi = 0; mi < scene.getObjects().size(); mi++) {
        const CRTMesh& m = scene.getObjects()[mi];
        fprintf(f, "  {\"n_vertices\":%zu,\"n_indices\":%zu,\"n_uvs\":%zu,\"material_index\":%d,\n",
                m.getVertices().size(), m.getIndices().size(), m.getUV().size(), m.getMaterialIndex());
        double sx = 0, sy = 0, sz = 0; long long si = 0;
        for (const CRTVector& v : m.getVertices()) { sx += v.getX(); sy += v.getY(); sz += v.getZ(); }
        for (int i : m.getIndices()) si += i;
        fprintf(f, "   \"vertex_sum\":[%.17g,%.17g,%.17g],\"index_sum\":%lld,\n", sx, sy, sz, si);
        fprintf(f, "   \"vertices\":[");
        for (size_t i = 0; i < m.getVertices().size(); i++) {
            if (i) fprintf(f, ",");
            putv(f, m.getVertices()[i]);
        }
        fprintf(f, "],\n   \"indices\":[");
        for (size_t i = 0; i < m.getIndices().size(); i++) fprintf(f, "%s%d", i ? "," : "", m.getIndices()[i]);
        fprintf(f, "],\n");
        fprintf(f, "   \"vertex_normals\":[");
        for (size_t i = 0; i < m.getVertexNormals().size(); i++) {
            if (i) fprintf(f, ",");
            putv(f, m.getVertexNormals()[i]);
        }
        fprintf(f, "]}%s\n", mi + 1 < scene.getObjects().size() ? "," : "");
    }
    fprintf(f, " ],\n");

    // face normals of the first 8 triangles of the last mesh via CRTTriangle (R/CRTTriangle.cpp:22-30)
    {
        const CRTMesh& m = scene.getObjects().back();
        fprintf(f, " \"face_normals_last_mesh_first8\":[");
        for (int t = 0; t < 8 && (size_t)(3 * t + 2) < m.getIndices().size(); t++) {
            CRTTriangle tri(m.getVertices()[m.getIndices()[3 * t]], m.getVertices()[m.getIndices()[3 * t + 1]],
                            m.getVertices()[m.getIndices()[3 * t + 2]]);
            if (t) fprintf(f, ",");
            putv(f, tri.getNormal());
        }
        fprintf(f, "],\n");
    }

    // camera operation sequences (R/CRTCamera.cpp:9-130); each entry is the state AFTER the op
    fprintf(f, " \"camera_sequence\":[\n");
    CRTCamera c = scene.getCamera();
    putcam(f, "init", c);
    c.rotate(10.f, 5.f);            putcam(f, "rotate(10,5)", c);
    c.zoom(2.5f);                   putcam(f, "zoom(2.5)", c);
    c.moveForward(-1.25f);          putcam(f, "moveForward(-1.25)", c);
    c.moveRight(3.f);               putcam(f, "moveRight(3)", c);
    c.rotate(-35.f, -120.f);        putcam(f, "rotate(-35,-120)", c);   // pitch clamps at -89 deg
    c.rotate(200.f, 300.f);         putcam(f, "rotate(200,300)", c);    // pitch clamps at +89 deg
    c.moveForward(0.5f);            putcam(f, "moveForward(0.5)", c);
    c.pan(30.f);                    putcam(f, "pan(30)", c);
    c.tilt(-12.f);                  putcam(f, "tilt(-12)", c);
    c.roll(7.f);                    putcam(f, "roll(7)", c);
    c.panAroundTarget(45.f, CRTVector(0.f, 0.f, 0.f)); putcam(f, "panAroundTarget(45,origin)", c, true);
    fprintf(f, " ],\n");

    // CRTVector * CRTMatrix is the row-vector product (R/CRTMatrix.cpp:26-38)
    {
        CRTMatrix m(1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f, 10.f);
        CRTVector v(0.5f, -1.5f, 2.f);
        fprintf(f, " \"row_vector_times_matrix\":{\"v\":");
        putv(f, v);
        fprintf(f, ",\"m\":");
        putm(f, m);
        fprintf(f, ",\"out\":");
        putv(f, v * m);
        fprintf(f, ",\"mm\":");
        putm(f, m * m);
        fprintf(f, "}\n}\n");
    }
    fclose(f);
    return 0;
}
