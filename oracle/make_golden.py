#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerate tests/golden/ from the reference's own scene layer.

Runs only where /root/reference exists (this container).  It executes oracle/_ref/ref_dump --
a tool linked against the reference's CRT* sources compiled in place (oracle/Makefile) -- and writes

  tests/golden/dragon_scene_layer.json   known answers of the reference scene layer for its one shipped
                                         scene: sizeof()s, parse results, CRTMesh::calculateVertexNormals
                                         output (R/CRTMesh.cpp:66-94), CRTCamera operation sequences
                                         (R/CRTCamera.cpp:9-130), CRTVector*CRTMatrix (R/CRTMatrix.cpp:26-38)
  tests/golden/texture_known_answers.json  getColor(u,v) of the reference's four texture classes (R/CRTTexture*.cpp) on a grid,
                                         the bitmap one reading tests/golden/tex7x5.ppm (written here, seeded)
  tests/golden/dragon.crtscene           the same scene DATA re-serialised from the reference's parsed
                                         values (float32 printed with 9 significant digits round-trips
                                         exactly) so the GPU box, which has no /root/reference, can load it.
"""
import json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SCENE = "/root/reference/DirectX-RayTracer/DirectX-RayTracer/Scenes/Dragon.crtscene"
MATERIAL_NAMES = {1: "diffuse", 2: "reflective", 3: "refractive", 4: "constant"}


def fmt(x):
    return repr(float(x)) if float(x) != int(x) else str(int(x))


def main():
    if not os.path.exists(REF_SCENE):
        print("reference tree absent; nothing to do")
        return 0
    subprocess.check_call(["make", "-C", HERE, "_ref/ref_dump"])
    tmp = os.path.join(HERE, "_ref", "dragon_ref.json")
    subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), REF_SCENE, tmp])
    d = json.load(open(tmp))
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)

    # 1) re-serialised scene data
    scene = {
        "settings": {"background_color": d["settings"]["background_color"],
                     "image_settings": {"width": d["settings"]["width"], "height": d["settings"]["height"]}},
        "camera": {"matrix": d["camera"]["matrix"], "position": d["camera"]["position"]},
        "lights": [{"intensity": l["intensity"], "position": l["position"]} for l in d["lights"]],
        "materials": [{"type": MATERIAL_NAMES[m["type"]], "albedo": m["albedo"],
                       "smooth_shading": bool(m["smooth_shading"])} for m in d["materials"]],
        "objects": [{"material_index": m["material_index"],
                     "vertices": [c for v in m["vertices"] for c in v],
                     "triangles": m["indices"]} for m in d["meshes"]],
    }
    with open(os.path.join(gold, "dragon.crtscene"), "w") as f:
        json.dump(scene, f, separators=(",", ":"))

    # 2) known answers (drop the bulky raw geometry: it lives in the scene file above)
    for m in d["meshes"]:
        del m["vertices"], m["indices"]
    with open(os.path.join(gold, "dragon_scene_layer.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    # 3) texture classes: a small binary PPM the reference's vendored stb_image can decode, and getColor known answers
    import numpy as np
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    with open(os.path.join(gold, "tex7x5.ppm"), "wb") as f:
        f.write(b"P6\n7 5\n255\n" + img.tobytes())
    ttmp = os.path.join(HERE, "_ref", "textures_ref.json")
    subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), "--textures", os.path.join(gold, "tex7x5.ppm"), ttmp])
    with open(os.path.join(gold, "texture_known_answers.json"), "w") as f:
        json.dump(json.load(open(ttmp)), f, separators=(",", ":"))
    print("wrote", gold)
    return 0


if __name__ == "__main__":
    sys.exit(main())
