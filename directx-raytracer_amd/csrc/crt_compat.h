// Spelling shim: code written against the reference's scene layer (CRTVector, CRTScene, ... R/CRT*.h) compiles
// against crt::* unchanged.  CRTTexture* are not mapped: the renderer never consumed them.
#pragma once
#include "scene.h"

using CRTVector = crt::Vector;
using CRTMatrix = crt::Matrix;
using CRTTriangle = crt::Triangle;
using CRTMesh = crt::Mesh;
using CRTCamera = crt::Camera;
using CRTLight = crt::Light;
using CRTMaterial = crt::Material;
using CRTMaterialType = crt::MaterialType;
using CRTSettings = crt::Settings;
using CRTScene = crt::Scene;
using CRTSceneParser = crt::SceneParser;
