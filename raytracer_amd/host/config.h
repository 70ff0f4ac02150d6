// Configuration file of the reference's command-line renderer (src/config.h:16-45,
// src/config.cpp:201-382): a TOML file naming the scene, the window size, the feature
// switches, any number of [[cameras]] and [[lights]].  Same names, same defaults, same
// quirks: enable_soft_shadow, enable_transparency and enable_bvh_sah_binning are not
// read by the reference (its command line can render neither area lights nor the SAH
// tree), so they are not read here either unless the caller asks for the extended keys.
// The reference parses with toml++; this file brings its own reader for the subset of
// TOML such a configuration uses (tables, arrays of tables, booleans, integers, floats,
// strings, nested arrays, comments).  Where the reference dereferences a missing node
// (a missing [features] key, an integer where it asks for a float) this reader takes the
// default or the integer's value.
#pragma once
#include <string>
#include <variant>
#include <vector>

#include "rt_types.h"

namespace rt {

// src/scene.h:16-29
enum SceneType { SingleTriangle, Cube, CubeTextured, CornellBox, CornellBoxParallelogramLight, Monkey, Teapot, Dragon, Spheres, Custom, TransparencyDebug, TextureDebug };

struct CameraConfig { // src/config.h:16-21
    float fieldOfView = 50.0f;      // degrees
    float distanceFromLookAt = 3.0f;
    vec3 lookAt { 0.0f, 0.0f, 0.0f };
    vec3 rotation { 20.0f, 20.0f, 0.0f }; // degrees
};

struct Config { // src/config.h:23-33
    Features features = {};
    bool cliRenderingEnabled = false;
    ivec2 windowSize { 800, 800 };
    std::string dataPath;
    std::variant<SceneType, std::string> scene = SingleTriangle; // a prebuilt scene, or the path of an OBJ file
    std::string outputDir;
    std::vector<CameraConfig> cameras;
    std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>> lights;
};

// src/config.cpp:384-448
std::string serialize(SceneType sceneType);
bool deserialize(const std::string& name, SceneType& out);

// src/config.cpp:201-382.  defaultDataPath stands in for the reference's DATA_DIR.  With
// extendedFeatures the three switches the reference forgets are read too
// ([features] enable_soft_shadow, [features.extra] enable_transparency, enable_bvh_sah_binning).
// Throws RtError-compatible std::runtime_error on a file that cannot be read or parsed, or a
// scene file that does not exist (the reference prints and exits).
Config readConfigFile(const std::string& configPath, const std::string& defaultDataPath, bool extendedFeatures = false);

// src/scene.cpp:5-103 and :105-114
Scene loadScenePrebuilt(SceneType type, const std::string& dataDir);
Scene loadSceneFromFile(const std::string& path, const std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>>& lights);

} // namespace rt
