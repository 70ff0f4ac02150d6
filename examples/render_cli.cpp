// The reference's command-line renderer (src/main.cpp:520-589) on the GPU path, headless:
// a TOML configuration names the scene, the window size, the feature switches, the cameras
// and the lights (src/config.cpp:201-382); every camera is rendered on a thread of its own
// with the driver the reference's router picks (src/main.cpp:50-63: depth of field, else
// motion blur, else transparency, else multiple rays per pixel, else the plain driver) and
// written to <output_dir>/<scene>_<time>_cam_<i>.bmp.  No window is opened (the reference
// creates a hidden one because its Trackball needs a Window for the aspect ratio).
//
//   render_cli <config.toml> [--data-dir DIR] [--extended-features] [--device-build]
//
// --data-dir           where "default" data_path points (the reference compiles its data/ in)
// --extended-features  also read enable_soft_shadow, enable_transparency and
//                      enable_bvh_sah_binning, which the reference's reader forgets
// --device-build       build the tree on the GPU (rtx_scene_build_device)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "config.h"
#include "raytracer.h"
#include "scene_io.h"

namespace {

void router(const rt::Scene& scene, const rt::Trackball& camera, const rt::BvhInterface& bvh, rt::Screen& screen, const rt::Features& features)
{
    if (features.extra.enableDepthOfField)
        rt::renderRayTracingDepthOfField(scene, camera, bvh, screen, features);
    else if (features.extra.enableMotionBlur)
        rt::renderRayTracingMotionBlur(scene, camera, bvh, screen, features);
    else if (features.extra.enableTransparency)
        rt::renderRayTracingTransparency(scene, camera, bvh, screen, features);
    else if (features.extra.enableMultipleRaysPerPixel)
        rt::renderRayTracingMRaysPerPixel(scene, camera, bvh, screen, features);
    else
        rt::renderRayTracing(scene, camera, bvh, screen, features);
}

void makeDirectories(const std::string& path)
{
    for (size_t at = 1; at <= path.size(); at++)
        if (at == path.size() || path[at] == '/')
            ::mkdir(path.substr(0, at).c_str(), 0777); // (existing directories: EEXIST, ignored)
}

} // namespace

int main(int argc, char** argv)
{
    std::string configPath, dataDir = "data/";
    bool extended = false;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--data-dir") && i + 1 < argc)
            dataDir = argv[++i];
        else if (!std::strcmp(argv[i], "--extended-features"))
            extended = true;
        else if (!std::strcmp(argv[i], "--device-build"))
            rt::buildTreeOnDevice = true;
        else if (configPath.empty())
            configPath = argv[i];
        else
            configPath.clear(), i = argc;
    }
    if (configPath.empty()) {
        std::fprintf(stderr, "usage: %s <config.toml> [--data-dir DIR] [--extended-features] [--device-build]\n", argv[0]);
        return 2;
    }
    try {
        rt::Config config = rt::readConfigFile(configPath, dataDir, extended);
        if (config.windowSize.x <= 0 || config.windowSize.y <= 0)
            throw rt::RtError("window_size is missing or empty (the reference's reader then makes a 0 x 0 window)");
        rt::Scene scene;
        std::string sceneName;
        if (std::holds_alternative<std::string>(config.scene)) {
            const std::string& path = std::get<std::string>(config.scene);
            scene = rt::loadSceneFromFile(path, config.lights);
            const size_t slash = path.find_last_of('/'), from = slash == std::string::npos ? 0 : slash + 1, dot = path.find_last_of('.');
            sceneName = path.substr(from, (dot != std::string::npos && dot > from) ? dot - from : std::string::npos);
        } else {
            scene = rt::loadScenePrebuilt(std::get<rt::SceneType>(config.scene), config.dataPath);
            sceneName = rt::serialize(std::get<rt::SceneType>(config.scene));
        }
        // the command line's one environment map (src/main.cpp:544): default.png, cylindrical, 120 degrees, grey
        // behind it; without the file a constant grey environment
        rt::EnvironmentMap cliMap;
        cliMap.backgroundColor = rt::vec3(0.5f, 0.5f, 0.5f);
        cliMap.mappingType = rt::CYLINDRICAL;
        cliMap.verticalFieldOfViewDegrees = 120.f;
        {
            const std::string file = config.dataPath + (config.dataPath.empty() || config.dataPath.back() == '/' ? "" : "/") + "default.png";
            if (std::ifstream(file))
                cliMap.texture = std::make_shared<rt::Image>(rt::loadImage(file));
        }
        scene.environmentMap.push_back(&cliMap);
        const rt::BvhInterface bvh(&scene, config.features);
        std::printf("scene %s: %zu meshes, %zu lights, BVH with %d levels and %d leaves\n", sceneName.c_str(), scene.meshes.size(), scene.lights.size(), bvh.numLevels(), bvh.numLeaves());
        makeDirectories(config.outputDir);
        char stamp[64];
        const std::time_t now = std::time(nullptr);
        std::strftime(stamp, sizeof(stamp), "%Y-%m-%d-%H:%M:%S", std::localtime(&now));
        const auto start = std::chrono::steady_clock::now();
        std::vector<std::thread> workers;
        std::mutex report;
        std::string failure;
        for (size_t index = 0; index < config.cameras.size(); index++)
            workers.emplace_back([&, index] {
                try {
                    // a frame slot per thread: the cameras' frames share the GPU instead of taking turns
                    rt::selectFrameSlot(int(index % RTX_FRAME_SLOTS));
                    const rt::CameraConfig& cc = config.cameras[index];
                    rt::Screen screen(config.windowSize, false);
                    screen.clear(rt::vec3(0.0f, 0.0f, 0.0f));
                    rt::Trackball camera(float(config.windowSize.x) / float(config.windowSize.y), rt::radians(cc.fieldOfView), cc.distanceFromLookAt);
                    camera.setCamera(cc.lookAt, rt::vec3(rt::radians(cc.rotation.x), rt::radians(cc.rotation.y), rt::radians(cc.rotation.z)), cc.distanceFromLookAt);
                    if (config.features.extra.enableMotionBlur) {
                        // the still frame, then the frame blurred towards a look-at point 0.3 further along z (src/main.cpp:568-571)
                        rt::renderRayTracingMotionBlur(scene, camera, bvh, screen, config.features);
                        rt::Trackball moved = camera;
                        moved.setCamera(camera.lookAt() + rt::vec3(0.f, 0.f, 0.3f), camera.rotationEulerAngles(), camera.distanceFromLookAt());
                        rt::renderRayTracingMotionBlur(scene, moved, bvh, screen, config.features);
                    } else {
                        router(scene, camera, bvh, screen, config.features);
                    }
                    const std::string file = config.outputDir + "/" + sceneName + "_" + stamp + "_cam_" + std::to_string(index) + ".bmp";
                    screen.writeBitmapToFile(file);
                    std::lock_guard<std::mutex> lock(report);
                    std::printf("Image %zu saved to %s\n", index, file.c_str());
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lock(report);
                    failure = e.what();
                }
            });
        for (std::thread& w : workers)
            w.join();
        if (!failure.empty())
            throw rt::RtError(failure);
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count();
        std::printf("Rendering took %lld ms, %zu images rendered.\n", ms, config.cameras.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
