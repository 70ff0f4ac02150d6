// Headless end-to-end use of the host layer, written the way the reference's
// command-line path is (src/main.cpp:520-589): load a mesh, build the BVH
// interface, render with one of the frame drivers, write a bitmap.
//
//   render_obj <file.obj> <out.bmp> [width height [samples]]
//
// One white point light at (-1, 1, -1) as in the reference's Dragon scene
// (src/scene.cpp:68-73); the default camera (src/config.h:16-21).  With
// samples > 1 the jittered multi-sample driver is used.
#include <cstdio>
#include <cstdlib>

#include "raytracer.h"
#include "scene_io.h"

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <file.obj> <out.bmp> [width height [samples]]\n", argv[0]);
        return 2;
    }
    const int width = argc > 4 ? std::atoi(argv[3]) : 512, height = argc > 4 ? std::atoi(argv[4]) : 512;
    const int samples = argc > 5 ? std::atoi(argv[5]) : 1;
    try {
        rt::Scene scene;
        scene.meshes = rt::loadMesh(argv[1], /*centerAndNormalize=*/true);
        scene.lights.emplace_back(rt::PointLight { rt::vec3(-1.f, 1.f, -1.f), rt::vec3(1.f, 1.f, 1.f) });

        rt::Features features;
        features.enableShading = true;
        features.enableHardShadow = true;
        features.enableNormalInterp = true;
        features.enableAccelStructure = true;
        features.extra.enableBvhSahBinning = true;

        const rt::BvhInterface bvh(&scene, features);
        std::printf("%zu meshes, BVH with %d levels and %d leaves\n", scene.meshes.size(), bvh.numLevels(), bvh.numLeaves());

        const rt::Trackball camera(float(width) / float(height), rt::radians(50.0f), rt::vec3(0.f, 0.f, 0.f), 3.0f, rt::radians(20.0f), rt::radians(20.0f));
        rt::Screen screen(rt::ivec2 { width, height });
        if (samples > 1)
            rt::renderRayTracingMRaysPerPixel(scene, camera, bvh, screen, features, samples);
        else
            rt::renderRayTracing(scene, camera, bvh, screen, features);
        screen.writeBitmapToFile(argv[2]);
        std::printf("wrote %s (%dx%d, %d sample%s per pixel)\n", argv[2], width, height, samples, samples == 1 ? "" : "s");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
