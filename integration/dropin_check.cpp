// A caller written against the reference's OWN headers only (src/bvh_interface.h,
// src/render.h, src/scene.h, src/screen.h, framework/trackball.h): it builds a Scene,
// constructs BvhInterface and calls the frame drivers, BvhInterface::intersect and
// getFinalColor exactly as src/main.cpp:50-63,159,226 do.  Linked against the two drop-in
// translation units of this directory (instead of the reference's src/bvh_interface.cpp,
// src/bounding_volume_hierarchy.cpp and src/render.cpp) it shows that the reference's
// callers compile and run unchanged on the GPU path.  Test tooling: the scene and the job
// come from files written by tests/test_dropin_binding.py, results go back as raw floats.
#include "bvh_interface.h"
#include "light.h"
#include "render.h"
#include "scene.h"
#include "screen.h"
#include "shading.h"
#include <framework/trackball.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

template <typename T>
std::vector<T> readArray(std::ifstream& in, size_t n)
{
    std::vector<T> v(n);
    in.read(reinterpret_cast<char*>(v.data()), std::streamsize(n * sizeof(T)));
    if (!in)
        throw std::runtime_error("scene file is short");
    return v;
}

Material material(const float* m)
{
    Material o;
    o.kd = { m[0], m[1], m[2] };
    o.ks = { m[3], m[4], m[5] };
    o.shininess = m[6];
    o.transparency = m[7];
    return o;
}

// scene file: int32 nMeshes, per mesh { int32 nv, nt; float v[nv][8]; uint32 t[nt][3]; float material[8] },
// int32 nLights, per light float[22] (kind, 3 points, 4 colours), int32 nSpheres, per sphere float[12]
Scene readScene(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in)
        throw std::runtime_error("cannot open " + path);
    Scene s;
    s.type = Custom;
    const int nMeshes = readArray<int32_t>(in, 1)[0];
    for (int i = 0; i < nMeshes; i++) {
        const auto n = readArray<int32_t>(in, 2);
        const auto v = readArray<float>(in, size_t(n[0]) * 8);
        const auto t = readArray<uint32_t>(in, size_t(n[1]) * 3);
        const auto m = readArray<float>(in, 8);
        Mesh mesh;
        for (int k = 0; k < n[0]; k++) {
            const float* p = &v[size_t(k) * 8];
            mesh.vertices.push_back(Vertex { { p[0], p[1], p[2] }, { p[3], p[4], p[5] }, { p[6], p[7] } });
        }
        for (int k = 0; k < n[1]; k++)
            mesh.triangles.push_back({ t[size_t(k) * 3], t[size_t(k) * 3 + 1], t[size_t(k) * 3 + 2] });
        mesh.material = material(m.data());
        s.meshes.push_back(std::move(mesh));
    }
    const int nLights = readArray<int32_t>(in, 1)[0];
    for (int i = 0; i < nLights; i++) {
        const auto l = readArray<float>(in, 22);
        auto p = [&](int k) { return glm::vec3(l[1 + 3 * k], l[2 + 3 * k], l[3 + 3 * k]); };
        auto c = [&](int k) { return glm::vec3(l[10 + 3 * k], l[11 + 3 * k], l[12 + 3 * k]); };
        if (int(l[0]) == 0)
            s.lights.push_back(PointLight { p(0), c(0) });
        else if (int(l[0]) == 1)
            s.lights.push_back(SegmentLight { p(0), p(1), c(0), c(1) });
        else
            s.lights.push_back(ParallelogramLight { p(0), p(1), p(2), c(0), c(1), c(2), c(3) });
    }
    const int nSpheres = readArray<int32_t>(in, 1)[0];
    for (int i = 0; i < nSpheres; i++) {
        const auto q = readArray<float>(in, 12);
        Sphere sp;
        sp.center = { q[0], q[1], q[2] };
        sp.radius = q[3];
        sp.material = material(&q[4]);
        s.spheres.push_back(sp);
    }
    return s;
}

struct Pose {
    float fovy = 0, distance = 0;
    glm::vec3 lookAt { 0 }, rotation { 0 };
};

Pose readPose(std::istringstream& ls)
{
    auto f = [&]() { std::string w; ls >> w; return std::strtof(w.c_str(), nullptr); }; // hex floats: exact bits
    Pose p;
    p.fovy = f(), p.distance = f();
    p.lookAt = { f(), f(), f() };
    p.rotation = { f(), f(), f() };
    return p;
}

} // namespace

int main(int argc, char** argv)
try {
    if (argc != 2) {
        std::fprintf(stderr, "usage: dropin_check job.txt\n");
        return 2;
    }
    std::ifstream job(argv[1]);
    std::string line, scenePath, outPath, raysPath, mode = "frame";
    std::vector<std::pair<int, std::string>> textures;
    std::string envPath;
    int envMapping = 1;
    float envFov = 180.f;
    glm::vec3 envBackground { 0.f };
    bool haveEnv = false;
    int W = 0, H = 0, driver = 0, samples = 0, depth = 0;
    float aperture = 0.1f, focal = 2.0f;
    Features features {};
    Pose cam, prev;
    bool havePrev = false;
    while (std::getline(job, line)) {
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "scene") ls >> scenePath;
        else if (key == "out") ls >> outPath;
        else if (key == "rays") ls >> raysPath;
        else if (key == "mode") ls >> mode;
        else if (key == "size") ls >> W >> H;
        else if (key == "depth") ls >> depth;
        else if (key == "texture") { int m; std::string p; ls >> m >> p; textures.push_back({ m, p }); }
        else if (key == "env") {
            std::string fov, r, g, b;
            ls >> envPath >> envMapping >> fov >> r >> g >> b;
            envFov = std::strtof(fov.c_str(), nullptr);
            envBackground = { std::strtof(r.c_str(), nullptr), std::strtof(g.c_str(), nullptr), std::strtof(b.c_str(), nullptr) };
            haveEnv = true;
        }
        else if (key == "driver") { std::string a, f; ls >> driver >> samples >> a >> f; aperture = std::strtof(a.c_str(), nullptr); focal = std::strtof(f.c_str(), nullptr); }
        else if (key == "globals") ls >> sampleSize >> numberOfRays; // the reference's own globals (src/light.cpp:14, src/shading.cpp:9)
        else if (key == "camera") cam = readPose(ls);
        else if (key == "prev") prev = readPose(ls), havePrev = true;
        else if (key == "features") {
            int b[12];
            for (int& x : b) ls >> x;
            features.enableShading = b[0], features.enableRecursive = b[1], features.enableHardShadow = b[2], features.enableSoftShadow = b[3];
            features.enableNormalInterp = b[4], features.enableTextureMapping = b[5], features.enableAccelStructure = b[6];
            features.extra.enableEnvironmentMapping = b[7], features.extra.enableBvhSahBinning = b[8];
            features.extra.enableBilinearTextureFiltering = b[9], features.extra.enableMipmapTextureFiltering = b[10];
            features.extra.enableGlossyReflection = b[11];
        }
    }
    Scene scene = readScene(scenePath);
    for (const auto& [m, p] : textures)
        scene.meshes[size_t(m)].material.kdTexture = std::make_shared<Image>(p);
    std::vector<EnvironmentMap> envStore;
    if (haveEnv) { // as src/main.cpp:155-156,544 set the scene's one environment map
        envStore.reserve(1);
        if (envPath == "-")
            envStore.emplace_back(envBackground);
        else
            envStore.push_back(EnvironmentMap::loadEnvironmentMap(envPath, EnvironmentMappingType(envMapping), envFov, envBackground));
        scene.environmentMap.push_back(&envStore[0]);
    }

    BvhInterface bvh { &scene, features }; // src/main.cpp:159,546
    std::ofstream out(outPath, std::ios::binary);
    const int32_t shape[2] = { bvh.numLevels(), bvh.numLeaves() };
    out.write(reinterpret_cast<const char*>(shape), sizeof(shape));

    if (mode == "frame") {
        // (the test harness's Trackball takes the aspect ratio where the reference passes its Window)
        float aspect = float(W) / float(H);
        Trackball camera(reinterpret_cast<Window*>(&aspect), cam.fovy, cam.lookAt, cam.distance, cam.rotation.x, cam.rotation.y);
        Screen screen(glm::ivec2(W, H), false);
        switch (driver) { // the router's choices, src/main.cpp:50-63
        case 0: renderRayTracing(scene, camera, bvh, screen, features); break;
        case 1: samples > 0 ? renderRayTracingMRaysPerPixel(scene, camera, bvh, screen, features, samples) : renderRayTracingMRaysPerPixel(scene, camera, bvh, screen, features); break;
        case 2: renderRayTracingDepthOfField(scene, camera, bvh, screen, features, aperture, focal, samples); break;
        case 3: renderRayTracingTransparency(scene, camera, bvh, screen, features); break;
        case 4:
            if (!havePrev)
                throw std::runtime_error("motion blur needs the previous pose");
            camera.setCamera(prev.lookAt, prev.rotation, prev.distance);
            renderRayTracingMotionBlur(scene, camera, bvh, screen, features, samples); // first call: still frame (src/main.cpp:568-571)
            camera.setCamera(cam.lookAt, cam.rotation, cam.distance);
            renderRayTracingMotionBlur(scene, camera, bvh, screen, features, samples);
            break;
        default: throw std::runtime_error("unknown driver");
        }
        out.write(reinterpret_cast<const char*>(screen.pixels().data()), std::streamsize(screen.pixels().size() * sizeof(glm::vec3)));
        // The reference's command line renders every camera on a thread of its own
        // (src/main.cpp:560-563): the same frame from three such threads at once must be the
        // frame above, bit for bit.  (Not the motion-blur driver, which keeps state between calls.)
        if (driver != 4) {
            std::vector<Screen> again(3, Screen(glm::ivec2(W, H), false));
            std::vector<std::thread> workers;
            std::string failure;
            std::mutex guard;
            for (Screen& s : again)
                workers.emplace_back([&, target = &s] {
                    try {
                        Trackball cameraOfThread(reinterpret_cast<Window*>(&aspect), cam.fovy, cam.lookAt, cam.distance, cam.rotation.x, cam.rotation.y);
                        switch (driver) {
                        case 0: renderRayTracing(scene, cameraOfThread, bvh, *target, features); break;
                        case 1: samples > 0 ? renderRayTracingMRaysPerPixel(scene, cameraOfThread, bvh, *target, features, samples) : renderRayTracingMRaysPerPixel(scene, cameraOfThread, bvh, *target, features); break;
                        case 2: renderRayTracingDepthOfField(scene, cameraOfThread, bvh, *target, features, aperture, focal, samples); break;
                        default: renderRayTracingTransparency(scene, cameraOfThread, bvh, *target, features); break;
                        }
                    } catch (const std::exception& e) {
                        std::lock_guard<std::mutex> lock(guard);
                        failure = e.what();
                    }
                });
            for (std::thread& w : workers)
                w.join();
            if (!failure.empty())
                throw std::runtime_error("on a thread of its own: " + failure);
            for (const Screen& s : again)
                if (std::memcmp(s.pixels().data(), screen.pixels().data(), screen.pixels().size() * sizeof(glm::vec3)) != 0)
                    throw std::runtime_error("the frame rendered on a thread of its own differs from the first");
        }
    } else {
        std::ifstream rin(raysPath, std::ios::binary);
        rin.seekg(0, std::ios::end);
        const size_t n = size_t(rin.tellg()) / 28;
        rin.seekg(0);
        const auto r = readArray<float>(rin, n * 7);
        for (size_t i = 0; i < n; i++) {
            Ray ray { { r[i * 7], r[i * 7 + 1], r[i * 7 + 2] }, { r[i * 7 + 3], r[i * 7 + 4], r[i * 7 + 5] }, r[i * 7 + 6] };
            if (mode == "trace") { // src/main.cpp:224-227
                HitInfo hit {};
                const bool ok = bvh.intersect(ray, hit, features);
                // hit flag, t, normal, barycentrics, uv, the triangle's vertex indices, mesh index, material kd
                float rec[16] = { ok ? 1.f : 0.f, ray.t, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1.f, 0, 0 };
                if (ok) {
                    rec[2] = hit.normal.x, rec[3] = hit.normal.y, rec[4] = hit.normal.z;
                    if (hit.mesh) {
                        rec[5] = hit.barycentricCoord.x, rec[6] = hit.barycentricCoord.y, rec[7] = hit.barycentricCoord.z;
                        rec[8] = hit.texCoord.x, rec[9] = hit.texCoord.y;
                        rec[10] = float(hit.triangle.x), rec[11] = float(hit.triangle.y), rec[12] = float(hit.triangle.z);
                        rec[13] = float(hit.mesh - scene.meshes.data());
                    }
                    rec[14] = hit.material.kd.x, rec[15] = hit.material.shininess;
                }
                out.write(reinterpret_cast<const char*>(rec), sizeof(rec));
            } else {
                const glm::vec3 c = getFinalColor(scene, bvh, ray, features, depth);
                out.write(reinterpret_cast<const char*>(&c), sizeof(c));
            }
        }
    }
    return 0;
} catch (const std::exception& e) {
    std::fprintf(stderr, "dropin_check: %s\n", e.what());
    return 1;
}
