// TEST INFRASTRUCTURE (oracle/_ref build only): a C ABI over the reference's
// own, unmodified hot-path translation units so that Python can drive them to
// produce golden vectors and to time the reference CPU path.  Nothing here is
// reference code; it only calls the reference's public functions.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <limits>
#include <memory>
#include <new>
#include <random>
#include <string>
#include <variant>
#include <vector>
#include <omp.h>

// The tree dump below reads two private members (BvhInterface::m_impl and
// BoundingVolumeHierarchy::nodes/root).  Access control does not change layout.
#define private public
#include "bounding_volume_hierarchy.h"
#include "bvh_interface.h"
#undef private

#include "config.h"
#include "dof.h"
// the radiance bins are a private member; the harness reads them for a known-answer test
#define private public
#include "environment_mapping.h"
#undef private
#include "sampling.h"
#include "interpolate.h"
#include "intersect.h"
#include "light.h"
#include "render.h"
#include "scene.h"
#include "screen.h"
#include "shading.h"
#include "texture.h"
#include "transparency.h"
#include <framework/trackball.h>

#include "../../include/rtx.h"

thread_local uint32_t rtx_shim_seed = 0;
thread_local uint32_t rtx_shim_pixel = 0;
thread_local uint32_t rtx_shim_sample = 0;
thread_local uint32_t rtx_shim_counter = 0;

namespace {

struct RefScene {
    Scene scene;
    std::vector<std::unique_ptr<EnvironmentMap>> envs;
};

glm::vec3 v3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }
void put3(float* d, const glm::vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// Image only has a from-file constructor; build one from raw pixels in place.
Image* makeImage(const float* rgb, int w, int h)
{
    void* mem = ::operator new(sizeof(Image));
    Image* im = static_cast<Image*>(mem);
    im->width = w;
    im->height = h;
    new (&im->pixels) std::vector<glm::vec3>(size_t(w) * size_t(h));
    std::memcpy(im->pixels.data(), rgb, size_t(w) * size_t(h) * 3 * sizeof(float));
    return im;
}
void freeImage(Image* im)
{
    im->pixels.~vector();
    ::operator delete(im);
}

Features toFeatures(const rtx_features* f)
{
    Features o;
    o.enableShading = f->enable_shading;
    o.enableRecursive = f->enable_recursive;
    o.enableHardShadow = f->enable_hard_shadow;
    o.enableSoftShadow = f->enable_soft_shadow;
    o.enableNormalInterp = f->enable_normal_interp;
    o.enableTextureMapping = f->enable_texture_mapping;
    o.enableAccelStructure = f->enable_accel_structure;
    o.extra.enableEnvironmentMapping = f->enable_environment_mapping;
    o.extra.enableBvhSahBinning = f->enable_bvh_sah_binning;
    o.extra.enableBilinearTextureFiltering = f->enable_bilinear_texture_filtering;
    o.extra.enableMipmapTextureFiltering = f->enable_mipmap_texture_filtering;
    o.extra.enableGlossyReflection = f->enable_glossy_reflection;
    return o;
}

} // namespace

extern "C" {

// readConfigFile (src/config.cpp:201-382) flattened: out[0] command_line_rendering, [1..2] window size,
// [3] the scene is a file (1) or a prebuilt scene (0), [4] the prebuilt scene's number, [5..10] the six
// basic feature switches, [11..18] bloom, multiple rays, motion blur, depth of field, glossy, environment
// map, bilinear, mipmap, [19..21] soft shadows, transparency, SAH binning, [22] number of cameras, then 8
// numbers per camera, then the number of lights and 22 numbers per light; sceneName as src/main.cpp:534-543.
int ref_read_config(const char* path, double* out, int capacity, char* sceneName, int nameCapacity)
{
    const Config c = readConfigFile(path);
    std::vector<double> v;
    v.push_back(c.cliRenderingEnabled);
    v.push_back(c.windowSize.x);
    v.push_back(c.windowSize.y);
    const bool file = std::holds_alternative<std::filesystem::path>(c.scene);
    v.push_back(file ? 1 : 0);
    v.push_back(file ? -1 : int(std::get<SceneType>(c.scene)));
    const Features& f = c.features;
    for (bool b : { f.enableShading, f.enableRecursive, f.enableHardShadow, f.enableNormalInterp, f.enableTextureMapping, f.enableAccelStructure,
             f.extra.enableBloomEffect, f.extra.enableMultipleRaysPerPixel, f.extra.enableMotionBlur, f.extra.enableDepthOfField, f.extra.enableGlossyReflection,
             f.extra.enableEnvironmentMapping, f.extra.enableBilinearTextureFiltering, f.extra.enableMipmapTextureFiltering, f.enableSoftShadow,
             f.extra.enableTransparency, f.extra.enableBvhSahBinning })
        v.push_back(b ? 1 : 0);
    v.push_back(double(c.cameras.size()));
    for (const CameraConfig& cam : c.cameras)
        for (float x : { cam.fieldOfView, cam.distanceFromLookAt, cam.lookAt.x, cam.lookAt.y, cam.lookAt.z, cam.rotation.x, cam.rotation.y, cam.rotation.z })
            v.push_back(x);
    v.push_back(double(c.lights.size()));
    for (const auto& l : c.lights) {
        double row[22] = {};
        auto put = [&](int at, const glm::vec3& p) { row[at] = p.x, row[at + 1] = p.y, row[at + 2] = p.z; };
        if (std::holds_alternative<PointLight>(l)) {
            const auto& q = std::get<PointLight>(l);
            row[0] = 0, put(1, q.position), put(10, q.color);
        } else if (std::holds_alternative<SegmentLight>(l)) {
            const auto& q = std::get<SegmentLight>(l);
            row[0] = 1, put(1, q.endpoint0), put(4, q.endpoint1), put(10, q.color0), put(13, q.color1);
        } else {
            const auto& q = std::get<ParallelogramLight>(l);
            row[0] = 2, put(1, q.v0), put(4, q.edge01), put(7, q.edge02), put(10, q.color0), put(13, q.color1), put(16, q.color2), put(19, q.color3);
        }
        v.insert(v.end(), row, row + 22);
    }
    if (int(v.size()) > capacity)
        return -1;
    std::copy(v.begin(), v.end(), out);
    const std::string name = file ? std::get<std::filesystem::path>(c.scene).stem().string() : serialize(std::get<SceneType>(c.scene));
    std::snprintf(sceneName, size_t(nameCapacity), "%s", name.c_str());
    return int(v.size());
}

// ---------------------------------------------------------------- scenes ----
void* ref_scene_prebuilt(int sceneType)
{
    auto* s = new RefScene;
    try {
        s->scene = loadScenePrebuilt(static_cast<SceneType>(sceneType), DATA_DIR);
    } catch (...) {
        delete s;
        return nullptr;
    }
    return s;
}

void* ref_scene_empty() { return new RefScene; }
// loadMesh (framework/src/mesh.cpp:52-149) on an arbitrary OBJ file; NULL on failure
void* ref_scene_load_obj(const char* path, int centerAndNormalize)
{
    try {
        auto* s = new RefScene;
        s->scene.meshes = loadMesh(path, centerAndNormalize != 0);
        return s;
    } catch (...) {
        return nullptr;
    }
}
void ref_scene_free(void* p) { delete static_cast<RefScene*>(p); }

// verts: nv x 8 floats (position, normal, texCoord); mat: kd3 ks3 shininess transparency
void ref_scene_add_mesh(void* p, const float* verts, int nv, const uint32_t* tris, int nt, const float* mat,
    const float* tex, int tw, int th)
{
    auto* s = static_cast<RefScene*>(p);
    Mesh m;
    m.vertices.resize(size_t(nv));
    for (int i = 0; i < nv; i++) {
        const float* v = verts + size_t(i) * 8;
        m.vertices[size_t(i)] = Vertex { v3(v), v3(v + 3), glm::vec2(v[6], v[7]) };
    }
    m.triangles.resize(size_t(nt));
    for (int i = 0; i < nt; i++)
        m.triangles[size_t(i)] = glm::uvec3(tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]);
    m.material.kd = v3(mat);
    m.material.ks = v3(mat + 3);
    m.material.shininess = mat[6];
    m.material.transparency = mat[7];
    if (tex)
        m.material.kdTexture = std::shared_ptr<Image>(makeImage(tex, tw, th), freeImage);
    s->scene.meshes.push_back(std::move(m));
}

void ref_scene_add_light(void* p, const rtx_light* l)
{
    auto* s = static_cast<RefScene*>(p);
    if (l->type == RTX_LIGHT_POINT)
        s->scene.lights.emplace_back(PointLight { v3(l->p[0]), v3(l->c[0]) });
    else if (l->type == RTX_LIGHT_SEGMENT)
        s->scene.lights.emplace_back(SegmentLight { v3(l->p[0]), v3(l->p[1]), v3(l->c[0]), v3(l->c[1]) });
    else
        s->scene.lights.emplace_back(ParallelogramLight { v3(l->p[0]), v3(l->p[1]), v3(l->p[2]), v3(l->c[0]), v3(l->c[1]), v3(l->c[2]), v3(l->c[3]) });
}
void ref_scene_clear_lights(void* p) { static_cast<RefScene*>(p)->scene.lights.clear(); }

void ref_scene_add_sphere(void* p, const float* center, float radius, const float* mat)
{
    auto* s = static_cast<RefScene*>(p);
    Sphere sp;
    sp.center = v3(center);
    sp.radius = radius;
    sp.material.kd = v3(mat);
    sp.material.ks = v3(mat + 3);
    sp.material.shininess = mat[6];
    sp.material.transparency = mat[7];
    s->scene.spheres.push_back(sp);
}

// rgb == NULL: constant-colour map
void ref_scene_set_envmap(void* p, const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* bg)
{
    auto* s = static_cast<RefScene*>(p);
    if (rgb) {
        Image* im = makeImage(rgb, w, h);
        s->envs.push_back(std::make_unique<EnvironmentMap>(*im, mapping == RTX_ENV_SPHERICAL ? SPHERICAL : CYLINDRICAL, vfovDegrees, v3(bg)));
        freeImage(im);
    } else {
        s->envs.push_back(std::make_unique<EnvironmentMap>(v3(bg)));
    }
    s->scene.environmentMap.clear();
    s->scene.environmentMap.push_back(s->envs.back().get());
}

int ref_scene_num_meshes(void* p) { return int(static_cast<RefScene*>(p)->scene.meshes.size()); }
void ref_mesh_sizes(void* p, int m, int* nv, int* nt, int* tw, int* th)
{
    const Mesh& mesh = static_cast<RefScene*>(p)->scene.meshes[size_t(m)];
    *nv = int(mesh.vertices.size());
    *nt = int(mesh.triangles.size());
    *tw = mesh.material.kdTexture ? mesh.material.kdTexture->width : 0;
    *th = mesh.material.kdTexture ? mesh.material.kdTexture->height : 0;
}
void ref_mesh_copy(void* p, int m, float* verts, uint32_t* tris, float* mat, float* tex)
{
    const Mesh& mesh = static_cast<RefScene*>(p)->scene.meshes[size_t(m)];
    for (size_t i = 0; i < mesh.vertices.size(); i++) {
        put3(verts + 8 * i, mesh.vertices[i].position);
        put3(verts + 8 * i + 3, mesh.vertices[i].normal);
        verts[8 * i + 6] = mesh.vertices[i].texCoord.x;
        verts[8 * i + 7] = mesh.vertices[i].texCoord.y;
    }
    for (size_t i = 0; i < mesh.triangles.size(); i++)
        for (int k = 0; k < 3; k++)
            tris[3 * i + size_t(k)] = mesh.triangles[i][k];
    put3(mat, mesh.material.kd);
    put3(mat + 3, mesh.material.ks);
    mat[6] = mesh.material.shininess;
    mat[7] = mesh.material.transparency;
    if (tex && mesh.material.kdTexture)
        std::memcpy(tex, mesh.material.kdTexture->pixels.data(), mesh.material.kdTexture->pixels.size() * 3 * sizeof(float));
}
int ref_scene_num_lights(void* p) { return int(static_cast<RefScene*>(p)->scene.lights.size()); }
void ref_scene_get_light(void* p, int i, rtx_light* out)
{
    const auto& l = static_cast<RefScene*>(p)->scene.lights[size_t(i)];
    std::memset(out, 0, sizeof(*out));
    if (std::holds_alternative<PointLight>(l)) {
        const auto& q = std::get<PointLight>(l);
        out->type = RTX_LIGHT_POINT;
        put3(out->p[0], q.position);
        put3(out->c[0], q.color);
    } else if (std::holds_alternative<SegmentLight>(l)) {
        const auto& q = std::get<SegmentLight>(l);
        out->type = RTX_LIGHT_SEGMENT;
        put3(out->p[0], q.endpoint0);
        put3(out->p[1], q.endpoint1);
        put3(out->c[0], q.color0);
        put3(out->c[1], q.color1);
    } else {
        const auto& q = std::get<ParallelogramLight>(l);
        out->type = RTX_LIGHT_PARALLELOGRAM;
        put3(out->p[0], q.v0);
        put3(out->p[1], q.edge01);
        put3(out->p[2], q.edge02);
        put3(out->c[0], q.color0);
        put3(out->c[1], q.color1);
        put3(out->c[2], q.color2);
        put3(out->c[3], q.color3);
    }
}
int ref_scene_num_spheres(void* p) { return int(static_cast<RefScene*>(p)->scene.spheres.size()); }
void ref_scene_get_sphere(void* p, int i, float* center, float* radius, float* mat)
{
    const Sphere& s = static_cast<RefScene*>(p)->scene.spheres[size_t(i)];
    put3(center, s.center);
    *radius = s.radius;
    put3(mat, s.material.kd);
    put3(mat + 3, s.material.ks);
    mat[6] = s.material.shininess;
    mat[7] = s.material.transparency;
}

// Load an image file through the reference's Image (stb, 8-bit -> float).
int ref_image_load(const char* path, int* w, int* h, float* rgb /* may be NULL to query size */)
{
    try {
        Image im { std::filesystem::path(path) };
        *w = im.width;
        *h = im.height;
        if (rgb)
            std::memcpy(rgb, im.pixels.data(), im.pixels.size() * 3 * sizeof(float));
        return 0;
    } catch (...) {
        return -1;
    }
}
const char* ref_data_dir() { return DATA_DIR; }

// ------------------------------------------------------------------- BVH ----
void* ref_bvh_create(void* p, const rtx_features* f)
{
    auto* s = static_cast<RefScene*>(p);
    return new BvhInterface(&s->scene, toFeatures(f));
}
void ref_bvh_free(void* b)
{
    auto* bi = static_cast<BvhInterface*>(b);
    delete bi->m_impl;
    delete bi;
}
int ref_bvh_levels(void* b) { return static_cast<BvhInterface*>(b)->numLevels(); }
int ref_bvh_leaves(void* b) { return static_cast<BvhInterface*>(b)->numLeaves(); }
int64_t ref_bvh_num_nodes(void* b) { return int64_t(static_cast<BvhInterface*>(b)->m_impl->nodes.size()); }
int64_t ref_bvh_num_children(void* b)
{
    int64_t n = 0;
    for (const Node& nd : static_cast<BvhInterface*>(b)->m_impl->nodes)
        n += int64_t(nd.children.size());
    return n;
}
// aabb: n x 6; isLeaf: n; childOffset: n+1; children: total
int64_t ref_bvh_dump(void* b, float* aabb, int32_t* isLeaf, int64_t* childOffset, int64_t* children)
{
    const auto* impl = static_cast<BvhInterface*>(b)->m_impl;
    int64_t off = 0;
    for (size_t i = 0; i < impl->nodes.size(); i++) {
        const Node& nd = impl->nodes[i];
        put3(aabb + 6 * i, nd.axisAlignedBox.lower);
        put3(aabb + 6 * i + 3, nd.axisAlignedBox.upper);
        isLeaf[i] = nd.isLeaf ? 1 : 0;
        childOffset[i] = off;
        for (size_t c : nd.children)
            children[off++] = int64_t(c);
    }
    childOffset[impl->nodes.size()] = off;
    return impl->nodes.empty() ? -1 : int64_t(impl->root);
}

// Batch of BvhInterface::intersect calls.  The triangle id is recovered from
// HitInfo::mesh / HitInfo::triangle (the reference does not export minTri).
void ref_intersect(void* b, void* p, const rtx_ray* rays, int64_t n, const rtx_features* f, rtx_hit* out)
{
    auto* s = static_cast<RefScene*>(p);
    const auto* bvh = static_cast<BvhInterface*>(b);
    const Features feat = toFeatures(f);
    std::vector<int64_t> meshBase(s->scene.meshes.size() + 1, 0);
    for (size_t m = 0; m < s->scene.meshes.size(); m++)
        meshBase[m + 1] = meshBase[m] + int64_t(s->scene.meshes[m].triangles.size());
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++) {
        Ray r { v3(rays[i].origin), v3(rays[i].direction), rays[i].t };
        HitInfo hi;
        hi.mesh = nullptr;
        hi.normal = glm::vec3(0.f);
        hi.barycentricCoord = glm::vec3(0.f);
        hi.texCoord = glm::vec2(0.f);
        const bool hit = bvh->intersect(r, hi, feat);
        rtx_hit& o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.t = r.t;
        o.hit = hit ? 1 : 0;
        o.tri = RTX_TRI_NONE;
        o.material = -1;
        if (hit) {
            put3(o.normal, hi.normal);
            if (hi.mesh) {
                const size_t m = size_t(hi.mesh - s->scene.meshes.data());
                const auto& tris = hi.mesh->triangles;
                // all triangles of this mesh with these indices and this hit distance
                int64_t found = -1;
                for (size_t j = 0; j < tris.size() && found < 0; j++)
                    if (tris[j] == hi.triangle)
                        found = int64_t(j);
                o.tri = int32_t(meshBase[m] + found);
                o.material = int32_t(m);
                put3(o.bary, hi.barycentricCoord);
                o.uv[0] = hi.texCoord.x;
                o.uv[1] = hi.texCoord.y;
            } else {
                // sphere hit: which sphere is not recorded by the reference
                o.tri = -2;
            }
        }
    }
}

// ---------------------------------------------------------------- render ----
static int defaultDepth(int driver)
{
    return (driver == RTX_DRIVER_MRAYS || driver == RTX_DRIVER_DOF) ? 0 : 1;
}

// mode 0: call the reference's own renderRayTracing* driver (RNG-driven drivers
//         run on one OpenMP thread so their default-seeded engine is one stream).
// mode 1: harness-owned pixel loop around getFinalColor /
//         calculateColorTransparency with the same per-pixel arithmetic, which
//         lets ray_depth be chosen and keys the RNG shim per pixel.
int ref_render(void* p, void* b, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* rp,
    float* outRgb, int mode, int threads)
{
    auto* s = static_cast<RefScene*>(p);
    const auto* bvh = static_cast<BvhInterface*>(b);
    const Features feat = toFeatures(f);
    const int W = rp->width, H = rp->height;
    sampleSize = rp->shadow_samples > 0 ? rp->shadow_samples : 50;
    numberOfRays = rp->glossy_rays > 0 ? rp->glossy_rays : 5;

    float aspect = float(W) / float(H); // Window::getAspectRatio, framework/src/window.cpp:379-384
    const float fovy = cam->fovy;
    Trackball camera(reinterpret_cast<Window*>(&aspect), fovy, v3(cam->look_at), cam->distance, cam->euler[0], cam->euler[1]);
    camera.setCamera(v3(cam->look_at), v3(cam->euler), cam->distance);
    Screen screen(glm::ivec2(W, H), false);

    const int driver = rp->driver;
    const int depth = rp->ray_depth >= 0 ? rp->ray_depth : defaultDepth(driver);
    const int spp = rp->samples > 0 ? rp->samples : (driver == RTX_DRIVER_MRAYS ? 64 : driver == RTX_DRIVER_DOF ? 100 : 7);
    const float aperture = rp->aperture > 0.f ? rp->aperture : 0.1f;
    const float focal = rp->focal_length > 0.f ? rp->focal_length : 2.0f;

    if (mode == 0) {
        const int prevThreads = omp_get_max_threads();
        switch (driver) {
        case RTX_DRIVER_PLAIN:
            renderRayTracing(s->scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_MRAYS:
            omp_set_num_threads(threads > 0 ? threads : 1);
            renderRayTracingMRaysPerPixel(s->scene, camera, *bvh, screen, feat, spp);
            break;
        case RTX_DRIVER_DOF:
            omp_set_num_threads(threads > 0 ? threads : 1);
            renderRayTracingDepthOfField(s->scene, camera, *bvh, screen, feat, aperture, focal, spp);
            break;
        case RTX_DRIVER_TRANSPARENCY:
            if (threads > 0)
                omp_set_num_threads(threads);
            renderRayTracingTransparency(s->scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_MOTION_BLUR: {
            omp_set_num_threads(threads > 0 ? threads : 1);
            Trackball prev = camera;
            prev.setCamera(v3(rp->prev_camera.look_at), v3(rp->prev_camera.euler), rp->prev_camera.distance);
            // first call stores the start pose (or blurs from a stale one); its image is discarded
            renderRayTracingMotionBlur(s->scene, prev, *bvh, screen, feat, spp);
            renderRayTracingMotionBlur(s->scene, camera, *bvh, screen, feat, spp);
        } break;
        default:
            return -1;
        }
        omp_set_num_threads(prevThreads);
    } else {
        // One sequential engine for the driver-level draws, consumed in pixel
        // order exactly like the reference's single-thread run.
        std::default_random_engine rng;
        std::uniform_real_distribution<float> dist01(0.f, 1.f);
        std::uniform_real_distribution<float> distAp(-aperture / 2, aperture / 2);
        rtx_shim_seed = rp->seed;
        for (int y = 0; y < H; y++) {
            for (int x = 0; x != W; x++) {
                rtx_shim_pixel = uint32_t(y) * uint32_t(W) + uint32_t(x);
                rtx_shim_sample = 0;
                rtx_shim_counter = 0;
                const glm::vec2 ndc { float(x) / float(W) * 2.0f - 1.0f, float(y) / float(H) * 2.0f - 1.0f };
                glm::vec3 col(0.f);
                switch (driver) {
                case RTX_DRIVER_PLAIN:
                    col = getFinalColor(s->scene, *bvh, camera.generateRay(ndc), feat, depth);
                    break;
                case RTX_DRIVER_MRAYS:
                    for (int i = 0; i < spp; i++) {
                        rtx_shim_sample = uint32_t(i);
                        rtx_shim_counter = 0;
                        const glm::vec2 q { float(x + dist01(rng)) / float(W) * 2.0f - 1.0f, float(y + dist01(rng)) / float(H) * 2.0f - 1.0f };
                        col += getFinalColor(s->scene, *bvh, camera.generateRay(q), feat, depth);
                    }
                    col *= glm::vec3(1.0f / (spp * 1.0f));
                    break;
                case RTX_DRIVER_DOF: {
                    Ray cr = camera.generateRay(ndc);
                    glm::vec3 fp = getFocalPoint(cr, focal);
                    std::vector<Ray> rays = sampledRays(fp, generateSamples(cr.origin, aperture, spp, rng, distAp));
                    uint32_t si = 0;
                    for (auto r : rays) {
                        rtx_shim_sample = si++;
                        rtx_shim_counter = 0;
                        col += getFinalColor(s->scene, *bvh, r, feat, depth);
                    }
                    col = col * glm::vec3(1.0f / (rays.size()));
                } break;
                case RTX_DRIVER_TRANSPARENCY: {
                    Ray cr = camera.generateRay(ndc);
                    col = calculateColorTransparency(s->scene, cr, *bvh, feat, depth);
                } break;
                default:
                    return -1;
                }
                screen.setPixel(x, y, col);
            }
        }
    }
    std::memcpy(outRgb, screen.pixels().data(), size_t(W) * size_t(H) * 3 * sizeof(float));
    return 0;
}

// getFinalColor for a batch of rays (sequential: the RNG shim's key is thread-local state)
void ref_shade(void* p, void* b, const rtx_ray* rays, int64_t n, const rtx_features* f, const rtx_render_params* rp, float* out)
{
    auto* s = static_cast<RefScene*>(p);
    const auto* bvh = static_cast<BvhInterface*>(b);
    const Features feat = toFeatures(f);
    sampleSize = rp->shadow_samples > 0 ? rp->shadow_samples : 50;
    numberOfRays = rp->glossy_rays > 0 ? rp->glossy_rays : 5;
    const int depth = rp->ray_depth >= 0 ? rp->ray_depth : 0;
    rtx_shim_seed = rp->seed;
    for (int64_t i = 0; i < n; i++) {
        rtx_shim_pixel = uint32_t(i);
        rtx_shim_sample = 0;
        rtx_shim_counter = 0;
        Ray r { v3(rays[i].origin), v3(rays[i].direction), rays[i].t };
        put3(out + 3 * i, getFinalColor(s->scene, *bvh, r, feat, depth));
    }
}

// ------------------------------------------------- per-function known answers
int ref_kat_slab(const float* lo, const float* hi, rtx_ray* ray)
{
    Ray r { v3(ray->origin), v3(ray->direction), ray->t };
    const bool ok = intersectRayWithShape(AxisAlignedBox { v3(lo), v3(hi) }, r);
    ray->t = r.t;
    return ok;
}
int ref_kat_triangle(const float* a, const float* b, const float* c, rtx_ray* ray, float* bary, float* n)
{
    Ray r { v3(ray->origin), v3(ray->direction), ray->t };
    HitInfo hi;
    hi.barycentricCoord = glm::vec3(0.f);
    hi.normal = glm::vec3(0.f);
    const bool ok = intersectRayWithTriangle(v3(a), v3(b), v3(c), r, hi);
    ray->t = r.t;
    put3(bary, hi.barycentricCoord);
    put3(n, hi.normal);
    return ok;
}
int ref_kat_sphere(const float* center, float radius, rtx_ray* ray, float* n)
{
    Ray r { v3(ray->origin), v3(ray->direction), ray->t };
    HitInfo hi;
    hi.normal = glm::vec3(0.f);
    Sphere sp;
    sp.center = v3(center);
    sp.radius = radius;
    const bool ok = intersectRayWithShape(sp, r, hi);
    ray->t = r.t;
    put3(n, hi.normal);
    return ok;
}
void ref_kat_barycentric(const float* a, const float* b, const float* c, const float* pnt, float* out)
{
    put3(out, computeBarycentricCoord(v3(a), v3(b), v3(c), v3(pnt)));
}
// computeShading with an untextured material; mat = kd3 ks3 shininess transparency
void ref_kat_shading(const float* lightPos, const float* lightCol, const rtx_ray* ray, const float* normal, const float* mat, float* out)
{
    Features feat;
    feat.enableShading = true;
    HitInfo hi;
    hi.normal = v3(normal);
    hi.barycentricCoord = glm::vec3(0.f);
    hi.texCoord = glm::vec2(0.f);
    hi.mesh = nullptr;
    hi.material.kd = v3(mat);
    hi.material.ks = v3(mat + 3);
    hi.material.shininess = mat[6];
    hi.material.transparency = mat[7];
    Ray r { v3(ray->origin), v3(ray->direction), ray->t };
    put3(out, computeShading(v3(lightPos), v3(lightCol), feat, r, hi));
}
void ref_kat_reflect(const rtx_ray* ray, const float* normal, rtx_ray* out)
{
    HitInfo hi;
    hi.normal = v3(normal);
    hi.mesh = nullptr;
    Ray r { v3(ray->origin), v3(ray->direction), ray->t };
    const Ray q = computeReflectionRay(r, hi);
    put3(out->origin, q.origin);
    put3(out->direction, q.direction);
    out->t = q.t;
}
// mode 0 nearest (acquireTexel), 1 bilinear; n lookups of (u,v)
void ref_kat_texture(const float* rgb, int w, int h, int mode, const float* uv, int n, float* out)
{
    Image* im = makeImage(rgb, w, h);
    Features feat;
    for (int i = 0; i < n; i++) {
        const glm::vec2 t(uv[2 * i], uv[2 * i + 1]);
        put3(out + 3 * i, mode == 0 ? acquireTexel(*im, t, feat) : bilinearInterpolation(*im, t, feat));
    }
    freeImage(im);
}
// mip chain through getMipMap; returns the level count; sizes[2*l] = w, h; pixels
// concatenated level after level.  Pass pixels == NULL to query.
int ref_kat_mipmap(const float* rgb, int w, int h, int* sizes, float* pixels)
{
    Image* im = makeImage(rgb, w, h);
    ImageMipMap mm = getMipMap(*im);
    size_t off = 0;
    for (size_t l = 0; l < mm.pixels.size(); l++) {
        if (sizes) {
            sizes[2 * l] = mm.width[l];
            sizes[2 * l + 1] = mm.height[l];
        }
        if (pixels)
            std::memcpy(pixels + off, mm.pixels[l].data(), mm.pixels[l].size() * 3 * sizeof(float));
        off += mm.pixels[l].size() * 3;
    }
    freeImage(im);
    return int(mm.pixels.size());
}
// EnvironmentMap::getColor for n directions
void ref_kat_envmap(const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* bg, int bilinear,
    const float* dirs, int n, float* out)
{
    Image* im = makeImage(rgb, w, h);
    EnvironmentMap env(*im, mapping == RTX_ENV_SPHERICAL ? SPHERICAL : CYLINDRICAL, vfovDegrees, v3(bg));
    Features feat;
    feat.extra.enableBilinearTextureFiltering = bilinear;
    for (int i = 0; i < n; i++) {
        Ray r { glm::vec3(0.f), v3(dirs + 3 * i) };
        put3(out + 3 * i, env.getColor(r, feat));
    }
    freeImage(im);
}
int ref_kat_env_bins(const float* rgb, int w, int h, float* out, int cap)
{
    Image* im = makeImage(rgb, w, h);
    EnvironmentMap env(*im, SPHERICAL, 180.f, glm::vec3(0.f));
    for (size_t i = 0; i < env.radianceBins.size() && int(i) < cap; i++) {
        out[4 * i + 0] = env.radianceBins[i].lower.x, out[4 * i + 1] = env.radianceBins[i].lower.y;
        out[4 * i + 2] = env.radianceBins[i].upper.x, out[4 * i + 3] = env.radianceBins[i].upper.y;
    }
    const int n = int(env.radianceBins.size());
    freeImage(im);
    return n;
}
void ref_kat_sample2d(const float* rect, int h, int k, uint32_t seed, uint32_t pixel, uint32_t index, float* out)
{
    rtx_shim_seed = seed;
    rtx_shim_pixel = pixel;
    rtx_shim_sample = 0;
    rtx_shim_counter = index;
    AxisAlignedRectangle r { glm::vec2(rect[0], rect[1]), glm::vec2(rect[2], rect[3]) };
    const std::vector<glm::vec2> v = sample2D(r, h, k);
    for (size_t i = 0; i < v.size(); i++)
        out[2 * i] = v[i].x, out[2 * i + 1] = v[i].y;
}
int ref_kat_env_rays(const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* position, const float* normal, int n,
    uint32_t seed, uint32_t pixel, rtx_ray* out, int cap)
{
    Image* im = makeImage(rgb, w, h);
    EnvironmentMap env(*im, mapping == RTX_ENV_SPHERICAL ? SPHERICAL : CYLINDRICAL, vfovDegrees, glm::vec3(0.f));
    rtx_shim_seed = seed;
    rtx_shim_pixel = pixel;
    rtx_shim_sample = 0;
    rtx_shim_counter = 0;
    const std::vector<Ray> rays = env.getSamplingRay(v3(position), v3(normal), n);
    for (size_t i = 0; i < rays.size() && int(i) < cap; i++) {
        put3(out[i].origin, rays[i].origin);
        put3(out[i].direction, rays[i].direction);
        out[i].t = rays[i].t;
    }
    freeImage(im);
    return int(rays.size());
}
// light sampling with the keyed RNG shim: n samples of light l for (seed,pixel), counter from 0
void ref_kat_light_samples(const rtx_light* l, uint32_t seed, uint32_t pixel, int n, float* pos, float* col)
{
    rtx_shim_seed = seed;
    rtx_shim_pixel = pixel;
    rtx_shim_sample = 0;
    rtx_shim_counter = 0;
    for (int i = 0; i < n; i++) {
        glm::vec3 pp, cc;
        if (l->type == RTX_LIGHT_SEGMENT)
            sampleSegmentLight(SegmentLight { v3(l->p[0]), v3(l->p[1]), v3(l->c[0]), v3(l->c[1]) }, pp, cc);
        else
            sampleParallelogramLight(ParallelogramLight { v3(l->p[0]), v3(l->p[1]), v3(l->p[2]), v3(l->c[0]), v3(l->c[1]), v3(l->c[2]), v3(l->c[3]) }, pp, cc);
        put3(pos + 3 * i, pp);
        put3(col + 3 * i, cc);
    }
}
// glossy ray fan with the keyed RNG shim
void ref_kat_glossy(const rtx_ray* reflection, float degreeBlur, int nRays, uint32_t seed, uint32_t pixel, rtx_ray* out)
{
    rtx_shim_seed = seed;
    rtx_shim_pixel = pixel;
    rtx_shim_sample = 0;
    rtx_shim_counter = 0;
    numberOfRays = nRays;
    Ray r { v3(reflection->origin), v3(reflection->direction), reflection->t };
    std::vector<Ray> rays = glossyRays(r, degreeBlur);
    for (size_t i = 0; i < rays.size(); i++) {
        put3(out[i].origin, rays[i].origin);
        put3(out[i].direction, rays[i].direction);
        out[i].t = rays[i].t;
    }
}
// camera: position + n rays for ndc points (the reference's Trackball methods; see camera_screen_tu.cpp)
void ref_kat_camera(const rtx_camera* cam, float fovy, float aspect, const float* ndc, int n, rtx_ray* out)
{
    Trackball camera(reinterpret_cast<Window*>(&aspect), fovy, v3(cam->look_at), cam->distance, cam->euler[0], cam->euler[1]);
    camera.setCamera(v3(cam->look_at), v3(cam->euler), cam->distance);
    for (int i = 0; i < n; i++) {
        const Ray r = camera.generateRay(glm::vec2(ndc[2 * i], ndc[2 * i + 1]));
        put3(out[i].origin, r.origin);
        put3(out[i].direction, r.direction);
        out[i].t = r.t;
    }
}

// draws of std::default_random_engine through std::uniform_real_distribution<float>(a, b)
// after `skip` discarded draws: the stream the reference's single-thread drivers consume
void ref_kat_minstd(uint64_t skip, int n, float a, float b, float* out)
{
    std::default_random_engine rng;
    std::uniform_real_distribution<float> dist(a, b);
    for (uint64_t i = 0; i < skip; i++)
        (void)dist(rng);
    for (int i = 0; i < n; i++)
        out[i] = dist(rng);
}

} // extern "C"
