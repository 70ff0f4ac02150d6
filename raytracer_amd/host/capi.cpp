// Flat C entry points over the C++ host layer so that Python (ctypes) can drive
// it: scene assembly from arrays, host-only flattening (no GPU needed), and the
// GPU-backed BvhInterface / render drivers.  Errors are returned as negative
// codes with text in rtxh_last_error().
#include <cstring>
#include <memory>
#include <string>

#include "config.h"
#include "raytracer.h"
#include "scene_io.h"

using namespace rt;

namespace {

thread_local std::string t_err;

struct HostScene {
    Scene scene;
    std::vector<std::unique_ptr<EnvironmentMap>> envs;
};

vec3 v3(const float* p) { return vec3(p[0], p[1], p[2]); }

Material toMaterial(const float* m, const float* tex, int tw, int th)
{
    Material o;
    o.kd = v3(m);
    o.ks = v3(m + 3);
    o.shininess = m[6];
    o.transparency = m[7];
    if (tex) {
        auto img = std::make_shared<Image>();
        img->width = tw;
        img->height = th;
        img->pixels.resize(size_t(tw) * size_t(th));
        std::memcpy(static_cast<void*>(img->pixels.data()), tex, size_t(tw) * size_t(th) * 12);
        o.kdTexture = img;
    }
    return o;
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

Trackball toTrackball(const rtx_camera* c, int W, int H)
{
    const float aspect = (W == 0 || H == 0) ? 1.0f : float(W) / float(H);
    Trackball t(aspect, c->fovy, v3(c->look_at), c->distance, c->euler[0], c->euler[1]);
    t.setCamera(v3(c->look_at), v3(c->euler), c->distance);
    return t;
}

template <class F>
int guarded(F&& fn)
{
    try {
        fn();
        return 0;
    } catch (const std::exception& e) {
        t_err = e.what();
        return -1;
    }
}

} // namespace

extern "C" {

const char* rtxh_last_error() { return t_err.c_str(); }

int rtxh_select_device(int device)
{
    return guarded([&] { selectDevice(device); });
}

// readConfigFile flattened for tests: out[0] command_line_rendering, [1..2] window size, [3] the scene is a
// file (1) or a prebuilt scene (0), [4] the prebuilt scene's number, [5..10] the six basic feature switches,
// [11..18] bloom, multiple rays, motion blur, depth of field, glossy, environment map, bilinear, mipmap,
// [19..21] soft shadows, transparency, SAH binning (extended keys), [22] number of cameras, then 8 numbers
// per camera, then the number of lights and 22 numbers per light (kind, three points, four colours);
// sceneName receives the scene's name as the output files carry it.  Returns the count of numbers written.
int rtxh_read_config(const char* path, const char* dataDir, int extended, double* out, int capacity, char* sceneName, int nameCapacity)
{
    int written = -1;
    const int rc = guarded([&] {
        const Config c = readConfigFile(path, dataDir, extended != 0);
        std::vector<double> v;
        v.push_back(c.cliRenderingEnabled);
        v.push_back(c.windowSize.x);
        v.push_back(c.windowSize.y);
        const bool file = std::holds_alternative<std::string>(c.scene);
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
            auto put = [&](int at, const vec3& p) { row[at] = p.x, row[at + 1] = p.y, row[at + 2] = p.z; };
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
            throw RtError("rtxh_read_config: output buffer too small");
        std::copy(v.begin(), v.end(), out);
        written = int(v.size());
        std::string name = file ? std::get<std::string>(c.scene) : serialize(std::get<SceneType>(c.scene));
        if (file) { // std::filesystem::path::stem
            const size_t slash = name.find_last_of('/');
            name = slash == std::string::npos ? name : name.substr(slash + 1);
            const size_t dot = name.find_last_of('.');
            if (dot != std::string::npos && dot > 0)
                name = name.substr(0, dot);
        }
        std::snprintf(sceneName, size_t(nameCapacity), "%s", name.c_str());
    });
    return rc ? rc : written;
}

// Where the next BvhInterface builds its tree: 0 on the host (the reference's builder), 1 on the GPU.
void rtxh_set_device_build(int on) { buildTreeOnDevice = on != 0; }

void* rtxh_scene_create() { return new HostScene; }
void rtxh_scene_free(void* p) { delete static_cast<HostScene*>(p); }

int rtxh_scene_add_mesh(void* p, const float* verts, int nv, const uint32_t* tris, int nt, const float* mat, const float* tex, int tw, int th)
{
    return guarded([&] {
        auto* s = static_cast<HostScene*>(p);
        Mesh m;
        m.vertices.resize(size_t(nv));
        for (int i = 0; i < nv; i++) {
            const float* v = verts + size_t(i) * 8;
            m.vertices[size_t(i)].position = v3(v);
            m.vertices[size_t(i)].normal = v3(v + 3);
            m.vertices[size_t(i)].texCoord = vec2 { v[6], v[7] };
        }
        m.triangles.resize(size_t(nt));
        for (int i = 0; i < nt; i++) {
            m.triangles[size_t(i)] = uvec3 { tris[3 * i], tris[3 * i + 1], tris[3 * i + 2] };
            for (int k = 0; k < 3; k++)
                if (tris[3 * i + k] >= uint32_t(nv))
                    throw RtError("triangle index out of range");
        }
        m.material = toMaterial(mat, tex, tw, th);
        s->scene.meshes.push_back(std::move(m));
    });
}

// loadMesh (framework/src/mesh.cpp:52-149): append the meshes of an OBJ file
int rtxh_scene_load_obj(void* p, const char* path, int centerAndNormalize)
{
    return guarded([&] {
        auto* s = static_cast<HostScene*>(p);
        for (Mesh& m : loadMesh(path, centerAndNormalize != 0))
            s->scene.meshes.push_back(std::move(m));
    });
}
int rtxh_scene_num_meshes(const void* p) { return int(static_cast<const HostScene*>(p)->scene.meshes.size()); }
void rtxh_mesh_sizes(const void* p, int m, int* nv, int* nt, int* tw, int* th)
{
    const Mesh& mesh = static_cast<const HostScene*>(p)->scene.meshes[size_t(m)];
    *nv = int(mesh.vertices.size());
    *nt = int(mesh.triangles.size());
    *tw = mesh.material.kdTexture ? mesh.material.kdTexture->width : 0;
    *th = mesh.material.kdTexture ? mesh.material.kdTexture->height : 0;
}
// verts: nv x 8 (position, normal, uv); tris: nt x 3; mat: kd, ks, shininess, transparency; tex: th x tw x 3 or NULL
void rtxh_mesh_copy(const void* p, int m, float* verts, uint32_t* tris, float* mat, float* tex)
{
    const Mesh& mesh = static_cast<const HostScene*>(p)->scene.meshes[size_t(m)];
    for (size_t i = 0; i < mesh.vertices.size(); i++) {
        const Vertex& v = mesh.vertices[i];
        const float row[8] = { v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.texCoord.x, v.texCoord.y };
        std::memcpy(verts + 8 * i, row, sizeof(row));
    }
    for (size_t i = 0; i < mesh.triangles.size(); i++)
        tris[3 * i] = mesh.triangles[i].x, tris[3 * i + 1] = mesh.triangles[i].y, tris[3 * i + 2] = mesh.triangles[i].z;
    const Material& mt = mesh.material;
    const float mrow[8] = { mt.kd.x, mt.kd.y, mt.kd.z, mt.ks.x, mt.ks.y, mt.ks.z, mt.shininess, mt.transparency };
    std::memcpy(mat, mrow, sizeof(mrow));
    if (tex && mt.kdTexture)
        std::memcpy(tex, static_cast<const void*>(mt.kdTexture->pixels.data()), mt.kdTexture->pixels.size() * 12);
}
// Image (framework/src/image.cpp:13-36): size query with rgb == NULL, then the pixels
int rtxh_image_load(const char* path, int* w, int* h, float* rgb)
{
    return guarded([&] {
        const Image img = loadImage(path);
        *w = img.width;
        *h = img.height;
        if (rgb)
            std::memcpy(rgb, static_cast<const void*>(img.pixels.data()), img.pixels.size() * 12);
    });
}

int rtxh_scene_add_light(void* p, const rtx_light* l)
{
    return guarded([&] {
        auto* s = static_cast<HostScene*>(p);
        if (l->type == RTX_LIGHT_POINT)
            s->scene.lights.emplace_back(PointLight { v3(l->p[0]), v3(l->c[0]) });
        else if (l->type == RTX_LIGHT_SEGMENT)
            s->scene.lights.emplace_back(SegmentLight { v3(l->p[0]), v3(l->p[1]), v3(l->c[0]), v3(l->c[1]) });
        else if (l->type == RTX_LIGHT_PARALLELOGRAM)
            s->scene.lights.emplace_back(ParallelogramLight { v3(l->p[0]), v3(l->p[1]), v3(l->p[2]), v3(l->c[0]), v3(l->c[1]), v3(l->c[2]), v3(l->c[3]) });
        else
            throw RtError("unknown light type");
    });
}

int rtxh_scene_add_sphere(void* p, const float* center, float radius, const float* mat)
{
    return guarded([&] {
        Sphere sp;
        sp.center = v3(center);
        sp.radius = radius;
        sp.material = toMaterial(mat, nullptr, 0, 0);
        static_cast<HostScene*>(p)->scene.spheres.push_back(sp);
    });
}

int rtxh_scene_set_envmap(void* p, const float* rgb, int w, int h, int mapping, float vfovDegrees, const float* bg)
{
    return guarded([&] {
        auto* s = static_cast<HostScene*>(p);
        auto e = std::make_unique<EnvironmentMap>();
        if (rgb) {
            e->texture = std::make_shared<Image>();
            e->texture->width = w;
            e->texture->height = h;
            e->texture->pixels.resize(size_t(w) * size_t(h));
            std::memcpy(static_cast<void*>(e->texture->pixels.data()), rgb, size_t(w) * size_t(h) * 12);
        }
        e->mappingType = mapping == RTX_ENV_SPHERICAL ? SPHERICAL : CYLINDRICAL;
        e->verticalFieldOfViewDegrees = vfovDegrees;
        e->backgroundColor = v3(bg);
        s->envs.push_back(std::move(e));
        s->scene.environmentMap.assign(1, s->envs.back().get());
    });
}

// ---- host-only flattening (no device) ---------------------------------------
void* rtxh_flatten(void* p, const rtx_features* f)
{
    FlatScene* out = nullptr;
    if (guarded([&] { out = new FlatScene(flattenScene(static_cast<HostScene*>(p)->scene, toFeatures(f))); }))
        return nullptr;
    return out;
}
void rtxh_flat_free(void* fl) { delete static_cast<FlatScene*>(fl); }
// out: nodes, tris, textures, texels, materials, lights, spheres, levels, leaves
void rtxh_flat_counts(const void* fl, int64_t* out)
{
    const auto* f = static_cast<const FlatScene*>(fl);
    out[0] = int64_t(f->nodes.size());
    out[1] = int64_t(f->tris.size());
    out[2] = int64_t(f->textures.size());
    out[3] = int64_t(f->texels.size() / 3);
    out[4] = int64_t(f->materials.size());
    out[5] = int64_t(f->lights.size());
    out[6] = int64_t(f->spheres.size());
    out[7] = f->numLevels;
    out[8] = f->numLeaves;
}
void rtxh_flat_copy(const void* fl, rtx_node* nodes, rtx_tri* tris, rtx_tri_attr* attrs, rtx_material* mats, rtx_texture* textures, float* texels)
{
    const auto* f = static_cast<const FlatScene*>(fl);
    // (an empty vector's data() may be null, which memcpy must not be handed even for 0 bytes)
    auto copy = [](void* dst, const auto& v) {
        if (dst && !v.empty())
            std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
    };
    copy(nodes, f->nodes);
    copy(tris, f->tris);
    copy(attrs, f->attrs);
    copy(mats, f->materials);
    copy(textures, f->textures);
    copy(texels, f->texels);
}
void rtxh_flat_root(const void* fl, int32_t* ref, float* lo, float* hi)
{
    const auto* f = static_cast<const FlatScene*>(fl);
    *ref = f->rootRef;
    std::memcpy(lo, f->rootLo, 12);
    std::memcpy(hi, f->rootHi, 12);
}
// the description rtx_scene_create takes, pointing into the FlatScene
void rtxh_flat_desc(const void* fl, rtx_scene_desc* out) { *out = static_cast<const FlatScene*>(fl)->desc(); }

// ---- GPU-backed objects --------------------------------------------------------
void* rtxh_bvh_create(void* p, const rtx_features* f)
{
    BvhInterface* b = nullptr;
    if (guarded([&] { b = new BvhInterface(&static_cast<HostScene*>(p)->scene, toFeatures(f)); }))
        return nullptr;
    return b;
}
void rtxh_bvh_free(void* b) { delete static_cast<BvhInterface*>(b); }
int rtxh_bvh_levels(const void* b) { return static_cast<const BvhInterface*>(b)->numLevels(); }
int rtxh_bvh_leaves(const void* b) { return static_cast<const BvhInterface*>(b)->numLeaves(); }
// out: levels, leaves, upload ms, build ms (the last two only for a tree built on the GPU)
int rtxh_bvh_tree_info(const void* b, float* out)
{
    rtx_tree_info info {};
    const int rc = rtx_scene_tree_info(static_cast<const BvhInterface*>(b)->deviceScene(), &info);
    out[0] = float(info.levels), out[1] = float(info.leaves), out[2] = info.upload_ms, out[3] = info.build_ms;
    return rc;
}
const void* rtxh_bvh_flat(const void* b) { return &static_cast<const BvhInterface*>(b)->flat(); }
const void* rtxh_bvh_device_scene(const void* b) { return static_cast<const BvhInterface*>(b)->deviceScene(); }

int rtxh_intersect_batch(const void* b, const rtx_ray* rays, int64_t n, const rtx_features* f, rtx_hit* hits, rtx_stats* stats)
{
    return guarded([&] { static_cast<const BvhInterface*>(b)->intersectBatch(rays, size_t(n), hits, toFeatures(f), stats); });
}

// BvhInterface::intersect for one ray, reported through HitInfo like the reference:
// out = normal3, bary3, uv2, kd3, ks3, shininess, transparency, triangle indices 3 (as float), mesh index
int rtxh_intersect_one(const void* b, rtx_ray* ray, const rtx_features* f, float* out, int* hit)
{
    return guarded([&] {
        const auto* bvh = static_cast<const BvhInterface*>(b);
        Ray r;
        r.origin = v3(ray->origin);
        r.direction = v3(ray->direction);
        r.t = ray->t;
        HitInfo hi;
        *hit = bvh->intersect(r, hi, toFeatures(f)) ? 1 : 0;
        ray->t = r.t;
        std::memset(out, 0, 20 * sizeof(float));
        if (*hit) {
            std::memcpy(out, &hi.normal, 12);
            std::memcpy(out + 3, &hi.barycentricCoord, 12);
            std::memcpy(out + 6, &hi.texCoord, 8);
            std::memcpy(out + 8, &hi.material.kd, 12);
            std::memcpy(out + 11, &hi.material.ks, 12);
            out[14] = hi.material.shininess;
            out[15] = hi.material.transparency;
            out[16] = float(hi.triangle.x), out[17] = float(hi.triangle.y), out[18] = float(hi.triangle.z);
            out[19] = hi.mesh ? float(hi.mesh - bvh->scene()->meshes.data()) : -1.f;
        }
    });
}

// getFinalColor for n rays (one batch), or ray by ray through the single-ray
// function when oneByOne is set
int rtxh_final_color(const void* b, const rtx_ray* rays, int64_t n, const rtx_features* f, int rayDepth, float* out, int oneByOne, rtx_stats* stats)
{
    return guarded([&] {
        const auto* bvh = static_cast<const BvhInterface*>(b);
        static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats");
        if (!oneByOne) {
            getFinalColorBatch(*bvh, rays, size_t(n), toFeatures(f), rayDepth, reinterpret_cast<vec3*>(out), stats);
            return;
        }
        for (int64_t i = 0; i < n; i++) {
            Ray r;
            r.origin = v3(rays[i].origin);
            r.direction = v3(rays[i].direction);
            r.t = rays[i].t;
            const vec3 c = getFinalColor(*bvh->scene(), *bvh, r, toFeatures(f), rayDepth);
            out[3 * i] = c.x, out[3 * i + 1] = c.y, out[3 * i + 2] = c.z;
        }
    });
}

// any driver, explicit parameters (renderFrame)
int rtxh_render(const void* b, const rtx_camera* cam, const rtx_features* f, const rtx_render_params* rp, float* out, rtx_stats* stats)
{
    return guarded([&] {
        const auto* bvh = static_cast<const BvhInterface*>(b);
        renderInto(*bvh, toTrackball(cam, rp->width, rp->height), rp->width, rp->height, out, toFeatures(f), *rp, stats);
    });
}

// the reference's five named drivers with their own defaults (src/render.h:18-26)
int rtxh_render_named(const void* b, const rtx_camera* cam, const rtx_camera* prevCam, const rtx_features* f, int driver, int W, int H, int samples, float aperture,
    float focal, float* out)
{
    return guarded([&] {
        const auto* bvh = static_cast<const BvhInterface*>(b);
        const Scene& scene = *bvh->scene();
        Screen screen(ivec2 { W, H });
        const Trackball camera = toTrackball(cam, W, H);
        const Features feat = toFeatures(f);
        switch (driver) {
        case RTX_DRIVER_PLAIN:
            renderRayTracing(scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_TRANSPARENCY:
            renderRayTracingTransparency(scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_DOF:
            if (samples > 0)
                renderRayTracingDepthOfField(scene, camera, *bvh, screen, feat, aperture, focal, samples);
            else
                renderRayTracingDepthOfField(scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_MRAYS:
            if (samples > 0)
                renderRayTracingMRaysPerPixel(scene, camera, *bvh, screen, feat, samples);
            else
                renderRayTracingMRaysPerPixel(scene, camera, *bvh, screen, feat);
            break;
        case RTX_DRIVER_MOTION_BLUR: {
            resetMotionBlur();
            const Trackball prev = toTrackball(prevCam, W, H);
            if (samples > 0) {
                renderRayTracingMotionBlur(scene, prev, *bvh, screen, feat, samples);
                renderRayTracingMotionBlur(scene, camera, *bvh, screen, feat, samples);
            } else {
                renderRayTracingMotionBlur(scene, prev, *bvh, screen, feat);
                renderRayTracingMotionBlur(scene, camera, *bvh, screen, feat);
            }
        } break;
        default:
            throw RtError("unknown driver");
        }
        std::memcpy(out, screen.pixels().data(), size_t(W) * size_t(H) * 12);
    });
}

// camera as the device receives it (derived fields filled in)
void rtxh_camera(const rtx_camera* cam, int W, int H, rtx_camera* out) { *out = toTrackball(cam, W, H).deviceCamera(); }

// Trackball::generateRay (framework/src/trackball.cpp:101-110) for n normalised pixel
// positions (x, y in [-1, 1], y up): the camera rays a caller hands to rtxh_intersect_batch
void rtxh_camera_rays(const rtx_camera* cam, int W, int H, const float* ndc, int64_t n, rtx_ray* out)
{
    const Trackball camera = toTrackball(cam, W, H);
    for (int64_t i = 0; i < n; i++) {
        const Ray r = camera.generateRay(vec2 { ndc[2 * i], ndc[2 * i + 1] });
        out[i].origin[0] = r.origin.x, out[i].origin[1] = r.origin.y, out[i].origin[2] = r.origin.z;
        out[i].direction[0] = r.direction.x, out[i].direction[1] = r.direction.y, out[i].direction[2] = r.direction.z;
        out[i].t = r.t;
    }
}

void rtxh_set_sampling(int shadowSamples, int glossyRays, uint32_t seed)
{
    if (shadowSamples > 0)
        sampleSize = shadowSamples;
    if (glossyRays > 0)
        numberOfRays = glossyRays;
    samplingSeed = seed;
}

int rtxh_write_bmp(const float* rgb, int W, int H, const char* path)
{
    return guarded([&] {
        Screen screen(ivec2 { W, H });
        std::memcpy(static_cast<void*>(screen.pixels().data()), rgb, size_t(W) * size_t(H) * 12);
        screen.writeBitmapToFile(path);
    });
}

} // extern "C"
