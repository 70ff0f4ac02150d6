#include "config.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

#include "scene_io.h"

namespace rt {
namespace {

// ---- a reader for the subset of TOML a configuration uses -------------------------------
struct Node;
using NodePtr = std::shared_ptr<Node>;
struct Node {
    enum Kind { Bool, Int, Float, String, Array, Table } kind = Table;
    bool b = false;
    long long i = 0;
    double f = 0;
    std::string s;
    std::vector<NodePtr> items;           // Array (also an array of tables)
    std::map<std::string, NodePtr> fields; // Table
    bool number() const { return kind == Int || kind == Float; }
    double asDouble() const { return kind == Int ? double(i) : f; }
    const Node* get(const std::string& key) const
    {
        if (kind != Table)
            return nullptr;
        auto it = fields.find(key);
        return it == fields.end() ? nullptr : it->second.get();
    }
};

struct Parser {
    const std::string& text;
    size_t at = 0;
    int line = 1;
    explicit Parser(const std::string& t) : text(t) {}

    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error("line " + std::to_string(line) + ": " + what); }
    bool done() const { return at >= text.size(); }
    char peek() const { return done() ? '\0' : text[at]; }
    void skipBlank(bool newlines)
    {
        for (;;) {
            const char c = peek();
            if (c == ' ' || c == '\t' || c == '\r' || (newlines && c == '\n')) {
                if (c == '\n')
                    line++;
                at++;
            } else if (c == '#') {
                while (!done() && peek() != '\n')
                    at++;
            } else {
                return;
            }
        }
    }
    std::string key()
    {
        skipBlank(false);
        std::string k;
        if (peek() == '"' || peek() == '\'') {
            k = string();
        } else {
            while (std::isalnum(static_cast<unsigned char>(peek())) || peek() == '_' || peek() == '-')
                k += text[at++];
        }
        if (k.empty())
            fail("expected a key");
        return k;
    }
    std::vector<std::string> dottedKey()
    {
        std::vector<std::string> path { key() };
        for (skipBlank(false); peek() == '.'; skipBlank(false)) {
            at++;
            path.push_back(key());
        }
        return path;
    }
    std::string string()
    {
        const char quote = text[at++];
        std::string out;
        while (!done() && peek() != quote) {
            char c = text[at++];
            if (c == '\n')
                fail("unterminated string");
            if (c == '\\' && quote == '"') {
                const char e = done() ? '\0' : text[at++];
                c = e == 'n' ? '\n' : e == 't' ? '\t' : e == '\\' ? '\\' : e == '"' ? '"' : e == 'r' ? '\r' : '\0';
                if (!c)
                    fail("unsupported escape in string");
            }
            out += c;
        }
        if (done())
            fail("unterminated string");
        at++;
        return out;
    }
    NodePtr value()
    {
        skipBlank(false);
        auto n = std::make_shared<Node>();
        const char c = peek();
        if (c == '"' || c == '\'') {
            n->kind = Node::String;
            n->s = string();
        } else if (c == '[') {
            at++;
            n->kind = Node::Array;
            for (;;) {
                skipBlank(true);
                if (peek() == ']') {
                    at++;
                    break;
                }
                n->items.push_back(value());
                skipBlank(true);
                if (peek() == ',')
                    at++;
                else if (peek() != ']')
                    fail("expected ',' or ']' in array");
            }
        } else if (text.compare(at, 4, "true") == 0) {
            n->kind = Node::Bool, n->b = true, at += 4;
        } else if (text.compare(at, 5, "false") == 0) {
            n->kind = Node::Bool, n->b = false, at += 5;
        } else {
            std::string tok;
            while (std::isalnum(static_cast<unsigned char>(peek())) || peek() == '+' || peek() == '-' || peek() == '.' || peek() == '_')
                tok += text[at++];
            tok.erase(std::remove(tok.begin(), tok.end(), '_'), tok.end());
            if (tok.empty())
                fail("expected a value");
            const bool isFloat = tok.find_first_of(".eE") != std::string::npos || tok.find("inf") != std::string::npos || tok.find("nan") != std::string::npos;
            char* end = nullptr;
            if (isFloat) {
                n->kind = Node::Float;
                n->f = std::strtod(tok.c_str(), &end);
            } else {
                n->kind = Node::Int;
                n->i = std::strtoll(tok.c_str(), &end, 10);
            }
            if (!end || *end)
                fail("cannot read the value '" + tok + "'");
        }
        return n;
    }
    static Node* descend(Node* table, const std::string& k, bool arrayOfTables, const Parser& p)
    {
        NodePtr& slot = table->fields[k];
        if (!slot) {
            slot = std::make_shared<Node>();
            slot->kind = arrayOfTables ? Node::Array : Node::Table;
        }
        if (slot->kind == Node::Array && !arrayOfTables) { // a path through an array of tables: its last element
            if (slot->items.empty() || slot->items.back()->kind != Node::Table)
                p.fail("'" + k + "' is not a table");
            return slot->items.back().get();
        }
        if (slot->kind != (arrayOfTables ? Node::Array : Node::Table))
            p.fail("'" + k + "' is already a value");
        return slot.get();
    }
    NodePtr document()
    {
        auto root = std::make_shared<Node>();
        Node* current = root.get();
        for (;;) {
            skipBlank(true);
            if (done())
                return root;
            if (peek() == '[') {
                at++;
                const bool aot = peek() == '[';
                if (aot)
                    at++;
                const std::vector<std::string> path = dottedKey();
                skipBlank(false);
                if (peek() != ']' || (aot && text.compare(at, 2, "]]") != 0))
                    fail("expected the end of the table header");
                at += aot ? 2 : 1;
                Node* t = root.get();
                for (size_t k = 0; k + 1 < path.size(); k++)
                    t = descend(t, path[k], false, *this);
                if (aot) {
                    Node* arr = descend(t, path.back(), true, *this);
                    arr->items.push_back(std::make_shared<Node>());
                    current = arr->items.back().get();
                } else {
                    current = descend(t, path.back(), false, *this);
                }
            } else {
                const std::vector<std::string> path = dottedKey();
                skipBlank(false);
                if (peek() != '=')
                    fail("expected '='");
                at++;
                Node* t = current;
                for (size_t k = 0; k + 1 < path.size(); k++)
                    t = descend(t, path[k], false, *this);
                t->fields[path.back()] = value();
            }
            skipBlank(false);
            if (!done() && peek() != '\n')
                fail("unexpected text after the value");
        }
    }
};

// tomlArrayToVec3, src/config.cpp:160-178: the first three numbers, zeros for the rest
vec3 toVec3(const Node* array, vec3 fallback)
{
    if (!array || array->kind != Node::Array)
        return fallback;
    vec3 out(0.f, 0.f, 0.f);
    int i = 0;
    for (const NodePtr& e : array->items)
        if (e->number() && i < 3)
            out[i++] = float(e->asDouble());
    return out;
}
const Node* item(const Node* array, size_t k)
{
    return (array && array->kind == Node::Array && k < array->items.size()) ? array->items[k].get() : nullptr;
}
bool flag(const Node* table, const char* key, bool fallback = false)
{
    const Node* n = table ? table->get(key) : nullptr;
    return (n && n->kind == Node::Bool) ? n->b : fallback;
}

} // namespace

std::string serialize(SceneType sceneType)
{
    switch (sceneType) {
    case SingleTriangle: return "single_triangle";
    case Cube: return "cube";
    case CubeTextured: return "cube_textured";
    case CornellBox: return "cornell_box";
    case CornellBoxParallelogramLight: return "cornell_box_parallelogram_light";
    case Monkey: return "monkey";
    case Teapot: return "teapot";
    case Dragon: return "dragon";
    case Spheres: return "spheres";
    case Custom: return "custom";
    case TextureDebug: return "texture_debug";
    case TransparencyDebug: return "transparency_debug";
    }
    return "unknown";
}

bool deserialize(const std::string& name, SceneType& out)
{
    std::string l;
    for (const char c : name)
        l += char(std::tolower(static_cast<unsigned char>(c)));
    static const std::pair<const char*, SceneType> names[] = { { "single_triangle", SingleTriangle }, { "singletriangle", SingleTriangle }, { "single-triangle", SingleTriangle },
        { "cube", Cube }, { "cube-textured", CubeTextured }, { "cube_textured", CubeTextured }, { "cubetextured", CubeTextured }, { "cornell_box", CornellBox },
        { "cornellbox", CornellBox }, { "cornell-box", CornellBox }, { "cornell_box_parallelogram_light", CornellBoxParallelogramLight },
        { "cornellboxparallelogramlight", CornellBoxParallelogramLight }, { "cornell-box-parallelogram-light", CornellBoxParallelogramLight }, { "monkey", Monkey },
        { "teapot", Teapot }, { "dragon", Dragon }, { "spheres", Spheres }, { "custom", Custom }, { "texture_debug", TextureDebug }, { "transparency_debug", TransparencyDebug } };
    for (const auto& [n, t] : names)
        if (l == n) {
            out = t;
            return true;
        }
    return false;
}

Config readConfigFile(const std::string& configPath, const std::string& defaultDataPath, bool extendedFeatures)
{
    std::ifstream in(configPath);
    if (!in)
        throw std::runtime_error("cannot open " + configPath);
    std::stringstream buffer;
    buffer << in.rdbuf();
    const std::string text = buffer.str();
    NodePtr root;
    try {
        root = Parser(text).document();
    } catch (const std::runtime_error& e) {
        throw std::runtime_error("failed parsing " + configPath + ": " + e.what());
    }
    Config config;
    const Node* n = root->get("command_line_rendering");
    config.cliRenderingEnabled = (n && n->kind == Node::Bool) ? n->b : true;
    {
        // tomlArrayToIVec2, src/config.cpp:181-198: the first two integers; a missing array gives
        // (0, 0), not the (800, 800) the caller's value_or names, because the helper always returns a value
        int out[2] = { 0, 0 }, i = 0;
        if ((n = root->get("window_size")) && n->kind == Node::Array)
            for (const NodePtr& e : n->items)
                if (e->kind == Node::Int && i < 2)
                    out[i++] = int(e->i);
        config.windowSize = ivec2 { out[0], out[1] };
    }
    n = root->get("data_path");
    config.dataPath = (n && n->kind == Node::String && n->s != "default") ? n->s : defaultDataPath;
    n = root->get("scene");
    if (n && n->kind == Node::Int) {
        config.scene = SceneType(n->i);
    } else {
        const std::string name = (n && n->kind == Node::String) ? n->s : "none";
        SceneType type;
        if (deserialize(name, type)) {
            config.scene = type;
        } else {
            std::string path = config.dataPath;
            if (!path.empty() && path.back() != '/')
                path += '/';
            path += name;
            if (!std::ifstream(path))
                throw std::runtime_error("scene file " + path + " does not exist");
            config.scene = path;
        }
    }
    n = root->get("output_dir");
    std::string outputDir = (n && n->kind == Node::String) ? n->s : "";
    if (outputDir.empty()) {
        outputDir = ".";
    } else if (const char* home = std::getenv("HOME")) {
        if (outputDir[0] == '~')
            outputDir.replace(0, 1, home);
        if (outputDir.rfind("$HOME", 0) == 0)
            outputDir.replace(0, 5, home);
    }
    config.outputDir = outputDir;

    const Node* f = root->get("features");
    config.features.enableShading = flag(f, "enable_shading");
    config.features.enableRecursive = flag(f, "enable_recursive");
    config.features.enableHardShadow = flag(f, "enable_hard_shadow");
    config.features.enableNormalInterp = flag(f, "enable_normal_interp");
    config.features.enableTextureMapping = flag(f, "enable_texture_mapping");
    config.features.enableAccelStructure = flag(f, "enable_accel_structure");
    const Node* x = f ? f->get("extra") : nullptr;
    config.features.extra.enableBloomEffect = flag(x, "enable_bloom_effect");
    config.features.extra.enableMultipleRaysPerPixel = flag(x, "enable_multiple_rays_per_pixel");
    config.features.extra.enableMotionBlur = flag(x, "enable_motion_blur");
    config.features.extra.enableDepthOfField = flag(x, "enable_depth_of_field");
    config.features.extra.enableGlossyReflection = flag(x, "enable_glossy_reflection");
    config.features.extra.enableEnvironmentMapping = flag(x, "enable_environment_mapping");
    config.features.extra.enableBilinearTextureFiltering = flag(x, "enable_bilinear_texture_filtering");
    config.features.extra.enableMipmapTextureFiltering = flag(x, "enable_mipmap_texture_filtering");
    if (extendedFeatures) { // (never read by the reference: src/config.cpp:265-323)
        config.features.enableSoftShadow = flag(f, "enable_soft_shadow");
        config.features.extra.enableTransparency = flag(x, "enable_transparency");
        config.features.extra.enableBvhSahBinning = flag(x, "enable_bvh_sah_binning");
    }

    if ((n = root->get("cameras")) && n->kind == Node::Array)
        for (const NodePtr& c : n->items) {
            CameraConfig cam;
            const Node* v = c->get("field_of_view");
            cam.fieldOfView = (v && v->number()) ? float(v->asDouble()) : 50.0f;
            v = c->get("distance_from_look_at");
            cam.distanceFromLookAt = (v && v->number()) ? float(v->asDouble()) : 3.0f;
            cam.lookAt = toVec3(c->get("look_at"), vec3(0.f, 0.f, 0.f));
            cam.rotation = toVec3(c->get("rotation"), vec3(20.f, 20.f, 0.f));
            // (tomlArrayToVec3 returns zeros, not the fallback, for a missing array: src/config.cpp:160-178
            // always hands back a value, so value_or never applies)
            if (!c->get("look_at"))
                cam.lookAt = vec3(0.f, 0.f, 0.f);
            if (!c->get("rotation"))
                cam.rotation = vec3(0.f, 0.f, 0.f);
            config.cameras.push_back(cam);
        }
    if ((n = root->get("lights")) && n->kind == Node::Array)
        for (const NodePtr& l : n->items) {
            const Node* t = l->get("type");
            const std::string type = (t && t->kind == Node::String) ? t->s : "none";
            const vec3 zero(0.f, 0.f, 0.f);
            if (type == "point") {
                config.lights.emplace_back(PointLight { toVec3(l->get("position"), zero), toVec3(l->get("color"), zero) });
            } else if (type == "segment") {
                const Node *e = l->get("endpoints"), *c = l->get("colors");
                config.lights.emplace_back(SegmentLight { toVec3(item(e, 0), zero), toVec3(item(e, 1), zero), toVec3(item(c, 0), zero), toVec3(item(c, 1), zero) });
            } else if (type == "parallelogram") {
                const Node *e = l->get("edges"), *c = l->get("colors");
                config.lights.emplace_back(ParallelogramLight { toVec3(l->get("corner"), zero), toVec3(item(e, 0), zero), toVec3(item(e, 1), zero), toVec3(item(c, 0), zero),
                    toVec3(item(c, 1), zero), toVec3(item(c, 2), zero), toVec3(item(c, 3), zero) });
            } // (an unknown type is skipped, src/config.cpp:372-374)
        }
    return config;
}

Scene loadScenePrebuilt(SceneType type, const std::string& dataDir)
{
    const std::string dir = (dataDir.empty() || dataDir.back() == '/') ? dataDir : dataDir + "/";
    Scene scene;
    auto meshes = [&](const char* file, bool normalize = false) { scene.meshes = loadMesh(dir + file, normalize); };
    const vec3 white(1.f, 1.f, 1.f);
    switch (type) {
    case SingleTriangle:
        meshes("triangle.obj");
        if (!scene.meshes.empty())
            scene.meshes[0].material.kd = white;
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), white });
        break;
    case Cube:
        meshes("cube.obj");
        scene.lights.emplace_back(SegmentLight { vec3(1.5f, 0.5f, -0.6f), vec3(-1, 0.5f, -0.5f), vec3(0.9f, 0.2f, 0.1f), vec3(0.2f, 1, 0.3f) });
        break;
    case CubeTextured:
        meshes("cube-textured.obj");
        scene.lights.emplace_back(PointLight { vec3(-1.0f, 1.5f, -1.0f), white });
        break;
    case CornellBox:
        meshes("CornellBox-Mirror-Rotated.obj", true);
        scene.lights.emplace_back(PointLight { vec3(0, 0.58f, 0), white });
        break;
    case CornellBoxParallelogramLight:
        meshes("CornellBox-Mirror-Rotated.obj", true);
        scene.lights.emplace_back(ParallelogramLight { vec3(-0.2f, 0.5f, 0), vec3(0.4f, 0, 0), vec3(0.0f, 0.0f, 0.4f), vec3(1, 0, 0), vec3(0, 1, 0), vec3(0, 0, 1), vec3(0, 1, 1) });
        break;
    case Monkey:
        meshes("monkey.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), white });
        scene.lights.emplace_back(PointLight { vec3(1, -1, -1), white });
        break;
    case Teapot:
        meshes("teapot.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), white });
        break;
    case Dragon:
        meshes("dragon.obj", true);
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), white });
        break;
    case Spheres: {
        auto sphere = [](vec3 c, float r, vec3 kd) {
            Sphere s;
            s.center = c;
            s.radius = r;
            s.material.kd = kd;
            return s;
        };
        scene.spheres.push_back(sphere(vec3(3.0f, -2.0f, 10.2f), 1.0f, vec3(0.8f, 0.2f, 0.2f)));
        scene.spheres.push_back(sphere(vec3(-2.0f, 2.0f, 4.0f), 2.0f, vec3(0.6f, 0.8f, 0.2f)));
        scene.spheres.push_back(sphere(vec3(0.0f, 0.0f, 6.0f), 0.75f, vec3(0.2f, 0.2f, 0.8f)));
        scene.lights.emplace_back(PointLight { vec3(3, 0, 3), vec3(15.f, 15.f, 15.f) });
    } break;
    case TransparencyDebug:
        meshes("cg3.obj");
        scene.lights.emplace_back(PointLight { vec3(-1, 1, -1), white });
        break;
    case TextureDebug:
        break; // (empty in the reference too)
    case Custom:
        meshes("texture_debug.obj");
        scene.lights.emplace_back(PointLight { vec3(float(2.514), float(0.743), float(1.955)), white });
        break;
    }
    return scene;
}

Scene loadSceneFromFile(const std::string& path, const std::vector<std::variant<PointLight, SegmentLight, ParallelogramLight>>& lights)
{
    Scene scene;
    scene.lights = lights;
    scene.meshes = loadMesh(path);
    return scene;
}

} // namespace rt
