#include "scene_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

namespace rt {
namespace {

// ------------------------------------------------------------- text -------
bool isSpace(char c) { return c == ' ' || c == '\t'; }
bool isDigit(char c) { return c >= '0' && c <= '9'; }

void skipSpace(const char*& p)
{
    while (isSpace(*p))
        p++;
}

// A decimal number the way tinyobjloader reads one (its own routine, not strtod,
// so the rounding of the last bit is its own): the digits are accumulated in a
// double, fraction digits weighted by a table of powers of ten, and a decimal
// exponent e applied as ldexp(m * 5^e, e).  Malformed text leaves `fallback`.
double parseNumber(const char* s, const char* end, double fallback)
{
    if (s >= end)
        return fallback;
    const char* c = s;
    double mantissa = 0.0;
    int exponent = 0;
    bool negative = false, negativeExp = false, leadingDot = false;
    if (*c == '+' || *c == '-') {
        negative = *c == '-';
        c++;
        if (c != end && *c == '.')
            leadingDot = true;
    } else if (*c == '.') {
        leadingDot = true;
    } else if (!isDigit(*c)) {
        return fallback;
    }
    int read = 0;
    if (!leadingDot) {
        while (c != end && isDigit(*c)) {
            mantissa *= 10;
            mantissa += int(*c - '0');
            c++;
            read++;
        }
        if (read == 0)
            return fallback;
    }
    bool done = c == end;
    if (!done) {
        if (*c == '.') {
            static const double weight[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
            c++;
            read = 1;
            while (c != end && isDigit(*c)) {
                mantissa += int(*c - '0') * (read < 8 ? weight[read] : std::pow(10.0, -read));
                read++;
                c++;
            }
        } else if (*c != 'e' && *c != 'E') {
            done = true;
        }
    }
    if (!done && c != end && (*c == 'e' || *c == 'E')) {
        c++;
        if (c != end && (*c == '+' || *c == '-')) {
            negativeExp = *c == '-';
            c++;
        } else if (!isDigit(*c)) {
            return fallback;
        }
        read = 0;
        while (c != end && isDigit(*c)) {
            if (exponent > 2147483647 / 10)
                return fallback;
            exponent *= 10;
            exponent += int(*c - '0');
            c++;
            read++;
        }
        if (negativeExp)
            exponent = -exponent;
        if (read == 0)
            return fallback;
    }
    const double magnitude = exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa;
    return (negative ? -1 : 1) * magnitude;
}

// next blank-separated field as a float (rounded once, from the double)
float nextReal(const char*& p, double fallback = 0.0)
{
    skipSpace(p);
    const char* end = p + std::strcspn(p, " \t\r");
    const double v = parseNumber(p, end, fallback);
    p = end;
    return float(v);
}

std::string nextWord(const char*& p)
{
    skipSpace(p);
    const size_t n = std::strcspn(p, " \t\r");
    std::string s(p, n);
    p += n;
    return s;
}

bool startsWith(const char* p, const char* word)
{
    const size_t n = std::strlen(word);
    return std::strncmp(p, word, n) == 0 && isSpace(p[n]);
}

// ------------------------------------------------------------- MTL --------
struct ObjMaterial {
    std::string name;
    float diffuse[3] = { 0, 0, 0 }, specular[3] = { 0, 0, 0 };
    float shininess = 1.f, dissolve = 1.f;
    std::string diffuseTexture;
};

void readMaterials(const std::string& path, std::vector<ObjMaterial>& materials, std::map<std::string, int>& byName)
{
    std::ifstream in(path);
    if (!in)
        return; // a missing library only means "no materials", as in the reference's reader
    ObjMaterial cur;
    bool any = false, hasD = false;
    std::string line;
    auto flush = [&] {
        if (!cur.name.empty()) {
            byName.emplace(cur.name, int(materials.size()));
            materials.push_back(cur);
        }
    };
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r'))
            line.pop_back();
        while (!line.empty() && isSpace(line.back()))
            line.pop_back();
        const char* p = line.c_str();
        skipSpace(p);
        if (*p == '\0' || *p == '#')
            continue;
        if (startsWith(p, "newmtl")) {
            if (any)
                flush();
            cur = ObjMaterial();
            hasD = false;
            any = true;
            p += 7;
            cur.name = p;
        } else if (p[0] == 'K' && p[1] == 'd' && isSpace(p[2])) {
            p += 2;
            for (float& c : cur.diffuse)
                c = nextReal(p);
        } else if (p[0] == 'K' && p[1] == 's' && isSpace(p[2])) {
            p += 2;
            for (float& c : cur.specular)
                c = nextReal(p);
        } else if (p[0] == 'N' && p[1] == 's' && isSpace(p[2])) {
            p += 2;
            cur.shininess = nextReal(p);
        } else if (p[0] == 'd' && isSpace(p[1])) {
            p += 1;
            cur.dissolve = nextReal(p);
            hasD = true;
        } else if (p[0] == 'T' && p[1] == 'r' && isSpace(p[2])) {
            p += 2;
            if (!hasD) // `d` wins over `Tr` whichever comes first
                cur.dissolve = 1.0f - nextReal(p);
        } else if (startsWith(p, "map_Kd")) {
            p += 7;
            skipSpace(p);
            if (*p == '-')
                throw RtIoError(path + ": texture options on map_Kd are not supported");
            cur.diffuseTexture = p; // Kd stays whatever the file says (0 if it says nothing)
        }
    }
    if (any)
        flush();
}

// ------------------------------------------------------------- OBJ --------
struct Corner {
    int v = -1, vt = -1, vn = -1;
};
struct Shape {
    std::vector<Corner> corners; // three per triangle
    std::vector<int> materialIds; // one per triangle
};

bool resolveIndex(int raw, int count, int& out)
{
    if (raw > 0)
        out = raw - 1;
    else if (raw < 0)
        out = count + raw; // relative to the end of what has been read so far
    else
        return false;
    return true;
}

// one corner of an `f` line: i, i/j, i//k or i/j/k
bool readCorner(const char*& p, int nv, int nvn, int nvt, Corner& c)
{
    c = Corner();
    if (!resolveIndex(std::atoi(p), nv, c.v))
        return false;
    p += std::strcspn(p, "/ \t\r");
    if (*p != '/')
        return true;
    p++;
    if (*p == '/') {
        p++;
        if (!resolveIndex(std::atoi(p), nvn, c.vn))
            return false;
        p += std::strcspn(p, "/ \t\r");
        return true;
    }
    if (!resolveIndex(std::atoi(p), nvt, c.vt))
        return false;
    p += std::strcspn(p, "/ \t\r");
    if (*p != '/')
        return true;
    p++;
    if (!resolveIndex(std::atoi(p), nvn, c.vn))
        return false;
    p += std::strcspn(p, "/ \t\r");
    return true;
}

struct ObjFile {
    std::vector<float> v, vn, vt;
    std::vector<Shape> shapes;
    std::vector<ObjMaterial> materials;
};

void readObj(const std::string& file, const std::string& baseDir, ObjFile& obj)
{
    std::ifstream in(file);
    if (!in)
        throw RtIoError("cannot open " + file);
    std::map<std::string, int> materialByName;
    struct Face {
        std::vector<Corner> corners;
    };
    std::vector<Face> pending; // faces since the last flush, all of material `material`
    int material = -1;
    Shape shape;

    // faces -> triangles of the current shape (quads split along the shorter diagonal)
    auto flushFaces = [&] {
        for (const Face& f : pending) {
            const size_t n = f.corners.size();
            if (n < 3)
                continue;
            auto emit = [&](int a, int b, int c) {
                shape.corners.push_back(f.corners[size_t(a)]);
                shape.corners.push_back(f.corners[size_t(b)]);
                shape.corners.push_back(f.corners[size_t(c)]);
                shape.materialIds.push_back(material);
            };
            if (n == 3) {
                emit(0, 1, 2);
            } else if (n == 4) {
                bool ok = true;
                for (const Corner& c : f.corners)
                    ok = ok && c.v >= 0 && size_t(3 * c.v + 2) < obj.v.size();
                if (!ok)
                    continue;
                const float* p0 = &obj.v[size_t(3 * f.corners[0].v)];
                const float* p1 = &obj.v[size_t(3 * f.corners[1].v)];
                const float* p2 = &obj.v[size_t(3 * f.corners[2].v)];
                const float* p3 = &obj.v[size_t(3 * f.corners[3].v)];
                const float ax = p2[0] - p0[0], ay = p2[1] - p0[1], az = p2[2] - p0[2];
                const float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
                const float d02 = ax * ax + ay * ay + az * az, d13 = bx * bx + by * by + bz * bz;
                if (d02 < d13) {
                    emit(0, 1, 2);
                    emit(0, 2, 3);
                } else {
                    emit(0, 1, 3);
                    emit(1, 2, 3);
                }
            } else {
                throw RtIoError(file + ": faces with more than four corners are not supported");
            }
        }
        pending.clear();
    };
    auto closeShape = [&] {
        flushFaces();
        if (!shape.corners.empty())
            obj.shapes.push_back(shape);
        shape = Shape();
    };

    std::string line;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r'))
            line.pop_back();
        const char* p = line.c_str();
        skipSpace(p);
        if (*p == '\0' || *p == '#')
            continue;
        if (p[0] == 'v' && isSpace(p[1])) {
            p += 2;
            for (int i = 0; i < 3; i++)
                obj.v.push_back(nextReal(p));
        } else if (p[0] == 'v' && p[1] == 'n' && isSpace(p[2])) {
            p += 3;
            for (int i = 0; i < 3; i++)
                obj.vn.push_back(nextReal(p));
        } else if (p[0] == 'v' && p[1] == 't' && isSpace(p[2])) {
            p += 3;
            for (int i = 0; i < 2; i++)
                obj.vt.push_back(nextReal(p));
        } else if (p[0] == 'f' && isSpace(p[1])) {
            p += 2;
            skipSpace(p);
            Face f;
            while (*p != '\0' && *p != '\r' && *p != '\n') {
                Corner c;
                if (!readCorner(p, int(obj.v.size() / 3), int(obj.vn.size() / 3), int(obj.vt.size() / 2), c))
                    throw RtIoError(file + ": bad face index in \"" + line + "\"");
                f.corners.push_back(c);
                p += std::strspn(p, " \t\r");
            }
            pending.push_back(std::move(f));
        } else if (startsWith(p, "usemtl")) {
            p += 6;
            const std::string name = nextWord(p);
            const auto it = materialByName.find(name);
            const int id = it == materialByName.end() ? -1 : it->second;
            if (id != material) {
                flushFaces(); // same shape, faces carry their material
                material = id;
            }
        } else if (startsWith(p, "mtllib")) {
            p += 7;
            std::istringstream names(p);
            std::string one;
            while (names >> one) {
                const size_t before = obj.materials.size();
                readMaterials(baseDir.empty() ? one : baseDir + "/" + one, obj.materials, materialByName);
                if (obj.materials.size() != before)
                    break; // the first library that loads is the one used
            }
        } else if ((p[0] == 'g' || p[0] == 'o') && isSpace(p[1])) {
            closeShape();
        }
    }
    closeShape();
}

bool sameVertex(const Vertex& a, const Vertex& b)
{
    return a.position.x == b.position.x && a.position.y == b.position.y && a.position.z == b.position.z && a.normal.x == b.normal.x
        && a.normal.y == b.normal.y && a.normal.z == b.normal.z && a.texCoord.x == b.texCoord.x && a.texCoord.y == b.texCoord.y;
}
struct VertexHash {
    size_t operator()(const Vertex& v) const
    {
        const float f[8] = { v.position.x, v.position.y, v.position.z, v.normal.x, v.normal.y, v.normal.z, v.texCoord.x, v.texCoord.y };
        size_t h = 1469598103934665603ull;
        for (float x : f) {
            if (x == 0.f)
                x = 0.f; // +0 and -0 compare equal, so they must hash alike
            uint32_t bits;
            std::memcpy(&bits, &x, 4);
            h = (h ^ bits) * 1099511628211ull;
        }
        return h;
    }
};
struct VertexEq {
    bool operator()(const Vertex& a, const Vertex& b) const { return sameVertex(a, b); }
};

std::string directoryOf(const std::string& file)
{
    const size_t slash = file.find_last_of('/');
    return slash == std::string::npos ? std::string() : file.substr(0, slash);
}

// centerAndScaleToUnitMesh, framework/src/mesh.cpp:151-176
void centerAndScale(std::vector<Mesh>& meshes)
{
    vec3 sum(0.0f);
    size_t count = 0;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices) {
            sum = sum + v.position;
            count++;
        }
    const float n = float(count);
    const vec3 center(sum.x / n, sum.y / n, sum.z / n);
    float maxD = 0.0f;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices)
            maxD = std::max(length(v.position - center), maxD);
    for (Mesh& m : meshes)
        for (Vertex& v : m.vertices) {
            const vec3 d = v.position - center;
            v.position = vec3(d.x / maxD, d.y / maxD, d.z / maxD);
        }
}

} // namespace

std::vector<Mesh> loadMesh(const std::string& file, bool centerAndNormalize)
{
    const std::string baseDir = directoryOf(file);
    ObjFile obj;
    readObj(file, baseDir, obj);
    std::map<std::string, std::shared_ptr<Image>> textures;

    std::vector<Mesh> out;
    for (const Shape& shape : obj.shapes) {
        const size_t nTri = shape.materialIds.size();
        // One mesh per run of equal material ids (framework/src/mesh.cpp:76-86).  The
        // reference's loop looks at the LAST triangle only to close the final run, so
        // a material change exactly there stays in the previous mesh; kept as is.
        size_t start = 0;
        int prev = shape.materialIds[0];
        for (size_t end = 0; end < nTri; ++end) {
            if (end == nTri - 1)
                ++end;
            else if (shape.materialIds[end] == prev)
                continue;
            else
                prev = shape.materialIds[end];

            Mesh mesh;
            std::unordered_map<Vertex, uint32_t, VertexHash, VertexEq> seen;
            for (size_t i = start * 3; i != end * 3; i += 3) {
                auto position = [&](const Corner& c) {
                    if (c.v < 0 || size_t(3 * c.v + 2) >= obj.v.size())
                        throw RtIoError(file + ": face refers to a vertex that does not exist");
                    return vec3(obj.v[size_t(3 * c.v)], obj.v[size_t(3 * c.v + 1)], obj.v[size_t(3 * c.v + 2)]);
                };
                const vec3 v0 = position(shape.corners[i]), v1 = position(shape.corners[i + 1]), v2 = position(shape.corners[i + 2]);
                const vec3 geometricNormal = normalize(cross(v1 - v0, v2 - v0));
                uint32_t tri[3];
                for (int j = 0; j < 3; j++) {
                    const Corner& c = shape.corners[i + size_t(j)];
                    Vertex vertex;
                    vertex.position = position(c);
                    if (c.vn != -1 && !obj.vn.empty()) {
                        if (c.vn < 0 || size_t(3 * c.vn + 2) >= obj.vn.size())
                            throw RtIoError(file + ": face refers to a normal that does not exist");
                        vertex.normal = vec3(obj.vn[size_t(3 * c.vn)], obj.vn[size_t(3 * c.vn + 1)], obj.vn[size_t(3 * c.vn + 2)]);
                    } else {
                        vertex.normal = geometricNormal;
                    }
                    if (c.vt != -1 && !obj.vt.empty()) {
                        if (c.vt < 0 || size_t(2 * c.vt + 1) >= obj.vt.size())
                            throw RtIoError(file + ": face refers to a texture coordinate that does not exist");
                        vertex.texCoord = vec2 { obj.vt[size_t(2 * c.vt)], obj.vt[size_t(2 * c.vt + 1)] };
                    }
                    const auto it = seen.find(vertex);
                    if (it != seen.end()) {
                        tri[j] = it->second;
                    } else {
                        tri[j] = uint32_t(mesh.vertices.size());
                        seen.emplace(vertex, tri[j]);
                        mesh.vertices.push_back(vertex);
                    }
                }
                mesh.triangles.push_back(uvec3 { tri[0], tri[1], tri[2] });
            }

            const int id = shape.materialIds[start];
            if (id == -1) {
                mesh.material.kd = vec3(1.0f);
                mesh.material.ks = vec3(0.0f);
                mesh.material.shininess = 1.0f;
            } else {
                const ObjMaterial& m = obj.materials[size_t(id)];
                mesh.material.kd = vec3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
                if (!m.diffuseTexture.empty()) {
                    auto& slot = textures[m.diffuseTexture];
                    if (!slot)
                        slot = std::make_shared<Image>(loadImage(baseDir.empty() ? m.diffuseTexture : baseDir + "/" + m.diffuseTexture));
                    mesh.material.kdTexture = slot;
                }
                mesh.material.ks = vec3(m.specular[0], m.specular[1], m.specular[2]);
                mesh.material.shininess = m.shininess;
                mesh.material.transparency = m.dissolve;
            }
            out.push_back(std::move(mesh));
            start = end;
        }
    }
    if (centerAndNormalize && !out.empty())
        centerAndScale(out);
    return out;
}

// ------------------------------------------------------------- PNG --------
namespace {

uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

} // namespace

Image decodeJpeg(const std::vector<unsigned char>& bytes, const std::string& file); // jpeg.cpp

namespace {

uint32_t le32(const unsigned char* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }

// Uncompressed Windows bitmaps of 8 (palette), 24 and 32 bits per pixel, rows bottom-up or
// top-down; 32-bit files with channel masks other than B, G, R in the low three bytes and
// every compressed kind are rejected.
Image decodeBmp(const std::vector<unsigned char>& d, const std::string& file)
{
    if (d.size() < 54)
        throw RtIoError(file + ": truncated BMP header");
    const uint32_t offset = le32(&d[10]), headerSize = le32(&d[14]);
    if (headerSize < 40)
        throw RtIoError(file + ": BMP core headers are not supported");
    const int32_t w = int32_t(le32(&d[18])), hRaw = int32_t(le32(&d[22]));
    const int bpp = d[28] | (d[29] << 8);
    const uint32_t compression = le32(&d[30]);
    const bool masks = compression == 3 && bpp == 32 && d.size() >= 66 && le32(&d[54]) == 0x00ff0000u && le32(&d[58]) == 0x0000ff00u && le32(&d[62]) == 0x000000ffu;
    if (compression != 0 && !masks)
        throw RtIoError(file + ": compressed BMPs are not supported");
    if (bpp != 8 && bpp != 24 && bpp != 32)
        throw RtIoError(file + ": only BMPs of 8, 24 or 32 bits per pixel are supported");
    const bool topDown = hRaw < 0;
    const int64_t h = topDown ? -int64_t(hRaw) : hRaw;
    if (w <= 0 || h <= 0 || w > 65536 || h > 65536)
        throw RtIoError(file + ": bad BMP size");
    const size_t stride = (size_t(w) * size_t(bpp) / 8 + 3) & ~size_t(3);
    // (both are the file's own words: the header must lie inside the file and the pixels behind it)
    if (14 + size_t(headerSize) > d.size() || size_t(offset) < 14 + size_t(headerSize))
        throw RtIoError(file + ": bad BMP header size or pixel offset");
    if (size_t(offset) + stride * size_t(h) > d.size())
        throw RtIoError(file + ": truncated BMP pixel data");
    const unsigned char* palette = nullptr;
    uint32_t colours = le32(&d[46]);
    if (bpp == 8) {
        if (colours == 0)
            colours = 256;
        if (14 + size_t(headerSize) + 4 * size_t(colours) > d.size() || colours > 256)
            throw RtIoError(file + ": truncated BMP palette");
        palette = d.data() + 14 + size_t(headerSize);
    }
    Image img;
    img.width = w;
    img.height = int(h);
    img.pixels.reserve(size_t(w) * size_t(h));
    for (int64_t y = 0; y < h; y++) {
        const unsigned char* row = &d[offset + stride * size_t(topDown ? y : h - 1 - y)];
        for (int x = 0; x < w; x++) {
            const unsigned char* px = bpp == 8 ? (row[x] < colours ? palette + 4 * size_t(row[x]) : nullptr) : row + size_t(x) * size_t(bpp / 8);
            if (!px)
                throw RtIoError(file + ": BMP palette index out of range");
            img.pixels.emplace_back(px[2] / 255.0f, px[1] / 255.0f, px[0] / 255.0f);
        }
    }
    return img;
}

} // namespace

Image loadImage(const std::string& file)
{
    std::ifstream in(file, std::ios::binary);
    if (!in)
        throw RtIoError("texture file " + file + " does not exist");
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const unsigned char magic[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (data.size() >= 3 && data[0] == 0xff && data[1] == 0xd8 && data[2] == 0xff)
        return decodeJpeg(data, file);
    if (data.size() >= 2 && data[0] == 'B' && data[1] == 'M')
        return decodeBmp(data, file);
    if (data.size() < 8 || std::memcmp(data.data(), magic, 8) != 0)
        throw RtIoError(file + ": only PNG, JPEG and BMP textures are supported");
    uint32_t width = 0, height = 0;
    int depth = 0, colorType = 0, interlace = 0;
    std::vector<unsigned char> compressed, palette;
    for (size_t pos = 8; pos + 12 <= data.size();) {
        const uint32_t len = be32(&data[pos]);
        const char* type = reinterpret_cast<const char*>(&data[pos + 4]);
        const unsigned char* body = &data[pos + 8];
        if (pos + 12 + size_t(len) > data.size())
            throw RtIoError(file + ": truncated PNG chunk");
        if (std::memcmp(type, "IHDR", 4) == 0 && len >= 13) {
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            colorType = body[9];
            interlace = body[12];
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            palette.assign(body, body + len);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            compressed.insert(compressed.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (width == 0 || height == 0 || width > 65536 || height > 65536)
        throw RtIoError(file + ": bad PNG header");
    if (interlace != 0)
        throw RtIoError(file + ": interlaced PNGs are not supported");
    int channels;
    switch (colorType) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: throw RtIoError(file + ": unknown PNG colour type");
    }
    const bool subByte = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || depth == 16 || (subByte && (colorType == 0 || colorType == 3))))
        throw RtIoError(file + ": unsupported PNG bit depth");
    const size_t stride = (size_t(width) * size_t(channels) * size_t(depth) + 7) / 8; // bytes per row
    const size_t unit = std::max<size_t>(1, size_t(channels) * size_t(depth) / 8);     // bytes per pixel as the filters see it
    std::vector<unsigned char> raw((stride + 1) * size_t(height));
    uLongf rawSize = uLongf(raw.size());
    if (uncompress(raw.data(), &rawSize, compressed.data(), uLong(compressed.size())) != Z_OK || rawSize != raw.size())
        throw RtIoError(file + ": PNG data does not decompress to the size its header announces");
    // undo the per-row filters
    std::vector<unsigned char> rows(stride * size_t(height));
    for (size_t y = 0; y < height; y++) {
        const unsigned char* src = &raw[y * (stride + 1)];
        unsigned char* row = &rows[y * stride];
        const unsigned char* up = y ? row - stride : nullptr;
        const int filter = src[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= unit ? row[i - unit] : 0;
            const int b = up ? up[i] : 0;
            const int c = (up && i >= unit) ? up[i - unit] : 0;
            int v = src[i + 1];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: throw RtIoError(file + ": unknown PNG row filter");
            }
            row[i] = static_cast<unsigned char>(v);
        }
    }
    // one byte per sample: 16-bit samples keep their high byte, sub-byte samples are
    // unpacked most significant bits first (grey levels stretched to 0..255)
    std::vector<unsigned char> pixels(size_t(width) * size_t(height) * size_t(channels));
    for (size_t y = 0; y < height; y++) {
        const unsigned char* row = &rows[y * stride];
        unsigned char* dst = &pixels[y * size_t(width) * size_t(channels)];
        const size_t samples = size_t(width) * size_t(channels);
        if (depth == 8) {
            std::memcpy(dst, row, samples);
        } else if (depth == 16) {
            for (size_t i = 0; i < samples; i++)
                dst[i] = row[2 * i];
        } else {
            const int perByte = 8 / depth, mask = (1 << depth) - 1;
            const int stretch = colorType == 0 ? 255 / mask : 1;
            for (size_t i = 0; i < samples; i++) {
                const int shift = 8 - depth * (int(i % size_t(perByte)) + 1);
                dst[i] = static_cast<unsigned char>(((row[i / size_t(perByte)] >> shift) & mask) * stretch);
            }
        }
    }
    Image img;
    img.width = int(width);
    img.height = int(height);
    img.pixels.reserve(size_t(width) * size_t(height));
    for (size_t i = 0; i < size_t(width) * size_t(height); i++) {
        const unsigned char* px = &pixels[i * size_t(channels)];
        unsigned char r, g, b;
        if (colorType == 3) {
            if (size_t(px[0]) * 3 + 2 >= palette.size())
                throw RtIoError(file + ": PNG palette index out of range");
            r = palette[size_t(px[0]) * 3], g = palette[size_t(px[0]) * 3 + 1], b = palette[size_t(px[0]) * 3 + 2];
        } else if (channels <= 2) {
            r = g = b = px[0]; // grey is replicated; alpha is dropped
        } else {
            r = px[0], g = px[1], b = px[2];
        }
        img.pixels.emplace_back(r / 255.0f, g / 255.0f, b / 255.0f);
    }
    return img;
}

} // namespace rt
