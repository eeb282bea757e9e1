// .fray scene-description parser and flattening (host side, no GPU).
//
// Behaviour follows the reference's DefaultSceneParser (src/scene.cpp:403-570) so that the
// shipped scene files load unchanged:
//   * `//` and `#` start a comment anywhere in a line; a line whose first two raw characters
//     are `/ *` opens a block comment that ends at a line whose first two raw characters are
//     `* /` (scene.cpp:431-448);
//   * `Class name {` opens a block, `Class {` a singleton whose name becomes "{" (scene.cpp:461-487);
//   * inside a block a line is `<prop> <rest of line>`, quotes around the rest are stripped
//     (scene.cpp:519-529); only the FIRST line with a given name is seen by a typed getter
//     (scene.cpp:136-146); scale/rotate/translate lines apply in file order (scene.cpp:297-320);
//   * blocks are interpreted in the order settings, camera, environment, lights, geometries,
//     textures, shaders, nodes (scene.cpp:532-538), so forward references by name work;
//   * nodes without a shader are dropped from the render list (scene.cpp:563-568).
#include "host_scene.h"

#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <sys/stat.h>

namespace frayhost {

namespace {

const double kPI = 3.141592653589793238;   // constants.h:31

struct ParseError : std::runtime_error {
    explicit ParseError(const std::string& m) : std::runtime_error(m) {}
};

[[noreturn]] void fail(int line, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw ParseError("line " + std::to_string(line) + ": " + buf);
}

bool path_exists(std::string p)
{
    if (!p.empty() && p.back() == '/') p.pop_back();   // util.cpp:57-66
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

std::vector<std::string> split_ws(const std::string& s)   // util.cpp:69-82
{
    std::vector<std::string> out;
    size_t i = 0, n = s.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)s[i])) i++;
        if (i >= n) break;
        size_t j = i;
        while (j < n && !isspace((unsigned char)s[j])) j++;
        out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}

std::string trim(const std::string& s)
{
    size_t b = 0, e = s.size();
    while (e > 0 && isspace((unsigned char)s[e - 1])) e--;
    while (b < e && isspace((unsigned char)s[b])) b++;
    return s.substr(b, e - b);
}

// ---- 3x3 matrix helpers, row-vector convention (matrix.h:53-60, matrix.cpp:29-115) ---------
struct Mat3 { double a[3][3]; };

Mat3 mat_diag(double d)
{
    Mat3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = (i == j) ? d : 0.0;
    return r;
}
Mat3 mat_mul(const Mat3& x, const Mat3& y)
{
    Mat3 c = mat_diag(0.0);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) c.a[i][j] += x.a[i][k] * y.a[k][j];
    return c;
}
// sincos() by name: the reference's g++ build merges sin(angle) and cos(angle) into that call, and its sine differs from sin()'s in the last
// place for one angle in 700
Mat3 rot_x(double ang) { double S, C; sincos(ang, &S, &C); Mat3 r = mat_diag(1.0); r.a[1][1] = C; r.a[2][1] = S; r.a[1][2] = -S; r.a[2][2] = C; return r; }
Mat3 rot_y(double ang) { double S, C; sincos(ang, &S, &C); Mat3 r = mat_diag(1.0); r.a[0][0] = C; r.a[2][0] = -S; r.a[0][2] = S; r.a[2][2] = C; return r; }
Mat3 rot_z(double ang) { double S, C; sincos(ang, &S, &C); Mat3 r = mat_diag(1.0); r.a[0][0] = C; r.a[1][0] = S; r.a[0][1] = -S; r.a[1][1] = C; return r; }
double to_rad(double deg) { return deg / 180.0 * kPI; }   // util.h:37

double mat_det(const Mat3& m)   // matrix.cpp:75-83, same term order
{
    const double (*a)[3] = m.a;
    return a[0][0] * a[1][1] * a[2][2] - a[0][0] * a[1][2] * a[2][1] - a[0][1] * a[1][0] * a[2][2]
         + a[0][1] * a[1][2] * a[2][0] + a[0][2] * a[1][0] * a[2][1] - a[0][2] * a[1][1] * a[2][0];
}
double mat_cofactor(const Mat3& m, int ii, int jj)   // matrix.cpp:85-96
{
    int rows[2], rc = 0, cols[2], cc = 0;
    for (int i = 0; i < 3; i++) if (i != ii) rows[rc++] = i;
    for (int j = 0; j < 3; j++) if (j != jj) cols[cc++] = j;
    double t = m.a[rows[0]][cols[0]] * m.a[rows[1]][cols[1]] - m.a[rows[1]][cols[0]] * m.a[rows[0]][cols[1]];
    if ((ii + jj) % 2) t = -t;
    return t;
}
Mat3 mat_inverse(const Mat3& m)   // matrix.cpp:98-108
{
    double D = mat_det(m);
    if (fabs(D) < 1e-12) return m;
    double rD = 1.0 / D;
    Mat3 r;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[i][j] = rD * mat_cofactor(m, j, i);
    return r;
}

struct Xform {   // Transform, matrix.h:72-98
    double off[3] = {0, 0, 0};
    Mat3 m = mat_diag(1.0), inv = mat_diag(1.0);
    void scale(double x, double y, double z)
    {
        Mat3 t = mat_diag(0.0);
        t.a[0][0] = x; t.a[1][1] = y; t.a[2][2] = z;
        m = mat_mul(m, t);
        inv = mat_inverse(m);
    }
    void rotate(double yaw, double pitch, double roll)
    {
        m = mat_mul(mat_mul(mat_mul(m, rot_z(to_rad(roll))), rot_x(to_rad(pitch))), rot_y(to_rad(yaw)));
        inv = mat_inverse(m);
    }
    void translate(double x, double y, double z) { off[0] += x; off[1] += y; off[2] += z; }
    void store(frayhip_transform& T) const
    {
        for (int i = 0; i < 3; i++) T.offset[i] = off[i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T.m[i * 3 + j] = m.a[i][j]; T.invM[i * 3 + j] = inv.a[i][j]; }
    }
    void point(const double p[3], double out[3]) const   // transformPoint: p*m + offset
    {
        for (int j = 0; j < 3; j++) out[j] = (p[0] * m.a[0][j] + p[1] * m.a[1][j] + p[2] * m.a[2][j]) + off[j];
    }
};

// ---- block model ------------------------------------------------------------------------------
struct PropLine { int line; std::string name, value; bool used = false; };

enum Kind { K_SETTINGS, K_CAMERA, K_ENV, K_LIGHT, K_GEOM, K_TEX, K_SHADER, K_NODE };

struct Block {
    std::string cls, name;
    Kind kind;
    int begin = 0, end = 0;
    int index = -1;               // position in the per-kind scene array
    std::vector<PropLine> props;

    PropLine* find(const char* n)
    {
        for (auto& p : props) if (p.name == n) { p.used = true; return &p; }
        return nullptr;
    }
    bool getInt(const char* n, int32_t& v, int lo = INT32_MIN, int hi = INT32_MAX)
    {
        PropLine* p = find(n); if (!p) return false;
        int x;
        if (sscanf(p->value.c_str(), "%d", &x) != 1) fail(p->line, "Invalid integer");
        if (x < lo || x > hi) fail(p->line, "Value outside the allowed bounds (%d .. %d)", lo, hi);
        v = x; return true;
    }
    bool getBool(const char* n, int32_t& v)
    {
        PropLine* p = find(n); if (!p) return false;
        v = !(p->value == "off" || p->value == "false" || p->value == "0");
        return true;
    }
    bool getFloat(const char* n, float& v, float lo = -1e17f, float hi = 1e17f)
    {
        PropLine* p = find(n); if (!p) return false;
        float x;
        if (sscanf(p->value.c_str(), "%f", &x) != 1) fail(p->line, "Invalid float");
        if (x < lo || x > hi) fail(p->line, "Value outside the allowed bounds (%f .. %f)", lo, hi);
        v = x; return true;
    }
    bool getDouble(const char* n, double& v, double lo = -1e120, double hi = 1e120)
    {
        PropLine* p = find(n); if (!p) return false;
        double x;
        if (sscanf(p->value.c_str(), "%lf", &x) != 1) fail(p->line, "Invalid double");
        if (x < lo || x > hi) fail(p->line, "Value outside the allowed bounds (%f .. %f)", lo, hi);
        v = x; return true;
    }
    static void unbrace(std::string& s) { for (auto& c : s) if (c == ',' || c == '(' || c == ')') c = ' '; }
    bool getColor(const char* n, float v[3])
    {
        PropLine* p = find(n); if (!p) return false;
        unbrace(p->value);
        float c[3];
        if (sscanf(p->value.c_str(), "%f%f%f", &c[0], &c[1], &c[2]) != 3) fail(p->line, "Invalid color");
        for (int i = 0; i < 3; i++) {
            if (c[i] < -1e17f || c[i] > 1e17f) fail(p->line, "Color value outside the allowed bounds");
            v[i] = c[i];
        }
        return true;
    }
    bool getVector(const char* n, double v[3])
    {
        PropLine* p = find(n); if (!p) return false;
        unbrace(p->value);
        double c[3];
        if (sscanf(p->value.c_str(), "%lf%lf%lf", &c[0], &c[1], &c[2]) != 3) fail(p->line, "Invalid vector");
        v[0] = c[0]; v[1] = c[1]; v[2] = c[2];
        return true;
    }
    void require(const char* n)
    {
        if (!find(n)) fail(end, "Required property `%s' not defined", n);
    }
    void getTransform(Xform& T)
    {
        for (auto& p : props) {
            int which = p.name == "scale" ? 0 : p.name == "rotate" ? 1 : p.name == "translate" ? 2 : -1;
            if (which < 0) continue;
            p.used = true;
            unbrace(p.value);
            double x, y, z;
            if (sscanf(p.value.c_str(), "%lf%lf%lf", &x, &y, &z) != 3) fail(p.line, "Expected three double values");
            if (which == 0) T.scale(x, y, z);
            else if (which == 1) T.rotate(x, y, z);
            else T.translate(x, y, z);
        }
    }
};

struct ClassInfo { const char* cls; Kind kind; int sub; };
// The class factory of scene.cpp:821-848.
const ClassInfo kClasses[] = {
    {"GlobalSettings", K_SETTINGS, 0}, {"Camera", K_CAMERA, 0}, {"CubemapEnvironment", K_ENV, 0},
    {"PointLight", K_LIGHT, FRAYHIP_LIGHT_POINT}, {"RectLight", K_LIGHT, FRAYHIP_LIGHT_RECT},
    {"Plane", K_GEOM, FRAYHIP_GEOM_PLANE}, {"Sphere", K_GEOM, FRAYHIP_GEOM_SPHERE}, {"Cube", K_GEOM, FRAYHIP_GEOM_CUBE},
    {"Mesh", K_GEOM, FRAYHIP_GEOM_MESH},
    {"CsgPlus", K_GEOM, 100 + FRAYHIP_CSG_PLUS}, {"CsgAnd", K_GEOM, 100 + FRAYHIP_CSG_AND}, {"CsgMinus", K_GEOM, 100 + FRAYHIP_CSG_MINUS},
    {"CheckerTexture", K_TEX, FRAYHIP_TEX_CHECKER}, {"BitmapTexture", K_TEX, FRAYHIP_TEX_BITMAP},
    {"BumpTexture", K_TEX, FRAYHIP_TEX_BUMP}, {"Fresnel", K_TEX, FRAYHIP_TEX_FRESNEL},
    {"Const", K_SHADER, FRAYHIP_SHADER_CONST}, {"Lambert", K_SHADER, FRAYHIP_SHADER_LAMBERT},
    {"Phong", K_SHADER, FRAYHIP_SHADER_PHONG}, {"Refl", K_SHADER, FRAYHIP_SHADER_REFL},
    {"Refr", K_SHADER, FRAYHIP_SHADER_REFR}, {"Layered", K_SHADER, FRAYHIP_SHADER_LAYERED},
    {"Node", K_NODE, 0},
};
const ClassInfo* find_class(const std::string& c)
{
    for (auto& k : kClasses) if (c == k.cls) return &k;
    return nullptr;
}

// randfloat(a,b) / randint(a,b) macro substitution, scene.cpp:609-653.  The generator is the
// reference's table entry 0 after initRandom(42) (random_generator.cpp:91-108, scene.cpp:405).
struct MacroRng {
    std::mt19937 gen;
    MacroRng()
    {
        unsigned seed = 42u ^ 0xbf14ef80u;
        gen.seed(seed);
        std::uniform_int_distribution<unsigned> raw;
        for (int i = 0; i < 1223; i++) raw(gen);
        raw(gen);                                               // seed of table entry 1
        std::uniform_int_distribution<int> warm(0, 1222); warm(gen);   // its warm-up count
    }
    float randfloat() { std::uniform_real_distribution<float> d; return d(gen); }
    int randint(int a, int b) { std::uniform_int_distribution<int> d(a, b); return d(gen); }
};

void substitute_macros(int lineNo, std::string& s, MacroRng& rng)
{
    for (int pass = 0; pass < 2; pass++) {
        const char* key = pass == 0 ? "randfloat" : "randint";
        size_t p;
        while ((p = s.find(key)) != std::string::npos) {
            size_t i = s.find('(', p), j = (i == std::string::npos) ? i : s.find(')', i);
            if (i == std::string::npos || j == std::string::npos) fail(lineNo, "%s in inexpected format", key);
            std::string args = s.substr(i + 1, j - i - 1);
            char text[40];
            if (pass == 0) {
                float f1, f2;
                if (sscanf(args.c_str(), "%f,%f", &f1, &f2) != 2) fail(lineNo, "bad randfloat format");
                if (f1 > f2) fail(lineNo, "bad randfloat format (min > max)");
                float res = rng.randfloat() * (f2 - f1) + f1;
                snprintf(text, sizeof text, "%.5f", res);
            } else {
                int a, b;
                if (sscanf(args.c_str(), "%d,%d", &a, &b) != 2) fail(lineNo, "bad randint format");
                if (a > b) fail(lineNo, "bad randint format (min > max)");
                snprintf(text, sizeof text, "%d", rng.randint(a, b));
            }
            std::string rep(j - p + 1, ' ');
            size_t l = strlen(text);
            if (l >= j - p) fail(lineNo, "random macro result does not fit");
            rep.replace(0, l, text);
            s.replace(p, j - p + 1, rep);
        }
    }
}

struct Loader {
    std::string rootDir;
    std::vector<Block> blocks;
    HostScene* hs = nullptr;
    // per-kind block lists in declaration order = the reference's scene arrays
    std::vector<int> geomBlocks, texBlocks, shaderBlocks, nodeBlocks, lightBlocks;
    int cameraBlock = -1, envBlock = -1;

    bool resolve(std::string& p) const   // scene.cpp:724-735
    {
        std::string full = rootDir + p;
        if (path_exists(full)) { p = full; return true; }
        return false;
    }
    int lookup(const std::vector<int>& list, const std::string& name) const
    {
        for (size_t i = 0; i < list.size(); i++) if (blocks[list[i]].name == name) return (int)i;
        return -1;
    }

    void read(const char* path)
    {
        FILE* f = fopen(path, "rt");
        if (!f) throw ParseError(std::string("Cannot open scene file `") + path + "'");
        std::string p = path;
        size_t slash = p.find_last_of("/\\");
        rootDir = slash == std::string::npos ? "" : p.substr(0, slash + 1);
        MacroRng rng;
        char raw[1024];
        int lineNo = 0;
        bool inComment = false;
        int cur = -1;
        try {
            while (fgets(raw, sizeof raw, f)) {
                lineNo++;
                if (inComment) {
                    if (raw[0] == '*' && raw[1] == '/') inComment = false;
                    continue;
                }
                std::string line = raw;
                size_t c1 = line.find("//"), c2 = line.find('#');
                size_t cut = c1 < c2 ? c1 : c2;
                if (cut != std::string::npos) line.erase(cut);
                line = trim(line);
                if (line.empty()) continue;
                if (line[0] == '/' && line.size() > 1 && line[1] == '*') { inComment = true; continue; }
                substitute_macros(lineNo, line, rng);
                std::vector<std::string> tok = split_ws(line);
                if (cur < 0) {
                    if (tok.size() == 1) fail(lineNo, "Unexpected token `%s'", tok[0].c_str());
                    if (tok.size() > 3) fail(lineNo, "Unexpected content");
                    if (tok.back() != "{") fail(lineNo, "An object definition should end with a `{'");
                    const ClassInfo* ci = find_class(tok[0]);
                    if (!ci) fail(lineNo, "Unknown object class `%s'", tok[0].c_str());
                    Block b;
                    b.cls = tok[0];
                    b.name = tok[1].substr(0, 63);          // SceneElement::name is char[64]
                    b.kind = ci->kind;
                    b.begin = lineNo;
                    blocks.push_back(b);
                    cur = (int)blocks.size() - 1;
                    switch (ci->kind) {
                        case K_GEOM: blocks[cur].index = (int)geomBlocks.size(); geomBlocks.push_back(cur); break;
                        case K_TEX: blocks[cur].index = (int)texBlocks.size(); texBlocks.push_back(cur); break;
                        case K_SHADER: blocks[cur].index = (int)shaderBlocks.size(); shaderBlocks.push_back(cur); break;
                        case K_NODE: blocks[cur].index = (int)nodeBlocks.size(); nodeBlocks.push_back(cur); break;
                        case K_LIGHT: blocks[cur].index = (int)lightBlocks.size(); lightBlocks.push_back(cur); break;
                        case K_CAMERA: cameraBlock = cur; break;
                        case K_ENV: envBlock = cur; break;
                        default: break;
                    }
                } else if (tok.size() == 1) {
                    if (tok[0] != "}") fail(lineNo, "Unexpected token in object definition: `%s'", tok[0].c_str());
                    blocks[cur].end = lineNo;
                    cur = -1;
                } else {
                    size_t i = tok[0].size();
                    while (i < line.size() && isspace((unsigned char)line[i])) i++;
                    size_t l = line.size() - 1;
                    std::string value;
                    if (i < l && line[i] == '"' && line[l] == '"') value = line.substr(i + 1, l - i - 1);
                    else value = line.substr(i);
                    if (value.size() > 255) value.resize(255);
                    PropLine pl;
                    pl.line = lineNo;
                    pl.name = tok[0].substr(0, 127);
                    pl.value = value;
                    blocks[cur].props.push_back(pl);
                }
            }
        } catch (...) { fclose(f); throw; }
        fclose(f);
        if (cur >= 0) throw ParseError("Unfinished object definition at EOF!");
    }

    // ---- per-class interpretation (each element's fillProperties) --------------------------------
    void fillSettings(Block& b)   // scene.cpp:799-814
    {
        frayhip_settings& s = hs->desc.settings;
        b.getInt("frameWidth", s.frameWidth);
        b.getInt("frameHeight", s.frameHeight);
        b.getColor("ambientLight", s.ambientLight);
        b.getInt("maxTraceDepth", s.maxTraceDepth);
        b.getBool("dbg", s.dbg);
        b.getBool("wantAA", s.wantAA);
        b.getFloat("saturation", s.saturation, 0, 1);
        b.getBool("wantPrepass", s.wantPrepass);
        b.getBool("gi", s.gi);
        b.getInt("pathsPerPixel", s.numPaths, 1);
        b.getInt("numThreads", s.numThreads);
        b.getBool("interactive", s.interactive);
        b.getBool("fullscreen", s.fullscreen);
    }
    void fillCamera(Block& b)   // camera.h:57-75
    {
        frayhip_camera& c = hs->desc.camera;
        if (!b.getVector("position", c.pos)) b.require("position");
        b.getDouble("aspectRatio", c.aspectRatio, 1e-6);
        b.getDouble("fov", c.fov, 0.0001, 179);
        b.getDouble("yaw", c.yaw);
        b.getDouble("pitch", c.pitch, -90, 90);
        b.getDouble("roll", c.roll);
        b.getBool("dof", c.dof);
        b.getDouble("fNumber", c.fNumber, 0);
        b.getInt("numSamples", c.numDOFSamples, 1);
        b.getDouble("focalPlaneDist", c.focalPlaneDist, 0.1);
        b.getBool("autofocus", c.autofocus);
        b.getDouble("stereoSeparation", c.stereoSeparation, 0.0);
        b.getColor("leftMask", c.leftMask);
        b.getColor("rightMask", c.rightMask);
    }
    int64_t addImage(const Image& img)
    {
        int64_t off = (int64_t)hs->texels.size();
        hs->texels.insert(hs->texels.end(), img.rgb.begin(), img.rgb.end());
        return off;
    }
    bool loadImage(const std::string& file, Image& img, std::string& err)   // Bitmap::loadImage, bitmap.cpp:286-291
    {
        size_t dot = file.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : file.substr(dot + 1);
        for (auto& c : ext) c = (char)toupper((unsigned char)c);
        if (ext == "BMP") return load_bmp(file.c_str(), img, err);
        if (ext == "EXR") return load_exr(file.c_str(), img, err);
        err = "unknown image extension";
        return false;
    }
    void fillEnvironment(Block& b)   // environment.h:66-76, environment.cpp:31-52
    {
        frayhip_environment& e = hs->desc.environment;
        e.present = 1;
        e.loaded = 0;
        PropLine* p = b.find("folder");
        if (!p) { b.require("folder"); return; }
        std::string folder = p->value;
        if (!resolve(folder)) fail(p->line, "Required file not found (%s)", p->value.c_str());
        const char* prefixes[2] = {"neg", "pos"};
        const char* axes[3] = {"x", "y", "z"};
        const char* suffixes[2] = {".bmp", ".exr"};
        int n = 0;
        for (int pi = 0; pi < 2; pi++)
            for (int ax = 0; ax < 3; ax++, n++) {
                Image img;
                std::string err;
                for (int si = 0; si < 2 && !img.ok(); si++) {
                    std::string fn = folder + "/" + prefixes[pi] + axes[ax] + suffixes[si];
                    if (path_exists(fn)) loadImage(fn, img, err);
                }
                if (!img.ok()) {
                    hs->warnings.push_back("CubemapEnvironment: could not load maps from `" + folder + "' (" + err + ")");
                    return;
                }
                e.width[n] = img.w; e.height[n] = img.h;
                e.texel_offset[n] = addImage(img);
            }
        e.loaded = 1;
    }
    void fillLight(Block& b, int sub)   // lights.h:52-56,60-65,82-88; lights.cpp:37-46
    {
        frayhip_light L{};
        L.kind = sub;
        L.color[0] = L.color[1] = L.color[2] = 1; L.power = 1;
        L.xSubd = L.ySubd = 1;
        Xform T;
        b.getColor("color", L.color);
        b.getFloat("power", L.power);
        if (sub == FRAYHIP_LIGHT_POINT) {
            b.getVector("pos", L.pos);
        } else {
            b.getInt("xSubd", L.xSubd, 1);
            b.getInt("ySubd", L.ySubd, 1);
            b.getTransform(T);
            const double o[3] = {0, 0, 0}, pa[3] = {-0.5, 0, -0.5}, pb[3] = {0.5, 0, -0.5}, pc[3] = {0.5, 0, 0.5};
            double a[3], bb[3], c[3];
            T.point(o, L.center); T.point(pa, a); T.point(pb, bb); T.point(pc, c);
            double d1[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, d2[3] = {bb[0] - c[0], bb[1] - c[1], bb[2] - c[2]};
            float width = (float)sqrt(d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2]);
            float height = (float)sqrt(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]);
            L.area = width * height;     // float product widened to double, as in the reference
        }
        T.store(L.T);
        hs->lights.push_back(L);
    }
    void fillGeometry(Block& b, int sub)
    {
        frayhip_geom_ref ref{};
        if (sub == FRAYHIP_GEOM_PLANE) {   // geometry.h:60-66
            frayhip_plane p{128, 0};
            b.getDouble("y", p.height);
            b.getDouble("limit", p.limit);
            ref = {FRAYHIP_GEOM_PLANE, (int32_t)hs->planes.size()};
            hs->planes.push_back(p);
        } else if (sub == FRAYHIP_GEOM_SPHERE) {   // geometry.h:82-87
            frayhip_sphere s{{0, 0, 0}, 1};
            b.getVector("O", s.O);
            b.getDouble("R", s.R);
            ref = {FRAYHIP_GEOM_SPHERE, (int32_t)hs->spheres.size()};
            hs->spheres.push_back(s);
        } else if (sub == FRAYHIP_GEOM_CUBE) {   // geometry.h:106-111
            frayhip_cube c{{0, 0, 0}, 1};
            b.getVector("O", c.O);
            b.getDouble("halfSide", c.halfSide);
            ref = {FRAYHIP_GEOM_CUBE, (int32_t)hs->cubes.size()};
            hs->cubes.push_back(c);
        } else if (sub == FRAYHIP_GEOM_MESH) {   // mesh.h:78-92, mesh.cpp:67-94
            MeshData md;
            PropLine* p = b.find("file");
            if (!p) b.require("file");
            std::string file = p->value;
            if (!resolve(file)) fail(p->line, "Required file not found (%s)", p->value.c_str());
            if (!load_obj(file.c_str(), md)) fail(b.end, "Could not parse OBJ file!");
            int32_t v;
            if (b.getBool("faceted", v)) md.faceted = v;
            if (b.getBool("backfaceCulling", v)) md.backfaceCulling = v;
            if (b.getBool("useKDTree", v)) md.useKD = v;
            build_kd(md);                                   // Mesh::beginRender
            if (md.normals.empty()) md.faceted = true;
            ref = {FRAYHIP_GEOM_MESH, (int32_t)hs->meshData.size()};
            hs->meshData.push_back(std::move(md));
        } else {   // CSG, geometry.h:117-131
            frayhip_csg c{};
            c.op = sub - 100;
            b.require("left");
            b.require("right");
            PropLine* l = b.find("left"); PropLine* r = b.find("right");
            c.left = lookup(geomBlocks, l->value);
            c.right = lookup(geomBlocks, r->value);
            if (c.left < 0) fail(l->line, "Geometry not defined");
            if (c.right < 0) fail(r->line, "Geometry not defined");
            ref = {FRAYHIP_GEOM_CSG, (int32_t)hs->csgs.size()};
            hs->csgs.push_back(c);
        }
        hs->geoms[b.index] = ref;
    }
    void fillTexture(Block& b, int sub)   // shading.h:42-110, 211-222
    {
        frayhip_texture t{};
        t.kind = sub;
        t.width = t.height = -1;
        t.scaling = 1; t.bumpIntensity = 10.0f; t.ior = 1;
        t.color1[0] = t.color1[1] = t.color1[2] = 0.7f;
        t.color2[0] = t.color2[1] = t.color2[2] = 0.2f;
        auto bitmapProp = [&](Image& img) {
            PropLine* p = b.find("file");
            if (!p) { b.require("file"); return; }
            std::string file = p->value, err;
            if (!resolve(file)) fail(p->line, "Required file not found (%s)", p->value.c_str());
            if (!loadImage(file, img, err)) fail(p->line, "cannot load image %s: %s", file.c_str(), err.c_str());
        };
        if (sub == FRAYHIP_TEX_CHECKER) {
            b.getColor("color1", t.color1);
            b.getColor("color2", t.color2);
            b.getDouble("scaling", t.scaling);
        } else if (sub == FRAYHIP_TEX_BITMAP) {
            b.getDouble("scaling", t.scaling);
            t.scaling = 1 / t.scaling;
            Image img;
            bitmapProp(img);
            t.width = img.w; t.height = img.h;
            t.texel_offset = addImage(img);
        } else if (sub == FRAYHIP_TEX_BUMP) {
            b.getDouble("strength", t.bumpIntensity);
            b.getDouble("scaling", t.scaling);
            Image img;
            bitmapProp(img);
            // BumpTexture::beginRender -> Bitmap::differentiate (bitmap.cpp:300-315)
            Image d;
            d.w = img.w; d.h = img.h;
            d.rgb.assign(img.rgb.size(), 0.0f);
            auto inten = [&](int x, int y) {
                const float* p = &img.rgb[(size_t)(x + y * img.w) * 3];
                return (p[0] + p[1] + p[2]) / 3;
            };
            for (int y = 0; y < img.h; y++)
                for (int x = 0; x < img.w; x++) {
                    float dx = inten(x, y) - inten((x + 1) % img.w, y);
                    float dy = inten(x, y) - inten(x, (y + 1) % img.h);
                    float* o = &d.rgb[(size_t)(x + y * img.w) * 3];
                    o[0] = dx; o[1] = dy; o[2] = 0;
                }
            t.width = d.w; t.height = d.h;
            t.texel_offset = addImage(d);
        } else {
            b.getDouble("ior", t.ior, 1e-6, 10);
        }
        hs->textures[b.index] = t;
    }
    void fillShader(Block& b, int sub)   // shading.h:138-255, shading.cpp:313-355
    {
        frayhip_shader s{};
        s.kind = sub;
        s.texture = -1;
        s.color[0] = s.color[1] = s.color[2] = 1;
        if (sub == FRAYHIP_SHADER_CONST) { s.color[1] = s.color[2] = 0; }
        s.specularColor[0] = s.specularColor[1] = s.specularColor[2] = 0.75f;
        s.mult[0] = s.mult[1] = s.mult[2] = 1;
        s.exponent = 10.0f; s.specularMultiplier = 0.25f;
        s.glossiness = 1.0; s.numSamples = 10; s.ior = 1;
        auto texProp = [&]() {
            PropLine* p = b.find("texture");
            if (!p) return;
            s.texture = lookup(texBlocks, p->value);
            if (s.texture < 0) fail(p->line, "Texture not defined");
        };
        if (sub == FRAYHIP_SHADER_LAMBERT) {
            b.getColor("color", s.color);
            texProp();
        } else if (sub == FRAYHIP_SHADER_PHONG) {
            b.getColor("color", s.color);
            texProp();
            b.getDouble("specularExponent", s.exponent);
            b.getDouble("specularMultiplier", s.specularMultiplier);
            b.getColor("specularColor", s.specularColor);
        } else if (sub == FRAYHIP_SHADER_REFL) {
            double m = 1;
            b.getDouble("multiplier", m);
            s.mult[0] = s.mult[1] = s.mult[2] = (float)m;
            b.getDouble("glossiness", s.glossiness, 0, 1);
            b.getInt("numSamples", s.numSamples, 1);
        } else if (sub == FRAYHIP_SHADER_REFR) {
            double m = 1;
            b.getDouble("multiplier", m);
            s.mult[0] = s.mult[1] = s.mult[2] = (float)m;
            b.getDouble("ior", s.ior, 1e-6, 10);
        } else if (sub == FRAYHIP_SHADER_LAYERED) {
            s.layer_begin = (int32_t)hs->layers.size();
            for (auto& p : b.props) {
                p.used = true;
                if (p.name != "layer") continue;
                // "layer <shader> (r, g, b) [texture]"
                std::vector<std::string> tok = split_ws(p.value);
                if (tok.size() < 2) fail(p.line, "Expected a line like `layer <shader>, <color>[, <texture>]'");
                auto strip = [](std::string s) { std::string o; for (char c : s) if (!isspace((unsigned char)c) && c != ',') o += c; return o; };
                std::string shaderName = strip(tok[0]);
                std::string rest = trim(p.value.substr(p.value.find(tok[0]) + tok[0].size()));
                std::string texName;
                if (!rest.empty() && rest.back() != ')') {
                    texName = strip(tok.back());
                    rest = rest.substr(0, rest.rfind(tok.back()));
                }
                if (texName == "NULL") texName.clear();
                frayhip_layer L{};
                L.shader = lookup(shaderBlocks, shaderName);
                L.texture = texName.empty() ? -1 : lookup(texBlocks, texName);
                if (L.shader < 0 || (!texName.empty() && L.texture < 0))
                    fail(p.line, "Expected a line like `layer <shader>, <color>[, <texture>]'");
                Block::unbrace(rest);
                double x, y, z;
                if (sscanf(rest.c_str(), "%lf%lf%lf", &x, &y, &z) != 3) fail(p.line, "Expected three double values");
                L.opacity[0] = (float)x; L.opacity[1] = (float)y; L.opacity[2] = (float)z;
                if (s.layer_count < 32) { hs->layers.push_back(L); s.layer_count++; }   // Layer layers[32]
            }
        }
        // Reflection::beginFrame (shading.h:197-201)
        s.deflectionScaling = pow(10.0, 2 - 4 * s.glossiness);
        hs->shaders[b.index] = s;
    }
    struct NodeTmp { frayhip_node n; bool hasShader; };
    NodeTmp fillNode(Block& b)   // geometry.h:168-175
    {
        NodeTmp r{};
        r.n.geom = -1; r.n.shader = -1; r.n.bump_tex = -1;
        if (PropLine* p = b.find("geometry")) {
            r.n.geom = lookup(geomBlocks, p->value);
            if (r.n.geom < 0) fail(p->line, "Geometry not defined");
        }
        if (PropLine* p = b.find("shader")) {
            r.n.shader = lookup(shaderBlocks, p->value);
            if (r.n.shader < 0) fail(p->line, "Shader not defined");
        }
        Xform T;
        b.getTransform(T);
        T.store(r.n.T);
        if (PropLine* p = b.find("bump")) {
            r.n.bump_tex = lookup(texBlocks, p->value);
            if (r.n.bump_tex < 0) fail(p->line, "Texture not defined");
        }
        r.hasShader = r.n.shader >= 0;
        return r;
    }

    void interpret()
    {
        // defaults: GlobalSettings ctor scene.cpp:783-797, Camera members camera.h:43-55
        frayhip_settings& s = hs->desc.settings;
        s.frameWidth = 800; s.frameHeight = 600; s.wantAA = 1; s.dbg = 0; s.maxTraceDepth = 4;
        s.ambientLight[0] = s.ambientLight[1] = s.ambientLight[2] = 0;
        s.saturation = 1; s.wantPrepass = 1; s.gi = 0; s.numPaths = 10; s.numThreads = 0;
        s.interactive = s.fullscreen = 0;
        frayhip_camera& c = hs->desc.camera;
        c.pos[0] = c.pos[1] = c.pos[2] = 0; c.yaw = c.pitch = c.roll = 0; c.fov = 90; c.aspectRatio = 1.3333;
        c.focalPlaneDist = 5; c.fNumber = 2; c.dof = 0; c.autofocus = 1; c.numDOFSamples = 32; c.stereoSeparation = 0;
        c.leftMask[0] = 1; c.leftMask[1] = c.leftMask[2] = 0;
        c.rightMask[0] = 0; c.rightMask[1] = c.rightMask[2] = 1;

        hs->geoms.resize(geomBlocks.size());
        hs->textures.resize(texBlocks.size());
        hs->shaders.resize(shaderBlocks.size());
        const Kind order[] = {K_SETTINGS, K_CAMERA, K_ENV, K_LIGHT, K_GEOM, K_TEX, K_SHADER, K_NODE};
        for (Kind k : order)
            for (auto& b : blocks) {
                if (b.kind != k) continue;
                const ClassInfo* ci = find_class(b.cls);
                switch (k) {
                    case K_SETTINGS: fillSettings(b); break;
                    case K_CAMERA: if (&b == &blocks[cameraBlock]) fillCamera(b); break;
                    case K_ENV: if (&b == &blocks[envBlock]) fillEnvironment(b); break;
                    case K_LIGHT: fillLight(b, ci->sub); break;
                    case K_GEOM: fillGeometry(b, ci->sub); break;
                    case K_TEX: fillTexture(b, ci->sub); break;
                    case K_SHADER: fillShader(b, ci->sub); break;
                    case K_NODE: {
                        NodeTmp n = fillNode(b);
                        if (n.hasShader) {
                            if (n.n.geom < 0) fail(b.end, "Node `%s' has a shader but no geometry", b.name.c_str());
                            hs->nodes.push_back(n.n);
                        }
                        break;
                    }
                }
                for (auto& p : b.props)
                    if (!p.used)
                        hs->warnings.push_back("line " + std::to_string(p.line) + ": the property `" + p.name + "' isn't recognized");
            }
        if (cameraBlock < 0) throw ParseError("scene has no Camera");
    }
};

}  // namespace

void HostScene::finalize()
{
    meshes.resize(meshData.size());
    for (size_t i = 0; i < meshData.size(); i++) {
        MeshData& d = meshData[i];
        frayhip_mesh& m = meshes[i];
        m = frayhip_mesh{};
        m.n_vertices = (int32_t)(d.vertices.size() / 3);
        m.n_normals = (int32_t)(d.normals.size() / 3);
        m.n_uvs = (int32_t)(d.uvs.size() / 3);
        m.n_triangles = (int32_t)d.triangles.size();
        m.n_kdnodes = (int32_t)d.kdnodes.size();
        m.n_trirefs = (int32_t)d.trirefs.size();
        m.faceted = d.faceted; m.backfaceCulling = d.backfaceCulling; m.has_kd = !d.kdnodes.empty();
        for (int k = 0; k < 3; k++) { m.bbox_min[k] = d.bbox_min[k]; m.bbox_max[k] = d.bbox_max[k]; }
        m.vertices = d.vertices.data(); m.normals = d.normals.data(); m.uvs = d.uvs.data();
        m.triangles = d.triangles.data(); m.kdnodes = d.kdnodes.data(); m.trirefs = d.trirefs.data();
        m.kd_max_depth = d.maxDepth; m.kd_depth_sum = d.depthSum;
    }
    desc.abi_version = FRAYHIP_ABI_VERSION;
    desc.n_nodes = (int32_t)nodes.size(); desc.nodes = nodes.data();
    desc.n_geoms = (int32_t)geoms.size(); desc.geoms = geoms.data();
    desc.n_planes = (int32_t)planes.size(); desc.planes = planes.data();
    desc.n_spheres = (int32_t)spheres.size(); desc.spheres = spheres.data();
    desc.n_cubes = (int32_t)cubes.size(); desc.cubes = cubes.data();
    desc.n_csgs = (int32_t)csgs.size(); desc.csgs = csgs.data();
    desc.n_meshes = (int32_t)meshes.size(); desc.meshes = meshes.data();
    desc.n_shaders = (int32_t)shaders.size(); desc.shaders = shaders.data();
    desc.n_layers = (int32_t)layers.size(); desc.layers = layers.data();
    desc.n_textures = (int32_t)textures.size(); desc.textures = textures.data();
    desc.n_lights = (int32_t)lights.size(); desc.lights = lights.data();
    desc.n_texels = (int64_t)texels.size(); desc.texels = texels.data();
}

HostScene* parse_scene_file(const char* path, std::string& err)
{
    HostScene* hs = new HostScene();
    try {
        Loader L;
        L.hs = hs;
        L.read(path);
        L.interpret();
        hs->finalize();
        return hs;
    } catch (const std::exception& e) {
        err = std::string(path) + ": " + e.what();
        delete hs;
        return nullptr;
    }
}

}  // namespace frayhost
