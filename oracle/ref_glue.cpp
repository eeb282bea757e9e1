// oracle/_ref glue -- TEST INFRASTRUCTURE, built only where /root/reference is mounted.
//
// libfray_ref.so = the reference translation units that compile here exactly as they are
// (camera, environment, geometry, heightfield, lights, matrix, mesh, scene, shading, triangle,
// util: compiled in place from /root/reference/src by oracle/Makefile.ref) + this file.
//
// Five reference files cannot be built in this image because they include SDL 1.2 or OpenEXR
// headers (main.cpp, sdl.cpp, cxxptl-sdl.cpp, random_generator.cpp, bitmap.cpp); no stand-in
// headers were written for them.  The handful of symbols the compiled files import from those
// five are defined HERE, in our own code, against the reference's own class declarations:
//   frameWidth/frameHeight            (sdl.cpp)      -> values set by the harness
//   Random::*, getRandomGen           (random_generator.cpp) -> the same libstdc++ calls; one
//                                      generator plays the per-thread table entry
//   Bitmap::*                         (bitmap.cpp)   -> own BMP reader; EXR faces only as pre-decoded texel files (Bitmap::loadEXR below)
//   visible, raytrace, hemisphereSample (+ pathtrace, the frame loop)   (main.cpp) -> restated
// So what this library pins is everything the compiled files own: the .fray parser, transforms,
// OBJ loading, the KD build and walk, every geometry / light / camera / shader / texture method.
// The integrator loops and the RNG plumbing are restatements here exactly as they are in the
// oracle; they are NOT reference object code.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <random>
#include <vector>

#include "bitmap.h"
#include "camera.h"
#include "color.h"
#include "environment.h"
#include "geometry.h"
#include "lights.h"
#include "main.h"
#include "mesh.h"
#include "random_generator.h"
#include "scene.h"
#include "shading.h"

// ------------------------------------------------------------------ sdl.cpp:77-88
static int g_w = 0, g_h = 0;
int frameWidth(void) { return g_w; }
int frameHeight(void) { return g_h; }

// ------------------------------------------------------------------ random_generator.cpp:41-80,110-131
Random::Random(unsigned s) { generator.seed(s); }
void Random::seed(unsigned s) { generator.seed(s); }
unsigned Random::_next(void) { std::uniform_int_distribution<unsigned> g; return g(generator); }
int Random::randint(int a, int b) { std::uniform_int_distribution<int> g(a, b); return g(generator); }
float Random::randfloat(void) { std::uniform_real_distribution<float> g; return g(generator); }
double Random::randdouble(void) { std::uniform_real_distribution<double> g; return g(generator); }
double Random::gaussian(double mean, double sigma) { std::normal_distribution<double> g(mean, sigma); return g(generator); }
void Random::unitDiscSample(double& x, double& y)
{
    double angle = randdouble() * 2 * PI;
    double rad = sqrt(randdouble());
    x = sin(angle) * rad;
    y = cos(angle) * rad;
}
static Random g_table(1);   // stands for rg_table[thread]
static Random g_local(1);   // stands for the worker's copy `rnd` (main.cpp:333)
Random& getRandomGen(void) { return g_table; }
Random& getRandomGen(int) { return g_table; }

// ------------------------------------------------------------------ bitmap.cpp (BMP part only)
Bitmap::Bitmap() { width = height = -1; data = NULL; }
Bitmap::~Bitmap() { freeMem(); }
void Bitmap::freeMem(void) { delete[] data; data = NULL; width = height = -1; }
int Bitmap::getWidth(void) const { return width; }
int Bitmap::getHeight(void) const { return height; }
bool Bitmap::isOK(void) const { return data != NULL; }
void Bitmap::generateEmptyImage(int w, int h)
{
    freeMem();
    if (w <= 0 || h <= 0) return;
    width = w; height = h;
    data = new Color[w * h];
    memset(data, 0, sizeof(Color) * w * h);
}
Color Bitmap::getPixel(int x, int y) const
{
    if (!data || x < 0 || x >= width || y < 0 || y >= height) return Color(0.0f, 0.0f, 0.0f);
    return data[x + y * width];
}
void Bitmap::setPixel(int x, int y, const Color& c)
{
    if (!data || x < 0 || x >= width || y < 0 || y >= height) return;
    data[x + y * width] = c;
}
void Bitmap::differentiate()
{
    std::vector<Color> out(width * height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float here = getPixel(x, y).intensity();
            float dx = here - getPixel((x + 1) % width, y).intensity();
            float dy = here - getPixel(x, (y + 1) % height).intensity();
            out[x + y * width] = Color(dx, dy, 0);
        }
    for (int i = 0; i < width * height; i++) data[i] = out[i];
}
bool Bitmap::loadBMP(const char* filename)
{
    freeMem();
    FILE* fp = fopen(filename, "rb");
    if (!fp) return false;
    unsigned char h[54];
    bool ok = fread(h, 1, 54, fp) == 54 && h[0] == 'B' && h[1] == 'M';
    auto i32 = [&](int o) { return (int)(h[o] | (h[o + 1] << 8) | (h[o + 2] << 16) | ((unsigned)h[o + 3] << 24)); };
    int off = i32(10), w = i32(18), hh = i32(22), bpp = h[28] | (h[29] << 8), colors = i32(46);
    ok = ok && (bpp == 8 || bpp == 24 || bpp == 32) && w > 0 && hh > 0;
    Color pal[256];
    int npal = 0;
    if (ok && bpp == 8) {
        npal = colors ? colors : 256;
        for (int i = 0; i < npal && ok; i++) {
            unsigned char q[4];
            ok = fread(q, 1, 4, fp) == 4;
            pal[i] = Color((unsigned)(q[0] | (q[1] << 8) | (q[2] << 16) | ((unsigned)q[3] << 24)));
        }
    }
    if (ok) {
        fseek(fp, off - (54 + npal * 4), SEEK_CUR);
        int k = bpp / 8, row = (w * k + 3) / 4 * 4;
        std::vector<unsigned char> buf(row);
        generateEmptyImage(w, hh);
        for (int j = hh - 1; j >= 0 && ok; j--) {
            ok = fread(buf.data(), 1, row, fp) > 0;
            for (int i = 0; i < w; i++)
                setPixel(i, j, bpp > 8 ? Color(buf[i * k + 2] / 255.0f, buf[i * k + 1] / 255.0f, buf[i * k] / 255.0f) : pal[buf[i * k]]);
        }
    }
    fclose(fp);
    if (!ok) freeMem();
    return ok;
}
bool Bitmap::saveBMP(const char*) { return false; }
// OpenEXR is absent, so bitmap.cpp:238-264 cannot be built.  When FRAY_REF_FACES_DIR is set, an .exr name is served
// from <dir>/<basename>.f32 instead: {int32 w, int32 h, w*h*3 float32 RGB, rows top to bottom} -- the texels
// oracle/make_golden.py dumped from this project's own EXR decoder.  That does NOT pin the decoder; it lets the
// reference's CubemapEnvironment::loadMaps / getEnvironment object code (environment.cpp:31-98) run on those texels.
bool Bitmap::loadEXR(const char* filename)
{
    freeMem();
    const char* dir = getenv("FRAY_REF_FACES_DIR");
    if (!dir) return false;
    const char* base = strrchr(filename, '/');
    std::string path = std::string(dir) + "/" + (base ? base + 1 : filename) + ".f32";
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    int wh[2] = {0, 0};
    bool ok = fread(wh, 4, 2, fp) == 2 && wh[0] > 0 && wh[1] > 0 && wh[0] <= 8192 && wh[1] <= 8192;
    if (ok) {
        generateEmptyImage(wh[0], wh[1]);
        std::vector<float> row((size_t)wh[0] * 3);
        for (int y = 0; y < wh[1] && ok; y++) {
            ok = fread(row.data(), 4, row.size(), fp) == row.size();
            for (int x = 0; x < wh[0] && ok; x++) data[x + y * width] = Color(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
        }
    }
    fclose(fp);
    if (!ok) freeMem();
    return ok;
}
bool Bitmap::saveEXR(const char*) { return false; }
bool Bitmap::loadImage(const char* fn)
{
    size_t l = strlen(fn);
    if (l > 4 && (!strcmp(fn + l - 4, ".bmp") || !strcmp(fn + l - 4, ".BMP"))) return loadBMP(fn);
    if (l > 4 && (!strcmp(fn + l - 4, ".exr") || !strcmp(fn + l - 4, ".EXR"))) return loadEXR(fn);   // bitmap.cpp:286-298
    return false;
}
bool Bitmap::saveImage(const char*) { return false; }

// ------------------------------------------------------------------ main.cpp:64-285, restated
namespace {
struct Nearest { Node* node = nullptr; Light* light = nullptr; int nodeIdx = -1, lightIdx = -1; IntersectionInfo info; };

Nearest nearestHit(const Ray& ray)   // the node loop, then the light loop (strict <)
{
    Nearest best;
    best.info.dist = 1e99;
    for (size_t i = 0; i < scene.nodes.size(); i++) {
        IntersectionInfo ii;
        if (scene.nodes[i]->intersect(ray, ii) && ii.dist < best.info.dist) { best.info = ii; best.node = scene.nodes[i]; best.nodeIdx = (int)i; }
    }
    for (size_t i = 0; i < scene.lights.size(); i++) {
        IntersectionInfo ii;
        if (scene.lights[i]->intersect(ray, ii) && ii.dist < best.info.dist) { best.info = ii; best.light = scene.lights[i]; best.lightIdx = (int)i; }
    }
    return best;
}
void bumpIfAny(Node& n, IntersectionInfo& info)
{
    if (!n.bump) return;
    if (void* itf = n.bump->getInterface(BumpMapperInterface::ID)) static_cast<BumpMapperInterface*>(itf)->modifyNormal(info);
}
Color sampleOneLight(const Ray& ray, const IntersectionInfo& info, const Color& weight, Shader* sh)   // main.cpp:118-169
{
    if (scene.lights.empty()) return Color(0, 0, 0);
    Light* L = scene.lights[g_local.randint(0, (int)scene.lights.size() - 1)];
    double omega = L->solidAngle(info);
    if (omega == 0) return Color(0, 0, 0);
    int k = g_local.randint(0, L->getNumSamples() - 1);
    Vector p;
    Color ignored;
    L->getNthSample(k, info.ip, p, ignored);
    if (!visible(info.ip + info.norm * 1e-6, p)) return Color(0, 0, 0);
    Color Le = L->getColor();
    Vector w = p - info.ip;
    w.normalize();
    Color f = sh->eval(info, ray.dir, w);
    if (f.intensity() == 0) return Color(0, 0, 0);
    float pArea = 1.0f / omega;
    float pPick = 1.0f / scene.lights.size();
    float prob = pArea * pPick;
    return Le * weight * f / prob;
}
}  // namespace

bool visible(const Vector& a, const Vector& b)
{
    Ray r;
    r.dir = b - a;
    r.start = a;
    double limit = distance(a, b);
    r.dir.normalize();
    for (Node* n : scene.nodes) {
        IntersectionInfo ii;
        if (n->intersect(r, ii) && ii.dist < limit) return false;
    }
    return true;
}

Vector hemisphereSample(const IntersectionInfo& info)
{
    double u = g_table.randdouble();
    double v = g_table.randdouble();
    double theta = 2 * PI * u;
    double phi = acos(2 * v - 1);
    Vector dir(sin(phi) * cos(theta), cos(phi), sin(phi) * sin(theta));
    return dot(dir, info.norm) > 0 ? dir : -dir;
}

Color raytrace(const Ray& ray)
{
    if (ray.depth > scene.settings.maxTraceDepth) return Color(0, 0, 0);
    Nearest h = nearestHit(ray);
    if (h.light) return h.light->getColor();
    if (!h.node) return (scene.environment && scene.environment->loaded) ? scene.environment->getEnvironment(ray.dir) : Color(0, 0, 0);
    bumpIfAny(*h.node, h.info);
    return h.node->shader->shade(ray, h.info);
}

static Color pathtraceRef(Ray ray, Color weight)
{
    std::vector<Color> direct;
    Color tail(0, 0, 0);
    for (;;) {
        if (ray.depth > scene.settings.maxTraceDepth || weight.intensity() < 0.01) break;
        Nearest h = nearestHit(ray);
        if (h.light) { if (!(ray.flags & RF_DIFFUSE)) tail = h.light->getColor() * weight; break; }
        if (!h.node) { if (scene.environment && scene.environment->loaded) tail = scene.environment->getEnvironment(ray.dir) * weight; break; }
        bumpIfAny(*h.node, h.info);
        Shader* sh = h.node->shader;
        Ray thrown = ray;
        thrown.depth++;
        thrown.start = h.info.ip + h.info.norm * 1e-6;
        Color c0;
        float p0;
        sh->spawnRay(h.info, ray, thrown, c0, p0);                 // discarded draw
        const Color contribLight = sampleOneLight(ray, h.info, weight, sh);   // evaluated (it draws random numbers) before the real spawn, main.cpp:227-228
        Ray next = ray;
        next.depth++;
        Color f;
        float pdf;
        sh->spawnRay(h.info, ray, next, f, pdf);
        // main.cpp:238-239: both sentinels return WITHOUT this bounce's light term (unreachable with the shaders fray ships: their spawnRay
        // yields 1 / 2pi, 1e9 or 1 -- shading.cpp:98,225,294,297, shading.h:133 -- but the restatement must not be the sloppier of the two)
        if (pdf == -1) { tail = Color(1, 0, 0); break; }
        if (pdf == 0) break;
        direct.push_back(contribLight);
        weight = weight * f / pdf;
        ray = next;
    }
    for (size_t k = direct.size(); k-- > 0;) tail = direct[k] + tail;
    return tail;
}

// ------------------------------------------------------------------ harness entry points
static unsigned fmix(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
static unsigned sampleSeed(unsigned seed, unsigned p, unsigned i) { unsigned h = fmix(seed ^ (p * 0x9e3779b1u)); return fmix(h ^ (i * 0x85ebca77u) ^ 0x27d4eb2fu); }

extern "C" {

// Parses with the reference's own parser; overrides follow "name=value;..." for the settings /
// camera fields the tests change.  Returns 0 on success.
int ref_load(const char* path, int W, int H, const char* overrides)
{
    static bool loaded = false;
    if (loaded) return -2;          // the reference's `scene` is a process-wide singleton
    loaded = true;
    {   // The parser draws its randfloat / randint macros from getRandomGen(0) (scene.cpp:405, 449, 609-653), which main() has put into the state
        // initRandom(42) leaves in table entry 0 (main.cpp:502, random_generator.cpp:91-108): seeded, warmed up 1223 times, then one word and
        // one randint(0, 1222) drawn for entry 1 (the later entries draw from their own predecessors).  Restated here for the one generator that
        // stands for the table.
        g_table.seed(42u ^ 0xbf14ef80u);
        for (int i = 0; i < 1223; i++) g_table._next();
        g_table._next();
        g_table.randint(0, 1222);
    }
    if (!scene.parseScene(path)) return -1;
    GlobalSettings& s = scene.settings;
    s.frameWidth = W; s.frameHeight = H; s.numThreads = 1; s.wantPrepass = false; s.interactive = false;
    std::string o = overrides ? overrides : "";
    size_t pos = 0;
    while (pos < o.size()) {
        size_t e = o.find(';', pos);
        if (e == std::string::npos) e = o.size();
        std::string kv = o.substr(pos, e - pos);
        pos = e + 1;
        size_t q = kv.find('=');
        if (q == std::string::npos) continue;
        std::string k = kv.substr(0, q);
        double v = atof(kv.c_str() + q + 1);
        if (k == "wantAA") s.wantAA = v != 0;
        else if (k == "gi") s.gi = v != 0;
        else if (k == "numPaths") s.numPaths = (int)v;
        else if (k == "maxTraceDepth") s.maxTraceDepth = (int)v;
        else if (k == "dof") scene.camera->dof = v != 0;
        else if (k == "numDOFSamples") scene.camera->numDOFSamples = (int)v;
        else if (k == "stereoSeparation") scene.camera->stereoSeparation = v;
        else return -3;
    }
    g_w = W; g_h = H;
    scene.beginRender();
    scene.beginFrame();
    return 0;
}

// parseScene + beginRender + beginFrame with NOTHING overridden afterwards (ref_load forces the frame size, one thread, no prepass): what
// ref_dump_scene (oracle/ref_dump.cpp) is compared on.  The macro generator stands where initRandom(42) leaves it, as in ref_load.
int ref_parse(const char* path)
{
    static bool parsed = false;
    if (parsed) return -2;
    parsed = true;
    g_table.seed(42u ^ 0xbf14ef80u);
    for (int i = 0; i < 1223; i++) g_table._next();
    g_table._next();
    g_table.randint(0, 1222);
    if (!scene.parseScene(path)) return -1;
    g_w = scene.settings.frameWidth; g_h = scene.settings.frameHeight;
    scene.beginRender();
    scene.beginFrame();
    return 0;
}

int ref_counts(int* nodes, int* lights) { *nodes = (int)scene.nodes.size(); *lights = (int)scene.lights.size(); return 0; }

// closest hit of an arbitrary ray: returns id (node, -1, -2-light); out = dist, ip3, norm3, u, v
int ref_probe(const double* start, const double* dir, double* out)
{
    Ray r(Vector(start[0], start[1], start[2]), Vector(dir[0], dir[1], dir[2]));
    Nearest h = nearestHit(r);
    out[0] = h.info.dist;
    if (h.node || h.light) {
        out[1] = h.info.ip.x; out[2] = h.info.ip.y; out[3] = h.info.ip.z;
        out[4] = h.info.norm.x; out[5] = h.info.norm.y; out[6] = h.info.norm.z;
        out[7] = h.light ? 0 : h.info.u; out[8] = h.light ? 0 : h.info.v;
    } else for (int i = 1; i < 9; i++) out[i] = 0;
    return h.light ? -2 - h.lightIdx : h.nodeIdx;
}

void ref_camera_ray(double x, double y, double* start, double* dir)
{
    Ray r = scene.camera->getScreenRay(x, y);
    start[0] = r.start.x; start[1] = r.start.y; start[2] = r.start.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

void ref_primary(int* ids, double* dist)
{
    for (int y = 0; y < g_h; y++)
        for (int x = 0; x < g_w; x++) {
            Nearest h = nearestHit(scene.camera->getScreenRay(x, y));
            ids[y * g_w + x] = h.light ? -2 - h.lightIdx : h.nodeIdx;
            dist[y * g_w + x] = h.info.dist;
        }
}

// Whole frame under the RNG contract (both generators re-seeded per camera sample).
void ref_render(float* rgb, unsigned seed)
{
    static const double aa[5][2] = {{0, 0}, {0.6, 0}, {0.3, 0.3}, {0, 0.6}, {0.6, 0.6}};
    int spp = scene.settings.wantAA ? 5 : 1;
    if (scene.camera->dof) spp = std::max(spp, scene.camera->numDOFSamples);
    if (scene.settings.gi) spp = std::max(spp, scene.settings.numPaths);
    const bool jitter = scene.camera->dof || scene.settings.gi;
    for (int y = 0; y < g_h; y++)
        for (int x = 0; x < g_w; x++) {
            Color sum(0, 0, 0);
            for (int i = 0; i < spp; i++) {
                unsigned s = sampleSeed(seed, (unsigned)(y * g_w + x), (unsigned)i);
                g_local.seed(s);
                g_table.seed(s);
                float ox, oy;
                if (jitter) { ox = g_local.randfloat(); oy = g_local.randfloat(); }
                else { ox = (float)aa[i][0]; oy = (float)aa[i][1]; }
                double fx = x + ox, fy = y + oy;
                // raytraceSinglePixel / getRay / trace, main.cpp:287-321 (restated): both eyes' rays are made before either is traced
                auto getRay = [&](WhichCamera which) { return scene.camera->dof ? scene.camera->getDOFRay(fx, fy, which) : scene.camera->getScreenRay(fx, fy, which); };
                auto trace = [&](const Ray& r) { return scene.settings.gi ? pathtraceRef(r, Color(1, 1, 1)) : raytrace(r); };
                if (scene.camera->stereoSeparation > 0) {
                    Ray leftRay = getRay(CAMERA_LEFT);
                    Ray rightRay = getRay(CAMERA_RIGHT);
                    Color colorLeft = trace(leftRay);
                    Color colorRight = trace(rightRay);
                    if (scene.settings.saturation != 1) {
                        colorLeft.adjustSaturation(scene.settings.saturation);
                        colorRight.adjustSaturation(scene.settings.saturation);
                    }
                    sum += colorLeft * scene.camera->leftMask + colorRight * scene.camera->rightMask;
                } else {
                    sum += trace(getRay(CAMERA_CENTER));
                }
            }
            sum = sum / spp;
            float* o = rgb + 3 * (y * g_w + x);
            o[0] = sum.r; o[1] = sum.g; o[2] = sum.b;
        }
}

}  // extern "C"
