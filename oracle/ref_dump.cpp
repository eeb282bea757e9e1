// Test infrastructure, part of oracle/_ref (built only where /root/reference is mounted): a text dump of what the reference's OWN parser
// (scene.o: DefaultSceneParser, scene.cpp:403-570, every fillProperties, the OBJ / BMP loaders, Transform's matrices) left in `Scene scene`
// after parseScene + beginRender + beginFrame -- settings, camera, lights and, per render-list node, its transform, its geometry tree and its
// shader tree with their textures.  oracle/scene_dump.py writes the same text from the PRODUCT's frayhip_scene_desc; tests/test_host_scene.py
// compares the two on the parser's edge cases (comments, quotes, singletons, transform order, Layered lines, randfloat / randint macros).
//
// The reference's headers are used as they are.  Several members this dump reads are private or protected (Layered::layers, Reflection's
// parameters, the lights' fields, Mesh's arrays): THIS translation unit only is compiled with g++'s -fno-access-control (oracle/Makefile.ref) --
// object layout does not depend on access, and no reference source is changed or copied.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdarg.h>
#include <functional>
#include <random>
#include <string>
#include <vector>
#include <typeinfo>
#include "bitmap.h"
#include "camera.h"
#include "color.h"
#include "environment.h"
#include "geometry.h"
#include "lights.h"
#include "mesh.h"
#include "scene.h"
#include "shading.h"

namespace {
std::string g_out;
void put(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_out += buf;
}
void d(double v) { put(" %a", v); }
void f(float v) { put(" %a", (double)v); }
void vec(const Vector& v) { d(v.x); d(v.y); d(v.z); }
void col(const Color& c) { f(c.r); f(c.g); f(c.b); }
void xform(const Transform& T)
{
    put(" T"); vec(T.offset);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d(T.m.m[i][j]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d(T.invM.m[i][j]);
}
void texture(Texture* t)
{
    if (!t) { put(" tex-none"); return; }
    if (auto* c = dynamic_cast<CheckerTexture*>(t)) { put(" tex-checker"); col(c->color1); col(c->color2); d(c->scaling); }
    else if (auto* b = dynamic_cast<BitmapTexture*>(t)) { put(" tex-bitmap %d %d", b->bmp.getWidth(), b->bmp.getHeight()); d(b->scaling); }
    else if (auto* u = dynamic_cast<BumpTexture*>(t)) { put(" tex-bump %d %d", u->bumpTex.getWidth(), u->bumpTex.getHeight()); d(u->scaling); d(u->bumpIntensity); }
    else if (auto* r = dynamic_cast<FresnelTexture*>(t)) { put(" tex-fresnel"); d(r->ior); }
    else put(" tex-unknown");
}
void shader(Shader* s, int depth)
{
    if (!s) { put(" shader-none"); return; }
    if (depth > 40) { put(" shader-too-deep"); return; }
    if (auto* c = dynamic_cast<ConstantShader*>(s)) { put(" const"); col(c->color); }
    else if (auto* l = dynamic_cast<Lambert*>(s)) { put(" lambert"); col(l->color); texture(l->diffuseTex); }
    else if (auto* p = dynamic_cast<Phong*>(s)) { put(" phong"); col(p->color); col(p->specularColor); d(p->exponent); d(p->specularMultiplier); texture(p->diffuseTex); }
    else if (auto* r = dynamic_cast<Reflection*>(s)) { put(" refl"); col(r->mult); d(r->glossiness); d(r->deflectionScaling); put(" %d", r->numSamples); }
    else if (auto* q = dynamic_cast<Refraction*>(s)) { put(" refr"); col(q->mult); d(q->ior); }
    else if (auto* y = dynamic_cast<Layered*>(s)) {
        put(" layered %d [", y->numLayers);
        for (int i = 0; i < y->numLayers; i++) { put(" layer"); col(y->layers[i].opacity); texture(y->layers[i].texture); shader(y->layers[i].shader, depth + 1); }
        put(" ]");
    } else put(" shader-unknown");
}
void geometry(Geometry* g, int depth)
{
    if (!g) { put(" geom-none"); return; }
    if (depth > 40) { put(" geom-too-deep"); return; }
    if (auto* p = dynamic_cast<Plane*>(g)) { put(" plane"); d(p->limit); d(p->height); }
    else if (auto* s = dynamic_cast<Sphere*>(g)) { put(" sphere"); vec(s->O); d(s->R); }
    else if (auto* c = dynamic_cast<Cube*>(g)) { put(" cube"); vec(c->O); d(c->halfSide); }
    else if (auto* m = dynamic_cast<Mesh*>(g)) {
        put(" mesh %d %d %d %d faceted %d culling %d kd %d", (int)m->vertices.size(), (int)m->normals.size(), (int)m->uvs.size(), (int)m->triangles.size(),
            (int)m->faceted, (int)m->backfaceCulling, m->kdRoot ? 1 : 0);
        vec(m->bbox.vmin); vec(m->bbox.vmax);
        // a checksum-free spot check of the loaders: the first and the last vertex and triangle
        if (!m->vertices.empty()) { vec(m->vertices.front()); vec(m->vertices.back()); }
        if (!m->triangles.empty()) {
            const Triangle* ts[2] = {&m->triangles.front(), &m->triangles.back()};
            for (const Triangle* t : ts) { put(" tri %d %d %d %d %d %d %d %d %d", t->v[0], t->v[1], t->v[2], t->n[0], t->n[1], t->n[2], t->t[0], t->t[1], t->t[2]); vec(t->gnormal); vec(t->AB); vec(t->AC); vec(t->ABcrossAC); }
        }
    } else if (auto* o = dynamic_cast<CsgOp*>(g)) {
        put(dynamic_cast<CsgPlus*>(o) ? " csg-plus (" : (dynamic_cast<CsgIntersect*>(o) ? " csg-and (" : " csg-minus ("));
        geometry(o->left, depth + 1);
        put(" ,");
        geometry(o->right, depth + 1);
        put(" )");
    } else put(" geom-unknown");
}
}  // namespace

extern "C" int ref_dump_scene(char* out, int cap)
{
    g_out.clear();
    const GlobalSettings& s = scene.settings;
    put("settings %d %d aa %d gi %d paths %d depth %d prepass %d", s.frameWidth, s.frameHeight, (int)s.wantAA, (int)s.gi, s.numPaths, s.maxTraceDepth, (int)s.wantPrepass);
    f(s.saturation); col(s.ambientLight); put("\n");
    const Camera& c = *scene.camera;
    put("camera"); vec(c.pos); d(c.yaw); d(c.pitch); d(c.roll); d(c.fov); d(c.aspectRatio); d(c.focalPlaneDist); d(c.fNumber); d(c.stereoSeparation);
    put(" dof %d autofocus %d samples %d", (int)c.dof, (int)c.autofocus, c.numDOFSamples); col(c.leftMask); col(c.rightMask); put("\n");
    put("environment %d\n", scene.environment ? 1 : 0);
    put("lights %d\n", (int)scene.lights.size());
    for (Light* L : scene.lights) {
        if (auto* p = dynamic_cast<PointLight*>(L)) { put("light point"); col(p->color); f(p->power); vec(p->pos); }
        else if (auto* r = dynamic_cast<RectLight*>(L)) { put("light rect"); col(r->color); f(r->power); put(" %d %d", r->xSubd, r->ySubd); xform(r->T); vec(r->center); d(r->area); }
        else put("light unknown");
        put("\n");
    }
    put("nodes %d\n", (int)scene.nodes.size());
    for (Node* n : scene.nodes) {
        put("node"); xform(n->T); put(" |"); geometry(n->geometry, 0); put(" |"); shader(n->shader, 0); put(" | bump"); texture(n->bump); put("\n");
    }
    if ((int)g_out.size() + 1 > cap) return -(int)g_out.size() - 1;
    memcpy(out, g_out.c_str(), g_out.size() + 1);
    return (int)g_out.size();
}
