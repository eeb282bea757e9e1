// Shading on the device: hit-attribute reconstruction, textures and bump mapping, light sampling,
// the Whitted direct-lighting evaluators (Lambert / Phong), the BRDF eval / spawnRay pairs of the
// path tracer and next-event estimation.  Precision follows the reference: FP64 geometry, FP32
// colour, with its float<->double conversions kept where it has them.
#pragma once
#include "dev_math.hpp"
#include "dev_rng.hpp"
#include "dev_scene.hpp"
#include "dev_trace.hpp"

enum { RF_DIFFUSE = 2 };   // vector.h:215-219

struct HitInfo {           // IntersectionInfo, geometry.h:33-39 (after Node::intersect)
    V3 ip, norm, dNdx, dNdy;
    double u, v;
};

// Rebuilds what Node::intersect + <Geometry>::intersect leave in `info` for the winning node
// (geometry.cpp:30-83, 196-208; mesh.cpp:112-137).  needUV gates the sphere's atan2/asin, whose
// result only textures and bump maps read.
// Local normal / uv of a non-CSG geometry at local point ipl (what its intersect() writes).
FD void prim_attributes(const DScene& S, int kind, int index, V3 ipl, int code, double l2, double l3, bool needUV, bool needBump, V3& nl, HitInfo& info)
{
    if (kind == 0) {
        nl = v3(0, 1, 0);
        info.u = ipl.x;
        info.v = ipl.z;
    } else if (kind == 1) {
        const FRAY_RO DSphere& Sp = S.spheres[index];
        nl = normalized(ipl - ld3(Sp.O));
        if (needUV) {
            info.u = ((atan2(nl.z, nl.x) / FRAY_PI * 180.0) + 180.0) / 360.0;
            info.v = 1 - ((asin(nl.y) / FRAY_PI * 180.0) + 90) / 180.0;
        }
    } else if (kind == 2) {   // Cube::intersect's per-side normal and uv lambdas, geometry.cpp:113-126
        const int ax = code >> 1;
        const double sg = (code & 1) ? +1.0 : -1.0;
        nl = v3(ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0);
        if (ax == 0) { info.u = ipl.y; info.v = ipl.z; }
        else if (ax == 1) { info.u = ipl.x; info.v = ipl.z; }
        else { info.u = ipl.x; info.v = ipl.y; }
    } else {
        const FRAY_RO DMesh& M = S.meshes[index];
        const FRAY_RO DTriAttr* A = M.attrs + code;
        if (M.smooth) {
            V3 nA = ld3(A->nA), nB = ld3(A->nB), nC = ld3(A->nC);
            nl = normalized(nA + (nB - nA) * l2 + (nC - nA) * l3);
        } else {
            nl = ld3(M.tris[code].g);
        }
        if (M.hasUV) {
            info.u = A->tA[0] + (A->tB[0] - A->tA[0]) * l2 + (A->tC[0] - A->tA[0]) * l3;
            info.v = A->tA[1] + (A->tB[1] - A->tA[1]) * l2 + (A->tC[1] - A->tA[1]) * l3;
        }
        if (needBump) {                       // only BumpTexture::modifyNormal reads them (shading.cpp:397-418)
            info.dNdx = ld3(A->dNdx);
            info.dNdy = ld3(A->dNdy);
        }
    }
}

// BARY: the hit record carries no barycentrics (the path tracer's hit queue): they are re-derived for meshes that read them.
template <int ST, bool BARY = false>
FD void finalize_hit(const DScene& S, const HitT<ST>& h, V3 o, V3 d, bool needUV, HitInfo& info)
{
    const FRAY_RO DNode& N = S.nodes[h.node];
    needUV = needUV && tex_variant(ST);                       // no texture in the scene reads (u, v)
    const bool needBump = tex_variant(ST) && N.bumpTex >= 0;
    V3 ls = mulM(o - ld3(N.T.off), N.T.inv);
    V3 ldir = normalized(mulM(d, N.T.inv));
    V3 ipl = ls + ldir * h.t;
    V3 nl;
    info.dNdx = v3(0, 0, 0);
    info.dNdy = v3(0, 0, 0);
    info.u = 0; info.v = 0;
    bool done = false;
    if constexpr ((ST & 2) != 0) {
        if (N.geomKind == 2) {
            ipl = h.ipl;                                          // Cube::intersect's own ip and side, kept by closest_hit
            prim_attributes(S, 2, N.geomIndex, ipl, h.tri, 0, 0, needUV, needBump, nl, info);
            done = true;
        } else if (N.geomKind == 4) {
            ipl = h.ipl;                                          // the winning intersection of CsgOp::intersect as its leaf geometry reported it
            prim_attributes(S, h.leafKind, h.leafIndex, ipl, h.tri, h.l2, h.l3, needUV, needBump, nl, info);
            done = true;
        }
    }
    if (!done) {
        double l2 = h.l2, l3 = h.l3;
        if (BARY && N.geomKind == 3) {
            const FRAY_RO DMesh& M = S.meshes[N.geomIndex];
            if (M.smooth || M.hasUV) tri_bary(M.tris + h.tri, ls, ldir, l2, l3);
        }
        prim_attributes(S, N.geomKind, N.geomIndex, ipl, h.tri, l2, l3, needUV, needBump, nl, info);
    }
    info.ip = mulM(ipl, N.T.m) + ld3(N.T.off);
    info.norm = normalized(mulM(nl, N.T.m));
}

// ---- textures -----------------------------------------------------------------------------------
template <int ST>
FD C3 texel(const FRAY_RO DTexture& T, int x, int y, Cnt& c)   // Bitmap::getPixel, bitmap.cpp:67-71
{
    bump<ST>(c.tex);
    if (T.width <= 0 || x < 0 || x >= T.width || y < 0 || y >= T.height) return c3(0, 0, 0);
    return ldc(T.texels + 3 * ((long long)x + (long long)y * T.width));
}
FD void wrap_texel(const FRAY_RO DTexture& T, double u, double v, int& ix, int& iy)   // shading.cpp:149-155, 404-410
{
    ix = int(floor(u * T.scaling * T.width));
    iy = int(floor(v * T.scaling * T.height));
    ix %= T.width;
    iy %= T.height;
    if (ix < 0) ix += T.width;
    if (iy < 0) iy += T.height;
}
FD float fresnel_schlick(V3 i, V3 n, float ior)   // shading.cpp:230-236
{
    float q = (1.0f - ior) / (1.0f + ior);
    float f = (float)((double)q * (double)q);       // sqr() works in double
    float NdotI = (float)-dot(n, i);
    return f + (1.0f - f) * powf(1.0f - NdotI, 5.0f);
}
template <int ST>
FD C3 texture_sample(const DScene& S, int t, V3 rayDir, const HitInfo& info, Cnt& c)
{
    if (!tex_variant(ST)) return c3(1, 1, 1);                 // never reached: the scene has no texture
    const FRAY_RO DTexture& T = S.textures[t];
    if (T.kind == 0) {   // CheckerTexture::sample, shading.cpp:40-46
        int ix = int(floor(info.u * T.scaling) / 5.0);
        int iy = int(floor(info.v * T.scaling) / 5.0);
        return ((ix + iy) % 2 == 0) ? ldc(T.color1) : ldc(T.color2);
    }
    if (T.kind == 1) {   // BitmapTexture::sample, shading.cpp:147-158
        int ix, iy;
        wrap_texel(T, info.u, info.v, ix, iy);
        return texel<ST>(T, ix, iy, c);
    }
    if (T.kind == 3) {   // FresnelTexture::sample, shading.cpp:369-385
        V3 n;
        double myIor;
        if (dot(rayDir, info.norm) < 0) { n = info.norm; myIor = T.ior; }
        else { n = -info.norm; myIor = 1.0 / T.ior; }
        float f = fresnel_schlick(rayDir, n, (float)myIor);
        return c3(f, f, f);
    }
    return c3(0, 0, 0);   // BumpTexture::sample, shading.cpp:392-395
}
// applyBumpMapping (main.cpp:82-90) -> BumpTexture::modifyNormal (shading.cpp:397-418)
template <int ST>
FD void apply_bump(const DScene& S, int nodeIdx, HitInfo& info, Cnt& c)
{
    if (!tex_variant(ST)) return;
    int bt = S.nodes[nodeIdx].bumpTex;
    if (bt < 0) return;
    const FRAY_RO DTexture& T = S.textures[bt];
    if (T.kind != 2) return;
    int ix, iy;
    wrap_texel(T, info.u, info.v, ix, iy);
    C3 t = texel<ST>(T, ix, iy, c);
    float dx = (float)(t.r * T.bumpIntensity);
    float dy = (float)(t.g * T.bumpIntensity);
    info.norm = info.norm + ((double)dx * info.dNdx + (double)dy * info.dNdy) * T.bumpIntensity;
    info.norm = normalized(info.norm);
}

// CubemapEnvironment::getEnvironment, environment.cpp:64-98
template <int ST>
FD C3 environment(const DScene& S, V3 dir, Cnt& c)
{
    if (!tex_variant(ST) || !S.env.present || !S.env.loaded) return c3(0, 0, 0);
    double maxVal = fabs(dir.x);
    int dim = 0;
    if (fabs(dir.y) > maxVal) { dim = 1; maxVal = fabs(dir.y); }
    if (fabs(dir.z) > maxVal) dim = 2;
    bool positive = comp(dir, dim) > 0;
    V3 on = dir * (1.0 / fabs(comp(dir, dim)));
    int face = (positive ? 3 : 0) + dim;
    double sx, sy;
    switch (face) {
        case 0: sx = on.z; sy = -on.y; break;
        case 3: sx = -on.z; sy = -on.y; break;
        case 1: sx = on.x; sy = -on.z; break;
        case 4: sx = on.x; sy = on.z; break;
        case 2: sx = on.x; sy = on.y; break;
        default: sx = on.x; sy = -on.y; break;
    }
    int W = S.env.width[face], H = S.env.height[face];
    int ix = (int)(((sx + 1) / 2) * W);
    int iy = (int)(((sy + 1) / 2) * H);
    bump<ST>(c.tex);
    if (ix < 0 || ix >= W || iy < 0 || iy >= H) return c3(0, 0, 0);
    return ldc(S.env.face[face] + 3 * ((long long)ix + (long long)iy * W));
}

// ---- lights (lights.h:41-47, lights.cpp:31-77,105-108) ---------------------------------------------
FD int light_num_samples(const FRAY_RO DLight& L) { return L.kind == 0 ? 1 : L.xSubd * L.ySubd; }
FD C3 light_color(const FRAY_RO DLight& L) { return ldc(L.color) * L.power; }
template <class G>
FD void light_nth_sample(const FRAY_RO DLight& L, int idx, V3 shadePos, G& tab, V3& samplePos, C3& color)
{
    if (L.kind == 0) {
        samplePos = ld3(L.pos);
        color = ldc(L.color) * L.power;
        return;
    }
    int col = idx % L.xSubd;
    int row = idx / L.xSubd;
    double cellW = L.areaXsize;
    double cellH = L.areaYsize;
    double x0 = col * cellW;
    double y0 = row * cellH;
    double p_x = x0 + cellW * rng_float(tab);
    double p_y = y0 + cellH * rng_float(tab);
    V3 pl = v3(p_x - 0.5, 0, p_y - 0.5);
    V3 sp = mulM(shadePos - ld3(L.T.off), L.T.inv);
    if (sp.y > 0) {
        color = c3(0, 0, 0);
    } else {
        float cw = float(fray_div(dot(v3(0, -1, 0), sp), length(sp)));
        color = ldc(L.color) * L.power * (float)L.area * cw;
    }
    samplePos = mulM(pl, L.T.m) + ld3(L.T.off);
}
FD double light_solid_angle(const FRAY_RO DLight& L, V3 ip)
{
    if (L.kind == 0) return 0;
    double q = lengthSqr(ip - ld3(L.center));
    return fray_div(L.area, 1.0 < q ? q : 1.0);   // std::max(1.0, q)
}

// ---- Whitted: Lambert::shade / Phong::shade (shading.cpp:48-80, 101-144) ---------------------------
template <int ST, class G>
FD C3 shade_direct(const DScene& S, const FRAY_RO DShader& sh, V3 rayDir, const HitInfo& info, G& tab, bool phong, Cnt& c)
{
    C3 diffuse = ldc(sh.color);
    if (sh.texture >= 0) diffuse = diffuse * texture_sample<ST>(S, sh.texture, rayDir, info, c);
    C3 result = diffuse * ldc(S.ambient);
    const int nl = S.nLights;
    for (int li = 0; li < nl; li++) {
        const FRAY_RO DLight& L = S.lights[li];
        const int ns = light_num_samples(L);
        C3 sum = c3(0, 0, 0);
        for (int k = 0; k < ns; k++) {
            C3 lc;
            V3 lp;
            light_nth_sample(L, k, info.ip, tab, lp, lc);
            double d2 = lengthSqr(info.ip - lp);
            V3 wl = normalized(lp - info.ip);
            V3 n = faceforward(rayDir, info.norm);
            float cosN = (float)dot(wl, n);
            float lam = (float)(cosN / d2);
            lam = lam > 0.0f ? lam : 0.0f;   // max(0.0f, x)
            if (visible<ST>(S, info.ip + n * 1e-6, lp, c)) {
                C3 r = diffuse * lc * lam;
                if (phong) {
                    V3 wi = -wl;
                    V3 rr = reflect(wi, n);
                    double cosV = dot(-rayDir, rr);
                    if (cosV > 0)
                        r = r + lc / (float)d2 * ldc(sh.specularColor) * (float)pow(cosV, sh.exponent) * (float)sh.specularMultiplier;
                }
                sum = sum + r;
            }
        }
        result = result + sum / (float)ns;
    }
    return result;
}

// ---- path tracing: BRDF::eval / BRDF::spawnRay ---------------------------------------------------------
template <class G>
FD V3 hemisphere_sample(G& tab, V3 norm)   // main.cpp:92-116
{
    double u = rng_double(tab);
    double v = rng_double(tab);
    double theta = 2 * FRAY_PI * u;
    double sp, cp, st, ct;          // phi = acos(2 v - 1); dev_trig.hpp: correctly rounded, which is what glibc's are in 99.85 % of calls
    fray_acos_sincos(2 * v - 1, &sp, &cp);
    fray_sincos(theta, &st, &ct);
    V3 dir = v3(sp * ct, cp, sp * st);
    if (dot(dir, norm) > 0) return dir;
    return -dir;
}
FD C3 brdf_eval(const FRAY_RO DShader& sh, const HitInfo& x, V3 w_out)
{
    if (sh.kind == 1) {   // Lambert::eval, shading.cpp:82-86
        double dd = dot(x.norm, w_out);
        float cosTerm = (float)(0.0 < dd ? dd : 0.0);
        return ldc(sh.color) * (float)fray_div(cosTerm, FRAY_PI);
    }
    if (sh.kind == 3 || sh.kind == 4) return c3(0, 0, 0);   // Reflection / Refraction::eval
    return c3(1, 0, 0);                                     // Shader::eval default, shading.h:124-127
}
// Number of table-generator words the discarded first spawnRay consumes (main.cpp:219-224).
FD int spawn_words(const FRAY_RO DShader& sh) { return sh.kind == 1 ? 4 : 0; }

struct PathRay { V3 o, d; int depth; unsigned flags; };

template <class G>
FD void spawn_ray(const FRAY_RO DShader& sh, const HitInfo& x, const PathRay& w_in, G& tab, PathRay& w_out, C3& brdf, float& pdf)
{
    w_out = w_in;
    if (sh.kind == 1) {   // Lambert::spawnRay, shading.cpp:88-99
        w_out.depth = w_in.depth + 1;
        w_out.o = x.ip + x.norm * 1e-6;
        w_out.d = hemisphere_sample(tab, x.norm);
        w_out.flags |= RF_DIFFUSE;
        double dd = dot(x.norm, w_out.d);
        float cosTerm = (float)(0.0 < dd ? dd : 0.0);
        brdf = ldc(sh.color) * (float)fray_div(cosTerm, FRAY_PI);
        pdf = (float)(1 / (2 * FRAY_PI));
        return;
    }
    if (sh.kind == 3) {   // Reflection::spawnRay, shading.cpp:217-227
        V3 n = faceforward(w_in.d, x.norm);
        w_out.depth = w_in.depth + 1;
        w_out.o = x.ip + n * 1e-6;
        w_out.d = reflect(w_in.d, x.norm);
        w_out.flags &= ~(unsigned)RF_DIFFUSE;
        brdf = ldc(sh.mult) * 1e9f;
        pdf = 1e9f;
        return;
    }
    if (sh.kind == 4) {   // Refraction::spawnRay, shading.cpp:270-299
        V3 n = faceforward(w_in.d, x.norm);
        double myIor = dot(n, x.norm) > 0 ? 1.0 / sh.ior : sh.ior / 1.0;
        V3 refr = refract(w_in.d, n, myIor);
        if (!(refr.x == 0 && refr.y == 0 && refr.z == 0)) {
            w_out.o = x.ip - n * 1e-6;
            w_out.d = refr;
            w_out.depth = w_in.depth + 1;
            w_out.flags &= ~(unsigned)RF_DIFFUSE;
            brdf = ldc(sh.mult) * 1e9f;
            pdf = 1e9f;
        } else {
            // brdf 0, pdf 1; w_out stays as pathtrace() initialised it: the incoming ray, depth + 1
            w_out.depth = w_in.depth + 1;
            brdf = c3(0, 0, 0);
            pdf = 1.0f;
        }
        return;
    }
    // Shader::spawnRay default (Const / Phong / Layered), shading.h:128-134: same ray again
    w_out.depth = w_in.depth + 1;
    brdf = c3(1, 0, 0);
    pdf = 1;
}

// explicitLightSample (main.cpp:118-169) with the visibility query split off: everything that draws
// random numbers or evaluates the BRDF happens here; the returned segment a->b and the contribution
// go to the shadow queue, and k_pt_shadow adds `contrib` iff visible(a, b).  The reference asks
// visible() before eval(); neither draws random numbers, so the order does not matter.
template <class GR, class GT>
FD bool nee_prepare(const DScene& S, V3 rayDir, const HitInfo& info, C3 pm, const FRAY_RO DShader& sh, GR& rnd, GT& tab, V3& a, V3& b, C3& contrib)
{
    if (S.nLights == 0) return false;
    int lightIdx = rng_int0(rnd, S.nLights - 1);
    const FRAY_RO DLight& L = S.lights[lightIdx];
    V3 x = info.ip;
    double solidAngle = light_solid_angle(L, x);
    if (solidAngle == 0) return false;
    int randSample = rng_int0(rnd, light_num_samples(L) - 1);
    V3 pl;
    C3 unused;
    light_nth_sample(L, randSample, x, tab, pl, unused);
    a = x + info.norm * 1e-6;
    b = pl;
    C3 Le = light_color(L);
    V3 w_out = normalized(pl - x);
    C3 brdfAtPoint = brdf_eval(sh, info, w_out);
    if (intensity(brdfAtPoint) == 0) { contrib = c3(0, 0, 0); return true; }
    float probHitLightArea = (float)fray_rcp(solidAngle);
    float probPickThisLight = S.probPickLight;
    float chooseLightProb = probHitLightArea * probPickThisLight;
    contrib = Le * pm * brdfAtPoint / chooseLightProb;
    return true;
}
