// Whitted integrator with recursion: raytrace() (main.cpp:246-285) calling Shader::shade, where
// Reflection / Refraction re-enter raytrace() and Layered calls shade() of every layer
// (shading.cpp:160-207, 238-263, 357-367).  The reference's call tree consumes the per-thread
// generator in depth-first order (RectLight samples, glossy-reflection normals), so the tree is
// walked depth-first here too, by one lane, with an explicit stack of pending shade() activations
// instead of recursion; the order of random draws and of FP32 colour arithmetic is the reference's.
#pragma once
#include "dev_shade.hpp"

enum { WF_MULT = 0, WF_GLOSSY = 1, WF_LAYERED = 2 };

struct WFrame {
    int kind, shader, i, count;
    C3 acc;        // MULT: the multiplier; GLOSSY: sum so far; LAYERED: result so far
    C3 opacity;    // LAYERED: opacity of the layer being evaluated
    V3 o, d;       // the ray being shaded
    int depth;
    HitInfo info;
};

#define FRAY_WSTACK 40

// One lane's raytrace() machine.  `mode` says where the lane stands; the kernel (k_whitted, kernels.hpp) drives all lanes of a wave
// through rounds -- every lane takes its cheap steps (wl_cheap_step: returns, loop heads of glossy / layered activations, pushes of
// recursive shaders) until each stands at a closest-hit search (WM_TRACE), at a direct-light loop (WM_SHADE of a Constant / Lambert /
// Phong surface) or at the end of its camera ray (WM_ROOT_RET); then ONE search runs for the whole wave, then ONE direct-light loop.
// A lane's own sequence of steps, random draws and FP32 colour operations is the reference's depth-first call tree.
enum { WM_TRACE = 0, WM_SHADE = 1, WM_RET = 2, WM_RESUME = 3, WM_ROOT_RET = 4, WM_NEXT_SAMPLE = 5, WM_NEXT_PIXEL = 6, WM_EXHAUSTED = 7, WM_DEFER = 8 };
struct WhittedLane {
    WFrame stack[FRAY_WSTACK];
    int sp;
    V3 o, d;
    int depth;
    HitInfo info;
    int shader;
    C3 ret;
    int mode;
    int spBase;       // activations BELOW this lane's stack that the reference's recursion would hold at this point: 0, except for a glossy fan's child traced ahead
                      // (k_whitted pass B), which in place runs on top of the fan's own activation and whatever Layered shaders enclose it
};

struct SpecBuf {
    int* counters;                    // [0] entries, [1] children; after pass C [2] children looked up, [3] fans given up at a child that drew
    int minCount;                     // fans of at least this many samples are filed
    int* eSlot; int* eChildBase;      // per entry: the work item (k * nItems + pixel item), its first child
    double* eo[3];                    //            the children's common origin
    int* cEntry; double* cd[3];       // per child: its entry, its direction
    float* cc[3]; unsigned char* cok; //            its colour; 1 = traced without a draw (between passes A and B: the stack level the child starts on in place)
    unsigned char* cdraws;            //            how many unit-disc samples its direction took (pass C skips their words instead of redoing the trigonometry), 0 = more than 255
};
struct SpecLane { int state, sp, base, looked, missed; };      // 0 = armed, 1 = the fan on stack level sp is being looked up from child `base`, 2 = off until the next sample

// the generator of pass B: reports that it was asked, and returns words that differ so that no rejection loop spins on them
struct MtSpy {
    uint32_t x; bool used;
    FD uint32_t next() { used = true; x = x * 747796405u + 2891336453u; return x ^ (x >> 15); }
};

FD void wl_start(WhittedLane& L, V3 o, V3 d)
{
    L.sp = 0; L.spBase = 0; L.o = o; L.d = d; L.depth = 0; L.shader = -1; L.ret = c3(0, 0, 0); L.mode = WM_TRACE;
}
// is the lane at a step that needs no search and no light loop?
FD bool wl_cheap(const DScene& S, const WhittedLane& L)
{
    return L.mode == WM_RET || L.mode == WM_RESUME || (L.mode == WM_SHADE && S.shaders[L.shader].kind >= 3);
}

template <int ST, class G, int MODE>
FD void wl_cheap_step(const DScene& S, WhittedLane& L, G& tab, Cnt& c, bool& overflow, const SpecBuf& SP, SpecLane& SL)
{
    if (L.mode == WM_SHADE) {                              // shader->shade(ray, info) of a recursive shader: push its activation
        const FRAY_RO DShader& sh = S.shaders[L.shader];
        const int kind = sh.kind;
        if (L.sp + L.spBase >= FRAY_WSTACK) { overflow = true; L.sp = 0; L.ret = c3(0, 0, 0); L.mode = WM_ROOT_RET; return; }
        WFrame& f = L.stack[L.sp];
        f.shader = L.shader; f.i = 0; f.count = 0; f.o = L.o; f.d = L.d; f.depth = L.depth; f.info = L.info;
        f.acc = c3(0, 0, 0); f.opacity = c3(0, 0, 0);
        if (kind == 3) {                                   // Reflection::shade
            V3 n = faceforward(L.d, L.info.norm);
            if (sh.glossiness == 1.0) {
                f.kind = WF_MULT; f.acc = ldc(sh.mult); L.sp++;
                L.o = L.info.ip + n * 1e-6;
                L.d = reflect(f.d, n);
                L.depth = L.depth + 1;
                L.mode = WM_TRACE;
            } else {
                f.kind = WF_GLOSSY;
                f.count = L.depth == 0 ? sh.numSamples : 3;    // LOW_GLOSSY_SAMPLES, constants.h:36
                L.sp++;
                L.mode = WM_RESUME;
                // the camera sample's first depth-0 fan: filed by pass A, looked up by pass C (the same test in both: they reach it by the same steps)
                if ((MODE == 1 || MODE == 3) && L.depth == 0 && f.count >= SP.minCount && SL.state == 0) {
                    if (MODE == 1) L.mode = WM_DEFER;
                    else { SL.state = 1; SL.sp = L.sp - 1; }
                }
            }
            return;
        }
        if (kind == 4) {                                   // Refraction::shade
            V3 n = faceforward(L.d, L.info.norm);
            double myIor = dot(n, L.info.norm) > 0 ? 1.0 / sh.ior : sh.ior / 1.0;
            V3 refr = refract(L.d, n, myIor);
            if (refr.x == 0 && refr.y == 0 && refr.z == 0) { L.ret = c3(0, 0, 0); L.mode = WM_RET; return; }
            f.kind = WF_MULT; f.acc = ldc(sh.mult); L.sp++;
            L.o = L.info.ip - n * 1e-6;
            L.d = refr;
            L.depth = L.depth + 1;
            L.mode = WM_TRACE;
            return;
        }
        f.kind = WF_LAYERED;                               // Layered::shade
        L.sp++;
        L.mode = WM_RESUME;
        return;
    }
    if (L.mode == WM_RET) {                                // a call returned `ret`
        if (L.sp == 0) { L.mode = WM_ROOT_RET; return; }
        WFrame& f = L.stack[L.sp - 1];
        if (f.kind == WF_MULT) { L.ret = L.ret * f.acc; L.sp--; return; }
        if (f.kind == WF_GLOSSY) {
            f.acc = f.acc + L.ret * ldc(S.shaders[f.shader].mult);
            f.i++;
        } else {
            f.acc = L.ret * f.opacity + (c3(1, 1, 1) - f.opacity) * f.acc;
            f.i++;
        }
        L.mode = WM_RESUME;
        return;
    }
    // WM_RESUME: continue the loop of the activation on top of the stack
    WFrame& f = L.stack[L.sp - 1];
    const FRAY_RO DShader& sh = S.shaders[f.shader];
    if (f.kind == WF_GLOSSY) {                             // shading.cpp:172-204
        if (f.i == f.count) {
            if (MODE == 3 && SL.state == 1 && SL.sp == L.sp - 1) SL.state = 2;
            L.ret = f.acc / (float)f.count; L.sp--; L.mode = WM_RET; return;
        }
        if (MODE == 3 && SL.state == 1 && SL.sp == L.sp - 1) {  // the filed fan: its children's raytrace() calls ran in pass B, from the rays pass A drew for them, unless a draw came in between
            const C3 mult = ldc(sh.mult);
            while (f.i < f.count) {                             // as many of them as drew nothing, in one go: the loop of shading.cpp:172-204 with raytrace() looked up
                const int ch = SL.base + f.i;
                const int it = SP.cdraws[ch];
                if (!(SP.cok[ch] && it)) break;
                for (int q = 0; q < 4 * it; q++) (void)tab.next();  // the words of its unit-disc samples (two doubles each)
                SL.looked++;
                f.acc = f.acc + c3(SP.cc[0][ch], SP.cc[1][ch], SP.cc[2][ch]) * mult;
                f.i++;
            }
            if (f.i == f.count) { SL.state = 2; L.ret = f.acc / (float)f.count; L.sp--; L.mode = WM_RET; return; }
            SL.missed++;
            SL.state = 2;                                       // child f.i drew: from here on the generator is not where pass A assumed it, and the fan goes on as usual
        }
        V3 n = faceforward(f.d, f.info.norm);
        V3 b, cc;
        orthonormalSystem(n, b, cc);
        V3 reflected;
        for (;;) {
            double x, y;
            rng_unit_disc(tab, x, y);
            x *= sh.deflectionScaling;
            y *= sh.deflectionScaling;
            V3 nn = normalized(n + b * x + cc * y);
            reflected = reflect(f.d, nn);
            if (dot(reflected, n) > 0) break;
        }
        L.o = f.info.ip + n * 1e-6;
        L.d = reflected;
        L.depth = f.depth + 1;
        L.mode = WM_TRACE;
        return;
    }
    // WF_LAYERED, shading.cpp:357-367
    if (f.i == sh.layerCount) { L.ret = f.acc; L.sp--; L.mode = WM_RET; return; }
    const FRAY_RO DLayer& Ly = S.layers[sh.layerBegin + f.i];
    f.opacity = Ly.texture >= 0 ? texture_sample<ST>(S, Ly.texture, f.d, f.info, c) : ldc(Ly.opacity);
    L.o = f.o; L.d = f.d; L.depth = f.depth; L.info = f.info;
    L.shader = Ly.shader;
    L.mode = WM_SHADE;
}

// raytrace(ray) up to the shader call (main.cpp:246-283), for a lane standing at WM_TRACE
template <int ST>
FD void wl_trace_step(const DScene& S, WhittedLane& L, Cnt& c)
{
    if (L.depth > S.maxTraceDepth) { L.ret = c3(0, 0, 0); L.mode = WM_RET; return; }
    HitT<ST> h;
    closest_hit<ST>(S, L.o, L.d, h, c);
    if (h.node <= -2) { L.ret = light_color(S.lights[-2 - h.node]); L.mode = WM_RET; }
    else if (h.node < 0) { L.ret = environment<ST>(S, L.d, c); L.mode = WM_RET; }
    else {
        const FRAY_RO DNode& N = S.nodes[h.node];
        L.shader = N.shader;
        finalize_hit<ST>(S, h, L.o, L.d, S.shaders[L.shader].usesUV || N.bumpTex >= 0, L.info);
        apply_bump<ST>(S, h.node, L.info, c);
        L.mode = WM_SHADE;
    }
}

// Constant / Lambert / Phong ::shade, for a lane standing at WM_SHADE of such a shader
template <int ST, class G>
FD void wl_direct_step(const DScene& S, WhittedLane& L, G& tab, Cnt& c)
{
    const FRAY_RO DShader& sh = S.shaders[L.shader];
    const int kind = sh.kind;
    if (kind == 0) L.ret = ldc(sh.color);
    else L.ret = shade_direct<ST, G>(S, sh, L.d, L.info, tab, kind == 2, c);
    L.mode = WM_RET;
}
