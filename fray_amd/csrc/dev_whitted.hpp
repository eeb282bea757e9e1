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

template <int ST, class G>
FD C3 raytrace_full(const DScene& S, V3 o0, V3 d0, G& tab, Cnt& c, bool& overflow)
{
    WFrame stack[FRAY_WSTACK];
    int sp = 0;
    // registers of the machine
    V3 o = o0, d = d0;
    int depth = 0;
    HitInfo info;
    int shader = -1;
    C3 ret = c3(0, 0, 0);
    enum { TRACE, SHADE, RET, RESUME } mode = TRACE;
    bool done = false;
    C3 result = c3(0, 0, 0);
    // The lanes of a wave walk different trees, so left alone they drift out of phase and the two expensive steps -- the closest-hit
    // search (TRACE) and the direct-light loop with its visible() queries (SHADE of a Lambert / Phong surface) -- would each run for
    // the few lanes that happen to stand there.  Every round therefore first lets ALL lanes take their cheap steps (returns, loop
    // heads of glossy / layered activations, pushes of recursive shaders) until each stands at a TRACE, at a direct SHADE or at the
    // end; then one TRACE for all, then one direct SHADE for all.  A lane's own sequence of steps is unchanged.
    for (;;) {
        for (;;) {
            const bool cheap = !done && (mode == RET || mode == RESUME || (mode == SHADE && S.shaders[shader].kind >= 3));
            if (!__any(cheap)) break;
            if (!cheap) continue;
            if (mode == SHADE) {                               // shader->shade(ray, info) of a recursive shader: push its activation
                const FRAY_RO DShader& sh = S.shaders[shader];
                const int kind = sh.kind;
                if (sp >= FRAY_WSTACK) { overflow = true; done = true; continue; }
                WFrame& f = stack[sp];
                f.shader = shader; f.i = 0; f.count = 0; f.o = o; f.d = d; f.depth = depth; f.info = info;
                f.acc = c3(0, 0, 0); f.opacity = c3(0, 0, 0);
                if (kind == 3) {                               // Reflection::shade
                    V3 n = faceforward(d, info.norm);
                    if (sh.glossiness == 1.0) {
                        f.kind = WF_MULT; f.acc = ldc(sh.mult); sp++;
                        o = info.ip + n * 1e-6;
                        d = reflect(f.d, n);
                        depth = depth + 1;
                        mode = TRACE;
                    } else {
                        f.kind = WF_GLOSSY;
                        f.count = depth == 0 ? sh.numSamples : 3;  // LOW_GLOSSY_SAMPLES, constants.h:36
                        sp++;
                        mode = RESUME;
                    }
                    continue;
                }
                if (kind == 4) {                               // Refraction::shade
                    V3 n = faceforward(d, info.norm);
                    double myIor = dot(n, info.norm) > 0 ? 1.0 / sh.ior : sh.ior / 1.0;
                    V3 refr = refract(d, n, myIor);
                    if (refr.x == 0 && refr.y == 0 && refr.z == 0) { ret = c3(0, 0, 0); mode = RET; continue; }
                    f.kind = WF_MULT; f.acc = ldc(sh.mult); sp++;
                    o = info.ip - n * 1e-6;
                    d = refr;
                    depth = depth + 1;
                    mode = TRACE;
                    continue;
                }
                f.kind = WF_LAYERED;                           // Layered::shade
                sp++;
                mode = RESUME;
                continue;
            }
            if (mode == RET) {                                 // a call returned `ret`
                if (sp == 0) { result = ret; done = true; continue; }
                WFrame& f = stack[sp - 1];
                if (f.kind == WF_MULT) { ret = ret * f.acc; sp--; continue; }
                if (f.kind == WF_GLOSSY) {
                    f.acc = f.acc + ret * ldc(S.shaders[f.shader].mult);
                    f.i++;
                } else {
                    f.acc = ret * f.opacity + (c3(1, 1, 1) - f.opacity) * f.acc;
                    f.i++;
                }
                mode = RESUME;
                continue;
            }
            // RESUME: continue the loop of the activation on top of the stack
            WFrame& f = stack[sp - 1];
            const FRAY_RO DShader& sh = S.shaders[f.shader];
            if (f.kind == WF_GLOSSY) {                         // shading.cpp:172-204
                if (f.i == f.count) { ret = f.acc / (float)f.count; sp--; mode = RET; continue; }
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
                o = f.info.ip + n * 1e-6;
                d = reflected;
                depth = f.depth + 1;
                mode = TRACE;
                continue;
            }
            // WF_LAYERED, shading.cpp:357-367
            if (f.i == sh.layerCount) { ret = f.acc; sp--; mode = RET; continue; }
            const FRAY_RO DLayer& L = S.layers[sh.layerBegin + f.i];
            f.opacity = L.texture >= 0 ? texture_sample<ST>(S, L.texture, f.d, f.info, c) : ldc(L.opacity);
            o = f.o; d = f.d; depth = f.depth; info = f.info;
            shader = L.shader;
            mode = SHADE;
        }
        if (!__any(!done)) break;
        if (!done && mode == TRACE) {                          // raytrace(ray)
            if (depth > S.maxTraceDepth) { ret = c3(0, 0, 0); mode = RET; }
            else {
                HitRec h;
                closest_hit<ST>(S, o, d, h, c);
                if (h.node <= -2) { ret = light_color(S.lights[-2 - h.node]); mode = RET; }
                else if (h.node < 0) { ret = environment<ST>(S, d, c); mode = RET; }
                else {
                    const FRAY_RO DNode& N = S.nodes[h.node];
                    shader = N.shader;
                    finalize_hit<ST>(S, h, o, d, S.shaders[shader].usesUV || N.bumpTex >= 0, info);
                    apply_bump<ST>(S, h.node, info, c);
                    mode = SHADE;
                }
            }
        }
        if (!done && mode == SHADE && S.shaders[shader].kind <= 2) {   // Constant / Lambert / Phong ::shade
            const FRAY_RO DShader& sh = S.shaders[shader];
            const int kind = sh.kind;
            if (kind == 0) ret = ldc(sh.color);
            else ret = shade_direct<ST, G>(S, sh, d, info, tab, kind == 2, c);
            mode = RET;
        }
    }
    return result;
}
