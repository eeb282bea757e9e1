// Device restatement of the reference's random numbers: std::mt19937 driven through libstdc++ 11's
// uniform_real_distribution<float/double> and uniform_int_distribution<int>
// (src/random_generator.cpp:41-80; /usr/include/c++/11/bits/random.tcc:3346-3380,
// bits/uniform_int_dist.h:243-318 -- third-party arithmetic the reference does not vendor).
//
// Under the RNG contract every camera sample starts from a freshly seeded
// generator and draws at most ~110 words, so only the first 227 outputs of mt19937 are ever
// needed.  Output j (j < 227) of the first twist depends only on the seeding recurrence
//     x[0] = seed,  x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i
// at i = j, j+1 and j+397, so a stream is three registers {j, x[j], x[j+397]} and no 2.5 KB
// state array.  Drawing more than 227 words is reported through DStats.rngOverflow (the render
// call then fails instead of returning numbers that differ from the reference).
#pragma once
#include "dev_math.hpp"
#include "dev_trig.hpp"

struct Mt { uint32_t j, a, b; FD uint32_t next(); };

// One step of the seeding recurrence x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i (libstdc++ bits/random.tcc, mersenne_twister_engine::seed).
// On the device the multiply and the add are ONE instruction, v_mad_u64_u32 (the low word of y * C + i): measured 1.43 ms against 2.01 ms for the
// v_mul_lo_u32 + v_add_u32 pair on a k_seed-shaped loop (tools/ubench), and every random word costs two of these steps.
FD uint32_t mt_lcg(uint32_t x, uint32_t i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t y = x ^ (x >> 30);
    unsigned long long r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(y), "s"(1812433253u), "v"((unsigned long long)i) : "vcc");
    return (uint32_t)r;
#else
    return 1812433253u * (x ^ (x >> 30)) + i;
#endif
}

FD Mt mt_seed(uint32_t s)
{
    Mt r;
    r.j = 0;
    r.a = s;
    uint32_t b = s;
    for (uint32_t i = 1; i <= 397; i++) b = mt_lcg(b, i);
    r.b = b;
    return r;
}

// Same, with x[397] already known (k_seed computes it for a whole batch, four independent chains
// per lane, because the 397-step recurrence is one long dependency chain).
FD Mt mt_seed_with(uint32_t s, uint32_t x397)
{
    Mt r;
    r.j = 0;
    r.a = s;
    r.b = x397;
    return r;
}

FD uint32_t mt_next(Mt& r)
{
    uint32_t a1 = mt_lcg(r.a, r.j + 1);
    uint32_t y = (r.a & 0x80000000u) | (a1 & 0x7fffffffu);
    uint32_t v = r.b ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r.a = a1;
    r.b = mt_lcg(r.b, r.j + 398);
    r.j++;
    v ^= v >> 11;
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= v >> 18;
    return v;
}

FD uint32_t Mt::next() { return mt_next(*this); }

FD void mt_skip(Mt& r, int n)   // draws whose values the reference discards
{
    for (int i = 0; i < n; i++) {
        r.a = mt_lcg(r.a, r.j + 1);
        r.b = mt_lcg(r.b, r.j + 398);
        r.j++;
    }
}

// Random::randfloat: generate_canonical<float, 24> -- one word, float(u) / 2^32, clamped below 1.
template <class G>
FD float rng_float(G& r)
{
    float f = (float)r.next();
    f = f / 4294967296.0f;
    if (f >= 1.0f) f = 0x1.fffffep-1f;   // nextafter(1.0f, 0.0f)
    return f;
}

// Random::randdouble: generate_canonical<double, 53> -- two words, (u0 + u1 * 2^32) / 2^64.
template <class G>
FD double rng_double(G& r)
{
    double sum = (double)r.next();
    sum = sum + (double)r.next() * 4294967296.0;
    double v = sum / 18446744073709551616.0;
    if (v >= 1.0) v = 0x1.fffffffffffffp-1;   // nextafter(1.0, 0.0)
    return v;
}

// Random::randint(0, hi): Lemire's nearly divisionless method on a 32-bit generator
// (uniform_int_distribution::_S_nd<uint64_t>).
template <class G>
FD int rng_int0(G& r, int hi)
{
    uint32_t range = (uint32_t)hi + 1u;
    uint64_t product = (uint64_t)r.next() * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)r.next() * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (int)(product >> 32);
}

// Random::unitDiscSample, random_generator.cpp:71-80
template <class G>
FD void rng_unit_disc(G& r, double& x, double& y)
{
    double angle = rng_double(r) * 2 * FRAY_PI;
    double rad = sqrt(rng_double(r));
    double sa, ca;
    fray_sincos(angle, &sa, &ca);
    x = sa * rad;
    y = ca * rad;
}

// Generator without the 227-word limit, for the Whitted kernel (a Lambert hit under a 15x15
// RectLight alone draws 450 words; glossy reflections draw an unbounded number).  The first 227
// outputs come from the register recurrence above; past that the full 624-word state is
// materialised in a per-thread slice of a global workspace (word k of thread t at
// st[k * stride + t], so lanes in step touch consecutive addresses) and advanced with the
// standard in-place twist.
struct MtLong {
    Mt r;
    uint32_t seed;
    uint32_t* st;       // this thread's column of the workspace
    uint32_t stride;
    int idx;            // next output position in the materialised state, -1 = not materialised

    FD void reseed(uint32_t s) { r = mt_seed(s); seed = s; idx = -1; }
    FD void reseed_with(uint32_t s, uint32_t x397) { r = mt_seed_with(s, x397); seed = s; idx = -1; }
    FD uint32_t& w(int k) { return st[(size_t)k * stride]; }
    FD static uint32_t tw(uint32_t u, uint32_t v)
    {
        uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    FD void twist()
    {
        for (int k = 0; k < 227; k++) w(k) = w(k + 397) ^ tw(w(k), w(k + 1));
        for (int k = 227; k < 623; k++) w(k) = w(k - 227) ^ tw(w(k), w(k + 1));
        w(623) = w(396) ^ tw(w(623), w(0));
    }
    // The full state after the first twist, in the thread's workspace column.  A function of the seed alone, so it may be made before the
    // register stream has reached its 227th word (k_whitted does that for a whole wave at once: made lane by lane, as each lane's glossy
    // rejection loop happens to cross 227 words, the 1 248-step dependent chain below ran up to 64 times per 8x8 tile -- 22 ms for one tile
    // at the horizon of hw9/dragon.fray).
    FD void materialise()
    {
        uint32_t x = seed;
        w(0) = x;
        for (uint32_t i = 1; i < 624; i++) { x = mt_lcg(x, i); w((int)i) = x; }
        twist();
        idx = 227;
    }
    FD uint32_t next()
    {
        if (r.j < 227) return mt_next(r);
        if (idx < 0) materialise();
        if (idx >= 624) { twist(); idx = 0; }
        uint32_t v = w(idx++);
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
};

// Generator of a PATH whose draws outlive one kernel and may pass 227 words (path tracing with maxTraceDepth >= 20: a Lambert bounce
// draws ten).  While j < 227 it is the three-register stream above; the draw after that materialises the 624-word state in the path's
// column of a global workspace (word k at col[k * stride]) and from then on the cursor is just the number of words drawn, with
// FRAY_MT_MATERIALISED set: position = count % 624, the standard twist whenever the position wraps.  {j, a, b} is what the path queue
// stores between bounces either way.
#define FRAY_MT_MATERIALISED 0x80000000u
struct MtPath {
    Mt r;
    uint32_t seed;      // the stream's seed (needed once, to materialise)
    uint32_t* col;
    size_t stride;
    FD uint32_t& w(int k) { return col[(size_t)k * stride]; }
    FD void twist()
    {
        for (int k = 0; k < 227; k++) w(k) = w(k + 397) ^ MtLong::tw(w(k), w(k + 1));
        for (int k = 227; k < 623; k++) w(k) = w(k - 227) ^ MtLong::tw(w(k), w(k + 1));
        w(623) = w(396) ^ MtLong::tw(w(623), w(0));
    }
    FD uint32_t next()
    {
        if (!(r.j & FRAY_MT_MATERIALISED)) {
            if (r.j < 227) return mt_next(r);
            uint32_t x = seed;                  // materialise: seeding recurrence, then the first twist
            w(0) = x;
            for (uint32_t i = 1; i < 624; i++) { x = mt_lcg(x, i); w((int)i) = x; }
            twist();
            r.j = 227u | FRAY_MT_MATERIALISED;
        }
        const uint32_t n = r.j & ~FRAY_MT_MATERIALISED;
        const int idx = (int)(n % 624u);
        if (idx == 0) twist();                  // n >= 624 here: the stream starts at word 227 of the first generation
        uint32_t v = w(idx);
        r.j = (n + 1u) | FRAY_MT_MATERIALISED;
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
};
FD void mt_skip(MtPath& g, int n) { for (int i = 0; i < n; i++) (void)g.next(); }

// Per-(pixel, sample) seed of the RNG contract; the oracle applies the same function
// (oracle/fray_oracle.cpp sample_seed).
FD uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
FD uint32_t sample_seed(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = fmix32(seed ^ (pixel * 0x9e3779b1u));
    return fmix32(h ^ (sample * 0x85ebca77u) ^ 0x27d4eb2fu);
}
