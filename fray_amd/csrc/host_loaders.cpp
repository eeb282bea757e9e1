// OBJ / BMP loaders and the KD-tree builder (host side).
//
// The KD builder has to reproduce the reference tree leaf for leaf: which triangle wins an
// equal-distance tie depends on leaf contents and order ("Hard parts").
// So the box/triangle overlap predicate follows BBox::intersectTriangle (src/bbox.h:164-199)
// literally, including its use of the any-hit box test on un-normalised edge rays.
#include "host_scene.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace frayhost {

namespace {

struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double m) { return {a.x * m, a.y * m, a.z * m}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 unit(V3 a) { double m = 1.0 / sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return a * m; }
inline double comp(const V3& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
inline void setc(V3& v, int i, double d) { (i == 0 ? v.x : i == 1 ? v.y : v.z) = d; }
inline void put(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
inline V3 get(const double* p) { return {p[0], p[1], p[2]}; }

int to_int(const std::string& s)   // mesh.cpp:167-173
{
    int x;
    if (s.empty() || sscanf(s.c_str(), "%d", &x) != 1) return 0;
    return x;
}
double to_double(const std::string& s)   // mesh.cpp:175-181
{
    double x;
    if (s.empty() || sscanf(s.c_str(), "%lf", &x) != 1) return 0;
    return x;
}

// "4" -> {4,0,0}; "4//5" -> {4,0,5}; "3/4/5" -> v/uv/normal (mesh.cpp:183-192).  A trailing
// empty field after the last '/' counts as a field, like the reference's split().
void parse_corner(const std::string& s, int& v, int& uv, int& n)
{
    std::vector<std::string> items;
    size_t i = 0, l = s.size();
    while (i < l) {
        size_t j = i;
        while (j < l && s[j] != '/') j++;
        items.push_back(s.substr(i, j - i));
        i = j + 1;
        if (j == l - 1) items.push_back("");
    }
    v = items.size() >= 1 ? to_int(items[0]) : 0;
    uv = items.size() >= 2 ? to_int(items[1]) : 0;
    n = items.size() >= 3 ? to_int(items[2]) : 0;
}

std::vector<std::string> words(const char* s)
{
    std::vector<std::string> out;
    while (*s) {
        while (*s && isspace((unsigned char)*s)) s++;
        if (!*s) break;
        const char* e = s;
        while (*e && !isspace((unsigned char)*e)) e++;
        out.emplace_back(s, e - s);
        s = e;
    }
    return out;
}

// ---- BBox predicates (bbox.h) ------------------------------------------------------------------
struct Box { V3 lo, hi; };
struct EdgeRay { V3 start, dir, rdir; };

inline void prepare(EdgeRay& r)   // RRay::prepareForTracing, bbox.h:49-54
{
    r.rdir.x = fabs(r.dir.x) > 1e-12 ? 1.0 / r.dir.x : 1e12;
    r.rdir.y = fabs(r.dir.y) > 1e-12 ? 1.0 / r.dir.y : 1e12;
    r.rdir.z = fabs(r.dir.z) > 1e-12 ? 1.0 / r.dir.z : 1e12;
}
inline bool inside(const Box& b, V3 v)   // bbox.h:79-84
{
    return b.lo.x - 1e-6 <= v.x && v.x <= b.hi.x + 1e-6 && b.lo.y - 1e-6 <= v.y && v.y <= b.hi.y + 1e-6 &&
           b.lo.z - 1e-6 <= v.z && v.z <= b.hi.z + 1e-6;
}
bool box_any_hit(const Box& b, const EdgeRay& r)   // BBox::testIntersect, bbox.h:87-134
{
    if (inside(b, r.start)) return true;
    for (int d = 0; d < 3; d++) {
        double dd = comp(r.dir, d), sd = comp(r.start, d), lo = comp(b.lo, d), hi = comp(b.hi, d);
        if ((dd < 0 && sd < lo) || (dd > 0 && sd > hi)) return false;
        if (fabs(dd) < 1e-9) continue;
        double mul = comp(r.rdir, d);
        int u = d == 0 ? 1 : 0, v = d == 2 ? 1 : 2;
        for (int face = 0; face < 2; face++) {
            double dist = ((face == 0 ? lo : hi) - sd) * mul;
            if (dist < 0) break;            // the reference's `continue` of the dim loop skips the far face too
            double x = comp(r.start, u) + comp(r.dir, u) * dist;
            if (comp(b.lo, u) <= x && x <= comp(b.hi, u)) {
                double y = comp(r.start, v) + comp(r.dir, v) * dist;
                if (comp(b.lo, v) <= y && y <= comp(b.hi, v)) return true;
            }
        }
    }
    return false;
}
// Triangle::intersect, triangle.cpp:33-64 (the build-time variant, edges recomputed).
bool slow_triangle_hit(const EdgeRay& ray, V3 A, V3 B, V3 C, double& minDist)
{
    V3 AB = B - A, AC = C - A;
    V3 D = {-ray.dir.x, -ray.dir.y, -ray.dir.z};
    double Dcr = dot(cross(AB, AC), D);
    if (fabs(Dcr) < 1e-12) return false;
    double rDcr = 1 / Dcr;
    V3 H = ray.start - A;
    double gamma = dot(cross(AB, AC), H) * rDcr;
    if (gamma < 0 || gamma > minDist) return false;
    double l2 = dot(cross(H, AC), D) * rDcr;
    if (l2 < 0 || l2 > 1) return false;
    double l3 = dot(cross(AB, H), D) * rDcr;
    if (l3 < 0 || l3 > 1) return false;
    double l1 = 1 - (l2 + l3);
    if (l1 < 0) return false;
    minDist = gamma;
    return true;
}
inline double sign_of(double x) { return x > 0 ? +1 : -1; }   // util.h:35

bool box_overlaps_triangle(const Box& b, V3 A, V3 B, V3 C)   // bbox.h:164-199
{
    if (inside(b, A) || inside(b, B) || inside(b, C)) return true;
    EdgeRay ray;
    V3 t[3] = {A, B, C};
    for (int i = 0; i < 3; i++)
        for (int j = i + 1; j < 3; j++) {
            ray.start = t[i];
            ray.dir = t[j] - t[i];
            prepare(ray);
            if (box_any_hit(b, ray)) {
                ray.start = t[j];
                ray.dir = t[i] - t[j];
                prepare(ray);
                if (box_any_hit(b, ray)) return true;
            }
        }
    V3 AB = B - A, AC = C - A;
    V3 N = cross(AB, AC);
    double D = dot(A, N);
    for (int mask = 0; mask < 7; mask++)
        for (int j = 0; j < 3; j++) {
            if (mask & (1 << j)) continue;
            ray.start = {(mask & 1) ? b.hi.x : b.lo.x, (mask & 2) ? b.hi.y : b.lo.y, (mask & 4) ? b.hi.z : b.lo.z};
            V3 end = ray.start;
            setc(end, j, comp(b.hi, j));
            if (sign_of(dot(ray.start, N) - D) != sign_of(dot(end, N) - D)) {
                ray.dir = end - ray.start;
                double gamma = 1.0000001;
                if (slow_triangle_hit(ray, A, B, C, gamma)) return true;
            }
        }
    return false;
}

struct KDBuilder {
    MeshData& m;
    explicit KDBuilder(MeshData& mesh) : m(mesh) {}

    // Mesh::buildKD, mesh.cpp:320-355.  `self` is already allocated; children are allocated
    // pairwise when a node turns out to be an inner node (mesh.cpp:57).
    void build(int self, const std::vector<int>& tris, Box box, int depth)
    {
        if (depth > m.maxDepth) m.maxDepth = depth;
        if ((int)tris.size() <= 20 || depth > 64) {   // MAX_TRIANGLES_PER_LEAF, MAX_DEPTH (constants.h:38-39)
            frayhip_kdnode& n = m.kdnodes[self];
            n.axis = 3;
            n.child0 = -1;
            n.tri_begin = (int32_t)m.trirefs.size();
            n.tri_count = (int32_t)tris.size();
            n.split = 0;
            m.trirefs.insert(m.trirefs.end(), tris.begin(), tris.end());
            m.depthSum += depth;
            return;
        }
        int axis = depth % 3;
        double split = (comp(box.lo, axis) + comp(box.hi, axis)) * 0.5;   // mesh.cpp:315-318
        Box left = box, right = box;                                      // BBox::split, bbox.h:205-211
        setc(left.hi, axis, split);
        setc(right.lo, axis, split);
        std::vector<int> lt, rt;
        for (int ti : tris) {
            const frayhip_triangle& T = m.triangles[ti];
            V3 A = get(&m.vertices[3 * (size_t)T.v[0]]), B = get(&m.vertices[3 * (size_t)T.v[1]]), C = get(&m.vertices[3 * (size_t)T.v[2]]);
            if (box_overlaps_triangle(left, A, B, C)) lt.push_back(ti);
            if (box_overlaps_triangle(right, A, B, C)) rt.push_back(ti);
        }
        int c0 = (int)m.kdnodes.size();
        m.kdnodes.push_back(frayhip_kdnode{});
        m.kdnodes.push_back(frayhip_kdnode{});
        m.kdnodes[c0].parent = m.kdnodes[c0 + 1].parent = self;
        {
            frayhip_kdnode& n = m.kdnodes[self];
            n.axis = axis;
            n.child0 = c0;
            n.tri_begin = n.tri_count = 0;
            n.split = split;
        }
        build(c0, lt, left, depth + 1);
        build(c0 + 1, rt, right, depth + 1);
        m.depthSum += depth;
    }
};

}  // namespace

// Mesh::loadFromOBJ + prepareTriangles, mesh.cpp:203-313.
bool load_obj(const char* path, MeshData& mesh)
{
    FILE* f = fopen(path, "rt");
    if (!f) return false;
    // index 0 of every attribute array is a dummy so that 1-based OBJ indices (and the 0 that a
    // missing or unparsable index becomes) are always valid
    mesh.vertices.assign(3, 0.0);
    mesh.uvs.assign(3, 0.0);
    mesh.normals.assign(3, 0.0);
    std::vector<char> lineBuf(10000);      // per call: scenes may be parsed from different threads (include/frayhip.h)
    char* const line = lineBuf.data();
    while (fgets(line, (int)lineBuf.size(), f)) {
        if (line[0] == '#') continue;
        std::vector<std::string> tok = words(line);
        if (tok.empty()) continue;
        auto num = [&](size_t i) { return i < tok.size() ? to_double(tok[i]) : 0.0; };
        if (tok[0] == "v") { mesh.vertices.push_back(num(1)); mesh.vertices.push_back(num(2)); mesh.vertices.push_back(num(3)); }
        if (tok[0] == "vn") { mesh.normals.push_back(num(1)); mesh.normals.push_back(num(2)); mesh.normals.push_back(num(3)); }
        if (tok[0] == "vt") { mesh.uvs.push_back(num(1)); mesh.uvs.push_back(num(2)); mesh.uvs.push_back(0.0); }
        if (tok[0] == "f") {
            for (int i = 0; i < (int)tok.size() - 3; i++) {   // fan around the first corner
                frayhip_triangle T{};
                parse_corner(tok[1], T.v[0], T.t[0], T.n[0]);
                parse_corner(tok[2 + i], T.v[1], T.t[1], T.n[1]);
                parse_corner(tok[3 + i], T.v[2], T.t[2], T.n[2]);
                mesh.triangles.push_back(T);
            }
        }
    }
    fclose(f);
    if (mesh.normals.size() == 3) mesh.normals.clear();
    // Indices the file does not back (beyond the arrays, or negative) read the dummy element 0; the
    // reference would index out of bounds there.
    {
        const int nv = (int)(mesh.vertices.size() / 3), nn = (int)(mesh.normals.size() / 3), nt = (int)(mesh.uvs.size() / 3);
        for (auto& t : mesh.triangles)
            for (int k = 0; k < 3; k++) {
                if (t.v[k] < 0 || t.v[k] >= nv) t.v[k] = 0;
                if (t.n[k] < 0 || t.n[k] >= nn) t.n[k] = 0;
                if (t.t[k] < 0 || t.t[k] >= nt) t.t[k] = 0;
            }
    }

    const bool haveUV = !mesh.uvs.empty(), haveN = !mesh.normals.empty();
    for (auto& t : mesh.triangles) {
        V3 A = get(&mesh.vertices[3 * (size_t)t.v[0]]), B = get(&mesh.vertices[3 * (size_t)t.v[1]]), C = get(&mesh.vertices[3 * (size_t)t.v[2]]);
        V3 AB = B - A, AC = C - A;
        V3 N = cross(AB, AC);
        put(t.AB, AB); put(t.AC, AC); put(t.ABcrossAC, N);
        put(t.gnormal, unit(N));
        if (haveUV && haveN) {
            V3 tA = get(&mesh.uvs[3 * (size_t)t.t[0]]), tB = get(&mesh.uvs[3 * (size_t)t.t[1]]), tC = get(&mesh.uvs[3 * (size_t)t.t[2]]);
            V3 tAB = tB - tA, tAC = tC - tA;
            // solve p*tAB + q*tAC = (1,0) and (0,1)   (solve2D, mesh.cpp:261-270)
            double m00 = tAB.x, m01 = tAC.x, m10 = tAB.y, m11 = tAC.y;
            double Dcr = m00 * m11 - m10 * m01;
            double px = (1.0 * m11 - 0.0 * m01) / Dcr, qx = (m00 * 0.0 - m10 * 1.0) / Dcr;
            double py = (0.0 * m11 - 1.0 * m01) / Dcr, qy = (m00 * 1.0 - m10 * 0.0) / Dcr;
            put(t.dNdx, unit(AB * px + AC * qx));
            put(t.dNdy, unit(AB * py + AC * qy));
        } else {
            put(t.dNdx, V3{0, 0, 0});
            put(t.dNdy, V3{0, 0, 0});
        }
    }
    return true;
}

// Mesh::computeBoundingGeometry, mesh.cpp:74-94.
void build_kd(MeshData& mesh)
{
    Box box{{+1e99, +1e99, +1e99}, {-1e99, -1e99, -1e99}};
    for (size_t i = 0; i + 2 < mesh.vertices.size(); i += 3) {   // includes the dummy vertex, as the reference does
        V3 v = get(&mesh.vertices[i]);
        box.lo.x = std::min(box.lo.x, v.x); box.hi.x = std::max(box.hi.x, v.x);
        box.lo.y = std::min(box.lo.y, v.y); box.hi.y = std::max(box.hi.y, v.y);
        box.lo.z = std::min(box.lo.z, v.z); box.hi.z = std::max(box.hi.z, v.z);
    }
    put(mesh.bbox_min, box.lo);
    put(mesh.bbox_max, box.hi);
    mesh.kdnodes.clear();
    mesh.trirefs.clear();
    mesh.maxDepth = mesh.depthSum = 0;
    if (mesh.useKD && mesh.triangles.size() > 20) {
        std::vector<int> all(mesh.triangles.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
        frayhip_kdnode root{};
        root.parent = -1;
        mesh.kdnodes.push_back(root);
        KDBuilder kb(mesh);
        kb.build(0, all, box, 0);
    }
}

// Bitmap::loadBMP, bitmap.cpp:117-195: 8-bit palettised (palette entries are 0xRRGGBB words),
// 24- and 32-bit BGR(A), rows bottom-up and padded to 4 bytes.
bool load_bmp(const char* path, Image& img, std::string& err)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = "cannot open file"; return false; }
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{fp};
    unsigned char hdr[54];
    if (fread(hdr, 1, 54, fp) != 54) { err = "short header"; return false; }
    auto u16 = [&](int o) { return (unsigned)hdr[o] | ((unsigned)hdr[o + 1] << 8); };
    auto i32 = [&](int o) { return (int32_t)((uint32_t)hdr[o] | ((uint32_t)hdr[o + 1] << 8) | ((uint32_t)hdr[o + 2] << 16) | ((uint32_t)hdr[o + 3] << 24)); };
    if (u16(0) != 19778) { err = "not a BMP file"; return false; }
    int imgOffset = i32(10), w = i32(18), h = i32(22);
    unsigned planes = u16(26), bpp = u16(28);
    int colors = i32(46);
    if (!(bpp == 8 || bpp == 24 || bpp == 32)) { err = "unsupported bpp"; return false; }
    if (planes != 1) { err = "multichannel bmp"; return false; }
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768) { err = "bad dimensions"; return false; }
    {   // a truncated or corrupt file must not drive a huge allocation
        long here = ftell(fp);
        fseek(fp, 0, SEEK_END);
        long size = ftell(fp);
        fseek(fp, here, SEEK_SET);
        long long rowBytes = ((long long)w * (bpp / 8) + 3) / 4 * 4;
        // every row the header promises must be in the file (the last row may lack its padding): a 100 KB file that claims
        // 32768 x 32768 pixels must not drive a 12.9 GB allocation
        if (imgOffset < 54 || imgOffset > size || rowBytes * (long long)h > (long long)(size - imgOffset) + 4) { err = "file is shorter than its header claims"; return false; }
    }
    float palette[256][3] = {};
    int toread = 0;
    if (bpp <= 8) {
        if (colors < 0) { err = "bad palette size"; return false; }
        toread = colors ? colors : (1 << bpp);
        if (toread > 256) { err = "bad palette size"; return false; }
        for (int i = 0; i < toread; i++) {
            unsigned char q[4];
            if (fread(q, 1, 4, fp) != 4) { err = "short palette"; return false; }
            uint32_t t = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            palette[i][2] = (t & 0xff) / 255.0f;            // Color(unsigned), color.h:50-55
            palette[i][1] = ((t >> 8) & 0xff) / 255.0f;
            palette[i][0] = ((t >> 16) & 0xff) / 255.0f;
        }
    }
    fseek(fp, imgOffset - (54 + toread * 4), SEEK_CUR);
    int k = bpp / 8;
    int rowsz = w * k;
    if (rowsz % 4 != 0) rowsz = (rowsz / 4 + 1) * 4;
    std::vector<unsigned char> row(rowsz);
    img.w = w; img.h = h;
    img.rgb.assign((size_t)w * h * 3, 0.0f);
    for (int j = h - 1; j >= 0; j--) {
        if (fread(row.data(), 1, rowsz, fp) == 0) { err = "short read"; img = Image(); return false; }
        for (int i = 0; i < w; i++) {
            float* o = &img.rgb[((size_t)j * w + i) * 3];
            if (bpp > 8) {
                o[0] = row[i * k + 2] / 255.0f; o[1] = row[i * k + 1] / 255.0f; o[2] = row[i * k] / 255.0f;
            } else {
                const float* p = palette[row[i * k]];
                o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            }
        }
    }
    return true;
}

}  // namespace frayhost
