// raytracer.hpp — C++ host surface of the MI355X render path.
//
// Same public names and float semantics as the reference's host classes so that code written against
// cgringmuth/raytracer (Vec3f, Ray, Mat3d, Color, Material, Sphere, Plane, Triangle, Model, Light, Scene,
// Camera, render()) keeps working; the difference is below render(): instead of tracing on CPU threads it
// flattens Scene + Camera into the plain structs of include/raytracer_amd.h and calls the C-ABI, which runs
// hand-written HIP kernels on gfx950. Nothing here intersects or shades anything.
//
// Every arithmetic statement that feeds the device (normalisation, rotation, bounding volume, PLY normals)
// follows the reference's operation order and float/double mix, cited per function as (ref file:line),
// because the parity bar is bit-exact host data.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "raytracer_amd.h"

typedef float Float;                    // ref src/common.h:33
constexpr Float EPS{0.000001};          // ref src/common.h:36

struct Mat3d;

struct Point2D {                        // ref src/container.h:17-30
    Float x_, y_;
    Point2D() : x_(0), y_(0) {}
    Point2D(Float x, Float y) : x_(x), y_(y) {}
    Float x() const { return x_; }
    Float y() const { return y_; }
};

// 3-vector of Float. Only the float instantiation the reference uses (Vec3f) is provided.
struct Vec3f {
    Float x, y, z;
    Vec3f() : x(0), y(0), z(0) {}
    Vec3f(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}

    Float& operator[](size_t i) { return i == 1 ? y : (i == 2 ? z : x); }
    const Float& operator[](size_t i) const { return i == 1 ? y : (i == 2 ? z : x); }

    Vec3f operator-() const { return Vec3f(-x, -y, -z); }
    Vec3f& operator+=(const Vec3f& r) { x += r.x; y += r.y; z += r.z; return *this; }
    Vec3f& operator-=(const Vec3f& r) { x -= r.x; y -= r.y; z -= r.z; return *this; }
    Vec3f& operator+=(Float d) { x += d; y += d; z += d; return *this; }
    Vec3f& operator*=(Float s) { x *= s; y *= s; z *= s; return *this; }
    // Division multiplies by a reciprocal taken in DOUBLE and narrowed (ref src/container.h:72-75).
    Vec3f& operator/=(Float v) { const Float inv = Float(1.0 / double(v)); return *this *= inv; }
    inline Vec3f& operator*=(const Mat3d& m);

    // (x*x + y*y) + z*z in float (ref src/container.h:134-138)
    Float dotProduct(const Vec3f& o) const { return (x * o.x) + (y * o.y) + (z * o.z); }
    // sqrt in double of the float dot, narrowed (ref src/container.h:140-142)
    Float length() const { return Float(std::sqrt(double(dotProduct(*this)))); }
    // multiply by the FLOAT quotient 1/length (ref src/container.h:144-146)
    Vec3f& normalize() { return *this *= (Float(1) / length()); }
    Vec3f cross_product(const Vec3f& v) const {     // ref src/container.h:151-158
        return Vec3f(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x);
    }
    // unit vector start->end (ref src/container.h:44-48) and unit normal of a x b (:50-54)
    static Vec3f getVec(const Vec3f& start, const Vec3f& end) { Vec3f v = end; v -= start; v.normalize(); return v; }
    static Vec3f getNormal(const Vec3f& a, const Vec3f& b) { Vec3f v = a.cross_product(b); v.normalize(); return v; }
};
inline Vec3f operator+(Vec3f a, const Vec3f& b) { return a += b; }
inline Vec3f operator-(Vec3f a, const Vec3f& b) { return a -= b; }
inline Vec3f operator*(Vec3f a, Float s) { return a *= s; }
inline Vec3f operator*(Float s, Vec3f a) { return a *= s; }
inline Vec3f operator/(Vec3f a, Float s) { return a /= s; }
inline Vec3f cross_product(const Vec3f& a, const Vec3f& b) { return a.cross_product(b); }
inline Float dotProduct(const Vec3f& a, const Vec3f& b) { return a.dotProduct(b); }

struct Ray {                            // ref src/container.h:247-274
    Vec3f origin_, direction_;
    Ray(const Vec3f& o, const Vec3f& d) : origin_(o), direction_(d) {}
    Vec3f getPoint(Float dist) const { return origin_ + direction_ * dist; }
    const Vec3f& origin() const { return origin_; }
    const Vec3f& direction() const { return direction_; }
};

// Row-major 3x3; M*v convention (ref src/container.h:276-353). at(x,y) = column x of row y.
struct Mat3d {
    Float v[9];
    Mat3d(Float a, Float b, Float c, Float d, Float e, Float f, Float g, Float h, Float i) {
        const Float t[9] = {a, b, c, d, e, f, g, h, i};
        std::memcpy(v, t, sizeof v);
    }
    explicit Mat3d(const Float* p) { std::memcpy(v, p, sizeof v); }
    Float& at(unsigned x, unsigned y) { return v[y * 3 + x]; }
    Float at(unsigned x, unsigned y) const { return v[y * 3 + x]; }
    Float operator[](size_t i) const { return v[i]; }

    // this = this * rhs, each element a left-to-right 3-term float sum (ref src/container.h:307-319)
    Mat3d& operator*=(const Mat3d& r) {
        const Mat3d t(*this);
        for (unsigned row = 0; row < 3; ++row)
            for (unsigned col = 0; col < 3; ++col)
                v[row * 3 + col] = t.v[row * 3 + 0] * r.v[0 * 3 + col] + t.v[row * 3 + 1] * r.v[1 * 3 + col] +
                                   t.v[row * 3 + 2] * r.v[2 * 3 + col];
        return *this;
    }
    // cos/sin in double, narrowed (ref src/container.h:324-352)
    static Mat3d rotationX(Float phi) {
        const Float c = Float(std::cos(double(phi))), s = Float(std::sin(double(phi)));
        return Mat3d(1, 0, 0, 0, c, -s, 0, s, c);
    }
    static Mat3d rotationY(Float phi) {
        const Float c = Float(std::cos(double(phi))), s = Float(std::sin(double(phi)));
        return Mat3d(c, 0, s, 0, 1, 0, -s, 0, c);
    }
    static Mat3d rotationZ(Float phi) {
        const Float c = Float(std::cos(double(phi))), s = Float(std::sin(double(phi)));
        return Mat3d(c, -s, 0, s, c, 0, 0, 0, 1);
    }
    // (Rz*Ry)*Rx (ref src/container.cpp:13-15)
    static Mat3d rotation(Float phiX, Float phiY, Float phiZ) {
        Mat3d m = rotationZ(phiZ);
        m *= rotationY(phiY);
        m *= rotationX(phiX);
        return m;
    }
    static Mat3d rotation(const Vec3f& r) { return rotation(r.x, r.y, r.z); }
};
inline Mat3d operator*(Mat3d a, const Mat3d& b) { return a *= b; }
// v = M*v, rows summed left to right (ref src/container.h:98-106)
inline Vec3f& Vec3f::operator*=(const Mat3d& m) {
    const Float tx = x * m.v[0] + y * m.v[1] + z * m.v[2];
    const Float ty = x * m.v[3] + y * m.v[4] + z * m.v[5];
    const Float tz = x * m.v[6] + y * m.v[7] + z * m.v[8];
    x = tx; y = ty; z = tz;
    return *this;
}
inline Vec3f operator*(Vec3f a, const Mat3d& m) { return a *= m; }

inline Float deg2rad(Float ang) { return Float(double(ang) * M_PI / 180); }   // ref src/helper.cpp:13-15

struct Color {                          // ref src/color.h:16-127
    Float r, g, b;
    Color() : r(0), g(0), b(0) {}
    explicit Color(Float v) : r(v), g(v), b(v) {}
    explicit Color(const Vec3f& v) : r(v.x), g(v.y), b(v.z) {}
    Color(Float r_, Float g_, Float b_) : r(r_), g(g_), b(b_) {}
    Color& operator*=(Float d) { r *= d; g *= d; b *= d; return *this; }
    Color& operator*=(const Color& c) { r *= c.r; g *= c.g; b *= c.b; return *this; }
    Color& operator+=(const Color& c) { r += c.r; g += c.g; b += c.b; return *this; }
    Color& operator/=(Float d) { const Float inv = Float(1) / d; return *this *= inv; }   // ref src/color.h:27-30
    Color& clamp(Float lo, Float hi) {                                                      // ref src/common.h:47-52
        Float* p[3] = {&r, &g, &b};
        for (Float* q : p) { *q = *q < lo ? lo : *q; *q = *q > hi ? hi : *q; }
        return *this;
    }
    static Color white() { return Color(1, 1, 1); }
    static Color black() { return Color(); }
    static Color red() { return Color(1, 0, 0); }
    static Color green() { return Color(0, 1, 0); }
    static Color blue() { return Color(0, 0, 1); }
    static Color gray() { return Color(0.5f); }
    static Color light_gray() { return Color(0.75f); }
    static Color yellow() { return Color(1, 1, 0); }
    static Color glass() { return Color(0.788f, 0.858f, 0.862f); }
};
inline Color operator+(Color a, const Color& b) { return a += b; }
inline Color operator*(Color a, Float d) { return a *= d; }
inline Color operator*(Float d, Color a) { return a *= d; }
inline Color operator*(Color a, const Color& b) { return a *= b; }
inline Color operator/(Color a, Float d) { return a /= d; }

struct Material {                       // ref src/material.h:13-47 (same defaults)
    Color color;
    Float ka, kd, ks, specRefExp;
    bool reflective;
    Float kr;
    bool refractive;
    Float refractiveIdx, kt;
    explicit Material(Float ka_ = 0.2f, Float kd_ = 0.7f, Float ks_ = 0.2f, Float kr_ = 0, Float kt_ = 0,
                      Float specRefExp_ = 8, Float refractiveIdx_ = 0, bool reflective_ = false,
                      bool refractive_ = false)
        : Material(Color(), ka_, kd_, ks_, kr_, kt_, specRefExp_, refractiveIdx_, reflective_, refractive_) {}
    explicit Material(const Color& c, Float ka_ = 0.2f, Float kd_ = 0.7f, Float ks_ = 0.2f, Float kr_ = 0,
                      Float kt_ = 0, Float specRefExp_ = 8, Float refractiveIdx_ = 0, bool reflective_ = false,
                      bool refractive_ = false)
        : color(c), ka(ka_), kd(kd_), ks(ks_), specRefExp(specRefExp_), reflective(reflective_), kr(kr_),
          refractive(refractive_), refractiveIdx(refractiveIdx_), kt(kt_) {}
    rt_material flat() const {
        rt_material m;
        m.color[0] = color.r; m.color[1] = color.g; m.color[2] = color.b;
        m.ka = ka; m.kd = kd; m.ks = ks; m.kr = kr; m.kt = kt; m.specRefExp = specRefExp;
        m.refractiveIdx = refractiveIdx; m.reflective = reflective; m.refractive = refractive;
        return m;
    }
};

struct Light {                          // ref src/scene.h:15-27
    Vec3f pos;
    Color color;
    Light(const Vec3f& p, const Color& c) : pos(p), color(c) {}
};

// Collects the SoA scene the C-ABI takes. Object order = call order (ref src/raytracer.cpp:292-300 tie-break).
class FlatSceneBuilder {
public:
    std::vector<rt_object> objects;
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_model> models;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<float> v0, v1, v2, n0, n1, n2;

    uint32_t addMaterial(const Material& m) { materials.push_back(m.flat()); return uint32_t(materials.size() - 1); }
    void addObject(int type, uint32_t index, uint32_t material) {
        rt_object o; o.type = type; o.index = index; o.material = material; o.reserved = 0;
        objects.push_back(o);
    }
    rt_flat_scene view() const {
        rt_flat_scene s;
        s.objects = objects.data(); s.n_objects = uint32_t(objects.size());
        s.spheres = spheres.data(); s.n_spheres = uint32_t(spheres.size());
        s.planes = planes.data(); s.n_planes = uint32_t(planes.size());
        s.models = models.data(); s.n_models = uint32_t(models.size());
        s.materials = materials.data(); s.n_materials = uint32_t(materials.size());
        s.lights = lights.data(); s.n_lights = uint32_t(lights.size());
        s.v0 = v0.data(); s.v1 = v1.data(); s.v2 = v2.data();
        s.n0 = n0.data(); s.n1 = n1.data(); s.n2 = n2.data();
        s.n_faces = uint32_t(v0.size() / 3);
        return s;
    }
};

// Base of everything renderable (ref src/primitives.h:18-28).
struct Primitive {
    Material material;
    explicit Primitive(const Color& c) : material(c) {}
    explicit Primitive(Material m = Material()) : material(std::move(m)) {}
    virtual void flatten(FlatSceneBuilder& b) const = 0;
    // ref src/primitives.h:23: ONE ray against THIS primitive - on the device, like every intersection here: the primitive
    // is flattened into a one-object scene and rt_intersect is asked (a scene build and a launch per call: for the odd
    // caller of the virtual; the renderer and rt_intersect on a whole Scene are the batch forms). Returns the reference's
    // bool; `dist` and `normal` are what its intersect() computes on a hit and are left as they were on a miss.
    // Throws on device errors.
    bool intersect(const Ray& ray, Float& dist, Vec3f& normal) const {
        FlatSceneBuilder fb;
        flatten(fb);
        const rt_flat_scene fs = fb.view();
        rt_scene* dev = nullptr;
        if (rt_scene_create(&fs, 0, &dev) != RT_OK) throw std::runtime_error(rt_last_error());
        const float r6[6] = {ray.origin().x, ray.origin().y, ray.origin().z, ray.direction().x, ray.direction().y, ray.direction().z};
        int32_t hit = -1;
        float d = 0, n[3] = {0, 0, 0};
        const int rc = rt_intersect(dev, 0, r6, 1, &hit, &d, n);
        const std::string err = rc == RT_OK ? std::string() : std::string(rt_last_error());
        rt_scene_destroy(dev);
        if (rc != RT_OK) throw std::runtime_error(err);
        if (hit < 0) return false;
        dist = d; normal = Vec3f(n[0], n[1], n[2]);
        return true;
    }
    virtual Vec3f getNormal(const Vec3f& vec) const = 0;      // ref src/primitives.h:24
    virtual ~Primitive() = default;
};

struct Plane : Primitive {              // ref src/primitives.h:31-62: ax+by+cz+d=0, normal normalised in the ctor
    Float a, b, c, d;
    Plane(Float a_, Float b_, Float c_, Float d_, const Material& m) : Primitive(m), d(d_) { setNormal(Vec3f(a_, b_, c_)); }
    Plane(Float a_, Float b_, Float c_, Float d_, const Color& col) : Plane(a_, b_, c_, d_, Material(col)) {}
    Plane(Vec3f normal, Float dist, const Material& m) : Primitive(m), d(dist) { setNormal(normal); }
    Plane(const Vec3f& normal, Float dist, const Color& col) : Plane(normal, dist, Material(col)) {}
    Vec3f getNormal(const Vec3f&) const override { return Vec3f(a, b, c); }                       // ref src/primitives.cpp:27-29
    void flatten(FlatSceneBuilder& fb) const override {
        rt_plane p; p.a = a; p.b = b; p.c = c; p.d = d;
        fb.planes.push_back(p);
        fb.addObject(RT_OBJ_PLANE, uint32_t(fb.planes.size() - 1), fb.addMaterial(material));
    }
private:
    void setNormal(Vec3f n) { n.normalize(); a = n.x; b = n.y; c = n.z; }
};

struct Sphere : Primitive {             // ref src/primitives.h:107-119
    Vec3f center;
    Float radius;
    Sphere(const Vec3f& c, Float r, const Color& col) : Primitive(col), center(c), radius(r) {}
    Sphere(const Vec3f& c, Float r, const Material& m) : Primitive(m), center(c), radius(r) {}
    Sphere() : Sphere(Vec3f(0, 0, 0), 1, Material()) {}
    Vec3f getNormal(const Vec3f& P) const override { Vec3f n(P - center); n /= radius; return n; }   // ref src/primitives.cpp:198-203
    void flatten(FlatSceneBuilder& fb) const override {
        rt_sphere s; s.center[0] = center.x; s.center[1] = center.y; s.center[2] = center.z; s.radius = radius;
        fb.spheres.push_back(s);
        fb.addObject(RT_OBJ_SPHERE, uint32_t(fb.spheres.size() - 1), fb.addMaterial(material));
    }
};

struct Triangle : Primitive {           // ref src/primitives.h:64-102
    Vec3f v0, v1, v2, n0, n1, n2, faceNormal;
    Triangle(const Vec3f& a, const Vec3f& b, const Vec3f& c, const Material& m = Material())
        : Primitive(m), v0(a), v1(b), v2(c) { n0 = calcNormal(); n1 = n0; n2 = n0; faceNormal = n0; }
    Triangle(const Vec3f& a, const Vec3f& b, const Vec3f& c, const Color& col) : Triangle(a, b, c, Material(col)) {}
    Triangle(const Vec3f& a, const Vec3f& b, const Vec3f& c, const Vec3f& na, const Vec3f& nb, const Vec3f& nc,
             const Material& m = Material())
        : Primitive(m), v0(a), v1(b), v2(c), n0(na), n1(nb), n2(nc) {
        // the reference's ctor calls calcNormal() for faceNormal, which also OVERWRITES n0 (ref src/primitives.h:75-78,
        // src/primitives.cpp:132-139); member order makes n0 the given value again only if initialised later - it is not:
        // n0 is initialised before faceNormal, so calcNormal()'s write to n0 wins. Kept.
        faceNormal = calcNormal();
    }
    // normalize(cross(normalize(v1-v0), normalize(v2-v0))), stored into n0 as a side effect (ref src/primitives.cpp:132-139)
    Vec3f calcNormal() {
        n0 = cross_product(Vec3f::getVec(v0, v1), Vec3f::getVec(v0, v2));
        n0.normalize();
        return n0;
    }
    Vec3f getNormal(const Vec3f&) const override { return faceNormal; }          // ref src/primitives.cpp:128-130
    void scale(Float s) { v0 *= s; v1 *= s; v2 *= s; }                          // ref src/primitives.cpp:141-145
    Triangle& operator+=(const Vec3f& t) { v0 += t; v1 += t; v2 += t; return *this; }   // :154-159
    Triangle& operator*=(const Mat3d& m) { v0 *= m; v1 *= m; v2 *= m; n0 *= m; n1 *= m; n2 *= m; return *this; }  // :161-169
    void appendTo(FlatSceneBuilder& fb) const {
        const Vec3f* src[6] = {&v0, &v1, &v2, &n0, &n1, &n2};
        std::vector<float>* dst[6] = {&fb.v0, &fb.v1, &fb.v2, &fb.n0, &fb.n1, &fb.n2};
        for (int k = 0; k < 6; ++k) { dst[k]->push_back(src[k]->x); dst[k]->push_back(src[k]->y); dst[k]->push_back(src[k]->z); }
    }
    // A lone Triangle in the scene is a one-face Model whose gate always passes is NOT equivalent (Model adds a
    // bounding-sphere gate), so it gets a gate that cannot reject: radius +inf.
    void flatten(FlatSceneBuilder& fb) const override {
        rt_model m; m.first_face = uint32_t(fb.v0.size() / 3); m.num_faces = 1;
        m.bb_center[0] = m.bb_center[1] = m.bb_center[2] = 0; m.bb_radius = -1.0f;   // < 0: no gate (see rt_model)
        appendTo(fb);
        fb.models.push_back(m);
        fb.addObject(RT_OBJ_MODEL, uint32_t(fb.models.size() - 1), fb.addMaterial(material));
    }
};

struct Model : Primitive {              // ref src/primitives.h:121-159
    std::vector<Triangle> faces;
    Sphere bbvol;                       // bounding sphere gate of Model::intersect (ref src/primitives.cpp:370)
    Model(const Material& m, const std::vector<Triangle>& f) : Primitive(m), faces(f), bbvol() { updateBBVol(); }
    Model(const Color& c, const std::vector<Triangle>& f) : Primitive(c), faces(f), bbvol() { updateBBVol(); }
    explicit Model(const Material& m) : Primitive(m), bbvol() {}
    Model() : bbvol() {}

    // centroid of all face corners (sequential float sums, divided via double reciprocal), radius = max distance
    // (ref src/primitives.cpp:205-236)
    void updateBBVol() {
        Vec3f center(0, 0, 0);
        const unsigned int numVertices = unsigned(faces.size() * 3);
        for (const auto& f : faces) { center += f.v0; center += f.v1; center += f.v2; }
        center /= Float(numVertices);
        Float radius = 0;
        for (const auto& f : faces) {
            const Vec3f* c[3] = {&f.v0, &f.v1, &f.v2};
            for (const Vec3f* p : c) { const Float t = (center - *p).length(); if (t > radius) radius = t; }
        }
        bbvol = Sphere(center, radius, Color(0));
    }
    Vec3f getNormal(const Vec3f&) const override { return Vec3f(0, 0, 0); }                       // ref src/primitives.cpp:399-401 ("currently broken")
    Model& scale(Float s) { for (auto& f : faces) f.scale(s); updateBBVol(); return *this; }      // :386-393
    Model& translate(const Vec3f& t) { return *this += t; }
    Model& operator*=(const Mat3d& m) { for (auto& f : faces) f *= m; return *this; }             // :403-408 (bbvol NOT updated)
    Model& operator+=(const Vec3f& t) { for (auto& f : faces) f += t; bbvol.center += t; return *this; }  // :410-416

    // ASCII PLY -> centred vertices -> unweighted smooth normals -> faces (ref src/primitives.cpp:238-362).
    static Model* load_ply(const std::string& fname, const Material& material, bool calcNormal = false);
    // The part of load_ply after the text is parsed (ref :287-362): re-centre, normals, faces. `vertices` is consumed.
    static Model* from_indexed(std::vector<Vec3f>& vertices, const std::vector<unsigned int>& idx,
                               const Material& material, bool calcNormal);

    void flatten(FlatSceneBuilder& fb) const override {
        rt_model m; m.first_face = uint32_t(fb.v0.size() / 3); m.num_faces = uint32_t(faces.size());
        m.bb_center[0] = bbvol.center.x; m.bb_center[1] = bbvol.center.y; m.bb_center[2] = bbvol.center.z;
        m.bb_radius = bbvol.radius;
        for (const auto& f : faces) f.appendTo(fb);
        fb.models.push_back(m);
        fb.addObject(RT_OBJ_MODEL, uint32_t(fb.models.size() - 1), fb.addMaterial(material));
    }
};

inline Model* Model::load_ply(const std::string& fname, const Material& material, bool calcNormal) {
    std::ifstream ifs(fname);
    if (!ifs.is_open()) throw std::runtime_error("Model " + fname + " could not be loaded. File does not exists.");
    std::string line, key;
    unsigned int nfaces = 0, nvertices = 0, val = 0;
    // header: first token decides; anything unknown is skipped (ref :252-271). Unlike the reference, EOF
    // before end_header is an error instead of an endless loop.
    for (;;) {
        if (!std::getline(ifs, line)) throw std::runtime_error("Model " + fname + ": no end_header");
        std::stringstream ss(line);
        key.clear();
        ss >> key;
        if (key == "end_header") break;
        if (key == "element") {
            ss >> key >> val;
            if (key == "vertex") nvertices = val; else if (key == "face") nfaces = val;
        }
    }
    std::vector<Vec3f> vertices;
    vertices.reserve(nvertices);
    for (unsigned i = 0; i < nvertices; ++i) {      // first three numbers of each line (ref :279-285)
        std::getline(ifs, line);
        std::stringstream ss(line);
        Float x = 0, y = 0, z = 0;
        ss >> x >> y >> z;
        vertices.emplace_back(x, y, z);
    }
    std::vector<unsigned int> idx;
    idx.reserve(size_t(nfaces) * 3);
    for (unsigned i = 0; i < nfaces; ++i) {         // "3 i j k" only (ref :310-319)
        std::getline(ifs, line);
        std::stringstream ss(line);
        unsigned int num = 0, a = 0, b = 0, c = 0;
        ss >> num >> a >> b >> c;
        if (num != 3) throw std::runtime_error("Only triangles are supported");
        if (a >= nvertices || b >= nvertices || c >= nvertices) throw std::runtime_error("Face index out of range");
        idx.push_back(a); idx.push_back(b); idx.push_back(c);
    }
    return from_indexed(vertices, idx, material, calcNormal);
}

inline Model* Model::from_indexed(std::vector<Vec3f>& vertices, const std::vector<unsigned int>& idx,
                                  const Material& material, bool calcNormal) {
    const unsigned nvertices = unsigned(vertices.size()), nfaces = unsigned(idx.size() / 3);
    std::vector<Vec3f> normals;
    {   // always re-centre on the vertex mean (ref :287-307; the file's faces come after, so the order is free)
        Vec3f center(0, 0, 0);
        for (const auto& v : vertices) center += v;
        center /= Float(nvertices);
        for (auto& v : vertices) v -= center;
    }
    if (calcNormal) {                               // per-vertex sum of unit face normals, in face order (ref :322-345)
        normals.assign(vertices.size(), Vec3f(0, 0, 0));
        for (unsigned i = 0; i < nfaces; ++i) {
            const Vec3f &a = vertices[idx[3 * i]], &b = vertices[idx[3 * i + 1]], &c = vertices[idx[3 * i + 2]];
            const Vec3f fn = Vec3f::getNormal(Vec3f::getVec(a, b), Vec3f::getVec(a, c));
            normals[idx[3 * i]] += fn; normals[idx[3 * i + 1]] += fn; normals[idx[3 * i + 2]] += fn;
        }
        for (auto& n : normals) n.normalize();
    }
    std::vector<Triangle> faces;
    faces.reserve(nfaces);
    for (unsigned i = 0; i < nfaces; ++i) {
        const unsigned a = idx[3 * i], b = idx[3 * i + 1], c = idx[3 * i + 2];
        if (calcNormal) faces.emplace_back(vertices[a], vertices[b], vertices[c], normals[a], normals[b], normals[c], material);
        else faces.emplace_back(vertices[a], vertices[b], vertices[c], material);
    }
    return new Model(material, faces);
}

class Camera {                          // ref src/camera.h:14-104; state is readable here (the device needs it)
    Vec3f eye, up, right, at;
    Float aspectRatio, fov;
    unsigned int imWidth, imHeight;
    std::vector<Point2D> antiAliasingPattern;
public:
    Camera(Float aspect, Float fov_, unsigned W, unsigned H)
        : Camera(Vec3f(0, 0, 0), Vec3f(0, 1, 0), Vec3f(0, 0, -1), aspect, fov_, W, H) {}
    Camera(const Vec3f& eye_, const Vec3f& up_, const Vec3f& at_, Float aspect, Float fov_, unsigned W, unsigned H,
           std::vector<Point2D> pattern = defaultAntiAliasingPattern())
        : eye(eye_), up(up_), at(at_), aspectRatio(aspect), fov(fov_), imWidth(W), imHeight(H),
          antiAliasingPattern(std::move(pattern)) {
        up.normalize(); at.normalize();                                     // ref :34-35
        // ref :36 writes cross_product(at, up) inside a constructor whose PARAMETERS are also called at/up, so the
        // un-normalised arguments are crossed, not the members just normalised. Kept: rays depend on it.
        right = cross_product(at_, up_);
    }
    static std::vector<Point2D> defaultAntiAliasingPattern() {             // ref :40-50
        return {Point2D(0.5f, 0.5f), Point2D(0.25f, 0.25f), Point2D(0.75f, 0.25f), Point2D(0.75f, 0.75f),
                Point2D(0.25f, 0.75f), Point2D(1.0f, 0), Point2D(1.0f, 1.0f), Point2D(0, 1.0f)};
    }
    static std::vector<Point2D> noAntiAliasing() { return {}; }
    Camera& rotate(Float alpha, Float beta, Float gamma) {                  // ref :81-89
        const Mat3d rot = Mat3d::rotation(alpha, beta, gamma);
        up *= rot; up.normalize(); at *= rot; at.normalize(); right = cross_product(at, up);
        return *this;
    }
    Camera& move(const Vec3f& t) { eye += t; return *this; }                // ref :91-94
    unsigned int getImWidth() const { return imWidth; }
    unsigned int getImHeight() const { return imHeight; }
    const std::vector<Point2D>& getAntiAliasingPattern() const { return antiAliasingPattern; }
    rt_camera flat() const {
        rt_camera c;
        const Vec3f* src[4] = {&eye, &up, &right, &at};
        float* dst[4] = {c.eye, c.up, c.right, c.at};
        for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
        c.aspectRatio = aspectRatio; c.fov = fov; c.imWidth = imWidth; c.imHeight = imHeight;
        return c;
    }
};

// Owns the primitives and lights (ref src/scene.h:29-39) and, lazily, their HBM copy.
class Scene {
public:
    Scene() = default;
    explicit Scene(const std::string& filename);          // YAML loader, raytracer_amd/host/scene_yaml.hpp
    ~Scene() { if (dev_) rt_scene_destroy(dev_); if (multi_) rt_multi_destroy(multi_); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    const std::vector<std::unique_ptr<Primitive>>& getObjects() const { return objects; }
    const std::vector<Light>& getLights() const { return lights; }
    void addObject(std::unique_ptr<Primitive> p) { objects.push_back(std::move(p)); invalidate(); }
    void addLight(const Light& l) { lights.push_back(l); invalidate(); }

    void flatten(FlatSceneBuilder& fb) const {
        for (const auto& o : objects) o->flatten(fb);
        for (const auto& l : lights) {
            rt_light fl;
            fl.pos[0] = l.pos.x; fl.pos[1] = l.pos.y; fl.pos[2] = l.pos.z;
            fl.color[0] = l.color.r; fl.color[1] = l.color.g; fl.color[2] = l.color.b;
            fb.lights.push_back(fl);
        }
    }
    // Device copy on `device`, created on first use (BVH build + upload) and kept until the scene changes.
    rt_scene* deviceScene(int device = 0) const {
        if (!dev_ || dev_device_ != device) {
            invalidate();
            FlatSceneBuilder fb;
            flatten(fb);
            const rt_flat_scene fs = fb.view();
            if (rt_scene_create(&fs, device, &dev_) != RT_OK) throw std::runtime_error(rt_last_error());
            dev_device_ = device;
        }
        return dev_;
    }
    // The GPUs render() spreads a whole frame over (default: device 0 alone). No counterpart in the reference, whose
    // workers are CPU threads (ref src/raytracer.cpp:577-627).
    void setDevices(std::vector<int> devices) { invalidate(); devices_ = std::move(devices); }
    const std::vector<int>& devices() const { return devices_; }
    // The replicas on devices(), created on first use and kept until the scene changes.
    rt_multi* multiScene() const {
        if (!multi_) {
            FlatSceneBuilder fb;
            flatten(fb);
            const rt_flat_scene fs = fb.view();
            if (rt_multi_create(&fs, devices_.data(), uint32_t(devices_.size()), 0u, &multi_) != RT_OK) throw std::runtime_error(rt_last_error());
        }
        return multi_;
    }
private:
    void invalidate() const {
        if (dev_) { rt_scene_destroy(dev_); dev_ = nullptr; }
        if (multi_) { rt_multi_destroy(multi_); multi_ = nullptr; }
    }
    std::vector<int> devices_;
    mutable rt_multi* multi_ = nullptr;
    std::vector<std::unique_ptr<Primitive>> objects;
    std::vector<Light> lights;
    mutable rt_scene* dev_ = nullptr;
    mutable int dev_device_ = -1;
};

using ImageType = unsigned char;        // ref src/raytracer.cpp:72

// Render limits the reference fixes at file scope; mutable here (ref src/raytracer.cpp:74).
inline int& maxDepth() { static int d = 10; return d; }

// Same signature and contract as the reference's render() (ref src/raytracer.cpp:422-463): writes the BGR8
// bytes of rectangle [x_start, x_start+cW) x [y_start, y_start+cH) of the full-image buffer `img`,
// adds the rectangle's pixels to finPix AS ITS PASSES COMPLETE (the reference: ++finPix per pixel, ref :450, read by
// the main thread for its pix/s line, ref :217-225; a frame of up to 8 M samples is one pass) and decrements
// thread_count. A scene with setDevices({...}) of more than one GPU renders whole-image calls on all of them.
// Throws on device errors (the reference's render cannot fail).
inline void render(ImageType* img, unsigned int x_start, unsigned int y_start, unsigned int cH, unsigned int cW,
                   const Camera& camera, const Scene& scene, const Color& background, unsigned int& finPix,
                   unsigned int& thread_count) {
    const rt_camera cam = camera.flat();
    std::vector<float> aa;
    for (const auto& p : camera.getAntiAliasingPattern()) { aa.push_back(p.x()); aa.push_back(p.y()); }
    rt_params prm;
    prm.background[0] = background.r; prm.background[1] = background.g; prm.background[2] = background.b;
    prm.max_depth = maxDepth();
    prm.n_aa = uint32_t(aa.size() / 2);
    prm.aa_xy = aa.data();
    const bool whole = x_start == 0 && y_start == 0 && cW == cam.imWidth && cH == cam.imHeight;
    if (scene.devices().size() > 1 && whole) {
        if (rt_multi_render(scene.multiScene(), &cam, &prm, img, nullptr) != RT_OK) throw std::runtime_error(rt_last_error());
        finPix += cW * cH;
    } else {
        struct Progress { unsigned int* fin; unsigned int base; } pg{&finPix, finPix};
        const rt_progress_fn cb = [](void* u, uint64_t done, uint64_t) {
            Progress* g = static_cast<Progress*>(u);
            *g->fin = g->base + static_cast<unsigned int>(done);
        };
        rt_scene* dev = scene.devices().empty() ? scene.deviceScene() : scene.deviceScene(scene.devices()[0]);
        if (rt_render_progress(dev, &cam, &prm, x_start, y_start, cW, cH, img, nullptr, cb, &pg) != RT_OK)
            throw std::runtime_error(rt_last_error());
    }
    if (thread_count > 0) --thread_count;
}
