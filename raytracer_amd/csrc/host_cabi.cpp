// host_cabi.cpp — C entry points over the C++ host surface (include/raytracer_amd_host.h).
// Compiled with -ffp-contract=off: the host data must match the reference's x86 build bit for bit.
#include "raytracer_amd_host.h"

#include "../host/raytracer.hpp"
#include "../host/scene_yaml.hpp"
#include "../host/image_io.hpp"

void rt_set_error(const char* fmt, ...);   // rt_api.cpp

namespace {
Vec3f v3(const float* p) { return Vec3f(p[0], p[1], p[2]); }
void put(float* d, const Vec3f& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
}  // namespace

struct rt_host_model {
    std::unique_ptr<Model> m;
};

struct rt_host_scene {
    std::unique_ptr<Scene> scene;
    FlatSceneBuilder fb;
    rt_flat_scene view;
};

extern "C" {

int rt_host_camera_make(const float* eye, const float* up, const float* at, float aspect, float fov,
                        uint32_t W, uint32_t H, const float* rot, const float* move, rt_camera* out) {
    if (!eye || !up || !at || !out) { rt_set_error("rt_host_camera_make: null argument"); return RT_ERR_INVALID; }
    Camera cam(v3(eye), v3(up), v3(at), aspect, fov, W, H, Camera::noAntiAliasing());
    if (rot) cam.rotate(rot[0], rot[1], rot[2]);
    if (move) cam.move(v3(move));
    *out = cam.flat();
    return RT_OK;
}

int rt_host_plane_make(const float* normal, float dist, rt_plane* out) {
    if (!normal || !out) { rt_set_error("rt_host_plane_make: null argument"); return RT_ERR_INVALID; }
    const Plane p(v3(normal), dist, Material());
    out->a = p.a; out->b = p.b; out->c = p.c; out->d = p.d;
    return RT_OK;
}

int rt_host_primitive_intersect(int kind, const float* prm, const float* ray6, const float* point3, int* hit_out,
                                float* dist_out, float* normal_out, float* getnormal_out) {
    if (!prm || !ray6 || !hit_out || !dist_out || !normal_out || (getnormal_out && !point3)) { rt_set_error("rt_host_primitive_intersect: null argument"); return RT_ERR_INVALID; }
    std::unique_ptr<Primitive> p;
    if (kind == 0) p.reset(new Sphere(v3(prm), prm[3], Material()));
    else if (kind == 1) p.reset(new Plane(v3(prm), prm[3], Material()));
    else if (kind == 2) p.reset(new Triangle(v3(prm), v3(prm + 3), v3(prm + 6), Material()));
    else { rt_set_error("rt_host_primitive_intersect: kind %d (0 sphere, 1 plane, 2 triangle)", kind); return RT_ERR_INVALID; }
    try {
        Float dist = 0;
        Vec3f normal(0, 0, 0);
        *hit_out = p->intersect(Ray(v3(ray6), v3(ray6 + 3)), dist, normal) ? 1 : 0;
        *dist_out = dist;
        put(normal_out, normal);
        if (getnormal_out) put(getnormal_out, p->getNormal(v3(point3)));
    } catch (const std::exception& e) {
        rt_set_error("%s", e.what());
        return RT_ERR_DEVICE;
    }
    return RT_OK;
}

int rt_host_mat3_rotation(float rx, float ry, float rz, float* out9) {
    if (!out9) { rt_set_error("rt_host_mat3_rotation: null argument"); return RT_ERR_INVALID; }
    const Mat3d m = Mat3d::rotation(rx, ry, rz);
    for (int i = 0; i < 9; ++i) out9[i] = m[i];
    return RT_OK;
}

int rt_host_bbvol(const float* v0, const float* v1, const float* v2, uint32_t n, float* bb) {
    if (!v0 || !v1 || !v2 || !bb) { rt_set_error("rt_host_bbvol: null argument"); return RT_ERR_INVALID; }
    std::vector<Triangle> faces;
    faces.reserve(n);
    const Vec3f z(0, 0, 0);
    for (uint32_t f = 0; f < n; ++f)
        faces.emplace_back(v3(v0 + 3 * size_t(f)), v3(v1 + 3 * size_t(f)), v3(v2 + 3 * size_t(f)), z, z, z);
    const Model m(Material(), faces);
    put(bb, m.bbvol.center);
    bb[3] = m.bbvol.radius;
    return RT_OK;
}

int rt_host_model_load_ply(const char* fname, rt_host_model** out) {
    if (!fname || !out) { rt_set_error("rt_host_model_load_ply: null argument"); return RT_ERR_INVALID; }
    try {
        std::unique_ptr<Model> m(Model::load_ply(fname, Material(), true));
        *out = new rt_host_model{std::move(m)};
        return RT_OK;
    } catch (const std::exception& e) {
        rt_set_error("rt_host_model_load_ply: %s", e.what());
        return RT_ERR_INVALID;
    }
}

int rt_host_model_from_arrays(const float* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_faces,
                              rt_host_model** out) {
    if (!vertices || !indices || !out || n_vertices == 0) {
        rt_set_error("rt_host_model_from_arrays: null argument or no vertices");
        return RT_ERR_INVALID;
    }
    for (size_t i = 0; i < size_t(n_faces) * 3; ++i)
        if (indices[i] >= n_vertices) { rt_set_error("rt_host_model_from_arrays: face index out of range"); return RT_ERR_INVALID; }
    try {
        std::vector<Vec3f> V;
        V.reserve(n_vertices);
        for (uint32_t i = 0; i < n_vertices; ++i) V.push_back(v3(vertices + 3 * size_t(i)));
        const std::vector<unsigned int> idx(indices, indices + size_t(n_faces) * 3);
        std::unique_ptr<Model> m(Model::from_indexed(V, idx, Material(), true));
        *out = new rt_host_model{std::move(m)};
        return RT_OK;
    } catch (const std::exception& e) {
        rt_set_error("rt_host_model_from_arrays: %s", e.what());
        return RT_ERR_INVALID;
    }
}

int rt_host_model_transform(rt_host_model* h, const float* rot, double scale, const float* tr) {
    if (!h || !rot || !tr) { rt_set_error("rt_host_model_transform: null argument"); return RT_ERR_INVALID; }
    *h->m *= Mat3d::rotation(v3(rot));
    h->m->scale(Float(scale));
    *h->m += v3(tr);
    return RT_OK;
}

uint32_t rt_host_model_num_faces(const rt_host_model* h) { return h ? uint32_t(h->m->faces.size()) : 0; }

int rt_host_model_get(const rt_host_model* h, float* v0, float* v1, float* v2, float* n0, float* n1, float* n2,
                      float* bb) {
    if (!h) { rt_set_error("rt_host_model_get: null model"); return RT_ERR_INVALID; }
    size_t i = 0;
    for (const auto& f : h->m->faces) {
        if (v0) put(v0 + 3 * i, f.v0);
        if (v1) put(v1 + 3 * i, f.v1);
        if (v2) put(v2 + 3 * i, f.v2);
        if (n0) put(n0 + 3 * i, f.n0);
        if (n1) put(n1 + 3 * i, f.n1);
        if (n2) put(n2 + 3 * i, f.n2);
        ++i;
    }
    if (bb) { put(bb, h->m->bbvol.center); bb[3] = h->m->bbvol.radius; }
    return RT_OK;
}

void rt_host_model_free(rt_host_model* h) { delete h; }

int rt_host_known_answers(float* out) {
    if (!out) { rt_set_error("rt_host_known_answers: null argument"); return RT_ERR_INVALID; }
    Vec3f v1, v2(1, 2, 3);
    v1 += v2;
    put(out, v1);
    put(out + 3, v1 * 2);
    Color c1, c2(1, 2, 3);
    c1 += c2;
    const Color cc = c1 * c2, c2x = c1 * 2;
    out[6] = cc.r; out[7] = cc.g; out[8] = cc.b;
    out[9] = c2x.r; out[10] = c2x.g; out[11] = c2x.b;
    const Vec3f cp = cross_product(Vec3f(1, 0, 0), Vec3f(0, 1, 0));
    put(out + 12, cp);
    out[15] = float(std::cos(double(cp.dotProduct(Vec3f(1, 0, 0)))));
    return RT_OK;
}


int rt_host_scene_load_yaml(const char* fname, rt_host_scene** out) {
    if (!fname || !out) { rt_set_error("rt_host_scene_load_yaml: null argument"); return RT_ERR_INVALID; }
    *out = nullptr;
    try {
        std::unique_ptr<rt_host_scene> h(new rt_host_scene());
        h->scene.reset(new Scene(std::string(fname)));
        h->scene->flatten(h->fb);
        h->view = h->fb.view();
        *out = h.release();
    } catch (const std::exception& e) {
        rt_set_error("%s", e.what());
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

int rt_host_write_image(const char* fname, const uint8_t* bgr, uint32_t W, uint32_t H) {
    if (!fname || !bgr) { rt_set_error("rt_host_write_image: null argument"); return RT_ERR_INVALID; }
    try {
        write_image(fname, bgr, W, H);
    } catch (const std::exception& e) {
        rt_set_error("%s", e.what());
        return RT_ERR_INVALID;
    }
    return RT_OK;
}

const rt_flat_scene* rt_host_scene_flat(const rt_host_scene* s) { return s ? &s->view : nullptr; }

void rt_host_scene_free(rt_host_scene* s) { delete s; }

}  // extern "C"
