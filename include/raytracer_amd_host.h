/*
 * raytracer_amd_host.h — C view of the C++ host surface (raytracer_amd/host/raytracer.hpp) for non-C++
 * callers (the Python tests and bench.py bind these through ctypes). Host-only: none of these touch the GPU, except
 * rt_host_primitive_intersect, which says so.
 * Each entry point names the reference code whose result it reproduces bit for bit.
 */
#ifndef RAYTRACER_AMD_HOST_H
#define RAYTRACER_AMD_HOST_H

#include "raytracer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces: Camera::Camera + Camera::rotate + Camera::move (reference src/camera.h:30-37, 81-94).
 * rot (alpha,beta,gamma) and move may be NULL. */
int rt_host_camera_make(const float* eye, const float* up, const float* at, float aspect, float fov,
                        uint32_t imWidth, uint32_t imHeight, const float* rot, const float* move,
                        rt_camera* out);

/* Replaces: Plane::Plane(Vec3f normal, Float dist, ...) normalisation (reference src/primitives.h:50-55). */
int rt_host_plane_make(const float* normal, float dist, rt_plane* out);

/* Replaces: Mat3d::rotation (reference src/container.cpp:13-15; src/container.h:324-352). 9 floats, row-major. */
int rt_host_mat3_rotation(float rx, float ry, float rz, float* out9);

/* Replaces: Model::updateBBVol (reference src/primitives.cpp:205-236). bb = {cx, cy, cz, radius}. */
int rt_host_bbvol(const float* v0, const float* v1, const float* v2, uint32_t n_faces, float* bb);

typedef struct rt_host_model rt_host_model;

/* Replaces: Model::load_ply(fname, material, calcNormal=true) (reference src/primitives.cpp:238-362). */
int rt_host_model_load_ply(const char* fname, rt_host_model** out);

/* Replaces: the part of Model::load_ply after the text is parsed (reference src/primitives.cpp:287-362:
 * re-centre on the vertex mean, per-vertex normals, faces), for meshes that are already in memory
 * (vertices: n_vertices*3 floats, indices: n_faces*3). Same result as writing the arrays as an ASCII .ply with
 * 9 significant digits and calling rt_host_model_load_ply. */
int rt_host_model_from_arrays(const float* vertices, uint32_t n_vertices, const uint32_t* indices,
                              uint32_t n_faces, rt_host_model** out);

/* Replaces: the mesh branch of the scene loader (reference src/scene.cpp:161-163):
 * model *= Mat3d::rotation(rot); model.scale(scale); model += translation. */
int rt_host_model_transform(rt_host_model* m, const float* rot, double scale, const float* translation);

uint32_t rt_host_model_num_faces(const rt_host_model* m);

/* Copies Triangle::v0,v1,v2,n0,n1,n2 (n_faces*3 floats each) and bbvol {cx,cy,cz,r} out of the model
 * (reference src/primitives.h:65,125). */
int rt_host_model_get(const rt_host_model* m, float* v0, float* v1, float* v2, float* n0, float* n1,
                      float* n2, float* bb);

void rt_host_model_free(rt_host_model* m);

/* The known answers of check_op_overloading() (reference src/raytracer.cpp:78-100) recomputed with the host
 * surface's operators: out[0..2] v1+=v2, [3..5] v1*2, [6..8] c1*c2, [9..11] c1*2, [12..14] cross(e1,e2),
 * [15] cos(cross.e1). */
int rt_host_known_answers(float* out16);

/* Scene::Scene(filename) (reference src/scene.cpp:32-189: colors, materials, lights, primitives of a scene.yaml;
 * meshes go through load_ply + the transform chain), without OpenCV. The flat view (object order = file order)
 * stays valid until rt_host_scene_free and can be passed to rt_scene_create as it is. */
typedef struct rt_host_scene rt_host_scene;
int rt_host_scene_load_yaml(const char* fname, rt_host_scene** out);
const rt_flat_scene* rt_host_scene_flat(const rt_host_scene* scene);
void rt_host_scene_free(rt_host_scene* scene);

/* Primitive::intersect and Primitive::getNormal (reference src/primitives.h:23-24, src/primitives.cpp:11-29, 42-81,
 * 128-130, 173-203) called on ONE primitive of the host surface: kind 0 = Sphere, prm {cx, cy, cz, radius}; 1 = Plane,
 * prm {nx, ny, nz, d} (the constructor normalises the normal); 2 = Triangle, prm {v0, v1, v2} (9 floats, flat normal).
 * ray6 = origin, direction. The intersection RUNS ON THE DEVICE (the host surface's intersect() flattens the primitive
 * into a one-object scene and calls rt_intersect): *hit_out 0/1, dist_out / normal_out (3 floats) as the reference's
 * intersect() leaves them on a hit. getnormal_out (3 floats, may be NULL) = getNormal(point3) (host arithmetic). */
int rt_host_primitive_intersect(int kind, const float* prm, const float* ray6, const float* point3, int* hit_out,
                                float* dist_out, float* normal_out, float* getnormal_out);

/* cv::imwrite(outFilename, img) of main() (reference src/raytracer.cpp:720) for the BGR8 frame rt_render fills:
 * .png (8-bit RGB, uncompressed deflate) or .ppm (P6). */
int rt_host_write_image(const char* fname, const uint8_t* bgr, uint32_t W, uint32_t H);

#ifdef __cplusplus
}
#endif
#endif
