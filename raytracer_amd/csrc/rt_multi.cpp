// rt_multi.cpp — one frame on several GPUs of ONE process (include/raytracer_amd.h: rt_multi_*).
//
// The reference's own model for spreading a frame is the tile pool of thread_render() (reference
// src/raytracer.cpp:577-627): fixed tiles handed to workers that all write into one shared `img`. Here the workers are
// GPUs: the scene is replicated (rt_scene per device), 32x32-pixel tiles are dealt round-robin (tile_index % N ==
// part, rt_render_tiles_device), every device renders its tiles into a tile-major buffer of its own, and the one
// exchange step of the path - the tiles have to end up in one frame - is a peer copy per device into the first
// device's gather buffer (hipMemcpyPeerAsync: within one process a direct xGMI copy, each peer over its own link; no
// communicator needed), after which the first device scatters the tiles into the row-major frame (rt_untile_device).
// Everything is built from the library's public entry points; bench.py's one-process-per-GPU path does the same with
// torch.distributed (RCCL) as the exchange.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "raytracer_amd.h"

void rt_set_error(const char* fmt, ...);      // rt_api.cpp (thread-local message)

#define MULTI_TRY(expr)                                                                       \
    do {                                                                                      \
        const hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                               \
            rt_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                          \
            return RT_ERR_DEVICE;                                                             \
        }                                                                                     \
    } while (0)

struct rt_multi {
    std::vector<int> devices;            // part q renders on devices[q] (a device may appear more than once)
    std::vector<rt_scene*> scenes;
    std::vector<uint8_t*> d_tiles;       // part q's tile-major buffer on devices[q] (part 0: inside d_all)
    std::vector<hipEvent_t> ev_rendered; // part q's tiles are complete (recorded on scene q's stream)
    hipEvent_t ev_gathered = nullptr;    // the peer copies of the previous frame have read the parts' buffers (device 0)
    uint8_t* d_all = nullptr;            // gather buffer on devices[0]: part q at q * slot_tiles tiles
    uint8_t* d_frame = nullptr;          // row-major frame on devices[0] (rt_multi_render)
    size_t part_bytes = 0, frame_bytes = 0;
    uint32_t tile = 32;
    bool gathered_once = false;
};

static void multi_free(rt_multi* m) {
    if (!m) return;
    for (size_t q = 0; q < m->scenes.size(); ++q) {
        if (m->scenes[q]) { rt_scene_status(m->scenes[q]); }      // wait for frames in flight
    }
    for (size_t q = 0; q < m->devices.size(); ++q) {
        if (!m->scenes[q]) continue;      // (creation failed there: nothing of ours on that device, which may not even exist)
        hipSetDevice(m->devices[q]);
        if (q < m->d_tiles.size() && q > 0) hipFree(m->d_tiles[q]);
        if (q < m->ev_rendered.size() && m->ev_rendered[q]) hipEventDestroy(m->ev_rendered[q]);
    }
    if (!m->devices.empty() && m->scenes[0]) {
        hipSetDevice(m->devices[0]);
        if (m->ev_gathered) hipEventDestroy(m->ev_gathered);
        hipFree(m->d_all);
        hipFree(m->d_frame);
    }
    for (rt_scene* s : m->scenes) rt_scene_destroy(s);
    (void)hipGetLastError();
    delete m;
}

extern "C" int rt_multi_create(const rt_flat_scene* scene, const int* devices, uint32_t n_devices, uint32_t flags, rt_multi** out) {
    if (!out) { rt_set_error("out is NULL"); return RT_ERR_INVALID; }
    *out = nullptr;
    if (!scene || !devices || n_devices == 0 || n_devices > RT_MULTI_MAX_DEVICES) {
        rt_set_error("rt_multi_create: 1..%d devices", RT_MULTI_MAX_DEVICES);
        return RT_ERR_INVALID;
    }
    rt_multi* m = new rt_multi();
    m->devices.assign(devices, devices + n_devices);
    m->scenes.assign(n_devices, nullptr);
    m->d_tiles.assign(n_devices, nullptr);
    m->ev_rendered.assign(n_devices, nullptr);
    // the replicas are built side by side (a host BVH build of 10 M faces takes over a second)
    std::vector<int> rcs(n_devices, RT_OK);
    std::vector<std::string> errs(n_devices);
    std::vector<std::thread> workers;
    for (uint32_t q = 0; q < n_devices; ++q)
        workers.emplace_back([&, q] {
            rcs[q] = rt_scene_create_ex(scene, m->devices[q], flags, &m->scenes[q]);
            if (rcs[q] != RT_OK) errs[q] = rt_last_error();
        });
    for (std::thread& t : workers) t.join();
    for (uint32_t q = 0; q < n_devices; ++q)
        if (rcs[q] != RT_OK) {
            rt_set_error("device %d: %s", m->devices[q], errs[q].c_str());
            const int rc = rcs[q];
            multi_free(m);
            return rc;
        }
    for (uint32_t q = 0; q < n_devices; ++q) {
        if (hipSetDevice(m->devices[q]) != hipSuccess || hipEventCreateWithFlags(&m->ev_rendered[q], hipEventDisableTiming) != hipSuccess) {
            rt_set_error("device %d: cannot create an event", m->devices[q]);
            multi_free(m);
            return RT_ERR_DEVICE;
        }
    }
    hipSetDevice(m->devices[0]);
    if (hipEventCreateWithFlags(&m->ev_gathered, hipEventDisableTiming) != hipSuccess) { multi_free(m); rt_set_error("cannot create an event"); return RT_ERR_DEVICE; }
    // direct peer copies into the first device where the hardware allows (xGMI); without it the copy is staged
    for (uint32_t q = 1; q < n_devices; ++q)
        if (m->devices[q] != m->devices[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m->devices[0], m->devices[q]) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(m->devices[q], 0);
                if (e != hipSuccess) (void)hipGetLastError();      // (already enabled is fine)
            }
        }
    *out = m;
    return RT_OK;
}

extern "C" void rt_multi_destroy(rt_multi* m) { multi_free(m); }
extern "C" uint32_t rt_multi_parts(const rt_multi* m) { return m ? uint32_t(m->devices.size()) : 0u; }
extern "C" rt_scene* rt_multi_scene(const rt_multi* m, uint32_t part) { return (m && part < m->scenes.size()) ? m->scenes[part] : nullptr; }

extern "C" int rt_multi_status(rt_multi* m) {
    if (!m) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = RT_OK;
    // part 0 last: its stream carries the gather and the untile, which wait for the others
    for (size_t q = m->scenes.size(); q-- > 0;) {
        const int r = rt_scene_status(m->scenes[q]);
        if (r != RT_OK && rc == RT_OK) rc = r;
    }
    return rc;
}

static int ensure_buffers(rt_multi* m, const rt_camera* cam, bool want_frame) {
    const uint32_t n = uint32_t(m->devices.size()), t = m->tile;
    const size_t need = size_t(rt_tiles_owned(cam->imWidth, cam->imHeight, t, t, 0, n)) * t * t * 3;      // part 0 owns the most
    if (need > m->part_bytes) {
        for (uint32_t q = 0; q < n; ++q) rt_scene_status(m->scenes[q]);      // nothing may still read the old buffers
        for (uint32_t q = 1; q < n; ++q) {
            MULTI_TRY(hipSetDevice(m->devices[q]));
            hipFree(m->d_tiles[q]); m->d_tiles[q] = nullptr;
            MULTI_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_tiles[q]), need));
        }
        MULTI_TRY(hipSetDevice(m->devices[0]));
        hipFree(m->d_all); m->d_all = nullptr;
        MULTI_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_all), need * n));
        m->d_tiles[0] = m->d_all;        // part 0 renders straight into its slot of the gather buffer
        m->part_bytes = need;
        m->gathered_once = false;
    }
    const size_t fb = size_t(cam->imWidth) * cam->imHeight * 3;
    if (want_frame && fb > m->frame_bytes) {
        rt_scene_status(m->scenes[0]);
        MULTI_TRY(hipSetDevice(m->devices[0]));
        hipFree(m->d_frame); m->d_frame = nullptr;
        MULTI_TRY(hipMalloc(reinterpret_cast<void**>(&m->d_frame), fb));
        m->frame_bytes = fb;
    }
    return RT_OK;
}

static void add_stats(rt_stats* sum, const rt_stats& s) {
    sum->trace_calls += s.trace_calls; sum->shadow_rays += s.shadow_rays;
    for (int l = 0; l < 16; ++l) sum->level_rays[l] += s.level_rays[l];
    sum->passes += s.passes; sum->retries += s.retries;
    sum->kernel_ms = std::max(sum->kernel_ms, s.kernel_ms);      // the parts run side by side
    for (int k = 0; k < 5; ++k) { sum->kernel_ms_by[k] += s.kernel_ms_by[k]; sum->kernel_launches_by[k] += s.kernel_launches_by[k]; }
}

// Asynchronous: returns when everything is enqueued; the frame is complete on devices[0] when the FIRST part's stream
// (rt_scene_stream(rt_multi_scene(m, 0))) has drained, which rt_multi_status waits for. With stats the parts run one
// after the other, each synchronously (their counters are read back), and the sums are returned.
extern "C" int rt_multi_render_device(rt_multi* m, const rt_camera* cam, const rt_params* prm, uint8_t* d_bgr_out, rt_stats* stats) {
    if (!m || !cam || !prm || !d_bgr_out) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    if (cam->imWidth == 0 || cam->imHeight == 0) { rt_set_error("empty image"); return RT_ERR_INVALID; }
    const uint32_t n = uint32_t(m->devices.size()), t = m->tile;
    int rc = ensure_buffers(m, cam, false);
    if (rc != RT_OK) return rc;
    if (stats) std::memset(stats, 0, sizeof *stats);
    const uint32_t slot_tiles = uint32_t(m->part_bytes / (size_t(t) * t * 3));
    hipStream_t s0 = static_cast<hipStream_t>(rt_scene_stream(m->scenes[0]));
    // A failure after some parts were enqueued must not leave half a frame behind: the parts already on their way are
    // waited for (their status is dropped: the call fails anyway) and the cross-device event state starts afresh, so the
    // next frame neither waits for an ev_gathered that was never recorded for it nor reads an ev_rendered of this one.
    auto abandon = [&](int code) {
        char keep[512];
        std::snprintf(keep, sizeof keep, "%s", rt_last_error());
        for (uint32_t p = 0; p < n; ++p) (void)rt_scene_status(m->scenes[p]);
        m->gathered_once = false;
        rt_set_error("%s", keep);
        return code;
    };
    for (uint32_t q = 0; q < n; ++q) {
        if (hipSetDevice(m->devices[q]) != hipSuccess) { rt_set_error("hipSetDevice(%d) failed", m->devices[q]); (void)hipGetLastError(); return abandon(RT_ERR_DEVICE); }
        hipStream_t sq = static_cast<hipStream_t>(rt_scene_stream(m->scenes[q]));
        // part q's buffer is free again once the previous frame's copies have read it
        if (q > 0 && m->gathered_once && hipStreamWaitEvent(sq, m->ev_gathered, 0) != hipSuccess) { rt_set_error("hipStreamWaitEvent failed"); (void)hipGetLastError(); return abandon(RT_ERR_DEVICE); }
        rt_stats ps;
        rc = rt_render_tiles_device(m->scenes[q], cam, prm, t, t, q, n, m->d_tiles[q], nullptr, stats ? &ps : nullptr);
        if (rc != RT_OK) return abandon(rc);
        if (stats) add_stats(stats, ps);
        if (q > 0 && hipEventRecord(m->ev_rendered[q], sq) != hipSuccess) { rt_set_error("hipEventRecord failed"); (void)hipGetLastError(); return abandon(RT_ERR_DEVICE); }
    }
#define MULTI_TRY_ABANDON(expr) do { if ((expr) != hipSuccess) { rt_set_error("%s failed (%s:%d)", #expr, __FILE__, __LINE__); (void)hipGetLastError(); return abandon(RT_ERR_DEVICE); } } while (0)
    MULTI_TRY_ABANDON(hipSetDevice(m->devices[0]));
    for (uint32_t q = 1; q < n; ++q) {
        const size_t bytes = size_t(rt_tiles_owned(cam->imWidth, cam->imHeight, t, t, q, n)) * t * t * 3;
        MULTI_TRY_ABANDON(hipStreamWaitEvent(s0, m->ev_rendered[q], 0));
        uint8_t* dst = m->d_all + size_t(q) * m->part_bytes;
        // (the branch between DIFFERENT devices - hipMemcpyPeerAsync, the cross-device wait above - has not run yet:
        // every box this was developed on had one GPU; INTEGRATION.md says so)
        if (m->devices[q] == m->devices[0]) MULTI_TRY_ABANDON(hipMemcpyAsync(dst, m->d_tiles[q], bytes, hipMemcpyDeviceToDevice, s0));
        else MULTI_TRY_ABANDON(hipMemcpyPeerAsync(dst, m->devices[0], m->d_tiles[q], m->devices[q], bytes, s0));
    }
    if (n > 1) { MULTI_TRY_ABANDON(hipEventRecord(m->ev_gathered, s0)); m->gathered_once = true; }
#undef MULTI_TRY_ABANDON
    return rt_untile_device(m->devices[0], cam->imWidth, cam->imHeight, t, t, n, slot_tiles, m->d_all, d_bgr_out, s0);
}

// render() for the whole image on all devices: the frame lands in the caller's host buffer (the reference's `img`).
extern "C" int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_params* prm, uint8_t* bgr_out, rt_stats* stats) {
    if (!m || !cam || !bgr_out) { rt_set_error("null argument"); return RT_ERR_INVALID; }
    int rc = ensure_buffers(m, cam, true);
    if (rc != RT_OK) return rc;
    // asynchronous frames cannot re-run a pass that ran out of ray pool; the scenes shrink their passes and the frame is
    // rendered again (results do not depend on the pass size)
    for (int attempt = 0; attempt < 8; ++attempt) {
        rc = rt_multi_render_device(m, cam, prm, m->d_frame, stats);
        if (rc != RT_OK) return rc;
        rc = rt_multi_status(m);
        if (rc != RT_ERR_OVERFLOW) break;
    }
    if (rc != RT_OK) return rc;
    MULTI_TRY(hipSetDevice(m->devices[0]));
    hipStream_t s0 = static_cast<hipStream_t>(rt_scene_stream(m->scenes[0]));
    MULTI_TRY(hipMemcpyAsync(bgr_out, m->d_frame, size_t(cam->imWidth) * cam->imHeight * 3, hipMemcpyDeviceToHost, s0));
    MULTI_TRY(hipStreamSynchronize(s0));
    return RT_OK;
}
