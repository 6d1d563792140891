// raytracer_main.cpp — the reference's main() (ref src/raytracer.cpp:629-723) on the MI355X path, without the OpenCV
// preview window: `raytracer_main scene.yaml [-o out.png] [-w W -h H] [-aa] [-d depth] [-devices 0,1,...]`. The frame is
// rendered by a worker thread while the main thread prints the reference's progress text ("42%; t: 3 s; 1234 pix/s",
// ref :217-225, every 200 ms as preview() does, ref :708) - to stdout instead of into the window.
// Same defaults as the reference: 800x600, camera at the origin looking down -z, FOV 60, moved by (-0.5, 0, -0.25),
// no anti-aliasing, background (0, 0.5, 0.5), MAX_DEPTH 10, output "raytracer.png".
//
// Build:  g++ -std=c++17 -O2 -ffp-contract=off -Iinclude examples/raytracer_main.cpp -Lraytracer_amd/csrc -lrt_amd \
//             -Wl,-rpath,$PWD/raytracer_amd/csrc -o raytracer_main
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <thread>

#include "../raytracer_amd/host/raytracer.hpp"
#include "../raytracer_amd/host/scene_yaml.hpp"
#include "../raytracer_amd/host/image_io.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "You have to provide a scene file.";
        return -1;
    }
    std::string outFilename{"raytracer.png"};
    unsigned int imWidth = 800, imHeight = 600;
    bool aa = false;
    std::vector<int> devices;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-o" && i + 1 < argc) outFilename = argv[++i];
        else if (a == "-w" && i + 1 < argc) imWidth = unsigned(std::atoi(argv[++i]));
        else if (a == "-h" && i + 1 < argc) imHeight = unsigned(std::atoi(argv[++i]));
        else if (a == "-d" && i + 1 < argc) maxDepth() = std::atoi(argv[++i]);
        else if (a == "-aa") aa = true;
        else if (a == "-devices" && i + 1 < argc) {      // GPUs to spread the frame over (a device may be named twice)
            std::stringstream ss(argv[++i]);
            for (std::string tok; std::getline(ss, tok, ',');) devices.push_back(std::atoi(tok.c_str()));
        }
    }
    try {
        Scene scene{std::string(argv[1])};
        std::cout << "... start ray tracer" << std::endl;
        std::cout << "... write to file: " << outFilename << std::endl;

        std::unique_ptr<ImageType[]> img(new ImageType[size_t(imWidth) * imHeight * 3]);
        std::memset(img.get(), 0, size_t(imWidth) * imHeight * 3);
        const Float ASPECT_RATIO = (Float)imWidth / imHeight;
        const Float FOV = 60;
        Color background{0, 0.5, 0.5};
        Camera camera{Vec3f{0, 0, 0}, Vec3f{0, 1, 0}, Vec3f{0, 0, -1}, ASPECT_RATIO, FOV, imWidth, imHeight,
                      aa ? Camera::defaultAntiAliasingPattern() : Camera::noAntiAliasing()};
        camera.move(Vec3f{-0.5, 0, -0.25});

        unsigned int finPix{0}, thread_count{1};
        const auto t0 = std::chrono::steady_clock::now();
        if (devices.size() > 1) { scene.setDevices(devices); scene.multiScene(); }      // BVH build + upload, reported apart
        else scene.deviceScene(devices.empty() ? 0 : devices[0]);                       // from the frame time
        if (devices.size() == 1) scene.setDevices(devices);
        const auto t1 = std::chrono::steady_clock::now();
        std::string failure;
        std::thread renderThread([&] {                                                  // ref :711-713
            try { render(img.get(), 0, 0, imHeight, imWidth, camera, scene, background, finPix, thread_count); }
            catch (const std::exception& e) { failure = e.what(); thread_count = 0; }
        });
        // preview() (ref :195-262) without the window: the text it overlays, whenever finPix has moved
        const unsigned int sumPix = imWidth * imHeight;
        unsigned int lastFinPix = 0;
        double lastElapsed = 0;
        while (thread_count > 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(thread_count > 0 && finPix == 0 ? 1 : 200));
            const unsigned int curFinPix = finPix;
            if (curFinPix == lastFinPix || curFinPix >= sumPix) continue;
            const double curElapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            std::cout << (unsigned int)((float)curFinPix / sumPix * 100) << "%; t: " << (unsigned int)curElapsed << " s; "
                      << (unsigned int)((curFinPix - lastFinPix) / (curElapsed - lastElapsed)) << " pix/s" << std::endl;
            lastFinPix = curFinPix; lastElapsed = curElapsed;
        }
        renderThread.join();
        if (!failure.empty()) throw std::runtime_error(failure);
        const auto t2 = std::chrono::steady_clock::now();
        const double build_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        const double frame_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        std::cout << "... scene upload " << build_ms << " ms, frame " << frame_ms << " ms, "
                  << finPix / (frame_ms * 1e-3) << " pix/s" << std::endl;
        write_image(outFilename, img.get(), imWidth, imHeight);
    } catch (const std::exception& e) {
        std::cerr << "raytracer: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
