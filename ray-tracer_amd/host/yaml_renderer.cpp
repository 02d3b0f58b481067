// yaml_renderer <scene.yaml> [--width W --height H] [--strict] [--ppm] [-o out]
// Command-line twin of the reference's src/bin/yaml_renderer.rs:432-441: parse the scene,
// render it (on the GPU), write ./<stem>.png. The extra flags override the resolution the YAML
// declares, select the evaluation mode and the output format.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "rt_scene.hpp"

int main(int argc, char** argv) {
    std::string path, out;
    int w = 0, h = 0, mode = RTAMD_MODE_FAST;
    bool ppm = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--width" && i + 1 < argc)
            w = std::atoi(argv[++i]);
        else if (a == "--height" && i + 1 < argc)
            h = std::atoi(argv[++i]);
        else if (a == "--strict")
            mode = RTAMD_MODE_STRICT;
        else if (a == "--ppm")
            ppm = true;
        else if (a == "-o" && i + 1 < argc)
            out = argv[++i];
        else if (path.empty())
            path = a;
        else {
            std::fprintf(stderr, "unexpected argument %s\n", a.c_str());
            return 2;
        }
    }
    if (path.empty()) {
        std::fprintf(stderr, "no yaml file provided\n");
        return 2;
    }
    try {
        auto t0 = std::chrono::steady_clock::now();
        auto scene = rt::load_yaml_scene(path);
        if (!scene->camera) throw rt::Error("no camera set");
        if (w > 0 && h > 0) {
            auto c = std::make_unique<rt::Camera>((size_t)w, (size_t)h, scene->camera->field_of_view);
            c->transform = scene->camera->transform;
            scene->camera = std::move(c);
        }
        std::printf("Parsed in: %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        std::printf("Rendering\n");
        rt::RenderStats st;
        rt::Canvas image = scene->camera->render(scene->world, mode, &st);
        std::printf("Rendered in: %.3f ms (kernel %.3f ms, %llu rays)\n", st.render_ms, st.gpu.kernel_ms,
                    (unsigned long long)st.gpu.rays);
        if (out.empty()) {
            std::string base = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
            size_t dot = base.find_last_of('.');
            if (dot != std::string::npos) base.resize(dot);
            out = "./" + base + (ppm ? ".ppm" : ".png");
        }
        if (ppm)
            image.save_ppm(out);
        else
            image.save_png(out);
    } catch (std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
