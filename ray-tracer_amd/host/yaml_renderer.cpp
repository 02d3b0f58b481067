// yaml_renderer <scene.yaml> [--width W --height H] [--strict] [--ppm] [--f64] [--repeat N] [-o out]
// Command-line twin of the reference's src/bin/yaml_renderer.rs:432-441: parse the scene,
// render it (on the GPU), write ./<stem>.png. The extra flags override the resolution the YAML
// declares, select the evaluation mode and the output format.
// An image file holds 8-bit channels (src/tuple.rs:226-237, :343-390), so that is all the program asks the
// device for: 3 bytes per pixel cross the PCIe link instead of 24.  --f64 fetches Canvas.pixels as well (and
// the file is then quantised on the host from them, as the reference does).  --repeat N renders the frame N more
// times into the same canvas and reports the median: the first frame of a process pays for the context's buffers
// and page-locked memory, the ones after it are what a host that renders frame after frame sees.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_scene.hpp"

int main(int argc, char** argv) {
    std::string path, out;
    int w = 0, h = 0, mode = RTAMD_MODE_FAST, repeat = 0;
    bool ppm = false, f64 = false;
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
        else if (a == "--f64")
            f64 = true;
        else if (a == "--repeat" && i + 1 < argc)
            repeat = std::atoi(argv[++i]);
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
        const int want = f64 ? RTAMD_WANT_F64 : RTAMD_WANT_U8;
        rt::Canvas image = scene->camera->render(scene->world, mode, &st, want);
        std::printf("Rendered in: %.3f ms (kernel %.3f ms, %llu rays)\n", st.render_ms, st.gpu.kernel_ms,
                    (unsigned long long)st.gpu.rays);
        if (repeat > 0) {
            std::vector<double> ms;
            for (int k = 0; k < repeat; k++) {
                scene->camera->render_into(image, scene->world, mode, &st);
                ms.push_back(st.render_ms);
            }
            std::sort(ms.begin(), ms.end());
            std::printf("Rendered again in: %.3f ms (median of %d, %s into %s memory; kernel %.3f ms)\n", ms[ms.size() / 2], repeat,
                        f64 ? "24 bytes per pixel" : "3 bytes per pixel",
                        (f64 ? image.pixels().page_locked() : image.rgb8().page_locked()) ? "page-locked" : "ordinary", st.gpu.kernel_ms);
        }
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
