// rtcli — command-line driver with the reference's flags, paths and log formats (Code/main.cpp:47-566), rendering
// through the C-ABI of librt_mi355x.so on one MI355X.  Host side only: argument parsing, config.json, scene file,
// timing log, BVH on/off sweep, PPM output.  Everything per-pixel happens in the library's HIP kernels.
//
// Reference flags (main.cpp:82-216): --no-bvh --time N --aa N --exposure F --shadows --glossy --parallel
//   --motion-blur F --fresnel --normals --bvh_testing --tonemap {reinhard|aces|filmic}; unknown arguments are ignored.
// Added (no counterpart in the reference, whose paths are hard-coded relative to Code/<build>/):
//   --scene PATH --config PATH --out PATH --texture-root DIR --device N --seed N --d0 --megakernel --pfm PATH
//   --bvh-builder {sah|sah_gpu|lbvh|median}   (rt_set_bvh_builder: host threads / built on the GPU / the reference's topology)
//   --gpus N   one process per GPU (devices --device .. --device + N - 1), forked here before anything touches a GPU: every
//              rank renders its interleaved row tiles, the slices are gathered on rank 0 over RCCL (rt_render_gathered),
//              and rank 0 -- this process -- prints and writes what the single-GPU run does
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <regex>
#include <string>
#include <vector>

#include <csignal>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/rt_mi355x.h"

namespace fs = std::filesystem;

static std::string timestamp() {  // main.cpp:35-45
    std::time_t t = std::time(nullptr);
    char buf[64];
    std::strftime(buf, sizeof(buf), "%Y-%m-%d_%H-%M-%S", std::localtime(&t));
    return buf;
}

struct Options {
    bool use_bvh = true, shadows = false, glossy = false, parallel = false, fresnel = false, normals = false;
    bool timing = false, bvh_testing = false, d0 = false, megakernel = false;
    int bvh_builder = RT_BVH_DEFAULT;
    int run_count = 1, tonemap = 0, device = 0, gpus = 1;
    bool have_aa = false, have_exposure = false, have_shutter = false;
    int aa = 1;
    double exposure = 1.0, shutter = 0.0;
    unsigned long long seed = 1;
    std::string pfm_path;
    std::string scene = "../../ASCII/scene.txt", config = "../config.json", out = "../../Output/scene_test.ppm", texture_root = "..";
    std::string all_args;
};

static bool need_value(int i, int argc, const char* flag, const char* what) {
    if (i + 1 < argc) return true;
    std::cerr << "Error: " << flag << " " << what << std::endl;
    std::exit(1);
}

int main(int argc, char* argv[]) {
    Options o;
    for (int i = 1; i < argc; ++i) o.all_args += std::string(argv[i]) + " ";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        try {
            if (a == "--no-bvh") { o.use_bvh = false; std::cout << "BVH disabled" << std::endl; }
            else if (a == "--time" && need_value(i, argc, "--time", "flag requires a number of runs.")) {
                o.run_count = std::max(1, std::stoi(argv[++i])); o.timing = true;
                std::cout << "Timing enabled: " << o.run_count << " runs." << std::endl;
            } else if (a == "--aa" && need_value(i, argc, "--aa", "flag requires a number of samples (e.g., --aa 16).")) {
                o.aa = std::stoi(argv[++i]); o.have_aa = true;
                std::cout << "Antialiasing enabled: " << o.aa << " samples/pixel." << std::endl;
            } else if (a == "--exposure" && need_value(i, argc, "--exposure", "flag requires a value (e.g., --exposure 0.5).")) {
                o.exposure = std::stod(argv[++i]); o.have_exposure = true;
                std::cout << "Exposure set to: " << o.exposure << std::endl;
            } else if (a == "--shadows") { o.shadows = true; std::cout << "Shadows enabled" << std::endl; }
            else if (a == "--glossy") { o.glossy = true; }
            else if (a == "--parallel") { o.parallel = true; std::cout << "Parallel rendering enabled" << std::endl; }
            else if (a == "--motion-blur" && need_value(i, argc, "--motion-blur", "requires a time value (e.g., --motion-blur 1.0).")) {
                o.shutter = std::stod(argv[++i]); o.have_shutter = true;
                std::cout << "Motion blur enabled. Shutter time: " << o.shutter << std::endl;
            } else if (a == "--fresnel") { o.fresnel = true; std::cout << "Fresnel effect enabled" << std::endl; }
            else if (a == "--normals") { o.normals = true; std::cout << "Debug: Rendering surface normals." << std::endl; }
            else if (a == "--bvh_testing") { o.bvh_testing = true; std::cout << "BVH testing mode enabled." << std::endl; }
            else if (a == "--tonemap" && need_value(i, argc, "--tonemap", "requires a mode (reinhard, aces, filmic).")) {
                std::string m = argv[++i];
                std::transform(m.begin(), m.end(), m.begin(), ::tolower);
                if (m == "reinhard") o.tonemap = RT_TONEMAP_REINHARD;
                else if (m == "aces") o.tonemap = RT_TONEMAP_ACES;
                else if (m == "filmic") o.tonemap = RT_TONEMAP_FILMIC;
                else { std::cerr << "Unknown tonemap mode: " << m << " (defaulting to none)" << std::endl; o.tonemap = RT_TONEMAP_NONE; }
                std::cout << "Tone mapping set to: " << m << std::endl;
            }
            else if (a == "--scene" && i + 1 < argc) o.scene = argv[++i];
            else if (a == "--config" && i + 1 < argc) o.config = argv[++i];
            else if (a == "--out" && i + 1 < argc) o.out = argv[++i];
            else if (a == "--texture-root" && i + 1 < argc) o.texture_root = argv[++i];
            else if (a == "--device" && i + 1 < argc) o.device = std::stoi(argv[++i]);
            else if (a == "--gpus" && i + 1 < argc) o.gpus = std::max(1, std::stoi(argv[++i]));
            else if (a == "--seed" && i + 1 < argc) o.seed = std::stoull(argv[++i]);
            else if (a == "--d0") o.d0 = true;
            else if (a == "--megakernel") o.megakernel = true;
            else if (a == "--bvh-builder" && i + 1 < argc) {
                const std::string b = argv[++i];
                o.bvh_builder = b == "lbvh" ? RT_BVH_LBVH : b == "median" ? RT_BVH_MEDIAN : b == "sah" ? RT_BVH_SAH : b == "sah_gpu" ? RT_BVH_SAH_GPU : RT_BVH_DEFAULT;
            }
            else if (a == "--pfm" && i + 1 < argc) o.pfm_path = argv[++i];   // also write the linear (pre tone map) frame
        } catch (const std::exception&) {
            std::cerr << "Error: Invalid value for " << a << " flag." << std::endl;
            return 1;
        }
    }

    rt_config_t cfg = nullptr;
    if (rt_config_load(o.config.c_str(), &cfg) != RT_OK) std::cerr << rt_last_error() << std::endl;  // warning only, config.cpp:9-12
    else std::cout << "Configuration loaded from " << o.config << std::endl;
    RtRenderParams p;
    rt_params_from_config(cfg, &p);
    if (o.have_aa) p.samples_per_pixel = o.aa;
    if (o.have_exposure) p.exposure = o.exposure;
    const double shutter = o.have_shutter ? o.shutter : rt_config_get_double(cfg, "image.shutter_time", 0.0);  // main.cpp:62
    p.shadows = o.shadows; p.fresnel = o.fresnel; p.normals = o.normals; p.tonemap = o.tonemap;
    if (o.glossy) {
        p.glossy_samples = rt_config_get_int(cfg, "render.glossy_samples", 8);  // main.cpp:152
        std::cout << "Glossy flag present: Enabled with " << p.glossy_samples << " samples." << std::endl;
    }
    p.rng_mode = o.d0 ? RT_RNG_D0 : RT_RNG_SEEDED;
    p.seed = o.seed;
    p.schedule = o.megakernel ? RT_SCHEDULE_MEGAKERNEL : RT_SCHEDULE_WAVEFRONT;

    (void)rt_set_bvh_builder(o.bvh_builder);
    // --gpus N: ranks 1 .. N-1 are forked here, before the first HIP call of this process.  Each reports over a socket pair
    // whether it got its GPU, and only when all have does rank 0 create the RCCL id and hand it out (a communicator whose
    // ranks never all arrive would wait for ever).  All ranks then run the same sequence of renders (rt_render_gathered is
    // collective); only rank 0 prints, times and writes files.
    int rank = 0;
    std::vector<pid_t> children;
    std::vector<int> id_pipes;
    int my_pipe = -1;
    std::signal(SIGPIPE, SIG_IGN);
    for (int r = 1; r < o.gpus; ++r) {
        int fd[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, fd) != 0) { std::cerr << "An error occurred: socketpair() failed" << std::endl; return 1; }
        std::cout.flush();
        const pid_t pid = fork();
        if (pid < 0) { std::cerr << "An error occurred: fork() failed" << std::endl; return 1; }
        if (pid == 0) {
            rank = r;
            close(fd[1]);
            for (int w : id_pipes) close(w);
            id_pipes.clear();
            children.clear();
            my_pipe = fd[0];
            std::cout.setstate(std::ios_base::failbit);   // one voice: rank 0's
            break;
        }
        close(fd[0]);
        children.push_back(pid);
        id_pipes.push_back(fd[1]);
    }
    auto finish = [&](int code) {   // rank 0 waits for the others; any failure fails the run
        for (pid_t c : children) {
            int status = 0;
            if (waitpid(c, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) code = code ? code : 1;
        }
        return code;
    };
    const bool have_gpu = rt_init(o.device + rank) == RT_OK;
    if (!have_gpu) std::cerr << "An error occurred: " << (o.gpus > 1 ? "rank " + std::to_string(rank) + ": " : std::string()) << rt_last_error() << std::endl;
    if (rank > 0) {
        const char ready = have_gpu ? 1 : 0;
        if (write(my_pipe, &ready, 1) != 1 || !have_gpu) return 1;
    } else if (!have_gpu) {
        for (int w : id_pipes) close(w);   // the other ranks see end of file and leave
        return finish(1);
    }
    if (o.gpus > 1) {
        unsigned char id[RT_COMM_ID_BYTES];
        bool ok = true;
        if (rank == 0) {
            for (int w : id_pipes) {
                char ready = 0;
                if (read(w, &ready, 1) != 1 || ready != 1) ok = false;
            }
            if (!ok) {
                std::cerr << "An error occurred: --gpus " << o.gpus << ": not every rank got its GPU; nothing rendered" << std::endl;
                for (int w : id_pipes) close(w);
                return finish(1);
            }
            ok = rt_comm_unique_id(id) == RT_OK;
            for (int w : id_pipes) {
                if (ok) ok = write(w, id, sizeof(id)) == (ssize_t)sizeof(id);
                close(w);
            }
        } else {
            size_t got = 0;
            while (got < sizeof(id)) {
                const ssize_t n = read(my_pipe, id + got, sizeof(id) - got);
                if (n <= 0) break;
                got += (size_t)n;
            }
            close(my_pipe);
            ok = got == sizeof(id);
        }
        if (!ok || rt_comm_init(rank, o.gpus, id) != RT_OK) {
            std::cerr << "An error occurred: rank " << rank << ": " << (ok ? rt_last_error() : "no RCCL id from rank 0") << std::endl;
            return finish(1);
        }
    }

    // render_scene_func of main.cpp:229-359: returns elapsed seconds of load + build + render (the file write is outside)
    auto render_scene = [&](const std::string& scene_path, bool use_bvh, const std::string& output_path, double& seconds) -> bool {
        auto t0 = std::chrono::high_resolution_clock::now();
        std::cout << "Loading scene: " << scene_path << (use_bvh ? " [BVH ON]" : " [BVH OFF]") << std::endl;
        rt_scene_file_t sf = nullptr;
        if (rt_scene_file_load(scene_path.c_str(), o.texture_root.c_str(), cfg, shutter, &sf) != RT_OK) {
            std::cerr << "An error occurred: " << rt_last_error() << std::endl;
            return false;
        }
        const RtSceneDesc* d = rt_scene_file_desc(sf);
        rt_scene_t sc = nullptr;
        if (d->n_shapes) std::cout << (use_bvh ? "Building BVH..." : "BVH build skipped.") << std::endl;
        if (rt_scene_create(d, &sc) != RT_OK) {
            std::cerr << "An error occurred: " << rt_last_error() << std::endl;
            rt_scene_file_destroy(sf);
            return false;
        }
        const int w = d->camera.resolution_x, h = d->camera.resolution_y;
        std::cout << "Rendering scene (" << w << "x" << h << ") with " << p.samples_per_pixel << " samples per pixel..." << std::endl;
        std::cout << "Device: HIP device " << o.device << (o.gpus > 1 ? " .. " + std::to_string(o.device + o.gpus - 1) + ", interleaved row tiles, RCCL gather" : std::string())
                  << " (" << rt_version() << ")" << std::endl;
        std::vector<uint8_t> img(static_cast<size_t>(w) * h * 3);
        RtRenderParams q = p;
        q.use_bvh = use_bvh ? 1 : 0;
        RtStats st{};
        bool ok = (o.gpus > 1 ? rt_render_gathered(sc, &q, img.data(), &st) : rt_render(sc, &q, img.data(), &st)) == RT_OK;
        if (!ok) std::cerr << "An error occurred: " << rt_last_error() << std::endl;
        seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        if (ok) {
            std::cout << "Rendering: 100% [" << h << "/" << h << "]" << std::endl;
            std::cout << "Rays: " << st.rays << ", device " << st.kernel_ms << " ms (" << (st.kernel_ms > 0 ? st.rays / st.kernel_ms / 1e3 : 0.0)
                      << " Mrays/s)" << std::endl;
            if (!output_path.empty() && rank == 0) {
                if (rt_write_ppm(output_path.c_str(), w, h, img.data()) != RT_OK) {
                    std::cerr << "An error occurred: " << rt_last_error() << std::endl;
                    ok = false;
                } else {
                    std::cout << "Image saved to '" << output_path << "'." << std::endl;
                }
            }
            if (ok && !o.pfm_path.empty() && o.gpus == 1) {   // linear radiance of the same frame (same seed: same samples), before tone mapping
                std::vector<double> lin(static_cast<size_t>(w) * h * 3);
                RtRenderParams ql = q;
                ql.schedule = RT_SCHEDULE_WAVEFRONT;
                if (rt_render_linear(sc, &ql, lin.data()) != RT_OK || rt_write_pfm(o.pfm_path.c_str(), w, h, lin.data()) != RT_OK) {
                    std::cerr << "An error occurred: " << rt_last_error() << std::endl;
                    ok = false;
                } else {
                    std::cout << "Linear frame saved to '" << o.pfm_path << "'." << std::endl;
                }
            }
        }
        rt_scene_destroy(sc);
        rt_scene_file_destroy(sf);
        return ok;
    };

    int rc = 0;
    if (o.bvh_testing) {  // main.cpp:362-479
        const std::string out_dir = "../../Output/testing/" + timestamp();
        const std::string src_dir = "../../ASCII/BVH_tests";
        try {
            if (rank == 0) {
                fs::create_directories(out_dir);
                std::cout << "Output directory created: " << out_dir << std::endl;
                if (fs::exists(src_dir)) fs::copy(src_dir, out_dir + "/BVH_tests", fs::copy_options::recursive | fs::copy_options::overwrite_existing);
                else std::cerr << "Warning: BVH_tests directory not found at " << src_dir << std::endl;
                std::ofstream(out_dir + "/flags.txt") << "Flags used: " << o.all_args << std::endl;
            }
            struct Item { int x; std::string path; };
            std::vector<Item> items;
            std::regex re(R"(scene_?(\d+)\.txt)", std::regex::icase);
            if (fs::exists(src_dir))
                for (const auto& e : fs::directory_iterator(src_dir)) {
                    if (!e.is_regular_file()) continue;
                    std::string name = e.path().filename().string();
                    std::smatch m;
                    if (std::regex_search(name, m, re)) items.push_back({std::stoi(m[1]), e.path().string()});
                    else if (name == "scene.txt") items.push_back({0, e.path().string()});
                }
            if (items.empty()) {
                std::cerr << "No matching scene files found! Please check filenames in " << src_dir << std::endl;
                return finish(1);
            }
            std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.x < b.x; });
            std::ofstream f_bvh, f_no;
            if (rank == 0) { f_bvh.open(out_dir + "/bvh_test.txt"); f_no.open(out_dir + "/no_bvh_test.txt"); }
            for (const auto& it : items) {
                std::cout << "\n--- Testing Scene X=" << it.x << " ---" << std::endl;
                for (int mode = 0; mode < 2; ++mode) {
                    double total = 0;
                    for (int k = 0; k < 3; ++k) {
                        std::cout << "Run " << (k + 1) << "/3 " << (mode == 0 ? "[BVH ON]" : "[BVH OFF]") << std::endl;
                        std::string img = k == 0 ? out_dir + (mode == 0 ? "/bvh_" : "/no_bvh_") + std::to_string(it.x) + ".ppm" : "";
                        double s = 0;
                        if (!render_scene(it.path, mode == 0, img, s)) return finish(1);
                        total += s;
                    }
                    if (rank == 0) (mode == 0 ? f_bvh : f_no) << total / 3.0 << " " << it.x << std::endl;
                }
            }
            std::cout << "\nBVH Testing Complete. Results saved to " << out_dir << std::endl;
        } catch (const std::exception& e) {
            std::cerr << "Error during BVH testing: " << e.what() << std::endl;
            return finish(1);
        }
    } else {  // main.cpp:481-566
        std::string ts, out_dir;
        if (o.timing) {
            ts = timestamp();
            out_dir = "../../Output/testing/" + ts;
            if (rank == 0) try {
                fs::create_directories(out_dir);
                std::cout << "Output directory created: " << out_dir << std::endl;
                fs::copy(o.scene, out_dir + "/scene.txt", fs::copy_options::overwrite_existing);
                std::cout << "Saved copy of scene.txt to output folder." << std::endl;
            } catch (const std::exception& e) {
                std::cerr << "Error creating output directory: " << e.what() << std::endl;
                return finish(1);
            }
        }
        std::vector<std::pair<double, std::string>> runs;
        for (int i = 0; i < o.run_count && rc == 0; ++i) {
            if (o.timing) std::cout << "\n--- Starting Run " << (i + 1) << " of " << o.run_count << " ---" << std::endl;
            std::string file = o.timing ? out_dir + "/output_" + ts + "_" + std::to_string(i + 1) + ".ppm" : o.out;
            double s = 0;
            if (!render_scene(o.scene, o.use_bvh, file, s)) { rc = 1; break; }
            if (o.timing) {
                std::cout << "Run " << (i + 1) << " completed in " << s << " seconds." << std::endl;
                runs.emplace_back(s, file);
            } else {
                std::cout << "Render complete! Image saved to '" << file << "'." << std::endl;
            }
        }
        if (o.timing && rc == 0 && rank == 0) {  // main.cpp:546-561
            std::ofstream log(out_dir + "/timing_log.txt");
            if (log.is_open()) {
                log << "args: [" << o.all_args << "]\n";
                for (const auto& r : runs) log << "[" << r.first << ", " << r.second << "]\n";
                std::cout << "Timing log saved to: " << out_dir + "/timing_log.txt" << std::endl;
            } else {
                std::cerr << "Failed to write log file to: " << out_dir + "/timing_log.txt" << std::endl;
            }
        }
    }
    rt_config_destroy(cfg);
    if (o.gpus > 1) (void)rt_comm_destroy();
    rt_shutdown();
    return finish(rc);
}
