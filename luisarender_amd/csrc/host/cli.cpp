// cli.cpp — `luisa-render-cli`, process-level contract of the reference's src/apps/cli.cpp:
//   luisa-render-cli -b <backend> [-d <index>] [-D key=value]... <scene-file>
// -D macros override scene `define`s (cli.cpp:105-152); unknown options warn; a missing scene file
// prints the help and exits with -1 (cli.cpp:59-99); each camera's image goes to its `file` property.
// The only backend is "hip" (MI355X / gfx950).
#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <mutex>
#include <unordered_map>

#include "luisa_render_shim.h"



namespace {

void print_help() {
    std::cout << "Usage:\n  luisa-render-cli [OPTION...] <file>\n\n"
                 "  -b, --backend <backend>    Compute backend name (hip)\n"
                 "  -d, --device <index>       Compute device index (default: -1); a list `0,1,2,3` (or LR_DEVICES) shards the frame over several GPUs\n"
                 "      --scene <file>         Path to scene description file\n"
                 "  -D, --define <key>=<value> Parameter definitions to override scene description macros.\n"
                 "  -h, --help                 Display this help message\n";
}

std::filesystem::path current_exe_directory() {
    char buf[4096];
    auto n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    if (n <= 0) { return std::filesystem::current_path(); }
    buf[n] = '\0';
    return std::filesystem::path{buf}.parent_path();
}

}// namespace

int main(int argc, char *argv[]) {
    lr::set_log_level(2);// log_level_info(), cli.cpp:156
    lr::MacroMap macros;
    std::string backend, scene_file;
    auto device_index = -1;
    std::string device_list;// `-d 2` as in the reference, or `-d 0,1,2,3`: one frame sharded over several GPUs
    std::vector<std::string> unknown;
    auto parse_macro = [&](std::string_view d) {
        auto p = d.find('=');
        if (p == std::string_view::npos) {
            lr::log_warning("Invalid definition: " + std::string{d});
            return;
        }
        auto key = std::string{d.substr(0, p)}, value = std::string{d.substr(p + 1)};
        if (auto it = macros.find(key); it != macros.end()) {
            lr::log_warning("Duplicate definition: " + key + " = '" + value + "'. Ignoring the previous one.");
        }
        macros[key] = value;
    };
    for (auto i = 1; i < argc; i++) {
        std::string_view arg{argv[i]};
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        if (arg == "-D" || arg == "--define") {
            if (auto v = next()) { parse_macro(v); } else { lr::log_warning("Missing definition after " + std::string{arg} + "."); }
        } else if (arg.rfind("-D", 0) == 0 && arg.size() > 2u) {
            parse_macro(arg.substr(2));
        } else if (arg == "-b" || arg == "--backend") {
            if (auto v = next()) { backend = v; }
        } else if (arg.rfind("--backend=", 0) == 0) {
            backend = arg.substr(10);
        } else if (arg == "-d" || arg == "--device") {
            if (auto v = next()) { device_list = v; }
        } else if (arg.rfind("--device=", 0) == 0) {
            device_list = std::string{arg.substr(9)};
        } else if (arg == "--scene") {
            if (auto v = next()) { scene_file = v; }
        } else if (arg == "-h" || arg == "--help") {
            print_help();
            return 0;
        } else if (!arg.empty() && arg[0] == '-') {
            unknown.emplace_back(arg);
        } else if (scene_file.empty()) {
            scene_file = arg;
        } else {
            unknown.emplace_back(arg);
        }
    }
    for (auto &[k, v] : macros) { lr::log_info("Found CLI Macro: " + k + " = " + v); }
    if (scene_file.empty()) {
        lr::log_warning("Scene file not specified.");
        print_help();
        return -1;
    }
    if (!unknown.empty()) {
        std::string opts{unknown.front()};
        for (size_t i = 1; i < unknown.size(); i++) { opts.append("; ").append(unknown[i]); }
        lr::log_warning("Unrecognized options: " + opts);
    }
    if (backend.empty()) {
        lr::log_warning("Backend not specified (-b); using 'hip'.");
        backend = "hip";
    } else if (backend != "hip" && backend != "HIP") {
        lr::log_warning("Backend '" + backend + "' is not available in this build; using 'hip' (MI355X / gfx950).");
    }
    try {
        if (device_list.empty()) {
            if (auto env = std::getenv("LR_DEVICES")) { device_list = env; }
        }
        std::vector<int> devices;
        for (size_t b = 0; b < device_list.size();) {
            auto e = device_list.find(',', b);
            if (e == std::string::npos) { e = device_list.size(); }
            if (e > b) { devices.emplace_back(std::atoi(device_list.substr(b, e - b).c_str())); }
            b = e + 1;
        }
        if (!devices.empty()) { device_index = devices.front(); }
        luisa::compute::Device device{"hip", device_index, devices.size() > 1u ? devices : std::vector<int>{}};
        auto t0 = std::chrono::steady_clock::now();
        auto desc = lr::parse_scene_file(scene_file, macros);
        auto parse_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        lr::log_info("Parsed scene description file '" + scene_file + "' in " + std::to_string(parse_ms) + " ms.");
        auto scene = luisa::render::Scene::create(current_exe_directory(), std::move(desc));
        luisa::compute::Stream stream{&device};
        auto pipeline = luisa::render::Pipeline::create(device, stream, *scene);
        pipeline->render(stream);
        stream.synchronize();
    } catch (const std::exception &e) {// LUISA_ERROR: log and abort
        std::fprintf(stderr, "[error] %s\n", e.what());
        return 1;
    }
    return 0;
}
