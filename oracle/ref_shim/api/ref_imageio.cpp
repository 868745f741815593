// ref_imageio.cpp -- image file IO behind the reference's util/imageio.h interface.
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim).  src/util/imageio.cpp sits on stb_image and tinyexr (src/ext: empty submodules in
// the snapshot) and cannot be compiled; this file implements the same declarations for what the parity tests feed libref:
//   * LoadedImage::load of `.pfm` (PF / Pf, rows stored bottom-up -> row 0 = top, RGB promoted to RGBA with alpha 1, grey kept
//     as one channel): the convention of luisarender_amd/csrc/host/image_io.cpp, so both sides see the same texels;
//   * save_image: keeps the float RGBA film in memory (ref_api.cpp hands it to the test) and, for debugging, writes it as `.npy`.
#include <util/imageio.h>
#include <util/half.h>
#include <core/logging.h>

#include <fstream>
#include <map>
#include <mutex>

namespace luisa::render {

LoadedImage::LoadedImage(void *pixels, storage_type storage, uint2 resolution, luisa::function<void(void *)> deleter) noexcept
    : _pixels{pixels}, _resolution{resolution}, _storage{storage}, _deleter{std::move(deleter)} {}
LoadedImage::~LoadedImage() noexcept { _destroy(); }
void LoadedImage::_destroy() noexcept {
    if (_pixels != nullptr && _deleter) { _deleter(_pixels); }
    _pixels = nullptr;
}
LoadedImage::LoadedImage(LoadedImage &&o) noexcept
    : _pixels{o._pixels}, _resolution{o._resolution}, _storage{o._storage}, _deleter{std::move(o._deleter)} { o._pixels = nullptr; }
LoadedImage &LoadedImage::operator=(LoadedImage &&rhs) noexcept {
    if (this != &rhs) {
        _destroy();
        _pixels = rhs._pixels, _resolution = rhs._resolution, _storage = rhs._storage, _deleter = std::move(rhs._deleter);
        rhs._pixels = nullptr;
    }
    return *this;
}
LoadedImage LoadedImage::create(uint2 resolution, storage_type storage) noexcept {
    auto bytes = compute::pixel_storage_size(storage, make_uint3(resolution, 1u));
    return {std::calloc(bytes, 1u), storage, resolution, [](void *p) noexcept { std::free(p); }};
}

LoadedImage LoadedImage::load(const std::filesystem::path &path) noexcept {
    auto ext = path.extension().string();
    for (auto &c : ext) { c = static_cast<char>(tolower(c)); }
    if (ext != ".pfm") { LUISA_ERROR("libref: only .pfm images can be loaded (stb_image / tinyexr are absent): '{}'.", path.string()); }
    std::ifstream f{path, std::ios::binary};
    if (!f) { LUISA_ERROR("libref: cannot open '{}'.", path.string()); }
    std::string magic;
    int w = 0, h = 0;
    double scale = 0.;
    f >> magic >> w >> h >> scale;
    f.get();
    auto channels = magic == "PF" ? 3u : 1u;
    if ((magic != "PF" && magic != "Pf") || w <= 0 || h <= 0 || scale >= 0.) {// little-endian files only (negative scale)
        LUISA_ERROR("libref: unsupported PFM header in '{}'.", path.string());
    }
    std::vector<float> rows(static_cast<size_t>(w) * h * channels);
    f.read(reinterpret_cast<char *>(rows.data()), static_cast<std::streamsize>(rows.size() * sizeof(float)));
    auto out_channels = channels == 3u ? 4u : 1u;
    auto image = create(make_uint2(static_cast<uint>(w), static_cast<uint>(h)),
                        out_channels == 4u ? storage_type::FLOAT4 : storage_type::FLOAT1);
    auto dst = static_cast<float *>(image.pixels());
    for (auto y = 0; y < h; y++) {
        auto src = rows.data() + static_cast<size_t>(h - 1 - y) * w * channels;
        for (auto x = 0; x < w; x++) {
            auto p = dst + (static_cast<size_t>(y) * w + x) * out_channels;
            if (channels == 3u) { p[0] = src[x * 3], p[1] = src[x * 3 + 1], p[2] = src[x * 3 + 2], p[3] = 1.f; }
            else { p[0] = src[x]; }
        }
    }
    return image;
}
LoadedImage LoadedImage::load(const std::filesystem::path &path, storage_type) noexcept { return load(path); }
LoadedImage::storage_type LoadedImage::parse_storage(const std::filesystem::path &path) noexcept { return load(path).pixel_storage(); }

}// namespace luisa::render

namespace libref {
std::map<std::string, std::vector<float>> &saved_images() noexcept {
    static std::map<std::string, std::vector<float>> images;
    return images;
}
}// namespace libref

namespace luisa::render {

void save_image(std::filesystem::path path, const float *pixels, uint2 resolution, uint components) noexcept {
    auto n = static_cast<size_t>(resolution.x) * resolution.y * components;
    libref::saved_images()[path.string()] = std::vector<float>(pixels, pixels + n);
    if (std::getenv("LIBREF_WRITE_IMAGES") != nullptr) {
        auto npy = path;
        npy.replace_extension(".npy");
        std::ofstream f{npy, std::ios::binary};
        auto header = luisa::format("{{'descr': '<f4', 'fortran_order': False, 'shape': ({}, {}, {}), }}", resolution.y, resolution.x, components);
        while ((10u + header.size() + 1u) % 64u != 0u) { header.push_back(' '); }
        header.push_back('\n');
        f.write("\x93NUMPY\x01\x00", 8);
        auto len = static_cast<uint16_t>(header.size());
        f.write(reinterpret_cast<const char *>(&len), 2);
        f.write(header.data(), static_cast<std::streamsize>(header.size()));
        f.write(reinterpret_cast<const char *>(pixels), static_cast<std::streamsize>(n * sizeof(float)));
    }
}
void save_image(std::filesystem::path, const uint8_t *, uint2, uint) noexcept { LUISA_ERROR("libref: LDR image output is not supported."); }

}// namespace luisa::render
