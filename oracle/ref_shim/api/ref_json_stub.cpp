// ref_json_stub.cpp -- libref reads the reference's TEXT scene format only.
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim).  src/sdl/scene_parser_json.cpp needs nlohmann::json (src/ext/json: an empty
// submodule in the snapshot); this file gives SceneParserJSON (declared in the reference's sdl/scene_parser_json.h) a body
// that refuses, so that sdl/scene_parser.cpp links.
#include <sdl/scene_parser_json.h>

namespace nlohmann {
class json {};
}// namespace nlohmann

namespace luisa::render {
SceneParserJSON::SceneParserJSON(SceneDesc &desc, const std::filesystem::path &path, const MacroMap &cli_macros) noexcept
    : _desc{desc}, _cli_macros{cli_macros}, _location{nullptr} { static_cast<void>(path); }
void SceneParserJSON::parse() const noexcept { LUISA_ERROR("libref: JSON scenes are not supported (nlohmann::json is absent)."); }
}// namespace luisa::render
