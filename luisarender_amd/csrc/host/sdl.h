// sdl.h — scene description layer: typed property bags (SceneNodeDesc), the scene
// description (SceneDesc) and the two parsers (text ".luisa" grammar and JSON).
//
// Re-creates the contract of the reference's src/sdl/ (grammar: scene_parser.cpp:71-451,
// JSON: scene_parser_json.cpp:27-195, tags: scene_node_tag.cpp:15-45, getters:
// scene_node_desc.h:212-380) on plain C++17 — the LuisaCompute/fast_float/nlohmann
// headers the reference is written against are not available.
#pragma once
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

namespace lr {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void log_info(const std::string &msg);
void log_warning(const std::string &msg);
void set_log_level(int level);// 0 = silent, 1 = warnings, 2 = info

enum class Tag : uint32_t {
    ROOT, INTERNAL, CAMERA, SHAPE, SURFACE, LIGHT, TRANSFORM, FILM, FILTER, SAMPLER,
    INTEGRATOR, LIGHT_SAMPLER, ENVIRONMENT, TEXTURE, TEXTURE_MAPPING, SPECTRUM,
    MEDIUM, PHASE_FUNCTION, DECLARATION
};
const char *tag_description(Tag tag);
Tag parse_tag(std::string desc);// unknown -> Tag::ROOT (scene_node_tag.cpp:47-50)

class NodeDesc {
public:
    using number_list = std::vector<double>;
    using bool_list = std::vector<bool>;
    using string_list = std::vector<std::string>;
    using node_list = std::vector<const NodeDesc *>;
    using value_list = std::variant<number_list, bool_list, string_list, node_list>;

private:
    std::string _identifier;
    Tag _tag;
    std::string _impl;        // lower-cased (scene_node_desc.cpp:28-30)
    std::string _source_file; // file that defined the node (for relative paths)
    uint32_t _line{0};
    const NodeDesc *_base{nullptr};
    std::map<std::string, value_list> _props;
    std::vector<std::unique_ptr<NodeDesc>> _internal;

    const value_list *_find(const std::string &name) const;
    template<typename L> const L *_raw(const std::string &name) const;

public:
    NodeDesc(std::string id, Tag tag) : _identifier{std::move(id)}, _tag{tag} {}
    const std::string &identifier() const { return _identifier; }
    Tag tag() const { return _tag; }
    const std::string &impl_type() const { return _impl; }
    const std::string &source_file() const { return _source_file; }
    std::string location() const;
    bool is_defined() const { return _tag != Tag::DECLARATION && !_impl.empty(); }
    void define(Tag tag, const std::string &impl, const std::string &file, uint32_t line,
                const NodeDesc *base = nullptr);
    NodeDesc *define_internal(const std::string &impl, const std::string &file, uint32_t line,
                              const NodeDesc *base = nullptr);
    void add_property(const std::string &name, value_list v);
    bool has_property(const std::string &name) const;
    bool is_string_property(const std::string &name) const;// defined (here or in the base) as a string list
    const std::map<std::string, value_list> &properties() const { return _props; }

    // getters; *_opt return nullopt when absent / wrong list type / too few values
    std::optional<double> number_opt(const std::string &name) const;
    std::optional<std::vector<double>> numbers_opt(const std::string &name) const;
    std::optional<std::vector<double>> vector_opt(const std::string &name, size_t n) const;
    std::optional<bool> bool_opt(const std::string &name) const;
    std::optional<std::string> string_opt(const std::string &name) const;
    std::optional<std::vector<std::string>> strings_opt(const std::string &name) const;
    const NodeDesc *node_or_null(const std::string &name) const;
    std::optional<node_list> nodes_opt(const std::string &name) const;

    float float_or(const std::string &name, float dv) const;
    uint32_t uint_or(const std::string &name, uint32_t dv) const;// errors on non-integral
    bool bool_or(const std::string &name, bool dv) const;
    std::string string_or(const std::string &name, const std::string &dv = {}) const;
    std::string path_or(const std::string &name, const std::string &dv = {}) const;
    std::vector<float> float_list_or_empty(const std::string &name) const;
    std::vector<float> float_list(const std::string &name) const;  // required
    std::vector<uint32_t> uint_list(const std::string &name) const;// required
    const NodeDesc *node(const std::string &name) const;           // required
    node_list node_list_required(const std::string &name) const;
    node_list node_list_or_empty(const std::string &name) const;

    // shared default nodes (scene_node_desc.cpp:51-72)
    static const NodeDesc *shared_default(Tag tag, std::string impl);
};

class SceneDesc {
    std::unordered_map<std::string, std::unique_ptr<NodeDesc>> _globals;
    NodeDesc _root{"render", Tag::ROOT};
    std::deque<std::string> _files;

public:
    static constexpr const char *root_node_identifier = "render";
    NodeDesc *root() { return &_root; }
    const NodeDesc *root() const { return &_root; }
    const NodeDesc *node(const std::string &id) const;
    const NodeDesc *reference(const std::string &id);// forward declaration allowed
    NodeDesc *define(const std::string &id, Tag tag, const std::string &impl,
                     const std::string &file, uint32_t line, const NodeDesc *base);
    NodeDesc *define_root(const std::string &file, uint32_t line);
    const std::string &register_file(const std::string &path);
    void validate() const;// every referenced node must be defined
    const std::unordered_map<std::string, std::unique_ptr<NodeDesc>> &globals() const { return _globals; }
};

using MacroMap = std::unordered_map<std::string, std::string>;

// SceneParser::parse (scene_parser.cpp:401-407): dispatches on the ".json" extension.
std::unique_ptr<SceneDesc> parse_scene_file(const std::string &path, const MacroMap &cli_macros);
// parse from memory (tests, generated scenes); `virtual_path` anchors relative imports
std::unique_ptr<SceneDesc> parse_scene_string(const std::string &source, const std::string &virtual_path,
                                              const MacroMap &cli_macros, bool json = false);

}// namespace lr
